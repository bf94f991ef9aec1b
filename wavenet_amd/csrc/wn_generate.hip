// Synthetic training examples produced on the device, a batch at a time, so that an end-to-end run
// does not ship its inputs over PCIe.  Counter-based: every value is a pure function of (seed,
// example number, item) through splitmix64, exactly as in wavenet_amd/generators.py and
// wavenet_amd/host/src/GeneratorBase.cxx, so host and device produce the same bits.
//
// Reference: the generators of include/Wavenet/Generators.h (Needle :50-62, Uniform :87-96, HepMC
// eta-phi histogram :377-393) behind GeneratorBase::next (include/Wavenet/GeneratorBase.h:32-103);
// those reseed from entropy on every reset and cannot be reproduced, which is why these exist.
#include <hip/hip_runtime.h>

#include <string.h>

#include <string>

#include "wavenet_hip.h"
#include "wn_internal.h"

namespace {

using wn::fail;
typedef unsigned long long u64;

__host__ __device__ inline u64 splitmix64(u64 x) {
    x += 0x9E3779B97F4A7C15ull;
    u64 z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__host__ __device__ inline u64 mix(u64 h, u64 counter) { return splitmix64(h ^ (counter * 0xD6E8FEB86659FD93ull)); }
__host__ __device__ inline double unit(u64 h) { return (double)(h >> 11) * (1.0 / 9007199254740992.0); }

// element i of the device layout (column-major: r + c * R) is item r * C + c of the example
__device__ inline u64 item_of(int i, int lgR, int C) { return (u64)(i & ((1 << lgR) - 1)) * (u64)C + (u64)(i >> lgR); }

__global__ __launch_bounds__(256) void kg_uniform(u64 seed, long long first, int lgR, int C, int n, double* X) {
    const u64 he = mix(splitmix64(seed), (u64)(first + blockIdx.y));
    double* Xe = X + (size_t)blockIdx.y * n;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) Xe[i] = unit(mix(he, item_of(i, lgR, C)));
}

// One workgroup per example: pixels below the threshold are zeroed; an image left empty keeps one pixel.
__global__ __launch_bounds__(256) void kg_needle(u64 seed, long long first, int lgR, int C, int n, double threshold, double* X) {
    __shared__ int any;
    const u64 e = (u64)(first + blockIdx.x);
    const u64 he = mix(splitmix64(seed), e);
    double* Xe = X + (size_t)blockIdx.x * n;
    if (threadIdx.x == 0) any = 0;
    __syncthreads();
    bool mine = false;
    for (int i = threadIdx.x; i < n; i += 256) {
        double v = unit(mix(he, item_of(i, lgR, C)));
        if (v < threshold) v = 0.0;
        mine = mine || v != 0.0;
        Xe[i] = v;
    }
    if (mine) any = 1;
    __syncthreads();
    if (!any && threadIdx.x == 0) {
        const u64 pos = mix(splitmix64(seed ^ 0x5EEDull), e) % (u64)n;   // natural index r * C + c
        const int r = (int)(pos / (u64)C), c = (int)(pos % (u64)C);
        Xe[r + ((size_t)c << lgR)] = 0.995 + 0.005 * unit(mix(splitmix64(seed ^ 0xBEEFull), e));
    }
}

__constant__ int kGaussOffsets[20] = {-3, -2, -2, -1, -1, -1, -1, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3};

// One wave per example: the lanes clear the image, lane 0 fills the deposits in their order (deposits
// in one bin add, and the sum depends on the order), the lanes round every bin to float.
__global__ __launch_bounds__(64) void kg_calorimeter(u64 seed, long long first, int lgR, int R, int C, int n,
                                                     const int* ptab, const double* etab, int jets, int jet_deposits,
                                                     double* X) {
    const u64 e = (u64)(first + blockIdx.x);
    const u64 he = mix(splitmix64(seed), e);
    double* Xe = X + (size_t)blockIdx.x * n;
    for (int i = threadIdx.x; i < n; i += 64) Xe[i] = 0.0;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int K = ptab[mix(he, 1) % 4096ull];
        for (int d = 0; d < K; d++) {
            const u64 h = mix(he, (u64)(2 + d));
            const int r = (int)(h % (u64)R), c = (int)((h >> 20) % (u64)C);
            Xe[r + ((size_t)c << lgR)] += 0.5 * etab[(h >> 40) % 1024ull];
        }
        for (int j = 0; j < jets; j++) {
            const u64 hc = mix(he, (u64)(1000 + j));
            const int r0 = (int)(hc % (u64)R), c0 = (int)((hc >> 20) % (u64)C);
            for (int dd = 0; dd < jet_deposits; dd++) {
                const u64 hj = mix(he, (u64)(2000 + 100 * j + dd));
                const int dr = kGaussOffsets[hj % 20ull], dc = kGaussOffsets[(hj >> 16) % 20ull];
                const int r = ((r0 + dr) % R + R) % R, c = ((c0 + dc) % C + C) % C;
                Xe[r + ((size_t)c << lgR)] += 4.0 * etab[(hj >> 40) % 1024ull];
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += 64) Xe[i] = (double)(float)Xe[i];
}

int ilog2_exact(int v) {
    if (v < 1 || (v & (v - 1))) return -1;
    int l = 0;
    while ((1 << l) < v) l++;
    return l;
}

int check(wn_context* ctx, int B, int R, int C, const double* X, int* lgR) {
    if (!ctx) return WN_ERR_ARG;
    if (B < 0) return fail(ctx, WN_ERR_ARG, "negative batch size");
    *lgR = ilog2_exact(R);
    if (*lgR < 0 || ilog2_exact(C) < 0)
        return fail(ctx, WN_ERR_BAD_SHAPE, "shape {" + std::to_string(R) + ", " + std::to_string(C) + "} is not radix 2");
    if (*lgR + ilog2_exact(C) > 30) return fail(ctx, WN_ERR_UNSUPPORTED, "more than 2^30 samples per example");
    if (B > 65535) return fail(ctx, WN_ERR_UNSUPPORTED, "more than 65535 examples per call");
    if (B > 0 && !X) return fail(ctx, WN_ERR_ARG, "null output");
    return WN_OK;
}

}  // namespace

extern "C" {

int wn_generate_uniform(wn_context* ctx, unsigned long long seed, long long first, int B, int R, int C, double* X) {
    wn::make_current(ctx);
    int lgR;
    const int rc = check(ctx, B, R, C, X, &lgR);
    if (rc || B == 0) return rc;
    const int n = R * C;
    const int gx = n / 256 < 1 ? 1 : (n / 256 > 1024 ? 1024 : n / 256);
    hipLaunchKernelGGL(kg_uniform, dim3(gx, B), dim3(256), 0, ctx->stream, seed, first, lgR, C, n, X);
    WN_HIP(ctx, hipGetLastError());
    return WN_OK;
}

int wn_generate_needle(wn_context* ctx, unsigned long long seed, long long first, int B, int R, int C, double threshold,
                       double* X) {
    wn::make_current(ctx);
    int lgR;
    const int rc = check(ctx, B, R, C, X, &lgR);
    if (rc || B == 0) return rc;
    hipLaunchKernelGGL(kg_needle, dim3(B), dim3(256), 0, ctx->stream, seed, first, lgR, C, R * C, threshold, X);
    WN_HIP(ctx, hipGetLastError());
    return WN_OK;
}

int wn_generate_calorimeter(wn_context* ctx, unsigned long long seed, long long first, int B, int R, int C,
                            const int* poisson_table, const double* energy_table, int jets, int jet_deposits, double* X) {
    wn::make_current(ctx);
    int lgR;
    const int rc = check(ctx, B, R, C, X, &lgR);
    if (rc || B == 0) return rc;
    if (!poisson_table || !energy_table || jets < 0 || jet_deposits < 0) return fail(ctx, WN_ERR_ARG, "bad tables or counts");
    if (!ctx->gen_tables) WN_HIP(ctx, hipMalloc(&ctx->gen_tables, 4096 * sizeof(int) + 1024 * sizeof(double)));
    int* ptab = reinterpret_cast<int*>(ctx->gen_tables);
    double* etab = reinterpret_cast<double*>(reinterpret_cast<char*>(ctx->gen_tables) + 4096 * sizeof(int));
    // the tables are the caller's (pageable) memory: synchronous copies, after the kernels that may still read them
    WN_HIP(ctx, hipStreamSynchronize(ctx->stream));
    WN_HIP(ctx, hipMemcpy(ptab, poisson_table, 4096 * sizeof(int), hipMemcpyHostToDevice));
    WN_HIP(ctx, hipMemcpy(etab, energy_table, 1024 * sizeof(double), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(kg_calorimeter, dim3(B), dim3(64), 0, ctx->stream, seed, first, lgR, R, C, R * C, ptab, etab, jets,
                       jet_deposits, X);
    WN_HIP(ctx, hipGetLastError());
    return WN_OK;
}

}  // extern "C"
