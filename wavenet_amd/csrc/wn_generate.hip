// Synthetic training examples produced on the device, a batch at a time, so that an end-to-end run
// does not ship its inputs over PCIe.  Counter-based: every value is a pure function of (seed,
// example number, item) through splitmix64, exactly as in wavenet_amd/generators.py and
// wavenet_amd/host/src/GeneratorBase.cxx, so host and device produce the same bits.
//
// Reference: the generators of include/Wavenet/Generators.h (Needle :50-62, Uniform :87-96, HepMC
// eta-phi histogram :377-393) behind GeneratorBase::next (include/Wavenet/GeneratorBase.h:32-103);
// those reseed from entropy on every reset and cannot be reproduced, which is why these exist.
#include <hip/hip_runtime.h>

#include <stdlib.h>
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

// The same images for examples of up to 4096 bins, built in LDS by the whole wave: 64 deposits per round, one per
// lane.  Deposits in one bin add in the order of their numbers (the sum depends on it), so within a round the lowest
// lane that aims at a bin goes first -- found with an atomic minimum on a tag per bin -- and the others try again.
// Collisions are rare (a few per cent of the bins are hit at all), so a round is two or three passes.
__global__ __launch_bounds__(64) void kg_calorimeter_lds(u64 seed, long long first, int lgR, int R, int C, int n,
                                                         const int* ptab, const double* etab, int jets, int jet_deposits,
                                                         double* X) {
    __shared__ double img[4096];
    __shared__ unsigned tag[4096];
    const int lane = threadIdx.x;
    const u64 e = (u64)(first + blockIdx.x);
    const u64 he = mix(splitmix64(seed), e);
    for (int i = lane; i < n; i += 64) { img[i] = 0.0; tag[i] = 0xFFFFFFFFu; }
    __syncthreads();
    const int K = ptab[mix(he, 1) % 4096ull];
    const int total = K + jets * jet_deposits;
    for (int d0 = 0; d0 < total; d0 += 64) {
        const int d = d0 + lane;
        bool live = d < total;
        int bin = 0;
        double en = 0.0;
        if (live) {
            if (d < K) {
                const u64 h = mix(he, (u64)(2 + d));
                const int r = (int)(h % (u64)R), c = (int)((h >> 20) % (u64)C);
                bin = r + (c << lgR);
                en = 0.5 * etab[(h >> 40) % 1024ull];
            } else {
                const int j = (d - K) / jet_deposits, dd = (d - K) % jet_deposits;
                const u64 hc = mix(he, (u64)(1000 + j));
                const int r0 = (int)(hc % (u64)R), c0 = (int)((hc >> 20) % (u64)C);
                const u64 hj = mix(he, (u64)(2000 + 100 * j + dd));
                const int dr = kGaussOffsets[hj % 20ull], dc = kGaussOffsets[(hj >> 16) % 20ull];
                const int r = ((r0 + dr) % R + R) % R, c = ((c0 + dc) % C + C) % C;
                bin = r + (c << lgR);
                en = 4.0 * etab[(hj >> 40) % 1024ull];
            }
        }
        while (__any(live)) {        // (one wave: the LDS operations of a pass are complete before the next one's)
            if (live) atomicMin(&tag[bin], (unsigned)lane);
            __syncthreads();
            const bool go = live && tag[bin] == (unsigned)lane;
            __syncthreads();
            if (go) { img[bin] += en; tag[bin] = 0xFFFFFFFFu; live = false; }
            __syncthreads();
        }
    }
    double* Xe = X + (size_t)blockIdx.x * n;
    for (int i = lane; i < n; i += 64) Xe[i] = (double)(float)img[i];
}

// exp(-t) from a table of exp(-i / 128), i = 0 .. 4096, with linear interpolation; 0 beyond t = 32.  Additions and
// multiplications only, in this order, on host and device alike.
__host__ __device__ inline double table_exp_neg(const double* tab, double t) {
    const double x = t * 128.0;
    if (!(x < 4096.0)) return 0.0;
    const int i = (int)x;
    const double fr = x - (double)i;
    return tab[i] + fr * (tab[i + 1] - tab[i]);
}

// One to three unit-height Gaussian bumps per example (GaussianGenerator of the host layer, same arithmetic).
__global__ __launch_bounds__(256) void kg_gaussian(u64 seed, long long first, int lgR, int R, int C, int n, const double* tab, double* X) {
    const u64 he = mix(splitmix64(seed), (u64)(first + blockIdx.y));
    const unsigned bumps = 1u + (unsigned)(mix(he, 0) % 3ull);
    double mr[3], mc[3], sr[3], sc[3];
    for (unsigned b = 0; b < 3; b++) {
        mr[b] = unit(mix(he, 10 + b)) * R;
        mc[b] = unit(mix(he, 20 + b)) * C;
        sr[b] = 0.5 + 0.15 * R * unit(mix(he, 30 + b));
        sc[b] = 0.5 + 0.15 * C * unit(mix(he, 40 + b));
    }
    double* Xe = X + (size_t)blockIdx.y * n;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const int r = i & ((1 << lgR) - 1), c = i >> lgR;
        double v = 0.0;
        for (unsigned b = 0; b < bumps; b++) {
            const double dr = ((double)r - mr[b]) / sr[b], dc = (C > 1) ? ((double)c - mc[b]) / sc[b] : 0.0;
            v += table_exp_neg(tab, 0.5 * (dr * dr + dc * dc));
        }
        Xe[i] = v;
    }
}

// Device copies of the callers' tables: [Poisson 4096 int | energy 1024 double | exp 4097 double].  A table is sent only
// when it differs from what was sent last (the callers pass the same tables batch after batch), and then synchronously,
// after the kernels that may still read the old one.
constexpr size_t kTabPoisson = 0, kTabEnergy = 4096 * sizeof(int), kTabExp = kTabEnergy + 1024 * sizeof(double),
                 kTabBytes = kTabExp + 4097 * sizeof(double);
int upload_table(wn_context* ctx, size_t offset, const void* src, size_t bytes, void** dev) {
    if (!ctx->gen_tables) {
        WN_HIP(ctx, hipMalloc(&ctx->gen_tables, kTabBytes));
        ctx->gen_tables_host = calloc(1, kTabBytes + 3);
        if (!ctx->gen_tables_host) return fail(ctx, WN_ERR_HIP, "out of host memory");
    }
    char* seen = reinterpret_cast<char*>(ctx->gen_tables_host);
    char* sent = seen + kTabBytes;          // one flag per table
    const int which = offset == kTabPoisson ? 0 : (offset == kTabEnergy ? 1 : 2);
    *dev = reinterpret_cast<char*>(ctx->gen_tables) + offset;
    if (sent[which] && memcmp(seen + offset, src, bytes) == 0) return WN_OK;
    WN_HIP(ctx, hipStreamSynchronize(ctx->stream));
    WN_HIP(ctx, hipMemcpy(*dev, src, bytes, hipMemcpyHostToDevice));
    memcpy(seen + offset, src, bytes);
    sent[which] = 1;
    return WN_OK;
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
    void *pt = nullptr, *et = nullptr;
    int rc2 = upload_table(ctx, kTabPoisson, poisson_table, 4096 * sizeof(int), &pt);
    if (!rc2) rc2 = upload_table(ctx, kTabEnergy, energy_table, 1024 * sizeof(double), &et);
    if (rc2) return rc2;
    const int* ptab = reinterpret_cast<const int*>(pt);
    const double* etab = reinterpret_cast<const double*>(et);
    if (R * C <= 4096)
        hipLaunchKernelGGL(kg_calorimeter_lds, dim3(B), dim3(64), 0, ctx->stream, seed, first, lgR, R, C, R * C, ptab, etab, jets,
                           jet_deposits, X);
    else
        hipLaunchKernelGGL(kg_calorimeter, dim3(B), dim3(64), 0, ctx->stream, seed, first, lgR, R, C, R * C, ptab, etab, jets,
                           jet_deposits, X);
    WN_HIP(ctx, hipGetLastError());
    return WN_OK;
}

int wn_generate_gaussian(wn_context* ctx, unsigned long long seed, long long first, int B, int R, int C, const double* exp_table,
                         double* X) {
    wn::make_current(ctx);
    int lgR;
    const int rc = check(ctx, B, R, C, X, &lgR);
    if (rc || B == 0) return rc;
    if (!exp_table) return fail(ctx, WN_ERR_ARG, "no table");
    void* tab = nullptr;
    const int rc2 = upload_table(ctx, kTabExp, exp_table, 4097 * sizeof(double), &tab);
    if (rc2) return rc2;
    const int n = R * C;
    const int gx = n / 256 < 1 ? 1 : (n / 256 > 1024 ? 1024 : n / 256);
    hipLaunchKernelGGL(kg_gaussian, dim3(gx, B), dim3(256), 0, ctx->stream, seed, first, lgR, R, C, n, reinterpret_cast<const double*>(tab), X);
    WN_HIP(ctx, hipGetLastError());
    return WN_OK;
}

}  // extern "C"
