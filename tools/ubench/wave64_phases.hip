// Per-phase cycle counts of the one-wave-per-example training kernel: the same phase functions as
// k_train64, with the shader clock read between phases.  Sparse random input, 4096 examples.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include "../../wavenet_amd/csrc/wn_wave64.h"

using namespace wn;
constexpr int NPH = 11;

#define TICK(i) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); long long t_ = clock64(); ph[i] += t_ - tprev; tprev = t_; __builtin_amdgcn_sched_barrier(0); }

template <int NP>
__global__ __launch_bounds__(256, 1) void k_phases(FusedArgs A, long long* phases) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const FilterDev* filt = A.filt;
    double a[NP], b[NP];
#pragma unroll
    for (int j = 0; j < NP; j++) { a[j] = filt->a[j]; b[j] = filt->b[j]; }
    long long ph[NPH];
    for (int i = 0; i < NPH; i++) ph[i] = 0;
    double acc = 0;
    const int nslots = gridDim.x * 4;
    long long tprev = clock64();
    for (int e = blockIdx.x * 4 + wave; e < A.B; e += nslots) {
        const double* Xe = A.X + (size_t)e * 4096;
        unsigned tile_off = (unsigned)wave * W64_TILE;
        asm volatile("" : "+s"(tile_off));
        double* tile = smem + tile_off;
        double* trow = tile + lane * 65;
        double v[64], w[64], p[64];
        w64_load_rows(v, Xe, lane);
        TICK(0)
        w64_fwd_levels<NP, 5, true>(v, p, w, trow, a, b);
        TICK(1)
        w64_read_transposed(w, tile, lane);
        TICK(2)
        w64_fwd_levels<NP, 5, false>(w, p, v, trow, a, b);
        TICK(3)
        W64Rank rank = w64_rank_sort(v, tile, lane);
#ifdef H_SKIP_GROUPS
        rank.any_group = false;
#endif
        TICK(4)
#pragma unroll
        for (int r = 0; r < 64; r++) asm volatile("" : "+v"(w[r]));
        w64_fwd_levels<NP, 5, false>(w, p, v, trow, a, b);
        const double S = w64_gini_finish(v, rank, tile, lane);
        TICK(5)
        double gL[NP], gH[NP];
#pragma unroll
        for (int u = 0; u < NP; u++) { gL[u] = 0; gH[u] = 0; }
#ifdef H_PREFETCH
        {   // touch every 64-byte line of the example: 512 lines, 8 per lane
            float sink = 0;
#pragma unroll
            for (int i = 0; i < 8; i++) sink += reinterpret_cast<const float*>(Xe)[(lane + 64 * i) * 16];
            asm volatile("" :: "v"(sink));
        }
#endif
        w64_pyramid_to_tile<NP, 5>(w, p, trow, a);
        w64_bwd_levels<NP, 0, W64_TOP_TILE, true>(v, w, trow, a, b, gL, gH);
        TICK(6)
        w64_read_transposed(v, tile, lane);
#ifndef H_SKIP_RELOAD
        w64_load_rows(w, Xe, lane);
#endif
        TICK(7)
        w64_pyramid_to_tile<NP, 5>(w, p, trow, a);
        w64_bwd_levels<NP, 0, W64_TOP_NONE, true>(v, w, trow, a, b, gL, gH);
        TICK(8)
        double mine = S;
#pragma unroll
        for (int kp = 0; kp < NP; kp++) {
            double c = filt->vL[kp] * gL[kp] + filt->sH[kp] * gH[NP - 1 - kp];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off, 64);
            mine += c;
        }
        acc += mine;
        TICK(9)
    }
    A.S[blockIdx.x * 256 + threadIdx.x] = acc;
    if (lane == 0)
        for (int i = 0; i < NPH; i++) phases[(blockIdx.x * 4 + wave) * NPH + i] = ph[i];
}

int main(int argc, char** argv) {
    const int B = 4096, NP = WN_NP_BENCH;
    std::vector<double> X((size_t)B * 4096, 0.0);
    srand(1);
    for (auto& x : X) if (rand() % 20 == 0) x = (float)(rand() / (double)RAND_MAX);
    FilterDev f{};
    const double db4[4] = {0.48296291314453414, 0.83651630373780794, 0.22414386804201339, -0.12940952255126037};
    f.N = NP; f.NP = NP; f.d = 0;
    for (int j = 0; j < NP; j++) { f.a[j] = db4[j % 4] * (j < 4 ? 1.0 : 0.01); f.vL[j] = 1; }
    for (int j = 0; j < NP; j++) { f.b[j] = ((j & 1) ? -1.0 : 1.0) * f.a[NP - 1 - j]; f.sH[j] = ((NP - 1 - j) & 1) ? -1.0 : 1.0; }
    double *dX, *dS; FilterDev* dF; long long* dP;
    hipMalloc(&dX, X.size() * 8); hipMemcpy(dX, X.data(), X.size() * 8, hipMemcpyHostToDevice);
    hipMalloc(&dS, 256 * 256 * 8); hipMalloc(&dF, sizeof(f)); hipMemcpy(dF, &f, sizeof(f), hipMemcpyHostToDevice);
    hipMalloc(&dP, 1024 * NPH * 8);
    FusedArgs A{}; A.B = B; A.lgR = 6; A.lgC = 6; A.filt = dF; A.X = dX; A.S = dS;
    const size_t shmem = 4 * W64_TILE * 8;
    hipFuncSetAttribute((const void*)k_phases<NP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    for (int it = 0; it < 3; it++) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_phases<NP>, dim3(256), dim3(256), shmem, 0, A, dP);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("launch %d: %.3f ms (%s)\n", it, ms, hipGetErrorString(hipGetLastError()));
    }
    std::vector<long long> h(1024 * NPH);
    hipMemcpy(h.data(), dP, h.size() * 8, hipMemcpyDeviceToHost);
    const char* names[NPH] = {"load X", "fwd rows", "transpose read", "fwd cols", "keys, sort, rank scatter", "fwd cols again, ranks, delta", "pyramid + bwd cols",
                              "transpose read + reload X", "pyramid + bwd rows", "gradient reduce", ""};
    double tot = 0;
    for (int i = 0; i < NPH - 1; i++) { double s = 0; for (int w = 0; w < 1024; w++) s += h[w * NPH + i]; tot += s / 1024 / 4; }
    for (int i = 0; i < NPH - 1; i++) {
        double s = 0; for (int w = 0; w < 1024; w++) s += h[w * NPH + i];
        printf("%-28s %9.0f ticks / example  %5.1f %%\n", names[i], s / 1024 / 4, 100.0 * s / 1024 / 4 / tot);
    }
    printf("total %.0f ticks / example\n", tot);
    return 0;
}
