// Time of the in-register bitonic sort of 4096 keys by one wave (w64_merge), alone on its CU or with
// three neighbours, with and without the LDS transposes / the compare-exchanges.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../../wavenet_amd/csrc/wn_wave64.h"
using namespace wn;

__global__ __launch_bounds__(256, 1) void k_sort(double* out, long long* cyc, int reps, int waves) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wave >= waves) return;
    double* tile = smem + wave * W64_TILE;
    double k[64];
    for (int r = 0; r < 64; r++) k[r] = (double)((lane * 64 + r) * 2654435761u % 1000003u) + 1.0;
    long long t0 = clock64();
    for (int i = 0; i < reps; i++) {
        w64_merge<1>(k, tile, lane);
#pragma unroll
        for (int r = 0; r < 64; r++) k[r] = -k[63 - r] * 1.0000001 + 3e6;   // unsort again, cheaply
    }
    long long t1 = clock64();
    double s = 0;
    for (int r = 0; r < 64; r++) s += k[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}

int main() {
    double* out; long long* cyc;
    hipMalloc(&out, 256 * 256 * 8); hipMalloc(&cyc, 1024 * 8);
    const size_t shmem = 4 * W64_TILE * 8;
    hipFuncSetAttribute((const void*)k_sort, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    const int reps = 50;
    for (int waves : {1, 2, 4}) {
        for (int it = 0; it < 2; it++) hipLaunchKernelGGL(k_sort, dim3(256), dim3(256), shmem, 0, out, cyc, reps, waves);
        hipDeviceSynchronize();
        std::vector<long long> h(1024);
        hipMemcpy(h.data(), cyc, 1024 * 8, hipMemcpyDeviceToHost);
        double s = 0; int n = 0;
        for (int b = 0; b < 256; b++) for (int w = 0; w < waves; w++) { s += h[b * 4 + w]; n++; }
        printf("%d wave(s) per CU: %.0f ticks per sort (%s)\n", waves, s / n / reps, hipGetErrorString(hipGetLastError()));
    }
    return 0;
}
