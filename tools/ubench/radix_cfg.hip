// Times rocprim::radix_sort_pairs on 32-bit keys + 32-bit values with several Onesweep configurations
// (radix bits per pass, sort kernel shape).  ./radix_cfg [millions of pairs]
#include <hip/hip_runtime.h>
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <class Config>
int run(const char* name, unsigned* kin, unsigned* kout, unsigned* vin, unsigned* vout, size_t n) {
    size_t bytes = 0;
    CK((rocprim::radix_sort_pairs<Config>(nullptr, bytes, kin, kout, vin, vout, n, 0, 32, (hipStream_t)0)));
    void* tmp = nullptr;
    CK(hipMalloc(&tmp, bytes));
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    for (int i = 0; i < 2; i++) CK((rocprim::radix_sort_pairs<Config>(tmp, bytes, kin, kout, vin, vout, n, 0, 32, (hipStream_t)0)));
    CK(hipEventRecord(a));
    const int reps = 5;
    for (int i = 0; i < reps; i++) CK((rocprim::radix_sort_pairs<Config>(tmp, bytes, kin, kout, vin, vout, n, 0, 32, (hipStream_t)0)));
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    std::vector<unsigned> head(4096);
    CK(hipMemcpy(head.data(), kout + n / 2, head.size() * sizeof(unsigned), hipMemcpyDeviceToHost));
    bool ok = true;
    for (size_t i = 1; i < head.size(); i++) ok = ok && head[i - 1] <= head[i];
    std::printf("%-28s %8.3f ms  %6.1f G pairs/s  %s\n", name, ms / reps, n / (ms / reps) / 1e6, ok ? "sorted" : "NOT SORTED");
    CK(hipFree(tmp));
    return 0;
}

template <unsigned BITS, unsigned BS, unsigned IPT>
using Cfg = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                       rocprim::radix_sort_onesweep_config<rocprim::kernel_config<256, 12>, rocprim::kernel_config<BS, IPT>, BITS>>;

int main(int argc, char** argv) {
    const size_t n = (size_t)(argc > 1 ? std::atoi(argv[1]) : 57) * 1000000;
    std::vector<unsigned> h(n);
    unsigned long long x = 88172645463325252ull;
    for (size_t i = 0; i < n; i++) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; h[i] = (unsigned)(x >> 16); }
    unsigned *kin, *kout, *vin, *vout;
    CK(hipMalloc(&kin, n * 4)); CK(hipMalloc(&kout, n * 4)); CK(hipMalloc(&vin, n * 4)); CK(hipMalloc(&vout, n * 4));
    CK(hipMemcpy(kin, h.data(), n * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(vin, h.data(), n * 4, hipMemcpyHostToDevice));
    if (run<rocprim::default_config>("default", kin, kout, vin, vout, n)) return 1;
    if (run<Cfg<8, 256, 8>>("8 bits, 256 x 8", kin, kout, vin, vout, n)) return 1;
    if (run<Cfg<8, 256, 12>>("8 bits, 256 x 12", kin, kout, vin, vout, n)) return 1;
    if (run<Cfg<8, 256, 16>>("8 bits, 256 x 16", kin, kout, vin, vout, n)) return 1;
    if (run<Cfg<8, 256, 22>>("8 bits, 256 x 22", kin, kout, vin, vout, n)) return 1;
    if (run<Cfg<7, 256, 16>>("7 bits, 256 x 16", kin, kout, vin, vout, n)) return 1;
    if (run<Cfg<7, 512, 12>>("7 bits, 512 x 12", kin, kout, vin, vout, n)) return 1;
    if (run<Cfg<6, 512, 12>>("6 bits, 512 x 12", kin, kout, vin, vout, n)) return 1;
    return 0;
}
