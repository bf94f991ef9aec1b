// Instantiates the LDS-resident kernels for one compiled tap count (-DWN_NP=2|4|8|16|32) and
// exports their addresses: kind 0 train, 1 forward, 2 inverse, 3 backprop; size class sc picks
// (workgroup size, log2 keys per thread): 0 (64,1)  1 (64,3)  2 (128,3)  3 (256,3)  4 (512,3).
#include "wn_kernels.h"

#ifndef WN_NP
#error "compile with -DWN_NP=<taps>"
#endif

#define WN_CAT2(a, b) a##b
#define WN_CAT(a, b) WN_CAT2(a, b)

#define WN_KERNELS(T, Q)                                                   \
    switch (kind) {                                                        \
        case 0: return (const void*)&wn::k_train<WN_NP, T, Q>;             \
        case 1: return (const void*)&wn::k_forward<WN_NP, T, Q>;           \
        case 2: return (const void*)&wn::k_inverse<WN_NP, T, Q>;           \
        case 3: return (const void*)&wn::k_backprop<WN_NP, T, Q>;          \
    }                                                                      \
    return nullptr;

extern "C" const void* WN_CAT(wn_kernel_np, WN_NP)(int kind, int sc) {
    switch (sc) {
        case 0: { WN_KERNELS(64, 1) }
        case 1: { WN_KERNELS(64, 3) }
        case 2: { WN_KERNELS(64, 4) }
        case 3: { WN_KERNELS(256, 3) }
        case 4: { WN_KERNELS(512, 3) }
    }
    return nullptr;
}
