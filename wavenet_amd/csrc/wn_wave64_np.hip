// Instantiates the one-wave-per-example kernels of 64x64 examples for one compiled tap count
// (-DWN_NP=2|4|8|16|32) and exports their addresses: kind 0 train, 1 forward, 2 inverse, 3 backprop,
// 4 cost grid.
#include "wn_wave64.h"

#ifndef WN_NP
#error "compile with -DWN_NP=<taps>"
#endif

#define WN_CAT2(a, b) a##b
#define WN_CAT(a, b) WN_CAT2(a, b)

extern "C" const void* WN_CAT(wn_kernel_wave64_np, WN_NP)(int kind) {
    switch (kind) {
        case 0: return (const void*)&wn::k_train64<WN_NP>;
        case 1: return (const void*)&wn::k_forward64<WN_NP>;
        case 2: return (const void*)&wn::k_inverse64<WN_NP>;
        case 3: return (const void*)&wn::k_backprop64<WN_NP>;
        case 4: return (const void*)&wn::k_cost64<WN_NP>;
    }
    return nullptr;
}
