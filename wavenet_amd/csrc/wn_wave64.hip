// Instantiates the one-wave-per-example training kernel of 64x64 examples for the compiled tap
// counts it supports and exports the kernel addresses.
#include "wn_wave64.h"

extern "C" const void* wn_kernel_wave64(int np) {
    switch (np) {
        case 2: return (const void*)&wn::k_train64<2>;
        case 4: return (const void*)&wn::k_train64<4>;
        case 8: return (const void*)&wn::k_train64<8>;
        case 16: return (const void*)&wn::k_train64<16>;
        case 32: return (const void*)&wn::k_train64<32>;
    }
    return nullptr;
}

extern "C" const void* wn_kernel_wave64_sparse() { return (const void*)&wn::k_sparse64; }

// kind 1: forward transform, 2: inverse transform, 3: backpropagation (numbering of the
// workgroup-per-example kernels)
extern "C" const void* wn_kernel_wave64_transform(int kind, int np) {
#define WN_W64_PAIR(NP)                                                  \
    return kind == 1 ? (const void*)&wn::k_forward64<NP>                 \
         : kind == 2 ? (const void*)&wn::k_inverse64<NP>                 \
                     : (const void*)&wn::k_backprop64<NP>;
    if (kind < 1 || kind > 3) return nullptr;
    switch (np) {
        case 2: WN_W64_PAIR(2)
        case 4: WN_W64_PAIR(4)
        case 8: WN_W64_PAIR(8)
        case 16: WN_W64_PAIR(16)
        case 32: WN_W64_PAIR(32)
    }
    return nullptr;
#undef WN_W64_PAIR
}
