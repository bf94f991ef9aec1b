// Instantiates the one-wave-per-example training kernel of 64x64 examples for the compiled tap
// counts it supports and exports the kernel addresses.
#include "wn_wave64.h"

extern "C" const void* wn_kernel_wave64(int np) {
    switch (np) {
        case 2: return (const void*)&wn::k_train64<2>;
        case 4: return (const void*)&wn::k_train64<4>;
        case 8: return (const void*)&wn::k_train64<8>;
        case 16: return (const void*)&wn::k_train64<16>;
    }
    return nullptr;
}
