// Instantiates the one-wave-per-example training kernels of 1D signals for one compiled tap count
// (-DWN_NP=2|4|8|16|32) and exports their addresses: 1024 samples (wn_wave1d.h) and 128, 256, 512
// samples (k_train_line in wn_lines.h; M = samples per lane).
#include "wn_lines.h"

#ifndef WN_NP
#error "compile with -DWN_NP=<taps>"
#endif

#define WN_CAT2(a, b) a##b
#define WN_CAT(a, b) WN_CAT2(a, b)

extern "C" const void* WN_CAT(wn_kernel_wave1d_np, WN_NP)() { return (const void*)&wn::k_train1d1024<WN_NP>; }

// used up to 8 taps, for 256 and 512 samples (measured: slower than the workgroup kernels beyond, equal at 128)
extern "C" const void* WN_CAT(wn_kernel_wave1d_line_np, WN_NP)(int M) {
#if WN_NP <= 8
    switch (M) {
        case 4: return (const void*)&wn::k_train_line<WN_NP, 4>;
        case 8: return (const void*)&wn::k_train_line<WN_NP, 8>;
    }
#endif
    (void)M;
    return nullptr;
}
