// Instantiates the one-wave-per-example training kernel of 1024-sample signals for one compiled tap
// count (-DWN_NP=2|4|8|16|32) and exports its address.
#include "wn_wave1d.h"

#ifndef WN_NP
#error "compile with -DWN_NP=<taps>"
#endif

#define WN_CAT2(a, b) a##b
#define WN_CAT(a, b) WN_CAT2(a, b)

extern "C" const void* WN_CAT(wn_kernel_wave1d_np, WN_NP)() { return (const void*)&wn::k_train1d1024<WN_NP>; }
