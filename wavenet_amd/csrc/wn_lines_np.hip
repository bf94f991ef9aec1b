// Instantiates the one-wave-per-line kernels of the streamed path (wn_lines.h) for one compiled tap count
// (-DWN_NP=2|4|8|16|32) and exports their addresses.  kind 0: columns forward, 1: columns backward from
// errors, 2: columns backward from ranks, 3: rows forward, 4: rows filter gradient, 5: columns forward with the
// ranking's histogram, 6: columns backward from (home index, rank) pairs; M = samples per lane
// (line length / 64): 2, 4, 8 or 16.
#include "wn_lines.h"

#ifndef WN_NP
#error "compile with -DWN_NP=<taps>"
#endif

#define WN_CAT2(a, b) a##b
#define WN_CAT(a, b) WN_CAT2(a, b)

namespace {
template <int M>
const void* pick(int kind) {
    switch (kind) {
        case 0: return (const void*)&wn::kc_fwd_cols<WN_NP, M>;
        case 1: return (const void*)&wn::kc_bwd_cols<WN_NP, M, false>;
        case 2: return (const void*)&wn::kc_bwd_cols<WN_NP, M, true>;
        case 3: return (const void*)&wn::kr_fwd_rows<WN_NP, M>;
        case 4: return (const void*)&wn::kr_grad_rows<WN_NP, M>;
        case 5: return (const void*)&wn::kc_fwd_cols_hist<WN_NP, M>;
        case 6: return (const void*)&wn::kc_bwd_cols_regions<WN_NP, M>;
    }
    return nullptr;
}
}  // namespace

extern "C" const void* WN_CAT(wn_kernel_lines_np, WN_NP)(int kind, int M) {
    switch (M) {
        case 2: return pick<2>(kind);
        case 4: return pick<4>(kind);
        case 8: return pick<8>(kind);
        case 16: return pick<16>(kind);
    }
    return nullptr;
}
