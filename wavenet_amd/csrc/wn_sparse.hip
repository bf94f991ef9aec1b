// Instantiates the Gini sparsity kernels (no filter taps involved), one per size class.
#include "wn_kernels.h"

extern "C" const void* wn_kernel_sparse(int sc) {
    switch (sc) {
        case 0: return (const void*)&wn::k_sparse<64, 1>;
        case 1: return (const void*)&wn::k_sparse<64, 3>;
        case 2: return (const void*)&wn::k_sparse<64, 4>;
        case 3: return (const void*)&wn::k_sparse<256, 3>;
        case 4: return (const void*)&wn::k_sparse<512, 3>;
    }
    return nullptr;
}
