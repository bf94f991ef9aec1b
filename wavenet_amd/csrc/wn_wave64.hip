// Entry points to the one-wave-per-example kernels of 64x64 examples (wn_wave64.h): the kernels that
// depend on the tap count are compiled per tap count in wn_wave64_np.hip.
#include "wn_wave64.h"

extern "C" const void* wn_kernel_wave64_np2(int kind);
extern "C" const void* wn_kernel_wave64_np4(int kind);
extern "C" const void* wn_kernel_wave64_np8(int kind);
extern "C" const void* wn_kernel_wave64_np16(int kind);
extern "C" const void* wn_kernel_wave64_np32(int kind);

// kind 0: training step, 1: forward transform, 2: inverse transform, 3: backpropagation (the
// numbering of the workgroup-per-example kernels), 4: sparsity cost over a table of filters
extern "C" const void* wn_kernel_wave64_transform(int kind, int np) {
    switch (np) {
        case 2: return wn_kernel_wave64_np2(kind);
        case 4: return wn_kernel_wave64_np4(kind);
        case 8: return wn_kernel_wave64_np8(kind);
        case 16: return wn_kernel_wave64_np16(kind);
        case 32: return wn_kernel_wave64_np32(kind);
    }
    return nullptr;
}

extern "C" const void* wn_kernel_wave64(int np) { return wn_kernel_wave64_transform(0, np); }

extern "C" const void* wn_kernel_wave64_sparse() { return (const void*)&wn::k_sparse64<0>; }

// training kernel of 1D signals of 1024 samples, one wave per example (wn_wave1d.h)
extern "C" const void* wn_kernel_wave1d_np2();
extern "C" const void* wn_kernel_wave1d_np4();
extern "C" const void* wn_kernel_wave1d_np8();
extern "C" const void* wn_kernel_wave1d_np16();
extern "C" const void* wn_kernel_wave1d_np32();
extern "C" const void* wn_kernel_wave1d(int np) {
    switch (np) {
        case 2: return wn_kernel_wave1d_np2();
        case 4: return wn_kernel_wave1d_np4();
        case 8: return wn_kernel_wave1d_np8();
        case 16: return wn_kernel_wave1d_np16();
        case 32: return wn_kernel_wave1d_np32();
    }
    return nullptr;
}

// training kernels of 1D signals of 128, 256 and 512 samples, one wave per example (k_train_line, wn_lines.h)
extern "C" const void* wn_kernel_wave1d_line_np2(int M);
extern "C" const void* wn_kernel_wave1d_line_np4(int M);
extern "C" const void* wn_kernel_wave1d_line_np8(int M);
extern "C" const void* wn_kernel_wave1d_line_np16(int M);
extern "C" const void* wn_kernel_wave1d_line_np32(int M);
extern "C" const void* wn_kernel_wave1d_line(int np, int M) {
    switch (np) {
        case 2: return wn_kernel_wave1d_line_np2(M);
        case 4: return wn_kernel_wave1d_line_np4(M);
        case 8: return wn_kernel_wave1d_line_np8(M);
        case 16: return wn_kernel_wave1d_line_np16(M);
        case 32: return wn_kernel_wave1d_line_np32(M);
    }
    return nullptr;
}
