// __global__ kernels of the LDS-resident path: one workgroup per example, persistent over
// the batch.  Templated on the compiled tap count NP, the workgroup size T and Q = log2 of the
// sort keys per thread (T << Q >= R*C).
#pragma once

#include "wn_device.h"
#include "wn_step.h"

namespace wn {

constexpr int PSTRIDE = MAXTAPS + 2;  // doubles per workgroup in the partial-sum buffer
constexpr int W64_TILE = 64 * 65;     // doubles of LDS per wave in the one-wave-per-example kernels (wn_wave64.h)

struct FusedArgs {
    int B, lgR, lgC;
    const FilterDev* filt;
    // inputs (device, per example R*C doubles, examples contiguous); unused ones are null
    const double* X;
    const double* Yin;      // inverse / sparse: coefficients
    const double* Din;      // backprop: errors on the coefficients
    const double* ProwIn;   // backprop: activations
    const double* Y1In;
    const double* PcolIn;
    // outputs, all optional
    double* Y;
    double* Prow;
    double* Y1;
    double* Pcol;
    double* Xout;     // inverse
    double* Delta;    // sparse term gradient / errors on the input (backprop)
    double* S;        // [B] sparse term per example
    double* grad;     // [B][N] sparsity gradient per example
    double* partial;  // [grid][PSTRIDE] per-workgroup sums: N gradient entries, S, non-finite count
    int F;            // cost grid only: number of filters in the table `filt` points to
    StepTail tail;    // k_train64: what the last workgroup does with the partial sums (wn_step.h)
};

template <int T, int K>
__device__ __forceinline__ void load_regs(double (&x)[K], const double* __restrict__ src, int n, int tid) {
#pragma unroll
    for (int it = 0; it < K / 2; it++) {
        const int i = 2 * (it * T + tid);
        double2 v = make_double2(0.0, 0.0);
        if (i + 1 < n) v = *reinterpret_cast<const double2*>(src + i);
        else if (i < n) v.x = src[i];
        x[2 * it] = v.x;
        x[2 * it + 1] = v.y;
    }
}
template <int T, int K>
__device__ __forceinline__ void store_regs_lds(const double (&x)[K], double* dst, int n, int tid) {
#pragma unroll
    for (int it = 0; it < K / 2; it++) {
        const int i = 2 * (it * T + tid);
        if (i + 1 < n) *reinterpret_cast<double2*>(dst + i) = make_double2(x[2 * it], x[2 * it + 1]);
        else if (i < n) dst[i] = x[2 * it];
    }
}
template <int T>
__device__ __forceinline__ void copy_in(double* dst, const double* __restrict__ src, int n, int tid) {
    for (int i = 2 * tid; i < n; i += 2 * T) {
        if (i + 1 < n) *reinterpret_cast<double2*>(dst + i) = *reinterpret_cast<const double2*>(src + i);
        else dst[i] = src[i];
    }
}
template <int T>
__device__ __forceinline__ void copy_out(double* __restrict__ dst, const double* src, int n, int tid) {
    for (int i = 2 * tid; i < n; i += 2 * T) {
        if (i + 1 < n) *reinterpret_cast<double2*>(dst + i) = *reinterpret_cast<const double2*>(src + i);
        else dst[i] = src[i];
    }
}

// Reduce the per-thread window sums to the N-tap gradient of this example.  Thread k' < NP
// returns tap k' - d of the (padded) filter; other threads return 0.  sg: the sign the high-pass
// window sums carry (-1 where a kernel worked with the high-pass taps times -1, k_train below).
template <int NP, int T>
__device__ __forceinline__ double reduce_gradient(const double (&gL)[NP], const double (&gH)[NP],
                                                  const FilterDev* filt, double* red, int tid, double sg = 1.0) {
    constexpr int NWV = T / 64;
    double cp[NP];
#pragma unroll
    for (int k = 0; k < NP; k++) cp[k] = filt->vL[k] * gL[k] + (sg * filt->sH[k]) * gH[NP - 1 - k];
    const int lane = tid & 63;
    const double mine = wave_scatter_sums<NP>(cp, lane);     // the wave's total of tap lane mod NP
    __syncthreads();
    if (lane < NP) red[(tid >> 6) * NP + lane] = mine;
    __syncthreads();
    double s = 0;
    if (tid < NP) {
#pragma unroll
        for (int w = 0; w < NWV; w++) s += red[w * NP + tid];
    }
    return s;
}

// ---------------------------------------------------------------------------------------------
// Training step of a batch (src/Wavenet.cxx:196-212 per example): forward, Gini gradient,
// backpropagation.  Writes per-workgroup sums; per-example outputs are optional.
// ---------------------------------------------------------------------------------------------
template <int NP, int T, int Q>
__global__ __launch_bounds__(T) void k_train(FusedArgs A) {
    constexpr int K = 1 << Q;
    constexpr int NS = T * K;
    constexpr int NWV = T / 64;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int tid = threadIdx.x;
    const int lgR = A.lgR, lgC = A.lgC, R = 1 << lgR, n = 1 << (lgR + lgC);
    // a 1D signal (one column) has no row stage: its row-stage coefficients are the signal itself,
    // and the buffer for them is not allocated (the host sizes the LDS accordingly)
    const bool twoD = lgC > 0;
    double* bufX = smem;
    double* bufY1 = twoD ? bufX + n : bufX;
    double* bufP = bufY1 + n;
    double* bufY = bufP + n;
    double* red = bufY + NS;
    int* flag = reinterpret_cast<int*>(red + NWV * NP + 2);

    const FilterDev* filt = A.filt;
    double a[NP], b[NP];
#pragma unroll
    for (int j = 0; j < NP; j++) { a[j] = filt->a[j]; b[j] = filt->b[j]; }
    const int N = filt->N, d = filt->d;
    // Above 16 taps the high-pass taps are held as signed low-pass taps, b'[j] = (-1)^j a[NP - 1 - j] = sg b with
    // sg = (-1)^d, which halves the registers for taps; the signs of coefficients and errors that follow from it
    // cancel on the way to the gradient, as spelled out in k_train64 (wn_wave64.h), and the results are the same bits.
    constexpr bool B_FROM_A = (NP > 16);
    if constexpr (B_FROM_A) {
#pragma unroll
        for (int j = 0; j < NP; j++) b[j] = (j & 1) ? -a[NP - 1 - j] : a[NP - 1 - j];
    }
    const double sg = (B_FROM_A && (d & 1)) ? -1.0 : 1.0;

    double accG = 0, accS = 0, accBad = 0;
    double xpre[K];
    int e = blockIdx.x;
    if (e < A.B) load_regs<T, K>(xpre, A.X + (size_t)e * n, n, tid);
    for (; e < A.B; e += gridDim.x) {
        store_regs_lds<T, K>(xpre, bufX, n, tid);
        __syncthreads();
        if (e + (int)gridDim.x < A.B) load_regs<T, K>(xpre, A.X + (size_t)(e + gridDim.x) * n, n, tid);

        if (twoD) fwd_pass<NP, T>(bufX, bufP, bufY1, R, 1, lgR, lgC, true, a, b, tid);   // rows  (Wavenet.cxx:587-601)
        fwd_pass<NP, T>(bufY1, bufP, bufY, 1, R, lgC, lgR, false, a, b, tid);  // cols  (Wavenet.cxx:604-619; :404-445 in 1D)

        const double S = gini_phase<T, Q, true>(bufY, n, red, flag, tid);      // CostFunctions.cxx:49-117

        double gL[NP], gH[NP];
#pragma unroll
        for (int k = 0; k < NP; k++) { gL[k] = 0; gH[k] = 0; }
        bwd_pass<NP, T, K, true>(bufY, bufY1, bufP, 1, R, lgC, lgR, false, lgC > 0, a, b, gL, gH, tid);  // cols (:687-698)
        if (lgC > 0) {
            fwd_pass_lo<NP, T>(bufX, bufP, R, 1, lgR, lgC, true, a, tid);
            bwd_pass<NP, T, K, true>(bufY, bufX, bufP, R, 1, lgR, lgC, true, false, a, b, gL, gH, tid);  // rows (:701-708)
        }
        const double ge = reduce_gradient<NP, T>(gL, gH, filt, red, tid, sg);
        const int k = tid - d;
        if (tid < NP && k >= 0 && k < N) {
            accG += ge;
            if (!isfinite(ge)) accBad += 1;
            if (A.grad) A.grad[(size_t)e * N + k] = ge;
        }
        if (tid == 0) {
            accS += S;
            if (!isfinite(S)) accBad += 1;
            if (A.S) A.S[e] = S;
        }
        __syncthreads();
    }
    // per-workgroup sums
    __syncthreads();
    if (tid < NP) red[tid] = accBad;
    __syncthreads();
    double* part = A.partial + (size_t)blockIdx.x * PSTRIDE;
    const int k = tid - d;
    if (tid < NP && k >= 0 && k < N) part[k] = accG;
    if (tid == 0) {
        double bad = 0;
        for (int j = 0; j < NP; j++) bad += red[j];
        part[N] = accS;
        part[N + 1] = bad;
    }
}

// Forward transform only (Wavenet::forward_, src/Wavenet.cxx:404-445, 562-622).
template <int NP, int T, int Q>
__global__ __launch_bounds__(T) void k_forward(FusedArgs A) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int tid = threadIdx.x;
    const int lgR = A.lgR, lgC = A.lgC, R = 1 << lgR, n = 1 << (lgR + lgC);
    double* bufX = smem;
    double* bufY1 = bufX + n;
    double* bufP = bufY1 + n;
    double* bufY = bufP + n;
    double a[NP], b[NP];
#pragma unroll
    for (int j = 0; j < NP; j++) { a[j] = A.filt->a[j]; b[j] = A.filt->b[j]; }
    for (int e = blockIdx.x; e < A.B; e += gridDim.x) {
        const size_t off = (size_t)e * n;
        copy_in<T>(bufX, A.X + off, n, tid);
        __syncthreads();
        fwd_pass<NP, T>(bufX, bufP, bufY1, R, 1, lgR, lgC, true, a, b, tid);
        if (A.Prow) copy_out<T>(A.Prow + off, bufP, n, tid);
        if (A.Y1) copy_out<T>(A.Y1 + off, bufY1, n, tid);
        __syncthreads();
        fwd_pass<NP, T>(bufY1, bufP, bufY, 1, R, lgC, lgR, false, a, b, tid);
        if (A.Pcol) copy_out<T>(A.Pcol + off, bufP, n, tid);
        if (A.Y) copy_out<T>(A.Y + off, bufY, n, tid);
        __syncthreads();
    }
}

// Inverse transform (Wavenet::inverse_, src/Wavenet.cxx:447-472, 624-663): columns, then rows.
template <int NP, int T, int Q>
__global__ __launch_bounds__(T) void k_inverse(FusedArgs A) {
    constexpr int K = 1 << Q;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int tid = threadIdx.x;
    const int lgR = A.lgR, lgC = A.lgC, R = 1 << lgR, n = 1 << (lgR + lgC);
    double* buf = smem;
    double a[NP], b[NP];
#pragma unroll
    for (int j = 0; j < NP; j++) { a[j] = A.filt->a[j]; b[j] = A.filt->b[j]; }
    double gL[NP], gH[NP];  // unused without GRAD
    for (int e = blockIdx.x; e < A.B; e += gridDim.x) {
        const size_t off = (size_t)e * n;
        copy_in<T>(buf, A.Yin + off, n, tid);
        __syncthreads();
        bwd_pass<NP, T, K, false>(buf, nullptr, nullptr, 1, R, lgC, lgR, false, true, a, b, gL, gH, tid);
        bwd_pass<NP, T, K, false>(buf, nullptr, nullptr, R, 1, lgR, lgC, true, true, a, b, gL, gH, tid);
        copy_out<T>(A.Xout + off, buf, n, tid);
        __syncthreads();
    }
}

// SparseTerm / SparseTermDeriv on a batch of coefficient sets (src/CostFunctions.cxx:6-117).
template <int T, int Q>
__global__ __launch_bounds__(T) void k_sparse(FusedArgs A) {
    constexpr int NS = T << Q;
    constexpr int NWV = T / 64;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int tid = threadIdx.x;
    const int n = 1 << (A.lgR + A.lgC);
    double* bufY = smem;
    double* red = bufY + NS;
    int* flag = reinterpret_cast<int*>(red + NWV + 2);
    for (int e = blockIdx.x; e < A.B; e += gridDim.x) {
        const size_t off = (size_t)e * n;
        copy_in<T>(bufY, A.Yin + off, n, tid);
        __syncthreads();
        double S;
        if (A.Delta) S = gini_phase<T, Q, true>(bufY, n, red, flag, tid);
        else         S = gini_phase<T, Q, false>(bufY, n, red, flag, tid);
        if (A.Delta) copy_out<T>(A.Delta + off, bufY, n, tid);
        if (A.S && tid == 0) A.S[e] = S;
        __syncthreads();
    }
}

// Backpropagation from given errors and activations (Wavenet::backpropagate_,
// src/Wavenet.cxx:474-556, 665-711).  Delta (optional) receives the errors after the column
// stage, i.e. the "errors on the input" of a 1D signal.
template <int NP, int T, int Q>
__global__ __launch_bounds__(T) void k_backprop(FusedArgs A) {
    constexpr int K = 1 << Q;
    constexpr int NWV = T / 64;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int tid = threadIdx.x;
    const int lgR = A.lgR, lgC = A.lgC, R = 1 << lgR, n = 1 << (lgR + lgC);
    double* bufX = smem;
    double* bufY1 = bufX + n;
    double* bufP = bufY1 + n;
    double* bufY = bufP + n;
    double* red = bufY + n;
    (void)NWV;
    const FilterDev* filt = A.filt;
    double a[NP], b[NP];
#pragma unroll
    for (int j = 0; j < NP; j++) { a[j] = filt->a[j]; b[j] = filt->b[j]; }
    const int N = filt->N, d = filt->d;
    for (int e = blockIdx.x; e < A.B; e += gridDim.x) {
        const size_t off = (size_t)e * n;
        copy_in<T>(bufX, A.X + off, n, tid);
        copy_in<T>(bufY1, A.Y1In + off, n, tid);
        copy_in<T>(bufP, A.PcolIn + off, n, tid);
        copy_in<T>(bufY, A.Din + off, n, tid);
        __syncthreads();
        double gL[NP], gH[NP];
#pragma unroll
        for (int k = 0; k < NP; k++) { gL[k] = 0; gH[k] = 0; }
        bwd_pass<NP, T, K, true>(bufY, bufY1, bufP, 1, R, lgC, lgR, false, true, a, b, gL, gH, tid);
        if (A.Delta) copy_out<T>(A.Delta + off, bufY, n, tid);
        if (lgC > 0) {
            __syncthreads();
            copy_in<T>(bufP, A.ProwIn + off, n, tid);
            __syncthreads();
            bwd_pass<NP, T, K, true>(bufY, bufX, bufP, R, 1, lgR, lgC, true, false, a, b, gL, gH, tid);
        }
        const double ge = reduce_gradient<NP, T>(gL, gH, filt, red, tid);
        const int k = tid - d;
        if (tid < NP && k >= 0 && k < N) A.grad[(size_t)e * N + k] = ge;
        __syncthreads();
    }
}

}  // namespace wn
