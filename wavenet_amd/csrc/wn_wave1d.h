// Training kernel for 1D signals of 1024 samples: ONE WAVE owns one example.
//
// Lane l holds 16 consecutive samples of the signal in registers, and every level of the cascade
// keeps that shape -- 8, 4, 2, 1 consecutive low-pass values per lane down to 64 values.  A lane
// needs N/2 - 1 samples of its neighbours on either side per level; they come through the
// cross-lane network (ds_bpermute: no LDS storage), and since 64 lanes times the block size is the
// whole period, rotating the lanes IS the periodic boundary.  Below 64 values a level is held
// "replicated": every lane holds the value whose index is its lane number modulo the level's
// length, and fetches its taps from the lanes that hold them.  All filter arithmetic is
// straight-line code on registers with compile-time indices; the low-pass pyramid of the backward
// pass simply stays in registers (36 doubles).  LDS holds only the 1024 coefficients for the Gini
// phase, which is the one of the workgroup-per-example kernels run by a single wave.
//
// Build switches for timing experiments (tools/build_variant.sh): W1D_SKIP_GINI, W1D_SKIP_BWD leave a
// phase out (wrong results), W1D_WAVES sets the waves per SIMD the register budget is compiled for.
#pragma once

#include "wn_kernels.h"

namespace wn {

// ds_bpermute takes a byte address and an immediate offset and uses (address + offset) / 4 modulo 64
// as the source lane, so a lane number may be given unreduced; the callers keep the variable part
// non-negative and the compile-time part in the immediate.
__device__ __forceinline__ double w1d_lane(double v, int src) {
    const int lo = __builtin_amdgcn_ds_bpermute(src << 2, __double2loint(v));
    const int hi = __builtin_amdgcn_ds_bpermute(src << 2, __double2hiint(v));
    return __hiloint2double(hi, lo);
}
// v of lane (lane - D) mod 64
template <int D>
__device__ __forceinline__ double w1d_from(double v, int lane) {
    return w1d_lane(v, lane + ((64 - D) & 63));
}

// A value that is the same in every lane, moved to scalar registers.
__device__ __forceinline__ double w1d_uniform(double v) {
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)),
                            __builtin_amdgcn_readfirstlane(__double2loint(v)));
}

// The lane's M values (own[M]) with HL values of the lanes before it and HR of the lanes after it in
// ext[HL + M + HR]: ext[HL + i] = own[i]; ext[HL - 1 - t] and ext[HL + M + t] continue the sequence on either side.
template <int M, int HL, int HR>
__device__ __forceinline__ void w1d_extend(const double* own, double* ext, int lane) {
#pragma unroll
    for (int i = 0; i < M; i++) ext[HL + i] = own[i];
#pragma unroll
    for (int t = 0; t < HL; t++) {
        const int d = 1 + t / M, reg = M - 1 - (t % M);
        // distance d is a compile-time value after unrolling; the switch keeps it a template argument
        double v;
        switch (d) {
            case 1: v = w1d_from<1>(own[reg], lane); break;
            case 2: v = w1d_from<2>(own[reg], lane); break;
            case 3: v = w1d_from<3>(own[reg], lane); break;
            case 4: v = w1d_from<4>(own[reg], lane); break;
            case 5: v = w1d_from<5>(own[reg], lane); break;
            case 6: v = w1d_from<6>(own[reg], lane); break;
            case 7: v = w1d_from<7>(own[reg], lane); break;
            case 8: v = w1d_from<8>(own[reg], lane); break;
            case 9: v = w1d_from<9>(own[reg], lane); break;
            case 10: v = w1d_from<10>(own[reg], lane); break;
            case 11: v = w1d_from<11>(own[reg], lane); break;
            case 12: v = w1d_from<12>(own[reg], lane); break;
            case 13: v = w1d_from<13>(own[reg], lane); break;
            case 14: v = w1d_from<14>(own[reg], lane); break;
            case 15: v = w1d_from<15>(own[reg], lane); break;
            default: v = w1d_from<16>(own[reg], lane); break;
        }
        ext[HL - 1 - t] = v;
    }
#pragma unroll
    for (int t = 0; t < HR; t++) {
        const int d = 1 + t / M, reg = t % M;
        double v;
        switch (d) {
            case 1: v = w1d_from<-1>(own[reg], lane); break;
            case 2: v = w1d_from<-2>(own[reg], lane); break;
            case 3: v = w1d_from<-3>(own[reg], lane); break;
            case 4: v = w1d_from<-4>(own[reg], lane); break;
            case 5: v = w1d_from<-5>(own[reg], lane); break;
            case 6: v = w1d_from<-6>(own[reg], lane); break;
            case 7: v = w1d_from<-7>(own[reg], lane); break;
            case 8: v = w1d_from<-8>(own[reg], lane); break;
            case 9: v = w1d_from<-9>(own[reg], lane); break;
            case 10: v = w1d_from<-10>(own[reg], lane); break;
            case 11: v = w1d_from<-11>(own[reg], lane); break;
            case 12: v = w1d_from<-12>(own[reg], lane); break;
            case 13: v = w1d_from<-13>(own[reg], lane); break;
            case 14: v = w1d_from<-14>(own[reg], lane); break;
            case 15: v = w1d_from<-15>(own[reg], lane); break;
            default: v = w1d_from<-16>(own[reg], lane); break;
        }
        ext[HL + M + t] = v;
    }
}

// ---- forward (src/Wavenet.cxx:404-445) ----------------------------------------------------------------

// One level with M input samples per lane (in[M]): lo[m], hi[m] (M/2 each) = outputs M/2 * lane + m.
// All arrays are indexed with compile-time constants only, so they stay in registers.
template <int NP, int M>
__device__ __forceinline__ void w1d_fwd_block(const double* in, double* lo, double* hi, const double (&a)[NP],
                                              const double (&b)[NP], int lane) {
    constexpr int H = NP / 2 - 1;
    double ext[H + M + H];
    w1d_extend<M, H, H>(in, ext, lane);
#pragma unroll
    for (int m = 0; m < M / 2; m++) {
        double sl = 0, sh = 0;
#pragma unroll
        for (int j = 0; j < NP; j++) {
            const double v = ext[H + 2 * m + NP / 2 - j];
            sl = __builtin_fma(a[j], v, sl);
            sh = __builtin_fma(b[j], v, sh);
        }
        lo[m] = sl;
        hi[m] = sh;
    }
}

// One level below 64 values: in is the value of index lane mod 2^(LG+1), lo / hi the outputs of
// index lane mod 2^LG.
template <int NP, int LG>
__device__ __forceinline__ void w1d_fwd_small(double in, double& lo, double& hi, const double (&a)[NP],
                                              const double (&b)[NP], int lane) {
    constexpr int half = 1 << LG;
    // source index 2k + N/2 - j modulo the length; any lane congruent to it holds the value
    const int base = 2 * (lane & (half - 1)) + 64 + NP / 2 - (NP - 1);
    double v[NP];   // all gathers first: they are independent, the sums are chains
#pragma unroll
    for (int j = 0; j < NP; j++) v[j] = w1d_lane(in, base + (NP - 1 - j));
    __builtin_amdgcn_sched_barrier(0);
    double sl = 0, sh = 0;
#pragma unroll
    for (int j = 0; j < NP; j++) {
        sl = __builtin_fma(a[j], v[j], sl);
        sh = __builtin_fma(b[j], v[j], sh);
    }
    lo = sl;
    hi = sh;
}

// ---- backward (src/Wavenet.cxx:509-545) ---------------------------------------------------------------

// One level with MR errors per lane (dlo[i], dhi[i] = index MR * lane + i) and its activations, 2 MR
// per lane.  Accumulates the filter-gradient correlation; with PROP, out receives the errors on the
// level's input, 2 MR per lane.
template <int NP, int MR, bool PROP>
__device__ __forceinline__ void w1d_bwd_block(const double* dlo, const double* dhi, const double* act, double* out,
                                              const double (&a)[NP], const double (&b)[NP], double (&gL)[NP], double (&gH)[NP],
                                              int lane) {
    {
        constexpr int H = NP / 2 - 1;
        double ext[H + 2 * MR + H];
        w1d_extend<2 * MR, H, H>(act, ext, lane);
#pragma unroll
        for (int i = 0; i < MR; i++) {
#pragma unroll
            for (int u = 0; u < NP; u++) {
                const double v = ext[H + 2 * i + NP / 2 - u];
                gL[u] = __builtin_fma(dlo[i], v, gL[u]);
                gH[u] = __builtin_fma(dhi[i], v, gH[u]);
            }
        }
    }
    if constexpr (PROP) {
        // out[2p + q] = sum_i a[Jq + 2i] dlo[p + i + Cq] + b[Jq + 2i] dhi[p + i + Cq]  (bwd_level, wn_device.h)
        constexpr int S = NP / 2;
        constexpr int JE = S & 1, JO = (S + 1) & 1;
        constexpr int CE = (JE - S) / 2, CO = (1 + JO - S) / 2;
        constexpr int CMIN = CE < CO ? CE : CO, CMAX = CE < CO ? CO : CE;
        constexpr int HL = -CMIN, HR = (NP / 2 - 1 + CMAX) > 0 ? (NP / 2 - 1 + CMAX) : 0;
        double el[HL + MR + HR], eh[HL + MR + HR];
        w1d_extend<MR, HL, HR>(dlo, el, lane);
        w1d_extend<MR, HL, HR>(dhi, eh, lane);
#pragma unroll
        for (int p = 0; p < MR; p++) {
            double se = 0, so = 0;
#pragma unroll
            for (int i = 0; i < NP / 2; i++) {
                se = __builtin_fma(a[JE + 2 * i], el[HL + p + i + CE], se);
                se = __builtin_fma(b[JE + 2 * i], eh[HL + p + i + CE], se);
                so = __builtin_fma(a[JO + 2 * i], el[HL + p + i + CO], so);
                so = __builtin_fma(b[JO + 2 * i], eh[HL + p + i + CO], so);
            }
            out[2 * p] = se;
            out[2 * p + 1] = so;
        }
    }
}

// One level below 64 values: dlo, dhi of index lane mod 2^LG, act and the result of index lane mod
// 2^(LG+1).  Only the lanes below 2^LG add to the gradient (the others hold copies).
template <int NP, int LG>
__device__ __forceinline__ double w1d_bwd_small(double dlo, double dhi, double act, const double (&a)[NP],
                                                const double (&b)[NP], double (&gL)[NP], double (&gH)[NP], int lane) {
    constexpr int len = 1 << LG, alen = 2 * len;
    const double wl = lane < len ? dlo : 0.0, wh = lane < len ? dhi : 0.0;
    const int gbase = 2 * (lane & (len - 1)) + 64 + NP / 2 - (NP - 1);
    {
        double v[NP];
#pragma unroll
        for (int u = 0; u < NP; u++) v[u] = w1d_lane(act, gbase + (NP - 1 - u));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < NP; u++) {
            gL[u] = __builtin_fma(wl, v[u], gL[u]);
            gH[u] = __builtin_fma(wh, v[u], gH[u]);
        }
    }
    constexpr int S = NP / 2;
    constexpr int JE = S & 1, JO = (S + 1) & 1;
    constexpr int CE = (JE - S) / 2, CO = (1 + JO - S) / 2;
    // Both parities are evaluated from one set of gathers and the lane keeps its own: indexing the
    // taps by the lane's parity would turn the tap arrays into memory.
    constexpr int CMIN = CE < CO ? CE : CO;
    const int pbase = ((lane & (alen - 1)) >> 1) + 64 + CMIN;
    double vl[NP / 2 + 1], vh[NP / 2 + 1];
#pragma unroll
    for (int t = 0; t <= NP / 2; t++) {
        vl[t] = w1d_lane(dlo, pbase + t);
        vh[t] = w1d_lane(dhi, pbase + t);
    }
    __builtin_amdgcn_sched_barrier(0);
    double se = 0, so = 0;
#pragma unroll
    for (int i = 0; i < NP / 2; i++) {
        se = __builtin_fma(a[JE + 2 * i], vl[i + CE - CMIN], se);
        se = __builtin_fma(b[JE + 2 * i], vh[i + CE - CMIN], se);
        so = __builtin_fma(a[JO + 2 * i], vl[i + CO - CMIN], so);
        so = __builtin_fma(b[JO + 2 * i], vh[i + CO - CMIN], so);
    }
    const double s = (lane & 1) ? so : se;
    return s;
}

// ---- kernel -------------------------------------------------------------------------------------------
// 64 threads = one wave = one example in flight per workgroup; grid-stride over the batch.
#ifndef W1D_WAVES
#define W1D_WAVES (NP > 16 ? 1 : 2)
#endif
template <int NP>
__global__ __launch_bounds__(64, W1D_WAVES) void k_train1d1024(FusedArgs A) {
    constexpr int n = 1024;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* bufY = smem;                       // 1024 coefficients / sort slots / errors
    double* red = bufY + n;                    // scratch of the Gini phase
    int* flag = reinterpret_cast<int*>(red + 4);
    const int lane = threadIdx.x;
    const FilterDev* filt = A.filt;
    double a[NP], b[NP];
#pragma unroll
    for (int j = 0; j < NP; j++) { a[j] = filt->a[j]; b[j] = filt->b[j]; }
    if constexpr (NP <= 16) {
        // the taps are the same in every lane: keep them in scalar registers
#pragma unroll
        for (int j = 0; j < NP; j++) { a[j] = w1d_uniform(a[j]); b[j] = w1d_uniform(b[j]); }
    }
    const int N = filt->N, dpad = filt->d;

    double accG = 0, accS = 0, accBad = 0;
    for (int e = blockIdx.x; e < A.B; e += gridDim.x) {
        // ---- forward
        double x[16], lo9[8], lo8[4], lo7[2], lo6[1], hi[8];
        {
            const double2* src = reinterpret_cast<const double2*>(A.X + (size_t)e * n + 16 * lane);
#pragma unroll
            for (int i = 0; i < 8; i++) { const double2 t = src[i]; x[2 * i] = t.x; x[2 * i + 1] = t.y; }
        }
        __syncthreads();   // the previous example's errors have been read
        w1d_fwd_block<NP, 16>(x, lo9, hi, a, b, lane);
#pragma unroll
        for (int m = 0; m < 8; m++) bufY[512 + 8 * lane + m] = hi[m];
        {
            double h8[4];
            w1d_fwd_block<NP, 8>(lo9, lo8, h8, a, b, lane);
#pragma unroll
            for (int m = 0; m < 4; m++) bufY[256 + 4 * lane + m] = h8[m];
            double h7[2];
            w1d_fwd_block<NP, 4>(lo8, lo7, h7, a, b, lane);
            bufY[128 + 2 * lane] = h7[0];
            bufY[128 + 2 * lane + 1] = h7[1];
            double h6[1];
            w1d_fwd_block<NP, 2>(lo7, lo6, h6, a, b, lane);
            bufY[64 + lane] = h6[0];
        }
        // levels below 64 values, replicated: los[i] = low-pass level i at index lane mod 2^i
        double los[6], his;
        w1d_fwd_small<NP, 5>(lo6[0], los[5], his, a, b, lane);
        if (lane >= 32) bufY[lane] = his;                  // coefficient 32 + k sits in lane 32 + k
        w1d_fwd_small<NP, 4>(los[5], los[4], his, a, b, lane);
        if (lane >= 16 && lane < 32) bufY[lane] = his;
        w1d_fwd_small<NP, 3>(los[4], los[3], his, a, b, lane);
        if (lane >= 8 && lane < 16) bufY[lane] = his;
        w1d_fwd_small<NP, 2>(los[3], los[2], his, a, b, lane);
        if (lane >= 4 && lane < 8) bufY[lane] = his;
        w1d_fwd_small<NP, 1>(los[2], los[1], his, a, b, lane);
        if (lane >= 2 && lane < 4) bufY[lane] = his;
        w1d_fwd_small<NP, 0>(los[1], los[0], his, a, b, lane);
        if (lane == 1) bufY[1] = his;
        if (lane == 0) bufY[0] = los[0];
        __syncthreads();

        // ---- Gini term and its gradient, in place (src/CostFunctions.cxx:6-37, 49-105)
#ifdef W1D_SKIP_GINI
        const double S = 1.0;
#else
        const double S = gini_phase<64, 4, true>(bufY, n, red, flag, lane);
#endif

        // ---- backward
        double gL[NP], gH[NP];
#pragma unroll
        for (int u = 0; u < NP; u++) { gL[u] = 0; gH[u] = 0; }
#ifndef W1D_SKIP_BWD
        // For the shorter filters the signal is read again for the last level rather than held in
        // registers through the Gini phase (measured: faster up to 8 taps, slower at 16).
        double xb[16];
        if constexpr (NP <= 8) {
            const double* again = A.X + (size_t)e * n + 16 * lane;
            asm volatile("" : "+v"(again));
            const double2* src = reinterpret_cast<const double2*>(again);
#pragma unroll
            for (int i = 0; i < 8; i++) { const double2 t = src[i]; xb[2 * i] = t.x; xb[2 * i + 1] = t.y; }
        } else {
#pragma unroll
            for (int i = 0; i < 16; i++) xb[i] = x[i];
        }
        double d = bufY[0];                                                                      // errors on lo_0
        d = w1d_bwd_small<NP, 0>(d, bufY[1], los[1], a, b, gL, gH, lane);                        // -> 2 values
        d = w1d_bwd_small<NP, 1>(d, bufY[2 + (lane & 1)], los[2], a, b, gL, gH, lane);           // -> 4
        d = w1d_bwd_small<NP, 2>(d, bufY[4 + (lane & 3)], los[3], a, b, gL, gH, lane);           // -> 8
        d = w1d_bwd_small<NP, 3>(d, bufY[8 + (lane & 7)], los[4], a, b, gL, gH, lane);           // -> 16
        d = w1d_bwd_small<NP, 4>(d, bufY[16 + (lane & 15)], los[5], a, b, gL, gH, lane);         // -> 32
        d = w1d_bwd_small<NP, 5>(d, bufY[32 + (lane & 31)], lo6[0], a, b, gL, gH, lane);         // -> 64: one per lane
        double d1[1] = {d}, h1[1] = {bufY[64 + lane]}, d2[2], h2[2], d4[4], h4[4], d8[8], h8[8], unused[16];
        w1d_bwd_block<NP, 1, true>(d1, h1, lo7, d2, a, b, gL, gH, lane);
        h2[0] = bufY[128 + 2 * lane]; h2[1] = bufY[128 + 2 * lane + 1];
        w1d_bwd_block<NP, 2, true>(d2, h2, lo8, d4, a, b, gL, gH, lane);
#pragma unroll
        for (int m = 0; m < 4; m++) h4[m] = bufY[256 + 4 * lane + m];
        w1d_bwd_block<NP, 4, true>(d4, h4, lo9, d8, a, b, gL, gH, lane);
#pragma unroll
        for (int m = 0; m < 8; m++) h8[m] = bufY[512 + 8 * lane + m];
        w1d_bwd_block<NP, 8, false>(d8, h8, xb, unused, a, b, gL, gH, lane);

#else
        gL[0] = bufY[lane];
#endif
        // tap k' collects gL[k'] and sH[k'] gH[NP-1-k'], summed over the 64 lanes
        double mine = 0;
#pragma unroll
        for (int kp = 0; kp < NP; kp++) {
            double c = filt->vL[kp] * gL[kp] + filt->sH[kp] * gH[NP - 1 - kp];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off, 64);
            if (lane == kp) mine = c;
        }
        const int k = lane - dpad;
        if (lane < NP && k >= 0 && k < N) {
            accG += mine;
            if (!isfinite(mine)) accBad += 1;
            if (A.grad) A.grad[(size_t)e * N + k] = mine;
        }
        if (lane == 0) {
            accS += S;
            if (!isfinite(S)) accBad += 1;
            if (A.S) A.S[e] = S;
        }
    }
    double bad = accBad;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) bad += __shfl_xor(bad, off, 64);
    double* part = A.partial + (size_t)blockIdx.x * PSTRIDE;
    const int k = lane - dpad;
    if (lane < NP && k >= 0 && k < N) part[k] = accG;
    if (lane == 0) { part[N] = accS; part[N + 1] = bad; }
}

}  // namespace wn
