// Device-side building blocks of the LDS-resident Wavenet kernels (gfx950).
//
// One workgroup owns one example (R x C doubles, column-major, R*C <= T*K) and
// keeps every intermediate of the training step in LDS:
//
//   bufX   input example                           n doubles
//   bufY1  row-stage coefficients                  n doubles
//   bufP   low-pass pyramid of the current stage   n doubles
//   bufY   coefficients / sort keys / errors       T*K doubles
//
// The dense 2^s x 2^(s+1) matrices of the reference (src/MatrixOperator.cxx)
// never exist here: row r of the low-pass matrix holds tap j at column
// (N/2 + 2r - j) mod 2^(s+1), so L*x, L^T*y and the Frobenius products with
// the 0/+-1 weight matrices (src/Wavenet.cxx:509-545) are periodic stride-2
// convolutions / correlations with the N taps held in registers.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace wn {

constexpr int MAXTAPS = 32;

// Filter state resident on the device.  Taps are zero-padded symmetrically from
// N to NP (the compiled tap count) with d = (NP - N) / 2 zeros on each side,
// which leaves both operators unchanged: a'[j + d] = a[j], N'/2 = N/2 + d.
struct FilterDev {
    double a[MAXTAPS];         // padded low-pass taps
    double b[MAXTAPS];         // padded high-pass taps, b[j] = (-1)^j a[N-1-j] (src/HighpassOperator.cxx:16-18)
    double vL[MAXTAPS];        // 1 on real taps, 0 on padding
    double sH[MAXTAPS];        // (-1)^(N-1-k) on real taps, 0 on padding
    double a_true[MAXTAPS];    // the N filter coefficients
    double momentum[MAXTAPS];  // SGD momentum (src/Wavenet.cxx:742-765)
    int N, NP, d, steps;
    int nonfinite;             // set by the update kernel when a batch's sums were not finite; later updates are skipped
    int reserved;
};

__device__ __forceinline__ unsigned swz(unsigned e) {
    // XOR swizzle of the sort buffer: makes the K-strided register layouts of
    // every sort window hit 32 distinct 8-byte banks per half wave.
    const unsigned h = e >> 5;
    const unsigned m1 = (h ^ (h >> 3) ^ (h >> 6)) & 1u;
    const unsigned m2 = ((h >> 1) ^ (h >> 4)) & 1u;
    const unsigned m3 = ((h >> 2) ^ (h >> 5)) & 1u;
    return e ^ m1 ^ (m2 * 10u) ^ (m3 * 20u);
}
constexpr unsigned swz_c(unsigned e) {
    return e ^ (((e >> 5) ^ (e >> 8) ^ (e >> 11)) & 1u) ^ ((((e >> 6) ^ (e >> 9)) & 1u) * 10u) ^
           ((((e >> 7) ^ (e >> 10)) & 1u) * 20u);
}

__device__ __forceinline__ void cmpx(double& lo, double& hi) {
    const double a = __builtin_fmin(lo, hi);
    const double b = __builtin_fmax(lo, hi);
    lo = a;
    hi = b;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

template <int M>
__device__ __forceinline__ double xlane(double v, int lane);   // v of lane (lane ^ M), defined with the sort below

// Sums over the wave's lanes of M values per lane at once (M a power of two, at most 64): returns, in lane l, the
// sum over all lanes of v[l % M].  Butterfly steps at distances of M and more add all M values; below that a
// lane keeps the half of its values that its own lane bit selects and hands the other half to its partner, so
// the steps move M/2, M/4, ... 1 values -- M - 1 exchanges in all instead of 6 M for M separate sums.
// v is destroyed.  A fixed order.
template <int M, int OFF>
__device__ __forceinline__ void wave_sum_scatter_step(double (&v)[M], int lane) {
    if constexpr (OFF >= M) {
#pragma unroll
        for (int i = 0; i < M; i++) v[i] += xlane<OFF>(v[i], lane);
    } else {
        const bool up = (lane & OFF) != 0;
#pragma unroll
        for (int i = 0; i < OFF; i++) {
            const double keep = up ? v[i + OFF] : v[i];
            const double send = up ? v[i] : v[i + OFF];
            v[i] = keep + xlane<OFF>(send, lane);
        }
    }
    if constexpr (OFF > 1) wave_sum_scatter_step<M, OFF / 2>(v, lane);
}
template <int M>
__device__ __forceinline__ double wave_sum_scatter(double (&v)[M], int lane) {
    static_assert(M >= 1 && M <= 64 && (M & (M - 1)) == 0, "a power of two up to the wave size");
    wave_sum_scatter_step<M, 32>(v, lane);
    return v[0];
}

// The window sums gL[NP], gH[NP] of every thread added up over the workgroup of T threads: thread t < 2 NP receives
// the sum of gL[t] (t < NP) or gH[t - NP]; the others receive 0.  red: (T / 64) * 2 NP doubles of LDS.  Contains barriers.
template <int NP, int T>
__device__ __forceinline__ double block_window_sums(const double (&gL)[NP], const double (&gH)[NP], double* red, int tid) {
    double v[2 * NP];
#pragma unroll
    for (int u = 0; u < NP; u++) { v[u] = gL[u]; v[NP + u] = gH[u]; }
    const int lane = tid & 63;
    const double mine = wave_sum_scatter<2 * NP>(v, lane);
    __syncthreads();
    if (lane < 2 * NP) red[(tid >> 6) * 2 * NP + lane] = mine;
    __syncthreads();
    double s = 0;
    if (tid < 2 * NP) {
#pragma unroll
        for (int wv = 0; wv < T / 64; wv++) s += red[wv * 2 * NP + tid];
    }
    return s;
}

// Sum of one value per thread over the workgroup, same result in every thread.
// Fixed order: lanes by shuffle tree, waves in index order.  `red` needs T/64 doubles.
template <int T>
__device__ __forceinline__ double block_sum(double v, double* red, int tid) {
    constexpr int NWV = T / 64;
    v = wave_sum(v);
    if constexpr (NWV == 1) {
        return __shfl(v, 0, 64);
    } else {
        __syncthreads();
        if ((tid & 63) == 0) red[tid >> 6] = v;
        __syncthreads();
        double s = 0;
#pragma unroll
        for (int w = 0; w < NWV; w++) s += red[w];
        return s;
    }
}

// ---------------------------------------------------------------------------------------------
// One level of the forward cascade for NL = 2^lgNL lines (src/Wavenet.cxx:423-434):
//   lo[k] = sum_j a[j] in[(N/2 + 2k - j) mod len],  hi[k] likewise with b.
// in/lo/hi are addressed as base + line * ls + index * es.
// ---------------------------------------------------------------------------------------------
template <int NP, int T>
__device__ __forceinline__ void fwd_level(const double* in, int es, int ls, double* lo, double* hi,
                                          int lgNL, int lgHalf, bool lanesOnLines,
                                          const double (&a)[NP], const double (&b)[NP], int tid) {
    const int half = 1 << lgHalf, lmask = 2 * half - 1;
    const int W = 1 << (lgNL + lgHalf);
    for (int w = tid; w < W; w += T) {
        int line, k;
        if (lanesOnLines) { line = w & ((1 << lgNL) - 1); k = w >> lgNL; }
        else              { k = w & (half - 1);           line = w >> lgHalf; }
        const double* src = in + line * ls;
        const int base = 2 * k + NP / 2;
        double sl = 0, sh = 0;
#pragma unroll
        for (int j = 0; j < NP; j++) {
            const double v = src[((base - j) & lmask) * es];
            sl = __builtin_fma(a[j], v, sl);
            sh = __builtin_fma(b[j], v, sh);
        }
        lo[line * ls + k * es] = sl;
        hi[line * ls + k * es] = sh;
    }
}

// Low-pass half only (rebuilds the row pyramid for the backward row pass).
template <int NP, int T>
__device__ __forceinline__ void fwd_level_lo(const double* in, int es, int ls, double* lo,
                                             int lgNL, int lgHalf, bool lanesOnLines,
                                             const double (&a)[NP], int tid) {
    const int half = 1 << lgHalf, lmask = 2 * half - 1;
    const int W = 1 << (lgNL + lgHalf);
    for (int w = tid; w < W; w += T) {
        int line, k;
        if (lanesOnLines) { line = w & ((1 << lgNL) - 1); k = w >> lgNL; }
        else              { k = w & (half - 1);           line = w >> lgHalf; }
        const double* src = in + line * ls;
        const int base = 2 * k + NP / 2;
        double sl = 0;
#pragma unroll
        for (int j = 0; j < NP; j++) sl = __builtin_fma(a[j], src[((base - j) & lmask) * es], sl);
        lo[line * ls + k * es] = sl;
    }
}

// Full-depth forward transform of NL lines of length 2^lgLen.
//   in : top-level input;  P : pyramid, P[2^i + k] = lo_i[k];  out : coefficients,
//   out[0] = lo_0, out[2^i + k] = hi_i[k] (src/Utilities.cxx:24-41).
// All three use element stride es and line stride ls.  Ends with a barrier.
template <int NP, int T>
__device__ __forceinline__ void fwd_pass(const double* in, double* P, double* out, int es, int ls,
                                         int lgNL, int lgLen, bool lanesOnLines,
                                         const double (&a)[NP], const double (&b)[NP], int tid) {
    if (lgLen == 0) {  // a line of one sample is its own coefficient
        for (int w = tid; w < (1 << lgNL); w += T) { out[w * ls] = in[w * ls]; P[w * ls] = 0; }
        __syncthreads();
        return;
    }
    for (int i = lgLen - 1; i >= 0; i--) {
        const double* src = (i == lgLen - 1) ? in : P + (2 << i) * es;
        fwd_level<NP, T>(src, es, ls, P + (1 << i) * es, out + (1 << i) * es, lgNL, i, lanesOnLines, a, b, tid);
        __syncthreads();
    }
    for (int w = tid; w < (1 << lgNL); w += T) { out[w * ls] = P[w * ls + es]; P[w * ls] = 0; }
    __syncthreads();
}

template <int NP, int T>
__device__ __forceinline__ void fwd_pass_lo(const double* in, double* P, int es, int ls, int lgNL,
                                            int lgLen, bool lanesOnLines, const double (&a)[NP], int tid) {
    // the backward pass never reads lo_0, so the cascade stops at level 1
    for (int i = lgLen - 1; i >= 1; i--) {
        const double* src = (i == lgLen - 1) ? in : P + (2 << i) * es;
        fwd_level_lo<NP, T>(src, es, ls, P + (1 << i) * es, lgNL, i, lanesOnLines, a, tid);
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// One level of the backward cascade (src/Wavenet.cxx:509-545) for NL lines.
//   D   : errors, in place: D[0..len) = delta_LP, D[len..2len) = delta_HP of this level;
//         on exit D[0..2len) = delta_LP of the next level.
//   act : activations(i+1, 0), length 2len.
// Filter-gradient sums are accumulated per thread, indexed by the position u of the
// activation inside the window of row r (column N/2 + 2r - u):
//   gL[u] += delta_LP[r] * act[..],   gH[u] += delta_HP[r] * act[..]
// so tap k of the filter collects gL[k] and (-1)^(N-1-k) gH[N-1-k].
// ---------------------------------------------------------------------------------------------
template <int NP, int T, int K, bool GRAD>
__device__ __forceinline__ void bwd_level(double* D, int es, int ls, const double* act, int lgNL,
                                          int lgLen, bool lanesOnLines, bool propagate,
                                          const double (&a)[NP], const double (&b)[NP],
                                          double (&gL)[NP], double (&gH)[NP], int tid) {
    const int len = 1 << lgLen, rmask = len - 1, amask = 2 * len - 1;
    const int W = 1 << (lgNL + lgLen);
    if constexpr (GRAD) {
        for (int w = tid; w < W; w += T) {
            int line, r;
            if (lanesOnLines) { line = w & ((1 << lgNL) - 1); r = w >> lgNL; }
            else              { r = w & rmask;                line = w >> lgLen; }
            const double dl = D[line * ls + r * es];
            const double dh = D[line * ls + (len + r) * es];
            const double* src = act + line * ls;
            const int base = 2 * r + NP / 2;
#pragma unroll
            for (int u = 0; u < NP; u++) {
                const double v = src[((base - u) & amask) * es];
                gL[u] = __builtin_fma(dl, v, gL[u]);
                gH[u] = __builtin_fma(dh, v, gH[u]);
            }
        }
    }
    if (!propagate) return;
    // new[c] = sum over (r, j) with (N/2 + 2r - j) mod 2len == c of a[j] dLP[r] + b[j] dHP[r]:
    // for c = 2p + q the taps are j = jq + 2i with jq = (N/2 - q) mod 2 and r = p + i + (q + jq - N/2)/2.
    constexpr int S = NP / 2;
    constexpr int JE = S & 1, JO = (S + 1) & 1;
    constexpr int CE = (JE - S) / 2, CO = (1 + JO - S) / 2;  // exact: both numerators are even
    constexpr int NIT = (K + 1) / 2;
    double ne[NIT], no[NIT];
#pragma unroll
    for (int it = 0; it < NIT; it++) {
        const int w = tid + it * T;
        ne[it] = 0;
        no[it] = 0;
        if (w < W) {
            int line, p;
            if (lanesOnLines) { line = w & ((1 << lgNL) - 1); p = w >> lgNL; }
            else              { p = w & rmask;                line = w >> lgLen; }
            const double* dl = D + line * ls;
            const double* dh = dl + len * es;
            double se = 0, so = 0;
#pragma unroll
            for (int i = 0; i < NP / 2; i++) {
                const int re = (p + i + CE) & rmask;
                const int ro = (p + i + CO) & rmask;
                se = __builtin_fma(a[JE + 2 * i], dl[re * es], se);
                se = __builtin_fma(b[JE + 2 * i], dh[re * es], se);
                so = __builtin_fma(a[JO + 2 * i], dl[ro * es], so);
                so = __builtin_fma(b[JO + 2 * i], dh[ro * es], so);
            }
            ne[it] = se;
            no[it] = so;
        }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < NIT; it++) {
        const int w = tid + it * T;
        if (w < W) {
            int line, p;
            if (lanesOnLines) { line = w & ((1 << lgNL) - 1); p = w >> lgNL; }
            else              { p = w & rmask;                line = w >> lgLen; }
            D[line * ls + (2 * p) * es] = ne[it];
            D[line * ls + (2 * p + 1) * es] = no[it];
        }
    }
    __syncthreads();
}

// Backward pass over NL lines of length 2^lgLen.  `in` is the line the forward pass
// started from (activations(m, 0)), P its pyramid.  With lastPropagate the errors on the
// input are left in D (needed by the row pass that follows, and by inverse transforms).
template <int NP, int T, int K, bool GRAD>
__device__ __forceinline__ void bwd_pass(double* D, const double* in, const double* P, int es, int ls,
                                         int lgNL, int lgLen, bool lanesOnLines, bool lastPropagate,
                                         const double (&a)[NP], const double (&b)[NP],
                                         double (&gL)[NP], double (&gH)[NP], int tid) {
    for (int i = 0; i < lgLen; i++) {
        const double* act = (i == lgLen - 1) ? in : P + (2 << i) * es;
        const bool prop = (i != lgLen - 1) || lastPropagate;
        bwd_level<NP, T, K, GRAD>(D, es, ls, act, lgNL, i, lanesOnLines, prop, a, b, gL, gH, tid);
    }
}

// ---------------------------------------------------------------------------------------------
// Register-blocked bitonic sort of T*K = 2^LGN doubles held in LDS at swz(e).
//
// A thread holds K = 2^Q keys whose indices differ in the Q bits of a "window"
// [bl, bl+Q).  Compare-exchanges on window bits are register-only; moving to the
// next window is one LDS store + barrier + load.  Windows are bl = 0, Q, 2Q, ...
// with the top one clamped to LGN-Q.  All compare-exchanges put the minimum at
// the lower index: each merge of size 2^KK starts by pairing e with its mirror
// e ^ (2^KK - 1); the registers of the upper half are loaded from (and stored
// back to) the mirrored addresses, which is a constant XOR because swz() is
// linear over GF(2).
// ---------------------------------------------------------------------------------------------
template <int LGN, int Q>
struct SortCfg {
    static constexpr int K = 1 << Q;
    static constexpr int NW = (LGN + Q - 1) / Q;
    static constexpr int wbl(int w) { return (w * Q < LGN - Q) ? w * Q : LGN - Q; }
    static constexpr int wtop(int KK) {
        int w = 0;
        while (wbl(w) + Q - 1 < KK - 1) w++;
        return w;
    }
};

template <int Q>
__device__ __forceinline__ void sort_regs(double (&v)[1 << Q]) {
    constexpr int K = 1 << Q;
#pragma unroll
    for (int kk = 1; kk <= Q; kk++) {
#pragma unroll
        for (int r = 0; r < K; r++) {
            if (!(r & (1 << (kk - 1)))) cmpx(v[r], v[r ^ ((1 << kk) - 1)]);
        }
#pragma unroll
        for (int bp = kk - 2; bp >= 0; bp--) {
#pragma unroll
            for (int r = 0; r < K; r++) {
                if (!(r & (1 << bp))) cmpx(v[r], v[r | (1 << bp)]);
            }
        }
    }
}

template <int LGN, int Q, int KK, int W>
__device__ __forceinline__ void sort_run(double* keys, unsigned t) {
    using Cfg = SortCfg<LGN, Q>;
    constexpr int K = 1 << Q;
    constexpr int bl = Cfg::wbl(W);
    constexpr bool first = (W == Cfg::wtop(KK));
    constexpr int hi = first ? KK - 1 : Cfg::wbl(W + 1) - 1;
    constexpr int tb = KK - 1 - bl;  // register bit of the merge's top bit (first run only)
    constexpr unsigned MIR = swz_c((1u << (KK - 1)) - 1u);
    const unsigned ebase = ((t >> bl) << (bl + Q)) | (t & ((1u << bl) - 1u));
    const unsigned pbase = swz(ebase);
    double v[K];
    unsigned addr[K];
#pragma unroll
    for (int r = 0; r < K; r++) {
        unsigned p = pbase ^ swz_c((unsigned)r << bl);
        if (first && ((r >> tb) & 1)) p ^= MIR;
        addr[r] = p;
        v[r] = keys[p];
    }
#pragma unroll
    for (int bp = hi - bl; bp >= 0; bp--) {
#pragma unroll
        for (int r = 0; r < K; r++) {
            if (r & (1 << bp)) continue;
            if (first && bp < tb && ((r >> tb) & 1)) cmpx(v[r | (1 << bp)], v[r]);
            else                                      cmpx(v[r], v[r | (1 << bp)]);
        }
    }
#pragma unroll
    for (int r = 0; r < K; r++) keys[addr[r]] = v[r];
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// Wave-local part of the sort.  With e = t*K + r the low Q bits of a key's index are the
// register, the next 6 bits the lane, the rest the wave.  Sub-stages on lane bits exchange
// keys between lanes with DPP / ds_swizzle / ds_bpermute (no LDS storage, no barrier): both
// lanes of a pair fetch the partner's key and keep the minimum or the maximum.  Only the
// sub-stages on wave bits go through LDS (sort_run on the top window).
// ---------------------------------------------------------------------------------------------
// CTRL: a permutation (every lane has a source).  Zero for "no source" plus bound_ctrl: the destination needs no
// initialising move.
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}

// v of lane (lane ^ M), M a compile-time mask.
template <int M>
__device__ __forceinline__ double xlane(double v, int lane) {
    if constexpr (M == 1) return dpp_mov<0xB1>(v);            // quad_perm [1,0,3,2]
    else if constexpr (M == 2) return dpp_mov<0x4E>(v);       // quad_perm [2,3,0,1]
    else if constexpr (M == 3) return dpp_mov<0x1B>(v);       // quad_perm [3,2,1,0]
    else if constexpr (M == 7) return dpp_mov<0x141>(v);      // row_half_mirror
    else if constexpr (M == 4) return dpp_mov<0x1B>(dpp_mov<0x141>(v));  // 7 ^ 3
    else if constexpr (M == 8) return dpp_mov<0x128>(v);      // row_ror:8
    else if constexpr (M == 15) return dpp_mov<0x140>(v);     // row_mirror
    else if constexpr (M < 32) {                              // ds_swizzle, bit mode: xor within 32 lanes
        int lo = __double2loint(v), hi = __double2hiint(v);
        lo = __builtin_amdgcn_ds_swizzle(lo, (M << 10) | 0x1F);
        hi = __builtin_amdgcn_ds_swizzle(hi, (M << 10) | 0x1F);
        return __hiloint2double(hi, lo);
    } else {
        const int addr = (lane ^ M) << 2;
        int lo = __double2loint(v), hi = __double2hiint(v);
        lo = __builtin_amdgcn_ds_bpermute(addr, lo);
        hi = __builtin_amdgcn_ds_bpermute(addr, hi);
        return __hiloint2double(hi, lo);
    }
}

// v summed over the wave by the butterfly at distances 32, 16, ... 1, the same bits in every lane -- the additions of
// a loop over __shfl_xor, with DPP moves instead of ds_bpermute for the distances below 16.
__device__ __forceinline__ double wave_allsum(double v, int lane) {
    v += xlane<32>(v, lane);
    v += xlane<16>(v, lane);
    v += xlane<8>(v, lane);
    v += xlane<4>(v, lane);
    v += xlane<2>(v, lane);
    v += xlane<1>(v, lane);
    return v;
}

// ---- the gradient's sums over the wave ------------------------------------------------------------------
// c[k], k < NP (a power of two up to 32), summed over the 64 lanes; lane l returns the total of c[l mod NP].  A
// butterfly that halves the set as it goes: on lane bit t a lane keeps the half of its values whose index has that
// bit equal to its own and hands the other half to its partner, so NP - 1 exchanges settle the NP sums over groups
// of NP lanes and one more per remaining lane bit completes them -- against six exchanges per value for a plain
// butterfly of every value.  A fixed order of additions: the same bits from run to run.
template <int NP, int CNT>
__device__ __forceinline__ void wave_scatter_step(double (&c)[NP], int lane) {
    if constexpr (CNT > 1) {
        constexpr int H = CNT / 2;
        const bool up = (lane & H) != 0;
#pragma unroll
        for (int i = 0; i < H; i++) {
            const double keep = up ? c[i + H] : c[i];
            const double send = up ? c[i] : c[i + H];
            c[i] = keep + xlane<H>(send, lane);
        }
        wave_scatter_step<NP, H>(c, lane);
    }
}
template <int NP>
__device__ __forceinline__ double wave_scatter_sums(double (&c)[NP], int lane) {
    static_assert(NP >= 1 && NP <= 32 && (NP & (NP - 1)) == 0, "a power of two up to 32");
    wave_scatter_step<NP, NP>(c, lane);
    double s = c[0];
    if constexpr (NP <= 1) s += xlane<1>(s, lane);
    if constexpr (NP <= 2) s += xlane<2>(s, lane);
    if constexpr (NP <= 4) s += xlane<4>(s, lane);
    if constexpr (NP <= 8) s += xlane<8>(s, lane);
    if constexpr (NP <= 16) s += xlane<16>(s, lane);
    s += xlane<32>(s, lane);
    return s;
}

// Half-cleaners on index bits B..0 (compare e with e ^ 2^b, minimum to the lower index).
template <int Q, int B>
__device__ __forceinline__ void wave_clean(double (&v)[1 << Q], int lane) {
    constexpr int K = 1 << Q;
    if constexpr (B >= Q) {
        constexpr int M = 1 << (B - Q);
        const bool is_lower = (lane & M) == 0;
#pragma unroll
        for (int r = 0; r < K; r++) {
            const double p = xlane<M>(v[r], lane);
            const bool take = (p < v[r]) == is_lower;
            v[r] = take ? p : v[r];
        }
    } else {
#pragma unroll
        for (int r = 0; r < K; r++) {
            if (!(r & (1 << B))) cmpx(v[r], v[r | (1 << B)]);
        }
    }
    if constexpr (B > 0) wave_clean<Q, B - 1>(v, lane);
}

// First sub-stage of the merge of size 2^KK inside a wave: e against e ^ (2^KK - 1).
template <int Q, int KK>
__device__ __forceinline__ void wave_mirror(double (&v)[1 << Q], int lane) {
    constexpr int K = 1 << Q;
    constexpr int M = (1 << (KK - Q)) - 1;
    const bool is_lower = ((lane >> (KK - 1 - Q)) & 1) == 0;
    double p[K];
#pragma unroll
    for (int r = 0; r < K; r++) p[r] = xlane<M>(v[r ^ (K - 1)], lane);
#pragma unroll
    for (int r = 0; r < K; r++) {
        const bool take = (p[r] < v[r]) == is_lower;
        v[r] = take ? p[r] : v[r];
    }
}

template <int LGN, int Q, int KK>
__device__ __forceinline__ void sort_wave_merges(double (&v)[1 << Q], int lane) {
    constexpr int WL = Q + 6;
    if constexpr (KK <= LGN && KK <= WL) {
        wave_mirror<Q, KK>(v, lane);
        wave_clean<Q, KK - 2>(v, lane);
        sort_wave_merges<LGN, Q, KK + 1>(v, lane);
    }
}

template <int LGN, int Q, int KK>
__device__ __forceinline__ void sort_cross_merges(double (&v)[1 << Q], double* keys, unsigned t, int lane) {
    using Cfg = SortCfg<LGN, Q>;
    constexpr int K = 1 << Q;
    constexpr int WL = Q + 6;
    if constexpr (KK <= LGN) {
        constexpr int W = Cfg::wtop(KK);
        static_assert(Cfg::wbl(W) <= WL, "the top sort window must reach down to the wave-local bits");
        const unsigned pbase = swz(t * K);
        __syncthreads();
#pragma unroll
        for (int r = 0; r < K; r++) keys[pbase ^ swz_c((unsigned)r)] = v[r];
        __syncthreads();
        sort_run<LGN, Q, KK, W>(keys, t);  // ends with a barrier
#pragma unroll
        for (int r = 0; r < K; r++) v[r] = keys[pbase ^ swz_c((unsigned)r)];
        wave_clean<Q, Cfg::wbl(W) - 1>(v, lane);
        sort_cross_merges<LGN, Q, KK + 1>(v, keys, t, lane);
    }
}

// Meeting point of the threads that work on one example.  WAVE: they are one wave, which is only one of several in
// its workgroup (each on an example of its own, so a workgroup barrier would wait for waves that never come): a
// wave's LDS operations execute in order, all it takes is that the compiler keeps them in order too.
template <bool WAVE>
__device__ __forceinline__ void group_sync() {
    if constexpr (WAVE) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    } else {
        __syncthreads();
    }
}

// Sorts T*K keys: v[r] is key t*K + r on entry; on exit keys[e] holds the e-th smallest.
template <int LGN, int Q, bool WAVE = false>
__device__ __forceinline__ void sort_full(double (&v)[1 << Q], double* keys, unsigned t) {
    static_assert(!WAVE || LGN == Q + 6, "one wave sorts its own keys without a workgroup barrier");
    constexpr int K = 1 << Q;
    constexpr int WL = Q + 6;
    const int lane = (int)(t & 63u);
    sort_regs<Q>(v);
    sort_wave_merges<LGN, Q, Q + 1>(v, lane);
    sort_cross_merges<LGN, Q, WL + 1>(v, keys, t, lane);
    group_sync<WAVE>();
#pragma unroll
    for (int r = 0; r < K; r++) keys[t * K + r] = v[r];  // final order is stored unswizzled
    group_sync<WAVE>();
}

// Number of keys strictly below v in the sorted buffer (v itself is in the buffer).
template <int LGN>
__device__ __forceinline__ unsigned lower_bound_sorted(const double* keys, double v) {
    unsigned pos = 0;
#pragma unroll
    for (int s = LGN - 1; s >= 0; s--) {
        const unsigned step = 1u << s;
        if (keys[pos + step - 1] < v) pos += step;
    }
    return pos;
}

// Utilities.h:69-70: sign(x) = x == 0 ? 0 : x / sqrt(x*x).  sqrt(x*x) == |x| exactly
// whenever x*x neither underflows nor overflows, so the quotient is +-1 there; outside
// that range the expression is evaluated as written.
__device__ __forceinline__ double sign_ref(double x) {
    const double ax = __builtin_fabs(x);
    if (ax >= 0x1p-500 && ax <= 0x1p500) return x > 0 ? 1.0 : -1.0;
    if (x == 0) return 0.0;
    return x / __builtin_sqrt(x * x);
}

// ---------------------------------------------------------------------------------------------
// Gini sparsity term and its gradient on one example (src/CostFunctions.cxx:6-37, 49-105).
//   S = 1 - f/g,  f = sum_i (2 i - n - 1) |c|_(i),  g = n sum |c|,
//   delta_j = -(df_j g - f dg_j) / g^2,  df_j = (2 rank_j - n - 1) sign(c_j),  dg_j = n sign(c_j).
// bufY holds the n coefficients in vectorise (column-major) order and T*K slots; on exit it
// holds delta in the same order when WRITE_DELTA.  Ranks of equal non-zero magnitudes
// ascend with the index (the reference's std::sort leaves that order unspecified).
// Returns S in every thread.  `red` needs T/64 doubles, `flag` one int.  WAVE: see group_sync (tid is then the lane).
// ---------------------------------------------------------------------------------------------
template <int T, int Q, bool WRITE_DELTA, bool WAVE = false>
__device__ __forceinline__ double gini_phase(double* bufY, int n, double* red, int* flag, int tid) {
    static_assert(!WAVE || T == 64, "WAVE: one wave per example");
    constexpr int K = 1 << Q;
    constexpr int LGN = __builtin_ctz(T) + Q;
    double yv[K], kv[K];
    const unsigned e0 = (unsigned)tid * K;
#pragma unroll
    for (int r = 0; r < K; r++) {
        const unsigned e = e0 + r;
        yv[r] = (e < (unsigned)n) ? bufY[e] : 0.0;
        kv[r] = (e < (unsigned)n) ? __builtin_fabs(yv[r]) : __builtin_inf();
    }
    if (tid == 0) *flag = 0;
    group_sync<WAVE>();
    {
        double sv[K];
#pragma unroll
        for (int r = 0; r < K; r++) sv[r] = kv[r];
        sort_full<LGN, Q, WAVE>(sv, bufY, (unsigned)tid);
    }

    unsigned lb[K];
    bool tie = false;
#pragma unroll
    for (int r = 0; r < K; r++) {
        lb[r] = 0;
        if (e0 + r < (unsigned)n) {
            lb[r] = lower_bound_sorted<LGN>(bufY, kv[r]);
            if (kv[r] != 0.0 && lb[r] + 1 < (1u << LGN) && bufY[lb[r] + 1] == kv[r]) tie = true;
        }
    }
    if (tie) *flag = 1;
    group_sync<WAVE>();
    if (*flag) {
        // Rare path: some non-zero magnitudes are equal.  Each tied group [lb, ub) of the
        // sorted buffer is reused as scratch: 32-bit word 0 counts members, words 1..mult list
        // their indices; a member's rank offset is the number of listed indices below its own.
        unsigned ub[K];
        bool mine[K];
        uint32_t* words = reinterpret_cast<uint32_t*>(bufY);
#pragma unroll
        for (int r = 0; r < K; r++) {
            ub[r] = lb[r] + 1;
            mine[r] = e0 + r < (unsigned)n && kv[r] != 0.0 && ub[r] < (1u << LGN) && bufY[ub[r]] == kv[r];
            if (mine[r]) {
                do { ub[r]++; } while (ub[r] < (1u << LGN) && bufY[ub[r]] == kv[r]);
            }
        }
        group_sync<WAVE>();
#pragma unroll
        for (int r = 0; r < K; r++) {
            if (mine[r]) words[2 * lb[r]] = 0u;
        }
        group_sync<WAVE>();
#pragma unroll
        for (int r = 0; r < K; r++) {
            if (mine[r]) {
                const unsigned slot = 1u + atomicAdd(&words[2 * lb[r]], 1u);
                words[2 * lb[r] + slot] = e0 + r;
            }
        }
        group_sync<WAVE>();
#pragma unroll
        for (int r = 0; r < K; r++) {
            if (mine[r]) {
                const unsigned mult = ub[r] - lb[r];
                unsigned below = 0;
                for (unsigned s = 1; s <= mult; s++) {
                    const unsigned other = words[2 * lb[r] + s];
                    below += (other < e0 + r) ? 1u : 0u;
                }
                lb[r] += below;
            }
        }
    }

    double fpart = 0, spart = 0;
    double coef[K];
#pragma unroll
    for (int r = 0; r < K; r++) {
        coef[r] = 2.0 * (double)(lb[r] + 1) - (double)n - 1.0;
        if (e0 + r < (unsigned)n) {
            fpart = __builtin_fma(coef[r], kv[r], fpart);
            spart += kv[r];
        }
    }
    const double f = block_sum<T>(fpart, red, tid);
    const double sabs = block_sum<T>(spart, red, tid);
    const double g = (double)n * sabs;
    if constexpr (WRITE_DELTA) {
        const double inv_g2 = 1.0 / (g * g);
        group_sync<WAVE>();
#pragma unroll
        for (int r = 0; r < K; r++) {
            if (e0 + r < (unsigned)n) {
                const double sg = sign_ref(yv[r]);
                const double df = coef[r] * sg;
                const double dg = (double)n * sg;
                bufY[e0 + r] = -(df * g - f * dg) * inv_g2;
            }
        }
        group_sync<WAVE>();
    }
    return 1.0 - f / g;
}

}  // namespace wn
