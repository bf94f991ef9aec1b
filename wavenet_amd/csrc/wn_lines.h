// Passes of the streamed path over lines of 128, 256, 512 or 1024 samples: ONE WAVE runs the full-depth
// cascade of one line in registers (M = 2, 4, 8 or 16 samples per lane), with the blocked / replicated
// level shapes and the cross-lane halos of wn_wave1d.h.  A line is read once and its coefficients written
// once (forward), or the errors and the line read once and the errors on the line written once (backward,
// which rebuilds the low-pass pyramid in registers instead of reading a stored one) -- against one read
// and one write per LEVEL in the level-by-level kernels.
//
// Columns are contiguous (kc_*).  Rows are strided by the number of rows: a workgroup moves 16 (8 for
// lines of 1024) adjacent rows through an LDS tile, 128 (64) contiguous bytes per column on the global
// side, one padded row per wave on the LDS side (kr_*).
//
// Reference stages: forward_ over the row and column lines (src/Wavenet.cxx:582-606), backpropagate_
// over them (src/Wavenet.cxx:680-702).
#pragma once

#include "wn_wave1d.h"

namespace wn {

constexpr int LN_GX = 64;   // workgroups per example; equals GXMAX of the streamed path's partial sums

// One line's values in the lane layout of the cascade: v[0] = index lane (the 64 values of the levels
// below 64; index 0 is lo_0 for coefficients, 0 for the pyramid), and for c = 1, 2, ..., M/2:
// v[c + m] = index 64 c + c lane + m, m < c (the level with c values per lane).
template <int M>
struct LineLane {
    double v[M];
};
// x: the line, M samples per lane; lo: its low-pass pyramid in the layout of LineLane (lo[0] unused);
// los[i]: level i < 6 at index lane mod 2^i.
template <int M>
struct LinePyramid {
    double x[M], lo[M], los[6];
};

template <int NP, int M, int Cc>   // the level with Cc outputs per lane, then the ones below it
__device__ __forceinline__ void ln_fwd_blocked(LinePyramid<M>& p, LineLane<M>& y, const double (&a)[NP], const double (&b)[NP],
                                               int lane) {
    if constexpr (2 * Cc == M) w1d_fwd_block<NP, 2 * Cc>(p.x, &p.lo[Cc], &y.v[Cc], a, b, lane);
    else w1d_fwd_block<NP, 2 * Cc>(&p.lo[2 * Cc], &p.lo[Cc], &y.v[Cc], a, b, lane);
    if constexpr (Cc > 1) ln_fwd_blocked<NP, M, Cc / 2>(p, y, a, b, lane);
}

// The cascade of one line: p.x in, pyramid left in p, coefficients in y; small_lo: the pyramid's value at
// index lane (layout of the level kernels: level i at [2^i, 2^(i+1)), Y[0] = lo_0, P[0] = 0).
template <int NP, int M>
__device__ __forceinline__ void ln_forward(LinePyramid<M>& p, const double (&a)[NP], const double (&b)[NP], int lane,
                                           LineLane<M>& y, double& small_lo_out) {
    ln_fwd_blocked<NP, M, M / 2>(p, y, a, b, lane);
    const double lo6 = p.lo[1];
    // levels below 64 values, replicated: lane 2^i + k holds low-pass and high-pass value k of level i
    double his, small_hi = 0, small_lo = 0;
    w1d_fwd_small<NP, 5>(lo6, p.los[5], his, a, b, lane);
    if (lane >= 32) { small_hi = his; small_lo = p.los[5]; }
    w1d_fwd_small<NP, 4>(p.los[5], p.los[4], his, a, b, lane);
    if (lane >= 16 && lane < 32) { small_hi = his; small_lo = p.los[4]; }
    w1d_fwd_small<NP, 3>(p.los[4], p.los[3], his, a, b, lane);
    if (lane >= 8 && lane < 16) { small_hi = his; small_lo = p.los[3]; }
    w1d_fwd_small<NP, 2>(p.los[3], p.los[2], his, a, b, lane);
    if (lane >= 4 && lane < 8) { small_hi = his; small_lo = p.los[2]; }
    w1d_fwd_small<NP, 1>(p.los[2], p.los[1], his, a, b, lane);
    if (lane >= 2 && lane < 4) { small_hi = his; small_lo = p.los[1]; }
    w1d_fwd_small<NP, 0>(p.los[1], p.los[0], his, a, b, lane);
    if (lane == 1) { small_hi = his; small_lo = p.los[0]; }
    if (lane == 0) { small_hi = p.los[0]; small_lo = 0; }   // Y[0] = lo_0, P[0] = 0
    y.v[0] = small_hi;
    small_lo_out = small_lo;
}

template <int NP, int M, int Cr, bool PROP>   // the level with Cr errors per lane, then the ones above it
__device__ __forceinline__ void ln_bwd_blocked(const LinePyramid<M>& p, const LineLane<M>& dy, double* dbuf, double* dtop,
                                               const double (&a)[NP], const double (&b)[NP], double (&gL)[NP], double (&gH)[NP],
                                               int lane) {
    if constexpr (2 * Cr == M) {
        w1d_bwd_block<NP, Cr, PROP>(&dbuf[Cr], &dy.v[Cr], p.x, dtop, a, b, gL, gH, lane);
    } else {
        w1d_bwd_block<NP, Cr, true>(&dbuf[Cr], &dy.v[Cr], &p.lo[2 * Cr], &dbuf[2 * Cr], a, b, gL, gH, lane);
        ln_bwd_blocked<NP, M, 2 * Cr, PROP>(p, dy, dbuf, dtop, a, b, gL, gH, lane);
    }
}

// Backward over one line: errors on its coefficients (lane layout) and its pyramid -> window sums of the
// filter gradient; with PROP the errors on the line itself, dtop[M] per lane.
template <int NP, int M, bool PROP>
__device__ __forceinline__ void ln_backward(const LinePyramid<M>& p, const LineLane<M>& dy, double* dtop, const double (&a)[NP],
                                            const double (&b)[NP], double (&gL)[NP], double (&gH)[NP], int lane) {
    const double dsmall = dy.v[0];
    double d = w1d_lane(dsmall, 0);                       // lo_0
    d = w1d_bwd_small<NP, 0>(d, w1d_lane(dsmall, 1), p.los[1], a, b, gL, gH, lane);
    d = w1d_bwd_small<NP, 1>(d, w1d_lane(dsmall, 2 + (lane & 1)), p.los[2], a, b, gL, gH, lane);
    d = w1d_bwd_small<NP, 2>(d, w1d_lane(dsmall, 4 + (lane & 3)), p.los[3], a, b, gL, gH, lane);
    d = w1d_bwd_small<NP, 3>(d, w1d_lane(dsmall, 8 + (lane & 7)), p.los[4], a, b, gL, gH, lane);
    d = w1d_bwd_small<NP, 4>(d, w1d_lane(dsmall, 16 + (lane & 15)), p.los[5], a, b, gL, gH, lane);
    d = w1d_bwd_small<NP, 5>(d, w1d_lane(dsmall, 32 + (lane & 31)), p.lo[1], a, b, gL, gH, lane);
    double dbuf[M];      // dbuf[c + m]: errors on the low-pass level with c values per lane
    dbuf[1] = d;
    ln_bwd_blocked<NP, M, 1, PROP>(p, dy, dbuf, dtop, a, b, gL, gH, lane);
}

// contiguous line in memory <-> lane layout / samples
template <int M>
__device__ __forceinline__ void ln_store(double* Y, const double* v, double small, int lane) {
    Y[lane] = small;
    Y[64 + lane] = v[1];
#pragma unroll
    for (int c = 2; c < M; c *= 2) {
        double2* t = reinterpret_cast<double2*>(Y + 64 * c + c * lane);
#pragma unroll
        for (int m = 0; m < c; m += 2) t[m / 2] = make_double2(v[c + m], v[c + m + 1]);
    }
}
template <int M>
__device__ __forceinline__ void ln_load(const double* Y, LineLane<M>& d, int lane) {
    d.v[0] = Y[lane];
    d.v[1] = Y[64 + lane];
#pragma unroll
    for (int c = 2; c < M; c *= 2) {
        const double2* t = reinterpret_cast<const double2*>(Y + 64 * c + c * lane);
#pragma unroll
        for (int m = 0; m < c; m += 2) { const double2 u = t[m / 2]; d.v[c + m] = u.x; d.v[c + m + 1] = u.y; }
    }
}
template <int M>
__device__ __forceinline__ void ln_load_samples(const double* src, double* x, int lane) {
    const double2* s = reinterpret_cast<const double2*>(src + M * lane);
#pragma unroll
    for (int i = 0; i < M / 2; i++) { const double2 t = s[i]; x[2 * i] = t.x; x[2 * i + 1] = t.y; }
}
template <int M>
__device__ __forceinline__ void ln_store_samples(double* dst, const double* x, int lane) {
    double2* s = reinterpret_cast<double2*>(dst + M * lane);
#pragma unroll
    for (int i = 0; i < M / 2; i++) s[i] = make_double2(x[2 * i], x[2 * i + 1]);
}

template <int NP>
__device__ __forceinline__ void ln_load_taps(const FilterDev* filt, double (&a)[NP], double (&b)[NP]) {
#pragma unroll
    for (int j = 0; j < NP; j++) { a[j] = filt->a[j]; b[j] = filt->b[j]; }
    if constexpr (NP <= 16) {   // the same in every lane: scalar registers
#pragma unroll
        for (int j = 0; j < NP; j++) { a[j] = w1d_uniform(a[j]); b[j] = w1d_uniform(b[j]); }
    }
}

// The error on one coefficient from its value and rank (kb_delta_home of the streamed path, same expression).
struct LineGini {
    double f, g, inv_g2, n;
    __device__ __forceinline__ double delta(double y, unsigned rank) const {
        const double sg = sign_ref(y);
        const double coef = (y != 0.0) ? 2.0 * (double)rank - n - 1.0 : 0.0;
        const double df = coef * sg;
        return -(df * g - f * (n * sg)) * inv_g2;
    }
};

// From 16 taps on the kernels are compiled for one wave per SIMD (512 registers): at two, the backward kernels
// of 16 taps keep 300 to 750 bytes per lane in scratch memory and run a quarter to a half slower.
// Errors on the coefficients of the levels with CC, 2 CC, ... values per lane from their ranks: a lane's CC ranks of
// a level are consecutive words (8- and 16-byte loads).
template <int M, int CC>
__device__ __forceinline__ void ln_delta_levels(LineLane<M>& dy, const LineGini& gini, const unsigned* rc, int lane) {
    if constexpr (CC < M) {
        if constexpr (CC == 2) {
            const uint2 r = *reinterpret_cast<const uint2*>(rc + 128 + 2 * lane);
            dy.v[2] = gini.delta(dy.v[2], r.x);
            dy.v[3] = gini.delta(dy.v[3], r.y);
        } else {
#pragma unroll
            for (int m = 0; m < CC; m += 4) {
                const uint4 r = *reinterpret_cast<const uint4*>(rc + 64 * CC + CC * lane + m);
                dy.v[CC + m] = gini.delta(dy.v[CC + m], r.x);
                dy.v[CC + m + 1] = gini.delta(dy.v[CC + m + 1], r.y);
                dy.v[CC + m + 2] = gini.delta(dy.v[CC + m + 2], r.z);
                dy.v[CC + m + 3] = gini.delta(dy.v[CC + m + 3], r.w);
            }
        }
        ln_delta_levels<M, 2 * CC>(dy, gini, rc, lane);
    }
}

#ifndef LN_ONE_WAVE_FROM
#define LN_ONE_WAVE_FROM 16
#endif
#define LN_BOUNDS(T) __launch_bounds__(T, (NP >= LN_ONE_WAVE_FROM ? 1 : 2))

// The thread index as a value the compiler cannot see through.  Taken at the top of every pass of a kernel's loop
// over lines or tiles: the addresses derived from it are then computed where they are used instead of being
// hoisted out of the loop and held in registers (or scratch memory) all the way through.
__device__ __forceinline__ int ln_opaque(int v) {
    asm volatile("" : "+v"(v));
    return v;
}

// ---- columns: contiguous lines of 64 M samples ---------------------------------------------------------
// grid (workgroups per example, examples), 256 threads = 4 waves, each on columns of its own.  in/Y/P: chunk arrays, example stride xs, column c
// of an example at c * 64 M.  P may be null.  counts (may be null): counts[i] receives (is added) the number
// of non-zero coefficients among the chunk's coefficients [2048 i, 2048 (i + 1)) -- what the streamed Gini
// phase needs first, gathered while the coefficients are still in registers.  kmax (may be null): kmax[example]
// is raised to the largest exponent word among the example's coefficients.
template <int NP, int M>
__global__ LN_BOUNDS(256) void kc_fwd_cols(const double* __restrict__ in, double* __restrict__ Y, double* __restrict__ P, size_t xs,
                                           int ncols, const FilterDev* filt, unsigned* counts, unsigned* kmax) {
    // four independent waves per workgroup (one per SIMD: waves of one-wave workgroups are not spread as evenly)
    const int lane0 = threadIdx.x & 63;
    const int slot = blockIdx.x * 4 + (threadIdx.x >> 6), nslots = gridDim.x * 4;
    double a[NP], b[NP];
    ln_load_taps<NP>(filt, a, b);
    const size_t off = (size_t)blockIdx.y * xs;
    unsigned kmine = 0;
    for (int c = slot; c < ncols; c += nslots) {
        const int lane = ln_opaque(lane0);
        const size_t lo = off + (size_t)c * (64 * M);
        LinePyramid<M> p;
        LineLane<M> y;
        double small_lo;
        ln_load_samples<M>(in + lo, p.x, lane);
        ln_forward<NP, M>(p, a, b, lane, y, small_lo);
        ln_store<M>(Y + lo, y.v, y.v[0], lane);
        if (P) ln_store<M>(P + lo, p.lo, small_lo, lane);
        if (counts) {
            unsigned nz = 0;
#pragma unroll
            for (int k = 0; k < M; k++) nz += (unsigned)__popcll(__ballot(y.v[k] != 0.0));
            if (lane == 0) atomicAdd(&counts[lo / 2048], nz);
        }
        if (kmax) {   // the largest exponent word (upper half of |c|) so far, for the ranking's histogram
#pragma unroll
            for (int k = 0; k < M; k++) kmine = max(kmine, (unsigned)__double2hiint(y.v[k]) & 0x7FFFFFFFu);
        }
    }
    if (kmax) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) kmine = max(kmine, (unsigned)__shfl_xor((int)kmine, o, 64));
        if (lane0 == 0 && kmine) atomicMax(&kmax[blockIdx.y], kmine);
    }
}

// kc_fwd_cols with the first step of the bucket ranking fused in: ONE workgroup of 16 waves per example writes the
// coefficients and, from the registers they are still in, the example's histogram over the ranking's NF logarithmic
// bins (wn_rank.h: fine_bin, counted down from the example's largest magnitude) and that largest magnitude's
// exponent word -- what kr_max and kr_hist would otherwise read the coefficients back for.
// The largest magnitude is only known at the end, so the workgroup counts over LN_HIST_BINS bins in LDS counted
// down from a bound that is known beforehand: the largest exponent word of the input (the maximum of the nparts
// words per example in kmax_in, left by the waves of the kernel that wrote the input) raised by LEVELS * log2(max(1, sum |a_j|, sum |b_j|)) octaves and one more.  Both
// binnings cut the same absolute grid (exponent word >> shift), so the histogram below the true maximum is the
// LDS one moved by (bound - maximum) bins, exactly.  When the bound lies more than LN_HIST_BINS - NF bins above
// the maximum (or anything exceeds it: non-finite values), need[example] is set instead and kr_hist does that
// example.  grid (examples), LN_HIST_T threads, LN_HIST_BINS * 4 bytes of dynamic LDS.
constexpr int LN_HIST_T = 1024;
constexpr int LN_HIST_BINS = 36864;
template <int NP, int M>
__global__ __launch_bounds__(LN_HIST_T) void kc_fwd_cols_hist(const double* __restrict__ in, double* __restrict__ Y, size_t xs, int ncols,
                                                              const FilterDev* filt, const unsigned* __restrict__ kmax_in, int nparts,
                                                              unsigned* kmax, unsigned* need, unsigned* hist, int NF, int lgBPO) {
    constexpr int LEVELS = 6 + (M == 2 ? 1 : M == 4 ? 2 : M == 8 ? 3 : 4);
    extern __shared__ unsigned ln_hist[];
    __shared__ unsigned wred[LN_HIST_T / 64];
    __shared__ unsigned tail_sum, bad_any;
    const int tid = threadIdx.x, lane0 = tid & 63, wave = tid >> 6, e = blockIdx.x;
    for (int j = tid; j < LN_HIST_BINS; j += LN_HIST_T) ln_hist[j] = 0;
    if (tid == 0) { tail_sum = 0; bad_any = 0; }
    double a[NP], b[NP];
    ln_load_taps<NP>(filt, a, b);
    double sa = 0, sb = 0;
#pragma unroll
    for (int j = 0; j < NP; j++) { sa += fabs(a[j]); sb += fabs(b[j]); }
    const double gain = fmax(1.0, fmax(sa, sb));
    const double octs = ceil((double)LEVELS * log2(gain) + 1e-6) + 1.0;
    const int shift = 20 - lgBPO;
    unsigned badv = 0;
    // the largest exponent word of the input: the maximum over the words its producer's waves left
    unsigned kin = 0;
    for (int i = tid; i < nparts; i += LN_HIST_T) kin = max(kin, kmax_in[(size_t)e * nparts + i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) kin = max(kin, (unsigned)__shfl_xor((int)kin, o, 64));
    if (lane0 == 0) wred[wave] = kin;
    __syncthreads();
#pragma unroll
    for (int w = 0; w < LN_HIST_T / 64; w++) kin = max(kin, wred[w]);
    unsigned kprov = 0x7FEFFFFFu;
    if (octs < 1024.0) {
        const unsigned long long kp = (unsigned long long)kin + ((unsigned long long)octs << 20);
        if (kp < 0x7FF00000ull) kprov = (unsigned)kp;
    }
    const unsigned Lprov = kprov >> shift;
    __syncthreads();                      // (wred is written again at the end)
    unsigned kmine = 0;
    const size_t off = (size_t)e * xs;
    for (int c = wave; c < ncols; c += LN_HIST_T / 64) {
        const int lane = ln_opaque(lane0);
        const size_t lo = off + (size_t)c * (64 * M);
        LinePyramid<M> p;
        LineLane<M> y;
        double small_lo;
        ln_load_samples<M>(in + lo, p.x, lane);
        ln_forward<NP, M>(p, a, b, lane, y, small_lo);
        ln_store<M>(Y + lo, y.v, y.v[0], lane);
#pragma unroll
        for (int k = 0; k < M; k++) {
            const unsigned key = (unsigned)__double2hiint(y.v[k]) & 0x7FFFFFFFu;
            kmine = max(kmine, key);
            if (y.v[k] != 0.0) {
                const unsigned L = key >> shift;
                if (L > Lprov) badv = 1;
                const unsigned d = (L > Lprov) ? 0u : Lprov - L;
                atomicAdd(&ln_hist[d < (unsigned)(LN_HIST_BINS - 1) ? d : (unsigned)(LN_HIST_BINS - 1)], 1u);
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) kmine = max(kmine, (unsigned)__shfl_xor((int)kmine, o, 64));
    if (lane0 == 0) wred[wave] = kmine;
    if (badv) bad_any = 1u;
    __syncthreads();
    unsigned kmaxW = 0;
#pragma unroll
    for (int w = 0; w < LN_HIST_T / 64; w++) kmaxW = max(kmaxW, wred[w]);
    const unsigned Lmax = kmaxW >> shift;
    const bool invalid = bad_any != 0 || Lmax > Lprov || Lprov - Lmax > (unsigned)(LN_HIST_BINS - NF);
    if (tid == 0) {
        kmax[e] = kmaxW;
        need[e] = invalid ? 1u : 0u;
    }
    if (invalid) return;
    const unsigned o = Lprov - Lmax;
    unsigned* out = hist + (size_t)e * NF;
    unsigned t = 0;      // bin 0 takes all that lies NF - 1 bins or more below the maximum
    for (unsigned j = o + (unsigned)NF - 1u + (unsigned)tid; j < (unsigned)LN_HIST_BINS; j += LN_HIST_T) t += ln_hist[j];
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) t += (unsigned)__shfl_xor((int)t, s, 64);
    if (lane0 == 0 && t) atomicAdd(&tail_sum, t);
    for (int f = 1 + tid; f < NF; f += LN_HIST_T) out[f] = ln_hist[o + (unsigned)(NF - 1 - f)];
    __syncthreads();
    if (tid == 0) out[0] = tail_sum;
}

// Errors on the coefficients of the columns and the columns themselves (in) -> errors on the columns
// (out, may be null: last pass) and the window sums of the filter gradient, added to
// partial[(example * LN_GX + workgroup) * 2 NP + u] as the level kernels do.  The errors on the
// coefficients are read from coef, or with FROM_RANKS computed on the way in from the coefficients
// (recomputed from the columns; coef is not read), their ranks and the example's sums fg = {f, g}: the array of
// errors is then never written.
// ranks: per slice of `slice` examples an array of slice * n ranks at ranks + slice_first_example * n
// DOUBLES (the streamed path's layout).
#ifndef LN_BWD_COLS_WAVES
#define LN_BWD_COLS_WAVES 2
#endif
#ifndef LN_BWD_COLS_ONE_WAVE_FROM
#define LN_BWD_COLS_ONE_WAVE_FROM 32      // (16 taps: 268 registers at one wave per SIMD; at two, 36 bytes of scratch and a quarter faster)
#endif
template <int NP, int M, bool FROM_RANKS>
__global__ __launch_bounds__(256, (NP >= LN_BWD_COLS_ONE_WAVE_FROM ? 1 : LN_BWD_COLS_WAVES)) void kc_bwd_cols(const double* __restrict__ coef, const double* __restrict__ in, double* __restrict__ out,
                                           size_t xs, int ncols, const FilterDev* filt, double* partial, const double* ranks,
                                           int slice, const double* fg) {
    // four independent waves per workgroup, each with a partial-sum slot of its own (grid.x * 4 <= LN_GX)
    const int lane0 = threadIdx.x & 63;
    const int slot = blockIdx.x * 4 + (threadIdx.x >> 6), nslots = gridDim.x * 4;
    double a[NP], b[NP];
    ln_load_taps<NP>(filt, a, b);
    double gL[NP], gH[NP];
#pragma unroll
    for (int u = 0; u < NP; u++) { gL[u] = 0; gH[u] = 0; }
    const size_t off = (size_t)blockIdx.y * xs;
    LineGini gini{0, 0, 0, (double)xs};
    const unsigned* rk = nullptr;
    if constexpr (FROM_RANKS) {
        const int e = blockIdx.y, e0 = (e / slice) * slice;
        rk = reinterpret_cast<const unsigned*>(ranks + (size_t)e0 * xs) + (size_t)(e - e0) * xs;
        gini.f = fg[2 * e];
        gini.g = fg[2 * e + 1];
        gini.inv_g2 = 1.0 / (gini.g * gini.g);
    }
    for (int c = slot; c < ncols; c += nslots) {
        const int lane = ln_opaque(lane0);
        const size_t lo = off + (size_t)c * (64 * M);
        LinePyramid<M> p;
        LineLane<M> dy;
        double unused_lo;
        ln_load_samples<M>(in + lo, p.x, lane);
        // FROM_RANKS: the coefficients come out of the cascade that rebuilds the pyramid -- the very instructions that
        // produced the stored ones, so the same bits -- and are not read from memory
        ln_forward<NP, M>(p, a, b, lane, dy, unused_lo);
        if constexpr (!FROM_RANKS) ln_load<M>(coef + lo, dy, lane);          // errors on the coefficients
        if constexpr (FROM_RANKS) {
            const unsigned* rc = rk + (size_t)c * (64 * M);
            dy.v[0] = gini.delta(dy.v[0], rc[lane]);
            dy.v[1] = gini.delta(dy.v[1], rc[64 + lane]);
            ln_delta_levels<M, 2>(dy, gini, rc, lane);
        }
        double dtop[M];
        if (out) {
            ln_backward<NP, M, true>(p, dy, dtop, a, b, gL, gH, lane);
            ln_store_samples<M>(out + lo, dtop, lane);
        } else {
            ln_backward<NP, M, false>(p, dy, dtop, a, b, gL, gH, lane);
        }
    }
    double* part = partial + ((size_t)blockIdx.y * LN_GX + slot) * 2 * NP;
#pragma unroll
    for (int u = 0; u < NP; u++) {
        double sl = gL[u], sh = gH[u];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { sl += __shfl_xor(sl, o, 64); sh += __shfl_xor(sh, o, 64); }
        if (lane0 == 0) { part[u] += sl; part[NP + u] += sh; }
    }
}

// kc_bwd_cols from the bucket ranking's (home index, rank) pairs instead of ranks at home positions.  The ranking then
// writes consecutive addresses (one 32-bit pair per slot of a bucket: home index mod region in the upper ib bits, the
// rank below, which takes examples of up to 2^(31 - ib) coefficients) instead of one cache line per rank, which is
// what bounds its last kernel: 537 M four-byte stores per step of 2048 images of 512x512 are 2 ms of L2 requests.
// A bucket's slots are ordered by region of 2^ib home positions (a whole number of columns), and the partition's
// cursor snapshots say where a region's run begins in every bucket: a workgroup gathers the runs of its region
// into LDS -- descriptors first, then sixteen loads in flight per wave -- which puts the region's ranks at
// (home index mod region), and then works through the region's columns as kc_bwd_cols does, four waves on columns
// of their own, the ranks read from LDS.  Layout of the ranking's tables (wn_rank.h): per example meta_stride words,
// word 1 = number of buckets, words 4 .. = bucket starts; snap[(example * regions + region) * nbe + bucket] = 16-bit
// offset of the region's first slot in the bucket; a bucket of more than LN_BUCKET_CAP slots has raised the flag.
// grid (workgroups per example: 4 * grid.x <= LN_GX, examples), 256 threads, dynamic LDS (4 << ib) + 6 * nbe bytes.
constexpr unsigned LN_BUCKET_CAP = 4096;
#ifndef LN_RUNS
#define LN_RUNS 8        // runs a wave fetches at a time: 16 loads in flight
#endif
template <int NP, int M>
__global__ __launch_bounds__(256, (NP >= LN_BWD_COLS_ONE_WAVE_FROM ? 1 : LN_BWD_COLS_WAVES)) void kc_bwd_cols_regions(
    const double* __restrict__ in, double* __restrict__ out, size_t xs, int ncols, const FilterDev* filt, double* partial,
    const unsigned* __restrict__ pairs, const unsigned short* __restrict__ snap, const unsigned* __restrict__ meta_all,
    int meta_stride, int nbe, int ib, const double* fg, const unsigned* flag) {
    extern __shared__ unsigned ln_rk[];
    const int lane0 = threadIdx.x & 63, wave0 = threadIdx.x >> 6;
    const int slot = blockIdx.x * 4 + wave0;
    double a[NP], b[NP];
    ln_load_taps<NP>(filt, a, b);
    double gL[NP], gH[NP];
#pragma unroll
    for (int u = 0; u < NP; u++) { gL[u] = 0; gH[u] = 0; }
    const int e = blockIdx.y;
    const size_t off = (size_t)e * xs;
    LineGini gini{fg[2 * e], fg[2 * e + 1], 0, (double)xs};
    gini.inv_g2 = 1.0 / (gini.g * gini.g);
    const int regions = (int)(xs >> ib), cpr = (1 << ib) / (64 * M);      // columns per region
    const unsigned* meta = meta_all + (size_t)e * meta_stride;
    const unsigned* bstart = meta + 4;
    const int nb = (*flag) ? 0 : min((int)meta[1], nbe);
    unsigned* run0 = ln_rk + (1u << ib);
    unsigned short* runn = reinterpret_cast<unsigned short*>(run0 + nbe);
    const int rb = 32 - ib;                               // a pair: (home index mod region) << rb | rank
    const unsigned rmask = (1u << rb) - 1u;
    const unsigned* pe = pairs + off;
    for (int r = blockIdx.x; r < regions; r += gridDim.x) {
        const int tid = ln_opaque((int)threadIdx.x), lane = tid & 63, wave = tid >> 6;
        __syncthreads();                                   // the ranks of the region before are done with
        const unsigned short* s0 = snap + ((size_t)e * regions + r) * nbe;
        for (int bk = tid; bk < nb; bk += 256) {
            const unsigned bs = bstart[bk], count = bstart[bk + 1] - bs;
            unsigned lo = 0, hi = 0;
            if (count <= LN_BUCKET_CAP) {
                lo = min((unsigned)s0[bk], count);
                hi = (r + 1 < regions) ? min((unsigned)s0[nbe + bk], count) : count;
                hi = max(hi, lo);
            }
            run0[bk] = bs + lo;
            runn[bk] = (unsigned short)(hi - lo);
        }
        __syncthreads();
        for (int b0 = wave; b0 < nb; b0 += 4 * LN_RUNS) {
            unsigned v[LN_RUNS][2];
            unsigned len[LN_RUNS], base[LN_RUNS];
#pragma unroll
            for (int q = 0; q < LN_RUNS; q++) {
                const int bk = b0 + 4 * q;
                len[q] = (bk < nb) ? (unsigned)runn[bk] : 0u;
                base[q] = (bk < nb) ? run0[bk] : 0u;
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const unsigned j = (unsigned)lane + 64u * (unsigned)h;
                    v[q][h] = pe[base[q] + (j < len[q] ? j : 0u)];
                }
            }
#pragma unroll
            for (int q = 0; q < LN_RUNS; q++) {
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    if ((unsigned)lane + 64u * (unsigned)h < len[q]) ln_rk[v[q][h] >> rb] = v[q][h] & rmask;
                }
                for (unsigned j = 128u + (unsigned)lane; j < len[q]; j += 64u) {
                    const unsigned w = pe[base[q] + j];
                    ln_rk[w >> rb] = w & rmask;
                }
            }
        }
        __syncthreads();
        for (int cl = wave; cl < cpr; cl += 4) {
            const int c = r * cpr + cl;
            if (c >= ncols) break;
            const size_t lo = off + (size_t)c * (64 * M);
            LinePyramid<M> p;
            LineLane<M> dy;
            double unused_lo;
            ln_load_samples<M>(in + lo, p.x, lane);
            ln_forward<NP, M>(p, a, b, lane, dy, unused_lo);       // the coefficients, recomputed (kc_bwd_cols)
            const unsigned* rc = ln_rk + cl * (64 * M);
            dy.v[0] = gini.delta(dy.v[0], rc[lane]);
            dy.v[1] = gini.delta(dy.v[1], rc[64 + lane]);
            ln_delta_levels<M, 2>(dy, gini, rc, lane);
            double dtop[M];
            if (out) {
                ln_backward<NP, M, true>(p, dy, dtop, a, b, gL, gH, lane);
                ln_store_samples<M>(out + lo, dtop, lane);
            } else {
                ln_backward<NP, M, false>(p, dy, dtop, a, b, gL, gH, lane);
            }
        }
    }
    double* part = partial + ((size_t)blockIdx.y * LN_GX + slot) * 2 * NP;
#pragma unroll
    for (int u = 0; u < NP; u++) {
        double sl = gL[u], sh = gH[u];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { sl += __shfl_xor(sl, o, 64); sh += __shfl_xor(sh, o, 64); }
        if (lane0 == 0) { part[u] += sl; part[NP + u] += sh; }
    }
}

// ---- rows: lines of 64 M samples strided by the number of rows -------------------------------------------
// Tile: ROWS rows x 64 M columns in LDS, row stride RS doubles, element c of a row at c + c / M: a lane's M
// samples sit M + 1 doubles from its neighbour's (conflict-free 8-byte reads), and the row pairs x columns
// that half a wave moves on the global side fall into 32 different banks (row stride = 2 or 4 mod 32).
template <int M>
struct LineTile {
    static constexpr int ROWS = (M == 16) ? 8 : 16;
    static constexpr int RS = 64 * (M + 1) + (ROWS == 16 ? 2 : 4);
    static constexpr size_t BYTES = (size_t)ROWS * RS * sizeof(double);
    static constexpr int PAIRS = ROWS / 2, COLS_PER_STEP = 256 / PAIRS, STEPS = 64 * M / COLS_PER_STEP;
    __device__ static __forceinline__ int pad(int c) { return c + c / M; }
};

// ROWS adjacent rows of one example between memory (element (r, c) at r + R c) and the tile; 256 threads
template <int M, bool TO_TILE>
__device__ __forceinline__ void ln_move_tile(double* tile, double* mem, int R, int tid) {
    using T = LineTile<M>;
    const int rp = tid % T::PAIRS;
#ifndef LN_MOVE_UNROLL
#define LN_MOVE_UNROLL 4
#endif
#pragma unroll LN_MOVE_UNROLL
    for (int it = 0; it < T::STEPS; it++) {
        const int c = tid / T::PAIRS + T::COLS_PER_STEP * it;
        double2* g = reinterpret_cast<double2*>(mem + 2 * rp + (size_t)R * c);
        double* t = tile + (2 * rp) * T::RS + T::pad(c);
        if constexpr (TO_TILE) {
            const double2 v = *g;
            t[0] = v.x;
            t[T::RS] = v.y;
        } else {
            *g = make_double2(t[0], t[T::RS]);
        }
    }
}
// The same move towards the tile in two halves: the global reads into registers (all in flight at once), and
// from the registers into the tile.  A kernel that needs two tiles one after the other through the one LDS
// buffer issues the reads of both before it waits for either.
template <int M>
struct LineTileRegs {
    double2 v[LineTile<M>::STEPS];
};
template <int M>
__device__ __forceinline__ void ln_tile_fetch(LineTileRegs<M>& r, const double* mem, int R, int tid) {
    using T = LineTile<M>;
    const int rp = tid % T::PAIRS;
#pragma unroll
    for (int it = 0; it < T::STEPS; it++) {
        const int c = tid / T::PAIRS + T::COLS_PER_STEP * it;
        r.v[it] = *reinterpret_cast<const double2*>(mem + 2 * rp + (size_t)R * c);
    }
}
template <int M>
__device__ __forceinline__ void ln_tile_put(double* tile, const LineTileRegs<M>& r, int tid) {
    using T = LineTile<M>;
    const int rp = tid % T::PAIRS;
#pragma unroll
    for (int it = 0; it < T::STEPS; it++) {
        const int c = tid / T::PAIRS + T::COLS_PER_STEP * it;
        double* t = tile + (2 * rp) * T::RS + T::pad(c);
        t[0] = r.v[it].x;
        t[T::RS] = r.v[it].y;
    }
}
// one row of the tile <-> samples / lane layout
template <int M>
__device__ __forceinline__ void ln_row_samples(const double* row, double* x, int lane) {
#pragma unroll
    for (int i = 0; i < M; i++) x[i] = row[(M + 1) * lane + i];
}
template <int M>
__device__ __forceinline__ void ln_row_store(double* row, const LineLane<M>& y, int lane) {
    row[LineTile<M>::pad(lane)] = y.v[0];
#pragma unroll
    for (int c = 1; c < M; c *= 2) {
#pragma unroll
        for (int m = 0; m < c; m++) row[LineTile<M>::pad(64 * c + c * lane + m)] = y.v[c + m];
    }
}
template <int M>
__device__ __forceinline__ void ln_row_load(const double* row, LineLane<M>& y, int lane) {
    y.v[0] = row[LineTile<M>::pad(lane)];
#pragma unroll
    for (int c = 1; c < M; c *= 2) {
#pragma unroll
        for (int m = 0; m < c; m++) y.v[c + m] = row[LineTile<M>::pad(64 * c + c * lane + m)];
    }
}

// Forward over the rows: grid (tiles per example <= R / ROWS, examples), 256 threads, LineTile<M>::BYTES of LDS.
// kmax_out (may be null): kmax_out[(example * workgroups per example + workgroup) * 4 + wave] receives the largest
// exponent word among the row coefficients the wave produced.
template <int NP, int M>
__global__ LN_BOUNDS(256) void kr_fwd_rows(const double* __restrict__ in, double* __restrict__ Y, size_t xs, int R,
                                           const FilterDev* filt, unsigned* kmax_out) {
    using T = LineTile<M>;
    extern __shared__ __attribute__((aligned(16))) double ln_tile[];
    double a[NP], b[NP];
    ln_load_taps<NP>(filt, a, b);
    unsigned kmine = 0;
    for (int tile = blockIdx.x; tile < R / T::ROWS; tile += gridDim.x) {
        const int tid = ln_opaque((int)threadIdx.x), lane = tid & 63, wave = tid >> 6;
        const size_t base = (size_t)blockIdx.y * xs + (size_t)tile * T::ROWS;
        __syncthreads();
        ln_move_tile<M, true>(ln_tile, const_cast<double*>(in) + base, R, tid);
        __syncthreads();
        for (int j = wave; j < T::ROWS; j += 4) {
            double* row = ln_tile + j * T::RS;
            LinePyramid<M> p;
            LineLane<M> y;
            double unused_lo;
            ln_row_samples<M>(row, p.x, lane);
            ln_forward<NP, M>(p, a, b, lane, y, unused_lo);
            ln_row_store<M>(row, y, lane);      // in place: the row is in this wave's registers
            if (kmax_out) {
#pragma unroll
                for (int k = 0; k < M; k++) kmine = max(kmine, (unsigned)__double2hiint(y.v[k]) & 0x7FFFFFFFu);
            }
        }
        __syncthreads();
        ln_move_tile<M, false>(ln_tile, Y + base, R, tid);
    }
    if (kmax_out) {
        // one word per wave, a plain store: no barrier and no atomic at the end of 65536 workgroups
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) kmine = max(kmine, (unsigned)__shfl_xor((int)kmine, o, 64));
        if ((threadIdx.x & 63) == 0) kmax_out[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 4 + (threadIdx.x >> 6)] = kmine;
    }
}

// Filter gradient over the rows (no errors on the input: the last pass of the training step): errors on
// the row coefficients (coef) and the rows themselves (in).  grid (tiles per example <= LN_GX, examples).
#ifndef LN_GRAD_ROWS_ONE_WAVE_FROM
#define LN_GRAD_ROWS_ONE_WAVE_FROM LN_ONE_WAVE_FROM
#endif
template <int NP, int M>
__global__ __launch_bounds__(256, (NP >= LN_GRAD_ROWS_ONE_WAVE_FROM ? 1 : 2)) void kr_grad_rows(const double* __restrict__ coef, const double* __restrict__ in, size_t xs, int R,
                                            const FilterDev* filt, double* partial) {
    using T = LineTile<M>;
    constexpr int PER_WAVE = T::ROWS / 4;
    extern __shared__ __attribute__((aligned(16))) double ln_tile[];
    double a[NP], b[NP];
    ln_load_taps<NP>(filt, a, b);
    double gL[NP], gH[NP];
#pragma unroll
    for (int u = 0; u < NP; u++) { gL[u] = 0; gH[u] = 0; }
    for (int tile = blockIdx.x; tile < R / T::ROWS; tile += gridDim.x) {
        const int tid = ln_opaque((int)threadIdx.x), lane = tid & 63, wave = tid >> 6;
        const size_t base = (size_t)blockIdx.y * xs + (size_t)tile * T::ROWS;
        double xr[PER_WAVE][M];
#ifdef LN_GRAD_EARLY
        {
            LineTileRegs<M> tin, tco;
            ln_tile_fetch<M>(tin, in + base, R, tid);
            ln_tile_fetch<M>(tco, coef + base, R, tid);
            __syncthreads();
            ln_tile_put<M>(ln_tile, tin, tid);
            __syncthreads();
#pragma unroll
            for (int jj = 0; jj < PER_WAVE; jj++) ln_row_samples<M>(ln_tile + (wave + 4 * jj) * T::RS, xr[jj], lane);
            __syncthreads();
            ln_tile_put<M>(ln_tile, tco, tid);
            __syncthreads();
        }
#else
        __syncthreads();
        ln_move_tile<M, true>(ln_tile, const_cast<double*>(in) + base, R, tid);
        __syncthreads();
#pragma unroll
        for (int jj = 0; jj < PER_WAVE; jj++) ln_row_samples<M>(ln_tile + (wave + 4 * jj) * T::RS, xr[jj], lane);
        __syncthreads();
        ln_move_tile<M, true>(ln_tile, const_cast<double*>(coef) + base, R, tid);
        __syncthreads();
#endif
#pragma unroll
        for (int jj = 0; jj < PER_WAVE; jj++) {
            LinePyramid<M> p;
            LineLane<M> unused_y, dy;
            double unused_lo, dtop[M];
#pragma unroll
            for (int i = 0; i < M; i++) p.x[i] = xr[jj][i];
            ln_forward<NP, M>(p, a, b, lane, unused_y, unused_lo);
            ln_row_load<M>(ln_tile + (wave + 4 * jj) * T::RS, dy, lane);
            ln_backward<NP, M, false>(p, dy, dtop, a, b, gL, gH, lane);
        }
    }
    __syncthreads();                        // the tile is free: sums of the four waves
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double* red = ln_tile;
#pragma unroll
    for (int u = 0; u < NP; u++) {
        double sl = gL[u], sh = gH[u];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { sl += __shfl_xor(sl, o, 64); sh += __shfl_xor(sh, o, 64); }
        if (lane == 0) { red[wave * 2 * NP + u] = sl; red[wave * 2 * NP + NP + u] = sh; }
    }
    __syncthreads();
    if (threadIdx.x < 2 * NP) {
        double s = 0;
#pragma unroll
        for (int w = 0; w < 4; w++) s += red[w * 2 * NP + threadIdx.x];
        partial[((size_t)blockIdx.y * LN_GX + blockIdx.x) * 2 * NP + threadIdx.x] += s;
    }
}

// ---- training step of 1D signals of 256 or 512 samples (128 works too) ---------------------------------------------------
// One wave per example as k_train1d1024 (wn_wave1d.h), from the line pieces above: cascade in registers,
// the coefficients through LDS for the Gini phase (gini_phase of the workgroup-per-example kernels, run by one
// wave; Q = log2 of the sort slots per lane), backward from the errors in the same lane layout.
template <int NP, int M>
__global__ __launch_bounds__(64, 2) void k_train_line(FusedArgs A) {
    constexpr int n = 64 * M;
    constexpr int Q = (M == 2) ? 1 : 3;              // sort slots per lane: 2 (128 samples) or 8 (256, 512)
    constexpr int SLOTS = 64 << Q;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* bufY = smem;                             // coefficients / sort slots / errors
    double* red = bufY + SLOTS;
    int* flag = reinterpret_cast<int*>(red + 4);
    const int lane = threadIdx.x;
    const FilterDev* filt = A.filt;
    double a[NP], b[NP];
    ln_load_taps<NP>(filt, a, b);
    const int N = filt->N, dpad = filt->d;
    // above 16 taps: the high-pass taps as signed low-pass taps (k_train1d1024, wn_wave1d.h, has the reasoning)
    constexpr bool B_FROM_A = (NP > 16);
    if constexpr (B_FROM_A) {
#pragma unroll
        for (int j = 0; j < NP; j++) b[j] = (j & 1) ? -a[NP - 1 - j] : a[NP - 1 - j];
    }
    const double sg = (B_FROM_A && (dpad & 1)) ? -1.0 : 1.0;
    double accG = 0, accS = 0, accBad = 0;
    for (int e = blockIdx.x; e < A.B; e += gridDim.x) {
        LinePyramid<M> p;
        LineLane<M> y;
        double unused_lo;
        ln_load_samples<M>(A.X + (size_t)e * n, p.x, lane);
        ln_forward<NP, M>(p, a, b, lane, y, unused_lo);
        __syncthreads();                             // the previous example's errors have been read
        ln_store<M>(bufY, y.v, y.v[0], lane);
        __syncthreads();
        const double S = gini_phase<64, Q, true>(bufY, n, red, flag, lane);
        LineLane<M> dy;
        ln_load<M>(bufY, dy, lane);
        double gL[NP], gH[NP], dtop[M];
#pragma unroll
        for (int u = 0; u < NP; u++) { gL[u] = 0; gH[u] = 0; }
        ln_backward<NP, M, false>(p, dy, dtop, a, b, gL, gH, lane);
        // tap k' collects gL[k'] and sH[k'] gH[NP-1-k'], summed over the 64 lanes
        double mine;
        {
            double c[NP];
#pragma unroll
            for (int kp = 0; kp < NP; kp++) c[kp] = filt->vL[kp] * gL[kp] + (sg * filt->sH[kp]) * gH[NP - 1 - kp];
            mine = wave_scatter_sums<NP>(c, lane);      // the total of tap lane mod NP
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
