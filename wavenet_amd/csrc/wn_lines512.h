// Column pass of the streamed path for images with 512 rows: ONE WAVE runs the full-depth cascade of one
// column (512 contiguous samples, 8 per lane) in registers, with the blocked / replicated level shapes
// and the cross-lane halos of wn_wave1d.h.  A column is read once and its coefficients written once
// (forward), or the errors and the column read once and the errors on the column written once
// (backward, which rebuilds the low-pass pyramid in registers instead of reading a stored one) --
// against one read and one write per LEVEL in the level-by-level kernels.
//
// Reference stages: forward_ over the column lines (src/Wavenet.cxx:594-606), backpropagate_ over them
// (src/Wavenet.cxx:680-690).
#pragma once

#include "wn_wave1d.h"

namespace wn {

constexpr int L512_GX = 64;   // workgroups (waves) per example; equals GXMAX of the streamed path's partial sums

template <int NP>
struct Lines512Pyramid {
    double x[8], lo8[4], lo7[2], lo6[1], los[6];
};

// The cascade of one column.  Y / P: where the coefficients and the low-pass values of this column go
// (layout of the level kernels: level i at [2^i, 2^(i+1)), Y[0] = lo_0, P[0] = 0); either may be null.
template <int NP>
__device__ __forceinline__ void l512_forward(Lines512Pyramid<NP>& p, const double (&a)[NP], const double (&b)[NP], int lane,
                                             double* Y, double* P) {
    double h8[4], h7[2], h6[1], his;
    w1d_fwd_block<NP, 8>(p.x, p.lo8, h8, a, b, lane);
    w1d_fwd_block<NP, 4>(p.lo8, p.lo7, h7, a, b, lane);
    w1d_fwd_block<NP, 2>(p.lo7, p.lo6, h6, a, b, lane);
    if (Y) {
        double2* y8 = reinterpret_cast<double2*>(Y + 256 + 4 * lane);
        y8[0] = make_double2(h8[0], h8[1]);
        y8[1] = make_double2(h8[2], h8[3]);
        *reinterpret_cast<double2*>(Y + 128 + 2 * lane) = make_double2(h7[0], h7[1]);
        Y[64 + lane] = h6[0];
    }
    if (P) {
        double2* p8 = reinterpret_cast<double2*>(P + 256 + 4 * lane);
        p8[0] = make_double2(p.lo8[0], p.lo8[1]);
        p8[1] = make_double2(p.lo8[2], p.lo8[3]);
        *reinterpret_cast<double2*>(P + 128 + 2 * lane) = make_double2(p.lo7[0], p.lo7[1]);
        P[64 + lane] = p.lo6[0];
    }
    // levels below 64 values, replicated: lane 2^i + k holds low-pass and high-pass value k of level i
    double small_hi = 0, small_lo = 0;
    w1d_fwd_small<NP, 5>(p.lo6[0], p.los[5], his, a, b, lane);
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
    if (Y) Y[lane] = small_hi;
    if (P) P[lane] = small_lo;
}

template <int NP>
__device__ __forceinline__ void l512_load_taps(const FilterDev* filt, double (&a)[NP], double (&b)[NP]) {
#pragma unroll
    for (int j = 0; j < NP; j++) { a[j] = filt->a[j]; b[j] = filt->b[j]; }
    if constexpr (NP <= 16) {
#pragma unroll
        for (int j = 0; j < NP; j++) { a[j] = w1d_uniform(a[j]); b[j] = w1d_uniform(b[j]); }
    }
}

__device__ __forceinline__ void l512_load_line(const double* src, double (&x)[8], int lane) {
    const double2* s = reinterpret_cast<const double2*>(src + 8 * lane);
#pragma unroll
    for (int i = 0; i < 4; i++) { const double2 t = s[i]; x[2 * i] = t.x; x[2 * i + 1] = t.y; }
}

// grid (workgroups per example, examples), 64 threads.  in/Y/P: chunk arrays, example stride xs, column c
// of an example at c * 512.
template <int NP>
__global__ __launch_bounds__(64, (NP > 16 ? 1 : 2)) void kc_fwd_cols512(const double* __restrict__ in, double* __restrict__ Y,
                                                                        double* __restrict__ P, size_t xs, int ncols,
                                                                        const FilterDev* filt) {
    const int lane = threadIdx.x;
    double a[NP], b[NP];
    l512_load_taps<NP>(filt, a, b);
    const size_t off = (size_t)blockIdx.y * xs;
    for (int c = blockIdx.x; c < ncols; c += gridDim.x) {
        const size_t lo = off + (size_t)c * 512;
        Lines512Pyramid<NP> p;
        l512_load_line(in + lo, p.x, lane);
        l512_forward<NP>(p, a, b, lane, Y + lo, P ? P + lo : nullptr);
    }
}

// The error on one coefficient from its value and rank (kb_delta_home of the streamed path, same expression).
struct L512Gini {
    double f, g, inv_g2, n;
    __device__ __forceinline__ double delta(double y, unsigned rank) const {
        const double sg = sign_ref(y);
        const double coef = (y != 0.0) ? 2.0 * (double)rank - n - 1.0 : 0.0;
        const double df = coef * sg;
        return -(df * g - f * (n * sg)) * inv_g2;
    }
};

// Errors on the coefficients of the columns and the columns themselves (in) -> errors on the columns
// (out, may be null: last pass) and the window sums of the filter gradient, added to
// partial[(example * L512_GX + workgroup) * 2 NP + u] as the level kernels do.  The errors on the
// coefficients are read from coef, or with FROM_RANKS computed on the way in from the coefficients
// (coef), their ranks and the example's sums fg = {f, g}: the array of errors is then never written.
// ranks: per slice of `slice` examples an array of slice * n ranks at ranks + slice_first_example * n
// DOUBLES (the streamed path's layout).
template <int NP, bool FROM_RANKS>
__global__ __launch_bounds__(64, (NP > 16 ? 1 : 2)) void kc_bwd_cols512(const double* __restrict__ coef, const double* __restrict__ in,
                                                                        double* __restrict__ out, size_t xs, int ncols,
                                                                        const FilterDev* filt, double* partial,
                                                                        const double* ranks, int slice, const double* fg) {
    const int lane = threadIdx.x;
    double a[NP], b[NP];
    l512_load_taps<NP>(filt, a, b);
    double gL[NP], gH[NP];
#pragma unroll
    for (int u = 0; u < NP; u++) { gL[u] = 0; gH[u] = 0; }
    const size_t off = (size_t)blockIdx.y * xs;
    L512Gini gini{0, 0, 0, (double)xs};
    const unsigned* rk = nullptr;
    if constexpr (FROM_RANKS) {
        const int e = blockIdx.y, e0 = (e / slice) * slice;
        rk = reinterpret_cast<const unsigned*>(ranks + (size_t)e0 * xs) + (size_t)(e - e0) * xs;
        gini.f = fg[2 * e];
        gini.g = fg[2 * e + 1];
        gini.inv_g2 = 1.0 / (gini.g * gini.g);
    }
    for (int c = blockIdx.x; c < ncols; c += gridDim.x) {
        const size_t lo = off + (size_t)c * 512;
        Lines512Pyramid<NP> p;
        l512_load_line(in + lo, p.x, lane);
        l512_forward<NP>(p, a, b, lane, nullptr, nullptr);
        const double* dc = coef + lo;
        const unsigned* rc = rk + (size_t)c * 512;
        // errors on the coefficients, in the lane layout of the cascade
        double dsmall, h1[1], h2[2], h4[4];
        {
            dsmall = dc[lane];
            h1[0] = dc[64 + lane];
            const double2 t2 = *reinterpret_cast<const double2*>(dc + 128 + 2 * lane);
            h2[0] = t2.x; h2[1] = t2.y;
            const double2* t = reinterpret_cast<const double2*>(dc + 256 + 4 * lane);
            const double2 t0 = t[0], t1 = t[1];
            h4[0] = t0.x; h4[1] = t0.y; h4[2] = t1.x; h4[3] = t1.y;
        }
        if constexpr (FROM_RANKS) {
            dsmall = gini.delta(dsmall, rc[lane]);
            h1[0] = gini.delta(h1[0], rc[64 + lane]);
            const uint2 r2 = *reinterpret_cast<const uint2*>(rc + 128 + 2 * lane);
            h2[0] = gini.delta(h2[0], r2.x); h2[1] = gini.delta(h2[1], r2.y);
            const uint4 r4 = *reinterpret_cast<const uint4*>(rc + 256 + 4 * lane);
            h4[0] = gini.delta(h4[0], r4.x); h4[1] = gini.delta(h4[1], r4.y);
            h4[2] = gini.delta(h4[2], r4.z); h4[3] = gini.delta(h4[3], r4.w);
        }
        double d = w1d_lane(dsmall, 0);                       // lo_0
        d = w1d_bwd_small<NP, 0>(d, w1d_lane(dsmall, 1), p.los[1], a, b, gL, gH, lane);
        d = w1d_bwd_small<NP, 1>(d, w1d_lane(dsmall, 2 + (lane & 1)), p.los[2], a, b, gL, gH, lane);
        d = w1d_bwd_small<NP, 2>(d, w1d_lane(dsmall, 4 + (lane & 3)), p.los[3], a, b, gL, gH, lane);
        d = w1d_bwd_small<NP, 3>(d, w1d_lane(dsmall, 8 + (lane & 7)), p.los[4], a, b, gL, gH, lane);
        d = w1d_bwd_small<NP, 4>(d, w1d_lane(dsmall, 16 + (lane & 15)), p.los[5], a, b, gL, gH, lane);
        d = w1d_bwd_small<NP, 5>(d, w1d_lane(dsmall, 32 + (lane & 31)), p.lo6[0], a, b, gL, gH, lane);
        double d1[1] = {d}, d2[2], d4[4], d8[8];
        w1d_bwd_block<NP, 1, true>(d1, h1, p.lo7, d2, a, b, gL, gH, lane);
        w1d_bwd_block<NP, 2, true>(d2, h2, p.lo8, d4, a, b, gL, gH, lane);
        if (out) {
            w1d_bwd_block<NP, 4, true>(d4, h4, p.x, d8, a, b, gL, gH, lane);
            double2* o = reinterpret_cast<double2*>(out + lo + 8 * lane);
#pragma unroll
            for (int i = 0; i < 4; i++) o[i] = make_double2(d8[2 * i], d8[2 * i + 1]);
        } else {
            w1d_bwd_block<NP, 4, false>(d4, h4, p.x, d8, a, b, gL, gH, lane);
        }
    }
    double* part = partial + ((size_t)blockIdx.y * L512_GX + blockIdx.x) * 2 * NP;
#pragma unroll
    for (int u = 0; u < NP; u++) {
        double sl = gL[u], sh = gH[u];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { sl += __shfl_xor(sl, o, 64); sh += __shfl_xor(sh, o, 64); }
        if (lane == 0) { part[u] += sl; part[NP + u] += sh; }
    }
}

}  // namespace wn
