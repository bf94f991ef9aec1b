// Passes of the streamed path over lines of 512 samples: ONE WAVE runs the full-depth cascade of one line
// (8 samples per lane) in registers, with the blocked / replicated level shapes and the cross-lane
// halos of wn_wave1d.h.  A line is read once and its coefficients written once (forward), or the errors
// and the line read once and the errors on the line written once (backward, which rebuilds the low-pass
// pyramid in registers instead of reading a stored one) -- against one read and one write per LEVEL in
// the level-by-level kernels.
//
// Columns of an image with 512 rows are contiguous (kc_*).  Rows of an image with 512 columns are
// strided by the number of rows: a workgroup moves 16 adjacent rows through an LDS tile, 128 contiguous
// bytes per column on the global side, one padded row per wave on the LDS side (kr_*).
//
// Reference stages: forward_ over the row and column lines (src/Wavenet.cxx:582-606), backpropagate_
// over them (src/Wavenet.cxx:680-702).
#pragma once

#include "wn_wave1d.h"

namespace wn {

constexpr int L512_GX = 64;   // workgroups (waves) per example; equals GXMAX of the streamed path's partial sums

template <int NP>
struct Lines512Pyramid {
    double x[8], lo8[4], lo7[2], lo6[1], los[6];
};

// One line's values in the lane layout of the cascade: index 256 + 4 lane + m (top level), 128 + 2 lane + m,
// 64 + lane, and lane (the 64 values of the levels below; index 0 is lo_0 for coefficients, 0 for the pyramid).
struct L512Lane {
    double top[4], mid[2], low, small;
};

// The cascade of one line: p.x in, pyramid left in p, coefficients in y, low-pass values in lp
// (layout of the level kernels: level i at [2^i, 2^(i+1)), Y[0] = lo_0, P[0] = 0).
template <int NP>
__device__ __forceinline__ void l512_forward(Lines512Pyramid<NP>& p, const double (&a)[NP], const double (&b)[NP], int lane,
                                             L512Lane& y, L512Lane& lp) {
    double h6[1], l6[1], his;
    w1d_fwd_block<NP, 8>(p.x, p.lo8, y.top, a, b, lane);
    w1d_fwd_block<NP, 4>(p.lo8, p.lo7, y.mid, a, b, lane);
    w1d_fwd_block<NP, 2>(p.lo7, l6, h6, a, b, lane);
    p.lo6[0] = l6[0];
    y.low = h6[0];
#pragma unroll
    for (int m = 0; m < 4; m++) lp.top[m] = p.lo8[m];
    lp.mid[0] = p.lo7[0];
    lp.mid[1] = p.lo7[1];
    lp.low = p.lo6[0];
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
    y.small = small_hi;
    lp.small = small_lo;
}

// Backward over one line: errors on its coefficients (lane layout) and its pyramid -> window sums of the
// filter gradient; with PROP the errors on the line itself, 8 per lane.
template <int NP, bool PROP>
__device__ __forceinline__ void l512_backward(const Lines512Pyramid<NP>& p, const L512Lane& dy, double (&d8)[8], const double (&a)[NP],
                                              const double (&b)[NP], double (&gL)[NP], double (&gH)[NP], int lane) {
    const double dsmall = dy.small;
    double d = w1d_lane(dsmall, 0);                       // lo_0
    d = w1d_bwd_small<NP, 0>(d, w1d_lane(dsmall, 1), p.los[1], a, b, gL, gH, lane);
    d = w1d_bwd_small<NP, 1>(d, w1d_lane(dsmall, 2 + (lane & 1)), p.los[2], a, b, gL, gH, lane);
    d = w1d_bwd_small<NP, 2>(d, w1d_lane(dsmall, 4 + (lane & 3)), p.los[3], a, b, gL, gH, lane);
    d = w1d_bwd_small<NP, 3>(d, w1d_lane(dsmall, 8 + (lane & 7)), p.los[4], a, b, gL, gH, lane);
    d = w1d_bwd_small<NP, 4>(d, w1d_lane(dsmall, 16 + (lane & 15)), p.los[5], a, b, gL, gH, lane);
    d = w1d_bwd_small<NP, 5>(d, w1d_lane(dsmall, 32 + (lane & 31)), p.lo6[0], a, b, gL, gH, lane);
    double d1[1] = {d}, h1[1] = {dy.low}, d2[2], d4[4];
    w1d_bwd_block<NP, 1, true>(d1, h1, p.lo7, d2, a, b, gL, gH, lane);
    w1d_bwd_block<NP, 2, true>(d2, dy.mid, p.lo8, d4, a, b, gL, gH, lane);
    w1d_bwd_block<NP, 4, PROP>(d4, dy.top, p.x, d8, a, b, gL, gH, lane);
}

// contiguous line in memory <-> lane layout
__device__ __forceinline__ void l512_store(double* Y, const L512Lane& v, int lane) {
    double2* t = reinterpret_cast<double2*>(Y + 256 + 4 * lane);
    t[0] = make_double2(v.top[0], v.top[1]);
    t[1] = make_double2(v.top[2], v.top[3]);
    *reinterpret_cast<double2*>(Y + 128 + 2 * lane) = make_double2(v.mid[0], v.mid[1]);
    Y[64 + lane] = v.low;
    Y[lane] = v.small;
}
__device__ __forceinline__ void l512_load(const double* Y, L512Lane& v, int lane) {
    const double2* t = reinterpret_cast<const double2*>(Y + 256 + 4 * lane);
    const double2 t0 = t[0], t1 = t[1];
    v.top[0] = t0.x; v.top[1] = t0.y; v.top[2] = t1.x; v.top[3] = t1.y;
    const double2 m = *reinterpret_cast<const double2*>(Y + 128 + 2 * lane);
    v.mid[0] = m.x; v.mid[1] = m.y;
    v.low = Y[64 + lane];
    v.small = Y[lane];
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
        L512Lane y, lp;
        l512_load_line(in + lo, p.x, lane);
        l512_forward<NP>(p, a, b, lane, y, lp);
        l512_store(Y + lo, y, lane);
        if (P) l512_store(P + lo, lp, lane);
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
        L512Lane unused_y, unused_lp, dy;
        l512_load_line(in + lo, p.x, lane);
        l512_forward<NP>(p, a, b, lane, unused_y, unused_lp);
        l512_load(coef + lo, dy, lane);          // errors on the coefficients, or the coefficients themselves
        if constexpr (FROM_RANKS) {
            const unsigned* rc = rk + (size_t)c * 512;
            dy.small = gini.delta(dy.small, rc[lane]);
            dy.low = gini.delta(dy.low, rc[64 + lane]);
            const uint2 r2 = *reinterpret_cast<const uint2*>(rc + 128 + 2 * lane);
            dy.mid[0] = gini.delta(dy.mid[0], r2.x); dy.mid[1] = gini.delta(dy.mid[1], r2.y);
            const uint4 r4 = *reinterpret_cast<const uint4*>(rc + 256 + 4 * lane);
            dy.top[0] = gini.delta(dy.top[0], r4.x); dy.top[1] = gini.delta(dy.top[1], r4.y);
            dy.top[2] = gini.delta(dy.top[2], r4.z); dy.top[3] = gini.delta(dy.top[3], r4.w);
        }
        double d8[8];
        if (out) {
            l512_backward<NP, true>(p, dy, d8, a, b, gL, gH, lane);
            double2* o = reinterpret_cast<double2*>(out + lo + 8 * lane);
#pragma unroll
            for (int i = 0; i < 4; i++) o[i] = make_double2(d8[2 * i], d8[2 * i + 1]);
        } else {
            l512_backward<NP, false>(p, dy, d8, a, b, gL, gH, lane);
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


// ---- rows of images with 512 columns ---------------------------------------------------------------------
// Tile: 16 rows x 512 columns in LDS, row stride L512_RS doubles, element c of a row at c + c / 8: a lane's
// 8 samples sit 72 bytes from its neighbour's (conflict-free 8-byte reads), and the 8 row pairs x 4
// columns that half a wave moves on the global side fall into 32 different banks (row stride = 2 mod 32).
constexpr int L512_RS = 578, L512_ROWS = 16;
constexpr size_t L512_TILE_BYTES = (size_t)L512_ROWS * L512_RS * sizeof(double);
__device__ __forceinline__ int l512_pad(int c) { return c + (c >> 3); }

// 16 adjacent rows (r0 .. r0 + 15) of one example between memory (element (r, c) at r + R c) and the tile
template <bool TO_TILE>
__device__ __forceinline__ void l512_move_tile(double* tile, double* mem, int R) {
    const int rp = threadIdx.x & 7;
#pragma unroll 4
    for (int it = 0; it < 16; it++) {
        const int c = (threadIdx.x >> 3) + 32 * it;
        double2* g = reinterpret_cast<double2*>(mem + 2 * rp + (size_t)R * c);
        double* t = tile + (2 * rp) * L512_RS + l512_pad(c);
        if constexpr (TO_TILE) {
            const double2 v = *g;
            t[0] = v.x;
            t[L512_RS] = v.y;
        } else {
            *g = make_double2(t[0], t[L512_RS]);
        }
    }
}
// one row of the tile <-> lane layout
__device__ __forceinline__ void l512_row_samples(const double* row, double (&x)[8], int lane) {
#pragma unroll
    for (int i = 0; i < 8; i++) x[i] = row[9 * lane + i];
}
__device__ __forceinline__ void l512_row_store(double* row, const L512Lane& v, int lane) {
#pragma unroll
    for (int m = 0; m < 4; m++) row[l512_pad(256 + 4 * lane + m)] = v.top[m];
    row[l512_pad(128 + 2 * lane)] = v.mid[0];
    row[l512_pad(128 + 2 * lane + 1)] = v.mid[1];
    row[l512_pad(64 + lane)] = v.low;
    row[l512_pad(lane)] = v.small;
}
__device__ __forceinline__ void l512_row_load(const double* row, L512Lane& v, int lane) {
#pragma unroll
    for (int m = 0; m < 4; m++) v.top[m] = row[l512_pad(256 + 4 * lane + m)];
    v.mid[0] = row[l512_pad(128 + 2 * lane)];
    v.mid[1] = row[l512_pad(128 + 2 * lane + 1)];
    v.low = row[l512_pad(64 + lane)];
    v.small = row[l512_pad(lane)];
}

// Forward over the rows: grid (R / 16, examples), 256 threads, L512_TILE_BYTES of dynamic LDS.
template <int NP>
__global__ __launch_bounds__(256, 2) void kr_fwd_rows512(const double* __restrict__ in, double* __restrict__ Y, size_t xs, int R,
                                                         const FilterDev* filt) {
    extern __shared__ __attribute__((aligned(16))) double l512_tile[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double a[NP], b[NP];
    l512_load_taps<NP>(filt, a, b);
    const size_t base = (size_t)blockIdx.y * xs + (size_t)blockIdx.x * L512_ROWS;
    l512_move_tile<true>(l512_tile, const_cast<double*>(in) + base, R);
    __syncthreads();
    for (int j = wave; j < L512_ROWS; j += 4) {
        double* row = l512_tile + j * L512_RS;
        Lines512Pyramid<NP> p;
        L512Lane y, lp;
        l512_row_samples(row, p.x, lane);
        l512_forward<NP>(p, a, b, lane, y, lp);
        l512_row_store(row, y, lane);      // in place: the row is in this wave's registers
    }
    __syncthreads();
    l512_move_tile<false>(l512_tile, Y + base, R);
}

// Filter gradient over the rows (no errors on the input: the last pass of the training step): errors on
// the row coefficients (coef) and the rows themselves (in).  grid (R / 16 <= L512_GX, examples), 256 threads.
template <int NP>
__global__ __launch_bounds__(256, 2) void kr_grad_rows512(const double* __restrict__ coef, const double* __restrict__ in, size_t xs,
                                                          int R, const FilterDev* filt, double* partial) {
    extern __shared__ __attribute__((aligned(16))) double l512_tile[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double a[NP], b[NP];
    l512_load_taps<NP>(filt, a, b);
    double gL[NP], gH[NP];
#pragma unroll
    for (int u = 0; u < NP; u++) { gL[u] = 0; gH[u] = 0; }
    const size_t base = (size_t)blockIdx.y * xs + (size_t)blockIdx.x * L512_ROWS;
    double xr[4][8];
    l512_move_tile<true>(l512_tile, const_cast<double*>(in) + base, R);
    __syncthreads();
#pragma unroll
    for (int jj = 0; jj < 4; jj++) l512_row_samples(l512_tile + (wave + 4 * jj) * L512_RS, xr[jj], lane);
    __syncthreads();
    l512_move_tile<true>(l512_tile, const_cast<double*>(coef) + base, R);
    __syncthreads();
#pragma unroll
    for (int jj = 0; jj < 4; jj++) {
        Lines512Pyramid<NP> p;
        L512Lane unused_y, unused_lp, dy;
#pragma unroll
        for (int i = 0; i < 8; i++) p.x[i] = xr[jj][i];
        l512_forward<NP>(p, a, b, lane, unused_y, unused_lp);
        l512_row_load(l512_tile + (wave + 4 * jj) * L512_RS, dy, lane);
        double d8[8];
        l512_backward<NP, false>(p, dy, d8, a, b, gL, gH, lane);
    }
    __syncthreads();                        // the tile is free: sums of the four waves
    double* red = l512_tile;
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
        partial[((size_t)blockIdx.y * L512_GX + blockIdx.x) * 2 * NP + threadIdx.x] += s;
    }
}

}  // namespace wn
