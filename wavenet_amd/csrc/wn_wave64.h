// Register-resident training kernel for 64x64 examples: ONE WAVE owns one example.
//
// Lane l holds row l (row stage) or column l (column stage) of the image in 64 registers, so the
// whole filter cascade of a line -- forward, transposed (backward) and the filter-gradient
// correlation -- is straight-line code on registers with compile-time indices: no LDS traffic, no
// address arithmetic, no barrier.  LDS is used only for 64x64 transposes (row <-> column
// orientation, and the lane <-> register exchange of the sort) in a per-wave 64x65 tile, and for
// handing the ranks back.  Four such waves share a CU, each at its own pace.
//
// The Gini ranks come from a bitonic sort of 4096 keys, 64 per lane.  A key is |c| with its low 12
// mantissa bits replaced by the element index: keys are unique, v_min_f64 / v_max_f64 carry the
// index for free, and equal magnitudes come out in ascending index order.  Magnitudes that differ
// only in those 12 bits come out as one group in index order; sparse inputs produce such groups all
// the time (the same product rounded along two paths), so they are re-ranked in place by (low 12
// bits, index) from small tables in LDS, which makes the order that of (exact magnitude, index).
// Descending merge directions are realised by negating keys, so every compare-exchange is
// min -> lower index.
#pragma once

#include "wn_kernels.h"

namespace wn {


// min -> lo, max -> hi.
__device__ __forceinline__ void cmpx_asm(double& lo, double& hi) {
    double mn, mx;
    asm volatile("v_min_f64 %0, %2, %3\n\tv_max_f64 %1, %2, %3" : "=&v"(mn), "=&v"(mx) : "v"(lo), "v"(hi));
    lo = mn;
    hi = mx;
}

// Order point between LDS writes and the reads that follow: a wave's LDS operations execute in
// order, so only the compiler has to be told.
__device__ __forceinline__ void w64_lds_order() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// ---- one line in registers ------------------------------------------------------------------------

// Input of cascade level LG (2^(LG+1) samples): the line itself at the top, else pyramid level LG+1.
template <int LG>
__device__ __forceinline__ double w64_act(const double (&x)[64], const double (&p)[64], int idx) {
    if constexpr (LG == 5) return x[idx];
    else return p[(2 << LG) + idx];
}

// Forward cascade of a 64-sample line.  p[2^i + k] = low-pass level i (p[0] unused).  The
// coefficients (out[0] = lo_0, out[2^i + k] = hi_i[k]) go to y, or with TO_TILE straight to the
// lane's row of the LDS tile, the first half of a transpose.
template <int NP, int LG, bool TO_TILE>
__device__ __forceinline__ void w64_fwd_levels(const double (&x)[64], double (&p)[64], double (&y)[64], double* trow,
                                               const double (&a)[NP], const double (&b)[NP]) {
    constexpr int half = 1 << LG, len = 2 * half;
#pragma unroll
    for (int k = 0; k < half; k++) {
        double sl = 0, sh = 0;
#pragma unroll
        for (int j = 0; j < NP; j++) {
            const int idx = (2 * k + NP / 2 - j) & (len - 1);
            const double v = w64_act<LG>(x, p, idx);
            sl = __builtin_fma(a[j], v, sl);
            sh = __builtin_fma(b[j], v, sh);
        }
        p[half + k] = sl;
        if constexpr (TO_TILE) trow[half + k] = sh;
        else y[half + k] = sh;
    }
    if constexpr (LG > 0) {
        w64_fwd_levels<NP, LG - 1, TO_TILE>(x, p, y, trow, a, b);
    } else {
        if constexpr (TO_TILE) trow[0] = p[1];
        else y[0] = p[1];
    }
}

// Low-pass pyramid only, levels 5..1 (what the backward pass reads), written to the lane's row of the
// LDS tile: trow[2^i + k] = lo_i[k].  Registers hold a level only until the next one is computed.
template <int NP, int LG>
__device__ __forceinline__ void w64_pyramid_to_tile(const double (&x)[64], double (&p)[64], double* trow, const double (&a)[NP]) {
    constexpr int half = 1 << LG, len = 2 * half;
#pragma unroll
    for (int k = 0; k < half; k++) {
        double sl = 0;
#pragma unroll
        for (int j = 0; j < NP; j++) {
            const int idx = (2 * k + NP / 2 - j) & (len - 1);
            sl = __builtin_fma(a[j], w64_act<LG>(x, p, idx), sl);
        }
        p[half + k] = sl;
        trow[half + k] = sl;
    }
    if constexpr (LG > 1) w64_pyramid_to_tile<NP, LG - 1>(x, p, trow, a);
}

// Backward cascade of a line, levels LG = 0..5 (src/Wavenet.cxx:509-545): d in place.  With GRAD the
// filter-gradient correlation is accumulated, its activations below the top level coming from the
// lane's row of the LDS tile (w64_pyramid_to_tile); without it this is the inverse transform of the
// line (src/Wavenet.cxx:447-472) and x, the tile row and gL / gH are not touched.  At the top level
// the errors on the line's input (the reconstructed line) are dropped (TOP = W64_TOP_NONE: row stage
// of the training step), go straight to the lane's tile row (W64_TOP_TILE: first half of a transpose;
// after the last read of the pyramid, which the in-order LDS keeps safe) or replace d (W64_TOP_REGS).
constexpr int W64_TOP_NONE = 0, W64_TOP_TILE = 1, W64_TOP_REGS = 2;
template <int NP, int LG, int TOP, bool GRAD>
__device__ __forceinline__ void w64_bwd_levels(double (&d)[64], const double (&x)[64], double* trow,
                                               const double (&a)[NP], const double (&b)[NP],
                                               double (&gL)[NP], double (&gH)[NP]) {
    constexpr int len = 1 << LG, alen = 2 * len;
    if constexpr (GRAD) {
        double act[LG < 5 ? alen : 1];
        if constexpr (LG < 5) {
#pragma unroll
            for (int i = 0; i < alen; i++) act[i] = trow[alen + i];
        }
#pragma unroll
        for (int r = 0; r < len; r++) {
            const double dl = d[r], dh = d[len + r];
#pragma unroll
            for (int u = 0; u < NP; u++) {
                const int idx = (2 * r + NP / 2 - u) & (alen - 1);
                double v;
                if constexpr (LG == 5) v = x[idx];
                else v = act[idx];
                gL[u] = __builtin_fma(dl, v, gL[u]);
                gH[u] = __builtin_fma(dh, v, gH[u]);
            }
        }
    }
    if constexpr (LG < 5 || TOP != W64_TOP_NONE) {
        constexpr int S = NP / 2;
        constexpr int JE = S & 1, JO = (S + 1) & 1;
        constexpr int CE = (JE - S) / 2, CO = (1 + JO - S) / 2;
        constexpr bool to_tile = (LG == 5 && TOP == W64_TOP_TILE);
        double nw[to_tile ? 2 : alen];
#pragma unroll
        for (int q = 0; q < len; q++) {
            double se = 0, so = 0;
#pragma unroll
            for (int i = 0; i < NP / 2; i++) {
                const int re = (q + i + CE) & (len - 1), ro = (q + i + CO) & (len - 1);
                se = __builtin_fma(a[JE + 2 * i], d[re], se);
                se = __builtin_fma(b[JE + 2 * i], d[len + re], se);
                so = __builtin_fma(a[JO + 2 * i], d[ro], so);
                so = __builtin_fma(b[JO + 2 * i], d[len + ro], so);
            }
            if constexpr (to_tile) {
                trow[2 * q] = se;
                trow[2 * q + 1] = so;
            } else {
                nw[2 * q] = se;
                nw[2 * q + 1] = so;
            }
        }
        if constexpr (!to_tile) {
#pragma unroll
            for (int c = 0; c < alen; c++) d[c] = nw[c];
        }
    }
    if constexpr (LG < 5) w64_bwd_levels<NP, LG + 1, TOP, GRAD>(d, x, trow, a, b, gL, gH);
}

// ---- 64x64 transpose through the wave's LDS tile -----------------------------------------------------
// Row stride 65 doubles: conflict-free both ways.  No fence between the writes and the reads: both
// are accesses to doubles that may alias as far as the compiler can tell, so it keeps their program
// order, and the LDS executes a wave's operations in order -- while the scheduler stays free to
// overlap the transfers with the arithmetic on either side.
__device__ __forceinline__ void w64_read_transposed(double (&v)[64], const double* tile, int lane) {
#pragma unroll
    for (int r = 0; r < 64; r++) v[r] = tile[r * 65 + lane];
}
// v[r] of lane l  ->  v[l] of lane r.  The reads are issued in the order in which a compare-exchange
// stage on register bit FIRST consumes them (r, r + 2^FIRST, ...), so that stage can start while
// the rest is still in flight.
template <int FIRST = 0>
__device__ __forceinline__ void w64_transpose(double (&v)[64], double* tile, int lane) {
#pragma unroll
    for (int r = 0; r < 64; r++) tile[lane * 65 + r] = v[r];
#pragma unroll
    for (int i = 0; i < 64; i++) {
        // i enumerates (pair, member): member bit goes to bit FIRST of r
        const int pair = i >> 1, member = i & 1;
        const int r = ((pair >> FIRST) << (FIRST + 1)) | (member << FIRST) | (pair & ((1 << FIRST) - 1));
        v[r] = tile[r * 65 + lane];
    }
}

// ---- sort ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void flip_sign(double& k, unsigned mask) {
    k = __hiloint2double(__double2hiint(k) ^ (int)mask, __double2loint(k));
}

// compare-exchanges on register bit RB of all 64 keys
template <int RB>
__device__ __forceinline__ void w64_substage(double (&k)[64]) {
#pragma unroll
    for (int r = 0; r < 64; r++) {
        if (!(r & (1 << RB))) cmpx_asm(k[r], k[r | (1 << RB)]);
    }
}
template <int RB>
__device__ __forceinline__ void w64_substages_down(double (&k)[64]) {
    w64_substage<RB>(k);
    if constexpr (RB > 0) w64_substages_down<RB - 1>(k);
}

// Merge step KK (blocks of 2^KK) of the bitonic sort.  Index e = lane * 64 + r in layout A (register
// bits are e's bits 0-5) and e = r * 64 + lane in layout B (register bits are e's bits 6-11).  A key
// is held negated while its block sorts descending, i.e. while bit KK of e is set.
template <int KK>
__device__ __forceinline__ void w64_merge(double (&k)[64], double* tile, int lane) {
    constexpr unsigned SIGN = 0x80000000u;
    if constexpr (KK <= 5) {
        // bits KK and KK-1 are register bits: flip where they differ (KK = 1: where bit 1 is set)
#pragma unroll
        for (int r = 0; r < 64; r++) {
            const int hi = (r >> KK) & 1, lo = (KK > 1) ? ((r >> (KK - 1)) & 1) : 0;
            if (hi ^ lo) flip_sign(k[r], SIGN);
        }
        w64_substages_down<KK - 1>(k);
    } else if constexpr (KK == 6) {
        // bit 6 is lane bit 0, bit 5 a register bit
        const unsigned m0 = (lane & 1) ? SIGN : 0u, m1 = m0 ^ SIGN;
#pragma unroll
        for (int r = 0; r < 64; r++) flip_sign(k[r], (r & 32) ? m1 : m0);
        w64_substages_down<5>(k);
    } else {
        // bits KK-1 .. 6 in layout B, then bits 5 .. 0 in layout A
        w64_transpose<KK - 7>(k, tile, lane);
#pragma unroll
        for (int r = 0; r < 64; r++) {
            const int hi = (KK < 12) ? ((r >> (KK - 6)) & 1) : 0, lo = (r >> (KK - 7)) & 1;
            if (hi ^ lo) flip_sign(k[r], SIGN);
        }
        w64_substages_down<KK - 7>(k);
        w64_transpose<5>(k, tile, lane);
        w64_substages_down<5>(k);
    }
    if constexpr (KK < 12) w64_merge<KK + 1>(k, tile, lane);
}

// Exact ranks by counting, for the rare example whose near-equal magnitudes the 12-bit payload would
// misorder: rank_j = 1 + #{i : (|c_i|, i) < (|c_j|, j)}.  tile holds |c| at [column * 65 + row].
__device__ __forceinline__ void w64_exact_ranks(const double (&y)[64], unsigned (&rank)[64], const double* tile, int lane) {
#pragma unroll
    for (int r = 0; r < 64; r++) rank[r] = 1;
    for (int c = 0; c < 64; c++) {
        for (int q = 0; q < 64; q++) {
            const double v = tile[c * 65 + q];
            const int i = q + 64 * c;
#pragma unroll
            for (int r = 0; r < 64; r++) {
                const double mine = __builtin_fabs(y[r]);
                const int j = r + 64 * lane;
                rank[r] += (v < mine || (v == mine && i < j)) ? 1u : 0u;
            }
        }
    }
}

// Byte offsets inside the wave's tile for the rank hand-over, three tables of 4096 16-bit entries:
// the low 12 bits of every exact magnitude by element index, (group flag | index) by sorted
// position, ranks by element index.
constexpr int W64_OFF_LOW = 0, W64_OFF_POS = 8192, W64_OFF_RANK = 16384;

// What the sorting half of the Gini phase hands to the finishing half.
struct W64Rank {
    bool odd_rank;              // some magnitude is neither zero nor a normal number: rank by counting
    bool odd_sign;              // some magnitude is outside the range where the fast sign is exact
    bool any_group;             // some keys agree in their upper 52 bits: re-rank those groups
    unsigned long long grouped; // bit r: sorted position lane * 64 + r belongs to such a group
};

// First half of the Gini phase: sorts the keys of the coefficients y (lane = column c, y[r] =
// Y(r, c), vectorise index r + 64 c) and leaves a provisional rank for every element in the tile.
// Needs nothing of y afterwards, so the caller can let it go and recompute it: 64 fewer live
// doubles during the sort are worth far more than the 504 multiply-adds.
__device__ __forceinline__ W64Rank w64_rank_sort(const double (&y)[64], double* tile, int lane_in) {
    // opaque copy of the lane id: keeps the 64 per-element indices from being hoisted out of the
    // example loop (where they would only be spilled)
    int lane = lane_in;
    asm volatile("" : "+v"(lane));
    W64Rank R;
    // The index-in-the-mantissa keys need magnitudes that are zero or normal numbers; the fast sign
    // needs them inside the range where x / sqrt(x*x) is exactly +-1.  Anything else is rare.
    bool odd_rank = false, odd_sign = false;
#pragma unroll
    for (int r = 0; r < 64; r++) {
        const double ay = __builtin_fabs(y[r]);
        odd_rank |= !((ay == 0.0) | ((ay >= 0x1p-1000) & (ay <= 0x1p1000)));
        odd_sign |= !((ay == 0.0) | ((ay >= 0x1p-500) & (ay <= 0x1p500)));
    }
    R.odd_rank = __any(odd_rank);
    R.odd_sign = __any(odd_sign);
    R.any_group = false;
    R.grouped = 0;
    if (R.odd_rank) return R;

    double k[64];
#pragma unroll
    for (int r = 0; r < 64; r++) {
        const double ay = __builtin_fabs(y[r]);
        const unsigned idx = (unsigned)(r + 64 * lane);
        k[r] = __hiloint2double(__double2hiint(ay), (int)((((unsigned)__double2loint(ay)) & ~0xFFFu) | idx));
    }
    w64_merge<1>(k, tile, lane);
    // Sorted position p = lane * 64 + r.  Everyone gets rank p + 1 first.
    w64_lds_order();
    unsigned short* rk = reinterpret_cast<unsigned short*>(reinterpret_cast<char*>(tile) + W64_OFF_RANK);
#pragma unroll
    for (int r = 0; r < 64; r++) rk[(unsigned)__double2loint(k[r]) & 0xFFFu] = (unsigned short)(lane * 64 + r + 1);
    // Keys that agree in the upper 52 bits (zeros aside) form a group whose members are in index
    // order and must be put in (exact magnitude, index) order.  gp bit r: position p belongs to the
    // same group as position p - 1.
    unsigned long long gp = 0;
    {
        const double prev_last = __shfl_up(k[63], 1, 64);
#pragma unroll
        for (int r = 0; r < 64; r++) {
            const unsigned long long kb = (unsigned long long)__double_as_longlong(k[r]);
            const unsigned long long pb = (unsigned long long)__double_as_longlong((r > 0) ? k[r - 1] : prev_last);
            const bool same = ((pb ^ kb) < 4096ull) & (kb >= 4096ull) & ((r > 0) | (lane > 0));
            gp |= same ? (1ull << r) : 0ull;
        }
    }
    R.any_group = __any(gp != 0);
    if (R.any_group) {
        const unsigned next_gp0 = __shfl_down((unsigned)(gp & 1ull), 1, 64);
        R.grouped = gp | (gp >> 1) | ((lane < 63 && next_gp0) ? (1ull << 63) : 0ull);
        // by sorted position: group flag and index
        unsigned short* pos = reinterpret_cast<unsigned short*>(reinterpret_cast<char*>(tile) + W64_OFF_POS);
#pragma unroll
        for (int q = 0; q < 8; q++) {
            unsigned wds[4];
#pragma unroll
            for (int h = 0; h < 4; h++) {
                const int r = 8 * q + 2 * h;
                const unsigned e0 = ((unsigned)__double2loint(k[r]) & 0xFFFu) | ((unsigned)((gp >> r) & 1ull) << 15);
                const unsigned e1 = ((unsigned)__double2loint(k[r + 1]) & 0xFFFu) | ((unsigned)((gp >> (r + 1)) & 1ull) << 15);
                wds[h] = e0 | (e1 << 16);
            }
            reinterpret_cast<uint4*>(pos + 64 * lane)[q] = make_uint4(wds[0], wds[1], wds[2], wds[3]);
        }
    }
    return R;
}

// Second half: from the coefficients y (again) and the ranks in the tile, the sparsity term (return
// value, in every lane) and its gradient, which replaces y.
__device__ __forceinline__ double w64_gini_finish(double (&y)[64], const W64Rank& R, double* tile, int lane_in) {
    constexpr int n = 4096;
    int lane = lane_in;
    asm volatile("" : "+v"(lane));
    unsigned short* rk = reinterpret_cast<unsigned short*>(reinterpret_cast<char*>(tile) + W64_OFF_RANK);
    if (R.odd_rank) {
#pragma unroll
        for (int r = 0; r < 64; r++) tile[lane * 65 + r] = __builtin_fabs(y[r]);
        w64_lds_order();
        unsigned rank[64];
        w64_exact_ranks(y, rank, tile, lane);
        w64_lds_order();
#pragma unroll
        for (int r = 0; r < 64; r++) rk[r + 64 * lane] = (unsigned short)rank[r];
    } else if (R.any_group) {
        // low 12 bits of the exact magnitudes, by element index
        uint4* low4 = reinterpret_cast<uint4*>(reinterpret_cast<char*>(tile) + W64_OFF_LOW) + 8 * lane;
#pragma unroll
        for (int q = 0; q < 8; q++) {
            unsigned wds[4];
#pragma unroll
            for (int h = 0; h < 4; h++) {
                const unsigned l0 = (unsigned)__double2loint(y[8 * q + 2 * h]) & 0xFFFu;
                const unsigned l1 = (unsigned)__double2loint(y[8 * q + 2 * h + 1]) & 0xFFFu;
                wds[h] = l0 | (l1 << 16);
            }
            low4[q] = make_uint4(wds[0], wds[1], wds[2], wds[3]);
        }
        w64_lds_order();
        // members of a group: rank = group start + number of members below in (low bits, index)
        const unsigned short* low = reinterpret_cast<const unsigned short*>(reinterpret_cast<char*>(tile) + W64_OFF_LOW);
        const unsigned short* pos = reinterpret_cast<const unsigned short*>(reinterpret_cast<char*>(tile) + W64_OFF_POS);
        unsigned long long todo = R.grouped;
        while (todo) {
            const unsigned p = (unsigned)(lane * 64 + __builtin_ctzll(todo));
            todo &= todo - 1;
            const unsigned me = pos[p];
            const unsigned mine = ((unsigned)low[me & 0xFFFu] << 12) | (me & 0xFFFu);
            unsigned gs = p, w = me;
            while (w >> 15) { gs--; w = pos[gs]; }
            unsigned below = 0;
            for (unsigned q = gs; q < (unsigned)n; q++) {
                const unsigned o = pos[q];
                if (q > gs && !(o >> 15)) break;
                const unsigned other = ((unsigned)low[o & 0xFFFu] << 12) | (o & 0xFFFu);
                below += (other < mine) ? 1u : 0u;
            }
            rk[me & 0xFFFu] = (unsigned short)(gs + below + 1);
        }
    }
    w64_lds_order();
    const uint4* rk4 = reinterpret_cast<const uint4*>(rk + 64 * lane);
    double f = 0, sabs = 0;
#pragma unroll
    for (int q = 0; q < 8; q++) {
        const uint4 pk = rk4[q];
        const unsigned wds[4] = {pk.x, pk.y, pk.z, pk.w};
#pragma unroll
        for (int h = 0; h < 8; h++) {
            const unsigned rank = (wds[h >> 1] >> (16 * (h & 1))) & 0xFFFFu;
            const double coef = 2.0 * (double)rank - (double)n - 1.0;
            const double ay = __builtin_fabs(y[8 * q + h]);
            f = __builtin_fma(coef, ay, f);
            sabs += ay;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        f += __shfl_xor(f, off, 64);
        sabs += __shfl_xor(sabs, off, 64);
    }
    const double g = (double)n * sabs;
    const double inv_g2 = 1.0 / (g * g);
    if (!R.odd_sign) {
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const uint4 pk = rk4[q];
            const unsigned wds[4] = {pk.x, pk.y, pk.z, pk.w};
#pragma unroll
            for (int h = 0; h < 8; h++) {
                const unsigned rank = (wds[h >> 1] >> (16 * (h & 1))) & 0xFFFFu;
                const double coef = 2.0 * (double)rank - (double)n - 1.0;
                const double v = y[8 * q + h];
                const double sg = (v == 0.0) ? 0.0 : __builtin_copysign(1.0, v);  // == sign_ref in this range
                y[8 * q + h] = -((coef * sg) * g - f * ((double)n * sg)) * inv_g2;
            }
        }
    } else {
#pragma unroll
        for (int r = 0; r < 64; r++) {
            const double coef = 2.0 * (double)rk[r + 64 * lane] - (double)n - 1.0;
            const double sg = sign_ref(y[r]);
            y[r] = -((coef * sg) * g - f * ((double)n * sg)) * inv_g2;
        }
    }
    w64_lds_order();
    return 1.0 - f / g;
}

// One example row-wise into registers: v[c] = X(lane, c).  Eight scalar bases 4 KiB apart, so that
// every load is scalar base + lane offset + immediate and no per-lane 64-bit address is kept.
__device__ __forceinline__ void w64_load_rows(double (&v)[64], const double* Xe, int lane) {
#pragma unroll
    for (int j = 0; j < 8; j++) {
        typedef const __attribute__((address_space(1))) double* global_ptr;
        global_ptr bj = (global_ptr)(Xe + 512 * j);
        asm volatile("" : "+s"(bj));
#pragma unroll
        for (int c = 0; c < 8; c++) v[8 * j + c] = bj[lane + 64 * c];
    }
}


// One example row-wise back to memory: X(lane, c) = v[c].  Same addressing as w64_load_rows.
__device__ __forceinline__ void w64_store_rows(const double (&v)[64], double* Xe, int lane) {
#pragma unroll
    for (int j = 0; j < 8; j++) {
        typedef __attribute__((address_space(1))) double* global_ptr;
        global_ptr bj = (global_ptr)(Xe + 512 * j);
        asm volatile("" : "+s"(bj));
#pragma unroll
        for (int c = 0; c < 8; c++) bj[lane + 64 * c] = v[8 * j + c];
    }
}

// Forward transform of 64x64 examples (Wavenet::forward_, src/Wavenet.cxx:562-622), one wave per example.
// Everything leaves in row orientation (lane = row), so that every store instruction covers 512
// consecutive bytes; the column-stage results take one more trip through the tile for that.
template <int NP>
__global__ __launch_bounds__(256, 1) void k_forward64(FusedArgs A) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const FilterDev* filt = A.filt;
    double a[NP], b[NP];
#pragma unroll
    for (int j = 0; j < NP; j++) { a[j] = filt->a[j]; b[j] = filt->b[j]; }
    const int nslots = gridDim.x * 4;
    for (int e = blockIdx.x * 4 + wave; e < A.B; e += nslots) {
        const size_t off = (size_t)e * 4096;
        unsigned tile_off = (unsigned)wave * W64_TILE;
        asm volatile("" : "+s"(tile_off));
        double* tile = smem + tile_off;
        double* trow = tile + lane * 65;
        double v[64], w[64], p[64];
        w64_load_rows(v, A.X + off, lane);
        w64_fwd_levels<NP, 5, false>(v, p, w, trow, a, b);       // w = Y1(lane, :), p = row pyramid
        p[0] = 0;
        if (A.Prow) w64_store_rows(p, A.Prow + off, lane);
        if (A.Y1) w64_store_rows(w, A.Y1 + off, lane);
#pragma unroll
        for (int c = 0; c < 64; c++) trow[c] = w[c];
        w64_read_transposed(w, tile, lane);                       // w[r] = Y1(r, lane)
        w64_fwd_levels<NP, 5, false>(w, p, v, trow, a, b);       // v = Y(:, lane), p = column pyramid
        p[0] = 0;
        if (A.Y) {
            w64_transpose(v, tile, lane);                         // v[c] = Y(lane, c)
            w64_store_rows(v, A.Y + off, lane);
        }
        if (A.Pcol) {
            w64_transpose(p, tile, lane);
            w64_store_rows(p, A.Pcol + off, lane);
        }
    }
}

// Inverse transform of 64x64 examples (Wavenet::inverse_, src/Wavenet.cxx:624-663): columns, then rows.
template <int NP>
__global__ __launch_bounds__(256, 1) void k_inverse64(FusedArgs A) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const FilterDev* filt = A.filt;
    double a[NP], b[NP];
#pragma unroll
    for (int j = 0; j < NP; j++) { a[j] = filt->a[j]; b[j] = filt->b[j]; }
    double gL[NP], gH[NP];   // not used without the gradient
    const int nslots = gridDim.x * 4;
    for (int e = blockIdx.x * 4 + wave; e < A.B; e += nslots) {
        const size_t off = (size_t)e * 4096;
        unsigned tile_off = (unsigned)wave * W64_TILE;
        asm volatile("" : "+s"(tile_off));
        double* tile = smem + tile_off;
        double* trow = tile + lane * 65;
        double v[64];
        w64_load_rows(v, A.Yin + off, lane);                      // v[c] = Y(lane, c)
        w64_transpose(v, tile, lane);                             // v[r] = Y(r, lane): lane = column
        w64_bwd_levels<NP, 0, W64_TOP_TILE, false>(v, v, trow, a, b, gL, gH);   // columns, result to the tile
        w64_read_transposed(v, tile, lane);                       // v[c] = Y1(lane, c): lane = row
        w64_bwd_levels<NP, 0, W64_TOP_REGS, false>(v, v, trow, a, b, gL, gH);   // rows
        w64_store_rows(v, A.Xout + off, lane);
    }
}


// SparseTerm / SparseTermDeriv (src/CostFunctions.cxx:6-117) on sets of 4096 coefficients, one wave per set.
template <int UNUSED = 0>   // a template only so that every translation unit may include this header
__global__ __launch_bounds__(256, 1) void k_sparse64(FusedArgs A) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nslots = gridDim.x * 4;
    for (int e = blockIdx.x * 4 + wave; e < A.B; e += nslots) {
        const size_t off = (size_t)e * 4096;
        unsigned tile_off = (unsigned)wave * W64_TILE;
        asm volatile("" : "+s"(tile_off));
        double* tile = smem + tile_off;
        double v[64];
        w64_load_rows(v, A.Yin + off, lane);
        w64_transpose(v, tile, lane);                 // v[r] = element r + 64 * lane
        const W64Rank rank = w64_rank_sort(v, tile, lane);
        const double S = w64_gini_finish(v, rank, tile, lane);
        if (A.Delta) {
            w64_transpose(v, tile, lane);
            w64_store_rows(v, A.Delta + off, lane);
        }
        if (A.S && lane == 0) A.S[e] = S;
    }
}

// Backpropagation of 64x64 examples from given errors and activations (Wavenet::backpropagate_,
// src/Wavenet.cxx:665-711).  Delta (optional) receives the errors after the column stage.
template <int NP>
__global__ __launch_bounds__(256, 1) void k_backprop64(FusedArgs A) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const FilterDev* filt = A.filt;
    double a[NP], b[NP];
#pragma unroll
    for (int j = 0; j < NP; j++) { a[j] = filt->a[j]; b[j] = filt->b[j]; }
    const int N = filt->N, dpad = filt->d;
    const int nslots = gridDim.x * 4;
    for (int e = blockIdx.x * 4 + wave; e < A.B; e += nslots) {
        const size_t off = (size_t)e * 4096;
        unsigned tile_off = (unsigned)wave * W64_TILE;
        asm volatile("" : "+s"(tile_off));
        double* tile = smem + tile_off;
        double* trow = tile + lane * 65;
        double v[64], w[64];
        w64_load_rows(v, A.Din + off, lane);
        w64_transpose(v, tile, lane);                 // v[r] = errors on Y(r, lane)
        w64_load_rows(w, A.Y1In + off, lane);
        w64_transpose(w, tile, lane);                 // w[r] = Y1(r, lane)
        {   // column pyramid, given row-wise, into the tile rows of its columns
            double p[64];
            w64_load_rows(p, A.PcolIn + off, lane);
#pragma unroll
            for (int c = 0; c < 64; c++) tile[c * 65 + lane] = p[c];
        }
        double gL[NP], gH[NP];
#pragma unroll
        for (int u = 0; u < NP; u++) { gL[u] = 0; gH[u] = 0; }
        w64_bwd_levels<NP, 0, W64_TOP_TILE, true>(v, w, trow, a, b, gL, gH);
        w64_read_transposed(v, tile, lane);           // v[c] = errors on Y1(lane, c)
        if (A.Delta) w64_store_rows(v, A.Delta + off, lane);
        w64_load_rows(w, A.X + off, lane);
        {
            double p[64];
            w64_load_rows(p, A.ProwIn + off, lane);
#pragma unroll
            for (int c = 0; c < 64; c++) trow[c] = p[c];
        }
        w64_bwd_levels<NP, 0, W64_TOP_NONE, true>(v, w, trow, a, b, gL, gH);
        double mine;
        {
            double c[NP];
#pragma unroll
            for (int kp = 0; kp < NP; kp++) c[kp] = filt->vL[kp] * gL[kp] + filt->sH[kp] * gH[NP - 1 - kp];
            mine = wave_scatter_sums<NP>(c, lane);      // the total of tap lane mod NP
        }
        const int k = lane - dpad;
        if (lane < NP && k >= 0 && k < N) A.grad[(size_t)e * N + k] = mine;
    }
}


// Sparsity cost of every 64x64 example under every filter of a table (Wavenet::costMap,
// src/Wavenet.cxx:278-324, evaluates cost(forward(X)) over a grid of filters): work item w = f * B + e
// is one wave's forward transform and Gini term, S[w] the result.  The taps change per item, so they
// are fetched inside the loop.
template <int NP>
__global__ __launch_bounds__(256, 1) void k_cost64(FusedArgs A) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nslots = gridDim.x * 4;
    const long long items = (long long)A.F * A.B;
    for (long long item = blockIdx.x * 4 + wave; item < items; item += nslots) {
        const int f = (int)(item / A.B), e = (int)(item - (long long)f * A.B);
        const FilterDev* filt = A.filt + f;
        double a[NP], b[NP];
#pragma unroll
        for (int j = 0; j < NP; j++) { a[j] = filt->a[j]; b[j] = filt->b[j]; }
        if constexpr (NP > 16) {   // as in k_train64: only the signs of coefficients change, which the Gini term does not see
#pragma unroll
            for (int j = 0; j < NP; j++) b[j] = (j & 1) ? -a[NP - 1 - j] : a[NP - 1 - j];
        }
        unsigned tile_off = (unsigned)wave * W64_TILE;
        asm volatile("" : "+s"(tile_off));
        double* tile = smem + tile_off;
        double* trow = tile + lane * 65;
        double v[64], w[64], p[64];
        w64_load_rows(v, A.X + (size_t)e * 4096, lane);
        w64_fwd_levels<NP, 5, true>(v, p, w, trow, a, b);
        w64_read_transposed(w, tile, lane);
        w64_fwd_levels<NP, 5, false>(w, p, v, trow, a, b);   // v = Y(:, lane)
        const W64Rank rank = w64_rank_sort(v, tile, lane);
        const double S = w64_gini_finish(v, rank, tile, lane);
        if (lane == 0) A.S[item] = S;
    }
}

// ---- kernel ---------------------------------------------------------------------------------------------
// 256 threads = 4 waves = 4 examples in flight per workgroup; grid-stride over the batch.
template <int NP>
__global__ __launch_bounds__(256, 1) void k_train64(FusedArgs A) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    // the wave index as a scalar: example addresses then live in SGPRs and the loads take the
    // scalar-base form, instead of 64-bit per-lane addresses that end up in scratch memory
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const FilterDev* filt = A.filt;
    double a[NP], b[NP];
#pragma unroll
    for (int j = 0; j < NP; j++) { a[j] = filt->a[j]; b[j] = filt->b[j]; }
    const int N = filt->N, dpad = filt->d;
    // Above 16 taps (at 16 it measures the same) the kernel holds the high-pass taps as signed low-pass taps, b'[j] = (-1)^j a[NP - 1 - j], not
    // the stored b = sg b', sg = (-1)^(padding in front of the filter): half the registers for taps.  No sign is put
    // right on the way: coefficient (r, c) comes out times s_r s_c, with s = sg on the high-pass indices and 1 on
    // index 0; the Gini term does not see that and its derivative is odd in a coefficient, so the errors carry the
    // same signs, the propagation multiplies two of them, and the high-pass window sums come out times sg, which the
    // sign table takes back.  Negations are exact: what leaves the kernel is the same bit for bit (k_train1d1024 too).
    constexpr bool B_FROM_A = (NP > 16);
    if constexpr (B_FROM_A) {
#pragma unroll
        for (int j = 0; j < NP; j++) b[j] = (j & 1) ? -a[NP - 1 - j] : a[NP - 1 - j];
    }
    const double sg = (B_FROM_A && (dpad & 1)) ? -1.0 : 1.0;

    double accG = 0, accS = 0, accBad = 0;
    const int nslots = gridDim.x * 4;
    for (int e = blockIdx.x * 4 + wave; e < A.B; e += nslots) {
        const double* Xe = A.X + (size_t)e * 4096;
        // LDS addresses are rebuilt from an opaque offset every iteration: hoisted out of the loop
        // they would only occupy registers (or scratch) all the way through
        unsigned tile_off = (unsigned)wave * W64_TILE;
        asm volatile("" : "+s"(tile_off));
        double* tile = smem + tile_off;
        double* trow = tile + lane * 65;
        double v[64], w[64], p[64];
        // rows: lane = row, v[c] = X(lane, c); the row-stage coefficients Y1(lane, :) go to the tile
        w64_load_rows(v, Xe, lane);
        w64_fwd_levels<NP, 5, true>(v, p, w, trow, a, b);
        w64_read_transposed(w, tile, lane);                  // w[r] = Y1(r, lane): lane = column
        w64_fwd_levels<NP, 5, false>(w, p, v, trow, a, b);   // v = Y(:, lane)
        const W64Rank rank = w64_rank_sort(v, tile, lane);
        // Y again, from values the compiler cannot tell are the same: otherwise it would keep the
        // first copy alive across the sort instead
#pragma unroll
        for (int r = 0; r < 64; r++) asm volatile("" : "+v"(w[r]));
        w64_fwd_levels<NP, 5, false>(w, p, v, trow, a, b);
        const double S = w64_gini_finish(v, rank, tile, lane);   // v = delta(:, lane)

        double gL[NP], gH[NP];
#pragma unroll
        for (int u = 0; u < NP; u++) { gL[u] = 0; gH[u] = 0; }
        w64_pyramid_to_tile<NP, 5>(w, p, trow, a);                // column pyramid into the lane's tile row
        w64_bwd_levels<NP, 0, W64_TOP_TILE, true>(v, w, trow, a, b, gL, gH);    // columns; errors on Y1(:, lane) to the tile
        w64_read_transposed(v, tile, lane);                        // v[c] = D1(lane, c): lane = row
        w64_load_rows(w, Xe, lane);
        w64_pyramid_to_tile<NP, 5>(w, p, trow, a);
        w64_bwd_levels<NP, 0, W64_TOP_NONE, true>(v, w, trow, a, b, gL, gH);   // rows

        // tap k' collects gL[k'] and sH[k'] gH[NP-1-k'], summed over the 64 lanes
        double mine;
        {
            double c[NP];
#pragma unroll
            for (int kp = 0; kp < NP; kp++) c[kp] = filt->vL[kp] * gL[kp] + (sg * filt->sH[kp]) * gH[NP - 1 - kp];
#ifndef W64_PLAIN_BUTTERFLY
            mine = wave_scatter_sums<NP>(c, lane);      // the total of tap lane mod NP
#else
            mine = 0;
#pragma unroll
            for (int kp = 0; kp < NP; kp++) {
                double t = c[kp];
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off, 64);
                if (lane == kp) mine = t;
            }
#endif
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
    // The four waves' running sums are added in wave order and leave as one slot per workgroup: the reduction at
    // the end of the step then has a quarter of the slots to read.
    double bad = accBad;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) bad += __shfl_xor(bad, off, 64);
    __syncthreads();                                    // every wave is done with the tiles
    double* wsum = smem + (size_t)wave * PSTRIDE;       // [4][PSTRIDE]
    const int k = lane - dpad;
    if (lane < NP && k >= 0 && k < N) wsum[k] = accG;
    if (lane == 0) { wsum[N] = accS; wsum[N + 1] = bad; }
    __syncthreads();
    const int tid = threadIdx.x;
    if (tid <= N + 1) {
        double s = smem[tid];
#pragma unroll
        for (int w = 1; w < 4; w++) s += smem[w * PSTRIDE + tid];
        store_partial(A.partial + (size_t)blockIdx.x * PSTRIDE + tid, s);
    }
    step_tail<256>(A.tail, A.partial, (int)gridDim.x, PSTRIDE, smem + 4 * PSTRIDE, tid);
}

}  // namespace wn
