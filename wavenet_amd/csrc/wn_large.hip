// Streamed path for examples that do not fit the LDS-resident kernels (R*C > 4096), e.g. 512x512
// images and 2^20-sample signals.  Pyramids, coefficients and errors live in a device workspace (288 GB of
// HBM make that cheap).  Lines of 128 .. 1024 samples run through the one-wave-per-line kernels of wn_lines.h;
// every other length level by level, one kernel per level (or per two or three levels) over a chunk of the batch,
// with the levels below 4096 samples of a contiguous line in one launch.  The Gini ranks come from the per-example
// bucket ranking of wn_rank.h; when that gives up on a chunk (too many magnitudes crowd into one bucket) from
// stable device-wide radix sorts (one sort of 32-bit keys per slice of the chunk plus an exact re-ranking of
// equal-key groups, see gini_large; two sorts on 64-bit keys behind that).  All reductions of data are two-stage
// and ordered (atomics only on counters, cursors and maxima).
//
// Reference stages: forward_ (src/Wavenet.cxx:404-445, 562-622), SparseTerm(Deriv)
// (src/CostFunctions.cxx:6-117), backpropagate_ (src/Wavenet.cxx:474-556, 665-711).
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include <algorithm>
#include <vector>

#include "wn_internal.h"
#include "wn_rank.h"

extern "C" const void* wn_kernel_lines_np2(int kind, int M);
extern "C" const void* wn_kernel_lines_np4(int kind, int M);
extern "C" const void* wn_kernel_lines_np8(int kind, int M);
extern "C" const void* wn_kernel_lines_np16(int kind, int M);

namespace wn {
namespace {

// One-wave-per-line kernels (wn_lines.h, compiled per tap count in wn_lines_np.hip, up to 16 taps).
enum { LN_FWD_COLS = 0, LN_BWD_COLS = 1, LN_BWD_COLS_RANKS = 2, LN_FWD_ROWS = 3, LN_GRAD_ROWS = 4, LN_FWD_COLS_HIST = 5, LN_BWD_COLS_REGIONS = 6 };
constexpr int LN_HIST_T = 1024, LN_HIST_BINS = 36864;   // kc_fwd_cols_hist (wn_lines.h)
const void* line_kernel(int np, int kind, int M) {
    switch (np) {
        case 2: return wn_kernel_lines_np2(kind, M);
        case 4: return wn_kernel_lines_np4(kind, M);
        case 8: return wn_kernel_lines_np8(kind, M);
        case 16: return wn_kernel_lines_np16(kind, M);
    }
    return nullptr;
}
// LDS tile of the row kernels: 16 rows (8 for lines of 1024) of 64 (M + 1) + 2 (4) doubles (LineTile in wn_lines.h)
int line_tile_rows(int M) { return M == 16 ? 8 : 16; }
size_t line_tile_bytes(int M) { return (size_t)line_tile_rows(M) * (64 * (M + 1) + (M == 16 ? 4 : 2)) * sizeof(double); }

constexpr int BT = 256;        // threads per workgroup
constexpr int GXMAX = 64;      // workgroups per example in the reducing kernels
constexpr int LN_GX = 64;      // workgroups per example of the one-wave-per-line kernels (wn_lines.h)
static_assert(GXMAX == LN_GX, "the line kernels of wn_lines.h write the same partial-sum slots");

struct Geom {                   // lines of one pass inside an example of n doubles
    int es, ls, lgNL, lgLen;
    int lanesOnLines;
};

__device__ __forceinline__ void decode(int w, int lgNL, int lgInner, int lanesOnLines, int& line, int& k) {
    if (lanesOnLines) { line = w & ((1 << lgNL) - 1); k = w >> lgNL; }
    else              { k = w & ((1 << lgInner) - 1); line = w >> lgInner; }
}


// The taps for the lifetime of a kernel: values that are the same in every lane, which the compiler keeps in scalar
// registers.  Two filters of 32 taps are 128 of those, more than a wave has: the compiler then parks them in the lanes
// of vector registers and fetches every tap back with a v_readlane in front of its multiply-add -- as many moves as
// arithmetic.  FROM_A: the high-pass taps are not loaded but written as what they are, b[j] = (-1)^j a[NP - 1 - j]
// (src/HighpassOperator.cxx:14-18; a negated operand costs nothing), which leaves 64 registers to hold.  The stored
// b carries the parity of the TRUE tap index, j minus the padding in front of the filter: FROM_A is for an even
// padding only (the host code sends the other case to the kernels that load b).  Padded entries come out as +-0
// instead of +0, which no sum can tell apart.
template <int NP, bool FROM_A = false>
__device__ __forceinline__ void load_taps(const FilterDev* filt, double (&a)[NP], double (&b)[NP]) {
#pragma unroll
    for (int j = 0; j < NP; j++) a[j] = filt->a[j];
#pragma unroll
    for (int j = 0; j < NP; j++) {
        if constexpr (FROM_A) b[j] = (j & 1) ? -a[NP - 1 - j] : a[NP - 1 - j];
        else b[j] = filt->b[j];
    }
}
#ifndef WN_TAPS_FROM_A_ABOVE
#define WN_TAPS_FROM_A_ABOVE 16
#endif
constexpr bool taps_from_a(int NP) { return NP > WN_TAPS_FROM_A_ABOVE; }   // the tap counts whose LDS-staged and tail kernels use FROM_A

// The ranking wants the largest exponent word (upper 32 bits of |c|) among an example's coefficients before anything
// else (kr_max, wn_rank.h: one more pass over them).  The forward kernels of long contiguous lines keep track of it
// for the coefficients they write, and every workgroup leaves its maximum in a slot of its own, kparts[example *
// stride + slot0 + blockIdx.x]; kr_hist takes the maximum over an example's slots.  (Not an atomic on one word per
// example: atomics of device scope are carried out behind the L2s, one after the other per address -- 512 of them per
// example and launch were a fifth of the forward kernel's time, and a look at the word first, which has to be of device
// scope as well, six times its time.)
struct KmaxSlots {
    unsigned* parts;     // null: not wanted
    int stride, slot0;   // slots per example; first slot of this launch
};
__device__ __forceinline__ unsigned exponent_word(double c) { return (unsigned)__double2hiint(c) & 0x7FFFFFFFu; }
template <int T>
__device__ __forceinline__ void leave_kmax(unsigned m, const KmaxSlots& ks, unsigned* wmax /* T / 64 words of LDS */) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 1; w < T / 64; w++) m = max(m, wmax[w]);
        ks.parts[(size_t)blockIdx.y * ks.stride + ks.slot0 + blockIdx.x] = m;
    }
}

// lo[k] = sum_j a[j] in[(N/2 + 2k - j) mod len], hi[k] with b; one level, whole chunk (grid.y = example).
template <int NP>
__global__ __launch_bounds__(BT) void kb_fwd_level(const double* __restrict__ in, double* __restrict__ lo,
                                                   double* __restrict__ hi, size_t xs, Geom g, int lgHalf,
                                                   const FilterDev* filt) {
    double a[NP], b[NP];
    load_taps<NP>(filt, a, b);
    const size_t off = (size_t)blockIdx.y * xs;
    const int W = 1 << (g.lgNL + lgHalf), lmask = (2 << lgHalf) - 1;
    for (int w = blockIdx.x * BT + threadIdx.x; w < W; w += gridDim.x * BT) {
        int line, k;
        decode(w, g.lgNL, lgHalf, g.lanesOnLines, line, k);
        const double* src = in + off + (size_t)line * g.ls;
        const int base = 2 * k + NP / 2;
        double sl = 0, sh = 0;
#pragma unroll
        for (int j = 0; j < NP; j++) {
            const double v = src[(size_t)((base - j) & lmask) * g.es];
            sl = __builtin_fma(a[j], v, sl);
            sh = __builtin_fma(b[j], v, sh);
        }
        lo[off + (size_t)line * g.ls + (size_t)k * g.es] = sl;
        hi[off + (size_t)line * g.ls + (size_t)k * g.es] = sh;
    }
}

// out[line][0] = P[line][1] (lo_0), P[line][0] = 0
__global__ void kb_finish_pass(double* out, double* P, size_t xs, Geom g) {
    const size_t off = (size_t)blockIdx.y * xs;
    for (int line = blockIdx.x * BT + threadIdx.x; line < (1 << g.lgNL); line += gridDim.x * BT) {
        out[off + (size_t)line * g.ls] = P[off + (size_t)line * g.ls + g.es];
        P[off + (size_t)line * g.ls] = 0;
    }
}

// Filter-gradient correlation of one level (see bwd_level in wn_device.h for the indexing).
// partial[(example * pgx + workgroup) * 2NP + u] accumulates over the launches of one backward (pgx: slots per
// example, GXMAX unless the chunk has so few examples that more workgroups per example are needed to fill the chip).
template <int NP>
__global__ __launch_bounds__(BT) void kb_grad_level(const double* __restrict__ lp, const double* __restrict__ hp,
                                                    const double* __restrict__ act, size_t xs, Geom g, int lgLen,
                                                    double* partial, int pgx) {
    __shared__ double red[(BT / 64) * 2 * NP];
    double gL[NP], gH[NP];
#pragma unroll
    for (int u = 0; u < NP; u++) { gL[u] = 0; gH[u] = 0; }
    const size_t off = (size_t)blockIdx.y * xs;
    const int W = 1 << (g.lgNL + lgLen), amask = (2 << lgLen) - 1;
    for (int w = blockIdx.x * BT + threadIdx.x; w < W; w += gridDim.x * BT) {
        int line, r;
        decode(w, g.lgNL, lgLen, g.lanesOnLines, line, r);
        const size_t lo = off + (size_t)line * g.ls;
        const double dl = lp[lo + (size_t)r * g.es];
        const double dh = hp[lo + (size_t)r * g.es];
        const int base = 2 * r + NP / 2;
#pragma unroll
        for (int u = 0; u < NP; u++) {
            const double v = act[lo + (size_t)((base - u) & amask) * g.es];
            gL[u] = __builtin_fma(dl, v, gL[u]);
            gH[u] = __builtin_fma(dh, v, gH[u]);
        }
    }
    const int tid = threadIdx.x;
    const double s = block_window_sums<NP, BT>(gL, gH, red, tid);
    if (tid < 2 * NP) partial[((size_t)blockIdx.y * pgx + blockIdx.x) * 2 * NP + tid] += s;
}

// out[c] = sum over (r, j) with (N/2 + 2r - j) mod 2len == c of a[j] lp[r] + b[j] hp[r]
template <int NP>
__global__ __launch_bounds__(BT) void kb_prop_level(const double* __restrict__ lp, const double* __restrict__ hp,
                                                    double* __restrict__ out, size_t xs, Geom g, int lgLen,
                                                    const FilterDev* filt) {
    double a[NP], b[NP];
    load_taps<NP>(filt, a, b);
    constexpr int S = NP / 2;
    constexpr int JE = S & 1, JO = (S + 1) & 1;
    constexpr int CE = (JE - S) / 2, CO = (1 + JO - S) / 2;
    const size_t off = (size_t)blockIdx.y * xs;
    const int W = 1 << (g.lgNL + lgLen), rmask = (1 << lgLen) - 1;
    for (int w = blockIdx.x * BT + threadIdx.x; w < W; w += gridDim.x * BT) {
        int line, p;
        decode(w, g.lgNL, lgLen, g.lanesOnLines, line, p);
        const size_t lo = off + (size_t)line * g.ls;
        double se = 0, so = 0;
#pragma unroll
        for (int i = 0; i < NP / 2; i++) {
            const size_t re = lo + (size_t)((p + i + CE) & rmask) * g.es;
            const size_t ro = lo + (size_t)((p + i + CO) & rmask) * g.es;
            se = __builtin_fma(a[JE + 2 * i], lp[re], se);
            se = __builtin_fma(b[JE + 2 * i], hp[re], se);
            so = __builtin_fma(a[JO + 2 * i], lp[ro], so);
            so = __builtin_fma(b[JO + 2 * i], hp[ro], so);
        }
        out[lo + (size_t)(2 * p) * g.es] = se;
        out[lo + (size_t)(2 * p + 1) * g.es] = so;
    }
}

// kb_grad_level and kb_prop_level of one level in one pass over lp / hp: the errors are read from HBM once.
template <int NP>
__global__ __launch_bounds__(BT) void kb_bwd_level(const double* __restrict__ lp, const double* __restrict__ hp,
                                                   const double* __restrict__ act, double* __restrict__ out, size_t xs, Geom g,
                                                   int lgLen, const FilterDev* filt, double* partial, int pgx) {
    __shared__ double red[(BT / 64) * 2 * NP];
    double a[NP], b[NP];
    load_taps<NP>(filt, a, b);
    double gL[NP], gH[NP];
#pragma unroll
    for (int u = 0; u < NP; u++) { gL[u] = 0; gH[u] = 0; }
    constexpr int S = NP / 2;
    constexpr int JE = S & 1, JO = (S + 1) & 1;
    constexpr int CE = (JE - S) / 2, CO = (1 + JO - S) / 2;
    const size_t off = (size_t)blockIdx.y * xs;
    const int W = 1 << (g.lgNL + lgLen), amask = (2 << lgLen) - 1, rmask = (1 << lgLen) - 1;
    for (int w = blockIdx.x * BT + threadIdx.x; w < W; w += gridDim.x * BT) {
        int line, r;
        decode(w, g.lgNL, lgLen, g.lanesOnLines, line, r);
        const size_t lo = off + (size_t)line * g.ls;
        const double dl = lp[lo + (size_t)r * g.es];
        const double dh = hp[lo + (size_t)r * g.es];
        const int base = 2 * r + NP / 2;
#pragma unroll
        for (int u = 0; u < NP; u++) {
            const double v = act[lo + (size_t)((base - u) & amask) * g.es];
            gL[u] = __builtin_fma(dl, v, gL[u]);
            gH[u] = __builtin_fma(dh, v, gH[u]);
        }
        double se = 0, so = 0;
#pragma unroll
        for (int i = 0; i < NP / 2; i++) {
            const size_t re = lo + (size_t)((r + i + CE) & rmask) * g.es;
            const size_t ro = lo + (size_t)((r + i + CO) & rmask) * g.es;
            se = __builtin_fma(a[JE + 2 * i], lp[re], se);
            se = __builtin_fma(b[JE + 2 * i], hp[re], se);
            so = __builtin_fma(a[JO + 2 * i], lp[ro], so);
            so = __builtin_fma(b[JO + 2 * i], hp[ro], so);
        }
        out[lo + (size_t)(2 * r) * g.es] = se;
        out[lo + (size_t)(2 * r + 1) * g.es] = so;
    }
    const int tid = threadIdx.x;
    const double s = block_window_sums<NP, BT>(gL, gH, red, tid);
    if (tid < 2 * NP) partial[((size_t)blockIdx.y * pgx + blockIdx.x) * 2 * NP + tid] += s;
}

// ---- level kernels for contiguous lines, operands staged in LDS ------------------------------------------
// For lines whose samples are contiguous (the column pass; 1D signals) a thread of the kernels above loads
// its N taps' worth of operands itself, 8 bytes at a 16-byte stride across the wave: at 16 and 32 taps that
// is bound by L1 requests.  Here a workgroup produces LT consecutive outputs of one line: it stages the
// 2 LT + N - 1 operands in LDS with coalesced loads and every thread reads its window from there, two
// doubles per instruction (the window of output k starts at the even offset 2 k).
constexpr int LT = 256;   // outputs per workgroup = threads
#ifndef WN_LDS_LEVEL_TAPS
#define WN_LDS_LEVEL_TAPS 16
#endif
constexpr int LDS_LEVEL_TAPS = WN_LDS_LEVEL_TAPS;   // from this compiled tap count on

template <int NP>
__device__ __forceinline__ void stage_window(double* tile, const double* __restrict__ line, int first, int mask) {
    // tile[t] = line[(first + t) & mask], t < 2 LT + NP (one more than needed keeps the pairs whole)
    for (int t = threadIdx.x; t < 2 * LT + NP; t += LT) tile[t] = line[(first + t) & mask];
}
// the window of thread tid in ascending order: win[u] = tile[2 tid + u] = operand of tap NP - 1 - u
template <int NP>
__device__ __forceinline__ void read_window(const double* tile, double (&win)[NP]) {
    const double2* p = reinterpret_cast<const double2*>(tile + 2 * threadIdx.x);
#pragma unroll
    for (int u = 0; u < NP / 2; u++) { const double2 t = p[u]; win[2 * u] = t.x; win[2 * u + 1] = t.y; }
}

// kb_fwd_level for contiguous lines with at least LT outputs: grid (lines * half / LT, examples)
template <int NP>
__global__ __launch_bounds__(LT) void kb_fwd_level_lds(const double* __restrict__ in, double* __restrict__ lo, double* __restrict__ hi,
                                                       size_t xs, Geom g, int lgHalf, const FilterDev* filt, KmaxSlots ks) {
    __shared__ __attribute__((aligned(16))) double tile[2 * LT + NP];
    __shared__ unsigned wmax[LT / 64];
    double a[NP], b[NP];
    load_taps<NP, taps_from_a(NP)>(filt, a, b);
    const int w0 = blockIdx.x * LT;
    const int line = w0 >> lgHalf, k0 = w0 & ((1 << lgHalf) - 1);
    const size_t off = (size_t)blockIdx.y * xs + (size_t)line * g.ls;
    stage_window<NP>(tile, in + off, 2 * k0 + NP / 2 - (NP - 1), (2 << lgHalf) - 1);
    __syncthreads();
    double win[NP];
    read_window<NP>(tile, win);
    double sl = 0, sh = 0;
#pragma unroll
    for (int j = 0; j < NP; j++) {
        sl = __builtin_fma(a[j], win[NP - 1 - j], sl);
        sh = __builtin_fma(b[j], win[NP - 1 - j], sh);
    }
    lo[off + k0 + threadIdx.x] = sl;
    hi[off + k0 + threadIdx.x] = sh;
    if (ks.parts) leave_kmax<LT>(exponent_word(sh), ks, wmax);
}

// NLEV consecutive levels of kb_fwd_level_lds in one launch: a workgroup produces LT outputs of the lowest of them and
// what that takes of the levels above -- 2 LT + NP - 1 low-pass values of the level before, and so on up to
// 2^NLEV LT + (2^NLEV - 1)(NP - 1) inputs -- with the intermediate low-pass values handed down through LDS instead of
// through memory.  The halos are computed again by the neighbouring workgroups ((NP - 1) / LT of the work per level).
// Every level's outputs still go to P / out as the single-level kernel writes them (the backward pass reads them).
// in: the input of the top level i (2^(i+1) samples per line).  grid (lines * 2^(i-NLEV+1) / LT, examples).
template <int NP, int NLEV>
struct FwdFused {
    // stage s = 0 .. NLEV-1 computes level i - s.  NEED[s]: low-pass outputs of stage s the workgroup computes;
    // C[s]: its first output, relative to (first owned output of the lowest stage) * 2^(NLEV-1-s)
    static constexpr int need(int s) { return s == NLEV - 1 ? LT : 2 * need(s + 1) + NP - 1; }
    static constexpr int first(int s) { return s == NLEV - 1 ? 0 : 2 * first(s + 1) + NP / 2 - (NP - 1); }
    static constexpr int IN = 2 * need(0) + NP - 1;                  // inputs staged
    static constexpr int IN_FIRST = 2 * first(0) + NP / 2 - (NP - 1);
    static constexpr int lds_doubles() {
        int t = IN + 1;
        for (int s = 0; s < NLEV - 1; s++) t += need(s) + 1;
        return t;
    }
};

template <int NP, int NLEV, int S>
__device__ __forceinline__ void fwd_fused_stage(const double* src, double* tiles, double* P, double* out, size_t off, int K0, int i,
                                                const double (&a)[NP], const double (&b)[NP], unsigned& kmine) {
    using F = FwdFused<NP, NLEV>;
    constexpr int NEED = F::need(S), OWN = LT << (NLEV - 1 - S), OWN_OFF = -F::first(S);
    const int lvl = i - S;
    const size_t own0 = (size_t)K0 << (NLEV - 1 - S);               // first owned output of this stage, within the line
    for (int q = threadIdx.x; q < NEED; q += LT) {
        // the window of output q in ascending order, win[u] = operand of tap NP - 1 - u; the taps are added in the
        // order of kb_fwd_level_lds, so a level comes out bit for bit as from there
        const double2* p = reinterpret_cast<const double2*>(src + 2 * q);
        double win[NP];
#pragma unroll
        for (int u = 0; u < NP / 2; u++) { const double2 t = p[u]; win[2 * u] = t.x; win[2 * u + 1] = t.y; }
        double sl = 0, sh = 0;
#pragma unroll
        for (int j = 0; j < NP; j++) {
            sl = __builtin_fma(a[j], win[NP - 1 - j], sl);
            sh = __builtin_fma(b[j], win[NP - 1 - j], sh);
        }
        if constexpr (S < NLEV - 1) tiles[q] = sl;
        const int o = q - OWN_OFF;
        if (o >= 0 && o < OWN) {
            P[off + ((size_t)1 << lvl) + own0 + o] = sl;
            out[off + ((size_t)1 << lvl) + own0 + o] = sh;
            kmine = max(kmine, exponent_word(sh));
        }
    }
    if constexpr (S < NLEV - 1) {
        __syncthreads();
        fwd_fused_stage<NP, NLEV, S + 1>(tiles, tiles + NEED + 1, P, out, off, K0, i, a, b, kmine);
    }
}

template <int NP, int NLEV>
__global__ __launch_bounds__(LT) void kb_fwd_levels_lds(const double* __restrict__ in, double* __restrict__ P, double* __restrict__ out,
                                                        size_t xs, Geom g, int i, const FilterDev* filt, KmaxSlots ks) {
    using F = FwdFused<NP, NLEV>;
    extern __shared__ __attribute__((aligned(16))) double fused_lds[];
    __shared__ unsigned wmax[LT / 64];
    double a[NP], b[NP];
    load_taps<NP, taps_from_a(NP)>(filt, a, b);
    const int lgLow = i - (NLEV - 1);                                // outputs per line of the lowest stage: 2^lgLow
    const int w0 = blockIdx.x * LT;
    const int line = w0 >> lgLow, K0 = w0 & ((1 << lgLow) - 1);
    const size_t off = (size_t)blockIdx.y * xs + (size_t)line * g.ls;
    // inputs (2 << i per line, periodic) from position (K0 << NLEV) + IN_FIRST on
    const int first = (K0 << NLEV) + F::IN_FIRST, mask = (2 << i) - 1;
    for (int t = threadIdx.x; t < F::IN + 1; t += LT) fused_lds[t] = in[off + ((first + t) & mask)];
    __syncthreads();
    unsigned kmine = 0;
    fwd_fused_stage<NP, NLEV, 0>(fused_lds, fused_lds + F::IN + 1, P, out, off, K0, i, a, b, kmine);
    if (ks.parts) leave_kmax<LT>(kmine, ks, wmax);
}

// kb_bwd_level / kb_grad_level (PROP false) for contiguous lines with at least LT errors per line:
// grid (workgroups per example <= pgx, examples); a workgroup walks over tiles of LT errors.  The operands of the
// next tile are fetched into registers while the current one is worked on from LDS, so that a step does not
// wait for a round trip to memory.
// EPT: consecutive errors per thread.  The windows of neighbouring errors overlap in all but two activations, and
// the errors their propagation reads in all but one: at two per thread a tile of 2 LT errors takes half the LDS reads
// per error (and half the barriers).
#ifndef WN_BWD_LDS_EPT
#define WN_BWD_LDS_EPT 2
#endif
constexpr int BWD_LDS_EPT = WN_BWD_LDS_EPT;
#ifndef WN_BWD_LDS_FILL
#define WN_BWD_LDS_FILL 2
#endif
constexpr int BWD_LDS_FILL = WN_BWD_LDS_FILL;   // workgroups per CU the grid of kb_bwd_level_lds is sized for
template <int NP, bool PROP, int EPT>
__global__ __launch_bounds__(LT) void kb_bwd_level_lds(const double* __restrict__ lp, const double* __restrict__ hp,
                                                       const double* __restrict__ act, double* __restrict__ out, size_t xs, Geom g,
                                                       int lgLen, const FilterDev* filt, double* partial, int pgx) {
    static_assert(EPT == 1 || EPT % 2 == 0, "one error per thread, or an even number (16-byte LDS reads)");
    constexpr int S = NP / 2;
    constexpr int JE = S & 1, JO = (S + 1) & 1;
    constexpr int CE = (JE - S) / 2, CO = (1 + JO - S) / 2;
    constexpr int CMIN = CE < CO ? CE : CO;
    constexpr int TE = EPT * LT;                         // errors per tile
    constexpr int EW = PROP ? TE + NP / 2 + 1 : TE;      // errors a tile reads: with PROP those its propagation needs, from r0 + CMIN on
    constexpr int E0 = PROP ? -CMIN : 0;                 // position of error r0 among them
    constexpr int AW = 2 * TE + NP;                      // activations a tile reads
    constexpr int NA = (AW + LT - 1) / LT, NE = (EW + LT - 1) / LT;
    constexpr int WN = NP + 2 * (EPT - 1);               // activations under a thread's windows
    constexpr int VN = EPT + NP / 2;                     // errors under a thread's propagation
    __shared__ __attribute__((aligned(16))) double tile[AW + 2];
    __shared__ __attribute__((aligned(16))) double tl[EW + 3];
    __shared__ __attribute__((aligned(16))) double th[EW + 3];
    __shared__ double red[(LT / 64) * 2 * NP];
    double a[NP], b[NP];
    load_taps<NP, taps_from_a(NP)>(filt, a, b);
    double gL[NP], gH[NP];
#pragma unroll
    for (int u = 0; u < NP; u++) { gL[u] = 0; gH[u] = 0; }
    const int tid = threadIdx.x;
    const int W = 1 << (g.lgNL + lgLen), rmask = (1 << lgLen) - 1, amask = (2 << lgLen) - 1;
    double fa[NA], fl[NE], fh[NE];
    auto fetch = [&](int w0) {
        const int line = w0 >> lgLen, r0 = w0 & rmask;
        const size_t off = (size_t)blockIdx.y * xs + (size_t)line * g.ls;
        const int first = 2 * r0 + NP / 2 - (NP - 1);
#pragma unroll
        for (int q = 0; q < NA; q++) {
            const int t = tid + q * LT;
            fa[q] = (t < AW) ? act[off + ((first + t) & amask)] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < NE; q++) {
            const int t = tid + q * LT;
            const bool on = t < EW;
            fl[q] = on ? lp[off + ((r0 - E0 + t) & rmask)] : 0.0;
            fh[q] = on ? hp[off + ((r0 - E0 + t) & rmask)] : 0.0;
        }
    };
    int w0 = blockIdx.x * TE;
    if (w0 < W) fetch(w0);
    for (; w0 < W; w0 += gridDim.x * TE) {
        const int line = w0 >> lgLen, r0 = w0 & rmask;
        const size_t off = (size_t)blockIdx.y * xs + (size_t)line * g.ls;
        rk::lds_barrier();       // the previous tile has been read (a barrier for LDS only: the store of its results may still be on its way)
#pragma unroll
        for (int q = 0; q < NA; q++) {
            const int t = tid + q * LT;
            if (t < AW) tile[t] = fa[q];
        }
#pragma unroll
        for (int q = 0; q < NE; q++) {
            const int t = tid + q * LT;
            if (t < EW) { tl[t] = fl[q]; th[t] = fh[q]; }
        }
        rk::lds_barrier();
        const int wnext = w0 + gridDim.x * TE;
        if (wnext < W) fetch(wnext);
        // the activations under the windows of errors EPT tid .. EPT tid + EPT - 1: win[t] = tile[2 EPT tid + t]
        double win[WN];
        {
            const double2* p = reinterpret_cast<const double2*>(tile + 2 * EPT * tid);
#pragma unroll
            for (int u = 0; u < WN / 2; u++) { const double2 t = p[u]; win[2 * u] = t.x; win[2 * u + 1] = t.y; }
        }
#pragma unroll
        for (int k = 0; k < EPT; k++) {
            const double dl = tl[E0 + EPT * tid + k], dh = th[E0 + EPT * tid + k];
#pragma unroll
            for (int u = 0; u < NP; u++) {
                gL[u] = __builtin_fma(dl, win[2 * k + NP - 1 - u], gL[u]);
                gH[u] = __builtin_fma(dh, win[2 * k + NP - 1 - u], gH[u]);
            }
        }
        if constexpr (PROP) {
            // vl[t] = tl[EPT tid + t]: the errors from r0 + EPT tid + CMIN on
            double vl[VN], vh[VN];
            if constexpr (EPT == 1) {
#pragma unroll
                for (int t = 0; t < VN; t++) { vl[t] = tl[tid + t]; vh[t] = th[tid + t]; }
            } else {
                const double2* pl2 = reinterpret_cast<const double2*>(tl + EPT * tid);
                const double2* ph2 = reinterpret_cast<const double2*>(th + EPT * tid);
#pragma unroll
                for (int t = 0; t < VN / 2; t++) {
                    const double2 x = pl2[t], y = ph2[t];
                    vl[2 * t] = x.x; vl[2 * t + 1] = x.y;
                    vh[2 * t] = y.x; vh[2 * t + 1] = y.y;
                }
            }
#pragma unroll
            for (int k = 0; k < EPT; k++) {
                double se = 0, so = 0;
#pragma unroll
                for (int i = 0; i < NP / 2; i++) {
                    se = __builtin_fma(a[JE + 2 * i], vl[k + i + CE - CMIN], se);
                    se = __builtin_fma(b[JE + 2 * i], vh[k + i + CE - CMIN], se);
                    so = __builtin_fma(a[JO + 2 * i], vl[k + i + CO - CMIN], so);
                    so = __builtin_fma(b[JO + 2 * i], vh[k + i + CO - CMIN], so);
                }
                *reinterpret_cast<double2*>(out + off + 2 * (r0 + EPT * tid + k)) = make_double2(se, so);
            }
        }
    }
    const double s = block_window_sums<NP, LT>(gL, gH, red, tid);
    if (tid < 2 * NP) partial[((size_t)blockIdx.y * pgx + blockIdx.x) * 2 * NP + tid] += s;
}

// ---- the levels below 4096 samples of a contiguous line in one launch ---------------------------------------
// One workgroup runs the rest of a line's cascade (fwd_pass / bwd_pass of wn_device.h, the code of the
// LDS-resident kernels, here on global memory: 4096 samples and their pyramid stay in the caches) instead of
// one launch per level.  TAIL_LG: the low levels of lines longer than 2^TAIL_LG samples, and up to 8 taps whole
// lines of up to that length (measured: at 16 and 32 taps the LDS-staged level kernels are faster for those).
constexpr int TAIL_LG = 12, TAIL_T = 256;

// in: the level's input (2^lgLen samples per line); P / out as in the level kernels.  grid (lines, examples).
template <int NP>
__global__ __launch_bounds__(TAIL_T) void kb_fwd_tail(const double* __restrict__ in, double* P, double* out, size_t xs, Geom g,
                                                      int lgLen, const FilterDev* filt, KmaxSlots ks) {
    __shared__ unsigned wmax[TAIL_T / 64];
    double a[NP], b[NP];
    load_taps<NP, taps_from_a(NP)>(filt, a, b);
    const size_t off = (size_t)blockIdx.y * xs + (size_t)blockIdx.x * g.ls;
    fwd_pass<NP, TAIL_T>(in + off, P + off, out + off, 1, 0, 0, lgLen, false, a, b, (int)threadIdx.x);
    if (ks.parts) {
        // the coefficients this workgroup has just written (2^lgLen of them, the lowest low-pass value among them)
        __syncthreads();
        unsigned m = 0;
        for (int i = threadIdx.x; i < (1 << lgLen); i += TAIL_T) m = max(m, exponent_word(out[off + i]));
        leave_kmax<TAIL_T>(m, ks, wmax);
    }
}

// D: errors on the coefficients of the lines, IN PLACE: on exit D[0 .. 2^lgLen) of every line holds the errors
// on the level's input (bwd_pass).  act: that input.  grid (workgroups per example <= GXMAX, examples); a
// workgroup walks over lines.
template <int NP>
__global__ __launch_bounds__(TAIL_T) void kb_bwd_tail(double* D, const double* act, const double* P, size_t xs, Geom g, int lgLen,
                                                      const FilterDev* filt, double* partial, int pgx) {
    __shared__ double red[(TAIL_T / 64) * 2 * NP];
    double a[NP], b[NP];
    load_taps<NP, taps_from_a(NP)>(filt, a, b);
    double gL[NP], gH[NP];
#pragma unroll
    for (int u = 0; u < NP; u++) { gL[u] = 0; gH[u] = 0; }
    const int tid = threadIdx.x;
    for (int line = blockIdx.x; line < (1 << g.lgNL); line += gridDim.x) {
        const size_t off = (size_t)blockIdx.y * xs + (size_t)line * g.ls;
        __syncthreads();
        bwd_pass<NP, TAIL_T, (1 << TAIL_LG) / TAIL_T, true>(D + off, act + off, P + off, 1, 0, 0, lgLen, false, true, a, b, gL, gH, tid);
    }
    const double s = block_window_sums<NP, TAIL_T>(gL, gH, red, tid);
    if (tid < 2 * NP) partial[((size_t)blockIdx.y * pgx + blockIdx.x) * 2 * NP + tid] += s;
}

// The same two kernels with the line in LDS (samples, pyramid and coefficients / errors: 3 * 2^lgLen doubles of
// dynamic LDS), for chunks with so few lines that a workgroup per line leaves most of the chip idle anyway:
// every level is then an LDS round trip instead of a trip to the L2.  grid (lines, examples).
constexpr int TAIL_LDS_T = 512;
template <int NP>
__global__ __launch_bounds__(TAIL_LDS_T) void kb_fwd_tail_lds(const double* __restrict__ in, double* P, double* out, size_t xs, Geom g,
                                                              int lgLen, const FilterDev* filt, KmaxSlots ks) {
    extern __shared__ __attribute__((aligned(16))) double tail_lds[];
    __shared__ unsigned wmax[TAIL_LDS_T / 64];
    const int len = 1 << lgLen, tid = threadIdx.x;
    double* sIn = tail_lds;
    double* sP = sIn + len;
    double* sOut = sP + len;
    double a[NP], b[NP];
    load_taps<NP, taps_from_a(NP)>(filt, a, b);
    const size_t off = (size_t)blockIdx.y * xs + (size_t)blockIdx.x * g.ls;
    for (int i = 2 * tid; i < len; i += 2 * TAIL_LDS_T) *reinterpret_cast<double2*>(sIn + i) = *reinterpret_cast<const double2*>(in + off + i);
    __syncthreads();
    fwd_pass<NP, TAIL_LDS_T>(sIn, sP, sOut, 1, 0, 0, lgLen, false, a, b, tid);
    unsigned m = 0;
    for (int i = 2 * tid; i < len; i += 2 * TAIL_LDS_T) {
        *reinterpret_cast<double2*>(P + off + i) = *reinterpret_cast<const double2*>(sP + i);
        const double2 c = *reinterpret_cast<const double2*>(sOut + i);
        *reinterpret_cast<double2*>(out + off + i) = c;
        m = max(m, max(exponent_word(c.x), exponent_word(c.y)));
    }
    if (ks.parts) leave_kmax<TAIL_LDS_T>(m, ks, wmax);
}

template <int NP>
__global__ __launch_bounds__(TAIL_LDS_T) void kb_bwd_tail_lds(double* D, const double* __restrict__ act, const double* __restrict__ P, size_t xs,
                                                              Geom g, int lgLen, const FilterDev* filt, double* partial, int pgx) {
    extern __shared__ __attribute__((aligned(16))) double tail_lds[];
    __shared__ double red[(TAIL_LDS_T / 64) * 2 * NP];
    const int len = 1 << lgLen, tid = threadIdx.x;
    double* sD = tail_lds;
    double* sAct = sD + len;
    double* sP = sAct + len;
    double a[NP], b[NP];
    load_taps<NP, taps_from_a(NP)>(filt, a, b);
    double gL[NP], gH[NP];
#pragma unroll
    for (int u = 0; u < NP; u++) { gL[u] = 0; gH[u] = 0; }
    const size_t off = (size_t)blockIdx.y * xs + (size_t)blockIdx.x * g.ls;
    for (int i = 2 * tid; i < len; i += 2 * TAIL_LDS_T) {
        *reinterpret_cast<double2*>(sD + i) = *reinterpret_cast<const double2*>(D + off + i);
        *reinterpret_cast<double2*>(sAct + i) = *reinterpret_cast<const double2*>(act + off + i);
        *reinterpret_cast<double2*>(sP + i) = *reinterpret_cast<const double2*>(P + off + i);
    }
    __syncthreads();
    bwd_pass<NP, TAIL_LDS_T, (1 << TAIL_LG) / TAIL_LDS_T, true>(sD, sAct, sP, 1, 0, 0, lgLen, false, true, a, b, gL, gH, tid);
    for (int i = 2 * tid; i < len; i += 2 * TAIL_LDS_T) *reinterpret_cast<double2*>(D + off + i) = *reinterpret_cast<const double2*>(sD + i);
    const double s = block_window_sums<NP, TAIL_LDS_T>(gL, gH, red, tid);
    if (tid < 2 * NP) partial[((size_t)blockIdx.y * pgx + blockIdx.x) * 2 * NP + tid] += s;
}

// Per-example gradient from the per-workgroup window sums: tap k collects gL[k] and (-1)^(N-1-k) gH[N-1-k].
// grid (examples), 256 threads: wave w takes taps w, w + 4, ...; lane l adds slots l, l + 64, ... in order and the
// lanes are combined by a butterfly -- a fixed order.
__global__ __launch_bounds__(256) void kb_grad_finish(const double* partial, int NP, const FilterDev* filt, double* grad_e, int pgx) {
    const int ex = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int kp = wave; kp < NP; kp += 4) {
        double gl = 0, gh = 0;
        for (int bx = lane; bx < pgx; bx += 64) {
            const double* p = partial + ((size_t)ex * pgx + bx) * 2 * NP;
            gl += p[kp];
            gh += p[NP + (NP - 1 - kp)];
        }
        gl = wave_sum(gl);
        gh = wave_sum(gh);
        const int k = kp - filt->d;
        if (lane == 0 && k >= 0 && k < filt->N) grad_e[(size_t)ex * filt->N + k] = filt->vL[kp] * gl + filt->sH[kp] * gh;
    }
}

// sums[0..N) += sum_e grad_e, sums[N] += sum_e S_e, sums[N+1] += number of non-finite values.  One wave per
// quantity (grid N + 2): lane l adds examples l, l + 64, ... in order, the lanes are combined by a
// butterfly -- a fixed order.
__global__ __launch_bounds__(256) void kb_batch_sums(const double* grad_e, const double* S_e, int B, int N, double* sums, int first,
                                                     const unsigned* skip) {
    __shared__ double red[4];
    const int t = blockIdx.x, tid = threadIdx.x;
    if (t > N + 1) return;
    if (skip && *skip) return;   // the chunk is going to be computed again (see train_impl): leave the sums alone
    double s = 0;
    if (t < N) {
        for (int e = tid; e < B; e += 256) s += grad_e[(size_t)e * N + t];
    } else if (t == N) {
        for (int e = tid; e < B; e += 256) s += S_e[e];
    } else {
        for (int e = tid; e < B; e += 256) {
            int bad = isfinite(S_e[e]) ? 0 : 1;
            for (int k = 0; k < N; k++) bad += isfinite(grad_e[(size_t)e * N + k]) ? 0 : 1;
            s += (double)bad;
        }
    }
    s = wave_sum(s);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) sums[t] = (first ? 0.0 : sums[t]) + (((red[0] + red[1]) + red[2]) + red[3]);   // a fixed order
}

// ---- Gini --------------------------------------------------------------------------------------------
__global__ void kb_abs_iota(const double* Y, double* keys, unsigned* idx, size_t total) {
    for (size_t i = (size_t)blockIdx.x * BT + threadIdx.x; i < total; i += (size_t)gridDim.x * BT) {
        keys[i] = __builtin_fabs(Y[i]);
        idx[i] = (unsigned)i;  // position in the chunk: example * n + index
    }
}
__global__ void kb_segment_keys(const unsigned* idx, unsigned* seg, size_t total, int lgn) {
    for (size_t i = (size_t)blockIdx.x * BT + threadIdx.x; i < total; i += (size_t)gridDim.x * BT) seg[i] = idx[i] >> lgn;
}
// part[(example * GXMAX + workgroup) * 2 + {0,1}] = partial f and partial sum |c|; pos[p] is the chunk
// position of the p-th smallest magnitude of the example.
__global__ __launch_bounds__(BT) void kb_gini_partial(const unsigned* pos, const double* Y, unsigned n, double* part) {
    __shared__ double red[2 * (BT / 64)];
    const unsigned* s = pos + (size_t)blockIdx.y * n;
    double f = 0, a = 0;
    for (unsigned p = blockIdx.x * BT + threadIdx.x; p < n; p += gridDim.x * BT) {
        const double v = __builtin_fabs(Y[s[p]]);
        f = __builtin_fma(2.0 * (double)(p + 1) - (double)n - 1.0, v, f);
        a += v;
    }
    f = wave_sum(f);
    a = wave_sum(a);
    const int tid = threadIdx.x;
    if ((tid & 63) == 0) { red[tid >> 6] = f; red[BT / 64 + (tid >> 6)] = a; }
    __syncthreads();
    if (tid < 2) {
        double t = 0;
        for (int wv = 0; wv < BT / 64; wv++) t += red[tid * (BT / 64) + wv];
        part[((size_t)blockIdx.y * GXMAX + blockIdx.x) * 2 + tid] = t;
    }
}
__global__ void kb_gini_finish(const double* part, int B, int gx, unsigned n, double* fg, double* S_e) {
    const int e = blockIdx.x * BT + threadIdx.x;
    if (e >= B) return;
    double f = 0, a = 0;
    for (int bx = 0; bx < gx; bx++) { f += part[((size_t)e * GXMAX + bx) * 2]; a += part[((size_t)e * GXMAX + bx) * 2 + 1]; }
    const double g = (double)n * a;
    fg[2 * e] = f;
    fg[2 * e + 1] = g;
    if (S_e) S_e[e] = 1.0 - f / g;
}
// delta_j = -((2 rank_j - n - 1) sign g - f n sign) / g^2, in place over the coefficients
__global__ __launch_bounds__(BT) void kb_gini_delta(const unsigned* pos, const double* Yin, double* Dout,
                                                    const double* fg, unsigned n) {
    const size_t off = (size_t)blockIdx.y * n;
    const double f = fg[2 * blockIdx.y], g = fg[2 * blockIdx.y + 1];
    const double inv_g2 = 1.0 / (g * g);
    for (unsigned p = blockIdx.x * BT + threadIdx.x; p < n; p += gridDim.x * BT) {
        const unsigned j = pos[off + p];
        const double sg = sign_ref(Yin[j]);
        const double df = (2.0 * (double)(p + 1) - (double)n - 1.0) * sg;
        Dout[j] = -(df * g - f * ((double)n * sg)) * inv_g2;
    }
}

// ---- Gini, fast path: 32-bit keys ------------------------------------------------------------------
// One radix sort of 32-bit keys instead of a 64-bit and a segment sort: key = example number in the top
// bits, then as many leading bits of the magnitude (exponent first) as fit.  Positions whose keys are
// equal form a group -- same example, magnitudes equal in those leading bits -- that the stable sort
// leaves in index order; kb_rank_home puts each group in (exact magnitude, index) order by counting.
// A group longer than GROUP_LIMIT would make the counting quadratic: kb_rank_home notices and the caller
// takes the 64-bit path.
constexpr unsigned GROUP_LIMIT = 1024;

// Exact zeros are left out of the sort: they rank first, in index order, need no rank of their own (the
// sign of a zero is zero) and add nothing to either sum; an example with z zeros has the ranks of its other
// coefficients start at z + 1.  Sparse inputs leave many coefficients exactly zero (needle inputs almost
// all of them, the 512x512 calorimeter images 15 % with a 4-tap filter), and everything from the sort to
// the scatter shrinks with them.  The compaction keeps index order (the stable sort's tie order):
// workgroups of CP_BLOCK elements count their non-zeros, one workgroup scans the counts, and each
// workgroup then writes its keys behind its offset.
//
// Key: example number above `magbits` bits of magnitude -- 7 bits of exponent, clamped to [2^-95, 2^33),
// and the leading mantissa bits that fit (20 for slices of 32 examples); magnitudes below the range share
// the lowest key, those above it the highest (the map stays monotone; kb_rank_home orders equal keys
// exactly).  Measured: 12 mantissa bits (24-bit keys, three radix passes instead of four) cost more in
// the re-ranking of equal keys than the pass saves -- smooth 2^20-sample signals put dozens of coefficients on each key.
constexpr int KEY_E0 = 1023 - 95;
constexpr int CP_ITEMS = 8, CP_BLOCK = BT * CP_ITEMS;
static_assert(CP_BLOCK == 2048, "kc_fwd_cols (wn_lines.h) counts non-zeros per 2048 coefficients");

__global__ __launch_bounds__(BT) void kb_count_nonzero(const double* Y, unsigned* counts) {
    __shared__ unsigned wsum[BT / 64];
    const size_t base = (size_t)blockIdx.x * CP_BLOCK;
    unsigned c = 0;
#pragma unroll
    for (int k = 0; k < CP_ITEMS; k++) c += (Y[base + (size_t)k * BT + threadIdx.x] != 0.0) ? 1u : 0u;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned t = 0;
        for (int w = 0; w < BT / 64; w++) t += wsum[w];
        counts[blockIdx.x] = t;
    }
}
// counts[b] -> number of non-zeros before workgroup b (in place); segs[2e], segs[2e+1] = first sorted
// position and number of non-zeros of example e (bpe workgroups each); *total = all of them.  One workgroup
// per slice of `slice` examples (the last one may be shorter); the arguments describe the whole chunk.
__global__ __launch_bounds__(BT) void kb_scan_counts(unsigned* counts, int bpe, int B, int slice, unsigned* segs, unsigned* total) {
    __shared__ unsigned part[BT];
    const int tid = threadIdx.x;
    const int e0 = blockIdx.x * slice, Bs = min(slice, B - e0), nblocks = Bs * bpe;
    counts += (size_t)e0 * bpe;
    segs += 2 * (size_t)e0;
    total += blockIdx.x;
    const int per = (nblocks + BT - 1) / BT;
    const int b0 = tid * per, b1 = min(b0 + per, nblocks);
    unsigned sum = 0;
    for (int b = b0; b < b1; b++) sum += counts[b];
    part[tid] = sum;
    __syncthreads();
    if (tid == 0) {
        unsigned run = 0;
        for (int t = 0; t < BT; t++) { const unsigned c = part[t]; part[t] = run; run += c; }
        *total = run;
    }
    __syncthreads();
    unsigned run = part[tid];
    for (int b = b0; b < b1; b++) { const unsigned c = counts[b]; counts[b] = run; run += c; }
    __syncthreads();
    __threadfence_block();
    for (int e = tid; e < Bs; e += BT) {
        const unsigned first = counts[e * bpe];
        const unsigned next = (e + 1 < Bs) ? counts[(e + 1) * bpe] : *total;
        segs[2 * e] = first;
        segs[2 * e + 1] = next - first;
    }
}
// key / idx of the non-zero coefficients, in index order, behind the workgroup's offset
__device__ __forceinline__ unsigned magnitude_key(unsigned long long bits, int magbits) {
    const int e = (int)(bits >> 52) - KEY_E0;
    const int mb = magbits - 7;
    if (e < 0) return 0u;
    if (e > 127) return 0xFFFFFFFFu >> (32 - magbits);
    return ((unsigned)e << mb) | ((unsigned)(bits >> (52 - mb)) & ((1u << mb) - 1u));
}
__global__ __launch_bounds__(BT) void kb_key32_compact(const double* Y, const unsigned* starts, unsigned* key, unsigned* idx,
                                                       int lgn, int magbits) {
    __shared__ unsigned wtot[CP_ITEMS * (BT / 64)];
    const size_t base = (size_t)blockIdx.x * CP_BLOCK;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned long long bits[CP_ITEMS];
    unsigned before[CP_ITEMS];
    unsigned nzmask = 0;
#pragma unroll
    for (int k = 0; k < CP_ITEMS; k++) {
        const double v = Y[base + (size_t)k * BT + tid];
        bits[k] = (unsigned long long)__double_as_longlong(v) & 0x7FFFFFFFFFFFFFFFull;
        const bool nz = (v != 0.0);
        const unsigned long long ballot = __ballot(nz);
        before[k] = (unsigned)__popcll(ballot & ((1ull << lane) - 1ull));
        if (lane == 0) wtot[k * (BT / 64) + wave] = (unsigned)__popcll(ballot);
        nzmask |= nz ? (1u << k) : 0u;
    }
    __syncthreads();
    unsigned run = starts[blockIdx.x];
#pragma unroll
    for (int k = 0; k < CP_ITEMS; k++) {
        unsigned mine = 0;
#pragma unroll
        for (int w = 0; w < BT / 64; w++) {
            if (w == wave) mine = run;
            run += wtot[k * (BT / 64) + w];
        }
        if (nzmask & (1u << k)) {
            const size_t i = base + (size_t)k * BT + tid;
            const unsigned p = mine + before[k];
            const unsigned ex = (unsigned)(i >> lgn);
            key[p] = (magbits < 32 ? (ex << magbits) : 0u) | magnitude_key(bits[k], magbits);
            idx[p] = (unsigned)i;
        }
    }
}
constexpr int FIX_TILE = 2048, FIX_HALO = 32;   // sorted positions staged in LDS per step, and on either side
constexpr int RANK_TILE = 1024, RANK_GX = 256;  // kb_rank_home: smaller steps, one example per XCD at a time

// ---- ranks written home -----------------------------------------------------------------------------------
// The sorted order is only read, never materialised as values: kb_rank_home walks the sorted (key, position)
// pairs of an example and writes each coefficient's rank to its home position (one 4-byte scatter).  A key
// that differs from both neighbours fixes the rank by itself; the members of a group of equal keys fetch
// their magnitudes (the only gathers) and are ordered exactly by (magnitude, index).  A group longer than
// GROUP_LIMIT sets *flag instead (the caller then takes the 64-bit path).  The sums and the errors are then
// plain streaming passes over the coefficients in home order (kb_sums_home, kb_delta_home); the rank stored
// at an exact zero is never written and never matters (it is multiplied by |c| = 0 or sign(0) = 0).
// 1D grid of gx * (examples rounded up to 8) workgroups: consecutive workgroups go to consecutive XCDs, so
// workgroup L works on example (L / 8 / gx) * 8 + L % 8 -- all gx workgroups of an example share one XCD,
// whose L2 then collects the example's scattered ranks (1 MB at 512x512) into whole lines.
__global__ __launch_bounds__(BT) void kb_rank_home(const unsigned* key, const unsigned* idx, const double* Y, unsigned n,
                                                   const unsigned* segs, unsigned* rank, unsigned* flag, int gx, int B) {
    __shared__ double la[RANK_TILE + 2 * FIX_HALO];
    __shared__ unsigned lk[RANK_TILE + 2 * FIX_HALO], li[RANK_TILE + 2 * FIX_HALO];
    const int slot = blockIdx.x >> 3;
    const int ex = (slot / gx) * 8 + (blockIdx.x & 7), bx = slot % gx;
    if (ex >= B) return;
    const size_t exstart = segs[2 * ex], exend = exstart + segs[2 * ex + 1];
    const unsigned zeros = n - segs[2 * ex + 1];
    const int tid = threadIdx.x;
    bool over = false;
    for (size_t t0 = exstart + (size_t)bx * RANK_TILE; t0 < exend; t0 += (size_t)gx * RANK_TILE) {
        const size_t tile = (exend - t0 < (size_t)RANK_TILE) ? exend - t0 : (size_t)RANK_TILE;
        const size_t lo = (t0 >= exstart + FIX_HALO) ? t0 - FIX_HALO : exstart;
        const size_t hi = (t0 + tile + FIX_HALO <= exend) ? t0 + tile + FIX_HALO : exend;
        const int cnt = (int)(hi - lo);
        __syncthreads();
        for (int l = tid; l < cnt; l += BT) {
            lk[l] = key[lo + l];
            li[l] = idx[lo + l];
        }
        __syncthreads();
        for (int l = tid; l < cnt; l += BT) {   // magnitudes of group members only
            const unsigned k = lk[l];
            if ((l > 0 && lk[l - 1] == k) || (l + 1 < cnt && lk[l + 1] == k)) la[l] = __builtin_fabs(Y[li[l]]);
        }
        __syncthreads();
        for (size_t p = t0 + tid; p < t0 + tile; p += BT) {
            const int l = (int)(p - lo);
            const unsigned k = lk[l];
            const unsigned me = li[l];
            unsigned rk = zeros + (unsigned)(p - exstart) + 1u;
            if ((l > 0 && lk[l - 1] == k) || (l + 1 < cnt && lk[l + 1] == k)) {
                const double av = la[l];
                int gs = l;
                while (gs > 0 && lk[gs - 1] == k) gs--;
                int ge = l + 1;
                while (ge < cnt && lk[ge] == k) ge++;
                unsigned below = 0;
                if ((gs > 0 || lo == exstart) && (ge < cnt || hi == exend)) {
                    if ((unsigned)(ge - gs) > GROUP_LIMIT) {
                        over = true;
                    } else {
                        for (int q = gs; q < ge; q++) below += (la[q] < av || (la[q] == av && li[q] < me)) ? 1u : 0u;
                    }
                    rk = zeros + (unsigned)(lo + gs - exstart) + below + 1u;
                } else {  // the group reaches past the staged range
                    size_t g0 = p;
                    while (g0 > exstart && key[g0 - 1] == k && p - g0 <= GROUP_LIMIT) g0--;
                    size_t q = g0;
                    for (; q < exend && key[q] == k && q - g0 <= GROUP_LIMIT; q++) {
                        const double aq = __builtin_fabs(Y[idx[q]]);
                        below += (aq < av || (aq == av && idx[q] < me)) ? 1u : 0u;
                    }
                    if (q - g0 > GROUP_LIMIT) over = true;
                    rk = zeros + (unsigned)(g0 - exstart) + below + 1u;
                }
            }
            rank[me] = rk;
        }
    }
    if (over) *flag = 1u;
}
// part[(example * GXMAX + workgroup) * 2 + {0,1}] = partial f, partial sum |c|, over home positions
__global__ __launch_bounds__(BT) void kb_sums_home(const double* Y, const unsigned* rank, unsigned n, double* part) {
    __shared__ double red[2 * (BT / 64)];
    const size_t off = (size_t)blockIdx.y * n;
    const unsigned span = n / gridDim.x;                  // gridDim.x divides n
    const unsigned j0 = blockIdx.x * span;
    double f = 0, a = 0;
    for (unsigned j = j0 + threadIdx.x; j < j0 + span; j += BT) {
        const double av = __builtin_fabs(Y[off + j]);
        if (av != 0.0) {                                   // also true for NaN
            f = __builtin_fma(2.0 * (double)rank[off + j] - (double)n - 1.0, av, f);
            a += av;
        }
    }
    f = wave_sum(f);
    a = wave_sum(a);
    const int tid = threadIdx.x;
    if ((tid & 63) == 0) { red[tid >> 6] = f; red[BT / 64 + (tid >> 6)] = a; }
    __syncthreads();
    if (tid < 2) {
        double t = 0;
        for (int wv = 0; wv < BT / 64; wv++) t += red[tid * (BT / 64) + wv];
        part[((size_t)blockIdx.y * GXMAX + blockIdx.x) * 2 + tid] = t;
    }
}
// delta_j = -((2 rank_j - n - 1) sign g - f n sign) / g^2 at home positions; Dout may be Yin
__global__ __launch_bounds__(BT) void kb_delta_home(const double* Yin, const unsigned* rank, double* Dout, const double* fg,
                                                    unsigned n) {
    const size_t off = (size_t)blockIdx.y * n;
    const double f = fg[2 * blockIdx.y], g = fg[2 * blockIdx.y + 1];
    const double inv_g2 = 1.0 / (g * g);
    for (unsigned j = blockIdx.x * BT + threadIdx.x; j < n; j += gridDim.x * BT) {
        const double y = Yin[off + j];
        const double sg = sign_ref(y);
        const double coef = (y != 0.0) ? 2.0 * (double)rank[off + j] - (double)n - 1.0 : 0.0;
        const double df = coef * sg;
        Dout[off + j] = -(df * g - f * ((double)n * sg)) * inv_g2;
    }
}

// ---- workspace ----------------------------------------------------------------------------------------
struct Carver {
    char* base;
    size_t used = 0;
    template <class T>
    T* take(size_t count) {
        used = (used + 255) & ~(size_t)255;
        T* p = base ? reinterpret_cast<T*>(base + used) : nullptr;
        used += count * sizeof(T);
        return p;
    }
};

int ensure_workspace(wn_context* ctx, size_t bytes) {
    if (ctx->workspace_bytes >= bytes) return WN_OK;
    WN_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->workspace) (void)hipFree(ctx->workspace);
    ctx->workspace = nullptr;
    ctx->workspace_bytes = 0;
    WN_HIP(ctx, hipMalloc(&ctx->workspace, bytes));
    ctx->workspace_bytes = bytes;
    return WN_OK;
}

inline dim3 grid_for(int items, int B, int cap = 4096) {
    int gx = (items + BT - 1) / BT;
    if (gx > cap) gx = cap;
    if (gx < 1) gx = 1;
    return dim3((unsigned)gx, (unsigned)B);
}

// ---- passes ----------------------------------------------------------------------------------------------
// the tail of the lines in LDS (kb_*_tail_lds) rather than through the L2: when the chunk has fewer lines than the chip
// has CUs, and the tail is long enough for the levels' round trips to matter
bool tail_in_lds(const wn_context* ctx, const Geom& g, int B, int lgTail) {
    return ((long long)B << g.lgNL) <= ctx->numCU && lgTail >= 8 && lgTail <= TAIL_LG;
}
// The LDS-staged and tail kernels write the high-pass taps from the low-pass ones where taps_from_a says so, which
// holds for an even padding in front of the filter (load_taps); a filter with an odd one goes level by level.
template <int NP>
bool staged_kernels_fit(const wn_context* ctx) {
    return !taps_from_a(NP) || (((NP - ctx->N) / 2) & 1) == 0;
}

// kparts (may be null): kstride words per example in which the workgroups leave the largest exponent word of the
// coefficients they wrote (KmaxSlots) when all the kernels are of the kind that can (long contiguous lines); *nparts
// receives the number of slots written per example, 0 when they were not.
template <int NP>
int fwd_pass_big(wn_context* ctx, const double* in, double* P, double* out, int B, size_t n, Geom g, unsigned* kparts = nullptr,
                 int kstride = 0, int* nparts = nullptr) {
    hipStream_t st = ctx->stream;
    const bool staged = staged_kernels_fit<NP>(ctx);
    // every level by an LDS-staged kernel and the rest by a tail kernel?
    if (!(staged && NP >= LDS_LEVEL_TAPS && g.es == 1 && !g.lanesOnLines && ctx->wave64 && g.lgLen > TAIL_LG)) kparts = nullptr;
    // a launch over level i has at most lines * 2^i / LT workgroups, the tail one per line: fewer than
    // lines * 2^lgLen / LT + lines slots in all; a caller that left less room gets none written
    if (kparts && (((size_t)1 << (g.lgNL + g.lgLen)) / LT + ((size_t)1 << g.lgNL)) > (size_t)kstride) kparts = nullptr;
    KmaxSlots ks{kparts, kstride, 0};
    if (nparts) *nparts = 0;
    if (g.lgLen == 0) {
        WN_HIP(ctx, hipMemcpyAsync(out, in, (size_t)B * n * sizeof(double), hipMemcpyDeviceToDevice, st));
        WN_HIP(ctx, hipMemsetAsync(P, 0, (size_t)B * n * sizeof(double), st));
        return WN_OK;
    }
    // contiguous lines: the levels below 2^TAIL_LG samples in one launch
    const int tail = (staged && g.es == 1 && !g.lanesOnLines && ctx->wave64 && (g.lgLen > TAIL_LG || NP <= 8)) ? std::min(g.lgLen, TAIL_LG) : 0;
    for (int i = g.lgLen - 1; i >= tail; i--) {
        const double* src = (i == g.lgLen - 1) ? in : P + (size_t)(2 << i) * g.es;
        // long contiguous lines: three (or two) levels per launch while that many are left above the tail
        if (staged && NP >= LDS_LEVEL_TAPS && g.es == 1 && !g.lanesOnLines && i - 1 >= tail && (1 << (i - 1)) >= 16 * LT) {
            const int nlev = (i - 2 >= tail && (1 << (i - 2)) >= 16 * LT) ? 3 : 2;
            const dim3 grid((1u << (g.lgNL + i - (nlev - 1))) / LT, B);
            if (nlev == 3) {
                const size_t lds = FwdFused<NP, 3>::lds_doubles() * sizeof(double);
                hipLaunchKernelGGL((kb_fwd_levels_lds<NP, 3>), grid, dim3(LT), lds, st, src, P, out, n, g, i, ctx->dfilt, ks);
            } else {
                const size_t lds = FwdFused<NP, 2>::lds_doubles() * sizeof(double);
                hipLaunchKernelGGL((kb_fwd_levels_lds<NP, 2>), grid, dim3(LT), lds, st, src, P, out, n, g, i, ctx->dfilt, ks);
            }
            ks.slot0 += (int)grid.x;
            i -= nlev - 1;
            continue;
        }
        if (staged && NP >= LDS_LEVEL_TAPS && g.es == 1 && !g.lanesOnLines && (1 << i) >= LT) {
            hipLaunchKernelGGL(kb_fwd_level_lds<NP>, dim3((1u << (g.lgNL + i)) / LT, B), dim3(LT), 0, st, src,
                               P + (size_t)(1 << i), out + (size_t)(1 << i), n, g, i, ctx->dfilt, ks);
            ks.slot0 += (int)((1u << (g.lgNL + i)) / LT);
            continue;
        }
        hipLaunchKernelGGL(kb_fwd_level<NP>, grid_for(1 << (g.lgNL + i), B), dim3(BT), 0, st, src,
                           P + (size_t)(1 << i) * g.es, out + (size_t)(1 << i) * g.es, n, g, i, ctx->dfilt);
    }
    if (tail) {
        const double* src = (tail == g.lgLen) ? in : P + ((size_t)1 << tail);
        if (tail_in_lds(ctx, g, B, tail)) {
            const size_t lds = (size_t)3 * sizeof(double) << tail;
            WN_HIP(ctx, hipFuncSetAttribute((const void*)kb_fwd_tail_lds<NP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(kb_fwd_tail_lds<NP>, dim3(1u << g.lgNL, B), dim3(TAIL_LDS_T), lds, st, src, P, out, n, g, tail, ctx->dfilt, ks);
        } else {
            hipLaunchKernelGGL(kb_fwd_tail<NP>, dim3(1u << g.lgNL, B), dim3(TAIL_T), 0, st, src, P, out, n, g, tail, ctx->dfilt, ks);
        }
        ks.slot0 += 1 << g.lgNL;
    } else {
        hipLaunchKernelGGL(kb_finish_pass, grid_for(1 << g.lgNL, B), dim3(BT), 0, st, out, P, n, g);
    }
    WN_HIP(ctx, hipGetLastError());
    if (ks.parts && nparts) *nparts = ks.slot0;
    return WN_OK;
}

// Backward over the lines of one pass.  coef: errors on the coefficients of these lines; in/P: the
// activations; pingA/pingB: two scratch arrays (distinct from coef).  *result points at the array
// holding the errors on the input afterwards.  partial may be null (inverse transform: no gradient).
template <int NP>
int bwd_pass_big(wn_context* ctx, const double* coef, const double* in, const double* P, double* pingA,
                 double* pingB, int B, size_t n, Geom g, bool lastPropagate, double* partial, int pgx, const double** result,
                 double* coefScratch = nullptr) {
    hipStream_t st = ctx->stream;
    const double* lp = coef;
    double* dst = pingA;
    // coefScratch: coef again, writable (the caller does not need the errors afterwards).  Contiguous lines
    // then have their levels below 2^TAIL_LG samples done in place by one launch; when that is the whole
    // pass, its result -- the errors on the lines -- is in coef.
    int first = 0;
    const bool staged = staged_kernels_fit<NP>(ctx);
    if (staged && coefScratch && partial && g.es == 1 && !g.lanesOnLines && ctx->wave64 && (g.lgLen > TAIL_LG || (NP <= 8 && lastPropagate))) {
        first = std::min(g.lgLen, TAIL_LG);
        const double* act = (first == g.lgLen) ? in : P + ((size_t)1 << first);
        if (tail_in_lds(ctx, g, B, first) && (1 << g.lgNL) <= pgx) {
            const size_t lds = (size_t)3 * sizeof(double) << first;
            WN_HIP(ctx, hipFuncSetAttribute((const void*)kb_bwd_tail_lds<NP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(kb_bwd_tail_lds<NP>, dim3(1u << g.lgNL, B), dim3(TAIL_LDS_T), lds, st, coefScratch, act, P, n, g, first,
                               ctx->dfilt, partial, pgx);
        } else {
            hipLaunchKernelGGL(kb_bwd_tail<NP>, dim3(std::min(1 << g.lgNL, pgx), B), dim3(TAIL_T), 0, st, coefScratch, act, P, n, g, first,
                               ctx->dfilt, partial, pgx);
        }
    }
    for (int i = first; i < g.lgLen; i++) {
        const double* hp = coef + (size_t)(1 << i) * g.es;
        const bool propagate = (i != g.lgLen - 1 || lastPropagate);
        if (staged && partial && NP >= LDS_LEVEL_TAPS && g.es == 1 && !g.lanesOnLines && (1 << i) >= LT) {
            const double* act = (i == g.lgLen - 1) ? in : P + (size_t)(2 << i);
            const bool wide = BWD_LDS_EPT > 1 && (1 << i) >= BWD_LDS_EPT * LT;   // (a line's errors fill the wider tile)
            const int te = wide ? BWD_LDS_EPT * LT : LT;
            // workgroups per example: the kernel runs two to a CU (registers), and every one of them ends with the
            // reduction of its window sums -- no more of them than fill the chip
            const int fill = std::max(1, (BWD_LDS_FILL * ctx->numCU + B - 1) / B);
            const dim3 grid((unsigned)std::min<size_t>(std::min<size_t>(((size_t)1 << (g.lgNL + i)) / te, pgx), fill), B);
            double* none = nullptr;
            if (propagate) {
                if (wide) hipLaunchKernelGGL((kb_bwd_level_lds<NP, true, BWD_LDS_EPT>), grid, dim3(LT), 0, st, lp, hp, act, dst, n, g, i, ctx->dfilt, partial, pgx);
                else hipLaunchKernelGGL((kb_bwd_level_lds<NP, true, 1>), grid, dim3(LT), 0, st, lp, hp, act, dst, n, g, i, ctx->dfilt, partial, pgx);
                lp = dst;
                dst = (dst == pingA) ? pingB : pingA;
            } else {
                if (wide) hipLaunchKernelGGL((kb_bwd_level_lds<NP, false, BWD_LDS_EPT>), grid, dim3(LT), 0, st, lp, hp, act, none, n, g, i, ctx->dfilt, partial, pgx);
                else hipLaunchKernelGGL((kb_bwd_level_lds<NP, false, 1>), grid, dim3(LT), 0, st, lp, hp, act, none, n, g, i, ctx->dfilt, partial, pgx);
            }
            continue;
        }
        if (partial && propagate) {
            const double* act = (i == g.lgLen - 1) ? in : P + (size_t)(2 << i) * g.es;
            hipLaunchKernelGGL(kb_bwd_level<NP>, grid_for(1 << (g.lgNL + i), B, pgx), dim3(BT), 0, st, lp, hp, act, dst, n, g, i,
                               ctx->dfilt, partial, pgx);
            lp = dst;
            dst = (dst == pingA) ? pingB : pingA;
            continue;
        }
        if (partial) {
            const double* act = (i == g.lgLen - 1) ? in : P + (size_t)(2 << i) * g.es;
            hipLaunchKernelGGL(kb_grad_level<NP>, grid_for(1 << (g.lgNL + i), B, pgx), dim3(BT), 0, st, lp, hp, act, n, g, i, partial, pgx);
        }
        if (i != g.lgLen - 1 || lastPropagate) {
            hipLaunchKernelGGL(kb_prop_level<NP>, grid_for(1 << (g.lgNL + i), B), dim3(BT), 0, st, lp, hp, dst, n, g, i, ctx->dfilt);
            lp = dst;
            dst = (dst == pingA) ? pingB : pingA;
        }
    }
    WN_HIP(ctx, hipGetLastError());
    if (result) *result = lp;
    return WN_OK;
}

// Y, D: the chunk's coefficients / errors (D may alias Y or be null).
int gini_big(wn_context* ctx, const double* Y, double* D, int B, size_t n, double* keysA, double* keysB,
             unsigned* idxA, unsigned* idxB, double* gpart, double* fg, double* S_e, void* sort_tmp,
             size_t sort_tmp_bytes) {
    hipStream_t st = ctx->stream;
    const size_t total = (size_t)B * n;
    const unsigned blocks = (unsigned)std::min<size_t>((total + BT - 1) / BT, 65536);
    int lgn = 0;
    while (((size_t)1 << lgn) < n) lgn++;
    hipLaunchKernelGGL(kb_abs_iota, dim3(blocks), dim3(BT), 0, st, Y, keysA, idxA, total);
    // by magnitude over the whole chunk (stable: equal magnitudes stay in position order) ...
    WN_HIP(ctx, rocprim::radix_sort_pairs(sort_tmp, sort_tmp_bytes, keysA, keysB, idxA, idxB, total, 0, 64, st));
    const unsigned* pos = idxB;
    if (B > 1) {
        // ... then by example number, which groups the examples and keeps the magnitude order inside each
        unsigned* segA = reinterpret_cast<unsigned*>(keysA);
        unsigned* segB = reinterpret_cast<unsigned*>(keysB);
        int bits = 1;
        while ((1 << bits) < B) bits++;
        hipLaunchKernelGGL(kb_segment_keys, dim3(blocks), dim3(BT), 0, st, idxB, segA, total, lgn);
        WN_HIP(ctx, rocprim::radix_sort_pairs(sort_tmp, sort_tmp_bytes, segA, segB, idxB, idxA, total, 0, bits, st));
        pos = idxA;
    }
    const int gx = (int)std::min<size_t>((n + BT - 1) / BT, GXMAX);
    hipLaunchKernelGGL(kb_gini_partial, dim3(gx, B), dim3(BT), 0, st, pos, Y, (unsigned)n, gpart);
    hipLaunchKernelGGL(kb_gini_finish, dim3((B + BT - 1) / BT), dim3(BT), 0, st, gpart, B, gx, (unsigned)n, fg, S_e);
    if (D) hipLaunchKernelGGL(kb_gini_delta, grid_for((int)std::min<size_t>(n, 1u << 30), B), dim3(BT), 0, st, pos, Y, D, fg, (unsigned)n);
    WN_HIP(ctx, hipGetLastError());
    return WN_OK;
}

// Histogram bins of the bucket ranking for examples of n coefficients, and log2 of the bins per octave (NF / RK_OCTAVES).
void rank_bins(size_t n, int* NF, int* lgBPO) {
    *NF = (int)std::min<size_t>(rk::RK_NFMAX, std::max<size_t>(1024, n / 2));
    *lgBPO = 0;
    while ((rk::RK_OCTAVES << *lgBPO) < *NF) (*lgBPO)++;
}

// Ranks by histogram and per-bucket LDS sort (wn_rank.h).  Leaves 32-bit ranks at home positions in keysB
// (example e at (unsigned*)keysB + e * n), {f, g} in fg, the sparsity terms in S_e, and with D the errors.
// *ok = false: some bucket would overflow (or the example is too long) -- nothing usable was written.
// kmax: B words, and 16 spare ones of which one serves as the give-up flag; kmax_ready: the producer of Y has raised
// them already (and its caller zeroed all B + 16 words before that).
// What a consumer of (home index, rank) pairs needs to find them (kc_bwd_cols_regions, wn_lines.h).  want: the caller can
// take pairs; used: pairs it is (32-bit ones in keysB, example e at word e * n; no ranks at home positions were written).
// Where the producer of the coefficients left the largest exponent word per workgroup: count words per example, stride apart.
struct KmaxParts {
    const unsigned* parts;
    int stride, count;
};
struct RankPairs {
    bool want = false, used = false;
    const unsigned* pairs = nullptr;
    const unsigned short* snap = nullptr;
    const unsigned* meta = nullptr;
    const unsigned* flag = nullptr;
    int meta_stride = 0, nbe = 0, ib = 0;
};

// hist_need (with kmax_ready): the producer has also left the histograms (in idxB, where kr_hist puts them) of all
// examples but those it marked in hist_need[e].
int gini_buckets(wn_context* ctx, const double* Y, double* D, int B, size_t n, double* keysA, double* keysB, unsigned* idxA,
                 unsigned* idxB, double* fg, double* S_e, unsigned* kmax, bool kmax_ready, bool* ok, const unsigned** pending = nullptr,
                 const unsigned* hist_need = nullptr, RankPairs* rp = nullptr, const KmaxParts* kp = nullptr) {
    using namespace rk;
    *ok = false;
    int lgn = 0;
    while (((size_t)1 << lgn) < n) lgn++;
    if (lgn > RK_LGN_MAX || n < 8192) return WN_OK;
    hipStream_t st = ctx->stream;
    int NF, lgBPO;
    rank_bins(n, &NF, &lgBPO);
    const int nbmax = nbmax_for(n);
    unsigned* hist = idxB;                                                           // B * NF words
    unsigned short* btab = reinterpret_cast<unsigned short*>(idxB + (size_t)B * NF);  // B * NF half words
    unsigned* rank = reinterpret_cast<unsigned*>(keysB);                              // B * n words
    // B * meta_words(n) words behind the bucket table (a multiple of 8 bytes in: NF is a multiple of 1024); the array of
    // ranks then has its second half free for the (home index, rank) pairs of kr_delta_regions
    unsigned* meta = reinterpret_cast<unsigned*>(btab + (size_t)B * NF);
    unsigned* flag = kmax + B + 4;   // beside the B words of kmax (the plan leaves 16 spare): nothing later in the chunk writes there
    unsigned long long* mag = reinterpret_cast<unsigned long long*>(keysA);
    // per context, i.e. per device: a function's attributes belong to the device it is loaded on
    if (!ctx->rank_configured) {
        WN_HIP(ctx, hipFuncSetAttribute((const void*)kr_hist, hipFuncAttributeMaxDynamicSharedMemorySize, RK_NFMAX * 4));
        WN_HIP(ctx, hipFuncSetAttribute((const void*)kr_scan, hipFuncAttributeMaxDynamicSharedMemorySize, RK_NFMAX * 4));
        WN_HIP(ctx, hipFuncSetAttribute((const void*)kr_part<true>, hipFuncAttributeMaxDynamicSharedMemorySize, RK_NFMAX * 2));
        WN_HIP(ctx, hipFuncSetAttribute((const void*)kr_part<false>, hipFuncAttributeMaxDynamicSharedMemorySize, RK_NFMAX * 2));
        WN_HIP(ctx, hipFuncSetAttribute((const void*)kr_part8, hipFuncAttributeMaxDynamicSharedMemorySize, RK_NFMAX * 2));
        WN_HIP(ctx, hipFuncSetAttribute((const void*)kr_count, hipFuncAttributeMaxDynamicSharedMemorySize, RK_NFMAX * 2));
        ctx->rank_configured = true;
    }
    // workgroups per example: enough of them to fill the chip when the chunk has few examples
    int G = 1;
    while ((long long)G * B < 2LL * ctx->numCU && n / (2 * (size_t)G) >= (size_t)RK_TILE) G *= 2;
    if (!kmax_ready) {
        // (the flag too: with kmax_ready the producer's caller has zeroed these B + 16 words before the producer ran)
        WN_HIP(ctx, hipMemsetAsync(kmax, 0, (size_t)(B + 16) * sizeof(unsigned), st));
        // (kp: the producer's workgroups left their maxima in slots; kr_hist takes the maximum over them and writes kmax)
        if (!kp) hipLaunchKernelGGL(kr_max, dim3(G, B), dim3(RK_PT), 0, st, Y, n, kmax);
    }
    const unsigned* kparts = (kp && !kmax_ready) ? kp->parts : nullptr;
    const int kstride = kp ? kp->stride : 0, kcount = kp ? kp->count : 0;
    if (kmax_ready && hist_need) {
        // the few examples the producer could not do: one workgroup each, plain stores
        hipLaunchKernelGGL(kr_hist, dim3(1, B), dim3(RK_PT), (size_t)NF * 4, st, Y, n, kmax, hist, NF, lgBPO, hist_need, (const unsigned*)nullptr, 0, 0);
    } else {
        if (G > 1) WN_HIP(ctx, hipMemsetAsync(hist, 0, (size_t)B * NF * sizeof(unsigned), st));
        hipLaunchKernelGGL(kr_hist, dim3(G, B), dim3(RK_PT), (size_t)NF * 4, st, Y, n, kmax, hist, NF, lgBPO, (const unsigned*)nullptr, kparts, kstride, kcount);
    }
    hipLaunchKernelGGL(kr_scan, dim3(B), dim3(RK_BT), (size_t)NF * 4, st, hist, kmax, NF, lgBPO, n, btab, meta, flag);
    const int sub = ctx->rank_sub > 0 ? std::min(ctx->rank_sub, B) : B;
    const bool packed = ctx->rank_packed;
    // errors wanted and an example's ranks beyond what an L2 merges into whole lines: pairs in slot order, taken home
    // region by region (one launch over the whole chunk, so not with sub-chunks)
    bool by_regions = D && packed && sub == B && lgn >= ctx->rank_regions_from;
    unsigned long long* pairs = reinterpret_cast<unsigned long long*>(keysB);
    int ib_all = 0;
    // no array of errors wanted, and the consumer of the ranks takes pairs: the same, with regions small enough for its LDS
    // (32-bit pairs: the rank, up to n, beside RK_IB_MIN bits of home index)
    const bool pairs_out = !D && rp && rp->want && packed && sub == B && lgn + 1 + RK_IB_MIN <= 32;
    const int steps_max = (int)(n >> RK_IB_MIN), nbe = nbmax_even(n);    // regions per example at most: what the snapshots are sized for
    // packed staging: cursor snapshots and (several workgroups per example) kr_count's counts in place of the index array
    unsigned short* snap = reinterpret_cast<unsigned short*>(idxA);
    unsigned* precnt = idxA + ((size_t)B * steps_max * nbe + 1) / 2;
    for (int s0 = 0; s0 < B; s0 += sub) {
        const int Bs = std::min(sub, B - s0);
        int Gs = 1;
        while ((long long)Gs * Bs < 2LL * ctx->numCU && n / (2 * (size_t)Gs) >= (size_t)std::max(RK_TILE, 1 << RK_IB_MIN)) Gs *= 2;   // (whole regions per workgroup)
        int ib = rank_ib(lgn, n / (size_t)Gs);
        if (by_regions) ib = std::min(ib, ctx->rank_regions_ib);    // regions of 64 KB of ranks: two workgroups of kr_delta_regions per CU
        if (pairs_out) ib = RK_IB_MIN;                              // regions of 32 KB of ranks
        const int steps = (int)(n >> ib);
        ib_all = ib;
        if (((size_t)4 << ib) > (size_t)128 * 1024) by_regions = false;     // a region's ranks have to fit the LDS
        const double* Ys = Y + (size_t)s0 * n;
        const unsigned short* bts = btab + (size_t)s0 * NF;
        unsigned* ms = meta + (size_t)s0 * meta_words(n);
        unsigned* rks = rank + (size_t)s0 * n;
        const unsigned bgrid = (unsigned)nbmax * (((unsigned)Bs + 7u) & ~7u);
        if (packed) {
            unsigned short* sn = snap + (size_t)s0 * steps * nbe;
            const unsigned* pc = nullptr;
            if (Gs > 1) {
                hipLaunchKernelGGL(kr_count, dim3(Gs, Bs), dim3(RK_PT), (size_t)NF * 2, st, Ys, n, bts, ms, NF, lgBPO, precnt, flag);
                pc = precnt;
            }
            hipLaunchKernelGGL(kr_part8, dim3(Gs, Bs), dim3(RK_PT), (size_t)NF * 2, st, Ys, n, bts, ms, NF, lgBPO, mag, sn, pc, flag, ib);
            hipLaunchKernelGGL((kr_bucket<RK_LGNSB, true>), dim3(bgrid), dim3(RK_KT), 0, st, mag, (const unsigned*)nullptr, sn, Ys, n, lgn, ms, rks, Bs, flag, NF, lgBPO, ib,
                               (by_regions || pairs_out) ? pairs : (unsigned long long*)nullptr, pairs_out ? ib : 0);
        } else {
            if (Gs == 1) hipLaunchKernelGGL(kr_part<true>, dim3(1, Bs), dim3(RK_PT), (size_t)NF * 2, st, Ys, n, bts, ms, NF, lgBPO, mag, idxA, flag);
            else hipLaunchKernelGGL(kr_part<false>, dim3(Gs, Bs), dim3(RK_PT), (size_t)NF * 2, st, Ys, n, bts, ms, NF, lgBPO, mag, idxA, flag);
            hipLaunchKernelGGL((kr_bucket<RK_LGNSB, false>), dim3(bgrid), dim3(RK_KT), 0, st, mag, idxA, (const unsigned short*)nullptr, Ys, n, lgn, ms, rks, Bs, flag, NF, lgBPO, 0,
                               (unsigned long long*)nullptr, 0);
        }
    }
    hipLaunchKernelGGL(kr_finish, dim3(B), dim3(64), 0, st, meta, n, B, fg, S_e);
    WN_HIP(ctx, hipGetLastError());
    if (pairs_out) {
        rp->used = true;
        rp->pairs = reinterpret_cast<const unsigned*>(pairs);
        rp->snap = snap;
        rp->meta = meta;
        rp->flag = flag;
        rp->meta_stride = (int)meta_words(n);
        rp->nbe = nbe;
        rp->ib = ib_all;
    }
    if (pending) {
        // the caller looks at the flag when it has queued everything that follows, and starts over if it is up; what the
        // kernels in between compute from unfinished ranks is arithmetic on garbage, never an address.  The event marks
        // the point from which the flag is final.
        if (!ctx->rank_event) WN_HIP(ctx, hipEventCreateWithFlags(&ctx->rank_event, hipEventDisableTiming));
        WN_HIP(ctx, hipEventRecord(ctx->rank_event, st));
        *pending = flag;
    } else {
        unsigned over = 0;
        WN_HIP(ctx, hipMemcpyAsync(&over, flag, sizeof(unsigned), hipMemcpyDeviceToHost, st));
        WN_HIP(ctx, hipStreamSynchronize(st));
        if (over) return WN_OK;
    }
    if (D && by_regions) {
        if (!ctx->regions_configured) {
            WN_HIP(ctx, hipFuncSetAttribute((const void*)kr_delta_regions, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
            ctx->regions_configured = true;
        }
        hipLaunchKernelGGL(kr_delta_regions, dim3((unsigned)(n >> ib_all), B), dim3(RK_KT), (size_t)4 << ib_all, st, Y, pairs, snap, meta, n,
                           ib_all, fg, D, flag);
        WN_HIP(ctx, hipGetLastError());
    } else if (D) {
        hipLaunchKernelGGL(kb_delta_home, grid_for((int)std::min<size_t>(n, 1u << 30), B), dim3(BT), 0, st, Y, rank, D, fg, (unsigned)n);
        WN_HIP(ctx, hipGetLastError());
    }
    *ok = true;
    return WN_OK;
}

// Gini term and gradient of a chunk: the short-key path when it applies, else gini_big.  The chunk is
// sorted in slices of at most 2^23 coefficients (32 examples of 512x512, 8 signals of 2^20 samples).
int gini_large(wn_context* ctx, const double* Y, double* D, int B, size_t n, double* keysA, double* keysB,
               unsigned* idxA, unsigned* idxB, unsigned* cnt, double* gpart, double* fg, double* S_e, void* sort_tmp,
               size_t sort_tmp_bytes, int* rank_slice = nullptr, bool counts_ready = false, unsigned* kmax = nullptr,
               bool kmax_ready = false, const unsigned** pending = nullptr, const unsigned* hist_need = nullptr, RankPairs* rp = nullptr,
               const KmaxParts* kp = nullptr) {
    // rank_slice (with D null): receives the number of examples per slice when the ranks were left in keysB
    // (slice s at keysB + s * slice * n, as 32-bit ranks at home positions) for a consumer that computes
    // the errors itself, 0 when this path did not apply and only the sums were computed.
    // counts_ready: cnt already holds the non-zero count of every CP_BLOCK coefficients of the chunk.
    if (rank_slice) *rank_slice = 0;
    int lgn = 0;
    while (((size_t)1 << lgn) < n) lgn++;
    if (ctx->bucket_rank && !ctx->exact_sort && kmax) {
        bool ok = false;
        const int rc = gini_buckets(ctx, Y, D, B, n, keysA, keysB, idxA, idxB, fg, S_e, kmax, kmax_ready, &ok, pending, hist_need, rp, kp);
        if (!ok && rp) rp->used = false;
        if (rc) return rc;
        if (ok) {
            if (rank_slice) *rank_slice = B;   // one slice: example e's ranks at (unsigned*)keysB + e * n
            return WN_OK;
        }
        counts_ready = false;   // the caller skipped the non-zero counts this path does not need
        const bool verbose = getenv("WAVENET_HIP_VERBOSE") != nullptr;
        if (verbose) fprintf(stderr, "wavenet_hip: bucket ranking gave up on a chunk of %d examples of %zu coefficients, radix sort instead\n", B, n);
    }
    const int exbits_max = 26 - lgn;   // slices of 2^26 coefficients
    if (exbits_max < 0 || ctx->exact_sort)
        return gini_big(ctx, Y, D, B, n, keysA, keysB, idxA, idxB, gpart, fg, S_e, sort_tmp, sort_tmp_bytes);
    const int slice = std::min(B, 1 << exbits_max);
    int exbits = 0;
    while ((1 << exbits) < slice) exbits++;
    const int magbits = 32 - exbits;
    hipStream_t st = ctx->stream;
    // cnt: per workgroup of CP_BLOCK coefficients the number of non-zeros before it (within its slice),
    // then per example {first sorted position, non-zeros}, then per slice the number of non-zeros, then a flag
    const int bpe = (int)(n / CP_BLOCK);                 // n is a power of two >= 8192
    unsigned* segs = cnt + (size_t)B * bpe;
    unsigned* totals = segs + 2 * (size_t)B;
    const int nslices = (B + slice - 1) / slice;
    unsigned* flag = totals + nslices;
    if (!counts_ready) hipLaunchKernelGGL(kb_count_nonzero, dim3((unsigned)B * bpe), dim3(BT), 0, st, Y, cnt);
    hipLaunchKernelGGL(kb_scan_counts, dim3(nslices), dim3(BT), 0, st, cnt, bpe, B, slice, segs, totals);
    WN_HIP(ctx, hipMemsetAsync(flag, 0, sizeof(unsigned), st));
    const int gx = (int)std::min<size_t>(n / FIX_TILE, GXMAX);
    std::vector<unsigned> nonzero(nslices);
    WN_HIP(ctx, hipMemcpyAsync(nonzero.data(), totals, nslices * sizeof(unsigned), hipMemcpyDeviceToHost, st));
    WN_HIP(ctx, hipStreamSynchronize(st));   // the sorts take their sizes from the host
    for (int e0 = 0, sl = 0; e0 < B; e0 += slice, sl++) {
        const int Bs = std::min(slice, B - e0);
        const size_t off = (size_t)e0 * n, total = nonzero[sl];
        if (total == 0) continue;
        unsigned* keyA = reinterpret_cast<unsigned*>(keysA + off);   // two 32-bit arrays in the slice of keysA
        unsigned* keyB = keyA + (size_t)Bs * n;
        hipLaunchKernelGGL(kb_key32_compact, dim3(Bs * bpe), dim3(BT), 0, st, Y + off, cnt + (size_t)e0 * bpe, keyA, idxA + off, lgn, magbits);
        WN_HIP(ctx, rocprim::radix_sort_pairs(sort_tmp, sort_tmp_bytes, keyA, keyB, idxA + off, idxB + off, total, 0, 32, st));
        // ranks go to the slice of keysB (free: the sorted keys sit in the second half of the slice of keysA)
        const int rgx = (int)std::min<size_t>(n / RANK_TILE, RANK_GX);
        hipLaunchKernelGGL(kb_rank_home, dim3((unsigned)rgx * (((unsigned)Bs + 7u) & ~7u)), dim3(BT), 0, st, keyB, idxB + off, Y + off,
                           (unsigned)n, segs + 2 * (size_t)e0, reinterpret_cast<unsigned*>(keysB + off), flag, rgx, Bs);
    }
    unsigned over = 0;
    WN_HIP(ctx, hipMemcpyAsync(&over, flag, sizeof(unsigned), hipMemcpyDeviceToHost, st));
    WN_HIP(ctx, hipStreamSynchronize(st));
    if (over) return gini_big(ctx, Y, D, B, n, keysA, keysB, idxA, idxB, gpart, fg, S_e, sort_tmp, sort_tmp_bytes);
    for (int e0 = 0; e0 < B; e0 += slice) {
        const int Bs = std::min(slice, B - e0);
        const size_t off = (size_t)e0 * n;
        const unsigned* rk = reinterpret_cast<const unsigned*>(keysB + off);
        hipLaunchKernelGGL(kb_sums_home, dim3(gx, Bs), dim3(BT), 0, st, Y + off, rk, (unsigned)n, gpart + (size_t)e0 * GXMAX * 2);
        hipLaunchKernelGGL(kb_gini_finish, dim3((Bs + BT - 1) / BT), dim3(BT), 0, st, gpart + (size_t)e0 * GXMAX * 2, Bs, gx,
                           (unsigned)n, fg + 2 * (size_t)e0, S_e ? S_e + e0 : nullptr);
        if (D) hipLaunchKernelGGL(kb_delta_home, grid_for((int)std::min<size_t>(n, 1u << 30), Bs), dim3(BT), 0, st, Y + off, rk,
                                  D + off, fg + 2 * (size_t)e0, (unsigned)n);
    }
    WN_HIP(ctx, hipGetLastError());
    if (rank_slice) *rank_slice = slice;
    return WN_OK;
}

size_t sort_temp_bytes(size_t total) {
    size_t a = 0, b = 0;
    (void)rocprim::radix_sort_pairs(nullptr, a, (double*)nullptr, (double*)nullptr, (unsigned*)nullptr,
                                    (unsigned*)nullptr, total, 0, 64, (hipStream_t)0);
    (void)rocprim::radix_sort_pairs(nullptr, b, (unsigned*)nullptr, (unsigned*)nullptr, (unsigned*)nullptr,
                                    (unsigned*)nullptr, total, 0, 32, (hipStream_t)0);
    return std::max(a, b);
}

struct Plan {
    int Bc;                 // examples per chunk
    int pgx;                // partial-sum slots per example the buffers are sized for (>= GXMAX)
    size_t n;
    double *Prow, *Y1, *Pcol, *YD, *A, *Bb, *gpart, *fg, *partial, *grad_e, *S_e;
    unsigned *idxA, *idxB, *cnt, *kmax;
    void* sort_tmp;
    size_t sort_tmp_bytes;
};

int make_plan(wn_context* ctx, int B, size_t n, int NP, int N, Plan* pl) {
    const size_t cap = (size_t)32 << 30;  // bytes of workspace at most
    const size_t per_example = n * (6 * sizeof(double) + 2 * sizeof(unsigned)) + GXMAX * (2 * NP + 2) * sizeof(double) + 64 * sizeof(double);
    size_t Bc = std::max<size_t>(1, std::min<size_t>((size_t)B, cap / per_example));
    Bc = std::min<size_t>(Bc, ((size_t)1 << 31) / n);  // chunk positions are kept in 32 bits
    Bc = std::min<size_t>(Bc, 65535);                   // grid.y
    if (ctx->chunk_cap > 0) Bc = std::min<size_t>(Bc, (size_t)ctx->chunk_cap);
    if (Bc < 1) return fail(ctx, WN_ERR_UNSUPPORTED, "example too large for the streamed path");
    pl->Bc = (int)Bc;
    pl->n = n;
    // partial-sum slots per example.  (More of them for chunks of few examples do not pay: the reducing kernels hold
    // two workgroups per CU in registers, and every further workgroup only adds its own prologue and reduction.)
    const int pgx = GXMAX;
    pl->pgx = pgx;
    pl->sort_tmp_bytes = sort_temp_bytes(Bc * n);
    for (int pass = 0; pass < 2; pass++) {
        Carver c{pass ? (char*)ctx->workspace : nullptr};
        pl->Prow = c.take<double>(Bc * n);
        pl->Y1 = c.take<double>(Bc * n);
        pl->Pcol = c.take<double>(Bc * n);
        pl->YD = c.take<double>(Bc * n);
        pl->A = c.take<double>(Bc * n);
        pl->Bb = c.take<double>(Bc * n);
        pl->idxA = c.take<unsigned>(Bc * n);
        pl->idxB = c.take<unsigned>(Bc * n);
        pl->cnt = c.take<unsigned>(Bc * (n / CP_BLOCK + 1) + 3 * Bc + 64);
        pl->kmax = c.take<unsigned>(2 * Bc + 16);   // kmax[Bc], 16 spare words, then kc_fwd_cols_hist's marks [Bc]
        pl->gpart = c.take<double>(Bc * GXMAX * 2);
        pl->fg = c.take<double>(Bc * 2);
        pl->partial = c.take<double>(Bc * (size_t)pgx * 2 * NP);
        pl->grad_e = c.take<double>(Bc * N);
        pl->S_e = c.take<double>(Bc);
        pl->sort_tmp = c.take<char>(pl->sort_tmp_bytes);
        if (!pass) {
            const int rc = ensure_workspace(ctx, c.used + 256);
            if (rc) return rc;
        }
    }
    return WN_OK;
}

Geom row_geom(int lgR, int lgC) { return Geom{1 << lgR, 1, lgR, lgC, 1}; }
Geom col_geom(int lgR, int lgC) { return Geom{1, 1 << lgR, lgC, lgR, 0}; }

template <int NP>
int train_impl(wn_context* ctx, const double* X, int B, int lgR, int lgC, double* grad, double* S, bool forward_only,
               double* Yout, double* ProwOut, double* Y1Out, double* PcolOut) {
    const size_t n = (size_t)1 << (lgR + lgC);
    const int N = ctx->N;
    Plan pl;
    int rc = make_plan(ctx, B, n, NP, N, &pl);
    if (rc) return rc;
    hipStream_t st = ctx->stream;
    const Geom rg = row_geom(lgR, lgC), cg = col_geom(lgR, lgC);
    // One chunk; with *pending set on return, the bucket ranking's verdict (a device flag) is still out.
    auto run_chunk = [&](int e0, bool try_buckets, const unsigned** pending) -> int {
        int rc = WN_OK;
        const int Bc = std::min(pl.Bc, B - e0);
        const double* Xc = X + (size_t)e0 * n;
        double* Prow = ProwOut ? ProwOut + (size_t)e0 * n : pl.Prow;
        double* Y1 = Y1Out ? Y1Out + (size_t)e0 * n : pl.Y1;
        double* Pcol = PcolOut ? PcolOut + (size_t)e0 * n : pl.Pcol;
        double* YD = Yout ? Yout + (size_t)e0 * n : pl.YD;
        // Lines of 128 .. 1024 samples go through the one-wave-per-line kernels (wn_lines.h): rows in tiles of
        // 16 (8) adjacent rows through LDS, columns directly.  No pyramid is written unless the caller asked
        // for it -- the backward kernels rebuild it in registers.  Up to 16 taps (at 16 for lines of 256 and 512
        // only: measured neutral at 1024, slower at 128): beyond, the cross-lane halos and the replicated small
        // levels of the register cascade cost more than the extra passes of the level kernels (measured at
        // 512x512, ms per 2048 images, line kernels against level kernels: 8 taps 21.6 / 28.3 per 1024; 16 taps,
        // compiled for one wave per SIMD so that nothing spills: rows and columns 31.7, columns only 38.8;
        // 32 taps, 512 images: columns only 34.1, rows only 48.9, neither 29.7).
        const int R = 1 << lgR, C = 1 << lgC;
        const int Mrow = C / 64, Mcol = R / 64;     // samples per lane
        const bool rowLines = (NP <= ctx->row_line_taps && (NP <= 8 || lgC == 8 || lgC == 9) && lgC >= 7 && lgC <= 10 && R >= line_tile_rows(Mrow) && ctx->wave64 && !ProwOut);
        const bool colLines = ((NP <= 8 || (NP == 16 && (lgR == 8 || lgR == 9))) && lgR >= 7 && lgR <= 10 && lgC > 0 && ctx->wave64);
        const bool buckets = try_buckets && ctx->bucket_rank && !ctx->exact_sort && (lgR + lgC) <= rk::RK_LGN_MAX;
        const double* Yc = YD;
        double* pcolArg = PcolOut ? Pcol : nullptr;
        // the column kernel also builds the ranking's histograms when both passes run on lines and there is an example per CU
        int NF = 0, lgBPO = 0;
        rank_bins(n, &NF, &lgBPO);
        const bool fusedHist = rowLines && colLines && buckets && !forward_only && !pcolArg && ctx->fused_hist && n >= 8192 &&
                               Bc >= ctx->numCU && NF <= LN_HIST_BINS;
        // (the row kernel's per-wave maxima sit in the index array of the ranking, which is idle until the coefficients exist)
        const int rowGroups = std::min(R / line_tile_rows(Mrow), LN_GX);
        int rowParts = 4 * rowGroups;
        unsigned* kmaxRows = fusedHist ? pl.idxA : nullptr;
        KmaxParts kmaxParts{nullptr, 0, 0};
        unsigned* histNeed = fusedHist ? pl.kmax + (size_t)Bc + 16 : nullptr;
        size_t xs = n;
        int ncols = C, nrows = R;
        const FilterDev* dfilt = ctx->dfilt;
        double* partial = pl.partial;
        if (lgC == 0 && !Y1Out && !ProwOut) {
            // 1D: the row pass is the identity, so the column pass reads the signals where they are
            Y1 = const_cast<double*>(Xc);
        } else if (rowLines) {
            const void* fwd = line_kernel(NP, LN_FWD_ROWS, Mrow);
            const void* grd = line_kernel(NP, LN_GRAD_ROWS, Mrow);
            // LDS beyond the default limit
            WN_HIP(ctx, hipFuncSetAttribute(fwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)line_tile_bytes(Mrow)));
            WN_HIP(ctx, hipFuncSetAttribute(grd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)line_tile_bytes(Mrow)));
            if (fusedHist) WN_HIP(ctx, hipMemsetAsync(pl.kmax, 0, (size_t)(2 * Bc + 16) * sizeof(unsigned), st));   // the ranking's flag among them
            void* args[] = {(void*)&Xc, (void*)&Y1, (void*)&xs, (void*)&nrows, (void*)&dfilt, (void*)&kmaxRows};
            WN_HIP(ctx, hipLaunchKernel(fwd, dim3(rowGroups, Bc), dim3(256), args, line_tile_bytes(Mrow), st));
        } else if ((rc = fwd_pass_big<NP>(ctx, Xc, Prow, Y1, Bc, n, rg))) {
            return rc;
        }
        if (fusedHist) {
            const double* y1c = Y1;
            const void* fn = line_kernel(NP, LN_FWD_COLS_HIST, Mcol);
            unsigned* hist = pl.idxB;      // where gini_buckets expects them
            unsigned* kmaxOut = pl.kmax;
            WN_HIP(ctx, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, LN_HIST_BINS * 4));
            void* args[] = {(void*)&y1c, (void*)&YD, (void*)&xs, (void*)&ncols, (void*)&dfilt, (void*)&kmaxRows, (void*)&rowParts, (void*)&kmaxOut,
                            (void*)&histNeed, (void*)&hist, (void*)&NF, (void*)&lgBPO};
            WN_HIP(ctx, hipLaunchKernel(fn, dim3(Bc), dim3(LN_HIST_T), args, (size_t)LN_HIST_BINS * 4, st));
        } else if (colLines) {
            const double* y1c = Y1;
            // the Gini phase's first step rides along: non-zero counts per CP_BLOCK coefficients
            // (bucket ranking: the largest exponent word per example; radix ranking: the non-zero counts)
            unsigned* counts = (forward_only || buckets) ? nullptr : pl.cnt;
            unsigned* kmaxArg = (!forward_only && buckets) ? pl.kmax : nullptr;
            if (counts) WN_HIP(ctx, hipMemsetAsync(counts, 0, (size_t)Bc * (n / CP_BLOCK) * sizeof(unsigned), st));
            if (kmaxArg) WN_HIP(ctx, hipMemsetAsync(kmaxArg, 0, (size_t)(Bc + 16) * sizeof(unsigned), st));   // and the ranking's flag behind them
            void* args[] = {(void*)&y1c, (void*)&YD, (void*)&pcolArg, (void*)&xs, (void*)&ncols, (void*)&dfilt, (void*)&counts, (void*)&kmaxArg};
            WN_HIP(ctx, hipLaunchKernel(line_kernel(NP, LN_FWD_COLS, Mcol), dim3((std::min(C, LN_GX) + 3) / 4, Bc), dim3(256), args, 0, st));
        } else {
            // long contiguous lines: the forward kernels leave the largest exponent words for the bucket ranking (in its
            // index array, which is idle until the coefficients are partitioned; fewer than n / 128 slots per example)
            unsigned* kparts = (!forward_only && buckets) ? pl.idxA : nullptr;
            kmaxParts.parts = kparts;
            kmaxParts.stride = (int)(n / 128);
            if ((rc = fwd_pass_big<NP>(ctx, Y1, Pcol, YD, Bc, n, cg, kparts, kmaxParts.stride, &kmaxParts.count))) return rc;
        }
        if (forward_only) return WN_OK;
        double* S_e = S ? S + e0 : pl.S_e;
        double* grad_e = grad ? grad + (size_t)e0 * N : pl.grad_e;
        // with the column kernels the errors on the coefficients are computed inside the backward kernel from
        // the ranks; otherwise (or when the short-key ranking did not apply) they replace the coefficients in YD
        int rank_slice = 0;
        RankPairs rp;
        // (up to 8 taps: at 16 the kernels run one wave per SIMD, where nothing hides the gather of a region's pairs: measured 25.7 against 24.3 ms)
        rp.want = colLines && buckets && ctx->rank_pairs_cols && NP <= ctx->rank_pairs_taps && ((size_t)1 << rk::RK_IB_MIN) % (size_t)R == 0 && n >= ((size_t)1 << rk::RK_IB_MIN);
        if (colLines && (rc = gini_large(ctx, YD, nullptr, Bc, n, pl.A, pl.Bb, pl.idxA, pl.idxB, pl.cnt, pl.gpart, pl.fg, S_e, pl.sort_tmp,
                                         pl.sort_tmp_bytes, &rank_slice, !buckets, buckets ? pl.kmax : nullptr, buckets, pending, histNeed, &rp)))
            return rc;
        if (!rank_slice &&
            (rc = gini_large(ctx, YD, YD, Bc, n, pl.A, pl.Bb, pl.idxA, pl.idxB, pl.cnt, pl.gpart, pl.fg, S_e, pl.sort_tmp, pl.sort_tmp_bytes,
                             nullptr, false, try_buckets ? pl.kmax : nullptr, false, pending, nullptr, nullptr, kmaxParts.count ? &kmaxParts : nullptr)))
            return rc;
        // the one-wave-per-line kernels write GXMAX slots per example
        const int pgx = (rowLines || colLines) ? GXMAX : pl.pgx;
        WN_HIP(ctx, hipMemsetAsync(pl.partial, 0, (size_t)Bc * pgx * 2 * NP * sizeof(double), st));
        const double* D1 = nullptr;
        if (colLines && rank_slice && rp.used) {
            const double* y1c = Y1;
            double* out = pl.A;
            const double* fgp = pl.fg;
            const void* fn = line_kernel(NP, LN_BWD_COLS_REGIONS, Mcol);
            const size_t lds = ((size_t)4 << rp.ib) + (size_t)6 * rp.nbe + 16;
            void* args[] = {(void*)&y1c, (void*)&out, (void*)&xs, (void*)&ncols, (void*)&dfilt, (void*)&partial, (void*)&rp.pairs,
                            (void*)&rp.snap, (void*)&rp.meta, (void*)&rp.meta_stride, (void*)&rp.nbe, (void*)&rp.ib, (void*)&fgp, (void*)&rp.flag};
            const int regions = (int)(n >> rp.ib);
            WN_HIP(ctx, hipLaunchKernel(fn, dim3(std::min(regions, LN_GX / 4), Bc), dim3(256), args, lds, st));
            D1 = pl.A;
        } else if (colLines) {
            const double* y1c = Y1;
            double* out = pl.A;
            const double* ranks = rank_slice ? pl.Bb : nullptr;
            const double* fg = rank_slice ? pl.fg : nullptr;
            int slice = rank_slice ? rank_slice : 1;
            void* args[] = {(void*)&Yc, (void*)&y1c, (void*)&out, (void*)&xs, (void*)&ncols, (void*)&dfilt, (void*)&partial,
                            (void*)&ranks, (void*)&slice, (void*)&fg};
            WN_HIP(ctx, hipLaunchKernel(line_kernel(NP, rank_slice ? LN_BWD_COLS_RANKS : LN_BWD_COLS, Mcol),
                                        dim3((std::min(C, LN_GX) + 3) / 4, Bc), dim3(256), args, 0, st));
            D1 = pl.A;
        } else if ((rc = bwd_pass_big<NP>(ctx, YD, Y1, Pcol, pl.A, pl.Bb, Bc, n, cg, lgC > 0, pl.partial, pgx, &D1, Yout ? nullptr : YD))) {
            return rc;
        }
        if (lgC > 0) {
            // D1 sits in A or Bb (or is YD itself when the column pass had no level); scratch: the other one and Pcol
            double* other = (D1 == pl.A) ? pl.Bb : pl.A;
            if (rowLines) {
                void* args[] = {(void*)&D1, (void*)&Xc, (void*)&xs, (void*)&nrows, (void*)&dfilt, (void*)&partial};
                WN_HIP(ctx, hipLaunchKernel(line_kernel(NP, LN_GRAD_ROWS, Mrow), dim3(std::min(R / line_tile_rows(Mrow), LN_GX), Bc),
                                            dim3(256), args, line_tile_bytes(Mrow), st));
            } else if ((rc = bwd_pass_big<NP>(ctx, D1, Xc, Prow, other, Pcol, Bc, n, rg, false, pl.partial, pgx, nullptr))) {
                return rc;
            }
        }
        hipLaunchKernelGGL(kb_grad_finish, dim3(Bc), dim3(256), 0, st, pl.partial, NP, ctx->dfilt, grad_e, pgx);
        hipLaunchKernelGGL(kb_batch_sums, dim3(N + 2), dim3(256), 0, st, grad_e, S_e, Bc, N, ctx->sums, e0 == 0 ? 1 : 0, *pending);
        WN_HIP(ctx, hipGetLastError());
        return WN_OK;
    };
    for (int e0 = 0; e0 < B; e0 += pl.Bc) {
        // The bucket ranking gives up on an example whose magnitudes crowd into one bucket, and says so in a device flag.
        // Rather than wait for that verdict in the middle of the chunk with the GPU idle, everything is queued first
        // (kb_batch_sums keeps its hands off the sums when the flag is up) and the chunk is done again without the
        // bucket ranking in the rare case that it has to be.
        const unsigned* pending = nullptr;
        if ((rc = run_chunk(e0, true, &pending))) return rc;
        if (pending) {
            // fetched beside the main stream: the host waits for the ranking only, not for the backward pass behind it,
            // and is back in time to queue the next batch before the GPU runs dry
            unsigned over = 0;
            if (!ctx->side_stream) WN_HIP(ctx, hipStreamCreateWithFlags(&ctx->side_stream, hipStreamNonBlocking));
            WN_HIP(ctx, hipStreamWaitEvent(ctx->side_stream, ctx->rank_event, 0));
            WN_HIP(ctx, hipMemcpyAsync(&over, pending, sizeof(unsigned), hipMemcpyDeviceToHost, ctx->side_stream));
            WN_HIP(ctx, hipStreamSynchronize(ctx->side_stream));
            if (over) {
                if (getenv("WAVENET_HIP_VERBOSE"))
                    fprintf(stderr, "wavenet_hip: bucket ranking gave up on a chunk of %d examples of %zu coefficients, radix sort instead\n",
                            std::min(pl.Bc, B - e0), n);
                pending = nullptr;
                if ((rc = run_chunk(e0, false, &pending))) return rc;
            }
        }
    }
    return WN_OK;
}


template <int NP>
int inverse_impl(wn_context* ctx, const double* Y, int B, int lgR, int lgC, double* Xout) {
    const size_t n = (size_t)1 << (lgR + lgC);
    Plan pl;
    int rc = make_plan(ctx, B, n, NP, ctx->N, &pl);
    if (rc) return rc;
    const Geom rg = row_geom(lgR, lgC), cg = col_geom(lgR, lgC);
    for (int e0 = 0; e0 < B; e0 += pl.Bc) {
        const int Bc = std::min(pl.Bc, B - e0);
        const double* Yc = Y + (size_t)e0 * n;
        const double* mid = nullptr;
        const double* fin = nullptr;
        if ((rc = bwd_pass_big<NP>(ctx, Yc, nullptr, nullptr, pl.A, pl.Bb, Bc, n, cg, true, nullptr, GXMAX, &mid))) return rc;
        double* s1 = (mid == pl.A) ? pl.Bb : pl.A;
        if ((rc = bwd_pass_big<NP>(ctx, mid, nullptr, nullptr, s1, pl.Pcol, Bc, n, rg, true, nullptr, GXMAX, &fin))) return rc;
        WN_HIP(ctx, hipMemcpyAsync(Xout + (size_t)e0 * n, fin, (size_t)Bc * n * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    }
    return WN_OK;
}

template <int NP>
int backprop_impl(wn_context* ctx, const double* Delta, const double* X, const double* Prow, const double* Y1,
                  const double* Pcol, int B, int lgR, int lgC, double* grad, double* din) {
    const size_t n = (size_t)1 << (lgR + lgC);
    const int N = ctx->N;
    Plan pl;
    int rc = make_plan(ctx, B, n, NP, N, &pl);
    if (rc) return rc;
    hipStream_t st = ctx->stream;
    const Geom rg = row_geom(lgR, lgC), cg = col_geom(lgR, lgC);
    for (int e0 = 0; e0 < B; e0 += pl.Bc) {
        const int Bc = std::min(pl.Bc, B - e0);
        const size_t o = (size_t)e0 * n;
        const int pgx = pl.pgx;
        WN_HIP(ctx, hipMemsetAsync(pl.partial, 0, (size_t)Bc * pgx * 2 * NP * sizeof(double), st));
        const double* D1 = nullptr;
        if ((rc = bwd_pass_big<NP>(ctx, Delta + o, Y1 + o, Pcol + o, pl.A, pl.Bb, Bc, n, cg, true, pl.partial, pgx, &D1))) return rc;
        if (din) WN_HIP(ctx, hipMemcpyAsync(din + o, D1, (size_t)Bc * n * sizeof(double), hipMemcpyDeviceToDevice, st));
        if (lgC > 0) {
            double* other = (D1 == pl.A) ? pl.Bb : pl.A;
            if ((rc = bwd_pass_big<NP>(ctx, D1, X + o, Prow + o, other, pl.Pcol, Bc, n, rg, false, pl.partial, pgx, nullptr))) return rc;
        }
        hipLaunchKernelGGL(kb_grad_finish, dim3(Bc), dim3(256), 0, st, pl.partial, NP, ctx->dfilt, grad + (size_t)e0 * N, pgx);
        WN_HIP(ctx, hipGetLastError());
    }
    return WN_OK;
}

}  // namespace

int large_train(wn_context* ctx, const double* X, int B, int lgR, int lgC, double* grad, double* S) {
#define CALL(NPV) train_impl<NPV>(ctx, X, B, lgR, lgC, grad, S, false, nullptr, nullptr, nullptr, nullptr)
    switch (ctx->NP) { case 2: return CALL(2); case 4: return CALL(4); case 8: return CALL(8); case 16: return CALL(16); case 32: return CALL(32); }
#undef CALL
    return fail(ctx, WN_ERR_UNSUPPORTED, "no kernels for this filter length");
}

int large_forward(wn_context* ctx, const double* X, int B, int lgR, int lgC, double* Y, double* Prow, double* Y1, double* Pcol) {
#define CALL(NPV) train_impl<NPV>(ctx, X, B, lgR, lgC, nullptr, nullptr, true, Y, Prow, Y1, Pcol)
    switch (ctx->NP) { case 2: return CALL(2); case 4: return CALL(4); case 8: return CALL(8); case 16: return CALL(16); case 32: return CALL(32); }
#undef CALL
    return fail(ctx, WN_ERR_UNSUPPORTED, "no kernels for this filter length");
}

int large_inverse(wn_context* ctx, const double* Y, int B, int lgR, int lgC, double* X) {
#define CALL(NPV) inverse_impl<NPV>(ctx, Y, B, lgR, lgC, X)
    switch (ctx->NP) { case 2: return CALL(2); case 4: return CALL(4); case 8: return CALL(8); case 16: return CALL(16); case 32: return CALL(32); }
#undef CALL
    return fail(ctx, WN_ERR_UNSUPPORTED, "no kernels for this filter length");
}

int large_backprop(wn_context* ctx, const double* Delta, const double* X, const double* Prow, const double* Y1,
                   const double* Pcol, int B, int lgR, int lgC, double* grad, double* din) {
#define CALL(NPV) backprop_impl<NPV>(ctx, Delta, X, Prow, Y1, Pcol, B, lgR, lgC, grad, din)
    switch (ctx->NP) { case 2: return CALL(2); case 4: return CALL(4); case 8: return CALL(8); case 16: return CALL(16); case 32: return CALL(32); }
#undef CALL
    return fail(ctx, WN_ERR_UNSUPPORTED, "no kernels for this filter length");
}

int large_sparse(wn_context* ctx, const double* Y, int B, long long n, double* Delta, double* S) {
    Plan pl;
    int rc = make_plan(ctx, B, (size_t)n, 2, 2, &pl);
    if (rc) return rc;
    for (int e0 = 0; e0 < B; e0 += pl.Bc) {
        const int Bc = std::min(pl.Bc, B - e0);
        const size_t o = (size_t)e0 * n;
        if ((rc = gini_large(ctx, Y + o, Delta ? Delta + o : nullptr, Bc, (size_t)n, pl.A, pl.Bb, pl.idxA, pl.idxB,
                           pl.cnt, pl.gpart, pl.fg, S ? S + e0 : nullptr, pl.sort_tmp, pl.sort_tmp_bytes, nullptr, false, pl.kmax, false)))
            return rc;
    }
    return WN_OK;
}

}  // namespace wn
