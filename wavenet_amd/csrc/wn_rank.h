// Gini ranks of examples beyond 4096 coefficients without a device-wide sort (src/CostFunctions.cxx:67-102:
// the rank of |c_j| among the example's magnitudes, ties in index order).
//
// A full sort moves every (key, position) pair through memory once per radix pass.  Here the only thing
// ever sorted is a bucket of at most 4096 magnitudes, in LDS:
//   1. kr_hist   per example a histogram of the non-zero magnitudes over NF logarithmic bins (bins per octave
//                fixed, counted down from the example's largest magnitude, whose exponent word the producer
//                of the coefficients left in kmax), built in LDS;
//   2. kr_scan   prefix sums of the histogram; consecutive bins are grouped into buckets by cutting the sorted
//                order every T positions (bucket of a bin = its first sorted position / T), so a bucket
//                holds fewer than T + (largest bin) magnitudes; T = RK_CAP - (the example's largest bin), at least RK_T;
//   3. kr_part   every non-zero magnitude, with its home index, goes to its bucket's range of a staging array
//                (order inside the bucket: as it comes);
//   4. kr_bucket one workgroup per bucket: a counting sort in LDS over 4096 sub-bins that split the bucket's
//                own magnitude range linearly, exact order inside a sub-bin from 64-bit keys (what is left of the
//                magnitude, then the home index), the rank written to the coefficient's home position, and the
//                bucket's share of the two Gini sums from the magnitudes in sorted order;
//   5. kr_finish adds the buckets' shares per example: lane l of a wave takes buckets l, l + 64, ... in order, the lanes are
//                combined by a butterfly.
// Exact zeros are left out (they rank first and need no rank of their own).  Every rank is decided by exact
// comparisons, so the result does not depend on the order in which atomics land; every sum is taken in a fixed
// order.  A bucket that would hold more than RK_CAP magnitudes (thousands of coefficients within half a percent
// of each other, or equal) raises a flag and the caller falls back to the radix-sort path.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "wn_device.h"

namespace wn {
namespace rk {

constexpr int RK_T = 3072;       // sorted positions per bucket
constexpr int RK_CAP = 4096;     // magnitudes a bucket may hold
constexpr int RK_NFMAX = 16384;  // histogram bins at most (512 per octave over RK_OCTAVES octaves)
constexpr int RK_OCTAVES = 32;   // octaves below the largest magnitude that get bins of their own; what lies below shares the lowest bin
constexpr int RK_TILE = 2048;    // coefficients per step of kr_part
constexpr int RK_PT = 512;       // threads of kr_hist / kr_part
constexpr int RK_BT = 1024;      // threads of kr_scan
constexpr int RK_KT = 512;       // threads of kr_bucket
constexpr int RK_KE = 8;         // magnitudes (and sub-bins) per thread of kr_bucket
constexpr int RK_LGNSB = 12;     // log2 of the sub-bins of a bucket
constexpr int RK_NBMAX = 1024;   // buckets per example at most: examples of up to 2^21 coefficients
constexpr int RK_LGN_MAX = 21;
// Packed staging keys carry the home index modulo a region of 2^ib positions (the rest of the index comes from cursor
// snapshots per region).  ib is chosen per launch: regions of at least 2^RK_IB_MIN positions (whole steps of kr_part8),
// about 32 of them per example -- kr_bucket follows the regions' boundaries slot by slot, so a bucket that is cut into
// many regions pays for each boundary -- and whole regions per workgroup of kr_part8.
constexpr int RK_IB_MIN = 13;
constexpr int RK_STEPS_MAX = (1 << RK_LGN_MAX) >> RK_IB_MIN;    // regions per example at most
__host__ __device__ inline int rank_ib(int lgn, size_t per_workgroup) {
    int ib = lgn - 5 > RK_IB_MIN ? lgn - 5 : RK_IB_MIN;
    while (ib > RK_IB_MIN && ((size_t)1 << ib) > per_workgroup) ib--;
    return ib;
}

// Workgroup barrier for hand-overs through LDS only: waits for this wave's LDS operations, not for its global
// loads and stores (__syncthreads() also drains those, which would put the memory latency of every prefetch and
// every scattered store on the critical path of each step).
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

__host__ __device__ inline int nbmax_for(size_t n) { return (int)(n / RK_T) + 2; }
// per example: u32 words {nz, nb, Lmax, kmax, bstart[nbmax + 1], binfirst[nbmax + 1], cursor[nbmax]}, then doubles bpart[2 nbmax]
__host__ __device__ inline size_t meta_words(size_t n) {
    const size_t nb = (size_t)nbmax_for(n);
    return ((4 + 2 * (nb + 1) + nb + 1) & ~(size_t)1) + 4 * nb;
}
__device__ __forceinline__ unsigned* meta_bstart(unsigned* m) { return m + 4; }
__device__ __forceinline__ unsigned* meta_binfirst(unsigned* m, int nbmax) { return m + 4 + nbmax + 1; }
__device__ __forceinline__ const unsigned* meta_bstart(const unsigned* m) { return m + 4; }
__device__ __forceinline__ const unsigned* meta_binfirst(const unsigned* m, int nbmax) { return m + 4 + nbmax + 1; }
__device__ __forceinline__ unsigned* meta_cursor(unsigned* m, int nbmax) { return m + 4 + 2 * (nbmax + 1); }
__device__ __forceinline__ double* meta_bpart(unsigned* m, int nbmax) {
    return reinterpret_cast<double*>(m + ((4 + 2 * (nbmax + 1) + nbmax + 1) & ~1));
}
// The magnitudes (bit patterns) that fall into bins [f0, f1] lie in [*lo, *hi].
__device__ __forceinline__ void bin_range(unsigned f0, unsigned f1, unsigned Lmax, unsigned kmax, int lgBPO, int NF,
                                          unsigned long long* lo, unsigned long long* hi) {
    const int shift = 20 - lgBPO;
    const unsigned d0 = (unsigned)(NF - 1) - f0, d1 = (unsigned)(NF - 1) - f1;
    const unsigned k0 = (f0 == 0 || d0 > Lmax) ? 0u : (Lmax - d0) << shift;                    // the lowest bin takes all that is smaller
    const unsigned k1 = (d1 == 0 || d1 > Lmax) ? kmax : (((Lmax - d1) + 1u) << shift) - 1u;  // the highest bin ends at the maximum
    *lo = (unsigned long long)k0 << 32;
    *hi = ((unsigned long long)k1 << 32) | 0xFFFFFFFFull;
}

// histogram bin of a magnitude's exponent word (the upper 32 bits of |c|): lgBPO = log2(bins per octave)
__device__ __forceinline__ unsigned fine_bin(unsigned key32, unsigned Lmax, int lgBPO, int NF) {
    const unsigned L = key32 >> (20 - lgBPO);
    const unsigned d = (L > Lmax) ? 0u : Lmax - L;
    return (unsigned)(NF - 1) - (d < (unsigned)(NF - 1) ? d : (unsigned)(NF - 1));
}

// kmax[e] = largest exponent word among the example's coefficients; grid (G, B).  Used when the kernel that
// produced the coefficients did not leave it.
__global__ __launch_bounds__(RK_PT) void kr_max(const double* __restrict__ Y, size_t n, unsigned* kmax) {
    const size_t per = n / gridDim.x;
    const double* src = Y + (size_t)blockIdx.y * n + (size_t)blockIdx.x * per;
    unsigned m = 0;
    for (size_t i0 = 0; i0 < per; i0 += 16 * RK_PT) {
        double2 v[8];   // eight loads in flight
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const size_t i = i0 + 2 * (size_t)(k * RK_PT + threadIdx.x);
            v[k] = (i < per) ? *reinterpret_cast<const double2*>(src + i) : make_double2(0.0, 0.0);
        }
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const unsigned k0 = (unsigned)__double2hiint(v[k].x) & 0x7FFFFFFFu, k1 = (unsigned)__double2hiint(v[k].y) & 0x7FFFFFFFu;
            m = max(m, max(k0, k1));
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
    // one atomic per workgroup: atomics on one address are served one after the other
    __shared__ unsigned wmax[RK_PT / 64];
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 1; w < RK_PT / 64; w++) m = max(m, wmax[w]);
        if (m) atomicMax(&kmax[blockIdx.y], m);
    }
}

// hist[e][f] (+)= number of non-zero coefficients of example e in bin f; grid (G, B), NF * 4 bytes of LDS.
// G == 1: plain stores of every bin; else hist must be zero beforehand and the workgroups add.
// only (may be null): the examples to do (only[e] != 0); the others have their histogram already.
// kparts (may be null): kmax[e] is not there yet -- it is the largest of the kcount words at kparts + e * kstride that
// the producer's workgroups left; every workgroup works it out for itself and the first of an example writes kmax[e].
__global__ __launch_bounds__(RK_PT) void kr_hist(const double* __restrict__ Y, size_t n, unsigned* __restrict__ kmax,
                                                 unsigned* hist, int NF, int lgBPO, const unsigned* __restrict__ only,
                                                 const unsigned* __restrict__ kparts, int kstride, int kcount) {
    extern __shared__ unsigned rk_h[];
    __shared__ unsigned wmax[RK_PT / 64];
    const int e = blockIdx.y;
    if (only && !only[e]) return;
    for (int f = threadIdx.x; f < NF; f += RK_PT) rk_h[f] = 0;
    unsigned km;
    if (kparts) {
        unsigned m = 0;
        for (int i = threadIdx.x; i < kcount; i += RK_PT) m = max(m, kparts[(size_t)e * kstride + i]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
        if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
        __syncthreads();
        km = wmax[0];
#pragma unroll
        for (int w = 1; w < RK_PT / 64; w++) km = max(km, wmax[w]);
        if (blockIdx.x == 0 && threadIdx.x == 0) kmax[e] = km;
    } else {
        km = kmax[e];
    }
    __syncthreads();
    const unsigned Lmax = km >> (20 - lgBPO);
    const size_t per = n / gridDim.x;
    const double* src = Y + (size_t)e * n + (size_t)blockIdx.x * per;
    for (size_t i0 = 0; i0 < per; i0 += 8 * RK_PT) {
        double2 v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const size_t i = i0 + 2 * (size_t)(k * RK_PT + threadIdx.x);
            v[k] = (i < per) ? *reinterpret_cast<const double2*>(src + i) : make_double2(0.0, 0.0);
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (v[k].x != 0.0) atomicAdd(&rk_h[fine_bin((unsigned)__double2hiint(v[k].x) & 0x7FFFFFFFu, Lmax, lgBPO, NF)], 1u);
            if (v[k].y != 0.0) atomicAdd(&rk_h[fine_bin((unsigned)__double2hiint(v[k].y) & 0x7FFFFFFFu, Lmax, lgBPO, NF)], 1u);
        }
    }
    __syncthreads();
    unsigned* out = hist + (size_t)e * NF;
    if (gridDim.x == 1) {
        for (int f = threadIdx.x; f < NF; f += RK_PT) out[f] = rk_h[f];
    } else {
        for (int f = threadIdx.x; f < NF; f += RK_PT) {
            const unsigned c = rk_h[f];
            if (c) atomicAdd(&out[f], c);
        }
    }
}

// Exclusive prefix sum over the workgroup of one value per thread (T threads); *total receives the sum.
template <int T = RK_BT>
__device__ __forceinline__ unsigned block_exclusive_scan(unsigned v, unsigned* wsum /* T / 64 + 1 */, unsigned* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned t = (unsigned)__shfl_up((int)incl, o, 64);
        if (lane >= o) incl += t;
    }
    lds_barrier();
    if (lane == 63) wsum[wave] = incl;
    lds_barrier();
    if (threadIdx.x == 0) {
        unsigned run = 0;
        for (int w = 0; w < T / 64; w++) { const unsigned c = wsum[w]; wsum[w] = run; run += c; }
        wsum[T / 64] = run;
    }
    lds_barrier();
    if (total) *total = wsum[T / 64];
    return wsum[wave] + incl - v;
}

// Per example: bins -> buckets.  btab[e][f] = bucket of bin f; meta: counts, bucket starts, cursors zeroed.
// grid (B), RK_BT threads, NF * 4 bytes of dynamic LDS.
__global__ __launch_bounds__(RK_BT) void kr_scan(const unsigned* __restrict__ hist, const unsigned* __restrict__ kmax, int NF, int lgBPO,
                                                 size_t n, unsigned short* btab, unsigned* meta_all, unsigned* flag) {
    extern __shared__ unsigned rk_c[];            // cum[NF]
    __shared__ unsigned wsum[RK_BT / 64 + 1];
    __shared__ unsigned bs[RK_NBMAX], bf0[RK_NBMAX];   // bucket starts and first bins: nbmax + 1 = n / RK_T + 3 entries
    const int e = blockIdx.x, tid = threadIdx.x;
    const int nbmax = nbmax_for(n);
    unsigned* meta = meta_all + (size_t)e * meta_words(n);
    const unsigned* h = hist + (size_t)e * NF;
    const int per = NF / RK_BT;                   // consecutive bins per thread (NF is a multiple of RK_BT)
    unsigned loc[RK_NFMAX / RK_BT];
    unsigned sum = 0;
    if (per % 4 == 0) {      // the thread's bins are consecutive words: 16-byte loads
        const uint4* h4 = reinterpret_cast<const uint4*>(h + tid * per);
#pragma unroll
        for (int q = 0; q < RK_NFMAX / RK_BT / 4; q++) {
            const uint4 v = (q < per / 4) ? h4[q] : make_uint4(0u, 0u, 0u, 0u);
            loc[4 * q] = v.x; loc[4 * q + 1] = v.y; loc[4 * q + 2] = v.z; loc[4 * q + 3] = v.w;
        }
    } else {
#pragma unroll
        for (int k = 0; k < RK_NFMAX / RK_BT; k++) loc[k] = (k < per) ? h[tid * per + k] : 0u;
    }
#pragma unroll
    for (int k = 0; k < RK_NFMAX / RK_BT; k++) sum += loc[k];
    unsigned nz;
    unsigned run = block_exclusive_scan(sum, wsum, &nz);
#pragma unroll
    for (int k = 0; k < RK_NFMAX / RK_BT; k++) {
        if (k < per) { rk_c[tid * per + k] = run; run += loc[k]; }
    }
    // Sorted positions per bucket for THIS example: a bucket holds fewer than T + (largest bin) magnitudes, so with
    // T = RK_CAP - largest bin it is as full as it can safely be (94 - 98 % for bins of a few hundred; RK_T, 75 %, is
    // the floor, below which the number of buckets would outgrow the tables: an example with a bin of more than
    // RK_CAP - RK_T magnitudes is cut as before, and raises the flag if a bucket then overflows).
    unsigned big = 0;
#pragma unroll
    for (int k = 0; k < RK_NFMAX / RK_BT; k++) big = max(big, loc[k]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) big = max(big, (unsigned)__shfl_xor((int)big, o, 64));
    __shared__ unsigned wbig[RK_BT / 64];
    if ((tid & 63) == 0) wbig[tid >> 6] = big;
    for (int b = tid; b <= nbmax; b += RK_BT) { bs[b] = nz; bf0[b] = (unsigned)NF; }
    __syncthreads();
#pragma unroll
    for (int w = 0; w < RK_BT / 64; w++) big = max(big, wbig[w]);
    const unsigned T = max((unsigned)RK_T, (unsigned)RK_CAP - min(big, (unsigned)RK_CAP));
    unsigned short* bt = btab + (size_t)e * NF;
    {
        // The thread's own `per` consecutive bins: first sorted positions from registers, position / T by a multiplication
        // with 1 / T and one correction either way (positions are below 2^22, quotients below 2^10), the bucket numbers
        // stored 16 bytes at a time.
        const float invT = 1.0f / (float)T;
        auto bucket_of = [&](unsigned pos) -> int {
            unsigned q = (unsigned)((float)pos * invT);
            if (q * T > pos) q--;
            if ((q + 1u) * T <= pos) q++;
            return (int)q;
        };
        const int f0 = tid * per;
        unsigned pos = f0 ? rk_c[f0] : 0u;
        int bprev = f0 ? bucket_of(rk_c[f0 - 1]) : -1;
        unsigned packed[RK_NFMAX / RK_BT / 2];
#pragma unroll
        for (int k = 0; k < RK_NFMAX / RK_BT; k++) {
            if (k < per) {
                const int bf = bucket_of(pos);
                for (int b = bprev + 1; b <= bf; b++) { bs[b] = pos; bf0[b] = (unsigned)(f0 + k); }   // the first bin that starts in bucket b's range
                if (k & 1) packed[k / 2] |= (unsigned)bf << 16; else packed[k / 2] = (unsigned)bf;
                bprev = bf;
                pos += loc[k];
            }
        }
        if (per % 8 == 0) {
            uint4* out = reinterpret_cast<uint4*>(bt + f0);
#pragma unroll
            for (int q = 0; q < RK_NFMAX / RK_BT / 8; q++) {
                if (q < per / 8) out[q] = make_uint4(packed[4 * q], packed[4 * q + 1], packed[4 * q + 2], packed[4 * q + 3]);
            }
        } else {
#pragma unroll
            for (int k = 0; k < RK_NFMAX / RK_BT; k++) {
                if (k < per) bt[f0 + k] = (unsigned short)((packed[k / 2] >> (16 * (k & 1))) & 0xFFFFu);
            }
        }
    }
    __syncthreads();
    const int nb = nz ? (int)((nz - 1) / T) + 1 : 0;
    bool over = false;
    for (int b = tid; b <= nbmax; b += RK_BT) {
        meta_bstart(meta)[b] = bs[b];
        meta_binfirst(meta, nbmax)[b] = bf0[b];
        if (b < nbmax) meta_cursor(meta, nbmax)[b] = 0;
        if (b < nb && bs[b + 1] - bs[b] > (unsigned)RK_CAP) over = true;
    }
    if (tid == 0) {
        meta[0] = nz;
        meta[1] = (unsigned)nb;
        meta[2] = kmax[e] >> (20 - lgBPO);
        meta[3] = kmax[e];
    }
    if (over) atomicOr(flag, 1u);
}

// Non-zero magnitudes (as bit patterns) and home indices into their buckets' ranges of mag / idx (per example n
// entries, bucket b at [bstart[b], bstart[b+1])).  A step takes RK_TILE coefficients: they are counted per bucket,
// gathered bucket by bucket in LDS and copied out from there, so that a wave's stores fall into a few runs of
// consecutive addresses instead of one address per lane.  grid (G, B), RK_PT threads; dynamic LDS: btab, NF * 2 bytes.
template <bool SINGLE>   // one workgroup per example: the cursors stay in LDS
__global__ __launch_bounds__(RK_PT) void kr_part(const double* __restrict__ Y, size_t n, const unsigned short* __restrict__ btab,
                                                 unsigned* meta_all, int NF, int lgBPO, unsigned long long* mag, unsigned* idx,
                                                 const unsigned* flag) {
    static_assert(RK_TILE == 4 * RK_PT && RK_NBMAX == 2 * RK_PT, "four coefficients and two buckets per thread");
    extern __shared__ unsigned short rk_bt[];
    __shared__ unsigned cnt[RK_NBMAX], gb[RK_NBMAX], bst[RK_NBMAX];
    __shared__ unsigned short lst[RK_NBMAX];     // 78 KB with the rest: two workgroups per CU
    __shared__ unsigned long long bitsL[RK_TILE];
    __shared__ unsigned idxL[RK_TILE], dstL[RK_TILE];
    __shared__ unsigned wsum[RK_PT / 64 + 1];
    if (*flag) return;     // some bucket overflows: the caller will rank another way
    const int e = blockIdx.y, tid = threadIdx.x;
    const int nbmax = nbmax_for(n);
    unsigned* meta = meta_all + (size_t)e * meta_words(n);
    const unsigned Lmax = meta[2];
    unsigned* cursor = meta_cursor(meta, nbmax);
    for (int f = tid; f < NF; f += RK_PT) rk_bt[f] = btab[(size_t)e * NF + f];
    for (int b = tid; b < RK_NBMAX; b += RK_PT) { cnt[b] = 0; bst[b] = (b < nbmax) ? meta_bstart(meta)[b] : 0u; }
    lds_barrier();
    const size_t per = n / gridDim.x, first = (size_t)blockIdx.x * per, last = first + per;
    const double* src = Y + (size_t)e * n;
    unsigned long long* mo = mag + (size_t)e * n;
    unsigned* io = idx + (size_t)e * n;
    double2 nx[2];
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const size_t i = first + 2 * (size_t)(k * RK_PT + tid);
        nx[k] = *reinterpret_cast<const double2*>(src + (i < last ? i : last - 2));   // (beyond the end: a valid address, the value is dropped when used)
        // Used here already, on purpose: the first loads are then complete when the loop is entered, and the compiler's
        // wait for the prefetched coefficients inside the loop can be "all but the eight stores behind them" on every
        // path -- with loads still pending on the entry path it falls back to waiting for everything, stores included.
        if (!(i < last)) nx[k] = make_double2(0.0, 0.0);
    }
    for (size_t i0 = first; i0 < last; i0 += RK_TILE) {
        unsigned long long bits[4];
        unsigned short bk[4];
        unsigned o[4];
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const double2 v = (i0 + 2 * (size_t)(k * RK_PT + tid) < last) ? nx[k] : make_double2(0.0, 0.0);
            bits[2 * k] = (unsigned long long)__double_as_longlong(v.x) & 0x7FFFFFFFFFFFFFFFull;
            bits[2 * k + 1] = (unsigned long long)__double_as_longlong(v.y) & 0x7FFFFFFFFFFFFFFFull;
            // (v != 0) rather than (bits != 0): the same test as everywhere else, true for NaN
            if (!(v.x != 0.0)) bits[2 * k] = 0;
            if (!(v.y != 0.0)) bits[2 * k + 1] = 0;
            const size_t i = i0 + RK_TILE + 2 * (size_t)(k * RK_PT + tid);      // the next step's coefficients, on their way early
            nx[k] = *reinterpret_cast<const double2*>(src + (i < last ? i : last - 2));
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            bk[k] = 0;
            o[k] = 0;
            if (bits[k]) {
                bk[k] = rk_bt[fine_bin((unsigned)(bits[k] >> 32), Lmax, lgBPO, NF)];
                o[k] = atomicAdd(&cnt[bk[k]], 1u);
            }
        }
        lds_barrier();
        unsigned total;
        {   // where each bucket's coefficients of this step start: in the LDS copy (lst) and in the staging arrays (gb)
            const unsigned c0 = cnt[2 * tid], c1 = cnt[2 * tid + 1];
            const unsigned run = block_exclusive_scan<RK_PT>(c0 + c1, wsum, &total);
            lst[2 * tid] = (unsigned short)run;
            lst[2 * tid + 1] = (unsigned short)(run + c0);
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int b = 2 * tid + h;
                const unsigned c = h ? c1 : c0;
                if (c) {
                    if (SINGLE) { gb[b] = bst[b]; bst[b] += c; }
                    else gb[b] = bst[b] + atomicAdd(&cursor[b], c);
                    cnt[b] = 0;
                }
            }
        }
        lds_barrier();
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (bits[k]) {
                const size_t i = i0 + 2 * (size_t)((k >> 1) * RK_PT + tid) + (k & 1);
                const unsigned lp = lst[bk[k]] + o[k];
                bitsL[lp] = bits[k];
                idxL[lp] = (unsigned)i;
                dstL[lp] = gb[bk[k]] + o[k];
            }
        }
        lds_barrier();
        // a fixed number of stores per step (lanes beyond the step's non-zeros repeat its last element), so that the wait for
        // the prefetched coefficients can leave the stores in flight
        {
            // (a step without non-zeros: the example then has at least RK_TILE zeros, so the last RK_TILE entries of its
            // staging arrays are never read and can take the stores)
#pragma unroll
            for (int k = 0; k < RK_TILE / RK_PT; k++) {
                const unsigned p = total ? min((unsigned)(tid + k * RK_PT), total - 1u) : 0u;
                const unsigned d = total ? dstL[p] : (unsigned)n - 1u - (unsigned)tid;
                mo[d] = bitsL[p];
                io[d] = idxL[p];
            }
        }
    }
}

// ---- packed staging: one 8-byte key per magnitude --------------------------------------------------------------
// kr_part above writes 12 bytes per magnitude into two arrays.  A bucket's magnitudes lie within an octave or two of
// its lower bound, so (magnitude - lower bound) takes at most 53 bits, and the 11 bits that are left hold the home
// index modulo RK_TILE.  The rest of the index comes from WHERE a key sits in its bucket's range: a workgroup takes
// the example's coefficients step by step (RK_TILE consecutive home positions per step) and appends to every bucket
// what the step holds for it, so the keys of step s occupy the positions [snap[s][b], snap[s + 1][b]) of bucket b.
// kr_part8 leaves those cursor snapshots (16-bit offsets from the bucket's start: a bucket holds at most RK_CAP keys)
// and kr_bucket recovers home index = (step << ib) | low bits by a search over its bucket's column of them.  A bucket
// wider than 2^53 bit patterns (the lowest one, which reaches down to zero; sparse tails) keeps whatever fits of its
// magnitudes; kr_bucket reads such a bucket's exact magnitudes back from the coefficients at the home positions.
// With several workgroups per example each one's cursors start where kr_count says the workgroups before it end, so
// the placement does not depend on the order in which workgroups run.

__host__ __device__ inline int nbmax_even(size_t n) { return (nbmax_for(n) + 1) & ~1; }

// precnt[(e * G + g) * nbmax_even + b] = number of non-zero coefficients of workgroup g's share of example e that go to
// bucket b.  grid (G, B), RK_PT threads, NF * 2 bytes of dynamic LDS.  Only needed for G > 1.
__global__ __launch_bounds__(RK_PT) void kr_count(const double* __restrict__ Y, size_t n, const unsigned short* __restrict__ btab,
                                                  const unsigned* __restrict__ meta_all, int NF, int lgBPO, unsigned* precnt,
                                                  const unsigned* flag) {
    extern __shared__ unsigned short rk_bt[];
    __shared__ unsigned cnt[RK_NBMAX];
    if (*flag) return;
    const int e = blockIdx.y, tid = threadIdx.x;
    const int nbe = nbmax_even(n);
    const unsigned* meta = meta_all + (size_t)e * meta_words(n);
    const unsigned Lmax = meta[2];
    for (int f = tid; f < NF; f += RK_PT) rk_bt[f] = btab[(size_t)e * NF + f];
    for (int b = tid; b < RK_NBMAX; b += RK_PT) cnt[b] = 0;
    lds_barrier();
    const size_t per = n / gridDim.x;
    const double* src = Y + (size_t)e * n + (size_t)blockIdx.x * per;
    for (size_t i0 = 0; i0 < per; i0 += 8 * RK_PT) {
        double2 v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const size_t i = i0 + 2 * (size_t)(k * RK_PT + tid);
            v[k] = *reinterpret_cast<const double2*>(src + (i < per ? i : per - 2));
            if (!(i < per)) v[k] = make_double2(0.0, 0.0);
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (v[k].x != 0.0) atomicAdd(&cnt[rk_bt[fine_bin((unsigned)__double2hiint(v[k].x) & 0x7FFFFFFFu, Lmax, lgBPO, NF)]], 1u);
            if (v[k].y != 0.0) atomicAdd(&cnt[rk_bt[fine_bin((unsigned)__double2hiint(v[k].y) & 0x7FFFFFFFu, Lmax, lgBPO, NF)]], 1u);
        }
    }
    __syncthreads();
    unsigned* out = precnt + ((size_t)e * gridDim.x + blockIdx.x) * nbe;
    for (int b = tid; b < nbe; b += RK_PT) out[b] = cnt[b];
}

// Packed keys of the non-zero magnitudes into their buckets' ranges of keys (per example n entries, bucket b at
// [bstart[b], bstart[b+1])), and the cursor snapshots snap[(e * steps + step) * nbmax_even + b].  Steps as in kr_part.
// grid (G, B), RK_PT threads; dynamic LDS: btab, NF * 2 bytes.  precnt: kr_count's counts (G > 1), else null.
// The coefficients pass through kr_part8 once: the loads are marked non-temporal (measured: 437 -> 397 us for 64
// signals of 2^20 samples, 2.05 -> 1.99 ms for 2048 images of 512x512; RK_PART_PLAIN_LOADS compiles ordinary loads).
// The bytes the kernel writes to memory do not change with it (1.0 GB for 0.54 GB of keys at 300 buckets per
// signal, whose runs per step are 7 keys long), so what it buys is not fewer evictions of half-written lines.
__device__ __forceinline__ double2 part_load(const double2* p) {
#ifndef RK_PART_PLAIN_LOADS
    const double* q = reinterpret_cast<const double*>(p);
    return make_double2(__builtin_nontemporal_load(q), __builtin_nontemporal_load(q + 1));
#else
    return *p;
#endif
}
__global__ __launch_bounds__(RK_PT) void kr_part8(const double* __restrict__ Y, size_t n, const unsigned short* __restrict__ btab,
                                                  const unsigned* __restrict__ meta_all, int NF, int lgBPO, unsigned long long* keys,
                                                  unsigned short* snap, const unsigned* __restrict__ precnt, const unsigned* flag, int ib) {
    static_assert(RK_TILE == 4 * RK_PT && RK_NBMAX == 2 * RK_PT && (1 << RK_IB_MIN) % RK_TILE == 0, "four coefficients and two buckets per thread; whole steps per region");
    extern __shared__ unsigned short rk_bt[];
    __shared__ unsigned cnt[RK_NBMAX], gb[RK_NBMAX], bst[RK_NBMAX], bs0[RK_NBMAX], lo32[RK_NBMAX];
    __shared__ unsigned short lst[RK_NBMAX];     // 79 KB with the rest: two workgroups per CU
    __shared__ unsigned long long keyL[RK_TILE];
    __shared__ unsigned dstL[RK_TILE];
    __shared__ unsigned wsum[RK_PT / 64 + 1];
    if (*flag) return;     // some bucket overflows: the caller will rank another way
    const int e = blockIdx.y, tid = threadIdx.x;
    const int nbmax = nbmax_for(n), nbe = nbmax_even(n);
    const unsigned* meta = meta_all + (size_t)e * meta_words(n);
    const unsigned Lmax = meta[2], kmx = meta[3];
    const int nb = (int)meta[1];
    for (int f = tid; f < NF; f += RK_PT) rk_bt[f] = btab[(size_t)e * NF + f];
    for (int b = tid; b < RK_NBMAX; b += RK_PT) {
        cnt[b] = 0;
        unsigned start = (b < nbmax) ? meta_bstart(meta)[b] : 0u, low = 0;
        bs0[b] = start;
        if (b < nb) {
            const unsigned f0 = meta_binfirst(meta, nbmax)[b], fnext = meta_binfirst(meta, nbmax)[b + 1];
            unsigned long long lo, hi;
            bin_range(f0, (fnext > f0 ? fnext : (unsigned)NF) - 1u, Lmax, kmx, lgBPO, NF, &lo, &hi);
            low = (unsigned)(lo >> 32);              // (the lower bound of a range of bins has no low word)
            if (precnt) {
                for (int g = 0; g < (int)blockIdx.x; g++) start += precnt[((size_t)e * gridDim.x + g) * nbe + b];
            }
        }
        bst[b] = start;
        lo32[b] = low;
    }
    lds_barrier();
    const size_t per = n / gridDim.x, first = (size_t)blockIdx.x * per, last = first + per;
    const double* src = Y + (size_t)e * n;
    unsigned long long* ko = keys + (size_t)e * n;
    unsigned* sn = reinterpret_cast<unsigned*>(snap + (size_t)e * (n >> ib) * nbe);    // two buckets per word
    const unsigned region_mask = (1u << ib) - 1u;
    double2 nx[2];
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const size_t i = first + 2 * (size_t)(k * RK_PT + tid);
        nx[k] = part_load(reinterpret_cast<const double2*>(src + (i < last ? i : last - 2)));
        if (!(i < last)) nx[k] = make_double2(0.0, 0.0);       // (used here on purpose, see kr_part)
    }
    for (size_t i0 = first; i0 < last; i0 += RK_TILE) {
        unsigned long long bits[4];
        unsigned short bk[4];
        unsigned o[4];
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const double2 v = (i0 + 2 * (size_t)(k * RK_PT + tid) < last) ? nx[k] : make_double2(0.0, 0.0);
            bits[2 * k] = (unsigned long long)__double_as_longlong(v.x) & 0x7FFFFFFFFFFFFFFFull;
            bits[2 * k + 1] = (unsigned long long)__double_as_longlong(v.y) & 0x7FFFFFFFFFFFFFFFull;
            if (!(v.x != 0.0)) bits[2 * k] = 0;
            if (!(v.y != 0.0)) bits[2 * k + 1] = 0;
            const size_t i = i0 + RK_TILE + 2 * (size_t)(k * RK_PT + tid);      // the next step's coefficients, on their way early
            nx[k] = part_load(reinterpret_cast<const double2*>(src + (i < last ? i : last - 2)));
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            bk[k] = 0;
            o[k] = 0;
            if (bits[k]) {
                bk[k] = rk_bt[fine_bin((unsigned)(bits[k] >> 32), Lmax, lgBPO, NF)];
                o[k] = atomicAdd(&cnt[bk[k]], 1u);
            }
        }
        lds_barrier();
        unsigned total;
        {   // where each bucket's keys of this step start: in the LDS copy (lst) and in the bucket's range (gb); the cursors as they
            // stand before the step are its snapshot
            const unsigned c0 = cnt[2 * tid], c1 = cnt[2 * tid + 1];
            const unsigned run = block_exclusive_scan<RK_PT>(c0 + c1, wsum, &total);
            lst[2 * tid] = (unsigned short)run;
            lst[2 * tid + 1] = (unsigned short)(run + c0);
            const unsigned p0 = bst[2 * tid], p1 = bst[2 * tid + 1];
            if (2 * tid < nbe && ((unsigned)i0 & region_mask) == 0)      // the first step of a region
                sn[(i0 >> ib) * (size_t)(nbe / 2) + tid] = ((p0 - bs0[2 * tid]) & 0xFFFFu) | ((p1 - bs0[2 * tid + 1]) << 16);
            gb[2 * tid] = p0;
            gb[2 * tid + 1] = p1;
            bst[2 * tid] = p0 + c0;
            bst[2 * tid + 1] = p1 + c1;
            cnt[2 * tid] = 0;
            cnt[2 * tid + 1] = 0;
        }
        lds_barrier();
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (bits[k]) {
                const unsigned within = ((unsigned)i0 & region_mask) + 2u * (unsigned)((k >> 1) * RK_PT + tid) + (unsigned)(k & 1);    // home index modulo the region
                const unsigned lp = lst[bk[k]] + o[k];
                keyL[lp] = ((bits[k] - ((unsigned long long)lo32[bk[k]] << 32)) << ib) | within;
                dstL[lp] = gb[bk[k]] + o[k];
            }
        }
        lds_barrier();
        // a fixed number of stores per step, as in kr_part
#pragma unroll
        for (int k = 0; k < RK_TILE / RK_PT; k++) {
            const unsigned p = total ? min((unsigned)(tid + k * RK_PT), total - 1u) : 0u;
            const unsigned d = total ? dstL[p] : (unsigned)n - 1u - (unsigned)tid;
            ko[d] = keyL[p];
        }
    }
}

// ---- kr_bucket ------------------------------------------------------------------------------------------------
// Inclusive prefix sum over the 64 lanes of a wave by DPP row shifts and row broadcasts (no LDS traffic); every
// lane of the wave must be active.
__device__ __forceinline__ unsigned wave_inclusive_scan(unsigned v) {
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);   // row_shr:1
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);   // row_shr:2
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);   // row_shr:4
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);   // row_shr:8
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);   // row_bcast:15 into rows 1 and 3
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);   // row_bcast:31 into rows 2 and 3
    return v;
}

#ifndef RK_SUMS_IN_PLACE
#define RK_SUMS_IN_PLACE 0
#endif
#ifndef RK_B2_WAVES
#define RK_B2_WAVES 6     // waves per SIMD kr_bucket is compiled for (6: 80 registers, three workgroups per CU)
#endif
// LDS of kr_bucket: keys and sub-bin counters only (48.3 KB at 4096 sub-bins: three workgroups per CU).
template <int LGNSB>
struct BucketLds {
    unsigned long long keyB[RK_CAP];
    unsigned st[(1 << LGNSB) + 4];
    unsigned short col[RK_STEPS_MAX + 2];      // packed staging: the bucket's column of cursor snapshots, then its size
    unsigned wsum[RK_KT / 64];
    unsigned long long wlo[RK_KT / 64], whi[RK_KT / 64];
    double red[2 * (RK_KT / 64)];
};

// The ranks inside the sub-bins.  A magnitude's key is what is left of it below its sub-bin bits with a payload
// behind it: the home index where both fit in 64 bits, else (SLOTS) the magnitude's slot in the bucket, which always
// fits; with slots, equal magnitudes are ordered by the home indices read from memory.
// Home index of the magnitude in slot j of a bucket: from the index array, or (packed staging) from the slot's step,
// found in the bucket's column of cursor snapshots, and the low bits of its key.
template <bool PACKED8>
struct BucketHomes {
    const unsigned* ii;
    const unsigned long long* mi;
    const unsigned short* col;
    int steps, ib;
    __device__ __forceinline__ unsigned step_of(unsigned j) const {
        int lo = 0, hi = steps;                      // col[lo] <= j < col[hi] (col[steps] = size of the bucket)
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (col[mid] <= j) lo = mid; else hi = mid;
        }
        return (unsigned)lo;
    }
    __device__ __forceinline__ unsigned operator()(unsigned j) const {
        if constexpr (!PACKED8) return ii[j];
        else return (step_of(j) << ib) | ((unsigned)mi[j] & ((1u << ib) - 1u));
    }
};

// slot of the k-th magnitude of a thread: strided over the workgroup, or (packed staging) strided over the wave, so that
// a wave's slots are 512 consecutive ones and a lane's step numbers can be followed from one slot to the next
template <bool PACKED8>
__device__ __forceinline__ unsigned bucket_slot(int tid, int k) {
    if constexpr (!PACKED8) return (unsigned)(tid + k * RK_KT);
    else return (unsigned)(((tid >> 6) * RK_KE + k) * 64 + (tid & 63));
}

// so[k]: sub-bin of the thread's k-th magnitude in the low half; its place inside the sub-bin comes in and its rank
// inside the bucket goes out in the high half (both below RK_CAP).
// The rank goes to the coefficient's home position in rk, or -- pr given -- into the slot's place of pr as the pair
// (home index << 32) | rank: consecutive addresses instead of one line per rank (kr_delta_regions takes them home).
// pr32 > 0: pr is an array of 32-bit words and the pair is (home index mod 2^pr32) << (32 - pr32) | rank -- the
// consumer knows the region from where it reads (kc_bwd_cols_regions); the rank must fit 32 - pr32 bits.
template <int LGNSB, bool SLOTS, bool PACKED8>
__device__ __forceinline__ void bucket_ranks(BucketLds<LGNSB>& L, unsigned long long (&key)[RK_KE], unsigned (&so)[RK_KE],
                                              unsigned count, int PB, const BucketHomes<PACKED8>& home, unsigned* rk, unsigned rank0,
                                              unsigned long long* pr, int pr32) {
    const int tid = threadIdx.x;
    const unsigned pmask = (1u << PB) - 1u;
    if constexpr (!SLOTS) {
        // Keys are unique, so the rank inside a sub-bin is the number of smaller keys in it.  The first RK_W keys of
        // every sub-bin are read at once, without a branch (what lies beyond the sub-bin's end is read and not
        // counted): sub-bins hold 3/4 of a key on average, and a loop over their members would wait for LDS once
        // per member.  Only sub-bins of more than RK_W keys go through a loop for the rest.
        constexpr int RK_W = 4;
        constexpr int RK_G = 2;                  // keys whose sub-bins are read together
#pragma unroll
        for (int h = 0; h < RK_KE; h += RK_G) {
            unsigned g0[RK_G], sz[RK_G];
            unsigned long long m[RK_G][RK_W];
#pragma unroll
            for (int k = 0; k < RK_G; k++) {
                g0[k] = L.st[so[h + k] & 0xFFFFu];
                sz[k] = L.st[(so[h + k] & 0xFFFFu) + 1] - g0[k];
            }
#pragma unroll
            for (int k = 0; k < RK_G; k++) {
#pragma unroll
                for (int w = 0; w < RK_W; w++) m[k][w] = L.keyB[g0[k] + w];    // (at most RK_W - 1 entries beyond the keys: the counters follow)
            }
#pragma unroll
            for (int k = 0; k < RK_G; k++) {
                unsigned r = g0[k];
#pragma unroll
                for (int w = 0; w < RK_W; w++) r += ((unsigned)w < sz[k] && m[k][w] < key[h + k]) ? 1u : 0u;
                const bool on = bucket_slot<PACKED8>(tid, h + k) < count;
                if (on && sz[k] > (unsigned)RK_W) {
                    for (unsigned q = g0[k] + RK_W; q < g0[k] + sz[k]; q++) r += (L.keyB[q] < key[h + k]) ? 1u : 0u;
                }
                if (on) {
                    const unsigned hm = (unsigned)key[h + k] & pmask;
                    if (pr32) reinterpret_cast<unsigned*>(pr)[bucket_slot<PACKED8>(tid, h + k)] = ((hm & ((1u << pr32) - 1u)) << (32 - pr32)) | (rank0 + r);
                    else if (pr) pr[bucket_slot<PACKED8>(tid, h + k)] = ((unsigned long long)hm << 32) | (unsigned long long)(rank0 + r);
                    else rk[hm] = rank0 + r;
                }
                so[h + k] = (so[h + k] & 0xFFFFu) | (r << 16);
            }
        }
    } else {
#pragma unroll
        for (int k = 0; k < RK_KE; k++) {
            const unsigned j = bucket_slot<PACKED8>(tid, k);
            if (j < count) {
                const unsigned g0 = L.st[so[k] & 0xFFFFu], g1 = L.st[(so[k] & 0xFFFFu) + 1];
                unsigned r = g0;
                if (g1 - g0 > 1) {
                    const unsigned long long mine = key[k] >> PB;
                    unsigned myid = 0xFFFFFFFFu;
                    for (unsigned q = g0; q < g1; q++) {
                        const unsigned long long kq = L.keyB[q];
                        if ((kq >> PB) != mine) r += ((kq >> PB) < mine) ? 1u : 0u;
                        else if (kq != key[k]) {      // an equal magnitude in another slot: the home indices decide
                            if (myid == 0xFFFFFFFFu) myid = home(j);
                            r += (home((unsigned)kq & pmask) < myid) ? 1u : 0u;
                        }
                    }
                }
                {
                    const unsigned hm = home(j);
                    if (pr32) reinterpret_cast<unsigned*>(pr)[j] = ((hm & ((1u << pr32) - 1u)) << (32 - pr32)) | (rank0 + r);
                    else if (pr) pr[j] = ((unsigned long long)hm << 32) | (unsigned long long)(rank0 + r);
                    else rk[hm] = rank0 + r;
                }
                so[k] = (so[k] & 0xFFFFu) | (r << 16);
            }
        }
    }
}

// One bucket: exact ranks home, the bucket's share of f = sum (2 rank - n - 1) |c| and of sum |c|.  A counting sort in
// LDS over the sub-bins (the leading LGNSB bits of magnitude - bucket minimum), the workgroup-wide prefix sum of the
// counts by DPP scans and one hand-over, exact order inside a sub-bin from the keys, then the magnitudes once more in
// sorted order for the two sums (a fixed order of addition, whatever order the bucket arrived in).  lgn = log2 n.
// 1D grid of nbmax * (examples rounded up to 8) workgroups: consecutive workgroups go to consecutive XCDs, so
// workgroup L works on example (L / 8 / nbmax) * 8 + L % 8 -- all buckets of an example share one XCD, whose L2
// then collects the example's scattered ranks into whole lines.
// PACKED8: the staging holds kr_part8's keys (mag) and cursor snapshots (snap); wide buckets read their magnitudes from Y.
// pairs (may be null): the ranks leave as (home index, rank) pairs in slot order instead of going home (bucket_ranks).
template <int LGNSB, bool PACKED8>
__global__ __launch_bounds__(RK_KT, RK_B2_WAVES) void kr_bucket(const unsigned long long* __restrict__ mag, const unsigned* __restrict__ idx,
                                                                 const unsigned short* __restrict__ snap, const double* __restrict__ Y, size_t n, int lgn,
                                                                 unsigned* meta_all, unsigned* rank, int B, const unsigned* flag, int NF, int lgBPO, int ib,
                                                                 unsigned long long* pairs, int pairs32) {
    constexpr int NSB = 1 << LGNSB;
    constexpr int SPT = NSB / RK_KT;            // sub-bins per thread in the prefix sum
    static_assert(SPT % 4 == 0 && RK_CAP == RK_KE * RK_KT, "whole 16-byte pieces of counters per thread; RK_KE magnitudes per thread");
    __shared__ __attribute__((aligned(16))) BucketLds<LGNSB> L;
    const int nbmax = nbmax_for(n);
    const int slot = blockIdx.x >> 3;
    const int e = (slot / nbmax) * 8 + (blockIdx.x & 7), b = slot % nbmax;
    if (e >= B || *flag) return;
    unsigned* meta = meta_all + (size_t)e * meta_words(n);
    const unsigned nz = meta[0];
    if (b >= (int)meta[1]) return;
    const unsigned s0 = meta_bstart(meta)[b], count = meta_bstart(meta)[b + 1] - s0;
    if (count == 0 || count > (unsigned)RK_CAP) return;      // an overfull bucket has raised the flag already
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned zeros = (unsigned)n - nz;
    const unsigned long long* mi = mag + (size_t)e * n + s0;
    const unsigned* ii = idx + (size_t)e * n + s0;
    unsigned long long key[RK_KE];
    unsigned id[RK_KE], so[RK_KE];
    static_assert(LGNSB <= 16 && RK_CAP <= 65535, "sub-bin and place share a register");
#pragma unroll
    for (int k = 0; k < RK_KE; k++) {
        const unsigned j = min(bucket_slot<PACKED8>(tid, k), count - 1u);   // (slots beyond the bucket repeat its last magnitude and are left out below)
        key[k] = mi[j];
        if constexpr (!PACKED8) id[k] = ii[j];
    }
    {
        uint4* z = reinterpret_cast<uint4*>(L.st) + tid * (SPT / 4);
#pragma unroll
        for (int q = 0; q < SPT / 4; q++) z[q] = make_uint4(0u, 0u, 0u, 0u);
        if (tid < 4) L.st[NSB + tid] = 0;
    }
    // the range of the bucket's magnitudes, from the bins it is made of
    const unsigned f0 = meta_binfirst(meta, nbmax)[b], fnext = meta_binfirst(meta, nbmax)[b + 1];
    unsigned long long lo, hi;
    bin_range(f0, (fnext > f0 ? fnext : (unsigned)NF) - 1u, meta[2], meta[3], lgBPO, NF, &lo, &hi);
    const int steps = PACKED8 ? (int)(n >> ib) : 0;
    if constexpr (PACKED8) {
        // the bucket's column of cursor snapshots; a slot's step is the last one whose snapshot does not exceed it
        const int nbe = nbmax_even(n);
        const unsigned short* sc = snap + (size_t)e * steps * nbe + b;
        for (int r = tid; r < steps; r += RK_KT) L.col[r] = sc[(size_t)r * nbe];
        if (tid == 0) L.col[steps] = (unsigned short)count;
        lds_barrier();
        const bool wide = ((hi - lo) >> (64 - ib)) != 0;      // kr_part8 could not keep this bucket's magnitudes whole
        // A wave's slots are 512 consecutive ones, 64 per k, and the regions' boundaries are the same for every lane: the
        // region of a block's first slot and the boundaries inside the block are followed in scalar registers (reads of
        // one LDS address by the whole wave), and a lane only counts the boundaries that do not exceed its slot.
        const BucketHomes<true> first{nullptr, mi, L.col, steps, ib};
        const unsigned base0 = (unsigned)__builtin_amdgcn_readfirstlane((int)bucket_slot<true>(tid & ~63, 0));
        unsigned rs = (base0 < count) ? first.step_of(base0) : 0u;
        rs = (unsigned)__builtin_amdgcn_readfirstlane((int)rs);
#pragma unroll
        for (int k = 0; k < RK_KE; k++) {
            const unsigned base = base0 + 64u * (unsigned)k, j = base + (unsigned)lane;
            unsigned bnd = (unsigned)__builtin_amdgcn_readfirstlane((int)L.col[rs + 1]);
            while (rs + 1 < (unsigned)steps && bnd <= base) {          // the last region that starts at or before the block
                rs++;
                bnd = (unsigned)__builtin_amdgcn_readfirstlane((int)L.col[rs + 1]);
            }
            unsigned r = rs, t = rs + 1;
            while (t < (unsigned)steps && bnd < base + 64u) {           // regions that start inside the block
                r += (j >= bnd) ? 1u : 0u;
                t++;
                bnd = (unsigned)__builtin_amdgcn_readfirstlane((int)L.col[t]);
            }
            id[k] = (r << ib) | ((unsigned)key[k] & ((1u << ib) - 1u));
            key[k] = lo + (key[k] >> ib);
            if (wide && j < count) key[k] = (unsigned long long)__double_as_longlong(Y[(size_t)e * n + id[k]]) & 0x7FFFFFFFFFFFFFFFull;
        }
    }
    if (f0 == 0) {
        // the lowest bin reaches down to zero: the bucket's own smallest and largest magnitude instead
        lo = ~0ull;
        hi = 0;
#pragma unroll
        for (int k = 0; k < RK_KE; k++) {
            if (bucket_slot<PACKED8>(tid, k) < count) {
                lo = key[k] < lo ? key[k] : lo;
                hi = key[k] > hi ? key[k] : hi;
            }
        }
#pragma unroll
        for (int s = 32; s > 0; s >>= 1) {
            const unsigned long long l2 = (unsigned long long)__shfl_xor((long long)lo, s, 64), h2 = (unsigned long long)__shfl_xor((long long)hi, s, 64);
            lo = l2 < lo ? l2 : lo;
            hi = h2 > hi ? h2 : hi;
        }
        if (lane == 0) { L.wlo[wave] = lo; L.whi[wave] = hi; }
    }
    lds_barrier();         // counters zero, the waves' ranges in place
    if (f0 == 0) {
#pragma unroll
        for (int w = 0; w < RK_KT / 64; w++) {
            lo = L.wlo[w] < lo ? L.wlo[w] : lo;
            hi = L.whi[w] > hi ? L.whi[w] : hi;
        }
    }
    const unsigned long long span = hi - lo;
    const int sh = (span >> LGNSB) ? (64 - __builtin_clzll(span) - LGNSB) : 0;   // (span >> sh) < NSB
    const bool slots = sh + lgn > 64;            // the home index does not fit behind what is left of a magnitude
    const int PB = slots ? 12 : lgn;
    static_assert(RK_CAP <= 4096, "a slot number takes 12 bits");
    const unsigned long long remmask = sh ? ((1ull << sh) - 1ull) : 0ull;
#pragma unroll
    for (int k = 0; k < RK_KE; k++) {
        const unsigned j = bucket_slot<PACKED8>(tid, k);
        const unsigned long long rel = key[k] - lo;
        // (a slot beyond the bucket holds no magnitude of it: sub-bin 0, so that whatever is read for it below is in range)
        // (and never beyond the last sub-bin: every loop below is then bounded by the bucket's size whatever the staging holds)
        const unsigned sb = (j < count) ? min((unsigned)(rel >> sh), (unsigned)(NSB - 1)) : 0u;
        key[k] = ((rel & remmask) << PB) | (slots ? j : id[k]);
        so[k] = sb;
        if (j < count) so[k] = sb | (atomicAdd(&L.st[sb], 1u) << 16);
    }
    lds_barrier();
    {   // counts -> exclusive prefix sums, in place: SPT consecutive sub-bins per thread
        unsigned c[SPT], s = 0;
        uint4* p4 = reinterpret_cast<uint4*>(L.st) + tid * (SPT / 4);
#pragma unroll
        for (int q = 0; q < SPT / 4; q++) {
            const uint4 t = p4[q];
            c[4 * q] = t.x; c[4 * q + 1] = t.y; c[4 * q + 2] = t.z; c[4 * q + 3] = t.w;
            s += t.x + t.y + t.z + t.w;
        }
        const unsigned incl = wave_inclusive_scan(s);
        if (lane == 63) L.wsum[wave] = incl;
        lds_barrier();
        unsigned run = incl - s;
#pragma unroll
        for (int w = 0; w < RK_KT / 64 - 1; w++) run += (w < wave) ? L.wsum[w] : 0u;
#pragma unroll
        for (int q = 0; q < SPT / 4; q++) {
            uint4 t;
            t.x = run; run += c[4 * q];
            t.y = run; run += c[4 * q + 1];
            t.z = run; run += c[4 * q + 2];
            t.w = run; run += c[4 * q + 3];
            p4[q] = t;
        }
        if (tid == RK_KT - 1) L.st[NSB] = run;
    }
    lds_barrier();
#pragma unroll
    for (int k = 0; k < RK_KE; k++) {
        if (bucket_slot<PACKED8>(tid, k) < count) L.keyB[L.st[so[k] & 0xFFFFu] + (so[k] >> 16)] = key[k];
    }
    lds_barrier();
    unsigned* rk = rank + (size_t)e * n;
    const unsigned rank0 = zeros + s0 + 1u;
    const BucketHomes<PACKED8> home{ii, mi, L.col, steps, ib};
    // (an array of its own: the staging is still read for ties; with 32-bit pairs, example e at word e * n)
    unsigned long long* pr = !pairs ? nullptr
                             : pairs32 ? reinterpret_cast<unsigned long long*>(reinterpret_cast<unsigned*>(pairs) + (size_t)e * n + s0)
                                       : pairs + (size_t)e * n + s0;
    if (!slots) bucket_ranks<LGNSB, false, PACKED8>(L, key, so, count, PB, home, rk, rank0, pr, pairs32);
    else bucket_ranks<LGNSB, true, PACKED8>(L, key, so, count, PB, home, rk, rank0, pr, pairs32);
    double f = 0, a = 0;
    const double w0 = 2.0 * (double)rank0 - (double)n - 1.0;     // weight of the bucket's smallest magnitude (exact: integers below 2^53)
    if constexpr (PACKED8 && RK_SUMS_IN_PLACE) {
        // Sums taken where the magnitudes are, in the order the bucket arrived in.  That order is not the same in every
        // run (inside a step of kr_part8 the places come from LDS atomics), so the sums differ in their last bits from
        // run to run: 5% faster (3.45 against 3.64 ms at 512x512), off by default.
#pragma unroll
        for (int k = 0; k < RK_KE; k++) {
            if (bucket_slot<PACKED8>(tid, k) < count) {
                const double av = __longlong_as_double((long long)(lo + ((unsigned long long)(so[k] & 0xFFFFu) << sh) + (key[k] >> PB)));
                f = __builtin_fma(w0 + 2.0 * (double)(so[k] >> 16), av, f);
                a += av;
            }
        }
    } else {
        lds_barrier();       // every comparison has read its keys: the array now takes the magnitudes in sorted order
        unsigned long long* magS = L.keyB;
#pragma unroll
        for (int k = 0; k < RK_KE; k++) {
            if (bucket_slot<PACKED8>(tid, k) < count) magS[so[k] >> 16] = lo + ((unsigned long long)(so[k] & 0xFFFFu) << sh) + (key[k] >> PB);
        }
        lds_barrier();
#pragma unroll
        for (int k = 0; k < RK_KE; k++) {
            const unsigned r = tid + k * RK_KT;
            if (r < count) {
                const double av = __longlong_as_double((long long)magS[r]);
                f = __builtin_fma(w0 + 2.0 * (double)r, av, f);
                a += av;
            }
        }
    }
    f = wave_sum(f);
    a = wave_sum(a);
    if (lane == 0) { L.red[wave] = f; L.red[RK_KT / 64 + wave] = a; }
    lds_barrier();
    if (tid < 2) {
        double t = 0;
#pragma unroll
        for (int w = 0; w < RK_KT / 64; w++) t += L.red[tid * (RK_KT / 64) + w];
        meta_bpart(meta, nbmax)[2 * b + tid] = t;
    }
}

// Errors on the coefficients from kr_bucket's (home index, rank) pairs, region by region: for examples whose ranks do
// not fit an L2 (2^20 coefficients: 4 MB), ranks stored home one by one leave the cache as partially written lines
// over and over (measured: 381 MB written for 32 MB of ranks).  A bucket's slots are ordered by region (kr_part8
// appends region after region, and its cursor snapshots say where each region's slots begin), so the workgroup of
// region r reads, bucket by bucket, the run of pairs that belongs to it -- consecutive addresses -- drops the ranks
// at (home index mod region) in LDS, and then streams over its 2^ib coefficients:
//   delta_j = -((2 rank_j - n - 1) sign g - f n sign) / g^2   (kb_delta_home's expression; Dout may be Yin).
// Exact zeros have no pair and need none.  grid (n >> ib, B), RK_KT threads, (4 << ib) bytes of dynamic LDS.
__global__ __launch_bounds__(RK_KT) void kr_delta_regions(const double* Yin, const unsigned long long* __restrict__ pairs,
                                                          const unsigned short* __restrict__ snap, const unsigned* __restrict__ meta_all,
                                                          size_t n, int ib, const double* __restrict__ fg, double* Dout, const unsigned* flag) {
    extern __shared__ __attribute__((aligned(16))) unsigned rk_home[];
    __shared__ unsigned run0[RK_NBMAX];              // per bucket: where the region's run of pairs starts, and its length
    __shared__ unsigned short runn[RK_NBMAX];
    if (*flag) return;
    const int r = blockIdx.x, e = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int steps = (int)(n >> ib), nbe = nbmax_even(n);
    const unsigned* meta = meta_all + (size_t)e * meta_words(n);
    const int nb = min((int)meta[1], RK_NBMAX);
    const unsigned short* s0 = snap + ((size_t)e * steps + r) * nbe;
    const unsigned mask = (1u << ib) - 1u;
    const unsigned long long* pe = pairs + (size_t)e * n;
    for (int b = tid; b < nb; b += RK_KT) {          // all buckets' runs at once: their descriptors come from memory
        const unsigned bs = meta_bstart(meta)[b], count = meta_bstart(meta)[b + 1] - bs;
        unsigned lo = 0, hi = 0;
        if (count <= (unsigned)RK_CAP) {             // (an overfull bucket has raised the flag)
            lo = min((unsigned)s0[b], count);
            hi = (r + 1 < steps) ? min((unsigned)s0[nbe + b], count) : count;
            hi = max(hi, lo);
        }
        run0[b] = bs + lo;
        runn[b] = (unsigned short)(hi - lo);
    }
    __syncthreads();
    // a wave takes four buckets at a time, two blocks of 64 pairs of each: eight loads in flight
    constexpr int NW = RK_KT / 64;
    for (int b0 = wave; b0 < nb; b0 += 4 * NW) {
        unsigned long long v[4][2];
        unsigned len[4], base[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int b = b0 + q * NW;
            len[q] = (b < nb) ? (unsigned)runn[b] : 0u;
            base[q] = (b < nb) ? run0[b] : 0u;
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const unsigned j = (unsigned)lane + 64u * (unsigned)h;
                v[q][h] = pe[base[q] + (j < len[q] ? j : 0u)];            // (the address stays inside the example either way)
            }
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
#pragma unroll
            for (int h = 0; h < 2; h++) {
                if ((unsigned)lane + 64u * (unsigned)h < len[q]) rk_home[(unsigned)(v[q][h] >> 32) & mask] = (unsigned)v[q][h];
            }
            for (unsigned j = 128u + (unsigned)lane; j < len[q]; j += 64u) {     // runs beyond 128 pairs
                const unsigned long long w = pe[base[q] + j];
                rk_home[(unsigned)(w >> 32) & mask] = (unsigned)w;
            }
        }
    }
    __syncthreads();
    const size_t off = (size_t)e * n + ((size_t)r << ib);
    const double f = fg[2 * e], g = fg[2 * e + 1], dn = (double)n;
    const double inv_g2 = 1.0 / (g * g);
    // two coefficients per thread and step (regions are whole multiples of 2 RK_KT positions: ib >= RK_IB_MIN)
    static_assert((1 << RK_IB_MIN) % (2 * RK_KT) == 0, "whole steps of two positions per thread");
    auto delta = [&](double y, unsigned rank) {
        const double sg = sign_ref(y);
        const double coef = (y != 0.0) ? 2.0 * (double)rank - dn - 1.0 : 0.0;
        const double df = coef * sg;
        return -(df * g - f * (dn * sg)) * inv_g2;
    };
    for (unsigned j = 2u * (unsigned)tid; j <= mask; j += 2 * RK_KT) {
        const double2 y = *reinterpret_cast<const double2*>(Yin + off + j);
        const uint2 rk2 = *reinterpret_cast<const uint2*>(rk_home + j);
        *reinterpret_cast<double2*>(Dout + off + j) = make_double2(delta(y.x, rk2.x), delta(y.y, rk2.y));
    }
}

// fg[2e] = f, fg[2e+1] = g = n sum |c|, S_e[e] = 1 - f / g.  One wave per example (grid B, 64 threads): lane l
// adds the shares of buckets l, l + 64, ... in order, the lanes are combined by a butterfly -- a fixed order.
__global__ __launch_bounds__(64) void kr_finish(unsigned* meta_all, size_t n, int B, double* fg, double* S_e) {
    const int e = blockIdx.x, lane = threadIdx.x;
    if (e >= B) return;
    const int nbmax = nbmax_for(n);
    unsigned* meta = meta_all + (size_t)e * meta_words(n);
    const int nb = (int)meta[1];
    const double* bp = meta_bpart(meta, nbmax);
    const unsigned* bs = meta_bstart(meta);
    double f = 0, a = 0;
    for (int b = lane; b < nb; b += 64) {
        if (bs[b + 1] != bs[b]) { f += bp[2 * b]; a += bp[2 * b + 1]; }
    }
    f = wave_sum(f);
    a = wave_sum(a);
    if (lane == 0) {
        const double g = (double)n * a;
        fg[2 * e] = f;
        fg[2 * e + 1] = g;
        if (S_e) S_e[e] = 1.0 - f / g;
    }
}

}  // namespace rk
}  // namespace wn
