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
// phase.  Its ranks come from a bitonic sort of the wave's 1024 keys, 16 per lane, in the manner of the
// 64x64 kernel (wn_wave64.h): a key is |c| with its low 10 mantissa bits replaced by the coefficient's
// index, so v_min_f64 / v_max_f64 carry the index along, keys are unique and equal magnitudes come out in
// index order; compare-exchanges run on register bits only, and three register layouts (index bits 0-3,
// 4-7 or 6-9 in the register number) exchanged through LDS bring every index bit into the registers.  An
// example whose keys collide in their upper 54 bits (so that the order by index may differ from the order
// by exact magnitude), or with magnitudes outside the range where the fast sign is exact, takes the general
// Gini phase of the workgroup-per-example kernels instead, run by this one wave.
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
// The wave rotated by one lane: lane i receives lane i - 1 (ror) or lane i + 1 (rol), modulo 64.  A DPP move in the
// vector pipeline: no LDS instruction, unlike ds_bpermute, and the periodic boundary comes with it.  Every lane of
// the wave must be active.  (Here and below: a zero for lanes without a source AND bound_ctrl, which says the same
// thing twice -- in that form the compiler knows the destination's old value is dead and does not emit a move that
// sets it to zero in front of every DPP move.)
__device__ __forceinline__ double w1d_ror1(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x13C, 0xf, 0xf, true);   // wave_ror:1
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x13C, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double w1d_rol1(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x134, 0xf, 0xf, true);   // wave_rol:1
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x134, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
// v of lane (lane - D) mod 64: neighbours at distance 1 and 2 by rotations, the others through ds_bpermute
template <int D>
__device__ __forceinline__ double w1d_from(double v, int lane) {
#ifndef WN_NO_DPP_ROTATE
    if constexpr (D == 1) return w1d_ror1(v);
    else if constexpr (D == -1) return w1d_rol1(v);
    else if constexpr (D == 2) return w1d_ror1(w1d_ror1(v));
    else if constexpr (D == -2) return w1d_rol1(w1d_rol1(v));
    else
#endif
    return w1d_lane(v, lane + ((64 - D) & 63));
}

// A value that is the same in every lane, moved to scalar registers.
__device__ __forceinline__ double w1d_uniform(double v) {
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)),
                            __builtin_amdgcn_readfirstlane(__double2loint(v)));
}

// Rotation inside the rows of 16 lanes: lane i receives lane (i - N) mod 16 of its row (DPP row_ror:N).
template <int N>
__device__ __forceinline__ double w1d_row_ror(double v) {
    static_assert(N >= 1 && N <= 15, "row_ror takes 1..15");
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x120 + N, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x120 + N, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
// Permutation inside every group of four lanes: lane 4 g + i receives lane 4 g + {A, B, C, D}[i] (DPP quad_perm).
template <int A, int B, int C, int D>
__device__ __forceinline__ double w1d_quad(double v) {
    constexpr int ctrl = A | (B << 2) | (C << 4) | (D << 6);
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), ctrl, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), ctrl, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
// out[t] = v of lane (lane + S0 + t), t < CNT, for a value that is periodic in the lane number with period P (a
// power of two up to 64).  Periods up to 16 repeat inside every row of 16 lanes, so one row rotation reaches any
// distance; the longer periods are followed one lane at a time by rotations of the whole wave, outwards from
// distance 0.  DPP moves only: nothing goes through the LDS crossbar.  Every lane of the wave must be active.
template <int S0, int CNT, int P>
__device__ __forceinline__ void w1d_shifts(double v, double* out) {
    static_assert(P >= 1 && P <= 64 && (P & (P - 1)) == 0, "period: a power of two up to 64");
    if constexpr (P <= 16) {
#pragma unroll
        for (int t = 0; t < CNT; t++) {
            const int s = (((S0 + t) % P) + P) % P;          // lane + s, 0 <= s < P
            switch (s) {                                      // (a compile-time value once unrolled)
                case 0: out[t] = v; break;
                case 1: out[t] = w1d_row_ror<15>(v); break;
                case 2: out[t] = w1d_row_ror<14>(v); break;
                case 3: out[t] = w1d_row_ror<13>(v); break;
                case 4: out[t] = w1d_row_ror<12>(v); break;
                case 5: out[t] = w1d_row_ror<11>(v); break;
                case 6: out[t] = w1d_row_ror<10>(v); break;
                case 7: out[t] = w1d_row_ror<9>(v); break;
                case 8: out[t] = w1d_row_ror<8>(v); break;
                case 9: out[t] = w1d_row_ror<7>(v); break;
                case 10: out[t] = w1d_row_ror<6>(v); break;
                case 11: out[t] = w1d_row_ror<5>(v); break;
                case 12: out[t] = w1d_row_ror<4>(v); break;
                case 13: out[t] = w1d_row_ror<3>(v); break;
                case 14: out[t] = w1d_row_ror<2>(v); break;
                default: out[t] = w1d_row_ror<1>(v); break;
            }
        }
    } else {
        static_assert(S0 <= 0 && S0 + CNT - 1 >= 0 && -S0 < P / 2 && S0 + CNT - 1 <= P / 2, "distances around 0, below half a period");
        out[-S0] = v;
        double cur = v;
#pragma unroll
        for (int s = 1; s <= S0 + CNT - 1; s++) {
            cur = w1d_rol1(cur);                              // lane + s
            out[s - S0] = cur;
        }
        cur = v;
#pragma unroll
        for (int s = -1; s >= S0; s--) {
            cur = w1d_ror1(cur);                              // lane - |s|
            out[s - S0] = cur;
        }
    }
}
// floor(c / 2) for either sign
__host__ __device__ constexpr int w1d_floor_half(int c) { return (c >= 0) ? c / 2 : -((1 - c) / 2); }

// The taps' inputs of one level below 64 values, v[j] = in[(2 k + NP/2 - j) mod 2^(LG+1)] for the lane's output
// k = lane mod 2^LG: `in` holds index lane mod 2^(LG+1).  The even and the odd inputs are gathered once each
// (E[l] = in[2 l], O[l] = in[2 l + 1], two ds_bpermute pairs); with them a tap is a neighbour's value,
// in[2 k + c] = E[k + c / 2] or O[k + (c - 1) / 2], reached by DPP moves (w1d_shifts) -- against one gather per tap.
template <int NP, int LG>
__device__ __forceinline__ void w1d_small_taps(double in, double* v, int lane) {
    constexpr int P = 1 << LG;
    double E, O;
    if constexpr (LG == 0) {             // two inputs, in lanes 0 and 1 of every four
        E = w1d_quad<0, 0, 0, 0>(in);
        O = w1d_quad<1, 1, 1, 1>(in);
    } else if constexpr (LG == 1) {      // four inputs, one per lane of every four
        E = w1d_quad<0, 2, 0, 2>(in);
        O = w1d_quad<1, 3, 1, 3>(in);
    } else {
        const int src = 2 * (lane & (P - 1));
        E = w1d_lane(in, src);
        O = w1d_lane(in, src + 1);
    }
    // c = NP/2 - j runs from NP/2 down to NP/2 - NP + 1
    constexpr int CHI = NP / 2, CLO = NP / 2 - NP + 1;
    constexpr int CEHI = (CHI % 2 == 0) ? CHI : CHI - 1, CELO = (CLO % 2 == 0) ? CLO : CLO + 1;     // even c
    constexpr int COHI = (CHI % 2 != 0) ? CHI : CHI - 1, COLO = (CLO % 2 != 0) ? CLO : CLO + 1;     // odd c
    constexpr int SE0 = w1d_floor_half(CELO), SE1 = w1d_floor_half(CEHI), SO0 = w1d_floor_half(COLO), SO1 = w1d_floor_half(COHI);
    double Es[SE1 - SE0 + 1], Os[SO1 - SO0 + 1];
    w1d_shifts<SE0, SE1 - SE0 + 1, P>(E, Es);
    w1d_shifts<SO0, SO1 - SO0 + 1, P>(O, Os);
#pragma unroll
    for (int j = 0; j < NP; j++) {
        const int c = NP / 2 - j;
        v[j] = (c % 2 == 0) ? Es[w1d_floor_half(c) - SE0] : Os[w1d_floor_half(c) - SO0];
    }
}

// The lane's M values (own[M]) with HL values of the lanes before it and HR of the lanes after it in
// ext[HL + M + HR]: ext[HL + i] = own[i]; ext[HL - 1 - t] and ext[HL + M + t] continue the sequence on either side.
template <int M, int HL, int HR>
__device__ __forceinline__ void w1d_extend(const double* own, double* ext, int lane) {
#pragma unroll
    for (int i = 0; i < M; i++) ext[HL + i] = own[i];
#ifndef WN_HALOS_BY_GATHER
    // Every register is followed outwards one lane at a time (wave rotations, DPP moves in the vector pipeline):
    // distance d is one more rotation of what distance d - 1 brought in.
    double cur[M];
#pragma unroll
    for (int t = 0; t < HL; t++) {
        const int d = 1 + t / M, reg = M - 1 - (t % M);
        cur[reg] = w1d_ror1(d == 1 ? own[reg] : cur[reg]);
        ext[HL - 1 - t] = cur[reg];
    }
#pragma unroll
    for (int t = 0; t < HR; t++) {
        const int d = 1 + t / M, reg = t % M;
        cur[reg] = w1d_rol1(d == 1 ? own[reg] : cur[reg]);
        ext[HL + M + t] = cur[reg];
    }
    (void)lane;
#else
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
#endif
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
    double v[NP];   // the inputs first: they are independent, the sums are chains
#ifdef WN_SMALL_LEVELS_BY_GATHER
    constexpr int half = 1 << LG;
    // source index 2k + N/2 - j modulo the length; any lane congruent to it holds the value
    const int base = 2 * (lane & (half - 1)) + 64 + NP / 2 - (NP - 1);
#pragma unroll
    for (int j = 0; j < NP; j++) v[j] = w1d_lane(in, base + (NP - 1 - j));
#else
    w1d_small_taps<NP, LG>(in, v, lane);
#endif
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
    {
        double v[NP];
#ifdef WN_SMALL_LEVELS_BY_GATHER
        const int gbase = 2 * (lane & (len - 1)) + 64 + NP / 2 - (NP - 1);
#pragma unroll
        for (int u = 0; u < NP; u++) v[u] = w1d_lane(act, gbase + (NP - 1 - u));
#else
        w1d_small_taps<NP, LG>(act, v, lane);
#endif
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
    double vl[NP / 2 + 1], vh[NP / 2 + 1];
#ifdef WN_SMALL_LEVELS_BY_GATHER
    const int pbase = ((lane & (alen - 1)) >> 1) + 64 + CMIN;
#pragma unroll
    for (int t = 0; t <= NP / 2; t++) {
        vl[t] = w1d_lane(dlo, pbase + t);
        vh[t] = w1d_lane(dhi, pbase + t);
    }
#else
    // Every lane works out both results of ITS error index p = lane mod 2^LG from its neighbours' errors (DPP moves);
    // the lanes of the level above then fetch theirs, index m = lane mod 2^(LG+1), from lane m / 2: one gather each
    // for the even and the odd result, against one per tap and array.
    w1d_shifts<CMIN, NP / 2 + 1, len>(dlo, vl);
    w1d_shifts<CMIN, NP / 2 + 1, len>(dhi, vh);
#endif
    __builtin_amdgcn_sched_barrier(0);
    double se = 0, so = 0;
#pragma unroll
    for (int i = 0; i < NP / 2; i++) {
        se = __builtin_fma(a[JE + 2 * i], vl[i + CE - CMIN], se);
        se = __builtin_fma(b[JE + 2 * i], vh[i + CE - CMIN], se);
        so = __builtin_fma(a[JO + 2 * i], vl[i + CO - CMIN], so);
        so = __builtin_fma(b[JO + 2 * i], vh[i + CO - CMIN], so);
    }
#ifndef WN_SMALL_LEVELS_BY_GATHER
    if constexpr (LG == 1) {             // two error indices: lanes 0, 1 of every four hold them
        se = w1d_quad<0, 0, 1, 1>(se);
        so = w1d_quad<0, 0, 1, 1>(so);
    } else if constexpr (LG > 1) {
        const int from = (lane & (alen - 1)) >> 1;
        se = w1d_lane(se, from);
        so = w1d_lane(so, from);
    }                                    // (LG == 0: one error index, the same results in every lane)
#endif
    const double s = (lane & 1) ? so : se;
    return s;
}


// ---- Gini ranks of 1024 coefficients in one wave -------------------------------------------------------
__device__ __forceinline__ void w1d_cmpx(double& lo, double& hi) {   // min -> lo, max -> hi
    double mn, mx;
    asm volatile("v_min_f64 %0, %2, %3\n\tv_max_f64 %1, %2, %3" : "=&v"(mn), "=&v"(mx) : "v"(lo), "v"(hi));
    lo = mn;
    hi = mx;
}
__device__ __forceinline__ void w1d_flip(double& k, unsigned mask) {
    k = __hiloint2double(__double2hiint(k) ^ (int)mask, __double2loint(k));
}
// Position of sort index e in the exchange buffer: one double of padding per 16, which makes the three
// layouts' accesses conflict-free.
__device__ __forceinline__ constexpr int w1d_pos(int e) { return e + (e >> 4); }
constexpr int W1D_SORT_DOUBLES = 1024 + 64;

// Sort index of register r in lane `lane`: layout A (register = index bits 0-3), M (bits 4-7), B (bits 6-9).
constexpr int W1D_LAYOUT_A = 0, W1D_LAYOUT_M = 4, W1D_LAYOUT_B = 6;
template <int LAYOUT>
__device__ __forceinline__ int w1d_index(int lane, int r) {
    if constexpr (LAYOUT == W1D_LAYOUT_A) return 16 * lane + r;
    else if constexpr (LAYOUT == W1D_LAYOUT_M) return 256 * (lane >> 4) + 16 * r + (lane & 15);
    else return 64 * r + lane;
}
// No fence between the writes and the reads: both are accesses to doubles that may alias as far as the compiler
// can tell, so it keeps their program order, and the LDS executes a wave's operations in order.  (A fence here
// also keeps the key array from being held in registers.)
template <int FROM, int TO>
__device__ __forceinline__ void w1d_exchange(double (&k)[16], double* buf, int lane) {
#pragma unroll
    for (int r = 0; r < 16; r++) buf[w1d_pos(w1d_index<FROM>(lane, r))] = k[r];
#pragma unroll
    for (int r = 0; r < 16; r++) k[r] = buf[w1d_pos(w1d_index<TO>(lane, r))];
}
// compare-exchanges on index bits HI .. LO, which are register bits HI - BASE .. LO - BASE in the current layout
template <int BASE, int HI, int LO>
__device__ __forceinline__ void w1d_substages(double (&k)[16]) {
    constexpr int RB = HI - BASE;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        if (!(r & (1 << RB))) w1d_cmpx(k[r], k[r | (1 << RB)]);
    }
    if constexpr (HI > LO) w1d_substages<BASE, HI - 1, LO>(k);
}
// Merge step KK of the bitonic sort (blocks of 2^KK); on entry and exit the keys are in layout A.  A key is held
// negated while its block sorts descending, i.e. while bit KK of its index is set, so every compare-exchange
// is minimum -> lower index.
template <int KK>
__device__ __forceinline__ void w1d_merge(double (&k)[16], double* buf, int lane) {
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int e = 16 * lane + r;
        // negated so far: where bit KK - 1 is set (nowhere before the first merge); from now on: where bit KK is set
        // (bit 10 is 0: the last merge ends with every key positive)
        // (an exclusive-or with a mask rather than a branch on the lane number)
        const int was = (KK > 1) ? (e >> (KK - 1)) & 1 : 0, now = (e >> KK) & 1;
        w1d_flip(k[r], (unsigned)(was ^ now) << 31);
    }
    if constexpr (KK <= 4) {
        w1d_substages<W1D_LAYOUT_A, KK - 1, 0>(k);
    } else if constexpr (KK <= 8) {
        w1d_exchange<W1D_LAYOUT_A, W1D_LAYOUT_M>(k, buf, lane);
        w1d_substages<W1D_LAYOUT_M, KK - 1, 4>(k);
        w1d_exchange<W1D_LAYOUT_M, W1D_LAYOUT_A>(k, buf, lane);
        w1d_substages<W1D_LAYOUT_A, 3, 0>(k);
    } else {
        w1d_exchange<W1D_LAYOUT_A, W1D_LAYOUT_B>(k, buf, lane);
        w1d_substages<W1D_LAYOUT_B, KK - 1, 8>(k);
        w1d_exchange<W1D_LAYOUT_B, W1D_LAYOUT_M>(k, buf, lane);
        w1d_substages<W1D_LAYOUT_M, 7, 4>(k);
        w1d_exchange<W1D_LAYOUT_M, W1D_LAYOUT_A>(k, buf, lane);
        w1d_substages<W1D_LAYOUT_A, 3, 0>(k);
    }
    if constexpr (KK < 10) w1d_merge<KK + 1>(k, buf, lane);
}

// Coefficient index of the lane's slot s: slots 0-7 the top level's high-pass values (512 + 8 lane + s), 8-11 the
// next level (256 + 4 lane + ..), 12-13 (128 + 2 lane + ..), 14 (64 + lane), 15 the levels below (lane).
__device__ __forceinline__ int w1d_coef_index(int lane, int s) {
    return s < 8 ? 512 + 8 * lane + s : s < 12 ? 256 + 4 * lane + (s - 8) : s < 14 ? 128 + 2 * lane + (s - 12) : s == 14 ? 64 + lane : lane;
}

// Gini term and gradient (src/CostFunctions.cxx:6-37, 49-105) from the coefficients in registers, y[s] the
// coefficient of index w1d_coef_index(lane, s).  Leaves the errors in bufY[0 .. 1024) in coefficient order and
// returns true -- or returns false without having touched y when the example needs the general path.
// buf: W1D_SORT_DOUBLES doubles (may be bufY itself); rk: 1024 16-bit words.
__device__ __forceinline__ bool w1d_gini_fast(const double (&y)[16], double* buf, unsigned short* rk, double* bufY, int lane,
                                              double* S_out) {
    constexpr int n = 1024;
    // the index-in-the-mantissa keys need magnitudes that are zero or normal numbers, the fast sign magnitudes
    // for which x / sqrt(x * x) is exactly +-1
    bool odd = false;
#pragma unroll
    for (int s = 0; s < 16; s++) {
        const double ay = __builtin_fabs(y[s]);
        odd |= !((ay == 0.0) | ((ay >= 0x1p-500) & (ay <= 0x1p500)));
    }
    if (__any(odd)) return false;
    double k[16];
#pragma unroll
    for (int s = 0; s < 16; s++) {
        const double ay = __builtin_fabs(y[s]);
        const unsigned idx = (unsigned)w1d_coef_index(lane, s);
        k[s] = __hiloint2double(__double2hiint(ay), (int)((((unsigned)__double2loint(ay)) & ~0x3FFu) | idx));
    }
    w1d_merge<1>(k, buf, lane);
    // sorted position p = 16 lane + r; keys that agree in their upper 54 bits (zeros aside) would have to be ordered
    // by their exact low bits: leave those examples to the general path
    bool same = false;
    {
        const double prev_last = __shfl_up(k[15], 1, 64);
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const unsigned long long kb = (unsigned long long)__double_as_longlong(k[r]);
            const unsigned long long pb = (unsigned long long)__double_as_longlong((r > 0) ? k[r - 1] : prev_last);
            same |= ((pb ^ kb) < 1024ull) & (kb >= 1024ull) & ((r > 0) | (lane > 0));
        }
    }
    if (__any(same)) return false;
#pragma unroll
    for (int r = 0; r < 16; r++) rk[(unsigned)__double2loint(k[r]) & 0x3FFu] = (unsigned short)(16 * lane + r + 1);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // the ranks of the lane's own coefficients: contiguous per level
    unsigned rank[16];
    {
        const uint4 r8 = *reinterpret_cast<const uint4*>(rk + 512 + 8 * lane);
        const uint2 r4 = *reinterpret_cast<const uint2*>(rk + 256 + 4 * lane);
        const unsigned r2 = *reinterpret_cast<const unsigned*>(rk + 128 + 2 * lane);
        const unsigned w8[4] = {r8.x, r8.y, r8.z, r8.w};
#pragma unroll
        for (int s = 0; s < 8; s++) rank[s] = (w8[s >> 1] >> (16 * (s & 1))) & 0xFFFFu;
        rank[8] = r4.x & 0xFFFFu; rank[9] = r4.x >> 16; rank[10] = r4.y & 0xFFFFu; rank[11] = r4.y >> 16;
        rank[12] = r2 & 0xFFFFu; rank[13] = r2 >> 16;
        rank[14] = rk[64 + lane];
        rank[15] = rk[lane];
    }
    double f = 0, sabs = 0;
#pragma unroll
    for (int s = 0; s < 16; s++) {
        const double coef = 2.0 * (double)rank[s] - (double)n - 1.0;
        const double ay = __builtin_fabs(y[s]);
        f = __builtin_fma(coef, ay, f);
        sabs += ay;
    }
    f = wave_allsum(f, lane);
    sabs = wave_allsum(sabs, lane);
    const double g = (double)n * sabs;
    const double inv_g2 = 1.0 / (g * g);
    double d[16];
#pragma unroll
    for (int s = 0; s < 16; s++) {
        const double coef = 2.0 * (double)rank[s] - (double)n - 1.0;
        const double sg = (y[s] == 0.0) ? 0.0 : __builtin_copysign(1.0, y[s]);   // == sign_ref in this range
        d[s] = -((coef * sg) * g - f * ((double)n * sg)) * inv_g2;
    }
    // (the exchange buffer may be bufY: the sort is over)
    {
        double2* t8 = reinterpret_cast<double2*>(bufY + 512 + 8 * lane);
#pragma unroll
        for (int m = 0; m < 4; m++) t8[m] = make_double2(d[2 * m], d[2 * m + 1]);
        double2* t4 = reinterpret_cast<double2*>(bufY + 256 + 4 * lane);
        t4[0] = make_double2(d[8], d[9]);
        t4[1] = make_double2(d[10], d[11]);
        *reinterpret_cast<double2*>(bufY + 128 + 2 * lane) = make_double2(d[12], d[13]);
        bufY[64 + lane] = d[14];
        bufY[lane] = d[15];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    *S_out = 1.0 - f / g;
    return true;
}

// The general Gini phase as a call of its own: inlined, its registers would add to those the fast path holds.
// (WAVE: this wave is one of four in its workgroup, each with an example of its own)
__device__ __attribute__((noinline)) double w1d_gini_general(double* bufY, double* red, int* flag, int lane) {
    return gini_phase<64, 4, true, true>(bufY, 1024, red, flag, lane);
}

// doubles of LDS per wave: sort exchange buffer / coefficients and errors, ranks, scratch of the general Gini phase,
// and with 16 taps the parked pyramid
template <int NP>
constexpr int w1d_lds_doubles() { return W1D_SORT_DOUBLES + 256 + 8 + (NP == 16 ? 15 * 64 : 0); }

// ---- kernel -------------------------------------------------------------------------------------------
// 256 threads = 4 waves = 4 examples in flight per workgroup, each wave with an LDS slice of its own and no barrier
// between them until the end; grid-stride over the batch.  (Four waves per workgroup rather than one so that the
// partial sums leave as one slot per workgroup and the workgroup that finishes last can end the step, wn_step.h.)
#ifndef W1D_WAVES
#define W1D_WAVES (NP > 16 ? 1 : 2)
#endif
template <int NP>
__global__ __launch_bounds__(256, W1D_WAVES) void k_train1d1024(FusedArgs A) {
    constexpr int n = 1024;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double* bufY = smem + (size_t)wave * w1d_lds_doubles<NP>();   // sort exchange buffer; 1024 coefficients / errors in coefficient order
    unsigned short* rk = reinterpret_cast<unsigned short*>(bufY + W1D_SORT_DOUBLES);   // ranks by coefficient index
    double* red = bufY + W1D_SORT_DOUBLES + 256;    // scratch of the general Gini phase
    int* flag = reinterpret_cast<int*>(red + 4);
    // 16 taps: the low-pass pyramid waits in LDS ([15][64] doubles) from the end of the forward pass until the backward
    // pass wants it, and the kernel then runs two waves per SIMD with few spills instead of one (measured: 7 % faster)
    constexpr bool PARK = (NP == 16);
    double* park = red + 8;
    const int lane0 = threadIdx.x & 63;
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
#ifndef W1D_B_LOADED
    // The high-pass taps are the low-pass taps reversed with alternating signs (src/HighpassOperator.cxx:14-18).  The
    // kernel does not hold the stored b but writes b'[j] = (-1)^j a[NP - 1 - j], a negated operand where the sign is
    // minus: half the registers for taps (two filters of 16 or 32 taps do not fit the scalar registers, and the
    // compiler then fetches spilled taps back lane by lane in front of the multiply-adds).  The stored b carries the
    // parity of the TRUE tap index: b = sg b' with sg = (-1)^(padding in front of the filter).  Nothing is corrected
    // for that on the way: the high-pass coefficients come out times sg, the Gini term does not see the sign of a
    // coefficient and its derivative is odd in it, so the errors on them are times sg as well, the errors propagated
    // to the low-pass side are products of the two, and the high-pass window sums gH are times sg -- which the sign
    // table takes back below.  A negation is exact, so every number that leaves the kernel is the same bit for bit.
#pragma unroll
    for (int j = 0; j < NP; j++) b[j] = (j & 1) ? -a[NP - 1 - j] : a[NP - 1 - j];
    const double sg = (dpad & 1) ? -1.0 : 1.0;
#else
    const double sg = 1.0;
#endif

    double accG = 0, accS = 0, accBad = 0;
    for (int e = blockIdx.x * 4 + wave; e < A.B; e += gridDim.x * 4) {
        // The lane number as a value the compiler cannot see through: the LDS and cross-lane addresses derived
        // from it are then computed where they are used, not all held in registers across the whole loop.
        int lane = lane0;
        asm volatile("" : "+v"(lane));
        // ---- forward
        double x[16], lo9[8], lo8[4], lo7[2], lo6[1], hi[8];
        {
            const double2* src = reinterpret_cast<const double2*>(A.X + (size_t)e * n + 16 * lane);
#pragma unroll
            for (int i = 0; i < 8; i++) { const double2 t = src[i]; x[2 * i] = t.x; x[2 * i + 1] = t.y; }
        }
        group_sync<true>();   // the previous example's errors have been read
        double y[16];      // the lane's coefficients, slot s = coefficient w1d_coef_index(lane, s)
        w1d_fwd_block<NP, 16>(x, lo9, hi, a, b, lane);
#pragma unroll
        for (int m = 0; m < 8; m++) y[m] = hi[m];
        {
            double h8[4];
            w1d_fwd_block<NP, 8>(lo9, lo8, h8, a, b, lane);
#pragma unroll
            for (int m = 0; m < 4; m++) y[8 + m] = h8[m];
            double h7[2];
            w1d_fwd_block<NP, 4>(lo8, lo7, h7, a, b, lane);
            y[12] = h7[0];
            y[13] = h7[1];
            double h6[1];
            w1d_fwd_block<NP, 2>(lo7, lo6, h6, a, b, lane);
            y[14] = h6[0];
        }
        // levels below 64 values, replicated: los[i] = low-pass level i at index lane mod 2^i
        double los[6], his, small = 0;
        w1d_fwd_small<NP, 5>(lo6[0], los[5], his, a, b, lane);
        if (lane >= 32) small = his;                       // coefficient 32 + k sits in lane 32 + k
        w1d_fwd_small<NP, 4>(los[5], los[4], his, a, b, lane);
        if (lane >= 16 && lane < 32) small = his;
        w1d_fwd_small<NP, 3>(los[4], los[3], his, a, b, lane);
        if (lane >= 8 && lane < 16) small = his;
        w1d_fwd_small<NP, 2>(los[3], los[2], his, a, b, lane);
        if (lane >= 4 && lane < 8) small = his;
        w1d_fwd_small<NP, 1>(los[2], los[1], his, a, b, lane);
        if (lane >= 2 && lane < 4) small = his;
        w1d_fwd_small<NP, 0>(los[1], los[0], his, a, b, lane);
        if (lane == 1) small = his;
        if (lane == 0) small = los[0];
        y[15] = small;

        if constexpr (PARK) {
#pragma unroll
            for (int m = 0; m < 8; m++) park[m * 64 + lane] = lo9[m];
#pragma unroll
            for (int m = 0; m < 4; m++) park[(8 + m) * 64 + lane] = lo8[m];
            park[12 * 64 + lane] = lo7[0];
            park[13 * 64 + lane] = lo7[1];
            park[14 * 64 + lane] = lo6[0];
            // (the register copies end here as far as the compiler can tell)
#pragma unroll
            for (int m = 0; m < 8; m++) asm volatile("" : "=v"(lo9[m]));
#pragma unroll
            for (int m = 0; m < 4; m++) asm volatile("" : "=v"(lo8[m]));
            asm volatile("" : "=v"(lo7[0]));
            asm volatile("" : "=v"(lo7[1]));
            asm volatile("" : "=v"(lo6[0]));
        }
        // ---- Gini term and its gradient (src/CostFunctions.cxx:6-37, 49-105): errors to bufY in coefficient order
        double S;
        if (!w1d_gini_fast(y, bufY, rk, bufY, lane, &S)) {
#pragma unroll
            for (int m = 0; m < 8; m++) bufY[512 + 8 * lane + m] = y[m];
#pragma unroll
            for (int m = 0; m < 4; m++) bufY[256 + 4 * lane + m] = y[8 + m];
            bufY[128 + 2 * lane] = y[12];
            bufY[128 + 2 * lane + 1] = y[13];
            bufY[64 + lane] = y[14];
            bufY[lane] = y[15];
            group_sync<true>();
            S = w1d_gini_general(bufY, red, flag, lane);
        }

        // ---- backward
        double gL[NP], gH[NP];
#pragma unroll
        for (int u = 0; u < NP; u++) { gL[u] = 0; gH[u] = 0; }
        double d = bufY[0];                                                                      // errors on lo_0
        d = w1d_bwd_small<NP, 0>(d, bufY[1], los[1], a, b, gL, gH, lane);                        // -> 2 values
        d = w1d_bwd_small<NP, 1>(d, bufY[2 + (lane & 1)], los[2], a, b, gL, gH, lane);           // -> 4
        d = w1d_bwd_small<NP, 2>(d, bufY[4 + (lane & 3)], los[3], a, b, gL, gH, lane);           // -> 8
        d = w1d_bwd_small<NP, 3>(d, bufY[8 + (lane & 7)], los[4], a, b, gL, gH, lane);           // -> 16
        d = w1d_bwd_small<NP, 4>(d, bufY[16 + (lane & 15)], los[5], a, b, gL, gH, lane);         // -> 32
        if constexpr (PARK) lo6[0] = park[14 * 64 + lane];
        d = w1d_bwd_small<NP, 5>(d, bufY[32 + (lane & 31)], lo6[0], a, b, gL, gH, lane);         // -> 64: one per lane
        double d1[1] = {d}, h1[1] = {bufY[64 + lane]}, d2[2], h2[2], d4[4], h4[4], d8[8], h8[8], unused[16];
        if constexpr (PARK) { lo7[0] = park[12 * 64 + lane]; lo7[1] = park[13 * 64 + lane]; }
        w1d_bwd_block<NP, 1, true>(d1, h1, lo7, d2, a, b, gL, gH, lane);
        h2[0] = bufY[128 + 2 * lane]; h2[1] = bufY[128 + 2 * lane + 1];
        if constexpr (PARK) {
#pragma unroll
            for (int m = 0; m < 4; m++) lo8[m] = park[(8 + m) * 64 + lane];
        }
        w1d_bwd_block<NP, 2, true>(d2, h2, lo8, d4, a, b, gL, gH, lane);
#pragma unroll
        for (int m = 0; m < 4; m++) h4[m] = bufY[256 + 4 * lane + m];
        if constexpr (PARK) {
#pragma unroll
            for (int m = 0; m < 8; m++) lo9[m] = park[m * 64 + lane];
        }
        w1d_bwd_block<NP, 4, true>(d4, h4, lo9, d8, a, b, gL, gH, lane);
        // the signal is read again for the last level, as late as possible, rather than held in registers until then
        double xb[16];
        {
            const double* again = A.X + (size_t)e * n + 16 * lane;
            asm volatile("" : "+v"(again));
            const double2* src = reinterpret_cast<const double2*>(again);
#pragma unroll
            for (int i = 0; i < 8; i++) { const double2 t = src[i]; xb[2 * i] = t.x; xb[2 * i + 1] = t.y; }
        }
#pragma unroll
        for (int m = 0; m < 8; m++) h8[m] = bufY[512 + 8 * lane + m];
        w1d_bwd_block<NP, 8, false>(d8, h8, xb, unused, a, b, gL, gH, lane);

        // tap k' collects gL[k'] and sH[k'] gH[NP-1-k'], summed over the 64 lanes
        double mine;
        {
            double c[NP];
#pragma unroll
            for (int kp = 0; kp < NP; kp++) c[kp] = filt->vL[kp] * gL[kp] + (sg * filt->sH[kp]) * gH[NP - 1 - kp];
#ifndef W1D_PLAIN_BUTTERFLY
            mine = wave_scatter_sums<NP>(c, lane);       // the total of tap lane mod NP
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
    // The four waves' running sums are added in wave order and leave as one slot per workgroup; the workgroup
    // that finishes last ends the step (step_tail).
    double bad = accBad;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) bad += __shfl_xor(bad, off, 64);
    __syncthreads();                                    // every wave is done with its slice
    double* wsum = smem + (size_t)wave * PSTRIDE;       // [4][PSTRIDE]
    const int k = lane0 - dpad;
    if (lane0 < NP && k >= 0 && k < N) wsum[k] = accG;
    if (lane0 == 0) { wsum[N] = accS; wsum[N + 1] = bad; }
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
