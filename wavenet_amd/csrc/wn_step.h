// The end of a training step inside the training kernel (k_train64: at most 256 partial-sum slots, which one
// workgroup adds up in a few microseconds): the workgroup that finishes last adds up the
// per-workgroup partial sums and, on a single GPU, applies the filter update, so that a whole step
// (src/Wavenet.cxx:184-242 for a batch, then flushBatchQueue_ / update_ / setFilter, :717-765, 8-36) is
// ONE kernel launch and consecutive batches queue up on the stream with no host code in between.
//
// Hand-over between workgroups: a workgroup stores its slot write-through (agent-scope stores), its waves wait
// for their stores to complete and meet at a barrier, one lane takes a ticket from a device-wide
// counter; the workgroup whose ticket is the last one acquires at agent scope and only then reads
// the other workgroups' slots.  Nothing ever waits for another workgroup, so no placement or
// residency assumption is made.
#pragma once

#include "wn_costfn.h"
#include "wn_device.h"

namespace wn {

struct UpdateParams {
    double Bf;            // float(batch size) as a double        (src/Wavenet.cxx:727, 733)
    double lambda_sum;    // sum over the batch's examples of lambda
    double alpha;
    double inertia_eff;   // the caller's evaluation of src/Wavenet.cxx:757
    int wavelet;
    int cap;              // entries of the log rings
    double* logF;         // [cap][MAXTAPS] filter after each update
    double* logC;         // [cap] batch-mean cost of each update
};

struct StepTail {
    unsigned* ticket;     // null: the kernel only writes its partial sums
    double* sums;         // N + 2 sums of the batch
    FilterDev* filt;      // the filter state to update (mode 2)
    int mode;             // 1: sums only (the caller all-reduces them and updates), 2: sums and update
    UpdateParams P;
};

struct UpdateShared {
    double a[MAXTAPS], b[MAXTAPS], prefA[MAXTAPS + 1], prefB[MAXTAPS + 1], anew[MAXTAPS], pnew[MAXTAPS], cost;
    double apad[3 * MAXTAPS];   // a with a margin of zeros on either side
    float devA[MAXTAPS + 1], devB[MAXTAPS + 1];
    int bad;
};

// LDS the tail needs at the start of the kernel's dynamic shared memory (every training kernel is done
// with its own use of it by then)
// layout: flag | the sums | UpdateShared | 64 partial results per quantity
constexpr int TAIL_OFF_SUMS = 16, TAIL_OFF_SHARED = TAIL_OFF_SUMS + (MAXTAPS + 2) * 8;
constexpr int TAIL_OFF_LSUM = (TAIL_OFF_SHARED + (int)sizeof(UpdateShared) + 15) & ~15;
constexpr int TAIL_LDS_BYTES = TAIL_OFF_LSUM + (MAXTAPS + 2) * 64 * 8;   // (MAXTAPS + 2 = the doubles of a partial-sum slot)

// The loops of the regularisation terms (wn_costfn.h) for operands in LDS.  This code runs once per step in a
// single wave, so it pays both for LDS round trips that wait on each other and for every instruction it has
// (fetched cold): the loops are kept as loops, unrolled by four with the loads of an unrolled group issued
// together, and the additions keep the order (and so the rounding) of the plain loops.

// lag_prefactor and lag_deviation_sq_float of lag m from one set of loads
__device__ __forceinline__ void lag_sums(const double* v, int N, int m, double& prefactor, float& dev_sq) {
    double pf = 0.;
    float term = 0.f;
    // k + 2m inside the filter: k from max(0, -2m) to min(N, N - 2m) - 1
    const int k0 = m < 0 ? -2 * m : 0, k1 = m > 0 ? N - 2 * m : N;
#pragma unroll 4
    for (int k = k0; k < k1; k++) {
        const double x = v[k] * v[k + 2 * m];
        pf += x;
        term = (float)((double)term + x);
    }
    pf -= (m == 0 ? 1. : 0);
    prefactor = pf * 2.;
    const float dev = term - (m == 0 ? 1.f : 0.f);
    dev_sq = dev * dev;
}

__device__ __forceinline__ double sum_taps_lds(const double* v, int N) {
    double s = 0;
#pragma unroll 4
    for (int i = 0; i < N; i++) s += v[i];
    return s;
}

// regd += sum over the lags l of pref[l] * lag_inner(a, N, t, l - N/2), in the order of the lags; a is read
// through a zero margin (apad[i] = a[i - MAXTAPS] inside the filter, 0 outside), which is what skipping the
// taps outside the filter amounts to: 0 + x and x + 0 are x
__device__ __forceinline__ double add_lag_terms(double regd, const double* pref, const double* apad, int N, int t) {
#pragma unroll 4
    for (int l = 0; l <= N; l++) {
        const int m = l - N / 2;
        double inner = 0;
        inner += apad[MAXTAPS + t + 2 * m];
        inner += apad[MAXTAPS + t - 2 * m];
        regd += pref[l] * inner;
    }
    return regd;
}

// Everything the update reads from the filter state, fetched in one round trip (and, by the callers that have
// something else to wait for first, before that).
struct UpdateInputs {
    int N, was_bad, steps, dpad;
    double a_mine, p_mine, a_margin;
    __device__ __forceinline__ UpdateInputs(const FilterDev* f, int t) {
        N = f->N;
        was_bad = f->nonfinite;
        steps = f->steps;
        dpad = f->d;
        a_mine = (t < N) ? f->a_true[t] : 0.0;
        p_mine = (t < N) ? f->momentum[t] : 0.0;
        a_margin = (t >= MAXTAPS && t < MAXTAPS + N) ? f->a_true[t - MAXTAPS] : 0.0;
    }
};

// Wavenet::flushBatchQueue_ / update_ / setFilter (src/Wavenet.cxx:717-765, 8-36) from the batch's sums.
// Called by every thread of the workgroup (uniformly); threads 0..63 do the work: lane l evaluates lag
// m = l - N/2 of the regularisation sums, lane i adds them up for tap i in the reference's order.
// A batch whose sums, new filter or cost are not finite is not applied (the reference's train() fails
// before anything is queued, src/Wavenet.cxx:234-241), and neither is any batch after it: the flag
// stays up until the host takes it down.
__device__ inline void update_body(FilterDev* f, const UpdateInputs& in, const double* sums, const UpdateParams& P, UpdateShared& sh, int t) {
    const int N = in.N, was_bad = in.was_bad, steps = in.steps, dpad = in.dpad;
    const double a_mine = in.a_mine, p_mine = in.p_mine, a_margin = in.a_margin;
    if (t == 0) sh.bad = was_bad;
    __syncthreads();
    if (t < 64) {
        const bool bad_here = (t <= N && !isfinite(sums[t])) || (t == N + 1 && sums[t] != 0.0);
        if (bad_here) sh.bad = 1;
        if (t < N) sh.a[t] = a_mine;
        sh.apad[t] = a_margin;
        if (t < MAXTAPS) sh.apad[64 + t] = 0.0;
    }
    __syncthreads();
    if (t < N) sh.b[t] = parity_sign(t) * sh.a[N - t - 1];
    __syncthreads();
    if (t <= N && !sh.bad) {
        const int m = t - N / 2;
        lag_sums(sh.a, N, m, sh.prefA[t], sh.devA[t]);
        lag_sums(sh.b, N, m, sh.prefB[t], sh.devB[t]);
    }
    __syncthreads();
    if (t < 64 && !sh.bad) {
        const double suma = sum_taps_lds(sh.a, N), sumb = sum_taps_lds(sh.b, N);
        if (t < N) {
            double regd = add_lag_terms(0., sh.prefA, sh.apad, N, t);                    // RegTermDeriv, tap t
            if (P.wavelet) {
                regd += 2 * (suma - 1.4142135623730951) * 1.0;
                regd = add_lag_terms(regd, sh.prefB, sh.apad, N, t);
                regd += 2 * (sumb - 0) * parity_sign(N - t - 1);
            }
            double gbar = sums[t] + P.lambda_sum * regd;
            gbar /= P.Bf;                                 // gradient /= float(size)          :727
            double p = p_mine * P.inertia_eff;            // scaleMomentum_                    :760
            p += -P.alpha * gbar;                         // addMomentum_                      :761
            sh.pnew[t] = p;
            sh.anew[t] = sh.a[t] + p;                     // setFilter(m_filter + m_momentum)  :762
        }
        if (t == 0) {
            double reg = 0.;                                                            // RegTerm
            float R2 = 0.f, R3 = 0.f;
#pragma unroll 4
            for (int l = 0; l <= N; l++) {
                R2 = R2 + sh.devA[l];
                R3 = R3 + sh.devB[l];
            }
            reg += R2;
            if (P.wavelet) {
                const double c1 = suma - 1.4142135623730951;
                reg += c1 * c1;
                reg += R3;
                reg += sumb * sumb;
            }
            reg = reg / 2.;
            sh.cost = (sums[N] + P.lambda_sum * reg) / P.Bf;                            // :227, :733
        }
    }
    __syncthreads();
    // the same for a filter or a cost that overflowed on the way
    if (!sh.bad && ((t < N && !isfinite(sh.anew[t])) || (t == 0 && !isfinite(sh.cost)))) sh.bad = 1;
    __syncthreads();
    if (sh.bad) {
        if (t == 0) f->nonfinite = 1;
        return;
    }
    // the new state: lane j writes entry j of the padded taps, of the plain filter, of the momentum and of the log
    const int idx = steps % P.cap;
    if (t < MAXTAPS) {
        const int j = t - dpad;
        const bool real = (j >= 0) & (j < N);
        f->a[t] = real ? sh.anew[real ? j : 0] : 0.0;
        f->b[t] = real ? parity_sign(j) * sh.anew[real ? N - 1 - j : 0] : 0.0;
        if (t < N) {
            f->a_true[t] = sh.anew[t];
            f->momentum[t] = sh.pnew[t];
        }
        P.logF[(size_t)idx * MAXTAPS + t] = (t < N) ? sh.anew[t] : 0.0;
    }
    if (t == 0) {
        P.logC[idx] = sh.cost;
        f->steps = steps + 1;
    }
}

// A partial sum leaves the workgroup write-through (an agent-scope store), so that the workgroup that ends the
// step finds it in memory without anyone having to write a whole L2 back.
__device__ __forceinline__ void store_partial(double* p, double v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Sum of quantity q over the partial-sum slots: lane l adds slots l, l + 64, ... in order, the 64 lane
// sums are combined by a shuffle tree -- a fixed order, so the result does not depend on timing.  Lane 0
// holds the result.
__device__ __forceinline__ double reduce_slots(const double* partial, int nslots, int stride, int q, int lane) {
    double s = 0;
    // sixteen loads in flight at a time, added in slot order
    for (int base = 0; base < nslots; base += 64 * 16) {
        double v[16];
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const int slot = base + 64 * i + lane;
            v[i] = (slot < nslots) ? partial[(size_t)slot * stride + q] : 0.0;
        }
#pragma unroll
        for (int i = 0; i < 16; i++) {
            if (base + 64 * i + lane < nslots) s += v[i];
        }
    }
    return wave_sum(s);
}

// Called by every thread of every workgroup of a training kernel after its partial sums have been stored
// (store_partial).  T threads per workgroup; nslots slots of `stride` doubles in all; lds: the start of
// the kernel's dynamic shared memory, at least TAIL_LDS_BYTES, 16-byte aligned, no longer in use.
template <int T>
__device__ __forceinline__ void step_tail(const StepTail& S, const double* partial, int nslots, int stride, void* lds, int tid) {
    if (!S.ticket) return;   // the same in every thread of the launch
    constexpr int NW = T / 64;
    int* lastp = reinterpret_cast<int*>(lds);
    double* ssum = reinterpret_cast<double*>(reinterpret_cast<char*>(lds) + TAIL_OFF_SUMS);
    UpdateShared& sh = *reinterpret_cast<UpdateShared*>(reinterpret_cast<char*>(lds) + TAIL_OFF_SHARED);
    // Hand-over of the slots to the workgroup that takes the last ticket.  In the HIP memory model this is a release on
    // the ticket by every workgroup and an acquire by every thread that reads the slots; build with -DWN_TAIL_RELEASE
    // for exactly that (measured: 4 us more per launch of 4096 images, +1.6 %, because an agent-scope release also
    // writes the XCD's whole L2 back, register spills and all).  The default form reaches the same ordering on gfx950
    // with less, and depends on these properties of that target, which is why other targets are refused below:
    //   1. store_partial's agent-scope atomic stores are write-through (sc1): the data is at the memory side, past
    //      the non-coherent per-XCD L2, when the store is acknowledged;
    //   2. vmcnt counts those stores, so "s_waitcnt vmcnt(0)" returns only after the acknowledgement -- before the
    //      workgroup barrier, hence before thread 0 takes its ticket (program order within the wave, the barrier for
    //      the other waves), and the ticket itself is an agent-scope atomic executed at the memory side;
    //   3. the acquire fence of ONE wave (buffer_inv sc1) drops the stale lines of its CU's vector cache and of its
    //      XCD's L2 for every wave of the workgroup, so the plain loads of the other 255 threads behind the second
    //      barrier miss and fetch the slots from memory.
    // tests/test_gpu_step.py runs hundreds of consecutive one-launch steps over 1024 and more slots and requires the
    // sums to be bit-identical to those of separate launches: a stale slot would show there.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(WN_TAIL_RELEASE)
#error "step_tail's default hand-over relies on gfx950 cache behaviour: build other targets with -DWN_TAIL_RELEASE"
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's slot has left the wave
    __syncthreads();                                    // ... and so has every other wave's of the workgroup
    if (tid == 0) {
#ifdef WN_TAIL_RELEASE
        const unsigned mine = __hip_atomic_fetch_add(S.ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
#else
        const unsigned mine = __hip_atomic_fetch_add(S.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
        const int is_last = (mine == gridDim.x - 1u) ? 1 : 0;
        if (is_last) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        *lastp = is_last;
    }
    __syncthreads();
    if (!*lastp) return;     // the same in every thread of the workgroup
#ifdef WN_TAIL_RELEASE
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
    const UpdateInputs state(S.filt, tid);     // on its way while the slots are added up
    const int N = state.N, nq = N + 2;
    const int lane = tid & 63, wave = tid >> 6;
    // The order of reduce_slots (slots l, l + 64, ... in order, then the 64 partial results by a shuffle tree), computed
    // from whole rows as k_finish does: a thread owns (l, quantity) pairs, consecutive threads consecutive quantities
    // of a slot, so the slots are read as rows and not as one cache line per lane and quantity.
    double (*lsum)[64] = reinterpret_cast<double (*)[64]>(reinterpret_cast<char*>(lds) + TAIL_OFF_LSUM);
    for (int pair = tid; pair < 64 * nq; pair += T) {
        const int l = pair / nq, q = pair - l * nq;
        double s = 0;
        for (int base = l; base < nslots; base += 64 * 16) {
            double v[16];
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const int slot = base + 64 * i;
                v[i] = (slot < nslots) ? partial[(size_t)slot * stride + q] : 0.0;
            }
#pragma unroll
            for (int i = 0; i < 16; i++) {
                if (base + 64 * i < nslots) s += v[i];
            }
        }
        lsum[q][l] = s;
    }
    if (tid == 0) __hip_atomic_store(S.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // for the next launch
    __syncthreads();
    for (int q = wave; q < nq; q += NW) {
        const double s = wave_sum(lsum[q][lane]);
        if (lane == 0) { ssum[q] = s; S.sums[q] = s; }
    }
    __syncthreads();
    if (S.mode == 2) update_body(S.filt, state, ssum, S.P, sh, tid);
}

}  // namespace wn
