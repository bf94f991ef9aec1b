// Regularisation term of the wavelet conditions and its gradient
// (reference src/CostFunctions.cxx:119-257 and :259-415), shared by the host entry points
// and the on-device filter update.  The reference accumulates the C2 and C3 sums of RegTerm
// in `float` (CostFunctions.cxx:140-160, 188-208); the float/double roundings are reproduced
// here, so this file must be compiled with -ffp-contract=off.
//
// The building blocks are per lag m so that the device update can spread the lags over lanes
// while adding them up in the reference's order.
#pragma once

#include <math.h>

#ifdef __HIPCC__
#define WN_HD __host__ __device__
#else
#define WN_HD
#endif

namespace wn {

WN_HD inline bool in_range(int idx, int N) { return idx >= 0 && idx < N; }
WN_HD inline double parity_sign(int i) { return (i & 1) ? -1.0 : 1.0; }  // pow(-1, i)

// sum_k v_k v_{k+2m} accumulated in float, then (.. - delta_{m0})^2 in float   (:146-159)
WN_HD inline float lag_deviation_sq_float(const double* v, int N, int m) {
    const float delta = (m == 0 ? 1.f : 0.f);
    float term = 0.f;
    for (int k = 0; k < N; k++) {
        if (in_range(k + 2 * m, N)) term = (float)((double)term + v[k] * v[k + 2 * m]);
    }
    const float dev = term - delta;
    return dev * dev;
}

// 2 (sum_k v_k v_{k+2m} - delta_{m0}) in double                                 (:290-303)
WN_HD inline double lag_prefactor(const double* v, int N, int m) {
    double prefactor = 0.;
    for (int k = 0; k < N; k++) {
        if (in_range(k + 2 * m, N)) prefactor += v[k] * v[k + 2 * m];
    }
    prefactor -= (m == 0 ? 1. : 0);
    prefactor *= 2.;
    return prefactor;
}

// a_{i+2m} + a_{i-2m}                                                            (:309-316)
WN_HD inline double lag_inner(const double* a, int N, int i, int m) {
    double inner = 0;
    if (in_range(i + 2 * m, N)) inner += a[i + 2 * m];
    if (in_range(i - 2 * m, N)) inner += a[i - 2 * m];
    return inner;
}

WN_HD inline void highpass_taps(const double* a, int N, double* b) {
    for (int i = 0; i < N; i++) b[i] = parity_sign(i) * a[N - i - 1];
}

WN_HD inline double sum_taps(const double* v, int N) {
    double s = 0;
    for (int i = 0; i < N; i++) s += v[i];
    return s;
}

// RegTerm(a, doWavelet).  N <= 64.
WN_HD inline double reg_term(const double* a, int N, bool doWavelet) {
    double R = 0.;
    float R2 = 0.f;
    for (int m = -N / 2; m <= N / 2; m++) R2 = R2 + lag_deviation_sq_float(a, N, m);
    R += R2;  // (C2)
    if (doWavelet) {
        double b[64];
        highpass_taps(a, N, b);
        const double c1 = sum_taps(a, N) - 1.4142135623730951;  // sqrt(2)
        R += c1 * c1;                                             // (C1)
        float R3 = 0.f;
        for (int m = -N / 2; m <= N / 2; m++) R3 = R3 + lag_deviation_sq_float(b, N, m);
        R += R3;                                                  // (C3)
        const double sumb = sum_taps(b, N);
        R += sumb * sumb;                                         // (C4)
    }
    return R / 2.;
}

// RegTermDeriv(a, doWavelet).  N <= 64.
WN_HD inline void reg_term_deriv(const double* a, int N, bool doWavelet, double* gradient) {
    for (int i = 0; i < N; i++) gradient[i] = 0;
    for (int m = -N / 2; m <= N / 2; m++) {  // (C2) :287-320
        const double prefactor = lag_prefactor(a, N, m);
        for (int i = 0; i < N; i++) gradient[i] += prefactor * lag_inner(a, N, i, m);
    }
    if (doWavelet) {
        double b[64];
        highpass_taps(a, N, b);
        const double c1 = 2 * (sum_taps(a, N) - 1.4142135623730951);  // (C1) :342
        for (int i = 0; i < N; i++) gradient[i] += c1 * 1.0;
        for (int m = -N / 2; m <= N / 2; m++) {                        // (C3) :356-389
            const double prefactor = lag_prefactor(b, N, m);
            for (int i = 0; i < N; i++) gradient[i] += prefactor * lag_inner(a, N, i, m);
        }
        const double c4 = 2 * (sum_taps(b, N) - 0);                    // (C4) :399
        for (int i = 0; i < N; i++) gradient[i] += c4 * parity_sign(N - i - 1);
    }
}

}  // namespace wn
