// Regularisation term of the wavelet conditions and its gradient
// (reference src/CostFunctions.cxx:119-257 and :259-415), shared by the host entry points
// and the on-device filter update.  The reference accumulates the C2 and C3 sums of RegTerm
// in `float` (CostFunctions.cxx:140-160, 188-208); the float/double roundings are reproduced
// here, so this file must be compiled with -ffp-contract=off.
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

// sum_m (sum_k v_k v_{k+2m} - delta_{m0})^2 with the reference's float accumulators.
WN_HD inline float orthonormality_float(const double* v, int N) {
    float R = 0.f;
    for (int m = -N / 2; m <= N / 2; m++) {
        const float delta = (m == 0 ? 1.f : 0.f);
        float term = 0.f;
        for (int k = 0; k < N; k++) {
            if (in_range(k + 2 * m, N)) term = (float)((double)term + v[k] * v[k + 2 * m]);
        }
        const float dev = term - delta;
        R = R + dev * dev;
    }
    return R;
}

// RegTerm(a, doWavelet).  N <= 64.
WN_HD inline double reg_term(const double* a, int N, bool doWavelet) {
    double R = 0.;
    R += orthonormality_float(a, N);  // (C2)
    if (doWavelet) {
        double b[64];
        for (int i = 0; i < N; i++) b[i] = parity_sign(i) * a[N - i - 1];
        double suma = 0;
        for (int i = 0; i < N; i++) suma += a[i];
        const double c1 = suma - 1.4142135623730951;  // sqrt(2)
        R += c1 * c1;                                  // (C1)
        R += orthonormality_float(b, N);               // (C3)
        double sumb = 0;
        for (int i = 0; i < N; i++) sumb += b[i];
        R += sumb * sumb;                              // (C4)
    }
    return R / 2.;
}

// gradient += sum_m 2 (sum_k v_k v_{k+2m} - delta_{m0}) (a_{i+2m} + a_{i-2m})
WN_HD inline void orthonormality_deriv(const double* v, const double* a, int N, double* gradient) {
    for (int m = -N / 2; m <= N / 2; m++) {
        double prefactor = 0.;
        for (int k = 0; k < N; k++) {
            if (in_range(k + 2 * m, N)) prefactor += v[k] * v[k + 2 * m];
        }
        prefactor -= (m == 0 ? 1. : 0);
        prefactor *= 2.;
        for (int i = 0; i < N; i++) {
            double inner = 0;
            if (in_range(i + 2 * m, N)) inner += a[i + 2 * m];
            if (in_range(i - 2 * m, N)) inner += a[i - 2 * m];
            gradient[i] += prefactor * inner;
        }
    }
}

// RegTermDeriv(a, doWavelet).  N <= 64.
WN_HD inline void reg_term_deriv(const double* a, int N, bool doWavelet, double* gradient) {
    for (int i = 0; i < N; i++) gradient[i] = 0;
    orthonormality_deriv(a, a, N, gradient);  // (C2) :287-320
    if (doWavelet) {
        double b[64];
        for (int i = 0; i < N; i++) b[i] = parity_sign(i) * a[N - i - 1];
        double suma = 0;
        for (int i = 0; i < N; i++) suma += a[i];
        const double c1 = 2 * (suma - 1.4142135623730951);  // (C1) :342
        for (int i = 0; i < N; i++) gradient[i] += c1 * 1.0;
        orthonormality_deriv(b, a, N, gradient);             // (C3) :356-389
        double sumb = 0;
        for (int i = 0; i < N; i++) sumb += b[i];
        const double c4 = 2 * (sumb - 0);                    // (C4) :399
        for (int i = 0; i < N; i++) gradient[i] += c4 * parity_sign(N - i - 1);
    }
}

}  // namespace wn
