#include "Wavenet/CostFunctions.h"

#include <cmath>
#include <limits>

#include "Wavenet/Logger.h"
#include "Wavenet/Utilities.h"
#include "wavenet_hip.h"

namespace wavenet {

namespace {

// One context shared by the free functions (the reference's are stateless); device 0.
wn_context* sharedContext() {
    static wn_context* ctx = nullptr;
    if (!ctx && wn_create(0, &ctx) != WN_OK) {
        ERROR("No device context: %s", wn_last_error(nullptr));
        ctx = nullptr;
    }
    return ctx;
}

// Runs SparseTerm / SparseTermDeriv on n values; delta may be null.  NaN on failure.
double sparseOnDevice(const double* c, std::size_t n, double* delta) {
    const double nan = std::numeric_limits<double>::quiet_NaN();
    wn_context* ctx = sharedContext();
    if (!ctx) return nan;
    if (!isRadix2((unsigned)n) || n > WN_MAX_FUSED) {
        ERROR("SparseTerm on %zu coefficients is not supported (radix 2, at most %d).", n, WN_MAX_FUSED);
        return nan;
    }
    double *dY = nullptr, *dD = nullptr, *dS = nullptr;
    double S = nan;
    bool ok = wn_malloc(ctx, (void**)&dY, n * sizeof(double)) == WN_OK &&
              wn_malloc(ctx, (void**)&dD, n * sizeof(double)) == WN_OK &&
              wn_malloc(ctx, (void**)&dS, sizeof(double)) == WN_OK;
    ok = ok && wn_memcpy_h2d(ctx, dY, c, n * sizeof(double)) == WN_OK;
    ok = ok && wn_sparse_term_deriv(ctx, dY, 1, (int)n, delta ? dD : nullptr, dS) == WN_OK;
    ok = ok && wn_memcpy_d2h(ctx, &S, dS, sizeof(double)) == WN_OK;
    if (ok && delta) ok = wn_memcpy_d2h(ctx, delta, dD, n * sizeof(double)) == WN_OK;
    if (!ok) {
        ERROR("%s", wn_last_error(ctx));
        S = nan;
    }
    wn_free(ctx, dY);
    wn_free(ctx, dD);
    wn_free(ctx, dS);
    return S;
}

}  // namespace

double SparseTerm(const Col& c) { return sparseOnDevice(c.memptr(), c.n_elem, nullptr); }

double SparseTerm(const Mat& C) { return sparseOnDevice(C.memptr(), C.n_elem, nullptr); }  // vectorise = memory order

Col SparseTermDeriv(const Col& c) {
    Col g(c.n_elem);
    sparseOnDevice(c.memptr(), c.n_elem, g.memptr());
    return g;
}

Mat SparseTermDeriv(const Mat& C) {
    Mat G(C.n_rows, C.n_cols);
    sparseOnDevice(C.memptr(), C.n_elem, G.memptr());
    return G;
}

double RegTerm(const Col& a, const bool& doWavelet) {
    double R = std::numeric_limits<double>::quiet_NaN();
    const int status = wn_reg_term(a.memptr(), (int)a.n_elem, doWavelet, &R);
    if (status != WN_OK) WARNING("RegTerm: %s", wn_status_string(status));
    return R;
}

Col RegTermDeriv(const Col& a, const bool& doWavelet) {
    Col g(a.n_elem);
    const int status = wn_reg_term_deriv(a.memptr(), (int)a.n_elem, doWavelet, g.memptr());
    if (status != WN_OK) WARNING("RegTermDeriv: %s", wn_status_string(status));
    return g;
}

}  // namespace wavenet
