#include "Wavenet/CostFunctions.h"

#include <algorithm>
#include <cmath>
#include <limits>
#include <vector>

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

// Device buffers of the free functions, kept from call to call and grown when a longer vector comes.
struct SharedBuffers {
    double *Y = nullptr, *D = nullptr, *S = nullptr;
    std::size_t capacity = 0;
};

double sparseOnDevice(const double* c, std::size_t n, double* delta);

// A length that is not a power of two (the reference takes any, src/CostFunctions.cxx:6-46): the vector goes to the
// device behind p zeros that fill it up to n' = 2^k.  Zeros rank first and carry no weight, so the padded sums are
// f' = f + p sum|c| and g' = n' sum|c|, from which the true values follow without sum|c|:
//   S = 1 - ((1 - S') n' - p) / n,     delta_j = delta'_j n' / n.
double sparsePadded(const double* c, std::size_t n, double* delta) {
    const double nan = std::numeric_limits<double>::quiet_NaN();
    if (n == 0) return nan;
    std::size_t np2 = 1;
    while (np2 < n) np2 <<= 1;
    static std::vector<double> in, out;
    in.assign(np2, 0.0);
    std::copy(c, c + n, in.begin());
    if (delta) out.resize(np2);
    const double Sp = sparseOnDevice(in.data(), np2, delta ? out.data() : nullptr);
    if (!(Sp == Sp)) return nan;
    const double scale = (double)np2 / (double)n;
    if (delta) {
        for (std::size_t j = 0; j < n; j++) delta[j] = out[j] * scale;
    }
    return 1.0 - ((1.0 - Sp) * (double)np2 - (double)(np2 - n)) / (double)n;
}

// Runs SparseTerm / SparseTermDeriv on n values (any radix-2 length the library takes: the streamed ranking beyond
// WN_MAX_FUSED coefficients; other lengths zero-padded, see above); delta may be null.  NaN on failure.
double sparseOnDevice(const double* c, std::size_t n, double* delta) {
    const double nan = std::numeric_limits<double>::quiet_NaN();
    static SharedBuffers buf;
    wn_context* ctx = sharedContext();
    if (!ctx) return nan;
    if (n > ((std::size_t)1 << 30)) {
        ERROR("SparseTerm on %zu coefficients is not supported (at most 2^30).", n);
        return nan;
    }
    if (!isRadix2((unsigned)n)) return sparsePadded(c, n, delta);
    if (buf.capacity < n) {
        if (buf.Y) wn_free(ctx, buf.Y);
        if (buf.D) wn_free(ctx, buf.D);
        buf.Y = buf.D = nullptr;
        buf.capacity = 0;
        if (wn_malloc(ctx, (void**)&buf.Y, n * sizeof(double)) != WN_OK || wn_malloc(ctx, (void**)&buf.D, n * sizeof(double)) != WN_OK) {
            ERROR("%s", wn_last_error(ctx));
            if (buf.Y) wn_free(ctx, buf.Y);
            buf.Y = buf.D = nullptr;
            return nan;
        }
        buf.capacity = n;
    }
    if (!buf.S && wn_malloc(ctx, (void**)&buf.S, sizeof(double)) != WN_OK) { buf.S = nullptr; return nan; }
    double S = nan;
    bool ok = wn_memcpy_h2d(ctx, buf.Y, c, n * sizeof(double)) == WN_OK;
    ok = ok && wn_sparse_term_deriv(ctx, buf.Y, 1, (int)n, delta ? buf.D : nullptr, buf.S) == WN_OK;
    ok = ok && wn_memcpy_d2h(ctx, &S, buf.S, sizeof(double)) == WN_OK;
    if (ok && delta) ok = wn_memcpy_d2h(ctx, delta, buf.D, n * sizeof(double)) == WN_OK;
    if (!ok) {
        ERROR("%s", wn_last_error(ctx));
        S = nan;
    }
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
