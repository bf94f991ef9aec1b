/*
 * wavenet_oracle.c -- CPU restatement of the Sardingfish/Wavenet training path.
 *
 * TEST INFRASTRUCTURE ONLY (see wavenet_oracle.h).  PARITY UNPINNED: no
 * reference build and no reference fixtures exist to check this file against;
 * it follows the reference sources line by line and is pinned by closed-form
 * answers, finite differences and the dense/banded identity in tests/.
 *
 * Citations are file:line in the reference tree.
 *
 * Known places where the reference's result is not fixed by its own source:
 *   - summation order inside Armadillo's accu/dot/sum and BLAS dgemv
 *     (sequential order is used here; deviation <= n * eps);
 *   - arma::sort_index is an unstable std::sort, so the order of tied |c| is
 *     unspecified; ties are broken by ascending index here.
 */
#include "wavenet_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------ */
/* small helpers                                                             */
/* ------------------------------------------------------------------------ */

static unsigned ilog2u(unsigned n) { /* log2 of a power of two */
    unsigned m = 0;
    while ((1u << m) < n) m++;
    return m;
}

static void *xmalloc(size_t bytes) {
    void *p = malloc(bytes ? bytes : 1);
    if (!p) abort();
    return p;
}

static void *xcalloc(size_t n, size_t sz) {
    void *p = calloc(n ? n : 1, sz ? sz : 1);
    if (!p) abort();
    return p;
}

typedef struct { double *d; size_t n, cap; } dvec;
static void dvec_push(dvec *v, double x) {
    if (v->n == v->cap) {
        v->cap = v->cap ? 2 * v->cap : 16;
        v->d = (double *)realloc(v->d, v->cap * sizeof(double));
        if (!v->d) abort();
    }
    v->d[v->n++] = x;
}
static void dvec_push_n(dvec *v, const double *x, int n) {
    for (int i = 0; i < n; i++) dvec_push(v, x[i]);
}

/* ------------------------------------------------------------------------ */
/* MatrixOperator                                                            */
/* ------------------------------------------------------------------------ */

void orc_highpass_filter(const double *a, int N, double *b) {
    /* HighpassOperator.cxx:16-18 */
    for (int i = 0; i < N; i++) b[i] = pow(-1, i) * a[N - i - 1];
}

/* MatrixOperator.cxx:115-143.  Rotates LEFT by `shift` (std::rotate). */
static void rowshift(double *row, unsigned length, int shift) {
    if (shift > 0 && (unsigned)shift < length) {
        double *tmp = (double *)xmalloc(length * sizeof(double));
        for (unsigned i = 0; i < length; i++) tmp[i] = row[(i + (unsigned)shift) % length];
        memcpy(row, tmp, length * sizeof(double));
        free(tmp);
    } else {
        if (length < (unsigned)abs(shift)) {
            int sgn = shift == 0 ? 0 : (shift > 0 ? 1 : -1);
            rowshift(row, length, sgn * (abs(shift) % (int)length));
        }
        if (shift < 0) rowshift(row, length, (int)length + shift);
    }
}

void orc_operator_by_rows(const double *f, int Ni, unsigned size, double *M) {
    /* MatrixOperator.cxx:31-61 */
    const unsigned nRows = 1u << size;
    const unsigned nCols = 1u << (size + 1);
    const unsigned N = (unsigned)Ni;
    memset(M, 0, (size_t)nRows * nCols * sizeof(double));

    double *v = (double *)xcalloc(nCols, sizeof(double));
    for (unsigned i = 0; i < N; i++) {
        v[(i - N / 2 + 1) % nCols] += f[N - i - 1]; /* unsigned wrap-around, :45 */
    }
    for (unsigned c = 0; c < nCols; c++) M[0 + (size_t)c * nRows] = v[c];
    for (unsigned i = nRows; i-- > 1;) {
        rowshift(v, nCols, 2);
        for (unsigned c = 0; c < nCols; c++) M[i + (size_t)c * nRows] = v[c];
    }
    free(v);
}

void orc_operator_by_indices(const double *f, int Ni, unsigned size, double *M) {
    /* MatrixOperator.cxx:63-113 */
    const unsigned nRows = 1u << size;
    const unsigned nCols = 1u << (size + 1);
    const unsigned N = (unsigned)Ni;
    memset(M, 0, (size_t)nRows * nCols * sizeof(double));

    unsigned long long *rows = (unsigned long long *)xmalloc(nRows * sizeof(*rows));
    unsigned long long *cols = (unsigned long long *)xmalloc(nRows * sizeof(*cols));
    for (unsigned irow = 0; irow < nRows; irow++) rows[irow] = irow;
    for (unsigned icol = 0; icol < nRows; icol++) cols[icol] = (N / 2 + 2 * icol) % nCols;

    for (unsigned i = 0; i < N; i++) {
        for (unsigned r = 0; r < nRows; r++) M[cols[r] * nRows + rows[r]] += f[i];
        for (unsigned r = 0; r < nRows; r++) {
            if (cols[r] == 0) cols[r] += nCols;
        }
        for (unsigned r = 0; r < nRows; r++) cols[r] -= 1;
    }
    free(rows);
    free(cols);
}

int orc_operator_construct(const double *f, int Ni, unsigned size, double *M) {
    /* MatrixOperator.cxx:18-26 */
    const unsigned N = (unsigned)Ni;
    if ((N == 2 && size >= 2) || (N <= 6 && size >= 3) || (N <= 20 && size >= 4) || (size >= 5)) {
        orc_operator_by_indices(f, Ni, size, M);
        return 1;
    }
    orc_operator_by_rows(f, Ni, size, M);
    return 0;
}

/* LowpassOperator(filter,size) / HighpassOperator(filter,size)
 * (LowpassOperator.h:44-45, HighpassOperator.h:48-49). */
static double *new_lowpass_operator(const double *a, int N, unsigned size) {
    double *M = (double *)xmalloc(((size_t)1 << size) * ((size_t)2 << size) * sizeof(double));
    orc_operator_construct(a, N, size, M);
    return M;
}
static double *new_highpass_operator(const double *a, int N, unsigned size) {
    double *b = (double *)xmalloc((size_t)N * sizeof(double));
    orc_highpass_filter(a, N, b);
    double *M = (double *)xmalloc(((size_t)1 << size) * ((size_t)2 << size) * sizeof(double));
    orc_operator_construct(b, N, size, M);
    free(b);
    return M;
}

/* ------------------------------------------------------------------------ */
/* CostFunctions                                                             */
/* ------------------------------------------------------------------------ */

static int cmp_double(const void *pa, const void *pb) {
    const double a = *(const double *)pa, b = *(const double *)pb;
    return (a > b) - (a < b);
}

typedef struct { double v; int i; } keyidx;
static int cmp_keyidx(const void *pa, const void *pb) {
    const keyidx *a = (const keyidx *)pa, *b = (const keyidx *)pb;
    if (a->v < b->v) return -1;
    if (a->v > b->v) return 1;
    return (a->i > b->i) - (a->i < b->i); /* ties: ascending index */
}

/* Utilities.h:57-70: sq, norm, sign for doubles. */
static double sq_d(double x) { return x * x; }
static double sign_d(double x) { return x == 0 ? 0 : x / sqrt(sq_d(x)); }

double orc_sparse_term(const double *c, int N) {
    /* CostFunctions.cxx:6-37 */
    double *cSortAbs = (double *)xmalloc((size_t)N * sizeof(double));
    for (int i = 0; i < N; i++) cSortAbs[i] = fabs(c[i]);
    qsort(cSortAbs, (size_t)N, sizeof(double), cmp_double);
    double f = 0, s = 0;
    for (int i = 0; i < N; i++) {
        const double index = (double)(i + 1);
        f += (2 * index - N - 1) * cSortAbs[i];
        s += cSortAbs[i];
    }
    const double g = N * s;
    free(cSortAbs);
    return 1. - f / g;
}

void orc_sparse_term_deriv(const double *c, int N, double *gradient) {
    /* CostFunctions.cxx:49-105 */
    keyidx *ks = (keyidx *)xmalloc((size_t)N * sizeof(keyidx));
    double *cSign = (double *)xmalloc((size_t)N * sizeof(double));
    double *df = (double *)xcalloc((size_t)N, sizeof(double));
    for (int i = 0; i < N; i++) {
        ks[i].v = fabs(c[i]);
        ks[i].i = i;
        cSign[i] = sign_d(c[i]);
    }
    qsort(ks, (size_t)N, sizeof(keyidx), cmp_keyidx); /* sort_index(cAbs,"ascend") :74 */

    double f = 0, s = 0;
    for (int i = 0; i < N; i++) {
        const double index = (double)(i + 1);
        f += (2 * index - N - 1) * ks[i].v; /* :85 */
        s += ks[i].v;
    }
    const double g = N * s; /* :86 */
    for (int i = 0; i < N; i++) {
        const double index = (double)(i + 1);
        df[ks[i].i] = (2 * index - N - 1); /* :89-93 */
    }
    const double g2 = pow(g, 2.);
    for (int i = 0; i < N; i++) {
        const double dfi = df[i] * cSign[i]; /* :96 */
        const double dgi = N * cSign[i];     /* :99 */
        gradient[i] = -(dfi * g - f * dgi) / g2; /* :102 */
    }
    free(ks);
    free(cSign);
    free(df);
}

static int in_range(long long idx, int N) { return idx >= 0 && idx < N; }

double orc_reg_term(const double *a, int N, int doWavelet) {
    /* CostFunctions.cxx:119-257.  R2, R3, term and delta are `float` there. */
    double R = 0.;

    float R2 = 0.;
    for (int m = -N / 2; m <= N / 2; m++) {
        float delta = (m == 0 ? 1 : 0);
        float term = 0.;
        for (int k = 0; k < N; k++) {
            if (in_range((long long)k + 2 * m, N)) {
                term = (float)((double)term + a[k] * a[k + 2 * m]); /* :154 float += double */
            }
        }
        const float dev = term - delta;
        R2 = R2 + dev * dev; /* :159 sq<float> */
    }
    R += R2;

    if (doWavelet) {
        double *b = (double *)xmalloc((size_t)N * sizeof(double));
        for (int i = 0; i < N; i++) b[i] = pow(-1, i) * a[N - i - 1];

        double suma = 0;
        for (int i = 0; i < N; i++) suma += a[i];
        R += sq_d(suma - sqrt(2)); /* :179 */

        float R3 = 0.;
        for (int m = -N / 2; m <= N / 2; m++) {
            float delta = (m == 0 ? 1 : 0);
            float term = 0.;
            for (int k = 0; k < N; k++) {
                if (in_range((long long)k + 2 * m, N)) {
                    term = (float)((double)term + b[k] * b[k + 2 * m]); /* :202 */
                }
            }
            const float dev = term - delta;
            R3 = R3 + dev * dev; /* :207 */
        }
        R += R3;

        double sumb = 0;
        for (int i = 0; i < N; i++) sumb += b[i];
        R += sq_d(sumb - 0); /* :218 */
        free(b);
    }
    return R / 2.;
}

void orc_reg_term_deriv(const double *a, int N, int doWavelet, double *gradient) {
    /* CostFunctions.cxx:259-415 */
    double *inner = (double *)xmalloc((size_t)N * sizeof(double));
    for (int i = 0; i < N; i++) gradient[i] = 0;

    for (int m = -N / 2; m <= N / 2; m++) { /* (C2) :287-320 */
        double prefactor = 0.;
        for (int k = 0; k < N; k++) {
            if (in_range((long long)k + 2 * m, N)) prefactor += a[k] * a[k + 2 * m];
        }
        prefactor -= (m == 0 ? 1. : 0);
        prefactor *= 2.;
        for (int i = 0; i < N; i++) {
            inner[i] = 0;
            if (in_range((long long)i + 2 * m, N)) inner[i] += a[i + 2 * m];
            if (in_range((long long)i - 2 * m, N)) inner[i] += a[i - 2 * m];
        }
        for (int i = 0; i < N; i++) gradient[i] += prefactor * inner[i];
    }

    if (doWavelet) {
        double *b = (double *)xmalloc((size_t)N * sizeof(double));
        double *bsign = (double *)xmalloc((size_t)N * sizeof(double));
        for (int i = 0; i < N; i++) {
            b[i] = pow(-1, i) * a[N - i - 1];
            bsign[i] = pow(-1, N - i - 1);
        }

        double suma = 0;
        for (int i = 0; i < N; i++) suma += a[i];
        const double c1 = 2 * (suma - sqrt(2)); /* (C1) :342 */
        for (int i = 0; i < N; i++) gradient[i] += c1 * 1.0;

        for (int m = -N / 2; m <= N / 2; m++) { /* (C3) :356-389 */
            double prefactor = 0.;
            for (int k = 0; k < N; k++) {
                if (in_range((long long)k + 2 * m, N)) prefactor += b[k] * b[k + 2 * m];
            }
            prefactor -= (m == 0 ? 1. : 0);
            prefactor *= 2.;
            for (int i = 0; i < N; i++) {
                inner[i] = 0;
                if (in_range((long long)i + 2 * m, N)) inner[i] += a[i + 2 * m];
                if (in_range((long long)i - 2 * m, N)) inner[i] += a[i - 2 * m];
            }
            for (int i = 0; i < N; i++) gradient[i] += prefactor * inner[i];
        }

        double sumb = 0;
        for (int i = 0; i < N; i++) sumb += b[i];
        const double c4 = 2 * (sumb - 0); /* (C4) :399 */
        for (int i = 0; i < N; i++) gradient[i] += c4 * bsign[i];
        free(b);
        free(bsign);
    }
    free(inner);
}

/* ------------------------------------------------------------------------ */
/* Wavenet                                                                   */
/* ------------------------------------------------------------------------ */

#define ORC_MAX_LEVELS 31

struct orc_wavenet {
    double lambda, alpha, inertia, inertiaTimeScale; /* Wavenet.h:712-762 */
    int N;
    double *filter, *momentum;
    int momentum_n;

    int hasCachedOperators;
    unsigned nCachedOperators; /* size(m_cachedLowpassOperators, 0) */
    double *L[ORC_MAX_LEVELS + 1], *H[ORC_MAX_LEVELS + 1];

    int hasCachedWeights;
    unsigned nCachedWeights; /* size(m_cachedLowpassWeights, 0) */
    int weightsN;
    double **WL[ORC_MAX_LEVELS + 1], **WH[ORC_MAX_LEVELS + 1];

    unsigned batchSize;
    dvec batchQueue; /* entries of N doubles */
    dvec filterLog;  /* entries of N doubles */
    int filterLogEntries;
    dvec costLog;
    int wavelet;
};

static void clearCachedOperators_(orc_wavenet *w) {
    /* Wavenet.cxx:788-798 */
    for (unsigned i = 0; i <= ORC_MAX_LEVELS; i++) {
        free(w->L[i]); w->L[i] = NULL;
        free(w->H[i]); w->H[i] = NULL;
    }
    w->nCachedOperators = 0;
    w->hasCachedOperators = 0;
}

static void clearCachedWeights_(orc_wavenet *w) {
    /* Wavenet.cxx:841-851 */
    for (unsigned i = 0; i <= ORC_MAX_LEVELS; i++) {
        if (w->WL[i]) {
            for (int k = 0; k < w->weightsN; k++) { free(w->WL[i][k]); free(w->WH[i][k]); }
            free(w->WL[i]); free(w->WH[i]);
            w->WL[i] = NULL; w->WH[i] = NULL;
        }
    }
    w->nCachedWeights = 0;
    w->hasCachedWeights = 0;
}

static void cacheOperators_(orc_wavenet *w, unsigned m) {
    /* Wavenet.cxx:767-786 */
    clearCachedOperators_(w);
    for (unsigned i = 0; i <= m; i++) {
        w->L[i] = new_lowpass_operator(w->filter, w->N, i);
        w->H[i] = new_highpass_operator(w->filter, w->N, i);
    }
    w->nCachedOperators = m + 1;
    w->hasCachedOperators = 1;
}

static void cacheWeights_(orc_wavenet *w, unsigned m) {
    /* Wavenet.cxx:800-839 */
    clearCachedWeights_(w);
    const int N = w->N;
    w->weightsN = N;
    double *t = (double *)xmalloc((size_t)N * sizeof(double));
    for (unsigned i = 0; i <= m; i++) {
        w->WL[i] = (double **)xmalloc((size_t)N * sizeof(double *));
        w->WH[i] = (double **)xmalloc((size_t)N * sizeof(double *));
        for (int k = 0; k < N; k++) {
            for (int j = 0; j < N; j++) t[j] = 0;
            t[k] = 1;
            w->WH[i][k] = new_highpass_operator(t, N, i);
            w->WL[i][k] = new_lowpass_operator(t, N, i);
        }
    }
    free(t);
    w->nCachedWeights = m + 1;
    w->hasCachedWeights = 1;
}

orc_wavenet *orc_wavenet_new(void) {
    orc_wavenet *w = (orc_wavenet *)xcalloc(1, sizeof(orc_wavenet));
    w->lambda = 10.0;
    w->alpha = 0.001;
    w->inertia = 0.;
    w->inertiaTimeScale = 0.;
    w->batchSize = 1;
    w->wavelet = 1;
    dvec_push(&w->costLog, 0); /* m_costLog = {0}, Wavenet.h:847 */
    return w;
}

void orc_wavenet_free(orc_wavenet *w) {
    if (!w) return;
    clearCachedOperators_(w);
    clearCachedWeights_(w);
    free(w->filter); free(w->momentum);
    free(w->batchQueue.d); free(w->filterLog.d); free(w->costLog.d);
    free(w);
}

int orc_set_filter(orc_wavenet *w, const double *a, int N) {
    /* Wavenet.cxx:8-36 */
    if (N <= 0) return 0;
    if (N % 2) return 0;
    if (N != w->N) {
        /* The reference keeps stale weight matrices when the filter length
         * changes (they depend on N only); they are dropped here so that the
         * next backpropagation rebuilds them with the right count. */
        clearCachedWeights_(w);
        /* Logs hold entries of one length; a length change restarts them. */
        w->filterLog.n = 0; w->filterLogEntries = 0;
        w->batchQueue.n = 0;
        w->filter = (double *)realloc(w->filter, (size_t)N * sizeof(double));
    }
    w->N = N;
    memcpy(w->filter, a, (size_t)N * sizeof(double));
    dvec_push_n(&w->filterLog, a, N);
    w->filterLogEntries++;
    clearCachedOperators_(w);
    if (w->momentum_n != N) {
        w->momentum = (double *)realloc(w->momentum, (size_t)N * sizeof(double));
        for (int i = 0; i < N; i++) w->momentum[i] = 0;
        w->momentum_n = N;
    }
    return 1;
}

void orc_set_lambda(orc_wavenet *w, double v) { w->lambda = v; }
void orc_set_alpha(orc_wavenet *w, double v) { w->alpha = v; }
void orc_set_inertia(orc_wavenet *w, double v) { w->inertia = v; }
void orc_set_inertia_time_scale(orc_wavenet *w, double v) { w->inertiaTimeScale = v; }
void orc_set_batch_size(orc_wavenet *w, unsigned v) { w->batchSize = v; }
void orc_do_wavelet(orc_wavenet *w, int flag) { w->wavelet = flag; }

void orc_clear(orc_wavenet *w) {
    /* Wavenet.cxx:244-250 */
    for (int i = 0; i < w->momentum_n; i++) w->momentum[i] = w->momentum[i] * 0.;
    w->filterLog.n = 0; w->filterLogEntries = 0;
    w->costLog.n = 0; dvec_push(&w->costLog, 0); /* resize(1, 0) */
    clearCachedOperators_(w);
}

int orc_num_coeffs(const orc_wavenet *w) { return w->N; }
double orc_lambda(const orc_wavenet *w) { return w->lambda; }
double orc_alpha(const orc_wavenet *w) { return w->alpha; }
unsigned orc_batch_size(const orc_wavenet *w) { return w->batchSize; }
void orc_get_filter(const orc_wavenet *w, double *a) { memcpy(a, w->filter, (size_t)w->N * sizeof(double)); }
void orc_get_momentum(const orc_wavenet *w, double *p) { memcpy(p, w->momentum, (size_t)w->N * sizeof(double)); }
int orc_cost_log_size(const orc_wavenet *w) { return (int)w->costLog.n; }
void orc_get_cost_log(const orc_wavenet *w, double *out) { memcpy(out, w->costLog.d, w->costLog.n * sizeof(double)); }
void orc_cost_log_pop_back(orc_wavenet *w) { if (w->costLog.n) w->costLog.n--; }
int orc_filter_log_size(const orc_wavenet *w) { return w->filterLogEntries; }
void orc_get_filter_log(const orc_wavenet *w, double *out) { memcpy(out, w->filterLog.d, w->filterLog.n * sizeof(double)); }
int orc_batch_queue_size(const orc_wavenet *w) { return w->N ? (int)(w->batchQueue.n / (size_t)w->N) : 0; }

/* dense products ---------------------------------------------------------- */

/* y = M * x, M is nr x nc column-major */
static void matvec(const double *M, unsigned nr, unsigned nc, const double *x, double *y) {
    for (unsigned r = 0; r < nr; r++) y[r] = 0;
    for (unsigned c = 0; c < nc; c++) {
        const double xc = x[c];
        const double *col = M + (size_t)c * nr;
        for (unsigned r = 0; r < nr; r++) y[r] += col[r] * xc;
    }
}
/* y = M^T * x, M is nr x nc column-major, x has nr entries, y has nc */
static void matvec_t(const double *M, unsigned nr, unsigned nc, const double *x, double *y) {
    for (unsigned c = 0; c < nc; c++) {
        const double *col = M + (size_t)c * nr;
        double acc = 0;
        for (unsigned r = 0; r < nr; r++) acc += col[r] * x[r];
        y[c] = acc;
    }
}

/* Wavenet.cxx:853-899; `out` is caller-allocated. */
static void lowpassfilter_(orc_wavenet *w, const double *x, unsigned n, double *out) {
    const unsigned m = ilog2u(n);
    if (!w->hasCachedOperators || w->nCachedOperators < m) cacheOperators_(w, m);
    matvec(w->L[m - 1], n / 2, n, x, out);
}
static void highpassfilter_(orc_wavenet *w, const double *x, unsigned n, double *out) {
    const unsigned m = ilog2u(n);
    if (!w->hasCachedOperators || w->nCachedOperators < m) cacheOperators_(w, m);
    matvec(w->H[m - 1], n / 2, n, x, out);
}
static void inv_lowpassfilter_(orc_wavenet *w, const double *y, unsigned n, double *out) {
    const unsigned m = ilog2u(n);
    if (!w->hasCachedOperators || w->nCachedOperators <= m) cacheOperators_(w, m);
    matvec_t(w->L[m], n, 2 * n, y, out);
}
static void inv_highpassfilter_(orc_wavenet *w, const double *y, unsigned n, double *out) {
    const unsigned m = ilog2u(n);
    if (!w->hasCachedOperators || w->nCachedOperators <= m) cacheOperators_(w, m);
    matvec_t(w->H[m], n, 2 * n, y, out);
}
static const double *lowpassweight_(orc_wavenet *w, unsigned level, unsigned filt) {
    /* Wavenet.cxx:901-908 */
    if (!w->hasCachedWeights || w->nCachedWeights <= level) cacheWeights_(w, level);
    return w->WL[level][filt];
}
static const double *highpassweight_(orc_wavenet *w, unsigned level, unsigned filt) {
    /* Wavenet.cxx:910-917 */
    if (!w->hasCachedWeights || w->nCachedWeights <= level) cacheWeights_(w, level);
    return w->WH[level][filt];
}

/* Wavenet.h:870-886 */
static double frobeniusProduct(const double *A, const double *B, size_t nelem) {
    double acc = 0;
    for (size_t i = 0; i < nelem; i++) acc += A[i] * B[i];
    return acc;
}
static void outerProduct(const double *a, unsigned na, const double *b, unsigned nb, double *E) {
    for (unsigned c = 0; c < nb; c++) {
        for (unsigned r = 0; r < na; r++) E[r + (size_t)c * na] = a[r] * b[c];
    }
}

/* 1D ------------------------------------------------------------------------ */

void orc_forward_1d(orc_wavenet *w, const double *x, int n, double *LO, double *HI) {
    /* Wavenet.cxx:404-445 */
    const unsigned m = ilog2u((unsigned)n);
    double *x_current = (double *)xmalloc((size_t)n * sizeof(double));
    memcpy(x_current, x, (size_t)n * sizeof(double));
    unsigned len = (unsigned)n;
    LO[0] = 0; HI[0] = 0;
    for (unsigned i = m; i-- > 0;) {
        lowpassfilter_(w, x_current, len, LO + (1u << i));
        highpassfilter_(w, x_current, len, HI + (1u << i));
        len /= 2;
        memcpy(x_current, LO + (1u << i), len * sizeof(double));
    }
    memcpy(LO + n, x, (size_t)n * sizeof(double));      /* activations(m,0) = x      */
    for (int i = 0; i < n; i++) HI[n + i] = 0;         /* activations(m,1) = zeros  */
    free(x_current);
}

void orc_coeffs_from_activations_1d(const double *LO, const double *HI, int n, double *y) {
    /* Utilities.cxx:24-41 */
    const unsigned m = ilog2u((unsigned)n);
    for (int i = 0; i < n; i++) y[i] = 0;
    y[0] = LO[1];
    for (unsigned i = 0; i < m; i++) {
        memcpy(y + (1u << i), HI + (1u << i), ((size_t)1 << i) * sizeof(double));
    }
}

void orc_inverse_1d(orc_wavenet *w, const double *y, int n, double *xout) {
    /* Wavenet.cxx:447-472 */
    const unsigned m = ilog2u((unsigned)n);
    double *x = (double *)xmalloc((size_t)n * sizeof(double));
    double *t1 = (double *)xmalloc((size_t)n * sizeof(double));
    double *t2 = (double *)xmalloc((size_t)n * sizeof(double));
    x[0] = y[0];
    unsigned len = 1;
    for (unsigned i = 0; i < m; i++) {
        inv_lowpassfilter_(w, x, len, t1);
        inv_highpassfilter_(w, y + (1u << i), len, t2);
        len *= 2;
        for (unsigned k = 0; k < len; k++) x[k] = t1[k] + t2[k];
    }
    memcpy(xout, x, (size_t)n * sizeof(double));
    free(x); free(t1); free(t2);
}

void orc_backprop_1d(orc_wavenet *w, const double *delta, int n,
                     const double *LO, const double *HI, double *gradient, double *delta_in) {
    /* Wavenet.cxx:474-556 */
    (void)HI;
    const int N = w->N;
    const unsigned m = ilog2u((unsigned)n);
    if (!w->hasCachedWeights) cacheWeights_(w, m);

    for (int k = 0; k < N; k++) gradient[k] = 0;
    double *delta_LP = (double *)xcalloc((size_t)n, sizeof(double));
    double *delta_HP = (double *)xcalloc((size_t)n, sizeof(double));
    double *t1 = (double *)xmalloc((size_t)n * sizeof(double));
    double *t2 = (double *)xmalloc((size_t)n * sizeof(double));
    double *error = (double *)xmalloc((size_t)(n / 2 ? n / 2 : 1) * (size_t)n * sizeof(double));

    delta_LP[0] = delta[0];
    if (n > 1) delta_HP[0] = delta[1];
    unsigned len = 1; /* length of delta_LP / delta_HP at level i */

    for (unsigned i = 0; i < m; i++) {
        const double *activ_LP = LO + (2u << i); /* activations(i + 1, 0), length 2^(i+1) */
        const unsigned alen = 2u << i;

        outerProduct(delta_LP, len, activ_LP, alen, error);
        for (int k = 0; k < N; k++) {
            gradient[k] += frobeniusProduct(lowpassweight_(w, i, (unsigned)k), error, (size_t)len * alen);
        }
        outerProduct(delta_HP, len, activ_LP, alen, error);
        for (int k = 0; k < N; k++) {
            gradient[k] += frobeniusProduct(highpassweight_(w, i, (unsigned)k), error, (size_t)len * alen);
        }

        inv_lowpassfilter_(w, delta_LP, len, t1);
        inv_highpassfilter_(w, delta_HP, len, t2);
        len *= 2;
        for (unsigned k = 0; k < len; k++) delta_LP[k] = t1[k] + t2[k];
        if (i != m - 1) {
            memcpy(delta_HP, delta + len, len * sizeof(double)); /* delta(span(2^(i+1), 2^(i+2)-1)) */
        } else {
            for (unsigned k = 0; k < len; k++) delta_HP[k] = 0;
        }
    }
    memcpy(delta_in, delta_LP, (size_t)n * sizeof(double));
    free(delta_LP); free(delta_HP); free(t1); free(t2); free(error);
}

/* 2D ------------------------------------------------------------------------ */

void orc_forward_2d(orc_wavenet *w, const double *X, int R, int C,
                    double *ActRows, double *ActCols, double *Yout) {
    /* Wavenet.cxx:562-622 and :200-203 */
    double *Y = (double *)xmalloc((size_t)R * C * sizeof(double));
    memcpy(Y, X, (size_t)R * C * sizeof(double));
    double *x = (double *)xmalloc((size_t)(R > C ? R : C) * sizeof(double));
    double *y = (double *)xmalloc((size_t)(R > C ? R : C) * sizeof(double));

    for (int irow = 0; irow < R; irow++) {
        for (int c = 0; c < C; c++) x[c] = Y[irow + (size_t)c * R];
        double *LO = ActRows + (size_t)irow * 4 * C, *HI = LO + 2 * C;
        orc_forward_1d(w, x, C, LO, HI);
        orc_coeffs_from_activations_1d(LO, HI, C, y);
        for (int c = 0; c < C; c++) Y[irow + (size_t)c * R] = y[c];
    }
    for (int icol = 0; icol < C; icol++) {
        memcpy(x, Y + (size_t)icol * R, (size_t)R * sizeof(double));
        double *LO = ActCols + (size_t)icol * 4 * R, *HI = LO + 2 * R;
        orc_forward_1d(w, x, R, LO, HI);
        orc_coeffs_from_activations_1d(LO, HI, R, y);
        memcpy(Y + (size_t)icol * R, y, (size_t)R * sizeof(double));
    }
    memcpy(Yout, Y, (size_t)R * C * sizeof(double));
    free(Y); free(x); free(y);
}

void orc_inverse_2d(orc_wavenet *w, const double *Y, int R, int C, double *Xout) {
    /* Wavenet.cxx:624-663 */
    double *X = (double *)xmalloc((size_t)R * C * sizeof(double));
    memcpy(X, Y, (size_t)R * C * sizeof(double));
    double *x = (double *)xmalloc((size_t)(R > C ? R : C) * sizeof(double));
    double *t = (double *)xmalloc((size_t)(R > C ? R : C) * sizeof(double));
    for (int icol = 0; icol < C; icol++) {
        memcpy(x, X + (size_t)icol * R, (size_t)R * sizeof(double));
        orc_inverse_1d(w, x, R, t);
        memcpy(X + (size_t)icol * R, t, (size_t)R * sizeof(double));
    }
    for (int irow = 0; irow < R; irow++) {
        for (int c = 0; c < C; c++) x[c] = X[irow + (size_t)c * R];
        orc_inverse_1d(w, x, C, t);
        for (int c = 0; c < C; c++) X[irow + (size_t)c * R] = t[c];
    }
    memcpy(Xout, X, (size_t)R * C * sizeof(double));
    free(X); free(x); free(t);
}

void orc_backprop_2d(orc_wavenet *w, const double *Delta, int R, int C,
                     const double *ActRows, const double *ActCols, double *gradient) {
    /* Wavenet.cxx:665-711 */
    const int N = w->N;
    double *X = (double *)xmalloc((size_t)R * C * sizeof(double));
    memcpy(X, Delta, (size_t)R * C * sizeof(double));
    for (int k = 0; k < N; k++) gradient[k] = 0;
    double *g = (double *)xmalloc((size_t)N * sizeof(double));
    double *x = (double *)xmalloc((size_t)(R > C ? R : C) * sizeof(double));
    double *din = (double *)xmalloc((size_t)(R > C ? R : C) * sizeof(double));

    for (int icol = 0; icol < C; icol++) {
        const double *LO = ActCols + (size_t)icol * 4 * R, *HI = LO + 2 * R;
        memcpy(x, X + (size_t)icol * R, (size_t)R * sizeof(double));
        orc_backprop_1d(w, x, R, LO, HI, g, din);
        for (int k = 0; k < N; k++) gradient[k] += g[k];
        memcpy(X + (size_t)icol * R, din, (size_t)R * sizeof(double));
    }
    for (int irow = 0; irow < R; irow++) {
        const double *LO = ActRows + (size_t)irow * 4 * C, *HI = LO + 2 * C;
        for (int c = 0; c < C; c++) x[c] = X[irow + (size_t)c * R];
        orc_backprop_1d(w, x, C, LO, HI, g, din);
        for (int k = 0; k < N; k++) gradient[k] += g[k];
    }
    free(X); free(g); free(x); free(din);
}

/* learning ------------------------------------------------------------------ */

double orc_cost(orc_wavenet *w, const double *Y, int n) {
    /* Wavenet.cxx:256-276 */
    const double S = orc_sparse_term(Y, n);
    const double Rg = w->lambda * orc_reg_term(w->filter, w->N, w->wavelet);
    return S + Rg;
}

static void update_(orc_wavenet *w, const double *gradient) {
    /* Wavenet.cxx:753-765 */
    const int N = w->N;
    const unsigned steps = (unsigned)w->costLog.n - 1;
    const double effectiveInertia =
        (w->inertiaTimeScale > 0. ? w->inertia * (1. - exp(-(float)steps / w->inertiaTimeScale)) : w->inertia);
    for (int i = 0; i < N; i++) w->momentum[i] = w->momentum[i] * effectiveInertia; /* scaleMomentum_ */
    for (int i = 0; i < N; i++) w->momentum[i] += -w->alpha * gradient[i];          /* addMomentum_   */
    double *nf = (double *)xmalloc((size_t)N * sizeof(double));
    for (int i = 0; i < N; i++) nf[i] = w->filter[i] + w->momentum[i];
    orc_set_filter(w, nf, N);
    free(nf);
}

void orc_flush_batch_queue(orc_wavenet *w) {
    /* Wavenet.cxx:717-740 */
    const int N = w->N;
    const size_t size = N ? w->batchQueue.n / (size_t)N : 0;
    if (!size) return;
    double *gradient = (double *)xcalloc((size_t)N, sizeof(double));
    for (size_t i = 0; i < size; i++) {
        for (int k = 0; k < N; k++) gradient[k] += w->batchQueue.d[i * (size_t)N + (size_t)k];
    }
    for (int k = 0; k < N; k++) gradient[k] /= (float)size;
    update_(w, gradient);
    w->costLog.d[w->costLog.n - 1] /= (float)size;
    dvec_push(&w->costLog, 0);
    w->batchQueue.n = 0;
    free(gradient);
}

int orc_train(orc_wavenet *w, const double *X, int R, int C) {
    /* Wavenet.cxx:184-242 */
    const int N = w->N;
    double *ActRows = (double *)xmalloc((size_t)R * 4 * C * sizeof(double));
    double *ActCols = (double *)xmalloc((size_t)C * 4 * R * sizeof(double));
    double *Y = (double *)xmalloc((size_t)R * C * sizeof(double));
    double *delta = (double *)xmalloc((size_t)R * C * sizeof(double));
    double *gS = (double *)xmalloc((size_t)N * sizeof(double));
    double *gR = (double *)xmalloc((size_t)N * sizeof(double));

    orc_forward_2d(w, X, R, C, ActRows, ActCols, Y);                 /* :196-203 */
    orc_sparse_term_deriv(Y, R * C, delta);                          /* :207 (vectorise = column-major) */
    orc_backprop_2d(w, delta, R, C, ActRows, ActCols, gS);           /* :212 */
    orc_reg_term_deriv(w->filter, N, w->wavelet, gR);                /* :216 */
    for (int k = 0; k < N; k++) gR[k] = w->lambda * gR[k];
    for (int k = 0; k < N; k++) gS[k] = gS[k] + gR[k];               /* :219 */
    dvec_push_n(&w->batchQueue, gS, N);                              /* :223 */
    w->costLog.d[w->costLog.n - 1] += orc_cost(w, Y, R * C);         /* :227 */
    if (w->batchQueue.n / (size_t)N >= w->batchSize) orc_flush_batch_queue(w); /* :230 */

    free(ActRows); free(ActCols); free(Y); free(delta); free(gS); free(gR);
    return 1;
}

static double norm2(const double *v, int n) {
    double s = 0;
    for (int i = 0; i < n; i++) s += v[i] * v[i];
    return sqrt(s);
}

long orc_coach_run(orc_wavenet *w, const double *init_filter, int N,
                   const double *examples, int R, int C,
                   int numEvents, unsigned numEpochs, unsigned flags, double targetPrecision) {
    /* Coach.cxx:119-274, one initialisation */
    const int useAdaptiveLearningRate = (flags & 1u) != 0;
    const int useAdaptiveBatchSize = (flags & 2u) != 0;
    const int useSimulatedAnnealing = (flags & 4u) != 0;
    const unsigned useLastN = 10;
    const double lambdaBare = w->lambda;
    long trained = 0;

    orc_clear(w);                       /* :139 */
    orc_set_filter(w, init_filter, N);  /* :144 with an explicit point */

    int done = 0;
    unsigned tail = 0;
    unsigned currentCostLogSize = 0, previousCostLogSize = 0;

    for (unsigned epoch = 0; epoch < numEpochs; epoch++) {
        int event = 0;
        do {
            if (useSimulatedAnnealing) { /* :176-180, f is computed through a float */
                const double f = (event + (int)epoch * numEvents) / (float)(numEvents * (int)numEpochs);
                const double effectiveLambda = lambdaBare * f / sq_d(2 - f);
                orc_set_lambda(w, effectiveLambda);
            }
            const int status = orc_train(w, examples + (size_t)event * R * C, R, C); /* :183 */
            trained++;
            if (!status) { done = 1; break; }

            if (useAdaptiveLearningRate || useAdaptiveBatchSize) { /* :192-250 */
                previousCostLogSize = currentCostLogSize;
                currentCostLogSize = (unsigned)w->costLog.n;
                const int changed = (currentCostLogSize != previousCostLogSize);
                if (changed && ++tail > useLastN) {
                    const unsigned filterLogSize = (unsigned)w->filterLogEntries;
                    const double *FL = w->filterLog.d;
                    double meanStepSize = 0;
                    double *step = (double *)xmalloc((size_t)N * sizeof(double));
                    for (unsigned i = 0; i < useLastN; i++) {
                        const double *fa = FL + (size_t)(filterLogSize - useLastN + i) * N;
                        const double *fb = FL + (size_t)(filterLogSize - useLastN + i - 1) * N;
                        for (int k = 0; k < N; k++) step[k] = fa[k] - fb[k];
                        meanStepSize += norm2(step, N);
                    }
                    meanStepSize /= (float)useLastN;
                    {
                        const double *fa = FL + (size_t)(filterLogSize - 1) * N;
                        const double *fb = FL + (size_t)(filterLogSize - 1 - useLastN) * N;
                        for (int k = 0; k < N; k++) step[k] = fa[k] - fb[k];
                    }
                    const double totalStepSize = norm2(step, N);
                    free(step);
                    if (targetPrecision != -1 && meanStepSize < targetPrecision && !useSimulatedAnnealing) {
                        done = 1;
                    } else if (totalStepSize < meanStepSize) {
                        if (useAdaptiveBatchSize) orc_set_batch_size(w, 2 * w->batchSize);
                        if (useAdaptiveLearningRate) orc_set_alpha(w, (1. / 2.) * w->alpha);
                        tail = 0;
                    }
                }
            }
            ++event;
        } while (!done && (event < numEvents || numEvents < 0));
        if (done) break;
    }
    orc_cost_log_pop_back(w); /* :274 */
    return trained;
}

/* ------------------------------------------------------------------------ */
/* banded restatement                                                        */
/* ------------------------------------------------------------------------ */
/*
 * Both operator constructions place tap j of row r at column
 * (N/2 + 2r - j) mod 2^(s+1) (by-indices: start column (N/2 + 2r) mod nCols,
 * one column to the left per tap, MatrixOperator.cxx:83-85,107-109; by-rows:
 * f[N-1-i] at (i - N/2 + 1) mod nCols rotated by 2 per row, :44-58), adding
 * when taps fold onto one column.  The weight matrix of tap k is the 0/1
 * indicator of those positions; for the high-pass operator the taps are
 * b_j = (-1)^j a_{N-1-j}, so tap k of `a` sits at j = N-1-k with sign
 * (-1)^(N-1-k).
 */

static inline unsigned wrapidx(long long v, unsigned len) { /* len is a power of two */
    return (unsigned)(v & (long long)(len - 1));
}

static void band_step_forward(const double *a, const double *b, int N,
                              const double *in, unsigned len, double *lo, double *hi) {
    const unsigned half = len / 2;
    for (unsigned r = 0; r < half; r++) {
        double sl = 0, sh = 0;
        for (int j = 0; j < N; j++) {
            const double v = in[wrapidx((long long)N / 2 + 2 * (long long)r - j, len)];
            sl += a[j] * v;
            sh += b[j] * v;
        }
        lo[r] = sl;
        hi[r] = sh;
    }
}

/* out[c] = sum over (r,j) with (N/2+2r-j) mod len == c of a[j] lo[r] + b[j] hi[r]; len = 2*half */
static void band_step_inverse(const double *a, const double *b, int N,
                              const double *lo, const double *hi, unsigned half, double *out) {
    const unsigned len = 2 * half;
    for (unsigned c = 0; c < len; c++) out[c] = 0;
    for (unsigned r = 0; r < half; r++) {
        for (int j = 0; j < N; j++) {
            const unsigned c = wrapidx((long long)N / 2 + 2 * (long long)r - j, len);
            out[c] += a[j] * lo[r] + b[j] * hi[r];
        }
    }
}

void orb_forward_1d(const double *a, int N, const double *x, int n, double *P, double *y) {
    const unsigned m = ilog2u((unsigned)n);
    double *b = (double *)xmalloc((size_t)N * sizeof(double));
    orc_highpass_filter(a, N, b);
    double *cur = (double *)xmalloc((size_t)n * sizeof(double));
    double *lo = (double *)xmalloc((size_t)n * sizeof(double));
    memcpy(cur, x, (size_t)n * sizeof(double));
    for (int i = 0; i < n; i++) { P[i] = 0; y[i] = 0; }
    unsigned len = (unsigned)n;
    for (unsigned i = m; i-- > 0;) {
        band_step_forward(a, b, N, cur, len, lo, y + (1u << i));
        len /= 2;
        memcpy(P + (1u << i), lo, len * sizeof(double));
        memcpy(cur, lo, len * sizeof(double));
    }
    y[0] = (n > 1) ? P[1] : x[0];
    free(b); free(cur); free(lo);
}

void orb_inverse_1d(const double *a, int N, const double *y, int n, double *x) {
    const unsigned m = ilog2u((unsigned)n);
    double *b = (double *)xmalloc((size_t)N * sizeof(double));
    orc_highpass_filter(a, N, b);
    double *cur = (double *)xmalloc((size_t)n * sizeof(double));
    double *nxt = (double *)xmalloc((size_t)n * sizeof(double));
    cur[0] = y[0];
    unsigned len = 1;
    for (unsigned i = 0; i < m; i++) {
        band_step_inverse(a, b, N, cur, y + (1u << i), len, nxt);
        len *= 2;
        memcpy(cur, nxt, len * sizeof(double));
    }
    memcpy(x, cur, (size_t)n * sizeof(double));
    free(b); free(cur); free(nxt);
}

void orb_backprop_1d(const double *a, int N, const double *delta, const double *x,
                     const double *P, int n, double *grad, double *delta_in) {
    const unsigned m = ilog2u((unsigned)n);
    double *b = (double *)xmalloc((size_t)N * sizeof(double));
    orc_highpass_filter(a, N, b);
    double *dLP = (double *)xcalloc((size_t)n, sizeof(double));
    double *dHP = (double *)xcalloc((size_t)n, sizeof(double));
    double *nxt = (double *)xmalloc((size_t)n * sizeof(double));
    for (int k = 0; k < N; k++) grad[k] = 0;
    dLP[0] = delta[0];
    if (n > 1) dHP[0] = delta[1];
    unsigned len = 1;
    for (unsigned i = 0; i < m; i++) {
        const unsigned alen = 2 * len;
        const double *act = (i + 1 == m) ? x : P + alen; /* activations(i+1, 0) */
        for (int k = 0; k < N; k++) {
            double sl = 0, sh = 0;
            for (unsigned r = 0; r < len; r++) {
                sl += dLP[r] * act[wrapidx((long long)N / 2 + 2 * (long long)r - k, alen)];
                /* tap k of `a` is tap j = N-1-k of `b`: column N/2 + 2r - (N-1-k) */
                sh += dHP[r] * act[wrapidx((long long)N / 2 + 2 * (long long)r - (N - 1 - k), alen)];
            }
            grad[k] += sl;
            grad[k] += pow(-1, N - 1 - k) * sh;
        }
        band_step_inverse(a, b, N, dLP, dHP, len, nxt);
        len = alen;
        memcpy(dLP, nxt, len * sizeof(double));
        if (i != m - 1) memcpy(dHP, delta + len, len * sizeof(double));
        else for (unsigned k = 0; k < len; k++) dHP[k] = 0;
    }
    if (delta_in) memcpy(delta_in, dLP, (size_t)n * sizeof(double));
    free(b); free(dLP); free(dHP); free(nxt);
}

void orb_forward_2d(const double *a, int N, const double *X, int R, int C,
                    double *Prow, double *Y1, double *Pcol, double *Y) {
    const int L = R > C ? R : C;
    double *x = (double *)xmalloc((size_t)L * sizeof(double));
    double *p = (double *)xmalloc((size_t)L * sizeof(double));
    double *y = (double *)xmalloc((size_t)L * sizeof(double));
    for (int r = 0; r < R; r++) {
        for (int c = 0; c < C; c++) x[c] = X[r + (size_t)c * R];
        orb_forward_1d(a, N, x, C, p, y);
        for (int c = 0; c < C; c++) { Prow[r + (size_t)c * R] = p[c]; Y1[r + (size_t)c * R] = y[c]; }
    }
    for (int c = 0; c < C; c++) {
        orb_forward_1d(a, N, Y1 + (size_t)c * R, R, Pcol + (size_t)c * R, Y + (size_t)c * R);
    }
    free(x); free(p); free(y);
}

void orb_inverse_2d(const double *a, int N, const double *Y, int R, int C, double *X) {
    const int L = R > C ? R : C;
    double *x = (double *)xmalloc((size_t)L * sizeof(double));
    double *t = (double *)xmalloc((size_t)L * sizeof(double));
    for (int c = 0; c < C; c++) orb_inverse_1d(a, N, Y + (size_t)c * R, R, X + (size_t)c * R);
    for (int r = 0; r < R; r++) {
        for (int c = 0; c < C; c++) x[c] = X[r + (size_t)c * R];
        orb_inverse_1d(a, N, x, C, t);
        for (int c = 0; c < C; c++) X[r + (size_t)c * R] = t[c];
    }
    free(x); free(t);
}

void orb_backprop_2d(const double *a, int N, const double *Delta, const double *X,
                     const double *Prow, const double *Y1, const double *Pcol,
                     int R, int C, double *grad) {
    const int L = R > C ? R : C;
    double *D1 = (double *)xmalloc((size_t)R * C * sizeof(double));
    double *g = (double *)xmalloc((size_t)N * sizeof(double));
    double *d = (double *)xmalloc((size_t)L * sizeof(double));
    double *x = (double *)xmalloc((size_t)L * sizeof(double));
    double *p = (double *)xmalloc((size_t)L * sizeof(double));
    for (int k = 0; k < N; k++) grad[k] = 0;
    for (int c = 0; c < C; c++) {
        orb_backprop_1d(a, N, Delta + (size_t)c * R, Y1 + (size_t)c * R, Pcol + (size_t)c * R, R, g,
                        D1 + (size_t)c * R);
        for (int k = 0; k < N; k++) grad[k] += g[k];
    }
    for (int r = 0; r < R; r++) {
        for (int c = 0; c < C; c++) {
            d[c] = D1[r + (size_t)c * R];
            x[c] = X[r + (size_t)c * R];
            p[c] = Prow[r + (size_t)c * R];
        }
        orb_backprop_1d(a, N, d, x, p, C, g, NULL);
        for (int k = 0; k < N; k++) grad[k] += g[k];
    }
    free(D1); free(g); free(d); free(x); free(p);
}

int orb_train_batch(const double *a, int N, const double *X, int B, int R, int C,
                    double *gsum, double *ssum) {
    const size_t n = (size_t)R * C;
    double *Prow = (double *)xmalloc(n * sizeof(double));
    double *Y1 = (double *)xmalloc(n * sizeof(double));
    double *Pcol = (double *)xmalloc(n * sizeof(double));
    double *Y = (double *)xmalloc(n * sizeof(double));
    double *D = (double *)xmalloc(n * sizeof(double));
    double *g = (double *)xmalloc((size_t)N * sizeof(double));
    int bad = 0;
    for (int k = 0; k < N; k++) gsum[k] = 0;
    *ssum = 0;
    for (int e = 0; e < B; e++) {
        const double *Xe = X + (size_t)e * n;
        orb_forward_2d(a, N, Xe, R, C, Prow, Y1, Pcol, Y);
        const double S = orc_sparse_term(Y, (int)n);
        orc_sparse_term_deriv(Y, (int)n, D);
        orb_backprop_2d(a, N, D, Xe, Prow, Y1, Pcol, R, C, g);
        int ok = isfinite(S);
        for (int k = 0; k < N; k++) ok = ok && isfinite(g[k]);
        if (!ok) bad++;
        for (int k = 0; k < N; k++) gsum[k] += g[k];
        *ssum += S;
    }
    free(Prow); free(Y1); free(Pcol); free(Y); free(D); free(g);
    return bad;
}
