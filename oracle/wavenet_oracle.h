/*
 * wavenet_oracle.h -- CPU restatement of the Sardingfish/Wavenet training path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the shipped
 * library: it is the checker the parity tests, the smoke test and the
 * cpu_baseline leg of the benchmark compare the HIP path against.
 *
 * PARITY UNPINNED: the reference needs Armadillo 7.500.2 + BLAS/LAPACK
 * (scripts/downloadArmadillo.sh:10, Makefile:72-74), neither of which exists
 * in this image, and the reference ships no tests, fixtures or golden vectors.
 * This restatement is therefore pinned only by closed-form known answers,
 * finite differences and the dense-vs-banded identity, not by reference
 * output.
 *
 * Two families of functions:
 *   orc_*  dense, literal restatement: materialises the 2^s x 2^(s+1) filter
 *          matrices, the 0/+-1 weight matrices, outer products and Frobenius
 *          products exactly as the reference does.  This is also the timed
 *          "reference CPU path" stand-in.
 *   orb_*  banded restatement of the same arithmetic, O(n N) per line, for
 *          sizes the dense form cannot hold.  Checked against orc_*.
 *
 * All matrices are column-major doubles, X(r,c) at r + c*R, like arma::Mat.
 */
#ifndef WAVENET_ORACLE_H
#define WAVENET_ORACLE_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- filter matrices (src/MatrixOperator.cxx, src/HighpassOperator.cxx) -- */

/* b[i] = pow(-1,i) * a[N-1-i]                       HighpassOperator.cxx:14-18 */
void orc_highpass_filter(const double *a, int N, double *b);

/* M is 2^size x 2^(size+1), column-major, caller-allocated.                   */
void orc_operator_by_rows(const double *f, int N, unsigned size, double *M);    /* MatrixOperator.cxx:31-61  */
void orc_operator_by_indices(const double *f, int N, unsigned size, double *M); /* MatrixOperator.cxx:63-113 */
/* picks one of the two by the reference's size heuristic; returns 1 if
 * by-indices was used, 0 if by-rows.                 MatrixOperator.cxx:5-29  */
int  orc_operator_construct(const double *f, int N, unsigned size, double *M);

/* ---- cost functions (src/CostFunctions.cxx) ------------------------------ */
double orc_sparse_term(const double *c, int n);                       /* :6-37    */
void   orc_sparse_term_deriv(const double *c, int n, double *grad);   /* :49-105  */
double orc_reg_term(const double *a, int N, int doWavelet);           /* :119-257 */
void   orc_reg_term_deriv(const double *a, int N, int doWavelet, double *grad); /* :259-415 */

/* ---- Wavenet object (include/Wavenet/Wavenet.h, src/Wavenet.cxx) --------- */
typedef struct orc_wavenet orc_wavenet;

orc_wavenet *orc_wavenet_new(void);
void   orc_wavenet_free(orc_wavenet *w);
/* 1 on success, 0 if empty or odd length (Wavenet.cxx:8-36) */
int    orc_set_filter(orc_wavenet *w, const double *a, int N);
void   orc_set_lambda(orc_wavenet *w, double v);
void   orc_set_alpha(orc_wavenet *w, double v);
void   orc_set_inertia(orc_wavenet *w, double v);
void   orc_set_inertia_time_scale(orc_wavenet *w, double v);
void   orc_set_batch_size(orc_wavenet *w, unsigned v);
void   orc_do_wavelet(orc_wavenet *w, int flag);
void   orc_clear(orc_wavenet *w);                                     /* :244-250 */

int    orc_num_coeffs(const orc_wavenet *w);
double orc_lambda(const orc_wavenet *w);
double orc_alpha(const orc_wavenet *w);
unsigned orc_batch_size(const orc_wavenet *w);
void   orc_get_filter(const orc_wavenet *w, double *a);
void   orc_get_momentum(const orc_wavenet *w, double *p);
int    orc_cost_log_size(const orc_wavenet *w);
void   orc_get_cost_log(const orc_wavenet *w, double *out);
void   orc_cost_log_pop_back(orc_wavenet *w);
int    orc_filter_log_size(const orc_wavenet *w);
void   orc_get_filter_log(const orc_wavenet *w, double *out /* size x N */);
int    orc_batch_queue_size(const orc_wavenet *w);

/*
 * Activations of one line of length n = 2^m are held as two flat arrays LO and
 * HI of 2n doubles each: level i (length 2^i) lives at offset 2^i, so
 * LO[2^i + k] = activations(i,0)(k), HI[2^i + k] = activations(i,1)(k),
 * LO[n .. 2n-1] = x and HI[n .. 2n-1] = 0 (Wavenet.cxx:410-442).  Index 0 is
 * unused.
 */
void orc_forward_1d(orc_wavenet *w, const double *x, int n, double *LO, double *HI);   /* :404-445 */
void orc_coeffs_from_activations_1d(const double *LO, const double *HI, int n, double *y); /* Utilities.cxx:24-41 */
void orc_inverse_1d(orc_wavenet *w, const double *y, int n, double *x);                /* :447-472 */
/* grad[N] is overwritten; delta_in[n] receives the "errors on the input". */
void orc_backprop_1d(orc_wavenet *w, const double *delta, int n,
                     const double *LO, const double *HI,
                     double *grad, double *delta_in);                                  /* :474-556 */

/*
 * 2D: ActRows holds R lines of length C (LO then HI, 2C doubles each, i.e.
 * 4C doubles per row), ActCols holds C lines of length R (4R doubles per
 * column).  Y (R x C) receives the wavelet coefficients (Wavenet.cxx:200-203).
 */
void orc_forward_2d(orc_wavenet *w, const double *X, int R, int C,
                    double *ActRows, double *ActCols, double *Y);                      /* :562-622 */
void orc_inverse_2d(orc_wavenet *w, const double *Y, int R, int C, double *X);         /* :624-663 */
void orc_backprop_2d(orc_wavenet *w, const double *Delta, int R, int C,
                     const double *ActRows, const double *ActCols, double *grad);      /* :665-711 */

double orc_cost(orc_wavenet *w, const double *Y, int n);                               /* :256-276 */
/* One training example (R x C).  Returns 1 like the reference's bool. */
int    orc_train(orc_wavenet *w, const double *X, int R, int C);                       /* :184-242 */
void   orc_flush_batch_queue(orc_wavenet *w);                                          /* :717-740 */

/*
 * The event loop of Coach::run (src/Coach.cxx:130-278) for ONE initialisation
 * with an explicit initial filter and a fixed array of examples that every
 * epoch replays (the reference's generators and PointOnNSphere reseed from
 * entropy, so they cannot be restated deterministically).  Snapshots and
 * printing are left out.  flags: bit0 adaptive learning rate, bit1 adaptive
 * batch size, bit2 simulated annealing.  Returns the number of events trained.
 */
long orc_coach_run(orc_wavenet *w, const double *init_filter, int N,
                   const double *examples, int R, int C,
                   int numEvents, unsigned numEpochs,
                   unsigned flags, double targetPrecision);

/* ---- banded restatement --------------------------------------------------- */
/* Same maths as the dense path, O(nN) per line.  P is the low-pass pyramid:
 * P[2^i + k] = lo_i[k] for i = 0..m-1, P[0] = 0; y is the coefficient vector. */
void orb_forward_1d(const double *a, int N, const double *x, int n, double *P, double *y);
void orb_inverse_1d(const double *a, int N, const double *y, int n, double *x);
void orb_backprop_1d(const double *a, int N, const double *delta, const double *x,
                     const double *P, int n, double *grad, double *delta_in);
/* Prow: R x C (row r's pyramid at Prow[r + k*R]); Y1: row-stage coefficients;
 * Pcol: R x C (column c's pyramid at Pcol[k + c*R]); Y: coefficients.          */
void orb_forward_2d(const double *a, int N, const double *X, int R, int C,
                    double *Prow, double *Y1, double *Pcol, double *Y);
void orb_inverse_2d(const double *a, int N, const double *Y, int R, int C, double *X);
void orb_backprop_2d(const double *a, int N, const double *Delta, const double *X,
                     const double *Prow, const double *Y1, const double *Pcol,
                     int R, int C, double *grad);
/*
 * Banded batch step: sums over B examples of the sparsity gradient (gsum[N])
 * and of SparseTerm (ssum).  Returns the count of examples with a non-finite
 * cost or gradient.
 */
int orb_train_batch(const double *a, int N, const double *X, int B, int R, int C,
                    double *gsum, double *ssum);

#ifdef __cplusplus
}
#endif
#endif
