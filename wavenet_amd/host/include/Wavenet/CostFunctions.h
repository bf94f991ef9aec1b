// Sparsity (Gini) and regularisation terms with their gradients: same free functions as the
// reference (include/Wavenet/CostFunctions.h:37-121).  The sparsity functions run on the GPU
// through wn_sparse_term / wn_sparse_term_deriv; the regularisation functions are O(N^2) host
// arithmetic on the N filter coefficients (wn_reg_term / wn_reg_term_deriv).
#ifndef WAVENET_COSTFUNCTIONS_H
#define WAVENET_COSTFUNCTIONS_H

#include "Wavenet/Matrix.h"

namespace wavenet {

double SparseTerm(const Col& c);
double SparseTerm(const Mat& C);
Col SparseTermDeriv(const Col& c);
Mat SparseTermDeriv(const Mat& C);

double RegTerm(const Col& a, const bool& doWavelet = true);
Col RegTermDeriv(const Col& a, const bool& doWavelet = true);

}  // namespace wavenet

#endif
