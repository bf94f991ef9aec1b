// Helpers the training path needs from the reference's Utilities (include/Wavenet/Utilities.h,
// src/Utilities.cxx): radix-2 test, sq / sign, a seeded PointOnNSphere, coeffsFromActivations.
#ifndef WAVENET_UTILITIES_H
#define WAVENET_UTILITIES_H

#include <array>
#include <cmath>
#include <cstdint>
#include <vector>

#include "Wavenet/Matrix.h"

// Activations of one line: entry i holds {low-pass, high-pass} of level i; the last entry holds
// {the line itself, zeros} (reference src/Wavenet.cxx:410-442).  The reference's type is
// arma::field<arma::Col<double>> indexed (level, 0|1); here activations[level][0|1].
typedef std::vector<std::array<wavenet::Col, 2>> Activations1D_t;
// [0][row] are the row lines, [1][col] the column lines (reference src/Wavenet.cxx:569-580).
typedef std::vector<std::vector<Activations1D_t>> Activations2D_t;

namespace wavenet {

inline bool isRadix2(unsigned x) { return x != 0 && (x & (x - 1)) == 0; }

template <class T>
inline T sq(const T& x) { return x * x; }

// x == 0 ? 0 : x / sqrt(x*x), as the reference writes it (Utilities.h:69-70)
inline double sign(double x) { return x == 0 ? 0 : x / std::sqrt(x * x); }

inline unsigned ilog2(unsigned x) {
    unsigned m = 0;
    while ((1u << m) < x) m++;
    return m;
}

// Counter-based hashing shared with wavenet_amd/generators.py: splitmix64 of the seed and counters.
std::uint64_t splitmix64(std::uint64_t x);
std::uint64_t hashCounters(std::uint64_t seed, std::initializer_list<std::uint64_t> counters);
double unitFromHash(std::uint64_t h);  // [0, 1), top 53 bits

// Uniformly random point on the unit N-sphere scaled by (1 + rho * gauss), like the reference's
// PointOnNSphere (src/Utilities.cxx:7-22) but reproducible: the reference reseeds from entropy.
Col PointOnNSphere(unsigned N, double rho = 0., std::uint64_t seed = 1);

// y[0] = activations[0][0](0), y[2^i .. 2^(i+1)-1] = activations[i][1]  (src/Utilities.cxx:24-41)
Col coeffsFromActivations(const Activations1D_t& activations);
// column c of the result = coefficients of the column lines               (src/Utilities.cxx:43-58)
Mat coeffsFromActivations(const Activations2D_t& Activations);

}  // namespace wavenet

#endif
