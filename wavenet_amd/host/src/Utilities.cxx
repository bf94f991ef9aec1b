#include "Wavenet/Utilities.h"

#include <cmath>

namespace wavenet {

double norm(const Col& v) {
    double s = 0;
    for (std::size_t i = 0; i < v.n_elem; i++) s += v(i) * v(i);
    return std::sqrt(s);
}

std::uint64_t splitmix64(std::uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    std::uint64_t z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

std::uint64_t hashCounters(std::uint64_t seed, std::initializer_list<std::uint64_t> counters) {
    std::uint64_t h = splitmix64(seed);
    for (std::uint64_t c : counters) h = splitmix64(h ^ (c * 0xD6E8FEB86659FD93ull));
    return h;
}

double unitFromHash(std::uint64_t h) { return (double)(h >> 11) * (1.0 / 9007199254740992.0); }

static double gaussFromHash(std::uint64_t seed, std::uint64_t i) {
    const double u1 = unitFromHash(hashCounters(seed, {i, 0})) + 1.0 / 9007199254740992.0;  // (0, 1]
    const double u2 = unitFromHash(hashCounters(seed, {i, 1}));
    return std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2);
}

Col PointOnNSphere(unsigned N, double rho, std::uint64_t seed) {
    Col coords(N);
    for (unsigned i = 0; i < N; i++) coords(i) = gaussFromHash(seed, i);
    const double scale = (1 + gaussFromHash(seed, N) * rho) / norm(coords);
    for (unsigned i = 0; i < N; i++) coords(i) *= scale;
    return coords;
}

Col coeffsFromActivations(const Activations1D_t& activations) {
    const unsigned m = (unsigned)activations.size() - 1;
    Col y((std::size_t)1 << m);
    y(0) = activations[0][0](0);
    for (unsigned i = 0; i < m; i++) {
        for (std::size_t k = 0; k < ((std::size_t)1 << i); k++) y(((std::size_t)1 << i) + k) = activations[i][1](k);
    }
    return y;
}

Mat coeffsFromActivations(const Activations2D_t& Activations) {
    const std::size_t nRows = Activations.at(0).size();
    const std::size_t nCols = Activations.at(1).size();
    Mat Y(nRows, nCols);
    for (std::size_t icol = 0; icol < nCols; icol++) {
        const Col y = coeffsFromActivations(Activations.at(1).at(icol));
        for (std::size_t r = 0; r < nRows; r++) Y(r, icol) = y(r);
    }
    return Y;
}

}  // namespace wavenet
