// Minimal column-major fp64 containers standing where the reference uses arma::Mat<double> /
// arma::Col<double>: element (r, c) lives at memptr()[r + c * n_rows], exactly Armadillo's
// layout, so buffers can be handed to the C ABI (include/wavenet_hip.h) without conversion.
#ifndef WAVENET_MATRIX_H
#define WAVENET_MATRIX_H

#include <cassert>
#include <cstddef>
#include <initializer_list>
#include <vector>

namespace wavenet {

class Mat {
public:
    Mat() {}
    Mat(std::size_t rows, std::size_t cols) : n_rows(rows), n_cols(cols), n_elem(rows * cols), m_data(rows * cols, 0.0) {}
    Mat(std::size_t rows, std::size_t cols, const double* src) : Mat(rows, cols) {
        for (std::size_t i = 0; i < n_elem; i++) m_data[i] = src[i];
    }
    virtual ~Mat() {}

    double& operator()(std::size_t r, std::size_t c) { assert(r < n_rows && c < n_cols); return m_data[r + c * n_rows]; }
    double operator()(std::size_t r, std::size_t c) const { assert(r < n_rows && c < n_cols); return m_data[r + c * n_rows]; }
    double& operator()(std::size_t i) { assert(i < n_elem); return m_data[i]; }
    double operator()(std::size_t i) const { assert(i < n_elem); return m_data[i]; }

    double* memptr() { return m_data.data(); }
    const double* memptr() const { return m_data.data(); }
    bool is_empty() const { return n_elem == 0; }
    std::size_t size() const { return n_elem; }

    void zeros(std::size_t rows, std::size_t cols) { set_size(rows, cols); fill(0.0); }
    void set_size(std::size_t rows, std::size_t cols) { n_rows = rows; n_cols = cols; n_elem = rows * cols; m_data.assign(n_elem, 0.0); }
    void fill(double v) { for (double& x : m_data) x = v; }

    std::size_t n_rows = 0, n_cols = 0, n_elem = 0;

protected:
    std::vector<double> m_data;
};

class Col : public Mat {
public:
    Col() {}
    explicit Col(std::size_t n) : Mat(n, 1) {}
    Col(std::size_t n, const double* src) : Mat(n, 1, src) {}
    Col(std::initializer_list<double> v) : Mat(v.size(), 1) {
        std::size_t i = 0;
        for (double x : v) m_data[i++] = x;
    }
    explicit Col(const std::vector<double>& v) : Mat(v.size(), 1, v.data()) {}
    void zeros(std::size_t n) { Mat::zeros(n, 1); }
};

inline Col operator+(const Col& a, const Col& b) {
    assert(a.n_elem == b.n_elem);
    Col r(a.n_elem);
    for (std::size_t i = 0; i < a.n_elem; i++) r(i) = a(i) + b(i);
    return r;
}
inline Col operator-(const Col& a, const Col& b) {
    assert(a.n_elem == b.n_elem);
    Col r(a.n_elem);
    for (std::size_t i = 0; i < a.n_elem; i++) r(i) = a(i) - b(i);
    return r;
}
inline Col operator*(const Col& a, double s) {
    Col r(a.n_elem);
    for (std::size_t i = 0; i < a.n_elem; i++) r(i) = a(i) * s;
    return r;
}
inline Col operator*(double s, const Col& a) { return a * s; }

double norm(const Col& v);  // Euclidean norm, as arma::norm

}  // namespace wavenet

#endif
