#include <cmath>

#include "Wavenet/GeneratorBase.h"
#include "Wavenet/Generators.h"
#include "wavenet_hip.h"
#include "Wavenet/Utilities.h"

namespace wavenet {

bool GeneratorBase::setShape(const std::vector<unsigned>& shape) {
    if (shape.empty() || shape.size() > 2) {
        WARNING("Only one- and two-dimensional shapes are supported.");
        return false;
    }
    std::vector<unsigned> s = shape;
    if (s.size() == 1) s.push_back(1);
    if (s[0] == 1 && s[1] > 1) { s[0] = s[1]; s[1] = 1; }  // {1, n} -> {n, 1}
    if (!isRadix2(s[0]) || !isRadix2(s[1])) {
        WARNING("Shape {%u, %u} is not radix 2.", s[0], s[1]);
        return false;
    }
    m_shape = s;
    m_data.zeros(s[0], s[1]);
    m_initialised = true;
    return true;
}

bool GeneratorBase::check_() const {
    if (!m_initialised) WARNING("Generator not initialised: no valid shape set.");
    return m_initialised;
}

const Mat& UniformGenerator::next() {
    if (!check_()) return m_data;
    const std::uint64_t e = m_event++;
    const std::size_t R = m_shape[0], C = m_shape[1];
    for (std::size_t r = 0; r < R; r++) {
        for (std::size_t c = 0; c < C; c++) m_data(r, c) = unitFromHash(hashCounters(m_seed, {e, r * C + c}));
    }
    return m_data;
}

const Mat& NeedleGenerator::next() {
    if (!check_()) return m_data;
    const std::uint64_t e = m_event++;
    const std::size_t R = m_shape[0], C = m_shape[1];
    bool any = false;
    for (std::size_t r = 0; r < R; r++) {
        for (std::size_t c = 0; c < C; c++) {
            double v = unitFromHash(hashCounters(m_seed, {e, r * C + c}));
            if (v < 0.995) v = 0.0;
            any = any || v != 0.0;
            m_data(r, c) = v;
        }
    }
    if (!any) {
        const std::size_t pos = hashCounters(m_seed ^ 0x5EEDull, {e}) % (R * C);
        m_data(pos / C, pos % C) = 0.995 + 0.005 * unitFromHash(hashCounters(m_seed ^ 0xBEEFull, {e}));
    }
    return m_data;
}

// The same sequences, a batch at a time, written by the device (wn_generate_* of wavenet_hip.h).
bool UniformGenerator::nextOnDevice(wn_context* ctx, double* X_dev, unsigned count) {
    if (!check_() || !ctx || count > 65535u) return false;
    if (wn_generate_uniform(ctx, m_seed, (long long)m_event, (int)count, (int)m_shape[0], (int)m_shape[1], X_dev) != WN_OK) return false;
    m_event += count;
    return true;
}

bool NeedleGenerator::nextOnDevice(wn_context* ctx, double* X_dev, unsigned count) {
    if (!check_() || !ctx || count > 65535u) return false;
    if (wn_generate_needle(ctx, m_seed, (long long)m_event, (int)count, (int)m_shape[0], (int)m_shape[1], 0.995, X_dev) != WN_OK)
        return false;
    m_event += count;
    return true;
}

const Mat& GaussianGenerator::next() {
    if (!check_()) return m_data;
    const std::uint64_t e = m_event++;
    const std::size_t R = m_shape[0], C = m_shape[1];
    m_data.fill(0.0);
    const unsigned bumps = 1 + (unsigned)(hashCounters(m_seed, {e, 0}) % 3);
    for (unsigned b = 0; b < bumps; b++) {
        const double mr = unitFromHash(hashCounters(m_seed, {e, 10 + b})) * R;
        const double mc = unitFromHash(hashCounters(m_seed, {e, 20 + b})) * C;
        const double sr = 0.5 + 0.15 * R * unitFromHash(hashCounters(m_seed, {e, 30 + b}));
        const double sc = 0.5 + 0.15 * C * unitFromHash(hashCounters(m_seed, {e, 40 + b}));
        for (std::size_t r = 0; r < R; r++) {
            for (std::size_t c = 0; c < C; c++) {
                const double dr = ((double)r - mr) / sr, dc = (C > 1) ? ((double)c - mc) / sc : 0.0;
                m_data(r, c) += std::exp(-0.5 * (dr * dr + dc * dc));
            }
        }
    }
    return m_data;
}

}  // namespace wavenet
