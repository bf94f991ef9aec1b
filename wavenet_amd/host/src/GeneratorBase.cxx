#include <algorithm>
#include <cmath>
#include <vector>

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

// exp(-i / 128), i = 0 .. 4096: the table behind GaussianGenerator on host and device (wn_generate_gaussian)
static const std::vector<double>& gaussExpTable() {
    static std::vector<double> tab;
    if (tab.empty()) {
        tab.resize(4097);
        for (int i = 0; i <= 4096; i++) tab[(std::size_t)i] = std::exp(-(double)i / 128.0);
    }
    return tab;
}
static double tableExpNeg(const double* tab, double t) {
    const double x = t * 128.0;
    if (!(x < 4096.0)) return 0.0;
    const int i = (int)x;
    const double fr = x - (double)i;
    return tab[i] + fr * (tab[i + 1] - tab[i]);
}

const Mat& GaussianGenerator::next() {
    if (!check_()) return m_data;
    const std::uint64_t e = m_event++;
    const std::size_t R = m_shape[0], C = m_shape[1];
    const double* tab = gaussExpTable().data();
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
                m_data(r, c) += tableExpNeg(tab, 0.5 * (dr * dr + dc * dc));
            }
        }
    }
    return m_data;
}

bool GaussianGenerator::nextOnDevice(wn_context* ctx, double* X_dev, unsigned count) {
    if (!check_() || !ctx || count > 65535u) return false;
    if (wn_generate_gaussian(ctx, m_seed, (long long)m_event, (int)count, (int)m_shape[0], (int)m_shape[1], gaussExpTable().data(), X_dev) != WN_OK)
        return false;
    m_event += count;
    return true;
}

// ---- calorimeter images ---------------------------------------------------------------------------------
static const int kGaussOffsets[20] = {-3, -2, -2, -1, -1, -1, -1, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3};

// Inverse CDF of Poisson(mean) sampled at (i + 0.5) / 4096, and the unit-mean exponential spectrum at (i + 0.5) / 1024
// (generators.py: _poisson_table, _EXP_TABLE).
void CalorimeterGenerator::tables_() {
    const std::size_t R = m_shape[0], C = m_shape[1];
    const double mean = m_meanDeposits > 0 ? m_meanDeposits : std::max(1.0, 200.0 * (double)(R * C) / 4096.0);
    if (mean == m_tablesFor) return;
    const int kmax = (int)(mean + 12 * std::sqrt(mean) + 20);
    std::vector<double> cdf((std::size_t)kmax + 1);
    double logp = -mean, run = 0;
    for (int k = 0; k <= kmax; k++) {
        if (k) logp = logp + std::log(mean / k);
        run += std::exp(logp);
        cdf[(std::size_t)k] = run;
    }
    m_poisson.resize(4096);
    for (int i = 0; i < 4096; i++) {
        const double u = (i + 0.5) / 4096;
        m_poisson[(std::size_t)i] = (int)(std::lower_bound(cdf.begin(), cdf.end(), u) - cdf.begin());     // numpy.searchsorted, side "left"
    }
    m_energy.resize(1024);
    for (int i = 0; i < 1024; i++) m_energy[(std::size_t)i] = -std::log((i + 0.5) / 1024.0);
    m_tablesFor = mean;
}

const Mat& CalorimeterGenerator::next() {
    if (!check_()) return m_data;
    tables_();
    const std::uint64_t e = m_event++;
    const std::uint64_t R = m_shape[0], C = m_shape[1];
    m_data.fill(0.0);
    const int K = m_poisson[(std::size_t)(hashCounters(m_seed, {e, 1}) % 4096ull)];
    for (int d = 0; d < K; d++) {
        const std::uint64_t h = hashCounters(m_seed, {e, (std::uint64_t)(2 + d)});
        m_data((std::size_t)(h % R), (std::size_t)((h >> 20) % C)) += 0.5 * m_energy[(std::size_t)((h >> 40) % 1024ull)];
    }
    for (int j = 0; j < m_jets; j++) {
        const std::uint64_t hc = hashCounters(m_seed, {e, (std::uint64_t)(1000 + j)});
        const long long r0 = (long long)(hc % R), c0 = (long long)((hc >> 20) % C);
        for (int dd = 0; dd < m_jetDeposits; dd++) {
            const std::uint64_t hj = hashCounters(m_seed, {e, (std::uint64_t)(2000 + 100 * j + dd)});
            const int dr = kGaussOffsets[hj % 20ull], dc = kGaussOffsets[(hj >> 16) % 20ull];
            const long long r = ((r0 + dr) % (long long)R + (long long)R) % (long long)R, c = ((c0 + dc) % (long long)C + (long long)C) % (long long)C;
            m_data((std::size_t)r, (std::size_t)c) += 4.0 * m_energy[(std::size_t)((hj >> 40) % 1024ull)];
        }
    }
    for (std::size_t k = 0; k < m_data.n_elem; k++) m_data(k) = (double)(float)m_data(k);      // TH2F bin contents are floats
    return m_data;
}

bool CalorimeterGenerator::nextOnDevice(wn_context* ctx, double* X_dev, unsigned count) {
    if (!check_() || !ctx || count > 65535u) return false;
    tables_();
    if (wn_generate_calorimeter(ctx, m_seed, (long long)m_event, (int)count, (int)m_shape[0], (int)m_shape[1], m_poisson.data(),
                                m_energy.data(), m_jets, m_jetDeposits, X_dev) != WN_OK)
        return false;
    m_event += count;
    return true;
}

// ---- a fixed set of examples resident on the device ------------------------------------------------------
ResidentGenerator::ResidentGenerator(GeneratorBase& source, unsigned events) : m_source(source), m_events(events) {
    if (source.initialised()) setShape(source.shape());
}

ResidentGenerator::~ResidentGenerator() {
    if (m_ctx) {
        if (m_dev) wn_free(m_ctx, m_dev);
        wn_destroy(m_ctx);
    }
}

// The set is produced through a context of the generator's own on the caller's GPU, so that it outlives the Wavenet
// objects that train on it; device memory is good for every context of the process on that GPU.
bool ResidentGenerator::fill_(wn_context* ctx) {
    if (m_dev) return true;
    int device = 0;
    if (!ctx || wn_get_device(ctx, &device) != WN_OK || !m_source.hasDeviceForm() || !m_source.initialised()) return false;
    if (!m_initialised && !setShape(m_source.shape())) return false;
    if (wn_create(device, &m_ctx) != WN_OK) { m_ctx = nullptr; return false; }
    m_device = device;
    const std::size_t n = (std::size_t)m_shape[0] * m_shape[1];
    if (wn_malloc(m_ctx, (void**)&m_dev, (std::size_t)m_events * n * sizeof(double)) != WN_OK) { m_dev = nullptr; return false; }
    m_source.reset();
    for (unsigned e0 = 0; e0 < m_events; e0 += 32768u) {
        if (!m_source.nextOnDevice(m_ctx, m_dev + (std::size_t)e0 * n, std::min(32768u, m_events - e0))) return false;
    }
    return wn_synchronize(m_ctx) == WN_OK;       // other contexts' streams read it from now on
}

const double* ResidentGenerator::deviceSpan(wn_context* ctx, unsigned count) {
    int device = -1;
    if (!fill_(ctx) || wn_get_device(ctx, &device) != WN_OK || device != m_device || m_event + count > m_events) return nullptr;
    const double* p = m_dev + (std::size_t)m_event * m_shape[0] * m_shape[1];
    m_event += count;
    return p;
}

bool ResidentGenerator::nextOnDevice(wn_context* ctx, double* X_dev, unsigned count) {
    const std::size_t n = (std::size_t)m_shape[0] * m_shape[1];
    if (!fill_(ctx) || m_event + count > m_events) return false;
    // through the host: the caller may sit on another GPU than the set
    std::vector<double> host((std::size_t)count * n);
    if (wn_memcpy_d2h(m_ctx, host.data(), m_dev + (std::size_t)m_event * n, host.size() * sizeof(double)) != WN_OK) return false;
    if (wn_memcpy_h2d(ctx, X_dev, host.data(), host.size() * sizeof(double)) != WN_OK) return false;
    m_event += count;
    return true;
}

const Mat& ResidentGenerator::next() {
    if (!m_initialised) setShape(m_source.shape());
    if (!m_dev || m_event >= m_events) { m_data.fill(0.0); return m_data; }      // (needs a first deviceSpan / nextOnDevice to exist)
    wn_memcpy_d2h(m_ctx, m_data.memptr(), m_dev + (std::size_t)m_event * m_data.n_elem, m_data.n_elem * sizeof(double));
    m_event++;
    return m_data;
}

}  // namespace wavenet
