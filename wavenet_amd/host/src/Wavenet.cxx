#include "Wavenet/Wavenet.h"
#include "Wavenet/GeneratorBase.h"

#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstring>
#include <string>

#include <sys/stat.h>

#include "wavenet_hip.h"

namespace wavenet {

// -------------------------------------------------------------------------------------------------
// Device side of one Wavenet object: the wn_context and grow-only device buffers.
// -------------------------------------------------------------------------------------------------
class DeviceSession {
public:
    enum Slot { X = 0, Y, PROW, Y1, PCOL, DELTA, GRAD, S, NSLOTS };

    explicit DeviceSession(int device) {
        if (wn_create(device, &ctx) != WN_OK) {
            ERROR("Cannot create a device context on GPU %d: %s", device, wn_last_error(nullptr));
            ctx = nullptr;
        }
        for (int i = 0; i < NSLOTS; i++) { m_buf[i] = nullptr; m_cap[i] = 0; }
    }
    ~DeviceSession() {
        if (!ctx) return;
        for (int i = 0; i < NSLOTS; i++) {
            if (m_buf[i]) wn_free(ctx, m_buf[i]);
        }
        wn_destroy(ctx);
    }
    double* buf(Slot slot, std::size_t doubles) {
        if (m_cap[slot] < doubles) {
            if (m_buf[slot]) wn_free(ctx, m_buf[slot]);
            m_buf[slot] = nullptr;
            if (wn_malloc(ctx, (void**)&m_buf[slot], doubles * sizeof(double)) != WN_OK) {
                ERROR("%s", wn_last_error(ctx));
                m_cap[slot] = 0;
                return nullptr;
            }
            m_cap[slot] = doubles;
        }
        return m_buf[slot];
    }
    bool ok(int status) const {
        if (status != WN_OK) ERROR("%s (%s)", wn_last_error(ctx), wn_status_string(status));
        return status == WN_OK;
    }
    bool up(double* dst, const double* src, std::size_t n) { return dst && ok(wn_memcpy_h2d(ctx, dst, src, n * sizeof(double))); }
    bool down(double* dst, const double* src, std::size_t n) { return src && ok(wn_memcpy_d2h(ctx, dst, src, n * sizeof(double))); }

    wn_context* ctx = nullptr;

private:
    double* m_buf[NSLOTS];
    std::size_t m_cap[NSLOTS];
};

// -------------------------------------------------------------------------------------------------

Wavenet::Wavenet() {}
Wavenet::Wavenet(const double& lambda) : m_lambda(lambda) {}
Wavenet::Wavenet(const double& lambda, const double& alpha) : m_lambda(lambda), m_alpha(alpha) {}
Wavenet::Wavenet(const Wavenet& other)
    : m_lambda(other.m_lambda), m_alpha(other.m_alpha), m_inertia(other.m_inertia),
      m_inertiaTimeScale(other.m_inertiaTimeScale), m_filter(other.m_filter), m_device(other.m_device) {}
Wavenet::~Wavenet() {
    for (DeviceSession* r : m_replicas) delete r;
    delete m_session;
}

void Wavenet::setDevices(const std::vector<int>& devices, bool useRccl) {
    if (m_session) {
        WARNING("setDevices() has no effect after the first device call.");
        return;
    }
    m_devices = devices;
    m_useRccl = useRccl;
    if (!devices.empty()) m_device = devices[0];
}

template <class F>
bool Wavenet::onReplicas_(F f) {
    if (m_session && !f(*m_session)) return false;
    for (DeviceSession* r : m_replicas) {
        if (!f(*r)) return false;
    }
    return true;
}

DeviceSession& Wavenet::device_() {
    if (!m_session) {
        m_session = new DeviceSession(m_device);
        for (std::size_t r = 1; r < m_devices.size() && m_session->ctx; r++) {
            m_replicas.push_back(new DeviceSession(m_devices[r]));
            if (!m_replicas.back()->ctx) {   // without all of its replicas the object runs on the first device alone
                for (DeviceSession* q : m_replicas) delete q;
                m_replicas.clear();
                m_devices.clear();
                break;
            }
        }
        if (!m_replicas.empty() && m_useRccl) {
            std::vector<wn_context*> ctxs(1, m_session->ctx);
            for (DeviceSession* q : m_replicas) ctxs.push_back(q->ctx);
            if (!m_session->ok(wn_comm_init_all(ctxs.data(), (int)ctxs.size()))) {
                WARNING("No RCCL communicator over the replicas: their sums are added on the host instead.");
                m_useRccl = false;
            }
        }
        onReplicas_([&](DeviceSession& d) {
            if (!d.ctx || m_filter.is_empty()) return true;
            d.ok(wn_set_filter(d.ctx, m_filter.memptr(), (int)m_filter.n_elem));
            if (m_momentum.n_elem == m_filter.n_elem) d.ok(wn_set_momentum(d.ctx, m_momentum.memptr(), (int)m_momentum.n_elem));
            return true;
        });
    }
    return *m_session;
}

/// Set method(s).
bool Wavenet::setLambda(const double& lambda) { assert(lambda >= 0); m_lambda = lambda; return true; }
bool Wavenet::setAlpha(const double& alpha) { assert(alpha > 0); m_alpha = alpha; return true; }
bool Wavenet::setInertia(const double& inertia) { assert(inertia >= 0 && inertia < 1); m_inertia = inertia; return true; }
bool Wavenet::setInertiaTimeScale(const double& inertiaTimeScale) {
    assert(inertiaTimeScale > 0);
    m_inertiaTimeScale = inertiaTimeScale;
    return true;
}

bool Wavenet::setFilter(const Col& filter) {
    if (filter.is_empty()) {
        WARNING("Input filter is empty.");
        return false;
    }
    if (filter.n_rows % 2) {
        WARNING("Input filter length is not a multiple of 2.");
        return false;
    }
    if (filter.n_elem > WN_MAX_TAPS) {
        WARNING("Input filter has %d coefficients; at most %d are supported.", (int)filter.n_elem, WN_MAX_TAPS);
        return false;
    }
    if (m_pending) fetchUpdates_();      // outstanding updates belong before this entry of the filter log
    m_filter = filter;
    m_filterLog.push_back(m_filter);
    if (m_momentum.n_elem != m_filter.n_elem) m_momentum.zeros(m_filter.n_elem);
    // stands for clearCachedOperators_(): the device taps are what the operators were built from
    return onReplicas_([&](DeviceSession& d) { return !d.ctx || d.ok(wn_set_filter(d.ctx, m_filter.memptr(), (int)m_filter.n_elem)); });
}

bool Wavenet::setMomentum(const Col& momentum) {
    assert(momentum.size() == m_filter.size());
    if (m_pending) fetchUpdates_();
    m_momentum = momentum;
    return onReplicas_([&](DeviceSession& d) { return !d.ctx || d.ok(wn_set_momentum(d.ctx, m_momentum.memptr(), (int)m_momentum.n_elem)); });
}

/// Get method(s).
double Wavenet::firstCost() const { return m_costLog.size() > 0 ? m_costLog[0] : -1.; }

double Wavenet::lastCost() const {
    if (m_costLog.size() > 0) {
        if (m_costLog.back() > 0) {
            if (m_batchQueue.size() > 0) return m_costLog.back() / float(m_batchQueue.size());
            return m_costLog.back();
        } else if (m_costLog.size() > 1) {
            return m_costLog[m_costLog.size() - 2];
        }
    }
    return -1.;
}

void Wavenet::print() const {
    std::string f, p;
    for (std::size_t i = 0; i < m_filter.n_elem; i++) f += (i ? ", " : "") + std::to_string(m_filter(i));
    for (std::size_t i = 0; i < m_momentum.n_elem; i++) p += (i ? ", " : "") + std::to_string(m_momentum(i));
    std::printf("Wavenet: lambda %g, alpha %g, inertia %g (time scale %g), batch size %u, queue %zu\n", m_lambda,
                m_alpha, m_inertia, m_inertiaTimeScale, m_batchSize, m_batchQueue.size());
    std::printf("  filter   : [%s]\n  momentum : [%s]\n", f.c_str(), p.c_str());
}

/// Storage method(s).
void Wavenet::save(Snapshot snap) const {
    const std::string file = snap.file();
    DEBUG("Saving snapshot '%s'.", file.c_str());
    if (file.substr(0, 1) == "/") {                                                     // :141-144
        WARNING("File '%s' not accepted. Only accepting realtive paths.", file.c_str());
        return;
    }
    if (snap.exists()) DEBUG("File '%s' already exists. Overwriting.", file.c_str());
    if (file.find("/") != std::string::npos) {                                          // :150-156
        const std::string dir = file.substr(0, file.find_last_of("/"));
        if (!dirExists(dir)) {
            WARNING("Directory '%s' does not exist. Creating it.", dir.c_str());
            for (std::size_t at = dir.find('/'); ; at = dir.find('/', at + 1)) {
                const std::string part = dir.substr(0, at);
                if (!part.empty() && !dirExists(part)) ::mkdir(part.c_str(), 0777);
                if (at == std::string::npos) break;
            }
        }
    }
    snap << *this;
}

void Wavenet::load(Snapshot snap) {
    DEBUG("Loading snapshot '%s'.", snap.file().c_str());
    if (!fileExists(snap.file())) {
        WARNING("File '%s' doesn't exists.", snap.file().c_str());
        return;
    }
    snap >> *this;
    // the loaded taps replace whatever the device-side operators were built from
    onReplicas_([&](DeviceSession& d) {
        if (!d.ctx || m_filter.is_empty()) return true;
        d.ok(wn_set_filter(d.ctx, m_filter.memptr(), (int)m_filter.n_elem));
        if (m_momentum.n_elem == m_filter.n_elem) d.ok(wn_set_momentum(d.ctx, m_momentum.memptr(), (int)m_momentum.n_elem));
        return true;
    });
}

bool Wavenet::checkShape_(unsigned nRows, unsigned nCols) const {
    if (m_filter.is_empty()) {
        ERROR("No filter set.");
        return false;
    }
    if (!isRadix2(nRows) || !isRadix2(nCols)) {
        ERROR("Shape {%u, %u} is not radix 2.", nRows, nCols);
        return false;
    }
    return true;
}

/// High-level learning method(s).
bool Wavenet::queueExamples_(const double* X_dev, unsigned count, unsigned nRows, unsigned nCols, const double* lambdas) {
    DeviceSession& dev = device_();
    const int N = (int)m_filter.n_elem;
    double* dG = dev.buf(DeviceSession::GRAD, (std::size_t)count * N);
    double* dS = dev.buf(DeviceSession::S, count);
    if (!dG || !dS) return false;
    if (!dev.ok(wn_train_batch(dev.ctx, X_dev, (int)count, (int)nRows, (int)nCols, dG, dS))) return false;
    std::vector<double> G((std::size_t)count * N), S(count);
    if (!dev.down(G.data(), dG, G.size()) || !dev.down(S.data(), dS, count)) return false;

    const Col regDeriv = RegTermDeriv(m_filter, m_wavelet);
    const double reg = RegTerm(m_filter, m_wavelet);
    for (unsigned e = 0; e < count; e++) {
        const double lambda = lambdas ? lambdas[e] : m_lambda;
        bool finite = std::isfinite(S[e]);
        for (int k = 0; k < N; k++) finite = finite && std::isfinite(G[(std::size_t)e * N + k]);
        if (!finite) return diverged_();                                                // :234-241
        Col gradientCombined(N);
        for (int k = 0; k < N; k++) gradientCombined(k) = G[(std::size_t)e * N + k] + lambda * regDeriv(k);
        m_batchQueue.push_back(gradientCombined);
        m_costLog.back() += S[e] + lambda * reg;
        if (lambdas) m_lambda = lambda;
        if (m_batchQueue.size() >= m_batchSize) flushBatchQueue_();
        // the filter only changes at a flush, and a flush only happens on the last queued example
        // of this call (callers size `count` accordingly)
    }
    return true;
}

bool Wavenet::train(Mat X) {
    return trainBatch(X.memptr(), 1, (unsigned)X.n_rows, (unsigned)X.n_cols);
}

bool Wavenet::trainBatch(const double* X, unsigned B, unsigned nRows, unsigned nCols, const double* lambdas) {
    if (!checkShape_(nRows, nCols)) return false;
    if (B == 0) return true;
    DeviceSession& dev = device_();
    if (!dev.ctx) return false;
    if (!fetchUpdates_()) return false;
    const std::size_t n = (std::size_t)nRows * nCols;
    unsigned done = 0;
    if (!m_replicas.empty() && m_batchQueue.empty() && B >= m_batchSize) {
        // whole batches go over all replicas, straight from the caller's memory
        auto place = [&](unsigned r, double* dst, unsigned first, unsigned count) {
            DeviceSession& d = r ? *m_replicas[r - 1] : dev;
            return d.up(dst, X + (std::size_t)first * n, (std::size_t)count * n);
        };
        while (B - done >= m_batchSize) {
            const unsigned nb = std::min((B - done) / m_batchSize, 8192u);   // the device keeps the last 16384 log entries
            if (!trainSharded_(place, nb, done, nRows, nCols, lambdas)) return false;
            done += nb * m_batchSize;
        }
        if (done == B) return true;
    }
    double* dX = dev.buf(DeviceSession::X, (std::size_t)(B - done) * n);
    if (!dev.up(dX, X + (std::size_t)done * n, (std::size_t)(B - done) * n)) return false;
    return trainOnDevice_(dX, B - done, nRows, nCols, lambdas ? lambdas + done : nullptr);
}

bool Wavenet::trainBatchFrom(GeneratorBase& generator, unsigned B, unsigned nRows, unsigned nCols, const double* lambdas, bool deferLogs) {
    if (!generator.hasDeviceForm() || !checkShape_(nRows, nCols)) return false;
    if (B == 0) return true;
    DeviceSession& dev = device_();
    if (!dev.ctx) return false;
    unsigned done = 0;
    if (!m_replicas.empty() && m_batchQueue.empty() && B >= m_batchSize) {
        if (!fetchUpdates_()) return false;
        // every replica's share is generated where it is trained; the shares are taken in the order of the examples,
        // so the generator's sequence is the one a single device would see
        auto place = [&](unsigned r, double* dst, unsigned, unsigned count) {
            DeviceSession& d = r ? *m_replicas[r - 1] : dev;
            return generator.nextOnDevice(d.ctx, dst, count);
        };
        while (B - done >= m_batchSize) {
            const unsigned nb = std::min((B - done) / m_batchSize, 8192u);
            if (!trainSharded_(place, nb, done, nRows, nCols, lambdas)) return false;
            done += nb * m_batchSize;
        }
        if (done == B) return true;
    }
    // examples that already sit in the generator's own device memory are trained where they are
    const double* span = m_replicas.empty() ? generator.deviceSpan(dev.ctx, B - done) : nullptr;
    if (span) return trainOnDevice_(span, B - done, nRows, nCols, lambdas ? lambdas + done : nullptr, deferLogs);
    // (batches still queued may read this buffer: the generator writes it on the same stream, behind them, and a
    // reallocation waits for the device)
    double* dX = dev.buf(DeviceSession::X, (std::size_t)(B - done) * nRows * nCols);
    if (!dX || !generator.nextOnDevice(dev.ctx, dX, B - done)) return false;
    return trainOnDevice_(dX, B - done, nRows, nCols, lambdas ? lambdas + done : nullptr, deferLogs);
}

// Contiguous share [begin, end) of `total` examples for replica r of G; sizes differ by at most one.
static void shareOf(unsigned total, unsigned r, unsigned G, unsigned* begin, unsigned* end) {
    const unsigned base = total / G, extra = total % G;
    *begin = r * base + std::min(r, extra);
    *end = *begin + base + (r < extra ? 1u : 0u);
}

// The sums of all replicas added up and handed back to every one of them.
bool Wavenet::exchangeSums_() {
    DeviceSession& dev = *m_session;
    const unsigned G = (unsigned)m_replicas.size() + 1;
    if (m_useRccl) {
        std::vector<wn_context*> ctxs(1, dev.ctx);
        for (DeviceSession* q : m_replicas) ctxs.push_back(q->ctx);
        return dev.ok(wn_allreduce_sums_all(ctxs.data(), (int)G));
    }
    // on the host, in replica order: the same result whatever the timing
    const int count = (int)m_filter.n_elem + 2;
    std::vector<double> total((std::size_t)count, 0.0), part((std::size_t)count);
    for (unsigned r = 0; r < G; r++) {
        DeviceSession& d = r ? *m_replicas[r - 1] : dev;
        int c = count;
        if (!d.ok(wn_get_sums(d.ctx, part.data(), &c))) return false;
        for (int k = 0; k < count; k++) total[(std::size_t)k] += part[(std::size_t)k];
    }
    return onReplicas_([&](DeviceSession& d) {
        double* sums = nullptr;
        int c = 0;
        return d.ok(wn_sums(d.ctx, &sums, &c)) && d.up(sums, total.data(), (std::size_t)count);
    });
}

template <class Place>
bool Wavenet::trainSharded_(Place place, unsigned nb, unsigned first, unsigned nRows, unsigned nCols, const double* lambdas) {
    DeviceSession& dev = *m_session;
    const unsigned G = (unsigned)m_replicas.size() + 1;
    const std::size_t n = (std::size_t)nRows * nCols;
    const int N = (int)m_filter.n_elem;
    // every replica's shares of all nb batches, batch after batch in its X buffer
    std::vector<double*> dX(G);
    std::vector<unsigned> shareSize(G), shareBegin(G);
    for (unsigned r = 0; r < G; r++) {
        unsigned b0, b1;
        shareOf(m_batchSize, r, G, &b0, &b1);
        shareBegin[r] = b0;
        shareSize[r] = b1 - b0;
        DeviceSession& d = r ? *m_replicas[r - 1] : dev;
        dX[r] = d.buf(DeviceSession::X, std::max<std::size_t>(1, (std::size_t)nb * shareSize[r] * n));
        if (!dX[r]) return false;
    }
    for (unsigned b = 0; b < nb; b++) {
        for (unsigned r = 0; r < G; r++) {
            if (shareSize[r] && !place(r, dX[r] + (std::size_t)b * shareSize[r] * n, first + b * m_batchSize + shareBegin[r], shareSize[r]))
                return false;
        }
    }
    if (!deviceSteps_()) return false;
    const int firstLog = m_devSteps;
    for (unsigned b = 0; b < nb; b++) {
        double lambdaSum = 0;
        for (unsigned e = 0; e < m_batchSize; e++) lambdaSum += lambdas ? lambdas[first + b * m_batchSize + e] : m_lambda;
        const unsigned steps = (unsigned)m_costLog.size() - 1 + b;
        for (unsigned r = 0; r < G; r++) {
            DeviceSession& d = r ? *m_replicas[r - 1] : dev;
            if (!d.ok(wn_train_batch(d.ctx, dX[r] + (std::size_t)b * shareSize[r] * n, (int)shareSize[r], (int)nRows, (int)nCols, nullptr, nullptr)))
                return false;
        }
        if (!exchangeSums_()) return false;
        const double inertia = effectiveInertia_(steps);
        if (!onReplicas_([&](DeviceSession& d) { return d.ok(wn_apply_update(d.ctx, (double)m_batchSize, lambdaSum, m_alpha, inertia, m_wavelet)); }))
            return false;
    }
    // every replica holds the same state: the logs, the momentum and the verdict come from the first, in one readback
    std::vector<double> F((std::size_t)nb * N), C(nb);
    int lastLog = 0;
    const int state = wn_get_updates(dev.ctx, firstLog, (int)nb, &lastLog, F.data(), C.data(), m_momentum.memptr());
    if (state != WN_OK && state != WN_ERR_NONFINITE) return dev.ok(state);
    const unsigned applied = (unsigned)std::min<int>((int)nb, std::max(0, lastLog - firstLog));
    m_devSteps = lastLog;
    for (unsigned b = 0; b < applied; b++) {
        m_filterLog.push_back(Col(N, F.data() + (std::size_t)b * N));
        m_costLog.back() = C[b];
        m_costLog.push_back(0);
    }
    if (applied) m_filter = m_filterLog.back();
    if (lambdas && applied) m_lambda = lambdas[first + applied * m_batchSize - 1];
    if (state == WN_ERR_NONFINITE) return diverged_();
    return true;
}

// The device's update counter, kept on the host between fetches (one readback the first time).
bool Wavenet::deviceSteps_() {
    if (m_devSteps >= 0) return true;
    DeviceSession& dev = device_();
    int steps = 0;
    const int state = wn_num_updates(dev.ctx, &steps);
    if (state == WN_ERR_NONFINITE) return diverged_();
    if (!dev.ok(state)) return false;
    m_devSteps = steps;
    return true;
}

// The logs of the updates queued since the last fetch, the momentum, and whether a batch was refused: one readback.
// A batch with non-finite sums was not applied, nor was any batch after it.
bool Wavenet::fetchUpdates_() {
    if (!m_pending) return true;
    DeviceSession& dev = device_();
    const int N = (int)m_filter.n_elem;
    std::vector<double> F((std::size_t)m_pending * N), C(m_pending);
    int steps = 0;
    if (m_momentum.n_elem != m_filter.n_elem) m_momentum.zeros(m_filter.n_elem);
    const int state = wn_get_updates(dev.ctx, m_devSteps, (int)m_pending, &steps, F.data(), C.data(), m_momentum.memptr());
    if (state != WN_OK && state != WN_ERR_NONFINITE) { m_pending = 0; m_pendingLambda.clear(); return dev.ok(state); }
    const unsigned applied = (unsigned)std::min<int>((int)m_pending, std::max(0, steps - m_devSteps));
    for (unsigned b = 0; b < applied; b++) {
        m_filterLog.push_back(Col(N, F.data() + (std::size_t)b * N));
        m_costLog.back() = C[b];
        m_costLog.push_back(0);
    }
    if (applied) {
        m_filter = m_filterLog.back();
        const double l = m_pendingLambda[applied - 1];
        if (l == l) m_lambda = l;
    }
    m_devSteps = steps;
    m_pending = 0;
    m_pendingLambda.clear();
    if (state == WN_ERR_NONFINITE) return diverged_();                                  // :234-241
    // batches that ran on the first device alone (a pending queue came first): the replicas follow
    for (DeviceSession* r : m_replicas) {
        if (!r->ok(wn_set_filter(r->ctx, m_filter.memptr(), N)) || !r->ok(wn_set_momentum(r->ctx, m_momentum.memptr(), N))) return false;
    }
    return true;
}

bool Wavenet::sync() { return fetchUpdates_(); }

// B examples already on the device: whole batches back to back, the rest through the batch queue.
bool Wavenet::trainOnDevice_(const double* dX, unsigned B, unsigned nRows, unsigned nCols, const double* lambdas, bool deferLogs) {
    DeviceSession& dev = device_();
    const std::size_t n = (std::size_t)nRows * nCols;
    unsigned done = 0;
    while (done < B) {
        const unsigned room = m_batchSize > m_batchQueue.size() ? m_batchSize - (unsigned)m_batchQueue.size() : 1;
        if (m_batchQueue.empty() && B - done >= m_batchSize && m_batchSize >= 1) {
            // whole batches back to back on the device: train kernel, batch reduction, update kernel
            // (at most 8192 outstanding: the device keeps the last 16384 log entries)
            if (m_pending >= 8192u && !fetchUpdates_()) return false;
            if (!deviceSteps_()) return false;
            const unsigned nb = std::min((B - done) / m_batchSize, 8192u - m_pending);
            for (unsigned b = 0; b < nb; b++) {
                const double* Xb = dX + (std::size_t)(done + b * m_batchSize) * n;
                double lambdaSum = 0;
                for (unsigned e = 0; e < m_batchSize; e++)
                    lambdaSum += lambdas ? lambdas[done + b * m_batchSize + e] : m_lambda;
                const unsigned steps = (unsigned)m_costLog.size() - 1 + m_pending;
                if (!dev.ok(wn_train_step(dev.ctx, Xb, (int)m_batchSize, (int)nRows, (int)nCols, (double)m_batchSize, lambdaSum, m_alpha,
                                          effectiveInertia_(steps), m_wavelet)))
                    return false;
                m_pending++;
                m_pendingLambda.push_back(lambdas ? lambdas[done + (b + 1) * m_batchSize - 1] : std::nan(""));
            }
            done += nb * m_batchSize;
            if (!deferLogs && !fetchUpdates_()) return false;
        } else {
            // fill the pending batch (or a trailing partial one) example by example
            if (!fetchUpdates_()) return false;
            const unsigned count = std::min(room, B - done);
            if (!queueExamples_(dX + (std::size_t)done * n, count, nRows, nCols, lambdas ? lambdas + done : nullptr)) return false;
            done += count;
        }
    }
    return true;
}

// The reference's train() catches the exception a NaN raises, prints this and returns false
// (src/Wavenet.cxx:234-241); Coach::run then ends the initialisation (src/Coach.cxx:186-189).
bool Wavenet::diverged_() const {
    ERROR("Non-finite cost or gradient.");
    ERROR("Most likely due to diverging solution. Try lowering the learning rate (alpha) or the regularisation term (lambda). Exiting.");
    return false;
}

void Wavenet::clear() {
    if (m_pending) fetchUpdates_();       // (waits for what is queued; the logs are dropped below)
    m_devSteps = m_session ? 0 : -1;      // wn_clear_logs restarts the device's update counter
    scaleMomentum_(0.);
    clearFilterLog();
    clearCostLog();
    onReplicas_([&](DeviceSession& d) {
        if (d.ctx) {
            wn_set_momentum(d.ctx, m_momentum.memptr(), (int)m_momentum.n_elem);
            wn_clear_logs(d.ctx);
        }
        return true;
    });
}

/// High-level cost method(s).
double Wavenet::cost(const Col& y) {
    const double S = SparseTerm(y);
    const double R = lambda() * RegTerm(m_filter, m_wavelet);
    return S + R;
}

double Wavenet::cost(const Mat& Y) {
    const double S = SparseTerm(Y);
    const double R = lambda() * RegTerm(m_filter, m_wavelet);
    return S + R;
}

std::vector<Mat> Wavenet::costMap(const std::vector<Mat>& X, const double& range, const unsigned& Ndiv) {
    std::vector<Mat> costs(3);
    for (Mat& M : costs) M.zeros(Ndiv, Ndiv);
    const unsigned nExample = (unsigned)X.size();
    if (!nExample) return costs;
    const unsigned R = (unsigned)X[0].n_rows, C = (unsigned)X[0].n_cols;
    const std::size_t n = (std::size_t)R * C;
    DeviceSession& dev = device_();
    std::vector<double> host((std::size_t)nExample * n), S(nExample);
    for (unsigned e = 0; e < nExample; e++) std::memcpy(host.data() + e * n, X[e].memptr(), n * sizeof(double));
    double* dX = dev.buf(DeviceSession::X, host.size());
    double* dY = dev.buf(DeviceSession::Y, host.size());
    double* dS = dev.buf(DeviceSession::S, nExample);
    if (!dev.up(dX, host.data(), host.size())) return costs;
    // all grid points in one call where the library has a batched kernel for the shape ...
    const std::size_t F = (std::size_t)Ndiv * Ndiv;
    std::vector<double> grid(2 * F), Sall(F * nExample);
    for (unsigned i = 0; i < Ndiv; i++) {
        for (unsigned j = 0; j < Ndiv; j++) {
            grid[2 * ((std::size_t)i * Ndiv + j)] = (2 * j / double(Ndiv - 1) - 1) * range;
            grid[2 * ((std::size_t)i * Ndiv + j) + 1] = (2 * i / double(Ndiv - 1) - 1) * range;
        }
    }
    double* dSall = dev.buf(DeviceSession::DELTA, Sall.size());
    const int st = dSall ? wn_sparse_cost_grid(dev.ctx, dX, (int)nExample, (int)R, (int)C, grid.data(), (int)F, 2, dSall) : WN_ERR_HIP;
    const bool batched = (st == WN_OK) && dev.down(Sall.data(), dSall, Sall.size());
    if (!batched && st != WN_ERR_UNSUPPORTED) { dev.ok(st); return costs; }
    for (unsigned i = 0; i < Ndiv; i++) {
        for (unsigned j = 0; j < Ndiv; j++) {
            const std::size_t f = (std::size_t)i * Ndiv + j;
            const Col point({grid[2 * f], grid[2 * f + 1]});
            if (batched) {
                for (unsigned e = 0; e < nExample; e++) S[e] = Sall[f * nExample + e];
                // setFilter's host-side effects at every grid point (src/Wavenet.cxx:297-299): the filter
                // log grows by one entry; the device gets the last point's filter after the loop
                m_filter = point;
                m_filterLog.push_back(m_filter);
            } else {
                // ... else point by point (examples beyond the fused kernels' size)
                setFilter(point);
                bool ok = dev.ok(wn_forward(dev.ctx, dX, (int)nExample, (int)R, (int)C, dY, nullptr, nullptr, nullptr)) &&
                          dev.ok(wn_sparse_term(dev.ctx, dY, (int)nExample, (int)n, dS)) && dev.down(S.data(), dS, nExample);
                if (!ok) return costs;
            }
            const double reg = lambda() * RegTerm(point, m_wavelet);
            for (unsigned e = 0; e < nExample; e++) {
                costs[0](i, j) += S[e] + reg;
                costs[1](i, j) += S[e];
                costs[2](i, j) += reg;
            }
        }
    }
    if (batched && F > 0) {
        if (m_momentum.n_elem != m_filter.n_elem) m_momentum.zeros(m_filter.n_elem);
        if (!dev.ok(wn_set_filter(dev.ctx, m_filter.memptr(), (int)m_filter.n_elem))) return costs;
    }
    for (Mat& M : costs) {
        for (std::size_t k = 0; k < M.n_elem; k++) M(k) /= (double)nExample;
    }
    return costs;
}

/// Basis function method(s).
Mat Wavenet::basisFunction1D(const unsigned& nRows, const unsigned& irow) {
    if (!isRadix2(nRows)) {
        WARNING("Cannot produce 1D basis function with length %d. Exiting.", nRows);
        return Mat();
    }
    if (irow >= nRows) {
        WARNING("Requested index (%d) is out of bounds with length %d. Exiting.", irow, nRows);
        return Mat();
    }
    Mat Y(nRows, 1);
    Y(irow, 0) = 1.;
    return inverse_(Y);
}

Mat Wavenet::basisFunction2D(const unsigned& nRows, const unsigned& nCols, const unsigned& irow, const unsigned& icol) {
    if (!isRadix2(nRows) || !isRadix2(nCols)) {
        WARNING("Cannot produce 2D basis function for shape {%d, %d}. Exiting.", nRows, nCols);
        return Mat();
    }
    if (irow >= nRows || icol >= nCols) {
        WARNING("Requested indices (%d, %d) are out of bounds with shape {%d, %d}. Exiting.", irow, icol, nRows, nCols);
        return Mat();
    }
    Mat Y(nRows, nCols);
    Y(irow, icol) = 1.;
    return inverse_(Y);
}

Mat Wavenet::basisFunction(const unsigned& nRows, const unsigned& nCols, const unsigned& irow, const unsigned& icol) {
    if (nCols == 1) { assert(icol < 1); return basisFunction1D(nRows, irow); }
    if (nRows == 1) { assert(irow < 1); return basisFunction1D(nCols, icol); }
    return basisFunction2D(nRows, nCols, irow, icol);
}

/// Transform method(s).
namespace {

// Activations of one line from its pyramid P (P[2^i + k] = low-pass level i), its coefficient
// vector y and the line itself.
Activations1D_t lineActivations(const double* x, const double* P, const double* y, std::size_t n, std::size_t stride) {
    const unsigned m = ilog2((unsigned)n);
    Activations1D_t act(m + 1);
    for (unsigned i = 0; i < m; i++) {
        const std::size_t len = (std::size_t)1 << i;
        act[i][0].zeros(len);
        act[i][1].zeros(len);
        for (std::size_t k = 0; k < len; k++) {
            act[i][0](k) = P[(len + k) * stride];
            act[i][1](k) = y[(len + k) * stride];
        }
    }
    act[m][0].zeros(n);
    act[m][1].zeros(n);
    for (std::size_t k = 0; k < n; k++) act[m][0](k) = x[k * stride];
    return act;
}

// The inverse of lineActivations for the arrays the backward kernels read.
void scatterLine(const Activations1D_t& act, double* x, double* P, std::size_t n, std::size_t stride) {
    const unsigned m = ilog2((unsigned)n);
    P[0] = 0;
    for (unsigned i = 0; i < m; i++) {
        const std::size_t len = (std::size_t)1 << i;
        for (std::size_t k = 0; k < len; k++) P[(len + k) * stride] = act[i][0](k);
    }
    for (std::size_t k = 0; k < n; k++) x[k * stride] = act[m][0](k);
}

}  // namespace

Activations2D_t Wavenet::forward_(const Mat& X) {
    Activations2D_t Activations(2);
    const unsigned R = (unsigned)X.n_rows, C = (unsigned)X.n_cols;
    if (!checkShape_(R, C)) return Activations;
    DeviceSession& dev = device_();
    const std::size_t n = (std::size_t)R * C;
    double* dX = dev.buf(DeviceSession::X, n);
    double* dY = dev.buf(DeviceSession::Y, n);
    double* dPr = dev.buf(DeviceSession::PROW, n);
    double* dY1 = dev.buf(DeviceSession::Y1, n);
    double* dPc = dev.buf(DeviceSession::PCOL, n);
    std::vector<double> Y(n), Pr(n), Y1(n), Pc(n);
    bool ok = dev.up(dX, X.memptr(), n) && dev.ok(wn_forward(dev.ctx, dX, 1, (int)R, (int)C, dY, dPr, dY1, dPc)) &&
              dev.down(Y.data(), dY, n) && dev.down(Pr.data(), dPr, n) && dev.down(Y1.data(), dY1, n) &&
              dev.down(Pc.data(), dPc, n);
    if (!ok) return Activations;
    Activations[0].resize(R);
    Activations[1].resize(C);
    for (unsigned r = 0; r < R; r++) Activations[0][r] = lineActivations(X.memptr() + r, Pr.data() + r, Y1.data() + r, C, R);
    for (unsigned c = 0; c < C; c++)
        Activations[1][c] = lineActivations(Y1.data() + (std::size_t)c * R, Pc.data() + (std::size_t)c * R, Y.data() + (std::size_t)c * R, R, 1);
    return Activations;
}

Activations1D_t Wavenet::forward_(const Col& x) {
    Mat X(x.n_elem, 1, x.memptr());
    Activations2D_t A = forward_(X);
    if (A[1].empty()) return Activations1D_t();
    return A[1][0];
}

Mat Wavenet::inverse_(const Mat& Y) {
    const unsigned R = (unsigned)Y.n_rows, C = (unsigned)Y.n_cols;
    Mat X(R, C);
    if (!checkShape_(R, C)) return X;
    DeviceSession& dev = device_();
    const std::size_t n = (std::size_t)R * C;
    double* dY = dev.buf(DeviceSession::Y, n);
    double* dX = dev.buf(DeviceSession::X, n);
    if (dev.up(dY, Y.memptr(), n) && dev.ok(wn_inverse(dev.ctx, dY, 1, (int)R, (int)C, dX))) dev.down(X.memptr(), dX, n);
    return X;
}

Col Wavenet::inverse_(const Col& y) {
    const Mat X = inverse_(Mat(y.n_elem, 1, y.memptr()));
    return Col(X.n_elem, X.memptr());
}

Col Wavenet::backpropagate_(const Mat& Delta, Activations2D_t Activations) {
    const unsigned R = (unsigned)Activations.at(0).size(), C = (unsigned)Activations.at(1).size();
    const int N = (int)m_filter.n_elem;
    Col gradient(N);
    if (!checkShape_(R, C) || Delta.n_rows != R || Delta.n_cols != C) return gradient;
    const std::size_t n = (std::size_t)R * C;
    std::vector<double> X(n), Pr(n), Y1(n), Pc(n);
    for (unsigned r = 0; r < R; r++) scatterLine(Activations[0][r], X.data() + r, Pr.data() + r, C, R);
    for (unsigned c = 0; c < C; c++) scatterLine(Activations[1][c], Y1.data() + (std::size_t)c * R, Pc.data() + (std::size_t)c * R, R, 1);
    DeviceSession& dev = device_();
    double* dD = dev.buf(DeviceSession::DELTA, n);
    double* dX = dev.buf(DeviceSession::X, n);
    double* dPr = dev.buf(DeviceSession::PROW, n);
    double* dY1 = dev.buf(DeviceSession::Y1, n);
    double* dPc = dev.buf(DeviceSession::PCOL, n);
    double* dG = dev.buf(DeviceSession::GRAD, N);
    bool ok = dev.up(dD, Delta.memptr(), n) && dev.up(dX, X.data(), n) && dev.up(dPr, Pr.data(), n) &&
              dev.up(dY1, Y1.data(), n) && dev.up(dPc, Pc.data(), n) &&
              dev.ok(wn_backprop(dev.ctx, dD, dX, dPr, dY1, dPc, 1, (int)R, (int)C, dG, nullptr));
    if (ok) dev.down(gradient.memptr(), dG, N);
    return gradient;
}

std::vector<Col> Wavenet::backpropagate_(const Col& delta, Activations1D_t activations) {
    const unsigned n = (unsigned)delta.n_elem;
    const int N = (int)m_filter.n_elem;
    std::vector<Col> output(2);
    output[0].zeros(N);
    output[1].zeros(n);
    if (!checkShape_(n, 1) || activations.size() != ilog2(n) + 1) return output;
    std::vector<double> x(n), P(n), zeros(n, 0.0);
    scatterLine(activations, x.data(), P.data(), n, 1);
    DeviceSession& dev = device_();
    double* dD = dev.buf(DeviceSession::DELTA, n);
    double* dX = dev.buf(DeviceSession::X, n);   // column stage input = the signal itself
    double* dPr = dev.buf(DeviceSession::PROW, n);
    double* dPc = dev.buf(DeviceSession::PCOL, n);
    double* dG = dev.buf(DeviceSession::GRAD, N);
    double* dIn = dev.buf(DeviceSession::Y, n);
    bool ok = dev.up(dD, delta.memptr(), n) && dev.up(dX, x.data(), n) && dev.up(dPr, zeros.data(), n) &&
              dev.up(dPc, P.data(), n) &&
              dev.ok(wn_backprop(dev.ctx, dD, dX, dPr, dX, dPc, 1, (int)n, 1, dG, dIn));
    if (ok) {
        dev.down(output[0].memptr(), dG, N);
        dev.down(output[1].memptr(), dIn, n);
    }
    return output;
}

/// Low-level learning method(s).
double Wavenet::effectiveInertia_(unsigned steps) const {
    return m_inertiaTimeScale > 0. ? m_inertia * (1. - std::exp(-float(steps) / m_inertiaTimeScale)) : m_inertia;
}

void Wavenet::flushBatchQueue_() {
    if (!m_batchQueue.size()) return;
    Col gradient(m_filter.n_elem);
    for (std::size_t i = 0; i < m_batchQueue.size(); i++) gradient = gradient + m_batchQueue.at(i);
    for (std::size_t k = 0; k < gradient.n_elem; k++) gradient(k) /= float(m_batchQueue.size());
    this->update_(gradient);
    m_costLog.back() /= float(m_batchQueue.size());
    m_costLog.push_back(0);
    m_batchQueue.clear();
}

void Wavenet::addMomentum_(const Col& gradient) {
    if (m_momentum.n_elem > 0) m_momentum = m_momentum + gradient;
    else m_momentum = gradient;
}

void Wavenet::scaleMomentum_(const double& factor) { m_momentum = m_momentum * factor; }

void Wavenet::update_(const Col& gradient) {
    const unsigned steps = (unsigned)m_costLog.size() - 1;
    scaleMomentum_(effectiveInertia_(steps));
    addMomentum_(gradient * (-m_alpha));
    setFilter(m_filter + m_momentum);
    onReplicas_([&](DeviceSession& d) {
        if (d.ctx) wn_set_momentum(d.ctx, m_momentum.memptr(), (int)m_momentum.n_elem);
        return true;
    });
}

}  // namespace wavenet
