// wavenet::Wavenet with the public surface of the reference class (include/Wavenet/Wavenet.h:76-860)
// on top of the C ABI in include/wavenet_hip.h: the transforms, the sparsity gradient and the
// backpropagation run as HIP kernels; the object keeps the reference's host-side state (filter,
// momentum, batch queue, filter log, cost log) and semantics.
//
// Not carried over: everything under USE_ROOT.
#ifndef WAVENET_WAVENET_H
#define WAVENET_WAVENET_H

#include <string>
#include <vector>

#include "Wavenet/CostFunctions.h"
#include "Wavenet/HighpassOperator.h"
#include "Wavenet/Logger.h"
#include "Wavenet/LowpassOperator.h"
#include "Wavenet/Matrix.h"
#include "Wavenet/Snapshot.h"
#include "Wavenet/Utilities.h"

struct wn_context;

namespace wavenet {

class GeneratorBase;
class DeviceSession;  // device buffers and the wn_context of one Wavenet object

class Wavenet {
public:
    Wavenet();
    Wavenet(const double& lambda);
    Wavenet(const double& lambda, const double& alpha);
    Wavenet(const Wavenet& other);  // copies the learning parameters and the filter (Wavenet.h:91-97)
    Wavenet& operator=(const Wavenet&) = delete;
    ~Wavenet();

    /// Get method(s).
    double lambda() const { return m_lambda; }
    double alpha() const { return m_alpha; }
    double inertia() const { return m_inertia; }
    double inertiaTimeScale() const { return m_inertiaTimeScale; }
    Col filter() const { return m_filter; }
    Col momentum() const { return m_momentum; }
    int batchSize() const { return m_batchSize; }
    bool wavelet() const { return m_wavelet; }
    std::vector<Col>& filterLog() { return m_filterLog; }
    void clearFilterLog() { m_filterLog.clear(); }
    std::vector<double>& costLog() { return m_costLog; }
    void clearCostLog() { m_costLog.assign(1, 0.); }
    // gradients queued for the pending batch (the reference keeps them private and carries them through
    // its snapshots; Coach restores them between initialisations)
    std::vector<Col>& batchQueue() { return m_batchQueue; }
    double firstCost() const;
    double lastCost() const;

    /// Set method(s).
    bool setLambda(const double& lambda);
    bool setAlpha(const double& alpha);
    bool setInertia(const double& inertia);
    bool setInertiaTimeScale(const double& inertiaTimeScale);
    bool setFilter(const Col& filter);
    bool setMomentum(const Col& momentum);
    bool setBatchSize(const unsigned& batchSize) { m_batchSize = batchSize; return true; }
    bool doWavelet(const bool& wavelet) { m_wavelet = wavelet; return true; }
    // GPU to run on (default 0); takes effect before the first device call.
    void setDevice(int device) { m_device = device; }
    // Data-parallel training over several GPUs of the node: every whole batch of trainBatch / trainBatchFrom
    // is split into contiguous shares, one per entry of `devices` (share r holds examples [r B / G, (r + 1) B / G)
    // in the sense of the sizes differing by at most one), each GPU runs forward, sparsity gradient and
    // backpropagation on its share, the N + 2 sums are added over the GPUs and every replica applies the same
    // update (src/Wavenet.cxx:717-765 is the one place where examples of a batch meet).  The sums travel by RCCL
    // (wn_comm_init_all / wn_allreduce_sums_all); with useRccl false they are added on the host in device order,
    // which also allows one GPU to appear several times (several shares on one card, for tests).  Takes effect
    // before the first device call; the first entry also becomes the device of everything else.
    void setDevices(const std::vector<int>& devices, bool useRccl = true);
    unsigned numReplicas() const { return m_devices.empty() ? 1u : (unsigned)m_devices.size(); }

    void print() const;

    /// Storage method(s) (src/Wavenet.cxx:136-178).
    // Writes the object to snap.file(), creating its directory; absolute paths are refused.
    void save(Snapshot snap) const;
    // Reads the object back; the device side, if it exists, receives the loaded filter and momentum.
    void load(Snapshot snap);

    /// High-level learning method(s).
    // One example, exactly the reference's train (src/Wavenet.cxx:184-242).
    bool train(Mat X);
    // B examples (column-major, contiguous, host memory) in arrival order, same result as B calls of
    // train(): whole batches run on the device back to back; a trailing partial batch is left in
    // the batch queue.  lambdas (optional, B values) is the regularisation constant per example
    // for Coach's annealing; otherwise lambda() is used.
    bool trainBatch(const double* X, unsigned B, unsigned nRows, unsigned nCols, const double* lambdas = nullptr);
    // The same with the B examples drawn from a generator that can write them on the device
    // (GeneratorBase::nextOnDevice); false without touching anything if it cannot.
    // With deferLogs the filter log, cost log, filter and momentum of this object are NOT brought back from the device
    // when the call returns (whole batches only): they follow at the next call without the flag, at sync(), or once
    // 8192 updates are outstanding.  Nothing but the readback is put off -- the device applies every update as it
    // comes -- so a loop that takes no decision from the logs (Coach without its adaptive rules) runs without a
    // host round trip per batch.  A rejected (non-finite) batch is reported when the logs arrive.
    bool trainBatchFrom(GeneratorBase& generator, unsigned B, unsigned nRows, unsigned nCols, const double* lambdas = nullptr,
                        bool deferLogs = false);
    // Brings outstanding logs and state over (no effect when there are none); false when a batch was rejected.
    bool sync();
    unsigned pendingUpdates() const { return m_pending; }
    void clear();

    /// High-level cost method(s).
    double cost(const Col& y);
    double cost(const Mat& Y);
    std::vector<Mat> costMap(const std::vector<Mat>& X, const double& range, const unsigned& Ndiv);

    /// Basis function method(s).
    Mat basisFunction1D(const unsigned& nRows, const unsigned& irow);
    Mat basisFunction2D(const unsigned& nRows, const unsigned& nCols, const unsigned& irow, const unsigned& icol);
    Mat basisFunction(const unsigned& nRows, const unsigned& nCols, const unsigned& irow, const unsigned& icol);

    /// The transform methods by their plain names.
    Activations1D_t forward(const Col& x) { return forward_(x); }
    Activations2D_t forward(const Mat& X) { return forward_(X); }
    Col inverse(const Col& y) { return inverse_(y); }
    Mat inverse(const Mat& Y) { return inverse_(Y); }
    std::vector<Col> backpropagate(const Col& delta, Activations1D_t activations) { return backpropagate_(delta, activations); }
    Col backpropagate(const Mat& Delta, Activations2D_t Activations) { return backpropagate_(Delta, Activations); }
    void update(const Col& gradient) { update_(gradient); }

protected:
    /// 1D wavenet transform method(s).
    Activations1D_t forward_(const Col& x);
    Col inverse_(const Col& y);
    // {gradient on the filter coefficients, errors on the input}
    std::vector<Col> backpropagate_(const Col& delta, Activations1D_t activations);

    /// 2D wavenet transform method(s).
    Activations2D_t forward_(const Mat& X);
    Mat inverse_(const Mat& Y);
    Col backpropagate_(const Mat& Delta, Activations2D_t Activations);

    /// Low-level learning method(s).
    void flushBatchQueue_();
    void addMomentum_(const Col& gradient);
    void scaleMomentum_(const double& factor);
    void update_(const Col& gradient);
    double effectiveInertia_(unsigned steps) const;

private:
    friend Snapshot& operator<<(Snapshot& snap, const Wavenet& wavenet);
    friend const Snapshot& operator>>(const Snapshot& snap, Wavenet& wavenet);

    DeviceSession& device_();
    // Runs f(session) on the session of every replica, the first one included; false as soon as one fails.
    template <class F> bool onReplicas_(F f);
    // Whole batches over all replicas: place(r, dst, first, count) puts examples [first, first + count) of the call
    // on replica r's device at dst; it is called in the order of the examples.
    template <class Place> bool trainSharded_(Place place, unsigned nb, unsigned done, unsigned nRows, unsigned nCols, const double* lambdas);
    bool exchangeSums_();
    bool checkShape_(unsigned nRows, unsigned nCols) const;
    // forward + sparsity gradient + backpropagation of `count` examples already on the device,
    // per-example results into the batch queue and the cost log (the slow, exact path of train()).
    bool trainOnDevice_(const double* dX, unsigned B, unsigned nRows, unsigned nCols, const double* lambdas, bool deferLogs = false);
    bool fetchUpdates_();    // the outstanding updates' logs, the filter and the momentum in one readback (wn_get_updates)
    bool deviceSteps_();     // makes m_devSteps valid
    bool queueExamples_(const double* X_dev, unsigned count, unsigned nRows, unsigned nCols, const double* lambdas);
    bool diverged_() const;  // prints the reference's "diverging solution" message, returns false

    double m_lambda = 10.0;
    double m_alpha = 0.001;
    double m_inertia = 0.;
    double m_inertiaTimeScale = 0.;
    Col m_filter;
    Col m_momentum;
    unsigned m_batchSize = 1;
    std::vector<Col> m_batchQueue;
    std::vector<Col> m_filterLog;
    std::vector<double> m_costLog = {0};
    bool m_wavelet = true;
    int m_device = 0;
    DeviceSession* m_session = nullptr;
    std::vector<int> m_devices;               // data-parallel: one entry per replica (empty: single device)
    std::vector<DeviceSession*> m_replicas;   // sessions of the replicas 1 .. G-1 (replica 0 is m_session)
    bool m_useRccl = true;
    // updates queued on the first device whose logs have not been fetched, the regularisation constant in force at the
    // end of each of them (NaN: unchanged), and the device's update counter as of the last fetch
    unsigned m_pending = 0;
    std::vector<double> m_pendingLambda;
    int m_devSteps = -1;
};

template <class T>
inline T frobeniusProduct(const Mat& A, const Mat& B) {
    T acc = 0;
    for (std::size_t i = 0; i < A.n_elem; i++) acc += A(i) * B(i);
    return acc;
}

}  // namespace wavenet

#endif
