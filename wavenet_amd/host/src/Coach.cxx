#include "Wavenet/Coach.h"

#include <cmath>
#include <cstring>
#include <vector>

namespace wavenet {

void Coach::setBasedir(const std::string& basedir) {
    m_basedir = basedir;
    if (!m_basedir.empty() && m_basedir.back() != '/') m_basedir.append("/");
}

void Coach::setNumEvents(const int& numEvents) {
    if (numEvents < 0 and numEvents != -1) {
        WARNING("Input number of events (%d) not supported.", numEvents);
        return;
    }
    m_numEvents = numEvents;
}

void Coach::setNumCoeffs(const unsigned& numCoeffs) {
    if (!isRadix2(numCoeffs)) {
        WARNING("Input number of coefficients (%d) is not radix 2.", numCoeffs);
        return;
    }
    m_numCoeffs = numCoeffs;
}

void Coach::setTargetPrecision(const double& targetPrecision) {
    if (targetPrecision != -1 && targetPrecision <= 0) {
        WARNING("Requested target precision (%f) is no good. Exiting.", targetPrecision);
        return;
    }
    if (!(useAdaptiveLearningRate() || useAdaptiveBatchSize())) {
        WARNING("Target precision only has an effect together with 'adaptive learning rate' or 'adaptive batch size'.");
    }
    m_targetPrecision = targetPrecision;
}

// The loop of the reference's Coach::run (src/Coach.cxx:130-278).  Events are gathered into
// chunks that end where the reference's batch queue would flush, so that every decision the
// reference takes per event (they all depend on flushes only) is taken at the same point.
bool Coach::run() {
    if (!m_wavenet) { ERROR("Wavenet object not set. Exiting."); return false; }
    if (!m_generator) { ERROR("Input generator not set. Exiting."); return false; }
    if (!m_generator->initialised()) {
        ERROR("Generator was not properly initialised. Did you remember to specify a valid shape? Exiting.");
        return false;
    }
    if (!m_name.size()) { ERROR("Coach name not set. Exiting."); return false; }
    if (m_numEvents < 0 && m_generator->remaining() < 0 && !((useAdaptiveLearningRate() || useAdaptiveBatchSize()) && targetPrecision() > 0.)) {
        // an endless generator, like the reference's Needle/Uniform/Gaussian ones (:94-109)
        WARNING("The number of events is %d, the generator has no natural epochs and no target precision is set. Exiting.", m_numEvents);
        return false;
    }

    if (!m_devices.empty()) m_wavenet->setDevices(m_devices, m_useRccl);
    const unsigned useLastN = 10;                       // :120
    const double lambdaBare = m_wavenet->lambda();      // :123
    const double alphaBare = m_wavenet->alpha();        // the reference restores these from its base snapshot (:138)
    const unsigned batchBare = (unsigned)m_wavenet->batchSize();
    const std::vector<Col> queueBare = m_wavenet->batchQueue();
    const std::vector<unsigned> shape = m_generator->shape();
    const unsigned nRows = shape[0], nCols = shape[1];
    const std::size_t n = (std::size_t)nRows * nCols;
    const bool adaptive = useAdaptiveLearningRate() || useAdaptiveBatchSize();

    std::vector<double> staging, lambdas;
    Snapshot snap(outdir() + "snapshots/" + m_name + ".%06u.snap", 0);                   // :127

    for (unsigned init = 0; init < m_numInits; init++) {
        if (m_printLevel > 0) INFO("Initialisation %d/%d", init + 1, m_numInits);
        m_wavenet->setLambda(lambdaBare);
        m_wavenet->setAlpha(alphaBare);
        m_wavenet->setBatchSize(batchBare);
        m_wavenet->batchQueue() = queueBare;
        m_wavenet->clear();                                                             // :139
        if (m_initialFilter.n_elem) m_wavenet->setFilter(m_initialFilter);
        else m_wavenet->setFilter(PointOnNSphere(m_numCoeffs, 0., m_seed + init));      // :144

        bool done = false;
        unsigned tail = 0;
        unsigned currentCostLogSize = 0, previousCostLogSize = 0;
        // The reference compares the cost log's size after every EVENT (:196-204).  After the first event
        // of an initialisation the size (1, no flush yet when the batch holds more than one example) differs
        // from the initial 0, which already counts as a change and bumps `tail`.  Events come in whole
        // batches here, so that first comparison is made up front.
        if (adaptive && m_wavenet->batchSize() > 1 && m_numEvents != 0) {
            currentCostLogSize = (unsigned)m_wavenet->costLog().size();
            tail = 1;
        }

        for (unsigned epoch = 0; epoch < m_numEpochs && !done; epoch++) {
            m_generator->reset();                                                       // :164
            if (m_printLevel > 1) INFO("  Epoch %d/%d", epoch + 1, m_numEpochs);
            int event = 0;
            bool generatorGood = true;
            while (!done && generatorGood && (event < m_numEvents || m_numEvents < 0)) {
                // chunk: up to the next flush when decisions depend on it, else many batches at once
                const unsigned bs = (unsigned)std::max(1, m_wavenet->batchSize());
                // (without the adaptive rules: as many whole batches as fit 2^27 samples, 1 GiB, and a generator call)
                const unsigned many = (unsigned)std::min<std::size_t>(65535u, std::max<std::size_t>(4096u, ((std::size_t)1 << 27) / n));
                unsigned chunk = adaptive ? bs : bs * std::max(1u, many / bs);
                if (m_numEvents >= 0) chunk = std::min<unsigned>(chunk, (unsigned)(m_numEvents - event));
                if (m_generator->remaining() >= 0) chunk = (unsigned)std::min<long long>(chunk, std::max<long long>(1, m_generator->remaining()));
                lambdas.resize(chunk);
                // a generator with a device form writes the chunk where the kernels read it
                const bool onDevice = m_generateOnDevice && m_generator->hasDeviceForm() && chunk <= 65535u;
                if (!onDevice) staging.resize((std::size_t)chunk * n);
                unsigned got = 0;
                if (onDevice) {
                    for (; got < chunk; got++) {
                        if (useSimulatedAnnealing()) {                                  // :176-180
                            const double f = (event + (int)got + (int)epoch * numEvents()) / float(numEvents() * (int)numEpochs());
                            lambdas[got] = lambdaBare * f / sq(2 - f);
                        } else {
                            lambdas[got] = m_wavenet->lambda();
                        }
                    }
                    // nothing here looks at the logs between batches unless the adaptive rules or the per-event print are on:
                    // they are then fetched once per epoch instead of once per chunk
                    const bool status = m_wavenet->trainBatchFrom(*m_generator, got, nRows, nCols,
                                                                  useSimulatedAnnealing() ? lambdas.data() : nullptr,
                                                                  !adaptive && m_printLevel <= 2);
                    if (!m_generator->good()) generatorGood = false;                    // :264
                    event += (int)got;
                    if (!status) { done = true; break; }                                // :186-189
                }
                for (; !onDevice && got < chunk; got++) {
                    if (useSimulatedAnnealing()) {                                      // :176-180
                        const double f = (event + (int)got + (int)epoch * numEvents()) / float(numEvents() * (int)numEpochs());
                        lambdas[got] = lambdaBare * f / sq(2 - f);
                    } else {
                        lambdas[got] = m_wavenet->lambda();
                    }
                    const Mat& X = m_generator->next();                                 // :183
                    std::memcpy(staging.data() + (std::size_t)got * n, X.memptr(), n * sizeof(double));
                    if (!m_generator->good()) { generatorGood = false; got++; break; }  // :264
                }
                if (!onDevice) {
                    const bool status = m_wavenet->trainBatch(staging.data(), got, nRows, nCols,
                                                              useSimulatedAnnealing() ? lambdas.data() : nullptr);
                    event += (int)got;
                    if (!status) { done = true; break; }                                // :186-189
                }

                if (adaptive) {                                                         // :192-250
                    previousCostLogSize = currentCostLogSize;
                    currentCostLogSize = (unsigned)m_wavenet->costLog().size();
                    const bool changed = (currentCostLogSize != previousCostLogSize);
                    if (changed && ++tail > useLastN) {
                        const std::vector<Col>& log = m_wavenet->filterLog();
                        const unsigned filterLogSize = (unsigned)log.size();
                        double meanStepSize = 0;
                        for (unsigned i = 0; i < useLastN; i++)
                            meanStepSize += norm(log.at(filterLogSize - useLastN + i) - log.at(filterLogSize - useLastN + i - 1));
                        meanStepSize /= float(useLastN);
                        const double totalStepSize = norm(log.at(filterLogSize - 1) - log.at(filterLogSize - 1 - useLastN));
                        if (targetPrecision() != -1 && meanStepSize < targetPrecision() && !useSimulatedAnnealing()) {
                            INFO("[Adaptive learning] Mean step size %f below the target precision %f. Done.", meanStepSize, targetPrecision());
                            done = true;
                        } else if (totalStepSize < meanStepSize) {
                            if (useAdaptiveBatchSize()) m_wavenet->setBatchSize(2 * m_wavenet->batchSize());
                            if (useAdaptiveLearningRate()) m_wavenet->setAlpha((1. / 2.) * m_wavenet->alpha());
                            tail = 0;
                        }
                    }
                }
                if (m_printLevel > 2) INFO("    Event %d (cost: %7.3f)", event, m_wavenet->lastCost());
            }
            if (!m_wavenet->sync()) done = true;                                        // outstanding logs; a rejected batch ends the initialisation (:186-189)
        }
        m_wavenet->costLog().pop_back();                                                // :274
        if (m_saveSnapshots) m_wavenet->save(snap++);                                   // :277
    }
    return true;
}

}  // namespace wavenet
