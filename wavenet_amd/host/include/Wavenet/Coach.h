// wavenet::Coach: the training driver with the reference's configuration surface
// (include/Wavenet/Coach.h:59-149) and the loop of Coach::run (src/Coach.cxx:71-296), feeding the
// Wavenet object whole batches instead of single events.  The snapshot of each initialisation's
// final state (outdir()/snapshots/<name>.%06u.snap, src/Coach.cxx:127, :277) is written on request
// (setSaveSnapshots); the run README is not written.
#ifndef WAVENET_COACH_H
#define WAVENET_COACH_H

#include <cstdint>
#include <string>

#include "Wavenet/GeneratorBase.h"
#include "Wavenet/Logger.h"
#include "Wavenet/Wavenet.h"

namespace wavenet {

class Coach {
public:
    Coach(const std::string& name) : m_name(name) {}
    ~Coach() {}  // wavenet and generator are not owned (Coach.h:65-69)

    /// Set method(s).
    void setName(const std::string& name) { m_name = name; }
    void setBasedir(const std::string& basedir);
    void setWavenet(Wavenet* wavenet) { m_wavenet = wavenet; }
    void setGenerator(GeneratorBase* generator) { m_generator = generator; }
    void setNumEvents(const int& numEvents);
    void setNumEpochs(const unsigned& numEpochs) { m_numEpochs = numEpochs; }
    void setNumInits(const unsigned& numInits) { m_numInits = numInits; }
    void setNumCoeffs(const unsigned& numCoeffs);
    void setUseAdaptiveLearningRate(const unsigned& flag = true) { m_useAdaptiveLearningRate = flag; }
    void setUseAdaptiveBatchSize(const unsigned& flag = true) { m_useAdaptiveBatchSize = flag; }
    void setUseSimulatedAnnealing(const unsigned& flag = true) { m_useSimulatedAnnealing = flag; }
    void setTargetPrecision(const double& targetPrecision);
    void setPrintLevel(const unsigned& printLevel) { m_printLevel = printLevel; }
    // The reference always starts from PointOnNSphere(numCoeffs) seeded from entropy
    // (src/Coach.cxx:144).  Here the start is reproducible: either this explicit filter (used for
    // every initialisation) or PointOnNSphere(numCoeffs, 0, seed + init).
    void setInitialFilter(const Col& filter) { m_initialFilter = filter; }
    void setSeed(std::uint64_t seed) { m_seed = seed; }
    // examples are generated on the device when the generator can do that (default); off: always next()
    void setGenerateOnDevice(bool flag = true) { m_generateOnDevice = flag; }
    bool generateOnDevice() const { return m_generateOnDevice; }
    // Data-parallel batch loop: the GPUs the wavenet's batches are sharded over (Wavenet::setDevices); set before run().
    void setDevices(const std::vector<int>& devices, bool useRccl = true) { m_devices = devices; m_useRccl = useRccl; }
    // write outdir()/snapshots/<name>.<init, 6 digits>.snap after every initialisation (default: off)
    void setSaveSnapshots(bool flag = true) { m_saveSnapshots = flag; }
    bool saveSnapshots() const { return m_saveSnapshots; }

    /// Get method(s).
    std::string name() const { return m_name; }
    std::string basedir() const { return m_basedir; }
    std::string outdir() const { return m_basedir + m_name + "/"; }
    Wavenet* wavenet() const { return m_wavenet; }
    GeneratorBase* generator() const { return m_generator; }
    int numEvents() const { return m_numEvents; }
    unsigned numEpochs() const { return m_numEpochs; }
    unsigned numInits() const { return m_numInits; }
    unsigned numCoeffs() const { return m_numCoeffs; }
    bool useAdaptiveLearningRate() const { return m_useAdaptiveLearningRate; }
    bool useAdaptiveBatchSize() const { return m_useAdaptiveBatchSize; }
    bool useSimulatedAnnealing() const { return m_useSimulatedAnnealing; }
    double targetPrecision() const { return m_targetPrecision; }
    unsigned printLevel() const { return m_printLevel; }

    /// High-level training method(s).
    bool run();

private:
    std::string m_name = "";
    std::string m_basedir = "./output/";
    Wavenet* m_wavenet = nullptr;
    GeneratorBase* m_generator = nullptr;
    int m_numEvents = 1000;
    unsigned m_numEpochs = 1;
    unsigned m_numInits = 1;
    unsigned m_numCoeffs = 2;
    bool m_useAdaptiveLearningRate = false;
    bool m_useAdaptiveBatchSize = false;
    bool m_useSimulatedAnnealing = false;
    double m_targetPrecision = -1;
    unsigned m_printLevel = 0;
    Col m_initialFilter;
    std::uint64_t m_seed = 1;
    bool m_generateOnDevice = true;
    bool m_saveSnapshots = false;
    std::vector<int> m_devices;
    bool m_useRccl = true;
};

}  // namespace wavenet

#endif
