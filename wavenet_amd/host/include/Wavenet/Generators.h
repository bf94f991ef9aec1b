// Deterministic counterparts of the reference's in-memory generators (include/Wavenet/Generators.h:
// NeedleGenerator :31-78, UniformGenerator :81-108, GaussianGenerator :111-187).  The reference
// reseeds Armadillo's RNG from entropy on every open(); these are counter-based (seed, example,
// pixel) so that a run can be repeated and compared.  reset() restarts the sequence.
#ifndef WAVENET_GENERATORS_H
#define WAVENET_GENERATORS_H

#include <cstdint>
#include <vector>

#include "Wavenet/GeneratorBase.h"

namespace wavenet {

class SeededGenerator : public GeneratorBase {
public:
    void setSeed(std::uint64_t seed) { m_seed = seed; }
    std::uint64_t seed() const { return m_seed; }
    bool good() override { return m_initialised; }
    bool open() override { m_event = 0; return true; }

protected:
    std::uint64_t m_seed = 1;
    std::uint64_t m_event = 0;
};

// Independent U[0,1) pixels.
class UniformGenerator : public SeededGenerator {
public:
    const Mat& next() override;
    bool nextOnDevice(wn_context* ctx, double* X_dev, unsigned count) override;
    bool hasDeviceForm() const override { return true; }
};

// U[0,1) pixels with everything below 0.995 set to zero; never an all-zero example.
class NeedleGenerator : public SeededGenerator {
public:
    const Mat& next() override;
    bool nextOnDevice(wn_context* ctx, double* X_dev, unsigned count) override;
    bool hasDeviceForm() const override { return true; }
};

// One to three unit-height Gaussian bumps at random positions and widths (GaussianGenerator, Generators.h:122-166).
// exp() comes from a 4097-entry table with linear interpolation, the same table and arithmetic on host and device,
// so that the two forms agree to the bit.
class GaussianGenerator : public SeededGenerator {
public:
    const Mat& next() override;
    bool nextOnDevice(wn_context* ctx, double* X_dev, unsigned count) override;
    bool hasDeviceForm() const override { return true; }
};

// Calorimeter-style eta-phi images standing for the reference's HepMCGenerator (Generators.h:368-402: a TH2F over
// eta x phi filled with pT / 1000 of the visible particles, float bin contents): a Poisson number of soft deposits
// with exponentially distributed energies in uniformly drawn bins, plus `jets` clusters of harder deposits spread
// over +-3 bins; deposits in one bin add, bin contents are rounded to float.  The same sequence as
// wn_generate_calorimeter and wavenet_amd/generators.py:calorimeter.
class CalorimeterGenerator : public SeededGenerator {
public:
    void setMeanDeposits(double mean) { m_meanDeposits = mean; m_tablesFor = -1.; }   // default: 200 per 4096 bins
    void setJets(int jets, int depositsPerJet) { m_jets = jets; m_jetDeposits = depositsPerJet; }
    const Mat& next() override;
    bool nextOnDevice(wn_context* ctx, double* X_dev, unsigned count) override;
    bool hasDeviceForm() const override { return true; }

private:
    void tables_();
    double m_meanDeposits = -1.;
    int m_jets = 2, m_jetDeposits = 30;
    double m_tablesFor = -1.;
    std::vector<int> m_poisson;
    std::vector<double> m_energy;
};

// A fixed set of `events` examples of another generator, produced once on the device and served from there epoch
// after epoch: what a file-based generator of the reference is to Coach (good() turns false after the last example,
// reset() starts over) without the file.  Wavenet::trainBatchFrom trains such examples where they sit.
class ResidentGenerator : public GeneratorBase {
public:
    // (the source's shape must be set already; it is restarted and drawn from once, at the first request for examples)
    ResidentGenerator(GeneratorBase& source, unsigned events);
    ~ResidentGenerator();
    const Mat& next() override;
    bool good() override { return m_initialised && m_event < m_events; }
    bool open() override { m_event = 0; return true; }
    bool nextOnDevice(wn_context* ctx, double* X_dev, unsigned count) override;
    bool hasDeviceForm() const override { return m_source.hasDeviceForm(); }
    const double* deviceSpan(wn_context* ctx, unsigned count) override;
    long long remaining() const override { return (long long)m_events - (long long)m_event; }

private:
    bool fill_(wn_context* ctx);
    GeneratorBase& m_source;
    unsigned m_events, m_event = 0;
    wn_context* m_ctx = nullptr;     // the generator's own context, on the GPU of the first caller
    int m_device = -1;
    double* m_dev = nullptr;
};

}  // namespace wavenet

#endif
