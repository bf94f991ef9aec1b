// Deterministic counterparts of the reference's in-memory generators (include/Wavenet/Generators.h:
// NeedleGenerator :31-78, UniformGenerator :81-108, GaussianGenerator :111-187).  The reference
// reseeds Armadillo's RNG from entropy on every open(); these are counter-based (seed, example,
// pixel) so that a run can be repeated and compared.  reset() restarts the sequence.
#ifndef WAVENET_GENERATORS_H
#define WAVENET_GENERATORS_H

#include <cstdint>

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

// One to three unit-height Gaussian bumps at random positions and widths.
class GaussianGenerator : public SeededGenerator {
public:
    const Mat& next() override;
};

}  // namespace wavenet

#endif
