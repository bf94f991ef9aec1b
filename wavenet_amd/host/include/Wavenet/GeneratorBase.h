// Source of training examples, same contract as the reference (include/Wavenet/GeneratorBase.h:32-103):
// next() returns a reference to an internal matrix of the configured radix-2 shape that stays valid
// until the following next(); a {1,n} or {n} shape becomes {n,1} (src/GeneratorBase.cxx:15-61).
#ifndef WAVENET_GENERATORBASE_H
#define WAVENET_GENERATORBASE_H

#include <vector>

#include "Wavenet/Logger.h"
#include "Wavenet/Matrix.h"

struct wn_context;

namespace wavenet {

class GeneratorBase {
public:
    virtual ~GeneratorBase() {}

    virtual const Mat& next() = 0;
    virtual bool good() = 0;
    virtual bool open() { return true; }
    virtual bool close() { return true; }
    bool reset() { return close() && open(); }

    // Optional fast path with no counterpart in the reference: the next `count` examples written
    // straight into device memory (column-major, contiguous: the layout of wavenet_hip.h), advancing
    // the sequence exactly as `count` calls of next() would.  Returns false (and generates nothing)
    // when the generator has no device form; Coach then gathers from next() as usual.
    virtual bool nextOnDevice(wn_context* /*ctx*/, double* /*X_dev*/, unsigned /*count*/) { return false; }
    virtual bool hasDeviceForm() const { return false; }
    // A generator whose examples already sit in device memory hands out the next `count` of them where they are
    // (valid until the generator is destroyed) and advances; null: no such storage, use nextOnDevice.
    virtual const double* deviceSpan(wn_context* /*ctx*/, unsigned /*count*/) { return nullptr; }
    // Examples left before good() turns false, for generators with a natural end (the reference's file generators,
    // include/Wavenet/Generators.h:194-241, 368-402); -1: endless.
    virtual long long remaining() const { return -1; }

    bool setShape(const std::vector<unsigned>& shape);
    std::vector<unsigned> shape() const { return m_shape; }
    unsigned dim() const { return m_shape.size() > 1 && m_shape[1] == 1 ? 1 : 2; }
    bool initialised() const { return m_initialised; }

protected:
    bool check_() const;

    bool m_initialised = false;
    std::vector<unsigned> m_shape;
    Mat m_data;
};

}  // namespace wavenet

#endif
