// Source of training examples, same contract as the reference (include/Wavenet/GeneratorBase.h:32-103):
// next() returns a reference to an internal matrix of the configured radix-2 shape that stays valid
// until the following next(); a {1,n} or {n} shape becomes {n,1} (src/GeneratorBase.cxx:15-61).
#ifndef WAVENET_GENERATORBASE_H
#define WAVENET_GENERATORBASE_H

#include <vector>

#include "Wavenet/Logger.h"
#include "Wavenet/Matrix.h"

namespace wavenet {

class GeneratorBase {
public:
    virtual ~GeneratorBase() {}

    virtual const Mat& next() = 0;
    virtual bool good() = 0;
    virtual bool open() { return true; }
    virtual bool close() { return true; }
    bool reset() { return close() && open(); }

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
