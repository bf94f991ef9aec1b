// HighpassOperator(filter, size): taps b_i = (-1)^i a_{N-1-i} derived from the low-pass filter
// (include/Wavenet/HighpassOperator.h:48-49, src/HighpassOperator.cxx:14-18).
#ifndef WAVENET_HIGHPASSOPERATOR_H
#define WAVENET_HIGHPASSOPERATOR_H

#include "Wavenet/MatrixOperator.h"

namespace wavenet {

class HighpassOperator : public MatrixOperator {
public:
    HighpassOperator() {}
    HighpassOperator(const Col& filter, unsigned size) {
        setSize(size);
        setFilter(filter);
        construct();
    }
    void setFilter(const Col& filter) override;
};

}  // namespace wavenet

#endif
