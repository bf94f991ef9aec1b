// LowpassOperator(filter, size): the taps are used as given (include/Wavenet/LowpassOperator.h:44-45).
#ifndef WAVENET_LOWPASSOPERATOR_H
#define WAVENET_LOWPASSOPERATOR_H

#include "Wavenet/MatrixOperator.h"

namespace wavenet {

class LowpassOperator : public MatrixOperator {
public:
    LowpassOperator() {}
    LowpassOperator(const Col& filter, unsigned size) {
        setSize(size);
        setFilter(filter);
        construct();
    }
    void setFilter(const Col& filter) override;
};

}  // namespace wavenet

#endif
