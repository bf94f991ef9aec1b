// Dense 2^size x 2^(size+1) filter matrices, kept for callers that want the reference's view
// (include/Wavenet/MatrixOperator.h:34-103).  The GPU path never builds them: see wn_set_filter.
#ifndef WAVENET_MATRIXOPERATOR_H
#define WAVENET_MATRIXOPERATOR_H

#include "Wavenet/Logger.h"
#include "Wavenet/Matrix.h"

namespace wavenet {

class MatrixOperator : public Mat {
public:
    MatrixOperator() {}
    virtual ~MatrixOperator() {}

    void setSize(unsigned size) {
        m_size = size;
        m_hasSize = true;
        m_complete = m_filter.n_elem > 0;
    }
    virtual void setFilter(const Col& filter) = 0;

    unsigned size() const { return m_size; }
    bool complete() const { return m_complete; }

    // Fills the matrix: entry (r, (N/2 + 2r - j) mod 2^(size+1)) += filter(j).
    void construct();

protected:
    void setComplete(bool complete = true) { m_complete = complete; }

    unsigned m_size = 0;
    bool m_hasSize = false;
    Col m_filter;
    bool m_complete = false;
};

}  // namespace wavenet

#endif
