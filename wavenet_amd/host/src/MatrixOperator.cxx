#include "Wavenet/HighpassOperator.h"
#include "Wavenet/LowpassOperator.h"
#include "Wavenet/MatrixOperator.h"
#include "wavenet_hip.h"

namespace wavenet {

void MatrixOperator::construct() {
    if (!complete()) {
        WARNING("Trying to construct incomplete operator.");
        return;
    }
    // m_filter already holds the taps this operator uses, so the low-pass entry point applies.
    set_size((std::size_t)1 << m_size, (std::size_t)2 << m_size);
    const int status = wn_lowpass_operator(m_filter.memptr(), (int)m_filter.n_elem, m_size, memptr());
    if (status != WN_OK) WARNING("Operator construction failed: %s", wn_status_string(status));
}

void LowpassOperator::setFilter(const Col& filter) {
    if (filter.n_elem % 2 != 0) {
        WARNING("Trying to set odd number of filter coefficients (%d).", (int)filter.n_elem);
        return;
    }
    m_filter = filter;
    if (m_hasSize) setComplete();
}

void HighpassOperator::setFilter(const Col& filter) {
    if (filter.n_elem % 2 != 0) {
        WARNING("Trying to set odd number of filter coefficients (%d).", (int)filter.n_elem);
        return;
    }
    const std::size_t N = filter.n_elem;
    m_filter.zeros(N);
    for (std::size_t i = 0; i < N; i++) m_filter(i) = ((i & 1) ? -1.0 : 1.0) * filter(N - i - 1);
    if (m_hasSize) setComplete();
}

}  // namespace wavenet
