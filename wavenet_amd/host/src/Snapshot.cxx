#include "Wavenet/Snapshot.h"

#include <sys/stat.h>

#include <cstdio>
#include <fstream>
#include <stdexcept>
#include <vector>

#include "Wavenet/Wavenet.h"

namespace wavenet {

bool fileExists(const std::string& path) {
    struct stat st;
    return ::stat(path.c_str(), &st) == 0;
}

bool dirExists(const std::string& path) {
    struct stat st;
    return ::stat(path.c_str(), &st) == 0 && S_ISDIR(st.st_mode);
}

std::string Snapshot::file() const {
    if (!hasFormatSpecifier_()) return m_pattern;
    char buff[100];                                             // the reference's limit (src/Snapshot.cxx:13)
    std::snprintf(buff, sizeof(buff), m_pattern.c_str(), m_number);
    return std::string(buff);
}

Snapshot& Snapshot::operator++() {
    if (!hasFormatSpecifier_()) WARNING("Calling pre-fix increment operator for snapshot with no format specifiers in file name pattern");
    ++m_number;
    return *this;
}

Snapshot& Snapshot::operator--() {
    if (!hasFormatSpecifier_()) WARNING("Calling pre-fix decrement operator for snapshot with no format specifiers in file name pattern");
    --m_number;
    return *this;
}

Snapshot Snapshot::operator++(int) {
    if (!hasFormatSpecifier_()) WARNING("Calling post-fix increment operator for snapshot with no format specifiers in file name pattern");
    Snapshot before(*this);
    m_number++;
    return before;
}

Snapshot Snapshot::operator--(int) {
    if (!hasFormatSpecifier_()) WARNING("Calling post-fix decrement operator for snapshot with no format specifiers in file name pattern");
    Snapshot before(*this);
    m_number--;
    return before;
}

namespace {

void putNumber(std::FILE* f, double v) { std::fprintf(f, "%.17g\n", v); }

// one coefficient per line, then an empty line and the "#" that closes a vector
void putVector(std::FILE* f, const Col& v) {
    for (std::size_t i = 0; i < v.n_elem; i++) std::fprintf(f, "  %.17g\n", v(i));
    std::fprintf(f, "\n#\n");
}

// std::stod as the reference calls it: a leading number is enough, anything else throws
bool toNumber(const std::string& token, double* value) {
    try {
        *value = std::stod(token);
        return true;
    } catch (const std::invalid_argument&) {
        return false;
    } catch (const std::out_of_range&) {
        *value = std::strtod(token.c_str(), nullptr);   // the reference would terminate; keep what strtod gives
        return true;
    }
}

Col toCol(const std::vector<double>& v) { return v.empty() ? Col() : Col(v.size(), v.data()); }

}  // namespace

Snapshot& operator<<(Snapshot& snap, const Wavenet& wavenet) {
    std::FILE* f = std::fopen(snap.file().c_str(), "w");
    if (!f) {
        ERROR("Cannot open '%s' for writing.", snap.file().c_str());
        return snap;
    }
    putNumber(f, wavenet.m_lambda);
    putNumber(f, wavenet.m_alpha);
    putNumber(f, wavenet.m_inertia);
    putNumber(f, wavenet.m_inertiaTimeScale);
    putVector(f, wavenet.m_filter);
    putVector(f, wavenet.m_momentum);
    std::fprintf(f, "%u\n", wavenet.m_batchSize);
    std::fprintf(f, "BATCHQUEUE\n");
    for (const Col& q : wavenet.m_batchQueue) putVector(f, q);
    std::fprintf(f, "FILTERLOG\n");
    for (const Col& l : wavenet.m_filterLog) putVector(f, l);
    std::fprintf(f, "COSTLOG\n");
    for (const double c : wavenet.m_costLog) putNumber(f, c);
    std::fclose(f);
    return snap;
}

const Snapshot& operator>>(const Snapshot& snap, Wavenet& wavenet) {
    std::string tmp;
    std::ifstream stream(snap.file());

    stream >> wavenet.m_lambda;
    stream >> wavenet.m_alpha;
    stream >> wavenet.m_inertia;
    stream >> wavenet.m_inertiaTimeScale;

    // filter and momentum: numbers up to the first token that holds a '#'      (src/Snapshot.cxx:97-113)
    double value;
    std::vector<double> numbers;
    while (stream >> tmp && tmp.find("#") == std::string::npos) {
        if (toNumber(tmp, &value)) numbers.push_back(value);
    }
    wavenet.m_filter = toCol(numbers);
    numbers.clear();
    while (stream >> tmp && tmp.find("#") == std::string::npos) {
        if (toNumber(tmp, &value)) numbers.push_back(value);
    }
    wavenet.m_momentum = toCol(numbers);

    stream >> wavenet.m_batchSize;

    // batch queue: vectors of numbers, each ended by a token that is no number   (src/Snapshot.cxx:117-129)
    stream >> tmp;
    wavenet.m_batchQueue.clear();
    while (tmp.find("FILTERLOG") == std::string::npos) {
        numbers.clear();
        while (stream >> tmp) {
            if (!toNumber(tmp, &value)) break;
            numbers.push_back(value);
        }
        if (numbers.empty()) break;
        wavenet.m_batchQueue.push_back(toCol(numbers));
    }

    // filter log                                                                  (src/Snapshot.cxx:131-142)
    wavenet.m_filterLog.clear();
    while (tmp.find("COSTLOG") == std::string::npos) {
        numbers.clear();
        while (stream >> tmp) {
            if (!toNumber(tmp, &value)) break;
            numbers.push_back(value);
        }
        if (numbers.empty()) break;
        wavenet.m_filterLog.push_back(toCol(numbers));
    }

    // cost log: every remaining number                                            (src/Snapshot.cxx:144-152)
    wavenet.m_costLog.clear();
    while (!stream.fail()) {
        while (stream >> tmp) {
            if (!toNumber(tmp, &value)) break;
            wavenet.m_costLog.push_back(value);
        }
    }
    return snap;
}

}  // namespace wavenet
