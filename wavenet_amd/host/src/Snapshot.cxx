#include "Wavenet/Snapshot.h"

#include <sys/stat.h>

#include <cstdio>
#include <fstream>
#include <stdexcept>
#include <string>
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
    const int len = std::snprintf(nullptr, 0, m_pattern.c_str(), m_number);
    if (len <= 0) return m_pattern;
    std::vector<char> text((std::size_t)len + 1);
    std::snprintf(text.data(), text.size(), m_pattern.c_str(), m_number);
    return std::string(text.data(), (std::size_t)len);
}

// Stepping the number of a snapshot whose file name has no place for it changes nothing on disk: say so once per call.
void Snapshot::warnFixedName_(const char* what) const {
    if (!hasFormatSpecifier_()) WARNING("The snapshot's file name '%s' holds no number: %s has no effect on it.", m_pattern.c_str(), what);
}

Snapshot& Snapshot::operator++() { warnFixedName_("++snapshot"); ++m_number; return *this; }
Snapshot& Snapshot::operator--() { warnFixedName_("--snapshot"); --m_number; return *this; }
Snapshot Snapshot::operator++(int) { warnFixedName_("snapshot++"); Snapshot before(*this); ++m_number; return before; }
Snapshot Snapshot::operator--(int) { warnFixedName_("snapshot--"); Snapshot before(*this); --m_number; return before; }

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

// Reads what operator<< above and the reference's writer (src/Snapshot.cxx:54-81) produce.  The file is cut into
// whitespace-separated tokens once; a cursor then walks through: four scalars, two vectors closed by a token that
// holds a '#', the batch size, a section name, and three sections.  In the queue and filter-log sections a run of
// numbers is one vector and whatever follows it closes it; a section ends at an empty run, at the next section's name
// or at the end of the file.  The cost log is every number that is left.  The same files are accepted, and read to
// the same values, as by the reference's reader (src/Snapshot.cxx:83-158).
const Snapshot& operator>>(const Snapshot& snap, Wavenet& wavenet) {
    std::vector<std::string> tokens;
    {
        std::ifstream stream(snap.file());
        std::string token;
        while (stream >> token) tokens.push_back(token);
    }
    std::size_t at = 0;
    const auto more = [&]() { return at < tokens.size(); };
    const auto scalar = [&](double fallback) {
        double v = fallback;
        if (more() && toNumber(tokens[at], &v)) at++;
        return v;
    };
    // numbers up to and including the first token with a '#' (other words in between are passed over)
    const auto closedVector = [&]() {
        std::vector<double> numbers;
        double v;
        while (more()) {
            const std::string& token = tokens[at++];
            if (token.find('#') != std::string::npos) break;
            if (toNumber(token, &v)) numbers.push_back(v);
        }
        return toCol(numbers);
    };
    // the run of numbers at the cursor; `closer` receives the token behind it (empty at the end of the file)
    const auto run = [&](std::string* closer) {
        std::vector<double> numbers;
        double v;
        while (more() && toNumber(tokens[at], &v)) { numbers.push_back(v); at++; }
        *closer = more() ? tokens[at++] : std::string();
        return numbers;
    };

    wavenet.m_lambda = scalar(wavenet.m_lambda);
    wavenet.m_alpha = scalar(wavenet.m_alpha);
    wavenet.m_inertia = scalar(wavenet.m_inertia);
    wavenet.m_inertiaTimeScale = scalar(wavenet.m_inertiaTimeScale);
    wavenet.m_filter = closedVector();
    wavenet.m_momentum = closedVector();
    wavenet.m_batchSize = (unsigned)scalar((double)wavenet.m_batchSize);

    enum Section { QUEUE, FILTERS, COSTS };
    const char* const nextName[] = {"FILTERLOG", "COSTLOG"};
    Section section = QUEUE;
    std::string closer = more() ? tokens[at++] : std::string();           // the queue's own name
    if (closer.find(nextName[QUEUE]) != std::string::npos) section = FILTERS;    // (a file without a queue section)
    wavenet.m_batchQueue.clear();
    wavenet.m_filterLog.clear();
    while (section != COSTS) {
        const std::vector<double> numbers = run(&closer);
        if (!numbers.empty()) (section == QUEUE ? wavenet.m_batchQueue : wavenet.m_filterLog).push_back(toCol(numbers));
        if (numbers.empty() || closer.empty() || closer.find(nextName[section]) != std::string::npos) section = (Section)(section + 1);
    }
    wavenet.m_costLog.clear();
    double v;
    for (; more(); at++) {
        if (toNumber(tokens[at], &v)) wavenet.m_costLog.push_back(v);
    }
    return snap;
}

}  // namespace wavenet
