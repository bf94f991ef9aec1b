// wavenet::Snapshot: a Wavenet object's state as a text file, in the layout of the reference's
// Snapshot (include/Wavenet/Snapshot.h:39-110, src/Snapshot.cxx:6-158).  The file name is a
// pattern with at most one printf integer specifier, completed by a serial number that the
// increment / decrement operators step through.
//
// Layout (src/Snapshot.cxx:54-81): lambda, alpha, inertia, inertia time scale, one per line; the
// filter, one coefficient per line, closed by an empty line and "#"; the momentum likewise; the
// batch size; "BATCHQUEUE" and the queued gradients, each closed by "#"; "FILTERLOG" and the
// logged filters likewise; "COSTLOG" and one cost per line.  The reader takes the file token by
// token exactly as the reference's does (src/Snapshot.cxx:83-158), so files written by the
// reference load here and files written here load there.  The reference prints vectors through
// Armadillo's stream operator, which keeps 4 decimal digits; this writer prints every number
// with 17 significant digits, so a state survives save + load unchanged.
#ifndef WAVENET_SNAPSHOT_H
#define WAVENET_SNAPSHOT_H

#include <string>

#include "Wavenet/Logger.h"

namespace wavenet {

class Wavenet;

bool fileExists(const std::string& path);   // regular file or directory entry of that name
bool dirExists(const std::string& path);

class Snapshot {
public:
    Snapshot() {}
    Snapshot(const std::string& pattern) : m_pattern(pattern) {}
    Snapshot(const std::string& pattern, const int& number) : m_pattern(pattern), m_number(number) {}

    void setPattern(const std::string& pattern) { m_pattern = pattern; }
    void setNumber(const int& number) { m_number = number; }
    int number() const { return m_number; }
    std::string pattern() const { return m_pattern; }

    bool exists() const { return fileExists(file()); }
    // the pattern itself, or the pattern completed with the number when it holds a '%'
    std::string file() const;

    Snapshot& operator++();
    Snapshot& operator--();
    Snapshot operator++(int);
    Snapshot operator--(int);

protected:
    bool hasFormatSpecifier_() const { return m_pattern.find("%") != std::string::npos; }
    void warnFixedName_(const char* what) const;

private:
    std::string m_pattern = "";
    int m_number = 0;
};

// write / read the Wavenet object to / from snap.file()
Snapshot& operator<<(Snapshot& snap, const Wavenet& wavenet);
const Snapshot& operator>>(const Snapshot& snap, Wavenet& wavenet);

}  // namespace wavenet

#endif
