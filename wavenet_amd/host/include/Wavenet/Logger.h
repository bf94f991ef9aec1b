// printf-style messages in the spirit of the reference's Logger mix-in (include/Wavenet/Logger.h):
// level and origin as a prefix, no colours, written to stderr so that tools can keep stdout clean.
#ifndef WAVENET_LOGGER_H
#define WAVENET_LOGGER_H

#include <cstdio>

namespace wavenet {
inline int& logLevel() { static int level = 1; return level; }  // 0 silent, 1 warnings/errors, 2 info, 3 debug
}

#define WN_LOG_(minlevel, tag, ...)                                        \
    do {                                                                   \
        if (::wavenet::logLevel() >= (minlevel)) {                         \
            std::fprintf(stderr, "<%s> %s: ", tag, __func__);              \
            std::fprintf(stderr, __VA_ARGS__);                             \
            std::fprintf(stderr, "\n");                                    \
        }                                                                  \
    } while (0)

#define DEBUG(...) WN_LOG_(3, "DEBUG", __VA_ARGS__)
#define INFO(...) WN_LOG_(2, "INFO", __VA_ARGS__)
#define WARNING(...) WN_LOG_(1, "WARNING", __VA_ARGS__)
#define ERROR(...) WN_LOG_(1, "ERROR", __VA_ARGS__)

#endif
