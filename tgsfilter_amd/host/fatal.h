// fatal.h -- how the program ends on a fatal path, from any thread: no static destructors, no runtime teardown -- flush what was
// said, let the output sink take back what it reserved ahead of the records (or the file it created, if nothing was written to
// it yet), and leave with the reference's exit status (-1).
#pragma once
#include <unistd.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <string>

namespace host {

// Test settings of the command line (TGSF_BATCH_BYTES, TGSF_STRIDE_BYTES, TGSF_STREAM_MIN_BYTES ...: each is explained where it
// is read) force small batches, strides and rare paths in the tests.  They are read only when TGSF_DEBUG_KNOBS=1 is in the
// environment -- the test suite sets it --: a stray variable never changes what a user's run does.  The settings a user may
// give are the few README.md lists (TGSF_TIMING, TGSF_DETACH, TGSF_NUMA, TGSF_WRITER, TGSF_CTX_PER_DEVICE, TGSF_SHARD_EXCHANGE,
// TGSF_LIB, TGSF_ASSETS; the library reads TGSF_SYNC) and are read with getenv where they apply.
inline const char* knob(const char* name)
{
    static const bool on = [] { const char* e = getenv("TGSF_DEBUG_KNOBS"); return e && e[0] == '1'; }();
    return on ? getenv(name) : nullptr;
}

inline std::atomic<void (*)()>& on_die() { static std::atomic<void (*)()> f{nullptr}; return f; }

[[noreturn]] inline void quit(int code)
{
    fflush(nullptr);
    if (void (*f)() = on_die().exchange(nullptr)) f();
    _exit(code);
}

[[noreturn]] inline void die(const std::string& msg)
{
    std::cerr << "Error: " << msg << std::endl;
    quit(255);
}

}  // namespace host
