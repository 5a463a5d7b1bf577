// fatal.h -- how the program ends on a fatal path, from any thread: no static destructors, no runtime teardown -- flush what was
// said, let the output sink take back what it reserved ahead of the records (or the file it created, if nothing was written to
// it yet), and leave with the reference's exit status (-1).
#pragma once
#include <unistd.h>

#include <atomic>
#include <cstdio>
#include <iostream>
#include <string>

namespace host {

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
