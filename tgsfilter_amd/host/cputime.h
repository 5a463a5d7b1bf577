// cputime.h -- CPU seconds of the run by stage (TGSF_TIMING's "CPU:" line).  Every thread of the command line charges the CPU
// time it used (CLOCK_THREAD_CPUTIME_ID: user + system, waiting excluded) to the stage it works for when it ends -- or when
// the stage's scope ends, for the main thread.  What the process used beyond the sum (getrusage) belongs to threads that
// are not the program's own: the HIP runtime's helpers, RCCL's proxies.  This is what bounds a job of N rank processes on a
// box whose control group allows C CPUs' worth of time: all ranks together cannot filter more than C / (CPU-s per Gbase).
#pragma once
#include <sys/resource.h>
#include <time.h>

#include <atomic>
#include <cstdint>

namespace host {

enum CpuStage {
    CPU_MAIN = 0,        // options, waiting, statistics, report (the main thread outside the scopes below)
    CPU_PREPASS,         // the pre-pass on the main thread (sample of reads, base content, adapter search: its host side)
    CPU_INDEX_LINES,     // line ends of the input text (LineScanner's threads: one memchr pass over the text)
    CPU_INDEX_RECORDS,   // record assembly from the line ends (RecordIndex's thread)
    CPU_BATCHER,         // records dealt into batches (the reader thread)
    CPU_FEEDER,          // tgsf_create + tgsf_submit: the staged copy of the text into pinned memory, launches, waiting for results
    CPU_PLANNER,         // results -> layout of the output (the writer thread; the single-stream writer's writev too)
    CPU_FILL,            // records copied into the mapped output file (fill threads)
    CPU_POPULATE,        // page-table entries for instantiated pages of the output (MADV_POPULATE_WRITE)
    CPU_FALLOCATE,       // instantiation of the output file's pages (fallocate: kernel time of the reserver thread)
    CPU_RELEASER,        // mappings of written batches dropped (input text, output file)
    CPU_LOADER,          // dlopen of the library, device bring-up (helper threads)
    CPU_DOWNSAMPLE,      // the second pass of a downsampling run (its threads)
    CPU_N
};
inline const char* cpu_stage_name(int s)
{
    static const char* const n[CPU_N] = {"main", "pre-pass", "line ends", "record assembly", "batching", "feeders (staging copy + submit)", "planner",
                                          "fill threads", "mapping output pages", "fallocate", "releasing mappings", "library + device", "downsampling pass"};
    return n[s];
}
inline std::atomic<uint64_t>* cpu_ns() { static std::atomic<uint64_t> t[CPU_N]; return t; }
inline uint64_t thread_cpu_ns()
{
    struct timespec ts;
    clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts);
    return (uint64_t)ts.tv_sec * 1000000000ull + (uint64_t)ts.tv_nsec;
}
// charges the calling thread's CPU time between construction and destruction to `stage`
struct CpuScope {
    explicit CpuScope(int stage) : stage_(stage), t0_(thread_cpu_ns()) {}
    ~CpuScope() { cpu_ns()[stage_].fetch_add(thread_cpu_ns() - t0_, std::memory_order_relaxed); }
    CpuScope(const CpuScope&) = delete;
    CpuScope& operator=(const CpuScope&) = delete;
private:
    int stage_;
    uint64_t t0_;
};
// user + system seconds of the whole process so far (every thread, the runtime's included)
inline double process_cpu_s()
{
    struct rusage ru;
    getrusage(RUSAGE_SELF, &ru);
    return (double)ru.ru_utime.tv_sec + ru.ru_utime.tv_usec * 1e-6 + (double)ru.ru_stime.tv_sec + ru.ru_stime.tv_usec * 1e-6;
}

}  // namespace host
