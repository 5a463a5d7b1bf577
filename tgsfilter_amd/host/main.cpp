// main.cpp -- `tgsfilter` for MI355X: the reference's command line, stderr lines and report around
// the batch pipeline  indexer -> batcher -> [queue] -> GPU feeders (tgsf_submit) -> [queue] -> ordered planner -> fill threads.
//
// Replaces main (src/TGSFilter.cpp:2945-3332) and TGSFilterTask (:1755-2162): the reference moves one
// read at a time as three std::string copies through lock-free queues to N worker threads; here reads
// are only INDEXED on the host (the mmap'ed FASTQ text itself is the batch buffer: sequence and quality
// lines are read in place by the kernels), filtered on the GPU through the C ABI, and the kept
// fragments are formatted straight from the input text in input order (= the reference's -t 1 order).
// The run itself is in run.h: main() parses the command line and calls its stages in order.
#include "run.h"

#include <malloc.h>
#include <signal.h>
#include <sys/prctl.h>
#include <sys/wait.h>

using namespace host;

// How the program ends.  A run maps tens of GB (the input text, the output file) and holds GPU contexts: taking all
// that down is ~1.3 s of kernel work per 57 GB of mappings.  By default the program is ONE process and everything is on
// the caller's clock: the mappings of written batches are dropped piece by piece during the run by a background thread
// (see the releaser in main), the rest at exit.  TGSF_DETACH=1 (opt-in) works in a child process instead, as linkers
// that map whole outputs do (mold): when the child has written and closed everything it hands its exit status to the
// waiting parent, which returns to the caller at once, and the child's address space is taken down in the background --
// the caller then shares the machine with that teardown for a moment (and schedulers see a short-lived orphan).
static int g_done_fd = -1;

namespace host {

bool detached() { return g_done_fd >= 0; }

// CPUs' worth of time this process may use: the hardware's threads, or less where a control group caps it (cgroup v2
// cpu.max, as container runtimes set it).  -t keeps the reference's meaning and clamp (:488-499: hardware threads); the
// pipeline's own pools are sized from this -- 32 threads indexing the input at once on a box capped at 16 CPUs are
// throttled together with everything else of the run (measured, 135-GB input: 7.2 s with 32 indexing threads, 6.2 s with 8).
int cpu_budget()
{
    int hw = (int)std::thread::hardware_concurrency();
    if (hw <= 0) hw = 1;
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char q[64] = {0};
        long long per = 0;
        if (fscanf(f, "%63s %lld", q, &per) == 2 && strcmp(q, "max") != 0 && per > 0) {
            const long long quota = atoll(q);
            if (quota > 0) hw = (int)std::min<long long>(hw, std::max<long long>(1, (quota + per - 1) / per));
        }
        fclose(f);
    }
    return hw;
}

[[noreturn]] void leave(int code)
{
    fflush(nullptr);
    if (code != 0) if (void (*f)() = on_die().exchange(nullptr)) f();      // (an output created ahead of the run does not stay behind)
    if (g_done_fd >= 0) {
        prctl(PR_SET_PDEATHSIG, 0);                   // the parent is about to leave in the ordinary way: no signal for that
        if (write(g_done_fd, &code, sizeof code) != (ssize_t)sizeof code) { /* the parent is gone: nothing to tell */ }
        close(0); close(1); close(2);                 // the caller's pipes see end-of-file now, not when the teardown is over
    }
    _exit(code);
}

}  // namespace host

static void work_in_a_child()
{
    const char* d = getenv("TGSF_DETACH");
    if (!d || !*d || *d == '0') return;
    int fds[2];
    if (pipe(fds) != 0) return;
    const pid_t pid = fork();                          // before any thread or GPU state exists
    if (pid < 0) { close(fds[0]); close(fds[1]); return; }
    if (pid > 0) {
        close(fds[1]);
        int code = 0;
        ssize_t n;
        do { n = read(fds[0], &code, sizeof code); } while (n < 0 && errno == EINTR);
        if (n == (ssize_t)sizeof code) _exit(code);    // everything is written and closed
        int st = 0;                                    // the child ended without saying so (a fatal path): its status is ours
        while (waitpid(pid, &st, 0) < 0 && errno == EINTR) {}
        _exit(WIFEXITED(st) ? WEXITSTATUS(st) : 128 + WTERMSIG(st));
    }
    close(fds[0]);
    g_done_fd = fds[1];
    prctl(PR_SET_PDEATHSIG, SIGTERM);                  // no orphan if the parent is killed
}


// Ranks above 0 of a sharded job (shard.h) run the same code as rank 0 but leave the run's INFO lines to it: a filter in
// front of std::cerr drops the lines that begin with "INFO:" (errors and warnings still pass).  Every rank, rank 0 too,
// hands its lines over WHOLE: the ranks share one stderr, and std::cerr on its own writes a line piece by piece -- another
// rank's line would land in its middle.
class InfoFilter : public std::streambuf {
public:
    explicit InfoFilter(std::streambuf* to, bool drop_info = true) : to_(to), drop_info_(drop_info) {}
protected:
    // (any thread may write to std::cerr -- a feeder that ends the run with an error, the reader's messages: one at a time)
    int overflow(int c) override {
        std::lock_guard<std::mutex> g(m_);
        if (c == traits_type::eof()) { flush_line(); return to_->pubsync() == 0 ? 0 : c; }
        line_.push_back((char)c);
        if (c == '\n') flush_line();
        return c;
    }
    std::streamsize xsputn(const char* s, std::streamsize n) override {
        std::lock_guard<std::mutex> g(m_);
        for (std::streamsize i = 0; i < n; i++) { line_.push_back(s[i]); if (s[i] == '\n') flush_line(); }
        return n;
    }
    int sync() override { std::lock_guard<std::mutex> g(m_); flush_line(); return to_->pubsync(); }
private:
    void flush_line() {
        if (drop_info_ && !line_.empty() && line_.back() == '\n' && line_.compare(0, 5, "INFO:") == 0) { line_.clear(); return; }
        if (!line_.empty() && line_.back() == '\n') { to_->sputn(line_.data(), (std::streamsize)line_.size()); line_.clear(); }
    }
    std::streambuf* to_;
    bool drop_info_;
    std::string line_;
    std::mutex m_;
};

int main(int argc, char** argv)
{
    double t_epoch0;
    { struct timespec ts; clock_gettime(CLOCK_REALTIME, &ts); t_epoch0 = (double)ts.tv_sec + ts.tv_nsec * 1e-9; }
    // (a rank above 0 started by another launcher: not even the command line's own INFO lines)
    {
        const int r = shard_rank_on_command_line(argc, argv);
        if (r >= 0) std::cerr.rdbuf(new InfoFilter(std::cerr.rdbuf(), r > 0));
    }
    // the library's waits for the GPU sleep instead of spinning (tgsf_lib.hip, set_wait_mode): this program is the device's only
    // user in its process, and under a CPU quota shared by N ranks a spinning feeder is a CPU the fill threads do not get.
    // TGSF_SYNC=spin in the environment keeps the runtime's default.
    setenv("TGSF_SYNC", "blocking", 0);
    Options o;
    if (parse_args(argc, argv, o)) return 1;
    Run run(o);
    run.t_epoch0 = t_epoch0;
    if (const int rc = run.set_up_ranks()) return rc;               // (--ranks: returns in the N children only)
    if (run.o.ranks >= 1) std::cerr.rdbuf(new InfoFilter(std::cerr.rdbuf(), run.link.rank > 0));
    if (!run.sharded) work_in_a_child();
    // SIGINT / SIGTERM: as on any fatal path, an output file created ahead of its records is removed, one partly written is
    // cut back to the records laid out (unlink / ftruncate: async-signal-safe)
    {
        struct sigaction sa;
        memset(&sa, 0, sizeof sa);
        sa.sa_handler = [](int sig) { if (void (*f)() = on_die().exchange(nullptr)) f(); _exit(128 + sig); };
        sigaction(SIGINT, &sa, nullptr);
        sigaction(SIGTERM, &sa, nullptr);
    }
    // big blocks stay in the heap instead of being mapped and unmapped one by one (see BatchStore)
    mallopt(M_MMAP_THRESHOLD, 1 << 30);
    mallopt(M_TRIM_THRESHOLD, -1);
    run.timing = getenv("TGSF_TIMING") != nullptr;      // stage wall times on stderr (not part of the surface)
    run.t_start = now_s();
    if (const int rc = run.check_file_types()) return rc;
    run.open_input();
    run.open_output_early();
    run.prepass();
    run.make_contexts();
    run.filter_pass();
    run.downsample();
    run.sum_tallies();
    run.report();
    if (run.timing) run.timing_lines();
    // everything is written and closed: skip the teardown of multi-GB mappings and of the HIP runtime
    leave(0);
}
