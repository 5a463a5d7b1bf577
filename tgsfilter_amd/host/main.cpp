// main.cpp -- `tgsfilter` for MI355X: the reference's command line, stderr lines and report around
// the batch pipeline  indexer -> batcher -> [queue] -> GPU feeders (tgsf_submit) -> [queue] -> ordered planner -> fill threads.
//
// Replaces main (src/TGSFilter.cpp:2945-3332) and TGSFilterTask (:1755-2162): the reference moves one
// read at a time as three std::string copies through lock-free queues to N worker threads; here reads
// are only INDEXED on the host (the mmap'ed FASTQ text itself is the batch buffer: sequence and quality
// lines are read in place by the kernels), filtered on the GPU through the C ABI, and the kept
// fragments are formatted straight from the input text in input order (= the reference's -t 1 order).
#include "api.h"
#include "pipeline.h"
#include "prepass.h"
#include "report.h"
#include "shard.h"
#include <memory>
#include <sys/stat.h>

#include <sched.h>
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

// CPUs' worth of time this process may use: the hardware's threads, or less where a control group caps it (cgroup v2
// cpu.max, as container runtimes set it).  -t keeps the reference's meaning and clamp (:488-499: hardware threads); the
// pipeline's own pools are sized from this -- 32 threads indexing the input at once on a box capped at 16 CPUs are
// throttled together with everything else of the run (measured, 135-GB input: 7.2 s with 32 indexing threads, 6.2 s with 8).
static int cpu_budget()
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

[[noreturn]] static void leave(int code)
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
    for (int i = 1; i + 1 < argc; i++) {
        std::string f = argv[i];
        f.erase(std::remove(f.begin(), f.end(), '-'), f.end());
        if (f != "shard") continue;
        const char* r = !strcmp(argv[i + 1], "env") ? getenv("RANK") : argv[i + 1];
        if (r) std::cerr.rdbuf(new InfoFilter(std::cerr.rdbuf(), atoi(r) > 0));
    }
    Options o;
    if (parse_args(argc, argv, o)) return 1;
    // ---- one process per GPU (shard.h) ----
    RankLink link;
    bool shard_may_use_rccl = false;
    if (o.ranks >= 1 && o.shard_world > 0) { std::cerr << "Error: --ranks starts the ranks itself; --shard is for a rank started by another launcher" << std::endl; return 1; }
    if (o.ranks >= 1 || o.shard_world >= 1) {
        const int world = o.ranks >= 1 ? o.ranks : o.shard_world;
        const char* why = nullptr;
        if (o.out_file.empty() && !o.only_qc && !o.only_adapters) why = "every rank writes a part file of its own: -o is needed";
        else if (file_type(o.in_file) == 2 || (o.in_file.size() > 3 && o.in_file.compare(o.in_file.size() - 3, 3, ".gz") == 0))
            why = "the ranks take byte ranges of a plain FASTQ / FASTA text";
        if (why) { std::cerr << "Error: --ranks / --shard: " << why << std::endl; return 1; }
        if (world > 1024) { std::cerr << "Error: --ranks " << world << std::endl; return 1; }
        // a GPU per rank? (--ranks knows before it forks; ranks started by another launcher find out when they meet)
        {
            std::vector<int> dv;
            for (int r = 0; r < world && o.ranks >= 1; r++) dv.push_back(o.devices.empty() ? r : o.devices[(size_t)r % o.devices.size()]);
            std::sort(dv.begin(), dv.end());
            shard_may_use_rccl = o.ranks < 1 || std::adjacent_find(dv.begin(), dv.end()) == dv.end();
            const char* ex = getenv("TGSF_SHARD_EXCHANGE");
            if (ex && !strcmp(ex, "socket")) shard_may_use_rccl = false;
        }
        if (o.ranks >= 1) {
            fork_ranks(world, link);                                       // returns in the N children only
            std::cerr.rdbuf(new InfoFilter(std::cerr.rdbuf(), link.rank > 0));
            // rank r on device r, or on the r-th entry of --devices (cyclically: several ranks may share a GPU)
            o.device = o.devices.empty() ? link.rank : o.devices[(size_t)link.rank % o.devices.size()];
        } else {
            if (o.rendezvous.empty()) { std::cerr << "Error: --shard needs --rendezvous <path>: where the ranks of the job meet (a unix socket)" << std::endl; return 1; }
            setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0", 0);                    // (as fork_ranks does for its children: shard.h)
            link.rendezvous(o.shard_rank, world, o.rendezvous);
        }
        o.devices.assign(1, o.device);
        // -t is the job's: every rank takes its share -- or, on a node whose CPUs outnumber it (the reference clamps -t to 32:
        // 4 threads a rank with 8 GPUs), its share of the CPUs this job may use, up to -t
        o.n_thread = std::max(2, std::min(o.n_thread, std::max(o.n_thread / world, cpu_budget() / world)));
    }
    const bool sharded = o.ranks >= 1 || o.shard_world >= 1;             // (also with one rank: the same program path, one part file)
    if (!sharded) work_in_a_child();
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
    const bool timing = getenv("TGSF_TIMING") != nullptr;      // stage wall times on stderr (not part of the surface)
    const double t_start = now_s();
    double t_drain = 0, t_prepass = 0, t_pipe = 0, t_parse = 0, t_gpu = 0, t_write = 0, t_widle = 0, t_first = 0;

    // file types and report name, :2993-3033
    o.in_type = file_type(o.in_file);
    const std::string prefix = file_prefix(o.in_file);
    std::string html = prefix + ".html";
    if (!o.out_file.empty() && !o.only_qc) {
        html = file_prefix(o.out_file) + ".html";
        o.out_type = file_type(o.out_file);
        if (file_extension(o.out_file) == "gz") o.out_gz = true;
    } else {
        o.out_type = o.fasta_out ? 0 : (o.in_type == 2 ? 1 : o.in_type);
    }
    if (o.in_type == 3 || o.out_type == 3) {
        std::cerr << "Error: The file name suffix should be '.[fastq|fq|fasta|fa][.gz] or .[sam|bam]'" << std::endl;
        if (o.in_type == 3) std::cerr << "Error: Please check your input file name: " << o.in_file << std::endl;
        else std::cerr << "Error: Please check your output file name: " << o.out_file << std::endl;
        return 1;
    }
    if (o.in_type == 0 && o.out_type == 1) { std::cerr << "Error: Fasta format input file can't output fastq format file" << std::endl; return 1; }
    // rows of SURVEY 8(f) that are not built yet fail loudly instead of silently doing something else
    const bool fasta_in = o.in_type == 0;                              // records without qualities: count-only tallies, no Q gate

    // loading the HIP library, device bring-up and kernel loading run beside the input open, the indexing and
    // the pre-pass (api.h)
    if (o.devices.empty()) o.devices.push_back(o.device);
    lib_start(o.devices);
    if (sharded && shard_may_use_rccl) rccl_start();                   // (librccl is large: loaded beside the pre-pass, only where it can be used)

    // Compressed / BAM / SAM input beyond a size is STREAMED: decoded piece by piece in bounded memory, once for the
    // pre-pass (which stops after its sample of reads) and once for the filter pass, as the reference reads it twice
    // (:949-1040, :1845-1917).  Smaller ones are decoded whole (below), plain files are mapped.  A downsampling run keeps
    // its kept fragments addressed in the input text, so it takes the whole-file way.
    const bool coded = o.in_type == 2 || (o.in_file.size() > 3 && o.in_file.compare(o.in_file.size() - 3, 3, ".gz") == 0);
    uint64_t stream_min = 256ull << 20;
    if (const char* e = knob("TGSF_STREAM_MIN_BYTES")) stream_min = strtoull(e, nullptr, 10);      // test knob
    InputBytes in;
    bool streaming = false;
    if (coded && !o.downsample) {
        if (!in.open_raw(o.in_file)) leave(1);
        streaming = in.size() >= stream_min;
    }
    if (!streaming && !in.open(o.in_file, o.in_type == 2)) leave(1);   // SAM/BAM: decoded to FASTQ text (read_bam, :1872-1917)
    const size_t chunk_bytes = [] { const char* e = knob("TGSF_CHUNK_BYTES"); return e ? (size_t)strtoull(e, nullptr, 10) : (size_t)(64u << 20); }();
    auto open_stream = [&]() {
        std::string err;
        std::unique_ptr<TextSource> src = open_text(in.data(), in.size(), o.in_type == 2, err);
        if (!src) { std::cerr << "Error: " << err << " (" << o.in_file << ")" << std::endl; fflush(nullptr); _exit(255); }
        return std::unique_ptr<ChunkReader>(new ChunkReader(std::move(src), !fasta_in, chunk_bytes, 20));   // <= 20 x 64 MB of text alive
    };
    // mapped / decoded input: the records are indexed once, in the background, for the pre-pass and for the filter pass
    const int budget = std::max(1, cpu_budget() / link.world);         // (a sharded job: every rank takes its share)
    // indexing runs ahead of everything else and is memory-bound from a few threads on: half the CPU budget at most
    const int scan_threads = std::max(1, std::min({o.n_thread, 32, std::max(2, budget / 2)}));
    // This rank's part of the text: all of it, or -- one process per GPU, shard.h -- the rank-th of `world` byte ranges,
    // cut where find_record_start proposes.  The proposal is checked against the record reader's own view from both
    // sides (the reader thread below): this rank's first record must begin exactly at the cut, and its last record must
    // end exactly at the next rank's cut.  By induction from rank 0, which starts at byte 0, every rank then reads its
    // records exactly as the reference's one sequential reader does (FastxReader, src/TGSFilter.cpp:521-782: its only
    // state between two records is the position in the text); a text that cannot be cut that way -- lines that make the
    // reader skip, a malformed record -- ends the run with a message instead of being read differently.
    size_t text_off = 0, text_size = in.size();
    if (sharded) {
        if (streaming || (in.size() > 0 && !in.mapped())) die("--ranks / --shard: the input is not a plain text file");
        const size_t lo = find_record_start(in.data(), in.size(), (size_t)((unsigned __int128)in.size() * (unsigned)link.rank / (unsigned)link.world), !fasta_in);
        const size_t hi = link.rank + 1 == link.world ? in.size()
                        : find_record_start(in.data(), in.size(), (size_t)((unsigned __int128)in.size() * (unsigned)(link.rank + 1) / (unsigned)link.world), !fasta_in);
        text_off = lo; text_size = hi > lo ? hi - lo : 0;
    }
    const char* const text = in.data() + text_off;
    std::unique_ptr<RecordIndex> records_p;
    if (!streaming) records_p.reset(new RecordIndex(text, text_size, !fasta_in, scan_threads));

    // The output file's pages are the critical path of a run that writes a tmpfs file: their instantiation
    // starts NOW, beside the pre-pass and the device bring-up -- if the file does not exist yet (an existing one is not
    // touched before the run is certain to write it: a run that ends in its pre-pass leaves it as it was, as the reference
    // does; a file created here is removed again on such a path).
    // (a rank of a sharded job writes its own part: the parts, concatenated in rank order, are the single process's file)
    const std::string out_path = sharded && !o.out_file.empty() ? o.out_file + ".part" + std::to_string(link.rank) : o.out_file;
    MappedSink sink;
    std::atomic<bool> early_stop{false};
    std::thread early;
    {
        const char* w = getenv("TGSF_WRITER");                         // "writev": always the single-stream writer
        uint64_t early_min = 256ull << 20;
        if (const char* e = knob("TGSF_EARLY_OPEN_MIN")) early_min = strtoull(e, nullptr, 10);           // tests: small inputs too
        const bool may_map_early = !o.only_qc && !o.out_gz && !o.downsample && (o.filter || o.only_qc) && !o.out_file.empty() &&
                                   !(w && !strcmp(w, "writev")) && !o.only_adapters && !streaming && in.mapped() &&
                                   (uint64_t)text_size >= early_min;
        if (may_map_early && sink.open(out_path, 4 * (uint64_t)text_size + (1ull << 30), true))
            early = std::thread([&] {
                CpuScope cpu(CPU_FALLOCATE);
                const uint64_t limit = (uint64_t)text_size / 4;        // what a run keeps is not known yet; a surplus is cut off at the end
                while (!early_stop.load() && sink.reserved() < limit)
                    if (!sink.reserve_to(std::min<uint64_t>(limit, sink.reserved() + (256u << 20)), false)) break;   // (a nearly full file system: not this thread's call)
            });
    }
    auto end_early = [&] { if (early.joinable()) { early_stop = true; early.join(); } };

    // ---- pre-pass, :3058-3126 ----
    PrepassResult pp;
    std::unique_ptr<CpuScope> cpu_prepass(new CpuScope(CPU_PREPASS));
    if (sharded && link.rank > 0) {
        // (rank 0 looks at the first reads of the WHOLE input, as the reference does, and broadcasts what it found: below)
    } else if (streaming) {
        std::unique_ptr<ChunkReader> cr = open_stream();
        std::shared_ptr<Chunk> ch;
        size_t at = 0;
        bool over = false;
        pp = run_prepass(o, [&](Rec& r) {
            while (!over && (!ch || at >= ch->recs.size())) {
                if (ch && !ch->message.empty()) std::cerr << ch->message << std::endl;
                if (ch && ch->last) { over = true; break; }
                ch = cr->next(o.in_file);
                at = 0;
                if (!ch) over = true;
            }
            if (over) return false;
            r = ch->recs[at++];
            return true;
        });
    } else if (sharded && link.world > 1) {
        // the sample may reach beyond this rank's part: an index of its own over the whole text, kept a little ahead of
        // the pre-pass and dropped when that has seen enough
        // (every other rank waits for what this pass finds: it may use more than this rank's share of the CPUs for a moment)
        RecordIndex whole(in.data(), in.size(), !fasta_in, std::max(scan_threads, std::min(8, cpu_budget() / 2)), 1u << 14);
        RecordIndex::Cursor cur(whole);
        pp = run_prepass(o, [&](Rec& r) { return cur.next(r); });
    } else {
        RecordIndex::Cursor cur(*records_p);
        pp = run_prepass(o, [&](Rec& r) { return cur.next(r); });
    }
    if (sharded) {                                                     // the pre-pass's constants, from rank 0 to every rank
        BlobOut b;
        if (link.rank == 0) { b.pod(pp.qtype); b.pod(pp.trim5p); b.pod(pp.trim3p); b.pod(pp.depth5p); b.pod(pp.depth3p); b.pod(o.min_q); b.str(pp.adapter5p); b.str(pp.adapter3p); }
        link.bcast(b.s);
        BlobIn r(b.s);
        r.pod(pp.qtype); r.pod(pp.trim5p); r.pod(pp.trim3p); r.pod(pp.depth5p); r.pod(pp.depth3p); r.pod(o.min_q); r.str(pp.adapter5p); r.str(pp.adapter3p);
    }
    cpu_prepass.reset();
    t_prepass = now_s() - t_start;
    std::vector<std::string> adapters;
    if (o.filter) {
        if (o.head_trim < 0) o.head_trim = pp.trim5p;
        if (o.tail_trim < 0) o.tail_trim = pp.trim3p;
        std::cerr << "INFO: trim 5' end length: " << o.head_trim << std::endl;
        std::cerr << "INFO: trim 3' end length: " << o.tail_trim << std::endl;
        std::cerr << "INFO: min output reads length: " << o.min_len << std::endl;
        if (!fasta_in) std::cerr << "INFO: min Phred average quality score: " << o.min_q << std::endl;     // :3074-3076
        auto add = [&](const std::string& a) { if (std::find(adapters.begin(), adapters.end(), a) == adapters.end()) adapters.push_back(a); };
        if (!o.adapter_file.empty()) {                                 // Get_adapters, :2923-2942
            InputBytes af;
            if (af.open(o.adapter_file)) {
                const int t = file_type(o.adapter_file);
                FastxReader rd(af.data(), af.size(), t == 1);
                Record r;
                while (rd.next(r)) { add(std::string(r.seq)); add(rev_comp(std::string(r.seq))); }
            }
            int num = 0;
            for (const std::string& a : adapters) std::cerr << "INFO: input adapter " << ++num << " :" << a << std::endl;
        } else {
            std::string a5 = pp.adapter5p, a3 = pp.adapter3p;
            float d5 = pp.depth5p, d3 = pp.depth3p;
            if (d5 > 5 * d3) { a3.clear(); d3 = 0; } else if (d3 > 5 * d5) { a5.clear(); d5 = 0; }   // :3086-3092
            std::cerr << "INFO: 5' adapter: " << a5 << std::endl;
            std::cerr << "INFO: 3' adapter: " << a3 << std::endl;
            std::cerr << "INFO: mean depth of 5' adapter: " << d5 << std::endl;
            std::cerr << "INFO: mean depth of 3' adapter: " << d3 << std::endl;
            if (o.only_adapters) leave(0);
            if (!a5.empty()) { add(a5); add(rev_comp(a5)); }
            if (!a3.empty()) { add(a3); add(rev_comp(a3)); }
            if (a5.empty() && a3.empty()) {                            // :3115-3125
                if (o.read_type == "hifi" || o.read_type == "clr") {
                    add(kAdapterLib[0]); add(kAdapterLib[1]);
                    std::cerr << "INFO: set PacBio blunt adapter to trim: " << kAdapterLib[0] << std::endl;
                } else if (o.read_type == "ont") {
                    add(kAdapterLib[8]); add(kAdapterLib[9]);
                    std::cerr << "INFO: set NanoPore rapid adapter to trim: " << kAdapterLib[8] << std::endl;
                }
            }
        }
    }

    // ---- contexts ----
    // batches are slices of the input text: sized in text bytes (about 2 bytes per base + headers)
    uint64_t batch_text = streaming ? std::min<uint64_t>(256ull << 20, chunk_bytes)
                                    : std::min<uint64_t>(256ull << 20, std::max<uint64_t>(text_size / 8 + 4096, 1 << 16));
    if (const char* e = knob("TGSF_BATCH_BYTES")) { const long long v = atoll(e); if (v > 0) batch_text = (uint64_t)v; }   // tuning / test knob
    // reads per batch: the library keeps traceback scratch for every (read, adapter, end) of a batch -- columns x words of
    // the longest alignment each; with the library adapters that is ~12 KB per read, with 256-bp adapters and loose
    // match lengths ~350 KB: keep it under 4 GB per context
    uint32_t batch_reads = 1u << 16;
    {
        // (the library's rule, tgsf_lib.hip: every alignment of a batch gets the columns of the longest one, and the
        // words of the widest column class any adapter needs -- 1 / 2 / 4 words up to 64 / 128 / 256 bp, 20 beyond)
        uint64_t per_read = 0, cols = 0, nw = 1;
        for (const std::string& a : adapters) {
            const int Q = (int)a.size();
            const int kmax = std::max(0, std::min(Q - 1, std::max(Q - o.end_match_len + 1, Q - o.mid_match_len + 1)));
            cols = std::max<uint64_t>(cols, (uint64_t)(Q + kmax + 2));
            nw = std::max<uint64_t>(nw, Q > 256 ? (uint64_t)(Q + 63) / 64 : Q > 128 ? 4 : Q > 64 ? 2 : 1);
        }
        per_read = cols * 2 * nw * 8;
        if (nw > 4) per_read = std::min<uint64_t>(per_read, (1ull << 20) + 16 * nw) + 32 * nw + 512;   // (beyond 1 MiB an alignment is cut by Hirschberg's scheme, as in edlib)
        per_read *= 3 * std::max<size_t>(adapters.size(), 1);          // two end windows + one middle alignment per adapter
        if (per_read) batch_reads = (uint32_t)std::min<uint64_t>(batch_reads, std::max<uint64_t>(256, (4ull << 30) / per_read));
    }
    const bool fastq_out = o.out_type == 1;
    const bool run_filter_pass = o.filter || o.only_qc;                // :3061; with -F the input goes straight to downsampling
    Output out;
    {
        const char* w = getenv("TGSF_WRITER");                         // "writev": always the single-stream writer
        const bool may_map = !o.only_qc && !o.out_gz && !o.downsample && run_filter_pass && !o.out_file.empty() &&
                             !(w && !strcmp(w, "writev"));
        // address space for the output mapping: what the input could turn into (a streamed input's text size is unknown)
        // (address space only: pages exist where records are laid out.  A record's header is repeated in front of each of
        // its fragments, so an output can outgrow its input -- by a factor only headers of kilobytes reach.)
        if (may_map && !sink.is_open()) sink.open(out_path, streaming ? std::max<uint64_t>(64ull << 30, 64ull * in.size())
                                                                      : 4 * (uint64_t)text_size + (1ull << 30));
        Options oo = o;
        oo.out_file = out_path;
        if (!o.only_qc && !sink.is_open() && !out.open(oo)) leave(1);
    }
    const Api& L = lib();                                              // joins the loader thread
    double t_load = 0, t_dev = 0;
    lib_times(t_load, t_dev);
    const double t_libwait = now_s() - t_start - t_prepass;
    // How the tallies of a sharded job will be summed at the end: on the devices, one RCCL all-reduce, when every rank has
    // a GPU of its own -- the communicator is set up NOW, on a helper thread beside the filtering (RCCL's first
    // initialisation takes longer than a small run) -- or over the ranks' sockets when ranks share a GPU (RCCL refuses two
    // ranks on one device), where the collective library is missing, or with TGSF_SHARD_EXCHANGE=socket.
    const RcclApi* R = nullptr;
    void* rccl_comm = nullptr;
    int rccl_rc = TGSF_OK;
    std::string rccl_err;
    std::thread rccl_up;
    struct RcclUp { std::atomic<int> done{0}; int rc = TGSF_OK; std::string err; void* comm = nullptr; };
    const std::shared_ptr<RcclUp> rccl_state = std::make_shared<RcclUp>();
    bool use_rccl = false;
    if (sharded) {
        const char* ex = getenv("TGSF_SHARD_EXCHANGE");
        if (shard_may_use_rccl) R = rccl_lib();
        char bus[64] = {0};
        int node = -1;
        if (L.device_location(o.device, bus, (int)sizeof bus, &node) != TGSF_OK) snprintf(bus, sizeof bus, "device%d", o.device);
        BlobOut mine;
        mine.str(bus);
        mine.pod<int>(R ? 1 : 0);
        const std::vector<std::string> all = link.gather(mine.s);
        std::string verdict(1, '0');
        if (link.rank == 0) {
            std::vector<std::string> seen;
            bool ok = R != nullptr;
            for (const std::string& a : all) {
                BlobIn in2(a);
                std::string b2; int have = 0;
                in2.str(b2); in2.pod(have);
                ok = ok && have && std::find(seen.begin(), seen.end(), b2) == seen.end();
                seen.push_back(b2);
            }
            if (ok) {
                char id[TGSF_RCCL_ID_BYTES];
                if (R->unique_id(id) == TGSF_OK) { verdict.assign(1, '1'); verdict.append(id, sizeof id); }
                else if (ex && !strcmp(ex, "rccl")) die(std::string("TGSF_SHARD_EXCHANGE=rccl: ") + R->last_error());
            } else if (ex && !strcmp(ex, "rccl")) die("TGSF_SHARD_EXCHANGE=rccl: ranks share a GPU, or libtgsf_rccl.so does not load on every rank");
        }
        link.bcast(verdict);
        use_rccl = verdict[0] == '1' && verdict.size() == 1 + TGSF_RCCL_ID_BYTES;
        if (use_rccl)
            // (the helper owns its state: should the communicator never come up, the run goes on without it -- below -- and the
            // helper is left behind where it waits)
            rccl_up = std::thread([R, up = rccl_state, verdict, device = o.device, rank = link.rank, world = link.world] {
                if (const char* e = knob("TGSF_RCCL_STALL_S")) usleep((useconds_t)(atof(e) * 1e6));       // test knob: a communicator that is late
                up->rc = R->comm_init(device, verdict.data() + 1, rank, world, &up->comm);
                if (up->rc != TGSF_OK) up->err = R->last_error();       // (thread-local text: taken on this thread)
                up->done.store(1, std::memory_order_release);
            });
    }
    tgsf_params p;
    memset(&p, 0, sizeof p);
    p.struct_size = sizeof p;
    p.min_len = o.min_len; p.max_len = o.max_len; p.min_q = o.min_q < 0 ? 0.f : o.min_q; p.max_q = o.max_q;
    p.bc_len = o.bc_len; p.head_trim = o.head_trim < 0 ? 0 : o.head_trim; p.tail_trim = o.tail_trim < 0 ? 0 : o.tail_trim;
    p.end_len = o.end_len; p.end_match_len = o.end_match_len; p.mid_match_len = o.mid_match_len; p.extra_len = o.extra_len;
    p.end_sim = o.end_sim; p.mid_sim = o.mid_sim; p.discard = o.discard; p.filter = o.filter; p.only_qc = o.only_qc;
    p.min_repeat = o.min_repeat; p.kmer = o.kmer; p.qtype = pp.qtype ? pp.qtype : 33;
    p.no_qual = fasta_in ? 1 : 0;
    if (adapters.size() > TGSF_MAX_ADAPTERS) die("more than " + std::to_string(TGSF_MAX_ADAPTERS) + " adapter sequences");
    p.n_adapters = (int)adapters.size();
    for (size_t a = 0; a < adapters.size(); a++) { p.adapters[a] = adapters[a].data(); p.adapter_len[a] = (int)adapters[a].size(); }
    p.max_batch_reads = batch_reads;
    // rows of the per-100-bp tables: from the longest read when the index is complete by now (it usually is: it runs
    // at tens of GB/s beside the device bring-up), else from what the file could hold
    if (streaming) p.max_read_len = 1u << 26;
    else if (sharded) {
        // every rank needs the same table rows (the all-reduce sums vectors of one layout): the longest read of the whole
        // job when the parts are indexed in a moment anyway, what the file could hold otherwise
        if (in.size() <= (1ull << 30)) { records_p->wait_complete(); p.max_read_len = (uint32_t)std::max<uint64_t>(link.max_u64(records_p->longest()), 1024); }
        else p.max_read_len = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(in.size() / 2, 1024), 1u << 26);
    }
    else if (records_p->complete()) p.max_read_len = std::max<uint32_t>(records_p->longest(), 1024);
    else p.max_read_len = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(in.size() / 2, 1024), 1u << 26);
    // capacity is in buffer bytes: a text slice must fit, and so must one record of the longest read on its own
    // (header + two lines of max_read_len)
    p.max_batch_bases = std::max<uint64_t>(batch_text, 2ull * p.max_read_len + (1u << 16)) + (1 << 20);
    // Per device of --devices: a few contexts, each with its own feeder thread.  A feeder's tgsf_submit is
    // synchronous (H2D from the pageable input mapping, kernels, D2H): one of them moves ~26 GB/s over the link, two
    // or three together saturate it (~55 GB/s) and keep the kernels of one batch under the copy of another.  Batches
    // are dealt to whichever feeder is free, the planner re-sequences them, the tallies are merged at the end
    // (host side).
    int per_dev = (streaming || text_size > (64u << 20)) ? 3 : 1;
    if (const char* e = getenv("TGSF_CTX_PER_DEVICE")) { const int v = atoi(e); if (v >= 1 && v <= 8) per_dev = v; }
    std::vector<int> ctx_dev;
    for (int d : o.devices) for (int k = 0; k < per_dev; k++) ctx_dev.push_back(d);
    std::vector<tgsf_ctx*> ctxs(ctx_dev.size(), nullptr);
    const double t_p0 = now_s();

    // ---- pipeline ----
    Channel<std::shared_ptr<Batch>> to_gpu(2 + ctxs.size()), to_writer(256);
    std::vector<int> raw_lens, clean_lens;
    uint64_t raw_bases = 0, clean_bases = 0;
    std::vector<CleanRec> clean_recs;                                  // only filled when downsampling follows
    if (!run_filter_pass) {                                            // get_fastx_SeqLen, :2256-2269
        RecordIndex::Cursor rd(*records_p);
        Rec r;
        while (rd.next(r)) {
            clean_recs.push_back({std::string_view(r.name, r.name_len), 1, r.seq, r.qual, r.len});
            clean_bases += r.len;
        }
    }

    BatchStore store;
    std::atomic<uint64_t> stream_text{0};                             // streamed input: text handed out so far ...
    std::atomic<double> stream_share{0.0};                            // ... out of this share of the file's bytes
    std::thread reader([&] {                                           // read_fastx, :1845-1870 (batches of indexed records)
        CpuScope cpu(CPU_BATCHER);
        if (!run_filter_pass) { for (size_t d = 0; d < ctxs.size(); d++) to_gpu.put(nullptr); return; }
        std::unique_ptr<RecordIndex::Cursor> rd;
        if (!streaming) rd.reset(new RecordIndex::Cursor(*records_p));
        std::unique_ptr<ChunkReader> cr;
        if (streaming) cr = open_stream();
        std::shared_ptr<Chunk> ch;                                     // streamed input: the chunk being dealt into batches
        size_t ch_at = 0;
        bool over = false;
        Rec r;
        auto fresh = [&] { return store.get(); };
        std::shared_ptr<Batch> b = fresh();
        const double t0 = now_s();
        double waited = 0;
        uint64_t next_id = 0;
        auto flush = [&] {
            if (b->recs.empty()) return;
            const double w0 = now_s();
            b->id = next_id++;
            to_gpu.put(std::move(b));
            waited += now_s() - w0;
            b = fresh();
        };
        auto next_record = [&]() {
            if (!streaming) return rd->next(r);
            while (!over && (!ch || ch_at >= ch->recs.size())) {
                flush();                                               // a batch never spans two chunks
                if (ch && !ch->message.empty()) std::cerr << ch->message << std::endl;
                if (ch && ch->last) { over = true; break; }
                ch = cr->next(o.in_file);
                ch_at = 0;
                if (!ch) { over = true; break; }
                stream_text.store(cr->text_bytes());
                stream_share.store(cr->consumed());
            }
            if (over) return false;
            r = ch->recs[ch_at++];
            return true;
        };
        // (a sharded job: the cuts are checked against what the reader sees, see text_off above)
        auto bad_cut = [&](size_t at) {
            die("--ranks / --shard: the input cannot be cut near byte " + std::to_string(at) + " the way one sequential reader would read it "
                "(lines the reader skips, or a malformed record, near there): run it without --ranks / --shard");
        };
        const char* last_end = nullptr;
        while (next_record()) {
            const size_t L = r.len;
            if (L > p.max_read_len) die("read longer than the supported maximum");
            const char* rec_end = (fasta_in ? r.seq : r.qual) + L;
            if (sharded && !last_end && link.rank > 0 && r.name != text + 1) bad_cut(text_off);
            last_end = rec_end;
            if (!b->recs.empty() && ((uint64_t)(rec_end - b->base) > batch_text || b->recs.size() >= batch_reads)) flush();
            if (b->recs.empty()) { b->base = r.name; b->hold = ch; }
            b->off.push_back((uint64_t)(r.seq - b->base));
            b->qoff.push_back(fasta_in ? b->off.back() : (uint64_t)(r.qual - b->base));
            b->len.push_back((uint32_t)L); b->recs.push_back(r);
            b->span = (uint64_t)(rec_end - b->base);
            if (b->span > p.max_batch_bases) die("record larger than a batch");
            b->bases += L;
            raw_bases += L; raw_lens.push_back((int)L);
        }
        if (sharded && link.rank + 1 < link.world) {                   // the last record ends exactly where the next rank begins
            const char* e = last_end ? last_end : text;
            const char* const end = text + text_size;
            if (e < end && *e == '\r') e++;
            if (e < end && *e == '\n') e++;
            if (e != end || !records_p->end_message().empty()) bad_cut(text_off + text_size);
        }
        flush();
        for (size_t d = 0; d < ctxs.size(); d++) to_gpu.put(nullptr);       // one end marker per feeder
        t_parse = now_s() - t0 - waited;
        std::sort(raw_lens.begin(), raw_lens.end());                   // for the statistics (:3151), beside the rest of the pipeline
    });

    std::mutex gpu_time_m;
    // Several GPUs: a device's feeders -- and the pinned staging buffers tgsf_create allocates from them -- stay
    // on the CPUs of the GPU's own NUMA node, so that no feeder pushes its copies across the socket link.  With one GPU
    // binding was measured within noise and is left off; TGSF_NUMA=1 / 0 forces it on / off.
    // (a job of rank processes on GPUs of their own is the same case, one device per process)
    bool numa_bind = o.devices.size() > 1 || (link.world > 1 && shard_may_use_rccl);
    if (const char* e = getenv("TGSF_NUMA")) numa_bind = atoi(e) > 0;
    std::vector<int> dev_node(ctx_dev.size(), -1);
    std::vector<double> dev_submit_s(ctx_dev.size(), 0.0);
    std::vector<uint64_t> dev_bytes(ctx_dev.size(), 0), dev_batches(ctx_dev.size(), 0);
    auto feed = [&](size_t k) {                                        // filter_sequence, :1919-2064, one batch per call
        CpuScope cpu(CPU_FEEDER);
        if (numa_bind) {
            int node = -1;
            char bus[64];
            if (L.device_location(ctx_dev[k], bus, (int)sizeof bus, &node) == TGSF_OK && node >= 0) {
                std::ifstream f("/sys/devices/system/node/node" + std::to_string(node) + "/cpulist");
                std::string list;
                cpu_set_t set;
                CPU_ZERO(&set);
                int n_set = 0;
                if (f && std::getline(f, list)) {                      // "0-63,128-191"
                    size_t i = 0;
                    while (i < list.size()) {
                        const int a = atoi(list.c_str() + i);
                        int b = a;
                        size_t j = list.find_first_of(",-", i);
                        if (j != std::string::npos && list[j] == '-') { b = atoi(list.c_str() + j + 1); j = list.find(',', j); }
                        for (int c2 = a; c2 <= b && c2 < CPU_SETSIZE; c2++) { CPU_SET(c2, &set); n_set++; }
                        i = j == std::string::npos ? list.size() : j + 1;
                    }
                }
                // within what the caller allows (taskset, numactl, a container's cpuset): never a wider mask than it came with
                cpu_set_t allowed;
                if (n_set > 0 && sched_getaffinity(0, sizeof allowed, &allowed) == 0) {
                    n_set = 0;
                    for (int c2 = 0; c2 < CPU_SETSIZE; c2++) {
                        if (CPU_ISSET(c2, &set) && !CPU_ISSET(c2, &allowed)) CPU_CLR(c2, &set);
                        if (CPU_ISSET(c2, &set)) n_set++;
                    }
                }
                if (n_set > 0 && sched_setaffinity(0, sizeof set, &set) == 0) dev_node[k] = node;
            }
        }
        if (L.create(&p, ctx_dev[k], &ctxs[k]) != TGSF_OK) die(L.last_error(nullptr));
        tgsf_ctx* fctx = ctxs[k];
        if (timing) (void)L.profile(fctx, 1);                          // HIP events around the stages of every batch (GPU: line)
        for (;;) {
            std::shared_ptr<Batch> b = to_gpu.get();
            if (!b) break;
            const double g0 = now_s();
            if (b->res.size() < b->recs.size()) b->res.resize(b->recs.size());
            const size_t fneed = (size_t)(b->bases / (uint64_t)std::max(p.min_len, 1)) + b->recs.size() + 16;
            if (b->frags.size() < fneed) b->frags.resize(fneed);
            const uint8_t* text = reinterpret_cast<const uint8_t*>(b->base);
            tgsf_batch_in bi;
            memset(&bi, 0, sizeof bi);
            bi.seq = text; bi.qual = text;                             // one buffer: the FASTQ text itself
            bi.offsets = b->off.data(); bi.qual_offsets = b->qoff.data(); bi.lengths = b->len.data();
            bi.n_reads = (uint32_t)b->recs.size(); bi.n_bytes = b->span;
            tgsf_batch_out bo{b->res.data(), b->frags.data(), (uint32_t)b->frags.size(), 0};
            if (L.submit(fctx, &bi, &bo) != TGSF_OK) die(L.last_error(fctx));
            { std::lock_guard<std::mutex> l(gpu_time_m); t_gpu += now_s() - g0; if (t_first == 0) t_first = now_s() - t_p0; }
            dev_submit_s[k] += now_s() - g0; dev_bytes[k] += b->span; dev_batches[k]++;
            b->n_frags = bo.n_frags;
            to_writer.put(std::move(b));
        }
        to_writer.put(nullptr);
    };
    std::vector<std::thread> feeders;
    for (size_t k = 0; k < ctxs.size(); k++) feeders.emplace_back(feed, k);

    // record formatting :2011-2053 + write_output :2095-2145.  The planner takes the batches in input order (= the
    // reference's -t 1 order), lays the records of a batch out in the output file and hands runs of them to the fill
    // threads (MappedSink); or, for the other kinds of output, gathers the pieces and writes them itself (Output).
    int fill_threads = std::max(1, std::min(o.n_thread, 16));
    uint64_t fill_min = 1u << 20;                                      // bytes worth a job of their own
    if (const char* e = knob("TGSF_FILL_MIN_BYTES")) { const long long v = atoll(e); if (v > 0) fill_min = (uint64_t)v; }   // test knob
    Pool pool(sink.is_open() ? fill_threads : 1);
    int populate_threads = std::max(1, std::min(o.n_thread, 32));      // short bursts between two fallocates: the more the shorter
    if (const char* e = knob("TGSF_POPULATE_THREADS")) { const int v = atoi(e); if (v >= 1 && v <= 64) populate_threads = v; }   // tuning knob (tests/manual/e2e_cpu.py)
    // (the pages of a reserved stride are mapped by many threads BETWEEN two fallocates: beside one, page faults on the file
    // take its inode's lock and both crawl -- measured, DESIGN appendix)
    Pool populate(sink.is_open() ? populate_threads : 0, CPU_POPULATE);
    // (a streamed input is decoder-bound: small strides keep the mapped part of the output -- it counts as resident -- small)
    uint64_t stride_bytes = streaming ? (128ull << 20) : (2ull << 30);
    if (const char* e = knob("TGSF_STRIDE_BYTES")) { const long long v = atoll(e); if (v > 0) stride_bytes = (uint64_t)v; }   // tuning / test knob
    // While the library loads and the device comes up pages of the output file are instantiated already, up to a quarter
    // of the input's size (what a run keeps is not known yet; a surplus is cut off at the end).
    end_early();                                                       // (what it reserved is mapped by the reserver's first round)
    Reserver reserver(sink, populate, stride_bytes, false);
    if (sink.is_open())
        reserver.start((!streaming && text_size > (256u << 20)) ? (uint64_t)text_size / 4 : 0);
    // A downsampling run writes its output only after the whole filter pass (the selection needs every fragment's length,
    // :2297-2344) -- but the file can be instantiated meanwhile: while the filter pass is busy with the link to the device,
    // pages for what the selection may keep are reserved and mapped (a quarter of the input at most, and no more than twice
    // the bases asked for with -g/-d; a surplus is cut off at the end).  Large plain outputs only.
    std::unique_ptr<MappedSink> dsink;
    std::unique_ptr<Pool> dpop;
    std::unique_ptr<Reserver> dres;
    auto open_dsink = [&](uint64_t capacity, uint64_t speculative) {
        const char* w = getenv("TGSF_WRITER");
        if (o.out_gz || o.out_file.empty() || (w && !strcmp(w, "writev"))) return false;
        std::unique_ptr<MappedSink> d(new MappedSink);
        if (!d->open(out_path, capacity)) return false;
        dsink = std::move(d);
        dpop.reset(new Pool(populate_threads, CPU_POPULATE));
        dres.reset(new Reserver(*dsink, *dpop, stride_bytes, false));
        dres->start(speculative);
        return true;
    };
    {
        uint64_t early_min = 1ull << 30;
        if (const char* e = knob("TGSF_DOWN_EARLY_MIN")) early_min = strtoull(e, nullptr, 10);          // tests: small inputs too
        if (o.downsample && !streaming && in.mapped() && (uint64_t)text_size >= early_min) {
            uint64_t spec = (uint64_t)text_size / 4;
            if (o.genome_size > 0 && o.desired_depth > 0) spec = std::min<uint64_t>(spec, (2 * o.genome_size * (uint64_t)o.desired_depth) / (uint64_t)link.world + (uint64_t)text_size / 64);
            open_dsink(4 * (uint64_t)text_size + (1ull << 30), spec);
        }
    }
    // Mappings of written batches (input text, output file).  One process (the default): the teardown is on the caller's
    // clock -- 90 ns per page of the input, ~200 ns per dirty page of the output if it all waited for the exit -- so the
    // mappings are dropped piece by piece during the run, by ONE background thread (several only get in each other's way).
    // Both mappings carry MADV_SEQUENTIAL: unmapping a page of a mapping without it marks the page accessed, and 20 M pages
    // moving between the LRU lists slow the fallocate beside them by a quarter.  TGSF_DETACH=1: nothing is
    // dropped during the run (the child's address space goes in the background), except where the resident size matters
    // (a streamed input: the mapped part of the output counts as resident).
    const bool sync_exit = g_done_fd < 0;                              // one process (the default): the teardown is on the clock
    bool release_input = sync_exit && !streaming && in.mapped() && !o.downsample;
    bool release_output = sync_exit || streaming;
    const uint64_t release_piece = 16u << 20;
    Channel<std::pair<const char*, uint64_t>> to_release(1 << 16);
    std::thread releaser([&] {
        CpuScope cpu(CPU_RELEASER);
        for (;;) {
            const std::pair<const char*, uint64_t> r = to_release.get();
            if (!r.first) break;
            for (uint64_t o2 = 0; o2 < r.second; o2 += release_piece) MappedSink::release(r.first + o2, std::min<uint64_t>(release_piece, r.second - o2));
        }
    });
    using Emit = Batch::Emit;
    // a written batch: its mappings go to the releaser thread where they are dropped at all (from the 16 fill threads at
    // once that cost 7-15 thread-seconds of a run and slowed everything beside them), the batch itself back to the store
    auto batch_done = [&](std::shared_ptr<Batch> b) {
        if (b->out_bytes && release_output) to_release.put({b->dst, b->out_bytes});
        if (release_input) to_release.put({b->base, b->span});
        store.put(std::move(b));
    };
    std::thread writer([&] {
        CpuScope cpu(CPU_PLANNER);
        const std::string lead(1, fastq_out ? '@' : '>'), nl("\n"), sep("\n+\n");
        std::string name;
        std::map<uint64_t, std::shared_ptr<Batch>> held;               // batches that arrived ahead of their turn
        uint64_t want = 0, in_seen = 0;
        size_t open_feeders = ctxs.size();
        for (;;) {
            std::shared_ptr<Batch> b;
            auto it = held.find(want);
            if (it != held.end()) { b = std::move(it->second); held.erase(it); }
            else {
                if (open_feeders == 0) break;
                const double i0 = now_s();
                b = to_writer.get();
                t_widle += now_s() - i0;
                if (!b) { open_feeders--; continue; }
                if (b->id != want) { const uint64_t id = b->id; held[id] = std::move(b); continue; }
            }
            want++;
            const double w0 = now_s();
            const bool fill = sink.is_open();
            uint64_t at = 0;
            for (size_t r = 0; r < b->recs.size(); r++) {
                int pass_num = 1;
                const tgsf_read_result& rr = b->res[r];
                const Rec& rec = b->recs[r];
                const std::string_view rname(rec.name, rec.name_len);
                for (uint32_t f = rr.frag_begin; f < rr.frag_begin + rr.n_frags; f++) {
                    const tgsf_fragment& fr = b->frags[f];
                    if (!(fr.flags & TGSF_FF_PASS)) continue;
                    if (o.downsample) {                                // kept in memory instead of a tmp file (:3129-3137)
                        clean_recs.push_back({rname, pass_num++, rec.seq + fr.start, rec.qual + fr.start, (uint32_t)fr.len});
                        clean_bases += (uint64_t)fr.len;
                        clean_lens.push_back(fr.len);
                        continue;
                    }
                    clean_bases += (uint64_t)fr.len;
                    clean_lens.push_back(fr.len);
                    if (o.only_qc) { pass_num++; continue; }
                    if (fill) {
                        b->em.push_back({(uint32_t)r, f, pass_num, at});
                        size_t nlen = rname.size();
                        if (pass_num >= 2) { int v = pass_num; nlen += 1; while (v) { nlen++; v /= 10; } }
                        at += 1 + nlen + 1 + (uint64_t)fr.len + (fastq_out ? 3 + (uint64_t)fr.len : 0) + 1;
                        pass_num++;
                        continue;
                    }
                    out.text(lead);
                    if (pass_num < 2) out.piece(rname.data(), rname.size());
                    else { name.clear(); append_name(name, rname, pass_num); out.text(name); }
                    pass_num++;
                    out.text(nl);
                    out.piece(rec.seq + fr.start, (size_t)fr.len);
                    if (fastq_out) {
                        out.text(sep);
                        out.piece(rec.qual + fr.start, (size_t)fr.len);
                    }
                    out.text(nl);
                    out.end_record();
                }
            }
            in_seen += b->span;
            if (fill && at) {
                if (sink.planned() + at > sink.capacity()) die("output more than four times the size of the input: larger than the space mapped for it (TGSF_WRITER=writev writes such a file)");
                {
                    // How far the file will go: what is left of the input times the share of it that was written so far
                    // (plus a little).  (A streamed input's text size is estimated from the share of the file decoded so far.)
                    const double share = in_seen ? (double)(sink.planned() + at) / (double)in_seen : 1.0;
                    const double sh = stream_share.load();
                    const uint64_t in_total = !streaming ? (uint64_t)text_size
                                            : (uint64_t)((double)stream_text.load() / (sh > 1e-6 ? sh : 1e-6));
                    uint64_t goal = sink.planned() + at + (uint64_t)(share * 1.02 * (double)(in_total - std::min<uint64_t>(in_seen, in_total)));
                    goal = std::min<uint64_t>(std::max<uint64_t>(goal, sink.planned() + at), sink.capacity());
                    reserver.want(goal, sink.planned() + at);
                    const double d0 = now_s();
                    reserver.wait_ready(sink.planned() + at);            // instantiated AND mapped: the fill jobs take no fault
                    t_drain += now_s() - d0;
                }
                b->dst = sink.place(at);
                b->out_bytes = at;
                const size_t n = b->em.size();
                const int parts = (int)std::min<size_t>((size_t)fill_threads, std::max<size_t>(1, at / fill_min));
                b->left = parts;
                size_t lo = 0;
                for (int k = 0; k < parts; k++) {                      // byte-balanced runs of records
                    size_t hi = n;
                    if (k + 1 < parts) {
                        const uint64_t target = at / (uint64_t)parts * (uint64_t)(k + 1);
                        hi = (size_t)(std::lower_bound(b->em.begin() + (long)lo, b->em.end(), target,
                                                       [](const Emit& e, uint64_t t) { return e.at < t; }) - b->em.begin());
                    }
                    pool.add([b, lo, hi, fastq_out, &batch_done] {
                        Batch& bb = *b;
                        std::string nm;
                        for (size_t i = lo; i < hi; i++) {
                            const Emit& e = bb.em[i];
                            const Rec& rec = bb.recs[e.read];
                            const tgsf_fragment& fr = bb.frags[e.frag];
                            char* d = bb.dst + e.at;
                            *d++ = fastq_out ? '@' : '>';
                            if (e.pass_num < 2) { memcpy(d, rec.name, rec.name_len); d += rec.name_len; }
                            else { nm.clear(); append_name(nm, std::string_view(rec.name, rec.name_len), e.pass_num); memcpy(d, nm.data(), nm.size()); d += nm.size(); }
                            *d++ = '\n';
                            stream_copy(d, rec.seq + fr.start, (size_t)fr.len); d += fr.len;
                            if (fastq_out) {
                                memcpy(d, "\n+\n", 3); d += 3;
                                stream_copy(d, rec.qual + fr.start, (size_t)fr.len); d += fr.len;
                            }
                            *d++ = '\n';
                        }
                        stream_fence();
                        if (--bb.left == 0) batch_done(b);
                    });
                    lo = hi;
                }
            }
            if (!o.only_qc && !fill) out.flush_iov();                   // the batch (and its views) goes away
            if (!(fill && at)) batch_done(b);                           // written (or nothing to write)
            t_write += now_s() - w0;
        }
        if (!o.downsample) std::sort(clean_lens.begin(), clean_lens.end());   // for the statistics (:3182), beside the last fill jobs
    });
    reader.join();
    for (std::thread& f : feeders) f.join();
    writer.join();
    reserver.finish();
    const bool mapped_out = sink.is_open();
    const double t_f0 = now_s();
    pool.finish();
    populate.finish();
    to_release.put({nullptr, 0});
    releaser.join();
    const double t_busy = pool.busy_s();
    const double t_fill_tail = now_s() - t_f0;
    sink.close();
    const double t_close = now_s() - t_f0 - t_fill_tail;
    t_pipe = now_s() - t_p0;
    tgsf_ctx* ctx = ctxs[0];

    // A rank's tally vector as it travels to rank 0 over the sockets: everything in front of the four per-100-bp tables, then
    // of each of those only the rows in use; and its sum into rank 0's vector (the four "rows used" words are maxima).
    auto pack_rows = [](const std::vector<uint64_t>& v, int32_t bc2, uint32_t nb2) {
        const size_t head = tgsf_ctr_bin_table(0, bc2, nb2);
        std::vector<uint64_t> out(v.begin(), v.begin() + (long)head);
        for (int b = 0; b < 4; b++) {
            const size_t at = tgsf_ctr_bin_table(b, bc2, nb2), n = (size_t)std::min<uint64_t>(v[TGSF_CTR_ROWS + (b >> 1)], nb2) * 5;
            out.insert(out.end(), v.begin() + (long)at, v.begin() + (long)(at + n));
        }
        return out;
    };
    auto add_rows = [](std::vector<uint64_t>& v, const std::vector<uint64_t>& ru, int32_t bc2, uint32_t nb2) {
        const size_t head = tgsf_ctr_bin_table(0, bc2, nb2);
        if (ru.size() < head) die("a rank of the job sent a tally vector of another layout");
        uint64_t rows[4];
        for (int q = 0; q < 4; q++) rows[q] = std::max(v[TGSF_CTR_ROWS + q], ru[TGSF_CTR_ROWS + q]);
        for (size_t i = 0; i < head; i++) v[i] += ru[i];
        size_t from = head;
        for (int b = 0; b < 4; b++) {
            const size_t at = tgsf_ctr_bin_table(b, bc2, nb2), n = (size_t)std::min<uint64_t>(ru[TGSF_CTR_ROWS + (b >> 1)], nb2) * 5;
            if (from + n > ru.size()) die("a rank of the job sent a tally vector of another layout");
            for (size_t i = 0; i < n; i++) v[at + i] += ru[from + i];
            from += n;
        }
        for (int q = 0; q < 4; q++) v[TGSF_CTR_ROWS + q] = rows[q];
    };

    // ---- downsampling: DownSampleTask, :2164-2568 ----
    // keep the longest reads until the target is met (:2297-2344), then a QC-only pass over the kept reads
    // (CalcAvgQuality / Get_5p/3p_base_qual again, :2436-2447) which also writes them, in input order.
    uint64_t down_bases = 0, down_job_recs = 0, down_job_bases = 0;
    std::vector<int> down_lens;
    std::vector<uint64_t> down_t;
    double t_d0 = now_s(), t_dsel = 0, t_dcreate = 0, t_dqc = 0, t_dwrite = 0, t_dclose = 0, t_dsubmit = 0, t_dfirst = 0, d_kept = 0, d_span = 0; int n_dsubmit = 0; bool d_in_place = false, d_mapped = false;
    if (o.downsample) {
        // Selection as the reference makes it (:2297-2344), container for container, so that ties at the cut fall
        // the same way when both programs are built with the same standard library: lengths keyed by record name
        // in an unordered_map filled in write order (:2105, :2267; a repeated name keeps its last length), handed
        // over by copy (:3142, :2169), listed in the map's iteration order, std::sort by length (descending),
        // names taken from the top; the second pass keeps every record whose name was taken (:2356).
        auto full_name = [&](const CleanRec& c) {
            std::string nm;
            append_name(nm, c.name, c.pass_num);
            return nm;
        };
        // (one process per GPU: the selection is over the kept fragments of ALL ranks, in input order = rank order; rank 0
        // receives every rank's names and lengths and makes it, the others wait for their keep flags)
        std::vector<std::string> all_names;                            // rank 0 of a sharded job: every fragment of the job, in input order
        std::vector<int> all_lens;
        std::vector<size_t> rank_first;                                // ... and where each rank's begin
        if (sharded) {
            BlobOut mine;
            std::vector<uint32_t> lens;
            std::string names;
            for (const CleanRec& c : clean_recs) { lens.push_back(c.len); const std::string nm = full_name(c); const uint32_t n = (uint32_t)nm.size(); names.append((const char*)&n, 4); names += nm; }
            mine.vec(lens); mine.str(names);
            const std::vector<std::string> all = link.gather(mine.s);
            for (const std::string& b : all) {
                BlobIn in2(b);
                std::vector<uint32_t> l2; std::string n2;
                in2.vec(l2); in2.str(n2);
                rank_first.push_back(all_names.size());
                size_t at = 0;
                for (uint32_t L : l2) {
                    uint32_t n = 0;
                    if (at + 4 > n2.size()) die("a rank of the job sent a list of names shorter than its lengths");
                    memcpy(&n, n2.data() + at, 4); at += 4;
                    all_names.emplace_back(n2.data() + at, n); at += n;
                    all_lens.push_back((int)L);
                }
            }
            rank_first.push_back(all_names.size());
        }
        const bool selects = !sharded || link.rank == 0;
        std::unordered_map<std::string, int> seq_lens;
        uint64_t total = 0;
        if (!sharded) for (const CleanRec& c : clean_recs) { seq_lens[full_name(c)] = (int)c.len; total += c.len; }
        else for (size_t i = 0; i < all_names.size(); i++) { seq_lens[all_names[i]] = all_lens[i]; total += (uint64_t)all_lens[i]; }
        const std::unordered_map<std::string, int> handed(seq_lens), task_lens(handed);
        std::vector<std::pair<std::string, int>> vec(task_lens.begin(), task_lens.end());
        std::sort(vec.begin(), vec.end(), [](const std::pair<std::string, int>& a, const std::pair<std::string, int>& b) {
            return a.second > b.second;
        });
        uint64_t desired = 0; int want_num = 0; bool by_size = true;
        if (o.genome_size > 0 && o.desired_depth > 0) desired = o.genome_size * (uint64_t)o.desired_depth;
        else if (o.desired_frac > 0) desired = (uint64_t)(o.desired_frac * total);        // float * uint64, :2322
        else { by_size = false; want_num = o.desired_num; }
        std::unordered_set<std::string> chosen;
        uint64_t added = 0; int added_num = 0;
        for (const auto& pr : vec) {
            if (!selects) break;
            chosen.insert(pr.first);
            added += (uint64_t)pr.second; added_num++;
            down_bases += (uint64_t)pr.second; down_lens.push_back(pr.second);
            if (by_size ? added >= desired : added_num >= want_num) break;
        }
        std::vector<char> keep(clean_recs.size(), 0);
        if (!sharded) for (size_t i = 0; i < clean_recs.size(); i++) keep[i] = chosen.count(full_name(clean_recs[i])) ? 1 : 0;
        else {
            std::vector<std::string> flags;
            if (link.rank == 0)
                for (int k = 0; k < link.world; k++) {
                    std::string f(rank_first[(size_t)k + 1] - rank_first[(size_t)k], '\0');
                    for (size_t i = 0; i < f.size(); i++) f[i] = chosen.count(all_names[rank_first[(size_t)k] + i]) ? 1 : 0;
                    flags.push_back(std::move(f));
                }
            std::string mine;
            link.scatter(flags, mine);
            if (mine.size() != keep.size()) die("the selection handed to this rank does not fit its fragments");
            if (!keep.empty()) memcpy(keep.data(), mine.data(), keep.size());
            // (the job's totals, for rank 0's statistics: every fragment that entered the selection)
            if (link.rank == 0) { down_job_recs = all_names.size(); down_job_bases = total; }
            std::vector<std::string>().swap(all_names);
        }
        t_dsel = now_s() - t_d0;
        tgsf_params qp = p;
        qp.filter = 0; qp.only_qc = 1; qp.n_adapters = 0; qp.min_repeat = 0;
        // The reference's second pass re-reads what the filter pass wrote (:3129-3137): after a FASTA output
        // (-f, or FASTA input) the records carry no qualities, so this pass takes the count-only tallies.
        const bool down_no_qual = fasta_in || (run_filter_pass && !fastq_out);
        qp.no_qual = down_no_qual ? 1 : 0;
        qp.max_batch_bases = (1ull << 30); qp.max_batch_reads = 1u << 16;
        if (const char* e = knob("TGSF_DOWN_BATCH_BYTES")) { const long long v = atoll(e); if (v > (1 << 20) + 65536) qp.max_batch_bases = (uint64_t)v; }   // test knob: several slices of a small input
        tgsf_ctx *qctx = nullptr, *qctx2 = nullptr;
        std::vector<uint8_t> bs, bq; std::vector<uint64_t> boff; std::vector<uint32_t> blen;
        std::vector<tgsf_read_result> bres; std::vector<tgsf_fragment> bfr(16);
        auto run = [&] {
            if (blen.empty()) return;
            bres.resize(blen.size());
            bs.resize(bs.size() + 64); bq.resize(bq.size() + 64);
            tgsf_batch_in bi; memset(&bi, 0, sizeof bi);
            bi.seq = bs.data(); bi.qual = bq.data(); bi.offsets = boff.data(); bi.lengths = blen.data();
            bi.n_reads = (uint32_t)blen.size(); bi.n_bytes = bs.size() - 64;
            tgsf_batch_out bo{bres.data(), bfr.data(), (uint32_t)bfr.size(), 0};
            const double s0 = now_s();
            if (L.submit(qctx, &bi, &bo) != TGSF_OK) die(L.last_error(qctx));
            t_dsubmit += now_s() - s0; if (!n_dsubmit++) t_dfirst = now_s() - s0;
            bs.clear(); bq.clear(); boff.clear(); blen.clear();
        };
        // The kept records go to the device on a thread of their own while this one writes them.  Where they make up a fair
        // part of the text between them they are read in place (the text itself is the batch, as in the filter pass: one
        // copy to the device, none on the host); a very thin selection is packed first.
        // (the writer took the batches as they came back from the feeders: the kept records are sorted by address first)
        std::vector<uint32_t> by_addr;
        uint64_t kept_bytes = 0, kept_span = 0;
        for (size_t i = 0; i < clean_recs.size(); i++) {
            if (!keep[i]) continue;
            by_addr.push_back((uint32_t)i);
            kept_bytes += (down_no_qual ? 1u : 2u) * (uint64_t)clean_recs[i].len;
        }
        std::sort(by_addr.begin(), by_addr.end(), [&](uint32_t x, uint32_t y) { return clean_recs[x].seq < clean_recs[y].seq; });
        if (!by_addr.empty()) {
            const CleanRec& a = clean_recs[by_addr.front()];
            const CleanRec& z = clean_recs[by_addr.back()];
            kept_span = (uint64_t)((down_no_qual ? z.seq : std::max(z.seq, z.qual)) + z.len - a.seq);
            for (uint32_t i : by_addr) if (!down_no_qual && clean_recs[i].qual < clean_recs[i].seq) kept_span = 0;   // (never: FASTQ text)
        }
        const char* force = knob("TGSF_DOWN_QC");                     // "text" / "packed": tests run both ways
        const bool in_place = kept_span > 0 && (force ? !strcmp(force, "text") : kept_bytes * 12 >= kept_span);   // (packing runs at a tenth of the copy to the device)
        d_in_place = in_place; d_kept = (double)kept_bytes; d_span = (double)kept_span;
        std::thread qc_pass([&] {
            CpuScope cpu(CPU_DOWNSAMPLE);
            const double q0 = now_s();
            if (L.create(&qp, o.devices[0], &qctx) != TGSF_OK) die(L.last_error(nullptr));
            t_dcreate = now_s() - q0;
            if (in_place) {
                // Slices of the text (up to 1 GB each, the kept records indexed in place) go to the device from two feeders
                // with a context each when there is much of it: one tgsf_submit stream moves 22-29 GB/s over the link, two
                // together about what it carries (as in the filter pass).
                struct TextBatch { const char* base = nullptr; uint64_t span = 0; std::vector<uint64_t> off, qoff; std::vector<uint32_t> len; };
                int workers = kept_span >= (2ull << 30) ? 2 : 1;
                if (const char* e = knob("TGSF_DOWN_FEEDERS")) workers = atoi(e) >= 2 ? 2 : 1;      // tests run both on small inputs
                Channel<std::shared_ptr<TextBatch>> todo(2);
                std::mutex tm;
                auto work = [&](tgsf_ctx* c) {
                    std::vector<tgsf_read_result> res;
                    std::vector<tgsf_fragment> fr(16);
                    for (;;) {
                        std::shared_ptr<TextBatch> tb = todo.get();
                        if (!tb) break;
                        res.resize(tb->len.size());
                        tgsf_batch_in bi; memset(&bi, 0, sizeof bi);
                        bi.seq = bi.qual = reinterpret_cast<const uint8_t*>(tb->base);
                        bi.offsets = tb->off.data(); bi.qual_offsets = tb->qoff.data(); bi.lengths = tb->len.data();
                        bi.n_reads = (uint32_t)tb->len.size(); bi.n_bytes = tb->span;
                        tgsf_batch_out bo{res.data(), fr.data(), (uint32_t)fr.size(), 0};
                        const double s0 = now_s();
                        if (L.submit(c, &bi, &bo) != TGSF_OK) die(L.last_error(c));
                        std::lock_guard<std::mutex> l(tm);
                        t_dsubmit += now_s() - s0; if (!n_dsubmit++) t_dfirst = now_s() - s0;
                    }
                };
                std::thread second;
                if (workers == 2) second = std::thread([&] {
                    CpuScope cpu2(CPU_DOWNSAMPLE);
                    if (L.create(&qp, o.devices[0], &qctx2) != TGSF_OK) die(L.last_error(nullptr));
                    work(qctx2);
                });
                std::thread first([&] { CpuScope cpu2(CPU_DOWNSAMPLE); work(qctx); });
                std::shared_ptr<TextBatch> tb(new TextBatch);
                auto flush_text = [&] {
                    if (tb->len.empty()) return;
                    todo.put(std::move(tb));
                    tb.reset(new TextBatch);
                };
                for (uint32_t i : by_addr) {
                    const CleanRec& c = clean_recs[i];
                    const char* e = down_no_qual ? c.seq + c.len : c.qual + c.len;
                    if (tb->base && ((uint64_t)(e - tb->base) > qp.max_batch_bases - (1u << 20) || tb->len.size() >= qp.max_batch_reads)) flush_text();
                    if (!tb->base) tb->base = c.seq;
                    tb->off.push_back((uint64_t)(c.seq - tb->base));
                    tb->qoff.push_back((uint64_t)((down_no_qual ? c.seq : c.qual) - tb->base));
                    tb->len.push_back(c.len);
                    tb->span = std::max(tb->span, (uint64_t)(e - tb->base));
                    if (tb->span > qp.max_batch_bases) die("record larger than a batch");
                }
                flush_text();
                for (int k = 0; k < workers; k++) todo.put(nullptr);
                first.join();
                if (second.joinable()) second.join();
            } else {
                const size_t room = (size_t)std::min<uint64_t>(qp.max_batch_bases, kept_bytes / (down_no_qual ? 1 : 2) + 16 * by_addr.size() + 128);
                bs.reserve(room); bq.reserve(room);
                for (size_t i = 0; i < clean_recs.size(); i++) {
                    if (!keep[i]) continue;
                    const CleanRec& c = clean_recs[i];
                    if (bs.size() + c.len > qp.max_batch_bases - (1u << 20) || blen.size() >= qp.max_batch_reads) run();
                    const size_t o0 = (bs.size() + 15) & ~size_t(15);
                    bs.resize(o0); bq.resize(o0);
                    bs.insert(bs.end(), c.seq, c.seq + c.len);
                    if (!down_no_qual) bq.insert(bq.end(), c.qual, c.qual + c.len); else bq.resize(bs.size());
                    boff.push_back(o0); blen.push_back(c.len);
                }
                run();
            }
            t_dqc = now_s() - q0 - t_dcreate;
        });
        const double w0 = now_s();
        const std::string lead(1, fastq_out ? '@' : '>'), nl("\n"), sep("\n+\n");
        std::string name;
        // A regular file of some size is written as the filter pass writes its own (MappedSink): every record's place is
        // known, so the pages are instantiated in one go and threads copy the records in -- a single writev stream is a
        // 6-GB/s copy under the inode lock.
        bool down_mapped = false;
        {
            const char* mn = knob("TGSF_DOWN_MAP_MIN");              // tests force the mapped way on small outputs
            const uint64_t map_min = mn ? strtoull(mn, nullptr, 10) : (256ull << 20);
            std::vector<uint32_t> kept;
            std::vector<uint64_t> at;
            uint64_t total_out = 0;
            const char* w2 = getenv("TGSF_WRITER");
            if (!o.out_gz && !o.out_file.empty() && !(w2 && !strcmp(w2, "writev"))) {
                for (size_t i = 0; i < clean_recs.size(); i++) {
                    if (!keep[i]) continue;
                    const CleanRec& c = clean_recs[i];
                    size_t nm = c.name.size();
                    if (c.pass_num >= 2) { name.clear(); append_name(name, c.name, c.pass_num); nm = name.size(); }
                    kept.push_back((uint32_t)i); at.push_back(total_out);
                    total_out += 1 + nm + 1 + c.len + (fastq_out ? 3 + (uint64_t)c.len : 0) + 1;
                }
            }
            if (!dsink && total_out >= std::max<uint64_t>(map_min, 1)) open_dsink(total_out, 0);
            if (dsink && !kept.empty()) {
                // stride by stride, as the filter pass writes its own output: the reserver instantiates and maps the file
                // ahead, the records of every piece that is ready are copied in by the pool's threads
                dres->want(total_out, total_out);
                const int T = std::max(1, std::min(o.n_thread, 16));
                Pool dfill(T);
                char* const base = dsink->place(0);
                // (one process: the mappings of written pieces are dropped behind the fill jobs by one thread, as in the filter pass)
                Channel<std::pair<const char*, uint64_t>> dropped(1 << 12);
                std::thread dropper([&] {
                    CpuScope cpu2(CPU_RELEASER);
                    for (;;) {
                        const std::pair<const char*, uint64_t> r = dropped.get();
                        if (!r.first) break;
                        MappedSink::release(r.first, r.second);
                    }
                });
                const uint64_t piece = std::max<uint64_t>(1, std::min<uint64_t>(64ull << 20, stride_bytes / 4 + 1));
                size_t r0 = 0;
                while (r0 < kept.size()) {
                    size_t r1 = (size_t)(std::lower_bound(at.begin() + (long)r0, at.end(), at[r0] + piece) - at.begin());
                    if (r1 <= r0) r1 = r0 + 1;
                    const uint64_t end = r1 < at.size() ? at[r1] : total_out;
                    dres->wait_ready(end);
                    dfill.add([&, r0, r1] {
                        std::string nm;
                        for (size_t r = r0; r < r1; r++) {
                            const CleanRec& c = clean_recs[kept[r]];
                            char* p = base + at[r];
                            *p++ = fastq_out ? '@' : '>';
                            if (c.pass_num < 2) { memcpy(p, c.name.data(), c.name.size()); p += c.name.size(); }
                            else { nm.clear(); append_name(nm, c.name, c.pass_num); memcpy(p, nm.data(), nm.size()); p += nm.size(); }
                            *p++ = '\n';
                            stream_copy(p, c.seq, c.len); p += c.len;
                            if (fastq_out) { memcpy(p, "\n+\n", 3); p += 3; stream_copy(p, c.qual, c.len); p += c.len; }
                            *p++ = '\n';
                        }
                        stream_fence();
                        if (release_output) dropped.put({base + at[r0], (r1 < at.size() ? at[r1] : total_out) - at[r0]});
                    });
                    r0 = r1;
                }
                dfill.finish();
                dropped.put({nullptr, 0});
                dropper.join();
                dres->finish();
                dpop->finish();
                dsink->place(total_out);
                dsink->close();
                down_mapped = true;
            } else if (dsink) {                                        // nothing kept: an empty file
                dres->finish(); dpop->finish(); dsink->place(0); dsink->close(); down_mapped = true;
            }
        }
        for (size_t i = 0; i < clean_recs.size() && !down_mapped; i++) {
            if (!keep[i]) continue;
            const CleanRec& c = clean_recs[i];
            out.text(lead);
            if (c.pass_num < 2) out.piece(c.name.data(), c.name.size());
            else { name.clear(); append_name(name, c.name, c.pass_num); out.text(name); }
            out.text(nl);
            out.piece(c.seq, c.len);
            if (fastq_out) { out.text(sep); out.piece(c.qual, c.len); }
            out.text(nl);
            out.end_record();
        }
        t_dwrite = now_s() - w0; d_mapped = down_mapped;
        qc_pass.join();
        uint64_t qnw = 0; int32_t qbc = 0; uint32_t qnb = 0;
        L.counters_len(qctx, &qnw, &qbc, &qnb);
        down_t.resize(qnw);
        if (L.counters(qctx, down_t.data(), qnw) != TGSF_OK) die(L.last_error(qctx));
        L.destroy(qctx);
        if (qctx2) {                                                   // the second feeder's tallies: sums, maxima for the "rows used" words
            std::vector<uint64_t> t2(qnw);
            if (L.counters(qctx2, t2.data(), qnw) != TGSF_OK) die(L.last_error(qctx2));
            L.destroy(qctx2);
            for (uint64_t i = 0; i < qnw; i++)
                down_t[i] = (i >= TGSF_CTR_ROWS && i < TGSF_CTR_ROWS + 4) ? std::max(down_t[i], t2[i]) : down_t[i] + t2[i];
        }
        if (sharded) {                                                 // the second pass's tallies of the whole job, on rank 0
            BlobOut mine;
            mine.vec(pack_rows(down_t, qbc, qnb));
            const std::vector<std::string> all = link.gather(mine.s);
            for (int k = 1; k < (int)all.size(); k++) {
                BlobIn in2(all[(size_t)k]);
                std::vector<uint64_t> ru;
                in2.vec(ru);
                add_rows(down_t, ru, qbc, qnb);
            }
        }
    }
    { const double c0 = now_s(); if (!o.only_qc && !mapped_out) out.close(); t_dclose = now_s() - c0; }

    // ---- statistics, stderr, report: :3146-3235, :3240-3279, :3285-3328 ----
    uint64_t nw = 0; int32_t bc = 0; uint32_t nbins = 0;
    L.counters_len(ctx, &nw, &bc, &nbins);
    std::vector<uint64_t> t(nw, 0), part(nw);
    // One process per GPU: the job's tallies = the sum over the ranks (src/TGSFilter.cpp:3208-3213 across GPUs).  With a
    // GPU per rank: this rank's contexts folded into one vector in HBM, then ONE all-reduce of it over RCCL / xGMI
    // (include/tgsf_rccl.h; every rank has entered tgsf_create with the same table rows, see max_read_len above).
    const double t_x0 = now_s();
    double t_rccl_wait = 0, t_allreduce = 0;
    int rccl_ranks = 0;
    std::vector<tgsf_ctx*> sum_ctxs = ctxs;
    if (sharded && use_rccl) {
        // The communicator has had the whole run to come up.  One that is still not there some time after the filtering is
        // over (a peer that cannot be reached, a fabric that does not answer) must not hold the job for ever: this rank says
        // so below, every rank then sums over the sockets, and the helper is left where it waits (the process leaves with _exit).
        double rccl_patience = 120.0;
        if (const char* e = knob("TGSF_RCCL_INIT_TIMEOUT_S")) rccl_patience = atof(e);          // test knob
        while (!rccl_state->done.load(std::memory_order_acquire) && now_s() - t_x0 < rccl_patience) usleep(2000);
        if (rccl_state->done.load(std::memory_order_acquire)) {
            rccl_up.join();
            rccl_rc = rccl_state->rc; rccl_err = rccl_state->err; rccl_comm = rccl_state->comm;
        } else {
            rccl_up.detach();
            rccl_rc = TGSF_E_HIP;
            rccl_err = "the communicator was not up " + std::to_string((int)rccl_patience) + " s after the filtering ended";
        }
        t_rccl_wait = now_s() - t_x0;
        // the communicator came up on every rank, or nobody uses it: the sockets carry the rows in use instead (the run's
        // results do not depend on which way the tallies travel)
        if (link.max_u64(rccl_rc != TGSF_OK ? 1 : 0) != 0) {
            const char* ex = getenv("TGSF_SHARD_EXCHANGE");
            if (ex && !strcmp(ex, "rccl")) die("RCCL communicator: " + (rccl_err.empty() ? std::string("it failed on another rank") : rccl_err));
            if (rccl_rc != TGSF_OK) std::cerr << "Warning: rank " << link.rank << ": RCCL communicator: " << rccl_err << " -- the tallies are summed over the ranks' sockets" << std::endl;
            // (a communicator that did come up here is left as it is: taking it down may wait for peers that are stuck)
            rccl_comm = nullptr;
            use_rccl = false;
        }
    }
    if (sharded && use_rccl) {
        for (size_t k = 1; k < ctxs.size(); k++)
            if (L.counters_merge(ctxs[0], ctxs[k]) != TGSF_OK) die(L.last_error(ctxs[0]));
        const double a0 = now_s();
        if (R->allreduce_counters(ctxs[0], rccl_comm, link.rank, link.world, 0, nullptr) != TGSF_OK) die(std::string("tally all-reduce: ") + R->last_error());
        t_allreduce = now_s() - a0;
        (void)R->comm_count(rccl_comm, &rccl_ranks);
        sum_ctxs.assign(1, ctxs[0]);                                   // (it holds the whole job's totals now, on every rank)
    }
    for (tgsf_ctx* c : sum_ctxs) {                                     // sums; the four "rows used" words are maxima
        uint64_t used[2] = {0, 0};                                     // of the bin tables only the rows in use travel
        if (L.counters_used(c, part.data(), nw, used) != TGSF_OK) die(L.last_error(c));
        uint64_t rows[4];
        for (int k = 0; k < 4; k++) rows[k] = std::max(t[TGSF_CTR_ROWS + k], part[TGSF_CTR_ROWS + k]);
        const size_t head = tgsf_ctr_bin_table(0, bc, nbins);
        for (size_t i = 0; i < head; i++) t[i] += part[i];
        for (int b = 0; b < 4; b++) {
            const size_t at = tgsf_ctr_bin_table(b, bc, nbins), n = (size_t)used[b >> 1] * 5;
            for (size_t i = 0; i < n; i++) t[at + i] += part[at + i];
        }
        for (int k = 0; k < 4; k++) t[TGSF_CTR_ROWS + k] = rows[k];
    }
    // Rank 0 of a sharded job receives every rank's read lengths (the statistics need them sorted: N50 and the like) and
    // -- when the tallies were not summed on the devices -- its tally rows in use; it alone prints the run's statistics
    // and writes the report.
    const bool reports = !sharded || link.rank == 0;
    uint64_t job_reads = raw_lens.size();
    if (sharded) {
        BlobOut mine;
        mine.pod(raw_bases); mine.pod(clean_bases);
        mine.vec(raw_lens); mine.vec(clean_lens);                      // (each sorted already, beside the pipeline)
        std::vector<uint64_t> rows_used;
        if (!use_rccl) rows_used = pack_rows(t, bc, nbins);
        mine.vec(rows_used);
        const std::vector<std::string> all = link.gather(mine.s);
        for (int k = 1; k < (int)all.size(); k++) {                    // (rank 0 only)
            BlobIn in2(all[(size_t)k]);
            uint64_t rb = 0, cb = 0;
            std::vector<int> rl, cl;
            std::vector<uint64_t> ru;
            in2.pod(rb); in2.pod(cb); in2.vec(rl); in2.vec(cl); in2.vec(ru);
            raw_bases += rb; clean_bases += cb;
            const size_t r0 = raw_lens.size(), c0 = clean_lens.size();
            raw_lens.insert(raw_lens.end(), rl.begin(), rl.end());
            clean_lens.insert(clean_lens.end(), cl.begin(), cl.end());
            std::inplace_merge(raw_lens.begin(), raw_lens.begin() + (long)r0, raw_lens.end());
            if (!o.downsample) std::inplace_merge(clean_lens.begin(), clean_lens.begin() + (long)c0, clean_lens.end());   // (a downsampling run reports the selected reads' lengths instead)
            if (!use_rccl) add_rows(t, ru, bc, nbins);
        }
        job_reads = raw_lens.size();
        if (timing)
            fprintf(stderr, "SHARD %d/%d: bytes [%zu, %zu) of the text on device %d -> %s | tallies: %s (communicator ready after %.3f s of waiting, all-reduce %.4f s, %d ranks in it), exchange + gather %.3f s\n",
                    link.rank, link.world, text_off, text_off + text_size, o.device, out_path.c_str(),
                    use_rccl ? "RCCL all-reduce on the devices" : "summed on rank 0 over the ranks' sockets", t_rccl_wait, t_allreduce, rccl_ranks, now_s() - t_x0);
        if (use_rccl) R->comm_destroy(rccl_comm);
    }
    (void)job_reads;
    auto tables = [&](const std::vector<uint64_t>& v, bool clean) {
        SideTables s;
        s.bin_qual = &v[tgsf_ctr_bin_table(clean ? TGSF_B_CLEAN_QUAL : TGSF_B_RAW_QUAL, bc, nbins)];
        s.bin_cnt = &v[tgsf_ctr_bin_table(clean ? TGSF_B_CLEAN_CNT : TGSF_B_RAW_CNT, bc, nbins)];
        s.bin_rows = v[TGSF_CTR_ROWS + (clean ? 1 : 0)];
        s.q5 = &v[tgsf_ctr_end_table(clean ? TGSF_T_CLEAN5P_QUAL : TGSF_T_RAW5P_QUAL, bc)];
        s.c5 = &v[tgsf_ctr_end_table(clean ? TGSF_T_CLEAN5P_CNT : TGSF_T_RAW5P_CNT, bc)];
        s.q3 = &v[tgsf_ctr_end_table(clean ? TGSF_T_CLEAN3P_QUAL : TGSF_T_RAW3P_QUAL, bc)];
        s.c3 = &v[tgsf_ctr_end_table(clean ? TGSF_T_CLEAN3P_CNT : TGSF_T_RAW3P_CNT, bc)];
        s.end_rows = v[TGSF_CTR_ROWS + (clean ? 3 : 2)];
        s.diff_qual = &v[clean ? TGSF_CTR_CLEAN_DIFFQ : TGSF_CTR_RAW_DIFFQ];
        return s;
    };
    SideStats raw, clean;
    const int clean_num = (int)clean_lens.size();
    if (run_filter_pass && reports) {
        if (raw_lens.empty()) die("no reads in the input");
        std::sort(raw_lens.begin(), raw_lens.end());
        side_stats(bc, raw_lens, raw_bases, tables(t, false), raw);
        if (!o.only_qc && !o.downsample) {
            if (clean_lens.empty()) die("no reads passed the filters");  // the reference dereferences an empty vector here (:3183)
            std::sort(clean_lens.begin(), clean_lens.end());
            side_stats(bc, clean_lens, clean_bases, tables(t, true), clean);
        }
        const uint64_t* d = &t[TGSF_CTR_DROPINFO];
        std::cerr << "INFO: " << raw_lens.size() << " reads with a total of " << raw_bases << " bases were input." << std::endl;
        if (!o.only_qc) {
            std::cerr << "INFO: " << d[0] << " reads were discarded with " << d[1] << " bases due to low quality." << std::endl;
            std::cerr << "INFO: " << d[2] << " reads have adapter at 5', 3' and middle." << std::endl;
            std::cerr << "INFO: " << d[3] << " reads have adapter at 5' and middle." << std::endl;
            std::cerr << "INFO: " << d[4] << " reads have adapter at 3' and middle." << std::endl;
            std::cerr << "INFO: " << d[5] << " reads have adapter at 5' and 3' end." << std::endl;
            std::cerr << "INFO: " << d[6] << " reads only have adapter at middle." << std::endl;
            std::cerr << "INFO: " << d[7] << " reads only have adapter at 5' end." << std::endl;
            std::cerr << "INFO: " << d[8] << " reads only have adapter at 3' end." << std::endl;
            std::cerr << "INFO: " << d[9] << " reads didn't have any adapter." << std::endl;
            std::cerr << "INFO: " << d[10] << " bases were trimmed due to the adapter or base content bias." << std::endl;
            std::cerr << "INFO: " << d[11] << " reads were discarded with " << d[12] << " bases due to the short length." << std::endl;
            std::cerr << "INFO: " << d[13] << " reads were discarded with " << d[14] << " bases due to low quality after split." << std::endl;
            if (o.min_repeat > 0)
                std::cerr << "INFO: " << d[15] << " reads were discarded with " << d[16] << " bases due to short repeat length." << std::endl;
            std::cerr << "INFO: " << clean_num << " reads with a total of " << clean_bases << " bases after filtering." << std::endl;
            if (!o.downsample && !o.out_file.empty()) {
                if (!sharded) std::cerr << "INFO: Filtered reads were written to: " << o.out_file << "." << std::endl;
                else std::cerr << "INFO: Filtered reads were written to: " << o.out_file << ".part0 ... " << o.out_file << ".part" << link.world - 1
                               << " (" << link.world << " parts; concatenated in this order they are the reads in input order)." << std::endl;
            }
        }
    }
    // (parts of an earlier job with MORE ranks beside this job's would end up in a `cat <out>.part*`: they go, with a word)
    if (reports && sharded && !o.out_file.empty() && !o.only_qc && !o.only_adapters)
        for (int k = link.world;; k++) {
            const std::string stale = o.out_file + ".part" + std::to_string(k);
            struct stat st;
            if (lstat(stale.c_str(), &st) != 0 || !S_ISREG(st.st_mode)) break;
            if (unlink(stale.c_str()) == 0) std::cerr << "Warning: " << stale << ", a part of an earlier job with more ranks, was removed." << std::endl;
        }
    if (o.downsample && reports) {                                     // :3240-3279
        if (down_lens.empty()) die("no reads to downsample");
        std::sort(down_lens.begin(), down_lens.end());
        side_stats(bc, down_lens, down_bases, tables(down_t, false), clean);
        clean.tab[8] = limit_decimals(std::round(clean.mean_qual * 1000) / 1000.0, 2);      // two places here, :3268
        if (!o.filter)
            std::cerr << "INFO: " << (sharded ? down_job_recs : clean_recs.size()) << " reads with a total of " << (sharded ? down_job_bases : clean_bases) << " bases were input." << std::endl;
        std::cerr << "INFO: " << down_lens.size() << " reads with a total of " << down_bases << " bases after downsampling." << std::endl;
        if (!o.out_file.empty()) {
            if (!sharded) std::cerr << "INFO: Downsampled reads were written to: " << o.out_file << "." << std::endl;
            else std::cerr << "INFO: Downsampled reads were written to: " << o.out_file << ".part0 ... " << o.out_file << ".part" << link.world - 1
                           << " (" << link.world << " parts; concatenated in this order they are the reads in input order)." << std::endl;
        }
    }
    std::string qc = fasta_in ? "0" : "1";                             // :3286-3291
    qc += o.only_qc ? "0" : ((!o.filter && o.downsample) ? "1" : "2"); // :3293-3299
    if (reports) {
        std::ofstream ofs(html);
        write_report(ofs, qc, raw, clean);
        ofs.close();
        std::cerr << "INFO: Quality control report was written to: " << html << "." << std::endl;
    }
    if (timing) {
        fprintf(stderr, "POOL: %zu jobs, busy %.3f, freeing job state %.3f, longest job %.3f, first job at %.3f, last job done at %.3f (pipeline start = 0, planner done at %.3f)\n",
                pool.jobs_, t_busy, pool.destroy_, pool.longest_, pool.first_ - t_p0, pool.last_ - t_p0, t_f0 - t_p0);
        fprintf(stderr, "TIMING: total %.3f s | index+prepass %.3f | waiting for the library %.3f (load %.3f + device %.3f, beside the pre-pass) | "
                        "pipeline %.3f (batching %.3f, tgsf_submit summed over %zu feeders %.3f, plan+write %.3f, planner waiting %.3f, "
                        "first batch filtered after %.3f, fill tail %.3f, closing the output %.3f; stages overlap) | stats+report %.3f | %s (fallocate %.3f, mapping the reserved pages %.3f, fill threads busy %.3f summed)\n",
                now_s() - t_start, t_prepass, t_libwait, t_load, t_dev, t_pipe, t_parse, ctxs.size(), t_gpu, t_write, t_widle, t_first,
                t_fill_tail, t_close, now_s() - t_p0 - t_pipe, mapped_out ? "output: fallocate + mapped fill" : "output: writev", sink.t_falloc, reserver.t_populate_wait, t_busy);
        fprintf(stderr, "RESERVE: planner waited %.3f s for pages of the output file\n", t_drain);
    }
    if (timing) {
        // kernel time of the run: the stage durations of every batch (HIP events inside the library), summed over the
        // contexts -- batches of different contexts overlap, so this is an upper bound of the time the GPU was busy
        float tot[TGSF_N_STAGES] = {0};
        uint32_t nb = 0;
        for (tgsf_ctx* c : ctxs) {
            float ms[TGSF_N_STAGES]; uint32_t n1 = 0;
            if (c && L.stage_times(c, ms, &n1) == TGSF_OK) { nb += n1; for (int i = 0; i < TGSF_N_STAGES; i++) tot[i] += ms[i]; }
        }
        double sum = 0;
        for (int i = 0; i < TGSF_N_STAGES; i++) sum += tot[i];
        // (ONE write for the line: the ranks of a sharded job share this stderr, and a line put together from several writes gets
        // another rank's lines into its middle)
        char piece[256];
        snprintf(piece, sizeof piece, "GPU: kernels %.3f s summed over %zu contexts and %u batches (upper bound of the busy time: contexts overlap) = %.4f of the run |", sum * 1e-3, ctxs.size(), nb,
                 sum * 1e-3 / std::max(1e-9, now_s() - t_start));
        std::string line = piece;
        for (int i = 0; i < TGSF_N_STAGES; i++) if (tot[i] > 0) { snprintf(piece, sizeof piece, " %s %.1f ms", L.stage_name(i), tot[i]); line += piece; }
        line += "\n";
        fputs(line.c_str(), stderr);
    }
    if (timing) {
        // per device: what its feeders moved (text in, records out: H2D + kernels + D2H inside tgsf_submit)
        for (size_t di = 0; di < o.devices.size(); di++) {
            if (std::find(o.devices.begin(), o.devices.begin() + (long)di, o.devices[di]) != o.devices.begin() + (long)di) continue;   // (listed twice)
            double sub = 0; uint64_t by = 0, nb2 = 0; int nf2 = 0, node = -1;
            for (size_t k = 0; k < ctx_dev.size(); k++)
                if (ctx_dev[k] == o.devices[di]) { sub += dev_submit_s[k]; by += dev_bytes[k]; nb2 += dev_batches[k]; nf2++; node = std::max(node, dev_node[k]); }
            fprintf(stderr, "DEVICE %d: %llu batches, %.2f GB of text through %d feeders, tgsf_submit %.3f s summed = %.1f GB/s per feeder, %.1f GB/s for the device over the pipeline's %.3f s%s\n",
                    o.devices[di], (unsigned long long)nb2, by * 1e-9, nf2, sub, sub > 0 ? by * 1e-9 / sub : 0.0, t_pipe > 0 ? by * 1e-9 / t_pipe : 0.0, t_pipe,
                    node >= 0 ? (" (feeders bound to NUMA node " + std::to_string(node) + ")").c_str() : "");
        }
    }
    if (timing) {
        // CPU seconds by stage (cputime.h): what the threads that have ended charged, the main thread's share up to here, and
        // what the process used beyond both (the runtime's own threads).  ONE write (ranks share this stderr).
        const double proc = process_cpu_s();
        double own = 0;
        std::string line;
        char piece[160];
        for (int s = 0; s < CPU_N; s++) {
            double v = (double)cpu_ns()[s].load() * 1e-9;
            if (s == CPU_MAIN) v += (double)thread_cpu_ns() * 1e-9 - (double)cpu_ns()[CPU_PREPASS].load() * 1e-9;      // (the main thread is still running)
            own += v;
            if (v >= 0.0005) { snprintf(piece, sizeof piece, "%s %s %.3f", line.empty() ? "" : ",", cpu_stage_name(s), v); line += piece; }
        }
        snprintf(piece, sizeof piece, "CPU: %.3f s of CPU time (user + system) for %.3f Gbases = %.4f CPU-s per Gbase |", proc, (double)raw_bases * 1e-9,
                 raw_bases ? proc / ((double)raw_bases * 1e-9) : 0.0);
        std::string head = piece;
        snprintf(piece, sizeof piece, " | threads of the runtime and others %.3f\n", proc - own);
        fputs((head + line + piece).c_str(), stderr);
    }
    if (timing && o.downsample)
        fprintf(stderr, "DOWN: selection %.3f | QC pass over the kept reads (%s, %.2f GB of them in %.2f GB of text): context %.3f, batches + submits %.3f (its own thread; %d submits %.3f, the first %.3f) | writing them %.3f (%s) | closing the output %.3f\n",
                t_dsel, d_in_place ? "read in place" : "packed", d_kept * 1e-9, d_span * 1e-9, t_dcreate, t_dqc, n_dsubmit, t_dsubmit, t_dfirst, t_dwrite, d_mapped ? "fallocate + threads into a mapping" : "writev", t_dclose);
    // everything is written and closed: skip the teardown of multi-GB mappings and of the HIP runtime
    if (timing) {
        struct timespec ts; clock_gettime(CLOCK_REALTIME, &ts);
        fprintf(stderr, "CLOCK: main entered at %.6f, leaving at %.6f (epoch seconds)\n", t_epoch0, (double)ts.tv_sec + ts.tv_nsec * 1e-9);
    }
    leave(0);
}
