// pipeline.h -- the moving parts of the command line's batch pipeline (main.cpp wires them together):
//   Batch / BatchStore   a slice of the input text + the index of its records + its results, recycled
//   Channel              small bounded queue between the stages
//   Output               single-stream writer: records gathered with writev straight from the input text, or per-record
//                        gzip members (the reference's writer, src/TGSFilter.cpp:2095-2145, :786-812)
//   MappedSink + Pool    plain output into a regular file from several threads (phased fallocate + mapped fill)
#pragma once
#include <dlfcn.h>
#include <fcntl.h>
#include <malloc.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/uio.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <deque>
#include <fstream>
#include <functional>
#include <iostream>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <unordered_map>
#include <unordered_set>

#include "cputime.h"
#include "fatal.h"
#include "fastx.h"
#include "options.h"
#include "tgsf.h"
#include "textsource.h"

namespace host {

// A batch is a slice [base, base+span) of the input text plus an index of its records: nothing is
// copied on the host; the library reads sequence and quality lines in place (tgsf_batch_in.qual_offsets).
struct Batch {
    const char* base = nullptr;              // first byte of the slice (inside the mmap'ed input)
    uint64_t span = 0;                       // bytes of the slice
    std::vector<uint64_t> off, qoff;         // sequence / quality line of each read, relative to base
    std::vector<uint32_t> len;
    std::vector<Rec> recs;                   // header / sequence / quality lines, in the input
    std::vector<tgsf_read_result> res;
    std::vector<tgsf_fragment> frags;
    uint32_t n_frags = 0;
    uint64_t bases = 0;
    uint64_t id = 0;                         // position in the input: the writer puts batches back in order
    struct Emit { uint32_t read, frag; int pass_num; uint64_t at; };   // a record to write, `at` bytes into the batch's output
    std::vector<Emit> em;
    std::shared_ptr<Chunk> hold;             // streamed input: the chunk of text this batch's records live in
    char* dst = nullptr;                     // where the batch's output starts in the mapped file
    uint64_t out_bytes = 0;
    std::atomic<int> left{0};                // fill jobs still running
    void reset() {
        base = nullptr; span = 0; off.clear(); qoff.clear(); len.clear(); recs.clear(); n_frags = 0; bases = 0; id = 0;
        em.clear(); hold.reset(); dst = nullptr; out_bytes = 0; left = 0;
    }
};

// Batches are recycled, never freed while the pipeline runs: their vectors are MBs each, which malloc takes from and
// gives back to the kernel with mmap/munmap -- and those need the address-space lock for writing, which the page
// faults of the fill threads hold for reading all the time (measured: 170 ms per freed batch, 12 s over a run).
class BatchStore {
public:
    std::shared_ptr<Batch> get() {
        {
            std::lock_guard<std::mutex> l(m_);
            if (!free_.empty()) { std::shared_ptr<Batch> b = std::move(free_.back()); free_.pop_back(); return b; }
        }
        std::shared_ptr<Batch> b(new Batch);
        b->off.reserve(4096); b->qoff.reserve(4096); b->len.reserve(4096); b->recs.reserve(4096);
        return b;
    }
    void put(std::shared_ptr<Batch> b) {
        b->reset();
        std::lock_guard<std::mutex> l(m_);
        free_.push_back(std::move(b));
    }
private:
    std::mutex m_;
    std::vector<std::shared_ptr<Batch>> free_;
};

// a kept fragment, addressed in the input text (used when a downsampling pass follows the filter pass)
struct CleanRec {
    std::string_view name;
    int pass_num;
    const char* seq;
    const char* qual;
    uint32_t len;
};

template <class T>
class Channel {                               // small bounded queue; nullptr closes it
public:
    explicit Channel(size_t cap) : cap_(cap) {}
    void put(T v) {
        std::unique_lock<std::mutex> l(m_);
        not_full_.wait(l, [&] { return q_.size() < cap_; });
        q_.push_back(std::move(v));
        not_empty_.notify_one();
    }
    T get() {
        std::unique_lock<std::mutex> l(m_);
        not_empty_.wait(l, [&] { return !q_.empty(); });
        T v = std::move(q_.front());
        q_.pop_front();
        not_full_.notify_one();
        return v;
    }
private:
    std::mutex m_;
    std::condition_variable not_full_, not_empty_;
    std::deque<T> q_;
    size_t cap_;
};

// newSeqName, src/TGSFilter.cpp:1680-1701: ":<n>" goes before the first whitespace of the header
inline void append_name(std::string& out, std::string_view raw, int number)
{
    if (number < 2) { out.append(raw); return; }
    const std::string add = ":" + std::to_string(number);
    size_t i = 0;
    while (i < raw.size() && !std::isspace((unsigned char)raw[i])) i++;
    out.append(raw.substr(0, i));
    out += add;
    out.append(raw.substr(i));
}

inline double now_s()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// (fatal paths: fatal.h)

// libdeflate, the compressor the reference writes its .gz records with (libdeflate_gzip_compress, src/TGSFilter.cpp:
// 786-812), bound at run time when the system has it (no header needed: four entry points of its stable C API);
// without it the members come from zlib -- slower, the same decompressed bytes.
class Deflater {
public:
    static const Deflater& get() { static const Deflater d; return d; }
    bool ok() const { return alloc_ && compress_ && bound_ && free__; }
    void* alloc(int level) const { return alloc_(level); }
    size_t bound(void* c, size_t n) const { return bound_(c, n); }
    size_t compress(void* c, const void* in, size_t n, void* out, size_t room) const { return compress_(c, in, n, out, room); }
    void free_(void* c) const { free__(c); }
private:
    Deflater() {
        if (knob("TGSF_ZLIB_OUTPUT")) return;                 // test knob: the zlib path
        for (const char* name : {"libdeflate.so.0", "libdeflate.so"}) {
            void* h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (!h) continue;
            alloc_ = reinterpret_cast<void* (*)(int)>(dlsym(h, "libdeflate_alloc_compressor"));
            compress_ = reinterpret_cast<size_t (*)(void*, const void*, size_t, void*, size_t)>(dlsym(h, "libdeflate_gzip_compress"));
            bound_ = reinterpret_cast<size_t (*)(void*, size_t)>(dlsym(h, "libdeflate_gzip_compress_bound"));
            free__ = reinterpret_cast<void (*)(void*)>(dlsym(h, "libdeflate_free_compressor"));
            if (ok()) return;
        }
    }
    void* (*alloc_)(int) = nullptr;
    size_t (*compress_)(void*, const void*, size_t, void*, size_t) = nullptr;
    size_t (*bound_)(void*, size_t) = nullptr;
    void (*free__)(void*) = nullptr;
};

class Output {                                // plain or per-record gzip members (:2020-2053, :786-812)
public:
    bool open(const Options& o) {
        gz_ = o.out_gz;
        if (o.out_file.empty()) f_ = stdout;
        else f_ = fopen(o.out_file.c_str(), "wb");
        if (!f_) { std::cerr << "Error: Failed to open file: " << o.out_file << std::endl; return false; }
        if (gz_) {
            setvbuf(f_, nullptr, _IOFBF, 8 << 20);
            level_ = o.comp_level;
            gz_threads_ = std::max(1, std::min(o.n_thread, 32));
        } else {
            fflush(f_);
            fd_ = fileno(f_);
        }
        return true;
    }
    // Plain output: the pieces of a record (header, sequence, separator, qualities) are gathered with
    // writev straight from the input text -- no intermediate record string.
    void piece(const char* p, size_t n) {
        if (!n) return;
        if (gz_) { gzbuf_.append(p, n); return; }
        iov_.push_back({const_cast<char*>(p), n});
        if (iov_.size() >= 1000) flush_iov();
    }
    // small generated text (":<n>" suffixes, separators that are not in the input)
    void text(const std::string& t) {
        if (gz_) { gzbuf_ += t; return; }
        if (pool_.size() + t.size() > pool_.capacity()) flush_iov();
        const size_t o0 = pool_.size();
        pool_ += t;
        iov_.push_back({&pool_[o0], t.size()});
    }
    void end_record() {
        if (!gz_) return;
        ends_.push_back(gzbuf_.size());
        if (gzbuf_.size() >= (256u << 20)) flush_gz();
    }
    // One gzip member per record, as the reference writes them (:786-812, compressed by its worker threads):
    // the records gathered since the last flush are split into byte-balanced runs, each run is compressed
    // member by member on its own thread, and the runs are written in order.
    void flush_gz() {
        if (ends_.empty()) return;
        const size_t nrec = ends_.size();
        const int T = (int)std::min<size_t>((size_t)gz_threads_, nrec);
        std::vector<std::vector<char>> outv((size_t)T);
        std::vector<size_t> cut((size_t)T + 1, nrec);
        cut[0] = 0;
        for (int t = 1; t < T; t++) {
            const size_t target = gzbuf_.size() / (size_t)T * (size_t)t;
            cut[(size_t)t] = (size_t)(std::lower_bound(ends_.begin(), ends_.end(), target) - ends_.begin());
        }
        std::atomic<bool> bad{false};
        const Deflater& ld = Deflater::get();
        auto work = [&](int t) {
            std::vector<char>& o = outv[(size_t)t];
            if (ld.ok()) {                                      // the reference's compressor, one member per record (:786-812)
                void* cz = ld.alloc(level_);
                if (!cz) { bad = true; return; }
                for (size_t r = cut[(size_t)t]; r < cut[(size_t)t + 1]; r++) {
                    const size_t b = r ? ends_[r - 1] : 0, n = ends_[r] - b;
                    const size_t at = o.size(), room = ld.bound(cz, n);
                    o.resize(at + room);
                    const size_t got = ld.compress(cz, gzbuf_.data() + b, n, o.data() + at, room);
                    if (!got) bad = true;
                    o.resize(at + got);
                }
                ld.free_(cz);
                return;
            }
            z_stream z;
            memset(&z, 0, sizeof z);
            if (deflateInit2(&z, level_, Z_DEFLATED, 15 + 16, 8, Z_DEFAULT_STRATEGY) != Z_OK) { bad = true; return; }
            for (size_t r = cut[(size_t)t]; r < cut[(size_t)t + 1]; r++) {
                const size_t b = r ? ends_[r - 1] : 0, n = ends_[r] - b;
                deflateReset(&z);
                const size_t at = o.size(), room = deflateBound(&z, (uLong)n) + 64;
                o.resize(at + room);
                z.next_in = (Bytef*)(gzbuf_.data() + b); z.avail_in = (uInt)n;
                z.next_out = (Bytef*)(o.data() + at); z.avail_out = (uInt)room;
                if (deflate(&z, Z_FINISH) != Z_STREAM_END) bad = true;
                o.resize(at + room - z.avail_out);
            }
            deflateEnd(&z);
        };
        std::vector<std::thread> th;
        for (int t = 1; t < T; t++) th.emplace_back(work, t);
        work(0);
        for (std::thread& x : th) x.join();
        if (bad) die("compression failed");
        for (const auto& o : outv) if (!o.empty() && fwrite(o.data(), 1, o.size(), f_) != o.size()) die("write failed");
        gzbuf_.clear();
        ends_.clear();
    }
    void flush_iov() {
        if (gz_) { flush_gz(); return; }
        size_t i = 0;
        while (i < iov_.size()) {
            ssize_t w = writev(fd_, &iov_[i], (int)std::min<size_t>(iov_.size() - i, 1000));
            if (w < 0) die("write failed");
            size_t left = (size_t)w;
            while (i < iov_.size() && left >= iov_[i].iov_len) { left -= iov_[i].iov_len; i++; }
            if (left) { iov_[i].iov_base = (char*)iov_[i].iov_base + left; iov_[i].iov_len -= left; }
        }
        iov_.clear();
        pool_.clear();
    }
    void close() {
        flush_iov();
        if (f_ && f_ != stdout) fclose(f_); else if (f_) fflush(f_);
        f_ = nullptr;
    }
    Output() { pool_.reserve(1 << 16); }
private:
    FILE* f_ = nullptr;
    int fd_ = -1;
    bool gz_ = false;
    int level_ = 6, gz_threads_ = 1;
    std::string gzbuf_, pool_;              // gz: the records since the last flush, back to back
    std::vector<size_t> ends_;              // ... and where each of them ends
    std::vector<iovec> iov_;
};

// Plain output into a regular file, from several threads.  Measured on the MI355X host (tools/hostio_probe.cpp,
// tools/sink_probe*.cpp, tests/manual/e2e_knobs.py): one write() stream into a tmpfs file is a single-thread copy under
// the inode lock, ~6 GB/s, and several streams into one file serialise on that lock.  What the kernel does faster:
// fallocate() instantiates pages at 14-19 GB/s (one thread), and threads storing into a shared mapping of EXISTING,
// ALREADY MAPPED pages run at memory speed.  Page FAULTS on the file while fallocate() runs on it drag both down (~7 GB/s
// together), so the work is arranged in strides (main.cpp, the planner): fallocate a stride of the file; then map its
// pages (MADV_POPULATE_WRITE, 32 threads, 0.2 s for 21 GB in all) while no fallocate runs; from then on the fill threads'
// copies into that stride take no fault and overlap the next stride's fallocate.  Used when the output is a regular file
// that takes fallocate; everything else (pipes, /dev/null, gzip) goes through Output above.
class MappedSink {
public:
    // only_new: succeed only if the file does not exist yet (it is created now): a caller that reserves pages before it
    // knows that the run will get as far as writing uses this, so that an existing file is not touched by a run that ends
    // in its pre-pass -- and a file created here is removed again if the run ends before a record is laid out (cut_back)
    bool open(const std::string& path, uint64_t virt_bytes, bool only_new = false) {
        fd_ = ::open(path.c_str(), only_new ? (O_RDWR | O_CREAT | O_EXCL) : (O_RDWR | O_CREAT | O_TRUNC), 0644);
        if (fd_ < 0) return false;
        created_ = only_new;
        path_ = path;
        struct stat st;
        bool ok = fstat(fd_, &st) == 0 && S_ISREG(st.st_mode) && fallocate(fd_, 0, 0, 4096) == 0 && ftruncate(fd_, 0) == 0;   // not every file system has fallocate
        if (ok) {
            cap_ = (virt_bytes + 4095) & ~uint64_t(4095);
            map_ = (char*)mmap(nullptr, cap_, PROT_READ | PROT_WRITE, MAP_SHARED, fd_, 0);      // beyond the end of the file for now
            ok = map_ != MAP_FAILED;
            // Every page of this mapping is written once and never looked at again: say so.  Without the advice, dropping
            // the mappings (MADV_DONTNEED behind the fill jobs, or the exit) marks each page accessed -- 20 M pages moved
            // between the LRU lists under the lock the fallocate beside it needs for every page it adds.
            if (ok) madvise(map_, cap_, MADV_SEQUENTIAL);
        }
        if (!ok) { ::close(fd_); fd_ = -1; if (created_) unlink(path_.c_str()); return false; }
        live() = this;
        on_die().store([] { if (MappedSink* s = live()) s->cut_back(); });
        return true;
    }
    // a fatal path (any thread, or the handler of SIGINT / SIGTERM: only unlink, ftruncate and atomics here): leave the
    // records laid out so far, not the pages reserved ahead of them.  A reserve_to in flight on another thread sees dead_
    // when its fallocate returns and cuts the file back again itself.
    void cut_back() {
        if (fd_ < 0) return;
        dead_.store(true);
        const uint64_t size = size_.load();
        if (created_ && size == 0) { unlink(path_.c_str()); return; }
        if (reserved_.load() > size && ftruncate(fd_, (off_t)size) != 0) { /* nothing more to do */ }
    }
    uint64_t reserved() const { return reserved_.load(); }
    uint64_t planned() const { return size_.load(); }
    uint64_t capacity() const { return cap_; }
    // instantiate the pages up to `end`.  must: a failure ends the run (the records need the space); otherwise (the
    // speculative early reserve) it just reports false
    bool reserve_to(uint64_t end, bool must = true) {
        const uint64_t have = reserved_.load();
        if (end <= have) return true;
        if (dead_.load()) return false;
        const double t0 = now_s();
        if (fallocate(fd_, 0, (off_t)have, (off_t)(end - have)) != 0) {
            if (must) die(std::string("cannot extend the output file: ") + strerror(errno));
            return false;
        }
        t_falloc += now_s() - t0;
        reserved_.store(end);
        if (dead_.load()) { cut_back(); return false; }                 // the run ended meanwhile: not one page more than it left
        return true;
    }
    // map the (instantiated) pages of [at, at+n) into the address space now, so that storing into them later takes no
    // page fault -- page faults on this file while fallocate() runs on it slow both down to a crawl
    // (MADV_POPULATE_READ would do on tmpfs -- a shared writable mapping of a tmpfs file needs no write notification, its
    // page-table entries are writable whichever fault installs them -- but on the MI355X host it costs MORE than the write
    // fault: 37-42 CPU-s and 1.9-2.2 s of wall time for 72 GB on 32 threads against 15-16 CPU-s and 0.9-1.0 s, 11-12 against
    // 7-8 CPU-s on 4 threads; profiles/r06_cpu_populate_read.txt.  Measured in round 6, not used.)
    void populate(uint64_t at, uint64_t n) {
        char* p = map_ + (at & ~uint64_t(4095));
        const size_t len = (size_t)(((at + n + 4095) & ~uint64_t(4095)) - (at & ~uint64_t(4095)));
        if (madvise(p, len, 23 /* MADV_POPULATE_WRITE, Linux 5.14 */) != 0)
            for (size_t o = 0; o < len; o += 4096) { volatile char* q = p + o; *q = *q; }     // older kernels: one touch per page
    }
    // the next n bytes of the file (inside what was reserved)
    char* place(uint64_t n) { char* p = map_ + size_.load(); size_.fetch_add(n); return p; }
    // the pages wholly inside [p, p+n) are done with: drop their mappings now, from the calling (fill) thread, instead of
    // all of them at exit from one
    static void release(const char* p, size_t n) {
        const uintptr_t lo = ((uintptr_t)p + 4095) & ~uintptr_t(4095), hi = ((uintptr_t)p + n) & ~uintptr_t(4095);
        if (hi > lo) madvise((void*)lo, hi - lo, MADV_DONTNEED);
    }
    void close() {
        if (fd_ < 0) return;
        if (reserved_.load() != size_.load() && ftruncate(fd_, (off_t)size_.load()) != 0) die("cannot set the size of the output file");
        ::close(fd_);
        fd_ = -1;
        if (live() == this) { live() = nullptr; on_die().store(nullptr); }
    }
    bool is_open() const { return fd_ >= 0; }
    double t_falloc = 0;
private:
    static MappedSink*& live() { static MappedSink* s = nullptr; return s; }
    int fd_ = -1;
    char* map_ = nullptr;
    std::atomic<uint64_t> size_{0}, reserved_{0};
    std::atomic<bool> dead_{false};
    uint64_t cap_ = 0;
    bool created_ = false;
    std::string path_;
};

// Copy for the fill jobs: the bulk of a sequence / quality line goes out with non-temporal stores.  The destination pages
// were zeroed by fallocate a moment ago and are not read again by this program: ordinary stores would first fetch every
// line they overwrite and push the input text out of the caches, next to a fallocate that is itself memory-bound.
#if defined(__x86_64__)
#include <immintrin.h>
__attribute__((target("avx2"))) inline void stream_copy_avx2(char* d, const char* s, size_t n)
{
    const size_t head = (size_t)(-(uintptr_t)d & 31u) < n ? (size_t)(-(uintptr_t)d & 31u) : n;      // up to a 32-byte boundary
    memcpy(d, s, head);
    d += head; s += head; n -= head;
    size_t i = 0;
    for (; i + 128 <= n; i += 128) {
        const __m256i a = _mm256_loadu_si256((const __m256i*)(s + i)), b = _mm256_loadu_si256((const __m256i*)(s + i + 32));
        const __m256i c = _mm256_loadu_si256((const __m256i*)(s + i + 64)), e = _mm256_loadu_si256((const __m256i*)(s + i + 96));
        _mm256_stream_si256((__m256i*)(d + i), a); _mm256_stream_si256((__m256i*)(d + i + 32), b);
        _mm256_stream_si256((__m256i*)(d + i + 64), c); _mm256_stream_si256((__m256i*)(d + i + 96), e);
    }
    memcpy(d + i, s + i, n - i);
}
inline void stream_copy(char* d, const char* s, size_t n)
{
    static const bool avx2 = __builtin_cpu_supports("avx2");
    if (avx2 && n >= 4096) stream_copy_avx2(d, s, n); else memcpy(d, s, n);
}
inline void stream_fence() { _mm_sfence(); }
#else
inline void stream_copy(char* d, const char* s, size_t n) { memcpy(d, s, n); }
inline void stream_fence() {}
#endif

// worker threads for the fill jobs
class Pool {
public:
    explicit Pool(int n, int cpu_stage = CPU_FILL) : stage_(cpu_stage) { for (int i = 0; i < n; i++) th_.emplace_back([this] { run(); }); }
    double busy_s() { std::lock_guard<std::mutex> l(m_); return busy_; }
    void add(std::function<void()> f) {
        { std::lock_guard<std::mutex> l(m_); q_.push_back(std::move(f)); open_++; }
        cv_.notify_one();
    }
    void drain() {                              // until every job added so far has run
        std::unique_lock<std::mutex> l(m_);
        idle_.wait(l, [&] { return open_ == 0; });
    }
    void finish() {                             // runs what is queued, then stops the threads
        { std::lock_guard<std::mutex> l(m_); stop_ = true; }
        cv_.notify_all();
        for (std::thread& t : th_) t.join();
        th_.clear();
    }
private:
    void run() {
        CpuScope cpu(stage_);
        for (;;) {
            std::function<void()> f;
            {
                std::unique_lock<std::mutex> l(m_);
                cv_.wait(l, [&] { return stop_ || !q_.empty(); });
                if (q_.empty()) return;
                f = std::move(q_.front());
                q_.pop_front();
            }
            const double t0 = now_s();
            f();
            const double dt = now_s() - t0;
            f = nullptr;                            // what the job held goes now, outside the lock
            const double dd = now_s() - t0 - dt;
            std::lock_guard<std::mutex> l(m_);
            busy_ += dt;
            destroy_ += dd;
            if (dt > longest_) longest_ = dt;
            jobs_++;
            if (first_ == 0) first_ = t0;
            last_ = std::max(last_, t0 + dt + dd);
            if (--open_ == 0) idle_.notify_all();
        }
    }
    double busy_ = 0;
    const int stage_;
public:
    double destroy_ = 0, longest_ = 0, first_ = 0, last_ = 0;
    size_t jobs_ = 0;
private:
    size_t open_ = 0;
    std::condition_variable idle_;
    std::mutex m_;
    std::condition_variable cv_;
    std::deque<std::function<void()>> q_;
    std::vector<std::thread> th_;
    bool stop_ = false;
};

// The output file's pages, made ready ahead of the planner by a thread of its own.  A stride of the file is instantiated
// (fallocate: one thread, 13-16 GB/s, the critical path of a run that writes a tmpfs file) and then its pages are mapped
// (MADV_POPULATE_WRITE on the populate pool) -- either between two fallocates (page faults on a file take its inode's
// i_lock while a fallocate is in progress on it, shmem_falloc_wait: dozens of faulting threads beside a fallocate drag
// both down), or, `beside` > 0, by a FEW threads while the next stride is being instantiated.  The planner says how far
// the file will probably go (want) and waits for the part it is about to hand to the fill threads (wait_ready).
class Reserver {
public:
    Reserver(MappedSink& sink, Pool& populate, uint64_t stride, bool beside) : sink_(sink), pop_(populate), stride_(stride), beside_(beside) {}
    ~Reserver() { finish(); }
    // speculative: up to `limit` while nothing else is known (a failure just ends the speculation)
    void start(uint64_t speculative_limit) {
        goal_ = speculative_limit;
        th_ = std::thread([this] { run(); });
    }
    // the file will probably reach `estimate`; [0, need) is needed for certain (its reservation failing ends the run)
    void want(uint64_t estimate, uint64_t need) {
        std::lock_guard<std::mutex> l(m_);
        if (need > need_) need_ = need;
        goal_ = estimate > need_ ? estimate : need_;                    // (an estimate may shrink: what is reserved beyond it is cut off at the end)
        cv_.notify_all();
    }
    void wait_ready(uint64_t upto) {
        std::unique_lock<std::mutex> l(m_);
        ready_cv_.wait(l, [&] { return ready_ >= upto; });
    }
    void finish() {
        { std::lock_guard<std::mutex> l(m_); stop_ = true; }
        cv_.notify_all();
        if (th_.joinable()) th_.join();
    }
    double t_populate_wait = 0;                 // fallocate waiting for the mapping of the stride before (serial mode)
private:
    void piece_done(size_t k) {
        std::lock_guard<std::mutex> l(m_);
        done_[k] = 1;
        while (next_ < done_.size() && done_[next_]) { ready_ = ends_[next_]; next_++; }
        ready_cv_.notify_all();
    }
    void run() {
        CpuScope cpu(CPU_FALLOCATE);
        uint64_t at = 0;                        // pages of [0, at) are instantiated and their mapping is under way or done
        for (;;) {
            uint64_t goal, need;
            {
                std::unique_lock<std::mutex> l(m_);
                cv_.wait(l, [&] { return stop_ || goal_ > at; });
                if (stop_ && at >= need_) return;                    // stopped: only what is needed for certain is still reserved
                if (goal_ <= at) return;        // stopped with nothing left to do
                goal = stop_ ? need_ : goal_; need = need_;
            }
            uint64_t upto = std::min<uint64_t>(goal, at + stride_);
            if (upto < sink_.reserved()) upto = sink_.reserved();
            if (!sink_.reserve_to(upto, false)) {                    // no room for the estimate: exactly what is needed
                if (need > at) { upto = need; sink_.reserve_to(upto); }
                else { std::lock_guard<std::mutex> l(m_); goal_ = at; continue; }   // the speculation ends here
            }
            const uint64_t lo = at & ~uint64_t(4095), piece = 32u << 20;
            const double d0 = now_s();
            for (uint64_t o = lo; o < upto; o += piece) {
                const uint64_t n = std::min<uint64_t>(piece, upto - o);
                size_t k;
                { std::lock_guard<std::mutex> l(m_); k = done_.size(); done_.push_back(0); ends_.push_back(o + n); }
                pop_.add([this, o, n, k] { sink_.populate(o, n); piece_done(k); });
            }
            if (!beside_) { pop_.drain(); t_populate_wait += now_s() - d0; }
            at = upto;
        }
    }
    MappedSink& sink_;
    Pool& pop_;
    const uint64_t stride_;
    const bool beside_;
    std::mutex m_;
    std::condition_variable cv_, ready_cv_;
    uint64_t goal_ = 0, need_ = 0, ready_ = 0;
    std::vector<char> done_;
    std::vector<uint64_t> ends_;
    size_t next_ = 0;
    bool stop_ = false;
    std::thread th_;
};



}  // namespace host
