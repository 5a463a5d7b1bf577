#include "fastx.h"
#include "cputime.h"

#include "fatal.h"

#include "bam.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <cstring>
#include <algorithm>
#include <cstdlib>
#include <iostream>
#include <thread>

namespace host {

static bool ends_with(const std::string& v, const std::string& e)
{
    return e.size() <= v.size() && std::equal(e.rbegin(), e.rend(), v.rbegin());
}

InputBytes::~InputBytes()
{
    if (map_) munmap(map_, map_len_);
}

bool InputBytes::open(const std::string& path, bool sam_or_bam)
{
    if (sam_or_bam) {
        InputBytes raw;
        if (!raw.open_plain(path)) { std::cerr << "Error: Failed to open file: " << path << std::endl; return false; }
        std::string err;
        if (!decode_sam_or_bam(raw.data(), raw.size(), owned_, err)) {
            std::cerr << "Error: " << err << " (" << path << ")" << std::endl;
            fflush(nullptr); _exit(255);      // other threads are running: no static destructors, the reference's status
        }
        data_ = owned_.data(); size_ = owned_.size();
        return true;
    }
    if (ends_with(path, ".gz")) {                       // multi-member gzip is fine (:632-639); bgzip'ed files inflate in parallel
        InputBytes raw;
        if (!raw.open_plain(path)) { std::cerr << "Failed to open file: " << path << std::endl; return false; }
        std::string err;
        if (!inflate_gzip(raw.data(), raw.size(), owned_, err)) {
            std::cerr << "Error: Error encountered while decompressing file: " << path << std::endl;
            fflush(nullptr); _exit(255);      // other threads are running: no static destructors, the reference's status
        }
        data_ = owned_.data(); size_ = owned_.size();
        return true;
    }
    if (!open_plain(path)) { std::cerr << "Failed to open file: " << path << std::endl; return false; }
    return true;
}

bool InputBytes::open_raw(const std::string& path)
{
    if (!open_plain(path)) { std::cerr << "Failed to open file: " << path << std::endl; return false; }
    return true;
}

bool InputBytes::open_plain(const std::string& path)
{
    const int fd = ::open(path.c_str(), O_RDONLY);
    if (fd < 0) return false;
    struct stat st;
    if (fstat(fd, &st) != 0) { close(fd); return false; }
    size_ = (size_t)st.st_size;
    if (size_ == 0) { close(fd); data_ = ""; return true; }
    void* m = mmap(nullptr, size_, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (m == MAP_FAILED) return false;
    madvise(m, size_, MADV_SEQUENTIAL);
    map_ = m; map_len_ = size_;
    data_ = static_cast<const char*>(m);
    return true;
}

int FastxReader::scan_threads_from_env(int dflt)
{
    const char* e = knob("TGSF_SCAN_THREADS");
    const int v = e ? atoi(e) : 0;
    return v > 0 ? std::min(v, 64) : dflt;
}

static size_t scan_block_bytes()
{
    static const size_t kBlock = [] {                      // TGSF_SCAN_BLOCK: test knob (block edges on small inputs)
        const char* e = knob("TGSF_SCAN_BLOCK");
        const size_t v = e ? (size_t)strtoull(e, nullptr, 10) : 0;
        return v ? v : (size_t)(16u << 20);
    }();
    return kBlock;
}

LineScanner::LineScanner(const char* begin, const char* end, int threads, size_t block_bytes)
    : begin_(begin), end_(end), block_(block_bytes)
{
    n_blocks_ = ((size_t)(end - begin) + block_ - 1) / block_;
    ring_ = (size_t)threads * 3;
    slot_.resize(ring_);
    ready_.assign(ring_, 0);
    for (int t = 0; t < threads; t++) th_.emplace_back([this] { work(); });
}

LineScanner::~LineScanner()
{
    { std::lock_guard<std::mutex> l(m_); stop_ = true; }
    cv_.notify_all();
    for (std::thread& t : th_) t.join();
}

void LineScanner::work()
{
    CpuScope cpu(CPU_INDEX_LINES);
    std::vector<const char*> found;
    for (;;) {
        size_t b;
        {
            std::unique_lock<std::mutex> l(m_);
            cv_.wait(l, [&] { return stop_ || next_claim_ >= n_blocks_ || next_claim_ < low_ + ring_; });
            if (stop_ || next_claim_ >= n_blocks_) return;
            b = next_claim_++;
        }
        found.clear();
        const char* a = begin_ + b * block_;
        const char* e = block_end(b);
        while (a < e) {
            const char* q = static_cast<const char*>(memchr(a, '\n', (size_t)(e - a)));
            if (!q) break;
            found.push_back(q);
            a = q + 1;
        }
        {
            std::lock_guard<std::mutex> l(m_);
            slot_[b % ring_].swap(found);
            ready_[b % ring_] = b + 1;
        }
        cv_.notify_all();
    }
}

const std::vector<const char*>& LineScanner::lines(size_t b)
{
    std::unique_lock<std::mutex> l(m_);
    if (b > low_) { low_ = b; cv_.notify_all(); }
    cv_.wait(l, [&] { return ready_[b % ring_] == b + 1; });
    return slot_[b % ring_];
}

FastxReader::FastxReader(const char* data, size_t size, bool fastq, int scan_threads, std::string* message)
    : p_(data), end_(data + size), scan_threads_(scan_threads_from_env(scan_threads)), fastq_(fastq), message_(message)
{
    if (scan_threads_ > 1 && size > 0) scan_.reset(new LineScanner(data, data + size, scan_threads_, scan_block_bytes()));
}

void FastxReader::say(const std::string& text, std::string_view name)
{
    if (message_) { *message_ = text; message_->append(name); }
    else std::cerr << text << name << std::endl;
}

const char* FastxReader::next_newline(const char* from)
{
    if (!scan_) return static_cast<const char*>(memchr(from, '\n', (size_t)(end_ - from)));
    for (;;) {
        if (from >= end_) return nullptr;
        const size_t b = scan_->block_of(from);
        if (b != block_) { nl_ = &scan_->lines(b); block_ = b; nl_at_ = 0; }
        while (nl_at_ < nl_->size() && (*nl_)[nl_at_] < from) nl_at_++;
        if (nl_at_ < nl_->size()) return (*nl_)[nl_at_];
        from = scan_->block_end(b);        // none left in this block: the line runs on into the next one
    }
}

std::string_view FastxReader::line()
{
    if (p_ >= end_) { done_ = true; hit_end_ = true; return {}; }
    const char* nl = next_newline(p_);
    if (!nl) hit_end_ = true;
    const char* e = nl ? nl : end_;          // a last line without '\n' is still a line here (the reference reads out of bounds)
    size_t n = (size_t)(e - p_);
    if (n > 0 && e[-1] == '\r') n--;         // :666-668
    std::string_view v(p_, n);
    p_ = nl ? nl + 1 : end_;
    return v;
}

bool FastxReader::next(Record& r) { return fastq_ ? next_fastq(r) : next_fasta(r); }

bool FastxReader::next_partial(Record& r, bool final, bool& incomplete)
{
    const char* save = p_;
    const bool save_done = done_;
    hit_end_ = false;
    std::string msg;
    std::string* sink = message_;
    message_ = &msg;
    const bool ok = next(r);
    message_ = sink;
    if (!final && hit_end_) { p_ = save; done_ = save_done; incomplete = true; return false; }
    incomplete = false;
    if (!ok && !msg.empty()) { if (message_) *message_ = msg; else std::cerr << msg << std::endl; }
    return ok;
}

bool FastxReader::next_fastq(Record& r)
{
    std::string_view name, seq, strand;
    for (int i = 0; i < 5; i++) {                                     // :689-698
        name = line();
        if (!name.empty() && name[0] == '@') {
            seq = line();
            strand = line();
            if (!strand.empty() && strand[0] == '+' && !seq.empty()) break;
        }
    }
    if (done_) return false;                                          // :700-702
    if (name.empty()) { say("Error: input format wrong!", {}); return false; }
    name.remove_prefix(1);                                            // :709
    std::string_view qual = line();
    if (qual.empty()) { say("Error: quality are empty:", name); return false; }
    if (qual.size() != seq.size()) { say("warning: sequence and quality have different length:", name); return false; }
    r.name = name; r.seq = seq; r.qual = qual;
    return true;
}

bool FastxReader::next_fasta(Record& r)
{
    std::string_view name;
    for (int i = 0; i < 3; i++) {                                     // :732-737
        name = line();
        if (!name.empty() && name[0] == '>') break;
    }
    if (done_) return false;
    if (name.empty()) { say("Error: input format wrong!", {}); return false; }
    name.remove_prefix(1);
    std::string_view seq = line();
    if (seq.empty()) { say("Error: sequence are empty:", name); return false; }
    r.name = name; r.seq = seq; r.qual = {};
    return true;
}

// ---------------------------------------------------------------------------
RecordIndex::RecordIndex(const char* data, size_t size, bool fastq, int threads, size_t lead) : lead_(lead)
{
    chunks_.resize(1u << 17);                    // x 32768 records: room for 4e9
    th_ = std::thread([=] { produce(data, size, fastq, threads); });
}

RecordIndex::~RecordIndex()
{
    { std::lock_guard<std::mutex> l(m_); cancel_ = true; }
    cv_.notify_all();
    if (th_.joinable()) th_.join();
}

void RecordIndex::produce(const char* data, size_t size, bool fastq, int threads)
{
    CpuScope cpu(CPU_INDEX_RECORDS);
    std::string msg;
    {
        FastxReader rd(data, size, fastq, threads, &msg);
        Record r;
        size_t n = 0, published = 0;
        uint32_t longest = 0;
        while (rd.next(r)) {
            if (n / kChunk >= chunks_.size()) {   // 4.3e9 records: never silently a shorter input
                std::cerr << "Error: more than " << chunks_.size() * kChunk << " records in the input: beyond what this build indexes" << std::endl;
                quit(255);
            }
            if (n % kChunk == 0) chunks_[n / kChunk].reset(new Rec[kChunk]);
            Rec& x = chunks_[n / kChunk][n % kChunk];
            x.name = r.name.data(); x.name_len = (uint32_t)r.name.size();
            x.seq = r.seq.data(); x.len = (uint32_t)r.seq.size();
            x.qual = fastq ? r.qual.data() : r.seq.data();
            if (x.len > longest) longest = x.len;
            n++;
            if (n - published >= 256) {           // publish in small groups: one wake-up per group
                longest_.store(longest);
                { std::lock_guard<std::mutex> l(m_); count_.store(n, std::memory_order_release); }
                cv_.notify_all();
                published = n;
                if (lead_) {                      // a bounded lead: wait for the reader (or for the end of the index)
                    std::unique_lock<std::mutex> l(m_);
                    cv_.wait(l, [&] { return cancel_ || n <= wanted_ + lead_; });
                    if (cancel_) break;
                }
            }
        }
        longest_.store(longest);
        std::lock_guard<std::mutex> l(m_);
        count_.store(n, std::memory_order_release);
        message_ = msg;
        done_ = true;
    }
    cv_.notify_all();
}

bool RecordIndex::get(size_t i, Rec& r)
{
    if (lead_) {
        std::lock_guard<std::mutex> l(m_);
        if (i > wanted_) { wanted_ = i; cv_.notify_all(); }
    }
    if (i >= count_.load(std::memory_order_acquire)) {
        std::unique_lock<std::mutex> l(m_);
        cv_.wait(l, [&] { return done_ || i < count_.load(std::memory_order_acquire); });
        if (i >= count_.load(std::memory_order_acquire)) return false;
    }
    r = chunks_[i / kChunk][i % kChunk];
    return true;
}

bool RecordIndex::complete()
{
    std::lock_guard<std::mutex> l(m_);
    return done_;
}

size_t RecordIndex::wait_complete()
{
    std::unique_lock<std::mutex> l(m_);
    cv_.wait(l, [&] { return done_; });
    return count_.load();
}

bool RecordIndex::Cursor::next(Rec& r)
{
    if (ix_.get(i_, r)) { i_++; return true; }
    if (!ix_.end_message().empty()) std::cerr << ix_.end_message() << std::endl;
    return false;
}

}  // namespace host
