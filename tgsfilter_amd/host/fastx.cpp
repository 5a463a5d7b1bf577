#include "fastx.h"

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
            exit(-1);
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
            exit(-1);
        }
        data_ = owned_.data(); size_ = owned_.size();
        return true;
    }
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
    const char* e = getenv("TGSF_SCAN_THREADS");
    const int v = e ? atoi(e) : 0;
    return v > 0 ? std::min(v, 64) : dflt;
}

void FastxReader::scan_block(const char* from)
{
    static const size_t kBlock = [] {                      // TGSF_SCAN_BLOCK: test knob (block edges on small inputs)
        const char* e = getenv("TGSF_SCAN_BLOCK");
        const size_t v = e ? (size_t)strtoull(e, nullptr, 10) : 0;
        return v ? v : (size_t)(256u << 20);
    }();
    block_begin_ = from;
    block_end_ = (size_t)(end_ - from) > kBlock ? from + kBlock : end_;
    const int T = scan_threads_;
    std::vector<std::vector<const char*>> part((size_t)T);
    std::vector<std::thread> th;
    const size_t per = ((size_t)(block_end_ - block_begin_) + (size_t)T - 1) / (size_t)T;
    for (int t = 0; t < T; t++)
        th.emplace_back([&, t] {
            const char* a = block_begin_ + std::min(per * (size_t)t, (size_t)(block_end_ - block_begin_));
            const char* b = block_begin_ + std::min(per * (size_t)(t + 1), (size_t)(block_end_ - block_begin_));
            while (a < b) {
                const char* q = static_cast<const char*>(memchr(a, '\n', (size_t)(b - a)));
                if (!q) break;
                part[(size_t)t].push_back(q);
                a = q + 1;
            }
        });
    for (std::thread& x : th) x.join();
    nl_.clear();
    for (const auto& v : part) nl_.insert(nl_.end(), v.begin(), v.end());
    nl_at_ = 0;
}

const char* FastxReader::next_newline(const char* from)
{
    if (scan_threads_ <= 1) return static_cast<const char*>(memchr(from, '\n', (size_t)(end_ - from)));
    for (;;) {
        if (from >= end_) return nullptr;
        if (!block_begin_ || from < block_begin_ || from >= block_end_) scan_block(from);
        while (nl_at_ < nl_.size() && nl_[nl_at_] < from) nl_at_++;
        if (nl_at_ < nl_.size()) return nl_[nl_at_];
        from = block_end_;                 // none left in this block: the line runs on into the next one
    }
}

std::string_view FastxReader::line()
{
    if (p_ >= end_) { done_ = true; return {}; }
    const char* nl = next_newline(p_);
    const char* e = nl ? nl : end_;          // a last line without '\n' is still a line here (the reference reads out of bounds)
    size_t n = (size_t)(e - p_);
    if (n > 0 && e[-1] == '\r') n--;         // :666-668
    std::string_view v(p_, n);
    p_ = nl ? nl + 1 : end_;
    return v;
}

bool FastxReader::next(Record& r) { return fastq_ ? next_fastq(r) : next_fasta(r); }

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
    if (name.empty()) { std::cerr << "Error: input format wrong!" << std::endl; return false; }
    name.remove_prefix(1);                                            // :709
    std::string_view qual = line();
    if (qual.empty()) { std::cerr << "Error: quality are empty:" << name << std::endl; return false; }
    if (qual.size() != seq.size()) {
        std::cerr << "warning: sequence and quality have different length:" << name << std::endl;
        return false;
    }
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
    if (name.empty()) { std::cerr << "Error: input format wrong!" << std::endl; return false; }
    name.remove_prefix(1);
    std::string_view seq = line();
    if (seq.empty()) { std::cerr << "Error: sequence are empty:" << name << std::endl; return false; }
    r.name = name; r.seq = seq; r.qual = {};
    return true;
}

}  // namespace host
