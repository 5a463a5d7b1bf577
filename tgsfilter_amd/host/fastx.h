// fastx.h -- FASTA/FASTQ record reader with the reference's record semantics
// (FastxReader, src/TGSFilter.cpp:521-782): strict 4-line FASTQ / 2-line FASTA, "\r\n" tolerated
// (:666-668), header = everything after '@'/'>' (:709, :748), up to 5 (FASTQ) / 3 (FASTA) lines are
// tried for a header (:689-698, :732-737), the stream ENDS at the first malformed record (:700-723).
// Input is a flat byte range: plain files are mmap'ed, .gz files are inflated into memory (zlib),
// SAM/BAM files are decoded into FASTQ text in memory (bam.h).
#pragma once
#include <cstddef>
#include <string>
#include <string_view>
#include <vector>

namespace host {

class InputBytes {
public:
    ~InputBytes();
    // prints "Failed to open file: <path>" on failure (:564).  sam_or_bam: the file is SAM/BAM (by name, as the
    // reference decides) and is decoded to FASTQ text in memory (bam.h).
    bool open(const std::string& path, bool sam_or_bam = false);
    const char* data() const { return data_; }
    size_t size() const { return size_; }
private:
    bool open_plain(const std::string& path);   // mmap, no message
    const char* data_ = nullptr;
    size_t size_ = 0;
    void* map_ = nullptr;
    size_t map_len_ = 0;
    std::vector<char> owned_;
};

struct Record { std::string_view name, seq, qual; };

class FastxReader {
public:
    // scan_threads > 1: line ends are located ahead of the parser, one 256-MB block of the text at a time, by
    // that many threads (finding the newlines is all the work of indexing long reads); the records are then
    // assembled from the lines exactly as without it.
    FastxReader(const char* data, size_t size, bool fastq, int scan_threads = 1)
        : p_(data), end_(data + size), scan_threads_(scan_threads_from_env(scan_threads)), fastq_(fastq) {}
    static int scan_threads_from_env(int dflt);   // TGSF_SCAN_THREADS overrides (test knob)
    bool next(Record& r);        // false: end of input (or first malformed record, after the reference's message)
private:
    std::string_view line();     // "" at end of input (then done_ is set, like getLine :676-680)
    const char* next_newline(const char* from);   // first '\n' in [from, end_), or nullptr
    void scan_block(const char* from);
    bool next_fastq(Record& r);
    bool next_fasta(Record& r);
    const char* p_;
    const char* end_;
    int scan_threads_ = 1;
    bool fastq_;
    bool done_ = false;
    std::vector<const char*> nl_;                 // newlines of the current block, ascending
    size_t nl_at_ = 0;
    const char* block_begin_ = nullptr;
    const char* block_end_ = nullptr;
};

}  // namespace host
