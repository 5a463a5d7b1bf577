// fastx.h -- FASTA/FASTQ record reader with the reference's record semantics
// (FastxReader, src/TGSFilter.cpp:521-782): strict 4-line FASTQ / 2-line FASTA, "\r\n" tolerated
// (:666-668), header = everything after '@'/'>' (:709, :748), up to 5 (FASTQ) / 3 (FASTA) lines are
// tried for a header (:689-698, :732-737), the stream ENDS at the first malformed record (:700-723).
// Input is a flat byte range: plain files are mmap'ed, .gz files are inflated into memory (zlib),
// SAM/BAM files are decoded into FASTQ text in memory (bam.h).
#pragma once
#include <atomic>
#include <condition_variable>
#include <cstddef>
#include <cstdint>
#include <memory>
#include <mutex>
#include <string>
#include <string_view>
#include <thread>
#include <vector>

namespace host {

class InputBytes {
public:
    ~InputBytes();
    // prints "Failed to open file: <path>" on failure (:564).  sam_or_bam: the file is SAM/BAM (by name, as the
    // reference decides) and is decoded to FASTQ text in memory (bam.h).
    bool open(const std::string& path, bool sam_or_bam = false);
    // the bytes of the file as they are (no decoding): for textsource.h, which decodes them piece by piece
    bool open_raw(const std::string& path);
    const char* data() const { return data_; }
    size_t size() const { return size_; }
    bool mapped() const { return map_ != nullptr; }   // a read-only file mapping (not decoded text in memory)
private:
    bool open_plain(const std::string& path);   // mmap, no message
    const char* data_ = nullptr;
    size_t size_ = 0;
    void* map_ = nullptr;
    size_t map_len_ = 0;
    std::vector<char> owned_;
};

struct Record { std::string_view name, seq, qual; };

// Line ends of a byte range, located AHEAD of the record assembler by a few threads: the range is cut into
// blocks, the threads claim blocks in order and stay at most `ring` blocks ahead of the consumer (finding the
// newlines is all the work of indexing long reads; at 4 lines per record the assembly itself is nothing).
class LineScanner {
public:
    LineScanner(const char* begin, const char* end, int threads, size_t block_bytes);
    ~LineScanner();
    size_t block_of(const char* p) const { return (size_t)(p - begin_) / block_; }
    const char* block_end(size_t b) const { return (size_t)(end_ - begin_) > (b + 1) * block_ ? begin_ + (b + 1) * block_ : end_; }
    // newlines of block b, ascending (blocks until scanned); blocks below b are given back to the scanners
    const std::vector<const char*>& lines(size_t b);
private:
    void work();
    const char* begin_;
    const char* end_;
    size_t block_, n_blocks_, ring_;
    std::vector<std::vector<const char*>> slot_;  // slot_[b % ring_]
    std::vector<size_t> ready_;                   // ready_[b % ring_] == b + 1 once block b is scanned
    size_t next_claim_ = 0, low_ = 0;             // first block not yet claimed / first block still held by the consumer
    bool stop_ = false;
    std::mutex m_;
    std::condition_variable cv_;
    std::vector<std::thread> th_;
};

class FastxReader {
public:
    // scan_threads > 1: line ends come from a LineScanner; the records are assembled from the lines exactly as
    // without it.  `message`: where the reference's message about a malformed record goes instead of stderr
    // (RecordIndex prints it when a pass over the records reaches that point).
    FastxReader(const char* data, size_t size, bool fastq, int scan_threads = 1, std::string* message = nullptr);
    static int scan_threads_from_env(int dflt);   // TGSF_SCAN_THREADS overrides (test knob)
    bool next(Record& r);        // false: end of input (or first malformed record, after the reference's message)
    // The same over a buffer that may end in the middle of a record (a chunk of a stream, textsource.h): when the
    // attempt ran into the end of the buffer and more text follows (!final) nothing is consumed, nothing is said,
    // `incomplete` is set and the caller retries from pos() with more text.
    bool next_partial(Record& r, bool final, bool& incomplete);
    const char* pos() const { return p_; }
private:
    std::string_view line();     // "" at end of input (then done_ is set, like getLine :676-680)
    const char* next_newline(const char* from);   // first '\n' in [from, end_), or nullptr
    bool next_fastq(Record& r);
    bool next_fasta(Record& r);
    void say(const std::string& text, std::string_view name);
    const char* p_;
    const char* end_;
    int scan_threads_ = 1;
    bool fastq_;
    bool done_ = false;
    bool hit_end_ = false;       // a line of the current attempt ended at the end of the buffer, not at a newline
    std::string* message_ = nullptr;
    std::unique_ptr<LineScanner> scan_;
    const std::vector<const char*>* nl_ = nullptr;   // newlines of the current block
    size_t nl_at_ = 0, block_ = (size_t)-1;
};

// The records of an input, indexed ONCE, in the background, for every pass over them (the reference reads its
// input twice: GetFilterParameterTask::read_fastx :949-982, then TGSFilterTask::read_fastx :1845-1870).  32 bytes
// per record stay in memory; the text itself is only referenced.
struct Rec {
    const char* name; const char* seq; const char* qual;    // qual == seq for FASTA
    uint32_t name_len, len;
};
class RecordIndex {
public:
    // lead > 0: the indexing stays at most `lead` records ahead of the furthest get() (a pass that looks at the first
    // reads only -- the pre-pass of a sharded job over the whole text -- does not index the rest), and ends when the
    // index is destroyed
    RecordIndex(const char* data, size_t size, bool fastq, int threads, size_t lead = 0);
    ~RecordIndex();
    bool get(size_t i, Rec& r);          // blocks until record i is indexed; false: the input has fewer records
    bool complete();                     // the whole input is indexed (non-blocking)
    size_t wait_complete();              // blocks; number of records
    uint32_t longest() const { return longest_.load(); }    // longest read so far
    const std::string& end_message() const { return message_; }   // valid once a get() returned false
    // one pass over the records; prints the reference's message when it ends at a malformed record
    class Cursor {
    public:
        explicit Cursor(RecordIndex& ix) : ix_(ix) {}
        bool next(Rec& r);
    private:
        RecordIndex& ix_;
        size_t i_ = 0;
    };
private:
    static constexpr size_t kChunk = 1u << 15;
    void produce(const char* data, size_t size, bool fastq, int threads);
    std::vector<std::unique_ptr<Rec[]>> chunks_;   // table preallocated: readers never see it move
    std::atomic<size_t> count_{0};
    std::atomic<uint32_t> longest_{0};
    size_t lead_ = 0, wanted_ = 0;               // (lead_ > 0) the furthest record asked for so far
    bool cancel_ = false;
    bool done_ = false;
    std::string message_;
    std::mutex m_;
    std::condition_variable cv_;
    std::thread th_;
};

}  // namespace host
