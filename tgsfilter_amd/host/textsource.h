// textsource.h -- input that is not a plain FASTA/FASTQ file, in bounded memory: gzip (any series of members;
// bgzip'ed files inflate block-parallel), unaligned BAM and SAM (decoded to FASTQ text, bam.h), produced piece by
// piece instead of whole.  The reference streams too: FastxReader inflates 1-MiB chunks (src/TGSFilter.cpp:567-640),
// read_bam takes one record at a time (:1872-1916) -- and opens its input twice (pre-pass, then filter pass); so does
// this reader when the input is too large to keep.
//
//   ByteStream   decompressed bytes of the file           (PlainBytes | GzBytes)
//   TextSource   FASTA/FASTQ text                          (the bytes themselves | BamText | SamText)
//   ChunkReader  that text cut into chunks of WHOLE records, each parsed (fastx.h semantics: the stream ends at the
//                first malformed record), a bounded number of chunks alive at a time
#pragma once
#include <condition_variable>
#include <cstddef>
#include <cstdint>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "fastx.h"

namespace host {

class ByteStream {
public:
    virtual ~ByteStream() {}
    // appends up to cap bytes at dst; returns how many; eof once nothing more will come.  false + err on corrupt input.
    virtual bool read(char* dst, size_t cap, size_t& got, bool& eof, std::string& err) = 0;
    virtual double consumed() const = 0;          // share of the file's bytes taken so far, 0..1
};

class TextSource {
public:
    virtual ~TextSource() {}
    virtual bool read(char* dst, size_t cap, size_t& got, bool& eof, std::string& err) = 0;
    virtual double consumed() const = 0;
};

// Opens `data` (the mapped file) as text: gz -> inflate; sam_or_bam -> by content, like hts_open.
// nullptr + err when the format is not supported (CRAM).
std::unique_ptr<TextSource> open_text(const char* data, size_t size, bool sam_or_bam, std::string& err);

struct Chunk {
    std::unique_ptr<char[]> buf;
    size_t cap = 0, size = 0;                     // bytes of whole records
    std::vector<Rec> recs;                        // pointers into buf
    bool last = false;                            // the input ends with this chunk
    std::string message;                          // the reference's message if the stream ended at a malformed record
};

class ChunkReader {
public:
    ChunkReader(std::unique_ptr<TextSource> src, bool fastq, size_t chunk_bytes, int max_live);
    // next chunk of whole records (blocks while max_live chunks are still referenced); nullptr after the last one.
    // Exits the program with the reference's message on corrupt compressed input, as FastxReader::readChunk does (:636).
    std::shared_ptr<Chunk> next(const std::string& path);
    double consumed() const { return src_->consumed(); }
    uint64_t text_bytes() const { return text_bytes_; }          // text handed out so far
private:
    std::unique_ptr<TextSource> src_;
    bool fastq_, eof_ = false, ended_ = false;
    size_t chunk_bytes_;
    std::vector<char> carry_;                     // bytes after the last whole record of the previous chunk
    uint64_t text_bytes_ = 0;
    // how many chunks are alive: shared with the chunks themselves, which may outlive the reader (a batch holds its
    // chunk until it is written)
    struct Gate { std::mutex m; std::condition_variable cv; int live = 0; };
    std::shared_ptr<Gate> gate_;
    int max_live_;
};

}  // namespace host
