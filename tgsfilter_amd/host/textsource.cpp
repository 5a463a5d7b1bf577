#include "textsource.h"

#include <unistd.h>

#include <cstring>
#include <iostream>

#include "bam.h"

namespace host {

namespace {

class PlainBytes : public ByteStream {
public:
    PlainBytes(const char* d, size_t n) : d_(d), n_(n) {}
    double consumed() const override { return n_ ? (double)at_ / (double)n_ : 1.0; }
    bool read(char* dst, size_t cap, size_t& got, bool& eof, std::string&) override {
        got = n_ - at_ < cap ? n_ - at_ : cap;
        memcpy(dst, d_ + at_, got);
        at_ += got;
        eof = at_ >= n_;
        return true;
    }
private:
    const char* d_;
    size_t n_, at_ = 0;
};

class PassText : public TextSource {              // the decompressed bytes are the text
public:
    explicit PassText(std::unique_ptr<ByteStream> b) : b_(std::move(b)) {}
    double consumed() const override { return b_->consumed(); }
    bool read(char* dst, size_t cap, size_t& got, bool& eof, std::string& err) override { return b_->read(dst, cap, got, eof, err); }
private:
    std::unique_ptr<ByteStream> b_;
};

// a ByteStream with its first bytes already read (to look at the format), handed out again first
class Peeked : public ByteStream {
public:
    Peeked(std::unique_ptr<ByteStream> b, std::vector<char> head, bool eof) : b_(std::move(b)), head_(std::move(head)), eof_(eof) {}
    double consumed() const override { return b_->consumed(); }
    bool read(char* dst, size_t cap, size_t& got, bool& eof, std::string& err) override {
        if (at_ < head_.size()) {
            got = head_.size() - at_ < cap ? head_.size() - at_ : cap;
            memcpy(dst, head_.data() + at_, got);
            at_ += got;
            eof = eof_ && at_ >= head_.size();
            return true;
        }
        if (eof_) { got = 0; eof = true; return true; }
        return b_->read(dst, cap, got, eof, err);
    }
private:
    std::unique_ptr<ByteStream> b_;
    std::vector<char> head_;
    size_t at_ = 0;
    bool eof_;
};

}  // namespace

std::unique_ptr<TextSource> open_text(const char* data, size_t size, bool sam_or_bam, std::string& err)
{
    const bool gz = size >= 2 && (unsigned char)data[0] == 0x1f && (unsigned char)data[1] == 0x8b;
    std::unique_ptr<ByteStream> bytes = gz ? make_gz_bytes(data, size) : std::unique_ptr<ByteStream>(new PlainBytes(data, size));
    if (!sam_or_bam) return std::unique_ptr<TextSource>(new PassText(std::move(bytes)));
    // BAM or SAM by content, as hts_open decides
    std::vector<char> head(1u << 16);
    size_t got = 0;
    bool eof = false;
    if (!bytes->read(head.data(), head.size(), got, eof, err)) return nullptr;
    head.resize(got);
    const bool bam = got >= 4 && memcmp(head.data(), "BAM\1", 4) == 0;
    if (got >= 4 && memcmp(head.data(), "CRAM", 4) == 0) { err = "CRAM input is not supported"; return nullptr; }
    std::unique_ptr<ByteStream> again(new Peeked(std::move(bytes), std::move(head), eof));
    return bam ? make_bam_text(std::move(again)) : make_sam_text(std::move(again));
}

ChunkReader::ChunkReader(std::unique_ptr<TextSource> src, bool fastq, size_t chunk_bytes, int max_live)
    : src_(std::move(src)), fastq_(fastq), chunk_bytes_(chunk_bytes), gate_(new Gate), max_live_(max_live) {}

std::shared_ptr<Chunk> ChunkReader::next(const std::string& path)
{
    if (ended_) return nullptr;
    {
        std::unique_lock<std::mutex> l(gate_->m);
        gate_->cv.wait(l, [&] { return gate_->live < max_live_; });
        gate_->live++;
    }
    std::shared_ptr<Gate> gate = gate_;
    std::shared_ptr<Chunk> c(new Chunk, [gate](Chunk* x) {
        delete x;
        { std::lock_guard<std::mutex> l(gate->m); gate->live--; }
        gate->cv.notify_one();
    });
    size_t cap = std::max(chunk_bytes_, carry_.size() * 2 + 4096);
    for (;;) {                                      // until the buffer holds at least one whole record (or the input ends)
        c->buf.reset(new char[cap]);
        c->cap = cap;
        if (!carry_.empty()) memcpy(c->buf.get(), carry_.data(), carry_.size());
        size_t have = carry_.size();
        while (!eof_ && have < cap) {
            size_t got = 0;
            std::string err;
            if (!src_->read(c->buf.get() + have, cap - have, got, eof_, err)) {
                if (err == "error while decompressing") std::cerr << "Error: Error encountered while decompressing file: " << path << std::endl;
                else std::cerr << "Error: " << err << " (" << path << ")" << std::endl;
                fflush(nullptr);
                _exit(255);
            }
            have += got;
            if (got == 0 && !eof_) break;             // the source's next piece does not fit in what is left
        }
        // whole records from the front; what follows the last one is carried into the next chunk
        FastxReader rd(c->buf.get(), have, fastq_, 1, &c->message);
        Record r;
        c->recs.clear();
        bool incomplete = false;
        while (rd.next_partial(r, eof_, incomplete)) {
            Rec x;
            x.name = r.name.data(); x.name_len = (uint32_t)r.name.size();
            x.seq = r.seq.data(); x.len = (uint32_t)r.seq.size();
            x.qual = fastq_ ? r.qual.data() : r.seq.data();
            c->recs.push_back(x);
        }
        const size_t used = (size_t)(rd.pos() - c->buf.get());
        if (incomplete && c->recs.empty() && !eof_) {     // one record larger than the buffer: a bigger one
            carry_.assign(c->buf.get(), c->buf.get() + have);
            cap *= 2;
            continue;
        }
        c->size = used;
        if (incomplete) carry_.assign(c->buf.get() + used, c->buf.get() + have);
        else { carry_.clear(); ended_ = true; }       // end of input, or the stream ended at a malformed record
        break;
    }
    c->last = ended_;
    text_bytes_ += c->size;
    return c;
}

}  // namespace host
