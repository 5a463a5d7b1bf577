#include "bam.h"
#include "fatal.h"

#include <dlfcn.h>
#include <sys/mman.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstring>
#include <thread>

namespace host {

namespace {

const char kBase[16] = {0, 'A', 'C', 0, 'G', 0, 0, 0, 'T', 0, 0, 0, 0, 0, 0, 'N'};   // src/TGSFilter.cpp:31

// SAM text -> 4-bit code (the order of "=ACMGRSVTWYHKDBN"), case-insensitive; digits 0-3 are A,C,G,T;
// anything else is N (15).
struct Nt16 {
    unsigned char t[256];
    Nt16() {
        memset(t, 15, sizeof t);
        const char* codes = "=ACMGRSVTWYHKDBN";
        for (int i = 0; i < 16; i++) {
            t[(unsigned char)codes[i]] = (unsigned char)i;
            if (codes[i] >= 'A' && codes[i] <= 'Z') t[(unsigned char)(codes[i] + 32)] = (unsigned char)i;
        }
        t['0'] = 1; t['1'] = 2; t['2'] = 4; t['3'] = 8;
    }
};

int worker_count()
{
    const unsigned hw = std::thread::hardware_concurrency();
    return (int)std::max(1u, std::min(hw ? hw - 1 : 1u, 16u));
}

template <class F>
void parallel_for(size_t n, F f)            // f(begin, end) on contiguous ranges
{
    const size_t T = std::min<size_t>((size_t)worker_count(), std::max<size_t>(n, 1));
    std::vector<std::thread> th;
    for (size_t t = 1; t < T; t++) th.emplace_back([=] { f(n * t / T, n * (t + 1) / T); });
    f(0, n / T);
    for (std::thread& x : th) x.join();
}

// BGZF: every gzip member carries its own size in a 'BC' extra subfield and its uncompressed size in the
// trailer, so the members can be located without inflating and inflated independently.
struct BgzfBlock { size_t at, payload, clen; uint32_t usize; size_t out; };
bool bgzf_index(const unsigned char* p, size_t n, std::vector<BgzfBlock>& blocks)
{
    size_t at = 0, out = 0;
    while (at < n) {
        if (n - at < 18 || p[at] != 0x1f || p[at + 1] != 0x8b || p[at + 2] != 8 || p[at + 3] != 4) return false;   // FEXTRA only: a member with a name, comment or header CRC goes the serial way
        const size_t xlen = (size_t)p[at + 10] | ((size_t)p[at + 11] << 8);
        if (n - at < 12 + xlen + 8) return false;
        size_t bsize = 0;
        for (size_t x = at + 12; x + 4 <= at + 12 + xlen;) {
            const size_t slen = (size_t)p[x + 2] | ((size_t)p[x + 3] << 8);
            if (p[x] == 'B' && p[x + 1] == 'C' && slen == 2 && x + 6 <= at + 12 + xlen) bsize = ((size_t)p[x + 4] | ((size_t)p[x + 5] << 8)) + 1;
            x += 4 + slen;
        }
        if (bsize < 12 + xlen + 8 || at + bsize > n) return false;
        const unsigned char* tr = p + at + bsize - 8;
        const uint32_t usize = (uint32_t)tr[4] | ((uint32_t)tr[5] << 8) | ((uint32_t)tr[6] << 16) | ((uint32_t)tr[7] << 24);
        blocks.push_back({at, at + 12 + xlen, bsize - 12 - xlen - 8, usize, out});
        out += usize;
        at += bsize;
    }
    return true;
}

bool inflate_serial(const char* data, size_t size, std::vector<char>& out, std::string& err);   // through GzBytes, below

bool inflate_members(const char* data, size_t size, std::vector<char>& out, std::string& err)
{
    std::vector<BgzfBlock> blocks;
    if (bgzf_index(reinterpret_cast<const unsigned char*>(data), size, blocks)) {
        out.resize(blocks.empty() ? 0 : blocks.back().out + blocks.back().usize);
        std::atomic<bool> bad{false};
        parallel_for(blocks.size(), [&](size_t b0, size_t b1) {
            z_stream z;
            memset(&z, 0, sizeof z);
            if (inflateInit2(&z, -15) != Z_OK) { bad = true; return; }
            for (size_t b = b0; b < b1; b++) {
                inflateReset(&z);
                z.next_in = (Bytef*)(data + blocks[b].payload); z.avail_in = (uInt)blocks[b].clen;
                z.next_out = (Bytef*)(out.data() + blocks[b].out); z.avail_out = blocks[b].usize;
                const int rc = blocks[b].usize || blocks[b].clen > 2 ? inflate(&z, Z_FINISH) : Z_STREAM_END;
                if (rc != Z_STREAM_END || z.avail_out != 0) bad = true;
            }
            inflateEnd(&z);
        });
        if (bad) { err = "error while decompressing"; return false; }
        return true;
    }
    return inflate_serial(data, size, out, err);
}

void emit(std::vector<char>& text, const char* name, size_t nlen, const std::string& seq, const std::string& qual)
{
    text.push_back('@');
    text.insert(text.end(), name, name + nlen);
    text.push_back('\n');
    text.insert(text.end(), seq.begin(), seq.end());
    text.push_back('\n'); text.push_back('+'); text.push_back('\n');
    text.insert(text.end(), qual.begin(), qual.end());
    text.push_back('\n');
}

uint32_t le32(const unsigned char* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

bool decode_bam(const unsigned char* p, size_t n, std::vector<char>& text, std::string& err)
{
    size_t at = 4;
    auto need = [&](size_t k) { return at + k <= n; };
    if (!need(4)) { err = "truncated BAM header"; return false; }
    const uint32_t l_text = le32(p + at); at += 4;
    if (!need((size_t)l_text + 4)) { err = "truncated BAM header"; return false; }
    at += l_text;
    const uint32_t n_ref = le32(p + at); at += 4;
    for (uint32_t i = 0; i < n_ref; i++) {
        if (!need(4)) { err = "truncated BAM reference list"; return false; }
        const uint32_t l_name = le32(p + at); at += 4;
        if (!need((size_t)l_name + 4)) { err = "truncated BAM reference list"; return false; }
        at += (size_t)l_name + 4;
    }
    // records are located first (each carries its size), then decoded side by side into their places
    struct Rec { const unsigned char* r; uint32_t name_len, l_seq; size_t o_seq, o_qual, out; };
    std::vector<Rec> recs;
    size_t out = text.size();
    while (need(4)) {
        const uint32_t block = le32(p + at); at += 4;
        if (block < 32 || !need(block)) break;                   // sam_read1 < 0 ends the reference's loop silently
        const unsigned char* r = p + at;
        at += block;
        const uint32_t l_read_name = r[8];
        const uint32_t n_cigar = (uint32_t)r[12] | ((uint32_t)r[13] << 8);
        const uint32_t l_seq = le32(r + 16);
        const size_t o_name = 32, o_seq = o_name + l_read_name + 4ull * n_cigar, o_qual = o_seq + (l_seq + 1ull) / 2;
        if (o_qual + l_seq > block) break;
        const char* name = reinterpret_cast<const char*>(r + o_name);
        const size_t nlen = strnlen(name, l_read_name);
        if (l_seq == 0) { err = "BAM record without a sequence: " + std::string(name, nlen); return false; }
        recs.push_back({r, (uint32_t)nlen, l_seq, o_seq, o_qual, out});
        out += 1 + nlen + 1 + l_seq + 3 + l_seq + 1;            // @name\n SEQ \n+\n QUAL \n
    }
    text.resize(out);
    std::atomic<int> bad_q{-1};
    parallel_for(recs.size(), [&](size_t a, size_t b) {
        for (size_t i = a; i < b; i++) {
            const Rec& rc = recs[i];
            char* o = text.data() + rc.out;
            *o++ = '@';
            memcpy(o, rc.r + 32, rc.name_len); o += rc.name_len;
            *o++ = '\n';
            const unsigned char* sq = rc.r + rc.o_seq;
            for (uint32_t k = 0; k < rc.l_seq; k++) { const unsigned char by = sq[k >> 1]; o[k] = kBase[(k & 1) ? (by & 15) : (by >> 4)]; }
            o += rc.l_seq;
            *o++ = '\n'; *o++ = '+'; *o++ = '\n';
            const unsigned char* ql = rc.r + rc.o_qual;
            for (uint32_t k = 0; k < rc.l_seq; k++) {
                const unsigned char q = (unsigned char)(ql[k] + 33);
                if (q == '\n') bad_q = (int)i;
                o[k] = (char)q;
            }
            o += rc.l_seq;
            *o++ = '\n';
        }
    });
    if (bad_q >= 0) { err = "unsupported quality value in " + std::string(reinterpret_cast<const char*>(recs[(size_t)bad_q].r + 32), recs[(size_t)bad_q].name_len); return false; }
    return true;
}

bool decode_sam(const char* p, size_t n, std::vector<char>& text, std::string& err)
{
    static const Nt16 nt16;
    const char* end = p + n;
    bool in_header = true;
    std::string seq, qual;
    while (p < end) {
        const char* nl = static_cast<const char*>(memchr(p, '\n', (size_t)(end - p)));
        const char* le = nl ? nl : end;
        const char* next = nl ? nl + 1 : end;
        if (le > p && le[-1] == '\r') le--;
        if (in_header && p < le && *p == '@') { p = next; continue; }
        in_header = false;
        if (p == le) { p = next; continue; }
        const char* f[12];
        int nf = 0;
        f[nf++] = p;
        for (const char* c = p; c < le && nf < 12; c++) if (*c == '\t') f[nf++] = c + 1;
        if (nf < 11) break;                                      // malformed alignment line: the stream ends
        const size_t nlen = (size_t)(f[1] - 1 - f[0]);
        const char* s = f[9]; const size_t slen = (size_t)(f[10] - 1 - f[9]);
        const char* q = f[10]; const size_t qlen = (size_t)((nf > 11 ? f[11] - 1 : le) - f[10]);
        if (slen == 1 && s[0] == '*') { err = "SAM record without a sequence: " + std::string(f[0], nlen); return false; }
        const bool noq = qlen == 1 && q[0] == '*';
        if (!noq && qlen != slen) break;
        seq.resize(slen); qual.resize(slen);
        for (size_t i = 0; i < slen; i++) {
            seq[i] = kBase[nt16.t[(unsigned char)s[i]]];
            qual[i] = noq ? (char)(unsigned char)(0xFF + 33) : q[i];
        }
        emit(text, f[0], nlen, seq, qual);
        p = next;
    }
    return true;
}

// ---------------------------------------------------------------------------
// streams
// ---------------------------------------------------------------------------
// one BGZF member at p[at..): false if it is not one (then the serial inflater takes over from there)
bool bgzf_peek(const unsigned char* p, size_t n, size_t at, BgzfBlock& b)
{
    if (n - at < 18 || p[at] != 0x1f || p[at + 1] != 0x8b || p[at + 2] != 8 || p[at + 3] != 4) return false;
    const size_t xlen = (size_t)p[at + 10] | ((size_t)p[at + 11] << 8);
    if (n - at < 12 + xlen + 8) return false;
    size_t bsize = 0;
    for (size_t x = at + 12; x + 4 <= at + 12 + xlen;) {
        const size_t slen = (size_t)p[x + 2] | ((size_t)p[x + 3] << 8);
        if (p[x] == 'B' && p[x + 1] == 'C' && slen == 2 && x + 6 <= at + 12 + xlen) bsize = ((size_t)p[x + 4] | ((size_t)p[x + 5] << 8)) + 1;
        x += 4 + slen;
    }
    if (bsize < 12 + xlen + 8 || at + bsize > n) return false;
    const unsigned char* tr = p + at + bsize - 8;
    b.at = at; b.payload = at + 12 + xlen; b.clen = bsize - 12 - xlen - 8;
    b.usize = (uint32_t)tr[4] | ((uint32_t)tr[5] << 8) | ((uint32_t)tr[6] << 16) | ((uint32_t)tr[7] << 24);
    b.out = 0;
    return true;
}

// libdeflate's whole-member gzip decompressor, bound at run time when the system has the library (as Output's compressor
// in pipeline.h): about three times zlib's speed, but it wants a member's output in one buffer -- so it serves files made
// of members of moderate size (concatenated per-run files, bgzip-less multi-member archives); a member that does not fit
// the buffer limit goes through zlib's streaming inflater as before.
class Inflater {
public:
    static const Inflater& get() { static const Inflater d; return d; }
    bool ok() const { return alloc_ && run_ && free__; }
    void* alloc() const { return alloc_(); }
    // 0 = done, 3 = the output buffer is too small, anything else = not decodable this way
    int run(void* d, const void* in, size_t n, void* out, size_t room, size_t* used, size_t* got) const { return run_(d, in, n, out, room, used, got); }
    void free_(void* d) const { free__(d); }
private:
    Inflater() {
        if (knob("TGSF_ZLIB_INPUT")) return;                  // test knob: zlib only
        for (const char* name : {"libdeflate.so.0", "libdeflate.so"}) {
            void* h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (!h) continue;
            alloc_ = reinterpret_cast<void* (*)()>(dlsym(h, "libdeflate_alloc_decompressor"));
            run_ = reinterpret_cast<int (*)(void*, const void*, size_t, void*, size_t, size_t*, size_t*)>(dlsym(h, "libdeflate_gzip_decompress_ex"));
            free__ = reinterpret_cast<void (*)(void*)>(dlsym(h, "libdeflate_free_decompressor"));
            if (ok()) return;
        }
    }
    void* (*alloc_)() = nullptr;
    int (*run_)(void*, const void*, size_t, void*, size_t, size_t*, size_t*) = nullptr;
    void (*free__)(void*) = nullptr;
};

class GzBytes : public ByteStream {
public:
    GzBytes(const char* data, size_t size) : p_(reinterpret_cast<const unsigned char*>(data)), n_(size) {
        memset(&z_, 0, sizeof z_);
        BgzfBlock b;
        bgzf_ = size > 0 && bgzf_peek(p_, n_, 0, b);
    }
    ~GzBytes() override { if (z_open_) inflateEnd(&z_); if (ld_) Inflater::get().free_(ld_); }
    double consumed() const override { return n_ ? (double)at_ / (double)n_ : 1.0; }
    // the mapping of the compressed bytes already decoded goes (they stay in the page cache for the second pass):
    // the resident size of a run does not grow with the file
    void drop_consumed() {
        const size_t done = at_ - (z_open_ ? (size_t)z_.avail_in : 0);
        const size_t upto = done > (1u << 20) ? (done - (1u << 20)) & ~size_t(4095) : 0;
        const uintptr_t lo = ((uintptr_t)p_ + dropped_ + 4095) & ~uintptr_t(4095), hi = ((uintptr_t)p_ + upto) & ~uintptr_t(4095);
        if (hi > lo + (8u << 20)) { madvise((void*)lo, hi - lo, MADV_DONTNEED); dropped_ = upto; }
    }
    bool read(char* dst, size_t cap, size_t& got, bool& eof, std::string& err) override {
        got = 0;
        drop_consumed();
        if (bgzf_) {
            // the members that fit, located from their 'BC' fields, inflated side by side
            std::vector<BgzfBlock> blocks;
            size_t out = 0;
            while (at_ < n_) {
                BgzfBlock b;
                if (!bgzf_peek(p_, n_, at_, b)) { if (blocks.empty()) bgzf_ = false; break; }
                if (out + b.usize > cap) break;
                b.out = out; out += b.usize;
                blocks.push_back(b);
                at_ = b.at + (b.payload - b.at) + b.clen + 8;
            }
            if (!blocks.empty() || bgzf_) {
                std::atomic<bool> bad{false};
                parallel_for(blocks.size(), [&](size_t b0, size_t b1) {
                    z_stream z;
                    memset(&z, 0, sizeof z);
                    if (inflateInit2(&z, -15) != Z_OK) { bad = true; return; }
                    for (size_t b = b0; b < b1; b++) {
                        inflateReset(&z);
                        z.next_in = (Bytef*)(p_ + blocks[b].payload); z.avail_in = (uInt)blocks[b].clen;
                        z.next_out = (Bytef*)(dst + blocks[b].out); z.avail_out = blocks[b].usize;
                        const int rc = blocks[b].usize || blocks[b].clen > 2 ? inflate(&z, Z_FINISH) : Z_STREAM_END;
                        if (rc != Z_STREAM_END || z.avail_out != 0) bad = true;
                    }
                    inflateEnd(&z);
                });
                if (bad) { err = "error while decompressing"; return false; }
                got = out;
                eof = at_ >= n_;
                return true;
            }
        }
        // any other gzip, member after member (:632-639): whole members through libdeflate while they fit ...
        static const size_t member_cap = [] { const char* e = knob("TGSF_GZ_MEMBER_CAP"); return e ? (size_t)strtoull(e, nullptr, 10) : (size_t)(512u << 20); }();
        const Inflater& fast = Inflater::get();
        while (fast.ok() && !in_member_ && got < cap) {
            if (mem_at_ < mem_.size()) {                             // what is left of the member decoded last
                const size_t k = std::min(cap - got, mem_.size() - mem_at_);
                memcpy(dst + got, mem_.data() + mem_at_, k);
                got += k; mem_at_ += k;
                continue;
            }
            if (at_ >= n_) { done_ = true; break; }
            if (!ld_) ld_ = fast.alloc();
            if (!ld_) break;
            size_t room = std::max<size_t>(mem_.capacity(), std::min<size_t>(member_cap, 64u << 20));
            int rc = 3;
            size_t used = 0, out = 0;
            while (rc == 3 && room <= member_cap) {
                mem_.resize(room);
                rc = fast.run(ld_, p_ + at_, n_ - at_, mem_.data(), room, &used, &out);
                if (rc == 3) room *= 2;
            }
            if (rc != 0) { mem_.clear(); mem_at_ = 0; in_member_ = true; break; }     // too large (or damaged): zlib takes this member
            mem_.resize(out); mem_at_ = 0;
            at_ += used;
        }
        if (done_ || (fast.ok() && !in_member_)) { eof = done_ && mem_at_ >= mem_.size(); return true; }
        // ... and zlib's streaming inflater for the rest
        if (!z_open_) {
            if (inflateInit2(&z_, 15 + 16) != Z_OK) { err = "zlib init failed"; return false; }
            z_open_ = true;
        }
        while (got < cap) {
            if (z_.avail_in == 0) {
                if (at_ >= n_) break;
                const size_t take = std::min<size_t>(n_ - at_, 1u << 30);
                z_.next_in = (Bytef*)(p_ + at_); z_.avail_in = (uInt)take;
                at_ += take;
            }
            z_.next_out = (Bytef*)(dst + got);
            const size_t room = std::min<size_t>(cap - got, 1u << 30);
            z_.avail_out = (uInt)room;
            const int rc = inflate(&z_, Z_NO_FLUSH);
            got += room - z_.avail_out;
            if (rc == Z_STREAM_END) {
                if (z_.avail_in == 0 && at_ >= n_) { done_ = true; break; }
                inflateReset(&z_);                              // next gzip member
                if (fast.ok()) {                                // ... which may fit the fast way again
                    at_ -= z_.avail_in; z_.avail_in = 0;
                    in_member_ = false;
                    break;
                }
            } else if (rc != Z_OK) { err = "error while decompressing"; return false; }
        }
        if (in_member_ && z_.avail_in == 0 && at_ >= n_ && got < cap) done_ = true;
        eof = done_;
        return true;
    }
private:
    const unsigned char* p_;
    size_t n_, at_ = 0, dropped_ = 0;
    bool bgzf_ = false, z_open_ = false, done_ = false;
    bool in_member_ = false;                  // zlib is in the middle of a member (one too large for the fast way)
    void* ld_ = nullptr;
    std::vector<char> mem_;                   // the member decoded last by libdeflate, handed out from mem_at_
    size_t mem_at_ = 0;
    z_stream z_;
};

// any series of gzip members into one vector (inputs small enough to be decoded whole): the same decoder as the stream
bool inflate_serial(const char* data, size_t size, std::vector<char>& out, std::string& err)
{
    GzBytes g(data, size);
    bool eof = false;
    while (!eof) {
        const size_t old = out.size(), piece = 64u << 20;
        out.resize(old + piece);
        size_t got = 0;
        if (!g.read(out.data() + old, piece, got, eof, err)) return false;
        out.resize(old + got);
    }
    return true;
}

// decompressed bytes with a window that keeps what a decoder has not consumed yet
class Window {
public:
    explicit Window(std::unique_ptr<ByteStream> b) : bytes_(std::move(b)) {}
    const unsigned char* data() const { return reinterpret_cast<const unsigned char*>(buf_.data()) + at_; }
    size_t size() const { return buf_.size() - at_; }
    void consume(size_t k) { at_ += k; }
    bool eof() const { return eof_; }
    double consumed() const { return bytes_->consumed(); }
    bool more(std::string& err) {                             // append another piece; false on error
        if (eof_) return true;
        if (at_) { buf_.erase(buf_.begin(), buf_.begin() + (long)at_); at_ = 0; }
        const size_t piece = 32u << 20, old = buf_.size();
        buf_.resize(old + piece);
        size_t got = 0;
        if (!bytes_->read(buf_.data() + old, piece, got, eof_, err)) return false;
        buf_.resize(old + got);
        return true;
    }
private:
    std::unique_ptr<ByteStream> bytes_;
    std::vector<char> buf_;
    size_t at_ = 0;
    bool eof_ = false;
};

class BamText : public TextSource {
public:
    explicit BamText(std::unique_ptr<ByteStream> b) : w_(std::move(b)) {}
    double consumed() const override { return w_.consumed(); }
    bool read(char* dst, size_t cap, size_t& got, bool& eof, std::string& err) override {
        got = 0; eof = false;
        if (done_) { eof = true; return true; }
        if (!header_done_) {
            for (;;) {
                const long h = header_len(w_.data(), w_.size());
                if (h >= 0) { w_.consume((size_t)h); header_done_ = true; break; }
                if (w_.eof()) { err = "truncated BAM header"; return false; }
                if (!w_.more(err)) return false;
            }
        }
        struct R { const unsigned char* r; uint32_t name_len, l_seq; size_t o_seq, o_qual, out; };
        std::vector<R> recs;
        size_t out = 0;
        bool ended = false;
        for (;;) {
            const unsigned char* p = w_.data();
            const size_t n = w_.size();
            size_t at = 0;
            recs.clear(); out = 0;
            bool full = false;
            while (at + 4 <= n) {
                const uint32_t block = le32(p + at);
                if (block < 32) { ended = true; break; }             // sam_read1 < 0 ends the reference's loop silently
                if (at + 4 + block > n) break;                       // the rest of this record is not here yet
                const unsigned char* r = p + at + 4;
                const uint32_t l_read_name = r[8];
                const uint32_t n_cigar = (uint32_t)r[12] | ((uint32_t)r[13] << 8);
                const uint32_t l_seq = le32(r + 16);
                const size_t o_seq = 32 + l_read_name + 4ull * n_cigar, o_qual = o_seq + (l_seq + 1ull) / 2;
                if (o_qual + l_seq > block) { ended = true; break; }
                const char* name = reinterpret_cast<const char*>(r + 32);
                const size_t nlen = strnlen(name, l_read_name);
                if (l_seq == 0) { err = "BAM record without a sequence: " + std::string(name, nlen); return false; }
                const size_t need = 1 + nlen + 1 + l_seq + 3 + (size_t)l_seq + 1;
                if (out + need > cap) { full = true; break; }
                recs.push_back({r, (uint32_t)nlen, l_seq, o_seq, o_qual, out});
                out += need;
                at += 4 + (size_t)block;
            }
            if (!recs.empty() || full || ended) { consume_ = at; break; }
            if (w_.eof()) { ended = true; consume_ = at; break; }    // a truncated last record: silently the end
            if (!w_.more(err)) return false;
        }
        std::atomic<int> bad_q{-1};
        parallel_for(recs.size(), [&](size_t a, size_t b) {
            for (size_t i = a; i < b; i++) {
                const R& rc = recs[i];
                char* o = dst + rc.out;
                *o++ = '@';
                memcpy(o, rc.r + 32, rc.name_len); o += rc.name_len;
                *o++ = '\n';
                const unsigned char* sq = rc.r + rc.o_seq;
                for (uint32_t k = 0; k < rc.l_seq; k++) { const unsigned char by = sq[k >> 1]; o[k] = kBase[(k & 1) ? (by & 15) : (by >> 4)]; }
                o += rc.l_seq;
                *o++ = '\n'; *o++ = '+'; *o++ = '\n';
                const unsigned char* ql = rc.r + rc.o_qual;
                for (uint32_t k = 0; k < rc.l_seq; k++) {
                    const unsigned char q = (unsigned char)(ql[k] + 33);
                    if (q == '\n') bad_q = (int)i;
                    o[k] = (char)q;
                }
                o += rc.l_seq;
                *o++ = '\n';
            }
        });
        if (bad_q >= 0) { err = "unsupported quality value in " + std::string(reinterpret_cast<const char*>(recs[(size_t)bad_q].r + 32), recs[(size_t)bad_q].name_len); return false; }
        w_.consume(consume_);
        got = out;
        if (ended) done_ = true;
        eof = done_ && got == 0;
        return true;
    }
private:
    // bytes of "BAM\1" header + reference list, or -1 while incomplete
    static long header_len(const unsigned char* p, size_t n) {
        size_t at = 4;
        if (n < 8) return -1;
        const uint32_t l_text = le32(p + at); at += 4;
        if (n < at + (size_t)l_text + 4) return -1;
        at += l_text;
        const uint32_t n_ref = le32(p + at); at += 4;
        for (uint32_t i = 0; i < n_ref; i++) {
            if (n < at + 4) return -1;
            const uint32_t l_name = le32(p + at); at += 4;
            if (n < at + (size_t)l_name + 4) return -1;
            at += (size_t)l_name + 4;
        }
        return (long)at;
    }
    Window w_;
    bool header_done_ = false, done_ = false;
    size_t consume_ = 0;
};

class SamText : public TextSource {
public:
    explicit SamText(std::unique_ptr<ByteStream> b) : w_(std::move(b)) {}
    double consumed() const override { return w_.consumed(); }
    bool read(char* dst, size_t cap, size_t& got, bool& eof, std::string& err) override {
        static const Nt16 nt16;
        got = 0; eof = false;
        if (done_) { eof = true; return true; }
        for (;;) {
            const char* base = reinterpret_cast<const char*>(w_.data());
            const char* p = base;
            const char* end = base + w_.size();
            bool full = false;
            while (p < end) {
                const char* nl = static_cast<const char*>(memchr(p, '\n', (size_t)(end - p)));
                if (!nl && !w_.eof()) break;                         // the rest of this line is not here yet
                const char* le = nl ? nl : end;
                const char* next = nl ? nl + 1 : end;
                if (le > p && le[-1] == '\r') le--;
                if (in_header_ && p < le && *p == '@') { p = next; continue; }
                in_header_ = false;
                if (p == le) { p = next; continue; }
                const char* f[12];
                int nf = 0;
                f[nf++] = p;
                for (const char* c = p; c < le && nf < 12; c++) if (*c == '\t') f[nf++] = c + 1;
                if (nf < 11) { done_ = true; break; }                // malformed alignment line: the stream ends
                const size_t nlen = (size_t)(f[1] - 1 - f[0]);
                const char* s = f[9]; const size_t slen = (size_t)(f[10] - 1 - f[9]);
                const char* q = f[10]; const size_t qlen = (size_t)((nf > 11 ? f[11] - 1 : le) - f[10]);
                if (slen == 1 && s[0] == '*') { err = "SAM record without a sequence: " + std::string(f[0], nlen); return false; }
                const bool noq = qlen == 1 && q[0] == '*';
                if (!noq && qlen != slen) { done_ = true; break; }
                const size_t need = 1 + nlen + 1 + slen + 3 + slen + 1;
                if (got + need > cap) { full = true; break; }
                char* o = dst + got;
                *o++ = '@'; memcpy(o, f[0], nlen); o += nlen; *o++ = '\n';
                for (size_t i = 0; i < slen; i++) o[i] = kBase[nt16.t[(unsigned char)s[i]]];
                o += slen;
                *o++ = '\n'; *o++ = '+'; *o++ = '\n';
                for (size_t i = 0; i < slen; i++) o[i] = noq ? (char)(unsigned char)(0xFF + 33) : q[i];
                o[slen] = '\n';
                got += need;
                p = next;
            }
            w_.consume((size_t)(p - base));
            if (done_ || full || got) break;
            if (w_.eof()) { done_ = true; break; }                   // every line has been taken
            if (!w_.more(err)) return false;
        }
        eof = done_ && got == 0;
        return true;
    }
private:
    Window w_;
    bool in_header_ = true, done_ = false;
};

}  // namespace

std::unique_ptr<ByteStream> make_gz_bytes(const char* data, size_t size) { return std::unique_ptr<ByteStream>(new GzBytes(data, size)); }
std::unique_ptr<TextSource> make_bam_text(std::unique_ptr<ByteStream> bytes) { return std::unique_ptr<TextSource>(new BamText(std::move(bytes))); }
std::unique_ptr<TextSource> make_sam_text(std::unique_ptr<ByteStream> bytes) { return std::unique_ptr<TextSource>(new SamText(std::move(bytes))); }

bool inflate_gzip(const char* data, size_t size, std::vector<char>& out, std::string& err)
{
    return inflate_members(data, size, out, err);
}

bool decode_sam_or_bam(const char* data, size_t size, std::vector<char>& text, std::string& err)
{
    std::vector<char> plain;
    if (size >= 2 && (unsigned char)data[0] == 0x1f && (unsigned char)data[1] == 0x8b) {
        if (!inflate_members(data, size, plain, err)) return false;
        data = plain.data(); size = plain.size();
    }
    if (size >= 4 && memcmp(data, "BAM\1", 4) == 0)
        return decode_bam(reinterpret_cast<const unsigned char*>(data), size, text, err);
    if (size >= 4 && memcmp(data, "CRAM", 4) == 0) { err = "CRAM input is not supported"; return false; }
    return decode_sam(data, size, text, err);
}

}  // namespace host
