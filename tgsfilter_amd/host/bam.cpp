#include "bam.h"

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
        if (n - at < 18 || p[at] != 0x1f || p[at + 1] != 0x8b || p[at + 2] != 8 || !(p[at + 3] & 4)) return false;
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
    z_stream z;
    memset(&z, 0, sizeof z);
    if (inflateInit2(&z, 15 + 16) != Z_OK) { err = "zlib init failed"; return false; }
    z.next_in = (Bytef*)data;
    size_t left = size;
    std::vector<char> chunk(4 << 20);
    for (;;) {
        if (z.avail_in == 0) {
            if (left == 0) break;
            const size_t take = left > (1u << 30) ? (1u << 30) : left;
            z.avail_in = (uInt)take; left -= take;
        }
        z.next_out = (Bytef*)chunk.data(); z.avail_out = (uInt)chunk.size();
        const int rc = inflate(&z, Z_NO_FLUSH);
        out.insert(out.end(), chunk.data(), chunk.data() + (chunk.size() - z.avail_out));
        if (rc == Z_STREAM_END) {
            if (z.avail_in == 0 && left == 0) break;
            inflateReset(&z);                                   // next gzip member
        } else if (rc != Z_OK) { inflateEnd(&z); err = "error while decompressing"; return false; }
    }
    inflateEnd(&z);
    return true;
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

}  // namespace

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
