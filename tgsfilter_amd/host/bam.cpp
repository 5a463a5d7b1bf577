#include "bam.h"

#include <zlib.h>

#include <cstdint>
#include <cstring>

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

bool inflate_members(const char* data, size_t size, std::vector<char>& out, std::string& err)
{
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
            inflateReset(&z);                                   // next BGZF block / gzip member
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
    std::string seq, qual;
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
        seq.resize(l_seq); qual.resize(l_seq);
        for (uint32_t i = 0; i < l_seq; i++) {
            const unsigned char b = r[o_seq + (i >> 1)];
            seq[i] = kBase[(i & 1) ? (b & 15) : (b >> 4)];
            const unsigned char q = (unsigned char)(r[o_qual + i] + 33);
            if (q == '\n') { err = "unsupported quality value in " + std::string(name, nlen); return false; }
            qual[i] = (char)q;
        }
        emit(text, name, nlen, seq, qual);
    }
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
