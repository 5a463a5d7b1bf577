// tgsf_core.h -- lane-level primitives of the per-read filtering hot path (gfx950).
//
// Device code: compiled by hipcc for the product (libtgsf.so) on top of tgsf_hip.h.  tests/emul compiles the same
// sources with -DTGSF_EMUL on top of tgsf_emul.h, a serial lane-by-lane emulation that lets the kernel logic be checked
// against the oracle on a box without a GPU (test infrastructure; the product has no CPU path).
//
// Reference behaviour restated here (file:line into /root/reference):
//   Myers/Hyyro column step           include/edlib.cpp:409-444  (calculateBlock)
//   infix (HW) scan, best + ends      include/edlib.cpp:547-704
//   start locations (reverse SHW)     include/edlib.cpp:223-258
//   path of the first location        include/edlib.cpp:945-1144 (traceback priority up,left,diag)
//   per-base QC columns               src/TGSFilter.cpp:1462-1476
#pragma once
#include <stdint.h>
#if defined(TGSF_EMUL)
#include "tgsf_emul.h"
#else
#include "tgsf_hip.h"
#endif

namespace tgsf {

constexpr int kMaxAdapters = 32;
constexpr int kMaxQ = 8192;        // adapters up to 256 bp run in registers (1..4 words); longer ones (-a accepts any length, the
                                   // reference's edlib is multi-block, include/edlib.cpp:182-185) in kWideNW-word arrays walked by
                                   // run-time loops.  The path of the first location comes from a traceback while
                                   // (2*8+4)*blocks*T + 8*T < 1 MiB (include/edlib.cpp:1191-1193: every adapter up to 1 280 bp), and
                                   // from Hirschberg's divide and conquer beyond, as in edlib (alignment_length_w)
constexpr int kPeqW = 4;           // words per symbol in the standard-layout Peq tables (adapters <= 256 bp)
constexpr int kWideNW = kMaxQ / 64;   // words per symbol in the wide tables (built only when an adapter needs them)
constexpr int kBin = 100;          // CalcAvgQuality bin width
constexpr int kTileBins = 64;      // one bin per lane
constexpr int kTileBases = kBin * kTileBins;   // 6400 bases per stats tile
constexpr int kSegCols = 1024;     // columns of the read middle owned by one lane of the infix scan
constexpr int kMidThreads = 256;   // lanes of a workgroup of the middle scans (k_mid_flat, k_mid_scan1): their launch bound
constexpr int kMaxRegions = 64;    // disjoint drop regions per read the region kernel can hold
constexpr int kMidListMax = 1024;  // candidates of one read beyond which the scan is redone into position-ordered arrays
constexpr int kSuffixMaxK = 12;    // adapters of 33..64 bp searched within at most this many differences go through the middle scan's 32-row filter (k_mid_flat)

// ---------------------------------------------------------------------------
// Bit-vector edit distance, standard layout: row r of the adapter is bit r%64 of
// word r/64.  Rows >= Q of the last word are don't-care (information only moves
// towards higher bits).  Used by every off-the-hot-loop search (end windows,
// start locations) and by adapters of 65..256 bp in the middle scan.
// ---------------------------------------------------------------------------
template <int NW>
struct Bv {
    uint64_t p[NW], m[NW];
    int score;           // value of row Q in the current column
};

template <int NW>
TGSF_HD void bv_init(Bv<NW>& s, int Q) {
#pragma unroll
    for (int w = 0; w < NW; w++) { s.p[w] = ~0ull; s.m[w] = 0ull; }
    s.score = Q;
}

// One text column.  eq[w]: rows equal to the text symbol.  hin_top: horizontal
// delta entering row 1: 0 for an infix search (free start, edlib.cpp:584), +1 for
// prefix/global searches.
template <int NW>
TGSF_HD void bv_step(Bv<NW>& s, const uint64_t* eq, int hin_top, int Q) {
    int hin = hin_top;
#pragma unroll
    for (int w = 0; w < NW; w++) {
        uint64_t Eq = eq[w], Pv = s.p[w], Mv = s.m[w];
        uint64_t Xv = Eq | Mv;
        if (hin < 0) Eq |= 1ull;
        uint64_t Xh = (((Eq & Pv) + Pv) ^ Pv) | Eq;
        uint64_t Ph = Mv | ~(Xh | Pv);
        uint64_t Mh = Pv & Xh;
        const int bit = (w == NW - 1) ? ((Q - 1) & 63) : 63;
        int hout = (int)((Ph >> bit) & 1ull) - (int)((Mh >> bit) & 1ull);
        Ph <<= 1; Mh <<= 1;
        if (hin < 0) Mh |= 1ull;
        if (hin > 0) Ph |= 1ull;
        s.p[w] = Mh | ~(Xv | Ph);
        s.m[w] = Ph & Xv;
        hin = hout;
    }
    s.score += hin;
}

// ---------------------------------------------------------------------------
// Hot-loop variant for Q <= 64 in infix mode: the adapter sits in the TOP Q bits
// of one 64-bit word; the low 64-Q bits are wildcard rows (Eq=1, Pv=Mv=0) which
// stay at distance 0 for ever (D[0][j]=0 boundary).  Same recurrences as
// edlib.cpp:416-441 with hin = 0, rearranged for gfx950, where every VALU op of a
// loop that contains 3-input ops issues in 4 cycles, so instruction COUNT is what
// matters (measured: tools/valu_rates.hip):
//   Xh | Pv  =  ((sum ^ Pv) | Eq) | Pv  =  sum | Pv | Eq          one v_or3 per half
//   Pv & Xh  =  (Pv & ~sum) | (Eq & Pv) =  Pv & (Eq | ~sum)        one v_bitop3 per half
//   Mv | ~u,  Mh' | ~x                                              one v_bfi / v_bitop3 per half
//   Ph' & Xv =  Ph' & (Eq | Mv),   Xv | Ph' = Eq | Mv | Ph'        one v_bitop3 / v_or3 per half
// (17 instructions per column: 2 and, 1 add64, 2 shift64, 12 three-input ops)
// and the bottom-row value is not carried along: D[Q][j] is the sum of the
// vertical deltas of column j, i.e. popcount(Pv) - popcount(Mv) (wildcard rows
// contribute 0), evaluated only where it is needed.
// ---------------------------------------------------------------------------
struct Hot {
    uint64_t p, m;
};
TGSF_HD void hot_init(Hot& s, int Q) {
    s.p = (Q >= 64) ? ~0ull : (~0ull << (64 - Q));
    s.m = 0ull;
}
TGSF_HD void hot_step(Hot& s, uint64_t Eq) {
    const uint64_t Pv = s.p, Mv = s.m;
    const uint64_t t = Eq & Pv;                          // 2 v_and
    const uint64_t sum = t + Pv;                         // 1 v_lshl_add_u64
    const uint64_t u = sum | Pv | Eq;                    // 2 v_or3           (= Xh | Pv)
    uint64_t Ph = Mv | ~u;                               // 2 v_bfi
    uint64_t Mh = (sum & t) | (~sum & Pv);               // 2 v_bfi           (= Pv & Xh)
    Ph <<= 1; Mh <<= 1;                                  // 2 v_lshlrev_b64 (operands are register pairs)
    pin(Ph);                                             // keep the shifted pair as is (else its high half is re-derived with a v_alignbit)
    const uint64_t x = Eq | Mv | Ph;                     // 2 v_or3           (= Xv | Ph')
    s.p = Mh | ~x;                                       // 2 v_bfi
    // Mv' = Ph' & (Eq | Mv) is one 3-input function per half; its result only feeds bitwise ops, so it
    // does not need to sit in a register pair (the builtin's results do not)
    const uint32_t nml = bitop3<0xE0>((uint32_t)Ph, (uint32_t)Eq, (uint32_t)Mv);
    const uint32_t nmh = bitop3<0xE0>((uint32_t)(Ph >> 32), (uint32_t)(Eq >> 32), (uint32_t)(Mv >> 32));
    s.m = ((uint64_t)nmh << 32) | nml;                   // 2 v_bitop3
}
TGSF_HD int hot_score(const Hot& s) { return (int)popc64(s.p) - (int)popc64(s.m); }
// bottom-row value <= lim  <=>  popcount(Pv) <= popcount(Mv) + lim: two accumulating v_bcnt_u32_b32 per side (plain
// popcounts: a chain of inline-assembly counts makes the compiler pad it with s_nop; the pin keeps the sum from being
// regrouped into two plain counts and a three-input add)
TGSF_HD bool hot_within(const Hot& s, int lim) {
    const uint32_t p = popc32((uint32_t)(s.p >> 32)) + popc32((uint32_t)s.p);
    uint32_t m = popc32((uint32_t)s.m) + (uint32_t)lim;
    pin(m);
    m = popc32((uint32_t)(s.m >> 32)) + m;
    return (int)p <= (int)m;
}
TGSF_HD uint64_t hot_eq(const Hot&, uint64_t top) { return top; }

// The same column for adapters of at most 32 bp (8 of the reference's 22 library adapters, src/TGSFilter.cpp:2974-2989:
// the ligation and barcoding kits): the adapter in the top Q bits of ONE 32-bit word -- 10 instructions instead of 17
// (1 and, 1 add, 2 shifts, 6 three-input ops), a dword of LDS per symbol instead of two.
struct Hot32 {
    uint32_t p, m;
};
TGSF_HD void hot_init(Hot32& s, int Q) {
    s.p = (Q >= 32) ? ~0u : (~0u << (32 - Q));
    s.m = 0u;
}
TGSF_HD void hot_step(Hot32& s, uint32_t Eq) {
    const uint32_t Pv = s.p, Mv = s.m;
    const uint32_t t = Eq & Pv;                          // v_and
    const uint32_t sum = t + Pv;                         // v_add_u32
    // every three-input function spelled out (the compiler forms v_or3 / v_bfi from 64-bit expressions, not from these)
    const uint32_t u = bitop3<0xFE>(sum, Pv, Eq);        // sum | Pv | Eq                 (= Xh | Pv)
    uint32_t Ph = bitop3<0xF3>(Mv, u, u);                // Mv | ~u
    uint32_t Mh = bitop3<0xD0>(Pv, Eq, sum);             // Pv & (Eq | ~sum)              (= Pv & Xh)
    Ph <<= 1; Mh <<= 1;                                  // 2 v_lshlrev_b32
    const uint32_t x = bitop3<0xFE>(Eq, Mv, Ph);         // Eq | Mv | Ph'                 (= Xv | Ph')
    s.p = bitop3<0xF3>(Mh, x, x);                        // Mh' | ~x
    s.m = bitop3<0xE0>(Ph, Eq, Mv);                      // Ph' & (Eq | Mv)
}
TGSF_HD int hot_score(const Hot32& s) { return (int)popc32(s.p) - (int)popc32(s.m); }
TGSF_HD bool hot_within(const Hot32& s, int lim) { return (int)popc32(s.p) <= (int)(popc32(s.m) + (uint32_t)lim); }   // (one accumulating v_bcnt per side)
// the 32-bit Eq row from the 64-bit top-aligned one: its high half (bits below the adapter are wildcards in both)
TGSF_HD uint32_t hot_eq(const Hot32&, uint64_t top) { return (uint32_t)(top >> 32); }

// ---------------------------------------------------------------------------
// QC columns for 4 bases at a time (SWAR).  s: 4 sequence bytes, q: 4 quality
// bytes (already masked to the valid ones; invalid bytes are 0 in both).
// cnt[c] += #bases of class c; qs[c] += 128 * sum of their quality bytes
// (c: A,T,G,C -- src/TGSFilter.cpp:1462-1474, case-insensitive); qs[4] += sum
// of all 4 quality bytes.  Quality bytes must be < 128.
// ---------------------------------------------------------------------------
TGSF_HD void qc_accum4(const QcConsts& kc, uint32_t s, uint32_t q, uint32_t* cnt /*[4]*/, uint32_t* qs /*[5]*/) {
    const uint32_t x7 = s & 0x5F5F5F5Fu;         // fold lower case onto upper case, drop bit 7
#pragma unroll
    for (int c = 0; c < 4; c++) {                // kc.k: A T G C
        const uint32_t m = ~(xad7f(x7, kc.k[c]) | s) & 0x80808080u;     // 0x80 where the base is of class c
        cnt[c] += popc32(m);
        qs[c] = udot4(q, m, qs[c]);
    }
    qs[4] = udot4(q, 0x01010101u, qs[4]);
}

// mean-quality gate: src/TGSFilter.cpp:1478 (double(sumQ)/len), :1947 (compare
// against float thresholds promoted to double).  sum is the uint64 accumulator.
TGSF_HD double mean_q(uint64_t sum, uint32_t len) { return (double)sum / (double)len; }
TGSF_HD bool q_fail(double mq, float min_q, float max_q) {
    return (mq < (double)min_q) || (mq > (double)max_q);
}

}  // namespace tgsf
