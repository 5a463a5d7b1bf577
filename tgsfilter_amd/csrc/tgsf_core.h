// tgsf_core.h -- lane-level primitives of the per-read filtering hot path (gfx950).
//
// Everything here is written once and compiled twice: by hipcc for the device
// (the product, libtgsf.so) and by g++ with -DTGSF_EMUL for tests/emul, a
// serial lane-by-lane emulation of the same kernels that lets the kernel logic
// be checked against the oracle on a box without a GPU.  The emulation is test
// infrastructure; the product has no CPU path.
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
#define TGSF_HD inline
#define TGSF_D inline
#else
#include <hip/hip_runtime.h>
#define TGSF_HD __host__ __device__ __forceinline__
#define TGSF_D __device__ __forceinline__
#endif

namespace tgsf {

constexpr int kMaxAdapters = 32;
constexpr int kMaxQ = 1280;        // adapters up to 256 bp run in registers (1..4 words); longer ones (-a accepts any length, the
                                   // reference's edlib is multi-block, include/edlib.cpp:182-185) in kWideNW-word arrays.  1 280 bp is
                                   // where exactness ends: edlib finds the path of the first location by traceback while
                                   // (2*8+4)*blocks*T + 8*T < 1 MiB (include/edlib.cpp:1191-1193; T <= 2Q-1 columns) -- true for every
                                   // adapter up to 20 blocks -- and by Hirschberg's divide and conquer beyond, which may choose another
                                   // of the equally good paths (a different match count): not restated, refused
constexpr int kPeqW = 4;           // words per symbol in the standard-layout Peq tables (adapters <= 256 bp)
constexpr int kWideNW = kMaxQ / 64;   // words per symbol in the wide tables (built only when an adapter needs them)
constexpr int kBin = 100;          // CalcAvgQuality bin width
constexpr int kTileBins = 64;      // one bin per lane
constexpr int kTileBases = kBin * kTileBins;   // 6400 bases per stats tile
constexpr int kSegCols = 1024;     // columns of the read middle owned by one lane of the infix scan
constexpr int kMaxRegions = 64;    // disjoint drop regions per read the region kernel can hold
constexpr int kMidListMax = 1024;  // candidates of one read beyond which the scan is redone into position-ordered arrays

// ---------------------------------------------------------------------------
// small intrinsics with host stand-ins (emulation only)
// ---------------------------------------------------------------------------
TGSF_HD uint32_t popc32(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_popcount(x);
#else
    return (uint32_t)__builtin_popcount(x);
#endif
}
// sum_i a.byte[i] * b.byte[i] + c
TGSF_HD uint32_t udot4(uint32_t a, uint32_t b, uint32_t c) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_udot4(a, b, c, false);
#else
    uint32_t s = c;
    for (int i = 0; i < 4; i++) s += ((a >> (8 * i)) & 0xFF) * ((b >> (8 * i)) & 0xFF);
    return s;
#endif
}
// bytes [sh, sh+4) of the 8-byte value hi:lo, sh in 0..3
TGSF_HD uint32_t alignbyte(uint32_t hi, uint32_t lo, uint32_t sh) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbyte(hi, lo, sh);
#else
    uint64_t v = ((uint64_t)hi << 32) | lo;
    return (uint32_t)(v >> (8 * sh));
#endif
}

// bits [sh, sh+32) of the 64-bit value hi:lo, sh in 0..31
TGSF_HD uint32_t alignbit(uint32_t hi, uint32_t lo, uint32_t sh) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbit(hi, lo, sh);
#else
    uint64_t v = ((uint64_t)hi << 32) | lo;
    return (uint32_t)(v >> (sh & 31u));
#endif
}
// byte i of the result = byte sel.byte[i] of the 8-byte value hi:lo (selectors 0..7 only)
TGSF_HD uint32_t perm_bytes(uint32_t hi, uint32_t lo, uint32_t sel) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_perm(hi, lo, sel);
#else
    uint64_t v = ((uint64_t)hi << 32) | lo;
    uint32_t r = 0;
    for (int i = 0; i < 4; i++) r |= (uint32_t)((v >> (8 * ((sel >> (8 * i)) & 7u))) & 0xFF) << (8 * i);
    return r;
#endif
}

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
TGSF_HD uint32_t popc64(uint64_t x) { return (uint32_t)__builtin_popcountll(x); }
TGSF_HD void hot_init(Hot& s, int Q) {
    s.p = (Q >= 64) ? ~0ull : (~0ull << (64 - Q));
    s.m = 0ull;
}
// any boolean function of three words in one instruction: bit i of the result is TT[a_i b_i c_i],
// TT = f(0xF0, 0xCC, 0xAA) (v_bitop3_b32)
template <int TT>
TGSF_HD uint32_t bitop3(uint32_t a, uint32_t b, uint32_t c) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_bitop3_b32(a, b, c, TT);
#else
    uint32_t r = 0;
    for (int i = 0; i < 32; i++) {
        const int idx = (int)(((a >> i) & 1u) << 2 | ((b >> i) & 1u) << 1 | ((c >> i) & 1u));
        r |= (uint32_t)((TT >> idx) & 1) << i;
    }
    return r;
#endif
}
TGSF_HD void hot_step(Hot& s, uint64_t Eq) {
    const uint64_t Pv = s.p, Mv = s.m;
    const uint64_t t = Eq & Pv;                          // 2 v_and
    const uint64_t sum = t + Pv;                         // 1 v_lshl_add_u64
    const uint64_t u = sum | Pv | Eq;                    // 2 v_or3           (= Xh | Pv)
    uint64_t Ph = Mv | ~u;                               // 2 v_bfi
    uint64_t Mh = (sum & t) | (~sum & Pv);               // 2 v_bfi           (= Pv & Xh)
    Ph <<= 1; Mh <<= 1;                                  // 2 v_lshlrev_b64 (operands are register pairs)
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(Ph));                         // keep the shifted pair as is (else its high half is re-derived with a v_alignbit)
#endif
    const uint64_t x = Eq | Mv | Ph;                     // 2 v_or3           (= Xv | Ph')
    s.p = Mh | ~x;                                       // 2 v_bfi
    // Mv' = Ph' & (Eq | Mv) is one 3-input function per half; its result only feeds bitwise ops, so it
    // does not need to sit in a register pair (the builtin's results do not)
    const uint32_t nml = bitop3<0xE0>((uint32_t)Ph, (uint32_t)Eq, (uint32_t)Mv);
    const uint32_t nmh = bitop3<0xE0>((uint32_t)(Ph >> 32), (uint32_t)(Eq >> 32), (uint32_t)(Mv >> 32));
    s.m = ((uint64_t)nmh << 32) | nml;                   // 2 v_bitop3
}
// popcount(x) + acc as two accumulating v_bcnt_u32_b32 (the compiler adds acc separately when popcount(x)
// has another use)
TGSF_HD int popc64_acc(uint64_t x, int acc) {
#if defined(__HIP_DEVICE_COMPILE__)
    int r;
    asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"((uint32_t)x), "v"(acc));
    asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(r) : "v"((uint32_t)(x >> 32)));
    return r;
#else
    return (int)popc64(x) + acc;
#endif
}
TGSF_HD int hot_score(const Hot& s) { return (int)popc64(s.p) - (int)popc64(s.m); }
// bottom-row value <= lim  <=>  popcount(Pv) <= popcount(Mv) + lim
TGSF_HD bool hot_within(const Hot& s, int lim) {
#if defined(__HIP_DEVICE_COMPILE__)
    // two accumulating v_bcnt_u32_b32 per side (the builtin: inline assembly makes the compiler pad the chain with s_nop)
    const uint32_t p = (uint32_t)__builtin_popcount((uint32_t)(s.p >> 32)) + (uint32_t)__builtin_popcount((uint32_t)s.p);
    uint32_t m = (uint32_t)__builtin_popcount((uint32_t)s.m) + (uint32_t)lim;
    asm volatile("" : "+v"(m));                          // (keeps the sum from being regrouped into two plain counts and a three-input add)
    m = (uint32_t)__builtin_popcount((uint32_t)(s.m >> 32)) + m;
    return (int)p <= (int)m;
#else
    return popc64_acc(s.p, 0) <= popc64_acc(s.m, lim);
#endif
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
#if defined(__HIP_DEVICE_COMPILE__)
    // every three-input function spelled out (the compiler forms v_or3 / v_bfi from 64-bit expressions, not from these)
    const uint32_t u = bitop3<0xFE>(sum, Pv, Eq);        // sum | Pv | Eq                 (= Xh | Pv)
    uint32_t Ph = bitop3<0xF3>(Mv, u, u);                // Mv | ~u
    uint32_t Mh = bitop3<0xD0>(Pv, Eq, sum);             // Pv & (Eq | ~sum)              (= Pv & Xh)
    Ph <<= 1; Mh <<= 1;                                  // 2 v_lshlrev_b32
    const uint32_t x = bitop3<0xFE>(Eq, Mv, Ph);         // Eq | Mv | Ph'                 (= Xv | Ph')
    s.p = bitop3<0xF3>(Mh, x, x);                        // Mh' | ~x
    s.m = bitop3<0xE0>(Ph, Eq, Mv);                      // Ph' & (Eq | Mv)
#else
    const uint32_t u = sum | Pv | Eq;
    uint32_t Ph = Mv | ~u;
    uint32_t Mh = Pv & (Eq | ~sum);
    Ph <<= 1; Mh <<= 1;
    const uint32_t x = Eq | Mv | Ph;
    s.p = Mh | ~x;
    s.m = Ph & (Eq | Mv);
#endif
}
TGSF_HD int hot_score(const Hot32& s) { return (int)popc32(s.p) - (int)popc32(s.m); }
TGSF_HD bool hot_within(const Hot32& s, int lim) {
#if defined(__HIP_DEVICE_COMPILE__)
    int r;
    asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"(s.m), "v"(lim));
    return (int)popc32(s.p) <= r;
#else
    return (int)popc32(s.p) <= (int)popc32(s.m) + lim;
#endif
}
// the 32-bit Eq row from the 64-bit top-aligned one: its high half (bits below the adapter are wildcards in both)
TGSF_HD uint32_t hot_eq(const Hot32&, uint64_t top) { return (uint32_t)(top >> 32); }

// ---------------------------------------------------------------------------
// QC columns for 4 bases at a time (SWAR).  s: 4 sequence bytes, q: 4 quality
// bytes (already masked to the valid ones; invalid bytes are 0 in both).
// cnt[c] += #bases of class c; qs[c] += 128 * sum of their quality bytes
// (c: A,T,G,C -- src/TGSFilter.cpp:1462-1474, case-insensitive); qs[4] += sum
// of all 4 quality bytes.  Quality bytes must be < 128.
// ---------------------------------------------------------------------------
TGSF_HD void qc_accum4(uint32_t s, uint32_t q, uint32_t* cnt /*[4]*/, uint32_t* qs /*[5]*/) {
    const uint32_t x7 = s & 0x5F5F5F5Fu;         // fold lower case onto upper case, drop bit 7
    const uint32_t K[4] = {0x41414141u, 0x54545454u, 0x47474747u, 0x43434343u};   // A T G C
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const uint32_t nz = ((x7 ^ K[c]) + 0x7F7F7F7Fu) | s;   // bit 7 of each byte: byte differs from K[c]
        const uint32_t m = ~nz & 0x80808080u;                   // 0x80 where the base is of class c
        cnt[c] += popc32(m);
        qs[c] = udot4(q, m, qs[c]);
    }
    qs[4] = udot4(q, 0x01010101u, qs[4]);
}

#if defined(__HIP_DEVICE_COMPILE__)
// The same tallies with the class constants held in VGPRs, so that (x7 ^ K) + 0x7F7F7F7F is one
// v_xad_u32 (a VOP3 instruction reads one scalar at most): 4 instructions per class and dword.
struct QcConsts { uint32_t k[4]; };
__device__ __forceinline__ QcConsts qc_consts() {
    QcConsts c;
    asm volatile("v_mov_b32 %0, 0x41414141" : "=v"(c.k[0]));
    asm volatile("v_mov_b32 %0, 0x54545454" : "=v"(c.k[1]));
    asm volatile("v_mov_b32 %0, 0x47474747" : "=v"(c.k[2]));
    asm volatile("v_mov_b32 %0, 0x43434343" : "=v"(c.k[3]));
    return c;
}
__device__ __forceinline__ void qc_accum4(const QcConsts& kc, uint32_t s, uint32_t q, uint32_t* cnt, uint32_t* qs) {
    const uint32_t x7 = s & 0x5F5F5F5Fu;
    const uint32_t c7f = 0x7F7F7F7Fu;
#pragma unroll
    for (int c = 0; c < 4; c++) {
        uint32_t t;
        asm("v_xad_u32 %0, %1, %2, %3" : "=v"(t) : "v"(x7), "v"(kc.k[c]), "s"(c7f));
        const uint32_t m = ~(t | s) & 0x80808080u;
        cnt[c] += popc32(m);
        qs[c] = udot4(q, m, qs[c]);
    }
    qs[4] = udot4(q, 0x01010101u, qs[4]);
}
#else
struct QcConsts { int unused; };
TGSF_HD QcConsts qc_consts() { return QcConsts{0}; }
TGSF_HD void qc_accum4(const QcConsts&, uint32_t s, uint32_t q, uint32_t* cnt, uint32_t* qs) { qc_accum4(s, q, cnt, qs); }
#endif

// mean-quality gate: src/TGSFilter.cpp:1478 (double(sumQ)/len), :1947 (compare
// against float thresholds promoted to double).  sum is the uint64 accumulator.
TGSF_HD double mean_q(uint64_t sum, uint32_t len) { return (double)sum / (double)len; }
TGSF_HD bool q_fail(double mq, float min_q, float max_q) {
    return (mq < (double)min_q) || (mq > (double)max_q);
}

}  // namespace tgsf
