// repeat_probe.hip -- formulations of the repeat gate's distinct-k-mer count side by side on random fragments of the
// bench's shape (lognormal lengths, mean 45 kb): what round 1 shipped, the steps to k_repeat (k <= 12) and to
// k_repeat_keys (k 13..31), and what was tried beside them.  Every variant's counts are checked against the round-1
// kernel's and, in small runs (<= 8192 fragments), against a sort on the host.
//   repeat_probe [fragments = 65536] [1 = longest first | 2 = none longer than a window]
//   hipcc -O3 --offload-arch=gfx950 tools/repeat_probe.hip -o tools/repeat_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
#include <random>
#include <algorithm>
#include <type_traits>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)


__device__ __forceinline__ uint32_t base_code(uint32_t c) {
    const uint32_t v = c - 0x41u;
    const uint32_t valid = (v < 20u ? 1u : 0u) & (0x80045u >> (v & 31u));
    const uint32_t x = (c >> 1) & 3u;
    return (x ^ (x >> 1)) & (0u - valid);
}
__device__ __forceinline__ uint32_t alignbyte(uint32_t hi, uint32_t lo, uint32_t s) { return __builtin_amdgcn_alignbyte(hi, lo, s); }

// fill codes[] for the window [w0, w0+wn) of the fragment
__device__ __forceinline__ void load_codes(uint32_t* codes, const uint8_t* seq, int L, int w0, int wn, int tid, int nthr)
{
    for (int g = tid; g * 16 < wn; g += nthr) {
        const int b0 = w0 + g * 16;
        const int nb = L - b0 < 16 ? L - b0 : 16;
        const uintptr_t a = (uintptr_t)(seq + b0);
        const uint32_t* w32 = reinterpret_cast<const uint32_t*>(a & ~(uintptr_t)3);
        const uint32_t bs = (uint32_t)(a & 3u);
        uint32_t wv[5];
#pragma unroll
        for (int q = 0; q < 5; q++) wv[q] = (q < 4 || bs) ? w32[q] : 0u;
        uint32_t word = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint32_t d = alignbyte(wv[q + 1], wv[q], bs);
#pragma unroll
            for (int r = 0; r < 4; r++) word |= base_code((d >> (8 * r)) & 0xFFu) << (2 * (4 * q + r));
        }
        if (nb < 16) word &= (1u << (2 * nb)) - 1u;
        codes[g] = word;
    }
}

// MODE 0: bitmap partition of 2^PART bits, divergent no-return ds_or (the shipped kernel)
// MODE 1: byte map of 2^PART entries, plain byte stores
// MODE 2: bitmap, lanes run ahead to their next owned k-mer so that every ds_or is issued by whole waves
// MODE 3: bitmap, interleaved ownership: lane handles k-mers i = tid, tid + nthr, ... (k-mer built from k code reads)
template <int MODE, int PART, int kRepWin>
__global__ void k_rep(const uint8_t* seqs, const uint64_t* foff, const uint32_t* flen, uint32_t nf, int k, uint32_t* distinct)
{
    constexpr uint32_t kSetBytes = MODE == 1 ? (1u << PART) : (1u << PART) / 8;
    __shared__ uint32_t bm[kSetBytes / 4];
    __shared__ uint32_t codes[kRepWin / 16 + 4];
    __shared__ uint32_t distinct_s;
    const uint32_t space_log2 = 2u * (uint32_t)k;
    const uint32_t part_log2 = space_log2 < (uint32_t)PART ? space_log2 : (uint32_t)PART;
    const uint32_t passes = 1u << (space_log2 - part_log2);
    const uint32_t part_words = MODE == 1 ? (1u << part_log2) / 4 : ((1u << part_log2) + 31u) / 32u;
    const uint32_t kmask = (1u << space_log2) - 1u;
    const uint32_t nthr = blockDim.x, tid = threadIdx.x;
    for (uint32_t f = blockIdx.x; f < nf; f += gridDim.x) {
        const int L = (int)flen[f];
        const int total = L - k + 1;
        const uint8_t* seq = seqs + foff[f];
        if (tid == 0) distinct_s = 0;
        uint32_t mine = 0;
        for (uint32_t pass = 0; pass < passes; pass++) {
            __syncthreads();
            for (uint32_t w = tid; w < part_words; w += nthr) bm[w] = 0;
            for (int w0 = 0; w0 < (total > 0 ? total : 0); w0 += kRepWin - (k - 1)) {
                int wn = L - w0;
                if (wn > kRepWin) wn = kRepWin;
                const int nk = wn - k + 1;
                __syncthreads();
                if (pass == 0 || total > kRepWin - (k - 1)) load_codes(codes, seq, L, w0, wn, (int)tid, (int)nthr);
                __syncthreads();
                const int per = (nk + (int)nthr - 1) / (int)nthr;
                const int i0 = (int)tid * per;
                int i1 = i0 + per;
                if (i1 > nk) i1 = nk;
                auto code_at = [&](int i) { return (codes[i >> 4] >> (2 * (i & 15))) & 3u; };
                if (MODE == 0 || MODE == 1) {
                    if (i0 < i1) {
                        uint32_t km = 0;
                        for (int j = i0; j < i0 + k - 1; j++) km = (km << 2) | code_at(j);
                        auto feed = [&](uint32_t c) {
                            km = ((km << 2) | c) & kmask;
                            if ((km >> part_log2) == pass) {
                                const uint32_t idx = km & ((1u << part_log2) - 1u);
                                if (MODE == 0) atomicOr(&bm[idx >> 5], 1u << (idx & 31u));
                                else reinterpret_cast<uint8_t*>(bm)[idx] = 1;
                            }
                        };
                        int pos = i0 + k - 1;
                        const int pend = i1 + k - 1;
                        while (pos < pend && (pos & 15)) { feed(code_at(pos)); pos++; }
                        while (pos + 16 <= pend) {
                            uint32_t w = codes[pos >> 4];
#pragma unroll
                            for (int q = 0; q < 16; q++) { feed(w & 3u); w >>= 2; }
                            pos += 16;
                        }
                        while (pos < pend) { feed(code_at(pos)); pos++; }
                    }
                } else if (MODE == 2) {
                    uint32_t km = 0;
                    int pos = i0 + k - 1;
                    const int pend = i0 < i1 ? i1 + k - 1 : pos;
                    if (i0 < i1) for (int j = i0; j < i0 + k - 1; j++) km = (km << 2) | code_at(j);
                    uint32_t w = pos < pend ? codes[pos >> 4] >> (2 * (pos & 15)) : 0u;
                    for (;;) {
                        bool have = false;
                        while (pos < pend) {
                            km = ((km << 2) | (w & 3u)) & kmask;
                            pos++;
                            w >>= 2;
                            if ((pos & 15) == 0) w = codes[pos >> 4];
                            if ((km >> part_log2) == pass) { have = true; break; }
                        }
                        if (!have) break;
                        const uint32_t idx = km & ((1u << part_log2) - 1u);
                        atomicOr(&bm[idx >> 5], 1u << (idx & 31u));
                    }
                } else {
                    // k-mer i from the two code words around it: bases i .. i+k-1 (k <= 13 fits in 64 bits of codes)
                    for (int i = (int)tid; i < nk; i += (int)nthr) {
                        const uint64_t two = (uint64_t)codes[i >> 4] | ((uint64_t)codes[(i >> 4) + 1] << 32);
                        const uint32_t fwd = (uint32_t)(two >> (2 * (i & 15)));     // base i at bits 0..1, i+1 at 2..3: reversed order
                        // reverse the 2-bit groups of the low 2k bits: k-mer has base i as the MOST significant pair
                        uint32_t r = fwd;
                        r = ((r >> 2) & 0x33333333u) | ((r & 0x33333333u) << 2);
                        r = ((r >> 4) & 0x0F0F0F0Fu) | ((r & 0x0F0F0F0Fu) << 4);
                        r = __builtin_bswap32(r);
                        const uint32_t km = (r >> (32 - space_log2)) & kmask;
                        if ((km >> part_log2) == pass) {
                            const uint32_t idx = km & ((1u << part_log2) - 1u);
                            atomicOr(&bm[idx >> 5], 1u << (idx & 31u));
                        }
                    }
                }
            }
            __syncthreads();
            if (MODE == 1) { for (uint32_t w = tid; w < part_words; w += nthr) mine += __popc(bm[w]); }
            else for (uint32_t w = tid; w < part_words; w += nthr) mine += __popc(bm[w]);
        }
        if (mine) atomicAdd(&distinct_s, mine);
        __syncthreads();
        if (tid == 0) distinct[f] = distinct_s;
        __syncthreads();
    }
}


// ---- second family: codes packed MSB-first (base j of a word at bits 31-2j, 30-2j), one lane per 16-base word, the k-mer at
// position 16g+j is a bit-field of the 64-bit pair {codes[g], codes[g+1]}; a pass owns the k-mers whose first P bases spell the
// pass number, found with a match mask per word; "distinct" counted from the value the ds_or returns; the set is cleared either
// whole or by running the same k-mers again.
__device__ __forceinline__ void load_codes_be(uint32_t* codes, const uint8_t* seq, int L, int w0, int wn, int tid, int nthr)
{
    const int nw = (wn + 15) / 16;
    for (int g = tid; g <= nw; g += nthr) {
        if (g == nw) { codes[g] = 0; continue; }
        const int b0 = w0 + g * 16;
        const int nb = L - b0 < 16 ? L - b0 : 16;
        const uintptr_t a = (uintptr_t)(seq + b0);
        const uint32_t* w32 = reinterpret_cast<const uint32_t*>(a & ~(uintptr_t)3);
        const uint32_t bs = (uint32_t)(a & 3u);
        uint32_t wv[5];
#pragma unroll
        for (int q = 0; q < 5; q++) wv[q] = (q < 4 || bs) ? w32[q] : 0u;
        uint32_t word = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint32_t d = alignbyte(wv[q + 1], wv[q], bs);
#pragma unroll
            for (int r = 0; r < 4; r++) word |= base_code((d >> (8 * r)) & 0xFFu) << (30 - 2 * (4 * q + r));
        }
        if (nb < 16) word &= ~(0xFFFFFFFFu >> (2 * nb));
        codes[g] = word;
    }
}
__device__ __forceinline__ uint32_t eq_mask(uint32_t w, uint32_t c) {       // bit 2s set where the pair at bits 2s+1,2s equals c
    const uint32_t x = w ^ (c * 0x55555555u);
    return ~(x | (x >> 1)) & 0x55555555u;
}
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    return v;
}

template <int LOOP, int kRepWin>      // LOOP 0: all 16 positions of a word, unowned ones OR a zero; 1: walk the set bits of the match mask
__global__ void __launch_bounds__(1024) k_rep2(const uint8_t* seqs, const uint64_t* foff, const uint32_t* flen, uint32_t nf, int k, uint32_t* distinct, uint32_t undo_below)
{
    __shared__ uint32_t bm[32768];
    __shared__ uint32_t codes[kRepWin / 16 + 4];
    __shared__ uint32_t distinct_s;
    const uint32_t space_log2 = 2u * (uint32_t)k;
    const uint32_t part_log2 = space_log2 < 20u ? space_log2 : 20u;
    const int P = (int)(space_log2 - part_log2) / 2;                  // prefix bases that select the pass
    const uint32_t passes = 1u << (2 * P);
    const uint32_t part_words = ((1u << part_log2) + 31u) / 32u;
    const int nthr = (int)blockDim.x, tid = (int)threadIdx.x;
    for (uint32_t w = tid; w < part_words; w += nthr) bm[w] = 0;
    for (uint32_t f = blockIdx.x; f < nf; f += gridDim.x) {
        const int L = (int)flen[f];
        const int total = L - k + 1;
        const uint8_t* seq = seqs + foff[f];
        if (tid == 0) distinct_s = 0;
        uint32_t mine = 0;
        const bool one_window = total <= kRepWin - (k - 1);
        const bool undo = one_window && (uint32_t)(total > 0 ? total : 0) / passes < undo_below;
        for (uint32_t pass = 0; pass < passes; pass++) {
            for (int w0 = 0; w0 < (total > 0 ? total : 0); w0 += kRepWin - (k - 1)) {
                int wn = L - w0;
                if (wn > kRepWin) wn = kRepWin;
                const int nk = wn - k + 1;
                if (pass == 0 || !one_window) {
                    __syncthreads();
                    load_codes_be(codes, seq, L, w0, wn, tid, nthr);
                }
                __syncthreads();
                for (int phase = 0; phase < (undo ? 2 : 1); phase++) {
                    if (phase) __syncthreads();
                    for (int g = tid; g * 16 < nk; g += nthr) {
                        const uint32_t hi = codes[g], lo = codes[g + 1];
                        const uint64_t win = ((uint64_t)hi << 32) | lo;
                        const int v = nk - g * 16;                         // valid positions in this word
                        if (LOOP == 0) {
#pragma unroll
                            for (int j = 0; j < 16; j++) {
                                const uint32_t t = j ? __builtin_amdgcn_alignbit(hi, lo, 32 - 2 * j) : hi;   // k-mer at bit 31 downwards
                                const uint32_t idx = (t << (2 * P)) >> (32 - part_log2);
                                uint32_t bit = 1u << (idx & 31u);
                                if (P && (t >> (32 - 2 * P)) != pass) bit = 0;
                                if (j >= v) bit = 0;
                                if (phase == 0) { const uint32_t old = atomicOr(&bm[idx >> 5], bit); mine += (bit & ~old) ? 1u : 0u; }
                                else if (bit) bm[idx >> 5] = 0;
                            }
                        } else {
                            uint32_t m;
                            if (P == 0) m = 0x55555555u;
                            else {
                                m = eq_mask(hi, (pass >> (2 * (P - 1))) & 3u);
                                if (P >= 2) m &= __builtin_amdgcn_alignbit(eq_mask(hi, (pass >> (2 * (P - 2))) & 3u), eq_mask(lo, (pass >> (2 * (P - 2))) & 3u), 30);
                                if (P >= 3) m &= __builtin_amdgcn_alignbit(eq_mask(hi, pass & 3u), eq_mask(lo, pass & 3u), 28);
                            }
                            if (v < 16) m &= ~(0xFFFFFFFFu >> (2 * v));
                            while (m) {
                                const int b = __builtin_ctz(m);                // even; position j = (30 - b) / 2
                                m &= m - 1;
                                const uint32_t t = (uint32_t)((win << (30 - b + 2 * P)) >> 32);
                                const uint32_t idx = t >> (32 - part_log2);
                                const uint32_t bit = 1u << (idx & 31u);
                                if (phase == 0) { const uint32_t old = atomicOr(&bm[idx >> 5], bit); mine += (bit & ~old) ? 1u : 0u; }
                                else bm[idx >> 5] = 0;
                            }
                        }
                    }
                }
            }
            if (!undo) {
                __syncthreads();
                uint4* b4 = reinterpret_cast<uint4*>(bm);
                for (uint32_t w = tid; w < part_words / 4; w += nthr) b4[w] = make_uint4(0, 0, 0, 0);
                if (part_words < 4) for (uint32_t w = tid; w < part_words; w += nthr) bm[w] = 0;
            }
        }
        mine = wave_sum(mine);
        if ((tid & 63) == 0 && mine) atomicAdd(&distinct_s, mine);
        __syncthreads();
        if (tid == 0) distinct[f] = distinct_s;
        __syncthreads();
    }
}

// ---- third family: as the second, plus: the text is read as aligned 16-byte chunks (one per lane, k-mer positions counted from the
// chunk the fragment starts in), the next fragment's first window is fetched into registers while this one is processed, four bases
// are converted at a time, the set bits are counted while the set is cleared (no returning atomics), fragments are dealt out by a counter.
__device__ __forceinline__ uint32_t codes4(uint32_t d)          // 4 text bytes -> 8 bits, first byte's code in bits 7..6
{
    const uint32_t x = (d >> 1) & 0x03030303u;
    uint32_t y = x ^ ((x >> 1) & 0x01010101u);                              // A0 C1 G2 T3 (other bytes: something in 0..3)
    const uint32_t diff = d ^ __builtin_amdgcn_perm(0u, 0x54474341u, y);    // zero byte where the text byte is exactly A C G T
    const uint32_t nz = (((diff & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | diff) >> 7 & 0x01010101u;
    y &= ~(nz | (nz << 1));
    return ((y << 6) | (y >> 4) | (y >> 14) | (y >> 24)) & 0xFFu;
}
__device__ __forceinline__ uint32_t codes16(uint4 r) { return (codes4(r.x) << 24) | (codes4(r.y) << 16) | (codes4(r.z) << 8) | codes4(r.w); }

template <int SLOTS>
__global__ void __launch_bounds__(1024) k_rep3(const uint8_t* seqs, const uint64_t* foff, const uint32_t* flen, uint32_t nf, int k, uint32_t* distinct, uint32_t* work_ctr)
{
    constexpr int NT = 1024;
    constexpr int W = SLOTS * NT;                                    // words of a window
    __shared__ uint4 bm4[8192];
    __shared__ uint32_t codes[W + 4];
    __shared__ uint32_t distinct_s, next_s;
    uint32_t* bm = reinterpret_cast<uint32_t*>(bm4);
    const uint32_t space_log2 = 2u * (uint32_t)k;
    const uint32_t part_log2 = space_log2 < 20u ? space_log2 : 20u;
    const int P = (int)(space_log2 - part_log2) / 2;
    const uint32_t passes = 1u << (2 * P);
    const uint32_t part_q = (((1u << part_log2) + 31u) / 32u + 3u) / 4u;   // uint4s of a partition
    const int tid = (int)threadIdx.x;
    for (uint32_t w = tid; w < 8192; w += NT) bm4[w] = make_uint4(0, 0, 0, 0);

    uint4 raw[SLOTS];
    auto chunks_of = [&](uint32_t f, const uint4*& base, int& a, int& L) {
        const uint8_t* s = seqs + foff[f];
        a = (int)((uintptr_t)s & 15u);
        base = reinterpret_cast<const uint4*>(s - a);
        L = (int)flen[f];
    };
    auto prefetch = [&](uint32_t f) {
        if (f >= nf) return;
        const uint4* base; int a, L;
        chunks_of(f, base, a, L);
        const int words = (a + L + 15) / 16;
#pragma unroll
        for (int sl = 0; sl < SLOTS; sl++) { const int g = tid + sl * NT; if (g < words) raw[sl] = base[g]; }
    };
    uint32_t f = blockIdx.x;
    prefetch(f);
    while (f < nf) {
        const uint4* base; int a, L;
        chunks_of(f, base, a, L);
        const int total = L - k + 1;
        const int words = (a + L + 15) / 16;                             // chunks holding the fragment
        const int kwords = total > 0 ? (a + total + 15) / 16 : 0;        // chunks in which a k-mer starts
        const bool one_window = words <= W;
        if (tid == 0) { distinct_s = 0; next_s = gridDim.x + atomicAdd(work_ctr, 1u); }
        __syncthreads();                                                 // (the previous fragment's readers of codes are done)
#pragma unroll
        for (int sl = 0; sl < SLOTS; sl++) { const int g = tid + sl * NT; if (g < words) codes[g] = codes16(raw[sl]); }
        __syncthreads();
        const uint32_t fnext = next_s;
        prefetch(fnext);
        uint32_t mine = 0;
        for (uint32_t pass = 0; pass < passes; pass++) {
            for (int wb = 0; wb < kwords; wb += W - 1) {                 // window = words [wb, wb + W)
                if (!(one_window && true) && !(pass == 0 && wb == 0)) {
                    __syncthreads();
                    for (int g = tid; g < W && wb + g < words; g += NT) codes[g] = codes16(base[wb + g]);
                    __syncthreads();
                }
                const int gend = (wb + W >= words) ? kwords - wb : W - 1;  // k-mers start in local words [0, gend)
                for (int g = tid; g < gend; g += NT) {
                    const uint32_t hi = codes[g], lo = codes[g + 1];
                    const int first = a - 16 * (wb + g);                  // positions before the fragment (word 0 only)
                    const int v = a + total - 16 * (wb + g);              // positions before the end of the k-mers
                    if (P == 0 && first <= 0 && v >= 16) {
#pragma unroll
                        for (int j = 0; j < 16; j++) {
                            const uint32_t t = j ? __builtin_amdgcn_alignbit(hi, lo, 32 - 2 * j) : hi;
                            const uint32_t idx = t >> (32 - part_log2);
                            atomicOr(&bm[idx >> 5], 1u << (idx & 31u));
                        }
                    } else {
                        uint32_t m;
                        if (P == 0) m = 0x55555555u;
                        else {
                            m = eq_mask(hi, (pass >> (2 * (P - 1))) & 3u);
                            if (P >= 2) m &= __builtin_amdgcn_alignbit(eq_mask(hi, (pass >> (2 * (P - 2))) & 3u), eq_mask(lo, (pass >> (2 * (P - 2))) & 3u), 30);
                            if (P >= 3) m &= __builtin_amdgcn_alignbit(eq_mask(hi, pass & 3u), eq_mask(lo, pass & 3u), 28);
                        }
                        if (first > 0) m &= 0xFFFFFFFFu >> (2 * first);
                        if (v < 16) m &= ~(0xFFFFFFFFu >> (2 * v));
                        const uint64_t win = ((uint64_t)hi << 32) | lo;
                        const int c0 = 30 + 2 * P;
                        while (m) {
                            const int b = __builtin_ctz(m);
                            m &= m - 1;
                            const uint32_t t = (uint32_t)((win << (c0 - b)) >> 32);
                            const uint32_t idx = t >> (32 - part_log2);
                            atomicOr(&bm[idx >> 5], 1u << (idx & 31u));
                        }
                    }
                }
                if (wb + W >= words) break;
            }
            __syncthreads();
            if (kwords > 0) for (uint32_t w = tid; w < part_q; w += NT) {
                const uint4 q = bm4[w];
                mine += __popc(q.x) + __popc(q.y) + __popc(q.z) + __popc(q.w);
                bm4[w] = make_uint4(0, 0, 0, 0);
            }
            __syncthreads();
        }
        mine = wave_sum(mine);
        if ((tid & 63) == 0 && mine) atomicAdd(&distinct_s, mine);
        __syncthreads();
        if (tid == 0) distinct[f] = distinct_s;
        f = fnext;
    }
}

// ---- fourth family (any k up to 31): the distinct k-mers of a pass are counted exactly by inserting the keys into an
// open-addressing table in LDS (32-bit keys for k <= 16, 64-bit above), a pass owning the k-mers that start with its PB
// leading bases; PB is the smallest for which a pass's share fits the table at a low load, and one more whenever a
// probe sequence grows long (skewed composition): the fragment then starts over.
template <bool KEY64>
__global__ void __launch_bounds__(1024) k_rep5(const uint8_t* seqs, const uint64_t* foff, const uint32_t* flen, uint32_t nf, int k, uint32_t* distinct, uint32_t* work_ctr)
{
    constexpr int NT = 1024, SLOTS = 6, W = SLOTS * NT;
    constexpr int OV = KEY64 ? 2 : 1;                                  // chunks shared by consecutive windows
    constexpr uint32_t S = KEY64 ? 16384u : 32768u;                     // slots of the whole table (128 KB)
    typedef typename std::conditional<KEY64, unsigned long long, uint32_t>::type key_t;
    __shared__ uint4 tab4[8192];
    __shared__ uint32_t codes[W + 8];
    __shared__ uint32_t distinct_s, next_s, over_s;
    key_t* tab = reinterpret_cast<key_t*>(tab4);
    const key_t EMPTY = ~(key_t)0;
    const int tid = (int)threadIdx.x;
    const uint4 ones4 = make_uint4(~0u, ~0u, ~0u, ~0u);

    uint4 raw[SLOTS];
    auto chunks_of = [&](uint32_t f, const uint4*& base, int& a, int& L) {
        const uint8_t* s = seqs + foff[f];
        a = (int)((uintptr_t)s & 15u);
        base = reinterpret_cast<const uint4*>(s - a);
        L = (int)flen[f];
    };
    auto prefetch = [&](uint32_t f) {
        if (f >= nf) return;
        const uint4* base; int a, L;
        chunks_of(f, base, a, L);
        const int words = (a + L + 15) / 16;
#pragma unroll
        for (int sl = 0; sl < SLOTS; sl++) { const int g = tid + sl * NT; if (g < words) raw[sl] = base[g]; }
    };
    uint32_t f = blockIdx.x;
    prefetch(f);
    while (f < nf) {
        const uint4* base; int a, L;
        chunks_of(f, base, a, L);
        const int total = L - k + 1;
        const int words = (a + L + 15) / 16;
        const int kwords = total > 0 ? (a + total + 15) / 16 : 0;
        const bool one_window = words <= W;
        if (tid == 0) { distinct_s = 0; over_s = 0; next_s = gridDim.x + atomicAdd(work_ctr, 1u); }
        __syncthreads();
#pragma unroll
        for (int sl = 0; sl < SLOTS; sl++) { const int g = tid + sl * NT; if (g < words) codes[g] = codes16(raw[sl]); }
        if (tid < 8) codes[(words < W ? words : W) + tid] = 0;
        __syncthreads();
        const uint32_t fnext = next_s;
        prefetch(fnext);
        int PB = (!KEY64 && k == 16) ? 1 : 0;                          // (a 32-bit key of all ones is the empty mark)
        uint32_t mine = 0;
        bool first_attempt = true;
        for (;;) {
            // table size for this fragment: a pass's share of the k-mers at load <= 3/8, a power of two
            while (PB < k - 1 && ((uint32_t)(total > 0 ? total : 0) >> (2 * PB)) > S * 3u / 8u) PB++;
            const uint32_t share = (uint32_t)(total > 0 ? total : 0) >> (2 * PB);
            uint32_t slog = 10;
            while ((1u << slog) < S && (3u << slog) / 8u < share) slog++;
            const uint32_t smask = (1u << slog) - 1u;
            const uint32_t tab_q = (uint32_t)((sizeof(key_t) << slog) / 16u);   // uint4s to clear
            const int kb = 2 * (k - PB);                                 // bits of a key
            const uint32_t passes = 1u << (2 * PB);
            mine = 0;
            for (uint32_t w = tid; w < tab_q; w += NT) tab4[w] = ones4;
            for (uint32_t pass = 0; pass < passes && kwords > 0; pass++) {
                for (int wb = 0;; wb += W - OV) {
                    if (!(one_window && first_attempt) && !(pass == 0 && wb == 0 && first_attempt)) {
                        if (!one_window || true) {
                            __syncthreads();
                            for (int g = tid; g < W + 4 && wb + g < words + 4; g += NT) codes[g] = wb + g < words ? codes16(base[wb + g]) : 0u;
                        }
                    }
                    __syncthreads();
                    const bool last = wb + W >= words;
                    const int gend = last ? kwords - wb : W - OV;
                    for (int g = tid; g < gend; g += NT) {
                        const uint32_t w0 = codes[g], w1 = codes[g + 1], w2 = codes[g + 2], w3 = codes[g + 3];
                        const int first = a - 16 * (wb + g);
                        const int v = a + total - 16 * (wb + g);
                        uint32_t m = 0x55555555u;
                        for (int q = 0; q < PB; q++) {
                            const uint32_t c = (pass >> (2 * (PB - 1 - q))) & 3u;
                            const uint32_t e0 = eq_mask(w0, c);
                            m &= q ? __builtin_amdgcn_alignbit(e0, eq_mask(w1, c), 32 - 2 * q) : e0;
                        }
                        if (first > 0) m &= 0xFFFFFFFFu >> (2 * first);
                        if (v < 16) m &= ~(0xFFFFFFFFu >> (2 * v));
                        while (m) {
                            const int b = __builtin_ctz(m);
                            m &= m - 1;
                            const int sh = 30 - b + 2 * PB;                 // bit offset of the key in the chunk sequence
                            const bool up = sh >= 32;
                            const uint32_t A = up ? w1 : w0, B = up ? w2 : w1, C = up ? w3 : w2;
                            const uint32_t sb = (uint32_t)sh & 31u;
                            const uint32_t x0 = sb ? __builtin_amdgcn_alignbit(A, B, 32 - sb) : A;
                            key_t key;
                            uint32_t h;
                            if (KEY64) {
                                const uint32_t x1 = sb ? __builtin_amdgcn_alignbit(B, C, 32 - sb) : B;
                                const unsigned long long k64 = (((unsigned long long)x0 << 32) | x1) >> (64 - kb);
                                key = (key_t)k64;
                                h = ((uint32_t)k64 ^ ((uint32_t)(k64 >> 32) * 0x85EBCA6Bu)) * 0x9E3779B1u;
                            } else {
                                const uint32_t k32 = x0 >> (32 - kb);
                                key = (key_t)k32;
                                h = k32 * 0x9E3779B1u;
                            }
                            uint32_t slot = h >> (32 - slog);
                            for (int probes = 0;; probes++) {
                                const key_t old = atomicCAS(&tab[slot], EMPTY, key);
                                if (old == EMPTY) { mine++; break; }
                                if (old == key) break;
                                if (probes >= 128) { over_s = 1; break; }
                                slot = (slot + 1) & smask;
                            }
                        }
                    }
                    if (last) break;
                }
                __syncthreads();
                if (over_s) break;
                for (uint32_t w = tid; w < tab_q; w += NT) tab4[w] = ones4;
            }
            __syncthreads();
            if (!over_s) break;
            __syncthreads();
            if (tid == 0) over_s = 0;
            PB++;
            first_attempt = false;
            __syncthreads();
        }
        mine = wave_sum(mine);
        if ((tid & 63) == 0 && mine) atomicAdd(&distinct_s, mine);
        __syncthreads();
        if (tid == 0) distinct[f] = distinct_s;
        f = fnext;
    }
}

// ---- fifth family (any k up to 31): two phases per pass.  Phase 1 marks a 2^19-bit map A with a hash of every owned k-mer
// and, when the returned word shows the bit already set, the same bit of a second map B: a hash value marked once stands for
// exactly one k-mer occurrence.  distinct = popcount(A & ~B) + the distinct keys among the occurrences whose hash is in B,
// which phase 2 counts exactly by inserting only those (a few percent of a random fragment) into a small table of full keys.
template <bool KEY64, bool FAST>
__global__ void __launch_bounds__(1024) k_rep6(const uint8_t* seqs, const uint64_t* foff, const uint32_t* flen, uint32_t nf, int k, uint32_t* distinct, uint32_t* work_ctr, uint32_t share_max)
{
    constexpr int NT = 1024, SLOTS = 6, W = SLOTS * NT;
    constexpr int OV = KEY64 ? 2 : 1;
    constexpr uint32_t TLOG = KEY64 ? 13u : 14u;                        // table slots (64 KB), log2
    typedef typename std::conditional<KEY64, unsigned long long, uint32_t>::type key_t;
    __shared__ uint4 A4[4096];
    __shared__ uint4 B4[4096];
    __shared__ uint32_t codes[W + 8];
    __shared__ uint32_t distinct_s, next_s, over_s;
    uint32_t* A = reinterpret_cast<uint32_t*>(A4);
    uint32_t* Bm = reinterpret_cast<uint32_t*>(B4);
    key_t* tab = reinterpret_cast<key_t*>(A4);
    const key_t EMPTY = ~(key_t)0;
    const int tid = (int)threadIdx.x;
    const uint4 ones4 = make_uint4(~0u, ~0u, ~0u, ~0u), zero4 = make_uint4(0, 0, 0, 0);
    for (uint32_t w = tid; w < 4096; w += NT) { A4[w] = zero4; B4[w] = zero4; }

    uint4 raw[SLOTS];
    auto chunks_of = [&](uint32_t f, const uint4*& base, int& a, int& L) {
        const uint8_t* s = seqs + foff[f];
        a = (int)((uintptr_t)s & 15u);
        base = reinterpret_cast<const uint4*>(s - a);
        L = (int)flen[f];
    };
    auto prefetch = [&](uint32_t f) {
        if (f >= nf) return;
        const uint4* base; int a, L;
        chunks_of(f, base, a, L);
        const int words = (a + L + 15) / 16;
#pragma unroll
        for (int sl = 0; sl < SLOTS; sl++) { const int g = tid + sl * NT; if (g < words) raw[sl] = base[g]; }
    };
    uint32_t f = blockIdx.x;
    prefetch(f);
    while (f < nf) {
        const uint4* base; int a, L;
        chunks_of(f, base, a, L);
        const int total = L - k + 1;
        const int words = (a + L + 15) / 16;
        const int kwords = total > 0 ? (a + total + 15) / 16 : 0;
        const bool one_window = words <= W;
        if (tid == 0) { distinct_s = 0; over_s = 0; next_s = gridDim.x + atomicAdd(work_ctr, 1u); }
        __syncthreads();
#pragma unroll
        for (int sl = 0; sl < SLOTS; sl++) { const int g = tid + sl * NT; if (g < words) codes[g] = codes16(raw[sl]); }
        if (tid < 8) codes[(words < W ? words : W) + tid] = 0;
        __syncthreads();
        const uint32_t fnext = next_s;
        prefetch(fnext);
        int PB = (!KEY64 && k == 16) ? 1 : 0;
        uint32_t mine = 0;
        bool in_lds = true;                                             // codes[] holds window 0 of this fragment
        for (;;) {
            while (PB < k - 1 && ((uint32_t)(total > 0 ? total : 0) >> (2 * PB)) > share_max) PB++;
            const int kb = 2 * (k - PB);
            const uint32_t passes = 1u << (2 * PB);
            mine = 0;
            for (uint32_t pass = 0; pass < passes && kwords > 0; pass++) {
                auto scan = [&](auto phase_tag) {
                    constexpr int phase = decltype(phase_tag)::value;
                    for (int wb = 0;; wb += W - OV) {
                        if (!(one_window && in_lds)) {
                            __syncthreads();
                            for (int g = tid; g < W + 4 && wb + g < words + 4; g += NT) codes[g] = wb + g < words ? codes16(base[wb + g]) : 0u;
                            in_lds = one_window;
                        }
                        __syncthreads();
                        const bool last = wb + W >= words;
                        const int gend = last ? kwords - wb : W - OV;
                        for (int g = tid; g < gend; g += NT) {
                            const uint32_t w0 = codes[g], w1 = codes[g + 1], w2 = codes[g + 2], w3 = codes[g + 3];
                            const int first = a - 16 * (wb + g);
                            const int v = a + total - 16 * (wb + g);
                            uint32_t m = 0x55555555u;
                            if (FAST && PB == 0 && first <= 0 && v >= 16) {
                                // all sixteen starts of the chunk, constant shifts; the rare table insertions are left to the walk below
                                uint32_t hbv[16];
                                uint32_t oldv[16];
#pragma unroll
                                for (int j = 0; j < 16; j++) {
                                    const uint32_t x0 = j ? __builtin_amdgcn_alignbit(w0, w1, 32 - 2 * j) : w0;
                                    uint32_t h;
                                    if (KEY64) {
                                        const uint32_t x1 = j ? __builtin_amdgcn_alignbit(w1, w2, 32 - 2 * j) : w1;
                                        const unsigned long long k64 = (((unsigned long long)x0 << 32) | x1) >> (64 - kb);
                                        h = ((uint32_t)k64 ^ ((uint32_t)(k64 >> 32) * 0x85EBCA6Bu)) * 0x9E3779B1u;
                                    } else {
                                        h = (x0 >> (32 - kb)) * 0x9E3779B1u;
                                    }
                                    hbv[j] = h >> 13;
                                    if (phase == 0) oldv[j] = atomicOr(&A[hbv[j] >> 5], 1u << (hbv[j] & 31u));
                                    else oldv[j] = Bm[hbv[j] >> 5];
                                }
                                m = 0;
#pragma unroll
                                for (int j = 0; j < 16; j++) {
                                    const uint32_t bit = 1u << (hbv[j] & 31u);
                                    if (phase == 0) atomicOr(&Bm[hbv[j] >> 5], oldv[j] & bit);
                                    else if (oldv[j] & bit) m |= 1u << (30 - 2 * j);
                                }
                            } else {
                            for (int q = 0; q < PB; q++) {
                                const uint32_t c = (pass >> (2 * (PB - 1 - q))) & 3u;
                                const uint32_t e0 = eq_mask(w0, c);
                                m &= q ? __builtin_amdgcn_alignbit(e0, eq_mask(w1, c), 32 - 2 * q) : e0;
                            }
                            if (first > 0) m &= 0xFFFFFFFFu >> (2 * first);
                            if (v < 16) m &= ~(0xFFFFFFFFu >> (2 * v));
                            }
                            while (m) {
                                const int b = __builtin_ctz(m);
                                m &= m - 1;
                                const int sh = 30 - b + 2 * PB;
                                const bool up = sh >= 32;
                                const uint32_t X = up ? w1 : w0, Y = up ? w2 : w1, Z = up ? w3 : w2;
                                const uint32_t sb = (uint32_t)sh & 31u;
                                const uint32_t x0 = sb ? __builtin_amdgcn_alignbit(X, Y, 32 - sb) : X;
                                key_t key;
                                uint32_t h, lo32, hi32 = 0;
                                if (KEY64) {
                                    const uint32_t x1 = sb ? __builtin_amdgcn_alignbit(Y, Z, 32 - sb) : Y;
                                    const unsigned long long k64 = (((unsigned long long)x0 << 32) | x1) >> (64 - kb);
                                    key = (key_t)k64;
                                    lo32 = (uint32_t)k64; hi32 = (uint32_t)(k64 >> 32);
                                    h = (lo32 ^ (hi32 * 0x85EBCA6Bu)) * 0x9E3779B1u;
                                } else {
                                    lo32 = x0 >> (32 - kb);
                                    key = (key_t)lo32;
                                    h = lo32 * 0x9E3779B1u;
                                }
                                const uint32_t hb = h >> 13;                     // 19 bits
                                const uint32_t bit = 1u << (hb & 31u);
                                if (phase == 0) {
                                    const uint32_t old = atomicOr(&A[hb >> 5], bit);
                                    if (old & bit) atomicOr(&Bm[hb >> 5], bit);
                                } else if (Bm[hb >> 5] & bit) {
                                    uint32_t slot = ((lo32 * 0xC2B2AE35u) ^ (hi32 * 0x27D4EB2Fu) ^ (lo32 >> 15)) * 0x165667B1u >> (32 - TLOG);
                                    for (int probes = 0;; probes++) {
                                        const key_t old = atomicCAS(&tab[slot], EMPTY, key);
                                        if (old == EMPTY) { mine++; break; }
                                        if (old == key) break;
                                        if (probes >= 64) { over_s = 1; break; }
                                        slot = (slot + 1) & ((1u << TLOG) - 1u);
                                    }
                                }
                            }
                        }
                        if (last) break;
                    }
                    __syncthreads();
                };
                for (int phase = 0; phase < 2; phase++) {
                    if (phase == 0) scan(std::integral_constant<int, 0>()); else scan(std::integral_constant<int, 1>());
                    if (phase == 0) {
                        // hash values marked once: one occurrence, one distinct k-mer each; A becomes the (empty) table
                        for (uint32_t w = tid; w < 4096; w += NT) {
                            const uint4 x = A4[w], y = B4[w];
                            mine += __popc(x.x & ~y.x) + __popc(x.y & ~y.y) + __popc(x.z & ~y.z) + __popc(x.w & ~y.w);
                            A4[w] = ones4;
                        }
                    } else {
                        for (uint32_t w = tid; w < 4096; w += NT) { A4[w] = zero4; B4[w] = zero4; }
                    }
                    __syncthreads();
                }
                if (over_s) break;
            }
            __syncthreads();
            if (!over_s) break;
            __syncthreads();
            if (tid == 0) over_s = 0;
            PB++;
            __syncthreads();
        }
        mine = wave_sum(mine);
        if ((tid & 63) == 0 && mine) atomicAdd(&distinct_s, mine);
        __syncthreads();
        if (tid == 0) distinct[f] = distinct_s;
        f = fnext;
    }
}

// ---- sixth family: as the fifth, but phase 2 first collects the keys whose hash value is in B into a list (HBM scratch of the
// workgroup, L2-resident) and then inserts the list in as many sub-passes as its length asks for, each owning a range of a second
// hash: the fragment is scanned twice whatever its length (up to share_max k-mers per pass), never once per sub-pass.
template <bool KEY64>
__global__ void __launch_bounds__(1024) k_rep7(const uint8_t* seqs, const uint64_t* foff, const uint32_t* flen, uint32_t nf, int k, uint32_t* distinct, uint32_t* work_ctr,
                                                uint32_t share_max, void* lists, uint32_t list_cap, int stop)
{
    constexpr int NT = 1024, SLOTS = 6, W = SLOTS * NT;
    constexpr int OV = KEY64 ? 2 : 1;
    constexpr uint32_t TLOG = KEY64 ? 13u : 14u;
    constexpr uint32_t CAP = (1u << TLOG) * 3u / 8u;                   // keys per sub-pass
    typedef typename std::conditional<KEY64, unsigned long long, uint32_t>::type key_t;
    __shared__ uint4 A4[4096];
    __shared__ uint4 B4[4096];
    __shared__ uint32_t codes[W + 8];
    __shared__ uint32_t distinct_s, next_s, over_s, list_n;
    uint32_t* A = reinterpret_cast<uint32_t*>(A4);
    uint32_t* Bm = reinterpret_cast<uint32_t*>(B4);
    key_t* tab = reinterpret_cast<key_t*>(A4);
    key_t* list = reinterpret_cast<key_t*>(lists) + (size_t)blockIdx.x * list_cap;
    const key_t EMPTY = ~(key_t)0;
    const int tid = (int)threadIdx.x;
    const uint4 ones4 = make_uint4(~0u, ~0u, ~0u, ~0u), zero4 = make_uint4(0, 0, 0, 0);
    for (uint32_t w = tid; w < 4096; w += NT) { A4[w] = zero4; B4[w] = zero4; }

    uint4 raw[SLOTS];
    auto chunks_of = [&](uint32_t f, const uint4*& base, int& a, int& L) {
        const uint8_t* s = seqs + foff[f];
        a = (int)((uintptr_t)s & 15u);
        base = reinterpret_cast<const uint4*>(s - a);
        L = (int)flen[f];
    };
    auto prefetch = [&](uint32_t f) {
        if (f >= nf) return;
        const uint4* base; int a, L;
        chunks_of(f, base, a, L);
        const int words = (a + L + 15) / 16;
#pragma unroll
        for (int sl = 0; sl < SLOTS; sl++) { const int g = tid + sl * NT; if (g < words) raw[sl] = base[g]; }
    };
    auto hash1 = [&](key_t key) -> uint32_t {
        if (KEY64) return ((uint32_t)key ^ ((uint32_t)((unsigned long long)key >> 32) * 0x85EBCA6Bu)) * 0x9E3779B1u;
        return (uint32_t)key * 0x9E3779B1u;
    };
    auto hash2 = [&](key_t key) -> uint32_t {
        const uint32_t lo = (uint32_t)key, hi = KEY64 ? (uint32_t)((unsigned long long)key >> 32) : 0u;
        return ((lo * 0xC2B2AE35u) ^ (hi * 0x27D4EB2Fu) ^ (lo >> 15)) * 0x165667B1u;
    };
    uint32_t f = blockIdx.x;
    prefetch(f);
    while (f < nf) {
        const uint4* base; int a, L;
        chunks_of(f, base, a, L);
        const int total = L - k + 1;
        const int words = (a + L + 15) / 16;
        const int kwords = total > 0 ? (a + total + 15) / 16 : 0;
        const bool one_window = words <= W;
        if (tid == 0) { distinct_s = 0; over_s = 0; list_n = 0; next_s = gridDim.x + atomicAdd(work_ctr, 1u); }
        __syncthreads();
#pragma unroll
        for (int sl = 0; sl < SLOTS; sl++) { const int g = tid + sl * NT; if (g < words) codes[g] = codes16(raw[sl]); }
        if (tid < 8) codes[(words < W ? words : W) + tid] = 0;
        __syncthreads();
        const uint32_t fnext = next_s;
        prefetch(fnext);
        int PB = (!KEY64 && k == 16) ? 1 : 0;
        while (PB < k - 1 && ((uint32_t)(total > 0 ? total : 0) >> (2 * PB)) > share_max) PB++;
        const int kb = 2 * (k - PB);
        const uint32_t passes = 1u << (2 * PB);
        uint32_t mine = 0;
        for (uint32_t pass = 0; pass < passes && kwords > 0 && stop != 1; pass++) {
            for (int phase = 0; phase < (stop == 2 || stop == 3 ? 1 : 2); phase++) {
                for (int wb = 0;; wb += W - OV) {
                    if (!one_window) {
                        __syncthreads();
                        for (int g = tid; g < W + 4 && wb + g < words + 4; g += NT) codes[g] = wb + g < words ? codes16(base[wb + g]) : 0u;
                    }
                    __syncthreads();
                    const bool last = wb + W >= words;
                    const int gend = last ? kwords - wb : W - OV;
                    for (int g = tid; g < gend; g += NT) {
                        const uint32_t w0 = codes[g], w1 = codes[g + 1], w2 = codes[g + 2], w3 = codes[g + 3];
                        const int first = a - 16 * (wb + g);
                        const int v = a + total - 16 * (wb + g);
                        uint32_t m = 0x55555555u;                   // starts to handle one by one below
                        if (PB == 0 && first <= 0 && v >= 16) {
                            uint32_t hbv[16], oldv[16];
#pragma unroll
                            for (int j = 0; j < 16; j++) {
                                const uint32_t x0 = j ? __builtin_amdgcn_alignbit(w0, w1, 32 - 2 * j) : w0;
                                key_t key;
                                if (KEY64) {
                                    const uint32_t x1 = j ? __builtin_amdgcn_alignbit(w1, w2, 32 - 2 * j) : w1;
                                    key = (key_t)((((unsigned long long)x0 << 32) | x1) >> (64 - kb));
                                } else key = (key_t)(x0 >> (32 - kb));
                                hbv[j] = hash1(key) >> 13;
                                if (phase == 0) oldv[j] = atomicOr(&A[hbv[j] >> 5], 1u << (hbv[j] & 31u));
                                else oldv[j] = Bm[hbv[j] >> 5];
                            }
                            m = 0;
#pragma unroll
                            for (int j = 0; j < 16; j++) {
                                const uint32_t bit = 1u << (hbv[j] & 31u);
                                if (phase == 0) { if (oldv[j] & bit) atomicOr(&Bm[hbv[j] >> 5], bit); }
                                else if (oldv[j] & bit) m |= 1u << (30 - 2 * j);
                            }
                        } else {
                            for (int q = 0; q < PB; q++) {
                                const uint32_t c = (pass >> (2 * (PB - 1 - q))) & 3u;
                                const uint32_t e0 = eq_mask(w0, c);
                                m &= q ? __builtin_amdgcn_alignbit(e0, eq_mask(w1, c), 32 - 2 * q) : e0;
                            }
                            if (first > 0) m &= 0xFFFFFFFFu >> (2 * first);
                            if (v < 16) m &= ~(0xFFFFFFFFu >> (2 * v));
                        }
                        while (m) {
                            const int b = __builtin_ctz(m);
                            m &= m - 1;
                            const int sh = 30 - b + 2 * PB;
                            const bool up = sh >= 32;
                            const uint32_t X = up ? w1 : w0, Y = up ? w2 : w1, Z = up ? w3 : w2;
                            const uint32_t sb = (uint32_t)sh & 31u;
                            const uint32_t x0 = sb ? __builtin_amdgcn_alignbit(X, Y, 32 - sb) : X;
                            key_t key;
                            if (KEY64) {
                                const uint32_t x1 = sb ? __builtin_amdgcn_alignbit(Y, Z, 32 - sb) : Y;
                                key = (key_t)((((unsigned long long)x0 << 32) | x1) >> (64 - kb));
                            } else key = (key_t)(x0 >> (32 - kb));
                            const uint32_t hb = hash1(key) >> 13;
                            const uint32_t bit = 1u << (hb & 31u);
                            if (phase == 0) {
                                const uint32_t old = atomicOr(&A[hb >> 5], bit);
                                if (old & bit) atomicOr(&Bm[hb >> 5], bit);
                            } else if (Bm[hb >> 5] & bit) {
                                const uint32_t at = atomicAdd(&list_n, 1u);
                                if (at < list_cap) list[at] = key;
                            }
                        }
                    }
                    if (last) break;
                }
                __syncthreads();
                if (phase == 0 && stop != 2) {
                    for (uint32_t w = tid; w < 4096; w += NT) {
                        const uint4 x = A4[w], y = B4[w];
                        mine += __popc(x.x & ~y.x) + __popc(x.y & ~y.y) + __popc(x.z & ~y.z) + __popc(x.w & ~y.w);
                        A4[w] = ones4;
                    }
                    __syncthreads();
                }
            }
            // phase 2b: the listed keys into the table, a range of the second hash at a time
            const uint32_t n_list = list_n < list_cap ? list_n : list_cap;      // (list_cap >= share_max: never cut)
            uint32_t nsub = stop == 4 ? 0u : (n_list + CAP - 1) / CAP;
            uint32_t mine2 = 0;
            for (;;) {
                mine2 = 0;
                for (uint32_t sp = 0; sp < nsub; sp++) {
                    for (uint32_t i = tid; i < n_list; i += NT) {
                        const key_t key = list[i];
                        const uint32_t h2 = hash2(key);
                        if ((((h2 & 0xFFFFu) * nsub) >> 16) != sp) continue;
                        uint32_t slot = h2 >> (32 - TLOG);
                        for (int probes = 0;; probes++) {
                            const key_t old = atomicCAS(&tab[slot], EMPTY, key);
                            if (old == EMPTY) { mine2++; break; }
                            if (old == key) break;
                            if (probes >= 64) { over_s = 1; break; }
                            slot = (slot + 1) & ((1u << TLOG) - 1u);
                        }
                    }
                    __syncthreads();
                    if (over_s) break;
                    if (sp + 1 < nsub) { for (uint32_t w = tid; w < 4096; w += NT) A4[w] = ones4; __syncthreads(); }
                }
                if (!over_s) break;
                __syncthreads();
                if (tid == 0) over_s = 0;
                for (uint32_t w = tid; w < 4096; w += NT) A4[w] = ones4;
                nsub *= 2;
                __syncthreads();
            }
            mine += mine2;
            if (tid == 0) list_n = 0;
            for (uint32_t w = tid; w < 4096; w += NT) { A4[w] = zero4; B4[w] = zero4; }
            __syncthreads();
        }
        mine = wave_sum(mine);
        if ((tid & 63) == 0 && mine) atomicAdd(&distinct_s, mine);
        __syncthreads();
        if (tid == 0) distinct[f] = distinct_s;
        f = fnext;
    }
}

// ---- seventh family: the sixth with the two scans compiled separately (no test of the phase between the sixteen LDS operations of
// a chunk, which made the compiler wait for each of them), an unconditional second mark (ORing zero where the bit was new), and
// the listed keys kept per lane (entry i of lane t at list[i * 1024 + t]): no counter, no scan, the lane that listed a key inserts it.
template <bool KEY64>
__global__ void __launch_bounds__(1024) k_rep8(const uint8_t* seqs, const uint64_t* foff, const uint32_t* flen, uint32_t nf, int k, uint32_t* distinct, uint32_t* work_ctr,
                                                uint32_t share_max, void* lists, uint32_t lane_cap, int stop)
{
    constexpr int NT = 1024, SLOTS = 6, W = SLOTS * NT;
    constexpr int OV = KEY64 ? 2 : 1;
    constexpr uint32_t TLOG = KEY64 ? 13u : 14u;
    constexpr uint32_t CAP = (1u << TLOG) * 3u / 8u;
    typedef typename std::conditional<KEY64, unsigned long long, uint32_t>::type key_t;
    __shared__ uint4 A4[4096];
    __shared__ uint4 B4[4096];
    __shared__ uint32_t codes[W + 8];
    __shared__ uint32_t distinct_s, next_s, over_s, list_n;
    uint32_t* A = reinterpret_cast<uint32_t*>(A4);
    uint32_t* Bm = reinterpret_cast<uint32_t*>(B4);
    key_t* tab = reinterpret_cast<key_t*>(A4);
    key_t* list = reinterpret_cast<key_t*>(lists) + (size_t)blockIdx.x * lane_cap * NT;
    const key_t EMPTY = ~(key_t)0;
    const int tid = (int)threadIdx.x;
    const uint4 ones4 = make_uint4(~0u, ~0u, ~0u, ~0u), zero4 = make_uint4(0, 0, 0, 0);
    for (uint32_t w = tid; w < 4096; w += NT) { A4[w] = zero4; B4[w] = zero4; }

    uint4 raw[SLOTS];
    auto chunks_of = [&](uint32_t f, const uint4*& base, int& a, int& L) {
        const uint8_t* s = seqs + foff[f];
        a = (int)((uintptr_t)s & 15u);
        base = reinterpret_cast<const uint4*>(s - a);
        L = (int)flen[f];
    };
    auto prefetch = [&](uint32_t f) {
        if (f >= nf) return;
        const uint4* base; int a, L;
        chunks_of(f, base, a, L);
        const int words = (a + L + 15) / 16;
#pragma unroll
        for (int sl = 0; sl < SLOTS; sl++) { const int g = tid + sl * NT; if (g < words) raw[sl] = base[g]; }
    };
    auto hash1 = [&](key_t key) -> uint32_t {
        if (KEY64) return ((uint32_t)key ^ ((uint32_t)((unsigned long long)key >> 32) * 0x85EBCA6Bu)) * 0x9E3779B1u;
        return (uint32_t)key * 0x9E3779B1u;
    };
    auto hash2 = [&](key_t key) -> uint32_t {
        const uint32_t lo = (uint32_t)key, hi = KEY64 ? (uint32_t)((unsigned long long)key >> 32) : 0u;
        return ((lo * 0xC2B2AE35u) ^ (hi * 0x27D4EB2Fu) ^ (lo >> 15)) * 0x165667B1u;
    };
    uint32_t f = blockIdx.x;
    prefetch(f);
    while (f < nf) {
        const uint4* base; int a, L;
        chunks_of(f, base, a, L);
        const int total = L - k + 1;
        const int words = (a + L + 15) / 16;
        const int kwords = total > 0 ? (a + total + 15) / 16 : 0;
        const bool one_window = words <= W;
        if (tid == 0) { distinct_s = 0; over_s = 0; list_n = 0; next_s = gridDim.x + atomicAdd(work_ctr, 1u); }
        __syncthreads();
#pragma unroll
        for (int sl = 0; sl < SLOTS; sl++) { const int g = tid + sl * NT; if (g < words) codes[g] = codes16(raw[sl]); }
        if (tid < 8) codes[(words < W ? words : W) + tid] = 0;
        __syncthreads();
        const uint32_t fnext = next_s;
        prefetch(fnext);
        int PB = 0;
        uint32_t mine = 0;
        for (;;) {                                                      // (again with more passes if a lane's list fills up)
            while (PB < k - 1 && ((uint32_t)(total > 0 ? total : 0) >> (2 * PB)) > share_max) PB++;
            const int kb = 2 * (k - PB);
            const uint32_t passes = 1u << (2 * PB);
            mine = 0;
            bool again = false;
            for (uint32_t pass = 0; pass < passes && kwords > 0 && stop != 1; pass++) {
                uint32_t ln = 0;                                        // keys this lane has listed
                // one scan of the fragment: PHASE 0 marks A and B, PHASE 1 lists the keys whose hash value is in B
                auto scan = [&](auto phase_tag) {
                    constexpr int PHASE = decltype(phase_tag)::value;
                    for (int wb = 0;; wb += W - OV) {
                        if (!one_window) {
                            __syncthreads();
                            for (int g = tid; g < W + 4 && wb + g < words + 4; g += NT) codes[g] = wb + g < words ? codes16(base[wb + g]) : 0u;
                        }
                        __syncthreads();
                        const bool last = wb + W >= words;
                        const int gend = last ? kwords - wb : W - OV;
                        for (int g = tid; g < gend; g += NT) {
                            const uint32_t w0 = codes[g], w1 = codes[g + 1], w2 = codes[g + 2], w3 = codes[g + 3];
                            const int first = a - 16 * (wb + g);
                            const int v = a + total - 16 * (wb + g);
                            if (PB == 0 && first <= 0 && v >= 16) {
                                uint32_t hbv[16], oldv[16];
                                key_t keyv[16];
#pragma unroll
                                for (int j = 0; j < 16; j++) {
                                    const uint32_t x0 = j ? __builtin_amdgcn_alignbit(w0, w1, 32 - 2 * j) : w0;
                                    if (KEY64) {
                                        const uint32_t x1 = j ? __builtin_amdgcn_alignbit(w1, w2, 32 - 2 * j) : w1;
                                        keyv[j] = (key_t)((((unsigned long long)x0 << 32) | x1) >> (64 - kb));
                                    } else keyv[j] = (key_t)(x0 >> (32 - kb));
                                    hbv[j] = hash1(keyv[j]) >> 13;
                                    if (PHASE == 0) oldv[j] = atomicOr(&A[hbv[j] >> 5], 1u << (hbv[j] & 31u));
                                    else oldv[j] = Bm[hbv[j] >> 5];
                                }
#pragma unroll
                                for (int j = 0; j < 16; j++) {
                                    const uint32_t hit = oldv[j] & (1u << (hbv[j] & 31u));
                                    if (PHASE == 0) atomicOr(&Bm[hbv[j] >> 5], hit);
                                    else if (hit) { if (ln < lane_cap) list[(size_t)ln * NT + tid] = keyv[j]; ln++; }
                                }
                            } else {
                                uint32_t m = 0x55555555u;
                                for (int q = 0; q < PB; q++) {
                                    const uint32_t c = (pass >> (2 * (PB - 1 - q))) & 3u;
                                    const uint32_t e0 = eq_mask(w0, c);
                                    m &= q ? __builtin_amdgcn_alignbit(e0, eq_mask(w1, c), 32 - 2 * q) : e0;
                                }
                                if (first > 0) m &= 0xFFFFFFFFu >> (2 * first);
                                if (v < 16) m &= ~(0xFFFFFFFFu >> (2 * v));
                                while (m) {
                                    const int b = __builtin_ctz(m);
                                    m &= m - 1;
                                    const int sh = 30 - b + 2 * PB;
                                    const bool up = sh >= 32;
                                    const uint32_t X = up ? w1 : w0, Y = up ? w2 : w1, Z = up ? w3 : w2;
                                    const uint32_t sb = (uint32_t)sh & 31u;
                                    const uint32_t x0 = sb ? __builtin_amdgcn_alignbit(X, Y, 32 - sb) : X;
                                    key_t key;
                                    if (KEY64) {
                                        const uint32_t x1 = sb ? __builtin_amdgcn_alignbit(Y, Z, 32 - sb) : Y;
                                        key = (key_t)((((unsigned long long)x0 << 32) | x1) >> (64 - kb));
                                    } else key = (key_t)(x0 >> (32 - kb));
                                    const uint32_t hb = hash1(key) >> 13;
                                    const uint32_t bit = 1u << (hb & 31u);
                                    if (PHASE == 0) {
                                        const uint32_t old = atomicOr(&A[hb >> 5], bit);
                                        if (old & bit) atomicOr(&Bm[hb >> 5], bit);
                                    } else if (Bm[hb >> 5] & bit) {
                                        if (ln < lane_cap) list[(size_t)ln * NT + tid] = key;
                                        ln++;
                                    }
                                }
                            }
                        }
                        if (last) break;
                    }
                    __syncthreads();
                };
                scan(std::integral_constant<int, 0>());
                if (stop == 2) continue;
                for (uint32_t w = tid; w < 4096; w += NT) {
                    const uint4 x = A4[w], y = B4[w];
                    mine += __popc(x.x & ~y.x) + __popc(x.y & ~y.y) + __popc(x.z & ~y.z) + __popc(x.w & ~y.w);
                    A4[w] = ones4;
                }
                __syncthreads();
                if (stop == 3) continue;
                scan(std::integral_constant<int, 1>());
                // the listed keys of all lanes: how many sub-passes the table needs; a lane whose list ran over asks for more passes
                {
                    const uint32_t t = wave_sum(ln < lane_cap ? ln : lane_cap);
                    if ((tid & 63) == 0 && t) atomicAdd(&list_n, t);
                    if (ln > lane_cap) over_s = 1;
                }
                __syncthreads();
                if (over_s) { again = true; break; }
                const uint32_t n_list = list_n;
                uint32_t nsub = stop == 4 ? 0u : (n_list + CAP - 1) / CAP;
                uint32_t mine2 = 0;
                for (;;) {
                    mine2 = 0;
                    for (uint32_t sp = 0; sp < nsub; sp++) {
                        for (uint32_t i = 0; i < ln; i++) {
                            const key_t key = list[(size_t)i * NT + tid];
                            const uint32_t h2 = hash2(key);
                            if ((((h2 & 0xFFFFu) * nsub) >> 16) != sp) continue;
                            uint32_t slot = h2 >> (32 - TLOG);
                            for (int probes = 0;; probes++) {
                                const key_t old = atomicCAS(&tab[slot], EMPTY, key);
                                if (old == EMPTY) { mine2++; break; }
                                if (old == key) break;
                                if (probes >= 64) { over_s = 1; break; }
                                slot = (slot + 1) & ((1u << TLOG) - 1u);
                            }
                        }
                        __syncthreads();
                        if (over_s) break;
                        if (sp + 1 < nsub) { for (uint32_t w = tid; w < 4096; w += NT) A4[w] = ones4; __syncthreads(); }
                    }
                    if (!over_s) break;
                    __syncthreads();
                    if (tid == 0) over_s = 0;
                    for (uint32_t w = tid; w < 4096; w += NT) A4[w] = ones4;
                    nsub *= 2;
                    __syncthreads();
                }
                mine += mine2;
                if (tid == 0) list_n = 0;
                for (uint32_t w = tid; w < 4096; w += NT) { A4[w] = zero4; B4[w] = zero4; }
                __syncthreads();
            }
            if (!again) break;
            __syncthreads();
            if (tid == 0) { over_s = 0; list_n = 0; }
            for (uint32_t w = tid; w < 4096; w += NT) { A4[w] = zero4; B4[w] = zero4; }
            PB++;
            __syncthreads();
        }
        if (stop == 2 || stop == 3) { __syncthreads(); for (uint32_t w = tid; w < 4096; w += NT) { A4[w] = zero4; B4[w] = zero4; } }
        mine = wave_sum(mine);
        if ((tid & 63) == 0 && mine) atomicAdd(&distinct_s, mine);
        __syncthreads();
        if (tid == 0) distinct[f] = distinct_s;
        f = fnext;
    }
}

struct Data {
    uint8_t* seq; uint64_t* off; uint32_t* len; uint32_t nf; uint64_t bases;
};

template <int MODE, int PART, int WIN = 96 * 1024>
static int run(const char* name, const Data& D, int k, int threads, int grid, std::vector<uint32_t>& ref, bool is_ref)
{
    uint32_t* d_out;
    CK(hipMalloc(&d_out, D.nf * 4));
    CK(hipMemset(d_out, 0xFF, D.nf * 4));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL((k_rep<MODE, PART, WIN>), dim3(grid), dim3(threads), 0, 0, D.seq, D.off, D.len, D.nf, k, d_out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((k_rep<MODE, PART, WIN>), dim3(grid), dim3(threads), 0, 0, D.seq, D.off, D.len, D.nf, k, d_out);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    std::vector<uint32_t> h(D.nf);
    CK(hipMemcpy(h.data(), d_out, D.nf * 4, hipMemcpyDeviceToHost));
    size_t bad = 0;
    if (is_ref) ref = h;
    else for (uint32_t i = 0; i < D.nf; i++) bad += h[i] != ref[i];
    printf("%-28s k %2d threads %4d grid %4d  %8.3f ms  %7.1f Gbases/s  -> %.2f ms per 2.93-Gbases batch  %s\n", name, k, threads, grid, ms,
           D.bases / (ms * 1e6), ms * 2.93e9 / D.bases, is_ref ? "(reference counts)" : bad ? "MISMATCH" : "same counts");
    CK(hipFree(d_out));
    return 0;
}

template <int LOOP>
static int run2(const char* name, const Data& D, int k, int threads, int grid, uint32_t undo_below, std::vector<uint32_t>& ref)
{
    uint32_t* d_out;
    CK(hipMalloc(&d_out, D.nf * 4));
    CK(hipMemset(d_out, 0xFF, D.nf * 4));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL((k_rep2<LOOP, 96 * 1024>), dim3(grid), dim3(threads), 0, 0, D.seq, D.off, D.len, D.nf, k, d_out, undo_below);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((k_rep2<LOOP, 96 * 1024>), dim3(grid), dim3(threads), 0, 0, D.seq, D.off, D.len, D.nf, k, d_out, undo_below);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    std::vector<uint32_t> h(D.nf);
    CK(hipMemcpy(h.data(), d_out, D.nf * 4, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (uint32_t i = 0; i < D.nf; i++) bad += h[i] != ref[i];
    printf("%-22s undo<%-6u k %2d threads %4d grid %4d  %8.3f ms  %7.1f Gbases/s  -> %.2f ms per 2.93-Gbases batch  %s\n", name, undo_below, k, threads, grid, ms,
           D.bases / (ms * 1e6), ms * 2.93e9 / D.bases, bad ? "MISMATCH" : "same counts");
    CK(hipFree(d_out));
    return 0;
}

template <int SLOTS>
static int run3(const char* name, const Data& D, int k, int grid, std::vector<uint32_t>& ref)
{
    uint32_t* d_out; uint32_t* d_ctr;
    CK(hipMalloc(&d_out, D.nf * 4));
    CK(hipMalloc(&d_ctr, 4));
    CK(hipMemset(d_out, 0xFF, D.nf * 4));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipMemset(d_ctr, 0, 4));
    hipLaunchKernelGGL((k_rep3<SLOTS>), dim3(grid), dim3(1024), 0, 0, D.seq, D.off, D.len, D.nf, k, d_out, d_ctr);
    CK(hipDeviceSynchronize());
    CK(hipMemset(d_ctr, 0, 4));
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((k_rep3<SLOTS>), dim3(grid), dim3(1024), 0, 0, D.seq, D.off, D.len, D.nf, k, d_out, d_ctr);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    std::vector<uint32_t> h(D.nf);
    CK(hipMemcpy(h.data(), d_out, D.nf * 4, hipMemcpyDeviceToHost));
    size_t bad = 0;
    std::vector<uint64_t> ho(D.nf); std::vector<uint32_t> hl(D.nf);
    CK(hipMemcpy(ho.data(), D.off, D.nf * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hl.data(), D.len, D.nf * 4, hipMemcpyDeviceToHost));
    for (uint32_t i = 0; i < D.nf; i++) if (h[i] != ref[i]) { if (bad < 6) printf("   f %u len %u align %u got %u want %u\n", i, hl[i], (unsigned)(ho[i] & 15), h[i], ref[i]); bad++; }
    if (bad) printf("   %zu of %u differ\n", bad, D.nf);
    printf("%-22s slots %d       k %2d threads 1024 grid %4d  %8.3f ms  %7.1f Gbases/s  -> %.2f ms per 2.93-Gbases batch  %s\n", name, SLOTS, k, grid, ms,
           D.bases / (ms * 1e6), ms * 2.93e9 / D.bases, bad ? "MISMATCH" : "same counts");
    CK(hipFree(d_out)); CK(hipFree(d_ctr));
    return 0;
}

static void host_counts(const std::vector<uint8_t>& seq, const std::vector<uint64_t>& off, const std::vector<uint32_t>& len, int k, std::vector<uint32_t>& out)
{
    out.assign(len.size(), 0);
    std::vector<unsigned long long> v;
    const unsigned long long mask = (1ull << (2 * k)) - 1ull;
    for (size_t f = 0; f < len.size(); f++) {
        const int total = (int)len[f] - k + 1;
        if (total <= 0) continue;
        v.resize(total);
        unsigned long long km = 0;
        for (int i = 0; i < (int)len[f]; i++) {
            const uint8_t c = seq[off[f] + i];
            km = ((km << 2) | (c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : 0)) & mask;
            if (i >= k - 1) v[i - k + 1] = km;
        }
        std::sort(v.begin(), v.end());
        out[f] = (uint32_t)(std::unique(v.begin(), v.end()) - v.begin());
    }
}

template <bool KEY64>
static int run5(const char* name, const Data& D, int k, int grid, const std::vector<uint32_t>* ref)
{
    uint32_t* d_out; uint32_t* d_ctr;
    CK(hipMalloc(&d_out, D.nf * 4));
    CK(hipMalloc(&d_ctr, 4));
    CK(hipMemset(d_out, 0xFF, D.nf * 4));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipMemset(d_ctr, 0, 4));
    hipLaunchKernelGGL((k_rep5<KEY64>), dim3(grid), dim3(1024), 0, 0, D.seq, D.off, D.len, D.nf, k, d_out, d_ctr);
    CK(hipDeviceSynchronize());
    CK(hipMemset(d_ctr, 0, 4));
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((k_rep5<KEY64>), dim3(grid), dim3(1024), 0, 0, D.seq, D.off, D.len, D.nf, k, d_out, d_ctr);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    std::vector<uint32_t> h(D.nf);
    CK(hipMemcpy(h.data(), d_out, D.nf * 4, hipMemcpyDeviceToHost));
    size_t bad = 0;
    if (ref) for (uint32_t i = 0; i < D.nf; i++) if (h[i] != (*ref)[i]) { if (bad < 6) printf("   f %u got %u want %u\n", i, h[i], (*ref)[i]); bad++; }
    printf("%-22s %s k %2d threads 1024 grid %4d  %8.3f ms  %7.1f Gbases/s  -> %.2f ms per 2.93-Gbases batch  %s\n", name, KEY64 ? "64-bit keys" : "32-bit keys", k, grid, ms,
           D.bases / (ms * 1e6), ms * 2.93e9 / D.bases, !ref ? "(not checked)" : bad ? "MISMATCH" : "same counts");
    CK(hipFree(d_out)); CK(hipFree(d_ctr));
    return 0;
}

template <bool KEY64, bool FAST = true>
static int run6(const char* name, const Data& D, int k, int grid, const std::vector<uint32_t>* ref, uint32_t share_max)
{
    uint32_t* d_out; uint32_t* d_ctr;
    CK(hipMalloc(&d_out, D.nf * 4));
    CK(hipMalloc(&d_ctr, 4));
    CK(hipMemset(d_out, 0xFF, D.nf * 4));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipMemset(d_ctr, 0, 4));
    hipLaunchKernelGGL((k_rep6<KEY64, FAST>), dim3(grid), dim3(1024), 0, 0, D.seq, D.off, D.len, D.nf, k, d_out, d_ctr, share_max);
    CK(hipDeviceSynchronize());
    CK(hipMemset(d_ctr, 0, 4));
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((k_rep6<KEY64, FAST>), dim3(grid), dim3(1024), 0, 0, D.seq, D.off, D.len, D.nf, k, d_out, d_ctr, share_max);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    std::vector<uint32_t> h(D.nf);
    CK(hipMemcpy(h.data(), d_out, D.nf * 4, hipMemcpyDeviceToHost));
    size_t bad = 0;
    if (ref) for (uint32_t i = 0; i < D.nf; i++) if (h[i] != (*ref)[i]) { if (bad < 6) printf("   f %u got %u want %u\n", i, h[i], (*ref)[i]); bad++; }
    printf("%-22s share<=%-6u %s k %2d threads 1024 grid %4d  %8.3f ms  %7.1f Gbases/s  -> %.2f ms per 2.93-Gbases batch  %s\n", name, share_max, KEY64 ? "64-bit keys" : "32-bit keys", k, grid, ms,
           D.bases / (ms * 1e6), ms * 2.93e9 / D.bases, !ref ? "(not checked)" : bad ? "MISMATCH" : "same counts");
    CK(hipFree(d_out)); CK(hipFree(d_ctr));
    return 0;
}

template <bool KEY64>
static int run7(const char* name, const Data& D, int k, int grid, const std::vector<uint32_t>* ref, uint32_t share_max, int stop = 0)
{
    uint32_t* d_out; uint32_t* d_ctr; void* d_lists;
    CK(hipMalloc(&d_lists, (size_t)grid * (share_max + 64) * 8));
    CK(hipMalloc(&d_out, D.nf * 4));
    CK(hipMalloc(&d_ctr, 4));
    CK(hipMemset(d_out, 0xFF, D.nf * 4));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipMemset(d_ctr, 0, 4));
    hipLaunchKernelGGL((k_rep7<KEY64>), dim3(grid), dim3(1024), 0, 0, D.seq, D.off, D.len, D.nf, k, d_out, d_ctr, share_max, d_lists, share_max + 64, stop);
    CK(hipDeviceSynchronize());
    CK(hipMemset(d_ctr, 0, 4));
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((k_rep7<KEY64>), dim3(grid), dim3(1024), 0, 0, D.seq, D.off, D.len, D.nf, k, d_out, d_ctr, share_max, d_lists, share_max + 64, stop);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    std::vector<uint32_t> h(D.nf);
    CK(hipMemcpy(h.data(), d_out, D.nf * 4, hipMemcpyDeviceToHost));
    size_t bad = 0;
    if (ref) for (uint32_t i = 0; i < D.nf; i++) if (h[i] != (*ref)[i]) { if (bad < 6) printf("   f %u got %u want %u\n", i, h[i], (*ref)[i]); bad++; }
    printf("%-22s share<=%-6u %s k %2d threads 1024 grid %4d  %8.3f ms  %7.1f Gbases/s  -> %.2f ms per 2.93-Gbases batch  %s\n", name, share_max, KEY64 ? "64-bit keys" : "32-bit keys", k, grid, ms,
           D.bases / (ms * 1e6), ms * 2.93e9 / D.bases, !ref ? "(not checked)" : bad ? "MISMATCH" : "same counts");
    CK(hipFree(d_out)); CK(hipFree(d_ctr)); CK(hipFree(d_lists));
    return 0;
}

template <bool KEY64>
static int run8(const char* name, const Data& D, int k, int grid, const std::vector<uint32_t>* ref, uint32_t share_max, int stop = 0)
{
    uint32_t* d_out; uint32_t* d_ctr; void* d_lists;
    CK(hipMalloc(&d_lists, (size_t)grid * (share_max + 65536) * 8));
    CK(hipMalloc(&d_out, D.nf * 4));
    CK(hipMalloc(&d_ctr, 4));
    CK(hipMemset(d_out, 0xFF, D.nf * 4));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipMemset(d_ctr, 0, 4));
    hipLaunchKernelGGL((k_rep8<KEY64>), dim3(grid), dim3(1024), 0, 0, D.seq, D.off, D.len, D.nf, k, d_out, d_ctr, share_max, d_lists, share_max / 1024 + 64, stop);
    CK(hipDeviceSynchronize());
    CK(hipMemset(d_ctr, 0, 4));
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((k_rep8<KEY64>), dim3(grid), dim3(1024), 0, 0, D.seq, D.off, D.len, D.nf, k, d_out, d_ctr, share_max, d_lists, share_max / 1024 + 64, stop);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    std::vector<uint32_t> h(D.nf);
    CK(hipMemcpy(h.data(), d_out, D.nf * 4, hipMemcpyDeviceToHost));
    size_t bad = 0;
    if (ref) for (uint32_t i = 0; i < D.nf; i++) if (h[i] != (*ref)[i]) { if (bad < 6) printf("   f %u got %u want %u\n", i, h[i], (*ref)[i]); bad++; }
    printf("%-22s share<=%-6u %s k %2d threads 1024 grid %4d  %8.3f ms  %7.1f Gbases/s  -> %.2f ms per 2.93-Gbases batch  %s\n", name, share_max, KEY64 ? "64-bit keys" : "32-bit keys", k, grid, ms,
           D.bases / (ms * 1e6), ms * 2.93e9 / D.bases, !ref ? "(not checked)" : bad ? "MISMATCH" : "same counts");
    CK(hipFree(d_out)); CK(hipFree(d_ctr)); CK(hipFree(d_lists));
    return 0;
}

int main(int argc, char** argv)
{
    const uint32_t nf = argc > 1 ? (uint32_t)atoi(argv[1]) : 65536u;
    std::mt19937_64 rng(7);
    std::vector<uint64_t> off(nf);
    std::vector<uint32_t> len(nf);
    uint64_t bases = 0;
    std::lognormal_distribution<double> ln(10.394, 0.8);      // the bench's shape: mean 45 kb
    for (uint32_t i = 0; i < nf; i++) {
        double v = ln(rng);
        if (v < 200) v = 200;
        if (v > 2e6) v = 2e6;
        len[i] = (uint32_t)v;
        off[i] = bases;
        bases += len[i] + 1;                                   // odd alignments, as in a text
    }
    if (argc > 2 && atoi(argv[2]) == 1) {        // longest first (offsets follow)
        std::sort(len.begin(), len.end(), [](uint32_t x, uint32_t y) { return x > y; });
        bases = 0;
        for (uint32_t i = 0; i < nf; i++) { off[i] = bases; bases += len[i] + 1; }
        printf("fragments sorted by length, longest first\n");
    }
    if (argc > 2 && atoi(argv[2]) == 2) {        // no fragment longer than one window
        bases = 0;
        for (uint32_t i = 0; i < nf; i++) { if (len[i] > 98000) len[i] = 98000; off[i] = bases; bases += len[i] + 1; }
        printf("fragments capped at 98000 bases\n");
    }
    std::vector<uint8_t> seq(bases + 64);
    const char* ACGT = "ACGT";
    uint64_t x = 88172645463325252ull;
    for (uint64_t i = 0; i < bases; i++) {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        seq[i] = (x >> 60) == 0 && (x & 1023) == 0 ? 'N' : ACGT[(x >> 33) & 3];
    }
    // a few low-complexity fragments
    for (uint32_t i = 0; i < nf; i += 97) for (uint32_t j = 0; j < len[i]; j++) seq[off[i] + j] = "ACG"[j % 3];
    Data D;
    D.nf = nf; D.bases = bases;
    CK(hipMalloc(&D.seq, seq.size()));
    CK(hipMalloc(&D.off, nf * 8));
    CK(hipMalloc(&D.len, nf * 4));
    CK(hipMemcpy(D.seq, seq.data(), seq.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(D.off, off.data(), nf * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(D.len, len.data(), nf * 4, hipMemcpyHostToDevice));
    printf("%u fragments, %.3f Gbases\n", nf, bases * 1e-9);
    std::vector<uint32_t> ref;
    const bool check = nf <= 8192;                 // small runs are checked against a sort on the host
    for (int k : {11, 13, 21}) {
        std::vector<uint32_t> hc;
        if (check) host_counts(seq, off, len, k, hc);
        const std::vector<uint32_t>* hp = check ? &hc : nullptr;
        if (k <= 13) {
            run<0, 20>("round 1: rolling k-mer", D, k, 256, 256, ref, true);
            if (check) { size_t bad = 0; for (uint32_t i = 0; i < nf; i++) bad += ref[i] != hc[i]; printf("   host counts vs the round-1 kernel: %zu differ\n", bad); }
            run<0, 20>("round 1: rolling k-mer", D, k, 1024, 256, ref, false);
            run2<1>("word/lane, mask walk", D, k, 1024, 256, 0, ref);
            run3<6>("= k_repeat", D, k, 256, ref);
        }
        if (k <= 15) run5<false>("LDS table of keys", D, k, 256, hp);
        else run5<true>("LDS table of keys", D, k, 256, hp);
        if (k <= 15) {
            run6<false, false>("two maps + table, walk", D, k, 256, hp, 49152);
            run6<false, true>("= k_repeat_keys", D, k, 256, hp, 65536);
            run8<false>("two maps, lane lists", D, k, 256, hp, 131072, 0);
        } else {
            run6<true, false>("two maps + table, walk", D, k, 256, hp, 49152);
            run6<true, true>("= k_repeat_keys", D, k, 256, hp, 65536);
            run8<true>("two maps, lane lists", D, k, 256, hp, 131072, 0);
        }
    }
    return 0;
}
