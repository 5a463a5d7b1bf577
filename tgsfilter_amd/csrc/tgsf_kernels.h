// tgsf_kernels.h -- the HIP kernels of the per-read filtering hot path (gfx950, wave64).
//
// Pipeline per batch (one stream plus an auxiliary one for the end kernels, no host round trip):
//   k_prepare        lengths, per-read state, stats-tile histogram
//   k_tile_scan/k_tile_scatter/k_build_work   counting sort of items by tile count, work list of k_stats
//   k_stats<raw>     CalcAvgQuality on every read: 100-bp bin tables (per batch) + sumQ   [HBM/VALU]
//   k_fold_raw       batch tables -> job tables (raw; clean too when tallied by difference)
//   k_gate_reads     mean-Q gate, rawDiffQual, middle-scan segment counts
//   k_end_tables<raw>   Get_5p/3p_base_qual                                  (auxiliary stream)
//   k_end_windows    GetEditDistance 5'/3' windows (edlib HW/PATH semantics) (auxiliary stream)
//   k_scan_u32       exclusive scan (segment bases)
//   k_mid_scan*      GetEditDistance middle: Myers infix scan, 1 lane = 1024 columns  [VALU-bound, dominant]
//   k_mid_resolve    start locations + path of the first location + similarity gates
//   k_regions<count|emit>   adapterMap: merge drop regions, keep regions, DropInfo
//   k_repeat / k_repeat_keys   GetKmerCount gate (-p/-k), only when asked for: LDS bitmap (k <= 11) / hashed map + keys (12..32)
//   k_clean_plan     reads kept whole; which way the clean tables are cheaper to tally
//   k_frag_prepare + sort + k_stats<clean> + k_gate_frags + k_end_tables<clean>
//   k_finalize       tgsf_read_result / tgsf_fragment records
//
// Reference lines are cited at each kernel.  Device code on the primitives of tgsf_hip.h; tests/emul compiles
// the same file on tgsf_emul.h (serial emulation, test infrastructure).
#pragma once
#include <type_traits>
#include "tgsf_dev.h"
#include "../../include/tgsf.h"
static_assert(TGSF_CTR_END_TABLES == 533, "tally layout");

namespace tgsf {

typedef unsigned long long ull;

// add a per-lane contribution to one global u64 word: one atomic per wave.
// Must be reached by all lanes of the wave (pass 0 for lanes with nothing to add).
TGSF_D void wave_add_u64(uint64_t* dst, uint64_t v) {
    uint64_t t = wave_sum(v);
    if (wave_leader() && t) atomicAdd((ull*)dst, (ull)t);
}
TGSF_D void wave_max_u64(uint64_t* dst, uint32_t v) {
    uint32_t t = wave_max(v);
    if (wave_leader() && t) atomicMax((ull*)dst, (ull)t);
}
TGSF_D void set_status(const DevBatch& B, uint32_t code, uint32_t detail) {
    if (atomicMax(&B.status[0], code) < code) B.status[1] = detail;
}

// The candidate pool of the middle scan overflowed: everything behind the scan leaves this batch alone (no tally, no
// record is written), and tgsf_wait runs the scan and those kernels again with a pool grown to fit (mid_mode).
TGSF_D bool pool_overflowed(const DevBatch& B) { return *B.ovf != 0u; }

// fragments actually stored (the count may exceed the capacity: DS_FRAG_CAP)
TGSF_D uint32_t stored_frags(const DevBatch& B) { uint32_t nf = B.nfr[B.n]; return nf < B.fcap ? nf : B.fcap; }

// ---------------------------------------------------------------------------
// The clean bin tables of a batch can be tallied two ways, both exact:
//   direct      scan every surviving fragment;
//   difference  clean += raw tallies of the batch - reads NOT kept whole + their fragments,
//               because a read kept whole adds to the clean tables exactly what it added to the raw ones.
// k_clean_plan counts the bases either way needs to scan; the cheaper one runs (difference must win by
// a quarter: it also pays one more pass over the batch's table rows).  Valid after k_clean_plan.
// ---------------------------------------------------------------------------
//   by-product  (round 5) the raw pass itself tallied [head_trim, L - tail_trim) of every read it expected to be kept
//               as exactly that (B.spec) into the clean tables: clean += - the speculated ranges of the reads that turned
//               out otherwise + their real fragments.  `whole` then means "kept as speculated".
// The batch speculated or it did not (B.bp_used, decided before the raw pass); when it did, that is the strategy.
TGSF_D bool clean_by_product(const DevBatch& B) { return B.bp_allowed && *B.bp_used != 0u; }
TGSF_D bool clean_by_difference(const DevBatch& B) {
    if (clean_by_product(B)) return true;            // (same bookkeeping: `whole` reads are not scanned again, the others taken out)
    if (B.clean_force) return B.clean_force == 2;
    const uint64_t direct = B.plan[2], diff = B.plan[3];
    return diff + (diff >> 2) < direct;
}
// the range of read r the clean tables already hold when the read is `whole`: the read itself, or what the raw pass speculated on
TGSF_D void held_range(const DevParams& P, const DevBatch& B, uint32_t r, uint32_t& s, uint32_t& e) {
    s = 0; e = B.len[r];
    if (clean_by_product(B)) {
        if (B.spec[r]) { s = (uint32_t)P.head_trim; e = B.len[r] - (uint32_t)P.tail_trim; }
        else e = 0;                                  // nothing held: nothing to take out
    }
}
// Items of a stats pass.  RAW: read i.  CLEAN: fragment i (< fcap), or read i - fcap to be taken back out.
// Length 0 = not part of the pass.
// (RAW, a read the batch speculates on: the raw pass goes as far as the speculated fragment does, [0, L - tail_trim) -- so
// that a lane's tallies of such a read are all of the fragment or in front of it --, the tail_trim bytes behind it are
// tallied by k_tail_fix)
template <bool CLEAN>
TGSF_D uint32_t stats_item_len(const DevParams& P, const DevBatch& B, uint32_t item, bool diff) {
    if (!CLEAN) return B.len[item] - ((B.bp_allowed && B.spec[item]) ? (uint32_t)P.tail_trim : 0u);
    if (item >= B.fcap) {
        if (!diff || B.whole[item - B.fcap]) return 0u;
        uint32_t s, e;
        held_range(P, B, item - B.fcap, s, e);
        return e > s ? e - s : 0u;
    }
    if (B.frag_flags[item] & TGSF_FF_REPEAT) return 0u;          // dropped before CalcAvgQuality (:1982-1989)
    if (diff && B.whole[B.frag_read[item]]) return 0u;
    return B.frag_len[item];
}
// Segments of a pass's work list (DevBatch::tile_hist).  The raw pass of a context that may speculate: the reads the batch
// speculates on, then the others.  The clean pass: the fragments, then the reads to be taken back out -- k_stats flushes
// its tallies whenever the sign of the items changes, and with the two kinds mixed as the counting sort leaves them that
// happened at every other item (measured, round 6: 7.8 M atomics in a pass of 48 000 work items, 0.80 ms; in two segments
// a wave changes sign once at most).
template <bool CLEAN> TGSF_D uint32_t stats_segments(const DevBatch& B) { return (CLEAN || B.bp_allowed) ? 2u : 1u; }
template <bool CLEAN> TGSF_D uint32_t stats_segment_of(const DevBatch& B, uint32_t item) {
    return CLEAN ? (item >= B.fcap ? 1u : 0u) : ((B.bp_allowed && !B.spec[item]) ? 1u : 0u);
}
// i-th candidate of the clean pass -> item number
TGSF_D uint32_t clean_item(const DevBatch& B, uint32_t i, uint32_t nf) { return i < nf ? i : B.fcap + (i - nf); }

TGSF_D uint32_t gtid() { return blockIdx.x * blockDim.x + threadIdx.x; }
TGSF_D uint32_t gsize() { return gridDim.x * blockDim.x; }

// largest i in [0,n) with a[i] <= v, for a non-decreasing a[0..n] with a[0] <= v < a[n]
TGSF_D uint32_t find_owner(const uint32_t* a, uint32_t n, uint32_t v) {
    uint32_t lo = 0, hi = n;
    while (hi - lo > 1) {
        uint32_t mid = (lo + hi) >> 1;
        if (a[mid] <= v) lo = mid; else hi = mid;
    }
    return lo;
}

// ---------------------------------------------------------------------------
// Block-local histogram of tile counts: same-address global atomics serialise in
// L2 (all reads of a batch fall into a handful of buckets), LDS atomics do not.
// ---------------------------------------------------------------------------
constexpr uint32_t kHistLds = 4096;

// ---------------------------------------------------------------------------
// k_prepare: per-read state.  Items of the raw stats pass are the reads.
// ---------------------------------------------------------------------------
TGSF_KERNEL k_prepare(DevParams P, DevBatch B, uint32_t max_read_len)
{
    TGSF_SHARED uint32_t h[kHistLds];
    const int A = P.n_adapters;
    const uint32_t seg_stride = B.max_tiles + 2;
    const uint32_t nbuck = seg_stride * stats_segments<false>(B);
    const bool use_lds = nbuck <= kHistLds;
    if (use_lds) for (uint32_t i = TGSF_COOP_BEGIN; i < nbuck; i += TGSF_COOP_STRIDE) h[i] = 0;
    TGSF_BLOCK_SYNC();
    uint32_t rows = 0, erows = 0;
    // does this batch speculate (see DevBatch::spec)?  Its first run decides from what the batch before left in bp_state
    // and notes it in the batch's own word; a second run after a pool overflow reads that word.
    bool speculate = false;
    if (B.bp_allowed) {
        speculate = B.replay ? *B.bp_used != 0u : B.bp_state[0] != 0u;
        if (!B.replay && gtid() == 0) { *B.bp_used = speculate ? 1u : 0u; if (speculate) B.bp_state[1]++; }
    }
    for (uint32_t r = gtid(); r < B.n; r += gsize()) {
        uint32_t L = B.len_in ? B.len_in[r] : (uint32_t)(B.off[r + 1] - B.off[r]);
        B.len[r] = L;
        if (B.bp_allowed) { B.spec[r] = 0; B.spec_sum[r] = 0; }
        B.sumq[r] = 0;
        B.flags[r] = 0;
        B.mid_head[r] = -1;
        B.mid_cnt[r] = 0;
        B.trimmed[r] = 0;
        B.seg_cnt[r] = 0;
        B.chk_cnt[r] = 0;
        B.nfr[r] = 0;
        for (int a = 0; a < A; a++) { B.clip5[(size_t)r * A + a] = 0; B.clip3[(size_t)r * A + a] = -1; B.mid_best[(size_t)r * A + a] = 0x7FFFFFFF; }
        if (L == 0 || L > max_read_len) { set_status(B, DS_BAD_LEN, r); B.len[r] = 0; continue; }
        uint32_t rw = L / kBin + 1;                                     // src/TGSFilter.cpp:1445
        rows = rw > rows ? rw : rows;
        uint32_t er = (uint32_t)P.bc_len < L ? (uint32_t)P.bc_len : L;  // :1490-1493
        erows = er > erows ? er : erows;
        if (speculate) {
            // kept as [head_trim, L - tail_trim) if nothing else happens to the read (adapterMap, src/TGSFilter.cpp:1334-1347,
            // :1388-1431: one keep region of a length the filters accept) ...
            const int64_t keep = (int64_t)L - P.head_trim - P.tail_trim;
            bool sp = keep >= P.min_len && keep <= P.max_len;
            // ... and if it passes the mean-quality gate (:1946-1953), which only the raw pass can tell: a guess from 64
            // quality bytes at four places of the read.  A wrong guess costs a second look at the read, never a result.
            if (sp && !P.no_qual) {
                const uint8_t* q = B.qual + B.qoff[r];
                uint32_t sum = 0, cntq = 0;
                for (int k = 0; k < 4; k++) {
                    const uint32_t at = (uint32_t)(((uint64_t)(L > 16 ? L - 16 : 0) * (uint32_t)k) / 3u);
                    for (uint32_t i = at; i < at + 16 && i < L; i++) { sum += (uint32_t)(int32_t)(int8_t)q[i] + 128u; cntq++; }
                }
                const double guess = ((double)sum - 128.0 * cntq) / (double)(cntq ? cntq : 1u) - (double)P.qtype;
                sp = !q_fail(guess, P.min_q, P.max_q);
            }
            B.spec[r] = sp ? 1u : 0u;
        }
        // (the read's place in the raw pass: its segment and the tiles of what that pass covers of it)
        const uint32_t Ls = stats_item_len<false>(P, B, r, false);
        const uint32_t v = (Ls + kTileBases - 1) / kTileBases + stats_segment_of<false>(B, r) * seg_stride;
        if (use_lds) atomicAdd(&h[v], 1u); else atomicAdd(&B.tile_hist[v], 1u);
    }
    wave_max_u64(&B.ctr[TGSF_CTR_ROWS + 0], rows);
    wave_max_u64(&B.plan[0], rows);
    wave_max_u64(&B.ctr[TGSF_CTR_ROWS + 2], erows);
    TGSF_BLOCK_SYNC();
    if (use_lds)
        for (uint32_t i = TGSF_COOP_BEGIN; i < nbuck; i += TGSF_COOP_STRIDE)
            if (h[i]) atomicAdd(&B.tile_hist[i], h[i]);
}

// k_clean_plan: which reads are kept whole, and the bases each clean-table strategy would scan.
TGSF_KERNEL k_clean_plan(DevParams P, DevBatch B)
{
    if (pool_overflowed(B)) return;
    const bool bp = clean_by_product(B);
    uint64_t direct = 0, diff = 0;
    for (uint32_t r0 = blockIdx.x * blockDim.x; r0 < B.n; r0 += gsize()) {
        const uint32_t r = r0 + threadIdx.x;
        if (r >= B.n) continue;
        const uint32_t L = B.len[r];
        uint32_t f0 = B.nfr[r], f1 = B.nfr[r + 1];
        if (f1 > B.fcap) f1 = B.fcap;
        uint64_t kept = 0;
        bool whole = false;
        uint32_t hs = 0, he = L;                       // what the clean tables hold of this read if it is `whole`
        if (bp) held_range(P, B, r, hs, he);
        for (uint32_t f = f0; f < f1; f++) {
            if (B.frag_flags[f] & TGSF_FF_REPEAT) continue;
            kept += B.frag_len[f];
            whole = (f1 - f0 == 1) && he > hs && B.frag_len[f] == he - hs && (uint32_t)B.frag_start[f] == hs;
        }
        if (L == 0 || P.only_qc) whole = false;
        B.whole[r] = whole ? 1u : 0u;
        direct += kept;
        if (!whole) diff += kept + (he > hs ? he - hs : 0u);
    }
    wave_add_u64(&B.plan[2], direct);
    wave_add_u64(&B.plan[3], diff);
}

// How the batch's speculation fared decides whether the next batch of this context speculates.  Since round 6 the by-product
// costs the raw pass little (the split-bin form: one pass over the bytes, 2.1 -> 2.3-2.4 ms on the C2 shape, 1.8 -> ~2 ms on
// C3's; profiles/r06_clean_tables_ab.txt) -- round 5's form ran the SWAR column twice over every speculated read and paid only
// where nearly every read was kept as expected.  What it still costs is the second look at the reads that turned out
// otherwise: their speculated range is taken back out and their real fragments put in, about twice their bases, where the
// direct way scans every kept fragment once.  So: go on while those bases (plan[3]) stay below three quarters of what
// scanning every fragment costs (plan[2]).  (A batch that did not speculate tries again after a while.)
TGSF_KERNEL k_clean_plan_next(DevBatch B)
{
    if (gtid() != 0 || !B.bp_allowed || B.clean_force) return;
    if (pool_overflowed(B)) return;
    if (clean_by_product(B)) {
        const uint64_t direct = B.plan[2], again = B.plan[3];
        B.bp_state[0] = (4 * again <= 3 * direct) ? 1u : 0u;
        B.bp_state[2] = 0;
    } else if (++B.bp_state[2] >= 64u) { B.bp_state[0] = 1u; B.bp_state[2] = 0; }     // (inputs change: look again now and then)
}

// ctr tables += this batch's raw tallies (rows the batch reached only).  The CLEAN instance is the
// last user of raw_tab and leaves it zeroed for the next batch.
template <bool CLEAN>
TGSF_KERNEL k_fold_raw(DevParams P, DevBatch B)
{
    // (the clean instance runs behind the middle scan: a batch whose candidate pool overflowed is left alone, but the
    // batch's table is still handed back empty -- the next batch's raw pass adds to it, and the second run of this one)
    const bool left_alone = CLEAN && pool_overflowed(B);
    const bool add = !left_alone && (!CLEAN || (clean_by_difference(B) && !clean_by_product(B)));   // (by-product: the raw pass put them there itself)
    uint64_t rows = B.plan[0];
    if (rows > P.n_bins) rows = P.n_bins;
    const size_t nw = (size_t)rows * 5, stride = (size_t)P.n_bins * 5;
    uint64_t* tq = B.ctr + ctr_bin_table(CLEAN ? TGSF_B_CLEAN_QUAL : TGSF_B_RAW_QUAL, P.bc_len, P.n_bins);
    uint64_t* tc = B.ctr + ctr_bin_table(CLEAN ? TGSF_B_CLEAN_CNT : TGSF_B_RAW_CNT, P.bc_len, P.n_bins);
    for (size_t i = gtid(); i < nw; i += gsize()) {
        const uint64_t q = B.raw_tab[i], c = B.raw_tab[stride + i];
        if (add && (c | q)) { tq[i] += q; tc[i] += c; }
        if (CLEAN && (c | q)) { B.raw_tab[i] = 0; B.raw_tab[stride + i] = 0; }
    }
}

// Items of the clean stats pass: the fragments (keep regions), plus -- by difference -- the reads to take out.
TGSF_KERNEL k_frag_prepare(DevParams P, DevBatch B)
{
    if (pool_overflowed(B)) return;
    TGSF_SHARED uint32_t h[kHistLds];
    const uint32_t seg_stride = B.max_tiles + 2;
    const uint32_t nbuck = seg_stride * stats_segments<true>(B);
    const bool use_lds = nbuck <= kHistLds;
    if (use_lds) for (uint32_t i = TGSF_COOP_BEGIN; i < nbuck; i += TGSF_COOP_STRIDE) h[i] = 0;
    TGSF_BLOCK_SYNC();
    const uint32_t nf = stored_frags(B);
    const bool diff = clean_by_difference(B);
    const uint32_t ni = nf + (diff ? B.n : 0u);
    uint32_t rows = 0;
    for (uint32_t i = gtid(); i < ni; i += gsize()) {
        const uint32_t item = clean_item(B, i, nf);
        if (i < nf && !(B.frag_flags[i] & TGSF_FF_REPEAT)) {
            uint32_t rw = B.frag_len[i] / kBin + 1;
            rows = rw > rows ? rw : rows;
        }
        const uint32_t L = stats_item_len<true>(P, B, item, diff);
        if (!L) continue;
        const uint32_t v = (L + kTileBases - 1) / kTileBases + stats_segment_of<true>(B, item) * seg_stride;
        if (use_lds) atomicAdd(&h[v], 1u); else atomicAdd(&B.tile_hist[v], 1u);
    }
    wave_max_u64(&B.ctr[TGSF_CTR_ROWS + 1], rows);
    TGSF_BLOCK_SYNC();
    if (use_lds)
        for (uint32_t i = TGSF_COOP_BEGIN; i < nbuck; i += TGSF_COOP_STRIDE)
            if (h[i]) atomicAdd(&B.tile_hist[i], h[i]);
}

// ---------------------------------------------------------------------------
// Counting sort of items by tile count, descending, so that "items with more
// than t tiles" is the prefix perm[0..cnt[t]).
//   hist[v]  #items with exactly v tiles
//   cnt[t]   #items with more than t tiles  (= first slot of value t in perm)
//   base[t]  #work items (item,tile) with tile index < t;  base[max_tiles+1] = total
// One block; thread i owns a contiguous chunk of buckets; chunk sums are combined
// by thread 0 (a few hundred buckets in all).
// ---------------------------------------------------------------------------
TGSF_KERNEL k_tile_scan(DevBatch B, uint32_t nseg)
{
    TGSF_SHARED uint32_t part[1024];
    const uint32_t mt = B.max_tiles;
    const uint32_t nb = mt + 1;                          // buckets 0..mt
    const uint32_t T = blockDim.x;
    const uint32_t per = (nb + T - 1) / T;
    const uint32_t lo = threadIdx.x * per;
    uint32_t hi = lo + per;
    if (hi > nb) hi = nb;
    for (uint32_t sg = 0; sg < nseg; sg++) {
        uint32_t* const hist = B.tile_hist + sg * (mt + 2);
        uint32_t* const cnt = B.tile_cnt + sg * (mt + 2);
        uint32_t* const base = B.tile_base + sg * (mt + 2);
        // suffix sums: cnt[v] = sum_{u>v} hist[u]
        uint32_t s = 0;
        for (uint32_t v = lo; v < hi; v++) s += hist[v];
        part[threadIdx.x] = s;
        TGSF_BLOCK_SYNC();
        if (threadIdx.x == 0) { uint32_t run = 0; for (int i = (int)T - 1; i >= 0; i--) { uint32_t x = part[i]; part[i] = run; run += x; } }
        TGSF_BLOCK_SYNC();
        uint32_t run = part[threadIdx.x];                    // sum of hist over all buckets above this chunk
        uint32_t csum = 0;
        for (int v = (int)hi - 1; v >= (int)lo; v--) { uint32_t hv = hist[v]; cnt[v] = run; csum += run; run += hv; }
        if (threadIdx.x == 0) cnt[mt + 1] = 0;
        TGSF_BLOCK_SYNC();
        // prefix sums of cnt: base[t] = sum_{u<t} cnt[u]
        part[threadIdx.x] = csum;
        TGSF_BLOCK_SYNC();
        if (threadIdx.x == 0) { uint32_t acc = 0; for (uint32_t i = 0; i < T; i++) { uint32_t x = part[i]; part[i] = acc; acc += x; } base[mt + 1] = acc; }
        TGSF_BLOCK_SYNC();
        uint32_t acc = part[threadIdx.x];
        for (uint32_t v = lo; v < hi; v++) { base[v] = acc; acc += cnt[v]; }
        TGSF_BLOCK_SYNC();
    }
    if (threadIdx.x == 0) {
        // (cnt[0] counts the items with more than 0 tiles: all of the segment's that take part)
        const uint32_t w0 = B.tile_base[mt + 1], w1 = nseg > 1 ? B.tile_base[(mt + 2) + mt + 1] : 0u;
        B.seg_info[0] = B.tile_cnt[0];
        B.seg_info[1] = w0;
        B.seg_info[2] = w0 + w1;
    }
}

template <bool CLEAN>
TGSF_KERNEL k_tile_scatter(DevParams P, DevBatch B)
{
    if (CLEAN && pool_overflowed(B)) return;
    TGSF_SHARED uint32_t h[kHistLds];
    TGSF_SHARED uint32_t hb[kHistLds];
    const uint32_t seg_stride = B.max_tiles + 2;
    const uint32_t nbuck = seg_stride * stats_segments<CLEAN>(B);
    const bool use_lds = nbuck <= kHistLds;
    const uint32_t nf = CLEAN ? stored_frags(B) : 0u;
    const bool diff = CLEAN && clean_by_difference(B);
    const uint32_t n = CLEAN ? nf + (diff ? B.n : 0u) : B.n;
    for (uint32_t i0 = blockIdx.x * blockDim.x; i0 < n; i0 += gsize()) {     // whole blocks stay in step
        const uint32_t i = i0 + threadIdx.x;
        if (use_lds) for (uint32_t k = TGSF_COOP_BEGIN; k < nbuck; k += TGSF_COOP_STRIDE) h[k] = 0;
        TGSF_BLOCK_SYNC();
        uint32_t L = 0, v = 0, local = 0, item = 0, first = 0;
        if (i < n) {
            item = CLEAN ? clean_item(B, i, nf) : i;
            L = stats_item_len<CLEAN>(P, B, item, diff);
            const uint32_t sg = stats_segment_of<CLEAN>(B, item);
            v = (L + kTileBases - 1) / kTileBases + sg * seg_stride;
            first = sg ? B.seg_info[0] : 0u;                  // where the segment's items begin in perm
            if (L && use_lds) local = atomicAdd(&h[v], 1u);
        }
        TGSF_BLOCK_SYNC();
        if (use_lds)
            for (uint32_t k = TGSF_COOP_BEGIN; k < nbuck; k += TGSF_COOP_STRIDE)
                if (h[k]) hb[k] = atomicAdd(&B.tile_fill[k], h[k]);
        TGSF_BLOCK_SYNC();
        if (L) {
            const uint32_t slot = first + (use_lds ? B.tile_cnt[v] + hb[v] + local
                                                   : B.tile_cnt[v] + atomicAdd(&B.tile_fill[v], 1u));
            B.perm[slot] = item;
        }
        TGSF_BLOCK_SYNC();
    }
}

// ---------------------------------------------------------------------------
// k_stats: CalcAvgQuality (src/TGSFilter.cpp:1436-1479) for every item.
//
// Work item = (item, tile of 64 bins x 100 bases).  Items are visited tile-index
// major, so consecutive work items of a wave hit the SAME 64 table rows: each
// lane owns one bin and keeps its 10 tallies in registers across items, and the
// tables see one atomic per lane per tile-index change instead of one per tile.
// k_build_work writes the (address, bases, item, tile) of every work item once,
// so a wave fetches the metadata of 64 work items with one coalesced load.
// The tile is staged through LDS with coalesced 16-byte loads (both streams,
// all 14 loads of a lane in flight together, and the NEXT tile's loads issued
// before the current one is reduced), then lane b reads its 100 bytes at stride
// 25 dwords (odd => bank-conflict free).
// Algorithmic traffic: 2 bytes per base, each read once.
// ---------------------------------------------------------------------------
constexpr int kStatsWaves = 4;
constexpr int kTileChunks = kTileBases / 16 + 2;   // +1 misalignment, +1 zero guard
constexpr int kLaneChunks = (kTileChunks + 63) / 64;   // 16-byte chunks a lane stages per stream

// work[2w] = { seq address lo, hi, bases | tile << 13, item },  work[2w+1] = { qual address lo, hi, speculated?, 0 }
// (whether the batch speculates on the item's read travels with the entry: k_stats needs it before it touches the tile, and a
// load of its own per tile sat in front of every tile's work -- measured on the C3 shape, round 6)
template <bool CLEAN>
TGSF_KERNEL k_build_work(DevParams P, DevBatch B)
{
    if (CLEAN && pool_overflowed(B)) return;
    const uint32_t mt = B.max_tiles;
    const uint32_t W0 = B.seg_info[1], W = B.seg_info[2], first1 = B.seg_info[0];
    const bool diff = CLEAN && clean_by_difference(B);
    (void)diff;
    for (uint32_t w = gtid(); w < W && w < B.work_cap; w += gsize()) {
        const uint32_t sg = w >= W0 ? 1u : 0u;                         // (segment 1: see stats_segments)
        const uint32_t* base = B.tile_base + sg * (mt + 2);
        const uint32_t ws = w - sg * W0;
        const uint32_t t = find_owner(base, mt + 1, ws);
        const uint32_t item = B.perm[sg * first1 + ws - base[t]];
        const bool frag = CLEAN && item < B.fcap;
        const uint32_t rd = CLEAN ? item - B.fcap : item;
        uint32_t hs = 0, he = frag ? 0u : stats_item_len<false>(P, B, rd, false);     // a read taken back out: the range the tables hold of it
        if (CLEAN && !frag) held_range(P, B, rd, hs, he);
        const uint32_t L = frag ? B.frag_len[item] : he - hs;
        const uint64_t a0 = (frag ? B.frag_off[item] : B.off[rd] + hs) + (uint64_t)t * kTileBases;
        uint32_t nb = L - t * kTileBases;
        if (nb > (uint32_t)kTileBases) nb = (uint32_t)kTileBases;
        const uint64_t aq = (frag ? B.frag_qoff[item] : B.qoff[rd] + hs) + (uint64_t)t * kTileBases;
        uint4 e, q;
        e.x = (uint32_t)a0; e.y = (uint32_t)(a0 >> 32); e.z = nb | (t << 13); e.w = item;
        q.x = (uint32_t)aq; q.y = (uint32_t)(aq >> 32); q.z = (!CLEAN && B.bp_allowed) ? B.spec[item] : 0u; q.w = 0;
        B.work[2 * (size_t)w] = e;
        B.work[2 * (size_t)w + 1] = q;
    }
}

// 51 KB of LDS per block already limits a CU to 3 blocks (3 waves per SIMD): take that register budget
// (the 14 staging registers per stream plus the work-list words spill at the default 128).
// NT: the text is fetched with non-temporal loads (each byte is wanted once by this kernel; what the later kernels of the
// batch re-read has long left the caches by then: a batch is GBs, the caches are MBs)
// BP (raw pass only): the clean tables as a by-product (DevBatch::spec), round 6's split-bin form.  A read the batch
// speculates on is expected to be kept as [head_trim, L - tail_trim): its clean bin j is the bytes [head_trim + 100 j, ...),
// i.e. with b = head_trim mod 100 and q = head_trim / 100 the TAIL piece [100 R + b, 100 R + 100) of raw bin R = q + j and
// the HEAD piece [100 (R+1), 100 (R+1) + b) of the next one.  So a lane tallies its raw bin of such a read in two pieces,
// split at b (launch-uniform) -- ONE pass over the bytes, two sets of ten tallies -- and when the tallies go to the tables
// raw row R gets head + tail, clean row R - q the tail and clean row R - q - 1 the head: no byte goes through the SWAR
// column twice (round 5's form ran the column again over the shifted window: the raw pass turned VALU-bound, 2.1 -> 3.5 ms),
// and nothing crosses lanes.  The raw pass covers such a read up to L - tail_trim only (stats_item_len; k_tail_fix tallies
// the rest), and the work list has the speculated reads first (DevBatch::tile_hist): a wave changes between the two kinds
// of item once at most, so both kinds share one set of accumulators.
template <bool CLEAN, bool NT = false, bool BP = false>
TGSF_KERNEL TGSF_BOUNDS(256, 3) k_stats(DevParams P, DevBatch B)
{
    static_assert(!(CLEAN && BP), "the by-product belongs to the raw pass");
    if (CLEAN && pool_overflowed(B)) return;
    TGSF_SHARED uint4 lds[kStatsWaves][2][kTileChunks];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t gw = blockIdx.x * kStatsWaves + wave, nw = gridDim.x * kStatsWaves;
    uint32_t W = B.seg_info[2];
    if (W > B.work_cap) W = B.work_cap;
    const uint32_t per = (W + nw - 1) / nw;
    uint32_t w0 = gw * per, w1 = w0 + per;
    if (w1 > W) w1 = W;
    if (w0 >= w1) return;

    // raw tallies go to the batch's own table first (k_fold_raw adds it to ctr, and to the clean tables
    // when those are tallied by difference)
    uint64_t* tab_q = CLEAN ? B.ctr + ctr_bin_table(TGSF_B_CLEAN_QUAL, P.bc_len, P.n_bins) : B.raw_tab;
    uint64_t* tab_c = CLEAN ? B.ctr + ctr_bin_table(TGSF_B_CLEAN_CNT, P.bc_len, P.n_bins) : B.raw_tab + (size_t)P.n_bins * 5;
    uint64_t* it_sum = CLEAN ? B.frag_sum : B.sumq;
    const int64_t qt = P.qtype;
    bool neg = false;                                  // accumulated tallies are to be taken OUT of the tables
    const QcConsts kc = qc_consts();

    // the lane's bin of the items gone through since the last flush -- (BP) of a speculated item: its tail piece
    uint32_t cnt[4] = {0, 0, 0, 0}, qs[5] = {0, 0, 0, 0, 0}, call = 0, since = 0;
    // (BP) ... and its head piece, the first bp_b bytes of the bin
    uint32_t hcnt[4] = {0, 0, 0, 0}, hqs[5] = {0, 0, 0, 0, 0}, hcall = 0;
    bool cur_sp = false;                               // (BP) the accumulated items are speculated ones
    const uint32_t bp_b = BP ? (uint32_t)P.head_trim % (uint32_t)kBin : 0u, bp_q = BP ? (uint32_t)P.head_trim / (uint32_t)kBin : 0u;
    uint64_t* ctab_q = B.ctr + ctr_bin_table(TGSF_B_CLEAN_QUAL, P.bc_len, P.n_bins);
    uint64_t* ctab_c = B.ctr + ctr_bin_table(TGSF_B_CLEAN_CNT, P.bc_len, P.n_bins);
    (void)hcall; (void)ctab_q; (void)ctab_c; (void)bp_b; (void)bp_q; (void)cur_sp;
    uint32_t t_acc = 0xFFFFFFFFu;
    uint32_t* S = reinterpret_cast<uint32_t*>(&lds[wave][0][0]);
    uint32_t* Qd = reinterpret_cast<uint32_t*>(&lds[wave][1][0]);

    // ten tallies into a row of a pair of tables
    auto put = [&](uint64_t* tc, uint64_t* tq, size_t row, const uint32_t* cn, const uint32_t* qv, uint32_t all, bool minus) TGSF_INLINE_LAMBDA {
#pragma unroll
        for (int c = 0; c < 4; c++) {
            if (cn[c]) {
                const int64_t dc = (int64_t)cn[c], dq = (int64_t)(qv[c] >> 7) - qt * (int64_t)cn[c];
                atomicAdd((ull*)&tc[row + c], (ull)(minus ? -dc : dc));
                atomicAdd((ull*)&tq[row + c], (ull)(minus ? -dq : dq));
            }
        }
        const int64_t dc = (int64_t)all, dq = (int64_t)qv[4] - qt * (int64_t)all;
        atomicAdd((ull*)&tc[row + 4], (ull)(minus ? -dc : dc));
        atomicAdd((ull*)&tq[row + 4], (ull)(minus ? -dq : dq));
    };
    auto flush = [&](uint32_t tt) {
        const uint32_t R = tt * kTileBins + lane;                      // the lane's row of the raw table
        if (BP && cur_sp && !B.replay) {
            // (a second run of the batch after a pool overflow: its first run has put these there already)
            if (call && R >= bp_q) put(ctab_c, ctab_q, (size_t)(R - bp_q) * 5, cnt, qs, call, false);
            if (hcall && R >= bp_q + 1u) put(ctab_c, ctab_q, (size_t)(R - bp_q - 1u) * 5, hcnt, hqs, hcall, false);
        }
        if (BP && hcall) {                                             // raw row R: head + tail
#pragma unroll
            for (int c = 0; c < 4; c++) { cnt[c] += hcnt[c]; qs[c] += hqs[c]; }       // (128 * sums of bytes: exact)
            qs[4] += hqs[4]; call += hcall;
        }
        if (call) put(tab_c, tab_q, (size_t)R * 5, cnt, qs, call, neg);
        cnt[0] = cnt[1] = cnt[2] = cnt[3] = 0;
        qs[0] = qs[1] = qs[2] = qs[3] = qs[4] = 0;
        call = 0; since = 0;
        if (BP) {
            hcnt[0] = hcnt[1] = hcnt[2] = hcnt[3] = 0;
            hqs[0] = hqs[1] = hqs[2] = hqs[3] = hqs[4] = 0;
            hcall = 0;
        }
    };

    // register staging of one tile: chunk c = lane + 64*k of each stream.  The two streams may sit at
    // different alignments (raw FASTQ text: seq and qual of a read are different lines of one buffer).
#if !defined(TGSF_EMUL)
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));   // plain vector values: stay in registers
    u32x4 rs[kLaneChunks], rq[kLaneChunks];
#endif
    // (a0, aq, nb are wave-uniform -- read with v_readlane from the work list -- so the chunk addresses are a
    // scalar base plus a per-lane offset.  Loads are not predicated: past the tile's last data chunk the
    // offset is clamped to that chunk (a line already being fetched) and the register is simply not stored.)
    auto issue = [&](uint64_t a0, uint64_t aq, uint32_t nb) TGSF_INLINE_LAMBDA {
#if !defined(TGSF_EMUL)
        const uint32_t shs = (uint32_t)(a0 & 15u), shq = (uint32_t)(aq & 15u);
        const uint8_t* ps = B.seq + (a0 - shs);
        const uint8_t* pq = B.qual + (aq - shq);
        const uint32_t lasts = (shs + nb - 1u) & ~15u, lastq = (shq + nb - 1u) & ~15u;   // nb >= 1
#pragma unroll
        for (int k = 0; k < kLaneChunks; k++) {
            const uint32_t cb = (lane + 64u * k) * 16u;
            const u32x4* as = reinterpret_cast<const u32x4*>(ps + (cb < lasts ? cb : lasts));
            const u32x4* aq4 = reinterpret_cast<const u32x4*>(pq + (cb < lastq ? cb : lastq));
            rs[k] = NT ? __builtin_nontemporal_load(as) : *as;
            if (P.no_qual) { const uint32_t w = (uint32_t)P.qtype * 0x01010101u; rq[k] = u32x4{w, w, w, w}; }   // q - qType = 0
            else rq[k] = NT ? __builtin_nontemporal_load(aq4) : *aq4;
        }
#else
        (void)a0; (void)aq; (void)nb;
#endif
    };
#if defined(TGSF_EMUL)
    auto mask_tail = [](uint4& v, uint32_t keep) {          // keep the first `keep` (1..15) bytes
        uint32_t* p = reinterpret_cast<uint32_t*>(&v);
#pragma unroll
        for (int d = 0; d < 4; d++) {
            int kb = (int)keep - 4 * d;
            p[d] &= kb >= 4 ? 0xFFFFFFFFu : (kb <= 0 ? 0u : ((1u << (8 * kb)) - 1u));
        }
    };
#endif
    auto commit = [&](uint64_t a0, uint64_t aq, uint32_t nb) TGSF_INLINE_LAMBDA {
        const uint32_t shs = (uint32_t)(a0 & 15u), shq = (uint32_t)(aq & 15u);
        const uint32_t ends = shs + nb, endq = shq + nb;
        const uint32_t nchs = (ends + 15u) / 16u + 1u, nchq = (endq + 15u) / 16u + 1u;   // + one all-zero guard chunk
#if !defined(TGSF_EMUL)
        // data chunks as loaded, then the bytes from the end of the tile to the end of the guard chunk are
        // zeroed in LDS by byte stores of lanes 0..31 (LDS operations of one wave complete in order)
#pragma unroll
        for (int k = 0; k < kLaneChunks; k++) {
            const uint32_t c = lane + 64u * k;
            if (c + 1u < nchs) *reinterpret_cast<u32x4*>(&lds[wave][0][c]) = rs[k];
            if (c + 1u < nchq) *reinterpret_cast<u32x4*>(&lds[wave][1][c]) = rq[k];
        }
        uint8_t* bs = reinterpret_cast<uint8_t*>(&lds[wave][0][0]);
        uint8_t* bq = reinterpret_cast<uint8_t*>(&lds[wave][1][0]);
        if (ends + lane < nchs * 16u) bs[ends + lane] = 0;
        if (endq + lane < nchq * 16u) bq[endq + lane] = 0;
#else
        // emulation: one lane at a time, so every emulated lane stages the whole tile
        for (uint32_t c = 0; c < nchs; c++) {
            uint4 v = {0, 0, 0, 0};
            const uint32_t cb = c * 16u;
            if (cb < ends) { v = *reinterpret_cast<const uint4*>(B.seq + (a0 - shs) + cb); if (cb + 16u > ends) mask_tail(v, ends - cb); }
            lds[wave][0][c] = v;
        }
        for (uint32_t c = 0; c < nchq; c++) {
            uint4 v = {0, 0, 0, 0};
            const uint32_t cb = c * 16u;
            if (cb < endq) {
                if (P.no_qual) { const uint32_t w = (uint32_t)P.qtype * 0x01010101u; v = uint4{w, w, w, w}; }
                else v = *reinterpret_cast<const uint4*>(B.qual + (aq - shq) + cb);
                if (cb + 16u > endq) mask_tail(v, endq - cb);
            }
            lds[wave][1][c] = v;
        }
#endif
    };

    for (uint32_t g0 = w0; g0 < w1; g0 += 64u) {
        const uint32_t ng = (w1 - g0) < 64u ? (w1 - g0) : 64u;
        uint4 me = {0, 0, 0, 0}, mq = {0, 0, 0, 0};
#if defined(TGSF_EMUL)
        (void)me; (void)mq;
#else
        if (lane < ng) { me = B.work[2 * (size_t)(g0 + lane)]; mq = B.work[2 * (size_t)(g0 + lane) + 1]; }
#endif
        auto entry = [&](uint32_t i, uint64_t& a0, uint64_t& aq, uint32_t& nb, uint32_t& tt, uint32_t& item, uint32_t& spz) {
#if defined(TGSF_EMUL)
            const uint4 e = B.work[2 * (size_t)(g0 + i)], q = B.work[2 * (size_t)(g0 + i) + 1];
            a0 = (uint64_t)e.x | ((uint64_t)e.y << 32); nb = e.z & 0x1FFFu; tt = e.z >> 13; item = e.w;
            aq = (uint64_t)q.x | ((uint64_t)q.y << 32); spz = q.z;
#else
            const uint32_t x = wave_pick(me.x, i), y = wave_pick(me.y, i), z = wave_pick(me.z, i);
            const uint32_t qx = wave_pick(mq.x, i), qy = wave_pick(mq.y, i);
            item = wave_pick(me.w, i);
            spz = BP ? wave_pick(mq.z, i) : 0u;
            a0 = (uint64_t)x | ((uint64_t)y << 32); nb = z & 0x1FFFu; tt = z >> 13;
            aq = (uint64_t)qx | ((uint64_t)qy << 32);
#endif
        };
        uint64_t a0, aq; uint32_t nb, tt, item, spz;
        entry(0, a0, aq, nb, tt, item, spz);
        issue(a0, aq, nb);
        for (uint32_t i = 0; i < ng; i++) {
            TGSF_WAVE_SYNC();                                  // previous tile fully consumed
            commit(a0, aq, nb);
            TGSF_WAVE_SYNC();
            const uint64_t ca0 = a0, caq = aq; const uint32_t cnb = nb, ctt = tt, citem = item, cspz = spz;
            if (i + 1 < ng) { entry(i + 1, a0, aq, nb, tt, item, spz); issue(a0, aq, nb); }   // in flight during the reduce
            const bool ineg = CLEAN && citem >= B.fcap;         // a read being taken back out
            const bool sp = BP && cspz != 0u;                   // (wave-uniform) a read the batch speculates on
            if (ctt != t_acc || ineg != neg || since >= 1024 || (BP && sp != cur_sp)) { // qs[c] carries 128*sum of bytes up to 255: stay below 2^32
                if (t_acc != 0xFFFFFFFFu) flush(t_acc);
                t_acc = ctt;
                neg = ineg;
                cur_sp = sp;
            }
            ++since;
            // nv bytes of the tile from byte `at` on (a lane's 100-base bin, or less at an item's end) into ten tallies;
            // returns the OR of the quality dwords
            auto bin = [&](uint32_t at, int nv, uint32_t* cn, uint32_t* qv) TGSF_INLINE_LAMBDA -> uint32_t {
                uint32_t qor = 0;
                const uint32_t bos = (uint32_t)(ca0 & 15u) + at, boq = (uint32_t)(caq & 15u) + at;
                const uint32_t ds = bos >> 2, bss = bos & 3u, dq = boq >> 2, bsq = boq & 3u;
                const int ndw = (nv + 3) >> 2;
                // A full bin (every tile but an item's last) runs a fixed 25 dwords, unrolled by 5: LDS reads
                // with immediate offsets, no per-dword loop bookkeeping.
                if ((bss | bsq) == 0) {
                    if (nv == kBin) {
#pragma unroll 5
                        for (int k = 0; k < kBin / 4; k++) {
                            const uint32_t q = Qd[dq + k];
                            qor |= q;
                            qc_accum4(kc, S[ds + k], q, cn, qv);
                        }
                    } else {
                        for (int k = 0; k < ndw; k++) {
                            uint32_t q = Qd[dq + k], sv = S[ds + k];
                            if (BP && k == ndw - 1 && (nv & 3)) {           // (a clean bin may end inside the read: bytes follow it)
                                const uint32_t m = (1u << (8 * (nv & 3))) - 1u;
                                q &= m; sv &= m;
                            }
                            qor |= q;
                            qc_accum4(kc, sv, q, cn, qv);
                        }
                    }
                } else {
                    // bytes past the item are zero in LDS; bytes before it never enter (the shift skips them)
                    uint32_t slo = S[ds], qlo = Qd[dq];
                    if (nv == kBin) {
#pragma unroll 5
                        for (int k = 0; k < kBin / 4; k++) {
                            const uint32_t shi = S[ds + k + 1], qhi = Qd[dq + k + 1];
                            const uint32_t q = alignbyte(qhi, qlo, bsq);
                            qor |= q;
                            qc_accum4(kc, alignbyte(shi, slo, bss), q, cn, qv);
                            slo = shi; qlo = qhi;
                        }
                    } else {
                        for (int k = 0; k < ndw; k++) {
                            const uint32_t shi = S[ds + k + 1], qhi = Qd[dq + k + 1];
                            uint32_t q = alignbyte(qhi, qlo, bsq), sv = alignbyte(shi, slo, bss);
                            if (BP && k == ndw - 1 && (nv & 3)) {           // (a clean bin may end inside the read: bytes follow it)
                                const uint32_t m = (1u << (8 * (nv & 3))) - 1u;
                                q &= m; sv &= m;
                            }
                            qor |= q;
                            qc_accum4(kc, sv, q, cn, qv);
                            slo = shi; qlo = qhi;
                        }
                    }
                }
                return qor;
            };
            // Quality bytes of 128 and above: the reference computes `qual[i] - qType` on a (signed) char
            // (src/TGSFilter.cpp:1455-1457), so such a byte stands for its value - 256.  The tallies took every byte as
            // unsigned (v_dot4_u32_u8); a bin that holds such bytes -- none in real data -- is put right on the spot: 256 per
            // high byte goes back out of the bin's row, class by class (sign: +1 where the tallies are being taken out).
            // Returns the number of high bytes (for the item's sum).
            auto high_bytes = [&](uint32_t at, int nv, uint64_t* tq, size_t row, bool minus) -> int32_t {
                const uint32_t bos = (uint32_t)(ca0 & 15u) + at, boq = (uint32_t)(caq & 15u) + at;
                const uint32_t ds = bos >> 2, bss = bos & 3u, dq = boq >> 2, bsq = boq & 3u;
                uint32_t hc[4] = {0, 0, 0, 0}, hq[5] = {0, 0, 0, 0, 0};
                for (int k = 0; k < ((nv + 3) >> 2); k++) {
                    uint32_t q = alignbyte(Qd[dq + k + 1], Qd[dq + k], bsq) & 0x80808080u;
                    if (k == ((nv + 3) >> 2) - 1 && (nv & 3)) q &= (1u << (8 * (nv & 3))) - 1u;
                    qc_accum4(kc, alignbyte(S[ds + k + 1], S[ds + k], bss), q >> 7, hc, hq);      // (a "quality" of 1 per high byte)
                }
#pragma unroll
                for (int c = 0; c < 5; c++) {
                    const int64_t d = 256 * (int64_t)(c < 4 ? (hq[c] >> 7) : hq[4]);
                    if (d && tq) atomicAdd((ull*)&tq[row + c], (ull)(minus ? d : -d));
                }
                return (int32_t)hq[4];
            };
            const int nvalid = (int)cnb - (int)lane * kBin;     // bases of this lane's bin in the tile
            const uint32_t q4_before = qs[4] + (BP ? hqs[4] : 0u), q4t_before = qs[4], q4h_before = BP ? hqs[4] : 0u;
            (void)q4t_before; (void)q4h_before;
            int nv = 0, nh = 0;                                 // bytes of the bin; (BP, a speculated read) of its head piece
            uint32_t qor = 0, qorh = 0;                         // OR of the quality dwords of the (tail piece of the) bin / of the head piece
            if (nvalid > 0) {
                nv = nvalid > kBin ? kBin : nvalid;
                if (BP && sp && bp_b) {
                    nh = nv < (int)bp_b ? nv : (int)bp_b;
                    // (two calls of the general loop.  Built and measured beside this, round 6, with the same raw-pass time either way: one
                    // pass over the bin's 25 dwords that changes tallies at the split; and, for a piece of a few bytes (C3's 5' trim of
                    // 7), the whole bin through the plain pass's unrolled loop + the small piece once more, the other piece as the difference)
                    qorh = bin(lane * kBin, nh, hcnt, hqs);
                    if (nv > nh) qor = bin(lane * kBin + bp_b, nv - nh, cnt, qs);
                    hcall += (uint32_t)nh;
                    call += (uint32_t)(nv - nh);
                } else {
                    qor = bin(lane * kBin, nv, cnt, qs);
                    call += (uint32_t)nv;
                }
            }
            const uint32_t R = ctt * kTileBins + lane;
            int32_t high_all = 0;
            const bool any_high = wave_any(((qor | qorh) & 0x80808080u) != 0u);          // (one ballot: no lane-to-lane reduction)
            if (any_high && nv > 0 && ((qor | qorh) & 0x80808080u))
                high_all = high_bytes(lane * kBin, nv, tab_q, (size_t)R * 5, ineg);
            // sumQ of the item: sum over the tile of (qual - qType), two's complement in u64 (:1457-1458)
            const int32_t part = (int32_t)(qs[4] + (BP ? hqs[4] : 0u) - q4_before) - (int32_t)qt * nv - 256 * high_all;
            const int32_t tot = wave_sum_i32(part);
            if (BP && sp) {
                // the same bytes as clean tallies: the tail piece belongs to clean row R - q, the head piece to row R - q - 1
                // (pieces in front of head_trim -- rows below those -- belong to no clean row); their share of the read's clean sum
                const bool t_in = R >= bp_q, h_in = R >= bp_q + 1u;
                int32_t chigh = 0;
                if (any_high && nv > 0) {                                // (quality bytes of 128 and above: 256 each comes back out, as in the raw row)
                    uint64_t* const ct = B.replay ? nullptr : ctab_q;
                    if (nh > 0 && h_in && (qorh & 0x80808080u)) chigh += high_bytes(lane * kBin, nh, ct, (size_t)(R - bp_q - 1u) * 5, false);
                    if (nv > nh && t_in && (qor & 0x80808080u)) chigh += high_bytes(lane * kBin + (uint32_t)nh, nv - nh, ct, (size_t)(R - bp_q) * 5, false);
                }
                int32_t cpart = -256 * chigh;
                if (t_in) cpart += (int32_t)(qs[4] - q4t_before) - (int32_t)qt * (nv - nh);
                if (h_in) cpart += (int32_t)(hqs[4] - q4h_before) - (int32_t)qt * nh;
                const int32_t ctot = wave_sum_i32(cpart);
                if (wave_leader()) atomicAdd((ull*)&B.spec_sum[citem], (ull)(int64_t)ctot);
            }
            if (!ineg && wave_leader()) atomicAdd((ull*)&it_sum[citem], (ull)(int64_t)tot);
        }
    }
    if (t_acc != 0xFFFFFFFFu) flush(t_acc);
}

// ---------------------------------------------------------------------------
// k_tail_fix: the raw pass of a read the batch speculates on stops at L - tail_trim (stats_item_len): the tail_trim bytes behind
// are tallied here, one lane a read, into the batch's raw table and the read's quality sum (CalcAvgQuality,
// src/TGSFilter.cpp:1436-1479, on those bytes).  A few bytes per read: the 3' trims the pre-pass finds are below 150.
// ---------------------------------------------------------------------------
// The reads of a batch are of similar length where such a trim exists (HiFi: 3' trims of a few bases), so their tails fall
// into a few dozen rows of the table: tallied straight into it, 262 144 reads spent 0.53 ms queueing at those rows' atomics.
// A workgroup tallies into LDS first (tables of up to kTailRows rows; longer tables -- reads spread over thousands of rows --
// go straight to memory) and adds what it gathered once.
constexpr int kTailRows = 1024;
TGSF_KERNEL k_tail_fix(DevParams P, DevBatch B)
{
    TGSF_SHARED int32_t acc[kTailRows][10];            // [row][5 counts, 5 sums of (quality - qType)]
    if (!B.bp_allowed || P.tail_trim <= 0) return;
    uint64_t* const tab_q = B.raw_tab;
    uint64_t* const tab_c = B.raw_tab + (size_t)P.n_bins * 5;
    const bool in_lds = P.n_bins <= (uint32_t)kTailRows;
    int32_t* const flat = &acc[0][0];
    if (in_lds) for (uint32_t i = TGSF_COOP_BEGIN; i < (uint32_t)kTailRows * 10u; i += TGSF_COOP_STRIDE) flat[i] = 0;
    TGSF_BLOCK_SYNC();
    for (uint32_t r = gtid(); r < B.n; r += gsize()) {
        const uint32_t L = B.len[r];
        if (!L || !B.spec[r]) continue;
        const uint8_t* sq = B.seq + B.off[r];
        const uint8_t* ql = B.qual + B.qoff[r];
        int64_t sum = 0;
        for (uint32_t i = L - (uint32_t)P.tail_trim; i < L; i++) {
            const uint32_t row = i / (uint32_t)kBin;
            const uint32_t b = (uint32_t)sq[i] & 0xDFu;
            const int c = (sq[i] & 0x80u) ? 4 : b == 'A' ? 0 : b == 'T' ? 1 : b == 'G' ? 2 : b == 'C' ? 3 : 4;
            // `qual[i] - qType` on a signed char (:1455-1457); records without qualities: 0
            const int32_t q = P.no_qual ? 0 : (int32_t)(int8_t)ql[i] - (int32_t)P.qtype;
            if (in_lds) {
                if (c < 4) { atomicAdd(&acc[row][c], 1); atomicAdd(&acc[row][5 + c], q); }
                atomicAdd(&acc[row][4], 1); atomicAdd(&acc[row][9], q);
            } else {
                if (c < 4) { atomicAdd((ull*)&tab_c[(size_t)row * 5 + c], (ull)1); atomicAdd((ull*)&tab_q[(size_t)row * 5 + c], (ull)(int64_t)q); }
                atomicAdd((ull*)&tab_c[(size_t)row * 5 + 4], (ull)1); atomicAdd((ull*)&tab_q[(size_t)row * 5 + 4], (ull)(int64_t)q);
            }
            sum += q;
        }
        atomicAdd((ull*)&B.sumq[r], (ull)sum);
    }
    TGSF_BLOCK_SYNC();
    if (in_lds)
        for (uint32_t i = TGSF_COOP_BEGIN; i < P.n_bins * 5u; i += TGSF_COOP_STRIDE) {
            const uint32_t row = i / 5u, c = i % 5u;
            const int32_t cn = acc[row][c];
            if (!cn) continue;
            atomicAdd((ull*)&tab_c[(size_t)row * 5 + c], (ull)(int64_t)cn);
            atomicAdd((ull*)&tab_q[(size_t)row * 5 + c], (ull)(int64_t)acc[row][5 + c]);
        }
}

// ---------------------------------------------------------------------------
// k_gate_reads: the raw mean-quality gate (src/TGSFilter.cpp:1943, :1946-1953) and
// the number of middle-scan segments of each surviving read.
// ---------------------------------------------------------------------------
TGSF_KERNEL k_gate_reads(DevParams P, DevBatch B)
{
    TGSF_SHARED ull hq[TGSF_N_QBINS];
    for (uint32_t i = TGSF_COOP_BEGIN; i < (uint32_t)TGSF_N_QBINS; i += TGSF_COOP_STRIDE) hq[i] = 0;
    TGSF_BLOCK_SYNC();
    uint64_t lowq_reads = 0, lowq_bases = 0;
    for (uint32_t r0 = blockIdx.x * blockDim.x; r0 < B.n; r0 += gsize()) {   // whole waves stay convergent
        const uint32_t r = r0 + threadIdx.x;
        if (r < B.n && B.len[r]) {
            const uint32_t L = B.len[r];
            const double mq = mean_q(B.sumq[r], L);
            if (!(mq >= 0.0 && mq < 256.0)) { set_status(B, DS_BAD_MEANQ, r); }
            else {
                if (!P.no_qual && !B.replay) atomicAdd(&hq[(int)mq], (ull)L);   // :1943 (records with qualities only, :1941)
                uint32_t segs = 0, chunks = 0;
                if (P.filter) {
                    if (!P.no_qual && q_fail(mq, P.min_q, P.max_q)) {
                        B.flags[r] = TGSF_RF_LOWQ;
                        if (!B.replay) { lowq_reads++; lowq_bases += L; }
                    } else {
                        int ML = (int)L - 2 * P.end_len;              // :1236 tsmLen
                        if (ML >= P.min_Q) {
                            // segments are seg_cols-aligned blocks of the absolute address space
                            const uint64_t a0 = (uint64_t)(uintptr_t)B.seq + B.off[r] + (uint64_t)P.end_len;
                            const uint64_t S = (uint64_t)P.seg_cols;
                            segs = (uint32_t)((a0 + (uint64_t)ML - 1) / S - a0 / S + 1);
                            chunks = ((uint32_t)ML + 15u) >> 4;           // k_mid_flat: 16 columns at a time from the window's start
                        }
                    }
                }
                B.seg_cnt[r] = segs;
                B.chk_cnt[r] = chunks;
            }
        }
    }
    wave_add_u64(&B.ctr[TGSF_CTR_DROPINFO + 0], lowq_reads);
    wave_add_u64(&B.ctr[TGSF_CTR_DROPINFO + 1], lowq_bases);
    TGSF_BLOCK_SYNC();
    for (uint32_t i = TGSF_COOP_BEGIN; i < (uint32_t)TGSF_N_QBINS; i += TGSF_COOP_STRIDE)
        if (hq[i]) atomicAdd((ull*)&B.ctr[TGSF_CTR_RAW_DIFFQ + i], hq[i]);
}

// ---------------------------------------------------------------------------
// k_scan_u32: exclusive prefix sum in place over a[0..n], a[n] receives the total.
// One block; each thread owns 16 consecutive elements per sweep.
// ---------------------------------------------------------------------------
// Exclusive prefix sums of a[0..n) in place, a[n] = total (n = the reads of a batch, ~1e5: the cost is latency, not
// work).  Three small launches instead of one block walking the whole array: every block scans its own kScanTile
// entries and leaves its total in part[]; one block scans the totals; every block adds its offset.
constexpr int kScanTile = 4096;
TGSF_KERNEL k_scan_tiles(uint32_t* a, uint32_t n, uint32_t* part)
{
    TGSF_SHARED uint32_t sums[1024];
    const uint32_t T = blockDim.x, per = (kScanTile + T - 1) / T;
    const uint32_t base = blockIdx.x * kScanTile;
    const uint32_t lo = base + threadIdx.x * per;
    uint32_t hi = lo + per;
    if (hi > base + kScanTile) hi = base + kScanTile;
    if (hi > n) hi = n;
    uint32_t s = 0;
    for (uint32_t i = lo; i < hi; i++) s += a[i];
    sums[threadIdx.x] = s;
    TGSF_BLOCK_SYNC();
    if (threadIdx.x == 0) {                                 // T <= 256 sums: a serial pass is as good as anything
        uint32_t acc = 0;
        for (uint32_t t = 0; t < T; t++) { const uint32_t x = sums[t]; sums[t] = acc; acc += x; }
        part[blockIdx.x] = acc;
    }
    TGSF_BLOCK_SYNC();
    uint32_t acc = sums[threadIdx.x];
    for (uint32_t i = lo; i < hi; i++) { const uint32_t x = a[i]; a[i] = acc; acc += x; }
}
TGSF_KERNEL k_scan_top(uint32_t* part, uint32_t nb, uint32_t* total_out)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    uint32_t acc = 0;
    for (uint32_t b = 0; b < nb; b++) { const uint32_t x = part[b]; part[b] = acc; acc += x; }
    *total_out = acc;
}
TGSF_KERNEL k_scan_add(uint32_t* a, uint32_t n, const uint32_t* part)
{
    const uint32_t T = blockDim.x, per = (kScanTile + T - 1) / T;
    const uint32_t base = blockIdx.x * kScanTile, off = part[blockIdx.x];
    if (!off) return;
    const uint32_t lo = base + threadIdx.x * per;
    uint32_t hi = lo + per;
    if (hi > base + kScanTile) hi = base + kScanTile;
    if (hi > n) hi = n;
    for (uint32_t i = lo; i < hi; i++) a[i] += off;
}

// ---------------------------------------------------------------------------
// k_end_tables: Get_5p_base_qual / Get_3p_base_qual (src/TGSFilter.cpp:1481-1575).
// Lane = position; a wave walks a strided set of items and keeps its tallies in
// lane-private LDS rows, so the [bc_len][5] tables see one atomic per lane-slot
// per wave, not per read.
// ---------------------------------------------------------------------------
constexpr int kEndWaves = 8;       // waves of a block share one 40 KB set of LDS tallies (LDS atomics)
constexpr int kMaxBcLen = 512;     // positions one launch tallies in LDS; a larger -e takes one launch per slab of 512 positions
constexpr int kMaxBcLenTotal = 1 << 20;
TGSF_D int base_col(uint32_t b) {
    b &= 0xDFu;
    return b == 'A' ? 0 : b == 'T' ? 1 : b == 'G' ? 2 : b == 'C' ? 3 : 4;
}
template <bool CLEAN>
TGSF_KERNEL k_end_tables(DevParams P, DevBatch B, uint32_t slab)
{
    if (CLEAN && pool_overflowed(B)) return;
    // [end][slot][10][lane]: 5 counts, 5 quality sums; position = slot*64 + lane
    TGSF_SHARED uint32_t acc[2][kMaxBcLen / 64][10][64];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t gw = blockIdx.x * kEndWaves + wave, nw = gridDim.x * kEndWaves;
    const uint32_t n = CLEAN ? stored_frags(B) : B.n;
    const uint32_t bc = (uint32_t)P.bc_len;
    const uint32_t p0 = slab * (uint32_t)kMaxBcLen;                      // first position of this launch's slab
    const uint32_t span = bc - p0 < (uint32_t)kMaxBcLen ? bc - p0 : (uint32_t)kMaxBcLen;
    const uint32_t slots = (span + 63u) / 64u;
    uint32_t* flat = &acc[0][0][0][0];
    for (uint32_t i = TGSF_COOP_BEGIN; i < 2u * (kMaxBcLen / 64) * 10u * 64u; i += TGSF_COOP_STRIDE) flat[i] = 0;
    TGSF_BLOCK_SYNC();
    auto add = [&](uint32_t e, uint32_t s, uint32_t b, uint32_t q) {
        int c = base_col(b);
        // count-only variants as coded: 'g' is not a G at the 5' end (:1629), 't' is not a T at the 3' end (:1667)
        if (P.no_qual && b == (e ? (uint32_t)'t' : (uint32_t)'g')) c = 4;
        if (q & 0x80u) q -= 256u;      // `qual[i] - qType` on a signed char (:1508): a byte of 128 and above stands for its value - 256
        if (c < 4) { atomicAdd(&acc[e][s][c][lane], 1u); atomicAdd(&acc[e][s][5 + c][lane], q); }
        atomicAdd(&acc[e][s][4][lane], 1u); atomicAdd(&acc[e][s][9][lane], q);
    };
    for (uint32_t i = gw; i < n; i += 2 * nw) {                 // two items in flight per wave
        uint32_t L[2] = {0, 0}; uint64_t off[2] = {0, 0}, qof[2] = {0, 0};
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const uint32_t it = i + u * nw;
            if (it < n) {
                if (CLEAN) { if (B.frag_flags[it] & TGSF_FF_PASS) { L[u] = B.frag_len[it]; off[u] = B.frag_off[it]; qof[u] = B.frag_qoff[it]; } }
                else { L[u] = B.len[it]; off[u] = B.off[it]; qof[u] = B.qoff[it]; }
            }
        }
        for (uint32_t s = 0; s < slots; s++) {
            const uint32_t p = p0 + s * 64u + lane;
            uint32_t b5[2], q5[2], b3[2], q3[2];
            bool on[2];
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const uint32_t m = bc < L[u] ? bc : L[u];
                on[u] = p < m;
                if (on[u]) {
                    b5[u] = B.seq[off[u] + p];
                    b3[u] = B.seq[off[u] + L[u] - 1 - p];                                          // :1554-1557
                    if (P.no_qual) q5[u] = q3[u] = (uint32_t)P.qtype;
                    else { q5[u] = B.qual[qof[u] + p]; q3[u] = B.qual[qof[u] + L[u] - 1 - p]; }
                }
            }
#pragma unroll
            for (int u = 0; u < 2; u++) if (on[u]) { add(0, s, b5[u], q5[u]); add(1, s, b3[u], q3[u]); }
        }
    }
    TGSF_BLOCK_SYNC();
    const int64_t qt = P.qtype;
    for (uint32_t k = TGSF_COOP_BEGIN; k < 2u * slots * 5u * 64u; k += TGSF_COOP_STRIDE) {
        const uint32_t ln = k & 63u, c = (k >> 6) % 5u, s = ((k >> 6) / 5u) % slots, e = (k >> 6) / (5u * slots);
        const uint32_t p = p0 + s * 64u + ln;
        if (p >= bc) continue;
        const uint32_t cn = acc[e][s][c][ln];
        if (!cn) continue;
        uint64_t* tq = B.ctr + ctr_end_table((CLEAN ? 4 : 0) + (e ? 2 : 0), P.bc_len);
        uint64_t* tc = B.ctr + ctr_end_table((CLEAN ? 4 : 0) + (e ? 2 : 0) + 1, P.bc_len);
        atomicAdd((ull*)&tc[(size_t)p * 5 + c], (ull)cn);
        atomicAdd((ull*)&tq[(size_t)p * 5 + c], (ull)((int64_t)(int32_t)acc[e][s][5 + c][ln] - qt * (int64_t)cn));   // (a signed sum: see add)
    }
}

// ---------------------------------------------------------------------------
// Alignment of one adapter against one short window with edlib's HW/PATH
// semantics as TGSFilter consumes them.
//
// mlen = alignmentLength - editDistance is the number of match columns on the
// path edlib's traceback picks.  With T' = end0-start0+1 window columns and
// `ins` up-moves on that path:  mlen = T' - best + ins,  0 <= ins - max(0,Q-T')
// and 2*ins <= best - (T'-Q).  When those bounds already decide "mlen >= need"
// the path is not needed; otherwise it is recovered bit-parallel: a global-mode
// Myers pass stores the vertical (+1) and horizontal (+1) delta words of every
// column, and the traceback (priority up > left > diagonal, edlib.cpp:1023/
// 1057/1088) is a walk over single bits: "up" iff Pv_j[i], "left" iff Ph_j[i].
// The per-column words live in a lane-interleaved scratch ([column][lane]).
// ---------------------------------------------------------------------------
struct WinAln {
    int best;          // -1: nothing within k
    int n;
    int first_end, last_end;
    int start0;
    int mlen;          // alignmentLength - editDistance of the first location
};

// one text column in global mode (hin = +1 at the top); returns the horizontal +1 words
template <int NW>
TGSF_HD void bv_step_global(Bv<NW>& s, const uint64_t* eq, uint64_t* ph_out, int Q) {
    int hin = 1;
#pragma unroll
    for (int w = 0; w < NW; w++) {
        uint64_t Eq = eq[w], Pv = s.p[w], Mv = s.m[w];
        uint64_t Xv = Eq | Mv;
        if (hin < 0) Eq |= 1ull;
        uint64_t Xh = (((Eq & Pv) + Pv) ^ Pv) | Eq;
        uint64_t Ph = Mv | ~(Xh | Pv);
        uint64_t Mh = Pv & Xh;
        ph_out[w] = Ph;
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

// lane-private scratch: word k of column j of this lane
struct LaneScratch {
    uint64_t* base;       // &scratch[wave region][lane]
    TGSF_HD uint64_t& at(int j, int k, int nwords) const { return base[((size_t)j * nwords + k) * 64]; }
};

// number of columns of the canonical global alignment of the adapter against t[0..T)
template <int NW>
TGSF_D int path_len_bv(const uint64_t* pf /*[256][2]*/, int Q, const uint8_t* t, int T, LaneScratch sc)
{
    Bv<NW> s;
    bv_init(s, Q);
    for (int j = 1; j <= T; j++) {
        uint64_t ph[NW];
        bv_step_global<NW>(s, pf + (size_t)t[j - 1] * kPeqW, ph, Q);
#pragma unroll
        for (int w = 0; w < NW; w++) { sc.at(j, w, 2 * NW) = s.p[w]; sc.at(j, NW + w, 2 * NW) = ph[w]; }
    }
    int i = Q, j = T, len = 0;
    while (i > 0 && j > 0) {
        const int r = i - 1;
        const uint64_t pv = sc.at(j, r >> 6, 2 * NW);
        if ((pv >> (r & 63)) & 1ull) { i--; }
        else {
            const uint64_t ph = sc.at(j, NW + (r >> 6), 2 * NW);
            if ((ph >> (r & 63)) & 1ull) j--; else { i--; j--; }
        }
        len++;
    }
    return len + i + j;        // straight along the border (edlib.cpp:1028-1032, :1062-1067)
}


// The bytes of a window, sixteen per load (any alignment; whole chunks only, the tail byte by byte): fn(j, byte).
// One lane walks its own window, so a load per byte would put the memory latency on every column.  (Fetching the
// sixteen Eq rows of a chunk ahead as well was measured slower: 1.9 ms against 0.8 for k_end_windows.)
template <class F>
TGSF_D void each_byte_fwd(const uint8_t* t, int T, F fn)
{
    int j = 0;
    for (; j + 16 <= T; j += 16) {
        uint4 v;
        __builtin_memcpy(&v, t + j, 16);
        const uint32_t c[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int q = 0; q < 16; q++) fn(j + q, (c[q >> 2] >> (8 * (q & 3))) & 0xFFu);
    }
    for (; j < T; j++) fn(j, (uint32_t)t[j]);
}
// the same backwards: fn(l, t[end - l]) for l = 0 .. n-1
template <class F>
TGSF_D void each_byte_bwd(const uint8_t* t, int end, int n, F fn)
{
    int l = 0;
    for (; l + 16 <= n; l += 16) {
        uint4 v;
        __builtin_memcpy(&v, t + end - l - 15, 16);
        const uint32_t c[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int q = 0; q < 16; q++) fn(l + q, (c[(15 - q) >> 2] >> (8 * ((15 - q) & 3))) & 0xFFu);
    }
    for (; l < n; l++) fn(l, (uint32_t)t[end - l]);
}

template <int NW>
TGSF_D int start_of(const DevParams& P, int a, const uint8_t* t, int end, int best)
{
    const int Q = P.Q[a];
    const uint64_t* pr = P.peq_rev + (size_t)a * 256 * kPeqW;
    Bv<NW> b;
    bv_init(b, Q);
    int maxl = end + 1;
    if (maxl > Q + best) maxl = Q + best;
    int best_l = 1;
    if constexpr (NW == 1) {
        each_byte_bwd(t, end, maxl, [&](int l, uint32_t sym) TGSF_INLINE_LAMBDA {
            bv_step<NW>(b, pr + (size_t)sym * kPeqW, 1, Q);
            if (b.score == best) best_l = l + 1;
        });
    } else {
        for (int l = 1; l <= maxl; l++) {
            bv_step<NW>(b, pr + (size_t)t[end - (l - 1)] * kPeqW, 1, Q);
            if (b.score == best) best_l = l;
        }
    }
    return end - best_l + 1;
}

// mlen of the first location, or -1 when the bounds prove it is below `need`
// (the caller only compares mlen with need), or a value >= need when they prove that.
template <int NW>
TGSF_D int first_mlen(const DevParams& P, int a, const uint8_t* t, int start0, int end0, int best, int need,
                      LaneScratch sc, bool exact)
{
    const int Q = P.Q[a];
    const int T = end0 - start0 + 1;
    if (!exact) {
        const int lo = T - best + (Q > T ? Q - T : 0);
        const int slack = best - (T - Q);
        const int hi = T - best + (slack > 0 ? slack / 2 : 0);
        if (lo >= need) return lo;          // passes whatever the path
        if (hi < need) return -1;           // fails whatever the path
    }
    return path_len_bv<NW>(P.peq_fwd + (size_t)a * 256 * kPeqW, Q, t + start0, T, sc) - best;
}

template <int NW>
TGSF_D WinAln align_window(const DevParams& P, int a, const uint8_t* t, int T, int kk, int need, LaneScratch sc, bool exact)
{
    const int Q = P.Q[a];
    const uint64_t* pf = P.peq_fwd + (size_t)a * 256 * kPeqW;
    WinAln r;
    r.best = -1; r.n = 0; r.first_end = r.last_end = -1; r.start0 = 0; r.mlen = 0;
    // peq tables are [256][kPeqW]; only the first NW words of each symbol are used
    Bv<NW> s;
    bv_init(s, Q);
    int cur = kk + 1;
    auto column = [&](int j, uint32_t sym) TGSF_INLINE_LAMBDA {
        bv_step<NW>(s, pf + (size_t)sym * kPeqW, 0, Q);
        if (s.score < cur) { cur = s.score; r.first_end = j; r.n = 0; }
        if (s.score == cur && cur <= kk) { r.last_end = j; r.n++; }
    };
    if constexpr (NW == 1) each_byte_fwd(t, T, column);
    else for (int j = 0; j < T; j++) column(j, (uint32_t)t[j]);
    if (cur > kk) return r;
    r.best = cur;
    r.start0 = start_of<NW>(P, a, t, r.first_end, cur);          // edlib.cpp:246-255
    r.mlen = first_mlen<NW>(P, a, t, r.start0, r.first_end, cur, need, sc, exact);
    return r;
}

// smallest start over all locations of a window whose optimum is `best`
template <int NW>
TGSF_D int min_start_all(const DevParams& P, int a, const uint8_t* t, int T, int best)
{
    const int Q = P.Q[a];
    const uint64_t* pf = P.peq_fwd + (size_t)a * 256 * kPeqW;
    Bv<NW> s;
    bv_init(s, Q);
    int mn = T;
    for (int j = 0; j < T; j++) {
        bv_step<NW>(s, pf + (size_t)t[j] * kPeqW, 0, Q);
        if (s.score == best) {
            int st = start_of<NW>(P, a, t, j, best);
            mn = st < mn ? st : mn;
        }
    }
    return mn;
}

// ---------------------------------------------------------------------------
// Adapters beyond 256 bp (only reachable with -a): the same searches with the column in kWideNW-word arrays walked by
// run-time loops (they live in scratch memory: correct for any length up to kMaxQ, not fast -- such adapters are a
// corner of the domain, the reference's multi-block edlib accepts them, include/edlib.cpp:182-185).
// ---------------------------------------------------------------------------
struct BvW {
    uint64_t p[kWideNW], m[kWideNW];
    int score;
};
TGSF_D void bvw_init(BvW& s, int Q, int nw) {
    for (int w = 0; w < nw; w++) { s.p[w] = ~0ull; s.m[w] = 0ull; }
    s.score = Q;
}
// one text column; hin_top 0 (infix) or +1 (prefix / global); ph_out (optional): the horizontal +1 words
TGSF_D void bvw_step(BvW& s, const uint64_t* eq, int hin_top, int Q, int nw, uint64_t* ph_out) {
    int hin = hin_top;
    for (int w = 0; w < nw; w++) {
        uint64_t Eq = eq[w], Pv = s.p[w], Mv = s.m[w];
        uint64_t Xv = Eq | Mv;
        if (hin < 0) Eq |= 1ull;
        uint64_t Xh = (((Eq & Pv) + Pv) ^ Pv) | Eq;
        uint64_t Ph = Mv | ~(Xh | Pv);
        uint64_t Mh = Pv & Xh;
        if (ph_out) ph_out[w] = Ph;
        const int bit = (w == nw - 1) ? ((Q - 1) & 63) : 63;
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
TGSF_D int start_of_w(const DevParams& P, int a, const uint8_t* t, int end, int best)
{
    const int Q = P.Q[a], nw = (Q + 63) >> 6;
    const uint64_t* pr = P.peq_rev_w + (size_t)a * 256 * kWideNW;
    BvW b;
    bvw_init(b, Q, nw);
    int maxl = end + 1;
    if (maxl > Q + best) maxl = Q + best;
    int best_l = 1;
    for (int l = 1; l <= maxl; l++) {
        bvw_step(b, pr + (size_t)t[end - (l - 1)] * kWideNW, 1, Q, nw, nullptr);
        if (b.score == best) best_l = l;
    }
    return end - best_l + 1;
}
// ---- the path of the first location beyond 1 MiB of traceback state: Hirschberg's divide and conquer as edlib does it ----
// 64 rows of a RANGE of the adapter (from row q0 + 64 w on) for one symbol; `row`: the symbol's kWideNW words of the wide
// table.  Rows below the range come along in the last word: they sit below every row that counts (carries run downwards).
TGSF_D uint64_t eq_range(const uint64_t* row, int q0, int w) {
    const int bit = q0 + 64 * w, wi = bit >> 6, sh = bit & 63;
    uint64_t v = wi < kWideNW ? row[wi] >> sh : 0ull;
    if (sh && wi + 1 < kWideNW) v |= row[wi + 1] << (64 - sh);
    return v;
}
// one text column of the GLOBAL alignment (hin = +1 at the top) of the Qn adapter rows from row q0 on
TGSF_D void bvw_step_range(BvW& s, const uint64_t* row, int q0, int Qn, int nw, uint64_t* ph_out) {
    int hin = 1;
    for (int w = 0; w < nw; w++) {
        uint64_t Eq = eq_range(row, q0, w), Pv = s.p[w], Mv = s.m[w];
        uint64_t Xv = Eq | Mv;
        if (hin < 0) Eq |= 1ull;
        uint64_t Xh = (((Eq & Pv) + Pv) ^ Pv) | Eq;
        uint64_t Ph = Mv | ~(Xh | Pv);
        uint64_t Mh = Pv & Xh;
        if (ph_out) ph_out[w] = Ph;
        const int bit = (w == nw - 1) ? ((Qn - 1) & 63) : 63;
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
// columns of the path edlib's traceback takes through the global alignment of adapter rows [q0, q0 + Qn) against
// t[0, Tn): from the bottom-right cell, up before left before diagonal (include/edlib.cpp:1023 / 1057 / 1088)
TGSF_D int path_columns_range(const uint64_t* pf, int q0, int Qn, const uint8_t* t, int Tn, LaneScratch sc)
{
    const int nw = (Qn + 63) >> 6;
    BvW s;
    bvw_init(s, Qn, nw);
    uint64_t ph[kWideNW];
    for (int j = 1; j <= Tn; j++) {
        bvw_step_range(s, pf + (size_t)t[j - 1] * kWideNW, q0, Qn, nw, ph);
        for (int w = 0; w < nw; w++) { sc.at(j, w, 2 * nw) = s.p[w]; sc.at(j, nw + w, 2 * nw) = ph[w]; }
    }
    int i = Qn, j = Tn, len = 0;
    while (i > 0 && j > 0) {
        const int r = i - 1;
        if ((sc.at(j, r >> 6, 2 * nw) >> (r & 63)) & 1ull) { i--; }
        else if ((sc.at(j, nw + (r >> 6), 2 * nw) >> (r & 63)) & 1ull) j--;
        else { i--; j--; }
        len++;
    }
    return len + i + j;
}
// alignmentLength of edlib's path for adapter a against t[0, T) with distance `best` (obtainAlignment,
// include/edlib.cpp:1164-1216): by traceback while (2*8+4)*blocks*columns + 8*columns < 1 MiB (:1191-1193), else the
// target is cut in the middle (:1250-1251), the query at the SMALLEST row h in 1..Qn-1 whose two half scores
// L[h] = NW(rows [0,h), left half) and R[h] = NW(rows [h,Qn), right half) add up to the score (:1321-1331), else at
// h = 0 (:1333-1340), else at h = Qn (:1341-1349); the quadrants recurse with their own scores (:1366-1384) and the
// lengths add up (:1392).  (edlib computes L and R inside a band; a row on an optimal path lies inside both bands and
// holds its exact value there, so the first h is the same: oracle/tgsf_oracle.c against the reference's own edlib.)
// The two half columns take 4 * nw words of the lane's scratch region, a leaf's traceback less than 1 MiB of it.
// -1: the scores are inconsistent (edlib: EDLIB_STATUS_ERROR) or the recursion outgrew its stack -- never seen.
TGSF_D int alignment_length_w(const DevParams& P, int a, const uint8_t* t, int T, int best, LaneScratch sc)
{
    const int Q = P.Q[a];
    const uint64_t* pf = P.peq_fwd_w + (size_t)a * 256 * kWideNW;
    const uint64_t* pr = P.peq_rev_w + (size_t)a * 256 * kWideNW;
    struct Node { int q0, qn, t0, tn, best; };
    Node st[40];
    int sp = 0, total = 0;
    st[sp++] = Node{0, Q, 0, T, best};
    while (sp > 0) {
        const Node nd = st[--sp];
        if (nd.qn == 0 || nd.tn == 0) { total += nd.qn + nd.tn; continue; }                       // :1171-1179
        const int nw = (nd.qn + 63) >> 6;
        const long long data = (2ll * 8 + 4) * nw * nd.tn + 8ll * nd.tn;
        if (data < (1ll << 20)) { total += path_columns_range(pf, nd.q0, nd.qn, t + nd.t0, nd.tn, sc); continue; }
        const int lw = nd.tn / 2, rw = nd.tn - lw;
        BvW s;
        bvw_init(s, nd.qn, nw);
        for (int j = 0; j < lw; j++) bvw_step_range(s, pf + (size_t)t[nd.t0 + j] * kWideNW, nd.q0, nd.qn, nw, nullptr);
        for (int w = 0; w < nw; w++) { sc.at(0, w, 1) = s.p[w]; sc.at(0, nw + w, 1) = s.m[w]; }     // the left half's last column
        // the right half: the reversed rows against the reversed text (rows [q0, q0+qn) reversed = rows
        // [Q - q0 - qn, Q - q0) of the reversed adapter)
        bvw_init(s, nd.qn, nw);
        const int q0r = Q - nd.q0 - nd.qn;
        for (int j = 0; j < rw; j++) bvw_step_range(s, pr + (size_t)t[nd.t0 + nd.tn - 1 - j] * kWideNW, q0r, nd.qn, nw, nullptr);
        auto dl = [&](int i) { return (int)((sc.at(0, i >> 6, 1) >> (i & 63)) & 1ull) - (int)((sc.at(0, nw + (i >> 6), 1) >> (i & 63)) & 1ull); };
        auto dr = [&](int i) { return (int)((s.p[i >> 6] >> (i & 63)) & 1ull) - (int)((s.m[i >> 6] >> (i & 63)) & 1ull); };
        int Rm = rw;                                    // R[1]: the last qn - 1 rows against the right half
        for (int i = 0; i + 1 < nd.qn; i++) Rm += dr(i);
        const int Rfull = Rm + dr(nd.qn - 1);           // R[0]
        int Lh = lw, h = -1, ls = 0, rs = 0;
        for (int x = 1; x <= nd.qn - 1; x++) {
            Lh += dl(x - 1);                            // L[x]
            if (Lh + Rm == nd.best) { h = x; ls = Lh; rs = Rm; break; }
            Rm -= dr(nd.qn - x - 1);                    // R[x + 1]
        }
        if (h < 0 && lw + Rfull == nd.best) { h = 0; ls = lw; rs = Rfull; }
        if (h < 0) {
            int Lq = lw;
            for (int i = 0; i < nd.qn; i++) Lq += dl(i);
            if (Lq + rw == nd.best) { h = nd.qn; ls = Lq; rs = rw; }
        }
        if (h < 0 || sp + 2 > 40) return -1;
        st[sp++] = Node{nd.q0 + h, nd.qn - h, nd.t0 + lw, rw, rs};
        st[sp++] = Node{nd.q0, h, nd.t0, lw, ls};
    }
    return total;
}

TGSF_D int first_mlen_w(const DevParams& P, int a, const uint8_t* t, int start0, int end0, int best, int need, LaneScratch sc, bool exact)
{
    const int Q = P.Q[a], nw = (Q + 63) >> 6;
    const int T = end0 - start0 + 1;
    if (!exact) {
        const int lo = T - best + (Q > T ? Q - T : 0);
        const int slack = best - (T - Q);
        const int hi = T - best + (slack > 0 ? slack / 2 : 0);
        if (lo >= need) return lo;
        if (hi < need) return -1;
    }
    if ((2ll * 8 + 4) * nw * T + 8ll * T >= (1ll << 20)) {                 // beyond 1 MiB of traceback state: as edlib, include/edlib.cpp:1191-1210
        const int len = alignment_length_w(P, a, t + start0, T, best, sc);
        return len < 0 ? -1 : len - best;
    }
    const uint64_t* pf = P.peq_fwd_w + (size_t)a * 256 * kWideNW;
    const uint8_t* tt = t + start0;
    BvW s;
    bvw_init(s, Q, nw);
    uint64_t ph[kWideNW];
    for (int j = 1; j <= T; j++) {
        bvw_step(s, pf + (size_t)tt[j - 1] * kWideNW, 1, Q, nw, ph);
        for (int w = 0; w < nw; w++) { sc.at(j, w, 2 * nw) = s.p[w]; sc.at(j, nw + w, 2 * nw) = ph[w]; }
    }
    int i = Q, j = T, len = 0;
    while (i > 0 && j > 0) {                              // up > left > diagonal (edlib.cpp:1023/1057/1088)
        const int r = i - 1;
        if ((sc.at(j, r >> 6, 2 * nw) >> (r & 63)) & 1ull) { i--; }
        else if ((sc.at(j, nw + (r >> 6), 2 * nw) >> (r & 63)) & 1ull) j--;
        else { i--; j--; }
        len++;
    }
    return len + i + j - best;
}
TGSF_D WinAln align_window_w(const DevParams& P, int a, const uint8_t* t, int T, int kk, int need, LaneScratch sc, bool exact)
{
    const int Q = P.Q[a], nw = (Q + 63) >> 6;
    const uint64_t* pf = P.peq_fwd_w + (size_t)a * 256 * kWideNW;
    WinAln r;
    r.best = -1; r.n = 0; r.first_end = r.last_end = -1; r.start0 = 0; r.mlen = 0;
    BvW s;
    bvw_init(s, Q, nw);
    int cur = kk + 1;
    for (int j = 0; j < T; j++) {
        bvw_step(s, pf + (size_t)t[j] * kWideNW, 0, Q, nw, nullptr);
        if (s.score < cur) { cur = s.score; r.first_end = j; r.n = 0; }
        if (s.score == cur && cur <= kk) { r.last_end = j; r.n++; }
    }
    if (cur > kk) return r;
    r.best = cur;
    r.start0 = start_of_w(P, a, t, r.first_end, cur);
    r.mlen = first_mlen_w(P, a, t, r.start0, r.first_end, cur, need, sc, exact);
    return r;
}
TGSF_D int min_start_all_w(const DevParams& P, int a, const uint8_t* t, int T, int best)
{
    const int Q = P.Q[a], nw = (Q + 63) >> 6;
    const uint64_t* pf = P.peq_fwd_w + (size_t)a * 256 * kWideNW;
    BvW s;
    bvw_init(s, Q, nw);
    int mn = T;
    for (int j = 0; j < T; j++) {
        bvw_step(s, pf + (size_t)t[j] * kWideNW, 0, Q, nw, nullptr);
        if (s.score == best) { const int st = start_of_w(P, a, t, j, best); mn = st < mn ? st : mn; }
    }
    return mn;
}

TGSF_D LaneScratch lane_scratch(const DevBatch& B, size_t first_wave)
{
    // one region per wave of the launch: [column][word][lane].  k_end_windows and k_mid_resolve may
    // run concurrently (different streams): they use disjoint wave ranges.
    const size_t wave_id = first_wave + ((size_t)gtid() >> 6);
    LaneScratch sc;
    sc.base = B.scratch + wave_id * B.scratch_wave_words + (threadIdx.x & 63u);
    return sc;
}

// Word count by adapter length = ceil(Q / 64): 1 (all library adapters), 2, and 3 or 4 (<= 256 bp; only kernels
// instantiated with MAXNW = 4 carry that path -- they are launched when an adapter of the run needs it, so the
// common configurations keep their register budget).
template <int MAXNW>
TGSF_D WinAln align_window_any(const DevParams& P, int a, const uint8_t* t, int T, int k, int need, const LaneScratch& sc, bool exact) {
    const int Q = P.Q[a];
    if (Q <= 64) return align_window<1>(P, a, t, T, k, need, sc, exact);
    if constexpr (MAXNW > 4) { if (Q > 256) return align_window_w(P, a, t, T, k, need, sc, exact); }
    if constexpr (MAXNW > 2) { if (Q > 192) return align_window<4>(P, a, t, T, k, need, sc, exact); if (Q > 128) return align_window<3>(P, a, t, T, k, need, sc, exact); }
    return align_window<2>(P, a, t, T, k, need, sc, exact);
}
template <int MAXNW>
TGSF_D int min_start_all_any(const DevParams& P, int a, const uint8_t* t, int T, int best) {
    const int Q = P.Q[a];
    if (Q <= 64) return min_start_all<1>(P, a, t, T, best);
    if constexpr (MAXNW > 4) { if (Q > 256) return min_start_all_w(P, a, t, T, best); }
    if constexpr (MAXNW > 2) { if (Q > 192) return min_start_all<4>(P, a, t, T, best); if (Q > 128) return min_start_all<3>(P, a, t, T, best); }
    return min_start_all<2>(P, a, t, T, best);
}
template <int MAXNW>
TGSF_D int start_of_any(const DevParams& P, int a, const uint8_t* t, int end, int best) {
    const int Q = P.Q[a];
    if (Q <= 64) return start_of<1>(P, a, t, end, best);
    if constexpr (MAXNW > 4) { if (Q > 256) return start_of_w(P, a, t, end, best); }
    if constexpr (MAXNW > 2) { if (Q > 192) return start_of<4>(P, a, t, end, best); if (Q > 128) return start_of<3>(P, a, t, end, best); }
    return start_of<2>(P, a, t, end, best);
}
template <int MAXNW>
TGSF_D int first_mlen_any(const DevParams& P, int a, const uint8_t* t, int s0, int e0, int best, int need, const LaneScratch& sc, bool exact) {
    const int Q = P.Q[a];
    if (Q <= 64) return first_mlen<1>(P, a, t, s0, e0, best, need, sc, exact);
    if constexpr (MAXNW > 4) { if (Q > 256) return first_mlen_w(P, a, t, s0, e0, best, need, sc, exact); }
    if constexpr (MAXNW > 2) { if (Q > 192) return first_mlen<4>(P, a, t, s0, e0, best, need, sc, exact); if (Q > 128) return first_mlen<3>(P, a, t, s0, e0, best, need, sc, exact); }
    return first_mlen<2>(P, a, t, s0, e0, best, need, sc, exact);
}

// ---------------------------------------------------------------------------
// k_end_windows: the 5' and 3' searches of GetEditDistance (src/TGSFilter.cpp:1266-1321).
// One lane per (read, adapter, end).  Every reported location pushes [0, end+1)
// (5') or [L-W5+start, L) (3'); only their union matters downstream, i.e. the
// last end / the smallest start.
// ---------------------------------------------------------------------------
template <int MAXNW>
TGSF_KERNEL k_end_windows(DevParams P, DevBatch B)
{
    // This kernel runs on the auxiliary stream beside the middle scan, whose waves are older and fill every SIMD: a SIMD
    // issues from its oldest ready wave first, so at equal priority these waves crawl until the scan's retire (1.9 ms for
    // 0.6 ms of work).  A few % of the scan's instructions: they go first.
    TGSF_WAVE_PRIO(2);
    const int A = P.n_adapters;
    const uint32_t idx = gtid();
    const uint32_t total = B.n * (uint32_t)A * 2u;
    if (idx >= total || !P.filter) return;
    const uint32_t r = idx / (2u * A);
    const int a = (int)((idx >> 1) % (uint32_t)A);
    const int e = (int)(idx & 1u);
    const uint32_t Lr = B.len[r];
    if (!Lr || (B.flags[r] & TGSF_RF_LOWQ)) return;
    const int L = (int)Lr;
    int W5 = P.w5[a];
    if (W5 > L) W5 = L;                                   // :1268-1270
    if (W5 < 5) return;                                   // :1274
    if (P.k_end[a] < 0) return;                           // match length > adapter: can never pass :1283
    const uint8_t* t = B.seq + B.off[r] + (e ? (L - W5) : 0);
    const LaneScratch sc = lane_scratch(B, 0);
    const int need = P.need_end[a];                        // mlen >= EndMatchLen && float(mlen)/Q >= EndSim (:1283-1288)
    WinAln w = align_window_any<MAXNW>(P, a, t, W5, P.k_end[a], need, sc, false);
    if (w.best < 0) return;
    if (w.mlen < need) return;
    if (e == 0) {
        B.clip5[(size_t)r * A + a] = w.last_end + 1;      // union of [0, end_i+1)
        atomicOr(&B.flags[r], (uint32_t)TGSF_RF_AD5P);
    } else {
        int mn = w.start0;
        if (w.n > 1) mn = min_start_all_any<MAXNW>(P, a, t, W5, w.best);
        B.clip3[(size_t)r * A + a] = L - W5 + mn;         // union of [L-W5+start_i, L)
        atomicOr(&B.flags[r], (uint32_t)TGSF_RF_AD3P);
    }
}

// ---------------------------------------------------------------------------
// k_mid_scan: the middle search of GetEditDistance (src/TGSFilter.cpp:1233-1264),
// i.e. edlib's infix scan (include/edlib.cpp:586-677) over read[E, L-E).
//
// One lane owns kSegCols columns of one read's middle window and scans them
// sequentially for up to AT adapters at once (independent dependency chains ->
// ILP).  Columns before the lane's own range are a warm-up: an alignment with
// distance <= k spans at most Q+k columns, so after Q+k warm-up columns every
// bottom-row value <= k is exact.  Reported locations are the columns at the
// global minimum; each lane appends the columns whose value ties or beats the
// best it has seen so far (a superset of the global-minimum columns it owns) to
// the read's candidate list; k_mid_resolve picks the minimum.  Candidates are
// rare (about 1e-6 per column on random sequence at default thresholds), so the
// value is only examined every 4 columns: it moves by at most 1 per column.
//
// Pure integer/bit work: ~30 VALU ops per column per adapter against one byte
// of sequence.  Peq rows live in LDS ([symbol][adapter]), sequence bytes are
// fetched 16 at a time per lane.
// ---------------------------------------------------------------------------
// Only the columns at the read's GLOBAL minimum matter.  A lane about to hand over the columns tying its own best value
// first compares that value with the best any lane of the read has handed over so far (one atomicMin): worse, and the
// columns are dropped here.  With thresholds close to Q every lane has a "best" at or below k; this keeps the pool to
// the lanes that tie or improve the read's running minimum.
TGSF_D bool worth_handing_over(const DevBatch& B, uint32_t r, int a, int A, int score)
{
    return score <= atomicMin(&B.mid_best[(size_t)r * A + a], score);
}
TGSF_D void push_candidate(const DevBatch& B, uint32_t r, int pos, int score, int a)
{
    if (*B.ovf) return;                                               // the scan will be redone anyway
    uint32_t idx = atomicAdd(B.pool_n, 1u);
    // Not errors: tgsf_wait re-runs the scan into a pool that fits, in position order (mid_mode).  A long list in the
    // order the lanes happened to reach it would cost the region kernel a pass over the list per out-of-order region.
    if (idx >= B.pool_cap) { *B.ovf = 1u; return; }
    if (atomicAdd(&B.mid_cnt[r], 1u) == (uint32_t)kMidListMax) *B.ovf = 1u;
    MidCand c;
    c.pos = pos;
    c.aux = score | (a << 16);
    c.state = 0;
    c.next = atomicExch(&B.mid_head[r], (int32_t)idx);
    B.pool[idx] = c;
}

// before the scans that follow a pool overflow: empty candidate lists (mid_best keeps the minima of the first scan)
TGSF_KERNEL k_mid_reset(DevBatch B, int A)
{
    for (uint32_t r = gtid(); r < B.n; r += gsize()) {
        B.mid_head[r] = -1;
        for (int a = 0; a < A; a++) B.mid_gate[(size_t)r * A + a] = 0u;
    }
}
// a candidate written at its own slot (mode 2): the slots of a read are consecutive, in ascending order of position
TGSF_D void place_candidate(const DevBatch& B, uint32_t idx, int pos, int score, int a)
{
    if (idx >= B.pool_cap) { *B.ovf = 1u; return; }                   // (cannot happen: the pool was sized from the counts)
    MidCand c;
    c.pos = pos;
    c.aux = score | (a << 16);
    c.state = 0;
    c.next = (int32_t)idx + 1;                                        // k_mid_link ends every read's list
    B.pool[idx] = c;
}
// first slot of read r's candidates (mode 2, after the prefix sum over seg_n)
TGSF_D uint32_t cand_begin(const DevBatch& B, uint32_t r, int A) { return B.seg_n[(size_t)B.seg_cnt[r] * (size_t)A]; }
TGSF_KERNEL k_mid_link(DevBatch B, int A)
{
    for (uint32_t r = gtid(); r < B.n; r += gsize()) {
        const uint32_t b = cand_begin(B, r, A), e = cand_begin(B, r + 1, A);
        B.mid_head[r] = b < e ? (int32_t)b : -1;
        if (b < e && e - 1u < B.pool_cap) B.pool[e - 1u].next = -1;
    }
}

// HT = Hot: adapters of 33..64 bp (one 64-bit word per column); HT = Hot32: adapters of at most 32 bp (one dword).
template <int AT, class HT = Hot>
TGSF_KERNEL TGSF_BOUNDS(kMidThreads, (AT <= 2 ? 4 : 2)) k_mid_scan1(DevParams P, DevBatch B, int a0, int na)   // (see k_mid_flat)
{
    typedef decltype(hot_eq(HT(), 0ull)) eq_t;
    TGSF_SHARED eq_t eqt[256][AT];
    // per lane and adapter: up to 4 columns that tie the lane's best value so far (slow path only)
    TGSF_SHARED int32_t tie_col[256][AT][4];
    for (uint32_t i = TGSF_COOP_BEGIN; i < 256u * AT; i += TGSF_COOP_STRIDE) {
        uint32_t sym = i / AT, j = i % AT;
        eqt[sym][j] = (int)j < na ? hot_eq(HT(), P.peq_top[(size_t)(a0 + j) * 256 + sym]) : (eq_t)0;
    }
    TGSF_BLOCK_SYNC();
    const uint32_t total = B.seg_cnt[B.n];
    // grid-stride over segments: the host may launch fewer workgroups than segments / 256 to leave part
    // of every CU to the HBM-bound kernels of the batch on the other stream (this kernel is VALU-bound)
    for (uint32_t g = gtid(); g < total; g += gsize()) {
    const uint32_t r = find_owner(B.seg_cnt, B.n, g);
    const uint32_t seg = g - B.seg_cnt[r];
    const int L = (int)B.len[r];
    const int E = P.end_len;
    const int ML = L - 2 * E;
    const uint8_t* mid = B.seq + B.off[r] + E;
    // this lane owns the seg-th seg_cols-aligned block (absolute addresses) that overlaps the window
    const uint64_t S = (uint64_t)P.seg_cols;
    const uint64_t amid = (uint64_t)(uintptr_t)mid;
    const uint64_t blk = amid / S + seg;
    const int c0 = blk * S > amid ? (int)(blk * S - amid) : 0;
    int c1 = (int)((blk + 1) * S - amid);
    if (c1 > ML) c1 = ML;

    HT st[AT];
    uint32_t slot[AT];        // mode 1: columns counted so far; mode 2: the next slot of this (lane, adapter)
    int lim[AT];              // best bottom-row value seen so far in the owned columns (k+1: none yet)
    int lim3[AT];             // the every-4th-column test: lim + 2 while nothing is recorded yet (only a value
                              // BELOW lim counts then), lim + 3 afterwards (ties count too)
    int ntie[AT];             // buffered columns attaining it
    int wu = 0;
    bool any_on = false;
#pragma unroll
    for (int j = 0; j < AT; j++) {
        const int a = a0 + j;
        bool on = j < na && ML >= P.Q[a] && P.k_mid[a] >= 0;          // :1237 tsmLen >= qLen
        int first_lim = on ? P.k_mid[a] + 1 : 0;
        if (on && B.mid_mode) {
            // second and third scan after a pool overflow: the (read, adapter)'s minimum is known from the first one
            // (every lane's best went through atomicMin whether or not its columns found room), so only the columns
            // AT that minimum are counted / handed over, and a lane whose pairs have nothing within k has nothing to do
            const int gmin = B.mid_best[(size_t)r * P.n_adapters + a];
            if (gmin > P.k_mid[a]) on = false; else first_lim = gmin + 1;
        }
        any_on |= on;
        slot[j] = (B.mid_mode == 2u && on) ? B.seg_n[(size_t)g * P.n_adapters + a] : 0u;
        hot_init(st[j], j < na ? P.Q[a] : 1);
        lim[j] = on ? first_lim : -1000;
        lim3[j] = lim[j] + 2;
        ntie[j] = 0;
        if (on) { int w = P.Q[a] + P.k_mid[a]; wu = w > wu ? w : wu; }
    }
    if (B.mid_mode && !any_on) continue;
    int c = c0 - wu;
    if (c < 0) c = 0;
    else if (c > 0) {
        // The 16-column body needs a 16-byte aligned start.  A few columns short of the boundary above are
        // cheaper to take one by one; otherwise the warm-up is lengthened down to the boundary below.
        const int mis = (int)((amid + (uint64_t)c) & 15u);
        if (mis && 16 - mis > 7) { const int al = c - mis; c = al > 0 ? al : c; }
    }

    auto step_all = [&](uint32_t byte) {
#pragma unroll
        for (int j = 0; j < AT; j++) hot_step(st[j], eqt[byte][j]);
    };
    // Only the columns at the GLOBAL minimum of the read matter (edlib.cpp:660-672), so a lane keeps
    // the columns tying ITS best value in a 4-slot buffer and pushes them to the read's candidate
    // list when the buffer fills or the block ends: the pool stays small for any threshold.
    int32_t (*ties)[4] = tie_col[threadIdx.x];
    auto flush_ties = [&](int j) {
        if (ntie[j] > 0) {
            if (B.mid_mode == 1u) slot[j] += (uint32_t)ntie[j];               // counting pass
            else if (B.mid_mode == 2u) for (int i = 0; i < ntie[j]; i++) place_candidate(B, slot[j]++, ties[j][i], lim[j], a0 + j);
            else if (worth_handing_over(B, r, a0 + j, P.n_adapters, lim[j]))
                for (int i = 0; i < ntie[j]; i++) push_candidate(B, r, ties[j][i], lim[j], a0 + j);
        }
        ntie[j] = 0;
    };
    auto note = [&](int j, int sc, int col) {
        if (sc < lim[j]) { lim[j] = sc; lim3[j] = sc + 3; ntie[j] = 0; }   // strictly better: older ties are void
        if (sc == lim[j]) {                                          // lim <= k here (init k+1 is never equalled... see below)
            if (ntie[j] == 4) flush_ties(j);
            ties[j][ntie[j]++] = col;
        }
    };
    auto check_col = [&](int col) {
#pragma unroll
        for (int j = 0; j < AT; j++) {
            const int sc = hot_score(st[j]);
            if (sc < lim[j] || (sc == lim[j] && ntie[j] > 0)) note(j, sc, col);
        }
    };
    // 16 columns starting at column cc0; `own` = the lane records candidates there (false during
    // the warm-up).  The bottom-row value moves by at most 1 per column, so it is evaluated after
    // every 4th column only; the 3 skipped columns are re-examined (from the states kept in
    // registers) when that value comes within 3 of the recording limit.
    auto chunk16 = [&](const uint4& v, int cc0, bool own) {
        const uint32_t d[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            HT h1[AT], h2[AT], h3[AT];
            step_all(d[k] & 0xFFu);
#pragma unroll
            for (int j = 0; j < AT; j++) h1[j] = st[j];
            step_all((d[k] >> 8) & 0xFFu);
#pragma unroll
            for (int j = 0; j < AT; j++) h2[j] = st[j];
            step_all((d[k] >> 16) & 0xFFu);
#pragma unroll
            for (int j = 0; j < AT; j++) h3[j] = st[j];
            step_all(d[k] >> 24);
            // value <= lim + 3  <=>  popcount(Pv) <= popcount(Mv) + (lim + 3): two accumulating v_bcnt per side
            bool any = false;
#pragma unroll
            for (int j = 0; j < AT; j++) any |= hot_within(st[j], lim3[j]);
            if (__builtin_expect(any && own, 0)) {          // a few % of the tests: keep this code out of the hot loop
#pragma unroll
                for (int j = 0; j < AT; j++) {
                    const int s4j = hot_score(st[j]);
                    const int cc = cc0 + 4 * k;
                    const int s1 = hot_score(h1[j]), s2 = hot_score(h2[j]), s3 = hot_score(h3[j]);
                    if (s1 < lim[j] || (s1 == lim[j] && ntie[j] > 0)) note(j, s1, cc);
                    if (s2 < lim[j] || (s2 == lim[j] && ntie[j] > 0)) note(j, s2, cc + 1);
                    if (s3 < lim[j] || (s3 == lim[j] && ntie[j] > 0)) note(j, s3, cc + 2);
                    if (s4j < lim[j] || (s4j == lim[j] && ntie[j] > 0)) note(j, s4j, cc + 3);
                }
            }
        }
    };

    // One instruction stream for warm-up and owned columns (a per-lane flag decides whether
    // candidates are recorded), so lanes with different warm-up lengths stay converged and the
    // 16-column body exists once in the code.  Block starts are seg_cols-aligned, so a 16-byte chunk
    // never straddles c0; only the first block of a read (c0 = 0, no warm-up) starts unaligned.
    while (c < c1 && ((amid + (uint64_t)c) & 15u)) { step_all(mid[c]); if (c >= c0) check_col(c); c++; }
    // 64 bytes per lane per fetch: a 128-byte line is touched by two fetch groups only
    // (16-byte fetches were re-fetching evicted lines: 2.1x the bytes, see profiles/).
    // The first group of a lane stops at the next 64-byte boundary, every later one covers a whole one.
    while (c + 16 <= c1) {
        const uint4* p4 = reinterpret_cast<const uint4*>(mid + c);
        const int room = 4 - (int)(((amid + (uint64_t)c) >> 4) & 3u);      // 16-B chunks up to the boundary
        uint4 v0 = p4[0], v1 = v0, v2 = v0, v3 = v0;
        int nq = 1;
        if (room >= 2 && c + 32 <= c1) { v1 = p4[1]; nq = 2; }
        if (room >= 3 && c + 48 <= c1) { v2 = p4[2]; nq = 3; }
        if (room >= 4 && c + 64 <= c1) { v3 = p4[3]; nq = 4; }
#pragma unroll 1
        for (int q = 0; q < nq; q++) {
            chunk16(v0, c, c >= c0);
            c += 16;
            v0 = v1; v1 = v2; v2 = v3;
        }
    }
    while (c < c1) { step_all(mid[c]); if (c >= c0) check_col(c); c++; }
#pragma unroll
    for (int j = 0; j < AT; j++) {
        flush_ties(j);
        if (B.mid_mode == 1u && j < na && slot[j]) B.seg_n[(size_t)g * P.n_adapters + a0 + j] = slot[j];
    }
    }
}

// ---------------------------------------------------------------------------
// k_mid_flat: the same search as k_mid_scan1 (first scan of a batch, mid_mode 0), with the work dealt differently.
//
// k_mid_scan1 gives every lane one 1 024-column block of the address space: each lane pays Q + k warm-up columns
// (7 %), the first and last block of a read are partial while their wave runs as long as its fullest lane, and
// unaligned heads and tails are stepped one column at a time in front of 63 waiting lanes.  Here the middle windows
// of a batch are ONE sequence of 16-column chunks (chk_cnt: chunks counted from each window's first column,
// prefix-summed over the reads), cut into stretches by flat_schedule (tgsf_dev.h): long ones (thousands of columns:
// one warm-up for all of them) for most of the sequence, shorter and shorter ones at its end, so that the launch ends
// everywhere within one short stretch.  Lane i of the grid takes stretch i -- no counter, no hand-out: the order in
// which the hardware starts the workgroups is the queue, as it is for k_mid_scan1 (a returning atomic on one word per
// stretch was measured: it paces the whole launch at ~20 stretches per microsecond) -- and the 64 lanes of a wave
// always hold 64 stretches of one length.  A stretch runs across read boundaries: the lane hands over what it holds for
// the read it leaves and starts the next window from its first column (no warm-up needed there).  Chunks are counted
// from the window's start, not from an aligned address, so the text is fetched with unaligned 16-byte loads, one chunk
// ahead of the columns, and no column is ever stepped alone; the last chunk of a window may hold fewer than 16
// columns: the bytes behind them (the read's 3' end window) go through the column as well -- the state is not used
// again -- and only columns inside the window are recorded.
//
// The candidates handed over and the minima in mid_best obey the same rules as k_mid_scan1's (a lane hands over
// the columns tying the best value it has seen in ITS stretch of a read, if that is no worse than the read's best so
// far), so everything behind the scan -- and the position-ordered scans after a pool overflow, which stay with
// k_mid_scan1 -- is unchanged.
// ---------------------------------------------------------------------------
// the first nvalid bytes at p (the rest zero): the last chunk of a window when fewer than 15 bytes follow it
// (sixteen predicated byte loads at constant shifts into four registers: an array indexed by the loop variable lived
// in scratch memory -- 112 bytes per lane in every instantiation of k_mid_flat)
TGSF_D uint4 load_upto16(const uint8_t* p, int nvalid)
{
    uint32_t d0 = 0u, d1 = 0u, d2 = 0u, d3 = 0u;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const uint32_t s = (i < nvalid ? (uint32_t)p[i] : 0u) << (8 * (i & 3));
        if ((i >> 2) == 0) d0 |= s;
        else if ((i >> 2) == 1) d1 |= s;
        else if ((i >> 2) == 2) d2 |= s;
        else d3 |= s;
    }
    uint4 v; v.x = d0; v.y = d1; v.z = d2; v.w = d3;
    return v;
}

//
// FS > 0 (with HT = Hot32, adapters of 33..64 bp searched within few differences: kSuffixMaxK): the scan as a FILTER.
// If the whole adapter ends at column j within k differences, so do its last 32 rows (the rows of an alignment below
// any row are an alignment of the adapter's tail, ending at the same column, with no more differences): only the last 32
// rows go through the column here -- the high dword of the top-aligned Eq row, the 10-instruction dword column -- and
// nothing is recorded but one bit per chunk and adapter (chk_mark) in which that value came within k; k_mid_recheck,
// launched right behind, puts the whole adapter through the chunks marked for it and hands over the candidates.  The value is looked at every
// FS-th column against k + FS - 1 (neighbouring bottom-row values differ by at most 1; marking more than needed costs a
// recheck, never a candidate).  On random sequence the 45-bp PacBio adapters at k = 11 mark 2 chunks in 1 000.
// Launched with 256 lanes a workgroup (kMidThreads); without a bound the compiler assumes 1 024 and caps the kernel at
// 128 registers: three and four adapters a pass (the ligation kits' distinct 5' and 3' adapters with their reverse
// complements, src/TGSFilter.cpp:3105-3113) then spilled up to 426 of them.  One and two adapters keep the cap they were
// tuned under (4 waves per SIMD); three and four take up to 256 registers at 2 waves per SIMD.
template <int AT, class HT = Hot, int FS = 0>
TGSF_KERNEL TGSF_BOUNDS(kMidThreads, (AT <= 2 ? 4 : 2)) k_mid_flat(DevParams P, DevBatch B, int a0, int na)
{
    static_assert(FS == 0 || (16 % FS) == 0, "the filter's test stride divides a chunk");
    typedef decltype(hot_eq(HT(), 0ull)) eq_t;
    TGSF_SHARED eq_t eqt[256][AT] __attribute__((aligned(16)));
    TGSF_SHARED int32_t tie_col[256][AT][4];
    for (uint32_t i = TGSF_COOP_BEGIN; i < 256u * AT; i += TGSF_COOP_STRIDE) {
        uint32_t sym = i / AT, j = i % AT;
        eqt[sym][j] = (int)j < na ? hot_eq(HT(), P.peq_top[(size_t)(a0 + j) * 256 + sym]) : (eq_t)0;
    }
    TGSF_BLOCK_SYNC();
    const int E = P.end_len;
    // values <= k are exact once the Q + k - 1 columns before them have gone through the column (an alignment within k
    // spans at most Q + k columns of text)
    int wu = 0;
    for (int j = 0; j < na; j++) { const int a = a0 + j; if (P.k_mid[a] >= 0) { const int w = P.Q[a] + P.k_mid[a] - 1; wu = w > wu ? w : wu; } }
    const uint32_t wuc = (uint32_t)(wu + 15) >> 4;
    int32_t (*ties)[4] = tie_col[threadIdx.x];
    const uint32_t n_chunks = B.chk_cnt[B.n];
    uint32_t ph_sh, ph_c0, ph_d0, ph_c1;
    const uint32_t n_stretch = flat_stretch(n_chunks, B.flat_pmax, B.flat_pmin, B.flat_f0, gtid(), ph_sh, ph_c0, ph_d0, ph_c1);

    // (the host launches one lane per stretch of the longest sequence the batch can have; the emulation's small grid strides)
    for (uint32_t d = gtid(); d < n_stretch; d += gsize()) {
        if (d != gtid()) (void)flat_stretch(n_chunks, B.flat_pmax, B.flat_pmin, B.flat_f0, d, ph_sh, ph_c0, ph_d0, ph_c1);
        const uint64_t first64 = (uint64_t)ph_c0 + ((uint64_t)(d - ph_d0) << ph_sh);
        if (first64 >= (uint64_t)ph_c1) continue;                        // (the last group of a phase may not be full)
        const uint32_t first = (uint32_t)first64;
        uint32_t left = ph_c1 - first;                                   // owned chunks still to do
        if (left > (1u << ph_sh)) left = 1u << ph_sh;
        uint32_t r = find_owner(B.chk_cnt, B.n, first);

        HT st[AT];
        int lim[AT];              // best bottom-row value seen so far in the owned columns of this read (k+1: none yet)
        int lim3[AT];             // the every-4th-column test (see k_mid_scan1)
        int ntie[AT];             // buffered columns attaining it
        const uint8_t* mid = nullptr;
        int ML = 0, c = 0, own_from = 0, cend = 0, cfull = 0;
        uint32_t cbase = 0u;      // (the filter) number of the window's first chunk in the batch's sequence of chunks
        auto open_read = [&](uint32_t ck) TGSF_INLINE_LAMBDA {
            if (FS) cbase = B.chk_cnt[r];
            ML = (int)B.len[r] - 2 * E;
            mid = B.seq + B.off[r] + E;
            cend = (ML + 15) & ~15;
            // chunks below cfull are fetched whole (16 bytes that lie inside the read): all of them unless -E is below 15
            cfull = E >= 15 ? cend : (ML + E) & ~15;
#pragma unroll
            for (int j = 0; j < AT; j++) {
                const int a = a0 + j;
                const bool on = j < na && ML >= P.Q[a] && P.k_mid[a] >= 0;            // :1237 tsmLen >= qLen
                hot_init(st[j], j < na ? P.Q[a] : 1);
                lim[j] = on ? P.k_mid[a] + (FS ? FS - 1 : 1) : -1000;             // (the filter: the bound of its test, fixed)
                lim3[j] = lim[j] + 2;
                ntie[j] = 0;
            }
            c = ck > wuc ? (int)((ck - wuc) << 4) : 0;
            own_from = (int)(ck << 4);
        };
        auto flush_ties = [&](int j) TGSF_INLINE_LAMBDA {
            if (ntie[j] > 0 && worth_handing_over(B, r, a0 + j, P.n_adapters, lim[j]))
                for (int i = 0; i < ntie[j]; i++) push_candidate(B, r, ties[j][i], lim[j], a0 + j);
            ntie[j] = 0;
        };
        auto note = [&](int j, int sc, int col) TGSF_INLINE_LAMBDA {
            if (sc < lim[j]) { lim[j] = sc; lim3[j] = sc + 3; ntie[j] = 0; }   // strictly better: older ties are void
            if (sc == lim[j]) {
                if (ntie[j] == 4) flush_ties(j);
                ties[j][ntie[j]++] = col;
            }
        };
        auto step_all = [&](uint32_t byte) TGSF_INLINE_LAMBDA {
#pragma unroll
            for (int j = 0; j < AT; j++) hot_step(st[j], eqt[byte][j]);
        };
        // 16 columns starting at column cc0 of the window; `own`: the lane records candidates there (not in the warm-up)
        auto chunk16 = [&](const uint4& v, int cc0, bool own) TGSF_INLINE_LAMBDA {
            const uint32_t dw[4] = {v.x, v.y, v.z, v.w};
            if (FS) {                                                 // the filter: one bit per adapter for the whole chunk
                bool hit[AT];
#pragma unroll
                for (int j = 0; j < AT; j++) hit[j] = false;
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    step_all((dw[k >> 2] >> (8 * (k & 3))) & 0xFFu);
                    if ((k % (FS ? FS : 1)) == (FS ? FS : 1) - 1) {
#pragma unroll
                        for (int j = 0; j < AT; j++) hit[j] |= hot_within(st[j], lim[j]);
                    }
                }
                bool any = false;
#pragma unroll
                for (int j = 0; j < AT; j++) any |= hit[j];
                if (__builtin_expect(any && own, 0)) {
                    const uint32_t g = cbase + ((uint32_t)cc0 >> 4);
#pragma unroll
                    for (int j = 0; j < AT; j++)
                        if (hit[j]) atomicOr(&B.chk_mark[(size_t)j * B.mark_stride + (g >> 5)], 1u << (g & 31u));
                }
                return;
            }
#pragma unroll
            for (int k = 0; k < 4; k++) {
                HT h1[AT], h2[AT], h3[AT];
                step_all(dw[k] & 0xFFu);
#pragma unroll
                for (int j = 0; j < AT; j++) h1[j] = st[j];
                step_all((dw[k] >> 8) & 0xFFu);
#pragma unroll
                for (int j = 0; j < AT; j++) h2[j] = st[j];
                step_all((dw[k] >> 16) & 0xFFu);
#pragma unroll
                for (int j = 0; j < AT; j++) h3[j] = st[j];
                step_all(dw[k] >> 24);
                bool any = false;
#pragma unroll
                for (int j = 0; j < AT; j++) any |= hot_within(st[j], lim3[j]);
                if (__builtin_expect(any && own, 0)) {          // a few % of the tests: keep this code out of the hot loop
                    const int cc = cc0 + 4 * k;
#pragma unroll
                    for (int j = 0; j < AT; j++) {
                        const int s4j = hot_score(st[j]);
                        const int s1 = hot_score(h1[j]), s2 = hot_score(h2[j]), s3 = hot_score(h3[j]);
                        // (columns at or beyond ML: bytes behind the window that went through the column with its last chunk)
                        if (cc < ML && (s1 < lim[j] || (s1 == lim[j] && ntie[j] > 0))) note(j, s1, cc);
                        if (cc + 1 < ML && (s2 < lim[j] || (s2 == lim[j] && ntie[j] > 0))) note(j, s2, cc + 1);
                        if (cc + 2 < ML && (s3 < lim[j] || (s3 == lim[j] && ntie[j] > 0))) note(j, s3, cc + 2);
                        if (cc + 3 < ML && (s4j < lim[j] || (s4j == lim[j] && ntie[j] > 0))) note(j, s4j, cc + 3);
                    }
                }
            }
        };

        open_read(first - B.chk_cnt[r]);
        // The loop below keeps every lane of the wave in one instruction stream without masks: the rare events of a lane
        // (its window has no further whole chunk; its stretch is done) are looked for with a wave-wide test before every
        // chunk and handled outside the hot loop, and a lane that is done "parks": it keeps stepping over the same 16
        // readable bytes, recording nothing, until its wave is done.
        // The text comes 64 bytes per lane at a time (a lane's 128-byte line is asked for twice, not eight times: the
        // lines of all resident lanes together outgrow the L2s, and every further touch would be another trip to the
        // fabric), one fetch group ahead of the columns: the four chunks of a trip of the hot loop sit in g0..g3; a chunk is
        // copied out as it is taken up, and while the last one goes through the columns the next four are on their way.
        // (Chunks past the window's last whole one: that one again, never used.)
        const uint8_t* const safe = reinterpret_cast<const uint8_t*>(P.peq_top);     // 16 readable bytes for parked lanes
        int dc = 16;
        bool parked = false;
        uint4 g0, g1, g2, g3;
        auto fetch4 = [&](int at) TGSF_INLINE_LAMBDA {                // the chunks at columns at, at + dc, at + 2 dc, at + 3 dc
            const int last = cfull - 16;
            const int p0 = at < last ? at : last, p1 = at + dc < last ? at + dc : last;
            const int p2 = at + 2 * dc < last ? at + 2 * dc : last, p3 = at + 3 * dc < last ? at + 3 * dc : last;
            g0 = load16u(mid + p0); g1 = load16u(mid + p1); g2 = load16u(mid + p2); g3 = load16u(mid + p3);
        };
        auto event = [&]() TGSF_INLINE_LAMBDA -> bool { return !parked && (left == 0u || c >= cfull); };
        auto chunk = [&](const uint4& v) TGSF_INLINE_LAMBDA {
            const bool own = c >= own_from;
            chunk16(v, c, own);
            c += dc;
            left -= own ? 1u : 0u;
        };
        for (;;) {
            // the hot loop: four chunks a trip, until some lane of the wave has an event
            if (!wave_any(event())) {
                if (c < cfull) fetch4(c); else { g0 = g1 = g2 = g3 = load16u(safe); }
                for (;;) {
                    { const uint4 cur = g0; chunk(cur); }
                    if (wave_any(event())) break;
                    { const uint4 cur = g1; chunk(cur); }
                    if (wave_any(event())) break;
                    { const uint4 cur = g2; chunk(cur); }
                    if (wave_any(event())) break;
                    { const uint4 cur = g3; fetch4(c + dc); chunk(cur); }
                    if (wave_any(event())) break;
                }
            }
            if (event()) {
                if (left == 0u) {                                         // the stretch is done
#pragma unroll
                    for (int j = 0; j < AT; j++) flush_ties(j);
                    parked = true; dc = 0; c = 0; own_from = 0x7FFFFFFF; cfull = 0x7FFFFFFF; cend = 0x7FFFFFFF; ML = 0;
                    mid = safe;
                } else if (c < cend) {
                    // the window's last columns with fewer than 15 bytes of the read behind them (-E below 15): byte by
                    // byte, never reading beyond the read
                    const bool own = c >= own_from;
                    chunk16(load_upto16(mid + c, ML + E - c), c, own);
                    c += 16;
                    left -= own ? 1u : 0u;
                } else {                                                  // the window is done: on to the next read that has one
#pragma unroll
                    for (int j = 0; j < AT; j++) flush_ties(j);
                    do { r++; } while (B.chk_cnt[r + 1] == B.chk_cnt[r]);
                    open_read(0u);
                }
            }
            if (!wave_any(!parked)) break;
        }
    }
}

// k_mid_marks + k_mid_recheck: behind a filtering pass of k_mid_flat (FS > 0, adapters a0 .. a0 + na - 1): every chunk
// marked for an adapter in chk_mark goes through that adapter's whole 64-bit column, from Q + k - 1 columns before it
// (values <= k are exact from there on, as in every scan here), and what its 16 columns hold is handed over by
// k_mid_scan1's rules: the columns tying the best value <= k seen in the chunk, if that is no worse than the read's best
// so far.  Marked chunks are few and scattered (a few in a thousand), and each costs some 70 columns: k_mid_marks
// turns the bitmaps into one list (a workgroup collects the marks of kRecheckWords words of each bitmap in LDS and
// appends them with one atomic), k_mid_recheck gives every lane of the chip one entry at a time.  Marks beyond the list's
// room (homopolymer reads against homopolymer adapters: every chunk) are rechecked on the spot by k_mid_marks.
constexpr int kRecheckWords = 4096;      // bitmap words (131 072 chunks, two million columns) per round of a workgroup
constexpr int kRecheckList = 4096;
constexpr int kRecheckPieces = 6;        // 16-byte pieces of text per marked chunk: warm-up (at most 64 + kSuffixMaxK - 1 columns) + the chunk
static_assert(64 + kSuffixMaxK - 1 + 16 <= 16 * kRecheckPieces, "k_mid_recheck: the warm-up and the chunk fit its pieces");
// item: chunk number * 4 + the adapter's place in the pass
TGSF_D void recheck_chunk(const DevParams& P, const DevBatch& B, const uint64_t (*eq)[4], int a0, uint32_t item)
{
    const uint32_t g = item >> 2;
    const int j = (int)(item & 3u), a = a0 + j;
    const uint32_t r = find_owner(B.chk_cnt, B.n, g);
    const int E = P.end_len, Q = P.Q[a], k = P.k_mid[a];
    const int ML = (int)B.len[r] - 2 * E;
    if (k < 0 || ML < Q) return;                                                  // :1237 (never marked)
    const uint8_t* mid = B.seq + B.off[r] + E;
    const int c1 = (int)((g - B.chk_cnt[r]) << 4);                                // the chunk's first column
    const int c2 = c1 + 16 < ML ? c1 + 16 : ML;
    Hot st;
    hot_init(st, Q);
    int lim = k + 1;
    uint32_t ties = 0u;                                                           // columns of the chunk at lim
    int c = c1 - (Q + k - 1);
    if (c < 0) c = 0;
    // the text in 16-byte pieces, all asked for at once (bytes behind the window belong to the read's 3' end window;
    // fewer than 16 of them left: byte by byte)
    auto piece = [&](int at) TGSF_INLINE_LAMBDA -> uint4 {
        if (at >= c2) { uint4 z; z.x = z.y = z.z = z.w = 0u; return z; }
        return at + 16 <= ML + E ? load16u(mid + at) : load_upto16(mid + at, ML + E - at);
    };
    uint4 p0 = piece(c), p1 = piece(c + 16), p2 = piece(c + 32), p3 = piece(c + 48), p4 = piece(c + 64), p5 = piece(c + 80);
    while (c < c2) {
        const uint32_t dw[4] = {p0.x, p0.y, p0.z, p0.w};
#pragma unroll
        for (int i = 0; i < 16; i++) {
            if (c + i < c2) {
                hot_step(st, eq[(dw[i >> 2] >> (8 * (i & 3))) & 0xFFu][j]);
                if (c + i >= c1) {
                    const int sc = hot_score(st);
                    if (sc < lim) { lim = sc; ties = 0u; }                    // (lim starts at k + 1: a column AT k + 1 is no candidate)
                    if (sc == lim && sc <= k) ties |= 1u << (c + i - c1);
                }
            }
        }
        c += 16;
        p0 = p1; p1 = p2; p2 = p3; p3 = p4; p4 = p5;
    }
    if (ties && worth_handing_over(B, r, a, P.n_adapters, lim))
        for (; ties; ties &= ties - 1u) push_candidate(B, r, c1 + __builtin_ctz(ties), lim, a);
}
TGSF_D void recheck_eq_rows(const DevParams& P, uint64_t (*eq)[4], int a0, int na)
{
    for (uint32_t i = TGSF_COOP_BEGIN; i < 256u * 4u; i += TGSF_COOP_STRIDE) {
        const uint32_t sym = i >> 2, j = i & 3u;
        eq[sym][j] = (int)j < na ? P.peq_top[(size_t)(a0 + j) * 256 + sym] : 0ull;
    }
}
TGSF_KERNEL k_mid_marks(DevParams P, DevBatch B, int a0, int na)
{
    TGSF_SHARED uint64_t eq[256][4];
    TGSF_SHARED uint32_t list[kRecheckList];
    TGSF_SHARED uint32_t list_n, list_at;
    recheck_eq_rows(P, eq, a0, na);
    const uint32_t total = B.chk_cnt[B.n], words = (total + 31u) >> 5;
    for (uint32_t w0 = blockIdx.x * (uint32_t)kRecheckWords; w0 < words; w0 += gridDim.x * (uint32_t)kRecheckWords) {
        if (TGSF_COOP_BEGIN == 0u) list_n = 0u;
        TGSF_BLOCK_SYNC();
        for (int j = 0; j < na; j++) {
            // (four words a load: mark_stride is a multiple of four, words past the batch's last are zero)
            const uint4* marks = reinterpret_cast<const uint4*>(B.chk_mark + (size_t)j * B.mark_stride + w0);
            for (uint32_t i = TGSF_COOP_BEGIN; i < (uint32_t)kRecheckWords / 4u && w0 + 4u * i < words; i += TGSF_COOP_STRIDE) {
                const uint4 m = marks[i];
                const uint32_t mw[4] = {m.x, m.y, m.z, m.w};
#pragma unroll
                for (uint32_t q = 0; q < 4u; q++)
                    for (uint32_t bits = mw[q]; bits; bits &= bits - 1u) {
                        const uint32_t item = ((((w0 + 4u * i + q) << 5) + (uint32_t)__builtin_ctz(bits)) << 2) | (uint32_t)j;
                        const uint32_t at = atomicAdd(&list_n, 1u);
                        if (at < (uint32_t)kRecheckList) list[at] = item;
                        else recheck_chunk(P, B, eq, a0, item);
                    }
            }
        }
        TGSF_BLOCK_SYNC();
        const uint32_t n = list_n < (uint32_t)kRecheckList ? list_n : (uint32_t)kRecheckList;
        if (TGSF_COOP_BEGIN == 0u) list_at = n ? atomicAdd(B.rc_n, n) : 0u;
        TGSF_BLOCK_SYNC();
        const uint32_t at = list_at;
        for (uint32_t i = TGSF_COOP_BEGIN; i < n; i += TGSF_COOP_STRIDE) {
            if (at + i < B.rc_cap) B.rc_list[at + i] = list[i];
            else recheck_chunk(P, B, eq, a0, list[i]);
        }
        TGSF_BLOCK_SYNC();
    }
}
TGSF_KERNEL k_mid_recheck(DevParams P, DevBatch B, int a0, int na)
{
    TGSF_SHARED uint64_t eq[256][4];
    recheck_eq_rows(P, eq, a0, na);
    TGSF_BLOCK_SYNC();
    const uint32_t n = *B.rc_n < B.rc_cap ? *B.rc_n : B.rc_cap;
    for (uint32_t i = gtid(); i < n; i += gsize()) recheck_chunk(P, B, eq, a0, B.rc_list[i]);
}

// adapters of 65..256 bp: NW-word standard layout (NW = ceil(Q / 64) = 2, 3 or 4), one adapter per pass
template <int NW>
TGSF_KERNEL k_mid_scanw(DevParams P, DevBatch B, int a)
{
    TGSF_SHARED uint64_t eqt[256][NW];
    TGSF_SHARED int32_t tie_col[256][4];
    for (uint32_t i = TGSF_COOP_BEGIN; i < 256u * NW; i += TGSF_COOP_STRIDE)
        eqt[i / NW][i % NW] = P.peq_fwd[(size_t)a * 256 * kPeqW + (size_t)(i / NW) * kPeqW + (i % NW)];
    TGSF_BLOCK_SYNC();
    const uint32_t total = B.seg_cnt[B.n];
    const uint32_t g = gtid();
    if (g >= total) return;
    const uint32_t r = find_owner(B.seg_cnt, B.n, g);
    const uint32_t seg = g - B.seg_cnt[r];
    const int L = (int)B.len[r];
    const int E = P.end_len;
    const int ML = L - 2 * E;
    const int Q = P.Q[a];
    if (ML < Q || P.k_mid[a] < 0) return;
    const uint8_t* mid = B.seq + B.off[r] + E;
    const uint64_t S = (uint64_t)P.seg_cols;
    const uint64_t amid = (uint64_t)(uintptr_t)mid;
    const uint64_t blk = amid / S + seg;
    const int c0 = blk * S > amid ? (int)(blk * S - amid) : 0;
    int c1 = (int)((blk + 1) * S - amid);
    if (c1 > ML) c1 = ML;
    Bv<NW> s;
    bv_init(s, Q);
    // As in k_mid_scan1: only the columns at the read's global minimum matter, so the lane keeps the columns tying ITS best
    // value (a 4-slot buffer, void as soon as a column does strictly better) and hands them over when the buffer fills or
    // the block ends -- with k close to Q every new low on the way down from Q would otherwise be a candidate.
    int32_t* ties = tie_col[threadIdx.x];
    int ntie = 0;
    int lim = P.k_mid[a] + 1;                                          // nothing at or below k yet
    int top = P.k_mid[a];                                              // values above it are never recorded
    if (B.mid_mode) {                                                  // after a pool overflow: see k_mid_scan1
        const int gmin = B.mid_best[(size_t)r * P.n_adapters + a];
        if (gmin > P.k_mid[a]) return;
        lim = gmin + 1;
        top = gmin;                                                    // (lim itself is NOT a value to record: only the minimum is)
    }
    uint32_t slot = B.mid_mode == 2u ? B.seg_n[(size_t)g * P.n_adapters + a] : 0u;
    auto hand_over = [&](int n) {
        if (B.mid_mode == 1u) slot += (uint32_t)n;
        else if (B.mid_mode == 2u) for (int i = 0; i < n; i++) place_candidate(B, slot++, ties[i], lim, a);
        else if (worth_handing_over(B, r, a, P.n_adapters, lim)) for (int i = 0; i < n; i++) push_candidate(B, r, ties[i], lim, a);
    };
    int c = c0 - (Q + P.k_mid[a]);
    if (c < 0) c = 0;
    for (; c < c0; c++) bv_step<NW>(s, eqt[mid[c]], 0, Q);
    for (; c < c1; c++) {
        bv_step<NW>(s, eqt[mid[c]], 0, Q);
        if (s.score < lim) { lim = s.score; ntie = 0; }
        if (s.score == lim && lim <= top) {
            if (ntie == 4) { hand_over(4); ntie = 0; }
            ties[ntie++] = c;
        }
    }
    if (ntie > 0) hand_over(ntie);
    if (B.mid_mode == 1u && slot) B.seg_n[(size_t)g * P.n_adapters + a] = slot;
}

// adapters beyond 256 bp: the wide column (see BvW), one adapter per pass, Eq rows read from the wide table in memory
TGSF_KERNEL k_mid_scan_wide(DevParams P, DevBatch B, int a)
{
    TGSF_SHARED int32_t tie_col[256][4];
    const uint32_t total = B.seg_cnt[B.n];
    const uint32_t g = gtid();
    if (g >= total) return;
    const uint32_t r = find_owner(B.seg_cnt, B.n, g);
    const uint32_t seg = g - B.seg_cnt[r];
    const int L = (int)B.len[r];
    const int E = P.end_len;
    const int ML = L - 2 * E;
    const int Q = P.Q[a], nw = (Q + 63) >> 6;
    if (ML < Q || P.k_mid[a] < 0) return;
    const uint64_t* pf = P.peq_fwd_w + (size_t)a * 256 * kWideNW;
    const uint8_t* mid = B.seq + B.off[r] + E;
    const uint64_t S = (uint64_t)P.seg_cols;
    const uint64_t amid = (uint64_t)(uintptr_t)mid;
    const uint64_t blk = amid / S + seg;
    const int c0 = blk * S > amid ? (int)(blk * S - amid) : 0;
    int c1 = (int)((blk + 1) * S - amid);
    if (c1 > ML) c1 = ML;
    BvW s;
    bvw_init(s, Q, nw);
    int32_t* ties = tie_col[threadIdx.x];
    int ntie = 0;
    int lim = P.k_mid[a] + 1;
    int top = P.k_mid[a];
    if (B.mid_mode) {                                                  // after a pool overflow: see k_mid_scan1, k_mid_scanw
        const int gmin = B.mid_best[(size_t)r * P.n_adapters + a];
        if (gmin > P.k_mid[a]) return;
        lim = gmin + 1;
        top = gmin;
    }
    uint32_t slot = B.mid_mode == 2u ? B.seg_n[(size_t)g * P.n_adapters + a] : 0u;
    auto hand_over = [&](int n) {
        if (B.mid_mode == 1u) slot += (uint32_t)n;
        else if (B.mid_mode == 2u) for (int i = 0; i < n; i++) place_candidate(B, slot++, ties[i], lim, a);
        else if (worth_handing_over(B, r, a, P.n_adapters, lim)) for (int i = 0; i < n; i++) push_candidate(B, r, ties[i], lim, a);
    };
    int c = c0 - (Q + P.k_mid[a]);
    if (c < 0) c = 0;
    for (; c < c1; c++) {
        bvw_step(s, pf + (size_t)mid[c] * kWideNW, 0, Q, nw, nullptr);
        if (c < c0) continue;                                           // warm-up columns
        if (s.score < lim) { lim = s.score; ntie = 0; }
        if (s.score == lim && lim <= top) {
            if (ntie == 4) { hand_over(4); ntie = 0; }
            ties[ntie++] = c;
        }
    }
    if (ntie > 0) hand_over(ntie);
    if (B.mid_mode == 1u && slot) B.seg_n[(size_t)g * P.n_adapters + a] = slot;
}

// ---------------------------------------------------------------------------
// k_mid_resolve: one lane per (read, adapter).  Among the read's candidates for
// this adapter the minimum value is edlib's editDistance and the columns that
// attain it are its endLocations (include/edlib.cpp:660-672).  The path of the
// FIRST location gives mlen, which gates ALL locations (src/TGSFilter.cpp:1245-1260).
// ---------------------------------------------------------------------------
template <int MAXNW>
TGSF_KERNEL k_mid_resolve(DevParams P, DevBatch B)
{
    const int A = P.n_adapters;
    const uint32_t idx = gtid();
    if (idx >= B.n * (uint32_t)A || !P.filter || pool_overflowed(B)) return;
    const uint32_t r = idx / (uint32_t)A;
    const int a = (int)(idx % (uint32_t)A);
    const int32_t head = B.mid_head[r];
    if (head < 0) return;
    int best = 1 << 30, e0 = 1 << 30;
    for (int32_t i = head; i >= 0; i = B.pool[i].next) {
        const int aux = B.pool[i].aux;
        if ((aux >> 16) != a) continue;
        const int sc = aux & 0xFFFF, pos = B.pool[i].pos;
        if (sc < best || (sc == best && pos < e0)) { best = sc; e0 = pos; }
        if (B.mid_mode) break;       // position-ordered arrays of columns AT the minimum: the first one is the first location
    }
    if (best == (1 << 30)) return;
    const int L = (int)B.len[r], E = P.end_len;
    const uint8_t* win = B.seq + B.off[r] + E;
    const LaneScratch sc = lane_scratch(B, B.scratch_mid_wave0);
    const int need = P.need_mid[a];                                      // :1246, :1250-1252
    const int s0 = start_of_any<MAXNW>(P, a, win, e0, best);
    const int mlen = first_mlen_any<MAXNW>(P, a, win, s0, e0, best, need, sc, false);
    if (mlen < need) return;
    atomicOr(&B.flags[r], (uint32_t)TGSF_RF_ADMID);
    if (B.mid_mode) { B.mid_gate[idx] = 1u; return; }                  // the locations themselves: k_mid_resolve_each, a lane each
    for (int32_t i = head; i >= 0; i = B.pool[i].next) {
        const int aux = B.pool[i].aux;
        if ((aux >> 16) != a || (aux & 0xFFFF) != best) continue;
        const int pos = B.pool[i].pos;
        const int st = (pos == e0) ? s0 : start_of_any<MAXNW>(P, a, win, pos, best);
        int ts = st + E - P.extra_len, te = pos + E + 1 + P.extra_len;   // :1248-1256
        if (ts < 0) ts = 0;
        if (te > L) te = L;
        // the slot is only ever touched by its own adapter's lane
        B.pool[i].pos = ts;
        B.pool[i].state = (te << 2) | 1;
    }
}

// The same for the position-ordered arrays (mid_mode): one lane per candidate slot -- a read may have millions -- finds its
// read by bisection over the reads' first slots and, if the (read, adapter)'s first location passed, its own start.
template <int MAXNW>
TGSF_KERNEL k_mid_resolve_each(DevParams P, DevBatch B)
{
    const int A = P.n_adapters;
    const uint32_t total = B.seg_n[(size_t)B.seg_cnt[B.n] * (size_t)A];
    for (uint32_t i = gtid(); i < total && i < B.pool_cap; i += gsize()) {
        uint32_t lo = 0, hi = B.n;                                     // largest r with cand_begin(r) <= i
        while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (cand_begin(B, mid, A) <= i) lo = mid; else hi = mid; }
        const uint32_t r = lo;
        const int aux = B.pool[i].aux, a = aux >> 16, best = aux & 0xFFFF, pos = B.pool[i].pos;
        if (!B.mid_gate[(size_t)r * A + a]) continue;
        const int L = (int)B.len[r], E = P.end_len;
        const uint8_t* win = B.seq + B.off[r] + E;
        const int st = start_of_any<MAXNW>(P, a, win, pos, best);
        int ts = st + E - P.extra_len, te = pos + E + 1 + P.extra_len;   // :1248-1256
        if (ts < 0) ts = 0;
        if (te > L) te = L;
        B.pool[i].pos = ts;
        B.pool[i].state = (te << 2) | 1;
    }
}

// ---------------------------------------------------------------------------
// k_regions: adapterMap (src/TGSFilter.cpp:1325-1434) for one read per lane.
// The reference sorts the drop regions by (start,end) and merges neighbours with
// prev.end >= next.start, i.e. it forms the union of the intervals with touching
// intervals joined; that union is built here by insertion into a small sorted
// list.  EMIT=false counts the keep regions, EMIT=true (after the scan of the
// counts) writes them and adds the DropInfo tallies.
// ---------------------------------------------------------------------------
struct RegList {
    int s[kMaxRegions], e[kMaxRegions];
    int n;
    bool overflow;
};
TGSF_D void reg_insert(RegList& R, int s, int e)
{
    // find the run of stored regions that overlap or touch [s,e)
    int i = 0;
    while (i < R.n && R.e[i] < s) i++;
    int j = i;
    while (j < R.n && R.s[j] <= e) {
        if (R.s[j] < s) s = R.s[j];
        if (R.e[j] > e) e = R.e[j];
        j++;
    }
    if (j == i) {                       // disjoint: open a slot at i
        if (R.n == kMaxRegions) { R.overflow = true; return; }
        for (int k = R.n; k > i; k--) { R.s[k] = R.s[k - 1]; R.e[k] = R.e[k - 1]; }
        R.n++;
    } else if (j - i > 1) {             // swallowed several: close the gap
        int d = j - i - 1;
        for (int k = i + 1; k + d < R.n; k++) { R.s[k] = R.s[k + d]; R.e[k] = R.e[k + d]; }
        R.n -= d;
    }
    R.s[i] = s; R.e[i] = e;
}

template <bool EMIT>
TGSF_KERNEL k_regions(DevParams P, DevBatch B)
{
    if (pool_overflowed(B)) return;
    const int A = P.n_adapters;
    uint64_t d[TGSF_N_DROPINFO];
#pragma unroll
    for (int k = 0; k < TGSF_N_DROPINFO; k++) d[k] = 0;
    for (uint32_t r0 = blockIdx.x * blockDim.x; r0 < B.n; r0 += gsize()) {
        const uint32_t r = r0 + threadIdx.x;
        if (r >= B.n) continue;
        const int L = (int)B.len[r];
        uint32_t nf = 0;
        const uint32_t fb = EMIT ? B.nfr[r] : 0u;
        auto keep = [&](int s, int l) {
            if (EMIT) {
                const uint32_t f = fb + nf;
                if (f < B.fcap) {
                    B.frag_off[f] = B.off[r] + (uint64_t)s;
                    B.frag_qoff[f] = B.qoff[r] + (uint64_t)s;
                    B.frag_len[f] = (uint32_t)l;
                    B.frag_sum[f] = 0;
                    B.frag_read[f] = r;
                    B.frag_start[f] = s;
                    B.frag_flags[f] = 0;
                } else set_status(B, DS_FRAG_CAP, r);
            }
            nf++;
        };
        const uint32_t fl = L ? B.flags[r] : (uint32_t)TGSF_RF_LOWQ;
        if (L == 0 || (fl & TGSF_RF_LOWQ)) {
            /* dropped before adapterMap (:1946-1953) */
        } else if (!P.filter) {
            if (!P.only_qc) keep(0, L);                                   // :1963-1965, :1976
        } else {
            // the raw drop regions of this read, in any order: fixed trims, end hits, resolved middle hits
            int n5 = 0, n3 = 0, nm = 0;
            auto each_raw = [&](auto&& f, bool count) {
                if (P.head_trim > 0) f(0, P.head_trim >= L ? L : P.head_trim);                  // :1334-1340
                if (P.tail_trim > 0) { if (P.tail_trim >= L) f(0, L); else f(L - P.tail_trim, L); }   // :1342-1348
                for (int a = 0; a < A; a++) {
                    const int c5 = B.clip5[(size_t)r * A + a], c3 = B.clip3[(size_t)r * A + a];
                    if (c5 > 0) { if (count) n5++; f(0, c5); }
                    if (c3 >= 0) { if (count) n3++; f(c3, L); }
                }
                for (int32_t i = B.mid_head[r]; i >= 0; i = B.pool[i].next) {
                    const int stt = B.pool[i].state;
                    if ((stt & 3) == 1) { if (count) nm++; f(B.pool[i].pos, stt >> 2); }
                }
            };
            RegList R;
            R.n = 0; R.overflow = false;
            each_raw([&](int s0, int e0) { reg_insert(R, s0, e0); }, true);
            if (EMIT) {                                                                          // :1354-1370
                const int cls = (nm > 0 && n5 > 0 && n3 > 0) ? 2 : (nm > 0 && n5 > 0) ? 3 : (nm > 0 && n3 > 0) ? 4
                        : (n5 > 0 && n3 > 0) ? 5 : (nm > 0) ? 6 : (n5 > 0) ? 7 : (n3 > 0) ? 8 : 9;
#pragma unroll
                for (int k = 2; k <= 9; k++) d[k] += (uint64_t)(cls == k);
            }
            if (nm > 0 && P.discard) {                                                           // :1372-1373
                if (EMIT) { d[10] += (uint64_t)L; B.trimmed[r] = (uint32_t)L; atomicOr(&B.flags[r], (uint32_t)TGSF_RF_DISCARDED); }
            } else if (R.n >= 1) {                                                               // :1396-1424
                int cur = 0;
                uint32_t tr = 0;
                auto on_region = [&](int rs, int re) {       // merged drop regions, ascending
                    const int dl = re - rs;
                    d[10] += (uint64_t)dl; tr += (uint32_t)dl;
                    if (dl == L) d[11]++;
                    if (rs > cur) {
                        const int kl = rs - cur;
                        if (kl >= P.min_len && kl <= P.max_len) keep(cur, kl);
                        else { d[11]++; d[12] += (uint64_t)kl; }
                    }
                    cur = re;
                };
                if (!R.overflow) {
                    for (int i = 0; i < R.n; i++) on_region(R.s[i], R.e[i]);
                } else {
                    // More disjoint regions than the list holds (a read studded with adapter copies): the same union,
                    // region by region, without storing it -- the next region starts at the smallest raw start beyond
                    // the previous one's end and grows while a raw region touches it (quadratic in the raw regions,
                    // which only such a read pays).
                    int last_e = -1;
                    for (;;) {
                        int ms = 0x7FFFFFFF, me = 0;
                        each_raw([&](int s0, int e0) { if (s0 > last_e && (s0 < ms || (s0 == ms && e0 > me))) { ms = s0; me = e0; } }, false);
                        if (ms == 0x7FFFFFFF) break;
                        for (bool grew = true; grew;) {
                            grew = false;
                            each_raw([&](int s0, int e0) { if (s0 >= ms && s0 <= me && e0 > me) { me = e0; grew = true; } }, false);
                        }
                        on_region(ms, me);
                        last_e = me;
                    }
                }
                if (cur < L) {
                    const int kl = L - cur;
                    if (kl >= P.min_len && kl <= P.max_len) keep(cur, kl);
                    else { d[11]++; d[12] += (uint64_t)kl; }
                }
                if (EMIT) B.trimmed[r] = tr;
            } else {                                                                             // :1425-1432
                if (L >= P.min_len && L <= P.max_len) keep(0, L);
                else { d[11]++; d[12] += (uint64_t)L; }
            }
        }
        if (P.only_qc) nf = 0;
        if (!EMIT) B.nfr[r] = nf;
    }
    if (EMIT) {
        for (int k = 2; k <= 12; k++) wave_add_u64(&B.ctr[TGSF_CTR_DROPINFO + k], d[k]);
    }
}

// ---------------------------------------------------------------------------
// k_repeat: the repeat gate, GetKmerCount (src/TGSFilter.cpp:1703-1753, :1982-1989), for k <= 13 (launched for k <= 11:
// from 12 on k_repeat_keys is faster).
// repeat = (#k-mers) - (#distinct k-mers) of a fragment; fragments below -p are dropped before any
// clean statistics.  k-mers are 2-bit codes (A0 C1 G2 T3, every other byte 0), first base in the top bits.
//
// One 1024-lane workgroup per fragment (one per CU: the 4^k-bit "seen" set is a 2^20-bit LDS bitmap, 128 KB;
// k = 11, 12, 13 take 4, 16, 64 passes, pass p owning the k-mers whose first k-10 bases spell p).
//   * The text is read as aligned 16-byte chunks, one per lane, and turned into one word of sixteen 2-bit codes
//     (four bytes at a time), first base of the chunk in the top bits; positions are counted from the chunk the
//     fragment starts in, so the load needs no shifting and a k-mer is a bit-field of two consecutive words.
//   * The next fragment's first window is fetched into registers before this one's passes and converted after
//     them: its HBM latency hides behind the passes.
//   * A lane owns the 16 k-mer starts of a word.  With one pass it marks all sixteen; with several it forms the
//     mask of the starts whose leading bases match the pass (three logic ops per base of the prefix, for the
//     whole word) and walks its set bits -- no pass touches a k-mer it does not own.
//   * Marks are no-return LDS ORs (measured on gfx950: random ds_or costs what a random ds_write_b32 does, 9-12
//     cycles per wave instruction on the CU); "distinct" is the popcount of the bitmap, taken while it is
//     cleared for the next pass.
//   * Fragments are dealt to the workgroups by a counter (B.rep_next, zeroed by the host before the launch).
// ---------------------------------------------------------------------------
constexpr int kRepThreads = 1024;
constexpr int kRepSlots = 6;                          // chunks a lane holds in registers for the prefetch
constexpr int kRepWords = kRepSlots * kRepThreads;    // words (16 bases each) of a window: 96 K bases, 24 KB
// 2-bit code of a base as GetKmerCount assigns it: exactly 'A' 'C' 'G' 'T' -> 0 1 2 3, any other byte 0
// (:1709-1724).  Branch-free: (c>>1)&3 is A0 C1 T2 G3, x^(x>>1) swaps the last two; validity from a bit
// mask over c - 'A'.
TGSF_D uint32_t base_code(uint32_t c) {
    const uint32_t v = c - 0x41u;
    const uint32_t valid = (v < 20u ? 1u : 0u) & (0x80045u >> (v & 31u));
    const uint32_t x = (c >> 1) & 3u;
    return (x ^ (x >> 1)) & (0u - valid);
}
// the same for four text bytes at once: 8 bits, the first byte's code in bits 7..6
TGSF_D uint32_t base_codes4(uint32_t d) {
    const uint32_t x = (d >> 1) & 0x03030303u;
    uint32_t y = x ^ ((x >> 1) & 0x01010101u);                                // A0 C1 G2 T3; any other byte: something in 0..3
    const uint32_t diff = d ^ perm_bytes(0u, 0x54474341u, y);                 // a zero byte where the text byte is exactly A C G T
    const uint32_t nz = ((((diff & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | diff) >> 7) & 0x01010101u;
    y &= ~(nz | (nz << 1));
    return ((y << 6) | (y >> 4) | (y >> 14) | (y >> 24)) & 0xFFu;
}
TGSF_D uint32_t base_codes16(const uint4& r) {
    return (base_codes4(r.x) << 24) | (base_codes4(r.y) << 16) | (base_codes4(r.z) << 8) | base_codes4(r.w);
}
// bit 2s set where the 2-bit group s of w equals c
TGSF_D uint32_t rep_eq_mask(uint32_t w, uint32_t c) {
    const uint32_t x = w ^ (c * 0x55555555u);
    return ~(x | (x >> 1)) & 0x55555555u;
}
TGSF_KERNEL TGSF_BOUNDS(kRepThreads, 4) k_repeat(DevParams P, DevBatch B)
{
    constexpr int W = kRepWords;
    TGSF_SHARED uint4 bm4[8192];                      // 128 KB: one partition of the 4^k-bit "seen" set
    TGSF_SHARED uint32_t codes[W + 4];                // a window of the fragment as 2-bit codes
    TGSF_SHARED uint32_t next_s;
    TGSF_SHARED int rep_s;                           // repeats found in the passes done so far: their k-mers less their distinct ones
    uint32_t* bm = reinterpret_cast<uint32_t*>(bm4);
    const int k = P.kmer;
    const uint32_t space_log2 = 2u * (uint32_t)k;                      // k <= 13 -> <= 26 bits
    const uint32_t part_log2 = space_log2 < 20u ? space_log2 : 20u;
    const int PB = (int)(space_log2 - part_log2) / 2;                  // leading bases that select the pass
    const uint32_t passes = 1u << (2 * PB);
    const uint32_t part_q = (((1u << part_log2) + 31u) / 32u + 3u) / 4u;   // uint4s of a partition
    const uint32_t nf = stored_frags(B);
    uint64_t drop_n = 0, drop_b = 0;
    if (pool_overflowed(B)) return;
    const int NT = kTgsfEmul ? 1 : kRepThreads, tid = kTgsfEmul ? 0 : (int)threadIdx.x;   // (emulation: one lane does the whole fragment)
    if (kTgsfEmul && threadIdx.x != 0) return;
    uint4 zero4;
    zero4.x = zero4.y = zero4.z = zero4.w = 0;
    for (uint32_t w = (uint32_t)tid; w < 8192u; w += (uint32_t)NT) bm4[w] = zero4;

    // the fragment as aligned 16-byte chunks: chunk 0 holds its first base at byte a
    auto chunks_of = [&](uint32_t f, const uint4*& base, int& a, int& L) TGSF_INLINE_LAMBDA {
        const uint8_t* s = B.seq + B.frag_off[f];
        a = (int)((uintptr_t)s & 15u);
        base = reinterpret_cast<const uint4*>(s - a);
        L = (int)B.frag_len[f];
    };
    // codes[0 .. ) <- chunks [wb, wb + W) of the fragment (as many as it has)
    auto load_window = [&](const uint4* base, int words, int wb) TGSF_INLINE_LAMBDA {
        for (int g = tid; g < W && wb + g < words; g += NT) codes[g] = base_codes16(base[wb + g]);
    };
    // the next fragment's text on its way into registers while this one is counted (the serial emulation reads windows)
    uint4 raw[kRepSlots];
    auto prefetch = [&](uint32_t f) TGSF_INLINE_LAMBDA {
        if (kTgsfEmul || f >= nf) return;
        const uint4* base; int a, L;
        chunks_of(f, base, a, L);
        const int words = (a + L + 15) / 16;
#pragma unroll
        for (int sl = 0; sl < kRepSlots; sl++) { const int g = tid + sl * NT; if (g < words) raw[sl] = base[g]; }
    };
    prefetch(blockIdx.x);
    uint32_t f = blockIdx.x;
    while (f < nf) {
        const uint4* base; int a, L;
        chunks_of(f, base, a, L);
        const int total = L - k + 1;                                   // number of k-mers
        const int words = (a + L + 15) / 16;                           // chunks holding the fragment
        const int kwords = total > 0 ? (a + total + 15) / 16 : 0;      // chunks in which a k-mer starts
        const bool one_window = words <= W;
        if (tid == 0) { rep_s = 0; next_s = gridDim.x + atomicAdd(B.rep_next, 1u); }
        TGSF_BLOCK_SYNC();                                             // the previous fragment's readers of codes[] are done
        if (kTgsfEmul) load_window(base, words, 0);
        else {
#pragma unroll
            for (int sl = 0; sl < kRepSlots; sl++) { const int g = tid + sl * NT; if (g < words) codes[g] = base_codes16(raw[sl]); }
        }
        TGSF_BLOCK_SYNC();
        const uint32_t fnext = next_s;
        prefetch(fnext);
        for (uint32_t pass = 0; pass < passes && kwords > 0; pass++) {
            uint32_t own = 0;                                          // k-mers this lane marks in this pass
            // windows of W chunks; a k-mer starting in chunk g reads chunk g+1 too, so consecutive windows share one chunk
            for (int wb = 0;; wb += W - 1) {
                if (!one_window && !(pass == 0 && wb == 0)) {
                    TGSF_BLOCK_SYNC();
                    load_window(base, words, wb);
                    TGSF_BLOCK_SYNC();
                }
                const bool last = wb + W >= words;
                const int gend = last ? kwords - wb : W - 1;           // k-mers start in chunks [wb, wb + gend)
                for (int g = tid; g < gend; g += NT) {
                    const uint32_t hi = codes[g], lo = codes[g + 1];
                    const int first = a - 16 * (wb + g);               // starts before the fragment's first base (chunk 0 only)
                    const int v = a + total - 16 * (wb + g);           // starts up to the last k-mer
                    if (PB == 0 && first <= 0 && v >= 16) {
                        own += 16u;
#pragma unroll
                        for (int j = 0; j < 16; j++) {
                            const uint32_t t = j ? alignbit(hi, lo, 32u - 2u * (uint32_t)j) : hi;   // the k-mer from bit 31 down
                            const uint32_t idx = t >> (32u - part_log2);
                            atomicOr(&bm[idx >> 5], 1u << (idx & 31u));     // result unused: a fire-and-forget ds_or
                        }
                    } else {
                        uint32_t m = 0x55555555u;                      // bit 30-2j: a k-mer of this pass starts at base j of the chunk
                        if (PB >= 1) m = rep_eq_mask(hi, (pass >> (2 * (PB - 1))) & 3u);
                        if (PB >= 2) m &= alignbit(rep_eq_mask(hi, (pass >> (2 * (PB - 2))) & 3u), rep_eq_mask(lo, (pass >> (2 * (PB - 2))) & 3u), 30u);
                        if (PB >= 3) m &= alignbit(rep_eq_mask(hi, pass & 3u), rep_eq_mask(lo, pass & 3u), 28u);
                        if (first > 0) m &= 0xFFFFFFFFu >> (2 * first);
                        if (v < 16) m &= ~(0xFFFFFFFFu >> (2 * v));
                        own += popc32(m);
                        const uint64_t win = ((uint64_t)hi << 32) | lo;
                        const int c0 = 30 + 2 * PB;
                        while (m) {
                            const int b = __builtin_ctz(m);
                            m &= m - 1u;
                            const uint32_t t = (uint32_t)((win << (c0 - b)) >> 32);   // the k-mer less its leading PB bases, from bit 31 down
                            const uint32_t idx = t >> (32u - part_log2);
                            atomicOr(&bm[idx >> 5], 1u << (idx & 31u));
                        }
                    }
                }
                if (last) break;
            }
            // distinct k-mers of this partition = set bits of the bitmap; leave it clear
            TGSF_BLOCK_SYNC();
            uint32_t mine = 0;
            for (uint32_t w = (uint32_t)tid; w < part_q; w += (uint32_t)NT) {
                const uint4 q = bm4[w];
                mine += popc32(q.x) + popc32(q.y) + popc32(q.z) + popc32(q.w);
                bm4[w] = zero4;
            }
            // the repeats of this pass: its k-mers less its distinct ones (never negative over the workgroup)
            const int part = wave_sum_i32((int)own - (int)mine);
            if (wave_leader() && part) atomicAdd(&rep_s, part);
            TGSF_BLOCK_SYNC();
            // What the gate asks is whether repeat reaches -p (:1984), and every pass only adds to it: a fragment is
            // accepted as soon as the passes so far hold -p repeats (a 150-kb read: after the first of its four passes).
            if (rep_s >= P.min_repeat) break;
        }
        if (tid == 0) {
            const int repeat = rep_s;                                  // all of them, or enough of them
            if (repeat < P.min_repeat) {                               // :1984-1988
                B.frag_flags[f] |= TGSF_FF_REPEAT;
                drop_n++; drop_b += (uint64_t)L;
            }
        }
        f = fnext;
    }
    if (tid == 0 && drop_n) {
        atomicAdd((ull*)&B.ctr[TGSF_CTR_DROPINFO + 15], (ull)drop_n);
        atomicAdd((ull*)&B.ctr[TGSF_CTR_DROPINFO + 16], (ull)drop_b);
    }
}

// ---------------------------------------------------------------------------
// k_repeat_keys: the same gate where the 4^k-bit set is too large to sweep as LDS bitmaps (k = 12..31; 32-bit keys up to
// k = 15, 64-bit above).  What the gate needs is exact only around -p (:1984: repeat < MinRepeat drops the fragment), and
// repeat = sum over the keys of (occurrences - 1) = the occurrences that are not the first of their key.  Per pass, in LDS:
//   0. every k-mer ORs a mask of three bits, chosen by its hash, into ONE word of a 96-KB map A (a Bloom filter whose block
//      is the word: one returning ds_or per k-mer, so the occurrences of a key are ordered by that one atomic and every
//      one but the first finds its three bits set).  An occurrence that finds them set is "flagged": every repeat is, and
//      so are a few first occurrences (1e-3 of them at 131 072 k-mers).  Flagged occurrences mark their hash value in a
//      32-KB map B.  flagged = 0 -> the pass holds no repeat;  flagged < -p on a fragment's last pass -> the fragment is
//      dropped, whatever the exact number is: no second scan (random fragments up to ~50 000 k-mers end here).
//   1. otherwise the fragment is scanned again: the occurrences whose hash value is in B -- all occurrences of a key or
//      none, and every repeated key's -- are counted (M) and their keys, compared in full, go into an open-addressing table
//      that takes A's place (I insertions).  The pass's repeat = M - I, exactly, whatever the hashes do.
//   A fragment of more than kRepShare k-mers takes 2, 4, ... passes, a pass owning the keys by a hash of the whole key (the
//   others OR a zero: every pass runs the same straight-line code over whole chunks, sixteen keys at constant shifts with
//   their LDS operations back to back); the passes double, the fragment starting over, when a pass's table fills up
//   (long fragments of many distinct repeats).  A fragment is accepted as soon as the passes so far hold -p repeats.
//   Windows, chunk layout, prefetch and work counter as in k_repeat.
//   k = 32 follows the reference's machine there (:1748, see the oracle): the first k-mer as built, every later one 0.
// ---------------------------------------------------------------------------
constexpr uint32_t kRepShare = 114688;                // k-mers of a pass for which the flagged few stay few (~2 300 of them, ~3 300 keys for the table)
// fragments of more than one pass (TGSF_KERNEL k_repeat_long lists them; k_repeat_keys hands them out first: a 1-Mb fragment
// is 16 passes on ONE workgroup, milliseconds that must not begin when the others are about to finish)
TGSF_D bool rep_long(const DevParams& P, const DevBatch& B, uint32_t f) { return (int)B.frag_len[f] - P.kmer + 1 > (int)kRepShare; }
TGSF_KERNEL k_repeat_long(DevParams P, DevBatch B)
{
    if (pool_overflowed(B)) return;
    const uint32_t nf = stored_frags(B);
    for (uint32_t f = blockIdx.x * blockDim.x + threadIdx.x; f < nf; f += gsize()) {
        if (!rep_long(P, B, f)) continue;
        const uint32_t i = atomicAdd(&B.rep_next[1], 1u);
        if (i < B.rep_long_cap) B.rep_next[2 + i] = f;                 // (always: the long fragments are disjoint pieces of the batch's bases)
        else set_status(B, DS_FRAG_CAP, B.frag_read[f]);
    }
}
constexpr uint32_t kRepMaxPlog = 10;                  // passes (log2) beyond which a fragment's k-mers are counted in memory instead
// The number of distinct k-mers among the `total` k-mers of the text at s (GetKmerCount's set, src/TGSFilter.cpp:1703-1753;
// k <= 31: 2-bit codes, non-ACGT and lower case = 0 as there, :1709-1724), by the whole workgroup: an open-addressing set
// of the full keys in B.rep_tab.  Every access to the table is an agent-scope atomic or a write-through store (the
// workgroups that use it one after the other may sit on different XCDs, whose L2s do not see each other's lines).
// acc: a word of LDS.  Called by all lanes of the workgroup together.
TGSF_D uint32_t rep_distinct_in_memory(const DevBatch& B, const uint8_t* s, int total, int k, int tid, int NT, uint32_t* acc)
{
    const ull kEmptyKey = ~0ull;                                       // no key has all its bits set (k <= 31)
    uint32_t lg = 4;
    while (lg < B.rep_tab_log2 && (1ull << lg) < 2ull * (ull)total) lg++;
    const ull mask = (1ull << lg) - 1ull;
    if (tid == 0) { while (atomicCAS(B.rep_lock, 0u, 1u) != 0u) { } *acc = 0u; }
    TGSF_BLOCK_SYNC();
    for (ull i = (ull)tid; i <= mask; i += (ull)NT) atomicExch(&B.rep_tab[i], kEmptyKey);
    TGSF_BLOCK_SYNC();
    uint32_t mine = 0;
    for (int i = tid; i < total; i += NT) {
        ull key = 0;
        for (int j = 0; j < k; j++) key = (key << 2) | base_code(s[i + j]);
        ull h = (key * 0x9E3779B97F4A7C15ull) >> (64u - lg);
        for (;;) {
            const ull old = atomicCAS(&B.rep_tab[h], kEmptyKey, key);
            if (old == kEmptyKey) { mine++; break; }
            if (old == key) break;
            h = (h + 1ull) & mask;
        }
    }
    mine = (uint32_t)wave_sum((uint64_t)mine);
    if (wave_leader() && mine) atomicAdd(acc, mine);
    TGSF_BLOCK_SYNC();
    const uint32_t distinct = *acc;
    TGSF_BLOCK_SYNC();
    if (tid == 0) { *acc = 0u; atomicExch(B.rep_lock, 0u); }
    TGSF_BLOCK_SYNC();
    return distinct;
}
constexpr uint32_t kRepAQ = 6144;                     // map A: 96 KB = 24 576 words
constexpr uint32_t kRepBQ = 2048;                     // map B: 32 KB = 2^18 bits
constexpr uint32_t kRepTabQ = 4096;                   // the table: the first 64 KB of A
template <bool KEY64>
TGSF_KERNEL TGSF_BOUNDS(kRepThreads, 4) k_repeat_keys(DevParams P, DevBatch B)
{
    constexpr int W = kRepWords;
    constexpr int OV = KEY64 ? 2 : 1;                                  // chunks shared by consecutive windows (a key spans up to 3 / 2)
    constexpr uint32_t TLOG = KEY64 ? 13u : 14u;                       // slots of the table (64 KB), log2
    typedef typename std::conditional<KEY64, ull, uint32_t>::type key_t;
    TGSF_SHARED uint4 A4[kRepAQ];                     // map A, then the table of full keys
    TGSF_SHARED uint4 B4[kRepBQ];                     // map B
    TGSF_SHARED uint32_t codes[W + 8];
    TGSF_SHARED uint32_t acc_s[3], next_s, over_s;
    uint32_t* Am = reinterpret_cast<uint32_t*>(A4);
    uint32_t* Bm = reinterpret_cast<uint32_t*>(B4);
    key_t* tab = reinterpret_cast<key_t*>(A4);
    const key_t kEmpty = ~(key_t)0;                                    // no key has all its bits set (k < 16 / < 32)
    const int k = P.kmer;
    const int kb = 2 * k;                                              // bits of a key
    const uint32_t tail_mask = KEY64 ? ~(0xFFFFFFFFu >> ((uint32_t)(kb - 32) & 31u)) : 0u;   // 64-bit keys: their bits in the second word
    const uint32_t nf = stored_frags(B);
    uint64_t drop_n = 0, drop_b = 0;
    if (pool_overflowed(B)) return;
    const int NT = kTgsfEmul ? 1 : kRepThreads, tid = kTgsfEmul ? 0 : (int)threadIdx.x;   // (emulation: one lane does the whole fragment)
    if (kTgsfEmul && threadIdx.x != 0) return;
    uint4 zero4, ones4;
    zero4.x = zero4.y = zero4.z = zero4.w = 0;
    ones4.x = ones4.y = ones4.z = ones4.w = ~0u;
    auto clear_maps = [&](bool b_too) TGSF_INLINE_LAMBDA {
        for (uint32_t w = (uint32_t)tid; w < kRepAQ; w += (uint32_t)NT) A4[w] = zero4;
        if (b_too) for (uint32_t w = (uint32_t)tid; w < kRepBQ; w += (uint32_t)NT) B4[w] = zero4;
    };
    clear_maps(true);

    auto chunks_of = [&](uint32_t f, const uint4*& base, int& a, int& L) TGSF_INLINE_LAMBDA {
        const uint8_t* s = B.seq + B.frag_off[f];
        a = (int)((uintptr_t)s & 15u);
        base = reinterpret_cast<const uint4*>(s - a);
        L = (int)B.frag_len[f];
    };
    // codes[0 .. ) <- chunks [wb, wb + W) of the fragment, zeros behind its last chunk
    auto load_window = [&](const uint4* base, int words, int wb) TGSF_INLINE_LAMBDA {
        for (int g = tid; g < W + 4 && wb + g < words + 4; g += NT) codes[g] = wb + g < words ? base_codes16(base[wb + g]) : 0u;
    };
    // the sum of a per-lane count over the workgroup (every lane gets it); slot = one of acc_s, zeroed by the caller
    auto block_sum = [&](uint32_t v, int slot) TGSF_INLINE_LAMBDA -> uint32_t {
        v = (uint32_t)wave_sum((uint64_t)v);
        if (wave_leader() && v) atomicAdd(&acc_s[slot], v);
        TGSF_BLOCK_SYNC();
        return acc_s[slot];
    };
    // the next fragment's text on its way into registers while this one is counted (the serial emulation reads windows)
    uint4 raw[kRepSlots];
    auto prefetch = [&](uint32_t f) TGSF_INLINE_LAMBDA {
        if (kTgsfEmul || f >= nf) return;
        const uint4* base; int a, L;
        chunks_of(f, base, a, L);
        const int words = (a + L + 15) / 16;
#pragma unroll
        for (int sl = 0; sl < kRepSlots; sl++) { const int g = tid + sl * NT; if (g < words) raw[sl] = base[g]; }
    };
    // work items: the long fragments in the order of their list, then every other fragment in the batch's order
    const uint32_t n_long = B.rep_next[1] < B.rep_long_cap ? B.rep_next[1] : B.rep_long_cap;
    auto take = [&]() TGSF_INLINE_LAMBDA -> uint32_t {                 // (one lane)
        for (;;) {
            const uint32_t w = atomicAdd(&B.rep_next[0], 1u);
            if (w < n_long) return B.rep_next[2 + w];
            const uint32_t g = w - n_long;
            if (g >= nf) return nf;
            if (!rep_long(P, B, g)) return g;
        }
    };
    if (tid == 0) next_s = take();
    TGSF_BLOCK_SYNC();
    uint32_t f = next_s;
    prefetch(f);
    while (f < nf) {
        const uint4* base; int a, L;
        chunks_of(f, base, a, L);
        const int total = L - k + 1;
        const int words = (a + L + 15) / 16;
        const int kwords = total > 0 ? (a + total + 15) / 16 : 0;
        const bool one_window = words <= W;
        TGSF_BLOCK_SYNC();                                             // (everyone has read next_s)
        if (tid == 0) { over_s = 0; next_s = take(); }
        TGSF_BLOCK_SYNC();
        if (kTgsfEmul) load_window(base, words, 0);
        else {
#pragma unroll
            for (int sl = 0; sl < kRepSlots; sl++) { const int g = tid + sl * NT; if (g < words) codes[g] = base_codes16(raw[sl]); }
            if (tid < 8) codes[(words < W ? words : W) + tid] = 0;
        }
        TGSF_BLOCK_SYNC();
        const uint32_t fnext = next_s;
        prefetch(fnext);
        bool drop = false;                                             // the gate's verdict (uniform over the workgroup)
        if (total <= 0) {
            drop = 0 < P.min_repeat;                                   // no k-mer: repeat = 0
        } else if (k >= 32) {
            ull first = 0;
            const uint8_t* seq = B.seq + B.frag_off[f];
            for (int i = 0; i < k; i++) first = (first << 2) | base_code(seq[i]);
            const int distinct = (total > 1 && first != 0ull) ? 2 : 1;
            drop = total - distinct < P.min_repeat;
        } else {
            uint32_t plog = 0;                                         // 2^plog passes
            uint32_t rot = 19u;                                        // (odd, one of sixteen: see hash_of)
            while (plog < 20u && ((uint32_t)total >> plog) > kRepShare) plog++;
            for (;;) {
                const uint32_t passes = 1u << plog;
                uint32_t repeat = 0;                                   // exact over the passes done so far
                uint32_t flagged = 0;
                bool decided = false;
                for (uint32_t pass = 0; pass < passes && !decided; pass++) {
                    // one scan of the fragment: PHASE 0 marks A (and B for the flagged), PHASE 1 counts and inserts the keys
                    // whose hash value is in B; MULTI: the pass owns part of the keys.  (Separate instances: a test of the
                    // phase between the LDS operations of a chunk makes the compiler wait for each of them in turn.)
                    uint32_t c0 = 0, c1 = 0;                           // PHASE 0: flagged;  PHASE 1: occurrences, insertions
                    auto scan = [&](auto phase_tag, auto multi_tag) TGSF_INLINE_LAMBDA {
                        constexpr int PHASE = decltype(phase_tag)::value;
                        constexpr bool MULTI = decltype(multi_tag)::value;
                        // what a key's hash selects
                        // (64-bit keys are hashed as they lie in the chunks: x0 = the first sixteen bases, x1 = the rest from bit 31
                        // down with the bits behind the key cleared -- no 64-bit shift, one multiplication.  The rotation is another
                        // with every restart: distinct keys that share a hash value under one share it under another four at a time
                        // at most (32-bit keys hash one to one))
                        auto hash_of = [&](uint32_t x0, uint32_t x1) TGSF_INLINE_LAMBDA -> uint32_t {
                            return KEY64 ? ((x1 & tail_mask) ^ alignbit(x0, x0, rot)) * 0x9E3779B1u : x0 * 0x9E3779B1u;
                        };
                        auto owned = [&](uint32_t h) TGSF_INLINE_LAMBDA -> bool {
                            return !MULTI || ((h ^ (h >> 7)) & (passes - 1u)) == pass;   // (low bits: the map's word comes from the top ones)
                        };
                        auto a_word = [&](uint32_t h) TGSF_INLINE_LAMBDA -> uint32_t { return ((h >> 16) * (kRepAQ * 4u)) >> 16; };
                        auto a_mask = [&](uint32_t h) TGSF_INLINE_LAMBDA -> uint32_t {
                            const uint32_t g = h ^ (h >> 15);
                            return (1u << (g & 31u)) | (1u << ((g >> 5) & 31u)) | (1u << ((g >> 10) & 31u));
                        };
                        for (int wb = 0;; wb += W - OV) {
                            if (!one_window) {
                                TGSF_BLOCK_SYNC();
                                load_window(base, words, wb);
                            }
                            TGSF_BLOCK_SYNC();
                            const bool last = wb + W >= words;
                            const int gend = last ? kwords - wb : W - OV;
                            for (int g = tid; g < gend; g += NT) {
                                const uint32_t w0 = codes[g], w1 = codes[g + 1], w2 = codes[g + 2];
                                const int first = a - 16 * (wb + g);
                                const int v = a + total - 16 * (wb + g);
                                uint32_t m = 0x55555555u;              // bit 30-2j: base j of the chunk starts a k-mer to handle below
                                if (first <= 0 && v >= 16) {
                                    // a whole chunk: sixteen keys at constant shifts, their LDS operations issued back to back;
                                    // what is left for the loop below is rare
                                    // (eight at a time: sixteen keys' hashes, masks and answers do not fit the registers)
                                    m = 0;
#pragma unroll
                                    for (int j0 = 0; j0 < 16; j0 += 8) {
                                        uint32_t hv[8], oldv[8], mkv[8];
#pragma unroll
                                        for (int jj = 0; jj < 8; jj++) {
                                            const int j = j0 + jj;
                                            const uint32_t x0 = j ? alignbit(w0, w1, 32u - 2u * (uint32_t)j) : w0;
                                            uint32_t h;
                                            if (KEY64) {
                                                const uint32_t x1 = j ? alignbit(w1, w2, 32u - 2u * (uint32_t)j) : w1;
                                                h = hash_of(x0, x1);
                                            } else {
                                                h = hash_of(x0 >> (32 - kb), 0u);
                                            }
                                            hv[jj] = h;
                                            if (PHASE == 0) {
                                                mkv[jj] = owned(h) ? a_mask(h) : 0u;
                                                oldv[jj] = atomicOr(&Am[a_word(h)], mkv[jj]);
                                            } else {
                                                oldv[jj] = Bm[h >> 19];
                                            }
                                        }
                                        uint32_t hits = 0;
#pragma unroll
                                        for (int jj = 0; jj < 8; jj++) {
                                            bool hit;
                                            if (PHASE == 0) hit = (mkv[jj] & ~oldv[jj]) == 0u && (!MULTI || mkv[jj] != 0u);   // every bit of the mask was there
                                            else hit = ((oldv[jj] >> ((hv[jj] >> 14) & 31u)) & 1u) && owned(hv[jj]);
                                            if (hit) hits |= 1u << (30 - 2 * (j0 + jj));
                                        }
                                        if (PHASE == 0) {
                                            if (hits) {                // flagged occurrences (rare): counted, their hash values into B
                                                c0 += popc32(hits);
#pragma unroll
                                                for (int jj = 0; jj < 8; jj++)
                                                    if (hits & (1u << (30 - 2 * (j0 + jj)))) atomicOr(&Bm[hv[jj] >> 19], 1u << ((hv[jj] >> 14) & 31u));
                                            }
                                        } else {
                                            m |= hits;
                                        }
                                    }
                                } else {
                                    if (first > 0) m &= 0xFFFFFFFFu >> (2 * first);
                                    if (v < 16) m &= ~(0xFFFFFFFFu >> (2 * v));
                                }
                                while (m) {
                                    const int b = __builtin_ctz(m);
                                    m &= m - 1u;
                                    const uint32_t sb = (uint32_t)(30 - b);    // bit offset of the key in the chunk sequence w0 w1 w2
                                    const uint32_t x0 = sb ? alignbit(w0, w1, 32u - sb) : w0;
                                    key_t key;
                                    uint32_t lo32, hi32 = 0, h;
                                    if (KEY64) {
                                        const uint32_t x1 = sb ? alignbit(w1, w2, 32u - sb) : w1;
                                        const ull k64 = (((ull)x0 << 32) | x1) >> (64 - kb);
                                        key = (key_t)k64;
                                        lo32 = (uint32_t)k64; hi32 = (uint32_t)(k64 >> 32);
                                        h = hash_of(x0, x1);
                                    } else {
                                        lo32 = x0 >> (32 - kb);
                                        key = (key_t)lo32;
                                        h = hash_of(lo32, 0u);
                                    }
                                    if (!owned(h)) continue;           // another pass's k-mer
                                    const uint32_t bbit = 1u << ((h >> 14) & 31u);
                                    if (PHASE == 0) {
                                        const uint32_t mk = a_mask(h);
                                        const uint32_t old = atomicOr(&Am[a_word(h)], mk);
                                        if ((mk & ~old) == 0u) { c0++; atomicOr(&Bm[h >> 19], bbit); }
                                    } else if (Bm[h >> 19] & bbit) {
                                        c0++;
                                        uint32_t slot = (((lo32 * 0xC2B2AE35u) ^ (hi32 * 0x27D4EB2Fu) ^ (lo32 >> 15)) * 0x165667B1u) >> (32u - TLOG);
                                        for (int probes = 0;; probes++) {
                                            const key_t old = atomicCAS(&tab[slot], kEmpty, key);
                                            if (old == kEmpty) { c1++; break; }
                                            if (old == key) break;
                                            if (probes >= 64) { over_s = 1; break; }
                                            slot = (slot + 1u) & ((1u << TLOG) - 1u);
                                        }
                                    }
                                }
                            }
                            if (last) break;
                        }
                    };
                    if (tid == 0) acc_s[0] = acc_s[1] = acc_s[2] = 0;
                    // (the scan's first barrier stands between this and the first use)
                    if (plog == 0) scan(std::integral_constant<int, 0>(), std::false_type());
                    else scan(std::integral_constant<int, 0>(), std::true_type());
                    TGSF_BLOCK_SYNC();
                    flagged = block_sum(c0, 0);
                    // no occurrence flagged: no repeat in this pass.  Fewer than -p could still be missing at the end of
                    // the last pass: dropped without counting them
                    const bool skip = flagged == 0 || (pass + 1 == passes && repeat + flagged < (uint32_t)P.min_repeat);
                    if (skip) {
                        if (flagged) repeat += flagged;                // an upper bound, on the last pass only
                        TGSF_BLOCK_SYNC();
                        clear_maps(flagged != 0);
                        TGSF_BLOCK_SYNC();
                        continue;
                    }
                    for (uint32_t w = (uint32_t)tid; w < kRepTabQ; w += (uint32_t)NT) A4[w] = ones4;    // A becomes the (empty) table
                    c0 = c1 = 0;
                    if (plog == 0) scan(std::integral_constant<int, 1>(), std::false_type());
                    else scan(std::integral_constant<int, 1>(), std::true_type());
                    TGSF_BLOCK_SYNC();
                    const uint32_t occ = block_sum(c0, 1), ins = block_sum(c1, 2);
                    clear_maps(true);
                    TGSF_BLOCK_SYNC();
                    if (over_s) break;
                    repeat += occ - ins;
                    if (repeat >= (uint32_t)P.min_repeat) decided = true;         // accepted, whatever the other passes hold
                }
                if (!over_s) { drop = !decided && repeat < (uint32_t)P.min_repeat; break; }
                // A pass's table could not take the keys the pass had to compare in full: more passes -- as many as leave a
                // pass's flagged occurrences half the table, twice as many at least -- and the fragment starts over.
                TGSF_BLOCK_SYNC();
                if (tid == 0) over_s = 0;
                uint32_t want = plog + 1u;
                while (want < B.rep_max_plog && (flagged >> (want - plog)) > (1u << (TLOG - 1u))) want++;
                if (plog >= B.rep_max_plog) {
                    // No number of passes the LDS table can afford separates this fragment's duplicated k-mers (thousands of
                    // distinct ones whichever way the key is hashed): count its distinct k-mers the plain way, as the
                    // reference does (:1703-1753, an unordered_set of all of them) -- every key, in full, into an
                    // open-addressing set in memory, one workgroup at a time.  repeat = k-mers - distinct ones, exactly.
                    const uint32_t distinct = rep_distinct_in_memory(B, B.seq + B.frag_off[f], total, k, tid, NT, &acc_s[0]);
                    drop = (uint32_t)total - distinct < (uint32_t)P.min_repeat;
                    break;
                }
                plog = want;
                rot = (rot + 6u) & 31u;
                TGSF_BLOCK_SYNC();
            }
        }
        if (tid == 0 && drop) {                                        // :1984-1988
            B.frag_flags[f] |= TGSF_FF_REPEAT;
            drop_n++; drop_b += (uint64_t)L;
        }
        f = fnext;
    }
    if (tid == 0 && drop_n) {
        atomicAdd((ull*)&B.ctr[TGSF_CTR_DROPINFO + 15], (ull)drop_n);
        atomicAdd((ull*)&B.ctr[TGSF_CTR_DROPINFO + 16], (ull)drop_b);
    }
}

// ---------------------------------------------------------------------------
// k_gate_frags: the post-split quality gate (src/TGSFilter.cpp:1995-2002).
// ---------------------------------------------------------------------------
TGSF_KERNEL k_gate_frags(DevParams P, DevBatch B)
{
    if (pool_overflowed(B)) return;
    TGSF_SHARED ull hq[TGSF_N_QBINS];
    for (uint32_t i = TGSF_COOP_BEGIN; i < (uint32_t)TGSF_N_QBINS; i += TGSF_COOP_STRIDE) hq[i] = 0;
    TGSF_BLOCK_SYNC();
    const uint32_t nf = stored_frags(B);
    const bool diff = clean_by_difference(B);
    uint64_t lq_n = 0, lq_b = 0;
    uint32_t erows = 0;
    for (uint32_t f0 = blockIdx.x * blockDim.x; f0 < nf; f0 += gsize()) {
        const uint32_t f = f0 + threadIdx.x;
        if (f >= nf || (B.frag_flags[f] & TGSF_FF_REPEAT)) continue;
        const uint32_t L = B.frag_len[f];
        if (diff && B.whole[B.frag_read[f]]) B.frag_sum[f] = clean_by_product(B) ? B.spec_sum[B.frag_read[f]] : B.sumq[B.frag_read[f]];   // not re-scanned
        const double cm = mean_q(B.frag_sum[f], L);
        if (!P.no_qual) {                                                 // :1992 rawQualLen > 0
            if (P.filter && q_fail(cm, P.min_q, P.max_q)) { lq_n++; lq_b += L; continue; }
            if (!(cm >= 0.0 && cm < 256.0)) { set_status(B, DS_BAD_MEANQ, B.frag_read[f]); continue; }
            atomicAdd(&hq[(int)cm], (ull)L);                              // :2002
        }
        B.frag_flags[f] = TGSF_FF_PASS;
        uint32_t er = (uint32_t)P.bc_len < L ? (uint32_t)P.bc_len : L;
        erows = er > erows ? er : erows;
    }
    wave_add_u64(&B.ctr[TGSF_CTR_DROPINFO + 13], lq_n);
    wave_add_u64(&B.ctr[TGSF_CTR_DROPINFO + 14], lq_b);
    wave_max_u64(&B.ctr[TGSF_CTR_ROWS + 3], erows);
    TGSF_BLOCK_SYNC();
    for (uint32_t i = TGSF_COOP_BEGIN; i < (uint32_t)TGSF_N_QBINS; i += TGSF_COOP_STRIDE)
        if (hq[i]) atomicAdd((ull*)&B.ctr[TGSF_CTR_CLEAN_DIFFQ + i], hq[i]);
}

// ---------------------------------------------------------------------------
// k_ctr_merge: tallies of one context added to another's (sums; the four "rows used" words are maxima).
// ---------------------------------------------------------------------------
TGSF_KERNEL k_ctr_merge(uint64_t* dst, const uint64_t* src, uint64_t n)
{
    for (uint64_t i = gtid(); i < n; i += gsize()) {
        const uint64_t a = dst[i], b = src[i];
        dst[i] = (i >= TGSF_CTR_ROWS && i < TGSF_CTR_ROWS + 4) ? (a > b ? a : b) : a + b;
    }
}

// ---------------------------------------------------------------------------
// k_finalize: the records handed back across the C ABI.
// ---------------------------------------------------------------------------
TGSF_KERNEL k_finalize(DevBatch B, tgsf_read_result* out_reads, tgsf_fragment* out_frags,
                       uint32_t out_fcap, uint32_t* out_nfrags)
{
    if (pool_overflowed(B)) {
        // no record of this batch is final: tgsf_wait runs it again.  A caller of tgsf_submit_device that looks at its
        // buffers before that finds TGSF_NFRAGS_NOT_FINAL in the fragment count.
        if (gtid() == 0 && out_nfrags) *out_nfrags = TGSF_NFRAGS_NOT_FINAL;
        return;
    }
    const uint32_t nf = B.nfr[B.n];
    if (gtid() == 0) {
        if (out_nfrags) *out_nfrags = nf;
        if (nf > out_fcap) set_status(B, DS_FRAG_CAP, nf);
    }
    for (uint32_t r = gtid(); r < B.n; r += gsize()) {
        tgsf_read_result o;
        o.sum_q = B.sumq[r];
        o.flags = B.flags[r];
        o.n_frags = B.nfr[r + 1] - B.nfr[r];
        o.frag_begin = B.nfr[r];
        o.trimmed = B.trimmed[r];
        o.reserved0 = 0; o.reserved1 = 0;
        out_reads[r] = o;
    }
    for (uint32_t f = gtid(); f < nf && f < out_fcap; f += gsize()) {
        tgsf_fragment o;
        o.sum_q = B.frag_sum[f];
        o.read = B.frag_read[f];
        o.start = B.frag_start[f];
        o.len = (int32_t)B.frag_len[f];
        o.flags = B.frag_flags[f];
        out_frags[f] = o;
    }
}

// ---------------------------------------------------------------------------
// k_align_windows: stand-alone edlib-compatible alignments (tgsf_align_windows).
// ---------------------------------------------------------------------------
template <int MAXNW>
TGSF_KERNEL k_align_windows(DevParams P, DevBatch B, const uint8_t* seq, const uint64_t* win_off, const uint32_t* win_len,
                            const uint8_t* adapter_id, const int32_t* kk, uint32_t n, int32_t* res, int32_t* ends)
{
    const uint32_t i = gtid();
    if (i >= n) return;
    const int a = adapter_id[i];
    const int Q = P.Q[a];
    int k = kk[i];
    if (k > Q) k = Q;                                    // edlib.cpp:565-567
    const LaneScratch sc = lane_scratch(B, 0);
    const uint8_t* t = seq + win_off[i];
    WinAln w = align_window_any<MAXNW>(P, a, t, (int)win_len[i], k, 0, sc, true);
    res[i * 4 + 0] = w.best;
    res[i * 4 + 1] = w.best < 0 ? 0 : w.n;
    res[i * 4 + 2] = w.best < 0 ? 0 : w.mlen + w.best;
    res[i * 4 + 3] = w.best < 0 ? -1 : w.start0;
    ends[i * 2 + 0] = w.best < 0 ? -1 : w.first_end;
    ends[i * 2 + 1] = w.best < 0 ? -1 : w.last_end;
}

}  // namespace tgsf
