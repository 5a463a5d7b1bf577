// tgsf_dev.h -- device-side data layout shared by the kernels and the host API.
#pragma once
#include <stdint.h>
#include "tgsf_core.h"

namespace tgsf {

// Resolved parameters + adapter tables, passed to kernels by value (kernarg).
struct DevParams {
    int min_len, max_len;
    float min_q, max_q;
    int bc_len, head_trim, tail_trim, end_len, end_match_len, mid_match_len, extra_len;
    float end_sim, mid_sim;
    int discard, filter, only_qc, qtype;
    int no_qual;               // records without qualities (FASTA): count-only tallies, no quality gate
    int min_repeat, kmer;             // -p / -k: repeat gate (GetKmerCount), 0 = off
    int n_adapters;
    int max_nw;                       // words of the longest adapter's column: 1, 2, 4, or kWideNW (an adapter beyond 256 bp)
    int Q[kMaxAdapters];
    int k_mid[kMaxAdapters];          // min(Q, Q - MidMatchLen + 1)   src/TGSFilter.cpp:1233, edlib.cpp:565
    int k_end[kMaxAdapters];          // min(Q, Q - EndMatchLen + 1)   :1271
    int w5[kMaxAdapters];             // EndLen + int(Q / EndSim)      :1267 (before clamping to L)
    int need_end[kMaxAdapters];       // smallest mlen with mlen >= EndMatchLen && float(mlen)/Q >= EndSim  (:1283-1288)
    int need_mid[kMaxAdapters];       // smallest mlen with mlen >= MidMatchLen && float(mlen)/Q >= MidSim  (:1246-1252)
    int min_Q;                        // shortest adapter
    int seg_cols;                     // columns of a read's middle owned by one lane of the infix scan
    uint32_t n_bins;                  // rows of the 100-bp tables
    const uint8_t* adapter;           // [kMaxAdapters][kMaxQ] bytes
    const uint64_t* peq_fwd;          // [kMaxAdapters][256][2]  standard layout
    const uint64_t* peq_rev;          // [kMaxAdapters][256][2]  reversed adapter
    const uint64_t* peq_top;          // [kMaxAdapters][256]     top-aligned single word (Q <= 64)
    const uint64_t* peq_fwd_w;        // [n_adapters][256][kWideNW]  standard layout, every adapter (nullptr unless max_nw == kWideNW)
    const uint64_t* peq_rev_w;        // ... reversed adapter
};

// Candidate column of the middle scan / resolved drop region (same 16-byte slot).
struct MidCand {
    int32_t pos;      // scan: end column in window coordinates;  resolved: region start (read coords)
    int32_t aux;      // scan: score | adapter << 16;             resolved: region end
    int32_t next;     // next slot of the same read, -1 = end of list
    int32_t state;    // 0 = candidate, 1 = resolved region, 2 = rejected
};

// Everything a batch needs on the device.  Arrays are sized for max_batch_reads /
// max_batch_bases at context creation.
struct DevBatch {
    const uint8_t* seq;
    const uint8_t* qual;
    const uint64_t* off;       // [n] (or [n+1] when len_in == nullptr)
    const uint64_t* qoff;      // [n] start of the qualities of read i in `qual` (== off unless the caller says otherwise)
    const uint32_t* len_in;    // optional explicit lengths
    uint32_t n;
    uint64_t n_bytes;

    uint32_t* len;             // [n] resolved lengths
    uint64_t* sumq;            // [n] raw sumQ
    uint32_t* flags;           // [n] TGSF_RF_*
    int32_t*  clip5;           // [n*A] 5' hit: end of drop region (0 = no hit)
    int32_t*  clip3;           // [n*A] 3' hit: start of drop region (-1 = no hit)
    int32_t*  mid_head;        // [n] head of the candidate list, -1
    int32_t*  mid_best;        // [n*A] best middle-scan value any lane has handed over so far (prunes the hand-overs)
    MidCand*  pool;            // candidate / region pool
    uint32_t  pool_cap;
    uint32_t* pool_n;          // number of slots used
    uint32_t  mid_mode;        // middle scan: 0 = one pass, candidates go to per-read lists in the order the lanes get to them
                               // (the pool usually holds them all, and a read has a handful).  When the pool overflows or a
                               // read collects more than kMidListMax of them (*ovf) the scan runs twice more with
                               // mid_best known: 1 = every lane counts the columns at its (read, adapter)'s minimum into
                               // seg_n, 2 = after a prefix sum over seg_n the lanes write exactly those columns at their
                               // own offsets: the pool (grown to fit) then holds every read's candidates as one array,
                               // in ascending order of position -- which the walks behind the scan rely on for long lists
    uint32_t* mid_cnt;         // [n] candidates handed over per read (mode 0)
    uint32_t* mid_gate;        // [n*A] (modes 1, 2) 1 = the first location's path passes the gates: every location counts
    uint32_t* seg_n;           // [segments*A + 1] (modes 1, 2) per (lane, adapter): count, then first slot
    uint32_t* seg_cnt;         // [n+1] middle segments per read, scanned in place to bases
    uint32_t* chk_cnt;         // [n+1] 16-column chunks of each read's middle window (counted from the window's first column),
                               // scanned in place: the flat scan (k_mid_flat) deals stretches of this one sequence of chunks
    uint32_t  flat_pmax, flat_pmin;   // chunks of the longest / shortest stretches, powers of two (see flat_schedule)
    uint32_t* chk_mark;        // [4][mark_stride] one bit per chunk of that sequence and adapter of a pass of the filtering flat scan: the LAST 32
                               // rows of the adapter came within k of a column of the chunk: k_mid_recheck puts the whole adapter there
    uint32_t  mark_stride;     // words per adapter (this batch's chunks / 32, rounded up)
    uint32_t* rc_list;         // [rc_cap] the marks of a pass as one list (chunk * 4 + the adapter's place in the pass): k_mid_marks -> k_mid_recheck
    uint32_t* rc_n;            // entries asked for (beyond rc_cap: rechecked by k_mid_marks itself)
    uint32_t  rc_cap;
    uint32_t  flat_f0;         // 256ths of the sequence dealt in the longest stretches
    uint32_t* nfr;             // [n+1] fragments per read, scanned in place to frag_begin
    uint32_t* scan_part;       // [n / kScanTile + 2] per-tile totals of the prefix scans
    uint32_t* trimmed;         // [n]
    uint32_t* rep_next;        // [2 + rep_long_cap] k_repeat*: [0] next work item to hand out, [1] number of long fragments
                               // (k_repeat_keys takes those that need several passes first: k_repeat_long lists them), [2..] the list
    uint32_t  rep_long_cap;
    unsigned long long* rep_tab;   // [1 << rep_tab_log2] k_repeat_keys: the set of a fragment's k-mers, in full, in memory -- for the fragment whose
    uint32_t  rep_tab_log2;        // duplicated k-mers no number of passes through the LDS table separates (one workgroup at a time:
    uint32_t* rep_lock;            // rep_lock)
    uint32_t  rep_max_plog;        // passes (log2) beyond which that happens (kRepMaxPlog; TGSF_REP_MAX_PLOG: tests)

    // stats work lists (raw: items are reads; clean: items are fragments [0,fcap) and, in the
    // "difference" strategy, whole reads to take back out, numbered fcap + read)
    // Two SEGMENTS (round 6): the raw pass of a context that may speculate (bp_allowed) lists the items the batch
    // speculates on first (segment 0) and the others behind them (segment 1), each sorted by tile count and laid out tile
    // index major, so that a wave of k_stats changes between the two kinds of item at most once.  Every other pass has
    // segment 0 only.  Segment s's buckets sit at [s * (max_tiles + 2), ...).
    uint32_t* tile_hist;       // [2][max_tiles+2]
    uint32_t* tile_cnt;        // [2][max_tiles+2]  cnt[t] = #items (of the segment) with more than t tiles
    uint32_t* tile_base;       // [2][max_tiles+2]  exclusive prefix of cnt
    uint32_t* tile_fill;       // [2][max_tiles+2]
    uint32_t* seg_info;        // [4] written by k_tile_scan: [0] first slot of segment 1 in perm, [1] work items of segment 0, [2] work items of both
    uint32_t* perm;            // items sorted by (segment,) tile count, descending
    uint4*    work;            // [2*work_cap] stats work items: {seq addr lo, hi, bases | tile<<13, item}, {qual addr lo, hi, 0, 0}
    uint32_t  work_cap;
    uint32_t  max_tiles;

    // fragments
    uint64_t* frag_off;        // [fcap] absolute byte offset of the fragment
    uint64_t* frag_qoff;       // [fcap] ... and of its qualities
    uint32_t* frag_len;        // [fcap]
    uint64_t* frag_sum;        // [fcap]
    uint32_t* frag_read;       // [fcap]
    int32_t*  frag_start;      // [fcap]
    uint32_t* frag_flags;      // [fcap]
    uint32_t  fcap;

    // clean-table strategy of this batch (see k_clean_plan)
    uint64_t* raw_tab;         // [2][n_bins*5] this batch's raw bin tallies: quality sums, then counts
    uint32_t* whole;           // [n] 1 = the read is kept whole (one fragment == the read, not dropped)
    uint64_t* plan;            // [4] {batch raw rows, -, bases to scan directly, bases to scan by difference}
    uint32_t  clean_force;     // 0 = choose per batch, 1 = always direct, 2 = always by difference
    // The clean tables as a BY-PRODUCT of the raw pass (round 5; see k_stats): a read that will probably be kept as one
    // fragment [head_trim, L - tail_trim) -- no adapter beyond the fixed trims, past both quality gates: what the raw pass
    // can know beforehand -- has that fragment tallied into the clean tables while its bytes are in LDS for the raw ones.
    // Afterwards only the reads that turned out otherwise are scanned again (taken back out, their real fragments put in).
    uint32_t  bp_allowed;      // this context may speculate at all (filtering run with fixed trims, no repeat gate; TGSF_CLEAN_TABLES)
    uint32_t* spec;            // [n] 1 = the raw pass tallies [head_trim, L - tail_trim) of this read into the clean tables
    uint64_t* spec_sum;        // [n] clean sum of (qual - qType) over that range
    uint32_t* bp_state;        // [2] (lives across batches) [0] speculate in the next batch: 1 = yes (set by k_clean_plan from how
                               // this batch's speculation fared); [1] batches that speculated
    uint32_t* bp_used;         // [1] THIS batch's word (one per enqueued batch, beside ovf): whether it speculated -- a second
                               // run of the batch after a pool overflow must take the same decisions

    uint64_t* scratch;         // traceback columns, one region per wave: [column][word][lane]
    size_t    scratch_wave_words;
    size_t    scratch_mid_wave0; // first wave region of k_mid_resolve (after those of k_end_windows)

    uint64_t* ctr;             // the flat tally vector (include/tgsf.h layout)
    uint32_t* status;          // [4] device-side words: [0] error code, [1] detail
    uint32_t* ovf;             // [1] THIS batch's word: the candidate pool overflowed (the kernels behind the middle scan then
                               // leave the batch alone: tgsf_wait runs the batch again, see mid_mode and `replay`)
    uint32_t  replay;          // 1 = the batch is run a second time after such an overflow: what its first run added to the
                               // tallies in front of the middle scan (raw tables, low-quality reads) is not added again
};

// include/tgsf.h layout of the tally vector, callable from device code
TGSF_HD size_t ctr_end_table(int t, int bc_len) { return (size_t)533 + (size_t)t * (size_t)bc_len * 5u; }
TGSF_HD size_t ctr_bin_table(int b, int bc_len, uint32_t n_bins) {
    return (size_t)533 + 8u * (size_t)bc_len * 5u + (size_t)b * (size_t)n_bins * 5u;
}


// The stretches k_mid_flat deals (see there): the chunk sequence [0, T) is cut into phases; phase 0 covers f0 / 256 of it
// in stretches of pmax chunks, every later phase half of what is left in stretches half as long, the last (pmin chunks)
// all the rest -- long stretches while there is plenty (one warm-up per several thousand columns), short ones at the
// end (the launch ends everywhere within one short stretch).  A phase is a whole number of groups of 64 equal
// stretches: the 64 lanes of a wave always work the same number of chunks.  pmax and pmin are powers of two (shifts only:
// every lane of the scan evaluates this).
struct FlatSchedule {
    uint32_t nph;
    uint32_t sh[8];            // log2 of the stretch length of the phase
    uint32_t c0[9];            // first chunk of the phase (c0[nph] = T)
    uint32_t d0[9];            // first stretch of the phase (d0[nph] = number of stretches, a multiple of 64)
};
TGSF_HD uint32_t flat_log2(uint32_t v) { uint32_t l = 0; while ((2u << l) <= v) l++; return l; }
TGSF_HD void flat_schedule(uint32_t T, uint32_t pmax, uint32_t pmin, uint32_t f0, FlatSchedule& s)
{
    uint64_t c = 0, d = 0, rest = T;
    const uint32_t shmin = flat_log2(pmin ? pmin : 1u);
    uint32_t sh = flat_log2(pmax ? pmax : 1u), k = 0;
    if (sh < shmin) sh = shmin;
    for (;;) {
        s.sh[k] = sh; s.c0[k] = (uint32_t)c; s.d0[k] = (uint32_t)d;
        const bool last = sh <= shmin || k == 7;
        const uint32_t gsh = sh + 6;                                    // a group: 64 stretches
        const uint64_t want = last ? rest : (k == 0 ? (rest * f0) >> 8 : rest >> 1);
        const uint64_t ng = last ? (rest + (1ull << gsh) - 1) >> gsh : want >> gsh;
        const uint64_t cover = (ng << gsh) < rest ? (ng << gsh) : rest;
        c += cover; d += ng << 6; rest -= cover; k++;
        if (last || rest == 0) break;
        sh--;
    }
    s.nph = k; s.c0[k] = (uint32_t)c; s.d0[k] = (uint32_t)d;
}
// The same schedule for ONE stretch, without the arrays (a lane of k_mid_flat asks for its own: arrays indexed by the
// phase lived in scratch memory, 112 bytes a lane): the phase stretch d falls in -- its shift, first chunk and first
// stretch, and the chunk the phase ends at.  Returns the number of stretches of the whole schedule; d beyond it leaves
// `sh`, `c0`, `d0`, `c1` at the last phase's values (the caller tests d against the count).
TGSF_HD uint32_t flat_stretch(uint32_t T, uint32_t pmax, uint32_t pmin, uint32_t f0, uint32_t dq,
                              uint32_t& sh_out, uint32_t& c0_out, uint32_t& d0_out, uint32_t& c1_out)
{
    uint64_t c = 0, d = 0, rest = T;
    const uint32_t shmin = flat_log2(pmin ? pmin : 1u);
    uint32_t sh = flat_log2(pmax ? pmax : 1u), k = 0;
    if (sh < shmin) sh = shmin;
    sh_out = sh; c0_out = 0; d0_out = 0; c1_out = 0;
    for (;;) {
        const bool last = sh <= shmin || k == 7;
        const uint32_t gsh = sh + 6;
        const uint64_t want = last ? rest : (k == 0 ? (rest * f0) >> 8 : rest >> 1);
        const uint64_t ng = last ? (rest + (1ull << gsh) - 1) >> gsh : want >> gsh;
        const uint64_t cover = (ng << gsh) < rest ? (ng << gsh) : rest;
        if (dq >= (uint32_t)d) { sh_out = sh; c0_out = (uint32_t)c; d0_out = (uint32_t)d; c1_out = (uint32_t)(c + cover); }
        c += cover; d += ng << 6; rest -= cover; k++;
        if (last || rest == 0) break;
        sh--;
    }
    return (uint32_t)d;
}

enum DevStatus : uint32_t {
    DS_OK = 0,
    DS_BAD_LEN = 1,        // read length 0 or > max_read_len
    DS_RESERVED_2 = 2,     // (was DS_BAD_QUAL until round 5: a quality byte of 128 and above stands for its value - 256, as in the reference)
    DS_POOL_FULL = 3,      // candidate pool overflow
    DS_TOO_MANY_REGIONS = 4,
    DS_FRAG_CAP = 5,
    DS_BAD_MEANQ = 6,      // mean quality outside [0,256): the reference indexes out of bounds
    DS_REPEAT_TABLE = 7,   // (no longer raised: such a fragment's k-mers are counted in memory, rep_distinct_in_memory)
};

}  // namespace tgsf
