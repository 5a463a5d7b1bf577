/*
 * tgsf.h -- C ABI of the MI355X per-read filtering hot path (libtgsf.so).
 *
 * This is the drop-in boundary for TGSFilter's worker body.  The reference has no
 * FFI of its own: the seam this ABI replaces is
 *
 *   S1  void TGSFilterTask::filter_sequence(int tid)      src/TGSFilter.cpp:1919-2064
 *         (per-read: CalcAvgQuality :1436, Get_5p/3p_base_qual :1481/:1528,
 *          adapterMap :1325, GetEditDistance :1218) and
 *   S2  EdlibAlignResult edlibAlign(query,qLen,target,tLen,config)
 *                                                          include/edlib.h:242-246
 *         as called in HW/PATH mode from src/TGSFilter.cpp:1239,1276,1301.
 *
 * The reference processes one read per call on std::string values; a GPU needs
 * batches, so the ABI is the batch-granular form of S1 with S2 folded in:
 * reads arrive CSR-packed (two byte buffers + offsets), results come back as
 * "kept-read set + trim coordinates" (per-read record + per-fragment records)
 * and the additive tallies (DropInfo[17] + QC accumulator tables) live in the
 * context until fetched with tgsf_counters().
 *
 * Conventions: plain C, fixed-width ints, no exceptions; every entry point
 * returns 0 (TGSF_OK) or a negative tgsf_status; tgsf_last_error() gives text.
 * One context per GPU.  Calls on one context are serialised by the caller;
 * different contexts may be driven concurrently from different host threads.
 * The library owns all device memory it allocates.  There is NO CPU fallback:
 * if no HIP device is usable tgsf_create fails with TGSF_E_NO_DEVICE.
 */
#ifndef TGSF_H
#define TGSF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TGSF_ABI_VERSION 3
#define TGSF_MAX_ADAPTERS 32      /* adapters.size(): 2 or 4 in practice (src/TGSFilter.cpp:3105-3125) */
#define TGSF_MAX_ADAPTER_LEN 8192 /* library adapters are 22..64 bp (:2970-2991); up to 256 bp (four 64-row blocks) the searches run in
                                     registers, beyond that (-a accepts any length) in word arrays walked by run-time loops: correct, not
                                     fast.  The first location's path: by traceback below 1 MiB of traceback state, by Hirschberg's
                                     divide and conquer beyond, exactly as edlib chooses (include/edlib.cpp:1191-1210) */
#define TGSF_N_DROPINFO 17        /* DropInfo row, src/TGSFilter.cpp:1776 */
#define TGSF_N_QBINS 256          /* raw/cleanDiffQualReadsBases, :1777-1778 */
#define TGSF_BIN_WIDTH 100        /* int(i/100) in CalcAvgQuality, :1460 */

typedef enum tgsf_status {
    TGSF_OK = 0,
    TGSF_E_INVALID = -1,      /* bad argument / parameter outside the supported domain */
    TGSF_E_NO_DEVICE = -2,    /* no usable HIP device (there is no CPU fallback)       */
    TGSF_E_HIP = -3,          /* a HIP runtime call failed                              */
    TGSF_E_CAPACITY = -4,     /* caller-provided output capacity too small              */
    TGSF_E_UNSUPPORTED = -5,  /* feature named by the reference but not on this path    */
    TGSF_E_DATA = -6          /* input the reference itself has no defined answer for (see tgsf_batch_in) */
} tgsf_status;

/*
 * The fields of Para_A24 (src/TGSFilter.cpp:82-172) and the globals the hot path
 * reads, AFTER the pre-pass resolved them (MinQ, HeadTrim/TailTrim >= 0, qType,
 * the adapter set).  Floats stay floats: the reference compares a double mean
 * quality against a float threshold (:1947) and float(mlen)/qLen against a
 * float similarity (:1250-1252), and so does the library.
 */
typedef struct tgsf_params {
    uint32_t struct_size;     /* = sizeof(tgsf_params); ABI guard                    */
    int32_t  min_len;         /* -l  MinLen                                           */
    int32_t  max_len;         /* -L  MaxLen                                           */
    float    min_q;           /* -q  MinQ (resolved, >= 0)                            */
    float    max_q;           /* -Q  MaxQ                                             */
    int32_t  bc_len;          /* -e  BCLen: positions in the 5'/3' QC tables          */
    int32_t  head_trim;       /* -5  HeadTrim (resolved; <= 0 disables, :1334)        */
    int32_t  tail_trim;       /* -3  TailTrim                                         */
    int32_t  end_len;         /* -E  EndLen                                           */
    int32_t  end_match_len;   /* -m  EndMatchLen                                      */
    int32_t  mid_match_len;   /* -M  MidMatchLen                                      */
    int32_t  extra_len;       /* -T  ExtraLen                                         */
    float    end_sim;         /* -s  EndSim                                           */
    float    mid_sim;         /* -S  MidSim                                           */
    int32_t  discard;         /* -D  discard reads with a middle adapter              */
    int32_t  filter;          /* Para_A24::Filter (0 with -F / --qc)                  */
    int32_t  only_qc;         /* --qc                                                 */
    int32_t  min_repeat;      /* -p  MinRepeat (0 disables the repeat gate)           */
    int32_t  kmer;            /* -k  Kmer, 1..32 when min_repeat > 0 (GetKmerCount :1703-1753) */
    int32_t  qtype;           /* global qType: 33 or 64 (:80, :1042-1053)             */
    int32_t  n_adapters;      /* size of the global `adapters` set (:1324)            */
    const char* adapters[TGSF_MAX_ADAPTERS];     /* not NUL-terminated necessarily    */
    int32_t  adapter_len[TGSF_MAX_ADAPTERS];
    /* sizing hints (not in the reference) */
    uint64_t max_batch_bases; /* largest sum of read lengths of one batch            */
    uint32_t max_batch_reads; /* largest number of reads of one batch                */
    uint32_t max_read_len;    /* longest read ever submitted (sizes the bin tables)  */
    /* FASTA input (records without qualities, src/TGSFilter.cpp:1954-1958, :2005-2009): tgsf_batch_in.qual is
     * ignored (may be NULL); the tallies are those of Get_base_counts (:1577-1606), Get_5p_base_counts
     * (:1608-1640, where a lower-case 'g' is not counted as G) and Get_3p_base_counts (:1642-1678, where a
     * lower-case 't' is not counted as T): counts only, all quality sums, sum_q and both DiffQual
     * histograms stay 0; there is no quality gate. */
    int32_t  no_qual;
    int32_t  reserved;
} tgsf_params;

/*
 * One batch of reads, CSR-packed.  Read i is seq[offsets[i] .. offsets[i+1]) and
 * its qualities are qual[offsets[i] .. offsets[i+1]) -- one offsets array for
 * both, because the reference drops the stream at the first record whose
 * lengths differ (src/TGSFilter.cpp:719-723).  Offsets are in bytes; aligning
 * each read start to 16 bytes (leaving gaps) is allowed and is the fast layout:
 * then pass `lengths` explicitly.  With lengths == NULL, read i has length
 * offsets[i+1]-offsets[i].  Supported domain: 1 <= length <= max_read_len.
 * Quality bytes are taken as the reference takes them: `qual[i] - qType` on a
 * (signed) char (src/TGSFilter.cpp:1455-1457, :1508), so a byte of 128 and above
 * stands for its value - 256.  A read whose mean of those values falls outside
 * [0, 256) fails the batch with TGSF_E_DATA: the reference indexes
 * rawDiffQualReadsBases[int(mean)] out of bounds there (:1943).
 */
typedef struct tgsf_batch_in {
    const uint8_t*  seq;
    const uint8_t*  qual;
    const uint64_t* offsets;   /* n_reads + 1 entries (n_reads if lengths given)     */
    const uint32_t* lengths;   /* optional, n_reads entries                           */
    uint32_t        n_reads;
    uint32_t        reserved;
    uint64_t        n_bytes;   /* bytes spanned by seq / qual (>= last offset+len)    */
    /* Optional: qualities of read i start at qual[qual_offsets[i]] instead of qual[offsets[i]]
     * (n_reads entries; requires `lengths`).  With seq == qual this lets both streams be read IN
     * PLACE from one buffer holding the raw FASTQ text: the caller only indexes the records. */
    const uint64_t* qual_offsets;
} tgsf_batch_in;

/* tgsf_read_result.flags */
#define TGSF_RF_LOWQ       0x01u  /* dropped by the raw mean-quality gate (:1946-1953)        */
#define TGSF_RF_AD5P       0x02u  /* num5p  > 0 (:1354-1370)                                  */
#define TGSF_RF_AD3P       0x04u  /* num3p  > 0                                               */
#define TGSF_RF_ADMID      0x08u  /* numMid > 0                                               */
#define TGSF_RF_DISCARDED  0x10u  /* numMid > 0 && -D: no regions (:1372-1373)                */

/* 32 bytes per read: the "trim coordinates" record */
typedef struct tgsf_read_result {
    uint64_t sum_q;        /* raw  Σ(qual[i]-qType) as uint64 (CalcAvgQuality sumQ, :1451-1458) */
    uint32_t flags;        /* TGSF_RF_*                                                        */
    uint32_t n_frags;      /* keepRegions.size() (:1960-1965); 0 if dropped                    */
    uint32_t frag_begin;   /* index of this read's first record in tgsf_batch_out.frags        */
    uint32_t trimmed;      /* bases this read added to DropInfo[10]                             */
    int32_t  reserved0;    /* 0 */
    int32_t  reserved1;    /* 0 */
} tgsf_read_result;

/* tgsf_fragment.flags */
#define TGSF_FF_PASS    0x01u /* survived the post-split quality gate (:1995-2001) => emitted  */
#define TGSF_FF_REPEAT  0x02u /* dropped by the repeat gate: GetKmerCount < -p (:1982-1989); no clean stats */

/* one keep-region {start,len} of adapterMap (:1404-1431), ascending within a read */
typedef struct tgsf_fragment {
    uint64_t sum_q;        /* clean Σ(qual-qType) over the fragment (:1994)                    */
    uint32_t read;         /* index of the read within the batch                               */
    int32_t  start;
    int32_t  len;
    uint32_t flags;        /* TGSF_FF_*                                                        */
} tgsf_fragment;

typedef struct tgsf_batch_out {
    tgsf_read_result* reads;        /* caller-allocated, n_reads entries                      */
    tgsf_fragment*    frags;        /* caller-allocated, frag_capacity entries                */
    uint32_t          frag_capacity;
    uint32_t          n_frags;      /* filled by the callee                                   */
} tgsf_batch_out;

/*
 * The additive tallies (T2 + T3 of SURVEY §8a) as one flat uint64 vector so
 * that a multi-GPU run can sum it with a single all-reduce.  Layout (u64 index):
 *   [0,17)                          DropInfo            (:1776; meaning :3214-3231)
 *   [17,273)                        rawDiffQualReadsBases   (:1777)
 *   [273,529)                       cleanDiffQualReadsBases (:1778)
 *   [529,531)                       rows used: raw bins, clean bins (max over reads of L/100+1)
 *   [531,533)                       rows used: raw 5'/3' tables, clean 5'/3' tables (max min(e,L))
 *   TGSF_CTR_END_TABLES + t*bc_len*5   t = 0..7, each [bc_len][5] :
 *        raw5pQual raw5pCounts raw3pQual raw3pCounts clean5pQual clean5pCounts clean3pQual clean3pCounts
 *   then 4 bin tables, each [n_bins][5] : rawBaseQual rawBaseCounts cleanBaseQual cleanBaseCounts
 * Column order of every [..][5] table is the reference's: A,T,G,C,all (:1462-1476).
 * The four "rows used" words are maxima, not sums: reduce them with MAX.
 */
#define TGSF_CTR_DROPINFO    0
#define TGSF_CTR_RAW_DIFFQ   17
#define TGSF_CTR_CLEAN_DIFFQ 273
#define TGSF_CTR_ROWS        529
#define TGSF_CTR_END_TABLES  533

enum { TGSF_T_RAW5P_QUAL = 0, TGSF_T_RAW5P_CNT, TGSF_T_RAW3P_QUAL, TGSF_T_RAW3P_CNT,
       TGSF_T_CLEAN5P_QUAL, TGSF_T_CLEAN5P_CNT, TGSF_T_CLEAN3P_QUAL, TGSF_T_CLEAN3P_CNT };
enum { TGSF_B_RAW_QUAL = 0, TGSF_B_RAW_CNT, TGSF_B_CLEAN_QUAL, TGSF_B_CLEAN_CNT };

static inline size_t tgsf_ctr_end_table(int t, int bc_len) {
    return (size_t)TGSF_CTR_END_TABLES + (size_t)t * (size_t)bc_len * 5u;
}
static inline size_t tgsf_ctr_bin_table(int b, int bc_len, uint32_t n_bins) {
    return (size_t)TGSF_CTR_END_TABLES + 8u * (size_t)bc_len * 5u + (size_t)b * (size_t)n_bins * 5u;
}
static inline size_t tgsf_ctr_len(int bc_len, uint32_t n_bins) {
    return tgsf_ctr_bin_table(4, bc_len, n_bins);
}
/* number of 100-bp rows a context created with max_read_len allocates */
static inline uint32_t tgsf_n_bins(uint32_t max_read_len) { return max_read_len / TGSF_BIN_WIDTH + 1u; }

typedef struct tgsf_ctx tgsf_ctx;

/* Library / ABI version (TGSF_ABI_VERSION). */
int tgsf_abi_version(void);

/*
 * What executes the kernels behind this library: "hip:gfx950" for the product (tgsfilter_amd/libtgsf.so).  The serial
 * emulation the tests build from the same kernel sources (tests/emul: test infrastructure for boxes without a GPU)
 * answers "emulation".  A host program that binds the library at run time checks this: the command line and the
 * Python binding refuse a library whose answer does not begin with "hip" -- a stray TGSF_LIB in the environment
 * must not turn a GPU run into a CPU run.  Static storage; never NULL.
 */
const char* tgsf_backend(void);

/*
 * Optional: bring the HIP runtime up on `device` and load the library's kernels (first use costs a few
 * tenths of a second).  Thread-safe; a host program can call it from a helper thread while it is still
 * reading its parameters, so that tgsf_create and the first batch do not pay for it.
 *
 * How host threads wait for the device inside tgsf_submit / tgsf_wait is the HOST PROGRAM's choice: with TGSF_SYNC=blocking in
 * the environment when the library brings a device up (here or in tgsf_create) the waits sleep (hipDeviceScheduleBlockingSync)
 * instead of spinning -- fewer CPU seconds at the same wall time; the tgsfilter command line sets it.  Leave it unset in a
 * process that shares the device with another user of the HIP runtime (PyTorch ...): the flag is the device's, not the library's.
 */
int tgsf_prepare_device(int device);

/*
 * Where a device sits in the host (nothing in the reference: it has no device): its PCI bus id ("0000:c1:00.0", NUL
 * terminated, into bus_id[0..len)) and the NUMA node of that PCI function (-1 where the platform does not say).
 * A multi-GPU caller binds each device's feeder threads -- and so the pinned staging buffers tgsf_create allocates from
 * them -- to the GPU's own node.  Does not initialise the device.
 */
int tgsf_device_location(int device, char* bus_id, int len, int* numa_node);

/*
 * Create a context on HIP device `device`.  Replaces the construction of
 * TGSFilterTask (:1757-1790: the accumulator rows) for one "worker" = one GPU.
 */
int tgsf_create(const tgsf_params* params, int device, tgsf_ctx** out_ctx);
void tgsf_destroy(tgsf_ctx* ctx);

/*
 * Filter one batch held in HOST memory: H2D, run the kernels, D2H of results.
 * The worker loop body (:1939-2061) for n_reads reads.  Counters accumulate inside the context.
 *
 * tgsf_submit_async only enqueues the work on the context's stream and returns; `in`, `out` and every
 * buffer they point to must stay valid and untouched until tgsf_wait(ctx) returns, which completes the
 * batch: out->n_frags is set, the fragment records are copied, asynchronous errors are reported.  One
 * batch may be pending per context; several contexts overlap each other's copies and kernels (copies are
 * only asynchronous from pinned host memory).  tgsf_submit = tgsf_submit_async + tgsf_wait.
 */
int tgsf_submit(tgsf_ctx* ctx, const tgsf_batch_in* in, tgsf_batch_out* out);
int tgsf_submit_async(tgsf_ctx* ctx, const tgsf_batch_in* in, tgsf_batch_out* out);

/*
 * Same, but every pointer in `in` and `out` is a DEVICE pointer already
 * resident in HBM and the work is enqueued on `hip_stream` (a hipStream_t, or
 * NULL for the context's own stream) without synchronising.
 * out->n_frags is not filled; the fragment count is written to
 * *d_n_frags (device uint32) if that pointer is non-NULL.
 *
 * Several batches may be enqueued one after the other (up to TGSF_MAX_ENQUEUED; one more makes the call wait for
 * the whole device and close the books of the earlier ones -- which fails with TGSF_E_INVALID if one of them has to be
 * run again, see below: only tgsf_wait does that, so call it at least every TGSF_MAX_ENQUEUED batches).  A batch's results and its share of the tallies are FINAL ONLY WHEN tgsf_wait(ctx) HAS RETURNED,
 * and every buffer `in` and `out` point to must stay valid and untouched until then: a batch whose middle-adapter
 * candidates outgrow the context's pool (a read whose best alignment is tied column after column: every tied column is
 * a location, include/edlib.cpp:660-672) is left alone by its first run and run again, from its inputs, inside
 * tgsf_wait.  A caller that only synchronises its stream and finds TGSF_NFRAGS_NOT_FINAL in *d_n_frags is looking at
 * such a batch: its records are not written yet.
 */
#define TGSF_MAX_ENQUEUED 64
#define TGSF_NFRAGS_NOT_FINAL 0xFFFFFFFFu
int tgsf_submit_device(tgsf_ctx* ctx, const tgsf_batch_in* in, tgsf_batch_out* out,
                       uint32_t* d_n_frags, void* hip_stream);

/* Block until everything submitted on this context has finished; run again the batches whose candidate pool
 * overflowed (see tgsf_submit_device); complete a pending tgsf_submit_async batch; report async errors. */
int tgsf_wait(tgsf_ctx* ctx);

/* Number of uint64 words tgsf_counters() writes, and the table geometry. */
int tgsf_counters_len(tgsf_ctx* ctx, uint64_t* n_words, int32_t* bc_len, uint32_t* n_bins);
/* Copy the tallies to host memory (synchronises the context). Replaces the
 * per-thread merge of src/TGSFilter.cpp:3208-3213 / :2673-2725 / :2586-2597. */
int tgsf_counters(tgsf_ctx* ctx, uint64_t* dst, uint64_t n_words);
/* Same, but of the four bin tables ([n_bins][5] each, n_bins sized by max_read_len at tgsf_create -- possibly far
 * more rows than any read of the run needed) only the rows in use are copied: rows[0] rows of the two raw tables,
 * rows[1] rows of the two clean ones (the values of dst[TGSF_CTR_ROWS], dst[TGSF_CTR_ROWS+1]).  Words of dst
 * beyond those rows are left as they were (zero them beforehand).  rows may be NULL. */
int tgsf_counters_used(tgsf_ctx* ctx, uint64_t* dst, uint64_t n_words, uint64_t rows[2]);
/* Add the tallies of `src` to those of `dst` (same device, same bc_len and max_read_len), in HBM; `src` keeps its
 * own.  Several contexts of one process (one per batch in flight) become one vector this way -- e.g. before the
 * job's all-reduce (include/tgsf_rccl.h).  Waits for both contexts. */
int tgsf_counters_merge(tgsf_ctx* dst, tgsf_ctx* src);
/* Device address of the same vector (for an in-place RCCL all-reduce). */
int tgsf_counters_device(tgsf_ctx* ctx, void** d_ptr, uint64_t* n_words);
/* Zero the tallies (start of a new run). */
int tgsf_reset_counters(tgsf_ctx* ctx);

/*
 * Stage timing (HIP events on the submit stream).  Enable, run batches, then read
 * the accumulated milliseconds per pipeline stage since the last reset.
 */
#define TGSF_N_STAGES 12
int tgsf_profile(tgsf_ctx* ctx, int enable);
int tgsf_stage_times(tgsf_ctx* ctx, float ms[TGSF_N_STAGES], uint32_t* n_batches);
const char* tgsf_stage_name(int stage);

/*
 * edlib-compatible single alignment on the device (S2), used by the parity tests
 * and by the parameter pre-pass (adapterSearch, src/TGSFilter.cpp:1163): for each
 * of n problems (adapter a[i], window = seq[win_off[i] .. +win_len[i])) with
 * threshold k[i], HW mode, PATH task, writes
 *   res[i*4+0] editDistance (-1 if > k), +1 numLocations, +2 alignmentLength,
 *   +3 startLocations[0] ; ends[i*2+0] endLocations[0], ends[i*2+1] endLocations[last]
 * All pointers are HOST pointers.
 */
int tgsf_align_windows(tgsf_ctx* ctx, const uint8_t* seq, uint64_t n_bytes,
                       const uint64_t* win_off, const uint32_t* win_len,
                       const uint8_t* adapter_id, const int32_t* k, uint32_t n,
                       int32_t* res, int32_t* ends);

const char* tgsf_last_error(tgsf_ctx* ctx);   /* ctx may be NULL: last create error */

#ifdef __cplusplus
}
#endif
#endif /* TGSF_H */
