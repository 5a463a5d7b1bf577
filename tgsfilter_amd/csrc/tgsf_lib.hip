// tgsf_lib.hip -- libtgsf.so: the C ABI of include/tgsf.h on top of the HIP kernels.
//
// Built two ways from this one source:
//   hipcc --offload-arch=gfx950            -> tgsfilter_amd/libtgsf.so   (the product)
//   g++ -x c++ -DTGSF_EMUL                 -> tests/emul/libtgsf_emul.so (serial lane-by-lane
//                                             emulation of the same kernels; test infrastructure)
// The product has no CPU path: tgsf_create fails when no HIP device is usable.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#if defined(TGSF_EMUL)
#include "tgsf_kernels.h"
namespace tgsf_emul { thread_local Dim3 threadIdx, blockIdx, blockDim, gridDim; }
typedef void* rt_stream;
typedef int rt_error;
#else
#include <hip/hip_runtime.h>
#include "tgsf_kernels.h"
typedef hipStream_t rt_stream;
#endif

using namespace tgsf;

// ---------------------------------------------------------------------------
// runtime layer
// ---------------------------------------------------------------------------
static thread_local std::string g_create_error;

// Test and diagnostic settings of the library (TGSF_POOL_CAP, TGSF_CLEAN_TABLES, TGSF_MID_FILTER, TGSF_FLAT_*, TGSF_TRACE_* ...:
// each is explained where it is read) force rare paths and strategies in the tests.  They are read only when
// TGSF_DEBUG_KNOBS=1 is in the environment -- the test suite sets it --: a stray variable in a user's environment never
// changes the path a run takes.  The library has no other setting: everything else comes through tgsf_params.
static const char* knob(const char* name)
{
    static const bool on = [] { const char* e = getenv("TGSF_DEBUG_KNOBS"); return e && e[0] == '1'; }();
    return on ? getenv(name) : nullptr;
}

struct tgsf_ctx {
    tgsf_params params;
    std::vector<std::string> adapters;
    DevParams P;
    DevBatch B;               // internal buffers (template for each batch)
    int device;
    rt_stream stream;
    rt_stream last_stream;     // stream of the most recent submit
    bool own_stream;
    std::string error;
    // capacities
    uint64_t cap_bases;
    uint32_t cap_reads, max_read_len, n_bins;
    static constexpr unsigned endtab_grid = 512;   // k_end_tables: LDS-atomic bound, 40 KB of LDS per block: two blocks per CU (128: 0.36 ms, 512: 0.19 ms)
    static constexpr unsigned stats_grid = 768;    // k_stats: 3 blocks (12 waves) per CU on 256 CUs, LDS-limited
    bool flat_scan = true;                    // first middle scan of a batch by k_mid_flat (TGSF_MID_FLAT=0: k_mid_scan1, as after a pool overflow)
    int suffix_filter = 2;                    // TGSF_MID_FILTER: test stride of k_mid_flat's 32-row filter for adapters of 33..64 bp at k <= kSuffixMaxK (0: off, 1, 2)
    bool no_hot32 = false;                    // TGSF_NO_HOT32=1: adapters <= 32 bp take the 64-bit column too (A/B, tests)
    uint64_t ctr_words;
    int scratch_cols;
    // internal input / output staging for tgsf_submit
    uint8_t *d_seq, *d_qual;
    uint64_t* d_off;
    uint64_t* d_qoff;
    uint32_t* d_lenin;
    tgsf_read_result* d_out_reads;
    tgsf_fragment* d_out_frags;
    uint32_t* d_out_nfrags;
    std::vector<void*> allocs;
    // profiling
    bool profile;
    float stage_ms[TGSF_N_STAGES];
    uint32_t prof_batches;
#if !defined(TGSF_EMUL)
    // ring of event sets: one set per profiled batch, harvested at tgsf_wait (no per-batch sync)
    static constexpr int kProfRing = 64;
    hipEvent_t ev[kProfRing][TGSF_N_STAGES + 1];
    hipEvent_t ev_aux[kProfRing][3];      // around the two kernels that run on the auxiliary stream
    int prof_pending;
    hipStream_t aux;                      // end-window / end-table kernels overlap the middle scan here
    hipEvent_t ev_fork, ev_join;
    uint32_t* h_pinned = nullptr;
    static constexpr size_t kStageBytes = 32u << 20;
    uint8_t* stage[2] = {nullptr, nullptr};   // pinned staging for host batches in pageable memory (text_h2d)
    hipEvent_t stage_ev[2] = {nullptr, nullptr};
#endif
    uint32_t h_words_[8];      // backing store of h_status / pend_nf when no pinned page is available (emulation)
    uint32_t* h_status;        // [4] device status words as last fetched  } one pinned allocation: the small D2H copies
    uint32_t* pend_nf_p;       // fragment count of the pending batch       } into it are truly asynchronous
    // batch enqueued by tgsf_submit_async, completed by tgsf_wait
    tgsf_batch_out* pend_out = nullptr;
    // the batches enqueued since the last tgsf_wait, as the pipeline saw them (device pointers): a batch whose candidate
    // pool overflowed in the middle scan (its word of ovf_ring) was left alone by every kernel behind the scan, and
    // tgsf_wait runs it again from its inputs -- which the caller keeps untouched until then -- with the raw-side
    // tallies of its first run not added twice (DevBatch::replay) and a pool grown to fit
    struct Pending { tgsf_batch_in in; tgsf_read_result* reads; tgsf_fragment* frags; uint32_t fcap; uint32_t* nfrags; rt_stream st; };
    Pending pending[TGSF_MAX_ENQUEUED];
    uint32_t n_pending = 0;
    uint32_t* ovf_ring = nullptr;          // [TGSF_MAX_ENQUEUED] device words, one per enqueued batch
    uint32_t* bp_ring = nullptr;           // [TGSF_MAX_ENQUEUED] ... and whether that batch speculated (DevBatch::bp_used)
    uint32_t h_ovf[TGSF_MAX_ENQUEUED];
    uint32_t pool_regrown = 0;            // times the pool had to grow (tests look at it through tgsf_last_error's sibling below)

};

static int fail(tgsf_ctx* c, int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->error = buf; else g_create_error = buf;
    return code;
}

#if defined(TGSF_EMUL)
#include "tgsf_emul_rt.h"                  // the runtime layer of the serial emulation (test infrastructure)
#else
static int rt_malloc(void** p, size_t n) { return (int)hipMalloc(p, n ? n : 1); }
static void rt_free(void* p) { (void)hipFree(p); }
static int rt_memset(void* p, int v, size_t n, rt_stream s) { return (int)hipMemsetAsync(p, v, n, s); }
static int rt_h2d(void* d, const void* s, size_t n, rt_stream st) { return (int)hipMemcpyAsync(d, s, n, hipMemcpyHostToDevice, st); }
static int rt_d2h(void* d, const void* s, size_t n, rt_stream st) { return (int)hipMemcpyAsync(d, s, n, hipMemcpyDeviceToHost, st); }
static int rt_sync(rt_stream s) { return (int)hipStreamSynchronize(s); }
static const char* rt_errstr(int e) { return hipGetErrorString((hipError_t)e); }
#define TGSF_LAUNCH(kernel, grid, block, stream, ...) hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), 0, (stream), __VA_ARGS__)
#define TGSF_LAUNCH_COOP TGSF_LAUNCH
// with `lds` bytes of dynamic LDS on top of the kernel's own (nothing uses them: they bound the workgroups a CU holds)
#define TGSF_LAUNCH_LDS(kernel, grid, block, lds, stream, ...) hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), (lds), (stream), __VA_ARGS__)
static unsigned grid_cap(unsigned g) { return g; }
#endif

template <class T>
static int dev_alloc(tgsf_ctx* c, T** p, size_t count)
{
    void* v = nullptr;
    int e = rt_malloc(&v, count * sizeof(T) + 64);
    if (e) return fail(c, TGSF_E_HIP, "device allocation of %zu bytes failed: %s", count * sizeof(T), rt_errstr(e));
    c->allocs.push_back(v);
    *p = (T*)v;
    return 0;
}

// ---------------------------------------------------------------------------
// adapter tables
// ---------------------------------------------------------------------------
static int build_tables(tgsf_ctx* c)
{
    const tgsf_params& p = c->params;
    DevParams& P = c->P;
    const int A = p.n_adapters;
    std::vector<uint8_t> ad((size_t)kMaxAdapters * kMaxQ, 0);
    const size_t per_ad = (size_t)256 * kPeqW;              // standard-layout Peq: [symbol][word]
    std::vector<uint64_t> fwd((size_t)kMaxAdapters * per_ad, 0), rev((size_t)kMaxAdapters * per_ad, 0),
        top((size_t)kMaxAdapters * 256, 0);
    P.max_nw = 1;
    P.min_Q = 1 << 30;
    for (int a = 0; a < A; a++) {
        const std::string& s = c->adapters[a];
        const int Q = (int)s.size();
        P.Q[a] = Q;
        if (Q > 64 && P.max_nw < 2) P.max_nw = 2;
        if (Q > 128 && P.max_nw < 4) P.max_nw = 4;
        if (Q > 256) P.max_nw = kWideNW;
        if (Q < P.min_Q) P.min_Q = Q;
        int km = Q - p.mid_match_len + 1;                 // src/TGSFilter.cpp:1233
        int ke = Q - p.end_match_len + 1;                 // :1271
        // edlib clamps k to Q (include/edlib.cpp:565-567); here it is clamped to Q - 1.  A best value of exactly Q
        // (reachable only with -m 1 / -M 1) means no character of the adapter matches anywhere in the window: edlib
        // then reports every column (and the location -1, :232-244, :670) with a path of Q insertions, i.e. mlen = 0,
        // which no gate passes (:1246, :1283) -- the same outcome as reporting nothing.  Values below Q are unaffected.
        P.k_mid[a] = km > Q - 1 ? Q - 1 : km;             // (negative: see k_end_windows)
        P.k_end[a] = ke > Q - 1 ? Q - 1 : ke;
        P.w5[a] = p.end_len + (int)((float)Q / p.end_sim);   // :1267 int(qLen / endSim), float division
        // the similarity gates are monotone in mlen: evaluate the reference's float predicates here, once
        P.need_end[a] = P.need_mid[a] = Q + 1;                // Q+1: cannot pass (mlen <= Q)
        for (int m = Q; m >= 0; m--) {
            if (m >= p.end_match_len && (float)m / (float)Q >= p.end_sim) P.need_end[a] = m;
            if (m >= p.mid_match_len && (float)m / (float)Q >= p.mid_sim) P.need_mid[a] = m;
        }
        memcpy(&ad[(size_t)a * kMaxQ], s.data(), (size_t)Q);
        for (int r = 0; r < Q && r < 64 * kPeqW; r++) {       // (adapters beyond 256 bp only use the wide tables below)
            uint8_t cf = (uint8_t)s[r], cr = (uint8_t)s[Q - 1 - r];
            fwd[(size_t)a * per_ad + (size_t)cf * kPeqW + (r >> 6)] |= 1ull << (r & 63);
            rev[(size_t)a * per_ad + (size_t)cr * kPeqW + (r >> 6)] |= 1ull << (r & 63);
        }
        if (Q <= 64) {
            const int sh = 64 - Q;
            const uint64_t pad = sh ? ((1ull << sh) - 1ull) : 0ull;    // wildcard rows below the adapter
            for (int sym = 0; sym < 256; sym++)
                top[(size_t)a * 256 + sym] = (fwd[(size_t)a * per_ad + (size_t)sym * kPeqW] << sh) | pad;
        }
    }
    if (A == 0) P.min_Q = 1 << 30;
    uint8_t* d_ad; uint64_t *d_f, *d_r, *d_t;
    int e;
    if ((e = dev_alloc(c, &d_ad, ad.size()))) return e;
    if ((e = dev_alloc(c, &d_f, fwd.size()))) return e;
    if ((e = dev_alloc(c, &d_r, rev.size()))) return e;
    if ((e = dev_alloc(c, &d_t, top.size()))) return e;
    rt_h2d(d_ad, ad.data(), ad.size(), c->stream);
    rt_h2d(d_f, fwd.data(), fwd.size() * 8, c->stream);
    rt_h2d(d_r, rev.data(), rev.size() * 8, c->stream);
    rt_h2d(d_t, top.data(), top.size() * 8, c->stream);
    rt_sync(c->stream);
    P.adapter = d_ad; P.peq_fwd = d_f; P.peq_rev = d_r; P.peq_top = d_t;
    P.peq_fwd_w = P.peq_rev_w = nullptr;
    if (P.max_nw > 4) {                                   // an adapter beyond 256 bp: kWideNW-word tables for every adapter
        const size_t per_w = (size_t)256 * kWideNW;
        std::vector<uint64_t> fw((size_t)A * per_w, 0), rw((size_t)A * per_w, 0);
        for (int a = 0; a < A; a++) {
            const std::string& s = c->adapters[a];
            const int Q = (int)s.size();
            for (int r = 0; r < Q; r++) {
                const uint8_t cf = (uint8_t)s[r], cr = (uint8_t)s[Q - 1 - r];
                fw[(size_t)a * per_w + (size_t)cf * kWideNW + (r >> 6)] |= 1ull << (r & 63);
                rw[(size_t)a * per_w + (size_t)cr * kWideNW + (r >> 6)] |= 1ull << (r & 63);
            }
        }
        uint64_t *d_fw, *d_rw;
        if ((e = dev_alloc(c, &d_fw, fw.size()))) return e;
        if ((e = dev_alloc(c, &d_rw, rw.size()))) return e;
        rt_h2d(d_fw, fw.data(), fw.size() * 8, c->stream);
        rt_h2d(d_rw, rw.data(), rw.size() * 8, c->stream);
        rt_sync(c->stream);
        P.peq_fwd_w = d_fw; P.peq_rev_w = d_rw;
    }
    return 0;
}

// ---------------------------------------------------------------------------
// create / destroy
// ---------------------------------------------------------------------------
extern "C" int tgsf_abi_version(void) { return TGSF_ABI_VERSION; }
extern "C" const char* tgsf_backend(void) { return kTgsfEmul ? "emulation" : "hip:gfx950"; }

#if !defined(TGSF_EMUL)
__global__ void k_noop(int* p) { if (p) *p = 0; }
#endif
#if !defined(TGSF_EMUL)
// How a host thread waits for the device (hipStreamSynchronize / hipEventSynchronize inside tgsf_submit and tgsf_wait).  The
// runtime's default spins where it sees spare CPUs.  TGSF_SYNC=blocking makes the waits SLEEP: the library then sets the
// device's scheduling flag before it brings the device up.  It is the HOST PROGRAM's choice, not the library's: the command
// line asks for it (host/main.cpp; measured on the MI355X host, one of C2's files, profiles/r06_cpu_blocking_sync.txt: the
// feeders' CPU time 7.6 -> 5.8 s in one process and 21 -> 7-10 s in three rank processes sharing the GPU, wall time
// unchanged -- what a job of N ranks under a CPU quota needs); a process that shares the device with another user of the
// runtime must not -- with PyTorch active in the same process (bench.py's kernel path) changing the flag hung the next wait.
static void set_wait_mode(int device)
{
    const char* e = getenv("TGSF_SYNC");
    if (!e || strcmp(e, "blocking") != 0) return;
    if (hipSetDevice(device) == hipSuccess) { (void)hipSetDeviceFlags(hipDeviceScheduleBlockingSync); (void)hipGetLastError(); }
}
#endif
extern "C" int tgsf_prepare_device(int device)
{
#if !defined(TGSF_EMUL)
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return TGSF_E_NO_DEVICE;
    set_wait_mode(device);
    if (hipSetDevice(device) != hipSuccess) return TGSF_E_HIP;
    hipLaunchKernelGGL(k_noop, dim3(1), dim3(64), 0, 0, (int*)nullptr);      // loads the code object
    if (hipDeviceSynchronize() != hipSuccess) return TGSF_E_HIP;
    // every kernel is resolved on its first launch: run a miniature batch (two 50-bp adapters, the usual
    // configuration) through the whole pipeline once
    {
        static const char a0[] = "GTTTTCGCATTTATCGTGAAACGCTTTCGCGTTTTTCGTGCGCCGCTTCA";
        static const char a1[] = "TGAAGCGGCGCACGAAAAACGCGAAAGCGTTTCACGATAAATGCGAAAAC";
        tgsf_params p;
        memset(&p, 0, sizeof p);
        p.struct_size = sizeof p;
        p.min_len = 100; p.max_len = 2147483647; p.min_q = 0; p.max_q = 255; p.bc_len = 150;
        p.end_len = 150; p.end_match_len = 4; p.mid_match_len = 35; p.extra_len = 50;
        p.end_sim = 0.75f; p.mid_sim = 0.9f; p.filter = 1; p.qtype = 33; p.kmer = 11;
        p.n_adapters = 2; p.adapters[0] = a0; p.adapters[1] = a1; p.adapter_len[0] = p.adapter_len[1] = 50;
        p.max_batch_bases = 1 << 16; p.max_batch_reads = 16; p.max_read_len = 1 << 14;
        tgsf_ctx* c = nullptr;
        if (tgsf_create(&p, device, &c) == TGSF_OK) {
            std::vector<uint8_t> seq(8192), qual(8192, (uint8_t)'5');
            for (size_t i = 0; i < seq.size(); i++) seq[i] = "ACGT"[(i * 2654435761u >> 13) & 3];
            memcpy(&seq[4096 + 1000], a0, 50);
            const uint64_t off[2] = {0, 4096};
            const uint32_t len[2] = {3000, 4000};
            tgsf_read_result res[2];
            tgsf_fragment fr[64];
            tgsf_batch_in in;
            memset(&in, 0, sizeof in);
            in.seq = seq.data(); in.qual = qual.data(); in.offsets = off; in.lengths = len; in.n_reads = 2; in.n_bytes = seq.size();
            tgsf_batch_out out{res, fr, 64, 0};
            (void)tgsf_submit(c, &in, &out);
            tgsf_destroy(c);
        }
    }
#else
    (void)device;
#endif
    return TGSF_OK;
}

extern "C" int tgsf_device_location(int device, char* bus_id, int len, int* numa_node)
{
    if (numa_node) *numa_node = -1;
    if (bus_id && len > 0) bus_id[0] = 0;
#if !defined(TGSF_EMUL)
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return TGSF_E_NO_DEVICE;
    char id[64] = {0};
    if (hipDeviceGetPCIBusId(id, (int)sizeof id, device) != hipSuccess) return TGSF_E_HIP;
    for (char* p = id; *p; p++) if (*p >= 'A' && *p <= 'F') *p = (char)(*p - 'A' + 'a');       // sysfs spells it in lower case
    if (bus_id && len > 0) { strncpy(bus_id, id, (size_t)len - 1); bus_id[len - 1] = 0; }
    if (numa_node) {
        const std::string path = std::string("/sys/bus/pci/devices/") + id + "/numa_node";
        if (FILE* f = fopen(path.c_str(), "r")) {
            int v = -1;
            if (fscanf(f, "%d", &v) == 1) *numa_node = v;
            fclose(f);
        }
    }
    return TGSF_OK;
#else
    // (emulation: a made-up place per device -- "emul:<n>", and the n-th entry of TGSF_EMUL_NUMA as its node -- so that the host
    // side's one-rank-per-GPU paths, the bus-id gather and the NUMA binding, can be tested on a box without a GPU)
    if (bus_id && len > 0) snprintf(bus_id, (size_t)len, "emul:%d", device);
    if (numa_node)
        if (const char* e = knob("TGSF_EMUL_NUMA")) {
            int k = 0;
            for (const char* c = e; *c; k++) {
                if (k == device) { *numa_node = atoi(c); break; }
                while (*c && *c != ',') c++;
                if (*c == ',') c++;
            }
        }
    return TGSF_OK;
#endif
}

extern "C" const char* tgsf_last_error(tgsf_ctx* ctx) { return ctx ? ctx->error.c_str() : g_create_error.c_str(); }

extern "C" void tgsf_destroy(tgsf_ctx* c)
{
    if (!c) return;
    if (c->B.bp_allowed && c->B.bp_state && knob("TGSF_TRACE_BP")) {     // how often the clean tables came as a by-product
        uint32_t w[4] = {0, 0, 0, 0};
        uint64_t pl[4] = {0, 0, 0, 0};
        uint32_t sg[4] = {0, 0, 0, 0};
        if (!rt_d2h(w, c->B.bp_state, sizeof w, c->stream) && !rt_d2h(pl, c->B.plan, sizeof pl, c->stream) && !rt_d2h(sg, c->B.seg_info, sizeof sg, c->stream) && !rt_sync(c->stream))
            fprintf(stderr, "tgsf: clean tables as a by-product of the raw pass: %u batches speculated; the next would%s; the last batch: %llu bases a direct clean pass scans, "
                            "%llu bases its own did, %u work items in that pass\n", w[1], w[0] ? "" : " not", (unsigned long long)pl[2], (unsigned long long)pl[3], sg[2]);
    }
#if !defined(TGSF_EMUL)
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    for (int k = 0; k < tgsf_ctx::kProfRing; k++) {
        for (int i = 0; i <= TGSF_N_STAGES; i++) if (c->ev[k][i]) (void)hipEventDestroy(c->ev[k][i]);
        for (int i = 0; i < 3; i++) if (c->ev_aux[k][i]) (void)hipEventDestroy(c->ev_aux[k][i]);
    }
    if (c->h_pinned) (void)hipHostFree(c->h_pinned);
    for (int i = 0; i < 2; i++) {
        if (c->stage[i]) (void)hipHostFree(c->stage[i]);
        if (c->stage_ev[i]) (void)hipEventDestroy(c->stage_ev[i]);
    }
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->aux) (void)hipStreamDestroy(c->aux);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
#endif
    for (void* p : c->allocs) rt_free(p);
    delete c;
}

extern "C" int tgsf_create(const tgsf_params* p, int device, tgsf_ctx** out)
{
    if (!p || !out) return fail(nullptr, TGSF_E_INVALID, "null argument");
    *out = nullptr;
    if (p->struct_size != sizeof(tgsf_params))
        return fail(nullptr, TGSF_E_INVALID, "tgsf_params.struct_size %u != %zu (ABI mismatch)", p->struct_size, sizeof(tgsf_params));
    if (p->n_adapters < 0 || p->n_adapters > TGSF_MAX_ADAPTERS)
        return fail(nullptr, TGSF_E_INVALID, "n_adapters %d outside [0,%d]", p->n_adapters, TGSF_MAX_ADAPTERS);
    if (p->min_repeat > 0 && (p->kmer < 1 || p->kmer > 32))
        return fail(nullptr, TGSF_E_UNSUPPORTED, "-k %d: the repeat gate supports k-mer sizes 1..32 (the reference's k-mers are 64 bits)", p->kmer);
    if (p->qtype != 33 && p->qtype != 64) return fail(nullptr, TGSF_E_INVALID, "qtype must be 33 or 64");
    if (p->bc_len < 0 || p->bc_len > kMaxBcLenTotal) return fail(nullptr, TGSF_E_INVALID, "bc_len (-e) outside [0,%d]", kMaxBcLenTotal);
    if (p->filter && p->n_adapters > 0) {
        if (!(p->end_sim > 0.f) || !(p->mid_sim > 0.f)) return fail(nullptr, TGSF_E_INVALID, "similarities must be > 0");
        if (p->end_len < 0 || p->extra_len < 0) return fail(nullptr, TGSF_E_INVALID, "end_len / extra_len must be >= 0");
        if (p->end_match_len < 1 || p->mid_match_len < 1)
            return fail(nullptr, TGSF_E_INVALID, "match lengths must be >= 1");
    }
    for (int a = 0; a < p->n_adapters; a++) {
        if (!p->adapters[a] || p->adapter_len[a] < 1 || p->adapter_len[a] > TGSF_MAX_ADAPTER_LEN)
            return fail(nullptr, TGSF_E_UNSUPPORTED, "adapter %d: length %d outside [1,%d]", a, p->adapter_len[a], TGSF_MAX_ADAPTER_LEN);
    }
    if (!p->max_batch_bases || !p->max_batch_reads || !p->max_read_len)
        return fail(nullptr, TGSF_E_INVALID, "max_batch_bases / max_batch_reads / max_read_len must be set");
    if (p->max_read_len > (1u << 28)) return fail(nullptr, TGSF_E_INVALID, "max_read_len above 2^28");

    tgsf_ctx* c = new tgsf_ctx();
    c->params = *p;
    c->device = device;
    c->profile = false;
    c->prof_batches = 0;
    memset(c->stage_ms, 0, sizeof c->stage_ms);
    memset(c->h_words_, 0, sizeof c->h_words_);
    c->h_status = c->h_words_;
    c->pend_nf_p = c->h_words_ + 4;
    for (int a = 0; a < p->n_adapters; a++) {
        c->adapters.emplace_back(p->adapters[a], (size_t)p->adapter_len[a]);
        c->params.adapters[a] = c->adapters.back().data();
    }
    for (int a = 0; a < p->n_adapters; a++) c->params.adapters[a] = c->adapters[a].data();
#if !defined(TGSF_EMUL)
    memset(c->ev, 0, sizeof c->ev);
    memset(c->ev_aux, 0, sizeof c->ev_aux);
    c->aux = nullptr; c->ev_fork = c->ev_join = nullptr;
    int ndev = 0;
    hipError_t he = hipGetDeviceCount(&ndev);
    if (he != hipSuccess || ndev <= 0) {
        delete c;
        return fail(nullptr, TGSF_E_NO_DEVICE, "no HIP device available (%s); libtgsf has no CPU fallback", hipGetErrorString(he));
    }
    if (device < 0 || device >= ndev) { delete c; return fail(nullptr, TGSF_E_NO_DEVICE, "device %d out of range (%d devices)", device, ndev); }
    set_wait_mode(device);
    if ((he = hipSetDevice(device)) != hipSuccess) { delete c; return fail(nullptr, TGSF_E_HIP, "hipSetDevice: %s", hipGetErrorString(he)); }
    if ((he = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess) {
        delete c; return fail(nullptr, TGSF_E_HIP, "hipStreamCreate: %s", hipGetErrorString(he));
    }
    c->own_stream = true;
    c->prof_pending = 0;
    {
        uint32_t* pw = nullptr;
        if (hipHostMalloc((void**)&pw, 64, hipHostMallocDefault) == hipSuccess && pw) {
            memset(pw, 0, 64);
            c->h_pinned = pw; c->h_status = pw; c->pend_nf_p = pw + 4;
        }
    }
    {
        // the auxiliary stream's short kernels (end windows, end tables) run beside the middle scan, whose workgroups fill
        // every CU for a millisecond each: with a priority above the scan's they get the slots that free up first
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);               // hi = numerically lowest = greatest priority
        he = hipStreamCreateWithPriority(&c->aux, hipStreamNonBlocking, hi);
    }
    if (he != hipSuccess ||
        (he = hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming)) != hipSuccess ||
        (he = hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming)) != hipSuccess) {
        tgsf_destroy(c); return fail(nullptr, TGSF_E_HIP, "auxiliary stream: %s", hipGetErrorString(he));
    }
#else
    c->stream = nullptr; c->own_stream = false;
#endif
    c->last_stream = c->stream;
#if !defined(TGSF_EMUL)
    if (p->max_batch_bases > (8u << 20)) {      // contexts for small batches (tests, the pre-pass) copy directly
        for (int i = 0; i < 2; i++)
            if (hipHostMalloc((void**)&c->stage[i], tgsf_ctx::kStageBytes, hipHostMallocDefault) != hipSuccess ||
                hipEventCreateWithFlags(&c->stage_ev[i], hipEventDisableTiming) != hipSuccess) {
                tgsf_destroy(c); return fail(nullptr, TGSF_E_HIP, "pinned staging buffers: allocation failed");
            }
    }
#endif
    DevParams& P = c->P;
    memset(&P, 0, sizeof P);
    P.min_len = p->min_len; P.max_len = p->max_len; P.min_q = p->min_q; P.max_q = p->max_q;
    P.bc_len = p->bc_len; P.head_trim = p->head_trim; P.tail_trim = p->tail_trim; P.end_len = p->end_len;
    P.end_match_len = p->end_match_len; P.mid_match_len = p->mid_match_len; P.extra_len = p->extra_len;
    P.end_sim = p->end_sim; P.mid_sim = p->mid_sim; P.discard = p->discard; P.filter = p->filter;
    P.only_qc = p->only_qc; P.qtype = p->qtype; P.n_adapters = p->n_adapters;
    P.no_qual = p->no_qual ? 1 : 0;
    P.min_repeat = p->min_repeat; P.kmer = p->kmer;
    c->cap_bases = p->max_batch_bases;
    c->cap_reads = p->max_batch_reads;
    c->max_read_len = p->max_read_len;
    c->n_bins = tgsf_n_bins(p->max_read_len);
    P.n_bins = c->n_bins;
    c->ctr_words = tgsf_ctr_len(p->bc_len, c->n_bins);
    P.seg_cols = kSegCols;
    if (const char* e = knob("TGSF_NO_HOT32")) c->no_hot32 = atoi(e) > 0;
    if (const char* e = knob("TGSF_SEG_COLS")) { int v = atoi(e); if (v >= 256 && v <= 65536) P.seg_cols = v & ~15; }   // tuning knob
    if (const char* e = knob("TGSF_MID_FLAT")) c->flat_scan = atoi(e) > 0;
    if (const char* e = knob("TGSF_MID_FILTER")) { int v = atoi(e); if (v >= 0 && v <= 2) c->suffix_filter = v; }

    int e = build_tables(c);
    DevBatch& B = c->B;
    memset(&B, 0, sizeof B);
    // k_mid_flat's schedule (flat_schedule): 7/8 of a batch's chunks in stretches of 256 (4 096 columns: the warm-up is
    // 2 % of that), the rest in stretches halving down to 16 chunks (the launch ends everywhere within 256 columns)
    B.flat_pmax = 256; B.flat_pmin = 16; B.flat_f0 = 224;
    if (const char* e = knob("TGSF_FLAT_PMIN")) { int v = atoi(e); if (v >= 1 && v <= (1 << 20)) B.flat_pmin = (uint32_t)v; }
    if (const char* e = knob("TGSF_FLAT_PMAX")) { int v = atoi(e); if (v >= 1 && v <= (1 << 20)) B.flat_pmax = (uint32_t)v; }
    if (const char* e = knob("TGSF_FLAT_F0")) { int v = atoi(e); if (v >= 0 && v <= 256) B.flat_f0 = (uint32_t)v; }
    B.flat_pmin = 1u << flat_log2(B.flat_pmin); B.flat_pmax = 1u << flat_log2(B.flat_pmax);
    if (B.flat_pmax < B.flat_pmin) B.flat_pmax = B.flat_pmin;
    const size_t n = c->cap_reads;
    const int A = p->n_adapters > 0 ? p->n_adapters : 1;
    // every read start may be padded to 16 bytes by the caller
    const uint64_t cap_bytes = c->cap_bases + 16ull * n + 64;
    const int min_len = p->min_len > 0 ? p->min_len : 1;
    uint64_t fcap64 = p->filter ? (c->cap_bases / (uint64_t)min_len + 16) : (uint64_t)n + 16;
    if (fcap64 > c->cap_bases + 16) fcap64 = c->cap_bases + 16;
    if (fcap64 < n + 16 && !p->filter) fcap64 = n + 16;
    B.fcap = (uint32_t)std::min<uint64_t>(fcap64, 0x7FFFFFF0ull);
    B.max_tiles = (c->max_read_len + kTileBases - 1) / kTileBases;
    // candidates are rare at sensible thresholds (1e-6 per column), and a lane only hands over the columns tying its best
    // value (4 at a time): the pool stays small for any threshold; some room per read and adapter on top
    uint64_t pool = c->cap_bases / 64 + 65536 + (uint64_t)n * 16u * (uint64_t)std::max(p->n_adapters, 1);
    B.pool_cap = (uint32_t)std::min<uint64_t>(pool, 1ull << 28);
    if (const char* e = knob("TGSF_POOL_CAP")) { const long long v = atoll(e); if (v >= 1 && v < (1ll << 28)) B.pool_cap = (uint32_t)v; }   // test knob: force the overflow path
    const size_t nitems = n + (size_t)B.fcap;         // clean pass: fragments, and reads to take back out
    if (!e) e = dev_alloc(c, &c->d_seq, cap_bytes);
    if (!e) e = dev_alloc(c, &c->d_qual, cap_bytes);
    if (!e) e = dev_alloc(c, &c->d_off, n + 1);
    if (!e) e = dev_alloc(c, &c->d_qoff, n + 1);
    if (!e) e = dev_alloc(c, &c->d_lenin, n);
    if (!e) e = dev_alloc(c, &B.len, n);
    if (!e) e = dev_alloc(c, &B.sumq, n);
    if (!e) e = dev_alloc(c, &B.flags, n);
    if (!e) e = dev_alloc(c, &B.clip5, n * A);
    if (!e) e = dev_alloc(c, &B.clip3, n * A);
    if (!e) e = dev_alloc(c, &B.mid_head, n);
    if (!e) e = dev_alloc(c, &B.mid_best, n * A);
    if (!e) e = dev_alloc(c, &B.mid_cnt, n);
    if (!e) e = dev_alloc(c, &B.mid_gate, n * A);
    if (!e) e = dev_alloc(c, &B.pool, (size_t)B.pool_cap);
    if (!e) e = dev_alloc(c, &B.pool_n, 4);
    if (!e) e = dev_alloc(c, &B.seg_cnt, n + 1);
    if (!e) e = dev_alloc(c, &B.chk_cnt, n + 1);
    if (c->cap_bases / 16u + n >= 0xFFFFFFF0ull) c->flat_scan = false;   // (chunk numbers are 32 bits: such a context keeps k_mid_scan1)
    if (c->cap_bases / 16u + n >= (1ull << 30) || !c->flat_scan) c->suffix_filter = 0;   // (k_mid_recheck's items: a chunk number and two bits)
    {
        // (no adapter the filter takes -- the ONT rapid adapters at the default -M, k = 16 --: none of its buffers either)
        bool any = false;
        for (int a = 0; a < p->n_adapters; a++) any |= c->P.Q[a] > 32 && c->P.Q[a] <= 64 && c->P.k_mid[a] >= 0 && c->P.k_mid[a] <= kSuffixMaxK;
        if (!any || !p->filter) c->suffix_filter = 0;
    }
    if (!e && c->suffix_filter) e = dev_alloc(c, &B.chk_mark, (size_t)4 * (size_t)((c->cap_bases / 16u + n) / 32u + 8u));
    if (c->suffix_filter) {
        // a mark in 25 chunks has room in the list (random sequence: a few in a thousand), the rest is done where it is found
        B.rc_cap = (uint32_t)std::min<uint64_t>((c->cap_bases / 16u + n) / 25u + 4096u, 1ull << 24);
        if (const char* e2 = knob("TGSF_RECHECK_CAP")) { int v = atoi(e2); if (v >= 0) B.rc_cap = (uint32_t)v; }   // test knob
        if (!e) e = dev_alloc(c, &B.rc_list, (size_t)B.rc_cap + 1);
        if (!e) e = dev_alloc(c, &B.rc_n, 4);
    }
    if (!e) e = dev_alloc(c, &B.nfr, n + 1);
    if (!e) e = dev_alloc(c, &B.scan_part, n / kScanTile + 2);
    if (!e) e = dev_alloc(c, &B.trimmed, n);
    B.rep_long_cap = (uint32_t)std::min<uint64_t>(c->cap_bases / kRepShare + 16, (uint64_t)B.fcap + 16);   // a long fragment holds more than kRepShare bases
    if (!e) e = dev_alloc(c, &B.rep_next, 2 + (size_t)B.rep_long_cap);
    if (p->min_repeat > 0 && p->kmer >= 12 && p->kmer <= 31) {
        // k_repeat_keys' last resort: room for the set of all k-mers of the longest fragment, half empty
        B.rep_tab_log2 = 4;
        while (B.rep_tab_log2 < 40 && (1ull << B.rep_tab_log2) < 2ull * c->max_read_len) B.rep_tab_log2++;
        if (!e) e = dev_alloc(c, &B.rep_tab, (size_t)1 << B.rep_tab_log2);
        if (!e) e = dev_alloc(c, &B.rep_lock, 4);
        if (!e) rt_memset(B.rep_lock, 0, 16, c->stream);
        B.rep_max_plog = kRepMaxPlog;
        if (const char* ev = knob("TGSF_REP_MAX_PLOG")) { int v = atoi(ev); if (v >= 0 && v <= 20) B.rep_max_plog = (uint32_t)v; }   // test knob
    }
    if (!e) e = dev_alloc(c, &B.tile_hist, 2 * ((size_t)B.max_tiles + 2));     // (two segments: DevBatch::tile_hist)
    if (!e) e = dev_alloc(c, &B.tile_cnt, 2 * ((size_t)B.max_tiles + 2));
    if (!e) e = dev_alloc(c, &B.tile_base, 2 * ((size_t)B.max_tiles + 2));
    if (!e) e = dev_alloc(c, &B.tile_fill, 2 * ((size_t)B.max_tiles + 2));
    if (!e) e = dev_alloc(c, &B.seg_info, 4);
    if (!e) e = dev_alloc(c, &B.perm, nitems);
    B.work_cap = (uint32_t)std::min<uint64_t>(2 * (c->cap_bases / kTileBases) + nitems + 16, 0x7FFFFFF0ull);
    if (!e) e = dev_alloc(c, &B.work, 2 * (size_t)B.work_cap);
    if (!e) e = dev_alloc(c, &B.frag_off, (size_t)B.fcap);
    if (!e) e = dev_alloc(c, &B.frag_qoff, (size_t)B.fcap);
    if (!e) e = dev_alloc(c, &B.frag_len, (size_t)B.fcap);
    if (!e) e = dev_alloc(c, &B.frag_sum, (size_t)B.fcap);
    if (!e) e = dev_alloc(c, &B.frag_read, (size_t)B.fcap);
    if (!e) e = dev_alloc(c, &B.frag_start, (size_t)B.fcap);
    if (!e) e = dev_alloc(c, &B.frag_flags, (size_t)B.fcap);
    if (!e) e = dev_alloc(c, &B.raw_tab, 2 * (size_t)c->n_bins * 5);
    if (!e) e = dev_alloc(c, &B.whole, n);
    if (!e) e = dev_alloc(c, &B.plan, 4);
    B.clean_force = p->only_qc ? 1u : 0u;
    // the clean tables as a by-product of the raw pass (DevBatch::spec): a filtering run with fixed trims in front of the
    // keep region (without trims a read kept whole is what the difference strategy handles already); the work list words hold
    // a staged length of 13 bits.  (Round 6: also with the repeat gate, -p.  Its verdict comes after the raw pass, and a
    // fragment it drops never reaches CalcAvgQuality, :1982-1994 -- but that is one more way for a read to turn out otherwise:
    // k_clean_plan finds such a read not `whole` and its speculated range is taken back out like any other's.)
    B.bp_allowed = (p->filter && !p->only_qc && (p->head_trim > 0 || p->tail_trim > 0) &&
                    p->head_trim >= 0 && p->tail_trim >= 0) ? 1u : 0u;
    static_assert(kTileBases < (1 << 13), "staged bytes of a tile fit the work list's 13 bits");
    if (const char* f = knob("TGSF_CLEAN_TABLES")) {      // test knob: "direct" | "difference" | "byproduct" (always speculate)
        if (!strcmp(f, "direct")) { B.clean_force = 1; B.bp_allowed = 0; }
        else if (!strcmp(f, "difference") && !p->only_qc) { B.clean_force = 2; B.bp_allowed = 0; }
        else if (!strcmp(f, "byproduct") && B.bp_allowed) B.clean_force = 3;      // (k_clean_plan_next leaves bp_state alone)
    }
    if (B.bp_allowed) {
        if (!e) e = dev_alloc(c, &B.spec, n);
        if (!e) e = dev_alloc(c, &B.spec_sum, n);
        if (!e) e = dev_alloc(c, &B.bp_state, 4);
        if (!e) e = dev_alloc(c, &c->bp_ring, TGSF_MAX_ENQUEUED);
    }
    {
        // traceback scratch: a window of the first location spans at most Q + k columns
        int maxcols = 1;
        for (int a = 0; a < p->n_adapters; a++) {
            int kmax = std::max(std::max(P.k_end[a], P.k_mid[a]), 0);
            maxcols = std::max(maxcols, P.Q[a] + std::min(P.Q[a], kmax) + 1);
        }
        c->scratch_cols = maxcols;
        // words per column: those of the widest adapter (1 / 2 / 4 in the register classes, ceil(Q / 64) beyond 256 bp).
        // An alignment whose columns would take 1 MiB and more is not traced back whole: edlib -- and alignment_length_w --
        // cut it by Hirschberg's scheme into pieces below that (include/edlib.cpp:1191-1193), so a lane's region never
        // needs more than 1 MiB (+ the two half columns of a cut).
        int col_words = P.max_nw;
        if (P.max_nw > 4) { col_words = 1; for (int a = 0; a < p->n_adapters; a++) col_words = std::max(col_words, (P.Q[a] + 63) / 64); }
        size_t lane_words = (size_t)(maxcols + 1) * 2 * (size_t)col_words;
        if (P.max_nw > 4) lane_words = std::min<size_t>(lane_words, (1u << 17) + 2 * (size_t)col_words) + 4 * (size_t)col_words + 64;
        B.scratch_wave_words = lane_words * 64;
        B.scratch_mid_wave0 = ((size_t)n * A * 2 + 63) / 64 + 1;
        const size_t waves = B.scratch_mid_wave0 + ((size_t)n * A + 63) / 64 + 1;
        if (!e) e = dev_alloc(c, &B.scratch, waves * B.scratch_wave_words);
    }
    if (!e) e = dev_alloc(c, &B.ctr, (size_t)c->ctr_words);
    if (!e) e = dev_alloc(c, &B.status, 4);
    if (!e) e = dev_alloc(c, &c->ovf_ring, TGSF_MAX_ENQUEUED);
    if (!e) e = dev_alloc(c, &c->d_out_reads, n);
    if (!e) e = dev_alloc(c, &c->d_out_frags, (size_t)B.fcap);
    if (!e) e = dev_alloc(c, &c->d_out_nfrags, 4);
    if (e) {
        g_create_error = c->error;
        tgsf_destroy(c);
        return e;
    }
    rt_memset(B.ctr, 0, c->ctr_words * 8, c->stream);
    rt_memset(B.status, 0, 16, c->stream);
    rt_memset(c->ovf_ring, 0, TGSF_MAX_ENQUEUED * 4, c->stream);
    B.ovf = c->ovf_ring;
    if (B.bp_allowed) {
        const uint32_t init[4] = {1u, 0u, 0u, 0u};          // the first batch speculates
        rt_memset(c->bp_ring, 0, TGSF_MAX_ENQUEUED * 4, c->stream);
        (void)rt_h2d(B.bp_state, init, sizeof init, c->stream);
        B.bp_used = c->bp_ring;
    }
    rt_memset(B.raw_tab, 0, 2 * (size_t)c->n_bins * 5 * 8, c->stream);
    rt_sync(c->stream);
    *out = c;
    return TGSF_OK;
}

// ---------------------------------------------------------------------------
// the pipeline
// ---------------------------------------------------------------------------
static const char* kStageNames[TGSF_N_STAGES] = {
    "prepare+sort", "stats_raw", "gate_reads", "end_tables_raw", "end_windows", "mid_scan",
    "mid_resolve", "regions", "repeat_gate", "stats_clean", "gate_frags+end_tables_clean", "finalize"};

extern "C" const char* tgsf_stage_name(int s) { return (s >= 0 && s < TGSF_N_STAGES) ? kStageNames[s] : ""; }

static unsigned blocks_for(uint64_t n, unsigned block) { return (unsigned)std::max<uint64_t>(1, (n + block - 1) / block); }

#if !defined(TGSF_EMUL)
// add the stage durations of every profiled batch not yet accounted for
static int harvest_profile(tgsf_ctx* c, rt_stream st)
{
    if (!c->prof_pending) return TGSF_OK;
    hipError_t he = hipStreamSynchronize(st);
    if (he != hipSuccess) return fail(c, TGSF_E_HIP, "stream synchronize failed: %s", hipGetErrorString(he));
    (void)hipStreamSynchronize(c->aux);
    for (int k = 0; k < c->prof_pending; k++)
        for (int i = 0; i < TGSF_N_STAGES; i++) {
            float ms = 0.f;
            // stages 3 (end_tables_raw) and 4 (end_windows) run on the auxiliary stream, beside stage 5
            hipError_t e = (i == 3 || i == 4) ? hipEventElapsedTime(&ms, c->ev_aux[k][i - 3], c->ev_aux[k][i - 2])
                                              : hipEventElapsedTime(&ms, c->ev[k][i], c->ev[k][i + 1]);
            if (e == hipSuccess) c->stage_ms[i] += ms;
        }
    c->prof_pending = 0;
    return TGSF_OK;
}
#endif

// exclusive prefix sums of a[0..n) in place, a[n] = total
static void scan_u32(const DevBatch& B, uint32_t* a, uint32_t n, rt_stream st, uint32_t* part = nullptr)
{
    if (!part) part = B.scan_part;                       // (sized for n = the reads of a batch)
    const unsigned nb = (n + kScanTile - 1) / kScanTile;
    TGSF_LAUNCH_COOP(k_scan_tiles, nb, 256, st, a, n, part);
    TGSF_LAUNCH_COOP(k_scan_top, 1, 64, st, part, (uint32_t)nb, a + n);
    TGSF_LAUNCH_COOP(k_scan_add, nb, 256, st, a, n, (const uint32_t*)part);
}

static int drain_pending(tgsf_ctx* c, bool implicit = false);

// redo = false: the whole pipeline of one batch, enqueued without a host round trip.
// redo = true (from tgsf_wait, a batch whose candidate pool overflowed, run again from its inputs): the kernels in front
// of the middle scan without their tallies (DevBatch::replay), the first scan for the minima, then the scan twice more
// -- counting the columns at each (read, adapter)'s minimum, and, with the pool grown to that many slots, handing
// them over in position order -- and everything behind it.  `slot`: the batch's word of ovf_ring.
static int run_pipeline(tgsf_ctx* c, const tgsf_batch_in* in, tgsf_read_result* d_reads, tgsf_fragment* d_frags,
                        uint32_t out_fcap, uint32_t* d_nfrags, rt_stream st, bool redo = false, uint32_t slot = 0)
{
    if (!redo) {
        if (c->n_pending == TGSF_MAX_ENQUEUED) { int e = drain_pending(c, true); if (e) return e; }
        slot = c->n_pending++;
        tgsf_ctx::Pending& pd = c->pending[slot];
        pd.in = *in; pd.reads = d_reads; pd.frags = d_frags; pd.fcap = out_fcap; pd.nfrags = d_nfrags; pd.st = st;
    }
    const bool profile = c->profile && !redo;
    (void)profile;
    DevBatch B = c->B;
    B.ovf = c->ovf_ring + slot;
    B.bp_used = c->bp_ring + slot;
    B.replay = redo ? 1u : 0u;
    const DevParams& P = c->P;
    B.seq = in->seq; B.qual = in->qual; B.off = in->offsets; B.len_in = in->lengths;
    B.qoff = in->qual_offsets ? in->qual_offsets : in->offsets;
    B.n = in->n_reads; B.n_bytes = in->n_bytes;
    const uint32_t n = B.n;
    const int A = P.n_adapters;
    const unsigned T = 256;
    static_assert(kMidThreads == 256, "k_mid_flat / k_mid_scan1 are launched with T lanes a workgroup");
    c->last_stream = st;
    const unsigned gsmall = grid_cap(std::min(blocks_for(n, T), 2048u));
    const unsigned gstats = grid_cap(c->stats_grid);   // 3 blocks (12 waves) per CU on 256 CUs: LDS-limited
    const size_t tl = 2 * ((size_t)B.max_tiles + 2) * 4;          // (both segments)
    const unsigned gfold = grid_cap(std::min(blocks_for((uint64_t)P.n_bins * 5, T), 1024u));
    int stage = 0;
#if !defined(TGSF_EMUL)
    if (profile && c->prof_pending == tgsf_ctx::kProfRing) { int e = harvest_profile(c, st); if (e) return e; }
    hipEvent_t* evs = c->ev[profile ? c->prof_pending : 0];
    hipEvent_t* evx = c->ev_aux[profile ? c->prof_pending : 0];
    if (profile) {
        for (int i = 0; i <= TGSF_N_STAGES; i++) if (!evs[i]) (void)hipEventCreate(&evs[i]);
        for (int i = 0; i < 3; i++) if (!evx[i]) (void)hipEventCreate(&evx[i]);
    }
    hipStream_t ax = c->aux;
#define STAGE_MARK() do { if (profile) (void)hipEventRecord(evs[stage], st); stage++; } while (0)
#else
#define STAGE_MARK() do { stage++; } while (0)
#endif
    STAGE_MARK();
    const unsigned gwork = grid_cap(std::min(blocks_for(in->n_bytes / kTileBases + n + 1, T), 4096u));
#if defined(TGSF_EMUL)
    rt_stream ax = st;
    (void)ax;
#endif
    {
    // -- prepare + counting sort of reads by tile count
    rt_memset(B.ovf, 0, 4, st);
    rt_memset(B.tile_hist, 0, tl, st);
    rt_memset(B.tile_fill, 0, tl, st);
    rt_memset(B.pool_n, 0, 4, st);
    rt_memset(B.plan, 0, 32, st);
    TGSF_LAUNCH(k_prepare, gsmall, T, st, P, B, c->max_read_len);
    TGSF_LAUNCH_COOP(k_tile_scan, 1, 64, st, B, B.bp_allowed ? 2u : 1u);       // a few hundred buckets: one wave
    TGSF_LAUNCH(k_tile_scatter<false>, gsmall, T, st, P, B);
    TGSF_LAUNCH(k_build_work<false>, gwork, T, st, P, B);
    STAGE_MARK();
    // -- raw stats
    // (a context that may speculate -- DevBatch::spec -- runs the variant of the raw pass that tallies the clean bins too;
    // the text is fetched with non-temporal loads: 2.24 -> 2.06 ms, 5.4 -> 5.8 TB/s, round 3)
    if (B.bp_allowed) {
        TGSF_LAUNCH((k_stats<false, true, true>), gstats, 64 * kStatsWaves, st, P, B);
        // (the bytes behind a speculated fragment: few, large workgroups -- each adds its LDS tallies to the table once)
        if (P.tail_trim > 0) TGSF_LAUNCH(k_tail_fix, grid_cap(std::min(blocks_for(n, 1024), 128u)), 1024, st, P, B);
    } else TGSF_LAUNCH((k_stats<false, true>), gstats, 64 * kStatsWaves, st, P, B);
    if (!redo) TGSF_LAUNCH(k_fold_raw<false>, gfold, T, st, P, B);   // (a second run: the batch's raw tallies are in the tables already)
    STAGE_MARK();
    TGSF_LAUNCH(k_gate_reads, gsmall, T, st, P, B);
    STAGE_MARK();
    // The 5'/3' QC tables and the end-window searches only need the gate; they are short,
    // latency-bound kernels, so they run on the auxiliary stream beside the middle scan.
#if !defined(TGSF_EMUL)
    (void)hipEventRecord(c->ev_fork, st);
    (void)hipStreamWaitEvent(ax, c->ev_fork, 0);
    if (profile) (void)hipEventRecord(evx[0], ax);
#endif
    for (uint32_t slab = 0; !redo && slab * (uint32_t)kMaxBcLen < (uint32_t)P.bc_len; slab++)
        TGSF_LAUNCH(k_end_tables<false>, grid_cap(c->endtab_grid), 64 * kEndWaves, ax, P, B, slab);
#if !defined(TGSF_EMUL)
    if (profile) (void)hipEventRecord(evx[1], ax);
#endif
    if (P.filter && A > 0) {
        const uint64_t nw = (uint64_t)n * A * 2;
        if (P.max_nw > 4) TGSF_LAUNCH(k_end_windows<kWideNW>, blocks_for(nw, 64), 64, ax, P, B);
        else if (P.max_nw > 2) TGSF_LAUNCH(k_end_windows<4>, blocks_for(nw, 64), 64, ax, P, B);
        else TGSF_LAUNCH(k_end_windows<2>, blocks_for(nw, 64), 64, ax, P, B);
    }
#if !defined(TGSF_EMUL)
    if (profile) (void)hipEventRecord(evx[2], ax);
    (void)hipEventRecord(c->ev_join, ax);
#endif
    }
    STAGE_MARK();
    STAGE_MARK();
    if (P.filter && A > 0) {
        // the first scan of a batch by the flat kernel (adapters of at most 64 bp); one lane per stretch of the schedule:
        // the number of stretches is known on the device only, so the grid covers the longest sequence the batch can have
        const bool flat = c->flat_scan;
        unsigned gflat = 1;
        uint64_t flat_chunks = 0;                  // upper bound of the batch's chunk count, known on the host
        {
            scan_u32(B, B.seg_cnt, n, st);
            if (flat) {
                scan_u32(B, B.chk_cnt, n, st);
                FlatSchedule S;
                const uint64_t tb = std::min<uint64_t>(in->n_bytes / 16u + (uint64_t)n, c->cap_bases / 16u + c->cap_reads);
                flat_chunks = tb;
                flat_schedule((uint32_t)tb, B.flat_pmax, B.flat_pmin, B.flat_f0, S);
                // The batch's real chunk count T' <= tb is on the device.  The number of stretches is not monotonic in it: a group
                // of 64 long stretches that a slightly smaller T' no longer fills falls to later phases of shorter stretches
                // (measured over all T' <= 200 000 and samples up to 30 M: at most 640 more than at tb for 256/16-chunk
                // stretches) -- room for one group of the longest stretches dealt as the shortest, twice; a lane beyond the
                // schedule's end exits at once, and the kernel strides by its grid should a schedule ever outgrow this.
                gflat = grid_cap(blocks_for((uint64_t)S.d0[S.nph] + 128ull * (B.flat_pmax / B.flat_pmin) + 512u, T));
            }
        }
        // upper bound of the segment count, known on the host: no device round trip
        const uint64_t max_segs = in->n_bytes / (uint64_t)P.seg_cols + 2ull * n + 1;
        const unsigned gseg = blocks_for(max_segs, T);
        const unsigned gmid = gseg;
        rt_stream ms = st;
        (void)ms;
        auto launch_scans = [&](uint32_t mode) {
            DevBatch Bm = B;
            Bm.mid_mode = mode;
            int a = 0;
            while (a < A) {
                if (P.Q[a] > 256) { TGSF_LAUNCH(k_mid_scan_wide, gseg, T, ms, P, Bm, a); a++; continue; }
                if (P.Q[a] > 192) { TGSF_LAUNCH(k_mid_scanw<4>, gseg, T, ms, P, Bm, a); a++; continue; }
                if (P.Q[a] > 128) { TGSF_LAUNCH(k_mid_scanw<3>, gseg, T, ms, P, Bm, a); a++; continue; }
                if (P.Q[a] > 64) { TGSF_LAUNCH(k_mid_scanw<2>, gseg, T, ms, P, Bm, a); a++; continue; }
                // up to four adapters of one word class per pass: <= 32 bp (one dword per column; two a pass, below), 33..64 bp
                // (one qword), or -- the flat scan only -- 33..64 bp within few differences (the last 32 rows as a filter, the
                // rest rechecked)
                auto cls = [&](int x) {
                    if (P.Q[x] <= 32 && !c->no_hot32) return 0;
                    if (flat && mode == 0 && c->suffix_filter && P.Q[x] > 32 && P.k_mid[x] >= 0 && P.k_mid[x] <= kSuffixMaxK) return 2;
                    return 1;
                };
                const int kind = cls(a);
                const bool narrow = kind == 0;
                // (measured, ligation 28- + 22-bp pairs at -M 22: four one-dword columns a lane take 193 registers = 2 waves per
                // SIMD and scan in 11.3 ms; two passes of two in 6.3 ms, 394 -> 610 Gbases/s.  Four qword columns a pass and
                // two passes of two are equal, 8.6 / 8.4 ms: those stay one pass.  profiles/r05_pass_width_ab.txt)
                const int width = narrow ? 2 : 4;
                int na = 0;
                while (a + na < A && na < width && P.Q[a + na] <= 64 && cls(a + na) == kind) na++;
                if (kind == 2) {
                    const unsigned lp = 0;
                    Bm.mark_stride = (uint32_t)((flat_chunks / 32u + 7u) & ~3ull);            // (k_mid_marks reads four words a load)
                    rt_memset(Bm.chk_mark, 0, (size_t)na * Bm.mark_stride * 4u, ms);
                    rt_memset(Bm.rc_n, 0, 4, ms);
                    if (c->suffix_filter == 1) switch (na) {
                    case 1: TGSF_LAUNCH_LDS((k_mid_flat<1, Hot32, 1>), gflat, T, lp, ms, P, Bm, a, na); break;
                    case 2: TGSF_LAUNCH_LDS((k_mid_flat<2, Hot32, 1>), gflat, T, lp, ms, P, Bm, a, na); break;
                    case 3: TGSF_LAUNCH_LDS((k_mid_flat<3, Hot32, 1>), gflat, T, lp, ms, P, Bm, a, na); break;
                    default: TGSF_LAUNCH_LDS((k_mid_flat<4, Hot32, 1>), gflat, T, lp, ms, P, Bm, a, na); break;
                    }
                    else switch (na) {
                    case 1: TGSF_LAUNCH_LDS((k_mid_flat<1, Hot32, 2>), gflat, T, lp, ms, P, Bm, a, na); break;
                    case 2: TGSF_LAUNCH_LDS((k_mid_flat<2, Hot32, 2>), gflat, T, lp, ms, P, Bm, a, na); break;
                    case 3: TGSF_LAUNCH_LDS((k_mid_flat<3, Hot32, 2>), gflat, T, lp, ms, P, Bm, a, na); break;
                    default: TGSF_LAUNCH_LDS((k_mid_flat<4, Hot32, 2>), gflat, T, lp, ms, P, Bm, a, na); break;
                    }
                    const unsigned grc = grid_cap(blocks_for(flat_chunks / 32u + 1u, (unsigned)kRecheckWords));
                    TGSF_LAUNCH_COOP(k_mid_marks, grc, T, ms, P, Bm, a, na);
                    TGSF_LAUNCH(k_mid_recheck, grid_cap(2048u), T, ms, P, Bm, a, na);
                    a += na;
                    continue;
                }
                if (flat && mode == 0) {
                    const unsigned lp = 0;
                    if (narrow) switch (na) {
                    case 1: TGSF_LAUNCH_LDS((k_mid_flat<1, Hot32>), gflat, T, lp, ms, P, Bm, a, na); break;
                    default: TGSF_LAUNCH_LDS((k_mid_flat<2, Hot32>), gflat, T, lp, ms, P, Bm, a, na); break;
                    }
                    else switch (na) {
                    case 1: TGSF_LAUNCH_LDS((k_mid_flat<1, Hot>), gflat, T, lp, ms, P, Bm, a, na); break;
                    case 2: TGSF_LAUNCH_LDS((k_mid_flat<2, Hot>), gflat, T, lp, ms, P, Bm, a, na); break;
                    case 3: TGSF_LAUNCH_LDS((k_mid_flat<3, Hot>), gflat, T, lp, ms, P, Bm, a, na); break;
                    default: TGSF_LAUNCH_LDS((k_mid_flat<4, Hot>), gflat, T, lp, ms, P, Bm, a, na); break;
                    }
                    a += na;
                    continue;
                }
                if (narrow) switch (na) {
                case 1: TGSF_LAUNCH((k_mid_scan1<1, Hot32>), gmid, T, ms, P, Bm, a, na); break;
                default: TGSF_LAUNCH((k_mid_scan1<2, Hot32>), gmid, T, ms, P, Bm, a, na); break;
                }
                else switch (na) {
                case 1: TGSF_LAUNCH((k_mid_scan1<1, Hot>), gmid, T, ms, P, Bm, a, na); break;
                case 2: TGSF_LAUNCH((k_mid_scan1<2, Hot>), gmid, T, ms, P, Bm, a, na); break;
                case 3: TGSF_LAUNCH((k_mid_scan1<3, Hot>), gmid, T, ms, P, Bm, a, na); break;
                default: TGSF_LAUNCH((k_mid_scan1<4, Hot>), gmid, T, ms, P, Bm, a, na); break;
                }
                a += na;
            }
        };
        launch_scans(0);
        if (redo) {
            // The first scan left every (read, adapter)'s minimum in mid_best.  Now every lane counts the columns AT those
            // minima (what edlib reports, include/edlib.cpp:660-672) per adapter; a prefix sum turns the counts into slots;
            // the pool is grown to the total; the lanes write their columns at their own slots: every read's candidates
            // end up as one array in ascending order of position.
            const uint64_t cells = max_segs * (uint64_t)A + 1;
            if (cells > 0xFFFFFFF0ull) return fail(c, TGSF_E_CAPACITY, "middle scan: too many (segment, adapter) cells for the position-ordered pass");
            uint32_t *seg_n = nullptr, *part = nullptr;
            if (rt_malloc((void**)&seg_n, (size_t)(cells + 1) * 4 + 64) || rt_malloc((void**)&part, (size_t)(cells / kScanTile + 4) * 4 + 64)) {
                if (seg_n) rt_free(seg_n);
                return fail(c, TGSF_E_CAPACITY, "middle scan: no device memory for the position-ordered pass (%.1f GB)", (double)cells * 4e-9);
            }
            B.seg_n = seg_n;
            int rc = TGSF_OK;
            do {
                rt_memset(seg_n, 0, (size_t)(cells + 1) * 4, ms);
                TGSF_LAUNCH(k_mid_reset, gsmall, T, ms, B, A);
                launch_scans(1);
                // prefix sums over the cells the batch really has (its segment count is on the device: sum the upper bound;
                // cells beyond the last segment hold zeros and seg_n[used] == seg_n[any later cell] == total)
                scan_u32(B, seg_n, (uint32_t)cells, ms, part);
                uint32_t need = 0;
                int he = rt_d2h(&need, seg_n + cells, 4, ms);
                if (!he) he = rt_sync(ms);
                if (he) { rc = fail(c, TGSF_E_HIP, "middle scan (counting pass) failed: %s", rt_errstr(he)); break; }
                if (need > B.pool_cap) {
                    const uint64_t want = (uint64_t)need + 1024;
                    if (want > 0x7FFFFFF0ull) { rc = fail(c, TGSF_E_CAPACITY, "middle-adapter candidates: %u columns tie their reads' minima, more than one batch can list", need); break; }
                    void* np = nullptr;
                    if (rt_malloc(&np, (size_t)want * sizeof(MidCand) + 64)) {
                        rc = fail(c, TGSF_E_CAPACITY, "middle-adapter candidates: no device memory for %llu slots (%.1f GB)", (unsigned long long)want, (double)want * sizeof(MidCand) * 1e-9);
                        break;
                    }
                    for (void*& q : c->allocs) if (q == (void*)c->B.pool) q = np;
                    rt_free(c->B.pool);
                    c->B.pool = (MidCand*)np; c->B.pool_cap = (uint32_t)want;
                    B.pool = c->B.pool; B.pool_cap = c->B.pool_cap;
                    c->pool_regrown++;
                }
                if (knob("TGSF_TRACE_POOL"))
                    fprintf(stderr, "tgsf: candidate pool overflow (or a read with a long candidate list): %u columns at their reads' minima, pool of %u slots%s; scanning again in position order\n", need, B.pool_cap,
                            c->pool_regrown ? " (grown)" : "");
                rt_memset(B.ovf, 0, 4, ms);
                launch_scans(2);
                TGSF_LAUNCH(k_mid_link, gsmall, T, ms, B, A);
                DevBatch Bm = B;
                Bm.mid_mode = 2;
                // first locations and gates per (read, adapter), then every location on a lane of its own
                if (P.max_nw > 4) TGSF_LAUNCH(k_mid_resolve<kWideNW>, blocks_for((uint64_t)n * A, 64), 64, ms, P, Bm);
                else if (P.max_nw > 2) TGSF_LAUNCH(k_mid_resolve<4>, blocks_for((uint64_t)n * A, 64), 64, ms, P, Bm);
                else TGSF_LAUNCH(k_mid_resolve<2>, blocks_for((uint64_t)n * A, 64), 64, ms, P, Bm);
                const unsigned geach = grid_cap(std::min(blocks_for(need, 64), 65536u));
                if (need) {
                    if (P.max_nw > 4) TGSF_LAUNCH(k_mid_resolve_each<kWideNW>, geach, 64, ms, P, Bm);
                    else if (P.max_nw > 2) TGSF_LAUNCH(k_mid_resolve_each<4>, geach, 64, ms, P, Bm);
                    else TGSF_LAUNCH(k_mid_resolve_each<2>, geach, 64, ms, P, Bm);
                }
                he = rt_sync(ms);                         // seg_n is released below
                if (he) rc = fail(c, TGSF_E_HIP, "middle scan (position-ordered pass) failed: %s", rt_errstr(he));
            } while (0);
            rt_free(seg_n);
            rt_free(part);
            B.seg_n = nullptr;
            if (rc) return rc;
        }
    }
    STAGE_MARK();
    if (P.filter && A > 0 && !redo) {
        if (P.max_nw > 4) TGSF_LAUNCH(k_mid_resolve<kWideNW>, blocks_for((uint64_t)n * A, 64), 64, st, P, B);
        else if (P.max_nw > 2) TGSF_LAUNCH(k_mid_resolve<4>, blocks_for((uint64_t)n * A, 64), 64, st, P, B);
        else TGSF_LAUNCH(k_mid_resolve<2>, blocks_for((uint64_t)n * A, 64), 64, st, P, B);
    }
    STAGE_MARK();
#if !defined(TGSF_EMUL)
    (void)hipStreamWaitEvent(st, c->ev_join, 0);      // regions need the end-window results
#endif
    TGSF_LAUNCH(k_regions<false>, gsmall, T, st, P, B);
    scan_u32(B, B.nfr, n, st);
    TGSF_LAUNCH(k_regions<true>, gsmall, T, st, P, B);
    STAGE_MARK();
    if (P.min_repeat > 0 && !P.only_qc) {
        // one 1024-lane workgroup per CU (152 KB of LDS each); fragments are handed out through B.rep_next.  k <= 11: the
        // 4^k-bit set swept as LDS bitmaps; above: a hashed map + the full keys of the few it cannot tell apart (k = 12
        // would be 16 sweeps: 18 ms a batch against 5)
        rt_memset(B.rep_next, 0, 2 * sizeof(uint32_t), st);
        const int keys_from = 12;
        if (P.kmer < keys_from && P.kmer <= 13) TGSF_LAUNCH(k_repeat, grid_cap(256u), kRepThreads, st, P, B);
        else {
            TGSF_LAUNCH(k_repeat_long, gsmall, T, st, P, B);
            if (P.kmer <= 15) TGSF_LAUNCH(k_repeat_keys<false>, grid_cap(256u), kRepThreads, st, P, B);
            else TGSF_LAUNCH(k_repeat_keys<true>, grid_cap(256u), kRepThreads, st, P, B);
        }
    }
    STAGE_MARK();
    // -- clean stats over the fragments
    rt_memset(B.tile_hist, 0, tl, st);
    rt_memset(B.tile_fill, 0, tl, st);
    const unsigned gfr = grid_cap(std::min(blocks_for((uint64_t)B.fcap + n, T), 2048u));
    TGSF_LAUNCH(k_clean_plan, gsmall, T, st, P, B);
    if (B.bp_allowed && !redo) TGSF_LAUNCH(k_clean_plan_next, 1, 64, st, B);
    TGSF_LAUNCH(k_fold_raw<true>, gfold, T, st, P, B);
    TGSF_LAUNCH(k_frag_prepare, gfr, T, st, P, B);
    TGSF_LAUNCH_COOP(k_tile_scan, 1, 64, st, B, 2u);   // a few hundred buckets: one wave (fragments; reads to take back out)
    TGSF_LAUNCH(k_tile_scatter<true>, gfr, T, st, P, B);
    TGSF_LAUNCH(k_build_work<true>, gwork, T, st, P, B);
    TGSF_LAUNCH((k_stats<true, true>), gstats, 64 * kStatsWaves, st, P, B);
    STAGE_MARK();
    TGSF_LAUNCH(k_gate_frags, gfr, T, st, P, B);
    for (uint32_t slab = 0; slab * (uint32_t)kMaxBcLen < (uint32_t)P.bc_len; slab++)
        TGSF_LAUNCH(k_end_tables<true>, grid_cap(c->endtab_grid), 64 * kEndWaves, st, P, B, slab);
    STAGE_MARK();
    TGSF_LAUNCH(k_finalize, gsmall, T, st, B, d_reads, d_frags, out_fcap, d_nfrags);
    STAGE_MARK();
#if !defined(TGSF_EMUL)
    hipError_t he = hipGetLastError();
    if (he != hipSuccess) return fail(c, TGSF_E_HIP, "kernel launch failed: %s", hipGetErrorString(he));
    if (profile) { c->prof_batches++; c->prof_pending++; }
#endif
    return TGSF_OK;
}

// Host-to-device copy of a batch's bytes.  Memory the runtime knows (pinned / registered) is copied as it is.  Plain
// pageable memory -- the command line hands over slices of its input mapping -- goes through the context's two pinned
// staging buffers, the CPU copy of one piece overlapping the DMA of the previous one: the runtime's own pageable path
// pins the caller's pages piece by piece instead, which is as fast but registers MMU notifiers on the caller's
// mapping (a caller that drops parts of that mapping meanwhile then stalls the device queues).
static int text_h2d(tgsf_ctx* c, uint8_t* dst, const uint8_t* src, uint64_t n, rt_stream st)
{
#if !defined(TGSF_EMUL)
    hipPointerAttribute_t at;
    const bool known = hipPointerGetAttributes(&at, src) == hipSuccess && at.type != hipMemoryTypeUnregistered;
    (void)hipGetLastError();
    if (!known && c->stage[0] && n > (1u << 20)) {
        int he = 0, k = 0;
        for (uint64_t o = 0; o < n; o += tgsf_ctx::kStageBytes, k ^= 1) {
            const size_t m = (size_t)std::min<uint64_t>(tgsf_ctx::kStageBytes, n - o);
            (void)hipEventSynchronize(c->stage_ev[k]);           // the DMA that last read this buffer
            memcpy(c->stage[k], src + o, m);
            he |= (int)hipMemcpyAsync(dst + o, c->stage[k], m, hipMemcpyHostToDevice, st);
            he |= (int)hipEventRecord(c->stage_ev[k], st);
        }
        return he;
    }
#endif
    (void)c;
    return rt_h2d(dst, src, n, st);
}

static int check_batch(tgsf_ctx* c, const tgsf_batch_in* in)
{
    if (!in || !in->seq || (!in->qual && !c->P.no_qual) || !in->offsets) return fail(c, TGSF_E_INVALID, "null batch pointer");
    if (in->n_reads == 0) return fail(c, TGSF_E_INVALID, "empty batch");
    if (in->n_reads > c->cap_reads) return fail(c, TGSF_E_CAPACITY, "batch has %u reads, context was sized for %u", in->n_reads, c->cap_reads);
    if (in->qual_offsets && !in->lengths) return fail(c, TGSF_E_INVALID, "qual_offsets requires explicit lengths");
    return TGSF_OK;
}

static int check_status(tgsf_ctx* c)
{
    const uint32_t code = c->h_status[0], detail = c->h_status[1];
    if (!code) return TGSF_OK;
    rt_memset(c->B.status, 0, 16, c->stream);
    rt_sync(c->stream);
    switch (code) {
    case DS_BAD_LEN: return fail(c, TGSF_E_DATA, "read %u: length 0 or above max_read_len %u", detail, c->max_read_len);
    case DS_TOO_MANY_REGIONS: return fail(c, TGSF_E_CAPACITY, "read %u has more than %d disjoint drop regions", detail, kMaxRegions);
    case DS_FRAG_CAP: return fail(c, TGSF_E_CAPACITY, "fragment capacity exceeded (%u)", detail);
    case DS_BAD_MEANQ: return fail(c, TGSF_E_DATA, "read %u: mean quality outside [0,256)", detail);
    default: return fail(c, TGSF_E_HIP, "device status %u", code);
    }
}

static int finish_pending(tgsf_ctx* c)
{
    tgsf_batch_out* out = c->pend_out;
    if (!out) return TGSF_OK;
    c->pend_out = nullptr;
    const uint32_t nf = *c->pend_nf_p;
    out->n_frags = nf;
    if (nf > out->frag_capacity || (nf && !out->frags))
        return fail(c, TGSF_E_CAPACITY, "batch produced %u fragments, caller provided room for %u", nf, out->frag_capacity);
    if (nf) {
        int he = rt_d2h(out->frags, c->d_out_frags, (size_t)nf * sizeof(tgsf_fragment), c->stream);
        if (!he) he = rt_sync(c->stream);
        if (he) return fail(c, TGSF_E_HIP, "device to host copy failed");
    }
    return TGSF_OK;
}

// Everything enqueued on the context completes: the streams are drained, the device's status words looked at, and
// every batch whose candidate pool overflowed (a read whose minimum is tied column after column, e.g. a homopolymer
// against a homopolymer adapter: nothing behind its middle scan has touched it) is run again from its inputs.
// implicit: not from tgsf_wait but from the submit that found TGSF_MAX_ENQUEUED batches enqueued.  The streams and buffers of
// those batches are the caller's, and only tgsf_wait is documented as the point up to which they must stay alive: here the
// whole device is waited for instead of the saved stream handles (a destroyed stream is no valid handle), and a batch that
// would have to be run again from its inputs (a pool overflow) is an error -- its inputs may be gone -- that says what to do.
static int drain_pending(tgsf_ctx* c, bool implicit)
{
    int e = 0;
    const uint32_t np = c->n_pending;
#if !defined(TGSF_EMUL)
    // batches enqueued with tgsf_submit_device run on the caller's streams (and the auxiliary one): all must be idle
    // before the status words mean anything
    (void)hipSetDevice(c->device);
    if (implicit) e = (int)hipDeviceSynchronize();
    for (uint32_t i = 0; i < np && !e && !implicit; i++) {
        bool seen = c->pending[i].st == c->stream;
        for (uint32_t j = 0; j < i && !seen; j++) seen = c->pending[j].st == c->pending[i].st;
        if (!seen) e = (int)hipStreamSynchronize(c->pending[i].st);
    }
    if (!e) e = (int)hipStreamSynchronize(c->aux);
#endif
    if (!e) e = rt_d2h(c->h_status, c->B.status, 16, c->stream);
    if (!e && np) e = rt_d2h(c->h_ovf, c->ovf_ring, (size_t)np * 4, c->stream);
    if (!e) e = rt_sync(c->stream);
    c->n_pending = 0;
    if (e) { c->pend_out = nullptr; return fail(c, TGSF_E_HIP, "stream synchronize failed: %s", rt_errstr(e)); }
    e = check_status(c);
    if (e) { c->pend_out = nullptr; return e; }
    for (uint32_t i = 0; i < np; i++) {
        if (!c->h_ovf[i]) continue;
        if (implicit) {
            c->pend_out = nullptr;
            return fail(c, TGSF_E_INVALID, "batch %u of the %u enqueued since the last tgsf_wait has to be run again from its inputs (its middle-adapter candidates "
                                           "outgrew the pool) and only tgsf_wait may do that: call tgsf_wait at least every TGSF_MAX_ENQUEUED (%d) batches", i, np, TGSF_MAX_ENQUEUED);
        }
        const tgsf_ctx::Pending pd = c->pending[i];
        e = run_pipeline(c, &pd.in, pd.reads, pd.frags, pd.fcap, pd.nfrags, pd.st, true, i);
        if (!e && c->pend_out && pd.reads == c->d_out_reads) {       // the batch of tgsf_submit_async: its copies again
            int he = rt_d2h(c->pend_nf_p, c->d_out_nfrags, 4, pd.st);
            he |= rt_d2h(c->pend_out->reads, c->d_out_reads, (size_t)pd.in.n_reads * sizeof(tgsf_read_result), pd.st);
            if (he) e = fail(c, TGSF_E_HIP, "device to host copy failed");
        }
        uint32_t again = 0;
        if (!e) {
            int he = rt_d2h(c->h_status, c->B.status, 16, pd.st);
            if (!he) he = rt_d2h(&again, c->ovf_ring + i, 4, pd.st);
            if (!he) he = rt_sync(pd.st);
#if !defined(TGSF_EMUL)
            if (!he) he = (int)hipStreamSynchronize(c->aux);
#endif
            if (he) e = fail(c, TGSF_E_HIP, "stream synchronize failed: %s", rt_errstr(he));
        }
        if (!e) e = check_status(c);
        if (!e && again) e = fail(c, TGSF_E_HIP, "middle-adapter candidate pool overflowed again after it was sized to fit");
        if (e) { c->pend_out = nullptr; return e; }
    }
    return TGSF_OK;
}

extern "C" int tgsf_wait(tgsf_ctx* c)
{
    if (!c) return TGSF_E_INVALID;
    const int e = drain_pending(c);
    if (e) return e;
    return finish_pending(c);
}

extern "C" int tgsf_submit_device(tgsf_ctx* c, const tgsf_batch_in* in, tgsf_batch_out* out,
                                  uint32_t* d_n_frags, void* hip_stream)
{
    if (!c) return TGSF_E_INVALID;
    int e = check_batch(c, in);
    if (e) return e;
    if (!out || !out->reads) return fail(c, TGSF_E_INVALID, "null output");
    if (((uintptr_t)in->seq & 15u) || (!c->P.no_qual && ((uintptr_t)in->qual & 15u))) return fail(c, TGSF_E_INVALID, "seq/qual device pointers must be 16-byte aligned");
#if !defined(TGSF_EMUL)
    (void)hipSetDevice(c->device);
#endif
    rt_stream st = hip_stream ? (rt_stream)hip_stream : c->stream;
    return run_pipeline(c, in, out->reads, out->frags, out->frags ? out->frag_capacity : 0u, d_n_frags, st);
}

extern "C" int tgsf_submit_async(tgsf_ctx* c, const tgsf_batch_in* in, tgsf_batch_out* out)
{
    if (!c) return TGSF_E_INVALID;
    if (c->pend_out) return fail(c, TGSF_E_INVALID, "a batch is already pending on this context: call tgsf_wait first");
    int e = check_batch(c, in);
    if (e) return e;
    if (!out || !out->reads) return fail(c, TGSF_E_INVALID, "null output");
    const uint32_t n = in->n_reads;
    uint64_t span = in->n_bytes;
    if (!span) {
        span = in->lengths ? in->offsets[n - 1] + in->lengths[n - 1] : in->offsets[n];
        if (in->qual_offsets) span = std::max<uint64_t>(span, in->qual_offsets[n - 1] + in->lengths[n - 1]);
    }
    if (span > c->cap_bases + 16ull * c->cap_reads) return fail(c, TGSF_E_CAPACITY, "batch spans %llu bytes, context was sized for %llu bases", (unsigned long long)span, (unsigned long long)c->cap_bases);
#if !defined(TGSF_EMUL)
    (void)hipSetDevice(c->device);
#endif
    rt_stream st = c->stream;
    int he = 0;
    const bool one_buffer = in->qual == in->seq || c->P.no_qual;   // raw FASTQ text: both streams are read in place (no_qual: none)
    he |= text_h2d(c, c->d_seq, in->seq, span, st);
    if (!one_buffer) he |= text_h2d(c, c->d_qual, in->qual, span, st);
    he |= rt_h2d(c->d_off, in->offsets, (size_t)(in->lengths ? n : n + 1) * 8, st);
    if (in->lengths) he |= rt_h2d(c->d_lenin, in->lengths, (size_t)n * 4, st);
    if (in->qual_offsets) he |= rt_h2d(c->d_qoff, in->qual_offsets, (size_t)n * 8, st);
    if (he) return fail(c, TGSF_E_HIP, "host to device copy failed");
    tgsf_batch_in din = *in;
    din.seq = c->d_seq; din.qual = one_buffer ? c->d_seq : c->d_qual; din.offsets = c->d_off;
    din.lengths = in->lengths ? c->d_lenin : nullptr;
    din.qual_offsets = in->qual_offsets ? c->d_qoff : nullptr;
    din.n_bytes = span;
    e = run_pipeline(c, &din, c->d_out_reads, c->d_out_frags, c->B.fcap, c->d_out_nfrags, st);
    if (e) return e;
    *c->pend_nf_p = 0;
    he |= rt_d2h(c->pend_nf_p, c->d_out_nfrags, 4, st);
    he |= rt_d2h(out->reads, c->d_out_reads, (size_t)n * sizeof(tgsf_read_result), st);
    if (he) return fail(c, TGSF_E_HIP, "device to host copy failed");
    c->pend_out = out;
    return TGSF_OK;
}

extern "C" int tgsf_submit(tgsf_ctx* c, const tgsf_batch_in* in, tgsf_batch_out* out)
{
    const int e = tgsf_submit_async(c, in, out);
    return e ? e : tgsf_wait(c);
}

// ---------------------------------------------------------------------------
// tallies
// ---------------------------------------------------------------------------
extern "C" int tgsf_counters_len(tgsf_ctx* c, uint64_t* n_words, int32_t* bc_len, uint32_t* n_bins)
{
    if (!c) return TGSF_E_INVALID;
    if (n_words) *n_words = c->ctr_words;
    if (bc_len) *bc_len = c->P.bc_len;
    if (n_bins) *n_bins = c->n_bins;
    return TGSF_OK;
}

extern "C" int tgsf_counters(tgsf_ctx* c, uint64_t* dst, uint64_t n_words)
{
    if (!c || !dst) return TGSF_E_INVALID;
    if (n_words < c->ctr_words) return fail(c, TGSF_E_CAPACITY, "counter buffer too small");
    int e = tgsf_wait(c);
    if (e) return e;
    int he = rt_d2h(dst, c->B.ctr, c->ctr_words * 8, c->stream);
    if (!he) he = rt_sync(c->stream);
    return he ? fail(c, TGSF_E_HIP, "device to host copy failed") : TGSF_OK;
}

extern "C" int tgsf_counters_used(tgsf_ctx* c, uint64_t* dst, uint64_t n_words, uint64_t rows[2])
{
    if (!c || !dst) return TGSF_E_INVALID;
    if (n_words < c->ctr_words) return fail(c, TGSF_E_CAPACITY, "counter buffer too small");
    int e = tgsf_wait(c);
    if (e) return e;
    // fixed part: DropInfo, both DiffQual histograms, the "rows used" words, the eight end tables
    const size_t head = tgsf_ctr_bin_table(0, c->P.bc_len, c->n_bins);
    int he = rt_d2h(dst, c->B.ctr, head * 8, c->stream);
    if (!he) he = rt_sync(c->stream);
    if (he) return fail(c, TGSF_E_HIP, "device to host copy failed");
    const uint64_t used[2] = {std::min<uint64_t>(dst[TGSF_CTR_ROWS], c->n_bins), std::min<uint64_t>(dst[TGSF_CTR_ROWS + 1], c->n_bins)};
    for (int b = 0; b < 4 && !he; b++) {
        const size_t at = tgsf_ctr_bin_table(b, c->P.bc_len, c->n_bins);
        const uint64_t r = used[b >> 1];
        if (r) he = rt_d2h(dst + at, c->B.ctr + at, (size_t)r * 5 * 8, c->stream);
    }
    if (!he) he = rt_sync(c->stream);
    if (rows) { rows[0] = used[0]; rows[1] = used[1]; }
    return he ? fail(c, TGSF_E_HIP, "device to host copy failed") : TGSF_OK;
}

extern "C" int tgsf_counters_merge(tgsf_ctx* dst, tgsf_ctx* src)
{
    if (!dst || !src || dst == src) return TGSF_E_INVALID;
    if (dst->ctr_words != src->ctr_words || dst->P.bc_len != src->P.bc_len || dst->device != src->device)
        return fail(dst, TGSF_E_INVALID, "tgsf_counters_merge: the contexts differ in tally layout or device");
    int e = tgsf_wait(src);
    if (e) return fail(dst, e, "%s", src->error.c_str());
    if ((e = tgsf_wait(dst))) return e;
    TGSF_LAUNCH(k_ctr_merge, grid_cap(blocks_for(dst->ctr_words, 256)), 256, dst->stream, dst->B.ctr, (const uint64_t*)src->B.ctr, dst->ctr_words);
    const int he = rt_sync(dst->stream);
    return he ? fail(dst, TGSF_E_HIP, "tgsf_counters_merge failed") : TGSF_OK;
}

extern "C" int tgsf_counters_device(tgsf_ctx* c, void** d_ptr, uint64_t* n_words)
{
    if (!c || !d_ptr) return TGSF_E_INVALID;
    *d_ptr = c->B.ctr;
    if (n_words) *n_words = c->ctr_words;
    return TGSF_OK;
}

extern "C" int tgsf_reset_counters(tgsf_ctx* c)
{
    if (!c) return TGSF_E_INVALID;
    int he = rt_memset(c->B.ctr, 0, c->ctr_words * 8, c->stream);
    if (!he) he = rt_memset(c->B.raw_tab, 0, 2 * (size_t)c->n_bins * 5 * 8, c->stream);
    if (!he) he = rt_sync(c->stream);
    return he ? fail(c, TGSF_E_HIP, "memset failed") : TGSF_OK;
}

extern "C" int tgsf_profile(tgsf_ctx* c, int enable)
{
    if (!c) return TGSF_E_INVALID;
#if !defined(TGSF_EMUL)
    { int e = harvest_profile(c, c->last_stream); if (e) return e; }
#endif
    c->profile = enable != 0;
    memset(c->stage_ms, 0, sizeof c->stage_ms);
    c->prof_batches = 0;
    return TGSF_OK;
}

extern "C" int tgsf_stage_times(tgsf_ctx* c, float ms[TGSF_N_STAGES], uint32_t* n_batches)
{
    if (!c || !ms) return TGSF_E_INVALID;
#if !defined(TGSF_EMUL)
    // batches submitted on a caller stream are complete once the caller synchronised it
    { int e = harvest_profile(c, c->last_stream); if (e) return e; }
#endif
    memcpy(ms, c->stage_ms, sizeof c->stage_ms);
    if (n_batches) *n_batches = c->prof_batches;
    return TGSF_OK;
}

// ---------------------------------------------------------------------------
// stand-alone alignments
// ---------------------------------------------------------------------------
extern "C" int tgsf_align_windows(tgsf_ctx* c, const uint8_t* seq, uint64_t n_bytes, const uint64_t* win_off,
                                  const uint32_t* win_len, const uint8_t* adapter_id, const int32_t* k, uint32_t n,
                                  int32_t* res, int32_t* ends)
{
    if (!c || !seq || !win_off || !win_len || !adapter_id || !k || !res || !ends) return TGSF_E_INVALID;
    if (!n) return TGSF_OK;
    for (uint32_t i = 0; i < n; i++) {
        if (adapter_id[i] >= c->P.n_adapters) return fail(c, TGSF_E_INVALID, "problem %u: adapter id out of range", i);
        if (win_len[i] == 0 || win_off[i] + win_len[i] > n_bytes) return fail(c, TGSF_E_INVALID, "problem %u: window outside the buffer", i);
        if (k[i] < 0) return fail(c, TGSF_E_INVALID, "problem %u: k < 0", i);
        const int Q = c->P.Q[adapter_id[i]];
        if (Q + std::min(Q, (int)k[i]) + 1 > c->scratch_cols)
            return fail(c, TGSF_E_CAPACITY, "problem %u: k larger than the context's thresholds allow for", i);
    }
    if ((size_t)n > (size_t)c->cap_reads * std::max(c->P.n_adapters, 1) * 2)
        return fail(c, TGSF_E_CAPACITY, "more alignment problems than the context was sized for");
#if !defined(TGSF_EMUL)
    (void)hipSetDevice(c->device);
#endif
    uint8_t *d_seq = nullptr, *d_aid = nullptr; uint64_t* d_off = nullptr; uint32_t* d_len = nullptr;
    int32_t *d_k = nullptr, *d_res = nullptr, *d_ends = nullptr;
    int e = 0;
    e |= rt_malloc((void**)&d_seq, n_bytes + 16);
    e |= rt_malloc((void**)&d_aid, n);
    e |= rt_malloc((void**)&d_off, (size_t)n * 8);
    e |= rt_malloc((void**)&d_len, (size_t)n * 4);
    e |= rt_malloc((void**)&d_k, (size_t)n * 4);
    e |= rt_malloc((void**)&d_res, (size_t)n * 16);
    e |= rt_malloc((void**)&d_ends, (size_t)n * 8);
    rt_stream st = c->stream;
    if (!e) {
        e |= rt_h2d(d_seq, seq, n_bytes, st);
        e |= rt_h2d(d_aid, adapter_id, n, st);
        e |= rt_h2d(d_off, win_off, (size_t)n * 8, st);
        e |= rt_h2d(d_len, win_len, (size_t)n * 4, st);
        e |= rt_h2d(d_k, k, (size_t)n * 4, st);
    }
    if (!e) {
        if (c->P.max_nw > 4)
            TGSF_LAUNCH(k_align_windows<kWideNW>, blocks_for(n, 64), 64, st, c->P, c->B, (const uint8_t*)d_seq, (const uint64_t*)d_off,
                        (const uint32_t*)d_len, (const uint8_t*)d_aid, (const int32_t*)d_k, n, d_res, d_ends);
        else if (c->P.max_nw > 2)
            TGSF_LAUNCH(k_align_windows<4>, blocks_for(n, 64), 64, st, c->P, c->B, (const uint8_t*)d_seq, (const uint64_t*)d_off,
                        (const uint32_t*)d_len, (const uint8_t*)d_aid, (const int32_t*)d_k, n, d_res, d_ends);
        else
            TGSF_LAUNCH(k_align_windows<2>, blocks_for(n, 64), 64, st, c->P, c->B, (const uint8_t*)d_seq, (const uint64_t*)d_off,
                        (const uint32_t*)d_len, (const uint8_t*)d_aid, (const int32_t*)d_k, n, d_res, d_ends);
        e |= rt_d2h(res, d_res, (size_t)n * 16, st);
        e |= rt_d2h(ends, d_ends, (size_t)n * 8, st);
        e |= rt_sync(st);
    }
    rt_free(d_seq); rt_free(d_aid); rt_free(d_off); rt_free(d_len); rt_free(d_k); rt_free(d_res); rt_free(d_ends);
    return e ? fail(c, TGSF_E_HIP, "tgsf_align_windows: device operation failed") : TGSF_OK;
}
