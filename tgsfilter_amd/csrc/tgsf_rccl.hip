// tgsf_rccl.hip -- libtgsf_rccl.so: the tally all-reduce of a multi-GPU job (include/tgsf_rccl.h).
// Built on the public ABI of libtgsf only (tgsf_wait, tgsf_counters_device) plus RCCL.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <string>

#include "tgsf_rccl.h"

static thread_local std::string g_err;

static int fail(int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

extern "C" const char* tgsf_rccl_last_error(void) { return g_err.c_str(); }

static_assert(sizeof(ncclUniqueId) == TGSF_RCCL_ID_BYTES, "ncclUniqueId is 128 bytes");

extern "C" int tgsf_rccl_unique_id(void* id128)
{
    if (!id128) return fail(TGSF_E_INVALID, "null argument");
    ncclUniqueId id;
    const ncclResult_t r = ncclGetUniqueId(&id);
    if (r != ncclSuccess) return fail(TGSF_E_HIP, "ncclGetUniqueId: %s", ncclGetErrorString(r));
    memcpy(id128, &id, sizeof id);
    return TGSF_OK;
}

extern "C" int tgsf_rccl_comm_init(int device, const void* id128, int rank, int world, void** nccl_comm)
{
    if (!id128 || !nccl_comm || world < 1 || rank < 0 || rank >= world) return fail(TGSF_E_INVALID, "bad argument");
    *nccl_comm = nullptr;
    if (hipSetDevice(device) != hipSuccess) return fail(TGSF_E_NO_DEVICE, "hipSetDevice(%d) failed", device);
    ncclUniqueId id;
    memcpy(&id, id128, sizeof id);
    ncclComm_t comm = nullptr;
    const ncclResult_t r = ncclCommInitRank(&comm, world, id, rank);
    if (r != ncclSuccess) return fail(TGSF_E_HIP, "ncclCommInitRank(rank %d of %d, device %d): %s", rank, world, device, ncclGetErrorString(r));
    *nccl_comm = comm;
    return TGSF_OK;
}

extern "C" int tgsf_rccl_comm_count(void* nccl_comm, int* n_ranks)
{
    if (!nccl_comm || !n_ranks) return fail(TGSF_E_INVALID, "null argument");
    const ncclResult_t r = ncclCommCount((ncclComm_t)nccl_comm, n_ranks);
    return r == ncclSuccess ? TGSF_OK : fail(TGSF_E_HIP, "ncclCommCount: %s", ncclGetErrorString(r));
}

extern "C" void tgsf_rccl_comm_destroy(void* nccl_comm)
{
    if (nccl_comm) (void)ncclCommDestroy((ncclComm_t)nccl_comm);
}

// buf = [ tallies (n words) | world slots of 4 words ]: this rank's "rows used" words move into its slot
__global__ void k_pack_rows(unsigned long long* buf, const unsigned long long* ctr, unsigned long long n, int rank, int world)
{
    const unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x;
    if (i < n) buf[i] = (i >= TGSF_CTR_ROWS && i < TGSF_CTR_ROWS + 4) ? 0ull : ctr[i];
    if (i < 4ull * world) buf[n + i] = ((int)(i / 4) == rank) ? ctr[TGSF_CTR_ROWS + (i & 3)] : 0ull;
}

__global__ void k_unpack_rows(unsigned long long* ctr, const unsigned long long* buf, unsigned long long n, int world)
{
    const unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned long long v = buf[i];
    if (i >= TGSF_CTR_ROWS && i < TGSF_CTR_ROWS + 4) {
        v = 0;
        for (int r = 0; r < world; r++) { const unsigned long long x = buf[n + 4ull * r + (i - TGSF_CTR_ROWS)]; v = x > v ? x : v; }
    }
    ctr[i] = v;
}

extern "C" int tgsf_rccl_allreduce_counters(tgsf_ctx* ctx, void* nccl_comm, int rank, int world, int check_layout, void* hip_stream)
{
    if (!ctx || !nccl_comm || world < 1 || rank < 0 || rank >= world) return fail(TGSF_E_INVALID, "bad argument");
    int e = tgsf_wait(ctx);                              // every batch of this rank is in the tallies
    if (e) return fail(e, "%s", tgsf_last_error(ctx));
    void* d_ctr = nullptr;
    uint64_t n = 0;
    if ((e = tgsf_counters_device(ctx, &d_ctr, &n))) return fail(e, "%s", tgsf_last_error(ctx));
    // the vector lives on the context's device: work there whatever the caller's current device is (and put that back)
    int prev_dev = -1, ctx_dev = -1;
    (void)hipGetDevice(&prev_dev);
    {
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, d_ctr) != hipSuccess) return fail(TGSF_E_HIP, "cannot tell the device of the tally vector");
        ctx_dev = at.device;
    }
    if (ctx_dev != prev_dev && hipSetDevice(ctx_dev) != hipSuccess) return fail(TGSF_E_HIP, "hipSetDevice(%d) failed", ctx_dev);
    struct Restore { int from, to; ~Restore() { if (from != to && to >= 0) (void)hipSetDevice(to); } } restore{ctx_dev, prev_dev};
    ncclComm_t comm = (ncclComm_t)nccl_comm;
    hipStream_t st = (hipStream_t)hip_stream;
    hipStream_t own = nullptr;
    if (!st) { if (hipStreamCreateWithFlags(&own, hipStreamNonBlocking) != hipSuccess) return fail(TGSF_E_HIP, "stream"); st = own; }
    unsigned long long* buf = nullptr;
    const size_t words = (size_t)n + 4u * (size_t)world + 2;
    if (hipMalloc((void**)&buf, words * 8) != hipSuccess) { if (own) (void)hipStreamDestroy(own); return fail(TGSF_E_HIP, "device allocation failed"); }
    int rc = TGSF_OK;
    if (check_layout) {                                  // max(n) == -max(-n) on every rank, or nobody sums anything
        long long h[2] = {(long long)n, -(long long)n};
        long long* d2 = (long long*)(buf + n + 4u * (size_t)world);
        (void)hipMemcpyAsync(d2, h, 16, hipMemcpyHostToDevice, st);
        ncclResult_t r = ncclAllReduce(d2, d2, 2, ncclInt64, ncclMax, comm, st);
        (void)hipMemcpyAsync(h, d2, 16, hipMemcpyDeviceToHost, st);
        if (r != ncclSuccess || hipStreamSynchronize(st) != hipSuccess) rc = fail(TGSF_E_HIP, "layout check: %s", ncclGetErrorString(r));
        else if (h[0] != -h[1]) rc = fail(TGSF_E_INVALID, "tally vectors differ in length across ranks (%llu words here, %lld..%lld over the job): create every context with the same bc_len and max_read_len",
                                          (unsigned long long)n, -h[1], h[0]);
    }
    if (rc == TGSF_OK) {
        const unsigned T = 256, G = (unsigned)((n + 4u * (size_t)world + T - 1) / T);
        hipLaunchKernelGGL(k_pack_rows, dim3(G), dim3(T), 0, st, buf, (const unsigned long long*)d_ctr, (unsigned long long)n, rank, world);
        ncclResult_t r = ncclAllReduce(buf, buf, (size_t)n + 4u * (size_t)world, ncclUint64, ncclSum, comm, st);   // the job's one collective
        hipLaunchKernelGGL(k_unpack_rows, dim3(G), dim3(T), 0, st, (unsigned long long*)d_ctr, (const unsigned long long*)buf, (unsigned long long)n, world);
        if (r != ncclSuccess) rc = fail(TGSF_E_HIP, "ncclAllReduce: %s", ncclGetErrorString(r));
        else if (hipStreamSynchronize(st) != hipSuccess) rc = fail(TGSF_E_HIP, "stream synchronize failed");
    }
    (void)hipFree(buf);
    if (own) (void)hipStreamDestroy(own);
    return rc;
}
