// tgsf_emul_rt.h -- TEST INFRASTRUCTURE: the runtime layer of tgsf_lib.hip for the serial CPU emulation (tests/emul):
// host memory for device memory, kernels run lane by lane on the calling thread.  Never part of the product.
#pragma once
static int rt_malloc(void** p, size_t n) { *p = calloc(n ? n : 1, 1); return *p ? 0 : 1; }
static void rt_free(void* p) { free(p); }
static int rt_memset(void* p, int v, size_t n, rt_stream) { memset(p, v, n); return 0; }
static int rt_h2d(void* d, const void* s, size_t n, rt_stream) { memcpy(d, s, n); return 0; }
static int rt_d2h(void* d, const void* s, size_t n, rt_stream) { memcpy(d, s, n); return 0; }
static int rt_sync(rt_stream) { return 0; }
static const char* rt_errstr(int) { return "emulation error"; }
template <class F>
static void emul_launch(unsigned grid, unsigned block, F f)
{
    using namespace tgsf_emul;
    gridDim = {grid, 1, 1}; blockDim = {block, 1, 1};
    // TGSF_EMUL_ORDER=reverse: lanes run last to first -- what they append to shared lists (the middle scan's candidates)
    // then arrives in descending order, as unlike the usual order as a GPU's may be: results must not depend on it
    const char* ord = getenv("TGSF_EMUL_ORDER");
    const bool reverse = ord && !strcmp(ord, "reverse");
    for (unsigned b0 = 0; b0 < grid; b0++)
        for (unsigned t0 = 0; t0 < block; t0++) {
            const unsigned b = reverse ? grid - 1 - b0 : b0, t = reverse ? block - 1 - t0 : t0;
            blockIdx = {b, 0, 0}; threadIdx = {t, 0, 0}; f();
        }
}
#define TGSF_LAUNCH(kernel, grid, block, stream, ...) emul_launch((grid), (block), [&] { kernel(__VA_ARGS__); })
#define TGSF_LAUNCH_LDS(kernel, grid, block, lds, stream, ...) emul_launch((grid), (block), [&] { (void)(lds); kernel(__VA_ARGS__); })
// block-cooperative kernels are written for any block size; emulate them with one thread
#define TGSF_LAUNCH_COOP(kernel, grid, block, stream, ...) emul_launch((grid), 1u, [&] { kernel(__VA_ARGS__); })
// the emulation trades speed for fidelity: small grids
static unsigned grid_cap(unsigned g) { return g > 8u ? 8u : g; }
