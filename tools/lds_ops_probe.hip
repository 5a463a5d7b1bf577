// lds_ops_probe.hip -- what a random LDS update costs on gfx950, per wave-instruction and per CU:
// no-return ds_or, returning ds_or, plain dword / byte stores, read-modify-write; all 64 lanes or a quarter of
// them active; one workgroup per CU holding 128 KB (the repeat gate's shape) with 4, 8 or 16 waves.
//   hipcc -O3 --offload-arch=gfx950 tools/lds_ops_probe.hip -o tools/lds_ops_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr uint32_t kWords = 32768;   // 128 KB

enum Op { OR_NORET = 0, OR_RET = 1, ST32 = 2, ST8 = 3, RMW = 4, OR_QUARTER = 5, ST8_QUARTER = 6, OR_NORET_DRAIN16 = 7, MAX_NORET = 8 };

template <int OP>
__global__ void k_probe(uint32_t iters, uint32_t* out)
{
    __shared__ uint32_t bm[kWords];
    for (uint32_t w = threadIdx.x; w < kWords; w += blockDim.x) bm[w] = 0;
    __syncthreads();
    uint32_t x = (blockIdx.x * 1024u + threadIdx.x) * 2654435761u + 12345u;
    uint32_t acc = 0;
    for (uint32_t i = 0; i < iters; i++) {
        x = x * 1664525u + 1013904223u;
        const uint32_t idx = x >> 12;                    // 20 bits
        const uint32_t word = idx >> 5, bit = 1u << (idx & 31u);
        if (OP == OR_NORET) atomicOr(&bm[word], bit);
        else if (OP == MAX_NORET) atomicMax(&bm[word], bit);
        else if (OP == OR_RET) acc += atomicOr(&bm[word], bit);
        else if (OP == ST32) bm[word] = bit;
        else if (OP == ST8) reinterpret_cast<uint8_t*>(bm)[idx >> 3] = 1;
        else if (OP == RMW) { bm[word] |= bit; }
        else if (OP == OR_QUARTER) { if (((x >> 8) & 3u) == (i & 3u)) atomicOr(&bm[word], bit); }
        else if (OP == ST8_QUARTER) { if (((x >> 8) & 3u) == (i & 3u)) reinterpret_cast<uint8_t*>(bm)[idx >> 3] = 1; }
        else if (OP == OR_NORET_DRAIN16) {
            atomicOr(&bm[word], bit);
            if ((i & 15u) == 15u) acc += bm[(x >> 3) & (kWords - 1)];      // a dependent read every 16 updates
        }
    }
    __syncthreads();
    uint32_t s = acc;
    for (uint32_t w = threadIdx.x; w < kWords; w += blockDim.x) s += __popc(bm[w]);
    atomicAdd(out, s);
}

template <int OP>
static int run(const char* name, int threads, uint32_t* d_out, double clock_ghz)
{
    const uint32_t total_per_cu = 1u << 22;               // lane-updates per CU
    const uint32_t iters = total_per_cu / (uint32_t)threads;
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL(k_probe<OP>, dim3(256), dim3(threads), 0, 0, iters, d_out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(k_probe<OP>, dim3(256), dim3(threads), 0, 0, iters, d_out);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    const double cycles = ms * 1e-3 * clock_ghz * 1e9;
    const double wave_instr = (double)iters * (threads / 64);
    printf("%-18s threads %4d  %8.3f ms  %7.1f cycles per wave-instruction on the CU  %6.2f cycles per lane-update\n",
           name, threads, ms, cycles / wave_instr, cycles / ((double)iters * threads));
    return 0;
}

int main()
{
    uint32_t* d_out;
    CK(hipMalloc(&d_out, 4));
    CK(hipMemset(d_out, 0, 4));
    int khz = 0;
    CK(hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0));
    const double ghz = khz * 1e-6;
    printf("clock %.2f GHz\n", ghz);
    for (int threads : {256, 512, 1024}) {
        run<OR_NORET>("ds_or no return", threads, d_out, ghz);
        run<MAX_NORET>("ds_max no return", threads, d_out, ghz);
        run<OR_RET>("ds_or returning", threads, d_out, ghz);
        run<ST32>("store b32", threads, d_out, ghz);
        run<ST8>("store b8", threads, d_out, ghz);
        run<RMW>("read-or-write", threads, d_out, ghz);
        run<OR_QUARTER>("ds_or 1/4 lanes", threads, d_out, ghz);
        run<ST8_QUARTER>("store b8 1/4 lanes", threads, d_out, ghz);
        run<OR_NORET_DRAIN16>("ds_or + read /16", threads, d_out, ghz);
    }
    return 0;
}
