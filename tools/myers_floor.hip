// myers_floor.hip -- the arithmetic floor of the middle scan: nothing but the 17-instruction Myers column
// (hot_step of tgsfilter_amd/csrc/tgsf_core.h) for two adapters per lane, Eq words taken from registers
// (no LDS, no global memory, no candidate test).  Prints lane-columns per second for the whole chip; the
// kernel's own figure (columns it scans / its duration) is to be read against this.
//   hipcc -O3 --offload-arch=gfx950 -I tgsfilter_amd/csrc tools/myers_floor.hip -o /tmp/myers_floor && /tmp/myers_floor
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "tgsf_core.h"
using namespace tgsf;

__global__ void __launch_bounds__(256) k_floor(uint64_t* out, const uint64_t* eqs, int iters)
{
    Hot a, b;
    hot_init(a, 50); hot_init(b, 50);
    uint64_t e0 = eqs[threadIdx.x & 3], e1 = eqs[(threadIdx.x >> 2) & 3], e2 = eqs[4], e3 = eqs[5];
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            hot_step(a, e0); hot_step(b, e1);
            hot_step(a, e2); hot_step(b, e3);
            hot_step(a, e1); hot_step(b, e2);
            hot_step(a, e3); hot_step(b, e0);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a.p ^ a.m ^ b.p ^ b.m;
}

// variant: Eq rows fetched from LDS by a byte of a register-resident dword, as the scan does (no global traffic)
__global__ void __launch_bounds__(256) k_floor_lds(uint64_t* out, const uint64_t* eqs, int iters)
{
    __shared__ uint64_t eqt[256][2];
    for (int i = threadIdx.x; i < 512; i += 256) eqt[i >> 1][i & 1] = eqs[(i * 7) % 6] ^ (uint64_t)i * 0x9e3779b97f4a7c15ull;
    __syncthreads();
    Hot a, b;
    hot_init(a, 50); hot_init(b, 50);
    uint32_t d = (uint32_t)(eqs[threadIdx.x & 3] >> 7) | 0x41434754u;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            hot_step(a, eqt[d & 0xFF][0]); hot_step(b, eqt[d & 0xFF][1]);
            hot_step(a, eqt[(d >> 8) & 0xFF][0]); hot_step(b, eqt[(d >> 8) & 0xFF][1]);
            hot_step(a, eqt[(d >> 16) & 0xFF][0]); hot_step(b, eqt[(d >> 16) & 0xFF][1]);
            hot_step(a, eqt[d >> 24][0]); hot_step(b, eqt[d >> 24][1]);
            d = d * 1664525u + 1013904223u;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a.p ^ a.m ^ b.p ^ b.m;
}

int main()
{
    const int blocks = 256 * 6 * 4, threads = 256, iters = 4096;        // 16 columns per iteration
    uint64_t h[6] = {0x123456789abcdef0ull, 0x0f0f00ff00ff0f0full, 0x8000000000000001ull, 0x5555aaaa5555aaaaull,
                     0x00000000ffffffffull, 0xdeadbeefcafef00dull};
    uint64_t *d_out, *d_eq;
    hipMalloc(&d_out, (size_t)blocks * threads * 8);
    hipMalloc(&d_eq, sizeof h);
    hipMemcpy(d_eq, h, sizeof h, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_floor, dim3(blocks), dim3(threads), 0, 0, d_out, d_eq, 64);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 5; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_floor, dim3(blocks), dim3(threads), 0, 0, d_out, d_eq, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    float best2 = 1e30f;
    for (int rep = 0; rep < 5; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_floor_lds, dim3(blocks), dim3(threads), 0, 0, d_out, d_eq, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best2) best2 = ms;
    }
    const double cols = (double)blocks * threads * iters * 16.0;          // lane-columns, two adapters each
    printf("same with the Eq rows read from LDS by a byte of a register: %.3f ms -> %.3e lane-columns/s\n", best2, cols / (best2 * 1e-3));
    printf("pure Myers column, 2 adapters per lane: %.3f ms for %.3e lane-columns -> %.3e lane-columns/s (%.2f ns per 1e3)\n",
           best, cols, cols / (best * 1e-3), best * 1e6 / (cols / 1e3));
    printf("at 34 VALU instructions per lane-column, 4 cycles each, 1024 SIMDs x 64 lanes: implied clock %.2f GHz\n",
           cols / (best * 1e-3) * 34.0 * 4.0 / (1024.0 * 64.0) / 1e9);
    return 0;
}
