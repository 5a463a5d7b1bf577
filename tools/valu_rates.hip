// valu_rates.hip -- measured issue rates of the integer VALU ops the Myers scan is made of (gfx950).
// hipcc --offload-arch=gfx950 -O3 tools/valu_rates.hip -o tools/valu_rates && ./tools/valu_rates
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define REP 256
#define ITERS 2000
#define STR2(x) #x
#define STR(x) STR2(x)
#define BENCH(NAME, ASM)                                                             \
    __global__ void NAME(uint32_t* out) {                                            \
        uint32_t a = threadIdx.x, b = a * 3 + 1, c = a ^ 5, d = a + 7;               \
        uint64_t x = a, y = b; uint32_t s = 0;                                       \
        for (int i = 0; i < ITERS; i++) {                                            \
            asm volatile(".rept " STR(REP) "\n" ASM "\n.endr\n"                      \
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(x), "+v"(y), "+v"(s) :: "vcc");  \
        }                                                                            \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + (uint32_t)x + (uint32_t)y + s; \
    }
// two independent chains per op so that dependent-issue latency is not the limit with >= 2 waves/SIMD
BENCH(k_and, "v_and_b32 %0, %0, %2\n v_and_b32 %1, %1, %3")
BENCH(k_bfi, "v_bfi_b32 %0, %2, %0, -1\n v_bfi_b32 %1, %3, %1, -1")
BENCH(k_add3, "v_add3_u32 %0, %0, %2, %3\n v_add3_u32 %1, %1, %2, %3")
BENCH(k_lshl_add_u64, "v_lshl_add_u64 %4, %4, 0, %5\n v_lshl_add_u64 %5, %5, 0, %4")
BENCH(k_lshlrev_b64, "v_lshlrev_b64 %4, 1, %4\n v_lshlrev_b64 %5, 1, %5")
BENCH(k_addco, "v_add_co_u32 %0, vcc, %0, %2\n v_addc_co_u32 %1, vcc, %1, %3, vcc")
BENCH(k_alignbit, "v_alignbit_b32 %0, %0, %2, 31\n v_alignbit_b32 %1, %1, %3, 31")
BENCH(k_sdwa, "v_lshlrev_b32_sdwa %0, %2, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n v_lshlrev_b32_sdwa %1, %3, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2")
BENCH(k_bfe, "v_bfe_u32 %0, %0, 8, 8\n v_bfe_u32 %1, %1, 16, 8")
BENCH(k_or3, "v_or3_b32 %0, %0, %2, %3\n v_or3_b32 %1, %1, %2, %3")
BENCH(k_andor, "v_and_or_b32 %0, %0, %2, %3\n v_and_or_b32 %1, %1, %2, %3")
BENCH(k_xor, "v_xor_b32 %0, %0, %2\n v_xor_b32 %1, %1, %3")
BENCH(k_dot4, "v_dot4_u32_u8 %0, %2, %3, %0\n v_dot4_u32_u8 %1, %2, %3, %1")
BENCH(k_bcnt, "v_bcnt_u32_b32 %0, %2, %0\n v_bcnt_u32_b32 %1, %3, %1")
BENCH(k_pk_add_u16, "v_pk_add_u16 %0, %0, %2\n v_pk_add_u16 %1, %1, %3")
BENCH(k_mov, "v_mov_b32 %0, %2\n v_mov_b32 %1, %3")
BENCH(k_min, "v_min_i32 %0, %0, %2\n v_min_i32 %1, %1, %3")
BENCH(k_cmp, "v_cmp_le_i32 vcc, %0, %2\n v_cmp_le_i32 vcc, %1, %3")


BENCH(k_or, "v_or_b32 %0, %0, %2\n v_or_b32 %1, %1, %3")
BENCH(k_not, "v_not_b32 %0, %0\n v_not_b32 %1, %1")
BENCH(k_xnor, "v_xnor_b32 %0, %0, %2\n v_xnor_b32 %1, %1, %3")
BENCH(k_add_u32, "v_add_u32 %0, %0, %2\n v_add_u32 %1, %1, %3")
BENCH(k_sub_u32, "v_sub_u32 %0, %0, %2\n v_sub_u32 %1, %1, %3")
BENCH(k_lshl32, "v_lshlrev_b32 %0, 1, %0\n v_lshlrev_b32 %1, 1, %1")
BENCH(k_lshr32, "v_lshrrev_b32 %0, 31, %0\n v_lshrrev_b32 %1, 31, %1")
BENCH(k_ashr32, "v_ashrrev_i32 %0, 31, %0\n v_ashrrev_i32 %1, 31, %1")
BENCH(k_cndmask, "v_cndmask_b32 %0, %0, %2, vcc\n v_cndmask_b32 %1, %1, %3, vcc")
BENCH(k_and_lit, "v_and_b32 %0, 0x7f7f7f7f, %0\n v_and_b32 %1, 0x7f7f7f7f, %1")
BENCH(k_add_f32, "v_add_f32 %0, %0, %2\n v_add_f32 %1, %1, %3")
BENCH(k_fma_f32, "v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3")
BENCH(k_pk_fma_f32, "v_pk_fma_f32 %4, %4, %5, %5\n v_pk_fma_f32 %5, %5, %4, %4")
BENCH(k_perm, "v_perm_b32 %0, %0, %2, %3\n v_perm_b32 %1, %1, %2, %3")
BENCH(k_lshl_or, "v_lshl_or_b32 %0, %0, 1, %2\n v_lshl_or_b32 %1, %1, 1, %3")
BENCH(k_max_u32, "v_max_u32 %0, %0, %2\n v_max_u32 %1, %1, %3")
BENCH(k_mul_u24, "v_mul_u32_u24 %0, %0, %2\n v_mul_u32_u24 %1, %1, %3")
BENCH(k_and_dpp, "v_and_b32_dpp %0, %0, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_and_b32_dpp %1, %1, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
BENCH(k_pk_mov, "v_pk_mov_b32 %4, %4, %5\n v_pk_mov_b32 %5, %5, %4")
BENCH(k_and_e64, "v_and_b32_e64 %0, %0, %2\n v_and_b32_e64 %1, %1, %3")
BENCH(k_mov_b64, "v_mov_b64 %4, %5\n v_mov_b64 %5, %4")
BENCH(k_pk_add_f32, "v_pk_add_f32 %4, %4, %5\n v_pk_add_f32 %5, %5, %4")
BENCH(k_and_or_mix, "v_and_b32 %0, %0, %2\n v_add3_u32 %1, %1, %2, %3")
BENCH(k_sad_u8, "v_sad_u8 %0, %2, %3, %0\n v_sad_u8 %1, %2, %3, %1")

template <class K> void run(const char* name, K k, uint32_t* d, int waves_per_simd) {
    int cus = 256;
    dim3 grid(cus * waves_per_simd), block(256);   // 4 waves per block = 1 per SIMD
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, grid, block, 0, 0, d);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, grid, block, 0, 0, d);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double instr_per_wave = 2.0 * REP * ITERS;
    double waves_per_simd_d = waves_per_simd;
    // cycles per wave-instruction per SIMD at 2.4 GHz nominal
    double cyc = ms * 1e-3 * 2.4e9 / (instr_per_wave * waves_per_simd_d);
    printf("%-16s waves/SIMD=%d  %.3f ms  -> %.2f cycles per wave-instruction per SIMD (at 2.4 GHz nominal)\n", name, waves_per_simd, ms, cyc);
}
int main() {
    uint32_t* d; hipMalloc(&d, 256 * 8 * 256 * 4);
#define R(k) run(#k, k, d, 2); run(#k, k, d, 4);
    R(k_or) R(k_not) R(k_xnor) R(k_add_u32) R(k_sub_u32) R(k_lshl32) R(k_lshr32) R(k_ashr32) R(k_cndmask) R(k_and_lit) R(k_add_f32) R(k_fma_f32) R(k_pk_fma_f32) R(k_perm) R(k_lshl_or) R(k_max_u32) R(k_mul_u24) R(k_and_dpp) R(k_pk_mov) R(k_and_e64) R(k_mov_b64) R(k_pk_add_f32) R(k_and_or_mix) R(k_sad_u8)
    R(k_and) R(k_xor) R(k_bfi) R(k_add3) R(k_or3) R(k_andor) R(k_lshl_add_u64) R(k_lshlrev_b64) R(k_addco) R(k_alignbit)
    R(k_sdwa) R(k_bfe) R(k_dot4) R(k_bcnt) R(k_pk_add_u16) R(k_mov) R(k_min) R(k_cmp)
    return 0;
}
