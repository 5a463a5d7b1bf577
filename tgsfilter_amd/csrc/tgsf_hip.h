// tgsf_hip.h -- the device primitives of the kernels as they are on gfx950 (wave64): intrinsics, wave-level
// operations, the execution-model vocabulary.  The product (libtgsf.so) is built from this file; the serial CPU
// emulation of tests/emul takes its stand-ins from tgsf_emul.h instead (test infrastructure, never in the product).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define TGSF_HD __host__ __device__ __forceinline__
#define TGSF_D __device__ __forceinline__
#define TGSF_KERNEL __global__ void
#define TGSF_INLINE_LAMBDA __attribute__((always_inline))
#define TGSF_BOUNDS(threads, waves_per_simd) __launch_bounds__(threads, waves_per_simd)
#define TGSF_SHARED __shared__
#define TGSF_BLOCK_SYNC() __syncthreads()
#define TGSF_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)
// cooperative loops of a workgroup / of a wave
#define TGSF_COOP_BEGIN threadIdx.x
#define TGSF_COOP_STRIDE blockDim.x
#define TGSF_WCOOP_BEGIN(lane) (lane)
#define TGSF_WCOOP_STRIDE 64u
// statements that exist on the device only / in the emulation only (register staging, prefetches; traces)
#define TGSF_ON_DEVICE(...) __VA_ARGS__
#define TGSF_ON_EMUL(...)
constexpr bool kTgsfEmul = false;
// issue priority of the wave from here on (0..3; priority outranks age in the SIMD's arbitration)
#define TGSF_WAVE_PRIO(p) __builtin_amdgcn_s_setprio(p)

namespace tgsf {

// ---- integer intrinsics ----
TGSF_HD uint32_t popc32(uint32_t x) { return (uint32_t)__builtin_popcount(x); }
TGSF_HD uint32_t popc64(uint64_t x) { return (uint32_t)__builtin_popcountll(x); }
// sum_i a.byte[i] * b.byte[i] + c
TGSF_D uint32_t udot4(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_udot4(a, b, c, false); }
// bytes [sh, sh+4) of the 8-byte value hi:lo, sh in 0..3
TGSF_D uint32_t alignbyte(uint32_t hi, uint32_t lo, uint32_t sh) { return __builtin_amdgcn_alignbyte(hi, lo, sh); }
// bits [sh, sh+32) of the 64-bit value hi:lo, sh in 0..31
TGSF_D uint32_t alignbit(uint32_t hi, uint32_t lo, uint32_t sh) { return __builtin_amdgcn_alignbit(hi, lo, sh); }
// byte i of the result = byte sel.byte[i] of the 8-byte value hi:lo (selectors 0..7 only)
TGSF_D uint32_t perm_bytes(uint32_t hi, uint32_t lo, uint32_t sel) { return __builtin_amdgcn_perm(hi, lo, sel); }
// any boolean function of three words in one instruction: bit i of the result is TT[a_i b_i c_i],
// TT = f(0xF0, 0xCC, 0xAA) (v_bitop3_b32)
template <int TT>
TGSF_D uint32_t bitop3(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, TT); }
// keeps a value in the registers it is in (no instruction): stops the compiler from re-deriving or regrouping it
TGSF_D void pin(uint64_t& v) { asm volatile("" : "+v"(v)); }
TGSF_D void pin(uint32_t& v) { asm volatile("" : "+v"(v)); }
// The class constants of the QC tallies held in VGPRs, so that (x7 ^ K) + 0x7F7F7F7F is one v_xad_u32
// (a VOP3 instruction reads one scalar at most): 4 instructions per class and dword.
struct QcConsts { uint32_t k[4]; };
TGSF_D QcConsts qc_consts() {
    QcConsts c;
    asm volatile("v_mov_b32 %0, 0x41414141" : "=v"(c.k[0]));
    asm volatile("v_mov_b32 %0, 0x54545454" : "=v"(c.k[1]));
    asm volatile("v_mov_b32 %0, 0x47474747" : "=v"(c.k[2]));
    asm volatile("v_mov_b32 %0, 0x43434343" : "=v"(c.k[3]));
    return c;
}
// ((x7 ^ k) + 0x7F7F7F7F): bit 7 of each byte set where the byte of x7 differs from the byte of k
TGSF_D uint32_t xad7f(uint32_t x7, uint32_t k) {
    uint32_t t;
    const uint32_t c7f = 0x7F7F7F7Fu;
    asm("v_xad_u32 %0, %1, %2, %3" : "=v"(t) : "v"(x7), "v"(k), "s"(c7f));
    return t;
}
// 16 bytes at any alignment (one global_load_dwordx4)
struct __attribute__((packed, aligned(1))) U4u { uint32_t x, y, z, w; };
TGSF_D uint4 load16u(const uint8_t* p) { const U4u* q = reinterpret_cast<const U4u*>(p); return make_uint4(q->x, q->y, q->z, q->w); }

// ---- wave-level operations ----
TGSF_D uint64_t wave_sum(uint64_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// The sum over the wave's 64 lanes, in every lane.  Six DPP adds (VALU, no LDS round trips: __shfl_xor is a ds_bpermute, and six
// of those in a row sat in front of every tile of k_stats, twice for a speculated read): within quads, within rows of 16, then
// row 0 into row 1 and row 2 into row 3 (row_bcast:15), rows 0-1 into rows 2-3 (row_bcast:31); lane 63 holds the total.
TGSF_D int32_t wave_sum_i32(int32_t v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, false);     // quad_perm:[1,0,3,2]
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, false);     // quad_perm:[2,3,0,1]
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, false);    // row_half_mirror
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, false);    // row_mirror: every lane of a row holds the row's sum
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);    // row_bcast:15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);    // row_bcast:31 into rows 2 and 3
    return __builtin_amdgcn_readlane(v, 63);
}
TGSF_D uint32_t wave_max(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { uint32_t u = __shfl_xor(v, o, 64); v = u > v ? u : v; }
    return v;
}
TGSF_D uint32_t wave_or(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v |= __shfl_xor(v, o, 64);
    return v;
}
TGSF_D bool wave_leader() { return (threadIdx.x & 63u) == 0u; }
TGSF_D bool wave_any(bool b) { return __builtin_amdgcn_ballot_w64(b) != 0ull; }
TGSF_D uint32_t wave_bcast(uint32_t v, uint32_t src_lane) { return (uint32_t)__shfl((int)v, (int)src_lane, 64); }
// lane src_lane's value as a wave-uniform (scalar) value; src_lane must be wave-uniform
TGSF_D uint32_t wave_pick(uint32_t v, uint32_t src_lane) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)src_lane); }

}  // namespace tgsf
