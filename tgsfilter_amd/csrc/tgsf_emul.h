// tgsf_emul.h -- TEST INFRASTRUCTURE: stand-ins for tgsf_hip.h that let the kernel sources compile as plain C++
// (g++ -DTGSF_EMUL, tests/emul): a serial emulation, one lane at a time, in which every lane is a wave of its own.
// Never part of the product: libtgsf.so is built from tgsf_hip.h and has no CPU path.
#pragma once
#include <stdint.h>
#include <string.h>

#define TGSF_HD inline
#define TGSF_D inline
#define TGSF_KERNEL static void
#define TGSF_INLINE_LAMBDA
#define TGSF_BOUNDS(threads, waves_per_simd)
#define TGSF_SHARED static thread_local      /* contexts may run on several host threads at once */
#define TGSF_BLOCK_SYNC() ((void)0)
#define TGSF_WAVE_SYNC() ((void)0)
// cooperative loops: every emulated thread performs all iterations (idempotent fills)
#define TGSF_COOP_BEGIN 0u
#define TGSF_COOP_STRIDE 1u
#define TGSF_WCOOP_BEGIN(lane) 0u
#define TGSF_WCOOP_STRIDE 1u
#define TGSF_ON_DEVICE(...)
#define TGSF_ON_EMUL(...) __VA_ARGS__
constexpr bool kTgsfEmul = true;
#define TGSF_WAVE_PRIO(p) ((void)0)

struct alignas(16) uint4 { uint32_t x, y, z, w; };      // (HIP's uint4 is 16-byte aligned: LDS arrays of it are reinterpreted as 64-bit words)
namespace tgsf_emul {
struct Dim3 { unsigned x, y, z; };
extern thread_local Dim3 threadIdx, blockIdx, blockDim, gridDim;
}
using tgsf_emul::threadIdx; using tgsf_emul::blockIdx; using tgsf_emul::blockDim; using tgsf_emul::gridDim;
template <class T> static inline T atomicAdd(T* p, T v) { T o = *p; *p = o + v; return o; }
template <class T> static inline T atomicMax(T* p, T v) { T o = *p; if (v > o) *p = v; return o; }
template <class T> static inline T atomicMin(T* p, T v) { T o = *p; if (v < o) *p = v; return o; }
template <class T> static inline T atomicExch(T* p, T v) { T o = *p; *p = v; return o; }
template <class T> static inline T atomicOr(T* p, T v) { T o = *p; *p = o | v; return o; }
template <class T> static inline T atomicCAS(T* p, T cmp, T v) { T o = *p; if (o == cmp) *p = v; return o; }

namespace tgsf {

TGSF_HD uint32_t popc32(uint32_t x) { return (uint32_t)__builtin_popcount(x); }
TGSF_HD uint32_t popc64(uint64_t x) { return (uint32_t)__builtin_popcountll(x); }
TGSF_HD uint32_t udot4(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t s = c;
    for (int i = 0; i < 4; i++) s += ((a >> (8 * i)) & 0xFF) * ((b >> (8 * i)) & 0xFF);
    return s;
}
TGSF_HD uint32_t alignbyte(uint32_t hi, uint32_t lo, uint32_t sh) { return (uint32_t)((((uint64_t)hi << 32) | lo) >> (8 * sh)); }
TGSF_HD uint32_t alignbit(uint32_t hi, uint32_t lo, uint32_t sh) { return (uint32_t)((((uint64_t)hi << 32) | lo) >> (sh & 31u)); }
TGSF_HD uint32_t perm_bytes(uint32_t hi, uint32_t lo, uint32_t sel) {
    uint64_t v = ((uint64_t)hi << 32) | lo;
    uint32_t r = 0;
    for (int i = 0; i < 4; i++) r |= (uint32_t)((v >> (8 * ((sel >> (8 * i)) & 7u))) & 0xFF) << (8 * i);
    return r;
}
template <int TT>
TGSF_HD uint32_t bitop3(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r = 0;
    for (int i = 0; i < 32; i++) {
        const int idx = (int)(((a >> i) & 1u) << 2 | ((b >> i) & 1u) << 1 | ((c >> i) & 1u));
        r |= (uint32_t)((TT >> idx) & 1) << i;
    }
    return r;
}
TGSF_HD void pin(uint64_t&) {}
TGSF_HD void pin(uint32_t&) {}
struct QcConsts { uint32_t k[4]; };
TGSF_HD QcConsts qc_consts() { return QcConsts{{0x41414141u, 0x54545454u, 0x47474747u, 0x43434343u}}; }
TGSF_HD uint32_t xad7f(uint32_t x7, uint32_t k) { return (x7 ^ k) + 0x7F7F7F7Fu; }
TGSF_HD uint4 load16u(const uint8_t* p) { uint4 v; memcpy(&v, p, 16); return v; }

// one lane at a time: every lane is the leader of a wave of one
TGSF_HD uint64_t wave_sum(uint64_t v) { return v; }
TGSF_HD int32_t wave_sum_i32(int32_t v) { return v; }
TGSF_HD uint32_t wave_max(uint32_t v) { return v; }
TGSF_HD uint32_t wave_or(uint32_t v) { return v; }
TGSF_HD bool wave_leader() { return true; }
TGSF_HD bool wave_any(bool b) { return b; }
TGSF_HD uint32_t wave_bcast(uint32_t v, uint32_t) { return v; }
TGSF_HD uint32_t wave_pick(uint32_t v, uint32_t) { return v; }

}  // namespace tgsf
