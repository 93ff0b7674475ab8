// Internal helpers shared by the HIP translation units of libgcl_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/gcl_amd.h"

namespace gcl {

void set_error(const char* fmt, ...);

#define GCL_CHECK_ARG(cond, ...)                 \
  do {                                           \
    if (!(cond)) {                               \
      gcl::set_error(__VA_ARGS__);               \
      return GCL_ERR_ARG;                        \
    }                                            \
  } while (0)

#define GCL_CHECK_HIP(expr)                                                          \
  do {                                                                               \
    hipError_t e_ = (expr);                                                          \
    if (e_ != hipSuccess) {                                                          \
      gcl::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
      return GCL_ERR_HIP;                                                            \
    }                                                                                \
  } while (0)

#define GCL_CHECK_LAUNCH() GCL_CHECK_HIP(hipGetLastError())

static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---- coordinate keys / hash table -----------------------------------------------------------------
constexpr unsigned long long EMPTY_KEY = ~0ull;
constexpr int COORD_OFF = 1 << 15;

struct Slot {
  unsigned long long key;
  long long val;   // row index (kept 64-bit so that a slot is one aligned 16-byte access)
};

__device__ __forceinline__ bool pack_ok(int b, int x, int y, int z) {
  return (unsigned)b < 65535u && (unsigned)(x + COORD_OFF) < 65536u && (unsigned)(y + COORD_OFF) < 65536u &&
         (unsigned)(z + COORD_OFF) < 65536u;
}
__device__ __forceinline__ unsigned long long pack_key(int b, int x, int y, int z) {
  return ((unsigned long long)(unsigned)b << 48) | ((unsigned long long)(unsigned)(x + COORD_OFF) << 32) |
         ((unsigned long long)(unsigned)(y + COORD_OFF) << 16) | (unsigned long long)(unsigned)(z + COORD_OFF);
}
__device__ __forceinline__ unsigned long long mix64(unsigned long long h) {
  h ^= h >> 33;
  h *= 0xff51afd7ed558ccdull;
  h ^= h >> 33;
  h *= 0xc4ceb9fe1a85ec53ull;
  h ^= h >> 33;
  return h;
}
// returns the slot that holds `key` after the call (inserting it if absent)
__device__ __forceinline__ long long table_insert(Slot* t, long long cap, unsigned long long key) {
  long long s = (long long)(mix64(key) & (unsigned long long)(cap - 1));
  while (true) {
    unsigned long long prev = atomicCAS(&t[s].key, EMPTY_KEY, key);
    if (prev == EMPTY_KEY || prev == key) return s;
    s = (s + 1) & (cap - 1);
  }
}
__device__ __forceinline__ long long table_find(const Slot* t, long long cap, unsigned long long key) {
  long long s = (long long)(mix64(key) & (unsigned long long)(cap - 1));
  while (true) {
    unsigned long long k = t[s].key;
    if (k == key) return s;
    if (k == EMPTY_KEY) return -1;
    s = (s + 1) & (cap - 1);
  }
}

// ---- wave helpers ---------------------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// ---- max|x| slots ("gcl_amax") --------------------------------------------------------------------------------
// A slot is GCL_AMAX_WORDS = 512 int32: 16 entries on separate 128-byte lines (entry i at word 32 i); the value is the
// maximum over the entries (bit patterns of non-negative floats order like ints).  Same-line atomics serialise at
// ~12 ns each on gfx950 (tools/micro/atomic_contention.hip: 4096 workgroups on one line 50 us, on 16 lines 1 us),
// so the workgroups of a producer spread over the 16 lines and skip the atomic when the line already covers them.
constexpr int AMAX_ENTRIES = 16, AMAX_STRIDE = 32, AMAX_WORDS = AMAX_ENTRIES * AMAX_STRIDE;
__device__ __forceinline__ int amax_slot_bits(const int* __restrict__ slot) {
  int m = 0;
#pragma unroll
  for (int i = 0; i < AMAX_ENTRIES; ++i) m = max(m, slot[i * AMAX_STRIDE]);
  return m;
}
__device__ __forceinline__ void amax_slot_publish(int* slot, int bits, unsigned wg) {
  int* p = slot + (wg & (AMAX_ENTRIES - 1)) * AMAX_STRIDE;
  if (bits > __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(p, bits);
}


// ---- fp16x3 operand format (csrc/conv.hip "fp16x3, range-extended") ----------------------------------------------------
// power-of-two scale that maps amax into [2^13, 2^14)
__device__ __forceinline__ float amax_scale(const int* __restrict__ amax_bits) {
  float amax = __int_as_float(amax_slot_bits(amax_bits));
  if (!(amax > 0.f) || !(amax < 3.0e38f)) return 1.f;
  int e;
  frexpf(amax, &e);   // amax = m * 2^e, m in [0.5, 1)
  int sh = 14 - e;
  sh = sh > 100 ? 100 : (sh < -100 ? -100 : sh);   // keep scale and 1/scale finite for degenerate tensors
  return ldexpf(1.f, sh);
}


constexpr float F16_LO_UP = 2048.f, F16_LO_DOWN = 1.f / 2048.f;
__device__ __forceinline__ void split_f16(float sv, _Float16& hi, _Float16& lo) {
  hi = (_Float16)sv;
  // (sv - hi) 2^11, exactly (the difference of a value and its fp16 rounding is exact in fp32, so is every scaling by 2^11):
  // written as one fused multiply-add with the fp16 value as a source (v_fma_mix_f32: no conversion back to fp32)
  lo = (_Float16)__builtin_fmaf((float)hi, -F16_LO_UP, sv * F16_LO_UP);
}


// plane image of an [n, ldc] fp32 tensor: per row and 32-channel slice 64 bytes of hi followed by 64 bytes of lo' (16-bit
// units: row * ldc * 2 + slice * 64 + plane * 32 + channel % 32).  Four consecutive channels col .. col + 3 of one row:
__device__ __forceinline__ void store_planes4(unsigned short* __restrict__ planes, long long row, long long ldc, int col,
                                              const float4& v, float s) {
  typedef _Float16 f16x4_ __attribute__((ext_vector_type(4)));
  const float sv[4] = {v.x * s, v.y * s, v.z * s, v.w * s};
  f16x4_ h, lo;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    _Float16 hh, ll;
    split_f16(sv[j], hh, ll);
    h[j] = hh;
    lo[j] = ll;
  }
  unsigned short* base = planes + (row * ldc + (long long)(col >> 5) * 32) * 2 + (col & 31);
  *reinterpret_cast<f16x4_*>(base) = h;
  *reinterpret_cast<f16x4_*>(base + 32) = lo;
}

}  // namespace gcl
