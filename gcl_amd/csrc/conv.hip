// Sparse convolution forward / input-gradient / weight-gradient for gfx950 (exact-f32 MFMA).
//
// Replaces MinkowskiEngine's per-offset gather -> GEMM -> atomic scatter with an
// OUTPUT-STATIONARY formulation over the k-major neighbour table built in coords.hip:
//   * one wave owns 32 output rows x TN output channels and walks the K kernel offsets; offsets for which none
//     of its 32 rows has a neighbour are skipped with one wave-wide ballot;
//   * gathered input rows are fetched as whole 128-byte lines (8 lanes x 16 B per row), staged through a
//     WAVE-PRIVATE LDS tile and read back as the A fragments of v_mfma_f32_32x32x2_f32 -- no workgroup barrier
//     anywhere in the main loop, the four waves of a workgroup only share the L1-resident weights;
//   * weights are pre-packed (gcl_pack_weights) into B-fragment order so that every wave-instruction reads
//     1 KiB of contiguous L2-resident data (k index of the MFMA is re-mapped so that A and B fragments are
//     float4 reads);
//   * accumulation in registers, one plain store per output element: no atomics, bitwise reproducible.
// The weight gradient runs over the compacted per-offset pair lists (exact sparse work) with per-wave partial
// slabs and an ordered reduction (deterministic).
#include "common.h"

namespace gcl {

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define WAVE_FENCE()                                              \
  do {                                                            \
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");        \
    __builtin_amdgcn_wave_barrier();                              \
  } while (0)

// ---------------------------------------------------------------------------------------------------
// weight packing:  wp[(((k*TNB + nb)*Q + q)*2 + h)*32 + j][e] = W_eff[k][8q + 4h + e][32 nb + j]
// ---------------------------------------------------------------------------------------------------
__global__ void k_pack_weights(const float* __restrict__ w, int K, int cin, int cout, int mode, float* wp) {
  long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long total = (long long)K * cin * cout;
  if (o >= total) return;
  int cin_e = mode == 0 ? cin : cout;
  int cout_e = mode == 0 ? cout : cin;
  int Q = cin_e / 8, TNB = cout_e / 32;
  int e = (int)(o & 3);
  int l = (int)((o >> 2) & 63);
  int h = l >> 5, j = l & 31;
  long long rest = o >> 8;
  int q = (int)(rest % Q);
  int nb = (int)((rest / Q) % TNB);
  int k = (int)(rest / ((long long)Q * TNB));
  int c = 8 * q + 4 * h + e;   // effective input channel
  int n = 32 * nb + j;         // effective output channel
  float v;
  if (mode == 0) {
    v = w[((long long)k * cin + c) * cout + n];
  } else {
    int ks = (mode == 2) ? (K - 1 - k) : k;
    v = w[((long long)ks * cin + n) * cout + c];
  }
  wp[o] = v;
}

// ---------------------------------------------------------------------------------------------------
// forward / input-gradient
// ---------------------------------------------------------------------------------------------------
constexpr int CONV_ROWS = 128;   // output rows per workgroup (4 waves x 32)

// XCD-aware tile index: workgroups are dealt round-robin over the 8 XCDs (blockIdx b and b+8 share an L2), so the
// bijective remap below gives every XCD a CONTIGUOUS range of row tiles -- tiles that are neighbours in the (windowed)
// sort order gather overlapping input rows and then hit the same L2.  Speed only, never correctness.
__device__ __forceinline__ unsigned xcd_tile(unsigned bid, unsigned nwg, int swizzle) {
  if (!swizzle || nwg < 16) return bid;
  unsigned q = nwg >> 3, r = nwg & 7u, xcd = bid & 7u;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}
constexpr int LDS_STRIDE = 36;   // 32 floats + 4 pad: conflict-free ds_read_b128 (MI355X_MICROARCH LDS table)

template <int NB>  // 32-column blocks per wave
__global__ void __launch_bounds__(256) k_conv_fwd(const float* __restrict__ X, const float4* __restrict__ Wp,
                                                  const int* __restrict__ tbl, const int* __restrict__ order,
                                                  const int* __restrict__ tile_mask, long long n_out, int K,
                                                  int cin, int cout, const float* __restrict__ bias,
                                                  float* __restrict__ Y, int swizzle) {
  __shared__ __attribute__((aligned(16))) float lds[4][32][LDS_STRIDE];
  __shared__ int idxs[4][27][32];   // [wave][k][row] (K <= 27)
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int i = l & 31, h = l >> 5;
  const long long tile = (long long)xcd_tile(blockIdx.x, gridDim.x, swizzle) * 4 + w;
  const long long row0 = tile * 32;
  if (row0 >= n_out) return;   // whole wave out of range (no workgroup barriers below)
  const int nb0 = blockIdx.y * NB;
  const int TNB = cout >> 5, Q = cin >> 3, CC = cin >> 5;
  const int p = l & 7, rsub = l >> 3;
  const bool row_ok = (l < 32) && (row0 + l < n_out);

  f32x16 acc[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;

  // offsets this tile has to visit (wave-uniform)
  unsigned kmask = tile_mask ? (unsigned)tile_mask[tile] : ((K >= 32) ? ~0u : ((1u << K) - 1u));
  kmask = __builtin_amdgcn_readfirstlane(kmask);

  // all neighbour indices of the tile go to (wave-private) LDS once: no index -> gather dependency in the loop
  for (int e = l; e < K * 32; e += 64) {
    int k = e >> 5, r = e & 31;
    int v = -1;
    if (row0 + r < n_out) v = tbl ? tbl[(long long)k * n_out + row0 + r] : (int)(row0 + r);
    idxs[w][k][r] = v;
  }
  WAVE_FENCE();
  auto gather = [&](int k, int cc, float4* st) {
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
      int ridx = idxs[w][k][rsub + 8 * ps];
      st[ps] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ridx >= 0) st[ps] = *reinterpret_cast<const float4*>(X + (long long)ridx * cin + cc * 32 + p * 4);
    }
  };

  if (kmask != 0u) {
    // software pipeline over the steps (k, cc): the gather of step s+1 is in flight while step s computes
    int k_cur = __builtin_ctz(kmask), cc_cur = 0;
    unsigned m_rest = kmask & (kmask - 1);
    float4 st[4];
    gather(k_cur, 0, st);
    while (true) {
      int k_nxt = k_cur, cc_nxt = cc_cur + 1;
      bool has_nxt = true;
      if (cc_nxt == CC) {
        cc_nxt = 0;
        if (m_rest) {
          k_nxt = __builtin_ctz(m_rest);
          m_rest &= m_rest - 1;
        } else {
          has_nxt = false;
        }
      }
      WAVE_FENCE();   // the previous step's fragment reads are done before the tile is overwritten
#pragma unroll
      for (int ps = 0; ps < 4; ++ps) *reinterpret_cast<float4*>(&lds[w][rsub + 8 * ps][p * 4]) = st[ps];
      if (has_nxt) gather(k_nxt, cc_nxt, st);
      WAVE_FENCE();
      float4 a[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) a[q] = *reinterpret_cast<const float4*>(&lds[w][i][8 * q + 4 * h]);
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const float4* wb = Wp + (((long long)k_cur * TNB + nb0 + b) * Q + cc_cur * 4) * 64 + l;
        float4 bv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) bv[q] = wb[q * 64];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].x, bv[q].x, acc[b], 0, 0, 0);
          acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].y, bv[q].y, acc[b], 0, 0, 0);
          acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].z, bv[q].z, acc[b], 0, 0, 0);
          acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].w, bv[q].w, acc[b], 0, 0, 0);
        }
      }
      if (!has_nxt) break;
      k_cur = k_nxt;
      cc_cur = cc_nxt;
    }
  }
  // epilogue: acc[b][r] is element (row = (r&3) + 8*(r>>2) + 4*h, col = i) of the wave's 32 x 32 block b
  int orow_l = -1;
  if (row_ok) orow_l = order ? order[row0 + l] : (int)(row0 + l);
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const int col = (nb0 + b) * 32 + i;
    float bvv = bias ? bias[col] : 0.f;
    // consume the (conditional) bias load HERE: otherwise every store below waits for all earlier stores (vmcnt(0))
    asm volatile("v_mov_b32 %0, %1" : "=v"(bvv) : "v"(bvv));
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int orow = __shfl(orow_l, (r & 3) + 8 * (r >> 2) + 4 * h);
      if (orow >= 0) Y[(long long)orow * cout + col] = acc[b][r] + bvv;
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// split-precision forward / input-gradient: fp32 operands split into PL bf16 planes (hi, mid[, lo]) and multiplied
// with v_mfma_f32_32x32x16_bf16 (fp32 accumulate).  PL = 3 keeps 24 significand bits: hh + hm + mh + mm + hl + lh,
// 6 MFMA terms -- measured error against the fp64 oracle equals native fp32 at 2.7x the MFMA rate of
// the exact-f32 instruction; PL = 2 (hh + hm + mh, ~1.5e-5) runs at 5.3x.
// Structure: 128 output rows per workgroup (wave w owns rows 32w..32w+31 and gathers them into its private LDS tile
// exactly as in k_conv_fwd), but the weight block of step (k, cc) is staged ONCE per workgroup in LDS and shared by
// the four waves [measured alternatives, all slower on the KITTI batch: 256 rows per workgroup (two tiles per wave)
// +6 %, every wave streaming its own weight fragments from L2 without barriers +13..20 %] (L2 -> CU weight traffic / 4: with the faster MFMA the per-wave weight stream of k_conv_fwd would
// exceed the L2 bandwidth share of a CU).  A wave whose own rows lack offset k skips its gather and MFMAs.
// ---------------------------------------------------------------------------------------------------
#ifdef GCL_STAMPS
// DIAGNOSTIC BUILD ONLY (tools/stamp_conv.py): per-phase cycle sums of k_conv_fwd_split, added up over all waves.
// [0] wait at barrier #1  [1] LDS writes + next-step load issue  [2] wait at barrier #2  [3] LDS reads + split + MFMA
// [4] prologue  [5] epilogue  [6] wave-steps  [7] wave-steps with MFMA work
__device__ unsigned long long g_stamps[8];
// per workgroup of the LAST k_conv_fwd_split launch: {start, end (s_memrealtime, 100 MHz), steps, s_memtime ticks start -> end}
// (tools/wg_trace.py)
constexpr int WG_TRACE_MAX = 16384;
__device__ unsigned long long g_wgtrace[WG_TRACE_MAX * 4];
#define STAMP(V)                                  \
  __builtin_amdgcn_sched_barrier(0);              \
  unsigned long long V = __builtin_amdgcn_s_memtime(); \
  __builtin_amdgcn_sched_barrier(0);
#else
#define STAMP(V)
#endif

#ifndef GCL_FWD_MIN_WAVES
#define GCL_FWD_MIN_WAVES(NB, PL) (((NB) <= 2 && (PL) != 3) ? 4 : 2)
#endif
// swizzle of the 16-byte pieces of an un-padded 128-byte LDS row: conflict-free ds_read_b128 of one piece of 32 rows
__device__ __forceinline__ int a_swz(int row) { return (row >> 1) & 7; }

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// Arithmetic modes of the split kernels (template parameter PL):
//   2 = "bf16x3": 2 bf16 planes, 3 MFMA terms            (~1.5e-5)
//   3 = "bf16x6": 3 bf16 planes, 6 MFMA terms            (= native fp32)
//   4 = "fp16x3": 2 fp16 planes (11 + 11 significand bits, round-to-nearest => 24 bits), 3 MFMA terms, operands
//       pre-scaled by a per-tensor power of two so that |x| * scale < 2^14 (fp16 range); = native fp32 at half the
//       MFMA work of bf16x6.  The scale comes from the tensor's max-abs (gcl_amax), the result is un-scaled exactly.
template <int PL>
struct Prec {
  static constexpr int planes = (PL == 3) ? 3 : 2;
};

// fp16x3, range-extended (round 5): the lo plane holds the remainder TIMES 2^11, i.e. lo' = fp16((x s - hi) 2^11).  The
// remainder of an fp16-rounded value is <= 2^-11 |x s|, so lo' lies in the same binade range as hi and is a NORMAL fp16
// number whenever hi is -- every element down to 2^-27 of the tensor's max-abs keeps its 22 significand bits.  (Rounds 1 - 4
// stored the remainder itself: it went subnormal for |x| < 2^-17 amax.  Forward activations never get there; the gradient
// tensors do -- a BatchNorm backward leaves a dense background ~ 1 / n below the few rows the loss touches, and the
// reductions downstream cancel the two against each other: the full-size gradient check showed the exact-f32 kernels 8 -
// 20 x closer to the fp64 oracle on the encoder's weight gradients.)  The cross terms hi lo' + lo' hi then carry a factor
// 2^11 and are summed in an accumulator of their own (`accx`), folded in once at the end: acc + accx 2^-11.  Same MFMA
// count; NB x 16 more accumulator registers.
// (F16_LO_UP / F16_LO_DOWN, split_f16 and amax_scale live in common.h: the BatchNorm passes write plane images too)
template <int PL>
__device__ __forceinline__ void split8(const float4& f0, const float4& f1, float scale, u32x4* pl) {
  float v[8] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w};
  if (PL == 4) {
    // two elements at a time: sv = v s and t = v (2^11 s) are both exact (s is a power of two); hi = fp16(sv) packed;
    // lo' = fp16(t - 2^11 hi) with the fp16 value as a source of ONE mixed-precision FMA (v_fma_mix_f32: no conversion of
    // hi back to fp32) -- six instructions per pair of elements, what the un-scaled lo plane of rounds 1 - 4 cost (the
    // compiler's own packed form needs seven).  Same values as split_f16.
    typedef _Float16 f16x2_ __attribute__((ext_vector_type(2)));
    const float scale_up = scale * F16_LO_UP, neg_up = -F16_LO_UP;
    typedef float f32x2_ __attribute__((ext_vector_type(2)));
    unsigned hw[4], lw[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f32x2_ vv = {v[2 * j], v[2 * j + 1]};
      const f32x2_ sv = vv * scale, tv = vv * scale_up;               // packed multiplies
      const f16x2_ h2 = __builtin_convertvector(sv, f16x2_);           // one packed conversion
      hw[j] = __builtin_bit_cast(unsigned, h2);
      f32x2_ r;
      asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r[0]) : "v"(hw[j]), "s"(neg_up), "v"(tv[0]));
      asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r[1]) : "v"(hw[j]), "s"(neg_up), "v"(tv[1]));
      lw[j] = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2_));
    }
    pl[0] = u32x4{hw[0], hw[1], hw[2], hw[3]};
    pl[1] = u32x4{lw[0], lw[1], lw[2], lw[3]};
  } else {
    bf16x8 p0, p1, p2;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      __bf16 hi = (__bf16)v[j];
      float r = v[j] - (float)hi;
      __bf16 mid = (__bf16)r;
      p0[j] = hi;
      p1[j] = mid;
      if (PL == 3) p2[j] = (__bf16)(r - (float)mid);
    }
    pl[0] = __builtin_bit_cast(u32x4, p0);
    pl[1] = __builtin_bit_cast(u32x4, p1);
    if (PL == 3) pl[2] = __builtin_bit_cast(u32x4, p2);
  }
}

template <int PL>
__device__ __forceinline__ f32x16 mfma16(const u32x4& a, const u32x4& b, const f32x16& c) {
  if (PL == 4)
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// all product terms of one (A planes, B planes) pair, small terms first.  PL == 4: the two cross terms go to `accx` (they
// carry the lo planes' factor 2^11), hi hi to `acc`; fold_cross adds them up.  The bf16 modes use `acc` only.
template <int PL>
__device__ __forceinline__ void mfma_terms(const u32x4* a, const u32x4* b, f32x16& acc, f32x16& accx) {
  if (PL == 4) {
    accx = mfma16<PL>(a[1], b[0], accx);
    accx = mfma16<PL>(a[0], b[1], accx);
    acc = mfma16<PL>(a[0], b[0], acc);
    return;
  }
  if (PL == 3) {
    acc = mfma16<PL>(a[2], b[0], acc);
    acc = mfma16<PL>(a[0], b[2], acc);
    acc = mfma16<PL>(a[1], b[1], acc);
  }
  acc = mfma16<PL>(a[1], b[0], acc);
  acc = mfma16<PL>(a[0], b[1], acc);
  acc = mfma16<PL>(a[0], b[0], acc);
}
// The weight-gradient kernels hold 64 - 96 accumulator registers per wave and have no room for a second set.  They do not
// need one: of their two operands only the output gradient (B) spans many binades; the layer input x (A) is an activation,
// where the round 1 - 4 format was adequate.  So A's two planes are re-scaled on the fly (packed fp16 multiplies by 2^-11,
// exact unless the value drops below fp16's normal range, i.e. for |x| < 2^-17 max|x|):
//   a[0] = hi_x,  a[1] = lo_x = lo'_x 2^-11 (the plain remainder),  a[2] = hi_x 2^-11
// and all three terms have the same scale:  hi_x hi_g + lo_x hi_g + (hi_x 2^-11) lo'_g  -> ONE accumulator.
__device__ __forceinline__ u32x4 f16x8_down(const u32x4& v) {
  f16x8 h = __builtin_bit_cast(f16x8, v);
  h = h * (_Float16)F16_LO_DOWN;
  return __builtin_bit_cast(u32x4, h);
}
// a: {hi, lo'} as stored -> {hi, lo, hi 2^-11}
__device__ __forceinline__ void dw_a_planes(u32x4* a) {
  a[2] = f16x8_down(a[0]);
  a[1] = f16x8_down(a[1]);
}
template <int PL>
__device__ __forceinline__ void mfma_terms_dw(const u32x4* a, const u32x4* b, f32x16& acc) {
  if (PL == 4) {
    acc = mfma16<PL>(a[1], b[0], acc);
    acc = mfma16<PL>(a[2], b[1], acc);
    acc = mfma16<PL>(a[0], b[0], acc);
    return;
  }
  f32x16 unused;
  mfma_terms<PL>(a, b, acc, unused);
}
template <int PL>
__device__ __forceinline__ void fold_cross(f32x16& acc, f32x16& accx) {
  if (PL == 4) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      acc[r] += accx[r] * F16_LO_DOWN;
      accx[r] = 0.f;
    }
  }
}

// max |x| of a tensor as the bit pattern of a non-negative float (integer atomicMax is order-independent)
__global__ void __launch_bounds__(256) k_amax(const float4* __restrict__ x, long long n4, const float* __restrict__ tail,
                                              int n_tail, int* amax_bits) {
  float m = 0.f;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n4; e += (long long)gridDim.x * blockDim.x) {
    float4 v = x[e];
    m = fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
  }
  if (blockIdx.x == 0 && (int)threadIdx.x < n_tail) m = fmaxf(m, fabsf(tail[threadIdx.x]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  __shared__ float wm[4];
  if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0)   // one atomic per workgroup: contended same-address atomics serialise (~10 ns each)
    amax_slot_publish(amax_bits, __float_as_int(fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]))), blockIdx.x);
}

// the same for a list of tensors in one launch (all convolution kernels of a model): grid = (chunks, tensors)
__global__ void __launch_bounds__(256) k_amax_multi(const float* const* __restrict__ ptrs,
                                                    const long long* __restrict__ sizes, int* amax_bits) {
  const float* x = ptrs[blockIdx.y];
  const long long n = sizes[blockIdx.y];
  float m = 0.f;
  if ((reinterpret_cast<unsigned long long>(x) & 15) == 0) {
    const long long n4 = n >> 2;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n4; e += (long long)gridDim.x * blockDim.x) {
      float4 v = reinterpret_cast<const float4*>(x)[e];
      m = fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
    }
    if (blockIdx.x == 0 && (long long)threadIdx.x < n - n4 * 4) m = fmaxf(m, fabsf(x[n4 * 4 + threadIdx.x]));
  } else {
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x)
      m = fmaxf(m, fabsf(x[e]));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  __shared__ float wm[4];
  if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0)
    amax_slot_publish(amax_bits + (size_t)blockIdx.y * AMAX_WORDS,
                      __float_as_int(fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]))), blockIdx.x);
}

// fp32 [n, c] -> fp16 plane image [n][c / 32][2][32]: hi = fp16(x s), lo = fp16((x s - hi) 2^11) (split_f16), s from the tensor's amax
// slot.  One thread per 4 channels (16-byte read, two 8-byte writes).
__global__ void __launch_bounds__(256) k_split_planes(const float4* __restrict__ x, long long total4, int c,
                                                      const int* __restrict__ amax, unsigned short* __restrict__ planes) {
  const float s = amax_scale(amax);
  const int q_per_row = c >> 2;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total4;
       e += (long long)gridDim.x * blockDim.x) {
    const float4 v = x[e];
    const long long row = e / q_per_row;
    const int col = (int)(e % q_per_row) * 4;            // first channel of this quad
    store_planes4(planes, row, c, col, v, s);
  }
}

// wp16 (8 x 16-bit units): [(((k*CC + cc)*TNB + nb)*2 + m)*planes + pl][lane = h*32 + j][jj]
//                         = plane pl of W_eff[k][cc*32 + 16m + 8h + jj][32 nb + j]   (fp16 mode: of W_eff * scale)
template <int PL>
__device__ __forceinline__ void pack_split_one(const float* __restrict__ w, int K, int cin, int cout, int mode,
                                               const int* __restrict__ w_amax, unsigned short* wp, long long o) {
  constexpr int NPL = Prec<PL>::planes;
  int cin_e = mode == 0 ? cin : cout;
  int cout_e = mode == 0 ? cout : cin;
  int CC = cin_e / 32, TNB = cout_e / 32;
  int jj = (int)(o & 7);
  int l = (int)((o >> 3) & 63);
  int h = l >> 5, j = l & 31;
  long long rest = o >> 9;
  int m = (int)(rest & 1);
  rest >>= 1;
  int nb = (int)(rest % TNB);
  int cc = (int)((rest / TNB) % CC);
  int k = (int)(rest / ((long long)TNB * CC));
  int c = cc * 32 + 16 * m + 8 * h + jj;
  int n = 32 * nb + j;
  float v;
  if (mode == 0) {
    v = w[((long long)k * cin + c) * cout + n];
  } else {
    int ks = (mode == 2) ? (K - 1 - k) : k;
    v = w[((long long)ks * cin + n) * cout + c];
  }
  long long blk = ((((long long)k * CC + cc) * TNB + nb) * 2 + m) * NPL;
  if (PL == 4) {
    _Float16 hi, lo;
    split_f16(v * amax_scale(w_amax), hi, lo);
    wp[((blk + 0) * 64 + l) * 8 + jj] = __builtin_bit_cast(unsigned short, hi);
    wp[((blk + 1) * 64 + l) * 8 + jj] = __builtin_bit_cast(unsigned short, lo);
  } else {
    __bf16 hi = (__bf16)v;
    float r = v - (float)hi;
    __bf16 mid = (__bf16)r;
    wp[((blk + 0) * 64 + l) * 8 + jj] = __builtin_bit_cast(unsigned short, hi);
    wp[((blk + 1) * 64 + l) * 8 + jj] = __builtin_bit_cast(unsigned short, mid);
    if (PL == 3) wp[((blk + 2) * 64 + l) * 8 + jj] = __builtin_bit_cast(unsigned short, (__bf16)(r - (float)mid));
  }
}

template <int PL>
__global__ void k_pack_weights_split(const float* __restrict__ w, int K, int cin, int cout, int mode,
                                     const int* __restrict__ w_amax, unsigned short* wp) {
  long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= (long long)K * cin * cout) return;
  pack_split_one<PL>(w, K, cin, cout, mode, w_amax, wp, o);
}

// all convolution kernels of a network in one launch.  desc[t] = {w pointer, K, cin, cout, mode, amax slot index,
// byte offset of the packed tensor in `out`, first workgroup}; a workgroup finds its tensor by binary search
struct PackDesc { long long w, K, cin, cout, mode, amax_index, out_off, first_wg; };
template <int PL>
__global__ void __launch_bounds__(256) k_pack_weights_multi(const PackDesc* __restrict__ desc, int n_tensors,
                                                            const int* __restrict__ amax_slots,
                                                            unsigned char* __restrict__ out) {
  int lo = 0, hi = n_tensors - 1;
  while (lo < hi) {
    int mid = (lo + hi + 1) >> 1;
    if (desc[mid].first_wg <= (long long)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const PackDesc d = desc[lo];
  const long long o = ((long long)blockIdx.x - d.first_wg) * 256 + threadIdx.x;
  if (o >= d.K * d.cin * d.cout) return;
  if ((d.cin & 31) || (d.cout & 31) || d.K > 27) {      // generic shape: fp32 W_eff[k][c][n] (k_pack_weights_generic)
    const int cin = (int)d.cin, cout = (int)d.cout, K = (int)d.K, mode = (int)d.mode;
    const float* w = reinterpret_cast<const float*>(d.w);
    const int cin_e = mode == 0 ? cin : cout, cout_e = mode == 0 ? cout : cin;
    const int n = (int)(o % cout_e), c = (int)((o / cout_e) % cin_e), k = (int)(o / ((long long)cin_e * cout_e));
    reinterpret_cast<float*>(out + d.out_off)[o] =
        (mode == 0) ? w[((long long)k * cin + c) * cout + n]
                    : w[((long long)((mode == 2) ? (K - 1 - k) : k) * cin + n) * cout + c];
    return;
  }
  pack_split_one<PL>(reinterpret_cast<const float*>(d.w), (int)d.K, (int)d.cin, (int)d.cout, (int)d.mode,
                     amax_slots + d.amax_index * AMAX_WORDS, reinterpret_cast<unsigned short*>(out + d.out_off), o);
}

// optional fused epilogue of the inference path: y = relu?(acc * col_scale + bias (+ residual)), amax(y) published
struct ConvEpi {
  const float* col_scale;   // per output column (BatchNorm in eval mode: gamma * rsqrt(var + eps)), NULL = 1
  const float* residual;    // [n_out, cout] added before the ReLU, NULL = none
  int relu;
  int* y_amax;              // zero-initialised amax slot of y, NULL = not wanted
  int residual_ld;          // row pitch of `residual` in floats (a column slice of a wider tensor), 0 = cout
};

// Epilogue of the four-wave forward kernels (k_conv_fwd_split, k_conv_fwd_dma): un-scale, bias, optional fused inference
// epilogue, scatter to the original row order, per-workgroup column sums for the BatchNorm that follows.  `tiles` = the
// workgroup's four wave-private 32 x 32-float A tiles (idle by now: scratch for the column sums).
template <int NB, bool EPI>
__device__ __forceinline__ void conv_fwd_epilogue(f32x16 (&acc)[NB], float* tiles, const int* __restrict__ order,
                                                  long long n_out, int cout, const float* __restrict__ bias,
                                                  float* __restrict__ Y, float* __restrict__ stats, float out_scale,
                                                  const ConvEpi& epi, long long tile, unsigned bxx, int nb0, bool active) {
  const int t = threadIdx.x, l = t & 63, w = t >> 6;
  const int i = l & 31, h = l >> 5;
  const long long row0 = tile * 32;
  if (!active && !stats) return;
  int orow_l = -1;
  if (active && (l < 32) && (row0 + l < n_out)) orow_l = order ? order[row0 + l] : (int)(row0 + l);
  float ymax = 0.f;
  // column sums of this wave's 32 rows go to its own (now idle) A tile; wave 0 adds the four waves in order and writes
  // ONE partial per workgroup (128 rows) for the BatchNorm that follows (saves its statistics pass over Y)
  float* const ssc = tiles + w * (32 * 32);   // wave w's tile; tiles are 32 x 32 floats apart
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const int col = (nb0 + b) * 32 + i;
    float bvv = bias ? bias[col] : 0.f;
    float csc = (EPI && epi.col_scale) ? epi.col_scale[col] * out_scale : out_scale;
    // consume the (conditional) loads HERE: otherwise every store below waits for all earlier stores (vmcnt(0))
    asm volatile("v_mov_b32 %0, %1" : "=v"(bvv) : "v"(bvv));
    asm volatile("v_mov_b32 %0, %1" : "=v"(csc) : "v"(csc));
    float s1 = 0.f, s2 = 0.f, lo = 3.0e38f, hi = -3.0e38f;      // column sum, sum of squares, minimum, maximum
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int orow = __shfl(orow_l, (r & 3) + 8 * (r >> 2) + 4 * h);
      if (orow >= 0) {
        float v = acc[b][r] * csc + bvv;
        if (EPI && epi.residual) {
          const float rsd = epi.residual[(long long)orow * (epi.residual_ld ? epi.residual_ld : cout) + col];
          v = epi.relu == 2 ? (rsd > 0.f ? v : 0.f) : v + rsd;      // relu == 2: threshold_backward, pass v where residual > 0
        }
        if (EPI && epi.relu == 1) v = fmaxf(v, 0.f);
        Y[(long long)orow * cout + col] = v;
        s1 += v;
        s2 += v * v;
        lo = fminf(lo, v);
        hi = fmaxf(hi, v);
        if (EPI) ymax = fmaxf(ymax, fabsf(v));
      }
    }
    if (stats) {
      s1 += __shfl_xor(s1, 32);
      s2 += __shfl_xor(s2, 32);
      lo = fminf(lo, __shfl_xor(lo, 32));
      hi = fmaxf(hi, __shfl_xor(hi, 32));
      if (h == 0) {
        ssc[b * 32 + i] = s1;
        ssc[NB * 32 + b * 32 + i] = s2;
        ssc[2 * NB * 32 + b * 32 + i] = lo;
        ssc[3 * NB * 32 + b * 32 + i] = hi;
      }
    }
  }
  if (stats) {
    __syncthreads();
    if (w == 0 && h == 0) {
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const int col = (nb0 + b) * 32 + i;
        float t1 = 0.f, t2 = 0.f, tlo = 3.0e38f, thi = -3.0e38f;
#pragma unroll
        for (int ww = 0; ww < 4; ++ww) {
          const float* o = ssc + ww * (32 * 32);
          t1 += o[b * 32 + i];
          t2 += o[NB * 32 + b * 32 + i];
          tlo = fminf(tlo, o[2 * NB * 32 + b * 32 + i]);
          thi = fmaxf(thi, o[3 * NB * 32 + b * 32 + i]);
        }
        // channel-major [4][cout][n_part] (sum, sum of squares, minimum, maximum): the finalisation reads a channel's
        // partials as contiguous runs ([n_part][2][cout] made it a walk of 16-byte pieces 256 bytes apart: 55 us per 0.5 M-row
        // layer).  Minimum / maximum (round 5): the exact range of the BatchNorm's output follows from them BEFORE its apply
        // pass runs (an affine map per channel is monotone), which is what lets that pass write fp16 plane images itself.
        const long long n_part = (n_out + 127) >> 7;
        stats[(long long)col * n_part + bxx] = t1;
        stats[((long long)cout + col) * n_part + bxx] = t2;
        stats[((long long)2 * cout + col) * n_part + bxx] = tlo;
        stats[((long long)3 * cout + col) * n_part + bxx] = thi;
      }
    }
    if (!active) return;
  }
  if (EPI && epi.y_amax) {   // one publish per wave (launches of this path are small: inference on single clouds)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ymax = fmaxf(ymax, __shfl_xor(ymax, o));
    if (l == 0) amax_slot_publish(epi.y_amax, __float_as_int(ymax), (unsigned)tile);
  }
}

// PRE: X is not the fp32 tensor but its fp16 plane image made by gcl_split_planes -- per row and 32-channel slice
// 64 bytes of hi followed by 64 bytes of lo, i.e. the same 128 bytes per (row, slice) and the same addressing as the
// fp32 rows.  The main loop then has no split at all (the kernels are instruction-issue-bound: the split was 48 of ~120
// instructions per step), a fragment is two 16-byte LDS reads.
template <int NB, int PL, bool PRE = false, bool EPI = false>
__global__ void __launch_bounds__(256, GCL_FWD_MIN_WAVES(NB, PL)) k_conv_fwd_split(const float* __restrict__ X, const u32x4* __restrict__ Wp,
                                                        const int* __restrict__ tbl, const int* __restrict__ order,
                                                        const int* __restrict__ tile_mask, long long n_out, int K,
                                                        int cin, int cout, const float* __restrict__ bias,
                                                        float* __restrict__ Y, int swizzle,
                                                        float* __restrict__ stats, const int* __restrict__ x_amax,
                                                        const int* __restrict__ w_amax, unsigned x_bytes, ConvEpi epi) {
  constexpr int NPL = Prec<PL>::planes;
  constexpr int BLK = NB * 2 * NPL * 64;                // uint4 per (k, cc) weight block of this workgroup
  constexpr int BREG = (BLK + 255) / 256;
  const float a_scale = (PL == 4) ? amax_scale(x_amax) : 1.f;
  const float out_scale = (PL == 4) ? 1.f / (a_scale * amax_scale(w_amax)) : 1.f;   // exact: powers of two
  __shared__ __attribute__((aligned(16))) float Asm[4][32][32];   // wave-private A tiles: un-padded 128-byte rows, 16-byte pieces XOR-swizzled with the row
  __shared__ __attribute__((aligned(16))) u32x4 Bsm[2][BLK];              // weight block, double-buffered
  // neighbour rows of the wave's tile, [wave][k][(row & 7) * 4 + (row >> 3)] (K <= 27): the four rows a lane gathers
  // (row, row + 8, row + 16, row + 24) are one 16-byte read.  The table holds HALF of the offsets at a time (k < 15, later
  // k >= 15: the loop visits offsets in ascending order; the second half waits in registers and is written when the first
  // unit with k >= 15 is issued) -- 7.5 KB instead of 14 KB per workgroup, which is what lets a fourth workgroup of the
  // NB = 2 instances share a CU's 160 KB LDS (occupancy is what hides the gather latency here)
  constexpr int ISM_H = 15;
  __shared__ __attribute__((aligned(16))) int Ism[4][ISM_H][32];
  __shared__ unsigned wmask[4];
  const int t = threadIdx.x, l = t & 63, w = t >> 6;
  const int i = l & 31, h = l >> 5;
  // swizzle 2 (1-D grid): the cout / (32 NB) column blocks that gather the SAME 128 rows get consecutive slots of one
  // XCD (workgroups are dealt round-robin over the 8 XCDs) and share its L2
  unsigned bxx = blockIdx.x, byy = blockIdx.y;
  // bit 4 of `swizzle`: process the row tiles in DESCENDING order.  The mask sort puts the rows that own rare offsets
  // last; their tiles visit 2 - 3x the average number of offsets (per-workgroup union: mean 8.9, p90 17, max 27 at
  // stride 1), so in ascending order the longest workgroups start last and the launch ends on a thin, slow tail
  const bool heavy_first = (swizzle & 16) != 0;
  swizzle &= 15;
  const unsigned nrw = (unsigned)((n_out + CONV_ROWS - 1) / CONV_ROWS);
  if (swizzle >= 2) {
    const unsigned ncb = (unsigned)(cout / (32 * NB));
    const unsigned xcd = bxx & 7u, slot = bxx >> 3;
    byy = slot % ncb;
    if (swizzle == 2) {
      bxx = (slot / ncb) * 8u + xcd;
      if (bxx >= nrw) return;   // whole workgroup, before any barrier
    } else {                    // 3: every XCD owns a CONTIGUOUS range of row tiles (spatially ordered tables)
      const unsigned q = nrw >> 3, r = nrw & 7u, mine = q + (xcd < r ? 1u : 0u), j = slot / ncb;
      if (j >= mine) return;
      bxx = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
  } else {
    bxx = xcd_tile(bxx, gridDim.x, swizzle);
  }
  if (heavy_first) bxx = nrw - 1u - bxx;
  const long long tile = (long long)bxx * 4 + w;
  const long long row0 = tile * 32;
  const bool active = row0 < n_out;
  const int nb0 = byy * NB;
  const int TNB = cout >> 5, CC = cin >> 5;
  const int p = l & 7, rsub = l >> 3;

  f32x16 acc[NB], accx[(PL == 4) ? NB : 1];      // accx: the fp16x3 cross terms (mfma_terms)
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
#pragma unroll
  for (int b = 0; b < (int)(sizeof(accx) / sizeof(accx[0])); ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) accx[b][r] = 0.f;

  unsigned mymask = 0u;
  if (active) mymask = tile_mask ? (unsigned)tile_mask[tile] : ((K >= 32) ? ~0u : ((1u << K) - 1u));
  mymask = __builtin_amdgcn_readfirstlane(mymask);
  if (l == 0) wmask[w] = mymask;
  // all neighbour indices of the tile go to LDS once: no index load (and no index -> gather dependency) in the loop
  // only the offsets of the wave's mask are ever looked up: the other table rows are not fetched (a wave tile visits
  // ~8 of 27 offsets on the KITTI batch: the index table is as large as a C = 32 output tile)
  for (int e = l; e < (K < ISM_H ? K : ISM_H) * 32; e += 64) {
    const int k = e >> 5, r = e & 31;
    int v = -1;
    if (active && ((mymask >> k) & 1u) && row0 + r < n_out) v = tbl ? tbl[(long long)k * n_out + row0 + r] : (int)(row0 + r);
    Ism[w][k][(r & 7) * 4 + (r >> 3)] = v;
  }
  int ib[6];                       // entries e = 32 ISM_H + l + 64 j (offsets k = e / 32 >= 15, rows e % 32), j = 0 .. 5
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const int e = 32 * ISM_H + l + 64 * j, k = e >> 5, r = e & 31;
    int v = -1;
    if (k < K && active && ((mymask >> k) & 1u) && row0 + r < n_out) v = tbl[(long long)k * n_out + row0 + r];
    ib[j] = v;
  }
  bool second_half = false;
  __syncthreads();
  const unsigned wgmask = wmask[0] | wmask[1] | wmask[2] | wmask[3];

  // Step (k, cc) = one kernel offset x one 32-channel slice.  Pipeline (ONE workgroup barrier per step):
  //   loads of step s+2 are issued at the end of step s and land in registers during step s+1,
  //   they are written to LDS at the end of step s+1 (A: wave-private tile; B: the buffer not being read),
  //   the barrier closing step s+1 publishes B(s+2).
  // Gather through a buffer resource: a missing neighbour (row -1) wraps to a byte offset beyond the tensor and the
  // hardware returns zeros -- no compare / branch / 64-bit address arithmetic per row (the loop is issue-bound)
  const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)X, 0, (int)x_bytes, 0x00020000);
  const unsigned row_bytes = (unsigned)cin * 4u;
#define GCL_GATHER_A(KK, CCV)                                                                                  \
  {                                                                                                            \
    if ((KK) >= ISM_H && !second_half) {   /* every gather of an offset < 14 has been issued: switch the table */ \
      second_half = true;                                                                                      \
      WAVE_FENCE();                                                                                            \
      _Pragma("unroll") for (int j_ = 0; j_ < 6; ++j_) {                                                       \
        const int e_ = 32 * ISM_H + l + 64 * j_;                                                               \
        if ((e_ >> 5) < K) Ism[w][(e_ >> 5) - ISM_H][((e_ & 31) & 7) * 4 + ((e_ & 31) >> 3)] = ib[j_];         \
      }                                                                                                        \
      WAVE_FENCE();                                                                                            \
    }                                                                                                          \
    const int4 ri_ = *reinterpret_cast<const int4*>(&Ism[w][(KK) >= ISM_H ? (KK) - ISM_H : (KK)][rsub * 4]);    \
    const unsigned co_ = (unsigned)(CCV)*128u + (unsigned)p * 16u;                                             \
    st[0] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, (int)((unsigned)ri_.x * row_bytes + co_), 0, 0)); \
    st[1] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, (int)((unsigned)ri_.y * row_bytes + co_), 0, 0)); \
    st[2] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, (int)((unsigned)ri_.z * row_bytes + co_), 0, 0)); \
    st[3] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, (int)((unsigned)ri_.w * row_bytes + co_), 0, 0)); \
  }
#define GCL_LOAD_B(KK, CCV)                                                                      \
  {                                                                                              \
    const u32x4* src_ = Wp + (((long long)(KK)*CC + (CCV)) * TNB + nb0) * (2 * NPL * 64);        \
    _Pragma("unroll") for (int e = 0; e < BREG; ++e) {                                           \
      if ((BLK % 256 == 0) || (e * 256 + t < BLK)) br[e] = src_[e * 256 + t];                    \
    }                                                                                            \
  }
#define GCL_STORE_LDS(MINE, BUF)                                                                             \
  {                                                                                                          \
    if (MINE) {                                                                                              \
      _Pragma("unroll") for (int ps = 0; ps < 4; ++ps)                                                       \
          *reinterpret_cast<float4*>(&Asm[w][rsub + 8 * ps][(p ^ a_swz(rsub + 8 * ps)) << 2]) = st[ps];      \
    }                                                                                                        \
    _Pragma("unroll") for (int e = 0; e < BREG; ++e) {                                                       \
      if ((BLK % 256 == 0) || (e * 256 + t < BLK)) Bsm[BUF][e * 256 + t] = br[e];                           \
    }                                                                                                        \
  }
#define GCL_ADVANCE(KV, CV, HAS)             \
  {                                          \
    CV += 1;                                 \
    if (CV == CC) {                          \
      CV = 0;                                \
      if (m_rest) {                          \
        KV = __builtin_ctz(m_rest);          \
        m_rest &= m_rest - 1;                \
      } else {                               \
        HAS = false;                         \
      }                                      \
    }                                        \
  }

#ifdef GCL_STAMPS
  const unsigned long long wg_t0 = __builtin_amdgcn_s_memrealtime(), wg_c0 = __builtin_amdgcn_s_memtime();
  unsigned long long wg_steps = 0;
#endif
  if (wgmask != 0u) {
    unsigned m_rest = wgmask & (wgmask - 1);
    float4 st[4];
    u32x4 br[BREG];
    // step 0 -> LDS, step 1 -> registers
    int k0 = __builtin_ctz(wgmask), c0 = 0;
    bool mine0 = (mymask >> k0) & 1u;
    if (mine0) GCL_GATHER_A(k0, 0);
    GCL_LOAD_B(k0, 0);
    int k1 = k0, c1 = 0;
    bool has1 = true;
    GCL_ADVANCE(k1, c1, has1);
    GCL_STORE_LDS(mine0, 0);
    bool mine1 = false;
    if (has1) {
      mine1 = (mymask >> k1) & 1u;
      if (mine1) GCL_GATHER_A(k1, c1);
      GCL_LOAD_B(k1, c1);
    }
    __syncthreads();
    int buf = 0;
    bool mine_cur = mine0;
#ifdef GCL_STAMPS
    unsigned long long st_c = 0, st_w = 0, st_i = 0, st_b = 0, st_n = 0, st_m = 0;
#endif
    while (true) {
      STAMP(ts0)
      // ---- compute the current step from Asm[w] (own tile) and Bsm[buf]
      if (mine_cur) {
        WAVE_FENCE();
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          u32x4 ap[3];
          if (PRE) {   // row image: dwords [0,16) = hi of channels 0..31, [16,32) = lo
            ap[0] = *reinterpret_cast<const u32x4*>(&Asm[w][i][((2 * m + h) ^ a_swz(i)) << 2]);
            ap[1] = *reinterpret_cast<const u32x4*>(&Asm[w][i][((4 + 2 * m + h) ^ a_swz(i)) << 2]);
          } else {
            float4 f0 = *reinterpret_cast<const float4*>(&Asm[w][i][((4 * m + 2 * h) ^ a_swz(i)) << 2]);
            float4 f1 = *reinterpret_cast<const float4*>(&Asm[w][i][((4 * m + 2 * h + 1) ^ a_swz(i)) << 2]);
            split8<PL>(f0, f1, a_scale, ap);
          }
#pragma unroll
          for (int b = 0; b < NB; ++b) {
            const u32x4* bb = &Bsm[buf][((b * 2 + m) * NPL) * 64 + l];
            u32x4 bp[3];
            bp[0] = bb[0];
            bp[1] = bb[64];
            if (NPL == 3) bp[2] = bb[128];
            mfma_terms<PL>(ap, bp, acc[b], accx[(PL == 4) ? b : 0]);
          }
        }
      }
      STAMP(ts1)
      if (!has1) break;   // no step staged in registers: done
      // ---- next step: registers -> LDS (A: own tile, after this wave's reads; B: the buffer nobody reads)
      WAVE_FENCE();
      GCL_STORE_LDS(mine1, buf ^ 1);
      STAMP(ts2)
      // ---- the step after: issue its loads (in flight across the barrier and the next compute phase)
      int k2 = k1, c2 = c1;
      bool has2 = true, mine2 = false;
      GCL_ADVANCE(k2, c2, has2);
      if (has2) {
        mine2 = (mymask >> k2) & 1u;
        if (mine2) GCL_GATHER_A(k2, c2);
        GCL_LOAD_B(k2, c2);
      }
      STAMP(ts3)
      __syncthreads();   // publishes the weight block just written; its buffer was last read two steps ago
#ifdef GCL_STAMPS
      {
        STAMP(ts4)
        st_c += ts1 - ts0; st_w += ts2 - ts1; st_i += ts3 - ts2; st_b += ts4 - ts3; st_n += 1; st_m += mine_cur ? 1 : 0;
      }
#endif
      buf ^= 1;
      mine_cur = mine1;
      mine1 = mine2;
      k1 = k2;
      c1 = c2;
      has1 = has2;
    }
#ifdef GCL_STAMPS
    wg_steps = st_n;
    if (l == 0) {      // [0] MFMA phase  [1] wait for the gather + LDS writes  [2] load issue  [3] barrier  [6] steps  [7] own steps
      atomicAdd(&g_stamps[0], st_c); atomicAdd(&g_stamps[1], st_w); atomicAdd(&g_stamps[2], st_i);
      atomicAdd(&g_stamps[3], st_b); atomicAdd(&g_stamps[6], st_n); atomicAdd(&g_stamps[7], st_m);
    }
#endif
  }
#undef GCL_GATHER_A
#undef GCL_LOAD_B
#undef GCL_STORE_LDS
#undef GCL_ADVANCE
#ifdef GCL_STAMPS
  if (t == 0 && blockIdx.x < WG_TRACE_MAX && gridDim.y == 1) {
    unsigned long long* o = g_wgtrace + (long long)blockIdx.x * 4;
    o[0] = wg_t0; o[1] = __builtin_amdgcn_s_memrealtime(); o[2] = wg_steps; o[3] = __builtin_amdgcn_s_memtime() - wg_c0;
  }
#endif
#pragma unroll
  for (int b = 0; b < NB; ++b) fold_cross<PL>(acc[b], accx[(PL == 4) ? b : 0]);
  conv_fwd_epilogue<NB, EPI>(acc, &Asm[0][0][0], order, n_out, cout, bias, Y, stats, out_scale, epi, tile, bxx, nb0, active);
}

// One LDS-DMA piece: every lane fetches 16 bytes at byte offset (voff + soff) of the buffer `rsrc` (beyond num_records: zeros)
// and the wave's 64 pieces land at LDS bytes lds_base + 16 lane (lds_base, soff, rsrc wave-uniform, in SGPRs).  Inline asm
// on purpose: hipcc makes every later LDS read wait (vmcnt(0)) for an LDS-DMA it knows about, which would put the DMA's
// latency back into the step; written this way it does not count them -- the kernel waits itself (dma_wait_all) in front
// of the barrier that precedes the reads.  M0 (the DMA's LDS base) is written in the same statement that uses it and
// restored: the compiler does not expect an asm statement to change it.
typedef unsigned rsrc_words __attribute__((ext_vector_type(4)));
__device__ __forceinline__ rsrc_words make_rsrc_words(const void* base, unsigned bytes) {
  const unsigned long long a = reinterpret_cast<unsigned long long>(base);
  rsrc_words r;
  r[0] = (unsigned)a;
  r[1] = (unsigned)(a >> 32) & 0xffffu;       // stride 0, no swizzle
  r[2] = bytes;                               // num_records (bytes, raw buffer)
  r[3] = 0x00020000u;
  return r;
}
template <int IMM>      // LDS destination = lds_base + IMM (bytes; the constant is added on the scalar unit inside the statement)
__device__ __forceinline__ void dma16(const rsrc_words& rsrc, unsigned lds_base, unsigned voff, unsigned soff) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_add_u32 m0, %1, %5\n\t"
      "s_nop 0\n\t"
      "buffer_load_dwordx4 %2, %3, %4 offen lds\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "s"(lds_base), "v"(voff), "s"(rsrc), "s"(soff), "i"(IMM)
      : "memory", "scc");
}
__device__ __forceinline__ void dma_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// ---- the same product with LDS-DMA staging (round 4; plane-image operands, fp16x3) -----------------------------------
// k_conv_fwd_split moves every operand global -> VGPR -> ds_write -> LDS: per wave and step 6 loads into 24 staging
// registers and 6 ds_write_b128 (13 issue cycles each on the VGPR -> LDS path, ~1200 of the ~5600 cycles a CU spends on a
// round of 16 wave-steps).  With plane images the gathered bytes ARE the LDS image, so here they go straight to LDS
// (`buffer_load_dwordx4 ... lds`: per-lane source offset, wave-uniform LDS base + 16 lane): no staging registers, no
// ds_write, and the XOR swizzle of the A tile is applied to the SOURCE piece a lane fetches.  Missing neighbours (row -1)
// wrap beyond the buffer resource and the DMA writes zeros (tools/micro/dma_probe.hip).  One workgroup barrier per step:
//   top of step s:  __syncthreads() (waits for this wave's DMAs: A(s) in its own tile, its part of B(s)), then
//                   B(s+1) -> the other weight buffer (last read in step s-1, before this barrier),
//                   A fragments of step s -> registers, then A(s+1) -> the wave's tile, then B fragments + MFMAs.
// Same products added in the same order as k_conv_fwd_split: bitwise the same y and column sums.
// GRP (round 6): the launch of ONE offset group of an inference layer (k_conv_fwd_tall's four fixed groups, k mod 4 == g, as
// four times as many ordinary workgroups instead of sixteen-wave ones): the group index rides in the column-block slot of the
// 1-D grid, the tile's mask is cut to the group's offsets, the arithmetic is k_conv_fwd_tall's (ONE accumulator, the
// activation planes re-scaled on the fly: mfma_terms_dw) and the RAW accumulators go to slab g of `Y`
// ([4][n_out][cout]); k_conv_groups_sum adds the slabs in group order and applies the epilogue.  fp32 rows only.
template <int NB, bool PRE, bool EPI, bool GRP = false>
__global__ void __launch_bounds__(256, (NB <= 2 ? 4 : 2)) k_conv_fwd_dma(const float* __restrict__ X, const u32x4* __restrict__ Wp,
                                                        const int* __restrict__ tbl, const int* __restrict__ order,
                                                        const int* __restrict__ tile_mask, long long n_out, int K,
                                                        int cin, int cout, const float* __restrict__ bias,
                                                        float* __restrict__ Y, int swizzle,
                                                        float* __restrict__ stats, const int* __restrict__ x_amax,
                                                        const int* __restrict__ w_amax, unsigned x_bytes, unsigned w_bytes,
                                                        ConvEpi epi) {
  static_assert(!GRP || (!PRE && !EPI), "offset-group launches: fp32 rows, raw accumulators");
  constexpr int NGRP = GRP ? 4 : 1;
  constexpr int PL = 4, NPL = 2;
  constexpr int BLK = NB * 2 * NPL * 64;                // uint4 per (k, cc) weight block of this workgroup = NB x 4 KB
  constexpr int ISM_H = 15;
  const float a_scale = amax_scale(x_amax);
  const float out_scale = 1.f / (a_scale * amax_scale(w_amax));   // exact: powers of two
  // ONE LDS object (a second one beside an LDS-DMA target makes hipcc wait for the DMA before unrelated reads)
  constexpr int A_BYTES = 4 * 32 * 32 * 4, B_BYTES = 2 * BLK * 16, I_BYTES = 4 * ISM_H * 32 * 4;
  __shared__ __attribute__((aligned(16))) unsigned char smem[A_BYTES + B_BYTES + I_BYTES + 16];
  float (*Asm)[32][32] = reinterpret_cast<float (*)[32][32]>(smem);
  u32x4* const Bsm = reinterpret_cast<u32x4*>(smem + A_BYTES);                       // [2][BLK]
  int (*Ism)[ISM_H][32] = reinterpret_cast<int (*)[ISM_H][32]>(smem + A_BYTES + B_BYTES);
  unsigned* const wmask = reinterpret_cast<unsigned*>(smem + A_BYTES + B_BYTES + I_BYTES);
  const int t = threadIdx.x, l = t & 63, w = t >> 6;
  const int i = l & 31, h = l >> 5;
  unsigned bxx = blockIdx.x, byy = blockIdx.y;
  const bool heavy_first = (swizzle & 16) != 0;
  swizzle &= 15;
  const unsigned nrw = (unsigned)((n_out + CONV_ROWS - 1) / CONV_ROWS);
  unsigned grp = 0u;
  if (swizzle >= 2) {
    const unsigned ncb1 = (unsigned)(cout / (32 * NB)), ncb = ncb1 * (unsigned)NGRP;      // GRP: (group, column block) slots
    const unsigned xcd = bxx & 7u, slot = bxx >> 3;
    byy = slot % ncb;
    if (GRP) {
      grp = byy / ncb1;
      byy -= grp * ncb1;
    }
    if (swizzle == 2) {
      bxx = (slot / ncb) * 8u + xcd;
      if (bxx >= nrw) return;
    } else {
      const unsigned q = nrw >> 3, r = nrw & 7u, mine = q + (xcd < r ? 1u : 0u), j = slot / ncb;
      if (j >= mine) return;
      bxx = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
  } else {
    bxx = xcd_tile(bxx, gridDim.x, swizzle);
  }
  if (heavy_first) bxx = nrw - 1u - bxx;
  if (GRP) Y += (long long)grp * n_out * cout;
  const long long tile = (long long)bxx * 4 + w;
  const long long row0 = tile * 32;
  const bool active = row0 < n_out;
  const int nb0 = byy * NB;
  const int TNB = cout >> 5, CC = cin >> 5;
  const int p = l & 7, rsub = l >> 3;

  f32x16 acc[NB], accx[NB];      // accx: the fp16x3 cross terms (mfma_terms)
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
#pragma unroll
  for (int b = 0; b < (int)(sizeof(accx) / sizeof(accx[0])); ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) accx[b][r] = 0.f;

  unsigned mymask = 0u;
  if (active) mymask = tile_mask ? (unsigned)tile_mask[tile] : ((K >= 32) ? ~0u : ((1u << K) - 1u));
  if (GRP) mymask &= 0x11111111u << grp;      // offsets k with k mod 4 == grp (k_conv_fwd_tall's groups)
  mymask = __builtin_amdgcn_readfirstlane(mymask);
  if (l == 0) wmask[w] = mymask;
  for (int e = l; e < (K < ISM_H ? K : ISM_H) * 32; e += 64) {
    const int k = e >> 5, r = e & 31;
    int v = -1;
    if (active && ((mymask >> k) & 1u) && row0 + r < n_out) v = tbl ? tbl[(long long)k * n_out + row0 + r] : (int)(row0 + r);
    Ism[w][k][(r & 7) * 4 + (r >> 3)] = v;
  }
  int ib[6];
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const int e = 32 * ISM_H + l + 64 * j, k = e >> 5, r = e & 31;
    int v = -1;
    if (k < K && active && ((mymask >> k) & 1u) && row0 + r < n_out) v = tbl[(long long)k * n_out + row0 + r];
    ib[j] = v;
  }
  bool second_half = false;
  __syncthreads();
  const unsigned wgmask = __builtin_amdgcn_readfirstlane(wmask[0] | wmask[1] | wmask[2] | wmask[3]);

  const rsrc_words xrsrc = make_rsrc_words(X, x_bytes), wrsrc = make_rsrc_words(Wp, w_bytes);
  const unsigned row_bytes = (unsigned)cin * 4u;
  // wave-uniform LDS byte addresses (an LDS pointer is its byte offset in the workgroup's allocation)
  const unsigned lds_a = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(&Asm[w][0][0]));
  const unsigned lds_b = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(Bsm + w * 64));
  // a lane's piece inside its four rows (rsub + 8 ps): LDS position p of row r holds source piece p ^ a_swz(r)
  unsigned pc[4];
#pragma unroll
  for (int ps = 0; ps < 4; ++ps) pc[ps] = (unsigned)((p ^ a_swz(rsub + 8 * ps)) << 4);
#define GCL_DMA_A(KK, CCV)                                                                                     \
  {                                                                                                            \
    if ((KK) >= ISM_H && !second_half) {                                                                       \
      second_half = true;                                                                                      \
      WAVE_FENCE();                                                                                            \
      _Pragma("unroll") for (int j_ = 0; j_ < 6; ++j_) {                                                       \
        const int e_ = 32 * ISM_H + l + 64 * j_;                                                               \
        if ((e_ >> 5) < K) Ism[w][(e_ >> 5) - ISM_H][((e_ & 31) & 7) * 4 + ((e_ & 31) >> 3)] = ib[j_];         \
      }                                                                                                        \
      WAVE_FENCE();                                                                                            \
    }                                                                                                          \
    const int4 ri_ = *reinterpret_cast<const int4*>(&Ism[w][(KK) >= ISM_H ? (KK) - ISM_H : (KK)][rsub * 4]);    \
    const unsigned co_ = (unsigned)(CCV)*128u;                                                                 \
    dma16<0>(xrsrc, lds_a, (unsigned)ri_.x * row_bytes + pc[0], co_);                                          \
    dma16<1024>(xrsrc, lds_a, (unsigned)ri_.y * row_bytes + pc[1], co_);                                       \
    dma16<2048>(xrsrc, lds_a, (unsigned)ri_.z * row_bytes + pc[2], co_);                                       \
    dma16<3072>(xrsrc, lds_a, (unsigned)ri_.w * row_bytes + pc[3], co_);                                       \
  }
  // weight block (k, cc) of this workgroup's columns: BLK uint4, lane-linear; wave w moves pieces e * 256 + 64 w .. + 63
#define GCL_DMA_B(KK, CCV, BUF)                                                                                \
  {                                                                                                            \
    const unsigned blk_ = (unsigned)((((KK)*CC + (CCV)) * TNB + nb0) * (2 * NPL * 64)) * 16u;                   \
    const unsigned dst_ = lds_b + (unsigned)(BUF) * (unsigned)(BLK * 16);                                      \
    dma16<0>(wrsrc, dst_, (unsigned)t * 16u, blk_);                                                            \
    if (NB >= 2) dma16<4096>(wrsrc, dst_, (unsigned)t * 16u + 4096u, blk_);                                    \
    if (NB >= 4) {                                                                                             \
      dma16<8192>(wrsrc, dst_, (unsigned)t * 16u + 8192u, blk_);                                               \
      dma16<12288>(wrsrc, dst_, (unsigned)t * 16u + 12288u, blk_);                                             \
    }                                                                                                          \
  }
#define GCL_ADVANCE(KV, CV, HAS)             \
  {                                          \
    CV += 1;                                 \
    if (CV == CC) {                          \
      CV = 0;                                \
      if (m_rest) {                          \
        KV = __builtin_ctz(m_rest);          \
        m_rest &= m_rest - 1;                \
      } else {                               \
        HAS = false;                         \
      }                                      \
    }                                        \
  }
  // every ordinary load of the prologue is consumed HERE: with loads of its own pending hipcc would put vmcnt waits into the
  // loop (it counts the DMAs it cannot see as nothing and would wait in the middle of a DMA sequence)
  asm volatile("" ::"v"(ib[0]), "v"(ib[1]), "v"(ib[2]), "v"(ib[3]), "v"(ib[4]), "v"(ib[5]) : "memory");
  if (wgmask != 0u) {
    unsigned m_rest = wgmask & (wgmask - 1);
    int kc = __builtin_ctz(wgmask), cc = 0;
    bool mine_cur = (mymask >> kc) & 1u;
    if (mine_cur) GCL_DMA_A(kc, 0);
    GCL_DMA_B(kc, 0, 0);
    int buf = 0;
    while (true) {
      dma_wait_all();       // this wave's DMAs have landed: A(s) in its tile, its part of B(s)
      __syncthreads();      // everybody's part of B(s) is in LDS; everybody's reads of step s-1 are done
      int kn = kc, cn = cc;
      bool hasn = true;
      GCL_ADVANCE(kn, cn, hasn);
      const bool mine_n = hasn && ((mymask >> kn) & 1u);
      if (hasn) GCL_DMA_B(kn, cn, buf ^ 1);
      if (mine_cur) {
        u32x4 ap[2][2];
        if (PRE) {   // row image: dwords [0,16) = hi of channels 0..31, [16,32) = lo
#pragma unroll
          for (int m = 0; m < 2; ++m) {
            ap[m][0] = *reinterpret_cast<const u32x4*>(&Asm[w][i][((2 * m + h) ^ a_swz(i)) << 2]);
            ap[m][1] = *reinterpret_cast<const u32x4*>(&Asm[w][i][((4 + 2 * m + h) ^ a_swz(i)) << 2]);
          }
          if (mine_n) {
            // the DMA overwrites the tile: this wave's fragment reads must have returned (the asm consumes them)
            asm volatile("" : "+v"(ap[0][0]), "+v"(ap[0][1]), "+v"(ap[1][0]), "+v"(ap[1][1])::"memory");
            GCL_DMA_A(kn, cn);
          }
        } else {     // fp32 rows: the raw pieces come out of the tile and are split into the two fp16 planes here
          typedef float f32x4 __attribute__((ext_vector_type(4)));
          f32x4 f[2][2];
#pragma unroll
          for (int m = 0; m < 2; ++m) {
            f[m][0] = *reinterpret_cast<const f32x4*>(&Asm[w][i][((4 * m + 2 * h) ^ a_swz(i)) << 2]);
            f[m][1] = *reinterpret_cast<const f32x4*>(&Asm[w][i][((4 * m + 2 * h + 1) ^ a_swz(i)) << 2]);
          }
          if (mine_n) {
            asm volatile("" : "+v"(f[0][0]), "+v"(f[0][1]), "+v"(f[1][0]), "+v"(f[1][1])::"memory");
            GCL_DMA_A(kn, cn);
          }
          if (!GRP) {
#pragma unroll
            for (int m = 0; m < 2; ++m)
              split8<PL>(make_float4(f[m][0][0], f[m][0][1], f[m][0][2], f[m][0][3]),
                         make_float4(f[m][1][0], f[m][1][1], f[m][1][2], f[m][1][3]), a_scale, ap[m]);
          } else {      // k_conv_fwd_tall's products, in its order: m, then the column blocks
#pragma unroll
            for (int m = 0; m < 2; ++m) {
              u32x4 ad[3];
              split8<PL>(make_float4(f[m][0][0], f[m][0][1], f[m][0][2], f[m][0][3]),
                         make_float4(f[m][1][0], f[m][1][1], f[m][1][2], f[m][1][3]), a_scale, ad);
              dw_a_planes(ad);
#pragma unroll
              for (int b = 0; b < NB; ++b) {
                const u32x4* bb = &Bsm[buf * BLK + ((b * 2 + m) * NPL) * 64 + l];
                u32x4 bp[3];
                bp[0] = bb[0];
                bp[1] = bb[64];
                mfma_terms_dw<PL>(ad, bp, acc[b]);
              }
            }
          }
        }
        if (!GRP) {
#pragma unroll
          for (int m = 0; m < 2; ++m) {
#pragma unroll
            for (int b = 0; b < NB; ++b) {
              const u32x4* bb = &Bsm[buf * BLK + ((b * 2 + m) * NPL) * 64 + l];
              u32x4 bp[2];
              bp[0] = bb[0];
              bp[1] = bb[64];
              mfma_terms<PL>(ap[m], bp, acc[b], accx[b]);
            }
          }
        }
      } else if (mine_n) {
        GCL_DMA_A(kn, cn);
      }
      if (!hasn) break;
      buf ^= 1;
      kc = kn;
      cc = cn;
      mine_cur = mine_n;
    }
  }
#undef GCL_DMA_A
#undef GCL_DMA_B
#undef GCL_ADVANCE
  if (!GRP) {
#pragma unroll
    for (int b = 0; b < NB; ++b) fold_cross<PL>(acc[b], accx[(PL == 4) ? b : 0]);
  }
  conv_fwd_epilogue<NB, EPI>(acc, &Asm[0][0][0], order, n_out, cout, bias, Y, stats, GRP ? 1.f : out_scale, epi, tile, bxx, nb0,
                             active);
}

// y = epilogue(((slab 0 + slab 1) + slab 2) + slab 3): k_conv_fwd_tall's group order and its epilogue expression, on the raw
// accumulator slabs the four offset-group launches of k_conv_fwd_dma<.., GRP> leave ([4][n][cout], rows in the output's order)
template <bool EPI, int ILP = 1>
__global__ void __launch_bounds__(256) k_conv_groups_sum(const float* __restrict__ slabs, long long n, int cout,
                                                         const float* __restrict__ bias, const int* __restrict__ x_amax,
                                                         const int* __restrict__ w_amax, ConvEpi epi, float* __restrict__ Y) {
  const float out_scale = 1.f / (amax_scale(x_amax) * amax_scale(w_amax));
  const long long total = n * cout;
  // ILP float4 groups per thread, a grid's width apart (coalesced), all sixteen slab loads in flight before the first sum
  const long long stride = (long long)gridDim.x * 1024;
  const long long e0 = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  float4 pp[ILP][4];
#pragma unroll
  for (int q = 0; q < ILP; ++q) {
    const long long e = e0 + q * stride;
    if (e < total) {
#pragma unroll
      for (int g = 0; g < 4; ++g) pp[q][g] = *reinterpret_cast<const float4*>(slabs + g * total + e);
    }
  }
  float ymax = 0.f;
#pragma unroll
  for (int q = 0; q < ILP; ++q) {
    const long long e = e0 + q * stride;
    if (e < total) {
      const long long row = e / cout;
      const int col = (int)(e - row * cout);
      const float4 p0 = pp[q][0], p1 = pp[q][1], p2 = pp[q][2], p3 = pp[q][3];
      const float a[4] = {((p0.x + p1.x) + p2.x) + p3.x, ((p0.y + p1.y) + p2.y) + p3.y, ((p0.z + p1.z) + p2.z) + p3.z,
                          ((p0.w + p1.w) + p2.w) + p3.w};
      float v4[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float bvv = bias ? bias[col + j] : 0.f;
        const float csc = (EPI && epi.col_scale) ? epi.col_scale[col + j] * out_scale : out_scale;
        float v = a[j] * csc + bvv;
        if (EPI && epi.residual) {
          const float rsd = epi.residual[row * (epi.residual_ld ? epi.residual_ld : cout) + col + j];
          v = epi.relu == 2 ? (rsd > 0.f ? v : 0.f) : v + rsd;
        }
        if (EPI && epi.relu == 1) v = fmaxf(v, 0.f);
        v4[j] = v;
        if (EPI) ymax = fmaxf(ymax, fabsf(v));
      }
      *reinterpret_cast<float4*>(Y + e) = make_float4(v4[0], v4[1], v4[2], v4[3]);
    }
  }
  if (EPI && epi.y_amax) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ymax = fmaxf(ymax, __shfl_xor(ymax, o));
    if ((threadIdx.x & 63) == 0) amax_slot_publish(epi.y_amax, __float_as_int(ymax), blockIdx.x * 4u + (threadIdx.x >> 6));
  }
}

// ---------------------------------------------------------------------------------------------------
// ---- inference on small clouds: SIXTEEN waves per workgroup, the offsets of a tile cut into four fixed groups -----------
// A pass over one or two clouds launches a handful of workgroups on the deep layers (17 k voxels: 28 row tiles at stride 4, 8
// at stride 8), and each of them walks up to 27 offsets x Cin / 32 dependent steps alone on its CU: 9 launches of ~136 us
// were half of a 2.4 ms pass.  Here four groups of four waves share the 128 rows of a tile; group g runs the offsets
// k with k mod 4 == g (FIXED by the offset index, not by the tile's mask; interleaved, so that the groups of a tile are
// about equally long: with the ranges 4 k / K == g eight pairs per pass ran at 77 instead of 80 M voxels/s) exactly like a k_conv_fwd_split
// workgroup -- own weight blocks, own wave-private A tiles -- and the four partial accumulators are added through LDS in
// group order: y = ((g0 + g1) + g2) + g3.  A row's result therefore depends on the layer's shape only, never on the rows
// it shares a tile or a batch with (batching clouds stays bitwise neutral), and nothing leaves the CU (no scratch, no
// atomics).  fp16x3 on fp32 rows (split in the kernel), 64 columns per workgroup, no BatchNorm statistics: the inference
// launches.  141.5 KB of LDS: one workgroup = 16 waves per CU, as many waves as four k_conv_fwd_split workgroups.
template <bool EPI>
__global__ void __launch_bounds__(1024, 1) k_conv_fwd_tall(const float* __restrict__ X, const u32x4* __restrict__ Wp,
                                                          const int* __restrict__ tbl, const int* __restrict__ order,
                                                          const int* __restrict__ tile_mask, long long n_out, int K, int cin,
                                                          int cout, const float* __restrict__ bias, float* __restrict__ Y,
                                                          int swizzle, const int* __restrict__ x_amax,
                                                          const int* __restrict__ w_amax, unsigned x_bytes, ConvEpi epi) {
  constexpr int PL = 4, NPL = 2, NB = 2, G = 4;
  constexpr int BLK = NB * 2 * NPL * 64;                // 512 uint4 per (k, cc) weight block of a group
  const float a_scale = amax_scale(x_amax);
  const float out_scale = 1.f / (a_scale * amax_scale(w_amax));
  // [A tiles: 16 waves x 4 KB][weight blocks: 4 groups x 2 buffers x 8 KB]; after the loop the same 128 KB hold the partial
  // accumulators of groups 1 - 3 (96 KB)
  __shared__ __attribute__((aligned(16))) float lds[16 * 1024 + G * 2 * BLK * 4];
  __shared__ __attribute__((aligned(16))) int Ism[4][27][32];
  __shared__ unsigned wmask[G][4];
  float (*Asm)[32][32] = reinterpret_cast<float (*)[32][32]>(lds);
  u32x4* const Ball = reinterpret_cast<u32x4*>(lds + 16 * 1024);
  const int t = threadIdx.x, l = t & 63, w16 = t >> 6, wt = w16 & 3, g = w16 >> 2, tg = t & 255;
  const int i = l & 31, h = l >> 5;
  u32x4* const Bg = Ball + g * 2 * BLK;
  unsigned bxx = blockIdx.x, byy;
  const bool heavy_first = (swizzle & 16) != 0;
  const unsigned nrw = (unsigned)((n_out + CONV_ROWS - 1) / CONV_ROWS);
  {
    const unsigned ncb = (unsigned)(cout / (32 * NB));
    const unsigned xcd = bxx & 7u, slot = bxx >> 3;
    byy = slot % ncb;
    bxx = (slot / ncb) * 8u + xcd;
    if (bxx >= nrw) return;
  }
  if (heavy_first) bxx = nrw - 1u - bxx;
  const long long tile = (long long)bxx * 4 + wt;
  const long long row0 = tile * 32;
  const bool active = row0 < n_out;
  const int nb0 = byy * NB;
  const int TNB = cout >> 5, CC = cin >> 5;
  const int p = l & 7, rsub = l >> 3;
  // offsets of group g: k with k mod 4 == g
  unsigned gsel = 0u;
  for (int k = 0; k < K; ++k)
    if ((k & 3) == g) gsel |= 1u << k;

  // ONE accumulator set (1024 threads per workgroup leave 128 registers per wave): both operands of an inference launch
  // are narrow-range (activations, weights), so the activation planes are re-scaled on the fly as in the weight-gradient
  // kernels (mfma_terms_dw) instead of summing the cross terms apart
  f32x16 acc[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;

  unsigned tmask = 0u;
  if (active) tmask = tile_mask ? (unsigned)tile_mask[tile] : ((1u << K) - 1u);
  tmask = __builtin_amdgcn_readfirstlane(tmask);
  const unsigned mymask = tmask & gsel;
  if (l == 0) wmask[g][wt] = mymask;
  for (int e = l + 64 * g; e < K * 32; e += 64 * G) {      // the four waves of a tile share the index loads
    const int k = e >> 5, r = e & 31;
    int v = -1;
    if (active && ((tmask >> k) & 1u) && row0 + r < n_out) v = tbl[(long long)k * n_out + row0 + r];
    Ism[wt][k][(r & 7) * 4 + (r >> 3)] = v;
  }
  __syncthreads();
  const unsigned wgmask = wmask[g][0] | wmask[g][1] | wmask[g][2] | wmask[g][3];
  int n_iter = 0;      // barriers every wave takes: the longest group's steps
#pragma unroll
  for (int q = 0; q < G; ++q) {
    const int c = __builtin_popcount(wmask[q][0] | wmask[q][1] | wmask[q][2] | wmask[q][3]) * CC;
    n_iter = c > n_iter ? c : n_iter;
  }
  const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)X, 0, (int)x_bytes, 0x00020000);
  const unsigned row_bytes = (unsigned)cin * 4u;
#define GCLT_GATHER_A(KK, CCV)                                                                                  \
  {                                                                                                             \
    const int4 ri_ = *reinterpret_cast<const int4*>(&Ism[wt][(KK)][rsub * 4]);                                 \
    const unsigned co_ = (unsigned)(CCV)*128u + (unsigned)p * 16u;                                              \
    st[0] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, (int)((unsigned)ri_.x * row_bytes + co_), 0, 0)); \
    st[1] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, (int)((unsigned)ri_.y * row_bytes + co_), 0, 0)); \
    st[2] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, (int)((unsigned)ri_.z * row_bytes + co_), 0, 0)); \
    st[3] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, (int)((unsigned)ri_.w * row_bytes + co_), 0, 0)); \
  }
#define GCLT_LOAD_B(KK, CCV)                                                                     \
  {                                                                                              \
    const u32x4* src_ = Wp + (((long long)(KK)*CC + (CCV)) * TNB + nb0) * (2 * NPL * 64);        \
    br[0] = src_[tg];                                                                            \
    br[1] = src_[256 + tg];                                                                      \
  }
#define GCLT_STORE_LDS(MINE, BUF)                                                                            \
  {                                                                                                          \
    if (MINE) {                                                                                              \
      _Pragma("unroll") for (int ps = 0; ps < 4; ++ps)                                                       \
          *reinterpret_cast<float4*>(&Asm[w16][rsub + 8 * ps][(p ^ a_swz(rsub + 8 * ps)) << 2]) = st[ps];    \
    }                                                                                                        \
    Bg[(BUF)*BLK + tg] = br[0];                                                                              \
    Bg[(BUF)*BLK + 256 + tg] = br[1];                                                                        \
  }
#define GCLT_ADVANCE(KV, CV, HAS)            \
  {                                          \
    CV += 1;                                 \
    if (CV == CC) {                          \
      CV = 0;                                \
      if (m_rest) {                          \
        KV = __builtin_ctz(m_rest);          \
        m_rest &= m_rest - 1;                \
      } else {                               \
        HAS = false;                         \
      }                                      \
    }                                        \
  }
  if (n_iter > 0) {
    unsigned m_rest = wgmask ? (wgmask & (wgmask - 1)) : 0u;
    float4 st[4];
    u32x4 br[2];
    bool has0 = wgmask != 0u;
    int k0 = has0 ? __builtin_ctz(wgmask) : 0;
    bool mine0 = has0 && ((mymask >> k0) & 1u);
    int k1 = k0, c1 = 0;
    bool has1 = has0;
    if (has0) {
      if (mine0) GCLT_GATHER_A(k0, 0);
      GCLT_LOAD_B(k0, 0);
      GCLT_ADVANCE(k1, c1, has1);
      GCLT_STORE_LDS(mine0, 0);
    }
    bool mine1 = false;
    if (has1) {
      mine1 = (mymask >> k1) & 1u;
      if (mine1) GCLT_GATHER_A(k1, c1);
      GCLT_LOAD_B(k1, c1);
    }
    __syncthreads();
    int buf = 0;
    bool mine_cur = mine0;
    for (int it = 0; it < n_iter; ++it) {      // every wave takes n_iter barriers; a group that has run out of steps idles
      if (mine_cur) {
        WAVE_FENCE();
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          u32x4 ap[3];
          float4 f0 = *reinterpret_cast<const float4*>(&Asm[w16][i][((4 * m + 2 * h) ^ a_swz(i)) << 2]);
          float4 f1 = *reinterpret_cast<const float4*>(&Asm[w16][i][((4 * m + 2 * h + 1) ^ a_swz(i)) << 2]);
          split8<PL>(f0, f1, a_scale, ap);
          dw_a_planes(ap);
#pragma unroll
          for (int b = 0; b < NB; ++b) {
            const u32x4* bb = &Bg[buf * BLK + ((b * 2 + m) * NPL) * 64 + l];
            u32x4 bp[3];
            bp[0] = bb[0];
            bp[1] = bb[64];
            mfma_terms_dw<PL>(ap, bp, acc[b]);
          }
        }
      }
      if (has1) {      // next step: registers -> LDS (A: own tile, after this wave's reads; B: the group's other buffer)
        WAVE_FENCE();
        GCLT_STORE_LDS(mine1, buf ^ 1);
      }
      int k2 = k1, c2 = c1;
      bool has2 = has1, mine2 = false;
      if (has1) GCLT_ADVANCE(k2, c2, has2);
      if (has2) {
        mine2 = (mymask >> k2) & 1u;
        if (mine2) GCLT_GATHER_A(k2, c2);
        GCLT_LOAD_B(k2, c2);
      }
      __syncthreads();
      buf ^= 1;
      mine_cur = has1 && mine1;
      mine1 = mine2;
      k1 = k2;
      c1 = c2;
      has1 = has2;
    }
  }
#undef GCLT_GATHER_A
#undef GCLT_LOAD_B
#undef GCLT_STORE_LDS
#undef GCLT_ADVANCE
  // partial accumulators of groups 1 - 3 through LDS (the loop's last barrier has passed: tiles and weight blocks are free)
  float* const red = lds;      // [g - 1][wt][b * 16 + r][64 lanes]
  if (g > 0) {
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) red[(((g - 1) * 4 + wt) * (NB * 16) + b * 16 + r) * 64 + l] = acc[b][r];
  }
  __syncthreads();
  if (g > 0 || !active) return;
#pragma unroll
  for (int q = 0; q < G - 1; ++q)      // group order: ((g0 + g1) + g2) + g3
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[b][r] += red[((q * 4 + wt) * (NB * 16) + b * 16 + r) * 64 + l];
  int orow_l = -1;
  if ((l < 32) && (row0 + l < n_out)) orow_l = order ? order[row0 + l] : (int)(row0 + l);
  float ymax = 0.f;
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const int col = (nb0 + b) * 32 + i;
    float bvv = bias ? bias[col] : 0.f;
    float csc = (EPI && epi.col_scale) ? epi.col_scale[col] * out_scale : out_scale;
    asm volatile("v_mov_b32 %0, %1" : "=v"(bvv) : "v"(bvv));
    asm volatile("v_mov_b32 %0, %1" : "=v"(csc) : "v"(csc));
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int orow = __shfl(orow_l, (r & 3) + 8 * (r >> 2) + 4 * h);
      if (orow >= 0) {
        float v = acc[b][r] * csc + bvv;
        if (EPI && epi.residual) {
          const float rsd = epi.residual[(long long)orow * (epi.residual_ld ? epi.residual_ld : cout) + col];
          v = epi.relu == 2 ? (rsd > 0.f ? v : 0.f) : v + rsd;      // relu == 2: threshold_backward, pass v where residual > 0
        }
        if (EPI && epi.relu == 1) v = fmaxf(v, 0.f);
        Y[(long long)orow * cout + col] = v;
        if (EPI) ymax = fmaxf(ymax, fabsf(v));
      }
    }
  }
  if (EPI && epi.y_amax) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ymax = fmaxf(ymax, __shfl_xor(ymax, o));
    if (l == 0) amax_slot_publish(epi.y_amax, __float_as_int(ymax), (unsigned)tile);
  }
}

// ---------------------------------------------------------------------------------------------------
// weight gradient over compacted pair lists
// ---------------------------------------------------------------------------------------------------
struct SegOffW {
  long long off[130];
};

// Flush of the weight-gradient accumulators: the four waves of a workgroup hold partial sums over disjoint pairs;
// they are added in wave order through LDS (deterministic) and ONE slab per (workgroup, offset) goes to memory.
template <int TCA, int TCB>
__device__ __forceinline__ void bwd_weight_flush(f32x16 (&acc)[TCA / 32][TCB / 32], float* lds, float* slab, int ca0,
                                                 int cb0, int cb) {
  constexpr int NBI = TCA / 32, NBJ = TCB / 32;
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();   // every wave is done with its gather tiles
#pragma unroll
  for (int a = 0; a < NBI; ++a)
#pragma unroll
    for (int b = 0; b < NBJ; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        lds[w * (TCA * TCB) + ((a * NBJ + b) * 16 + r) * 64 + l] = acc[a][b][r];
        acc[a][b][r] = 0.f;
      }
  __syncthreads();
  for (int e = threadIdx.x; e < TCA * TCB; e += 256) {
    float v = lds[e] + lds[TCA * TCB + e] + lds[2 * TCA * TCB + e] + lds[3 * TCA * TCB + e];
    int ll = e & 63, r = (e >> 6) & 15, blk = e >> 10;
    int a = blk / NBJ, b = blk % NBJ;
    int row = ca0 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * (ll >> 5);
    slab[(long long)row * cb + cb0 + b * 32 + (ll & 31)] = v;
  }
  __syncthreads();   // LDS goes back to the gather tiles
}

template <int TCA, int TCB>
__global__ void __launch_bounds__(256) k_conv_bwd_weight(const float* __restrict__ A, const float* __restrict__ B,
                                                         const int* __restrict__ pair_a,
                                                         const int* __restrict__ pair_b, SegOffW seg, int K, int ca,
                                                         int cb, long long n_chunks, int per, float* slabs) {
  constexpr int NBI = TCA / 32, NBJ = TCB / 32;
  constexpr int PA = TCA / 4, PB = TCB / 4;         // 16-byte pieces per row
  constexpr int RA = 64 / PA, RB = 64 / PB;          // rows per load pass
  __shared__ __attribute__((aligned(16))) float lds_all[4 * 32 * (TCA + TCB)];
  float (*As)[32][TCA] = reinterpret_cast<float (*)[32][TCA]>(lds_all);
  float (*Bs)[32][TCB] = reinterpret_cast<float (*)[32][TCB]>(lds_all + 4 * 32 * TCA);
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int i = l & 31, h = l >> 5;
  const int tiles_b = cb / TCB;
  const int ca0 = (blockIdx.y / tiles_b) * TCA, cb0 = (blockIdx.y % tiles_b) * TCB;
  const long long c0 = (long long)blockIdx.x * per;
  const long long c1 = (c0 + per < n_chunks) ? c0 + per : n_chunks;
  if (c0 >= c1) return;   // uniform over the workgroup

  f32x16 acc[NBI][NBJ];
#pragma unroll
  for (int a = 0; a < NBI; ++a)
#pragma unroll
    for (int b = 0; b < NBJ; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  auto flush = [&](int k) {
    bwd_weight_flush<TCA, TCB>(acc, lds_all, slabs + (long long)(blockIdx.x + k) * ((long long)ca * cb), ca0, cb0, cb);
  };

  int kcur = 0;
  while (seg.off[kcur + 1] <= c0 * GCL_PAIR_CHUNK) ++kcur;
  for (long long c = c0; c < c1; ++c) {
    const long long pbase = c * GCL_PAIR_CHUNK;
    if (pbase >= seg.off[kcur + 1]) {
      flush(kcur);
      while (seg.off[kcur + 1] <= pbase) ++kcur;
    }
    const long long p0 = pbase + w * 32;
    int ia = -1, ib = -1;
    if (l < 32) {
      ia = pair_a[p0 + l];
      ib = pair_b[p0 + l];
    }
    WAVE_FENCE();
#pragma unroll
    for (int ps = 0; ps < 32 / RA; ++ps) {
      int r = l / PA + RA * ps;
      int ridx = __shfl(ia, r);
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ridx >= 0) v = *reinterpret_cast<const float4*>(A + (long long)ridx * ca + ca0 + (l % PA) * 4);
      *reinterpret_cast<float4*>(&As[w][r][(l % PA) * 4]) = v;
    }
#pragma unroll
    for (int ps = 0; ps < 32 / RB; ++ps) {
      int r = l / PB + RB * ps;
      int ridx = __shfl(ib, r);
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ridx >= 0) v = *reinterpret_cast<const float4*>(B + (long long)ridx * cb + cb0 + (l % PB) * 4);
      *reinterpret_cast<float4*>(&Bs[w][r][(l % PB) * 4]) = v;
    }
    WAVE_FENCE();
#pragma unroll 4
    for (int s = 0; s < 16; ++s) {
      float av[NBI], bv[NBJ];
#pragma unroll
      for (int a = 0; a < NBI; ++a) av[a] = As[w][2 * s + h][a * 32 + i];
#pragma unroll
      for (int b = 0; b < NBJ; ++b) bv[b] = Bs[w][2 * s + h][b * 32 + i];
#pragma unroll
      for (int a = 0; a < NBI; ++a)
#pragma unroll
        for (int b = 0; b < NBJ; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a], bv[b], acc[a][b], 0, 0, 0);
    }
  }
  flush(kcur);
}

// ---- weight gradient on PRE-SPLIT operands (fp16 plane images from gcl_split_planes) -----------------------------
// dW = A^T B wants K-major fragments (8 consecutive PAIRS of one channel per lane) while rows are channel-major.  With
// fp32 rows that is 8 scalar LDS reads + a split per fragment; with plane images the gathered 16-byte pieces are copied
// into LDS unchanged and a fragment is two hardware-transposed 8-byte reads per plane (gfx950 ds_read_b64_tr_b16):
// no VALU work at all between the gather and the MFMA.  LDS row image = the row's TC * 4 bytes, 32-byte blocks
// [slice][plane][16-channel half] XOR-swizzled with the row so that the 4 rows x 2 column halves a 32-lane half reads
// hit 64 different banks.
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
template <int IMM>
__device__ __forceinline__ u32x2 lds_read_tr16(unsigned base) {      // one base VGPR, compile-time byte offset
  u32x2 r;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(base), "n"(IMM) : "memory");
  return r;
}
template <int TC>
__device__ __forceinline__ int tr_swz(int row) { return (TC == 64) ? ((row & 3) << 1) : (((row >> 1) & 1) << 1); }
// both planes of the fragment of 32-channel slice (base addresses precomputed per lane) for the 16-pair half HALF
template <int TC, int HALF, int EXTRA = 0>      // EXTRA: constant byte offset added to every read (a second tile buffer)
__device__ __forceinline__ void tr_fragment(unsigned base_hi, unsigned base_lo, u32x4* f) {
  constexpr int ROWB = TC * 4;
  u32x2 h0 = lds_read_tr16<(16 * HALF) * ROWB + EXTRA>(base_hi), h1 = lds_read_tr16<(16 * HALF + 4) * ROWB + EXTRA>(base_hi);
  u32x2 l0 = lds_read_tr16<(16 * HALF) * ROWB + EXTRA>(base_lo), l1 = lds_read_tr16<(16 * HALF + 4) * ROWB + EXTRA>(base_lo);
  f[0] = u32x4{h0.x, h0.y, h1.x, h1.y};
  f[1] = u32x4{l0.x, l0.y, l1.x, l1.y};
}
template <int TCA, int TCB, int HALF, int EXTRA = 0>
__device__ __forceinline__ void tr_mma_half(const unsigned (*baseA)[2], const unsigned (*baseB)[2],
                                            f32x16 (&acc)[TCA / 32][TCB / 32]) {
  constexpr int NBI = TCA / 32, NBJ = TCB / 32;
  u32x4 fa[NBI][3], fb[NBJ][2];
#pragma unroll
  for (int a = 0; a < NBI; ++a) tr_fragment<TCA, HALF, EXTRA>(baseA[a][0], baseA[a][1], fa[a]);
#pragma unroll
  for (int b = 0; b < NBJ; ++b) tr_fragment<TCB, HALF, EXTRA>(baseB[b][0], baseB[b][1], fb[b]);
  // the asm reads complete asynchronously and the compiler does not know: wait, then pin every fragment behind the
  // wait with an empty asm (volatile asms keep their order) so that no MFMA is scheduled above it
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
  for (int a = 0; a < NBI; ++a) {
    asm volatile("" : "+v"(fa[a][0]));
    asm volatile("" : "+v"(fa[a][1]));
  }
#pragma unroll
  for (int b = 0; b < NBJ; ++b) {
    asm volatile("" : "+v"(fb[b][0]));
    asm volatile("" : "+v"(fb[b][1]));
  }
  // a outermost: one slice's re-scaled planes are live at a time (the 128 x 128 kernel has no registers to spare)
#pragma unroll
  for (int a = 0; a < NBI; ++a) {
    dw_a_planes(fa[a]);
#pragma unroll
    for (int b = 0; b < NBJ; ++b) mfma_terms_dw<4>(fa[a], fb[b], acc[a][b]);
  }
}

// split-precision weight gradient: both operands are gathered activations, split on the fly into PL bf16 planes.
// Same decomposition as k_conv_bwd_weight (wave-private LDS tiles, per-wave slabs, no workgroup barrier), plus a
// two-deep software pipeline: pair indices are fetched two chunks ahead and rows one chunk ahead of their use.
// PRE (PL == 4 only): A and B point at plane images; see above.
// RG ("range-grouped", round 3): instead of a contiguous run of the k-major pair list, a workgroup takes ONE (row range j,
// offset k) cell -- the pairs of offset k whose row on the SORTED side of the list (the out rows of the kernel map) lies in
// [j RR, (j+1) RR); `bounds` holds the cell limits.  The K cells of a range get consecutive slots of ONE XCD, so the
// sorted side's rows (dY for a forward convolution) are fetched into that L2 once and re-read K - 1 times from there
// instead of once per offset from the fabric: P (Ca + Cb) -> ~ P Ca + N Cb bytes.  One slab per cell, summed over the
// ranges in order by k_bwd_weight_reduce_rg (deterministic).  Single channel tile only (Ca = TCA, Cb = TCB).
template <int TCA, int TCB, int PL, bool PRE = false, bool RG = false>
__global__ void __launch_bounds__(256) k_conv_bwd_weight_split(const float* __restrict__ A, const float* __restrict__ B,
                                                               const int* __restrict__ pair_a,
                                                               const int* __restrict__ pair_b, SegOffW seg, int K,
                                                               int ca, int cb, long long n_chunks, int per,
                                                               float* slabs, const int* __restrict__ a_amax,
                                                               const int* __restrict__ b_amax, int n_wg_x, int n_tiles,
                                                               unsigned a_bytes, unsigned b_bytes,
                                                               const int* __restrict__ bounds = nullptr, int n_ranges = 0) {
  constexpr int NBI = TCA / 32, NBJ = TCB / 32;
  constexpr int PA = TCA / 4, PB = TCB / 4;         // 16-byte pieces per row
  constexpr int RA = 64 / PA, RB = 64 / PB;          // rows per load pass
  constexpr int NPA = 32 / RA, NPB = 32 / RB;        // load passes per 32-pair tile
  const float sa = (PL == 4) ? amax_scale(a_amax) : 1.f, sb = (PL == 4) ? amax_scale(b_amax) : 1.f;
  const float out_scale = 1.f / (sa * sb);
  __shared__ __attribute__((aligned(16))) float lds_all[4 * 32 * (TCA + TCB)];
  float (*As)[32][TCA] = reinterpret_cast<float (*)[32][TCA]>(lds_all);
  float (*Bs)[32][TCB] = reinterpret_cast<float (*)[32][TCB]>(lds_all + 4 * 32 * TCA);
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int i = l & 31, h = l >> 5;
  // PRE: wave-private row images (same bytes as the fp32 tiles); lane 16 g + 4 q + p of a transposed read supplies
  // row q (+ 8 in the upper half of the wave), 16-channel half g & 1, 8-byte piece p
  unsigned char* const imgA = reinterpret_cast<unsigned char*>(&As[w][0][0]);
  unsigned char* const imgB = reinterpret_cast<unsigned char*>(&Bs[w][0][0]);
  unsigned trA[NBI][2], trB[NBJ][2];
  if (PRE) {
    const unsigned ldsA = (unsigned)(unsigned long long)((__attribute__((address_space(3))) unsigned char*)imgA);
    const unsigned ldsB = (unsigned)(unsigned long long)((__attribute__((address_space(3))) unsigned char*)imgB);
    const int q = (l & 15) >> 2, pp = l & 3, c16 = (l >> 4) & 1, trow = 8 * (l >> 5) + q;
#pragma unroll
    for (int a = 0; a < NBI; ++a)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl)
        trA[a][pl] = ldsA + trow * (TCA * 4) + (((a * 4 + pl * 2 + c16) ^ tr_swz<TCA>(q)) * 32) + pp * 8;
#pragma unroll
    for (int b = 0; b < NBJ; ++b)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl)
        trB[b][pl] = ldsB + trow * (TCB * 4) + (((b * 4 + pl * 2 + c16) ^ tr_swz<TCB>(q)) * 32) + pp * 8;
  }
  // XCD-aware launch order (1-D grid): workgroups are dealt round-robin to the 8 XCDs, so the n_tiles channel tiles
  // that re-read the SAME pair range are given consecutive slots of ONE XCD and share its L2
  int bx, by;
  long long rg_lo = 0, rg_hi = 0;
  int rg_k = 0;
  if (RG) {
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    rg_k = slot % K;
    const int j = (slot / K) * 8 + xcd;
    if (j >= n_ranges) return;
    rg_lo = bounds[rg_k * (n_ranges + 1) + j];
    rg_hi = bounds[rg_k * (n_ranges + 1) + j + 1];
    bx = j * K;              // slab of the cell = bx + rg_k
    by = 0;
  } else if (n_tiles > 0) {
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    by = slot % n_tiles;
    bx = (slot / n_tiles) * 8 + xcd;
    if (bx >= n_wg_x) return;
  } else {
    bx = blockIdx.x;
    by = blockIdx.y;
  }
  const int tiles_b = cb / TCB;
  const int ca0 = (by / tiles_b) * TCA, cb0 = (by % tiles_b) * TCB;
  const long long c0 = RG ? 0 : (long long)bx * per;
  const long long c1 = RG ? (rg_hi - rg_lo + GCL_PAIR_CHUNK - 1) / GCL_PAIR_CHUNK
                          : ((c0 + per < n_chunks) ? c0 + per : n_chunks);
  if (!RG && c0 >= c1) return;   // uniform over the workgroup (an empty RG cell still writes its zero slab)

  f32x16 acc[NBI][NBJ];
#pragma unroll
  for (int a = 0; a < NBI; ++a)
#pragma unroll
    for (int b = 0; b < NBJ; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  auto flush = [&](int k) {
    if (PL == 4) {
#pragma unroll
      for (int a = 0; a < NBI; ++a)
#pragma unroll
        for (int b = 0; b < NBJ; ++b)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[a][b][r] *= out_scale;
    }
    bwd_weight_flush<TCA, TCB>(acc, lds_all, slabs + (long long)(bx + k) * ((long long)ca * cb), ca0, cb0, cb);
  };
  auto load_pairs = [&](long long c, int& ia, int& ib) {
    ia = -1;
    ib = -1;
    if (c < c1 && l < 32) {
      long long p0 = (RG ? rg_lo : 0) + c * GCL_PAIR_CHUNK + w * 32 + l;
      if (!RG || p0 < rg_hi) {
        ia = pair_a[p0];
        ib = pair_b[p0];
      }
    }
  };

  float4 ga[NPA], gb[NPB];
  // rows through buffer resources: a padding pair (row -1) wraps beyond the tensor and reads zeros (no branch, no
  // 64-bit address arithmetic per row)
  const __amdgpu_buffer_rsrc_t arsrc = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (int)a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t brsrc = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, (int)b_bytes, 0x00020000);
  const unsigned a_row = (unsigned)ca * 4u, b_row = (unsigned)cb * 4u;
  const unsigned a_col = (unsigned)(ca0 + (l % PA) * 4) * 4u, b_col = (unsigned)(cb0 + (l % PB) * 4) * 4u;
#define GCL_GATHER(IA, IB)                                                                              \
  {                                                                                                     \
    _Pragma("unroll") for (int ps = 0; ps < NPA; ++ps) {                                                \
      unsigned ridx = (unsigned)__shfl(IA, l / PA + RA * ps);                                           \
      ga[ps] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(arsrc, (int)(ridx * a_row + a_col), 0, 0)); \
    }                                                                                                   \
    _Pragma("unroll") for (int ps = 0; ps < NPB; ++ps) {                                                \
      unsigned ridx = (unsigned)__shfl(IB, l / PB + RB * ps);                                           \
      gb[ps] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(brsrc, (int)(ridx * b_row + b_col), 0, 0)); \
    }                                                                                                   \
  }

  int kcur = RG ? rg_k : 0;
  if (!RG)
    while (seg.off[kcur + 1] <= c0 * GCL_PAIR_CHUNK) ++kcur;
  int ia1, ib1, ia2, ib2;
  load_pairs(c0, ia1, ib1);
  load_pairs(c0 + 1, ia2, ib2);
  GCL_GATHER(ia1, ib1);
  for (long long c = c0; c < c1; ++c) {
    const long long pbase = c * GCL_PAIR_CHUNK;
    if (!RG && pbase >= seg.off[kcur + 1]) {
      flush(kcur);
      while (seg.off[kcur + 1] <= pbase) ++kcur;
    }
    WAVE_FENCE();
    if (PRE) {   // 16-byte pieces go to LDS unchanged, 32-byte blocks swizzled with the row
#pragma unroll
      for (int ps = 0; ps < NPA; ++ps) {
        const int row = l / PA + RA * ps, pc = l % PA;
        *reinterpret_cast<float4*>(imgA + row * (TCA * 4) + (((pc >> 1) ^ tr_swz<TCA>(row)) * 32) + (pc & 1) * 16) = ga[ps];
      }
#pragma unroll
      for (int ps = 0; ps < NPB; ++ps) {
        const int row = l / PB + RB * ps, pc = l % PB;
        *reinterpret_cast<float4*>(imgB + row * (TCB * 4) + (((pc >> 1) ^ tr_swz<TCB>(row)) * 32) + (pc & 1) * 16) = gb[ps];
      }
    } else {
#pragma unroll
      for (int ps = 0; ps < NPA; ++ps) *reinterpret_cast<float4*>(&As[w][l / PA + RA * ps][(l % PA) * 4]) = ga[ps];
#pragma unroll
      for (int ps = 0; ps < NPB; ++ps) *reinterpret_cast<float4*>(&Bs[w][l / PB + RB * ps][(l % PB) * 4]) = gb[ps];
    }
    // rows of chunk c+1 (indices arrived a chunk ago), indices of chunk c+2
    ia1 = ia2;
    ib1 = ib2;
    if (c + 1 < c1) GCL_GATHER(ia1, ib1);
    load_pairs(c + 2, ia2, ib2);
    WAVE_FENCE();
    if (PRE) {
      tr_mma_half<TCA, TCB, 0>(trA, trB, acc);
      tr_mma_half<TCA, TCB, 1>(trA, trB, acc);
      continue;
    }
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      u32x4 pa[NBI][3];
#pragma unroll
      for (int a = 0; a < NBI; ++a) {
        float v[8];
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) v[jj] = As[w][16 * half + 8 * h + jj][a * 32 + i];
        split8<PL>(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]), sa, pa[a]);
        if (PL == 4) dw_a_planes(pa[a]);
      }
#pragma unroll
      for (int b = 0; b < NBJ; ++b) {
        float v[8];
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) v[jj] = Bs[w][16 * half + 8 * h + jj][b * 32 + i];
        u32x4 pb[3];
        split8<PL>(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]), sb, pb);
#pragma unroll
        for (int a = 0; a < NBI; ++a) mfma_terms_dw<PL>(pa[a], pb, acc[a][b]);
      }
    }
  }
  flush(kcur);
#undef GCL_GATHER
}

// ---- weight gradient on plane images, 128 x 128 output block per workgroup (round 4) ---------------------------------
// k_conv_bwd_weight_split gives every wave its own 32 pairs and a 64 x 64 block, so a layer with Ca = Cb = 128 gathers every
// row twice (once per channel tile of the OTHER operand), 256 x 256 four times; per CU the launches have 8 waves x 16 KB of
// gathers in flight and deliver what that buys at ~2.5 us per round trip (profiles/r04_conv_experiments.txt, 23 / 27) --
// bytes per MFMA are the lever.  Here the four waves of a workgroup share the SAME 32 pairs: the rows (128 channels of each
// operand) are gathered once into a double-buffered workgroup tile, wave (wi, wj) multiplies channel half wi of A with half
// wj of B (its 64 x 64 block, the same registers as before), and every wave stages a quarter of each sub-chunk two
// sub-chunks ahead: the same 16 KB in flight per wave for twice the MFMAs.  No cross-wave sums: a wave flushes its own
// block.  One workgroup barrier per sub-chunk of 32 pairs.  Deterministic (fixed order); NOT bitwise equal to the
// 64 x 64 kernel (one accumulator per block walks the pairs in order instead of four interleaved ones).
template <bool DUMMY = false>
__global__ void __launch_bounds__(256, 2) k_conv_bwd_weight_wg128(const float* __restrict__ A, const float* __restrict__ B,
                                                               const int* __restrict__ pair_a,
                                                               const int* __restrict__ pair_b, SegOffW seg, int K,
                                                               int ca, int cb, long long n_chunks, int per,
                                                               float* slabs, const int* __restrict__ a_amax,
                                                               const int* __restrict__ b_amax, int n_wg_x, int n_tiles,
                                                               unsigned a_bytes, unsigned b_bytes) {
  constexpr int TC = 64;                   // a wave's channel half of each operand
  constexpr int SUB = GCL_PAIR_CHUNK / 32; // sub-chunks of 32 pairs per 128-pair chunk
  const float out_scale = 1.f / (amax_scale(a_amax) * amax_scale(b_amax));
  // [buffer][operand A | B][channel half][32 rows][256 B]: each [32][256 B] piece is the row image of the 64-channel kernels
  __shared__ __attribute__((aligned(16))) unsigned char tiles[2][2][2][32 * TC * 4];
  const int t = threadIdx.x, l = t & 63, w = t >> 6, wi = w >> 1, wj = w & 1;
  constexpr int BUF_BYTES = 2 * 2 * 32 * TC * 4;      // one buffer: both operands, both halves (32 KB)
  unsigned trA[2][2], trB[2][2];           // buffer 0: [32-channel slice of the half][plane]; buffer 1 = + BUF_BYTES (immediate)
  {
    const int q = (l & 15) >> 2, pp = l & 3, c16 = (l >> 4) & 1, trow = 8 * (l >> 5) + q;
    const unsigned la = (unsigned)(unsigned long long)((__attribute__((address_space(3))) unsigned char*)&tiles[0][0][wi][0]);
    const unsigned lb = (unsigned)(unsigned long long)((__attribute__((address_space(3))) unsigned char*)&tiles[0][1][wj][0]);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) {
        const unsigned off = trow * (TC * 4) + (((a * 4 + pl * 2 + c16) ^ tr_swz<TC>(q)) * 32) + pp * 8;
        trA[a][pl] = la + off;
        trB[a][pl] = lb + off;
      }
  }
  int bx, by;
  if (n_tiles > 0) {
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    by = slot % n_tiles;
    bx = (slot / n_tiles) * 8 + xcd;
    if (bx >= n_wg_x) return;
  } else {
    bx = blockIdx.x;
    by = blockIdx.y;
  }
  const int tiles_b = cb / 128;
  const int ca0 = (by / tiles_b) * 128, cb0 = (by % tiles_b) * 128;
  const long long c0 = (long long)bx * per;
  const long long c1 = (c0 + per < n_chunks) ? c0 + per : n_chunks;
  if (c0 >= c1) return;
  const long long s0 = c0 * SUB, s1 = c1 * SUB;      // sub-chunk range of this workgroup

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  const __amdgpu_buffer_rsrc_t arsrc = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (int)a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t brsrc = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, (int)b_bytes, 0x00020000);
  const unsigned a_row = (unsigned)ca * 4u, b_row = (unsigned)cb * 4u;
  // a lane's share of a sub-chunk: rows 8 w + 2 ps + (l >> 5), ps = 0 .. 3, 16-byte piece l & 31 of the 512-byte row
  const int pc = l & 31, rsub = l >> 5;
  const unsigned a_col = (unsigned)ca0 * 4u + (unsigned)pc * 16u, b_col = (unsigned)cb0 * 4u + (unsigned)pc * 16u;
  // where piece pc of row r goes: half pc / 16, then the 64-channel row image (32-byte blocks XOR-swizzled with the row)
  unsigned dst[4];
#pragma unroll
  for (int ps = 0; ps < 4; ++ps) {
    const int row = 8 * w + 2 * ps + rsub, p16 = pc & 15;
    dst[ps] = (unsigned)((pc >> 4) * (32 * TC * 4) + row * (TC * 4) + (((p16 >> 1) ^ tr_swz<TC>(row)) * 32) + (p16 & 1) * 16);
  }
  auto load_pairs = [&](long long sc, int& ia, int& ib) {      // lanes 0 .. 31: the pair of row `lane` of sub-chunk sc
    ia = -1;
    ib = -1;
    if (sc < s1 && l < 32) {
      const long long p0 = sc * 32 + l;
      ia = pair_a[p0];
      ib = pair_b[p0];
    }
  };
#define GCLW_GATHER(IA, IB, GA, GB)                                                                            \
  {                                                                                                            \
    _Pragma("unroll") for (int ps = 0; ps < 4; ++ps) {                                                         \
      const int row_ = 8 * w + 2 * ps + rsub;                                                                  \
      const unsigned ra_ = (unsigned)__shfl(IA, row_), rb_ = (unsigned)__shfl(IB, row_);                       \
      GA[ps] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(arsrc, (int)(ra_ * a_row + a_col), 0, 0)); \
      GB[ps] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(brsrc, (int)(rb_ * b_row + b_col), 0, 0)); \
    }                                                                                                          \
  }
#define GCLW_STORE(BUF, GA, GB)                                                                                \
  {                                                                                                            \
    _Pragma("unroll") for (int ps = 0; ps < 4; ++ps) {                                                         \
      *reinterpret_cast<float4*>(&tiles[BUF][0][0][0] + dst[ps]) = GA[ps];                                     \
      *reinterpret_cast<float4*>(&tiles[BUF][1][0][0] + dst[ps]) = GB[ps];                                     \
    }                                                                                                          \
  }
  auto flush = [&](int k) {      // this wave's 64 x 64 block of the slab of (workgroup, offset k)
    int cbv = cb;
    asm volatile("" : "+s"(cbv));      // opaque here: keeps the 64 element offsets out of the loop's live registers (they spilled)
    float* base = slabs + (long long)(bx + k) * ((long long)ca * cb) +
                  (long long)(ca0 + wi * 64 + 4 * (l >> 5)) * cb + cb0 + wj * 64 + (l & 31);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          base[(unsigned)((a * 32 + (r & 3) + 8 * (r >> 2)) * cbv + b * 32)] = acc[a][b][r] * out_scale;
          acc[a][b][r] = 0.f;
        }
  };

  int kcur = 0;
  while (seg.off[kcur + 1] <= c0 * GCL_PAIR_CHUNK) ++kcur;
  float4 ga0[4], gb0[4], ga1[4], gb1[4];
  int ia, ib, ian, ibn;
  // sub-chunks s0 and s0 + 1 -> registers, s0 -> buffer 0, then s0 + 2 -> the freed registers; indices two ahead
  load_pairs(s0, ia, ib);
  load_pairs(s0 + 1, ian, ibn);
  GCLW_GATHER(ia, ib, ga0, gb0);
  GCLW_GATHER(ian, ibn, ga1, gb1);
  load_pairs(s0 + 2, ia, ib);
  load_pairs(s0 + 3, ian, ibn);
  GCLW_STORE(0, ga0, gb0);
  GCLW_GATHER(ia, ib, ga0, gb0);          // rows of s0 + 2 (out of range: row -1 reads zeros, never used)
  load_pairs(s0 + 4, ia, ib);
  __syncthreads();
  // steady state, two sub-chunks per trip (static register sets): at the top of the trip for sub-chunk s, buffer s & 1 holds
  // s, set 1 holds s + 1, set 0 holds s + 2 (in flight), `ian/ibn` = pairs of s + 3, `ia/ib` = pairs of s + 4
#define GCLW_COMPUTE(BUF)                                                                                      \
  {                                                                                                            \
    tr_mma_half<TC, TC, 0, (BUF)*BUF_BYTES>(trA, trB, acc);                                                    \
    tr_mma_half<TC, TC, 1, (BUF)*BUF_BYTES>(trA, trB, acc);                                                    \
  }
#define GCLW_SEGMENT(S)                                                                                        \
  if (((S) + 1) % SUB == 0) {        /* last sub-chunk of a 128-pair chunk: does the next chunk start a new offset? */ \
    const long long nb_ = ((S) + 1) / SUB * GCL_PAIR_CHUNK;                                                    \
    if ((S) + 1 < s1 && nb_ >= seg.off[kcur + 1]) {                                                            \
      flush(kcur);                                                                                             \
      while (seg.off[kcur + 1] <= nb_) ++kcur;                                                                 \
    }                                                                                                          \
  }
  for (long long s = s0; s < s1; s += 2) {
    // ---- sub-chunk s (buffer 0 of this trip's pair = (s - s0) & 1 == 0 -> buffer index alternates with s - s0)
    GCLW_COMPUTE(0);
    GCLW_SEGMENT(s);
    GCLW_STORE(1, ga1, gb1);              // s + 1 -> buffer 1 (last read one sub-chunk ago, before the previous barrier)
    GCLW_GATHER(ian, ibn, ga1, gb1);      // s + 3
    load_pairs(s + 5, ian, ibn);
    __syncthreads();
    if (s + 1 >= s1) break;
    // ---- sub-chunk s + 1
    GCLW_COMPUTE(1);
    GCLW_SEGMENT(s + 1);
    GCLW_STORE(0, ga0, gb0);              // s + 2 -> buffer 0
    GCLW_GATHER(ia, ib, ga0, gb0);        // s + 4
    load_pairs(s + 6, ia, ib);
    __syncthreads();
  }
  flush(kcur);
#undef GCLW_SEGMENT
#undef GCLW_COMPUTE
#undef GCLW_STORE
#undef GCLW_GATHER
}

__global__ void __launch_bounds__(256) k_bwd_weight_reduce(const float* __restrict__ slabs, SegOffW seg, int per,
                                                           long long mat, float* dw) {
  const int k = blockIdx.y;
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= mat) return;
  float s = 0.f;
  if (seg.off[k + 1] > seg.off[k]) {
    long long first = seg.off[k] / GCL_PAIR_CHUNK, last = seg.off[k + 1] / GCL_PAIR_CHUNK - 1;
    long long lo = first / per, hi = last / per;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;     // four slab loads in flight; fixed summation order
    long long bx = lo;
    for (; bx + 3 <= hi; bx += 4) {
      s0 += slabs[(bx + k) * mat + e];
      s1 += slabs[(bx + 1 + k) * mat + e];
      s2 += slabs[(bx + 2 + k) * mat + e];
      s3 += slabs[(bx + 3 + k) * mat + e];
    }
    for (; bx <= hi; ++bx) s0 += slabs[(bx + k) * mat + e];
    s = (s0 + s1) + (s2 + s3);
  }
  dw[(long long)k * mat + e] = s;
}

// RG mode: dw[k] = sum over the row ranges j of slab (j K + k), in order (four loads in flight, fixed tree)
__global__ void __launch_bounds__(256) k_bwd_weight_reduce_rg(const float* __restrict__ slabs, int n_ranges, int K,
                                                              long long mat, float* dw) {
  const int k = blockIdx.y;
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= mat) return;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int j = 0;
  for (; j + 3 < n_ranges; j += 4) {
    s0 += slabs[((long long)j * K + k) * mat + e];
    s1 += slabs[((long long)(j + 1) * K + k) * mat + e];
    s2 += slabs[((long long)(j + 2) * K + k) * mat + e];
    s3 += slabs[((long long)(j + 3) * K + k) * mat + e];
  }
  for (; j < n_ranges; ++j) s0 += slabs[((long long)j * K + k) * mat + e];
  dw[(long long)k * mat + e] = (s0 + s1) + (s2 + s3);
}

// ---- weight gradient of a kernel_size-1 convolution (identity map): dW = A^T B as a row STREAM (round 5) -----------------
// conv1_tr 96 -> 64 and final 64 -> 32 (model/resunet.py:153-171) own one "pair" per row, (i, i): nothing to gather.  The
// pair-list kernels still walked an index list and staged 32-row tiles through LDS for them (144 / 211 us at 0.53 M rows,
// 0.18 - 0.22 of HBM for bytes that are compulsory).  Here a wave takes 16 consecutive rows per step straight from global
// memory into MFMA fragment order: lane (i, h) loads rows r0 + 8 h + j (j = 0..7), channel 32 t + i -- every load instruction
// covers two full 128-byte row segments -- one step ahead of the step it multiplies; no LDS, no index reads, no barrier in the
// loop.  Rows beyond n read zero through the buffer resource.  fp16x3 split on the fly (the tensors are C < 128: no plane
// images).  Deterministic: a wave sums its steps in order, the four waves are added in wave order through LDS, one
// slab per workgroup, slabs summed in order by k_rows_slab_sum.
template <int NBI, int NBJ>
__global__ void __launch_bounds__(256, 2) k_bwd_weight_rows(const float* __restrict__ A, const float* __restrict__ B,
                                                            long long n_rows, int steps_per_wg, float* slabs,
                                                            const int* __restrict__ a_amax, const int* __restrict__ b_amax,
                                                            unsigned a_bytes, unsigned b_bytes) {
  constexpr int CA = NBI * 32, CB = NBJ * 32;
  __shared__ __attribute__((aligned(16))) float lds_sum[CA * CB];
  const float sa = amax_scale(a_amax), sb = amax_scale(b_amax);
  const float out_scale = 1.f / (sa * sb);
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int i = l & 31, h = l >> 5;
  const __amdgpu_buffer_rsrc_t arsrc = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (int)a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t brsrc = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, (int)b_bytes, 0x00020000);
  const long long n_steps = (n_rows + 15) / 16;
  const long long s0 = (long long)blockIdx.x * steps_per_wg;
  const long long s1 = (s0 + steps_per_wg < n_steps) ? s0 + steps_per_wg : n_steps;
  f32x16 acc[NBI][NBJ];
#pragma unroll
  for (int a = 0; a < NBI; ++a)
#pragma unroll
    for (int b = 0; b < NBJ; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  float ra[2][NBI][8], rb[2][NBJ][8];
  // One base offset per operand and step (byte offsets fit 32 bits: operands < 4 GiB, checked by the entry), the (row j,
  // slice) part is an instruction immediate.  A step beyond the wave's share is redirected past the tensor: the buffer
  // resource returns zeros, and a product of zeros leaves the sums unchanged -- no branch anywhere in the loop.
  const int wv = __builtin_amdgcn_readfirstlane(w);
  constexpr unsigned BEYOND = 0xFFFF0000u;      // + the largest immediate (< 4096) still does not wrap
  auto load = [&](long long st, float (&xa)[NBI][8], float (&xb)[NBJ][8]) {
    const unsigned r0 = (unsigned)(st * 16) + 8u * h;
    const int va = (int)(st < s1 ? (r0 * CA + i) * 4u : BEYOND), vb = (int)(st < s1 ? (r0 * CB + i) * 4u : BEYOND);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
#pragma unroll
      for (int a = 0; a < NBI; ++a)
        xa[a][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(arsrc, va + (j * CA + a * 32) * 4, 0, 0));
#pragma unroll
      for (int b = 0; b < NBJ; ++b)
        xb[b][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(brsrc, vb + (j * CB + b * 32) * 4, 0, 0));
    }
  };
  auto mma = [&](float (&xa)[NBI][8], float (&xb)[NBJ][8]) {
    u32x4 pa[NBI][3];
#pragma unroll
    for (int a = 0; a < NBI; ++a)
    {
      split8<4>(make_float4(xa[a][0], xa[a][1], xa[a][2], xa[a][3]), make_float4(xa[a][4], xa[a][5], xa[a][6], xa[a][7]), sa, pa[a]);
      dw_a_planes(pa[a]);
    }
#pragma unroll
    for (int b = 0; b < NBJ; ++b) {
      u32x4 pb[3];
      split8<4>(make_float4(xb[b][0], xb[b][1], xb[b][2], xb[b][3]), make_float4(xb[b][4], xb[b][5], xb[b][6], xb[b][7]), sb, pb);
#pragma unroll
      for (int a = 0; a < NBI; ++a) mfma_terms_dw<4>(pa[a], pb, acc[a][b]);
    }
  };
  // the four waves interleave over the workgroup's steps (the workgroup streams one contiguous row range); two steps per
  // iteration so that the register buffers alternate without copies; loads run one step ahead of their products
  long long st = s0 + wv;
  load(st, ra[0], rb[0]);
  for (; st < s1; st += 8) {
    load(st + 4, ra[1], rb[1]);
    mma(ra[0], rb[0]);
    load(st + 8, ra[0], rb[0]);
    mma(ra[1], rb[1]);
  }
  // the four waves' sums are added in wave order through ONE slab image in LDS (lane-owned cells: ((w0 + w1) + w2) + w3)
#pragma unroll 1
  for (int ww = 0; ww < 4; ++ww) {
    if (w == ww) {
#pragma unroll
      for (int a = 0; a < NBI; ++a)
#pragma unroll
        for (int b = 0; b < NBJ; ++b)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float* cell = &lds_sum[((a * NBJ + b) * 16 + r) * 64 + l];
            *cell = (ww == 0) ? acc[a][b][r] : *cell + acc[a][b][r];
          }
    }
    __syncthreads();
  }
  float* slab = slabs + (long long)blockIdx.x * (CA * CB);
  for (int e = threadIdx.x; e < CA * CB; e += 256) {
    const int ll = e & 63, r = (e >> 6) & 15, blk = e >> 10;
    const int a = blk / NBJ, b = blk % NBJ;
    const int row = a * 32 + (r & 3) + 8 * (r >> 2) + 4 * (ll >> 5);
    slab[row * CB + b * 32 + (ll & 31)] = lds_sum[e] * out_scale;
  }
}

// dw[e] = sum of n_slabs slabs in a FIXED order: thread (e, q) sums the slabs congruent q mod 4 in ascending order on two
// chains, the four q are added in order through LDS.  64 elements per workgroup: mat / 64 workgroups share the read
__global__ void __launch_bounds__(256) k_rows_slab_sum(const float* __restrict__ slabs, int n_slabs, int mat, float* dw) {
  __shared__ float part[4][64];
  const int e = blockIdx.x * 64 + (threadIdx.x & 63), q = threadIdx.x >> 6;
  float c0 = 0.f, c1 = 0.f;
  if (e < mat) {
    int j = q;
    for (; j + 4 < n_slabs; j += 8) {
      c0 += slabs[(long long)j * mat + e];
      c1 += slabs[(long long)(j + 4) * mat + e];
    }
    if (j < n_slabs) c0 += slabs[(long long)j * mat + e];
  }
  part[q][threadIdx.x & 63] = c0 + c1;
  __syncthreads();
  if (q == 0 && e < mat) dw[e] = (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
}

// cell limits of the RG mode: bounds[k (n_ranges + 1) + j] = first position p of offset k's segment whose sorted-side row
// is >= j rr (the segment ascends in that row and ends in -1 padding, which counts as +infinity)
__global__ void __launch_bounds__(256) k_pair_bounds(const int* __restrict__ sorted_rows, SegOffW seg, int K, int rr,
                                                     int n_ranges, int* __restrict__ bounds) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= K * (n_ranges + 1)) return;
  const int k = t / (n_ranges + 1), j = t % (n_ranges + 1);
  long long lo = seg.off[k], hi = seg.off[k + 1];
  const long long want = (long long)j * rr;
  while (lo < hi) {
    const long long mid = (lo + hi) >> 1;
    const int v = sorted_rows[mid];
    if (v >= 0 && v < want) lo = mid + 1; else hi = mid;
  }
  bounds[t] = (int)lo;
}

// rows per range of the RG mode: a power of two in [512, 4096] that gives ~2048 or more (range, offset) cells
static int dw_range_rows(long long n_rows, int K) {
  long long want = n_rows * K / 2048;
  int rr = 512;
  while (rr < 4096 && rr * 2 <= want) rr *= 2;
  return rr;
}
static bool dw_rg_shape(int K, int ca, int cb, int prec, int planes, int sorted_side, long long n_rows) {
  static const int on = [] { const char* e = getenv("GCL_DW_RANGES"); return e ? atoi(e) : 1; }();
  return on && sorted_side != 0 && K > 1 && K <= 27 && !planes && prec != 0 && (ca == 32 || ca == 64) &&
         (cb == 32 || cb == 64) && n_rows >= 32768;
}

// columns per wave (32 NB) of a forward launch: 128 when Cout allows, but 64 when the 128-wide launch would have
// 513..1024 workgroups -- a second, poorly filled round on 256 CUs x 2 resident workgroups -- where twice as many
// half-width workgroups (3 resident per CU) finish earlier (measured -7 % on the 128->128 / 256->256 layers of the
// KITTI batch).  GCL_NB_POLICY=0 disables the rule.
static int conv_fwd_nb(long long n_out, int cout, int prec) {
  int nb = (cout % 128 == 0) ? 4 : ((cout % 64 == 0) ? 2 : 1);
  static const int pol = [] { const char* s = getenv("GCL_NB_POLICY"); return s ? atoi(s) : 1; }();
  static const int force = [] { const char* s = getenv("GCL_FORCE_NB"); return s ? atoi(s) : 0; }();   // diagnostic
  if (force == 1 || (force == 2 && nb > 2)) return force;
  if (prec == 3 && nb == 4) nb = 2;   // three planes: the double-buffered weight block of NB = 4 would not fit twice
  long long wgs = cdiv(n_out, 128) * (cout / (32 * nb));
  if (prec != 0 && pol == 1 && nb == 4 && wgs > 512 && wgs <= 1024) nb = 2;
  // small launches: narrower column blocks until the launch has a workgroup per CU (the weight layout does not depend on NB and
  // a column's sum is the same chain of products whatever block holds it: bitwise the same result)
  static const long long small = [] { const char* s = getenv("GCL_NB_SMALL_WGS"); return s ? atoll(s) : 256ll; }();
  while (prec != 0 && nb > 1 && cdiv(n_out, 128) * (cout / (32 * nb)) < small) nb /= 2;
  return nb;
}

static int bwd_weight_wgs(long long n_chunks) {
  static const int cap = [] { const char* e = getenv("GCL_DW_MAX_WGS"); int v = e ? atoi(e) : 512; return v < 64 ? 64 : (v > 8192 ? 8192 : v); }();
  long long w = n_chunks / 8;     // (more, smaller slabs measured slower: GCL_DW_MAX_WGS 1024 / 2048 -> wgrad sum 3.19 -> 3.28 / 3.38 ms per step)
  if (w < 1) w = 1;
  if (w > cap) w = cap;
  return (int)w;
}

// ---------------------------------------------------------------------------------------------------
// first layer (Cin <= 4 -> 32): VALU kernels over the neighbour table
// ---------------------------------------------------------------------------------------------------
// OCCUPANCY ROWS (presence != NULL, cin == 1, not_ones[v] == 0): every feature row v gathers is exactly 1.0f -- what the
// reference's loaders feed (torch.ones((n, 1))): its test loaders and scripts for every cloud, its training loaders for
// the six neighbour clouds of a sample; only the centre cloud carries lib/transforms.py:18 Jitter (prob. 0.95;
// lib/colocation_data_loader.py:401-415).  A row's neighbours lie in its own cloud (the batch index is part of the key),
// so the flag is per cloud, spread to rows by gcl_not_ones_rows.  x[nbr[k][v]] is then bit k of the row's presence
// words: the forward adds W[k] over the set bits (k_stem_fwd_occ; this kernel skips the row), the weight gradient fills
// its 0/1 tile from the words.  Same values in the same order: bitwise identical to the table path, which every other
// row takes (device-side flags, no host decision).  Measured at 0.53 M rows, K = 125, all rows occupancy: forward
// 150 -> 48 us, weight gradient 195 -> 157 us (tools/micro/stem_time.py).
static __device__ __forceinline__ void stem_fwd_table(const float* __restrict__ x, const float* __restrict__ w,
                                                      const int* __restrict__ nbr, long long n_out, int K, int cin,
                                                      int cout, float* __restrict__ y, const unsigned* __restrict__ presence,
                                                      const int* __restrict__ not_ones, const long long bx) {
  // thread = output row; the weights W[k][ci][0..31] are wave-uniform and come through the scalar cache (s_load),
  // so the inner product costs one v_fmac with an SGPR operand per (offset, channel) and no LDS traffic
  long long v = bx * blockDim.x + threadIdx.x;
  if (v >= n_out) v = n_out - 1;   // keep the wave uniform; the duplicate rows are not stored
  float acc[32];
#pragma unroll
  for (int c = 0; c < 32; ++c) acc[c] = 0.f;
  constexpr int STEM_B = 25;
  if (presence && cin == 1 && not_ones[v] == 0) return;    // k_stem_fwd_occ, enqueued right behind, writes this row
  {
  // offsets in batches of STEM_B: all table reads of a batch are issued first, then all feature gathers, then the FMAs --
  // the walk is bound by the latency of these two dependent loads (one table entry per offset and row, 265 MB at K = 125
  // and 0.5 M rows), so the number in flight per thread is what counts (was 5: 221 us; the accumulation order per output
  // element is unchanged: k ascending, ci ascending)
  for (int k0 = 0; k0 < K; k0 += STEM_B) {
    int idx[STEM_B];
#pragma unroll
    for (int j = 0; j < STEM_B; ++j) idx[j] = (k0 + j < K) ? nbr[(long long)(k0 + j) * n_out + v] : -1;
    for (int ci = 0; ci < cin; ++ci) {
      float xv[STEM_B];
#pragma unroll
      for (int j = 0; j < STEM_B; ++j) xv[j] = idx[j] >= 0 ? x[(long long)idx[j] * cin + ci] : 0.f;
#pragma unroll
      for (int j = 0; j < STEM_B; ++j) {
        if (k0 + j < K) {      // wave-uniform
          const float* wr = w + (long long)((k0 + j) * cin + ci) * cout + blockIdx.y * 32;     // blockIdx.y = 32-column block
#pragma unroll
          for (int c = 0; c < 32; ++c) acc[c] = fmaf(xv[j], wr[c], acc[c]);
        }
      }
    }
  }
  }
  if (bx * blockDim.x + threadIdx.x >= n_out) return;
  float4* yo = reinterpret_cast<float4*>(y + v * cout + blockIdx.y * 32);
#pragma unroll
  for (int c4 = 0; c4 < 8; ++c4) yo[c4] = make_float4(acc[4 * c4], acc[4 * c4 + 1], acc[4 * c4 + 2], acc[4 * c4 + 3]);
}

// Occupancy input, forward: y[v][c] = sum over the SET bits k of the row's presence words of W[k][c], k ascending -- the
// order in which the general kernel adds the same terms (its absent offsets add fma(0, w, acc) = acc), so the result is
// bitwise identical.  8 lanes per row (4 columns each), the 32-column slice of W in LDS (one ds_read_b128 per set bit
// and lane): ~20 adds per row instead of 125 table reads + gathers + 4000 FMAs.
constexpr int STEM_OCC_ROWS = 256;      // rows per workgroup (8 passes of 32)
static __device__ __forceinline__ void stem_fwd_occ(const float* __restrict__ w, const unsigned* __restrict__ presence,
                                                    const int* __restrict__ not_ones, long long n_out, int K, int cout,
                                                    float* __restrict__ y, const long long bx) {
  __shared__ __attribute__((aligned(16))) float Ws[128 * 32];
  const int cb0 = blockIdx.y * 32;
  for (int e = threadIdx.x; e < K * 8; e += 256) {
    const int k = e >> 3, q = e & 7;
    *reinterpret_cast<float4*>(&Ws[k * 32 + q * 4]) = *reinterpret_cast<const float4*>(w + (long long)k * cout + cb0 + q * 4);
  }
  __syncthreads();
  const int words = (K + 31) >> 5;
  const int sub = threadIdx.x & 7;
  for (int pass = 0; pass < STEM_OCC_ROWS / 32; ++pass) {
    const long long v = bx * STEM_OCC_ROWS + pass * 32 + (threadIdx.x >> 3);
    if (v >= n_out || not_ones[v] != 0) continue;      // flagged rows: stem_fwd_table's walk
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int q = 0; q < words; ++q) {
      unsigned bits = presence[v * words + q];
      while (bits) {
        const int k = q * 32 + __builtin_ctz(bits);
        bits &= bits - 1;
        const float4 wv = *reinterpret_cast<const float4*>(&Ws[k * 32 + sub * 4]);
        acc.x = fmaf(1.f, wv.x, acc.x);
        acc.y = fmaf(1.f, wv.y, acc.y);
        acc.z = fmaf(1.f, wv.z, acc.z);
        acc.w = fmaf(1.f, wv.w, acc.w);
      }
    }
    *reinterpret_cast<float4*>(y + v * cout + cb0 + sub * 4) = acc;
  }
}

__global__ void __launch_bounds__(256) k_stem_fwd(const float* __restrict__ x, const float* __restrict__ w,
                                                  const int* __restrict__ nbr, long long n_out, int K, int cin, int cout,
                                                  float* __restrict__ y, const unsigned* __restrict__ presence,
                                                  const int* __restrict__ not_ones) {
  stem_fwd_table(x, w, nbr, n_out, K, cin, cout, y, presence, not_ones, (long long)blockIdx.x);
}
// Two launches, not one: merged (table workgroups first, occupancy workgroups behind them in the same grid) the
// occupancy rows inherit the table walk's registers and lose the occupancy that hides their LDS reads -- all-ones input
// (every inference pass) 49 -> 72 us, the bench's training batch 97 -> 85 us.
__global__ void __launch_bounds__(256) k_stem_fwd_occ(const float* __restrict__ w, const unsigned* __restrict__ presence,
                                                      const int* __restrict__ not_ones, long long n_out, int K, int cout,
                                                      float* __restrict__ y) {
  stem_fwd_occ(w, presence, not_ones, n_out, K, cout, y, (long long)blockIdx.x);
}

constexpr int STEM_KMAX = 125;

// dW[k][ci][c] = sum_v x[nbr[k][v]][ci] * dY[v][c]  as an exact-f32 MFMA GEMM: M = K offsets (4 blocks of 32),
// N = 32 channels, reduction over the output rows v.  Per tile of 128 rows the workgroup first resolves the gather
// with coalesced reads of the k-major neighbour table (lanes along rows) into an LDS tile A[k][row]; the MFMA phase
// then reads A[i = offset][kk = row] from LDS (row pitch 129: conflict-free) and B[kk = row][j = channel] as a
// coalesced row of dY.  Per-workgroup slabs + ordered reduction (k_stem_reduce).
// TILE = rows per tile.  64 (default): 33 KB of LDS and <= 128 registers, FOUR workgroups per CU -- the kernel is a chain
// of dependent phases per tile (flags / words or table -> LDS tile -> barrier -> dY values -> 32 MFMAs per wave) and what
// hides their latencies is other workgroups; at 128 rows (66 KB, two per CU) the launch at the end of the backward pass,
// alone on the chip with the optimizer waiting, took 155 us at 0.53 M rows (table path 197), 92 us now (118).
constexpr int STEM_TILE = 64;
template <int TILE>
__global__ void __launch_bounds__(256, TILE == 64 ? 4 : 2)
    k_stem_bwd_weight(const float* __restrict__ x, const float* __restrict__ dy, const int* __restrict__ nbr, long long n_out,
                      int K, int cin, int cout, float* slabs, const unsigned* __restrict__ presence,
                      const int* __restrict__ not_ones, long long rows_per_wg) {
  constexpr int LD = TILE + 1;            // A[k][row] pitch: conflict-free column reads
  constexpr int G = 256 / TILE;           // fill: thread -> row t & (TILE - 1), offsets k = t / TILE + G j
  constexpr int RW = TILE / 4;            // rows of a tile per wave in the MFMA phase
  const bool occ_rows = presence && cin == 1;      // see k_stem_fwd: per row, not_ones[row] == 0
  __shared__ float As[128 * LD > 4096 ? 128 * LD : 4096];      // also the 16 KB buffer of the cross-wave sum
  const int cb0 = blockIdx.y * 32;             // 32-column block of dY / dW
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int i = l & 31, h = l >> 5;
  const long long r_begin = (long long)blockIdx.x * rows_per_wg;
  long long r_end = r_begin + rows_per_wg;
  if (r_end > n_out) r_end = n_out;
  for (int ci = 0; ci < cin; ++ci) {
    f32x16 acc[4];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[kb][r] = 0.f;
    for (long long r0 = r_begin; r0 < r_end; r0 += TILE) {
      __syncthreads();
      {
        const int r = threadIdx.x & (TILE - 1);
        int kq = threadIdx.x / TILE;
        asm volatile("" : "+v"(kq));      // opaque per tile: the 128 / G products k * n_out of the table path were hoisted
                                          // out of the tile loop and spilled (652 bytes of scratch)
        const long long row = r0 + r;
        const bool rv = row < r_end;
        const bool occupancy = occ_rows && (!rv || not_ones[row] == 0);
        if (occupancy) {      // A[k][row] = bit k of the row's presence words (1.0 / 0.0), no table read, no gather
          const int words = (K + 31) >> 5;
          unsigned bits[4] = {0u, 0u, 0u, 0u};
          if (rv)
            for (int q = 0; q < words && q < 4; ++q) bits[q] = presence[row * words + q];
          for (int j = 0; j < 128 / G; ++j) {
            const int k = kq + G * j;
            As[k * LD + r] = (k < K && ((bits[k >> 5] >> (k & 31)) & 1u)) ? 1.f : 0.f;
          }
        } else
        for (int j0 = 0; j0 < 128 / G; j0 += 8) {      // 8 table reads, then 8 gathers, in flight at a time
          int idx[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int k = kq + G * (j0 + j);
            idx[j] = (rv && k < K) ? nbr[(long long)k * n_out + row] : -1;
          }
          float a[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) a[j] = idx[j] >= 0 ? x[(long long)idx[j] * cin + ci] : 0.f;
#pragma unroll
          for (int j = 0; j < 8; ++j) As[(kq + G * (j0 + j)) * LD + r] = a[j];
        }
      }
      __syncthreads();
      float bv[RW / 2];      // the lane's dY values of the tile: all loads in flight before the first MFMA
#pragma unroll
      for (int s = 0; s < RW / 2; ++s) {
        const long long row = r0 + w * RW + 2 * s + h;
        bv[s] = (row < r_end) ? dy[row * cout + cb0 + i] : 0.f;
      }
#pragma unroll
      for (int s = 0; s < RW / 2; ++s) {
        const int rl = w * RW + 2 * s + h;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
          acc[kb] = __builtin_amdgcn_mfma_f32_32x32x2f32(As[(kb * 32 + i) * LD + rl], bv[s], acc[kb], 0, 0, 0);
      }
    }
    // cross-wave sum ((w0 + w1) + w2) + w3 through one 16 KB buffer, then wave 3 writes the workgroup's slab
    float* red = As;
    for (int src = 0; src < 3; ++src) {
      __syncthreads();
      if (w == src) {
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
          for (int r = 0; r < 16; ++r) red[(kb * 16 + r) * 64 + l] = acc[kb][r];
      }
      __syncthreads();
      if (w == src + 1) {
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[kb][r] = red[(kb * 16 + r) * 64 + l] + acc[kb][r];
      }
    }
    if (w == 3) {
      int h_o = h;      // opaque: keeps the 64 store offsets out of the kernel's prologue (they were hoisted and spilled)
      asm volatile("" : "+v"(h_o));
      float* slab = slabs + ((long long)blockIdx.x * K * cin + ci) * cout + cb0 + i;
      const int kstep = cin * cout;
#pragma unroll
      for (int kb = 0; kb < 4; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int k = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h_o;
          if (k < K) slab[k * kstep] = acc[kb][r];
        }
    }
  }
}

// ordered sum of the per-workgroup slabs: thread = (element e = t & 15, part = t >> 4) adds slabs part, part + 16, ...
// (4 loads in flight), the sixteen parts are then added as a fixed tree.  16 elements per workgroup: 250 workgroups for the
// 4000 elements of the 5^3 x 1 x 32 kernel (64 elements x 4 parts was 63 workgroups walking ~230 slabs each in series).
__global__ void __launch_bounds__(256) k_stem_reduce(const float* __restrict__ slabs, int n_slabs, long long mat,
                                                     float* dw) {
  __shared__ float red[16][17];
  const int el = threadIdx.x & 15, part = threadIdx.x >> 4;
  const long long e = (long long)blockIdx.x * 16 + el;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (e < mat) {
    int b = part;
    for (; b + 48 < n_slabs; b += 64) {
      s0 += slabs[(long long)b * mat + e];
      s1 += slabs[(long long)(b + 16) * mat + e];
      s2 += slabs[(long long)(b + 32) * mat + e];
      s3 += slabs[(long long)(b + 48) * mat + e];
    }
    for (; b < n_slabs; b += 16) s0 += slabs[(long long)b * mat + e];
  }
  red[part][el] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (part == 0 && e < mat) {
    float q[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) q[g] = (red[4 * g][el] + red[4 * g + 1][el]) + (red[4 * g + 2][el] + red[4 * g + 3][el]);
    dw[e] = (q[0] + q[1]) + (q[2] + q[3]);
  }
}


// ---------------------------------------------------------------------------------------------------
// generic shapes (any Cin / Cout, K up to 125): exact-fp32 VALU kernels.  The MFMA kernels above need channel
// counts that are multiples of 32 and K <= 27; everything else -- demo.py:29's 16-dim head (`final` 64 -> 16), a
// 5^3 convolution on wide features -- runs here, so that the C ABI takes every shape the reference's models can have.
// These layers are small: a thread owns one output row x TC output columns, the weights W_eff[k][c][n0 .. n0+TC) are
// wave-uniform and come through the scalar cache (as in k_stem_fwd), rows are read 16 bytes at a time when aligned.
// ---------------------------------------------------------------------------------------------------
__global__ void k_pack_weights_generic(const float* __restrict__ w, int K, int cin, int cout, int mode, float* wp) {
  long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= (long long)K * cin * cout) return;
  const int cin_e = mode == 0 ? cin : cout, cout_e = mode == 0 ? cout : cin;
  const int n = (int)(o % cout_e), c = (int)((o / cout_e) % cin_e), k = (int)(o / ((long long)cin_e * cout_e));
  float v;
  if (mode == 0) v = w[((long long)k * cin + c) * cout + n];
  else v = w[((long long)((mode == 2) ? (K - 1 - k) : k) * cin + n) * cout + c];
  wp[o] = v;
}

template <int TC>
__global__ void __launch_bounds__(256) k_conv_generic(const float* __restrict__ X, const float* __restrict__ W,
                                                      const int* __restrict__ tbl, const int* __restrict__ order,
                                                      long long n_out, int K, int cin, int cout,
                                                      const float* __restrict__ bias, float* __restrict__ Y,
                                                      ConvEpi epi) {
  long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = r < n_out;
  if (!live) r = n_out - 1;             // keep the wave uniform; the duplicates are not stored
  const int n0 = blockIdx.y * TC;
  const int ncol = (cout - n0 < TC) ? cout - n0 : TC;
  float acc[TC];
#pragma unroll
  for (int j = 0; j < TC; ++j) acc[j] = 0.f;
  const bool vec = (cin & 3) == 0;
  for (int k = 0; k < K; ++k) {
    const int idx = tbl ? tbl[(long long)k * n_out + r] : (int)r;
    if (idx < 0) continue;
    const float* xr = X + (long long)idx * cin;
    const float* wk = W + (long long)k * cin * cout + n0;
    if (vec) {
      for (int c = 0; c < cin; c += 4) {
        const float4 xv = *reinterpret_cast<const float4*>(xr + c);
        const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float* wr = wk + (long long)(c + q) * cout;
#pragma unroll
          for (int j = 0; j < TC; ++j) acc[j] = fmaf(xs[q], wr[j < ncol ? j : 0], acc[j]);
        }
      }
    } else {
      for (int c = 0; c < cin; ++c) {
        const float xv = xr[c];
        const float* wr = wk + (long long)c * cout;
#pragma unroll
        for (int j = 0; j < TC; ++j) acc[j] = fmaf(xv, wr[j < ncol ? j : 0], acc[j]);
      }
    }
  }
  float ymax = 0.f;
  if (live) {
    const long long orow = order ? order[r] : r;
#pragma unroll
    for (int j = 0; j < TC; ++j) {
      if (j < ncol) {
        const int col = n0 + j;
        float v = acc[j] * (epi.col_scale ? epi.col_scale[col] : 1.f) + (bias ? bias[col] : 0.f);
        if (epi.residual) {
          const float rsd = epi.residual[orow * cout + col];
          v = epi.relu == 2 ? (rsd > 0.f ? v : 0.f) : v + rsd;
        }
        if (epi.relu == 1) v = fmaxf(v, 0.f);
        Y[orow * cout + col] = v;
        ymax = fmaxf(ymax, fabsf(v));
      }
    }
  }
  if (epi.y_amax) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ymax = fmaxf(ymax, __shfl_xor(ymax, o));
    if ((threadIdx.x & 63) == 0) amax_slot_publish(epi.y_amax, __float_as_int(ymax), blockIdx.x * 4 + (threadIdx.x >> 6));
  }
}

// weight gradient for generic shapes: workgroup = (range of 128-pair chunks, 16 x 16 tile of dW); thread (i, j) keeps
// one element.  Per chunk the 128 x 16 pieces of both operands are staged in LDS.  Same slab / ordered-reduction scheme
// as the MFMA kernels (k_bwd_weight_reduce), so the result is deterministic.
__global__ void __launch_bounds__(256) k_conv_bwd_weight_generic(const float* __restrict__ A, const float* __restrict__ B,
                                                                 const int* __restrict__ pair_a,
                                                                 const int* __restrict__ pair_b, SegOffW seg, int K,
                                                                 int ca, int cb, long long n_chunks, int per,
                                                                 float* slabs) {
  __shared__ float As[GCL_PAIR_CHUNK][17], Bs[GCL_PAIR_CHUNK][17];
  const int t = threadIdx.x, i = t >> 4, j = t & 15;
  const int tiles_b = (cb + 15) / 16;
  const int ca0 = (blockIdx.y / tiles_b) * 16, cb0 = (blockIdx.y % tiles_b) * 16;
  const long long c0 = (long long)blockIdx.x * per;
  const long long c1 = (c0 + per < n_chunks) ? c0 + per : n_chunks;
  if (c0 >= c1) return;
  const long long mat = (long long)ca * cb;
  float acc = 0.f;
  int kcur = 0;
  while (seg.off[kcur + 1] <= c0 * GCL_PAIR_CHUNK) ++kcur;
  auto flush = [&](int k) {
    if (ca0 + i < ca && cb0 + j < cb) slabs[(long long)(blockIdx.x + k) * mat + (long long)(ca0 + i) * cb + cb0 + j] = acc;
    acc = 0.f;
  };
  for (long long c = c0; c < c1; ++c) {
    const long long pbase = c * GCL_PAIR_CHUNK;
    if (pbase >= seg.off[kcur + 1]) {
      flush(kcur);
      while (seg.off[kcur + 1] <= pbase) ++kcur;
    }
    __syncthreads();
    for (int e = t; e < GCL_PAIR_CHUNK * 16; e += 256) {
      const int p = e >> 4, ch = e & 15;
      const int ia = pair_a[pbase + p], ib = pair_b[pbase + p];
      As[p][ch] = (ia >= 0 && ca0 + ch < ca) ? A[(long long)ia * ca + ca0 + ch] : 0.f;
      Bs[p][ch] = (ib >= 0 && cb0 + ch < cb) ? B[(long long)ib * cb + cb0 + ch] : 0.f;
    }
    __syncthreads();
#pragma unroll 8
    for (int p = 0; p < GCL_PAIR_CHUNK; ++p) acc = fmaf(As[p][i], Bs[p][j], acc);
  }
  flush(kcur);
}

// shapes the MFMA kernels do not take
static bool generic_shape(int K, int cin, int cout) { return (cin % 32) != 0 || (cout % 32) != 0 || K > 27; }

}  // namespace gcl

using namespace gcl;

extern "C" {

#ifdef GCL_STAMPS
int gcl_debug_wgtrace(unsigned long long* out_host, int n_wg) {
  if (n_wg > WG_TRACE_MAX) n_wg = WG_TRACE_MAX;
  GCL_CHECK_HIP(hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_wgtrace), sizeof(unsigned long long) * 4 * n_wg));
  return GCL_OK;
}
int gcl_debug_stamps(unsigned long long* out_host, int reset) {
  if (out_host) GCL_CHECK_HIP(hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 8));
  if (reset) {
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    GCL_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), z, sizeof(z)));
  }
  return GCL_OK;
}
#endif

int64_t gcl_pack_weights_bytes(int32_t K, int32_t cin, int32_t cout, int32_t prec) {
  long long n = (long long)K * cin * cout;
  if (generic_shape(K, cin, cout)) return n * 4;      // fp32 W_eff[k][c][n] for the generic kernels
  return prec == 0 ? n * 4 : n * 2 * (prec == 3 ? 3 : 2);
}

static bool prec_ok(int prec) { return prec == 0 || prec == 2 || prec == 3 || prec == 4; }

int gcl_amax(const float* x, int64_t n, int32_t* amax_bits, int32_t zeroed, void* stream) {
  GCL_CHECK_ARG(x && amax_bits && n > 0, "gcl_amax: bad argument");
  hipStream_t st = (hipStream_t)stream;
  if (!zeroed) GCL_CHECK_HIP(hipMemsetAsync(amax_bits, 0, AMAX_WORDS * sizeof(int32_t), st));
  long long n4 = n / 4;
  long long g = cdiv(n4 > 0 ? n4 : 1, 256);
  if (g > 512) g = 512;
  hipLaunchKernelGGL(k_amax, dim3((unsigned)g), dim3(256), 0, st, (const float4*)x, n4, x + n4 * 4, (int)(n - n4 * 4),
                     amax_bits);
  GCL_CHECK_LAUNCH();
  return GCL_OK;
}

int gcl_amax_multi(const float* const* ptrs, const int64_t* sizes, int32_t n_tensors, int32_t* amax_bits,
                   void* stream) {
  GCL_CHECK_ARG(ptrs && sizes && amax_bits, "gcl_amax_multi: null pointer");
  GCL_CHECK_ARG(n_tensors > 0 && n_tensors <= 65535, "gcl_amax_multi: 1 <= n_tensors <= 65535");
  hipStream_t st = (hipStream_t)stream;
  GCL_CHECK_HIP(hipMemsetAsync(amax_bits, 0, (size_t)n_tensors * AMAX_WORDS * sizeof(int32_t), st));
  hipLaunchKernelGGL(k_amax_multi, dim3(32, (unsigned)n_tensors), dim3(256), 0, st, ptrs, (const long long*)sizes,
                     amax_bits);
  GCL_CHECK_LAUNCH();
  return GCL_OK;
}

int gcl_split_planes(const float* x, int64_t n, int32_t c, const int32_t* amax, void* planes, void* stream) {
  GCL_CHECK_ARG(x && amax && planes, "gcl_split_planes: null pointer");
  GCL_CHECK_ARG(n > 0 && c > 0 && c % 32 == 0, "gcl_split_planes: c must be a positive multiple of 32");
  long long total4 = n * (c / 4);
  long long g = cdiv(total4, 256);
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(k_split_planes, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, (const float4*)x, total4, c,
                     amax, (unsigned short*)planes);
  GCL_CHECK_LAUNCH();
  return GCL_OK;
}

int gcl_pack_weights(const float* w, int32_t K, int32_t cin, int32_t cout, int32_t mode, int32_t prec,
                     const int32_t* w_amax, void* wp, void* stream) {
  GCL_CHECK_ARG(w && wp, "gcl_pack_weights: null pointer");
  GCL_CHECK_ARG(K >= 1 && K <= 125 && cin > 0 && cout > 0, "gcl_pack_weights: bad shape (K %d, Cin %d, Cout %d)", K, cin, cout);
  GCL_CHECK_ARG(mode >= 0 && mode <= 2, "gcl_pack_weights: mode must be 0, 1 or 2");
  GCL_CHECK_ARG(prec_ok(prec), "gcl_pack_weights: prec must be 0 (f32), 2 (bf16x3), 3 (bf16x6) or 4 (fp16x3)");
  long long total = (long long)K * cin * cout;
  dim3 grid((unsigned)cdiv(total, 256));
  hipStream_t st = (hipStream_t)stream;
  if (generic_shape(K, cin, cout)) {
    hipLaunchKernelGGL(k_pack_weights_generic, grid, dim3(256), 0, st, w, K, cin, cout, mode, (float*)wp);
    GCL_CHECK_LAUNCH();
    return GCL_OK;
  }
  GCL_CHECK_ARG(prec != 4 || w_amax, "gcl_pack_weights: fp16x3 needs the weight tensor's gcl_amax");
  if (prec == 0) hipLaunchKernelGGL(k_pack_weights, grid, dim3(256), 0, st, w, K, cin, cout, mode, (float*)wp);
  else if (prec == 2) hipLaunchKernelGGL(k_pack_weights_split<2>, grid, dim3(256), 0, st, w, K, cin, cout, mode, w_amax, (unsigned short*)wp);
  else if (prec == 3) hipLaunchKernelGGL(k_pack_weights_split<3>, grid, dim3(256), 0, st, w, K, cin, cout, mode, w_amax, (unsigned short*)wp);
  else hipLaunchKernelGGL(k_pack_weights_split<4>, grid, dim3(256), 0, st, w, K, cin, cout, mode, w_amax, (unsigned short*)wp);
  GCL_CHECK_LAUNCH();
  return GCL_OK;
}

int gcl_pack_weights_multi(const int64_t* desc, int32_t n_tensors, int64_t total_wgs, int32_t prec,
                           const int32_t* amax_slots, void* out, void* stream) {
  GCL_CHECK_ARG(desc && out && n_tensors > 0 && total_wgs > 0, "gcl_pack_weights_multi: bad argument");
  GCL_CHECK_ARG(prec == 2 || prec == 3 || prec == 4, "gcl_pack_weights_multi: split precisions only (2, 3, 4)");
  GCL_CHECK_ARG(prec != 4 || amax_slots, "gcl_pack_weights_multi: fp16x3 needs the amax slots");
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((unsigned)total_wgs);
  if (prec == 2)
    hipLaunchKernelGGL(k_pack_weights_multi<2>, grid, dim3(256), 0, st, (const PackDesc*)desc, n_tensors, amax_slots,
                       (unsigned char*)out);
  else if (prec == 3)
    hipLaunchKernelGGL(k_pack_weights_multi<3>, grid, dim3(256), 0, st, (const PackDesc*)desc, n_tensors, amax_slots,
                       (unsigned char*)out);
  else
    hipLaunchKernelGGL(k_pack_weights_multi<4>, grid, dim3(256), 0, st, (const PackDesc*)desc, n_tensors, amax_slots,
                       (unsigned char*)out);
  GCL_CHECK_LAUNCH();
  return GCL_OK;
}

int32_t gcl_conv_fwd_nb(int64_t n_out, int32_t cout, int32_t prec) {
  if (n_out <= 0 || cout <= 0 || cout % 32) return 0;
  return conv_fwd_nb(n_out, cout, prec);
}

int gcl_conv_fwd_fused(const float* x, int64_t n_in, int32_t x_is_planes, const void* wp, int32_t prec,
                       const int32_t* x_amax, const int32_t* w_amax, const int32_t* tbl, const int32_t* order,
                       const int32_t* tile_mask, int64_t n_out, int32_t K, int32_t cin, int32_t cout, const float* bias,
                       const float* col_scale, const float* residual, int32_t relu, int32_t* y_amax, float* y,
                       float* stats, int32_t flags, void* stream);

int64_t gcl_conv_fwd_groups_scratch_len(int64_t n_out, int32_t K, int32_t cin, int32_t cout) {
  static const int on = [] { const char* e = getenv("GCL_FWD_GROUPS"); return e ? atoi(e) : 1; }();
  static const long long max_rows = [] { const char* e = getenv("GCL_FWD_GROUPS_MAX_ROWS"); return e ? atoll(e) : 65536ll; }();
  static const int tall_min = [] { const char* e = getenv("GCL_FWD_TALL_MIN_STEPS"); return e ? atoi(e) : 108; }();
  if (!on || n_out <= 0 || n_out > max_rows || K < 8 || K > 27 || (cin % 32) || (cout % 64) || K * (cin / 32) < tall_min) return 0;
  return 4ll * n_out * cout;
}

int gcl_conv_fwd(const float* x, int64_t n_in, int32_t x_is_planes, const void* wp, int32_t prec, const int32_t* x_amax,
                 const int32_t* w_amax, const int32_t* tbl, const int32_t* order, const int32_t* tile_mask,
                 int64_t n_out, int32_t K, int32_t cin, int32_t cout, const float* bias, float* y, float* stats,
                 int32_t flags, void* stream) {
  return gcl_conv_fwd_fused(x, n_in, x_is_planes, wp, prec, x_amax, w_amax, tbl, order, tile_mask, n_out, K, cin, cout,
                            bias, nullptr, nullptr, 0, nullptr, y, stats, flags, stream);
}

int gcl_conv_fwd_fused(const float* x, int64_t n_in, int32_t x_is_planes, const void* wp, int32_t prec,
                       const int32_t* x_amax, const int32_t* w_amax, const int32_t* tbl, const int32_t* order,
                       const int32_t* tile_mask, int64_t n_out, int32_t K, int32_t cin, int32_t cout, const float* bias,
                       const float* col_scale, const float* residual, int32_t relu, int32_t* y_amax, float* y,
                       float* stats, int32_t flags, void* stream) {
  return gcl_conv_fwd_fused_ld(x, n_in, x_is_planes, wp, prec, x_amax, w_amax, tbl, order, tile_mask, n_out, K, cin, cout, bias,
                               col_scale, residual, 0, relu, y_amax, y, stats, flags, stream);
}

int gcl_conv_fwd_fused_ld(const float* x, int64_t n_in, int32_t x_is_planes, const void* wp, int32_t prec,
                          const int32_t* x_amax, const int32_t* w_amax, const int32_t* tbl, const int32_t* order,
                          const int32_t* tile_mask, int64_t n_out, int32_t K, int32_t cin, int32_t cout, const float* bias,
                          const float* col_scale, const float* residual, int32_t residual_ld, int32_t relu,
                          int32_t* y_amax, float* y, float* stats, int32_t flags, void* stream) {
  GCL_CHECK_ARG(residual_ld == 0 || (residual && residual_ld >= cout && !generic_shape(K, cin, cout)),
                "gcl_conv_fwd_fused_ld: residual_ld needs a residual, >= Cout, MFMA-shaped launches");
  GCL_CHECK_ARG(relu == 0 || relu == 1 || (relu == 2 && residual), "gcl_conv_fwd_fused: relu must be 0, 1, or 2 with a residual");
  const ConvEpi epi{col_scale, residual, relu, y_amax, residual_ld};
  const bool use_epi = col_scale || residual || relu || y_amax;
  GCL_CHECK_ARG(prec != 0 || generic_shape(K, cin, cout) || (!col_scale && !residual && !relu && !y_amax),
                "gcl_conv_fwd_fused: the fused epilogue needs a split-precision mode");
  GCL_CHECK_ARG(x && wp && y, "gcl_conv_fwd: null pointer");
  if (generic_shape(K, cin, cout)) {     // any Cin / Cout, K <= 125: exact-fp32 VALU kernel (wp = fp32 W_eff, see gcl_pack_weights)
    GCL_CHECK_ARG(n_in > 0 && n_out > 0 && K >= 1 && K <= 125 && cin > 0 && cout > 0, "gcl_conv_fwd: bad shape");
    GCL_CHECK_ARG(tbl || K == 1, "gcl_conv_fwd: a neighbour table is required when K > 1");
    GCL_CHECK_ARG(!stats, "gcl_conv_fwd: fused BN statistics need Cin, Cout multiples of 32 and K <= 27");
    GCL_CHECK_ARG(!x_is_planes, "gcl_conv_fwd: plane images need Cin, Cout multiples of 32 and K <= 27");
    hipStream_t gst = (hipStream_t)stream;
    if (cout > 8) {
      hipLaunchKernelGGL(k_conv_generic<16>, dim3((unsigned)cdiv(n_out, 256), (unsigned)cdiv(cout, 16)), dim3(256), 0, gst,
                         x, (const float*)wp, tbl, order, (long long)n_out, K, cin, cout, bias, y, epi);
    } else {
      hipLaunchKernelGGL(k_conv_generic<8>, dim3((unsigned)cdiv(n_out, 256), (unsigned)cdiv(cout, 8)), dim3(256), 0, gst,
                         x, (const float*)wp, tbl, order, (long long)n_out, K, cin, cout, bias, y, epi);
    }
    GCL_CHECK_LAUNCH();
    return GCL_OK;
  }
  GCL_CHECK_ARG(n_in > 0 && (long long)n_in * cin * 4 < (1ll << 32) - (1ll << 20),
                "gcl_conv_fwd: the input tensor must be non-empty and smaller than 4 GiB (buffer addressing)");
  const unsigned x_bytes = (unsigned)((long long)n_in * cin * 4);
  GCL_CHECK_ARG(n_out > 0 && K >= 1 && K <= 27, "gcl_conv_fwd: n_out must be positive and 1 <= K <= 27");
  GCL_CHECK_ARG(tbl || K == 1, "gcl_conv_fwd: a neighbour table is required when K > 1");
  GCL_CHECK_ARG((order == nullptr) == (tile_mask == nullptr), "gcl_conv_fwd: order and tile_mask go together");
  GCL_CHECK_ARG(cin % 32 == 0 && cout % 32 == 0 && cin > 0 && cout > 0,
                "gcl_conv_fwd: Cin (%d) and Cout (%d) must be positive multiples of 32", cin, cout);
  GCL_CHECK_ARG(prec_ok(prec), "gcl_conv_fwd: prec must be 0 (f32), 2 (bf16x3), 3 (bf16x6) or 4 (fp16x3)");
  GCL_CHECK_ARG(!stats || prec != 0, "gcl_conv_fwd: fused BN statistics need a split-precision mode");
  GCL_CHECK_ARG(prec != 4 || (x_amax && w_amax), "gcl_conv_fwd: fp16x3 needs gcl_amax of x and of the weights");
  GCL_CHECK_ARG(!x_is_planes || prec == 4, "gcl_conv_fwd: plane images are the fp16x3 operand format");
  hipStream_t st = (hipStream_t)stream;
  static const int swz = [] {   // tuning knob, default off (measured: -5 % with the global sort, +7 % with the windowed sort)
    const char* e = getenv("GCL_XCD_SWIZZLE");
    return e ? atoi(e) : 0;
  }();
  unsigned gx = (unsigned)cdiv(n_out, CONV_ROWS);
  const int nb = conv_fwd_nb(n_out, cout, prec);
  dim3 grid(gx, cout / (32 * nb));
  static const int colgroup = [] { const char* e = getenv("GCL_FWD_COLGROUP"); return e ? atoi(e) : 1; }();
  const bool cg = colgroup && grid.y > 1 && !swz;
  dim3 sgrid = cg ? dim3((unsigned)(cdiv(gx, 8) * 8 * grid.y)) : grid;
  const bool ranges = (flags & GCL_CONV_XCD_RANGES) != 0;     // spatially ordered table: contiguous tile range per XCD
  static const int heavy_first = [] { const char* e = getenv("GCL_CONV_HEAVY_FIRST"); return e ? atoi(e) : 1; }();
  const int sswz = (cg ? (ranges ? 3 : 2) : (ranges ? 1 : swz)) | ((heavy_first && tile_mask && !ranges) ? 16 : 0);
#define LAUNCH_F32(NBV)                                                                                          \
  hipLaunchKernelGGL(k_conv_fwd<NBV>, grid, dim3(256), 0, st, x, (const float4*)wp, tbl, order, tile_mask,       \
                     (long long)n_out, K, cin, cout, bias, y, swz)
#define LAUNCH_SPLIT_I(NBV, PLV, PREV, EPIV)                                                                     \
  hipLaunchKernelGGL((k_conv_fwd_split<NBV, PLV, PREV, EPIV>), sgrid, dim3(256), 0, st, x, (const u32x4*)wp, tbl, \
                     order, tile_mask, (long long)n_out, K, cin, cout, bias, y, sswz, stats, x_amax, w_amax,   \
                     x_bytes, epi)
#define LAUNCH_SPLIT(NBV, PLV)                                                                                   \
  do {                                                                                                           \
    if (PLV == 4 && x_is_planes) {                                                                               \
      if (use_epi) LAUNCH_SPLIT_I(NBV, 4, true, true); else LAUNCH_SPLIT_I(NBV, 4, true, false);                 \
    } else {                                                                                                     \
      if (use_epi) LAUNCH_SPLIT_I(NBV, PLV, false, true); else LAUNCH_SPLIT_I(NBV, PLV, false, false);           \
    }                                                                                                            \
  } while (0)
#define LAUNCH_SPLIT_NB(PLV)                                                             \
  {                                                                                      \
    if (nb == 4) LAUNCH_SPLIT(4, PLV); else if (nb == 2) LAUNCH_SPLIT(2, PLV); else LAUNCH_SPLIT(1, PLV); \
  }
#define LAUNCH_SPLIT_NB2(PLV)                                                            \
  {                                                                                      \
    if (nb >= 2) LAUNCH_SPLIT(2, PLV); else LAUNCH_SPLIT(1, PLV);                        \
  }
  // inference launches (flag GCL_CONV_TALL): sixteen-wave workgroups, the tile's offsets in four fixed groups, for layers
  // with at least tall_min = 108 steps per full tile (27 offsets x Cin / 32 >= 4).  Decided by the layer's shape only, so that
  // a row's bits never depend on the launch it is in.  Measured (bench.py secondary, one pair / eight pairs per pass,
  // M voxels/s): off 15.5 / 80.3, Cin >= 128 (default) 20.1 / 80 (77.0 with ranges of k instead of k mod 4), Cin >= 64 20.9 / 65.6;
  // eval_pairs 146 -> 160 - 165 pairs/s.
  static const int tall = [] { const char* e = getenv("GCL_FWD_TALL"); return e ? atoi(e) : 1; }();
  static const int tall_min = [] { const char* e = getenv("GCL_FWD_TALL_MIN_STEPS"); return e ? atoi(e) : 108; }();
  // Round 6: the same four offset groups as FOUR TIMES AS MANY ordinary workgroups (k_conv_fwd_dma<2, false, false, GRP>) + one
  // sum / epilogue launch, when the caller hands over scratch for the four accumulator slabs (`stats` of a GCL_CONV_TALL
  // launch: gcl_conv_fwd_groups_scratch_len floats).  A pass over one pair leaves 52 sixteen-wave workgroups on 256 CUs,
  // each walking 54 dependent steps at the round trip of a weight block with one step of prefetch (1.3 us per step: 72 us per
  // launch, ten launches = 0.72 of a 1.75 ms pass); as 832 four-wave workgroups three of them share a CU and cover each
  // other's round trips.  Same products in the same order per group, the same group order, the same epilogue expression:
  // bitwise k_conv_fwd_tall's result.
  if (tall && (flags & GCL_CONV_TALL) && stats && gcl_conv_fwd_groups_scratch_len(n_out, K, cin, cout) > 0 && prec == 4 &&
      !x_is_planes && tbl && tile_mask && colgroup && !swz && !ranges) {
    const dim3 ggrid((unsigned)(cdiv(gx, 8) * 8 * (cout / 64) * 4));
    const int gswz = 2 | (heavy_first ? 16 : 0);
    const unsigned w_bytes = (unsigned)((long long)K * cin * cout * 4);
    const ConvEpi none{nullptr, nullptr, 0, nullptr, 0};
    hipLaunchKernelGGL((k_conv_fwd_dma<2, false, false, true>), ggrid, dim3(256), 0, st, x, (const u32x4*)wp, tbl, order, tile_mask,
                       (long long)n_out, K, cin, cout, (const float*)nullptr, stats, gswz, (float*)nullptr, x_amax, w_amax, x_bytes,
                       w_bytes, none);
    // four float4 groups per thread, all sixteen slab loads in flight (GCL_GROUPS_SUM_ILP=1: one group per thread, the same
    // bits): 14.7 -> 9.6 us per launch on a pass over one pair, 27.8 -> 29.0 M voxels/s; eight groups: 28.5
    static const int sum_ilp = [] { const char* e = getenv("GCL_GROUPS_SUM_ILP"); return e ? atoi(e) : 4; }();
    if (sum_ilp == 4) {
      const unsigned sg = (unsigned)cdiv((long long)n_out * cout, 4096);
      if (use_epi)
        hipLaunchKernelGGL((k_conv_groups_sum<true, 4>), dim3(sg), dim3(256), 0, st, (const float*)stats, (long long)n_out, cout,
                           bias, x_amax, w_amax, epi, y);
      else
        hipLaunchKernelGGL((k_conv_groups_sum<false, 4>), dim3(sg), dim3(256), 0, st, (const float*)stats, (long long)n_out, cout,
                           bias, x_amax, w_amax, epi, y);
    } else {
      const unsigned sg = (unsigned)cdiv((long long)n_out * cout, 1024);
      if (use_epi)
        hipLaunchKernelGGL((k_conv_groups_sum<true>), dim3(sg), dim3(256), 0, st, (const float*)stats, (long long)n_out, cout, bias,
                           x_amax, w_amax, epi, y);
      else
        hipLaunchKernelGGL((k_conv_groups_sum<false>), dim3(sg), dim3(256), 0, st, (const float*)stats, (long long)n_out, cout, bias,
                           x_amax, w_amax, epi, y);
    }
    GCL_CHECK_LAUNCH();
    return GCL_OK;
  }
  if (tall && (flags & GCL_CONV_TALL)) stats = nullptr;      // scratch of the group launches, not wanted by this shape / size
  if (tall && (flags & GCL_CONV_TALL) && prec == 4 && !x_is_planes && !stats && tbl && tile_mask && K >= 8 &&
      K * (cin / 32) >= tall_min && cout % 64 == 0 && colgroup && !swz && !ranges) {
    const dim3 tgrid((unsigned)(cdiv(gx, 8) * 8 * (cout / 64)));
    const int tswz = 2 | (heavy_first ? 16 : 0);
    if (use_epi)
      hipLaunchKernelGGL((k_conv_fwd_tall<true>), tgrid, dim3(1024), 0, st, x, (const u32x4*)wp, tbl, order, tile_mask,
                         (long long)n_out, K, cin, cout, bias, y, tswz, x_amax, w_amax, x_bytes, epi);
    else
      hipLaunchKernelGGL((k_conv_fwd_tall<false>), tgrid, dim3(1024), 0, st, x, (const u32x4*)wp, tbl, order, tile_mask,
                         (long long)n_out, K, cin, cout, bias, y, tswz, x_amax, w_amax, x_bytes, epi);
    GCL_CHECK_LAUNCH();
    return GCL_OK;
  }
  // plane-image launches with LDS-DMA staging (k_conv_fwd_dma; default, GCL_FWD_DMA=0 / flag GCL_CONV_NO_DMA select the
  // register-staged k_conv_fwd_split): bitwise the same results, 15 - 21 % shorter launches on the C >= 128 layers of the
  // KITTI batch (profiles/r04_conv_experiments.txt, 19)
  static const int dma = [] { const char* e = getenv("GCL_FWD_DMA"); return e ? atoi(e) : 1; }();
  // fp32-row launches too (the C = 32 / 64 layers; GCL_FWD_DMA_ROWS=0: plane images only)
  static const int dma_rows = [] { const char* e = getenv("GCL_FWD_DMA_ROWS"); return e ? atoi(e) : 1; }();
  if ((dma || (flags & GCL_CONV_DMA)) && !(flags & GCL_CONV_NO_DMA) && prec == 4 && (x_is_planes || dma_rows)) {
    const unsigned w_bytes = (unsigned)((long long)K * cin * cout * 4);
#define LAUNCH_DMA(NBV, PREV, EPIV)                                                                               \
  hipLaunchKernelGGL((k_conv_fwd_dma<NBV, PREV, EPIV>), sgrid, dim3(256), 0, st, x, (const u32x4*)wp, tbl, order,  \
                     tile_mask, (long long)n_out, K, cin, cout, bias, y, sswz, stats, x_amax, w_amax, x_bytes, w_bytes, epi)
#define LAUNCH_DMA_P(NBV)                                                                                         \
  do {                                                                                                            \
    if (x_is_planes) { if (use_epi) LAUNCH_DMA(NBV, true, true); else LAUNCH_DMA(NBV, true, false); }             \
    else { if (use_epi) LAUNCH_DMA(NBV, false, true); else LAUNCH_DMA(NBV, false, false); }                       \
  } while (0)
    if (nb == 4) LAUNCH_DMA_P(4); else if (nb == 2) LAUNCH_DMA_P(2); else LAUNCH_DMA_P(1);
#undef LAUNCH_DMA_P
#undef LAUNCH_DMA
    GCL_CHECK_LAUNCH();
    return GCL_OK;
  }
  if (prec == 0) {
    if (nb == 4) LAUNCH_F32(4); else if (nb == 2) LAUNCH_F32(2); else LAUNCH_F32(1);
  } else if (prec == 2) LAUNCH_SPLIT_NB(2)
  else if (prec == 3) {   // three planes: the double-buffered weight block of NB = 4 would not leave room for 2 WGs/CU
    if (nb == 4) grid = dim3(gx, cout / 64);
    LAUNCH_SPLIT_NB2(3)
  } else LAUNCH_SPLIT_NB(4)
#undef LAUNCH_F32
#undef LAUNCH_SPLIT
#undef LAUNCH_SPLIT_I
#undef LAUNCH_SPLIT_NB
#undef LAUNCH_SPLIT_NB2
  GCL_CHECK_LAUNCH();
  return GCL_OK;
}

int64_t gcl_conv_bwd_weight_scratch_len(int32_t K, int32_t ca, int32_t cb, int64_t n_pairs_padded, int64_t n_sorted_rows) {
  long long nc = n_pairs_padded / GCL_PAIR_CHUNK;
  long long len = (long long)(bwd_weight_wgs(nc) + K) * ca * cb;
  // range-grouped mode (one slab per (row range, offset) + the cell limits): reserved only for the shapes dw_rg_shape can
  // select -- the C >= 128 layers always take the classic path and would otherwise reserve ~100 MB each for nothing
  if (n_sorted_rows >= 32768 && K > 1 && K <= 27 && (ca == 32 || ca == 64) && (cb == 32 || cb == 64)) {
    const long long nr = cdiv(n_sorted_rows, dw_range_rows(n_sorted_rows, K));
    const long long rg = nr * K * ca * cb + (long long)K * (nr + 1) + 64;
    if (rg > len) len = rg;
  }
  return len;
}

int64_t gcl_conv_bwd_weight_bounds_len(int32_t K, int64_t n_sorted_rows) {
  if (n_sorted_rows < 32768 || K <= 1 || K > 27) return 0;
  return (long long)K * (cdiv(n_sorted_rows, dw_range_rows(n_sorted_rows, K)) + 1);
}

int gcl_conv_bwd_weight_bounds(const int32_t* sorted_rows, const int64_t* seg_off_host, int32_t K, int64_t n_sorted_rows,
                               int32_t* bounds, void* stream) {
  GCL_CHECK_ARG(sorted_rows && seg_off_host && bounds && gcl_conv_bwd_weight_bounds_len(K, n_sorted_rows) > 0,
                "gcl_conv_bwd_weight_bounds: bad argument (the range-grouped mode needs 1 < K <= 27 and >= 32768 rows)");
  SegOffW seg;
  for (int k = 0; k <= K; ++k) seg.off[k] = seg_off_host[k];
  const int rr = dw_range_rows(n_sorted_rows, K);
  const int nr = (int)cdiv(n_sorted_rows, rr);
  hipLaunchKernelGGL(k_pair_bounds, dim3((unsigned)cdiv((long long)K * (nr + 1), 256)), dim3(256), 0, (hipStream_t)stream,
                     sorted_rows, seg, K, rr, nr, bounds);
  GCL_CHECK_LAUNCH();
  return GCL_OK;
}

int gcl_conv_bwd_weight(const float* a, int64_t n_a, const float* b, int64_t n_b, int32_t planes, int32_t sorted_side,
                        const int32_t* pair_a, const int32_t* pair_b, const int64_t* seg_off_host, int32_t K,
                        int32_t ca, int32_t cb, int32_t prec,
                        const int32_t* a_amax, const int32_t* b_amax, float* scratch, float* dw, void* stream) {
  return gcl_conv_bwd_weight_rg(a, n_a, b, n_b, planes, sorted_side, pair_a, pair_b, seg_off_host, K, ca, cb, prec, a_amax,
                                b_amax, scratch, dw, nullptr, stream);
}

int gcl_conv_bwd_weight_rg(const float* a, int64_t n_a, const float* b, int64_t n_b, int32_t planes, int32_t sorted_side,
                           const int32_t* pair_a, const int32_t* pair_b, const int64_t* seg_off_host, int32_t K,
                           int32_t ca, int32_t cb, int32_t prec, const int32_t* a_amax, const int32_t* b_amax,
                           float* scratch, float* dw, const int32_t* rg_bounds, void* stream) {
  GCL_CHECK_ARG(a && b && pair_a && pair_b && seg_off_host && scratch && dw, "gcl_conv_bwd_weight: null pointer");
  const bool legacy_dw = (planes & 2) != 0;      // bit 1 of `planes`: the 64 x 64-block kernel for this launch (tests)
  planes &= 1;
  GCL_CHECK_ARG(K >= 1 && K <= 125, "gcl_conv_bwd_weight: bad K");
  GCL_CHECK_ARG(ca > 0 && cb > 0, "gcl_conv_bwd_weight: channel counts (%d, %d) must be positive", ca, cb);
  GCL_CHECK_ARG(prec_ok(prec), "gcl_conv_bwd_weight: prec must be 0, 2, 3 or 4");
  if ((ca % 32) != 0 || (cb % 32) != 0) {     // generic shapes: exact-fp32 VALU kernel, same slabs + ordered reduction
    GCL_CHECK_ARG(!planes, "gcl_conv_bwd_weight: plane images need channel counts that are multiples of 32");
    GCL_CHECK_ARG(n_a > 0 && n_b > 0, "gcl_conv_bwd_weight: empty operand");
    hipStream_t gst = (hipStream_t)stream;
    SegOffW gseg;
    for (int k = 0; k <= K; ++k) gseg.off[k] = seg_off_host[k];
    const long long gnc = gseg.off[K] / GCL_PAIR_CHUNK, gmat = (long long)ca * cb;
    const int gW = bwd_weight_wgs(gnc), gper = (int)cdiv(gnc > 0 ? gnc : 1, gW);
    if (gnc > 0)
      hipLaunchKernelGGL(k_conv_bwd_weight_generic, dim3(gW, (unsigned)(cdiv(ca, 16) * cdiv(cb, 16))), dim3(256), 0, gst, a, b,
                         pair_a, pair_b, gseg, K, ca, cb, gnc, gper, scratch);
    hipLaunchKernelGGL(k_bwd_weight_reduce, dim3((unsigned)cdiv(gmat, 256), K), dim3(256), 0, gst, (const float*)scratch,
                       gseg, gper, gmat, dw);
    GCL_CHECK_LAUNCH();
    return GCL_OK;
  }
  GCL_CHECK_ARG(prec != 4 || (a_amax && b_amax), "gcl_conv_bwd_weight: fp16x3 needs gcl_amax of both operands");
  GCL_CHECK_ARG(!planes || prec == 4, "gcl_conv_bwd_weight: plane images are the fp16x3 operand format");
  GCL_CHECK_ARG(n_a > 0 && n_b > 0 && (long long)n_a * ca * 4 < (1ll << 32) - (1ll << 20) &&
                    (long long)n_b * cb * 4 < (1ll << 32) - (1ll << 20),
                "gcl_conv_bwd_weight: operands must be non-empty and smaller than 4 GiB (buffer addressing)");
  const unsigned a_bytes = (unsigned)((long long)n_a * ca * 4), b_bytes = (unsigned)((long long)n_b * cb * 4);
  hipStream_t st = (hipStream_t)stream;
  SegOffW seg;
  for (int k = 0; k <= K; ++k) seg.off[k] = seg_off_host[k];
  long long nc = seg.off[K] / GCL_PAIR_CHUNK;
  long long mat = (long long)ca * cb;
  int W = bwd_weight_wgs(nc);
  int per = (int)cdiv(nc > 0 ? nc : 1, W);
  GCL_CHECK_ARG(sorted_side >= 0 && sorted_side <= 2, "gcl_conv_bwd_weight: sorted_side must be 0, 1 (pair_a) or 2 (pair_b)");
  const long long n_sorted = sorted_side == 1 ? n_a : n_b;
  if (nc > 0 && dw_rg_shape(K, ca, cb, prec, planes, sorted_side, n_sorted)) {
    const int rr = dw_range_rows(n_sorted, K);
    const int nr = (int)cdiv(n_sorted, rr);
    const int* bounds = (const int*)rg_bounds;      // gcl_conv_bwd_weight_bounds of the sorted list, made once per map ...
    if (!bounds) {                                  // ... or here, per launch
      int* own = (int*)(scratch + (long long)nr * K * mat);
      hipLaunchKernelGGL(k_pair_bounds, dim3((unsigned)cdiv((long long)K * (nr + 1), 256)), dim3(256), 0, st,
                         sorted_side == 1 ? pair_a : pair_b, seg, K, rr, nr, own);
      bounds = own;
    }
    dim3 rgrid((unsigned)(cdiv(nr, 8) * 8 * K));
#define LAUNCH_RG(TA, TB, PLV)                                                                                      \
  hipLaunchKernelGGL((k_conv_bwd_weight_split<TA, TB, PLV, false, true>), rgrid, dim3(256), 0, st, a, b, pair_a, pair_b, \
                     seg, K, ca, cb, nc, per, scratch, a_amax, b_amax, 0, 0, a_bytes, b_bytes, (const int*)bounds, nr)
#define LAUNCH_RG_P(TA, TB)                                            \
  {                                                                    \
    if (prec == 2) LAUNCH_RG(TA, TB, 2);                               \
    else if (prec == 3) LAUNCH_RG(TA, TB, 3);                          \
    else LAUNCH_RG(TA, TB, 4);                                         \
  }
    if (ca == 64 && cb == 64) LAUNCH_RG_P(64, 64)
    else if (ca == 64) LAUNCH_RG_P(64, 32)
    else if (cb == 64) LAUNCH_RG_P(32, 64)
    else LAUNCH_RG_P(32, 32)
#undef LAUNCH_RG_P
#undef LAUNCH_RG
    hipLaunchKernelGGL(k_bwd_weight_reduce_rg, dim3((unsigned)cdiv(mat, 256), K), dim3(256), 0, st, (const float*)scratch,
                       nr, K, mat, dw);
    GCL_CHECK_LAUNCH();
    return GCL_OK;
  }
  // plane images with Ca, Cb multiples of 128: one 128 x 128 block per workgroup, rows gathered once and shared by its four
  // waves (k_conv_bwd_weight_wg128; GCL_DW_WG128=0 / bit 1 of `planes`: the 64 x 64 kernel)
  static const int wg128 = [] { const char* e = getenv("GCL_DW_WG128"); return e ? atoi(e) : 1; }();
  if (nc > 0 && prec == 4 && planes && wg128 && !legacy_dw && ca % 128 == 0 && cb % 128 == 0) {
    const int tiles = (ca / 128) * (cb / 128);
    static const int dwswz2 = [] { const char* s = getenv("GCL_DW_SWIZZLE"); return s ? atoi(s) : 1; }();
    const int stiles = (dwswz2 && tiles > 1) ? tiles : 0;
    const dim3 g2 = stiles ? dim3((unsigned)(cdiv(W, 8) * 8 * stiles)) : dim3(W, tiles);
    hipLaunchKernelGGL((k_conv_bwd_weight_wg128<false>), g2, dim3(256), 0, st, a, b, pair_a, pair_b, seg, K, ca, cb, nc, per,
                       scratch, a_amax, b_amax, W, stiles, a_bytes, b_bytes);
    hipLaunchKernelGGL(k_bwd_weight_reduce, dim3((unsigned)cdiv(mat, 256), K), dim3(256), 0, st, (const float*)scratch,
                       seg, per, mat, dw);
    GCL_CHECK_LAUNCH();
    return GCL_OK;
  }
  if (nc > 0) {
    int tca = (ca % 64 == 0) ? 64 : 32, tcb = (cb % 64 == 0) ? 64 : 32;
    dim3 grid(W, (ca / tca) * (cb / tcb));
    static const int dwswz = [] { const char* s = getenv("GCL_DW_SWIZZLE"); return s ? atoi(s) : 1; }();
    const int stiles = (dwswz && grid.y > 1) ? (int)grid.y : 0;
    dim3 sgrid = stiles ? dim3((unsigned)(cdiv(W, 8) * 8 * stiles)) : grid;
#define LAUNCH_BWS(TA, TB, PLV)                                                                                     \
  hipLaunchKernelGGL((k_conv_bwd_weight_split<TA, TB, PLV>), sgrid, dim3(256), 0, st, a, b, pair_a, pair_b, seg, K, \
                     ca, cb, nc, per, scratch, a_amax, b_amax, W, stiles, a_bytes, b_bytes)
#define LAUNCH_BW(TA, TB)                                                                                          \
  {                                                                                                                \
    if (prec == 0)                                                                                                 \
      hipLaunchKernelGGL((k_conv_bwd_weight<TA, TB>), grid, dim3(256), 0, st, a, b, pair_a, pair_b, seg, K, ca, cb, \
                         nc, per, scratch);                                                                        \
    else if (prec == 2) LAUNCH_BWS(TA, TB, 2);                                                                     \
    else if (prec == 3) LAUNCH_BWS(TA, TB, 3);                                                                     \
    else if (planes)                                                                                               \
      hipLaunchKernelGGL((k_conv_bwd_weight_split<TA, TB, 4, true>), sgrid, dim3(256), 0, st, a, b, pair_a, pair_b, \
                         seg, K, ca, cb, nc, per, scratch, a_amax, b_amax, W, stiles, a_bytes, b_bytes);           \
    else LAUNCH_BWS(TA, TB, 4);                                                                                    \
  }
    if (tca == 64 && tcb == 64) LAUNCH_BW(64, 64)
    else if (tca == 64) LAUNCH_BW(64, 32)
    else if (tcb == 64) LAUNCH_BW(32, 64)
    else LAUNCH_BW(32, 32)
#undef LAUNCH_BW
#undef LAUNCH_BWS
  }
  hipLaunchKernelGGL(k_bwd_weight_reduce, dim3((unsigned)cdiv(mat, 256), K), dim3(256), 0, st, (const float*)scratch,
                     seg, per, mat, dw);
  GCL_CHECK_LAUNCH();
  return GCL_OK;
}

// kernel_size-1 convolutions: dW = A^T B over ALL rows (pair (i, i) for every i), streamed (k_bwd_weight_rows)
static bool dw_rows_shape(int ca, int cb, int prec) {
  static const int on = [] { const char* e = getenv("GCL_DW_ROWS"); return e ? atoi(e) : 1; }();
  return on && prec == 4 && ca % 32 == 0 && cb % 32 == 0 && ca >= 32 && cb >= 32 && ca <= 128 && cb <= 64 &&
         (ca / 32) * (cb / 32) <= 6;
}
static int dw_rows_steps_per_wg(long long n_rows) {
  const long long n_steps = cdiv(n_rows, 16);
  long long per = cdiv(n_steps, 512);        // <= 512 workgroups (two per CU), every wave of a workgroup the same number of steps
  per = cdiv(per, 4) * 4;
  return (int)(per < 4 ? 4 : per);
}

int64_t gcl_conv_bwd_weight_rows_scratch_len(int32_t ca, int32_t cb, int32_t prec, int64_t n_rows) {
  if (n_rows <= 0 || !dw_rows_shape(ca, cb, prec)) return 0;
  return cdiv(cdiv(n_rows, 16), dw_rows_steps_per_wg(n_rows)) * (long long)ca * cb;
}

int gcl_conv_bwd_weight_rows(const float* a, const float* b, int64_t n_rows, int32_t ca, int32_t cb, int32_t prec,
                             const int32_t* a_amax, const int32_t* b_amax, float* scratch, float* dw, void* stream) {
  GCL_CHECK_ARG(a && b && scratch && dw && a_amax && b_amax, "gcl_conv_bwd_weight_rows: null pointer");
  GCL_CHECK_ARG(gcl_conv_bwd_weight_rows_scratch_len(ca, cb, prec, n_rows) > 0,
                "gcl_conv_bwd_weight_rows: shape (%d, %d) / arithmetic %d not taken by the row stream (scratch_len == 0: use "
                "gcl_conv_bwd_weight with identity pairs)", ca, cb, prec);
  GCL_CHECK_ARG((long long)n_rows * ca * 4 < (1ll << 32) - (1ll << 20) && (long long)n_rows * cb * 4 < (1ll << 32) - (1ll << 20),
                "gcl_conv_bwd_weight_rows: operands must be smaller than 4 GiB (buffer addressing)");
  const unsigned a_bytes = (unsigned)((long long)n_rows * ca * 4), b_bytes = (unsigned)((long long)n_rows * cb * 4);
  const int per = dw_rows_steps_per_wg(n_rows);
  const int W = (int)cdiv(cdiv(n_rows, 16), per);
  const int mat = ca * cb;
  hipStream_t st = (hipStream_t)stream;
#define LAUNCH_ROWS(NI, NJ)                                                                                           \
  hipLaunchKernelGGL((k_bwd_weight_rows<NI, NJ>), dim3(W), dim3(256), 0, st, a, b, (long long)n_rows, per, scratch, a_amax, \
                     b_amax, a_bytes, b_bytes)
  const int ni = ca / 32, nj = cb / 32;
  if (nj == 1) {
    if (ni == 1) LAUNCH_ROWS(1, 1); else if (ni == 2) LAUNCH_ROWS(2, 1); else if (ni == 3) LAUNCH_ROWS(3, 1); else LAUNCH_ROWS(4, 1);
  } else {
    if (ni == 1) LAUNCH_ROWS(1, 2); else if (ni == 2) LAUNCH_ROWS(2, 2); else LAUNCH_ROWS(3, 2);
  }
#undef LAUNCH_ROWS
  hipLaunchKernelGGL(k_rows_slab_sum, dim3((unsigned)cdiv(mat, 64)), dim3(256), 0, st, (const float*)scratch, W, mat, dw);
  GCL_CHECK_LAUNCH();
  return GCL_OK;
}

int gcl_stem_fwd(const float* x, const float* w, const int32_t* nbr, int64_t n_out, int32_t K, int32_t cin,
                 int32_t cout, float* y, const uint32_t* presence, const int32_t* not_ones, void* stream) {
  GCL_CHECK_ARG((presence == nullptr) == (not_ones == nullptr), "gcl_stem_fwd: presence and not_ones go together");
  GCL_CHECK_ARG(x && w && nbr && y, "gcl_stem_fwd: null pointer");
  GCL_CHECK_ARG(cin >= 1 && cin <= 4 && cout > 0 && cout % 32 == 0 && K >= 1 && K <= STEM_KMAX && n_out > 0,
                "gcl_stem_fwd: supports Cin <= 4, Cout a multiple of 32, K <= 125 (got %d, %d, %d)", cin, cout, K);
  hipLaunchKernelGGL(k_stem_fwd, dim3((unsigned)cdiv(n_out, 256), (unsigned)(cout / 32)), dim3(256), 0, (hipStream_t)stream,
                     x, w, nbr, (long long)n_out, K, cin, cout, y, (const unsigned*)presence, (const int*)not_ones);
  GCL_CHECK_LAUNCH();
  if (presence && cin == 1) {
    // both kernels are enqueued and the per-row device flags pick the rows each one writes (no host read of the flags)
    hipLaunchKernelGGL(k_stem_fwd_occ, dim3((unsigned)cdiv(n_out, STEM_OCC_ROWS), (unsigned)(cout / 32)), dim3(256), 0,
                       (hipStream_t)stream, w, (const unsigned*)presence, (const int*)not_ones, (long long)n_out, K, cout, y);
    GCL_CHECK_LAUNCH();
  }
  return GCL_OK;
}

// Rows per workgroup of k_stem_bwd_weight: the tiles are dealt evenly to at most `slots` workgroups -- four per CU of an
// MI355X, one round (a fixed 1024 rows per workgroup ran 518 workgroups on 512 places at 0.53 M rows: a second round of
// six).  A constant, not the device's CU count: the slab count fixes the summation order of the result.
static long long stem_rows_per_wg(long long n_out, int cout) {
  static const int slots = [] { const char* e = getenv("GCL_STEM_DW_WGS"); int v = e ? atoi(e) : 1024; return v < 1 ? 1 : v; }();
  int col_blocks = cout / 32;
  long long s = slots / (col_blocks > 0 ? col_blocks : 1);
  if (s < 1) s = 1;
  long long per = cdiv(cdiv(n_out, STEM_TILE), s);
  if (per < 8) per = 8;      // small inputs: few slabs rather than many workgroups
  return per * STEM_TILE;
}

int64_t gcl_stem_bwd_weight_scratch_len(int32_t K, int32_t cin, int32_t cout, int64_t n_out) {
  return cdiv(n_out, stem_rows_per_wg(n_out, cout)) * (long long)K * cin * cout;
}

int gcl_stem_bwd_weight(const float* x, const float* dy, const int32_t* nbr, int64_t n_out, int32_t K, int32_t cin,
                        int32_t cout, float* scratch, float* dw, const uint32_t* presence, const int32_t* not_ones,
                        void* stream) {
  GCL_CHECK_ARG((presence == nullptr) == (not_ones == nullptr), "gcl_stem_bwd_weight: presence and not_ones go together");
  GCL_CHECK_ARG(x && dy && nbr && scratch && dw, "gcl_stem_bwd_weight: null pointer");
  GCL_CHECK_ARG(cin >= 1 && cin <= 4 && cout > 0 && cout % 32 == 0 && K >= 1 && K <= STEM_KMAX && n_out > 0,
                "gcl_stem_bwd_weight: supports Cin <= 4, Cout a multiple of 32, K <= 125 (got %d, %d, %d)", cin, cout, K);
  hipStream_t st = (hipStream_t)stream;
  const long long rows_per_wg = stem_rows_per_wg(n_out, cout);
  int nwg = (int)cdiv(n_out, rows_per_wg);
  long long mat = (long long)K * cin * cout;
  hipLaunchKernelGGL(k_stem_bwd_weight<STEM_TILE>, dim3(nwg, (unsigned)(cout / 32)), dim3(256), 0, st, x, dy, nbr,
                     (long long)n_out, K, cin, cout, scratch, (const unsigned*)presence, (const int*)not_ones, rows_per_wg);
  hipLaunchKernelGGL(k_stem_reduce, dim3((unsigned)cdiv(mat, 16)), dim3(256), 0, st, (const float*)scratch, nwg, mat,
                     dw);
  GCL_CHECK_LAUNCH();
  return GCL_OK;
}

}  // extern "C"
