// BatchNorm over the rows of [n, c] fused with BasicBlock's residual add and ReLU (HBM-bound, float4 access).
//
// Reference path: model/common.py:4-6 (ME.MinkowskiBatchNorm == BatchNorm1d on the feature matrix),
// model/residual_block.py:37-53 (norm -> relu, norm -> += residual -> relu).  One statistics pass (double
// accumulators, per-workgroup partials, ordered finalisation => deterministic) and one apply pass that also
// performs the residual add / ReLU, instead of the reference's separate BN, add and ReLU launches.
#include "common.h"

namespace gcl {

// rows per workgroup of the statistics passes: 1024, halved (down to 64) until the launch has >= 1024 workgroups --
// the deep levels have few, wide rows.  A function of (n, c) only, so the summation order is reproducible.
static inline int bn_rows_per_wg(long long n, int c) {
  int rows = 1024;
  while (rows > 64 && n / rows < 1024) rows >>= 1;
  (void)c;
  return rows;
}

// ReLU sign bits of the forward output: element quad e (float4 granularity) -> bit (e & 63) of the four words
// mask[(e >> 6) * 4 + component] (written by k_bn_apply with one ballot per component; every kernel below maps quad e
// to lane e & 63, so a wave reads four wave-uniform words instead of 1 KB of y).  Returned as a float4 of 1 / 0.
__device__ __forceinline__ float4 mask_as_y(const unsigned long long* __restrict__ mask, long long e) {
  const unsigned long long* m = mask + (e >> 6) * 4;
  const int b = (int)(e & 63);
  return make_float4((float)((m[0] >> b) & 1), (float)((m[1] >> b) & 1), (float)((m[2] >> b) & 1),
                     (float)((m[3] >> b) & 1));
}

// thread t -> channel quad cq = t % (c/4), row lane rl = t / (c/4)
template <bool BWD>
__global__ void __launch_bounds__(256) k_bn_reduce(const float* __restrict__ x, const float* __restrict__ dy,
                                                   const float* __restrict__ y, long long n, int c,
                                                   const float* __restrict__ mean, const float* __restrict__ rstd,
                                                   int relu, const unsigned long long* __restrict__ mask,
                                                   int rows_per_wg, double* partial, int dy_ld = 0, int want_gmax = 0) {
  // want_gmax (BWD): a third run of partials, the workgroup's max|g| per channel after the ReLU gate (k_bn_bwd_final's bound)
  // dy_ld: row pitch of dy in floats (0 = c): dy may be a column slice of a wider tensor (the gradient of an ME.cat input)
  const long long gld = dy_ld ? dy_ld : c;
  __shared__ double red[2][256][4];
  const int cq_n = c >> 2;
  const int cq = threadIdx.x % cq_n, rl = threadIdx.x / cq_n, rstep = 256 / cq_n;
  const long long r_begin = (long long)blockIdx.x * rows_per_wg;
  long long r_end = r_begin + rows_per_wg;
  if (r_end > n) r_end = n;
  double s0[4] = {0, 0, 0, 0}, s1[4] = {0, 0, 0, 0};
  float gm[4] = {0.f, 0.f, 0.f, 0.f};
  float4 mu = make_float4(0, 0, 0, 0), rs = make_float4(1, 1, 1, 1);
  if (BWD) {
    mu = reinterpret_cast<const float4*>(mean)[cq];
    rs = reinterpret_cast<const float4*>(rstd)[cq];
  }
#define BN_ACC(xv, g, yv)                                                                                     \
  if (!BWD) {                                                                                                 \
    s0[0] += xv.x; s0[1] += xv.y; s0[2] += xv.z; s0[3] += xv.w;                                               \
    s1[0] += (double)xv.x * xv.x; s1[1] += (double)xv.y * xv.y;                                               \
    s1[2] += (double)xv.z * xv.z; s1[3] += (double)xv.w * xv.w;                                               \
  } else {                                                                                                    \
    if (relu) {                                                                                               \
      g.x = yv.x > 0.f ? g.x : 0.f; g.y = yv.y > 0.f ? g.y : 0.f;                                             \
      g.z = yv.z > 0.f ? g.z : 0.f; g.w = yv.w > 0.f ? g.w : 0.f;                                             \
    }                                                                                                         \
    s0[0] += g.x; s0[1] += g.y; s0[2] += g.z; s0[3] += g.w;                                                   \
    gm[0] = fmaxf(gm[0], fabsf(g.x)); gm[1] = fmaxf(gm[1], fabsf(g.y));                                       \
    gm[2] = fmaxf(gm[2], fabsf(g.z)); gm[3] = fmaxf(gm[3], fabsf(g.w));                                       \
    s1[0] += (double)g.x * ((xv.x - mu.x) * rs.x); s1[1] += (double)g.y * ((xv.y - mu.y) * rs.y);             \
    s1[2] += (double)g.z * ((xv.z - mu.z) * rs.z); s1[3] += (double)g.w * ((xv.w - mu.w) * rs.w);             \
  }
  // two rows in flight per thread (all loads of an iteration are issued before the first use)
  const float4 z4 = make_float4(0, 0, 0, 0);
  long long r = r_begin + rl;
  for (; r + rstep < r_end; r += 2 * rstep) {
    const long long q = r + rstep;
    float4 xa = reinterpret_cast<const float4*>(x + r * c)[cq], xb = reinterpret_cast<const float4*>(x + q * c)[cq];
    float4 ga = z4, gb = z4, ya = z4, yb = z4;
    if (BWD) {
      ga = reinterpret_cast<const float4*>(dy + r * gld)[cq];
      gb = reinterpret_cast<const float4*>(dy + q * gld)[cq];
      if (relu) {
        if (mask) {
          ya = mask_as_y(mask, r * cq_n + cq);
          yb = mask_as_y(mask, q * cq_n + cq);
        } else {
          ya = reinterpret_cast<const float4*>(y + r * c)[cq];
          yb = reinterpret_cast<const float4*>(y + q * c)[cq];
        }
      }
    }
    BN_ACC(xa, ga, ya)
    BN_ACC(xb, gb, yb)
  }
  if (r < r_end) {
    float4 xa = reinterpret_cast<const float4*>(x + r * c)[cq];
    float4 ga = z4, ya = z4;
    if (BWD) {
      ga = reinterpret_cast<const float4*>(dy + r * gld)[cq];
      if (relu) ya = mask ? mask_as_y(mask, r * cq_n + cq) : reinterpret_cast<const float4*>(y + r * c)[cq];
    }
    BN_ACC(xa, ga, ya)
  }
#undef BN_ACC
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    red[0][threadIdx.x][j] = s0[j];
    red[1][threadIdx.x][j] = s1[j];
  }
  __syncthreads();
  if (rl == 0) {      // channel-major partials [2][c][gridDim.x]: a channel's run is contiguous for reduce_runs
    const long long nwg = gridDim.x;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      double a = 0, b = 0;
      for (int q = 0; q < rstep; ++q) {
        a += red[0][q * cq_n + cq][j];
        b += red[1][q * cq_n + cq][j];
      }
      partial[(long long)(cq * 4 + j) * nwg + blockIdx.x] = a;
      partial[(long long)(c + cq * 4 + j) * nwg + blockIdx.x] = b;
    }
  }
  if (BWD && want_gmax) {      // uniform over the workgroup
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) red[0][threadIdx.x][j] = (double)gm[j];
    __syncthreads();
    if (rl == 0) {
      const long long nwg = gridDim.x;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        double a = 0;
        for (int q = 0; q < rstep; ++q) a = fmax(a, red[0][q * cq_n + cq][j]);
        partial[(long long)(2 * c + cq * 4 + j) * nwg + blockIdx.x] = a;
      }
    }
  }
}

// ordered (deterministic) sum of ONE channel's two runs of per-workgroup partials (channel-major layout) by a 256-thread
// workgroup: thread t adds elements t, t + 256, ... in fp64 (four independent chains, every load coalesced), the 256
// thread sums are then added as a fixed tree -- 16 groups of 16 consecutive threads, then the 16 group sums in order -- so
// the result does not depend on timing.  Valid in thread 0.  (Rounds 2 - 3 kept the partials workgroup-major and read
// them with c / 4 workgroups as 16 / 32-byte pieces a row apart: 11 us per BatchNorm backward, 55 us per 0.5 M-row
// forward layer, all on the training stream.)
template <typename T>
__device__ __forceinline__ void reduce_runs(const T* __restrict__ p1, const T* __restrict__ p2, int n_part, double& s,
                                            double& ss) {
  __shared__ double red[2][256];
  __shared__ double red2[2][16];
  const int t = threadIdx.x;
  double a0 = 0, a1 = 0, a2 = 0, a3 = 0, b0 = 0, b1 = 0, b2 = 0, b3 = 0;
  int w = t;
  for (; w + 768 < n_part; w += 1024) {
    a0 += (double)p1[w];
    a1 += (double)p1[w + 256];
    a2 += (double)p1[w + 512];
    a3 += (double)p1[w + 768];
    b0 += (double)p2[w];
    b1 += (double)p2[w + 256];
    b2 += (double)p2[w + 512];
    b3 += (double)p2[w + 768];
  }
  for (; w < n_part; w += 256) {
    a0 += (double)p1[w];
    b0 += (double)p2[w];
  }
  red[0][t] = (a0 + a1) + (a2 + a3);
  red[1][t] = (b0 + b1) + (b2 + b3);
  __syncthreads();
  if (t < 32) {      // lanes 0..15: the sums' 16 groups, lanes 16..31: the squares'
    const int which = t >> 4, g = t & 15;
    double u = 0;
#pragma unroll
    for (int q = 0; q < 16; ++q) u += red[which][g * 16 + q];
    red2[which][g] = u;
  }
  __syncthreads();
  s = 0;
  ss = 0;
  if (t == 0) {
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      s += red2[0][q];
      ss += red2[1][q];
    }
  }
}

__global__ void __launch_bounds__(256) k_bn_stats_final(const double* __restrict__ partial, int nwg, long long n,
                                                        int c, float eps, float momentum, float* running_mean,
                                                        float* running_var, float* mean, float* rstd) {
  const int ch = blockIdx.x;      // one workgroup per channel
  double s, ss;
  reduce_runs(partial + (long long)ch * nwg, partial + ((long long)c + ch) * nwg, nwg, s, ss);
  if (threadIdx.x != 0) return;
  double m = s / (double)n;
  double var = ss / (double)n - m * m;
  if (var < 0) var = 0;
  mean[ch] = (float)m;
  rstd[ch] = (float)(1.0 / sqrt(var + (double)eps));
  if (running_mean) {
    double unb = (n > 1) ? var * (double)n / (double)(n - 1) : var;
    running_mean[ch] = (float)((1.0 - momentum) * running_mean[ch] + momentum * m);
    running_var[ch] = (float)((1.0 - momentum) * running_var[ch] + momentum * unb);
  }
}

// One-launch finalisation straight from the convolution epilogue's fp32 partials, CHANNEL-MAJOR [2][c][n_part] (one
// partial per 128-row workgroup): one workgroup per channel, thread t adds partials t, t + 256, ... of the channel's two
// contiguous runs in fp64 (four independent chains, every load coalesced), the 256 thread sums are then added as a fixed
// tree (16 groups of 16, then the 16 group sums in order) -- deterministic.  The [n_part][2][c] layout of rounds 2 - 3 was
// read by c / 4 workgroups as 16-byte pieces 2 c floats apart: 55 us per 0.5 M-row layer (4143 partials), on the training
// stream between every convolution and its BatchNorm apply pass.
constexpr int BN_DIRECT_MAX = 1 << 20;
// minimum of run p1 and maximum of run p2 (n_part floats each) over a 256-thread workgroup; order-independent by nature.
// Valid in thread 0.  (Its own LDS: may follow reduce_runs without a barrier in between.)
__device__ __forceinline__ void reduce_min_max(const float* __restrict__ p1, const float* __restrict__ p2, int n_part,
                                               float& lo, float& hi) {
  __shared__ float mm[2][256];
  const int t = threadIdx.x;
  float a = 3.0e38f, b = -3.0e38f;
  for (int w = t; w < n_part; w += 256) {
    a = fminf(a, p1[w]);
    b = fmaxf(b, p2[w]);
  }
  mm[0][t] = a;
  mm[1][t] = b;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (t < o) {
      mm[0][t] = fminf(mm[0][t], mm[0][t + o]);
      mm[1][t] = fmaxf(mm[1][t], mm[1][t + o]);
    }
    __syncthreads();
  }
  lo = mm[0][0];
  hi = mm[1][0];
}

// What the BatchNorm apply pass of channel `ch` will write, bounded BEFORE it runs (round 5): y = (x - mu) rs w + b is a
// monotone function of x in fp32 arithmetic too (every step rounds monotonically), so max|y| over the channel is attained
// at the channel's minimum or maximum of x -- both known from the convolution epilogue's partials -- and is EXACT without
// a residual.  With a residual: + max|residual| (an upper bound); with ReLU only the positive side counts.  `max_with`:
// another tensor that shares the slot (the other input of an ME.cat whose output this BatchNorm writes in place).
struct BnRange {
  float* xrange;           // out [2][c]: channel minimum, maximum of x (kept for the backward pass's bound), or NULL
  const float* weight;     // BatchNorm weight / bias [c]
  const float* bias;
  int relu;
  const int* add_amax;     // amax slot of the residual, or NULL
  const int* max_with;     // amax slot of a tensor that shares y's slot, or NULL
  int* y_amax;             // zero-initialised slot that receives the bound, or NULL: nothing of the above is computed
};
__global__ void __launch_bounds__(256) k_bn_stats_direct(const float* __restrict__ partial, int n_part, long long n, int c,
                                                         float eps, float momentum, float* running_mean,
                                                         float* running_var, float* mean, float* rstd, BnRange rg) {
  const int ch = blockIdx.x;
  double s, ss;
  reduce_runs(partial + (long long)ch * n_part, partial + ((long long)c + ch) * n_part, n_part, s, ss);
  float xlo = 0.f, xhi = 0.f;
  if (rg.y_amax)
    reduce_min_max(partial + ((long long)2 * c + ch) * n_part, partial + ((long long)3 * c + ch) * n_part, n_part, xlo, xhi);
  if (threadIdx.x != 0) return;
  double m = s / (double)n;
  double var = ss / (double)n - m * m;
  if (var < 0) var = 0;
  const float mu = (float)m, rs = (float)(1.0 / sqrt(var + (double)eps));
  mean[ch] = mu;
  rstd[ch] = rs;
  if (running_mean) {
    double unb = (n > 1) ? var * (double)n / (double)(n - 1) : var;
    running_mean[ch] = (float)((1.0 - momentum) * running_mean[ch] + momentum * m);
    running_var[ch] = (float)((1.0 - momentum) * running_var[ch] + momentum * unb);
  }
  if (rg.y_amax) {
    if (rg.xrange) {
      rg.xrange[ch] = xlo;
      rg.xrange[c + ch] = xhi;
    }
    const float w = rg.weight[ch], b = rg.bias[ch];
    const float f_lo = (xlo - mu) * rs * w + b, f_hi = (xhi - mu) * rs * w + b;      // bn_fwd_one's expression
    float bound = rg.relu ? fmaxf(fmaxf(f_lo, f_hi), 0.f) : fmaxf(fabsf(f_lo), fabsf(f_hi));
    if (rg.add_amax) bound += __int_as_float(amax_slot_bits(rg.add_amax));
    if (rg.max_with) bound = fmaxf(bound, __int_as_float(amax_slot_bits(rg.max_with)));
    amax_slot_publish(rg.y_amax, __float_as_int(bound), (unsigned)ch);
  }
}

// Bound of the BatchNorm backward's dx = w rs (g - sum_g / n - xhat sum_gx / n) per channel, before its apply pass runs:
// |dx| <= |w rs| (max|g| + |sum_g| / n + max|xhat| |sum_gx| / n), max|g| of the channel from k_bn_reduce's third run (after
// the ReLU gate), max|xhat| from the forward pass's channel range of x.  Tight to within the triangle inequality.
struct BnBwdRange {
  const float* xrange;     // [2][c] from the forward pass (BnRange.xrange)
  const float* mean;
  const float* rstd;
  const float* weight;
  long long n;
  int* dx_amax;            // zero-initialised slot that receives the bound, or NULL (then the partials have two runs)
};
__global__ void __launch_bounds__(256) k_bn_bwd_final(const double* __restrict__ partial, int nwg, int c,
                                                      float* sum_g, float* sum_gx, BnBwdRange rg) {
  const int ch = blockIdx.x;
  double s, ss;
  reduce_runs(partial + (long long)ch * nwg, partial + ((long long)c + ch) * nwg, nwg, s, ss);
  double gmax = 0;
  if (rg.dx_amax) {      // third run: per-workgroup max|g| of the channel
    __shared__ double gm[256];
    double a = 0;
    const double* p3 = partial + ((long long)2 * c + ch) * nwg;
    for (int w = threadIdx.x; w < nwg; w += 256) a = fmax(a, p3[w]);
    gm[threadIdx.x] = a;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if ((int)threadIdx.x < o) gm[threadIdx.x] = fmax(gm[threadIdx.x], gm[threadIdx.x + o]);
      __syncthreads();
    }
    gmax = gm[0];
  }
  if (threadIdx.x != 0) return;
  sum_g[ch] = (float)s;
  sum_gx[ch] = (float)ss;
  if (rg.dx_amax) {
    const float mu = rg.mean[ch], rs = rg.rstd[ch], inv_n = 1.0f / (float)rg.n;
    const float xh = fmaxf(fabsf(rg.xrange[ch] - mu), fabsf(rg.xrange[c + ch] - mu)) * rs;
    const float bound = fabsf(rg.weight[ch] * rs) * ((float)gmax + fabsf((float)s) * inv_n + xh * fabsf((float)ss) * inv_n);
    // one part in 2^16 of slack: the apply pass rounds its own way (a bound that is low by an ulp would only matter at a
    // power of two, where the fp16 planes still have 4 x headroom)
    amax_slot_publish(rg.dx_amax, __float_as_int(bound * 1.0000153f), (unsigned)ch);
  }
}

// max |v| of the values a workgroup produced -> the gcl_amax slot (common.h)
__device__ __forceinline__ void publish_amax(float m, int* amax_bits) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  __shared__ float wm[4];
  if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    amax_slot_publish(amax_bits, __float_as_int(fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]))), blockIdx.x);
  }
}
__device__ __forceinline__ float amax4(float m, const float4& v) {
  return fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
}

// Both apply kernels: thread -> element quads e, e + S, e + 2S, ... with S = gridDim.x * 256.  HOIST (S a multiple of
// c / 4, the case for every power-of-two width): the thread's channel quad never changes, so its per-channel constants are
// loaded once instead of per quad (the 64-bit e % (c/4) and four to five 16-byte loads per quad went with them).  Two
// quads are in flight per thread (all loads of a pair issued before the first use).  Per-element arithmetic unchanged.
struct BnFwdC { float4 mu, rs, wv, bv; };
__device__ __forceinline__ BnFwdC bn_fwd_c(const float* mean, const float* rstd, const float* weight, const float* bias, int cq) {
  BnFwdC k;
  k.mu = reinterpret_cast<const float4*>(mean)[cq]; k.rs = reinterpret_cast<const float4*>(rstd)[cq];
  k.wv = reinterpret_cast<const float4*>(weight)[cq]; k.bv = reinterpret_cast<const float4*>(bias)[cq];
  return k;
}
__device__ __forceinline__ float4 bn_fwd_one(const float4& xv, const float4& rv, bool has_res, int relu, const BnFwdC& k) {
  float4 o;
  o.x = (xv.x - k.mu.x) * k.rs.x * k.wv.x + k.bv.x;
  o.y = (xv.y - k.mu.y) * k.rs.y * k.wv.y + k.bv.y;
  o.z = (xv.z - k.mu.z) * k.rs.z * k.wv.z + k.bv.z;
  o.w = (xv.w - k.mu.w) * k.rs.w * k.wv.w + k.bv.w;
  if (has_res) { o.x += rv.x; o.y += rv.y; o.z += rv.z; o.w += rv.w; }
  if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
  return o;
}
__device__ __forceinline__ void bn_mask_store(unsigned long long* __restrict__ mask, long long e, const float4& o, bool ok) {
  // e - lane is a multiple of 64 (grid stride and block size are): one ballot per component
  unsigned long long b0 = __ballot(ok && o.x > 0.f), b1 = __ballot(ok && o.y > 0.f), b2 = __ballot(ok && o.z > 0.f),
                     b3 = __ballot(ok && o.w > 0.f);
  if (ok && (threadIdx.x & 63) == 0) {
    unsigned long long* m = mask + (e >> 6) * 4;
    m[0] = b0; m[1] = b1; m[2] = b2; m[3] = b3;
  }
}

template <bool HOIST>
__global__ void __launch_bounds__(256) k_bn_apply(const float* __restrict__ x, long long total4, int c,
                                                  const float* __restrict__ mean, const float* __restrict__ rstd,
                                                  const float* __restrict__ weight, const float* __restrict__ bias,
                                                  const float* __restrict__ residual, int relu,
                                                  float* __restrict__ y, unsigned long long* __restrict__ mask,
                                                  int* amax_bits, int y_ld = 0, unsigned short* __restrict__ planes = nullptr,
                                                  const int* __restrict__ planes_amax = nullptr) {
  // planes (round 5): the fp16 plane image of y is written in the same pass (row pitch = y's, y_ld or c channels), scaled
  // by the slot `planes_amax` -- k_bn_stats_direct's bound of max|y|, final before this launch starts; no publication then
  const float pscale = planes ? amax_scale(planes_amax) : 1.f;
  const int cq_n = c >> 2;
  const long long S = (long long)gridDim.x * blockDim.x;
  const long long e0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  // y_ld != 0: y is a column slice (row pitch y_ld floats) of a wider tensor -- the left columns of an ME.cat's output,
  // written in place (no cat copy of this input).  HOIST: the thread's quad column is fixed, its row advances by S / cq_n.
  const long long yld4 = (y_ld ? y_ld : c) >> 2;
  const long long rstep = HOIST ? S / cq_n : 0;
  long long row_e = e0 / cq_n;
  const int q0 = (int)(e0 % cq_n);
#define BN_Y(E, ROW) (y_ld ? (HOIST ? reinterpret_cast<float4*>(y)[(ROW)*yld4 + q0]                        \
                                    : reinterpret_cast<float4*>(y)[((E) / cq_n) * yld4 + ((E) % cq_n)])   \
                           : reinterpret_cast<float4*>(y)[E])
  const float4 z4 = make_float4(0, 0, 0, 0);
  float am = 0.f;
  BnFwdC ka, kb;
  if (HOIST) ka = kb = bn_fwd_c(mean, rstd, weight, bias, (int)(e0 % cq_n));
  // wave-uniform trip count (e - lane is the same for the wave's lanes; the ballots need the whole wave)
  for (long long e = e0; e - (threadIdx.x & 63) < total4; e += 2 * S) {
    const long long f = e + S;
    const bool oka = e < total4, okb = f < total4;
    if (!HOIST) {
      ka = bn_fwd_c(mean, rstd, weight, bias, (int)((oka ? e : 0) % cq_n));
      kb = bn_fwd_c(mean, rstd, weight, bias, (int)((okb ? f : 0) % cq_n));
    }
    float4 xa = z4, xb = z4, ra = z4, rb = z4;
    if (oka) xa = reinterpret_cast<const float4*>(x)[e];
    if (okb) xb = reinterpret_cast<const float4*>(x)[f];
    if (residual) {
      if (oka) ra = reinterpret_cast<const float4*>(residual)[e];
      if (okb) rb = reinterpret_cast<const float4*>(residual)[f];
    }
    const float4 oa = bn_fwd_one(xa, ra, residual != nullptr, relu, ka);
    const float4 ob = bn_fwd_one(xb, rb, residual != nullptr, relu, kb);
    if (oka) { BN_Y(e, row_e) = oa; am = amax4(am, oa); }
    if (okb) { BN_Y(f, row_e + rstep) = ob; am = amax4(am, ob); }
    if (planes) {
      const long long pld = yld4 * 4;
      if (oka) store_planes4(planes, HOIST ? row_e : e / cq_n, pld, HOIST ? q0 * 4 : (int)(e % cq_n) * 4, oa, pscale);
      if (okb) store_planes4(planes, HOIST ? row_e + rstep : f / cq_n, pld, HOIST ? q0 * 4 : (int)(f % cq_n) * 4, ob, pscale);
    }
    row_e += 2 * rstep;
    if (mask) {
      bn_mask_store(mask, e, oa, oka);
      if (f - (threadIdx.x & 63) < total4) bn_mask_store(mask, f, ob, okb);
    }
  }
#undef BN_Y
  if (amax_bits && !planes) publish_amax(am, amax_bits);
}

struct BnBwdC { float4 mu, rs, wv, sg, sx; };
__device__ __forceinline__ BnBwdC bn_bwd_c(const float* mean, const float* rstd, const float* weight, const float* sum_g,
                                           const float* sum_gx, int cq) {
  BnBwdC k;
  k.mu = reinterpret_cast<const float4*>(mean)[cq]; k.rs = reinterpret_cast<const float4*>(rstd)[cq];
  k.wv = reinterpret_cast<const float4*>(weight)[cq];
  k.sg = reinterpret_cast<const float4*>(sum_g)[cq]; k.sx = reinterpret_cast<const float4*>(sum_gx)[cq];
  return k;
}
__device__ __forceinline__ float4 bn_bwd_one(const float4& xv, const float4& g, float inv_n, const BnBwdC& k) {
  float4 o;
  o.x = k.wv.x * k.rs.x * (g.x - k.sg.x * inv_n - (xv.x - k.mu.x) * k.rs.x * k.sx.x * inv_n);
  o.y = k.wv.y * k.rs.y * (g.y - k.sg.y * inv_n - (xv.y - k.mu.y) * k.rs.y * k.sx.y * inv_n);
  o.z = k.wv.z * k.rs.z * (g.z - k.sg.z * inv_n - (xv.z - k.mu.z) * k.rs.z * k.sx.z * inv_n);
  o.w = k.wv.w * k.rs.w * (g.w - k.sg.w * inv_n - (xv.w - k.mu.w) * k.rs.w * k.sx.w * inv_n);
  return o;
}
__device__ __forceinline__ void relu_gate(float4& g, const float4& yv) {
  g.x = yv.x > 0.f ? g.x : 0.f; g.y = yv.y > 0.f ? g.y : 0.f;
  g.z = yv.z > 0.f ? g.z : 0.f; g.w = yv.w > 0.f ? g.w : 0.f;
}

template <bool HOIST>
__global__ void __launch_bounds__(256) k_bn_bwd_apply(const float* __restrict__ x, const float* __restrict__ dy,
                                                      const float* __restrict__ y, long long total4, int c,
                                                      float inv_n, const float* __restrict__ mean,
                                                      const float* __restrict__ rstd,
                                                      const float* __restrict__ weight,
                                                      const float* __restrict__ sum_g,
                                                      const float* __restrict__ sum_gx, int relu,
                                                      const unsigned long long* __restrict__ mask,
                                                      float* __restrict__ dx, float* __restrict__ dres,
                                                      int* amax_bits, int dy_ld = 0,
                                                      unsigned short* __restrict__ planes = nullptr,
                                                      const int* __restrict__ planes_amax = nullptr) {
  // planes: the fp16 plane image of dx in the same pass, scaled by k_bn_bwd_final's bound (slot `planes_amax`)
  const float pscale = planes ? amax_scale(planes_amax) : 1.f;
  const int cq_n = c >> 2;
  const long long S = (long long)gridDim.x * blockDim.x;
  const long long e0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  // dy_ld != 0: dy is a column slice (row pitch dy_ld floats) of a wider tensor; element quad e = (row e / cq_n, quad e % cq_n)
  const long long gld4 = (dy_ld ? dy_ld : c) >> 2;
  // HOIST: the thread's quad column never changes and its row advances by S / cq_n per stride -- no division in the loop
  const long long rstep = HOIST ? S / cq_n : 0;
  long long row_e = e0 / cq_n;
  const int q0 = (int)(e0 % cq_n);
#define BN_DY(E, ROW) (dy_ld ? (HOIST ? reinterpret_cast<const float4*>(dy)[(ROW)*gld4 + q0]                       \
                                      : reinterpret_cast<const float4*>(dy)[((E) / cq_n) * gld4 + ((E) % cq_n)])  \
                             : reinterpret_cast<const float4*>(dy)[E])
  const float4 z4 = make_float4(0, 0, 0, 0);
  float am = 0.f;
  BnBwdC ka, kb;
  if (HOIST) ka = kb = bn_bwd_c(mean, rstd, weight, sum_g, sum_gx, (int)(e0 % cq_n));
  for (long long e = e0; e < total4; e += 2 * S) {
    const long long f = e + S;
    const bool okb = f < total4;
    if (!HOIST) {
      ka = bn_bwd_c(mean, rstd, weight, sum_g, sum_gx, (int)(e % cq_n));
      kb = bn_bwd_c(mean, rstd, weight, sum_g, sum_gx, (int)((okb ? f : 0) % cq_n));
    }
    float4 xa = reinterpret_cast<const float4*>(x)[e], ga = BN_DY(e, row_e);
    float4 xb = z4, gb = z4, ya = z4, yb = z4;
    if (okb) {
      xb = reinterpret_cast<const float4*>(x)[f];
      gb = BN_DY(f, row_e + rstep);
    }
    row_e += 2 * rstep;
    if (relu) {
      ya = mask ? mask_as_y(mask, e) : reinterpret_cast<const float4*>(y)[e];
      if (okb) yb = mask ? mask_as_y(mask, f) : reinterpret_cast<const float4*>(y)[f];
      relu_gate(ga, ya);
      relu_gate(gb, yb);
    }
    const float4 oa = bn_bwd_one(xa, ga, inv_n, ka);
    reinterpret_cast<float4*>(dx)[e] = oa;
    am = amax4(am, oa);
    if (planes) store_planes4(planes, e / cq_n, c, (int)(e % cq_n) * 4, oa, pscale);
    if (dres) reinterpret_cast<float4*>(dres)[e] = ga;
    if (okb) {
      const float4 ob = bn_bwd_one(xb, gb, inv_n, kb);
      reinterpret_cast<float4*>(dx)[f] = ob;
      am = amax4(am, ob);
      if (planes) store_planes4(planes, f / cq_n, c, (int)(f % cq_n) * 4, ob, pscale);
      if (dres) reinterpret_cast<float4*>(dres)[f] = gb;
    }
  }
#undef BN_DY
  if (amax_bits && !planes) publish_amax(am, amax_bits);
}

// ---- row-wise L2 normalisation  y = x / ||x||_2  (model/resunet.py:226-230) -------------------------------------
// c/4 lanes per row (power of two <= 64), float4 per lane.  No epsilon, like the reference: an all-zero row gives NaN.
// backward: dx = (dy - y (y . dy)) / ||x||
template <bool BWD>
__global__ void __launch_bounds__(256) k_row_normalize(const float* __restrict__ a, const float* __restrict__ dy,
                                                       const float* __restrict__ norm_in, long long n, int c,
                                                       float* __restrict__ out, float* __restrict__ norm_out,
                                                       int* out_amax) {
  const int lpr = c >> 2;
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long row = e / lpr;
  const bool ok = row < n;
  float4 v = make_float4(0, 0, 0, 0), g = make_float4(0, 0, 0, 0);
  if (ok) {
    v = reinterpret_cast<const float4*>(a)[e];
    if (BWD) g = reinterpret_cast<const float4*>(dy)[e];
  }
  float s = BWD ? (v.x * g.x + v.y * g.y + v.z * g.z + v.w * g.w) : (v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w);
  for (int o = 1; o < lpr; o <<= 1) s += __shfl_xor(s, o);
  float m = 0.f;
  if (ok) {
    float4 o4;
    if (!BWD) {
      const float nr = sqrtf(s);
      o4 = make_float4(v.x / nr, v.y / nr, v.z / nr, v.w / nr);
      if ((e % lpr) == 0) norm_out[row] = nr;
    } else {
      const float nr = norm_in[row];
      o4 = make_float4((g.x - v.x * s) / nr, (g.y - v.y * s) / nr, (g.z - v.z * s) / nr, (g.w - v.w * s) / nr);
    }
    reinterpret_cast<float4*>(out)[e] = o4;
    m = amax4(m, o4);
  }
  if (out_amax) publish_amax(m, out_amax);      // every thread of the workgroup gets here
}

// ---- SGD with momentum and weight decay over a LIST of tensors in one launch ---------------------------------------
// torch.optim.SGD's update (lib/colocation_trainer.py:73-77: lr, momentum, weight_decay; dampening 0, no Nesterov):
//   d = g + wd * p;   buf = first ? d : momentum * buf + d;   p -= lr * buf
// table[t] = {p, g, buf} device pointers, sizes[t] elements; grid = (chunks, tensors).  Replaces the 7 multi-tensor
// launches of torch's foreach implementation (and its ~0.6 ms of host time) per optimizer step.
struct SgdPtrs { float* p; const float* g; float* buf; };
__global__ void __launch_bounds__(256) k_sgd_multi(const SgdPtrs* __restrict__ table, const long long* __restrict__ sizes,
                                                   float lr, float momentum, float wd, int first) {
  const SgdPtrs t = table[blockIdx.y];
  const long long n = sizes[blockIdx.y];
  // float4 body when the three pointers are 16-byte aligned (tensors seated in a flat buffer at arbitrary offsets may not
  // be), scalar tail / fallback; the arithmetic per element is the same either way
  const bool vec = ((((unsigned long long)t.p) | ((unsigned long long)t.g) | ((unsigned long long)t.buf)) & 15ull) == 0;
  const long long n4 = vec ? n / 4 : 0;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n4; e += (long long)gridDim.x * blockDim.x) {
    float4 p = reinterpret_cast<const float4*>(t.p)[e];
    const float4 g = reinterpret_cast<const float4*>(t.g)[e];
    float4 b = make_float4(0, 0, 0, 0);
    if (!first) b = reinterpret_cast<const float4*>(t.buf)[e];
    const float d0 = g.x + wd * p.x, d1 = g.y + wd * p.y, d2 = g.z + wd * p.z, d3 = g.w + wd * p.w;
    b.x = first ? d0 : momentum * b.x + d0;
    b.y = first ? d1 : momentum * b.y + d1;
    b.z = first ? d2 : momentum * b.z + d2;
    b.w = first ? d3 : momentum * b.w + d3;
    reinterpret_cast<float4*>(t.buf)[e] = b;
    p.x -= lr * b.x;
    p.y -= lr * b.y;
    p.z -= lr * b.z;
    p.w -= lr * b.w;
    reinterpret_cast<float4*>(t.p)[e] = p;
  }
  for (long long e = n4 * 4 + (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x) {
    const float p = t.p[e];
    const float d = t.g[e] + wd * p;
    const float b = first ? d : momentum * t.buf[e] + d;
    t.buf[e] = b;
    t.p[e] = p - lr * b;
  }
}

static bool bn_c_ok(int c) { return c >= 4 && c % 4 == 0 && (256 % (c / 4)) == 0; }

// column sums of a matrix from k_bn_reduce<false>'s per-workgroup fp64 partials (ordered => deterministic)
__global__ void __launch_bounds__(256) k_col_sum_final(const double* __restrict__ partial, int nwg, int c, float* out) {
  const int ch = blockIdx.x;
  double s, ss;
  reduce_runs(partial + (long long)ch * nwg, partial + ((long long)c + ch) * nwg, nwg, s, ss);
  if (threadIdx.x != 0) return;
  out[ch] = (float)s;
}

}  // namespace gcl

using namespace gcl;

extern "C" {

int64_t gcl_bn_scratch_len(int64_t n, int32_t c) { return cdiv(n, bn_rows_per_wg(n, c)) * 3 * c; }      // sum, xhat-weighted sum, max|g|

int gcl_bn_stats(const float* x, int64_t n, int32_t c, float eps, float momentum, float* running_mean,
                 float* running_var, double* scratch, float* mean, float* rstd, void* stream) {
  GCL_CHECK_ARG(x && scratch && mean && rstd, "gcl_bn_stats: null pointer");
  GCL_CHECK_ARG(n > 0 && bn_c_ok(c), "gcl_bn_stats: unsupported shape n=%lld c=%d (c/4 must divide 256)", (long long)n, c);
  hipStream_t st = (hipStream_t)stream;
  const int rows = bn_rows_per_wg(n, c);
  int nwg = (int)cdiv(n, rows);
  hipLaunchKernelGGL(k_bn_reduce<false>, dim3(nwg), dim3(256), 0, st, x, (const float*)nullptr,
                     (const float*)nullptr, (long long)n, c, (const float*)nullptr, (const float*)nullptr, 0,
                     (const unsigned long long*)nullptr, rows, scratch);
  hipLaunchKernelGGL(k_bn_stats_final, dim3((unsigned)c), dim3(256), 0, st, (const double*)scratch, nwg,
                     (long long)n, c, eps, momentum, running_mean, running_var, mean, rstd);
  GCL_CHECK_LAUNCH();
  return GCL_OK;
}

int64_t gcl_bn_tiles_scratch_len(int64_t n_tiles, int32_t c) { (void)n_tiles; return 2 * (int64_t)c; }      // unused since round 4

int gcl_bn_stats_from_tiles(const float* partial, int64_t n_tiles, int64_t n, int32_t c, float eps, float momentum,
                            float* running_mean, float* running_var, double* scratch, float* mean, float* rstd,
                            void* stream) {
  (void)scratch;
  return gcl_bn_stats_from_tiles_range(partial, n_tiles, n, c, eps, momentum, running_mean, running_var, mean, rstd, nullptr,
                                       nullptr, nullptr, 0, nullptr, nullptr, nullptr, stream);
}

int gcl_bn_stats_from_tiles_range(const float* partial, int64_t n_tiles, int64_t n, int32_t c, float eps, float momentum,
                                  float* running_mean, float* running_var, float* mean, float* rstd, float* xrange,
                                  const float* weight, const float* bias, int32_t relu, const int32_t* add_amax,
                                  const int32_t* max_with, int32_t* y_amax, void* stream) {
  GCL_CHECK_ARG(partial && mean && rstd, "gcl_bn_stats_from_tiles: null pointer");
  GCL_CHECK_ARG(n > 0 && n_tiles > 0 && n_tiles <= BN_DIRECT_MAX && c > 0, "gcl_bn_stats_from_tiles: bad sizes");
  GCL_CHECK_ARG(!y_amax || (weight && bias), "gcl_bn_stats_from_tiles_range: the bound of max|y| needs the BatchNorm's weight and bias");
  const BnRange rg{xrange, weight, bias, relu, (const int*)add_amax, (const int*)max_with, (int*)y_amax};
  hipLaunchKernelGGL(k_bn_stats_direct, dim3((unsigned)c), dim3(256), 0, (hipStream_t)stream, partial, (int)n_tiles,
                     (long long)n, c, eps, momentum, running_mean, running_var, mean, rstd, rg);
  GCL_CHECK_LAUNCH();
  return GCL_OK;
}

int gcl_bn_apply(const float* x, int64_t n, int32_t c, const float* mean, const float* rstd, const float* weight,
                 const float* bias, const float* residual, int32_t relu, float* y, uint64_t* relu_mask,
                 int32_t* y_amax, void* stream) {
  return gcl_bn_apply_ld(x, n, c, mean, rstd, weight, bias, residual, relu, y, 0, relu_mask, y_amax, stream);
}

int gcl_bn_apply_ld(const float* x, int64_t n, int32_t c, const float* mean, const float* rstd, const float* weight,
                    const float* bias, const float* residual, int32_t relu, float* y, int32_t y_ld, uint64_t* relu_mask,
                    int32_t* y_amax, void* stream) {
  return gcl_bn_apply_planes(x, n, c, mean, rstd, weight, bias, residual, relu, y, y_ld, relu_mask, y_amax, nullptr, stream);
}

int gcl_bn_apply_planes(const float* x, int64_t n, int32_t c, const float* mean, const float* rstd, const float* weight,
                        const float* bias, const float* residual, int32_t relu, float* y, int32_t y_ld, uint64_t* relu_mask,
                        int32_t* y_amax, void* planes, void* stream) {
  GCL_CHECK_ARG(x && mean && rstd && weight && bias && y, "gcl_bn_apply: null pointer");
  GCL_CHECK_ARG(n > 0 && c >= 4 && c % 4 == 0, "gcl_bn_apply: unsupported shape");
  GCL_CHECK_ARG(y_ld == 0 || (y_ld >= c && y_ld % 4 == 0), "gcl_bn_apply_ld: y_ld must be 0 or a multiple of 4 >= c");
  GCL_CHECK_ARG(!planes || (y_amax && c % 32 == 0 && (y_ld % 32) == 0),
                "gcl_bn_apply_planes: a plane image needs the slot that holds the bound of max|y| and widths that are multiples of 32");
  long long total4 = n * (c / 4);
  long long g = cdiv(total4, 256);
  if (g > 4096) g = 4096;
  if ((g * 256) % (c / 4) == 0)
    hipLaunchKernelGGL(k_bn_apply<true>, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, x, total4, c, mean, rstd,
                       weight, bias, residual, relu, y, (unsigned long long*)(relu ? relu_mask : nullptr), y_amax, y_ld,
                       (unsigned short*)planes, (const int*)y_amax);
  else
    hipLaunchKernelGGL(k_bn_apply<false>, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, x, total4, c, mean, rstd,
                       weight, bias, residual, relu, y, (unsigned long long*)(relu ? relu_mask : nullptr), y_amax, y_ld,
                       (unsigned short*)planes, (const int*)y_amax);
  GCL_CHECK_LAUNCH();
  return GCL_OK;
}

int64_t gcl_bn_mask_len(int64_t n, int32_t c) { return cdiv(n * (c / 4), 64) * 4; }

int gcl_bn_bwd_reduce(const float* x, const float* dy, const float* y, const uint64_t* relu_mask, int64_t n,
                      int32_t c, const float* mean, const float* rstd, int32_t relu, double* scratch, float* sum_g,
                      float* sum_gx, void* stream) {
  return gcl_bn_bwd_reduce_ld(x, dy, 0, y, relu_mask, n, c, mean, rstd, relu, scratch, sum_g, sum_gx, stream);
}

int gcl_bn_bwd_reduce_ld(const float* x, const float* dy, int32_t dy_ld, const float* y, const uint64_t* relu_mask, int64_t n,
                         int32_t c, const float* mean, const float* rstd, int32_t relu, double* scratch, float* sum_g,
                         float* sum_gx, void* stream) {
  return gcl_bn_bwd_reduce_range(x, dy, dy_ld, y, relu_mask, n, c, mean, rstd, relu, scratch, sum_g, sum_gx, nullptr, nullptr,
                                 nullptr, stream);
}

int gcl_bn_bwd_reduce_range(const float* x, const float* dy, int32_t dy_ld, const float* y, const uint64_t* relu_mask,
                            int64_t n, int32_t c, const float* mean, const float* rstd, int32_t relu, double* scratch,
                            float* sum_g, float* sum_gx, const float* xrange, const float* weight, int32_t* dx_amax,
                            void* stream) {
  GCL_CHECK_ARG(dy_ld == 0 || (dy_ld >= c && dy_ld % 4 == 0), "gcl_bn_bwd_reduce: dy_ld must be 0 or a multiple of 4 >= c");
  GCL_CHECK_ARG(x && dy && mean && rstd && scratch && sum_g && sum_gx, "gcl_bn_bwd_reduce: null pointer");
  GCL_CHECK_ARG(!relu || y || relu_mask, "gcl_bn_bwd_reduce: y or relu_mask is required when relu is set");
  GCL_CHECK_ARG(n > 0 && bn_c_ok(c), "gcl_bn_bwd_reduce: unsupported shape n=%lld c=%d", (long long)n, c);
  GCL_CHECK_ARG(!dx_amax || (xrange && weight), "gcl_bn_bwd_reduce_range: the bound of max|dx| needs xrange (forward pass) and the weight");
  hipStream_t st = (hipStream_t)stream;
  const int rows = bn_rows_per_wg(n, c);
  int nwg = (int)cdiv(n, rows);
  hipLaunchKernelGGL(k_bn_reduce<true>, dim3(nwg), dim3(256), 0, st, x, dy, y, (long long)n, c, mean, rstd, relu,
                     (const unsigned long long*)relu_mask, rows, scratch, dy_ld, dx_amax ? 1 : 0);
  const BnBwdRange rg{xrange, mean, rstd, weight, (long long)n, (int*)dx_amax};
  hipLaunchKernelGGL(k_bn_bwd_final, dim3((unsigned)c), dim3(256), 0, st, (const double*)scratch, nwg, c,
                     sum_g, sum_gx, rg);
  GCL_CHECK_LAUNCH();
  return GCL_OK;
}

int gcl_bn_bwd_apply(const float* x, const float* dy, const float* y, const uint64_t* relu_mask, int64_t n, int32_t c,
                     const float* mean, const float* rstd, const float* weight, const float* sum_g,
                     const float* sum_gx, int32_t relu, float* dx, float* dres, int32_t* dx_amax, void* stream) {
  return gcl_bn_bwd_apply_ld(x, dy, 0, y, relu_mask, n, c, mean, rstd, weight, sum_g, sum_gx, relu, dx, dres, dx_amax, stream);
}

int gcl_bn_bwd_apply_ld(const float* x, const float* dy, int32_t dy_ld, const float* y, const uint64_t* relu_mask, int64_t n,
                        int32_t c, const float* mean, const float* rstd, const float* weight, const float* sum_g,
                        const float* sum_gx, int32_t relu, float* dx, float* dres, int32_t* dx_amax, void* stream) {
  return gcl_bn_bwd_apply_planes(x, dy, dy_ld, y, relu_mask, n, c, mean, rstd, weight, sum_g, sum_gx, relu, dx, dres, dx_amax,
                                 nullptr, stream);
}

int gcl_bn_bwd_apply_planes(const float* x, const float* dy, int32_t dy_ld, const float* y, const uint64_t* relu_mask, int64_t n,
                            int32_t c, const float* mean, const float* rstd, const float* weight, const float* sum_g,
                            const float* sum_gx, int32_t relu, float* dx, float* dres, int32_t* dx_amax, void* planes,
                            void* stream) {
  GCL_CHECK_ARG(dy_ld == 0 || (dy_ld >= c && dy_ld % 4 == 0), "gcl_bn_bwd_apply: dy_ld must be 0 or a multiple of 4 >= c");
  GCL_CHECK_ARG(x && dy && mean && rstd && weight && sum_g && sum_gx && dx, "gcl_bn_bwd_apply: null pointer");
  GCL_CHECK_ARG(!relu || y || relu_mask, "gcl_bn_bwd_apply: y or relu_mask is required when relu is set");
  GCL_CHECK_ARG(n > 0 && c >= 4 && c % 4 == 0, "gcl_bn_bwd_apply: unsupported shape");
  GCL_CHECK_ARG(!planes || (dx_amax && c % 32 == 0),
                "gcl_bn_bwd_apply_planes: a plane image needs the slot that holds the bound of max|dx| and c a multiple of 32");
  long long total4 = n * (c / 4);
  long long g = cdiv(total4, 256);
  if (g > 4096) g = 4096;
  if ((g * 256) % (c / 4) == 0)
    hipLaunchKernelGGL(k_bn_bwd_apply<true>, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, x, dy, y, total4, c,
                       1.0f / (float)n, mean, rstd, weight, sum_g, sum_gx, relu, (const unsigned long long*)relu_mask, dx,
                       dres, dx_amax, dy_ld, (unsigned short*)planes, (const int*)dx_amax);
  else
    hipLaunchKernelGGL(k_bn_bwd_apply<false>, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, x, dy, y, total4, c,
                       1.0f / (float)n, mean, rstd, weight, sum_g, sum_gx, relu, (const unsigned long long*)relu_mask, dx,
                       dres, dx_amax, dy_ld, (unsigned short*)planes, (const int*)dx_amax);
  GCL_CHECK_LAUNCH();
  return GCL_OK;
}

static bool rownorm_c_ok(int c) { return c >= 4 && c <= 256 && (c & (c - 1)) == 0; }

int gcl_row_normalize_fwd(const float* x, int64_t n, int32_t c, float* y, float* norm, void* stream) {
  GCL_CHECK_ARG(x && y && norm, "gcl_row_normalize_fwd: null pointer");
  GCL_CHECK_ARG(n > 0 && rownorm_c_ok(c), "gcl_row_normalize_fwd: c must be a power of two in [4, 256] (got %d)", c);
  long long total = n * (c / 4);
  hipLaunchKernelGGL(k_row_normalize<false>, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, x,
                     (const float*)nullptr, (const float*)nullptr, (long long)n, c, y, norm, (int*)nullptr);
  GCL_CHECK_LAUNCH();
  return GCL_OK;
}

int gcl_row_normalize_bwd(const float* y, const float* dy, const float* norm, int64_t n, int32_t c, float* dx,
                          int32_t* dx_amax, void* stream) {
  GCL_CHECK_ARG(y && dy && norm && dx, "gcl_row_normalize_bwd: null pointer");
  GCL_CHECK_ARG(n > 0 && rownorm_c_ok(c), "gcl_row_normalize_bwd: c must be a power of two in [4, 256] (got %d)", c);
  long long total = n * (c / 4);
  hipLaunchKernelGGL(k_row_normalize<true>, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, y, dy,
                     norm, (long long)n, c, dx, (float*)nullptr, dx_amax);
  GCL_CHECK_LAUNCH();
  return GCL_OK;
}

int gcl_col_sum(const float* x, int64_t n, int32_t c, double* scratch, float* out, void* stream) {
  GCL_CHECK_ARG(x && scratch && out, "gcl_col_sum: null pointer");
  GCL_CHECK_ARG(n > 0 && bn_c_ok(c), "gcl_col_sum: unsupported shape n=%lld c=%d (c/4 must divide 256)", (long long)n, c);
  hipStream_t st = (hipStream_t)stream;
  const int rows = bn_rows_per_wg(n, c);
  int nwg = (int)cdiv(n, rows);
  hipLaunchKernelGGL(k_bn_reduce<false>, dim3(nwg), dim3(256), 0, st, x, (const float*)nullptr,
                     (const float*)nullptr, (long long)n, c, (const float*)nullptr, (const float*)nullptr, 0,
                     (const unsigned long long*)nullptr, rows, scratch);
  hipLaunchKernelGGL(k_col_sum_final, dim3((unsigned)c), dim3(256), 0, st, (const double*)scratch, nwg, c, out);
  GCL_CHECK_LAUNCH();
  return GCL_OK;
}

int gcl_sgd_multi(const void* table, const int64_t* sizes, int32_t n_tensors, float lr, float momentum,
                  float weight_decay, int32_t first, void* stream) {
  GCL_CHECK_ARG(table && sizes && n_tensors > 0 && n_tensors <= 65535, "gcl_sgd_multi: bad argument");
  hipLaunchKernelGGL(k_sgd_multi, dim3(128, (unsigned)n_tensors), dim3(256), 0, (hipStream_t)stream,
                     (const SgdPtrs*)table, (const long long*)sizes, lr, momentum, weight_decay, first);
  GCL_CHECK_LAUNCH();
  return GCL_OK;
}

}  // extern "C"
