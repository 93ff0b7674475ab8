// GCL group-wise contrastive loss kernels and feature-space 1-NN (fp32).
//
// Reference: lib/colocation_trainer.py:430-535 (finest_contrastive_loss, square_loss path), lib/metrics.py:22-29
// (pdist), lib/eval.py:18-48 (find_nn_gpu).  The reference runs a Python loop of <=1024 groups x ~8 micro
// launches, materialises an [M, M, C] broadcast tensor for pdist and returns to the host for the positive-pair
// mask; here each selected group is one wavefront (lane = channel), the pairwise distance + row minimum never
// leaves registers/LDS, and the positive-pair mask is evaluated on the device from (index, group) directly.
#include "common.h"

#include <limits.h>
#include <math.h>

namespace gcl {

// ---- positive / finest loss: one wave per selected group, lane = channel ------------------------------
// flags (config switches of finest_contrastive_loss, lib/colocation_trainer.py:463-488 / location_contrastive_loss
// :768-776):  GL_SQRT   square_loss == False        distances enter as sqrt(d2 + 1e-7)
//             GL_BLOCK  block_finest_gradient        mean of the NON-finest members vs the detached finest member
//                                                    (always the sqrt form, :480-481)
//             GL_PAIR   use_pair_group_positive_loss two drawn members instead of the spread about the mean
//             GL_NOFIN  location_contrastive_loss    no finest term
constexpr int GL_SQRT = 1, GL_BLOCK = 2, GL_PAIR = 4, GL_NOFIN = 8;

struct GroupStats {
  float mean, mean_b, ft;     // lane's channel of: mean of all members, mean of the non-finest members, finest member
  long long tpos;             // position (in index) of the first finest member
  float inv_n, inv_nb;
};

__device__ __forceinline__ GroupStats group_stats(const float* __restrict__ f, int c,
                                                  const long long* __restrict__ index,
                                                  const unsigned char* __restrict__ flag, long long b, long long e,
                                                  int lane) {
  GroupStats g;
  float s = 0.f, sb = 0.f;
  int nb = 0;
  g.tpos = -1;
  for (long long j = b; j < e; ++j) {
    const float v = (lane < c) ? f[index[j] * c + lane] : 0.f;
    s += v;
    if (flag[j]) {
      if (g.tpos < 0) g.tpos = j;
    } else {
      sb += v;
      ++nb;
    }
  }
  g.inv_n = 1.f / (float)(e - b);
  g.inv_nb = 1.f / (float)nb;          // nb == 0: inf, and mean_b = 0 * inf = NaN like torch.mean of an empty set
  g.mean = s * g.inv_n;
  g.mean_b = sb * g.inv_nb;
  if (g.tpos < 0) g.tpos = b;          // malformed input (no flag): fall back to the first member
  g.ft = (lane < c) ? f[index[g.tpos] * c + lane] : 0.f;
  return g;
}

__device__ __forceinline__ float relu_keep_nan(float v) { return v != v ? v : fmaxf(v, 0.f); }

// the two scalar terms (before the threshold) of one group; pp = the two drawn member positions (GL_PAIR)
__device__ __forceinline__ void group_terms(const float* __restrict__ f, int c, const long long* __restrict__ index,
                                            long long b, long long e, int lane, const GroupStats& g, int flags,
                                            const int* __restrict__ pp, float& posv, float& finv) {
  if (flags & GL_PAIR) {
    const float fa = (lane < c) ? f[index[b + pp[0]] * c + lane] : 0.f;
    const float fb = (lane < c) ? f[index[b + pp[1]] * c + lane] : 0.f;
    const float d2 = wave_sum((fa - fb) * (fa - fb));
    posv = (flags & GL_SQRT) ? sqrtf(d2 + 1e-7f) : d2;
  } else {
    float acc = 0.f;
    for (long long j = b; j < e; ++j) {
      const float d = (lane < c) ? g.mean - f[index[j] * c + lane] : 0.f;
      if (flags & GL_SQRT) acc += sqrtf(wave_sum(d * d) + 1e-7f);
      else acc += d * d;
    }
    posv = ((flags & GL_SQRT) ? acc : wave_sum(acc)) * g.inv_n;
  }
  const float dt = (lane < c) ? ((flags & GL_BLOCK) ? g.mean_b : g.mean) - g.ft : 0.f;
  const float d2 = wave_sum(dt * dt);
  finv = (flags & (GL_SQRT | GL_BLOCK)) ? sqrtf(d2 + 1e-7f) : d2;
}

__global__ void __launch_bounds__(64) k_group_loss_fwd(const float* __restrict__ f, int c,
                                                       const long long* __restrict__ index,
                                                       const long long* __restrict__ goff,
                                                       const unsigned char* __restrict__ flag,
                                                       const long long* __restrict__ sel, float pos_thresh,
                                                       float finest_thresh, int flags, const int* __restrict__ pairpos,
                                                       float* pos, float* fin) {
  const int lane = threadIdx.x;
  const long long gi = sel[blockIdx.x];
  const long long b = goff[gi], e = goff[gi + 1];
  const GroupStats g = group_stats(f, c, index, flag, b, e, lane);
  float posv, finv;
  group_terms(f, c, index, b, e, lane, g, flags, pairpos ? pairpos + 2 * blockIdx.x : nullptr, posv, finv);
  if (lane == 0) {
    pos[blockIdx.x] = relu_keep_nan(posv - pos_thresh);
    fin[blockIdx.x] = (flags & GL_NOFIN) ? 0.f : relu_keep_nan(finv - finest_thresh);
  }
}

__global__ void __launch_bounds__(64) k_group_loss_bwd(const float* __restrict__ f, int c,
                                                       const long long* __restrict__ index,
                                                       const long long* __restrict__ goff,
                                                       const unsigned char* __restrict__ flag,
                                                       const long long* __restrict__ sel, float pos_thresh,
                                                       float finest_thresh, int flags, const int* __restrict__ pairpos,
                                                       const float* __restrict__ gpos,
                                                       const float* __restrict__ gfin, float* df) {
  const int lane = threadIdx.x;
  const long long gi = sel[blockIdx.x];
  const long long b = goff[gi], e = goff[gi + 1];
  const GroupStats g = group_stats(f, c, index, flag, b, e, lane);
  const int* pp = pairpos ? pairpos + 2 * blockIdx.x : nullptr;
  float posv, finv;
  group_terms(f, c, index, b, e, lane, g, flags, pp, posv, finv);
  const float gp = (posv - pos_thresh > 0.f) ? gpos[blockIdx.x] : 0.f;
  const float gf = (!(flags & GL_NOFIN) && finv - finest_thresh > 0.f) ? gfin[blockIdx.x] : 0.f;
  if (gp == 0.f && gf == 0.f) return;            // wave-uniform
  const bool sq = !(flags & GL_SQRT);

  // finest term: d/df_j of |mu - f_t|^2 (or its sqrt), mu = mean over all / over the non-finest members
  const float mu = (flags & GL_BLOCK) ? g.mean_b : g.mean;
  const float dt = (lane < c) ? mu - g.ft : 0.f;
  const float fin_scale = gf * ((flags & (GL_SQRT | GL_BLOCK)) ? 1.f / finv : 2.f);    // dL/d(d2) * 2, times dt below
  const float fin_all = fin_scale * ((flags & GL_BLOCK) ? g.inv_nb : g.inv_n) * dt;    // via the mean

  // positive term, non-pair sqrt form: u = (1/n) sum_j (m - f_j) / s_j
  float u = 0.f;
  if (!(flags & GL_PAIR) && !sq && gp != 0.f) {
    for (long long j = b; j < e; ++j) {
      const float d = (lane < c) ? g.mean - f[index[j] * c + lane] : 0.f;
      u += d / sqrtf(wave_sum(d * d) + 1e-7f);
    }
    u *= g.inv_n;
  }
  for (long long j = b; j < e; ++j) {
    const long long row = index[j];
    const float fj = (lane < c) ? f[row * c + lane] : 0.f;
    float gr = 0.f;
    if (!(flags & GL_PAIR) && gp != 0.f) {
      if (sq) gr += gp * 2.f * g.inv_n * (fj - g.mean);
      else {
        const float d = (lane < c) ? g.mean - fj : 0.f;
        const float sj = sqrtf(wave_sum(d * d) + 1e-7f);      // all lanes take part in the reduction
        gr += gp * g.inv_n * (u - d / sj);
      }
    }
    if (gf != 0.f) {
      if (flags & GL_BLOCK) {
        if (!flag[j]) gr += fin_all;                          // the finest member is detached (:480-481)
      } else {
        gr += fin_all;
        if (j == g.tpos) gr -= fin_scale * dt;
      }
    }
    if (lane < c && gr != 0.f) atomicAdd(&df[row * c + lane], gr);
  }
  if ((flags & GL_PAIR) && gp != 0.f && lane < c) {
    const long long ra = index[b + pp[0]], rb = index[b + pp[1]];
    const float diff = f[ra * c + lane] - f[rb * c + lane];
    const float coef = gp * (sq ? 2.f : 1.f / posv);
    atomicAdd(&df[ra * c + lane], coef * diff);
    atomicAdd(&df[rb * c + lane], -coef * diff);
  }
}

// ---- location_circle_loss (lib/colocation_trainer.py:538-681): per-group terms, one wave per selected group ----
//   term(v) = softplus(logsumexp_i(S v_i max(v_i, 0).detach())) / S,  S = log_scale = 16
//   pos:    v_i = dist(mean, f_i) - pos_thresh / 2 over all members                       (:607-618)
//           (GL_PAIR: softplus(dist(f_a, f_b) - pos_thresh), :597-605)
//   finest: v_i = dist(f_i, f_t) - finest_thresh over all members, or over the non-finest members with f_t detached
//           (GL_BLOCK)                                                                     (:620-640)
//   dist = squared distance, or sqrt(. + 1e-7) with GL_SQRT.  Also writes the group's mean feature (:585) for the
//   negative term, which the host forms on the [M, M] matrices of group means.
__device__ __forceinline__ float softplus_f(float x) { return x > 20.f ? x : log1pf(expf(x)); }   // torch threshold 20
__device__ __forceinline__ float sigmoid_f(float x) { return 1.f / (1.f + expf(-x)); }

// v of member j for the positive (WHICH = 0) or finest (WHICH = 1) term; `use` = whether the member takes part
__device__ __forceinline__ float circle_v(int which, const float* __restrict__ f, int c, long long row, int lane,
                                          const GroupStats& g, int flags, float thresh, bool is_finest, bool& use) {
  const float fj = (lane < c) ? f[row * c + lane] : 0.f;
  const float d = (lane < c) ? (which == 0 ? g.mean - fj : fj - g.ft) : 0.f;
  const float d2 = wave_sum(d * d);
  use = which == 0 || !((flags & GL_BLOCK) && is_finest);
  return ((flags & GL_SQRT) ? sqrtf(d2 + 1e-7f) : d2) - thresh;
}

// logsumexp over the members of S v max(v, 0) (empty set: -inf, like torch)
__device__ __forceinline__ float circle_lse(int which, const float* __restrict__ f, int c,
                                            const long long* __restrict__ index,
                                            const unsigned char* __restrict__ flag, long long b, long long e, int lane,
                                            const GroupStats& g, int flags, float thresh, float S) {
  float mx = -INFINITY;
  for (long long j = b; j < e; ++j) {
    bool use;
    const float v = circle_v(which, f, c, index[j], lane, g, flags, thresh, flag[j] != 0, use);
    if (use) mx = fmaxf(mx, S * v * fmaxf(v, 0.f));
  }
  if (mx == -INFINITY) return -INFINITY;
  float s = 0.f;
  for (long long j = b; j < e; ++j) {
    bool use;
    const float v = circle_v(which, f, c, index[j], lane, g, flags, thresh, flag[j] != 0, use);
    if (use) s += expf(S * v * fmaxf(v, 0.f) - mx);
  }
  return mx + logf(s);
}

__global__ void __launch_bounds__(64) k_circle_group_fwd(const float* __restrict__ f, int c,
                                                         const long long* __restrict__ index,
                                                         const long long* __restrict__ goff,
                                                         const unsigned char* __restrict__ flag,
                                                         const long long* __restrict__ sel, float pos_thresh,
                                                         float finest_thresh, float S, int flags,
                                                         const int* __restrict__ pairpos, float* pos, float* fin,
                                                         float* mean_out) {
  const int lane = threadIdx.x;
  const long long gi = sel[blockIdx.x];
  const long long b = goff[gi], e = goff[gi + 1];
  const GroupStats g = group_stats(f, c, index, flag, b, e, lane);
  if (lane < c) mean_out[(long long)blockIdx.x * c + lane] = g.mean;
  float pv;
  if (flags & GL_PAIR) {
    const int* pp = pairpos + 2 * blockIdx.x;
    const float fa = (lane < c) ? f[index[b + pp[0]] * c + lane] : 0.f;
    const float fb = (lane < c) ? f[index[b + pp[1]] * c + lane] : 0.f;
    const float d2 = wave_sum((fa - fb) * (fa - fb));
    pv = softplus_f(((flags & GL_SQRT) ? sqrtf(d2 + 1e-7f) : d2) - pos_thresh);
  } else {
    pv = softplus_f(circle_lse(0, f, c, index, flag, b, e, lane, g, flags, 0.5f * pos_thresh, S)) / S;
  }
  const float fv = softplus_f(circle_lse(1, f, c, index, flag, b, e, lane, g, flags, finest_thresh, S)) / S;
  if (lane == 0) {
    pos[blockIdx.x] = pv;
    fin[blockIdx.x] = fv;
  }
}

// dF += gpos dpos/dF + gfin dfin/dF + (1/n) gmean (gradient arriving at the group's mean feature), float atomics
__global__ void __launch_bounds__(64) k_circle_group_bwd(const float* __restrict__ f, int c,
                                                         const long long* __restrict__ index,
                                                         const long long* __restrict__ goff,
                                                         const unsigned char* __restrict__ flag,
                                                         const long long* __restrict__ sel, float pos_thresh,
                                                         float finest_thresh, float S, int flags,
                                                         const int* __restrict__ pairpos,
                                                         const float* __restrict__ gpos, const float* __restrict__ gfin,
                                                         const float* __restrict__ gmean, float* df) {
  const int lane = threadIdx.x;
  const long long gi = sel[blockIdx.x];
  const long long b = goff[gi], e = goff[gi + 1];
  const GroupStats g = group_stats(f, c, index, flag, b, e, lane);
  const float gp = gpos[blockIdx.x], gf = gfin[blockIdx.x];
  const float gm = (gmean && lane < c) ? gmean[(long long)blockIdx.x * c + lane] * g.inv_n : 0.f;
  const bool sq = !(flags & GL_SQRT);
  // positive term: d/dv_i = sigmoid(lse) softmax_i max(v_i, 0); dv_i/df_j = k_i (m - f_i)(1/n - [i == j]),
  // k_i = 2 (squared) or 1 / (v_i + thresh) (sqrt)
  float lse_p = 0.f, sig_p = 0.f, u = 0.f;
  if (!(flags & GL_PAIR) && gp != 0.f) {
    lse_p = circle_lse(0, f, c, index, flag, b, e, lane, g, flags, 0.5f * pos_thresh, S);
    sig_p = sigmoid_f(lse_p);
    for (long long j = b; j < e; ++j) {
      bool use;
      const float v = circle_v(0, f, c, index[j], lane, g, flags, 0.5f * pos_thresh, false, use);
      const float wgt = sig_p * expf(S * v * fmaxf(v, 0.f) - lse_p) * fmaxf(v, 0.f);
      const float kk = sq ? 2.f : 1.f / (v + 0.5f * pos_thresh);
      const float d = (lane < c) ? g.mean - f[index[j] * c + lane] : 0.f;
      u += wgt * kk * d;
    }
    u *= g.inv_n;
  }
  const float lse_f = (gf != 0.f) ? circle_lse(1, f, c, index, flag, b, e, lane, g, flags, finest_thresh, S) : 0.f;
  const float sig_f = sigmoid_f(lse_f);
  float gt = 0.f;                                        // gradient collected for the finest member
  for (long long j = b; j < e; ++j) {
    const long long row = index[j];
    const float fj = (lane < c) ? f[row * c + lane] : 0.f;
    float gr = gm;
    if (!(flags & GL_PAIR) && gp != 0.f) {
      bool use;
      const float v = circle_v(0, f, c, row, lane, g, flags, 0.5f * pos_thresh, false, use);
      const float wgt = sig_p * expf(S * v * fmaxf(v, 0.f) - lse_p) * fmaxf(v, 0.f);
      const float kk = sq ? 2.f : 1.f / (v + 0.5f * pos_thresh);
      gr += gp * (u - wgt * kk * (g.mean - fj));
    }
    if (gf != 0.f && lse_f != -INFINITY) {
      bool use;
      const float v = circle_v(1, f, c, row, lane, g, flags, finest_thresh, flag[j] != 0, use);
      if (use) {
        const float wgt = sig_f * expf(S * v * fmaxf(v, 0.f) - lse_f) * fmaxf(v, 0.f);
        const float kk = sq ? 2.f : 1.f / (v + finest_thresh);
        const float dfi = gf * wgt * kk * (fj - g.ft);
        gr += dfi;
        if (!(flags & GL_BLOCK)) gt -= dfi;
      }
    }
    if (lane < c && gr != 0.f) atomicAdd(&df[row * c + lane], gr);
  }
  if (lane < c && gt != 0.f) atomicAdd(&df[index[g.tpos] * c + lane], gt);
  if ((flags & GL_PAIR) && gp != 0.f && lane < c) {
    const int* pp = pairpos + 2 * blockIdx.x;
    const long long ra = index[b + pp[0]], rb = index[b + pp[1]];
    const float diff = f[ra * c + lane] - f[rb * c + lane];
    const float d2 = wave_sum(diff * diff);
    const float dist = sq ? d2 : sqrtf(d2 + 1e-7f);
    const float coef = gp * sigmoid_f(dist - pos_thresh) * (sq ? 2.f : 1.f / dist);
    atomicAdd(&df[ra * c + lane], coef * diff);
    atomicAdd(&df[rb * c + lane], -coef * diff);
  }
}

// ---- pairwise squared distance + row minimum ----------------------------------------------------------
// Round 5 form.  A thread owns ONE A row, held as (a_c, a_c) register pairs; the B rows come two at a time from a copy
// of B[rows_b] interleaved as [row pair][channel][2] (k_nn_interleave), read with SCALAR loads (the address is the same
// for the whole wave), so that one v_pk_add_f32 / v_pk_fma_f32 with an SGPR-pair operand serves two B rows and the loop
// touches no LDS: 38 us for 5000 x 5000 x 32 against 66 us of the LDS-broadcast form (tools/micro/nn_variants.hip, which
// also holds the five other forms tried).  The distance keeps the exact (a - b)^2 form of lib/metrics.py:22-25 (no
// |a|^2 + |b|^2 - 2ab) as ONE chain per pair, written out (nn_chain8) so that the compiler has no choice in it:
// d2 = (a_0 - b_0)^2 rounded, then d2 = fma(a_c - b_c, a_c - b_c, d2) for c = 1 .. C - 1.
// Grid = (64-row A tiles) x (B chunks, an even number of rows each); the four waves of a workgroup take every fourth row
// pair of the chunk.  A chunk's (minimum, lowest index) goes to part_v / part_i [chunk][row]; k_nn_merge folds the chunks
// in ascending order (strict <: the lowest index wins ties, as torch.min over the reference's distance matrix does).
constexpr int NN_TA = 64;    // A rows per workgroup
constexpr int NN_GRAN = 8;   // B rows per chunk: a multiple of (4 waves x 2 rows)
typedef float nn_f2 __attribute__((ext_vector_type(2)));

// eight channels of the chain  acc = fma(a_c - b_c, a_c - b_c, acc)  for two B rows at once; every dependent pair of packed
// operations has one independent instruction between them (the packed-fp32 forwarding hazard of gfx950 needs one wait
// state: the compiler writes s_nop 0 there, and does not look inside an asm block).
#define NN_SUB(D, A, B) "v_pk_add_f32 " D ", " A ", " B " neg_lo:[0,1] neg_hi:[0,1]\n"
#define NN_FMA(D) "v_pk_fma_f32 %[acc], " D ", " D ", %[acc]\n"
template <bool FIRST>
__device__ __forceinline__ void nn_chain8(nn_f2& acc, const nn_f2* a, const nn_f2* b) {
  nn_f2 d0, d1;
  if (FIRST)
    asm volatile(NN_SUB("%[d0]", "%[a0]", "%[b0]") NN_SUB("%[d1]", "%[a1]", "%[b1]")
                 "v_pk_mul_f32 %[acc], %[d0], %[d0]\n"
                 NN_SUB("%[d0]", "%[a2]", "%[b2]") NN_FMA("%[d1]") NN_SUB("%[d1]", "%[a3]", "%[b3]") NN_FMA("%[d0]")
                 NN_SUB("%[d0]", "%[a4]", "%[b4]") NN_FMA("%[d1]") NN_SUB("%[d1]", "%[a5]", "%[b5]") NN_FMA("%[d0]")
                 NN_SUB("%[d0]", "%[a6]", "%[b6]") NN_FMA("%[d1]") NN_SUB("%[d1]", "%[a7]", "%[b7]") NN_FMA("%[d0]")
                 "s_nop 0\n" NN_FMA("%[d1]")
                 : [acc] "=&v"(acc), [d0] "=&v"(d0), [d1] "=&v"(d1)
                 : [a0] "v"(a[0]), [a1] "v"(a[1]), [a2] "v"(a[2]), [a3] "v"(a[3]), [a4] "v"(a[4]), [a5] "v"(a[5]),
                   [a6] "v"(a[6]), [a7] "v"(a[7]), [b0] "s"(b[0]), [b1] "s"(b[1]), [b2] "s"(b[2]), [b3] "s"(b[3]),
                   [b4] "s"(b[4]), [b5] "s"(b[5]), [b6] "s"(b[6]), [b7] "s"(b[7]));
  else
    asm volatile(NN_SUB("%[d0]", "%[a0]", "%[b0]") NN_SUB("%[d1]", "%[a1]", "%[b1]") NN_FMA("%[d0]")
                 NN_SUB("%[d0]", "%[a2]", "%[b2]") NN_FMA("%[d1]") NN_SUB("%[d1]", "%[a3]", "%[b3]") NN_FMA("%[d0]")
                 NN_SUB("%[d0]", "%[a4]", "%[b4]") NN_FMA("%[d1]") NN_SUB("%[d1]", "%[a5]", "%[b5]") NN_FMA("%[d0]")
                 NN_SUB("%[d0]", "%[a6]", "%[b6]") NN_FMA("%[d1]") NN_SUB("%[d1]", "%[a7]", "%[b7]") NN_FMA("%[d0]")
                 "s_nop 0\n" NN_FMA("%[d1]")
                 : [acc] "+v"(acc), [d0] "=&v"(d0), [d1] "=&v"(d1)
                 : [a0] "v"(a[0]), [a1] "v"(a[1]), [a2] "v"(a[2]), [a3] "v"(a[3]), [a4] "v"(a[4]), [a5] "v"(a[5]),
                   [a6] "v"(a[6]), [a7] "v"(a[7]), [b0] "s"(b[0]), [b1] "s"(b[1]), [b2] "s"(b[2]), [b3] "s"(b[3]),
                   [b4] "s"(b[4]), [b5] "s"(b[5]), [b6] "s"(b[6]), [b7] "s"(b[7]));
}
template <int C>
__global__ void __launch_bounds__(256) k_nn_interleave(const float* __restrict__ b, const long long* __restrict__ rows_b,
                                                       int mb, float* __restrict__ bi) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;       // one (row pair, channel)
  const int p = (int)(e / C), c = (int)(e % C);
  if (2 * p >= mb) return;
  const long long r0 = rows_b ? rows_b[2 * p] : (long long)(2 * p);
  float v1 = 0.f;                                                        // an odd count's last partner: never compared
  if (2 * p + 1 < mb) v1 = b[(rows_b ? rows_b[2 * p + 1] : (long long)(2 * p + 1)) * C + c];
  *reinterpret_cast<nn_f2*>(bi + 2 * e) = nn_f2{b[r0 * C + c], v1};
}

template <int C>
__global__ void __launch_bounds__(256) k_nn_rowmin(const float* __restrict__ a, const long long* __restrict__ rows_a,
                                                   int ma, const float* __restrict__ bi_, int mb, int chunk, int l2,
                                                   float* __restrict__ out_v, int* __restrict__ out_i) {
  __shared__ float rv[4][NN_TA];
  __shared__ int ri[4][NN_TA];
  const nn_f2* __restrict__ bi = reinterpret_cast<const nn_f2*>(bi_);
  const int t = threadIdx.x, ar = t & 63, cg = __builtin_amdgcn_readfirstlane(t >> 6);
  const int arow = blockIdx.x * NN_TA + ar;
  nn_f2 av[C];
  {
    long long src = (arow < ma) ? (rows_a ? rows_a[arow] : (long long)arow) : -1;
#pragma unroll
    for (int q = 0; q < C / 4; ++q) {
      float4 v = make_float4(0, 0, 0, 0);
      if (src >= 0) v = reinterpret_cast<const float4*>(a + src * C)[q];
      av[4 * q] = nn_f2{v.x, v.x}; av[4 * q + 1] = nn_f2{v.y, v.y};
      av[4 * q + 2] = nn_f2{v.z, v.z}; av[4 * q + 3] = nn_f2{v.w, v.w};
    }
  }
  float best = INFINITY;
  int besti = 0;
  const int jb = blockIdx.y * chunk;
  const int je = (jb + chunk < mb) ? jb + chunk : mb;
  for (int p = jb / 2 + cg; 2 * p < je; p += 4) {       // ascending within a thread: strict < keeps the lowest index
    const nn_f2* __restrict__ row = bi + (long long)p * C;
    nn_f2 bv[C];
#pragma unroll
    for (int c = 0; c < C; ++c) bv[c] = row[c];
    nn_f2 acc;
    nn_chain8<true>(acc, av, bv);
#pragma unroll
    for (int c = 8; c < C; c += 8) nn_chain8<false>(acc, av + c, bv + c);
    asm volatile("s_nop 0" : "+v"(acc));      // the last packed write before the compiler's own reads of acc
    if (acc.x < best) {
      best = acc.x;
      besti = 2 * p;
    }
    if (2 * p + 1 < je && acc.y < best) {
      best = acc.y;
      besti = 2 * p + 1;
    }
  }
  rv[cg][ar] = best;
  ri[cg][ar] = besti;
  __syncthreads();
  if (cg == 0 && arow < ma) {
    float bv = rv[0][ar];
    int bix = ri[0][ar];
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      float v = rv[w][ar];
      int i2 = ri[w][ar];
      if (v < bv || (v == bv && i2 < bix)) {
        bv = v;
        bix = i2;
      }
    }
    if (gridDim.y == 1) {
      out_v[arow] = l2 ? sqrtf(bv + 1e-7f) : bv;
      out_i[arow] = bix;
    } else {
      out_v[(long long)blockIdx.y * ma + arow] = bv;
      out_i[(long long)blockIdx.y * ma + arow] = bix;
    }
  }
}

__global__ void __launch_bounds__(256) k_nn_merge(const float* __restrict__ part_v, const int* __restrict__ part_i, int ma,
                                                  int n_chunks, int l2, float* __restrict__ dmin, int* __restrict__ argmin) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= ma) return;
  float bv = part_v[r];
  int bi = part_i[r];
  for (int c = 1; c < n_chunks; ++c) {      // chunks hold ascending index ranges: strict < keeps the lowest index
    const float v = part_v[(long long)c * ma + r];
    if (v < bv) {
      bv = v;
      bi = part_i[(long long)c * ma + r];
    }
  }
  dmin[r] = l2 ? sqrtf(bv + 1e-7f) : bv;
  argmin[r] = bi;
}

// B rows per chunk: as many chunks as it takes to put ~1536 workgroups on the chip, each a multiple of NN_GRAN rows
static int nn_chunk_rows(int ma, int mb) {
  const long long a_tiles = cdiv(ma, NN_TA);
  long long want = cdiv(1536, a_tiles);                 // chunks wanted
  const long long grans = cdiv(mb, NN_GRAN);
  if (want > grans) want = grans;
  if (want < 1) want = 1;
  return (int)(cdiv(grans, want) * NN_GRAN);
}
// scratch (32-bit words): the interleaved copy of B, then the chunks' partial (minimum, index) pairs
static long long nn_interleaved_words(int mb, int c) { return (long long)cdiv(mb, 2) * 2 * c; }

// ---- negative-pair mask ---------------------------------------------------------------------------------
__global__ void k_table_fill2(Slot* t, long long cap) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < cap) {
    t[i].key = EMPTY_KEY;
    t[i].val = LLONG_MAX;
  }
}

__global__ void k_neg_mask_init(const long long* __restrict__ sel1, const long long* __restrict__ sel2,
                                const int* __restrict__ arg, int m, Slot* t, long long cap, unsigned char* keep) {
  int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= m) return;
  long long a = sel1[r], bb = sel2[arg[r]];
  keep[r] = (a != bb) ? 1 : 0;
  long long s = table_insert(t, cap, (unsigned long long)a);
  atomicMin(&t[s].val, (long long)r);   // sel1 is drawn without replacement; duplicates would keep the lowest r
}

__global__ void k_neg_mask_scan(const long long* __restrict__ sel1, const long long* __restrict__ sel2,
                                const int* __restrict__ arg, const long long* __restrict__ index,
                                const long long* __restrict__ goff, long long n_groups, long long n_index,
                                const Slot* __restrict__ t, long long cap, unsigned char* keep) {
  long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_index) return;
  long long row = index[e];
  long long s = table_find(t, cap, (unsigned long long)row);
  if (s < 0) return;
  int r = (int)t[s].val;
  long long target = sel2[arg[r]];
  // group of entry e: largest g with goff[g] <= e
  long long lo = 0, hi = n_groups;
  while (hi - lo > 1) {
    long long mid = (lo + hi) >> 1;
    if (goff[mid] <= e) lo = mid; else hi = mid;
  }
  for (long long j = goff[lo]; j < goff[lo + 1]; ++j)
    if (index[j] == target && target != row) {
      keep[r] = 0;
      break;
    }
}

__global__ void __launch_bounds__(256) k_neg_loss_fwd(const float* __restrict__ dmin,
                                                      const unsigned char* __restrict__ keep, int m, float thresh,
                                                      float* out) {
  __shared__ float ss[256];
  __shared__ int sc[256];
  float s = 0.f;
  int cnt = 0;
  for (int r = threadIdx.x; r < m; r += 256)
    if (keep[r]) {
      float v = fmaxf(thresh - dmin[r], 0.f);
      s += v * v;
      ++cnt;
    }
  ss[threadIdx.x] = s;
  sc[threadIdx.x] = cnt;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) {
      ss[threadIdx.x] += ss[threadIdx.x + o];
      sc[threadIdx.x] += sc[threadIdx.x + o];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out[0] = ss[0] / (float)sc[0];   // 0/0 = NaN, like torch's mean of an empty tensor
    out[1] = (float)sc[0];
  }
}

__global__ void __launch_bounds__(64) k_neg_loss_bwd(const float* __restrict__ f, int c,
                                                     const long long* __restrict__ sel1,
                                                     const long long* __restrict__ sel2,
                                                     const int* __restrict__ arg, const float* __restrict__ dmin,
                                                     const unsigned char* __restrict__ keep, float thresh,
                                                     const float* __restrict__ out, const float* __restrict__ gneg,
                                                     float* df) {
  const int r = blockIdx.x, lane = threadIdx.x;
  if (!keep[r] || lane >= c) return;
  float D = dmin[r];
  if (!(thresh - D > 0.f)) return;
  long long ra = sel1[r], rb = sel2[arg[r]];
  float coef = gneg[0] * (-2.f * (thresh - D) / out[1]) / D;   // dL/dD * dD/d(d2) * 2 == ... * (a - b) / D
  float diff = f[ra * c + lane] - f[rb * c + lane];
  atomicAdd(&df[ra * c + lane], coef * diff);
  atomicAdd(&df[rb * c + lane], -coef * diff);
}


// The step's scalar arithmetic in one launch: pos_mean = sum(pos) / n_sel, fin_mean = sum(fin) / n_sel (lib/colocation_
// trainer.py:533-535), total = w_pos * pos_mean + w_fin * fin_mean + w_neg * neg (:865-868).  One workgroup, fp64 thread
// sums in a fixed tree (deterministic).  out = {total, pos_mean, fin_mean, neg}.
__global__ void __launch_bounds__(256) k_loss_combine(const float* __restrict__ pos, const float* __restrict__ fin, int n_sel,
                                                      const float* __restrict__ neg, float w_pos, float w_fin, float w_neg,
                                                      float* __restrict__ out) {
  __shared__ double red[2][256];
  const int t = threadIdx.x;
  double a = 0, b = 0;
  for (int i = t; i < n_sel; i += 256) {
    a += (double)pos[i];
    b += (double)fin[i];
  }
  red[0][t] = a;
  red[1][t] = b;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (t < o) {
      red[0][t] += red[0][t + o];
      red[1][t] += red[1][t + o];
    }
    __syncthreads();
  }
  if (t == 0) {
    const float pm = (float)red[0][0] / (float)n_sel, fm = (float)red[1][0] / (float)n_sel, ng = neg[0];
    out[0] = (w_pos * pm + w_fin * fm) + w_neg * ng;
    out[1] = pm;
    out[2] = fm;
    out[3] = ng;
  }
}
// ... and its backward: the upstream gradients of the three terms from the gradient of the total
__global__ void __launch_bounds__(256) k_loss_seed(const float* __restrict__ g_total, float w_pos, float w_fin, float w_neg,
                                                   int n_sel, float* __restrict__ gpos, float* __restrict__ gfin,
                                                   float* __restrict__ gneg) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const float g = g_total[0];
  if (i < n_sel) {
    gpos[i] = (g * w_pos) / (float)n_sel;
    gfin[i] = (g * w_fin) / (float)n_sel;
  }
  if (i == 0) gneg[0] = g * w_neg;
}

}  // namespace gcl

using namespace gcl;

extern "C" {

int gcl_group_loss_fwd(const float* f, int32_t c, const int64_t* index, const int64_t* goff,
                       const uint8_t* finest_flag, const int64_t* sel, int32_t n_sel, float pos_thresh,
                       float finest_thresh, int32_t flags, const int32_t* pairpos, float* pos, float* fin,
                       void* stream) {
  GCL_CHECK_ARG(f && index && goff && finest_flag && sel && pos && fin, "gcl_group_loss_fwd: null pointer");
  GCL_CHECK_ARG(flags >= 0 && flags < 16 && (!(flags & GL_PAIR) || pairpos),
                "gcl_group_loss_fwd: bad flags / pair positions missing");
  GCL_CHECK_ARG(c >= 1 && c <= 64, "gcl_group_loss_fwd: feature width must be <= 64 (got %d)", c);
  if (n_sel <= 0) return GCL_OK;
  hipLaunchKernelGGL(k_group_loss_fwd, dim3(n_sel), dim3(64), 0, (hipStream_t)stream, f, c, (const long long*)index,
                     (const long long*)goff, finest_flag, (const long long*)sel, pos_thresh, finest_thresh, flags,
                     (flags & GL_PAIR) ? pairpos : nullptr, pos, fin);
  GCL_CHECK_LAUNCH();
  return GCL_OK;
}

int gcl_group_loss_bwd(const float* f, int32_t c, const int64_t* index, const int64_t* goff,
                       const uint8_t* finest_flag, const int64_t* sel, int32_t n_sel, float pos_thresh,
                       float finest_thresh, int32_t flags, const int32_t* pairpos, const float* gpos,
                       const float* gfin, float* df, void* stream) {
  GCL_CHECK_ARG(f && index && goff && finest_flag && sel && gpos && gfin && df, "gcl_group_loss_bwd: null pointer");
  GCL_CHECK_ARG(flags >= 0 && flags < 16 && (!(flags & GL_PAIR) || pairpos),
                "gcl_group_loss_bwd: bad flags / pair positions missing");
  GCL_CHECK_ARG(c >= 1 && c <= 64, "gcl_group_loss_bwd: feature width must be <= 64 (got %d)", c);
  if (n_sel <= 0) return GCL_OK;
  hipLaunchKernelGGL(k_group_loss_bwd, dim3(n_sel), dim3(64), 0, (hipStream_t)stream, f, c, (const long long*)index,
                     (const long long*)goff, finest_flag, (const long long*)sel, pos_thresh, finest_thresh, flags,
                     (flags & GL_PAIR) ? pairpos : nullptr, gpos, gfin, df);
  GCL_CHECK_LAUNCH();
  return GCL_OK;
}

int gcl_circle_group_fwd(const float* f, int32_t c, const int64_t* index, const int64_t* goff,
                         const uint8_t* finest_flag, const int64_t* sel, int32_t n_sel, float pos_thresh,
                         float finest_thresh, float log_scale, int32_t flags, const int32_t* pairpos, float* pos,
                         float* fin, float* mean_out, void* stream) {
  GCL_CHECK_ARG(f && index && goff && finest_flag && sel && pos && fin && mean_out, "gcl_circle_group_fwd: null pointer");
  GCL_CHECK_ARG(c >= 1 && c <= 64, "gcl_circle_group_fwd: feature width must be <= 64 (got %d)", c);
  GCL_CHECK_ARG(flags >= 0 && flags < 8 && (!(flags & GL_PAIR) || pairpos) && log_scale > 0,
                "gcl_circle_group_fwd: bad flags / pair positions missing / log_scale");
  if (n_sel <= 0) return GCL_OK;
  hipLaunchKernelGGL(k_circle_group_fwd, dim3(n_sel), dim3(64), 0, (hipStream_t)stream, f, c, (const long long*)index,
                     (const long long*)goff, finest_flag, (const long long*)sel, pos_thresh, finest_thresh, log_scale,
                     flags, pairpos, pos, fin, mean_out);
  GCL_CHECK_LAUNCH();
  return GCL_OK;
}

int gcl_circle_group_bwd(const float* f, int32_t c, const int64_t* index, const int64_t* goff,
                         const uint8_t* finest_flag, const int64_t* sel, int32_t n_sel, float pos_thresh,
                         float finest_thresh, float log_scale, int32_t flags, const int32_t* pairpos,
                         const float* gpos, const float* gfin, const float* gmean, float* df, void* stream) {
  GCL_CHECK_ARG(f && index && goff && finest_flag && sel && gpos && gfin && df, "gcl_circle_group_bwd: null pointer");
  GCL_CHECK_ARG(c >= 1 && c <= 64, "gcl_circle_group_bwd: feature width must be <= 64 (got %d)", c);
  GCL_CHECK_ARG(flags >= 0 && flags < 8 && (!(flags & GL_PAIR) || pairpos) && log_scale > 0,
                "gcl_circle_group_bwd: bad flags / pair positions missing / log_scale");
  if (n_sel <= 0) return GCL_OK;
  hipLaunchKernelGGL(k_circle_group_bwd, dim3(n_sel), dim3(64), 0, (hipStream_t)stream, f, c, (const long long*)index,
                     (const long long*)goff, finest_flag, (const long long*)sel, pos_thresh, finest_thresh, log_scale,
                     flags, pairpos, gpos, gfin, gmean, df);
  GCL_CHECK_LAUNCH();
  return GCL_OK;
}

int64_t gcl_nn_rowmin_scratch_len(int32_t ma, int32_t mb) {
  if (ma <= 0 || mb <= 0) return 0;
  const int chunk = nn_chunk_rows(ma, mb);
  const long long n_chunks = cdiv(mb, chunk);
  return nn_interleaved_words(mb, 64) + (n_chunks > 1 ? 2 * n_chunks * (long long)ma : 0);   // sized for the widest rows
}

int gcl_nn_rowmin(const float* a, const int64_t* rows_a, int32_t ma, const float* b, const int64_t* rows_b,
                  int32_t mb, int32_t c, int32_t l2, int32_t* scratch, float* dmin, int32_t* argmin, void* stream) {
  GCL_CHECK_ARG(a && b && dmin && argmin, "gcl_nn_rowmin: null pointer");
  GCL_CHECK_ARG(ma > 0 && mb > 0, "gcl_nn_rowmin: empty input");
  GCL_CHECK_ARG(c == 16 || c == 32 || c == 64, "gcl_nn_rowmin: feature width must be 16, 32 or 64 (got %d)", c);
  GCL_CHECK_ARG(scratch, "gcl_nn_rowmin: scratch (int32[gcl_nn_rowmin_scratch_len]) is required");
  hipStream_t st = (hipStream_t)stream;
  const int chunk = nn_chunk_rows(ma, mb);
  const int n_chunks = (int)cdiv(mb, chunk);
  dim3 grid((unsigned)cdiv(ma, NN_TA), (unsigned)n_chunks);
  float* bi = (float*)scratch;
  int32_t* part = scratch + nn_interleaved_words(mb, 64);
  float* pv = n_chunks > 1 ? (float*)part : dmin;
  int* pi = n_chunks > 1 ? part + (long long)n_chunks * ma : argmin;
  const unsigned il_grid = (unsigned)cdiv(cdiv(mb, 2) * c, 256);
#define LAUNCH_NN(CC)                                                                                              \
  do {                                                                                                             \
    hipLaunchKernelGGL(k_nn_interleave<CC>, dim3(il_grid), dim3(256), 0, st, b, (const long long*)rows_b, mb, bi); \
    hipLaunchKernelGGL(k_nn_rowmin<CC>, grid, dim3(256), 0, st, a, (const long long*)rows_a, ma, (const float*)bi, \
                       mb, chunk, l2, pv, pi);                                                                     \
  } while (0)
  if (c == 16) LAUNCH_NN(16);
  else if (c == 32) LAUNCH_NN(32);
  else LAUNCH_NN(64);
#undef LAUNCH_NN
  if (n_chunks > 1)
    hipLaunchKernelGGL(k_nn_merge, dim3((unsigned)cdiv(ma, 256)), dim3(256), 0, st, (const float*)pv, (const int*)pi, ma,
                       n_chunks, l2, dmin, argmin);
  GCL_CHECK_LAUNCH();
  return GCL_OK;
}

int gcl_neg_mask(const int64_t* sel1, const int64_t* sel2, const int32_t* arg, int32_t m, const int64_t* index,
                 const int64_t* goff, int64_t n_groups, int64_t n_index, int64_t* table, int64_t cap, uint8_t* keep,
                 void* stream) {
  GCL_CHECK_ARG(sel1 && sel2 && arg && table && keep, "gcl_neg_mask: null pointer");
  GCL_CHECK_ARG(m > 0 && cap >= 2 * (int64_t)m && (cap & (cap - 1)) == 0, "gcl_neg_mask: cap must be a power of two >= 2m");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_table_fill2, dim3((unsigned)cdiv(cap, 256)), dim3(256), 0, st, (Slot*)table, (long long)cap);
  hipLaunchKernelGGL(k_neg_mask_init, dim3((unsigned)cdiv(m, 256)), dim3(256), 0, st, (const long long*)sel1,
                     (const long long*)sel2, arg, m, (Slot*)table, (long long)cap, keep);
  if (n_index > 0 && n_groups > 0) {
    GCL_CHECK_ARG(index && goff, "gcl_neg_mask: null index / goff");
    hipLaunchKernelGGL(k_neg_mask_scan, dim3((unsigned)cdiv(n_index, 256)), dim3(256), 0, st, (const long long*)sel1,
                       (const long long*)sel2, arg, (const long long*)index, (const long long*)goff,
                       (long long)n_groups, (long long)n_index, (const Slot*)table, (long long)cap, keep);
  }
  GCL_CHECK_LAUNCH();
  return GCL_OK;
}

int gcl_neg_loss_fwd(const float* dmin, const uint8_t* keep, int32_t m, float thresh, float* out, void* stream) {
  GCL_CHECK_ARG(dmin && keep && out && m > 0, "gcl_neg_loss_fwd: bad argument");
  hipLaunchKernelGGL(k_neg_loss_fwd, dim3(1), dim3(256), 0, (hipStream_t)stream, dmin, keep, m, thresh, out);
  GCL_CHECK_LAUNCH();
  return GCL_OK;
}

int gcl_neg_loss_bwd(const float* f, int32_t c, const int64_t* sel1, const int64_t* sel2, const int32_t* arg,
                     const float* dmin, const uint8_t* keep, int32_t m, float thresh, const float* out,
                     const float* gneg, float* df, void* stream) {
  GCL_CHECK_ARG(f && sel1 && sel2 && arg && dmin && keep && out && gneg && df && m > 0, "gcl_neg_loss_bwd: bad argument");
  GCL_CHECK_ARG(c >= 1 && c <= 64, "gcl_neg_loss_bwd: feature width must be <= 64");
  hipLaunchKernelGGL(k_neg_loss_bwd, dim3(m), dim3(64), 0, (hipStream_t)stream, f, c, (const long long*)sel1,
                     (const long long*)sel2, arg, dmin, keep, thresh, out, gneg, df);
  GCL_CHECK_LAUNCH();
  return GCL_OK;
}

int gcl_loss_combine(const float* pos, const float* fin, int32_t n_sel, const float* neg, float w_pos, float w_fin,
                     float w_neg, float* out, void* stream) {
  GCL_CHECK_ARG(pos && fin && neg && out && n_sel > 0, "gcl_loss_combine: bad argument");
  hipLaunchKernelGGL(k_loss_combine, dim3(1), dim3(256), 0, (hipStream_t)stream, pos, fin, n_sel, neg, w_pos, w_fin, w_neg, out);
  GCL_CHECK_LAUNCH();
  return GCL_OK;
}

int gcl_loss_seed(const float* g_total, float w_pos, float w_fin, float w_neg, int32_t n_sel, float* gpos, float* gfin,
                  float* gneg, void* stream) {
  GCL_CHECK_ARG(g_total && gpos && gfin && gneg && n_sel > 0, "gcl_loss_seed: bad argument");
  hipLaunchKernelGGL(k_loss_seed, dim3((unsigned)cdiv(n_sel, 256)), dim3(256), 0, (hipStream_t)stream, g_total, w_pos, w_fin,
                     w_neg, n_sel, gpos, gfin, gneg);
  GCL_CHECK_LAUNCH();
  return GCL_OK;
}

}  // extern "C"
