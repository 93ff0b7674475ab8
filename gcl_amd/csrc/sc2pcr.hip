// SC2-PCR registration back-end on the device: the algorithm of scripts/SC2_PCR/SC2_PCR.py
// (Matcher.SC2_PCR :304-381) for one pair of clouds, restated for gfx950.
//
// The reference materialises four [N, N] float matrices (N <= 8000: 256 MB each) and a [S, N] x [N, N] float matmul
// on 0/1 values; here the compatibility of a correspondence pair is recomputed from the six coordinates wherever it
// is needed (25-64 M pair evaluations are cheaper than one pass over such a matrix), the tight compatibility is kept
// as a BIT matrix (N x N/64 words, 8 MB at N = 8000) and the second-order measure
//   SC2[s][j] = sum_m tight[seed_s][m] tight[m][j] * hard[seed_s][j]                                      (:353-361)
// becomes AND + popcount over 125 words.  Per-seed work (k1 = 30 nearest by SC2, local second-order selection of
// k2 = 20, 20 x 20 power iteration, weighted Kabsch with a 3 x 3 Jacobi SVD) runs as one wavefront per seed.
// Ties (argsort / argmax of equal values, unspecified in the reference) go to the LOWEST index; per-seed power
// iterations always run num_iterations steps (see oracle/sc2pcr_oracle.py).
#include "common.h"

#include <math.h>
#include <stdlib.h>

#include <algorithm>

namespace gcl {

constexpr int SC_TILE = 256;      // correspondences per LDS tile
constexpr int SC_CHUNKS = 8;      // column chunks of the matvec (partials added in fixed order)
constexpr int SC_MAXN = 8192;     // correspondences per registration (max_points = 8000 in the reference's configs)

struct P3 { float x, y, z; };
__device__ __forceinline__ float dist3(const P3& a, const P3& b) {
  float dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z;
  // torch.norm(a - b): no epsilon.  The sum of squares is written out as ONE chain: left to the compiler, kernels that compare
  // the same length difference against a threshold (k_sc_tight_bits, k_sc_sparse_build, k_sc_seed_knn) got different
  // contractions (packed multiplies + one fma + add in one, multiply + two fmas in another) and disagreed on borderline pairs.
  return sqrtf(__builtin_fmaf(dz, dz, __builtin_fmaf(dx, dx, dy * dy)));
}
__device__ __forceinline__ P3 ld3(const float* __restrict__ p, int i) { return P3{p[3 * i], p[3 * i + 1], p[3 * i + 2]}; }

// ---- confidence = leading eigenvector of SC (:337, :345, :167-185) -------------------------------------------
// partial[chunk][i] = sum_{j in chunk} clamp(1 - (|s_i s_j| - |t_i t_j|)^2 / d^2, 0) x_j
__global__ void __launch_bounds__(SC_TILE) k_sc_matvec(const float* __restrict__ src, const float* __restrict__ tgt,
                                                       int n, float d2_thre, const float* __restrict__ x,
                                                       const int* __restrict__ done, float* partial) {
  if (*done) return;
  __shared__ float ts[SC_TILE][7];
  const int i = blockIdx.x * SC_TILE + threadIdx.x;
  const bool ok = i < n;
  const P3 si = ok ? ld3(src, i) : P3{0, 0, 0}, ti = ok ? ld3(tgt, i) : P3{0, 0, 0};
  const int per = (n + SC_CHUNKS - 1) / SC_CHUNKS;
  const int j0 = blockIdx.y * per, j1 = min(n, j0 + per);
  float acc = 0.f;
  for (int jb = j0; jb < j1; jb += SC_TILE) {
    __syncthreads();
    const int j = jb + threadIdx.x;
    if (j < j1) {
      ts[threadIdx.x][0] = src[3 * j]; ts[threadIdx.x][1] = src[3 * j + 1]; ts[threadIdx.x][2] = src[3 * j + 2];
      ts[threadIdx.x][3] = tgt[3 * j]; ts[threadIdx.x][4] = tgt[3 * j + 1]; ts[threadIdx.x][5] = tgt[3 * j + 2];
      ts[threadIdx.x][6] = x[j];
    }
    __syncthreads();
    const int m = min(SC_TILE, j1 - jb);
    for (int q = 0; q < m; ++q) {
      const float cd = fabsf(dist3(si, P3{ts[q][0], ts[q][1], ts[q][2]}) - dist3(ti, P3{ts[q][3], ts[q][4], ts[q][5]}));
      acc = __builtin_fmaf(fmaxf(1.f - cd * cd / d2_thre, 0.f), ts[q][6], acc);      // one fma, as the sparse walks write it
    }
  }
  if (ok) partial[(size_t)blockIdx.y * n + i] = acc;
}

// ---- the same matvec over the NON-ZERO entries of the compatibility matrix, kept from one build (round 5) ---------------------
// The power iteration multiplies the same n x n matrix 20 times and k_sc_matvec re-derives every entry each time from six
// coordinates and two correctly rounded square roots (64 M entries at n = 8000: 74 us per product, 1.5 ms per registration).
// Most entries are zero -- a pair of correspondences is compatible only when its two lengths agree to d_thre.  ONE build pass
// keeps the non-zero entries (column, value) of row i in column chunk c (the same eight chunks) in ascending column order, in
// that segment's own fixed place of `per` = chunk-length entries (an ELL layout: n^2 entries of address space, 512 MB at n =
// 8000, of which only the non-zero ones are ever touched), and count[c][i].  A product then walks a segment in that order:
// the same non-zero terms in the same order as k_sc_matvec -- the skipped terms are exact zeros -- so partial[][] is BITWISE
// what k_sc_matvec writes.
struct ScEntry { int j; float m; };
__device__ __forceinline__ float sc_first_order(const P3& si, const P3& ti, const float* ts_row, float d2_thre) {
  const float cd = fabsf(dist3(si, P3{ts_row[0], ts_row[1], ts_row[2]}) - dist3(ti, P3{ts_row[3], ts_row[4], ts_row[5]}));
  return fmaxf(1.f - cd * cd / d2_thre, 0.f);
}
// One WAVE per row: lane = column, 64 columns per trip, the non-zero ones written in column order behind a ballot's prefix
// count -- n x 8 waves (40 000 at n = 5000) instead of n x 8 threads (625 waves on 1024 SIMDs, every one of them a serial
// chain of square roots and divisions): 191 -> 116 us.  A workgroup takes SB_ROWS rows of one column chunk and holds the
// chunk's coordinates in LDS.
// The trips are aligned to 64 columns of the whole row, and with `bits` the same pass also writes the TIGHT compatibility
// bit matrix k_sc_tight_bits makes (bit j of row i = |.| < tight_thr, the same |.| the entry's value comes from): a chunk
// owns the words that START in it, and holds the up to 63 columns beyond its end that such a word needs.
constexpr int SB_ROWS = 16;
constexpr int SB_PER_MAX = SC_MAXN / SC_CHUNKS;
__global__ void __launch_bounds__(256) k_sc_sparse_build(const float* __restrict__ src, const float* __restrict__ tgt, int n,
                                                         float d2_thre, int* count, ScEntry* entries, float tight_thr,
                                                         unsigned long long* bits) {
  __shared__ float cs[6][SB_PER_MAX + 128];
  const int per = (n + SC_CHUNKS - 1) / SC_CHUNKS;
  const int j0 = blockIdx.y * per, j1 = min(n, j0 + per);
  const int g0 = j0 & ~63, g1 = min(n, (j1 + 63) & ~63);       // the columns held: the chunk widened to whole words
  for (int q = threadIdx.x; q < g1 - g0; q += 256) {
    const int j = g0 + q;
    cs[0][q] = src[3 * j]; cs[1][q] = src[3 * j + 1]; cs[2][q] = src[3 * j + 2];
    cs[3][q] = tgt[3 * j]; cs[4][q] = tgt[3 * j + 1]; cs[5][q] = tgt[3 * j + 2];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned long long below = (1ull << lane) - 1ull;
  const int words = (n + 63) >> 6;
  for (int r = 0; r < SB_ROWS / 4; ++r) {
    const int i = blockIdx.x * SB_ROWS + wave * (SB_ROWS / 4) + r;
    if (i >= n) break;
    const P3 si = ld3(src, i), ti = ld3(tgt, i);
    ScEntry* const seg = entries + ((size_t)blockIdx.y * n + i) * per;
    int cnt = 0;
    for (int g = g0; g < j1; g += 64) {
      const int j = g + lane;
      const bool held = j < g1;
      const int q = held ? j - g0 : 0;
      const float cd = fabsf(dist3(si, P3{cs[0][q], cs[1][q], cs[2][q]}) - dist3(ti, P3{cs[3][q], cs[4][q], cs[5][q]}));
      if (bits && g >= j0) {                                   // uniform over the wave: this chunk owns the word
        const unsigned long long w = __ballot(held && cd < tight_thr);
        if (lane == 0) bits[(size_t)i * words + (g >> 6)] = w;
      }
      // The ROUNDED value decides what is kept: the empty asm makes it opaque.  Without it the compiler derives the predicate
      // from intermediates (q < 1 instead of max(1 - q, 0) != 0): an entry whose rounded value is not zero could be dropped,
      // and the sums would differ from the dense kernel's in their last bits (profiles/r05_conv_experiments.txt 52).
      float v = fmaxf(1.f - cd * cd / d2_thre, 0.f);
      asm("" : "+v"(v));
      const bool nz = j >= j0 && j < j1 && v != 0.f;      // (a NaN entry counts as non-zero: it must reach the sum as in the dense loop)
      const unsigned long long mask = __ballot(nz);
      if (nz) seg[cnt + __popcll(mask & below)] = ScEntry{j, v};
      cnt += __popcll(mask);
    }
    if (lane == 0) count[(size_t)blockIdx.y * n + i] = cnt;
  }
}

// FOLDED form (the default): product k does the normalisation of product k - 1 ITSELF -- every workgroup sums the previous
// launch's partials (8 n floats from L2), reduces |y| exactly as k_sc_normalize does (256 threads standing for its 1024: the
// same per-thread sums, the same tree) and keeps x = y / (|y| + 1e-6) in LDS, where the product then gathers it from;
// workgroup (0, 0) also writes x and makes the allclose test against the x before it.  20 + 1 launches per registration
// instead of 40 and no cross-workgroup synchronisation inside a launch (a last-workgroup ticket was tried first: the
// device-scope fences it needs cost 40 us per launch); partials alternate between two buffers.  x is bitwise what the
// two-kernel form writes.  When the test sets `done` the other workgroups of that launch may still write a (never read)
// partial; x then stays at the converged vector as in the reference's `break` (:181).
template <bool FIRST>
__global__ void __launch_bounds__(SC_TILE) k_sc_matvec_folded(int n, float* x, int* done, const float* __restrict__ prev,
                                                              float* __restrict__ partial, const int* __restrict__ count,
                                                              const ScEntry* __restrict__ entries) {
  if (*done) return;
  __shared__ float xs[SC_MAXN];
  __shared__ float red[1024];
  __shared__ int allc;
  const int t = threadIdx.x;
  if (FIRST) {
    for (int i = t; i < n; i += SC_TILE) xs[i] = x[i];
  } else {
    for (int v = t; v < 1024; v += SC_TILE) {
      float ss = 0.f;
      for (int i = v; i < n; i += 1024) {
        float y = 0.f;
        for (int c = 0; c < SC_CHUNKS; ++c) y += prev[(size_t)c * n + i];
        xs[i] = y;
        ss += y * y;
      }
      red[v] = ss;
    }
    if (t == 0) allc = 1;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
      for (int v = t; v < o; v += SC_TILE) red[v] += red[v + o];
      __syncthreads();
    }
    const float inv = 1.f / (sqrtf(red[0]) + 1e-6f);
    const bool writer = blockIdx.x == 0 && blockIdx.y == 0;
    bool close = true;
    for (int i = t; i < n; i += SC_TILE) {
      const float xn = xs[i] * inv;
      xs[i] = xn;
      if (writer) {
        const float xo = x[i];
        close = close && (fabsf(xn - xo) <= 1e-8f + 1e-5f * fabsf(xo));
        x[i] = xn;
      }
    }
    if (writer) {
      if (!close) allc = 0;      // benign race: every writer stores 0
      __syncthreads();
      if (t == 0 && allc) *done = 1;
    }
  }
  __syncthreads();
  const int i = blockIdx.x * SC_TILE + t;
  if (i >= n) return;
  const int per = (n + SC_CHUNKS - 1) / SC_CHUNKS;
  const size_t seg = (size_t)blockIdx.y * n + i;
  const ScEntry* e = entries + seg * per;
  const int cnt = count[seg];
  // four entries per trip as two 16-byte loads (a segment starts on a multiple of 8 bytes x per: 16-byte aligned when per is
  // even, else entry by entry); the terms are added in the segment's order by explicit fmas (left to the compiler, the
  // unrolled body became packed multiplies + adds: not the dense kernel's roundings)
  float acc = 0.f;
  int q = 0;
  if ((per & 1) == 0) {
    const int4* e4 = reinterpret_cast<const int4*>(e);
    for (; q + 4 <= cnt; q += 4) {
      const int4 a = e4[q >> 1], b = e4[(q >> 1) + 1];
      acc = __builtin_fmaf(__int_as_float(a.y), xs[a.x], acc);
      acc = __builtin_fmaf(__int_as_float(a.w), xs[a.z], acc);
      acc = __builtin_fmaf(__int_as_float(b.y), xs[b.x], acc);
      acc = __builtin_fmaf(__int_as_float(b.w), xs[b.z], acc);
    }
  }
  for (; q < cnt; ++q) acc = __builtin_fmaf(e[q].m, xs[e[q].j], acc);
  partial[seg] = acc;
}
__global__ void __launch_bounds__(SC_TILE) k_sc_matvec_sparse(int n, const float* __restrict__ x, const int* __restrict__ done,
                                                              float* partial, const int* __restrict__ count,
                                                              const ScEntry* __restrict__ entries) {
  if (*done) return;
  const int i = blockIdx.x * SC_TILE + threadIdx.x;
  if (i >= n) return;
  const int per = (n + SC_CHUNKS - 1) / SC_CHUNKS;
  const size_t seg = (size_t)blockIdx.y * n + i;
  const ScEntry* e = entries + seg * per;
  const int cnt = count[seg];
  float acc = 0.f;
  for (int q = 0; q < cnt; ++q) acc = __builtin_fmaf(e[q].m, x[e[q].j], acc);
  partial[seg] = acc;
}

// one workgroup: y = sum of the partials, x_new = y / (|y| + 1e-6), done = allclose(x_new, x_old) (:176-181).
// `nt` threads stand for 1024: thread t for the rows of t, t + nt, ... < 1024, each with its own sum (red[1024]).
__device__ void sc_normalize_by(int t, int nt, const float* partial, int n, float* x, int* done, float* red, int* allc) {
  const float* pv = partial;
  for (int v = t; v < 1024; v += nt) {
    float ss = 0.f;
    for (int i = v; i < n; i += 1024) {
      float y = 0.f;
      for (int c = 0; c < SC_CHUNKS; ++c) y += pv[(size_t)c * n + i];
      ss += y * y;
    }
    red[v] = ss;
  }
  if (t == 0) *allc = 1;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    for (int v = t; v < o; v += nt) red[v] += red[v + o];
    __syncthreads();
  }
  const float inv = 1.f / (sqrtf(red[0]) + 1e-6f);
  bool close = true;
  for (int i = t; i < n; i += nt) {
    float y = 0.f;
    for (int c = 0; c < SC_CHUNKS; ++c) y += pv[(size_t)c * n + i];
    const float xn = y * inv, xo = x[i];
    close = close && (fabsf(xn - xo) <= 1e-8f + 1e-5f * fabsf(xo));
    x[i] = xn;
  }
  if (!close) *allc = 0;      // benign race: every writer stores 0
  __syncthreads();
  if (t == 0 && *allc) *done = 1;
}
__global__ void __launch_bounds__(1024) k_sc_normalize(const float* __restrict__ partial, int n, float* x, int* done) {
  if (*done) return;
  __shared__ float red[1024];
  __shared__ int allc;
  sc_normalize_by(threadIdx.x, 1024, partial, n, x, done, red, &allc);
}

// ---- seeds: non-maximum suppression (:32-58) -----------------------------------------------------------------
// is_max[i] = all_j (conf_i >= conf_j  or  |s_i s_j| >= R);   is_max pre-set to 1.
// Round 5: every pair is looked at ONCE: within the radius the one of lower confidence loses, whichever side it is on (equal
// confidences: neither) -- the same flags from half the distance evaluations.  A loser is cleared by a plain store (every
// writer stores 0).  Workgroup (bx, y) takes row tile bx against column tile (bx + y) mod tiles, y = 0 .. tiles / 2: every
// unordered pair of tiles exactly once (y = 0: the columns behind the row only; y = tiles / 2 of an even count: the lower
// half of bx only), one tile pair per workgroup so that the launch balances.
__global__ void __launch_bounds__(SC_TILE) k_sc_local_max(const float* __restrict__ src, const float* __restrict__ conf,
                                                          int n, float radius, int* is_max) {
  __shared__ float ts[SC_TILE][4];
  const int n_tiles = (n + SC_TILE - 1) / SC_TILE;
  const int y = blockIdx.y;
  if (2 * y == n_tiles && (int)blockIdx.x >= y) return;
  const int jt = ((int)blockIdx.x + y) % n_tiles;
  const int jb = jt * SC_TILE;
  const int i = blockIdx.x * SC_TILE + threadIdx.x;
  const bool ok = i < n;
  const P3 si = ok ? ld3(src, i) : P3{0, 0, 0};
  const float ci = ok ? conf[i] : 0.f;
  {
    const int j = jb + threadIdx.x;
    if (j < n) {
      ts[threadIdx.x][0] = src[3 * j]; ts[threadIdx.x][1] = src[3 * j + 1]; ts[threadIdx.x][2] = src[3 * j + 2];
      ts[threadIdx.x][3] = conf[j];
    }
  }
  __syncthreads();
  if (!ok) return;
  const int m = min(SC_TILE, n - jb);
  bool good = true;
  for (int q = (y == 0) ? (int)threadIdx.x + 1 : 0; q < m; ++q) {
    if (dist3(si, P3{ts[q][0], ts[q][1], ts[q][2]}) < radius) {
      const float cj = ts[q][3];
      if (ci < cj) good = false;
      else if (cj < ci) is_max[jb + q] = 0;
    }
  }
  if (!good) is_max[i] = 0;
}

// ---- tight compatibility as a bit matrix: bit j of row i = (cross_ij < thr)  (:354) ---------------------------
// one wavefront per row i: lanes = 64 consecutive columns j (coalesced point reads), one ballot per word
__global__ void __launch_bounds__(256) k_sc_tight_bits(const float* __restrict__ src, const float* __restrict__ tgt,
                                                       int n, int words, float thr, unsigned long long* bits) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) return;                                   // whole wave
  const P3 si = ld3(src, i), ti = ld3(tgt, i);
  for (int w = 0; w < words; ++w) {
    const int j = w * 64 + lane;
    bool ok = false;
    if (j < n) ok = fabsf(dist3(si, ld3(src, j)) - dist3(ti, ld3(tgt, j))) < thr;
    const unsigned long long b = __ballot(ok);
    if (lane == 0) bits[(size_t)i * words + w] = b;
  }
}

// ---- per seed: second-order measure row and its k1 largest entries (:353-361, :85-86) -------------------------
// one workgroup per seed; selection by (value desc, index asc).
// Round 5: (1) the AND + popcount of a compatible column j is done by a WAVE -- lanes read the 125 words of bit row j as two
// coalesced pieces (lane, lane + 64) against the seed's words held in registers, and four columns share one packed
// reduction -- instead of one lane walking the row word by word (64 lanes, 64 different 1 KB rows: every load instruction
// touched 64 cache lines; a seed that is an inlier has thousands of compatible columns); (2) the k1 selection rounds keep
// every thread's 32 values in registers and reduce with shuffles + one exchange through LDS (two barriers per round, not
// nine).  Same integers, same tie rule: the k1 lists are those of the round-1 kernel.
__global__ void __launch_bounds__(256) k_sc_seed_knn(const float* __restrict__ src, const float* __restrict__ tgt,
                                                     const unsigned long long* __restrict__ bits, int n, int words,
                                                     const long long* __restrict__ seeds, float d_thre, int k1,
                                                     int* knn) {
  __shared__ int vals[SC_MAXN];
  __shared__ int wv[4], wi[4];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int r = (int)seeds[blockIdx.x];
  // the seed's bit row: words lane and lane + 64 (words <= 128)
  const unsigned long long* br = bits + (size_t)r * words;
  const unsigned long long r0 = lane < words ? br[lane] : 0ull, r1 = lane + 64 < words ? br[lane + 64] : 0ull;
  const P3 sr = ld3(src, r), tr = ld3(tgt, r);
  for (int jb = 0; jb < n; jb += 256) {
    const int j = jb + t;
    bool hard = false;
    if (j < n) {
      hard = fabsf(dist3(sr, ld3(src, j)) - dist3(tr, ld3(tgt, j))) < d_thre;      // hard[seed][j]: others score 0
      vals[j] = 0;
    }
    unsigned long long m = __ballot(hard);      // this wave's 64 columns jb + 64 w + bit
    const int j0 = jb + 64 * w;
    while (m) {      // four compatible columns per trip: eight coalesced loads in flight, one packed reduction
      int jj[4];
      unsigned long long pk = 0;      // 4 x 16-bit lane counts (each <= 128; their wave sums <= 8192)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        jj[u] = -1;
        if (m) {
          jj[u] = j0 + __builtin_ctzll(m);
          m &= m - 1;
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (jj[u] >= 0) {      // uniform over the wave
          const unsigned long long* bj = bits + (size_t)jj[u] * words;
          const unsigned long long a0 = lane < words ? bj[lane] : 0ull, a1 = lane + 64 < words ? bj[lane + 64] : 0ull;
          pk |= (unsigned long long)(__popcll(r0 & a0) + __popcll(r1 & a1)) << (16 * u);
        }
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) pk += __shfl_xor(pk, o);      // four 16-bit sums at once (no carry: <= 8192 each)
      if (lane == 0) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (jj[u] >= 0) vals[jj[u]] = (int)((pk >> (16 * u)) & 0xffffull);
      }
    }
  }
  __syncthreads();
  // selection: thread t owns columns t, t + 256, ... (ascending), its values in registers
  int v[SC_MAXN / 256];
#pragma unroll
  for (int q = 0; q < SC_MAXN / 256; ++q) {
    const int j = t + 256 * q;
    v[q] = j < n ? vals[j] : -2;      // -2: no such column; -1: taken
  }
  for (int round = 0; round < k1; ++round) {
    int best = -2, besti = 0x7fffffff;
#pragma unroll
    for (int q = 0; q < SC_MAXN / 256; ++q)
      if (v[q] > best) { best = v[q]; besti = t + 256 * q; }      // ascending column per thread: first maximum = lowest index
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const int v2 = __shfl_xor(best, o), i2 = __shfl_xor(besti, o);
      if (v2 > best || (v2 == best && i2 < besti)) { best = v2; besti = i2; }
    }
    if (lane == 0) { wv[w] = best; wi[w] = besti; }
    __syncthreads();
    best = wv[0];
    besti = wi[0];
#pragma unroll
    for (int u = 1; u < 4; ++u)
      if (wv[u] > best || (wv[u] == best && wi[u] < besti)) { best = wv[u]; besti = wi[u]; }
    if (t == 0) knn[blockIdx.x * k1 + round] = besti;
    if ((besti & 255) == t) {      // the owner takes it out
#pragma unroll
      for (int q = 0; q < SC_MAXN / 256; ++q)
        if (q == (besti >> 8)) v[q] = -1;
    }
    __syncthreads();
  }
}

// ---- the same second-order rows, SEED-BLOCKED (round 6) ----------------------------------------------------------------------
// k_sc_seed_knn streams the 1 KB bit row of every compatible column ONCE PER SEED: a seed that is an inlier is compatible with
// every other inlier, so at an inlier share p the 0.2 n seeds read ~ 0.2 p n^2 KB from L2 -- 3.8 GB per registration at p = 0.3,
// n = 8000 (624 us; 1148 us at p = 0.6; the zero-inlier pairs of round 5's benchmark: 148 us).  Here a workgroup takes SK_TS
// seeds x one of the SC_CHUNKS column chunks: every wave holds the SK_TS seed rows in registers (2 x SK_TS 64-bit words per
// lane), the chunk's columns that are hard-compatible with ANY of the seeds are listed once, and a listed column's row is
// read once for all SK_TS seeds (lane = word, coalesced): SK_TS x fewer bytes, the AND + popcount work unchanged.  The 64
// lanes' partial counts of the SK_TS seeds are packed four to a 64-bit word and summed by a reduce-scatter butterfly (7 word
// shuffles per column instead of 6 per column and seed).  Counts that are not hard-compatible are zero, as in k_sc_seed_knn;
// the values go to vals[seed][column] (uint16: a count is <= n <= 8192) and k_sc_seed_topk makes the same selection from them.
// Integers throughout: the k1 lists are those of k_sc_seed_knn (tests: one call == staged calls, bit for bit).
constexpr int SK_TS = 16;
__global__ void __launch_bounds__(256) k_sc_seed_sc2(const float* __restrict__ src, const float* __restrict__ tgt,
                                                     const unsigned long long* __restrict__ bits, int n, int words,
                                                     const long long* __restrict__ seeds, int n_seeds, float d_thre,
                                                     unsigned short* __restrict__ vals) {
  __shared__ unsigned short out[SK_TS][SB_PER_MAX];
  __shared__ unsigned short lst[SB_PER_MAX], hm[SB_PER_MAX];
  __shared__ float sp[SK_TS][6];
  __shared__ int sid[SK_TS];
  __shared__ int cnt;
  const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int per = (n + SC_CHUNKS - 1) / SC_CHUNKS;
  const int j0 = blockIdx.y * per, j1 = min(n, j0 + per), m = j1 - j0;
  const int s0 = blockIdx.x * SK_TS;
  if (t < SK_TS) {
    const int r = s0 + t < n_seeds ? (int)seeds[s0 + t] : -1;
    sid[t] = r;
    for (int a = 0; a < 3; ++a) { sp[t][a] = r >= 0 ? src[3 * r + a] : 0.f; sp[t][3 + a] = r >= 0 ? tgt[3 * r + a] : 0.f; }
  }
  if (t == 0) cnt = 0;
  for (int q = t; q < SK_TS * SB_PER_MAX; q += 256) (&out[0][0])[q] = 0;
  __syncthreads();
  // hard[seed][j] for the chunk's columns (the same |.| < d_thre k_sc_seed_knn evaluates), columns with any bit listed
  for (int q = t; q < m; q += 256) {
    const P3 sj = ld3(src, j0 + q), tj = ld3(tgt, j0 + q);
    unsigned mask = 0;
#pragma unroll
    for (int u = 0; u < SK_TS; ++u) {
      const bool hard = fabsf(dist3(P3{sp[u][0], sp[u][1], sp[u][2]}, sj) - dist3(P3{sp[u][3], sp[u][4], sp[u][5]}, tj)) < d_thre;
      mask |= (unsigned)(hard && sid[u] >= 0) << u;
    }
    if (mask) {
      const int at = atomicAdd(&cnt, 1);      // any order: a column's counts do not depend on its place in the list
      lst[at] = (unsigned short)q;
      hm[at] = (unsigned short)mask;
    }
  }
  // the seeds' bit rows: words lane and lane + 64 (words <= 128)
  unsigned long long r0[SK_TS], r1[SK_TS];
#pragma unroll
  for (int u = 0; u < SK_TS; ++u) {
    const int r = sid[u];
    const unsigned long long* br = bits + (size_t)(r >= 0 ? r : 0) * words;
    r0[u] = (r >= 0 && lane < words) ? br[lane] : 0ull;
    r1[u] = (r >= 0 && lane + 64 < words) ? br[lane + 64] : 0ull;
  }
  __syncthreads();
  const int listed = cnt;
  const bool hi32 = lane & 32, hi16 = lane & 16;
  for (int c = w; c < listed; c += 4) {
    const int q = lst[c];
    const unsigned long long* bj = bits + (size_t)(j0 + q) * words;
    const unsigned long long a0 = lane < words ? bj[lane] : 0ull, a1 = lane + 64 < words ? bj[lane + 64] : 0ull;
    unsigned long long pk[SK_TS / 4];
#pragma unroll
    for (int g = 0; g < SK_TS / 4; ++g) {
      unsigned long long v = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k)
        v |= (unsigned long long)(__popcll(r0[4 * g + k] & a0) + __popcll(r1[4 * g + k] & a1)) << (16 * k);
      pk[g] = v;
    }
    // reduce-scatter over the lanes (16-bit fields: a wave's sum is <= 8192, no carry): lanes with bit 5 clear keep packs
    // 0, 1, the others 2, 3; then bit 4 picks one of the two; then a plain butterfly inside each group of 16 lanes
    unsigned long long A = hi32 ? pk[2] : pk[0], B = hi32 ? pk[3] : pk[1];
    A += __shfl_xor(hi32 ? pk[0] : pk[2], 32);
    B += __shfl_xor(hi32 ? pk[1] : pk[3], 32);
    unsigned long long X = hi16 ? B : A;
    X += __shfl_xor(hi16 ? A : B, 16);
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) X += __shfl_xor(X, o);
    if ((lane & 15) == 0) {
      const int g = (hi32 ? 2 : 0) + (hi16 ? 1 : 0);
      const unsigned mask = hm[c];
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if ((mask >> (4 * g + k)) & 1u) out[4 * g + k][q] = (unsigned short)((X >> (16 * k)) & 0xffffull);
    }
  }
  __syncthreads();
  for (int u = 0; u < SK_TS; ++u) {
    if (s0 + u >= n_seeds) break;
    unsigned short* row = vals + (size_t)(s0 + u) * n + j0;
    for (int q = t; q < m; q += 256) row[q] = out[u][q];
  }
}

// the k1 largest entries of a seed's row of `vals` (value descending, index ascending), ONE WAVE per seed over a two-level
// tournament: the row sits in LDS (16 KB), lane l keeps the maxima of the 64-column blocks l and l + 64 as keys
// (value << 13 | 8191 - column) + 1 -- the largest key is the largest value at the lowest column -- and a round is a butterfly
// over the block maxima, one store that marks the winner taken, and a butterfly over the winner's block read back by all 64
// lanes: ~ 20 instructions per round.  (Round 6's first forms kept every column of a thread in registers and let the owner of
// the winner scan them again: 32 columns x 4 waves + an exchange through LDS, or 125 columns in one wave -- 82 - 104 us per
// registration for 1600 seeds, all of it a single lane scanning while its wave waits.)
constexpr unsigned short SK_TAKEN = 0xffff;      // a count is <= 8192
__device__ __forceinline__ unsigned sc_topk_key(unsigned short v, int j, int n) {
  return (j < n && v != SK_TAKEN) ? ((((unsigned)v) << 13) | (unsigned)(8191 - j)) + 1u : 0u;
}
// maximum over the wave, uniform: four DPP steps inside every row of 16 lanes (quad swaps, half-row mirror, row mirror),
// then the four rows through scalar registers -- ~ 10 short instructions; a butterfly of ds_bpermute shuffles was a chain
// of six LDS-pipe round trips (k_sc_seed_topk: 82 -> 60 us with it, the rounds were nothing but that latency)
__device__ __forceinline__ unsigned sc_wave_umax(unsigned v) {
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true));       // quad_perm [1,0,3,2]
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true));       // quad_perm [2,3,0,1]
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true));      // row_half_mirror
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, true));      // row_mirror
  const unsigned a = (unsigned)__builtin_amdgcn_readlane((int)v, 0), b = (unsigned)__builtin_amdgcn_readlane((int)v, 16);
  const unsigned c = (unsigned)__builtin_amdgcn_readlane((int)v, 32), d = (unsigned)__builtin_amdgcn_readlane((int)v, 48);
  return max(max(a, b), max(c, d));
}
__global__ void __launch_bounds__(64) k_sc_seed_topk(const unsigned short* __restrict__ vals, int n, int k1, int* knn) {
  __shared__ unsigned short row[SC_MAXN];
  const int lane = threadIdx.x;
  const unsigned short* g = vals + (size_t)blockIdx.x * n;
  for (int j = lane; j < SC_MAXN; j += 64) row[j] = j < n ? g[j] : SK_TAKEN;
  __syncthreads();
  // lane l: the maxima of blocks l and l + 64, its 64 columns read in a rotated order (lanes on different banks)
  unsigned bm[2] = {0u, 0u};
#pragma unroll
  for (int h = 0; h < 2; ++h)
    for (int k = 0; k < 64; ++k) {
      const int j = 64 * (lane + 64 * h) + ((k + lane) & 63);
      bm[h] = max(bm[h], sc_topk_key(row[j], j, n));
    }
  for (int round = 0; round < k1; ++round) {
    const unsigned best = sc_wave_umax(max(bm[0], bm[1]));
    const int besti = 8191 - (int)((best - 1u) & 8191u);      // k1 <= n: there is always an entry left
    const int b = besti >> 6;
    if (lane == 0) {
      knn[blockIdx.x * k1 + round] = besti;
      row[besti] = SK_TAKEN;
    }
    __syncthreads();      // one wave: orders the store before the block's reads
    const int j = 64 * b + lane;
    const unsigned nb = sc_wave_umax(sc_topk_key(row[j], j, n));
    if (lane == (b & 63)) bm[b >> 6] = nb;
  }
}

// ---- 3 x 3 SVD (one-sided Jacobi, fp64) and the weighted Kabsch solution (common.py:7-45) ---------------------
// H = A^T W B;  R = V diag(1, 1, det(V U^T)) U^T with singular values in DESCENDING order (torch.svd);  t = cb - R ca
__device__ void kabsch_from_H(const double H[9], const double ca[3], const double cb[3], float* T12) {
  double A[3][3], V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) A[i][j] = H[3 * i + j];
  // H = U S V^T  <=>  one-sided Jacobi on the columns of H: H J1 J2 ... = U S, V = J1 J2 ...
  for (int sweep = 0; sweep < 30; ++sweep) {
    double off = 0;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        double al = 0, be = 0, ga = 0;
        for (int i = 0; i < 3; ++i) { al += A[i][p] * A[i][p]; be += A[i][q] * A[i][q]; ga += A[i][p] * A[i][q]; }
        off = fmax(off, fabs(ga) / (sqrt(al * be) + 1e-300));
        if (fabs(ga) <= 1e-300) continue;
        const double zeta = (be - al) / (2.0 * ga);
        const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
        const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
        for (int i = 0; i < 3; ++i) {
          const double ap = A[i][p], aq = A[i][q];
          A[i][p] = c * ap - s * aq; A[i][q] = s * ap + c * aq;
          const double vp = V[i][p], vq = V[i][q];
          V[i][p] = c * vp - s * vq; V[i][q] = s * vp + c * vq;
        }
      }
    if (off < 1e-15) break;
  }
  double sg[3];
  int ord[3] = {0, 1, 2};
  for (int j = 0; j < 3; ++j) sg[j] = sqrt(A[0][j] * A[0][j] + A[1][j] * A[1][j] + A[2][j] * A[2][j]);
  for (int a = 0; a < 2; ++a)
    for (int b = a + 1; b < 3; ++b)
      if (sg[ord[b]] > sg[ord[a]]) { int tmp = ord[a]; ord[a] = ord[b]; ord[b] = tmp; }
  double U[3][3], Vs[3][3];
  for (int j = 0; j < 3; ++j) {
    const int c = ord[j];
    for (int i = 0; i < 3; ++i) { Vs[i][j] = V[i][c]; U[i][j] = sg[c] > 1e-300 ? A[i][c] / sg[c] : 0.0; }
  }
  // rank-deficient H: complete U to an orthonormal basis (the result is then not unique, as in the reference)
  const double smax = sg[ord[0]];
  if (sg[ord[1]] <= 1e-12 * smax || smax <= 1e-300) {
    if (smax <= 1e-300) { U[0][0] = 1; U[1][0] = 0; U[2][0] = 0; }
    const int m = fabs(U[0][0]) < 0.9 ? 0 : 1;     // any vector not parallel to u0
    double e[3] = {0, 0, 0};
    e[m] = 1;
    double dp = e[0] * U[0][0] + e[1] * U[1][0] + e[2] * U[2][0], nn = 0;
    for (int i = 0; i < 3; ++i) { U[i][1] = e[i] - dp * U[i][0]; nn += U[i][1] * U[i][1]; }
    nn = sqrt(nn);
    for (int i = 0; i < 3; ++i) U[i][1] /= nn;
  }
  if (sg[ord[2]] <= 1e-12 * smax || smax <= 1e-300) {
    U[0][2] = U[1][0] * U[2][1] - U[2][0] * U[1][1];
    U[1][2] = U[2][0] * U[0][1] - U[0][0] * U[2][1];
    U[2][2] = U[0][0] * U[1][1] - U[1][0] * U[0][1];
  }
  // d = det(V U^T) = det(V) det(U)
  auto det3 = [](const double M[3][3]) {
    return M[0][0] * (M[1][1] * M[2][2] - M[1][2] * M[2][1]) - M[0][1] * (M[1][0] * M[2][2] - M[1][2] * M[2][0]) +
           M[0][2] * (M[1][0] * M[2][1] - M[1][1] * M[2][0]);
  };
  const double d = det3(Vs) * det3(U);
  double R[3][3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) R[i][j] = Vs[i][0] * U[j][0] + Vs[i][1] * U[j][1] + d * Vs[i][2] * U[j][2];
  for (int i = 0; i < 3; ++i) {
    const float r0 = (float)R[i][0], r1 = (float)R[i][1], r2 = (float)R[i][2];
    T12[4 * i] = r0; T12[4 * i + 1] = r1; T12[4 * i + 2] = r2;
    T12[4 * i + 3] = (float)(cb[i] - ((double)r0 * ca[0] + (double)r1 * ca[1] + (double)r2 * ca[2]));
  }
}

// ---- per seed: local consensus, power iteration, weighted transformation (:60-147), one wave per seed ---------
__global__ void __launch_bounds__(64) k_sc_seed_trans(const float* __restrict__ src, const float* __restrict__ tgt,
                                                      const int* __restrict__ knn, int k1, int k2, float d_thre,
                                                      int num_iterations, float* trans) {
  __shared__ float ps[32][3], pt[32][3], fs[32][3], ft[32][3], xv[32];
  __shared__ unsigned rowbits[32];
  const int lane = threadIdx.x, s = blockIdx.x;
  if (lane < k1) {
    const int r = knn[s * k1 + lane];
    for (int a = 0; a < 3; ++a) { ps[lane][a] = src[3 * r + a]; pt[lane][a] = tgt[3 * r + a]; }
  }
  __syncthreads();
  unsigned rb = 0;
  if (lane < k1)
    for (int b = 0; b < k1; ++b) {
      const float cd = fabsf(dist3(P3{ps[lane][0], ps[lane][1], ps[lane][2]}, P3{ps[b][0], ps[b][1], ps[b][2]}) -
                             dist3(P3{pt[lane][0], pt[lane][1], pt[lane][2]}, P3{pt[b][0], pt[b][1], pt[b][2]}));
      rb |= (unsigned)(cd < d_thre) << b;
    }
  if (lane < 32) rowbits[lane] = lane < k1 ? rb : 0u;
  __syncthreads();
  // local second-order score (:97) and its k2 largest (value desc, index asc): rank = position in the fine list
  const int score = lane < k1 ? __popc(rowbits[0] & rb) : -1;
  int rank = 0;
  for (int c = 0; c < k1; ++c) {
    const int sc = __popc(rowbits[0] & rowbits[c]);
    rank += (sc > score || (sc == score && c < lane)) ? 1 : 0;
  }
  if (lane < k1 && rank < k2)
    for (int a = 0; a < 3; ++a) { fs[rank][a] = ps[lane][a]; ft[rank][a] = pt[lane][a]; }
  __syncthreads();
  // soft 20 x 20 measure with zero diagonal (:119-131); lane p keeps row p
  float M[32];
#pragma unroll
  for (int q = 0; q < 32; ++q) {
    float m = 0.f;
    if (lane < k2 && q < k2 && q != lane) {
      const float cd = fabsf(dist3(P3{fs[lane][0], fs[lane][1], fs[lane][2]}, P3{fs[q][0], fs[q][1], fs[q][2]}) -
                             dist3(P3{ft[lane][0], ft[lane][1], ft[lane][2]}, P3{ft[q][0], ft[q][1], ft[q][2]}));
      m = fmaxf(1.f - cd * cd / (d_thre * d_thre), 0.f);
    }
    M[q] = m;
  }
  float x = lane < k2 ? 1.f : 0.f;
  for (int it = 0; it < num_iterations; ++it) {
    if (lane < 32) xv[lane] = x;
    __syncthreads();
    float y = 0.f;
#pragma unroll
    for (int q = 0; q < 32; ++q) y += M[q] * xv[q];
    const float nrm = sqrtf(wave_sum(lane < k2 ? y * y : 0.f));
    x = lane < k2 ? y / (nrm + 1e-6f) : 0.f;
    __syncthreads();
  }
  const float w = x / (wave_sum(x) + 1e-6f);                      // :132
  // weighted Kabsch (common.py:18-33): fp32 centroids and covariance like the reference, SVD in fp64
  const float sw = wave_sum(w) + 1e-6f;
  float a[3] = {0, 0, 0}, b[3] = {0, 0, 0};
  if (lane < k2)
    for (int c = 0; c < 3; ++c) { a[c] = fs[lane][c]; b[c] = ft[lane][c]; }
  float ca[3], cb[3];
  for (int c = 0; c < 3; ++c) { ca[c] = wave_sum(a[c] * w) / sw; cb[c] = wave_sum(b[c] * w) / sw; }
  double H[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) H[3 * i + j] = (double)wave_sum(lane < k2 ? (a[i] - ca[i]) * w * (b[j] - cb[j]) : 0.f);
  if (lane == 0) {
    const double cad[3] = {ca[0], ca[1], ca[2]}, cbd[3] = {cb[0], cb[1], cb[2]};
    kabsch_from_H(H, cad, cbd, trans + 12 * s);
  }
}

// ---- inlier count of every hypothesis (:149-161) --------------------------------------------------------------
__global__ void __launch_bounds__(256) k_sc_fitness(const float* __restrict__ src, const float* __restrict__ tgt, int n,
                                                    const float* __restrict__ trans, float thr, float* fitness) {
  __shared__ int red[256];
  const float* T = trans + 12 * blockIdx.x;
  int cnt = 0;
  for (int j = threadIdx.x; j < n; j += 256) {
    const P3 p = ld3(src, j), q = ld3(tgt, j);
    const float x = T[0] * p.x + T[1] * p.y + T[2] * p.z + T[3] - q.x;
    const float y = T[4] * p.x + T[5] * p.y + T[6] * p.z + T[7] - q.y;
    const float z = T[8] * p.x + T[9] * p.y + T[10] * p.z + T[11] - q.z;
    cnt += sqrtf(x * x + y * y + z * z) < thr;
  }
  red[threadIdx.x] = cnt;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) fitness[blockIdx.x] = (float)red[0];
}

// ---- post refinement (:238-279): weighted Kabsch over the inliers until the inlier count stops changing -------
constexpr int RF_BLOCKS = 64, RF_TERMS = 17;   // sum w, sum w a (3), sum w b (3), sum w a b^T (9), count
__global__ void __launch_bounds__(256) k_sc_refine_accum(const float* __restrict__ src, const float* __restrict__ tgt,
                                                         int n, const float* __restrict__ T, float thr,
                                                         const int* __restrict__ state, double* partial) {
  if (state[0]) return;
  __shared__ double red[256];
  double acc[RF_TERMS];
  for (int k = 0; k < RF_TERMS; ++k) acc[k] = 0;
  for (int j = blockIdx.x * 256 + threadIdx.x; j < n; j += RF_BLOCKS * 256) {
    const P3 p = ld3(src, j), q = ld3(tgt, j);
    const float x = T[0] * p.x + T[1] * p.y + T[2] * p.z + T[3] - q.x;
    const float y = T[4] * p.x + T[5] * p.y + T[6] * p.z + T[7] - q.y;
    const float z = T[8] * p.x + T[9] * p.y + T[10] * p.z + T[11] - q.z;
    const float d = sqrtf(x * x + y * y + z * z);
    if (d < thr) {
      const float r = d / thr;
      const double w = 1.f / (1.f + r * r);
      const double a[3] = {p.x, p.y, p.z}, b[3] = {q.x, q.y, q.z};
      acc[0] += w;
      for (int c = 0; c < 3; ++c) { acc[1 + c] += w * a[c]; acc[4 + c] += w * b[c]; }
      for (int i = 0; i < 3; ++i)
        for (int k = 0; k < 3; ++k) acc[7 + 3 * i + k] += w * a[i] * b[k];
      acc[16] += 1.0;
    }
  }
  for (int k = 0; k < RF_TERMS; ++k) {
    red[threadIdx.x] = acc[k];
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
      __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x * RF_TERMS + k] = red[0];
    __syncthreads();
  }
}

// state[0] = done, state[1] = previous inlier count
__global__ void k_sc_refine_solve(const double* __restrict__ partial, int* state, float* T) {
  if (state[0]) return;
  double s[RF_TERMS];
  for (int k = 0; k < RF_TERMS; ++k) {
    double v = 0;
    for (int b = 0; b < RF_BLOCKS; ++b) v += partial[b * RF_TERMS + k];
    s[k] = v;
  }
  const int cnt = (int)s[16];
  if (abs(cnt - state[1]) < 1) {      // :266-267
    state[0] = 1;
    return;
  }
  state[1] = cnt;
  const double sw = s[0] + 1e-6;      // common.py:22-23
  double ca[3], cb[3], H[9];
  for (int c = 0; c < 3; ++c) { ca[c] = s[1 + c] / sw; cb[c] = s[4 + c] / sw; }
  // sum w (a - ca)(b - cb)^T = sum w a b^T - ca (sum w b)^T - (sum w a) cb^T + (sum w) ca cb^T
  for (int i = 0; i < 3; ++i)
    for (int k = 0; k < 3; ++k)
      H[3 * i + k] = s[7 + 3 * i + k] - ca[i] * s[4 + k] - s[1 + i] * cb[k] + s[0] * ca[i] * cb[k];
  kabsch_from_H(H, ca, cb, T);
}

// The whole refinement in ONE launch (round 5): a single 1024-thread workgroup runs the iterations and stops at convergence
// (the two-kernel form launched 2 x 20 kernels per registration, most of them no-ops behind the `done` flag once the inlier
// count stands still: 0.3 ms of dependent launches per pair).  A thread owns points t, t + 1024, ... and sums their 17
// weighted-Kabsch terms in fp64 in that order; the 64 lanes of a wave are combined by a butterfly of shuffles, the 16 waves
// in wave order by thread 0 -- a fixed order, so the result is reproducible; it is NOT the summation order of the two-kernel
// form (64 blocks of 256, a halving tree each): the fp64 sums may differ in their last bits, the fp32 transformation
// practically never (tests/test_gpu_parity.py compares the two forms on the golden problems).
__global__ void __launch_bounds__(1024) k_sc_refine_all(const float* __restrict__ src, const float* __restrict__ tgt, int n,
                                                        float thr, int iterations, int* state, float* T) {
  __shared__ double ws[16][RF_TERMS];
  __shared__ float Ts[12];
  __shared__ int done;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  if (t < 12) Ts[t] = T[t];
  if (t == 0) done = 0;
  int prev = 0;      // thread 0's
  __syncthreads();
  for (int it = 0; it < iterations; ++it) {
    double acc[RF_TERMS];
#pragma unroll
    for (int k = 0; k < RF_TERMS; ++k) acc[k] = 0;
    for (int j = t; j < n; j += 1024) {
      const P3 p = ld3(src, j), q = ld3(tgt, j);
      const float x = Ts[0] * p.x + Ts[1] * p.y + Ts[2] * p.z + Ts[3] - q.x;
      const float y = Ts[4] * p.x + Ts[5] * p.y + Ts[6] * p.z + Ts[7] - q.y;
      const float z = Ts[8] * p.x + Ts[9] * p.y + Ts[10] * p.z + Ts[11] - q.z;
      const float d = sqrtf(x * x + y * y + z * z);
      if (d < thr) {
        const float r = d / thr;
        const double wgt = 1.f / (1.f + r * r);
        const double a[3] = {p.x, p.y, p.z}, b[3] = {q.x, q.y, q.z};
        acc[0] += wgt;
#pragma unroll
        for (int c = 0; c < 3; ++c) { acc[1 + c] += wgt * a[c]; acc[4 + c] += wgt * b[c]; }
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int k = 0; k < 3; ++k) acc[7 + 3 * i + k] += wgt * a[i] * b[k];
        acc[16] += 1.0;
      }
    }
#pragma unroll
    for (int k = 0; k < RF_TERMS; ++k) {
      double v = acc[k];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
      if (lane == 0) ws[w][k] = v;
    }
    __syncthreads();
    if (t == 0) {
      double s[RF_TERMS];
      for (int k = 0; k < RF_TERMS; ++k) {
        double v = 0;
        for (int u = 0; u < 16; ++u) v += ws[u][k];
        s[k] = v;
      }
      const int cnt = (int)s[16];
      if (abs(cnt - prev) < 1) {      // :266-267
        done = 1;
      } else {
        prev = cnt;
        const double sw = s[0] + 1e-6;      // common.py:22-23
        double ca[3], cb[3], H[9];
        for (int c = 0; c < 3; ++c) { ca[c] = s[1 + c] / sw; cb[c] = s[4 + c] / sw; }
        for (int i = 0; i < 3; ++i)
          for (int k = 0; k < 3; ++k)
            H[3 * i + k] = s[7 + 3 * i + k] - ca[i] * s[4 + k] - s[1 + i] * cb[k] + s[0] * ca[i] * cb[k];
        float Tn[12];
        kabsch_from_H(H, ca, cb, Tn);
        for (int k = 0; k < 12; ++k) Ts[k] = Tn[k];
      }
    }
    __syncthreads();
    if (done) break;
  }
  if (t < 12) T[t] = Ts[t];
  if (t == 0) { state[0] = done; state[1] = prev; }
}

// ---- one call per registration (round 5): what scripts/SC2_PCR.py did between the stages, on the device ------------------------
__global__ void k_sc_reg_init(float* conf, int* is_max, int* rank, int* done, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    conf[i] = 1.f;
    is_max[i] = 1;
    rank[i] = 0;
  }
  if (i == 0) *done = 0;
}

// seeds = the n_seeds first of argsort(-(conf * is_max), stable) (:56-58): value descending, index ascending.  By RANK: the
// place of correspondence i is the number of correspondences ordered before it -- n^2 comparisons spread over the chip
// (row tiles x SC_CHUNKS column chunks, integer atomics: any order gives the same count) -- then seeds[rank[i]] = i.  (A
// bitonic sort in LDS by one workgroup was the first form: 111 us; torch's sort took 45 us in four launches.)  NaN last, as
// torch.sort places it.
__device__ __forceinline__ unsigned sc_seed_key(const float* __restrict__ conf, const int* __restrict__ is_max, int i) {
  float v = conf[i] * (float)is_max[i];
  if (v == 0.f) v = 0.f;                                                // -0 and +0 are one value to the reference's sort
  const unsigned b = __float_as_uint(v);
  const unsigned asc = b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u);     // ascending in v
  return (v != v) ? 0xFFFFFFFFu : ~asc;                                 // ascending key = descending value
}
__global__ void __launch_bounds__(SC_TILE) k_sc_seed_rank(const float* __restrict__ conf, const int* __restrict__ is_max, int n,
                                                          int* rank) {
  __shared__ unsigned tk[SC_TILE];
  const int i = blockIdx.x * SC_TILE + threadIdx.x;
  const bool ok = i < n;
  const unsigned ki = ok ? sc_seed_key(conf, is_max, i) : 0u;
  const int per = (n + SC_CHUNKS - 1) / SC_CHUNKS;
  const int j0 = blockIdx.y * per, j1 = min(n, j0 + per);
  int before = 0;
  for (int jb = j0; jb < j1; jb += SC_TILE) {
    __syncthreads();
    if (jb + (int)threadIdx.x < j1) tk[threadIdx.x] = sc_seed_key(conf, is_max, jb + threadIdx.x);
    __syncthreads();
    const int m = min(SC_TILE, j1 - jb);
    for (int q = 0; q < m; ++q) {
      const unsigned kj = tk[q];
      before += (kj < ki || (kj == ki && jb + q < i)) ? 1 : 0;
    }
  }
  if (ok && before) atomicAdd(&rank[i], before);
}
__global__ void k_sc_seed_place(const int* __restrict__ rank, int n, int n_seeds, long long* __restrict__ seeds) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && rank[i] < n_seeds) seeds[rank[i]] = i;
}

// best = lowest index of the maximum fitness (torch.sort(-fitness, stable)[1][0]); T = that seed's [R | t]
__global__ void __launch_bounds__(256) k_sc_best(const float* __restrict__ fitness, const float* __restrict__ trans, int n_seeds,
                                                 int* best_out, float* T) {
  __shared__ float bv[256];
  __shared__ int bi[256];
  float v = -INFINITY;
  int ix = 0x7fffffff;
  for (int i = threadIdx.x; i < n_seeds; i += 256) {      // ascending i per thread: the first maximum is the lowest index
    const float f = fitness[i];
    if (ix == 0x7fffffff || f > v) { v = f; ix = i; }
  }
  bv[threadIdx.x] = v;
  bi[threadIdx.x] = ix;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) {
      const float v2 = bv[threadIdx.x + o];
      const int i2 = bi[threadIdx.x + o];
      if (i2 != 0x7fffffff && (bi[threadIdx.x] == 0x7fffffff || v2 > bv[threadIdx.x] ||
                               (v2 == bv[threadIdx.x] && i2 < bi[threadIdx.x]))) {
        bv[threadIdx.x] = v2;
        bi[threadIdx.x] = i2;
      }
    }
    __syncthreads();
  }
  const int best = bi[0];
  if (threadIdx.x == 0) *best_out = best;
  if (threadIdx.x < 12) T[threadIdx.x] = trans[(size_t)best * 12 + threadIdx.x];
}

// the [4, 4] transformation and the inlier labels |R s + t - t'| < thr of Matcher.estimator (:404-409)
__global__ void k_sc_finish(const float* __restrict__ src, const float* __restrict__ tgt, int n, const float* __restrict__ T,
                            float thr, float* out16, float* labels) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < 16) out16[i] = i < 12 ? T[i] : (i == 15 ? 1.f : 0.f);
  if (i >= n) return;
  const P3 s = ld3(src, i), g = ld3(tgt, i);
  const float wx = T[0] * s.x + T[1] * s.y + T[2] * s.z + T[3] - g.x;
  const float wy = T[4] * s.x + T[5] * s.y + T[6] * s.z + T[7] - g.y;
  const float wz = T[8] * s.x + T[9] * s.y + T[10] * s.z + T[11] - g.z;
  labels[i] = sqrtf(wx * wx + wy * wy + wz * wz) < thr ? 1.f : 0.f;
}

}  // namespace gcl

using namespace gcl;

extern "C" {

int32_t gcl_sc2_chunks(void) { return SC_CHUNKS; }
int32_t gcl_sc2_refine_partial_len(void) { return RF_BLOCKS * RF_TERMS; }

int gcl_sc2_confidence(const float* src, const float* tgt, int32_t n, float d_thre, int32_t num_iterations,
                       float* partial, float* x, int32_t* done, void* stream) {
  GCL_CHECK_ARG(src && tgt && partial && x && done, "gcl_sc2_confidence: null pointer");
  GCL_CHECK_ARG(n > 0 && n <= SC_MAXN && d_thre > 0 && num_iterations >= 0, "gcl_sc2_confidence: 0 < n <= %d", SC_MAXN);
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((unsigned)cdiv(n, SC_TILE), SC_CHUNKS);
  for (int it = 0; it < num_iterations; ++it) {
    hipLaunchKernelGGL(k_sc_matvec, grid, dim3(SC_TILE), 0, st, src, tgt, n, d_thre * d_thre, (const float*)x,
                       (const int*)done, partial);
    hipLaunchKernelGGL(k_sc_normalize, dim3(1), dim3(1024), 0, st, (const float*)partial, n, x, done);
  }
  GCL_CHECK_LAUNCH();
  return GCL_OK;
}

int64_t gcl_sc2_confidence_scratch_bytes(int32_t n) {
  if (n <= 0 || n > SC_MAXN) return 0;
  const long long per = cdiv(n, SC_CHUNKS);
  return (long long)SC_CHUNKS * n * 4 + 256 + (long long)SC_CHUNKS * n * per * (long long)sizeof(ScEntry) + (long long)SC_CHUNKS * n * 4;
}

static bool sc_folded_normalize() {      // GCL_SC2_FOLDED_NORMALIZE=0: every product followed by its own k_sc_normalize launch
  static const bool on = [] {
    const char* e = getenv("GCL_SC2_FOLDED_NORMALIZE");
    return !(e && e[0] == '0');
  }();
  return on;
}

static int sc_confidence_sparse(const float* src, const float* tgt, int32_t n, float d_thre, int32_t num_iterations,
                                float* partial, float* x, int32_t* done, void* scratch, float tight_thr,
                                unsigned long long* bits, void* stream) {
  GCL_CHECK_ARG(src && tgt && partial && x && done && scratch, "gcl_sc2_confidence_sparse: null pointer");
  GCL_CHECK_ARG(n > 0 && n <= SC_MAXN && d_thre > 0 && num_iterations >= 0, "gcl_sc2_confidence: 0 < n <= %d", SC_MAXN);
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((unsigned)cdiv(n, SC_TILE), SC_CHUNKS);
  int* count = (int*)scratch;
  const size_t entries_off = ((size_t)SC_CHUNKS * n * 4 + 255) & ~(size_t)255;
  ScEntry* entries = (ScEntry*)((char*)scratch + entries_off);
  float* partial2 = (float*)((char*)scratch + entries_off + (size_t)SC_CHUNKS * n * cdiv(n, SC_CHUNKS) * sizeof(ScEntry));
  const float d2 = d_thre * d_thre;
  if (num_iterations > 0)
    hipLaunchKernelGGL(k_sc_sparse_build, dim3((unsigned)cdiv(n, SB_ROWS), SC_CHUNKS), dim3(256), 0, st, src, tgt, n, d2, count,
                       entries, tight_thr, bits);
  if (sc_folded_normalize() && num_iterations > 0) {
    float* buf[2] = {partial, partial2};
    hipLaunchKernelGGL(k_sc_matvec_folded<true>, grid, dim3(SC_TILE), 0, st, n, x, done, (const float*)nullptr, buf[0],
                       (const int*)count, (const ScEntry*)entries);
    for (int it = 1; it < num_iterations; ++it)
      hipLaunchKernelGGL(k_sc_matvec_folded<false>, grid, dim3(SC_TILE), 0, st, n, x, done, (const float*)buf[(it - 1) & 1],
                         buf[it & 1], (const int*)count, (const ScEntry*)entries);
    hipLaunchKernelGGL(k_sc_normalize, dim3(1), dim3(1024), 0, st, (const float*)buf[(num_iterations - 1) & 1], n, x, done);
  } else {
    for (int it = 0; it < num_iterations; ++it) {
      hipLaunchKernelGGL(k_sc_matvec_sparse, grid, dim3(SC_TILE), 0, st, n, (const float*)x, (const int*)done, partial,
                         (const int*)count, (const ScEntry*)entries);
      hipLaunchKernelGGL(k_sc_normalize, dim3(1), dim3(1024), 0, st, (const float*)partial, n, x, done);
    }
  }
  GCL_CHECK_LAUNCH();
  return GCL_OK;
}

int gcl_sc2_confidence_sparse(const float* src, const float* tgt, int32_t n, float d_thre, int32_t num_iterations,
                              float* partial, float* x, int32_t* done, void* scratch, void* stream) {
  return sc_confidence_sparse(src, tgt, n, d_thre, num_iterations, partial, x, done, scratch, 0.f, nullptr, stream);
}

int gcl_sc2_local_max(const float* src, const float* conf, int32_t n, float radius, int32_t* is_max, void* stream) {
  GCL_CHECK_ARG(src && conf && is_max && n > 0, "gcl_sc2_local_max: bad argument");
  hipLaunchKernelGGL(k_sc_local_max, dim3((unsigned)cdiv(n, SC_TILE), (unsigned)(cdiv(n, SC_TILE) / 2 + 1)), dim3(SC_TILE), 0,
                     (hipStream_t)stream, src, conf, n, radius, is_max);
  GCL_CHECK_LAUNCH();
  return GCL_OK;
}

static bool sc_seed_blocked() {      // GCL_SC2_SEED_BLOCKED=0: k_sc_seed_knn (one workgroup per seed) in the one-call form too
  static const bool on = [] {
    const char* e = getenv("GCL_SC2_SEED_BLOCKED");
    return !(e && e[0] == '0');
  }();
  return on;
}

// `vals` (uint16[n_seeds * n], may be null): scratch of the seed-blocked form; without it one workgroup per seed streams the
// rows of its compatible columns (k_sc_seed_knn, the staged entry point's form -- same integers, same lists)
static int sc_seed_knn(const float* src, const float* tgt, int32_t n, const int64_t* seeds, int32_t n_seeds,
                       float d_thre, int32_t k1, uint64_t* bits, bool make_bits, int32_t* knn, void* stream,
                       unsigned short* vals = nullptr) {
  GCL_CHECK_ARG(src && tgt && seeds && bits && knn, "gcl_sc2_seed_knn: null pointer");
  GCL_CHECK_ARG(n > 0 && n <= SC_MAXN && n_seeds > 0 && k1 >= 1 && k1 <= 32 && k1 <= n,
                "gcl_sc2_seed_knn: need n <= %d, 1 <= k1 <= min(32, n)", SC_MAXN);
  hipStream_t st = (hipStream_t)stream;
  const int words = (n + 63) / 64;
  if (make_bits)
    hipLaunchKernelGGL(k_sc_tight_bits, dim3((unsigned)cdiv(n, 4)), dim3(256), 0, st, src, tgt, n, words, d_thre * 0.5f,
                       (unsigned long long*)bits);
  if (vals && sc_seed_blocked()) {
    hipLaunchKernelGGL(k_sc_seed_sc2, dim3((unsigned)cdiv(n_seeds, SK_TS), SC_CHUNKS), dim3(256), 0, st, src, tgt,
                       (const unsigned long long*)bits, n, words, (const long long*)seeds, n_seeds, d_thre, vals);
    hipLaunchKernelGGL(k_sc_seed_topk, dim3(n_seeds), dim3(64), 0, st, (const unsigned short*)vals, n, k1, knn);
  } else {
    hipLaunchKernelGGL(k_sc_seed_knn, dim3(n_seeds), dim3(256), 0, st, src, tgt, (const unsigned long long*)bits, n,
                       words, (const long long*)seeds, d_thre, k1, knn);
  }
  GCL_CHECK_LAUNCH();
  return GCL_OK;
}

int gcl_sc2_seed_knn(const float* src, const float* tgt, int32_t n, const int64_t* seeds, int32_t n_seeds,
                     float d_thre, int32_t k1, uint64_t* bits, int32_t* knn, void* stream) {
  return sc_seed_knn(src, tgt, n, seeds, n_seeds, d_thre, k1, bits, true, knn, stream);
}

int gcl_sc2_seed_trans(const float* src, const float* tgt, int32_t n, const int32_t* knn, int32_t n_seeds, int32_t k1,
                       int32_t k2, float d_thre, int32_t num_iterations, float inlier_thresh, float* trans,
                       float* fitness, void* stream) {
  GCL_CHECK_ARG(src && tgt && knn && trans && fitness, "gcl_sc2_seed_trans: null pointer");
  GCL_CHECK_ARG(n > 0 && n_seeds > 0 && k1 >= 1 && k1 <= 32 && k2 >= 1 && k2 <= k1, "gcl_sc2_seed_trans: bad k1 / k2");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_sc_seed_trans, dim3(n_seeds), dim3(64), 0, st, src, tgt, knn, k1, k2, d_thre, num_iterations,
                     trans);
  hipLaunchKernelGGL(k_sc_fitness, dim3(n_seeds), dim3(256), 0, st, src, tgt, n, (const float*)trans, inlier_thresh,
                     fitness);
  GCL_CHECK_LAUNCH();
  return GCL_OK;
}

int gcl_sc2_refine(const float* src, const float* tgt, int32_t n, float thr, int32_t iterations, double* partial,
                   int32_t* state, float* T, void* stream) {
  GCL_CHECK_ARG(src && tgt && partial && state && T && n > 0 && iterations >= 0, "gcl_sc2_refine: bad argument");
  hipStream_t st = (hipStream_t)stream;
  static const int one_launch = [] { const char* e = getenv("GCL_SC2_REFINE_ONE_LAUNCH"); return e ? atoi(e) : 1; }();
  if (one_launch) {
    hipLaunchKernelGGL(k_sc_refine_all, dim3(1), dim3(1024), 0, st, src, tgt, n, thr, iterations, state, T);
    GCL_CHECK_LAUNCH();
    return GCL_OK;
  }
  GCL_CHECK_HIP(hipMemsetAsync(state, 0, 2 * sizeof(int32_t), st));
  for (int it = 0; it < iterations; ++it) {
    hipLaunchKernelGGL(k_sc_refine_accum, dim3(RF_BLOCKS), dim3(256), 0, st, src, tgt, n, (const float*)T, thr,
                       (const int*)state, partial);
    hipLaunchKernelGGL(k_sc_refine_solve, dim3(1), dim3(1), 0, st, (const double*)partial, state, T);
  }
  GCL_CHECK_LAUNCH();
  return GCL_OK;
}


static size_t sc_up256(size_t b) { return (b + 255) & ~(size_t)255; }
struct ScRegLayout { size_t partial, done, sparse, is_max, rank, bits, T, rpart, state, total; };
static ScRegLayout sc_reg_layout(int n) {
  ScRegLayout L;
  size_t o = 0;
  L.partial = o; o += sc_up256((size_t)SC_CHUNKS * n * 4);
  L.done = o;    o += 256;
  L.sparse = o;  o += sc_up256((size_t)gcl_sc2_confidence_scratch_bytes(n));
  L.is_max = o;  o += sc_up256((size_t)n * 4);
  L.rank = o;    o += sc_up256((size_t)n * 4);
  L.bits = o;    o += sc_up256((size_t)n * ((n + 63) / 64) * 8);
  L.T = o;       o += 256;
  L.rpart = o;   o += sc_up256((size_t)RF_BLOCKS * RF_TERMS * 8);
  L.state = o;   o += 256;
  L.total = o;
  return L;
}

int64_t gcl_sc2_register_scratch_bytes(int32_t n) {
  if (n <= 0 || n > SC_MAXN) return 0;
  return (int64_t)sc_reg_layout(n).total;
}

int gcl_sc2_register(const float* src, const float* tgt, int32_t n, float d_thre, int32_t num_iterations, float nms_radius,
                     int32_t n_seeds, int32_t k1, int32_t k2, float inlier_thresh, float refine_thr, int32_t refine_iters,
                     void* scratch, float* conf, int64_t* seeds, int32_t* knn, float* seed_trans, float* fitness,
                     int32_t* best, float* trans16, float* labels, void* stream) {
  GCL_CHECK_ARG(src && tgt && scratch && conf && seeds && knn && seed_trans && fitness && best && trans16 && labels,
                "gcl_sc2_register: null pointer");
  GCL_CHECK_ARG(n > 0 && n <= SC_MAXN && n_seeds >= 1 && n_seeds <= n, "gcl_sc2_register: need 1 <= n_seeds <= n <= %d",
                SC_MAXN);
  hipStream_t st = (hipStream_t)stream;
  const ScRegLayout L = sc_reg_layout(n);
  char* base = (char*)scratch;
  float* partial = (float*)(base + L.partial);
  int* done = (int*)(base + L.done);
  int* is_max = (int*)(base + L.is_max);
  float* T = (float*)(base + L.T);
  int* rank = (int*)(base + L.rank);
  hipLaunchKernelGGL(k_sc_reg_init, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, st, conf, is_max, rank, done, n);
  // the tight compatibility bits come out of the confidence's build pass (one evaluation of the n^2 length differences, not two)
  const bool bits_from_build = num_iterations > 0;
  int rc = sc_confidence_sparse(src, tgt, n, d_thre, num_iterations, partial, conf, done, base + L.sparse, d_thre * 0.5f,
                                bits_from_build ? (unsigned long long*)(base + L.bits) : nullptr, stream);
  if (rc != GCL_OK) return rc;
  rc = gcl_sc2_local_max(src, conf, n, nms_radius, is_max, stream);
  if (rc != GCL_OK) return rc;
  hipLaunchKernelGGL(k_sc_seed_rank, dim3((unsigned)cdiv(n, SC_TILE), SC_CHUNKS), dim3(SC_TILE), 0, st, (const float*)conf,
                     (const int*)is_max, n, rank);
  hipLaunchKernelGGL(k_sc_seed_place, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, st, (const int*)rank, n, n_seeds,
                     (long long*)seeds);
  // the second-order values of the seed-blocked form (uint16[n_seeds * n] <= 2 n^2 bytes) take the place of the confidence's
  // entry slab (8 n^2 bytes, behind its counts), which nothing reads after the last product
  unsigned short* vals = (unsigned short*)(base + L.sparse + (((size_t)SC_CHUNKS * n * 4 + 255) & ~(size_t)255));
  rc = sc_seed_knn(src, tgt, n, seeds, n_seeds, d_thre, k1, (uint64_t*)(base + L.bits), !bits_from_build, knn, stream, vals);
  if (rc != GCL_OK) return rc;
  rc = gcl_sc2_seed_trans(src, tgt, n, knn, n_seeds, k1, k2, d_thre, num_iterations, inlier_thresh, seed_trans, fitness,
                          stream);
  if (rc != GCL_OK) return rc;
  hipLaunchKernelGGL(k_sc_best, dim3(1), dim3(256), 0, st, (const float*)fitness, (const float*)seed_trans, n_seeds, best, T);
  rc = gcl_sc2_refine(src, tgt, n, refine_thr, refine_iters, (double*)(base + L.rpart), (int32_t*)(base + L.state), T, stream);
  if (rc != GCL_OK) return rc;
  hipLaunchKernelGGL(k_sc_finish, dim3((unsigned)cdiv(std::max(n, 16), 256)), dim3(256), 0, st, src, tgt, n, (const float*)T,
                     inlier_thresh, trans16, labels);
  GCL_CHECK_LAUNCH();
  return GCL_OK;
}

}  // extern "C"
