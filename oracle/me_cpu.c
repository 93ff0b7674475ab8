/* CPU ORACLE (test infrastructure, NOT product code): C / OpenMP restatement of MinkowskiEngine's CPU algorithm for
 * the generalized sparse convolution -- coordinate hash map -> per-offset in/out index lists (kernel map) -> per offset:
 * gather rows, GEMM with W_k, scatter-add (Choy et al., CVPR'19, sec. 4; the structure SURVEY.md 2.1 / 8(d) describes
 * for ME's CPU backend).  MinkowskiEngine itself is absent from /root/reference and not installable here, so this is a
 * restatement ("port"), pinned by tests/test_oracle_conv.py against oracle/me_oracle.py, which in turn is pinned against
 * dense torch conv3d / conv_transpose3d.  Used by bench.py's cpu_baseline leg (timed on the GPU box's host cores) and by
 * tests; never by the product package.
 *
 * Semantics (identical to oracle/me_oracle.py): coordinates int32 [N,4] = (batch,x,y,z); strided map = unique
 * floor(c / t) * t in first-occurrence order; kernel offsets x fastest in -(ks/2)..+(ks/2), scaled by `step`, region
 * centred on the OUTPUT coordinate; W [K, Cin, Cout]; transposed conv = forward map with in/out swapped.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <omp.h>

#define EMPTY (~0ull)

static inline uint64_t pack4(const int32_t* c) {
  return ((uint64_t)(uint32_t)c[0] << 48) | ((uint64_t)(uint32_t)(c[1] + 32768) << 32) |
         ((uint64_t)(uint32_t)(c[2] + 32768) << 16) | (uint64_t)(uint32_t)(c[3] + 32768);
}
static inline uint64_t mix64(uint64_t h) {
  h ^= h >> 33; h *= 0xff51afd7ed558ccdull; h ^= h >> 33; h *= 0xc4ceb9fe1a85ec53ull; h ^= h >> 33;
  return h;
}
static inline int in_range(const int32_t* c) {
  return (uint32_t)c[0] < 65535u && (uint32_t)(c[1] + 32768) < 65536u && (uint32_t)(c[2] + 32768) < 65536u &&
         (uint32_t)(c[3] + 32768) < 65536u;
}

/* keys[cap] (cap = power of two >= 2n), vals[cap].  Inserts rows in order; a duplicate keeps its first row.
 * Returns the number of distinct keys, or -1 if a coordinate is outside the packable range. */
int64_t me_map_build(const int32_t* coords, int64_t n, uint64_t* keys, int32_t* vals, int64_t cap) {
  for (int64_t i = 0; i < cap; ++i) keys[i] = EMPTY;
  int64_t distinct = 0;
  for (int64_t i = 0; i < n; ++i) {
    if (!in_range(coords + 4 * i)) return -1;
    uint64_t k = pack4(coords + 4 * i);
    uint64_t s = mix64(k) & (uint64_t)(cap - 1);
    while (keys[s] != EMPTY && keys[s] != k) s = (s + 1) & (uint64_t)(cap - 1);
    if (keys[s] == EMPTY) { keys[s] = k; vals[s] = (int32_t)i; ++distinct; }
  }
  return distinct;
}

static inline int32_t map_find(const uint64_t* keys, const int32_t* vals, int64_t cap, uint64_t k) {
  uint64_t s = mix64(k) & (uint64_t)(cap - 1);
  while (1) {
    uint64_t q = keys[s];
    if (q == k) return vals[s];
    if (q == EMPTY) return -1;
    s = (s + 1) & (uint64_t)(cap - 1);
  }
}

static inline int32_t floordiv(int32_t a, int32_t b) { int32_t q = a / b; return (a % b != 0 && ((a < 0) != (b < 0))) ? q - 1 : q; }

/* Strided coordinate map: out = unique(floor(c / t) * t) in first-occurrence order, hash map of the out coordinates in
 * keys / vals (cap >= 2 n).  Returns n_out. */
int64_t me_stride_coords(const int32_t* coords_in, int64_t n, int32_t t, int32_t* coords_out, uint64_t* keys,
                         int32_t* vals, int64_t cap) {
  for (int64_t i = 0; i < cap; ++i) keys[i] = EMPTY;
  int64_t m = 0;
  for (int64_t i = 0; i < n; ++i) {
    int32_t c[4] = {coords_in[4 * i], floordiv(coords_in[4 * i + 1], t) * t, floordiv(coords_in[4 * i + 2], t) * t,
                    floordiv(coords_in[4 * i + 3], t) * t};
    uint64_t k = pack4(c);
    uint64_t s = mix64(k) & (uint64_t)(cap - 1);
    while (keys[s] != EMPTY && keys[s] != k) s = (s + 1) & (uint64_t)(cap - 1);
    if (keys[s] == EMPTY) {
      keys[s] = k;
      vals[s] = (int32_t)m;
      memcpy(coords_out + 4 * m, c, sizeof(c));
      ++m;
    }
  }
  return m;
}

/* Kernel map: for every offset k the compacted (in, out) row pairs, ascending out row.  in_rows / out_rows: [K * n_out]
 * (segment k starts at k * n_out), counts[K].  Parallel over offsets. */
void me_kernel_map(const int32_t* coords_out, int64_t n_out, const uint64_t* keys_in, const int32_t* vals_in,
                   int64_t cap_in, int32_t ks, int32_t step, int32_t* in_rows, int32_t* out_rows, int64_t* counts) {
  const int K = ks * ks * ks, r = ks / 2;
#pragma omp parallel for schedule(dynamic, 1)
  for (int k = 0; k < K; ++k) {
    const int32_t ox = (k % ks - r) * step, oy = ((k / ks) % ks - r) * step, oz = (k / (ks * ks) - r) * step;
    int32_t* pi = in_rows + (int64_t)k * n_out;
    int32_t* po = out_rows + (int64_t)k * n_out;
    int64_t cnt = 0;
    for (int64_t v = 0; v < n_out; ++v) {
      int32_t c[4] = {coords_out[4 * v], coords_out[4 * v + 1] + ox, coords_out[4 * v + 2] + oy, coords_out[4 * v + 3] + oz};
      if (!in_range(c)) continue;
      int32_t u = map_find(keys_in, vals_in, cap_in, pack4(c));
      if (u >= 0) { pi[cnt] = u; po[cnt] = (int32_t)v; ++cnt; }
    }
    counts[k] = cnt;
  }
}

/* ---- dense block kernels -------------------------------------------------------------------------------------- */
typedef float v8 __attribute__((vector_size(32), aligned(4), may_alias));
#define BLK 32      /* pairs per block of the forward / input-gradient pass */
#define BLKW 128    /* pairs per block of the weight gradient (the dW tile is re-read once per block) */

/* C[nr][n] = A[nr][kdim] * B[kdim][n]  (row-major, A and C contiguous with their own leading dimension) */
static void block_gemm(const float* A, int nr, int kdim, const float* B, int n, float* C) {
  int j0 = 0;
  for (; j0 + 16 <= n; j0 += 16) {
    int r0 = 0;
    for (; r0 + 4 <= nr; r0 += 4) {
      v8 a00 = {0}, a01 = {0}, a10 = {0}, a11 = {0}, a20 = {0}, a21 = {0}, a30 = {0}, a31 = {0};
      const float* A0 = A + (size_t)r0 * kdim;
      for (int c = 0; c < kdim; ++c) {
        const v8 w0 = *(const v8*)(B + (size_t)c * n + j0), w1 = *(const v8*)(B + (size_t)c * n + j0 + 8);
        const float x0 = A0[c], x1 = A0[kdim + c], x2 = A0[2 * kdim + c], x3 = A0[3 * kdim + c];
        a00 += x0 * w0; a01 += x0 * w1; a10 += x1 * w0; a11 += x1 * w1;
        a20 += x2 * w0; a21 += x2 * w1; a30 += x3 * w0; a31 += x3 * w1;
      }
      *(v8*)(C + (size_t)(r0 + 0) * n + j0) = a00; *(v8*)(C + (size_t)(r0 + 0) * n + j0 + 8) = a01;
      *(v8*)(C + (size_t)(r0 + 1) * n + j0) = a10; *(v8*)(C + (size_t)(r0 + 1) * n + j0 + 8) = a11;
      *(v8*)(C + (size_t)(r0 + 2) * n + j0) = a20; *(v8*)(C + (size_t)(r0 + 2) * n + j0 + 8) = a21;
      *(v8*)(C + (size_t)(r0 + 3) * n + j0) = a30; *(v8*)(C + (size_t)(r0 + 3) * n + j0 + 8) = a31;
    }
    for (; r0 < nr; ++r0) {
      v8 a0 = {0}, a1 = {0};
      for (int c = 0; c < kdim; ++c) {
        const float x = A[(size_t)r0 * kdim + c];
        a0 += x * *(const v8*)(B + (size_t)c * n + j0);
        a1 += x * *(const v8*)(B + (size_t)c * n + j0 + 8);
      }
      *(v8*)(C + (size_t)r0 * n + j0) = a0; *(v8*)(C + (size_t)r0 * n + j0 + 8) = a1;
    }
  }
  for (; j0 < n; ++j0)
    for (int r0 = 0; r0 < nr; ++r0) {
      float s = 0.f;
      for (int c = 0; c < kdim; ++c) s += A[(size_t)r0 * kdim + c] * B[(size_t)c * n + j0];
      C[(size_t)r0 * n + j0] = s;
    }
}

/* D[m][n] += A[nr][m]^T * B[nr][n] */
static void block_gemm_tn(const float* A, int nr, int m, const float* B, int n, float* D) {
  int j0 = 0;
  for (; j0 + 16 <= n; j0 += 16) {
    int c0 = 0;
    for (; c0 + 4 <= m; c0 += 4) {
      v8 a00 = {0}, a01 = {0}, a10 = {0}, a11 = {0}, a20 = {0}, a21 = {0}, a30 = {0}, a31 = {0};
      for (int r = 0; r < nr; ++r) {
        const v8 w0 = *(const v8*)(B + (size_t)r * n + j0), w1 = *(const v8*)(B + (size_t)r * n + j0 + 8);
        const float* a = A + (size_t)r * m + c0;
        a00 += a[0] * w0; a01 += a[0] * w1; a10 += a[1] * w0; a11 += a[1] * w1;
        a20 += a[2] * w0; a21 += a[2] * w1; a30 += a[3] * w0; a31 += a[3] * w1;
      }
      float* d = D + (size_t)c0 * n + j0;
      *(v8*)(d) += a00; *(v8*)(d + 8) += a01; *(v8*)(d + n) += a10; *(v8*)(d + n + 8) += a11;
      *(v8*)(d + 2 * (size_t)n) += a20; *(v8*)(d + 2 * (size_t)n + 8) += a21;
      *(v8*)(d + 3 * (size_t)n) += a30; *(v8*)(d + 3 * (size_t)n + 8) += a31;
    }
    for (; c0 < m; ++c0) {
      v8 a0 = {0}, a1 = {0};
      for (int r = 0; r < nr; ++r) {
        const float x = A[(size_t)r * m + c0];
        a0 += x * *(const v8*)(B + (size_t)r * n + j0);
        a1 += x * *(const v8*)(B + (size_t)r * n + j0 + 8);
      }
      *(v8*)(D + (size_t)c0 * n + j0) += a0; *(v8*)(D + (size_t)c0 * n + j0 + 8) += a1;
    }
  }
  for (; j0 < n; ++j0)
    for (int c0 = 0; c0 < m; ++c0) {
      float s = 0.f;
      for (int r = 0; r < nr; ++r) s += A[(size_t)r * m + c0] * B[(size_t)r * n + j0];
      D[(size_t)c0 * n + j0] += s;
    }
}

/* y[dst] += x[src] * W_k for every offset (y zero-initialised by the caller).  `seg` = row stride between the per-offset
 * index segments.  Inside one offset every destination row occurs at most once, so the scatter-add of a parallel loop
 * over that offset's pairs is race free; offsets run one after the other (as ME's CPU path does). */
void me_conv_apply(const float* x, int cin, const float* W, int cout, int K, const int32_t* src_rows,
                   const int32_t* dst_rows, const int64_t* counts, int64_t seg, float* y) {
#pragma omp parallel
  {
    float* A = (float*)malloc(sizeof(float) * BLK * (size_t)cin);
    float* C = (float*)malloc(sizeof(float) * BLK * (size_t)cout);
    for (int k = 0; k < K; ++k) {
      const int32_t* ps = src_rows + (int64_t)k * seg;
      const int32_t* pd = dst_rows + (int64_t)k * seg;
      const int64_t nk = counts[k];
      const float* Wk = W + (size_t)k * cin * cout;
#pragma omp for schedule(static)
      for (int64_t b0 = 0; b0 < nk; b0 += BLK) {
        const int nr = (int)((nk - b0 < BLK) ? nk - b0 : BLK);
        for (int r = 0; r < nr; ++r) memcpy(A + (size_t)r * cin, x + (size_t)ps[b0 + r] * cin, sizeof(float) * cin);
        block_gemm(A, nr, cin, Wk, cout, C);
        for (int r = 0; r < nr; ++r) {
          float* yr = y + (size_t)pd[b0 + r] * cout;
          const float* cr = C + (size_t)r * cout;
          for (int j = 0; j < cout; ++j) yr[j] += cr[j];
        }
      }   /* implicit barrier: the next offset may touch the same destination rows */
    }
    free(A);
    free(C);
  }
}

/* dW[k] = sum over the pairs of offset k of x[src]^T dy[dst]  (dW zero-initialised by the caller) */
void me_conv_grad_weight(const float* x, int cin, const float* dy, int cout, int K, const int32_t* src_rows,
                         const int32_t* dst_rows, const int64_t* counts, int64_t seg, float* dW) {
  const int nt = omp_get_max_threads();
  float* part = (float*)calloc((size_t)nt * cin * cout, sizeof(float));
#pragma omp parallel
  {
    float* A = (float*)malloc(sizeof(float) * BLKW * (size_t)cin);
    float* B = (float*)malloc(sizeof(float) * BLKW * (size_t)cout);
    float* mine = part + (size_t)omp_get_thread_num() * cin * cout;
    for (int k = 0; k < K; ++k) {
      const int32_t* ps = src_rows + (int64_t)k * seg;
      const int32_t* pd = dst_rows + (int64_t)k * seg;
      const int64_t nk = counts[k];
      memset(mine, 0, sizeof(float) * (size_t)cin * cout);
#pragma omp for schedule(static)
      for (int64_t b0 = 0; b0 < nk; b0 += BLKW) {
        const int nr = (int)((nk - b0 < BLKW) ? nk - b0 : BLKW);
        for (int r = 0; r < nr; ++r) {
          memcpy(A + (size_t)r * cin, x + (size_t)ps[b0 + r] * cin, sizeof(float) * cin);
          memcpy(B + (size_t)r * cout, dy + (size_t)pd[b0 + r] * cout, sizeof(float) * cout);
        }
        block_gemm_tn(A, nr, cin, B, cout, mine);
      }
      float* dWk = dW + (size_t)k * cin * cout;
#pragma omp for schedule(static)
      for (int64_t e = 0; e < (int64_t)cin * cout; ++e) {
        float s = 0.f;
        for (int t = 0; t < nt; ++t) s += part[(size_t)t * cin * cout + e];      /* fixed order: deterministic */
        dWk[e] = s;
      }
    }
    free(A);
    free(B);
  }
  free(part);
}

int me_num_threads(void) { return omp_get_max_threads(); }
void me_set_num_threads(int n) { omp_set_num_threads(n); }
