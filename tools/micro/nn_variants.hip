// Feature 1-NN (gcl_nn_rowmin) inner-loop variants, 5000 x 5000 x 32, against the shipped form (bitwise).
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/nn_variants tools/micro/nn_variants.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int C = 32;

// V0: the shipped form (one A row per thread, B rows broadcast from LDS)
__global__ void __launch_bounds__(256) k_v0(const float* __restrict__ a, int ma, const float* __restrict__ b, int mb, int chunk,
                                            float* __restrict__ out_v, int* __restrict__ out_i) {
  constexpr int TA = 64, TB = 128;
  __shared__ __attribute__((aligned(16))) float bt[TB][C];
  __shared__ float rv[4][TA];
  __shared__ int ri[4][TA];
  const int t = threadIdx.x, ar = t & 63, cg = t >> 6;
  const int arow = blockIdx.x * TA + ar;
  float av[C];
  for (int q = 0; q < C / 4; ++q) {
    float4 v = make_float4(0, 0, 0, 0);
    if (arow < ma) v = reinterpret_cast<const float4*>(a + (long long)arow * C)[q];
    av[4 * q] = v.x; av[4 * q + 1] = v.y; av[4 * q + 2] = v.z; av[4 * q + 3] = v.w;
  }
  float best = INFINITY; int besti = 0;
  const int jb = blockIdx.y * chunk, je = min(jb + chunk, mb);
  for (int j0 = jb; j0 < je; j0 += TB) {
    __syncthreads();
    for (int e = t; e < TB * (C / 4); e += 256) {
      int r = e / (C / 4), q = e % (C / 4);
      float4 v = make_float4(0, 0, 0, 0);
      if (j0 + r < je) v = reinterpret_cast<const float4*>(b + (long long)(j0 + r) * C)[q];
      reinterpret_cast<float4*>(&bt[r][0])[q] = v;
    }
    __syncthreads();
    int jn = min(je - j0, TB);
    for (int r = cg; r < jn; r += 4) {
      float d2 = 0.f;
#pragma unroll
      for (int q = 0; q < C / 4; ++q) {
        float4 v = reinterpret_cast<const float4*>(&bt[r][0])[q];
        float d0 = av[4 * q] - v.x, d1 = av[4 * q + 1] - v.y, d2a = av[4 * q + 2] - v.z, d3 = av[4 * q + 3] - v.w;
        d2 += d0 * d0; d2 += d1 * d1; d2 += d2a * d2a; d2 += d3 * d3;
      }
      if (d2 < best) { best = d2; besti = j0 + r; }
    }
  }
  rv[cg][ar] = best; ri[cg][ar] = besti;
  __syncthreads();
  if (cg == 0 && arow < ma) {
    float bv = rv[0][ar]; int bi = ri[0][ar];
    for (int w = 1; w < 4; ++w) { float v = rv[w][ar]; int i2 = ri[w][ar]; if (v < bv || (v == bv && i2 < bi)) { bv = v; bi = i2; } }
    out_v[(long long)blockIdx.y * ma + arow] = bv; out_i[(long long)blockIdx.y * ma + arow] = bi;
  }
}

// V1: two A rows per thread packed in register pairs, B tile duplicated in LDS ((b, b) pairs), NB B rows per iteration
template <int NB>
__global__ void __launch_bounds__(256) k_v1(const float* __restrict__ a, int ma, const float* __restrict__ b, int mb, int chunk,
                                            float* __restrict__ out_v, int* __restrict__ out_i) {
  constexpr int TA = 128, TB = 64;
  __shared__ __attribute__((aligned(16))) f2 bt[TB][C];
  __shared__ float rv[4][TA];
  __shared__ int ri[4][TA];
  const int t = threadIdx.x, ar = t & 63, cg = t >> 6;
  const int arow0 = blockIdx.x * TA + ar, arow1 = arow0 + 64;
  f2 av[C];
  for (int q = 0; q < C / 4; ++q) {
    float4 v0 = make_float4(0, 0, 0, 0), v1 = v0;
    if (arow0 < ma) v0 = reinterpret_cast<const float4*>(a + (long long)arow0 * C)[q];
    if (arow1 < ma) v1 = reinterpret_cast<const float4*>(a + (long long)arow1 * C)[q];
    av[4 * q] = f2{v0.x, v1.x}; av[4 * q + 1] = f2{v0.y, v1.y}; av[4 * q + 2] = f2{v0.z, v1.z}; av[4 * q + 3] = f2{v0.w, v1.w};
  }
  f2 best = {INFINITY, INFINITY}; int bi0 = 0, bi1 = 0;
  const int jb = blockIdx.y * chunk, je = min(jb + chunk, mb);
  for (int j0 = jb; j0 < je; j0 += TB) {
    __syncthreads();
    for (int e = t; e < TB * (C / 4); e += 256) {
      int r = e / (C / 4), q = e % (C / 4);
      float4 v = make_float4(0, 0, 0, 0);
      if (j0 + r < je) v = reinterpret_cast<const float4*>(b + (long long)(j0 + r) * C)[q];
      reinterpret_cast<f4*>(&bt[r][4 * q])[0] = f4{v.x, v.x, v.y, v.y};
      reinterpret_cast<f4*>(&bt[r][4 * q])[1] = f4{v.z, v.z, v.w, v.w};
    }
    __syncthreads();
    int jn = min(je - j0, TB);
    for (int r = cg * NB; r < jn; r += 4 * NB) {
      f2 acc[NB];
#pragma unroll
      for (int u = 0; u < NB; ++u) acc[u] = f2{0.f, 0.f};
#pragma unroll
      for (int q = 0; q < C / 2; ++q) {
#pragma unroll
        for (int u = 0; u < NB; ++u) {
          f4 v = *reinterpret_cast<const f4*>(&bt[r + u][2 * q]);
          f2 d0 = av[2 * q] - f2{v.x, v.y};
          acc[u] = d0 * d0 + acc[u];
          f2 d1 = av[2 * q + 1] - f2{v.z, v.w};
          acc[u] = d1 * d1 + acc[u];
        }
      }
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        if (r + u < jn) {
          if (acc[u].x < best.x) { best.x = acc[u].x; bi0 = j0 + r + u; }
          if (acc[u].y < best.y) { best.y = acc[u].y; bi1 = j0 + r + u; }
        }
      }
    }
  }
  rv[cg][ar] = best.x; ri[cg][ar] = bi0; rv[cg][ar + 64] = best.y; ri[cg][ar + 64] = bi1;
  __syncthreads();
  if (t < TA && blockIdx.x * TA + t < ma) {
    float bv = rv[0][t]; int bi = ri[0][t];
    for (int w = 1; w < 4; ++w) { float v = rv[w][t]; int i2 = ri[w][t]; if (v < bv || (v == bv && i2 < bi)) { bv = v; bi = i2; } }
    out_v[(long long)blockIdx.y * ma + blockIdx.x * TA + t] = bv; out_i[(long long)blockIdx.y * ma + blockIdx.x * TA + t] = bi;
  }
}

// V3: V1 with the next B tile's rows fetched into registers before the current tile is computed
template <int NB, int TB>
__global__ void __launch_bounds__(256) k_v3(const float* __restrict__ a, int ma, const float* __restrict__ b, int mb, int chunk,
                                            float* __restrict__ out_v, int* __restrict__ out_i) {
  constexpr int TA = 128, PF = TB * (C / 4) / 256;
  __shared__ __attribute__((aligned(16))) f2 bt[TB][C];
  __shared__ float rv[4][TA];
  __shared__ int ri[4][TA];
  const int t = threadIdx.x, ar = t & 63, cg = t >> 6;
  const int arow0 = blockIdx.x * TA + ar, arow1 = arow0 + 64;
  f2 av[C];
  for (int q = 0; q < C / 4; ++q) {
    float4 v0 = make_float4(0, 0, 0, 0), v1 = v0;
    if (arow0 < ma) v0 = reinterpret_cast<const float4*>(a + (long long)arow0 * C)[q];
    if (arow1 < ma) v1 = reinterpret_cast<const float4*>(a + (long long)arow1 * C)[q];
    av[4 * q] = f2{v0.x, v1.x}; av[4 * q + 1] = f2{v0.y, v1.y}; av[4 * q + 2] = f2{v0.z, v1.z}; av[4 * q + 3] = f2{v0.w, v1.w};
  }
  f2 best = {INFINITY, INFINITY}; int bi0 = 0, bi1 = 0;
  const int jb = blockIdx.y * chunk, je = min(jb + chunk, mb);
  float4 pf[PF];
  auto fetch = [&](int j0) {
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int e = t + 256 * u, r = e / (C / 4), q = e % (C / 4);
      pf[u] = make_float4(0, 0, 0, 0);
      if (j0 + r < je) pf[u] = reinterpret_cast<const float4*>(b + (long long)(j0 + r) * C)[q];
    }
  };
  fetch(jb);
  for (int j0 = jb; j0 < je; j0 += TB) {
    __syncthreads();
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int e = t + 256 * u, r = e / (C / 4), q = e % (C / 4);
      const float4 v = pf[u];
      reinterpret_cast<f4*>(&bt[r][4 * q])[0] = f4{v.x, v.x, v.y, v.y};
      reinterpret_cast<f4*>(&bt[r][4 * q])[1] = f4{v.z, v.z, v.w, v.w};
    }
    __syncthreads();
    if (j0 + TB < je) fetch(j0 + TB);
    int jn = min(je - j0, TB);
    for (int r = cg * NB; r < jn; r += 4 * NB) {
      f2 acc[NB];
#pragma unroll
      for (int u = 0; u < NB; ++u) acc[u] = f2{0.f, 0.f};
#pragma unroll
      for (int q = 0; q < C / 2; ++q) {
#pragma unroll
        for (int u = 0; u < NB; ++u) {
          f4 v = *reinterpret_cast<const f4*>(&bt[r + u][2 * q]);
          f2 d0 = av[2 * q] - f2{v.x, v.y};
          acc[u] = d0 * d0 + acc[u];
          f2 d1 = av[2 * q + 1] - f2{v.z, v.w};
          acc[u] = d1 * d1 + acc[u];
        }
      }
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        if (r + u < jn) {
          if (acc[u].x < best.x) { best.x = acc[u].x; bi0 = j0 + r + u; }
          if (acc[u].y < best.y) { best.y = acc[u].y; bi1 = j0 + r + u; }
        }
      }
    }
  }
  rv[cg][ar] = best.x; ri[cg][ar] = bi0; rv[cg][ar + 64] = best.y; ri[cg][ar + 64] = bi1;
  __syncthreads();
  if (t < TA && blockIdx.x * TA + t < ma) {
    float bv = rv[0][t]; int bi = ri[0][t];
    for (int w = 1; w < 4; ++w) { float v = rv[w][t]; int i2 = ri[w][t]; if (v < bv || (v == bv && i2 < bi)) { bv = v; bi = i2; } }
    out_v[(long long)blockIdx.y * ma + blockIdx.x * TA + t] = bv; out_i[(long long)blockIdx.y * ma + blockIdx.x * TA + t] = bi;
  }
}

// V2: NA A rows per thread (each splat into pairs), two B rows per register pair (LDS tile interleaved [r/2][c][2])
template <int NA>
__global__ void __launch_bounds__(256) k_v2(const float* __restrict__ a, int ma, const float* __restrict__ b, int mb, int chunk,
                                            float* __restrict__ out_v, int* __restrict__ out_i) {
  constexpr int TA = 64 * NA, TB = 128;
  __shared__ __attribute__((aligned(16))) f2 bt[TB / 2][C];
  __shared__ float rv[4][TA];
  __shared__ int ri[4][TA];
  const int t = threadIdx.x, ar = t & 63, cg = t >> 6;
  f2 av[NA][C];
#pragma unroll
  for (int u = 0; u < NA; ++u) {
    const int arow = blockIdx.x * TA + ar + 64 * u;
    for (int q = 0; q < C / 4; ++q) {
      float4 v = make_float4(0, 0, 0, 0);
      if (arow < ma) v = reinterpret_cast<const float4*>(a + (long long)arow * C)[q];
      av[u][4 * q] = f2{v.x, v.x}; av[u][4 * q + 1] = f2{v.y, v.y}; av[u][4 * q + 2] = f2{v.z, v.z}; av[u][4 * q + 3] = f2{v.w, v.w};
    }
  }
  float best[NA]; int besti[NA];
#pragma unroll
  for (int u = 0; u < NA; ++u) { best[u] = INFINITY; besti[u] = 0; }
  const int jb = blockIdx.y * chunk, je = min(jb + chunk, mb);
  for (int j0 = jb; j0 < je; j0 += TB) {
    __syncthreads();
    for (int e = t; e < TB * (C / 4); e += 256) {
      int r = e / (C / 4), q = e % (C / 4);
      float4 v = make_float4(0, 0, 0, 0);
      if (j0 + r < je) v = reinterpret_cast<const float4*>(b + (long long)(j0 + r) * C)[q];
      float* dst = reinterpret_cast<float*>(&bt[r >> 1][4 * q]) + (r & 1);
      dst[0] = v.x; dst[2] = v.y; dst[4] = v.z; dst[6] = v.w;
    }
    __syncthreads();
    int jn = min(je - j0, TB);
    for (int p = cg; 2 * p < jn; p += 4) {
      f2 acc[NA];
#pragma unroll
      for (int u = 0; u < NA; ++u) acc[u] = f2{0.f, 0.f};
#pragma unroll
      for (int q = 0; q < C / 2; ++q) {
        f4 v = *reinterpret_cast<const f4*>(&bt[p][2 * q]);
#pragma unroll
        for (int u = 0; u < NA; ++u) {
          f2 d0 = av[u][2 * q] - f2{v.x, v.y};
          acc[u] = d0 * d0 + acc[u];
          f2 d1 = av[u][2 * q + 1] - f2{v.z, v.w};
          acc[u] = d1 * d1 + acc[u];
        }
      }
#pragma unroll
      for (int u = 0; u < NA; ++u) {
        if (acc[u].x < best[u]) { best[u] = acc[u].x; besti[u] = j0 + 2 * p; }
        if (2 * p + 1 < jn && acc[u].y < best[u]) { best[u] = acc[u].y; besti[u] = j0 + 2 * p + 1; }
      }
    }
  }
#pragma unroll
  for (int u = 0; u < NA; ++u) { rv[cg][ar + 64 * u] = best[u]; ri[cg][ar + 64 * u] = besti[u]; }
  __syncthreads();
  if (t < TA && blockIdx.x * TA + t < ma) {
    float bv = rv[0][t]; int bi = ri[0][t];
    for (int w = 1; w < 4; ++w) { float v = rv[w][t]; int i2 = ri[w][t]; if (v < bv || (v == bv && i2 < bi)) { bv = v; bi = i2; } }
    out_v[(long long)blockIdx.y * ma + blockIdx.x * TA + t] = bv; out_i[(long long)blockIdx.y * ma + blockIdx.x * TA + t] = bi;
  }
}

// V4: B row pairs from scalar loads (B interleaved in memory as [r / 2][c][2]): no LDS in the loop
__global__ void k_interleave(const float* __restrict__ b, int mb, float* __restrict__ bi) {
  const int e = blockIdx.x * 256 + threadIdx.x;       // one (row pair, channel)
  const int p = e / C, c = e % C;
  if (2 * p >= mb) return;
  bi[2 * (size_t)e] = b[(size_t)(2 * p) * C + c];
  bi[2 * (size_t)e + 1] = (2 * p + 1 < mb) ? b[(size_t)(2 * p + 1) * C + c] : 0.f;
}
template <int NA>
__global__ void __launch_bounds__(256) k_v4(const float* __restrict__ a, int ma, const float* __restrict__ bi_, int mb, int chunk,
                                            float* __restrict__ out_v, int* __restrict__ out_i) {
  constexpr int TA = 64 * NA;
  __shared__ float rv[4][TA];
  __shared__ int ri[4][TA];
  const f2* __restrict__ bi = reinterpret_cast<const f2*>(bi_);
  const int t = threadIdx.x, ar = t & 63, cg = __builtin_amdgcn_readfirstlane(t >> 6);
  f2 av[NA][C];
#pragma unroll
  for (int u = 0; u < NA; ++u) {
    const int arow = blockIdx.x * TA + ar + 64 * u;
    for (int q = 0; q < C / 4; ++q) {
      float4 v = make_float4(0, 0, 0, 0);
      if (arow < ma) v = reinterpret_cast<const float4*>(a + (long long)arow * C)[q];
      av[u][4 * q] = f2{v.x, v.x}; av[u][4 * q + 1] = f2{v.y, v.y}; av[u][4 * q + 2] = f2{v.z, v.z}; av[u][4 * q + 3] = f2{v.w, v.w};
    }
  }
  float best[NA]; int besti[NA];
#pragma unroll
  for (int u = 0; u < NA; ++u) { best[u] = INFINITY; besti[u] = 0; }
  const int jb = blockIdx.y * chunk, je = min(jb + chunk, mb);      // chunk even
  for (int p = jb / 2 + cg; 2 * p < je; p += 4) {
    const f2* __restrict__ row = bi + (size_t)p * C;
    f2 acc[NA];
#pragma unroll
    for (int u = 0; u < NA; ++u) acc[u] = f2{0.f, 0.f};
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const f2 v = row[c];
#pragma unroll
      for (int u = 0; u < NA; ++u) {
        f2 d = av[u][c] - v;
        acc[u] = d * d + acc[u];
      }
    }
#pragma unroll
    for (int u = 0; u < NA; ++u) {
      if (acc[u].x < best[u]) { best[u] = acc[u].x; besti[u] = 2 * p; }
      if (2 * p + 1 < je && acc[u].y < best[u]) { best[u] = acc[u].y; besti[u] = 2 * p + 1; }
    }
  }
#pragma unroll
  for (int u = 0; u < NA; ++u) { rv[cg][ar + 64 * u] = best[u]; ri[cg][ar + 64 * u] = besti[u]; }
  __syncthreads();
  if (t < TA && blockIdx.x * TA + t < ma) {
    float bv = rv[0][t]; int bi2 = ri[0][t];
    for (int w = 1; w < 4; ++w) { float v = rv[w][t]; int i2 = ri[w][t]; if (v < bv || (v == bv && i2 < bi2)) { bv = v; bi2 = i2; } }
    out_v[(long long)blockIdx.y * ma + blockIdx.x * TA + t] = bv; out_i[(long long)blockIdx.y * ma + blockIdx.x * TA + t] = bi2;
  }
}

static int chunk_rows(int ma, int mb, int ta, int tb, int want_wgs) {
  long long a_tiles = (ma + ta - 1) / ta, want = (want_wgs + a_tiles - 1) / a_tiles, tiles_b = (mb + tb - 1) / tb;
  if (want > tiles_b) want = tiles_b;
  if (want < 1) want = 1;
  return (int)(((tiles_b + want - 1) / want) * tb);
}

template <typename K>
static float run(K kern, int ta, int tb, int want, const float* a, int ma, const float* b, int mb, float* pv, int* pi,
                 std::vector<float>& hv, std::vector<int>& hi) {
  int chunk = chunk_rows(ma, mb, ta, tb, want), nch = (mb + chunk - 1) / chunk;
  dim3 grid((ma + ta - 1) / ta, nch);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, grid, dim3(256), 0, 0, a, ma, b, mb, chunk, pv, pi);
  hipEventRecord(e0);
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(kern, grid, dim3(256), 0, 0, a, ma, b, mb, chunk, pv, pi);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<float> v((size_t)nch * ma); std::vector<int> ix((size_t)nch * ma);
  hipMemcpy(v.data(), pv, v.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(ix.data(), pi, ix.size() * 4, hipMemcpyDeviceToHost);
  hv.assign(ma, 0); hi.assign(ma, 0);
  for (int r = 0; r < ma; ++r) {
    float bv = v[r]; int bi = ix[r];
    for (int c = 1; c < nch; ++c) if (v[(size_t)c * ma + r] < bv) { bv = v[(size_t)c * ma + r]; bi = ix[(size_t)c * ma + r]; }
    hv[r] = bv; hi[r] = bi;
  }
  printf("  grid %d x %d (chunk %d): %.1f us\n", grid.x, grid.y, chunk, ms * 1000 / 20);
  return ms * 1000 / 20;
}

int main(int argc, char** argv) {
  int ma = argc > 1 ? atoi(argv[1]) : 5000, mb = argc > 2 ? atoi(argv[2]) : 5000;
  std::vector<float> ha((size_t)ma * C), hb((size_t)mb * C);
  srand(1);
  auto fill = [](std::vector<float>& x) {
    for (size_t r = 0; r < x.size() / C; ++r) { double s = 0; for (int c = 0; c < C; ++c) { x[r * C + c] = rand() / (float)RAND_MAX - 0.5f; s += x[r * C + c] * x[r * C + c]; }
      for (int c = 0; c < C; ++c) x[r * C + c] /= (float)sqrt(s); } };
  fill(ha); fill(hb);
  for (int r = 0; r < 50 && r < ma && r + 7 < mb; ++r) for (int c = 0; c < C; ++c) hb[(size_t)(r + 7) * C + c] = hb[(size_t)r * C + c];   // exact ties
  float *a, *b, *pv; int* pi;
  hipMalloc(&a, ha.size() * 4); hipMalloc(&b, hb.size() * 4); hipMalloc(&pv, (size_t)64 * ma * 4 + 4096); hipMalloc(&pi, (size_t)64 * ma * 4 + 4096);
  hipMemcpy(a, ha.data(), ha.size() * 4, hipMemcpyHostToDevice); hipMemcpy(b, hb.data(), hb.size() * 4, hipMemcpyHostToDevice);
  std::vector<float> v0, v; std::vector<int> i0, ix;
  auto cmp = [&](const char* name) { size_t bad = 0; for (int r = 0; r < ma; ++r) bad += (v[r] != v0[r]) || (ix[r] != i0[r]); printf("  %s vs shipped: %zu rows differ\n", name, bad); };
  printf("V0 shipped\n"); run(k_v0, 64, 128, 1024, a, ma, b, mb, pv, pi, v0, i0);
  for (int want : {1536}) {
    printf("want %d workgroups\n", want);
    printf(" V1<1>\n"); run(k_v1<1>, 128, 64, want, a, ma, b, mb, pv, pi, v, ix); cmp("V1<1>");
    printf(" V1<2>\n"); run(k_v1<2>, 128, 64, want, a, ma, b, mb, pv, pi, v, ix); cmp("V1<2>");
    printf(" V2<1>\n"); run(k_v2<1>, 64, 128, want, a, ma, b, mb, pv, pi, v, ix); cmp("V2<1>");
    printf(" V2<2>\n"); run(k_v2<2>, 128, 128, want, a, ma, b, mb, pv, pi, v, ix); cmp("V2<2>");
    std::vector<float> v1 = v; std::vector<int> i1 = ix;
    printf(" V3<2,64>\n"); run(k_v3<2, 64>, 128, 64, want, a, ma, b, mb, pv, pi, v, ix); cmp("V3<2,64>");
    { size_t bad = 0; for (int r = 0; r < ma; ++r) bad += (v[r] != v1[r]) || (ix[r] != i1[r]); printf("  V3 vs V2<2>: %zu rows differ\n", bad); }
    printf(" V3<2,128>\n"); run(k_v3<2, 128>, 128, 128, want, a, ma, b, mb, pv, pi, v, ix); cmp("V3<2,128>");
    { float* bil; hipMalloc(&bil, ((size_t)mb + 2) * C * 4);
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      const int ne = ((mb + 1) / 2) * C;
      hipLaunchKernelGGL(k_interleave, dim3((ne + 255) / 256), dim3(256), 0, 0, b, mb, bil);
      hipEventRecord(e0);
      for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k_interleave, dim3((ne + 255) / 256), dim3(256), 0, 0, b, mb, bil);
      hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
      printf(" interleave %.1f us\n", ms * 1000 / 20);
      printf(" V4<1>\n"); run(k_v4<1>, 64, 8, want, a, ma, bil, mb, pv, pi, v, ix); cmp("V4<1>");
      { size_t bad = 0; for (int r = 0; r < ma; ++r) bad += (v[r] != v1[r]) || (ix[r] != i1[r]); printf("  V4<1> vs V2<2>: %zu rows differ\n", bad);
        int shown = 0; for (int r = 0; r < ma && shown < 6; ++r) if ((v[r] != v1[r]) || (ix[r] != i1[r])) { printf("    row %d: V4 (%.9g, %d)  V2 (%.9g, %d)\n", r, v[r], ix[r], v1[r], i1[r]); ++shown; } }
      printf(" V4<2>\n"); run(k_v4<2>, 128, 8, want, a, ma, bil, mb, pv, pi, v, ix); cmp("V4<2>");
      { size_t bad = 0; for (int r = 0; r < ma; ++r) bad += (v[r] != v1[r]) || (ix[r] != i1[r]); printf("  V4<2> vs V2<2>: %zu rows differ\n", bad); }
      hipFree(bil); }
    printf(" V3<4,128>\n"); run(k_v3<4, 128>, 128, 128, want, a, ma, b, mb, pv, pi, v, ix); cmp("V3<4,128>");
  }
  return 0;
}
