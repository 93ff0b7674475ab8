#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split2(float x0, float x1, float s, unsigned& hi, unsigned& lo) {
  f16x2 h;
  h[0] = (_Float16)(x0 * s);
  h[1] = (_Float16)(x1 * s);
  hi = __builtin_bit_cast(unsigned, h);
  unsigned L;
  asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(L) : "v"(x0), "v"(s), "v"(hi));
  asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(L) : "v"(x1), "v"(s), "v"(hi));
  lo = L;
}
__global__ void k(const float* x, float s, unsigned* out, unsigned* ref, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (2 * i + 1 >= n) return;
  float x0 = x[2 * i], x1 = x[2 * i + 1];
  unsigned hi, lo;
  split2(x0, x1, s, hi, lo);
  out[2 * i] = hi; out[2 * i + 1] = lo;
  f16x2 h, l;
  float s0 = x0 * s, s1 = x1 * s;
  h[0] = (_Float16)s0; h[1] = (_Float16)s1;
  l[0] = (_Float16)(s0 - (float)h[0]); l[1] = (_Float16)(s1 - (float)h[1]);
  ref[2 * i] = __builtin_bit_cast(unsigned, h); ref[2 * i + 1] = __builtin_bit_cast(unsigned, l);
}
int main() {
  const int n = 1 << 20;
  float* hx = new float[n];
  srand(1);
  for (int i = 0; i < n; ++i) hx[i] = ((rand() / (float)RAND_MAX) - 0.5f) * powf(2.f, (rand() % 24) - 12);
  float *dx; unsigned *o, *r;
  hipMalloc(&dx, n * 4); hipMalloc(&o, n * 4); hipMalloc(&r, n * 4);
  hipMemcpy(dx, hx, n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 2 / 256), dim3(256), 0, 0, dx, 4096.f, o, r, n);
  unsigned* ho = new unsigned[n]; unsigned* hr = new unsigned[n];
  hipMemcpy(ho, o, n * 4, hipMemcpyDeviceToHost); hipMemcpy(hr, r, n * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < n; ++i) if (ho[i] != hr[i]) { if (bad < 5) printf("mismatch %d: %08x vs %08x (x=%g %g)\n", i, ho[i], hr[i], hx[i & ~1], hx[i | 1]); ++bad; }
  printf("mismatches: %d of %d\n", bad, n);
  return 0;
}
