// DIAGNOSTIC micro-benchmark (not product): cost of end-of-kernel "publish a maximum" patterns on gfx950.
// hipcc --offload-arch=gfx950 -O3 -o /tmp/atomic_contention tools/micro/atomic_contention.hip && /tmp/atomic_contention
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void __launch_bounds__(256) k_pub(int* a, int mode, int slots, int stride, int work) {
  // a little streaming-free work so that workgroups do not all retire in the same cycle
  float m = (float)(blockIdx.x * 2654435761u >> 8) * 1e-3f + threadIdx.x * 1e-6f;
  for (int i = 0; i < work; ++i) m = m * 1.0000001f + 1e-7f;
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  __shared__ float wm[4];
  if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    int bits = __float_as_int(fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3])));
    int* p = a + (blockIdx.x % slots) * stride;
    if (mode == 0) atomicMax(p, bits);
    else if (mode == 1) { if (bits > __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(p, bits); }
    else if (mode == 2) { if (bits > *(volatile int*)p) atomicMax(p, bits); }
    else if (mode == 3) a[4096 + blockIdx.x] = bits;     // plain store, reduced by a second kernel
  }
}
__global__ void k_red(const int* a, int n, int* out) {
  int m = 0;
  for (int i = threadIdx.x; i < n; i += 256) m = max(m, a[4096 + i]);
  for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o));
  __shared__ int wm[4];
  if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = max(max(wm[0], wm[1]), max(wm[2], wm[3]));
}

int main() {
  int* a;
  hipMalloc(&a, 1 << 20);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const char* names[] = {"atomicMax", "agent-load check + atomicMax", "volatile-load check + atomicMax", "store + reduce kernel"};
  for (int grid : {512, 2048, 4096})
    for (int mode = 0; mode < 4; ++mode)
      for (int cfg = 0; cfg < 5; ++cfg) {
        int slots = cfg == 0 ? 1 : (cfg < 3 ? 16 : (cfg == 3 ? 64 : 256)), stride = cfg >= 2 ? 32 : 1;
        if (mode == 3 && cfg) continue;
        float best = 1e9f;
        for (int rep = 0; rep < 20; ++rep) {
          hipMemsetAsync(a, 0, 1 << 20, 0);
          hipEventRecord(e0, 0);
          hipLaunchKernelGGL(k_pub, dim3(grid), dim3(256), 0, 0, a, mode, slots, stride, 0);
          if (mode == 3) hipLaunchKernelGGL(k_red, dim3(1), dim3(256), 0, 0, a, grid, a);
          hipEventRecord(e1, 0);
          hipEventSynchronize(e1);
          float ms;
          hipEventElapsedTime(&ms, e0, e1);
          if (rep > 2 && ms < best) best = ms;
        }
        printf("grid %4d  %-34s slots %2d stride %2d : %7.1f us\n", grid, names[mode], slots, stride, best * 1e3f);
      }
  // baseline: no publish at all
  return 0;
}
