// DIAGNOSTIC (round 4): what `buffer_load_dwordx4 ... lds` (LDS-DMA through a buffer resource) does on gfx950 --
//  (1) lane l's 16 bytes land at LDS base + 16 l (wave-uniform base in M0, per-lane SOURCE offset),
//  (2) a lane whose source offset lies beyond the resource's num_records writes ZEROS (the gather's "row -1" trick),
//  (3) the XOR-swizzled source offsets give the image k_conv_fwd_split's A tile has.
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/dma_probe.hip -o tools/micro/dma_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
__global__ void k(const float* X, unsigned bytes, const int* idx, float* out) {
  __shared__ __attribute__((aligned(16))) float tile[4][32][32];
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int e = threadIdx.x; e < 4096; e += 256) (&tile[0][0][0])[e] = -7.f;     // poison: OOB lanes must overwrite it
  __syncthreads();
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)X, 0, (int)bytes, 0x00020000);
  const int p = l & 7, rsub = l >> 3;
#pragma unroll
  for (int ps = 0; ps < 4; ++ps) {
    const int row = idx[w * 32 + rsub + 8 * ps];
    const unsigned off = (unsigned)row * 128u + (unsigned)((p ^ (((rsub + 8 * ps) >> 1) & 7)) * 16);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)&tile[w][8 * ps][0], 16, (int)off, 0,
                                             0, 0);
  }
  __syncthreads();
  for (int e = l; e < 1024; e += 64) out[w * 1024 + e] = (&tile[w][0][0])[e];
}
int main() {
  const int n = 1000;
  std::vector<float> hx((size_t)n * 32);
  for (size_t i = 0; i < hx.size(); ++i) hx[i] = (float)i;
  std::vector<int> hidx(128);
  srand(1);
  for (int i = 0; i < 128; ++i) hidx[i] = (i % 5 == 3) ? -1 : rand() % n;
  float *dx, *dout;
  int* didx;
  hipMalloc(&dx, hx.size() * 4 + 4096);
  hipMemset(dx, 0xff, hx.size() * 4 + 4096);      // bytes past the tensor are NaN patterns: must never show up
  hipMalloc(&dout, 4096 * 4);
  hipMalloc(&didx, 128 * 4);
  hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(didx, hidx.data(), 128 * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, dx, (unsigned)(hx.size() * 4), didx, dout);
  std::vector<float> ho(4096);
  hipMemcpy(ho.data(), dout, 4096 * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int w = 0; w < 4; ++w)
    for (int r = 0; r < 32; ++r)
      for (int q = 0; q < 8; ++q)            // LDS position q of row r holds source piece q ^ swz(r)
        for (int j = 0; j < 4; ++j) {
          const int row = hidx[w * 32 + r], piece = q ^ ((r >> 1) & 7);
          const float want = row < 0 ? 0.f : hx[(size_t)row * 32 + piece * 4 + j];
          const float got = ho[w * 1024 + r * 32 + q * 4 + j];
          if (!(got == want)) {
            if (bad < 8) printf("mismatch w%d r%d q%d j%d: got %g want %g (row %d)\n", w, r, q, j, got, want, row);
            ++bad;
          }
        }
  printf("dma_probe: %s (%d mismatches; OOB rows %s)\n", bad ? "FAIL" : "OK", bad, bad ? "?" : "read zero");
  return bad ? 1 : 0;
}
