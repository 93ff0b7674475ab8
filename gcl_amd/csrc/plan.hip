// Native step runtime: one C call enqueues a whole pass (include/gcl_amd.h, "Native step runtime").
//
//   gcl_maps_build     = what CoordinateManager.__init__ / _build_stride_maps / get_kernel_map / KernelMap.sorted_table /
//                        KernelMap.pairs do from Python (gcl_amd/MinkowskiEngine/core.py), for all maps of a network;
//   gcl_plan_forward / = what ops.Tape records and replays (gcl_amd/MinkowskiEngine/ops.py: _SparseConvFn, _BatchNormFn,
//   gcl_plan_backward    Tape.backward) for a network given as operator records.
//
// Reference path: model/resunet.py:173-232 (ResUNet2.forward), model/residual_block.py:37-53, and the coordinate manager
// ME builds behind lib/colocation_trainer.py:843-845.  Both families call the SAME extern "C" entries the per-operator
// path calls, with the same arguments in the same order, so results are bitwise identical (tests/test_gpu_plan.py); what
// disappears is ~560 Python -> ctypes round trips (~40 us each) per training step.
// Memory: bump allocation from the caller's arena, nothing is freed inside a pass (288 GB of HBM: a 0.5 M-voxel step
// uses a few GB); a dry run of the same code sizes the arena.
#include "common.h"

#include <string.h>

#include <new>
#include <utility>
#include <vector>

namespace gcl {

// ---------------------------------------------------------------------------------------------------
// small elementwise kernels of the plan path
// ---------------------------------------------------------------------------------------------------
// max |v| of what a workgroup wrote -> the tensor's amax slot (same value gcl_amax would measure in a pass of its own)
__device__ __forceinline__ float amax4p(float m, const float4& v) {
  return fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
}
__device__ __forceinline__ void publish_amax_wg(float m, int* slot) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  __shared__ float wm[4];
  if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0)
    amax_slot_publish(slot, __float_as_int(fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]))), blockIdx.x);
}

// MEF.relu: torch.relu semantics (NaN propagates); amax (optional): zero-initialised slot that receives max|y|
__global__ void __launch_bounds__(256) k_relu_fwd(const float4* __restrict__ x, long long n4, float4* __restrict__ y,
                                                  int* amax) {
  float m = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    float4 v = x[i];
    v.x = v.x < 0.f ? 0.f : v.x;
    v.y = v.y < 0.f ? 0.f : v.y;
    v.z = v.z < 0.f ? 0.f : v.z;
    v.w = v.w < 0.f ? 0.f : v.w;
    y[i] = v;
    m = amax4p(m, v);
  }
  if (amax) publish_amax_wg(m, amax);
}
// aten::threshold_backward(g, y, 0): g where y > 0, else 0
__global__ void __launch_bounds__(256) k_relu_bwd(const float4* __restrict__ g, const float4* __restrict__ y, long long n4,
                                                  float4* __restrict__ gx, int* amax) {
  float m = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    float4 a = g[i];
    const float4 b = y[i];
    a.x = b.x > 0.f ? a.x : 0.f;
    a.y = b.y > 0.f ? a.y : 0.f;
    a.z = b.z > 0.f ? a.z : 0.f;
    a.w = b.w > 0.f ? a.w : 0.f;
    gx[i] = a;
    m = amax4p(m, a);
  }
  if (amax) publish_amax_wg(m, amax);
}
// ME.cat(a, b): y[r] = a[r] | b[r]   (channel counts are multiples of 4)
// sa / sb (both or neither): the inputs' amax slots -- the cat's slot becomes their elementwise maximum (a valid slot of
// max(value a, value b); was a launch of its own, k_slot_max) and nothing is measured
__global__ void __launch_bounds__(256) k_cat2(const float4* __restrict__ a, int ca4, const float4* __restrict__ b, int cb4,
                                              long long n, float4* __restrict__ y, int* amax, const int* __restrict__ sa,
                                              const int* __restrict__ sb) {
  if (sa && blockIdx.x == 0)
    for (int w = threadIdx.x; w < AMAX_WORDS; w += 256) amax[w] = max(sa[w], sb[w]);
  if (sa) amax = nullptr;
  const int c4 = ca4 + cb4;
  const long long total = n * c4;
  float m = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long r = i / c4;
    const int q = (int)(i - r * c4);
    const float4 v = q < ca4 ? a[r * ca4 + q] : b[r * cb4 + (q - ca4)];
    y[i] = v;
    m = amax4p(m, v);
  }
  if (amax) publish_amax_wg(m, amax);
}
// ME.cat whose left input was written in place by its producer: copy the right input's columns, y[r][ca..] = b[r]
// planes (round 5): the cat's plane image gets these columns too, at the scale of `amax` -- then a slot that already holds a
// bound of the WHOLE cat (the left input's BatchNorm statistics launch folded this input's maximum in): nothing is published
__global__ void __launch_bounds__(256) k_cat_right(const float4* __restrict__ b, int cb4, long long n, int ca4,
                                                   float4* __restrict__ y, int* amax, unsigned short* __restrict__ planes) {
  const int c4 = ca4 + cb4;
  const long long total = n * cb4;
  const float pscale = planes ? amax_scale(amax) : 1.f;
  float m = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long r = i / cb4;
    const int q = (int)(i - r * cb4);
    const float4 v = b[i];
    y[r * c4 + ca4 + q] = v;
    if (planes) store_planes4(planes, r, (long long)c4 * 4, (ca4 + q) * 4, v, pscale);
    m = amax4p(m, v);
  }
  if (amax && !planes) publish_amax_wg(m, amax);
}
__global__ void __launch_bounds__(256) k_split2(const float4* __restrict__ g, int ca4, int cb4, long long n,
                                                float4* __restrict__ ga, float4* __restrict__ gb) {
  const int c4 = ca4 + cb4;
  const long long total = n * c4;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long r = i / c4;
    const int q = (int)(i - r * c4);
    const float4 v = g[i];
    if (q < ca4) ga[r * ca4 + q] = v; else gb[r * cb4 + (q - ca4)] = v;
  }
}
// y = a + b for row-major [n, c] operands with row pitches (floats; a column slice of a wider tensor has pitch > c);
// b == NULL: y = a (materialises a slice)
__global__ void __launch_bounds__(256) k_add2_ld(const float* __restrict__ a, int a_ld, const float* __restrict__ b, int b_ld,
                                                 long long n, int c4, float4* __restrict__ y) {
  const long long total = n * c4;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long r = i / c4;
    const int q = (int)(i - r * c4);
    float4 u = reinterpret_cast<const float4*>(a + r * a_ld)[q];
    if (b) {
      const float4 v = reinterpret_cast<const float4*>(b + r * b_ld)[q];
      u = make_float4(u.x + v.x, u.y + v.y, u.z + v.z, u.w + v.w);
    }
    y[i] = u;
  }
}
__global__ void __launch_bounds__(256) k_add2(const float4* __restrict__ a, const float4* __restrict__ b, long long n4,
                                              float4* __restrict__ y) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const float4 u = a[i], v = b[i];
    y[i] = make_float4(u.x + v.x, u.y + v.y, u.z + v.z, u.w + v.w);
  }
}
// identity pair list of a kernel_size-1 convolution: p[i] = i for i < n, -1 padding (CoordinateManager.identity_pairs)
__global__ void __launch_bounds__(256) k_identity_pairs(int* __restrict__ p, long long n, long long total) {
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < total) p[i] = i < n ? (int)i : -1;
}

// coords.hip (internal entry points shared by the two translation units)
void fill32(void* p, long long n_words, unsigned v, hipStream_t st);
void tables_fill(int64_t* tables, long long slots, int32_t* zero4, hipStream_t st);
int coords_insert_impl(const int32_t* coords, int64_t n, int64_t* table, int64_t cap, int32_t* status, bool table_ready,
                       void* stream);
int stride_map_impl(const int32_t* coords_in, int64_t n_in, const int32_t* n_in_dev, int32_t t_out, int64_t* table_out,
                    int64_t cap_out, int32_t* scratch, int32_t* coords_out, int32_t* n_out_dev, int32_t* status,
                    int64_t* index_out, bool table_ready, void* stream);

static inline unsigned grid_for(long long items, unsigned cap = 4096) {
  long long g = cdiv(items > 0 ? items : 1, 256);
  return (unsigned)(g > cap ? cap : g);
}

// ---------------------------------------------------------------------------------------------------
// arena
// ---------------------------------------------------------------------------------------------------
struct Arena {
  char* base;
  long long size, off;
  bool dry;          // size the pass only: no launch, pointers are never dereferenced
  void* take(long long bytes) {
    off = (off + 255) & ~255ll;
    char* p = base + off;
    off += bytes > 0 ? bytes : 0;
    return p;
  }
  template <typename T>
  T* take_n(long long count) { return (T*)take(count * (long long)sizeof(T)); }
  bool fits() const { return dry || off <= size; }
};

static char* const DRY_BASE = (char*)0x100000;    // dry runs hand out fake non-null addresses that are never dereferenced

static long long pow2_cap(long long n) {
  long long cap = 64;
  while (cap < 2 * n) cap *= 2;
  return cap;
}

#define PLAN_CALL(expr)          \
  do {                           \
    if (!A.dry) {                \
      int rc_ = (expr);          \
      if (rc_ != GCL_OK) return rc_; \
    }                            \
  } while (0)

// ---------------------------------------------------------------------------------------------------
// maps
// ---------------------------------------------------------------------------------------------------
static int level_of(int t) {
  int l = 0;
  while ((1 << l) < t) ++l;
  return ((1 << l) == t) ? l : -1;
}

static int maps_build(const int32_t* coords, long long n, const gcl_map_spec* specs, int n_specs, int n_levels, Arena& A,
                      int32_t* pinned, gcl_maps_desc* out, hipStream_t st, hipStream_t side = nullptr) {
  void* stream = (void*)st;
  memset(out, 0, sizeof(*out));
  out->n_levels = n_levels;
  out->n_maps = n_specs;
  // level 0: the input coordinates and their hash table (CoordinateManager.__init__)
  const long long cap0 = pow2_cap(n);
  out->coords[0] = (int32_t*)coords;
  out->n_rows[0] = n;
  out->cap[0] = cap0;
  // the hash tables of ALL levels side by side (same capacity each): one EMPTY fill for the lot
  int64_t* tables_all = A.take_n<int64_t>(cap0 * 2 * n_levels);
  for (int l = 0; l < n_levels; ++l) {
    out->table[l] = tables_all + (long long)l * cap0 * 2;
    out->cap[l] = cap0;
  }
  // status words, the levels' meta words and every map's counts in ONE block: one zero fill, one read-back copy
  const long long n_meta = 8 * (n_levels > 1 ? n_levels - 1 : 1);
  int32_t* status = A.take_n<int32_t>(4 + n_meta + 128ll * GCL_MAX_MAPS);
  if (!A.dry) tables_fill(tables_all, cap0 * n_levels, status, st);
  PLAN_CALL(coords_insert_impl(coords, n, out->table[0], cap0, status, true, stream));
  // levels 1 ..: one chain of launches, row counts stay on the device (CoordinateManager._build_stride_maps)
  // one zero fill for the levels' meta words AND every map's per-offset counts (gcl_kernel_map then counts by integer
  // atomics: no reduction launch per map)
  int32_t* meta = status + 4;
  int32_t* counts_all = meta + n_meta;
  if (!A.dry) GCL_CHECK_HIP(hipMemsetAsync(meta, 0, sizeof(int32_t) * (size_t)(n_meta + 128ll * GCL_MAX_MAPS), st));
  const int32_t* cb = coords;
  const int32_t* n_dev = nullptr;
  for (int l = 1; l < n_levels; ++l) {
    int32_t* scratch = A.take_n<int32_t>(gcl_scan_scratch_len(n));
    out->coords[l] = A.take_n<int32_t>(n * 4);
    // (the level's status words, meta + 8 (l - 1) + 4, are zero from the block's fill above)
    PLAN_CALL(stride_map_impl(cb, n, n_dev, 1 << l, out->table[l], cap0, scratch, out->coords[l], meta + 8 * (l - 1),
                              meta + 8 * (l - 1) + 4, nullptr, true, stream));
    cb = out->coords[l];
    n_dev = meta + 8 * (l - 1);
    out->n_rows[l] = n;        // upper bound until the read-back below
  }
  if (!A.dry) {     // the first of the two host syncs: input status + level sizes
    GCL_CHECK_HIP(hipMemcpyAsync(pinned, status, sizeof(int32_t) * (4 + (n_levels > 1 ? 8 * (n_levels - 1) : 0)),
                                 hipMemcpyDeviceToHost, st));
    GCL_CHECK_HIP(hipStreamSynchronize(st));
    for (int j = 0; j < 4; ++j) out->status[j] = pinned[j];
    GCL_CHECK_ARG(pinned[0] == 0, "%d coordinates outside the packable range (batch < 65535, |x|,|y|,|z| < 32768)", pinned[0]);
    GCL_CHECK_ARG(pinned[1] == 0, "%d duplicate coordinates: ME.SparseTensor expects unique rows (use ME.utils.sparse_quantize)",
                  pinned[1]);
    for (int l = 1; l < n_levels; ++l) {
      GCL_CHECK_ARG(pinned[4 + 8 * (l - 1) + 4] == 0, "%d strided coordinates outside the packable range",
                    pinned[4 + 8 * (l - 1) + 4]);
      out->n_rows[l] = pinned[4 + 8 * (l - 1)];
    }
  }
  // kernel maps (CoordinateManager.get_kernel_map), one presence bitmap per coordinate table
  int32_t* bitmap[2][GCL_MAX_LEVELS] = {{nullptr}, {nullptr}};      // [built on the side stream][level]
  bool any_pairs = false;      // pair lists wanted (training): their counts are read back; inference skips the D2H copies
  for (int s = 0; s < n_specs; ++s) any_pairs = any_pairs || (specs[s].pairs != 0 && specs[s].kernel_size > 1);
  // SPLIT build (gcl_maps_build_split, inference): once the level sizes are on the host nothing below waits for the host
  // again, so the maps of the input level alone (what the first layers of a network use) stay on `stream` and every other
  // map and its sorted tables go to `side` -- they are built BESIDE the first layers' convolutions, which the caller can
  // enqueue on `stream` straight away (gcl_maps_desc.late_mask / ready_event tell the plan where to wait).  The dry run
  // reserves for the split (one more presence bitmap per level).
  const bool split = (side != nullptr || A.dry) && !any_pairs;
  auto on_side = [&](const gcl_map_spec& sp) { return split && !(sp.t_in == 1 && sp.stride == 1); };
  // SMALL builds (<= 65 536 input rows, one stream): every neighbour table of the build is pre-filled with -1 by ONE launch
  // over the arena range that holds them (gcl_kernel_map flag bit 3) instead of a fill per map -- a pass over one pair of
  // clouds pays ~ 4.8 us of dispatch latency per dependent launch.  The launches are collected first, issued after the fill.
  struct MapLaunch { int s, src5, flags; int32_t *bitmap, *scratch; void* stream; };
  MapLaunch launches[GCL_MAX_MAPS];
  int n_launches = 0;
  const bool prefill = !A.dry && !side && n <= 65536;
  char* fill_lo = nullptr;
  for (int s = 0; s < n_specs; ++s) {
    const gcl_map_spec& sp = specs[s];
    gcl_map_desc& d = out->maps[s];
    d.t_in = sp.t_in;
    d.kernel_size = sp.kernel_size;
    d.stride = sp.stride;
    d.K = sp.kernel_size * sp.kernel_size * sp.kernel_size;
    d.level_in = level_of(sp.t_in);
    d.level_out = level_of(sp.t_in * sp.stride);
    GCL_CHECK_ARG(d.level_in >= 0 && d.level_out >= 0 && d.level_out < n_levels && (sp.stride == 1 || sp.stride == 2),
                  "gcl_maps_build: map %d (t_in %d, stride %d) is outside the %d levels built", s, sp.t_in, sp.stride, n_levels);
    GCL_CHECK_ARG(sp.kernel_size == 1 || sp.kernel_size == 3 || sp.kernel_size == 5, "gcl_maps_build: kernel size 1, 3 or 5");
    d.n_in = out->n_rows[d.level_in];
    d.n_out = out->n_rows[d.level_out];
    if (sp.kernel_size == 1) {
      GCL_CHECK_ARG(sp.stride == 1, "gcl_maps_build: kernel_size 1 with stride > 1");
      continue;      // identity pairs only, after the counts are known (nothing to count here)
    }
    const int sd = on_side(sp) ? 1 : 0;
    void* mstream = (sd && side) ? (void*)side : stream;
    if (sd) out->late_mask |= 1 << s;
    d.nbr = A.take_n<int32_t>((long long)d.K * d.n_out);
    if (!fill_lo) fill_lo = (char*)d.nbr;
    const bool same = sp.stride == 1;
    d.nbr_t = same ? nullptr : A.take_n<int32_t>((long long)d.K * d.n_in);
    d.counts = counts_all + 128 * s;
    // a 3^3 stride-1 map of a table whose 5^3 stride-1 map is already built: 27 of its rows (GCL_MAP3_FROM5=0: probe again)
    static const bool from5 = [] { const char* e = getenv("GCL_MAP3_FROM5"); return !(e && e[0] == '0'); }();
    int src5 = -1;
    for (int q = 0; q < s && from5 && same && sp.kernel_size == 3; ++q)
      if (specs[q].t_in == sp.t_in && specs[q].kernel_size == 5 && specs[q].stride == 1 && out->maps[q].nbr) src5 = q;
    if (src5 >= 0) {
      launches[n_launches++] = MapLaunch{s, src5, 0, nullptr, nullptr, mstream};
      continue;
    }
    // the presence bitmap (2 MB) pays when the table is much larger than it; a table of <= 8 MB (<= 256 k slots: a pass over
    // a few clouds) sits in the caches itself and the bitmap's fill + build are two launches per level for nothing
    const bool use_bitmap = A.dry || out->cap[d.level_in] * 16 > 4 * gcl_kernel_map_bitmap_len() * (long long)sizeof(int32_t);
    const bool bitmap_valid = use_bitmap && bitmap[sd][d.level_in] != nullptr;     // shared by the maps of ONE stream only
    if (use_bitmap && !bitmap_valid) bitmap[sd][d.level_in] = A.take_n<int32_t>(gcl_kernel_map_bitmap_len());
    int32_t* scratch = A.take_n<int32_t>(gcl_kernel_map_scratch_len(sp.kernel_size, d.n_out));
    launches[n_launches++] = MapLaunch{s, -1, (same ? 1 : 0) | (bitmap_valid ? 2 : 0) | 4 | (prefill ? 8 : 0),
                                       use_bitmap ? bitmap[sd][d.level_in] : nullptr, scratch, mstream};
  }
  if (prefill && fill_lo) fill32(fill_lo, (long long)((A.base + A.off) - fill_lo) / 4, 0xFFFFFFFFu, st);
  for (int q = 0; q < n_launches; ++q) {
    const MapLaunch& L = launches[q];
    const gcl_map_spec& sp = specs[L.s];
    gcl_map_desc& d = out->maps[L.s];
    if (L.src5 >= 0)
      PLAN_CALL(gcl_kernel_map_3_from_5(out->maps[L.src5].nbr, out->maps[L.src5].counts, d.n_out, d.nbr, d.counts, L.stream));
    else
      PLAN_CALL(gcl_kernel_map(out->coords[d.level_out], d.n_out, out->table[d.level_in], out->cap[d.level_in],
                               sp.kernel_size, sp.t_in, L.flags, L.bitmap, L.scratch, d.nbr, d.nbr_t, d.n_in, d.counts, L.stream));
    if (!A.dry && any_pairs)   // without pair lists nobody waits for this copy: it would outlive the call (pinned re-use)
      GCL_CHECK_HIP(hipMemcpyAsync(pinned + 128 * (L.s + 1), d.counts, sizeof(int32_t) * d.K, hipMemcpyDeviceToHost, st));
  }
  // mask-sorted tables (KernelMap.sorted_table) of all maps in ONE gcl_table_sort_multi sequence (14 launches instead of
  // 14 per table; same results); K > 27 tables are used as they are
  gcl_sort_job jobs[2][2 * GCL_MAX_MAPS];      // [on the side stream][job]
  int n_jobs[2] = {0, 0};
  for (int s = 0; s < n_specs; ++s) {
    const gcl_map_spec& sp = specs[s];
    gcl_map_desc& d = out->maps[s];
    if (sp.kernel_size == 1) continue;
    for (int tr = 0; tr < 2; ++tr) {
      if (!(sp.tables & (1 << tr))) continue;
      const int32_t* tbl = tr ? d.nbr_t : d.nbr;
      GCL_CHECK_ARG(tbl, "gcl_maps_build: map %d has no transposed table (stride 1)", s);
      const long long rows = tr ? d.n_in : d.n_out;
      GCL_CHECK_ARG(d.K == 27, "gcl_maps_build: sorted tables are built for 3^3 kernels");
      int32_t* scratch = A.take_n<int32_t>(gcl_table_sort_scratch_len(rows));
      int32_t* order = A.take_n<int32_t>(rows);
      int32_t* sorted = A.take_n<int32_t>((long long)d.K * rows);
      int32_t* mask = A.take_n<int32_t>(cdiv(rows, 32));
      const int sd = on_side(sp) ? 1 : 0;
      jobs[sd][n_jobs[sd]++] = gcl_sort_job{tbl, d.K, rows, scratch, order, sorted, mask, d.counts};
      if (tr) { d.tbl_t = sorted; d.order_t = order; d.mask_t = mask; }
      else { d.tbl_n = sorted; d.order_n = order; d.mask_n = mask; }
    }
  }
  static const int sort_multi = [] { const char* e = getenv("GCL_SORT_MULTI"); return e ? atoi(e) : 1; }();
  for (int s = 0; s < n_specs; ++s) {      // presence words of the first layer's table (occupancy path of the stem kernels)
    if (!(specs[s].tables & 4) || specs[s].kernel_size == 1) continue;
    gcl_map_desc& d = out->maps[s];
    d.presence = A.take_n<uint32_t>(d.n_out * ((d.K + 31) / 32));
    PLAN_CALL(gcl_presence_bits(d.nbr, d.K, d.n_out, d.presence, (on_side(specs[s]) && side) ? (void*)side : stream));
  }
  for (int sd = 0; sd < 2; ++sd) {      // the main stream's tables first: the first layers wait for them
    void* sstream = (sd && side) ? (void*)side : stream;
    if (n_jobs[sd] && sort_multi) PLAN_CALL(gcl_table_sort_multi(jobs[sd], n_jobs[sd], sstream));
    if (n_jobs[sd] && !sort_multi)
      for (int q = 0; q < n_jobs[sd]; ++q)
        PLAN_CALL(gcl_table_sort_pre(jobs[sd][q].tbl, jobs[sd][q].K, jobs[sd][q].n, 0, nullptr, jobs[sd][q].scratch,
                                     jobs[sd][q].order, jobs[sd][q].tbl_sorted, jobs[sd][q].tile_mask, sstream));
  }
  if (!side) out->late_mask = 0;       // (dry run / one stream: nothing is late)
  // the second host sync: per-offset pair counts -> padded segment offsets -> pair lists (KernelMap.pairs); skipped when
  // no map asks for pair lists (inference): counts_host / n_pairs / seg_off then stay zero
  if (!A.dry && any_pairs) GCL_CHECK_HIP(hipStreamSynchronize(st));
  for (int s = 0; s < n_specs; ++s) {
    const gcl_map_spec& sp = specs[s];
    gcl_map_desc& d = out->maps[s];
    if (sp.kernel_size == 1) {
      if (!sp.pairs) continue;
      const long long total = cdiv(d.n_in, GCL_PAIR_CHUNK) * GCL_PAIR_CHUNK;
      d.pair_in = d.pair_out = A.take_n<int32_t>(total);
      d.seg_off[0] = 0;
      d.seg_off[1] = total;
      d.n_pairs = d.n_in;
      d.counts_host[0] = (int32_t)d.n_in;
      if (!A.dry) {
        hipLaunchKernelGGL(k_identity_pairs, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, st, d.pair_in, (long long)d.n_in,
                           total);
        GCL_CHECK_LAUNCH();
      }
      continue;
    }
    if (!any_pairs && !A.dry) continue;
    long long total = 0;
    d.seg_off[0] = 0;
    for (int k = 0; k < d.K; ++k) {
      // dry run: every output row could own every offset
      const long long c = A.dry ? d.n_out : (long long)pinned[128 * (s + 1) + k];
      d.counts_host[k] = (int32_t)c;
      d.n_pairs += c;
      total += cdiv(c, GCL_PAIR_CHUNK) * GCL_PAIR_CHUNK;
      d.seg_off[k + 1] = total;
    }
    if (!sp.pairs) continue;
    const long long len = total > 0 ? total : 1;
    int32_t* both = A.take_n<int32_t>(2 * len);
    d.pair_in = both;
    d.pair_out = both + len;
    int32_t* scratch = A.take_n<int32_t>((long long)d.K * cdiv(d.n_out, 1024) + d.K + 1);
    PLAN_CALL(gcl_kernel_map_pairs(d.nbr, d.K, d.n_out, d.seg_off, scratch, d.pair_in, d.pair_out, stream));
    // cell limits of the weight gradient's range-grouped launches: a function of the map, made here once (side stream)
    // instead of by every convolution that uses the map (k_pair_bounds: 9 launches per step on the weight-gradient stream)
    // (dry run: n_out is an upper bound of the level's rows and gcl_conv_bwd_weight_bounds_len is not monotone in the row
    // count -- the range length doubles as the level grows -- so the bound reserves for the SHORTEST range, 512 rows)
    const long long nb = !A.dry ? gcl_conv_bwd_weight_bounds_len(d.K, d.n_out)
                                : (d.n_out >= 32768 && d.K > 1 && d.K <= 27 ? (long long)d.K * (cdiv(d.n_out, 512) + 1) : 0);
    if (nb > 0 && total > 0) {
      d.dw_bounds = A.take_n<int32_t>(nb);
      PLAN_CALL(gcl_conv_bwd_weight_bounds(d.pair_out, d.seg_off, d.K, d.n_out, d.dw_bounds, stream));
    }
  }
  out->arena_used = A.off;
  if (!A.fits()) {
    set_error("gcl_maps_build: arena too small (%lld bytes needed, %lld given)", A.off, A.size);
    return GCL_ERR_ARENA;
  }
  return GCL_OK;
}

// ---------------------------------------------------------------------------------------------------
// plan
// ---------------------------------------------------------------------------------------------------
struct TState {            // a tensor of the pass (forward value or gradient)
  float* ptr = nullptr;
  int32_t* amax = nullptr;   // amax slot that holds max|tensor|, or NULL = not measured yet
  void* planes = nullptr;    // gcl_split_planes image, or NULL = not made yet
  int ld = 0;                // row pitch in floats when the tensor is a column slice of a wider one (gradients of ME.cat
                             // inputs: slices of the gradient of the cat's output), 0 = its own channel count
};

struct OpSaved {           // what the backward pass of a record needs from its forward pass
  float* conv_out = nullptr;            // CONVBN: the convolution output (the BatchNorm input)
  unsigned long long* mask = nullptr;   // CONVBN with relu: sign bits of the output
  float *mean = nullptr, *rstd = nullptr;
  float* xrange = nullptr;              // CONVBN in bound mode: per-channel minimum / maximum of conv_out ([2][c])
  int32_t* x_amax = nullptr;
  float* norm = nullptr;                // ROWNORM
};

struct ProfRec {
  hipEvent_t e0, e1;
  double kind, pairs, cin, cout, n_in, n_out, K;
};

struct PassState {        // everything one forward pass leaves behind for its backward pass
  const gcl_maps_desc* maps = nullptr;
  gcl_maps_desc maps_copy;
  std::vector<TState> t, g;
  std::vector<OpSaved> saved;
  std::vector<void*> params;
  Arena A{DRY_BASE, 0, 0, true};
  int32_t* slot_pool = nullptr;
  long long n_slots = 0, next_slot = 0;
  bool slots_exhausted = false;
  int32_t* w_amax = nullptr;        // [n_weights][GCL_AMAX_WORDS]
  unsigned char *pack_fwd = nullptr, *pack_bwd = nullptr;
  void* state = nullptr;
  bool forward_done = false, bwd_packed = false;
  std::vector<char> relu_masked;    // backward pass: RELU record whose backward the consumer's epilogue has already applied
  hipEvent_t fwd_packs = nullptr;      // recorded on the aux stream behind the forward packs of the pass
  bool fwd_packs_pending = false;
  int32_t* not_ones = nullptr;      // per input row of the pass: its cloud has a feature that differs from 1.0f (table path)
  void* key = nullptr;              // the pass's arena: how gcl_plan_backward / gcl_plan_release find it
};

struct Plan : PassState {   // the base part is the pass being enqueued right now
  std::vector<PassState*> passes;   // passes waiting for their backward pass (several forwards may be outstanding)
  std::vector<gcl_plan_op> ops;
  int n_tensors = 0, n_params = 0, n_bn = 0, presplit = 128;
  std::vector<int> worder;          // parameter ids of the MFMA-shaped kernels (amax slot / pack order)
  std::vector<int> widx;            // parameter id -> position in worder, -1 otherwise
  std::vector<int> wmode;           // per position: input-gradient pack mode (1 | 2), 0 = no input gradient needed
  std::vector<long long> wK, wcin, wcout, off_fwd, off_bwd;
  long long bytes_fwd = 0, bytes_bwd = 0, wgs_fwd = 0, wgs_bwd = 0;
  int n_bwd = 0;
  std::vector<char> made;           // tensor id -> produced by a record of the plan
  std::vector<char> fuse_relu;      // record i is a CONV whose only consumer is the RELU record i + 1: one launch writes relu(y)
  std::vector<int> cat_left;        // record i is a CONVBN whose output is only the LEFT input of CAT record cat_left[i]: its
                                    // BatchNorm apply pass writes into the cat's output (row pitch = the cat's width)
  std::vector<int> mask_from;       // record i is a convolution whose input is the output of RELU record mask_from[i] and that
                                    // ReLU's only consumer: its input-gradient epilogue applies the ReLU's backward (-1: no)
  // device tables (inside the caller's `state` buffer), re-uploaded when a parameter pointer changes
  std::vector<long long> host_tables, uploaded;
  // inference passes (gcl_plan_forward_eval): BatchNorm in eval mode folded into the convolution epilogue
  bool eval = false;
  void* const* bn_eval = nullptr;   // per BatchNorm: scale, shift, mean, rstd (device pointers)
  bool eval_repack = true;
  // optional second stream for the weight gradients (off the critical path of the backward pass)
  hipStream_t aux = nullptr;
  std::vector<hipEvent_t> events;
  size_t events_used = 0;
  bool aux_dirty = false;
  // profiling
  bool profile = false;
  std::vector<ProfRec> prof;
  size_t prof_used = 0;
};

static long long state_words(const Plan& P) { return (long long)P.worder.size() * (2 + 8 + 8); }

// GCL_BN_PLANES=0: the BatchNorm passes publish the measured max-abs and every plane image is a gcl_split_planes pass of
// its own (rounds 1 - 4); ops.py reads the same variable
static bool bound_mode() {
  static const bool on = [] { const char* e = getenv("GCL_BN_PLANES"); return !(e && e[0] == '0'); }();
  return on;
}

static int32_t* new_slot(Plan& P) {      // slots are used once per pass (they start zeroed); the pool is sized for the worst case
  if (P.next_slot >= P.n_slots) {
    P.slots_exhausted = true;
    return P.slot_pool;
  }
  return P.slot_pool + (P.next_slot++) * GCL_AMAX_WORDS;
}

static int ensure_amax(Plan& P, TState& ts, long long numel, hipStream_t st) {
  Arena& A = P.A;
  if (ts.amax) return GCL_OK;
  ts.amax = new_slot(P);
  PLAN_CALL(gcl_amax(ts.ptr, numel, ts.amax, 1, (void*)st));
  return GCL_OK;
}

static int ensure_planes(Plan& P, TState& ts, long long n, int c, hipStream_t st) {
  Arena& A = P.A;
  if (ts.planes) return GCL_OK;
  ts.planes = A.take(n * c * 4);
  PLAN_CALL(gcl_split_planes(ts.ptr, n, c, ts.amax, ts.planes, (void*)st));
  return GCL_OK;
}

// The weight gradient of a record is needed only by the optimizer: with an aux stream it is enqueued there, ordered
// behind everything the main stream has issued so far (its operands), and the main stream goes on with the input-gradient
// chain.  join_aux makes the main stream wait for all of it (end of a backward segment).
static hipStream_t fork_aux(Plan& P, hipStream_t st) {
  if (!P.aux || P.A.dry) return st;
  if (P.events_used == P.events.size()) {
    hipEvent_t e;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return st;
    P.events.push_back(e);
  }
  hipEvent_t e = P.events[P.events_used++];
  if (hipEventRecord(e, st) != hipSuccess || hipStreamWaitEvent(P.aux, e, 0) != hipSuccess) return st;
  P.aux_dirty = true;
  return P.aux;
}

static int join_aux(Plan& P, hipStream_t st) {
  if (!P.aux || !P.aux_dirty || P.A.dry) return GCL_OK;
  if (P.events_used == P.events.size()) {
    hipEvent_t e;
    GCL_CHECK_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    P.events.push_back(e);
  }
  hipEvent_t e = P.events[P.events_used++];
  GCL_CHECK_HIP(hipEventRecord(e, P.aux));
  GCL_CHECK_HIP(hipStreamWaitEvent(st, e, 0));
  P.aux_dirty = false;
  return GCL_OK;
}

struct ProfScope {      // brackets one launch with events when profiling is armed
  Plan& P;
  hipStream_t st;
  ProfRec* r = nullptr;
  ProfScope(Plan& P_, hipStream_t st_, int kind, double pairs, int cin, int cout, long long n_in, long long n_out, int K)
      : P(P_), st(st_) {
    if (!P.profile || P.A.dry) return;
    if (P.prof_used == P.prof.size()) {
      ProfRec nr{};
      if (hipEventCreate(&nr.e0) != hipSuccess || hipEventCreate(&nr.e1) != hipSuccess) return;
      P.prof.push_back(nr);
    }
    r = &P.prof[P.prof_used++];
    r->kind = kind; r->pairs = pairs; r->cin = cin; r->cout = cout; r->n_in = (double)n_in; r->n_out = (double)n_out; r->K = K;
    (void)hipEventRecord(r->e0, st);
  }
  ~ProfScope() { if (r) (void)hipEventRecord(r->e1, st); }
};

static bool is_stem(const gcl_plan_op& op, const gcl_map_desc& m) {
  return op.cin <= 4 && (op.cout % 32) == 0 && !op.transpose && m.kernel_size > 1 && op.bias < 0;
}

// occupancy rows of the first layer: measured once per pass, per cloud (the reference's training loaders jitter the centre
// cloud of a sample only), spread to rows -- device flags, no host decision
static int stem_flag(Plan& P, const gcl_plan_op& op, const gcl_map_desc& m, const TState& x, long long n_in, hipStream_t st) {
  Arena& A = P.A;
  if (!m.presence || P.not_ones) return GCL_OK;
  constexpr int CLOUD_FLAGS = 4096;      // batch indices beyond this take the table path
  int32_t* cloud = A.take_n<int32_t>(CLOUD_FLAGS);
  P.not_ones = A.take_n<int32_t>(n_in);
  PLAN_CALL(gcl_not_ones_rows(x.ptr, op.cin, P.maps->coords[op.level_in], n_in, cloud, CLOUD_FLAGS, P.not_ones, (void*)st));
  return GCL_OK;
}

// the convolution of a CONVBN / CONV record (ops._SparseConvFn.forward); returns its output in *y_out
static int conv_forward(Plan& P, int i, float** y_out, float** stats_out, hipStream_t st, int32_t** relu_amax_out) {
  Arena& A = P.A;
  const gcl_plan_op& op = P.ops[i];
  const gcl_maps_desc& M = *P.maps;
  const gcl_map_desc& m = M.maps[op.map];
  const long long n_in = M.n_rows[op.level_in], n_out = M.n_rows[op.level_out];
  TState& x = P.t[op.x];
  float* y = A.take_n<float>(n_out * op.cout);
  *y_out = y;
  *stats_out = nullptr;
  *relu_amax_out = nullptr;
  const float* W = (const float*)P.params[op.w];
  if (is_stem(op, m)) {
    int rcs = stem_flag(P, op, m, x, n_in, st);
    if (rcs) return rcs;
    PLAN_CALL(gcl_stem_fwd(x.ptr, W, m.nbr, n_out, op.K, op.cin, op.cout, y, m.presence, m.presence ? P.not_ones : nullptr,
                           (void*)st));
    return GCL_OK;
  }
  const int wi = P.widx[op.w];
  if (P.fwd_packs_pending && !A.dry) {      // first convolution that reads packed weights
    GCL_CHECK_HIP(hipStreamWaitEvent(st, P.fwd_packs, 0));
    P.fwd_packs_pending = false;
  }
  int rc = ensure_amax(P, x, n_in * op.cin, st);
  if (rc) return rc;
  P.saved[i].x_amax = x.amax;
  const bool pl = op.cin >= P.presplit;
  if (pl && (rc = ensure_planes(P, x, n_in, op.cin, st))) return rc;
  float* stats = nullptr;
  if (op.kind == GCL_OP_CONVBN) stats = A.take_n<float>(cdiv(n_out, 128) * 4 * op.cout);      // sum, squares, min, max per column and 128-row tile
  *stats_out = stats;
  const int32_t *tbl = nullptr, *order = nullptr, *mask = nullptr;
  if (m.kernel_size > 1) {
    tbl = op.transpose ? m.tbl_t : m.tbl_n;
    order = op.transpose ? m.order_t : m.order_n;
    mask = op.transpose ? m.mask_t : m.mask_n;
    GCL_CHECK_ARG(A.dry || tbl, "gcl_plan_forward: record %d needs a sorted table the maps do not carry", i);
  }
  const float* bias = op.bias >= 0 ? (const float*)P.params[op.bias] : nullptr;
  ProfScope ps(P, st, 0, (double)(m.kernel_size > 1 ? m.n_pairs : n_out), op.cin, op.cout, n_in, n_out, op.K);
  int32_t* relu_amax = (op.kind == GCL_OP_CONV && P.fuse_relu[i]) ? new_slot(P) : nullptr;
  *relu_amax_out = relu_amax;
  PLAN_CALL(gcl_conv_fwd_fused(pl ? (const float*)x.planes : x.ptr, n_in, pl ? 1 : 0, P.pack_fwd + P.off_fwd[wi], 4, x.amax,
                               P.w_amax + (long long)wi * GCL_AMAX_WORDS, tbl, order, mask, n_out, op.K, op.cin, op.cout, bias,
                               nullptr, nullptr, relu_amax ? 1 : 0, relu_amax, y, stats, 0, (void*)st));
  return GCL_OK;
}

static int upload_tables(Plan& P, hipStream_t st) {
  // [ptrs n | sizes n | fwd desc n x 8 | bwd desc n_bwd x 8]  (WeightAmaxGroup.refresh / .packed)
  const size_t n = P.worder.size();
  std::vector<long long>& h = P.host_tables;
  h.assign((size_t)state_words(P), 0);
  long long wg = 0;
  for (size_t q = 0; q < n; ++q) {
    h[q] = (long long)(uintptr_t)P.params[P.worder[q]];
    h[n + q] = P.wK[q] * P.wcin[q] * P.wcout[q];
    long long* d = &h[2 * n + 8 * q];
    d[0] = h[q]; d[1] = P.wK[q]; d[2] = P.wcin[q]; d[3] = P.wcout[q]; d[4] = 0; d[5] = (long long)q; d[6] = P.off_fwd[q]; d[7] = wg;
    wg += cdiv(h[n + q], 256);
  }
  P.wgs_fwd = wg;
  wg = 0;
  size_t row = 0;
  for (size_t q = 0; q < n; ++q) {
    if (!P.wmode[q]) continue;
    long long* d = &h[2 * n + 8 * n + 8 * row++];
    d[0] = h[q]; d[1] = P.wK[q]; d[2] = P.wcin[q]; d[3] = P.wcout[q]; d[4] = P.wmode[q]; d[5] = (long long)q; d[6] = P.off_bwd[q]; d[7] = wg;
    wg += cdiv(h[n + q], 256);
  }
  P.wgs_bwd = wg;
  if (h != P.uploaded) {      // parameters were (re-)seated: rare; the copy is finished before the vector can change
    GCL_CHECK_HIP(hipMemcpyAsync(P.state, h.data(), h.size() * sizeof(long long), hipMemcpyHostToDevice, st));
    GCL_CHECK_HIP(hipStreamSynchronize(st));
    P.uploaded = h;
  }
  return GCL_OK;
}

static int plan_forward(Plan& P, const float* x_in, void* const* bn_stats, float** y_out, hipStream_t st) {
  Arena& A = P.A;
  const gcl_maps_desc& M = *P.maps;
  const size_t nw = P.worder.size();
  P.t.assign(P.n_tensors, TState());
  P.g.assign(P.n_tensors, TState());
  P.saved.assign(P.ops.size(), OpSaved());
  if (P.profile && !A.dry) P.prof_used = 0;
  if (!A.dry) P.events_used = 0;      // fork / join events are re-used pass after pass (recording re-arms them)      // records of a profiled pass stay readable until the next profiled pass
  // amax slots of the pass: one zero fill (ops.amax_slot hands out slots of a zero-filled pool)
  P.n_slots = 4 * (long long)P.n_tensors + 2 * (long long)P.ops.size() + 16;
  P.next_slot = 0;
  P.slots_exhausted = false;
  P.not_ones = nullptr;
  P.fwd_packs_pending = false;
  P.slot_pool = A.take_n<int32_t>(P.n_slots * GCL_AMAX_WORDS);
  if (!A.dry) GCL_CHECK_HIP(hipMemsetAsync(P.slot_pool, 0, (size_t)P.n_slots * GCL_AMAX_WORDS * sizeof(int32_t), st));
  // all convolution kernels: max|W| in one launch, forward packs in one launch (WeightAmaxGroup)
  if (P.eval) {      // inference: max|W| and the packed kernels persist in the caller's state buffer from pass to pass
    char* base = (char*)P.state + ((state_words(P) * 8 + 255) & ~255ll);
    P.w_amax = (int32_t*)base;
    P.pack_fwd = (unsigned char*)(base + (((long long)nw * GCL_AMAX_WORDS * 4 + 255) & ~255ll));
  } else {
    P.w_amax = A.take_n<int32_t>((long long)nw * GCL_AMAX_WORDS);
    P.pack_fwd = (unsigned char*)A.take(P.bytes_fwd);
  }
  P.bwd_packed = false;
  if (nw && !A.dry) {
    int rc = upload_tables(P, st);
    if (rc) return rc;
    if (!P.eval || P.eval_repack) {
      const long long* tab = (const long long*)P.state;
      // training with an aux stream: max|W| and the forward packs are made THERE, beside the first layer (whose kernels
      // read the fp32 weights); the main stream waits for them in front of its first packed convolution (conv_forward)
      hipStream_t ps = st;
      if (!P.eval && P.aux && P.n_bwd) ps = fork_aux(P, st);
      PLAN_CALL(gcl_amax_multi((const float* const*)tab, (const int64_t*)(tab + nw), (int32_t)nw, P.w_amax, (void*)ps));
      PLAN_CALL(gcl_pack_weights_multi((const int64_t*)(tab + 2 * nw), (int32_t)nw, P.wgs_fwd, 4, P.w_amax, P.pack_fwd, (void*)ps));
      if (ps != st) {
        if (P.events_used == P.events.size()) {
          hipEvent_t e;
          GCL_CHECK_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
          P.events.push_back(e);
        }
        P.fwd_packs = P.events[P.events_used++];
        GCL_CHECK_HIP(hipEventRecord(P.fwd_packs, ps));
        P.fwd_packs_pending = true;
      }
    }
  }
  // with an aux stream the input-gradient packs (needed by the first record of the backward pass) are made there now,
  // beside the forward convolutions, instead of at the head of the backward pass
  if (!P.eval && P.aux && P.n_bwd) {
    P.pack_bwd = (unsigned char*)A.take(P.bytes_bwd);
    if (!A.dry) {
      hipStream_t ws = fork_aux(P, st);
      const long long* tab = (const long long*)P.state;
      PLAN_CALL(gcl_pack_weights_multi((const int64_t*)(tab + 2 * nw + 8 * nw), P.n_bwd, P.wgs_bwd, 4, P.w_amax, P.pack_bwd,
                                       (void*)ws));
    }
    P.bwd_packed = true;
  }
  P.t[0].ptr = (float*)x_in;
  // maps of a split build (gcl_maps_build_split): the first record that uses a map made on the side stream waits for it
  bool late_pending = !A.dry && M.ready_event && M.late_mask;
  for (size_t i = 0; i < P.ops.size(); ++i) {
    const gcl_plan_op& op = P.ops[i];
    const long long n_out = M.n_rows[op.level_out];
    if (late_pending && (op.kind == GCL_OP_CONVBN || op.kind == GCL_OP_CONV) && ((M.late_mask >> op.map) & 1)) {
      GCL_CHECK_HIP(hipStreamWaitEvent(st, (hipEvent_t)M.ready_event, 0));
      late_pending = false;
    }
    TState& y = P.t[op.y];
    int rc;
    switch (op.kind) {
      case GCL_OP_CONVBN: {
        if (P.eval) {      // ME.conv_bn in inference (ops.conv_bn_eval): conv + BatchNorm(running stats) + residual + ReLU, one launch
          const gcl_map_desc& m = M.maps[op.map];
          const long long n_in = M.n_rows[op.level_in];
          TState& x = P.t[op.x];
          const int c = op.cout;
          const float* res = op.x2 >= 0 ? P.t[op.x2].ptr : nullptr;
          y.ptr = A.take_n<float>(n_out * c);
          y.amax = new_slot(P);
          void* const* be = P.bn_eval ? P.bn_eval + 4 * op.bn : nullptr;
          if (is_stem(op, m)) {      // the Cin <= 4 first layer: VALU convolution, then the BatchNorm apply pass
            float* cy = A.take_n<float>(n_out * c);
            unsigned long long* mask = op.relu ? A.take_n<unsigned long long>(gcl_bn_mask_len(n_out, c)) : nullptr;
            if ((rc = stem_flag(P, op, m, x, n_in, st))) return rc;
            PLAN_CALL(gcl_stem_fwd(x.ptr, (const float*)P.params[op.w], m.nbr, n_out, op.K, op.cin, c, cy, m.presence,
                                   m.presence ? P.not_ones : nullptr, (void*)st));
            PLAN_CALL(gcl_bn_apply(cy, n_out, c, (const float*)be[2], (const float*)be[3], (const float*)P.params[op.bn_w],
                                   (const float*)P.params[op.bn_b], res, op.relu, y.ptr, (uint64_t*)mask, y.amax, (void*)st));
            break;
          }
          const int wi = P.widx[op.w];
          if ((rc = ensure_amax(P, x, n_in * op.cin, st))) return rc;
          const int32_t *tbl = nullptr, *order = nullptr, *mask = nullptr;
          if (m.kernel_size > 1) {
            tbl = op.transpose ? m.tbl_t : m.tbl_n;
            order = op.transpose ? m.order_t : m.order_n;
            mask = op.transpose ? m.mask_t : m.mask_n;
            GCL_CHECK_ARG(A.dry || tbl, "gcl_plan_forward_eval: record %d needs a sorted table the maps do not carry", (int)i);
          }
          // scratch of the offset-group launches (small deep layers of an inference pass; 0: the shape / size takes another kernel)
          const long long gs_len = tbl ? gcl_conv_fwd_groups_scratch_len(n_out, op.K, op.cin, c) : 0;
          float* gscratch = gs_len > 0 ? A.take_n<float>(gs_len) : nullptr;
          ProfScope ps(P, st, 0, (double)(m.kernel_size > 1 ? m.n_pairs : n_out), op.cin, c, n_in, n_out, op.K);
          PLAN_CALL(gcl_conv_fwd_fused(x.ptr, n_in, 0, P.pack_fwd + P.off_fwd[wi], 4, x.amax,
                                       P.w_amax + (long long)wi * GCL_AMAX_WORDS, tbl, order, mask, n_out, op.K, op.cin, c,
                                       (const float*)be[1], (const float*)be[0], res, op.relu, y.amax, y.ptr, gscratch,
                                       GCL_CONV_TALL, (void*)st));
          break;
        }
        float *cy, *stats;
        int32_t* unused_slot;
        if ((rc = conv_forward(P, (int)i, &cy, &stats, st, &unused_slot))) return rc;
        OpSaved& sv = P.saved[i];
        sv.conv_out = cy;
        const int c = op.cout;
        float* mr = A.take_n<float>(2 * c);
        sv.mean = mr;
        sv.rstd = mr + c;
        float* rm = (float*)bn_stats[2 * op.bn];
        float* rv = (float*)bn_stats[2 * op.bn + 1];
        // Bound mode (round 5): this BatchNorm's output is at least `presplit` channels wide, i.e. its consumers read a
        // plane image: the statistics launch bounds max|y| from the epilogue's column ranges and the apply pass writes the
        // image itself -- into the image of the ME.cat it writes in place, if so (no gcl_split_planes pass; ops.py mirrors the
        // slot values: _BatchNormFn, cat_features).  Needs the residual's / the cat partner's max-abs slots, which their
        // producers have published.
        const int32_t* res_amax = op.x2 >= 0 ? P.t[op.x2].amax : nullptr;
        const int32_t* cat_amax = P.cat_left[i] >= 0 ? P.t[P.ops[P.cat_left[i]].x2].amax : nullptr;
        const bool bound = stats && bound_mode() && c >= P.presplit && (c % 32) == 0 && (op.x2 < 0 || res_amax) &&
                           (P.cat_left[i] < 0 || (cat_amax && (P.ops[P.cat_left[i]].cout % 32) == 0));
        int32_t* bound_slot = bound ? new_slot(P) : nullptr;
        if (stats) {       // column sums (and ranges) from the convolution epilogue
          const long long nt = cdiv(n_out, 128);
          if (bound) sv.xrange = A.take_n<float>(2 * c);
          PLAN_CALL(gcl_bn_stats_from_tiles_range(stats, nt, n_out, c, op.eps, op.momentum, rm, rv, sv.mean, sv.rstd, sv.xrange,
                                                  (const float*)P.params[op.bn_w], (const float*)P.params[op.bn_b], op.relu,
                                                  res_amax, cat_amax, bound_slot, (void*)st));
        } else {
          double* scratch = A.take_n<double>(gcl_bn_scratch_len(n_out, c));
          PLAN_CALL(gcl_bn_stats(cy, n_out, c, op.eps, op.momentum, rm, rv, scratch, sv.mean, sv.rstd, (void*)st));
        }
        if (P.cat_left[i] >= 0) {
          // y is only the left input of an ME.cat: written straight into the cat's output (row pitch = its width); the
          // cat record then copies the other input's columns only.  max|y| goes to the cat's slot (the right half adds its own)
          const gcl_plan_op& cat = P.ops[P.cat_left[i]];
          TState& cty = P.t[cat.y];
          cty = TState();
          cty.ptr = A.take_n<float>(n_out * cat.cout);
          cty.amax = bound ? bound_slot : new_slot(P);
          if (bound) cty.planes = A.take(n_out * cat.cout * 4);
          y.ptr = cty.ptr;
          y.ld = cat.cout;
          y.amax = cty.amax;
          y.planes = nullptr;      // the image belongs to the cat (this tensor's only consumer)
        } else {
          y.ptr = A.take_n<float>(n_out * c);
          y.ld = 0;
          y.amax = bound ? bound_slot : new_slot(P);
          if (bound) y.planes = A.take(n_out * c * 4);
        }
        if (op.relu) sv.mask = A.take_n<unsigned long long>(gcl_bn_mask_len(n_out, c));
        const float* res = op.x2 >= 0 ? P.t[op.x2].ptr : nullptr;
        void* img = !bound ? nullptr : (P.cat_left[i] >= 0 ? P.t[P.ops[P.cat_left[i]].y].planes : y.planes);
        PLAN_CALL(gcl_bn_apply_planes(cy, n_out, c, sv.mean, sv.rstd, (const float*)P.params[op.bn_w],
                                      (const float*)P.params[op.bn_b], res, op.relu, y.ptr, y.ld, (uint64_t*)sv.mask, y.amax,
                                      img, (void*)st));
        break;
      }
      case GCL_OP_CONV: {
        float *cy, *stats;
        int32_t* relu_amax;
        if ((rc = conv_forward(P, (int)i, &cy, &stats, st, &relu_amax))) return rc;
        y.ptr = cy;
        if (relu_amax) {      // the epilogue applied the ReLU record that follows: both tensor ids name relu(conv(x))
          y.amax = relu_amax;
          P.t[P.ops[i + 1].y] = y;
        }
        break;
      }
      case GCL_OP_RELU: {
        if (i > 0 && P.fuse_relu[i - 1]) break;      // written by the convolution in front of it
        const long long n4 = n_out * op.cout / 4;
        y.ptr = A.take_n<float>(n_out * op.cout);
        y.amax = new_slot(P);        // published by the kernel itself: the consumer needs no gcl_amax pass
        if (!A.dry) {
          hipLaunchKernelGGL(k_relu_fwd, dim3(grid_for(n4)), dim3(256), 0, st, (const float4*)P.t[op.x].ptr, n4, (float4*)y.ptr,
                             y.amax);
          GCL_CHECK_LAUNCH();
        }
        break;
      }
      case GCL_OP_CAT: {
        const int ca = op.cin, cb = op.cout - op.cin;
        if (P.t[op.x].ld == op.cout && P.t[op.x].ptr == y.ptr && y.ptr) {      // left input already in place (cat_left)
          if (!A.dry) {
            hipLaunchKernelGGL(k_cat_right, dim3(grid_for(n_out * cb / 4)), dim3(256), 0, st, (const float4*)P.t[op.x2].ptr,
                               cb / 4, n_out, ca / 4, (float4*)y.ptr, y.amax, (unsigned short*)y.planes);
            GCL_CHECK_LAUNCH();
          }
          break;
        }
        y.ptr = A.take_n<float>(n_out * op.cout);
        y.amax = new_slot(P);
        // max|cat| = the larger of the inputs' slots when both are known (ops.cat_features: a slot may hold a BatchNorm's
        // bound instead of the measured maximum, and both paths must hand the consumers the same value); else measured
        const int32_t *sa = P.t[op.x].amax, *sb = P.t[op.x2].amax;
        if (!A.dry) {
          const bool both = sa && sb;
          hipLaunchKernelGGL(k_cat2, dim3(grid_for(n_out * op.cout / 4)), dim3(256), 0, st, (const float4*)P.t[op.x].ptr, ca / 4,
                             (const float4*)P.t[op.x2].ptr, cb / 4, n_out, (float4*)y.ptr, y.amax, both ? (const int*)sa : nullptr,
                             both ? (const int*)sb : nullptr);
          GCL_CHECK_LAUNCH();
        }
        break;
      }
      case GCL_OP_ROWNORM: {
        y.ptr = A.take_n<float>(n_out * op.cout);
        P.saved[i].norm = A.take_n<float>(n_out);
        PLAN_CALL(gcl_row_normalize_fwd(P.t[op.x].ptr, n_out, op.cout, y.ptr, P.saved[i].norm, (void*)st));
        break;
      }
      default:
        set_error("gcl_plan_forward: unknown record kind %d", op.kind);
        return GCL_ERR_ARG;
    }
  }
  if (late_pending) GCL_CHECK_HIP(hipStreamWaitEvent(st, (hipEvent_t)M.ready_event, 0));      // (the arena outlives the side stream's work)
  *y_out = P.t[P.ops.back().y].ptr;
  return GCL_OK;
}

// Tape.backward's give(): the first gradient of a tensor is kept, later ones are added in arrival order
// gptr: [rows, c] with row pitch ld floats (0 = c: contiguous; numel = rows * c)
static int give(Plan& P, int tensor, float* gptr, long long numel, hipStream_t st, int32_t* amax = nullptr, int ld = 0,
                int c = 0) {
  Arena& A = P.A;
  if (tensor < 0 || !P.made[tensor] || !gptr) return GCL_OK;
  TState& g = P.g[tensor];
  if (!g.ptr) {
    g = TState();
    g.ptr = gptr;
    g.ld = ld;
    g.amax = amax;        // published by the producing kernel (valid while the gradient stays this one tensor)
    return GCL_OK;
  }
  float* sum = A.take_n<float>(numel);
  if (!A.dry) {
    if (g.ld || ld) {
      GCL_CHECK_ARG(c > 0 && c % 4 == 0, "gcl_plan_backward: a sliced gradient needs its channel count");
      hipLaunchKernelGGL(k_add2_ld, dim3(grid_for(numel / 4)), dim3(256), 0, st, (const float*)g.ptr, g.ld ? g.ld : c,
                         (const float*)gptr, ld ? ld : c, numel / c, c / 4, (float4*)sum);
    } else {
      hipLaunchKernelGGL(k_add2, dim3(grid_for(numel / 4)), dim3(256), 0, st, (const float4*)g.ptr, (const float4*)gptr,
                         numel / 4, (float4*)sum);
    }
    GCL_CHECK_LAUNCH();
  }
  g = TState();
  g.ptr = sum;
  return GCL_OK;
}

// a gradient that is a column slice, for consumers that want contiguous rows: copied once (rare paths only -- the
// BatchNorm backward kernels and the convolution epilogue read slices as they are)
static int contiguous(Plan& P, TState& g, long long rows, int c, hipStream_t st) {
  if (!g.ld || !g.ptr) return GCL_OK;
  Arena& A = P.A;
  float* out = A.take_n<float>(rows * c);
  if (!A.dry) {
    hipLaunchKernelGGL(k_add2_ld, dim3(grid_for(rows * c / 4)), dim3(256), 0, st, (const float*)g.ptr, g.ld, (const float*)nullptr,
                       0, rows, c / 4, (float4*)out);
    GCL_CHECK_LAUNCH();
  }
  g = TState();
  g.ptr = out;
  return GCL_OK;
}

// ops._SparseConvFn.backward for record i with output gradient dy (dy.amax set when a producer published it)
static int conv_backward(Plan& P, int i, TState dy, void* const* grads, hipStream_t st) {
  Arena& A = P.A;
  const gcl_plan_op& op = P.ops[i];
  const gcl_maps_desc& M = *P.maps;
  const gcl_map_desc& m = M.maps[op.map];
  const long long n_in = M.n_rows[op.level_in], n_out = M.n_rows[op.level_out];
  TState& x = P.t[op.x];
  float* dW = (float*)grads[op.w];
  int rc;
  if (is_stem(op, m)) {
    float* scratch = A.take_n<float>(gcl_stem_bwd_weight_scratch_len(op.K, op.cin, op.cout, n_out));
    hipStream_t ws = fork_aux(P, st);
    ProfScope ps(P, ws, 2, (double)m.n_pairs, op.cin, op.cout, n_in, n_out, op.K);
    PLAN_CALL(gcl_stem_bwd_weight(x.ptr, dy.ptr, m.nbr, n_out, op.K, op.cin, op.cout, scratch, dW, m.presence,
                                  m.presence ? P.not_ones : nullptr, (void*)ws));
    return GCL_OK;
  }
  const int wi = P.widx[op.w];
  if ((rc = ensure_amax(P, dy, n_out * op.cout, st))) return rc;
  const int32_t* w_amax = P.w_amax + (long long)wi * GCL_AMAX_WORDS;
  const double pairs = (double)(m.kernel_size > 1 ? m.n_pairs : n_out);
  const bool pl_w = op.cin >= P.presplit && op.cout >= P.presplit;      // weight gradient on plane images
  const bool pl_x = op.cout >= P.presplit;                              // input gradient reads dy's plane image
  if ((pl_w || (pl_x && P.made[op.x])) && (rc = ensure_planes(P, dy, n_out, op.cout, st))) return rc;
  if (pl_w) {
    x.amax = P.saved[i].x_amax;
    if ((rc = ensure_planes(P, x, n_in, op.cin, st))) return rc;
  }
  // every operand of the weight gradient is enqueued: fork here, so that it runs beside the input gradient below
  hipStream_t ws = fork_aux(P, st);
  if (P.made[op.x]) {      // input gradient: the same output-stationary kernel over the opposite table
    int mode;
    const int32_t *tbl = nullptr, *order = nullptr, *mask = nullptr;
    if (m.kernel_size == 1) mode = 1;
    else if (op.transpose) { mode = 1; tbl = m.tbl_n; order = m.order_n; mask = m.mask_n; }
    else if (m.stride == 1) { mode = 2; tbl = m.tbl_n; order = m.order_n; mask = m.mask_n; }
    else { mode = 1; tbl = m.tbl_t; order = m.order_t; mask = m.mask_t; }
    GCL_CHECK_ARG(mode == P.wmode[wi], "gcl_plan_backward: record %d: input-gradient pack mode mismatch", i);
    GCL_CHECK_ARG(A.dry || m.kernel_size == 1 || tbl, "gcl_plan_backward: record %d needs a sorted table the maps do not carry", i);
    const bool pl = pl_x;
    float* acc = P.g[op.x].ptr;       // a gradient that already reached x through another path: added in the epilogue
    const int acc_ld = P.g[op.x].ld;  // ... possibly a column slice of a cat's gradient
    float* dx = A.take_n<float>(n_in * op.cin);
    int32_t* dx_amax = nullptr;
    {
      ProfScope ps(P, st, acc ? 1 : 0, pairs, op.cout, op.cin, n_out, n_in, op.K);
      if (acc)
        PLAN_CALL(gcl_conv_fwd_fused_ld(pl ? (const float*)dy.planes : dy.ptr, n_out, pl ? 1 : 0, P.pack_bwd + P.off_bwd[wi], 4,
                                        dy.amax, w_amax, tbl, order, mask, n_in, op.K, op.cout, op.cin, nullptr, nullptr, acc,
                                        acc_ld, 0, nullptr, dx, nullptr, 0, (void*)st));
      else if (P.mask_from[i] >= 0) {
        // x came out of a ReLU that nothing else reads: its backward (g where y > 0) in this epilogue, max|g| published
        dx_amax = new_slot(P);
        PLAN_CALL(gcl_conv_fwd_fused(pl ? (const float*)dy.planes : dy.ptr, n_out, pl ? 1 : 0, P.pack_bwd + P.off_bwd[wi], 4,
                                     dy.amax, w_amax, tbl, order, mask, n_in, op.K, op.cout, op.cin, nullptr, nullptr, x.ptr, 2,
                                     dx_amax, dx, nullptr, 0, (void*)st));
        if (P.relu_masked.size() != P.ops.size()) P.relu_masked.assign(P.ops.size(), 0);
        P.relu_masked[P.mask_from[i]] = 1;
      } else
        PLAN_CALL(gcl_conv_fwd(pl ? (const float*)dy.planes : dy.ptr, n_out, pl ? 1 : 0, P.pack_bwd + P.off_bwd[wi], 4, dy.amax,
                               w_amax, tbl, order, mask, n_in, op.K, op.cout, op.cin, nullptr, dx, nullptr, 0, (void*)st));
    }
    P.g[op.x] = TState();
    P.g[op.x].ptr = dx;
    P.g[op.x].amax = dx_amax;
  }
  const long long rows_len = m.kernel_size == 1 ? gcl_conv_bwd_weight_rows_scratch_len(op.cin, op.cout, 4, n_in) : 0;
  if (rows_len > 0) {      // kernel_size 1: both operands streamed once, no pair list (as ops._ConvFn.backward decides)
    float* scratch = A.take_n<float>(rows_len);
    ProfScope ps(P, ws, 3, pairs, op.cin, op.cout, n_in, n_out, op.K);
    PLAN_CALL(gcl_conv_bwd_weight_rows(x.ptr, dy.ptr, n_in, op.cin, op.cout, 4, P.saved[i].x_amax, dy.amax, scratch, dW,
                                       (void*)ws));
  } else {      // weight gradient over the compacted pair lists
    const int32_t *pa = m.pair_in, *pb = m.pair_out;
    if (op.transpose) { pa = m.pair_out; pb = m.pair_in; }
    GCL_CHECK_ARG(A.dry || (pa && pb), "gcl_plan_backward: record %d needs pair lists the maps do not carry", i);
    const int sorted_side = m.kernel_size == 1 ? 0 : (op.transpose ? 1 : 2);     // the map's out rows ascend per offset
    const long long n_sorted = sorted_side == 1 ? n_in : (sorted_side == 2 ? n_out : 0);
    float* scratch = A.take_n<float>(gcl_conv_bwd_weight_scratch_len(op.K, op.cin, op.cout, m.seg_off[op.K], n_sorted));
    const bool pl = pl_w;
    ProfScope ps(P, ws, 2, pairs, op.cin, op.cout, n_in, n_out, op.K);
    static const bool map_bounds = [] { const char* e = getenv("GCL_DW_MAP_BOUNDS"); return !(e && e[0] == '0'); }();
    // m.dw_bounds: over the map's pair_out with n_out(map) rows -- the sorted list of this launch on either side
    const int32_t* rgb = (map_bounds && sorted_side != 0 && n_sorted == m.n_out) ? m.dw_bounds : nullptr;
    PLAN_CALL(gcl_conv_bwd_weight_rg(pl ? (const float*)x.planes : x.ptr, n_in, pl ? (const float*)dy.planes : dy.ptr, n_out,
                                     pl ? 1 : 0, sorted_side, pa, pb, m.seg_off, op.K, op.cin, op.cout, 4, P.saved[i].x_amax,
                                     dy.amax, scratch, dW, rgb, (void*)ws));
  }
  if (op.bias >= 0) {
    double* scratch = A.take_n<double>(gcl_bn_scratch_len(n_out, op.cout));
    PLAN_CALL(gcl_col_sum(dy.ptr, n_out, op.cout, scratch, (float*)grads[op.bias], (void*)ws));
  }
  return GCL_OK;
}

static int plan_backward(Plan& P, const float* dy, void* const* grads, int first, int last, hipStream_t st) {
  Arena& A = P.A;
  const gcl_maps_desc& M = *P.maps;
  const size_t nw = P.worder.size();
  if (last == (int)P.ops.size()) {
    P.g[P.ops.back().y] = TState();
    P.g[P.ops.back().y].ptr = (float*)dy;
  }
  if (P.bwd_packed && last == (int)P.ops.size()) {
    int rcj = join_aux(P, st);      // packs made on the aux stream during the forward pass
    if (rcj) return rcj;
  }
  if (!P.bwd_packed) {       // input-gradient packs of all kernels in one launch (WeightAmaxGroup.packed("bwd"))
    P.pack_bwd = (unsigned char*)A.take(P.bytes_bwd);
    if (P.n_bwd && !A.dry) {
      const long long* tab = (const long long*)P.state;
      PLAN_CALL(gcl_pack_weights_multi((const int64_t*)(tab + 2 * nw + 8 * nw), P.n_bwd, P.wgs_bwd, 4, P.w_amax, P.pack_bwd,
                                       (void*)st));
    }
    P.bwd_packed = true;
  }
  for (int i = last - 1; i >= first; --i) {
    const gcl_plan_op& op = P.ops[i];
    const long long n_out = M.n_rows[op.level_out], n_in = M.n_rows[op.level_in];
    TState g = P.g[op.y];
    P.g[op.y] = TState();
    if (!g.ptr) continue;
    int rc;
    switch (op.kind) {
      case GCL_OP_CONVBN: {
        const OpSaved& sv = P.saved[i];
        const int c = op.cout;
        float* sum_g = (float*)grads[op.bn_b];
        float* sum_gx = (float*)grads[op.bn_w];
        double* scratch = A.take_n<double>(gcl_bn_scratch_len(n_out, c));
        // bound mode: dx is read as a plane image by the input gradient (and by the weight gradient when both widths allow)
        const bool bound = sv.xrange && c >= P.presplit && !is_stem(op, M.maps[op.map]);
        TState d;
        d.ptr = A.take_n<float>(n_out * c);
        d.amax = new_slot(P);
        PLAN_CALL(gcl_bn_bwd_reduce_range(sv.conv_out, g.ptr, g.ld, nullptr, (const uint64_t*)sv.mask, n_out, c, sv.mean, sv.rstd,
                                          op.relu, scratch, sum_g, sum_gx, sv.xrange, (const float*)P.params[op.bn_w],
                                          bound ? d.amax : nullptr, (void*)st));
        if (bound) d.planes = A.take(n_out * c * 4);
        float* dres = op.x2 >= 0 ? A.take_n<float>(n_out * c) : nullptr;
        PLAN_CALL(gcl_bn_bwd_apply_planes(sv.conv_out, g.ptr, g.ld, nullptr, (const uint64_t*)sv.mask, n_out, c, sv.mean,
                                          sv.rstd, (const float*)P.params[op.bn_w], sum_g, sum_gx, op.relu, d.ptr, dres,
                                          d.amax, d.planes, (void*)st));
        if ((rc = conv_backward(P, i, d, grads, st))) return rc;
        if ((rc = give(P, op.x2, dres, n_out * c, st, nullptr, 0, c))) return rc;
        break;
      }
      case GCL_OP_CONV:
        if ((rc = contiguous(P, g, n_out, op.cout, st))) return rc;
        if ((rc = conv_backward(P, i, g, grads, st))) return rc;
        break;
      case GCL_OP_RELU: {
        if ((size_t)i < P.relu_masked.size() && P.relu_masked[i]) {      // applied by the consumer's input-gradient epilogue
          P.relu_masked[i] = 0;
          if ((rc = give(P, op.x, g.ptr, n_out * op.cout, st, g.amax, g.ld, op.cout))) return rc;
          break;
        }
        if ((rc = contiguous(P, g, n_out, op.cout, st))) return rc;
        const long long numel = n_out * op.cout;
        float* gx = A.take_n<float>(numel);
        int32_t* slot = new_slot(P);
        if (!A.dry) {
          hipLaunchKernelGGL(k_relu_bwd, dim3(grid_for(numel / 4)), dim3(256), 0, st, (const float4*)g.ptr,
                             (const float4*)P.t[op.y].ptr, numel / 4, (float4*)gx, slot);
          GCL_CHECK_LAUNCH();
        }
        if ((rc = give(P, op.x, gx, numel, st, slot, 0, op.cout))) return rc;
        break;
      }
      case GCL_OP_CAT: {
        // the gradients of the two inputs ARE the column slices of g: handed on as views (row pitch = the cat's width); their
        // readers -- the BatchNorm backward kernels, a convolution epilogue that adds a waiting gradient -- take the pitch
        const int ca = op.cin, cb = op.cout - op.cin;
        static const int cat_views = [] { const char* e = getenv("GCL_CAT_VIEWS"); return e ? atoi(e) : 1; }();
        if ((rc = contiguous(P, g, n_out, op.cout, st))) return rc;
        if (cat_views) {
          if ((rc = give(P, op.x, g.ptr, n_out * ca, st, nullptr, op.cout, ca))) return rc;
          if ((rc = give(P, op.x2, g.ptr + ca, n_out * cb, st, nullptr, op.cout, cb))) return rc;
          break;
        }
        float* ga = A.take_n<float>(n_out * ca);
        float* gb = A.take_n<float>(n_out * cb);
        if (!A.dry) {
          hipLaunchKernelGGL(k_split2, dim3(grid_for(n_out * op.cout / 4)), dim3(256), 0, st, (const float4*)g.ptr, ca / 4, cb / 4,
                             n_out, (float4*)ga, (float4*)gb);
          GCL_CHECK_LAUNCH();
        }
        if ((rc = give(P, op.x, ga, n_out * ca, st))) return rc;
        if ((rc = give(P, op.x2, gb, n_out * cb, st))) return rc;
        break;
      }
      case GCL_OP_ROWNORM: {
        if ((rc = contiguous(P, g, n_out, op.cout, st))) return rc;
        float* dx = A.take_n<float>(n_in * op.cin);
        int32_t* slot = new_slot(P);
        PLAN_CALL(gcl_row_normalize_bwd(P.t[op.y].ptr, g.ptr, P.saved[i].norm, n_out, op.cout, dx, slot, (void*)st));
        if ((rc = give(P, op.x, dx, n_in * op.cin, st, slot, 0, op.cin))) return rc;
        break;
      }
      default:
        return GCL_ERR_ARG;
    }
  }
  return join_aux(P, st);      // the caller may hand the gradients of this segment to a collective / the optimizer next
}

static int check_maps(const Plan& P, const gcl_maps_desc* M) {
  GCL_CHECK_ARG(M && M->n_levels >= 1 && M->n_levels <= GCL_MAX_LEVELS && M->n_maps >= 0 && M->n_maps <= GCL_MAX_MAPS,
                "gcl_plan: bad maps descriptor");
  for (size_t i = 0; i < P.ops.size(); ++i) {
    const gcl_plan_op& op = P.ops[i];
    GCL_CHECK_ARG(op.level_in < M->n_levels && op.level_out < M->n_levels, "gcl_plan: record %d uses a level the maps lack", (int)i);
    if (op.kind == GCL_OP_CONVBN || op.kind == GCL_OP_CONV) {
      GCL_CHECK_ARG(op.map >= 0 && op.map < M->n_maps, "gcl_plan: record %d names map %d of %d", (int)i, op.map, M->n_maps);
      const gcl_map_desc& m = M->maps[op.map];
      GCL_CHECK_ARG(m.K == op.K, "gcl_plan: record %d expects K = %d, map %d has %d", (int)i, op.K, op.map, m.K);
      const int lin = op.transpose ? m.level_out : m.level_in, lout = op.transpose ? m.level_in : m.level_out;
      GCL_CHECK_ARG(lin == op.level_in && lout == op.level_out, "gcl_plan: record %d and map %d disagree on the levels", (int)i, op.map);
    }
  }
  return GCL_OK;
}

}  // namespace gcl

using namespace gcl;

extern "C" {

int64_t gcl_maps_arena_bytes(int64_t n, const gcl_map_spec* specs_host, int32_t n_specs, int32_t n_levels) {
  if (n <= 0 || !specs_host || n_specs < 0 || n_specs > GCL_MAX_MAPS || n_levels < 1 || n_levels > GCL_MAX_LEVELS) return -1;
  Arena A{DRY_BASE, 0, 0, true};
  gcl_maps_desc d;
  if (maps_build(nullptr, n, specs_host, n_specs, n_levels, A, nullptr, &d, nullptr) != GCL_OK) return -1;
  return A.off + 4096;
}

int gcl_maps_build_split(const int32_t* coords, int64_t n, const gcl_map_spec* specs_host, int32_t n_specs, int32_t n_levels,
                         void* arena, int64_t arena_bytes, void* pinned_host, gcl_maps_desc* out_host, void* stream,
                         void* side_stream) {
  GCL_CHECK_ARG(coords && specs_host && arena && pinned_host && out_host, "gcl_maps_build: null pointer");
  GCL_CHECK_ARG(side_stream != stream || !side_stream, "gcl_maps_build_split: the side stream must differ from the stream");
  GCL_CHECK_ARG(n > 0, "gcl_maps_build: empty SparseTensor");
  GCL_CHECK_ARG(n_specs >= 0 && n_specs <= GCL_MAX_MAPS && n_levels >= 1 && n_levels <= GCL_MAX_LEVELS,
                "gcl_maps_build: at most %d maps and %d levels", GCL_MAX_MAPS, GCL_MAX_LEVELS);
  // read-back area (int32): row 0 = input status + per-level meta, row s + 1 = the pair counts of map s; 128 words per row
  // size check BEFORE the first launch (the dry run's upper bound -- every level as large as the input -- is what
  // gcl_maps_arena_bytes tells callers to reserve): a short arena must not be written past its end
  const int64_t need = gcl_maps_arena_bytes(n, specs_host, n_specs, n_levels);
  GCL_CHECK_ARG(need >= 0, "gcl_maps_build: bad map specification");
  if (arena_bytes < need) {
    set_error("gcl_maps_build: arena too small (%lld bytes needed, %lld given)", (long long)need, (long long)arena_bytes);
    return GCL_ERR_ARENA;
  }
  Arena A{(char*)arena, arena_bytes, 0, false};
  return maps_build(coords, n, specs_host, n_specs, n_levels, A, (int32_t*)pinned_host, out_host, (hipStream_t)stream,
                    (hipStream_t)side_stream);
}

int gcl_maps_build(const int32_t* coords, int64_t n, const gcl_map_spec* specs_host, int32_t n_specs, int32_t n_levels,
                   void* arena, int64_t arena_bytes, void* pinned_host, gcl_maps_desc* out_host, void* stream) {
  return gcl_maps_build_split(coords, n, specs_host, n_specs, n_levels, arena, arena_bytes, pinned_host, out_host, stream, nullptr);
}

void* gcl_plan_create(const gcl_plan_op* ops_host, int32_t n_ops, int32_t n_tensors, int32_t n_params,
                      const int32_t* weight_order_host, int32_t n_weights, int32_t presplit_min_c) {
  if (!ops_host || n_ops <= 0 || n_tensors <= 1 || n_params <= 0 || n_weights < 0 || (n_weights && !weight_order_host)) {
    set_error("gcl_plan_create: bad argument");
    return nullptr;
  }
  Plan* P = new (std::nothrow) Plan();
  if (!P) {
    set_error("gcl_plan_create: out of host memory");
    return nullptr;
  }
  P->ops.assign(ops_host, ops_host + n_ops);
  P->n_tensors = n_tensors;
  P->n_params = n_params;
  P->presplit = presplit_min_c > 0 ? presplit_min_c : 128;
  P->worder.assign(weight_order_host, weight_order_host + n_weights);
  P->widx.assign(n_params, -1);
  P->made.assign(n_tensors, 0);
  const size_t nw = (size_t)n_weights;
  P->wmode.assign(nw, 0);
  P->wK.assign(nw, 0);
  P->wcin.assign(nw, 0);
  P->wcout.assign(nw, 0);
  P->off_fwd.assign(nw, 0);
  P->off_bwd.assign(nw, 0);
  bool ok = true;
  for (int q = 0; q < n_weights && ok; ++q) {
    const int p = weight_order_host[q];
    ok = p >= 0 && p < n_params && P->widx[p] < 0;
    if (ok) P->widx[p] = q;
  }
  std::vector<char> seen_param(n_params, 0);
  for (int i = 0; i < n_ops && ok; ++i) {
    const gcl_plan_op& op = P->ops[i];
    auto tensor_ok = [&](int t) { return t >= 0 && t < n_tensors; };
    ok = tensor_ok(op.x) && tensor_ok(op.y) && op.y != 0 && !P->made[op.y] && (op.x == 0 || P->made[op.x]) &&
         op.level_in >= 0 && op.level_in < GCL_MAX_LEVELS && op.level_out >= 0 && op.level_out < GCL_MAX_LEVELS &&
         op.cin > 0 && op.cout > 0;
    if (!ok) break;
    if (op.x2 >= 0) ok = tensor_ok(op.x2) && P->made[op.x2];
    if (!ok) break;
    auto claim = [&](int p) {       // every parameter belongs to exactly one record (its gradient is written, not added)
      if (p < 0 || p >= n_params || seen_param[p]) return false;
      seen_param[p] = 1;
      return true;
    };
    if (op.kind == GCL_OP_CONVBN || op.kind == GCL_OP_CONV) {
      ok = claim(op.w) && op.map >= 0 && op.map < GCL_MAX_MAPS && op.K >= 1 && op.K <= 125;
      if (ok && op.bias >= 0) ok = claim(op.bias) && op.kind == GCL_OP_CONV;
      if (ok && op.kind == GCL_OP_CONVBN) ok = claim(op.bn_w) && claim(op.bn_b) && op.bn >= 0 && op.cout % 4 == 0;
      if (ok && op.kind == GCL_OP_CONVBN) P->n_bn = op.bn + 1 > P->n_bn ? op.bn + 1 : P->n_bn;
      if (!ok) break;
      const bool stem_shape = op.cin <= 4 && (op.cout % 32) == 0 && !op.transpose && op.K > 1 && op.bias < 0;
      if (!stem_shape) {
        // MFMA-shaped kernels only (the generic VALU shapes stay on the per-operator path)
        ok = (op.cin % 32) == 0 && (op.cout % 32) == 0 && op.K <= 27 && P->widx[op.w] >= 0;
        if (!ok) break;
        const int q = P->widx[op.w];
        P->wK[q] = op.K; P->wcin[q] = op.cin; P->wcout[q] = op.cout;
        // input-gradient layout recorded by the forward pass: mirrored offsets on a stride-1 map, plain transpose otherwise
        // (the caller tells stride-1 3^3 / 5^3 maps apart through `transpose` and level_in == level_out)
        if (op.x != 0) P->wmode[q] = (op.K > 1 && !op.transpose && op.level_in == op.level_out) ? 2 : 1;
      } else {
        ok = op.x == 0;      // the Cin <= 4 first layer has no input gradient
      }
    } else if (op.kind == GCL_OP_CAT) {
      ok = op.x2 >= 0 && op.cin % 4 == 0 && (op.cout - op.cin) > 0 && (op.cout - op.cin) % 4 == 0 && op.level_in == op.level_out;
    } else if (op.kind == GCL_OP_RELU || op.kind == GCL_OP_ROWNORM) {
      ok = op.cin == op.cout && op.cout % 4 == 0 && op.level_in == op.level_out;
    } else {
      ok = false;
    }
    P->made[op.y] = 1;
  }
  // CONV directly followed by the RELU that is its only consumer (model/resunet.py:222-224: conv1_tr, MEF.relu, final): the
  // convolution's epilogue applies the ReLU and publishes max|y| (GCL_PLAN_FUSE_RELU=0: two launches, same values)
  P->fuse_relu.assign(P->ops.size(), 0);
  static const bool fuse_relu_on = [] { const char* e = getenv("GCL_PLAN_FUSE_RELU"); return !(e && e[0] == '0'); }();
  for (size_t i = 0; ok && fuse_relu_on && i + 1 < P->ops.size(); ++i) {
    const gcl_plan_op &a = P->ops[i], &b = P->ops[i + 1];
    if (a.kind != GCL_OP_CONV || b.kind != GCL_OP_RELU || b.x != a.y || a.cin <= 4) continue;
    bool other = a.y == P->ops.back().y;       // the plan's output keeps its own tensor
    for (size_t j = 0; j < P->ops.size(); ++j)
      if (j != i + 1 && (P->ops[j].x == a.y || P->ops[j].x2 == a.y)) other = true;
    if (!other) P->fuse_relu[i] = 1;
  }
  P->cat_left.assign(P->ops.size(), -1);
  static const bool cat_inplace = [] { const char* e = getenv("GCL_CAT_INPLACE"); return !(e && e[0] == '0'); }();
  for (size_t i = 0; ok && cat_inplace && i < P->ops.size(); ++i) {
    const gcl_plan_op& a = P->ops[i];
    if (a.kind != GCL_OP_CONVBN || a.cin <= 4 || a.y == P->ops.back().y) continue;
    int cat = -1, consumers = 0;
    for (size_t j = 0; j < P->ops.size(); ++j) {
      if (P->ops[j].kind == GCL_OP_CAT && P->ops[j].x == a.y && P->ops[j].x2 != a.y && j > i) cat = (int)j;
      consumers += (P->ops[j].x == a.y) + (P->ops[j].x2 == a.y);
    }
    if (cat >= 0 && consumers == 1) P->cat_left[i] = cat;
  }
  P->mask_from.assign(P->ops.size(), -1);
  for (size_t i = 0; ok && fuse_relu_on && i < P->ops.size(); ++i) {
    const gcl_plan_op& a = P->ops[i];
    if ((a.kind != GCL_OP_CONV && a.kind != GCL_OP_CONVBN) || a.cin <= 4 || a.x == 0) continue;
    int r = -1, consumers = 0;
    for (size_t j = 0; j < P->ops.size(); ++j) {
      if (P->ops[j].kind == GCL_OP_RELU && P->ops[j].y == a.x) r = (int)j;
      consumers += (P->ops[j].x == a.x) + (P->ops[j].x2 == a.x);
    }
    if (r >= 0 && consumers == 1 && a.x != P->ops.back().y) P->mask_from[i] = r;
  }
  if (!ok) {
    set_error("gcl_plan_create: malformed or unsupported operator records");
    delete P;
    return nullptr;
  }
  for (size_t q = 0; q < nw; ++q) {
    if (!P->wK[q]) {
      set_error("gcl_plan_create: weight_order names a parameter no record uses");
      delete P;
      return nullptr;
    }
    const long long bytes = (gcl_pack_weights_bytes((int)P->wK[q], (int)P->wcin[q], (int)P->wcout[q], 4) + 255) / 256 * 256;
    P->off_fwd[q] = P->bytes_fwd;
    P->bytes_fwd += bytes;
    if (P->wmode[q]) {
      P->off_bwd[q] = P->bytes_bwd;
      P->bytes_bwd += bytes;
      ++P->n_bwd;
    }
  }
  return P;
}

void gcl_plan_destroy(void* plan) {
  Plan* P = (Plan*)plan;
  if (!P) return;
  for (ProfRec& r : P->prof) {
    (void)hipEventDestroy(r.e0);
    (void)hipEventDestroy(r.e1);
  }
  for (PassState* q : P->passes) delete q;
  for (hipEvent_t e : P->events) (void)hipEventDestroy(e);
  delete P;
}

int64_t gcl_plan_state_bytes(const void* plan) {
  if (!plan) return -1;
  return state_words(*(const Plan*)plan) * 8 + 256;
}

int64_t gcl_plan_arena_bytes(void* plan, const gcl_maps_desc* maps_host) {
  Plan* P = (Plan*)plan;
  if (!P || check_maps(*P, maps_host) != GCL_OK) return -1;
  P->maps = maps_host;
  P->A = Arena{DRY_BASE, 0, 0, true};
  P->params.assign(P->n_params, nullptr);
  std::vector<void*> nulls(2 * (P->n_bn > 0 ? P->n_bn : 1), nullptr), gn(P->n_params, nullptr);
  float* y = nullptr;
  const bool prof = P->profile;
  P->profile = false;
  int rc = plan_forward(*P, nullptr, nulls.data(), &y, nullptr);
  if (rc == GCL_OK) rc = plan_backward(*P, (const float*)DRY_BASE, gn.data(), 0, (int)P->ops.size(), nullptr);
  P->profile = prof;
  P->forward_done = false;
  return rc == GCL_OK ? P->A.off + 4096 : -1;
}

int gcl_plan_forward(void* plan, const gcl_maps_desc* maps_host, const float* x, void* const* params_host,
                     void* const* bn_stats_host, void* state, void* arena, int64_t arena_bytes, float** y_out_host,
                     void* stream) {
  Plan* P = (Plan*)plan;
  GCL_CHECK_ARG(P && maps_host && x && params_host && bn_stats_host && state && arena && y_out_host, "gcl_plan_forward: null pointer");
  int rc = check_maps(*P, maps_host);
  if (rc) return rc;
  P->maps_copy = *maps_host;        // the backward pass runs after the caller may have released its descriptor
  P->maps = &P->maps_copy;
  P->params.assign(params_host, params_host + P->n_params);
  P->state = state;
  P->A = Arena{(char*)arena, arena_bytes, 0, false};
  P->forward_done = false;
  // size check first (cheap dry run of both passes): never launch into an arena that cannot hold the pass
  {
    Arena real = P->A;
    P->A = Arena{DRY_BASE, 0, 0, true};
    std::vector<void*> gn(P->n_params, nullptr);
    float* yy = nullptr;
    const bool prof = P->profile;
    P->profile = false;
    rc = plan_forward(*P, x, bn_stats_host, &yy, nullptr);
    if (rc == GCL_OK) rc = plan_backward(*P, (const float*)DRY_BASE, gn.data(), 0, (int)P->ops.size(), nullptr);
    P->profile = prof;
    const long long need = P->A.off;
    P->A = real;
    if (rc) return rc;
    if (need > arena_bytes) {
      set_error("gcl_plan_forward: arena too small (%lld bytes needed, %lld given)", need, (long long)arena_bytes);
      return GCL_ERR_ARENA;
    }
  }
  rc = plan_forward(*P, x, bn_stats_host, y_out_host, (hipStream_t)stream);
  P->forward_done = rc == GCL_OK;
  if (rc != GCL_OK) return rc;
  // park the pass under its arena until gcl_plan_backward / gcl_plan_release asks for it
  PassState* slot = nullptr;
  for (PassState* q : P->passes)
    if (!q->forward_done || q->key == arena) slot = q;
  if (!slot) {
    slot = new (std::nothrow) PassState();
    GCL_CHECK_ARG(slot, "gcl_plan_forward: out of host memory");
    P->passes.push_back(slot);
  }
  P->key = arena;
  *slot = std::move(static_cast<PassState&>(*P));
  slot->maps = &slot->maps_copy;
  static_cast<PassState&>(*P) = PassState();
  return GCL_OK;
}

int64_t gcl_plan_eval_state_bytes(const void* plan) {
  if (!plan) return -1;
  const Plan& P = *(const Plan*)plan;
  return ((state_words(P) * 8 + 255) & ~255ll) + (((long long)P.worder.size() * GCL_AMAX_WORDS * 4 + 255) & ~255ll) +
         P.bytes_fwd + 256;
}

int64_t gcl_plan_eval_arena_bytes(void* plan, const gcl_maps_desc* maps_host) {
  Plan* P = (Plan*)plan;
  if (!P || check_maps(*P, maps_host) != GCL_OK) return -1;
  P->maps = maps_host;
  P->A = Arena{DRY_BASE, 0, 0, true};
  P->params.assign(P->n_params, nullptr);
  P->eval = true;
  P->bn_eval = nullptr;
  P->state = DRY_BASE;
  std::vector<void*> nulls(2 * (P->n_bn > 0 ? P->n_bn : 1), nullptr);
  float* y = nullptr;
  const bool prof = P->profile;
  P->profile = false;
  int rc = plan_forward(*P, nullptr, nulls.data(), &y, nullptr);
  P->profile = prof;
  P->eval = false;
  P->forward_done = false;
  return rc == GCL_OK ? P->A.off + 4096 : -1;
}

int gcl_plan_forward_eval(void* plan, const gcl_maps_desc* maps_host, const float* x, void* const* params_host,
                          void* const* bn_eval_host, int32_t repack, void* state, void* arena, int64_t arena_bytes,
                          float** y_out_host, void* stream) {
  Plan* P = (Plan*)plan;
  GCL_CHECK_ARG(P && maps_host && x && params_host && bn_eval_host && state && arena && y_out_host,
                "gcl_plan_forward_eval: null pointer");
  int rc = check_maps(*P, maps_host);
  if (rc) return rc;
  P->maps_copy = *maps_host;
  P->maps = &P->maps_copy;
  P->params.assign(params_host, params_host + P->n_params);
  P->state = state;
  P->eval = true;
  P->bn_eval = bn_eval_host;
  P->eval_repack = repack != 0;
  std::vector<void*> nulls(2 * (P->n_bn > 0 ? P->n_bn : 1), nullptr);
  P->A = Arena{DRY_BASE, 0, 0, true};      // size check first
  float* yy = nullptr;
  const bool prof = P->profile;
  P->profile = false;
  rc = plan_forward(*P, x, nulls.data(), &yy, nullptr);
  P->profile = prof;
  const long long need = P->A.off;
  if (rc == GCL_OK && need > arena_bytes) {
    set_error("gcl_plan_forward_eval: arena too small (%lld bytes needed, %lld given)", need, (long long)arena_bytes);
    rc = GCL_ERR_ARENA;
  }
  if (rc == GCL_OK) {
    P->A = Arena{(char*)arena, arena_bytes, 0, false};
    rc = plan_forward(*P, x, nulls.data(), y_out_host, (hipStream_t)stream);
  }
  if (rc == GCL_OK && P->slots_exhausted) {
    set_error("gcl_plan_forward_eval: amax slot pool exhausted");
    rc = GCL_ERR_ARG;
  }
  P->eval = false;
  P->bn_eval = nullptr;
  P->forward_done = false;
  static_cast<PassState&>(*P) = PassState();      // nothing waits for a backward pass
  return rc;
}

int gcl_plan_release(void* plan, void* arena) {
  Plan* P = (Plan*)plan;
  GCL_CHECK_ARG(P, "gcl_plan_release: null plan");
  bool found = false;
  for (PassState* q : P->passes)
    if (q->forward_done && q->key == arena) found = true;
  // the forward pass forked packs / max|W| onto the aux stream; they are joined at the head of gcl_plan_backward, which
  // never comes for this pass.  The caller frees the arena to the MAIN stream's allocator next: drain the aux stream
  // first (rare path -- a training-mode pass under no_grad, or an output dropped without backward)
  if (found && P->aux && P->aux_dirty) {
    GCL_CHECK_HIP(hipStreamSynchronize(P->aux));
    P->aux_dirty = false;
  }
  for (PassState* q : P->passes)
    if (q->forward_done && q->key == arena) *q = PassState();
  return GCL_OK;
}

int gcl_plan_backward(void* plan, void* arena, const float* dy, void* const* grads_host, int32_t first_op,
                      int32_t last_op, void* stream) {
  Plan* P = (Plan*)plan;
  GCL_CHECK_ARG(P && grads_host, "gcl_plan_backward: null pointer");
  PassState* slot = nullptr;
  for (PassState* q : P->passes)
    if (q->forward_done && q->key == arena) slot = q;
  GCL_CHECK_ARG(slot, "gcl_plan_backward: no forward pass to differentiate in this arena");
  GCL_CHECK_ARG(first_op >= 0 && first_op < last_op && last_op <= (int)P->ops.size(), "gcl_plan_backward: bad record range");
  GCL_CHECK_ARG(last_op != (int)P->ops.size() || dy, "gcl_plan_backward: the last segment needs dy");
  static_cast<PassState&>(*P) = std::move(*slot);
  P->maps = &P->maps_copy;
  int rc = plan_backward(*P, dy, grads_host, first_op, last_op, (hipStream_t)stream);
  if (rc == GCL_OK && !P->A.fits()) {
    set_error("gcl_plan_backward: arena overrun");
    rc = GCL_ERR_ARENA;
  }
  if (rc == GCL_OK && P->slots_exhausted) {
    set_error("gcl_plan_backward: amax slot pool exhausted");
    rc = GCL_ERR_ARG;
  }
  if (first_op == 0 || rc != GCL_OK) P->forward_done = false;      // the pass is finished (or broken): its slot is free again
  *slot = std::move(static_cast<PassState&>(*P));
  slot->maps = &slot->maps_copy;
  static_cast<PassState&>(*P) = PassState();
  return rc;
}

int gcl_plan_set_aux_stream(void* plan, void* stream) {
  Plan* P = (Plan*)plan;
  GCL_CHECK_ARG(P, "gcl_plan_set_aux_stream: null plan");
  P->aux = (hipStream_t)stream;
  return GCL_OK;
}

// A HIP stream restricted to a share of the chip's CUs (hipExtStreamCreateWithCUMask): the lowest `percent` % of the mask
// bits.  Experiment hook for the weight-gradient stream (GCL_AUX_CU_PCT, native.py): how much of the main chain's slowdown
// beside the weight gradients is bought back by giving them fewer CUs.
int gcl_stream_create_cu_share(int32_t percent, int32_t low_priority, void** stream_out) {
  GCL_CHECK_ARG(stream_out && percent >= 1 && percent <= 100, "gcl_stream_create_cu_share: percent must be 1..100");
  int dev = 0;
  hipDeviceProp_t prop;
  GCL_CHECK_HIP(hipGetDevice(&dev));
  GCL_CHECK_HIP(hipGetDeviceProperties(&prop, dev));
  const int n_cu = prop.multiProcessorCount, words = (n_cu + 31) / 32;
  int on = (int)((long long)n_cu * percent / 100);
  if (on < 8) on = 8;
  std::vector<uint32_t> mask((size_t)words, 0u);
  for (int i = 0; i < on && i < n_cu; ++i) mask[(size_t)i >> 5] |= 1u << (i & 31);
  hipStream_t st = nullptr;
  GCL_CHECK_HIP(hipExtStreamCreateWithCUMask(&st, (uint32_t)words, mask.data()));
  (void)low_priority;      // hipExtStreamCreateWithCUMask has no priority argument
  *stream_out = (void*)st;
  return GCL_OK;
}

int gcl_stream_destroy(void* stream) {
  if (stream) GCL_CHECK_HIP(hipStreamDestroy((hipStream_t)stream));
  return GCL_OK;
}

int gcl_plan_profile(void* plan, int32_t enable) {
  Plan* P = (Plan*)plan;
  GCL_CHECK_ARG(P, "gcl_plan_profile: null plan");
  P->profile = enable != 0;
  return GCL_OK;
}

int gcl_plan_profile_read(void* plan, double* records_host, int32_t max_records) {
  Plan* P = (Plan*)plan;
  if (!P || !records_host) return GCL_ERR_ARG;
  int n = 0;
  for (size_t i = 0; i < P->prof_used && n < max_records; ++i) {
    const ProfRec& r = P->prof[i];
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess) continue;
    double* o = records_host + 8 * n++;
    o[0] = r.kind; o[1] = ms; o[2] = r.pairs; o[3] = r.cin; o[4] = r.cout; o[5] = r.n_in; o[6] = r.n_out; o[7] = r.K;
  }
  return n;
}

}  // extern "C"
