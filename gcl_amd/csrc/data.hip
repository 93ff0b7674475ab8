// "Next" rows of the scope table: the loader work either side of the hot path, on the GPU.
//
//  * gcl_voxelize            = ME.utils.sparse_quantize (util/misc.py:118, lib/colocation_data_loader.py:379,388):
//                              floor(xyz / voxel) -> one row per voxel (first occurrence, ascending row order).
//  * gcl_colocation_hits/emit = get_matching_indices_colocation (util/pointcloud.py:69-132): for every centre voxel,
//                              the <= K nearest points within `radius` in the centre cloud (itself first) and in every
//                              neighbour cloud, the group size, the member rows and the "finest" flag.
//    The reference runs one open3d KD-tree radius query per point per cloud in a Python loop; here every voxelised
//    cloud already has a coordinate hash map (one point per voxel), so a radius query is a scan of the (2R+1)^3
//    voxels around the query point, R = floor(radius / voxel) + 1.
// Integer / index work: results are compared bit-exactly with the CPU oracle (distances are evaluated in fp64 from the
// same fp32 inputs and without fused multiply-add, like the oracle's KD-tree).
#include "common.h"

#include <limits.h>

namespace gcl {

__device__ __forceinline__ int floor_to_int(float v) { return (int)floorf(v); }

__global__ void k_voxel_coords(const float* __restrict__ xyz, long long p, float voxel, int batch_id, int4* coords) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= p) return;
  // floor(xyz / voxel_size) with the correctly rounded fp32 division numpy performs (util/misc.py:117)
  coords[i] = make_int4(batch_id, floor_to_int(__fdiv_rn(xyz[3 * i], voxel)), floor_to_int(__fdiv_rn(xyz[3 * i + 1], voxel)),
                        floor_to_int(__fdiv_rn(xyz[3 * i + 2], voxel)));
}

// ---- co-location groups ------------------------------------------------------------------------------------------
constexpr int KMAX = 8;

struct Affine {
  double m[12];   // row-major 3x4: q = R p + t
};
struct Affines {
  Affine a[16];
};

// hits of centre point i in cloud c: up to K nearest within radius, ascending (d2, row).
// HITS_LANES lanes share one (centre point, cloud) query: the (2R+1)^3 candidate voxels are dealt round-robin, every lane keeps
// the K best of ITS candidates in registers, and K rounds of "smallest head over the lanes" merge them -- (d2, row) is a total
// order (a row appears once), so the result is the list one lane walking all candidates would keep.  One lane per query
// (rounds 3-5) was a chain of ~ 125 dependent probes on 2 waves per SIMD: 285 us per sample at 19 k centre voxels x 7 clouds.
constexpr int HITS_LANES = 8;
__global__ void __launch_bounds__(256) k_colocation_hits(const float* __restrict__ xyz_own,   // [Ntot,3] own frames
                                                         const float* __restrict__ xyz_cf,    // [Ntot,3] centre frame
                                                         long long n_center, int n_clouds, Affines to_cloud,
                                                         const Slot* __restrict__ table, long long cap,
                                                         float inv_voxel, double radius, int R, int K,
                                                         int* __restrict__ hits,        // [n_center, n_clouds, K]
                                                         int* __restrict__ cnt,         // [n_center, n_clouds]
                                                         double* __restrict__ first_rng,    // [n_center, n_clouds]
                                                         long long row0, int cloud0) {
  // row0 / cloud0 (batch form): the sample's centre voxels are rows row0 .. of xyz_*, its clouds carry the table's batch ids
  // cloud0 .. cloud0 + n_clouds - 1, and the table's values -- the hits written -- are rows of the whole batch
  const int sub = threadIdx.x & (HITS_LANES - 1);
  long long i = (long long)blockIdx.x * (256 / HITS_LANES) + (threadIdx.x / HITS_LANES);
  const int c = blockIdx.y;
  const bool live = i < n_center;
  if (!live) i = n_center - 1;     // the lanes of a dead query run along (their shuffles must execute) and write nothing
  const long long ic = row0 + i;
  const double px = xyz_cf[3 * ic], py = xyz_cf[3 * ic + 1], pz = xyz_cf[3 * ic + 2];
  // query point in the frame of cloud c (only to find candidate voxels)
  const double* m = to_cloud.a[c].m;
  float qx = (float)(m[0] * px + m[1] * py + m[2] * pz + m[3]);
  float qy = (float)(m[4] * px + m[5] * py + m[6] * pz + m[7]);
  float qz = (float)(m[8] * px + m[9] * py + m[10] * pz + m[11]);
  int bx = floor_to_int(qx * inv_voxel), by = floor_to_int(qy * inv_voxel), bz = floor_to_int(qz * inv_voxel);
  double bd[KMAX];
  int bi[KMAX];
  int n = 0;
  const double r2 = __dmul_rn(radius, radius);
  // Round 6: a voxel whose BOX is farther from the query than the radius cannot hold a hit (its point lies inside the box,
  // and the neighbour -> centre transforms are rigid): skipped before its table probe -- ~ 60 % of the (2R+1)^3 voxels at
  // radius 1.5 voxels.  Per-axis gaps in voxel units, shortened by 0.004 voxels: the query's position in the cloud's frame is
  // rounded (fp32 of an fp64 affine, ~ 3e-5 voxels at 100 m) and the test must only ever err towards probing.
  const float fx = qx * inv_voxel - (float)bx, fy = qy * inv_voxel - (float)by, fz = qz * inv_voxel - (float)bz;
  const float rv = (float)radius * inv_voxel, rv2 = rv * rv * 1.0001f;
  auto gap2 = [](int d, float f) {
    const float g = fmaxf((d > 0 ? (float)d - f : (d < 0 ? f - (float)(d + 1) : 0.f)) - 0.004f, 0.f);
    return g * g;
  };
  const int W = 2 * R + 1, W3 = W * W * W;
  const unsigned MW = 65535u / (unsigned)W + 1u;      // (v * MW) >> 16 == v / W for v < 729 (W <= 9): no integer division
  // pass 1 (no memory): which of this lane's candidates survive the box test -- a bit each (W3 <= 729: at most 92 per lane);
  // pass 2 probes only those, so a wave runs max-over-lanes(survivors) probe iterations instead of all W3 / 8
  unsigned keep[3] = {0u, 0u, 0u};
  {
    int slot = 0;
    for (int cand = sub; cand < W3; cand += HITS_LANES, ++slot) {
      const unsigned t = ((unsigned)cand * MW) >> 16, uz = (t * MW) >> 16;
      const int dz = (int)uz - R, dy = (int)(t - uz * W) - R, dx = (int)((unsigned)cand - t * W) - R;
      if (gap2(dx, fx) + gap2(dy, fy) + gap2(dz, fz) > rv2) continue;
      if (!pack_ok(cloud0 + c, bx + dx, by + dy, bz + dz)) continue;
      const unsigned bit = 1u << (slot & 31);
      if (slot < 32) keep[0] |= bit;
      else if (slot < 64) keep[1] |= bit;
      else keep[2] |= bit;
    }
  }
#pragma unroll
  for (int wd = 0; wd < 3; ++wd)
  for (unsigned rest = keep[wd]; rest; rest &= rest - 1) {
    const int cand = sub + HITS_LANES * (32 * wd + __builtin_ctz(rest));
    const unsigned t = ((unsigned)cand * MW) >> 16, uz = (t * MW) >> 16;
    const int dz = (int)uz - R, dy = (int)(t - uz * W) - R, dx = (int)((unsigned)cand - t * W) - R;
    int x = bx + dx, y = by + dy, z = bz + dz;
    const unsigned long long key = pack_key(cloud0 + c, x, y, z);
    long long s = table_find(table, cap, key);
    if (s < 0) continue;
    int q = (int)table[s].val;
    double ex = __dsub_rn((double)xyz_cf[3 * (long long)q], px);
    double ey = __dsub_rn((double)xyz_cf[3 * (long long)q + 1], py);
    double ez = __dsub_rn((double)xyz_cf[3 * (long long)q + 2], pz);
    double d2 = __dadd_rn(__dadd_rn(__dmul_rn(ex, ex), __dmul_rn(ey, ey)), __dmul_rn(ez, ez));
    if (!(d2 < r2)) continue;   // strictly inside, as a KD-tree query with distance_upper_bound
    // keep the K best, ascending (d2, row)
    if (n == K && !(d2 < bd[K - 1] || (d2 == bd[K - 1] && q < bi[K - 1]))) continue;
    int pos = (n < K) ? n : K - 1;
    while (pos > 0 && (bd[pos - 1] > d2 || (bd[pos - 1] == d2 && bi[pos - 1] > q))) {
      bd[pos] = bd[pos - 1];
      bi[pos] = bi[pos - 1];
      --pos;
    }
    bd[pos] = d2;
    bi[pos] = q;
    if (n < K) ++n;
  }
  // merge: K rounds; every lane offers the head of its list, the smallest (d2, row) over the query's lanes is the next hit
  constexpr int NOROW = 0x7fffffff;
#pragma unroll
  for (int j = 0; j < KMAX; ++j)
    if (j >= n) {
      bd[j] = 1e300;
      bi[j] = NOROW;
    }
  int merged = 0, first = -1;
  const long long o = (i * n_clouds + c);
  for (int j = 0; j < K; ++j) {
    double hd = bd[0];
    int hq = bi[0];
#pragma unroll
    for (int w = 1; w < HITS_LANES; w <<= 1) {
      const double od = __shfl_xor(hd, w, HITS_LANES);
      const int oq = __shfl_xor(hq, w, HITS_LANES);
      if (od < hd || (od == hd && oq < hq)) {
        hd = od;
        hq = oq;
      }
    }
    if (hq != NOROW) {
      if (j == 0) first = hq;
      ++merged;
      if (hq == bi[0]) {       // this lane's head won: pop it
#pragma unroll
        for (int t = 0; t + 1 < KMAX; ++t) {
          bd[t] = bd[t + 1];
          bi[t] = bi[t + 1];
        }
        bd[KMAX - 1] = 1e300;
        bi[KMAX - 1] = NOROW;
      }
    }
    if (live && sub == 0) hits[o * K + j] = hq != NOROW ? hq : -1;
  }
  if (!live || sub != 0) return;
  cnt[o] = merged;
  double rng = 1e300;
  if (c == 0) {
    double a = xyz_own[3 * ic], b = xyz_own[3 * ic + 1], d = xyz_own[3 * ic + 2];
    rng = sqrt(__dadd_rn(__dadd_rn(__dmul_rn(a, a), __dmul_rn(b, b)), __dmul_rn(d, d)));   // centre voxel's sensor range
  } else if (merged > 0) {
    long long q = first;
    double a = xyz_own[3 * q], b = xyz_own[3 * q + 1], d = xyz_own[3 * q + 2];
    rng = sqrt(__dadd_rn(__dadd_rn(__dmul_rn(a, a), __dmul_rn(b, b)), __dmul_rn(d, d)));   // nearest hit, ITS frame
  }
  first_rng[o] = rng;
}

// group size of every centre point (0 = no neighbour-cloud match => no group) and its kept flag
__global__ void k_colocation_sizes(const int* __restrict__ cnt, long long n_center, int n_clouds, int* gsize, int* kept) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_center) return;
  int tot = 0, ng = 0;
  for (int c = 0; c < n_clouds; ++c) {
    int v = cnt[i * n_clouds + c];
    tot += v;
    if (c > 0) ng += v;
  }
  gsize[i] = ng > 0 ? tot : 0;
  kept[i] = ng > 0 ? 1 : 0;
}

__global__ void k_colocation_emit(const int* __restrict__ hits, const int* __restrict__ cnt,
                                  const double* __restrict__ first_rng, long long n_center, int n_clouds, int K,
                                  const int* __restrict__ gsize, const int* __restrict__ goff,
                                  const int* __restrict__ gidx, int* group, long long* index, unsigned char* finest,
                                  int* totals) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_center) return;
  if (i == n_center - 1) {
    totals[0] = gidx[i] + (gsize[i] > 0 ? 1 : 0);   // number of groups
    totals[1] = goff[i] + gsize[i];                 // number of index entries
  }
  if (gsize[i] == 0) return;
  // finest member: first arg-min over [centre range, nearest-hit range of cloud 1, 2, ...] (strict < in the reference)
  int best = 0;
  double bv = first_rng[i * n_clouds];
  for (int c = 1; c < n_clouds; ++c) {
    double v = first_rng[i * n_clouds + c];
    if (cnt[i * n_clouds + c] > 0 && v < bv) {
      bv = v;
      best = c;
    }
  }
  long long o = goff[i];
  int fpos = 0, run = 0;
  for (int c = 0; c < n_clouds; ++c) {
    int n = cnt[i * n_clouds + c];
    if (c == best) fpos = (c == 0) ? 0 : run;
    for (int j = 0; j < n; ++j) index[o + run + j] = hits[(i * n_clouds + c) * K + j];
    run += n;
  }
  for (int j = 0; j < run; ++j) finest[o + j] = (j == fpos) ? 1 : 0;
  group[gidx[i]] = run;
}


// ---- a whole BATCH of samples in one pass (round 6): every cloud of every sample voxelised, deduplicated and looked up
// through ONE coordinate table whose batch id is the cloud's number in the batch (lib/colocation_data_loader.py:440-446 gives
// the clouds of a batch consecutive ids too), so that the loader makes two host synchronisations per batch instead of ~ 9 per
// sample, and no host round trip for the neighbours' centre-frame points ---------------------------------------------------
struct CloudOffsets {
  long long off[65];      // point offsets of up to 64 clouds + the total
};
__global__ void k_voxel_coords_multi(const float* __restrict__ xyz, long long p, CloudOffsets co, int n_clouds, float voxel,
                                     int4* coords) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= p) return;
  int lo = 0, hi = n_clouds - 1;      // the cloud of point i: last c with off[c] <= i
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (co.off[mid] <= i) lo = mid; else hi = mid - 1;
  }
  coords[i] = make_int4(lo, floor_to_int(__fdiv_rn(xyz[3 * i], voxel)), floor_to_int(__fdiv_rn(xyz[3 * i + 1], voxel)),
                        floor_to_int(__fdiv_rn(xyz[3 * i + 2], voxel)));
}

// starts[c] = first unique row of cloud c (the unique rows keep the input's cloud-major order), starts[n_clouds] = n;
// a cloud without rows gets the start of the next one (filled by the host from the right)
__global__ void k_cloud_row_starts(const int4* __restrict__ coords, const int* __restrict__ n_dev, int n_clouds, int* starts) {
  const long long n = *n_dev;
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) starts[n_clouds] = (int)n;
  if (i >= n) return;
  const int b = coords[i].x;
  if (i == 0 || coords[i - 1].x != b) starts[b] = (int)i;
}

// voxel representatives: the first point of every voxel in its OWN sensor frame (xyz_raw[index]) and in the frame of its
// sample's centre cloud -- q = fp32(R p + t) in fp64 from the fp32 point, as the host path's numpy expression
// (colocation_data_gpu.build_sample_gpu; the reference transforms fp64 copies: util/pointcloud.py:87-89)
__global__ void k_loader_points(const float* __restrict__ xyz_raw, const long long* __restrict__ index,
                                const int4* __restrict__ coords, const int* __restrict__ n_dev,
                                const double* __restrict__ to_center,      // [n_clouds][12], row-major 3 x 4
                                float* __restrict__ xyz_own, float* __restrict__ xyz_cf) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= *n_dev) return;
  const long long r = index[i];
  const float x = xyz_raw[3 * r], y = xyz_raw[3 * r + 1], z = xyz_raw[3 * r + 2];
  xyz_own[3 * i] = x; xyz_own[3 * i + 1] = y; xyz_own[3 * i + 2] = z;
  const double* m = to_center + 12 * coords[i].x;
  // (x m0 + y m1) + z m2, then + t: the order of a 3-term dot product followed by the broadcast add, no contraction
  xyz_cf[3 * i] = (float)__dadd_rn(__dadd_rn(__dadd_rn(__dmul_rn(x, m[0]), __dmul_rn(y, m[1])), __dmul_rn(z, m[2])), m[3]);
  xyz_cf[3 * i + 1] = (float)__dadd_rn(__dadd_rn(__dadd_rn(__dmul_rn(x, m[4]), __dmul_rn(y, m[5])), __dmul_rn(z, m[6])), m[7]);
  xyz_cf[3 * i + 2] = (float)__dadd_rn(__dadd_rn(__dadd_rn(__dmul_rn(x, m[8]), __dmul_rn(y, m[9])), __dmul_rn(z, m[10])), m[11]);
}
}  // namespace gcl

using namespace gcl;

// device-wide scan lives in coords.hip


extern "C" {

int gcl_voxel_coords(const float* xyz, int64_t p, float voxel, int32_t batch_id, int32_t* coords, void* stream) {
  GCL_CHECK_ARG(xyz && coords && p > 0 && voxel > 0, "gcl_voxel_coords: bad argument");
  hipLaunchKernelGGL(k_voxel_coords, dim3((unsigned)cdiv(p, 256)), dim3(256), 0, (hipStream_t)stream, xyz,
                     (long long)p, voxel, batch_id, (int4*)coords);
  GCL_CHECK_LAUNCH();
  return GCL_OK;
}

int gcl_voxel_coords_multi(const float* xyz, int64_t p, const int64_t* cloud_offsets_host, int32_t n_clouds, float voxel,
                            int32_t* coords, void* stream) {
  GCL_CHECK_ARG(xyz && coords && cloud_offsets_host && p > 0 && voxel > 0, "gcl_voxel_coords_multi: bad argument");
  GCL_CHECK_ARG(n_clouds >= 1 && n_clouds <= 64 && cloud_offsets_host[0] == 0 && cloud_offsets_host[n_clouds] == p,
                "gcl_voxel_coords_multi: 1 <= clouds <= 64, offsets[0] = 0, offsets[clouds] = p");
  CloudOffsets co;
  for (int c = 0; c <= n_clouds; ++c) co.off[c] = cloud_offsets_host[c];
  hipLaunchKernelGGL(k_voxel_coords_multi, dim3((unsigned)cdiv(p, 256)), dim3(256), 0, (hipStream_t)stream, xyz, (long long)p,
                     co, n_clouds, voxel, (int4*)coords);
  GCL_CHECK_LAUNCH();
  return GCL_OK;
}

int gcl_cloud_row_starts(const int32_t* coords, int64_t n_max, const int32_t* n_dev, int32_t n_clouds, int32_t* starts,
                         void* stream) {
  GCL_CHECK_ARG(coords && n_dev && starts && n_max > 0 && n_clouds >= 1, "gcl_cloud_row_starts: bad argument");
  GCL_CHECK_HIP(hipMemsetAsync(starts, 0xFF, (size_t)(n_clouds + 1) * sizeof(int32_t), (hipStream_t)stream));      // -1: no row
  hipLaunchKernelGGL(k_cloud_row_starts, dim3((unsigned)cdiv(n_max, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const int4*)coords, (const int*)n_dev, n_clouds, starts);
  GCL_CHECK_LAUNCH();
  return GCL_OK;
}

int gcl_loader_points(const float* xyz_raw, const int64_t* index, const int32_t* coords, int64_t n_max, const int32_t* n_dev,
                      const double* to_center_dev, float* xyz_own, float* xyz_cf, void* stream) {
  GCL_CHECK_ARG(xyz_raw && index && coords && n_dev && to_center_dev && xyz_own && xyz_cf && n_max > 0,
                "gcl_loader_points: bad argument");
  hipLaunchKernelGGL(k_loader_points, dim3((unsigned)cdiv(n_max, 256)), dim3(256), 0, (hipStream_t)stream, xyz_raw,
                     (const long long*)index, (const int4*)coords, (const int*)n_dev, to_center_dev, xyz_own, xyz_cf);
  GCL_CHECK_LAUNCH();
  return GCL_OK;
}

int gcl_colocation_hits_at(const float* xyz_own, const float* xyz_cf, int64_t row0, int64_t n_center, int32_t cloud0,
                           int32_t n_clouds, const double* to_cloud_host, const int64_t* table, int64_t cap, float inv_voxel,
                           double radius, int32_t K, int32_t* hits, int32_t* cnt, double* first_rng, void* stream);

int gcl_colocation_hits(const float* xyz_own, const float* xyz_cf, int64_t n_center, int32_t n_clouds,
                        const double* to_cloud_host, const int64_t* table, int64_t cap, float inv_voxel, double radius,
                        int32_t K, int32_t* hits, int32_t* cnt, double* first_rng, void* stream) {
  return gcl_colocation_hits_at(xyz_own, xyz_cf, 0, n_center, 0, n_clouds, to_cloud_host, table, cap, inv_voxel, radius, K,
                                hits, cnt, first_rng, stream);
}

int gcl_colocation_hits_at(const float* xyz_own, const float* xyz_cf, int64_t row0, int64_t n_center, int32_t cloud0,
                           int32_t n_clouds, const double* to_cloud_host, const int64_t* table, int64_t cap, float inv_voxel,
                           double radius, int32_t K, int32_t* hits, int32_t* cnt, double* first_rng, void* stream) {
  GCL_CHECK_ARG(row0 >= 0 && cloud0 >= 0 && cloud0 + n_clouds <= 65535, "gcl_colocation_hits_at: bad row / cloud offset");
  GCL_CHECK_ARG(xyz_own && xyz_cf && to_cloud_host && table && hits && cnt && first_rng, "gcl_colocation_hits: null pointer");
  GCL_CHECK_ARG(n_center > 0 && n_clouds >= 1 && n_clouds <= 16 && K >= 1 && K <= KMAX,
                "gcl_colocation_hits: need 1 <= clouds <= 16 and 1 <= K <= %d", KMAX);
  GCL_CHECK_ARG(radius > 0 && inv_voxel > 0, "gcl_colocation_hits: radius and voxel size must be positive");
  Affines aff;
  for (int c = 0; c < n_clouds; ++c)
    for (int j = 0; j < 12; ++j) aff.a[c].m[j] = to_cloud_host[c * 12 + j];
  int R = (int)(radius * (double)inv_voxel) + 1;
  GCL_CHECK_ARG(R <= 4, "gcl_colocation_hits: radius / voxel too large (R = %d)", R);
  hipLaunchKernelGGL(k_colocation_hits, dim3((unsigned)cdiv(n_center, 256 / HITS_LANES), n_clouds), dim3(256), 0, (hipStream_t)stream,
                     xyz_own, xyz_cf, (long long)n_center, n_clouds, aff, (const Slot*)table, (long long)cap, inv_voxel,
                     radius, R, K, hits, cnt, first_rng, (long long)row0, cloud0);
  GCL_CHECK_LAUNCH();
  return GCL_OK;
}

/* scratch: int32[4 * n_center + gcl_scan_scratch_len(n_center)]; totals: device int32[2] = {#groups, #index entries};
 * group / index / finest need capacity n_center / n_center * n_clouds * K. */
int gcl_colocation_emit(const int32_t* hits, const int32_t* cnt, const double* first_rng, int64_t n_center,
                        int32_t n_clouds, int32_t K, int32_t* scratch, int32_t* group, int64_t* index, uint8_t* finest,
                        int32_t* totals, void* stream) {
  GCL_CHECK_ARG(hits && cnt && first_rng && scratch && group && index && finest && totals, "gcl_colocation_emit: null pointer");
  GCL_CHECK_ARG(n_center > 0 && n_clouds >= 1 && K >= 1 && K <= KMAX, "gcl_colocation_emit: bad sizes");
  hipStream_t st = (hipStream_t)stream;
  int* gsize = scratch;
  int* kept = scratch + n_center;
  int* goff = scratch + 2 * n_center;
  int* gidx = scratch + 3 * n_center;
  int* sc = scratch + 4 * n_center;
  unsigned g = (unsigned)cdiv(n_center, 256);
  hipLaunchKernelGGL(k_colocation_sizes, dim3(g), dim3(256), 0, st, cnt, (long long)n_center, n_clouds, gsize, kept);
  GCL_CHECK_LAUNCH();
  int rc = gcl_exclusive_scan_i32(gsize, n_center, goff, sc, stream);
  if (rc) return rc;
  rc = gcl_exclusive_scan_i32(kept, n_center, gidx, sc, stream);
  if (rc) return rc;
  hipLaunchKernelGGL(k_colocation_emit, dim3(g), dim3(256), 0, st, hits, cnt, first_rng, (long long)n_center, n_clouds, K,
                     (const int*)gsize, (const int*)goff, (const int*)gidx, group, (long long*)index, finest, totals);
  GCL_CHECK_LAUNCH();
  return GCL_OK;
}

/* ---------------------------------------------------------------------------------------------------------------
 * HOST function (no GPU): numpy's legacy ``np.random.choice(n, k, replace=False)`` -- i.e. ``permutation(n)[:k]``, the
 * call the reference makes three times per training step (lib/colocation_trainer.py:457, :506-507) -- reproduced bit for
 * bit from the RandomState's MT19937 state, outside the Python interpreter lock.  At n = 0.5 M rows numpy needs 8 ms per
 * call and holds the interpreter lock while it shuffles, which stalls the thread that enqueues the GPU work.
 *   key[624], *pos: the generator state as returned by np.random.get_state() (updated in place: the caller hands it back
 *   with np.random.set_state, so the random stream continues exactly where numpy's own call would have left it);
 *   out[k] = the first k entries of the shuffled arange(n); work: int64[n + n / 32 + 64] scratch.
 * Algorithm = numpy/random/mtrand.pyx (_shuffle_raw: for i = n-1 .. 1: j = random_interval(i); swap) with
 * distributions.c random_interval (smallest mask >= max, rejection on next_uint32 / next_uint64). */
namespace {
struct MT {
  uint32_t* key;
  int pos;
  uint32_t out[624];      // tempered outputs of the current state block (filled in bulk: the loop vectorises)
  bool fresh = false;
};
inline void mt_temper_block(MT& s) {
  for (int i = 0; i < 624; ++i) {
    uint32_t y = s.key[i];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    s.out[i] = y;
  }
  s.fresh = true;
}
inline void mt_gen(uint32_t* mt) {
  const int N = 624, M = 397;
  const uint32_t UPPER = 0x80000000u, LOWER = 0x7fffffffu, MAT = 0x9908b0dfu;
  int i;
  uint32_t y;
  for (i = 0; i < N - M; i++) {
    y = (mt[i] & UPPER) | (mt[i + 1] & LOWER);
    mt[i] = mt[i + M] ^ (y >> 1) ^ (-(int32_t)(y & 1) & MAT);
  }
  for (; i < N - 1; i++) {
    y = (mt[i] & UPPER) | (mt[i + 1] & LOWER);
    mt[i] = mt[i + (M - N)] ^ (y >> 1) ^ (-(int32_t)(y & 1) & MAT);
  }
  y = (mt[N - 1] & UPPER) | (mt[0] & LOWER);
  mt[N - 1] = mt[M - 1] ^ (y >> 1) ^ (-(int32_t)(y & 1) & MAT);
}
inline uint32_t mt_next32(MT& s) {
  if (s.pos == 624) {
    mt_gen(s.key);
    s.pos = 0;
    s.fresh = false;
  }
  if (!s.fresh) mt_temper_block(s);
  return s.out[s.pos++];
}
inline uint64_t mt_interval(MT& s, uint64_t max) {
  if (max == 0) return 0;
  uint64_t mask = max, value;
  mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16; mask |= mask >> 32;
  if (max <= 0xffffffffull) {
    while ((value = (mt_next32(s) & mask)) > max) {}
  } else {
    while (true) {
      uint64_t hi = mt_next32(s), lo = mt_next32(s);
      value = ((hi << 32) | lo) & mask;
      if (value <= max) break;
    }
  }
  return value;
}
}  // namespace

int gcl_host_legacy_choice(uint32_t* key, int32_t* pos, int64_t n, int64_t k, int64_t* work, int64_t* out) {
  GCL_CHECK_ARG(key && pos && work && out, "gcl_host_legacy_choice: null pointer");
  GCL_CHECK_ARG(n >= 1 && k >= 0 && k <= n && *pos >= 0 && *pos <= 624, "gcl_host_legacy_choice: bad n / k / state");
  MT s;
  s.key = key;
  s.pos = *pos;
  if (k * 8 > n || n < 4096) {            // dense selection: shuffle the whole array as numpy does
    for (int64_t i = 0; i < n; ++i) work[i] = i;
    for (int64_t i = n - 1; i >= 1; --i) {
      const int64_t j = (int64_t)mt_interval(s, (uint64_t)i);
      const int64_t t = work[j];
      work[j] = work[i];
      work[i] = t;
    }
    for (int64_t i = 0; i < k; ++i) out[i] = work[i];
    *pos = s.pos;
    return GCL_OK;
  }
  // Sparse selection (k << n): only the first k entries of the shuffled arange(n) are wanted.  The swap partners
  // j_i (i = n-1 .. 1) are drawn in numpy's order -- the random stream is consumed exactly as by the full shuffle -- and
  // kept as 32-bit numbers; then the swaps are UNDONE from the last one (i = 1) back to the first (i = n-1) for k
  // tokens that start at the final positions 0 .. k-1: a token sitting on i moves to j_i and vice versa, and after the
  // last undo its position is its value in arange(n).  A presence bitmap (n / 8 bytes, cache resident) answers "is a
  // token on i or on j_i?" -- no for all but a few thousand of the n steps -- instead of 16 random bytes per step.
  uint32_t* js = reinterpret_cast<uint32_t*>(work);                      // n entries
  if (n > 0xffffffffll) {
    for (int64_t i = n - 1; i >= 1; --i) js[i] = (uint32_t)mt_interval(s, (uint64_t)i);
  } else {
    // random_interval(i) for i = n-1 .. 1 without an unpredictable branch per draw: all i that share a mask are served
    // from the tempered block in one loop -- every output is masked and stored to js[i], and i steps down only when
    // the value was accepted (<= i); the outputs are consumed one by one exactly as numpy's rejection loop does
    int64_t i = n - 1;
    while (i >= 1) {
      uint64_t mask = (uint64_t)i;
      mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
      const int64_t lo = (int64_t)(mask >> 1) + 1;          // the smallest i with this mask
      const uint32_t m32 = (uint32_t)mask;
      while (i >= lo) {
        if (s.pos == 624) {
          mt_gen(s.key);
          s.pos = 0;
          s.fresh = false;
        }
        if (!s.fresh) mt_temper_block(s);
        const uint32_t* o = s.out + s.pos;
        const int avail = 624 - s.pos;
        int c = 0;
        for (; c < avail && i >= lo; ++c) {
          const uint32_t v = o[c] & m32;
          js[i] = v;
          i -= (v <= (uint32_t)i) ? 1 : 0;
        }
        s.pos += c;
      }
    }
  }
  *pos = s.pos;
  int32_t* tok = reinterpret_cast<int32_t*>(work + (n + 1) / 2 + 1);     // token on a position (valid where its bit is set)
  uint64_t* bits = reinterpret_cast<uint64_t*>(work + 2 * ((n + 1) / 2 + 1));
  const int64_t nw = (n + 63) / 64;
  for (int64_t w = 0; w < nw; ++w) bits[w] = 0;
  int64_t* where = out;                                                  // where[t] = current position of token t
  for (int64_t t = 0; t < k; ++t) {
    where[t] = t;
    tok[t] = (int32_t)t;
    bits[t >> 6] |= 1ull << (t & 63);
  }
  for (int64_t i = 1; i < n; ++i) {
    const int64_t j = js[i];
    const bool hi = (bits[i >> 6] >> (i & 63)) & 1ull, hj = (bits[j >> 6] >> (j & 63)) & 1ull;
    if (!(hi | hj) || i == j) continue;
    const int32_t ti = hi ? tok[i] : -1, tj = hj ? tok[j] : -1;
    if (ti >= 0) {
      where[ti] = j;
      tok[j] = ti;
    }
    if (tj >= 0) {
      where[tj] = i;
      tok[i] = tj;
    }
    if (hi != hj) {                       // exactly one token moved: update the bitmap
      bits[i >> 6] ^= 1ull << (i & 63);
      bits[j >> 6] ^= 1ull << (j & 63);
    }
  }
  return GCL_OK;                          // out[t] = where[t] = permutation(n)[t]
}

}  // extern "C"
