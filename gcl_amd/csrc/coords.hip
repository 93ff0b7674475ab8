// Coordinate maps, strided maps and kernel maps (integer work, bit-exact, deterministic).
//
// Replaces MinkowskiEngine's coordinate manager (hash-map insert -> stride map -> kernel map) for the calls
// the reference makes through ME.SparseTensor / ME.MinkowskiConvolution.
// Design for gfx950: one 16-byte slot per key (single aligned access per probe, the table of a 0.5 M voxel
// batch is ~16 MB and lives in L2 / Infinity Cache); output row order is by FIRST OCCURRENCE (atomicMin of the
// row id + flag + prefix sum), so it does not depend on insertion timing; neighbour tables are k-major
// [K][N_out] so that every later gather reads its indices fully coalesced; per-offset pair lists are compacted
// with wave ballots + block prefix sums (no atomics on the data path => deterministic).
#include "common.h"

#include <limits.h>
#include <stdlib.h>

#include <algorithm>

namespace gcl {

thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// ---------------------------------------------------------------------------------------------------
// (also zeroes the four status words of the insertion that follows: one launch instead of a memset + a launch -- a pass
// over one pair of clouds is a chain of ~ 130 small dependent launches, every one of them ~ 4.5 us of somebody's time)
__global__ void k_table_fill(Slot* t, long long cap, int* zero4) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < cap) {
    t[i].key = EMPTY_KEY;
    t[i].val = LLONG_MAX;
  }
  if (zero4 && i < 4) zero4[i] = 0;
}

// p[0 .. n) = v: the fills of the map builders as ONE launch each (hipMemsetAsync splits a length that is not a multiple of its
// vector width into two dispatches and costs the enqueuing thread about twice a kernel launch)
__global__ void __launch_bounds__(256) k_fill32(unsigned* __restrict__ p, long long n, unsigned v) {
  const long long i0 = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i0 + 4 <= n && (reinterpret_cast<uintptr_t>(p + i0) & 15) == 0) {
    *reinterpret_cast<uint4*>(p + i0) = make_uint4(v, v, v, v);
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (i0 + j < n) p[i0 + j] = v;
  }
}
void fill32(void* p, long long n_words, unsigned v, hipStream_t st) {      // (also called by plan.hip's map build)
  if (n_words > 0)
    hipLaunchKernelGGL(k_fill32, dim3((unsigned)cdiv(n_words, 1024)), dim3(256), 0, st, (unsigned*)p, n_words, v);
}

__global__ void k_coords_insert(const int4* __restrict__ coords, long long n, Slot* t, long long cap,
                                int* status) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int4 c = coords[i];
  if (!pack_ok(c.x, c.y, c.z, c.w)) {
    atomicAdd(&status[0], 1);
    return;
  }
  unsigned long long key = pack_key(c.x, c.y, c.z, c.w);
  long long s = (long long)(mix64(key) & (unsigned long long)(cap - 1));
  while (true) {
    unsigned long long prev = atomicCAS(&t[s].key, EMPTY_KEY, key);
    if (prev == EMPTY_KEY) break;
    if (prev == key) {
      atomicAdd(&status[1], 1);
      break;
    }
    s = (s + 1) & (cap - 1);
  }
  atomicMin(&t[s].val, i);
}

__device__ __forceinline__ int floor_div(int a, int b) {
  int q = a / b;
  return (a % b != 0 && ((a < 0) != (b < 0))) ? q - 1 : q;
}

__global__ void k_stride_insert(const int4* __restrict__ coords, long long n, const int* __restrict__ n_dev,
                                int t_out, Slot* t, long long cap, int* status) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (n_dev) n = *n_dev;
  if (i >= n) return;
  int4 c = coords[i];
  c.y = floor_div(c.y, t_out) * t_out;
  c.z = floor_div(c.z, t_out) * t_out;
  c.w = floor_div(c.w, t_out) * t_out;
  if (!pack_ok(c.x, c.y, c.z, c.w)) {
    atomicAdd(&status[0], 1);
    return;
  }
  long long s = table_insert(t, cap, pack_key(c.x, c.y, c.z, c.w));
  atomicMin(&t[s].val, i);
}

__global__ void k_stride_flag(const int4* __restrict__ coords, long long n_max, const int* __restrict__ n_dev,
                              int t_out, const Slot* t, long long cap, int* flag) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_max) return;
  if (n_dev && i >= *n_dev) {
    flag[i] = 0;
    return;
  }
  int4 c = coords[i];
  c.y = floor_div(c.y, t_out) * t_out;
  c.z = floor_div(c.z, t_out) * t_out;
  c.w = floor_div(c.w, t_out) * t_out;
  int f = 0;
  if (pack_ok(c.x, c.y, c.z, c.w)) {
    long long s = table_find(t, cap, pack_key(c.x, c.y, c.z, c.w));
    f = (s >= 0 && t[s].val == i) ? 1 : 0;
  }
  flag[i] = f;
}

__global__ void k_stride_emit(const int4* __restrict__ coords, long long n_max, int t_out, Slot* t, long long cap,
                              const int* __restrict__ flag, const int* __restrict__ pos, int4* coords_out,
                              int* n_out, long long* index_out) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_max) return;
  if (i == n_max - 1) *n_out = pos[i] + flag[i];
  if (index_out && flag[i]) index_out[pos[i]] = i;
  if (!flag[i]) return;
  int4 c = coords[i];
  c.y = floor_div(c.y, t_out) * t_out;
  c.z = floor_div(c.z, t_out) * t_out;
  c.w = floor_div(c.w, t_out) * t_out;
  long long s = table_find(t, cap, pack_key(c.x, c.y, c.z, c.w));
  t[s].val = pos[i];
  coords_out[pos[i]] = c;
}

// ---- device-wide exclusive scan of int32 (3 launches) -----------------------------------------------
constexpr int SCAN_T = 256;
constexpr int SCAN_I = 8;
constexpr int SCAN_B = SCAN_T * SCAN_I;

// exclusive scan of one value per thread across the workgroup; returns the exclusive prefix, total in *tot
__device__ __forceinline__ int block_excl_scan(int v, int* lds /*[5]*/, int* tot) {
  int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    int y = __shfl_up(inc, o);
    if (lane >= o) inc += y;
  }
  if (lane == 63) lds[w] = inc;
  __syncthreads();
  int base = 0, total = 0;
  for (int j = 0; j < (int)(blockDim.x >> 6); ++j) {
    int s = lds[j];
    if (j < w) base += s;
    total += s;
  }
  __syncthreads();
  *tot = total;
  return base + inc - v;
}

__global__ void __launch_bounds__(SCAN_T) k_scan_reduce(const int* __restrict__ in, long long n, int* bs) {
  __shared__ int lds[8];
  long long base = (long long)blockIdx.x * SCAN_B + threadIdx.x * SCAN_I;
  int s = 0;
#pragma unroll
  for (int j = 0; j < SCAN_I; ++j)
    if (base + j < n) s += in[base + j];
  int tot;
  block_excl_scan(s, lds, &tot);
  if (threadIdx.x == 0) bs[blockIdx.x] = tot;
}

__global__ void __launch_bounds__(SCAN_T) k_scan_blocksums(int* bs, long long nb) {
  __shared__ int lds[8];
  int carry = 0;
  for (long long c = 0; c < nb; c += SCAN_T) {
    long long i = c + threadIdx.x;
    int v = (i < nb) ? bs[i] : 0;
    int tot;
    int ex = block_excl_scan(v, lds, &tot);
    if (i < nb) bs[i] = carry + ex;
    carry += tot;
  }
  if (threadIdx.x == 0) bs[nb] = carry;
}

// raw != 0: bs holds the blocks' RAW totals (k_scan_reduce's output, at most SCAN_T of them) and every block adds up the
// totals in front of it itself -- two launches instead of three for arrays of up to SCAN_T * SCAN_B = 524 288 entries
__global__ void __launch_bounds__(SCAN_T) k_scan_final(const int* __restrict__ in, long long n,
                                                       const int* __restrict__ bs, int* out, int raw) {
  __shared__ int lds[8];
  int before = 0;
  if (raw) {
    int mine = ((int)threadIdx.x < (int)blockIdx.x) ? bs[threadIdx.x] : 0;
    block_excl_scan(mine, lds, &before);
  }
  long long base = (long long)blockIdx.x * SCAN_B + threadIdx.x * SCAN_I;
  int v[SCAN_I];
  int s = 0;
#pragma unroll
  for (int j = 0; j < SCAN_I; ++j) {
    v[j] = (base + j < n) ? in[base + j] : 0;
    s += v[j];
  }
  int tot;
  int ex = block_excl_scan(s, lds, &tot) + (raw ? before : bs[blockIdx.x]);
#pragma unroll
  for (int j = 0; j < SCAN_I; ++j) {
    if (base + j < n) out[base + j] = ex;
    ex += v[j];
  }
}

// single-workgroup exclusive scan for short arrays (one launch instead of three): tiles of 4096 elements, coalesced
// int4 loads, wave-shuffle scan + LDS carry between tiles
constexpr int SCAN1_T = 1024;
__global__ void __launch_bounds__(SCAN1_T) k_scan_single(const int* __restrict__ in, long long n, int* out) {
  __shared__ int wsum[SCAN1_T / 64];
  __shared__ int tile_total;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  int carry = 0;
  for (long long base = 0; base < n; base += SCAN1_T * 4) {
    long long i0 = base + (long long)t * 4;
    int v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = (i0 + j < n) ? in[i0 + j] : 0;
    int s = v[0] + v[1] + v[2] + v[3];
    int inc = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      int y = __shfl_up(inc, o);
      if (lane >= o) inc += y;
    }
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    if (t < SCAN1_T / 64) {          // 16 wave totals: exclusive scan by the first 16 lanes of wave 0
      int x = wsum[t], incw = x;
#pragma unroll
      for (int o = 1; o < SCAN1_T / 64; o <<= 1) {
        int y = __shfl_up(incw, o);
        if (t >= o) incw += y;
      }
      wsum[t] = incw - x;
      if (t == SCAN1_T / 64 - 1) tile_total = incw;
    }
    __syncthreads();
    int ex = carry + wsum[w] + inc - s;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (i0 + j < n) out[i0 + j] = ex;
      ex += v[j];
    }
    carry += tile_total;
    __syncthreads();
  }
}

static int device_scan(const int* in, long long n, int* out, int* bs, hipStream_t st) {
  // one workgroup walks 4096-entry tiles at ~ 2.6 us each: one launch up to two tiles; up to SCAN_T blocks the final pass
  // adds up the raw block totals itself (two launches); beyond that the block totals get their own scan (three).
  // (A pass over one pair of clouds scans the INPUT's row bound, 36 k entries, once per strided level: 23.4 us as one
  // workgroup, 3 x 4.8 us of dispatch latency as three launches, two launches now.)
  if (n <= 8192) {
    hipLaunchKernelGGL(k_scan_single, dim3(1), dim3(SCAN1_T), 0, st, in, n, out);
    GCL_CHECK_LAUNCH();
    return GCL_OK;
  }
  long long nb = cdiv(n, SCAN_B);
  hipLaunchKernelGGL(k_scan_reduce, dim3((unsigned)nb), dim3(SCAN_T), 0, st, in, n, bs);
  if (nb <= SCAN_T) {
    hipLaunchKernelGGL(k_scan_final, dim3((unsigned)nb), dim3(SCAN_T), 0, st, in, n, (const int*)bs, out, 1);
    GCL_CHECK_LAUNCH();
    return GCL_OK;
  }
  hipLaunchKernelGGL(k_scan_blocksums, dim3(1), dim3(SCAN_T), 0, st, bs, nb);
  hipLaunchKernelGGL(k_scan_final, dim3((unsigned)nb), dim3(SCAN_T), 0, st, in, n, (const int*)bs, out, 0);
  GCL_CHECK_LAUNCH();
  return GCL_OK;
}

// ---- kernel map ------------------------------------------------------------------------------------
// Two devices keep the hash table out of the way (it is 16 MB at 0.5 M voxels and most probes miss it in L2):
//  * a 2 MB presence bitmap (one bit per hashed key, L2-resident) answers most ABSENT neighbours without touching
//    the table -- 83 % of the 5^3 stem's lookups are misses;
//  * for a map onto itself (stride 1) offset k and its mirror K-1-k describe the same pairs, so only the first half
//    of the offsets is looked up and the mirror entry is written from the hit (half the lookups).
constexpr int BITMAP_BITS_LOG2 = 24;
constexpr long long BITMAP_WORDS = 1ll << (BITMAP_BITS_LOG2 - 5);

__device__ __forceinline__ unsigned bitmap_pos(unsigned long long key) {
  return (unsigned)(mix64(key ^ 0x9e3779b97f4a7c15ull) >> (64 - BITMAP_BITS_LOG2));
}

__global__ void k_bitmap_fill(const Slot* __restrict__ t, long long cap, unsigned* bitmap) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= cap) return;
  unsigned long long key = t[i].key;
  if (key == EMPTY_KEY) return;
  unsigned b = bitmap_pos(key);
  atomicOr(&bitmap[b >> 5], 1u << (b & 31));
}

// A thread answers KPT offsets of its voxel (round 5; it was one: 63 x 2070 workgroups of one dependent chain -- coordinate,
// presence word, table slot, row -- each at the 5^3 stem map of 0.5 M voxels, wait_any 0.81): the coordinate is read once, the KPT
// presence words are independent loads in flight together, and only then the (rare: 17 %) table probes follow.  Same entries,
// same per-(offset, block) counts.
// per-offset number of threads of the 256-thread block with a hit, written by threads 0 .. KPT - 1: plain stores into the
// per-block array (k_count_reduce adds them up), or -- `acc` -- integer atomics straight into the zeroed counts (mir: the mirror
// offset of a same-map): the same totals without the reduction launch, for builds of few blocks
template <int KPT>
__device__ __forceinline__ void block_counts(const bool (&pred)[KPT], int* const (&dst)[KPT], int* const (&mir)[KPT], bool acc) {
  __shared__ int wc[KPT][4];
#pragma unroll
  for (int q = 0; q < KPT; ++q) {
    const unsigned long long m = __ballot(pred[q]);
    if ((threadIdx.x & 63) == 0) wc[q][threadIdx.x >> 6] = __popcll(m);
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < KPT; ++q)
    if ((int)threadIdx.x == q && dst[q]) {
      const int c = wc[q][0] + wc[q][1] + wc[q][2] + wc[q][3];
      if (!acc) {
        *dst[q] = c;
      } else if (c) {
        atomicAdd(dst[q], c);
        if (mir[q]) atomicAdd(mir[q], c);
      }
    }
}

// rows of the KPT neighbours c + o_k of one voxel (-1: absent / out of the packed range / k >= nk)
template <int KPT>
__device__ __forceinline__ void lookup_rows(const Slot* __restrict__ t, long long cap, const unsigned* __restrict__ bitmap,
                                            const int4& c, int ks, int step, int k0, int nk, int (&u)[KPT]) {
  const int r = ks / 2;
  unsigned long long key[KPT];
  bool maybe[KPT];
#pragma unroll
  for (int q = 0; q < KPT; ++q) {
    const int k = k0 + q;
    const int x = c.y + (k % ks - r) * step, y = c.z + ((k / ks) % ks - r) * step, z = c.w + (k / (ks * ks) - r) * step;
    maybe[q] = k < nk && pack_ok(c.x, x, y, z);
    key[q] = pack_key(c.x, x, y, z);
    u[q] = -1;
  }
  if (bitmap) {
    unsigned word[KPT];
#pragma unroll
    for (int q = 0; q < KPT; ++q) word[q] = maybe[q] ? bitmap[bitmap_pos(key[q]) >> 5] : 0u;
#pragma unroll
    for (int q = 0; q < KPT; ++q) maybe[q] = maybe[q] && ((word[q] >> (bitmap_pos(key[q]) & 31)) & 1u);
  }
#pragma unroll
  for (int q = 0; q < KPT; ++q) {
    if (maybe[q]) {
      const long long s = table_find(t, cap, key[q]);
      if (s >= 0) u[q] = (int)t[s].val;
    }
  }
}

template <int KPT>
__global__ void __launch_bounds__(256) k_kernel_map(const int4* __restrict__ coords_out, long long n_out,
                                                    const Slot* __restrict__ t, long long cap,
                                                    const unsigned* __restrict__ bitmap, int ks, int step,
                                                    int* __restrict__ nbr, int* __restrict__ nbr_t,
                                                    long long n_in, int* blockcnt, int acc) {
  const long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int K = ks * ks * ks, k0 = blockIdx.y * KPT;
  int u[KPT];
  bool hit[KPT];
  int* dst[KPT];
  int* mir[KPT];
#pragma unroll
  for (int q = 0; q < KPT; ++q) {
    u[q] = -1;
    mir[q] = nullptr;
    dst[q] = (k0 + q < K) ? (acc ? blockcnt + (k0 + q) : blockcnt + (long long)(k0 + q) * gridDim.x + blockIdx.x) : nullptr;
  }
  if (v < n_out) {
    lookup_rows<KPT>(t, cap, bitmap, coords_out[v], ks, step, k0, K, u);
#pragma unroll
    for (int q = 0; q < KPT; ++q) {
      const int k = k0 + q;
      if (k < K) {
        nbr[(long long)k * n_out + v] = u[q];
        if (nbr_t != nullptr && u[q] >= 0) nbr_t[(long long)k * n_in + u[q]] = (int)v;
      }
    }
  }
#pragma unroll
  for (int q = 0; q < KPT; ++q) hit[q] = u[q] >= 0;
  block_counts<KPT>(hit, dst, mir, acc != 0);
}

// same-map variant: offsets k in [0, K/2]; the mirror half [K/2+1, K) was pre-filled with -1
template <int KPT>
__global__ void __launch_bounds__(256) k_kernel_map_sym(const int4* __restrict__ coords, long long n,
                                                        const Slot* __restrict__ t, long long cap,
                                                        const unsigned* __restrict__ bitmap, int ks, int step,
                                                        int* __restrict__ nbr, int* blockcnt, int acc) {
  const long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int K = ks * ks * ks, k0 = blockIdx.y * KPT;
  int u[KPT];
  bool hit[KPT];
  int* dst[KPT];
  int* mir[KPT];
#pragma unroll
  for (int q = 0; q < KPT; ++q) {
    u[q] = -1;
    const int k = k0 + q;
    dst[q] = (k <= K / 2) ? (acc ? blockcnt + k : blockcnt + (long long)k * gridDim.x + blockIdx.x) : nullptr;
    mir[q] = (acc && k < K / 2) ? blockcnt + (K - 1 - k) : nullptr;
  }
  if (v < n) {
    lookup_rows<KPT>(t, cap, bitmap, coords[v], ks, step, k0, K / 2, u);      // the centre (k = K / 2) is the voxel itself
#pragma unroll
    for (int q = 0; q < KPT; ++q) {
      const int k = k0 + q;
      if (k == K / 2) u[q] = (int)v;
      if (k <= K / 2) nbr[(long long)k * n + v] = u[q];
      if (k < K / 2 && u[q] >= 0) nbr[(long long)(K - 1 - k) * n + u[q]] = (int)v;   // c_u + o_{K-1-k} = c_v
    }
  }
#pragma unroll
  for (int q = 0; q < KPT; ++q) hit[q] = u[q] >= 0;
  block_counts<KPT>(hit, dst, mir, acc != 0);
}

// counts[k] = sum of the per-block counts (ordered, no atomics); mirror offsets share the count in a same-map
__global__ void __launch_bounds__(256) k_count_reduce(const int* __restrict__ blockcnt, int nblk, int K, int sym,
                                                      int* counts) {
  __shared__ int red[256];
  const int k = blockIdx.x;
  int s = 0;
  for (int b = threadIdx.x; b < nblk; b += 256) s += blockcnt[(long long)k * nblk + b];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    counts[k] = red[0];
    if (sym && k != K / 2) counts[K - 1 - k] = red[0];
  }
}

struct SegOff {
  long long off[130];
};

constexpr int PAIR_B = 1024;  // out rows per compaction block (256 threads x 4)

__global__ void __launch_bounds__(256) k_pairs_count(const int* __restrict__ nbr, long long n_out, int nb,
                                                     int* bc) {
  __shared__ int lds[8];
  int k = blockIdx.y;
  long long base = (long long)blockIdx.x * PAIR_B + threadIdx.x * 4;
  int s = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (base + j < n_out) s += (nbr[(long long)k * n_out + base + j] >= 0);
  int tot;
  block_excl_scan(s, lds, &tot);
  if (threadIdx.x == 0) bc[(long long)k * nb + blockIdx.x] = tot;
}

__global__ void __launch_bounds__(256) k_pairs_scan(int* bc, int nb) {
  __shared__ int lds[8];
  int* row = bc + (long long)blockIdx.x * nb;
  int carry = 0;
  for (int c = 0; c < nb; c += 256) {
    int i = c + threadIdx.x;
    int v = (i < nb) ? row[i] : 0;
    int tot;
    int ex = block_excl_scan(v, lds, &tot);
    if (i < nb) row[i] = carry + ex;
    carry += tot;
  }
}

__global__ void __launch_bounds__(256) k_pairs_emit(const int* __restrict__ nbr, long long n_out, int nb,
                                                    const int* __restrict__ bc, SegOff seg, int* pair_in,
                                                    int* pair_out) {
  __shared__ int lds[8];
  int k = blockIdx.y;
  long long base = (long long)blockIdx.x * PAIR_B + threadIdx.x * 4;
  int u[4];
  int s = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    u[j] = (base + j < n_out) ? nbr[(long long)k * n_out + base + j] : -1;
    s += (u[j] >= 0);
  }
  int tot;
  const long long blk0 = seg.off[k] + bc[(long long)k * nb + blockIdx.x];
  long long p = blk0 + block_excl_scan(s, lds, &tot);
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (u[j] >= 0) {
      pair_in[p] = u[j];
      pair_out[p] = (int)(base + j);
      ++p;
    }
  // the segment's padding (< GCL_PAIR_CHUNK entries behind its last pair) is written here, by the offset's last block --
  // the lists used to be filled with -1 as a whole first (two passes over every list, ~15 fills per batch)
  if ((int)blockIdx.x == nb - 1)
    for (long long q = blk0 + tot + threadIdx.x; q < seg.off[k + 1]; q += blockDim.x) pair_in[q] = pair_out[q] = -1;
}


// ---- row ordering by neighbour-presence mask (cuts the MFMA work of the output-stationary conv) ---------
// Rows whose 27-bit presence masks are equal or close end up in the same 32-row wave tile, so the tile's OR-mask
// (the offsets the wave must visit) is close to each row's own mask.  On LiDAR sheets this halves the MFMA work
// relative to the loader's row order (measured on the synthetic KITTI batch: 2.8x -> 1.4x the exact sparse work).
// Stable LSD radix sort, 8-bit digits, one wave per 2048-element block (wave-local ranking by ballots).
#ifndef GCL_RS_BLOCK
#define GCL_RS_BLOCK 512
#endif
constexpr int RS_BLOCK = GCL_RS_BLOCK;   // one wave per block: small blocks => >= 4 waves per CU at 0.5 M rows

__device__ __forceinline__ void row_masks_body(const int* __restrict__ tbl, int K, long long n, unsigned* keys, int* vals,
                                               unsigned bx) {
  long long v = (long long)bx * blockDim.x + threadIdx.x;
  if (v >= n) return;
  unsigned m = 0;
  for (int k = 0; k < K; ++k) m |= (tbl[(long long)k * n + v] >= 0 ? 1u : 0u) << k;
  keys[v] = m;
  vals[v] = (int)v;
}
__global__ void k_row_masks(const int* __restrict__ tbl, int K, long long n, unsigned* keys, int* vals) {
  row_masks_body(tbl, K, n, keys, vals, blockIdx.x);
}

// bit_count[k] (32 zero-initialised ints) += number of rows whose mask has offset k.  A small fixed grid walks the
// masks, every thread counts in registers, the workgroup adds up in LDS and issues ONE integer atomic per offset
// (512 x 27 atomics per table: same-address atomics serialise at ~10 ns each, one per wave cost 2 ms per table)
__device__ __forceinline__ void mask_bit_count_body(const unsigned* __restrict__ keys, long long n, int K, int* bit_count,
                                                    unsigned bx, unsigned nbx) {
  __shared__ int tot[32];
  if (threadIdx.x < 32) tot[threadIdx.x] = 0;
  __syncthreads();
  int c[27];
#pragma unroll
  for (int k = 0; k < 27; ++k) c[k] = 0;
  for (long long v = (long long)bx * blockDim.x + threadIdx.x; v < n; v += (long long)nbx * blockDim.x) {
    const unsigned m = keys[v];
#pragma unroll
    for (int k = 0; k < 27; ++k) c[k] += (int)((m >> k) & 1u);
  }
#pragma unroll
  for (int k = 0; k < 27; ++k) {
    int x = c[k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
    if ((threadIdx.x & 63) == 0 && x) atomicAdd(&tot[k], x);
  }
  __syncthreads();
  if ((int)threadIdx.x < K && tot[threadIdx.x]) atomicAdd(&bit_count[threadIdx.x], tot[threadIdx.x]);
}
__global__ void __launch_bounds__(256) k_mask_bit_count(const unsigned* __restrict__ keys, long long n, int K, int* bit_count) {
  mask_bit_count_body(keys, n, K, bit_count, blockIdx.x, gridDim.x);
}

// Sort key = the mask with its bits re-ordered by how often each offset occurs in THIS table: the rarest offset
// becomes the most significant bit, the most frequent one the least significant (ties: lower offset index lower).
// Rows that own a rare offset then sit together, so the unit that offset costs is shared by a whole tile, while the
// trailing (frequent) bits are set in most rows anyway.  Measured on the KITTI batch against the offset-index order:
// 7.75 vs 8.26 units per 32-row tile at tensor stride 1, 9.88 vs 10.99 at stride 4, 11.41 vs 13.11 at stride 8
// (exact sparse work: 6.3 / 7.3 / 7.7).  pos[k] = key bit of offset k, written for k_tile_masks.
__device__ __forceinline__ void mask_keys_body(unsigned* keys, long long n, int K, const int* __restrict__ bit_count,
                                               int* pos_out, unsigned bx) {
  __shared__ int pos[32];
  if (threadIdx.x < 32) {
    const int k = threadIdx.x;
    int p = 0;
    if (k < K) {
      const int ck = bit_count[k];
      for (int j = 0; j < K; ++j) {
        const int cj = bit_count[j];
        p += (cj > ck || (cj == ck && j < k)) ? 1 : 0;
      }
    }
    pos[k] = p;
    if (bx == 0 && k < K) pos_out[k] = p;
  }
  __syncthreads();
  long long v = (long long)bx * blockDim.x + threadIdx.x;
  if (v >= n) return;
  const unsigned m = keys[v];
  unsigned key = 0;
  for (int k = 0; k < K; ++k) key |= ((m >> k) & 1u) << pos[k];
  // (taking the key's rank in the reflected Gray-code sequence would save another 2 - 3 % of the units -- measured --
  // but its pseudo-random low digits turn every radix scatter into 64 single-row writes per wave: +1 .. 2 ms per step)
  keys[v] = key;
}
__global__ void __launch_bounds__(256) k_mask_keys(unsigned* keys, long long n, int K, const int* __restrict__ bit_count,
                                                   int* pos_out) {
  mask_keys_body(keys, n, K, bit_count, pos_out, blockIdx.x);
}

__device__ __forceinline__ void radix_hist_body(const unsigned* __restrict__ keys, long long n, int shift, int nblk, int* hist,
                                                unsigned bx) {
  __shared__ int cnt[256];
  const int lane = threadIdx.x;
  for (int d = lane; d < 256; d += 64) cnt[d] = 0;
  __syncthreads();
  const long long base = (long long)bx * RS_BLOCK;
  for (int c = 0; c < RS_BLOCK / 64; ++c) {
    long long i = base + c * 64 + lane;
    if (i < n) atomicAdd(&cnt[(keys[i] >> shift) & 255u], 1);
  }
  __syncthreads();
  for (int d = lane; d < 256; d += 64) hist[(long long)d * nblk + bx] = cnt[d];
}
__global__ void __launch_bounds__(64) k_radix_hist(const unsigned* __restrict__ keys, long long n, int shift,
                                                   int nblk, int* hist) {
  radix_hist_body(keys, n, shift, nblk, hist, blockIdx.x);
}

// one workgroup per digit d: within[d][b] = sum_{b' < b} hist[d][b'], total[d] = sum_b hist[d][b]  (the digit bases,
// a 256-entry prefix sum of the totals, are formed by every scatter block itself: one launch instead of a 3-kernel
// scan over the whole 256 x nblk histogram)
__device__ __forceinline__ void radix_digit_scan_body(const int* __restrict__ hist, int nblk, int* within, int* total,
                                                      unsigned bx) {
  __shared__ int part[256];
  const int d = (int)bx, t = threadIdx.x;
  const int per = (nblk + 255) / 256;
  const int b0 = t * per, b1 = min(nblk, b0 + per);
  const int* h = hist + (long long)d * nblk;
  int s = 0;
  for (int b = b0; b < b1; ++b) s += h[b];
  part[t] = s;
  __syncthreads();
  for (int o = 1; o < 256; o <<= 1) {      // inclusive Hillis-Steele scan of the 256 chunk sums
    int v = (t >= o) ? part[t - o] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  int run = part[t] - s;
  int* w = within + (long long)d * nblk;
  for (int b = b0; b < b1; ++b) {
    w[b] = run;
    run += h[b];
  }
  if (t == 255) total[d] = part[255];
}
__global__ void __launch_bounds__(256) k_radix_digit_scan(const int* __restrict__ hist, int nblk, int* within,
                                                          int* total) {
  radix_digit_scan_body(hist, nblk, within, total, blockIdx.x);
}

__device__ __forceinline__ void radix_scatter_body(const unsigned* __restrict__ keys, const int* __restrict__ vals,
                                                   long long n, int shift, int nblk, const int* __restrict__ within,
                                                   const int* __restrict__ total, unsigned* keys_out, int* vals_out,
                                                   unsigned bx) {
  __shared__ int base_[256];
  const int lane = threadIdx.x;
  {   // digit bases = exclusive prefix sum of total[0..255]: lane l owns digits 4l .. 4l+3
    const int t0 = total[4 * lane], t1 = total[4 * lane + 1], t2 = total[4 * lane + 2], t3 = total[4 * lane + 3];
    const int mine = t0 + t1 + t2 + t3;
    int inc = mine;
    for (int o = 1; o < 64; o <<= 1) {
      int v = __shfl_up(inc, o);
      if (lane >= o) inc += v;
    }
    const int ex = inc - mine;
    const long long bb = bx;
    base_[4 * lane] = ex + within[(long long)(4 * lane) * nblk + bb];
    base_[4 * lane + 1] = ex + t0 + within[(long long)(4 * lane + 1) * nblk + bb];
    base_[4 * lane + 2] = ex + t0 + t1 + within[(long long)(4 * lane + 2) * nblk + bb];
    base_[4 * lane + 3] = ex + t0 + t1 + t2 + within[(long long)(4 * lane + 3) * nblk + bb];
  }
  __syncthreads();
  const long long base = (long long)bx * RS_BLOCK;
  const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  for (int c = 0; c < RS_BLOCK / 64; ++c) {
    long long i = base + c * 64 + lane;
    bool valid = i < n;
    unsigned key = valid ? keys[i] : 0u;
    int val = valid ? vals[i] : 0;
    unsigned d = (key >> shift) & 255u;
    unsigned long long peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      unsigned long long bb = __ballot(valid && ((d >> b) & 1u));
      peers &= ((d >> b) & 1u) ? bb : ~bb;
    }
    int rank = __popcll(peers & lt);
    int pos = 0;
    if (valid) pos = base_[d] + rank;
    __syncthreads();   // single-wave block: orders the reads above before the updates below
    if (valid && rank == 0) base_[d] += __popcll(peers);
    __syncthreads();
    if (valid) {
      keys_out[pos] = key;
      vals_out[pos] = val;
    }
  }
}
__global__ void __launch_bounds__(64) k_radix_scatter(const unsigned* __restrict__ keys, const int* __restrict__ vals,
                                                      long long n, int shift, int nblk,
                                                      const int* __restrict__ within, const int* __restrict__ total,
                                                      unsigned* keys_out, int* vals_out) {
  radix_scatter_body(keys, vals, n, shift, nblk, within, total, keys_out, vals_out, blockIdx.x);
}

// Windowed variant: rows are mask-sorted only INSIDE windows of WIN consecutive rows of the loader's order, which is
// spatially coherent (scan order) -- a wave tile then gathers from a compact region (L2 reuse of the gathered rows)
// at the price of more distinct masks per tile.  One workgroup sorts one window in LDS (bitonic on (mask, row)
// pairs: unique keys, so the result equals a stable sort); no global passes.
template <int WIN>
__global__ void __launch_bounds__(256) k_window_sort(const int* __restrict__ tbl, int K, long long n, int* order,
                                                     unsigned* keys_sorted, const int* __restrict__ pre) {
  __shared__ unsigned long long kv[WIN];
  const long long base = (long long)blockIdx.x * WIN;
  for (int e = threadIdx.x; e < WIN; e += 256) {
    long long v = base + e;
    unsigned long long key = ~0ull;            // padding sorts last
    if (v < n) {
      const long long row = pre ? pre[v] : v;  // windows of a spatial pre-order (gcl_spatial_order) or of the row order
      unsigned m = 0;
      for (int k = 0; k < K; ++k) m |= (tbl[(long long)k * n + row] >= 0 ? 1u : 0u) << k;
      key = ((unsigned long long)m << 32) | (unsigned)e;
    }
    kv[e] = key;
  }
  __syncthreads();
  for (int size = 2; size <= WIN; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int e = threadIdx.x; e < WIN / 2; e += 256) {
        int lo = 2 * e - (e & (stride - 1));   // index of the lower element of the pair
        int hi = lo + stride;
        bool up = ((lo & size) == 0);
        unsigned long long a = kv[lo], b = kv[hi];
        if ((a > b) == up) {
          kv[lo] = b;
          kv[hi] = a;
        }
      }
      __syncthreads();
    }
  }
  for (int e = threadIdx.x; e < WIN; e += 256) {
    long long j = base + e;
    if (j < n) {
      unsigned long long key = kv[e];
      const long long v = base + (unsigned)(key & 0xffffffffu);
      order[j] = pre ? pre[v] : (int)v;
      keys_sorted[j] = (unsigned)(key >> 32);
    }
  }
}

// 16-bit spatial key of a row: cloud id (5 bits, wraps) above an 11-bit Morton code of its cell -- 16 x 16 voxels in x / y
// (4 bits each: 256 voxels before the code wraps), 4 voxels in z (3 bits) -- so that rows close in space get close keys.
// A LOCALITY heuristic only (which rows run together on a CU / XCD and re-use each other's gathered neighbours in L2):
// results never depend on it.
__global__ void k_spatial_keys(const int4* __restrict__ coords, long long n, int tstride, unsigned* keys, int* vals) {
  long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= n) return;
  const int4 c = coords[v];
  const unsigned cx = (unsigned)((c.y / tstride + 4096) >> 4), cy = (unsigned)((c.z / tstride + 4096) >> 4),
                 cz = (unsigned)((c.w / tstride + 4096) >> 2);
  unsigned m = 0;
#pragma unroll
  for (int b = 3; b >= 0; --b) {       // interleaved from the most significant bit: coarse cells first
    m = (m << 1) | ((cx >> b) & 1u);
    m = (m << 1) | ((cy >> b) & 1u);
    if (b < 3) m = (m << 1) | ((cz >> b) & 1u);
  }
  keys[v] = (((unsigned)c.x & 31u) << 11) | (m & 2047u);
  vals[v] = (int)v;
}

// tile_mask (optional): an offset missing from a 32-row tile's mask is -1 for all of its rows by definition -- written
// without the (scattered, 4-byte) read of the source table: ~19 of 27 offsets on the KITTI batch
__device__ __forceinline__ void permute_table_body(const int* __restrict__ tbl, const int* __restrict__ order, long long n,
                                                   int* tbl_sorted, const int* __restrict__ tile_mask, unsigned bx, int k) {
  long long j = (long long)bx * blockDim.x + threadIdx.x;
  if (j >= n) return;
  int v = -1;
  if (!tile_mask || ((tile_mask[j >> 5] >> k) & 1)) v = tbl[(long long)k * n + order[j]];
  tbl_sorted[(long long)k * n + j] = v;
}
__global__ void k_permute_table(const int* __restrict__ tbl, const int* __restrict__ order, long long n,
                                int* tbl_sorted, const int* __restrict__ tile_mask) {
  permute_table_body(tbl, order, n, tbl_sorted, tile_mask, blockIdx.x, (int)blockIdx.y);
}

// pos (optional): the keys are masks with bit k moved to pos[k] (k_mask_keys); the tile mask is returned in offset order.
// Lane = row (coalesced key reads), OR over the 32 lanes of a tile by butterfly shuffles.
__device__ __forceinline__ void tile_masks_body(const unsigned* __restrict__ keys_sorted, long long n, long long n_tiles,
                                                int* tile_mask, const int* __restrict__ pos, int K, unsigned bx) {
  const long long j = (long long)bx * blockDim.x + threadIdx.x;
  unsigned m = (j < n) ? keys_sorted[j] : 0u;
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) m |= __shfl_xor(m, o);
  if ((threadIdx.x & 31) != 0) return;
  const long long t = j >> 5;
  if (t >= n_tiles) return;
  if (pos) {
    unsigned u = 0;
    for (int k = 0; k < K; ++k) u |= ((m >> pos[k]) & 1u) << k;
    m = u;
  }
  tile_mask[t] = (int)m;
}
__global__ void __launch_bounds__(256) k_tile_masks(const unsigned* __restrict__ keys_sorted, long long n, long long n_tiles,
                                                    int* tile_mask, const int* __restrict__ pos, int K) {
  tile_masks_body(keys_sorted, n, n_tiles, tile_mask, pos, K, blockIdx.x);
}

// The 3^3 stride-1 map of a level from its 5^3 stride-1 map: offset (dx, dy, dz) in {-1, 0, 1}^3 is row
// (dx + 2) + 5 (dy + 2) + 25 (dz + 2) of the 5^3 table and row (dx + 1) + 3 (dy + 1) + 9 (dz + 1) of the 3^3 one -- the same
// lookups, already answered (a copy of 27 rows instead of 13 hash probes per voxel).  blockIdx.y = 3^3 offset.
__global__ void __launch_bounds__(256) k_map3_from_map5(const int* __restrict__ nbr5, const int* __restrict__ counts5,
                                                        long long n, int* __restrict__ nbr3, int* __restrict__ counts3) {
  const int k3 = blockIdx.y;
  const int k5 = (k3 % 3 + 1) + 5 * ((k3 / 3) % 3 + 1) + 25 * (k3 / 9 + 1);
  const long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (v < n) nbr3[(long long)k3 * n + v] = nbr5[(long long)k5 * n + v];
  if (blockIdx.x == 0 && threadIdx.x == 0) counts3[k3] = counts5[k5];
}

// presence words of a neighbour table: bit (k & 31) of bits[v][k >> 5] = nbr[k][v] >= 0 (the first layer's occupancy path)
// one thread per (row, word of 32 offsets), eight independent loads per trip (round 6: one thread per row walked its 125
// offsets as a chain of dependent or-accumulations: 161 us on the benchmark batch's 265 MB table, 34 us on a 36 k-voxel pass)
__global__ void __launch_bounds__(256) k_presence_bits(const int* __restrict__ nbr, int K, long long n, unsigned* bits) {
  const int words = (K + 31) >> 5;
  const long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int q = blockIdx.y;
  if (v >= n) return;
  const int k0 = q * 32, k1 = (k0 + 32 < K) ? k0 + 32 : K;
  unsigned m = 0;
  int k = k0;
  for (; k + 8 <= k1; k += 8) {
    int r[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = nbr[(long long)(k + j) * n + v];
#pragma unroll
    for (int j = 0; j < 8; ++j) m |= (r[j] >= 0 ? 1u : 0u) << ((k + j) & 31);
  }
  for (; k < k1; ++k) m |= (nbr[(long long)k * n + v] >= 0 ? 1u : 0u) << (k & 31);
  bits[v * words + q] = m;
}
// per-cloud flags: cloud[b] = 1 when some feature of a row with batch index b differs from 1.0f (cloud[] zeroed by the
// entry; rows whose batch index does not fit the array are flagged themselves in the second pass)
__global__ void __launch_bounds__(256) k_not_ones_clouds(const float* __restrict__ x, int cin, const int* __restrict__ coords,
                                                         long long n, int* cloud, int n_cloud) {
  for (long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x; v < n; v += (long long)gridDim.x * blockDim.x) {
    bool bad = false;
    for (int c = 0; c < cin; ++c) bad = bad || (x[v * cin + c] != 1.0f);
    const int b = coords[v * 4];
    if (bad && b >= 0 && b < n_cloud) cloud[b] = 1;      // same value from every writer
  }
}
// rows[v] = 0 when every feature of v's cloud is 1.0f -- a row's kernel-map neighbours share its batch index
__global__ void __launch_bounds__(256) k_not_ones_rows(const int* __restrict__ coords, long long n,
                                                       const int* __restrict__ cloud, int n_cloud, int* rows) {
  const long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= n) return;
  const int b = coords[v * 4];
  rows[v] = (b >= 0 && b < n_cloud) ? cloud[b] : 1;
}

// ---- the same mask sort for SEVERAL tables per launch (gcl_table_sort_multi) --------------------------------------
// blockIdx.y (permute: z) = table; a table's blocks beyond its own grid return at once.  Same bodies, same results: what
// changes is the number of launches -- a network's 12 tables take 14 launches instead of 168, which is what a
// single-cloud inference pass (a chain of tiny dependent launches) and the trainer's map stream are made of.
constexpr int SORT_MAX_JOBS = 16;
struct SortJob {
  const int* tbl;
  long long n;
  int K, nblk, shift;
  const unsigned* kin;
  unsigned* kout;
  const int* vin;
  int* vout;
  int *hist, *offs, *bs, *bit_count, *key_pos, *order, *tbl_sorted, *tile_mask;
};
struct SortJobs {
  int count;
  SortJob j[SORT_MAX_JOBS];
};
__global__ void __launch_bounds__(256) k_row_masks_multi(SortJobs J) {
  const SortJob& q = J.j[blockIdx.y];
  if (blockIdx.x == 0 && threadIdx.x < 32 && q.bit_count) q.bit_count[threadIdx.x] = 0;      // read by the NEXT launch
  if ((long long)blockIdx.x * 256 >= q.n) return;
  row_masks_body(q.tbl, q.K, q.n, q.kout, q.vout, blockIdx.x);
}
__global__ void __launch_bounds__(256) k_mask_bit_count_multi(SortJobs J) {
  const SortJob& q = J.j[blockIdx.y];
  mask_bit_count_body(q.kin, q.n, q.K, q.bit_count, blockIdx.x, gridDim.x);
}
__global__ void __launch_bounds__(256) k_mask_keys_multi(SortJobs J) {
  const SortJob& q = J.j[blockIdx.y];
  if ((long long)blockIdx.x * 256 >= q.n) return;
  mask_keys_body(q.kout, q.n, q.K, q.bit_count, q.key_pos, blockIdx.x);
}
// masks AND sort keys in one launch, when the per-offset row counts are known beforehand (gcl_sort_job.counts: the kernel
// map's pair counts ARE the masks' bit counts -- offset k of a row is set exactly when the row owns a pair of offset k, in
// the table and in its transpose): no k_mask_bit_count pass, no separate key pass
__global__ void __launch_bounds__(256) k_row_masks_keys_multi(SortJobs J) {
  const SortJob& q = J.j[blockIdx.y];
  if ((long long)blockIdx.x * 256 >= q.n) return;
  __shared__ int pos[32];
  if (threadIdx.x < 32) {
    const int k = threadIdx.x;
    int p = 0;
    if (k < q.K) {
      const int ck = q.bit_count[k];
      for (int j = 0; j < q.K; ++j) {
        const int cj = q.bit_count[j];
        p += (cj > ck || (cj == ck && j < k)) ? 1 : 0;
      }
    }
    pos[k] = p;
    if (blockIdx.x == 0 && k < q.K) q.key_pos[k] = p;
  }
  __syncthreads();
  const long long v = (long long)blockIdx.x * 256 + threadIdx.x;
  if (v >= q.n) return;
  unsigned key = 0;
  for (int k = 0; k < q.K; ++k) key |= (q.tbl[(long long)k * q.n + v] >= 0 ? 1u : 0u) << pos[k];
  q.kout[v] = key;
  q.vout[v] = (int)v;
}
__global__ void __launch_bounds__(64) k_radix_hist_multi(SortJobs J) {
  const SortJob& q = J.j[blockIdx.y];
  if ((int)blockIdx.x >= q.nblk) return;
  radix_hist_body(q.kin, q.n, q.shift, q.nblk, q.hist, blockIdx.x);
}
__global__ void __launch_bounds__(256) k_radix_digit_scan_multi(SortJobs J) {
  const SortJob& q = J.j[blockIdx.y];
  radix_digit_scan_body(q.hist, q.nblk, q.offs, q.bs, blockIdx.x);
}
__global__ void __launch_bounds__(64) k_radix_scatter_multi(SortJobs J) {
  const SortJob& q = J.j[blockIdx.y];
  if ((int)blockIdx.x >= q.nblk) return;
  radix_scatter_body(q.kin, q.vin, q.n, q.shift, q.nblk, q.offs, q.bs, q.kout, q.vout, blockIdx.x);
}
__global__ void __launch_bounds__(256) k_tile_masks_multi(SortJobs J) {
  const SortJob& q = J.j[blockIdx.y];
  const long long n_tiles = (q.n + 31) / 32;
  if ((long long)blockIdx.x * 256 >= n_tiles * 32) return;
  tile_masks_body(q.kin, q.n, n_tiles, q.tile_mask, q.key_pos, q.K, blockIdx.x);
}
__global__ void __launch_bounds__(256) k_permute_table_multi(SortJobs J) {
  const SortJob& q = J.j[blockIdx.z];
  if ((long long)blockIdx.x * 256 >= q.n || (int)blockIdx.y >= q.K) return;
  permute_table_body(q.tbl, q.order, q.n, q.tbl_sorted, q.tile_mask, blockIdx.x, (int)blockIdx.y);
}

// ---- strided level, short form: flags + block totals in one launch, final scan + emit in another ------------------------
// (k_stride_flag + k_scan_reduce, k_scan_final + k_stride_emit: a pass over one pair of clouds walks three strided levels of
// <= 36 k rows, six small dependent launches each at ~ 4.8 us of dispatch latency; now three.)  flag[i] = slot + 1 of the
// row's voxel when row i is its first occurrence, else 0 -- the emit pass needs no second probe.
__global__ void __launch_bounds__(SCAN_T) k_stride_flag_reduce(const int4* __restrict__ coords, long long n_max,
                                                               const int* __restrict__ n_dev, int t_out, const Slot* t,
                                                               long long cap, int* __restrict__ flag, int* __restrict__ bs) {
  __shared__ int lds[8];
  const long long base = (long long)blockIdx.x * SCAN_B + threadIdx.x * SCAN_I;
  const long long nvalid = n_dev ? (long long)*n_dev : n_max;
  int s = 0;
#pragma unroll
  for (int j = 0; j < SCAN_I; ++j) {
    const long long i = base + j;
    int f = 0;
    if (i < n_max && i < nvalid) {
      int4 c = coords[i];
      c.y = floor_div(c.y, t_out) * t_out;
      c.z = floor_div(c.z, t_out) * t_out;
      c.w = floor_div(c.w, t_out) * t_out;
      if (pack_ok(c.x, c.y, c.z, c.w)) {
        const long long sl = table_find(t, cap, pack_key(c.x, c.y, c.z, c.w));
        if (sl >= 0 && t[sl].val == i) f = (int)sl + 1;
      }
    }
    if (i < n_max) flag[i] = f;
    s += f != 0;
  }
  int tot;
  block_excl_scan(s, lds, &tot);
  if (threadIdx.x == 0) bs[blockIdx.x] = tot;
}

__global__ void __launch_bounds__(SCAN_T) k_stride_final_emit(const int4* __restrict__ coords, long long n_max, int t_out, Slot* t,
                                                              const int* __restrict__ flag, const int* __restrict__ bs,
                                                              int4* __restrict__ coords_out, int* n_out, long long* index_out) {
  __shared__ int lds[8];
  int before = 0;
  {
    const int mine = ((int)threadIdx.x < (int)blockIdx.x) ? bs[threadIdx.x] : 0;      // at most SCAN_T blocks
    block_excl_scan(mine, lds, &before);
  }
  const long long base = (long long)blockIdx.x * SCAN_B + threadIdx.x * SCAN_I;
  int f[SCAN_I];
  int s = 0;
#pragma unroll
  for (int j = 0; j < SCAN_I; ++j) {
    f[j] = (base + j < n_max) ? flag[base + j] : 0;
    s += f[j] != 0;
  }
  int tot;
  int ex = block_excl_scan(s, lds, &tot) + before;
#pragma unroll
  for (int j = 0; j < SCAN_I; ++j) {
    const long long i = base + j;
    if (i < n_max && f[j]) {
      int4 c = coords[i];
      c.y = floor_div(c.y, t_out) * t_out;
      c.z = floor_div(c.z, t_out) * t_out;
      c.w = floor_div(c.w, t_out) * t_out;
      t[f[j] - 1].val = ex;
      coords_out[ex] = c;
      if (index_out) index_out[ex] = i;
      ++ex;
    }
    if (i == n_max - 1) *n_out = ex;
  }
}

static bool is_pow2(int64_t v) { return v > 0 && (v & (v - 1)) == 0; }

// EMPTY-fills `slots` consecutive table slots (several levels' tables at once) and zeroes the four status words
void tables_fill(int64_t* tables, long long slots, int32_t* zero4, hipStream_t st) {
  hipLaunchKernelGGL(k_table_fill, dim3((unsigned)cdiv(slots, 256)), dim3(256), 0, st, (Slot*)tables, slots, zero4);
}

// table_ready: the table is EMPTY-filled and the status words are zero already (tables_fill over all levels' tables)
int coords_insert_impl(const int32_t* coords, int64_t n, int64_t* table, int64_t cap, int32_t* status, bool table_ready,
                       void* stream) {
  GCL_CHECK_ARG(coords && table && status, "gcl_coords_insert: null pointer");
  GCL_CHECK_ARG(is_pow2(cap) && cap >= 2 * n && cap >= 64, "gcl_coords_insert: cap must be a power of two >= 2n");
  hipStream_t st = (hipStream_t)stream;
  if (!table_ready)
    hipLaunchKernelGGL(k_table_fill, dim3((unsigned)cdiv(cap, 256)), dim3(256), 0, st, (Slot*)table, (long long)cap, status);
  if (n > 0)
    hipLaunchKernelGGL(k_coords_insert, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, st, (const int4*)coords,
                       (long long)n, (Slot*)table, (long long)cap, status);
  GCL_CHECK_LAUNCH();
  return GCL_OK;
}

int stride_map_impl(const int32_t* coords_in, int64_t n_in, const int32_t* n_in_dev, int32_t t_out, int64_t* table_out,
                    int64_t cap_out, int32_t* scratch, int32_t* coords_out, int32_t* n_out_dev, int32_t* status,
                    int64_t* index_out, bool table_ready, void* stream) {
  GCL_CHECK_ARG(coords_in && table_out && scratch && coords_out && n_out_dev && status, "gcl_stride_map: null pointer");
  GCL_CHECK_ARG(n_in > 0 && t_out >= 1, "gcl_stride_map: n_in and t_out must be positive");
  GCL_CHECK_ARG(is_pow2(cap_out) && cap_out >= 2 * n_in && cap_out >= 64, "gcl_stride_map: cap must be a power of two >= 2n");
  hipStream_t st = (hipStream_t)stream;
  int* flag = scratch;
  int* pos = scratch + n_in;
  int* bs = scratch + 2 * n_in;
  unsigned g = (unsigned)cdiv(n_in, 256);
  if (!table_ready)
    hipLaunchKernelGGL(k_table_fill, dim3((unsigned)cdiv(cap_out, 256)), dim3(256), 0, st, (Slot*)table_out,
                       (long long)cap_out, status);
  hipLaunchKernelGGL(k_stride_insert, dim3(g), dim3(256), 0, st, (const int4*)coords_in, (long long)n_in,
                     (const int*)n_in_dev, t_out, (Slot*)table_out, (long long)cap_out, status);
  const long long nb = cdiv(n_in, SCAN_B);
  if (nb <= SCAN_T) {      // short form: three launches per level instead of six
    hipLaunchKernelGGL(k_stride_flag_reduce, dim3((unsigned)nb), dim3(SCAN_T), 0, st, (const int4*)coords_in, (long long)n_in,
                       (const int*)n_in_dev, t_out, (const Slot*)table_out, (long long)cap_out, flag, bs);
    hipLaunchKernelGGL(k_stride_final_emit, dim3((unsigned)nb), dim3(SCAN_T), 0, st, (const int4*)coords_in, (long long)n_in,
                       t_out, (Slot*)table_out, (const int*)flag, (const int*)bs, (int4*)coords_out, n_out_dev,
                       (long long*)index_out);
    GCL_CHECK_LAUNCH();
    return GCL_OK;
  }
  hipLaunchKernelGGL(k_stride_flag, dim3(g), dim3(256), 0, st, (const int4*)coords_in, (long long)n_in,
                     (const int*)n_in_dev, t_out, (const Slot*)table_out, (long long)cap_out, flag);
  GCL_CHECK_LAUNCH();
  int rc = device_scan(flag, n_in, pos, bs, st);
  if (rc) return rc;
  hipLaunchKernelGGL(k_stride_emit, dim3(g), dim3(256), 0, st, (const int4*)coords_in, (long long)n_in, t_out,
                     (Slot*)table_out, (long long)cap_out, (const int*)flag, (const int*)pos, (int4*)coords_out,
                     n_out_dev, (long long*)index_out);
  GCL_CHECK_LAUNCH();
  return GCL_OK;
}

}  // namespace gcl

using namespace gcl;

extern "C" {

const char* gcl_last_error(void) { return g_err; }
int gcl_version(void) { return 100; }
int gcl_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return GCL_ERR_NO_DEVICE;
  return n;
}

int gcl_coords_insert(const int32_t* coords, int64_t n, int64_t* table, int64_t cap, int32_t* status,
                      void* stream) {
  return coords_insert_impl(coords, n, table, cap, status, false, stream);
}

int64_t gcl_scan_scratch_len(int64_t n) { return 2 * n + cdiv(n, SCAN_B) + 64; }

int gcl_stride_map(const int32_t* coords_in, int64_t n_in, const int32_t* n_in_dev, int32_t t_out, int64_t* table_out,
                   int64_t cap_out, int32_t* scratch, int32_t* coords_out, int32_t* n_out_dev, int32_t* status,
                   void* stream) {
  return stride_map_impl(coords_in, n_in, n_in_dev, t_out, table_out, cap_out, scratch, coords_out, n_out_dev, status,
                         nullptr, false, stream);
}

int gcl_unique_coords(const int32_t* coords_in, int64_t n_in, int64_t* table_out, int64_t cap_out, int32_t* scratch,
                      int32_t* coords_out, int64_t* index_out, int32_t* n_out_dev, int32_t* status, void* stream) {
  GCL_CHECK_ARG(index_out, "gcl_unique_coords: null pointer");
  return stride_map_impl(coords_in, n_in, nullptr, 1, table_out, cap_out, scratch, coords_out, n_out_dev, status,
                         index_out, false, stream);
}

int gcl_exclusive_scan_i32(const int32_t* in, int64_t n, int32_t* out, int32_t* scratch, void* stream) {
  GCL_CHECK_ARG(in && out && scratch && n > 0, "gcl_exclusive_scan_i32: bad argument");
  return device_scan(in, n, out, scratch, (hipStream_t)stream);
}

int64_t gcl_kernel_map_bitmap_len(void) { return BITMAP_WORDS; }
int64_t gcl_kernel_map_scratch_len(int32_t ks, int64_t n_out) { return (long long)ks * ks * ks * cdiv(n_out, 256); }

int gcl_kernel_map(const int32_t* coords_out, int64_t n_out, const int64_t* table_in, int64_t cap_in, int32_t ks,
                   int32_t step, int32_t same_map, int32_t* bitmap, int32_t* scratch, int32_t* nbr, int32_t* nbr_t,
                   int64_t n_in, int32_t* counts, void* stream) {
  GCL_CHECK_ARG(coords_out && table_in && nbr && counts && scratch, "gcl_kernel_map: null pointer");
  GCL_CHECK_ARG(ks >= 1 && (ks & 1) && ks <= 5, "gcl_kernel_map: kernel size must be 1, 3 or 5");
  GCL_CHECK_ARG(n_out > 0 && step >= 1 && is_pow2(cap_in), "gcl_kernel_map: bad sizes");
  const bool bitmap_valid = (same_map & 2) != 0;      // bit 1: `bitmap` already describes table_in (built by an earlier call)
  // bit 2: `counts` is zero on entry and the build is short (<= 1024 blocks): per-offset counts by integer atomics, no
  // k_count_reduce launch, `scratch` unused
  // (64 blocks = 16 384 rows.  The K counters share two cache lines: at 256 / 1024 blocks the atomics cost more than the
  // launch they save -- eight pairs per pass 97.4 -> 89 - 96 / 87 - 94 M voxels/s, the training step 11.94 -> 12.0 / 12.35 ms)
  static const long long acc_max = [] { const char* e = getenv("GCL_MAP_ACC_MAX_BLOCKS"); return e ? atoll(e) : 64ll; }();
  const int acc = ((same_map & 4) != 0 && cdiv(n_out, 256) <= acc_max) ? 1 : 0;
  const bool prefilled = (same_map & 8) != 0;      // bit 3: nbr / nbr_t hold -1 everywhere on entry (one fill for all maps)
  same_map &= 1;
  GCL_CHECK_ARG(!same_map || (nbr_t == nullptr && n_in == n_out), "gcl_kernel_map: same_map excludes nbr_t");
  hipStream_t st = (hipStream_t)stream;
  int K = ks * ks * ks;
  int nblk = (int)cdiv(n_out, 256);
  static const int kpt = [] {      // GCL_MAP_KPT = 1 | 2 | 4: offsets per thread of the lookup kernels
    const char* e = getenv("GCL_MAP_KPT");
    const int v = e ? atoi(e) : 4;
    return (v == 1 || v == 2) ? v : 4;
  }();
  if (bitmap && !bitmap_valid) {
    fill32(bitmap, BITMAP_WORDS, 0u, st);
    hipLaunchKernelGGL(k_bitmap_fill, dim3((unsigned)cdiv(cap_in, 256)), dim3(256), 0, st, (const Slot*)table_in,
                       (long long)cap_in, (unsigned*)bitmap);
  }
  if (same_map) {
    if (K > 1 && !prefilled)
      fill32(nbr + (size_t)(K / 2 + 1) * n_out, (long long)(K / 2) * n_out, 0xFFFFFFFFu, st);
    if (kpt == 4)
      hipLaunchKernelGGL(k_kernel_map_sym<4>, dim3(nblk, (unsigned)cdiv(K / 2 + 1, 4)), dim3(256), 0, st, (const int4*)coords_out,
                         (long long)n_out, (const Slot*)table_in, (long long)cap_in, (const unsigned*)bitmap, ks, step,
                         nbr, acc ? counts : scratch, acc);
    else if (kpt == 2)
      hipLaunchKernelGGL(k_kernel_map_sym<2>, dim3(nblk, (unsigned)cdiv(K / 2 + 1, 2)), dim3(256), 0, st, (const int4*)coords_out,
                         (long long)n_out, (const Slot*)table_in, (long long)cap_in, (const unsigned*)bitmap, ks, step,
                         nbr, acc ? counts : scratch, acc);
    else
      hipLaunchKernelGGL(k_kernel_map_sym<1>, dim3(nblk, K / 2 + 1), dim3(256), 0, st, (const int4*)coords_out,
                         (long long)n_out, (const Slot*)table_in, (long long)cap_in, (const unsigned*)bitmap, ks, step,
                         nbr, acc ? counts : scratch, acc);
    if (!acc) hipLaunchKernelGGL(k_count_reduce, dim3(K / 2 + 1), dim3(256), 0, st, (const int*)scratch, nblk, K, 1, counts);
  } else {
    if (nbr_t && !prefilled) fill32(nbr_t, (long long)K * n_in, 0xFFFFFFFFu, st);
    if (kpt == 4)
      hipLaunchKernelGGL(k_kernel_map<4>, dim3(nblk, (unsigned)cdiv(K, 4)), dim3(256), 0, st, (const int4*)coords_out,
                         (long long)n_out, (const Slot*)table_in, (long long)cap_in, (const unsigned*)bitmap, ks, step, nbr,
                         nbr_t, (long long)n_in, acc ? counts : scratch, acc);
    else if (kpt == 2)
      hipLaunchKernelGGL(k_kernel_map<2>, dim3(nblk, (unsigned)cdiv(K, 2)), dim3(256), 0, st, (const int4*)coords_out,
                         (long long)n_out, (const Slot*)table_in, (long long)cap_in, (const unsigned*)bitmap, ks, step, nbr,
                         nbr_t, (long long)n_in, acc ? counts : scratch, acc);
    else
      hipLaunchKernelGGL(k_kernel_map<1>, dim3(nblk, K), dim3(256), 0, st, (const int4*)coords_out, (long long)n_out,
                         (const Slot*)table_in, (long long)cap_in, (const unsigned*)bitmap, ks, step, nbr, nbr_t,
                         (long long)n_in, acc ? counts : scratch, acc);
    if (!acc) hipLaunchKernelGGL(k_count_reduce, dim3(K), dim3(256), 0, st, (const int*)scratch, nblk, K, 0, counts);
  }
  GCL_CHECK_LAUNCH();
  return GCL_OK;
}

int gcl_kernel_map_pairs(const int32_t* nbr, int32_t K, int64_t n_out, const int64_t* seg_off_host, int32_t* scratch,
                         int32_t* pair_in, int32_t* pair_out, void* stream) {
  GCL_CHECK_ARG(nbr && seg_off_host && scratch && pair_in && pair_out, "gcl_kernel_map_pairs: null pointer");
  GCL_CHECK_ARG(K >= 1 && K <= 125 && n_out > 0, "gcl_kernel_map_pairs: bad K / n_out");
  hipStream_t st = (hipStream_t)stream;
  SegOff seg;
  for (int k = 0; k <= K; ++k) {
    seg.off[k] = seg_off_host[k];
    GCL_CHECK_ARG(seg.off[k] % GCL_PAIR_CHUNK == 0, "gcl_kernel_map_pairs: segment offsets must be multiples of %d",
                  GCL_PAIR_CHUNK);
  }
  int nb = (int)cdiv(n_out, PAIR_B);
  hipLaunchKernelGGL(k_pairs_count, dim3(nb, K), dim3(256), 0, st, nbr, (long long)n_out, nb, scratch);
  hipLaunchKernelGGL(k_pairs_scan, dim3(K), dim3(256), 0, st, scratch, nb);
  hipLaunchKernelGGL(k_pairs_emit, dim3(nb, K), dim3(256), 0, st, nbr, (long long)n_out, nb, (const int*)scratch, seg,
                     pair_in, pair_out);
  GCL_CHECK_LAUNCH();
  return GCL_OK;
}

int64_t gcl_table_sort_scratch_len(int64_t n) {
  long long nblk = cdiv(n, RS_BLOCK);
  long long hist = 256 * nblk;
  return 4 * n + 2 * hist + cdiv(hist, SCAN_B) + 64 + 256 + 64;   // keys/vals ping-pong, hist, within, digit totals, offset counts + key bit positions
}

int gcl_spatial_order(const int32_t* coords, int64_t n, int32_t tensor_stride, int32_t* scratch, int32_t* order,
                      void* stream) {
  GCL_CHECK_ARG(coords && scratch && order, "gcl_spatial_order: null pointer");
  GCL_CHECK_ARG(n > 0 && tensor_stride >= 1, "gcl_spatial_order: bad sizes");
  hipStream_t st = (hipStream_t)stream;
  const int nblk = (int)cdiv(n, RS_BLOCK);
  const long long hist_len = 256ll * nblk;
  unsigned* ka = (unsigned*)scratch;
  unsigned* kb = ka + n;
  int* va = scratch + 2 * n;
  int* vb = scratch + 3 * n;
  int* hist = scratch + 4 * n;
  int* offs = hist + hist_len;
  int* bs = offs + hist_len;
  hipLaunchKernelGGL(k_spatial_keys, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, st, (const int4*)coords, (long long)n,
                     tensor_stride, ka, va);
  for (int p = 0; p < 2; ++p) {      // 16-bit keys: two stable 8-bit passes
    hipLaunchKernelGGL(k_radix_hist, dim3(nblk), dim3(64), 0, st, (const unsigned*)ka, (long long)n, 8 * p, nblk, hist);
    hipLaunchKernelGGL(k_radix_digit_scan, dim3(256), dim3(256), 0, st, (const int*)hist, nblk, offs, bs);
    int* vout = (p == 1) ? order : vb;
    hipLaunchKernelGGL(k_radix_scatter, dim3(nblk), dim3(64), 0, st, (const unsigned*)ka, (const int*)va, (long long)n,
                       8 * p, nblk, (const int*)offs, (const int*)bs, kb, vout);
    unsigned* tk = ka; ka = kb; kb = tk;
    int* tv = va; va = vb; vb = tv;
  }
  GCL_CHECK_LAUNCH();
  return GCL_OK;
}

int gcl_table_sort(const int32_t* tbl, int32_t K, int64_t n, int32_t window, int32_t* scratch, int32_t* order,
                   int32_t* tbl_sorted, int32_t* tile_mask, void* stream) {
  return gcl_table_sort_pre(tbl, K, n, window, nullptr, scratch, order, tbl_sorted, tile_mask, stream);
}

int gcl_table_sort_pre(const int32_t* tbl, int32_t K, int64_t n, int32_t window, const int32_t* pre, int32_t* scratch,
                       int32_t* order, int32_t* tbl_sorted, int32_t* tile_mask, void* stream) {
  GCL_CHECK_ARG(tbl && scratch && order && tbl_sorted && tile_mask, "gcl_table_sort: null pointer");
  GCL_CHECK_ARG(!pre || window, "gcl_table_sort_pre: a pre-order needs window = 2048 or 4096");
  GCL_CHECK_ARG(K >= 1 && K <= 27 && n > 0, "gcl_table_sort: K must be <= 27 (3^3 kernels), n > 0");
  GCL_CHECK_ARG(window == 0 || window == 2048 || window == 4096, "gcl_table_sort: window must be 0, 2048 or 4096");
  hipStream_t st = (hipStream_t)stream;
  unsigned* ka = (unsigned*)scratch;
  int* bit_count = nullptr;
  int* key_pos = nullptr;
  if (window) {
    if (window == 2048)
      hipLaunchKernelGGL(k_window_sort<2048>, dim3((unsigned)cdiv(n, 2048)), dim3(256), 0, st, tbl, K, (long long)n,
                         order, ka, pre);
    else
      hipLaunchKernelGGL(k_window_sort<4096>, dim3((unsigned)cdiv(n, 4096)), dim3(256), 0, st, tbl, K, (long long)n,
                         order, ka, pre);
  } else {
    int nblk = (int)cdiv(n, RS_BLOCK);
    long long hist_len = 256ll * nblk;
    unsigned* kb = ka + n;
    int* va = scratch + 2 * n;
    int* vb = scratch + 3 * n;
    int* hist = scratch + 4 * n;
    int* offs = hist + hist_len;
    int* bs = offs + hist_len;
    static const int freq_order = [] { const char* e = getenv("GCL_SORT_FREQ_ORDER"); return e ? atoi(e) : 1; }();
    if (freq_order) {
      bit_count = scratch + gcl_table_sort_scratch_len(n) - 64;
      key_pos = bit_count + 32;
      GCL_CHECK_HIP(hipMemsetAsync(bit_count, 0, 32 * sizeof(int32_t), st));
    }
    hipLaunchKernelGGL(k_row_masks, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, st, tbl, K, (long long)n, ka, va);
    if (freq_order) {
      hipLaunchKernelGGL(k_mask_bit_count, dim3(512), dim3(256), 0, st, (const unsigned*)ka, (long long)n, K, bit_count);
      hipLaunchKernelGGL(k_mask_keys, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, st, ka, (long long)n, K,
                         (const int*)bit_count, key_pos);
    }
    int passes = (K + 7) / 8;
    // at most three 8-bit passes: a 27-offset mask is ordered by its 24 most significant bits (offsets 3..26); the
    // three dropped bits only permute rows inside runs that already share 24 bits (measured: MFMA work unchanged,
    // one pass of ~45 us per table saved)
    static const int max_passes = [] { const char* e = getenv("GCL_SORT_PASSES"); int v = e ? atoi(e) : 3; return v < 1 ? 1 : (v > 4 ? 4 : v); }();
    int base = 0;
    if (passes > max_passes) {
      base = K - 8 * max_passes;
      passes = max_passes;
    }
    for (int p = 0; p < passes; ++p) {
      hipLaunchKernelGGL(k_radix_hist, dim3(nblk), dim3(64), 0, st, (const unsigned*)ka, (long long)n, base + 8 * p, nblk, hist);
      GCL_CHECK_LAUNCH();
      hipLaunchKernelGGL(k_radix_digit_scan, dim3(256), dim3(256), 0, st, (const int*)hist, nblk, offs, bs);
      int* vout = (p == passes - 1) ? order : vb;
      hipLaunchKernelGGL(k_radix_scatter, dim3(nblk), dim3(64), 0, st, (const unsigned*)ka, (const int*)va, (long long)n,
                         base + 8 * p, nblk, (const int*)offs, (const int*)bs, kb, vout);
      unsigned* tk = ka; ka = kb; kb = tk;
      if (p != passes - 1) { int* tv = va; va = vb; vb = tv; }
    }
  }
  long long n_tiles = cdiv(n, 32);
  hipLaunchKernelGGL(k_tile_masks, dim3((unsigned)cdiv(n_tiles * 32, 256)), dim3(256), 0, st, (const unsigned*)ka,
                     (long long)n, n_tiles, tile_mask, (const int*)key_pos, K);
  hipLaunchKernelGGL(k_permute_table, dim3((unsigned)cdiv(n, 256), K), dim3(256), 0, st, tbl, (const int*)order,
                     (long long)n, tbl_sorted, (const int*)tile_mask);
  GCL_CHECK_LAUNCH();
  return GCL_OK;
}

int gcl_kernel_map_3_from_5(const int32_t* nbr5, const int32_t* counts5, int64_t n, int32_t* nbr3, int32_t* counts3,
                            void* stream) {
  GCL_CHECK_ARG(nbr5 && counts5 && nbr3 && counts3 && n > 0, "gcl_kernel_map_3_from_5: bad argument");
  hipLaunchKernelGGL(k_map3_from_map5, dim3((unsigned)cdiv(n, 256), 27), dim3(256), 0, (hipStream_t)stream, nbr5, counts5,
                     (long long)n, nbr3, counts3);
  GCL_CHECK_LAUNCH();
  return GCL_OK;
}

int gcl_presence_bits(const int32_t* nbr, int32_t K, int64_t n, uint32_t* bits, void* stream) {
  GCL_CHECK_ARG(nbr && bits && K >= 1 && K <= 128 && n > 0, "gcl_presence_bits: bad argument");
  hipLaunchKernelGGL(k_presence_bits, dim3((unsigned)cdiv(n, 256), (unsigned)((K + 31) / 32)), dim3(256), 0, (hipStream_t)stream,
                     nbr, K, (long long)n, (unsigned*)bits);
  GCL_CHECK_LAUNCH();
  return GCL_OK;
}

int gcl_not_ones_rows(const float* x, int32_t cin, const int32_t* coords, int64_t n, int32_t* cloud_flags,
                      int32_t n_cloud_flags, int32_t* row_flags, void* stream) {
  GCL_CHECK_ARG(x && coords && cloud_flags && row_flags && n > 0 && cin >= 1 && n_cloud_flags >= 1,
                "gcl_not_ones_rows: bad argument");
  hipStream_t st = (hipStream_t)stream;
  GCL_CHECK_HIP(hipMemsetAsync(cloud_flags, 0, sizeof(int32_t) * (size_t)n_cloud_flags, st));
  long long g = cdiv(n, 256 * 4);
  if (g > 1024) g = 1024;
  hipLaunchKernelGGL(k_not_ones_clouds, dim3((unsigned)g), dim3(256), 0, st, x, cin, (const int*)coords, (long long)n,
                     (int*)cloud_flags, n_cloud_flags);
  hipLaunchKernelGGL(k_not_ones_rows, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, st, (const int*)coords, (long long)n,
                     (const int*)cloud_flags, n_cloud_flags, (int*)row_flags);
  GCL_CHECK_LAUNCH();
  return GCL_OK;
}

int gcl_table_sort_multi(const gcl_sort_job* jobs_host, int32_t n_jobs, void* stream) {
  GCL_CHECK_ARG(jobs_host && n_jobs >= 1, "gcl_table_sort_multi: bad argument");
  hipStream_t st = (hipStream_t)stream;
  static const int max_passes = [] { const char* e = getenv("GCL_SORT_PASSES"); int v = e ? atoi(e) : 3; return v < 1 ? 1 : (v > 4 ? 4 : v); }();
  for (int j0 = 0; j0 < n_jobs; j0 += SORT_MAX_JOBS) {
    const int T = (n_jobs - j0 < SORT_MAX_JOBS) ? n_jobs - j0 : SORT_MAX_JOBS;
    SortJobs J;
    J.count = T;
    long long n_max = 0;
    bool counts_known = true;
    int nblk_max = 0, K0 = jobs_host[j0].K;
    unsigned* ka[SORT_MAX_JOBS];
    unsigned* kb[SORT_MAX_JOBS];
    int* va[SORT_MAX_JOBS];
    int* vb[SORT_MAX_JOBS];
    for (int t = 0; t < T; ++t) {
      const gcl_sort_job& g = jobs_host[j0 + t];
      GCL_CHECK_ARG(g.tbl && g.scratch && g.order && g.tbl_sorted && g.tile_mask, "gcl_table_sort_multi: null pointer");
      GCL_CHECK_ARG(g.K >= 1 && g.K <= 27 && g.n > 0 && g.K == K0, "gcl_table_sort_multi: tables of one call share K <= 27");
      SortJob& q = J.j[t];
      const long long n = g.n;
      const int nblk = (int)cdiv(n, RS_BLOCK);
      const long long hist_len = 256ll * nblk;
      ka[t] = (unsigned*)g.scratch;
      kb[t] = ka[t] + n;
      va[t] = g.scratch + 2 * n;
      vb[t] = g.scratch + 3 * n;
      q.tbl = g.tbl; q.n = n; q.K = g.K; q.nblk = nblk; q.shift = 0;
      q.hist = g.scratch + 4 * n;
      q.offs = q.hist + hist_len;
      q.bs = q.offs + hist_len;
      q.bit_count = g.scratch + gcl_table_sort_scratch_len(n) - 64;
      q.key_pos = q.bit_count + 32;
      if (g.counts) q.bit_count = (int*)g.counts;       // read only on this path
      else counts_known = false;
      q.order = g.order; q.tbl_sorted = g.tbl_sorted; q.tile_mask = g.tile_mask;
      q.kin = ka[t]; q.kout = ka[t]; q.vin = va[t]; q.vout = va[t];
      if (n > n_max) n_max = n;
      if (nblk > nblk_max) nblk_max = nblk;
    }
    const unsigned gn = (unsigned)cdiv(n_max, 256);
    if (counts_known) {
      hipLaunchKernelGGL(k_row_masks_keys_multi, dim3(gn, T), dim3(256), 0, st, J);     // sort keys -> ka, row ids -> va
    } else {
    for (int t = 0; t < T; ++t) {      // (a mixed call measures every table)
      J.j[t].bit_count = jobs_host[j0 + t].scratch + gcl_table_sort_scratch_len(jobs_host[j0 + t].n) - 64;
    }
    hipLaunchKernelGGL(k_row_masks_multi, dim3(gn, T), dim3(256), 0, st, J);            // masks -> ka, row ids -> va
    // (grid-stride body: as many workgroups as the largest table has 1024-row pieces, at most 512 -- a pass over one pair has
    // 36 of them, and 512 workgroups per table each reduced 27 counters for nothing: 38 us)
    hipLaunchKernelGGL(k_mask_bit_count_multi, dim3((unsigned)std::min<long long>(512, cdiv(n_max, 1024)), T), dim3(256), 0, st, J);
    hipLaunchKernelGGL(k_mask_keys_multi, dim3(gn, T), dim3(256), 0, st, J);             // ka: masks -> sort keys
    }
    int passes = (K0 + 7) / 8, base = 0;
    if (passes > max_passes) {
      base = K0 - 8 * max_passes;
      passes = max_passes;
    }
    for (int p = 0; p < passes; ++p) {
      for (int t = 0; t < T; ++t) {
        SortJob& q = J.j[t];
        q.shift = base + 8 * p;
        q.kin = ka[t]; q.vin = va[t];
        q.kout = kb[t];
        q.vout = (p == passes - 1) ? q.order : vb[t];
      }
      hipLaunchKernelGGL(k_radix_hist_multi, dim3(nblk_max, T), dim3(64), 0, st, J);
      hipLaunchKernelGGL(k_radix_digit_scan_multi, dim3(256, T), dim3(256), 0, st, J);
      hipLaunchKernelGGL(k_radix_scatter_multi, dim3(nblk_max, T), dim3(64), 0, st, J);
      for (int t = 0; t < T; ++t) {
        unsigned* tk = ka[t]; ka[t] = kb[t]; kb[t] = tk;
        if (p != passes - 1) { int* tv = va[t]; va[t] = vb[t]; vb[t] = tv; }
      }
    }
    for (int t = 0; t < T; ++t) J.j[t].kin = ka[t];                                      // the sorted keys
    hipLaunchKernelGGL(k_tile_masks_multi, dim3((unsigned)cdiv(cdiv(n_max, 32) * 32, 256), T), dim3(256), 0, st, J);
    hipLaunchKernelGGL(k_permute_table_multi, dim3(gn, K0, T), dim3(256), 0, st, J);
    GCL_CHECK_LAUNCH();
  }
  return GCL_OK;
}

}  // extern "C"
