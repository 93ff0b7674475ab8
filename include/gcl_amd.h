/*
 * gcl_amd.h -- C ABI of libgcl_hip.so, the MI355X (gfx950) native library under the
 * MinkowskiEngine-compatible operator surface used by liuQuan98/GCL's hot path.
 *
 * Boundary.  The reference reaches its native code only through the third-party
 * Python module `MinkowskiEngine` (model/resunet.py:3-4); the entry points below are what that module's
 * native half has to provide for the calls the reference makes:
 *
 *   ME.SparseTensor(feats, coordinates=...)          lib/colocation_trainer.py:843-845,
 *                                                    scripts/test_kitti.py:143-147, util/misc.py:128
 *        -> gcl_coords_insert                        (coordinate map of the stride-1 tensor)
 *   ME.MinkowskiConvolution / ...Transpose           model/resunet.py:38-171, model/residual_block.py:23-33
 *        -> gcl_stride_map, gcl_kernel_map, gcl_kernel_map_pairs   (coordinate manager, cached per key)
 *        -> gcl_pack_weights, gcl_conv_fwd (forward AND input-gradient), gcl_conv_bwd_weight,
 *           gcl_stem_fwd / gcl_stem_bwd_weight       (Cin <= 4 first layer, model/resunet.py:38-45)
 *   ME.MinkowskiBatchNorm + MEF.relu + `out += residual`   model/common.py:4-6, model/residual_block.py:37-53
 *        -> gcl_bn_stats, gcl_bn_apply, gcl_bn_bwd_reduce, gcl_bn_bwd_apply
 *   out.F / torch.norm(out.F, p=2, dim=1, keepdim=True)   model/resunet.py:226-230  -> gcl_row_normalize_fwd/bwd
 *   finest_contrastive_loss                          lib/colocation_trainer.py:430-535
 *        -> gcl_group_loss_fwd/bwd, gcl_nn_rowmin (pdist + min, lib/metrics.py:22-25),
 *           gcl_neg_mask, gcl_neg_loss_fwd/bwd
 *   find_nn_gpu                                      lib/eval.py:18-48  -> gcl_nn_rowmin
 *   Matcher.estimator (SC2-PCR)                      scripts/test_kitti.py:172-180, scripts/SC2_PCR/SC2_PCR.py
 *        -> gcl_nn_rowmin, gcl_sc2_*
 *
 * Conventions
 *   - plain C types only; every pointer is a DEVICE pointer unless its name ends in _host;
 *   - the caller owns all memory (the Python host allocates torch tensors); the library never allocates
 *     persistent device memory and never synchronises the stream;
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream);
 *   - return 0 on success, <0 on error; gcl_last_error() gives the message of the calling thread's last error;
 *   - row-major contiguous matrices; features are fp32, indices int32, coordinates int32 [N,4] = (batch,x,y,z).
 */
#ifndef GCL_AMD_H
#define GCL_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GCL_OK 0
#define GCL_ERR_ARG (-1)       /* bad argument (shape not supported, null pointer, ...) */
#define GCL_ERR_HIP (-2)       /* a HIP runtime call failed */
#define GCL_ERR_NO_DEVICE (-3) /* no gfx950 device visible */

#define GCL_PAIR_CHUNK 128     /* pair lists are padded per kernel offset to a multiple of this */

const char* gcl_last_error(void);
int gcl_version(void);
/* number of visible devices, or <0; never throws -- used by the host to fail loudly without a GPU */
int gcl_device_count(void);

/* ------------------------------------------------------------------------------------------------
 * Coordinate maps.  A map is an open-addressing hash table the CALLER allocates:
 *   table: int64 [cap, 2]  (slot = {packed key, row index}), cap = power of two >= 2 * n.
 * Keys pack (batch, x, y, z) into 16 bits each (x,y,z offset by 2^15).
 * status (int32[4], device): [0] = #rows with a coordinate outside the packable range,
 *                            [1] = #duplicate rows met by gcl_coords_insert.
 * ---------------------------------------------------------------------------------------------- */
int gcl_coords_insert(const int32_t* coords, int64_t n, int64_t* table, int64_t cap,
                      int32_t* status, void* stream);

/* Strided coordinate map: out = unique(floor(c / t_out) * t_out), rows ordered by first occurrence in
 * `coords_in` (deterministic).  Writes coords_out (capacity n_in rows), n_out_dev (device int32) and
 * fills `table_out` (cap_out >= 2 * n_in) with out-key -> out-row.  scratch: int32[gcl_scan_scratch_len(n_in)].
 * n_in_dev (optional, device int32): the true row count when n_in is only an upper bound -- lets the host chain
 * the maps of several levels without reading a count back in between. */
int64_t gcl_scan_scratch_len(int64_t n);
int gcl_stride_map(const int32_t* coords_in, int64_t n_in, const int32_t* n_in_dev, int32_t t_out,
                   int64_t* table_out, int64_t cap_out, int32_t* scratch,
                   int32_t* coords_out, int32_t* n_out_dev, int32_t* status, void* stream);

/* Device-wide exclusive prefix sum of int32 (scratch: int32[n / 2048 + 64]); used by the map builders. */
int gcl_exclusive_scan_i32(const int32_t* in, int64_t n, int32_t* out, int32_t* scratch, void* stream);

/* ------------------------------------------------------------------------------------------------
 * "Next" rows: the loader work either side of the hot path, on the device.
 * gcl_voxel_coords + gcl_unique_coords = ME.utils.sparse_quantize(xyz / voxel, return_index=True)
 *   (util/misc.py:117-118, lib/colocation_data_loader.py:379,388): coords = floor(xyz / voxel) (correctly rounded
 *   fp32 division, floor toward -inf), one row per voxel = first occurrence, kept rows in ascending order;
 *   index_out[j] = input row of output row j; table_out maps the kept coordinates to their output rows.
 * gcl_colocation_hits + gcl_colocation_emit = get_matching_indices_colocation (util/pointcloud.py:69-132):
 *   xyz_own [Ntot,3]: voxel-representative points of the centre cloud followed by the neighbour clouds, each in its
 *   OWN sensor frame; xyz_cf: the same points in the CENTRE frame (neighbours transformed by list_M, fp32);
 *   table: coordinate map of floor(xyz_own / voxel) with batch id = cloud id; to_cloud_host: double[n_clouds][12],
 *   row-major 3x4 transforms centre frame -> cloud frame (identity for cloud 0), used only to find candidate voxels.
 *   hits [n_center, n_clouds, K] (rows into xyz_*, ascending distance, -1 padded), cnt, first_rng: per (centre point,
 *   cloud).  emit: group [G], index [sum], finest [sum] (exactly one 1 per group), totals = {G, sum} on the device.
 * ---------------------------------------------------------------------------------------------- */
int gcl_voxel_coords(const float* xyz, int64_t p, float voxel, int32_t batch_id, int32_t* coords, void* stream);
int gcl_unique_coords(const int32_t* coords_in, int64_t n_in, int64_t* table_out, int64_t cap_out, int32_t* scratch,
                      int32_t* coords_out, int64_t* index_out, int32_t* n_out_dev, int32_t* status, void* stream);
int gcl_colocation_hits(const float* xyz_own, const float* xyz_cf, int64_t n_center, int32_t n_clouds,
                        const double* to_cloud_host, const int64_t* table, int64_t cap, float inv_voxel, double radius,
                        int32_t K, int32_t* hits, int32_t* cnt, double* first_rng, void* stream);
int gcl_colocation_emit(const int32_t* hits, const int32_t* cnt, const double* first_rng, int64_t n_center,
                        int32_t n_clouds, int32_t K, int32_t* scratch, int32_t* group, int64_t* index, uint8_t* finest,
                        int32_t* totals, void* stream);

/* A whole BATCH of samples in one pass (round 6; what ColocationKittiDataset.__getitem__ x batch_size + collate_colocation_fn
 * produce, lib/colocation_data_loader.py:315-475): the clouds of ALL samples share ONE coordinate table whose batch id is the
 * cloud's number in the batch (the ids sparse_collate gives them, :440-446), so a batch costs two host synchronisations.
 *   gcl_voxel_coords_multi  gcl_voxel_coords over the concatenated points of n_clouds <= 64 clouds; cloud_offsets_host:
 *                           HOST int64[n_clouds + 1] point offsets (offsets[0] = 0, offsets[n_clouds] = p).
 *   gcl_cloud_row_starts    after gcl_unique_coords (rows keep the cloud-major input order): starts int32[n_clouds + 1] on the
 *                           device, starts[c] = first row of cloud c (-1: the cloud has none), starts[n_clouds] = *n_dev.
 *   gcl_loader_points       voxel representatives xyz_own[i] = xyz_raw[index[i]] and, in the centre frame of the row's sample,
 *                           xyz_cf[i] = fp32(R p + t) evaluated in fp64 without contraction; to_center_dev: DEVICE
 *                           double[n_clouds][12] (row-major 3 x 4; identity for a centre cloud).  n_max bounds the launch,
 *                           *n_dev rows are written.
 *   gcl_colocation_hits_at  gcl_colocation_hits for ONE sample of the batch: its centre voxels are rows row0 .. row0 + n_center
 *                           of xyz_*, its clouds are the table's batch ids cloud0 .. cloud0 + n_clouds - 1; hits are rows of
 *                           the batch (= the `index` collate_colocation_fn makes by adding the sample's offset, :434-437).
 *                           hits / cnt / first_rng point at the sample's slice; gcl_colocation_emit then runs once over the
 *                           concatenated centre voxels of all samples. */
int gcl_voxel_coords_multi(const float* xyz, int64_t p, const int64_t* cloud_offsets_host, int32_t n_clouds, float voxel,
                            int32_t* coords, void* stream);
int gcl_cloud_row_starts(const int32_t* coords, int64_t n_max, const int32_t* n_dev, int32_t n_clouds, int32_t* starts,
                         void* stream);
int gcl_loader_points(const float* xyz_raw, const int64_t* index, const int32_t* coords, int64_t n_max, const int32_t* n_dev,
                      const double* to_center_dev, float* xyz_own, float* xyz_cf, void* stream);
int gcl_colocation_hits_at(const float* xyz_own, const float* xyz_cf, int64_t row0, int64_t n_center, int32_t cloud0,
                           int32_t n_clouds, const double* to_cloud_host, const int64_t* table, int64_t cap, float inv_voxel,
                           double radius, int32_t K, int32_t* hits, int32_t* cnt, double* first_rng, void* stream);

/* HOST function (no GPU, every pointer is a HOST pointer): numpy's legacy np.random.choice(n, k, replace=False) =
 * permutation(n)[:k] -- the three draws of every training step (lib/colocation_trainer.py:457, :506-507) -- reproduced
 * bit for bit from the RandomState's MT19937 state (key[624], *pos as in np.random.get_state(); both updated in place so
 * that np.random.set_state continues the stream), outside the interpreter lock: numpy needs 8 ms per call at 0.5 M rows and
 * holds the lock meanwhile, which stalls the thread that enqueues the GPU work.  work: int64[n + n / 32 + 64] scratch,
 * out: int64[k]. */
int gcl_host_legacy_choice(uint32_t* key, int32_t* pos, int64_t n, int64_t k, int64_t* work, int64_t* out);

/* Kernel map for kernel size ks^3 (x fastest in k), offsets scaled by `step` (= input tensor stride x dilation),
 * region centred on the OUTPUT coordinate:  nbr[k * n_out + v] = input row at c_out[v] + o_k * step, or -1.
 * same_map != 0: coords_out IS the input map (stride-1 conv): only offsets k <= K/2 are looked up, the mirror
 *   entries nbr[(K-1-k) * n + u] = v are written from the hits; nbr_t must be NULL.
 * same_map == 0 and nbr_t != NULL (n_in rows): nbr_t[k * n_in + u] = v for every pair (filled with -1 first).
 * same_map bit 1 (value 2): `bitmap` was already filled for THIS table_in by an earlier call (the three kernel maps that
 *   read the stride-1 table of a step share one fill); bit 0 is the same-map flag described above.
 * same_map bit 2 (value 4): `counts` is ZERO on entry (gcl_maps_build zeroes every map's counts with one fill): builds of
 *   <= 64 blocks (16 384 output rows) then add their per-block counts with integer atomics -- the same totals, one launch
 *   less per map, `scratch` unused; longer builds take the ordered reduction as before (the counters share two cache lines).
 * same_map bit 3 (value 8): nbr (and nbr_t) hold -1 in EVERY entry on entry (gcl_maps_build pre-fills the tables of a small
 *   build with one launch): the call's own fills are skipped.
 * bitmap: optional int32[gcl_kernel_map_bitmap_len()] scratch: a presence bit per hashed key lets most absent
 *   neighbours return without probing the table.  scratch: int32[gcl_kernel_map_scratch_len(ks, n_out)] (per-block
 *   pair counts, summed in order -- no contended atomics).  counts[k] (int32[K], device) = #pairs of offset k. */
int64_t gcl_kernel_map_bitmap_len(void);
int64_t gcl_kernel_map_scratch_len(int32_t ks, int64_t n_out);
int gcl_kernel_map(const int32_t* coords_out, int64_t n_out, const int64_t* table_in, int64_t cap_in,
                   int32_t ks, int32_t step, int32_t same_map, int32_t* bitmap, int32_t* scratch, int32_t* nbr,
                   int32_t* nbr_t, int64_t n_in, int32_t* counts, void* stream);

/* Compact per-offset pair lists from nbr (out-major, ascending out row inside an offset), each offset's
 * segment padded with -1 to a multiple of GCL_PAIR_CHUNK:
 *   seg_off_host: int64[K+1] HOST array computed by the caller from `counts` (padded prefix sums);
 *   pair_in / pair_out: int32[seg_off_host[K]].  scratch: int32[K * ceil(n_out/1024) + K + 1]. */
int gcl_kernel_map_pairs(const int32_t* nbr, int32_t K, int64_t n_out, const int64_t* seg_off_host,
                         int32_t* scratch, int32_t* pair_in, int32_t* pair_out, void* stream);

/* Row ordering for the output-stationary convolution: sorts the rows of a [K][n] neighbour table (K <= 27) by
 * their K-bit presence mask (stable radix sort), so that the rows of a 32-row wave tile need nearly the same offsets.
 * Global mode: the sort key is the mask with its bits re-ordered by how often each offset occurs in THIS table -- the
 * rarest offset is the most significant key bit, the most frequent the least significant, ties: lower offset lower --
 * and the sort runs on the key's 24 most significant bits (key >> max(0, K - 24), three 8-bit passes).  Rows that own a
 * rare offset then share its visit.  The convolution walks the tiles from the END of this order (the tiles with the
 * most offsets first).  order[j] = original row at sorted position j; tbl_sorted[k*n + j] = tbl[k*n + order[j]];
 * tile_mask[t] = OR of the masks of sorted rows 32t .. 32t+31.  scratch: int32[gcl_table_sort_scratch_len(n)].
 * window = 0: one global sort.  window = 2048 | 4096: rows are sorted only inside windows of that many consecutive
 * rows (one LDS bitonic sort per window): keeps the loader's spatial coherence for the gathers. */
int64_t gcl_table_sort_scratch_len(int64_t n);
int gcl_table_sort(const int32_t* tbl, int32_t K, int64_t n, int32_t window, int32_t* scratch, int32_t* order,
                   int32_t* tbl_sorted, int32_t* tile_mask, void* stream);
/* gcl_table_sort (global mode, window 0) of SEVERAL tables in one sequence of 14 launches (a network has 12 such tables:
 * 168 launches one by one).  Same results bit for bit.  All tables of a call share K. */
typedef struct gcl_sort_job {
  const int32_t* tbl;
  int32_t K;
  int64_t n;
  int32_t* scratch;      /* int32[gcl_table_sort_scratch_len(n)] */
  int32_t* order;
  int32_t* tbl_sorted;
  int32_t* tile_mask;
  const int32_t* counts; /* optional int32[K], device: rows per offset (= gcl_kernel_map's counts, for nbr and for nbr_t alike);
                            when EVERY table of the call has it the bit-count and key passes fold into the first launch */
} gcl_sort_job;
int gcl_table_sort_multi(const gcl_sort_job* jobs_host, int32_t n_jobs, void* stream);
/* Spatial pre-order of the rows of a coordinate map at tensor stride `tensor_stride`: order[j] = row at position j when
 * rows are sorted by (cloud id, Morton code of the 4^3-voxel cell); stable, deterministic.  gcl_table_sort_pre with
 * window = 2048 | 4096 then mask-sorts inside windows of THAT order: the rows a workgroup (and, with
 * GCL_CONV_XCD_RANGES, an XCD) processes together are spatial neighbours and re-use each other's gathered input rows in
 * L2 instead of fetching every row from the Infinity Cache once per kernel offset.  A locality heuristic only: results
 * do not depend on it.  scratch: int32[gcl_table_sort_scratch_len(n)]. */
int gcl_spatial_order(const int32_t* coords, int64_t n, int32_t tensor_stride, int32_t* scratch, int32_t* order,
                      void* stream);
int gcl_table_sort_pre(const int32_t* tbl, int32_t K, int64_t n, int32_t window, const int32_t* pre, int32_t* scratch,
                       int32_t* order, int32_t* tbl_sorted, int32_t* tile_mask, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Sparse convolution (fp32, exact-f32 MFMA).
 * gcl_pack_weights: W [K, Cin, Cout] -> MFMA B-fragment order.
 *   mode 0: forward weights            (Cin_eff = Cin,  Cout_eff = Cout, k kept)
 *   mode 1: transposed per offset      (Cin_eff = Cout, Cout_eff = Cin,  k kept)     input-gradient, in != out map
 *   mode 2: transposed + offsets mirrored (k -> K-1-k)                               input-gradient, same map
 *   prec 0: exact-f32 MFMA (v_mfma_f32_32x32x2_f32), wp = float[K * Cin * Cout];
 *   prec 3: fp32 operands split into 3 bf16 planes, 6 bf16-MFMA terms, fp32 accumulate ("bf16x6": error vs fp64
 *           equal to native fp32, 2.7x the MFMA rate);  prec 2: 2 planes, 3 terms ("bf16x3", ~1.5e-5, 5.3x);
 *   prec 4: 2 fp16 planes of the operands scaled by a per-tensor power of two (gcl_amax), 3 fp16-MFMA terms
 *           ("fp16x3": 11+11 significand bits with round-to-nearest = 24 bits, i.e. native-fp32 accuracy at half the
 *           MFMA work of bf16x6); x_amax / w_amax are required in this mode only;
 *           wp = gcl_pack_weights_bytes(...) bytes of bf16 planes.
 *   Shapes: the MFMA kernels take Cin, Cout multiples of 32 and K <= 27.  EVERY OTHER shape (any Cin, Cout > 0, K <= 125;
 *   e.g. the 16-dim head of demo.py:29, `final` 64 -> 16) is a "generic shape": gcl_pack_weights then writes plain fp32
 *   W_eff[k][Cin_eff][Cout_eff] (prec ignored), gcl_conv_fwd / gcl_conv_fwd_fused / gcl_conv_bwd_weight run exact-fp32
 *   VALU kernels (amax pointers, plane images and `stats` are not used / not accepted there).
 * gcl_conv_fwd: Y[row(j)] = sum_k X[tbl[k*n_out+j]] . Wp_k (+ bias); tbl == NULL means K == 1, identity.
 *   x_is_planes != 0 (prec 4 only): x points at the gcl_split_planes image of X instead of X (no split in the kernel).
 *   X has n_in rows (n_in * Cin * 4 < 4 GiB: rows are gathered through a buffer resource, absent neighbours read 0).
 *   With (order, tile_mask) from gcl_table_sort, `tbl` is the PERMUTED table, row(j) = order[j] and each 32-row
 *   wave tile visits only the offsets of tile_mask; with NULLs, row(j) = j and every offset is visited.
 *   Output-stationary (no atomics, deterministic).  The same entry computes the input gradient when given
 *   the opposite table and mode-1/2 weights.  K <= 27.
 *   stats (optional, split precisions): float[4][cout][ceil(n_out/128)] (channel-major) -- column sums of y and y^2,
 *   column minimum and maximum per 128-row workgroup tile (its four waves combined in order), consumed by
 *   gcl_bn_stats_from_tiles(_range) (the BatchNorm that follows then needs no statistics pass over y; n_tiles =
 *   ceil(n_out/128), its `scratch` is unused). */
int64_t gcl_pack_weights_bytes(int32_t K, int32_t cin, int32_t cout, int32_t prec);
/* max |x| of a tensor in an "amax slot": GCL_AMAX_WORDS device int32, 16 entries on separate 128-byte lines (entry i
 * at word 32 i), value = max over the entries, each the bit pattern of a non-negative float.  (Workgroups publish to
 * different lines: same-line atomics serialise on gfx950.)  The fp16x3 mode derives its exact power-of-two operand
 * scales from it.  zeroed != 0: the caller guarantees the slot is all zero on the stream (slots handed out from a
 * zero-filled pool), so no memset is issued. */
#define GCL_AMAX_WORDS 512
int gcl_amax(const float* x, int64_t n, int32_t* amax_bits, int32_t zeroed, void* stream);
/* gcl_amax of n_tensors tensors in one launch: ptrs / sizes are DEVICE arrays (float* and element counts),
 * amax_bits[n_tensors * GCL_AMAX_WORDS] (one slot per tensor) is cleared here.  Used once per step for all convolution kernels of a model. */
int gcl_amax_multi(const float* const* ptrs, const int64_t* sizes, int32_t n_tensors, int32_t* amax_bits,
                   void* stream);
int gcl_pack_weights(const float* w, int32_t K, int32_t cin, int32_t cout, int32_t mode, int32_t prec,
                     const int32_t* w_amax, void* wp, void* stream);
/* gcl_pack_weights for a list of tensors in ONE launch (all convolution kernels of a network, once per optimizer step
 * and direction).  desc: DEVICE int64[n_tensors][8] = {w pointer, K, cin, cout, mode, amax slot index, byte offset of
 * the packed tensor inside `out`, first workgroup}, first workgroup = running sum of ceil(K*cin*cout / 256);
 * total_wgs = that sum over all tensors; amax_slots: the slots written by gcl_amax_multi (prec 4).  prec 2, 3 or 4. */
int gcl_pack_weights_multi(const int64_t* desc, int32_t n_tensors, int64_t total_wgs, int32_t prec,
                           const int32_t* amax_slots, void* out, void* stream);
/* NB of the kernel instance gcl_conv_fwd will launch for this shape (a wave covers 32 NB output columns): 4, 2 or 1;
 * diagnostic (profile labels). */
int32_t gcl_conv_fwd_nb(int64_t n_out, int32_t cout, int32_t prec);
/* fp16x3 operand image of an activation tensor: planes = uint16 [n][c / 32][2][32] -- per row and 32-channel slice the
 * fp16 hi values (64 bytes) then the fp16 lo values (64 bytes) of x * scale(amax), i.e. 4 bytes per element like x.
 * A tensor that several launches consume (forward, weight gradient; input gradient, weight gradient) is split once. */
int gcl_split_planes(const float* x, int64_t n, int32_t c, const int32_t* amax, void* planes, void* stream);
/* flags: GCL_CONV_XCD_RANGES -- the table comes from gcl_table_sort_pre on a spatial pre-order: give every XCD (L2) a
 * CONTIGUOUS range of row tiles, so that the input rows a tile gathers are mostly the ones its neighbours on the same XCD
 * just gathered.  Launch-order hint only. */
#define GCL_CONV_XCD_RANGES 1
#define GCL_CONV_TALL 4     /* inference launches: sixteen waves per workgroup, a tile's offsets in four FIXED groups (k with
                               k mod 4 == g) whose partial sums are added in group order -- shortens the chain of dependent
                               steps a small cloud's deep layers are made of; a row's result depends on the layer's shape only
                               (batching stays bitwise neutral).  fp16x3 on fp32 rows, no BatchNorm statistics, K >= 8,
                               K Cin / 32 >= 108 (Cin >= 128 at K = 27), Cout a multiple of 64; ignored elsewhere.  Results differ from the launch
                               without the flag in the last bits (another summation order).
                               Round 6: with this flag the `stats` argument of gcl_conv_fwd(_fused(_ld)) is SCRATCH of
                               gcl_conv_fwd_groups_scratch_len(n_out, K, Cin, Cout) floats, or NULL.  With scratch (and a
                               non-zero length: <= 65536 output rows) the four groups run as four times as many ordinary
                               workgroups + one sum / epilogue launch -- bitwise the sixteen-wave kernel's result, 2 - 3 x
                               shorter on the deep layers of a pass over one or two clouds. */
#define GCL_CONV_DMA 2      /* plane-image launches (fp16x3): operands staged by LDS-DMA (`buffer_load ... lds`: no staging
                               registers, no ds_write); bitwise the same results.  The default (GCL_FWD_DMA=0 turns it off
                               unless this flag is set) */
#define GCL_CONV_NO_DMA 8   /* ... and this flag selects the register-staged kernel for a launch regardless */
int gcl_conv_fwd(const float* x, int64_t n_in, int32_t x_is_planes, const void* wp, int32_t prec, const int32_t* x_amax,
                 const int32_t* w_amax, const int32_t* tbl, const int32_t* order, const int32_t* tile_mask,
                 int64_t n_out, int32_t K,
                 int32_t cin, int32_t cout, const float* bias, float* y, float* stats, int32_t flags, void* stream);

/* gcl_conv_fwd with a fused inference epilogue: y = relu?(conv * col_scale + bias (+ residual)), i.e. convolution +
 * BatchNorm in eval mode (col_scale = gamma * rsqrt(var + eps), bias = beta - mean * col_scale) + BasicBlock's residual
 * add + ReLU in ONE launch (model/residual_block.py:37-53 with running statistics); y_amax (optional, zero-initialised
 * amax slot) receives max|y| for the next fp16x3 convolution.  Split-precision modes only.
 * relu == 2 (needs `residual`): aten::threshold_backward instead of the add -- y = conv where residual > 0, else 0: the
 * input gradient of a convolution whose input came out of a ReLU, with that ReLU's backward in the epilogue (`residual` =
 * the ReLU's output). */
int64_t gcl_conv_fwd_groups_scratch_len(int64_t n_out, int32_t K, int32_t cin, int32_t cout);
int gcl_conv_fwd_fused(const float* x, int64_t n_in, int32_t x_is_planes, const void* wp, int32_t prec,
                       const int32_t* x_amax, const int32_t* w_amax, const int32_t* tbl, const int32_t* order,
                       const int32_t* tile_mask, int64_t n_out, int32_t K, int32_t cin, int32_t cout, const float* bias,
                       const float* col_scale, const float* residual, int32_t relu, int32_t* y_amax, float* y,
                       float* stats, int32_t flags, void* stream);
/* the same with `residual` a column slice of a wider row-major tensor: row pitch residual_ld floats (0 = cout).  Used by the
 * backward pass for the gradient of an ME.cat input, which is a slice of the gradient of the cat's output (no split copy). */
int gcl_conv_fwd_fused_ld(const float* x, int64_t n_in, int32_t x_is_planes, const void* wp, int32_t prec,
                          const int32_t* x_amax, const int32_t* w_amax, const int32_t* tbl, const int32_t* order,
                          const int32_t* tile_mask, int64_t n_out, int32_t K, int32_t cin, int32_t cout, const float* bias,
                          const float* col_scale, const float* residual, int32_t residual_ld, int32_t relu,
                          int32_t* y_amax, float* y, float* stats, int32_t flags, void* stream);

/* dW[k] = sum over pairs of offset k of  A[pair_a]^T . B[pair_b]   (A: [*, ca], B: [*, cb]) -> dw [K, ca, cb].
 * Forward conv: A = X, pair_a = pair_in, B = dY, pair_b = pair_out.  Transposed conv: roles swapped.
 * Deterministic: per-wave partial slabs + ordered reduction.  prec as in gcl_conv_fwd (both operands are split
 * on the fly for prec 2 / 3).
 * planes != 0 (prec 4 only): a AND b point at gcl_split_planes images (no split, hardware-transposed LDS reads).
 * A has n_a rows, B n_b rows (each tensor < 4 GiB: rows are gathered through buffer resources, padding pairs read 0).
 * sorted_side: 0 = nothing known about the pair order; 1 / 2 = pair_a / pair_b ascends inside every offset segment (the
 *   out rows of gcl_kernel_map_pairs: pair_b for a forward convolution, pair_a for a transposed one).  With it, and Ca, Cb
 *   in {32, 64}, 1 < K <= 27, no plane images and >= 32768 rows on that side, the launch is "range-grouped": a workgroup
 *   takes the pairs of ONE offset whose sorted-side row lies in one row range, the K workgroups of a range share an XCD,
 *   and the sorted side's rows are fetched from the fabric once per range instead of once per offset (same sums, other
 *   fixed order).  scratch: float[gcl_conv_bwd_weight_scratch_len(K, ca, cb, n_pairs_padded, rows of the sorted side or 0)].
 * planes: bit 0 = a and b are plane images (gcl_split_planes); with Ca and Cb multiples of 128 the launch then gives every
 *   workgroup a 128 x 128 block of dW[k] whose rows are gathered once and shared by its four waves (half the gathered bytes
 *   per MFMA).  Bit 1 = keep the 64 x 64-block kernel for this launch (same result to rounding: another fixed order). */
int64_t gcl_conv_bwd_weight_scratch_len(int32_t K, int32_t ca, int32_t cb, int64_t n_pairs_padded, int64_t n_sorted_rows);
int gcl_conv_bwd_weight(const float* a, int64_t n_a, const float* b, int64_t n_b, int32_t planes, int32_t sorted_side,
                        const int32_t* pair_a, const int32_t* pair_b, const int64_t* seg_off_host, int32_t K,
                        int32_t ca, int32_t cb, int32_t prec, const int32_t* a_amax, const int32_t* b_amax, float* scratch, float* dw,
                        void* stream);
/* kernel_size-1 convolutions (conv1_tr, final: model/resunet.py:153-171; ME binds them to the same GEMM path with an
 * identity kernel map): dW[ca][cb] = sum over ALL n_rows rows of a[i][:]^T b[i][:] -- pair (i, i) for every row, so no pair
 * list is read and nothing is gathered: both operands are streamed once (k_bwd_weight_rows, fp16x3 split on the fly,
 * per-workgroup slabs summed in a fixed order: deterministic).  gcl_conv_bwd_weight_rows_scratch_len returns the scratch
 * floats, or 0 when the shape / arithmetic is not taken (then use gcl_conv_bwd_weight with identity pairs). */
int64_t gcl_conv_bwd_weight_rows_scratch_len(int32_t ca, int32_t cb, int32_t prec, int64_t n_rows);
int gcl_conv_bwd_weight_rows(const float* a, const float* b, int64_t n_rows, int32_t ca, int32_t cb, int32_t prec,
                             const int32_t* a_amax, const int32_t* b_amax, float* scratch, float* dw, void* stream);
/* The range-grouped mode's cell limits (first position of every (offset, row range) cell in the sorted pair list) depend
 * on the map alone: gcl_conv_bwd_weight_bounds makes them once (int32[gcl_conv_bwd_weight_bounds_len(K, rows of the sorted
 * side)], 0 = the mode does not apply), gcl_conv_bwd_weight_rg takes them (`rg_bounds`, NULL = computed per launch as
 * gcl_conv_bwd_weight does).  gcl_maps_build makes them for every map with pair lists (gcl_map_desc.dw_bounds: over
 * pair_out, the sorted list of the convolution and of its transpose alike). */
int64_t gcl_conv_bwd_weight_bounds_len(int32_t K, int64_t n_sorted_rows);
int gcl_conv_bwd_weight_bounds(const int32_t* sorted_rows, const int64_t* seg_off_host, int32_t K, int64_t n_sorted_rows,
                               int32_t* bounds, void* stream);
int gcl_conv_bwd_weight_rg(const float* a, int64_t n_a, const float* b, int64_t n_b, int32_t planes, int32_t sorted_side,
                           const int32_t* pair_a, const int32_t* pair_b, const int64_t* seg_off_host, int32_t K,
                           int32_t ca, int32_t cb, int32_t prec, const int32_t* a_amax, const int32_t* b_amax,
                           float* scratch, float* dw, const int32_t* rg_bounds, void* stream);

/* First layer (Cin <= 4, Cout a multiple of 32, any ks <= 5): VALU kernels over the nbr table (one 32-column block per
 * workgroup column).  Other first-layer widths go through gcl_conv_fwd / gcl_conv_bwd_weight (generic shapes). */
/* Occupancy rows: `presence` (uint32 [n_out][ceil(K / 32)], gcl_presence_bits of `nbr`) and `not_ones` (device int32
 * [n_out], 0 = every feature row v gathers equals 1.0f; gcl_not_ones_rows) are optional and go together; cin must be 1.
 * The reference's loaders feed torch.ones((n, 1)): its test loaders and scripts for every cloud, its training loaders for
 * the neighbour clouds of a sample -- only the centre cloud carries lib/transforms.py:18 Jitter
 * (lib/colocation_data_loader.py:401-415).  For a row with flag 0, x[nbr[k][v]] is bit k of the row's presence words:
 * gcl_stem_fwd adds W[k] over the SET bits, k ascending (k_stem_fwd_occ; the table kernel is enqueued too and skips
 * those rows), gcl_stem_bwd_weight fills its 0/1 tile from the words.  Same values added in the same order: bitwise
 * identical to the table path, which every flagged row takes (flags are read on the device, no host decision).
 * gcl_not_ones_rows: a row's kernel-map neighbours share its batch index (coords[v][0], part of the key), so the flag is
 * per cloud: cloud_flags (int32 [n_cloud_flags] scratch, zeroed here) gets 1 for a batch index with a feature != 1.0f,
 * row_flags[v] = cloud_flags[batch index of v] (1 when the index does not fit the scratch). */
/* the 3^3 stride-1 kernel map of a coordinate table from its 5^3 stride-1 map (the same neighbour lookups, already
 * answered: nbr3 = 27 rows of nbr5, counts3 = their counts) -- gcl_maps_build uses it when a network asks for both
 * (conv1 5^3 + block1 3^3 on the input table, model/resunet.py:38-46, :59-60); bit-exact the direct build */
int gcl_kernel_map_3_from_5(const int32_t* nbr5, const int32_t* counts5, int64_t n, int32_t* nbr3, int32_t* counts3,
                            void* stream);
int gcl_presence_bits(const int32_t* nbr, int32_t K, int64_t n, uint32_t* bits, void* stream);
int gcl_not_ones_rows(const float* x, int32_t cin, const int32_t* coords, int64_t n, int32_t* cloud_flags,
                      int32_t n_cloud_flags, int32_t* row_flags, void* stream);
int gcl_stem_fwd(const float* x, const float* w, const int32_t* nbr, int64_t n_out, int32_t K,
                 int32_t cin, int32_t cout, float* y, const uint32_t* presence, const int32_t* not_ones, void* stream);
int64_t gcl_stem_bwd_weight_scratch_len(int32_t K, int32_t cin, int32_t cout, int64_t n_out);
int gcl_stem_bwd_weight(const float* x, const float* dy, const int32_t* nbr, int64_t n_out, int32_t K,
                        int32_t cin, int32_t cout, float* scratch, float* dw, const uint32_t* presence,
                        const int32_t* not_ones, void* stream);

/* ------------------------------------------------------------------------------------------------
 * BatchNorm over rows of [n, c] fused with the residual add and ReLU of BasicBlock.
 *   gcl_bn_stats:  mean[c], rstd[c] (biased variance, eps) and running-stat update (momentum, unbiased var).
 *                  scratch: double[gcl_bn_scratch_len(n, c)].
 *   gcl_bn_apply:  y = x * scale + shift (+ residual) (relu);   scale = w * rstd, shift = b - mean * scale.
 *                  In eval mode the host passes running stats as mean / rstd.
 *   backward:      g = dy * (relu ? y > 0 : 1);  gcl_bn_bwd_reduce -> sum_g[c], sum_gx[c] (xhat-weighted);
 *                  gcl_bn_bwd_apply -> dx, (dres = g).
 *   relu_mask (optional, uint64[gcl_bn_mask_len(n, c)]): with relu, gcl_bn_apply also writes the sign bits of y
 *                  (1 bit per element); the backward passes then take the mask instead of re-reading y (pass
 *                  y = NULL), which removes a third of their HBM traffic.
 *   y_amax / dx_amax (optional, ZERO-INITIALISED amax slots): the apply passes also publish gcl_amax of the tensor they
 *                  write, so the fp16x3 convolution that consumes it needs no extra pass over it.
 * ---------------------------------------------------------------------------------------------- */
int64_t gcl_bn_scratch_len(int64_t n, int32_t c);
int gcl_bn_stats(const float* x, int64_t n, int32_t c, float eps, float momentum,
                 float* running_mean, float* running_var, double* scratch,
                 float* mean, float* rstd, void* stream);
int64_t gcl_bn_tiles_scratch_len(int64_t n_tiles, int32_t c);   /* doubles */
int gcl_bn_stats_from_tiles(const float* partial, int64_t n_tiles, int64_t n, int32_t c, float eps, float momentum,
                            float* running_mean, float* running_var, double* scratch, float* mean, float* rstd,
                            void* stream);
int gcl_bn_apply(const float* x, int64_t n, int32_t c, const float* mean, const float* rstd,
                 const float* weight, const float* bias, const float* residual, int32_t relu,
                 float* y, uint64_t* relu_mask, int32_t* y_amax, void* stream);
/* the same with y a column slice of a wider row-major tensor (row pitch y_ld floats, 0 = c): the plan writes the decoder
 * input of an ME.cat (model/resunet.py:206,213,220) straight into the cat's output */
int gcl_bn_apply_ld(const float* x, int64_t n, int32_t c, const float* mean, const float* rstd,
                    const float* weight, const float* bias, const float* residual, int32_t relu,
                    float* y, int32_t y_ld, uint64_t* relu_mask, int32_t* y_amax, void* stream);
int64_t gcl_bn_mask_len(int64_t n, int32_t c);                  /* uint64 words */
int gcl_bn_bwd_reduce(const float* x, const float* dy, const float* y, const uint64_t* relu_mask, int64_t n,
                      int32_t c, const float* mean, const float* rstd, int32_t relu, double* scratch,
                      float* sum_g, float* sum_gx, void* stream);
/* ... with dy a column slice of a wider tensor: row pitch dy_ld floats (0 = c) */
int gcl_bn_bwd_reduce_ld(const float* x, const float* dy, int32_t dy_ld, const float* y, const uint64_t* relu_mask, int64_t n,
                         int32_t c, const float* mean, const float* rstd, int32_t relu, double* scratch, float* sum_g,
                         float* sum_gx, void* stream);
int gcl_bn_bwd_apply(const float* x, const float* dy, const float* y, const uint64_t* relu_mask, int64_t n, int32_t c,
                     const float* mean, const float* rstd, const float* weight,
                     const float* sum_g, const float* sum_gx, int32_t relu,
                     float* dx, float* dres, int32_t* dx_amax, void* stream);
int gcl_bn_bwd_apply_ld(const float* x, const float* dy, int32_t dy_ld, const float* y, const uint64_t* relu_mask, int64_t n,
                        int32_t c, const float* mean, const float* rstd, const float* weight, const float* sum_g,
                        const float* sum_gx, int32_t relu, float* dx, float* dres, int32_t* dx_amax, void* stream);

/* Round 5: plane images from the producer.  A tensor at least 128 channels wide is consumed by the fp16x3 convolutions as a
 * plane image (gcl_split_planes); the BatchNorm passes that write such a tensor write the image in the same pass, which
 * needs the tensor's power-of-two scale BEFORE the pass -- i.e. max|tensor| bounded from what is known by then:
 *   forward   gcl_bn_stats_from_tiles_range: the convolution epilogue's partials carry per-column minimum / maximum
 *             (float[4][c][n_tiles]: sum, squares, min, max); y = (x - mean) rstd w + b is monotone in x, so max|y| of a
 *             channel is attained at its minimum or maximum of x: EXACT without residual; + max|residual| (add_amax: its
 *             slot) with one; max'ed with `max_with` (the slot of the other input of an ME.cat written in place).  The
 *             bound goes to the zeroed slot y_amax, the channel ranges to xrange[2][c] (kept for the backward pass).
 *             gcl_bn_apply_planes: as gcl_bn_apply_ld; with planes != NULL it also writes the image (row pitch = y's) at
 *             the scale of y_amax's value and publishes nothing.
 *   backward  gcl_bn_bwd_reduce_range: k_bn_reduce also keeps max|g| per channel; with it, xrange and the sums,
 *             |dx| <= |w rstd| (max|g| + |sum_g| / n + max|xhat| |sum_gx| / n) goes to the zeroed slot dx_amax;
 *             gcl_bn_bwd_apply_planes writes dx's image at that scale.
 * A bound above the true maximum costs range at the bottom of the lo plane (which the range-extended format has to spare),
 * never correctness; consumers derive their scale from the same slot. */
int gcl_bn_stats_from_tiles_range(const float* partial, int64_t n_tiles, int64_t n, int32_t c, float eps, float momentum,
                                  float* running_mean, float* running_var, float* mean, float* rstd, float* xrange,
                                  const float* weight, const float* bias, int32_t relu, const int32_t* add_amax,
                                  const int32_t* max_with, int32_t* y_amax, void* stream);
int gcl_bn_apply_planes(const float* x, int64_t n, int32_t c, const float* mean, const float* rstd, const float* weight,
                        const float* bias, const float* residual, int32_t relu, float* y, int32_t y_ld, uint64_t* relu_mask,
                        int32_t* y_amax, void* planes, void* stream);
int gcl_bn_bwd_reduce_range(const float* x, const float* dy, int32_t dy_ld, const float* y, const uint64_t* relu_mask,
                            int64_t n, int32_t c, const float* mean, const float* rstd, int32_t relu, double* scratch,
                            float* sum_g, float* sum_gx, const float* xrange, const float* weight, int32_t* dx_amax,
                            void* stream);
int gcl_bn_bwd_apply_planes(const float* x, const float* dy, int32_t dy_ld, const float* y, const uint64_t* relu_mask, int64_t n,
                            int32_t c, const float* mean, const float* rstd, const float* weight, const float* sum_g,
                            const float* sum_gx, int32_t relu, float* dx, float* dres, int32_t* dx_amax, void* planes,
                            void* stream);

/* Row-wise L2 normalisation of the output features, y = x / ||x||_2 (model/resunet.py:226-230; no epsilon, as there).
 * norm[n] keeps the row norms for the backward pass: dx = (dy - y (y . dy)) / norm.  c: power of two in [4, 256].
 * dx_amax (optional, ZERO-INITIALISED amax slot): receives gcl_amax of dx (the next consumer is the `final` convolution's
 * input gradient, which needs it in the fp16x3 arithmetic). */
int gcl_row_normalize_fwd(const float* x, int64_t n, int32_t c, float* y, float* norm, void* stream);
int gcl_row_normalize_bwd(const float* y, const float* dy, const float* norm, int64_t n, int32_t c, float* dx,
                          int32_t* dx_amax, void* stream);

/* torch.optim.SGD's step (lib/colocation_trainer.py:73-77, :887: lr, momentum, weight_decay; dampening 0, no Nesterov)
 * for a list of tensors in ONE launch:  d = g + wd p;  buf = first ? d : momentum buf + d;  p -= lr buf.
 * table: DEVICE array of n_tensors x {float* p, const float* g, float* buf}; sizes: DEVICE int64[n_tensors]. */
int gcl_sgd_multi(const void* table, const int64_t* sizes, int32_t n_tensors, float lr, float momentum,
                  float weight_decay, int32_t first, void* stream);

/* ------------------------------------------------------------------------------------------------
 * GCL loss (lib/colocation_trainer.py:430-535) and feature-space 1-NN.
 *   group g = rows index[goff[g] .. goff[g+1]) of F [n, c] (c <= 64); sel[s] = selected group ids.
 *   flags == 0 (the configuration of scripts/train_gcl_kitti.sh):
 *     pos[s]  = relu(mean_j |mean - f_j|^2 - pos_thresh)          (:474)
 *     fin[s]  = relu(|mean - f_finest|^2 - finest_thresh)         (:484-485)
 *   flags (the other config switches of the same function, and location_contrastive_loss :768-776):
 *     GCL_LOSS_SQRT   square_loss == False: distances enter as sqrt(d2 + 1e-7)                      (:470,:476,:487)
 *     GCL_LOSS_BLOCK  block_finest_gradient: mean of the NON-finest members vs the detached finest (:479-481)
 *     GCL_LOSS_PAIR   use_pair_group_positive_loss: pairpos[2 s], pairpos[2 s + 1] = the two drawn member
 *                     positions inside group sel[s]                                                  (:466-470)
 *     GCL_LOSS_NOFIN  no finest term (location_contrastive_loss)
 *   backward adds  gpos * dpos[s]/dF + gfin * dfin[s]/dF  into dF with float atomics (dF pre-zeroed by caller).
 * ---------------------------------------------------------------------------------------------- */
#define GCL_LOSS_SQRT 1
#define GCL_LOSS_BLOCK 2
#define GCL_LOSS_PAIR 4
#define GCL_LOSS_NOFIN 8
int gcl_group_loss_fwd(const float* f, int32_t c, const int64_t* index, const int64_t* goff,
                       const uint8_t* finest_flag, const int64_t* sel, int32_t n_sel,
                       float pos_thresh, float finest_thresh, int32_t flags, const int32_t* pairpos,
                       float* pos, float* fin, void* stream);
int gcl_group_loss_bwd(const float* f, int32_t c, const int64_t* index, const int64_t* goff,
                       const uint8_t* finest_flag, const int64_t* sel, int32_t n_sel,
                       float pos_thresh, float finest_thresh, int32_t flags, const int32_t* pairpos,
                       const float* gpos, const float* gfin, float* df, void* stream);
/* location_circle_loss (lib/colocation_trainer.py:538-681), per-group part: with S = log_scale (16 in the reference)
 *   pos[s] = softplus(logsumexp_i(S v_i max(v_i, 0))) / S,  v_i = dist(mean, f_i) - pos_thresh / 2      (:607-618)
 *            (GCL_LOSS_PAIR: softplus(dist(f_a, f_b) - pos_thresh), :597-605)
 *   fin[s] = the same form over v_i = dist(f_i, f_finest) - finest_thresh (GCL_LOSS_BLOCK: non-finest members only,
 *            finest detached)                                                                          (:620-640)
 *   mean_out[s][c] = the group's mean feature (:585); the negative term (:642-676) is formed by the host on the
 *   [n_sel, n_sel] matrices of these means.  max(v, 0) is a constant in the backward pass, as in the reference.
 *   backward: dF += gpos dpos/dF + gfin dfin/dF + gmean[s] / n_s (gmean may be NULL), float atomics.
 *   flags: GCL_LOSS_SQRT | GCL_LOSS_BLOCK | GCL_LOSS_PAIR. */
int gcl_circle_group_fwd(const float* f, int32_t c, const int64_t* index, const int64_t* goff,
                         const uint8_t* finest_flag, const int64_t* sel, int32_t n_sel, float pos_thresh,
                         float finest_thresh, float log_scale, int32_t flags, const int32_t* pairpos, float* pos,
                         float* fin, float* mean_out, void* stream);
int gcl_circle_group_bwd(const float* f, int32_t c, const int64_t* index, const int64_t* goff,
                         const uint8_t* finest_flag, const int64_t* sel, int32_t n_sel, float pos_thresh,
                         float finest_thresh, float log_scale, int32_t flags, const int32_t* pairpos,
                         const float* gpos, const float* gfin, const float* gmean, float* df, void* stream);

/* Row-wise nearest neighbour: for every row i of A[rows_a[i]] (rows_a may be NULL = identity) the column j
 * minimising sum_c (a - b)^2 over B[rows_b[j]]; ties -> lowest j.  dmin = that squared distance, or
 * sqrt(d2 + 1e-7) when l2 != 0 (lib/metrics.py:24-25).  The search runs as (64-row A tiles) x (chunks of B) workgroups
 * over a pair-interleaved copy of B[rows_b] made in scratch, followed by a merge over the chunks;
 * scratch: int32[gcl_nn_rowmin_scratch_len(ma, mb)], always required. */
int64_t gcl_nn_rowmin_scratch_len(int32_t ma, int32_t mb);
int gcl_nn_rowmin(const float* a, const int64_t* rows_a, int32_t ma, const float* b, const int64_t* rows_b,
                  int32_t mb, int32_t c, int32_t l2, int32_t* scratch, float* dmin, int32_t* argmin, void* stream);

/* keep[r] = (sel1[r] != sel2[arg[r]]) && the pair {sel1[r], sel2[arg[r]]} shares no positive group
 * (equivalent to the reference's `~np.isin(_neg_hash(...), index_hash)` :521-529: the symmetric key is
 * collision-free).  table: int64[cap,2] scratch hash table (cap = pow2 >= 2*m). */
int gcl_neg_mask(const int64_t* sel1, const int64_t* sel2, const int32_t* arg, int32_t m,
                 const int64_t* index, const int64_t* goff, int64_t n_groups, int64_t n_index,
                 int64_t* table, int64_t cap, uint8_t* keep, void* stream);
/* neg = mean over kept rows of relu(thresh - dmin)^2 (NaN when nothing is kept, as torch's mean of empty);
 * out[0] = neg, out[1] = #kept.  Backward scatters into dF (atomics). */
int gcl_neg_loss_fwd(const float* dmin, const uint8_t* keep, int32_t m, float thresh, float* out, void* stream);
int gcl_neg_loss_bwd(const float* f, int32_t c, const int64_t* sel1, const int64_t* sel2, const int32_t* arg,
                     const float* dmin, const uint8_t* keep, int32_t m, float thresh, const float* out,
                     const float* gneg, float* df, void* stream);
/* The step's scalar arithmetic in one launch each way (lib/colocation_trainer.py:533-535, :865-868):
 * out = {total, pos_mean, fin_mean, neg} with pos_mean = sum(pos) / n_sel, fin_mean = sum(fin) / n_sel,
 * total = w_pos * pos_mean + w_fin * fin_mean + w_neg * neg[0]; gcl_loss_seed writes the upstream gradients of the
 * terms for gcl_group_loss_bwd / gcl_neg_loss_bwd from the gradient of the total: gpos[i] = gfin-alike
 * (g * w) / n_sel, gneg[0] = g * w_neg. */
int gcl_loss_combine(const float* pos, const float* fin, int32_t n_sel, const float* neg, float w_pos, float w_fin,
                     float w_neg, float* out, void* stream);
int gcl_loss_seed(const float* g_total, float w_pos, float w_fin, float w_neg, int32_t n_sel, float* gpos, float* gfin,
                  float* gneg, void* stream);

/* ------------------------------------------------------------------------------------------------
 * SC2-PCR registration back-end (scripts/SC2_PCR/SC2_PCR.py, Matcher.SC2_PCR :304-381) for ONE pair:
 * src / tgt float [n, 3] = the putative correspondences (n <= 8192).  No [n, n] matrix is materialised; ties of the
 * reference's argsort / argmax go to the lowest index.  Call order (the host keeps the few glue steps):
 *   gcl_sc2_confidence  leading eigenvector of the first-order compatibility matrix by power iteration with the
 *                       reference's torch.allclose early stop (:337-345, :167-185).  x float[n] must hold ones,
 *                       done int32[1] zero; partial float[gcl_sc2_chunks() * n].  Result in x.
 *   gcl_sc2_local_max   non-maximum suppression flags (:43-47); is_max int32[n] must hold ones.
 *                       host: seeds = first int(n * ratio) of a stable descending sort of conf * is_max.
 *   gcl_sc2_seed_knn    tight-compatibility bit matrix (bits uint64[n * ceil(n / 64)]), second-order measure of every
 *                       seed and its k1 (<= 32) best correspondences, knn int32[n_seeds * k1] (:353-361, :85-86).
 *   gcl_sc2_seed_trans  per seed: k2-subset by the local second-order measure, k2 x k2 power iteration
 *                       (num_iterations steps), weighted Kabsch -> trans float[n_seeds * 12] (rows of [R | t]);
 *                       fitness float[n_seeds] = inlier count under inlier_thresh (:88-161).
 *                       host: best = lowest index of the maximum fitness.
 *   gcl_sc2_refine      post refinement in place on T float[12] (:238-279): up to `iterations` weighted Kabsch steps
 *                       over the inliers under thr, stopping when the inlier count repeats.
 *                       partial double[gcl_sc2_refine_partial_len()], state int32[2].
 * ---------------------------------------------------------------------------------------------- */
int32_t gcl_sc2_chunks(void);
int32_t gcl_sc2_refine_partial_len(void);
int gcl_sc2_confidence(const float* src, const float* tgt, int32_t n, float d_thre, int32_t num_iterations,
                       float* partial, float* x, int32_t* done, void* stream);
/* the same power iteration over the NON-ZERO entries of the compatibility matrix, kept from ONE build pass per registration
 * in an ELL layout (every (row, column chunk) segment has its own place of chunk-length entries; scratch:
 * gcl_sc2_confidence_scratch_bytes(n) bytes = n^2 entries of address space, 512 MB at n = 8000, of which only the non-zero
 * ones are touched): the same non-zero terms in the same order, i.e. bitwise the result x of gcl_sc2_confidence, without
 * re-deriving 64 M entries (two square roots each) in every one of the 20 products.  `partial` is working space here (the
 * products alternate between it and a second buffer in scratch: product k normalises product k - 1 itself, 21 launches
 * instead of 40; GCL_SC2_FOLDED_NORMALIZE=0 restores one normalisation launch per product) */
int64_t gcl_sc2_confidence_scratch_bytes(int32_t n);
int gcl_sc2_confidence_sparse(const float* src, const float* tgt, int32_t n, float d_thre, int32_t num_iterations,
                              float* partial, float* x, int32_t* done, void* scratch, void* stream);
/* One call per registration (round 5): Matcher.SC2_PCR + the labels of Matcher.estimator (:304-381, :404-409) as ONE launch
 * sequence -- gcl_sc2_confidence_sparse, gcl_sc2_local_max, the seed order (stable argsort of -(conf * is_max): value
 * descending, index ascending), gcl_sc2_seed_knn, gcl_sc2_seed_trans, the best seed (lowest index of the maximum fitness),
 * gcl_sc2_refine, and the [4, 4] result with labels[i] = |R s_i + t - t'_i| < inlier_thresh -- without the host in between
 * (the Python Matcher issued ~25 torch operations and 7 calls per pair: 1 ms of host time in a loop that is host-bound).
 * Outputs: conf float[n], seeds int64[n_seeds], knn int32[n_seeds * k1], seed_trans float[n_seeds * 12],
 * fitness float[n_seeds], best int32[1], trans16 float[16] (row-major [4, 4]), labels float[n] (0 / 1).
 * scratch: gcl_sc2_register_scratch_bytes(n) bytes.  FOOTPRINT: ~ 8 n^2 bytes of ADDRESS SPACE -- 528 MB at n = 8000 -- for the
 * confidence's kept non-zero entries (an ELL slab: n^2 (column, value) places of 8 bytes, of which only the non-zero entries
 * are ever touched, i.e. only those cost bandwidth or physical pages' worth of traffic) + the 8 MB tight-compatibility bit
 * matrix; the seed stage's uint16 second-order rows (2 n_seeds n bytes) reuse the slab once the last product has run.  That
 * is two of the four dense [n, n] float matrices the reference allocates (scripts/SC2_PCR/SC2_PCR.py:327-361); with several
 * registrations in flight (GCL_EVAL_STREAMS > 1) every one needs its own block.  Same results as the staged calls. */
int64_t gcl_sc2_register_scratch_bytes(int32_t n);
int gcl_sc2_register(const float* src, const float* tgt, int32_t n, float d_thre, int32_t num_iterations, float nms_radius,
                     int32_t n_seeds, int32_t k1, int32_t k2, float inlier_thresh, float refine_thr, int32_t refine_iters,
                     void* scratch, float* conf, int64_t* seeds, int32_t* knn, float* seed_trans, float* fitness,
                     int32_t* best, float* trans16, float* labels, void* stream);
int gcl_sc2_local_max(const float* src, const float* conf, int32_t n, float radius, int32_t* is_max, void* stream);
int gcl_sc2_seed_knn(const float* src, const float* tgt, int32_t n, const int64_t* seeds, int32_t n_seeds,
                     float d_thre, int32_t k1, uint64_t* bits, int32_t* knn, void* stream);
int gcl_sc2_seed_trans(const float* src, const float* tgt, int32_t n, const int32_t* knn, int32_t n_seeds, int32_t k1,
                       int32_t k2, float d_thre, int32_t num_iterations, float inlier_thresh, float* trans,
                       float* fitness, void* stream);
int gcl_sc2_refine(const float* src, const float* tgt, int32_t n, float thr, int32_t iterations, double* partial,
                   int32_t* state, float* T, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Native step runtime (round 3): ONE call enqueues a whole pass.
 *
 * The per-operator entries above are what ME's operator surface binds; a training step calls ~560 of them, and from
 * Python each costs ~40 us of interpreter + ctypes time -- the step was bound by the host.  The two families below issue
 * the same launches, with the same arguments and in the same order, from C:
 *   gcl_maps_*  everything CoordinateManager builds for a network (lib/colocation_trainer.py:843-845 -> ME's
 *               coordinate manager: stride maps, kernel maps, mask-sorted tables, pair lists) in one call;
 *   gcl_plan_*  the forward / backward pass of a network described as a list of operator records (what
 *               model/resunet.py:173-232 and model/residual_block.py:37-53 call on the ME surface).
 * Results are bitwise identical to the per-operator path (tests/test_gpu_plan.py).
 * Memory stays with the caller: both families carve every tensor they need from an ARENA the caller passes (a device
 * buffer; size from gcl_maps_arena_bytes / gcl_plan_arena_bytes); plan handles own host memory only.
 * ---------------------------------------------------------------------------------------------- */
#define GCL_MAX_LEVELS 8
#define GCL_MAX_MAPS 16
#define GCL_ERR_ARENA (-4)     /* the arena is too small */
#define GCL_MAPS_PINNED_BYTES (512 * (GCL_MAX_MAPS + 1))

/* one kernel map to build: CoordinateManager.get_kernel_map(t_in, kernel_size, stride);
 * tables bit 0 / bit 1: gcl_table_sort of nbr / of nbr_t; bit 2: presence words of nbr (the Cin <= 4 first layer's
 * occupancy path); pairs != 0: the weight gradient's pair lists.
 * kernel_size 1 (stride 1): only the identity pair list of level t_in (pairs != 0). */
typedef struct gcl_map_spec {
  int32_t t_in, kernel_size, stride, tables, pairs;
} gcl_map_spec;

/* one kernel map as built; pointers are DEVICE pointers into the arena (NULL = not requested), seg_off HOST values */
typedef struct gcl_map_desc {
  int32_t t_in, kernel_size, stride, K;
  int32_t level_in, level_out;          /* indices into gcl_maps_desc.n_rows (level i = tensor stride 2^i) */
  int64_t n_in, n_out, n_pairs;
  int32_t *nbr, *nbr_t, *counts;
  int32_t *tbl_n, *order_n, *mask_n;    /* gcl_table_sort(nbr) */
  int32_t *tbl_t, *order_t, *mask_t;    /* gcl_table_sort(nbr_t) */
  int32_t *pair_in, *pair_out;
  uint32_t* presence;                   /* gcl_presence_bits(nbr) when the spec asks for it (tables bit 2), else NULL */
  int32_t* dw_bounds;                   /* gcl_conv_bwd_weight_bounds over pair_out (maps with pair lists and >= 32768 out rows), else NULL */
  int64_t seg_off[128];                 /* padded prefix sums of the per-offset pair counts, K + 1 used */
  int32_t counts_host[128];
} gcl_map_desc;

typedef struct gcl_maps_desc {
  int32_t n_levels, n_maps;
  int64_t n_rows[GCL_MAX_LEVELS];
  int32_t* coords[GCL_MAX_LEVELS];      /* int32 [n_rows, 4] */
  int64_t* table[GCL_MAX_LEVELS];       /* coordinate hash table of the level, int64 [cap, 2] */
  int64_t cap[GCL_MAX_LEVELS];
  int32_t status[4];                    /* HOST copy of gcl_coords_insert's status words */
  int64_t arena_used;
  void* ready_event;                    /* split build: hipEvent_t the CALLER records on the side stream after the call (else NULL) */
  int32_t late_mask;                    /* split build: bit s = map s was enqueued on the side stream (gcl_maps_build_split) */
  int32_t reserved;
  gcl_map_desc maps[GCL_MAX_MAPS];
} gcl_maps_desc;

/* Upper bound of the arena gcl_maps_build needs for n stride-1 rows. */
int64_t gcl_maps_arena_bytes(int64_t n, const gcl_map_spec* specs_host, int32_t n_specs, int32_t n_levels);
/* Builds levels 0 .. n_levels-1 (tensor strides 1, 2, 4, ...) from `coords` (int32 [n,4], device) and every map of
 * `specs_host`, with exactly the launches of the per-operator entries.  SYNCHRONISES `stream` twice (level sizes, pair
 * counts) -- call it from a loader-side thread on a side stream.  pinned_host: >= GCL_MAPS_PINNED_BYTES of page-locked HOST
 * memory used for the two read-backs.  Fills *out_host; returns GCL_ERR_ARG for duplicate / out-of-range coordinates
 * (out_host->status tells which). */
int gcl_maps_build(const int32_t* coords, int64_t n, const gcl_map_spec* specs_host, int32_t n_specs, int32_t n_levels,
                   void* arena, int64_t arena_bytes, void* pinned_host, gcl_maps_desc* out_host, void* stream);
/* The same build on TWO streams, for a pass over few rows (one pair of clouds per call: util/misc.py:128-130,
 * scripts/test_kitti.py:143-150), whose ~ 90 small dependent launches are a third of the pass.  Everything up to the read-back
 * of the level sizes runs on `stream`; after it -- when no map asks for pair lists, i.e. nothing else waits for the host --
 * the maps of the input level alone (kernel maps with t_in = 1, stride 1: what a network's first layers use) and their sorted
 * tables stay on `stream`, every other map goes to `side_stream` (out_host->late_mask).  The caller records an event on
 * side_stream after the call and stores it in out_host->ready_event: gcl_plan_forward / _eval make `stream` wait for it in
 * front of the first record that uses a late map, so the first layers' convolutions run beside the deeper levels' map
 * building.  Same launches, same results as gcl_maps_build (side_stream NULL, or pair lists wanted: exactly that call). */
int gcl_maps_build_split(const int32_t* coords, int64_t n, const gcl_map_spec* specs_host, int32_t n_specs, int32_t n_levels,
                         void* arena, int64_t arena_bytes, void* pinned_host, gcl_maps_desc* out_host, void* stream,
                         void* side_stream);

/* Operator records of a network pass.  Tensors are numbered 0 .. n_tensors-1 (0 = the input features); every record
 * names its input(s) and its output; parameters are numbered in the order of the `params` / `grads` pointer arrays. */
#define GCL_OP_CONVBN 1    /* y = BatchNorm(conv(x)) (+ residual x2) (relu): MinkowskiConvolution(+Transpose) + MinkowskiBatchNorm */
#define GCL_OP_CONV 2      /* y = conv(x) (+ bias) */
#define GCL_OP_RELU 3      /* MEF.relu */
#define GCL_OP_CAT 4       /* ME.cat(x, x2) */
#define GCL_OP_ROWNORM 5   /* y = x / ||x||_2 per row (model/resunet.py:226-230) */
typedef struct gcl_plan_op {
  int32_t kind;
  int32_t x, x2, y;          /* tensor ids; x2 = residual (CONVBN) / second input (CAT), -1 = none */
  int32_t level_in, level_out, cin, cout;
  int32_t map, transpose, K; /* map = index into gcl_maps_desc.maps (kernel_size 1: the identity-pair entry) */
  int32_t w, bias;           /* parameter ids: kernel [K, cin, cout], bias [1, cout] or -1 */
  int32_t bn_w, bn_b, bn;    /* parameter ids of the BatchNorm affine pair; bn = index of its running-statistics pair */
  int32_t relu;
  float momentum, eps;
} gcl_plan_op;

/* Plan handle: host memory only.  weight_order[n_weights] = parameter ids of the MFMA-shaped convolution kernels in
 * the order their max-abs slots / packed images are laid out (WeightAmaxGroup).  NULL on error (gcl_last_error). */
void* gcl_plan_create(const gcl_plan_op* ops_host, int32_t n_ops, int32_t n_tensors, int32_t n_params,
                      const int32_t* weight_order_host, int32_t n_weights, int32_t presplit_min_c);
void gcl_plan_destroy(void* plan);
/* persistent DEVICE state the caller keeps for a plan (pointer / descriptor tables), bytes */
int64_t gcl_plan_state_bytes(const void* plan);
/* arena bytes of one forward + backward pass over `maps` */
int64_t gcl_plan_arena_bytes(void* plan, const gcl_maps_desc* maps_host);
/* Forward pass (training mode: batch statistics, running statistics updated).  x: features of tensor 0
 * [n_rows[level_in of op 0], cin]; params_host[n_params]: DEVICE pointers of the parameters; bn_stats_host[2 * n_bn]:
 * DEVICE pointers running_mean, running_var per BatchNorm.  *y_out_host receives the device pointer (inside the arena)
 * of the last record's output.  The arena must stay untouched until gcl_plan_backward has run. */
int gcl_plan_forward(void* plan, const gcl_maps_desc* maps_host, const float* x, void* const* params_host,
                     void* const* bn_stats_host, void* state, void* arena, int64_t arena_bytes, float** y_out_host,
                     void* stream);
/* Inference pass of the same records (model.eval(), torch.no_grad(): util/misc.py:58-130, scripts/test_kitti.py:141-152):
 * every GCL_OP_CONVBN is ONE gcl_conv_fwd_fused launch -- BatchNorm with running statistics folded into the epilogue as
 * y = relu?(conv * scale + shift (+ residual)) -- the Cin <= 4 first layer is gcl_stem_fwd + gcl_bn_apply(mean, rstd).
 * bn_eval_host[4 * n_bn]: DEVICE pointers scale, shift, mean, rstd per BatchNorm (scale = gamma rsqrt(var + eps),
 * shift = beta - mean scale, rstd = rsqrt(var + eps)).  The packed kernels and their max-abs slots persist in `state`
 * (gcl_plan_eval_state_bytes) from pass to pass: repack != 0 re-measures and re-packs them (first pass, or after the
 * parameters changed).  No backward pass follows; the arena may be released once the output has been consumed. */
int64_t gcl_plan_eval_state_bytes(const void* plan);
int64_t gcl_plan_eval_arena_bytes(void* plan, const gcl_maps_desc* maps_host);
int gcl_plan_forward_eval(void* plan, const gcl_maps_desc* maps_host, const float* x, void* const* params_host,
                          void* const* bn_eval_host, int32_t repack, void* state, void* arena, int64_t arena_bytes,
                          float** y_out_host, void* stream);
/* Backward pass of the forward pass that ran in `arena` (several forward passes of one plan may be outstanding, each in
 * its own arena): the records first_op <= i < last_op in reverse order (the caller may cut the pass into segments,
 * highest records first, e.g. to start a gradient bucket's all-reduce in between).  dy: gradient of the forward output
 * (read by the segment that contains the last record).  grads_host[n_params]: DEVICE pointers that RECEIVE (are
 * overwritten with) the parameter gradients. */
int gcl_plan_backward(void* plan, void* arena, const float* dy, void* const* grads_host, int32_t first_op,
                      int32_t last_op, void* stream);
/* Forget the forward pass parked under `arena` (its output was dropped without a backward pass). */
int gcl_plan_release(void* plan, void* arena);
/* Optional second stream: the weight gradients (needed only by the optimizer) are enqueued there, ordered behind their
 * operands by events, while `stream` continues with the input-gradient chain; every gcl_plan_backward call ends with
 * `stream` waiting for them.  NULL (default) = everything on one stream.  Results do not depend on it. */
int gcl_plan_set_aux_stream(void* plan, void* stream);
/* a stream whose kernels run on the lowest `percent` % of the device's CUs (hipExtStreamCreateWithCUMask) -- measurement hook
 * for the weight-gradient stream (GCL_AUX_CU_PCT); gcl_stream_destroy frees it */
int gcl_stream_create_cu_share(int32_t percent, int32_t low_priority, void** stream_out);
int gcl_stream_destroy(void* stream);
/* Per-launch timing of the convolution launches of the NEXT forward + backward pass (events on `stream`):
 * gcl_plan_profile(plan, 1) arms it; after the stream has been synchronised gcl_plan_profile_read copies up to
 * max_records records of 8 doubles {kind (0 fwd/dx, 1 dW), ms, pairs, cin, cout, n_in, n_out, K | flags} and returns
 * their number. */
int gcl_plan_profile(void* plan, int32_t enable);
int gcl_plan_profile_read(void* plan, double* records_host, int32_t max_records);

/* Elementwise helpers of the plan path (also used by the per-operator path so that both stay bitwise equal):
 * gcl_col_sum: out[c] = sum over rows of x [n, c] (fp64 partials, ordered) -- the bias gradient of `final`
 * (model/resunet.py:165-171); scratch: double[gcl_bn_scratch_len(n, c)]. */
int gcl_col_sum(const float* x, int64_t n, int32_t c, double* scratch, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GCL_AMD_H */
