"""DIAGNOSTIC (GPU box): how much of the forward / input-gradient MFMA work is padding.  For every kernel map of the
training batch: pairs (real row x offset products), 32-row-tile x offset units the kernel visits (popcount of the tile
masks x 32), and what 16-row tiles over the same sorted order would visit."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import gcl_amd.MinkowskiEngine as ME
from gcl_amd import synthetic
from gcl_amd.model import load_model
batch = synthetic.make_train_batch(100, batch_size=4, group_mode="fixed16")
dev = "cuda:0"
C = batch["sinput_C"].to(dev)
m = load_model("ResUNetBN2C")(1, 32, bn_momentum=0.05, normalize_feature=True, conv1_kernel_size=5, D=3)
mgr = ME.CoordinateManager(C)
for t_in, ks, stride, tables, pairs in m.map_specs():
    km = mgr.get_kernel_map(t_in, ks, stride)
    if km.K > 27:
        continue
    for tr in tables:
        tbl, order, mask = km.sorted_table(transposed=tr)
        t = tbl.cpu().numpy() >= 0                       # [K, n] in sorted order
        K, n = t.shape
        real = int(t.sum())
        def units(rows):
            pad = (-n) % rows
            tt = np.concatenate([t, np.zeros((K, pad), bool)], 1).reshape(K, -1, rows)
            return int(tt.any(2).sum()) * rows
        u32, u16, u64 = units(32), units(16), units(64)
        pad = (-n) % 128
        wg = np.concatenate([t, np.zeros((K, pad), bool)], 1).reshape(K, -1, 128).any(2).sum(0)      # offsets per 128-row workgroup
        print(f"    workgroup (128 rows) offset unions: tiles {len(wg)} mean {wg.mean():.1f} p50 {np.percentile(wg, 50):.0f} "
              f"p90 {np.percentile(wg, 90):.0f} max {wg.max()}  sum {int(wg.sum())}")
        print(f"map ({t_in},{ks},{stride}) {'T' if tr else 'N'}: rows {n:7d} K {K:2d} pairs {real:9d}  "
              f"visited/pairs: 64-row {u64 / real:5.2f}  32-row {u32 / real:5.2f}  16-row {u16 / real:5.2f}   "
              f"pairs/row {real / n:5.2f}")
