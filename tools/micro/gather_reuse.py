"""CPU analysis (numpy, no GPU): how often does a 3^3 stride-1 convolution launch gather the same input row inside ONE tile?
Mask-sorted tiles (the product's row order: rows with the same set of present offsets share a tile, so the MFMA steps of a tile
are dense) against spatially compact tiles (Morton order): distinct gathered rows / gathers per 32-row MFMA tile and per 128-row
workgroup.  The ratio bounds what staging a tile's distinct rows ONCE in LDS could save of the L2 -> CU gather traffic that
bounds the C <= 64 layers.  Usage: python tools/micro/gather_reuse.py [seed]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from gcl_amd import synthetic

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 100
b = synthetic.make_train_batch(seed, batch_size=1)          # one sample: a centre cloud + 6 neighbour clouds
C = np.asarray(b["sinput_C"]).astype(np.int64)
n = len(C)
key = ((C[:, 0] << 48) | ((C[:, 1] + 32768) << 32) | ((C[:, 2] + 32768) << 16) | (C[:, 3] + 32768))
srt = np.argsort(key)
ks = key[srt]
nbr = np.full((27, n), -1, np.int64)
k = 0
for dz in (-1, 0, 1):
    for dy in (-1, 0, 1):
        for dx in (-1, 0, 1):
            q = ((C[:, 0] << 48) | ((C[:, 1] + dx + 32768) << 32) | ((C[:, 2] + dy + 32768) << 16) | (C[:, 3] + dz + 32768))
            pos = np.searchsorted(ks, q)
            pos[pos >= n] = 0
            hit = ks[pos] == q
            nbr[k, hit] = srt[pos[hit]]
            k += 1
present = nbr >= 0
print(f"{n} voxels, {present.sum()} pairs ({present.sum() / n:.1f} per row)")


def morton(C):
    def spread(v):
        v = v.astype(np.uint64) & 0x1FFFFF
        v = (v | (v << 32)) & 0x1F00000000FFFF
        v = (v | (v << 16)) & 0x1F0000FF0000FF
        v = (v | (v << 8)) & 0x100F00F00F00F00F
        v = (v | (v << 4)) & 0x10C30C30C30C30C3
        v = (v | (v << 2)) & 0x1249249249249249
        return v
    m = spread(C[:, 1] + 32768) | (spread(C[:, 2] + 32768) << 1) | (spread(C[:, 3] + 32768) << 2)
    return np.lexsort((m, C[:, 0]))


freq = present.sum(1)
rank = np.argsort(np.argsort(freq))            # rarest offset -> most significant bit, as gcl_table_sort does
mask = np.zeros(n, np.int64)
for kk in range(27):
    mask |= present[kk].astype(np.int64) << int(rank[kk])
orders = {"mask-sorted (the product's tiles)": np.argsort(mask, kind="stable"), "Morton order (compact tiles)": morton(C),
          "input order": np.arange(n)}
for name, order in orders.items():
    for T in (32, 128):
        g = d = steps = 0
        for t0 in range(0, n, T):
            rows = order[t0:t0 + T]
            sub = nbr[:, rows]
            v = sub[sub >= 0]
            g += len(v)
            d += len(np.unique(v))
            steps += int((sub >= 0).any(1).sum())
        print(f"{name:36s} tiles of {T:3d}: {g / max(d, 1):5.2f} gathers per distinct row in the tile "
              f"(distinct / gathers = {d / g:.3f}); offsets with any row per tile {steps / (n / T):5.1f} of 27")
