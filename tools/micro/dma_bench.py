"""DIAGNOSTIC (round 4): plane-image forward launches of the network's C >= 64 layer shapes on the benchmark batch with
register staging (k_conv_fwd_split) and with LDS-DMA staging (k_conv_fwd_dma, flag GCL_CONV_DMA), alternating in ONE process.
Usage on the GPU box:  python tools/micro/dma_bench.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import gcl_amd.MinkowskiEngine as ME  # noqa: E402
from gcl_amd import _lib, synthetic  # noqa: E402

batch = synthetic.make_train_batch(100, batch_size=int(os.environ.get("LB_BATCH", "4")), group_mode="fixed16")
dev = "cuda:0"
C = batch["sinput_C"].to(dev)
# (tensor stride of the kernel map's fine side, stride, transposed, cin, cout)
# last field: operand = plane image (what production feeds the C >= 128 layers) or fp32 rows
LAYERS = [(4, 1, False, 128, 128, 1), (8, 1, False, 256, 256, 1), (4, 2, False, 128, 256, 1), (4, 2, True, 256, 128, 1),
          (2, 2, True, 256, 64, 1), (1, 2, True, 128, 64, 1), (2, 1, False, 64, 64, 0), (1, 1, False, 64, 64, 0),
          (1, 1, False, 32, 32, 0), (1, 2, False, 32, 64, 0), (2, 2, False, 64, 128, 0), (1, 2, True, 64, 32, 0)]
rounds, reps = int(os.environ.get("LB_ROUNDS", "5")), int(os.environ.get("LB_REPS", "10"))
lib = _lib.load()
mgr = ME.CoordinateManager(C)
tot = {8: 0.0, 2: 0.0}
with torch.cuda.device(dev):
    for (t_in, stride, tr, cin, cout, pre) in LAYERS:
        km = mgr.get_kernel_map(t_in, 3, stride)
        tbl, order, mask = km.sorted_table(transposed=tr)
        n_out = tbl.shape[1]
        n_in = mgr.num_rows(t_in * stride if tr else t_in)
        K = 27
        g = torch.Generator().manual_seed(cin)
        x = torch.randn(n_in, cin, generator=g).to(dev)
        W = (0.1 * torch.randn(K, cin, cout, generator=g)).to(dev)
        xa, wa = ME.ops.amax_slot(x.device), ME.ops.amax_slot(x.device)
        _lib.check(lib.gcl_amax(_lib.ptr(x), x.numel(), _lib.ptr(xa), 1, _lib.stream()), "gcl_amax")
        _lib.check(lib.gcl_amax(_lib.ptr(W), W.numel(), _lib.ptr(wa), 1, _lib.stream()), "gcl_amax")
        planes = torch.empty((n_in, cin), dtype=torch.int32, device=dev)
        _lib.check(lib.gcl_split_planes(_lib.ptr(x), n_in, cin, _lib.ptr(xa), _lib.ptr(planes), _lib.stream()), "split")
        wp = torch.empty(lib.gcl_pack_weights_bytes(K, cin, cout, 4), dtype=torch.uint8, device=dev)
        _lib.check(lib.gcl_pack_weights(_lib.ptr(W), K, cin, cout, 0, 4, _lib.ptr(wa), _lib.ptr(wp), _lib.stream()), "pack")
        if os.environ.get("LB_HOT_ROWS"):      # diagnostic: every gather hits the same few rows (L1 / L2 hot) -- same
            hot = int(os.environ["LB_HOT_ROWS"])   # instruction stream, no fabric traffic for the A operand
            tbl = torch.where(tbl >= 0, tbl % hot, tbl).contiguous()
        y = {f: torch.empty((n_out, cout), device=dev) for f in (8, 2)}
        stats = torch.empty((4, cout, (n_out + 127) // 128), device=dev)

        def run(flags):
            _lib.check(lib.gcl_conv_fwd(_lib.ptr(planes if pre else x), n_in, pre, _lib.ptr(wp), 4, _lib.ptr(xa), _lib.ptr(wa), _lib.ptr(tbl),
                                        _lib.ptr(order), _lib.ptr(mask), n_out, K, cin, cout, None, _lib.ptr(y[flags]),
                                        _lib.ptr(stats), flags, _lib.stream()), "gcl_conv_fwd")
        times = {8: [], 2: []}
        for r in range(rounds):
            for f in (8, 2):
                run(f)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    run(f)
                e1.record()
                torch.cuda.synchronize()
                times[f].append(e0.elapsed_time(e1) / reps * 1e3)
        same = torch.equal(y[8], y[2])
        line = f"t={t_in} {cin:3d}->{cout:3d} s{stride}{' tr' if tr else '   '} n_out={n_out:7d} {'planes' if pre else 'rows  '} nb={lib.gcl_conv_fwd_nb(n_out, cout, 4)}:"
        for f, name in ((8, "regs"), (2, "dma ")):
            med = sorted(times[f])[len(times[f]) // 2]
            tot[f] += med
            line += f"  {name} {med:6.1f} ({min(times[f]):6.1f})"
        print(line + f"  bitwise_equal={same}", flush=True)
print(f"sum regs {tot[8]:.1f} us  dma {tot[2]:.1f} us")
