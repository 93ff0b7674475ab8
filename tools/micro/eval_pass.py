"""DIAGNOSTIC (GPU box): one inference pass (a pair of clouds per forward, configs[1] at one pair per pass), repeated;
run it under `rocprofv3 --kernel-trace --stats` to see what the pass is made of.  Prints the wall time per pass."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from gcl_amd import synthetic
from gcl_amd.model import load_model
from gcl_amd.scripts.test_kitti import forward_clouds
dev = torch.device("cuda:0")
torch.manual_seed(0); np.random.seed(0)
model = load_model("ResUNetBN2C")(1, 32, bn_momentum=0.05, normalize_feature=True, conv1_kernel_size=5, D=3).to(dev).eval()
PP = int(os.environ.get("EP_PAIRS", "1"))        # pairs per pass (EP_PAIRS=8: the eval loop's batch_pairs = 8 shape)
pairs = [synthetic.make_eval_pair(100 + s, baseline=15.0 + 5.0 * (s % 6)) for s in range(4 * PP)]
d = [{k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in p.items()} for p in pairs]
N = int(os.environ.get("EP_PASSES", "40"))
with torch.no_grad(), torch.cuda.device(dev):
    def one(j):
        forward_clouds(model, [(d[(j % 4) * PP + q][f"sinput{k}_F"], d[(j % 4) * PP + q][f"sinput{k}_C"]) for q in range(PP) for k in (0, 1)])
    for j in range(4):
        one(j)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for j in range(N):
        one(j)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / N
nv = np.mean([len(p["sinput0_C"]) + len(p["sinput1_C"]) for p in pairs]) * PP
print(f"{dt * 1e3:.3f} ms per pass of {nv:.0f} voxels = {nv / dt / 1e6:.1f} M voxels/s")

if os.environ.get("EP_HOST_SPLIT"):     # where the enqueuing thread spends a pass (perf_counter around the two native calls)
    import gcl_amd.MinkowskiEngine as ME
    from gcl_amd.MinkowskiEngine import native
    acc = {"build": 0.0, "eval": 0.0}
    ob, oe = ME.CoordinateManager.build_native.__func__, native.NetworkPlan.run_eval
    def tb(cls, *a, **k):
        t = time.perf_counter(); r = ob(cls, *a, **k); acc["build"] += time.perf_counter() - t; return r
    def te(self, *a, **k):
        t = time.perf_counter(); r = oe(self, *a, **k); acc["eval"] += time.perf_counter() - t; return r
    ME.CoordinateManager.build_native = classmethod(tb)
    native.NetworkPlan.run_eval = te
    from gcl_amd import _lib
    lib = _lib.require_gpu()
    acc["c_eval"] = acc["c_build"] = acc["c_bytes"] = 0.0
    for nm_, key_ in (("gcl_plan_forward_eval", "c_eval"), ("gcl_maps_build", "c_build"), ("gcl_plan_eval_arena_bytes", "c_bytes")):
        fn = getattr(lib, nm_)
        def wrap(*a, _fn=fn, _k=key_):
            t = time.perf_counter(); r = _fn(*a); acc[_k] += time.perf_counter() - t; return r
        setattr(lib, nm_, wrap)
    with torch.no_grad(), torch.cuda.device(dev):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for j in range(N):
            one(j)
        t_enq = time.perf_counter() - t0
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
    print(f"host per pass: build_native {acc['build'] / N * 1e6:.0f} us (incl. its read-back wait), run_eval {acc['eval'] / N * 1e6:.0f} us, "
          f"everything else {(t_enq - acc['build'] - acc['eval']) / N * 1e6:.0f} us; loop {t_enq / N * 1e6:.0f} us, with the final sync {t_all / N * 1e6:.0f} us")
    print(f"  inside: gcl_maps_build {acc['c_build'] / N * 1e6:.0f} us, gcl_plan_eval_arena_bytes {acc['c_bytes'] / N * 1e6:.0f} us, "
          f"gcl_plan_forward_eval {acc['c_eval'] / N * 1e6:.0f} us -> Python around the build {(acc['build'] - acc['c_build']) / N * 1e6:.0f} us, "
          f"around the pass {(acc['eval'] - acc['c_eval'] - acc['c_bytes']) / N * 1e6:.0f} us")
