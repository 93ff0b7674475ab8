"""Feature-space nearest neighbour + correspondence search (interfaces of lib/eval.py:18-48 and
scripts/test_kitti.py:29-43 / lib/colocation_trainer.py:381-395)."""
import numpy as np
import torch

from gcl_amd.lib.metrics import pdist_min


def host_to_device(array, device):
    """A small host array (numpy or CPU tensor) on ``device`` WITHOUT stopping the enqueuing thread: through a pinned block of
    torch's caching host allocator (reused only after the copy has run) and a non-blocking copy.  A pageable ``.to(device)``
    costs 80 - 90 us of host time per call here and the eval loop makes six per pair."""
    t = torch.from_numpy(array) if isinstance(array, np.ndarray) else array
    dev = torch.device(device)
    if dev.type != "cuda" or t.is_cuda:
        return t.to(dev)
    return t.pin_memory().to(dev, non_blocking=True)


def find_nn_gpu(F0, F1, nn_max_n=-1, return_distance=False, dist_type="SquareL2"):
    """1-NN of every F0 row in F1.  Returns CPU int64 indices (and CPU [N, 1] distances) like the reference.
    ``nn_max_n`` (the reference's chunk size against its [n, N, C] temporary) is accepted and irrelevant here:
    the HIP kernel keeps the running minimum in registers.  Ties resolve to the lowest index."""
    dmin, arg = pdist_min(F0, F1, dist_type)
    inds = arg.long().cpu()
    if return_distance:
        return inds, dmin.unsqueeze(1).cpu()
    return inds


class DeferredCorr:
    """``find_corr`` with the host half postponed: the random draws are made and the feature 1-NN is ENQUEUED when the
    object is built (same numpy call order as scripts/test_kitti.py:29-43), the nearest-neighbour indices stay on the
    device until ``resolve()`` -- so a loop over pairs need not stop the GPU once per pair to read 5000 indices it only
    needs for the optional distance statistics (scripts/test_kitti.py:155)."""

    def __init__(self, xyz0, xyz1, F0, F1, subsample_size=-1, nn_max_n=500):
        self.xyz0, self.xyz1 = xyz0, xyz1
        self.subsampled = subsample_size > 0 and len(F0) > subsample_size
        if self.subsampled:
            N0, N1 = min(len(F0), subsample_size), min(len(F1), subsample_size)
            self.inds0 = np.random.choice(len(F0), N0, replace=False)
            self.inds1 = np.random.choice(len(F1), N1, replace=False)
            F0, F1 = F0[host_to_device(self.inds0, F0.device)], F1[host_to_device(self.inds1, F1.device)]
        _, arg = pdist_min(F0, F1, "SquareL2")
        self.nn_dev = arg              # int32 on the device; ties -> lowest index

    def resolve(self):
        nn_inds = self.nn_dev.long().cpu()
        if self.subsampled:
            return self.xyz0[self.inds0], self.xyz1[self.inds1[nn_inds.numpy()]]
        return self.xyz0, self.xyz1[nn_inds]


def find_corr(xyz0, xyz1, F0, F1, subsample_size=-1, nn_max_n=500):
    """scripts/test_kitti.py:29-43: random subsample (np.random.choice, same call order), kNN, matched points."""
    return DeferredCorr(xyz0, xyz1, F0, F1, subsample_size, nn_max_n).resolve()


def forward_pair(model, F0, C0, F1, C1):
    """Features of the two clouds of an evaluation pair from ONE forward pass (scripts/test_kitti.py:141-152 runs the
    model twice).  In eval mode the network treats the clouds of a batch independently (running BatchNorm statistics,
    per-cloud coordinate maps), and every output row is accumulated in a fixed offset order whatever its tile mates
    are, so the result equals the two separate passes bit for bit -- at half the launches of this launch-bound case.
    ``C0`` / ``C1`` int32 [N, 4] with batch column 0."""
    import gcl_amd.MinkowskiEngine as ME
    if model.training:
        raise RuntimeError("forward_pair needs model.eval(): batch statistics would mix the two clouds")
    C1b = C1.clone()
    C1b[:, 0] = 1
    out = model(ME.SparseTensor(torch.cat([F0, F1]), coordinates=torch.cat([C0, C1b]))).F
    return out[:len(F0)], out[len(F0):]
