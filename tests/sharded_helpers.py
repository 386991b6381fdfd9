"""Test doubles for the sharded voxelizer: CPU compute steps built on the oracle (so that the distributed
orchestration can run under gloo without a GPU) and an in-process thread communicator (K virtual ranks on one GPU)."""
import threading

import numpy as np
import torch

import oracle


class NumpyOps:
    """same interface as d3d_amd.voxel.sharded.HipOps, computed with the CPU oracle + numpy"""

    def voxelize_reduce(self, points, shape, bounds, reduction, index_offset):
        pts = points.cpu().numpy()
        n, c = pts.shape
        red = {1: 1, 2: 2, 3: 3, 4: 1}[int(reduction)]
        r = oracle.voxelize_3d_dense(pts, shape, bounds, 1, max(n, 1), red)
        coords, cnt = r["coords"], r["voxel_npoints"]
        agg = r["aggregates"].astype(np.float32)
        if int(reduction) == 4:
            agg = (agg.astype(np.float64) * cnt[:, None]).astype(np.float32)
        b = np.asarray(bounds, np.float32)
        size = ((b[1::2] - b[0::2]) / np.asarray(shape, np.float32)).astype(np.float32)
        with np.errstate(invalid="ignore", over="ignore"):
            q = ((pts[:, :3] - b[0::2]) / size).astype(np.float32)
            ok = np.all(np.isfinite(q) & (np.abs(q) < 2.0 ** 31), 1)
            idx = np.where(ok[:, None], q, 0).astype(np.int64)
        ok &= np.all((idx >= 0) & (idx < np.asarray(shape)), 1)
        lut = {tuple(cc): v for v, cc in enumerate(coords.tolist())}
        mapping = np.array([lut[tuple(i)] if o else -1 for i, o in zip(idx.tolist(), ok)], np.int64).reshape(-1)
        first = np.full((len(coords),), -1, np.int64)
        for i in range(n - 1, -1, -1):
            if mapping[i] >= 0:
                first[mapping[i]] = i + index_offset
        T = torch.from_numpy
        return T(coords), T(cnt), T(agg), T(first), T(mapping)

    def compact_index(self, keys, ncells):
        u = np.unique(keys.cpu().numpy())
        return u, len(u)

    def compact_lookup(self, handle, keys):
        k = keys.cpu().numpy()
        pos = np.searchsorted(handle, k)
        pos = np.where((pos < len(handle)) & (handle[np.minimum(pos, len(handle) - 1)] == k), pos, -1)
        return torch.from_numpy(pos.astype(np.int64)).to(keys.device)


class LockedOps:
    """serialises the multi-kernel ops of several virtual ranks that share one GPU stream and scratch arena"""

    def __init__(self, ops, lock):
        self._ops, self._lock = ops, lock

    def __getattr__(self, name):
        fn = getattr(self._ops, name)

        def call(*a, **k):
            with self._lock:
                out = fn(*a, **k)
                torch.cuda.synchronize()
                return out
        return call


class ThreadWorld:
    """K virtual ranks = K threads of one process; collectives through shared lists + a barrier"""

    def __init__(self, world):
        self.world = world
        self.barrier = threading.Barrier(world)
        self.slots = [None] * world

    def comm(self, rank):
        return _ThreadComm(self, rank)


class _ThreadComm:
    def __init__(self, w, rank):
        self.w, self.rank, self.world = w, rank, w.world

    def _exchange(self, value):
        self.w.slots[self.rank] = value
        self.w.barrier.wait()
        vals = list(self.w.slots)
        self.w.barrier.wait()
        return vals

    def all_gather_int(self, value, device):
        return [int(v) for v in self._exchange(int(value))]

    def all_gather_var(self, t, sizes):
        return torch.cat(self._exchange(t.clone()))

    def all_reduce(self, t, op):
        vals = self._exchange(t.clone())
        st = torch.stack(vals)
        r = st.sum(0) if op == "sum" else (st.max(0).values if op == "max" else st.min(0).values)
        t.copy_(r.to(t.dtype))
        return t
