"""Test doubles for the sharded voxelizer: CPU compute steps built on the oracle (so that the distributed
orchestration can run under gloo without a GPU) and an in-process thread communicator (K virtual ranks on one GPU)."""
import threading

import numpy as np
import torch

import oracle


class NumpyOps:
    """same interface as d3d_amd.voxel.sharded.HipOps, computed with the CPU oracle + numpy"""

    def voxelize_reduce(self, points, shape, bounds, reduction, index_offset, plain=False, want_coords=True, max_points=0):
        pts = points.cpu().numpy()
        n, c = pts.shape
        red = {1: 1, 2: 2, 3: 3, 4: 1}[int(reduction)]
        r = oracle.voxelize_3d_dense(pts, shape, bounds, max(int(max_points), 1), max(n, 1), red)
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
        v = len(coords)

        def padded(a, fill, extra=0):   # device contract: buffers sized for n voxels, rows >= V are garbage
            out = np.full((n + extra,) + a.shape[1:], fill, a.dtype)
            out[:v] = a
            return torch.from_numpy(out)
        counts = torch.tensor([v, 0, 0, 0], dtype=torch.int64)
        keys = (coords[:, 0] * shape[1] + coords[:, 1]) * shape[2] + coords[:, 2]
        ret = (padded(coords, 7), padded(cnt, 99), padded(agg, 1e30), padded(first, 123), torch.from_numpy(mapping),
               padded(keys.astype(np.int64), -1, extra=1), counts)      # keys[n] = -1 - status (0)
        if max_points:       # every voxel's first min(count, max_points) rows at seg[voxel] (segments of `count` rows, like the kernels)
            seg = np.zeros((max(n, 1),), np.int32)
            seg[:v] = np.cumsum(cnt) - cnt
            rows = np.full((n + 8, 4), np.nan, np.float32)
            for q in range(v):
                k = min(int(cnt[q]), int(max_points))
                rows[seg[q]:seg[q] + k] = r["voxels"][q, :k]
            ret = ret + (torch.from_numpy(seg), torch.from_numpy(rows))
        return ret

    def compact_index(self, keys, ncells, status_stride=None):
        k = keys.cpu().numpy()
        status = 0
        if status_stride:
            for flag in k[status_stride - 1::status_stride]:
                status |= -1 - int(flag)
        u = np.unique(k[k >= 0])
        return u, len(u), status

    def bitmap_mark(self, keys, n, ncells):
        k = keys.cpu().numpy()
        nw = (int(ncells) + 63) // 64
        bits = np.zeros((nw * 64,), np.uint8)
        bits[k[:n][k[:n] >= 0]] = 1
        words = np.packbits(bits.reshape(nw, 64), axis=1, bitorder="little").view("<u8").reshape(nw)
        return torch.from_numpy(np.concatenate([words.view(np.int64), k[n:n + 1]]))

    def compact_from_bitmaps(self, parts_all, world, ncells, rank=None):
        nw = (int(ncells) + 63) // 64
        parts = parts_all.cpu().numpy().reshape(world, nw + 1)
        status = 0
        for flag in parts[:, nw]:
            status |= -1 - int(flag)
        words = np.ascontiguousarray(parts[:, :nw]).view(np.uint64)
        bits = np.unpackbits(words.view(np.uint8).reshape(world, nw, 8), axis=2, bitorder="little").reshape(world, -1)[:, :ncells]
        u = np.flatnonzero(bits.any(0)).astype(np.int64)
        if rank is None:
            return u, len(u), status
        seen = np.zeros((ncells,), bool)
        newc, lower = [], None
        for q in range(world):
            if q == rank:
                lower = seen.copy()
            newc.append(int(np.sum(bits[q].astype(bool) & ~seen)))
            seen |= bits[q].astype(bool)
        return dict(keys=u, lower=lower, newc=newc), len(u), status

    def compact_keys(self, handle, nvox):
        keys = handle["keys"] if isinstance(handle, dict) else handle
        return torch.from_numpy(keys[:nvox].copy())

    def build_table_owned(self, handle, keys_local, n_local, nvox, c, reduction, agg, cnt, rank):
        k = keys_local.cpu().numpy()[:n_local]
        slot = self._lookup(handle["keys"], k)
        mean = int(reduction) == 1
        ident = 0.0 if mean else (-np.inf if int(reduction) == 2 else np.inf)
        table = np.full((nvox, c + 2 if mean else c + 1), ident, np.float32)
        cnt_t = None if mean else np.zeros((nvox,), np.int32)
        mine = slot >= 0
        owned = mine & ~handle["lower"][np.where(mine, k, 0)]
        vid = sum(handle["newc"][:rank]) + np.cumsum(owned) - owned          # exclusive scan in local order
        table[slot[mine], :c] = agg.numpy()[:n_local][mine]
        if mean:
            table[slot[mine], c] = cnt.numpy()[:n_local][mine].astype(np.float32)
        else:
            cnt_t[slot[mine]] = cnt.numpy()[:n_local][mine]
        table[slot[owned], -1] = vid[owned].astype(np.float32)
        t = torch.from_numpy
        return t(table), (None if mean else t(cnt_t)), t(slot)

    def finalize_owned(self, nvox, c, key_of_slot, table, mean, cnt_in, shape):
        t = table.numpy()[:nvox]
        vid = np.round(t[:, -1]).astype(np.int64)
        k = key_of_slot.numpy()[:nvox]
        sy, sz = shape[1], shape[2]
        coords = np.empty((nvox, 3), np.int64)
        coords[vid] = np.stack([k // (sy * sz), (k // sz) % sy, k % sz], 1)
        cnt = np.empty((nvox,), np.int32)
        feats = np.empty((nvox, c), np.float32)
        if mean:
            cnt[vid] = np.round(t[:, c]).astype(np.int32)
            feats[vid] = t[:, :c] / t[:, c:c + 1]
        else:
            cnt[vid] = cnt_in.numpy()[:nvox]
            feats[vid] = t[:, :c]
        return torch.from_numpy(coords), torch.from_numpy(cnt), torch.from_numpy(feats), torch.from_numpy(vid)

    def _lookup(self, handle, k):
        if len(handle) == 0:
            return np.full((len(k),), -1, np.int64)
        pos = np.searchsorted(handle, k)
        return np.where((pos < len(handle)) & (handle[np.minimum(pos, len(handle) - 1)] == k), pos, -1).astype(np.int64)

    def build_table(self, handle, keys_all, begin, n_local, nvox, c, reduction, agg, cnt, first_local, want_keys=True):
        k = keys_all.cpu().numpy()
        slot_all = self._lookup(handle, k)
        mean = int(reduction) == 1
        ident = 0.0 if mean else (-np.inf if int(reduction) == 2 else np.inf)
        table = np.full((nvox, c + 1 if mean else c), ident, np.float32)
        cnt_t = None if mean else np.zeros((nvox,), np.int32)
        first = np.full((nvox,), np.iinfo(np.int64).max, np.int64)
        key_of_slot = np.empty((nvox,), np.int64)
        ok = slot_all >= 0
        key_of_slot[slot_all[ok]] = k[ok]
        slot = slot_all[begin:begin + n_local].copy()
        mine = slot >= 0
        table[slot[mine], :c] = agg.numpy()[:n_local][mine]
        if mean:
            table[slot[mine], c] = cnt.numpy()[:n_local][mine].astype(np.float32)
        else:
            cnt_t[slot[mine]] = cnt.numpy()[:n_local][mine]
        first[slot[mine]] = first_local.numpy()[:n_local][mine]
        t = torch.from_numpy
        return t(table), (None if mean else t(cnt_t)), t(first), t(key_of_slot), t(slot)

    def finalize(self, nvox, c, first, n_total, key_of_slot, table, mean, cnt_in, shape):
        f = first.numpy()[:nvox]
        vid = np.argsort(np.argsort(f, kind="stable"), kind="stable").astype(np.int64)   # rank among the first indices
        k = key_of_slot.numpy()[:nvox]
        t = table.numpy()[:nvox]
        sy, sz = shape[1], shape[2]
        coords = np.empty((nvox, 3), np.int64)
        coords[vid] = np.stack([k // (sy * sz), (k // sz) % sy, k % sz], 1)
        cnt = np.empty((nvox,), np.int32)
        feats = np.empty((nvox, c), np.float32)
        if mean:
            cnt[vid] = np.round(t[:, c]).astype(np.int32)
            feats[vid] = t[:, :c] / t[:, c:c + 1]
        else:
            cnt[vid] = cnt_in.numpy()[:nvox]
            feats[vid] = t[:, :c]
        return torch.from_numpy(coords), torch.from_numpy(cnt), torch.from_numpy(feats), torch.from_numpy(vid)

    # ---- owner-computes exchange: numpy twins of owner.hip (same record layout, any deterministic owner function) ----
    @staticmethod
    def _owner(keys, world):
        h = keys.astype(np.uint64)
        with np.errstate(over="ignore"):
            h ^= h >> np.uint64(33); h *= np.uint64(0xff51afd7ed558ccd)
            h ^= h >> np.uint64(33); h *= np.uint64(0xc4ceb9fe1a85ec53)
            h ^= h >> np.uint64(33)
        return (((h >> np.uint64(32)) * np.uint64(world)) >> np.uint64(32)).astype(np.int64)

    def owner_pack(self, keys, cnt, agg, first, counts, n, c, world, max_points=0, seg=None, rows=None, points_in_shard=None):
        v = int(counts[0])
        k = keys.numpy()[:v]
        words = (5 + c + 1) & ~1
        own = self._owner(k, world)
        perm = np.argsort(own, kind="stable").astype(np.int32)
        send = np.zeros((n, words), np.int32)
        rec = send[:v]
        rec[:, 0:2] = k[perm].astype(np.int64).reshape(-1, 1).view(np.int32)
        rec[:, 2:4] = first.numpy()[:v][perm].reshape(-1, 1).view(np.int32)
        rec[:, 4] = cnt.numpy()[:v][perm]
        rec[:, 5:5 + c] = agg.numpy()[:v][perm].view(np.int32)
        sc = np.zeros((2 * world + 2,), np.int64)
        sc[2 * world + 1] = -1 if points_in_shard is None else int(points_in_shard)
        sc[:world] = np.bincount(own, minlength=world)
        sc[world] = -1 - int(keys.numpy()[n])
        full_perm = np.zeros((n,), np.int32)
        full_perm[:v] = perm
        pos = np.zeros((n,), np.int32)
        pos[perm] = np.arange(v, dtype=np.int32)
        send_rows = None
        if max_points:
            kept = np.minimum(cnt.numpy()[:v][perm], max_points)
            dest = own[perm]
            send_rows = np.zeros((max(n, 1), 4), np.float32)
            at = 0
            for d in range(world):
                inbatch = 0
                for j in np.flatnonzero(dest == d):
                    k = int(kept[j])
                    rec[j, words - 1] = inbatch
                    b = int(seg.numpy()[perm[j]])
                    send_rows[at + inbatch:at + inbatch + k] = rows.numpy()[b:b + k]
                    inbatch += k
                sc[world + 1 + d] = inbatch
                at += inbatch
            send_rows = torch.from_numpy(send_rows)
        return torch.from_numpy(send), torch.from_numpy(full_perm), torch.from_numpy(pos), send_rows, torch.from_numpy(sc)

    def owner_merge(self, recv, recv_counts, world, c, reduction, shape, flags=0, point_off=None):
        r = recv.numpy()
        R = len(r)
        keys = np.ascontiguousarray(r[:, 0:2]).view(np.int64).reshape(-1)
        first = np.ascontiguousarray(r[:, 2:4]).view(np.int64).reshape(-1)
        if point_off is not None:                                 # local indices + the source rank's first point = global
            first = first + np.repeat(np.asarray(point_off, np.int64), [int(k) for k in recv_counts])
        cnt = r[:, 4].copy()
        agg = np.ascontiguousarray(r[:, 5:5 + c]).view(np.float32)
        red = int(reduction)
        ident = 0.0 if red == 1 else (-np.inf if red == 2 else np.inf)
        owned, order = {}, []                                     # voxels in order of their first record (= lowest source rank)
        rec_owned = np.zeros((R,), np.int32)
        for i in range(R):                                        # records arrive grouped by source rank, in rank order
            k = int(keys[i])
            if k not in owned:
                owned[k] = len(order)
                order.append(k)
            rec_owned[i] = owned[k]
        vo = len(order)
        agg_o = np.full((R, c), ident, np.float32)
        cnt_o = np.zeros((R,), np.int32)
        first_o = np.full((R,), np.iinfo(np.int64).max, np.int64)
        for i in range(R):
            o = rec_owned[i]
            if red == 1:
                agg_o[o] = agg_o[o] + agg[i]
            elif red == 2:
                agg_o[o] = np.maximum(agg_o[o], agg[i])
            else:
                agg_o[o] = np.minimum(agg_o[o], agg[i])
            cnt_o[o] += cnt[i]
            first_o[o] = min(first_o[o], first[i])
        if red == 1:
            agg_o[:vo] = agg_o[:vo] / cnt_o[:vo, None].astype(np.float32)
        k = np.asarray(order, np.int64)
        sy, sz = shape[1], shape[2]
        coords = np.zeros((R, 3), np.int64)
        coords[:vo] = np.stack([k // (sy * sz), (k // sz) % sy, k % sz], 1) if vo else np.zeros((0, 3), np.int64)
        t = torch.from_numpy
        handle = dict(recv=r, rec_owned=rec_owned, cnt=cnt, vo=vo, counts=recv_counts, total=cnt_o)
        return t(first_o), t(coords), t(cnt_o), t(agg_o), t(rec_owned), torch.tensor([vo, 0, 0, 0], dtype=torch.int64), handle

    def owner_dense(self, handle, recv_rows, recv_row_counts, max_points):
        r, R, P = handle["recv"], len(handle["recv"]), int(max_points)
        rows = recv_rows.numpy()
        voxels = np.zeros((R, P, 4), np.float32)
        pmask = np.zeros((R, P), np.uint8)
        have = np.zeros((R,), np.int64)
        src_of = np.repeat(np.arange(len(handle["counts"])), handle["counts"])
        rbase = np.concatenate([[0], np.cumsum(recv_row_counts)])
        for i in range(R):                                        # receive order = rank order
            o = handle["rec_owned"][i]
            k = min(int(handle["cnt"][i]), P)
            take = min(k, P - int(have[o]))
            b = int(rbase[src_of[i]]) + int(r[i, -1])
            voxels[o, have[o]:have[o] + take] = rows[b:b + take]
            have[o] += take
        for o in range(handle["vo"]):
            pmask[o, :min(int(handle["total"][o]), P)] = 1
        return torch.from_numpy(voxels), torch.from_numpy(pmask)

    def owner_mark_first(self, first_o, counts_o, n_total):
        nw = (max(n_total, 1) + 63) // 64
        bits = np.zeros((nw * 64,), np.uint8)
        bits[first_o.numpy()[:int(counts_o[0])]] = 1
        return torch.from_numpy(np.packbits(bits.reshape(nw, 64), axis=1, bitorder="little").view("<u8").reshape(nw).view(np.int64).copy())

    def owner_number(self, gbits, n_total, first_o, counts_o):
        w = gbits.numpy().view(np.uint64)
        g = np.unpackbits(w.view(np.uint8).reshape(-1, 8), axis=1, bitorder="little").reshape(-1).astype(np.int64)
        gpre = np.cumsum(g) - g
        vo = int(counts_o[0])
        vids = np.zeros((len(first_o),), np.int64)
        vids[:vo] = gpre[first_o.numpy()[:vo]]
        return torch.from_numpy(vids), torch.tensor([int(g.sum()), 0, 0, 0], dtype=torch.int64)

    def owner_reply(self, rec_owned, vids):
        return vids[rec_owned.long()]

    def owner_map(self, local_map, pos_of_local, back):
        m = local_map.numpy()
        out = np.full(m.shape, -1, np.int64)
        ok = m >= 0
        out[ok] = back.numpy()[pos_of_local.numpy()[m[ok]]]
        return torch.from_numpy(out)

    def owner_replicate(self, nvox, vids, coords_in, cnt_in, feats_in, sizes=None):
        v = vids.numpy()
        coords = np.zeros((nvox, 3), np.int64); cnt = np.zeros((nvox,), np.int32)
        feats = np.zeros((nvox, feats_in.shape[1]), np.float32)
        coords[v] = coords_in.numpy(); cnt[v] = cnt_in.numpy(); feats[v] = feats_in.numpy()
        return torch.from_numpy(coords), torch.from_numpy(cnt), torch.from_numpy(feats)

    def compose_map(self, local_map, slot_of_local, nvox, vid_of_slot):
        m = local_map.numpy()
        out = np.full(m.shape, -1, np.int64)
        ok = m >= 0
        out[ok] = vid_of_slot.numpy()[slot_of_local.numpy()[m[ok]]]
        return torch.from_numpy(out)


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

    def exchange_counts(self, counts):
        return [v.tolist() for v in self._exchange(counts.clone())]

    def all_to_all(self, send, send_counts, recv_counts):
        off = [0]
        for k in send_counts:
            off.append(off[-1] + int(k))
        parts = self._exchange([send[off[d]:off[d + 1]].clone() for d in range(self.world)])
        out = torch.cat([parts[s][self.rank] for s in range(self.world)])
        assert [int(parts[s][self.rank].shape[0]) for s in range(self.world)] == [int(k) for k in recv_counts]
        return out
