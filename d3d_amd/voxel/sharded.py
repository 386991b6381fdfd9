"""Point-sharded voxelization over the GPUs of one node (north_star; the reference has no distributed code).

One process per GPU; rank r holds a contiguous slice of the frame's points (rank order = point order).  The result is the
voxel feature grid of the WHOLE frame -- numbered exactly like the single-GPU contract (first-seen order over the global point
index, voxelize.cpp:119) -- plus the global voxel id of each of the rank's own points.

DEFAULT: owner-computes (`exchange="owner"`, rounds 3-5; kernels in csrc/owner.hip).  A cell has an owner rank,
owner(cell) = mix64(cell) * W >> 64, and per-rank work scales with the SHARD, not with the frame:

  1. local    d3d_voxelize_3d_reduce on the shard alone, point indices local to it (no rank needs another's size first):
              per-voxel partial SUM / MAX / MIN, count, first point, cell key, point -> local voxel map          [HIP]
  2. pack     one partial RECORD per local voxel {cell, first point, count, C partial features}, grouped by owner   [HIP]
  3. counts   ONE small all-gather: records / rows per destination, status bits and the shard's point count of every rank
              (the sizes of the all-to-alls and every rank's global point offset) -- host synchronisation 1 of 2  [RCCL]
  4. records  all_to_all_single of the records: a sparse reduce-scatter of the feature grid                       [RCCL, xGMI]
  5. merge    the owner groups the records of a cell (LDS buckets, d3d_owner_merge), folds the partials in source-rank order
              and turns the leader's local first point into a global index (point_off)                           [HIP]
  6. number   every owner sets bit `first point` of a one-bit-per-point bitmap over the frame; the bitmaps are SUM-all-reduced
              (disjoint sets: SUM = OR; RCCL has no OR) and a popcount scan gives every owned voxel its global id  [RCCL + HIP]
  7. reply    all_to_all_single of the ids back along the records' path; every point's local voxel -> global id   [RCCL + HIP]
  8. sizes    the owned / global voxel counts -- host synchronisation 2 of 2

Four collectives and two host synchronisations per call (five collectives with the dense contract, `max_points=P`, whose
candidate rows travel in a second all-to-all).  `replicate=False` ends here: each rank returns ITS voxels (coords, counts,
features, global ids) -- the scalable form.  `replicate=True` adds an all-gather of the finished rows and a merge into
voxel-id order on every rank (work sized by the frame again; for callers that want the replicated grid).

Exact: coords, counts, numbering, MAX / MIN, the dense contract's rows.  MEAN: partial sums are folded in source-rank order
(the same on every run) and match a sequential pass within fp32 rounding (rtol 1e-5).
Collective payloads (config 5, 8 x 1 M points, 5.9 M voxels, per rank): records 31 MB, ids 6 MB, bitmap 1 MB (+ rows 14 MB).

LEGACY: the replicated-grid exchanges of rounds 1-2 (`exchange="keys" | "bitmap" | "auto"`, still tested): all-gather of the
occupied cells (key lists or occupancy bitmaps), identical compact slots on every rank, all-reduce of the compact voxel table,
every rank finalises the WHOLE grid (`_run`).  Kept for communicators that only implement the round 1-2 protocol.

Never run on more than one GPU in any round (no multi-GPU box was available): world-2 gloo tests on CPU, 8 virtual ranks on
one GPU at config 5's full size, and RCCL (backend "nccl") with one rank.  `tools/sharded_profile.py` prints a MODELLED step
(assumed link and latency constants) next to both single-GPU bases.
"""
import ctypes

import torch

from .. import _lib
from ..utils import Dict

_REDUCTIONS = {"MEAN": 1, "MAX": 2, "MIN": 3}
_SUM = 4
_I64_MAX = (1 << 63) - 1
_MAX_OWNER_WORLD = 16      # kMaxOwnerWorld of grid.hip


class TorchComm:
    """torch.distributed (backend "nccl" = RCCL on ROCm, "gloo" on CPU) behind the three collectives used."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self._dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)

    def all_gather_int(self, value, device):
        t = torch.tensor([int(value)], dtype=torch.int64, device=device)
        out = torch.empty((self.world,), dtype=torch.int64, device=device)
        self._dist.all_gather_into_tensor(out, t, group=self.group)
        return [int(x) for x in out.tolist()]

    def all_gather_var(self, t, sizes):
        """concatenate 1-D tensors of different lengths in rank order (pad to the max, one collective)"""
        cap = max(max(sizes), 1)
        if min(sizes) == cap and t.numel() == cap:           # equal shards: no padding, no slicing
            out = torch.empty((self.world * cap,), dtype=t.dtype, device=t.device)
            self._dist.all_gather_into_tensor(out, t.contiguous(), group=self.group)
            return out
        pad = torch.zeros((cap,), dtype=t.dtype, device=t.device)
        pad[:t.numel()] = t
        out = torch.empty((self.world * cap,), dtype=t.dtype, device=t.device)
        self._dist.all_gather_into_tensor(out, pad, group=self.group)
        return torch.cat([out[r * cap:r * cap + sizes[r]] for r in range(self.world)])

    def all_reduce(self, t, op):
        ops = {"sum": self._dist.ReduceOp.SUM, "max": self._dist.ReduceOp.MAX, "min": self._dist.ReduceOp.MIN}
        self._dist.all_reduce(t, op=ops[op], group=self.group)
        return t

    def exchange_counts(self, counts):
        """counts: int64[k] on the device, one row per rank -> the world x k matrix on the host (a host synchronisation)"""
        out = torch.empty((self.world * counts.numel(),), dtype=torch.int64, device=counts.device)
        self._dist.all_gather_into_tensor(out, counts.contiguous(), group=self.group)
        return out.view(self.world, -1).tolist()

    def all_to_all(self, send, send_counts, recv_counts):
        """rows of `send` grouped by destination (send_counts rows each) -> rows grouped by source (recv_counts)"""
        out = torch.empty((sum(recv_counts),) + tuple(send.shape[1:]), dtype=send.dtype, device=send.device)
        self._dist.all_to_all_single(out, send.contiguous(), output_split_sizes=list(recv_counts),
                                     input_split_sizes=list(send_counts), group=self.group)
        return out


class LocalComm:
    """world of one rank: the sharded operator without torch.distributed (single-GPU feature grid; bench.py's
    same-operator base of the weak-scaling curve)"""
    rank, world = 0, 1

    def all_gather_int(self, value, device):
        return [int(value)]

    def all_gather_var(self, t, sizes):
        return t

    def all_reduce(self, t, op):
        return t

    def exchange_counts(self, counts):
        return [counts.tolist()]

    def all_to_all(self, send, send_counts, recv_counts):
        return send


class _RowSource:
    """what HipOps.voxelize_reduce(max_points) hands to HipOps.owner_pack: the `rows` buffer of d3d_voxelize_3d_reduce (staged
    rows, or ranked point indices into `points`) together with the shard it came from"""
    __slots__ = ("buf", "points", "index_offset")

    def __init__(self, buf, points, index_offset):
        self.buf, self.points, self.index_offset = buf, points, index_offset


class HipOps:
    """the compute steps, on the HIP kernels of libd3d_hip.so (no host synchronisation except where noted)"""

    def voxelize_reduce(self, points, shape, bounds, reduction, index_offset, plain=False, want_coords=True, max_points=0):
        """-> coords[n,3], cnt[n], agg[n,c], first[n], mapping[n], keys[n], counts[4] -- all on the device and all
        sized for n voxels; only the first counts[0] rows are meaningful.  keys has n + 1 entries: -1 beyond the
        voxels, and keys[n] = -1 - status bits (so the status reaches every rank with the key all-gather)."""
        lib = _lib.load()
        pts = points.contiguous()
        dev = pts.device
        n, c = pts.shape
        shape_h = (ctypes.c_int32 * 3)(*[int(x) for x in shape])
        bound_h = (ctypes.c_float * 6)(*[float(x) for x in bounds])
        with torch.cuda.device(dev):
            coords = torch.empty((n, 3), dtype=torch.int64, device=dev) if want_coords else None
            cnt = torch.empty((n,), dtype=torch.int32, device=dev)
            agg = torch.empty((n, c), dtype=torch.float32, device=dev)
            first = torch.empty((n,), dtype=torch.int64, device=dev)
            mapping = torch.empty((n,), dtype=torch.int64, device=dev)
            keys = torch.empty((n + 1,), dtype=torch.int64, device=dev)
            counts = torch.empty((_lib.NUM_COUNTS,), dtype=torch.int64, device=dev)
            seg = rows = None
            if max_points:     # dense contract: every voxel's first min(count, max_points) rows, in point order
                seg = torch.empty((max(n, 1),), dtype=torch.int32, device=dev)
                rows = torch.empty((lib.d3d_voxelize_reduce_rows(n), 4), dtype=torch.float32, device=dev)
            ws = _lib.workspace(lib.d3d_voxelize_workspace_bytes(n, 0), dev)
            from .. import options
            flags = options.current().voxel_flags
            if plain:      # the retry after a status overflow: general slots in the hash table (no bucket capacity to outgrow)
                flags |= _lib.VOXEL_PATH_HASH | _lib.VOXEL_PLAIN_SLOTS
            rc = lib.d3d_voxelize_3d_reduce(
                _lib.ptr(pts), n, c, ctypes.cast(shape_h, ctypes.c_void_p), ctypes.cast(bound_h, ctypes.c_void_p),
                int(reduction), int(index_offset), _lib.ptr(coords), _lib.ptr(cnt), _lib.ptr(agg), _lib.ptr(first),
                _lib.ptr(mapping), _lib.ptr(keys), int(max_points), _lib.ptr(seg), _lib.ptr(rows), _lib.ptr(counts), _lib.ptr(ws),
                ws.numel(), _lib.stream_ptr(), flags)
            if rc == _lib.ERR_UNSUPPORTED and max_points:
                raise ValueError("the sharded dense contract needs points[n, 4] float32, 16-byte aligned")
            _lib.check(rc, "voxelize_3d_reduce")
        if max_points:
            # (on the binned index `rows` holds ranked point indices, not rows -- counts[3] says which -- and owner_pack gathers
            # the rows from the points: they travel with the buffer)
            return coords, cnt, agg, first, mapping, keys, counts, seg, _RowSource(rows, pts, int(index_offset))
        return coords, cnt, agg, first, mapping, keys, counts

    def compact_index(self, keys, ncells, status_stride=None):
        """-> (handle, number of distinct keys, status bits OR-ed over the ranks); handle feeds build_table.
        Negative keys are ignored; with status_stride, every status_stride-th key (the last of each rank's block)
        is a status row.  Reading the total is THE host synchronisation of a sharded call."""
        lib = _lib.load()
        dev = keys.device
        keys = keys.contiguous()
        with torch.cuda.device(dev):
            ws = torch.empty((lib.d3d_grid_compact_workspace_bytes(ncells),), dtype=torch.uint8, device=dev)
            counts = torch.empty((_lib.NUM_COUNTS,), dtype=torch.int64, device=dev)
            rc = lib.d3d_grid_compact_index(_lib.ptr(keys), keys.numel(), int(ncells), _lib.ptr(counts), _lib.ptr(ws),
                                            ws.numel(), _lib.stream_ptr())
            _lib.check(rc, "grid_compact_index")
            if status_stride:
                host = torch.cat([counts[:1], keys[status_stride - 1::status_stride]]).tolist()
            else:
                host = counts[:1].tolist()
        status = 0
        for flag in host[1:]:
            status |= -1 - int(flag)
        return (ws, int(ncells)), int(host[0]), status

    def bitmap_mark(self, keys, n, ncells):
        """this rank's occupancy bitmap as int64 words + one trailing status word (keys[n] = -1 - status bits)"""
        lib = _lib.load()
        dev = keys.device
        nw = (int(ncells) + 63) // 64
        with torch.cuda.device(dev):
            part = torch.empty((nw + 1,), dtype=torch.int64, device=dev)
            rc = lib.d3d_grid_bitmap_mark(_lib.ptr(keys), int(n), int(ncells), _lib.ptr(part), _lib.stream_ptr())
            _lib.check(rc, "grid_bitmap_mark")
            part[nw:] = keys[n:n + 1]
        return part

    def compact_from_bitmaps(self, parts_all, world, ncells, rank=None):
        """-> (handle, number of occupied cells, status bits OR-ed over the ranks) from the all-gathered bitmaps;
        the host read-back of a sharded call in bitmap mode.  With `rank` the handle also carries the ownership
        bookkeeping (cells a lower rank has, cells owned per rank) that build_table_owned needs."""
        lib = _lib.load()
        dev = parts_all.device
        nw = (int(ncells) + 63) // 64
        with torch.cuda.device(dev):
            ws = torch.empty((lib.d3d_grid_compact_workspace_bytes(ncells),), dtype=torch.uint8, device=dev)
            ows = None
            if rank is not None:
                ows = torch.empty((lib.d3d_grid_owner_workspace_bytes(ncells),), dtype=torch.uint8, device=dev)
            counts = torch.empty((_lib.NUM_COUNTS,), dtype=torch.int64, device=dev)
            rc = lib.d3d_grid_compact_from_bitmaps(_lib.ptr(parts_all), nw + 1, int(world), int(ncells), _lib.ptr(counts),
                                                   _lib.ptr(ws), ws.numel(), int(rank or 0), _lib.ptr(ows),
                                                   ows.numel() if ows is not None else 0, _lib.stream_ptr())
            _lib.check(rc, "grid_compact_from_bitmaps")
            host = torch.cat([counts[:1], parts_all[nw::nw + 1]]).tolist()
        status = 0
        for flag in host[1:]:
            status |= -1 - int(flag)
        return (ws, int(ncells), ows), int(host[0]), status

    def build_table_owned(self, handle, keys_local, n_local, nvox, c, reduction, agg, cnt, rank):
        """identity-filled all-reduce operand with this rank's partial rows in place and, for the voxels this rank owns,
        the global voxel id in the last column: -> table[nvox, c + 2 | c + 1], cnt_table[nvox] | None, slot_of_local"""
        lib = _lib.load()
        ws, ncells, ows = handle
        dev = keys_local.device
        mean = int(reduction) == 1
        with torch.cuda.device(dev):
            table = torch.empty((nvox, c + 2 if mean else c + 1), dtype=torch.float32, device=dev)
            cnt_t = None if mean else torch.empty((nvox,), dtype=torch.int32, device=dev)
            slot = torch.empty((n_local,), dtype=torch.int64, device=dev)
            sws = torch.empty(((n_local // 1024 + 2) * 8 + 1024,), dtype=torch.uint8, device=dev)
            rc = lib.d3d_sharded_scatter_owned(_lib.ptr(keys_local), int(n_local), ncells, _lib.ptr(ws), ws.numel(),
                                               _lib.ptr(ows), ows.numel(), int(rank), int(nvox), int(c), int(reduction),
                                               _lib.ptr(agg), _lib.ptr(cnt), _lib.ptr(table), table.shape[1], _lib.ptr(cnt_t),
                                               _lib.ptr(slot), _lib.ptr(sws), sws.numel(), _lib.stream_ptr())
            _lib.check(rc, "sharded_scatter_owned")
        return table, cnt_t, slot

    def finalize_owned(self, nvox, c, key_of_slot, table, mean, cnt_in, shape):
        """slot-ordered reduced table (id in the last column) -> voxel-id-ordered (coords, counts, features) + vid_of_slot"""
        lib = _lib.load()
        dev = table.device
        shape_h = (ctypes.c_int32 * 3)(*[int(x) for x in shape])
        with torch.cuda.device(dev):
            coords = torch.empty((nvox, 3), dtype=torch.int64, device=dev)
            cnt = torch.empty((nvox,), dtype=torch.int32, device=dev)
            feats = torch.empty((nvox, c), dtype=torch.float32, device=dev)
            vid = torch.empty((nvox,), dtype=torch.int64, device=dev)
            rc = lib.d3d_sharded_finalize_owned(nvox, c, _lib.ptr(key_of_slot), _lib.ptr(table), table.shape[1],
                                                1 if mean else 0, _lib.ptr(cnt_in), ctypes.cast(shape_h, ctypes.c_void_p),
                                                _lib.ptr(vid), _lib.ptr(coords), _lib.ptr(cnt), _lib.ptr(feats),
                                                _lib.stream_ptr())
            _lib.check(rc, "sharded_finalize_owned")
        return coords, cnt, feats, vid

    def compact_keys(self, handle, nvox):
        lib = _lib.load()
        ws, ncells = handle[0], handle[1]
        with torch.cuda.device(ws.device):
            key_of_slot = torch.empty((nvox,), dtype=torch.int64, device=ws.device)
            rc = lib.d3d_grid_compact_keys(ncells, _lib.ptr(ws), ws.numel(), _lib.ptr(key_of_slot), _lib.stream_ptr())
            _lib.check(rc, "grid_compact_keys")
        return key_of_slot

    def build_table(self, handle, keys_all, begin, n_local, nvox, c, reduction, agg, cnt, first_local, want_keys=True):
        """identity-filled all-reduce operands with this rank's partial rows in place:
        -> table[nvox, c(+1)], cnt_table[nvox] (None for mean), first[nvox], key_of_slot[nvox], slot_of_local[n_local]"""
        lib = _lib.load()
        ws, ncells = handle[0], handle[1]
        dev = keys_all.device
        mean = int(reduction) == 1
        with torch.cuda.device(dev):
            table = torch.empty((nvox, c + 1 if mean else c), dtype=torch.float32, device=dev)
            cnt_t = None if mean else torch.empty((nvox,), dtype=torch.int32, device=dev)
            first = torch.empty((nvox,), dtype=torch.int64, device=dev)
            key_of_slot = torch.empty((nvox,), dtype=torch.int64, device=dev) if want_keys else None
            slot = torch.empty((n_local,), dtype=torch.int64, device=dev)
            rc = lib.d3d_sharded_scatter(_lib.ptr(keys_all), keys_all.numel(), int(begin), int(n_local), ncells, _lib.ptr(ws),
                                         ws.numel(), int(nvox), int(c), int(reduction), _lib.ptr(agg), _lib.ptr(cnt),
                                         _lib.ptr(first_local), _lib.ptr(table), table.shape[1], _lib.ptr(cnt_t),
                                         _lib.ptr(first), _lib.ptr(key_of_slot), _lib.ptr(slot), _lib.stream_ptr())
            _lib.check(rc, "sharded_scatter")
        return table, cnt_t, first, key_of_slot, slot

    def finalize(self, nvox, c, first, n_total, key_of_slot, table, mean, cnt_in, shape):
        """slot-ordered reduced table -> voxel-id-ordered (coords, counts, features) + vid_of_slot"""
        lib = _lib.load()
        dev = table.device
        shape_h = (ctypes.c_int32 * 3)(*[int(x) for x in shape])
        with torch.cuda.device(dev):
            coords = torch.empty((nvox, 3), dtype=torch.int64, device=dev)
            cnt = torch.empty((nvox,), dtype=torch.int32, device=dev)
            feats = torch.empty((nvox, c), dtype=torch.float32, device=dev)
            vid = torch.empty((nvox,), dtype=torch.int64, device=dev)
            counts = torch.empty((_lib.NUM_COUNTS,), dtype=torch.int64, device=dev)
            ws = torch.empty((lib.d3d_grid_compact_workspace_bytes(max(n_total, 1)),), dtype=torch.uint8, device=dev)
            rc = lib.d3d_sharded_finalize(nvox, c, _lib.ptr(first), int(n_total), _lib.ptr(counts), _lib.ptr(ws), ws.numel(),
                                          _lib.ptr(key_of_slot), _lib.ptr(table), table.shape[1], 1 if mean else 0,
                                          _lib.ptr(cnt_in), ctypes.cast(shape_h, ctypes.c_void_p), _lib.ptr(vid),
                                          _lib.ptr(coords), _lib.ptr(cnt), _lib.ptr(feats), _lib.stream_ptr())
            _lib.check(rc, "sharded_finalize")
        return coords, cnt, feats, vid

    # ---- owner-computes exchange (owner.hip) ----
    def owner_pack(self, keys, cnt, agg, first, counts, n, c, world, max_points=0, seg=None, rows=None, points_in_shard=None):
        """local voxels -> records grouped by owner rank: send[n, words] int32, perm[n] (send position -> local voxel),
        pos_of_local[n] (its inverse), send_rows[n, 4] | None (dense contract: the voxels' ranked rows, same grouping),
        send_counts[2 world + 2] (device: records per destination, status bits, rows per destination, points_in_shard)"""
        lib = _lib.load()
        dev = keys.device
        words = lib.d3d_owner_record_words(c)
        with torch.cuda.device(dev):
            send = torch.empty((n, words), dtype=torch.int32, device=dev)
            perm = torch.empty((n,), dtype=torch.int32, device=dev)
            pos = torch.empty((n,), dtype=torch.int32, device=dev)
            send_rows = torch.empty((max(n, 1), 4), dtype=torch.float32, device=dev) if max_points else None
            # (one word more than the C entry point fills: the shard's point count, which travels with the record counts so
            # that no separate size exchange is needed -- ShardedVoxelGenerator._run_owner)
            sc = torch.empty((2 * world + 2,), dtype=torch.int64, device=dev)
            sc[2 * world + 1:].fill_(int(points_in_shard) if points_in_shard is not None else -1)
            ws = torch.empty((lib.d3d_owner_pack_workspace_bytes(n, world),), dtype=torch.uint8, device=dev)
            src = rows if isinstance(rows, _RowSource) else _RowSource(rows, None, 0)
            rc = lib.d3d_owner_pack(_lib.ptr(keys), _lib.ptr(cnt), _lib.ptr(agg), _lib.ptr(first), _lib.ptr(counts), n, c, world,
                                    int(max_points), _lib.ptr(seg), _lib.ptr(src.buf), _lib.ptr(send), _lib.ptr(perm), _lib.ptr(pos),
                                    _lib.ptr(send_rows), _lib.ptr(sc), _lib.ptr(ws), ws.numel(), _lib.stream_ptr(),
                                    _lib.ptr(src.points), int(src.index_offset))
            _lib.check(rc, "owner_pack")
        return send, perm, pos, send_rows, sc

    def owner_merge(self, recv, recv_counts, world, c, reduction, shape, flags=0, point_off=None):
        """records grouped by source rank -> this owner's voxels in global id order, finished:
        first_o, coords, npoints, feats (R rows allocated), rec_owned[R], counts (device; [0] = owned voxels, [2] = status:
        BIN_OVERFLOW asks for flags=OWNER_MERGE_CHAINS), and a handle (leader records + the cells' record lists) for
        owner_dense.  point_off (list of world ints, or None): the records carry point indices LOCAL to their source rank's shard;
        point_off[s] = global index of rank s's first point (added to the leader's index: first_o is global either way)"""
        lib = _lib.load()
        dev = recv.device
        R = int(recv.shape[0])
        off = [0]
        for k in recv_counts:
            off.append(off[-1] + int(k))
        shape_h = (ctypes.c_int32 * 3)(*[int(x) for x in shape])
        with torch.cuda.device(dev):
            # record offsets and (optionally) point offsets of the sources: ONE small host -> device copy
            both = torch.tensor(off + ([int(x) for x in point_off] if point_off is not None else []), dtype=torch.int64, device=dev)
            src_off = both[:world + 1]
            poff = both[world + 1:] if point_off is not None else None
            first_o = torch.empty((R,), dtype=torch.int64, device=dev)
            coords = torch.empty((R, 3), dtype=torch.int64, device=dev)
            npoints = torch.empty((R,), dtype=torch.int32, device=dev)
            feats = torch.empty((R, c), dtype=torch.float32, device=dev)
            rec_owned = torch.empty((R,), dtype=torch.int32, device=dev)
            lead = torch.empty((R,), dtype=torch.int32, device=dev)
            counts = torch.empty((_lib.NUM_COUNTS,), dtype=torch.int64, device=dev)
            ws = torch.empty((lib.d3d_owner_merge_workspace_bytes(R, world),), dtype=torch.uint8, device=dev)   # kept for owner_dense
            rc = lib.d3d_owner_merge(_lib.ptr(recv), R, _lib.ptr(src_off), world, c, int(reduction),
                                     ctypes.cast(shape_h, ctypes.c_void_p), _lib.ptr(first_o), _lib.ptr(coords), _lib.ptr(npoints),
                                     _lib.ptr(feats), _lib.ptr(rec_owned), _lib.ptr(lead), _lib.ptr(counts), _lib.ptr(ws), ws.numel(),
                                     _lib.stream_ptr(), int(flags), _lib.ptr(poff))
            _lib.check(rc, "owner_merge")
        return first_o, coords, npoints, feats, rec_owned, counts, (recv, lead, npoints, counts, ws, world, int(flags))

    def owner_dense(self, handle, recv_rows, recv_row_counts, max_points, resident=None):
        """the dense contract of the owned voxels (id order): voxels[R, max_points, 4], pmask[R, max_points] uint8.
        resident: a d3d_amd.voxel.DenseOutputBuffer of at least R voxels -- `voxels` is then that buffer"""
        lib = _lib.load()
        recv, lead, npoints, counts, ws, world, flags = handle
        dev = recv.device
        R = int(recv.shape[0])
        off = [0]
        for k in recv_row_counts:
            off.append(off[-1] + int(k))
        with torch.cuda.device(dev):
            roff = torch.tensor(off, dtype=torch.int64, device=dev)
            if resident is not None and (resident.capacity < R or resident.max_points != int(max_points) or resident.device != dev):
                raise ValueError("resident output: capacity, max_points or device do not fit")
            voxels = resident.voxels if resident is not None else torch.empty((R, max_points, 4), dtype=torch.float32, device=dev)
            pmask = torch.empty((R, max_points), dtype=torch.uint8, device=dev)
            rc = lib.d3d_owner_dense(_lib.ptr(recv), R, _lib.ptr(recv_rows), _lib.ptr(roff), world, int(max_points), _lib.ptr(lead),
                                     _lib.ptr(npoints), _lib.ptr(counts), R, _lib.ptr(ws), ws.numel(), _lib.ptr(voxels),
                                     _lib.ptr(pmask), _lib.stream_ptr(), flags,
                                     _lib.ptr(resident.row_state) if resident is not None else None)
            _lib.check(rc, "owner_dense")
        return voxels, pmask

    def owner_mark_first(self, first_o, counts_o, n_total):
        """-> int64 words of the bitmap over the frame's point indices with this owner's first points set, + one word: this
        owner's merge has to be repeated (summed over the ranks by the same all-reduce)"""
        lib = _lib.load()
        dev = first_o.device
        with torch.cuda.device(dev):
            bits = torch.empty(((max(n_total, 1) + 63) // 64 + 1,), dtype=torch.int64, device=dev)
            rc = lib.d3d_owner_mark_first(_lib.ptr(first_o), _lib.ptr(counts_o), first_o.numel(), n_total, _lib.ptr(bits),
                                          _lib.stream_ptr())
            _lib.check(rc, "owner_mark_first")
        return bits

    def owner_number(self, gbits, n_total, first_o, counts_o):
        """-> vids[R] (global id of every owned voxel), counts_out (device; [0] = voxels of the whole frame)"""
        lib = _lib.load()
        dev = first_o.device
        R = int(first_o.numel())
        with torch.cuda.device(dev):
            vids = torch.empty((R,), dtype=torch.int64, device=dev)
            counts = torch.empty((_lib.NUM_COUNTS,), dtype=torch.int64, device=dev)
            ws = _lib.workspace(lib.d3d_owner_number_workspace_bytes(n_total), dev)
            rc = lib.d3d_owner_number(_lib.ptr(gbits), n_total, _lib.ptr(first_o), _lib.ptr(counts_o), R, _lib.ptr(vids),
                                      _lib.ptr(counts), _lib.ptr(ws), ws.numel(), _lib.stream_ptr())
            _lib.check(rc, "owner_number")
        return vids, counts

    def owner_reply(self, rec_owned, vids):
        lib = _lib.load()
        R = int(rec_owned.numel())
        with torch.cuda.device(rec_owned.device):
            reply = torch.empty((R,), dtype=torch.int64, device=rec_owned.device)
            _lib.check(lib.d3d_owner_reply(R, _lib.ptr(rec_owned), _lib.ptr(vids), _lib.ptr(reply), _lib.stream_ptr()),
                       "owner_reply")
        return reply

    def owner_map(self, local_map, pos_of_local, back):
        lib = _lib.load()
        dev = local_map.device
        n = int(local_map.numel())
        with torch.cuda.device(dev):
            gmap = torch.empty((n,), dtype=torch.int64, device=dev)
            _lib.check(lib.d3d_owner_map(n, _lib.ptr(local_map), _lib.ptr(pos_of_local), _lib.ptr(back), _lib.ptr(gmap),
                                         _lib.stream_ptr()), "owner_map")
        return gmap

    def owner_replicate(self, nvox, vids, coords_in, cnt_in, feats_in, sizes=None):
        """rows of all owners -> voxel-id order.  sizes: rows per rank when the rows are the ranks' blocks in rank order, each
        ascending in id (the all-gather's layout): merged with coalesced accesses instead of scattered row by row"""
        lib = _lib.load()
        dev = vids.device
        c = int(feats_in.shape[1])
        with torch.cuda.device(dev):
            src_off = ws = None
            if sizes is not None:
                off = [0]
                for k in sizes:
                    off.append(off[-1] + int(k))
                src_off = torch.tensor(off, dtype=torch.int64, device=dev)
                ws = torch.empty(((nvox // 1024 + 2) * len(sizes),), dtype=torch.int64, device=dev)
            coords = torch.empty((nvox, 3), dtype=torch.int64, device=dev)
            cnt = torch.empty((nvox,), dtype=torch.int32, device=dev)
            feats = torch.empty((nvox, c), dtype=torch.float32, device=dev)
            _lib.check(lib.d3d_owner_replicate(nvox, _lib.ptr(vids), _lib.ptr(coords_in), _lib.ptr(cnt_in), _lib.ptr(feats_in), c,
                                               _lib.ptr(coords), _lib.ptr(cnt), _lib.ptr(feats), _lib.stream_ptr(),
                                               _lib.ptr(src_off), len(sizes) if sizes is not None else 0,
                                               _lib.ptr(ws), ws.numel() * 8 if ws is not None else 0),
                       "owner_replicate")
        return coords, cnt, feats

    def compose_map(self, local_map, slot_of_local, nvox, vid_of_slot):
        lib = _lib.load()
        dev = local_map.device
        with torch.cuda.device(dev):
            gmap = torch.empty_like(local_map)
            rc = lib.d3d_sharded_map(local_map.numel(), _lib.ptr(local_map), _lib.ptr(slot_of_local), nvox,
                                     _lib.ptr(vid_of_slot), _lib.ptr(gmap), _lib.stream_ptr())
            _lib.check(rc, "sharded_map")
        return gmap


def _status_retry(counts_host):
    status = int(counts_host[_lib.COUNT_STATUS])
    if status & _lib.STATUS_TABLE_FULL:
        raise RuntimeError("voxelize_3d_reduce: internal hash table overflow")
    # a voxel outgrew the packed slot counter, or a bucket of the binned index its workgroup: redo on the general path
    return bool(status & (_lib.STATUS_PACK_OVERFLOW | _lib.STATUS_BIN_OVERFLOW))


def voxelize_reduce(points, shape, bounds, reduction="mean"):
    """single-GPU "dynamic voxelization": Dict(coords, voxel_npoints, aggregates, points_mapping, voxel_first)"""
    red = _REDUCTIONS[reduction.upper()]
    ops = HipOps()
    out = ops.voxelize_reduce(points, shape, bounds, red, 0)
    host = out[6].cpu()
    if _status_retry(host):
        out = ops.voxelize_reduce(points, shape, bounds, red, 0, plain=True)
        host = out[6].cpu()
    v = int(host[_lib.COUNT_VOXELS])
    coords, cnt, agg, first, mapping, _, _ = out
    return Dict(coords=coords[:v], voxel_npoints=cnt[:v], aggregates=agg[:v], points_mapping=mapping,
                voxel_first=first[:v])


class ShardedVoxelGenerator:
    """Voxel feature grid of a frame whose points are sharded over the ranks of `group` (contiguous slices in
    rank order).  Grid arguments as d3d.voxel.VoxelGenerator (bounds, shape); reduction in {mean, max, min}."""

    def __init__(self, bounds, shape, reduction="mean", group=None, comm=None, ops=None, exchange="owner", replicate=True,
                 max_points=None, debug_checks=False, merge_flags=0, resident=False):
        key = (reduction or "").upper()
        if key not in _REDUCTIONS:
            raise ValueError("Unsupported reduction type in VoxelGenerator!")
        self._red = _REDUCTIONS[key]
        self._bounds = [float(b) for b in bounds]
        self._shape = [int(s) for s in shape]
        if len(self._bounds) != 6 or len(self._shape) != 3 or min(self._shape) <= 0:
            raise ValueError("bounds must have 6 entries and shape 3 positive entries")
        self._ncells = self._shape[0] * self._shape[1] * self._shape[2]
        self._comm = comm if comm is not None else TorchComm(group)
        self._ops = ops if ops is not None else HipOps()
        if exchange not in ("owner", "auto", "keys", "bitmap"):
            raise ValueError("exchange must be owner, auto, keys or bitmap")
        # "owner" (default): owner-computes -- partial voxel records go to the cell's owner rank (all-to-all), which merges,
        # numbers and finishes 1/world of the frame's voxels; per-rank work scales with the shard.  replicate=False then
        # returns each rank's OWNED voxels (in global id order, with their ids) instead of all-gathering the grid.
        # "keys" / "bitmap" / "auto": the replicated-grid exchanges (all-gather of the occupied cells + all-reduce of a
        # compact table every rank finalises in full).
        # A caller-supplied `comm` written for the replicated-grid protocol of rounds 1-2 (all_gather_int, all_gather_var,
        # all_reduce) lacks the two collectives the owner-computes exchange adds (exchange_counts, all_to_all -- protocol in
        # INTEGRATION.md section 6): the default then stays what that comm can run; asking for what it cannot is an error
        # here, not an AttributeError in the middle of a collective sequence.
        if exchange == "owner" and not (hasattr(self._comm, "exchange_counts") and hasattr(self._comm, "all_to_all")):
            if replicate and not max_points:
                exchange = "auto"
            else:
                raise TypeError("comm lacks exchange_counts / all_to_all: replicate=False and max_points need the "
                                "owner-computes exchange (INTEGRATION.md section 6)")
        self._exchange = exchange
        self._replicate = bool(replicate)
        # debug_checks: assert the two invariants the global numbering rests on (costs a device reduction per call): the
        # owned voxels arrive ordered by their first point, i.e. local voxel ids follow first-seen order, and shards are
        # contiguous point ranges in rank order
        self._debug_checks = bool(debug_checks)
        self._merge_flags = int(merge_flags)        # d3d_owner_merge's flags (tests: OWNER_MERGE_CHAINS, OWNER_MERGE_TEST_TINY)
        if not self._replicate and exchange != "owner":
            raise ValueError("replicate=False needs exchange='owner'")
        # max_points: also the dense contract's voxels[V, max_points, 4] + voxel_pmask (voxelize.cpp:128-134: the first
        # max_points points of every voxel by global index), assembled by the voxel's owner from the ranks' candidate rows
        self._max_points = int(max_points) if max_points else 0
        # resident (dense contract, replicate=False): the owned voxels' voxels[Vo, max_points, 4] comes back as a view of a
        # buffer this generator keeps on the device -- only rows with points and stale rows are stored per frame, the zero
        # padding stays (d3d_amd.voxel.DenseOutputBuffer, d3d_owner_dense's row_state); valid until the next call
        self._resident, self._resident_buf = bool(resident), None
        if self._resident and (self._replicate or not self._max_points):
            raise ValueError("resident needs the dense contract (max_points) with replicate=False")
        if self._max_points and exchange != "owner":
            raise ValueError("max_points (the dense contract) needs exchange='owner'")
        if self._max_points > 256:
            raise ValueError("max_points <= 256 on the sharded path")
        self.last_stats = None

    def _layout(self, n, dev):
        """shard sizes over the ranks -> (offset of this rank's first point, total points, largest shard).  Exchanged on
        EVERY call (one 8-byte all-gather + host read): whether a rank's own size changed says nothing about the other
        ranks', and a collective that only some ranks enter desynchronises the whole sequence."""
        sizes = self._comm.all_gather_int(n, dev)
        return sum(sizes[:self._comm.rank]), sum(sizes), max(max(sizes), 1)

    def __call__(self, points):
        run = self._run_owner if self._exchange == "owner" else self._run
        for plain in (False, True):
            out = run(points, plain)
            if out is not None:
                return out
        raise RuntimeError("sharded voxelization failed")

    def _run_owner(self, points, plain):
        """owner-computes exchange (module docstring).  FOUR collectives (five with the dense contract's rows) and TWO host
        synchronisations per call: the count matrix -- which since round 5 also carries every shard's point count, so the local
        pass runs on shard-local point indices and nobody waits for a size exchange first -- and the output sizes at the end."""
        comm, ops = self._comm, self._ops
        dev = points.device
        n, c = points.shape
        W = comm.world
        mean = self._red == 1
        kw = {"plain": True} if plain else {}
        P = self._max_points
        if P and c != 4:
            raise ValueError("the sharded dense contract needs points[n, 4]")
        out = ops.voxelize_reduce(points, self._shape, self._bounds, _SUM if mean else self._red, 0, want_coords=False,
                                  max_points=P, **kw)                  # (first points as indices into THIS shard)
        _, cnt_r, agg_r, first_r, map_r, keys_r, counts_r = out[:7]
        seg_r, rows_r = out[7:] if P else (None, None)
        send, perm, pos_r, send_rows, sc_dev = ops.owner_pack(keys_r, cnt_r, agg_r, first_r, counts_r, n, c, W, P, seg_r, rows_r,
                                                              points_in_shard=n)
        # [src][dst] records, [src][W] status bits, [src][W+1+dst] rows, [src][2W+1] points of the shard -- host sync 1 of 2
        mat = comm.exchange_counts(sc_dev)
        shard = [int(row[2 * W + 1]) for row in mat]
        if min(shard) < 0:
            raise RuntimeError("owner_pack of some rank did not report its shard size (an `ops` of the round-4 protocol?)")
        point_off = [0] * W
        for r in range(1, W):
            point_off[r] = point_off[r - 1] + shard[r - 1]
        n_total = point_off[-1] + shard[-1]
        status = 0
        for row in mat:
            status |= int(row[W])
        if status & _lib.STATUS_TABLE_FULL:
            raise RuntimeError("voxelize_3d_reduce: internal hash table overflow")
        if status & (_lib.STATUS_PACK_OVERFLOW | _lib.STATUS_BIN_OVERFLOW) and not plain:
            return None        # some rank hit a capacity limit of the fast index (rare): all ranks redo on the general path
        sc = [int(x) for x in mat[comm.rank][:W]]
        rc = [int(mat[s][comm.rank]) for s in range(W)]
        recv = comm.all_to_all(send[:sum(sc)], sc, rc)
        first_o, coords, npoints, feats, rec_owned, counts_o, handle = ops.owner_merge(
            recv, rc, W, c, self._red, self._shape, flags=self._merge_flags | (_lib.OWNER_MERGE_CHAINS if plain else 0),
            point_off=point_off)
        voxels = pmask = None
        if P:
            rsc = [int(x) for x in mat[comm.rank][W + 1:2 * W + 1]]
            rrc = [int(mat[s][W + 1 + comm.rank]) for s in range(W)]
            recv_rows = comm.all_to_all(send_rows[:sum(rsc)], rsc, rrc)
            if self._resident:
                from . import DenseOutputBuffer
                buf, need = self._resident_buf, int(recv.shape[0])        # (rows are indexed by owned voxel id < records received)
                if buf is None or buf.capacity < need or buf.device != recv.device:
                    grow = need if buf is None else max(need, buf.capacity + buf.capacity // 4)
                    buf = self._resident_buf = DenseOutputBuffer(grow, P, recv.device)
                voxels, pmask = ops.owner_dense(handle, recv_rows, rrc, P, resident=buf)
            else:
                voxels, pmask = ops.owner_dense(handle, recv_rows, rrc, P)
        lbits = ops.owner_mark_first(first_o, counts_o, n_total)
        gbits = comm.all_reduce(lbits, "sum")                   # disjoint bit sets: their sum is their OR
        vids, counts_out = ops.owner_number(gbits, n_total, first_o, counts_o)
        back = comm.all_to_all(ops.owner_reply(rec_owned, vids), rc, sc)
        gmap = ops.owner_map(map_r, pos_r, back)
        host = torch.stack([counts_out, counts_o]).tolist()     # the output sizes -- host sync 2 of 2
        if int(host[0][_lib.COUNT_STATUS]) & _lib.STATUS_BIN_OVERFLOW and not plain:
            return None        # some owner's merge outgrew a bucket (every rank reads the same word): all redo on the general path
        nvox, nown = int(host[0][_lib.COUNT_VOXELS]), int(host[1][_lib.COUNT_VOXELS])
        words = int(send.shape[1])
        self.last_stats = dict(
            exchange="owner", numbering="first-point bitmap", voxels=nvox, owned_voxels=nown, ranks=W,
            all_to_all_bytes_sent=(sum(sc) - sc[comm.rank]) * 4 * words, all_to_all_bytes_received=(sum(rc) - rc[comm.rank]) * 4 * words,
            reply_bytes_sent=(sum(rc) - rc[comm.rank]) * 8, all_reduce_bytes=int(gbits.numel()) * 8,
            all_gather_bytes_per_rank=nown * (8 + 24 + 4 + 4 * c) if self._replicate else 0)
        vids, coords, npoints, feats = vids[:nown], coords[:nown], npoints[:nown], feats[:nown]
        if self._debug_checks:
            # the status word behind the bitmap is a SUM over the ranks of 0 / 1 flags: anything else means the comm's
            # all_reduce is not an int64 SUM (the lock-step redo rests on every rank reading the same word)
            if int(gbits.numel()) == (max(n_total, 1) + 63) // 64 + 1:       # (HipOps: one status word behind the bitmap)
                assert 0 <= int(gbits[-1]) <= W, "all_reduce('sum') of the first-point bitmap did not add the ranks' status words"
        if self._debug_checks and nown > 1:
            fo = first_o[:nown]
            assert bool((fo[1:] > fo[:-1]).all()), "owned voxels are not ordered by their first point"
            assert bool((vids[1:] > vids[:-1]).all()), "global voxel ids of the owned voxels are not ascending"
            assert 0 <= int(fo[0]) and int(fo[-1]) < n_total, "first-point index outside the frame"
        if P:
            voxels, pmask = voxels[:nown], pmask[:nown].view(torch.bool)
            self.last_stats["rows_all_to_all_bytes_sent"] = (sum(rsc) - rsc[comm.rank]) * 16
        if not self._replicate:
            ret = Dict(coords=coords, voxel_npoints=npoints, aggregates=feats, voxel_ids=vids, points_mapping=gmap,
                       num_voxels=nvox)
            if P:
                ret.voxels, ret.voxel_pmask = voxels, pmask
            return ret
        sizes = comm.all_gather_int(nown, dev)
        packed = torch.cat([vids.view(-1, 1), coords, npoints.to(torch.int64).view(-1, 1)], 1)      # one int64 gather
        packed = comm.all_gather_var(packed.reshape(-1), [5 * k for k in sizes]).view(-1, 5)
        feats_all = comm.all_gather_var(feats.reshape(-1), [c * k for k in sizes]).view(-1, c)
        coords_f, cnt_f, feats_f = ops.owner_replicate(nvox, packed[:, 0].contiguous(), packed[:, 1:4].contiguous(),
                                                        packed[:, 4].to(torch.int32).contiguous(), feats_all, sizes=sizes)
        ret = Dict(coords=coords_f, voxel_npoints=cnt_f, aggregates=feats_f, points_mapping=gmap)
        if P:
            # the replicated dense tensor (V x max_points x 16 bytes on EVERY rank: a convenience for parity tests, sized by
            # the frame; production callers keep replicate=False): gather the owners' blocks, then place them by voxel id
            blocks = comm.all_gather_var(voxels.reshape(-1), [4 * P * k for k in sizes]).view(-1, P, 4)
            masks = comm.all_gather_var(pmask.view(torch.uint8).reshape(-1), [P * k for k in sizes]).view(-1, P)
            order = packed[:, 0]
            ret.voxels = torch.empty_like(blocks).index_copy_(0, order, blocks)
            ret.voxel_pmask = torch.empty_like(masks).index_copy_(0, order, masks).view(torch.bool)
        return ret

    def _run(self, points, plain):
        comm, ops = self._comm, self._ops
        dev = points.device
        n, c = points.shape
        offset, n_total, cap = self._layout(n, dev)
        mean = self._red == 1
        # 1. local hash + partial reduction; nothing is read back: rows >= V_r carry key -1, row n the status
        kw = {"plain": True} if plain else {}
        _, cnt_r, agg_r, first_r, map_r, keys_r, _ = ops.voxelize_reduce(
            points, self._shape, self._bounds, _SUM if mean else self._red, offset, **kw)
        nw = (self._ncells + 63) // 64
        # auto: bitmaps when the grid's bitmap is not larger than a key list (KITTI-size grids), else key lists
        bitmap_mode = self._exchange == "bitmap" or (self._exchange == "auto" and nw <= cap + 1)
        owned = False
        if bitmap_mode:
            # 2b. all-gather the ranks' occupancy bitmaps (+ status word); 3b. OR them: a streaming pass instead of
            #     one atomic per gathered key, and the keys of the slots fall out of the merged bitmap
            part = ops.bitmap_mark(keys_r, n, self._ncells)
            parts_all = comm.all_gather_var(part, [nw + 1] * comm.world)
            owned = comm.world <= _MAX_OWNER_WORLD
            handle, nvox, status = ops.compact_from_bitmaps(parts_all, comm.world, self._ncells,
                                                            rank=comm.rank if owned else None)
            owned = owned and nvox < (1 << 24)         # voxel ids travel in an fp32 column of the table
        else:
            # 2. all-gather the occupied-cell keys (+ status row), padded to the largest shard
            pad = keys_r if n == cap else torch.cat([keys_r[:n], keys_r.new_full((cap - n,), -1), keys_r[n:]])
            keys_all = comm.all_gather_var(pad, [cap + 1] * comm.world)
            # 3. identical compact slots on every rank; the one host read-back: global voxel count + every rank's status
            handle, nvox, status = ops.compact_index(keys_all, self._ncells, status_stride=cap + 1)
        if status & _lib.STATUS_TABLE_FULL:
            raise RuntimeError("voxelize_3d_reduce: internal hash table overflow")
        if status & (_lib.STATUS_PACK_OVERFLOW | _lib.STATUS_BIN_OVERFLOW) and not plain:
            return None        # some rank hit a capacity limit of the fast index (rare): all ranks redo on the general path
        # what this call moves between the ranks (bytes each rank contributes / receives; bench.py reports them)
        tcols = (c + 2 if bitmap_mode and owned else c + 1) if mean else (c + 1 if bitmap_mode and owned else c)
        self.last_stats = dict(
            exchange="bitmap" if bitmap_mode else "keys", numbering="ownership" if bitmap_mode and owned else "first-index",
            voxels=nvox, ranks=comm.world,
            all_gather_bytes_per_rank=8 * ((nw + 1) if bitmap_mode else (cap + 1)),
            all_reduce_bytes=nvox * (4 * tcols + (0 if mean else 4) + (0 if bitmap_mode and owned else 8)))
        # 4. all-reduce the compact voxel table
        if bitmap_mode and owned:
            # numbering by ownership: the lowest rank that has a cell holds its first point, so the voxel ids follow from
            # the bitmaps and every rank's local order; they ride in an extra column of the table -- no first-index exchange
            key_of_slot = ops.compact_keys(handle, nvox)
            table, cnt_t, slot_r = ops.build_table_owned(handle, keys_r[:n], n, nvox, c, self._red, agg_r, cnt_r, comm.rank)
            if nvox > 0:
                comm.all_reduce(table, "sum" if mean else ("max" if self._red == 2 else "min"))
                if cnt_t is not None:
                    comm.all_reduce(cnt_t, "sum")
            coords, out_cnt, out_feats, vid_of_slot = ops.finalize_owned(nvox, c, key_of_slot, table, mean, cnt_t, self._shape)
            gmap = ops.compose_map(map_r, slot_r, nvox, vid_of_slot)
            return Dict(coords=coords, voxel_npoints=out_cnt, aggregates=out_feats, points_mapping=gmap)
        if bitmap_mode:
            key_of_slot = ops.compact_keys(handle, nvox)
            table, cnt_t, first, _, slot_r = ops.build_table(handle, keys_r[:n], 0, n, nvox, c, self._red, agg_r, cnt_r,
                                                             first_r, want_keys=False)
        else:
            # the cells of the slots come out of the marked bitmap in one streaming pass; only this rank's own voxels need
            # the per-key lookup (looking all world x cap gathered keys up again cost 8 x the requests at world 8)
            key_of_slot = ops.compact_keys(handle, nvox)
            table, cnt_t, first, _, slot_r = ops.build_table(handle, keys_r[:n], 0, n, nvox, c, self._red, agg_r, cnt_r,
                                                             first_r, want_keys=False)
        if nvox > 0:
            comm.all_reduce(table, "sum" if mean else ("max" if self._red == 2 else "min"))
            if cnt_t is not None:
                comm.all_reduce(cnt_t, "sum")
            comm.all_reduce(first, "min")
        # 5. first-seen numbering: rank of each voxel's first point index among all first indices
        coords, out_cnt, out_feats, vid_of_slot = ops.finalize(nvox, c, first, n_total, key_of_slot, table, mean, cnt_t,
                                                               self._shape)
        gmap = ops.compose_map(map_r, slot_r, nvox, vid_of_slot)
        return Dict(coords=coords, voxel_npoints=out_cnt, aggregates=out_feats, points_mapping=gmap)


__all__ = ["ShardedVoxelGenerator", "TorchComm", "LocalComm", "HipOps", "voxelize_reduce"]
