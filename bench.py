#!/usr/bin/env python3
"""bench.py -- throughput of the d3d voxel/box hot path on MI355X (contract: see README/DESIGN.md).

  python bench.py --gpus N --steps K --warmup W

N > 1 without a launcher (WORLD_SIZE unset): bench.py starts `python -m torch.distributed.run --nproc-per-node N` on
itself as a CHILD process before anything touches the GPU, relays rank 0's JSON line and exits with the child's code.

A "step" is one pass of the hot path over one batch of synthetic input already resident in HBM:
  N = 1 : BASELINE.json config 2 -- 1 M LiDAR-like KITTI-range points, 0.1 m voxels, max 32 points/voxel,
          dense contract + MEAN reduction through d3d_amd.voxel.VoxelGenerator (includes the one host read-back of
          the voxel count that the operator's variable-size return contract needs).
  N > 1 : BASELINE.json config 5's shards -- rank k holds points [k M, (k+1) M) of the 8 M-point Waymo-scale frame
          (0.05 m voxels, 3008 x 3008 x 120 grid); at N = 8 that is config 5 itself, at N = 2 / 4 the frame's first
          2 M / 4 M points (weak scaling: 1 M points per rank): local binned voxelization -> RCCL all-gather of the
          per-rank occupied-cell key lists -> global numbering -> RCCL all-reduce of the voxel feature grid
          (d3d_amd.voxel.sharded).
Rank 0 prints ONE JSON line.  `value` is whole-job Mpoints/s.  At N = 1 the line also carries `roofline` (dominant kernel,
timed with HIP events on its launch stream in a separate pass of the same K steps), `roofline_large` (the same fill kernel
on config 5's whole 8 M-point frame on one GPU: 3 GB of output, beyond the 256 MB Infinity Cache, next to the stream
bandwidth measured on the same box), `cpu_baseline` (the REAL reference voxelizer built from /root/reference into
oracle/_ref, or the C port when that binary is absent) and `extra` (uniform cloud, sparse contract, rotated IoU Mpairs/s,
NMS boxes/s, iou3d Mpairs/s).
"""
import argparse
import ctypes
import json
import os
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
REQ_PEAK_GS = 44.0      # scattered 8-byte requests per ns for the hash probe's mix (one coherent load + one CAS per
                        # thread), measured: tools/atomic_bench.hip (atomics alone: 23-26 per ns)


def sync():
    torch.cuda.synchronize()


def timed(fn, steps, warmup, barrier=None):
    for _ in range(warmup):
        fn()
    if barrier:
        barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    if barrier:
        barrier()
    sync()
    return time.perf_counter() - t0


def spread(fn, steps, reps=5):
    """the timed region again, `reps` more times: median / min / max ms per step (the headline `value` stays the FIRST
    region of exactly `steps` steps, as the contract asks)"""
    ms = sorted(1e3 * timed(fn, steps, 0) / steps for _ in range(reps))
    return dict(reps=reps, median_ms=round(ms[len(ms) // 2], 4), min_ms=round(ms[0], 4), max_ms=round(ms[-1], 4))


def kernel_profile(fn, steps):
    """per-kernel durations from HIP events recorded by the library on its launch stream"""
    from d3d_amd import _lib
    lib = _lib.load()
    lib.d3d_profile_enable.argtypes = [ctypes.c_int]
    lib.d3d_profile_report.argtypes = [ctypes.c_char_p, ctypes.c_size_t]
    sync()
    lib.d3d_profile_enable(1)
    for _ in range(steps):
        fn()
    sync()
    lib.d3d_profile_enable(0)
    buf = ctypes.create_string_buffer(1 << 16)
    lib.d3d_profile_report(buf, len(buf))
    out = {}
    for line in buf.value.decode().strip().splitlines():
        name, calls, ms = line.rsplit(",", 2)
        out[name] = dict(calls=int(calls), total_ms=float(ms), avg_us=1e3 * float(ms) / max(int(calls), 1))
    return out


TRAFFIC_SOURCES = {      # the kernel sources a workload's traffic figures depend on (tools/summarize_pmc.py hashes the same files)
    "voxel": ("voxel.hip", "common.hpp", "lds_sort.hpp"),
    "box": ("box.hip", "geom.hpp", "sort.hip", "common.hpp", "lds_sort.hpp"),
}


def source_hash(text):
    """hash of a kernel source's CODE: comments and white space do not count (a reworded comment must not retire the PMC passes)"""
    import hashlib
    import re
    code = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    code = re.sub(r"//[^\n]*", " ", code)
    return hashlib.sha1(" ".join(code.split()).encode()).hexdigest()[:16]


def source_hashes():
    d = os.path.join(ROOT, "d3d_amd", "csrc")
    return {f: source_hash(open(os.path.join(d, f), errors="replace").read())
            for fs in TRAFFIC_SOURCES.values() for f in fs if os.path.exists(os.path.join(d, f))}


def load_traffic(kernel, workload):
    """HBM bytes per launch from the COMMITTED rocprofv3 --pmc passes (profiles/traffic.json, made by
    tools/collect_profiles.sh + tools/summarize_pmc.py): PMC counters cannot be collected from inside this process, so
    the figure is the one of the last profiled build, not of this run -> (bytes | None, source label).  The file carries the
    hashes of the kernel sources it was collected on: when one of the files behind `workload` has changed since, the figure is
    withheld (None, "stale ...") instead of being passed off as this build's."""
    p = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        t = json.load(open(p))
        group = "box" if workload.startswith(("config3", "config4")) else "voxel"
        then, now = t.get("_sources", {}).get(workload, {}), source_hashes()
        changed = [f for f in TRAFFIC_SOURCES[group] if then.get(f) != now.get(f)]
        if changed:
            return None, "stale: %s changed since profiles/traffic.json's %s collection (%s)" % (
                ", ".join(changed), workload, t.get("_profile", {}).get(workload, "unlabelled") if isinstance(t.get("_profile"), dict)
                else t.get("_profile", "unlabelled"))
        prof = t.get("_profile", "unlabelled")
        if isinstance(prof, dict):
            prof = prof.get(workload, "unlabelled")
        return t.get(workload, {}).get(kernel), "profiles/traffic.json (committed rocprofv3 --pmc passes, %s)" % prof
    except Exception:
        return None, "unavailable"


def event_overhead_us(iters=50):
    """HIP-event duration of an EMPTY launch (d3d_stream_probe mode 6) behind a busy stream: what the event pair of d3d_profile_*
    adds to every kernel's figure on top of the dispatch's own begin-to-end time that rocprofv3 reports"""
    from d3d_amd import _lib
    lib = _lib.load()
    buf = torch.empty((1 << 20,), dtype=torch.uint8, device="cuda")

    def run():
        _lib.check(lib.d3d_stream_probe(0, _lib.ptr(buf), buf.numel(), _lib.stream_ptr()), "stream_probe")      # (a launch before it,
        _lib.check(lib.d3d_stream_probe(6, _lib.ptr(buf), buf.numel(), _lib.stream_ptr()), "stream_probe")      #  as inside an operator)
    prof = kernel_profile(run, iters)
    return round(prof["k_probe_empty"]["avg_us"], 2)


def stream_probe(nbytes, iters=5):
    """achievable stream bandwidth of THIS box (GB/s): nontemporal stores, copy, read sweep over an nbytes buffer --
    the access patterns of the HBM-bound kernels with the work stripped off (d3d_stream_probe, api.hip)"""
    from d3d_amd import _lib
    lib = _lib.load()
    buf = torch.empty((nbytes,), dtype=torch.uint8, device="cuda")
    out = {}
    for mode, name, moved in ((0, "store_nt", nbytes), (4, "store_nt_chunked", nbytes), (5, "store_nt_dealt", nbytes),
                              (1, "copy", nbytes // 32 * 32), (2, "read", nbytes), (3, "memset", nbytes)):
        run = lambda: _lib.check(lib.d3d_stream_probe(mode, _lib.ptr(buf), nbytes, _lib.stream_ptr()), "stream_probe")  # noqa: E731
        run()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters):
            run()
        b.record()
        b.synchronize()
        out[name] = round(moved * iters / (a.elapsed_time(b) * 1e-3) / 1e9, 1)
    del buf
    torch.cuda.empty_cache()
    return out


def emit_bytes(n, V, P):
    """ALGORITHMIC bytes of the dense contract per SURVEY 8(d) -- compulsory operator I/O only, no scratch: the point tensor
    read once (n x 16 B) and every output written (voxels[V,P,4], voxel_pmask, coords, voxel_npoints, aggregates).  k_emit is
    the launch that moves them (it reads the rows it stores straight from the point tensor)."""
    return n * 16 + V * (P * 16 + P + 24 + 4 + 16)


def dense_last_plan():
    """what the last dense call launched: (split, voxels whose zero padding was stored under the index launches, bytes of zeros
    stored by k_tile_sort's fillers, by k_first_count's)"""
    from d3d_amd import _lib
    out = (ctypes.c_int64 * 4)()
    _lib.load().d3d_voxelize_dense_last_plan(out)
    return tuple(int(x) for x in out)


def emit_split_bytes(n, P, npoints, plan):
    """ALGORITHMIC bytes of k_emit_split: the dense contract's 8(d) bytes (emit_bytes) minus the zero padding that the filler
    workgroups of the index launches stored and this launch therefore does not: for a voxel id below plan[1], every row from
    min(P, 8 * ceil(min(count, P) / 8)) on (a voxel of 255 points and more: none) -- computed from the call's own voxel_npoints"""
    V = int(npoints.shape[0])
    pre = min(plan[1], V)
    cnt = npoints[:pre].to(torch.int64)
    lim = torch.clamp(((torch.clamp(cnt, max=P) + 7) // 8) * 8, max=P)
    lim = torch.where(cnt >= 255, torch.full_like(lim, P), lim)
    skipped = int(((P - lim) * 16).sum())
    return emit_bytes(n, V, P) - skipped, skipped


def emit_scratch_bytes(npad, V, kept, multi_rows):
    """what k_emit reads on top of that from the index's scratch: one 4-byte first-point entry per point index (numbering +
    count + segment of the voxel, round 5) and one 4-byte ranked index per kept row that is not a voxel's first point"""
    return npad * 4 + multi_rows * 4


def large_frame_leg(steps=5):
    """config 5's whole 8 M-point frame through the dense operator on ONE GPU: 5.9 M voxels -> voxels[V,32,4] = 3 GB, far
    beyond the 256 MB Infinity Cache, so k_fill_c4's rate here is an HBM rate (at config 2 the 300 MB output partly drains
    through the cache).  Priced against the 8 TB/s spec AND the store / copy bandwidth measured on this box."""
    from d3d_amd import synth
    from d3d_amd.voxel import VoxelGenerator
    n, P = 8000000, 32
    cloud = torch.from_numpy(synth.lidar_like(n, 3, synth.WAYMO_BOUNDS)).cuda()
    gen = VoxelGenerator(synth.WAYMO_BOUNDS, synth.WAYMO_SHAPE, dense=True, reduction="mean", max_points=P, max_voxels=n)
    res = gen(cloud)
    V = int(res.coords.shape[0])
    kept = int(torch.clamp(res.voxel_npoints, max=P).sum())
    plan = dense_last_plan()
    b_split = emit_split_bytes(n, P, res.voxel_npoints, plan)[0]
    del res
    step = lambda: gen(cloud)  # noqa: E731
    dt = timed(step, steps, 2)
    prof = kernel_profile(step, steps)
    del cloud, gen
    torch.cuda.empty_cache()
    npad = -(-n // 16384) * 16384
    kern = "k_emit_split" if "k_emit_split" in prof else "k_emit" if "k_emit" in prof else "k_fill_c4"
    b_alg = b_split if kern == "k_emit_split" else emit_bytes(n, V, P) if kern == "k_emit" else V * P * 16 + kept * 16 + V * 16
    us = prof[kern]["avg_us"]
    ach = b_alg / (us * 1e-6) / 1e9
    probe = stream_probe(3 << 30)
    traffic, src = load_traffic(kern, "config5_1gpu")
    return dict(bound="hbm", kernel=kern, workload="config 5 frame on one GPU: 8 M LiDAR-like points, 0.05 m voxels "
                "(3008x3008x120), dense+MEAN, max 32 pts/voxel", voxels=V, achieved=round(ach, 1), peak=HBM_PEAK_GBS,
                unit="GB/s", frac=round(ach / HBM_PEAK_GBS, 4), peak_measured=probe,
                frac_of_measured_store=round(ach / probe["store_nt"], 4),
                # k_emit's own store pattern (every wavefront through a stretch of its own) without any of its loads
                frac_of_measured_chunked_store=round(ach / probe["store_nt_chunked"], 4),
                # the fill moves reads as well as writes: against the best streaming rate of either kind measured on this box
                frac_of_measured_best=round(ach / max(probe.values()), 4), avg_us=round(us, 2), algorithmic_bytes=b_alg,
                scratch_bytes=emit_scratch_bytes(npad, V, kept, kept - V) if kern in ("k_emit", "k_emit_split") else None,
                traffic=traffic, traffic_source=src, op_ms=round(1e3 * dt / steps, 3),
                op_mpoints_per_s=round(n * steps / dt / 1e6, 1),
                kernels_us={k: round(v["avg_us"], 2) for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["total_ms"])})


def cpu_baseline_voxel(cloud, bounds, shape, max_points, max_voxels):
    """reference voxelizer (oracle/_ref, built from /root/reference) on the host, 1 core (it is sequential)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    kind, fn = None, None
    try:
        from oracle.build_ref import load_ref
        ref = load_ref()
        if ref is not None:
            pts = torch.from_numpy(cloud)
            sh = torch.tensor(shape, dtype=torch.int32)
            bd = torch.tensor(bounds, dtype=torch.float)
            kind = "reference"
            fn = lambda: ref.voxelize_3d_dense(pts, sh, bd, max_points, max_voxels, ref.ReductionType.MEAN)  # noqa: E731
    except Exception as e:   # pragma: no cover
        print("cpu_baseline: reference binary unavailable (%s), using the C port" % e, file=sys.stderr)
    if fn is None:
        import oracle
        kind = "port"
        fn = lambda: oracle.voxelize_3d_dense(cloud, shape, bounds, max_points, max_voxels, 1)  # noqa: E731
    torch.set_num_threads(1)
    t0 = time.perf_counter()
    fn()
    dt = time.perf_counter() - t0
    runs = 1
    while dt < 10.0 and runs < 64:     # bounded sample: about 10 s of CPU work on one core
        t1 = time.perf_counter()
        fn()
        dt += time.perf_counter() - t1
        runs += 1
    n = cloud.shape[0]
    return dict(value=round(n * runs / dt / 1e6, 4), unit="Mpoints/s", cores=1, kind=kind,
                sample="%d run(s) of the full %d-point cloud, dense+MEAN, max_points=%d" % (runs, n, max_points))


def cpu_allcore_voxel(cloud, bounds, shape, max_points, max_voxels):
    """the same sequential algorithm on every host core at once, one whole frame per thread (the reference has no
    intra-frame parallelism, voxelize.cpp:94, so frame-parallel is the only all-core form): C port through ctypes
    (releases the GIL); at most 32 threads (512 MB of output buffers each)."""
    import threading
    import oracle
    cores = max(1, min(os.cpu_count() or 1, 32))
    oracle.lib()
    run = lambda: oracle.voxelize_3d_dense(cloud, shape, bounds, max_points, max_voxels, 1)  # noqa: E731
    ts = [threading.Thread(target=run) for _ in range(cores)]
    t0 = time.perf_counter()
    [t.start() for t in ts]
    [t.join() for t in ts]
    dt = time.perf_counter() - t0
    return dict(value=round(cloud.shape[0] * cores / dt / 1e6, 4), unit="Mpoints/s", cores=cores, kind="port",
                sample="one full %d-point frame per thread on %d threads at once (frame-parallel)" % (cloud.shape[0], cores))


def iou_leg(n3, steps=2, warmup=1):
    """config 3: n3 rotated boxes fp64, all n3^2 pairs in ONE result (80 GB at 100 k boxes; 288 GB of HBM; row blocks only
    beyond 100 GB) through the public operator, and the roofline of its dominant kernel (8 B per pair, write-bound)"""
    from d3d_amd import synth
    from d3d_amd.box import box2d_iou
    ex = {}
    b, _ = synth.boxes2d_sparse(n3, 1)
    bt = torch.from_numpy(b).cuda()
    rows = max(1, min(n3, int(100e9 // (8 * n3))))

    def all_pairs():
        for r0 in range(0, n3, rows):
            box2d_iou(bt[r0:r0 + rows], bt, method="rbox")       # the public operator (fp64 in, precise=True)
    dt = timed(all_pairs, steps, warmup)
    ex["iou2d_rbox_fp64_mpairs_per_s"] = round(n3 * n3 * steps / dt / 1e6, 1)
    ex["iou2d_rbox_fp64_GBps_written"] = round(n3 * n3 * 8 * steps / dt / 1e9, 1)
    if rows >= n3:      # HIP events on the launch stream
        prof = kernel_profile(all_pairs, steps)
        kp = prof.get("k_iou_pre")
        if kp:
            b_alg = n3 * n3 * 8 + 2 * n3 * 5 * 8
            ach = b_alg / (kp["avg_us"] * 1e-6) / 1e9
            tr, src = load_traffic("k_iou_pre", "config3_iou")
            ex["iou2d_rbox_fp64_roofline"] = dict(bound="hbm", kernel="k_iou_pre", achieved=round(ach, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                                                  frac=round(ach / HBM_PEAK_GBS, 4), avg_us=round(kp["avg_us"], 1),
                                                  algorithmic_bytes=b_alg, traffic=tr, traffic_source=src,
                                                  op_frac=round(n3 * n3 * 8 * steps / dt / 1e9 / HBM_PEAK_GBS, 4),
                                                  kernels_us={k: round(v["avg_us"], 2) for k, v in prof.items()})
            # the stream probes at THIS kernel's footprint (an 80 GB buffer: the 3 GB probes of roofline_large undersell a
            # fill that runs for 13 ms over every channel of the 288 GB) -- no fraction of a measured rate may exceed 1
            del bt
            torch.cuda.empty_cache()
            pr = stream_probe(n3 * n3 * 8, iters=2)
            fb = ach / max(pr.values())
            ex["iou2d_rbox_fp64_roofline"].update(peak_measured=pr, frac_of_measured_best=round(fb, 4))
            if fb > 1:      # the probes are FLOORS of what the box can store (their loops are not this kernel's): say so
                ex["iou2d_rbox_fp64_roofline"]["probe_note"] = ("the kernel's fill ran faster than every store probe on this box: the "
                                                               "probes bound the achievable rate from below, not from above; "
                                                               "traffic (WRITE_SIZE pass of the same launch) is the evidence")
    torch.cuda.empty_cache()
    return ex


def nms_leg(n3, steps=20, warmup=1):
    """config 3: box2d_nms (rbox, fp64, threshold 0.5) on n3 boxes.  SURVEY 8(d): compulsory I/O is tiny (boxes + scores in, one
    byte per box out); the reference design's figure is the N x ceil(N / 64) x 8 B suppression matrix written and read by the
    sweep (nms_cuda.cu:17-110) -- this build does not materialise it (grid broad phase -> candidate lists -> resolve), so the
    roofline object prices the launch sequence on the bytes it DOES move (PMC passes) and says what bounds it instead."""
    from d3d_amd import synth
    from d3d_amd.box import box2d_nms
    b, s = synth.boxes2d_sparse(n3, 1)
    bt, st = torch.from_numpy(b).cuda(), torch.from_numpy(s).cuda()
    step = lambda: box2d_nms(bt, st, iou_method="rbox", iou_threshold=0.5)  # noqa: E731
    dt = timed(step, steps, warmup)
    ex = {"nms_rbox_fp64_boxes_per_s": round(n3 * steps / dt, 1)}
    prof = kernel_profile(step, steps)
    ksum = sum(v["total_ms"] for v in prof.values()) / steps * 1e3
    launches = sum(v["calls"] for v in prof.values()) / steps
    tr, src = load_traffic("nms_op_total", "config3_nms")
    compulsory = n3 * (5 + 1) * 8 + n3
    us = 1e6 * dt / steps
    ex["nms_rbox_fp64_roofline"] = dict(
        bound="hbm", kernel="d3d_nms2d (all launches of a call)", achieved=round((tr or compulsory) / (us * 1e-6) / 1e9, 2), peak=HBM_PEAK_GBS,
        unit="GB/s", frac=round((tr or compulsory) / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 5), us_per_call=round(us, 1),
        kernel_sum_us=round(ksum, 1), launches_per_call=round(launches, 1), algorithmic_bytes=compulsory, traffic=tr, traffic_source=src,
        reference_mask_bytes=n3 * ((n3 + 63) // 64) * 8,
        note="achieved = PMC traffic of one call (or, without it, the compulsory bytes) / the call's duration: the operator is bound by "
             "its chain of ~%d dependent launches over a few MB, not by bandwidth; reference_mask_bytes is what nms_cuda.cu's dense "
             "bit matrix would write (and the sweep read) -- not moved here" % round(launches),
        kernels_us={k: round(v["avg_us"], 2) for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["total_ms"])[:8]})
    del bt, st
    torch.cuda.empty_cache()
    return ex


def iou3d_leg(steps=20, warmup=3):
    """config 4: 20 k predictions x 5 k ground truths, rbox iou3d fp32 (4 B per pair written + 7 floats per box read)"""
    from d3d_amd import synth
    from d3d_amd.box import iou3d
    p, g = synth.boxes3d_eval(5000, 4, 2)
    pt, gt = torch.from_numpy(p).cuda(), torch.from_numpy(g).cuda()
    step = lambda: iou3d(pt, gt)  # noqa: E731
    dt = timed(step, steps, warmup)
    npairs = len(p) * len(g)
    ex = {"iou3d_rbox_fp32_mpairs_per_s": round(npairs * steps / dt / 1e6, 1)}
    prof = kernel_profile(step, steps)
    b_alg = npairs * 4 + (len(p) + len(g)) * 7 * 4
    us = 1e6 * dt / steps
    dom = max(prof.items(), key=lambda kv: kv[1]["total_ms"])
    tr, src = load_traffic("iou3d_op_total", "config4_iou3d")
    ex["iou3d_rbox_fp32_roofline"] = dict(
        bound="hbm", kernel="d3d_iou3d_forward (all launches of a call)", achieved=round(b_alg / (us * 1e-6) / 1e9, 1), peak=HBM_PEAK_GBS,
        unit="GB/s", frac=round(b_alg / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4), us_per_call=round(us, 2), algorithmic_bytes=b_alg,
        traffic=tr, traffic_source=src, dominant_kernel=dom[0],
        dominant_kernel_frac=round(b_alg / (dom[1]["avg_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
        kernels_us={k: round(v["avg_us"], 2) for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["total_ms"])})
    del pt, gt
    torch.cuda.empty_cache()
    return ex


def extras(args):
    """secondary metrics of BASELINE.json (configs 2-sparse, 3, 4); short runs, GPU + bounded CPU samples"""
    import oracle
    from d3d_amd import synth
    from d3d_amd.box import box2d_iou, box2d_nms, iou3d
    from d3d_amd.voxel import VoxelGenerator
    ex = {}
    ncpu = os.cpu_count() or 1
    # config 2, sparse contract + trim
    cloud = torch.from_numpy(synth.lidar_like(args.points, 0)).cuda()
    gen = VoxelGenerator(synth.KITTI_BOUNDS, synth.KITTI_SHAPE, max_points=32, max_points_filter="trim")
    res = gen(cloud)
    nkept, vkept = int(res.points.shape[0]), int(res.coords.shape[0])
    del res
    dt = timed(lambda: gen(cloud), 20, 3)
    ex["voxelize_sparse_trim_mpoints_per_s"] = round(args.points * 20 / dt / 1e6, 2)
    # the reference's calling convention -- CPU tensor in, CPU tensors out (its voxelizer is a CPU operator): the PCIe-inclusive rate
    # (never `value`): staged in, computed on the device, results back through the pinned host allocator (_lib.to_caller)
    cloud_h = cloud.cpu()
    gen_d = VoxelGenerator(synth.KITTI_BOUNDS, synth.KITTI_SHAPE, dense=True, reduction="mean", max_points=32, max_voxels=args.points)
    pcie = {}
    for key, g in (("dense_mean", gen_d), ("sparse_trim", gen)):
        for _ in range(4):           # (the pinned allocator's cache fills: two result sets are alive at a time in this loop)
            r = g(cloud_h)
        t0 = time.perf_counter()
        for _ in range(3):
            r = g(cloud_h)
        dth = (time.perf_counter() - t0) / 3
        pcie[key] = dict(ms_per_call=round(1e3 * dth, 2), mpoints_per_s=round(args.points / dth / 1e6, 1),
                         bytes_back=int(sum(v.numel() * v.element_size() for v in r.values() if torch.is_tensor(v))))
        del r
    ex["voxelize_cpu_tensors_in_and_out_pcie_inclusive"] = pcie
    del cloud_h, gen_d
    # compulsory bytes of sparse + filter = N C 4 in + N' (C 4 + 8 + 8) + V' (24 + 4) out (62 B/point at config 2).
    # No single kernel of this path is HBM-bound (ten latency- / request-bound launches over 62 MB): the whole-operator rate
    # against the peak is the figure, the per-kernel durations say where the time goes.
    sp = kernel_profile(lambda: gen(cloud), 20)
    b_sp = args.points * 16 + nkept * 32 + vkept * 28
    ex["roofline_sparse"] = dict(
        bound="hbm", workload="config 2, sparse contract + max_points_filter=trim (the reference's default mode)",
        kept_points=nkept, kept_voxels=vkept, algorithmic_bytes=b_sp, ms_per_step=round(1e3 * dt / 20, 4),
        achieved=round(b_sp * 20 / dt / 1e9, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(b_sp * 20 / dt / 1e9 / HBM_PEAK_GBS, 4),
        kernels_us={k: round(v["avg_us"] * v["calls"] / 20, 2) for k, v in sorted(sp.items(), key=lambda kv: -kv[1]["total_ms"])},
        kernels_sum_us=round(sum(v["total_ms"] for v in sp.values()) * 1e3 / 20, 1),
        traffic=load_traffic("sparse_op_total", "config2_sparse")[0], traffic_source=load_traffic("sparse_op_total", "config2_sparse")[1])
    del cloud
    # SURVEY 8d "batched variant": a stream of config-2 frames through VoxelGenerator.stream -- frame k + 1's index launches on a
    # side stream under frame k's output launch (two frames in flight).  An EXTRA: the headline stays the sequential operator.
    ca = torch.from_numpy(synth.lidar_like(args.points, 0)).cuda()
    cb = torch.from_numpy(synth.lidar_like(args.points, 1)).cuda()
    gen = VoxelGenerator(synth.KITTI_BOUNDS, synth.KITTI_SHAPE, dense=True, reduction="mean", max_points=32, max_voxels=args.points)
    nfr = 60

    def run_stream():
        nv = 0
        for r in gen.stream(((ca if k & 1 else cb) for k in range(nfr)), pipelined=True):
            nv += r.coords.shape[0]
        return nv
    run_stream()
    sync()
    t0 = time.perf_counter()
    run_stream()
    sync()
    dtp = time.perf_counter() - t0
    ex["voxelize_dense_pipelined_mpoints_per_s"] = round(args.points * nfr / dtp / 1e6, 2)
    ex["voxelize_dense_pipelined_us_per_frame"] = round(dtp / nfr * 1e6, 1)
    ex["voxelize_dense_pipelined_note"] = ("two frames in flight on two streams: no gain on this stack -- one hardware queue serialises "
                                           "them, two queues pay 50-60 us per cross-queue event wait (DESIGN.md 4d); off by default")
    del gen
    # the dense contract into a RESIDENT output (d3d_voxelize_3d_dense_resident): voxels[V,P,4] comes back as a view
    # of a buffer the generator keeps on the device, of which a frame stores only the rows that hold points and zeros over the
    # previous frame's -- same values as the headline's fresh tensor, bit for bit, without re-writing 95 % zero padding.  An EXTRA
    # (a different output contract: the result is valid until the next call); alternating frames, so that every voxel id changes hands.
    gen = VoxelGenerator(synth.KITTI_BOUNDS, synth.KITTI_SHAPE, dense=True, reduction="mean", max_points=32, max_voxels=args.points,
                         resident=True)
    kfr = [0]

    def resident_step():
        kfr[0] += 1
        return gen(ca if kfr[0] & 1 else cb)
    dtr = timed(resident_step, 40, 4)
    rp = kernel_profile(resident_step, 20)
    ex["voxelize_dense_resident_mpoints_per_s"] = round(args.points * 40 / dtr / 1e6, 2)
    ex["voxelize_dense_resident_us_per_frame"] = round(dtr / 40 * 1e6, 1)
    ex["voxelize_dense_resident_kernels_us"] = {k: round(v["avg_us"], 2) for k, v in sorted(rp.items(), key=lambda kv: -kv[1]["total_ms"])}
    ex["voxelize_dense_resident_traffic"] = load_traffic("k_emit_resident", "config2_resident")[0]
    ex["voxelize_dense_resident_note"] = ("output buffer resident on the device: rows with points + stale rows stored, zero padding kept "
                                          "(same values as the headline's tensor; valid until the next call); not the headline contract")
    del ca, cb, gen
    torch.cuda.empty_cache()
    # config 2 on the UNIFORM cloud (SURVEY 8d's worst case: ~0.98 voxels per point, 500 MB of voxels[V,32,4])
    cloud = torch.from_numpy(synth.uniform_cloud(args.points, 0)).cuda()
    gen = VoxelGenerator(synth.KITTI_BOUNDS, synth.KITTI_SHAPE, dense=True, reduction="mean", max_points=32, max_voxels=args.points)
    vu = int(gen(cloud).coords.shape[0])
    dt = timed(lambda: gen(cloud), 10, 2)
    ex["voxelize_dense_uniform_mpoints_per_s"] = round(args.points * 10 / dt / 1e6, 2)
    ex["voxelize_dense_uniform_voxels"] = vu
    ex["voxelize_dense_uniform_op_algorithmic_GBps"] = round((args.points * 16 + vu * (32 * 16 + 32 + 24 + 4 + 16)) * 10 / dt / 1e9, 1)
    del cloud, gen
    torch.cuda.empty_cache()
    n3 = args.boxes
    ex.update(iou_leg(n3))
    b, s = synth.boxes2d_sparse(n3, 1)
    bt, st = torch.from_numpy(b).cuda(), torch.from_numpy(s).cuda()
    torch.cuda.empty_cache()
    bd, _ = synth.boxes2d_dense(5000, 1)      # the reference's own benchmark distribution (ALU-bound case)
    bdt = torch.from_numpy(bd).cuda()
    dt = timed(lambda: box2d_iou(bdt, bdt, method="rbox"), 10, 2)
    ex["iou2d_rbox_fp64_dense5k_mpairs_per_s"] = round(25e6 * 10 / dt / 1e6, 1)
    # box2d_iou's default form on fp32 boxes (precise=True): fp64 arithmetic with an fp32 matrix (D3D_F64_M32) against the chain the
    # reference's Python layer spells out -- boxes.double(), fp64 kernels, ious.to(float32) (box/__init__.py:204-205, 224)
    from d3d_amd.box import Iou2DR
    b32 = torch.from_numpy(b[:20000].astype(np.float32)).cuda()
    dt = timed(lambda: box2d_iou(b32, b32, method="rbox"), 5, 1)
    dtc = timed(lambda: Iou2DR.apply(b32.double(), b32.double()).to(torch.float32), 5, 1)
    # call latency at a detector's sizes, fp32 tensors through the default (precise=True) forms: D3D_F32_WIDE, no cast launches
    bsm = torch.from_numpy(synth.boxes2d_dense(2000, 7)[0].astype(np.float32)).cuda()
    ssm = torch.from_numpy(synth.boxes2d_dense(2000, 7)[1].astype(np.float32)).cuda()
    lat = {}
    for key, fn in (("box2d_iou_100x50", lambda: box2d_iou(bsm[:100], bsm[100:150], method="rbox")),
                    ("box2d_iou_1000x200", lambda: box2d_iou(bsm[:1000], bsm[1000:1200], method="rbox")),
                    ("box2d_nms_500", lambda: box2d_nms(bsm[:500], ssm[:500], iou_method="rbox", iou_threshold=0.3)),
                    ("box2d_nms_2000", lambda: box2d_nms(bsm, ssm, iou_method="rbox", iou_threshold=0.3))):
        lat[key] = round(min(1e6 * timed(fn, 200, 5) / 200 for _ in range(3)), 1)      # (best of three: an allocator stall lands in one)
    from d3d_amd.box import box3dp_crop
    pts_c = torch.rand((120000, 3), device="cuda") * 100
    box_c = torch.from_numpy(synth.boxes3d_eval(50, 1, 2)[1]).cuda()
    lat["box3dp_crop_120k_points_x_50_boxes"] = round(min(1e6 * timed(lambda: box3dp_crop(pts_c, box_c), 200, 5) / 200 for _ in range(3)), 1)
    del pts_c, box_c
    ex["call_latency_fp32_tensors_us"] = lat
    del bsm, ssm
    ex["box2d_iou_precise_fp32_boxes_20kx20k_ms"] = dict(fp32_matrix=round(1e3 * dt / 5, 3), fp64_matrix_then_cast=round(1e3 * dtc / 5, 3))
    del b32
    torch.cuda.empty_cache()
    # (no warm-up streak: since round 4 every call decides from ITS OWN grid whether the level kernels are launched)
    ex.update(nms_leg(n3, 20, 1))
    try:        # the same operator captured into a HIP graph by the caller and replayed (~28 launches without host work)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            box2d_nms(bt, st, iou_method="rbox", iou_threshold=0.5)
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            box2d_nms(bt, st, iou_method="rbox", iou_threshold=0.5)
        dt = timed(graph.replay, 20, 2)
        ex["nms_rbox_fp64_boxes_per_s_graph_replay"] = round(n3 * 20 / dt, 1)
        del graph
    except Exception as e:      # capture is a convenience of the caller's runtime, not of the library
        ex["nms_rbox_fp64_boxes_per_s_graph_replay"] = "unavailable: %s" % type(e).__name__
    del bt, st
    torch.cuda.empty_cache()
    # a detector's top-k: 2000 boxes in clusters of 50 around 40 objects (the small-set path), and the bare score argsort
    rng = np.random.default_rng(7)
    cen = np.stack([rng.random(40) * 400, rng.random(40) * 400, rng.random(40) * 20 + 10, rng.random(40) * 20 + 10, rng.random(40) * 6.28], 1)
    bk = torch.from_numpy(np.repeat(cen, 50, 0) + rng.normal(0, 1, (2000, 5)) * [1.5, 1.5, 1.0, 1.0, 0.05]).cuda()
    sk = torch.from_numpy(rng.random(2000)).cuda()
    dt = timed(lambda: box2d_nms(bk, sk, iou_method="rbox", iou_threshold=0.5), 50, 5)
    ex["nms_rbox_fp64_topk2000_clustered_us_per_call"] = round(dt / 50 * 1e6, 1)
    # a detector's RAW output: 100 k boxes in clusters around the objects (tools/nms_cluster_profile.py; the general path with
    # its level kernels, which every call enqueues or not by its own grid's density)
    for nobj, per in ((200, 500), (1000, 100)):
        rc = np.random.default_rng(1)
        cc = np.stack([rc.random(nobj) * 2000, rc.random(nobj) * 2000, rc.random(nobj) * 20 + 10, rc.random(nobj) * 20 + 10,
                       rc.random(nobj) * 6.28], 1)
        bc = torch.from_numpy(np.repeat(cc, per, 0) + rc.normal(0, 1, (nobj * per, 5)) * [1.5, 1.5, 1.0, 1.0, 0.05]).cuda()
        sc = torch.from_numpy(rc.random(nobj * per)).cuda()
        dt = timed(lambda: box2d_nms(bc, sc, iou_method="rbox", iou_threshold=0.5), 10, 3)
        ex["nms_rbox_fp64_clusters_%dx%d_ms" % (nobj, per)] = round(dt / 10 * 1e3, 3)
        if per == 500:      # ONE call on clusters right after calls on scattered boxes (rounds 2-3 guessed from history: 2.9 ms here)
            bs_, ss_ = synth.boxes2d_sparse(20000, 9)
            bst, sst = torch.from_numpy(bs_).cuda(), torch.from_numpy(ss_).cuda()
            for _ in range(6):
                box2d_nms(bst, sst, iou_method="rbox", iou_threshold=0.5)
                torch.cuda.synchronize()
            dt1 = timed(lambda: box2d_nms(bc, sc, iou_method="rbox", iou_threshold=0.5), 1, 0)
            ex["nms_rbox_fp64_clusters_200x500_first_call_after_sparse_streak_ms"] = round(dt1 * 1e3, 3)
            del bst, sst
        del bc, sc
    from d3d_amd.box import argsort_desc
    s100 = torch.from_numpy(np.random.default_rng(1).random(n3)).cuda()
    dt = timed(lambda: argsort_desc(s100), 50, 5)
    ex["argsort_desc_fp64_100k_us_per_call"] = round(dt / 50 * 1e6, 1)
    del bk, sk, s100
    # config 4: 20k x 5k iou3d fp32
    p, g = synth.boxes3d_eval(5000, 4, 2)
    pt, gt = torch.from_numpy(p).cuda(), torch.from_numpy(g).cuda()
    ex.update(iou3d_leg(20, 3))
    # config 4 as the evaluator uses it: [n,9] ingress + clip + 1 - riou, then ONE score-ordered association
    # for all 40 score thresholds of DetectionEvaluator.calc_stats (the reference re-sorts the matrix per threshold)
    from d3d_amd.benchmarks import DetectionEvaluator
    from d3d_amd.tracking import DistanceTypes, prepare_boxes
    rng = np.random.default_rng(5)
    gt9 = np.concatenate([rng.integers(1, 3, (len(g), 1)), np.zeros((len(g), 1)), g], 1).astype(np.float32)
    dt9 = np.concatenate([np.repeat(gt9[:, :1], 4, axis=0), rng.random((len(p), 1)), p], 1).astype(np.float32)
    dt9t, gt9t = torch.from_numpy(dt9).cuda(), torch.from_numpy(gt9).cuda()
    dt = timed(lambda: prepare_boxes(dt9t, gt9t, DistanceTypes.RIoU), 20, 3)
    ex["match_distance_riou_fp32_mpairs_per_s"] = round(1e8 * 20 / dt / 1e6, 1)
    # (the default association is the reference's own: one per score threshold with matcher.pyx:142-162's pairing; reference_compat=False:
    # every detection its own nearest ground truths, ONE association for all 40 thresholds -- INTEGRATION.md 5)
    for key, compat in (("evaluator_calc_stats_20kx5k_ms", True), ("evaluator_calc_stats_20kx5k_one_association_ms", False)):
        ev = DetectionEvaluator([1, 2], [0.7, 0.5], reference_compat=compat)
        ev.calc_stats(gt9, dt9)
        t0 = time.perf_counter()
        for _ in range(3):
            ev.calc_stats(gt9, dt9)
        sync()
        ex[key] = round((time.perf_counter() - t0) / 3 * 1e3, 2)
        # a frame's worth -- 200 detections x 50 ground truths: what an evaluation over a dataset calls thousands of times
        fg, fd = gt9[:50], dt9[:200]
        ev.calc_stats(fg, fd)
        t0 = time.perf_counter()
        for _ in range(20):
            ev.calc_stats(fg, fd)
        sync()
        ex[key.replace("20kx5k", "200x50_frame")] = round((time.perf_counter() - t0) / 20 * 1e3, 3)
    del dt9t, gt9t
    # loss path: GIoU / DIoU have a value for EVERY pair: 8 B/pair written + the hull (diameter) of the two
    # rectangles per pair -- round 5: a pair kernel for the boxes that are apart, the pairs that need the clip listed and done one
    # per lane; fp64, 10 k x 10 k of config 3's boxes; forward + backward on 2 k x 2 k
    nl = 10000
    bl = torch.from_numpy(b[:nl]).cuda()
    for method in ("grbox", "drbox"):
        dt = timed(lambda: box2d_iou(bl, bl, method=method), 5, 1)
        ex["iou2d_%s_fp64_mpairs_per_s" % method] = round(nl * nl * 5 / dt / 1e6, 1)
    b1g, b2g = bl[:2000].clone().requires_grad_(True), bl[2000:4000].clone().requires_grad_(True)

    def fwd_bwd():
        b1g.grad = b2g.grad = None
        box2d_iou(b1g, b2g, method="grbox").sum().backward()
    dt = timed(fwd_bwd, 5, 1)
    ex["iou2d_grbox_fp64_fwd_bwd_2kx2k_ms"] = round(dt / 5 * 1e3, 3)

    def fwd_bwd_d():
        b1g.grad = b2g.grad = None
        box2d_iou(b1g, b2g, method="drbox").sum().backward()
    dt = timed(fwd_bwd_d, 5, 1)
    ex["iou2d_drbox_fp64_fwd_bwd_2kx2k_ms"] = round(dt / 5 * 1e3, 3)
    # rotated IoU backward where almost no pair overlaps (config 3's density, 20 k x 20 k, a weight on every pair): marks, then the
    # marked pairs compacted globally (the dense regime is the reference's protocol below)
    from d3d_amd.box import iou2dr_backward
    if len(b) >= 40000:
        gsp = torch.ones((20000, 20000), dtype=torch.float64, device="cuda")
        sp1, sp2 = torch.from_numpy(b[:20000]).cuda(), torch.from_numpy(b[20000:40000]).cuda()
        dt = timed(lambda: iou2dr_backward(sp1, sp2, gsp), 5, 1)
        ex["iou2dr_backward_fp64_20kx20k_sparse_ms"] = round(dt / 5 * 1e3, 3)
        del gsp, sp1, sp2
    # the reference's OWN IoU benchmark protocol (test/compare/benchmark_riou.py:53-118): n x n rotated IoU of fp32 boxes
    # through box2d_iou (precise=True: fp64 inside), forward, then .sum().backward() into boxes2, for n = 1 ... 5000; boxes as
    # there (centres in +-5, sizes in [0, 5), angles in +-5 rad: 28 % of the pairs overlap).  Times with a device
    # synchronisation on both sides (the reference's script stops its clock without one).
    proto = {}
    rngp = np.random.default_rng(11)
    for n_p, rep in ((1, 50), (1, 200), (2, 200), (5, 100), (10, 100), (20, 50), (50, 20), (100, 10), (200, 10), (500, 5), (1000, 5),
                     (2000, 5), (5000, 5)):
        tf = tb = 0.0
        for _ in range(rep):
            mk = lambda: np.stack([(rngp.random(n_p) - 0.5) * 10, (rngp.random(n_p) - 0.5) * 10, rngp.random(n_p) * 5,  # noqa: E731
                                   rngp.random(n_p) * 5, (rngp.random(n_p) - 0.5) * 10], 1).astype(np.float32)
            p1, p2 = torch.from_numpy(mk()).cuda(), torch.from_numpy(mk()).cuda().requires_grad_(True)
            sync(); t0 = time.perf_counter()
            r = box2d_iou(p1, p2, method="rbox")
            sync(); t1 = time.perf_counter()
            r.sum().backward()
            sync(); t2 = time.perf_counter()
            tf += t1 - t0; tb += t2 - t1
        proto[str(n_p)] = dict(forward_ms=round(1e3 * tf / rep, 4), backward_ms=round(1e3 * tb / rep, 4))   # the first n = 1 round
    ex["reference_riou_protocol_ms"] = proto                                                               # is the warm-up
    # ALU rooflines of the clip-bound kernels: VALU counters cannot be read from inside this process, so the
    # figures are those of the committed rocprofv3 --pmc collection (tools/alu_roofline.sh -> profiles/alu_roofline.json)
    try:
        alu = json.load(open(os.path.join(ROOT, "profiles", "alu_roofline.json")))
        ex["roofline_alu"] = {
            k: dict(bound="valu_fp64", achieved=v["achieved_Tlaneops"], peak_fp64=v["peak_fp64_Tlaneops"], unit="T lane-operations/s",
                    frac=v["frac"], valu_busy=v["valu_busy"], valu_insts_per_wave=v["valu_insts_per_wave"], avg_us=v["duration_us"],
                    source="profiles/alu_roofline.json (%s)" % alu.get("_profile", "unlabelled"))
            for k, v in alu.items() if isinstance(v, dict) and "frac" in v}
    except Exception:
        ex["roofline_alu"] = None
    from d3d_amd.box import pdist2dr_forward
    pts2 = torch.rand((1000000, 2), dtype=torch.float32, device="cuda") * 3000
    bx32 = bl[:2000].to(torch.float32)
    dt = timed(lambda: pdist2dr_forward(pts2, bx32), 5, 1)          # 2000 boxes x 1 M points: 8 GB of distances + 2 GB of iedge
    ex["pdist2dr_fp32_mpairs_per_s"] = round(2000 * 1e6 * 5 / dt / 1e6, 1)
    del pts2, bx32, bl, b1g, b2g
    torch.cuda.empty_cache()
    if not args.skip_cpu:
        k = 3000
        t0 = time.perf_counter(); oracle.iou2d_forward(b[:k], b[:k], "rbox", nthreads=ncpu); dt = time.perf_counter() - t0
        ex["cpu_iou2d_rbox_fp64_mpairs_per_s"] = dict(value=round(k * k / dt / 1e6, 2), cores=ncpu, kind="port",
                                                       sample="%dx%d pairs of config 3" % (k, k))
        k = 20000
        b2, s2 = synth.boxes2d_sparse(k, 1)
        t0 = time.perf_counter(); oracle.box2d_nms(b2, s2, iou_method="rbox", iou_threshold=0.5); dt = time.perf_counter() - t0
        ex["cpu_nms_rbox_fp64_boxes_per_s"] = dict(value=round(k / dt, 1), cores=1, kind="port",
                                                    sample="%d boxes at config 3's density" % k)
        t0 = time.perf_counter(); oracle.iou3d(p[:2000], g, "rbox", nthreads=1); dt = time.perf_counter() - t0
        ex["cpu_iou3d_rbox_fp32_mpairs_per_s"] = dict(value=round(2000 * 5000 / dt / 1e6, 2), cores=1, kind="port",
                                                       sample="2000x5000 pairs of config 4")
        # the evaluator as the reference runs it (one association PER threshold, python restatement of the Cython loops)
        sub_g, sub_d = gt9[:100], dt9[:400]
        t0 = time.perf_counter(); oracle.calc_stats(sub_g, sub_d, [1, 2], {1: 0.3, 2: 0.5}, ev.score_thresholds); dt = time.perf_counter() - t0
        ex["cpu_evaluator_calc_stats_ms"] = dict(value=round(dt * 1e3, 1), cores=1, kind="port",
                                                  sample="400 detections x 100 ground truths, 40 thresholds (python loops)")
    return ex


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--points", type=int, default=1000000, help="points per GPU")
    ap.add_argument("--boxes", type=int, default=100000, help="boxes of config 3 (extra metrics)")
    ap.add_argument("--skip-cpu", action="store_true")
    ap.add_argument("--skip-extra", action="store_true")
    ap.add_argument("--dist", choices=["lidar", "uniform"], default="lidar")
    ap.add_argument("--force-sharded", action="store_true", help="run the sharded (RCCL) path even with one rank")
    ap.add_argument("--skip-large", action="store_true", help="skip the 8 M-point beyond-cache roofline leg")
    ap.add_argument("--large-only", action="store_true", help="run only that leg and print it (rocprofv3 --pmc passes)")
    ap.add_argument("--sparse-only", action="store_true", help="run only config 2's sparse + trim operator (rocprofv3 --pmc passes)")
    ap.add_argument("--iou-only", action="store_true", help="run only config 3's box2d_iou (one n x n rbox fp64 launch) and print its leg")
    ap.add_argument("--nms-only", action="store_true", help="run only config 3's box2d_nms and print its leg")
    ap.add_argument("--iou3d-only", action="store_true", help="run only config 4's iou3d (20 k x 5 k fp32) and print its leg")
    ap.add_argument("--master-port", type=int, default=29533, help="rendezvous port when bench.py launches the ranks itself")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # bare `python bench.py --gpus N`: one rank per GPU under torch.distributed.run, started as a CHILD process --
        # nothing in this process has touched the GPU yet (importing torch does not), and this process never will
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(args.master_port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        raise SystemExit(subprocess.call(cmd, env=env))
    if args.gpus != world:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))

    from d3d_amd import synth
    from d3d_amd.voxel import VoxelGenerator

    torch.cuda.set_device(local_rank)
    if args.iou_only or args.nms_only or args.iou3d_only:
        leg = iou_leg(args.boxes, max(1, min(args.steps, 2)), 1) if args.iou_only else \
            nms_leg(args.boxes, args.steps, args.warmup) if args.nms_only else iou3d_leg(args.steps, args.warmup)
        print(json.dumps(leg))
        return
    if args.large_only:
        print(json.dumps({"roofline_large": large_frame_leg()}))
        return
    if args.sparse_only:
        cloud = torch.from_numpy(synth.lidar_like(args.points, 0)).cuda()
        gen = VoxelGenerator(synth.KITTI_BOUNDS, synth.KITTI_SHAPE, max_points=32, max_points_filter="trim")
        dt = timed(lambda: gen(cloud), args.steps, args.warmup)
        print(json.dumps({"sparse_trim_ms_per_step": round(1e3 * dt / args.steps, 4)}))
        return
    barrier = None
    sharded = world > 1 or args.force_sharded
    if sharded:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(args.master_port))
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        barrier = dist.barrier

    P, n = 32, args.points
    out = {}
    if not sharded:
        mk = synth.lidar_like if args.dist == "lidar" else synth.uniform_cloud
        cloud_h = mk(n, 0)                      # config 2's cloud
        cloud = torch.from_numpy(cloud_h).cuda()
        gen = VoxelGenerator(synth.KITTI_BOUNDS, synth.KITTI_SHAPE, dense=True, reduction="mean", max_points=P,
                             max_voxels=n)
        res = gen(cloud)
        V = int(res.coords.shape[0])
        kept = int(torch.clamp(res.voxel_npoints, max=P).sum())
        plan = dense_last_plan()
        split_bytes, split_skipped = emit_split_bytes(n, P, res.voxel_npoints, plan)
        del res
        step = lambda: gen(cloud)  # noqa: E731
        dt = timed(step, args.steps, args.warmup)
        value = n * args.steps / dt / 1e6
        out["timed_region_repeats"] = spread(step, args.steps)
        workload = "config2: %d %s points, KITTI range, 0.1 m voxels (704x800x40), dense+MEAN, max 32 pts/voxel" % (
            n, "LiDAR-like" if args.dist == "lidar" else "uniform")
        prof = kernel_profile(step, args.steps)
        dom = max(prof.items(), key=lambda kv: kv[1]["total_ms"])
        # compulsory bytes of each kernel per launch (DESIGN.md "kernels"); C = 4
        npad = -(-n // 16384) * 16384
        algo = {
            # voxels[V,P,4] written; kept rows (16 B) gathered from the staged segments, one 16-byte record per voxel read
            "k_fill_c4": V * P * 16 + kept * 16 + V * 16,
            # binned index (n >= 32 k points): partition, per-bucket index in LDS, numbering + per-voxel outputs
            "k_tile_sort": n * 16 + n * 8 + npad * 4,               # rows read; {cell, index} entries + firstmap reset written
            "k_bin_count": n * 16 + npad * 8,                       # rows read; bucket word + firstmap reset written
            "k_bin_scatter": n * (16 + 4) + n * (16 + 4),           # rows + bucket words read; rows + indices written
            "k_bucket_index": n * (16 + 4) + kept * 16 + V * (16 + 4),   # bucket read; ranked rows, records, firstmap written
            "k_meta_first": npad * 4 + V * 16 + kept * 16 + V * (16 + 24 + 4 + P + 16),
            "k_emit": emit_bytes(n, V, P),                          # 16 B/point in + every output (no scratch)
            # round 6: the same minus the zero padding stored under the index launches (k_tile_sort's / k_first_count's fillers)
            "k_emit_split": split_bytes,
            # hash-table index (other inputs)
            "k_insert": n * 16 + n * 8 + n * 8,          # points read, pslot+arrival written, one 8-byte slot touched
            "k_scatter": n * 8 + n * 8 + n * 4 + n * 8,  # pslot+arrival read, aux read, index written, (cnt,base) written
            "k_select": n * 8 + kept * 4,
            "k_init": None,
        }
        name = dom[0]
        b_alg = algo.get(name) or (n * 16)
        ach = b_alg / (dom[1]["avg_us"] * 1e-6) / 1e9
        out["roofline"] = dict(bound="hbm", kernel=name, achieved=round(ach, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                               frac=round(ach / HBM_PEAK_GBS, 4), traffic=load_traffic(name, "config2")[0],
                               traffic_source=load_traffic(name, "config2")[1], avg_us=round(dom[1]["avg_us"], 2), algorithmic_bytes=b_alg,
                               timing="HIP events on the launch stream, separate pass of the same %d steps" % args.steps,
                               cache_note="config 2's 345 MB of outputs partly drain through the 256 MB Infinity Cache: this fraction "
                                          "is cache-assisted; the HBM claim is roofline_large.frac (same kernel, 3.3 GB of outputs)",
                               bytes_note="algorithmic_bytes = SURVEY 8(d): 16 B/point read + every output of the dense contract "
                                          "written; scratch_bytes (index entries the launch also reads) are not part of it")
        if name == "k_emit_split":
            out["roofline"].update(
                algorithmic_bytes_operator=emit_bytes(n, V, P), zero_padding_stored_under_index_launches=split_skipped,
                filler_bytes=dict(k_tile_sort=plan[2], k_first_count=plan[3]), prefilled_voxels=plan[1],
                split_note="k_emit_split moves the dense contract's 8(d) bytes EXCEPT the zero padding of voxel ids below prefilled_voxels: "
                           "filler workgroups on the CUs k_tile_sort / k_first_count leave idle store those (plus the rows' lines of the same "
                           "voxels, which this launch overwrites: filler_bytes > zero_padding_stored).  The operator-level figure on ALL 8(d) "
                           "bytes is roofline_operator")
        ov = event_overhead_us()
        out["roofline"].update(
            event_overhead_us=ov, avg_us_net=round(dom[1]["avg_us"] - ov, 2),
            frac_net=round(b_alg / ((dom[1]["avg_us"] - ov) * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
            event_note="avg_us = HIP events recorded around the launch on its stream; an EMPTY launch measures event_overhead_us that way "
                       "(the first event waits for the previous launch, the dispatch follows).  rocprofv3's per-dispatch durations (profiles/"
                       "*_kernel_stats_config2_only.csv) are begin-to-end of the dispatch and compare with avg_us_net; `frac` stays on avg_us")
        if name in ("k_emit", "k_emit_split"):
            # what the launch reads from the index's scratch on top of the 8(d) bytes, and the stream probes at THIS launch's
            # footprint (the outputs' 345 MB, partly absorbed by the 256 MB Infinity Cache exactly as the kernel's are): the
            # store forms with the work stripped off, on the same box in the same process
            out["roofline"]["scratch_bytes"] = emit_scratch_bytes(npad, V, kept, kept - V)
            pr = stream_probe(V * (P * 16 + P + 24 + 4 + 16))
            out["roofline"].update(peak_measured=pr, frac_of_measured_store=round(ach / pr["store_nt"], 4),
                                   frac_of_measured_chunked_store=round(ach / pr["store_nt_chunked"], 4),
                                   frac_of_measured_best=round(ach / max(pr.values()), 4),
                                   probe_note="probes on a buffer of the outputs' size (V x 588 B): floors of what the box stores in "
                                              "that form, not ceilings")
        if name in ("k_bin_scatter", "k_bucket_index"):
            # limited by scattered 4..16-byte stores, not bytes: measured ceiling ~80 G/s (profiles/r01_g_atomic_bench.txt)
            req = 2 * n if name == "k_bin_scatter" else kept + V
            out["roofline"]["requests"] = dict(per_launch=req, achieved_G_per_s=round(req / dom[1]["avg_us"] / 1e3, 2),
                                               measured_peak_G_per_s=80.0, frac=round(req / dom[1]["avg_us"] / 1e3 / 80.0, 3))
        if name == "k_insert":
            # the kernel's real limiter: scattered 8-byte requests (>= one coherent probe load + one atomic per point);
            # ceiling measured with tools/atomic_bench.hip on MI355X (profiles/r01_g_atomic_bench.txt); 2 per point is a lower
            # bound (linear probing adds ~0.2 loads, dense voxels serialise on one slot)
            req = 2 * n
            out["roofline"]["requests"] = dict(per_launch=req, achieved_G_per_s=round(req / dom[1]["avg_us"] / 1e3, 2),
                                               measured_peak_G_per_s=REQ_PEAK_GS,
                                               frac=round(req / dom[1]["avg_us"] / 1e3 / REQ_PEAK_GS, 3))
        if "k_fill_c4" in prof and name != "k_fill_c4":      # the HBM-bound kernel of the op, priced the same way
            f_us = prof["k_fill_c4"]["avg_us"]
            f_ach = algo["k_fill_c4"] / (f_us * 1e-6) / 1e9
            out["roofline_streaming"] = dict(bound="hbm", kernel="k_fill_c4", achieved=round(f_ach, 1), peak=HBM_PEAK_GBS,
                                             unit="GB/s", frac=round(f_ach / HBM_PEAK_GBS, 4),
                                             traffic=load_traffic("k_fill_c4", "config2")[0], avg_us=round(f_us, 2),
                                             algorithmic_bytes=algo["k_fill_c4"])
        out["kernels_us"] = {k: round(v["avg_us"], 2) for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["total_ms"])}
        out["op_algorithmic_GBps"] = round((n * 16 + V * (P * 16 + P + 24 + 4 + 16)) * args.steps / dt / 1e9, 1)
        # the whole operator (every launch of a step) on the 8(d) bytes: what a caller sees
        out["roofline_operator"] = dict(bound="hbm", achieved=out["op_algorithmic_GBps"], peak=HBM_PEAK_GBS, unit="GB/s",
                                        frac=round(out["op_algorithmic_GBps"] / HBM_PEAK_GBS, 4), algorithmic_bytes=emit_bytes(n, V, P),
                                        us_per_step=round(1e6 * dt / args.steps, 2))
        out["voxels"] = V
        parallelism = "1 GPU"
    else:
        from d3d_amd.voxel.sharded import LocalComm, ShardedVoxelGenerator
        # config 5: rank k holds points [k n, (k+1) n) of the frame (world = 8, n = 1 M: the 8 M-point frame itself)
        frame = synth.lidar_like(world * n, 3, synth.WAYMO_BOUNDS)
        cloud_h = np.ascontiguousarray(frame[rank * n:(rank + 1) * n])
        cloud = torch.from_numpy(cloud_h).cuda()
        # strong-scaling bases on ONE GPU, whole world x n frame: (a) the same CONTRACT without an exchange -- what one would
        # actually run on one GPU: sharded.voxelize_reduce (local index + feature grid, nothing packed / merged / replied) --
        # and (b) the same OPERATOR with a world of one (the full pack / merge / number / reply machinery on one rank)
        whole = best = None
        if rank == 0:
            from d3d_amd.voxel.sharded import voxelize_reduce
            fr = torch.from_numpy(frame).cuda()
            k = max(args.steps // 4, 3)
            best = 1e3 * timed(lambda: voxelize_reduce(fr, synth.WAYMO_SHAPE, synth.WAYMO_BOUNDS, "mean"), k, 2) / k
            solo = ShardedVoxelGenerator(synth.WAYMO_BOUNDS, synth.WAYMO_SHAPE, reduction="mean", comm=LocalComm(), replicate=False)
            whole = 1e3 * timed(lambda: solo(fr), k, 2) / k
            del solo, fr
            torch.cuda.empty_cache()
        del frame
        # ... and on this rank's shard alone (the base of the weak-scaling curve)
        solo = ShardedVoxelGenerator(synth.WAYMO_BOUNDS, synth.WAYMO_SHAPE, reduction="mean", comm=LocalComm(), replicate=False)
        dt_solo = timed(lambda: solo(cloud), max(args.steps // 2, 3), 2)
        solo_ms = 1e3 * dt_solo / max(args.steps // 2, 3)
        del solo
        # the scalable form: owner-computes exchange, every rank keeps its owned 1/world of the voxels (replicate=False)
        gen = ShardedVoxelGenerator(synth.WAYMO_BOUNDS, synth.WAYMO_SHAPE, reduction="mean", replicate=False)
        step = lambda: gen(cloud)  # noqa: E731
        out["voxels_global"] = int(step().num_voxels)
        st = dict(gen.last_stats)
        # what RCCL itself saw: an all-reduce of ones over the group, and the distinct devices behind the ranks
        ones = torch.ones((1,), dtype=torch.int64, device="cuda")
        dist.all_reduce(ones)
        ids = [None] * world
        props = torch.cuda.get_device_properties(local_rank)
        dist.all_gather_object(ids, str(getattr(props, "uuid", "")) or "%s#%d" % (props.name, local_rank))
        out["rccl_ranks"] = int(ones.item())
        out["rccl_distinct_devices"] = len(set(ids))
        out["exchange"] = st["exchange"]
        out["numbering"] = st["numbering"]
        out["collectives_per_step"] = dict(
            {k: v for k, v in st.items() if "bytes" in k}, size_all_gather_bytes=8, count_matrix_all_gather_bytes=8 * (2 * world + 1),
            note="per rank: all-to-all of the partial voxel records to the cells' owner ranks (a sparse reduce-scatter of the "
                 "feature grid), SUM all-reduce of the one-bit-per-point first-point bitmap (= OR of disjoint sets), all-to-all "
                 "of the voxel ids back; no all-gather of the grid (every rank keeps its owned 1/world of the voxels)")
        dt_local = timed(step, args.steps, args.warmup, barrier)
        # the same with the final all-gather of the finished grid (replicated result, the frame-sized part)
        gen_rep = ShardedVoxelGenerator(synth.WAYMO_BOUNDS, synth.WAYMO_SHAPE, reduction="mean", replicate=True)
        k = max(args.steps // 4, 3)
        dt_rep = timed(lambda: gen_rep(cloud), k, 2, barrier) / k
        del gen_rep
        # the N = 1 line's own workload on every rank at once -- config 2, dense + MEAN, each rank ITS OWN frame, no collective: the
        # data-parallel use of the operator (a training job voxelizes one batch per GPU) and the curve that is comparable with N = 1
        gen2 = VoxelGenerator(synth.KITTI_BOUNDS, synth.KITTI_SHAPE, dense=True, reduction="mean", max_points=P, max_voxels=n)
        cloud2 = torch.from_numpy(synth.lidar_like(n, rank)).cuda()
        dt_frames = timed(lambda: gen2(cloud2), args.steps, args.warmup, barrier)
        del gen2, cloud2
        t = torch.tensor([dt_local, solo_ms, dt_rep, dt_frames], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t[0].item())
        out["independent_frames_per_rank"] = dict(
            ms_per_step=round(1e3 * float(t[3].item()) / args.steps, 4), mpoints_per_s=round(n * world * args.steps / float(t[3].item()) / 1e6, 2),
            note="config 2 (dense + MEAN, the workload of the N = 1 line) on every rank's own %d-point frame at once, no collective, max "
                 "over ranks: the data-parallel use of the operator; `value` above is the point-sharded single frame of config 5" % n)
        value = n * world * args.steps / dt / 1e6
        out["same_operator_one_rank"] = dict(ms_per_step=round(float(t[1].item()), 4),
                                             mpoints_per_s=round(n / float(t[1].item()) / 1e3, 2),
                                             note="the same operator on one rank's shard without collectives (max over ranks)")
        if whole is not None:
            out["single_gpu_best"] = dict(
                ms_per_step=round(best, 4), mpoints_per_s=round(n * world / best / 1e3, 2),
                note="sharded.voxelize_reduce on the WHOLE %d-point frame on one GPU: the same contract (feature grid + point -> "
                     "voxel map) with no exchange at all -- the honest single-GPU base of the speed-up" % (n * world))
            out["speedup_vs_single_gpu_best"] = round(best / (1e3 * dt / args.steps), 3)
            out["single_gpu_whole_frame"] = dict(
                ms_per_step=round(whole, 4), mpoints_per_s=round(n * world / whole / 1e3, 2),
                note="the same OPERATOR (pack / merge / number / reply, world of one) on the whole frame on one GPU: an upper "
                     "bound of what one GPU needs, not what one would run there")
            out["speedup_vs_single_gpu_whole_frame"] = round(whole / (1e3 * dt / args.steps), 3)
        out["replicated_result"] = dict(ms_per_step=round(1e3 * float(t[2].item()), 4),
                                        note="with replicate=True: + all-gather of the owners' finished rows and the scatter into "
                                             "voxel-id order on every rank (sized by the frame, not the shard)")
        workload = ("config5 shards: rank k = points [k*%d, (k+1)*%d) of a %d-point LiDAR-like Waymo-range frame (seed 3), "
                    "0.05 m voxels (3008x3008x120), MEAN feature grid: local voxelization + RCCL all-to-all of the partial voxel "
                    "records to the cells' owners + bitmap all-reduce for the first-seen numbering; every rank returns its owned "
                    "voxels with their global ids and the global voxel id of each of its points" % (n, n, world * n))
        parallelism = "points sharded over %d GPUs, voxels owned by hash(cell) %% %d" % (world, world)

    if rank == 0:
        line = {
            "metric": "voxelize_mpoints_per_s", "value": round(value, 2), "unit": "Mpoints/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload, "points_per_gpu": n, "parallelism": parallelism},
        }
        line.update(out)
        if not sharded:
            del cloud, gen
            torch.cuda.empty_cache()
        if not sharded and not args.skip_large:
            line["roofline_large"] = large_frame_leg()
        if not sharded and not args.skip_cpu:
            line["cpu_baseline"] = cpu_baseline_voxel(cloud_h, synth.KITTI_BOUNDS, synth.KITTI_SHAPE, P, n)
            line["cpu_baseline_all_cores"] = cpu_allcore_voxel(cloud_h, synth.KITTI_BOUNDS, synth.KITTI_SHAPE, P, n)
        if not sharded and not args.skip_extra:
            line["extra"] = extras(args)
        print(json.dumps(line))
    if sharded:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
