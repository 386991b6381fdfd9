"""pipelined frames under rocprofv3 --kernel-trace: do frame k + 1's index kernels overlap frame k's k_emit?
usage (GPU box): cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/trace -o st -- python3 $GRAFT_REPO_ROOT/tools/stream_trace.py run
       then:     python3 tools/stream_trace.py parse gpurun_out/trace"""
import sys
if sys.argv[1] == "run":
    import torch
    sys.path.insert(0, __file__.rsplit("/", 2)[0])
    from d3d_amd import synth
    from d3d_amd.voxel import VoxelGenerator
    n = 1000000
    ca = torch.from_numpy(synth.lidar_like(n, 0)).cuda()
    cb = torch.from_numpy(synth.lidar_like(n, 1)).cuda()
    gen = VoxelGenerator(synth.KITTI_BOUNDS, synth.KITTI_SHAPE, dense=True, reduction="mean", max_points=32, max_voxels=n)
    for rep in range(2):
        print(sum(r.coords.shape[0] for r in gen.stream(((ca if k & 1 else cb) for k in range(12)), pipelined=True)))
    torch.cuda.synchronize()
else:
    import csv
    import glob
    f = sorted(glob.glob(sys.argv[2] + "/**/*kernel_trace.csv", recursive=True))[-1]
    rows = [r for r in csv.DictReader(open(f)) if any(k in r["Kernel_Name"] for k in ("k_emit", "k_tile_sort", "k_bucket_index", "k_first_count"))]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    t0 = int(rows[0]["Start_Timestamp"])
    for r in rows[-24:]:
        name = r["Kernel_Name"].split("(")[0].split("::")[-1].split("<")[0]
        print("%-16s queue %s  start %9.1f us  end %9.1f us  (%.1f)" % (name, r.get("Queue_Id", "?"), (int(r["Start_Timestamp"]) - t0) / 1e3,
              (int(r["End_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
