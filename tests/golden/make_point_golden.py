"""Generate tests/golden/point_ref_cases.npz by running the REAL reference aligned_scatter
(/root/reference/d3d/point/{impl,scatter}.cpp built by oracle/build_ref.py) in this container.  Data only."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))


def main():
    from oracle.build_ref import build, load_ref
    assert build(name="point_impl") is not None
    ref = load_ref("point_impl")
    torch.set_num_threads(1)
    # NB the reference's CPU aligned_scatter_backward never runs its kernel (the dispatch lambda at
    # scatter.cpp:196-206 is constructed but not invoked): "bwd" below is therefore all zeros; it is stored to
    # document that, the intended semantics are pinned by tests/test_point.py through the adjoint identity.
    rng = np.random.default_rng(77)
    out = {}
    k = 0
    for dtype in (np.float32, np.float64):
        for dim, dims in ((1, (7,)), (2, (5, 6)), (3, (3, 4, 5)), (3, (3, 3, 3))):
            B, C, n = 2, 5, 300
            img = rng.random((B, C) + dims).astype(dtype)
            coord = np.empty((n, dim + 1), dtype)
            coord[:, 0] = rng.integers(0, B, n)
            span = np.array(dims, np.float64)
            coord[:, 1:] = (rng.random((n, dim)) * (span + 1.0) - 0.5).astype(dtype)      # some outside both ends
            coord[:20, 1:] = np.round(coord[:20, 1:])                                      # exact integers (weight quirk)
            grad = rng.random((n, C)).astype(dtype)
            for at, name in ((ref.AlignType.MEAN, "mean"), (ref.AlignType.LINEAR, "linear")):
                fwd = ref.aligned_scatter_forward(torch.from_numpy(coord), torch.from_numpy(img), at).numpy()
                ig = torch.zeros(img.shape, dtype=torch.from_numpy(img).dtype)
                ref.aligned_scatter_backward(torch.from_numpy(coord), torch.from_numpy(grad), at, ig)
                p = "c%d" % k
                out[p + "/coord"], out[p + "/img"], out[p + "/grad"] = coord, img, grad
                out[p + "/fwd"], out[p + "/bwd"] = fwd, ig.numpy()
                out[p + "/atype"] = np.array([1 if name == "mean" else 2])
                k += 1
    # the reference's own test vectors (test/test_point.py:11-13)
    out["t/coord"] = np.array([[0, 0.25, 0.25, 0.25], [0, 1.25, 1.25, 1.25], [1, 2.25, 2.25, 2.25]], np.float32)
    np.savez_compressed(os.path.join(HERE, "point_ref_cases.npz"), **out)
    print("wrote", k, "cases")


if __name__ == "__main__":
    main()
