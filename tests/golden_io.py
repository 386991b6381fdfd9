"""Loader for tests/golden/voxel_ref_cases.npz (written by tests/golden/make_voxel_golden.py)."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_voxel_cases():
    z = np.load(os.path.join(GOLDEN, "voxel_ref_cases.npz"))
    meta = json.loads(bytes(z["__meta__"]).decode())
    cases = {}
    for name, m in meta.items():
        out = {k.split("/out/")[1]: z[k] for k in z.files if k.startswith(name + "/out/")}
        cases[name] = dict(meta=m, cloud=z[name + "/cloud"], out=out,
                           size=z[name + "/size"] if (name + "/size") in z.files else None)
    return cases


def derived_pmask(npoints, max_points):
    return np.arange(max_points)[None, :] < np.minimum(npoints, max_points)[:, None]
