"""Known-answer vectors carried by the reference's own tests for the box path
(reference test/test_box.py:12-123; test/test_benchmark.py:45-71).  Data only."""
import numpy as np

sq2 = np.sqrt(2)
d90 = np.pi / 4      # [sic] name from test_box.py:8
eps = 1e-3

AA_B1 = np.array([[1, 1, 2, 2, eps], [2, 2, 2, 2, eps], [3, 3, 2, 2, eps]], np.float32)
AA_B2 = np.array([[3, 1, 2, 2, -eps], [2, 2, 2, 2, -eps], [1, 3, 2, 2, -eps]], np.float32)
AA_EXPECTED = np.array([[0, 1 / 7, 0], [1 / 7, 1, 1 / 7], [0, 1 / 7, 0]], np.float32)   # box atol eps, rbox 4eps

ROT_B1 = np.array([[0, 0, 2, 2, 0], [-1, 1, 2, 2, 0], [1, 1, 2, 2, 0]], np.float32)
ROT_B2 = np.array([[-1, 1, 2 * sq2 - eps, 2 * sq2 - eps, d90 - eps], [1, 1, sq2 + eps, sq2 + eps, d90 + eps]], np.float32)
ROT_BOX_EXPECTED = np.array([[1 / 4, 1 / 7], [1 / 4, 0], [1 / 9, 1]], np.float32)      # atol 2eps
ROT_RBOX_EXPECTED = np.array([[1 / 5, 1 / 11], [1 / 2, 0], [1 / 11, 1 / 2]], np.float32)  # atol 4eps

APART_BOX = np.array([[1, 2, 3, 3, 0], [-2, 1, 3, 3, 0], [-1, -2, 3, 3, 0], [2, -1, 3, 3, 0]], np.float32)
APART_RBOX = np.array([[0, 0, 2, 2, 0], [2, 2, 2 * sq2, 2 * sq2, d90 + eps], [-2, 2, 2 * sq2, 2 * sq2, d90 + 2 * eps],
                       [2, -2, 2 * sq2, 2 * sq2, d90 + 3 * eps], [-2, -2, 2 * sq2, 2 * sq2, d90 + 4 * eps]], np.float32)

NMS_BOXES = np.array([[1, 1, 2 - 10 * eps, 2 - 10 * eps, 0], [2, 2, 2 - 10 * eps, 2 - 10 * eps, eps],
                      [3, 3, 2 - 10 * eps, 2 - 10 * eps, 2 * eps], [3, 1, 1, 2, 3 * eps], [4, 2, 1, 2, 4 * eps],
                      [5, 3, 1, 2, 5 * eps]], np.float32)
NMS_SCORES = np.array([0.5, 0.3, 0.4, 0.4, 0.2, 0.1], np.float32)
NMS_EXPECTED = np.array([True, False, True, True, False, True])

SOFT_BOXES = np.array([[1, 1, 2, 2, 0], [2, 2, 2, 2, 0], [3, 3, 2, 2, 0], [3, 1, 1, 1, 0], [4, 2, 1, 1, 0],
                       [5, 3, 1, 1, 0]], np.float32)

# test_benchmark.py:45-71: dt1 (0,0,0; 2,2,2; yaw 0) vs gt2 (-1,1,0; 2.1^3; yaw 0.01): acc_iou > 0.1
EVAL_DT = np.array([[0, 0, 0, 2, 2, 2, 0.0]], np.float32)
EVAL_GT = np.array([[-1, 1, 0, 2.1, 2.1, 2.1, 0.01]], np.float32)
EVAL_IOU = 0.14369   # SURVEY.md App. C, independent fp64 clip


def random_boxes_like_reference(n, seed):
    """test_box.py:125-131"""
    rng = np.random.default_rng(seed)
    return np.stack([rng.random(n) * 200, rng.random(n) * 400, rng.random(n) * 20 + 10, rng.random(n) * 30 + 5,
                     rng.random(n) * 2 - 1], 1).astype(np.float32), rng.random(n).astype(np.float32)
