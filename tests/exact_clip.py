"""Exact-rational convex clipping (python fractions): an independent pin for the geometry.
Corners are taken as the float64 values an implementation would compute; the intersection area of
those exact polygons is then computed without rounding (Sutherland-Hodgman over Fractions)."""
from fractions import Fraction as F
import math


def corners(x, y, w, h, r):
    s, c = math.sin(r), math.cos(r)
    dxs, dxc, dys, dyc = w * s / 2, w * c / 2, h * s / 2, h * c / 2
    return [(x - dxc + dys, y - dxs - dyc), (x + dxc + dys, y + dxs - dyc),
            (x + dxc - dys, y + dxs + dyc), (x - dxc - dys, y - dxs + dyc)]


def area(poly):
    s = F(0)
    for k in range(len(poly)):
        (ax, ay), (bx, by) = poly[k], poly[(k + 1) % len(poly)]
        s += ax * by - bx * ay
    return s / 2


def clip(subj, clipper):
    out = [(F(px), F(py)) for px, py in subj]
    cl = [(F(px), F(py)) for px, py in clipper]
    for e in range(len(cl)):
        if not out:
            break
        (ax, ay), (bx, by) = cl[e], cl[(e + 1) % len(cl)]
        ex, ey = bx - ax, by - ay
        inp, out = out, []
        for k in range(len(inp)):
            (px, py), (qx, qy) = inp[k], inp[(k + 1) % len(inp)]
            dp = ex * (py - ay) - ey * (px - ax)
            dq = ex * (qy - ay) - ey * (qx - ax)
            if dp >= 0:
                out.append((px, py))
            if (dp >= 0) != (dq >= 0):
                t = dp / (dp - dq)
                out.append((px + t * (qx - px), py + t * (qy - py)))
    return out


def iou_exact(b1, b2):
    p1, p2 = corners(*map(float, b1)), corners(*map(float, b2))
    inter = area(clip(p1, p2)) if True else 0
    a1, a2 = area([(F(x), F(y)) for x, y in p1]), area([(F(x), F(y)) for x, y in p2])
    if inter <= 0:
        return 0.0
    return float(inter / (a1 + a2 - inter))


def hull_exact(points):
    """convex hull (monotone chain, collinear points dropped) of exact points -> CCW vertex list"""
    pts = sorted(set((F(x), F(y)) for x, y in points))
    if len(pts) < 3:
        return pts

    def cross(o, a, b):
        return (a[0] - o[0]) * (b[1] - o[1]) - (a[1] - o[1]) * (b[0] - o[0])
    lower, upper = [], []
    for p in pts:
        while len(lower) >= 2 and cross(lower[-2], lower[-1], p) <= 0:
            lower.pop()
        lower.append(p)
    for p in reversed(pts):
        while len(upper) >= 2 and cross(upper[-2], upper[-1], p) <= 0:
            upper.pop()
        upper.append(p)
    return lower[:-1] + upper[:-1]


def loss_iou_exact(b1, b2, kind):
    """GIoU ("grbox") / DIoU ("drbox") of the exact polygons whose corners are the float64 values an implementation
    computes: hull area, intersection area and squared lengths without rounding; only the final quotient is rounded"""
    p1, p2 = corners(*map(float, b1)), corners(*map(float, b2))
    e1, e2 = [(F(x), F(y)) for x, y in p1], [(F(x), F(y)) for x, y in p2]
    a1, a2 = area(e1), area(e2)
    if a1 <= 0 or a2 <= 0:
        return 0.0
    inter = area(clip(p1, p2))
    if inter < 0:
        inter = F(0)
    union = a1 + a2 - inter
    if kind == "grbox":
        hull = area(hull_exact(e1 + e2))
        return float(inter / union - (hull - union) / hull)
    d2 = (F(float(b1[0])) - F(float(b2[0]))) ** 2 + (F(float(b1[1])) - F(float(b2[1]))) ** 2
    pts = e1 + e2
    diam2 = max((p[0] - q[0]) ** 2 + (p[1] - q[1]) ** 2 for p in pts for q in pts)
    return float(inter / union - d2 / diam2)
