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
