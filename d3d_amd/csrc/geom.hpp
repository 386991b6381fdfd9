// geom.hpp -- device-side 2D geometry for rotated boxes (our restatement of what the
// reference takes from the un-vendored dgal/geometry.hpp; call sites: reference
// d3d/box/utils.h:15-34, iou.cpp:30,116, nms.cpp:51, dgal_wrap.h:45-91).
//
// Design for CDNA4 (one box pair per lane, no MFMA -- this is branching geometry):
//  * a box is expanded ONCE into `BoxGeom` (trig + corners + AABB + area); tiles of them are
//    staged in LDS, so the pair loop never evaluates sin/cos (the reference re-runs
//    poly2_from_xywhr twice per pair, iou.cpp:113-114);
//  * the intersection area is the Green's-theorem line integral over the boundary of A∩B:
//    each of the 8 edges is clipped parametrically (Cyrus-Beck) against the other convex quad
//    and contributes cross(start, end)/2.  Fixed trip counts, registers only -- no
//    dynamically indexed vertex list (which hipcc would spill to scratch), unlike a
//    Sutherland-Hodgman vertex loop.  All coordinates are taken relative to A's centre,
//    which keeps fp32 cross products well conditioned.
#pragma once
#include <hip/hip_runtime.h>

template <typename T> struct BoxGeom {
    T cx, cy;          // centre
    T ux, uy;          // (w/2) * (cos r, sin r)   half-extent along the local x axis
    T vx, vy;          // (h/2) * (-sin r, cos r)  half-extent along the local y axis
    T xmin, xmax, ymin, ymax;
    T area;
};

__device__ __forceinline__ void d3d_sincos(float r, float *s, float *c) { sincosf(r, s, c); }
__device__ __forceinline__ void d3d_sincos(double r, double *s, double *c) { sincos(r, s, c); }

// corners (CCW, starting at local (-w/2,-h/2)) = c - u - v, c + u - v, c + u + v, c - u + v
template <typename T>
__device__ __forceinline__ BoxGeom<T> make_geom(T x, T y, T w, T h, T r)
{
    T s, c;
    d3d_sincos(r, &s, &c);
    BoxGeom<T> g;
    g.cx = x; g.cy = y;
    g.ux = w * c / 2; g.uy = w * s / 2;
    g.vx = -(h * s / 2); g.vy = h * c / 2;
    T ex = fabs(g.ux) + fabs(g.vx), ey = fabs(g.uy) + fabs(g.vy);
    // AABB of the four corners (dgal::aabox2_from_poly2)
    T x0 = x - g.ux - g.vx, x1 = x + g.ux - g.vx, x2 = x + g.ux + g.vx, x3 = x - g.ux + g.vx;
    T y0 = y - g.uy - g.vy, y1 = y + g.uy - g.vy, y2 = y + g.uy + g.vy, y3 = y - g.uy + g.vy;
    g.xmin = fmin(fmin(x0, x1), fmin(x2, x3)); g.xmax = fmax(fmax(x0, x1), fmax(x2, x3));
    g.ymin = fmin(fmin(y0, y1), fmin(y2, y3)); g.ymax = fmax(fmax(y0, y1), fmax(y2, y3));
    (void)ex; (void)ey;
    // signed shoelace area of the quad = 4 * cross(u, v)  (= w*h for positive sizes)
    g.area = 4 * (g.ux * g.vy - g.uy * g.vx);
    return g;
}

// IoU of the axis-aligned bounding boxes (method "box": dgal::iou(AABox2, AABox2))
template <typename T>
__device__ __forceinline__ T iou_aabb(const BoxGeom<T> &a, const BoxGeom<T> &b)
{
    T ix = fmin(a.xmax, b.xmax) - fmax(a.xmin, b.xmin);
    T iy = fmin(a.ymax, b.ymax) - fmax(a.ymin, b.ymin);
    if (!(ix > 0) || !(iy > 0)) return 0;
    T inter = ix * iy;
    T a1 = (a.xmax - a.xmin) * (a.ymax - a.ymin);
    T a2 = (b.xmax - b.xmin) * (b.ymax - b.ymin);
    return inter / (a1 + a2 - inter);
}

template <typename T>
__device__ __forceinline__ bool aabb_disjoint(const BoxGeom<T> &a, const BoxGeom<T> &b)
{
    return !(a.xmin < b.xmax && b.xmin < a.xmax && a.ymin < b.ymax && b.ymin < a.ymax);
}

// Clip segment P + t*D, t in [0,1], against the CCW convex quad with vertices (qx, qy);
// returns cross(start, end) of the surviving piece (0 if none).
// CLOSED: a segment lying exactly on a quad edge that runs in the same direction counts as
// inside (used for A's edges so that a shared boundary is integrated exactly once; collinear
// edges running in opposite directions -- boxes touching from outside -- are dropped from both).
template <typename T, bool CLOSED>
__device__ __forceinline__ T clip_edge_cross(T px, T py, T dx, T dy, const T (&qx)[4], const T (&qy)[4])
{
    T t0 = 0, t1 = 1;
    bool alive = true;
#pragma unroll
    for (int e = 0; e < 4; e++) {
        const T ex = qx[(e + 1) & 3] - qx[e], ey = qy[(e + 1) & 3] - qy[e];
        const T n0 = ex * (py - qy[e]) - ey * (px - qx[e]);   // cross(E, P - Q_e): >= 0 inside
        const T nd = ex * dy - ey * dx;                        // d/dt of the above
        if (nd > 0) {
            t0 = fmax(t0, -n0 / nd);
        } else if (nd < 0) {
            t1 = fmin(t1, -n0 / nd);
        } else {
            bool in = n0 > 0;
            if (CLOSED) in = in || (n0 == 0 && (ex * dx + ey * dy) > 0);
            alive = alive && in;
        }
    }
    if (!alive || !(t0 < t1)) return 0;
    const T sx = px + t0 * dx, sy = py + t0 * dy;
    const T ex_ = px + t1 * dx, ey_ = py + t1 * dy;
    return sx * ey_ - sy * ex_;
}

// area of A ∩ B for two CCW quads
template <typename T>
__device__ __forceinline__ T intersection_area(const BoxGeom<T> &a, const BoxGeom<T> &b)
{
    // corners relative to A's centre
    const T ax[4] = {-a.ux - a.vx, a.ux - a.vx, a.ux + a.vx, -a.ux + a.vx};
    const T ay[4] = {-a.uy - a.vy, a.uy - a.vy, a.uy + a.vy, -a.uy + a.vy};
    const T ox = b.cx - a.cx, oy = b.cy - a.cy;
    const T bx[4] = {ox - b.ux - b.vx, ox + b.ux - b.vx, ox + b.ux + b.vx, ox - b.ux + b.vx};
    const T by[4] = {oy - b.uy - b.vy, oy + b.uy - b.vy, oy + b.uy + b.vy, oy - b.uy + b.vy};
    T acc = 0;
#pragma unroll
    for (int k = 0; k < 4; k++)
        acc += clip_edge_cross<T, true>(ax[k], ay[k], ax[(k + 1) & 3] - ax[k], ay[(k + 1) & 3] - ay[k], bx, by);
#pragma unroll
    for (int k = 0; k < 4; k++)
        acc += clip_edge_cross<T, false>(bx[k], by[k], bx[(k + 1) & 3] - bx[k], by[(k + 1) & 3] - by[k], ax, ay);
    return acc / 2;
}

// rotated IoU (method "rbox": dgal::iou(Quad2, Quad2))
template <typename T>
__device__ __forceinline__ T iou_rbox(const BoxGeom<T> &a, const BoxGeom<T> &b)
{
    if (!(a.area > 0) || !(b.area > 0)) return 0;   // degenerate (zero / negative size) boxes: IoU 0, never NaN
    if (aabb_disjoint(a, b)) return 0;
    T inter = intersection_area(a, b);
    if (!(inter > 0)) return 0;
    return inter / (a.area + b.area - inter);
}
