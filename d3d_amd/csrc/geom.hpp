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

// Clip segment P + t*D, t in [0,1], against the CCW convex quad with vertices (qx, qy): the surviving piece
// [S, E] (false if none).
// CLOSED: a segment lying exactly on a quad edge that runs in the same direction counts as
// inside (used for A's edges so that a shared boundary is integrated exactly once; collinear
// edges running in opposite directions -- boxes touching from outside -- are dropped from both).
template <typename T, bool CLOSED>
__device__ __forceinline__ bool clip_edge_piece(T px, T py, T dx, T dy, const T (&qx)[4], const T (&qy)[4], T &sx, T &sy,
                                                T &ex_, T &ey_)
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
    if (!alive || !(t0 < t1)) return false;
    sx = px + t0 * dx; sy = py + t0 * dy;
    ex_ = px + t1 * dx; ey_ = py + t1 * dy;
    return true;
}

// cross(start, end) of the surviving piece (0 if none): its contribution to the Green's-theorem area integral
template <typename T, bool CLOSED>
__device__ __forceinline__ T clip_edge_cross(T px, T py, T dx, T dy, const T (&qx)[4], const T (&qy)[4])
{
    T sx, sy, ex, ey;
    if (!clip_edge_piece<T, CLOSED>(px, py, dx, dy, qx, qy, sx, sy, ex, ey)) return 0;
    return sx * ey - sy * ex;
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

// ---------------------------------------------------------------- gradients (loss path)
// d IoU / d (x, y, w, h, r) of both boxes.  The intersection area changes only through the normal motion of the
// pieces of each box's OWN boundary that lie inside the other box (Reynolds transport):
//   dI/dtheta_A = sum over pieces [S,E] of A's edges inside B of  v(mid) x (E - S),   v = d(boundary point)/dtheta
// and the velocity fields of a box are linear in position (translation, rotation about the centre, stretching
// along its axes), so the midpoint rule is exact.  IoU = I / U, U = A1 + A2 - I:
//   dIoU = (dI (U + I) - I dA) / U^2,   dA1/dw1 = h1, dA1/dh1 = w1.
// ga / gb receive the 5 partials (zero when the boxes do not overlap).  Returns the IoU.
template <typename T>
__device__ __forceinline__ void piece_grad(T sx, T sy, T ex, T ey, T ox, T oy, T ux, T uy, T vx, T vy, T w, T h, T (&g)[5])
{
    // midpoint relative to the owning box's centre (ox, oy), piece vector d
    const T mx = (sx + ex) / 2 - ox, my = (sy + ey) / 2 - oy, dx = ex - sx, dy = ey - sy;
    g[0] += dy;                                   // v = (1, 0)
    g[1] += -dx;                                  // v = (0, 1)
    g[4] += -my * dy - mx * dx;                   // v = (-my, mx)
    const T uc = mx * ux + my * uy, vc = mx * vx + my * vy;      // local coordinates along the unit axes
    g[2] += (uc / w) * (ux * dy - uy * dx);       // v = (uc / w) * u_hat
    g[3] += (vc / h) * (vx * dy - vy * dx);       // v = (vc / h) * v_hat
}

template <typename T>
__device__ __forceinline__ T iou_rbox_grad(const BoxGeom<T> &a, const BoxGeom<T> &b, T w1, T h1, T w2, T h2, T (&ga)[5],
                                           T (&gb)[5])
{
#pragma unroll
    for (int k = 0; k < 5; k++) { ga[k] = 0; gb[k] = 0; }
    if (!(a.area > 0) || !(b.area > 0) || aabb_disjoint(a, b)) return 0;
    const T ax[4] = {-a.ux - a.vx, a.ux - a.vx, a.ux + a.vx, -a.ux + a.vx};
    const T ay[4] = {-a.uy - a.vy, a.uy - a.vy, a.uy + a.vy, -a.uy + a.vy};
    const T ox = b.cx - a.cx, oy = b.cy - a.cy;
    const T bx[4] = {ox - b.ux - b.vx, ox + b.ux - b.vx, ox + b.ux + b.vx, ox - b.ux + b.vx};
    const T by[4] = {oy - b.uy - b.vy, oy + b.uy - b.vy, oy + b.uy + b.vy, oy - b.uy + b.vy};
    // unit axes of both boxes
    const T aux = 2 * a.ux / w1, auy = 2 * a.uy / w1, avx = 2 * a.vx / h1, avy = 2 * a.vy / h1;
    const T bux = 2 * b.ux / w2, buy = 2 * b.uy / w2, bvx = 2 * b.vx / h2, bvy = 2 * b.vy / h2;
    T acc = 0, da[5] = {0, 0, 0, 0, 0}, db[5] = {0, 0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 4; k++) {
        T sx, sy, ex, ey;
        if (clip_edge_piece<T, true>(ax[k], ay[k], ax[(k + 1) & 3] - ax[k], ay[(k + 1) & 3] - ay[k], bx, by, sx, sy, ex, ey)) {
            acc += sx * ey - sy * ex;
            piece_grad<T>(sx, sy, ex, ey, (T)0, (T)0, aux, auy, avx, avy, w1, h1, da);
        }
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
        T sx, sy, ex, ey;
        if (clip_edge_piece<T, false>(bx[k], by[k], bx[(k + 1) & 3] - bx[k], by[(k + 1) & 3] - by[k], ax, ay, sx, sy, ex, ey)) {
            acc += sx * ey - sy * ex;
            piece_grad<T>(sx, sy, ex, ey, ox, oy, bux, buy, bvx, bvy, w2, h2, db);
        }
    }
    const T I = acc / 2;
    if (!(I > 0)) return 0;
    const T U = a.area + b.area - I, U2 = U * U;
    const T dA1[5] = {0, 0, h1, w1, 0}, dA2[5] = {0, 0, h2, w2, 0};
#pragma unroll
    for (int k = 0; k < 5; k++) {
        ga[k] = (da[k] * (U + I) - I * dA1[k]) / U2;
        gb[k] = (db[k] * (U + I) - I * dA2[k]) / U2;
    }
    return I / U;
}

// d IoU(AABB of rotated rect A, AABB of rotated rect B) / d params (method "box")
template <typename T>
__device__ __forceinline__ void aabb_param_grad(T w, T h, T r, T dIdx0, T dIdx1, T dIdy0, T dIdy1, T dAdhx, T dAdhy,
                                                T I, T U, T (&g)[5])
{
    // bounds: x -+ hx, y -+ hy with hx = (|w c| + |h s|)/2, hy = (|w s| + |h c|)/2
    T s, c;
    d3d_sincos(r, &s, &c);
    const T sc = c < 0 ? (T)-1 : (T)1, ss = s < 0 ? (T)-1 : (T)1, sw = w < 0 ? (T)-1 : (T)1, sh = h < 0 ? (T)-1 : (T)1;
    const T dhx[3] = {fabs(c) * sw / 2, fabs(s) * sh / 2, (-fabs(w) * s * sc + fabs(h) * c * ss) / 2};   // d hx / d(w,h,r)
    const T dhy[3] = {fabs(s) * sw / 2, fabs(c) * sh / 2, (fabs(w) * c * ss - fabs(h) * s * sc) / 2};
    const T dI[5] = {dIdx0 + dIdx1, dIdy0 + dIdy1,
                     (dIdx1 - dIdx0) * dhx[0] + (dIdy1 - dIdy0) * dhy[0],
                     (dIdx1 - dIdx0) * dhx[1] + (dIdy1 - dIdy0) * dhy[1],
                     (dIdx1 - dIdx0) * dhx[2] + (dIdy1 - dIdy0) * dhy[2]};
    const T dA[5] = {0, 0, dAdhx * dhx[0] + dAdhy * dhy[0], dAdhx * dhx[1] + dAdhy * dhy[1],
                     dAdhx * dhx[2] + dAdhy * dhy[2]};
    const T U2 = U * U;
#pragma unroll
    for (int k = 0; k < 5; k++) g[k] = (dI[k] * (U + I) - I * dA[k]) / U2;
}

template <typename T>
__device__ __forceinline__ T iou_aabb_grad(const BoxGeom<T> &a, const BoxGeom<T> &b, const T *pa, const T *pb, T (&ga)[5],
                                           T (&gb)[5])
{
#pragma unroll
    for (int k = 0; k < 5; k++) { ga[k] = 0; gb[k] = 0; }
    const T ix = fmin(a.xmax, b.xmax) - fmax(a.xmin, b.xmin);
    const T iy = fmin(a.ymax, b.ymax) - fmax(a.ymin, b.ymin);
    if (!(ix > 0) || !(iy > 0)) return 0;
    const T I = ix * iy;
    const T ax = a.xmax - a.xmin, ay = a.ymax - a.ymin, bx = b.xmax - b.xmin, by = b.ymax - b.ymin;
    const T U = ax * ay + bx * by - I;
    // dI / d bounds (the active side of each min / max)
    const T a_x1 = a.xmax < b.xmax ? iy : 0, b_x1 = a.xmax < b.xmax ? 0 : iy;
    const T a_x0 = a.xmin > b.xmin ? -iy : 0, b_x0 = a.xmin > b.xmin ? 0 : -iy;
    const T a_y1 = a.ymax < b.ymax ? ix : 0, b_y1 = a.ymax < b.ymax ? 0 : ix;
    const T a_y0 = a.ymin > b.ymin ? -ix : 0, b_y0 = a.ymin > b.ymin ? 0 : -ix;
    // area = (2 hx)(2 hy): dA/dhx = 2 * (2 hy) = 2 * height
    aabb_param_grad<T>(pa[2], pa[3], pa[4], a_x0, a_x1, a_y0, a_y1, 2 * ay, 2 * ax, I, U, ga);
    aabb_param_grad<T>(pb[2], pb[3], pb[4], b_x0, b_x1, b_y0, b_y1, 2 * by, 2 * bx, I, U, gb);
    return I / U;
}
