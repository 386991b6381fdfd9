/*
 * oracle/box_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C CPU restatement of the reference's rotated/axis-aligned 2D IoU, NMS and
 * the BEV x z "3D IoU" (cmpute/d3d: d3d/box/iou.cpp, d3d/box/nms.cpp,
 * d3d/dgal_wrap.h, d3d/tracking/matcher.pyx).  Only tests/, smoke() and
 * bench.py's cpu_baseline leg may use it.
 *
 * Parity status: the polygon arithmetic lives in the third-party header
 * dgal/geometry.hpp (github.com/cmpute/dgal, git submodule, NOT vendored in the
 * reference snapshot and unpinned), so the box sources cannot be compiled here.
 * The oracle is pinned on VALUES by every known-answer vector the reference's
 * tests hold for this path (test/test_box.py:12-155, test/test_benchmark.py:45-71;
 * see tests/test_oracle_box.py and tests/golden/box_known_answers.json) and by an
 * exact-rational clipper (tests/exact_clip.py).  Bit-level parity with dgal is
 * UNPINNED (no source, no binary): stated in DESIGN.md.
 */
#include <stdint.h>
#include <stdlib.h>
#include <math.h>

#define T float
#define FN(n) n##_f32
#define SIN sinf
#define COS cosf
#define POW powf
#define EXP expf
#include "box_oracle_impl.h"
#undef T
#undef FN
#undef SIN
#undef COS
#undef POW
#undef EXP

#define T double
#define FN(n) n##_f64
#define SIN sin
#define COS cos
#define POW pow
#define EXP exp
#include "box_oracle_impl.h"
#undef T
#undef FN
#undef SIN
#undef COS
#undef POW
#undef EXP

/* box3dr_iou (rotated=1, dgal_wrap.h:45-67) / box3d_iou (rotated=0, :69-91):
 * BEV IoU times 1-D z-interval IoU, everything in fp32.
 * box = (x, y, z, lx, ly, lz, rz). */
static float box3d_pair(const float *a, const float *b, int rotated)
{
    quad_f32 q1 = quad_from_xywhr_f32(a[0], a[1], a[3], a[4], a[6]);
    quad_f32 q2 = quad_from_xywhr_f32(b[0], b[1], b[3], b[4], b[6]);
    float iou2d;
    if (rotated) iou2d = iou_quad_f32(&q1, &q2);
    else {
        aabox_f32 b1 = aabox_from_quad_f32(&q1), b2 = aabox_from_quad_f32(&q2);
        iou2d = iou_aabox_f32(&b1, &b2);
    }
    float z1max = a[2] + a[5] / 2, z1min = a[2] - a[5] / 2;
    float z2max = b[2] + b[5] / 2, z2min = b[2] - b[5] / 2;
    float imax = z1max < z2max ? z1max : z2max;
    float imin = z1min > z2min ? z1min : z2min;
    float umax = z1max > z2max ? z1max : z2max;
    float umin = z1min < z2min ? z1min : z2min;
    float i = imax - imin > 0.f ? imax - imin : 0.f;
    float u = umax - umin > (float)1e-6 ? umax - umin : (float)1e-6;
    return iou2d * (i / u);
}

/* pairwise loop of BaseMatcher.prepare_boxes (matcher.pyx:57-80) without the
 * "1 -" (that is applied by the caller); src[n,7], dst[m,7] -> out[n,m]. */
void oracle_iou3d(const float *src, int64_t n, const float *dst, int64_t m, int rotated,
                  int64_t row_begin, int64_t row_end, float *out)
{
    (void)n;
    for (int64_t i = row_begin; i < row_end; i++)
        for (int64_t j = 0; j < m; j++)
            out[i * m + j] = box3d_pair(src + i * 7, dst + j * 7, rotated);
}


/* box3dr_contains (dgal_wrap.h:6-19), fp32: the z interval is CLOSED (a point is rejected only when strictly above z + lz/2
 * or strictly below z - lz/2), then aabox.contains, then box.contains -- restated as in oracle_crop_2dr (closed tests).
 * box = (x, y, z, lx, ly, lz, rz). */
static int box3dr_contains_f32(const float *b, float xq, float yq, float zq)
{
    quad_f32 q = quad_from_xywhr_f32(b[0], b[1], b[3], b[4], b[6]);
    aabox_f32 a = aabox_from_quad_f32(&q);
    if (zq > b[2] + b[5] / 2 || zq < b[2] - b[5] / 2) return 0;
    if (!(xq >= a.xmin && xq <= a.xmax && yq >= a.ymin && yq <= a.ymax)) return 0;
    for (int e = 0; e < 4; e++) {
        pt_f32 v = q.v[e], w = q.v[(e + 1) & 3];
        float cr = (w.x - v.x) * (yq - v.y) - (w.y - v.y) * (xq - v.x);
        if (!(cr >= 0)) return 0;
    }
    return 1;
}

/* Target3DArray._crop_points (abstraction.pyx:654-660) -> ObjectTarget3D._crop (:308-319): result[i][j] =
 * box3dr_contains(box i, cloud[j, 0..2]).  cloud[n, pstride], box row i at boxes + i * bstride + boff. */
void oracle_crop_3dr(const float *cloud, int64_t n, int pstride, const float *boxes, int64_t m, int bstride, int boff,
                     uint8_t *out)
{
    for (int64_t i = 0; i < m; i++)
        for (int64_t j = 0; j < n; j++)
            out[i * n + j] = (uint8_t)box3dr_contains_f32(boxes + i * bstride + boff, cloud[j * pstride], cloud[j * pstride + 1],
                                                          cloud[j * pstride + 2]);
}

/* Target3DArray.paint_label (abstraction.pyx:662-682), literally: the mask of all boxes first, idarr zeroed, then the boxes
 * from the LAST to the first ("assuming scores are sorted descendingly"): idarr[ip] = ib + 1 (uint16) where the mask is set
 * and semantics[ip] equals the box's class. */
void oracle_paint_label(const float *cloud, int64_t n, int pstride, const uint8_t *semantics, const float *boxes, int64_t m,
                        int bstride, int boff, const uint8_t *labels, uint16_t *idarr)
{
    uint8_t *mask = (uint8_t *)malloc((size_t)(m > 0 ? m : 1) * (size_t)(n > 0 ? n : 1));
    oracle_crop_3dr(cloud, n, pstride, boxes, m, bstride, boff, mask);
    for (int64_t ip = 0; ip < n; ip++) idarr[ip] = 0;
    for (int64_t ib = m - 1; ib >= 0; ib--) {
        uint8_t target_cls = labels[ib];
        for (int64_t ip = 0; ip < n; ip++)
            if (mask[ib * n + ip] && semantics[ip] == target_cls) idarr[ip] = (uint16_t)(ib + 1);
    }
    free(mask);
}
