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
