/*
 * oracle/point_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 * Plain-C restatement of reference d3d/point/scatter.cpp (aligned_scatter forward/backward, MEAN and LINEAR).
 * Parity status: PINNED against the real reference built from /root/reference/d3d/point/{impl,scatter}.cpp by
 * oracle/build_ref.py (tests/golden/point_ref_cases.npz, generator tests/golden/make_point_golden.py).
 */
#include <stdint.h>
#define T float
#define FN(n) n##_f32
#include "point_oracle_impl.h"
#undef T
#undef FN
#define T double
#define FN(n) n##_f64
#include "point_oracle_impl.h"
#undef T
#undef FN
