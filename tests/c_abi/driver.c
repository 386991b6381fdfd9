/* A C program (no Python, no torch) that uses libd3d_hip.so through include/d3d_hip.h the way a C / C++ host of the
 * reference would after replacing its compiled modules: device buffers from the HIP runtime, one call per operator, results
 * compared with the CPU oracle (oracle/liboracle.so -- this file is test infrastructure, the only place besides the Python
 * tests where the oracle is linked).  Built and run by tests/test_gpu_cabi.py:
 *   gcc -std=c99 -O1 tests/c_abi/driver.c -Iinclude -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ -Ld3d_amd -ld3d_hip -Loracle -loracle
 *       -L/opt/rocm/lib -lamdhip64 -lm
 * Exit code 0 = every comparison passed. */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "d3d_hip.h"

/* oracle/voxel_oracle.c, oracle/box_oracle.c */
int64_t oracle_voxelize_3d_dense(const float *points, int64_t n, int32_t c, const int32_t *shape, const float *bound,
                                 int32_t max_points, int32_t max_voxels, int32_t reduction, float *voxels, int64_t *coords,
                                 uint8_t *pmask, int32_t *npoints, float *aggregates);
void oracle_iou2d_f64(const double *b1, int64_t n, const double *b2, int64_t m, int method /* 1 BOX, 2 RBOX */, int64_t row_begin,
                      int64_t row_end, double *ious);
void oracle_nms2d_f64(const double *boxes, const double *scores, int64_t n, const int64_t *order /* stable descending argsort */,
                      int method, int supp, float iou_threshold, float score_threshold, float supp_param, uint8_t *suppressed);

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %d at %s:%d\n", (int)e_, __FILE__, __LINE__); return 2; } } while (0)
#define CHECK_D3D(x) do { int r_ = (x); if (r_ != D3D_OK) { fprintf(stderr, "d3d error %d (%s) at %s:%d\n", r_, d3d_status_string(r_), __FILE__, __LINE__); return 3; } } while (0)

static uint32_t rng_state = 12345u;
static float frand(void) { rng_state = rng_state * 1664525u + 1013904223u; return (float)(rng_state >> 8) * (1.0f / 16777216.0f); }

static int voxel_case(void)
{
    const int64_t n = 50000;
    const int32_t c = 4, P = 8, max_voxels = 50000, shape[3] = {176, 200, 10};
    const float bound[6] = {0.f, 70.4f, -40.f, 40.f, -3.f, 1.f};
    float *pts = (float *)malloc(sizeof(float) * n * c);
    for (int64_t i = 0; i < n; i++) {
        pts[i * 4 + 0] = frand() * 72.f - 1.f;            /* a few points outside the range */
        pts[i * 4 + 1] = frand() * 82.f - 41.f;
        pts[i * 4 + 2] = frand() * 4.2f - 3.1f;
        pts[i * 4 + 3] = frand();
    }
    float *d_pts, *d_vox, *d_agg;
    int64_t *d_coords, *d_counts;
    uint8_t *d_pmask;
    int32_t *d_np;
    void *d_ws;
    const size_t ws_bytes = d3d_voxelize_workspace_bytes(n, 0);
    CHECK_HIP(hipMalloc((void **)&d_pts, sizeof(float) * n * c));
    CHECK_HIP(hipMalloc((void **)&d_vox, sizeof(float) * max_voxels * P * c));
    CHECK_HIP(hipMalloc((void **)&d_agg, sizeof(float) * max_voxels * c));
    CHECK_HIP(hipMalloc((void **)&d_coords, sizeof(int64_t) * max_voxels * 3));
    CHECK_HIP(hipMalloc((void **)&d_counts, sizeof(int64_t) * D3D_NUM_COUNTS));
    CHECK_HIP(hipMalloc((void **)&d_pmask, (size_t)max_voxels * P));
    CHECK_HIP(hipMalloc((void **)&d_np, sizeof(int32_t) * max_voxels));
    CHECK_HIP(hipMalloc(&d_ws, ws_bytes));
    CHECK_HIP(hipMemcpy(d_pts, pts, sizeof(float) * n * c, hipMemcpyHostToDevice));
    int bad = 0;
    for (uint32_t flags = 0; flags <= 1; flags++) {        /* binned index, then D3D_VOXEL_PATH_HASH */
        CHECK_D3D(d3d_voxelize_3d_dense(d_pts, n, c, shape, bound, P, max_voxels, D3D_REDUCE_MAX, d_vox, d_coords, d_pmask, d_np,
                                        d_agg, d_counts, d_ws, ws_bytes, NULL, flags));
        int64_t counts[D3D_NUM_COUNTS];
        CHECK_HIP(hipMemcpy(counts, d_counts, sizeof(counts), hipMemcpyDeviceToHost));     /* (synchronises with the null stream) */
        const int64_t V = counts[D3D_COUNT_VOXELS];
        float *e_vox = (float *)malloc(sizeof(float) * max_voxels * P * c), *e_agg = (float *)malloc(sizeof(float) * max_voxels * c);
        int64_t *e_coords = (int64_t *)malloc(sizeof(int64_t) * max_voxels * 3);
        uint8_t *e_pmask = (uint8_t *)malloc((size_t)max_voxels * P);
        int32_t *e_np = (int32_t *)malloc(sizeof(int32_t) * max_voxels);
        const int64_t EV = oracle_voxelize_3d_dense(pts, n, c, shape, bound, P, max_voxels, D3D_REDUCE_MAX, e_vox, e_coords, e_pmask,
                                                    e_np, e_agg);
        float *g_vox = (float *)malloc(sizeof(float) * V * P * c), *g_agg = (float *)malloc(sizeof(float) * V * c);
        int64_t *g_coords = (int64_t *)malloc(sizeof(int64_t) * V * 3);
        uint8_t *g_pmask = (uint8_t *)malloc((size_t)V * P);
        int32_t *g_np = (int32_t *)malloc(sizeof(int32_t) * V);
        CHECK_HIP(hipMemcpy(g_vox, d_vox, sizeof(float) * V * P * c, hipMemcpyDeviceToHost));
        CHECK_HIP(hipMemcpy(g_agg, d_agg, sizeof(float) * V * c, hipMemcpyDeviceToHost));
        CHECK_HIP(hipMemcpy(g_coords, d_coords, sizeof(int64_t) * V * 3, hipMemcpyDeviceToHost));
        CHECK_HIP(hipMemcpy(g_pmask, d_pmask, (size_t)V * P, hipMemcpyDeviceToHost));
        CHECK_HIP(hipMemcpy(g_np, d_np, sizeof(int32_t) * V, hipMemcpyDeviceToHost));
        const int ok = V == EV && counts[D3D_COUNT_STATUS] == 0 && !memcmp(g_vox, e_vox, sizeof(float) * V * P * c) &&
                       !memcmp(g_agg, e_agg, sizeof(float) * V * c) && !memcmp(g_coords, e_coords, sizeof(int64_t) * V * 3) &&
                       !memcmp(g_pmask, e_pmask, (size_t)V * P) && !memcmp(g_np, e_np, sizeof(int32_t) * V);
        printf("voxelize_3d_dense flags=%u: %lld voxels (oracle %lld) %s\n", flags, (long long)V, (long long)EV, ok ? "bit-exact" : "MISMATCH");
        bad += !ok;
        free(e_vox); free(e_agg); free(e_coords); free(e_pmask); free(e_np);
        free(g_vox); free(g_agg); free(g_coords); free(g_pmask); free(g_np);
    }
    hipFree(d_pts); hipFree(d_vox); hipFree(d_agg); hipFree(d_coords); hipFree(d_counts); hipFree(d_pmask); hipFree(d_np); hipFree(d_ws);
    free(pts);
    return bad;
}

/* the reference's default mode (sparse + filter) through the prepared argument block: voxel/__init__.py:93-103 in one call */
int64_t oracle_voxelize_3d_sparse(const float *points, int64_t n, int32_t c, const float *voxel_size, int64_t *points_mapping,
                                  int64_t *coords, int32_t *npoints);
int32_t oracle_voxelize_3d_filter(const float *feats, int64_t n, int32_t c, const int64_t *points_mapping, const int64_t *coords,
                                  const int32_t *voxel_npoints, int64_t nvox, const int64_t *coords_bound, int32_t min_points,
                                  int32_t max_points, int32_t max_voxels, int32_t max_points_filter, int32_t max_voxels_filter,
                                  float *out_feats, int64_t *out_mask, int64_t *out_mapping, int32_t *out_npoints, int64_t *out_coords,
                                  int64_t *counts);

static int sparse_call_case(void)
{
    const int64_t n = 60000;
    const int32_t c = 4;
    float *pts = (float *)malloc(sizeof(float) * n * c);
    for (int64_t i = 0; i < n; i++) {
        pts[i * 4 + 0] = frand() * 74.f - 2.f;            /* some points outside the coordinate bounds */
        pts[i * 4 + 1] = frand() * 84.f - 42.f;
        pts[i * 4 + 2] = frand() * 4.4f - 3.2f;
        pts[i * 4 + 3] = frand();
    }
    D3DSparseFilterCall call;
    memset(&call, 0, sizeof(call));
    call.n = n; call.c = c;
    call.min_points = 2; call.max_points = 3; call.max_voxels = 20000;
    call.max_points_filter = 1 /* TRIM */; call.max_voxels_filter = 1 /* TRIM */;
    call.voxel_size[0] = 0.4f; call.voxel_size[1] = 0.4f; call.voxel_size[2] = 0.4f;
    const int64_t cb[6] = {0, 176, -100, 100, -7, 2};       /* voxel coordinates [lo, hi) per axis */
    memcpy(call.coords_bound, cb, sizeof(cb));
    call.coord_offset[0] = 0; call.coord_offset[1] = -100; call.coord_offset[2] = -7; call.has_coord_offset = 1;
    size_t off[5];
    call.outputs_bytes = d3d_voxelize_3d_sparse_filter_call_layout(n, c, off);
    call.workspace_bytes = d3d_voxelize_3d_sparse_filter_call_workspace_bytes(n);
    float *d_pts;
    CHECK_HIP(hipMalloc((void **)&d_pts, sizeof(float) * n * c));
    CHECK_HIP(hipMalloc(&call.outputs, call.outputs_bytes));
    CHECK_HIP(hipMalloc(&call.workspace, call.workspace_bytes));
    CHECK_HIP(hipMemcpy(d_pts, pts, sizeof(float) * n * c, hipMemcpyHostToDevice));
    call.points = d_pts;
    CHECK_D3D(d3d_voxelize_3d_sparse_filter_call(&call));
    int64_t counts[2 * D3D_NUM_COUNTS];                     /* the device count rows: the front of the workspace */
    CHECK_HIP(hipMemcpy(counts, call.workspace, sizeof(counts), hipMemcpyDeviceToHost));
    const int64_t k = counts[D3D_NUM_COUNTS + D3D_COUNT_POINTS], v = counts[D3D_NUM_COUNTS + D3D_COUNT_VOXELS];
    /* oracle: the two reference calls, then coords - offset */
    int64_t *m0 = (int64_t *)malloc(8 * n), *c0 = (int64_t *)malloc(24 * n), *e_mask = (int64_t *)malloc(8 * n), *e_map = (int64_t *)malloc(8 * n),
            *e_crd = (int64_t *)malloc(24 * n), ecnt[D3D_NUM_COUNTS];
    int32_t *n0 = (int32_t *)malloc(4 * n), *e_np = (int32_t *)malloc(4 * n);
    float *e_feats = (float *)malloc(sizeof(float) * n * c);
    const int64_t nv0 = oracle_voxelize_3d_sparse(pts, n, c, call.voxel_size, m0, c0, n0);
    const int32_t rc = oracle_voxelize_3d_filter(pts, n, c, m0, c0, n0, nv0, cb, call.min_points, call.max_points, call.max_voxels, 1, 1,
                                                 e_feats, e_mask, e_map, e_np, e_crd, ecnt);
    const int64_t ek = ecnt[0], ev = ecnt[1];                /* (the oracle's own order: kept points, kept voxels) */
    for (int64_t q = 0; q < ev; q++) { e_crd[q * 3 + 1] += 100; e_crd[q * 3 + 2] += 7; }
    int ok = rc == 0 && k == ek && v == ev && counts[D3D_COUNT_STATUS] == 0 && k > 1000 && v > 500;
    if (ok) {
        char *host = (char *)malloc(call.outputs_bytes);
        CHECK_HIP(hipMemcpy(host, call.outputs, call.outputs_bytes, hipMemcpyDeviceToHost));
        ok = !memcmp(host + off[0], e_feats, sizeof(float) * k * c) && !memcmp(host + off[1], e_mask, 8 * k) &&
             !memcmp(host + off[2], e_map, 8 * k) && !memcmp(host + off[3], e_np, 4 * v) && !memcmp(host + off[4], e_crd, 24 * v);
        free(host);
    }
    printf("voxelize_3d_sparse_filter_call: %lld points in %lld voxels kept (oracle %lld / %lld) %s\n", (long long)k, (long long)v,
           (long long)ek, (long long)ev, ok ? "bit-exact" : "MISMATCH");
    hipFree(d_pts); hipFree(call.outputs); hipFree(call.workspace);
    free(pts); free(m0); free(c0); free(e_mask); free(e_map); free(e_crd); free(n0); free(e_np); free(e_feats);
    return !ok;
}

static const double *sort_scores;
static int by_score_desc(const void *pa, const void *pb)           /* ties in ascending index: the order of d3d_argsort_desc */
{
    const int64_t a = *(const int64_t *)pa, b = *(const int64_t *)pb;
    if (sort_scores[a] != sort_scores[b]) return sort_scores[a] > sort_scores[b] ? -1 : 1;
    return a < b ? -1 : (a > b);
}

static int box_case(void)
{
    const int64_t n = 1500;
    double *b = (double *)malloc(sizeof(double) * n * 5), *s = (double *)malloc(sizeof(double) * n);
    for (int64_t i = 0; i < n; i++) {
        b[i * 5 + 0] = frand() * 300.; b[i * 5 + 1] = frand() * 300.;
        b[i * 5 + 2] = frand() * 20. + 5.; b[i * 5 + 3] = frand() * 20. + 5.; b[i * 5 + 4] = (frand() - 0.5) * 6.;
        s[i] = floor(frand() * 64.) / 64.;                 /* ties */
    }
    double *d_b, *d_s, *d_iou;
    uint8_t *d_sup;
    void *d_ws;
    size_t ws_bytes = d3d_iou2d_workspace_bytes(n, n, D3D_F64);
    const size_t nms_bytes = d3d_nms2d_workspace_bytes(n);
    if (nms_bytes > ws_bytes) ws_bytes = nms_bytes;
    CHECK_HIP(hipMalloc((void **)&d_b, sizeof(double) * n * 5));
    CHECK_HIP(hipMalloc((void **)&d_s, sizeof(double) * n));
    CHECK_HIP(hipMalloc((void **)&d_iou, sizeof(double) * n * n));
    CHECK_HIP(hipMalloc((void **)&d_sup, (size_t)n));
    CHECK_HIP(hipMalloc(&d_ws, ws_bytes));
    CHECK_HIP(hipMemcpy(d_b, b, sizeof(double) * n * 5, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_s, s, sizeof(double) * n, hipMemcpyHostToDevice));
    int bad = 0;
    CHECK_D3D(d3d_iou2d_forward(d_b, n, d_b, n, D3D_IOU_RBOX, D3D_F64, d_iou, d_ws, ws_bytes, NULL, 0));
    double *g = (double *)malloc(sizeof(double) * n * n), *e = (double *)malloc(sizeof(double) * n * n);
    CHECK_HIP(hipMemcpy(g, d_iou, sizeof(double) * n * n, hipMemcpyDeviceToHost));
    oracle_iou2d_f64(b, n, b, n, D3D_IOU_RBOX, 0, n, e);
    double worst = 0;
    for (int64_t i = 0; i < n * n; i++) { const double d = fabs(g[i] - e[i]); if (d > worst) worst = d; }
    printf("iou2d_forward rbox fp64 %lld x %lld: max |diff| = %.3g %s\n", (long long)n, (long long)n, worst, worst < 1e-9 ? "ok" : "MISMATCH");
    bad += !(worst < 1e-9);
    /* order = NULL: the library sorts the scores itself, as nms2d does (nms.cpp:103) */
    CHECK_D3D(d3d_nms2d(d_b, d_s, NULL, n, D3D_IOU_RBOX, D3D_SUPPRESS_HARD, D3D_F64, 0.3f, 0.2f, 0.f, d_sup, d_ws, ws_bytes, NULL, 0));
    uint8_t *gs = (uint8_t *)malloc((size_t)n), *es = (uint8_t *)malloc((size_t)n);
    CHECK_HIP(hipMemcpy(gs, d_sup, (size_t)n, hipMemcpyDeviceToHost));
    int64_t *order = (int64_t *)malloc(sizeof(int64_t) * n);
    for (int64_t i = 0; i < n; i++) order[i] = i;
    sort_scores = s;
    qsort(order, (size_t)n, sizeof(int64_t), by_score_desc);
    oracle_nms2d_f64(b, s, n, order, D3D_IOU_RBOX, D3D_SUPPRESS_HARD, 0.3f, 0.2f, 0.f, es);
    free(order);
    int64_t kept = 0, diff = 0;
    for (int64_t i = 0; i < n; i++) { kept += !gs[i]; diff += (gs[i] != 0) != (es[i] != 0); }
    printf("nms2d rbox fp64 hard: kept %lld of %lld, %lld differences %s\n", (long long)kept, (long long)n, (long long)diff, diff ? "MISMATCH" : "bit-exact");
    bad += diff != 0;
    hipFree(d_b); hipFree(d_s); hipFree(d_iou); hipFree(d_sup); hipFree(d_ws);
    free(b); free(s); free(g); free(e); free(gs); free(es);
    return bad;
}

int main(void)
{
    printf("libd3d_hip ABI version %d\n", d3d_abi_version());
    const int bad = voxel_case() + sparse_call_case() + box_case();
    printf(bad ? "FAILED\n" : "all comparisons passed\n");
    return bad ? 1 : 0;
}
