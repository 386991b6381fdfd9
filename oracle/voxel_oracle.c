/*
 * oracle/voxel_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C, single-threaded CPU restatement of the reference voxelizer
 * (cmpute/d3d, d3d/voxel/voxelize.cpp).  It exists only so that tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg can check / time the
 * HIP path against the reference algorithm on a machine where /root/reference
 * is absent.  Nothing under d3d_amd/ may import, link or call it.
 *
 * Parity status: PINNED.  Checked bit-for-bit against (a) the reference's own
 * fixture test/voxel_data.npz (tests/golden/voxel_data_ref.npz, reference
 * test/test_voxel.py:80-88) and (b) outputs of the real reference compiled from
 * /root/reference by oracle/build_ref.py on randomized + edge-case clouds
 * (tests/golden/voxel_*.npz, generator tests/golden/make_voxel_golden.py).
 *
 * Each function cites the reference lines it follows.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <limits.h>

/* ------------------------------------------------------------------------- */
/* coord -> voxel-id map.  Reference: std::unordered_map<tuple<int,int,int>,int>
 * with hash (x*997+y)*997+z (voxelize.cpp:16-42).  Only the mapping matters for
 * results; we keep the same hash formula so the timed CPU baseline behaves alike. */
typedef struct { int32_t x, y, z, id; } cell_t;
typedef struct { cell_t *cells; uint64_t mask; uint64_t used; } cmap_t;

static uint64_t cmap_hash(int32_t x, int32_t y, int32_t z)
{
    /* voxelize.cpp:20-26 (size_t arithmetic, p = 997) followed by a bit mix so
     * that power-of-two bucket counts behave like libstdc++'s prime modulo */
    uint64_t h = (uint64_t)(int64_t)x;
    h = h * 997u + (uint64_t)(int64_t)y;
    h = h * 997u + (uint64_t)(int64_t)z;
    h ^= h >> 33; h *= 0xff51afd7ed558ccdULL; h ^= h >> 33;
    return h;
}

static int cmap_init(cmap_t *m, uint64_t expect)
{
    uint64_t cap = 64;
    while (cap < expect * 2) cap <<= 1;
    m->cells = (cell_t *)malloc(cap * sizeof(cell_t));
    if (!m->cells) return -1;
    for (uint64_t i = 0; i < cap; i++) m->cells[i].id = -1;
    m->mask = cap - 1; m->used = 0;
    return 0;
}

static int cmap_grow(cmap_t *m);

/* returns pointer to the cell for (x,y,z); cell->id == -1 when absent */
static cell_t *cmap_find(cmap_t *m, int32_t x, int32_t y, int32_t z)
{
    uint64_t h = cmap_hash(x, y, z) & m->mask;
    for (;;) {
        cell_t *c = &m->cells[h];
        if (c->id < 0) return c;
        if (c->x == x && c->y == y && c->z == z) return c;
        h = (h + 1) & m->mask;
    }
}

static int cmap_insert(cmap_t *m, cell_t *slot, int32_t x, int32_t y, int32_t z, int32_t id)
{
    slot->x = x; slot->y = y; slot->z = z; slot->id = id;
    m->used++;
    if (m->used * 2 > m->mask + 1) return cmap_grow(m);
    return 0;
}

static int cmap_grow(cmap_t *m)
{
    cmap_t n;
    uint64_t cap = (m->mask + 1) * 2;
    n.cells = (cell_t *)malloc(cap * sizeof(cell_t));
    if (!n.cells) return -1;
    for (uint64_t i = 0; i < cap; i++) n.cells[i].id = -1;
    n.mask = cap - 1; n.used = m->used;
    for (uint64_t i = 0; i <= m->mask; i++) {
        cell_t *c = &m->cells[i];
        if (c->id >= 0) { cell_t *d = cmap_find(&n, c->x, c->y, c->z); *d = *c; }
    }
    free(m->cells);
    *m = n;
    return 0;
}

/* float -> int the way the x86-64 build of the reference does it (cvttss2si):
 * truncation toward zero; NaN / out-of-range -> INT_MIN ("integer indefinite"). */
static int32_t f2i_trunc(float q)
{
    if (!(q > -2147483904.0f && q < 2147483648.0f)) return INT_MIN;
    return (int32_t)q;
}

/* ------------------------------------------------------------------------- */
/* voxelize_3d_dense_templated<R>  (voxelize.cpp:45-180)
 * reduction: 0 NONE, 1 MEAN, 2 MAX, 3 MIN (voxelize.h:5).
 * Caller provides buffers sized for max_voxels voxels:
 *   voxels[max_voxels*max_points*c] (this function zero-fills, :56)
 *   coords[max_voxels*3] i64, pmask[max_voxels*max_points] u8 (zero-filled here;
 *   the reference leaves it uninitialised, :58), npoints[max_voxels] i32,
 *   aggregates[max_voxels*c] (may be NULL when reduction==0).
 * Returns nvoxels (>=0) or -1 on allocation failure. */
int64_t oracle_voxelize_3d_dense(
    const float *points, int64_t n, int32_t c,
    const int32_t *shape, const float *bound,
    int32_t max_points, int32_t max_voxels, int32_t reduction,
    float *voxels, int64_t *coords, uint8_t *pmask, int32_t *npoints, float *aggregates)
{
    memset(voxels, 0, sizeof(float) * (size_t)max_voxels * max_points * c);
    memset(pmask, 0, (size_t)max_voxels * max_points);
    memset(npoints, 0, sizeof(int32_t) * (size_t)max_voxels);
    if (reduction != 0) {
        /* :66-81 */
        float init = reduction == 1 ? 0.0f : (reduction == 2 ? -INFINITY : INFINITY);
        for (size_t i = 0; i < (size_t)max_voxels * c; i++) aggregates[i] = init;
    }

    /* :84-86  float(hi - lo) / int -> float */
    float size[3];
    for (int d = 0; d < 3; d++)
        size[d] = (bound[(d << 1) | 1] - bound[d << 1]) / (float)shape[d];

    cmap_t map;
    if (cmap_init(&map, 1024)) return -1;
    int32_t nvoxels = 0;
    for (int64_t i = 0; i < n; i++) {
        const float *p = points + i * c;
        int32_t ct[3];
        int out = 0;
        for (int d = 0; d < 3; d++) {
            /* :100-101  C truncation toward zero, then range test */
            int32_t idx = f2i_trunc((p[d] - bound[d << 1]) / size[d]);
            if (idx < 0 || idx >= shape[d]) { out = 1; break; }
            ct[d] = idx;
        }
        if (out) continue;

        /* :111-125 first-seen numbering, capped by max_voxels */
        int32_t v;
        cell_t *cell = cmap_find(&map, ct[0], ct[1], ct[2]);
        if (cell->id < 0) {
            if (nvoxels >= max_voxels) continue;
            v = nvoxels++;
            if (cmap_insert(&map, cell, ct[0], ct[1], ct[2], v)) { free(map.cells); return -1; }
            for (int d = 0; d < 3; d++) coords[(size_t)v * 3 + d] = ct[d];
        } else v = cell->id;

        /* :128-134 copy first max_points points; count everything */
        int32_t k = npoints[v]++;
        if (k < max_points) {
            pmask[(size_t)v * max_points + k] = 1;
            memcpy(voxels + ((size_t)v * max_points + k) * c, p, sizeof(float) * c);
        }
        /* :137-157 reductions run over ALL in-range points */
        float *agg = aggregates ? aggregates + (size_t)v * c : NULL;
        switch (reduction) {
        case 1: for (int d = 0; d < c; d++) agg[d] += p[d]; break;
        case 2: for (int d = 0; d < c; d++) agg[d] = agg[d] < p[d] ? p[d] : agg[d]; break; /* std::max(a,b) = (a<b)?b:a */
        case 3: for (int d = 0; d < c; d++) agg[d] = p[d] < agg[d] ? p[d] : agg[d]; break; /* std::min(a,b) = (b<a)?b:a */
        default: break;
        }
    }
    /* :161-164 */
    if (reduction == 1)
        for (int32_t v = 0; v < nvoxels; v++)
            for (int d = 0; d < c; d++)
                aggregates[(size_t)v * c + d] /= (float)npoints[v];
    free(map.cells);
    return nvoxels;
}

/* ------------------------------------------------------------------------- */
/* voxelize_sparse (voxelize.cpp:288-335; bound to Python as voxelize_3d_sparse,
 * impl.cpp:5).  coords is sized [n*3] by the caller.  Returns nvoxels. */
int64_t oracle_voxelize_3d_sparse(
    const float *points, int64_t n, int32_t c, const float *voxel_size,
    int64_t *points_mapping, int64_t *coords, int32_t *npoints)
{
    cmap_t map;
    if (cmap_init(&map, 1024)) return -1;
    int32_t nvoxels = 0;
    for (int64_t i = 0; i < n; i++) {
        const float *p = points + i * c;
        int32_t ct[3];
        for (int d = 0; d < 3; d++) {
            /* :309  int = std::floor(float / float) */
            float q = floorf(p[d] / voxel_size[d]);
            ct[d] = f2i_trunc(q);
        }
        int32_t v;
        cell_t *cell = cmap_find(&map, ct[0], ct[1], ct[2]);
        if (cell->id < 0) {
            v = nvoxels++;
            if (cmap_insert(&map, cell, ct[0], ct[1], ct[2], v)) { free(map.cells); return -1; }
            npoints[v] = 1;
            for (int d = 0; d < 3; d++) coords[(size_t)v * 3 + d] = ct[d];
        } else { v = cell->id; npoints[v] += 1; }
        points_mapping[i] = v;
    }
    free(map.cells);
    return nvoxels;
}

/* ------------------------------------------------------------------------- */
/* voxelize_filter (voxelize.cpp:337-484; bound as voxelize_3d_filter).
 * max_points_filter: 0 NONE, 1 TRIM (2 FARTHEST_SAMPLING throws, :469-471 -> -2)
 * max_voxels_filter: 0 NONE, 1 TRIM, 2 DESCENDING.
 * DESCENDING: the reference uses an unstable torch::argsort (:406); the spec here
 * is a STABLE descending order (ties keep ascending voxel id).
 * Outputs (caller-sized): out_feats[n*c], out_mask[n] (ascending kept point idx),
 * out_mapping[n], out_npoints[nvox], out_coords[nvox*3].
 * counts[0] = kept points, counts[1] = kept voxels.  Returns 0, or <0 on error. */
typedef struct { int32_t cnt; int32_t id; } cntid_t;
static int cmp_desc_stable(const void *a, const void *b)
{
    const cntid_t *x = (const cntid_t *)a, *y = (const cntid_t *)b;
    if (x->cnt != y->cnt) return x->cnt > y->cnt ? -1 : 1;
    return x->id < y->id ? -1 : (x->id > y->id ? 1 : 0);
}

int32_t oracle_voxelize_3d_filter(
    const float *feats, int64_t n, int32_t c,
    const int64_t *points_mapping, const int64_t *coords, const int32_t *voxel_npoints, int64_t nvox,
    const int64_t *coords_bound /* [3][2] */,
    int32_t min_points, int32_t max_points, int32_t max_voxels,
    int32_t max_points_filter, int32_t max_voxels_filter,
    float *out_feats, int64_t *out_mask, int64_t *out_mapping,
    int32_t *out_npoints, int64_t *out_coords, int64_t *counts)
{
    if (max_points_filter == 2) return -2;
    if (max_points_filter < 0 || max_points_filter > 2 || max_voxels_filter < 0 || max_voxels_filter > 2) return -3;
    int32_t *newid = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nvox > 0 ? nvox : 1));
    if (!newid) return -1;
    for (int64_t i = 0; i < nvox; i++) newid[i] = -1;
    int32_t kept = 0;

#define OUT_OF_BOUND(i) ( \
        coords[(i) * 3 + 0] < coords_bound[0] || coords[(i) * 3 + 0] >= coords_bound[1] || \
        coords[(i) * 3 + 1] < coords_bound[2] || coords[(i) * 3 + 1] >= coords_bound[3] || \
        coords[(i) * 3 + 2] < coords_bound[4] || coords[(i) * 3 + 2] >= coords_bound[5])

    switch (max_voxels_filter) {
    case 0: /* :381-390 (max_voxels ignored) */
        for (int64_t i = 0; i < nvox; i++) {
            if (voxel_npoints[i] < min_points) continue;
            if (OUT_OF_BOUND(i)) continue;
            newid[i] = kept++;
        }
        break;
    case 1: /* :392-403 */
        for (int64_t i = 0; i < nvox; i++) {
            if (kept >= max_voxels) break;
            if (voxel_npoints[i] < min_points) continue;
            if (OUT_OF_BOUND(i)) continue;
            newid[i] = kept++;
        }
        break;
    case 2: { /* :405-418 */
        cntid_t *ord = (cntid_t *)malloc(sizeof(cntid_t) * (size_t)(nvox > 0 ? nvox : 1));
        if (!ord) { free(newid); return -1; }
        for (int64_t i = 0; i < nvox; i++) { ord[i].cnt = voxel_npoints[i]; ord[i].id = (int32_t)i; }
        qsort(ord, (size_t)nvox, sizeof(cntid_t), cmp_desc_stable);
        for (int64_t k = 0; k < nvox; k++) {
            int64_t i = ord[k].id;
            if (kept >= max_voxels) break;
            if (voxel_npoints[i] < min_points) break;
            if (OUT_OF_BOUND(i)) continue;
            newid[i] = kept++;
        }
        free(ord);
        break; }
    }
#undef OUT_OF_BOUND

    /* :422-426 */
    for (int64_t i = 0; i < nvox; i++)
        if (newid[i] >= 0)
            for (int d = 0; d < 3; d++) out_coords[(size_t)newid[i] * 3 + d] = coords[i * 3 + d];
    for (int32_t v = 0; v < kept; v++) out_npoints[v] = 0;

    /* :434-467 then :474-476 (where + index_select) */
    int64_t np = 0;
    for (int64_t i = 0; i < n; i++) {
        int64_t old = points_mapping[i];
        int32_t vid = (old >= 0 && old < nvox) ? newid[old] : -1;
        if (vid < 0) continue;
        if (max_points_filter == 1 && out_npoints[vid] >= max_points) continue;
        out_npoints[vid]++;
        out_mask[np] = i;
        out_mapping[np] = vid;
        memcpy(out_feats + (size_t)np * c, feats + (size_t)i * c, sizeof(float) * c);
        np++;
    }
    counts[0] = np; counts[1] = kept;
    free(newid);
    return 0;
}
