/*
 * oracle/box_oracle_impl.h -- TEST INFRASTRUCTURE (included twice by box_oracle.c
 * with T = float and T = double; FN(name) appends _f32 / _f64).
 *
 * The geometry the reference gets from the un-vendored third-party header
 * dgal/geometry.hpp (cmpute/dgal, submodule unpinned; reference .gitmodules:4-6)
 * is restated here from its published behaviour: rectangle corners from
 * (x,y,w,h,r), convex-polygon clipping (Sutherland-Hodgman), shoelace areas,
 * IoU = A_int / (A1 + A2 - A_int).  Call sites followed: reference
 * d3d/box/utils.h:15-34 (make_box), d3d/box/iou.cpp:12-46,95-141,
 * d3d/box/nms.cpp:10-119, d3d/dgal_wrap.h:45-91.
 */

typedef struct { T x, y; } FN(pt);
typedef struct { FN(pt) v[4]; } FN(quad);
typedef struct { T xmin, xmax, ymin, ymax; } FN(aabox);

/* dgal::poly2_from_xywhr (call site utils.h:19).  w spans the local x axis,
 * h the local y axis, r = CCW angle of the local x axis; corners CCW starting
 * at local (-w/2,-h/2)  (SURVEY.md App. D convention, pinned by test_box.py). */
static FN(quad) FN(quad_from_xywhr)(T x, T y, T w, T h, T r)
{
    T s = SIN(r), c = COS(r);
    T dxs = w * s / 2, dxc = w * c / 2;
    T dys = h * s / 2, dyc = h * c / 2;
    FN(quad) q;
    q.v[0].x = x - dxc + dys; q.v[0].y = y - dxs - dyc;
    q.v[1].x = x + dxc + dys; q.v[1].y = y + dxs - dyc;
    q.v[2].x = x + dxc - dys; q.v[2].y = y + dxs + dyc;
    q.v[3].x = x - dxc - dys; q.v[3].y = y - dxs + dyc;
    return q;
}

/* dgal::aabox2_from_poly2 (call site utils.h:32) */
static FN(aabox) FN(aabox_from_quad)(const FN(quad) *q)
{
    FN(aabox) a = { q->v[0].x, q->v[0].x, q->v[0].y, q->v[0].y };
    for (int k = 1; k < 4; k++) {
        if (q->v[k].x < a.xmin) a.xmin = q->v[k].x;
        if (q->v[k].x > a.xmax) a.xmax = q->v[k].x;
        if (q->v[k].y < a.ymin) a.ymin = q->v[k].y;
        if (q->v[k].y > a.ymax) a.ymax = q->v[k].y;
    }
    return a;
}

/* dgal::iou(AABox2, AABox2) (call site iou.cpp:30) */
static T FN(iou_aabox)(const FN(aabox) *a, const FN(aabox) *b)
{
    T ix = (a->xmax < b->xmax ? a->xmax : b->xmax) - (a->xmin > b->xmin ? a->xmin : b->xmin);
    T iy = (a->ymax < b->ymax ? a->ymax : b->ymax) - (a->ymin > b->ymin ? a->ymin : b->ymin);
    if (!(ix > 0) || !(iy > 0)) return 0;
    T inter = ix * iy;
    T a1 = (a->xmax - a->xmin) * (a->ymax - a->ymin);
    T a2 = (b->xmax - b->xmin) * (b->ymax - b->ymin);
    return inter / (a1 + a2 - inter);
}

static T FN(poly_area)(const FN(pt) *p, int n)
{
    if (n < 3) return 0;
    T s = 0;
    for (int k = 0; k < n; k++) {
        const FN(pt) *a = &p[k], *b = &p[(k + 1) % n];
        s += a->x * b->y - b->x * a->y;
    }
    return s / 2;
}

/* Sutherland-Hodgman: clip subject polygon (<=8 verts) by the CCW convex quad c.
 * A vertex on the clip line counts as inside (identical boxes -> full overlap). */
static int FN(clip_quad)(const FN(quad) *subj, const FN(quad) *c, FN(pt) *out)
{
    FN(pt) buf[2][16];
    int n = 4, cur = 0;
    for (int k = 0; k < 4; k++) buf[0][k] = subj->v[k];
    for (int e = 0; e < 4 && n > 0; e++) {
        FN(pt) a = c->v[e], b = c->v[(e + 1) & 3];
        T ex = b.x - a.x, ey = b.y - a.y;
        FN(pt) *in = buf[cur], *o = buf[cur ^ 1];
        int m = 0;
        for (int k = 0; k < n; k++) {
            FN(pt) p = in[k], q = in[(k + 1) % n];
            T dp = ex * (p.y - a.y) - ey * (p.x - a.x);
            T dq = ex * (q.y - a.y) - ey * (q.x - a.x);
            int pin = dp >= 0, qin = dq >= 0;
            if (pin) o[m++] = p;
            if (pin != qin) {
                T t = dp / (dp - dq);
                FN(pt) x = { p.x + t * (q.x - p.x), p.y + t * (q.y - p.y) };
                o[m++] = x;
            }
        }
        n = m; cur ^= 1;
    }
    for (int k = 0; k < n; k++) out[k] = buf[cur][k];
    return n;
}

/* dgal::iou(Quad2, Quad2) (call sites iou.cpp:116, nms.cpp:51, dgal_wrap.h:50) */
static T FN(iou_quad)(const FN(quad) *a, const FN(quad) *b)
{
    FN(pt) poly[16];
    T a1 = FN(poly_area)(a->v, 4), a2 = FN(poly_area)(b->v, 4);
    /* policy for degenerate (zero / negative size) boxes, SURVEY.md App. D: IoU 0, never NaN/inf */
    if (!(a1 > 0) || !(a2 > 0)) return 0;
    int n = FN(clip_quad)(a, b, poly);
    T inter = FN(poly_area)(poly, n);
    if (!(inter > 0)) return 0;
    return inter / (a1 + a2 - inter);
}

/* one pair; method 1 = BOX (AABB of rotated rect), 2 = RBOX (box/common.h:5-9).
 * make_box is evaluated per pair exactly like iou.cpp:27-28,113-114. */
static T FN(pair_iou)(const T *bi, const T *bj, int method)
{
    FN(quad) qi = FN(quad_from_xywhr)(bi[0], bi[1], bi[2], bi[3], bi[4]);
    FN(quad) qj = FN(quad_from_xywhr)(bj[0], bj[1], bj[2], bj[3], bj[4]);
    if (method == 1) {
        FN(aabox) ai = FN(aabox_from_quad)(&qi), aj = FN(aabox_from_quad)(&qj);
        return FN(iou_aabox)(&ai, &aj);
    }
    return FN(iou_quad)(&qi, &qj);
}

/* iou2d_forward / iou2dr_forward (iou.cpp:12-46, 95-141): ious[N,M] row-major.
 * Rows [row_begin,row_end) only, so callers can split across threads the way
 * at::parallel_for does (iou.cpp:21,106). */
void FN(oracle_iou2d)(const T *b1, int64_t n, const T *b2, int64_t m, int method,
                      int64_t row_begin, int64_t row_end, T *ious)
{
    (void)n;
    for (int64_t i = row_begin; i < row_end; i++)
        for (int64_t j = 0; j < m; j++)
            ious[i * m + j] = FN(pair_iou)(b1 + i * 5, b2 + j * 5, method);
}

/* ---- loss-path functions: GIoU / DIoU of rotated boxes and the autograd bookkeeping (reference iou.cpp:213-419, the
 * dgal calls giou / diou / iou(.., nx, flags) at iou.cpp:116,225,334).  dgal's source is not available; the PUBLISHED
 * definitions are restated:  GIoU = IoU - (H - U) / H with H = area of the convex hull of the two rectangles and
 * U = A1 + A2 - I;  DIoU = IoU - d^2 / D^2 with d = distance of the centres and D = diameter of that hull (largest
 * distance between two of the eight corners).  A rectangle of non-positive area gives 0 (never NaN), like IoU. */

/* convex hull of 8 points: Andrew's monotone chain (collinear points dropped) -> vertex indices CCW, starting at the
 * lowest (x, y) point; returns the vertex count */
static int FN(hull8)(const FN(pt) *p, uint8_t *hull)
{
    int idx[8], n = 8;
    for (int k = 0; k < n; k++) idx[k] = k;
    for (int a = 1; a < n; a++) {                     /* insertion sort by (x, y, index) */
        int v = idx[a], b = a - 1;
        while (b >= 0 && (p[idx[b]].x > p[v].x || (p[idx[b]].x == p[v].x && p[idx[b]].y > p[v].y))) { idx[b + 1] = idx[b]; b--; }
        idx[b + 1] = v;
    }
    int st[17], m = 0;
    for (int k = 0; k < n; k++) {                     /* lower hull */
        while (m >= 2) {
            const FN(pt) *a = &p[st[m - 2]], *b = &p[st[m - 1]], *c = &p[idx[k]];
            if ((b->x - a->x) * (c->y - a->y) - (b->y - a->y) * (c->x - a->x) <= 0) m--; else break;
        }
        st[m++] = idx[k];
    }
    const int lower = m + 1;
    for (int k = n - 2; k >= 0; k--) {                /* upper hull */
        while (m >= lower) {
            const FN(pt) *a = &p[st[m - 2]], *b = &p[st[m - 1]], *c = &p[idx[k]];
            if ((b->x - a->x) * (c->y - a->y) - (b->y - a->y) * (c->x - a->x) <= 0) m--; else break;
        }
        st[m++] = idx[k];
    }
    m--;                                              /* the start point is repeated at the end */
    if (m < 0) m = 0;
    for (int k = 0; k < m && k < 8; k++) hull[k] = (uint8_t)st[k];
    return m < 8 ? m : 8;
}

static T FN(hull_area)(const FN(quad) *a, const FN(quad) *b, int *nm, uint8_t *mflags)
{
    FN(pt) p[8], h[8];
    for (int k = 0; k < 4; k++) { p[k] = a->v[k]; p[4 + k] = b->v[k]; }
    uint8_t hull[8];
    int n = FN(hull8)(p, hull);
    for (int k = 0; k < n; k++) h[k] = p[hull[k]];
    if (nm) *nm = n;
    if (mflags) for (int k = 0; k < 8; k++) mflags[k] = k < n ? hull[k] : 0xff;
    return FN(poly_area)(h, n);
}

/* largest squared distance between two of the eight corners; i1 < i2 = the first pair that reaches it */
static T FN(diameter2)(const FN(quad) *a, const FN(quad) *b, int *i1, int *i2)
{
    FN(pt) p[8];
    for (int k = 0; k < 4; k++) { p[k] = a->v[k]; p[4 + k] = b->v[k]; }
    T best = -1;
    for (int x = 0; x < 8; x++)
        for (int y = x + 1; y < 8; y++) {
            T dx = p[x].x - p[y].x, dy = p[x].y - p[y].y, d = dx * dx + dy * dy;
            if (d > best) { best = d; if (i1) *i1 = x; if (i2) *i2 = y; }
        }
    return best;
}

/* Sutherland-Hodgman with the origin of every vertex: subject = box 1, clipped by the edges of box 2 in order.
 * flags[k] of vertex k (CCW): 0x00 | i = corner i of box 1;  0x10 | j = corner j of box 2;
 * 0x20 | i << 2 | j = crossing of edge i of box 1 (corner i -> i + 1) with edge j of box 2.  Returns the vertex count. */
static int FN(clip_quad_flags)(const FN(quad) *subj, const FN(quad) *c, FN(pt) *out, uint8_t *flags)
{
    FN(pt) buf[2][16];
    uint8_t vf[2][16], ef[2][16];        /* vertex flag; origin of the edge that STARTS at the vertex: 0..3 box-1 edge, 4..7 box-2 edge */
    int n = 4, cur = 0;
    for (int k = 0; k < 4; k++) { buf[0][k] = subj->v[k]; vf[0][k] = (uint8_t)k; ef[0][k] = (uint8_t)k; }
    for (int e = 0; e < 4 && n > 0; e++) {
        FN(pt) a = c->v[e], b = c->v[(e + 1) & 3];
        T ex = b.x - a.x, ey = b.y - a.y;
        int m = 0, nxt = cur ^ 1;
        for (int k = 0; k < n; k++) {
            FN(pt) p = buf[cur][k], q = buf[cur][(k + 1) % n];
            const uint8_t o = ef[cur][k];
            T dp = ex * (p.y - a.y) - ey * (p.x - a.x);
            T dq = ex * (q.y - a.y) - ey * (q.x - a.x);
            int pin = dp >= 0, qin = dq >= 0;
            if (pin) { buf[nxt][m] = p; vf[nxt][m] = vf[cur][k]; ef[nxt][m] = o; m++; }
            if (pin != qin) {
                T t = dp / (dp - dq);
                FN(pt) x = { p.x + t * (q.x - p.x), p.y + t * (q.y - p.y) };
                uint8_t f;
                if (o < 4) f = (uint8_t)(0x20 | (o << 2) | e);
                else {                                  /* two edges of box 2 meet in one of its corners */
                    const int e2 = o - 4;
                    f = (uint8_t)(0x10 | (((e2 + 1) & 3) == e ? e : e2));
                }
                buf[nxt][m] = x; vf[nxt][m] = f;
                ef[nxt][m] = pin ? (uint8_t)(4 + e) : o;    /* leaving: continue along the clip edge; entering: along p -> q */
                m++;
            }
        }
        n = m; cur = nxt;
    }
    for (int k = 0; k < n && k < 8; k++) { out[k] = buf[cur][k]; flags[k] = vf[cur][k]; }
    for (int k = n; k < 8; k++) flags[k] = 0xff;
    return n < 8 ? n : 8;
}

/* one pair: kind 0 = GIoU (IouType GRBOX), 1 = DIoU (DRBOX) */
static T FN(pair_loss_iou)(const T *bi, const T *bj, int kind)
{
    FN(quad) qi = FN(quad_from_xywhr)(bi[0], bi[1], bi[2], bi[3], bi[4]);
    FN(quad) qj = FN(quad_from_xywhr)(bj[0], bj[1], bj[2], bj[3], bj[4]);
    T a1 = FN(poly_area)(qi.v, 4), a2 = FN(poly_area)(qj.v, 4);
    if (!(a1 > 0) || !(a2 > 0)) return 0;
    FN(pt) poly[16];
    int n = FN(clip_quad)(&qi, &qj, poly);
    T inter = FN(poly_area)(poly, n);
    if (!(inter > 0)) inter = 0;
    T uni = a1 + a2 - inter, iou = inter / uni;
    if (kind == 0) {
        T hull = FN(hull_area)(&qi, &qj, 0, 0);
        return iou - (hull - uni) / hull;
    }
    T dx = bi[0] - bj[0], dy = bi[1] - bj[1];
    return iou - (dx * dx + dy * dy) / FN(diameter2)(&qi, &qj, 0, 0);
}

void FN(oracle_loss_iou2dr)(const T *b1, int64_t n, const T *b2, int64_t m, int kind, int64_t row_begin, int64_t row_end, T *out)
{
    (void)n;
    for (int64_t i = row_begin; i < row_end; i++)
        for (int64_t j = 0; j < m; j++) out[i * m + j] = FN(pair_loss_iou)(b1 + i * 5, b2 + j * 5, kind);
}

/* bookkeeping outputs of iou2dr_forward / giou2dr_forward / diou2dr_forward (iou.cpp:115-141, 224-258, 333-367):
 * nx[n,m] intersection vertex count, xflags[n,m,8] (see clip_quad_flags), nm[n,m] hull vertex count, mflags[n,m,8]
 * = corner index (0..3 box 1, 4..7 box 2) of every hull vertex, far[n,m,2] = the farthest corner pair.  Any may be NULL. */
void FN(oracle_iou2dr_flags)(const T *b1, int64_t n, const T *b2, int64_t m, uint8_t *nx, uint8_t *xflags, uint8_t *nm,
                             uint8_t *mflags, uint8_t *far)
{
    for (int64_t i = 0; i < n; i++)
        for (int64_t j = 0; j < m; j++) {
            const T *bi = b1 + i * 5, *bj = b2 + j * 5;
            FN(quad) qi = FN(quad_from_xywhr)(bi[0], bi[1], bi[2], bi[3], bi[4]);
            FN(quad) qj = FN(quad_from_xywhr)(bj[0], bj[1], bj[2], bj[3], bj[4]);
            FN(pt) poly[16];
            uint8_t fl[8], mf[8];
            int k = FN(clip_quad_flags)(&qi, &qj, poly, fl), hm = 0, f1 = 0, f2 = 0;
            if (k < 3) { k = 0; for (int t = 0; t < 8; t++) fl[t] = 0xff; }       /* no area: no polygon */
            FN(hull_area)(&qi, &qj, &hm, mf);
            FN(diameter2)(&qi, &qj, &f1, &f2);
            const int64_t e = i * m + j;
            if (nx) nx[e] = (uint8_t)k;
            if (xflags) for (int t = 0; t < 8; t++) xflags[e * 8 + t] = fl[t];
            if (nm) nm[e] = (uint8_t)hm;
            if (mflags) for (int t = 0; t < 8; t++) mflags[e * 8 + t] = mf[t];
            if (far) { far[e * 2] = (uint8_t)f1; far[e * 2 + 1] = (uint8_t)f2; }
        }
}

/* pdist2dr_forward (dist.cpp:10-52; dgal::distance(Quad2, Point2, iedge)): SIGNED distance from a point to the boundary of
 * a rotated box, positive inside (the convention box3dr_pdist relies on, box/__init__.py:370-381); iedge = the nearest
 * edge k (corner k -> k + 1), or 4 + k when the nearest boundary point is corner k.  dist[m,n], iedge[m,n]: box-major. */
void FN(oracle_pdist2dr)(const T *points, int64_t n, const T *boxes, int64_t m, T *dist, uint8_t *iedge)
{
    for (int64_t i = 0; i < m; i++) {
        const T *b = boxes + i * 5;
        FN(quad) q = FN(quad_from_xywhr)(b[0], b[1], b[2], b[3], b[4]);
        for (int64_t j = 0; j < n; j++) {
            T px = points[j * 2], py = points[j * 2 + 1], best = -1;
            int inside = 1, feat = 0;
            for (int e = 0; e < 4; e++) {
                FN(pt) v = q.v[e], w = q.v[(e + 1) & 3];
                T ex = w.x - v.x, ey = w.y - v.y, rx = px - v.x, ry = py - v.y;
                if (ex * ry - ey * rx < 0) inside = 0;
                T len2 = ex * ex + ey * ey, t = len2 > 0 ? (rx * ex + ry * ey) / len2 : 0;
                int f = e;
                if (t <= 0) { t = 0; f = 4 + e; } else if (t >= 1) { t = 1; f = 4 + ((e + 1) & 3); }
                T dx = rx - t * ex, dy = ry - t * ey, d2 = dx * dx + dy * dy;
                if (best < 0 || d2 < best) { best = d2; feat = f; }
            }
            T d = (T)sqrt((double)best);
            dist[i * n + j] = inside ? d : -d;
            if (iedge) iedge[i * n + j] = (uint8_t)feat;
        }
    }
}

/* the same per-pair arithmetic on a LIST of pairs (pi[k], pj[k]) -> out[k]: lets a test check a matrix far too large
 * to recompute densely on the CPU (config 3: 1e10 pairs) at every candidate pair of a CPU-side AABB sweep */
void FN(oracle_iou2d_pairs)(const T *b1, const T *b2, const int64_t *pi, const int64_t *pj, int64_t k, int method, T *out)
{
    for (int64_t t = 0; t < k; t++) out[t] = FN(pair_iou)(b1 + pi[t] * 5, b2 + pj[t] * 5, method);
}

/* nms2d + nms2d_templated (nms.cpp:10-119).  order must be the descending
 * argsort of scores (nms.cpp:103; STABLE here).  scores is copied (nms.cpp:104).
 * supp: 0 HARD, 1 LINEAR, 2 GAUSSIAN.  Thresholds are C floats compared against
 * scalar_t (nms.h:9-10, nms.cpp:26,53). Writes suppressed[N] (u8). */
void FN(oracle_nms2d)(const T *boxes, const T *scores_in, int64_t n_, const int64_t *order_in,
                      int method, int supp, float iou_threshold, float score_threshold,
                      float supp_param, uint8_t *suppressed)
{
    const int N = (int)n_;
    T *scores = (T *)malloc(sizeof(T) * (size_t)(N > 0 ? N : 1));
    int64_t *order = (int64_t *)malloc(sizeof(int64_t) * (size_t)(N > 0 ? N : 1));
    for (int i = 0; i < N; i++) { scores[i] = scores_in[i]; order[i] = order_in[i]; suppressed[i] = 0; }

    /* nms.cpp:23-29  (never touches sorted position 0) */
    for (int _i = N - 1; _i > 0; _i--) {
        int i = (int)order[_i];
        if (scores[i] > score_threshold) break;
        suppressed[i] = 1;
    }
    /* nms.cpp:32-95 */
    for (int _i = 0; _i < N; _i++) {
        int i = (int)order[_i];
        if (suppressed[i]) {
            if (supp == 0) continue;
            else break;
        }
        for (int _j = _i + 1; _j < N; _j++) {
            int j = (int)order[_j];
            if (supp == 0 && suppressed[j]) continue;
            T iou = FN(pair_iou)(boxes + (size_t)i * 5, boxes + (size_t)j * 5, method);
            if (iou > iou_threshold) {
                switch (supp) {
                case 0: suppressed[j] = 1; break;
                case 1: scores[j] *= 1 - POW(iou, supp_param);
                        suppressed[j] = scores[j] < score_threshold; break;
                case 2: scores[j] *= EXP(-iou * iou / supp_param);
                        suppressed[j] = scores[j] < score_threshold; break;
                }
            }
        }
        if (supp != 0) {
            /* nms.cpp:74-94 re-sort the tail, suppressed entries sink */
            int S = N - 1;
            while (S > _i && !suppressed[order[S]]) S--;
            for (int _j = S - 1; _j > _i; _j--) {
                int j = (int)order[_j];
                int _k = _j + 1;
                while (_k < S && (suppressed[j] || scores[order[_k]] > scores[j])) {
                    order[_k - 1] = order[_k];
                    _k++;
                }
                order[_k - 1] = j;
            }
        }
    }
    free(scores); free(order);
}

/* Soft NMS of oracle_nms2d (nms.cpp:60-94) with the inner loop restricted to candidate pairs: a pair whose bounding boxes do
 * not even touch has IoU 0, which never exceeds a threshold >= 0, so rescaling only the neighbours of i that stand at a later
 * position gives the SAME scores, order and suppressed mask -- the updates of different j are independent of each other.
 * The insertion pass is the literal one (O(n) per round: plain integer work, seconds at 100 k boxes); pos[] follows the
 * boxes through it.  Equality with oracle_nms2d is itself tested (tests/test_oracle_box.py). */
void FN(oracle_nms2d_soft_candidates)(const T *boxes, const T *scores_in, int64_t n_, const int64_t *order_in,
                                      const int64_t *nbr_off, const int64_t *nbr, int method, int supp, float iou_threshold,
                                      float score_threshold, float supp_param, uint8_t *suppressed)
{
    const int N = (int)n_;
    T *scores = (T *)malloc(sizeof(T) * (size_t)(N > 0 ? N : 1));
    int64_t *order = (int64_t *)malloc(sizeof(int64_t) * (size_t)(N > 0 ? N : 1));
    int64_t *pos = (int64_t *)malloc(sizeof(int64_t) * (size_t)(N > 0 ? N : 1));
    for (int i = 0; i < N; i++) { scores[i] = scores_in[i]; order[i] = order_in[i]; suppressed[i] = 0; }
    for (int p = 0; p < N; p++) pos[order[p]] = p;
    for (int _i = N - 1; _i > 0; _i--) {                     /* nms.cpp:23-29 */
        int i = (int)order[_i];
        if (scores[i] > score_threshold) break;
        suppressed[i] = 1;
    }
    for (int _i = 0; _i < N; _i++) {
        int i = (int)order[_i];
        if (suppressed[i]) break;                            /* nms.cpp:38 (soft) */
        for (int64_t t = nbr_off[i]; t < nbr_off[i + 1]; t++) {
            int j = (int)nbr[t];
            if (pos[j] <= _i) continue;                      /* only the boxes at later positions (nms.cpp:41) */
            T iou = FN(pair_iou)(boxes + (size_t)i * 5, boxes + (size_t)j * 5, method);
            if (iou > iou_threshold) {
                if (supp == 1) scores[j] *= 1 - POW(iou, supp_param);
                else scores[j] *= EXP(-iou * iou / supp_param);
                suppressed[j] = scores[j] < score_threshold;
            }
        }
        int S = N - 1;                                       /* nms.cpp:74-94 */
        while (S > _i && !suppressed[order[S]]) S--;
        for (int _j = S - 1; _j > _i; _j--) {
            int j = (int)order[_j];
            int _k = _j + 1;
            while (_k < S && (suppressed[j] || scores[order[_k]] > scores[j])) {
                order[_k - 1] = order[_k];
                pos[order[_k - 1]] = _k - 1;
                _k++;
            }
            order[_k - 1] = j;
            pos[j] = _k - 1;
        }
    }
    free(scores); free(order); free(pos);
}

/* Hard NMS of oracle_nms2d restricted to candidate pairs: the inner loop of nms.cpp:41-58 visits every later box j, but a
 * pair whose bounding boxes do not even touch has IoU 0, which never exceeds a threshold >= 0 -- so visiting only the
 * neighbours of i (CSR adjacency nbr_off / nbr from a CPU AABB sweep, any order) gives the SAME suppressed mask.  Lets a
 * test run the reference's greedy loop on config 3's 100 k boxes in seconds instead of minutes; equality with
 * oracle_nms2d is itself tested (tests/test_oracle_box.py).  rank[i] = position of box i in the descending order. */
void FN(oracle_nms2d_hard_candidates)(const T *boxes, const T *scores, int64_t n_, const int64_t *order,
                                      const int64_t *nbr_off, const int64_t *nbr, int method, float iou_threshold,
                                      float score_threshold, uint8_t *suppressed)
{
    const int N = (int)n_;
    int64_t *rank = (int64_t *)malloc(sizeof(int64_t) * (size_t)(N > 0 ? N : 1));
    for (int p = 0; p < N; p++) { rank[order[p]] = p; suppressed[p] = 0; }
    for (int _i = N - 1; _i > 0; _i--) {                     /* nms.cpp:23-29 */
        int i = (int)order[_i];
        if (scores[i] > score_threshold) break;
        suppressed[i] = 1;
    }
    for (int _i = 0; _i < N; _i++) {                         /* nms.cpp:32-59, HARD */
        int i = (int)order[_i];
        if (suppressed[i]) continue;
        for (int64_t t = nbr_off[i]; t < nbr_off[i + 1]; t++) {
            int j = (int)nbr[t];
            if (rank[j] <= _i || suppressed[j]) continue;
            T iou = FN(pair_iou)(boxes + (size_t)i * 5, boxes + (size_t)j * 5, method);
            if (iou > iou_threshold) suppressed[j] = 1;
        }
    }
    free(rank);
}

/* crop_2dr_templated (utils.cpp:9-36): indicators[i][j] = aabox.contains(p_j) && box_i.contains(p_j).
 * dgal's contains() is restated as the closed point-in-convex-polygon test (cross(edge, p - v) >= 0 for the
 * four CCW edges) behind the closed AABB test. */
void FN(oracle_crop_2dr)(const T *points, int64_t n, const T *boxes, int64_t m, uint8_t *out)
{
    for (int64_t i = 0; i < m; i++) {
        const T *b = boxes + i * 5;
        FN(quad) q = FN(quad_from_xywhr)(b[0], b[1], b[2], b[3], b[4]);
        FN(aabox) a = FN(aabox_from_quad)(&q);
        for (int64_t j = 0; j < n; j++) {
            T px = points[j * 2], py = points[j * 2 + 1];
            int in = px >= a.xmin && px <= a.xmax && py >= a.ymin && py <= a.ymax;
            for (int e = 0; e < 4 && in; e++) {
                FN(pt) v = q.v[e], w = q.v[(e + 1) & 3];
                T cr = (w.x - v.x) * (py - v.y) - (w.y - v.y) * (px - v.x);
                if (!(cr >= 0)) in = 0;
            }
            out[i * n + j] = (uint8_t)in;
        }
    }
}
