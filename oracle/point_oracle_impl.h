/* oracle/point_oracle_impl.h -- TEST INFRASTRUCTURE (included by point_oracle.c for float and double).
 * Restatement of reference d3d/point/scatter.cpp: _floor/_ceil (:22-33), _fill_lcoords (:34-79),
 * aligned_scatter_forward_templated (:81-141), aligned_scatter_backward_templated (:143-180). */

static int FN(fl)(T v) { int i = (int)v; return (i > v) ? i - 1 : i; }   /* scatter.cpp:22-27 */
static int FN(ce)(T v) { int i = (int)v; return (i < v) ? i + 1 : i; }   /* scatter.cpp:28-33 */

/* neighbours of one coordinate row; atype 1 = MEAN, 2 = LINEAR (scatter.h:37) */
static void FN(fill)(const int64_t *dims, int dim, int atype, const T *coord, int lcoord[8][3], T lw[8])
{
    const int nb = 1 << dim;
    if (atype == 2) for (int j = 0; j < nb; j++) lw[j] = 1;
    for (int j = 0; j < nb; j++)
        for (int d = 0; d < dim; d++) {
            const int dmin = 0, dmax = (int)dims[d] - 1;
            const T dc = coord[d + 1];
            if (dc > dmax) { lcoord[j][d] = dmax; if (atype == 2) lw[j] *= (T)0.5; }
            else if (dc < dmin) { lcoord[j][d] = dmin; if (atype == 2) lw[j] *= (T)0.5; }
            else if (j & (1u << d)) { lcoord[j][d] = FN(ce)(dc); if (atype == 2) lw[j] *= 1 + dc - lcoord[j][d]; }
            else { lcoord[j][d] = FN(fl)(dc); if (atype == 2) lw[j] *= 1 - dc + lcoord[j][d]; }
        }
}

static int64_t FN(off)(const int64_t *dims, int dim, const int *lc)
{
    int64_t o = 0;
    for (int d = 0; d < dim; d++) o = o * dims[d] + lc[d];
    return o;
}

/* coord[n, dim+1] (batch index first), image[B, C, D1..Ddim], out[n, C] */
void FN(oracle_aligned_scatter_forward)(const T *coord, int64_t n, int dim, const T *image, int64_t C, const int64_t *dims,
                                        int atype, T *out)
{
    int64_t vol = 1;
    for (int d = 0; d < dim; d++) vol *= dims[d];
    const int nb = 1 << dim;
    for (int64_t i = 0; i < n; i++) {
        const T *cr = coord + i * (dim + 1);
        const int b = (int)cr[0];
        int lc[8][3];
        T lw[8];
        FN(fill)(dims, dim, atype, cr, lc, lw);
        for (int64_t c = 0; c < C; c++) {
            const T *img = image + ((int64_t)b * C + c) * vol;
            T sum = 0;
            if (atype == 1) { for (int j = 0; j < nb; j++) sum += img[FN(off)(dims, dim, lc[j])]; out[i * C + c] = sum / nb; }
            else { for (int j = 0; j < nb; j++) sum += img[FN(off)(dims, dim, lc[j])] * lw[j]; out[i * C + c] = sum; }
        }
    }
}

/* image_grad[B, C, D...] += ... (sequential: the reference's parallel_for version races, scatter.cpp:150-178) */
void FN(oracle_aligned_scatter_backward)(const T *coord, int64_t n, int dim, const T *grad, int64_t C, const int64_t *dims,
                                         int atype, T *image_grad)
{
    int64_t vol = 1;
    for (int d = 0; d < dim; d++) vol *= dims[d];
    const int nb = 1 << dim;
    for (int64_t i = 0; i < n; i++) {
        const T *cr = coord + i * (dim + 1);
        const int b = (int)cr[0];
        int lc[8][3];
        T lw[8];
        FN(fill)(dims, dim, atype, cr, lc, lw);
        for (int64_t c = 0; c < C; c++) {
            T *img = image_grad + ((int64_t)b * C + c) * vol;
            for (int j = 0; j < nb; j++) {
                if (atype == 1) img[FN(off)(dims, dim, lc[j])] += grad[i * C + c] / nb;
                else img[FN(off)(dims, dim, lc[j])] += grad[i * C + c] * lw[j];
            }
        }
    }
}
