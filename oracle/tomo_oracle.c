/*
 * tomo_oracle.c -- CPU restatement of the reference's hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the parity oracle: tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg are the only callers.  Nothing under tomo_tv_amd/ may import,
 * link or execute it.  It restates, in plain C on the reference's own data layout
 * (volume [slice][y][z], sinogram [slice][angle*Nray+ray], explicit CSR matrix A),
 * the algorithms of
 *
 *   tomofusion/cpu/utils/pytvlib.py:8-130   parallelRay / rmepsilon        -> orc_parallel_ray
 *   tomofusion/cpu/utils/ctvlib.cpp:309-315 loadA                          -> orc_csr_from_coo
 *   tomofusion/cpu/utils/ctvlib.cpp:101-115,279-293 create/forward proj.   -> orc_forward
 *   tomofusion/cpu/utils/ctvlib.cpp:194-202 lipschits                      -> orc_lipschitz
 *   tomofusion/cpu/utils/ctvlib.cpp:205-231 SIRT (Landweber) + positivity  -> orc_sirt
 *   tomofusion/cpu/utils/ctvlib.cpp:198-199,212-216,245-251 Cimmino branch  -> orc_sirt_cimmino, orc_lipschitz_cimmino
 *   tomofusion/cpu/utils/ctvlib.cpp:137-155,234-242 ART + normalization    -> orc_art, orc_row_inner
 *   tomofusion/cpu/utils/ctvlib.cpp:260-306 matrix_2norm/data_distance/rmse-> orc_sqdiff (+ callers)
 *   tomofusion/cpu/utils/ctvlib.cpp:336-367 tv_3D                          -> orc_tv
 *   tomofusion/cpu/utils/ctvlib.cpp:406-462 tv_gd_3D                       -> orc_tv_gd
 *   tomofusion/gpu/utils/regularizers/tv_fgp.cu:44-115,192-281 FGP-TV      -> orc_tv_fgp
 *   tomofusion/gpu/utils/container/matrix_ops.cu:77-86 +
 *   tomofusion/gpu/utils/tomoengine.cpp:381-384 Nesterov step              -> orc_fista_momentum
 *   tomofusion/gpu/utils/tomoengine.cpp:293-315 poisson_ML                 -> orc_poisson_ml
 *
 * SART and the row/column-normalised SIRT are executed by ASTRA in the reference
 * (tomoengine.cpp:162-205); ASTRA is an un-vendored, un-pinned submodule
 * (.gitmodules:4-6), so those two are PARITY-UNPINNED: they are defined here on the
 * parallelRay matrix by the published SART/SIRT formulas (orc_sart, orc_sirt_norm).
 *
 * Pinning: orc_parallel_ray is checked bit-for-bit against tests/golden/A_*.npz, which
 * tools/gen_golden.py produced by importing the reference's parallelRay in the build
 * container.  Everything else has no reference-held vector (the reference has no tests).
 *
 * Documented deviations from the reference source:
 *   Q1  copy_recon copies the whole volume (reference copies 8 bytes).
 *   Q2  reduction variables start at 0 and accumulate in double (reference: uninitialised float).
 *   Q2b tv_norm is reset for every inner TV iteration.
 *   Q4  TV epsilon is a parameter.
 * Arithmetic on volumes/sinograms is fp32 in the reference's evaluation order
 * (Eigen row-major CSR: inner indices ascending; A^T*v accumulates in ascending row order).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------
 * parallelRay (tomofusion/cpu/utils/pytvlib.py:8-121)
 * Output: coo arrays rows/cols (as int64) and vals (fp32) in generation order.
 * Caller provides capacity 2*N*P*N entries (same bound as pytvlib.py:28-30).
 * Returns nnz, or -1 on overflow.
 * ------------------------------------------------------------------------------------------ */
typedef struct { double t; int idx; } orc_key;

static int orc_key_cmp(const void *pa, const void *pb)
{
    const orc_key *a = (const orc_key *)pa, *b = (const orc_key *)pb;
    /* NaNs last, like numpy argsort (pytvlib.py:58) */
    int an = isnan(a->t), bn = isnan(b->t);
    if (an || bn) { if (an && bn) return (a->idx > b->idx) - (a->idx < b->idx); return an ? 1 : -1; }
    if (a->t < b->t) return -1;
    if (a->t > b->t) return 1;
    return (a->idx > b->idx) - (a->idx < b->idx);
}

static double orc_rmeps(double v) { return fabs(v) < 1e-10 ? 0.0 : v; } /* pytvlib.py:124-130 */

int64_t orc_parallel_ray(int N, int P, const double *angles_deg,
                         int64_t *rows, int64_t *cols, float *vals, int64_t cap)
{
    const int M = N + 1;
    int64_t nnz = 0;
    int overflow = 0;
    double *grid = (double *)malloc(sizeof(double) * M);
    double *offs = (double *)malloc(sizeof(double) * N);
    /* np.linspace(-(N-1)/2, (N-1)/2, N)  and np.linspace(-N/2, N/2, N+1): pytvlib.py:20-24 */
    for (int j = 0; j < N; j++) {
        double start = -((double)N - 1.0) / 2.0, stop = ((double)N - 1.0) / 2.0;
        double step = (N > 1) ? (stop - start) / (double)(N - 1) : 0.0;
        offs[j] = (j == N - 1 && N > 1) ? stop : start + j * step;
    }
    for (int m = 0; m < M; m++) {
        double start = -N * 0.5, stop = N * 0.5, step = (stop - start) / (double)N;
        grid[m] = (m == M - 1) ? stop : start + m * step;
    }
    const double half = N / 2.0;

    /* one angle at a time; rays of an angle in parallel into private buffers so the
       output keeps the reference's (angle, ray, path) generation order */
    int64_t *cnt = (int64_t *)malloc(sizeof(int64_t) * N);
    int64_t *tmp_c = (int64_t *)malloc(sizeof(int64_t) * (size_t)N * 2 * M);
    float *tmp_v = (float *)malloc(sizeof(float) * (size_t)N * 2 * M);

    for (int i = 0; i < P && !overflow; i++) {
        double ang = angles_deg[i] * M_PI / 180.0;                 /* :34 */
        double ca = cos(ang), sa = sin(ang);
        double a = orc_rmeps(-sa), b = orc_rmeps(ca);              /* :41-44 */
#pragma omp parallel
        {
            orc_key *key = (orc_key *)malloc(sizeof(orc_key) * 2 * M);
            double *px = (double *)malloc(sizeof(double) * 2 * M);
            double *py = (double *)malloc(sizeof(double) * 2 * M);
            double *qx = (double *)malloc(sizeof(double) * 2 * M);
            double *qy = (double *)malloc(sizeof(double) * 2 * M);
#pragma omp for schedule(dynamic, 8)
            for (int j = 0; j < N; j++) {
                cnt[j] = 0;
                double x0 = ca * offs[j], y0 = sa * offs[j];       /* :36-37 */
                if (fabs(x0) < 1e-8) x0 = 0.0;                      /* :38-39 */
                if (fabs(y0) < 1e-8) y0 = 0.0;
                /* crossing points with the x-grid then the y-grid: :48-56 */
                for (int m = 0; m < M; m++) {
                    double tx = (grid[m] - x0) / a;
                    key[m].t = tx; key[m].idx = m;
                    px[m] = grid[m]; py[m] = b * tx + y0;
                    double ty = (grid[m] - y0) / b;
                    key[M + m].t = ty; key[M + m].idx = M + m;
                    px[M + m] = a * ty + x0; py[M + m] = grid[m];
                }
                qsort(key, 2 * M, sizeof(orc_key), orc_key_cmp);    /* :58 */
                /* in-grid filter :63-69 (NaN compares false) */
                int n = 0;
                for (int m = 0; m < 2 * M; m++) {
                    double xx = px[key[m].idx], yy = py[key[m].idx];
                    if (xx >= -half && xx <= half && yy >= -half && yy <= half) { qx[n] = xx; qy[n] = yy; n++; }
                }
                if (n == 0) continue;                               /* :72, :113-115 */
                /* drop point k when point k+1 is within 1e-8 of it: :74-79 */
                int n2 = 0;
                for (int k = 0; k < n; k++) {
                    int dup = 0;
                    if (k + 1 < n) dup = (fabs(qx[k + 1] - qx[k]) <= 1e-8) && (fabs(qy[k + 1] - qy[k]) <= 1e-8);
                    if (!dup) { qx[n2] = qx[k]; qy[n2] = qy[k]; n2++; }
                }
                int numvals = n2 - 1;                               /* :82-84 */
                /* boundary-ray rejection :88-92 */
                int check1 = (b == 0.0) && (fabs(y0 - half) < 1e-15);
                int check2 = (a == 0.0) && (fabs(x0 - half) < 1e-15);
                if (numvals <= 0 || check1 || check2) continue;
                int64_t *oc = tmp_c + (size_t)j * 2 * M;
                float *ov = tmp_v + (size_t)j * 2 * M;
                for (int k = 0; k < numvals; k++) {
                    double dx = qx[k + 1] - qx[k], dy = qy[k + 1] - qy[k];
                    double len = sqrt(dx * dx + dy * dy);           /* :82 */
                    double mx = orc_rmeps(0.5 * (qx[k] + qx[k + 1])); /* :98-99 */
                    double my = orc_rmeps(0.5 * (qy[k] + qy[k + 1]));
                    double pix = floor(half - my) * N + floor(mx + half); /* :101-103 */
                    oc[k] = (int64_t)(float)pix;                    /* stored as float32 :29,:111 */
                    ov[k] = (float)len;
                }
                cnt[j] = numvals;
            }
            free(key); free(px); free(py); free(qx); free(qy);
        }
        for (int j = 0; j < N; j++) {
            if (nnz + cnt[j] > cap) { overflow = 1; break; }
            for (int64_t k = 0; k < cnt[j]; k++) {
                rows[nnz] = (int64_t)(float)((double)i * N + j);    /* :110 (float32 storage) */
                cols[nnz] = tmp_c[(size_t)j * 2 * M + k];
                vals[nnz] = tmp_v[(size_t)j * 2 * M + k];
                nnz++;
            }
        }
    }
    free(grid); free(offs); free(cnt); free(tmp_c); free(tmp_v);
    return overflow ? -1 : nnz;
}

/* ------------------------------------------------------------------------------------------
 * loadA (ctvlib.cpp:309-315): A.coeffRef(r,c) = v into an Eigen RowMajor sparse matrix.
 * Result: CSR with ascending column index inside each row; a repeated (r,c) keeps the LAST
 * value.  ptr has nrow+1 entries; idx/val have capacity nnz.  Returns the final nnz.
 * ------------------------------------------------------------------------------------------ */
typedef struct { int64_t c; int64_t seq; float v; } orc_ent;
static int orc_ent_cmp(const void *pa, const void *pb)
{
    const orc_ent *a = (const orc_ent *)pa, *b = (const orc_ent *)pb;
    if (a->c != b->c) return (a->c > b->c) - (a->c < b->c);
    return (a->seq > b->seq) - (a->seq < b->seq);
}

int64_t orc_csr_from_coo(int64_t nrow, int64_t ncol, int64_t nnz,
                         const int64_t *rows, const int64_t *cols, const float *vals,
                         int64_t *ptr, int32_t *idx, float *val)
{
    int64_t *count = (int64_t *)calloc((size_t)nrow + 1, sizeof(int64_t));
    for (int64_t k = 0; k < nnz; k++) {
        if (rows[k] < 0 || rows[k] >= nrow || cols[k] < 0 || cols[k] >= ncol) { free(count); return -1; }
        count[rows[k] + 1]++;
    }
    for (int64_t r = 0; r < nrow; r++) count[r + 1] += count[r];
    orc_ent *ent = (orc_ent *)malloc(sizeof(orc_ent) * (size_t)(nnz > 0 ? nnz : 1));
    int64_t *fill = (int64_t *)malloc(sizeof(int64_t) * (size_t)nrow);
    for (int64_t r = 0; r < nrow; r++) fill[r] = count[r];
    for (int64_t k = 0; k < nnz; k++) {
        int64_t pos = fill[rows[k]]++;
        ent[pos].c = cols[k]; ent[pos].seq = k; ent[pos].v = vals[k];
    }
    int64_t out = 0;
    ptr[0] = 0;
    for (int64_t r = 0; r < nrow; r++) {
        int64_t b = count[r], e = count[r + 1];
        qsort(ent + b, (size_t)(e - b), sizeof(orc_ent), orc_ent_cmp);
        for (int64_t k = b; k < e; k++) {
            if (k + 1 < e && ent[k + 1].c == ent[k].c) continue;   /* later assignment wins */
            idx[out] = (int32_t)ent[k].c; val[out] = ent[k].v; out++;
        }
        ptr[r + 1] = out;
    }
    free(count); free(ent); free(fill);
    return out;
}

/* ------------------------------------------------------------------------------------------
 * Sparse kernels on one slice (fp32, Eigen evaluation order)
 * ------------------------------------------------------------------------------------------ */
static void spmv(int64_t nrow, const int64_t *ptr, const int32_t *idx, const float *val,
                 const float *x, float *y)
{
    for (int64_t r = 0; r < nrow; r++) {
        float acc = 0.0f;
        for (int64_t k = ptr[r]; k < ptr[r + 1]; k++) acc += val[k] * x[idx[k]];
        y[r] = acc;
    }
}

static void spmv_t(int64_t nrow, int64_t ncol, const int64_t *ptr, const int32_t *idx, const float *val,
                   const float *y, float *x)
{
    for (int64_t c = 0; c < ncol; c++) x[c] = 0.0f;
    for (int64_t r = 0; r < nrow; r++) {
        float yr = y[r];
        for (int64_t k = ptr[r]; k < ptr[r + 1]; k++) x[idx[k]] += val[k] * yr;
    }
}

/* forward_projection / create_projections: g(s,i) = A.row(i).dot(x_s)  (ctvlib.cpp:101-115,279-293) */
void orc_forward(int nslice, int64_t nrow, int64_t ncol, const int64_t *ptr, const int32_t *idx,
                 const float *val, const float *vol, float *g)
{
#pragma omp parallel for
    for (int s = 0; s < nslice; s++)
        spmv(nrow, ptr, idx, val, vol + (size_t)s * ncol, g + (size_t)s * nrow);
}

/* back projection of a full sinogram: v_s = A^T r_s (the A.transpose()* of ctvlib.cpp:200,216) */
void orc_back(int nslice, int64_t nrow, int64_t ncol, const int64_t *ptr, const int32_t *idx,
              const float *val, const float *sino, float *vol)
{
#pragma omp parallel for
    for (int s = 0; s < nslice; s++)
        spmv_t(nrow, ncol, ptr, idx, val, sino + (size_t)s * nrow, vol + (size_t)s * ncol);
}

/* lipschits(): (A^T (A 1)).maxCoeff()   (ctvlib.cpp:194-202, Landweber branch) */
float orc_lipschitz(int64_t nrow, int64_t ncol, const int64_t *ptr, const int32_t *idx, const float *val)
{
    float *one = (float *)malloc(sizeof(float) * (size_t)ncol);
    float *t = (float *)malloc(sizeof(float) * (size_t)nrow);
    float *u = (float *)malloc(sizeof(float) * (size_t)ncol);
    for (int64_t c = 0; c < ncol; c++) one[c] = 1.0f;
    spmv(nrow, ptr, idx, val, one, t);
    spmv_t(nrow, ncol, ptr, idx, val, t, u);
    float m = u[0];
    for (int64_t c = 1; c < ncol; c++) if (u[c] > m) m = u[c];
    free(one); free(t); free(u);
    return m;
}

/* positivity(): ctvlib.cpp:224-231 */
void orc_positivity(int64_t n, float *vol)
{
#pragma omp parallel for
    for (int64_t i = 0; i < n; i++) if (vol[i] < 0.0f) vol[i] = 0.0f;
}

/* SIRT(beta), Landweber branch + positivity: ctvlib.cpp:205-221 */
void orc_sirt(int nslice, int64_t nrow, int64_t ncol, const int64_t *ptr, const int32_t *idx,
              const float *val, const float *b, float *vol, float beta, int niter)
{
    for (int it = 0; it < niter; it++) {
#pragma omp parallel
        {
            float *t = (float *)malloc(sizeof(float) * (size_t)nrow);
            float *u = (float *)malloc(sizeof(float) * (size_t)ncol);
#pragma omp for
            for (int s = 0; s < nslice; s++) {
                float *x = vol + (size_t)s * ncol;
                const float *bs = b + (size_t)s * nrow;
                spmv(nrow, ptr, idx, val, x, t);
                for (int64_t r = 0; r < nrow; r++) t[r] = bs[r] - t[r];
                spmv_t(nrow, ncol, ptr, idx, val, t, u);
                for (int64_t c = 0; c < ncol; c++) x[c] += u[c] * beta;
            }
            free(t); free(u);
        }
        orc_positivity((int64_t)nslice * ncol, vol);
    }
}

/* SIRT(beta), Cimmino branch + positivity: ctvlib.cpp:212-216 with M = diag(A.row(i).dot(A.row(i))) from
 * cimminos_method() (ctvlib.cpp:245-251).  Eigen evaluates  A^T * M * r * (beta/Nrow)  left to right: the sparse
 * product (A^T M) has entries a_ij * m_i (rounded to fp32), applied to r, the result scaled by beta/Nrow.
 * (M multiplies by the row norms where Cimmino's method divides: quirk Q10, reproduced as written.) */
void orc_sirt_cimmino(int nslice, int64_t nrow, int64_t ncol, const int64_t *ptr, const int32_t *idx,
                      const float *val, const float *inner, const float *b, float *vol, float beta, int niter)
{
    const float scale = beta / (float)nrow;
    for (int it = 0; it < niter; it++) {
#pragma omp parallel
        {
            float *t = (float *)malloc(sizeof(float) * (size_t)nrow);
            float *u = (float *)malloc(sizeof(float) * (size_t)ncol);
#pragma omp for
            for (int s = 0; s < nslice; s++) {
                float *x = vol + (size_t)s * ncol;
                const float *bs = b + (size_t)s * nrow;
                spmv(nrow, ptr, idx, val, x, t);
                for (int64_t r = 0; r < nrow; r++) t[r] = bs[r] - t[r];
                for (int64_t c = 0; c < ncol; c++) u[c] = 0.0f;
                for (int64_t r = 0; r < nrow; r++)
                    for (int64_t k = ptr[r]; k < ptr[r + 1]; k++) u[idx[k]] += (val[k] * inner[r]) * t[r];
                for (int64_t c = 0; c < ncol; c++) x[c] += u[c] * scale;
            }
            free(t); free(u);
        }
        orc_positivity((int64_t)nslice * ncol, vol);
    }
}

/* lipschits(), Cimmino branch: (A^T * M * (A 1)).maxCoeff()   ctvlib.cpp:198-199 */
float orc_lipschitz_cimmino(int64_t nrow, int64_t ncol, const int64_t *ptr, const int32_t *idx, const float *val,
                            const float *inner)
{
    float *one = (float *)malloc(sizeof(float) * (size_t)ncol);
    float *t = (float *)malloc(sizeof(float) * (size_t)nrow);
    float *u = (float *)calloc((size_t)ncol, sizeof(float));
    for (int64_t c = 0; c < ncol; c++) one[c] = 1.0f;
    spmv(nrow, ptr, idx, val, one, t);
    for (int64_t r = 0; r < nrow; r++)
        for (int64_t k = ptr[r]; k < ptr[r + 1]; k++) u[idx[k]] += (val[k] * inner[r]) * t[r];
    float m = u[0];
    for (int64_t c = 1; c < ncol; c++) if (u[c] > m) m = u[c];
    free(one); free(t); free(u);
    return m;
}

/* normalization(): innerProduct(i) = A.row(i).dot(A.row(i))  ctvlib.cpp:234-242 */
void orc_row_inner(int64_t nrow, const int64_t *ptr, const float *val, float *inner)
{
    for (int64_t r = 0; r < nrow; r++) {
        float acc = 0.0f;
        for (int64_t k = ptr[r]; k < ptr[r + 1]; k++) acc += val[k] * val[k];
        inner[r] = acc;
    }
}

/* ART(beta): ctvlib.cpp:137-155.  Rows in natural order; clamp after the full sweep.
 * A row with zero inner product (empty ray) would divide by zero in the reference; it is skipped. */
void orc_art(int nslice, int64_t nrow, int64_t ncol, const int64_t *ptr, const int32_t *idx,
             const float *val, const float *inner, const float *b, float *vol, float beta, const int32_t *order)
{
#pragma omp parallel for
    for (int s = 0; s < nslice; s++) {
        float *x = vol + (size_t)s * ncol;
        const float *bs = b + (size_t)s * nrow;
        for (int64_t q = 0; q < nrow; q++) {
            int64_t j = order ? order[q] : q;          /* randART as a permutation sweep (ctvlib.cpp:158-179, quirk Q9) */
            if (!(inner[j] > 0.0f)) continue;
            float dot = 0.0f;
            for (int64_t k = ptr[j]; k < ptr[j + 1]; k++) dot += val[k] * x[idx[k]];
            float a = (bs[j] - dot) / inner[j];
            for (int64_t k = ptr[j]; k < ptr[j + 1]; k++) x[idx[k]] += val[k] * a * beta;
        }
    }
    orc_positivity((int64_t)nslice * ncol, vol);
}

/* Sum of squared differences (double accumulator).  Used by:
 *   matrix_2norm  = sqrt(sum)                          ctvlib.cpp:260-269 / tomoengine.cpp:407
 *   data_distance = sqrt(sum)/size (CPU) or sqrt(sum)  ctvlib.cpp:272-276 / tomoengine.cpp:410-413
 *   rmse          = sqrt(sum/(Nslice*Ny*Nz))           ctvlib.cpp:296-306 */
double orc_sqdiff(int64_t n, const float *a, const float *b)
{
    double acc = 0.0;
#pragma omp parallel for reduction(+ : acc)
    for (int64_t i = 0; i < n; i++) { float d = a[i] - b[i]; acc += (double)(d * d); }
    return acc;
}

/* ------------------------------------------------------------------------------------------
 * SART on the parallelRay matrix (PARITY-UNPINNED: ASTRA CCudaSartAlgorithm is absent).
 * One call = niter sweeps over the nproj angles in the given order (tomoengine.cpp:162-179:
 * run(Nproj*nIter)); per angle i:
 *     r_j = (b_j - A_j x) / (A_j 1)                     rays j of angle i (0 where A_j 1 == 0)
 *     x_p = max(0, x_p + beta * (sum_j A_jp r_j) / (sum_j A_jp))   (unchanged where denominator 0)
 * min-constraint 0 after every angle (tomoengine.cpp:157).
 * ------------------------------------------------------------------------------------------ */
void orc_sart(int nslice, int nray, int nproj, int64_t ncol, const int64_t *ptr, const int32_t *idx,
              const float *val, const float *b, float *vol, float beta, int niter, const int32_t *order)
{
    int64_t nrow = (int64_t)nray * nproj;
    float *rowsum = (float *)malloc(sizeof(float) * (size_t)nrow);
    for (int64_t r = 0; r < nrow; r++) {
        float acc = 0.0f;
        for (int64_t k = ptr[r]; k < ptr[r + 1]; k++) acc += val[k];
        rowsum[r] = acc;
    }
#pragma omp parallel
    {
        float *num = (float *)malloc(sizeof(float) * (size_t)ncol);
        float *den = (float *)malloc(sizeof(float) * (size_t)ncol);
#pragma omp for
        for (int s = 0; s < nslice; s++) {
            float *x = vol + (size_t)s * ncol;
            const float *bs = b + (size_t)s * nrow;
            for (int it = 0; it < niter; it++) {
                for (int q = 0; q < nproj; q++) {
                    int i = order ? order[q] : q;
                    for (int64_t c = 0; c < ncol; c++) { num[c] = 0.0f; den[c] = 0.0f; }
                    for (int64_t r = (int64_t)i * nray; r < (int64_t)(i + 1) * nray; r++) {
                        float dot = 0.0f;
                        for (int64_t k = ptr[r]; k < ptr[r + 1]; k++) dot += val[k] * x[idx[k]];
                        float res = rowsum[r] > 0.0f ? (bs[r] - dot) / rowsum[r] : 0.0f;
                        for (int64_t k = ptr[r]; k < ptr[r + 1]; k++) {
                            num[idx[k]] += val[k] * res;
                            den[idx[k]] += val[k];
                        }
                    }
                    for (int64_t c = 0; c < ncol; c++) {
                        float upd = den[c] > 0.0f ? num[c] / den[c] : 0.0f;
                        float v = x[c] + beta * upd;
                        x[c] = v < 0.0f ? 0.0f : v;
                    }
                }
            }
        }
        free(num); free(den);
    }
    free(rowsum);
}

/* Row/column-sum normalised SIRT (PARITY-UNPINNED: ASTRA CCudaSirtAlgorithm is absent;
 * tomoengine.cpp:181-205 with setConstraints(true,0,...)):
 *     x = max(0, x + C A^T R (b - A x)),  R = 1/(A 1), C = 1/(A^T 1), 1/0 := 0 */
void orc_sirt_norm(int nslice, int64_t nrow, int64_t ncol, const int64_t *ptr, const int32_t *idx,
                   const float *val, const float *b, float *vol, int niter)
{
    float *rowsum = (float *)malloc(sizeof(float) * (size_t)nrow);
    float *colsum = (float *)calloc((size_t)ncol, sizeof(float));
    for (int64_t r = 0; r < nrow; r++) {
        float acc = 0.0f;
        for (int64_t k = ptr[r]; k < ptr[r + 1]; k++) { acc += val[k]; colsum[idx[k]] += val[k]; }
        rowsum[r] = acc;
    }
    for (int it = 0; it < niter; it++) {
#pragma omp parallel
        {
            float *t = (float *)malloc(sizeof(float) * (size_t)nrow);
            float *u = (float *)malloc(sizeof(float) * (size_t)ncol);
#pragma omp for
            for (int s = 0; s < nslice; s++) {
                float *x = vol + (size_t)s * ncol;
                const float *bs = b + (size_t)s * nrow;
                spmv(nrow, ptr, idx, val, x, t);
                for (int64_t r = 0; r < nrow; r++) t[r] = rowsum[r] > 0.0f ? (bs[r] - t[r]) / rowsum[r] : 0.0f;
                spmv_t(nrow, ncol, ptr, idx, val, t, u);
                for (int64_t c = 0; c < ncol; c++) {
                    float upd = colsum[c] > 0.0f ? u[c] / colsum[c] : 0.0f;
                    float v = x[c] + upd;
                    x[c] = v < 0.0f ? 0.0f : v;
                }
            }
            free(t); free(u);
        }
    }
    free(rowsum); free(colsum);
}

/* ------------------------------------------------------------------------------------------
 * 3-D TV (volume [nx][ny][nz], periodic in all three dims)
 * ------------------------------------------------------------------------------------------ */
#define V(i, j, k) vol[((size_t)(i) * ny + (j)) * nz + (k)]

/* tv_3D(): ctvlib.cpp:336-367 (eps 1e-8 there; 1e-6 in tv_gd.cu:29) */
double orc_tv(int nx, int ny, int nz, const float *vol, float eps)
{
    double tv = 0.0;
#pragma omp parallel for reduction(+ : tv)
    for (int i = 0; i < nx; i++) {
        int ip = (i + 1) % nx;
        for (int j = 0; j < ny; j++) {
            int jp = (j + 1) % ny;
            for (int k = 0; k < nz; k++) {
                int kp = (k + 1) % nz;
                float c = V(i, j, k);
                float d1 = c - V(ip, j, k), d2 = c - V(i, jp, k), d3 = c - V(i, j, kp);
                tv += (double)sqrtf(eps + d1 * d1 + d2 * d2 + d3 * d3);
            }
        }
    }
    return tv;
}

/* TV gradient tensor of ctvlib.cpp:431-447 into g; returns sum g^2 */
static double tv_grad(int nx, int ny, int nz, const float *vol, float *g, float eps)
{
    double nrm = 0.0;
#pragma omp parallel for reduction(+ : nrm)
    for (int i = 0; i < nx; i++) {
        int ip = (i + 1) % nx, im = (i - 1 + nx) % nx;
        for (int j = 0; j < ny; j++) {
            int jp = (j + 1) % ny, jm = (j - 1 + ny) % ny;
            for (int k = 0; k < nz; k++) {
                int kp = (k + 1) % nz, km = (k - 1 + nz) % nz;
                float c = V(i, j, k);
                /* ctvlib.cpp:431 writes 3.0*recon[i](j,k) - ...: the literal is a double, so the whole numerator is
                 * evaluated in double and rounded to float ONCE (round 3 had 3.0f * c: three float roundings) */
                float v1n = (float)(3.0 * (double)c - (double)V(ip, j, k) - (double)V(i, jp, k) - (double)V(i, j, kp));
                float v1d = sqrtf(eps + (c - V(ip, j, k)) * (c - V(ip, j, k))
                                      + (c - V(i, jp, k)) * (c - V(i, jp, k))
                                      + (c - V(i, j, kp)) * (c - V(i, j, kp)));
                float a = V(im, j, k);
                float v2n = c - a;
                float v2d = sqrtf(eps + (a - c) * (a - c)
                                      + (a - V(im, jp, k)) * (a - V(im, jp, k))
                                      + (a - V(im, j, kp)) * (a - V(im, j, kp)));
                float bb = V(i, jm, k);
                float v3n = c - bb;
                float v3d = sqrtf(eps + (bb - V(ip, jm, k)) * (bb - V(ip, jm, k))
                                      + (bb - c) * (bb - c)
                                      + (bb - V(i, jm, kp)) * (bb - V(i, jm, kp)));
                float d = V(i, j, km);
                float v4n = c - d;
                float v4d = sqrtf(eps + (d - V(ip, j, km)) * (d - V(ip, j, km))
                                      + (d - V(i, jp, km)) * (d - V(i, jp, km))
                                      + (d - c) * (d - c));
                float gv = v1n / v1d + v2n / v2d + v3n / v3d + v4n / v4d;
                g[((size_t)i * ny + j) * nz + k] = gv;
                nrm += (double)(gv * gv);
            }
        }
    }
    return nrm;
}

/* tv_gd_3D(ng, dPOCS): ctvlib.cpp:406-462.  Returns the TV value BEFORE descent
 * (what cuda_tv_gd_3D returns: tv_gd.cu:177-183,217). */
double orc_tv_gd(int nx, int ny, int nz, float *vol, float *scratch, int ng, float dPOCS, float eps)
{
    double tv0 = orc_tv(nx, ny, nz, vol, eps);
    size_t n = (size_t)nx * ny * nz;
    for (int it = 0; it < ng; it++) {
        float tv_norm = (float)sqrt(tv_grad(nx, ny, nz, vol, scratch, eps));
#pragma omp parallel for
        for (int64_t i = 0; i < (int64_t)n; i++) vol[i] -= dPOCS * scratch[i] / tv_norm;
    }
    orc_positivity((int64_t)n, vol);
    return tv0;
}
#undef V

/* tv_gd_3D (ctvlib.cpp:406-462) evaluated in binary64 from a float32 start: NOT the reference's arithmetic, but the
 * yardstick for it.  Ten fixed-length steps along g/|g| are ill-conditioned (one ulp on the start moves the fp32 result by
 * ~5e-5 at 512^3), so two fp32 implementations cannot be held to 1e-5 of each other there; what CAN be asked is that the
 * HIP path is no further from this exact-arithmetic trajectory than the reference's own fp32 evaluation (orc_tv_gd) is.
 * work: 2 volumes of doubles.  out: the result rounded to float32 once, after the clamp. */
void orc_tv_gd_f64(int nx, int ny, int nz, const float *vol_in, double *work, float *out, int ng, double dPOCS, double eps)
{
    size_t n = (size_t)nx * ny * nz;
    double *vol = work, *g = work + n;
#pragma omp parallel for
    for (int64_t i = 0; i < (int64_t)n; i++) vol[i] = (double)vol_in[i];
#define W(i, j, k) vol[((size_t)(i) * ny + (j)) * nz + (k)]
#define DD(i, j, k, ip, jp, kp) sqrt(eps + (W(i, j, k) - W(ip, j, k)) * (W(i, j, k) - W(ip, j, k)) \
                                         + (W(i, j, k) - W(i, jp, k)) * (W(i, j, k) - W(i, jp, k)) \
                                         + (W(i, j, k) - W(i, j, kp)) * (W(i, j, k) - W(i, j, kp)))
    for (int it = 0; it < ng; it++) {
        double nrm = 0.0;
#pragma omp parallel for reduction(+ : nrm)
        for (int i = 0; i < nx; i++) {
            int ip = (i + 1) % nx, im = (i - 1 + nx) % nx;
            for (int j = 0; j < ny; j++) {
                int jp = (j + 1) % ny, jm = (j - 1 + ny) % ny;
                for (int k = 0; k < nz; k++) {
                    int kp = (k + 1) % nz, km = (k - 1 + nz) % nz;
                    double c = W(i, j, k);
                    double gv = (3.0 * c - W(ip, j, k) - W(i, jp, k) - W(i, j, kp)) / DD(i, j, k, ip, jp, kp)
                              + (c - W(im, j, k)) / DD(im, j, k, i, jp, kp)
                              + (c - W(i, jm, k)) / DD(i, jm, k, ip, j, kp)
                              + (c - W(i, j, km)) / DD(i, j, km, ip, jp, k);
                    g[((size_t)i * ny + j) * nz + k] = gv;
                    nrm += gv * gv;
                }
            }
        }
        double inv = dPOCS / sqrt(nrm);
#pragma omp parallel for
        for (int64_t i = 0; i < (int64_t)n; i++) vol[i] -= inv * g[i];
    }
#undef DD
#undef W
#pragma omp parallel for
    for (int64_t i = 0; i < (int64_t)n; i++) out[i] = (float)(vol[i] < 0.0 ? 0.0 : vol[i]);
}

/* ------------------------------------------------------------------------------------------
 * FGP-TV (tv_fgp.cu:192-281, methodTV = 0 isotropic, nonneg = 1).  Volume [N][M][Z].
 * Returns TV(input) with eps 1e-6 (tv_fgp.cu:170-189,231-238).
 * work: 4 volumes (D, P1, P2, P3).
 * ------------------------------------------------------------------------------------------ */
double orc_tv_fgp(int N, int M, int Z, float *vol, float *work, int iter, float lambda)
{
    size_t n = (size_t)N * M * Z;
    float *D = work, *P1 = work + n, *P2 = work + 2 * n, *P3 = work + 3 * n;
    memset(work, 0, sizeof(float) * 4 * n);
    double tv = orc_tv(N, M, Z, vol, 1e-6f);
    float multip = 1.0f / (26.0f * lambda);                         /* tv_fgp.cu:241 */
#define IX(i, j, k) (((size_t)(i) * M + (j)) * Z + (k))
    for (int it = 0; it < iter; it++) {
        /* Obj_func3D_kernel :44-65 + nonneg3D_kernel :143-154 */
#pragma omp parallel for
        for (int i = 0; i < N; i++)
            for (int j = 0; j < M; j++)
                for (int k = 0; k < Z; k++) {
                    size_t q = IX(i, j, k);
                    float v1 = i <= 0 ? 0.0f : P1[IX(i - 1, j, k)];
                    float v2 = j <= 0 ? 0.0f : P2[IX(i, j - 1, k)];
                    float v3 = k <= 0 ? 0.0f : P3[IX(i, j, k - 1)];
                    float d = vol[q] - lambda * (P1[q] + P2[q] + P3[q] - v1 - v2 - v3);
                    D[q] = d < 0.0f ? 0.0f : d;
                }
        /* Grad_func3D_kernel :67-91 + Proj_func3D_iso_kernel :93-115 */
#pragma omp parallel for
        for (int i = 0; i < N; i++)
            for (int j = 0; j < M; j++)
                for (int k = 0; k < Z; k++) {
                    size_t q = IX(i, j, k);
                    float v1 = i >= N - 1 ? 0.0f : D[q] - D[IX(i + 1, j, k)];
                    float v2 = j >= M - 1 ? 0.0f : D[q] - D[IX(i, j + 1, k)];
                    float v3 = k >= Z - 1 ? 0.0f : D[q] - D[IX(i, j, k + 1)];
                    float p1 = P1[q] + multip * v1, p2 = P2[q] + multip * v2, p3 = P3[q] + multip * v3;
                    float denom = p1 * p1 + p2 * p2 + p3 * p3;
                    if (denom > 1.0f) {
                        float sq = 1.0f / sqrtf(denom);
                        p1 *= sq; p2 *= sq; p3 *= sq;
                    }
                    P1[q] = p1; P2[q] = p2; P3[q] = p3;
                }
    }
#undef IX
    if (iter > 0) memcpy(vol, D, sizeof(float) * n);                 /* :272 copies d_update back */
    else memset(vol, 0, sizeof(float) * n);                          /* d_update is the memset-0 buffer */
    return tv;
}

/* fista_nesterov_momentum(beta): tomoengine.cpp:381-384 + matrix_ops.cu:77-86
 *   recon <- yk ; yk <- recon + beta*(recon - recon_old) ; recon_old <- recon */
void orc_fista_momentum(int64_t n, float *recon, float *yk, float *recon_old, float beta)
{
#pragma omp parallel for
    for (int64_t i = 0; i < n; i++) {
        float r = yk[i];
        recon[i] = r;
        yk[i] = r + beta * (r - recon_old[i]);
        recon_old[i] = r;
    }
}

/* poisson_ML(lambda): tomoengine.cpp:293-315, on the parallelRay matrix (FP/BP themselves are
 * ASTRA in the reference).  L = max(A^T A 1) (tomoengine.cpp:241-242).  Returns the cost. */
double orc_poisson_ml(int nslice, int64_t nrow, int64_t ncol, const int64_t *ptr, const int32_t *idx,
                      const float *val, const float *b, float *vol, float lambda, float L)
{
    const float eps = 1e-1f;
    double cost = 0.0;
#pragma omp parallel
    {
        float *ax = (float *)malloc(sizeof(float) * (size_t)nrow);
        float *t = (float *)malloc(sizeof(float) * (size_t)nrow);
        float *u = (float *)malloc(sizeof(float) * (size_t)ncol);
#pragma omp for reduction(+ : cost)
        for (int s = 0; s < nslice; s++) {
            float *x = vol + (size_t)s * ncol;
            const float *bs = b + (size_t)s * nrow;
            spmv(nrow, ptr, idx, val, x, ax);
            for (int64_t r = 0; r < nrow; r++) t[r] = (ax[r] - bs[r]) / (ax[r] + eps);
            spmv_t(nrow, ncol, ptr, idx, val, t, u);
            for (int64_t c = 0; c < ncol; c++) x[c] -= (lambda / L) * u[c];
            for (int64_t r = 0; r < nrow; r++) cost += (double)(ax[r] - bs[r] * logf(ax[r] + eps));
        }
        free(ax); free(t); free(u);
    }
    orc_positivity((int64_t)nslice * ncol, vol);
    return cost;
}

/* The checker runs on whatever host the tests land on: the caller bounds the team (usable CPUs, cgroup quota). */
void orc_set_num_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

int orc_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
