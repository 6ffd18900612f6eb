// kernels_misc.hip.h -- element-wise passes and reductions, multimodal steps, per-slice scalars, the WBP filter
// Part of kernels.hip.h (include that, not this: the families share helpers and constants in the order kernels.hip.h lists them).
#pragma once

namespace tomo {

// ---- element-wise and reductions (float4 grid-stride; n4 = element count / 4) -------------------------
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_clamp(f4 *__restrict__ x, int64_t n4)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        f4 v = x[i];
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        x[i] = v;
    }
}

__device__ __forceinline__ float soft1(float v, float l)
{   // matrix_ops.cu:64-75: signbit(l - |v|) * copysign(|v| - l, v)
    float a = fabsf(v);
    return a > l ? copysignf(a - l, v) : 0.f;
}

__global__ __launch_bounds__(256) void k_soft_threshold(f4 *__restrict__ x, float l, int64_t n4)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        f4 v = x[i];
        v.x = soft1(v.x, l); v.y = soft1(v.y, l); v.z = soft1(v.z, l); v.w = soft1(v.w, l);
        x[i] = v;
    }
}

// Nesterov step (tomoengine.cpp:381-384: recon <- yk ; yk <- recon + beta (recon - recon_old) ; recon_old <- recon).  The two
// copies are not stores here: the engine rotates the recon / yk buffers and keeps "recon_old == recon" as a flag, so this pass
// reads r (the prox result) and old and writes the extrapolated point; out may be the buffer old lives in (same index: read
// before write in one thread).
__global__ __launch_bounds__(256) void k_momentum(const f4 *r_in, const f4 *old, f4 *out, float beta, int64_t n4)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        typedef VecOf<4>::T V;
        V r = nt_ld<512>(reinterpret_cast<const V *>(r_in) + i), o = nt_ld<512>(reinterpret_cast<const V *>(old) + i);
        nt_st<512>(r + beta * (r - o), reinterpret_cast<V *>(out) + i);
    }
}

__global__ __launch_bounds__(256) void k_sqdiff(const f4 *__restrict__ a, const f4 *__restrict__ b,
                                                 double *__restrict__ part, int64_t n4)
{
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        f4 d = a[i] - b[i];
        acc += (double)(d.x * d.x) + (double)(d.y * d.y) + (double)(d.z * d.z) + (double)(d.w * d.w);
    }
    block_accumulate(acc, part);
}

__global__ __launch_bounds__(256) void k_l1(const f4 *__restrict__ a, double *__restrict__ part, int64_t n4)
{
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        f4 v = a[i];
        acc += (double)fabsf(v.x) + (double)fabsf(v.y) + (double)fabsf(v.z) + (double)fabsf(v.w);
    }
    block_accumulate(acc, part);
}

__global__ __launch_bounds__(256) void k_scale(f4 *__restrict__ x, float f, int64_t n4)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) x[i] = x[i] * f;
}

// max over rays and slices of one projection (block p handles projection p)
__global__ __launch_bounds__(256) void k_proj_max(const float *__restrict__ g, float *__restrict__ out, int n, int nx, int sx)
{
    __shared__ float red[256];
    const float *base = g + (size_t)blockIdx.x * n * sx;
    float m = -3.402823466e38f;
    for (int64_t i = threadIdx.x; i < (int64_t)n * sx; i += 256) {
        int s = (int)(i % sx);
        if (s < nx) m = fmaxf(m, base[i]);
    }
    red[threadIdx.x] = m;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + o]);
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = red[0];
}

// b <- (b / div[p]) * mul[p] for projection p (the two steps of multimodal.cpp:325-326)
__global__ __launch_bounds__(256) void k_proj_scale(float *__restrict__ g, const float *__restrict__ f, int n, int sx)
{
    float *base = g + (size_t)blockIdx.x * n * sx;
    float d = f[blockIdx.x], m = f[gridDim.x + blockIdx.x];
    for (int64_t i = threadIdx.x; i < (int64_t)n * sx; i += 256) base[i] = (base[i] / d) * m;
}

// ---- multimodal (ChemicalTomo) element-wise steps -----------------------------------------------------------
// Sigma of fusion_helper.py:5-32 has one weight per element and pixel-diagonal structure, so
// Sigma*x = sum_e w_e x_e and Sigma^T v = (w_e v)_e: no sparse matrix is needed.
constexpr int MM_MAX_EL = 8;
struct MMArgs { float *x[MM_MAX_EL]; float *u[MM_MAX_EL]; float w[MM_MAX_EL]; int nel; float gamma; };

// x^g for x >= 0 (the tomograms are clamped at zero) as exp2(g log2 x): the correctly rounded powf costs ~60 vector
// instructions per element and made the two fusion kernels 4x slower than their memory traffic (1.2 ms per pass at 2 x 512^3);
// this form is good to ~2e-6 relative at |g log2 x| <= 20, 0 -> 0 for g > 0 (log2 0 = -inf, exp2 -inf = 0).
// Domain: the fast path serves x > 0 (tomograms are clamped after every update); x == 0 and x < 0 (a caller-supplied start
// volume with negative voxels, an integer gamma) take powf's value exactly as numpy's ** / std::pow in the reference would
// (exp2(g log2 x) alone returned NaN there and for 0^0).  The slow branch is taken per lane only where needed.
__device__ __forceinline__ float pow_pos(float x, float g)
{
    if (__builtin_expect(x > 0.f, 1)) return exp2f(g * log2f(x));
    return x == 0.f ? (g == 0.f ? 1.f : (g > 0.f ? 0.f : INFINITY)) : powf(x, g);
}
__device__ __forceinline__ f4 pow4(f4 v, float g)
{
    f4 r; r.x = pow_pos(v.x, g); r.y = pow_pos(v.y, g); r.z = pow_pos(v.z, g); r.w = pow_pos(v.w, g); return r;
}

__global__ __launch_bounds__(256) void k_mm_model(MMArgs a, f4 *__restrict__ model, int64_t n4)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        f4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int e = 0; e < a.nel; ++e) {
            f4 v = reinterpret_cast<const f4 *>(a.x[e])[i];
            if (a.gamma != 1.0f) v = pow4(v, a.gamma);
            acc += a.w[e] * v;
        }
        model[i] = acc;
    }
}

__global__ __launch_bounds__(256) void k_mm_update(MMArgs a, const f4 *__restrict__ upd, const f4 *__restrict__ model,
                                                    float lamC_over_L, float lamH, int64_t n4)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        f4 d = {0.f, 0.f, 0.f, 0.f};
        if (lamH != 0.f) d = upd[i] - model[i];
        for (int e = 0; e < a.nel; ++e) {
            f4 x = reinterpret_cast<f4 *>(a.x[e])[i];
            f4 uc = reinterpret_cast<const f4 *>(a.u[e])[i];
            f4 uh = a.w[e] * d;                                   // Sigma^T (updateVol - modelHAADF)
            if (a.gamma != 1.0f) uh = (a.gamma * pow4(x, a.gamma - 1.0f)) * uh;
            f4 v = x - (lamC_over_L * uc - lamH * uh);
            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
            reinterpret_cast<f4 *>(a.x[e])[i] = v;
        }
    }
}

// ---- per-slice scalars (CGLS: every slice is its own least-squares problem with its own alpha, beta) -------
// sums[s] += sum_m v[m][s]^2 over the rows m of this workgroup; a thread owns 4 consecutive slices (one float4 per row), so a
// wave reads 1 KiB contiguous per row.  (Round 1's scalar form with a 64-bit modulo per element made a CGLS iteration
// spend twice as long in these helpers as in the projectors.)
// Round 3: two passes without atomics -- a workgroup leaves ITS rows' sums in part[blockIdx.y][slice] and k_slice_sumsq_finish adds
// the workgroups' sums in ascending order.  (4096 workgroups x 512 double atomics onto the same 512 addresses was most of the
// kernel's 300-416 us for a 537 MB volume, and arrival order made the per-slice alpha / beta differ in the last bits between runs.)
// All 256 threads load: the two halves of a workgroup take alternate groups of 8 rows and meet in LDS.
__global__ __launch_bounds__(256) void k_slice_sumsq(const float *__restrict__ v, double *__restrict__ part, int64_t m,
                                                      int sx, int rows_per_block)
{
    __shared__ double sh[128 * 4];
    const int cols = sx / 4;                                          // float4 columns of a row
    const int per = cols >= 256 ? 256 : (cols >= 128 ? 128 : 64);     // threads side by side on one row
    const int half = threadIdx.x / per, nhalf = 256 / per;            // row phases of this workgroup (1, 2 or 4)
    const int s4 = blockIdx.x * per + (threadIdx.x % per);            // float4 column: slices 4*s4 .. 4*s4+3
    const bool live = s4 < cols;
    const int64_t m0 = (int64_t)blockIdx.y * rows_per_block, m1 = min(m, m0 + rows_per_block);
    const f4 *p = reinterpret_cast<const f4 *>(v) + (live ? s4 : 0);
    const int64_t pitch4 = cols;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    for (int64_t r = m0 + 8 * half; live && r < m1; r += 8 * nhalf) {  // 8 independent loads per trip, rows in ascending order
        f4 a[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) a[u] = (r + u < m1) ? p[(r + u) * pitch4] : f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            a0 += (double)(a[u].x * a[u].x); a1 += (double)(a[u].y * a[u].y); a2 += (double)(a[u].z * a[u].z); a3 += (double)(a[u].w * a[u].w);
        }
    }
    // the row phases of a column meet in LDS, phase 0 adds them in ascending phase order
    for (int h = 1; h < nhalf; ++h) {
        if (half == h && per <= 128) { double *q = sh + (threadIdx.x % per) * 4; q[0] = a0; q[1] = a1; q[2] = a2; q[3] = a3; }
        __syncthreads();
        if (half == 0 && per <= 128) { const double *q = sh + (threadIdx.x % per) * 4; a0 += q[0]; a1 += q[1]; a2 += q[2]; a3 += q[3]; }
        __syncthreads();
    }
    if (half == 0 && live) {
        double *o = part + (size_t)blockIdx.y * sx + 4 * (size_t)s4;
        o[0] = a0; o[1] = a1; o[2] = a2; o[3] = a3;
    }
}

// sums[s] = sum over the nb workgroups' partial sums, ascending
__global__ void k_slice_sumsq_finish(const double *__restrict__ part, double *__restrict__ sums, int nb, int sx)
{
    int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= sx) return;
    double a = 0.0;
    for (int b = 0; b < nb; ++b) a += part[(size_t)b * sx + s];
    sums[s] = a;
}

// coef[s] = num[s] / den[s] (0 when den == 0)
__global__ void k_slice_ratio(const double *__restrict__ num, const double *__restrict__ den, float *__restrict__ coef, int sx)
{
    int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < sx) coef[s] = den[s] > 0.0 ? (float)(num[s] / den[s]) : 0.f;
}

// y[m][s] = y[m][s] + sign * coef[s] * x[m][s]: float4 grid-stride over n4 = n/4 elements, sx4 = sx/4 float4 per row.
// The grid stride is a multiple of sx4 (the launcher rounds it), so a thread's slice group -- and its 4 coefficients -- never change.
__global__ __launch_bounds__(256) void k_slice_axpy(f4 *__restrict__ y, const f4 *__restrict__ x,
                                                     const f4 *__restrict__ coef, float sign, int64_t n4, int sx4)
{
    const int64_t i0 = (int64_t)blockIdx.x * 256 + threadIdx.x, stride = (int64_t)gridDim.x * 256;
    const f4 c = sign * coef[i0 % sx4];
    for (int64_t i = i0; i < n4; i += stride) y[i] = y[i] + c * x[i];
}

// p[m][s] = z[m][s] + coef[s] * p[m][s]
__global__ __launch_bounds__(256) void k_slice_xpay(f4 *__restrict__ p, const f4 *__restrict__ z,
                                                     const f4 *__restrict__ coef, int64_t n4, int sx4)
{
    const int64_t i0 = (int64_t)blockIdx.x * 256 + threadIdx.x, stride = (int64_t)gridDim.x * 256;
    const f4 c = coef[i0 % sx4];
    for (int64_t i = i0; i < n4; i += stride) p[i] = z[i] + c * p[i];
}

// filtered sinogram for WBP: out[i*N + j][s] = sum_k h[|j - k|] in[i*N + k][s]; one wave = one output ray x 64*VEC slices
template <int VEC>
__global__ __launch_bounds__(256) void k_filter_rows(const float *__restrict__ in, float *__restrict__ out,
                                                      const float *__restrict__ h, int n, int nrows, int sx, int nchunk)
{
    typedef typename VecOf<VEC>::T V;
    int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    int64_t gw = (int64_t)blockIdx.x * 4 + wave;
    int chunk = (int)(gw / nrows);
    if (chunk >= nchunk) return;
    int row = (int)(gw - (int64_t)chunk * nrows);
    int i = row / n, j = row - i * n;
    int off = chunk * (64 * VEC) + lane * VEC;
    const float *base = in + (size_t)i * n * sx + off;
    V acc = vzero<VEC>();
#pragma unroll 8
    for (int k = 0; k < n; ++k) {
        int d = j - k;
        acc += h[d < 0 ? -d : d] * *reinterpret_cast<const V *>(base + (size_t)k * sx);
    }
    *reinterpret_cast<V *>(out + (size_t)row * sx + off) = acc;
}

}  // namespace tomo
