// kernels.hip.h -- gfx950 device kernels of the tomography hot path.
//
// Device layout ("slab-interleaved"): every field is stored with the slice index
// fastest,   volume  x[pixel = y*N + z][s]   and   sinogram  g[row = angle*N + ray][s],   row pitch sx floats
// (sx = Nslice rounded up to 64, padding slices are identically zero).  All slices share one system
// matrix, so a matrix entry (row, pixel, w) is wave-uniform scalar data and its use is one coalesced
// vector AXPY across slices: lanes index slices, the scalar unit walks the ray / voxel tables.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tomo {

// ---- small vector helpers -----------------------------------------------------------------------
template <int V> struct VecOf;
template <> struct VecOf<1> { typedef float T; };
template <> struct VecOf<2> { typedef float T __attribute__((ext_vector_type(2))); };
template <> struct VecOf<4> { typedef float T __attribute__((ext_vector_type(4))); };

template <int V> __device__ __forceinline__ typename VecOf<V>::T vzero();
template <> __device__ __forceinline__ float vzero<1>() { return 0.f; }
template <> __device__ __forceinline__ VecOf<2>::T vzero<2>() { VecOf<2>::T v = {0.f, 0.f}; return v; }
template <> __device__ __forceinline__ VecOf<4>::T vzero<4>() { VecOf<4>::T v = {0.f, 0.f, 0.f, 0.f}; return v; }

template <int V> __device__ __forceinline__ float velem(const typename VecOf<V>::T &v, int i);
template <> __device__ __forceinline__ float velem<1>(const float &v, int) { return v; }
template <> __device__ __forceinline__ float velem<2>(const VecOf<2>::T &v, int i) { return v[i]; }
template <> __device__ __forceinline__ float velem<4>(const VecOf<4>::T &v, int i) { return v[i]; }

template <int V> __device__ __forceinline__ void vset(typename VecOf<V>::T &v, int i, float f);
template <> __device__ __forceinline__ void vset<1>(float &v, int, float f) { v = f; }
template <> __device__ __forceinline__ void vset<2>(VecOf<2>::T &v, int i, float f) { v[i] = f; }
template <> __device__ __forceinline__ void vset<4>(VecOf<4>::T &v, int i, float f) { v[i] = f; }

// ---- block reduction of a double into one of NPART partial slots ----------------------------------
constexpr int NPART = 256;

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// all threads of a 256-thread block call this; one atomic per block
__device__ __forceinline__ void block_accumulate(double v, double *__restrict__ part)
{
    __shared__ double red[4];
    v = wave_sum(v);
    int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(&part[blockIdx.x & (NPART - 1)], red[0] + red[1] + red[2] + red[3]);
}

__global__ void k_finalize(const double *__restrict__ part, double *__restrict__ dst)
{
    double v = part[threadIdx.x];  // launched with NPART threads
    __shared__ double red[NPART / 64];
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0;
        for (int i = 0; i < NPART / 64; ++i) s += red[i];
        *dst = s;
    }
}

// ---- layout conversion at the host boundary -------------------------------------------------------
// host [ns][m]  ->  device [m][sx]   (padding slices written as zero)
__global__ __launch_bounds__(256) void k_transpose_in(const float *__restrict__ src, float *__restrict__ dst,
                                                       int ns, int64_t m, int sx)
{
    __shared__ float tile[32][33];
    int64_t m0 = (int64_t)blockIdx.x * 32;
    int s0 = blockIdx.y * 32;
    int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        int s = s0 + r;
        int64_t mm = m0 + tx;
        tile[r][tx] = (s < ns && mm < m) ? src[(int64_t)s * m + mm] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        int64_t mm = m0 + r;
        int s = s0 + tx;
        if (mm < m && s < sx) dst[mm * sx + s] = tile[tx][r];
    }
}

// device [m][sx] -> host [ns][m]
__global__ __launch_bounds__(256) void k_transpose_out(const float *__restrict__ src, float *__restrict__ dst,
                                                        int ns, int64_t m, int sx)
{
    __shared__ float tile[32][33];
    int64_t m0 = (int64_t)blockIdx.x * 32;
    int s0 = blockIdx.y * 32;
    int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        int64_t mm = m0 + r;
        int s = s0 + tx;
        tile[r][tx] = (mm < m && s < sx) ? src[mm * sx + s] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        int s = s0 + r;
        int64_t mm = m0 + tx;
        if (s < ns && mm < m) dst[(int64_t)s * m + mm] = tile[tx][r];
    }
}

__global__ void k_scatter_slice(const float *__restrict__ img, float *__restrict__ vol, int64_t m, int sx, int s)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) vol[i * sx + s] = img[i];
}

__global__ void k_gather_slice(const float *__restrict__ vol, float *__restrict__ img, int64_t m, int sx, int s)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) img[i] = vol[i * sx + s];
}

// ---- forward projector: ray-driven, one workgroup per (ray, slice chunk) ---------------------------
// g[row][s] = sum_k w_k * x[col_k][s].  The four waves of a workgroup split the ray's entry list; lanes hold
// VEC consecutive slices each.  Entry (col, w) pairs are wave-uniform: fetched by the scalar unit.
enum { FP_STORE = 0, FP_RESID = 1, FP_RESID_NORM = 2, FP_DD = 3, FP_POISSON = 4 };

template <int VEC, int MODE>
__global__ __launch_bounds__(256) void k_fp_rows(const float *__restrict__ x, const uint32_t *__restrict__ rptr,
                                                  const uint2 *__restrict__ rent, const float *__restrict__ b,
                                                  const float *__restrict__ rowsum, float *__restrict__ out,
                                                  double *__restrict__ part, int row0, int nrows, int sx)
{
    typedef typename VecOf<VEC>::T V;
    // XCD-aware order: consecutive rays (which share pixels) land on the same XCD's L2
    int nb = gridDim.x, bid = blockIdx.x;
    int v = (nb & 7) == 0 ? (bid & 7) * (nb >> 3) + (bid >> 3) : bid;
    int chunk = v / nrows;
    int row = row0 + (v - chunk * nrows);
    int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int lane = threadIdx.x & 63;
    uint32_t beg = rptr[row], end = rptr[row + 1];
    uint32_t seg = (end - beg + 3u) >> 2;
    uint32_t kb = min(beg + wave * seg, end), ke = min(kb + seg, end);
    int off = chunk * (64 * VEC) + lane * VEC;
    const float *xp = x + off;
    V acc = vzero<VEC>();
#pragma unroll 8
    for (uint32_t k = kb; k < ke; ++k) {
        uint2 e = rent[k];
        float w = __uint_as_float(e.y);
        V xv = *reinterpret_cast<const V *>(xp + (size_t)e.x * sx);
        acc += w * xv;
    }
    __shared__ V red[3][64];
    if (wave > 0) red[wave - 1][lane] = acc;
    __syncthreads();
    double local = 0.0;
    if (wave == 0) {
        acc = ((acc + red[0][lane]) + red[1][lane]) + red[2][lane];
        size_t o = (size_t)row * sx + off;
        if (MODE == FP_STORE) {
            *reinterpret_cast<V *>(out + o) = acc;
        } else {
            V bv = *reinterpret_cast<const V *>(b + o);
            V r;
            if (MODE == FP_RESID) {
                r = bv - acc;
            } else if (MODE == FP_RESID_NORM) {
                float rs = rowsum[row];
                r = rs > 0.f ? (bv - acc) / rs : vzero<VEC>();
            } else if (MODE == FP_DD) {
                r = acc;
#pragma unroll
                for (int i = 0; i < VEC; ++i) {
                    float d = velem<VEC>(acc, i) - velem<VEC>(bv, i);
                    local += (double)(d * d);
                }
            } else {  // FP_POISSON: tomoengine.cpp:302,311
                const float eps = 1e-1f;
#pragma unroll
                for (int i = 0; i < VEC; ++i) {
                    float a = velem<VEC>(acc, i), bb = velem<VEC>(bv, i);
                    vset<VEC>(r, i, (a - bb) / (a + eps));
                    local += (double)(a - bb * logf(a + eps));
                }
            }
            *reinterpret_cast<V *>(out + o) = r;
        }
    }
    if (MODE == FP_DD || MODE == FP_POISSON) {
        if (wave == 0) {
            local = wave_sum(local);
            if (lane == 0) atomicAdd(&part[blockIdx.x & (NPART - 1)], local);
        }
    }
}

// ---- voxel-driven back-projector, one angle (the SART update) ---------------------------------------
// x[p][s] = max(0, x[p][s] + beta * (w0 r[j0][s] + w1 r[j1][s]) / (w0 + w1))
// cell[p] = {j0, w0, j1, w1}: the (at most two) rays of this angle through pixel p.  r = this angle's
// normalised residual rows (N rows, L2 resident).  One wave owns PPW consecutive pixels of a slice chunk.
struct CellD { uint32_t r0; float w0; uint32_t r1; float w1; };

template <int VEC, int PPW>
__global__ __launch_bounds__(256) void k_bp_angle(float *__restrict__ x, const CellD *__restrict__ cell,
                                                   const float *__restrict__ r, float beta, int npix, int sx,
                                                   int ngroups)
{
    typedef typename VecOf<VEC>::T V;
    int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int lane = threadIdx.x & 63;
    int gw = blockIdx.x * 4 + wave;  // global wave id
    int chunk = gw / ngroups;
    int grp = gw - chunk * ngroups;
    int p0 = grp * PPW;
    if (p0 >= npix) return;
    int off = chunk * (64 * VEC) + lane * VEC;
    V xv[PPW], r0[PPW], r1[PPW];
    CellD c[PPW];
#pragma unroll
    for (int q = 0; q < PPW; ++q) {
        int p = min(p0 + q, npix - 1);
        c[q] = cell[p];
        xv[q] = *reinterpret_cast<const V *>(x + (size_t)p * sx + off);
        r0[q] = *reinterpret_cast<const V *>(r + (size_t)c[q].r0 * sx + off);
        r1[q] = *reinterpret_cast<const V *>(r + (size_t)c[q].r1 * sx + off);
    }
#pragma unroll
    for (int q = 0; q < PPW; ++q) {
        int p = p0 + q;
        if (p < npix) {
            float cs = c[q].w0 + c[q].w1;
            V num = c[q].w0 * r0[q];
            num += c[q].w1 * r1[q];
            V upd = cs > 0.f ? num / cs : vzero<VEC>();
            V nv = xv[q] + beta * upd;
#pragma unroll
            for (int i = 0; i < VEC; ++i) vset<VEC>(nv, i, fmaxf(velem<VEC>(nv, i), 0.f));
            *reinterpret_cast<V *>(x + (size_t)p * sx + off) = nv;
        }
    }
}

// ---- voxel-driven back-projector, all angles (SIRT / Landweber / plain A^T / Poisson) ----------------
// acc[p][s] = sum_i (w0 r[i*N+j0][s] + w1 r[i*N+j1][s])      rows in ascending order, like Eigen's A^T*v
// epilogue:  v = alpha*x + beta * (colsum ? acc/colsum[p] : acc);  x = clamp ? max(0, v) : v
template <int VEC, int PPW>
__global__ __launch_bounds__(256) void k_bp_all(float *__restrict__ x, const CellD *__restrict__ cell,
                                                 const float *__restrict__ r, const float *__restrict__ colsum,
                                                 float alpha, float beta, int clamp, int nproj, int nray, int npix,
                                                 int sx, int ngroups)
{
    typedef typename VecOf<VEC>::T V;
    int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int lane = threadIdx.x & 63;
    int gw = blockIdx.x * 4 + wave;
    int chunk = gw / ngroups;
    int grp = gw - chunk * ngroups;
    int p0 = grp * PPW;
    if (p0 >= npix) return;
    int off = chunk * (64 * VEC) + lane * VEC;
    V acc[PPW];
#pragma unroll
    for (int q = 0; q < PPW; ++q) acc[q] = vzero<VEC>();
    for (int i = 0; i < nproj; ++i) {
        const CellD *ci = cell + (size_t)i * npix;
        const float *ri = r + (size_t)i * nray * sx + off;
#pragma unroll
        for (int q = 0; q < PPW; ++q) {
            int p = min(p0 + q, npix - 1);
            CellD c = ci[p];
            V a0 = *reinterpret_cast<const V *>(ri + (size_t)c.r0 * sx);
            V a1 = *reinterpret_cast<const V *>(ri + (size_t)c.r1 * sx);
            acc[q] += c.w0 * a0;
            acc[q] += c.w1 * a1;
        }
    }
#pragma unroll
    for (int q = 0; q < PPW; ++q) {
        int p = p0 + q;
        if (p < npix) {
            V a = acc[q];
            if (colsum) {
                float cs = colsum[p];
                a = cs > 0.f ? a / cs : vzero<VEC>();
            }
            float *xp = x + (size_t)p * sx + off;
            V nv = beta * a;
            if (alpha != 0.f) nv = alpha * (*reinterpret_cast<const V *>(xp)) + nv;
            if (clamp) {
#pragma unroll
                for (int i = 0; i < VEC; ++i) vset<VEC>(nv, i, fmaxf(velem<VEC>(nv, i), 0.f));
            }
            *reinterpret_cast<V *>(xp) = nv;
        }
    }
}

// ---- ART (Kaczmarz), row-sequential by definition (ctvlib.cpp:137-155) -------------------------------
// One wave per 64 slices walks every row in order: a = (b_j - A_j x)/|A_j|^2 ; x += A_j^T a beta.
__global__ __launch_bounds__(64) void k_art(float *__restrict__ x, const uint32_t *__restrict__ rptr,
                                             const uint2 *__restrict__ rent, const float *__restrict__ b,
                                             const float *__restrict__ inner, float beta, int nrows, int sx)
{
    int off = blockIdx.x * 64 + threadIdx.x;
    float *xp = x + off;
    for (int row = 0; row < nrows; ++row) {
        float ip = inner[row];
        if (!(ip > 0.f)) continue;
        uint32_t beg = rptr[row], end = rptr[row + 1];
        float dot = 0.f;
        for (uint32_t k = beg; k < end; ++k) {
            uint2 e = rent[k];
            dot += __uint_as_float(e.y) * xp[(size_t)e.x * sx];
        }
        float a = (b[(size_t)row * sx + off] - dot) / ip;
        for (uint32_t k = beg; k < end; ++k) {
            uint2 e = rent[k];
            xp[(size_t)e.x * sx] += __uint_as_float(e.y) * a * beta;
        }
    }
}

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

// recon <- yk ; yk <- recon + beta (recon - recon_old) ; recon_old <- recon   (tomoengine.cpp:381-384)
__global__ __launch_bounds__(256) void k_momentum(f4 *__restrict__ recon, f4 *__restrict__ yk,
                                                   f4 *__restrict__ old, float beta, int64_t n4)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        f4 r = yk[i], o = old[i];
        recon[i] = r;
        yk[i] = r + beta * (r - o);
        old[i] = r;
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

// ---- 3-D TV stencils ---------------------------------------------------------------------------------
// Index map to the reference's (i, j, k): i = slice s (periodic over the GLOBAL slice count, neighbours
// of the slab's end slices come from halo planes), j = y, k = z (periodic over N).
// One wave = one pixel x 64 slices; waves stride over (pixel, chunk) items.
struct Halo { const float *lo; const float *hi; };

__device__ __forceinline__ float ldx(const float *__restrict__ x, const Halo &h, int pix, int s, int nx, int sx)
{
    if (s < 0) return h.lo[pix];
    if (s >= nx) return h.hi[pix];
    return x[(size_t)pix * sx + s];
}

__global__ void k_halo_pack(const float *__restrict__ x, float *__restrict__ dst, int npix, int sx, int s)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < npix) dst[i] = x[(size_t)i * sx + s];
}

// sum sqrt(eps + (x - x_ip)^2 + (x - x_jp)^2 + (x - x_kp)^2)     (ctvlib.cpp:336-367, tv_gd.cu:27-47)
__global__ __launch_bounds__(256) void k_tv_value(const float *__restrict__ x, Halo h, double *__restrict__ part,
                                                   float eps, int n, int nx, int sx)
{
    int lane = threadIdx.x & 63;
    int nchunk = sx >> 6;
    int64_t items = (int64_t)n * n * nchunk;
    int64_t wstride = (int64_t)gridDim.x * 4;
    double acc = 0.0;
    for (int64_t it = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); it < items; it += wstride) {
        int chunk = (int)(it / ((int64_t)n * n));
        int p = (int)(it - (int64_t)chunk * n * n);
        int y = p / n, z = p - y * n;
        int pjp = (y + 1 == n ? 0 : y + 1) * n + z;
        int pkp = y * n + (z + 1 == n ? 0 : z + 1);
        int s = chunk * 64 + lane;
        if (s < nx) {
            float c = x[(size_t)p * sx + s];
            float d1 = c - ldx(x, h, p, s + 1, nx, sx);
            float d2 = c - x[(size_t)pjp * sx + s];
            float d3 = c - x[(size_t)pkp * sx + s];
            acc += (double)sqrtf(eps + d1 * d1 + d2 * d2 + d3 * d3);
        }
    }
    block_accumulate(acc, part);
}

// TV gradient tensor g (ctvlib.cpp:431-447) + fused sum g^2
__global__ __launch_bounds__(256) void k_tv_grad(const float *__restrict__ x, Halo h, float *__restrict__ g,
                                                  double *__restrict__ part, float eps, int n, int nx, int sx)
{
    int lane = threadIdx.x & 63;
    int nchunk = sx >> 6;
    int64_t items = (int64_t)n * n * nchunk;
    int64_t wstride = (int64_t)gridDim.x * 4;
    double acc = 0.0;
    for (int64_t it = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); it < items; it += wstride) {
        int chunk = (int)(it / ((int64_t)n * n));
        int p = (int)(it - (int64_t)chunk * n * n);
        int y = p / n, z = p - y * n;
        int yp = (y + 1 == n ? 0 : y + 1), ym = (y == 0 ? n - 1 : y - 1);
        int zp = (z + 1 == n ? 0 : z + 1), zm = (z == 0 ? n - 1 : z - 1);
        int pjp = yp * n + z, pjm = ym * n + z, pkp = y * n + zp, pkm = y * n + zm;
        int pjm_kp = ym * n + zp, pjp_km = yp * n + zm;
        int s = chunk * 64 + lane;
        if (s < nx) {
            float c = x[(size_t)p * sx + s];
            float x_ip = ldx(x, h, p, s + 1, nx, sx);
            float x_jp = x[(size_t)pjp * sx + s];
            float x_kp = x[(size_t)pkp * sx + s];
            float v1n = 3.0f * c - x_ip - x_jp - x_kp;
            float v1d = sqrtf(eps + (c - x_ip) * (c - x_ip) + (c - x_jp) * (c - x_jp) + (c - x_kp) * (c - x_kp));
            float a = ldx(x, h, p, s - 1, nx, sx);
            float a_jp = ldx(x, h, pjp, s - 1, nx, sx);
            float a_kp = ldx(x, h, pkp, s - 1, nx, sx);
            float v2n = c - a;
            float v2d = sqrtf(eps + (a - c) * (a - c) + (a - a_jp) * (a - a_jp) + (a - a_kp) * (a - a_kp));
            float bb = x[(size_t)pjm * sx + s];
            float b_ip = ldx(x, h, pjm, s + 1, nx, sx);
            float b_kp = x[(size_t)pjm_kp * sx + s];
            float v3n = c - bb;
            float v3d = sqrtf(eps + (bb - b_ip) * (bb - b_ip) + (bb - c) * (bb - c) + (bb - b_kp) * (bb - b_kp));
            float d = x[(size_t)pkm * sx + s];
            float d_ip = ldx(x, h, pkm, s + 1, nx, sx);
            float d_jp = x[(size_t)pjp_km * sx + s];
            float v4n = c - d;
            float v4d = sqrtf(eps + (d - d_ip) * (d - d_ip) + (d - d_jp) * (d - d_jp) + (d - c) * (d - c));
            float gv = v1n / v1d + v2n / v2d + v3n / v3d + v4n / v4d;
            g[(size_t)p * sx + s] = gv;
            acc += (double)(gv * gv);
        }
    }
    block_accumulate(acc, part);
}

// x -= dPOCS * g / ||g||   (ctvlib.cpp:452-458); gnorm2 = global sum g^2 on the device; optional clamp (:461)
__global__ __launch_bounds__(256) void k_tv_update(f4 *__restrict__ x, const f4 *__restrict__ g,
                                                    const double *__restrict__ gnorm2, float dPOCS, int clamp,
                                                    int64_t n4)
{
    float nrm = (float)sqrt(*gnorm2);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        f4 v = x[i] - (dPOCS * g[i]) / nrm;
        if (clamp) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        x[i] = v;
    }
}

// FGP-TV (tv_fgp.cu).  Non-periodic: i-1 below the first GLOBAL slice and i+1 above the last are "0" terms.
// D = max(0, A - lambda (P1 + P2 + P3 - P1[i-1] - P2[j-1] - P3[k-1]))        (:44-65, :143-154)
__global__ __launch_bounds__(256) void k_fgp_obj(const float *__restrict__ A, float *__restrict__ D,
                                                  const float *__restrict__ P1, const float *__restrict__ P2,
                                                  const float *__restrict__ P3, const float *__restrict__ p1_lo,
                                                  int first, float lambda, int n, int nx, int sx)
{
    int lane = threadIdx.x & 63;
    int nchunk = sx >> 6;
    int64_t items = (int64_t)n * n * nchunk;
    int64_t wstride = (int64_t)gridDim.x * 4;
    for (int64_t it = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); it < items; it += wstride) {
        int chunk = (int)(it / ((int64_t)n * n));
        int p = (int)(it - (int64_t)chunk * n * n);
        int y = p / n, z = p - y * n;
        int s = chunk * 64 + lane;
        if (s < nx) {
            size_t q = (size_t)p * sx + s;
            float v1 = s > 0 ? P1[q - 1] : (first ? 0.f : p1_lo[p]);
            float v2 = y > 0 ? P2[q - (size_t)n * sx] : 0.f;
            float v3 = z > 0 ? P3[q - sx] : 0.f;
            float d = A[q] - lambda * (P1[q] + P2[q] + P3[q] - v1 - v2 - v3);
            D[q] = fmaxf(d, 0.f);
        }
    }
}

// P += (1/(26 lambda)) * forward-diff(D), then isotropic projection                (:67-115)
__global__ __launch_bounds__(256) void k_fgp_grad(const float *__restrict__ D, float *__restrict__ P1,
                                                   float *__restrict__ P2, float *__restrict__ P3,
                                                   const float *__restrict__ d_hi, int last, float multip, int n,
                                                   int nx, int sx)
{
    int lane = threadIdx.x & 63;
    int nchunk = sx >> 6;
    int64_t items = (int64_t)n * n * nchunk;
    int64_t wstride = (int64_t)gridDim.x * 4;
    for (int64_t it = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); it < items; it += wstride) {
        int chunk = (int)(it / ((int64_t)n * n));
        int p = (int)(it - (int64_t)chunk * n * n);
        int y = p / n, z = p - y * n;
        int s = chunk * 64 + lane;
        if (s < nx) {
            size_t q = (size_t)p * sx + s;
            float dc = D[q];
            float v1 = s + 1 < nx ? dc - D[q + 1] : (last ? 0.f : dc - d_hi[p]);
            float v2 = y + 1 < n ? dc - D[q + (size_t)n * sx] : 0.f;
            float v3 = z + 1 < n ? dc - D[q + sx] : 0.f;
            float a = P1[q] + multip * v1, b = P2[q] + multip * v2, c = P3[q] + multip * v3;
            float denom = a * a + b * b + c * c;
            if (denom > 1.0f) {
                float sq = 1.0f / sqrtf(denom);
                a *= sq; b *= sq; c *= sq;
            }
            P1[q] = a; P2[q] = b; P3[q] = c;
        }
    }
}

}  // namespace tomo
