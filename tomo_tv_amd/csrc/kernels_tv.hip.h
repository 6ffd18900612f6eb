// kernels_tv.hip.h -- 3-D TV: value, gradient (direct, LDS march, register marches), update
// Part of kernels.hip.h (include that, not this: the families share helpers and constants in the order kernels.hip.h lists them).
#pragma once

namespace tomo {

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

// dst = sum of n device doubles (the partial sums several slab engines on one device hold for the same quantity)
// scalar read-back without the copy engine: the device writes the slots straight into pinned host memory (a D2H hipMemcpyAsync
// of 128 bytes left a ~50 us bubble on the stream after it: rocprofv3 gap analysis, round 3)
__global__ void k_scalars_to_host(const double *__restrict__ src, double *__restrict__ host_dst, int n)
{
    int i = threadIdx.x;
    if (i < n) host_dst[i] = src[i];
}

struct SumSrc { const double *p[8]; int n; };
__global__ void k_sum_doubles(SumSrc src, double *__restrict__ dst)
{
    double s = 0.0;
    for (int i = 0; i < src.n; ++i) s += *src.p[i];
    *dst = s;
}

// periodic wrap of a single slab in one launch: lo = last slice, hi = slice 0
__global__ void k_halo_wrap(const float *__restrict__ x, float *__restrict__ lo, float *__restrict__ hi, int npix, int sx, int nx)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < npix) { lo[i] = x[(size_t)i * sx + nx - 1]; hi[i] = x[(size_t)i * sx]; }
}

// sum sqrt(eps + (x - x_ip)^2 + (x - x_jp)^2 + (x - x_kp)^2)     (ctvlib.cpp:336-367, tv_gd.cu:27-47)
__global__ __launch_bounds__(256) void k_tv_value(const float *__restrict__ x, Halo h, double *__restrict__ part,
                                                   float eps, int n, int nx, int sx)
{
    int lane = threadIdx.x & 63;
    int nchunk = (nx + 63) >> 6   /* computed width, not the row pitch: the pitch may carry padding */;
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
    int nchunk = (nx + 63) >> 6   /* computed width, not the row pitch: the pitch may carry padding */;
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
            float v1n = ((c - x_ip) + (c - x_jp)) + (c - x_kp);   // 3 c - x_ip - x_jp - x_kp without the cancellation at 2c (tv_v1n)
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

// LDS-tiled form of k_tv_grad.  The direct form re-reads every voxel from up to 7 pixel rows that lie ~N*sx
// floats apart, which the L2 cannot hold (measured: 5x the compulsory HBM traffic), and evaluates 4 square roots
// and 4 divisions per voxel (VALU-bound once the traffic is fixed).  Here a workgroup owns TZ z-columns x 64
// slices and marches along y with the pixel rows y-1 .. y+2 in a 4-slot LDS ring (one-element halo in z and s):
//  * every volume element is fetched once per workgroup column, the next row's loads fly during compute;
//  * the four denominators of ctvlib.cpp:431-447 are one field, D(p) = sqrt(eps + sum_d (x_p - x_{p+d})^2),
//    taken at p, p-i, p-j, p-k (same term order as the reference), so D is evaluated ONCE per voxel, its
//    reciprocal R = 1/D (<= 1 ulp) is shared through LDS, and the gradient is
//    g = (3c - x_ip - x_jp - x_kp) R(p) + (c - x_im) R(p-i) + (c - x_jm) R(p-j) + (c - x_km) R(p-k).
//    (v * (1/D) instead of v / D: at most one ulp per term away from the reference's expression.)
// The gradient value from its thirteen inputs, with every rounding written out (explicit fma / mul / sub): the march kernels
// are instantiated in several modes (store / norm only / recompute-and-update; LDS or register march) and the compiler's
// contraction choices differ between instantiations -- this keeps all of them bit-identical.
// A product / sum / difference that keeps ITS OWN rounding.  HIP's __fmul_rn / __fadd_rn / __fsub_rn are plain * + - (see
// __clang_hip_math.h) and device code is compiled with -ffp-contract=fast-honor-pragmas: a*b + c written with them is fused into
// one FMA wherever the instruction selector likes, differently in every kernel that inlines the expression (round 3 found the
// three march forms an ulp apart that way).  The pragma takes the `contract` flag off these instructions, inlined or not.
__device__ __forceinline__ float nc_mul(float a, float b)
{
#pragma clang fp contract(off)
    return a * b;
}
__device__ __forceinline__ float nc_add(float a, float b)
{
#pragma clang fp contract(off)
    return a + b;
}
__device__ __forceinline__ float nc_sub(float a, float b)
{
#pragma clang fp contract(off)
    return a - b;
}
// The same three on a PAIR of values: gfx950 issues v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 at the rate of their scalar
// forms (two IEEE results per lane per issue, each rounded exactly like the scalar instruction), so arithmetic written on pairs
// costs half the vector-ALU cycles and keeps every bit.
// A wave-uniform pointer pinned in scalar registers, accessed with a 32-bit per-lane BYTE offset: the "scalar base + vector
// offset" form of the global instructions, no address arithmetic per access when the offsets are loop invariants.  Left alone,
// the optimiser re-associates (row base + column offset) + lane into (row base + lane) + column offset and pays a 64-bit VECTOR
// add per access (30 of the ~270 vector instructions of a TV march row).
struct SBase { const __attribute__((address_space(1))) char *p; };
__device__ __forceinline__ SBase sgpr_base(const void *p)
{
    asm("" : "+s"(p));
    return SBase{(const __attribute__((address_space(1))) char *)p};   // (the barrier hides that p is global memory: say so)
}
// (the offset is re-pinned at every use, in place: its zero-extension to 64 bits must sit next to the access for the instruction
// selector to fold it -- hoisted out of the loop it costs a register pair per offset and a 64-bit vector add per access again)
__device__ __forceinline__ float ld_so(SBase b, unsigned &byte_off)
{
    asm("" : "+v"(byte_off));
    return *(const __attribute__((address_space(1))) float *)(b.p + byte_off);
}
template <bool NT> __device__ __forceinline__ void st_so(SBase b, unsigned &byte_off, float v)
{
    asm("" : "+v"(byte_off));
    auto q = (__attribute__((address_space(1))) float *)(b.p + byte_off);
    if (NT) __builtin_nontemporal_store(v, q);
    else *q = v;
}
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f nc_mul2(v2f a, v2f b)
{
#pragma clang fp contract(off)
    return a * b;
}
__device__ __forceinline__ v2f nc_sub2(v2f a, v2f b)
{
#pragma clang fp contract(off)
    return a - b;
}
__device__ __forceinline__ v2f nc_add2(v2f a, v2f b)
{
#pragma clang fp contract(off)
    return a + b;
}
__device__ __forceinline__ v2f fma2(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }

// The numerator of the first term, 3 c - x_ip - x_jp - x_kp (ctvlib.cpp:431).  The reference writes it with the double literal
// 3.0, so it is evaluated in binary64 and rounded once: no cancellation error.  In fp32 `fma(3, c, -x_ip) - x_jp - x_kp` rounds at
// the magnitude of 2c (an absolute error of ~1e-7 on a numerator that is a small difference of neighbouring voxels: 1e-4 ... 0.1
// relative); the sum of the three forward differences (c - x_ip) + (c - x_jp) + (c - x_kp) -- which the march has in hand, they
// are what R is made of, and which are exact wherever neighbours lie within a factor of two (Sterbenz) -- rounds at the magnitude
// of the numerator itself and costs one instruction less (round 4; TV_V1N_DIFFS 0 restores the round-3 expression for A/B runs).
#ifndef TV_V1N_DIFFS
#define TV_V1N_DIFFS 1
#endif
__device__ __forceinline__ float tv_v1n(float c, float xip, float xjp, float xkp)
{
#if TV_V1N_DIFFS
    return nc_add(nc_add(nc_sub(c, xip), nc_sub(c, xjp)), nc_sub(c, xkp));
#else
    return nc_sub(nc_sub(__fmaf_rn(3.0f, c, -xip), xjp), xkp);
#endif
}

__device__ __forceinline__ float tv_gval(float c, float xip, float xjp, float xkp, float r0, float xim, float rim,
                                          float xjm, float rjm, float xkm, float rkm)
{
    // Round 3: the four terms are four ROUNDED products added left to right -- the structure of the reference's
    // v1n/v1d + v2n/v2d + v3n/v3d + v4n/v4d (ctvlib.cpp:431-447; round 2 chained FMAs) -- which is what lets the register march
    // take the three backward terms from where they are cheapest: (c - x_im) R(p-i) is the product (x_ip - c) R formed at the
    // neighbouring slice (one lane shift of a product instead of two shifts of its factors), (c - x_jm) R(p-j) the product formed
    // one row earlier, (c - x_km) R(p-k) the one formed one column earlier.  Same operands, same roundings: bit-identical.
    float v1n = tv_v1n(c, xip, xjp, xkp);
    float gv = nc_mul(v1n, r0);
    gv = nc_add(gv, nc_mul(nc_sub(c, xim), rim));
    gv = nc_add(gv, nc_mul(nc_sub(c, xjm), rjm));
    gv = nc_add(gv, nc_mul(nc_sub(c, xkm), rkm));
    return gv;
}

// R = 1/sqrt(q) for the TV gradient: the hardware estimate v_rsq_f32 (1 ulp).  TV_RSQ_NEWTON adds one Newton step
// (y (1.5 - 0.5 q y^2), 4 more instructions per voxel = 12 % of the march's vector work) -- round 1 carried it; the estimate
// alone keeps every parity figure (the gradient is v * R with v a difference of voxels: its relative error stays ~1e-7).
// One definition for every form of the march, so they stay bit-identical.
#ifndef TV_RSQ_NEWTON
#define TV_RSQ_NEWTON 0
#endif
__device__ __forceinline__ float tv_rsqrt(float q)
{
    float y = __frsqrt_rn(q);
#if TV_RSQ_NEWTON
    float e = __fmaf_rn(-__fmul_rn(q, y), __fmul_rn(0.5f, y), 0.5f);   // 0.5 - 0.5 q y^2
    y = __fmaf_rn(y, e, y);
#endif
    return y;
}

// The descent step x - dPOCS g / ||g|| (ctvlib.cpp:452-458).  The reference evaluates (dPOCS * g) / ||g|| per voxel; here the
// step length dPOCS / ||g|| is formed ONCE per pass (one IEEE division) and the voxel update is one fused multiply-add: the
// IEEE division per voxel was ~10 of the ~40 vector instructions a voxel of the update pass costs (round 3).  At most 1.5 ulp of
// the STEP away from the reference's expression.  One definition for every form (march, stored-gradient update, halo planes),
// so they stay bit-identical to each other.
// (the length is capped at FLT_MAX: with ||g|| zero or denormal dPOCS / ||g|| overflows and -g * inf would turn a voxel whose gradient
// is zero into NaN, where the reference's (dPOCS * g) / ||g|| stays finite unless every g is zero -- and there the capped form leaves the
// volume as it is instead of the reference's 0 / 0)
__device__ __forceinline__ float tv_step_len(float dPOCS, const double *gnorm2) { return fminf(__fdiv_rn(dPOCS, (float)sqrt(*gnorm2)), 3.402823466e38f); }
__device__ __forceinline__ float tv_step(float c, float gv, float len) { return __fmaf_rn(-gv, len, c); }

constexpr int TVL_TZ = 8;          // z-columns per workgroup of the FGP kernel (2 per wave)
constexpr int TVL_PITCH = 66;      // 64 slices + halo each side

__device__ __forceinline__ float tv_ld(const float *__restrict__ x, const Halo &h, int pix, int s, int nx, int sx)
{
    // one load through a selected address (three guarded loads compile to a branch ladder per element)
    const float *p = x + (size_t)pix * sx + s;
    p = (s < 0) ? h.lo + pix : p;
    p = (s >= nx) ? h.hi + pix : p;
    return *p;
}

// WITH_TV: D(p) is exactly the TV integrand (ctvlib.cpp:336-367), so the first gradient pass of a tv_gd call also
// returns the TV value "before descent" (tv_gd.cu:177-183) without a separate pass over the volume.
// GRAD = false: the TV value alone (the march with its single read of x, without the gradient stencil and the g store).
template <int TZ, bool WITH_TV, bool GRAD = true>
__global__ __launch_bounds__(256) void k_tv_grad_lds(const float *__restrict__ x, Halo h, float *__restrict__ g,
                                                      double *__restrict__ part, float eps, int n, int nx, int sx,
                                                      int yseg, double *__restrict__ part_tv)
{
    __shared__ float ring[4][TZ + 2][TVL_PITCH];      // x planes; row zi = column z0-1+zi, element si = slice s0-1+si
    __shared__ float rinv[2][TZ + 1][TVL_PITCH];      // R planes; rows zi = 0..TZ, elements si = 0..64
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nzb = (n + TZ - 1) / TZ;
    int bz = blockIdx.x % nzb;
    int bs = blockIdx.x / nzb;                 // slice chunk
    int y0 = blockIdx.y * yseg;
    int y1 = min(y0 + yseg, n);
    int z0 = bz * TZ, s0 = bs * 64;
    // full modulo: with n < TZ + 2 the halo columns (and with n = 1 the prefetched rows) wrap more than once
    auto zcol = [&](int zi) { int z = (z0 - 1 + zi) % n; return z < 0 ? z + n : z; };
    auto yrow = [&](int y) { int r = y % n; return r < 0 ? r + n : r; };
    constexpr int NR = (TZ + 2 + 3) / 4;       // plane rows per wave
    float v[NR], vh;
    auto fetch = [&](int y) {                  // rows (wave, wave+4, ...) x column lane+1, + halo columns
        int yy = yrow(y);
        int s = s0 + lane;
#pragma unroll
        for (int t = 0; t < NR; ++t) {
            int r = wave + 4 * t;
            v[t] = r < TZ + 2 ? tv_ld(x, h, yy * n + zcol(r), s, nx, sx) : 0.f;
        }
        vh = 0.f;
        if (wave == 3 && lane < 2 * (TZ + 2)) {
            int zi = lane >> 1, side = lane & 1;
            vh = tv_ld(x, h, yy * n + zcol(zi), side ? s0 + 64 : s0 - 1, nx, sx);
        }
    };
    auto stash = [&](int slot) {
#pragma unroll
        for (int t = 0; t < NR; ++t) {
            int r = wave + 4 * t;
            if (r < TZ + 2) ring[slot][r][lane + 1] = v[t];
        }
        if (wave == 3 && lane < 2 * (TZ + 2)) ring[slot][lane >> 1][(lane & 1) ? 65 : 0] = vh;
    };
    // R of the plane in slot a, whose +y neighbour plane is in slot b
    double tvacc = 0.0;
    auto compute_r = [&](int a, int b, int rslot, bool own_plane) {
        for (int e = threadIdx.x; e < (TZ + 1) * 65; e += 256) {
            int zi = e / 65, si = e - zi * 65;
            float c = ring[a][zi][si];
            float d1 = c - ring[a][zi][si + 1];
            float d2 = c - ring[b][zi][si];
            float d3 = c - ring[a][zi + 1][si];
            // R = 1/sqrt(q) from the hardware estimate plus one Newton step (<= 1 ulp); the IEEE sqrt followed by an
            // IEEE division costs 10 % of the whole pass.  D = q R is the TV integrand.
            // (explicit fma/mul intrinsics: the sequence must round identically in every instantiation of this kernel)
            float q_ = __fmaf_rn(d3, d3, __fmaf_rn(d2, d2, __fmaf_rn(d1, d1, eps)));
            float rr_ = tv_rsqrt(q_);
            float D = __fmul_rn(q_, rr_);
            rinv[rslot][zi][si] = rr_;
            if (WITH_TV && own_plane && zi >= 1 && si >= 1 && z0 + zi - 1 < n && s0 + si - 1 < nx) tvacc += (double)D;
        }
    };
    fetch(y0 - 1); stash(0);
    fetch(y0);     stash(1);
    fetch(y0 + 1); stash(2);
    __syncthreads();
    compute_r(0, 1, 0, false);                 // R(y0-1)
    double acc = 0.0;
    const int si = lane + 1;
    const int s = s0 + lane;
    for (int y = y0; y < y1; ++y) {
        int t = y - y0;
        int m0 = t & 3, m1 = (t + 1) & 3, m2 = (t + 2) & 3, m3 = (t + 3) & 3;   // slots of y-1, y, y+1, free
        int rc = (t + 1) & 1, rp = t & 1;                                      // R(y), R(y-1)
        bool more = y + 1 < y1;
        if (more) fetch(y + 2);                // in flight while this row is computed
        compute_r(m1, m2, rc, true);
        __syncthreads();
#pragma unroll
        for (int q = 0; GRAD && q < TZ / 4; ++q) {
            int zi = 1 + wave * (TZ / 4) + q;
            int z = z0 + zi - 1;
            if (z < n && s < nx) {
                float c = ring[m1][zi][si];
                float gv = tv_gval(c, ring[m1][zi][si + 1], ring[m2][zi][si], ring[m1][zi + 1][si], rinv[rc][zi][si],
                                   ring[m1][zi][si - 1], rinv[rc][zi][si - 1], ring[m0][zi][si], rinv[rp][zi][si],
                                   ring[m1][zi - 1][si], rinv[rc][zi - 1][si]);
                g[(size_t)(y * n + z) * sx + s] = gv;
                acc += (double)(gv * gv);
            }
        }
        if (more) stash(m3);                   // plane y+2 into the free slot
        __syncthreads();
    }
    if (GRAD) block_accumulate(acc, part);
    if (WITH_TV) {
        __syncthreads();
        block_accumulate(tvacc, part_tv);
    }
}

// ---- TV gradient, register march: no LDS, no barriers -----------------------------------------------------------
// What-if timing of k_tv_grad_lds shows half of its time in its own skeleton (LDS stash, two barriers per
// row, 15 LDS operations per output).  Here ONE WAVE owns TZ z-columns x 64 slices and marches along y with the rows
// y-1, y, y+1 of its TZ+2 columns in registers: z neighbours are other registers of the same lane, y neighbours are the
// rolling rows, slice neighbours come by DPP wave_shr / wave_shl.  The two values beyond a chunk's edges (slices s0-1
// and s0+64) are loaded into lanes 0 and 63 of a per-column edge register, which is exactly the DPP `old` operand the
// shifts leave in those lanes; R of the phantom slice s0-1 (needed by lane 0's R(p-i)) is the same formula evaluated on
// the edge registers.  Same arithmetic, operand order and rounding sequence as k_tv_grad_lds.
// GRAD = false: the TV value alone (rows y, y+1 only; no phantom slice, no gradient, no store).
// Round-2 experiments on this kernel (512^3, 288 us = 3.7 TB/s on its 8V compulsory bytes), none of which moved its time:
//  * the XCD-aware item map below cut the L2-side reads from 1.72x to 1.29x compulsory (PMC) -- the duplicate halo reads had
//    been Infinity-Cache hits, not HBM traffic;
//  * a form on float2 z-pairs (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32, -27 % vector instructions): 608 vs 598 us per
//    inner iteration; the whole library built WITHOUT packed fp32 (-target-feature -packed-fp32-ops): the same;
//  * a workgroup-cooperative form (the 4 waves of a workgroup = 4 adjacent chunks hand lane 63's R to the neighbour through
//    LDS instead of re-evaluating the phantom slice: -78 instructions per row; buffer loads with scalar row offsets: -28):
//    616 vs 602 us.
// With the gradient no longer stored (MODE below) the norm pass takes 231 us for 0.69 GB of reads and the update pass 332 us: a
// what-if build without the phantom-slice evaluation (-104 of ~440 instructions per row) runs an inner iteration in 470 instead
// of 530 us, and 4 z-columns per wave (68 VGPRs, 7 waves per SIMD instead of 4) in the same 530: the passes are about half
// instruction-bound, not occupancy-bound.  Handing R across chunk edges costs what it saves in every form tried.
// What did help a little: evaluating R at the phantom slice once per row for all columns on PACKED inputs (lane j = column j,
// two gather loads per row) instead of once per column on the edge registers: -72 vector instructions per row, an inner
// iteration 516 -> 500 us at 512 slices, 93 -> 90.5 us at 64 (same box, both libraries side by side).
// workgroups (4 waves) of the march kernels' item space, for the XCD-aware map above
inline unsigned tv_march_grid(int n, int tz, int nchunk, int nys)
{
    const int nzb = (n + tz - 1) / tz;
    if ((nzb & 7) == 0) return 8u * (unsigned)(((int64_t)(nzb >> 3) * nchunk * nys + 3) / 4);
    return (unsigned)(((int64_t)nzb * nchunk * nys + 3) / 4);
}

// MODE (round 2).  HBM WRITES are the scarce resource on this part (a 537 MB memset runs at 3.0 TB/s, a read stream at ~6;
// tools/whatif_sart.py), and a tv_gd inner iteration as "gradient pass (write g) + update pass (read x, g; write x)" writes the
// volume twice.  So the gradient is never stored:
//   TVM_NORM    the pass only accumulates sum g^2 (and, WITH_TV, the TV value): reads x, writes nothing;
//   TVM_UPDATE  the pass re-evaluates g (bit for bit the same arithmetic) and writes x_new = x - (dPOCS g)/||g|| into a SECOND
//               buffer (neighbours still read the old x), clamp / wrapped halo planes / tracked norm + snapshot as in
//               k_tv_update.  One volume write per inner iteration instead of two, 8 instead of 12 bytes read.
//   TVM_STORE   the round-1 form (g stored; k_tv_update applies it): kept for the A/B option and the other kernel forms.
enum { TVM_STORE = 0, TVM_NORM = 1, TVM_UPDATE = 2, TVM_VALUE = 3 };   // TVM_VALUE (k_tv_march4 only): the TV value alone, no gradient
struct TvUpd { float *x_out; const double *gnorm2; float dPOCS; int clamp; float *track; float *wrap_lo; float *wrap_hi;
               int stream;      // stream: non-temporal stores of x_new / the snapshot (slabs beyond the Infinity Cache: -3 %; thin slabs: +3 %)
               // slab-sharded descent (round 6): the update pass also advances the halo planes -- the neighbours' update of the slices they
               // hold, k_halo_apply's expression on the gradient planes received from them -- into a SECOND pair of planes (this pass still
               // reads the old ones), instead of one more launch per inner iteration (k_tv_march4 only; null = not asked for)
               const float *hg_lo; const float *hg_hi; float *ho_lo; float *ho_hi; };

template <int TZ, bool WITH_TV, bool GRAD = true, int MODE = TVM_STORE>
__global__ __launch_bounds__(256) void k_tv_grad_reg(const float *__restrict__ x, Halo h, float *__restrict__ g,
                                                      double *__restrict__ part, float eps, int n, int nx, int sx,
                                                      int yseg, double *__restrict__ part_tv, TvUpd up = TvUpd{})
{
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nzb = (n + TZ - 1) / TZ, nchunk = (nx + 63) >> 6   /* computed width, not the row pitch: the pitch may carry padding */, nys = (n + yseg - 1) / yseg;
    double acc = 0.0, tvacc = 0.0;
    float nrm_ = 1.f;
    if (MODE == TVM_UPDATE) nrm_ = tv_step_len(up.dPOCS, up.gnorm2);      // the step length dPOCS / ||g||
    // Item = (y segment, z block, chunk).  Neighbouring z blocks share two of their ten columns and neighbouring chunks a
    // slice on either side: when the neighbours run on different XCDs every shared line is fetched from HBM once per XCD
    // (PMC, round 2: 1.72x the compulsory reads, and the kernel is bound by exactly that traffic: 1.46 GB in 288 us).
    // Workgroups b and b+8 share an XCD, so each XCD is given a contiguous slab of z blocks and walks it chunk-fastest:
    // the neighbours are then in flight on the same L2 at the same time.  (tv_march_items sizes the grid.)
    int bs, bz, ys;
    bool live;
    if ((nzb & 7) == 0) {
        const int zpx = nzb >> 3;
        const int64_t li = (int64_t)(blockIdx.x >> 3) * 4 + wave;
        bs = (int)(li % nchunk); bz = (int)(blockIdx.x & 7) * zpx + (int)((li / nchunk) % zpx); ys = (int)(li / ((int64_t)nchunk * zpx));
        live = ys < nys;
    } else {
        const int64_t item = (int64_t)blockIdx.x * 4 + wave;        // chunk fastest
        bs = (int)(item % nchunk); bz = (int)((item / nchunk) % nzb); ys = (int)(item / ((int64_t)nchunk * nzb));
        live = ys < nys;
    }
    if (live) {
        const int y0 = ys * yseg, y1 = min(y0 + yseg, n);
        const int z0 = bz * TZ, s0 = bs * 64, s = s0 + lane;
        // edge register: lane 0 <- slice s0-1, lane 63 <- slice s0+64; the other lanes re-read their own slice (same
        // cache lines as the column load: an unconditional load costs less than a two-lane branch per column)
        const int se = lane == 0 ? s0 - 1 : (lane == 63 ? s0 + 64 : s);
        int zc[TZ + 2];
#pragma unroll
        for (int j = 0; j < TZ + 2; ++j) { int z = (z0 - 1 + j) % n; zc[j] = z < 0 ? z + n : z; }
        auto yrow = [&](int y) { int r = y % n; return r < 0 ? r + n : r; };
        float cm[TZ + 2], c0[TZ + 2], cp[TZ + 2], cn[TZ + 2], E0[TZ + 2], Ep[TZ + 2], En[TZ + 2], Rm[TZ + 1], R0[TZ + 1];
        // a chunk strictly inside the slab needs no halo planes: wave-uniform row pointers + a lane offset
        const bool interior = s0 > 0 && s0 + 64 < nx;
        auto fetch = [&](int y, float *c, float *E) {
            int yy = yrow(y) * n;
            if (interior) {
#pragma unroll
                for (int j = 0; j < TZ + 2; ++j) {
                    const float *rp = x + (size_t)(yy + zc[j]) * sx;   // wave-uniform; unsigned lane offsets -> saddr form
                    c[j] = nt_ld<8>(rp + (unsigned)s);    // (non-temporal on all columns: +14 %; on the wave's own columns only: neutral)
                    E[j] = nt_ld<8>(rp + (unsigned)se);
                }
            } else {
#pragma unroll
                for (int j = 0; j < TZ + 2; ++j) {
                    c[j] = tv_ld(x, h, yy + zc[j], s, nx, sx);
                    E[j] = tv_ld(x, h, yy + zc[j], se, nx, sx);
                }
            }
        };
        // R = 1/sqrt(q), q = eps + d1^2 + d2^2 + d3^2: the rounding sequence of k_tv_grad_lds::compute_r
#define TVR_RINV(C, IP, JP, KP, RR, DD)                                                                   \
        {                                                                                                 \
            float d1_ = (C) - (IP), d2_ = (C) - (JP), d3_ = (C) - (KP);                                   \
            float q_ = __fmaf_rn(d3_, d3_, __fmaf_rn(d2_, d2_, __fmaf_rn(d1_, d1_, eps)));                \
            RR = tv_rsqrt(q_);                                                                            \
            DD = __fmul_rn(q_, RR);                                                                       \
        }
        auto shr = [&](float old, float v) {                    // lane l <- lane l-1 ; lane 0 keeps `old`
            return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, false));
        };
        auto shl = [&](float old, float v) {                    // lane l <- lane l+1 ; lane 63 keeps `old`
            return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, false));
        };
        // Phantom slice s0-1, packed: lane j (< TZ+2) holds column j's value at slice s0-1 (PE*) and at slice s0 (PC*) of a
        // row, so R at the phantom slice is ONE evaluation per row for all columns (lanes 1..TZ) instead of one full-wave
        // evaluation per column of which only lane 0 was used (104 of ~440 vector instructions per row).  Two gather loads
        // per row (hits: the column loads of this wave and of the neighbouring chunk touch the same lines).
        float PE0 = 0.f, PEp = 0.f, PEn = 0.f, PC0 = 0.f, PCn = 0.f;
        int zl;
        { int z = (z0 - 1 + (lane < TZ + 2 ? lane : 0)) % n; zl = z < 0 ? z + n : z; }
        auto fetch_ph = [&](int y, float &pe, float &pc) {
            int pix = yrow(y) * n + zl;
            pe = tv_ld(x, h, pix, s0 - 1, nx, sx);
            pc = tv_ld(x, h, pix, s0, nx, sx);
        };
        if (GRAD) fetch(y0 - 1, cm, En);
        fetch(y0, c0, E0);
        fetch(y0 + 1, cp, Ep);
        if (GRAD) { float pcp; fetch_ph(y0, PE0, PC0); fetch_ph(y0 + 1, PEp, pcp); PCn = pcp; }
        // R of row y0-1 for the output columns (its +y neighbour is row y0)
#pragma unroll
        for (int j = 1; GRAD && j <= TZ; ++j) {
            float xip = shl(En[j], cm[j]), dd;
            TVR_RINV(cm[j], xip, c0[j], cm[j + 1], Rm[j], dd)
            (void)dd;
        }
        for (int y = y0; y < y1; ++y) {
            float PCp = PCn;                                    // slice s0 of row y+1 (fetched with its PE)
            if (y + 1 < y1) { fetch(y + 2, cn, En); if (GRAD) fetch_ph(y + 2, PEn, PCn); }   // in flight while this row is computed
            // R at the phantom slice of row y, all columns at once: lane j <- column j
            float REp;
            if (GRAD) {
                float kp = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, PE0), 0x101, 0xf, 0xf, false));   // row_shl:1 -> column j+1
                float dd;
                TVR_RINV(PE0, PC0, PEp, kp, REp, dd)
                (void)dd;
            }
            float xip[TZ + 1];
#pragma unroll
            for (int j = 0; j <= TZ; ++j) {
                float dd;
                xip[j] = shl(E0[j], c0[j]);
                TVR_RINV(c0[j], xip[j], cp[j], c0[j + 1], R0[j], dd)
                if (WITH_TV && j >= 1 && z0 + j - 1 < n && s < nx) tvacc += (double)dd;
            }
#pragma unroll
            for (int j = 1; GRAD && j <= TZ; ++j) {
                // R at the phantom slice s0-1 (lane j of the packed evaluation), then R(p-i) by the shift
                float re = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, REp), j));
                float rim = shr(re, R0[j]);
                float xim = shr(E0[j], c0[j]);
                float c = c0[j];
                float gv = tv_gval(c, xip[j], cp[j], c0[j + 1], R0[j], xim, rim, cm[j], Rm[j], c0[j - 1], R0[j - 1]);
                int z = z0 + j - 1;
                if (z < n && s < nx) {
                    if (MODE == TVM_STORE) {
                        float *gr = g + (size_t)(y * n + z) * sx;
                        gr[(unsigned)s] = gv;
                        acc += (double)(gv * gv);
                    } else if (MODE == TVM_NORM) {
                        acc += (double)(gv * gv);
                        if (up.wrap_lo) {      // slab-sharded descent: the gradient's first / last slice for the neighbours
                            const size_t pix = (size_t)(y * n + z);
                            if (s == 0) up.wrap_hi[pix] = gv;
                            if (s == nx - 1) up.wrap_lo[pix] = gv;
                        }
                    } else {   // TVM_UPDATE: the expression of k_tv_update
                        const size_t pix = (size_t)(y * n + z);
                        float v = tv_step(c, gv, nrm_);   // = k_tv_update's step
                        if (up.clamp) v = fmaxf(v, 0.f);
                        if (up.stream) __builtin_nontemporal_store(v, up.x_out + pix * sx + (unsigned)s);
                        else up.x_out[pix * sx + (unsigned)s] = v;
                        if (up.wrap_lo) {
                            if (s == 0) up.wrap_hi[pix] = v;
                            if (s == nx - 1) up.wrap_lo[pix] = v;
                        }
                        if (up.track) {
                            float *tr = up.track + pix * sx;
                            float d = v - tr[(unsigned)s];
                            acc += (double)(d * d);
                            if (up.stream) __builtin_nontemporal_store(v, tr + (unsigned)s);
                            else tr[(unsigned)s] = v;
                        }
                    }
                }
            }
            // rotate the rows by register moves (rotating them by name, a 4x unrolled loop, costs a wave of occupancy:
            // 141 VGPRs, 10 % slower)
#pragma unroll
            for (int j = 0; j < TZ + 2; ++j) { cm[j] = c0[j]; c0[j] = cp[j]; cp[j] = cn[j]; E0[j] = Ep[j]; Ep[j] = En[j]; }
            PE0 = PEp; PEp = PEn; PC0 = PCp;
#pragma unroll
            for (int j = 1; j <= TZ; ++j) Rm[j] = R0[j];
        }
#undef TVR_RINV
    }
    if (GRAD) block_accumulate(acc, part);
    if (WITH_TV) {
        __syncthreads();
        block_accumulate(tvacc, part_tv);
    }
}

// ---- register march without the row rotation ---------------------------------------------------------------------------
// In k_tv_grad_reg a quarter of the vector instructions of a row are register moves: the rows y-1, y, y+1 and the prefetched
// y+2 (and their edge registers) rotate by v_mov every row.  Here the four rows live in four fixed slots and the loop is
// unrolled four times with the slots' roles rotating by NAME, and the per-column edge registers are gone: the values beyond
// the chunk's edges are gathered once per row into packed registers (lane j = column j: slice s0-1, slice s0+64 and, for
// the phantom slice's R, slice s0) and reach lane 0 / lane 63 of a column through v_readlane + the DPP `old` operand.
// Same arithmetic, operand order and rounding sequence as k_tv_grad_reg (bit-identical); gradient modes only.
// 92-98 VGPRs (5 waves per SIMD; k_tv_grad_reg: 112-121, 4 waves).  Measured, same box: a TV-GD inner iteration 496 -> 446 us at
// 512 slices, 86 -> 76 us at 64.  16 z-columns per wave (18 loaded for 16 outputs instead of 10 for 8; 150 VGPRs): 481-496 us.
// Overlapping chunks (a wave loads 64 slices and owns the 62 in the middle, so every slice shift is a plain DPP: no packed edge
// values, no readlane fix-ups, no phantom R; 71-78 VGPRs): fewer instructions but 475 against 429 us at 512 slices and 103 against
// 76 at 64 -- the misaligned 248-byte rows and the extra chunk cost more than the ~20 % of vector instructions they save.
// Occupancy: the update pass fits 96 VGPRs (5 waves per SIMD) without a spill, the norm pass does not (62 spilled registers at 5
// waves: 816 us measured with an earlier form); TV4_UPD_WAVES (build-time) asks for 5 on the update pass only -- measured 377.7 vs
// 376.0 us per inner iteration at 512 slices and 65.6 vs 57.8 at 64: the compiler's own choice (120 VGPRs, 4 waves) stays.
#ifndef TV4_PACKED
#define TV4_PACKED 1
#endif
#ifndef TV4_PACKED_EDGE
#define TV4_PACKED_EDGE 0      // the predicated EDGE forms spill with the pairs (28-36 B) and lose: 450 vs 415 us at 500 slices, 127 vs 112 at 100
#endif
#ifndef TV4_UPD_WAVES
#define TV4_UPD_WAVES 4
#endif
#define TV4_OCC __attribute__((amdgpu_waves_per_eu((MODE == TVM_UPDATE && !TRACK) ? TV4_UPD_WAVES : 4, (MODE == TVM_UPDATE && !TRACK) ? TV4_UPD_WAVES : 8)))
template <int TZ, bool WITH_TV, int MODE, bool EDGE, bool TRACK = false, bool STREAM = false>
__global__ __launch_bounds__(256) TV4_OCC void k_tv_march4(const float *__restrict__ x, Halo h, double *__restrict__ part, float eps,
                                                    int n, int nx, int sx, int yseg, double *__restrict__ part_tv, TvUpd up)
{
    static_assert(MODE == TVM_NORM || MODE == TVM_UPDATE || MODE == TVM_VALUE, "modes without a stored gradient");
    static_assert(MODE != TVM_VALUE || WITH_TV, "the value mode sums the TV integrand");
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nzb = (n + TZ - 1) / TZ, nchunk = (nx + 63) >> 6   /* computed width, not the row pitch: the pitch may carry padding */, nys = (n + yseg - 1) / yseg;
    double acc = 0.0, tvacc = 0.0;
    float nrm_ = 1.f;
    if (MODE == TVM_UPDATE) nrm_ = tv_step_len(up.dPOCS, up.gnorm2);      // the step length dPOCS / ||g||
    int bs, bz, ys;
    if ((nzb & 7) == 0) {       // the XCD-aware item map of k_tv_grad_reg
        const int zpx = nzb >> 3;
        const int64_t li = (int64_t)(blockIdx.x >> 3) * 4 + wave;
        bs = (int)(li % nchunk); bz = (int)(blockIdx.x & 7) * zpx + (int)((li / nchunk) % zpx); ys = (int)(li / ((int64_t)nchunk * zpx));
    } else {
        const int64_t item = (int64_t)blockIdx.x * 4 + wave;
        bs = (int)(item % nchunk); bz = (int)((item / nchunk) % nzb); ys = (int)(item / ((int64_t)nchunk * nzb));
    }
    if (ys < nys) {
        const int y0 = ys * yseg, y1 = min(y0 + yseg, n);
        const int z0 = bz * TZ, s0 = bs * 64, s = s0 + lane;
        int zc[TZ + 2];
#pragma unroll
        for (int j = 0; j < TZ + 2; ++j) { int z = (z0 - 1 + j) % n; zc[j] = z < 0 ? z + n : z; }
        int zl;
        { int z = (z0 - 1 + (lane < TZ + 2 ? lane : 0)) % n; zl = z < 0 ? z + n : z; }
        auto yrow = [&](int y) { int r = y % n; return r < 0 ? r + n : r; };
        float rows[4][TZ + 2], Ta[TZ + 1], Tb[TZ + 1];     // Ta / Tb: the -y terms (x_jp - c) R of the previous row, alternating
        float pe[4], pf[4], pc[4];          // packed edge values of the row in slot k: slices s0-1, s0+64, s0
        // EDGE = false (the launcher picks the instantiation when the slab is a multiple of 64 slices and the image side a
        // multiple of TZ -- every BASELINE shape): every lane of every wave owns a voxel, so the row carries NO predicate and no
        // branch: the columns load straight from the volume through a row pointer formed once per row, whether the slice below /
        // above the chunk is a neighbour's halo plane is a wave-uniform question answered ONCE (scalar select of the base pointer,
        // hoisted per-lane offset), the eight results of a row are stored back to back after the arithmetic, and the wrap planes
        // are written by the two chunks that hold them.  Round 2 predicated every column (s_and_saveexec + s_cbranch_execz + a
        // join per column: ~25 tiny basic blocks per row, which also kept the scheduler from filling the DPP / readlane hazard
        // slots: 28 s_nop per row) and sent every load of a chunk touching the slab's first or last slice -- on a 64- or 128-slice
        // slab: all of them -- through a per-lane three-way address select (~8 vector instructions per load).
        // EDGE = true keeps the predicated per-lane form for everything else (ragged last chunk, partial last z block).
        const bool lo_in = s0 > 0, hi_in = s0 + 64 < nx;
        const unsigned zls = (unsigned)zl * (unsigned)sx;
        const unsigned off_lo = lo_in ? zls + (unsigned)(s0 - 1) : (unsigned)zl, off_hi = hi_in ? zls + (unsigned)(s0 + 64) : (unsigned)zl;
        // (the norm pass that also sums the TV value is two registers over the 128 of four waves per SIMD with the ten offsets
        // held: that one instantiation forms its addresses per access instead -- 9 dwords of scratch otherwise)
        constexpr bool SOFF = !EDGE && !(WITH_TV && MODE == TVM_NORM);   // (and the predicated EDGE forms spill with them too)
        unsigned vb[TZ + 2];                // byte offsets of the lane's voxel in the columns of a row: loop invariants
#pragma unroll
        for (int j = 0; j < TZ + 2; ++j) vb[j] = ((unsigned)zc[j] * (unsigned)sx + (unsigned)s0 + (unsigned)lane) * 4u;
        unsigned eb_c = (zls + (unsigned)s0) * 4u, eb_lo = off_lo * 4u, eb_hi = off_hi * 4u;
        auto fetch = [&](int y, float *c, float &e_lo, float &e_hi, float &e_c) __attribute__((always_inline)) {
            int yy = yrow(y) * n;
            if (!EDGE) {
                const float *rowp = x + (size_t)yy * sx;            // wave-uniform
                if constexpr (SOFF) {
                    const SBase rb = sgpr_base(rowp);
#pragma unroll
                    for (int j = 0; j < TZ + 2; ++j) c[j] = ld_so(rb, vb[j]);
                    e_c = ld_so(rb, eb_c);
                    e_lo = ld_so(sgpr_base(lo_in ? rowp : h.lo + yy), eb_lo);
                    e_hi = ld_so(sgpr_base(hi_in ? rowp : h.hi + yy), eb_hi);
                } else {
#pragma unroll
                    for (int j = 0; j < TZ + 2; ++j) c[j] = (rowp + ((unsigned)zc[j] * (unsigned)sx + (unsigned)s0))[(unsigned)lane];
                    e_c = rowp[zls + (unsigned)s0];
                    e_lo = (lo_in ? rowp : h.lo + yy)[off_lo];
                    e_hi = (hi_in ? rowp : h.hi + yy)[off_hi];
                }
            } else {
#pragma unroll
                for (int j = 0; j < TZ + 2; ++j) c[j] = tv_ld(x, h, yy + zc[j], s, nx, sx);
                e_lo = tv_ld(x, h, yy + zl, s0 - 1, nx, sx);
                e_hi = tv_ld(x, h, yy + zl, s0 + 64, nx, sx);
                e_c = tv_ld(x, h, yy + zl, s0, nx, sx);
            }
        };
        const float vmin = up.clamp ? 0.f : -INFINITY;              // positivity as one v_max whatever the flag
        const bool planes = up.wrap_lo != nullptr && (EDGE || s0 == 0 || s0 + 64 == nx);   // this chunk holds slice 0 or nx-1
        // R = 1/sqrt(q) with the three differences it is made of left in D1..D3 (they are the numerators of the backward terms)
#define TV4_RINV(C, IP, JP, KP, RR, DD, D1, D2, D3)                                                       \
        {                                                                                                 \
            D1 = (C) - (IP); D2 = (C) - (JP); D3 = (C) - (KP);                                            \
            float q_ = __fmaf_rn(D3, D3, __fmaf_rn(D2, D2, __fmaf_rn(D1, D1, eps)));                      \
            RR = tv_rsqrt(q_);                                                                            \
            DD = __fmul_rn(q_, RR);                                                                       \
        }
        auto shr = [&](float old, float v) {                    // lane l <- lane l-1 ; lane 0 keeps `old`
            return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, false));
        };
        auto shl = [&](float old, float v) {                    // lane l <- lane l+1 ; lane 63 keeps `old`
            return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, false));
        };
        auto col = [&](float packed, int j) {                   // column j's value of a packed register, wave-uniform
            return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, packed), j));
        };
        // one row: c0 / cp = rows y, y+1; cn receives row y+2; Tp = the -y terms formed in row y-1, Tn receives this row's.
        // (tv_gval's expression, term by term: G1 = v1n R(p); t2 = (x_ip - c) R of the slice below, shifted in; Tp; Tk of column j-1)
        auto row = [&](int y, const float *c0, const float *cp, float *cn, float pe0, float pf0, float pc0, float pep,
                       float &pen, float &pfn, float &pcn, const float *Tp, float *Tn
                       ) __attribute__((always_inline)) {
            if (y + 1 < y1) fetch(y + 2, cn, pen, pfn, pcn);    // in flight while this row is computed
            float TEp = 0.f;
            float out[TZ + 1];                                  // the row's results: g (norm pass) or x_new (update pass)
            float tk_prev = 0.f;                                // -(x_kp - c) R of column j-1: minus the -z term of column j
            if constexpr (TV4_PACKED && (!EDGE || TV4_PACKED_EDGE)) {
            // The columns two at a time on packed fp32 instructions (nc_mul2 / nc_sub2 / fma2: same roundings as the scalar
            // column loop below, half the issue slots).  Column 0, of which only the -z term is needed, shares its evaluation
            // with the slice-direction term of lane 0.
            static_assert((TZ & 1) == 0, "column pairs");
#define TV4_RINV2(C, IP, JP, KP, RR, DD, D1, D2, D3)                                                      \
            {                                                                                             \
                D1 = (C) - (IP); D2 = (C) - (JP); D3 = (C) - (KP);                                        \
                const v2f q_ = fma2(D3, D3, fma2(D2, D2, fma2(D1, D1, v2f{eps, eps})));                   \
                RR = v2f{tv_rsqrt(q_.x), tv_rsqrt(q_.y)};                                                 \
                DD = nc_mul2(q_, RR);                                                                     \
            }
            {
                const float kpe = MODE != TVM_VALUE ? shl(0.f, pe0) : 0.f;   // column j+1 (wave shift: the packed columns may pass lane 15)
                const v2f c = {c0[0], pe0}, ip = {shl(col(pf0, 0), c0[0]), pc0}, jp = {cp[0], pep}, kp = {c0[1], kpe};
                v2f r, dd, d1, d2, d3;
                TV4_RINV2(c, ip, jp, kp, r, dd, d1, d2, d3)
                (void)dd; (void)d2;
                tk_prev = nc_mul(d3.x, r.x);
                if (MODE != TVM_VALUE) TEp = nc_mul(d1.y, r.y);         // = -(x[s0] - x[s0-1]) R(s0-1): the terms are kept negated ...
            }
#pragma unroll
            for (int j = 1; j < TZ; j += 2) {
                const v2f c = {c0[j], c0[j + 1]}, xip = {shl(col(pf0, j), c0[j]), shl(col(pf0, j + 1), c0[j + 1])};
                const v2f jp = {cp[j], cp[j + 1]}, kp = {c0[j + 1], c0[j + 2]};
                v2f r, dd, d1, d2, d3;
                TV4_RINV2(c, xip, jp, kp, r, dd, d1, d2, d3)
                const bool ok0 = !EDGE || (z0 + j - 1 < n && s < nx), ok1 = !EDGE || (z0 + j < n && s < nx);
                if (WITH_TV) { tvacc += (double)(ok0 ? dd.x : 0.f); tvacc += (double)(ok1 ? dd.y : 0.f); }
                const v2f tk = nc_mul2(d3, r);                  // .y is handed to the next pair
                if (MODE != TVM_VALUE) {
                    const v2f ti = nc_mul2(d1, r);              // -(x_ip - c) R: minus the -slice term of the lane above
                    const v2f tn = nc_mul2(d2, r);              // -(x_jp - c) R: minus the -y term of the next row
                    Tn[j] = tn.x; Tn[j + 1] = tn.y;
                    #if TV_V1N_DIFFS
                    const v2f g1 = nc_mul2(nc_add2(nc_add2(d1, d2), d3), r);   // tv_v1n: the three forward differences R is made of
#else
                    const v2f g1 = nc_mul2(nc_sub2(nc_sub2(fma2(v2f{3.0f, 3.0f}, c, -xip), jp), kp), r);
#endif
                    const v2f t2 = {shr(col(TEp, j), ti.x), shr(col(TEp, j + 1), ti.y)};
                    const v2f tp = {Tp[j], Tp[j + 1]}, tkp = {tk_prev, tk.x};
                    const v2f gv = nc_sub2(nc_sub2(nc_sub2(g1, t2), tp), tkp);   // the terms are kept negated and subtracted: a - (-t) == a + t
                    if (MODE == TVM_NORM) {
                        out[j] = gv.x; out[j + 1] = gv.y;
                        const v2f g2 = gv * gv;
                        acc += (double)(ok0 ? g2.x : 0.f);
                        acc += (double)(ok1 ? g2.y : 0.f);
                    } else {
                        const v2f xn = fma2(-gv, v2f{nrm_, nrm_}, c);            // tv_step, the expression of k_tv_update
                        out[j] = fmaxf(xn.x, vmin); out[j + 1] = fmaxf(xn.y, vmin);
                    }
                }
                tk_prev = tk.y;
            }
#undef TV4_RINV2
            } else {
            if (MODE != TVM_VALUE) {   // the slice-direction term for lane 0, all columns at once (lane j <- column j): (x[s0] - x[s0-1]) R(s0-1)
                float kp = shl(0.f, pe0), r_, dd_, d1, d2, d3;  // column j+1 (wave shift: the packed columns may pass lane 15)
                TV4_RINV(pe0, pc0, pep, kp, r_, dd_, d1, d2, d3)
                (void)dd_; (void)d2; (void)d3;
                TEp = nc_mul(d1, r_);                           // = -(x[s0] - x[s0-1]) R(s0-1): the terms are kept negated ...
            }
#pragma unroll
            for (int j = 0; j <= TZ; ++j) {                     // one pass over the columns: R, the shared products, the gradient
                float dd, d1, d2, d3, r;
                const float c = c0[j], xip = shl(col(pf0, j), c);
                TV4_RINV(c, xip, cp[j], c0[j + 1], r, dd, d1, d2, d3)
                if (WITH_TV && j >= 1) tvacc += (double)((!EDGE || (z0 + j - 1 < n && s < nx)) ? dd : 0.f);
                const float tk = nc_mul(d3, r);                 // handed to column j+1
                if (j >= 1 && MODE != TVM_VALUE) {
                    const float ti = nc_mul(d1, r);             // -(x_ip - c) R: minus the -slice term of the lane above
                    Tn[j] = nc_mul(d2, r);                      // -(x_jp - c) R: minus the -y term of the next row
                    #if TV_V1N_DIFFS
                    const float g1 = nc_mul(nc_add(nc_add(d1, d2), d3), r);     // tv_v1n: d1..d3 are c - x_ip, c - x_jp, c - x_kp
#else
                    const float g1 = nc_mul(nc_sub(nc_sub(__fmaf_rn(3.0f, c, -xip), cp[j]), c0[j + 1]), r);
#endif
                    const float t2 = shr(col(TEp, j), ti);
                    const float gv = nc_sub(nc_sub(nc_sub(g1, t2), Tp[j]), tk_prev);   // the terms are kept negated and subtracted: a - (-t) == a + t
                    const bool ok = !EDGE || (z0 + j - 1 < n && s < nx);
                    if (MODE == TVM_NORM) {
                        out[j] = gv;
                        float g2 = gv * gv;
                        acc += (double)(ok ? g2 : 0.f);
                    } else {
                        out[j] = fmaxf(tv_step(c, gv, nrm_), vmin);   // the expression of k_tv_update
                    }
                }
                tk_prev = tk;
            }
            }
            const size_t pix0 = (size_t)(y * n + z0);                 // wave-uniform
            if (MODE == TVM_UPDATE) {
                // (column j >= 1 of a row is pixel y n + z0 + j - 1 whenever it is stored: vb[j] is its offset in the output row too)
                const SBase xo = sgpr_base(up.x_out + (size_t)y * n * sx);
                float *xo_e = up.x_out + pix0 * sx + (unsigned)s0;          // (EDGE: address per access)
#pragma unroll
                for (int j = 1; j <= TZ; ++j) {
                    if (EDGE && !(z0 + j - 1 < n && s < nx)) continue;
                    if constexpr (SOFF) st_so<STREAM>(xo, vb[j], out[j]);
                    else if (STREAM) __builtin_nontemporal_store(out[j], xo_e + (size_t)(j - 1) * sx + (unsigned)lane);
                    else (xo_e + (size_t)(j - 1) * sx)[(unsigned)lane] = out[j];
                }
                if (TRACK) {
                    const SBase tr = sgpr_base(up.track + (size_t)y * n * sx);
                    float *tr_e = up.track + pix0 * sx + (unsigned)s0;
                    float told[TZ + 1];
#pragma unroll
                    for (int j = 1; j <= TZ; ++j) {
                        if constexpr (SOFF) told[j] = ld_so(tr, vb[j]);
                        else told[j] = (z0 + j - 1 < n && s < nx) ? (tr_e + (size_t)(j - 1) * sx)[(unsigned)lane] : out[j];
                    }
#pragma unroll
                    for (int j = 1; j <= TZ; ++j) {
                        float d = out[j] - told[j];
                        acc += (double)(d * d);
                        if (EDGE && !(z0 + j - 1 < n && s < nx)) continue;
                        if constexpr (SOFF) st_so<STREAM>(tr, vb[j], out[j]);
                        else if (STREAM) __builtin_nontemporal_store(out[j], tr_e + (size_t)(j - 1) * sx + (unsigned)lane);
                        else (tr_e + (size_t)(j - 1) * sx)[(unsigned)lane] = out[j];
                    }
                }
            }
            if (planes) {      // the result's first / last slice: the wrap planes (single slab) or what the neighbours receive (sharded)
                if (s == 0) {                                       // one lane, the row's eight values back to back
#pragma unroll
                    for (int j = 1; j <= TZ; ++j) if (!EDGE || z0 + j - 1 < n) up.wrap_hi[pix0 + (j - 1)] = out[j];
                }
                if (s == nx - 1) {
#pragma unroll
                    for (int j = 1; j <= TZ; ++j) if (!EDGE || z0 + j - 1 < n) up.wrap_lo[pix0 + (j - 1)] = out[j];
                }
            }
            if (MODE == TVM_UPDATE && up.hg_lo != nullptr && (EDGE || s0 == 0 || s0 + 64 == nx)) {
                if (!EDGE) {
                    // the old planes' values of this row are already here: lane j of the packed edge registers holds column z0 - 1 + j
                    // of slice s0 - 1 (pe0) / s0 + 64 (pf0) -- the halo planes themselves in the chunks that touch them.  Lanes 1 .. TZ
                    // advance the row's eight pixels of each plane at once: one 32-byte load of the received gradient plane, one store.
                    const unsigned hp = (unsigned)(y * n) + (unsigned)zl;
                    if (lane >= 1 && lane <= TZ) {
                        if (s0 == 0) up.ho_lo[hp] = fmaxf(tv_step(pe0, up.hg_lo[hp], nrm_), vmin);
                        if (s0 + 64 == nx) up.ho_hi[hp] = fmaxf(tv_step(pf0, up.hg_hi[hp], nrm_), vmin);
                    }
                } else {
                    if (s == 0) {
#pragma unroll
                        for (int j = 1; j <= TZ; ++j)
                            if (z0 + j - 1 < n) up.ho_lo[pix0 + (j - 1)] = fmaxf(tv_step(h.lo[pix0 + (j - 1)], up.hg_lo[pix0 + (j - 1)], nrm_), vmin);
                    }
                    if (s == nx - 1) {
#pragma unroll
                        for (int j = 1; j <= TZ; ++j)
                            if (z0 + j - 1 < n) up.ho_hi[pix0 + (j - 1)] = fmaxf(tv_step(h.hi[pix0 + (j - 1)], up.hg_hi[pix0 + (j - 1)], nrm_), vmin);
                    }
                }
            }
        };
        fetch(y0 - 1, rows[0], pe[0], pf[0], pc[0]);
        fetch(y0, rows[1], pe[1], pf[1], pc[1]);
        fetch(y0 + 1, rows[2], pe[2], pf[2], pc[2]);
        // the -y terms of row y0: (x(y0) - x(y0-1)) R(row y0-1) for the output columns
#pragma unroll
        for (int j = 1; MODE != TVM_VALUE && j <= TZ; ++j) {
            float xi = shl(col(pf[0], j), rows[0][j]), r, dd, d1, d2, d3;
            TV4_RINV(rows[0][j], xi, rows[1][j], rows[0][j + 1], r, dd, d1, d2, d3)
            (void)dd; (void)d1; (void)d3;
            Ta[j] = nc_mul(d2, r);
        }
#define TV4_ROW(S0, SP, SN, TP, TN) row(y, rows[S0], rows[SP], rows[SN], pe[S0], pf[S0], pc[S0], pe[SP], pe[SN], pf[SN], pc[SN], TP, TN)
        for (int y = y0; y < y1;) {
            TV4_ROW(1, 2, 3, Ta, Tb); if (++y >= y1) break;
            TV4_ROW(2, 3, 0, Tb, Ta); if (++y >= y1) break;
            TV4_ROW(3, 0, 1, Ta, Tb); if (++y >= y1) break;
            TV4_ROW(0, 1, 2, Tb, Ta); ++y;
        }
#undef TV4_ROW
#undef TV4_RINV
    }
    if (MODE != TVM_VALUE) block_accumulate(acc, part);
    if (WITH_TV) {
        __syncthreads();
        block_accumulate(tvacc, part_tv);
    }
}

// Slab-sharded TV descent with ONE communication round per inner iteration: a rank receives the gradient's boundary slices
// of its neighbours (with the global sum g^2) and advances its halo planes itself -- the neighbour's update of those slices,
// same expression, same bits -- instead of receiving the updated slices in a second round.
__global__ __launch_bounds__(256) void k_halo_apply(float *__restrict__ halo_lo, float *__restrict__ halo_hi,
                                                     const float *__restrict__ g_lo, const float *__restrict__ g_hi,
                                                     const double *__restrict__ gnorm2, float dPOCS, int clamp, int npix)
{
    const float len = tv_step_len(dPOCS, gnorm2);
    int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= npix) return;
    float a = tv_step(halo_lo[p], g_lo[p], len);
    float b = tv_step(halo_hi[p], g_hi[p], len);
    if (clamp) { a = fmaxf(a, 0.f); b = fmaxf(b, 0.f); }
    halo_lo[p] = a;
    halo_hi[p] = b;
}

// x -= dPOCS * g / ||g||   (ctvlib.cpp:452-458); gnorm2 = global sum g^2 on the device; optional clamp (:461)
// TRACK: also sum (x_new - track)^2 -> part[] and track = x_new (the step norm and snapshot after the TV descent)
// wrap_lo / wrap_hi (single slab, periodic in the slice direction): the pass also leaves the new last / first slice
// in the halo planes the next gradient pass reads, instead of a gather launch between the two.
template <bool TRACK>
__global__ __launch_bounds__(256) void k_tv_update(f4 *__restrict__ x, const f4 *__restrict__ g,
                                                    const double *__restrict__ gnorm2, float dPOCS, int clamp,
                                                    int64_t n4, f4 *__restrict__ track, double *__restrict__ part,
                                                    float *__restrict__ wrap_lo, float *__restrict__ wrap_hi, int nx, int sx4)
{
    const float len = tv_step_len(dPOCS, gnorm2);
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        f4 xv = x[i], gv = g[i], v;
        v.x = tv_step(xv.x, gv.x, len); v.y = tv_step(xv.y, gv.y, len); v.z = tv_step(xv.z, gv.z, len); v.w = tv_step(xv.w, gv.w, len);
        if (clamp) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        x[i] = v;
        if (wrap_lo) {
            int64_t pix = i / sx4;
            int s = (int)(i - pix * sx4) * 4;
            if (s == 0) wrap_hi[pix] = v.x;
            int d = nx - 1 - s;
            if (d >= 0 && d < 4) wrap_lo[pix] = d == 0 ? v.x : d == 1 ? v.y : d == 2 ? v.z : v.w;
        }
        if (TRACK) {
            f4 d = v - track[i];
            acc += (double)(d.x * d.x) + (double)(d.y * d.y) + (double)(d.z * d.z) + (double)(d.w * d.w);
            track[i] = v;
        }
    }
    if (TRACK) block_accumulate(acc, part);
}

}  // namespace tomo
