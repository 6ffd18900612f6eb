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

// Non-temporal access by experiment bit (make EXTRA=-DTOMO_NT=<mask>); bits that paid are folded into NT_DEFAULT.
#ifndef TOMO_NT
#define TOMO_NT 0
#endif
#ifndef TOMO_NT_OFF   // experiment: bits of NT_DEFAULT switched off
#define TOMO_NT_OFF 0
#endif
constexpr int NT_DEFAULT = 32 | 256;   // 32: k_fp_tile partial sums (SIRT iteration -5 %); 256: k_fgp_fused outputs (-4 %)
template <int BIT, typename T>
__device__ __forceinline__ T nt_ld(const T *p)
{
    if constexpr (((TOMO_NT | NT_DEFAULT) & ~TOMO_NT_OFF & BIT) != 0) return __builtin_nontemporal_load(p);
    else return *p;
}
template <int BIT, typename T>
__device__ __forceinline__ void nt_st(T v, T *p)
{
    if constexpr (((TOMO_NT | NT_DEFAULT) & ~TOMO_NT_OFF & BIT) != 0) {
        __builtin_nontemporal_store(v, p);
    } else *p = v;
}

__global__ void k_finalize(double *__restrict__ part, double *__restrict__ dst)
{
    double v = part[threadIdx.x];  // launched with NPART threads
    part[threadIdx.x] = 0.0;       // leave the buffer ready for the next reduction (no memset launch per reduction)
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

// ---- workgroup -> (ray, slice chunk) map for the ray-driven kernels ---------------------------------------
// Neighbouring rays share pixels, so they should meet in one XCD's L2: workgroups b and b+8 share an XCD
// (round-robin dispatch; speed only, never correctness).  Ray lengths fall off towards the detector edges, so
// an XCD must not own one contiguous block of rays (the edge XCDs would idle): rays are dealt to XCDs in
// groups of RAY_GROUP neighbours instead.
constexpr int RAY_GROUP = 16;

__device__ __forceinline__ void ray_block_map(int bid, int nb, int nrows, int &chunk, int &rowidx)
{
    if ((nrows % (8 * RAY_GROUP)) == 0 && (nb & 7) == 0) {
        int xcd = bid & 7, l = bid >> 3;
        int per = nrows >> 3;               // rays per XCD per chunk
        chunk = l / per;
        int li = l - chunk * per;
        int g = li / RAY_GROUP, w = li - g * RAY_GROUP;
        rowidx = (g * 8 + xcd) * RAY_GROUP + w;
    } else {
        chunk = bid / nrows;
        rowidx = bid - chunk * nrows;
    }
}

// ---- forward projector: ray-driven, one workgroup per (ray, slice chunk) ---------------------------
// g[row][s] = sum_k w_k * x[col_k][s].  The four waves of a workgroup split the ray's entry list; lanes hold
// VEC consecutive slices each.  Entry (col, w) pairs are wave-uniform: fetched by the scalar unit.
// FP_RESID_MUL: r = (b - Ax) * m[row] with m passed in the rowsum argument (Cimmino weights, ctvlib.cpp:215)
enum { FP_STORE = 0, FP_RESID = 1, FP_RESID_NORM = 2, FP_DD = 3, FP_POISSON = 4, FP_RESID_MUL = 5 };

template <int VEC, int MODE>
__global__ __launch_bounds__(256) void k_fp_rows(const float *__restrict__ x, const uint32_t *__restrict__ rptr,
                                                  const uint2 *__restrict__ rent, const float *__restrict__ b,
                                                  const float *__restrict__ rowsum, float *__restrict__ out,
                                                  double *__restrict__ part, int row0, int nrows, int sx)
{
    typedef typename VecOf<VEC>::T V;
    int chunk, rowidx;
    ray_block_map(blockIdx.x, gridDim.x, nrows, chunk, rowidx);
    int row = row0 + rowidx;
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
            } else if (MODE == FP_RESID_MUL) {
                r = (bv - acc) * rowsum[row];
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

// ---- forward projector, narrow-chunk form: LPR lanes x float4 per ray, 64/LPR neighbouring rays per wave ------
// A slice chunk is LPR*4 slices (64 for LPR = 16), so one chunk of a 512^2 volume is 67 MB: launched chunk-major
// over ALL angles, the chunk stays resident in the 256 MB Infinity Cache after the first angle has touched it
// (the wide form's 256-slice chunk is 268 MB and streams from HBM 90 times).  It is also the efficient form for
// narrow slabs (64/128 slices per GPU when a volume is sharded 8 ways).  Each lane group walks its own ray;
// entry (pixel, weight) pairs are fetched with lane-group-uniform vector loads.
template <int LPR, int MODE>
__global__ __launch_bounds__(256) void k_fp_rows_g(const float *__restrict__ x, const uint32_t *__restrict__ rptr,
                                                    const uint2 *__restrict__ rent, const float *__restrict__ b,
                                                    const float *__restrict__ rowsum, float *__restrict__ out,
                                                    double *__restrict__ part, int row0, int nrows, int sx,
                                                    int nchunk)
{
    typedef VecOf<4>::T V;
    constexpr int R = 64 / LPR;                       // rays per wave
    constexpr int U = 8;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = lane / LPR, gl = lane - grp * LPR;
    const int ngw = (nrows + R - 1) / R;              // ray groups per chunk
    int64_t gw = (int64_t)blockIdx.x * 4 + wave;      // chunk-major
    int chunk = (int)(gw / ngw);
    if (chunk >= nchunk) return;                      // grid is rounded up to whole workgroups
    int rowidx = (int)(gw - (int64_t)chunk * ngw) * R + grp;
    bool valid = rowidx < nrows;
    int row = row0 + (valid ? rowidx : 0);
    uint32_t kb = rptr[row], ke = valid ? rptr[row + 1] : kb;
    int off = chunk * (LPR * 4) + gl * 4;
    const float *xp = x + off;
    V acc = vzero<4>();
    // software pipeline: the entry (pixel, weight) loads of trip t+1 are issued before the row loads of trip t are
    // consumed, so a trip costs one memory round trip instead of two dependent ones
    uint2 e[U], en[U];
#pragma unroll
    for (int u = 0; u < U; ++u) e[u] = (kb + u < ke) ? rent[kb + u] : make_uint2(0u, 0u);
    for (uint32_t k = kb; __any(k < ke); k += U) {
        V xv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) xv[u] = *reinterpret_cast<const V *>(xp + (size_t)e[u].x * sx);
#pragma unroll
        for (int u = 0; u < U; ++u) en[u] = (k + U + u < ke) ? rent[k + U + u] : make_uint2(0u, 0u);
#pragma unroll
        for (int u = 0; u < U; ++u) acc += __uint_as_float(e[u].y) * xv[u];
#pragma unroll
        for (int u = 0; u < U; ++u) e[u] = en[u];
    }
    double local = 0.0;
    if (valid) {
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
                r = rs > 0.f ? (bv - acc) / rs : vzero<4>();
            } else if (MODE == FP_RESID_MUL) {
                r = (bv - acc) * rowsum[row];
            } else if (MODE == FP_DD) {
                r = acc;
#pragma unroll
                for (int i = 0; i < 4; ++i) { float d = acc[i] - bv[i]; local += (double)(d * d); }
            } else {
                const float eps = 1e-1f;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float a = acc[i], bb = bv[i];
                    r[i] = (a - bb) / (a + eps);
                    local += (double)(a - bb * logf(a + eps));
                }
            }
            *reinterpret_cast<V *>(out + o) = r;
        }
    }
    if (MODE == FP_DD || MODE == FP_POISSON) {
        local = wave_sum(local);
        if (lane == 0) atomicAdd(&part[blockIdx.x & (NPART - 1)], local);
    }
}

// ---- forward projector, tile-stationary form: all angles from one LDS-resident image tile -----------------------
// The ray-driven forms re-read every pixel once per angle through L1/L2 (nnz x 256 B per 64-slice chunk: 61 GB at
// 512^3 x 90).  Here a workgroup stages a FT_TY x FT_TZ pixel tile x 64 slices (128 KiB) in LDS once and computes, for
// every angle, the partial sums of the rays crossing the tile ("tile segments", host table sysmat.cpp:build_tiles);
// k_fp_tile_reduce then adds a row's segments in ascending tile order (fixed order: bit-reproducible, no float
// atomics) and applies the epilogue.  HBM traffic: the volume once + the partials written and read once
// (~(1.3 (TY+TZ)/2 + 1)/(TY*TZ) of the volume per angle).
// A lane group of 16 lanes x float4 (64 slices) owns one entry stream.  Entries arrive by coalesced vector loads,
// 8 per group and batch (lanes 8-15 hold a second copy); inside a batch lane l takes the entries in the rotated
// order l, l+1, ... (DPP row_ror), so no broadcast is needed: every lane still adds all 8, each to its own slices.
// ds_read_b128 of 256-B pixel images is conflict-free at 256 B/clk whatever the pixels (MI355X_MICROARCH.md, LDS).
constexpr int FT_TY = 32, FT_TZ = 16, FT_PIX = FT_TY * FT_TZ, FT_THREADS = 1024, FT_SLOTS = FT_THREADS / 16, FT_BATCH = 8;
constexpr int FT_LDS_BYTES = (FT_PIX + 1) * 256;         // + one zero pixel for the padding entries

template <int J> __device__ __forceinline__ uint32_t row_ror(uint32_t v)
{
    if (J == 0) return v;
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x120 + J, 0xf, 0xf, true);
}

__global__ __launch_bounds__(FT_THREADS) void k_fp_tile(const float *__restrict__ x, const uint32_t *__restrict__ slot_ptr,
                                                         const uint32_t *__restrict__ slot_seg0, const uint2 *__restrict__ tent,
                                                         float *__restrict__ part, int n, int sx, int tiles_z, int ntiles,
                                                         int chunk0, int ncp)
{
    typedef VecOf<4>::T V;
    extern __shared__ V ft_tile[];                      // [FT_PIX + 1][16]
    // all chunks of a tile run back to back on one XCD (workgroups b and b+8 share an XCD): they read the same table
    const int xcd = blockIdx.x & 7, l = blockIdx.x >> 3;
    const int tile = (l / ncp) * 8 + xcd, c = l % ncp;
    if (tile >= ntiles) return;
    const int ty = tile / tiles_z, tz = tile - ty * tiles_z;
    const int t = threadIdx.x, gl = t & 15;
    {
        V v[FT_PIX / 64];
#pragma unroll
        for (int k = 0; k < FT_PIX / 64; ++k) {
            int lp = (t >> 4) + 64 * k;
            int y = ty * FT_TY + lp / FT_TZ, z = tz * FT_TZ + lp % FT_TZ;
            v[k] = (y < n && z < n) ? nt_ld<16>(reinterpret_cast<const V *>(x + ((size_t)y * n + z) * sx + (size_t)(chunk0 + c) * 64 + gl * 4))
                                    : vzero<4>();
        }
#pragma unroll
        for (int k = 0; k < FT_PIX / 64; ++k) ft_tile[((t >> 4) + 64 * k) * 16 + gl] = v[k];
        if (t < 16) ft_tile[FT_PIX * 16 + t] = vzero<4>();
    }
    __syncthreads();
    const size_t slot = (size_t)tile * FT_SLOTS + (t >> 4);
    const uint32_t b0 = slot_ptr[slot], b1 = slot_ptr[slot + 1];
    uint32_t seg = slot_seg0[slot];
    const uint2 *ep = tent + (size_t)b0 * FT_BATCH + (gl & 7);
    const uint32_t nb = b1 - b0;
    // FT_PF batches in flight, each in its own registers and reloaded in place once consumed (no rotation: a trip
    // then waits only for the oldest load).  The table is padded by FT_PF batches, so the reload needs no bounds test;
    // what lies past the stream's end is replaced by the zero entry.
    const uint32_t zoff = (uint32_t)FT_PIX * 256u;
    const char *base = reinterpret_cast<const char *>(ft_tile) + gl * 16;
    V acc = vzero<4>();
    uint2 e0 = ep[0], e1 = ep[FT_BATCH], e2 = ep[2 * FT_BATCH], e3 = ep[3 * FT_BATCH];
    uint2 e4 = ep[4 * FT_BATCH], e5 = ep[5 * FT_BATCH], e6 = ep[6 * FT_BATCH], e7 = ep[7 * FT_BATCH];
    // Software pipeline: the eight LDS reads of entry batch I+1 are issued before the FMAs of batch I (two register sets; the
    // additions keep their order: bit-identical).  Left to itself the compiler waits for every pair of reads right before its
    // FMAs, so a wave alternates between the LDS and the vector ALU instead of overlapping them (1 workgroup = 2 waves per SIMD).
    // Measured: neither this nor the trimmed address arithmetic (39 instead of 46 vector instructions per batch) nor two
    // accumulation chains moved the kernel (1.00 ms at 512^3 x 90).  PMC: vector ALU 61 % busy, LDS 51 %, no bank conflicts --
    // the two add up instead of overlapping with two waves per SIMD.
#define FT_LD1(XV, OFF, J) XV[J] = *reinterpret_cast<const V *>(base + row_ror<J>(OFF));
#define FT_FM1(XV, WB, J) acc += __uint_as_float(row_ror<J>(WB)) * XV[J];
#define FT_ISSUE(E, I, XV, WB, LAST)                                                                       \
    {                                                                                                     \
        const bool in = b + (I) < nb;                      /* past the stream's end: the zero entry */   \
        const uint32_t off = in ? (E.x & 0x7FFFFFFFu) : zoff;                                             \
        WB = in ? E.y : 0u;                                                                               \
        LAST = in && (E.x >> 31) != 0;                                                                    \
        FT_LD1(XV, off, 0) FT_LD1(XV, off, 1) FT_LD1(XV, off, 2) FT_LD1(XV, off, 3)                       \
        FT_LD1(XV, off, 4) FT_LD1(XV, off, 5) FT_LD1(XV, off, 6) FT_LD1(XV, off, 7)                       \
    }
#define FT_CONSUME(E, I, XV, WB, LAST)                                                                    \
    {                                                                                                     \
        E = epn[(I) * FT_BATCH];                           /* constant offset from the trip's pointer */ \
        FT_FM1(XV, WB, 0) FT_FM1(XV, WB, 1) FT_FM1(XV, WB, 2) FT_FM1(XV, WB, 3)                           \
        FT_FM1(XV, WB, 4) FT_FM1(XV, WB, 5) FT_FM1(XV, WB, 6) FT_FM1(XV, WB, 7)                           \
        if (LAST) {                                                                                       \
            nt_st<32>(acc, reinterpret_cast<V *>(pp));                                                    \
            acc = vzero<4>();                                                                             \
            pp += pstep;                                   /* the group's next segment */                \
        }                                                                                                 \
    }
    constexpr int FT_PF = 8;
    float *pp = part + ((size_t)seg * ncp + c) * 64 + gl * 4;
    const size_t pstep = (size_t)ncp * 64;
    V xa[FT_BATCH], xb[FT_BATCH];
    uint32_t wa, wb;
    bool la, lb;
    uint32_t b = 0;
    FT_ISSUE(e0, 0, xa, wa, la)
    const uint2 *epn = ep + (size_t)FT_PF * FT_BATCH;      // entries of the NEXT trip (reloaded in place once consumed)
    for (; __any(b < nb); b += FT_PF, epn += FT_PF * FT_BATCH) {
        FT_ISSUE(e1, 1, xb, wb, lb) FT_CONSUME(e0, 0, xa, wa, la)
        FT_ISSUE(e2, 2, xa, wa, la) FT_CONSUME(e1, 1, xb, wb, lb)
        FT_ISSUE(e3, 3, xb, wb, lb) FT_CONSUME(e2, 2, xa, wa, la)
        FT_ISSUE(e4, 4, xa, wa, la) FT_CONSUME(e3, 3, xb, wb, lb)
        FT_ISSUE(e5, 5, xb, wb, lb) FT_CONSUME(e4, 4, xa, wa, la)
        FT_ISSUE(e6, 6, xa, wa, la) FT_CONSUME(e5, 5, xb, wb, lb)
        FT_ISSUE(e7, 7, xb, wb, lb) FT_CONSUME(e6, 6, xa, wa, la)
        FT_ISSUE(e0, FT_PF, xa, wa, la)                  /* first batch of the next trip (e0 was reloaded above) */
        FT_CONSUME(e7, 7, xb, wb, lb)
    }
#undef FT_CONSUME
#undef FT_ISSUE
#undef FT_FM1
#undef FT_LD1
}

// ---- forward projector, sheared-strip form: the ray sums stay in registers while a workgroup marches a strip ------------------
// k_fp_tile emits one partial sum per (ray, 32 x 16 tile): 27.5 per ray at 512^2 x 90, 5.3 of the 6.1 GB the projection moves
// (written, then read again by the reduce kernel).  The partial sums are fewer the longer a ray stays with one workgroup, and
// what bounds that is where the running sums live: here they live in REGISTERS (K accumulators of 64 slices per 16-lane
// group: 128 KB per workgroup on top of its 64 KB of LDS).  The angles are split into passes of similar direction (host:
// sysmat.cpp build_fp_strips); a workgroup owns one strip of FS_W pixels across the pass's mean ray direction, sheared along
// it, and marches it tile by tile (FS_H march steps): stage the tile in LDS (each pixel a 256-byte image, conflict-free
// ds_read_b128 as in k_fp_tile), then every 16-lane group walks its entry stream -- slot by slot, the slot's batches of this
// tile into the slot's accumulator -- and a ray that leaves the strip is flushed as ONE partial sum (flag in its last batch).
// A ray owns its slot from the tile where it enters to the tile where it leaves; the four lane groups of a wave hold four
// neighbouring rays of one angle per slot, so the batch counts per (tile, wave, slot) are wave-uniform loop bounds (scalar).
// 5.4 partial sums per ray at 512^2 x 90 in 5 passes (the volume is staged 5 times instead of once).
// Entry sharing inside a batch is k_fp_tile's: 8 entries per group and batch, lane l takes them in the rotated order (DPP).
constexpr int FS_W = 16, FS_H = 16, FS_PIX = FS_W * FS_H, FS_THREADS = 512, FS_WAVES = FS_THREADS / 64, FS_GROUPS = FS_THREADS / 16;
constexpr int FS_RING = 7;                               // entry batches in flight per lane group (LDS ring slots of 64 B)
struct FsItemD { int pass, v0; uint32_t tile0, ntiles, cnt0, g0, work, pad; };
#ifndef FS_WHATIF
#define FS_WHATIF 0
#endif
#ifndef FS_DMA_STAGE
#define FS_DMA_STAGE 1
#endif

// The entry stream of a lane group (8 entries of 8 bytes per batch) comes from L2 at best, and from HBM for whichever of an
// item's chunk workgroups touches a line first; the batch loops have run-time trip counts, so a register ring cannot be kept
// ahead of them (rotating it by moves makes the waitcnt pass wait for the youngest load: measured, the kernel then sits on the
// table's latency -- 1.79 ms against 1.25 ms with the loads taken out).  The stream is therefore fetched by LDS-DMA
// (global_load_lds_dword: 64 lanes x 4 bytes = one batch for each of the wave's four lane groups per instruction, no vector
// register touched) into a per-wave ring of FS_RING slots, FS_RING - 1 batches ahead of the batch being worked on; the entry of
// the NEXT batch is read from the ring (ds_read_b64, issued by inline asm: the waitcnt pass would otherwise drain the DMAs in
// flight before every LDS read it cannot tell apart from their destination -- the ring is an LDS object of its own for the same
// reason, so the tile reads are not held back) while the current batch's pixel images are read.
// (Measured and not kept: the weight's lane rotation folded into the multiply-add -- four v_fmac_f32 with a DPP source per entry
// instead of one DPP move + two packed FMAs -- 1.33 against 1.27 ms per projection at 512^3 x 90.)
template <int K>
__global__ __launch_bounds__(FS_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4)))
void k_fp_strip(const float *__restrict__ x, const FsItemD *__restrict__ items, const int *__restrict__ orient,
                const int *__restrict__ shift, const uint4 *__restrict__ cnt, const uint32_t *__restrict__ gstart,
                const uint32_t *__restrict__ gseg0, const uint2 *__restrict__ ent, float *__restrict__ part, int n, int sx,
                int nitems, int chunk0, int ncp, const float *__restrict__ zero16)
{
    typedef VecOf<4>::T V;
    static_assert(K >= 1 && K <= 16, "slots per lane group");
    __shared__ V fs_tile[(FS_PIX + 1) * 16];            // the tile, 256-byte pixel images, + the zero pixel of the padding entries
    __shared__ uint2 fs_ring[FS_WAVES * FS_RING * 32];  // [wave][slot][lane group][8 entries]: two workgroups per CU (80,128 B each)
    // all chunks of an item run back to back on one XCD (workgroups b and b+8 share an XCD): they read the same tables
    const int xcd = blockIdx.x & 7, l = blockIdx.x >> 3;
    const int it = (l / ncp) * 8 + xcd, c = l % ncp;
    if (it >= nitems) return;
    const FsItemD I = items[it];
    const int t = threadIdx.x, gl = t & 15, g = t >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int o = orient[I.pass];
    const int *sh = shift + (size_t)I.pass * n;
    const float *xc = x + (size_t)(chunk0 + c) * 64 + gl * 4;
    // LDS-DMA source of this lane: 4 bytes of its group's batch (16 lanes x 4 B = the batch's 64 bytes); destination: the wave's slot
    const char *gsrc = reinterpret_cast<const char *>(ent + (size_t)gstart[I.g0 + g] * FT_BATCH) + gl * 4;
    uint2 *ring_w = fs_ring + wave * (FS_RING * 32);
    const uint32_t ring_l = (uint32_t)(size_t)(__attribute__((address_space(3))) uint2 *)(ring_w + ((t >> 4) & 3) * 8 + (gl & 7));
#define FS_DMA(SLOT) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gsrc, \
                                                      (__attribute__((address_space(3))) void *)(ring_w + (SLOT) * 32), 4, 0, 0); gsrc += 64;
#pragma unroll
    for (int b = 0; b < FS_RING; ++b) { FS_DMA(b) }     // batches 0 .. FS_RING-1 (the table is padded past its last stream)
    float *pp = part + ((size_t)gseg0[I.g0 + g] * ncp + c) * 64 + gl * 4;
    const size_t pstep = (size_t)ncp * 64;
    V acc[K];
#pragma unroll
    for (int k = 0; k < K; ++k) acc[k] = vzero<4>();
    const char *base = reinterpret_cast<const char *>(fs_tile) + gl * 16;
    const uint4 *cp = cnt + I.cnt0 + wave;
    if (t < 16) fs_tile[FS_PIX * 16 + t] = vzero<4>();
    uint32_t slot = 0;                                  // wave-uniform: ring slot of the batch whose entry `en` holds
    uint2 en;
    asm volatile("s_waitcnt vmcnt(%1)\n\tds_read_b64 %0, %2" : "=v"(en) : "n"(FS_RING - 1), "v"(ring_l) : "memory");
    for (uint32_t tt = 0; tt < I.ntiles; ++tt) {
        if (tt) __syncthreads();                        // every group is done with the previous tile
#if FS_DMA_STAGE
        {   // the tile by LDS-DMA: a wave-instruction moves the 64-slice images of four neighbouring pixels (one per lane group, 16 bytes
            // per lane) straight into their 1 KB of the tile -- no vector registers, no ds_write pass; pixels outside the image read zeros
            const int u0 = (int)(I.tile0 + tt) * FS_H;
#pragma unroll
            for (int i = 0; i < FS_PIX / FS_GROUPS; ++i) {
                const int q = g + FS_GROUPS * i;
                const int u = u0 + q / FS_W;
                const int vv = I.v0 + sh[min(u, n - 1)] + q % FS_W;
                const bool ok = u < n && (unsigned)vv < (unsigned)n;
                const size_t pix = o ? (size_t)vv * n + u : (size_t)u * n + vv;
                const float *src = ok ? xc + pix * sx : zero16 + gl * 4;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                 (__attribute__((address_space(3))) void *)(fs_tile + (4 * wave + FS_GROUPS * i) * 16), 16, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
#else
        {
            const int u0 = (int)(I.tile0 + tt) * FS_H;
            V v[FS_PIX / FS_GROUPS];
#pragma unroll
            for (int i = 0; i < FS_PIX / FS_GROUPS; ++i) {
                const int q = g + FS_GROUPS * i;
                const int u = u0 + q / FS_W;
                const int vv = I.v0 + sh[min(u, n - 1)] + q % FS_W;
                const bool ok = u < n && (unsigned)vv < (unsigned)n;
                const size_t pix = o ? (size_t)vv * n + u : (size_t)u * n + vv;
#if FS_WHATIF & 2
                v[i] = (ok && tt == 0) ? nt_ld<16>(reinterpret_cast<const V *>(xc + pix * sx)) : vzero<4>();   // timing experiment: one tile staged
#else
                v[i] = ok ? nt_ld<16>(reinterpret_cast<const V *>(xc + pix * sx)) : vzero<4>();
#endif
            }
#pragma unroll
            for (int i = 0; i < FS_PIX / FS_GROUPS; ++i) fs_tile[(g + FS_GROUPS * i) * 16 + gl] = v[i];
        }
#endif
        __syncthreads();
        const uint4 c4 = cp[(size_t)tt * FS_WAVES];
        // One stream unit: the entry of the unit after it is read from the ring while this one's pixel images are read.  FULL: a
        // batch of 8 entries; !FULL: a half batch, 4 entries stored twice in the unit, so the first four rotations meet all of them.
#define FS_LD1(J) xv[J] = *reinterpret_cast<const V *>(base + row_ror<J>(off));
#define FS_FM1(J) acc[k] += __uint_as_float(row_ror<J>(wb)) * xv[J];
#if FS_WHATIF & 1
#define FS_NEXT slot = slot == FS_RING - 1 ? 0u : slot + 1u;
#else
#define FS_NEXT                                                                                           \
        FS_DMA(slot)           /* this unit's slot is free (its entry is in registers): fetch the unit FS_RING ahead into it */ \
        slot = slot == FS_RING - 1 ? 0u : slot + 1u;                                                      \
        /* the NEXT unit's entry: its DMA is the oldest of the FS_RING now in flight */                    \
        asm volatile("s_waitcnt vmcnt(%1)\n\tds_read_b64 %0, %2" : "=v"(en) : "n"(FS_RING - 1), "v"(ring_l + slot * 256u) : "memory");
#endif
#define FS_UNIT(FULL)                                                                                     \
        {                                                                                                 \
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(en) : : "memory");   /* the ring read issued one unit ago */ \
            const uint2 e = en;                                                                           \
            FS_NEXT                                                                                       \
            const uint32_t off = e.x & 0x7FFFFFFFu, wb = e.y;                                             \
            V xv[FT_BATCH];                                                                               \
            FS_LD1(0) FS_LD1(1) FS_LD1(2) FS_LD1(3)                                                       \
            if (FULL) { FS_LD1(4) FS_LD1(5) FS_LD1(6) FS_LD1(7) }                                         \
            FS_FM1(0) FS_FM1(1) FS_FM1(2) FS_FM1(3)                                                       \
            if (FULL) { FS_FM1(4) FS_FM1(5) FS_FM1(6) FS_FM1(7) }                                         \
            if (!(FS_WHATIF & 1) && (e.x >> 31)) {   /* the ray leaves the strip: its sum is this group's next partial sum */ \
                nt_st<32>(acc[k], reinterpret_cast<V *>(pp));                                             \
                acc[k] = vzero<4>();                                                                      \
                pp += pstep;                                                                              \
            }                                                                                             \
        }
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const uint32_t word = k < 4 ? c4.x : k < 8 ? c4.y : k < 12 ? c4.z : c4.w;
            const uint32_t h = __builtin_amdgcn_readfirstlane((word >> (8 * (k & 3))) & 0xFFu);   // half batches of slot k in this tile
            for (uint32_t b = h >> 1; b > 0; --b) FS_UNIT(true)
            if (h & 1) FS_UNIT(false)
        }
#undef FS_UNIT
#undef FS_NEXT
#undef FS_FM1
#undef FS_LD1
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(en) : : "memory");   // nothing of this wave's is in flight when its LDS is released
#undef FS_DMA
}

// row sums of the tile partials + epilogue.  LPR lanes x float4 cover LPR/16 chunks of one row; part = [seg][ncp][64]
template <int LPR, int MODE>
__global__ __launch_bounds__(256) void k_fp_tile_reduce(const float *__restrict__ part, const uint32_t *__restrict__ rsptr,
                                                         const uint32_t *__restrict__ rsidx, const float *__restrict__ b,
                                                         const float *__restrict__ rowsum, float *__restrict__ out,
                                                         double *__restrict__ dpart, int nrows, int sx, int chunk0, int ncp)
{
    typedef VecOf<4>::T V;
    constexpr int R = 64 / LPR, U = 8;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = lane / LPR, gl = lane - grp * LPR;
    const int spans = ncp * 16 / LPR;                    // lane-group spans per row in this pass
    int64_t item = ((int64_t)blockIdx.x * 4 + wave) * R + grp;
    int row = (int)(item / spans), span = (int)(item - (int64_t)row * spans);
    bool valid = row < nrows;
    if (!__any(valid)) return;
    if (!valid) row = 0;
    uint32_t kb = rsptr[row], ke = valid ? rsptr[row + 1] : kb;
    const float *pp = part + (size_t)span * (LPR * 4) + gl * 4;
    V acc = vzero<4>();
    uint32_t sidx[U], sn[U];
#pragma unroll
    for (int u = 0; u < U; ++u) sidx[u] = (kb + u < ke) ? rsidx[kb + u] : 0xFFFFFFFFu;
    for (uint32_t k = kb; __any(k < ke); k += U) {
        V pv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) pv[u] = (sidx[u] != 0xFFFFFFFFu) ? nt_ld<32>(reinterpret_cast<const V *>(pp + (size_t)sidx[u] * ncp * 64)) : vzero<4>();
#pragma unroll
        for (int u = 0; u < U; ++u) sn[u] = (k + U + u < ke) ? rsidx[k + U + u] : 0xFFFFFFFFu;
#pragma unroll
        for (int u = 0; u < U; ++u) acc += pv[u];
#pragma unroll
        for (int u = 0; u < U; ++u) sidx[u] = sn[u];
    }
    double local = 0.0;
    if (valid) {
        size_t o = (size_t)row * sx + (size_t)chunk0 * 64 + (size_t)span * (LPR * 4) + gl * 4;
        if (MODE == FP_STORE) {
            *reinterpret_cast<V *>(out + o) = acc;
        } else {
            V bv = *reinterpret_cast<const V *>(b + o);
            V r;
            if (MODE == FP_RESID) {
                r = bv - acc;
            } else if (MODE == FP_RESID_NORM) {
                float rs = rowsum[row];
                r = rs > 0.f ? (bv - acc) / rs : vzero<4>();
            } else if (MODE == FP_RESID_MUL) {
                r = (bv - acc) * rowsum[row];
            } else if (MODE == FP_DD) {
                r = acc;
#pragma unroll
                for (int i = 0; i < 4; ++i) { float d = acc[i] - bv[i]; local += (double)(d * d); }
            } else {
                const float eps = 1e-1f;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float a = acc[i], bb = bv[i];
                    r[i] = (a - bb) / (a + eps);
                    local += (double)(a - bb * logf(a + eps));
                }
            }
            *reinterpret_cast<V *>(out + o) = r;
        }
    }
    if (MODE == FP_DD || MODE == FP_POISSON) {
        local = wave_sum(local);
        if (lane == 0) atomicAdd(&dpart[blockIdx.x & (NPART - 1)], local);
    }
}

// residual rows from a projection already in hand: the epilogues of k_fp_tile_reduce / k_fp_rows in FP_RESID and FP_RESID_NORM
// mode applied to a stored g = A x (same expressions, so the same bits as projecting again)
template <int MODE>
__global__ __launch_bounds__(256) void k_sino_resid(const float *__restrict__ b, const float *__restrict__ g,
                                                     const float *__restrict__ rowsum, float *__restrict__ out, int64_t n4, int sx4)
{
    typedef VecOf<4>::T V;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        V bv = reinterpret_cast<const V *>(b)[i], acc = reinterpret_cast<const V *>(g)[i], r;
        if (MODE == FP_RESID) {
            r = bv - acc;
        } else {
            float rs = rowsum[i / sx4];
            r = rs > 0.f ? (bv - acc) / rs : vzero<4>();
        }
        reinterpret_cast<V *>(out)[i] = r;
    }
}

// q <- (1 + beta) g - beta p, p <- g : the projection of the Nesterov point y = r + beta (r - r_old) by linearity from A r (g) and
// A r_old (p), and A r saved as the next step's A r_old, in one pass.  g = the model sinogram G is only read: it stays A * recon.
__global__ __launch_bounds__(256) void k_sino_extrapolate(const VecOf<4>::T *__restrict__ g, VecOf<4>::T *__restrict__ p,
                                                          VecOf<4>::T *__restrict__ q, float beta, int64_t n4)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        VecOf<4>::T a = g[i], b = p[i];
        q[i] = a + beta * (a - b);
        p[i] = a;
    }
}

// ---- voxel-driven back-projector, one angle (the SART update) ---------------------------------------
// x[p][s] = max(0, x[p][s] + beta * (w0 r[j0][s] + w1 r[j1][s]) / (w0 + w1))
// cell[p] = {j0, w0, j1, w1}: the (at most two) rays of this angle through pixel p.  r = this angle's
// normalised residual rows (N rows, L2 resident).  One wave owns PPW consecutive pixels of a slice chunk.
struct CellD { uint32_t r0; float w0; uint32_t r1; float w1; };

// TRACK: the same pass also leaves sum (x_new - track)^2 in part[] and overwrites track with x_new -- the step norm and the
// snapshot copy that an ASD-POCS iteration takes after its SART sweep, without two more passes over the slab.
template <int VEC, int PPW, bool TRACK>
__global__ __launch_bounds__(256) void k_bp_angle(float *__restrict__ x, const CellD *__restrict__ cell,
                                                   const float *__restrict__ r, float beta, int npix, int sx,
                                                   int ngroups, int nchunk, float *__restrict__ track,
                                                   double *__restrict__ part, int chunk0)
{
    typedef typename VecOf<VEC>::T V;
    int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int lane = threadIdx.x & 63;
    int gw = blockIdx.x * 4 + wave;  // global wave id
    int chunk = gw / ngroups;
    int grp = gw - chunk * ngroups;
    int p0 = grp * PPW;
    if (p0 >= npix || chunk >= nchunk) return;   // grid is rounded up to whole workgroups
    int off = (chunk0 + chunk) * (64 * VEC) + lane * VEC;   // chunk0: first chunk of the sub-slab this launch covers
    V xv[PPW], r0[PPW], r1[PPW], tk[PPW];
    CellD c[PPW];
#pragma unroll
    for (int q = 0; q < PPW; ++q) {
        int p = min(p0 + q, npix - 1);
        c[q] = cell[p];
        xv[q] = nt_ld<1>(reinterpret_cast<const V *>(x + (size_t)p * sx + off));
        r0[q] = *reinterpret_cast<const V *>(r + (size_t)c[q].r0 * sx + off);
        r1[q] = *reinterpret_cast<const V *>(r + (size_t)c[q].r1 * sx + off);
        if (TRACK) tk[q] = nt_ld<1>(reinterpret_cast<const V *>(track + (size_t)p * sx + off));
    }
    double local = 0.0;
#pragma unroll
    for (int q = 0; q < PPW; ++q) {
        int p = p0 + q;
        if (p < npix) {
            float cs = c[q].w0 + c[q].w1;
            const float inv = 1.0f / (cs > 0.f ? cs : 1.0f);   // cs == 0 means w0 == w1 == 0, so num == 0; the formula of k_sart_tile
            // every rounding written out (mul, fma, mul, fma -- what the float4 code of k_sart_tile compiles to): the compiler's
            // contraction choices differ between vector widths, and a sub-slab of a two-chain sweep may run at another width
            V nv;
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                float num = __fmaf_rn(c[q].w1, velem<VEC>(r1[q], i), __fmul_rn(c[q].w0, velem<VEC>(r0[q], i)));
                float v = __fmaf_rn(beta, __fmul_rn(num, inv), velem<VEC>(xv[q], i));
                vset<VEC>(nv, i, fmaxf(v, 0.f));
            }
            // in place: a pixel's 64*VEC-slice piece whose bits did not change is not stored (see k_sart_tile)
            bool chx = false, cht = false;
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                chx |= __float_as_uint(velem<VEC>(nv, i)) != __float_as_uint(velem<VEC>(xv[q], i));
                if (TRACK) cht |= __float_as_uint(velem<VEC>(nv, i)) != __float_as_uint(velem<VEC>(tk[q], i));
            }
            if (__any(chx)) nt_st<1>(nv, reinterpret_cast<V *>(x + (size_t)p * sx + off));
            if (TRACK) {
#pragma unroll
                for (int i = 0; i < VEC; ++i) { float d = velem<VEC>(nv, i) - velem<VEC>(tk[q], i); local += (double)(d * d); }
                if (__any(cht)) nt_st<1>(nv, reinterpret_cast<V *>(track + (size_t)p * sx + off));
            }
        }
    }
    if (TRACK) {
        local = wave_sum(local);
        if (lane == 0) atomicAdd(&part[blockIdx.x & (NPART - 1)], local);
    }
}

// ---- fused SART step: back-projection of angle "prev" + forward projection of angle "next" ------------
// Ray-driven over the rays of "next": every pixel on the ray first receives the pending voxel update of
// "prev" (same arithmetic as k_bp_angle), the updated value feeds this ray's line integral, and the visit that
// owns the pixel stores it.  The walk lists make the rays of one angle visit every pixel with exactly one
// owner, so x_new is fully written; reads come from x_old only (ping-pong), so the 1-2 rays that share a pixel
// never see a half-updated volume.  Per angle the slab is read once and written once: 8 B/voxel instead of the
// 12 B/voxel of a separate FP + BP pair.
//
// Work decomposition: one workgroup per (ray, chunk) finishes when its longest ray does, and with ~4 workgroups
// per CU there is no second round to even things out (measured: 242 us against 187 us at the streaming rate).
// So a ray's walk list is cut into segments of <= seg_len visits (host: build_segments) and ONE WAVE runs one
// segment: many short equal items, dealt to the XCDs in groups of neighbouring rays.  Each item leaves its
// partial line integral in partial[id][s]; k_resid_finish adds a ray's segments in order and forms the residual.
// A visit-at-a-time loop serialises four dependent memory round trips per pixel; the loop runs U visits per trip
// in phases (entries, cells, 3U row loads, then arithmetic and the owner stores).
// FUSED = false is the plain per-angle forward projection (no pending voxel update, no volume write).
struct SegItemD { uint32_t id, kbeg, kend, pad; };

template <int VEC, int U, bool FUSED>
__global__ __launch_bounds__(64) void k_sart_seg(const float *__restrict__ x_old, float *__restrict__ x_new,
                                                  const SegItemD *__restrict__ exec, int L,
                                                  const uint2 *__restrict__ went, const CellD *__restrict__ cell_prev,
                                                  const float *__restrict__ r_prev, float beta,
                                                  float *__restrict__ partial, int sx)
{
    typedef typename VecOf<VEC>::T V;
    int bid = blockIdx.x;
    int xcd = bid & 7, l = bid >> 3;
    int chunk = l / L;
    int li = l - chunk * L;
    SegItemD it = exec[xcd * L + li];
    uint32_t kb = it.kbeg, ke = it.kend;
    if (kb >= ke) return;  // padding item
    int lane = threadIdx.x;
    int off = chunk * (64 * VEC) + lane * VEC;
    const float *xp = x_old + off;
    V acc = vzero<VEC>();
    for (uint32_t k = kb; k < ke; k += U) {
        uint2 e[U];
        V xv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) e[u] = went[min(k + u, ke - 1)];
        if (FUSED) {
            const float *rp = r_prev + off;
            float *xo = x_new + off;
            CellD c[U];
            V a0[U], a1[U];
#pragma unroll
            for (int u = 0; u < U; ++u) c[u] = cell_prev[e[u].x & 0x7fffffffu];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                xv[u] = *reinterpret_cast<const V *>(xp + (size_t)(e[u].x & 0x7fffffffu) * sx);
                a0[u] = *reinterpret_cast<const V *>(rp + (size_t)c[u].r0 * sx);
                a1[u] = *reinterpret_cast<const V *>(rp + (size_t)c[u].r1 * sx);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) asm volatile("" : "+v"(xv[u]), "+v"(a0[u]), "+v"(a1[u]));
#pragma unroll
            for (int u = 0; u < U; ++u) {
                bool live = k + u < ke;
                float cs = c[u].w0 + c[u].w1;
                V num = c[u].w0 * a0[u];
                num += c[u].w1 * a1[u];
                V upd = num * (1.0f / (cs > 0.f ? cs : 1.0f));
                V nv = xv[u] + beta * upd;
#pragma unroll
                for (int i = 0; i < VEC; ++i) vset<VEC>(nv, i, fmaxf(velem<VEC>(nv, i), 0.f));
                float w = live ? __uint_as_float(e[u].y) : 0.f;
                acc += w * nv;
                if (live && (e[u].x & 0x80000000u))
                    *reinterpret_cast<V *>(xo + (size_t)(e[u].x & 0x7fffffffu) * sx) = nv;
            }
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u) xv[u] = *reinterpret_cast<const V *>(xp + (size_t)(e[u].x & 0x7fffffffu) * sx);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                float w = (k + u < ke) ? __uint_as_float(e[u].y) : 0.f;
                acc += w * xv[u];
            }
        }
    }
    *reinterpret_cast<V *>(partial + (size_t)it.id * sx + off) = acc;
}

// r[row][s] = (b - sum of the row's partials) / rowsum   (0 where rowsum == 0).  A row's partials have consecutive ids.
// One workgroup per (row, chunk): its four waves each add a quarter of the list (the tile form leaves ~N/11 partials
// per ray), the quarters are combined in fixed order through LDS.
constexpr int RF_U = 12;
// SUM: r_out = the plain row sum (the forward projection itself; b and rowsum unused) -- the chained ART sweep
template <int VEC, bool SUM = false>
__global__ __launch_bounds__(256) void k_resid_finish(const float *__restrict__ partial,
                                                       const uint32_t *__restrict__ row_first,
                                                       const uint32_t *__restrict__ row_nseg,
                                                       const float *__restrict__ b, const float *__restrict__ rowsum,
                                                       float *__restrict__ r_out, int row0, int nrows, int nchunk, int sx,
                                                       int chunk0)
{
    typedef typename VecOf<VEC>::T V;
    __shared__ V red[3][64];
    int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int lane = threadIdx.x & 63;
    int chunk = blockIdx.x / nrows;
    int row = row0 + (blockIdx.x - chunk * nrows);
    int off = (chunk0 + chunk) * (64 * VEC) + lane * VEC;
    uint32_t first = row_first[row], ns = row_nseg[row];
    // the measurement row and the row sum do not depend on the partials: in flight from the start (wave 0 uses them)
    size_t o = (size_t)row * sx + off;
    V bv = vzero<VEC>();
    float rs = 0.f;
    if (wave == 0 && !SUM) { bv = *reinterpret_cast<const V *>(b + o); rs = rowsum[row]; }
    uint32_t q = (ns + 3u) >> 2;
    uint32_t sb = min(wave * q, ns), se = min(sb + q, ns);
    V acc = vzero<VEC>();
    const float *pp = partial + (size_t)first * sx + off;
    for (uint32_t s = sb; s < se; s += RF_U) {        // RF_U independent loads per trip (one trip at the tile form's ~N/11
        V t[RF_U];                                    // partials per ray), summed in segment order
#pragma unroll
        for (int u = 0; u < RF_U; ++u) t[u] = (s + u < se) ? nt_ld<2>(reinterpret_cast<const V *>(pp + (size_t)(s + u) * sx)) : vzero<VEC>();
#pragma unroll
        for (int u = 0; u < RF_U; ++u) acc += t[u];
    }
    if (wave > 0) red[wave - 1][lane] = acc;
    __syncthreads();
    if (wave != 0) return;
    acc = ((acc + red[0][lane]) + red[1][lane]) + red[2][lane];
    V r = SUM ? acc : (rs > 0.f ? (bv - acc) / rs : vzero<VEC>());
    *reinterpret_cast<V *>(r_out + o) = r;
}

// ---- voxel-driven back-projector, all angles (SIRT / Landweber / plain A^T / Poisson) ----------------
// acc[p][s] = sum_i (w0 r[i*N+j0][s] + w1 r[i*N+j1][s])      rows in ascending order, like Eigen's A^T*v
// epilogue:  v = alpha*x + beta * (colsum ? acc/colsum[p] : acc);  x = clamp ? max(0, v) : v
// alpha * x + beta * a of the all-angle back-projectors' epilogues, in ONE arithmetic for every form and vector width: the product
// beta * a rounded, then one FMA (left to the contraction pass, the scalar and the vector forms of "alpha * x + beta * a" came out
// as different FMAs: 1-ulp differences between k_bp_all<1> and the others).
template <typename V>
__device__ __forceinline__ V bp_axpby(float alpha, V x, float beta, V a)
{
#pragma clang fp contract(off)
    V t = beta * a;
    return __builtin_elementwise_fma((V)alpha, x, t);
}

template <typename V>
__device__ __forceinline__ V bp_fma(float w, V a, V acc) { return __builtin_elementwise_fma((V)w, a, acc); }
template <>
__device__ __forceinline__ float bp_fma<float>(float w, float a, float acc) { return __builtin_fmaf(w, a, acc); }

template <int VEC, int PPW>
__global__ __launch_bounds__(256) void k_bp_all(float *__restrict__ x, const CellD *__restrict__ cell,
                                                 const float *__restrict__ r, const float *__restrict__ colsum,
                                                 float alpha, float beta, int clamp, int nproj, int nray, int npix,
                                                 int sx, int ngroups, int nchunk)
{
    typedef typename VecOf<VEC>::T V;
    int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int lane = threadIdx.x & 63;
    int gw = blockIdx.x * 4 + wave;
    int chunk = gw / ngroups;
    int grp = gw - chunk * ngroups;
    int p0 = grp * PPW;
    if (p0 >= npix || chunk >= nchunk) return;   // grid is rounded up to whole workgroups
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
            // two FMAs, written out (round 5): left to the compiler, the one-float-per-lane build packed the four pixels' products
            // and sums of one of the two statements into v_pk_mul_f32 + v_pk_add_f32 (two roundings) where every other build and
            // every other back-projector contracts to an FMA -- the 1-ulp difference of k_bp_all<1> that round 4 could not place
            acc[q] = bp_fma(c.w0, a0, acc[q]);
            acc[q] = bp_fma(c.w1, a1, acc[q]);
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
            if (alpha != 0.f) nv = bp_axpby(alpha, *reinterpret_cast<const V *>(xp), beta, a);
            if (clamp) {
#pragma unroll
                for (int i = 0; i < VEC; ++i) vset<VEC>(nv, i, fmaxf(velem<VEC>(nv, i), 0.f));
            }
            *reinterpret_cast<V *>(xp) = nv;
        }
    }
}

// ---- fused SART step, tile form: BP(prev) + FP(next) on image tiles streamed through LDS -------------------------
// k_sart_seg walks rays: every pixel is a 1-KiB gather along a ray, and the achieved HBM rate stays ~15 % under that
// of a streaming pass (k_bp_angle).  Here a workgroup owns a ST_T x ST_T pixel tile x 64 slices: it streams the tile
// in (coalesced), applies the pending voxel update of angle "prev" pixel-driven from the tile's window of residual rows
// (staged in LDS; same arithmetic as k_bp_angle, bit-identical), streams the tile out, keeps the updated tile as an LDS
// image and forms, for angle "next", the partial sums of the ray segments inside the tile from that image (one segment
// per 16-lane group, entry batches shared by DPP rotation as in k_fp_tile).  k_resid_finish adds a ray's partials
// (consecutive ids, ascending tile) and forms the residual.  Per angle: the slab read once and written once, plus ~9 % for the partials.
// Two workgroups per CU (75 KB LDS each) overlap one's streaming with the other's LDS phase.  In-place is safe: a
// workgroup reads and writes only its own tile.
// Tile shape: ST_TY rows x ST_TZ columns, 8 pixels per 16-lane group.  Measured at 512^3 x 90 (MI355X, round 2):
// 16 x 16 tiles, 512 threads, 75 KB of LDS (two workgroups per CU): 218-224 us per fused step; 16 x 8 tiles (tall: rays
// of a -70..70 degree series run closer to the y axis), 256 threads, 40 KB (FOUR workgroups per CU): 231 us, the
// per-angle FP 160 instead of 144 us -- more independent phases per CU did not pay for 40 % more partial sums (again with the
// non-temporal tile accesses below: 204.7 against 200.3 us per angle).  The kernel
// is not HBM-bound either: a 128-slice slab that sits in the 256 MB Infinity Cache runs at the same rate per byte.
constexpr int ST_TY = 16, ST_TZ = 16, ST_PIX = ST_TY * ST_TZ, ST_THREADS = 512, ST_MAXR = 26, ST_MAXSEG = 32;
constexpr int ST_NG = ST_THREADS / 16, ST_SPG = ST_MAXSEG / ST_NG;   // 16-lane groups; ray segments per group
constexpr int ST_MAXB = (ST_TY + ST_TZ - 1 + 7) / 8;                 // entry batches of the longest segment (TY + TZ - 1 pixels)
static_assert(ST_PIX == ST_NG * 8, "a group owns 8 pixels");
constexpr int ST_LDS_V = (ST_PIX + 1) * 16 + (ST_MAXR + 1) * 16 + ST_PIX;     // image + zero pixel, window + zero row, cells

// Voxel update: num * (1 / colsum) -- one IEEE division per pixel instead of four (num / colsum per component): -3.7 % per
// launch (round 2; the kernel is not HBM-bound, see above).  Cells with host-normalised weights (two FMAs per component, no
// division at all) measured the same 215 us, so the cells keep the raw weights and the formula of k_bp_angle / k_sart_seg:
// the three forms are bit-identical.
#ifdef TOMO_WHATIF   // measurement builds only (make EXTRA=-DTOMO_WHATIF): switch parts of k_sart_tile off, results are WRONG
__device__ int g_sart_whatif = 0;   // 1 no x stores, 2 no BP arithmetic, 4 no FP phase, 8 no window / cell staging, 16 no tile loads
#define ST_WI(bit) (wi_ & (bit))
#else
#define ST_WI(bit) 0
#endif

// ---- cooperative residual rows (COOP) -----------------------------------------------------------------------------
// The chain "tile step; k_resid_finish; tile step; ..." pays one short kernel and two launch boundaries per angle for the
// residual rows (12 + 4 us of 230 at 512^3, 4 + 4 of 37 on a 64-slice slab of a multi-GPU run).  In the COOP form the tile
// step of link k first turns the partial sums that link k-1 left (they are complete: kernel boundary) into the residual
// rows of angle "prev" itself: the first `nred` workgroups of the grid -- the ones that start first -- each take a share
// of the (row, 64-slice chunk) items, one wave per item with its four 16-lane quarters in the role of k_resid_finish's four
// waves (same split, same order of additions, same division: bit-identical rows).  Rows are published write-through
// (sc1 stores, s_waitcnt vmcnt(0), then one agent-scope flag store per row and chunk carrying this launch's epoch); a
// tile workgroup polls the flags of its window rows (one wave, sc1 loads) and stages the rows with sc1 loads.  The reducer
// duty comes before any wait, so nothing can deadlock whatever the dispatch order or residency; a workgroup whose rows are
// not flagged after `spin` polls computes them itself from the partials (same arithmetic, into LDS only).
// Measured: no gain (see "sart_coop" in tomo_engine.hip) -- kept as an option with its tests (tests/test_gpu_sart_coop.py).
struct StCoop {
    const float *p_read;            // partial sums of angle "prev" (written by the previous link)
    const uint32_t *row_first, *row_nseg;   // of angle prev
    const float *b;                 // measured rows of angle prev
    const float *rowsum;            // of angle prev
    float *r_out;                   // residual rows of angle prev (= r_prev of the tile step)
    uint32_t *flags;                // [row][chunk of the whole slab]
    uint32_t epoch;
    int nred, nitems, nchunk_all, spin;
};

__device__ __forceinline__ void st_store_sc1(float *p, VecOf<4>::T v)
{
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");   // s_nop: see st_xstore
}
__device__ __forceinline__ VecOf<4>::T st_load_sc1(const float *base, uint32_t byte_off)
{
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base), 0, 0x7fffffff, 0x00020000);
    return __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off, 0, 16);   // aux 16 = sc1
}

// One wave: the residual row `row` (of angle prev) for 64-slice chunk cc.  Returns the row in the 16 lanes of quarter 0.
// U = loads in flight per trip (the additions run in list order whatever U is).
template <int U>
__device__ __forceinline__ VecOf<4>::T st_resid_row(const StCoop &co, int row, int cc, int sx)
{
    typedef VecOf<4>::T V;
    const int ln = threadIdx.x & 63, qd = ln >> 4, l16 = ln & 15;
    const uint32_t first = co.row_first[row], ns = co.row_nseg[row];
    const int off = cc * 64 + l16 * 4;
    const size_t o = (size_t)row * sx + off;
    V bv = *reinterpret_cast<const V *>(co.b + o);
    const float rs = co.rowsum[row];
    const uint32_t q = (ns + 3u) >> 2;
    const uint32_t sb = min((uint32_t)qd * q, ns), se = min(sb + q, ns);
    V acc = vzero<4>();
    const float *pp = co.p_read + (size_t)first * sx + off;
    for (uint32_t s = sb; s < se; s += U) {
        V t[U];
#pragma unroll
        for (int u = 0; u < U; ++u) t[u] = (s + u < se) ? *reinterpret_cast<const V *>(pp + (size_t)(s + u) * sx) : vzero<4>();
#pragma unroll
        for (int u = 0; u < U; ++u) acc += t[u];
    }
    V a1, a2, a3;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        a1[c] = __shfl(acc[c], l16 + 16, 64);
        a2[c] = __shfl(acc[c], l16 + 32, 64);
        a3[c] = __shfl(acc[c], l16 + 48, 64);
    }
    acc = ((acc + a1) + a2) + a3;
    return rs > 0.f ? (bv - acc) / rs : vzero<4>();
}

// The tile's voxels are read once and written once per launch: non-temporal loads and stores keep them from displacing the
// residual rows, partial sums and tables the launch (and k_resid_finish after it) re-reads from L2.  Measured (512^3 x 90, a
// sweep incl. k_resid_finish, same box): 222 us per angle plain, 211 nt loads only, 226 nt stores only, 202.5 both.
// NT = false (slabs that fit the 256 MB Infinity Cache: the thin slabs of a multi-GPU run) keeps plain accesses -- there the
// next angle's launch finds the slab cached and the streaming forms lose (64 slices: 33.5 against 30.8 us per step, 128: 57.6
// against 53.2; 256 slices: 100.1 against 104).
// Stores: nt 202.8 us per angle, sc1 201.5, sc0 sc1 201.5, sc1 nt 199.5, sc0 sc1 nt 199.3 (write-through and not kept in L2);
// loads: nt 200.5, sc1 205, nt sc1 200.6 (same run).  The store is inline asm (no builtin carries sc1 nt): 16 bytes per lane,
// whole 256-byte pieces per 16-lane group.  (An inline-asm store is invisible to the compiler's hazard recogniser: a 128-bit
// store needs a wait state before its data registers are written again -- the s_nop; without it a k_fp_tile trial of this
// store lost data.)
template <bool NT>
__device__ __forceinline__ VecOf<4>::T st_xload(const VecOf<4>::T *p)
{
    if constexpr (NT) return __builtin_nontemporal_load(p);
    else return *p;
}
template <bool NT>
__device__ __forceinline__ void st_xstore(VecOf<4>::T v, VecOf<4>::T *p)
{
    if constexpr (NT) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
    else *p = v;
}
// ART = true: the pending voxel update is the Kaczmarz one of k_bp_art (x += (w a) beta per ray in ascending ray order, no
// normalisation, no clamp: ctvlib.cpp:137-155 keeps the clamp for the end of the sweep) -- the chained ART sweep then runs as
// the same fused steps as SART, with k_art_chain in the place of the residual normalisation.
template <bool FUSED, bool COOP = false, bool NT = true, bool ART = false>
__global__ __launch_bounds__(ST_THREADS) void k_sart_tile(const float *x_old, float *x_new,
                                                           const uint4 *__restrict__ cells, const uint32_t *__restrict__ wins,
                                                           const float *__restrict__ r_prev, float beta,
                                                           const uint2 *__restrict__ segs, const uint32_t *__restrict__ segid,
                                                           const uint2 *__restrict__ ent, float *__restrict__ partial,
                                                           int n, int sx, int tiles_z, int ntiles, int nchunk, int chunk0,
                                                           int skip_same, StCoop co)
{
    typedef VecOf<4>::T V;
#ifdef TOMO_WHATIF
    const int wi_ = g_sart_whatif;
#endif
    static_assert(!COOP || FUSED, "the cooperative residual rows feed the voxel update");
    if (COOP && (int)blockIdx.x < co.nred) {
        // reducer duty (before anything this workgroup could wait for): items (row, chunk), one wave each
        const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
        for (int it = blockIdx.x * (ST_THREADS / 64) + wv; it < co.nitems; it += co.nred * (ST_THREADS / 64)) {
            const int row = it / nchunk, cc = chunk0 + it - row * nchunk;
            V rr = st_resid_row<RF_U>(co, row, cc, sx);
            if (ln < 16) st_store_sc1(co.r_out + (size_t)row * sx + cc * 64 + ln * 4, rr);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (ln == 0) __hip_atomic_store(co.flags + (size_t)row * co.nchunk_all + cc, co.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    extern __shared__ V st_lds[];                       // ST_LDS_V float4 (dynamic)
    V *img = st_lds, *win = st_lds + (ST_PIX + 1) * 16;
    uint4 *cel = reinterpret_cast<uint4 *>(st_lds + (ST_PIX + 1) * 16 + (ST_MAXR + 1) * 16);
    // the chunks of a tile run back to back on one XCD (workgroups b and b+8 share an XCD): they stream the same
    // pixel lines and read the same tables.  (A persistent form with the next tile prefetched into registers while
    // the current one is in its LDS phases measured 3 % slower.)
    const int xcd = blockIdx.x & 7, l = blockIdx.x >> 3;
    const int tile = (l / nchunk) * 8 + xcd, c = chunk0 + l % nchunk;   // chunk0: first 64-slice chunk of this launch's sub-slab
    if (tile >= ntiles) return;
    const int ty = tile / tiles_z, tz = tile - ty * tiles_z;
    const int t = threadIdx.x, gl = t & 15, g = t >> 4;
    const int off = c * 64 + gl * 4;
    // the group's 8 pixels: local indices g*8 .. g*8+7 (y-major inside the tile)
    const int y = ty * ST_TY + (g * 8) / ST_TZ, z0 = tz * ST_TZ + (g * 8) % ST_TZ;
    // the group's ray segments of "next" and all their entry batches are fetched first, so the forward-projection phase
    // at the end touches LDS only (its two dependent loads cost 10 us per launch when issued there)
    // COOP: the first look at the window rows' flags is issued ahead of the tile loads (loads return in order: issued
    // behind them it would come back only after the whole tile, and the rows could be requested only then)
    uint32_t wflag = 0, wbase = 0, wcnt = 0;
    const uint32_t *wfp = nullptr;
    if (FUSED && COOP) {
        const uint32_t w = wins[tile];
        wbase = w & 0xFFFFu; wcnt = w >> 16;
        if (t < 64) {
            wfp = co.flags + (size_t)(wbase + min((uint32_t)t, wcnt ? wcnt - 1 : 0u)) * co.nchunk_all + c;
            wflag = ((uint32_t)t < wcnt) ? __hip_atomic_load(wfp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : co.epoch;
        }
    }
    uint2 sd[ST_SPG];
    uint32_t pid[ST_SPG];
#pragma unroll
    for (int q = 0; q < ST_SPG; ++q) {
        sd[q] = segs[(size_t)tile * ST_MAXSEG + g + q * ST_NG];
        pid[q] = segid[(size_t)tile * ST_MAXSEG + g + q * ST_NG];
    }
    V xv[8];
#pragma unroll
    for (int J = 0; J < 8; ++J)
        xv[J] = (y < n && z0 + J < n && !ST_WI(16)) ? st_xload<NT>(reinterpret_cast<const V *>(x_old + ((size_t)y * n + z0 + J) * sx + off)) : vzero<4>();
    uint2 eb[ST_SPG][ST_MAXB];
#pragma unroll
    for (int q = 0; q < ST_SPG; ++q) {
        const uint2 *ep = ent + (size_t)sd[q].x * FT_BATCH + (gl & 7);
#pragma unroll
        for (int b = 0; b < ST_MAXB; ++b) eb[q][b] = ((uint32_t)b < sd[q].y) ? ep[(size_t)b * FT_BATCH] : make_uint2((uint32_t)ST_PIX * 256u, 0u);
    }
    if (FUSED && COOP) {
        __shared__ int st_rows_ready;
        if (t < 64) {   // one wave polls the flags of the window's rows
            bool ok;
            int spins = 0;
            for (;;) {
                ok = __all(wflag == co.epoch);
                if (ok || ++spins > co.spin) break;
                __builtin_amdgcn_s_sleep(2);
                wflag = ((uint32_t)t < wcnt) ? __hip_atomic_load(wfp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : co.epoch;
            }
            if (co.spin < 0) ok = false;                 // tests: every workgroup takes the do-it-yourself path
            if (t == 0) st_rows_ready = ok;
        }
        if (t < ST_PIX) cel[t] = cells[(size_t)tile * ST_PIX + t];
        __syncthreads();
        if (st_rows_ready) {
            for (int i = t; i < (ST_MAXR + 1) * 16; i += ST_THREADS) {
                int j = i >> 4;
                win[i] = ((uint32_t)j < wcnt) ? st_load_sc1(r_prev, (uint32_t)((((size_t)wbase + j) * sx + c * 64 + (i & 15) * 4) * sizeof(float))) : vzero<4>();
            }
        } else {        // rows not published in time (reducer workgroups not resident yet): this workgroup's own copy
            const int wv = t >> 6, ln = t & 63;
            for (int j = wv; j < ST_MAXR + 1; j += ST_THREADS / 64) {
                V rr = ((uint32_t)j < wcnt) ? st_resid_row<2>(co, (int)wbase + j, c, sx) : vzero<4>();
                if (ln < 16) win[j * 16 + ln] = rr;
            }
        }
    } else if (FUSED && !ST_WI(8)) {
        uint32_t w = wins[tile];
        for (int i = t; i < (ST_MAXR + 1) * 16; i += ST_THREADS) {
            int j = i >> 4;
            win[i] = ((uint32_t)j < (w >> 16)) ? *reinterpret_cast<const V *>(r_prev + ((size_t)(w & 0xFFFFu) + j) * sx + c * 64 + (i & 15) * 4) : vzero<4>();
        }
        if (t < ST_PIX) cel[t] = cells[(size_t)tile * ST_PIX + t];
    }
    if (t < 16) img[ST_PIX * 16 + t] = vzero<4>();
    if (FUSED && !ST_WI(2)) {
        __syncthreads();
        const char *wb = reinterpret_cast<const char *>(win) + gl * 16;
#pragma unroll
        for (int J = 0; J < 8; ++J) {
            uint4 ce = cel[g * 8 + J];
            V a0 = *reinterpret_cast<const V *>(wb + ce.x), a1 = *reinterpret_cast<const V *>(wb + ce.z);
            float w0 = __uint_as_float(ce.y), w1 = __uint_as_float(ce.w);
            const V ov = xv[J];
            V nv;
            if constexpr (ART) {
                // k_bp_art's expression: ascending ray order (window offsets order like ray indices), each term (w a) beta,
                // zero weights skipped
                uint32_t oa = ce.x, ob = ce.z;
                if (w1 != 0.f && (w0 == 0.f || ob < oa)) { V tv = a0; a0 = a1; a1 = tv; float tw = w0; w0 = w1; w1 = tw; }
                nv = ov;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float v = nv[i];
                    if (w0 != 0.f) v = __fadd_rn(v, __fmul_rn(__fmul_rn(w0, a0[i]), beta));
                    if (w1 != 0.f) v = __fadd_rn(v, __fmul_rn(__fmul_rn(w1, a1[i]), beta));
                    nv[i] = v;
                }
            } else {
                float cs = w0 + w1;
                V num = w0 * a0;
                num += w1 * a1;
                V upd = num * (1.0f / (cs > 0.f ? cs : 1.0f));
                nv = ov + beta * upd;
                nv[0] = fmaxf(nv[0], 0.f); nv[1] = fmaxf(nv[1], 0.f); nv[2] = fmaxf(nv[2], 0.f); nv[3] = fmaxf(nv[3], 0.f);
            }
            xv[J] = nv;
            // In place, a 256-byte piece (one pixel x 64 slices = one 16-lane group) whose bits did not change needs no store:
            // voxels held at zero by the positivity clamp, pixels no ray of this angle crosses, rays with a zero residual.
            bool wr = true;
            if (skip_same) {
                const bool mine = ((__float_as_uint(nv[0]) ^ __float_as_uint(ov[0])) | (__float_as_uint(nv[1]) ^ __float_as_uint(ov[1])) |
                                   (__float_as_uint(nv[2]) ^ __float_as_uint(ov[2])) | (__float_as_uint(nv[3]) ^ __float_as_uint(ov[3]))) != 0u;
                wr = ((__ballot(mine) >> (t & 48)) & 0xFFFFull) != 0;
            }
            if (y < n && z0 + J < n && wr && !ST_WI(1)) st_xstore<NT>(nv, reinterpret_cast<V *>(x_new + ((size_t)y * n + z0 + J) * sx + off));
        }
    }
    if (ST_WI(4)) {
        if (FUSED && ST_WI(2) && !ST_WI(1)) {   // copy-through when the update is off but the stores are on
#pragma unroll
            for (int J = 0; J < 8; ++J) if (y < n && z0 + J < n) *reinterpret_cast<V *>(x_new + ((size_t)y * n + z0 + J) * sx + off) = xv[J];
        }
        return;
    }
#pragma unroll
    for (int J = 0; J < 8; ++J) img[(g * 8 + J) * 16 + gl] = xv[J];
    __syncthreads();
    // forward projection of "next": group g owns the tile's ray segments g, g + ST_NG, ...
    const char *ib = reinterpret_cast<const char *>(img) + gl * 16;
#define ST_LOAD(J) q_[J] = *reinterpret_cast<const V *>(ib + row_ror<J>(e.x));
#define ST_FMA(J) acc += __uint_as_float(row_ror<J>(e.y)) * q_[J];
#pragma unroll
    for (int q = 0; q < ST_SPG; ++q) {
        if (sd[q].y == 0) continue;                       // uniform inside a 16-lane DPP row
        V acc = vzero<4>();
#pragma unroll
        for (int b = 0; b < ST_MAXB; ++b) {
            if ((uint32_t)b < sd[q].y) {
                uint2 e = eb[q][b];
                V q_[FT_BATCH];
                ST_LOAD(0) ST_LOAD(1) ST_LOAD(2) ST_LOAD(3) ST_LOAD(4) ST_LOAD(5) ST_LOAD(6) ST_LOAD(7)
                ST_FMA(0) ST_FMA(1) ST_FMA(2) ST_FMA(3) ST_FMA(4) ST_FMA(5) ST_FMA(6) ST_FMA(7)
            }
        }
        if (sd[q].y > ST_MAXB) {                          // longer segments (only a user matrix can have them)
            const uint2 *ep = ent + (size_t)sd[q].x * FT_BATCH + (gl & 7);
            for (uint32_t b = ST_MAXB; b < sd[q].y; ++b) {
                uint2 e = ep[(size_t)b * FT_BATCH];
#pragma unroll
                for (int J = 0; J < 8; ++J) {
                    // generic lane exchange (__shfl) instead of the compile-time DPP rotation: rare path
                    uint32_t ox = (uint32_t)__shfl((int)e.x, (gl + J) & 7, 16), wy = (uint32_t)__shfl((int)e.y, (gl + J) & 7, 16);
                    acc += __uint_as_float(wy) * *reinterpret_cast<const V *>(ib + ox);
                }
            }
        }
        nt_st<128>(acc, reinterpret_cast<V *>(partial + (size_t)pid[q] * sx + off));
    }
#undef ST_FMA
#undef ST_LOAD
}

// ---- back-projector, all angles, tile-stationary form ------------------------------------------------------------
// k_bp_all gathers 2 x 256 B per pixel, angle and 64-slice chunk from L2 (96 GB at 512^3 x 90).  Here a workgroup owns
// a FT_TY x FT_TZ pixel tile x 64 slices, keeps the 512 x 64 sums in registers (8 pixels per 16-lane group) and stages,
// FB_A angles at a time and double-buffered, the window of residual rows that cross the tile (<= FB_MAXR per angle)
// in LDS; the two row reads per pixel and angle then come from LDS.  Cells {row offset, weight} x 2 arrive by
// coalesced loads, 8 pixels per group and angle, and are shared by DPP row rotation as in k_fp_tile: at step J lane l
// works on pixel (l + J) mod 8 of its group, always into acc[J], so the sums never move between lanes.
// Same two FMAs per pixel and angle in the same order as k_bp_all: results are bit-identical.
constexpr int FB_A = 4, FB_MAXR = 40, FB_BUF = (FB_A * FB_MAXR + 1) * 256;
constexpr int FB_LDS_BYTES = FT_PIX * 256;              // two stage buffers (82 KB); the epilogue reuses it as a 128 KiB tile image
static_assert(2 * FB_BUF <= FB_LDS_BYTES, "stage buffers must fit the tile image");
constexpr int FB_MAX_PROJ = 4096;                       // ray windows of all angles sit in LDS (4 B each)
constexpr int FB_SLOTS = FB_A * FB_MAXR * 16, FB_Q = (FB_SLOTS + FT_THREADS - 1) / FT_THREADS;

__global__ __launch_bounds__(FT_THREADS) void k_bp_tile(float *__restrict__ x, const uint4 *__restrict__ tcell,
                                                         const uint32_t *__restrict__ win, const float *__restrict__ r,
                                                         const float *__restrict__ colsum, float alpha, float beta, int clamp,
                                                         int nproj, int n, int sx, int tiles_z, int ntiles, int nchunk)
{
    typedef VecOf<4>::T V;
    extern __shared__ V fb_lds[];
    const int xcd = blockIdx.x & 7, l = blockIdx.x >> 3;
    const int tile = (l / nchunk) * 8 + xcd, c = l % nchunk;
    if (tile >= ntiles) return;
    const int ty = tile / tiles_z, tz = tile - ty * tiles_z;
    const int t = threadIdx.x, gl = t & 15, g = t >> 4;
    const uint32_t *wn = win + (size_t)tile * nproj;
    const float *rc = r + (size_t)c * 64;
    const int nstage = (nproj + FB_A - 1) / FB_A;
    if (t < 16) { fb_lds[FB_A * FB_MAXR * 16 + t] = vzero<4>(); fb_lds[FB_BUF / 16 + FB_A * FB_MAXR * 16 + t] = vzero<4>(); }
    // the tile's ray windows, all angles, behind the stage buffers: the staging loads then depend on an LDS read only
    uint32_t *lwin = reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(fb_lds) + FB_LDS_BYTES);
    for (int i = t; i < nproj; i += FT_THREADS) lwin[i] = wn[i];
    __syncthreads();
    V sreg[FB_Q];
    bool sval[FB_Q];
#define FB_STAGE_LOAD(S)                                                                                  \
    _Pragma("unroll") for (int q = 0; q < FB_Q; ++q) {                                                    \
        int f = t + FT_THREADS * q;                                                                       \
        int a = f / (FB_MAXR * 16), j = (f - a * (FB_MAXR * 16)) >> 4;                                    \
        int i = (S) * FB_A + a;                                                                           \
        sval[q] = false;                                                                                  \
        if (f < FB_SLOTS && i < nproj) {                                                                  \
            uint32_t w = lwin[i];                                                                         \
            if ((uint32_t)j < (w >> 16)) {                                                                \
                sval[q] = true;                                                                           \
                sreg[q] = *reinterpret_cast<const V *>(rc + ((size_t)i * n + (w & 0xFFFFu) + j) * sx + (f & 15) * 4); \
            }                                                                                             \
        }                                                                                                 \
    }
#define FB_STAGE_STORE(B)                                                                                 \
    _Pragma("unroll") for (int q = 0; q < FB_Q; ++q)                                                      \
        if (sval[q]) fb_lds[(B) * (FB_BUF / 16) + t + FT_THREADS * q] = sreg[q];
    FB_STAGE_LOAD(0)
    FB_STAGE_STORE(0)
    __syncthreads();
    const uint4 *cp = tcell + (size_t)tile * nproj * FT_PIX + g * 8 + (gl & 7);
    uint4 e0 = cp[0], e1 = cp[FT_PIX], e2 = cp[2 * FT_PIX], e3 = cp[3 * FT_PIX];   // table padded by 2*FB_A angles:
    // the in-place reloads of the last stage reach angle 4*nstage + 3 <= P + 2*FB_A - 2
    V acc[8];
#pragma unroll
    for (int J = 0; J < 8; ++J) acc[J] = vzero<4>();
    const uint32_t zoff = FB_A * FB_MAXR * 256;
#define FB_ROW(O, J) (*reinterpret_cast<const V *>(base + row_ror<J>(O)))
#define FB_HALF(J0)                                                                                       \
    {                                                                                                     \
        V a0 = FB_ROW(o0, J0), a1 = FB_ROW(o1, J0), b0 = FB_ROW(o0, J0 + 1), b1 = FB_ROW(o1, J0 + 1);     \
        V c0 = FB_ROW(o0, J0 + 2), c1 = FB_ROW(o1, J0 + 2), d0 = FB_ROW(o0, J0 + 3), d1 = FB_ROW(o1, J0 + 3); \
        acc[J0] += __uint_as_float(row_ror<J0>(w0)) * a0;     acc[J0] += __uint_as_float(row_ror<J0>(w1)) * a1;         \
        acc[J0 + 1] += __uint_as_float(row_ror<J0 + 1>(w0)) * b0; acc[J0 + 1] += __uint_as_float(row_ror<J0 + 1>(w1)) * b1; \
        acc[J0 + 2] += __uint_as_float(row_ror<J0 + 2>(w0)) * c0; acc[J0 + 2] += __uint_as_float(row_ror<J0 + 2>(w1)) * c1; \
        acc[J0 + 3] += __uint_as_float(row_ror<J0 + 3>(w0)) * d0; acc[J0 + 3] += __uint_as_float(row_ror<J0 + 3>(w1)) * d1; \
        /* pin: pure FMAs carry no chain, the DAG would otherwise sink all of a stage's FMAs below all of its reads */ \
        asm volatile("" : "+v"(acc[J0]), "+v"(acc[J0 + 1]), "+v"(acc[J0 + 2]), "+v"(acc[J0 + 3]));        \
    }
#define FB_STEP(E, I)                                                                                     \
    {                                                                                                     \
        const bool in = s * FB_A + (I) < nproj;                                                           \
        const uint32_t o0 = in ? E.x : zoff, w0 = in ? E.y : 0u, o1 = in ? E.z : zoff, w1 = in ? E.w : 0u; \
        E = cp[(size_t)(s * FB_A + (I) + FB_A) * FT_PIX];                                                 \
        FB_HALF(0) FB_HALF(4)                                                                             \
    }
    for (int s = 0; s < nstage; ++s) {
        if (s + 1 < nstage) { FB_STAGE_LOAD(s + 1) }
        const char *base = reinterpret_cast<const char *>(fb_lds) + (s & 1) * FB_BUF + gl * 16;
        FB_STEP(e0, 0) FB_STEP(e1, 1) FB_STEP(e2, 2) FB_STEP(e3, 3)
        if (s + 1 < nstage) { FB_STAGE_STORE((s + 1) & 1) }
        __syncthreads();
    }
#undef FB_STEP
#undef FB_HALF
#undef FB_ROW
#undef FB_STAGE_STORE
#undef FB_STAGE_LOAD
    // Un-rotate through LDS (the stage buffers are dead): lane l holds pixel (l + J) mod 8 in acc[J]; stored as is,
    // a wave instruction would scatter 16-byte pieces over 8 pixels (PMC: 3.1x the bytes written).  Afterwards
    // every group reads its pixels in order and the x read / write are whole 256-byte pieces.
    __syncthreads();
#define FB_PUT(J) fb_lds[(g * 8 + (int)row_ror<J>((uint32_t)(gl & 7))) * 16 + gl] = acc[J];
    FB_PUT(0) FB_PUT(1) FB_PUT(2) FB_PUT(3) FB_PUT(4) FB_PUT(5) FB_PUT(6) FB_PUT(7)
#undef FB_PUT
    __syncthreads();
    // the 8 column sums and the 8 reads of x go out together (clamped addresses for pixels outside the image, so that no branch
    // separates them: one after the other they were 16 memory round trips in a row), then pixel by pixel the update and the store
    const int off = c * 64 + gl * 4;
    float csv[8];
    V xv[8];
#pragma unroll
    for (int J = 0; J < 8; ++J) {
        const int lp = g * 8 + J;
        const int y = min(ty * FT_TY + lp / FT_TZ, n - 1), z = min(tz * FT_TZ + lp % FT_TZ, n - 1);
        const size_t p = (size_t)y * n + z;
        csv[J] = colsum ? colsum[p] : 1.f;
        if (alpha != 0.f) xv[J] = nt_ld<64>(reinterpret_cast<const V *>(x + p * sx + off));
    }
#pragma unroll
    for (int J = 0; J < 8; ++J) {
        int lp = g * 8 + J;
        int y = ty * FT_TY + lp / FT_TZ, z = tz * FT_TZ + lp % FT_TZ;
        if (y < n && z < n) {
            size_t p = (size_t)y * n + z;
            V a = fb_lds[lp * 16 + gl];
            if (colsum) { float cs = csv[J]; a = cs > 0.f ? a / cs : vzero<4>(); }
            float *xp = x + p * sx + off;
            V nv = beta * a;
            if (alpha != 0.f) nv = bp_axpby(alpha, xv[J], beta, a);
            if (clamp) { nv[0] = fmaxf(nv[0], 0.f); nv[1] = fmaxf(nv[1], 0.f); nv[2] = fmaxf(nv[2], 0.f); nv[3] = fmaxf(nv[3], 0.f); }
            nt_st<64>(nv, reinterpret_cast<V *>(xp));
        }
    }
}

// ---- back-projector, all angles, tile-stationary form with WAVE-UNIFORM entry lists (round 4) -----------------------------------
// k_bp_tile shares a cell {row offset, weight} x 2 among the 16 lanes of a group by DPP rotation: 28 lane moves and 16 address adds
// for every 8 pixels and angle, next to the 32 packed FMAs that do the work, and BOTH row reads of every pixel -- although a pixel has
// a second ray of an angle in one case of four (the second read then fetches the zero row: 39 % of the LDS reads and of the FMAs).
// Here a wave covers 128 slices (64 lanes x float2) and owns 32 pixels of a 16 x 16 tile (2 registers each: v[64:127]); what it
// has to do in a stage of BL_A = 3 angles is a LIST of entries {window byte offset | accumulator register, weight}, one per NONZERO
// weight (sysmat.cpp: build_bp_lists; 1.22 per pixel and angle), fetched 16 at a time by scalar loads.  An entry costs one
// v_and_or_b32 (the address), one ds_read_b64 and one v_pk_fma_f32 whose accumulator is picked by the VGPR index mode
// (s_set_gpr_idx_on: M0[7:0] is added to the register number of src2 and dst), the weight being the scalar operand: no lane moves,
// no branches, no reads of zeros.  The loop is one asm block on fixed registers (the index mode cannot be expressed otherwise):
// entries s[36:67] / s[68:99] (two sets: the scalar loads of the next batch go out before this batch's reads; a counted lgkmcnt
// stays valid beside them, see BL_FMAS), rows v[32:63], list pointer in vcc.  The residual rows of a stage (<= 26 per angle,
// 512 bytes each) are staged by LDS-DMA, the next stage into the other half of the workgroup's LDS while this one is worked on
// (2 x 39 KB; no registers, which the fixed blocks leave no room for); 8 waves, two workgroups per CU.  The lists stream from HBM
// once: a wave touches its next list with one vector load a stage ahead so that the scalar loads hit the L2, and the list bounds
// and window words of all stages sit in registers (one stage per lane).  A pixel's FMAs keep the order of k_bp_all (angles
// ascending, first ray before second); a skipped zero weight would have added +-0 to a sum that is never -0: bit-identical.
// Measured at 512^3 x 90 (profiles/r04_bp_list_development.md): 1.01 ms against 1.35 ms for k_bp_tile; the entry work is bound by
// vector-ALU issue (v_and_or_b32 and v_pk_fma_f32 are 4 cycles each: 9 cycles per entry and SIMD measured in isolation, 11 with the
// LDS reads), the rest is the staging (DMA issue + the wait at the stage's end) and the epilogue.
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v32f __attribute__((ext_vector_type(32)));
typedef uint32_t u16v __attribute__((ext_vector_type(16)));
constexpr int BL_TY = 16, BL_TZ = 16, BL_PIX = BL_TY * BL_TZ, BL_THREADS = 512, BL_WAVES = BL_THREADS / 64, BL_PPW = BL_PIX / BL_WAVES;
constexpr int BL_A = 3, BL_MAXR = 26, BL_ROWB = 512;   // angles per stage, rows of a 16 x 16 tile's window (<= 16 sqrt 2 + 2), bytes of a row
constexpr int BL_BUF = BL_A * BL_MAXR * BL_ROWB, BL_LDS_BYTES = 2 * BL_BUF;         // 79,872 bytes: two workgroups per CU
constexpr int BL_PAIRS = BL_MAXR / 2, BL_STAGE_PAIRS = BL_A * BL_PAIRS;              // DMA pieces (row pairs) of an angle / a stage
static_assert(BL_PPW == 32 && BL_MAXR % 2 == 0 && 2 * BL_LDS_BYTES <= 160 * 1024, "k_bp_list geometry");
constexpr int BL_BATCH = 8;                             // pairs per batch (two s_load_dwordx16)
// pixel q of wave w inside the tile (= Tables::bl_pixel, sysmat.h): waves own blocks of 8 x 4 pixels
__device__ __forceinline__ int bl_ly(int w, int q) { return (w >> 2) * 8 + (q >> 2); }
__device__ __forceinline__ int bl_lz(int w, int q) { return (w & 3) * 4 + (q & 3); }
// pair K of the set that starts at SGPR SB: s[SB+4K] = row offset | register of its first pixel (rows are 512 bytes apart: the low
// 9 bits of the offset are free; M0 takes the register from bits 7:0, the address is (entry & ~511) | 8 * lane), s[SB+4K+1] = that
// pixel's weight, s[SB+4K+2] = register of the second pixel, s[SB+4K+3] = its weight; the row lands in v[32+2K:33+2K]
#define BL_RD(SB, K)                                                                                      \
    "v_and_or_b32 v[32+2*" #K "], s[" #SB "+4*" #K "], %[mask], %[base]\n"                                \
    "ds_read_b64 v[32+2*" #K ":33+2*" #K "], v[32+2*" #K "]\n"
#define BL_FMA(SB, K, H)                                                                                  \
    "s_set_gpr_idx_on s[" #SB "+4*" #K "+" #H "], gpr_idx(SRC2,DST)\n"                                    \
    "v_pk_fma_f32 v[64:65], s[" #SB "+4*" #K "+" #H ":" #SB "+4*" #K "+" #H "+1], v[32+2*" #K ":33+2*" #K "], v[64:65] op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
#define BL_WFMA(SB, K, W) "s_waitcnt lgkmcnt(" #W ")\n" BL_FMA(SB, K, 0) BL_FMA(SB, K, 2)
#define BL_READS(SB) BL_RD(SB, 0) BL_RD(SB, 1) BL_RD(SB, 2) BL_RD(SB, 3) BL_RD(SB, 4) BL_RD(SB, 5) BL_RD(SB, 6) BL_RD(SB, 7)
// (a counted wait stays valid with the scalar loads of the next batch in flight: lgkmcnt(7 - K) leaves at most 7 - K of the
// 8 + 2 operations outstanding, so at least K + 1 LDS reads -- which return in order -- have landed whatever the scalar loads do)
#define BL_FMAS(SB)                                                                                       \
    BL_WFMA(SB, 0, 7) BL_WFMA(SB, 1, 6) BL_WFMA(SB, 2, 5) BL_WFMA(SB, 3, 4) BL_WFMA(SB, 4, 3) BL_WFMA(SB, 5, 2) BL_WFMA(SB, 6, 1) BL_WFMA(SB, 7, 0) \
    "s_set_gpr_idx_off\n"
#define BL_CLOB4(P, A, B, C, D) #P #A, #P #B, #P #C, #P #D
#define BL_CLOBBERS                                                                                       \
    BL_CLOB4(s, 68, 69, 70, 71), BL_CLOB4(s, 72, 73, 74, 75), BL_CLOB4(s, 76, 77, 78, 79), BL_CLOB4(s, 80, 81, 82, 83),      \
    BL_CLOB4(s, 84, 85, 86, 87), BL_CLOB4(s, 88, 89, 90, 91), BL_CLOB4(s, 92, 93, 94, 95), BL_CLOB4(s, 96, 97, 98, 99),      \
    "s33",                                                                                                \
    BL_CLOB4(v, 32, 33, 34, 35), BL_CLOB4(v, 36, 37, 38, 39), BL_CLOB4(v, 40, 41, 42, 43), BL_CLOB4(v, 44, 45, 46, 47),      \
    BL_CLOB4(v, 48, 49, 50, 51), BL_CLOB4(v, 52, 53, 54, 55), BL_CLOB4(v, 56, 57, 58, 59), BL_CLOB4(v, 60, 61, 62, 63),      \
    "vcc", "scc", "memory"

__global__ __launch_bounds__(BL_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4)))
void k_bp_list(float *__restrict__ x, const uint4 *__restrict__ lent, const uint32_t *__restrict__ lptr, const uint32_t *__restrict__ win,
               const float *__restrict__ r, const float *__restrict__ colsum, float alpha, float beta, int clamp,
               int nproj, int n, int sx, int tiles_z, int ntiles, int nchunk2)
{
    typedef VecOf<4>::T V;
    extern __shared__ V bl_lds[];
    const int xcd = blockIdx.x & 7, l = blockIdx.x >> 3;
    const int tile = (l / nchunk2) * 8 + xcd, c2 = l % nchunk2;
    if (tile >= ntiles) return;
    const int ty = tile / tiles_z, tz = tile - ty * tiles_z;
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const uint32_t *wn = win + (size_t)tile * nproj;
    const int nstage = (nproj + BL_A - 1) / BL_A;
    // One DMA instruction moves 64 x 16 bytes = a PAIR of consecutive rows of a window (lanes 0-31 the even row, 32-63 the odd one).
    // A stage has BL_A x BL_MAXR / 2 = 39 pairs: wave w moves pairs w, w + 8, ... (pair k = row pair k % 13 of angle k / 13).
    // The window words {first ray | rays << 16} of all angles sit in 3 registers, stage s in lane s (nstage <= 64), so that a
    // stage's staging depends on no scalar load; everything but the odd row's lane offset is scalar arithmetic.
    static_assert(BL_A == 3, "window words of a stage");
    const int jl = lane >> 5;
    const float *rc = r + (size_t)c2 * 128 + (lane & 31) * 4 + (size_t)jl * sx;
    uint32_t wv0 = 0, wv1 = 0, wv2 = 0;
    if (lane < nstage) {
        const int i0 = lane * BL_A;
        wv0 = wn[i0];
        if (i0 + 1 < nproj) wv1 = wn[i0 + 1];
        if (i0 + 2 < nproj) wv2 = wn[i0 + 2];
    }
#define BL_DMA1(S, K)                                                                                     \
    if ((K) < BL_STAGE_PAIRS) {                                                                           \
        const int a = (K) / BL_PAIRS, pr = (K) - a * BL_PAIRS;                                            \
        const uint32_t ww = a == 0 ? w0 : a == 1 ? w1 : w2;                                               \
        if ((uint32_t)(2 * pr) < (ww >> 16)) {                                                            \
            if ((uint32_t)(2 * pr + jl) < (ww >> 16))                                                     \
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(rc + ((size_t)((S) * BL_A + a) * n + (ww & 0xFFFFu) + 2 * pr) * sx), \
                                                 (__attribute__((address_space(3))) void *)(bl_lds + ((S) & 1) * (BL_BUF / 16) + (a * BL_MAXR + 2 * pr) * (BL_ROWB / 16)), 16, 0, 0); \
        }                                                                                                 \
    }
#define BL_STAGE_DMA(S)                                                                                   \
    {                                                                                                     \
        const uint32_t w0 = __builtin_amdgcn_readlane(wv0, (S)), w1 = __builtin_amdgcn_readlane(wv1, (S)), w2 = __builtin_amdgcn_readlane(wv2, (S)); \
        _Pragma("unroll") for (int q = 0; q < (BL_STAGE_PAIRS + BL_WAVES - 1) / BL_WAVES; ++q) { BL_DMA1(S, wave + BL_WAVES * q) } \
    }
    BL_STAGE_DMA(0)
    // The lists stream from HBM once and a scalar load has nobody to hide a miss behind: every wave touches the lines of its NEXT
    // list with one vector load a stage ahead (lane k: batch k of the list), so that the scalar loads hit the L2.
    // (the list bounds of all stages, one stage per lane, so that no stage starts behind a scalar miss: nstage <= 64)
    const uint32_t *lp = lptr + (size_t)tile * nstage * BL_WAVES + wave;
    uint32_t pv0 = 0, pv1 = 0;
    if (lane < nstage) { pv0 = lp[(size_t)lane * BL_WAVES]; pv1 = lp[(size_t)lane * BL_WAVES + 1]; }
#define BL_TOUCH(S)                                                                                       \
    {                                                                                                     \
        const uint32_t t0 = __builtin_amdgcn_readlane(pv0, (S)), t1 = __builtin_amdgcn_readlane(pv1, (S)); \
        if (t0 + lane < t1) touched = *reinterpret_cast<const uint32_t *>(lent + (size_t)(t0 + lane) * BL_BATCH);   /* a list has <= 16 batches */ \
    }
    uint32_t touched = 0;
    BL_TOUCH(0)
    // the column sums of the wave's pixels, pixel q in lane q (as scalar loads in the epilogue they were 32 misses in a row)
    float csv = 0.f;
    if (colsum && lane < BL_PPW) {
        const int y = ty * BL_TY + bl_ly(wave, lane), z = tz * BL_TZ + bl_lz(wave, lane);
        if (y < n && z < n) csv = colsum[(size_t)y * n + z];
    }
    v32f acc_lo, acc_hi;                                // pixel q of the wave: registers 2q, 2q+1 of v[64:127]
#pragma unroll
    for (int q = 0; q < 32; ++q) { acc_lo[q] = 0.f; acc_hi[q] = 0.f; }
    // (the dynamic LDS block is the kernel's only one, so it starts at LDS address 0 and a row offset IS its address)
    if ((uint32_t)(size_t)(__attribute__((address_space(3))) V *)bl_lds != 0u) __builtin_trap();
    const uint32_t base = (uint32_t)lane * 8u, mask = ~(uint32_t)(BL_ROWB - 1);
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(touched) :: "memory");
    __syncthreads();
    for (int s = 0; s < nstage; ++s) {
        // the stage's first batch is requested before anything else of the stage (as two 16-dword values bound to the registers the
        // loop keeps its first entry set in), so that it arrives behind the staging code instead of in front of the loop
        const uint32_t b0 = __builtin_amdgcn_readlane(pv0, s);
        uint32_t nb = __builtin_amdgcn_readlane(pv1, s) - b0;
        const uint4 *ep = lent + (size_t)b0 * BL_BATCH;
        u16v ea, eb;
        asm volatile("s_load_dwordx16 %0, %2, 0x0\n"
                     "s_load_dwordx16 %1, %2, 0x40\n" : "={s[36:51]}"(ea), "={s[52:67]}"(eb) : "s"(ep));   // (waited for inside the loop's block)
        if (s + 1 < nstage) { BL_STAGE_DMA(s + 1) BL_TOUCH(s + 1) }
        {   // (an empty list is skipped inside the block: a branch around it would make the accumulators merge values)
            asm volatile("s_waitcnt lgkmcnt(0)\n"         /* the first batch (also when the list is empty: nothing may land later) */
                         "s_cmp_eq_u32 %[nb], 0\n"
                         "s_cbranch_scc1 3f\n"
                         "s_mov_b32 s33, m0\n"
                         "s_mov_b64 vcc, %[ep]\n"
                         "1:\n"
                         "s_load_dwordx16 s[68:83], vcc, 0x80\n"
                         "s_load_dwordx16 s[84:99], vcc, 0xc0\n"
                         BL_READS(36)
                         BL_FMAS(36)
                         "s_sub_u32 %[nb], %[nb], 1\n"
                         "s_cmp_eq_u32 %[nb], 0\n"
                         "s_cbranch_scc1 2f\n"
                         "s_add_u32 vcc_lo, vcc_lo, 0x100\n"
                         "s_addc_u32 vcc_hi, vcc_hi, 0\n"
                         "s_load_dwordx16 s[36:51], vcc, 0x0\n"
                         "s_load_dwordx16 s[52:67], vcc, 0x40\n"
                         BL_READS(68)
                         BL_FMAS(68)
                         "s_sub_u32 %[nb], %[nb], 1\n"
                         "s_cmp_lg_u32 %[nb], 0\n"
                         "s_cbranch_scc1 1b\n"
                         "2:\n"
                         "s_waitcnt lgkmcnt(0)\n"
                         "s_mov_b32 m0, s33\n"
                         "3:\n"
                         : "+{v[64:95]}"(acc_lo), "+{v[96:127]}"(acc_hi), [nb] "+s"(nb), "+{s[36:51]}"(ea), "+{s[52:67]}"(eb)
                         : [ep] "s"(ep), [base] "v"(base), [mask] "v"(mask)
                         : BL_CLOBBERS);
        }
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(touched) :: "memory");     // this wave's pieces of the next stage have landed
        __syncthreads();                                                    // ... everybody's have, and every wave is done with this stage's rows
    }
#undef BL_STAGE_DMA
#undef BL_DMA1
#undef BL_TOUCH
    // epilogue in two halves of 16 pixels: the 16 reads of x go out together (clamped addresses for pixels outside the image, so
    // that no branch separates them), then pixel by pixel the update and the store
    const int off = c2 * 128 + lane * 2;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        v2f xv[16];
        if (alpha != 0.f) {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int y = min(ty * BL_TY + bl_ly(wave, h * 16 + k), n - 1), z = min(tz * BL_TZ + bl_lz(wave, h * 16 + k), n - 1);
                xv[k] = nt_ld<64>(reinterpret_cast<const v2f *>(x + ((size_t)y * n + z) * sx + off));
            }
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int q = h * 16 + k;
            const int y = ty * BL_TY + bl_ly(wave, q), z = tz * BL_TZ + bl_lz(wave, q);
            if (y < n && z < n) {
                v2f a = h == 0 ? v2f{acc_lo[2 * k], acc_lo[2 * k + 1]} : v2f{acc_hi[2 * k], acc_hi[2 * k + 1]};
                if (colsum) { const float cs = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(csv), q)); a = cs > 0.f ? a / cs : v2f{0.f, 0.f}; }
                v2f nv = beta * a;
                if (alpha != 0.f) nv = bp_axpby(alpha, xv[k], beta, a);
                if (clamp) { nv.x = fmaxf(nv.x, 0.f); nv.y = fmaxf(nv.y, 0.f); }
                nt_st<64>(nv, reinterpret_cast<v2f *>(x + ((size_t)y * n + z) * sx + off));
            }
        }
    }
}
#undef BL_CLOBBERS
#undef BL_CLOB4
#undef BL_FMAS
#undef BL_READS
#undef BL_WFMA
#undef BL_FMA
#undef BL_RD

// ---- forward projector, all angles, sheared strips with WAVE-UNIFORM entry lists (round 4) -------------------------------------------
// The strip decomposition of k_fp_strip (passes of neighbouring angles, strips of 16 pixels sheared with the pass's mean direction,
// march segments; sysmat.cpp: build_fp_lists) with the machinery of k_bp_list: a wave covers 128 slices (64 lanes x float2) and owns
// the rays of ONE angle of the pass, ray j in accumulator j mod 32 (v[64:127]); per tile of 8 march steps (16 x 8 pixels x 512 B,
// staged by LDS-DMA into one half of the LDS while the other is worked on; pixels outside the image read zeros) it works through a
// list of entries {byte offset of the pixel in the staged tiles | accumulator register, weight} -- one v_and_or_b32, one ds_read_b64
// and one v_pk_fma_f32 into the accumulator M0 picks, no lane moves and no per-lane entry loads -- and then through the tile's flush
// records {accumulator register, partial-sum id}: the sums of the rays that leave the strip here are stored (read through the index
// mode as well) and cleared.  16 waves (up to 16 angles of a pass side by side), one workgroup per CU.  A ray's entries keep their
// order, so a partial sum is the same FMA chain as in k_fp_strip; k_fp_tile_reduce adds a ray's partial sums in ascending strip order.
constexpr int FL_W = 16, FL_TH = 8, FL_PIX = FL_W * FL_TH, FL_THREADS = 1024, FL_WAVES = FL_THREADS / 64, FL_BATCH = 16, FL_PIXB = 512;
constexpr int FL_TILE_BYTES = 2 * FL_PIX * FL_PIXB, FL_MAXSTEPS = 64 * FL_TH, FL_LDS_BYTES = FL_TILE_BYTES + FL_MAXSTEPS * 4;
struct FlItemD { int pass, v0; uint32_t tile0, ntiles, lp0, work, pad0, pad1; };

#define FL_RD(SB, K)                                                                                      \
    "v_and_or_b32 v[32+2*" #K "], s[" #SB "+2*" #K "], %[mask], %[base]\n"                                \
    "ds_read_b64 v[32+2*" #K ":33+2*" #K "], v[32+2*" #K "]\n"
#define FL_FMA(SB, K, W)                                                                                  \
    "s_waitcnt lgkmcnt(" #W ")\n"                                                                         \
    "s_set_gpr_idx_on s[" #SB "+2*" #K "], gpr_idx(SRC2,DST)\n"                                           \
    "v_pk_fma_f32 v[64:65], s[" #SB "+2*" #K ":" #SB "+2*" #K "+1], v[32+2*" #K ":33+2*" #K "], v[64:65] op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
#define FL_READS(SB)                                                                                      \
    FL_RD(SB, 0) FL_RD(SB, 1) FL_RD(SB, 2) FL_RD(SB, 3) FL_RD(SB, 4) FL_RD(SB, 5) FL_RD(SB, 6) FL_RD(SB, 7)             \
    FL_RD(SB, 8) FL_RD(SB, 9) FL_RD(SB, 10) FL_RD(SB, 11) FL_RD(SB, 12) FL_RD(SB, 13) FL_RD(SB, 14) FL_RD(SB, 15)
#define FL_FMAS(SB)                                                                                       \
    FL_FMA(SB, 0, 15) FL_FMA(SB, 1, 14) FL_FMA(SB, 2, 13) FL_FMA(SB, 3, 12) FL_FMA(SB, 4, 11) FL_FMA(SB, 5, 10) FL_FMA(SB, 6, 9) FL_FMA(SB, 7, 8) \
    FL_FMA(SB, 8, 7) FL_FMA(SB, 9, 6) FL_FMA(SB, 10, 5) FL_FMA(SB, 11, 4) FL_FMA(SB, 12, 3) FL_FMA(SB, 13, 2) FL_FMA(SB, 14, 1) FL_FMA(SB, 15, 0) \
    "s_set_gpr_idx_off\n"
#define FL_CLOB4(P, A, B, C, D) #P #A, #P #B, #P #C, #P #D
#define FL_CLOBBERS                                                                                       \
    FL_CLOB4(s, 68, 69, 70, 71), FL_CLOB4(s, 72, 73, 74, 75), FL_CLOB4(s, 76, 77, 78, 79), FL_CLOB4(s, 80, 81, 82, 83),      \
    FL_CLOB4(s, 84, 85, 86, 87), FL_CLOB4(s, 88, 89, 90, 91), FL_CLOB4(s, 92, 93, 94, 95), FL_CLOB4(s, 96, 97, 98, 99),      \
    "s33",                                                                                                \
    FL_CLOB4(v, 32, 33, 34, 35), FL_CLOB4(v, 36, 37, 38, 39), FL_CLOB4(v, 40, 41, 42, 43), FL_CLOB4(v, 44, 45, 46, 47),      \
    FL_CLOB4(v, 48, 49, 50, 51), FL_CLOB4(v, 52, 53, 54, 55), FL_CLOB4(v, 56, 57, 58, 59), FL_CLOB4(v, 60, 61, 62, 63),      \
    "vcc", "scc", "memory"

__global__ __launch_bounds__(FL_THREADS)
void k_fp_list(const float *__restrict__ x, const FlItemD *__restrict__ items, const int *__restrict__ orient, const int *__restrict__ shift,
               const uint2 *__restrict__ lent, const uint32_t *__restrict__ lptr, const uint2 *__restrict__ fent, const uint32_t *__restrict__ fptr,
               float *__restrict__ part, int n, int sx, int nitems, int cpair0, int ncpp, int ncp, const float *__restrict__ zero)
{
    typedef VecOf<4>::T V;
    extern __shared__ V fl_lds[];                       // [2][FL_PIX][32]: tile parity, pixel, 512 bytes; then the item's shifts
    // all 128-slice pieces of an item run back to back on one XCD (workgroups b and b+8 share an XCD): they read the same tables
    const int xcd = blockIdx.x & 7, l = blockIdx.x >> 3;
    const int it = (l / ncpp) * 8 + xcd, c2 = l % ncpp;
    if (it >= nitems) return;
    const FlItemD I = items[it];
    const int t = threadIdx.x, lane = t & 63, jl = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int o = orient[I.pass];
    const int ntiles = (int)I.ntiles;
    int *sh_l = reinterpret_cast<int *>(reinterpret_cast<char *>(fl_lds) + FL_TILE_BYTES);
    {
        const int *sh = shift + (size_t)I.pass * n;
        for (int i = t; i < ntiles * FL_TH; i += FL_THREADS) sh_l[i] = sh[min((int)I.tile0 * FL_TH + i, n - 1)];
    }
    const float *xc = x + (size_t)(cpair0 + c2) * 128 + (lane & 31) * 4;
    const float *zsrc = zero + (lane & 31) * 4;
    // the list and flush-list bounds of all tiles, one tile per lane (ntiles <= 64)
    const uint32_t *lp = lptr + I.lp0 + wave, *fp = fptr + I.lp0 + wave;
    uint32_t pv0 = 0, pv1 = 0, fv0 = 0, fv1 = 0;
    if (lane < ntiles) {
        pv0 = lp[(size_t)lane * FL_WAVES]; pv1 = lp[(size_t)lane * FL_WAVES + 1];
        fv0 = fp[(size_t)lane * FL_WAVES]; fv1 = fp[(size_t)lane * FL_WAVES + 1];
    }
    __syncthreads();                                    // the shifts are in place
    // One DMA instruction moves 64 x 16 bytes = the 512-byte images of two neighbouring pixels (lanes 0-31 the even one); a tile has
    // 64 such pairs, wave w moves pairs w, w + 16, w + 32, w + 48 (pair p = pixels 2p, 2p + 1 of march step p / 8)
#define FL_STAGE(TT)                                                                                      \
    _Pragma("unroll") for (int q = 0; q < FL_PIX / 2 / FL_WAVES; ++q) {                                   \
        const int p = wave + FL_WAVES * q;                                                                \
        const int lu = p >> 3, u = ((int)I.tile0 + (TT)) * FL_TH + lu;                                    \
        const int vv = I.v0 + sh_l[(TT) * FL_TH + lu] + 2 * (p & 7) + jl;                                 \
        const bool ok = u < n && (unsigned)vv < (unsigned)n;                                              \
        const size_t pix = o ? (size_t)vv * n + u : (size_t)u * n + vv;                                   \
        const float *src = ok ? xc + pix * sx : zsrc;                                                     \
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,             \
                                         (__attribute__((address_space(3))) void *)(fl_lds + (((TT) & 1) * FL_PIX + 2 * p) * (FL_PIXB / 16)), 16, 0, 0); \
    }
#define FL_TOUCH(TT)                                                                                      \
    {                                                                                                     \
        const uint32_t t0 = __builtin_amdgcn_readlane(pv0, (TT)), t1 = __builtin_amdgcn_readlane(pv1, (TT)); \
        if (t0 + lane < t1) touched = *reinterpret_cast<const uint32_t *>(lent + (size_t)(t0 + lane) * FL_BATCH);   /* (a list of > 64 batches is touched in part) */ \
    }
    uint32_t touched = 0;
    FL_STAGE(0)
    FL_TOUCH(0)
    v32f acc_lo, acc_hi;                                // ray j of the wave's angle: registers 2 (j mod 32), + 1 of v[64:127]
#pragma unroll
    for (int q = 0; q < 32; ++q) { acc_lo[q] = 0.f; acc_hi[q] = 0.f; }
    // (the dynamic LDS block is the kernel's only one, so it starts at LDS address 0 and a pixel's offset IS its address)
    if ((uint32_t)(size_t)(__attribute__((address_space(3))) V *)fl_lds != 0u) __builtin_trap();
    const uint32_t base = (uint32_t)lane * 8u, mask = ~(uint32_t)(FL_PIXB - 1);
    // partial sum `id` of this 128-slice piece: part[(id * ncp + 2 c2) * 64 + 2 lane]
    const uint64_t pb = (uint64_t)(size_t)(part + (size_t)c2 * 128);
    const uint32_t pb_lo = (uint32_t)pb, pb_hi = (uint32_t)(pb >> 32), pstride = (uint32_t)ncp * 256u;
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(touched) :: "memory");
    __syncthreads();
    for (int tt = 0; tt < ntiles; ++tt) {
        // the tile's first batch is requested before anything else of the tile (two 16-dword values bound to the registers the loop
        // keeps its first entry set in): it arrives behind the staging code instead of in front of the loop
        const uint32_t b0 = __builtin_amdgcn_readlane(pv0, tt);
        uint32_t nb = __builtin_amdgcn_readlane(pv1, tt) - b0;
        const uint2 *ep = lent + (size_t)b0 * FL_BATCH;
        u16v ea, eb;
        asm volatile("s_load_dwordx16 %0, %2, 0x0\n"
                     "s_load_dwordx16 %1, %2, 0x40\n" : "={s[36:51]}"(ea), "={s[52:67]}"(eb) : "s"(ep));   // (waited for inside the loop's block)
        if (tt + 1 < ntiles) { FL_STAGE(tt + 1) FL_TOUCH(tt + 1) }
        {   // (an empty list is skipped inside the block: a branch around it would make the accumulators merge values, i.e. copies)
            asm volatile("s_waitcnt lgkmcnt(0)\n"         /* the first batch (also when the list is empty: nothing may land later) */
                         "s_cmp_eq_u32 %[nb], 0\n"
                         "s_cbranch_scc1 3f\n"
                         "s_mov_b32 s33, m0\n"
                         "s_mov_b64 vcc, %[ep]\n"
                         "1:\n"
                         "s_load_dwordx16 s[68:83], vcc, 0x80\n"
                         "s_load_dwordx16 s[84:99], vcc, 0xc0\n"
                         FL_READS(36)
                         FL_FMAS(36)
                         "s_sub_u32 %[nb], %[nb], 1\n"
                         "s_cmp_eq_u32 %[nb], 0\n"
                         "s_cbranch_scc1 2f\n"
                         "s_add_u32 vcc_lo, vcc_lo, 0x100\n"
                         "s_addc_u32 vcc_hi, vcc_hi, 0\n"
                         "s_load_dwordx16 s[36:51], vcc, 0x0\n"
                         "s_load_dwordx16 s[52:67], vcc, 0x40\n"
                         FL_READS(68)
                         FL_FMAS(68)
                         "s_sub_u32 %[nb], %[nb], 1\n"
                         "s_cmp_lg_u32 %[nb], 0\n"
                         "s_cbranch_scc1 1b\n"
                         "2:\n"
                         "s_waitcnt lgkmcnt(0)\n"
                         "s_mov_b32 m0, s33\n"
                         "3:\n"
                         : "+{v[64:95]}"(acc_lo), "+{v[96:127]}"(acc_hi), [nb] "+s"(nb), "+{s[36:51]}"(ea), "+{s[52:67]}"(eb)
                         : [ep] "s"(ep), [base] "v"(base), [mask] "v"(mask)
                         : FL_CLOBBERS);
        }
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(touched) :: "memory");     // this wave's pieces of the next tile have landed (waited for before the flush, so that
        // the flush's stores are not: they have the whole next tile to complete)
        const uint32_t f0 = __builtin_amdgcn_readlane(fv0, tt);
        uint32_t nf = __builtin_amdgcn_readlane(fv1, tt) - f0;
        {   // the rays that leave the strip in this tile: accumulator (read through the index mode) -> partial sum, accumulator cleared
            const uint2 *fr = fent + f0;
            asm volatile("s_cmp_eq_u32 %[nf], 0\n"
                         "s_cbranch_scc1 3f\n"
                         "s_mov_b32 s33, m0\n"
                         "s_mov_b64 vcc, %[fr]\n"
                         "1:\n"
                         "s_load_dwordx2 s[36:37], vcc, 0x0\n"
                         "s_waitcnt lgkmcnt(0)\n"
                         "s_set_gpr_idx_on s36, gpr_idx(SRC0)\n"
                         "v_mov_b32 v32, v64\n"
                         "v_mov_b32 v33, v65\n"
                         "s_set_gpr_idx_on s36, gpr_idx(DST)\n"
                         "v_mov_b32 v64, 0\n"
                         "v_mov_b32 v65, 0\n"
                         "s_set_gpr_idx_off\n"
                         "s_mul_hi_u32 s39, s37, %[pstride]\n"
                         "s_mul_i32 s38, s37, %[pstride]\n"
                         "s_add_u32 s38, s38, %[pb_lo]\n"
                         "s_addc_u32 s39, s39, %[pb_hi]\n"
                         "global_store_dwordx2 %[base], v[32:33], s[38:39]\n"
                         "s_add_u32 vcc_lo, vcc_lo, 8\n"
                         "s_addc_u32 vcc_hi, vcc_hi, 0\n"
                         "s_sub_u32 %[nf], %[nf], 1\n"
                         "s_cmp_lg_u32 %[nf], 0\n"
                         "s_cbranch_scc1 1b\n"
                         "s_mov_b32 m0, s33\n"
                         "3:\n"
                         : "+{v[64:95]}"(acc_lo), "+{v[96:127]}"(acc_hi), [nf] "+s"(nf)
                         : [fr] "s"(fr), [base] "v"(base), [pstride] "s"(pstride), [pb_lo] "s"(pb_lo), [pb_hi] "s"(pb_hi)
                         : "s33", "s36", "s37", "s38", "s39", "v32", "v33", "vcc", "scc", "memory");
        }
        __syncthreads();                                                    // ... everybody's have, and every wave is done with this tile
    }
#undef FL_TOUCH
#undef FL_STAGE
}
#undef FL_CLOBBERS
#undef FL_CLOB4
#undef FL_FMAS
#undef FL_READS
#undef FL_FMA
#undef FL_RD

// ---- ART (Kaczmarz), row-sequential by definition (ctvlib.cpp:137-155) -------------------------------
// a = (b_j - A_j x)/|A_j|^2 ; x += A_j^T a beta, one row after the other: row j+1 shares pixels with row j, so
// rows cannot run side by side.  The parallelism that exists is across slices (lanes) and inside a row: one
// 1024-thread workgroup owns 64 slices, its 16 waves split the row's entries for the dot product (LDS reduce)
// and again for the update.  Two barriers per row; the grid is only Nslice/64 workgroups, so ART uses a small
// part of the chip -- it is the reference CPU path's default algorithm, kept for completeness.
constexpr int ART_WAVES = 16;

__global__ __launch_bounds__(1024) void k_art(float *__restrict__ x, const uint32_t *__restrict__ rptr,
                                               const uint2 *__restrict__ rent, const float *__restrict__ b,
                                               const float *__restrict__ inner, float beta, int nrows, int sx,
                                               const int32_t *__restrict__ order)
{
    __shared__ float red[ART_WAVES][64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int off = blockIdx.x * 64 + lane;
    float *xp = x + off;
    for (int q = 0; q < nrows; ++q) {
        int row = order ? order[q] : q;                // randART: a permutation of the rows (ctvlib.cpp:158-179)
        float ip = inner[row];
        if (!(ip > 0.f)) continue;                     // uniform: an empty ray would divide by zero in the reference
        uint32_t beg = rptr[row], end = rptr[row + 1];
        uint32_t seg = (end - beg + ART_WAVES - 1) / ART_WAVES;
        uint32_t kb = min(beg + wave * seg, end), ke = min(kb + seg, end);
        float dot = 0.f;
#pragma unroll 4
        for (uint32_t k = kb; k < ke; ++k) {
            uint2 e = rent[k];
            dot += __uint_as_float(e.y) * xp[(size_t)e.x * sx];
        }
        red[wave][lane] = dot;
        __syncthreads();
        float tot = 0.f;
#pragma unroll
        for (int w = 0; w < ART_WAVES; ++w) tot += red[w][lane];
        float a = (b[(size_t)row * sx + off] - tot) / ip;
#pragma unroll 4
        for (uint32_t k = kb; k < ke; ++k) {
            uint2 e = rent[k];
            xp[(size_t)e.x * sx] += __uint_as_float(e.y) * a * beta;
        }
        __syncthreads();                               // the next row reads what this one wrote
    }
}

// ---- ART in natural row order, one angle at a time ------------------------------------------------------------------
// Two rays of one angle that are not neighbours share no pixel (a unit pixel is crossed by at most two unit-spaced rays).
// So within an angle the Kaczmarz chain a_j = (b_j - A_j x^{(j)}) / |A_j|^2, x^{(j+1)} = x^{(j)} + beta a_j A_j^T only
// couples neighbours:  A_j x^{(j)} = A_j x^{(0)} + beta a_{j-1} (A_j . A_{j-1}).  One angle of the sweep is therefore
//   d = A_i x (a forward projection of the angle),
//   a_j = (b_j - d_j - beta a_{j-1} G_{j-1}) / |A_j|^2   (k_art_chain: a scalar recurrence along the rays, lanes = slices),
//   x += beta A_i^T a (k_bp_art: the two updates of a pixel in ray order, (w a) beta like ctvlib.cpp:152),
// the same iterates as the row-sequential k_art up to the rounding of the dot products (d + correction instead of a dot
// over the updated pixels): 60 x 3 launches instead of 15360 row steps with two barriers each at 256^3 x 60.
// The recurrence is affine, a_j = u_j + v_j a_{j-1} with u_j = (b_j - d_j)/|A_j|^2, v_j = -beta G_{j-1}/|A_j|^2, so it need not be
// walked ray by ray (512 dependent steps on 8 waves took 85 us per angle at 512^3, 22 % of an ART sweep): a workgroup of
// ART_CW waves owns 64 slices, wave w composes the maps of its segment of rays (U_w, V_w), the segment start values follow
// from at most ART_CW - 1 compositions through LDS, and every wave then REPLAYS its segment with the reference's own
// expression from its start value.  Inside a segment the arithmetic is the sequential one; across segments the start value
// carries the rounding of the composed maps (~1e-7 relative).  2 * ceil(N / ART_CW) + ART_CW dependent steps.
constexpr int ART_CW = 16;

__global__ __launch_bounds__(64 * ART_CW) void k_art_chain(const float *__restrict__ d, const float *__restrict__ b,
                                                            const float *__restrict__ inner, const float *__restrict__ cross,
                                                            float *__restrict__ a_out, float beta, int row0, int nray, int sx,
                                                            int chunk0)
{
    __shared__ float su[ART_CW][64], sv[ART_CW][64];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int s = (chunk0 + blockIdx.x) * 64 + lane;         // sx is a multiple of 64; chunk0: first chunk of a sub-slab
    const int L = (nray + ART_CW - 1) / ART_CW;
    const int j0 = min(wave * L, nray), j1 = min(j0 + L, nray);
    constexpr int U = 8;                                     // the loads of U rays are independent of the chain: issue them together
    // phase 1: the composed map of the segment, a_{j1-1} = cu + cv * a_{j0-1}
    float cu = 0.f, cv = 1.f;
    for (int j = j0; j < j1; j += U) {
        float dv[U], bv[U], ipv[U], gv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int jj = min(j + u, nray - 1), row = row0 + jj;
            const size_t o = (size_t)row * sx + s;
            dv[u] = d[o]; bv[u] = b[o]; ipv[u] = inner[row]; gv[u] = jj > 0 ? cross[row - 1] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (j + u < j1) {
                float uj = 0.f, vj = 0.f;
                if (ipv[u] > 0.f) { uj = (bv[u] - dv[u]) / ipv[u]; vj = -(beta * gv[u]) / ipv[u]; }   // an empty ray: a = 0
                cu = uj + vj * cu; cv = vj * cv;
            }
        }
    }
    su[wave][lane] = cu; sv[wave][lane] = cv;
    __syncthreads();
    // phase 2: a of the ray before this segment
    float aprev = 0.f;
    for (int k = 0; k < wave; ++k) aprev = su[k][lane] + sv[k][lane] * aprev;
    // phase 3: the segment itself, with the expression of the row-sequential form (ctvlib.cpp:146-148)
    float gprev = j0 > 0 ? cross[row0 + j0 - 1] : 0.f;
    for (int j = j0; j < j1; j += U) {
        float dv[U], bv[U], ipv[U], gv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int row = row0 + min(j + u, nray - 1);
            const size_t o = (size_t)row * sx + s;
            dv[u] = d[o]; bv[u] = b[o]; ipv[u] = inner[row]; gv[u] = cross[row];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (j + u < j1) {
                float a = 0.f;
                if (ipv[u] > 0.f) a = (bv[u] - (dv[u] + beta * aprev * gprev)) / ipv[u];   // an empty ray is skipped (a = 0)
                a_out[(size_t)(row0 + j + u) * sx + s] = a;
                aprev = a; gprev = gv[u];
            }
        }
    }
}

template <int VEC, int PPW>
// stream: non-temporal voxel accesses (slabs beyond the Infinity Cache, like k_sart_tile: the chained ART sweep 29.3 -> 26.0 ms)
__global__ __launch_bounds__(256) void k_bp_art(float *__restrict__ x, const CellD *__restrict__ cell,
                                                 const float *__restrict__ a, float beta, int npix, int sx,
                                                 int ngroups, int nchunk, int stream, int chunk0)
{
    typedef typename VecOf<VEC>::T V;
    int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int lane = threadIdx.x & 63;
    int gw = blockIdx.x * 4 + wave;
    int chunk = gw / ngroups;
    int grp = gw - chunk * ngroups;
    int p0 = grp * PPW;
    if (p0 >= npix || chunk >= nchunk) return;   // grid is rounded up to whole workgroups
    int off = (chunk0 + chunk) * (64 * VEC) + lane * VEC;
#pragma unroll
    for (int q = 0; q < PPW; ++q) {
        int p = p0 + q;
        if (p >= npix) break;
        CellD c = cell[p];
        if (c.w0 == 0.f && c.w1 == 0.f) continue;
        const V *xp = reinterpret_cast<const V *>(x + (size_t)p * sx + off);
        V xv = stream ? __builtin_nontemporal_load(xp) : *xp;
        // ascending ray order, each term rounded like `val * a * beta`
        uint32_t ra = c.r0, rb = c.r1; float wa = c.w0, wb = c.w1;
        if (wb != 0.f && (wa == 0.f || rb < ra)) { uint32_t tr = ra; ra = rb; rb = tr; float tw = wa; wa = wb; wb = tw; }
        if (wa != 0.f) {
            V av = *reinterpret_cast<const V *>(a + (size_t)ra * sx + off);
#pragma unroll
            for (int i = 0; i < VEC; ++i) vset<VEC>(xv, i, __fadd_rn(velem<VEC>(xv, i), __fmul_rn(__fmul_rn(wa, velem<VEC>(av, i)), beta)));
        }
        if (wb != 0.f) {
            V bv = *reinterpret_cast<const V *>(a + (size_t)rb * sx + off);
#pragma unroll
            for (int i = 0; i < VEC; ++i) vset<VEC>(xv, i, __fadd_rn(velem<VEC>(xv, i), __fmul_rn(__fmul_rn(wb, velem<VEC>(bv, i)), beta)));
        }
        if (stream) __builtin_nontemporal_store(xv, reinterpret_cast<V *>(x + (size_t)p * sx + off));
        else *reinterpret_cast<V *>(x + (size_t)p * sx + off) = xv;
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
               int stream; };   // stream: non-temporal stores of x_new / the snapshot (slabs beyond the Infinity Cache: -3 %; thin slabs: +3 %)

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
                       float &pen, float &pfn, float &pcn, const float *Tp, float *Tn) __attribute__((always_inline)) {
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

// FGP-TV (tv_fgp.cu).  Non-periodic: i-1 below the first GLOBAL slice and i+1 above the last are "0" terms.
// D = max(0, A - lambda (P1 + P2 + P3 - P1[i-1] - P2[j-1] - P3[k-1]))        (:44-65, :143-154)
// The two expressions of an FGP iteration, spelled out operation by operation (no contraction left to the compiler), so that
// every kernel that evaluates them -- one iteration per pass, two per pass -- rounds alike: the forms are compared bit for bit.
__device__ __forceinline__ float fgp_d_of(float a, float lambda, float p1, float p2, float p3, float v1, float v2, float v3)
{
#pragma clang fp contract(off)
    const float t = p1 + p2 + p3 - v1 - v2 - v3;
    return fmaxf(__builtin_fmaf(-lambda, t, a), 0.f);
}
__device__ __forceinline__ void fgp_p_of(float &a, float &b, float &c, float multip, float v1, float v2, float v3)
{
#pragma clang fp contract(off)
    a = __builtin_fmaf(multip, v1, a); b = __builtin_fmaf(multip, v2, b); c = __builtin_fmaf(multip, v3, c);
    const float denom = __builtin_fmaf(c, c, __builtin_fmaf(b, b, a * a));
    if (denom > 1.0f) {
        const float sq = 1.0f / sqrtf(denom);
        a *= sq; b *= sq; c *= sq;
    }
}

// D may be A itself (the fused single-slab form finishes in place: each voxel reads only its own A)
__global__ __launch_bounds__(256) void k_fgp_obj(const float *A, float *D,
                                                  const float *__restrict__ P1, const float *__restrict__ P2,
                                                  const float *__restrict__ P3, const float *__restrict__ p1_lo,
                                                  int first, float lambda, int n, int nx, int sx)
{
    int lane = threadIdx.x & 63;
    int nchunk = (nx + 63) >> 6   /* computed width, not the row pitch: the pitch may carry padding */;
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
            D[q] = fgp_d_of(A[q], lambda, P1[q], P2[q], P3[q], v1, v2, v3);
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
    int nchunk = (nx + 63) >> 6   /* computed width, not the row pitch: the pitch may carry padding */;
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
            float a = P1[q], b = P2[q], c = P3[q];
            fgp_p_of(a, b, c, multip, v1, v2, v3);
            P1[q] = a; P2[q] = b; P3[q] = c;
        }
    }
}

// ---- fused FGP iteration (single slab): D = max(0, A - lambda div P) is NOT written, only P_new ---------------
// The reference runs Obj, nonneg, Grad, Proj as four full-volume kernels per iteration (tv_fgp.cu:244-268,
// ~80 B/voxel); the two-kernel form above moves 48 B/voxel.  Here one kernel per iteration reads A and P (16 B),
// rebuilds D for the pixel rows y and y+1 in LDS and writes P_new (12 B): 28 B/voxel.  P is ping-ponged because a
// neighbouring workgroup still needs the old values of this workgroup's border voxels.  Boundaries are the
// reference's: lower neighbours of the first slice/row/column and upper differences at the last are zero.
// Slab-sharded use (FgpEdge): an interior slab face is not a boundary.  D of the neighbour's first slice (needed by the
// slice difference of this slab's last slice) is rebuilt here from that slice's A, P1, P2, P3 planes (hi, 4 planes) and
// this slab's own last P1; D of this slab's first slice takes P1 of the neighbour's last slice (p1_lo).  The pass also
// leaves P_new of its first slice (planes 1..3 of send_first; plane 0 = A's first slice, packed once per call) and P1_new
// of its last slice (send_last): exactly what the ring exchange before the next iteration sends -- one exchange per
// iteration instead of the two of the Obj / Grad pair (tv_fgp.cu:57,81; mpi_ctvlib.cpp:400-422).
struct FgpEdge { const float *p1_lo; const float *hi; float *send_first; float *send_last; int first, last; };

template <bool SHARDED>
__global__ __launch_bounds__(256) void k_fgp_fused(const float *__restrict__ A, const float *__restrict__ P1i,
                                                    const float *__restrict__ P2i, const float *__restrict__ P3i,
                                                    float *__restrict__ P1o, float *__restrict__ P2o,
                                                    float *__restrict__ P3o, float lambda, float multip, int n, int nx,
                                                    int sx, int yseg, int zero_p, FgpEdge ed)
{
    // zero_p: first iteration of a call, P = 0 is known and neither zero-filled beforehand nor read here
    __shared__ float pl[3][2][TVL_TZ + 2][TVL_PITCH];     // P1,P2,P3 planes (parity ring); row zi = column z0-1+zi
    __shared__ float al[TVL_TZ + 2][TVL_PITCH];           // A plane being turned into D
    __shared__ float dl[2][TVL_TZ + 1][TVL_PITCH];        // D planes: row zi' = column z0+zi', element si' = slice s0+si'
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nzb = (n + TVL_TZ - 1) / TVL_TZ, nchunk = (nx + 63) >> 6   /* computed width, not the row pitch: the pitch may carry padding */, nys = (n + yseg - 1) / yseg;
    // one workgroup per (y segment, z block, chunk); XCD-aware like k_tv_grad_reg: an XCD (workgroups b, b+8, ...) owns a
    // contiguous slab of z blocks and walks it chunk-fastest, so the halo columns / slices two neighbours share are fetched
    // once per L2 (PMC, round 2, blockIdx-ordered z blocks: reads 1.77x compulsory)
    int bz, bs, ysi;
    if ((nzb & 7) == 0) {
        const int zpx = nzb >> 3;
        const int64_t li = blockIdx.x >> 3;
        bs = (int)(li % nchunk); bz = (int)(blockIdx.x & 7) * zpx + (int)((li / nchunk) % zpx); ysi = (int)(li / ((int64_t)nchunk * zpx));
    } else {
        bs = (int)(blockIdx.x % nchunk); bz = (int)((blockIdx.x / nchunk) % nzb); ysi = (int)(blockIdx.x / ((int64_t)nchunk * nzb));
    }
    if (ysi >= nys) return;
    const int y0 = ysi * yseg, y1 = min(y0 + yseg, n);
    const int z0 = bz * TVL_TZ, s0 = bs * 64;
    const size_t npix = (size_t)n * n;
    // fid: 0 = A, 1..3 = P1..P3 (the plane order of the hi / send_first buffers)
    auto ld = [&](const float *__restrict__ f, int fid, int y, int zi, int si) -> float {
        int z = z0 - 1 + zi, s = s0 - 1 + si;
        if (!SHARDED) {      // single slab: every face is a global edge (the form measured at 821 us per iteration)
            if (y < 0 || y >= n || z < 0 || z >= n || s < 0 || s >= nx) return 0.f;
            return f[(size_t)(y * n + z) * sx + s];
        }
        if (y < 0 || y >= n || z < 0 || z >= n) return 0.f;
        if (s < 0) return (fid == 1 && !ed.first && s == -1) ? ed.p1_lo[y * n + z] : 0.f;
        if (s >= nx) return (!ed.last && s == nx) ? ed.hi[fid * npix + y * n + z] : 0.f;
        return f[(size_t)(y * n + z) * sx + s];
    };
    // a pixel row (A and the three P fields, with halo) travels global -> registers -> LDS; the fetch of row
    // y+2 is issued a full iteration before it is needed
    float rg[4][3], rh[4];
    auto fetch = [&](int y) {
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            int r = wave + 4 * t;
            bool ok = r < TVL_TZ + 2;
            const bool okp = ok && !zero_p;
            rg[0][t] = okp ? ld(P1i, 1, y, r, lane + 1) : 0.f;
            rg[1][t] = okp ? ld(P2i, 2, y, r, lane + 1) : 0.f;
            rg[2][t] = okp ? ld(P3i, 3, y, r, lane + 1) : 0.f;
            rg[3][t] = ok ? ld(A, 0, y, r, lane + 1) : 0.f;
        }
        if (wave == 3 && lane < 2 * (TVL_TZ + 2)) {
            int r = lane >> 1, si = (lane & 1) ? 65 : 0;
            rh[0] = zero_p ? 0.f : ld(P1i, 1, y, r, si); rh[1] = zero_p ? 0.f : ld(P2i, 2, y, r, si);
            rh[2] = zero_p ? 0.f : ld(P3i, 3, y, r, si); rh[3] = ld(A, 0, y, r, si);
        }
    };
    auto stash = [&](int par) {
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            int r = wave + 4 * t;
            if (r < TVL_TZ + 2) {
                pl[0][par][r][lane + 1] = rg[0][t]; pl[1][par][r][lane + 1] = rg[1][t];
                pl[2][par][r][lane + 1] = rg[2][t]; al[r][lane + 1] = rg[3][t];
            }
        }
        if (wave == 3 && lane < 2 * (TVL_TZ + 2)) {
            int r = lane >> 1, si = (lane & 1) ? 65 : 0;
            pl[0][par][r][si] = rh[0]; pl[1][par][r][si] = rh[1]; pl[2][par][r][si] = rh[2]; al[r][si] = rh[3];
        }
    };
    // D of the row staged in slot `par` (its -y neighbour row of P2 is in slot par^1)
    auto compute_d = [&](int y, int par) {
        for (int e = threadIdx.x; e < (TVL_TZ + 1) * 65; e += 256) {
            int zq = e / 65, sq = e - zq * 65;
            int zi = zq + 1, si = sq + 1;
            float v1 = pl[0][par][zi][si - 1];                       // P1(s-1): zero-loaded below slice 0
            float v2 = y > 0 ? pl[1][par ^ 1][zi][si] : 0.f;         // P2(y-1)
            float v3 = pl[2][par][zi - 1][si];                       // P3(z-1): zero-loaded left of column 0
            dl[par][zq][sq] = fgp_d_of(al[zi][si], lambda, pl[0][par][zi][si], pl[1][par][zi][si], pl[2][par][zi][si], v1, v2, v3);
        }
    };
    fetch(y0 - 1); stash((y0 + 1) & 1);        // only P2(y0-1) is used
    __syncthreads();
    fetch(y0); stash(y0 & 1);
    fetch(y0 + 1);
    __syncthreads();
    compute_d(y0, y0 & 1);
    __syncthreads();
    for (int y = y0; y < y1; ++y) {
        int par = y & 1, nxt = par ^ 1;
        float keep[2][3];
#pragma unroll
        for (int q = 0; q < 2; ++q) {                                // old P of this thread's outputs
            int zi = 1 + wave * 2 + q, si = lane + 1;
            keep[q][0] = pl[0][par][zi][si]; keep[q][1] = pl[1][par][zi][si]; keep[q][2] = pl[2][par][zi][si];
        }
        stash(nxt);                                                  // row y+1 replaces row y-1 (and A of row y)
        if (y + 1 < y1) fetch(y + 2);
        __syncthreads();
        compute_d(y + 1, nxt);
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            int zq = wave * 2 + q, sq = lane;
            int z = z0 + zq, s = s0 + sq;
            if (z < n && s < nx) {
                float dc = dl[par][zq][sq];
                float v1 = (s + 1 < nx || (SHARDED && !ed.last)) ? dc - dl[par][zq][sq + 1] : 0.f;
                float v2 = y + 1 < n ? dc - dl[nxt][zq][sq] : 0.f;
                float v3 = z + 1 < n ? dc - dl[par][zq + 1][sq] : 0.f;
                float a = keep[q][0], b = keep[q][1], c = keep[q][2];
                fgp_p_of(a, b, c, multip, v1, v2, v3);
                size_t o = (size_t)(y * n + z) * sx + s;
                nt_st<256>(a, P1o + o); nt_st<256>(b, P2o + o); nt_st<256>(c, P3o + o);
                if (SHARDED) {
                    const size_t pix = (size_t)y * n + z;
                    if (s == 0) { ed.send_first[npix + pix] = a; ed.send_first[2 * npix + pix] = b; ed.send_first[3 * npix + pix] = c; }
                    if (s == nx - 1) ed.send_last[pix] = a;
                }
            }
        }
        __syncthreads();
    }
}

// ---- TWO fused FGP iterations per pass (single slab; round 4) -------------------------------------------------------------------
// k_fgp_fused moves 28 B per voxel and iteration (A and P in, P_new out) and is bound by exactly that (0.72 ms at the 5.2 TB/s a
// read + write stream gets, against ~0.4 ms of arithmetic).  Here a workgroup carries P through two iterations before it stores it:
// per pixel row it rebuilds D^k on its tile + 2 halo cells, P^(k+1) on tile + 1 (the halo cells are recomputed, not exchanged:
// (TZ+2)(64+2) / (TZ 64) = 1.29 x the tile), D^(k+1), and stores P^(k+2) of the tile: 28 B per voxel for TWO iterations, against
// 2.27 x the arithmetic of one.  Every value is computed by the expressions of k_fgp_fused on the same operands, in the same order:
// the result equals two passes of it bit for bit.  Rows travel global -> registers -> LDS one iteration ahead; rings of two rows.
//   needs, for the stored row y:   D1(y), D1(y+1)  <-  P1(y), P1(y+1), P1_2(y-1)  <-  D0(y) .. D0(y+2)  <-  P0(y-1 .. y+2), A
#ifndef F2_TZ_V
#define F2_TZ_V 8
#endif
#ifndef F2_SC_V
#define F2_SC_V 32
#endif
constexpr int F2_TZ = F2_TZ_V, F2_R = F2_TZ + 4, F2_SC = F2_SC_V, F2_S = F2_SC + 4;   // columns x slices of a tile (8 x 32: 31 KB of LDS, five workgroups per CU;
                                                                                        // 8 x 64 = 59 KB, two per CU, ran at 880 us per iteration against 604)   // rows zi = column z0-2+zi, elements si = slice s0-2+si

// FINAL: the call's last pass -- one iteration and then D = max(0, A - lambda div P) of the result, which is all the last iteration
// of tv_fgp.cu needs (:272): D^(k+1) of the tile goes to P1o (a scratch volume: A's halo cells are other tiles' outputs, so the
// result cannot land on A in place; the engine swaps the buffers), P^(k+1) is never stored.
template <bool FINAL>
__global__ __launch_bounds__(256) void k_fgp_fused2(const float *__restrict__ A, const float *__restrict__ P1i,
                                                     const float *__restrict__ P2i, const float *__restrict__ P3i,
                                                     float *__restrict__ P1o, float *__restrict__ P2o, float *__restrict__ P3o,
                                                     float lambda, float multip, int n, int nx, int sx, int yseg, int zero_p)
{
    constexpr int PL = F2_R * F2_S;                     // a staged plane: element zi * F2_S + si
    __shared__ float pk[3][2][PL];                      // P^k, rows r (slot r & 1)
    __shared__ float ak[2][PL];                         // A
    __shared__ float dk[2][PL];                         // D^k
    __shared__ float pn[3][2][PL];                      // P^(k+1)
    __shared__ float dn[2][PL];                         // D^(k+1)
    const int tid = threadIdx.x;
    const int nzb = (n + F2_TZ - 1) / F2_TZ, nchunk = (nx + F2_SC - 1) / F2_SC, nys = (n + yseg - 1) / yseg;
    int bz, bs, ysi;                                    // the item map of k_fgp_fused
    if ((nzb & 7) == 0) {
        const int zpx = nzb >> 3;
        const int64_t li = blockIdx.x >> 3;
        bs = (int)(li % nchunk); bz = (int)(blockIdx.x & 7) * zpx + (int)((li / nchunk) % zpx); ysi = (int)(li / ((int64_t)nchunk * zpx));
    } else {
        bs = (int)(blockIdx.x % nchunk); bz = (int)((blockIdx.x / nchunk) % nzb); ysi = (int)(blockIdx.x / ((int64_t)nchunk * nzb));
    }
    if (ysi >= nys) return;
    const int y0 = ysi * yseg, y1 = min(y0 + yseg, n);
    const int z0 = bz * F2_TZ, s0 = bs * F2_SC;
    // Every phase works on a rectangle of the staged plane, 256 elements a round; which elements are this thread's, where they sit
    // and what the volume's faces make of them does not change along y: worked out once.
    //   staged row (fetch / stash): zi 0 .. R-1, si 0 .. S-1       D^k:      zi 1 .. R-1, si 1 .. S-1
    //   P^(k+1):                    zi 1 .. R-2, si 1 .. S-2       D^(k+1):  zi 2 .. R-2, si 2 .. S-2      output: zi 2 .. R-3, si 2 .. S-3
    constexpr int NT = (PL + 255) / 256;
    constexpr int ND = (F2_R - 1) * (F2_S - 1), RD = (ND + 255) / 256;
    constexpr int NP = (F2_R - 2) * (F2_S - 2), RP = (NP + 255) / 256;
    constexpr int NN = (F2_R - 3) * (F2_S - 3), RN = (NN + 255) / 256;
    static_assert(F2_TZ * F2_SC == 256, "one output per thread and row");
    int eo[NT]; size_t eg[NT]; bool einv[NT];           // staged element: plane offset (-1: none), offset in a volume row, inside in z and s
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int e = tid + 256 * t, zi = e / F2_S, si = e - zi * F2_S;
        const int z = z0 - 2 + zi, s = s0 - 2 + si;
        eo[t] = e < PL ? e : -1;
        einv[t] = e < PL && z >= 0 && z < n && s >= 0 && s < nx;
        eg[t] = einv[t] ? (size_t)z * sx + s : 0;
    }
    int od[RD], op[RP], on[RN];
    unsigned pf_[RP];                                   // P^(k+1) element: bit 0 inside the volume in z and s, bit 1 s+1 < nx, bit 2 z+1 < n
#pragma unroll
    for (int r = 0; r < RD; ++r) { const int e = tid + 256 * r, zq = e / (F2_S - 1); od[r] = e < ND ? (zq + 1) * F2_S + (e - zq * (F2_S - 1)) + 1 : -1; }
#pragma unroll
    for (int r = 0; r < RP; ++r) {
        const int e = tid + 256 * r, zq = e / (F2_S - 2), zi = zq + 1, si = e - zq * (F2_S - 2) + 1;
        const int z = z0 - 2 + zi, s = s0 - 2 + si;
        op[r] = e < NP ? zi * F2_S + si : -1;
        pf_[r] = (z >= 0 && z < n && s >= 0 && s < nx ? 1u : 0u) | (s + 1 < nx ? 2u : 0u) | (z + 1 < n ? 4u : 0u);
    }
#pragma unroll
    for (int r = 0; r < RN; ++r) { const int e = tid + 256 * r, zq = e / (F2_S - 3); on[r] = e < NN ? (zq + 2) * F2_S + (e - zq * (F2_S - 3)) + 2 : -1; }
    const int ozi = 2 + tid / F2_SC, osi = 2 + tid % F2_SC, oo = ozi * F2_S + osi;
    const int oz = z0 - 2 + ozi, os = s0 - 2 + osi;
    const bool oin = oz < n && os < nx, os1 = os + 1 < nx, oz1 = oz + 1 < n;
    const size_t og = (size_t)oz * sx + os;
    float rg[4][NT];
    auto fetch = [&](int y) {
        const bool yin = y >= 0 && y < n;
        const size_t row = yin ? (size_t)y * n * sx : 0;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const bool ok = yin && einv[t];
            const size_t o = row + eg[t];
            rg[0][t] = (ok && !zero_p) ? P1i[o] : 0.f;
            rg[1][t] = (ok && !zero_p) ? P2i[o] : 0.f;
            rg[2][t] = (ok && !zero_p) ? P3i[o] : 0.f;
            rg[3][t] = ok ? A[o] : 0.f;
        }
    };
    auto stash = [&](int par) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
            if (eo[t] >= 0) { pk[0][par][eo[t]] = rg[0][t]; pk[1][par][eo[t]] = rg[1][t]; pk[2][par][eo[t]] = rg[2][t]; ak[par][eo[t]] = rg[3][t]; }
    };
    // D of row r from the P fields `pf` (pk or pn) at this thread's elements `off` (od or on) into `df`
#define F2_D(pf, df, r, off, NRND)                                                                        \
    {                                                                                                     \
        const int par = (r) & 1;                                                                          \
        _Pragma("unroll") for (int q = 0; q < NRND; ++q) {                                                \
            const int o = off[q];                                                                         \
            if (o >= 0) {                                                                                 \
                float v1 = pf[0][par][o - 1];                                                             \
                float v2 = (r) > 0 ? pf[1][par ^ 1][o] : 0.f;                                             \
                float v3 = pf[2][par][o - F2_S];                                                          \
                df[par][o] = fgp_d_of(ak[par][o], lambda, pf[0][par][o], pf[1][par][o], pf[2][par][o], v1, v2, v3); \
            }                                                                                             \
        }                                                                                                 \
    }
    // P^(k+1) of row r (zero outside the volume, as a load of it would give)
    auto compute_pn = [&](int r) {
        const int par = r & 1;
        const bool rin = r >= 0 && r < n, r1 = r + 1 < n;
#pragma unroll
        for (int q = 0; q < RP; ++q) {
            const int o = op[q];
            if (o >= 0) {
                float a = 0.f, b = 0.f, c = 0.f;
                if (rin && (pf_[q] & 1u)) {
                    a = pk[0][par][o]; b = pk[1][par][o]; c = pk[2][par][o];
                    const float dc = dk[par][o];
                    const float v1 = (pf_[q] & 2u) ? dc - dk[par][o + 1] : 0.f;
                    const float v2 = r1 ? dc - dk[par ^ 1][o] : 0.f;
                    const float v3 = (pf_[q] & 4u) ? dc - dk[par][o + F2_S] : 0.f;
                    fgp_p_of(a, b, c, multip, v1, v2, v3);
                }
                pn[0][par][o] = a; pn[1][par][o] = b; pn[2][par][o] = c;
            }
        }
    };
    fetch(y0 - 2); stash(y0 & 1);                       // only P2(y0-2) is used
    __syncthreads();
    fetch(y0 - 1); stash((y0 - 1) & 1);
    fetch(y0);
    __syncthreads();
    F2_D(pk, dk, y0 - 1, od, RD)
    __syncthreads();
    stash(y0 & 1);                                      // row y0 replaces row y0-2
    fetch(y0 + 1);
    __syncthreads();
    F2_D(pk, dk, y0, od, RD)
    __syncthreads();
    compute_pn(y0 - 1);
    __syncthreads();
    stash((y0 + 1) & 1);                                // row y0+1 replaces row y0-1 (P, A) ...
    fetch(y0 + 2);
    __syncthreads();
    F2_D(pk, dk, y0 + 1, od, RD)                        // ... and its D
    __syncthreads();
    compute_pn(y0);
    __syncthreads();
    F2_D(pn, dn, y0, on, RN)
    __syncthreads();
    for (int y = y0; y < y1; ++y) {
        const int par = y & 1, nxt = par ^ 1;
        stash(par);                                     // row y+2 replaces row y (P^k, A)
        if (y + 1 < y1) fetch(y + 3);
        __syncthreads();
        F2_D(pk, dk, y + 2, od, RD)                     // D^k(y+2) replaces D^k(y)
        __syncthreads();
        compute_pn(y + 1);                              // P^(k+1)(y+1) replaces P^(k+1)(y-1)
        __syncthreads();
        F2_D(pn, dn, y + 1, on, RN)                     // D^(k+1)(y+1) replaces D^(k+1)(y-1)
        __syncthreads();
        if (FINAL) {
            if (oin) nt_st<256>(dn[par][oo], P1o + (size_t)y * n * sx + og);
        } else if (oin) {                               // the tile's 256 outputs of this row
            float a = pn[0][par][oo], b = pn[1][par][oo], c = pn[2][par][oo];
            const float dc = dn[par][oo];
            const float v1 = os1 ? dc - dn[par][oo + 1] : 0.f;
            const float v2 = y + 1 < n ? dc - dn[nxt][oo] : 0.f;
            const float v3 = oz1 ? dc - dn[par][oo + F2_S] : 0.f;
            fgp_p_of(a, b, c, multip, v1, v2, v3);
            const size_t o = (size_t)y * n * sx + og;
            nt_st<256>(a, P1o + o); nt_st<256>(b, P2o + o); nt_st<256>(c, P3o + o);
        }
        // (the next iteration's stash / D^k / P^(k+1) phases write slots this phase does not read; its D^(k+1) phase, which does,
        // comes behind three barriers)
    }
#undef F2_D
}

}  // namespace tomo
