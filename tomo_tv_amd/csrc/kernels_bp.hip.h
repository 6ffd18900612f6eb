// kernels_bp.hip.h -- back projectors: one angle (the SART update), all angles pixel-driven, tile-stationary, entry lists
// Part of kernels.hip.h (include that, not this: the families share helpers and constants in the order kernels.hip.h lists them).
#pragma once

namespace tomo {

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
// Register assumptions of the asm block below (v[32:127] and s[33], s[36:99] bound by NUMBER, amdgpu_waves_per_eu(4, 4)): written against and
// verified on ROCm 7.2.0 (hipcc = AMD clang 20, gfx950).  A compiler that needs one of these registers for a kernel argument or a
// live value fails to BUILD (constraint conflict) rather than miscompile; tests/test_gpu_parity.py holds the kernel bit-identical to k_bp_all.
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
               int nproj, int n, int sx, int tiles_z, int ntiles, int nchunk2, int band)
{
    typedef VecOf<4>::T V;
    extern __shared__ V bl_lds[];
    const int xcd = blockIdx.x & 7, l = blockIdx.x >> 3;
    // band = 0: consecutive tiles go round the eight XCDs (rounds 4: a tile's neighbours sit on seven other L2s);
    // band = 1 (round 5): an XCD owns a contiguous band of tiles, so the tiles that share residual rows -- a ray
    // crosses neighbouring tiles -- find them in ONE L2
    const int tpx = (ntiles + 7) >> 3;
    const int tile = band ? xcd * tpx + l / nchunk2 : (l / nchunk2) * 8 + xcd, c2 = l % nchunk2;
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

}  // namespace tomo
