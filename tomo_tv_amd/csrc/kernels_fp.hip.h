// kernels_fp.hip.h -- forward projectors: ray-driven, tile-stationary, sheared strips and their list form, the reduce kernel, sinogram passes
// Part of kernels.hip.h (include that, not this: the families share helpers and constants in the order kernels.hip.h lists them).
#pragma once

namespace tomo {

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
                v[i] = ok ? nt_ld<16>(reinterpret_cast<const V *>(xc + pix * sx)) : vzero<4>();
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
#define FS_NEXT                                                                                           \
        FS_DMA(slot)           /* this unit's slot is free (its entry is in registers): fetch the unit FS_RING ahead into it */ \
        slot = slot == FS_RING - 1 ? 0u : slot + 1u;                                                      \
        /* the NEXT unit's entry: its DMA is the oldest of the FS_RING now in flight */                    \
        asm volatile("s_waitcnt vmcnt(%1)\n\tds_read_b64 %0, %2" : "=v"(en) : "n"(FS_RING - 1), "v"(ring_l + slot * 256u) : "memory");
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
            if (e.x >> 31) {   /* the ray leaves the strip: its sum is this group's next partial sum */ \
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
// (register assumptions as for BL_CLOBBERS, kernels_bp.hip.h: bound by number, verified on ROCm 7.2.0 / clang 20 / gfx950; the bit-identity
// tests against k_fp_strip / k_fp_tile are the check for another toolchain)
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

}  // namespace tomo
