// kernels_sart.hip.h -- the streamed SART / ART sweeps: ray-walk and tile forms of the fused step, residual finish, Kaczmarz kernels
// Part of kernels.hip.h (include that, not this: the families share helpers and constants in the order kernels.hip.h lists them).
#pragma once

namespace tomo {

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
        xv[J] = (y < n && z0 + J < n) ? st_xload<NT>(reinterpret_cast<const V *>(x_old + ((size_t)y * n + z0 + J) * sx + off)) : vzero<4>();
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
    } else if (FUSED) {
        uint32_t w = wins[tile];
        for (int i = t; i < (ST_MAXR + 1) * 16; i += ST_THREADS) {
            int j = i >> 4;
            win[i] = ((uint32_t)j < (w >> 16)) ? *reinterpret_cast<const V *>(r_prev + ((size_t)(w & 0xFFFFu) + j) * sx + c * 64 + (i & 15) * 4) : vzero<4>();
        }
        if (t < ST_PIX) cel[t] = cells[(size_t)tile * ST_PIX + t];
    }
    if (t < 16) img[ST_PIX * 16 + t] = vzero<4>();
    if (FUSED) {
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
            if (y < n && z0 + J < n && wr) st_xstore<NT>(nv, reinterpret_cast<V *>(x_new + ((size_t)y * n + z0 + J) * sx + off));
        }
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

}  // namespace tomo
