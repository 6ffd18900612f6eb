// kernels_fgp.hip.h -- FGP-TV: Obj / Grad pair, one and two fused iterations per pass
// Part of kernels.hip.h (include that, not this: the families share helpers and constants in the order kernels.hip.h lists them).
#pragma once

namespace tomo {

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
//
// SHARDED (round 6): the slab-sharded form.  Two iterations reach two slices across an interior slab face -- P^(k+2)(s) <- D^(k+1)(s),
// D^(k+1)(s+1) <- P^(k+1)(s-1 .. s+1) <- D^k(s-1 .. s+2) <- P^k(s-2 .. s+2), A(s-1 .. s+2) -- so the halo is two slices deep and is
// exchanged once per TWO iterations (tv_fgp.cu:57,81 name the neighbours; mpi_ctvlib.cpp:400-422 the ring):
//   lo  (5 planes, from the slab below)   [P1(-1), A(-1), P2(-1), P3(-1), P1(-2)]
//   hi  (8 planes, from the slab above)   [A, P1, P2, P3](nx), [A, P1, P2, P3](nx + 1)
//   send_first (8 planes, my slices 0, 1) [A, P1, P2, P3](0), [A, P1, P2, P3](1)             -> the lower neighbour's hi
//   send_last  (5 planes)                 [P1(nx-1), A(nx-1), P2(nx-1), P3(nx-1), P1(nx-2)]  -> the upper neighbour's lo
// The planes the one-iteration form k_fgp_fused uses (P1(-1); [A, P1, P2, P3](nx); its send planes) are the PREFIXES of these, so one
// set of buffers serves both.  The pass stores the P planes of its send buffers (the A planes are packed once per call).  Every slab
// must hold at least two slices (the host checks); the halo cells are recomputed here, like the tile's own ring, by the same helpers
// on the same operands: bit-identical to two one-iteration passes with an exchange between them.
struct Fgp2Edge { const float *lo; const float *hi; float *send_first; float *send_last; int first, last; };

template <bool FINAL, bool SHARDED = false>
__global__ __launch_bounds__(256) void k_fgp_fused2(const float *__restrict__ A, const float *__restrict__ P1i,
                                                     const float *__restrict__ P2i, const float *__restrict__ P3i,
                                                     float *__restrict__ P1o, float *__restrict__ P2o, float *__restrict__ P3o,
                                                     float lambda, float multip, int n, int nx, int sx, int yseg, int zero_p,
                                                     Fgp2Edge ed = Fgp2Edge{})
{
    static_assert(!(FINAL && SHARDED), "the sharded odd iteration out runs as k_fgp_fused + k_fgp_obj");
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
    // staged element: plane offset (-1: none); where it comes from (ek: 0 the volume, 1 / 2 the slices -1 / -2 of the slab below,
    // 3 / 4 the slices nx / nx + 1 of the slab above, -1 outside the global volume: zero); its offset in a volume row or its column
    int eo[NT]; size_t eg[NT]; int ek[NT];
    const size_t npix = (size_t)n * n;
    const bool lo_face = SHARDED && !ed.first, hi_face = SHARDED && !ed.last;      // interior slab faces
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int e = tid + 256 * t, zi = e / F2_S, si = e - zi * F2_S;
        const int z = z0 - 2 + zi, s = s0 - 2 + si;
        eo[t] = e < PL ? e : -1;
        int kind = -1;
        if (e < PL && z >= 0 && z < n) {
            if (s >= 0 && s < nx) kind = 0;
            else if (lo_face && s == -1) kind = 1;
            else if (lo_face && s == -2) kind = 2;
            else if (hi_face && s == nx) kind = 3;
            else if (hi_face && s == nx + 1) kind = 4;
        }
        ek[t] = kind;
        eg[t] = kind == 0 ? (size_t)z * sx + s : (kind > 0 ? (size_t)z : 0);
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
        // (sharded: slice -1 of the slab below and slice nx of the slab above are cells of the global volume too, and their upper
        // neighbours exist -- every slab holds at least two slices)
        const bool s_in = (s >= 0 && s < nx) || (lo_face && s == -1) || (hi_face && s == nx);
        const bool s_up = s + 1 < nx || (hi_face && s + 1 <= nx + 1);
        pf_[r] = (z >= 0 && z < n && s_in ? 1u : 0u) | (s_up ? 2u : 0u) | (z + 1 < n ? 4u : 0u);
    }
#pragma unroll
    for (int r = 0; r < RN; ++r) { const int e = tid + 256 * r, zq = e / (F2_S - 3); on[r] = e < NN ? (zq + 2) * F2_S + (e - zq * (F2_S - 3)) + 2 : -1; }
    const int ozi = 2 + tid / F2_SC, osi = 2 + tid % F2_SC, oo = ozi * F2_S + osi;
    const int oz = z0 - 2 + ozi, os = s0 - 2 + osi;
    const bool oin = oz < n && os < nx, os1 = os + 1 < nx || hi_face, oz1 = oz + 1 < n;
    const size_t og = (size_t)oz * sx + os;
    float rg[4][NT];
    auto fetch = [&](int y) {
        const bool yin = y >= 0 && y < n;
        const size_t row = yin ? (size_t)y * n * sx : 0;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const bool ok = yin && ek[t] == 0;
            const size_t o = row + eg[t];
            rg[0][t] = (ok && !zero_p) ? P1i[o] : 0.f;
            rg[1][t] = (ok && !zero_p) ? P2i[o] : 0.f;
            rg[2][t] = (ok && !zero_p) ? P3i[o] : 0.f;
            rg[3][t] = ok ? A[o] : 0.f;
            if (SHARDED && yin && ek[t] > 0) {          // a cell of a neighbouring slab: from the exchanged planes
                const size_t pix = (size_t)y * n + eg[t];
                const int kd = ek[t];
                if (kd == 1) {
                    rg[3][t] = ed.lo[npix + pix];
                    if (!zero_p) { rg[0][t] = ed.lo[pix]; rg[1][t] = ed.lo[2 * npix + pix]; rg[2][t] = ed.lo[3 * npix + pix]; }
                } else if (kd == 2) {
                    if (!zero_p) rg[0][t] = ed.lo[4 * npix + pix];          // only P1(-2) is ever used (by D^k(-1))
                } else {
                    const float *hp = ed.hi + (kd == 3 ? 0 : 4) * npix + pix;
                    rg[3][t] = hp[0];
                    if (!zero_p) { rg[0][t] = hp[npix]; rg[1][t] = hp[2 * npix]; rg[2][t] = hp[3 * npix]; }
                }
            }
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
            if (SHARDED) {                              // what the next exchange sends (either depth)
                const size_t pix = (size_t)y * n + oz;
                if (os == 0) { ed.send_first[npix + pix] = a; ed.send_first[2 * npix + pix] = b; ed.send_first[3 * npix + pix] = c; }
                if (os == 1) { ed.send_first[5 * npix + pix] = a; ed.send_first[6 * npix + pix] = b; ed.send_first[7 * npix + pix] = c; }
                if (os == nx - 1) { ed.send_last[pix] = a; ed.send_last[2 * npix + pix] = b; ed.send_last[3 * npix + pix] = c; }
                if (os == nx - 2) ed.send_last[4 * npix + pix] = a;
            }
        }
        // (the next iteration's stash / D^k / P^(k+1) phases write slots this phase does not read; its D^(k+1) phase, which does,
        // comes behind three barriers)
    }
#undef F2_D
}

}  // namespace tomo
