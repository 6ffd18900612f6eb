// sart_resident.hip.h -- the volume-RESIDENT SART sweep (round 5).  Included by tomo_engine.hip after kernels.hip.h.
//
// k_sart_tile streams the slab once per angle: 8 B per voxel and angle at the read + write ceiling of the memory system.  Here a
// 64-slice chunk of the WHOLE image stays in vector registers for all angles of a sweep (ASTRA's SART as the reference calls it,
// tomoengine.cpp:162-179: one forward projection, one residual, one clamped back-projection per angle, angles in sequence):
//   * one workgroup of 16 waves per 32 x 32-pixel tile, one workgroup per CU (N <= 512: at most 256 tiles); wave w holds the
//     8 x 8 block (w / 4, w % 4) of the tile in v[64:127], one register per pixel, lane = slice;
//   * back-projection of angle a:  x[q] = max(0, x[q] + beta * ((w0 r[s0] + w1 r[s0+1]) * inv))   -- the expression of k_bp_angle,
//     rounding for rounding; the (at most 14) residual rows that cross the block sit in v[32:45], picked through the VGPR index
//     mode (s_set_gpr_idx: M0[7:0] is added to the register number of the operands the mode names), the cell {s0, w0, w1, inv}
//     is SCALAR data (s_load_dwordx16: four pixels per load).  A pixel's two rays of an angle are neighbours, so ONE index per
//     pixel serves both: the second operand is written one register higher;
//   * forward projection of angle a+1:  acc[s0] += w0 x[q]; acc[s0+1] += w1 x[q]  with the pixel STATIC in the instruction and the
//     ray's accumulator (v[32:45] again) picked through the index mode -- no LDS image, no lane moves; the SAME cells;
//   * exchange, per angle: the workgroup adds its waves' block sums per ray of the tile's window (LDS, fixed order) and publishes
//     them; a ray's reducer (16 / rpt waves of tile j / rpt) adds the tile sums in ascending tile order, forms
//     r = (b - sum) / rowsum (k_resid_finish's expression) and publishes the row; every workgroup picks up the <= 48 rows of its window.
//     Everything published travels as 8-byte {value, tag} granules, one agent-scope store per lane (write-through), polled by
//     agent-scope loads until the tag is this step's epoch: the data is the flag (cdna_hip_programming.md, Guideline 16 R2) -- no
//     drain, no separate flag, no fence, and a stale line can only show an older tag.
//   * buffers: tile sums single-buffered per (tile, window slot) -- a tile writes slot i for angle a+1 only after it has consumed
//     the residual rows of angle a, which the reducers formed from its sums of angle a; residual rows per (ANGLE, ray) -- the tiles
//     that trigger a rewrite (they sit on the ray of that angle) are exactly the tiles that consumed the old value.
// Every spin is bounded, and a sweep that cannot finish leaves the volume as it found it (round 6): a wave that gives up sets *abort
// (every other workgroup stops at its next look) and marks its workgroup; after the last angle the workgroups of a chunk COMMIT --
// a word per chunk {poison bit, count of clean workgroups}: a clean workgroup adds one and waits for the count to reach the number
// of tiles, one that is not clean sets the poison bit (and so does one that gives up waiting, by compare-and-swap on the value it
// saw, so that "full" and "poisoned" exclude each other; rs_commit).  Only a full, unpoisoned count lets the chunk's
// workgroups store -- all of them or none -- and tile 0 then writes the launch sequence into the chunk's word of a pinned host array.
// The host (launch_sart_resident, tomo_engine.hip) waits for the launch, reads that array and sweeps the chunks that did not commit
// with the streamed chain (k_sart_tile): the sweep either happens or the volume is untouched, as in tomoengine.cpp:162-179.
// The summation ORDER of a ray sum differs from k_sart_tile's (blocks of 8 x 8 inside tiles of 32 x 32 instead of segments of
// 16 x 16 tiles): sweeps agree to ~1e-7 relative, not to the bit; the voxel update itself is bit-identical given the same rows.
#pragma once

namespace tomo {

typedef float v16f __attribute__((ext_vector_type(16)));
typedef unsigned long long rs_u64;
typedef uint32_t rs_u2 __attribute__((ext_vector_type(2)));
constexpr int RS_T = 32, RS_WAVES = 16, RS_THREADS = RS_WAVES * 64, RS_MAXWIN = 48, RS_RL = 32, RS_USABLE = 14;
struct RsHdrD { uint16_t jbase, nrays; uint32_t dw[4]; uint32_t pad[3]; };
static_assert(sizeof(RsHdrD) == 32, "header layout (resident.h)");
struct RsArgs {
    float *x;                       // the swept volume [pixel][sx]
    const float *b, *rowsum;        // measured rows [row][sx]; row sums
    const RsHdrD *hdr;
    const uint4 *cell;              // [angle][tile][wave][pixel] {s0, w0, w1, inv}
    const uint2 *ts;                // [angle][tile][window ray] eight block sums wave << 4 | slot
    const uint16_t *rl;
    rs_u64 *pb, *rb;                // granules: tile sums [group][tile][RS_MAXWIN][64], residual rows [group][angle][ray][64]
    const int *angs;                // angle of step k (steps entries)
    float *track;                   // nullptr, or the volume that receives a copy of the result (the sum of squared differences goes to part)
    double *part;
    int *abort_word, *abort_host;   // device word every spin looks at; pinned host word the host looks at (both set by the wave that gives up)
    unsigned *commit;               // [64-slice chunk of the slab] poison << 31 | workgroups that finished a sweep of the chunk clean (since cleared)
    int *done_host;                 // pinned [chunk]: the sequence number of the launch whose workgroups stored the chunk
    unsigned seq, commit_base;      // this launch; what every commit word held when it started
    int test_fail;                  // tests: chunk + 1 whose tile 0 declares itself not clean at the commit (0 = none)
    int n, sx, np, ntiles, tiles, rpt, steps, chunk0, nchunk;
    unsigned epoch0, spin_limit;
    float beta;
    long long *prof;                // measurement builds (-DRS_PROF): 8 phase totals per workgroup, in ticks of s_memrealtime (100 MHz)
};
#ifdef RS_PROF
#define RS_STAMP(q) { const long long now_ = (long long)__builtin_amdgcn_s_memrealtime(); if (wave == 0) tacc[q] += now_ - tprev; tprev = now_; }
#define RS_TL(q) { if (k == 40 && blockIdx.x == 37 && lane == 0) rs_args()->prof[2048 + wave * 8 + (q)] = (long long)__builtin_amdgcn_s_memrealtime(); }
#else
#define RS_STAMP(q)
#define RS_TL(q)
#endif

__device__ __forceinline__ rs_u64 rs_gld(const rs_u64 *p)
{
    return __hip_atomic_load((const __attribute__((address_space(1))) rs_u64 *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void rs_gst(rs_u64 *p, float v, unsigned ep)
{
    __hip_atomic_store((__attribute__((address_space(1))) rs_u64 *)p, ((rs_u64)ep << 32) | (rs_u64)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int rs_abort_ld(const int *p)
{
    return __hip_atomic_load((const __attribute__((address_space(1))) int *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// one look at the give-up conditions every 32 polls (wave-uniform); true = stop spinning
__device__ __forceinline__ bool rs_give_up(unsigned &spins, unsigned spin_limit, int *abort_word, int *abort_host, int lane, int code)
{
    if (((++spins) & 31u) != 0u && spin_limit != 0u) return false;       // (a limit of 0: the first look that finds nothing gives up -- tests)
    if (spins > spin_limit) {
        if (lane == 0) {
            __hip_atomic_store((__attribute__((address_space(1))) int *)abort_word, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (abort_host) __hip_atomic_store((__attribute__((address_space(1))) int *)abort_host, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        return true;
    }
    return rs_abort_ld(abort_word) != 0;
}

// ---- all or nothing: the chunk's workgroups store only when every one of them finished clean (one lane per workgroup calls this) --------
// The word counts clean workgroups since the words were last cleared (bits 0..30) and carries a poison bit (31).  Every launch covers
// every chunk of the slab and a launch that commits adds exactly `ntiles` to each word, so all words start a launch at the same `base`
// (a kernel argument; after a launch that did not commit the host clears the words and starts again at 0).  The join is ONE atomic add
// (clean) or one atomic or (not clean) -- a compare-and-swap join of 256 workgroups on one word cost 0.6 ms per chunk (measured, round 6:
// 17.6 instead of 11.0 us per angle) -- and only a workgroup that gives up WAITING uses a compare-and-swap on the value it last saw,
// so that "full" and "poisoned" exclude each other.
constexpr unsigned RS_POISON = 0x80000000u;
__device__ __forceinline__ bool rs_commit(unsigned *w, unsigned base, unsigned ntiles, bool clean, unsigned spin_limit, const int *abort_word)
{
    typedef __attribute__((address_space(1))) unsigned *gp;
    if (!clean) { __hip_atomic_fetch_or((gp)w, RS_POISON, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return false; }
    unsigned seen = __hip_atomic_fetch_add((gp)w, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    const unsigned full = base + ntiles;
    for (unsigned spins = 0;;) {
        if (seen & RS_POISON) return false;
        if (seen == full) return true;
        if (++spins > spin_limit || ((spins & 31u) == 0u && rs_abort_ld(abort_word) != 0)) {
            // give up -- on the value last seen: if the count moved meanwhile (it may just have become full), look again
            unsigned expected = seen;
            if (__hip_atomic_compare_exchange_strong((gp)w, &expected, seen | RS_POISON, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return false;
            seen = expected;
            continue;
        }
        __builtin_amdgcn_s_sleep(4);
        seen = __hip_atomic_load((gp)w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---- the two static loops over a wave's 64 pixels --------------------------------------------------------------------------------
// Cells arrive 16 pixels (64 dwords = all of s[36:99]) at a time: four s_load_dwordx16, one wait, sixteen pixels of work.  A scalar
// load is ~0.2 us from the L2 whatever its size and only lgkmcnt(0) is a safe wait (scalar loads return out of order), so what
// counts is round trips per pass: double-buffered sets of 8 pixels (the first form: 8 round trips, each hidden behind 8 pixels = 0.03
// to 0.07 us of this wave's own work) took 1.4 us (back projection) and 2.1 us (forward projection) per wave and pass; one set of 16
// pixels has four, and the other three waves of the SIMD fill them (profiles/r05_resident_sweep.md).
// SB = first SGPR of the set (36), K = pixel inside the batch, Q = pixel of the block.
#define RS_LD(OFF)                                                                                        \
    "s_load_dwordx16 s[36:51], %[cp], " #OFF "\n"                                                         \
    "s_load_dwordx16 s[52:67], %[cp], " #OFF "+0x40\n"                                                    \
    "s_load_dwordx16 s[68:83], %[cp], " #OFF "+0x80\n"                                                    \
    "s_load_dwordx16 s[84:99], %[cp], " #OFF "+0xc0\n"
#define RS_IDX(SB, K) "s_set_gpr_idx_idx s[" #SB "+4*" #K "]\n"
// back-projection, cell {s0, w0, w1, inv}; index mode on SRC1; rows in v[32:47] (v46 = v47 = 0: where a pixel without rays points)
#define RS_BP1(SB, K, Q)                                                                                  \
    RS_IDX(SB, K)                                                                                         \
    "v_mul_f32 v48, s[" #SB "+4*" #K "+1], v32\n"                                                         \
    "v_fma_f32 v48, s[" #SB "+4*" #K "+2], v33, v48\n"                                                    \
    "v_mul_f32 v48, v48, s[" #SB "+4*" #K "+3]\n"                                                         \
    "v_fma_f32 v[64+" #Q "], v48, %[beta], v[64+" #Q "]\n"                                                \
    "v_max_f32 v[64+" #Q "], v[64+" #Q "], 0\n"
// forward projection, the same cell; index mode on SRC2 and DST; ray sums in v[32:47] (v46, v47: the sink of pixels without rays,
// which only ever receives 0 * x)
#define RS_FP1(SB, K, Q)                                                                                  \
    RS_IDX(SB, K)                                                                                         \
    "v_fma_f32 v32, s[" #SB "+4*" #K "+1], v[64+" #Q "], v32\n"                                           \
    "v_fma_f32 v33, s[" #SB "+4*" #K "+2], v[64+" #Q "], v33\n"
#define RS_BATCH(OP, B)                                                                                   \
    OP(36, 0, 16*B+0) OP(36, 1, 16*B+1) OP(36, 2, 16*B+2) OP(36, 3, 16*B+3) OP(36, 4, 16*B+4) OP(36, 5, 16*B+5) OP(36, 6, 16*B+6) OP(36, 7, 16*B+7) \
    OP(36, 8, 16*B+8) OP(36, 9, 16*B+9) OP(36, 10, 16*B+10) OP(36, 11, 16*B+11) OP(36, 12, 16*B+12) OP(36, 13, 16*B+13) OP(36, 14, 16*B+14) OP(36, 15, 16*B+15)
#define RS_WAIT "s_waitcnt lgkmcnt(0)\n"
#define RS_SWEEP(OP)                                                                                      \
    RS_BATCH(OP, 0) RS_LD(0x100) RS_WAIT RS_BATCH(OP, 1) RS_LD(0x200) RS_WAIT RS_BATCH(OP, 2) RS_LD(0x300) RS_WAIT RS_BATCH(OP, 3)
// Register assumptions of the asm blocks (x in v[64:127], rows / sums in v[32:47], v[48:59] scratch, s[33], s[36:99] cells; 128 VGPRs at
// amdgpu_waves_per_eu(4, 4)): written against and verified on ROCm 7.2.0 (hipcc = AMD clang 20, gfx950).  Another toolchain: a register the
// compiler needs elsewhere is a BUILD error (constraint conflict); tools/experiments/resident_probe.hip replays the kernel bit for bit against a
// CPU loop in its order of operations, tests/test_gpu_sart_resident.py holds it to the streamed chain and the oracle.
#define RS_CLOB4(P, A, B, C, D) #P #A, #P #B, #P #C, #P #D
#define RS_CLOBBERS                                                                                       \
    RS_CLOB4(s, 36, 37, 38, 39), RS_CLOB4(s, 40, 41, 42, 43), RS_CLOB4(s, 44, 45, 46, 47), RS_CLOB4(s, 48, 49, 50, 51),      \
    RS_CLOB4(s, 52, 53, 54, 55), RS_CLOB4(s, 56, 57, 58, 59), RS_CLOB4(s, 60, 61, 62, 63), RS_CLOB4(s, 64, 65, 66, 67),      \
    RS_CLOB4(s, 68, 69, 70, 71), RS_CLOB4(s, 72, 73, 74, 75), RS_CLOB4(s, 76, 77, 78, 79), RS_CLOB4(s, 80, 81, 82, 83),      \
    RS_CLOB4(s, 84, 85, 86, 87), RS_CLOB4(s, 88, 89, 90, 91), RS_CLOB4(s, 92, 93, 94, 95), RS_CLOB4(s, 96, 97, 98, 99),      \
    "s33", "v48", "memory"

// the chunk in and out: a running scalar pointer to the pixel's row of slices + the lane's byte offset; 8 pixels of a block row are
// RS_PS bytes apart, the next block row follows after RS_RS more
#define RS_XROW(OP, R)                                                                                    \
    OP(8*R+0) OP(8*R+1) OP(8*R+2) OP(8*R+3) OP(8*R+4) OP(8*R+5) OP(8*R+6) OP(8*R+7)                       \
    "s_add_u32 s36, s36, %[rs]\n" "s_addc_u32 s37, s37, 0\n"
#define RS_XALL(OP) RS_XROW(OP, 0) RS_XROW(OP, 1) RS_XROW(OP, 2) RS_XROW(OP, 3) RS_XROW(OP, 4) RS_XROW(OP, 5) RS_XROW(OP, 6) RS_XROW(OP, 7)
#define RS_XLD1(Q) "global_load_dword v[64+" #Q "], %[voff], s[36:37]\n" "s_add_u32 s36, s36, %[ps]\n" "s_addc_u32 s37, s37, 0\n"
#define RS_XZ1(Q) "v_mov_b32 v[64+" #Q "], 0\n"
#define RS_XZROW(R) RS_XZ1(8*R+0) RS_XZ1(8*R+1) RS_XZ1(8*R+2) RS_XZ1(8*R+3) RS_XZ1(8*R+4) RS_XZ1(8*R+5) RS_XZ1(8*R+6) RS_XZ1(8*R+7)
#define RS_XST1(Q) "global_store_dword %[voff], v[64+" #Q "], s[36:37]\n" "s_add_u32 s36, s36, %[ps]\n" "s_addc_u32 s37, s37, 0\n"
// tracked store, one block row at a time: d = x - snapshot, sum += (double)(d * d) (k_bp_angle<TRACK>'s expression), x to both volumes
#define RS_TLD1(C) "global_load_dword v[49+" #C "], %[voff], s[38:39]\n" "s_add_u32 s38, s38, %[ps]\n" "s_addc_u32 s39, s39, 0\n"
#define RS_TST1(R, C)                                                                                     \
    "v_sub_f32 v48, v[64+8*" #R "+" #C "], v[49+" #C "]\n"                                                \
    "v_mul_f32 v48, v48, v48\n"                                                                           \
    "v_cvt_f64_f32 v[58:59], v48\n"                                                                       \
    "v_add_f64 %[sum], %[sum], v[58:59]\n"                                                                \
    "global_store_dword %[voff], v[64+8*" #R "+" #C "], s[36:37]\n" "s_add_u32 s36, s36, %[ps]\n" "s_addc_u32 s37, s37, 0\n" \
    "global_store_dword %[voff], v[64+8*" #R "+" #C "], s[40:41]\n" "s_add_u32 s40, s40, %[ps]\n" "s_addc_u32 s41, s41, 0\n"
#define RS_TROW(R)                                                                                        \
    RS_TLD1(0) RS_TLD1(1) RS_TLD1(2) RS_TLD1(3) RS_TLD1(4) RS_TLD1(5) RS_TLD1(6) RS_TLD1(7)               \
    "s_waitcnt vmcnt(0)\n"                                                                                \
    RS_TST1(R, 0) RS_TST1(R, 1) RS_TST1(R, 2) RS_TST1(R, 3) RS_TST1(R, 4) RS_TST1(R, 5) RS_TST1(R, 6) RS_TST1(R, 7) \
    "s_add_u32 s36, s36, %[rs]\n" "s_addc_u32 s37, s37, 0\n"                                              \
    "s_add_u32 s38, s38, %[rs]\n" "s_addc_u32 s39, s39, 0\n"                                              \
    "s_add_u32 s40, s40, %[rs]\n" "s_addc_u32 s41, s41, 0\n"

// the kernel's arguments are read from the kernarg segment through a pointer that is made opaque at the head of every phase: taken as
// plain by-value arguments the compiler keeps all of them (and every loop-invariant product of them) alive across the sweep -- in
// ~36 SGPRs, because the two loops above own s[36:99]: hundreds of lane spills and, behind them, vector spills to scratch
// (constant address space: every read of an argument, and of the tables cast the same way below, is a scalar load)
#define RS_K __attribute__((address_space(4)))
typedef const RS_K RsArgs *RsArgsP;
__device__ __forceinline__ RsArgsP rs_args()
{
    RsArgsP p = (RsArgsP)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p) :: "memory");
    return p;
}

__device__ __forceinline__ int rs_angle(RsArgsP A, int k)
{
    return ((const RS_K int *)A->angs)[k];
}

// The cells stream from HBM once per chunk and angle, and a scalar load has nothing to hide a miss behind: a step ahead of their first
// use (the forward projection; the back projection a step later finds them in the L2) every wave pulls the lines of its 1-KB cell
// block into the L2 with one LDS-DMA load per 128-byte line (no register to wait for: the data goes to a dump row of the LDS that
// nobody reads)
__device__ __forceinline__ void rs_touch(RsArgsP A, int tile, int wave, int lane, int an, float *dump)
{
    if (lane < 8 && an >= 0) {
        const uint32_t *tp = reinterpret_cast<const uint32_t *>(A->cell + (((size_t)an * A->ntiles + tile) * RS_WAVES + wave) * 64) + lane * 32;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)tp, (__attribute__((address_space(3))) void *)dump, 4, 0, 0);
    }
}

__global__ __launch_bounds__(RS_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4)))
void k_sart_resident(const RsArgs unused_by_name)
{
    __shared__ float rs_pbuf[RS_WAVES][16][64];          // the waves' block sums of a forward projection (64 KB)
    __shared__ float rs_rbuf[RS_MAXWIN][64];             // the residual rows of the tile's window (12 KB)
    __shared__ float rs_sbuf[RS_WAVES][64];              // a reducer wave's share of a ray sum (4 KB)
    __shared__ float rs_dump[RS_WAVES][64];              // where the cell prefetches land (never read)
    __shared__ int rs_dirty, rs_go, rs_store;                   // a wave of this workgroup left a spin without its data; the chunk's verdict
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    v32f xlo, xhi;
    v16f rr;
#ifdef RS_PROF
    long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = (long long)__builtin_amdgcn_s_memrealtime();
    const long long clk0 = (long long)__builtin_amdgcn_s_memtime(), rt0 = tprev;
#endif
    int tile, grp, ngrp, chunk, chunk_end, steps;
    {
        RsArgsP A = rs_args();
        tile = (int)blockIdx.x % A->ntiles; grp = (int)blockIdx.x / A->ntiles; ngrp = (int)gridDim.x / A->ntiles;
        chunk = A->chunk0 + grp; chunk_end = A->chunk0 + A->nchunk; steps = A->steps;
    }
    for (int it = 0; chunk < chunk_end; chunk += ngrp, ++it) {
        const uint32_t voff = (uint32_t)(chunk * 64 + lane) * 4u;
        {   // a sweep that has given up touches no further chunk (the host redoes what did not commit)
            if (threadIdx.x == 0) { rs_dirty = 0; rs_go = rs_abort_ld(rs_args()->abort_word) == 0 ? 1 : 0; }
            __syncthreads();
            if (rs_go == 0) break;
        }
        {   // ---- the chunk comes in (a block outside the image holds zeros: its cells carry no weights)
            RsArgsP A = rs_args();
            rs_touch(A, tile, wave, lane, rs_angle(A, 0), rs_dump[wave]);
            if (steps > 1) rs_touch(A, tile, wave, lane, rs_angle(A, 1), rs_dump[wave]);
            const int n = A->n, sx = A->sx;
            const int ty = tile / A->tiles, tz = tile - ty * A->tiles;
            const int y0 = ty * RS_T + (wave >> 2) * 8, z0 = tz * RS_T + (wave & 3) * 8;
            const int in = __builtin_amdgcn_readfirstlane((y0 < n && z0 < n) ? 1 : 0);          // n is a multiple of 8: a block is inside or outside as a whole
            const float *xb = A->x + ((size_t)min(y0, n - 1) * n + min(z0, n - 1)) * sx;
            const uint32_t ps = (uint32_t)sx * 4u, rs = (uint32_t)(n - 8) * ps;
            asm volatile("s_cmp_eq_u32 %[in], 0\n"
                         "s_cbranch_scc1 1f\n"
                         "s_mov_b64 s[36:37], %[xb]\n"
                         RS_XALL(RS_XLD1)
                         "s_waitcnt vmcnt(0)\n"
                         "s_branch 2f\n"
                         "1:\n"
                         RS_XZROW(0) RS_XZROW(1) RS_XZROW(2) RS_XZROW(3) RS_XZROW(4) RS_XZROW(5) RS_XZROW(6) RS_XZROW(7)
                         "2:\n"
                         : "={v[64:95]}"(xlo), "={v[96:127]}"(xhi)
                         : [in] "s"(in), [xb] "s"(xb), [voff] "v"(voff), [ps] "s"(ps), [rs] "s"(rs)
                         : "s36", "s37", "scc", "memory");
        }
        int c_jbase = 0, c_nr = 0, c_dwv = 0;       // the window of the angle whose rows are awaited (read in the forward phase before)
        for (int k = -1; k < steps; ++k) {
            // What this step's forward phase needs that does not depend on the sweep's state -- the window of its angle, the first
            // entries of this wave's reducer list, the measured row and the row sum of its ray -- is requested here, in front of the
            // wait for the residual rows, so that none of these (cold) loads sits on the path from the rows to the published sums.
            // reducer duty: rays [tile * rpt, tile * rpt + rpt) of every angle; a ray's list is shared by wpr waves
            int f_nr = 0, f_jbase = 0, f_rpt = 1, f_rpt2 = 1, f_wpr = RS_WAVES, f_cpw = RS_RL / RS_WAVES, f_jr0 = 0, f_sub = 0;
            uint32_t f_dwv = 0;
            rs_u2 f_ts0 = {0u, 0u}, f_ts1 = {0u, 0u}, f_ts2 = {0u, 0u};
            rs_u2 f_ids0 = {0xFFFFFFFFu, 0xFFFFFFFFu};
            float f_rs0 = 0.f, f_bv0 = 0.f;
            if (k + 1 < steps) {
                RsArgsP A = rs_args();
                const int n = A->n;
                const int a = rs_angle(A, k + 1);
                const RS_K RsHdrD *h = (const RS_K RsHdrD *)A->hdr + (size_t)a * A->ntiles + tile;
                f_nr = h->nrays; f_jbase = h->jbase;
                f_dwv = (h->dw[wave >> 2] >> ((wave & 3) * 8)) & 255u;
                {   // the block sums behind this wave's (at most three) window rays
                    const RS_K rs_u2 *tsp = (const RS_K rs_u2 *)A->ts + ((size_t)a * A->ntiles + tile) * RS_MAXWIN;
                    f_ts0 = tsp[wave]; f_ts1 = tsp[wave + 16]; f_ts2 = tsp[wave + 32];
                }
                f_rpt = A->rpt;
                while (f_rpt2 < f_rpt && f_rpt2 < RS_WAVES) f_rpt2 *= 2;
                f_wpr = RS_WAVES / f_rpt2; f_cpw = RS_RL / f_wpr;
                f_jr0 = wave / f_wpr; f_sub = wave % f_wpr;
                const int j0 = min(tile * f_rpt + f_jr0, n - 1);
                f_ids0 = ((const RS_K rs_u2 *)A->rl)[(((size_t)a * n + j0) * RS_RL + f_sub * f_cpw) >> 2];
                f_rs0 = ((const RS_K float *)A->rowsum)[(size_t)a * n + j0];
                f_bv0 = A->b[((size_t)a * n + j0) * A->sx + (voff >> 2)];
            }
            RS_STAMP(7)
            if (k >= 0) {
                // ---- the residual rows of angle a: pick up the tile's window, then the wave's own rows into v[32:45]
                RsArgsP A = rs_args();
                const int n = A->n;
                const int a = rs_angle(A, k);
                const unsigned ep = A->epoch0 + (unsigned)it * (unsigned)steps + (unsigned)k + 1u;     // step k carries this tag
                const int jbase = c_jbase, nr = c_nr, dwv = c_dwv;
                const rs_u64 *rrow = A->rb + (((size_t)grp * A->np + a) * n + jbase) * 64 + lane;
                {
                    const int i0 = wave, i1 = wave + 16, i2 = wave + 32;
                    rs_u64 g0 = 0, g1 = 0, g2 = 0;
                    unsigned spins = 0;
                    for (;;) {
                        if (i0 < nr) g0 = rs_gld(rrow + (size_t)i0 * 64);
                        if (i1 < nr) g1 = rs_gld(rrow + (size_t)i1 * 64);
                        if (i2 < nr) g2 = rs_gld(rrow + (size_t)i2 * 64);
                        bool ok = (i0 >= nr || (unsigned)(g0 >> 32) == ep) && (i1 >= nr || (unsigned)(g1 >> 32) == ep) && (i2 >= nr || (unsigned)(g2 >> 32) == ep);
                        if (__all(ok)) break;
                        if (rs_give_up(spins, A->spin_limit, A->abort_word, A->abort_host, lane, 1)) { rs_dirty = 1; break; }
                        __builtin_amdgcn_s_sleep(1);
                    }
                    if (i0 < nr) rs_rbuf[i0][lane] = __uint_as_float((uint32_t)g0);
                    if (i1 < nr) rs_rbuf[i1][lane] = __uint_as_float((uint32_t)g1);
                    if (i2 < nr) rs_rbuf[i2][lane] = __uint_as_float((uint32_t)g2);
                }
                RS_STAMP(0)
                RS_TL(0)
                __syncthreads();
#pragma unroll
                for (int s = 0; s < 16; ++s) rr[s] = (s < RS_USABLE && dwv + s < nr) ? rs_rbuf[dwv + s][lane] : 0.f;
                const uint4 *cp = A->cell + (((size_t)a * A->ntiles + tile) * RS_WAVES + wave) * 64;
                const float beta = A->beta;
                rs_touch(A, tile, wave, lane, k + 2 < steps ? rs_angle(A, k + 2) : -1, rs_dump[wave]);
                RS_STAMP(1)
                RS_TL(1)
                asm volatile("s_mov_b32 s33, m0\n"
                             RS_LD(0x000)
                             RS_WAIT
                             "s_set_gpr_idx_on s36, gpr_idx(SRC1)\n"
                             RS_SWEEP(RS_BP1)
                             "s_set_gpr_idx_off\n"
                             "s_mov_b32 m0, s33\n"
                             : "+{v[64:95]}"(xlo), "+{v[96:127]}"(xhi), "+{v[32:47]}"(rr)
                             : [cp] "s"(cp), [beta] "s"(beta)
                             : RS_CLOBBERS);
                RS_STAMP(2)
                RS_TL(2)
            }
            if (k + 1 < steps) {
                // ---- forward projection of the next angle, the workgroup's sums per window ray, the reducers' rows
                RsArgsP A = rs_args();
                const int n = A->n;
                const int a = rs_angle(A, k + 1);
                const unsigned ep = A->epoch0 + (unsigned)it * (unsigned)steps + (unsigned)k + 2u;
                const int nr = f_nr, rpt = f_rpt, rpt2 = f_rpt2, wpr = f_wpr, cpw = f_cpw, jr0 = f_jr0, sub = f_sub;
                const rs_u2 ids0 = f_ids0;
                const float rs0 = f_rs0, bv0 = f_bv0;
                c_jbase = f_jbase; c_nr = nr;
                c_dwv = (int)f_dwv;
#pragma unroll
                for (int s = 0; s < 16; ++s) rr[s] = 0.f;
                const uint4 *cp = A->cell + (((size_t)a * A->ntiles + tile) * RS_WAVES + wave) * 64;
                asm volatile("s_mov_b32 s33, m0\n"
                             RS_LD(0x000)
                             RS_WAIT
                             "s_set_gpr_idx_on s36, gpr_idx(SRC2,DST)\n"
                             RS_SWEEP(RS_FP1)
                             "s_set_gpr_idx_off\n"
                             "s_mov_b32 m0, s33\n"
                             : "+{v[64:95]}"(xlo), "+{v[96:127]}"(xhi), "+{v[32:47]}"(rr)
                             : [cp] "s"(cp)
                             : RS_CLOBBERS);
                RS_STAMP(3)
                RS_TL(3)
#pragma unroll
                for (int s = 0; s < 16; ++s) rs_pbuf[wave][s][lane] = rr[s];
                __syncthreads();
                rs_u64 *pb = A->pb + ((size_t)grp * A->ntiles + tile) * RS_MAXWIN * 64 + lane;
                // the tile's sum of window ray i = the block sums its list names (bytes wave << 4 | slot = row of rs_pbuf), in wave order;
                // the list is padded to eight with a row that is always zero.  (The first form asked all sixteen waves "is ray i in
                // your window": 350 instructions per wave -- 2.6 us between the last projection and the last published sum.)
#pragma unroll
                for (int r3 = 0; r3 < 3; ++r3) {
                    const int i = wave + 16 * r3;
                    if (i < nr) {
                        const rs_u2 e = r3 == 0 ? f_ts0 : r3 == 1 ? f_ts1 : f_ts2;
                        const float *pr = &rs_pbuf[0][0][0] + lane;
                        float v[8];
#pragma unroll
                        for (int c = 0; c < 8; ++c) v[c] = pr[(((c < 4 ? e.x : e.y) >> ((c & 3) * 8)) & 255u) * 64];
                        float acc = v[0];
#pragma unroll
                        for (int c = 1; c < 8; ++c) acc += v[c];
                        rs_gst(pb + (size_t)i * 64, acc, ep);
                    }
                }
                RS_STAMP(4)
                RS_TL(4)
                const rs_u64 *pg = A->pb + (size_t)grp * A->ntiles * RS_MAXWIN * 64 + lane;
                for (int r0 = 0; r0 < rpt; r0 += rpt2) {
                    const int jr = r0 + jr0, j = tile * rpt + jr;
                    const bool active = jr < rpt && j < n;
                    if (active) {
                        float acc = 0.f;
                        const RS_K rs_u2 *list = (const RS_K rs_u2 *)A->rl + ((((size_t)a * n + j) * RS_RL + sub * cpw) >> 2);
                        for (int e0 = 0; e0 < cpw; e0 += 4) {
                            const rs_u2 ids = (r0 == 0 && e0 == 0) ? ids0 : list[e0 >> 2];
                            const int id0 = ids.x & 0xFFFF, id1 = ids.x >> 16, id2 = ids.y & 0xFFFF, id3 = ids.y >> 16;
                            if (id0 == 0xFFFF) break;
                            rs_u64 g0 = 0, g1 = 0, g2 = 0, g3 = 0;
                            unsigned spins = 0;
                            for (;;) {
                                g0 = rs_gld(pg + (size_t)id0 * 64);
                                if (id1 != 0xFFFF) g1 = rs_gld(pg + (size_t)id1 * 64);
                                if (id2 != 0xFFFF) g2 = rs_gld(pg + (size_t)id2 * 64);
                                if (id3 != 0xFFFF) g3 = rs_gld(pg + (size_t)id3 * 64);
                                bool ok = (unsigned)(g0 >> 32) == ep && (id1 == 0xFFFF || (unsigned)(g1 >> 32) == ep) &&
                                          (id2 == 0xFFFF || (unsigned)(g2 >> 32) == ep) && (id3 == 0xFFFF || (unsigned)(g3 >> 32) == ep);
                                if (__all(ok)) break;
                                if (rs_give_up(spins, A->spin_limit, A->abort_word, A->abort_host, lane, 2)) { rs_dirty = 1; break; }
                                __builtin_amdgcn_s_sleep(1);
                            }
                            acc += __uint_as_float((uint32_t)g0);
                            if (id1 != 0xFFFF) acc += __uint_as_float((uint32_t)g1);
                            if (id2 != 0xFFFF) acc += __uint_as_float((uint32_t)g2);
                            if (id3 != 0xFFFF) acc += __uint_as_float((uint32_t)g3);
                        }
                        rs_sbuf[wave][lane] = acc;
                    }
                    RS_STAMP(5)
                    RS_TL(5)
                    __syncthreads();
                    if (active && sub == 0) {
                        float tot = rs_sbuf[wave][lane];
                        for (int u = 1; u < wpr; ++u) tot += rs_sbuf[wave + u][lane];
                        const size_t row = (size_t)a * n + j;
                        const float bv = r0 == 0 ? bv0 : A->b[row * A->sx + (voff >> 2)], rs = r0 == 0 ? rs0 : ((const RS_K float *)A->rowsum)[row];
                        const float rv = rs > 0.f ? (bv - tot) / rs : 0.f;      // k_resid_finish's expression
                        rs_gst(A->rb + (((size_t)grp * A->np) * n + row) * 64 + lane, rv, ep);
                    }
                    if (r0 + rpt2 < rpt) __syncthreads();
                }
                RS_STAMP(6)
                RS_TL(6)
            }
        }
        {   // ---- commit: every workgroup of the chunk stores, or none does
            __syncthreads();
            if (threadIdx.x == 0) {
                RsArgsP A = rs_args();
                const bool clean = rs_dirty == 0 && rs_abort_ld(A->abort_word) == 0 && !(tile == 0 && A->test_fail == chunk + 1);
                const bool ok = rs_commit(A->commit + chunk, A->commit_base, (unsigned)A->ntiles, clean, A->spin_limit, A->abort_word);
                if (ok && tile == 0)
                    __hip_atomic_store((__attribute__((address_space(1))) int *)(A->done_host + chunk), (int)A->seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                rs_store = ok ? 1 : 0;
            }
            __syncthreads();
            if (rs_store == 0) continue;       // (a sweep that gave up stops at the next chunk's first look)
        }
        {   // ---- the chunk goes back (and, tracked, into the snapshot volume with the squared step in part[])
            RsArgsP A = rs_args();
            const int n = A->n, sx = A->sx;
            const int ty = tile / A->tiles, tz = tile - ty * A->tiles;
            const int y0 = ty * RS_T + (wave >> 2) * 8, z0 = tz * RS_T + (wave & 3) * 8;
            const int in = __builtin_amdgcn_readfirstlane((y0 < n && z0 < n) ? 1 : 0);
            const size_t o0 = ((size_t)min(y0, n - 1) * n + min(z0, n - 1)) * sx;
            float *xb = A->x + o0;
            const uint32_t ps = (uint32_t)sx * 4u, rs = (uint32_t)(n - 8) * ps;
            if (A->track) {
                float *tb = A->track + o0;
                double local = 0.0;
                asm volatile("s_cmp_eq_u32 %[in], 0\n"
                             "s_cbranch_scc1 1f\n"
                             "s_mov_b64 s[36:37], %[xb]\n"
                             "s_mov_b64 s[38:39], %[tb]\n"
                             "s_mov_b64 s[40:41], %[tb]\n"
                             RS_TROW(0) RS_TROW(1) RS_TROW(2) RS_TROW(3) RS_TROW(4) RS_TROW(5) RS_TROW(6) RS_TROW(7)
                             "s_waitcnt vmcnt(0)\n"
                             "1:\n"
                             : [sum] "+v"(local)
                             : "{v[64:95]}"(xlo), "{v[96:127]}"(xhi), [in] "s"(in), [xb] "s"(xb), [tb] "s"(tb), [voff] "v"(voff), [ps] "s"(ps), [rs] "s"(rs)
                             : "s36", "s37", "s38", "s39", "s40", "s41", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v58", "v59", "scc", "memory");
                // the wave's sum through its own row of the LDS, added by lane 0 (a shuffle tree and the compiler's atomic optimiser both
                // want a lane count that the compiler then keeps alive across the whole sweep -- in scratch)
                double *red = reinterpret_cast<double *>(&rs_pbuf[wave][0][0]);
                red[lane] = local;
                __builtin_amdgcn_wave_barrier();
                if (lane == 0) {
                    local = 0.0;
                    for (int l = 0; l < 64; ++l) local += red[l];
                    const uint32_t slot = (blockIdx.x & (NPART - 1)) * 8u;
                    double *part = A->part;
                    asm volatile("global_atomic_add_f64 %0, %1, %2" :: "v"(slot), "v"(local), "s"(part) : "memory");
                }
            } else {
                asm volatile("s_cmp_eq_u32 %[in], 0\n"
                             "s_cbranch_scc1 1f\n"
                             "s_mov_b64 s[36:37], %[xb]\n"
                             RS_XALL(RS_XST1)
                             "s_waitcnt vmcnt(0)\n"
                             "1:\n"
                             :
                             : "{v[64:95]}"(xlo), "{v[96:127]}"(xhi), [in] "s"(in), [xb] "s"(xb), [voff] "v"(voff), [ps] "s"(ps), [rs] "s"(rs)
                             : "s36", "s37", "scc", "memory");
            }
        }
    }
#ifdef RS_PROF
    if (threadIdx.x == 0) {
        RsArgsP A = rs_args();
        for (int q = 0; q < 8; ++q) A->prof[blockIdx.x * 8 + q] = tacc[q];
        if (blockIdx.x == 0) { A->prof[2048 + 126] = (long long)__builtin_amdgcn_s_memtime() - clk0; A->prof[2048 + 127] = (long long)__builtin_amdgcn_s_memrealtime() - rt0; }
    }
#endif
}
#undef RS_K
#undef RS_TROW
#undef RS_TST1
#undef RS_TLD1
#undef RS_XST1
#undef RS_XZROW
#undef RS_XZ1
#undef RS_XLD1
#undef RS_XALL
#undef RS_XROW
#undef RS_CLOBBERS
#undef RS_CLOB4
#undef RS_SWEEP
#undef RS_WAIT
#undef RS_BATCH
#undef RS_FP1
#undef RS_IDX
#undef RS_BP1
#undef RS_LD

}  // namespace tomo
