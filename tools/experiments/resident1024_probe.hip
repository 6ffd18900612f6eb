// Probe for a volume-RESIDENT SART sweep at N = 1024 (config 4: 1024^3 x 120, one GPU's shard 128 x 1024^2).
//
// k_sart_resident (sart_resident.hip.h) keeps a 64-slice chunk of a 512^2 image in registers with lane = slice.  At 1024^2 the register
// file of the chip (256 CUs x 512 KB) holds a 16-slice chunk: 64 x 64 pixels x 16 slices = 256 KB per CU.  With 16 slices a lane cannot
// be a slice any more; the layout probed here:
//   * one workgroup of 16 waves per 64 x 64 tile = one per CU; wave w holds rows 4w .. 4w+3 of the tile, LANE = COLUMN, and a pixel's
//     16 slices sit in 16 registers of its lane: x[4 rows][16 slices] = 64 VGPRs (as much as today);
//   * the cell {s0, w0, w1, 1/(w0+w1)} of a pixel is per-LANE data now (one coalesced 16-byte load per lane and row, used for 16 slices);
//   * back projection: the residual rows of the tile's window (<= 96 rays x 16 slices) sit in LDS; a lane reads the 2 x 16 values of its
//     two rays (8 x ds_read_b128) and applies k_bp_angle's expression to its 16 registers;
//   * forward projection: per wave a private LDS array of ray sums [<= 72 rays][16 slices]; a lane ADDS w0 x and w1 x for its 16
//     slices (ds_add_f32: lanes = columns of ONE image row are on different rays except for runs of <= 3 neighbours at the steepest
//     angles, so an instruction has few conflicts); then the tile's sum per window ray = the waves' sums in ascending wave order.
// What is measured: the COMPUTE side of a step (rows LDS <- global stand-in for the exchange's pick-up, back projection, forward
// projection, tile sums -> global stand-in for the publish) without any waiting between workgroups, per angle and 16-slice chunk, and
// bit-compared with a CPU replay in the kernel's order (conflicting lanes of one ds_add_f32 applied in ascending lane order).
// The exchange itself (two hand-offs per angle) is what k_sart_resident already pays: ~5 us (profiles/r06_resident_phases.txt).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I../../tomo_tv_amd/csrc resident1024_probe.hip ../../tomo_tv_amd/csrc/sysmat.cpp -lpthread -o resident1024_probe
//   ./resident1024_probe [N=1024] [P=120] [steps=P] [reps=3] [mode=3: bit 0 back projection, bit 1 forward projection by LDS float adds,
//                        bit 2 forward without the LDS adds, bit 3 forward projection WITHOUT atomics (mode 9 = full step in that form, bit-compared):
//                        the lanes of an image row are sorted by ray and runs of equal rays are <= 4 lanes (|theta| <= 70 deg: a line 20 deg off the row direction crosses <= 4 pixels of a row), so three DPP
//                        shifts with 0 / 1 masks form the run totals in the run's last lane, which adds them to the wave's OWN LDS rows by a
//                        plain read - add - write (distinct rays per instruction; the wave's four rows one after the other)]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "sysmat.h"

using namespace tomo;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int T = 64, SL = 16, WAVES = 16, RPW = 4, MAXWIN = 96, WWIN = 72;
struct PCell { int32_t s0; float w0, w1, inv; };                 // s0: first ray of the pixel (index inside the angle); its second ray is s0 + 1
struct THdr { int32_t jbase, nr; };                              // the tile's window of an angle
struct WHdr { int32_t wbase, wnr; };                             // a wave's window (its four rows)

static inline float hashf(uint64_t i, uint32_t salt)
{
    uint64_t z = (i + 0x9E3779B97F4A7C15ull * (salt + 1));
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
    return (float)(z >> 40) * (1.0f / 16777216.0f);
}

template <int mode>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(4, 4)))
void k_probe(float *__restrict__ xg, const PCell *__restrict__ cells, const THdr *__restrict__ thdr, const WHdr *__restrict__ whdr,
             const float *__restrict__ rows, float *__restrict__ tsum, int n, int tiles, int nproj, int steps, float beta)
{
    __shared__ float r_lds[MAXWIN + 2][SL];                      // residual rows of the tile's window (+ a zero row pair for pixels without rays)
    __shared__ float acc[WAVES][WWIN][SL];                       // per wave: the sums of the rays through its four rows
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tile = blockIdx.x, ty = tile / tiles, tz = tile - ty * tiles, ntiles = tiles * tiles;
    const int z = tz * T + lane;
    float x[RPW][SL];
#pragma unroll
    for (int g = 0; g < RPW; ++g) {
        const int y = ty * T + wave * RPW + g;
        const float4 *p = reinterpret_cast<const float4 *>(xg + ((size_t)y * n + z) * SL);
#pragma unroll
        for (int q = 0; q < 4; ++q) { float4 v = p[q]; x[g][4 * q] = v.x; x[g][4 * q + 1] = v.y; x[g][4 * q + 2] = v.z; x[g][4 * q + 3] = v.w; }
    }
    for (int k = -1; k < steps; ++k) {
        if (k >= 0 && (mode & 1)) {
            // ---- back projection of angle a: rows of the window into LDS (the exchange's pick-up), then the voxel update
            const int a = k % nproj;
            const THdr h = thdr[(size_t)a * ntiles + tile];
            __syncthreads();
            for (int i = threadIdx.x; i < (h.nr + 2) * SL; i += 1024)
                r_lds[0][i] = i < h.nr * SL ? rows[((size_t)a * n + h.jbase) * SL + i] : 0.f;
            __syncthreads();
#pragma unroll
            for (int g = 0; g < RPW; ++g) {
                const int y = ty * T + wave * RPW + g;
                const PCell c = cells[((size_t)a * n + y) * n + z];
                const float4 *rp = reinterpret_cast<const float4 *>(&r_lds[max(c.s0, h.jbase) - h.jbase][0]);      // (a pixel without rays: s0 = -1, weights 0)
                float r0[SL], r1[SL];
#pragma unroll
                for (int q = 0; q < 4; ++q) { float4 v = rp[q]; r0[4 * q] = v.x; r0[4 * q + 1] = v.y; r0[4 * q + 2] = v.z; r0[4 * q + 3] = v.w; }
#pragma unroll
                for (int q = 0; q < 4; ++q) { float4 v = rp[4 + q]; r1[4 * q] = v.x; r1[4 * q + 1] = v.y; r1[4 * q + 2] = v.z; r1[4 * q + 3] = v.w; }
#pragma unroll
                for (int s = 0; s < SL; ++s) {
                    float t = c.w0 * r0[s];
                    t = __builtin_fmaf(c.w1, r1[s], t);
                    t = t * c.inv;
                    x[g][s] = fmaxf(__builtin_fmaf(t, beta, x[g][s]), 0.f);
                }
            }
        }
        if (k + 1 < steps && (mode & 14)) {
            // ---- forward projection of angle a + 1: per-wave ray sums by LDS adds, then the tile's sums per window ray (the publish)
            const int a = (k + 1) % nproj;
            const THdr h = thdr[(size_t)a * ntiles + tile];
            const WHdr wh = whdr[((size_t)a * ntiles + tile) * WAVES + wave];
            for (int i = lane; i < WWIN * SL; i += 64) acc[wave][0][i] = 0.f;
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int g = 0; g < RPW; ++g) {
                const int y = ty * T + wave * RPW + g;
                const PCell c = cells[((size_t)a * n + y) * n + z];
                float *ap = &acc[wave][max(c.s0, wh.wbase) - wh.wbase][0];
                if (mode & 8) {
                    // run structure of this row: lanes sorted by ray; m1 / m2: the lane one / two to the left is on the same ray; tail: last of its run
                    const int s0 = c.s0;
                    const int l1 = __builtin_amdgcn_update_dpp(-2, s0, 0x138, 0xf, 0xf, false);        // wave_shr:1 (lane 0 keeps -2)
                    const int l2 = __builtin_amdgcn_update_dpp(-2, l1, 0x138, 0xf, 0xf, false);
                    const int l3 = __builtin_amdgcn_update_dpp(-2, l2, 0x138, 0xf, 0xf, false);
                    const int r1 = __builtin_amdgcn_update_dpp(-2, s0, 0x130, 0xf, 0xf, false);        // wave_shl:1 (lane 63 keeps -2)
                    const float m1 = (s0 >= 0 && l1 == s0) ? 1.f : 0.f, m2 = (s0 >= 0 && l2 == s0) ? 1.f : 0.f, m3 = (s0 >= 0 && l3 == s0) ? 1.f : 0.f;
                    const bool tail = s0 >= 0 && r1 != s0;
#pragma unroll
                    for (int pass = 0; pass < 2; ++pass) {
                        const float w = pass ? c.w1 : c.w0;
                        float4 *q4 = reinterpret_cast<float4 *>(ap + pass * SL);
#pragma unroll
                        for (int q = 0; q < 4; ++q) {               // four slices at a time: one 16-byte read - add - write per run
                            float tot[4];
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                const float v = w * x[g][4 * q + u];
                                const float a1 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, false));
                                const float a2 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a1), 0x138, 0xf, 0xf, false));
                                const float a3 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a2), 0x138, 0xf, 0xf, false));
                                tot[u] = __builtin_fmaf(m3, a3, __builtin_fmaf(m2, a2, __builtin_fmaf(m1, a1, v)));
                            }
                            if (tail) {
                                float4 o = q4[q];
                                o.x += tot[0]; o.y += tot[1]; o.z += tot[2]; o.w += tot[3];
                                q4[q] = o;
                            }
                        }
                        __builtin_amdgcn_wave_barrier();
                    }
                } else if (mode & 2) {
#pragma unroll
                    for (int s = 0; s < SL; ++s) { __hip_atomic_fetch_add(ap + s, c.w0 * x[g][s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
#pragma unroll
                    for (int s = 0; s < SL; ++s) { __hip_atomic_fetch_add(ap + SL + s, c.w1 * x[g][s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
                } else {            // (bit 2 alone: the products without the LDS adds -- what the adds cost)
                    float u = 0.f;
#pragma unroll
                    for (int s = 0; s < SL; ++s) u += c.w0 * x[g][s] + c.w1 * x[g][s];
                    if (u == 1.2345e-30f) ap[0] = u;
                }
            }
            __syncthreads();
            for (int i = threadIdx.x; i < h.nr * SL; i += 1024) {
                const int ray = h.jbase + i / SL, s = i % SL;
                float tot = 0.f;
                for (int w = 0; w < WAVES; ++w) {
                    const WHdr o = whdr[((size_t)a * ntiles + tile) * WAVES + w];
                    const int rel = ray - o.wbase;
                    if (rel >= 0 && rel < o.wnr) tot += acc[w][rel][s];
                }
                tsum[(((size_t)a * ntiles + tile) * MAXWIN) * SL + i] = tot;
            }
        }
    }
#pragma unroll
    for (int g = 0; g < RPW; ++g) {
        const int y = ty * T + wave * RPW + g;
        float4 *p = reinterpret_cast<float4 *>(xg + ((size_t)y * n + z) * SL);
#pragma unroll
        for (int q = 0; q < 4; ++q) p[q] = make_float4(x[g][4 * q], x[g][4 * q + 1], x[g][4 * q + 2], x[g][4 * q + 3]);
    }
}

int main(int argc, char **argv)
{
    const int N = argc > 1 ? atoi(argv[1]) : 1024, P = argc > 2 ? atoi(argv[2]) : 120;
    const int steps = argc > 3 ? atoi(argv[3]) : P, reps = argc > 4 ? atoi(argv[4]) : 3, mode = argc > 5 ? atoi(argv[5]) : 3;
    const float beta = 0.7f;
    hipDeviceProp_t prop;
    const bool host_only = hipGetDeviceProperties(&prop, 0) != hipSuccess;      // (no device: build and check the tables only)
    if (host_only) { prop.multiProcessorCount = 256; std::snprintf(prop.name, sizeof(prop.name), "no device"); }
    const int tiles = N / T, ntiles = tiles * tiles;
    printf("%s: %d CUs; N %d P %d steps %d mode %d: %d tiles of %d x %d pixels x %d slices\n", prop.name, prop.multiProcessorCount, N, P, steps, mode, ntiles, T, T, SL);
    if (N % T || ntiles > prop.multiProcessorCount) { printf("needs N a multiple of %d and at most one tile per CU\n", T); return 1; }
    std::vector<double> ang(P);
    for (int i = 0; i < P; ++i) ang[i] = (P > 1 ? -70.0 + 140.0 * i / (P - 1) : 0.0) * M_PI / 180.0;
    Coo m; build_parallel_ray(N, P, ang.data(), m); sort_rows(m);
    Tables t; std::string err;
    if (!build_tables(m, N, P, t, err)) { printf("build_tables: %s\n", err.c_str()); return 1; }
    if (!t.art_chain_ok) { printf("a pixel's two rays are not neighbours\n"); return 1; }
    const size_t npix = (size_t)N * N;
    std::vector<PCell> cells((size_t)P * npix);
    std::vector<THdr> thdr((size_t)P * ntiles);
    std::vector<WHdr> whdr((size_t)P * ntiles * WAVES);
    int worst_win = 0, worst_wwin = 0;
    for (int a = 0; a < P; ++a) {
        for (int k = 0; k < ntiles; ++k) {
            int lo = 1 << 30, hi = -1;
            for (int w = 0; w < WAVES; ++w) {
                int wlo = 1 << 30, whi = -1;
                for (int g = 0; g < RPW; ++g) for (int l = 0; l < 64; ++l) {
                    const int y = (k / tiles) * T + w * RPW + g, z = (k % tiles) * T + l;
                    const Cell &c = t.cell[(size_t)a * npix + (size_t)y * N + z];
                    if (c.w0 == 0.f && c.w1 == 0.f) continue;
                    const int j0 = (int)c.r0, j1 = c.w1 != 0.f ? (int)c.r1 : j0;      // (Cell::r0 / r1 count inside the angle)
                    wlo = std::min(wlo, j0); whi = std::max(whi, std::max(j1, j0 + 1));
                }
                whdr[((size_t)a * ntiles + k) * WAVES + w] = whi < 0 ? WHdr{-1, 0} : WHdr{wlo, whi - wlo + 1};
                if (whi < 0) continue;
                worst_wwin = std::max(worst_wwin, whi - wlo + 1);
                lo = std::min(lo, wlo); hi = std::max(hi, whi);
            }
            if (hi < 0) { lo = 0; hi = 1; }                                   // a tile no ray of this angle crosses
            thdr[(size_t)a * ntiles + k] = THdr{lo, std::min(hi, N - 1) - lo + 1};
            worst_win = std::max(worst_win, std::min(hi, N - 1) - lo + 1);
            for (int w = 0; w < WAVES; ++w) {                                 // a wave without rays: an (unused) window at the tile's first ray
                WHdr &wh = whdr[((size_t)a * ntiles + k) * WAVES + w];
                if (wh.wbase < 0) wh = WHdr{lo, 2};
            }
        }
        for (size_t p = 0; p < npix; ++p) {
            const Cell &c = t.cell[(size_t)a * npix + p];
            PCell q;
            if (c.w0 == 0.f && c.w1 == 0.f) {
                const int k = (int)((p / N) / T) * tiles + (int)((p % N) / T);
                const THdr &h = thdr[(size_t)a * ntiles + k];
                const WHdr &wh = whdr[((size_t)a * ntiles + k) * WAVES + (int)(((p / N) % T) / RPW)];
                (void)h; (void)wh;
                q = PCell{-1, 0.f, 0.f, 1.f};                                  // no ray: the kernel clamps the index, the weights are zero
            } else {
                const float sum = c.w0 + c.w1;
                q = PCell{(int32_t)c.r0, c.w0, c.w1, 1.0f / sum};
            }
            cells[(size_t)a * npix + p] = q;
        }
    }
    // the no-atomics form needs: rays non-decreasing along a row of a tile, runs of equal rays <= 4 lanes, pixels without rays only at the ends
    int longest_run = 0, order_bad = 0;
    for (int a = 0; a < P; ++a)
        for (size_t y = 0; y < (size_t)N; ++y)
            for (int z0 = 0; z0 < N; z0 += T) {
                int run = 0, prev = -1, seen_real = 0, ended = 0;
                for (int l = 0; l < T; ++l) {
                    const int s0 = cells[(size_t)a * npix + y * N + z0 + l].s0;
                    if (s0 < 0) { if (seen_real) ended = 1; run = 0; prev = -1; continue; }
                    if (ended) ++order_bad;                       // a real pixel behind a gap
                    if (prev >= 0 && s0 < prev) ++order_bad;
                    run = (s0 == prev) ? run + 1 : 1;
                    longest_run = std::max(longest_run, run);
                    prev = s0; seen_real = 1;
                }
            }
    printf("rows of a tile: longest run of lanes on one ray %d (limit 4), order violations %d\n", longest_run, order_bad);
    printf("widest tile window %d (limit %d), widest wave window %d (limit %d)\n", worst_win, MAXWIN, worst_wwin, WWIN);
    if ((mode & 8) && (longest_run > 4 || order_bad)) { printf("the no-atomics form cannot run on this geometry\n"); return 1; }
    if (worst_win > MAXWIN || worst_wwin + 1 > WWIN) { printf("windows do not fit\n"); return 1; }
    if (host_only) return 0;
    std::vector<float> x0(npix * SL), rows((size_t)P * N * SL);
    for (size_t i = 0; i < x0.size(); ++i) x0[i] = hashf(i, 1);
    for (size_t i = 0; i < rows.size(); ++i) rows[i] = 0.02f * (hashf(i, 2) - 0.5f);
    float *dx, *drows, *dts; PCell *dc; THdr *dth; WHdr *dwh;
    CK(hipMalloc(&dx, x0.size() * 4)); CK(hipMalloc(&drows, rows.size() * 4)); CK(hipMalloc(&dts, (size_t)P * ntiles * MAXWIN * SL * 4));
    CK(hipMalloc(&dc, cells.size() * sizeof(PCell))); CK(hipMalloc(&dth, thdr.size() * sizeof(THdr))); CK(hipMalloc(&dwh, whdr.size() * sizeof(WHdr)));
    CK(hipMemcpy(drows, rows.data(), rows.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dc, cells.data(), cells.size() * sizeof(PCell), hipMemcpyHostToDevice));
    CK(hipMemcpy(dth, thdr.data(), thdr.size() * sizeof(THdr), hipMemcpyHostToDevice));
    CK(hipMemcpy(dwh, whdr.data(), whdr.size() * sizeof(WHdr), hipMemcpyHostToDevice));
    CK(hipMemset(dts, 0, (size_t)P * ntiles * MAXWIN * SL * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < reps; ++rep) {
        CK(hipMemcpy(dx, x0.data(), x0.size() * 4, hipMemcpyHostToDevice));
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
#define LAUNCH(M) case M: hipLaunchKernelGGL(k_probe<M>, dim3(ntiles), dim3(1024), 0, 0, dx, dc, dth, dwh, drows, dts, N, tiles, P, steps, beta); break;
        switch (mode) { LAUNCH(1) LAUNCH(2) LAUNCH(3) LAUNCH(4) LAUNCH(8) LAUNCH(9) default: printf("mode %d is not built\n", mode); return 1; }
#undef LAUNCH
        CK(hipEventRecord(e1));
        CK(hipGetLastError());
        CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms);
        printf("rep %d: %.3f ms = %.2f us per angle and 16-slice chunk (compute side only)\n", rep, ms, 1000.0 * ms / steps);
    }
    std::vector<float> got(x0.size()), gts((size_t)P * ntiles * MAXWIN * SL);
    CK(hipMemcpy(got.data(), dx, got.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(gts.data(), dts, gts.size() * 4, hipMemcpyDeviceToHost));
    int bad = 0, bad_ts = 0;
    if (mode == 3 || mode == 9) {
        // ---- CPU replay of three tiles in the kernel's order (a ds_add_f32's conflicting lanes in ascending lane order)
        const int check_tiles[] = {0, ntiles / 2 + tiles / 3, ntiles - 1};
        for (int ct = 0; ct < 3; ++ct) {
            const int k = check_tiles[ct], ty = k / tiles, tz = k % tiles;
            std::vector<float> x((size_t)T * T * SL);
            for (int y = 0; y < T; ++y) for (int zz = 0; zz < T; ++zz) for (int s = 0; s < SL; ++s)
                x[((size_t)y * T + zz) * SL + s] = x0[(((size_t)(ty * T + y)) * N + tz * T + zz) * SL + s];
            std::vector<float> last_ts((size_t)MAXWIN * SL);
            int last_a = -1;
            for (int kk = -1; kk < steps; ++kk) {
                if (kk >= 0) {
                    const int a = kk % P;
                    for (int y = 0; y < T; ++y) for (int zz = 0; zz < T; ++zz) {
                        const PCell &c = cells[(size_t)a * npix + (size_t)(ty * T + y) * N + tz * T + zz];
                        for (int s = 0; s < SL; ++s) {
                            const THdr &h = thdr[(size_t)a * ntiles + k];
                            const int j0 = std::max(c.s0, h.jbase);
                            const float r0 = rows[((size_t)a * N + j0) * SL + s];
                            const float r1 = j0 + 1 < N ? rows[((size_t)a * N + j0 + 1) * SL + s] : 0.f;
                            const float r0w = (j0 - h.jbase) < h.nr ? r0 : 0.f, r1w = (j0 + 1 - h.jbase) < h.nr ? r1 : 0.f;
                            float tt = c.w0 * r0w;
                            tt = fmaf(c.w1, r1w, tt);
                            tt = tt * c.inv;
                            float &xv = x[((size_t)y * T + zz) * SL + s];
                            const float v = fmaf(tt, beta, xv);
                            xv = v > 0.f ? v : 0.f;
                        }
                    }
                }
                if (kk + 1 < steps) {
                    const int a = (kk + 1) % P;
                    const THdr &h = thdr[(size_t)a * ntiles + k];
                    std::vector<float> wacc((size_t)WAVES * WWIN * SL, 0.f);
                    for (int w = 0; w < WAVES; ++w) {
                        const WHdr &wh = whdr[((size_t)a * ntiles + k) * WAVES + w];
                        for (int g = 0; g < RPW; ++g) {
                            if (mode & 8) {       // run totals (last lane + the one / two lanes before it), one read - add - write per run
                                const int y = w * RPW + g;
                                for (int pass = 0; pass < 2; ++pass) for (int s = 0; s < SL; ++s) for (int l = 0; l < 64; ++l) {
                                    const PCell &c = cells[(size_t)a * npix + (size_t)(ty * T + y) * N + tz * T + l];
                                    if (c.s0 < 0) continue;
                                    const bool tail = l == 63 || cells[(size_t)a * npix + (size_t)(ty * T + y) * N + tz * T + l + 1].s0 != c.s0;
                                    if (!tail) continue;
                                    auto val = [&](int ll) { const PCell &cc = cells[(size_t)a * npix + (size_t)(ty * T + y) * N + tz * T + ll];
                                                             const volatile float pr = (pass ? cc.w1 : cc.w0) * x[((size_t)y * T + ll) * SL + s]; return (float)pr; };
                                    auto same = [&](int ll) { return ll >= 0 && cells[(size_t)a * npix + (size_t)(ty * T + y) * N + tz * T + ll].s0 == c.s0; };
                                    float tot = val(l);
                                    if (same(l - 1)) tot = tot + val(l - 1);
                                    if (same(l - 2)) tot = tot + val(l - 2);
                                    if (same(l - 3)) tot = tot + val(l - 3);
                                    float &dst = wacc[((size_t)w * WWIN + (c.s0 - wh.wbase) + pass) * SL + s];
                                    dst = dst + tot;
                                }
                                continue;
                            }
                            for (int pass = 0; pass < 2; ++pass) for (int s = 0; s < SL; ++s) for (int l = 0; l < 64; ++l) {
                                const int y = w * RPW + g;
                                const PCell &c = cells[(size_t)a * npix + (size_t)(ty * T + y) * N + tz * T + l];
                                if (c.s0 < 0) continue;
                                const float xv = x[((size_t)y * T + l) * SL + s];
                                const volatile float prod = (pass ? c.w1 : c.w0) * xv;          // (rounded product, then the add: no contraction)
                                float &dst = wacc[((size_t)w * WWIN + (c.s0 - wh.wbase) + pass) * SL + s];
                                dst = dst + prod;
                            }
                        }
                    }
                    for (int i = 0; i < h.nr; ++i) for (int s = 0; s < SL; ++s) {
                        float tot = 0.f;
                        for (int w = 0; w < WAVES; ++w) {
                            const WHdr &o = whdr[((size_t)a * ntiles + k) * WAVES + w];
                            const int rel = h.jbase + i - o.wbase;
                            if (rel >= 0 && rel < o.wnr) tot += wacc[((size_t)w * WWIN + rel) * SL + s];
                        }
                        last_ts[(size_t)i * SL + s] = tot;
                    }
                    last_a = a;
                    if (steps <= P) {                                               // one sweep: every angle's sums are in the buffer once
                        for (int i = 0; i < h.nr * SL; ++i) {
                            const float g = gts[(((size_t)a * ntiles + k) * MAXWIN) * SL + i];
                            if (!(g == last_ts[i])) { if (bad_ts < 5) printf("  tile %d angle %d sum %d: gpu %.9g cpu %.9g\n", k, a, i, g, last_ts[i]); ++bad_ts; }
                        }
                    }
                }
            }
            (void)last_a;
            for (int y = 0; y < T; ++y) for (int zz = 0; zz < T; ++zz) for (int s = 0; s < SL; ++s) {
                const float g = got[(((size_t)(ty * T + y)) * N + tz * T + zz) * SL + s], c = x[((size_t)y * T + zz) * SL + s];
                if (!(g == c)) { if (bad < 5) printf("  tile %d pixel (%d, %d) slice %d: gpu %.9g cpu %.9g\n", k, y, zz, s, g, c); ++bad; }
            }
        }
        printf("CPU replay of 3 tiles: %d voxels and %d tile sums differ\n", bad, bad_ts);
    }
    printf("RESULT N %d P %d mode %d: best %.3f ms, %.2f us per angle and 16-slice chunk (compute side)\n", N, P, mode, best, 1000.0 * best / steps);
    return bad || bad_ts ? 3 : 0;
}
