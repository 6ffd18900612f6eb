// Standalone harness of the volume-resident SART sweep (tomo_tv_amd/csrc/sart_resident.hip.h): builds the parallel-ray matrix and
// the resident tables, runs k_sart_resident on pseudo-random data, replays the sweep on the CPU IN THE KERNEL'S ORDER of operations
// (fmaf where the kernel has an FMA) for a few slices and compares bit for bit, and times the sweep.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I../../tomo_tv_amd/csrc resident_probe.hip ../../tomo_tv_amd/csrc/sysmat.cpp \
//         ../../tomo_tv_amd/csrc/resident.cpp -lpthread -o resident_probe
//   ./resident_probe [N=512] [P=90] [nslice=64] [sweeps=1] [reps=3]
#include "kernels.hip.h"
#include "sart_resident.hip.h"
#include "sysmat.h"
#include "resident.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

using namespace tomo;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static inline float hashf(uint64_t i, uint32_t salt)
{
    uint64_t z = (i + 0x9E3779B97F4A7C15ull * (salt + 1));
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
    return (float)(z >> 40) * (1.0f / 16777216.0f);
}
static inline float bitsf(uint32_t b) { float f; std::memcpy(&f, &b, 4); return f; }

int main(int argc, char **argv)
{
    const int N = argc > 1 ? atoi(argv[1]) : 512, P = argc > 2 ? atoi(argv[2]) : 90, nx = argc > 3 ? atoi(argv[3]) : 64;
    const int sweeps = argc > 4 ? atoi(argv[4]) : 1, reps = argc > 5 ? atoi(argv[5]) : 3, tracked = argc > 6 ? atoi(argv[6]) : 0;
    const int sx = (nx + 63) / 64 * 64, nchunk = sx / 64, steps = sweeps * P;
    const float beta = 0.7f;
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("%s: %d CUs; N %d P %d slices %d (%d chunks) steps %d\n", prop.name, prop.multiProcessorCount, N, P, nx, nchunk, steps);
    std::vector<double> ang(P);
    for (int i = 0; i < P; ++i) ang[i] = (P > 1 ? -70.0 + 140.0 * i / (P - 1) : 0.0) * M_PI / 180.0;
    Coo m; build_parallel_ray(N, P, ang.data(), m); sort_rows(m);
    Tables t; std::string err;
    if (!build_tables(m, N, P, t, err)) { printf("build_tables: %s\n", err.c_str()); return 1; }
    Resident R; build_sart_resident(N, P, t, prop.multiProcessorCount, R);
    if (!R.ok) { printf("resident tables: %s\n", R.why.c_str()); return 1; }
    const int ntiles = R.ntiles;
    int ngrp = std::max(1, std::min(prop.multiProcessorCount / ntiles, nchunk));
    printf("tiles %d, rpt %d, groups %d; cell table %.1f MB\n", ntiles, R.rpt, ngrp, R.cell.size() * 4 / 1e6);
    const size_t npix = (size_t)N * N, nrows = (size_t)N * P;
    std::vector<float> x0(npix * sx), b(nrows * sx);
    for (size_t i = 0; i < x0.size(); ++i) x0[i] = ((int)(i % sx) < nx) ? hashf(i, 1) : 0.f;
    for (size_t r = 0; r < nrows; ++r)
        for (int s = 0; s < sx; ++s) b[r * sx + s] = s < nx ? t.rowsum[r] * (0.3f + 0.4f * hashf(r * sx + s, 2)) : 0.f;
    float *dx, *db, *drs; RsHdrD *dh; uint4 *dcell; uint2 *dts; uint16_t *drl; rs_u64 *dpb, *drb; int *dab; double *dpart;
    CK(hipMalloc(&dx, x0.size() * 4)); CK(hipMalloc(&db, b.size() * 4)); CK(hipMalloc(&drs, nrows * 4));
    CK(hipMalloc(&dh, R.hdr.size() * 32)); CK(hipMalloc(&dcell, R.cell.size() * 4)); CK(hipMalloc(&dts, R.ts.size())); CK(hipMalloc(&drl, R.rl.size() * 2));
    const size_t pbn = (size_t)ngrp * ntiles * RS_MAXWIN * 64, rbn = (size_t)ngrp * P * N * 64;
    CK(hipMalloc(&dpb, pbn * 8)); CK(hipMalloc(&drb, rbn * 8)); CK(hipMalloc(&dab, 4)); CK(hipMalloc(&dpart, NPART * 8));
    CK(hipMemset(dpb, 0, pbn * 8)); CK(hipMemset(drb, 0, rbn * 8)); CK(hipMemset(dab, 0, 4)); CK(hipMemset(dpart, 0, NPART * 8));
    CK(hipMemcpy(db, b.data(), b.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(drs, t.rowsum.data(), nrows * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dh, R.hdr.data(), R.hdr.size() * 32, hipMemcpyHostToDevice)); CK(hipMemcpy(dcell, R.cell.data(), R.cell.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dts, R.ts.data(), R.ts.size(), hipMemcpyHostToDevice)); CK(hipMemcpy(drl, R.rl.data(), R.rl.size() * 2, hipMemcpyHostToDevice));
    RsArgs A{};
    A.x = dx; A.b = db; A.rowsum = drs; A.hdr = dh; A.cell = dcell; A.ts = dts; A.rl = drl; A.pb = dpb; A.rb = drb; A.angs = nullptr; A.track = nullptr; A.part = dpart;
    A.abort_word = dab; A.n = N; A.sx = sx; A.np = P; A.ntiles = ntiles; A.tiles = R.tiles; A.rpt = R.rpt; A.steps = steps; A.chunk0 = 0; A.nchunk = nchunk;
    A.spin_limit = 1u << 20; A.beta = beta;
    unsigned *dcommit; int *hdone;      // the commit words and the pinned per-chunk verdicts (round 6: a chunk is stored by all of its workgroups or by none)
    CK(hipMalloc(&dcommit, nchunk * 4)); CK(hipMemset(dcommit, 0, nchunk * 4));
    CK(hipHostMalloc((void **)&hdone, nchunk * sizeof(int), hipHostMallocMapped)); std::memset(hdone, 0, nchunk * sizeof(int));
    A.commit = dcommit; A.done_host = hdone; A.seq = 0; A.commit_base = 0; A.test_fail = 0;
    std::vector<int> hang(steps); for (int k = 0; k < steps; ++k) hang[k] = k % P;
    int *dang; CK(hipMalloc(&dang, steps * 4)); CK(hipMemcpy(dang, hang.data(), steps * 4, hipMemcpyHostToDevice)); A.angs = dang;
    long long *dprof; CK(hipMalloc(&dprof, (2048 + 128) * 8)); CK(hipMemset(dprof, 0, (2048 + 128) * 8)); A.prof = dprof;
    std::vector<float> tk0;
    float *dtk = nullptr;
    if (tracked) {
        tk0.resize(x0.size());
        for (size_t i = 0; i < tk0.size(); ++i) tk0[i] = ((int)(i % sx) < nx) ? hashf(i, 3) : 0.f;
        CK(hipMalloc(&dtk, tk0.size() * 4));
        A.track = dtk;
    }
    unsigned epoch = 0;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> got(x0.size());
    float best = 1e30f;
    for (int rep = 0; rep < reps; ++rep) {
        CK(hipMemcpy(dx, x0.data(), x0.size() * 4, hipMemcpyHostToDevice));
        if (tracked) { CK(hipMemcpy(dtk, tk0.data(), tk0.size() * 4, hipMemcpyHostToDevice)); CK(hipMemset(dpart, 0, NPART * 8)); }
        A.epoch0 = epoch; epoch += (unsigned)((nchunk + ngrp - 1) / ngrp) * (unsigned)steps;
        ++A.seq; A.commit_base = (A.seq - 1) * (unsigned)ntiles;
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_sart_resident, dim3(ntiles * ngrp), dim3(RS_THREADS), 0, 0, A);
        CK(hipEventRecord(e1));
        CK(hipGetLastError());
        CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms);
        int ab; CK(hipMemcpy(&ab, dab, 4, hipMemcpyDeviceToHost));
        printf("rep %d: %.3f ms = %.2f us per angle and chunk-round; abort word %d\n", rep, ms, 1000.0 * ms / steps / ((nchunk + ngrp - 1) / ngrp), ab);
        if (ab) { printf("ABORTED\n"); return 2; }
        for (int c = 0; c < nchunk; ++c) if (hdone[c] != (int)A.seq) { printf("chunk %d did not commit (%d, launch %u)\n", c, hdone[c], A.seq); return 2; }
    }
    CK(hipMemcpy(got.data(), dx, got.size() * 4, hipMemcpyDeviceToHost));
#ifdef RS_PROF
    {
        std::vector<long long> hp(256 * 8); CK(hipMemcpy(hp.data(), dprof, hp.size() * 8, hipMemcpyDeviceToHost));
        const int nwg = ntiles * ngrp; const double per = 0.01 / nwg / steps / ((nchunk + ngrp - 1) / ngrp);
        double av[8] = {0};
        for (int w = 0; w < nwg; ++w) for (int q = 0; q < 8; ++q) av[q] += hp[w * 8 + q] * per;
        if (steps > 41 && ntiles * ngrp > 37) {
            std::vector<long long> tl(128); CK(hipMemcpy(tl.data(), dprof + 2048, 128 * 8, hipMemcpyDeviceToHost));
            long long t0 = tl[0]; for (int w = 0; w < 16; ++w) t0 = std::min(t0, tl[w * 8]);
            printf("TIMELINE step 40, workgroup 37, us after the first wave had its rows (per wave: rows in | BP starts | BP done | FP done | sums published | reducer polled | final done)\n");
            for (int w = 0; w < 16; ++w) { printf("  wave %2d:", w); for (int q = 0; q < 7; ++q) printf(" %6.2f", (tl[w * 8 + q] - t0) * 0.01); printf("\n"); }
        }
        { std::vector<long long> ck(2); CK(hipMemcpy(ck.data(), dprof + 2048 + 126, 16, hipMemcpyDeviceToHost));
          printf("CLOCK s_memtime %lld ticks over %.1f us = %.0f MHz\n", ck[0], ck[1] * 0.01, ck[0] / (ck[1] * 0.01)); }
        printf("PROF us per angle (wave 0, mean over workgroups): wait rows %.2f | barrier+rows->regs %.2f | BP %.2f | FP %.2f | block sums->LDS, barrier, tile sums, publish %.2f | reducer poll %.2f | barrier, final, publish %.2f | loop head %.2f\n",
               av[0], av[1], av[2], av[3], av[4], av[5], av[6], av[7]);
    }
#endif
    int track_bad = 0;
    if (tracked) {
        std::vector<float> tk(x0.size()); CK(hipMemcpy(tk.data(), dtk, tk.size() * 4, hipMemcpyDeviceToHost));
        std::vector<double> hpart(NPART); CK(hipMemcpy(hpart.data(), dpart, NPART * 8, hipMemcpyDeviceToHost));
        double gsum = 0, want = 0; size_t ndiff = 0;
        for (double v : hpart) gsum += v;
        for (size_t i = 0; i < got.size(); ++i) { float d = got[i] - tk0[i]; want += (double)(d * d); if (!(tk[i] == got[i])) ++ndiff; }
        printf("TRACK: sum (x - snapshot)^2 gpu %.10g cpu %.10g (rel %.2e); snapshot differs from x in %zu places\n", gsum, want, fabs(gsum - want) / want, ndiff);
        track_bad = (ndiff != 0) || fabs(gsum - want) > 1e-9 * want;
    }
    // ---- CPU replay in the kernel's order, selected slices
    int rpt2 = 1; while (rpt2 < R.rpt && rpt2 < 16) rpt2 *= 2;
    const int wpr = 16 / rpt2, cpw = 32 / wpr;
    const int check[] = {0, 17, 63, nx - 1};
    int nbad_total = 0;
    double worst = 0;
    for (int ci = 0; ci < 4; ++ci) {
        const int s = check[ci];
        if (s < 0 || s >= nx || (ci > 0 && s == check[ci - 1])) continue;
        std::vector<float> x(npix);
        for (size_t p = 0; p < npix; ++p) x[p] = x0[p * sx + s];
        std::vector<float> pbuf((size_t)ntiles * 16 * 16), tsum((size_t)ntiles * 48), r(N);
        auto fp = [&](int a) {
            for (int k = 0; k < ntiles; ++k) for (int w = 0; w < 16; ++w) {
                float sl[16] = {0};
                const uint32_t *c = R.cell.data() + (((size_t)a * ntiles + k) * 16 + w) * 256;
                for (int q = 0; q < 64; ++q) {
                    int ly, lz; Resident::pixel(w, q, ly, lz);
                    const int y = (k / R.tiles) * 32 + ly, z = (k % R.tiles) * 32 + lz;
                    const float xv = (y < N && z < N) ? x[(size_t)y * N + z] : 0.f;
                    sl[c[q * 4]] = fmaf(bitsf(c[q * 4 + 1]), xv, sl[c[q * 4]]);
                    sl[c[q * 4] + 1] = fmaf(bitsf(c[q * 4 + 2]), xv, sl[c[q * 4] + 1]);
                }
                std::memcpy(&pbuf[((size_t)k * 16 + w) * 16], sl, 64);
            }
            for (int k = 0; k < ntiles; ++k) {
                const Resident::Hdr &h = R.hdr[(size_t)a * ntiles + k];
                for (int i = 0; i < h.nrays; ++i) {
                    const uint8_t *e = R.ts.data() + (((size_t)a * ntiles + k) * 48 + i) * 8;
                    float acc = pbuf[(size_t)k * 256 + e[0]];
                    for (int c8 = 1; c8 < 8; ++c8) acc += pbuf[(size_t)k * 256 + e[c8]];
                    tsum[(size_t)k * 48 + i] = acc;
                }
            }
            for (int j = 0; j < N; ++j) {
                const uint16_t *list = R.rl.data() + ((size_t)a * N + j) * 32;
                float tot = 0.f;
                for (int sub = 0; sub < wpr; ++sub) {
                    float acc = 0.f;
                    for (int e = 0; e < cpw; ++e) { int id = list[sub * cpw + e]; if (id == 0xFFFF) break; acc += tsum[id]; }
                    tot = sub == 0 ? acc : tot + acc;
                }
                const size_t row = (size_t)a * N + j;
                const float rs = t.rowsum[row];
                r[j] = rs > 0.f ? (b[row * sx + s] - tot) / rs : 0.f;
            }
        };
        auto bp = [&](int a) {
            for (int k = 0; k < ntiles; ++k) {
                const Resident::Hdr &h = R.hdr[(size_t)a * ntiles + k];
                for (int w = 0; w < 16; ++w) {
                    float rr[16];
                    for (int sl = 0; sl < 16; ++sl) rr[sl] = (sl < 14 && h.dw[w] + sl < h.nrays) ? r[h.jbase + h.dw[w] + sl] : 0.f;
                    const uint32_t *c = R.cell.data() + (((size_t)a * ntiles + k) * 16 + w) * 256;
                    for (int q = 0; q < 64; ++q) {
                        int ly, lz; Resident::pixel(w, q, ly, lz);
                        const int y = (k / R.tiles) * 32 + ly, z = (k % R.tiles) * 32 + lz;
                        if (y >= N || z >= N) continue;
                        float tt = bitsf(c[q * 4 + 1]) * rr[c[q * 4]];
                        tt = fmaf(bitsf(c[q * 4 + 2]), rr[c[q * 4] + 1], tt);
                        tt = tt * bitsf(c[q * 4 + 3]);
                        float v = fmaf(tt, beta, x[(size_t)y * N + z]);
                        x[(size_t)y * N + z] = v > 0.f ? v : 0.f;
                    }
                }
            }
        };
        fp(0);
        for (int k = 0; k < steps; ++k) { bp(k % P); if (k + 1 < steps) fp((k + 1) % P); }
        int nbad = 0; double num = 0, den = 0;
        for (size_t p = 0; p < npix; ++p) {
            const float g = got[p * sx + s];
            if (!(g == x[p])) { if (nbad < 5) printf("  slice %d pixel %zu: gpu %.9g cpu %.9g\n", s, p, g, x[p]); ++nbad; }
            num += (double)(g - x[p]) * (g - x[p]); den += (double)x[p] * x[p];
        }
        const double rel = std::sqrt(num / std::max(den, 1e-300));
        worst = std::max(worst, rel);
        printf("slice %d: %d of %zu voxels differ from the CPU replay; rel L2 %.3e\n", s, nbad, npix, rel);
        nbad_total += nbad;
    }
    printf("RESULT N %d P %d nx %d: best %.3f ms, %.2f us per angle and chunk-round, mismatching voxels %d, worst rel %.3e\n", N, P, nx, best,
           1000.0 * best / steps / ((nchunk + ngrp - 1) / ngrp), nbad_total, worst);
    return nbad_total ? 3 : track_bad ? 4 : 0;
}
