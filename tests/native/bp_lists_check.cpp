// Host-only check of the entry lists of the all-angle back-projector k_bp_list (sysmat.cpp: build_bp_lists).
// Replays what the kernel does with the tables -- per (tile, stage, wave): whole batches of PAIRS {window byte offset | accumulator
// register, weight, accumulator register, weight} (two weights on one row, read once), the offset resolved through the stage's staged windows (LDS buffer = stage parity, slot = angle % BL_A, row =
// ray - first ray of the window), the accumulator through the wave's pixel numbering (Tables::bl_pixel) -- and requires that the entries are exactly
// the nonzero weights of the cell table: every (pixel, angle, ray) once, a pixel's entries in the order of k_bp_all (angles ascending,
// first ray before second: rows ascend inside an angle), padding only with weight 0 on a staged row.  In double precision on one slice it also compares the
// back projection with the plain CSR transpose product.
// Usage: bp_lists_check N P max_abs_angle_deg [quiet]
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include "sysmat.h"
using namespace tomo;

#define REQUIRE(c, ...) do { if (!(c)) { std::printf("FAIL %s:%d: ", __FILE__, __LINE__); std::printf(__VA_ARGS__); std::printf("\n"); return 1; } } while (0)

int main(int argc, char **argv)
{
    const int N = argc > 1 ? std::atoi(argv[1]) : 50, P = argc > 2 ? std::atoi(argv[2]) : 9;
    const double amax = argc > 3 ? std::atof(argv[3]) : 70.0;
    const bool quiet = argc > 4;
    constexpr int TY = Tables::BL_TY, TZ = Tables::BL_TZ, WAVES = Tables::BL_WAVES, A = Tables::BL_A, MAXR = Tables::BL_MAXR, ROWB = Tables::BL_ROWB,
                  BATCH = Tables::BL_BATCH, REGS = Tables::BL_REGS, PPW = TY * TZ / WAVES;
    std::vector<double> ang(P);
    for (int i = 0; i < P; ++i) ang[i] = (P > 1 ? -amax + 2 * amax * i / (P - 1) : 0.0) * M_PI / 180;
    Coo m; build_parallel_ray(N, P, ang.data(), m); sort_rows(m);
    Tables t; std::string why;
    REQUIRE(build_tables(m, N, P, t, why), "build_tables: %s", why.c_str());
    build_bp_lists(N, P, TY, TZ, A, MAXR, ROWB, WAVES, BATCH, REGS, t);
    REQUIRE(t.bl_ok, "build_bp_lists gave up");
    const int64_t nrows = (int64_t)N * P, npix = (int64_t)N * N;
    const int tiles_z = (N + TZ - 1) / TZ, ntiles = ((N + TY - 1) / TY) * tiles_z, nstage = (P + A - 1) / A;
    REQUIRE(t.bl_ptr.size() == (size_t)ntiles * nstage * WAVES + 1 && t.bl_win.size() == (size_t)ntiles * P, "table sizes");
    std::mt19937 rng(5); std::uniform_real_distribution<double> U(-1.0, 1.0);
    std::vector<double> r(nrows), want(npix, 0.0), got(npix, 0.0);
    for (auto &v : r) v = U(rng);
    for (int64_t row = 0; row < nrows; ++row) for (int64_t k = m.ptr[row]; k < m.ptr[row + 1]; ++k) want[m.col[k]] += (double)m.val[k] * r[row];
    // what is left to see of every cell: 0 = nothing seen, 1 = first ray seen, 2 = both
    std::vector<uint8_t> seen((size_t)P * npix, 0);
    uint64_t real = 0, slots = 0;
    for (int k = 0; k < ntiles; ++k) {
        const int y0 = (k / tiles_z) * TY, z0 = (k % tiles_z) * TZ;
        for (int s = 0; s < nstage; ++s)
            for (int w = 0; w < WAVES; ++w) {
                const size_t li = ((size_t)k * nstage + s) * WAVES + w;
                REQUIRE(t.bl_ptr[li] <= t.bl_ptr[li + 1], "list bounds");
                int last_angle = -1;
                uint32_t last_ray = 0;
                for (size_t pr = (size_t)t.bl_ptr[li] * BATCH; pr < (size_t)t.bl_ptr[li + 1] * BATCH; ++pr) {
                    // one pair: {row offset | register 0, weight 0} {register 1, weight 1}, both weights on the SAME row
                    const uint32_t e0 = (uint32_t)t.bl_ent[2 * pr];
                    const uint32_t off = e0 & ~(uint32_t)(ROWB - 1);
                    const uint32_t buf = off / (A * MAXR * ROWB), in = off % (A * MAXR * ROWB), slot = in / (MAXR * ROWB), row = in % (MAXR * ROWB) / ROWB;
                    REQUIRE(buf == (uint32_t)(s & 1), "pair of stage %d in LDS buffer %u", s, buf);
                    const int i = s * A + (int)slot;
                    REQUIRE(i < P, "pair of an angle beyond the last");
                    const uint32_t win = t.bl_win[(size_t)k * P + i], lo = win & 0xFFFFu, nr = win >> 16;
                    REQUIRE(row < nr, "row %u outside the staged window of %u rows (tile %d angle %d)", row, nr, k, i);
                    const uint32_t ray = lo + row;
                    REQUIRE((uint32_t)(t.bl_ent[2 * pr + 1]) < (uint32_t)ROWB, "second half of a pair carries no offset");
                    for (int h = 0; h < 2; ++h, ++slots) {
                        const uint32_t reg = (uint32_t)t.bl_ent[2 * pr + h] & (uint32_t)(ROWB - 1), wb = (uint32_t)(t.bl_ent[2 * pr + h] >> 32);
                        float wt; std::memcpy(&wt, &wb, 4);
                        REQUIRE(reg % REGS == 0 && reg / REGS < (uint32_t)PPW, "accumulator register %u", reg);
                        if (wt == 0.f) { REQUIRE(reg == 0, "padding must go to accumulator 0"); continue; }
                        REQUIRE(i > last_angle || (i == last_angle && ray >= last_ray), "angles, and rows inside an angle, must ascend");
                        last_angle = i; last_ray = ray;
                        int ly, lz;
                        Tables::bl_pixel(w, (int)(reg / REGS), ly, lz);
                        const int y = y0 + ly, z = z0 + lz;
                        REQUIRE(y < N && z < N, "entry of a pixel outside the image");
                        const int64_t p = (int64_t)y * N + z;
                        const Cell &c = t.cell[(size_t)i * npix + p];
                        uint8_t &st = seen[(size_t)i * npix + p];
                        if (c.w0 != 0.f && st == 0 && ray == c.r0 && wt == c.w0) st = (c.w1 != 0.f) ? 1 : 2;
                        else if (c.w1 != 0.f && st == (c.w0 != 0.f ? 1 : 0) && ray == c.r1 && wt == c.w1) st = 2;
                        else REQUIRE(false, "entry (pixel %lld angle %d ray %u w %g) is not the next weight of its cell {%u %g %u %g} (state %d)",
                                     (long long)p, i, ray, wt, c.r0, c.w0, c.r1, c.w1, (int)st);
                        got[p] += (double)wt * r[(int64_t)i * N + ray];
                        ++real;
                    }
                }
            }
    }
    // every nonzero weight was seen; every earlier angle of a pixel comes earlier in ITS wave's lists by construction (stages ascend)
    uint64_t nnz = 0;
    for (int i = 0; i < P; ++i)
        for (int64_t p = 0; p < npix; ++p) {
            const Cell &c = t.cell[(size_t)i * npix + p];
            const int need = (c.w0 != 0.f || c.w1 != 0.f) ? 2 : 0;
            nnz += (c.w0 != 0.f) + (c.w1 != 0.f);
            REQUIRE(seen[(size_t)i * npix + p] == need, "cell (angle %d pixel %lld) not fully listed", i, (long long)p);
        }
    REQUIRE(real == nnz, "listed %llu of %llu weights", (unsigned long long)real, (unsigned long long)nnz);
    double err = 0, nrm = 0;
    for (int64_t p = 0; p < npix; ++p) { err += (got[p] - want[p]) * (got[p] - want[p]); nrm += want[p] * want[p]; }
    REQUIRE(std::sqrt(err) <= 1e-12 * std::sqrt(nrm) + 1e-300, "back projection differs: %g", std::sqrt(err / (nrm + 1e-300)));
    if (!quiet) std::printf("N %d P %d: %llu weights in %llu slots of %llu row reads (padding %.1f %%), %.2f weights per pixel and angle\n", N, P,
                            (unsigned long long)real, (unsigned long long)slots, (unsigned long long)(slots / 2), 100.0 * (slots - real) / slots,
                            (double)real / ((double)npix * P));
    std::printf("ok\n");
    return 0;
}
