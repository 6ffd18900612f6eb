// Host-only check of the entry lists of the all-angle forward projector k_fp_list (sysmat.cpp: build_fp_lists).
// Replays what the kernel does with the tables -- per item: tiles of FL_TH march steps staged alternately into two LDS buffers, per
// (tile, wave) whole batches of entries {byte offset of the pixel in the staged tile | accumulator register, weight} and then the flush
// records {accumulator register, partial-sum id} of the rays that leave the strip -- in double precision on one slice, sums every ray's
// partial sums in the order of its row list, and compares with the plain CSR product.  Also: every accumulator is clean when an item
// ends, every partial sum is written exactly once and belongs to exactly one ray.
// Usage: fp_lists_check N P max_abs_angle_deg [quiet]
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
    constexpr int W = Tables::FL_W, TH = Tables::FL_TH, WAVES = Tables::FL_WAVES, ACC = Tables::FL_ACC, BATCH = Tables::FL_BATCH, PIXB = Tables::FL_PIXB,
                  REGS = Tables::FL_REGS;
    std::vector<double> ang(P);
    for (int i = 0; i < P; ++i) ang[i] = (P > 1 ? -amax + 2 * amax * i / (P - 1) : 0.0) * M_PI / 180;
    Coo m; build_parallel_ray(N, P, ang.data(), m); sort_rows(m);
    Tables t; std::string why;
    REQUIRE(build_fp_lists(m, N, P, t, why), "build_fp_lists: %s", why.c_str());
    const int64_t nrows = (int64_t)N * P, npix = (int64_t)N * N, nnz = m.ptr[nrows];
    std::mt19937 rng(7); std::uniform_real_distribution<double> U(0.0, 1.0);
    std::vector<double> x(npix), g(nrows, 0.0);
    for (auto &v : x) v = U(rng);
    for (int64_t r = 0; r < nrows; ++r) for (int64_t k = m.ptr[r]; k < m.ptr[r + 1]; ++k) g[r] += (double)m.val[k] * x[m.col[k]];
    std::vector<double> part(t.fl_nseg, 0.0);
    std::vector<int> written(t.fl_nseg, 0);
    uint64_t real = 0, slots = 0;
    std::vector<double> tile((size_t)2 * W * TH);
    for (size_t it = 0; it < t.fl_item.size(); ++it) {
        const Tables::FlItem &I = t.fl_item[it];
        REQUIRE(I.pass >= 0 && I.pass < t.fl_npass && I.ntiles >= 1 && I.ntiles <= 64, "item %zu: pass %d, %u tiles", it, I.pass, I.ntiles);
        const int o = t.fl_orient[I.pass];
        const int32_t *sh = t.fl_shift.data() + (size_t)I.pass * N;
        std::vector<double> acc((size_t)WAVES * ACC, 0.0);
        std::vector<uint8_t> dirty((size_t)WAVES * ACC, 0);
        for (uint32_t tt = 0; tt < I.ntiles; ++tt) {
            // stage the tile into buffer tt & 1 (pixels outside the image read zero)
            for (int q = 0; q < W * TH; ++q) {
                const int u = (int)(I.tile0 + tt) * TH + q / W, vv = I.v0 + sh[std::min(u, N - 1)] + q % W;
                const bool ok = u < N && vv >= 0 && vv < N;
                tile[(size_t)(tt & 1) * W * TH + q] = ok ? x[o ? (int64_t)vv * N + u : (int64_t)u * N + vv] : 0.0;
            }
            for (int w = 0; w < WAVES; ++w) {
                const size_t li = (size_t)I.lp0 + (size_t)tt * WAVES + w;
                for (size_t e = (size_t)t.fl_ptr[li] * BATCH; e < (size_t)t.fl_ptr[li + 1] * BATCH; ++e, ++slots) {
                    const uint32_t e0 = Tables::fs_off_of(t.fl_ent[e]); const float wt = Tables::fs_w_of(t.fl_ent[e]);
                    const uint32_t off = e0 & ~(uint32_t)(PIXB - 1), reg = e0 & (uint32_t)(PIXB - 1);
                    REQUIRE(off % PIXB == 0 && off / PIXB < (uint32_t)(2 * W * TH) && off / PIXB / (W * TH) == (tt & 1), "entry reads pixel image %u of tile %u", off / PIXB, tt);
                    REQUIRE(reg % REGS == 0 && reg / REGS < (uint32_t)ACC, "accumulator register %u", reg);
                    if (wt == 0.f) { REQUIRE(reg == 0, "padding must go to accumulator 0"); continue; }
                    acc[(size_t)w * ACC + reg / REGS] += (double)wt * tile[off / PIXB];
                    dirty[(size_t)w * ACC + reg / REGS] = 1;
                    ++real;
                }
                for (uint32_t fr = t.fl_fptr[li]; fr < t.fl_fptr[li + 1]; ++fr) {
                    const uint32_t reg = (uint32_t)t.fl_flush[fr], id = (uint32_t)(t.fl_flush[fr] >> 32);
                    REQUIRE(id == fr, "partial-sum id %u of flush record %u", id, fr);
                    REQUIRE(reg % REGS == 0 && reg / REGS < (uint32_t)ACC, "flushed register %u", reg);
                    REQUIRE(!written[id], "partial sum %u written twice", id);
                    written[id] = 1;
                    part[id] = acc[(size_t)w * ACC + reg / REGS];
                    acc[(size_t)w * ACC + reg / REGS] = 0.0; dirty[(size_t)w * ACC + reg / REGS] = 0;
                }
            }
        }
        for (size_t k = 0; k < dirty.size(); ++k) REQUIRE(!dirty[k], "item %zu ends with a sum left in accumulator %zu", it, k);
    }
    REQUIRE((int64_t)real == nnz, "lists hold %llu of %lld matrix entries", (unsigned long long)real, (long long)nnz);
    for (uint32_t id = 0; id < t.fl_nseg; ++id) REQUIRE(written[id], "partial sum %u never written", id);
    std::vector<int> used(t.fl_nseg, 0);
    double err = 0, nrm = 0;
    for (int64_t r = 0; r < nrows; ++r) {
        double s = 0;
        for (uint32_t k = t.fl_rseg_ptr[r]; k < t.fl_rseg_ptr[r + 1]; ++k) { const uint32_t id = t.fl_rseg_idx[k]; REQUIRE(id < t.fl_nseg && !used[id], "row list of ray %lld", (long long)r); used[id] = 1; s += part[id]; }
        err += (s - g[r]) * (s - g[r]); nrm += g[r] * g[r];
    }
    for (uint32_t id = 0; id < t.fl_nseg; ++id) REQUIRE(used[id], "partial sum %u belongs to no ray", id);
    REQUIRE(std::sqrt(err) <= 1e-12 * std::sqrt(nrm) + 1e-300, "forward projection differs: %g", std::sqrt(err / (nrm + 1e-300)));
    if (!quiet) {
        size_t nl = 0; for (auto &I : t.fl_item) nl += (size_t)I.ntiles * WAVES;
        std::printf("N %d P %d: %d passes, %zu items, %llu entries in %llu slots (padding %.1f %%), %.2f partial sums per ray, %.2f pixels staged per image pixel\n", N, P,
                    t.fl_npass, t.fl_item.size(), (unsigned long long)real, (unsigned long long)slots, 100.0 * (slots - real) / slots, (double)t.fl_nseg / nrows,
                    (double)t.fl_staged_pixels / npix);
    }
    std::printf("ok\n");
    return 0;
}
