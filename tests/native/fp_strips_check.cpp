// Host-only check of the sheared-strip tables of the all-angle forward projector (sysmat.cpp: build_fp_strips).
// Replays what k_fp_strip / k_fp_tile_reduce do with the tables -- per item and lane group: the entry stream walked tile by
// tile and slot by slot with the per-(tile, wave, slot) batch counts, K accumulators, a partial sum emitted at every flagged
// batch -- in double precision on one slice, and compares with the plain CSR product.  Also prints the tables' statistics
// (passes, partial sums per ray, padding, accumulator slots, volume pixels staged).
// Usage: fp_strips_check N P max_abs_angle_deg [nchunk] [quiet]
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include "sysmat.h"
using namespace tomo;

#define REQUIRE(c, ...) do { if (!(c)) { std::printf("FAIL %s:%d: ", __FILE__, __LINE__); std::printf(__VA_ARGS__); std::printf("\n"); return 1; } } while (0)

int main(int argc, char **argv)
{
    int N = argc > 1 ? std::atoi(argv[1]) : 50, P = argc > 2 ? std::atoi(argv[2]) : 9;
    double amax = argc > 3 ? std::atof(argv[3]) : 70.0;
    const int nchunk = argc > 4 ? std::atoi(argv[4]) : 8;
    const bool quiet = argc > 5;
    const int PIXB = 256;
    constexpr int W = Tables::FS_W, H = Tables::FS_H, WAVES = Tables::FS_WAVES, GROUPS = Tables::FS_GROUPS, NB = Tables::TILE_BATCH;
    std::vector<double> ang(P);
    for (int i = 0; i < P; ++i) ang[i] = (P > 1 ? -amax + 2 * amax * i / (P - 1) : 0.0) * M_PI / 180;
    Coo m; build_parallel_ray(N, P, ang.data(), m); sort_rows(m);
    Tables t; std::string why;
    REQUIRE(build_fp_strips(m, N, P, PIXB, nchunk, t, why), "build_fp_strips: %s", why.c_str());
    const int64_t nrows = (int64_t)N * P, npix = (int64_t)N * N, nnz = m.ptr[nrows];
    std::mt19937 rng(7); std::uniform_real_distribution<double> U(0.0, 1.0);
    std::vector<double> x(npix), g(nrows, 0.0);
    for (auto &v : x) v = U(rng);
    for (int64_t r = 0; r < nrows; ++r) for (int64_t k = m.ptr[r]; k < m.ptr[r + 1]; ++k) g[r] += (double)m.val[k] * x[m.col[k]];
    std::vector<double> part(t.fs_nseg, 0.0);
    std::vector<int> seen(t.fs_nseg, 0);
    int64_t real = 0;
    uint64_t wave_batches_max = 0, wave_batches_sum = 0;
    for (size_t it = 0; it < t.fs_item.size(); ++it) {
        const Tables::FsItem &I = t.fs_item[it];
        REQUIRE(I.pass >= 0 && I.pass < t.fs_npass, "pass out of range");
        const int o = t.fs_orient[I.pass];
        const int32_t *sh = t.fs_shift.data() + (size_t)I.pass * N;
        for (int grp = 0; grp < GROUPS; ++grp) {
            const int wave = grp >> 2;
            size_t b = t.fs_gstart[I.g0 + grp];
            uint32_t seg = t.fs_gseg0[I.g0 + grp];
            double acc[16] = {0};
            for (uint32_t tt = 0; tt < I.ntiles; ++tt) {
                const uint8_t *cnt = t.fs_cnt.data() + ((size_t)I.cnt0 + (size_t)tt * WAVES + wave) * 16;
                for (int k = 0; k < 16; ++k) {
                    REQUIRE(k < t.fs_kused || cnt[k] == 0, "slot %d beyond fs_kused = %d has batches", k, t.fs_kused);
                    const int h = cnt[k], units = (h + 1) >> 1;              // half batches; a trailing half batch is a unit stored twice
                    for (int q = 0; q < units; ++q, ++b) {
                        const bool half = q >= (h >> 1);
                        bool last = false;
                        for (int j = 0; j < NB; ++j) {
                            const size_t at = b * NB + j;
                            uint32_t off = Tables::fs_off_of(t.fs_ent[at]); float w = Tables::fs_w_of(t.fs_ent[at]);
                            if (j == 0) last = off >> 31; else REQUIRE((off >> 31) == (uint32_t)last, "flag differs inside batch %zu", b);
                            if (half && j >= NB / 2) { REQUIRE(t.fs_ent[at] == t.fs_ent[at - NB / 2], "half batch %zu is not stored twice", b); continue; }
                            uint32_t lp = (off & 0x7FFFFFFFu) / PIXB;
                            REQUIRE((off & 0x7FFFFFFFu) % PIXB == 0 && lp <= (uint32_t)(W * H), "bad offset");
                            if (lp == (uint32_t)(W * H)) { REQUIRE(w == 0.f, "padding entry with weight"); continue; }
                            int u = (int)(I.tile0 + tt) * H + (int)(lp / W), v;
                            REQUIRE(u < N, "march coordinate %d outside the image", u);
                            v = I.v0 + sh[u] + (int)(lp % W);
                            REQUIRE(v >= 0 && v < N, "cross coordinate %d outside the image", v);
                            acc[k] += (double)w * x[o ? (int64_t)v * N + u : (int64_t)u * N + v];
                            ++real;
                        }
                        if (last) { REQUIRE(seg < t.fs_nseg, "segment id out of range"); part[seg] = acc[k]; seen[seg]++; acc[k] = 0.0; ++seg; }
                    }
                }
            }
            for (int k = 0; k < 16; ++k) REQUIRE(acc[k] == 0.0, "item %zu group %d slot %d ends with an unflushed sum", it, grp, k);
            REQUIRE(b == t.fs_gstart[I.g0 + grp + 1], "stream of item %zu group %d ends at batch %zu, the next starts at %u", it, grp, b, t.fs_gstart[I.g0 + grp + 1]);
            REQUIRE(seg == t.fs_gseg0[I.g0 + grp + 1], "segments of item %zu group %d end at %u, the next start at %u", it, grp, seg, t.fs_gseg0[I.g0 + grp + 1]);
            if ((grp & 3) == 0) {
                uint64_t nb = t.fs_gstart[I.g0 + grp + 1] - t.fs_gstart[I.g0 + grp];
                wave_batches_sum += nb;
                wave_batches_max = std::max(wave_batches_max, nb);
            }
        }
    }
    REQUIRE(real == nnz, "streams hold %ld entries, matrix has %ld", (long)real, (long)nnz);
    for (uint32_t s = 0; s < t.fs_nseg; ++s) REQUIRE(seen[s] == 1, "partial sum %u written %d times", s, seen[s]);
    std::vector<int> used(t.fs_nseg, 0);
    double worst = 0;
    for (int64_t r = 0; r < nrows; ++r) {
        double s = 0;
        for (uint32_t k = t.fs_rseg_ptr[r]; k < t.fs_rseg_ptr[r + 1]; ++k) { REQUIRE(t.fs_rseg_idx[k] < t.fs_nseg, "row list id out of range"); s += part[t.fs_rseg_idx[k]]; used[t.fs_rseg_idx[k]]++; }
        worst = std::max(worst, std::fabs(s - g[r]) / (1.0 + std::fabs(g[r])));
    }
    for (uint32_t s = 0; s < t.fs_nseg; ++s) REQUIRE(used[s] == 1, "partial sum %u used %d times by the row lists", s, used[s]);
    REQUIRE(worst < 1e-9, "strip replay differs from the CSR product by %.3e", worst);
    if (!quiet) {
        // critical path of an item = its slowest wave; a wave's work = its batches (all four lane groups run them together)
        uint64_t crit = 0, all = 0;
        for (size_t it = 0; it < t.fs_item.size(); ++it) {
            const Tables::FsItem &I = t.fs_item[it];
            uint64_t mx = 0;
            for (int w = 0; w < WAVES; ++w) { uint64_t nb = t.fs_gstart[I.g0 + 4 * w + 1] - t.fs_gstart[I.g0 + 4 * w]; mx = std::max(mx, nb); all += nb; }
            crit += mx * WAVES;
        }
        // a workgroup meets at a barrier after every tile: the tile costs its slowest wave
        uint64_t tile_crit = 0, tile_all = 0, empty_slots = 0, used_slots = 0;
        for (size_t it = 0; it < t.fs_item.size(); ++it) {
            const Tables::FsItem &I = t.fs_item[it];
            for (uint32_t tt = 0; tt < I.ntiles; ++tt) {
                uint64_t mx = 0;
                for (int w = 0; w < WAVES; ++w) {
                    const uint8_t *cnt = t.fs_cnt.data() + ((size_t)I.cnt0 + (size_t)tt * WAVES + w) * 16;
                    uint64_t nb = 0;
                    for (int k = 0; k < 16; ++k) { nb += cnt[k]; if (k < t.fs_kused) { if (cnt[k]) ++used_slots; else ++empty_slots; } }
                    mx = std::max(mx, nb); tile_all += nb;
                }
                tile_crit += mx * WAVES;
            }
        }
        std::printf("per-tile wave balance %.3f (sum over tiles of the mean wave / the slowest wave); slots in use per tile %.2f of %d\n",
                    (double)tile_all / tile_crit, (double)used_slots / (used_slots + empty_slots) * t.fs_kused, t.fs_kused);
        std::printf("N=%d P=%d: %d passes, %zu items, K used %d; partial sums %u = %.2f per ray; entries %ld, padded slots %lu (fill %.3f); "
                    "wave balance inside items %.3f; volume staged %.2f x\n",
                    N, P, t.fs_npass, t.fs_item.size(), t.fs_kused, t.fs_nseg, (double)t.fs_nseg / nrows, (long)nnz, (unsigned long)t.fs_slots,
                    (double)nnz / t.fs_slots, (double)all / crit, (double)t.fs_staged_pixels / npix);
        for (int ps = 0; ps < t.fs_npass; ++ps) {
            int cnt = 0; uint64_t tiles = 0;
            for (auto &I : t.fs_item) if (I.pass == ps) { ++cnt; tiles += I.ntiles; }
            std::printf("  pass %d: orientation %d, shift %d..%d, %d strips, %lu tiles\n", ps, t.fs_orient[ps], t.fs_shift[(size_t)ps * N], t.fs_shift[(size_t)ps * N + N - 1], cnt, (unsigned long)tiles);
        }
    }
    std::printf("ok\n");
    return 0;
}
