// Host-only check of the tile tables the all-angle projector kernels consume (sysmat.cpp: build_tiles, build_bp_tiles).
// Replays what k_fp_tile / k_fp_tile_reduce and k_bp_tile do with the tables, in double precision on one slice, and
// compares with the plain CSR product.  Usage: sysmat_tables_check N P max_abs_angle_deg
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
    int N = argc > 1 ? std::atoi(argv[1]) : 50, P = argc > 2 ? std::atoi(argv[2]) : 9;
    double amax = argc > 3 ? std::atof(argv[3]) : 89.0;
    const int TY = 32, TZ = 16, PIXB = 256, A = 4, MAXR = 40;   // = FT_TY, FT_TZ, FB_A, FB_MAXR of kernels.hip.h
    std::vector<double> ang(P);
    for (int i = 0; i < P; ++i) ang[i] = (P > 1 ? -amax + 2 * amax * i / (P - 1) : 0.0) * M_PI / 180;
    Coo m; build_parallel_ray(N, P, ang.data(), m); sort_rows(m);
    Tables t; std::string err;
    REQUIRE(build_tables(m, N, P, t, err), "%s", err.c_str());
    build_tiles(m, N, P, TY, TZ, PIXB, t);
    build_bp_tiles(N, P, TY, TZ, A, MAXR, PIXB, A, t);
    const int64_t nrows = (int64_t)N * P, npix = (int64_t)N * N, nnz = m.ptr[nrows];
    const int tiles_z = t.tiles_z, ntiles = t.tiles_y * t.tiles_z, NS = Tables::TILE_SLOTS, NB = Tables::TILE_BATCH;
    std::mt19937 rng(7); std::uniform_real_distribution<double> U(0.0, 1.0);
    std::vector<double> x(npix), g(nrows, 0.0);
    for (auto &v : x) v = U(rng);
    for (int64_t r = 0; r < nrows; ++r) for (int64_t k = m.ptr[r]; k < m.ptr[r + 1]; ++k) g[r] += (double)m.val[k] * x[m.col[k]];
    // ---- forward: partial sums per segment from the streams, then the row lists
    std::vector<double> part(t.tile_nseg, 0.0);
    std::vector<int> seen(t.tile_nseg, 0);
    int64_t real_entries = 0;
    for (int k = 0; k < ntiles; ++k) {
        int y0 = (k / tiles_z) * TY, z0 = (k % tiles_z) * TZ;
        for (int q = 0; q < NS; ++q) {
            size_t slot = (size_t)k * NS + q;
            uint32_t seg = t.tile_slot_seg0[slot];
            double acc = 0.0;
            for (uint32_t b = t.tile_slot_ptr[slot]; b < t.tile_slot_ptr[slot + 1]; ++b) {
                bool last = false;
                for (int j = 0; j < NB; ++j) {
                    uint32_t off = t.tile_off[(size_t)b * NB + j]; float w = t.tile_w[(size_t)b * NB + j];
                    if (j == 0) last = off >> 31; else REQUIRE((off >> 31) == (uint32_t)last, "flag differs inside batch %u", b);
                    uint32_t lp = (off & 0x7FFFFFFFu) / PIXB;
                    REQUIRE((off & 0x7FFFFFFFu) % PIXB == 0 && lp <= (uint32_t)(TY * TZ), "bad offset");
                    if (lp == (uint32_t)(TY * TZ)) { REQUIRE(w == 0.f, "padding entry with weight"); continue; }
                    int y = y0 + lp / TZ, z = z0 + lp % TZ;
                    REQUIRE(y < N && z < N, "entry outside the image");
                    acc += (double)w * x[(int64_t)y * N + z];
                    ++real_entries;
                }
                if (last) { REQUIRE(seg < t.tile_nseg, "segment id out of range"); part[seg] = acc; seen[seg]++; acc = 0.0; ++seg; }
            }
            uint32_t next = slot + 1 < (size_t)ntiles * NS ? t.tile_slot_seg0[slot + 1] : t.tile_nseg;
            REQUIRE(seg == next, "stream %zu ends at segment %u, next stream starts at %u", slot, seg, next);
        }
    }
    REQUIRE(real_entries == nnz, "streams hold %ld entries, matrix has %ld", (long)real_entries, (long)nnz);
    for (uint32_t s = 0; s < t.tile_nseg; ++s) REQUIRE(seen[s] == 1, "segment %u written %d times", s, seen[s]);
    std::vector<int> used(t.tile_nseg, 0);
    double worst = 0;
    for (int64_t r = 0; r < nrows; ++r) {
        double s = 0;
        for (uint32_t k = t.rseg_ptr[r]; k < t.rseg_ptr[r + 1]; ++k) { s += part[t.rseg_idx[k]]; used[t.rseg_idx[k]]++; }
        worst = std::max(worst, std::fabs(s - g[r]));
    }
    for (uint32_t s = 0; s < t.tile_nseg; ++s) REQUIRE(used[s] == 1, "segment %u used by %d rows", s, used[s]);
    REQUIRE(worst < 1e-9, "forward mismatch %g", worst);
    // balance of the 64 streams of a tile
    double imb = 0;
    for (int k = 0; k < ntiles; ++k) {
        uint32_t mx = 0, sum = 0;
        for (int q = 0; q < NS; ++q) { uint32_t l = t.tile_slot_ptr[(size_t)k * NS + q + 1] - t.tile_slot_ptr[(size_t)k * NS + q]; mx = std::max(mx, l); sum += l; }
        if (sum >= 64 * 8) imb = std::max(imb, mx * 64.0 / sum);
    }
    // ---- backward: windows and tile cells against the cell table
    REQUIRE(t.bp_tile_ok, "a ray window exceeds %d rows", MAXR);
    std::vector<double> rr(nrows);
    for (auto &v : rr) v = U(rng);
    const uint32_t zoff = (uint32_t)(A * MAXR) * PIXB;
    double worst_bp = 0; uint32_t maxwin = 0;
    for (int k = 0; k < ntiles; ++k) {
        int y0 = (k / tiles_z) * TY, z0 = (k % tiles_z) * TZ;
        for (int lp = 0; lp < TY * TZ; ++lp) {
            int y = y0 + lp / TZ, z = z0 + lp % TZ;
            double acc = 0, ref = 0;
            for (int i = 0; i < P; ++i) {
                uint32_t w = t.bp_win[(size_t)k * P + i], lo = w & 0xFFFFu, nr = w >> 16;
                maxwin = std::max(maxwin, nr);
                REQUIRE(nr <= (uint32_t)MAXR && lo + nr <= (uint32_t)N, "bad window");
                const Tables::TileCell &c = t.bp_cell[((size_t)k * P + i) * (TY * TZ) + lp];
                for (int h = 0; h < 2; ++h) {
                    uint32_t off = h ? c.off1 : c.off0; float wt = h ? c.w1 : c.w0;
                    if (off == zoff) { REQUIRE(wt == 0.f, "zero row with weight"); continue; }
                    uint32_t slot = off / ((uint32_t)MAXR * PIXB), j = (off / PIXB) % MAXR;
                    REQUIRE(off % PIXB == 0 && slot == (uint32_t)(i % A) && j < nr, "bad cell offset");
                    acc += (double)wt * rr[(int64_t)i * N + lo + j];
                }
                if (y < N && z < N) {
                    const Cell &cc = t.cell[(size_t)i * npix + (int64_t)y * N + z];
                    if (cc.w0 != 0.f) ref += (double)cc.w0 * rr[(int64_t)i * N + cc.r0];
                    if (cc.w1 != 0.f) ref += (double)cc.w1 * rr[(int64_t)i * N + cc.r1];
                }
            }
            worst_bp = std::max(worst_bp, std::fabs(acc - ref));
        }
    }
    REQUIRE(worst_bp == 0.0, "backward mismatch %g", worst_bp);
    // ---- fused SART step: per-angle tile segments and cells
    const int SY = 16, SZ = 16, SMAXR = 26, MS = Tables::ST_MAXSEG;   // = ST_TY, ST_TZ, ST_MAXR of kernels.hip.h
    build_sart_tiles(m, N, P, SY, SZ, SMAXR, PIXB, t);
    REQUIRE(t.st_ok, "SART tile tables rejected");
    const int stz = t.st_tiles_z, snt = t.st_tiles;
    double worst_st = 0, worst_stbp = 0;
    for (int i = 0; i < P; ++i) {
        std::vector<double> ps(t.st_max_ids, 0.0);
        std::vector<int> wr(t.st_max_ids, 0);
        for (int k = 0; k < snt; ++k) {
            int y0 = (k / stz) * SY, z0 = (k % stz) * SZ;
            bool ended = false;
            for (int q = 0; q < MS; ++q) {
                uint32_t b0 = t.st_seg[(((size_t)i * snt + k) * MS + q) * 2], nb = t.st_seg[(((size_t)i * snt + k) * MS + q) * 2 + 1];
                if (nb == 0) { ended = true; continue; }
                REQUIRE(!ended, "segment slots of a tile are not contiguous");
                double acc = 0;
                for (uint32_t e = 0; e < nb * NB; ++e) {
                    uint32_t off = t.st_off[(size_t)b0 * NB + e]; float w = t.st_w[(size_t)b0 * NB + e];
                    uint32_t lp = off / PIXB;
                    REQUIRE(off % PIXB == 0 && lp <= (uint32_t)(SY * SZ), "bad SART entry offset");
                    if (lp == (uint32_t)(SY * SZ)) { REQUIRE(w == 0.f, "padding with weight"); continue; }
                    int y = y0 + lp / SZ, z = z0 + lp % SZ;
                    REQUIRE(y < N && z < N, "SART entry outside the image");
                    acc += (double)w * x[(int64_t)y * N + z];
                }
                uint32_t pid = t.st_segid[((size_t)i * snt + k) * MS + q];
                REQUIRE(pid < t.st_max_ids, "partial id out of range");
                ps[pid] = acc; wr[pid]++;
            }
            // cells against the cell table
            uint32_t w = t.st_win[(size_t)i * snt + k], lo = w & 0xFFFFu, nr = w >> 16;
            REQUIRE(nr <= (uint32_t)SMAXR, "SART window too wide");
            for (int lp = 0; lp < SY * SZ; ++lp) {
                int y = y0 + lp / SZ, z = z0 + lp % SZ;
                const Tables::TileCell &c = t.st_cell[((size_t)i * snt + k) * (SY * SZ) + lp];
                double a = 0, ref = 0;
                for (int h = 0; h < 2; ++h) {
                    uint32_t off = h ? c.off1 : c.off0; float wt = h ? c.w1 : c.w0;
                    if (off == (uint32_t)SMAXR * PIXB) { REQUIRE(wt == 0.f, "zero row with weight"); continue; }
                    REQUIRE(off % PIXB == 0 && off / PIXB < nr, "bad SART cell offset");
                    a += (double)wt * rr[(int64_t)i * N + lo + off / PIXB];
                }
                if (y < N && z < N) {
                    const Cell &cc = t.cell[(size_t)i * npix + (int64_t)y * N + z];
                    if (cc.w0 != 0.f) ref += (double)cc.w0 * rr[(int64_t)i * N + cc.r0];
                    if (cc.w1 != 0.f) ref += (double)cc.w1 * rr[(int64_t)i * N + cc.r1];
                }
                worst_stbp = std::max(worst_stbp, std::fabs(a - ref));
            }
        }
        for (int j = 0; j < N; ++j) {
            int64_t r = (int64_t)i * N + j;
            double sum = 0;
            for (uint32_t q = t.st_row_first[r]; q < t.st_row_first[r] + t.st_row_nseg[r]; ++q) { REQUIRE(q < t.st_max_ids && wr[q] == 1, "row uses an unwritten partial"); sum += ps[q]; wr[q] = 2; }
            worst_st = std::max(worst_st, std::fabs(sum - g[r]));
        }
        for (uint32_t q = 0; q < t.st_max_ids; ++q) REQUIRE(wr[q] != 1, "partial %u of angle %d belongs to no row", q, i);
    }
    REQUIRE(worst_st < 1e-9 && worst_stbp == 0.0, "SART tile mismatch fp %g bp %g", worst_st, worst_stbp);
    // ---- build_tables: the cell table and the three column sums against the plain row loop (ctvlib.cpp:194-202 order), BIT FOR BIT:
    // the builder spreads the work over threads (cells per angle, sums per pixel range) and must not change a rounding
    {
        std::vector<Cell> cell((size_t)P * npix, Cell{0u, 0.f, 0u, 0.f});
        std::vector<float> cs(npix, 0.f), a1(npix, 0.f), am(npix, 0.f);
        for (int64_t r = 0; r < nrows; ++r) {
            const int i = (int)(r / N);
            const uint32_t j = (uint32_t)(r % N);
            for (int64_t k = m.ptr[r]; k < m.ptr[r + 1]; ++k) {
                const uint32_t p = m.col[k];
                const float w = m.val[k];
                cs[p] += w; a1[p] += w * t.rowsum[r]; am[p] += (w * t.rowinner[r]) * t.rowsum[r];
                if (w == 0.f) continue;
                Cell &c = cell[(size_t)i * npix + p];
                if (c.w0 == 0.f) { c.r0 = j; c.w0 = w; } else { c.r1 = j; c.w1 = w; }
            }
        }
        float L = 0.f, Lm = 0.f;
        for (int64_t p = 0; p < npix; ++p) { L = std::max(L, a1[p]); Lm = std::max(Lm, am[p]); }
        REQUIRE(std::memcmp(cs.data(), t.colsum_all.data(), npix * sizeof(float)) == 0, "column sums differ from the row loop");
        REQUIRE(L == t.lipschitz && Lm == t.lipschitz_cimmino, "Lipschitz constants differ from the row loop: %.9g / %.9g vs %.9g / %.9g", t.lipschitz, t.lipschitz_cimmino, L, Lm);
        REQUIRE(std::memcmp(cell.data(), t.cell.data(), cell.size() * sizeof(Cell)) == 0, "cell table differs from the row loop");
    }
    std::printf("OK N=%d P=%d nnz=%ld nseg=%u padded=%.3f stream_imbalance=%.3f max_window=%u\n", N, P, (long)nnz, t.tile_nseg,
                (double)t.tile_off.size() / std::max<int64_t>(1, nnz), imb, maxwin);
    return 0;
}
