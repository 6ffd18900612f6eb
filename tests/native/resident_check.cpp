// Host-only check of the tables of the volume-resident SART sweep (tomo_tv_amd/csrc/resident.cpp: build_sart_resident).
// Replays what k_sart_resident does with them, in double precision on one slice:
//   forward projection  -- per (tile, wave) the block sums acc[s0] += w0 x[pixel], acc[s0 + 1] += w1 x[pixel] from the cells, per tile
//                          the sums of its window rays from the block sums the ts lists name (wave << 4 | slot, ascending wave, padded
//                          with a row that must stay zero), per ray the tile sums named by its reducer list -- must equal the CSR
//                          product A x, every nonzero weight used exactly once;
//   back projection     -- the cell of every pixel must name the rays / weights of the cell table through the wave's window
//                          (jbase + dw + s0, and the NEXT ray for the second weight) and carry 1 / (w0 + w1) as the single-precision
//                          quotient;
//   windows             -- a tile's window holds <= MAXWIN rays, a wave's <= USABLE, every list <= RL entries, ascending tile.
// Usage: resident_check N P max_abs_angle_deg [max_tiles=256] [quiet]
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include "resident.h"
using namespace tomo;

#define REQUIRE(c, ...) do { if (!(c)) { std::printf("FAIL %s:%d: ", __FILE__, __LINE__); std::printf(__VA_ARGS__); std::printf("\n"); return 1; } } while (0)
static inline float bitsf(uint32_t b) { float f; std::memcpy(&f, &b, 4); return f; }

int main(int argc, char **argv)
{
    const int N = argc > 1 ? std::atoi(argv[1]) : 64, P = argc > 2 ? std::atoi(argv[2]) : 9;
    const double amax = argc > 3 ? std::atof(argv[3]) : 70.0;
    const int max_tiles = argc > 4 ? std::atoi(argv[4]) : 256;
    const bool quiet = argc > 5;
    std::vector<double> ang(P);
    for (int i = 0; i < P; ++i) ang[i] = (P > 1 ? -amax + 2 * amax * i / (P - 1) : 0.0) * M_PI / 180;
    Coo m; build_parallel_ray(N, P, ang.data(), m); sort_rows(m);
    Tables t; std::string why;
    REQUIRE(build_tables(m, N, P, t, why), "build_tables: %s", why.c_str());
    Resident R; build_sart_resident(N, P, t, max_tiles, R);
    if (!R.ok) { std::printf("NOT BUILT: %s\n", R.why.c_str()); return 2; }
    constexpr int T = Resident::T, W = Resident::WAVES, Q = Resident::PPW;
    const int ntiles = R.ntiles;
    const int64_t npix = (int64_t)N * N;
    REQUIRE(R.tiles == (N + T - 1) / T && ntiles == R.tiles * R.tiles && ntiles <= max_tiles, "tile counts");
    REQUIRE(R.rpt * ntiles >= N, "every ray needs a reducer");
    REQUIRE(R.hdr.size() == (size_t)P * ntiles && R.cell.size() == (size_t)P * ntiles * W * Q * 4 && R.ts.size() == (size_t)P * ntiles * Resident::MAXWIN * Resident::TSN && R.rl.size() == (size_t)P * N * Resident::RL, "table sizes");
    std::mt19937 rng(7); std::uniform_real_distribution<double> U(0.1, 1.0);
    std::vector<double> x(npix);
    for (auto &v : x) v = U(rng);
    uint64_t nz_cells = 0, nz_matrix = 0;
    int max_win = 0, max_wwin = 0, max_list = 0;
    for (int i = 0; i < P; ++i) {
        const Cell *ci = t.cell.data() + (size_t)i * npix;
        std::vector<double> tsum((size_t)ntiles * Resident::MAXWIN, 0.0);
        for (int k = 0; k < ntiles; ++k) {
            const Resident::Hdr &h = R.hdr[(size_t)i * ntiles + k];
            REQUIRE(h.nrays <= Resident::MAXWIN && h.jbase + h.nrays <= N, "window of tile %d angle %d", k, i);
            max_win = std::max<int>(max_win, h.nrays);
            const int y0 = (k / R.tiles) * T, z0 = (k % R.tiles) * T;
            std::vector<double> pbuf((size_t)W * 16, 0.0);
            for (int w = 0; w < W; ++w) {
                const uint32_t *cc = R.cell.data() + (((size_t)i * ntiles + k) * W + w) * Q * 4;
                int top = -1;
                for (int q = 0; q < Q; ++q) {
                    int ly, lz; Resident::pixel(w, q, ly, lz);
                    const int y = y0 + ly, z = z0 + lz;
                    const uint32_t s0 = cc[q * 4];
                    const float w0 = bitsf(cc[q * 4 + 1]), w1 = bitsf(cc[q * 4 + 2]);
                    REQUIRE(s0 <= (uint32_t)Resident::SINK, "slot range");
                    REQUIRE((w0 != 0.f) == (s0 < (uint32_t)Resident::USABLE) && (w1 == 0.f || (w0 != 0.f && s0 + 1 < (uint32_t)Resident::USABLE)),
                            "a pixel without rays must aim at the sink, a real first ray at slots 0..13, a second ray needs a first (tile %d wave %d pixel %d angle %d)", k, w, q, i);
                    const float cs = w0 + w1, inv = 1.0f / (cs > 0.f ? cs : 1.0f);
                    REQUIRE(bitsf(cc[q * 4 + 3]) == inv, "divisor");
                    if (y >= N || z >= N) { REQUIRE(w0 == 0.f && w1 == 0.f, "weight outside the image"); continue; }
                    const Cell &c = ci[(int64_t)y * N + z];
                    const int base = h.jbase + h.dw[w];
                    if (c.w0 != 0.f) { REQUIRE(w0 == c.w0 && base + (int)s0 == (int)c.r0, "first ray of pixel (%d,%d) angle %d", y, z, i); ++nz_cells; }
                    else REQUIRE(w0 == 0.f, "phantom first weight");
                    if (c.w1 != 0.f) { REQUIRE(w1 == c.w1 && base + (int)s0 + 1 == (int)c.r1, "second ray of pixel (%d,%d) angle %d", y, z, i); ++nz_cells; }
                    else REQUIRE(w1 == 0.f, "phantom second weight");
                    pbuf[(size_t)w * 16 + s0] += (double)w0 * x[(int64_t)y * N + z];
                    pbuf[(size_t)w * 16 + s0 + 1] += (double)w1 * x[(int64_t)y * N + z];
                    if (w0 != 0.f) top = std::max(top, (int)s0);
                    if (w1 != 0.f) top = std::max(top, (int)s0 + 1);
                }
                REQUIRE(top < 0 || h.dw[w] + top < h.nrays, "a wave's window reaches beyond the tile's");
                REQUIRE(pbuf[(size_t)w * 16 + 14] == 0.0 && pbuf[(size_t)w * 16 + 15] == 0.0, "the sink rows must stay zero");
                max_wwin = std::max(max_wwin, top + 1);
            }
            for (int r = 0; r < h.nrays; ++r) {
                const uint8_t *e = R.ts.data() + (((size_t)i * ntiles + k) * Resident::MAXWIN + r) * Resident::TSN;
                double acc = 0.0;
                int prev = -1;
                bool padded = false;
                for (int c = 0; c < Resident::TSN; ++c) {
                    if (e[c] == Resident::TS_PAD) { padded = true; continue; }
                    REQUIRE(!padded, "entry behind the padding");
                    const int w = e[c] >> 4, sl = e[c] & 15;
                    REQUIRE(w > prev && sl < Resident::USABLE && h.dw[w] + sl == r, "tile-sum entry %d of ray %d tile %d angle %d", c, r, k, i);
                    acc += pbuf[(size_t)w * 16 + sl];
                    prev = w;
                }
                // every block sum that carries weight on this ray is in the list (the others are exact zeros)
                double all = 0.0;
                for (int w = 0; w < W; ++w) { const int sl = r - h.dw[w]; if ((unsigned)sl < (unsigned)Resident::USABLE) all += pbuf[(size_t)w * 16 + sl]; }
                REQUIRE(acc == all, "tile sum of ray %d tile %d angle %d misses a block", r, k, i);
                tsum[(size_t)k * Resident::MAXWIN + r] = acc;
            }
        }
        for (int j = 0; j < N; ++j) {
            const uint16_t *list = R.rl.data() + ((size_t)i * N + j) * Resident::RL;
            double got = 0.0;
            int prev_tile = -1, cnt = 0;
            for (int e = 0; e < Resident::RL; ++e) {
                if (list[e] == 0xFFFF) { for (int f = e; f < Resident::RL; ++f) REQUIRE(list[f] == 0xFFFF, "hole in a reducer list"); break; }
                const int tile = list[e] / Resident::MAXWIN, slot = list[e] % Resident::MAXWIN;
                REQUIRE(tile > prev_tile && tile < ntiles, "reducer list not ascending");
                const Resident::Hdr &h = R.hdr[(size_t)i * ntiles + tile];
                REQUIRE(h.jbase + slot == j && slot < h.nrays, "reducer entry names another ray");
                got += tsum[list[e]];
                prev_tile = tile; ++cnt;
            }
            max_list = std::max(max_list, cnt);
            // every tile whose window holds the ray is in the list
            int holders = 0;
            for (int k = 0; k < ntiles; ++k) { const Resident::Hdr &h = R.hdr[(size_t)i * ntiles + k]; if (j >= h.jbase && j < h.jbase + h.nrays) ++holders; }
            REQUIRE(holders == cnt, "ray %d angle %d: %d windows hold it, %d listed", j, i, holders, cnt);
            double want = 0.0;
            const int64_t row = (int64_t)i * N + j;
            for (int64_t e = m.ptr[row]; e < m.ptr[row + 1]; ++e) { want += (double)m.val[e] * x[m.col[e]]; if (m.val[e] != 0.f) ++nz_matrix; }
            REQUIRE(std::fabs(got - want) <= 1e-10 * (1.0 + std::fabs(want)), "ray %d angle %d: %.12g vs %.12g", j, i, got, want);
        }
    }
    REQUIRE(nz_cells == nz_matrix, "%llu weights in the cells, %llu in the matrix", (unsigned long long)nz_cells, (unsigned long long)nz_matrix);
    if (!quiet) std::printf("OK N %d P %d: %d tiles, rpt %d, widest tile window %d, widest block window %d, longest reducer list %d, %llu weights\n", N, P, ntiles, R.rpt,
                            max_win, max_wwin, max_list, (unsigned long long)nz_matrix);
    return 0;
}
