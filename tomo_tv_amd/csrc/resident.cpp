// resident.cpp -- tables of the volume-resident SART sweep (see resident.h).  Host code only.
#include "resident.h"

#include <sched.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <thread>

namespace tomo {

static unsigned resident_threads()
{
    if (const char *s = std::getenv("TOMO_BUILD_THREADS")) { int v = std::atoi(s); if (v > 0) return (unsigned)v; }
    unsigned hw = std::thread::hardware_concurrency();
#ifdef __linux__
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) { int c = CPU_COUNT(&set); if (c > 0) hw = std::min<unsigned>(hw ? hw : (unsigned)c, (unsigned)c); }
#endif
    return hw ? std::min(hw, 32u) : 1u;
}

static inline uint32_t fbits(float f) { uint32_t b; std::memcpy(&b, &f, 4); return b; }

void build_sart_resident(int N, int P, const Tables &t, int max_tiles, Resident &r)
{
    constexpr int T = Resident::T, W = Resident::WAVES, Q = Resident::PPW;
    r.ok = false; r.why.clear();
    r.tiles = (N + T - 1) / T;
    r.ntiles = r.tiles * r.tiles;
    if (r.ntiles > max_tiles) { r.why = "more tiles than resident workgroups"; return; }
    if (t.cell.size() != (size_t)P * N * N) { r.why = "no cell table"; return; }
    if (!t.art_chain_ok) { r.why = "a pixel's two rays of an angle are not neighbours"; return; }
    r.rpt = (N + r.ntiles - 1) / r.ntiles;
    const int64_t npix = (int64_t)N * N;
    const int ntiles = r.ntiles;
    r.hdr.assign((size_t)P * ntiles, Resident::Hdr{});
    r.cell.assign((size_t)P * ntiles * W * Q * 4, 0u);
    r.ts.assign((size_t)P * ntiles * Resident::MAXWIN * Resident::TSN, (uint8_t)Resident::TS_PAD);
    r.rl.assign((size_t)P * N * Resident::RL, (uint16_t)0xFFFFu);
    const int nth = (int)std::max(1u, std::min<unsigned>(resident_threads(), (unsigned)P));
    std::vector<std::string> bad(nth);
    auto work = [&](int th) {
        for (int i = th; i < P; i += nth) {
            const Cell *ci = t.cell.data() + (size_t)i * npix;
            for (int k = 0; k < ntiles; ++k) {
                const int y0 = (k / r.tiles) * T, z0 = (k % r.tiles) * T;
                Resident::Hdr &h = r.hdr[(size_t)i * ntiles + k];
                uint32_t wlo[W], whi[W], tlo = 0xFFFFFFFFu, thi = 0;
                for (int w = 0; w < W; ++w) {
                    uint32_t lo = 0xFFFFFFFFu, hi = 0;
                    for (int q = 0; q < Q; ++q) {
                        int ly, lz; Resident::pixel(w, q, ly, lz);
                        const int y = y0 + ly, z = z0 + lz;
                        if (y >= N || z >= N) continue;
                        const Cell &c = ci[(int64_t)y * N + z];
                        if (c.w0 != 0.f) { lo = std::min(lo, c.r0); hi = std::max(hi, c.r0); }
                        if (c.w1 != 0.f) { lo = std::min(lo, c.r1); hi = std::max(hi, c.r1); }
                    }
                    wlo[w] = lo; whi[w] = hi;
                    if (lo != 0xFFFFFFFFu) {
                        if (hi - lo + 1 > (uint32_t)Resident::USABLE && bad[th].empty())
                            bad[th] = "a block is crossed by more than " + std::to_string(Resident::USABLE) + " rays of angle " + std::to_string(i);
                        tlo = std::min(tlo, lo); thi = std::max(thi, hi);
                    }
                }
                const uint32_t nr = tlo == 0xFFFFFFFFu ? 0u : thi - tlo + 1;
                if (nr == 0) tlo = 0;
                if (nr > (uint32_t)Resident::MAXWIN) { if (bad[th].empty()) bad[th] = "a tile's ray window exceeds " + std::to_string(Resident::MAXWIN) + " rays at angle " + std::to_string(i); continue; }
                if (!bad[th].empty()) continue;
                h.jbase = (uint16_t)tlo; h.nrays = (uint16_t)nr;
                uint8_t *ts = r.ts.data() + ((size_t)i * ntiles + k) * Resident::MAXWIN * Resident::TSN;
                int tsn[Resident::MAXWIN] = {0};
                for (int w = 0; w < W; ++w) {
                    const uint32_t base = wlo[w] == 0xFFFFFFFFu ? tlo : wlo[w];
                    h.dw[w] = (uint8_t)(base - tlo);
                    uint32_t *cp = r.cell.data() + (((size_t)i * ntiles + k) * W + w) * Q * 4;
                    bool used[Resident::USABLE] = {false};
                    for (int q = 0; q < Q; ++q) {
                        int ly, lz; Resident::pixel(w, q, ly, lz);
                        const int y = y0 + ly, z = z0 + lz;
                        uint32_t s0 = Resident::SINK;
                        float w0 = 0.f, w1 = 0.f;
                        if (y < N && z < N) {
                            const Cell &c = ci[(int64_t)y * N + z];
                            if (c.w0 != 0.f) { s0 = c.r0 - base; w0 = c.w0; if (s0 < (uint32_t)Resident::USABLE) used[s0] = true; }
                            if (c.w1 != 0.f) {
                                // (the neighbour property: the second ray is the next one; a first ray with weight 0 cannot have a second)
                                if ((c.w0 == 0.f || c.r1 != c.r0 + 1 || s0 + 1 >= (uint32_t)Resident::USABLE) && bad[th].empty()) bad[th] = "second ray of a pixel is not the next ray of its block window";
                                w1 = c.w1;
                                if (s0 + 1 < (uint32_t)Resident::USABLE) used[s0 + 1] = true;
                            }
                        }
                        // the divisor of k_bp_angle / k_sart_tile: 1 / (w0 + w1), 1 where no ray crosses the pixel (IEEE single division)
                        const float cs = w0 + w1;
                        const float inv = 1.0f / (cs > 0.f ? cs : 1.0f);
                        cp[q * 4 + 0] = s0; cp[q * 4 + 1] = fbits(w0); cp[q * 4 + 2] = fbits(w1); cp[q * 4 + 3] = fbits(inv);
                    }
                    // this wave's block sums, by window ray of the tile (waves ascend, so every list is in wave order)
                    for (int sl = 0; sl < Resident::USABLE; ++sl) {
                        if (!used[sl]) continue;
                        const int row = (int)(base - tlo) + sl;
                        if (tsn[row] >= Resident::TSN) { if (bad[th].empty()) bad[th] = "more than " + std::to_string(Resident::TSN) + " blocks of a tile on one ray"; continue; }
                        ts[row * Resident::TSN + tsn[row]++] = (uint8_t)(w << 4 | sl);
                    }
                }
            }
            if (!bad[th].empty()) continue;
            // reducer lists: the tiles whose window holds ray j, ascending tile
            std::vector<int> cnt(N, 0);
            for (int k = 0; k < ntiles; ++k) {
                const Resident::Hdr &h = r.hdr[(size_t)i * ntiles + k];
                for (uint32_t s = 0; s < h.nrays; ++s) {
                    const int j = (int)h.jbase + (int)s;
                    if (j >= N) { if (bad[th].empty()) bad[th] = "ray window beyond the detector"; break; }
                    if (cnt[j] >= Resident::RL) { if (bad[th].empty()) bad[th] = "a ray crosses more than " + std::to_string(Resident::RL) + " tile windows at angle " + std::to_string(i); break; }
                    r.rl[((size_t)i * N + j) * Resident::RL + cnt[j]++] = (uint16_t)(k * Resident::MAXWIN + s);
                }
            }
        }
    };
    std::vector<std::thread> thr;
    for (int th = 1; th < nth; ++th) thr.emplace_back(work, th);
    work(0);
    for (auto &x : thr) x.join();
    for (auto &b : bad) if (!b.empty()) { r.why = b; r.hdr.clear(); r.cell.clear(); r.ts.clear(); r.rl.clear(); r.hdr.shrink_to_fit(); r.cell.shrink_to_fit(); r.ts.shrink_to_fit(); r.rl.shrink_to_fit(); return; }
    r.ok = true;
}

}  // namespace tomo
