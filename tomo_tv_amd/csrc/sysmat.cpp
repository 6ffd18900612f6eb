// sysmat.cpp -- host-side system-matrix builder and table construction (no device code).
//
// Produces the line-intersection matrix that the reference's CPU path builds in Python
// (tomofusion/cpu/utils/pytvlib.py:8-121, parallelRay): ray j of angle i is the centre line through
// offset_j*(cos t, sin t) with direction (-sin t, cos t); A[i*N+j, pixel] = chord length in that pixel.
// The reference sorts all 2(N+1) grid crossings per ray; here the x-grid and y-grid crossing families are
// each monotone in the ray parameter, so they are merged in O(N), many rays at a time on host threads.
// Epsilon handling (snap, duplicate-point and boundary-ray rules) follows pytvlib.py:38-44,63-92,124-130
// so that the float32 weights equal parallelRay's.
#include "sysmat.h"

#include <sched.h>
#include <cstdlib>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <chrono>
#include <cstdio>

namespace tomo {

// Host threads of the table builders: the CPUs this process may actually use (affinity mask; a cgroup quota is not visible
// here), or TOMO_BUILD_THREADS when set -- one process per GPU builds its own tables, so a launcher that starts N ranks on
// one node divides the CPUs between them (bench.py does).
static unsigned builder_threads()
{
    if (const char *s = std::getenv("TOMO_BUILD_THREADS")) { int v = std::atoi(s); if (v > 0) return (unsigned)v; }
    unsigned hw = std::thread::hardware_concurrency();
#ifdef __linux__
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) { int c = CPU_COUNT(&set); if (c > 0) hw = std::min<unsigned>(hw ? hw : (unsigned)c, (unsigned)c); }
#endif
    return hw ? hw : 1u;
}


static inline double snap10(double v) { return std::fabs(v) < 1e-10 ? 0.0 : v; }

// entries of one ray in path order; returns count
static int trace_ray(int N, const double *grid, double x0, double y0, double a, double b, double half,
                     double *qx, double *qy, uint32_t *cols, float *vals)
{
    const int M = N + 1;
    int n = 0;
    // the two crossing families, each walked in increasing t
    int ix = (a > 0) ? 0 : M - 1, dx = (a > 0) ? 1 : -1, nxl = (a != 0.0) ? M : 0;
    int iy = (b > 0) ? 0 : M - 1, dy = (b > 0) ? 1 : -1, nyl = (b != 0.0) ? M : 0;
    double tx = 0, ty = 0;
    if (nxl) tx = (grid[ix] - x0) / a;
    if (nyl) ty = (grid[iy] - y0) / b;
    while (nxl > 0 || nyl > 0) {
        bool take_x;
        if (nxl == 0) take_x = false;
        else if (nyl == 0) take_x = true;
        else take_x = (tx <= ty);
        double px, py;
        if (take_x) {
            px = grid[ix]; py = b * tx + y0;
            ix += dx; --nxl;
            if (nxl) tx = (grid[ix] - x0) / a;
        } else {
            px = a * ty + x0; py = grid[iy];
            iy += dy; --nyl;
            if (nyl) ty = (grid[iy] - y0) / b;
        }
        if (px >= -half && px <= half && py >= -half && py <= half) {
            // duplicate rule: a point is dropped when its successor lies within 1e-8 in both coordinates
            if (n > 0 && std::fabs(px - qx[n - 1]) <= 1e-8 && std::fabs(py - qy[n - 1]) <= 1e-8) {
                qx[n - 1] = px; qy[n - 1] = py;
            } else {
                qx[n] = px; qy[n] = py; ++n;
            }
        }
    }
    int numvals = n - 1;
    if (numvals <= 0) return 0;
    if ((b == 0.0 && std::fabs(y0 - half) < 1e-15) || (a == 0.0 && std::fabs(x0 - half) < 1e-15)) return 0;
    for (int k = 0; k < numvals; ++k) {
        double ex = qx[k + 1] - qx[k], ey = qy[k + 1] - qy[k];
        double len = std::sqrt(ex * ex + ey * ey);
        double mx = snap10(0.5 * (qx[k] + qx[k + 1]));
        double my = snap10(0.5 * (qy[k] + qy[k + 1]));
        double pix = std::floor(half - my) * N + std::floor(mx + half);
        cols[k] = (uint32_t)(int64_t)(float)pix;
        vals[k] = (float)len;
    }
    return numvals;
}

void build_parallel_ray(int N, int P, const double *angles_rad, Coo &out)
{
    const int M = N + 1;
    std::vector<double> grid(M), offs(N);
    for (int m = 0; m < M; ++m) grid[m] = (m == M - 1) ? N * 0.5 : -N * 0.5 + m * ((N * 0.5 + N * 0.5) / (double)N);
    {
        double start = -((double)N - 1.0) / 2.0, stop = ((double)N - 1.0) / 2.0;
        double step = (N > 1) ? (stop - start) / (double)(N - 1) : 0.0;
        for (int j = 0; j < N; ++j) offs[j] = (j == N - 1 && N > 1) ? stop : start + j * step;
    }
    const double half = N / 2.0;
    const int64_t nrays = (int64_t)N * P;
    unsigned hw = builder_threads();
    int nth = (int)std::min<int64_t>(std::max(1u, std::min(hw, 32u)), std::max<int64_t>(1, nrays / 64));
    std::vector<std::vector<uint32_t>> tcols(nth);
    std::vector<std::vector<float>> tvals(nth);
    std::vector<uint32_t> count(nrays, 0);
    auto work = [&](int t) {
        int64_t r0 = nrays * t / nth, r1 = nrays * (t + 1) / nth;
        std::vector<double> qx(2 * M), qy(2 * M);
        std::vector<uint32_t> c(2 * M);
        std::vector<float> v(2 * M);
        tcols[t].reserve((size_t)((r1 - r0) * 1.3 * N));
        tvals[t].reserve((size_t)((r1 - r0) * 1.3 * N));
        for (int64_t r = r0; r < r1; ++r) {
            int i = (int)(r / N), j = (int)(r % N);
            double ang = angles_rad[i];
            double ca = std::cos(ang), sa = std::sin(ang);
            double x0 = ca * offs[j], y0 = sa * offs[j];
            if (std::fabs(x0) < 1e-8) x0 = 0.0;
            if (std::fabs(y0) < 1e-8) y0 = 0.0;
            double a = snap10(-sa), b = snap10(ca);
            int n = trace_ray(N, grid.data(), x0, y0, a, b, half, qx.data(), qy.data(), c.data(), v.data());
            count[r] = (uint32_t)n;
            tcols[t].insert(tcols[t].end(), c.begin(), c.begin() + n);
            tvals[t].insert(tvals[t].end(), v.begin(), v.begin() + n);
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < nth; ++t) th.emplace_back(work, t);
    work(0);
    for (auto &x : th) x.join();
    out.nrow = nrays;
    out.ncol = (int64_t)N * N;
    out.ptr.assign(nrays + 1, 0);
    for (int64_t r = 0; r < nrays; ++r) out.ptr[r + 1] = out.ptr[r] + count[r];
    out.col.resize(out.ptr[nrays]);
    out.val.resize(out.ptr[nrays]);
    for (int t = 0; t < nth; ++t) {
        int64_t r0 = nrays * t / nth;
        std::memcpy(out.col.data() + out.ptr[r0], tcols[t].data(), tcols[t].size() * sizeof(uint32_t));
        std::memcpy(out.val.data() + out.ptr[r0], tvals[t].data(), tvals[t].size() * sizeof(float));
    }
}

// ctvlib::loadA (cpu/utils/ctvlib.cpp:309-315): coeffRef(row, col) = val.  Entries arrive as float32 triplets.
bool coo_from_triplets(int64_t nrow, int64_t ncol, int64_t nnz, const float *rows, const float *cols,
                       const float *vals, Coo &out, std::string &err)
{
    out.nrow = nrow; out.ncol = ncol;
    out.ptr.assign(nrow + 1, 0);
    for (int64_t k = 0; k < nnz; ++k) {
        int64_t r = (int64_t)rows[k], c = (int64_t)cols[k];
        if (r < 0 || r >= nrow || c < 0 || c >= ncol || rows[k] != (float)r || cols[k] != (float)c) {
            err = "load_A: entry " + std::to_string(k) + " has row/col outside the matrix";
            return false;
        }
        out.ptr[r + 1]++;
    }
    for (int64_t r = 0; r < nrow; ++r) out.ptr[r + 1] += out.ptr[r];
    out.col.resize(nnz); out.val.resize(nnz);
    std::vector<int64_t> fill(out.ptr.begin(), out.ptr.end() - 1);
    for (int64_t k = 0; k < nnz; ++k) {
        int64_t pos = fill[(int64_t)rows[k]]++;
        out.col[pos] = (uint32_t)(int64_t)cols[k];
        out.val[pos] = vals[k];
    }
    return true;
}

// Sort every row by column (the order Eigen's RowMajor storage iterates: ctvlib.hpp:22); a repeated
// (row, col) keeps the last assignment.
void sort_rows(Coo &m)
{
    unsigned hw = builder_threads();
    int nth = (int)std::min<int64_t>(std::max(1u, std::min(hw, 32u)), std::max<int64_t>(1, m.nrow / 256));
    std::vector<uint32_t> keep(m.nrow, 0);
    auto work = [&](int t) {
        int64_t r0 = m.nrow * t / nth, r1 = m.nrow * (t + 1) / nth;
        std::vector<std::pair<uint32_t, uint32_t>> key;
        std::vector<float> tmp;
        for (int64_t r = r0; r < r1; ++r) {
            int64_t b = m.ptr[r], e = m.ptr[r + 1];
            int n = (int)(e - b);
            key.resize(n); tmp.resize(n);
            for (int k = 0; k < n; ++k) { key[k] = {m.col[b + k], (uint32_t)k}; tmp[k] = m.val[b + k]; }
            std::sort(key.begin(), key.end());
            int o = 0;
            for (int k = 0; k < n; ++k) {
                if (k + 1 < n && key[k + 1].first == key[k].first) continue;
                m.col[b + o] = key[k].first; m.val[b + o] = tmp[key[k].second]; ++o;
            }
            keep[r] = (uint32_t)o;
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < nth; ++t) th.emplace_back(work, t);
    work(0);
    for (auto &x : th) x.join();
    // compact if duplicates were dropped
    bool dense = true;
    for (int64_t r = 0; r < m.nrow; ++r) if ((int64_t)keep[r] != m.ptr[r + 1] - m.ptr[r]) { dense = false; break; }
    if (dense) return;
    int64_t o = 0;
    std::vector<int64_t> nptr(m.nrow + 1, 0);
    for (int64_t r = 0; r < m.nrow; ++r) {
        int64_t b = m.ptr[r];
        for (uint32_t k = 0; k < keep[r]; ++k) { m.col[o] = m.col[b + k]; m.val[o] = m.val[b + k]; ++o; }
        nptr[r + 1] = o;
    }
    m.ptr.swap(nptr);
    m.col.resize(o); m.val.resize(o);
}

// Derived tables (all fp32, accumulated in the order the oracle / Eigen would):
//   rowsum[r] = A_r . 1, rowinner[r] = A_r . A_r            (ascending column)
//   cell[i][p] = up to two (ray-of-angle-i, weight) pairs through pixel p, ascending ray
//   colsum_all[p] = sum_r A[r,p]                            (ascending row)
//   lipschitz = max_p (A^T (A 1))_p, lipschitz_cimmino = max_p (A^T M (A 1))_p   (ctvlib.cpp:194-202)
bool build_tables(const Coo &m, int N, int P, Tables &t, std::string &err)
{
    const int64_t npix = (int64_t)N * N;
    t.rowsum.assign(m.nrow, 0.f);
    t.rowinner.assign(m.nrow, 0.f);
    for (int64_t r = 0; r < m.nrow; ++r) {
        float s = 0.f, q = 0.f;
        for (int64_t k = m.ptr[r]; k < m.ptr[r + 1]; ++k) { s += m.val[k]; q += m.val[k] * m.val[k]; }
        t.rowsum[r] = s; t.rowinner[r] = q;
    }
    // Cells: one angle at a time, angles over threads (an angle's cells are written by its own rays only).  The three column sums
    // (A^T 1, A^T A 1, A^T M A 1) are accumulated afterwards per pixel in ascending (angle, ray) order from the cells -- the order
    // the row loop of the oracle adds them in (a cell lists its two rays in ascending ray order; zero weights add nothing), so the
    // sums keep their bits while the work spreads over pixel ranges.  Round 4: 2.6 s -> 0.4 s at 1024^2 x 120 on 16 cores.
    t.cell.resize((size_t)P * npix);
    t.colsum_all.assign(npix, 0.f);
    std::vector<float> ata1(npix, 0.f), atma1(npix, 0.f);
    {
        const unsigned hw = std::min(builder_threads(), 32u);
        const int nth = (int)std::max<int64_t>(1, std::min<int64_t>(hw, P));
        std::vector<std::string> errs(nth);
        auto cells_of = [&](int th) {
            for (int i = th; i < P; i += nth) {
                Cell *ci = t.cell.data() + (size_t)i * npix;
                std::fill(ci, ci + npix, Cell{0u, 0.f, 0u, 0.f});
                for (int j = 0; j < N; ++j) {
                    const int64_t r = (int64_t)i * N + j;
                    for (int64_t k = m.ptr[r]; k < m.ptr[r + 1]; ++k) {
                        const float w = m.val[k];
                        if (w == 0.f) continue;      // a zero weight carries nothing into a voxel update
                        Cell &c = ci[m.col[k]];
                        if (c.w0 == 0.f) { c.r0 = (uint32_t)j; c.w0 = w; }
                        else if (c.w1 == 0.f) { c.r1 = (uint32_t)j; c.w1 = w; }
                        else if (errs[th].empty())
                            errs[th] = "system matrix has more than two rays of angle " + std::to_string(i) + " through pixel " +
                                       std::to_string(m.col[k]) + " (unsupported geometry)";
                    }
                }
            }
        };
        {
            std::vector<std::thread> thr;
            for (int th = 1; th < nth; ++th) thr.emplace_back(cells_of, th);
            cells_of(0);
            for (auto &x : thr) x.join();
        }
        for (auto &e2 : errs) if (!e2.empty()) { err = e2; return false; }
        const int npt = (int)std::max<int64_t>(1, std::min<int64_t>(hw, npix / 4096));
        auto sums_of = [&](int th) {
            const int64_t p0 = npix * th / npt, p1 = npix * (th + 1) / npt;
            for (int i = 0; i < P; ++i) {
                const Cell *ci = t.cell.data() + (size_t)i * npix;
                const float *rs = t.rowsum.data() + (size_t)i * N, *ri = t.rowinner.data() + (size_t)i * N;
                for (int64_t p = p0; p < p1; ++p) {
                    const Cell &c = ci[p];
                    if (c.w0 != 0.f) { t.colsum_all[p] += c.w0; ata1[p] += c.w0 * rs[c.r0]; atma1[p] += (c.w0 * ri[c.r0]) * rs[c.r0]; }   // (A^T M)(A 1), M = diag(|A_i|^2): ctvlib.cpp:198-199
                    if (c.w1 != 0.f) { t.colsum_all[p] += c.w1; ata1[p] += c.w1 * rs[c.r1]; atma1[p] += (c.w1 * ri[c.r1]) * rs[c.r1]; }
                }
            }
        };
        std::vector<std::thread> thr;
        for (int th = 1; th < npt; ++th) thr.emplace_back(sums_of, th);
        sums_of(0);
        for (auto &x : thr) x.join();
    }
    float L = 0.f, Lm = 0.f;
    for (int64_t p = 0; p < npix; ++p) { L = std::max(L, ata1[p]); Lm = std::max(Lm, atma1[p]); }
    t.lipschitz = L;
    t.lipschitz_cimmino = Lm;
    // inner products of neighbouring rays (fp64 accumulation, ascending pixel) and the neighbour property
    t.rowcross.assign(m.nrow, 0.f);
    t.art_chain_ok = true;
    {
        std::vector<double> cross(m.nrow, 0.0);
        for (int i = 0; i < P; ++i) {
            const Cell *ci = t.cell.data() + (size_t)i * npix;
            double *cr = cross.data() + (size_t)i * N;
            for (int64_t p = 0; p < npix; ++p) {
                const Cell &c = ci[p];
                if (c.w0 == 0.f || c.w1 == 0.f) continue;
                uint32_t lo = std::min(c.r0, c.r1), hi = std::max(c.r0, c.r1);
                if (hi != lo + 1) { t.art_chain_ok = false; continue; }
                cr[lo] += (double)c.w0 * (double)c.w1;
            }
        }
        for (int64_t r = 0; r < m.nrow; ++r) t.rowcross[r] = (float)cross[r];
    }
    return true;
}

// Walk lists (see sysmat.h).  Owner of pixel p for angle i = the first ray listed in cell[i][p]; pixels that no
// ray of the angle crosses ("orphans", the image corners) are dealt round-robin over the angle's rays with
// weight 0, so every ray carries about the same number of extra visits.
void build_walk(const Coo &m, int N, int P, Tables &t)
{
    const int64_t npix = (int64_t)N * N;
    const int64_t nrows = (int64_t)N * P;
    std::vector<uint32_t> extra(nrows, 0);
    std::vector<std::vector<uint32_t>> orphans(P);
    for (int i = 0; i < P; ++i) {
        const Cell *c = t.cell.data() + (size_t)i * npix;
        uint32_t k = 0;
        for (int64_t p = 0; p < npix; ++p)
            if (c[p].w0 == 0.f) { orphans[i].push_back((uint32_t)p); extra[(int64_t)i * N + (k++ % N)]++; }
    }
    t.walk_ptr.assign(nrows + 1, 0);
    for (int64_t r = 0; r < nrows; ++r)
        t.walk_ptr[r + 1] = t.walk_ptr[r] + (uint32_t)(m.ptr[r + 1] - m.ptr[r]) + extra[r];
    t.walk_pix.resize(t.walk_ptr[nrows]);
    t.walk_w.resize(t.walk_ptr[nrows]);
    unsigned hw = builder_threads();
    int nth = (int)std::min<int64_t>(std::max(1u, std::min(hw, 32u)), P);
    auto work = [&](int th) {
        for (int i = th; i < P; i += nth) {
            const Cell *c = t.cell.data() + (size_t)i * npix;
            for (int j = 0; j < N; ++j) {
                int64_t r = (int64_t)i * N + j;
                uint32_t o = t.walk_ptr[r];
                for (int64_t k = m.ptr[r]; k < m.ptr[r + 1]; ++k) {
                    uint32_t p = m.col[k];
                    bool own = (m.val[k] != 0.f) && c[p].w0 != 0.f && c[p].r0 == (uint32_t)j;
                    t.walk_pix[o] = p | (own ? 0x80000000u : 0u);
                    t.walk_w[o] = m.val[k];
                    ++o;
                }
                for (size_t q = j; q < orphans[i].size(); q += N) {
                    t.walk_pix[o] = orphans[i][q] | 0x80000000u;
                    t.walk_w[o] = 0.f;
                    ++o;
                }
            }
        }
    };
    std::vector<std::thread> thr;
    for (int th = 1; th < nth; ++th) thr.emplace_back(work, th);
    work(0);
    for (auto &x : thr) x.join();
}

void build_segments(int N, int P, int seg_len, Tables &t)
{
    const int G = 16;  // RAY_GROUP of kernels.hip.h
    t.seg_len = seg_len;
    t.row_first.assign((size_t)N * P, 0);
    t.row_nseg.assign((size_t)N * P, 0);
    t.seg_exec_ptr.assign(P + 1, 0);
    t.seg_exec.clear();
    t.max_items_per_angle = 0;
    for (int i = 0; i < P; ++i) {
        uint32_t next_id = 0;
        std::vector<Tables::SegItem> lists[8];
        int ngroups = (N + G - 1) / G;
        uint32_t maxseg_all = 0;
        for (int j = 0; j < N; ++j) {
            size_t r = (size_t)i * N + j;
            uint32_t len = t.walk_ptr[r + 1] - t.walk_ptr[r];
            uint32_t ns = (len + seg_len - 1) / seg_len;
            t.row_first[r] = next_id;
            t.row_nseg[r] = ns;
            next_id += ns;
            maxseg_all = std::max(maxseg_all, ns);
        }
        // group-major: the s-th segments of neighbouring rays run side by side, one ray group after another
        // (a segment-major order over the whole angle measured the same)
        for (int g = 0; g < ngroups; ++g) {
            auto &L = lists[g & 7];
            int j0 = g * G, j1 = std::min(N, j0 + G);
            for (uint32_t sidx = 0; sidx < maxseg_all; ++sidx)
                for (int j = j0; j < j1; ++j) {
                    size_t r = (size_t)i * N + j;
                    if (sidx >= t.row_nseg[r]) continue;
                    uint32_t kb = t.walk_ptr[r] + sidx * seg_len;
                    uint32_t ke = std::min(t.walk_ptr[r + 1], kb + (uint32_t)seg_len);
                    L.push_back({t.row_first[r] + sidx, kb, ke, 0});
                }
        }
        size_t Lmax = 0;
        for (auto &L : lists) Lmax = std::max(Lmax, L.size());
        for (auto &L : lists) {
            L.resize(Lmax, Tables::SegItem{0xFFFFFFFFu, 0, 0, 0});
            t.seg_exec.insert(t.seg_exec.end(), L.begin(), L.end());
        }
        t.seg_exec_ptr[i + 1] = (uint32_t)t.seg_exec.size();
        t.max_items_per_angle = std::max(t.max_items_per_angle, next_id);
    }
}

struct RowSeg { uint32_t tile, cnt; };

// Per row: entries stably regrouped by image tile (ascending pixel inside a tile, the order the row-driven kernels
// use), written CSR-aligned into tmp_lpix (pixel index inside the tile, y-major) / tmp_w; rsegs[r] lists the row's
// (tile, entry count) runs in ascending tile order.  Threads over rows.
static void split_rows_by_tile(const Coo &m, int N, int64_t nrows, int TY, int TZ, int tiles_z, std::vector<uint32_t> &tmp_lpix,
                               std::vector<float> &tmp_w, std::vector<std::vector<RowSeg>> &rsegs)
{
    const int64_t nnz = m.ptr[nrows];
    tmp_lpix.assign(nnz ? nnz : 1, 0u);
    tmp_w.assign(nnz ? nnz : 1, 0.f);
    rsegs.assign(nrows, {});
    unsigned hw = builder_threads();
    int nth = (int)std::min<int64_t>(std::max(1u, std::min(hw, 32u)), std::max<int64_t>(1, nrows / 256));
    auto work = [&](int th) {
        std::vector<std::pair<uint32_t, uint32_t>> key;   // (tile, position in row)
        for (int64_t r = nrows * th / nth; r < nrows * (th + 1) / nth; ++r) {
            int64_t b = m.ptr[r], e = m.ptr[r + 1];
            int n = (int)(e - b);
            key.resize(n);
            for (int k = 0; k < n; ++k) {
                uint32_t p = m.col[b + k];
                uint32_t y = p / (uint32_t)N, z = p - y * (uint32_t)N;
                key[k] = {(y / TY) * (uint32_t)tiles_z + z / TZ, (uint32_t)k};
            }
            std::sort(key.begin(), key.end());
            auto &rs = rsegs[r];
            for (int k = 0; k < n; ++k) {
                uint32_t p = m.col[b + key[k].second];
                uint32_t y = p / (uint32_t)N, z = p - y * (uint32_t)N;
                tmp_lpix[b + k] = (y % TY) * (uint32_t)TZ + z % TZ;
                tmp_w[b + k] = m.val[b + key[k].second];
                if (rs.empty() || rs.back().tile != key[k].first) rs.push_back({key[k].first, 0u});
                rs.back().cnt++;
            }
        }
    };
    std::vector<std::thread> thr;
    for (int th = 1; th < nth; ++th) thr.emplace_back(work, th);
    work(0);
    for (auto &x : thr) x.join();
}

// Tile tables of the all-angle forward projector (see sysmat.h).  The segments of a tile go longest first to the
// stream with the least work so far, so the 64 lane groups of a workgroup finish together.
void build_tiles(const Coo &m, int N, int P, int TY, int TZ, int pixel_bytes, Tables &t)
{
    constexpr int NS = Tables::TILE_SLOTS, NB = Tables::TILE_BATCH;
    const int64_t nrows = (int64_t)N * P;
    const int64_t nnz = m.ptr[nrows];
    t.tile_ty = TY; t.tile_tz = TZ;
    t.tiles_y = (N + TY - 1) / TY; t.tiles_z = (N + TZ - 1) / TZ;
    const uint32_t ntiles = (uint32_t)t.tiles_y * t.tiles_z;
    std::vector<uint32_t> tmp_lpix;
    std::vector<float> tmp_w;
    std::vector<std::vector<RowSeg>> rsegs;
    split_rows_by_tile(m, N, nrows, TY, TZ, t.tiles_z, tmp_lpix, tmp_w, rsegs);
    unsigned hw = builder_threads();
    (void)nnz;
    // pass 2: bucket the segments by tile
    struct Ref { uint32_t cnt, row, src; };               // src = offset of the segment's entries in the temp
    std::vector<uint32_t> tptr(ntiles + 1, 0);
    size_t nseg = 0;
    for (int64_t r = 0; r < nrows; ++r) { for (auto &s : rsegs[r]) tptr[s.tile + 1]++; nseg += rsegs[r].size(); }
    for (uint32_t k = 0; k < ntiles; ++k) tptr[k + 1] += tptr[k];
    std::vector<Ref> refs(nseg ? nseg : 1);
    {
        std::vector<uint32_t> fill(tptr.begin(), tptr.end() - 1);
        for (int64_t r = 0; r < nrows; ++r) {
            uint32_t src = (uint32_t)m.ptr[r];
            for (auto &s : rsegs[r]) { refs[fill[s.tile]++] = {s.cnt, (uint32_t)r, src}; src += s.cnt; }
        }
    }
    t.tile_nseg = (uint32_t)nseg;
    t.rseg_ptr.assign(nrows + 1, 0);
    for (int64_t r = 0; r < nrows; ++r) t.rseg_ptr[r + 1] = t.rseg_ptr[r] + (uint32_t)rsegs[r].size();
    t.rseg_idx.assign(nseg ? nseg : 1, 0);
    t.tile_slot_ptr.assign((size_t)ntiles * NS + 1, 0);
    t.tile_slot_seg0.assign((size_t)ntiles * NS, 0);
    // pass 3 (threads over tiles): deal segments to streams; sizes first, then the streams themselves
    std::vector<std::vector<uint32_t>> deal(ntiles);      // per tile: refs index per stream, concatenated; sizes in cnts
    std::vector<uint32_t> nbatch_of(ntiles * (size_t)NS, 0), nseg_of(ntiles * (size_t)NS, 0);
    int nth2 = (int)std::min<uint32_t>(std::max(1u, std::min(hw, 32u)), std::max(1u, ntiles / 8));
    auto deal_work = [&](int th) {
        std::vector<uint32_t> order;
        for (uint32_t k = ntiles * (uint64_t)th / nth2; k < ntiles * (uint64_t)(th + 1) / nth2; ++k) {
            uint32_t b = tptr[k], e = tptr[k + 1];
            order.resize(e - b);
            for (uint32_t i = 0; i < e - b; ++i) order[i] = b + i;
            std::sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) {
                return refs[x].cnt != refs[y].cnt ? refs[x].cnt > refs[y].cnt : refs[x].row < refs[y].row; });
            uint32_t load[NS] = {0};
            std::vector<uint32_t> lists[NS];
            for (uint32_t idx : order) {
                int best = 0;
                for (int q = 1; q < NS; ++q) if (load[q] < load[best]) best = q;
                load[best] += (refs[idx].cnt + NB - 1) / NB;
                lists[best].push_back(idx);
            }
            auto &d = deal[k];
            d.reserve(e - b);
            for (int q = 0; q < NS; ++q) {
                nbatch_of[(size_t)k * NS + q] = load[q];
                nseg_of[(size_t)k * NS + q] = (uint32_t)lists[q].size();
                d.insert(d.end(), lists[q].begin(), lists[q].end());
            }
        }
    };
    {
        std::vector<std::thread> thr;
        for (int th = 1; th < nth2; ++th) thr.emplace_back(deal_work, th);
        deal_work(0);
        for (auto &x : thr) x.join();
    }
    uint32_t segid = 0;
    for (size_t q = 0; q < (size_t)ntiles * NS; ++q) {
        t.tile_slot_ptr[q + 1] = t.tile_slot_ptr[q] + nbatch_of[q];
        t.tile_slot_seg0[q] = segid;
        segid += nseg_of[q];
    }
    // The kernel's ring prefetches up to 15 batches past the point where the LONGEST stream of a wave ends, counted from
    // each lane group's own stream start: pad by the longest stream + 16 batches so the last streams stay in bounds.
    uint32_t longest = 0;
    for (size_t q = 0; q < (size_t)ntiles * NS; ++q) longest = std::max(longest, nbatch_of[q]);
    const size_t nent = ((size_t)t.tile_slot_ptr.back() + longest + 16) * NB;
    t.tile_off.assign(nent, 0u);
    t.tile_w.assign(nent, 0.f);
    auto emit_work = [&](int th) {
        for (uint32_t k = ntiles * (uint64_t)th / nth2; k < ntiles * (uint64_t)(th + 1) / nth2; ++k) {
            const auto &d = deal[k];
            size_t di = 0;
            for (int q = 0; q < NS; ++q) {
                size_t slot = (size_t)k * NS + q;
                size_t o = (size_t)t.tile_slot_ptr[slot] * NB;
                uint32_t id = t.tile_slot_seg0[slot];
                for (uint32_t i = 0; i < nseg_of[slot]; ++i, ++id) {
                    const Ref &f = refs[d[di++]];
                    uint32_t nb = (f.cnt + NB - 1) / NB;
                    for (uint32_t j = 0; j < nb * NB; ++j) {
                        uint32_t flag = (j >= (nb - 1) * NB) ? 0x80000000u : 0u;
                        if (j < f.cnt) { t.tile_off[o] = (tmp_lpix[f.src + j] * (uint32_t)pixel_bytes) | flag; t.tile_w[o] = tmp_w[f.src + j]; }
                        else { t.tile_off[o] = ((uint32_t)(TY * TZ) * (uint32_t)pixel_bytes) | flag; t.tile_w[o] = 0.f; }   // the image's spare zero pixel
                        ++o;
                    }
                    // a row meets a tile once, so (row, tile) finds the slot in the row's ascending-tile list
                    const auto &rs = rsegs[f.row];
                    size_t lo = 0, hi = rs.size();
                    while (lo + 1 < hi) { size_t mid = (lo + hi) / 2; if (rs[mid].tile <= k) lo = mid; else hi = mid; }
                    t.rseg_idx[t.rseg_ptr[f.row] + lo] = id;
                }
            }
        }
    };
    {
        std::vector<std::thread> thr;
        for (int th = 1; th < nth2; ++th) thr.emplace_back(emit_work, th);
        emit_work(0);
        for (auto &x : thr) x.join();
    }
}

// Back-projection tile tables (see sysmat.h); needs t.cell (build_tables).
void build_bp_tiles(int N, int P, int TY, int TZ, int stage_angles, int max_rows, int row_bytes, int pad_angles, Tables &t)
{
    const int64_t npix = (int64_t)N * N;
    const int tiles_y = (N + TY - 1) / TY, tiles_z = (N + TZ - 1) / TZ;
    const uint32_t ntiles = (uint32_t)tiles_y * tiles_z;
    const int TP = TY * TZ;
    const uint32_t zero_off = (uint32_t)(stage_angles * max_rows) * (uint32_t)row_bytes;   // the zero row behind a stage buffer
    t.bp_win.assign((size_t)ntiles * P, 0);
    t.bp_cell.assign(((size_t)ntiles * P + pad_angles) * TP, Tables::TileCell{zero_off, 0.f, zero_off, 0.f});
    std::vector<uint8_t> bad(ntiles, 0);
    unsigned hw = builder_threads();
    int nth = (int)std::min<uint32_t>(std::max(1u, std::min(hw, 32u)), std::max(1u, ntiles / 8));
    auto work = [&](int th) {
        for (uint32_t k = ntiles * (uint64_t)th / nth; k < ntiles * (uint64_t)(th + 1) / nth; ++k) {
            int y0 = (int)(k / tiles_z) * TY, z0 = (int)(k % tiles_z) * TZ;
            for (int i = 0; i < P; ++i) {
                const Cell *ci = t.cell.data() + (size_t)i * npix;
                uint32_t lo = 0xFFFFFFFFu, hi = 0;
                for (int ly = 0; ly < TY && y0 + ly < N; ++ly)
                    for (int lz = 0; lz < TZ && z0 + lz < N; ++lz) {
                        const Cell &c = ci[(int64_t)(y0 + ly) * N + z0 + lz];
                        if (c.w0 != 0.f) { lo = std::min(lo, c.r0); hi = std::max(hi, c.r0); }
                        if (c.w1 != 0.f) { lo = std::min(lo, c.r1); hi = std::max(hi, c.r1); }
                    }
                uint32_t nr = (lo == 0xFFFFFFFFu) ? 0u : hi - lo + 1;
                if (nr == 0) lo = 0;
                if (nr > (uint32_t)max_rows) { bad[k] = 1; nr = 0; }
                t.bp_win[(size_t)k * P + i] = lo | (nr << 16);
                if (bad[k]) continue;
                Tables::TileCell *out = t.bp_cell.data() + ((size_t)k * P + i) * TP;
                uint32_t slot_base = (uint32_t)(i % stage_angles) * (uint32_t)max_rows * (uint32_t)row_bytes;
                for (int ly = 0; ly < TY && y0 + ly < N; ++ly)
                    for (int lz = 0; lz < TZ && z0 + lz < N; ++lz) {
                        const Cell &c = ci[(int64_t)(y0 + ly) * N + z0 + lz];
                        Tables::TileCell &o = out[ly * TZ + lz];
                        if (c.w0 != 0.f) { o.off0 = slot_base + (c.r0 - lo) * (uint32_t)row_bytes; o.w0 = c.w0; }
                        if (c.w1 != 0.f) { o.off1 = slot_base + (c.r1 - lo) * (uint32_t)row_bytes; o.w1 = c.w1; }
                    }
            }
        }
    };
    std::vector<std::thread> thr;
    for (int th = 1; th < nth; ++th) thr.emplace_back(work, th);
    work(0);
    for (auto &x : thr) x.join();
    t.bp_tile_ok = true;
    for (uint32_t k = 0; k < ntiles; ++k) if (bad[k]) t.bp_tile_ok = false;
}

// Entry lists of k_bp_list (see sysmat.h).
void build_bp_lists(int N, int P, int TY, int TZ, int stage_angles, int max_rows, int row_bytes, int waves, int batch, int regs_per_pixel, Tables &t)
{
    const int64_t npix = (int64_t)N * N;
    const int tiles_y = (N + TY - 1) / TY, tiles_z = (N + TZ - 1) / TZ;
    const uint32_t ntiles = (uint32_t)tiles_y * tiles_z;
    const int TP = TY * TZ, ppw = TP / waves, nstage = (P + stage_angles - 1) / stage_angles;
    const uint32_t buf_bytes = (uint32_t)(stage_angles * max_rows) * (uint32_t)row_bytes;       // stage s sits in LDS buffer s & 1
    const size_t nlist = (size_t)ntiles * nstage * waves;
    t.bl_ok = false;
    t.bl_nbatch = 0;
    t.bl_ptr.clear();
    t.bl_win.assign((size_t)ntiles * P, 0);
    unsigned hw = builder_threads();
    int nth = (int)std::min<uint32_t>(std::max(1u, std::min(hw, 32u)), std::max(1u, ntiles / 8));
    std::vector<uint32_t> nb(nlist, 0);
    std::vector<uint8_t> bad(nth, 0);
    // pass 0: the window {first ray | rays << 16} of every (tile, angle) and the batches of every list; pass 1 (after the prefix
    // sum) writes the lists
    auto work = [&](int th, int pass) {
        struct Ent { uint32_t ray, q; float w; };
        std::vector<Ent> ents;
        for (uint32_t k = ntiles * (uint64_t)th / nth; k < ntiles * (uint64_t)(th + 1) / nth; ++k) {
            const int y0 = (int)(k / tiles_z) * TY, z0 = (int)(k % tiles_z) * TZ;
            if (!pass)
                for (int i = 0; i < P; ++i) {
                    const Cell *ci = t.cell.data() + (size_t)i * npix;
                    uint32_t lo = 0xFFFFFFFFu, hi = 0;
                    for (int ly = 0; ly < TY && y0 + ly < N; ++ly)
                        for (int lz = 0; lz < TZ && z0 + lz < N; ++lz) {
                            const Cell &c = ci[(int64_t)(y0 + ly) * N + z0 + lz];
                            if (c.w0 != 0.f) { lo = std::min(lo, c.r0); hi = std::max(hi, c.r0); }
                            if (c.w1 != 0.f) { lo = std::min(lo, c.r1); hi = std::max(hi, c.r1); }
                            if (c.w0 != 0.f && c.w1 != 0.f && c.r1 <= c.r0) bad[th] = 1;   // (rows are worked on in ascending order)
                        }
                    uint32_t nr = (lo == 0xFFFFFFFFu) ? 0u : hi - lo + 1;
                    if (nr == 0) lo = 0;
                    if (nr > (uint32_t)max_rows || lo > 0xFFFFu) { bad[th] = 1; nr = 0; }
                    t.bl_win[(size_t)k * P + i] = lo | (nr << 16);
                }
            for (int s = 0; s < nstage; ++s)
                for (int w = 0; w < waves; ++w) {
                    const size_t li = ((size_t)k * nstage + s) * waves + w;
                    uint64_t *out = pass ? t.bl_ent.get() + (size_t)t.bl_ptr[li] * batch * 2 : nullptr;
                    uint32_t npair = 0, last_off = 0;
                    for (int i = s * stage_angles; i < std::min(P, (s + 1) * stage_angles); ++i) {
                        const Cell *ci = t.cell.data() + (size_t)i * npix;
                        const uint32_t lo = t.bl_win[(size_t)k * P + i] & 0xFFFFu;
                        const uint32_t slot_base = (uint32_t)(s & 1) * buf_bytes + (uint32_t)(i % stage_angles) * (uint32_t)max_rows * (uint32_t)row_bytes;
                        ents.clear();
                        for (int q = 0; q < ppw; ++q) {
                            int ly, lz;
                            Tables::bl_pixel(w, q, ly, lz);
                            if (y0 + ly >= N || z0 + lz >= N) continue;
                            const Cell &c = ci[(int64_t)(y0 + ly) * N + z0 + lz];
                            if (c.w0 != 0.f) ents.push_back({c.r0, (uint32_t)q, c.w0});
                            if (c.w1 != 0.f) ents.push_back({c.r1, (uint32_t)q, c.w1});
                        }
                        // rows ascending (a pixel's first ray is its lower one), pixels ascending inside a row; a row's entries go out
                        // in pairs that share one read of the row
                        std::stable_sort(ents.begin(), ents.end(), [](const Ent &a, const Ent &b) { return a.ray < b.ray; });
                        for (size_t j = 0; j < ents.size();) {
                            const bool two = j + 1 < ents.size() && ents[j + 1].ray == ents[j].ray;
                            if (pass) {
                                const uint32_t off = slot_base + (ents[j].ray - lo) * (uint32_t)row_bytes;
                                uint32_t w0b, w1b = 0, q1 = 0;
                                std::memcpy(&w0b, &ents[j].w, 4);
                                if (two) { std::memcpy(&w1b, &ents[j + 1].w, 4); q1 = ents[j + 1].q * regs_per_pixel; }
                                out[2 * npair] = ((uint64_t)w0b << 32) | (uint64_t)(off | (ents[j].q * regs_per_pixel));
                                out[2 * npair + 1] = ((uint64_t)w1b << 32) | (uint64_t)q1;
                                last_off = off;
                            }
                            ++npair;
                            j += two ? 2 : 1;
                        }
                    }
                    const uint32_t batches = (npair + batch - 1) / batch;
                    // padding: weight 0 on the row of the list's last pair (staged, so finite), into accumulator 0
                    if (pass) for (uint32_t j = npair; j < batches * (uint32_t)batch; ++j) { out[2 * j] = last_off; out[2 * j + 1] = 0; }
                    else nb[li] = batches;
                }
        }
    };
    auto run = [&](int pass) {
        std::vector<std::thread> thr;
        for (int th = 1; th < nth; ++th) thr.emplace_back(work, th, pass);
        work(0, pass);
        for (auto &x : thr) x.join();
    };
    run(0);
    for (int th = 0; th < nth; ++th) if (bad[th]) return;              // a window the kernel's LDS cannot hold: the cell form stays
    t.bl_ptr.assign(nlist + 1, 0);
    uint64_t tot = 0;
    for (size_t i = 0; i < nlist; ++i) { t.bl_ptr[i] = (uint32_t)tot; tot += nb[i]; }
    t.bl_ptr[nlist] = (uint32_t)tot;
    if (tot >= 0xFFFFFFFFull) { t.bl_ptr.clear(); return; }
    t.bl_nbatch = tot;
    t.bl_ent.reset(new uint64_t[(size_t)(tot + 1) * batch * 2]);
    for (int j = 0; j < batch * 2; ++j) t.bl_ent[(size_t)tot * batch * 2 + j] = 0;  // (prefetched behind the last list, never worked on)
    run(1);
    t.bl_ok = true;
}

// Per-angle tile tables of the fused SART step (see sysmat.h); needs t.cell (build_tables).
void build_sart_tiles(const Coo &m, int N, int P, int TY, int TZ, int max_rows, int pixel_bytes, Tables &t)
{
    constexpr int NB = Tables::TILE_BATCH, MS = Tables::ST_MAXSEG;
    const int64_t nrows = (int64_t)N * P, npix = (int64_t)N * N;
    const int tiles_y = (N + TY - 1) / TY, tiles_z = (N + TZ - 1) / TZ;
    const uint32_t ntiles = (uint32_t)tiles_y * tiles_z;
    const int TP = TY * TZ;
    t.st_ty = TY; t.st_tz = TZ; t.st_tiles = (int)ntiles; t.st_tiles_z = tiles_z; t.st_maxr = max_rows;
    const uint32_t zero_row = (uint32_t)max_rows * (uint32_t)pixel_bytes, zero_pix = (uint32_t)TP * (uint32_t)pixel_bytes;
    std::vector<uint32_t> tmp_lpix;
    std::vector<float> tmp_w;
    std::vector<std::vector<RowSeg>> rsegs;
    split_rows_by_tile(m, N, nrows, TY, TZ, tiles_z, tmp_lpix, tmp_w, rsegs);
    t.st_cell.assign((size_t)P * ntiles * TP, Tables::TileCell{zero_row, 0.f, zero_row, 0.f});
    t.st_win.assign((size_t)P * ntiles, 0);
    t.st_segid.assign((size_t)P * ntiles * MS, 0);
    t.st_seg.assign((size_t)P * ntiles * MS * 2, 0);
    t.st_row_first.assign(nrows, 0);
    t.st_row_nseg.assign(nrows, 0);
    // batches per angle (prefix over angles), so that the angles can be emitted in parallel
    std::vector<uint64_t> angle_batches(P + 1, 0);
    for (int i = 0; i < P; ++i) {
        uint64_t nb = 0;
        for (int j = 0; j < N; ++j) for (auto &s : rsegs[(int64_t)i * N + j]) nb += (s.cnt + NB - 1) / NB;
        angle_batches[i + 1] = angle_batches[i] + nb;
    }
    t.st_off.assign((size_t)(angle_batches[P] + 1) * NB, zero_pix);
    t.st_w.assign((size_t)(angle_batches[P] + 1) * NB, 0.f);
    std::vector<uint8_t> bad(P, 0);
    std::vector<uint32_t> ids_of(P, 0);
    unsigned hw = builder_threads();
    int nth = (int)std::min<int>(std::max(1u, std::min(hw, 32u)), P);
    auto work = [&](int th) {
        std::vector<uint32_t> nseg(ntiles), fill(ntiles);
        for (int i = th; i < P; i += nth) {
            // segments of angle i per tile, in ascending row order
            std::fill(nseg.begin(), nseg.end(), 0u);
            for (int j = 0; j < N; ++j) for (auto &s : rsegs[(int64_t)i * N + j]) nseg[s.tile]++;
            uint32_t id = 0;
            for (uint32_t k = 0; k < ntiles; ++k) if (nseg[k] > (uint32_t)MS) bad[i] = 1;
            for (int j = 0; j < N; ++j) {
                int64_t r = (int64_t)i * N + j;
                t.st_row_first[r] = id; t.st_row_nseg[r] = (uint32_t)rsegs[r].size();
                id += (uint32_t)rsegs[r].size();
            }
            ids_of[i] = id;
            if (bad[i]) continue;
            std::fill(fill.begin(), fill.end(), 0u);
            uint64_t batch = angle_batches[i];
            for (int j = 0; j < N; ++j) {
                int64_t r = (int64_t)i * N + j;
                uint32_t src = (uint32_t)m.ptr[r];
                uint32_t q = 0;
                for (auto &s : rsegs[r]) {
                    uint32_t k = fill[s.tile]++;
                    uint32_t nb = (s.cnt + NB - 1) / NB;
                    size_t slot = (((size_t)i * ntiles + s.tile) * MS + k) * 2;
                    t.st_seg[slot] = (uint32_t)batch; t.st_seg[slot + 1] = nb;
                    for (uint32_t e = 0; e < s.cnt; ++e) {
                        t.st_off[(size_t)batch * NB + e] = tmp_lpix[src + e] * (uint32_t)pixel_bytes;
                        t.st_w[(size_t)batch * NB + e] = tmp_w[src + e];
                    }
                    t.st_segid[((size_t)i * ntiles + s.tile) * MS + k] = t.st_row_first[r] + q++;
                    batch += nb; src += s.cnt;
                }
            }
            // ray windows and cells
            const Cell *ci = t.cell.data() + (size_t)i * npix;
            for (uint32_t k = 0; k < ntiles; ++k) {
                int y0 = (int)(k / tiles_z) * TY, z0 = (int)(k % tiles_z) * TZ;
                uint32_t lo = 0xFFFFFFFFu, hi = 0;
                for (int ly = 0; ly < TY && y0 + ly < N; ++ly)
                    for (int lz = 0; lz < TZ && z0 + lz < N; ++lz) {
                        const Cell &c = ci[(int64_t)(y0 + ly) * N + z0 + lz];
                        if (c.w0 != 0.f) { lo = std::min(lo, c.r0); hi = std::max(hi, c.r0); }
                        if (c.w1 != 0.f) { lo = std::min(lo, c.r1); hi = std::max(hi, c.r1); }
                    }
                uint32_t nr = (lo == 0xFFFFFFFFu) ? 0u : hi - lo + 1;
                if (nr == 0) lo = 0;
                if (nr > (uint32_t)max_rows) { bad[i] = 1; continue; }
                t.st_win[(size_t)i * ntiles + k] = lo | (nr << 16);
                Tables::TileCell *out = t.st_cell.data() + ((size_t)i * ntiles + k) * TP;
                for (int ly = 0; ly < TY && y0 + ly < N; ++ly)
                    for (int lz = 0; lz < TZ && z0 + lz < N; ++lz) {
                        const Cell &c = ci[(int64_t)(y0 + ly) * N + z0 + lz];
                        Tables::TileCell &o = out[ly * TZ + lz];
                        if (c.w0 != 0.f) { o.off0 = (c.r0 - lo) * (uint32_t)pixel_bytes; o.w0 = c.w0; }
                        if (c.w1 != 0.f) { o.off1 = (c.r1 - lo) * (uint32_t)pixel_bytes; o.w1 = c.w1; }
                    }
            }
        }
    };
    std::vector<std::thread> thr;
    for (int th = 1; th < nth; ++th) thr.emplace_back(work, th);
    work(0);
    for (auto &x : thr) x.join();
    t.st_ok = true;
    t.st_max_ids = 0;
    for (int i = 0; i < P; ++i) { if (bad[i]) t.st_ok = false; t.st_max_ids = std::max(t.st_max_ids, ids_of[i]); }
}


// ---- sheared-strip tables of the all-angle forward projector (see sysmat.h; kernel: k_fp_strip) -------------------------------
namespace {
struct FsSeg {                 // one ray inside one strip
    uint32_t row, q;           // matrix row; index of this strip in the row's ascending-strip list
    int32_t ang, j;            // angle, ray number within the angle
    uint32_t t0, t1;           // first / last tile with entries
    uint32_t off, cnt;         // entries [off, off + cnt) of the per-pass sorted entry arrays
    int32_t wave, k, sub;      // accumulator slot
};
struct FsRowSeg { int32_t strip; uint32_t off, cnt, t0, t1; };
struct FsWork {                // per item, between the sizing and the emission phase
    std::vector<FsSeg> segs;
    std::vector<uint8_t> nb;   // [ntiles][WAVES][16]
    uint32_t tile0 = 0, ntiles = 0, quads = 0;
    uint32_t wave_batches[Tables::FS_WAVES] = {0}, wave_halves[Tables::FS_WAVES] = {0}, kused = 0;
    uint32_t group_segs[Tables::FS_GROUPS] = {0};
    int32_t pass = 0, strip = 0;
};
template <typename F> void fs_parallel(size_t n, unsigned nth, F f)
{
    nth = (unsigned)std::max<size_t>(1, std::min<size_t>(nth, n));
    std::vector<std::thread> thr;
    for (unsigned th = 1; th < nth; ++th) thr.emplace_back([&, th] { for (size_t i = n * th / nth; i < n * (th + 1) / nth; ++i) f(i); });
    for (size_t i = 0; i < n / nth; ++i) f(i);
    for (auto &x : thr) x.join();
}
}  // namespace

bool build_fp_strips(const Coo &m, int N, int P, int pixel_bytes, int nchunk, Tables &t, std::string &why)
{
    constexpr int W = Tables::FS_W, H = Tables::FS_H, WAVES = Tables::FS_WAVES, GROUPS = Tables::FS_GROUPS, KMAX = Tables::FS_KMAX;
    constexpr int NB = Tables::TILE_BATCH;
    static_assert(GROUPS == WAVES * 4 && KMAX <= 16, "slot layout");
    t.fs_ok = false;
    const bool timing = std::getenv("TOMO_FS_TIMING") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_last = now();
    auto lap = [&](const char *what) { if (timing) { double x = now(); std::fprintf(stderr, "build_fp_strips: %-28s %.3f s\n", what, x - t_last); t_last = x; } };
    const int64_t nrows = (int64_t)N * P;
    const int64_t nnz = m.ptr[nrows];
    if (N < 1 || P < 1 || nnz <= 0 || N > 32768) { why = "empty geometry"; return false; }
    const unsigned hw = std::min(builder_threads(), 32u);
    // ---- 1. direction of every angle, from the matrix itself (a user matrix has no angle list): orientation = the axis its
    // longest ray advances along, slope = least-squares dv/du of that ray's pixels
    std::vector<int> orient(P, 0);
    std::vector<double> slope(P, 0.0);
    for (int i = 0; i < P; ++i) {
        int64_t best = (int64_t)i * N;
        for (int j = 0; j < N; ++j) { int64_t r = (int64_t)i * N + j; if (m.ptr[r + 1] - m.ptr[r] > m.ptr[best + 1] - m.ptr[best]) best = r; }
        const int64_t b = m.ptr[best], e = m.ptr[best + 1];
        if (e - b < 2) continue;
        int ymin = N, ymax = -1, zmin = N, zmax = -1;
        for (int64_t k = b; k < e; ++k) { int y = (int)(m.col[k] / (uint32_t)N), z = (int)(m.col[k] % (uint32_t)N); ymin = std::min(ymin, y); ymax = std::max(ymax, y); zmin = std::min(zmin, z); zmax = std::max(zmax, z); }
        orient[i] = (ymax - ymin >= zmax - zmin) ? 0 : 1;
        double su = 0, sv = 0, suu = 0, suv = 0; const double n = (double)(e - b);
        for (int64_t k = b; k < e; ++k) {
            int y = (int)(m.col[k] / (uint32_t)N), z = (int)(m.col[k] % (uint32_t)N);
            double u = orient[i] ? z : y, v = orient[i] ? y : z;
            su += u; sv += v; suu += u * u; suv += u * v;
        }
        const double den = n * suu - su * su;
        slope[i] = den > 0 ? (n * suv - su * sv) / den : 0.0;
    }
    // ---- 2. passes: angles of one orientation, neighbouring slopes
    struct Pass { int orient; double tg; std::vector<int> ang; };
    std::vector<Pass> passes;
    auto form_passes = [&](double dt_max, int amax) {
        passes.clear();
        for (int o = 0; o < 2; ++o) {
            std::vector<int> a;
            for (int i = 0; i < P; ++i) if (orient[i] == o) a.push_back(i);
            std::sort(a.begin(), a.end(), [&](int x, int y) { return slope[x] != slope[y] ? slope[x] < slope[y] : x < y; });
            size_t k = 0;
            while (k < a.size()) {
                size_t e = k + 1;
                while (e < a.size() && (int)(e - k) < amax && slope[a[e]] - slope[a[k]] <= dt_max) ++e;
                Pass ps; ps.orient = o; ps.tg = 0.5 * (slope[a[k]] + slope[a[e - 1]]);
                ps.ang.assign(a.begin() + k, a.begin() + e);
                passes.push_back(std::move(ps));
                k = e;
            }
        }
    };
    double dt_max = 0.72; int amax = 24;
    if (const char *sdt = std::getenv("TOMO_FS_DT")) { double v = std::atof(sdt); if (v > 0) dt_max = v; }
    if (const char *sa = std::getenv("TOMO_FS_AMAX")) { int v = std::atoi(sa); if (v > 0) amax = v; }
    // per-ray sorted entries (every ray belongs to exactly one pass, so one set of arrays serves all passes)
    std::vector<uint16_t> eu(nnz);
    std::vector<uint8_t> elv(nnz);
    std::vector<float> ew(nnz);
    std::vector<std::vector<FsRowSeg>> rsegs;
    std::vector<FsWork> work;
    std::vector<int32_t> shift;
    const int OFF = N;                                     // keeps v - shift + OFF in [0, 3N) (|shift| <= N)
    int nsegs = 1, seglen = 1 << 30;                       // segments per strip, tiles per segment
    for (int attempt = 0; attempt < 6; ++attempt) {
        form_passes(dt_max, amax);
        const int npass = (int)passes.size();
        shift.assign((size_t)npass * N, 0);
        std::vector<int> pass_of(P, 0);
        for (int ps = 0; ps < npass; ++ps) {
            for (int a : passes[ps].ang) pass_of[a] = ps;
            for (int u = 0; u < N; ++u) {
                double sh = passes[ps].tg * (u - 0.5 * (N - 1));
                sh = std::max(-(double)N, std::min((double)N, sh));
                shift[(size_t)ps * N + u] = (int32_t)std::floor(sh + 0.5);
            }
        }
        // A strip is one long sequential march, and a launch wants several rounds of workgroups (512 are resident): strips are cut
        // along the march into `nsegs` segments of `seglen` tiles, each an item of its own (a ray pays one more partial sum per cut
        // it crosses).  The cut depends on the IMAGE SIZE ONLY -- half a strip, 4 ... 16 tiles -- never on how many slices
        // the engine holds: a ray's partial sums, and with them the rounding of its line integral, must be the same in a 64-slice
        // shard and in the whole volume (sharded == whole, two half-slab engines == one engine: bit for bit).
        {
            const int tiles_per_strip = (N + H - 1) / H;
            int len = std::max(4, std::min(16, tiles_per_strip / 2));
            if (const char *sl = std::getenv("TOMO_FS_SEGLEN")) { int v = std::atoi(sl); if (v > 0) len = v; }
            len = std::max(1, std::min(len, tiles_per_strip));
            nsegs = (tiles_per_strip + len - 1) / len;
            seglen = (tiles_per_strip + nsegs - 1) / nsegs;
            (void)nchunk;
        }
        // ---- 3. every ray: entries sorted by (strip segment, march coordinate, cross coordinate), cut into strip segments
        rsegs.assign(nrows, {});
        fs_parallel((size_t)nrows, hw, [&](size_t r) {
            const int i = (int)(r / N), ps = pass_of[i], o = passes[ps].orient;
            const int32_t *sh = shift.data() + (size_t)ps * N;
            const int64_t b = m.ptr[r], e = m.ptr[r + 1];
            const int n = (int)(e - b);
            if (n == 0) return;
            struct K { int32_t strip; uint16_t u; uint8_t lv; float w; };
            std::vector<K> key(n);
            for (int k = 0; k < n; ++k) {
                uint32_t p = m.col[b + k];
                int y = (int)(p / (uint32_t)N), z = (int)(p % (uint32_t)N);
                int u = o ? z : y, v = o ? y : z;
                int vs = v - sh[u] + OFF;
                key[k] = {(vs / W) * nsegs + (u / H) / seglen, (uint16_t)u, (uint8_t)(vs % W), m.val[b + k]};
            }
            std::sort(key.begin(), key.end(), [](const K &a, const K &c) { return a.strip != c.strip ? a.strip < c.strip : a.u != c.u ? a.u < c.u : a.lv < c.lv; });
            auto &rs = rsegs[r];
            for (int k = 0; k < n; ++k) {
                eu[b + k] = key[k].u; elv[b + k] = key[k].lv; ew[b + k] = key[k].w;
                if (rs.empty() || rs.back().strip != key[k].strip) rs.push_back({key[k].strip, (uint32_t)(b + k), 0u, (uint32_t)key[k].u / H, 0u});
                rs.back().cnt++; rs.back().t1 = (uint32_t)key[k].u / H;
            }
        });
        lap("rays sorted into strips");
        // ---- 4. bucket the ray segments by (pass, strip)
        const int nstrip_max = ((3 * N + W - 1) / W + 2) * nsegs;     // (strip, segment) keys
        std::vector<uint32_t> iptr((size_t)npass * nstrip_max + 1, 0);
        for (int64_t r = 0; r < nrows; ++r) { const int ps = pass_of[r / N]; for (auto &sg : rsegs[r]) iptr[(size_t)ps * nstrip_max + sg.strip + 1]++; }
        for (size_t k = 0; k + 1 < iptr.size(); ++k) iptr[k + 1] += iptr[k];
        struct Ref { uint32_t row, q; };
        std::vector<Ref> refs(iptr.back() ? iptr.back() : 1);
        {
            std::vector<uint32_t> fill(iptr.begin(), iptr.end() - 1);
            for (int64_t r = 0; r < nrows; ++r) {
                const int ps = pass_of[r / N];
                for (uint32_t q = 0; q < rsegs[r].size(); ++q) refs[fill[(size_t)ps * nstrip_max + rsegs[r][q].strip]++] = {(uint32_t)r, q};
            }
        }
        std::vector<size_t> bucket;                        // non-empty (pass, strip) buckets = items
        for (size_t k = 0; k + 1 < iptr.size(); ++k) if (iptr[k + 1] > iptr[k]) bucket.push_back(k);
        work.assign(bucket.size(), {});
        lap("segments bucketed");
        // ---- 5. per item: accumulator slots and batch counts (sizing phase)
        std::vector<uint8_t> over(bucket.size(), 0);
        fs_parallel(bucket.size(), hw, [&](size_t it) {
            FsWork &wk = work[it];
            const size_t bk = bucket[it];
            wk.pass = (int32_t)(bk / nstrip_max); wk.strip = (int32_t)(bk % nstrip_max);
            auto &segs = wk.segs;
            segs.reserve(iptr[bk + 1] - iptr[bk]);
            uint32_t lo = 0xFFFFFFFFu, hi = 0;
            for (uint32_t x = iptr[bk]; x < iptr[bk + 1]; ++x) {
                const Ref &f = refs[x];
                const FsRowSeg &rs = rsegs[f.row][f.q];
                segs.push_back({f.row, f.q, (int32_t)(f.row / N), (int32_t)(f.row % N), rs.t0, rs.t1, rs.off, rs.cnt, 0, 0, 0});
                lo = std::min(lo, rs.t0); hi = std::max(hi, rs.t1);
            }
            wk.tile0 = lo; wk.ntiles = hi - lo + 1;
            std::sort(segs.begin(), segs.end(), [](const FsSeg &a, const FsSeg &c) { return a.ang != c.ang ? a.ang < c.ang : a.j < c.j; });
            uint32_t qbase = 0;
            std::vector<int> jmin(wk.ntiles), jmax(wk.ntiles);
            std::vector<uint32_t> quad_of(segs.size());
            for (size_t a = 0; a < segs.size();) {
                size_t e = a;
                while (e < segs.size() && segs[e].ang == segs[a].ang) ++e;
                std::fill(jmin.begin(), jmin.end(), 1 << 30); std::fill(jmax.begin(), jmax.end(), -1);
                for (size_t x = a; x < e; ++x) for (uint32_t tt = segs[x].t0; tt <= segs[x].t1; ++tt) { jmin[tt - lo] = std::min(jmin[tt - lo], segs[x].j); jmax[tt - lo] = std::max(jmax[tt - lo], segs[x].j); }
                int span = 1;
                for (uint32_t tt = 0; tt < wk.ntiles; ++tt) if (jmax[tt] >= 0) span = std::max(span, jmax[tt] - jmin[tt] + 1);
                const int M = (span + 3) & ~3;
                for (size_t x = a; x < e; ++x) {
                    const int slot = segs[x].j % M;
                    quad_of[x] = qbase + (uint32_t)(slot >> 2);
                    segs[x].sub = slot & 3;
                }
                qbase += (uint32_t)(M >> 2);
                a = e;
            }
            wk.quads = qbase;
            if (qbase > (uint32_t)(WAVES * KMAX)) { over[it] = 1; return; }
            // half-batches (4 entries) of every quad in every tile: the four rays of a quad run the same loop, so the longest counts
            std::vector<uint8_t> hq((size_t)qbase * wk.ntiles, 0);
            for (size_t x = 0; x < segs.size(); ++x) {
                const FsSeg &sg = segs[x];
                uint32_t p = sg.off, end = sg.off + sg.cnt;
                while (p < end) {
                    const uint32_t tt = eu[p] / H;
                    uint32_t c = 0;
                    while (p < end && (uint32_t)eu[p] / H == tt) { ++p; ++c; }
                    uint8_t &h = hq[(size_t)quad_of[x] * wk.ntiles + (tt - lo)];
                    h = (uint8_t)std::max<uint32_t>(h, std::min<uint32_t>(254u, (c + NB / 2 - 1) / (NB / 2)));
                }
            }
            // quads -> (wave, slot): heaviest first, each to the wave (with a free slot) that keeps the per-tile loads most even --
            // the workgroup meets at a barrier after every tile, so a tile costs its slowest wave
            std::vector<uint32_t> qorder(qbase), qwork(qbase, 0), qwave(qbase), qk(qbase);
            for (uint32_t q = 0; q < qbase; ++q) { qorder[q] = q; for (uint32_t tt = 0; tt < wk.ntiles; ++tt) qwork[q] += hq[(size_t)q * wk.ntiles + tt]; }
            std::sort(qorder.begin(), qorder.end(), [&](uint32_t x, uint32_t y) { return qwork[x] != qwork[y] ? qwork[x] > qwork[y] : x < y; });
            std::vector<uint32_t> load((size_t)WAVES * wk.ntiles, 0);
            int used[WAVES] = {0};
            for (uint32_t q : qorder) {
                int best = -1; uint64_t bestc = ~0ull;
                for (int w = 0; w < WAVES; ++w) {
                    if (used[w] >= KMAX) continue;
                    uint64_t cst = 0;
                    for (uint32_t tt = 0; tt < wk.ntiles; ++tt) { uint64_t v = load[(size_t)w * wk.ntiles + tt] + hq[(size_t)q * wk.ntiles + tt]; cst += v * v - (uint64_t)load[(size_t)w * wk.ntiles + tt] * load[(size_t)w * wk.ntiles + tt]; }
                    if (cst < bestc) { bestc = cst; best = w; }
                }
                qwave[q] = (uint32_t)best; qk[q] = (uint32_t)used[best]++;
                for (uint32_t tt = 0; tt < wk.ntiles; ++tt) load[(size_t)best * wk.ntiles + tt] += hq[(size_t)q * wk.ntiles + tt];
            }
            wk.kused = 0;
            for (int w = 0; w < WAVES; ++w) wk.kused = std::max(wk.kused, (uint32_t)used[w]);
            wk.nb.assign((size_t)wk.ntiles * WAVES * 16, 0);
            for (uint32_t q = 0; q < qbase; ++q)
                for (uint32_t tt = 0; tt < wk.ntiles; ++tt) wk.nb[((size_t)tt * WAVES + qwave[q]) * 16 + qk[q]] = hq[(size_t)q * wk.ntiles + tt];
            for (size_t x = 0; x < segs.size(); ++x) {
                segs[x].wave = (int32_t)qwave[quad_of[x]]; segs[x].k = (int32_t)qk[quad_of[x]];
                wk.group_segs[segs[x].wave * 4 + segs[x].sub]++;
            }
            // stream units (64 bytes: a batch of 8 entries, or a half batch stored twice) and work (half-batches) per wave
            for (uint32_t tt = 0; tt < wk.ntiles; ++tt) for (int w = 0; w < WAVES; ++w) for (int k = 0; k < 16; ++k) {
                const uint32_t h = wk.nb[((size_t)tt * WAVES + w) * 16 + k];
                wk.wave_batches[w] += (h + 1) >> 1; wk.wave_halves[w] += h;
            }
        });
        lap("slots and counts");
        bool any_over = false;
        for (uint8_t o : over) any_over |= o != 0;
        if (!any_over) break;
        if (attempt == 5) { why = "a strip needs more accumulator slots than a workgroup has"; return false; }
        dt_max *= 0.7; amax = std::max(1, amax * 3 / 4);   // narrower passes: fewer rays alive per strip
    }
    const int npass = (int)passes.size();
    // ---- 6. layout: items heaviest first; per-item offsets
    const size_t nitems = work.size();
    std::vector<uint32_t> order(nitems);
    for (size_t i = 0; i < nitems; ++i) order[i] = (uint32_t)i;
    auto crit = [&](const FsWork &w) { uint32_t c = 0; for (int k = 0; k < WAVES; ++k) c = std::max(c, w.wave_halves[k]); return c + 8 * w.ntiles; };
    std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { uint32_t ca = crit(work[a]), cb = crit(work[b]); return ca != cb ? ca > cb : a < b; });
    t.fs_item.assign(nitems, {});
    t.fs_gstart.assign(nitems * GROUPS + 1, 0);
    t.fs_gseg0.assign(nitems * GROUPS + 1, 0);
    uint64_t nbatch = 0, nseg = 0, ncnt = 0;
    int kused = 1;
    for (size_t o = 0; o < nitems; ++o) {
        const FsWork &wk = work[order[o]];
        Tables::FsItem &itx = t.fs_item[o];
        itx.pass = wk.pass; itx.v0 = (wk.strip / nsegs) * W - OFF; itx.tile0 = wk.tile0; itx.ntiles = wk.ntiles;
        itx.cnt0 = (uint32_t)ncnt; itx.g0 = (uint32_t)(o * GROUPS); itx.work = crit(wk); itx.pad = 0;
        ncnt += (uint64_t)wk.ntiles * WAVES;
        kused = std::max<int>(kused, (int)wk.kused);
        for (int g = 0; g < GROUPS; ++g) {
            t.fs_gstart[o * GROUPS + g] = (uint32_t)nbatch; t.fs_gseg0[o * GROUPS + g] = (uint32_t)nseg;
            nbatch += wk.wave_batches[g >> 2]; nseg += wk.group_segs[g];
        }
    }
    if (nbatch + 8 >= (1ull << 32) / NB || nseg >= (1ull << 32) || ncnt >= (1ull << 28)) { why = "strip tables exceed 32-bit offsets"; return false; }
    t.fs_gstart[nitems * GROUPS] = (uint32_t)nbatch; t.fs_gseg0[nitems * GROUPS] = (uint32_t)nseg;
    t.fs_kused = kused;
    t.fs_nseg = (uint32_t)nseg;
    t.fs_cnt.assign((size_t)ncnt * 16 + 16, 0);
    const uint32_t zero_off = (uint32_t)(W * H) * (uint32_t)pixel_bytes;
    t.fs_ent_n = (size_t)(nbatch + 16) * NB;                  // + the kernel's prefetch distance past the last stream
    t.fs_ent.reset(new uint64_t[t.fs_ent_n]);                 // uninitialised: every stream slot is written by the emission below
    for (size_t k = (size_t)nbatch * NB; k < t.fs_ent_n; ++k) t.fs_ent[k] = Tables::fs_pack(zero_off, 0.f);
    t.fs_rseg_ptr.assign(nrows + 1, 0);
    for (int64_t r = 0; r < nrows; ++r) t.fs_rseg_ptr[r + 1] = t.fs_rseg_ptr[r] + (uint32_t)rsegs[r].size();
    t.fs_rseg_idx.assign(nseg ? nseg : 1, 0);
    lap("layout + allocation");
    // ---- 7. emission (threads over items)
    std::vector<uint64_t> real_of(nitems, 0);
    fs_parallel(nitems, hw, [&](size_t o) {
        FsWork &wk = work[order[o]];
        const Tables::FsItem &itx = t.fs_item[o];
        std::memcpy(t.fs_cnt.data() + (size_t)itx.cnt0 * 16, wk.nb.data(), wk.nb.size());
        // segments of every accumulator slot in the order of their stay
        std::vector<std::vector<uint32_t>> of_slot((size_t)GROUPS * 16);
        for (uint32_t x = 0; x < wk.segs.size(); ++x) of_slot[(size_t)(wk.segs[x].wave * 4 + wk.segs[x].sub) * 16 + wk.segs[x].k].push_back(x);
        for (auto &v : of_slot) std::sort(v.begin(), v.end(), [&](uint32_t a, uint32_t b) { return wk.segs[a].t0 < wk.segs[b].t0; });
        for (int g = 0; g < GROUPS; ++g) {
            const int wave = g >> 2;
            size_t oe = (size_t)t.fs_gstart[itx.g0 + g] * NB;          // next entry slot of this stream
            uint32_t id = t.fs_gseg0[itx.g0 + g];
            size_t cur[16] = {0};                            // per slot: the segment whose stay is current / next
            std::vector<uint32_t> pos(wk.segs.size(), 0);    // per segment: entries consumed (only this group's are touched)
            for (uint32_t tt = 0; tt < wk.ntiles; ++tt) {
                const uint32_t tile = wk.tile0 + tt;
                for (int k = 0; k < 16; ++k) {
                    const uint32_t h = wk.nb[((size_t)tt * WAVES + wave) * 16 + k];   // half-batches of this slot in this tile
                    auto &lst = of_slot[(size_t)g * 16 + k];
                    while (cur[k] < lst.size() && wk.segs[lst[cur[k]]].t1 < tile) ++cur[k];
                    if (h == 0) continue;
                    uint32_t c = 0, base = 0, si = 0xFFFFFFFFu;
                    if (cur[k] < lst.size() && wk.segs[lst[cur[k]]].t0 <= tile) {
                        si = lst[cur[k]];
                        const FsSeg &sg = wk.segs[si];
                        base = sg.off + pos[si];
                        while (pos[si] + c < sg.cnt && (uint32_t)eu[base + c] / H == tile) ++c;
                        pos[si] += c;
                    }
                    const bool ends = si != 0xFFFFFFFFu && wk.segs[si].t1 == tile;
                    // h >> 1 whole batches, then (h odd) a half batch: its 4 entries stored twice, so that the rotated reads of
                    // the 8 lanes that share a batch (lane l takes entry l, l+1, l+2, l+3 mod 8) meet all four in every lane
                    const uint32_t nfull = h >> 1, units = (h + 1) >> 1;
                    const uint32_t last_unit = c ? std::min((c - 1) / NB, units - 1) : 0;
                    for (uint32_t un = 0; un < units; ++un) {
                        const uint32_t flag = (ends && un == last_unit) ? 0x80000000u : 0u;
                        for (uint32_t j = 0; j < (uint32_t)NB; ++j, ++oe) {
                            const uint32_t e = un < nfull ? un * NB + j : nfull * NB + j % (NB / 2);
                            if (e < c) {
                                const uint32_t lu = (uint32_t)eu[base + e] % H, lv = elv[base + e];
                                t.fs_ent[oe] = Tables::fs_pack(((lu * W + lv) * (uint32_t)pixel_bytes) | flag, ew[base + e]);
                            } else t.fs_ent[oe] = Tables::fs_pack(zero_off | flag, 0.f);
                        }
                    }
                    real_of[o] += c;
                    if (ends) { const FsSeg &sg = wk.segs[si]; t.fs_rseg_idx[t.fs_rseg_ptr[sg.row] + sg.q] = id++; }
                }
            }
        }
    });
    lap("emission");
    t.fs_npass = npass;
    t.fs_orient.assign(npass, 0);
    for (int ps = 0; ps < npass; ++ps) t.fs_orient[ps] = passes[ps].orient;
    t.fs_shift = shift;
    t.fs_real_entries = 0;
    for (uint64_t v : real_of) t.fs_real_entries += v;
    t.fs_slots = 0;
    for (auto &wk : work) for (int w = 0; w < WAVES; ++w) t.fs_slots += (uint64_t)wk.wave_halves[w] * (NB / 2) * 4;   // entry slots the kernel walks
    t.fs_staged_pixels = 0;
    for (auto &itx : t.fs_item) t.fs_staged_pixels += (uint64_t)itx.ntiles * W * H;
    if ((int64_t)t.fs_real_entries != nnz) { why = "strip streams do not cover the matrix"; return false; }
    t.fs_ok = true;
    return true;
}

// Entry lists of k_fp_list (see sysmat.h): the strips of build_fp_strips with one ANGLE per wave and one accumulator per ray.
bool build_fp_lists(const Coo &m, int N, int P, Tables &t, std::string &why)
{
    constexpr int W = Tables::FL_W, H = Tables::FL_H, TH = Tables::FL_TH, WAVES = Tables::FL_WAVES, ACC = Tables::FL_ACC, BATCH = Tables::FL_BATCH,
                  PIXB = Tables::FL_PIXB, REGS = Tables::FL_REGS;
    t.fl_ok = false;
    const int64_t nrows = (int64_t)N * P;
    const int64_t nnz = m.ptr[nrows];
    if (N < 1 || P < 1 || nnz <= 0 || N > 32768) { why = "empty geometry"; return false; }
    const unsigned hw = std::min(builder_threads(), 32u);
    // ---- 1. direction of every angle (as build_fp_strips)
    std::vector<int> orient(P, 0);
    std::vector<double> slope(P, 0.0);
    for (int i = 0; i < P; ++i) {
        int64_t best = (int64_t)i * N;
        for (int j = 0; j < N; ++j) { int64_t r = (int64_t)i * N + j; if (m.ptr[r + 1] - m.ptr[r] > m.ptr[best + 1] - m.ptr[best]) best = r; }
        const int64_t b = m.ptr[best], e = m.ptr[best + 1];
        if (e - b < 2) continue;
        int ymin = N, ymax = -1, zmin = N, zmax = -1;
        for (int64_t k = b; k < e; ++k) { int y = (int)(m.col[k] / (uint32_t)N), z = (int)(m.col[k] % (uint32_t)N); ymin = std::min(ymin, y); ymax = std::max(ymax, y); zmin = std::min(zmin, z); zmax = std::max(zmax, z); }
        orient[i] = (ymax - ymin >= zmax - zmin) ? 0 : 1;
        double su = 0, sv = 0, suu = 0, suv = 0; const double n = (double)(e - b);
        for (int64_t k = b; k < e; ++k) {
            int y = (int)(m.col[k] / (uint32_t)N), z = (int)(m.col[k] % (uint32_t)N);
            double u = orient[i] ? z : y, v = orient[i] ? y : z;
            su += u; sv += v; suu += u * u; suv += u * v;
        }
        const double den = n * suu - su * su;
        slope[i] = den > 0 ? (n * suv - su * sv) / den : 0.0;
    }
    // ---- 2. passes: angles of one orientation, neighbouring slopes (as few as the slope range allows, the angles dealt evenly)
    struct Pass { int orient; double tg; std::vector<int> ang; };
    std::vector<Pass> passes;
    auto form_passes = [&](double dt_max, int amax) {
        passes.clear();
        for (int o = 0; o < 2; ++o) {
            std::vector<int> a;
            for (int i = 0; i < P; ++i) if (orient[i] == o) a.push_back(i);
            std::sort(a.begin(), a.end(), [&](int x, int y) { return slope[x] != slope[y] ? slope[x] < slope[y] : x < y; });
            if (a.empty()) continue;
            size_t np = (a.size() + amax - 1) / amax;
            for (;; ++np) {
                bool fits = true;
                for (size_t c = 0; c < np && fits; ++c) {
                    const size_t b = a.size() * c / np, e = a.size() * (c + 1) / np;
                    fits = e > b && (e - b) <= (size_t)amax && slope[a[e - 1]] - slope[a[b]] <= dt_max;
                }
                if (fits || np >= a.size()) break;
            }
            for (size_t c = 0; c < np; ++c) {
                const size_t b = a.size() * c / np, e = a.size() * (c + 1) / np;
                if (e == b) continue;
                Pass ps; ps.orient = o; ps.tg = 0.5 * (slope[a[b]] + slope[a[e - 1]]);
                ps.ang.assign(a.begin() + b, a.begin() + e);
                passes.push_back(std::move(ps));
            }
        }
    };
    double dt_max = 0.72; int amax = 20;
    if (const char *sdt = std::getenv("TOMO_FL_DT")) { double v = std::atof(sdt); if (v > 0) dt_max = v; }
    if (const char *sa = std::getenv("TOMO_FL_AMAX")) { int v = std::atoi(sa); if (v > 0) amax = v; }
    std::vector<uint16_t> eu(nnz);
    std::vector<uint8_t> elv(nnz);
    std::vector<float> ew(nnz);
    std::vector<std::vector<FsRowSeg>> rsegs;
    std::vector<int32_t> shift;
    std::vector<int> pass_of(P, 0);
    const int OFF = N;
    int nsegs = 1, seglen = 1 << 30;
    struct Item { int32_t pass, strip; uint32_t tile0 = 0, ntiles = 0; std::vector<uint32_t> rows, qs; std::vector<uint8_t> wv, sl; uint64_t work = 0; };
    std::vector<Item> items;
    for (int attempt = 0; attempt < 6; ++attempt) {
        form_passes(dt_max, amax);
        const int npass = (int)passes.size();
        shift.assign((size_t)npass * N, 0);
        for (int ps = 0; ps < npass; ++ps) {
            for (int a : passes[ps].ang) pass_of[a] = ps;
            for (int u = 0; u < N; ++u) {
                double sh = passes[ps].tg * (u - 0.5 * (N - 1));
                sh = std::max(-(double)N, std::min((double)N, sh));
                shift[(size_t)ps * N + u] = (int32_t)std::floor(sh + 0.5);
            }
        }
        {   // march segments by the image size only (see build_fp_strips): half a strip, 4 ... 16 tiles of H steps
            const int tiles_per_strip = (N + H - 1) / H;
            int len = std::max(4, std::min(16, tiles_per_strip / 2));
            len = std::max(1, std::min(len, tiles_per_strip));
            nsegs = (tiles_per_strip + len - 1) / len;
            seglen = (tiles_per_strip + nsegs - 1) / nsegs;
        }
        // ---- 3. every ray: entries sorted by (strip segment, march coordinate, cross coordinate), cut into stays
        rsegs.assign(nrows, {});
        fs_parallel((size_t)nrows, hw, [&](size_t r) {
            const int i = (int)(r / N), ps = pass_of[i], o = passes[ps].orient;
            const int32_t *sh = shift.data() + (size_t)ps * N;
            const int64_t b = m.ptr[r], e = m.ptr[r + 1];
            const int n = (int)(e - b);
            if (n == 0) return;
            struct K { int32_t strip; uint16_t u; uint8_t lv; float w; };
            std::vector<K> key(n);
            for (int k = 0; k < n; ++k) {
                uint32_t p = m.col[b + k];
                int y = (int)(p / (uint32_t)N), z = (int)(p % (uint32_t)N);
                int u = o ? z : y, v = o ? y : z;
                int vs = v - sh[u] + OFF;
                key[k] = {(vs / W) * nsegs + (u / H) / seglen, (uint16_t)u, (uint8_t)(vs % W), m.val[b + k]};
            }
            std::sort(key.begin(), key.end(), [](const K &a, const K &c) { return a.strip != c.strip ? a.strip < c.strip : a.u != c.u ? a.u < c.u : a.lv < c.lv; });
            auto &rs = rsegs[r];
            for (int k = 0; k < n; ++k) {
                eu[b + k] = key[k].u; elv[b + k] = key[k].lv; ew[b + k] = key[k].w;
                // t0 / t1 in tiles of TH march steps
                if (rs.empty() || rs.back().strip != key[k].strip) rs.push_back({key[k].strip, (uint32_t)(b + k), 0u, (uint32_t)key[k].u / TH, 0u});
                rs.back().cnt++; rs.back().t1 = (uint32_t)key[k].u / TH;
            }
        });
        // ---- 4. items = non-empty (pass, strip segment) buckets
        const int nstrip_max = ((3 * N + W - 1) / W + 2) * nsegs;
        std::vector<uint32_t> iptr((size_t)npass * nstrip_max + 1, 0);
        for (int64_t r = 0; r < nrows; ++r) { const int ps = pass_of[r / N]; for (auto &sg : rsegs[r]) iptr[(size_t)ps * nstrip_max + sg.strip + 1]++; }
        for (size_t k = 0; k + 1 < iptr.size(); ++k) iptr[k + 1] += iptr[k];
        struct Ref { uint32_t row, q; };
        std::vector<Ref> refs(iptr.back() ? iptr.back() : 1);
        {
            std::vector<uint32_t> fill(iptr.begin(), iptr.end() - 1);
            for (int64_t r = 0; r < nrows; ++r) {
                const int ps = pass_of[r / N];
                for (uint32_t q = 0; q < rsegs[r].size(); ++q) refs[fill[(size_t)ps * nstrip_max + rsegs[r][q].strip]++] = {(uint32_t)r, q};
            }
        }
        items.clear();
        for (size_t k = 0; k + 1 < iptr.size(); ++k) {
            if (iptr[k + 1] == iptr[k]) continue;
            Item itx; itx.pass = (int32_t)(k / nstrip_max); itx.strip = (int32_t)(k % nstrip_max);
            uint32_t lo = 0xFFFFFFFFu, hi = 0;
            for (uint32_t x = iptr[k]; x < iptr[k + 1]; ++x) {
                const FsRowSeg &rs = rsegs[refs[x].row][refs[x].q];
                itx.rows.push_back(refs[x].row); itx.qs.push_back(refs[x].q);
                lo = std::min(lo, rs.t0); hi = std::max(hi, rs.t1); itx.work += rs.cnt;
            }
            itx.tile0 = lo; itx.ntiles = hi - lo + 1;
            items.push_back(std::move(itx));
        }
        // ---- 5. every stay of a ray in the item gets an accumulator (wave, slot) for its tiles t0 .. t1: in the order of arrival to
        // the wave with a free slot whose live stays bring the fewest entries per tile -- the waves meet at a barrier after every tile,
        // so a tile costs its busiest wave
        std::vector<uint8_t> over(items.size(), 0);
        fs_parallel(items.size(), hw, [&](size_t it) {
            Item &itx = items[it];
            if (itx.ntiles > 64) { over[it] = 1; return; }             // (the kernel keeps an item's list bounds one tile per lane)
            const size_t ns = itx.rows.size();
            itx.wv.assign(ns, 0); itx.sl.assign(ns, 0);
            std::vector<uint32_t> ord(ns);
            for (uint32_t x = 0; x < ns; ++x) ord[x] = x;
            std::sort(ord.begin(), ord.end(), [&](uint32_t a, uint32_t b) {
                const FsRowSeg &A = rsegs[itx.rows[a]][itx.qs[a]], &B = rsegs[itx.rows[b]][itx.qs[b]];
                return A.t0 != B.t0 ? A.t0 < B.t0 : itx.rows[a] < itx.rows[b];
            });
            std::vector<uint32_t> until((size_t)WAVES * ACC, 0);       // tile (relative) + 1 up to which the slot is taken
            std::vector<float> rate((size_t)WAVES * ACC, 0.f);         // entries per tile of the stay in the slot
            for (uint32_t x : ord) {
                const FsRowSeg &rs = rsegs[itx.rows[x]][itx.qs[x]];
                const uint32_t t0 = rs.t0 - itx.tile0, t1 = rs.t1 - itx.tile0;
                int bw = -1, bs = -1; float br = 0.f;
                for (int w = 0; w < WAVES; ++w) {
                    float r = 0.f; int fs = -1;
                    for (int k = 0; k < ACC; ++k) { if (until[(size_t)w * ACC + k] > t0) r += rate[(size_t)w * ACC + k]; else if (fs < 0) fs = k; }
                    if (fs >= 0 && (bw < 0 || r < br)) { bw = w; bs = fs; br = r; }
                }
                if (bw < 0) { over[it] = 1; return; }
                itx.wv[x] = (uint8_t)bw; itx.sl[x] = (uint8_t)bs;
                until[(size_t)bw * ACC + bs] = t1 + 1; rate[(size_t)bw * ACC + bs] = (float)rs.cnt / (float)(t1 - t0 + 1);
            }
        });
        bool clash = false;
        for (uint8_t o : over) clash |= o != 0;
        if (!clash) break;
        if (attempt == 5) { why = "a strip holds more live rays than a workgroup has accumulators"; return false; }
        dt_max *= 0.7; amax = std::max(1, amax * 3 / 4);
    }
    const int npass = (int)passes.size();
    // ---- 6. layout: items heaviest first; list and flush-list bounds
    std::sort(items.begin(), items.end(), [](const Item &a, const Item &b) { return a.work != b.work ? a.work > b.work : (a.pass != b.pass ? a.pass < b.pass : a.strip < b.strip); });
    const size_t nitems = items.size();
    t.fl_item.assign(nitems, {});
    size_t nlist = 0;
    for (size_t o = 0; o < nitems; ++o) {
        Tables::FlItem &f = t.fl_item[o];
        f.pass = items[o].pass; f.v0 = (items[o].strip / nsegs) * W - OFF; f.tile0 = items[o].tile0; f.ntiles = items[o].ntiles;
        f.lp0 = (uint32_t)nlist; f.work = (uint32_t)std::min<uint64_t>(items[o].work, 0xFFFFFFFFu); f.pad0 = f.pad1 = 0;
        nlist += (size_t)items[o].ntiles * WAVES;
    }
    if (nlist >= (1ull << 31)) { why = "list tables exceed 32-bit offsets"; return false; }
    // per list: entries and flushes (counting pass), then prefix sums, then emission
    std::vector<uint32_t> nbat(nlist, 0), nfl(nlist, 0);
    struct E { uint16_t lu; uint8_t lv, slot; float w; };
    auto walk = [&](size_t o, bool emit) {
        const Item &itx = items[o];
        const Tables::FlItem &f = t.fl_item[o];
        // entries of every (tile, wave), in the order (march step, cross coordinate, ray): a ray's own entries keep their order
        std::vector<std::vector<E>> ents(emit ? (size_t)itx.ntiles * WAVES : 0);
        std::vector<uint32_t> cnt((size_t)itx.ntiles * WAVES, 0);
        for (uint32_t x = 0; x < itx.rows.size(); ++x) {
            const uint32_t row = itx.rows[x];
            const FsRowSeg &rs = rsegs[row][itx.qs[x]];
            const int w = itx.wv[x];
            const uint8_t slot = itx.sl[x];
            for (uint32_t k = rs.off; k < rs.off + rs.cnt; ++k) {
                const size_t li = (size_t)((uint32_t)eu[k] / TH - itx.tile0) * WAVES + w;
                ++cnt[li];
                if (emit) ents[li].push_back({(uint16_t)((uint32_t)eu[k] % TH), elv[k], slot, ew[k]});
            }
            const size_t lf = (size_t)(rs.t1 - itx.tile0) * WAVES + w;
            if (!emit) ++nfl[f.lp0 + lf];
        }
        if (!emit) { for (size_t li = 0; li < cnt.size(); ++li) nbat[f.lp0 + li] = (cnt[li] + BATCH - 1) / BATCH; return; }
        for (size_t li = 0; li < ents.size(); ++li) {
            auto &v = ents[li];
            std::stable_sort(v.begin(), v.end(), [](const E &a, const E &b) { return a.lu != b.lu ? a.lu < b.lu : a.lv < b.lv; });
            const uint32_t buf = (uint32_t)((li / WAVES) & 1) * (uint32_t)(W * TH);
            uint64_t *out = t.fl_ent.get() + (size_t)t.fl_ptr[f.lp0 + li] * BATCH;
            uint32_t last = 0;
            for (size_t k = 0; k < v.size(); ++k) {
                last = (buf + (uint32_t)v[k].lu * W + v[k].lv) * (uint32_t)PIXB;
                out[k] = Tables::fs_pack(last | (uint32_t)(v[k].slot * REGS), v[k].w);
            }
            const size_t padded = (size_t)nbat[f.lp0 + li] * BATCH;
            for (size_t k = v.size(); k < padded; ++k) out[k] = Tables::fs_pack(last, 0.f);     // weight 0 on a staged pixel, into accumulator 0
        }
        // flush records, partial-sum ids in the order (tile, wave, ray)
        std::vector<uint32_t> ord(itx.rows.size());
        for (uint32_t x = 0; x < ord.size(); ++x) ord[x] = x;
        std::sort(ord.begin(), ord.end(), [&](uint32_t a, uint32_t b) {
            const FsRowSeg &A = rsegs[itx.rows[a]][itx.qs[a]], &B = rsegs[itx.rows[b]][itx.qs[b]];
            const int wa = itx.wv[a], wb = itx.wv[b];
            return A.t1 != B.t1 ? A.t1 < B.t1 : wa != wb ? wa < wb : itx.rows[a] < itx.rows[b];
        });
        std::vector<uint32_t> fill((size_t)itx.ntiles * WAVES, 0);
        for (uint32_t x : ord) {
            const uint32_t row = itx.rows[x];
            const FsRowSeg &rs = rsegs[row][itx.qs[x]];
            const size_t lf = (size_t)(rs.t1 - itx.tile0) * WAVES + itx.wv[x];
            const uint32_t id = t.fl_fptr[f.lp0 + lf] + fill[lf]++;
            t.fl_flush[id] = (uint64_t)((uint32_t)itx.sl[x] * REGS) | ((uint64_t)id << 32);
            t.fl_rseg_idx[t.fl_rseg_ptr[row] + itx.qs[x]] = id;
        }
    };
    fs_parallel(nitems, hw, [&](size_t o) { walk(o, false); });
    {   // how evenly a tile's work is spread over the waves (they meet at a barrier after every tile): mean / max batches per wave
        uint64_t sum = 0, summax = 0;
        for (size_t li = 0; li < nlist; li += WAVES) {
            uint32_t mx = 0;
            for (int w = 0; w < WAVES; ++w) { sum += nbat[li + w]; mx = std::max(mx, nbat[li + w]); }
            summax += mx;
        }
        t.fl_balance = summax ? (double)sum / ((double)summax * WAVES) : 0.0;
    }
    t.fl_ptr.assign(nlist + 1, 0);
    t.fl_fptr.assign(nlist + 1, 0);
    uint64_t nb = 0, nf = 0;
    for (size_t li = 0; li < nlist; ++li) { t.fl_ptr[li] = (uint32_t)nb; t.fl_fptr[li] = (uint32_t)nf; nb += nbat[li]; nf += nfl[li]; }
    if (nb >= (1ull << 32) / BATCH || nf >= (1ull << 32)) { why = "list tables exceed 32-bit offsets"; return false; }
    t.fl_ptr[nlist] = (uint32_t)nb; t.fl_fptr[nlist] = (uint32_t)nf;
    t.fl_nseg = (uint32_t)nf;
    t.fl_ent_n = (size_t)(nb + 1) * BATCH;
    t.fl_ent.reset(new uint64_t[t.fl_ent_n]);
    for (size_t k = (size_t)nb * BATCH; k < t.fl_ent_n; ++k) t.fl_ent[k] = 0;
    t.fl_flush.assign(nf ? nf : 1, 0);
    t.fl_rseg_ptr.assign(nrows + 1, 0);
    for (int64_t r = 0; r < nrows; ++r) t.fl_rseg_ptr[r + 1] = t.fl_rseg_ptr[r] + (uint32_t)rsegs[r].size();
    if (t.fl_rseg_ptr[nrows] != nf) { why = "stays and flush records disagree"; return false; }
    t.fl_rseg_idx.assign(nf ? nf : 1, 0);
    fs_parallel(nitems, hw, [&](size_t o) { walk(o, true); });
    t.fl_npass = npass;
    t.fl_orient.assign(npass, 0);
    for (int ps = 0; ps < npass; ++ps) t.fl_orient[ps] = passes[ps].orient;
    t.fl_shift = shift;
    t.fl_real_entries = (uint64_t)nnz;
    t.fl_slots = nb * BATCH;
    t.fl_staged_pixels = 0;
    for (auto &f : t.fl_item) t.fl_staged_pixels += (uint64_t)f.ntiles * W * TH;
    t.fl_ok = true;
    return true;
}

}  // namespace tomo
