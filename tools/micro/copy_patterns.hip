// Micro-benchmark: in-place read-modify-write of a [N*N pixels][SX slices] float volume with the access patterns
// of the projector kernels.  hipcc -O3 --offload-arch=gfx950 copy_patterns.hip -o copy_patterns
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float V __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// A: one wave = 4 consecutive pixels x 256 slices (1 KiB contiguous per instruction), like k_bp_angle
__global__ __launch_bounds__(256) void k_linear(float *x, int npix, int sx)
{
    int gw = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    int nchunk = sx / 256, ngroups = npix / 4;
    int chunk = gw / ngroups, grp = gw - chunk * ngroups;
    if (chunk >= nchunk) return;
    V v[4];
    for (int q = 0; q < 4; ++q) v[q] = *(const V *)(x + (size_t)(grp * 4 + q) * sx + chunk * 256 + lane * 4);
    for (int q = 0; q < 4; ++q) *(V *)(x + (size_t)(grp * 4 + q) * sx + chunk * 256 + lane * 4) = v[q] + 1.0f;
}

// B: workgroup = T x T pixel tile x (LPP*4) slices; LPP lanes per pixel; each thread owns PPT pixels.  XCD-aware: sibling
// chunks of a tile consecutive on one XCD.  order=1: chunk-major instead
// POL: 0 plain, 1 nt loads + nt stores, 2 nt loads + sc1 nt stores (inline asm), 3 plain loads + sc1 nt stores
template <int T, int LPP, int THREADS, int POL = 0>
__global__ __launch_bounds__(THREADS) void k_tile(float *x, int n, int sx, int tiles_z, int ntiles, int nchunk, int order)
{
    constexpr int GROUPS = THREADS / LPP, PPT = T * T / GROUPS;
    int tile, c;
    if (order == 0) { int xcd = blockIdx.x & 7, l = blockIdx.x >> 3; tile = (l / nchunk) * 8 + xcd; c = l % nchunk; }
    else { c = blockIdx.x / ntiles; tile = blockIdx.x % ntiles; }
    if (tile >= ntiles) return;
    int ty = tile / tiles_z, tz = tile % tiles_z;
    int t = threadIdx.x, gl = t % LPP, g = t / LPP;
    V v[PPT];
    for (int J = 0; J < PPT; ++J) {
        int lp = g * PPT + J;
        int y = ty * T + lp / T, z = tz * T + lp % T;
        const V *p = (const V *)(x + ((size_t)y * n + z) * sx + c * (LPP * 4) + gl * 4);
        v[J] = (POL == 1 || POL == 2) ? __builtin_nontemporal_load(p) : *p;
    }
    for (int J = 0; J < PPT; ++J) {
        int lp = g * PPT + J;
        int y = ty * T + lp / T, z = tz * T + lp % T;
        V *p = (V *)(x + ((size_t)y * n + z) * sx + c * (LPP * 4) + gl * 4);
        V w = v[J] + 1.0f;
        if (POL == 1) __builtin_nontemporal_store(w, p);
        else if (POL >= 2) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt\n\ts_nop 1" ::"v"(p), "v"(w) : "memory");
        else *p = w;
    }
}

int main(int argc, char **argv)
{
    int n = argc > 1 ? atoi(argv[1]) : 512, sx = argc > 2 ? atoi(argv[2]) : 512;
    size_t elems = (size_t)n * n * sx;
    float *x; CK(hipMalloc(&x, elems * 4)); CK(hipMemset(x, 0, elems * 4));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto timeit = [&](const char *name, auto launch) {
        launch(); CK(hipDeviceSynchronize());
        CK(hipEventRecord(a));
        for (int i = 0; i < 20; ++i) launch();
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 20;
        printf("%-44s %8.1f us  %6.0f GB/s\n", name, ms * 1e3, 2.0 * elems * 4 / ms / 1e6);
    };
    int npix = n * n;
    timeit("A linear 1 KiB/wave-instr", [&] { int waves = npix / 4 * (sx / 256); hipLaunchKernelGGL(k_linear, dim3((waves + 3) / 4), dim3(256), 0, 0, x, npix, sx); });
    { int tz = n / 16, nt = tz * tz, nc = sx / 64;
      timeit("B tile16x16 x 64 sl, 512 thr, nt ld + nt st", [&] { hipLaunchKernelGGL((k_tile<16, 16, 512, 1>), dim3(8 * ((nt + 7) / 8) * nc), dim3(512), 0, 0, x, n, sx, tz, nt, nc, 0); });
      timeit("B tile16x16 x 64 sl, 512 thr, nt ld + sc1 nt st", [&] { hipLaunchKernelGGL((k_tile<16, 16, 512, 2>), dim3(8 * ((nt + 7) / 8) * nc), dim3(512), 0, 0, x, n, sx, tz, nt, nc, 0); });
      timeit("B tile16x16 x 64 sl, 512 thr, plain ld + sc1 nt st", [&] { hipLaunchKernelGGL((k_tile<16, 16, 512, 3>), dim3(8 * ((nt + 7) / 8) * nc), dim3(512), 0, 0, x, n, sx, tz, nt, nc, 0); }); }
    for (int order = 0; order < 2; ++order) {
        char nm[96];
        { int tz = n / 16, nt = tz * tz, nc = sx / 64; snprintf(nm, 96, "B tile16x16 x 64 sl (256 B), 512 thr, order %d", order);
          timeit(nm, [&] { hipLaunchKernelGGL((k_tile<16, 16, 512>), dim3(8 * ((nt + 7) / 8) * nc), dim3(512), 0, 0, x, n, sx, tz, nt, nc, order); }); }
        { int tz = n / 16, nt = tz * tz, nc = sx / 128; snprintf(nm, 96, "C tile16x16 x 128 sl (512 B), 1024 thr, order %d", order);
          timeit(nm, [&] { hipLaunchKernelGGL((k_tile<16, 32, 1024>), dim3(8 * ((nt + 7) / 8) * nc), dim3(1024), 0, 0, x, n, sx, tz, nt, nc, order); }); }
        { int tz = n / 16, nt = tz * tz, nc = sx / 256; snprintf(nm, 96, "D tile16x16 x 256 sl (1 KiB), 1024 thr, order %d", order);
          timeit(nm, [&] { hipLaunchKernelGGL((k_tile<16, 64, 1024>), dim3(8 * ((nt + 7) / 8) * nc), dim3(1024), 0, 0, x, n, sx, tz, nt, nc, order); }); }
        { int tz = n / 8, nt = tz * tz, nc = sx / 256; snprintf(nm, 96, "E tile8x8 x 256 sl (1 KiB), 512 thr, order %d", order);
          timeit(nm, [&] { hipLaunchKernelGGL((k_tile<8, 64, 512>), dim3(8 * ((nt + 7) / 8) * nc), dim3(512), 0, 0, x, n, sx, tz, nt, nc, order); }); }
        { int tz = n / 16, nt = tz * tz, nc = sx / 64; snprintf(nm, 96, "F tile16x16 x 64 sl (256 B), 256 thr, order %d", order);
          timeit(nm, [&] { hipLaunchKernelGGL((k_tile<16, 16, 256>), dim3(8 * ((nt + 7) / 8) * nc), dim3(256), 0, 0, x, n, sx, tz, nt, nc, order); }); }
    }
    return 0;
}
