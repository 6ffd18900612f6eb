// Micro-benchmark: the per-angle exchange of a volume-resident SART sweep, without the projector arithmetic.
// 256 workgroups (one per CU) each "own" a block of the image; per angle every workgroup publishes one partial row
// (64 slices = 256 B) for each of 96 rays, a ray's 48 partials are summed by a fixed reducer workgroup (2 rays each),
// the reducer publishes the residual row, and every workgroup then reads back the 96 residual rows of its window.
// No grid barrier: per-ray arrival counters and per-ray ready flags (all agent scope, payload stored and loaded sc1).
// Mode 1 replaces the counters / flags by two flat grid barriers per angle.
// hipcc -O3 --offload-arch=gfx950 resident_sync.hip -o resident_sync
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float V __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int NWG = 256, NI = 48, NRAY = 2 * NWG, NCONTRIB = NI;   // rays 2m+b get a partial from workgroups (m - i) mod 256, i < 48
constexpr int SPIN_MAX = 1 << 22;

__device__ __forceinline__ void st_sc1(float *p, V v) { asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ V ld_sc1(const float *base, uint32_t byte_off)
{
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base), 0, 0x7fffffff, 0x00020000);
    return __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 16);
}
__device__ __forceinline__ int ld_flag(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// flat monotonic-counter barrier (mode 1)
__device__ __forceinline__ bool grid_barrier(int *ctr, int target, int *err)
{
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while (ld_flag(ctr) < target) { __builtin_amdgcn_s_sleep(1); if (++spins > SPIN_MAX) { *err = 1; ok = false; break; } }
    }
    __syncthreads();
    return ok;
}

template <int THREADS, int MODE>
__global__ __launch_bounds__(THREADS) void k_exchange(float *partial, float *resid, int *cnt, int *rdy, int *bar, int *err,
                                                       float *check, int nangles, long long *stamps)
{
    long long tacc[6] = {0, 0, 0, 0, 0, 0}, tprev = 0;
#define STAMP(q) { long long now_ = wall_clock64(); tacc[q] += now_ - tprev; tprev = now_; }
    tprev = wall_clock64();
    const int w = blockIdx.x, t = threadIdx.x, gl = t & 15, g = t >> 4, wave = t >> 6, lane = t & 63;
    constexpr int NG = THREADS / 16;
    __shared__ V win[2 * NI * 16];
    __shared__ int s_ok;
    float acc_check = 0.f;
    for (int k = 0; k < nangles; ++k) {
        // ---- publish: group g stores partial rows (i, b) = slots g, g + NG, ...
        for (int s = g; s < 2 * NI; s += NG) {
            int i = s >> 1, b = s & 1;
            V v;
            v[0] = v[1] = v[2] = v[3] = (float)(k + 1) * 0.001f + (float)(w * 2 * NI + s) * 1e-6f;
            st_sc1(partial + ((size_t)(w * NI + i) * 2 + b) * 64 + gl * 4, v);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        STAMP(0)
        if (MODE == 2) {
            if (t == 0) __hip_atomic_store(cnt + (size_t)k * NRAY + w, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (MODE == 0) {
            if (wave == 0) {   // one wave signals for the whole workgroup: 96 counters
                for (int s = lane; s < 2 * NI; s += 64) {
                    int i = s >> 1, b = s & 1, m = (w + i) & (NWG - 1);
                    __hip_atomic_fetch_add(cnt + (size_t)k * NRAY + 2 * m + b, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        } else {
            if (!grid_barrier(bar, (2 * k + 1) * NWG, err)) return;
        }
        // ---- reduce: waves 0 and 1 own rays 2w and 2w + 1
        if (wave < 2) {
            int ray = 2 * w + wave;
            bool ok = true;
            if (MODE == 2) {   // the 48 contributing workgroups' publish flags
                int spins = 0;
                for (;;) {
                    int f = lane < NI ? ld_flag(cnt + (size_t)k * NRAY + ((w - lane) & (NWG - 1))) : 1;
                    if (__all(f)) break;
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > SPIN_MAX) { ok = false; break; }
                }
            }
            if (MODE == 0) {
                int spins = 0;
                while (ld_flag(cnt + (size_t)k * NRAY + ray) < NCONTRIB) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > SPIN_MAX) { ok = false; break; }
                }
            }
            if (wave == 0) STAMP(1)
            if (!ok) { if (lane == 0) *err = 2; }
            else {
                // contributions come from workgroups (w - i) mod 256; a wave instruction covers 4 rows (16 lanes x 16 B each)
                V acc = {0.f, 0.f, 0.f, 0.f};
                {
                    V tmp[NI / 4];
#pragma unroll
                    for (int u = 0; u < NI / 4; ++u) {
                        int i = u * 4 + (lane >> 4);
                        int src = (w - i) & (NWG - 1);
                        tmp[u] = ld_sc1(partial, (uint32_t)((((size_t)(src * NI + i) * 2 + wave) * 64 + (lane & 15) * 4) * 4));
                    }
#pragma unroll
                    for (int u = 0; u < NI / 4; ++u) acc += tmp[u];
                }
                // combine the four 16-lane quarters
                for (int c = 0; c < 4; ++c) {
                    acc[c] += __shfl_xor(acc[c], 16, 64);
                    acc[c] += __shfl_xor(acc[c], 32, 64);
                }
                if (lane < 16) st_sc1(resid + ((size_t)k * NRAY + ray) * 64 + lane * 4, acc);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (MODE != 1 && lane == 0) __hip_atomic_store(rdy + (size_t)k * NRAY + ray, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (wave == 0) STAMP(2)
        }
        // ---- consume: wait for the 96 rays of the window, then stage their rows
        if (MODE != 1) {
            if (wave == 2) {
                int spins = 0;
                bool ok = true;
                for (;;) {
                    int m0 = (w + (lane >> 1)) & (NWG - 1), m1 = (w + 32 + (lane >> 1)) & (NWG - 1);
                    int f0 = ld_flag(rdy + (size_t)k * NRAY + 2 * m0 + (lane & 1));
                    int f1 = lane < 32 ? ld_flag(rdy + (size_t)k * NRAY + 2 * m1 + (lane & 1)) : 1;
                    if (__all(f0 && f1)) break;
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > SPIN_MAX) { ok = false; break; }
                }
                if (lane == 0) { s_ok = ok; if (!ok) *err = 3; }
            }
            __syncthreads();
            if (!s_ok) return;
        } else {
            if (!grid_barrier(bar, (2 * k + 2) * NWG, err)) return;
        }
        if (wave == 0) STAMP(3)
        for (int s = g; s < 2 * NI; s += NG) {
            int i = s >> 1, b = s & 1, m = (w + i) & (NWG - 1);
            win[s * 16 + gl] = ld_sc1(resid, (uint32_t)((((size_t)k * NRAY + 2 * m + b) * 64 + gl * 4) * 4));
        }
        __syncthreads();
        for (int s = g; s < 2 * NI; s += NG) acc_check += win[s * 16 + gl][0];
        __syncthreads();
        if (wave == 0) STAMP(4)
    }
    if (t == 0) for (int q = 0; q < 6; ++q) stamps[w * 6 + q] = tacc[q];
    atomicAdd(check + w, acc_check);
}

template <int THREADS, int MODE>
static void run(int nangles, int reps)
{
    float *partial, *resid, *check;
    int *cnt, *rdy, *bar, *err;
    long long *stamps; CK(hipMalloc(&stamps, NWG * 6 * 8));
    CK(hipMalloc(&partial, (size_t)NWG * NI * 2 * 64 * 4));
    CK(hipMalloc(&resid, (size_t)nangles * NRAY * 64 * 4));
    CK(hipMalloc(&cnt, (size_t)nangles * NRAY * 4));
    CK(hipMalloc(&rdy, (size_t)nangles * NRAY * 4));
    CK(hipMalloc(&bar, 256));
    CK(hipMalloc(&err, 4));
    CK(hipMalloc(&check, NWG * 4));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
        CK(hipMemset(cnt, 0, (size_t)nangles * NRAY * 4));
        CK(hipMemset(rdy, 0, (size_t)nangles * NRAY * 4));
        CK(hipMemset(bar, 0, 256));
        CK(hipMemset(err, 0, 4));
        CK(hipMemset(check, 0, NWG * 4));
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((k_exchange<THREADS, MODE>), dim3(NWG), dim3(THREADS), 0, 0, partial, resid, cnt, rdy, bar, err, check, nangles, stamps);
        CK(hipEventRecord(b));
        CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        best = ms < best ? ms : best;
        int herr; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
        if (herr) { printf("threads %d mode %d: spin limit hit (err %d)\n", THREADS, MODE, herr); break; }
    }
    // expected check: sum over angles and the window's 96 rays of the ray sum (first component)
    std::vector<float> hc(NWG);
    CK(hipMemcpy(hc.data(), check, NWG * 4, hipMemcpyDeviceToHost));
    double worst = 0;
    for (int w = 0; w < NWG; ++w) {
        double exp_sum = 0;
        for (int k = 0; k < nangles; ++k)
            for (int s = 0; s < 2 * NI; ++s) {
                int i = s >> 1, b2 = s & 1, m = (w + i) & (NWG - 1);
                for (int ii = 0; ii < NI; ++ii) {
                    int src = (m - ii) & (NWG - 1);
                    exp_sum += (double)((float)(k + 1) * 0.001f + (float)(src * 2 * NI + ii * 2 + b2) * 1e-6f);
                }
            }
        exp_sum *= (THREADS / 16 >= 2 * NI ? 1 : 1);
        // every 16-lane group adds its rows' first component: all 16 lanes of a group add the same rows -> x16 per row
        double got = hc[w] / 16.0;
        double rel = fabs(got - exp_sum) / exp_sum;
        worst = rel > worst ? rel : worst;
    }
    {
        std::vector<long long> hs(NWG * 6);
        CK(hipMemcpy(hs.data(), stamps, NWG * 6 * 8, hipMemcpyDeviceToHost));
        double av[6] = {0};
        for (int w = 0; w < NWG; ++w) for (int q = 0; q < 6; ++q) av[q] += (double)hs[w * 6 + q] / NWG / nangles * 0.01;   // 100 MHz clock -> us
        printf("   us per angle (wave 0 of every workgroup): publish+drain+barrier %.2f | signal+wait count %.2f | reduce+publish %.2f | wait flags %.2f | stage window %.2f\n", av[0], av[1], av[2], av[3], av[4]);
    }
    printf("threads %4d mode %d: %8.3f ms for %d angles = %6.2f us per angle; check rel err %.2e\n", THREADS, MODE, best, nangles,
           1000.0 * best / nangles, worst);
    CK(hipFree(partial)); CK(hipFree(resid)); CK(hipFree(cnt)); CK(hipFree(rdy)); CK(hipFree(bar)); CK(hipFree(err)); CK(hipFree(check));
}

int main(int argc, char **argv)
{
    int nangles = argc > 1 ? atoi(argv[1]) : 720;
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    printf("%s: %d CUs\n", p.name, p.multiProcessorCount);
    if (p.multiProcessorCount < NWG) { printf("needs %d CUs\n", NWG); return 0; }
    run<1024, 0>(nangles, 3);
    run<1024, 2>(nangles, 3);
    run<512, 2>(nangles, 3);
    run<1024, 1>(nangles, 3);
    return 0;
}
