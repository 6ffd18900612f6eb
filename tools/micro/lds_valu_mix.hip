// Micro-benchmark: the inner loop of k_fp_tile / k_bp_tile in isolation -- per 16-lane group and batch, 8 ds_read_b128 of 256-byte
// pixel images (address = entry offset shared by DPP row rotation + the lane's 16 bytes) and 8 x 2 packed FMAs with the entry's
// weight (shared by DPP too).  No global memory inside the loop.  Which of {LDS array, vector ALU, their overlap} sets the time?
//   hipcc -O3 --offload-arch=gfx950 lds_valu_mix.hip -o lds_valu_mix ; ./lds_valu_mix
// MODE 0: reads + FMAs, software-pipelined like k_fp_tile (reads of batch i+1 issued before the FMAs of batch i)
// MODE 1: the reads alone (data kept alive by an empty asm)            MODE 2: the vector work alone (DPP + FMAs on registers)
// MODE 3: reads + FMAs, every lane on its OWN entry (no DPP: plain add / plain weight)
// MODE 4: as 0 but not pipelined (reads of a batch, then its FMAs)
// MODE 5: DPP for the addresses only (weights plain)                   MODE 6: DPP for the weights only (addresses plain)
// MODE 7: as 0, the weights of two consecutive entries kept as ONE register pair (the packed FMAs select a half of it)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float V __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int J> __device__ __forceinline__ uint32_t row_ror(uint32_t v)
{
    if (J == 0) return v;
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x120 + J, 0xf, 0xf, true);
}

constexpr int PIX = 512, THREADS = 1024;

template <int MODE>
__global__ __launch_bounds__(THREADS) void k_mix(float *out, int nbatch)
{
    extern __shared__ V tile[];                     // [PIX][16]
    const int t = threadIdx.x, gl = t & 15;
    for (int i = t; i < PIX * 16; i += THREADS) tile[i] = V{(float)(i & 7), 1.f, 2.f, 3.f} * 1e-3f;
    __syncthreads();
    const char *base = reinterpret_cast<const char *>(tile) + gl * 16;
    // the lane's own entry of a batch: a pixel offset and a weight (fixed: the loop measures issue rates, not data)
    uint32_t off = (uint32_t)(((t * 37 + 11) & (PIX - 1)) * 256), wbits = __float_as_uint(1.0f + (t & 7) * 0.125f);
    V acc = {0.f, 0.f, 0.f, 0.f};
    V xa[8], xb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { xa[j] = V{1.f, 2.f, 3.f, 4.f}; xb[j] = V{4.f, 3.f, 2.f, 1.f}; }
#define LD1(XV, J) XV[J] = *reinterpret_cast<const V *>(base + ((MODE == 3 || MODE == 6) ? ((off + (uint32_t)(J) * 4352u) & (uint32_t)(PIX * 256 - 1)) : row_ror<J>(off)));
#define FM1(XV, J) acc += __uint_as_float((MODE == 3 || MODE == 5) ? wbits + (uint32_t)(J) : row_ror<J>(wbits)) * XV[J];
#define FM2(XV, J)                                                                                                   \
    {                                                                                                                \
        v2f w = {__uint_as_float(row_ror<J>(wbits)), __uint_as_float(row_ror<J + 1>(wbits))};                        \
        asm volatile("" : "+v"(w));                                                                                  \
        const v2f w0 = __builtin_shufflevector(w, w, 0, 0), w1 = __builtin_shufflevector(w, w, 1, 1);                \
        alo = __builtin_elementwise_fma(__builtin_shufflevector(XV[J], XV[J], 0, 1), w0, alo);                       \
        ahi = __builtin_elementwise_fma(__builtin_shufflevector(XV[J], XV[J], 2, 3), w0, ahi);                       \
        alo = __builtin_elementwise_fma(__builtin_shufflevector(XV[J + 1], XV[J + 1], 0, 1), w1, alo);               \
        ahi = __builtin_elementwise_fma(__builtin_shufflevector(XV[J + 1], XV[J + 1], 2, 3), w1, ahi);               \
    }
#define CONSUME2(XV) { FM2(XV, 0) FM2(XV, 2) FM2(XV, 4) FM2(XV, 6) }
#define STEP { off = (off + 256u * 9u) & (uint32_t)(PIX * 256 - 1); wbits += 0x100u; }
#define ISSUE(XV) { LD1(XV, 0) LD1(XV, 1) LD1(XV, 2) LD1(XV, 3) LD1(XV, 4) LD1(XV, 5) LD1(XV, 6) LD1(XV, 7) }
#define CONSUME(XV) { FM1(XV, 0) FM1(XV, 1) FM1(XV, 2) FM1(XV, 3) FM1(XV, 4) FM1(XV, 5) FM1(XV, 6) FM1(XV, 7) }
#define KEEP(XV) { _Pragma("unroll") for (int j = 0; j < 8; ++j) asm volatile("" :: "v"(XV[j])); }
    if (MODE == 0 || MODE == 3 || MODE == 5 || MODE == 6) {
        ISSUE(xa) STEP
        for (int b = 0; b < nbatch; b += 2) {
            ISSUE(xb) STEP CONSUME(xa)
            ISSUE(xa) STEP CONSUME(xb)
        }
    } else if (MODE == 7) {
        v2f alo = {0.f, 0.f}, ahi = {0.f, 0.f};
        ISSUE(xa) STEP
        for (int b = 0; b < nbatch; b += 2) {
            ISSUE(xb) STEP CONSUME2(xa)
            ISSUE(xa) STEP CONSUME2(xb)
        }
        acc = V{alo.x, alo.y, ahi.x, ahi.y};
    } else if (MODE == 4) {
        for (int b = 0; b < nbatch; b += 2) {
            ISSUE(xa) STEP CONSUME(xa)
            ISSUE(xb) STEP CONSUME(xb)
        }
    } else if (MODE == 1) {
        for (int b = 0; b < nbatch; b += 2) {
            ISSUE(xa) STEP KEEP(xa)
            ISSUE(xb) STEP KEEP(xb)
        }
    } else {
        for (int b = 0; b < nbatch; b += 2) {
            uint32_t o = 0;
            o += row_ror<1>(off) + row_ror<2>(off) + row_ror<3>(off) + row_ror<4>(off) + row_ror<5>(off) + row_ror<6>(off) + row_ror<7>(off) + off;
            CONSUME(xa)
            off = (off + 256u * 8u + (o & 1u)) & (uint32_t)(PIX * 256 - 1);
            o += row_ror<1>(off) + row_ror<2>(off) + row_ror<3>(off) + row_ror<4>(off) + row_ror<5>(off) + row_ror<6>(off) + row_ror<7>(off) + off;
            CONSUME(xb)
            wbits ^= (o & 1u);
        }
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) out[blockIdx.x * THREADS + t] = acc[0];
}

template <int MODE> static void run(const char *what, int nwg, int nbatch)
{
    float *out;
    CK(hipMalloc(&out, (size_t)nwg * THREADS * 4));
    CK(hipFuncSetAttribute((const void *)k_mix<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, PIX * 256));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL(k_mix<MODE>, dim3(nwg), dim3(THREADS), PIX * 256, 0, out, 64);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(k_mix<MODE>, dim3(nwg), dim3(THREADS), PIX * 256, 0, out, nbatch);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    // one workgroup per CU at a time (128 KB of LDS): nwg / 256 workgroups in a row per CU, 16 waves each
    const double rounds = (double)nbatch * (nwg / 256.0);                 // a round = one batch of every wave of a workgroup
    printf("%-58s %8.3f ms  -> %7.1f ns per round (16 waves x 1 batch; 512 clk = %.0f ns at 2.4 GHz if LDS and vector ALU overlap fully)\n", what, ms,
           ms * 1e6 / rounds, 512 / 2.4);
    CK(hipFree(out));
}

int main(int argc, char **argv)
{
    int nbatch = argc > 1 ? atoi(argv[1]) : 4000, nwg = 256 * (argc > 2 ? atoi(argv[2]) : 4);
    run<0>("reads + FMAs, pipelined (k_fp_tile's loop)", nwg, nbatch);
    run<4>("reads + FMAs, batch after batch", nwg, nbatch);
    run<1>("the 8 ds_read_b128 per batch alone", nwg, nbatch);
    run<2>("the DPP + packed-FMA work alone", nwg, nbatch);
    run<3>("reads + FMAs, no DPP (own entry per lane)", nwg, nbatch);
    run<5>("reads + FMAs, DPP for the addresses only", nwg, nbatch);
    run<6>("reads + FMAs, DPP for the weights only", nwg, nbatch);
    run<7>("k_fp_tile's loop, weights of two entries in one register pair", nwg, nbatch);
    return 0;
}
