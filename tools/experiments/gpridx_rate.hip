// What does an indexed v_pk_fma_f32 cost?  256 x 8 workgroups of 16 waves, LOOPS x 16 FMAs each; variants:
// 0: plain v_pk_fma (fixed registers)   1: s_set_gpr_idx_on before every FMA   2: _on once, s_set_gpr_idx_idx before every FMA
// 3: as 1 plus an s_lshr + v_add_u32 + ds_read_b64 per FMA (the k_bp_list entry, without its scalar loads)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float v32f __attribute__((ext_vector_type(32)));
#define R4(X) X X X X
#define R16(X) R4(X) R4(X) R4(X) R4(X)
template <int VAR>
__global__ __launch_bounds__(1024) void rate(float *out, int loops, uint32_t i0, uint32_t i1, float w)
{
    __shared__ float lds[8192];
    v32f acc;
    for (int i = 0; i < 32; ++i) acc[i] = 0.f;
    for (int i = threadIdx.x; i < 8192; i += 1024) lds[i] = 1.f;
    __syncthreads();
    uint64_t e0 = ((uint64_t)__float_as_uint(w) << 32) | i0, e1 = ((uint64_t)__float_as_uint(w) << 32) | i1;
    uint32_t base = (threadIdx.x & 63) * 8;
    for (int k = 0; k < (VAR >= 4 ? 0 : loops); ++k) {
        if (VAR == 0)
            asm volatile(R16("v_pk_fma_f32 v[64:65], %1, v[32:33], v[64:65] op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
                             "v_pk_fma_f32 v[70:71], %2, v[32:33], v[70:71] op_sel:[1,0,0] op_sel_hi:[1,1,1]\n")
                         : "+{v[64:95]}"(acc) : "s"(e0), "s"(e1) : "v32", "v33");
        else if (VAR == 1)
            asm volatile(R16("s_set_gpr_idx_on %3, gpr_idx(SRC2,DST)\n"
                             "v_pk_fma_f32 v[64:65], %1, v[32:33], v[64:65] op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
                             "s_set_gpr_idx_on %4, gpr_idx(SRC2,DST)\n"
                             "v_pk_fma_f32 v[64:65], %2, v[32:33], v[64:65] op_sel:[1,0,0] op_sel_hi:[1,1,1]\n")
                         "s_set_gpr_idx_off\n"
                         : "+{v[64:95]}"(acc) : "s"(e0), "s"(e1), "s"(i0), "s"(i1) : "v32", "v33");
        else if (VAR == 2)
            asm volatile("s_set_gpr_idx_on %3, gpr_idx(SRC2,DST)\n"
                         R16("s_set_gpr_idx_idx %3\n"
                             "v_pk_fma_f32 v[64:65], %1, v[32:33], v[64:65] op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
                             "s_set_gpr_idx_idx %4\n"
                             "v_pk_fma_f32 v[64:65], %2, v[32:33], v[64:65] op_sel:[1,0,0] op_sel_hi:[1,1,1]\n")
                         "s_set_gpr_idx_off\n"
                         : "+{v[64:95]}"(acc) : "s"(e0), "s"(e1), "s"(i0), "s"(i1) : "v32", "v33");
        else
            asm volatile(R16("s_lshr_b32 s34, %3, 8\n v_add_u32_e32 v32, s34, %5\n ds_read_b64 v[32:33], v32\n"
                             "s_lshr_b32 s35, %4, 8\n v_add_u32_e32 v34, s35, %5\n ds_read_b64 v[34:35], v34\n"
                             "s_waitcnt lgkmcnt(1)\n"
                             "s_set_gpr_idx_on %3, gpr_idx(SRC2,DST)\n"
                             "v_pk_fma_f32 v[64:65], %1, v[32:33], v[64:65] op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
                             "s_waitcnt lgkmcnt(0)\n"
                             "s_set_gpr_idx_on %4, gpr_idx(SRC2,DST)\n"
                             "v_pk_fma_f32 v[64:65], %2, v[34:35], v[64:65] op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
                             "s_set_gpr_idx_off\n")
                         : "+{v[64:95]}"(acc) : "s"(e0), "s"(e1), "s"(i0), "s"(i1), "v"(base) : "v32", "v33", "v34", "v35", "s34", "s35", "memory");
    }
    if (VAR >= 4) {
        uint32_t mask = ~511u;
#define RD(K, E) "v_and_or_b32 v[32+2*" #K "], " E ", %6, %5\n" DSR(K)
#define FM(K, E, I, W) "s_waitcnt lgkmcnt(" #W ")\n s_set_gpr_idx_on " I ", gpr_idx(SRC2,DST)\n v_pk_fma_f32 v[64:65], " E ", v[32+2*" #K ":33+2*" #K "], v[64:65] op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
        for (int k = 0; k < loops; ++k) {
            if (VAR == 4) {
#define DSR(K) "ds_read_b64 v[32+2*" #K ":33+2*" #K "], v[32+2*" #K "]\n"
            asm volatile(RD(0, "%3") RD(1, "%4") RD(2, "%3") RD(3, "%4") RD(4, "%3") RD(5, "%4") RD(6, "%3") RD(7, "%4")
                         RD(8, "%3") RD(9, "%4") RD(10, "%3") RD(11, "%4") RD(12, "%3") RD(13, "%4") RD(14, "%3") RD(15, "%4")
                         FM(0, "%1", "%3", 15) FM(1, "%2", "%4", 14) FM(2, "%1", "%3", 13) FM(3, "%2", "%4", 12) FM(4, "%1", "%3", 11) FM(5, "%2", "%4", 10) FM(6, "%1", "%3", 9) FM(7, "%2", "%4", 8)
                         FM(8, "%1", "%3", 7) FM(9, "%2", "%4", 6) FM(10, "%1", "%3", 5) FM(11, "%2", "%4", 4) FM(12, "%1", "%3", 3) FM(13, "%2", "%4", 2) FM(14, "%1", "%3", 1) FM(15, "%2", "%4", 0)
                         "s_set_gpr_idx_off\n"
                         : "+{v[64:95]}"(acc) : "s"(e0), "s"(e1), "s"(i0), "s"(i1), "v"(base), "v"(mask)
                         : "v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63","memory");
#undef DSR
            } else {
#define DSR(K) ""
            asm volatile(RD(0, "%3") RD(1, "%4") RD(2, "%3") RD(3, "%4") RD(4, "%3") RD(5, "%4") RD(6, "%3") RD(7, "%4")
                         RD(8, "%3") RD(9, "%4") RD(10, "%3") RD(11, "%4") RD(12, "%3") RD(13, "%4") RD(14, "%3") RD(15, "%4")
                         FM(0, "%1", "%3", 15) FM(1, "%2", "%4", 14) FM(2, "%1", "%3", 13) FM(3, "%2", "%4", 12) FM(4, "%1", "%3", 11) FM(5, "%2", "%4", 10) FM(6, "%1", "%3", 9) FM(7, "%2", "%4", 8)
                         FM(8, "%1", "%3", 7) FM(9, "%2", "%4", 6) FM(10, "%1", "%3", 5) FM(11, "%2", "%4", 4) FM(12, "%1", "%3", 3) FM(13, "%2", "%4", 2) FM(14, "%1", "%3", 1) FM(15, "%2", "%4", 0)
                         "s_set_gpr_idx_off\n"
                         : "+{v[64:95]}"(acc) : "s"(e0), "s"(e1), "s"(i0), "s"(i1), "v"(base), "v"(mask)
                         : "v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63","memory");
#undef DSR
            }
        }
    }
    float sum = 0.f;
    for (int i = 0; i < 32; ++i) sum += acc[i];
    out[blockIdx.x * 1024 + threadIdx.x] = sum;
}
int main()
{
    float *out; hipMalloc(&out, 2048 * 1024 * 4);
    const int loops = 2000;
    for (int var = 0; var < 6; ++var) {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(a);
            if (var == 0) hipLaunchKernelGGL(rate<0>, 2048, 1024, 0, 0, out, loops, 0u, 6u, 0.5f);
            if (var == 1) hipLaunchKernelGGL(rate<1>, 2048, 1024, 0, 0, out, loops, 0u, 6u, 0.5f);
            if (var == 2) hipLaunchKernelGGL(rate<2>, 2048, 1024, 0, 0, out, loops, 0u, 6u, 0.5f);
            if (var == 3) hipLaunchKernelGGL(rate<3>, 2048, 1024, 0, 0, out, loops, 0u, 6u, 0.5f);
            if (var == 4) hipLaunchKernelGGL(rate<4>, 2048, 1024, 0, 0, out, loops, 512u * 3, 512u * 7 + 6u, 0.5f);
            if (var == 5) hipLaunchKernelGGL(rate<5>, 2048, 1024, 0, 0, out, loops, 512u * 3, 512u * 7 + 6u, 0.5f);
            hipEventRecord(b); hipEventSynchronize(b);
        }
        float ms; hipEventElapsedTime(&ms, a, b);
        // per SIMD: 8 workgroups x 4 waves x loops x 32 FMAs
        double fmas = 8.0 * 4 * loops * 32;
        float h; hipMemcpy(&h, out, 4, hipMemcpyDeviceToHost);
        printf("variant %d: %.3f ms, %.2f cycles per FMA per SIMD at 2.4 GHz (check %g)\n", var, ms, ms * 1e-3 * 2.4e9 / fmas, h);
    }
    return 0;
}
