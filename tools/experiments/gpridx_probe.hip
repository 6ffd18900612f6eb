// Probe: does VGPR index mode (s_set_gpr_idx_on, M0-relative dst/src2) work on gfx950 for v_fma_f32 and v_pk_fma_f32?
// Build: hipcc -O3 --offload-arch=gfx950 gpridx_probe.hip -o gpridx_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <cmath>
#include <cstring>
typedef float v32f __attribute__((ext_vector_type(32)));

template <int PK>
__global__ void probe(const uint2 *__restrict__ ent, int n, const float4 *__restrict__ data, float *__restrict__ out)
{
    v32f acc;
    for (int i = 0; i < 32; ++i) acc[i] = 0.f;
    const int lane = threadIdx.x;
    for (int e = 0; e < n; ++e) {
        const uint2 c = ent[e];
        const float4 d = data[e * 64 + lane];
        if (PK == 0)
            asm volatile("s_set_gpr_idx_on %1, gpr_idx(SRC2,DST)\n"
                         "v_fma_f32 v64, %2, %3, v64\n"
                         "v_fma_f32 v65, %2, %4, v65\n"
                         "v_fma_f32 v66, %2, %5, v66\n"
                         "v_fma_f32 v67, %2, %6, v67\n"
                         "s_set_gpr_idx_off\n"
                         : "+{v[64:95]}"(acc) : "s"(c.x), "s"(c.y), "v"(d.x), "v"(d.y), "v"(d.z), "v"(d.w) );
        else {
            typedef float v2f __attribute__((ext_vector_type(2)));
            v2f lo = {d.x, d.y}, hi = {d.z, d.w};
            uint64_t cc = ((uint64_t)c.y << 32) | c.x;
            asm volatile("s_set_gpr_idx_on %4, gpr_idx(SRC2,DST)\n"
                         "v_pk_fma_f32 v[64:65], %1, %2, v[64:65] op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
                         "v_pk_fma_f32 v[66:67], %1, %3, v[66:67] op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
                         "s_set_gpr_idx_off\n"
                         : "+{v[64:95]}"(acc) : "s"(cc), "v"(lo), "v"(hi), "s"(c.x));
        }
    }
    for (int i = 0; i < 32; ++i) out[i * 64 + lane] = acc[i];
}

int main()
{
    const int n = 200;
    std::vector<uint2> ent(n);
    std::vector<float> data((size_t)n * 256), want(32 * 64, 0.f);
    srand(1);
    for (int e = 0; e < n; ++e) {
        int q = rand() % 8;
        float w = (rand() % 1000) / 1000.f;
        uint32_t wb; std::memcpy(&wb, &w, 4);
        ent[e] = make_uint2((uint32_t)(q * 4) | ((uint32_t)(rand() & 0xFFFF) << 8), wb);   // junk above bit 7 must be ignored
        for (int i = 0; i < 256; ++i) data[(size_t)e * 256 + i] = (rand() % 2000) / 1000.f - 1.f;
        for (int l = 0; l < 64; ++l)
            for (int k = 0; k < 4; ++k) {
                float &a = want[(q * 4 + k) * 64 + l];
                a = fmaf(w, data[(size_t)e * 256 + l * 4 + k], a);
            }
    }
    uint2 *d_ent; float4 *d_data; float *d_out;
    hipMalloc(&d_ent, n * 8); hipMalloc(&d_data, data.size() * 4); hipMalloc(&d_out, 32 * 64 * 4);
    hipMemcpy(d_ent, ent.data(), n * 8, hipMemcpyHostToDevice);
    hipMemcpy(d_data, data.data(), data.size() * 4, hipMemcpyHostToDevice);
    int rc = 0;
    for (int pk = 0; pk < 2; ++pk) {
        hipMemset(d_out, 0xff, 32 * 64 * 4);
        if (pk) hipLaunchKernelGGL(probe<1>, 1, 64, 0, 0, d_ent, n, d_data, d_out);
        else hipLaunchKernelGGL(probe<0>, 1, 64, 0, 0, d_ent, n, d_data, d_out);
        std::vector<float> got(32 * 64);
        hipError_t err = hipMemcpy(got.data(), d_out, got.size() * 4, hipMemcpyDeviceToHost);
        int bad = 0;
        for (size_t i = 0; i < got.size(); ++i) bad += !(got[i] == want[i]);
        printf("pk=%d: %s, %d of %zu values differ (first: got %g want %g)\n", pk, hipGetErrorString(err), bad, got.size(), got[0], want[0]);
        rc |= bad != 0;
    }
    return rc;
}
