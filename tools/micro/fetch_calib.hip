// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access widths k_sart_resident uses.
// /opt/skills/guides/MI355X_MICROARCH.md (HBM): FETCH_SIZE reports exactly half the bytes of a 16-B-per-lane streaming read; "other
// access widths are uncalibrated: calibrate on a known byte count in your own access pattern".  Each kernel here reads (or writes)
// every byte of a 1 GiB buffer exactly once in one of those patterns; run under
//   rocprofv3 --pmc FETCH_SIZE -- ./fetch_calib      and      rocprofv3 --pmc WRITE_SIZE -- ./fetch_calib
// and divide the counter (KiB) by 1 GiB (tools/fetch_calib.sh does both and prints the factors).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef unsigned long long u64;

// 16 B per lane, coalesced (the guide's calibrated case)
__global__ void k_read_vec16(const uint4 *p, size_t n16, uint32_t *sink)
{
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) { uint4 v = p[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) *sink = acc;
}
// 8 B per lane, agent-scope atomic loads (rs_gld: how the granules are polled)
__global__ void k_read_poll8(const u64 *p, size_t n8, uint32_t *sink)
{
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
        u64 v = __hip_atomic_load((const __attribute__((address_space(1))) u64 *)(p + i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        acc ^= (uint32_t)v ^ (uint32_t)(v >> 32);
    }
    if (acc == 0x12345678u) *sink = acc;
}
// 4 B per lane (the chunk's rows on the way in: global_load_dword, 256 B per wave and instruction)
__global__ void k_read_vec4(const uint32_t *p, size_t n4, uint32_t *sink)
{
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) acc ^= p[i];
    if (acc == 0x12345678u) *sink = acc;
}
// scalar loads, 64 B per s_load_dwordx16 (the cells); one wave per workgroup walks its own contiguous piece
__global__ void k_read_scalar(const uint32_t *p, size_t n4, uint32_t *sink)
{
    const size_t per = n4 / gridDim.x;                       // dwords per wave (a multiple of 16)
    const uint32_t *q = p + (size_t)blockIdx.x * per;
    uint32_t acc = 0;
    for (size_t i = 0; i < per; i += 16) {
        uint32_t v0, v1;
        asm volatile("s_load_dwordx16 s[36:51], %2, 0x0\n s_waitcnt lgkmcnt(0)\n s_xor_b32 %0, s36, s43\n s_xor_b32 %1, s44, s51\n"
                     : "=s"(v0), "=s"(v1) : "s"(q + i)
                     : "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "scc");
        acc ^= v0 ^ v1;
    }
    if (acc == 0x12345678u && threadIdx.x == 0) *sink = acc;
}
// one 4-byte LDS-DMA load per 128-byte line (rs_touch: pulls a line into the L2 without a register)
__global__ void k_touch_lines(const uint32_t *p, size_t nlines, uint32_t *sink)
{
    __shared__ float dump[64];
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nlines; i += (size_t)gridDim.x * blockDim.x)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(p + i * 32), (__attribute__((address_space(3))) void *)dump, 4, 0, 0);
    __syncthreads();
    if (dump[threadIdx.x & 63] == 1.2345e-30f) *sink = 1;
}
// 8 B per lane, agent-scope atomic stores (rs_gst: how the granules are published; write-through)
__global__ void k_write_gran8(u64 *p, size_t n8)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x)
        __hip_atomic_store((__attribute__((address_space(1))) u64 *)(p + i), (u64)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// 4 B per lane plain stores (the chunk's rows on the way out) and 16 B per lane (the calibrated case)
__global__ void k_write_vec4(uint32_t *p, size_t n4)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) p[i] = (uint32_t)i;
}
__global__ void k_write_vec16(uint4 *p, size_t n16)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) p[i] = make_uint4((uint32_t)i, 1, 2, 3);
}

int main()
{
    const size_t bytes = (size_t)1 << 30;
    void *buf; uint32_t *sink;
    CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(buf, 1, bytes));
    void *flush; CK(hipMalloc(&flush, bytes));
    auto evict = [&] { CK(hipMemset(flush, 2, bytes)); CK(hipDeviceSynchronize()); };    // 1 GiB of other stores: nothing of buf stays in the 256 MiB Infinity Cache
    const int grid = 256 * 8;
    evict(); hipLaunchKernelGGL(k_read_vec16, dim3(grid), dim3(256), 0, 0, (const uint4 *)buf, bytes / 16, sink); CK(hipDeviceSynchronize());
    evict(); hipLaunchKernelGGL(k_read_poll8, dim3(grid), dim3(256), 0, 0, (const u64 *)buf, bytes / 8, sink); CK(hipDeviceSynchronize());
    evict(); hipLaunchKernelGGL(k_read_vec4, dim3(grid), dim3(256), 0, 0, (const uint32_t *)buf, bytes / 4, sink); CK(hipDeviceSynchronize());
    evict(); hipLaunchKernelGGL(k_read_scalar, dim3(4096), dim3(64), 0, 0, (const uint32_t *)buf, bytes / 4, sink); CK(hipDeviceSynchronize());
    evict(); hipLaunchKernelGGL(k_touch_lines, dim3(grid), dim3(256), 0, 0, (const uint32_t *)buf, bytes / 128, sink); CK(hipDeviceSynchronize());
    evict(); hipLaunchKernelGGL(k_write_gran8, dim3(grid), dim3(256), 0, 0, (u64 *)buf, bytes / 8); CK(hipDeviceSynchronize());
    evict(); hipLaunchKernelGGL(k_write_vec4, dim3(grid), dim3(256), 0, 0, (uint32_t *)buf, bytes / 4); CK(hipDeviceSynchronize());
    evict(); hipLaunchKernelGGL(k_write_vec16, dim3(grid), dim3(256), 0, 0, (uint4 *)buf, bytes / 16); CK(hipDeviceSynchronize());
    printf("fetch_calib: every kernel touched %zu bytes once\n", bytes);
    return 0;
}
