// Test helper (not product code): hold workgroups on the device for a while, so that a launch that needs every CU at once -- the
// volume-resident SART sweep, tomo_tv_amd/csrc/sart_resident.hip.h -- finds part of the chip taken (tests/test_gpu_sart_resident.py).
// A workgroup of 1024 threads with 100 KB of LDS leaves no room for a resident workgroup (86 KB of LDS, every vector register of the
// four SIMDs) on the CU it sits on.
#include <hip/hip_runtime.h>
#include <cstdint>

__global__ __launch_bounds__(1024) void k_pin(long long ticks, int *sink)
{
    extern __shared__ int lds[];
    lds[threadIdx.x] = (int)threadIdx.x;
    __syncthreads();
    const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();        // 100 MHz
    while ((long long)__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
    if (lds[(threadIdx.x + 1) & 1023] < 0) *sink = 1;                      // (keeps the LDS allocation alive)
}

extern "C" {

// launches nwg pinned workgroups for `ms` milliseconds on a stream of their own and returns at once; *handle waits / cleans up
int pin_start(int device, int nwg, double ms, void **handle)
{
    if (hipSetDevice(device) != hipSuccess) return 1;
    hipStream_t st;
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) return 2;
    const int lds = 100 * 1024;
    if (hipFuncSetAttribute((const void *)k_pin, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return 3;
    int *sink = nullptr;
    if (hipMalloc((void **)&sink, sizeof(int)) != hipSuccess) return 4;
    hipLaunchKernelGGL(k_pin, dim3((unsigned)nwg), dim3(1024), lds, st, (long long)(ms * 1e5), sink);
    if (hipGetLastError() != hipSuccess) return 5;
    *handle = (void *)st;
    return 0;
}

int pin_running(void *handle) { return hipStreamQuery((hipStream_t)handle) == hipErrorNotReady ? 1 : 0; }

int pin_wait(void *handle)
{
    hipStream_t st = (hipStream_t)handle;
    if (hipStreamSynchronize(st) != hipSuccess) return 1;
    return hipStreamDestroy(st) == hipSuccess ? 0 : 2;
}

}
