// kernels_common.hip.h -- vector helpers, block reductions, layout conversion at the host boundary
// Part of kernels.hip.h (include that, not this: the families share helpers and constants in the order kernels.hip.h lists them).
#pragma once

namespace tomo {


// ---- small vector helpers -----------------------------------------------------------------------
template <int V> struct VecOf;
template <> struct VecOf<1> { typedef float T; };
template <> struct VecOf<2> { typedef float T __attribute__((ext_vector_type(2))); };
template <> struct VecOf<4> { typedef float T __attribute__((ext_vector_type(4))); };

template <int V> __device__ __forceinline__ typename VecOf<V>::T vzero();
template <> __device__ __forceinline__ float vzero<1>() { return 0.f; }
template <> __device__ __forceinline__ VecOf<2>::T vzero<2>() { VecOf<2>::T v = {0.f, 0.f}; return v; }
template <> __device__ __forceinline__ VecOf<4>::T vzero<4>() { VecOf<4>::T v = {0.f, 0.f, 0.f, 0.f}; return v; }

template <int V> __device__ __forceinline__ float velem(const typename VecOf<V>::T &v, int i);
template <> __device__ __forceinline__ float velem<1>(const float &v, int) { return v; }
template <> __device__ __forceinline__ float velem<2>(const VecOf<2>::T &v, int i) { return v[i]; }
template <> __device__ __forceinline__ float velem<4>(const VecOf<4>::T &v, int i) { return v[i]; }

template <int V> __device__ __forceinline__ void vset(typename VecOf<V>::T &v, int i, float f);
template <> __device__ __forceinline__ void vset<1>(float &v, int, float f) { v = f; }
template <> __device__ __forceinline__ void vset<2>(VecOf<2>::T &v, int i, float f) { v[i] = f; }
template <> __device__ __forceinline__ void vset<4>(VecOf<4>::T &v, int i, float f) { v[i] = f; }

// ---- block reduction of a double into one of NPART partial slots ----------------------------------
constexpr int NPART = 256;

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// all threads of a 256-thread block call this; one atomic per block
__device__ __forceinline__ void block_accumulate(double v, double *__restrict__ part)
{
    __shared__ double red[4];
    v = wave_sum(v);
    int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(&part[blockIdx.x & (NPART - 1)], red[0] + red[1] + red[2] + red[3]);
}

// Non-temporal access by experiment bit (make EXTRA=-DTOMO_NT=<mask>); bits that paid are folded into NT_DEFAULT.
#ifndef TOMO_NT
#define TOMO_NT 0
#endif
#ifndef TOMO_NT_OFF   // experiment: bits of NT_DEFAULT switched off
#define TOMO_NT_OFF 0
#endif
constexpr int NT_DEFAULT = 32 | 256;   // 32: k_fp_tile partial sums (SIRT iteration -5 %); 256: k_fgp_fused outputs (-4 %)
template <int BIT, typename T>
__device__ __forceinline__ T nt_ld(const T *p)
{
    if constexpr (((TOMO_NT | NT_DEFAULT) & ~TOMO_NT_OFF & BIT) != 0) return __builtin_nontemporal_load(p);
    else return *p;
}
template <int BIT, typename T>
__device__ __forceinline__ void nt_st(T v, T *p)
{
    if constexpr (((TOMO_NT | NT_DEFAULT) & ~TOMO_NT_OFF & BIT) != 0) {
        __builtin_nontemporal_store(v, p);
    } else *p = v;
}

__global__ void k_finalize(double *__restrict__ part, double *__restrict__ dst)
{
    double v = part[threadIdx.x];  // launched with NPART threads
    part[threadIdx.x] = 0.0;       // leave the buffer ready for the next reduction (no memset launch per reduction)
    __shared__ double red[NPART / 64];
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0;
        for (int i = 0; i < NPART / 64; ++i) s += red[i];
        *dst = s;
    }
}

// ---- layout conversion at the host boundary -------------------------------------------------------
// host [ns][m]  ->  device [m][sx]   (padding slices written as zero)
__global__ __launch_bounds__(256) void k_transpose_in(const float *__restrict__ src, float *__restrict__ dst,
                                                       int ns, int64_t m, int sx)
{
    __shared__ float tile[32][33];
    int64_t m0 = (int64_t)blockIdx.x * 32;
    int s0 = blockIdx.y * 32;
    int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        int s = s0 + r;
        int64_t mm = m0 + tx;
        tile[r][tx] = (s < ns && mm < m) ? src[(int64_t)s * m + mm] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        int64_t mm = m0 + r;
        int s = s0 + tx;
        if (mm < m && s < sx) dst[mm * sx + s] = tile[tx][r];
    }
}

// device [m][sx] -> host [ns][m]
__global__ __launch_bounds__(256) void k_transpose_out(const float *__restrict__ src, float *__restrict__ dst,
                                                        int ns, int64_t m, int sx)
{
    __shared__ float tile[32][33];
    int64_t m0 = (int64_t)blockIdx.x * 32;
    int s0 = blockIdx.y * 32;
    int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        int64_t mm = m0 + r;
        int s = s0 + tx;
        tile[r][tx] = (mm < m && s < sx) ? src[mm * sx + s] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        int s = s0 + r;
        int64_t mm = m0 + tx;
        if (s < ns && mm < m) dst[(int64_t)s * m + mm] = tile[tx][r];
    }
}

__global__ void k_scatter_slice(const float *__restrict__ img, float *__restrict__ vol, int64_t m, int sx, int s)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) vol[i * sx + s] = img[i];
}

__global__ void k_gather_slice(const float *__restrict__ vol, float *__restrict__ img, int64_t m, int sx, int s)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) img[i] = vol[i * sx + s];
}

// vector register blocks the asm loops of the list projectors and of the resident sweep bind (v[64:127], s[36:67] ...)
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v32f __attribute__((ext_vector_type(32)));
typedef uint32_t u16v __attribute__((ext_vector_type(16)));

}  // namespace tomo
