// Shared host/device helpers for libacimg (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>

#include "../../include/acimg.h"

namespace acimg {

// ---- per-thread error text -------------------------------------------------------------
inline char* err_buf() {
    static thread_local char buf[512] = {0};
    return buf;
}
inline int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}
inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(ACIMG_ELAUNCH, "%s: %s", what, hipGetErrorString(e));
    return ACIMG_OK;
}
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// ---- device helpers ---------------------------------------------------------------------
using f32x4 = __attribute__((ext_vector_type(4))) float;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// block-wide sum for blockDim.x <= 1024 (multiple of 64); result valid in every thread
__device__ __forceinline__ float block_sum(float v, float* smem /* >= 17 floats */) {
    v = wave_sum(v);
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    __syncthreads();
    if (lane == 0) smem[wid] = v;
    __syncthreads();
    float r = 0.f;
    for (int i = 0; i < nw; ++i) r += smem[i];
    return r;
}

__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == ACIMG_ACT_RELU) return fmaxf(v, 0.f);
    if (act == ACIMG_ACT_SIGMOID) return 1.f / (1.f + __expf(-v));
    return v;
}

}  // namespace acimg
