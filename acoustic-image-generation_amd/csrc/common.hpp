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

// ---- split-fp16 ("f16x3") operand format ------------------------------------------------------
// A value v is kept as hi = f16(v*SCALE), lo = f16(v*SCALE - hi): 22 mantissa bits in 4 bytes.  The
// power-of-two scales keep fp16 in range: activations x2^-2 (finite up to 2.6e5), weights x2^10.
constexpr float SPLIT3_WSCALE = 1024.f;
constexpr float SPLIT3_ASCALE = 0.25f;
constexpr float SPLIT3_OUTSCALE = 1.f / 256.f;   // undoes both on the fp32 accumulators
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));

// v (already scaled) -> packed hi / lo halves
__device__ __forceinline__ void split4_scaled(const float4 v, uint2& hi, uint2& lo) {
    const h16x4 h = {(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
    const h16x4 l = {(_Float16)(v.x - (float)h[0]), (_Float16)(v.y - (float)h[1]), (_Float16)(v.z - (float)h[2]),
                     (_Float16)(v.w - (float)h[3])};
    hi = __builtin_bit_cast(uint2, h);
    lo = __builtin_bit_cast(uint2, l);
}
// packed hi / lo halves -> fp32 values (unscaled)
__device__ __forceinline__ float4 unsplit4(const uint2 hi, const uint2 lo, float inv_scale) {
    const h16x4 h = __builtin_bit_cast(h16x4, hi), l = __builtin_bit_cast(h16x4, lo);
    return make_float4(((float)h[0] + (float)l[0]) * inv_scale, ((float)h[1] + (float)l[1]) * inv_scale,
                       ((float)h[2] + (float)l[2]) * inv_scale, ((float)h[3] + (float)l[3]) * inv_scale);
}

__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == ACIMG_ACT_RELU) return fmaxf(v, 0.f);
    if (act == ACIMG_ACT_SIGMOID) return 1.f / (1.f + __expf(-v));
    return v;
}

}  // namespace acimg
