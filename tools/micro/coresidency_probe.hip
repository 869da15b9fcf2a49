// Can an HBM-bound kernel share a CU with an MFMA-bound one when the register file leaves room?  (round 4)
//   hipcc --offload-arch=gfx950 -O3 tools/micro/coresidency_probe.hip -o gpurun_out/coresidency_probe && gpurun_out/coresidency_probe
// Kernel M: the trunk kernel's footprint - 512 threads, 68 KiB of LDS, two workgroups per CU - running K-step-shaped work
// (12 ds_read_b128 + 24 MFMAs 16x16x32 f16 per step and wave) with its VGPR allocation forced (a clobbered top register) to 128 (4 waves x 128 = the
// whole file, as shipped) or 96 (one more wave per SIMD fits).  Kernel C: a streaming float4 copy (256 threads, few
// registers, no LDS).  Timed: M alone, C alone, and M on one stream with C launched right behind it on another.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define VGTOP_128 "v127"
#define VGTOP_96 "v95"
#define VGTOP(VG) VGTOP_##VG
#define MFMA_LIKE(VG) \
__global__ __launch_bounds__(512) void mfma_like_##VG(const h16x8* src, float* out, int steps) { \
    extern __shared__ __attribute__((aligned(16))) char lds[]; \
    asm volatile("" ::: VGTOP(VG)); \
    const int tid = threadIdx.x; \
    for (int i = tid; i < 68 * 1024 / 16; i += 512) reinterpret_cast<h16x8*>(lds)[i] = src[(blockIdx.x * 131 + i) & 65535]; \
    __syncthreads(); \
    f32x4 acc[8]; \
_Pragma("unroll") \
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f}; \
    const int lane = tid & 63, wid = tid >> 6; \
    for (int s = 0; s < steps; ++s) { \
        h16x8 a[4], al[4], b[2], bl[2]; \
        const int base = ((s & 1) * 32768) + ((wid * 64 + lane) * 16) % 16384; \
_Pragma("unroll") \
        for (int i = 0; i < 4; ++i) { \
            a[i] = *reinterpret_cast<const h16x8*>(lds + base + i * 1024); \
            al[i] = *reinterpret_cast<const h16x8*>(lds + base + 8192 + i * 1024); \
        } \
_Pragma("unroll") \
        for (int j = 0; j < 2; ++j) { \
            b[j] = *reinterpret_cast<const h16x8*>(lds + base + 16384 + j * 1024); \
            bl[j] = *reinterpret_cast<const h16x8*>(lds + base + 24576 + j * 1024); \
        } \
_Pragma("unroll") \
        for (int i = 0; i < 4; ++i) \
_Pragma("unroll") \
            for (int j = 0; j < 2; ++j) acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl[j], a[i], acc[i * 2 + j], 0, 0, 0); \
_Pragma("unroll") \
        for (int i = 0; i < 4; ++i) \
_Pragma("unroll") \
            for (int j = 0; j < 2; ++j) acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[j], al[i], acc[i * 2 + j], 0, 0, 0); \
_Pragma("unroll") \
        for (int i = 0; i < 4; ++i) \
_Pragma("unroll") \
            for (int j = 0; j < 2; ++j) acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[j], a[i], acc[i * 2 + j], 0, 0, 0); \
        __builtin_amdgcn_s_barrier(); \
    } \
    f32x4 t = acc[0]; \
_Pragma("unroll") \
    for (int i = 1; i < 8; ++i) t += acc[i]; \
    if (t[0] == 123.456f) out[blockIdx.x * 512 + tid] = t[1]; \
}
MFMA_LIKE(128)
MFMA_LIKE(96)
template <int VG> struct Pick;
template <> struct Pick<128> { static constexpr auto fn = mfma_like_128; };
template <> struct Pick<96> { static constexpr auto fn = mfma_like_96; };

__global__ __launch_bounds__(256) void copy4(const float4* __restrict__ in, float4* __restrict__ out, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) out[i] = in[i];
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int VG>
int run(const h16x8* src, float* out, const float4* cin, float4* cout, long n4, hipStream_t sa, hipStream_t sb) {
    const int lds = 68 * 1024, steps = 600, grid = 512 * 6;     // six rounds of 512 resident workgroups
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(Pick<VG>::fn), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t e0, e1, e2;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
    float tm = 0, tc = 0, tb = 0, tb_c = 0;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, sa));
        hipLaunchKernelGGL(Pick<VG>::fn, dim3(grid), dim3(512), lds, sa, src, out, steps);
        CK(hipEventRecord(e1, sa));
        CK(hipDeviceSynchronize());
        CK(hipEventElapsedTime(&tm, e0, e1));
        CK(hipEventRecord(e0, sb));
        hipLaunchKernelGGL(copy4, dim3(8192), dim3(256), 0, sb, cin, cout, n4);
        CK(hipEventRecord(e1, sb));
        CK(hipDeviceSynchronize());
        CK(hipEventElapsedTime(&tc, e0, e1));
        // both: M first, C right behind it on the other stream
        CK(hipEventRecord(e0, sa));
        hipLaunchKernelGGL(Pick<VG>::fn, dim3(grid), dim3(512), lds, sa, src, out, steps);
        CK(hipEventRecord(e1, sa));
        CK(hipStreamWaitEvent(sb, e0, 0));
        hipLaunchKernelGGL(copy4, dim3(8192), dim3(256), 0, sb, cin, cout, n4);
        CK(hipEventRecord(e2, sb));
        CK(hipDeviceSynchronize());
        CK(hipEventElapsedTime(&tb, e0, e1));
        CK(hipEventElapsedTime(&tb_c, e0, e2));
    }
    printf("VGPR allocation %3d: MFMA-like alone %.3f ms | copy alone %.3f ms | together: MFMA-like done at %.3f ms, copy done at %.3f ms "
           "(sum of the two alone %.3f)\n", VG, tm, tc, tb, tb_c, tm + tc);
    return 0;
}

int main() {
    const long n4 = 48L << 20;                       // 768 MB in, 768 MB out
    h16x8* src; float* out; float4 *cin, *cout;
    CK(hipMalloc(&src, 65536 * 16)); CK(hipMalloc(&out, 512 * 6 * 512 * 4)); CK(hipMalloc(&cin, n4 * 16)); CK(hipMalloc(&cout, n4 * 16));
    std::vector<_Float16> h(65536 * 8);
    for (auto& v : h) v = (_Float16)((rand() % 2001 - 1000) / 1000.f);
    CK(hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemset(cin, 1, n4 * 16));
    hipStream_t sa, sb;
    CK(hipStreamCreate(&sa)); CK(hipStreamCreate(&sb));
    if (run<128>(src, out, cin, cout, n4, sa, sb)) return 1;
    if (run<96>(src, out, cin, cout, n4, sa, sb)) return 1;
    if (run<128>(src, out, cin, cout, n4, sa, sb)) return 1;
    if (run<96>(src, out, cin, cout, n4, sa, sb)) return 1;
    return 0;
}
