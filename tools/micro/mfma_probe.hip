// What does the matrix pipe give a 16x16x32 f16 MFMA stream shaped like the trunk kernels' K step?  (round 3)
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_probe.hip -o tools/debug/mfma_probe && tools/debug/mfma_probe
// Each variant: one or two 512-thread workgroups per CU on all 256 CUs, a loop of STEPS "K steps" of NM MFMAs per wave on
// independent accumulators, operands in registers (random, loaded once).  Reported: shader cycles per MFMA and SIMD
// (s_memtime around the loop, median over waves), wall time -> the clock the chip held, TFLOP/s.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

// MODE 0: 48 MFMAs per step in the ring kernel's order (3 sweeps over a 4x4 accumulator tile: bl.ah, bh.al, bh.ah)
// MODE 1: the same with one s_barrier per step
// MODE 2: 24 MFMAs per step (4x2 tile: the shipped kernels' wave tile), barrier per step
// MODE 3: MODE 1 plus 16 ds_read_b128 per step from LDS (conflict-free rows), results kept live
// MODE 4: MODE 1 plus 6 LDS-DMA requests (buffer_load_dwordx4 ... lds, 1 KiB each) per wave and step from an L2-resident
//         buffer into a rotating LDS slot (48 KiB per workgroup and step: the ring kernel's operand traffic), waited for
//         with vmcnt(6) at the end of the step (one step of latency hiding)
// MODE 5: MODE 3 + MODE 4 (everything a ring-kernel K step does except address arithmetic)
template <int MODE>
__global__ __launch_bounds__(512) void probe(const h16x8* src, float* sink, unsigned long long* cyc, int steps) {
    constexpr int TM = 4, TN = (MODE == 2) ? 2 : 4;
    __shared__ __attribute__((aligned(16))) char lds[MODE == 3 ? 49152 : (MODE >= 4 ? 3 * 49152 : 16)];
    const __amdgpu_buffer_rsrc_t rs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<h16x8*>(src), 0, 4096 * 16, 0x00020000);
    const int tid = threadIdx.x;
    h16x8 ah[TM], al[TM], bh[TN], bl[TN];
    for (int i = 0; i < TM; ++i) { ah[i] = src[(tid + 64 * i) & 4095]; al[i] = src[(tid + 64 * i + 17) & 4095]; }
    for (int j = 0; j < TN; ++j) { bh[j] = src[(tid + 64 * j + 33) & 4095]; bl[j] = src[(tid + 64 * j + 49) & 4095]; }
    if (MODE == 3 || MODE == 5) {
        for (int k = tid; k < 49152 / 16; k += 512) reinterpret_cast<h16x8*>(lds)[k] = src[k & 4095];
        __syncthreads();
    }
    f32x4 acc[TM][TN];
    for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int lane = tid & 63, wid = tid >> 6;
    const char* rd = lds + ((wid & 3) * 64 + (lane & 15)) * 64 + ((lane >> 4) << 4);
    h16x8 nx[16];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    int slot = 1;
    const unsigned goff = (unsigned)(((blockIdx.x * 7 + wid * 3) & 31) * 1024 + (tid & 63) * 16);   // L2-resident source
    for (int s = 0; s < steps; ++s) {
        if (MODE >= 1) __builtin_amdgcn_s_barrier();
        if (MODE >= 4) slot = slot == 2 ? 1 : 2;      // slot 0 holds the fragments MODE 5 reads
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl[j], ah[i], acc[i][j], 0, 0, 0);
            if (MODE == 3 || MODE == 5) { nx[2 * i] = *reinterpret_cast<const h16x8*>(rd + i * 2048); nx[2 * i + 1] = *reinterpret_cast<const h16x8*>(rd + i * 2048 + 1024); }
            if (MODE >= 4) {
                char* dst = lds + slot * 49152 + (wid * 6 + i) * 1024;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)dst, 16, goff, i * 4096, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[j], al[i], acc[i][j], 0, 0, 0);
            if (MODE == 3 || MODE == 5) { nx[8 + 2 * i] = *reinterpret_cast<const h16x8*>(rd + 16384 + i * 2048); nx[9 + 2 * i] = *reinterpret_cast<const h16x8*>(rd + 16384 + i * 2048 + 1024); }
            if (MODE >= 4 && i < 2) {
                char* dst = lds + slot * 49152 + (wid * 6 + 4 + i) * 1024;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)dst, 16, goff, 16384 + i * 4096, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[j], ah[i], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (MODE >= 4) __builtin_amdgcn_s_waitcnt((6 & 15) | 0x0F70);      // vmcnt(6): the previous step's requests landed
        if (MODE == 3 || MODE == 5) {        // the reads become the next step's operands (keeps them live, as in the kernel)
            __builtin_amdgcn_s_waitcnt(0xC07F);
#pragma unroll
            for (int i = 0; i < TM; ++i) { ah[i] = nx[2 * i]; al[i] = nx[8 + 2 * i]; }
#pragma unroll
            for (int j = 0; j < TN; ++j) { bh[j] = nx[2 * j + 1]; bl[j] = nx[9 + 2 * j]; }
        }
    }
    if (MODE >= 4) __builtin_amdgcn_s_waitcnt(0x0F70);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) s += acc[i][j][0] + acc[i][j][3];
    sink[blockIdx.x * 512 + tid] = s;
    if (lane == 0) cyc[blockIdx.x * 8 + wid] = t1 - t0;
}

template <int MODE>
static void run(const char* name, int wgs_per_cu, int steps, const h16x8* src, float* sink, unsigned long long* cyc) {
    const int nwg = 256 * wgs_per_cu;
    const int nm = (MODE == 2 ? 24 : 48);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(probe<MODE>, dim3(nwg), dim3(512), 0, 0, src, sink, cyc, steps);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int rep = 0; rep < 5; ++rep) hipLaunchKernelGGL(probe<MODE>, dim3(nwg), dim3(512), 0, 0, src, sink, cyc, steps);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    std::vector<unsigned long long> h(nwg * 8);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double med = (double)h[h.size() / 2];
    const double waves_per_simd = 2.0 * wgs_per_cu;
    const double per_mfma_simd = med / ((double)steps * nm * waves_per_simd);
    const double flops = (double)nwg * 8 * steps * nm * 2.0 * 16 * 16 * 32;
    printf("%-58s %d wg/CU: %.0f cycles/step and wave, %.2f cycles per MFMA and SIMD, %.3f ms, clock %.2f GHz, %.0f TFLOP/s (hw)\n",
           name, wgs_per_cu, med / steps, per_mfma_simd, ms, med / (ms * 1e-3) / 1e9, flops / (ms * 1e-3) / 1e12);
}

int main() {
    h16x8* src; float* sink; unsigned long long* cyc;
    std::vector<_Float16> h(4096 * 8);
    srand(1);
    for (auto& v : h) v = (_Float16)((rand() % 2001 - 1000) / 1000.0f);
    hipMalloc(&src, h.size() * 2); hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipMalloc(&sink, 512 * 512 * 4); hipMalloc(&cyc, 512 * 8 * 8);
    const int steps = 2000;
    run<0>("48 MFMA / step, 3 sweeps of a 4x4 tile, no barrier", 1, steps, src, sink, cyc);
    run<1>("48 MFMA / step + one s_barrier per step", 1, steps, src, sink, cyc);
    run<3>("48 MFMA / step + barrier + 16 ds_read_b128 (operands refreshed)", 1, steps, src, sink, cyc);
    run<4>("48 MFMA / step + barrier + 6 LDS-DMA requests per wave (48 KiB / step)", 1, steps, src, sink, cyc);
    run<5>("48 MFMA / step + barrier + 16 ds_read_b128 + 6 LDS-DMA requests", 1, steps, src, sink, cyc);
    run<2>("24 MFMA / step (4x2 tile) + barrier", 2, steps, src, sink, cyc);
    return 0;
}
