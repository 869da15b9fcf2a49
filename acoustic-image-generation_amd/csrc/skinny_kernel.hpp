// Skinny GEMMs of the VAE heads: the two 12x16 VALID "convs" over the whole 12x16x148 feature map are ONE dense layer
// [batch <= 64][28 416] x [28 416][300] (models/unet_acresnet.py:73-76) - 34 MB of weights against a few rows of
// activations.  On the general implicit-GEMM kernels (64-row tiles, split-K slabs) its data gradient and its weight
// gradient ran at 0.9 / 1.4 TB/s of weight traffic; these two kernels read (write) the weight matrix exactly once, in
// rows, with exact-f32 MFMA (v_mfma_f32_16x16x4_f32: an fmaf chain, the arithmetic of the kernels they replace):
//
//   skinny_dgrad_kernel:  dx[m][k] = sum_n g[m][n] W[k][n]  (+ residual, masked by mask > 0)
//       a wave owns 16 weight rows k and walks n in steps of 16; a lane loads 16 bytes = 4 consecutive n, register j of
//       the load is the operand of MFMA j, whose reduction index therefore runs over n0 + 4 q + j - the same n set on
//       the g side, so no shuffles are needed;
//   skinny_wgrad_kernel:  dW[k][n] = sum_m x[m][k] g[m][n],  db[n] = sum_m g[m][n]
//       a wave owns 64 rows k x 64 columns n (16 accumulator blocks: rows k0 + 4 i + j, columns n0 + 4 i' + jn, again
//       from 16-byte loads whose registers feed different MFMAs), 4 batch rows per MFMA step; the four jn blocks of a
//       lane are 4 consecutive n: 16-byte stores, 256 contiguous bytes per weight row and wave.
//   skinny_fwd_kernel + skinny_fwd_reduce_kernel:  y[m][n] = act(bias[n] + sum_k x[m][k] W[k][n])
//       the reduction runs over the weight ROWS, so a wave takes a slab of SKINNY_KS rows x 64 columns (four 256-byte row
//       segments per load instruction: whole lines), leaves its [M][64] partial in the caller's workspace, and a second
//       launch adds the slabs in slab order (deterministic) with bias and activation.
// Rows / columns out of range come in as zeros through the buffer descriptors' range check (N % 4 == 0 is required, so
// a 16-byte load is wholly inside or wholly outside a row).
#pragma once
#include "igemm_kernel.hpp"

namespace acimg {

struct SkinnyParams {
    const float* W;      // [C][ldw]
    const float* G;      // [M][ldg]   (gradient w.r.t. the layer's output)
    const float* X;      // [M][ldx]   (wgrad: the layer's input)
    float* out;          // dgrad: dx [M][ldo];  wgrad: dW [C][ldw]
    float* db;           // wgrad: [N] or null
    const float* res;    // dgrad: optional residual [M][ldres]
    const float* mask;   // dgrad: optional mask [M][ldmask] (dx zeroed where mask <= 0)
    int M, C, N;         // batch rows, weight rows (input channels), weight columns (output channels, % 4 == 0)
    int ldw, ldg, ldx, ldo, ldres, ldmask;
    // forward: partial sums [slabs][M][N] in `part`, then out = act(bias + sum of slabs)
    float* part;
    const float* bias;
    int act, slabs;
};

constexpr int SKINNY_KS = 128;       // weight rows per forward slab (1110 one-wave workgroups for the 28 416-row heads)

template <int MB>        // 16-row blocks of the batch (M <= 16 MB)
__global__ __launch_bounds__(256) void skinny_dgrad_kernel(const SkinnyParams p) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int k0 = (blockIdx.x * 4 + wid) * 16;
    if (k0 >= p.C) return;
    const int i = lane & 15, q = lane >> 4;
    const __amdgpu_buffer_rsrc_t rsW =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.W), 0, (unsigned)((long)p.C * p.ldw * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsG =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.G), 0, (unsigned)((long)p.M * p.ldg * 4), 0x00020000);
    const bool krow_ok = k0 + i < p.C;
    const unsigned wrow = (unsigned)(k0 + i) * (unsigned)p.ldw * 4u;
    unsigned grow[MB];
    bool g_ok[MB];
#pragma unroll
    for (int b = 0; b < MB; ++b) {
        g_ok[b] = b * 16 + i < p.M;
        grow[b] = (unsigned)(b * 16 + i) * (unsigned)p.ldg * 4u;
    }
    f32x4 acc[MB];
#pragma unroll
    for (int b = 0; b < MB; ++b) acc[b] = f32x4{0.f, 0.f, 0.f, 0.f};
    // four n steps per trip: their 4 + 4 MB loads are in flight together (fewer than two waves share a SIMD here, so the
    // loop itself has to keep the memory pipe full)
    for (int n0 = 0; n0 < p.N; n0 += 64) {
        f32x4 w4[4], g4[4][MB];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int n = n0 + 16 * u + 4 * q;
            const bool n_ok = n < p.N;                     // N % 4 == 0: the whole 16-byte chunk is in or out
            w4[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                  rsW, (krow_ok && n_ok) ? wrow + (unsigned)n * 4u : OOB, 0, 0));
#pragma unroll
            for (int b = 0; b < MB; ++b)
                g4[u][b] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                         rsG, (g_ok[b] && n_ok) ? grow[b] + (unsigned)n * 4u : OOB, 0, 0));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int b = 0; b < MB; ++b)  // D[m][k] += sum_q g[m][n + 4 q + j] * W[k][n + 4 q + j]
                    acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(g4[u][b][j], w4[u][j], acc[b], 0, 0, 0);
    }
    // lane holds D[m = 16 b + 4 q + r][k = k0 + i]
    const int k = k0 + i;
    if (k < p.C) {
#pragma unroll
        for (int b = 0; b < MB; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = b * 16 + 4 * q + r;
                if (m < p.M) {
                    float v = acc[b][r];
                    if (p.res) v += p.res[(long)m * p.ldres + k];
                    if (p.mask && !(p.mask[(long)m * p.ldmask + k] > 0.f)) v = 0.f;
                    p.out[(long)m * p.ldo + k] = v;
                }
            }
    }
}

__global__ __launch_bounds__(256) void skinny_wgrad_kernel(const SkinnyParams p, const int ngroups) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int task = blockIdx.x * 4 + wid;
    const int kb = task / ngroups, ng = task - kb * ngroups;
    const int k0 = kb * 64, n0 = ng * 64;
    const int i = lane & 15, q = lane >> 4;
    if (k0 >= p.C) return;
    const __amdgpu_buffer_rsrc_t rsX =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.X), 0, (unsigned)((long)p.M * p.ldx * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsG =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.G), 0, (unsigned)((long)p.M * p.ldg * 4), 0x00020000);
    const int kx = k0 + 4 * i, ngc = n0 + 4 * i;
    const bool kx_ok = kx < p.C;               // C % 4 == 0 is required too
    const bool ng_ok = ngc < p.N;
    f32x4 acc[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int jn = 0; jn < 4; ++jn) acc[j][jn] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 gsum = f32x4{0.f, 0.f, 0.f, 0.f};          // bias gradient: this lane's share of the column sums of g
    for (int m0 = 0; m0 < p.M; m0 += 16) {           // four MFMA steps per trip, their loads in flight together
        f32x4 x4[4], g4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int m = m0 + 4 * u + q;
            const bool m_ok = m < p.M;
            x4[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                  rsX, (m_ok && kx_ok) ? ((unsigned)m * (unsigned)p.ldx + (unsigned)kx) * 4u : OOB, 0, 0));
            g4[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                  rsG, (m_ok && ng_ok) ? ((unsigned)m * (unsigned)p.ldg + (unsigned)ngc) * 4u : OOB, 0, 0));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            gsum += g4[u];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int jn = 0; jn < 4; ++jn)    // D[i][i'] += sum_q x[m + q][k0 + 4 i + j] * g[m + q][n0 + 4 i' + jn]
                    acc[j][jn] = __builtin_amdgcn_mfma_f32_16x16x4f32(x4[u][j], g4[u][jn], acc[j][jn], 0, 0, 0);
        }
    }
    if (p.db && kb == 0) {
        // rows m = q (mod 4) were summed by lane group q: the four groups meet through two shuffles (fixed order)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float v = gsum[c];
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            gsum[c] = v;
        }
        if (q == 0 && ng_ok) *reinterpret_cast<f32x4*>(p.db + ngc) = gsum;
    }
    // lane holds, for block (j, jn), D[row 4 q + r][column i]: dW[k0 + 4 (4 q + r) + j][n0 + 4 i + jn]
    if (ngc < p.N) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k = k0 + 4 * (4 * q + r) + j;
                if (k < p.C) {
                    const f32x4 v = {acc[j][0][r], acc[j][1][r], acc[j][2][r], acc[j][3][r]};
                    *reinterpret_cast<f32x4*>(p.out + (long)k * p.ldw + ngc) = v;
                }
            }
    }
}

template <int MB>
__global__ __launch_bounds__(64) void skinny_fwd_kernel(const SkinnyParams p) {
    const int lane = threadIdx.x;
    const int slab = blockIdx.x, n0 = blockIdx.y * 64;
    const int i = lane & 15, q = lane >> 4;
    const __amdgpu_buffer_rsrc_t rsW =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.W), 0, (unsigned)((long)p.C * p.ldw * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsX =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.X), 0, (unsigned)((long)p.M * p.ldx * 4), 0x00020000);
    const int nc = n0 + 4 * i;
    const bool n_ok = nc < p.N;
    unsigned xrow[MB];
    bool x_ok[MB];
#pragma unroll
    for (int b = 0; b < MB; ++b) {
        x_ok[b] = b * 16 + i < p.M;
        xrow[b] = (unsigned)(b * 16 + i) * (unsigned)p.ldx * 4u;
    }
    f32x4 acc[MB][4];
#pragma unroll
    for (int b = 0; b < MB; ++b)
#pragma unroll
        for (int jn = 0; jn < 4; ++jn) acc[b][jn] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int kend = min(p.C, (slab + 1) * SKINNY_KS);
    for (int k0 = slab * SKINNY_KS; k0 < kend; k0 += 64) {      // four 16-row steps per trip: 16 KiB of weights in flight per wave
        f32x4 w4[4][4], x4[4][MB];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = k0 + 16 * u + 4 * q;                   // this lane's rows k .. k + 3 (C % 4 == 0)
            const bool k_ok = k < kend;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                w4[u][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                         rsW, (k_ok && n_ok) ? ((unsigned)(k + j) * (unsigned)p.ldw + (unsigned)nc) * 4u : OOB, 0, 0));
#pragma unroll
            for (int b = 0; b < MB; ++b)
                x4[u][b] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                         rsX, (k_ok && x_ok[b]) ? xrow[b] + (unsigned)k * 4u : OOB, 0, 0));
        }
        __builtin_amdgcn_sched_barrier(0);           // every load of the trip is issued before its first MFMA
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int b = 0; b < MB; ++b)
#pragma unroll
                    for (int jn = 0; jn < 4; ++jn)   // D[m][i'] += sum_q x[m][k + 4 q + j] * W[k + 4 q + j][n0 + 4 i' + jn]
                        acc[b][jn] = __builtin_amdgcn_mfma_f32_16x16x4f32(x4[u][b][j], w4[u][j][jn], acc[b][jn], 0, 0, 0);
    }
    // lane holds, for block jn, D[m = 16 b + 4 q + r][column i]: part[slab][m][n0 + 4 i + jn]
    if (n_ok) {
        float* dst = p.part + (long)slab * p.M * p.N;
#pragma unroll
        for (int b = 0; b < MB; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = b * 16 + 4 * q + r;
                if (m < p.M)
                    *reinterpret_cast<f32x4*>(dst + (long)m * p.N + nc) = f32x4{acc[b][0][r], acc[b][1][r], acc[b][2][r], acc[b][3][r]};
            }
    }
}

__global__ __launch_bounds__(256) void skinny_fwd_reduce_kernel(const SkinnyParams p) {
    const int t = blockIdx.x * 256 + threadIdx.x;       // one float4 of the [M][N] result
    const int nq = p.N >> 2;
    if (t >= p.M * nq) return;
    const int m = t / nq, n = (t - m * nq) * 4;
    const float* src = p.part + (long)m * p.N + n;
    const long stride = (long)p.M * p.N;
    f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
    int sl = 0;
    for (; sl + 8 <= p.slabs; sl += 8) {                // eight loads in flight, added in slab order
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4*>(src + (sl + u) * stride);
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; sl < p.slabs; ++sl) s += *reinterpret_cast<const f32x4*>(src + sl * stride);
    if (p.bias) s += *reinterpret_cast<const f32x4*>(p.bias + n);
#pragma unroll
    for (int c = 0; c < 4; ++c) s[c] = apply_act(s[c], p.act);
    *reinterpret_cast<f32x4*>(p.out + (long)m * p.ldo + n) = s;
}

}  // namespace acimg
