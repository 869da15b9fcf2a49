// HBM-bound elementwise / reduction kernels of the acoustic-image train step (gfx950).
// Each is a single pass with 16-byte accesses where the layout allows, wave-64 shuffle
// reductions and fp32 (or fp64 where noted) accumulation.
#include "common.hpp"

namespace acimg {

// ------------------------------------------------------------------------------------------
// batch-norm statistics -> scale/shift (+ moving averages)
// block = 1024 threads = (1024 / CPB) row groups x CPB channels; 4 independent partial sums per thread keep
// several loads in flight (the partials are tiny, the kernel is pure load latency)
// ------------------------------------------------------------------------------------------
// CPB channels per workgroup (32, or 8 when there are thousands of partial rows and few channels: the stem, the
// full-resolution U-Net layers), 1024 / CPB row groups
template <int CPB>
__global__ __launch_bounds__(1024) void bn_finalize_kernel(
    const float* stats, int rows, int C, int ld, double count, const float* gamma, const float* beta,
    float* moving_mean, float* moving_var, float decay, float eps, int training, float* scale,
    float* shift, float* save_mean, float* save_invstd) {
    // Latency-bound kernel (54 launches per step): (1) the per-channel constants are requested FIRST so that their
    // global-load latency overlaps the partial-sum loads; (2) the row groups are combined by a 4-way split of the
    // final loop instead of one thread walking all NRG partials of a channel through dependent LDS reads (that loop
    // alone was ~3 us of the 6 us a launch took).
    constexpr int NRG = 1024 / CPB;
    constexpr int Q = 4;                       // threads per channel in the final combine
    static_assert(NRG % Q == 0, "row groups per combining thread");
    __shared__ double red[2][NRG][CPB];
    __shared__ double red2[2][Q][CPB];
    const int cl = threadIdx.x % CPB, rg = threadIdx.x / CPB;
    const int c = blockIdx.x * CPB + cl;
    float g = 1.f, b = 0.f, mm = 0.f, mv = 1.f;
    if (rg == 0 && c < C) {
        if (gamma) g = gamma[c];
        if (beta) b = beta[c];
        if (moving_mean) {
            mm = moving_mean[c];
            mv = moving_var[c];
        }
    }
    double s1 = 0.0, s2 = 0.0;
    if (training && c < C) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, b0 = 0.f, b1 = 0.f, b2 = 0.f, b3 = 0.f;
        int r = rg;
        for (; r + 3 * NRG < rows; r += 4 * NRG) {
            a0 += stats[((long)r * 2 + 0) * ld + c];
            b0 += stats[((long)r * 2 + 1) * ld + c];
            a1 += stats[((long)(r + NRG) * 2 + 0) * ld + c];
            b1 += stats[((long)(r + NRG) * 2 + 1) * ld + c];
            a2 += stats[((long)(r + 2 * NRG) * 2 + 0) * ld + c];
            b2 += stats[((long)(r + 2 * NRG) * 2 + 1) * ld + c];
            a3 += stats[((long)(r + 3 * NRG) * 2 + 0) * ld + c];
            b3 += stats[((long)(r + 3 * NRG) * 2 + 1) * ld + c];
        }
        for (; r < rows; r += NRG) {
            a0 += stats[((long)r * 2 + 0) * ld + c];
            b0 += stats[((long)r * 2 + 1) * ld + c];
        }
        s1 = ((double)a0 + (double)a1) + ((double)a2 + (double)a3);
        s2 = ((double)b0 + (double)b1) + ((double)b2 + (double)b3);
    }
    float mean, var;
    if (training) {
        red[0][rg][cl] = s1;
        red[1][rg][cl] = s2;
        __syncthreads();
        if (rg < Q) {                          // Q threads per channel, NRG / Q partials each, in row-group order
            double t1 = 0.0, t2 = 0.0;
#pragma unroll
            for (int i = 0; i < NRG / Q; ++i) {
                t1 += red[0][rg * (NRG / Q) + i][cl];
                t2 += red[1][rg * (NRG / Q) + i][cl];
            }
            red2[0][rg][cl] = t1;
            red2[1][rg][cl] = t2;
        }
        __syncthreads();
        if (rg != 0 || c >= C) return;
        s1 = (red2[0][0][cl] + red2[0][1][cl]) + (red2[0][2][cl] + red2[0][3][cl]);
        s2 = (red2[1][0][cl] + red2[1][1][cl]) + (red2[1][2][cl] + red2[1][3][cl]);
        const double m = s1 / count;
        double v = s2 / count - m * m;
        if (v < 0.0) v = 0.0;
        mean = (float)m;
        var = (float)v;
        if (moving_mean) {
            const double unbiased = count > 1.0 ? v * (count / (count - 1.0)) : v;
            moving_mean[c] = decay * mm + (1.f - decay) * mean;
            moving_var[c] = decay * mv + (1.f - decay) * (float)unbiased;
        }
    } else {
        if (rg != 0 || c >= C) return;
        mean = mm;
        var = mv;
    }
    const float invstd = 1.f / sqrtf(var + eps);
    const float sc = g * invstd;
    scale[c] = sc;
    shift[c] = b - mean * sc;
    if (save_mean) save_mean[c] = mean;
    if (save_invstd) save_invstd[c] = invstd;
}

// out = relu(a*sa+ta + (b*sb+tb | b))
__global__ __launch_bounds__(256) void bn_add_relu_kernel(const float* a, const float* sa,
                                                          const float* ta, const float* b,
                                                          const float* sb, const float* tb,
                                                          float* out, long total4, int OH, int OW,
                                                          int C4, int BH, int BW, int bstride) {
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total4; idx += (long)gridDim.x * 256) {
        const int c4 = (int)(idx % C4);
        const long pix = idx / C4;
        const int c = c4 * 4;
        float4 va = *reinterpret_cast<const float4*>(a + pix * (C4 * 4) + c);
        const float4 s = *reinterpret_cast<const float4*>(sa + c);
        const float4 t = *reinterpret_cast<const float4*>(ta + c);
        long bpix = pix;
        if (bstride != 1) {
            const int ow = (int)(pix % OW);
            const long t2 = pix / OW;
            const int oh = (int)(t2 % OH);
            const long n = t2 / OH;
            bpix = (n * BH + (long)oh * bstride) * BW + (long)ow * bstride;
        }
        float4 vb = *reinterpret_cast<const float4*>(b + bpix * (C4 * 4) + c);
        if (sb) {
            const float4 s2 = *reinterpret_cast<const float4*>(sb + c);
            const float4 t2 = *reinterpret_cast<const float4*>(tb + c);
            vb.x = vb.x * s2.x + t2.x;
            vb.y = vb.y * s2.y + t2.y;
            vb.z = vb.z * s2.z + t2.z;
            vb.w = vb.w * s2.w + t2.w;
        }
        float4 o;
        o.x = fmaxf(va.x * s.x + t.x + vb.x, 0.f);
        o.y = fmaxf(va.y * s.y + t.y + vb.y, 0.f);
        o.z = fmaxf(va.z * s.z + t.z + vb.z, 0.f);
        o.w = fmaxf(va.w * s.w + t.w + vb.w, 0.f);
        *reinterpret_cast<float4*>(out + pix * (C4 * 4) + c) = o;
    }
}

__global__ __launch_bounds__(256) void bn_relu_maxpool_kernel(const float* x, const float* scale,
                                                              const float* shift, float* out,
                                                              long total4, int H, int W, int C4,
                                                              int OH, int OW, int pad_t, int pad_l) {
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total4; idx += (long)gridDim.x * 256) {
        const int c = (int)(idx % C4) * 4;
        long t = idx / C4;
        const int ow = (int)(t % OW);
        t /= OW;
        const int oh = (int)(t % OH);
        const long n = t / OH;
        const float4 s = *reinterpret_cast<const float4*>(scale + c);
        const float4 sh = *reinterpret_cast<const float4*>(shift + c);
        float4 m = make_float4(0.f, 0.f, 0.f, 0.f);  // post-ReLU values are >= 0
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int ih = oh * 2 - pad_t + r;
            if ((unsigned)ih >= (unsigned)H) continue;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int iw = ow * 2 - pad_l + q;
                if ((unsigned)iw >= (unsigned)W) continue;
                const float4 v = *reinterpret_cast<const float4*>(x + ((n * H + ih) * W + iw) * (C4 * 4) + c);
                m.x = fmaxf(m.x, v.x * s.x + sh.x);
                m.y = fmaxf(m.y, v.y * s.y + sh.y);
                m.z = fmaxf(m.z, v.z * s.z + sh.z);
                m.w = fmaxf(m.w, v.w * s.w + sh.w);
            }
        }
        *reinterpret_cast<float4*>(out + idx * 4) = m;
    }
}

__global__ __launch_bounds__(256) void bn_relu_kernel(const float* x, const float* scale,
                                                      const float* shift, float* y, long rows, int C,
                                                      int ldx, int ldy) {
    const long total = rows * C;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const long r = idx / C;
        const int c = (int)(idx - r * C);
        y[r * ldy + c] = fmaxf(x[r * ldx + c] * scale[c] + shift[c], 0.f);
    }
}

// one block per channel
__global__ __launch_bounds__(1024) void bn_relu_bwd_kernel(const float* x, const float* y,
                                                          const float* gy, const float* gamma,
                                                          const float* mean, const float* invstd,
                                                          float* gx, float* dgamma, float* dbeta,
                                                          long rows, int C) {
    __shared__ float sm[32];
    const int c = blockIdx.x;
    const float mu = mean[c], is = invstd[c];
    float s1 = 0.f, s2 = 0.f;
    for (long r = threadIdx.x; r < rows; r += blockDim.x) {
        const float g = y[r * C + c] > 0.f ? gy[r * C + c] : 0.f;
        s1 += g;
        s2 += g * (x[r * C + c] - mu) * is;
    }
    s1 = block_sum(s1, sm);
    s2 = block_sum(s2, sm);
    if (threadIdx.x == 0) {
        dbeta[c] = s1;
        dgamma[c] = s2;
    }
    const float k = gamma[c] * is;
    const float inv_n = 1.f / (float)rows;
    for (long r = threadIdx.x; r < rows; r += blockDim.x) {
        const float g = y[r * C + c] > 0.f ? gy[r * C + c] : 0.f;
        const float xh = (x[r * C + c] - mu) * is;
        gx[r * C + c] = k * (g - s1 * inv_n - xh * s2 * inv_n);
    }
}

// ------------------------------------------------------------------------------------------
// Training-mode batch-norm backward for the conv-BN-ReLU stacks of the RGB / spectrogram U-Nets
// (models/unet_architecture.py:161-166): many row blocks per launch, float4 over channels.
//   g   = gy * [raw*scale + shift > 0]                    (ReLU mask recomputed, y is not re-read)
//   s1  = sum_rows g = dbeta,   s2 = sum_rows g * xhat = dgamma,   xhat = (raw - mean) * invstd
//   gx  = gamma * invstd * (g - s1/n - xhat * s2/n)
// pass 1 leaves per-block partials [blocks][2][C]; pass 2 (one block) adds them in block order
// (deterministic) and writes dgamma/dbeta; pass 3 is elementwise.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* x, int ldx, const float* gy, int ldgy,
                                                            const float* scale, const float* shift,
                                                            const float* mean, const float* invstd, long rows,
                                                            int C, long rows_per_block, float* partial) {
    extern __shared__ float sm[];                 // [256][8]
    const int c4n = C >> 2;
    const int tc = threadIdx.x % c4n, tr = threadIdx.x / c4n;
    const int rstep = 256 / c4n;                  // rows covered per sweep (threads beyond rstep*c4n idle)
    const int c = tc * 4;
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    if (tr < rstep) {
        const float4 sc = *reinterpret_cast<const float4*>(scale + c), sh = *reinterpret_cast<const float4*>(shift + c);
        const float4 mu = *reinterpret_cast<const float4*>(mean + c), is = *reinterpret_cast<const float4*>(invstd + c);
        const float scs[4] = {sc.x, sc.y, sc.z, sc.w}, shs[4] = {sh.x, sh.y, sh.z, sh.w};
        const float mus[4] = {mu.x, mu.y, mu.z, mu.w}, iss[4] = {is.x, is.y, is.z, is.w};
        const long r0 = (long)blockIdx.x * rows_per_block;
        const long r1 = min(rows, r0 + rows_per_block);
        // four rows per trip with sums of their own: eight 16-byte loads in flight per thread (one row per trip streamed the
        // two largest U-Net shapes at 2.5 TB/s - 54 us for 137 MB - while the apply pass, more bytes, took 33 us)
        float t1[4][4], t2[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int k = 0; k < 4; ++k) t1[u][k] = t2[u][k] = 0.f;
        long r = r0 + tr;
        for (; r + 3L * rstep < r1; r += 4L * rstep) {
            float4 xv[4], gv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                xv[u] = *reinterpret_cast<const float4*>(x + (r + (long)u * rstep) * ldx + c);
                gv[u] = *reinterpret_cast<const float4*>(gy + (r + (long)u * rstep) * ldgy + c);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float xs[4] = {xv[u].x, xv[u].y, xv[u].z, xv[u].w}, gs[4] = {gv[u].x, gv[u].y, gv[u].z, gv[u].w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float g = xs[k] * scs[k] + shs[k] > 0.f ? gs[k] : 0.f;
                    t1[u][k] += g;
                    t2[u][k] += g * (xs[k] - mus[k]) * iss[k];
                }
            }
        }
        for (; r < r1; r += rstep) {
            const float4 xv = *reinterpret_cast<const float4*>(x + r * ldx + c);
            const float4 gv = *reinterpret_cast<const float4*>(gy + r * ldgy + c);
            const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, gs[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float g = xs[k] * scs[k] + shs[k] > 0.f ? gs[k] : 0.f;
                t1[0][k] += g;
                t2[0][k] += g * (xs[k] - mus[k]) * iss[k];
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            s1[k] = (t1[0][k] + t1[1][k]) + (t1[2][k] + t1[3][k]);
            s2[k] = (t2[0][k] + t2[1][k]) + (t2[2][k] + t2[3][k]);
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        sm[threadIdx.x * 8 + k] = s1[k];
        sm[threadIdx.x * 8 + 4 + k] = s2[k];
    }
    __syncthreads();
    if (threadIdx.x < c4n) {                      // thread tc sums its column group over the row slots, in order
        float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int j = 0; j < rstep; ++j)
#pragma unroll
            for (int k = 0; k < 8; ++k) a[k] += sm[(j * c4n + threadIdx.x) * 8 + k];
        float* dst = partial + (long)blockIdx.x * 2 * C;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            dst[c + k] = a[k];
            dst[C + c + k] = a[4 + k];
        }
    }
}

// one workgroup per 32 channels x {dbeta, dgamma}: 8 row groups of 32 lanes walk the blocks, fixed-order tree
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float* partial, int blocks, int C,
                                                              float* dgamma, float* dbeta, float* sums, int cw) {
    // 256 threads = cw channels x 256 / cw partial-row groups (cw = 8 / 16 / 32 by the channel count: with few channels more
    // groups walk the <= 256 partial rows - 8 dependent loads per thread instead of 32; the kernel is pure latency, 22 launches
    // per U-Net step); two sums in flight per thread; groups combined in order
    __shared__ float red[32][32];
    const int ng = 256 / cw;
    const int cl = threadIdx.x % cw, rg = threadIdx.x / cw;
    const int which = blockIdx.y;                       // 0: sum g (dbeta), 1: sum g*xhat (dgamma)
    const int c = blockIdx.x * cw + cl;
    float a0 = 0.f, a1 = 0.f;
    if (c < C) {
        int j = rg;
        for (; j + ng < blocks; j += 2 * ng) {
            a0 += partial[(long)j * 2 * C + which * C + c];
            a1 += partial[(long)(j + ng) * 2 * C + which * C + c];
        }
        if (j < blocks) a0 += partial[(long)j * 2 * C + which * C + c];
    }
    red[rg][cl] = a0 + a1;
    __syncthreads();
    if (rg == 0 && c < C) {
        float t = 0.f;
        for (int i = 0; i < ng; ++i) t += red[i][cl];
        (which ? dgamma : dbeta)[c] = t;
        sums[which * C + c] = t;
    }
}

__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* x, int ldx, const float* gy, int ldgy,
                                                           const float* scale, const float* shift,
                                                           const float* mean, const float* invstd,
                                                           const float* gamma, const float* sums, long rows, int C,
                                                           float* gx, int ldgx) {
    const int c4n = C >> 2;
    const float inv_n = 1.f / (float)rows;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < rows * c4n; idx += (long)gridDim.x * 256) {
        const long r = idx / c4n;
        const int c = (int)(idx - r * c4n) * 4;
        const float4 xv = *reinterpret_cast<const float4*>(x + r * ldx + c);
        const float4 gv = *reinterpret_cast<const float4*>(gy + r * ldgy + c);
        const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, gs[4] = {gv.x, gv.y, gv.z, gv.w};
        float o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float g = xs[k] * scale[c + k] + shift[c + k] > 0.f ? gs[k] : 0.f;
            const float xh = (xs[k] - mean[c + k]) * invstd[c + k];
            o[k] = gamma[c + k] * invstd[c + k] * (g - sums[c + k] * inv_n - xh * sums[C + c + k] * inv_n);
        }
        *reinterpret_cast<float4*>(gx + r * ldgx + c) = make_float4(o[0], o[1], o[2], o[3]);
    }
}

__global__ __launch_bounds__(256) void pad_channels_kernel(const float* x, float* y, long pixels,
                                                           int C, int Cp) {
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < pixels * Cp; idx += (long)gridDim.x * 256) {
        const long p = idx / Cp;
        const int c = (int)(idx - p * Cp);
        y[idx] = c < C ? x[p * C + c] : 0.f;
    }
}

// [N,H,W,C] -> interior of a zeroed [N,Hp,Wp,Cp] image at (pt, pl); the border and the pad channels are never
// written (the caller zeroes the buffer once)
__global__ __launch_bounds__(256) void pad_image_kernel(const float* x, float* y, int H, int W, int C, int Cp,
                                                        int Hp, int Wp, int pt, int pl, long pixels) {
    // one pixel per thread: C contiguous floats in, one (zero-padded) Cp-float pixel out
    for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < pixels; p += (long)gridDim.x * 256) {
        const int w = (int)(p % W);
        const long q = p / W;
        const int h = (int)(q % H);
        const long n = q / H;
        const float* src = x + p * C;
        float* dst = y + ((n * Hp + h + pt) * Wp + w + pl) * Cp;
        if (C == 3 && Cp == 4) {
            *reinterpret_cast<float4*>(dst) = make_float4(src[0], src[1], src[2], 0.f);
        } else {
            for (int c = 0; c < C; ++c) dst[c] = src[c];
        }
    }
}

__global__ __launch_bounds__(256) void tile_mfcc_kernel(const float* mfcc, float* out, int HW, int C,
                                                        long total) {
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int c = (int)(idx % C);
        const long n = idx / ((long)HW * C);
        out[idx] = mfcc[n * C + c];
    }
}

// ------------------------------------------------------------------------------------------
// per-sample min-max normalisation; S pixel chunks per sample (grid S x N), two phases each way:
// partial (min, max) / partial gradient sums per chunk into the workspace, then every chunk combines the S
// partials of its sample in a fixed order and does its share of the elementwise work.  Tie counts are
// integer-valued float atomics (exact, order-independent), so results are bit-reproducible.
// ------------------------------------------------------------------------------------------
static inline int minmax_chunks(int P, int C) {
    long s = ((long)P * C + 2047) / 2048;
    if (s > 32) s = 32;
    if (s > P) s = P;
    if (s < 1) s = 1;
    const int ppc = (P + (int)s - 1) / (int)s;
    return (P + ppc - 1) / ppc;
}

__global__ __launch_bounds__(256) void minmax_part_kernel(const float* x, int ldx, float* part, float* mm,
                                                          int P, int C, int ppc) {
    __shared__ float sm[32];
    const int n = blockIdx.y, sidx = blockIdx.x, S = gridDim.x;
    const int p0 = sidx * ppc, np = min(ppc, P - p0);
    const float* xs = x + ((long)n * P + p0) * ldx;
    const int cnt = np * C;
    float mn = INFINITY, mx = -INFINITY;
    for (int i = threadIdx.x; i < cnt; i += 256) {
        const int p = i / C, c = i - p * C;
        const float v = xs[(long)p * ldx + c];
        mn = fminf(mn, v);
        mx = fmaxf(mx, v);
    }
    mn = wave_min(mn);
    mx = wave_max(mx);
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) {
        sm[wid] = mn;
        sm[4 + wid] = mx;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        part[((long)n * S + sidx) * 2 + 0] = fminf(fminf(sm[0], sm[1]), fminf(sm[2], sm[3]));
        part[((long)n * S + sidx) * 2 + 1] = fmaxf(fmaxf(sm[4], sm[5]), fmaxf(sm[6], sm[7]));
        if (sidx == 0) {
            mm[n * 4 + 2] = 0.f;
            mm[n * 4 + 3] = 0.f;
        }
    }
}

__global__ __launch_bounds__(256) void minmax_apply_kernel(const float* x, int ldx, float* out, int ldo,
                                                           const float* part, float* mm, int P, int C, int ppc) {
    __shared__ float sm[32];
    const int n = blockIdx.y, sidx = blockIdx.x, S = gridDim.x;
    float mn = INFINITY, mx = -INFINITY;
    for (int k = 0; k < S; ++k) {
        mn = fminf(mn, part[((long)n * S + k) * 2 + 0]);
        mx = fmaxf(mx, part[((long)n * S + k) * 2 + 1]);
    }
    const float D = mx - mn;
    const int p0 = sidx * ppc, np = min(ppc, P - p0);
    const float* xs = x + ((long)n * P + p0) * ldx;
    float* os = out + ((long)n * P + p0) * ldo;
    const int cnt = np * C;
    float cmin = 0.f, cmax = 0.f;
    for (int i = threadIdx.x; i < cnt; i += 256) {
        const int p = i / C, c = i - p * C;
        const float v = xs[(long)p * ldx + c];
        cmin += (v == mn) ? 1.f : 0.f;
        cmax += (v == mx) ? 1.f : 0.f;
        os[(long)p * ldo + c] = (v - mn) / D;
    }
    cmin = block_sum(cmin, sm);
    cmax = block_sum(cmax, sm);
    if (threadIdx.x == 0) {
        if (sidx == 0) {
            mm[n * 4 + 0] = mn;
            mm[n * 4 + 1] = mx;
        }
        if (cmin != 0.f) atomicAdd(&mm[n * 4 + 2], cmin);
        if (cmax != 0.f) atomicAdd(&mm[n * 4 + 3], cmax);
    }
}

__global__ __launch_bounds__(256) void minmax_bwd_part_kernel(const float* x, int ldx, const float* go, int ldgo,
                                                              const float* mm, float* part, int P, int C, int ppc) {
    __shared__ float sm[32];
    const int n = blockIdx.y, sidx = blockIdx.x, S = gridDim.x;
    const int p0 = sidx * ppc, np = min(ppc, P - p0);
    const float* xs = x + ((long)n * P + p0) * ldx;
    const float* gs = go + ((long)n * P + p0) * ldgo;
    const float mn = mm[n * 4 + 0], D = mm[n * 4 + 1] - mn;
    const int cnt = np * C;
    float s1 = 0.f, s2 = 0.f;
    for (int i = threadIdx.x; i < cnt; i += 256) {
        const int p = i / C, c = i - p * C;
        const float g = gs[(long)p * ldgo + c];
        const float o = (xs[(long)p * ldx + c] - mn) / D;
        s1 += g;
        s2 += g * o;
    }
    s1 = block_sum(s1, sm);
    s2 = block_sum(s2, sm);
    if (threadIdx.x == 0) {
        part[((long)n * S + sidx) * 2 + 0] = s1;
        part[((long)n * S + sidx) * 2 + 1] = s2;
    }
}

__global__ __launch_bounds__(256) void minmax_bwd_kernel(const float* x, int ldx, const float* go,
                                                         int ldgo, const float* mm, const float* part, float* gx,
                                                         int ldgx, int P, int C, int ppc, int accumulate,
                                                         int mask_relu) {
    const int n = blockIdx.y, sidx = blockIdx.x, S = gridDim.x;
    const int p0 = sidx * ppc, np = min(ppc, P - p0);
    const float* xs = x + ((long)n * P + p0) * ldx;
    const float* gs = go + ((long)n * P + p0) * ldgo;
    float* gxs = gx + ((long)n * P + p0) * ldgx;
    const float mn = mm[n * 4 + 0], mx = mm[n * 4 + 1], cmin = mm[n * 4 + 2], cmax = mm[n * 4 + 3];
    const float D = mx - mn;
    float s1 = 0.f, s2 = 0.f;
    for (int k = 0; k < S; ++k) {
        s1 += part[((long)n * S + k) * 2 + 0];
        s2 += part[((long)n * S + k) * 2 + 1];
    }
    const float gmin = -(s1 - s2) / D / cmin;
    const float gmax = -s2 / D / cmax;
    const int cnt = np * C;
    for (int i = threadIdx.x; i < cnt; i += 256) {
        const int p = i / C, c = i - p * C;
        const float v = xs[(long)p * ldx + c];
        float g = gs[(long)p * ldgo + c] / D;
        if (v == mn) g += gmin;
        if (v == mx) g += gmax;
        if (accumulate) g += gxs[(long)p * ldgx + c];
        if (mask_relu && !(v > 0.f)) g = 0.f;
        gxs[(long)p * ldgx + c] = g;
    }
}

// ------------------------------------------------------------------------------------------
// latent: softplus, reparameterisation, KL ; one block per sample
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float softplus_f(float x) {
    // log(1+exp(x)) evaluated as TF does: max(x,0) + log1p(exp(-|x|))
    return fmaxf(x, 0.f) + log1pf(expf(-fabsf(x)));
}

__global__ __launch_bounds__(256) void latent_fwd_kernel(const float* heads, const float* eps,
                                                         float* z, int ldz, float* sigma, float* kl,
                                                         int Z) {
    __shared__ float sm[32];
    const int n = blockIdx.x;
    float acc = 0.f;
    for (int j = threadIdx.x; j < Z; j += 256) {
        const float mu = heads[(long)n * 2 * Z + j];
        const float sg = softplus_f(heads[(long)n * 2 * Z + Z + j]);
        sigma[(long)n * Z + j] = sg;
        z[(long)n * ldz + j] = mu + sg * eps[(long)n * Z + j];
        acc += mu * mu + sg * sg - logf(1e-8f + sg * sg) - 1.f;
    }
    acc = block_sum(acc, sm);
    if (threadIdx.x == 0) kl[n] = 0.5f * acc;
}

__global__ __launch_bounds__(256) void latent_bwd_kernel(const float* heads, const float* eps,
                                                         const float* sigma, const float* gz, int ldgz,
                                                         float klw, float* gheads, int N, int Z) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= N * Z) return;
    const int n = idx / Z, j = idx - n * Z;
    const float mu = heads[(long)n * 2 * Z + j];
    const float sraw = heads[(long)n * 2 * Z + Z + j];
    const float sg = sigma[idx];
    const float g = gz[(long)n * ldgz + j];
    const float dmu = g + klw * mu;
    const float dsg = g * eps[idx] + klw * (sg - sg / (1e-8f + sg * sg));
    const float dsp = 1.f / (1.f + expf(-sraw));  // d softplus
    gheads[(long)n * 2 * Z + j] = dmu;
    gheads[(long)n * 2 * Z + Z + j] = dsg * dsp;
}

// softplus and its backward (the std tower of the latent associators, models/multimodal.py:48,107)
__global__ __launch_bounds__(256) void softplus_fwd_kernel(const float* x, int ldx, float* y, int ldy, int rows, int C) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= rows * C) return;
    const int r = idx / C, c = idx - r * C;
    y[(long)r * ldy + c] = softplus_f(x[(long)r * ldx + c]);
}
__global__ __launch_bounds__(256) void softplus_bwd_kernel(const float* x, int ldx, const float* gy, int ldgy, float* gx,
                                                           int ldgx, int rows, int C) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= rows * C) return;
    const int r = idx / C, c = idx - r * C;
    gx[(long)r * ldgx + c] = gy[(long)r * ldgy + c] / (1.f + expf(-x[(long)r * ldx + c]));
}

// the RGB / spectrogram U-Nets use the second head as sigma directly (models/unet_architecture.py:66-69):
// z = mean + variance * eps, kl[n] = 0.5 * sum_j (mu^2 + s^2 - log(1e-8 + s^2) - 1)
__global__ __launch_bounds__(256) void latent_linear_fwd_kernel(const float* heads, const float* eps, float* z,
                                                                int ldz, float* kl, int Z) {
    __shared__ float sm[32];
    const int n = blockIdx.x;
    float acc = 0.f;
    for (int j = threadIdx.x; j < Z; j += 256) {
        const float mu = heads[(long)n * 2 * Z + j];
        const float sg = heads[(long)n * 2 * Z + Z + j];
        z[(long)n * ldz + j] = mu + sg * eps[(long)n * Z + j];
        acc += mu * mu + sg * sg - logf(1e-8f + sg * sg) - 1.f;
    }
    acc = block_sum(acc, sm);
    if (threadIdx.x == 0) kl[n] = 0.5f * acc;
}

__global__ __launch_bounds__(256) void latent_linear_bwd_kernel(const float* heads, const float* eps,
                                                                const float* gz, int ldgz, float klw,
                                                                float* gheads, int N, int Z) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= N * Z) return;
    const int n = idx / Z, j = idx - n * Z;
    const float mu = heads[(long)n * 2 * Z + j];
    const float sg = heads[(long)n * 2 * Z + Z + j];
    const float g = gz[(long)n * ldgz + j];
    gheads[(long)n * 2 * Z + j] = g + klw * mu;
    gheads[(long)n * 2 * Z + Z + j] = g * eps[idx] + klw * (sg - sg / (1e-8f + sg * sg));
}

// ------------------------------------------------------------------------------------------
// DualCamNet classifier pieces (models/dualcamnet.py:82-106, models/base.py:34-38): non-overlapping VALID max
// pooling, the spatial reduce_sum, and the clip-level softmax cross-entropy of
// trainer/trainer_reconstructed_class.py:49-56 (logits averaged over the clip's frames).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const float* x, int ldx, float* y, int ldy, int H, int W,
                                                          int C, int OH, int OW, int k, long total) {
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int c = (int)(idx % C);
        long t = idx / C;
        const int ow = (int)(t % OW);
        t /= OW;
        const int oh = (int)(t % OH);
        const long n = t / OH;
        float m = -INFINITY;
        for (int r = 0; r < k; ++r)
            for (int q = 0; q < k; ++q)
                m = fmaxf(m, x[((n * H + oh * k + r) * W + ow * k + q) * ldx + c]);
        y[((n * OH + oh) * OW + ow) * ldy + c] = m;
    }
}

// gx = gradient w.r.t. the PRE-activation of the ReLU layer that feeds the pool: the window's gradient goes to
// its first maximum (tf.nn.max_pool's argmax) and only where the activation is positive
__global__ __launch_bounds__(256) void maxpool_relu_bwd_kernel(const float* x, int ldx, const float* gy, int ldgy,
                                                               float* gx, int ldgx, int H, int W, int C, int OH,
                                                               int OW, int k, long total) {
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int c = (int)(idx % C);
        long t = idx / C;
        const int w = (int)(t % W);
        t /= W;
        const int h = (int)(t % H);
        const long n = t / H;
        const int oh = h / k, ow = w / k;
        float g = 0.f;
        const float v = x[((n * H + h) * W + w) * ldx + c];
        if (oh < OH && ow < OW && v > 0.f) {
            bool take = true;
            for (int r = 0; r < k; ++r)
                for (int q = 0; q < k; ++q) {
                    const int hh = oh * k + r, ww = ow * k + q;
                    const float u = x[((n * H + hh) * W + ww) * ldx + c];
                    const bool before = hh < h || (hh == h && ww < w);
                    if (u > v || (u == v && before)) take = false;
                }
            if (take) g = gy[((n * OH + oh) * OW + ow) * ldgy + c];
        }
        gx[((n * H + h) * W + w) * ldgx + c] = g;
    }
}

// y[n][c] = sum_p x[n][p][c]; one workgroup per (sample, 64-channel group)
__global__ __launch_bounds__(256) void spatial_sum_kernel(const float* x, int ldx, float* y, int P, int C) {
    __shared__ float red[4][64];
    const int n = blockIdx.x, c = blockIdx.y * 64 + (threadIdx.x & 63), rg = threadIdx.x >> 6;
    float s = 0.f;
    if (c < C)
        for (int p = rg; p < P; p += 4) s += x[((long)n * P + p) * ldx + c];
    red[rg][threadIdx.x & 63] = s;
    __syncthreads();
    if (rg == 0 && c < C) y[(long)n * C + c] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

__global__ __launch_bounds__(256) void spatial_sum_relu_bwd_kernel(const float* x, int ldx, const float* gy, float* gx,
                                                                   int ldgx, int P, int C, long total) {
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int c = (int)(idx % C);
        const long np = idx / C;
        const long n = np / P;
        gx[np * ldgx + c] = x[np * ldx + c] > 0.f ? gy[n * C + c] : 0.f;
    }
}

// one workgroup per clip: logits averaged over the F frames, softmax cross-entropy against `label`;
// out[0] += loss / clips, out[1] += (argmax == label); g[n*F+f][c] = (p_c - [c == label]) / (clips * F)
__global__ __launch_bounds__(64) void clip_softmax_ce_kernel(const float* logits, int ldl, int F, int K,
                                                             const int* labels, int clips, float* out, float* g,
                                                             int ldg) {
    const int n = blockIdx.x, c = threadIdx.x;
    float m = -INFINITY;
    if (c < K) {
        float a = 0.f;
        for (int f = 0; f < F; ++f) a += logits[((long)n * F + f) * ldl + c];
        m = a / (float)F;
    }
    const float mx = wave_max(m);
    const float e = c < K ? expf(m - mx) : 0.f;
    const float se = wave_sum(e);
    const float p = e / se;
    const int lab = labels[n];
    // argmax with the lowest index among ties (tf.argmax)
    int am = c < K && m == mx ? c : 1 << 20;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) am = min(am, __shfl_xor(am, o, 64));
    if (c == lab) {
        atomicAdd(out, -(m - mx - logf(se)) / (float)clips);
        if (am == lab) atomicAdd(out + 1, 1.f);
    }
    if (g && c < K) {
        const float gv = (p - (c == lab ? 1.f : 0.f)) / ((float)clips * (float)F);
        for (int f = 0; f < F; ++f) g[((long)n * F + f) * ldg + c] = gv;
    }
}

// ------------------------------------------------------------------------------------------
// reconstruction loss (MSE + Huber delta=1) and gradient w.r.t. pre-sigmoid logits
// ------------------------------------------------------------------------------------------
// Ordered combination of per-workgroup partial sums (bit-reproducible, unlike float atomics): every workgroup
// parks its NV partials in `scratch` (agent-scope stores), takes a ticket, and the LAST arriver adds all
// partials in workgroup order with a fixed tree and accumulates them into dst[0..NV).  scratch: int ticket at
// [0] (zero before the first use; the last arriver resets it), partials from float index 4 on.  Call from all
// 256 threads; v is meaningful on thread 0.  scratch == nullptr: plain float atomics.
template <int NV>
__device__ __forceinline__ void ordered_block_sums(const float (&v)[NV], float* scratch, float* dst, float* sm) {
    if (scratch == nullptr) {
        if (threadIdx.x == 0)
#pragma unroll
            for (int i = 0; i < NV; ++i) atomicAdd(&dst[i], v[i]);
        return;
    }
    int* ticket = reinterpret_cast<int*>(scratch);
    float* part = scratch + 4;
    __shared__ int last_flag;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i)
            __hip_atomic_store(&part[(long)blockIdx.x * NV + i], v[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        const int t = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last_flag = t == (int)gridDim.x - 1;
        if (last_flag) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    if (!last_flag) return;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        float acc = 0.f;
        for (int b = threadIdx.x; b < (int)gridDim.x; b += 256)
            acc += __hip_atomic_load(&part[(long)b * NV + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        acc = block_sum(acc, sm);
        if (threadIdx.x == 0) dst[i] += acc;
    }
    if (threadIdx.x == 0) __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ __launch_bounds__(256) void recon_loss_kernel(const float* yhat, const float* target,
                                                         float* glogit, float* sums, long count,
                                                         float w_mse, float w_huber, float* scratch) {
    __shared__ float sm[32];
    float s_mse = 0.f, s_hub = 0.f;
    const float inv = 1.f / (float)count;
    const long n4 = count >> 2;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const float4 yh = reinterpret_cast<const float4*>(yhat)[i];
        const float4 tg = reinterpret_cast<const float4*>(target)[i];
        float4 g;
        const float yv[4] = {yh.x, yh.y, yh.z, yh.w};
        const float tv[4] = {tg.x, tg.y, tg.z, tg.w};
        float gv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float e = yv[k] - tv[k];
            const float ae = fabsf(e);
            const float q = fminf(ae, 1.f);
            s_mse += e * e;
            s_hub += 0.5f * q * q + (ae - q);
            const float dl = (w_mse * 2.f * e + w_huber * fminf(fmaxf(e, -1.f), 1.f)) * inv;
            gv[k] = dl * yv[k] * (1.f - yv[k]);
        }
        g.x = gv[0];
        g.y = gv[1];
        g.z = gv[2];
        g.w = gv[3];
        if (glogit) reinterpret_cast<float4*>(glogit)[i] = g;
    }
    // tail (count not a multiple of 4)
    for (long i = (n4 << 2) + (long)blockIdx.x * 256 + threadIdx.x; i < count; i += (long)gridDim.x * 256) {
        const float e = yhat[i] - target[i];
        const float ae = fabsf(e);
        const float q = fminf(ae, 1.f);
        s_mse += e * e;
        s_hub += 0.5f * q * q + (ae - q);
        if (glogit)
            glogit[i] = (w_mse * 2.f * e + w_huber * fminf(fmaxf(e, -1.f), 1.f)) * inv * yhat[i] * (1.f - yhat[i]);
    }
    s_mse = block_sum(s_mse, sm);
    s_hub = block_sum(s_hub, sm);
    const float v[2] = {s_mse, s_hub};
    ordered_block_sums<2>(v, scratch, sums, sm);
}

__global__ __launch_bounds__(256) void sumsq_kernel(const float* x, long n, float* out, float* scratch) {
    __shared__ float sm[32];
    float s = 0.f;
    const long n4 = n >> 2;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    for (long i = (n4 << 2) + (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
        s += x[i] * x[i];
    s = block_sum(s, sm);
    const float v[1] = {s};
    ordered_block_sums<1>(v, scratch, out, sm);
}

__global__ __launch_bounds__(256) void axpy_kernel(float a, const float* x, float* y, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) y[i] += a * x[i];
}

__global__ __launch_bounds__(256) void adam_kernel(float* p, const float* g, float* m, float* v, long n,
                                                   float lr_t, float b1, float b2, float eps, float gs) {
    const long n4 = n >> 2;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        float4 pv = reinterpret_cast<float4*>(p)[i];
        const float4 gv = reinterpret_cast<const float4*>(g)[i];
        float4 mv = reinterpret_cast<float4*>(m)[i];
        float4 vv = reinterpret_cast<float4*>(v)[i];
        float* pp = &pv.x;
        const float* gp = &gv.x;
        float* mp = &mv.x;
        float* vp = &vv.x;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float gg = gp[k] * gs;
            mp[k] = b1 * mp[k] + (1.f - b1) * gg;
            vp[k] = b2 * vp[k] + (1.f - b2) * gg * gg;
            pp[k] -= lr_t * mp[k] / (sqrtf(vp[k]) + eps);
        }
        reinterpret_cast<float4*>(p)[i] = pv;
        reinterpret_cast<float4*>(m)[i] = mv;
        reinterpret_cast<float4*>(v)[i] = vv;
    }
    for (long i = (n4 << 2) + (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float gg = g[i] * gs;
        m[i] = b1 * m[i] + (1.f - b1) * gg;
        v[i] = b2 * v[i] + (1.f - b2) * gg * gg;
        p[i] -= lr_t * m[i] / (sqrtf(v[i]) + eps);
    }
}


// dst[p][c] = [accumulate ? dst : 0] + src[p][c], zeroed where mask[p][c] <= 0 (mask optional)
__global__ __launch_bounds__(256) void grad_slice_kernel(const float* src, int ldsrc, float* dst, int lddst,
                                                         const float* mask, int ldmask, long pixels, int C,
                                                         int accumulate) {
    const long total = pixels * C;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const long p = idx / C;
        const int c = (int)(idx - p * C);
        float v = src[p * ldsrc + c];
        if (accumulate) v += dst[p * lddst + c];
        if (mask && !(mask[p * ldmask + c] > 0.f)) v = 0.f;
        dst[p * lddst + c] = v;
    }
}

// out = {mse, huber, latent, reg, total}
__global__ void loss_finalize_kernel(const float* sums, const float* kl, int N, double count, float latent_w,
                                     float half_wd, float w_mse, float w_huber, float* out) {
    __shared__ float sm[32];
    float a = 0.f;
    if (kl)
        for (int i = threadIdx.x; i < N; i += blockDim.x) a += kl[i];
    a = block_sum(a, sm);
    if (threadIdx.x == 0) {
        const float mse = (float)((double)sums[0] / count);
        const float hub = (float)((double)sums[1] / count);
        const float lat = kl ? latent_w * (a / (float)N) : 0.f;
        const float reg = half_wd * sums[2];
        out[0] = mse;
        out[1] = hub;
        out[2] = lat;
        out[3] = reg;
        out[4] = lat + w_mse * mse + w_huber * hub + reg;
    }
}


// ------------------------------------------------------------------------------------------
// standard normal samples, Philox4x32-10 counter RNG + Box-Muller (stands where the reference
// calls tf.random_normal, models/unet_acresnet.py:77): 4 samples per counter, stateless.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void philox_round(uint32_t& c0, uint32_t& c1, uint32_t& c2, uint32_t& c3,
                                             uint32_t k0, uint32_t k1) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
}

__global__ __launch_bounds__(256) void randn_kernel(float* out, long n, uint64_t seed, uint64_t offset) {
    const long quads = (n + 3) >> 2;
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < quads; q += (long)gridDim.x * 256) {
        const uint64_t ctr = (uint64_t)q + offset;
        uint32_t c0 = (uint32_t)ctr, c1 = (uint32_t)(ctr >> 32), c2 = 0x9E3779B9u, c3 = 0xBB67AE85u;
        uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            philox_round(c0, c1, c2, c3, k0, k1);
            k0 += 0x9E3779B9u;
            k1 += 0xBB67AE85u;
        }
        const float u0 = ((float)(c0 >> 8) + 0.5f) * (1.0f / 16777216.0f);
        const float u1 = ((float)(c1 >> 8) + 0.5f) * (1.0f / 16777216.0f);
        const float u2 = ((float)(c2 >> 8) + 0.5f) * (1.0f / 16777216.0f);
        const float u3 = ((float)(c3 >> 8) + 0.5f) * (1.0f / 16777216.0f);
        const float r0 = sqrtf(-2.f * logf(u0)), r1 = sqrtf(-2.f * logf(u2));
        float s0, cs0, s1, cs1;
        sincospif(2.f * u1, &s0, &cs0);
        sincospif(2.f * u3, &s1, &cs1);
        const float v[4] = {r0 * cs0, r0 * s0, r1 * cs1, r1 * s1};
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (q * 4 + k < n) out[q * 4 + k] = v[k];
    }
}


// per-channel sum of squared errors: out[c] += sum_p (a[p][c]-b[p][c])^2  (C <= 64)
__global__ __launch_bounds__(256) void sqerr_channels_kernel(const float* a, const float* b, long pixels, int C,
                                                             float* out) {
    __shared__ float acc[64];
    if (threadIdx.x < 64) acc[threadIdx.x] = 0.f;
    __syncthreads();
    const long total = pixels * C;
    // each thread keeps a fixed channel: stride of the loop is a multiple of C
    const long stride = ((long)gridDim.x * 256 / C) * C;
    const long start = (long)blockIdx.x * 256 + threadIdx.x;
    float s = 0.f;
    int c = -1;
    if (start < stride) {
        c = (int)(start % C);
        for (long i = start; i < total; i += stride) {
            const float e = a[i] - b[i];
            s += e * e;
        }
    }
    if (c >= 0) atomicAdd(&acc[c], s);
    __syncthreads();
    if (threadIdx.x < C) atomicAdd(&out[threadIdx.x], acc[threadIdx.x]);
}


// ------------------------------------------------------------------------------------------
// producers of the split-fp16 activation format consumed by the trunk kernels (igemm_split3d_kernel.hpp, "bricks"): two
// fp16 planes, lo plane lo_off bytes after the hi plane, values carry the 2^-2 scale.  A plane of a [P pixels][C] tensor
// (C % 32 == 0) is ceil(P / 16) x C / 32 bricks of 1 KiB: brick (f >> 4, c >> 5) holds 16 rows (f & 15) of 64 bytes, the
// logical 16-byte chunk (c >> 3) & 3 of a row at physical chunk ((c >> 3) ^ swz(f)) & 3, swz(f) = -(f >> 2) & 3 - the LDS
// image of 16 tile rows, so the consumer's LDS-DMA request is one contiguous KiB.
// Work items (one float4 = 4 channels each) are enumerated in brick order: a wave covers 8 pixels x 32 channels, i.e. it
// reads eight whole 128-byte lines of the fp32 source and writes 512 contiguous bytes (four whole lines) of each plane.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned plane_off(unsigned f, unsigned c, unsigned CQ) {      // c % 4 == 0
    return ((f >> 4) * CQ + (c >> 5)) * 1024u + ((f & 15u) << 6) + ((((c >> 3) ^ (0u - (f >> 2))) & 3u) << 4) + ((c & 4u) << 1);
}
struct BrickItem {
    unsigned pix, c, off;      // pixel, first of its 4 channels, byte offset inside a plane
};
__device__ __forceinline__ BrickItem brick_item(unsigned u, unsigned CQ, bool pow2, int sh) {
    const unsigned l = u & 63u, hb = u >> 6;
    const unsigned ph = pow2 ? hb >> sh : hb / CQ;
    const unsigned cq = hb - ph * CQ;
    BrickItem it;
    it.pix = ph * 8u + (l >> 3);
    it.c = cq * 32u + (l & 7u) * 4u;
    it.off = plane_off(it.pix, it.c, CQ);
    return it;
}

__global__ __launch_bounds__(256) void bn_relu_split_kernel(const float* x, const float* scale, const float* shift,
                                                            int relu, char* out, long lo_off, unsigned items, unsigned P,
                                                            int C) {
    const unsigned CQ = (unsigned)C >> 5;
    const bool pow2 = (CQ & (CQ - 1)) == 0;
    const int sh = 31 - __builtin_clz(CQ);
    for (unsigned u = blockIdx.x * 256u + threadIdx.x; u < items; u += gridDim.x * 256u) {
        const BrickItem it = brick_item(u, CQ, pow2, sh);
        if (it.pix >= P) continue;
        float4 v = *reinterpret_cast<const float4*>(x + (long)it.pix * C + it.c);
        if (scale) {
            const float4 s = *reinterpret_cast<const float4*>(scale + it.c);
            const float4 t = *reinterpret_cast<const float4*>(shift + it.c);
            v.x = v.x * s.x + t.x; v.y = v.y * s.y + t.y; v.z = v.z * s.z + t.z; v.w = v.w * s.w + t.w;
        }
        if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        v.x *= SPLIT3_ASCALE; v.y *= SPLIT3_ASCALE; v.z *= SPLIT3_ASCALE; v.w *= SPLIT3_ASCALE;
        uint2 hi, lo;
        split4_scaled(v, hi, lo);
        *reinterpret_cast<uint2*>(out + it.off) = hi;
        *reinterpret_cast<uint2*>(out + lo_off + it.off) = lo;
    }
}

// out = relu(a*sa+ta + shortcut) -> split planes (+ optional fp32 copy); shortcut = b32*sb+tb (projection,
// raw fp32) or the previous unit output read back from ITS split planes (identity, optional subsampling)
__global__ __launch_bounds__(256) void bn_add_relu_split_kernel(
    const float* a, const float* sa, const float* ta, const float* b32, const float* sb, const float* tb,
    const char* bsp, long b_lo_off, char* out, long out_lo_off, float* out32, unsigned items, unsigned P, int OH, int OW,
    int C, int BH, int BW, int bstride) {
    const unsigned CQ = (unsigned)C >> 5;
    const bool pow2 = (CQ & (CQ - 1)) == 0;
    const int sh = 31 - __builtin_clz(CQ);
    for (unsigned u = blockIdx.x * 256u + threadIdx.x; u < items; u += gridDim.x * 256u) {
        const BrickItem it = brick_item(u, CQ, pow2, sh);
        if (it.pix >= P) continue;
        const long idx = (long)it.pix * C + it.c;
        const float4 va = *reinterpret_cast<const float4*>(a + idx);
        const float4 s = *reinterpret_cast<const float4*>(sa + it.c);
        const float4 t = *reinterpret_cast<const float4*>(ta + it.c);
        unsigned bpix = it.pix;
        if (bstride != 1) {
            const unsigned ow = it.pix % (unsigned)OW;
            const unsigned t2 = it.pix / (unsigned)OW;
            const unsigned oh = t2 % (unsigned)OH;
            const unsigned n = t2 / (unsigned)OH;
            bpix = (n * (unsigned)BH + oh * (unsigned)bstride) * (unsigned)BW + ow * (unsigned)bstride;
        }
        float4 vb;
        if (b32) {
            vb = *reinterpret_cast<const float4*>(b32 + (long)bpix * C + it.c);
            const float4 s2 = *reinterpret_cast<const float4*>(sb + it.c);
            const float4 t2 = *reinterpret_cast<const float4*>(tb + it.c);
            vb.x = vb.x * s2.x + t2.x; vb.y = vb.y * s2.y + t2.y; vb.z = vb.z * s2.z + t2.z; vb.w = vb.w * s2.w + t2.w;
        } else {
            const unsigned e = bstride != 1 ? plane_off(bpix, it.c, CQ) : it.off;
            vb = unsplit4(*reinterpret_cast<const uint2*>(bsp + e), *reinterpret_cast<const uint2*>(bsp + b_lo_off + e),
                          1.f / SPLIT3_ASCALE);
        }
        float4 o;
        o.x = fmaxf(va.x * s.x + t.x + vb.x, 0.f);
        o.y = fmaxf(va.y * s.y + t.y + vb.y, 0.f);
        o.z = fmaxf(va.z * s.z + t.z + vb.z, 0.f);
        o.w = fmaxf(va.w * s.w + t.w + vb.w, 0.f);
        if (out32) *reinterpret_cast<float4*>(out32 + idx) = o;
        if (out) {
            o.x *= SPLIT3_ASCALE; o.y *= SPLIT3_ASCALE; o.z *= SPLIT3_ASCALE; o.w *= SPLIT3_ASCALE;
            uint2 hi, lo;
            split4_scaled(o, hi, lo);
            *reinterpret_cast<uint2*>(out + it.off) = hi;
            *reinterpret_cast<uint2*>(out + out_lo_off + it.off) = lo;
        }
    }
}

__global__ __launch_bounds__(256) void bn_relu_maxpool_split_kernel(const float* x, const float* scale,
                                                                    const float* shift, char* out, long lo_off,
                                                                    unsigned items, unsigned P, int H, int W, int C, int OH,
                                                                    int OW, int pad_t, int pad_l) {
    const unsigned CQ = (unsigned)C >> 5;
    const bool pow2 = (CQ & (CQ - 1)) == 0;
    const int shq = 31 - __builtin_clz(CQ);
    for (unsigned u = blockIdx.x * 256u + threadIdx.x; u < items; u += gridDim.x * 256u) {
        const BrickItem it = brick_item(u, CQ, pow2, shq);
        if (it.pix >= P) continue;
        const int c = (int)it.c;
        const int ow = (int)(it.pix % (unsigned)OW);
        const unsigned t = it.pix / (unsigned)OW;
        const int oh = (int)(t % (unsigned)OH);
        const long n = t / (unsigned)OH;
        const float4 s = *reinterpret_cast<const float4*>(scale + c);
        const float4 sh = *reinterpret_cast<const float4*>(shift + c);
        float4 m = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int ih = oh * 2 - pad_t + r;
            if ((unsigned)ih >= (unsigned)H) continue;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int iw = ow * 2 - pad_l + q;
                if ((unsigned)iw >= (unsigned)W) continue;
                const float4 v = *reinterpret_cast<const float4*>(x + ((n * H + ih) * W + iw) * C + c);
                m.x = fmaxf(m.x, v.x * s.x + sh.x);
                m.y = fmaxf(m.y, v.y * s.y + sh.y);
                m.z = fmaxf(m.z, v.z * s.z + sh.z);
                m.w = fmaxf(m.w, v.w * s.w + sh.w);
            }
        }
        m.x *= SPLIT3_ASCALE; m.y *= SPLIT3_ASCALE; m.z *= SPLIT3_ASCALE; m.w *= SPLIT3_ASCALE;
        uint2 hi, lo;
        split4_scaled(m, hi, lo);
        *reinterpret_cast<uint2*>(out + it.off) = hi;
        *reinterpret_cast<uint2*>(out + lo_off + it.off) = lo;
    }
}

static inline int ew_grid(long work_items) {
    long b = (work_items + 255) / 256;
    if (b > 4096) b = 4096;
    if (b < 1) b = 1;
    return (int)b;
}

}  // namespace acimg

using namespace acimg;

extern "C" {

int acimg_version(void) { return ACIMG_VERSION; }

int acimg_last_error(char* buf, size_t len) {
    if (!buf || len == 0) return ACIMG_EINVAL;
    strncpy(buf, err_buf(), len - 1);
    buf[len - 1] = 0;
    return ACIMG_OK;
}

int acimg_bn_finalize(const float* stats, int rows, int C, int ldstats, double count,
                      const float* gamma, const float* beta, float* moving_mean,
                      float* moving_var, float decay, float eps, int training, float* scale,
                      float* shift, float* save_mean, float* save_invstd, void* stream) {
    if (C <= 0 || (training && (!stats || rows <= 0 || count <= 0)) || (!training && (!moving_mean || !moving_var)))
        return fail(ACIMG_EINVAL, "bn_finalize: bad arguments");
    // (8 channels x 128 row groups for every layer was measured in round 2: 7.6 us per launch against 6.2 us)
    if (training && rows > 1024 && C <= 64)
        hipLaunchKernelGGL(bn_finalize_kernel<8>, dim3(cdiv(C, 8)), dim3(1024), 0, (hipStream_t)stream, stats,
                           rows, C, ldstats, count, gamma, beta, moving_mean, moving_var, decay, eps,
                           training, scale, shift, save_mean, save_invstd);
    else
        hipLaunchKernelGGL(bn_finalize_kernel<32>, dim3(cdiv(C, 32)), dim3(1024), 0, (hipStream_t)stream, stats,
                           rows, C, ldstats, count, gamma, beta, moving_mean, moving_var, decay, eps,
                           training, scale, shift, save_mean, save_invstd);
    return check_launch("bn_finalize");
}

int acimg_bn_add_relu(const float* a, const float* sa, const float* ta, const float* b,
                      const float* sb, const float* tb, float* out, int N, int OH, int OW, int C,
                      int BH, int BW, int bstride, void* stream) {
    if (C & 3) return fail(ACIMG_EINVAL, "bn_add_relu: C must be a multiple of 4");
    const long total4 = (long)N * OH * OW * (C / 4);
    hipLaunchKernelGGL(bn_add_relu_kernel, dim3(ew_grid(total4)), dim3(256), 0, (hipStream_t)stream, a,
                       sa, ta, b, sb, tb, out, total4, OH, OW, C / 4, BH, BW, bstride);
    return check_launch("bn_add_relu");
}

int acimg_bn_relu_maxpool(const float* x, const float* scale, const float* shift, float* out,
                          int N, int H, int W, int C, int OH, int OW, int pad_t, int pad_l,
                          void* stream) {
    if (C & 3) return fail(ACIMG_EINVAL, "bn_relu_maxpool: C must be a multiple of 4");
    const long total4 = (long)N * OH * OW * (C / 4);
    hipLaunchKernelGGL(bn_relu_maxpool_kernel, dim3(ew_grid(total4)), dim3(256), 0, (hipStream_t)stream,
                       x, scale, shift, out, total4, H, W, C / 4, OH, OW, pad_t, pad_l);
    return check_launch("bn_relu_maxpool");
}

int acimg_bn_relu(const float* x, const float* scale, const float* shift, float* y, long rows,
                  int C, int ldx, int ldy, void* stream) {
    hipLaunchKernelGGL(bn_relu_kernel, dim3(ew_grid(rows * C)), dim3(256), 0, (hipStream_t)stream, x,
                       scale, shift, y, rows, C, ldx, ldy);
    return check_launch("bn_relu");
}

int acimg_bn_relu_bwd(const float* x, const float* y, const float* gy, const float* gamma,
                      const float* save_mean, const float* save_invstd, float* gx, float* dgamma,
                      float* dbeta, long rows, int C, void* stream) {
    hipLaunchKernelGGL(bn_relu_bwd_kernel, dim3(C), dim3(1024), 0, (hipStream_t)stream, x, y, gy, gamma,
                       save_mean, save_invstd, gx, dgamma, dbeta, rows, C);
    return check_launch("bn_relu_bwd");
}

int acimg_pad_channels(const float* x, float* y, long pixels, int C, int Cp, void* stream) {
    hipLaunchKernelGGL(pad_channels_kernel, dim3(ew_grid(pixels * Cp)), dim3(256), 0,
                       (hipStream_t)stream, x, y, pixels, C, Cp);
    return check_launch("pad_channels");
}

int acimg_pad_image(const float* x, float* y, int N, int H, int W, int C, int Cp, int Hp, int Wp, int pad_t,
                    int pad_l, void* stream) {
    if (!x || !y || C > Cp || pad_t < 0 || pad_l < 0 || H + pad_t > Hp || W + pad_l > Wp)
        return fail(ACIMG_EINVAL, "pad_image: the image does not fit the padded frame");
    const long pixels = (long)N * H * W;
    if (C == 3 && Cp == 4 && !aligned16(y)) return fail(ACIMG_EINVAL, "pad_image: the frame must be 16-byte aligned");
    hipLaunchKernelGGL(pad_image_kernel, dim3(ew_grid(pixels)), dim3(256), 0, (hipStream_t)stream, x, y, H, W, C, Cp,
                       Hp, Wp, pad_t, pad_l, pixels);
    return check_launch("pad_image");
}

int acimg_tile_mfcc(const float* mfcc, float* out, int N, int HW, int C, void* stream) {
    const long total = (long)N * HW * C;
    hipLaunchKernelGGL(tile_mfcc_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, mfcc,
                       out, HW, C, total);
    return check_launch("tile_mfcc");
}

size_t acimg_minmax_workspace(int N, int P, int C) {
    if (N <= 0 || P <= 0 || C <= 0) return 0;
    return (size_t)N * minmax_chunks(P, C) * 2 * sizeof(float);
}

int acimg_minmax_fwd(const float* x, int ldx, float* out, int ldo, float* mm, int N, int P, int C, void* ws,
                     size_t ws_bytes, void* stream) {
    if (!x || !out || !mm || N <= 0 || P <= 0 || C <= 0) return fail(ACIMG_EINVAL, "minmax_fwd: bad argument");
    if (!ws || ws_bytes < acimg_minmax_workspace(N, P, C))
        return fail(ACIMG_EWORKSPACE, "minmax_fwd: workspace %zu < %zu", ws_bytes, acimg_minmax_workspace(N, P, C));
    const int S = minmax_chunks(P, C), ppc = cdiv(P, S);
    float* part = static_cast<float*>(ws);
    hipLaunchKernelGGL(minmax_part_kernel, dim3(S, N), dim3(256), 0, (hipStream_t)stream, x, ldx, part, mm, P, C, ppc);
    hipLaunchKernelGGL(minmax_apply_kernel, dim3(S, N), dim3(256), 0, (hipStream_t)stream, x, ldx, out, ldo, part,
                       mm, P, C, ppc);
    return check_launch("minmax_fwd");
}

int acimg_minmax_bwd(const float* x, int ldx, const float* go, int ldgo, const float* mm,
                     float* gx, int ldgx, int N, int P, int C, int accumulate, int mask_relu, void* ws,
                     size_t ws_bytes, void* stream) {
    if (!x || !go || !mm || !gx || N <= 0 || P <= 0 || C <= 0) return fail(ACIMG_EINVAL, "minmax_bwd: bad argument");
    if (!ws || ws_bytes < acimg_minmax_workspace(N, P, C))
        return fail(ACIMG_EWORKSPACE, "minmax_bwd: workspace %zu < %zu", ws_bytes, acimg_minmax_workspace(N, P, C));
    const int S = minmax_chunks(P, C), ppc = cdiv(P, S);
    float* part = static_cast<float*>(ws);
    hipLaunchKernelGGL(minmax_bwd_part_kernel, dim3(S, N), dim3(256), 0, (hipStream_t)stream, x, ldx, go, ldgo, mm,
                       part, P, C, ppc);
    hipLaunchKernelGGL(minmax_bwd_kernel, dim3(S, N), dim3(256), 0, (hipStream_t)stream, x, ldx, go, ldgo,
                       mm, part, gx, ldgx, P, C, ppc, accumulate, mask_relu);
    return check_launch("minmax_bwd");
}

int acimg_latent_fwd(const float* heads, const float* eps, float* z, int ldz, float* sigma,
                     float* kl, int N, int Z, void* stream) {
    hipLaunchKernelGGL(latent_fwd_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, heads, eps, z, ldz,
                       sigma, kl, Z);
    return check_launch("latent_fwd");
}

int acimg_latent_bwd(const float* heads, const float* eps, const float* sigma, const float* gz,
                     int ldgz, float kl_weight, float* g_heads, int N, int Z, void* stream) {
    hipLaunchKernelGGL(latent_bwd_kernel, dim3(cdiv((long)N * Z, 256)), dim3(256), 0, (hipStream_t)stream,
                       heads, eps, sigma, gz, ldgz, kl_weight, g_heads, N, Z);
    return check_launch("latent_bwd");
}

int acimg_softplus_fwd(const float* x, int ldx, float* y, int ldy, int rows, int C, void* stream) {
    hipLaunchKernelGGL(softplus_fwd_kernel, dim3(cdiv((long)rows * C, 256)), dim3(256), 0, (hipStream_t)stream, x, ldx, y,
                       ldy, rows, C);
    return check_launch("softplus_fwd");
}

int acimg_softplus_bwd(const float* x, int ldx, const float* gy, int ldgy, float* gx, int ldgx, int rows, int C,
                       void* stream) {
    hipLaunchKernelGGL(softplus_bwd_kernel, dim3(cdiv((long)rows * C, 256)), dim3(256), 0, (hipStream_t)stream, x, ldx, gy,
                       ldgy, gx, ldgx, rows, C);
    return check_launch("softplus_bwd");
}

int acimg_latent_linear_fwd(const float* heads, const float* eps, float* z, int ldz, float* kl, int N, int Z,
                            void* stream) {
    hipLaunchKernelGGL(latent_linear_fwd_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, heads, eps, z, ldz, kl, Z);
    return check_launch("latent_linear_fwd");
}

int acimg_latent_linear_bwd(const float* heads, const float* eps, const float* gz, int ldgz, float kl_weight,
                            float* g_heads, int N, int Z, void* stream) {
    hipLaunchKernelGGL(latent_linear_bwd_kernel, dim3(cdiv((long)N * Z, 256)), dim3(256), 0, (hipStream_t)stream,
                       heads, eps, gz, ldgz, kl_weight, g_heads, N, Z);
    return check_launch("latent_linear_bwd");
}

int acimg_maxpool_fwd(const float* x, int ldx, float* y, int ldy, int N, int H, int W, int C, int k, void* stream) {
    if (k <= 0 || H < k || W < k) return fail(ACIMG_EINVAL, "maxpool_fwd: window %d does not fit %dx%d", k, H, W);
    const int OH = H / k, OW = W / k;
    const long total = (long)N * OH * OW * C;
    hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, x, ldx, y, ldy, H, W,
                       C, OH, OW, k, total);
    return check_launch("maxpool_fwd");
}

int acimg_maxpool_relu_bwd(const float* x, int ldx, const float* gy, int ldgy, float* gx, int ldgx, int N, int H,
                           int W, int C, int k, void* stream) {
    if (k <= 0 || H < k || W < k) return fail(ACIMG_EINVAL, "maxpool_relu_bwd: window %d does not fit %dx%d", k, H, W);
    const long total = (long)N * H * W * C;
    hipLaunchKernelGGL(maxpool_relu_bwd_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, x, ldx, gy, ldgy,
                       gx, ldgx, H, W, C, H / k, W / k, k, total);
    return check_launch("maxpool_relu_bwd");
}

int acimg_spatial_sum(const float* x, int ldx, float* y, int N, int P, int C, void* stream) {
    hipLaunchKernelGGL(spatial_sum_kernel, dim3(N, cdiv(C, 64)), dim3(256), 0, (hipStream_t)stream, x, ldx, y, P, C);
    return check_launch("spatial_sum");
}

int acimg_spatial_sum_relu_bwd(const float* x, int ldx, const float* gy, float* gx, int ldgx, int N, int P, int C,
                               void* stream) {
    const long total = (long)N * P * C;
    hipLaunchKernelGGL(spatial_sum_relu_bwd_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, x, ldx, gy,
                       gx, ldgx, P, C, total);
    return check_launch("spatial_sum_relu_bwd");
}

int acimg_clip_softmax_ce(const float* logits, int ldl, int clips, int F, int K, const int* labels, float* out,
                          float* g_logits, int ldg, void* stream) {
    if (K <= 0 || K > 64 || F <= 0) return fail(ACIMG_EINVAL, "clip_softmax_ce: classes must be in 1..64 (got %d)", K);
    hipLaunchKernelGGL(clip_softmax_ce_kernel, dim3(clips), dim3(64), 0, (hipStream_t)stream, logits, ldl, F, K, labels,
                       clips, out, g_logits, ldg);
    return check_launch("clip_softmax_ce");
}

// partial-sum workgroups of the reduce pass: one per CU (round 4 measured four per CU: 22 launches 1133 -> 1223 us, the
// finalize walks four times the partials)
static constexpr long BN_BWD_BLOCKS = 256;

size_t acimg_bn_bwd_workspace(long rows, int C) {
    long blocks = (rows + 255) / 256;
    if (blocks > BN_BWD_BLOCKS) blocks = BN_BWD_BLOCKS;
    if (blocks < 1) blocks = 1;
    return (size_t)(blocks + 1) * 2 * C * sizeof(float);
}

int acimg_bn_bwd(const float* x, int ldx, const float* gy, int ldgy, const float* scale, const float* shift,
                 const float* save_mean, const float* save_invstd, const float* gamma, long rows, int C,
                 float* gx, int ldgx, float* dgamma, float* dbeta, void* ws, size_t ws_bytes, void* stream) {
    if ((C & 3) || C > 1024 || (ldx & 3) || (ldgy & 3) || (ldgx & 3) || !aligned16(x) || !aligned16(gy) ||
        !aligned16(gx))
        return fail(ACIMG_EINVAL, "bn_bwd: C=%d must be a multiple of 4 (<= 1024), buffers 16-byte aligned", C);
    if (ws_bytes < acimg_bn_bwd_workspace(rows, C) || !ws) return fail(ACIMG_EWORKSPACE, "bn_bwd: workspace too small");
    long blocks = (rows + 255) / 256;
    if (blocks > BN_BWD_BLOCKS) blocks = BN_BWD_BLOCKS;   // the finalize pass walks the partials 8 at a time
    if (blocks < 1) blocks = 1;
    const long rpb = (rows + blocks - 1) / blocks;
    float* partial = static_cast<float*>(ws);
    float* sums = partial + blocks * 2 * C;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3((int)blocks), dim3(256), 256 * 8 * sizeof(float), st, x, ldx, gy, ldgy,
                       scale, shift, save_mean, save_invstd, rows, C, rpb, partial);
    const int cw = C <= 8 ? 8 : (C <= 16 ? 16 : 32);
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(cdiv(C, cw), 2), dim3(256), 0, st, partial, (int)blocks, C, dgamma, dbeta, sums, cw);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(ew_grid(rows * (C / 4))), dim3(256), 0, st, x, ldx, gy, ldgy, scale, shift,
                       save_mean, save_invstd, gamma, sums, rows, C, gx, ldgx);
    return check_launch("bn_bwd");
}

size_t acimg_loss_scratch_bytes(void) { return (size_t)(4 + 2 * 1024) * sizeof(float); }

int acimg_recon_loss(const float* yhat, const float* target, float* g_logit, float* sums,
                     long count, float w_mse, float w_huber, void* scratch, size_t scratch_bytes, void* stream) {
    if (!aligned16(yhat) || !aligned16(target) || (g_logit && !aligned16(g_logit)))
        return fail(ACIMG_EINVAL, "recon_loss: buffers must be 16-byte aligned");
    if (scratch && scratch_bytes < acimg_loss_scratch_bytes())
        return fail(ACIMG_EWORKSPACE, "recon_loss: scratch %zu < %zu", scratch_bytes, acimg_loss_scratch_bytes());
    long blocks = (count / 4 + 255) / 256;
    if (blocks > 160) blocks = 160;      // one ordered hand-off (or two float atomics) per workgroup: keep them few
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(recon_loss_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, yhat,
                       target, g_logit, sums, count, w_mse, w_huber, static_cast<float*>(scratch));
    return check_launch("recon_loss");
}

int acimg_sumsq(const float* x, long n, float* out, void* scratch, size_t scratch_bytes, void* stream) {
    if (!aligned16(x)) return fail(ACIMG_EINVAL, "sumsq: buffer must be 16-byte aligned");
    if (scratch && scratch_bytes < acimg_loss_scratch_bytes())
        return fail(ACIMG_EWORKSPACE, "sumsq: scratch %zu < %zu", scratch_bytes, acimg_loss_scratch_bytes());
    long blocks = (n / 4 + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(sumsq_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, x, n, out,
                       static_cast<float*>(scratch));
    return check_launch("sumsq");
}

int acimg_axpy(float a, const float* x, float* y, long n, void* stream) {
    hipLaunchKernelGGL(axpy_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, a, x, y, n);
    return check_launch("axpy");
}

int acimg_adam_step(float* p, const float* g, float* m, float* v, long n, float lr_t, float beta1,
                    float beta2, float eps, float grad_scale, void* stream) {
    if (!aligned16(p) || !aligned16(g) || !aligned16(m) || !aligned16(v))
        return fail(ACIMG_EINVAL, "adam_step: buffers must be 16-byte aligned");
    hipLaunchKernelGGL(adam_kernel, dim3(ew_grid(n / 4 + 1)), dim3(256), 0, (hipStream_t)stream, p, g, m,
                       v, n, lr_t, beta1, beta2, eps, grad_scale);
    return check_launch("adam_step");
}

int acimg_grad_slice(const float* src, int ldsrc, float* dst, int lddst, const float* mask, int ldmask,
                     long pixels, int C, int accumulate, void* stream) {
    hipLaunchKernelGGL(grad_slice_kernel, dim3(ew_grid(pixels * C)), dim3(256), 0, (hipStream_t)stream, src,
                       ldsrc, dst, lddst, mask, ldmask, pixels, C, accumulate);
    return check_launch("grad_slice");
}

int acimg_loss_finalize(const float* sums, const float* kl, int N, double count, float latent_w,
                        float half_wd, float w_mse, float w_huber, float* out, void* stream) {
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, sums, kl, N, count,
                       latent_w, half_wd, w_mse, w_huber, out);
    return check_launch("loss_finalize");
}

}  // extern "C"
namespace acimg {
__global__ void spin_kernel(unsigned long long cycles, unsigned* out) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned long long t = t0;
    while (t - t0 < cycles) {          // every wave reaches the exit: the clock only moves forward
        __builtin_amdgcn_s_sleep(32);
        t = __builtin_amdgcn_s_memtime();
    }
    if (threadIdx.x == 0 && out) out[0] = (unsigned)(t - t0);
}
}  // namespace acimg
extern "C" {

int acimg_spin(uint64_t cycles, void* out, void* stream) {
    if (cycles > 0xFFFFFFFFull) return fail(ACIMG_EINVAL, "spin: at most 2^32 cycles");
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (unsigned long long)cycles,
                       static_cast<unsigned*>(out));
    return check_launch("spin");
}

int acimg_zero(void* ptr, size_t bytes, void* stream) {
    hipError_t e = hipMemsetAsync(ptr, 0, bytes, (hipStream_t)stream);
    if (e != hipSuccess) return fail(ACIMG_ELAUNCH, "zero: %s", hipGetErrorString(e));
    return ACIMG_OK;
}

int acimg_randn(float* out, long n, uint64_t seed, uint64_t offset, void* stream) {
    if (n <= 0) return fail(ACIMG_EINVAL, "randn: n must be positive");
    hipLaunchKernelGGL(randn_kernel, dim3(ew_grid((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, out, n,
                       seed, offset);
    return check_launch("randn");
}

int acimg_sqerr_channels(const float* a, const float* b, long pixels, int C, float* out, void* stream) {
    if (C <= 0 || C > 64) return fail(ACIMG_EINVAL, "sqerr_channels: C must be in 1..64");
    long blocks = (pixels * C + 255) / 256;
    if (blocks > 512) blocks = 512;
    if (blocks * 256 < C) blocks = (C + 255) / 256;
    hipLaunchKernelGGL(sqerr_channels_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, a, b, pixels,
                       C, out);
    return check_launch("sqerr_channels");
}

size_t acimg_split_plane_bytes(long rows, int C) { return (size_t)((rows + 15) / 16) * 16 * (size_t)C * 2; }

// items of a [rows][C] tensor in brick order (8 pixels x 32 channels per wave), or 0 if it cannot be one
static unsigned split_items(long rows, int C) {
    if (rows <= 0 || C <= 0 || (C & 31)) return 0;
    const long items = (rows + 7) / 8 * 8 * (C / 4);
    return items < (1L << 30) ? (unsigned)items : 0u;
}

int acimg_bn_relu_split(const float* x, const float* scale, const float* shift, int relu, void* out,
                        size_t lo_off, long rows, int C, void* stream) {
    const unsigned items = split_items(rows, C);
    if (!items || (lo_off & 15) || lo_off < acimg_split_plane_bytes(rows, C) || !aligned16(x) || !aligned16(out))
        return fail(ACIMG_EINVAL, "bn_relu_split: C must be a multiple of 32, at most 2^32 elements, buffers aligned, "
                                  "lo_off >= acimg_split_plane_bytes(rows, C)");
    hipLaunchKernelGGL(bn_relu_split_kernel, dim3(ew_grid(items)), dim3(256), 0, (hipStream_t)stream, x, scale,
                       shift, relu, static_cast<char*>(out), (long)lo_off, items, (unsigned)rows, C);
    return check_launch("bn_relu_split");
}

int acimg_bn_add_relu_split(const float* a, const float* sa, const float* ta, const float* b32,
                            const float* sb, const float* tb, const void* b_planes, size_t b_lo_off,
                            void* out_planes, size_t out_lo_off, float* out32, int N, int OH, int OW, int C,
                            int BH, int BW, int bstride, void* stream) {
    const long rows = (long)N * OH * OW;
    const unsigned items = split_items(rows, C);
    if (!items) return fail(ACIMG_EINVAL, "bn_add_relu_split: C must be a multiple of 32, at most 2^32 elements");
    if ((b32 == nullptr) == (b_planes == nullptr))
        return fail(ACIMG_EINVAL, "bn_add_relu_split: exactly one of b32 / b_planes");
    if (b32 && (!sb || !tb)) return fail(ACIMG_EINVAL, "bn_add_relu_split: projection shortcut needs scale/shift");
    if (out_planes && out_lo_off < acimg_split_plane_bytes(rows, C))
        return fail(ACIMG_EINVAL, "bn_add_relu_split: out_lo_off < acimg_split_plane_bytes(rows, C)");
    if (b_planes && b_lo_off < acimg_split_plane_bytes((long)N * BH * BW, C))
        return fail(ACIMG_EINVAL, "bn_add_relu_split: b_lo_off < acimg_split_plane_bytes of the shortcut tensor");
    // one float4 per thread (no grid-stride loop): measured 4 % faster than 4096 persistent workgroups on this
    // three-stream pass (16 M float4 at the 56x75x512 stage)
    const long nblk = ((long)items + 255) / 256;
    hipLaunchKernelGGL(bn_add_relu_split_kernel, dim3((unsigned)(nblk < (1L << 22) ? nblk : (1L << 22))), dim3(256), 0, (hipStream_t)stream, a, sa,
                       ta, b32, sb, tb, static_cast<const char*>(b_planes), (long)b_lo_off,
                       static_cast<char*>(out_planes), (long)out_lo_off, out32, items, (unsigned)rows, OH, OW, C, BH, BW,
                       bstride);
    return check_launch("bn_add_relu_split");
}

int acimg_bn_relu_maxpool_split(const float* x, const float* scale, const float* shift, void* out,
                                size_t lo_off, int N, int H, int W, int C, int OH, int OW, int pad_t, int pad_l,
                                void* stream) {
    const long rows = (long)N * OH * OW;
    const unsigned items = split_items(rows, C);
    if (!items || lo_off < acimg_split_plane_bytes(rows, C))
        return fail(ACIMG_EINVAL, "bn_relu_maxpool_split: C must be a multiple of 32, lo_off >= acimg_split_plane_bytes");
    hipLaunchKernelGGL(bn_relu_maxpool_split_kernel, dim3(ew_grid(items)), dim3(256), 0, (hipStream_t)stream, x,
                       scale, shift, static_cast<char*>(out), (long)lo_off, items, (unsigned)rows, H, W, C, OH, OW, pad_t,
                       pad_l);
    return check_launch("bn_relu_maxpool_split");
}

}  // extern "C"
