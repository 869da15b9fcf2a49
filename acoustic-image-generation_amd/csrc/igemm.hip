// fp32 implicit-GEMM convolution kernels for gfx950 (MI355X), exact-f32 MFMA
// (v_mfma_f32_16x16x4_f32: bit-for-bit an fmaf chain, 64 FLOP/clk/SIMD).
//
// One kernel template covers every conv-shaped op of the acoustic-image train step:
//   forward conv / strided conv / 1x1 / dense        A = im2col(x),  B = W[k][n]      (NN)
//   data gradient (stride 1)                         A = im2col(gy), B = W^T flipped  (NT)
//   kernel==stride patch ops (deconv fwd, pool dgrad)A = pixels,     B = W rows, scatter epilogue
// and a second template does the weight gradient dW = im2col(x)^T * gy (TN, split over pixels).
//
// Tiling: 256 threads = 4 waves, block tile BM x BN x BK(=32), wave tile (BM/WGM) x (BN/WGN) made
// of 16x16 MFMA tiles.  A is staged [row][k] (k contiguous, ds_read_b128 gives 4 k per lane), B is
// staged [k][n] (NN, ds_read_b32) or [n][k] (NT, ds_read_b128).  The k order inside a 16-deep step
// is permuted identically for A and B (lane group g owns k = 4g..4g+3), which leaves the sum
// unchanged.  Global->register prefetch of tile t+1 overlaps the MFMAs of tile t.
#include "wgrad_split3_kernel.hpp"
#include "igemm_split3d_kernel.hpp"
#include "igemm_split3dp_kernel.hpp"
#include "igemm_split3r_kernel.hpp"
#include "igemm_split3h_kernel.hpp"
#include "skinny_kernel.hpp"
#include <cstdlib>
#include <algorithm>
#include <type_traits>

namespace acimg {

// split-K reducer: sums the slabs and runs the epilogue
__global__ __launch_bounds__(256) void igemm_splitk_reduce_kernel(const float* slab, int splits,
                                                                  int M, int Ngemm, int slab_ld,
                                                                  const EpiParams e) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)M * Ngemm;
    if (idx >= total) return;
    const int m = (int)(idx / Ngemm);
    const int n = (int)(idx - (long)m * Ngemm);
    float v = 0.f;
    for (int z = 0; z < splits; ++z) v += slab[((long)z * M + m) * slab_ld + n];
    if (n >= e.Nstore) return;
    int cn, pixoff, r, q, oh, ow;
    epi_col(e, n, cn, pixoff, r, q);
    const long rp = epi_row_pix(e, m, oh, ow);
    if (epi_lands(e, oh, ow, r, q)) epi_store(e, rp + pixoff, cn, v);
}

// ------------------------------------------------------------------------------------------
// SUB-PIXEL form of the stride-2 transposed convolutions with overlapping taps (round 4): the transposed conv
// `conv2d_transpose(k > 2, s = 2)` (models/unet_architecture.py:192-206: upconv_2D with (2,3) kernels) and the data
// gradient of a stride-2 conv (the strided "pool" convs, :168-176).  Both compute
//     Y[s i + a + oy0][s j + b + ox0][ko] = sum_{u, v, kin} A[i - u][j - v][kin] * w[a + 2 u][b + 2 v][ko][kin]
// i.e. an output pixel of parity class (a, b) only sees the kernel taps of its class: ceil(R/2) x ceil(S/2) of them.
// Round 1 ran these as a stride-1 correlation over a ZERO-INSERTED copy of A (4x the pixels, 3 of 4 products against
// zeros, plus the copy's 4x write and read).  Here the four classes are the column groups of ONE implicit GEMM over A's own
// grid - rows (i, j), K = (u', v', kin) with a ceil(R/2) x ceil(S/2) gather, columns (a, b, ko) - whose epilogue scatters
// element (i, j, a, b, ko) to pixel (2 i + a + oy0, 2 j + b + ox0): the k <= s scatter of the non-overlapping transposed
// convs, bounds-checked (EpiParams::scatter = 2).  The combined weight matrix is gathered from the layer's kernel by
// `subpixel_weights_kernel` (the kernels change every step).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void subpixel_weights_kernel(const float* w, int R, int S, int Ko, int Kin, int ldw, int U,
                                                               int V, float* wc, int total) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int ncol = 4 * Ko;
    const int row = idx / ncol, col = idx - row * ncol;
    const int kin = row % Kin, tap = row / Kin;
    const int up = tap / V, vp = tap - up * V;
    const int cls = col / Ko, ko = col - cls * Ko;
    const int a = cls >> 1, b = cls & 1;
    const int r = a + 2 * (U - 1 - up), q = b + 2 * (V - 1 - vp);
    wc[idx] = (r < R && q < S) ? w[((long)(r * S + q) * Ko + ko) * ldw + kin] : 0.f;
}

// ------------------------------------------------------------------------------------------
// weight gradient kernel: dW[kk][n] = sum_m A(m,kk) G[m][n]
// ------------------------------------------------------------------------------------------
// WGM x WGN = 4 waves: 2 x 2, or 4 x 1 for the 16-column tile of the few-column problems (N <= 16: conv_map's 12
// outputs, the 8-channel layers of the RGB / spectrogram U-Nets) where a 32-wide tile would be mostly padding
template <int BMO, int BN, int WGM = 2>
__global__ __launch_bounds__(256) void wgrad_f32_kernel(const WgradParams p) {
    constexpr int BKR = 16;  // pixels per step
    constexpr int LDA_S = BMO + 4;
    constexpr int LDB_S = BN + 4;
    constexpr int WGN = 4 / WGM;
    constexpr int WTM = BMO / WGM, WTN = BN / WGN;
    static_assert(WTM % 16 == 0 && WTN % 16 == 0, "wave tile");
    constexpr int TM = WTM / 16, TN = WTN / 16;
    constexpr int AQ = BMO / 4;               // float4 per A row
    constexpr int ARPP = 256 / AQ;            // A rows per pass
    constexpr int NA = (BKR + ARPP - 1) / ARPP;
    constexpr int BQ = BN / 4;
    constexpr int BRPP = 256 / BQ;
    constexpr int NB = (BKR + BRPP - 1) / BRPP;

    __shared__ __attribute__((aligned(16))) float As[BKR * LDA_S];
    __shared__ __attribute__((aligned(16))) float Bs[BKR * LDB_S];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WGN, wn = wid % WGN;
    const int li = lane & 15, g = lane >> 4;
    const int kk0 = blockIdx.x * BMO, n0 = blockIdx.y * BN;

    const int m_begin = blockIdx.z * p.rows_per_split;
    const int m_end = min(p.M, m_begin + p.rows_per_split);

    // this thread's fixed A column (kk -> tap, c)
    const int aq = tid % AQ;
    const int arow0 = tid / AQ;
    const int kk = kk0 + aq * 4;
    const bool kk_ok = kk < p.KK;
    const bool kk_ones = p.db_out != nullptr && kk == p.KK;   // the all-ones column (bias gradient)
    int r = 0, s = 0, c = 0;
    if (kk_ok) {
        const int tap = kk / p.C;
        c = kk - tap * p.C;
        r = tap / p.S;
        s = tap - r * p.S;
    }
    const int bq = tid % BQ;
    const int brow0 = tid / BQ;
    const int nb = n0 + bq * 4;
    const bool nb_ok = nb < p.Nld;
    const int ohw = p.OH * p.OW;

    // pixel coordinates of this thread's A rows, advanced incrementally by BKR pixels per step (two integer
    // divisions per load were the bulk of this kernel's time on the few-channel layers: no hardware divider)
    int x_img[NA], x_oh[NA], x_ow[NA];
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        const int m = m_begin + arow0 + j * ARPP;
        x_img[j] = m / ohw;
        const int rem = m - x_img[j] * ohw;
        x_oh[j] = rem / p.OW;
        x_ow[j] = rem - x_oh[j] * p.OW;
    }

    float4 ra[NA], rb[NB];
    auto load_tiles = [&](int mb) {
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            const int row = arow0 + j * ARPP;
            const int m = mb + row;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < BKR && kk_ones && m < m_end) v.x = 1.f;
            if (row < BKR && kk_ok && m < m_end) {
                const int ih = x_oh[j] * p.stride - p.pad_t + r;
                const int iw = x_ow[j] * p.stride - p.pad_l + s;
                if ((unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W)
                    v = *reinterpret_cast<const float4*>(
                        p.X + ((long)(x_img[j] * p.H + ih) * p.W + iw) * p.ldx + c);
            }
            ra[j] = v;
            x_ow[j] += BKR;                    // this row slot moves BKR pixels ahead for the next call
            while (x_ow[j] >= p.OW) {
                x_ow[j] -= p.OW;
                if (++x_oh[j] == p.OH) {
                    x_oh[j] = 0;
                    ++x_img[j];
                }
            }
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int row = brow0 + j * BRPP;
            const int m = mb + row;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < BKR && nb_ok && m < m_end)
                v = *reinterpret_cast<const float4*>(p.G + (long)m * p.ldg + nb);
            rb[j] = v;
        }
    };
    auto store_tiles = [&]() {
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            const int row = arow0 + j * ARPP;
            if (row < BKR) *reinterpret_cast<float4*>(&As[row * LDA_S + aq * 4]) = ra[j];
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int row = brow0 + j * BRPP;
            if (row < BKR) *reinterpret_cast<float4*>(&Bs[row * LDB_S + bq * 4]) = rb[j];
        }
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (m_begin < m_end) {
        load_tiles(m_begin);
        store_tiles();
    }
    __syncthreads();
    for (int mb = m_begin; mb < m_end; mb += BKR) {
        const bool more = (mb + BKR) < m_end;
        if (more) load_tiles(mb + BKR);
        float af[TM][4], bf[TN][4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i][t] = As[(4 * g + t) * LDA_S + wm * WTM + i * 16 + li];
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j][t] = Bs[(4 * g + t) * LDB_S + wn * WTN + j * 16 + li];
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][t], bf[j][t], acc[i][j], 0, 0, 0);
        __syncthreads();
        if (more) {
            store_tiles();
            __syncthreads();
        }
    }

    float* out = p.out + (long)blockIdx.z * p.KK * p.ldo;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            const int row = kk0 + wm * WTM + i * 16 + g * 4 + rg;
            if (row > p.KK || (row == p.KK && p.db_out == nullptr)) continue;
            float* dst = row < p.KK ? out + (long)row * p.ldo : p.db_out + (long)blockIdx.z * p.ldo;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + wn * WTN + j * 16 + li;
                if (n < p.Ngemm) dst[n] = acc[i][j][rg];
            }
        }
}

// sums `splits` slabs of [rows][ld] (only cols < ncols) into out[rows][ld]; workgroups >= nb1 do the same for a
// second, one-row job (the bias gradient of the same launch: one reduce launch per weight gradient, not two)
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* slab, int splits, long rows,
                                                          int ncols, int ld, float* out, int nb1,
                                                          const float* slab2, float* out2) {
    long bid = blockIdx.x;
    if (bid >= nb1) {
        bid -= nb1;
        slab = slab2;
        out = out2;
        rows = 1;
    }
    const long idx = bid * 256 + threadIdx.x;
    if (idx >= rows * ncols) return;
    const long row = idx / ncols;
    const int col = (int)(idx - row * ncols);
    float v = 0.f;
    for (int z = 0; z < splits; ++z) v += slab[((long)z * rows + row) * ld + col];
    out[row * ld + col] = v;
}

// the same for many slabs (few-channel weight gradients use up to 2048 pixel splits): OUTS outputs per workgroup, 256 / OUTS
// lane groups walk the slabs with four loads in flight each, fixed-order combine -> deterministic.  OUTS = 32 for large
// gradients; 8 when there are few outputs (a 9 x 8 x 8 kernel is 576 numbers: 18 workgroups of 32 outputs each walked 512
// slabs in 16 dependent rounds - 8 to 20 us of pure latency per launch; 72 workgroups with 32 lane groups need 4 rounds)
template <int OUTS>
__global__ __launch_bounds__(256) void slab_reduce_wide_kernel(const float* slab, int splits, long rows,
                                                               int ncols, int ld, float* out, int nb1,
                                                               const float* slab2, float* out2) {
    constexpr int NG = 256 / OUTS;
    __shared__ float red[NG][OUTS];
    const int ol = threadIdx.x % OUTS, rg = threadIdx.x / OUTS;
    long bid = blockIdx.x;
    if (bid >= nb1) {                  // second job: the one-row bias gradient
        bid -= nb1;
        slab = slab2;
        out = out2;
        rows = 1;
    }
    const long idx = bid * OUTS + ol;
    const bool ok = idx < rows * ncols;
    const long row = ok ? idx / ncols : 0;
    const int col = ok ? (int)(idx - row * ncols) : 0;
    const float* src = slab + row * ld + col;
    const long zs = rows * ld;
    float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
    if (ok) {
        int z = rg;
        for (; z + 3 * NG < splits; z += 4 * NG) {
            v0 += src[(long)z * zs];
            v1 += src[(long)(z + NG) * zs];
            v2 += src[(long)(z + 2 * NG) * zs];
            v3 += src[(long)(z + 3 * NG) * zs];
        }
        for (; z < splits; z += NG) v0 += src[(long)z * zs];
    }
    red[rg][ol] = (v0 + v1) + (v2 + v3);
    __syncthreads();
    if (rg == 0 && ok) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < NG; ++i) t += red[i][ol];
        out[row * ld + col] = t;
    }
}
// slab [splits][rows][ld] -> out [rows][ld] (ncols of them); optionally slab2 [splits][ld] -> out2 [ncols] (bias gradient)
static void launch_slab_reduce_wide(const float* slab, int splits, long rows, int ncols, int ld, float* out, const float* slab2,
                                    float* out2, hipStream_t st) {
    const long total = rows * ncols;
    if (total <= 4096) {
        const int nb1 = (int)cdiv(total, 8), nb2 = out2 ? cdiv(ncols, 8) : 0;
        hipLaunchKernelGGL(slab_reduce_wide_kernel<8>, dim3(nb1 + nb2), dim3(256), 0, st, slab, splits, rows, ncols, ld, out, nb1,
                           slab2, out2);
    } else {
        const int nb1 = (int)cdiv(total, 32), nb2 = out2 ? cdiv(ncols, 32) : 0;
        hipLaunchKernelGGL(slab_reduce_wide_kernel<32>, dim3(nb1 + nb2), dim3(256), 0, st, slab, splits, rows, ncols, ld, out, nb1,
                           slab2, out2);
    }
}

// ------------------------------------------------------------------------------------------
// "tap GEMM" form of a stride-1 VALID convolution with FEW output channels (conv_map: 3x4, 2048 -> 12):
// as an implicit GEMM its N is one MFMA column and every output row gathers R*S*C inputs (no reuse across N:
// L2-bound); instead Z[input pixel][tap*K + k] = X[input pixel][:] . W[tap][:][k] is ONE plain GEMM with
// N = R*S*K columns that reads X once, and y[oh][ow][k] = sum_taps Z[oh+r][ow+s][tap*K + k] is a tiny gather.
// The weight gradient is the same GEMM transposed: dWt = X^T . GZ with GZ the tap-scattered output gradient.
// ------------------------------------------------------------------------------------------
// wt[c][tap*K + k] = w[tap][c][k]
__global__ __launch_bounds__(256) void tapconv_pack_kernel(const float* w, int taps, int C, int K, int ldw,
                                                           float* wt, int ldwt) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const int TK = taps * K;
    if (idx >= (long)C * TK) return;
    const int c = (int)(idx / TK), n = (int)(idx - (long)c * TK);
    const int tap = n / K, k = n - tap * K;
    wt[(long)c * ldwt + n] = w[((long)tap * C + c) * ldw + k];
}

// dw[tap][c][k] = dwt[c][tap*K + k] + decay * w[tap][c][k]
__global__ __launch_bounds__(256) void tapconv_unpack_kernel(const float* dwt, int ldwt, int taps, int C, int K,
                                                             int ldw, const float* w, float decay, float* dw) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)taps * C * K) return;
    const int k = (int)(idx % K);
    const long tc = idx / K;
    const int c = (int)(tc % C), tap = (int)(tc / C);
    float v = dwt[(long)c * ldwt + tap * K + k];
    if (w) v = fmaf(decay, w[tc * ldw + k], v);
    dw[tc * ldw + k] = v;
}

constexpr int TG_PPB = 32;     // output pixels per workgroup of the gather (= rows per batch-norm partial)
// y[(n,oh,ow)][k] = sum_{r,s} z[(n,oh+r,ow+s)][(r*S+s)*K + k]; TG_PPB output pixels per workgroup, one thread per
// (pixel, k); optional batch-norm partials stats[block][2][stats_ld] (rows past the end count as zeros)
__global__ __launch_bounds__(256) void tapconv_gather_kernel(const float* z, int ldz, int H, int W, int R, int S,
                                                             int K, int OH, int OW, long Mout, float* y, int ldy,
                                                             float* stats, int stats_ld) {
    extern __shared__ __attribute__((aligned(16))) float tg_smem[];     // [128][K] tile of outputs
    const int ppb = TG_PPB;
    const long m0 = (long)blockIdx.x * ppb;
    for (int e = threadIdx.x; e < ppb * K; e += 256) {
        const int pl = e / K, k = e - pl * K;
        const long m = m0 + pl;
        float acc = 0.f;
        if (m < Mout) {
            const int ow = (int)(m % OW);
            const long t = m / OW;
            const int oh = (int)(t % OH);
            const long img = t / OH;
            for (int r = 0; r < R; ++r)
                for (int q = 0; q < S; ++q)
                    acc += z[((img * H + oh + r) * W + ow + q) * ldz + (r * S + q) * K + k];
            y[m * ldy + k] = acc;
        }
        tg_smem[e] = acc;
    }
    if (!stats) return;
    __syncthreads();
    if ((int)threadIdx.x < 2 * K) {
        const int which = threadIdx.x / K, k = threadIdx.x - which * K;
        float sum = 0.f;
        for (int pl = 0; pl < ppb; ++pl) {
            const float v = tg_smem[pl * K + k];
            sum += which ? v * v : v;
        }
        stats[((long)blockIdx.x * 2 + which) * stats_ld + k] = sum;
    }
}

// gz[(n,ih,iw)][(r*S+s)*K + k] = gy[(n,ih-r,iw-s)][k] (0 outside the output)
__global__ __launch_bounds__(256) void tapconv_scatter_kernel(const float* gy, int ldgy, int H, int W, int R, int S,
                                                              int K, int OH, int OW, long Min, float* gz, int ldgz) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const int TK = R * S * K;
    if (idx >= Min * TK) return;
    const long pix = idx / TK;
    const int n = (int)(idx - pix * TK);
    const int tap = n / K, k = n - tap * K;
    const int r = tap / S, q = tap - r * S;
    const int iw = (int)(pix % W);
    const long t = pix / W;
    const int ih = (int)(t % H);
    const long img = t / H;
    const int oh = ih - r, ow = iw - q;
    float v = 0.f;
    if ((unsigned)oh < (unsigned)OH && (unsigned)ow < (unsigned)OW) v = gy[((img * OH + oh) * OW + ow) * ldgy + k];
    gz[pix * ldgz + n] = v;
}

// column sums of G[rows][ld] (cols < ncols) -> out[ncols]; one block per 64 columns, 256 threads
// = 4 row-groups x 64 columns.
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* G, long rows, int ncols,
                                                             int ld, long rows_per_block,
                                                             float* partial /*[gridDim.y][ncols]*/) {
    __shared__ float red[4][64];
    const int col = blockIdx.x * 64 + (threadIdx.x & 63);
    const int rg = threadIdx.x >> 6;
    const long rb = (long)blockIdx.y * rows_per_block;
    const long re = min(rows, rb + rows_per_block);
    float s = 0.f;
    if (col < ncols)
        for (long r = rb + rg; r < re; r += 4) s += G[r * ld + col];
    red[rg][threadIdx.x & 63] = s;
    __syncthreads();
    if (rg == 0 && col < ncols)
        partial[(long)blockIdx.y * ncols + col] = red[0][threadIdx.x] + red[1][threadIdx.x] +
                                                  red[2][threadIdx.x] + red[3][threadIdx.x];
}
// narrow variant (ncols <= 64, multiple of 4, ld % 4 == 0): a workgroup sweeps rows_per_block rows with float4
// loads, (ncols/4) lanes per row; partial sums are combined in lane order -> deterministic
__global__ __launch_bounds__(256) void colsum_narrow_kernel(const float* G, long rows, int ncols, int ld,
                                                            long rows_per_block, float* partial) {
    __shared__ float sm[256 * 4];
    const int c4n = ncols >> 2;
    const int tc = threadIdx.x % c4n, tr = threadIdx.x / c4n;
    const int rstep = 256 / c4n;
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    if (tr < rstep) {
        const long r0 = (long)blockIdx.x * rows_per_block;
        const long r1 = min(rows, r0 + rows_per_block);
        // four rows in flight per thread, sums of their own, combined in a fixed order
        float b[3][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        long r = r0 + tr;
        for (; r + 3L * rstep < r1; r += 4L * rstep) {
            const float4 v0 = *reinterpret_cast<const float4*>(G + r * ld + tc * 4);
            const float4 v1 = *reinterpret_cast<const float4*>(G + (r + rstep) * ld + tc * 4);
            const float4 v2 = *reinterpret_cast<const float4*>(G + (r + 2L * rstep) * ld + tc * 4);
            const float4 v3 = *reinterpret_cast<const float4*>(G + (r + 3L * rstep) * ld + tc * 4);
            a[0] += v0.x; a[1] += v0.y; a[2] += v0.z; a[3] += v0.w;
            b[0][0] += v1.x; b[0][1] += v1.y; b[0][2] += v1.z; b[0][3] += v1.w;
            b[1][0] += v2.x; b[1][1] += v2.y; b[1][2] += v2.z; b[1][3] += v2.w;
            b[2][0] += v3.x; b[2][1] += v3.y; b[2][2] += v3.z; b[2][3] += v3.w;
        }
        for (; r < r1; r += rstep) {
            const float4 v = *reinterpret_cast<const float4*>(G + r * ld + tc * 4);
            a[0] += v.x; a[1] += v.y; a[2] += v.z; a[3] += v.w;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) a[k] = (a[k] + b[0][k]) + (b[1][k] + b[2][k]);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) sm[threadIdx.x * 4 + k] = a[k];
    __syncthreads();
    if (threadIdx.x < c4n) {
        float t[4] = {0.f, 0.f, 0.f, 0.f};
        for (int j = 0; j < rstep; ++j)
#pragma unroll
            for (int k = 0; k < 4; ++k) t[k] += sm[(j * c4n + threadIdx.x) * 4 + k];
#pragma unroll
        for (int k = 0; k < 4; ++k) partial[(long)blockIdx.x * ncols + threadIdx.x * 4 + k] = t[k];
    }
}
// 256 threads = 4 part groups x 64 columns - or, for up to 16 columns (where the 4-group form is 64 dependent loads per
// thread, ~13 us of latency for a kilobyte of result), 16 part groups x 16 columns; fixed-order combine
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* partial, int parts, int ncols, float* out) {
    __shared__ float red[16][64];
    const bool narrow = ncols <= 16;
    const int cw = narrow ? 16 : 64, ng = 256 / cw;
    const int cl = threadIdx.x % cw, pg = threadIdx.x / cw;
    const int col = blockIdx.x * cw + cl;
    float s0 = 0.f, s1 = 0.f;
    if (col < ncols) {
        int i = pg;
        for (; i + ng < parts; i += 2 * ng) {
            s0 += partial[(long)i * ncols + col];
            s1 += partial[(long)(i + ng) * ncols + col];
        }
        if (i < parts) s0 += partial[(long)i * ncols + col];
    }
    red[pg][cl] = s0 + s1;
    __syncthreads();
    if (pg == 0 && col < ncols) {
        float t = 0.f;
        for (int g = 0; g < ng; ++g) t += red[g][cl];
        out[col] = t;
    }
}

// per-256-row-block column sums / sums of squares of y[M][K] (pixel stride ldy) -> stats[blk][2][ld]:
// the batch-norm partials for convs that ran split-K (their epilogue never sees a full accumulator)
__global__ __launch_bounds__(256) void partial_stats_kernel(const float* y, int ldy, int M, int K,
                                                            float* stats, int ld, int rows_per_block) {
    __shared__ float red[2][8][32];
    const int cl = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const int r0 = blockIdx.x * rows_per_block;
    const int r1 = min(M, r0 + rows_per_block);
    for (int cb = 0; cb < K; cb += 32) {
        const int c = cb + cl;
        float s1 = 0.f, s2 = 0.f;
        if (c < K) {
            // four rows in flight per thread (one at a time this pass streamed 17 MB in 33 us)
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, b0 = 0.f, b1 = 0.f, b2 = 0.f, b3 = 0.f;
            int r = r0 + rg;
            for (; r + 24 < r1; r += 32) {
                const float v0 = y[(long)r * ldy + c], v1 = y[(long)(r + 8) * ldy + c];
                const float v2 = y[(long)(r + 16) * ldy + c], v3 = y[(long)(r + 24) * ldy + c];
                a0 += v0; b0 += v0 * v0;
                a1 += v1; b1 += v1 * v1;
                a2 += v2; b2 += v2 * v2;
                a3 += v3; b3 += v3 * v3;
            }
            for (; r < r1; r += 8) {
                const float v = y[(long)r * ldy + c];
                a0 += v;
                b0 += v * v;
            }
            s1 = (a0 + a1) + (a2 + a3);
            s2 = (b0 + b1) + (b2 + b3);
        }
        red[0][rg][cl] = s1;
        red[1][rg][cl] = s2;
        __syncthreads();
        if (rg == 0 && c < K) {
            float a = 0.f, b = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                a += red[0][i][cl];
                b += red[1][i][cl];
            }
            stats[((long)blockIdx.x * 2 + 0) * ld + c] = a;
            stats[((long)blockIdx.x * 2 + 1) * ld + c] = b;
        }
        __syncthreads();
    }
}

// writes bias to the output positions of a kernel<stride transposed conv that no patch covers
__global__ __launch_bounds__(256) void deconv_gap_fill_kernel(float* y, int ldy, const float* bias,
                                                              long pixels, int OH, int OW, int K,
                                                              int R, int S, int stride) {
    const int k4 = K / 4;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= pixels * k4) return;
    const long pix = idx / k4;
    const int c = (int)(idx - pix * k4) * 4;
    const int ox = (int)(pix % OW);
    const int oy = (int)((pix / OW) % OH);
    if ((oy % stride) < R && (ox % stride) < S) return;
    float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
    if (bias) b = *reinterpret_cast<const float4*>(bias + c);
    *reinterpret_cast<float4*>(y + pix * ldy + c) = b;
}

// zero insertion: out[n, h*s, w*s, :] = in[n, h, w, :], zeros elsewhere; out is [N][(H-1)s+1][(W-1)s+1][C].
// Turns the data gradient of a stride-s conv (and the forward of a transposed conv whose kernel exceeds
// its stride) into a stride-1 correlation over a 4x larger, mostly-zero tensor: used only for the small
// stride-2 layers of the RGB / spectrogram U-Nets.
__global__ __launch_bounds__(256) void dilate2d_kernel(const float* in, int ldin, float* out, long opixels,
                                                       int H, int W, int OH, int OW, int C, int s) {
    const int c4 = C / 4;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= opixels * c4) return;
    const long pix = idx / c4;
    const int c = (int)(idx - pix * c4) * 4;
    const int ow = (int)(pix % OW);
    const int oh = (int)((pix / OW) % OH);
    const long n = pix / ((long)OW * OH);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (oh % s == 0 && ow % s == 0)
        v = *reinterpret_cast<const float4*>(in + ((n * H + oh / s) * W + ow / s) * ldin + c);
    *reinterpret_cast<float4*>(out + pix * C + c) = v;
}

// ------------------------------------------------------------------------------------------
// Direct convolution for FEW-CHANNEL layers (C*K <= 512: the 4/8/16-channel full-resolution layers of the RGB /
// spectrogram U-Nets, models/unet_architecture.py:55-60,78-85).  There the implicit GEMM is a bad fit: a
// workgroup runs 3 K steps on tiles that are mostly padding and never amortises its prologue.  Here a lane owns
// one output pixel and 8 output channels, walks the taps with 16-byte loads (neighbouring lanes hit the same
// lines) and takes the weights as wave-uniform LDS broadcasts: 8 FMAs per input value, the work is VALU- and
// HBM-shaped.  mode 0: forward, weights HWIO w[tap][c][k]; mode 1: stride-1 data gradient read as a forward
// conv over gy with flipped taps, weights w[ntaps-1-tap][kout][cin].
// ------------------------------------------------------------------------------------------
struct DirectParams {
    const float* x; int ldx, H, W, C;
    float* y; int ldy, OH, OW, K;
    int R, S, stride, pad_t, pad_l;
    const float* w; int ldw, mode, wrows;   // wrows: rows per tap of the weight tensor (C fwd, Kout dgrad)
    const float* bias; int act;
    const float* res; int ldres;
    float* stats; int stats_ld;   // optional: per-256-pixel-block (sum, sum^2) of conv + bias, [blocks][2][stats_ld]
    long M;
};

// weights -> [K/8][ntaps][C][8] (8 consecutive output channels innermost), zero beyond K
__global__ __launch_bounds__(256) void direct_prepare_kernel(const DirectParams p, float* wprep, int total) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int ntaps = p.R * p.S;
    const int k = i & 7, c = (i >> 3) % p.C, tap = ((i >> 3) / p.C) % ntaps, kg = ((i >> 3) / p.C) / ntaps;
    const int ko = kg * 8 + k;
    float v = 0.f;
    if (ko < p.K)
        v = p.mode == 0 ? p.w[((long)tap * p.wrows + c) * p.ldw + ko]
                        : p.w[((long)(ntaps - 1 - tap) * p.wrows + ko) * p.ldw + c];
    wprep[i] = v;
}

// TR, TS, TC > 0: compile-time kernel extent / channel count (the tap and channel loops unroll completely: all the
// pixel loads of a thread are in flight together and the weights arrive as batched scalar loads); 0: run-time
template <int TR, int TS, int TC>
__global__ __launch_bounds__(256) void direct_conv_kernel(const DirectParams p, const float* __restrict__ wprep) {
    // the weight addresses below are wave-uniform: they become scalar loads (s_load_dwordx8), the FMAs take the
    // weights from SGPRs, no LDS and no vector-memory traffic for them
    const int R = TR ? TR : p.R, S = TS ? TS : p.S, C = TC ? TC : p.C;
    const int ntaps = R * S;
    // XCD-aware order: the dispatcher deals workgroups to the 8 XCDs round-robin, and a 256-pixel block shares its input
    // rows with the blocks one image row above and below (and with the other output-channel groups of its own pixels).
    // Dealt out in launch order those neighbours sit behind three different L2s and every input row is fetched three
    // times (counters: 243 MB read per launch for a 68 MB input); here XCD j walks the contiguous range
    // [j * per, (j + 1) * per) of (pixel block, channel group) pairs, channel groups innermost.
    const int ny = (p.K + 7) >> 3;
    const long total = ((p.M + 255) >> 8) * ny, per = (total + 7) >> 3;
    const long unit = (long)(blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if (unit >= total) return;
    const long bx = unit / ny;
    const int by = (int)(unit - bx * ny);
    const int kg = by * 8;
    const float* __restrict__ wl = wprep + (long)by * ntaps * C * 8;
    const long m_raw = bx * 256 + threadIdx.x;
    const bool live = m_raw < p.M;
    const long m = live ? m_raw : p.M - 1;         // dead lanes recompute the last pixel and contribute nothing
    const int ow = (int)(m % p.OW);
    const long t = m / p.OW;
    const int oh = (int)(t % p.OH);
    const long n = t / p.OH;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 acc2[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) acc2[k] = f32x2{0.f, 0.f};
    const int ih0 = oh * p.stride - p.pad_t, iw0 = ow * p.stride - p.pad_l;
    const float* const img = p.x + n * p.H * p.W * p.ldx;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int ih = ih0 + r;
        const int ihc = min(max(ih, 0), p.H - 1);
#pragma unroll
        for (int q = 0; q < S; ++q) {
            const int iw = iw0 + q;
            const bool ok = (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
            // clamped address: the load is unconditional (and can be hoisted), padding taps are zeroed by select
            const float* src = img + ((long)ihc * p.W + min(max(iw, 0), p.W - 1)) * p.ldx;
            const float* __restrict__ wt = wl + (r * S + q) * C * 8;
#pragma unroll
            for (int c = 0; c < C; c += 4) {
                float4 xv = *reinterpret_cast<const float4*>(src + c);
                const float xs[4] = {ok ? xv.x : 0.f, ok ? xv.y : 0.f, ok ? xv.z : 0.f, ok ? xv.w : 0.f};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    // packed fp32 FMAs (v_pk_fma_f32: two accumulators per instruction, same rounding as fmaf)
                    const f32x2 xx = {xs[i], xs[i]};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const f32x2 ww = *reinterpret_cast<const f32x2*>(wt + (c + i) * 8 + 2 * j);
                        acc2[j] = __builtin_elementwise_fma(xx, ww, acc2[j]);
                    }
                }
            }
        }
    }
    float acc[8];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        acc[2 * k] = acc2[k][0];
        acc[2 * k + 1] = acc2[k][1];
    }
    if (p.stats) {
        // batch-norm partials of this 256-pixel row block (conv + bias, before any activation): lanes -> waves -> LDS
        __shared__ float sred[4][16];
        const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float v = live ? acc[k] + (p.bias && kg + k < p.K ? p.bias[kg + k] : 0.f) : 0.f;
            const float s1 = wave_sum(v), s2 = wave_sum(v * v);
            if (lane == 0) {
                sred[wid][k] = s1;
                sred[wid][8 + k] = s2;
            }
        }
        __syncthreads();
        if (threadIdx.x < 16 && kg + (threadIdx.x & 7) < p.K) {
            const int k = threadIdx.x & 7, which = threadIdx.x >> 3;
            p.stats[(bx * 2 + which) * p.stats_ld + kg + k] =
                (sred[0][threadIdx.x] + sred[1][threadIdx.x]) + (sred[2][threadIdx.x] + sred[3][threadIdx.x]);
        }
    }
    if (!live) return;
    float* dst = p.y + m * p.ldy + kg;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        if (kg + 4 * h >= p.K) break;
        float4 o = make_float4(acc[4 * h], acc[4 * h + 1], acc[4 * h + 2], acc[4 * h + 3]);
        if (p.bias) {
            const float4 b = *reinterpret_cast<const float4*>(p.bias + kg + 4 * h);
            o.x += b.x; o.y += b.y; o.z += b.z; o.w += b.w;
        }
        if (p.res) {
            const float4 rr = *reinterpret_cast<const float4*>(p.res + m * p.ldres + kg + 4 * h);
            o.x += rr.x; o.y += rr.y; o.z += rr.z; o.w += rr.w;
        }
        o.x = apply_act(o.x, p.act); o.y = apply_act(o.y, p.act);
        o.z = apply_act(o.z, p.act); o.w = apply_act(o.w, p.act);
        const int left = p.K - (kg + 4 * h);          // pad columns (K % 4 != 0) are written as zeros
        if (left < 4) {
            o.w = 0.f;
            if (left < 3) o.z = 0.f;
            if (left < 2) o.y = 0.f;
        }
        *reinterpret_cast<float4*>(dst + 4 * h) = o;
    }
}

// ------------------------------------------------------------------------------------------
// host side: configuration choice and launch
// ------------------------------------------------------------------------------------------
struct TileCfg {
    int bm, bn;
};

static TileCfg pick_cfg(int M, int Ngemm) {
    if (Ngemm <= 16) return {256, 16};
    if (Ngemm <= 64) return {128, 64};
    const long tiles128 = (long)cdiv(M, 128) * cdiv(Ngemm, 128);
    if (tiles128 < 192) return {64, 64};
    return {128, 128};
}

// Tuning record (acimg_configure): plain ints, defaults compiled in, written only by acimg_configure and never by a
// launch; the launch heuristics below read it instead of the process environment.
static AcimgConfig g_cfg = {320, 768, 1, 128, 1, 0, 0, 1, 0, 1, 0, 0, 0, 1, 0};

static int pick_splits(int M, int Ngemm, TileCfg c, int kiters) {
    // measured on the generator's 12x16 layers (192-288 tiles of 64x64, 36+ K steps: tools/splitk_sweep.sh):
    // below ~1.25 workgroups per CU a K split towards ~3 per CU pays for its reduce pass
    const int cut = g_cfg.splitk_cut, target = g_cfg.splitk_target;
    const long tiles = (long)cdiv(M, c.bm) * cdiv(Ngemm, c.bn);
    if (tiles >= cut || kiters < 8) return 1;
    long s = (target + tiles - 1) / tiles;
    if (s > kiters / 4) s = kiters / 4;
    if (s > 64) s = 64;
    if (s < 1) s = 1;
    return (int)s;
}

static size_t igemm_ws_bytes(int M, int Ngemm, int kiters) {
    TileCfg c = pick_cfg(M, Ngemm);
    int s = pick_splits(M, Ngemm, c, kiters);
    if (s <= 1) return 0;
    const int ld = (Ngemm + 3) & ~3;
    const size_t rows = (size_t)s * M * ld * sizeof(float);                                   // reduce-launch layout
    const size_t tiled = (size_t)s * cdiv(M, c.bm) * cdiv(Ngemm, c.bn) * c.bm * c.bn * sizeof(float);   // hand-off layout
    return rows > tiled ? rows : tiled;
}

template <int BM, int BN, int WGM, int WGN, int NTHR>
static void launch_cfg(const IgemmParams& p, bool nt, bool cal, dim3 grid, hipStream_t st) {
    constexpr int BK = 32;
    constexpr int a_elems = BM * (BK + 4);
    constexpr int bnt = BN * (BK + 4), bnn = BK * (BN + 4);
    const int stage = a_elems + (nt ? bnt : bnn);
    const int stat_elems = WGM * 2 * BN;
    const int elems = 2 * stage > stat_elems ? 2 * stage : stat_elems;
    const size_t shm = (size_t)elems * sizeof(float);
    if (nt) {
        if (cal) hipLaunchKernelGGL((igemm_f32_kernel<BM, BN, WGM, WGN, NTHR, true, true>), grid, dim3(NTHR), shm, st, p);
        else hipLaunchKernelGGL((igemm_f32_kernel<BM, BN, WGM, WGN, NTHR, true, false>), grid, dim3(NTHR), shm, st, p);
    } else {
        if (cal) hipLaunchKernelGGL((igemm_f32_kernel<BM, BN, WGM, WGN, NTHR, false, true>), grid, dim3(NTHR), shm, st, p);
        else hipLaunchKernelGGL((igemm_f32_kernel<BM, BN, WGM, WGN, NTHR, false, false>), grid, dim3(NTHR), shm, st, p);
    }
}

// tickets: ACIMG_TICKET_WORDS zeroed ints owned by the caller (or nullptr: split-K combines through a reduce launch)
static int launch_igemm(IgemmParams p, bool nt, void* ws, size_t ws_bytes, void* tickets, hipStream_t st) {
    if (p.M <= 0 || p.Ngemm <= 0) return fail(ACIMG_EINVAL, "igemm: empty problem");
    if ((p.C & 3) || (p.lda & 3) || (p.ldb & 3))
        return fail(ACIMG_EINVAL, "igemm: C=%d lda=%d ldb=%d must be multiples of 4", p.C, p.lda, p.ldb);
    if (!aligned16(p.A) || !aligned16(p.B)) return fail(ACIMG_EINVAL, "igemm: operands must be 16-byte aligned");
    const int nseg = p.rowrun ? p.R : p.R * p.S;
    p.L = p.rowrun ? p.S * p.C : p.C;
    p.cps = cdiv(p.L, 32);
    p.kiters = nseg * p.cps;
    p.ntaps = p.R * p.S;
    const bool cal = (p.C % 32) == 0;
    // buffer descriptors: extents of A (gathered NHWC tensor) and B
    const long nimg = p.M / ((long)p.OH * p.OW);
    const long a_bytes = ((nimg * p.H * p.W - 1) * p.lda + p.C) * 4;
    const long b_bytes = nt ? (((long)(p.ntaps - 1) * p.tap_stride + (long)(p.Ngemm - 1) * p.ldb + p.C) * 4)
                            : ((long)p.R * p.S * p.C * p.ldb * 4);
    if (a_bytes >= (1L << 31) || b_bytes >= (1L << 31) || a_bytes <= 0 || b_bytes <= 0)
        return fail(ACIMG_EINVAL, "igemm: operand extent %ld / %ld bytes outside (0, 2 GiB)", a_bytes, b_bytes);
    p.a_bytes = (unsigned)a_bytes;
    p.b_bytes = (unsigned)b_bytes;
    EpiParams& e = p.e;
    e.vec = aligned16(e.Y) && (e.ldy & 3) == 0 && (!e.bias || aligned16(e.bias)) &&
            (!e.res || (aligned16(e.res) && (e.ldres & 3) == 0)) &&
            (!e.mask || (aligned16(e.mask) && (e.ldmask & 3) == 0)) && (!e.scatter || (e.Ko & 3) == 0);
    TileCfg c = pick_cfg(p.M, p.Ngemm);
    p.splits = pick_splits(p.M, p.Ngemm, c, p.kiters);
    float* stats_after = nullptr;  // split-K + BN statistics: a small pass over y afterwards
    if (e.stats && p.splits > 1) {
        // (a bias is fine: the statistics pass reads y = acc + bias, which is what the batch norm normalises)
        if (e.scatter || e.res || e.mask || e.act != ACIMG_ACT_NONE)
            return fail(ACIMG_EINVAL, "igemm: statistics with split-K need a raw (conv + bias) output");
        stats_after = e.stats;
        e.stats = nullptr;
    }
    p.slab = nullptr;
    p.slab_ld = (p.Ngemm + 3) & ~3;
    p.ts_counters = nullptr;
    dim3 grid(cdiv(p.M, c.bm), cdiv(p.Ngemm, c.bn), p.splits);
    if (p.splits > 1) {
        size_t need = (size_t)p.splits * p.M * p.slab_ld * sizeof(float);
        const size_t tiles = (size_t)grid.x * grid.y;
        if (tickets && (reinterpret_cast<uintptr_t>(tickets) & 15))
            return fail(ACIMG_EINVAL, "igemm: ticket words must be 16-byte aligned");
        const bool handoff = tickets != nullptr && tiles <= ACIMG_TICKET_WORDS && g_cfg.splitk_handoff;
        if (handoff) need = (size_t)p.splits * tiles * c.bm * c.bn * sizeof(float);
        if (ws == nullptr || ws_bytes < need)
            return fail(ACIMG_EWORKSPACE, "igemm: workspace %zu < %zu", ws_bytes, need);
        p.slab = static_cast<float*>(ws);
        if (handoff) p.ts_counters = static_cast<int*>(tickets);
    }
    if (c.bm == 128 && c.bn == 128) launch_cfg<128, 128, 2, 4, 512>(p, nt, cal, grid, st);
    else if (c.bm == 128 && c.bn == 64) launch_cfg<128, 64, 2, 2, 256>(p, nt, cal, grid, st);
    else if (c.bm == 64 && c.bn == 64) launch_cfg<64, 64, 2, 2, 256>(p, nt, cal, grid, st);
    else launch_cfg<256, 16, 4, 1, 256>(p, nt, cal, grid, st);
    int rc = check_launch("igemm");
    if (rc) return rc;
    if (p.splits > 1) {
        if (p.ts_counters == nullptr) {
            const long total = (long)p.M * p.Ngemm;
            hipLaunchKernelGGL(igemm_splitk_reduce_kernel, dim3(cdiv(total, 256)), dim3(256), 0, st,
                               p.slab, p.splits, p.M, p.Ngemm, p.slab_ld, p.e);
            rc = check_launch("igemm_splitk_reduce");
        }
        if (!rc && stats_after) {
            hipLaunchKernelGGL(partial_stats_kernel, dim3(cdiv(p.M, 256)), dim3(256), 0, st, e.Y, e.ldy, p.M,
                               e.Nstore, stats_after, e.stats_ld, 256);
            rc = check_launch("partial_stats");
        }
    }
    return rc;
}

static int pick_wgrad_splits(int M, int KK, int Ngemm, int bmo, int bn) {
    const long tiles = (long)cdiv(KK, bmo) * cdiv(Ngemm, bn);
    long s = (512 + tiles - 1) / tiles;   // ~2 workgroups per CU; every split costs a KK x N slab round trip
    const int minpix = g_cfg.wgrad_minpix;
    const long maxs = (M + minpix - 1) / minpix;  // at least `minpix` pixels per split
    long cap = 128;
    if ((long)KK * Ngemm <= 8192) {       // few-channel layers (the RGB / spectrogram U-Nets: 72 x 8 ... 288 x 32
        s = (2048 + tiles - 1) / tiles;   // weights, millions of pixels): slabs are a few KB, the pixel stream is
        cap = 2048;                       // everything -> ~8 workgroups per CU
    }
    if (s > maxs) s = maxs;
    if (s > cap) s = cap;
    if (s < 1) s = 1;
    return (int)s;
}
static void wgrad_tile(int Ngemm, int& bmo, int& bn) {
    bmo = 128;
    bn = Ngemm <= 16 ? 16 : (Ngemm <= 32 ? 32 : (Ngemm <= 64 ? 64 : 128));
    // column counts just above a multiple of 128 (the generator's 133- and 144-column layers): 64-column tiles
    // cover them with fewer padded columns (136 -> 192 instead of 256)
    if (Ngemm > 64 && cdiv(Ngemm, 64) * 64 < cdiv(Ngemm, 128) * 128) bn = 64;
}
static int wgrad_split3_bn(int Ngemm) {      // column tile of the split-MFMA weight-gradient kernel
    return Ngemm > 64 ? 128 : (Ngemm > 32 ? 64 : 32);   // (64-column tiles for 144 columns: measured equal / slower)
}
static size_t wgrad_ws_bytes(int M, int KK, int Ngemm, int ldo) {
    // one sizing query serves acimg_conv2d_wgrad and acimg_conv2d_wgrad_split3 / _bf16: the larger of their slab counts
    int bmo, bn;
    wgrad_tile(Ngemm, bmo, bn);
    int s = std::max(pick_wgrad_splits(M, KK, Ngemm, bmo, bn), pick_wgrad_splits(M, KK, Ngemm, bmo, wgrad_split3_bn(Ngemm)));
    // the halo form of the 3x3 32- / 64-channel -> 32-column layers (wgrad_halo16_kernel) leaves one slab per CU
    if (Ngemm <= 32 && KK % 9 == 0 && KK <= 9 * 64 && M >= 65536 && s < (KK <= 9 * 16 ? 512 : 256)) s = KK <= 9 * 16 ? 512 : 256;
    return s > 1 ? (size_t)(s + 1) * ((size_t)KK + 1) * ldo * sizeof(float) : 0;
}

// ------------------------------------------------------------------------------------------
// few-channel weight gradient (C <= 16 input channels, <= 32 output channels; the full-resolution layers of the
// RGB / spectrogram U-Nets: 72 x 8 ... 288 x 32 weights, millions of pixels).  As an implicit GEMM the im2col
// gather re-reads x once per tap through L2 (9x for 3x3); here a workgroup stages an 8 x 32 output-pixel tile of
// gy and the matching x tile WITH ITS HALO in LDS once and builds every tap from there:
//   dW[kk][n] += x_patch(pixel, kk) * gy[pixel][n]   on exact-f32 MFMA (16x16x4: 16 kk rows x 16 columns x 4
//   pixels), operands read from LDS per lane (ds_read_b32), the bias gradient as an all-ones row kk = KK.
// Workgroups walk tiles grid-stride and keep their sums in registers; one partial slab per workgroup, then the
// ordinary deterministic slab reduce.
// ------------------------------------------------------------------------------------------
struct WgradHaloParams {
    const float* X; int H, W, C, ldx;
    const float* G; int OH, OW, Kp, ldg;
    int R, S, stride, pad_t, pad_l;
    int XH, XW;                 // x tile extent incl. halo
    int tiles_x, tiles_y; long tiles;
    int KK;                     // R*S*C
    float* out; float* db_out; int ldo;   // slabs [gridDim.x][KK][ldo], [gridDim.x][ldo]
};
constexpr int WH_TH = 8, WH_TW = 32, WH_MAXT = 10;

// NT: 16-column tiles (1 or 2); NKT: 16-row tiles of (R*S*C + 1) the instance has accumulators for
template <int NT, int NKT>
__global__ __launch_bounds__(256, 3) void wgrad_halo_kernel(const WgradHaloParams p) {
    extern __shared__ __attribute__((aligned(16))) float wh_smem[];
    const int xsz = p.XH * p.XW * p.C;
    float* xs = wh_smem;                       // [XH][XW][C], then {1.0f, 0.0f, 0, 0}
    float* gs = wh_smem + xsz + 4;             // [8][32][Kp]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int row16 = lane & 15, quad = lane >> 4;
    // LDS offset of this lane's x element for row tile t, relative to the pixel's window origin; the bias row reads
    // the constant 1, rows past it the constant 0 (absolute addresses: their pixel offset is masked away)
    int xoff[NKT], xmask[NKT];
#pragma unroll
    for (int t = 0; t < NKT; ++t) {
        const int kk = 16 * t + row16;
        if (kk < p.KK) {
            const int tap = kk / p.C, c = kk - tap * p.C;
            const int r = tap / p.S, q = tap - r * p.S;
            xoff[t] = (r * p.XW + q) * p.C + c;
            xmask[t] = -1;
        } else {
            xoff[t] = xsz + (kk == p.KK ? 0 : 1);      // the constants 1 (bias row) and 0
            xmask[t] = 0;
        }
    }
    int bcol[NT];
    float bscale[NT];               // columns past Kp multiply a valid (finite) element by 0
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        bscale[j] = row16 + 16 * j < p.Kp ? 1.f : 0.f;
        bcol[j] = min(row16 + 16 * j, p.Kp - 1);
    }
    f32x4 acc[NKT][NT];
#pragma unroll
    for (int t = 0; t < NKT; ++t)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[t][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (tid == 0) {
        xs[xsz] = 1.f;
        xs[xsz + 1] = 0.f;
    }
    const int c4n = p.C >> 2, k4n = p.Kp >> 2;
    const int c4sh = c4n == 1 ? 0 : (c4n == 2 ? 1 : 2);
    // Software pipeline: the global loads of tile i+1 are issued (into registers) before tile i is multiplied, so
    // every workgroup keeps a tile's worth of HBM requests in flight all the time.
    // Staging is division-free and branch-free (stride 1, R, S <= 3: at most 10 x 34 x pixels): wave w takes x rows
    // w, w+4, w+8, its lanes the row's float4s lane, lane+64, lane+128; every load goes to a clamped (valid) address
    // and is zeroed by select, so all 9 + 4 loads of a thread are in flight together.
    float4 xv[3][3], gv[2][2];
    const int rowlen = p.XW << c4sh;
    auto issue = [&](long tile) {
        const int tx = (int)(tile % p.tiles_x);
        const long t2 = tile / p.tiles_x;
        const int ty = (int)(t2 % p.tiles_y);
        const long img = t2 / p.tiles_y;
        const int oh0 = ty * WH_TH, ow0 = tx * WH_TW;
        const int ih0 = oh0 * p.stride - p.pad_t, iw0 = ow0 * p.stride - p.pad_l;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const int iy = wid + 4 * a;
            const int ih = ih0 + iy;
            const int ihc = min(max(ih, 0), p.H - 1);
            const float* grow = p.X + ((img * p.H + ihc) * p.W) * p.ldx;
#pragma unroll
            for (int bq = 0; bq < 3; ++bq) {
                const int e = lane + 64 * bq;
                const int ix = e >> c4sh, c4 = e & (c4n - 1);
                const int iw = iw0 + ix;
                const bool ok = iy < p.XH && e < rowlen && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
                const float4 v = *reinterpret_cast<const float4*>(grow + (long)min(max(iw, 0), p.W - 1) * p.ldx + 4 * c4);
                xv[a][bq] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int py = 2 * wid + half, px = lane & 31;
            const int oh = oh0 + py, ow = ow0 + px;
            const bool ok = oh < p.OH && ow < p.OW;
            const float* gsrc = p.G + ((img * p.OH + min(oh, p.OH - 1)) * p.OW + min(ow, p.OW - 1)) * p.ldg;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int k4 = (lane >> 5) + 2 * j;
                const float4 v = *reinterpret_cast<const float4*>(gsrc + 4 * min(k4, k4n - 1));
                gv[half][j] = ok && k4 < k4n ? v : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const int iy = wid + 4 * a;
#pragma unroll
            for (int bq = 0; bq < 3; ++bq) {
                const int e = lane + 64 * bq;
                if (iy < p.XH && e < rowlen)
                    *reinterpret_cast<float4*>(xs + iy * p.XW * p.C + (e >> c4sh) * p.C + 4 * (e & (c4n - 1))) = xv[a][bq];
            }
        }
#pragma unroll
        for (int half = 0; half < 2; ++half)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int k4 = (lane >> 5) + 2 * j;
                if (k4 < k4n)
                    *reinterpret_cast<float4*>(gs + ((2 * wid + half) * WH_TW + (lane & 31)) * p.Kp + 4 * k4) = gv[half][j];
            }
    };
    issue(blockIdx.x);
    for (long tile = blockIdx.x; tile < p.tiles; tile += gridDim.x) {
        __syncthreads();                                   // the previous tile has been consumed
        commit();
        __syncthreads();
        if (tile + gridDim.x < p.tiles) issue(tile + gridDim.x);
        // this wave: tile rows 2*wid, 2*wid+1 = 64 pixels, 4 at a time (quad = which of the 4); all addresses
        // advance incrementally (4 pixels per step, one row jump half way)
        {
            // register double buffer: the LDS reads of step g+1 are issued before the MFMAs of step g (all NKT row
            // tiles unconditionally - rows past KK + 1 read the constant 0 - so the loop has no branch)
            int xb = ((2 * wid) * p.XW + quad) * p.C;
            const float* gp = gs + ((2 * wid) * WH_TW + quad) * p.Kp;
            float a[2][NKT], b[2][NT];
#pragma unroll
            for (int t = 0; t < NKT; ++t) a[0][t] = xs[xoff[t] + (xb & xmask[t])];
#pragma unroll
            for (int j = 0; j < NT; ++j) b[0][j] = gp[bcol[j]] * bscale[j];
#pragma unroll 1
            for (int g2 = 0; g2 < 8; ++g2) {            // two steps per trip: buffers 0 -> 1 -> 0
#pragma unroll
                for (int cur = 0; cur < 2; ++cur) {
                    const int nxt = cur ^ 1;
                    const int gq = 2 * g2 + cur;
                    if (gq < 15) {
                        gp += 4 * p.Kp;
                        xb += 4 * p.C + (gq == 7 ? (p.XW - WH_TW) * p.C : 0);
#pragma unroll
                        for (int t = 0; t < NKT; ++t) a[nxt][t] = xs[xoff[t] + (xb & xmask[t])];
#pragma unroll
                        for (int j = 0; j < NT; ++j) b[nxt][j] = gp[bcol[j]] * bscale[j];
                    }
#pragma unroll
                    for (int t = 0; t < NKT; ++t)
#pragma unroll
                        for (int j = 0; j < NT; ++j)
                            acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[cur][t], b[cur][j], acc[t][j], 0, 0, 0);
                }
            }
        }
    }
    // the 4 waves add their sums in wave order (deterministic) in LDS: red[kk][NT*16]
    __syncthreads();
    float* red = wh_smem;
    const int rw = NT * 16;
    for (int w = 0; w < 4; ++w) {
        if (wid == w) {
#pragma unroll
            for (int t = 0; t < NKT; ++t) {
#pragma unroll
                for (int j = 0; j < NT; ++j)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float* dst = red + (16 * t + 4 * quad + i) * rw + 16 * j + row16;
                        *dst = (w == 0 ? 0.f : *dst) + acc[t][j][i];
                    }
            }
        }
        __syncthreads();
    }
    float* out = p.out + (long)blockIdx.x * p.KK * p.ldo;
    for (int e = tid; e < (p.KK + 1) * p.Kp; e += 256) {
        const int kk = e / p.Kp, n = e - kk * p.Kp;
        const float v = red[kk * rw + n];
        if (kk < p.KK) out[(long)kk * p.ldo + n] = v;
        else if (p.db_out) p.db_out[(long)blockIdx.x * p.ldo + n] = v;
    }
}

static size_t wgrad_halo_lds(int R, int S, int C, int Kp, int stride) {
    const int XH = (WH_TH - 1) * stride + R, XW = (WH_TW - 1) * stride + S;
    const size_t stage = ((size_t)XH * XW * C + 4 + (size_t)WH_TH * WH_TW * Kp) * sizeof(float);
    const size_t red = (size_t)WH_MAXT * 16 * (Kp > 16 ? 32 : 16) * sizeof(float);   // any instance's [16*NKT][16*NT]
    return stage > red ? stage : red;
}
static bool wgrad_halo_ok(const WgradParams& p) {
    return (p.C == 4 || p.C == 8 || p.C == 16) && p.Nld <= 16 && p.Ngemm == p.Nld && p.stride == 1 && p.R <= 3 && p.S <= 3 && (long)p.M >= 65536 && (p.ldx & 3) == 0 && (p.ldg & 3) == 0 &&
           (p.KK + 1 + 15) / 16 <= WH_MAXT && wgrad_halo_lds(p.R, p.S, p.C, p.Nld, p.stride) <= 65536 &&
           g_cfg.wgrad_halo;
}

// db (optional): fused bias gradient, db[n] = sum_m G[m][n] for n < Ngemm
// ------------------------------------------------------------------------------------------
// HALO form of the bf16 / bf16x3 weight gradient of the 3x3 / stride-1 layers with 32 or 64 input channels and 32 output
// channels (round 4; the 112x149 and 56x74 stages of the RGB / spectrogram U-Nets, models/unet_architecture.py:161-166,
// configs[1]).  As an implicit GEMM (wgrad_split3_kernel) these layers re-gather x once per tap through L2 - nine times the
// tensor for 2 x 576 x 32 MACs per pixel: 112x149 64->32 took 225 us against 41 us of HBM time for x and gy.  Here a
// workgroup stages a TH x 32 output-pixel tile of gy and the x tile WITH ITS HALO once, as bf16 (hi [, lo]) planes in LDS
// ([pixel][channel], 16-byte chunks swizzled by the pixel's COLUMN so that the transposing fragment reads - 4 pixels x 16
// channels per 16-lane group - are conflict free at every tap shift, and row offsets stay compile-time immediates), and
// forms all nine taps from there: K = the tile's pixels, one 32-pixel row per step, no barrier inside a tile.  gy^T sits in
// the A slot (a lane's 4 accumulators are 4 consecutive output channels of one dW row: 16-byte slab stores); the bias
// gradient rides as an MFMA against a constant ones fragment.  One workgroup per CU with the NEXT tile's global loads held in
// registers while the current one is multiplied (2 waves per SIMD: the 256-register budget pays for that); workgroups walk
// tiles grid-stride and keep their sums in registers: one partial slab each, then the deterministic slab reduce.
// Waves: C = 64: 4 channel tiles x 2 column tiles; C = 32: 2 x 2 x the tile's even / odd rows (summed through LDS at the end).
// ------------------------------------------------------------------------------------------
// the producer's batch norm + ReLU on a staged item (after ALL of a tile's loads were issued: applied inside the load loop it
// made every load wait for the one before it - 224x298 8->8 weight gradient 68 -> 132 us)
__device__ __forceinline__ float4 affine_relu4(float4 v, const float4 sc, const float4 sh, const bool relu) {
    v.x = fmaf(v.x, sc.x, sh.x); v.y = fmaf(v.y, sc.y, sh.y); v.z = fmaf(v.z, sc.z, sh.z); v.w = fmaf(v.w, sc.w, sh.w);
    if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    return v;
}

struct WgradHalo16Params {
    const float* X; int H, W, ldx;
    const float* G; int ldg;
    int tiles_x, tiles_y; long tiles;
    float* out; float* db_out; int ldo;      // slabs [gridDim.x][9 creal][ldo], [gridDim.x][ldo]
    int creal, nreal;                        // channels really there (multiples of 4; the rest of the C x 32 tile is zeros)
    // the producer's deferred batch norm on load: x' = relu(x * a_scale[c] + a_shift[c]) for pixels INSIDE the image (the
    // conv's zero padding applies after the affine); null = x as stored
    const float* a_scale; const float* a_shift; int a_relu;
};

// C: channel width of the x image (64, 32, or 16 for the few-channel layers: fewer real channels are zero padded);
// NNT: 16-column tiles of gy (2, or 1 for <= 16 output channels: the waves that would multiply padding take tile rows instead)
template <int C, int TERMS, int NNT = 2>
__global__ __launch_bounds__(512, C == 16 ? 4 : 1) void wgrad_halo16_kernel(const WgradHalo16Params p) {
    constexpr int TH = (TERMS == 1 || C == 16) ? 8 : 4, TW = 32, XH = TH + 2, XWV = TW + 2, XW = 36;
    constexpr int PITCH = C * 2;                      // bytes per pixel and plane
    constexpr int XPL = XH * XW * PITCH;              // one x plane
    constexpr int GPL = TH * TW * 64;                 // one gy plane (32 columns of bf16)
    constexpr int NCT = C / 16;                       // 16-channel tiles
    constexpr int NJ = 8 / (NCT * NNT);               // row groups: waves with the same (channel tile, column tile)
    constexpr int NXL = (XH * XWV * (C / 4) + 511) / 512, NGL = TH * TW * 8 / 512;     // float4 loads per thread and tile
    static_assert((C == 64 || C == 32 || C == 16) && TH % NJ == 0 && TH * TW * 8 % 512 == 0, "shape");
    extern __shared__ __attribute__((aligned(16))) float wh16_smem[];
    char* const lds = reinterpret_cast<char*>(wh16_smem);
    char* const gl = lds + (TERMS == 3 ? 2 : 1) * XPL;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int li = lane & 15, g = lane >> 4, q = li >> 2, pp = li & 3;
    const int ct = wid % NCT, nt = (wid / NCT) % NNT, jh = wid / (NCT * NNT);
    // (C = 16: 32-byte pixels, no room to swizzle: pixels 8 apart share banks, a 2-way conflict the HBM-bound kernel absorbs)
    auto swx = [](int col) {
        return C == 64 ? 2 * (((col >> 1) & 1) | (((col >> 3) & 1) << 1)) : (C == 32 ? 2 * ((col >> 3) & 1) : 0);
    };
    auto swg = [](int col) { return 2 * ((col >> 3) & 1); };

    // fragment read addresses: lane (q, pp) of group g supplies pixel column s + 8 g + 4 h + q, channels 4 pp .. + 3 of its
    // 16-channel tile; the tile row is a compile-time distance
    int xb[3][2], gb[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int col = 8 * g + 4 * h + q;
        gb[h] = (jh * TW + col) * 64 + (((2 * nt + (pp >> 1)) ^ swg(col)) << 4) + 8 * (pp & 1);
#pragma unroll
        for (int s_ = 0; s_ < 3; ++s_) {
            const int cx = col + s_;
            xb[s_][h] = (jh * XW + cx) * PITCH + (((2 * ct + (pp >> 1)) ^ swx(cx)) << 4) + 8 * (pp & 1);
        }
    }

    // this thread's items of a tile: x float4 (pixel of the XH x 34 window, 4 channels), gy float4 (pixel, 4 columns)
    float4 rx[NXL], rg[NGL];
    unsigned okm = 0;                          // which of rx[] came from inside the image (the affine applies to those only)
    // (a thread's x items are always the same four channels: 512 is a multiple of C / 4)
    const bool aff_ch = p.a_scale != nullptr && (tid % (C / 4)) * 4 < p.creal;
    const float4 asc = aff_ch ? *reinterpret_cast<const float4*>(p.a_scale + (tid % (C / 4)) * 4) : make_float4(1.f, 1.f, 1.f, 1.f);
    const float4 ash = aff_ch ? *reinterpret_cast<const float4*>(p.a_shift + (tid % (C / 4)) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    auto load_tile = [&](long tile) {
        okm = 0;
        const int tx = (int)(tile % p.tiles_x);
        const long t2 = tile / p.tiles_x;
        const int ty = (int)(t2 % p.tiles_y);
        const long img = t2 / p.tiles_y;
        const float* xi = p.X + img * p.H * p.W * p.ldx;
        const float* gi = p.G + img * p.H * p.W * p.ldg;
#pragma unroll
        for (int k = 0; k < NXL; ++k) {
            const int i = tid + 512 * k;
            const int c4 = i % (C / 4), pix = i / (C / 4);
            const int row = pix / XWV, col = pix - row * XWV;
            const int iy = ty * TH + row - 1, ix = tx * TW + col - 1;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < XH && c4 * 4 < p.creal && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) {
                v = *reinterpret_cast<const float4*>(xi + ((long)iy * p.W + ix) * p.ldx + c4 * 4);
                okm |= 1u << k;
            }
            rx[k] = v;
        }
#pragma unroll
        for (int k = 0; k < NGL; ++k) {
            const int i = tid + 512 * k;
            const int c4 = i & 7, pix = i >> 3;
            const int row = pix / TW, col = pix - row * TW;
            const int oy = ty * TH + row, ox = tx * TW + col;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (oy < p.H && ox < p.W && c4 * 4 < p.nreal) v = *reinterpret_cast<const float4*>(gi + ((long)oy * p.W + ox) * p.ldg + c4 * 4);
            rg[k] = v;
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int k = 0; k < NXL; ++k) {
            const int i = tid + 512 * k;
            const int c4 = i % (C / 4), pix = i / (C / 4);
            const int row = pix / XWV, col = pix - row * XWV;
            if (row < XH) {
                const int off = (row * XW + col) * PITCH + (((c4 >> 1) ^ swx(col)) << 4) + 8 * (c4 & 1);
                uint2 hi, lo;
                split4<SplitBF16>(p.a_scale && ((okm >> k) & 1u) ? affine_relu4(rx[k], asc, ash, p.a_relu != 0) : rx[k], hi, lo);
                *reinterpret_cast<uint2*>(lds + off) = hi;
                if (TERMS == 3) *reinterpret_cast<uint2*>(lds + XPL + off) = lo;
            }
        }
#pragma unroll
        for (int k = 0; k < NGL; ++k) {
            const int i = tid + 512 * k;
            const int c4 = i & 7, pix = i >> 3;
            const int col = pix & (TW - 1);
            const int off = pix * 64 + (((c4 >> 1) ^ swg(col)) << 4) + 8 * (c4 & 1);
            uint2 hi, lo;
            split4<SplitBF16>(rg[k], hi, lo);
            *reinterpret_cast<uint2*>(gl + off) = hi;
            if (TERMS == 3) *reinterpret_cast<uint2*>(gl + GPL + off) = lo;
        }
    };
    typedef short s16x4_ __attribute__((ext_vector_type(4)));
    typedef short s16x8_ __attribute__((ext_vector_type(8)));
    auto frag = [&](const char* base, int a0, int a1) -> b16x8 {
        const s16x4_ v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_*)(base + a0));
        const s16x4_ v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_*)(base + a1));
        const s16x8_ v = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        return __builtin_bit_cast(b16x8, v);
    };

    f32x4 acc[9], accb = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const __bf16 one = (__bf16)1.f;
    const b16x8 ones = {one, one, one, one, one, one, one, one};

    long tile = blockIdx.x;
    if (tile < p.tiles) load_tile(tile);
    for (; tile < p.tiles; tile += gridDim.x) {
        __syncthreads();                               // everyone has finished reading the previous tile
        store_tile();
        __syncthreads();
        if (tile + gridDim.x < p.tiles) load_tile(tile + gridDim.x);     // in flight while this tile is multiplied
#pragma unroll
        for (int jj = 0; jj < TH / NJ; ++jj) {
            const int j = jj * NJ;                     // (+ jh: in the lane bases)
            const b16x8 gh = frag(gl + j * TW * 64, gb[0], gb[1]);
            b16x8 glo;
            if (TERMS == 3) glo = frag(gl + GPL + j * TW * 64, gb[0], gb[1]);
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int s_ = 0; s_ < 3; ++s_) {
                    const char* xrow = lds + (j + r) * XW * PITCH;
                    const b16x8 xh = frag(xrow, xb[s_][0], xb[s_][1]);
                    if (TERMS == 3) {
                        const b16x8 xl = frag(xrow + XPL, xb[s_][0], xb[s_][1]);
                        acc[r * 3 + s_] = SplitBF16::mfma(glo, xh, acc[r * 3 + s_]);
                        acc[r * 3 + s_] = SplitBF16::mfma(gh, xl, acc[r * 3 + s_]);
                    }
                    acc[r * 3 + s_] = SplitBF16::mfma(gh, xh, acc[r * 3 + s_]);
                }
            if (ct == 0) {
                if (TERMS == 3) accb = SplitBF16::mfma(glo, ones, accb);
                accb = SplitBF16::mfma(gh, ones, accb);
            }
        }
    }
    // the row groups of one (channel tile, column tile) meet through LDS (C = 32), then lane (li, g) of acc[tap] holds
    // dW[tap * C + 16 ct + li][16 nt + 4 g .. + 3]
    if (NJ > 1) {
        // one round per row group (in group order: deterministic): its waves park their sums, group 0 adds them
        f32x4* red = reinterpret_cast<f32x4*>(lds);
        const int w0 = wid % (NCT * NNT);            // this wave's (channel tile, column tile) slot
        for (int r = 1; r < NJ; ++r) {
            __syncthreads();
            if (jh == r) {
#pragma unroll
                for (int t = 0; t < 9; ++t) red[(w0 * 10 + t) * 64 + lane] = acc[t];
                red[(w0 * 10 + 9) * 64 + lane] = accb;
            }
            __syncthreads();
            if (jh == 0) {
#pragma unroll
                for (int t = 0; t < 9; ++t) acc[t] += red[(w0 * 10 + t) * 64 + lane];
                accb += red[(w0 * 10 + 9) * 64 + lane];
            }
        }
        if (jh != 0) return;
    }
    float* out = p.out + (long)blockIdx.x * (9 * p.creal) * p.ldo;
    const bool nok = nt * 16 + 4 * g < p.nreal;
    if (nok && ct * 16 + li < p.creal) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
            *reinterpret_cast<f32x4*>(out + (long)(t * p.creal + ct * 16 + li) * p.ldo + nt * 16 + 4 * g) = acc[t];
    }
    if (p.db_out && ct == 0 && li == 0 && nok)
        *reinterpret_cast<f32x4*>(p.db_out + (long)blockIdx.x * p.ldo + nt * 16 + 4 * g) = accb;
}

// split3: the caller asked for 16-bit matrix-core arithmetic (acimg_conv2d_wgrad_split3 / _bf16).  The FEW-CHANNEL layers
// (fewer than 32 input channels: the full-resolution layers of the RGB / spectrogram U-Nets, which arrive through the fp32
// entry acimg_conv2d_wgrad) take the same kernel in its three-term form - fp32-class results (5e-6) - on a zero-padded
// 32 x 32 channel tile: 224x298 16->8 189 -> ~60 us against the exact-f32 halo kernel (round 4)
static bool wgrad_halo16_ok(const WgradParams& p, bool split3) {
    const bool few = p.C < 32;
    return (split3 || few) && p.R == 3 && p.S == 3 && p.stride == 1 && p.pad_t == 1 && p.pad_l == 1 &&
           (p.C == 64 || (p.C <= 32 && p.C % 4 == 0)) && p.Ngemm <= 32 && p.Ngemm % 4 == 0 && p.Nld == p.Ngemm && p.OH == p.H &&
           p.OW == p.W && p.ldo >= p.Ngemm && (long)p.M >= 65536 && g_cfg.wgrad_halo;
}

static int launch_wgrad(WgradParams p, float* dw, float* db, void* ws, size_t ws_bytes, hipStream_t st,
                        bool split3 = false, int terms = 3) {
    if ((p.C & 3) || (p.ldx & 3) || (p.ldg & 3) || (p.ldo & 3))
        return fail(ACIMG_EINVAL, "wgrad: C=%d ldx=%d ldg=%d ldo=%d must be multiples of 4", p.C, p.ldx, p.ldg, p.ldo);
    if (!aligned16(p.X) || !aligned16(p.G) || !aligned16(dw))
        return fail(ACIMG_EINVAL, "wgrad: operands must be 16-byte aligned");
    if (p.a_scale && !wgrad_halo16_ok(p, split3))
        return fail(ACIMG_EINVAL, "wgrad: an input affine is only taken by the halo form (3x3 / stride 1 / SAME, <= 32 columns, >= 65536 pixels)");
    if (p.a_scale && (!p.a_shift || !aligned16(p.a_scale) || !aligned16(p.a_shift)))
        return fail(ACIMG_EINVAL, "wgrad: the input affine needs scale and shift, 16-byte aligned");
    int bmo, bn;
    wgrad_tile(p.Ngemm, bmo, bn);
    if (wgrad_halo16_ok(p, split3)) {
        WgradHalo16Params q{};
        q.X = p.X; q.H = p.H; q.W = p.W; q.ldx = p.ldx; q.G = p.G; q.ldg = p.ldg; q.ldo = p.ldo;
        q.creal = p.C; q.nreal = p.Ngemm;
        q.a_scale = p.a_scale; q.a_shift = p.a_shift; q.a_relu = p.a_relu;
        if (p.C < 32) terms = 3;                                       // few-channel layers: always the fp32-class form
        const int cpad = p.C == 64 ? 64 : (p.C > 16 ? 32 : 16);
        const int nnt = (cpad == 16 && p.Ngemm <= 16) ? 1 : 2;
        const int th = (terms == 1 || cpad == 16) ? 8 : 4;
        q.tiles_x = cdiv(p.W, 32); q.tiles_y = cdiv(p.H, th);
        q.tiles = (long)(p.M / (p.H * p.W)) * q.tiles_x * q.tiles_y;
        int nb = p.C <= 16 ? 512 : 256;     // one workgroup per CU; two for the few-channel instances (56 KiB of LDS, tiny slabs:
        if (nb > q.tiles) nb = (int)q.tiles;       // their load / store / multiply phases overlap across workgroups)
        const size_t need = (size_t)nb * ((size_t)p.KK + 1) * p.ldo * sizeof(float);
        if (ws != nullptr && ws_bytes >= need) {      // (the sizing query covers it: pick_wgrad_splits gives these shapes >= 256 slabs)
            q.out = static_cast<float*>(ws);
            float* db_slab = q.out + (size_t)nb * p.KK * p.ldo;
            q.db_out = db ? db_slab : nullptr;
            const int xh = th + 2, planes = terms == 3 ? 2 : 1;
            int lds = planes * (xh * 36 * cpad * 2 + th * 32 * 64);
            if (lds < 4 * 10 * 64 * 16) lds = 4 * 10 * 64 * 16;                   // the row groups' final sums through LDS
#define ACIMG_WH16(Cv, Tv)                                                                                              \
    do {                                                                                                                \
        static bool attr_set = false;                                                                                   \
        if (!attr_set) {                                                                                                \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_halo16_kernel<Cv, Tv>),                       \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (10 * 36 * 64 * 2 + 8 * 32 * 64)); \
            attr_set = true;                                                                                            \
        }                                                                                                               \
        hipLaunchKernelGGL((wgrad_halo16_kernel<Cv, Tv>), dim3(nb), dim3(512), lds, st, q);                             \
    } while (0)
            if (cpad == 64 && terms == 1) ACIMG_WH16(64, 1);
            else if (cpad == 64) ACIMG_WH16(64, 3);
            else if (cpad == 32 && terms == 1) ACIMG_WH16(32, 1);
            else if (cpad == 32) ACIMG_WH16(32, 3);
            else if (nnt == 2) ACIMG_WH16(16, 3);
            else {
                static bool attr16 = false;
                if (!attr16) {
                    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_halo16_kernel<16, 3, 1>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
                    attr16 = true;
                }
                hipLaunchKernelGGL((wgrad_halo16_kernel<16, 3, 1>), dim3(nb), dim3(512), lds, st, q);
            }
#undef ACIMG_WH16
            int rc = check_launch("wgrad_halo16");
            if (rc) return rc;
            launch_slab_reduce_wide(q.out, nb, (long)p.KK, p.Ngemm, p.ldo, dw, db ? db_slab : nullptr, db, st);
            return check_launch("wgrad_reduce");
        }
    }
    if (p.a_scale) return fail(ACIMG_EWORKSPACE, "wgrad: workspace too small for the halo form, which the input affine needs");
    if (wgrad_halo_ok(p)) {
        WgradHaloParams q{};
        q.X = p.X; q.H = p.H; q.W = p.W; q.C = p.C; q.ldx = p.ldx;
        q.G = p.G; q.OH = p.OH; q.OW = p.OW; q.Kp = p.Nld; q.ldg = p.ldg;
        q.R = p.R; q.S = p.S; q.stride = p.stride; q.pad_t = p.pad_t; q.pad_l = p.pad_l;
        q.XH = (WH_TH - 1) * p.stride + p.R; q.XW = (WH_TW - 1) * p.stride + p.S;
        q.tiles_x = cdiv(p.OW, WH_TW); q.tiles_y = cdiv(p.OH, WH_TH);
        q.tiles = (long)(p.M / (p.OH * p.OW)) * q.tiles_x * q.tiles_y;
        q.KK = p.KK; q.ldo = p.ldo;
        int nb = pick_wgrad_splits(p.M, p.KK, p.Ngemm, bmo, bn);     // = what the workspace was sized for
        if (nb > 768) nb = 768;                                      // 3 resident workgroups per CU, each pipelined
        if (nb > q.tiles) nb = (int)q.tiles;
        if (nb < 2) nb = 2;
        const size_t need = (size_t)nb * ((size_t)p.KK + 1) * p.ldo * sizeof(float);
        if (ws == nullptr || ws_bytes < need) return fail(ACIMG_EWORKSPACE, "wgrad: workspace %zu < %zu", ws_bytes, need);
        q.out = static_cast<float*>(ws);
        float* db_slab = q.out + (size_t)nb * p.KK * p.ldo;
        q.db_out = db ? db_slab : nullptr;
        const size_t lds = wgrad_halo_lds(p.R, p.S, p.C, p.Nld, p.stride);
        const int nkt = (p.KK + 1 + 15) / 16;
#define ACIMG_WH(NTv, NKTv) hipLaunchKernelGGL((wgrad_halo_kernel<NTv, NKTv>), dim3(nb), dim3(256), lds, st, q)
        if (nkt <= 1) ACIMG_WH(1, 1);
        else if (nkt <= 2) ACIMG_WH(1, 2);
        else if (nkt <= 3) ACIMG_WH(1, 3);
        else if (nkt <= 5) ACIMG_WH(1, 5);
        else ACIMG_WH(1, 10);
#undef ACIMG_WH
        int rc = check_launch("wgrad_halo");
        if (rc) return rc;
        launch_slab_reduce_wide(q.out, nb, (long)p.KK, p.Ngemm, p.ldo, dw, db ? db_slab : nullptr, db, st);
        return check_launch("wgrad_reduce");
    }
    if (split3) bn = wgrad_split3_bn(p.Ngemm);
    p.splits = pick_wgrad_splits(p.M, p.KK, p.Ngemm, bmo, bn);
    int rps = cdiv(p.M, p.splits);
    rps = ((rps + 31) / 32) * 32;
    p.rows_per_split = rps;
    p.splits = cdiv(p.M, rps);
    float* db_slab = nullptr;
    if (p.splits > 1) {
        const size_t need = (size_t)p.splits * ((size_t)p.KK + 1) * p.ldo * sizeof(float);
        if (ws == nullptr || ws_bytes < need)
            return fail(ACIMG_EWORKSPACE, "wgrad: workspace %zu < %zu", ws_bytes, need);
        p.out = static_cast<float*>(ws);
        db_slab = p.out + (size_t)p.splits * p.KK * p.ldo;
        p.db_out = db ? db_slab : nullptr;
    } else {
        p.out = dw;
        p.db_out = db;
    }
    const int rows = p.KK + (db ? 1 : 0);
    dim3 grid(cdiv(rows, bmo), cdiv(p.Ngemm, bn), p.splits);
    if (split3 && terms == 1 && bn == 128) hipLaunchKernelGGL((wgrad_split3_kernel<128, 1, 512>), grid, dim3(512), 65536, st, p);
    else if (split3 && terms == 1 && bn == 64) hipLaunchKernelGGL((wgrad_split3_kernel<64, 1, 512>), grid, dim3(512), 65536, st, p);
    else if (split3 && terms == 1) hipLaunchKernelGGL((wgrad_split3_kernel<32, 1>), grid, dim3(256), 65536, st, p);
    else if (split3 && bn == 128) hipLaunchKernelGGL((wgrad_split3_kernel<128, 3, 512>), grid, dim3(512), 65536, st, p);
    else if (split3 && bn == 64) hipLaunchKernelGGL((wgrad_split3_kernel<64, 3, 512>), grid, dim3(512), 65536, st, p);
    else if (split3) hipLaunchKernelGGL((wgrad_split3_kernel<32>), grid, dim3(256), 65536, st, p);
    else if (bn == 128) hipLaunchKernelGGL((wgrad_f32_kernel<128, 128>), grid, dim3(256), 0, st, p);
    else if (bn == 64) hipLaunchKernelGGL((wgrad_f32_kernel<128, 64>), grid, dim3(256), 0, st, p);
    else if (bn == 32) hipLaunchKernelGGL((wgrad_f32_kernel<128, 32>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((wgrad_f32_kernel<128, 16, 4>), grid, dim3(256), 0, st, p);
    int rc = check_launch("wgrad");
    if (rc) return rc;
    if (p.splits > 1) {
        const long total = (long)p.KK * p.Ngemm;
        if (p.splits > 32) {
            launch_slab_reduce_wide(p.out, p.splits, (long)p.KK, p.Ngemm, p.ldo, dw, db ? db_slab : nullptr, db, st);
        } else {
            const int nb1 = (int)cdiv(total, 256), nb2 = db ? cdiv(p.Ngemm, 256) : 0;
            hipLaunchKernelGGL(slab_reduce_kernel, dim3(nb1 + nb2), dim3(256), 0, st, p.out, p.splits,
                               (long)p.KK, p.Ngemm, p.ldo, dw, nb1, db_slab, db);
        }
        rc = check_launch("wgrad_reduce");
    }
    return rc;
}

// column sums; workspace: parts*ncols floats
static int launch_colsum(const float* G, long rows, int ncols, int ld, float* out, void* ws,
                         size_t ws_bytes, hipStream_t st) {
    int parts = (int)((rows + 255) / 256);
    if (parts > 256) parts = 256;
    if (parts < 1) parts = 1;
    const long rpb = (rows + parts - 1) / parts;
    const size_t need = (size_t)parts * ncols * sizeof(float);
    if (ws == nullptr || ws_bytes < need) return fail(ACIMG_EWORKSPACE, "colsum: workspace %zu < %zu", ws_bytes, need);
    float* partial = static_cast<float*>(ws);
    if (ncols <= 64 && (ncols & 3) == 0 && (ld & 3) == 0 && aligned16(G))
        hipLaunchKernelGGL(colsum_narrow_kernel, dim3(parts), dim3(256), 0, st, G, rows, ncols, ld, rpb, partial);
    else
        hipLaunchKernelGGL(colsum_partial_kernel, dim3(cdiv(ncols, 64), parts), dim3(256), 0, st, G, rows,
                           ncols, ld, rpb, partial);
    hipLaunchKernelGGL(colsum_final_kernel, dim3(cdiv(ncols, 64)), dim3(256), 0, st, partial, parts, ncols, out);
    return check_launch("colsum");
}
static size_t colsum_ws_bytes(int ncols) { return (size_t)256 * ncols * sizeof(float); }

// row_run: the caller's kernel accepts ldx < C with S == 1: the C "channels" of a tap are then a run of C / ldx
// consecutive pixels of one input row (windows of neighbouring outputs overlap), see acimg_conv2d_fwd_split3
static int check_desc(const AcimgConvDesc* d, const char* who, bool row_run = false) {
    if (!d) return fail(ACIMG_EINVAL, "%s: null descriptor", who);
    if (d->N <= 0 || d->H <= 0 || d->W <= 0 || d->C <= 0 || d->K <= 0 || d->OH <= 0 || d->OW <= 0 ||
        d->R <= 0 || d->S <= 0 || d->stride <= 0)
        return fail(ACIMG_EINVAL, "%s: non-positive dimension", who);
    const bool run_view = row_run && d->S == 1 && d->pad_l == 0 && d->ldx > 0 && d->C % d->ldx == 0 &&
                          (d->OW - 1) * d->stride + d->C / d->ldx <= d->W;
    if ((d->C & 3) || (d->ldx & 3) || (d->ldw & 3) || (d->ldx < d->C && !run_view))
        return fail(ACIMG_EINVAL, "%s: C=%d ldx=%d ldw=%d must be multiples of 4 (ldx>=C)", who, d->C, d->ldx, d->ldw);
    if ((long)d->N * d->H * d->W * d->ldx >= (1L << 31) || (long)d->N * d->OH * d->OW * (long)d->ldy >= (1L << 31))
        return fail(ACIMG_EINVAL, "%s: tensor exceeds 2^31 elements", who);
    return ACIMG_OK;
}
static inline int up4(int v) { return (v + 3) & ~3; }

}  // namespace acimg

using namespace acimg;

// ==========================================================================================
// C ABI
// ==========================================================================================
extern "C" {

// few-channel direct path: C <= 16 (wider pixels stop coalescing across lanes), K <= 32 and a multiple of 8
static bool direct_ok(int C, int K, int ldy, int ldres, const float* y, const float* bias, const float* res,
                      bool affine, const float* mask) {
    return !affine && !mask && C <= 16 && K <= 32 && ldy >= ((K + 3) & ~3) && (C & 3) == 0 && (ldy & 3) == 0 &&
           (!res || ((ldres & 3) == 0 && aligned16(res))) && aligned16(y) && (!bias || aligned16(bias));
}
static size_t direct_ws_bytes(int R, int S, int C, int K) { return ((size_t)R * S * C * K * sizeof(float) + 255) & ~(size_t)255; }
static int launch_direct(const DirectParams& q, void* ws, size_t ws_bytes, hipStream_t st) {
    if (!aligned16(q.x) || (q.ldx & 3)) return fail(ACIMG_EINVAL, "direct conv: input must be 16-byte aligned");
    const int total = q.R * q.S * q.C * ((q.K + 7) & ~7);
    if (!ws || ws_bytes < (size_t)total * sizeof(float)) return fail(ACIMG_EWORKSPACE, "direct conv: workspace too small");
    float* wprep = static_cast<float*>(ws);
    hipLaunchKernelGGL(direct_prepare_kernel, dim3(cdiv(total, 256)), dim3(256), 0, st, q, wprep, total);
    const long units = (long)cdiv(q.M, 256) * cdiv(q.K, 8);
    const dim3 grid((unsigned)(8 * ((units + 7) / 8)));          // 8 equal ranges, one per XCD (see the kernel)
#define ACIMG_DIRECT(RR, SS, CC) \
    hipLaunchKernelGGL((direct_conv_kernel<RR, SS, CC>), grid, dim3(256), 0, st, q, wprep)
    if (q.R == 3 && q.S == 3 && q.C == 4) ACIMG_DIRECT(3, 3, 4);
    else if (q.R == 3 && q.S == 3 && q.C == 8) ACIMG_DIRECT(3, 3, 8);
    else if (q.R == 3 && q.S == 3 && q.C == 16) ACIMG_DIRECT(3, 3, 16);
    else if (q.R == 1 && q.S == 1 && q.C == 8) ACIMG_DIRECT(1, 1, 8);
    else if (q.R == 2 && q.S == 2 && q.C == 8) ACIMG_DIRECT(2, 2, 8);
    else ACIMG_DIRECT(0, 0, 0);
#undef ACIMG_DIRECT
    return check_launch("direct_conv");
}

extern "C++" {
// ------------------------------------------------------------------------------------------
// MFMA form of the FEW-CHANNEL 3x3 / stride-1 / SAME layers (8 or 16 channels in, up to 32 out: the full-resolution
// layers of the RGB / spectrogram U-Nets; round 4).  The direct kernel above does these with packed fp32 FMAs at ~2x its
// VALU bound (224x298 8->8: 61 us for 27 us of bytes), every input value fetched nine times through the L1.  Here the taps
// are the GEMM's K axis: a pixel's CIN channels are one 16- or 32-byte run of a 16-bit plane, so the 8 k-values a lane
// holds of a 16x16x32 MFMA operand are ONE tap's channels of ONE pixel - a single ds_read_b128 at the tap's shift, four
// (two) taps per MFMA, 9 taps in 3 (5) MFMAs per term with the spare tap slots multiplied by zero weights.  A workgroup
// stages a 16 x 32 pixel tile WITH ITS HALO once (fp32 -> hi / lo planes on the way), keeps the whole weight image in
// registers (weights in the A slot: a lane's 4 accumulators are 4 consecutive output channels of one pixel, 16-byte
// stores), 3-term split product (fp32-class: f16 hi/lo forward, bf16 hi/lo for gradients).  MODE 0: forward - bias, raw
// output, batch-norm partials of conv + bias: one statistics row per workgroup; MODE 1: data gradient as a forward conv of
// gy with the flipped / transposed image, residual added.  Persistent workgroups, two per CU; XCD j walks the contiguous
// tile range [j * per, (j + 1) * per) so that neighbouring tiles share an L2; the next tile's loads are held in registers
// while the current one is multiplied.
// ------------------------------------------------------------------------------------------
struct FewParams {
    const float* X; int H, W, ldx;               // H, W: the OUTPUT grid (tiles); the tensor that is convolved:
    int Hin, Win, SH, SW, dil, pad_t, pad_l;     //   Hin x Win pixels, stored SH x SW (dil 2: zero-inserted view of a stride-2 gy)
    const char* Wimg; unsigned w_lo_off;         // 16-bit image [NOUTP][KTOT] (k = tap slot * CIN + c), hi plane; lo plane w_lo_off bytes on
    float* Y; int ldy, nout;                     // nout: real output channels (multiple of 4)
    const float* bias; const float* res; int ldres;
    float* stats; int stats_ld;                  // [gridDim.x][2][stats_ld] or null
    int tiles_x, tiles_y; long tiles, per;
    // the producer's deferred batch norm on load: x' = relu(x * a_scale[c] + a_shift[c]) for pixels INSIDE the image (the
    // conv's zero padding applies after the affine); null = x as stored
    const float* a_scale; const float* a_shift; int a_relu;
    // weight preparation
    const float* w; int ldw, wrows, cin, mode;
};
constexpr int FEW16_WGS = 512;
constexpr int few16_ktot(int cin) { return ((9 + 32 / cin - 1) / (32 / cin)) * 32; }

// w (fp32 HWIO, possibly the flipped / transposed view of a data gradient) -> the [NOUTP][KTOT] hi / lo image
template <typename TR>
__global__ __launch_bounds__(256) void few16_prepare_kernel(const FewParams p, int CIN, int NOUTP, typename TR::T* img) {
    typedef typename TR::T T;
    const int KTOT = few16_ktot(CIN);
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= NOUTP * KTOT) return;
    const int row = i / KTOT, k = i - row * KTOT;
    const int slot = k / CIN, c = k - slot * CIN;
    float v = 0.f;
    if (slot < 9 && row < p.nout && c < p.cin)
        v = p.mode == 0 ? p.w[((long)slot * p.wrows + c) * p.ldw + row] : p.w[((long)(8 - slot) * p.wrows + row) * p.ldw + c];
    v *= TR::WSCALE;
    const T h = (T)v;
    img[i] = h;
    img[(size_t)NOUTP * KTOT + i] = (T)(v - (float)h);
}

template <typename TR, int CIN, int NOUTP, int MODE, int CLOAD = CIN>
__global__ __launch_bounds__(512, (NOUTP == 32 && MODE == 0) ? 2 : 4) void conv_few16_kernel(const FewParams p) {
    typedef typename TR::V8 V8;
    constexpr int TH = 16, TW = 32, XH = TH + 2, XWV = TW + 2, XW = 36;
    constexpr int PB = CIN * 2;                        // bytes per pixel and plane
    constexpr int XPL = XH * XW * PB;
    constexpr int TPK = 32 / CIN;                      // taps per 32-deep MFMA
    constexpr int NKB = (9 + TPK - 1) / TPK;           // MFMAs per term and tile
    constexpr int KTOT = NKB * 32;
    constexpr int NT = NOUTP / 16;
    constexpr int NXL = (XH * XWV * (CLOAD / 4) + 511) / 512;
    static_assert((CIN == 8 || CIN == 16) && (CLOAD == CIN || (CIN == 8 && CLOAD == 4)), "few-channel instance");
    static_assert(2 * XPL >= 8 * 2 * NOUTP * 4, "statistics scratch");
    __shared__ __attribute__((aligned(16))) char xl[2 * XPL];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int li = lane & 15, g = lane >> 4;

    // the weight image -> registers (once per workgroup): lane (li, g) of (n, kb) holds row 16 n + li, k = 32 kb + 8 g .. + 7
    V8 wh[NT][NKB], wlo[NT][NKB];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            const size_t off = ((size_t)(n * 16 + li) * KTOT + kb * 32 + g * 8) * 2;
            wh[n][kb] = *reinterpret_cast<const V8*>(p.Wimg + off);
            wlo[n][kb] = *reinterpret_cast<const V8*>(p.Wimg + p.w_lo_off + off);
        }
    // this lane's tap shift of each MFMA: slot = kb * TPK + g / (4 / TPK); spare slots read tap 0 (their weights are zero)
    int boff[NKB];
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
        int slot = kb * TPK + (CIN == 8 ? g : g >> 1);
        if (slot > 8) slot = 0;
        const int r = slot / 3, q = slot - 3 * r;
        boff[kb] = (r * XW + q) * PB + (CIN == 8 ? 0 : (g & 1) * 16);
    }

    float4 rx[NXL];
    unsigned okm = 0;                          // which of rx[] came from inside the image (the affine applies to those only)
    // (a thread's items are always the same four channels: 512 is a multiple of CLOAD / 4)
    const float4 asc = p.a_scale ? *reinterpret_cast<const float4*>(p.a_scale + (tid % (CLOAD / 4)) * 4) : make_float4(1.f, 1.f, 1.f, 1.f);
    const float4 ash = p.a_scale ? *reinterpret_cast<const float4*>(p.a_shift + (tid % (CLOAD / 4)) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    auto load_tile = [&](long tile) {
        okm = 0;
        const int tx = (int)(tile % p.tiles_x);
        const long t2 = tile / p.tiles_x;
        const int ty = (int)(t2 % p.tiles_y);
        const long img = t2 / p.tiles_y;
        const float* xi = p.X + img * p.SH * p.SW * p.ldx;
#pragma unroll
        for (int k = 0; k < NXL; ++k) {
            const int i = tid + 512 * k;
            const int c4 = i % (CLOAD / 4), pix = i / (CLOAD / 4);
            const int row = pix / XWV, col = pix - row * XWV;
            const int iy = ty * TH + row - p.pad_t, ix = tx * TW + col - p.pad_l;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            bool ok = row < XH && (unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win;
            int sy = iy, sx = ix;
            if (p.dil == 2) {               // the zero-inserted view: only the even positions hold data
                ok = ok && !((iy | ix) & 1);
                sy >>= 1; sx >>= 1;
            }
            if (ok) {
                v = *reinterpret_cast<const float4*>(xi + ((long)sy * p.SW + sx) * p.ldx + c4 * 4);
                okm |= 1u << k;
            }
            rx[k] = v;
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int k = 0; k < NXL; ++k) {
            const int i = tid + 512 * k;
            const int c4 = i % (CLOAD / 4), pix = i / (CLOAD / 4);
            const int row = pix / XWV, col = pix - row * XWV;
            if (row < XH) {
                const int off = (row * XW + col) * PB + c4 * 8;
                uint2 hi, lo;
                split4<TR>(p.a_scale && ((okm >> k) & 1u) ? affine_relu4(rx[k], asc, ash, p.a_relu != 0) : rx[k], hi, lo);
                *reinterpret_cast<uint2*>(xl + off) = hi;
                *reinterpret_cast<uint2*>(xl + XPL + off) = lo;
            }
        }
    };
    if (CLOAD < CIN) {                      // 4 real channels in an 8-channel image: the upper half stays zero
        for (int i = tid; i < 2 * XH * XW; i += 512)
            *reinterpret_cast<uint2*>(xl + (i / (XH * XW)) * XPL + (i % (XH * XW)) * PB + 8) = make_uint2(0u, 0u);
    }

    f32x4 s1[NT], s2[NT], bv[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        s1[n] = s2[n] = f32x4{0.f, 0.f, 0.f, 0.f};
        bv[n] = (MODE == 0 && p.bias && n * 16 + 4 * g < p.nout) ? *reinterpret_cast<const f32x4*>(p.bias + n * 16 + 4 * g)
                                                                 : f32x4{0.f, 0.f, 0.f, 0.f};
    }

    // XCD j = blockIdx.x % 8 walks tiles [j * per, (j + 1) * per), its gridDim.x / 8 workgroups interleaved
    const int xcd = blockIdx.x & 7, nslot = gridDim.x >> 3;
    const long t_end = min((long)(xcd + 1) * p.per, p.tiles);
    long tile = (long)xcd * p.per + (blockIdx.x >> 3);
    if (tile < t_end) load_tile(tile);
    for (; tile < t_end; tile += nslot) {
        __syncthreads();                               // everyone has finished reading the previous tile
        store_tile();
        __syncthreads();
        const int tx = (int)(tile % p.tiles_x);
        const long t2 = tile / p.tiles_x;
        const int ty = (int)(t2 % p.tiles_y);
        const long img = t2 / p.tiles_y;
        if (tile + nslot < t_end) load_tile(tile + nslot);     // in flight while this tile is multiplied
        // wave wid: tile rows 2 wid, 2 wid + 1, both 16-pixel halves
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int urow = 2 * wid + (u >> 1), ucol = (u & 1) * 16;
            const int base = (urow * XW + ucol + li) * PB;
            f32x4 acc[NT];
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) {
                const V8 xh = *reinterpret_cast<const V8*>(xl + base + boff[kb]);
                const V8 xlo = *reinterpret_cast<const V8*>(xl + XPL + base + boff[kb]);
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    acc[n] = TR::mfma(wlo[n][kb], xh, acc[n]);
                    acc[n] = TR::mfma(wh[n][kb], xlo, acc[n]);
                    acc[n] = TR::mfma(wh[n][kb], xh, acc[n]);
                }
            }
            // lane (li, g) of acc[n] holds output pixel (row urow, column ucol + li), channels 16 n + 4 g .. + 3
            const int oy = ty * TH + urow, ox = tx * TW + ucol + li;
            if (oy < p.H && ox < p.W) {
                const long pix = (img * p.H + oy) * p.W + ox;
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    if (n * 16 + 4 * g < p.nout) {
                        f32x4 v = acc[n] * TR::OUTSCALE + bv[n];
                        if (MODE == 1) {
                            if (p.res) v += *reinterpret_cast<const f32x4*>(p.res + pix * p.ldres + n * 16 + 4 * g);
                        } else {
                            s1[n] += v;
                            s2[n] += v * v;
                        }
                        *reinterpret_cast<f32x4*>(p.Y + pix * p.ldy + n * 16 + 4 * g) = v;
                    }
                }
            }
        }
    }
    if (MODE == 0 && p.stats) {
        // the workgroup's statistics row: 16 pixel lanes by DPP, 8 waves through LDS, in wave order
        __syncthreads();
        float* red = reinterpret_cast<float*>(xl);     // [8][2][NOUTP]
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float a = row16_sum(s1[n][c]), b = row16_sum(s2[n][c]);
                if (li == 0) {
                    red[(wid * 2 + 0) * NOUTP + n * 16 + 4 * g + c] = a;
                    red[(wid * 2 + 1) * NOUTP + n * 16 + 4 * g + c] = b;
                }
            }
        __syncthreads();
        if (tid < 2 * NOUTP) {
            const int which = tid / NOUTP, n = tid % NOUTP;
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) t += red[(w * 2 + which) * NOUTP + n];
            if (n < p.nout) p.stats[((long)blockIdx.x * 2 + which) * p.stats_ld + n] = t;
        }
    }
}

// shapes the few-channel MFMA kernel takes (cin = channels of the tensor that is convolved, nout = channels written)
// cin = channels of the tensor that is convolved (4 forward only: an 8-channel image with a zero upper half), nout = written
static bool few16_channels(int cin, int nout, long pixels) {
    return (cin == 4 || cin == 8 || cin == 16) && nout >= 4 && nout <= 32 && (nout & 3) == 0 && !(cin != 8 && nout > 16) &&
           pixels >= 65536 && g_cfg.wgrad_halo;
}
static bool few16_fwd_shape(const AcimgConvDesc* d) {
    return d->R == 3 && d->S == 3 && d->stride == 1 && d->pad_t == 1 && d->pad_l == 1 && d->OH == d->H && d->OW == d->W &&
           few16_channels(d->C, d->K, (long)d->N * d->OH * d->OW) && d->act == ACIMG_ACT_NONE && d->ldx >= d->C;
}
// the data gradient of a 3x3 conv is a 3x3 / stride-1 conv of gy (K channels, padded to 4) into C channels: of gy itself
// (stride 1) or of its zero-inserted view (stride 2: the view is formed while the tile is staged, no copy)
static bool few16_dgrad_shape(const AcimgConvDesc* d) {
    const int ca = (d->K + 3) & ~3;
    return d->R == 3 && d->S == 3 && (d->stride == 1 || d->stride == 2) && ca != 4 && d->pad_t <= 2 && d->pad_l <= 2 &&
           few16_channels(ca, d->C, (long)d->N * d->H * d->W);
}
static size_t few16_ws_bytes(int cin, int nout) {
    return ((size_t)2 * (nout <= 16 ? 16 : 32) * few16_ktot(cin == 4 ? 8 : cin) * 2 + 255) & ~(size_t)255;
}

template <typename TR, int CIN, int NOUTP, int MODE, int CLOAD = CIN>
static int launch_few16(FewParams q, int N, void* ws, hipStream_t st) {
    constexpr int KTOT = few16_ktot(CIN);
    q.tiles_x = cdiv(q.W, 32); q.tiles_y = cdiv(q.H, 16);
    q.tiles = (long)N * q.tiles_x * q.tiles_y;
    q.per = (q.tiles + 7) / 8;
    typename TR::T* img = static_cast<typename TR::T*>(ws);
    q.Wimg = static_cast<const char*>(ws); q.w_lo_off = NOUTP * KTOT * 2;
    hipLaunchKernelGGL((few16_prepare_kernel<TR>), dim3(cdiv(NOUTP * KTOT, 256)), dim3(256), 0, st, q, CIN, NOUTP, img);
    hipLaunchKernelGGL((conv_few16_kernel<TR, CIN, NOUTP, MODE, CLOAD>), dim3(FEW16_WGS), dim3(512), 0, st, q);
    return check_launch("conv_few16");
}
// forward (MODE 0, f16 hi / lo) or data gradient (MODE 1, bf16 hi / lo) on the few-channel MFMA kernel
template <int MODE>
static int dispatch_few16(const FewParams& q, int N, int cin, int nout, void* ws, size_t ws_bytes, hipStream_t st) {
    if (!ws || ws_bytes < few16_ws_bytes(cin, nout) || !aligned16(ws)) return fail(ACIMG_EWORKSPACE, "conv_few16: workspace too small");
    typedef typename std::conditional<MODE == 0, SplitF16, SplitBF16>::type TR;
    if (cin == 4) return launch_few16<TR, 8, 16, MODE, 4>(q, N, ws, st);
    if (cin == 8) return nout <= 16 ? launch_few16<TR, 8, 16, MODE>(q, N, ws, st) : launch_few16<TR, 8, 32, MODE>(q, N, ws, st);
    return launch_few16<TR, 16, 16, MODE>(q, N, ws, st);
}
}  // extern "C++"

extern "C++" {
// ------------------------------------------------------------------------------------------
// 2x2 / stride-2 transposed conv with 32 input and 8 output channels (models/unet_architecture.py upsample_9 at 112x149 ->
// 224x298; round 4): patches do not overlap, so per INPUT pixel it is one 32 x 32 product - y'[(tap, k)] = W[(tap, k)][c] x[c],
// dx[c] = W^T[c][(tap, k)] gy'[(tap, k)] - and both operands can be loaded from global memory directly in MFMA layout: a
// lane's 8 k-values are 8 consecutive channels of one pixel (forward) or the 8 channels of one of the pixel's four output
// positions (data gradient).  No LDS, no scatter pass: forward stores are 64 contiguous bytes per pixel and output row (1 KiB
// runs per wave), data-gradient stores 128.  Weights (the 32 x 32 matrix, hi / lo) live in registers.  3-term split product:
// f16 hi / lo forward, bf16 hi / lo for the gradient.  The implicit GEMM with a scatter epilogue these replace ran at 90 /
// 65 us for 136 MB each way.
// MODE 0: y[n][2i + r][2j + s][k] = bias[k] + sum_c x[n][i][j][c] w[r][s][k][c]
// MODE 1: dx[n][i][j][c] = sum_{r,s,k} gy[n][2i + r][2j + s][k] w[r][s][k][c]   (optional ReLU mask on dx)
// ------------------------------------------------------------------------------------------
struct Patch2Params {
    const float* X; int ldx;       // MODE 0: x [N][H][W] pixels of ldx floats; MODE 1: gy [N][2H][2W] pixels of ldx floats
    float* Y; int ldy;             // MODE 0: y [N][2H][2W]; MODE 1: dx [N][H][W]
    const float* w; int ldw;       // [2][2][8][ldw >= 32]
    const float* bias; const float* mask; int ldmask; int act;
    int H, W; long pixels;         // the low-resolution grid
};

template <typename TR, int MODE>
__global__ __launch_bounds__(256) void patch2_32x8_kernel(const Patch2Params p) {
    typedef typename TR::V8 V8;
    typedef typename TR::T T;
    const int lane = threadIdx.x & 63, li = lane & 15, g = lane >> 4;
    // the weight matrix in the A slot, rows 16 n + li: MODE 0 rows are (tap, k) and the lane's 8 k-values channels 8 g ..;
    // MODE 1 rows are channels c and the lane's 8 k-values are (tap g, k 0 .. 7)
    V8 wh[2], wl[2];
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        float v[8];
        if (MODE == 0) {
            const float* src = p.w + (long)(16 * n + li) * p.ldw + 8 * g;
            const float4 a = *reinterpret_cast<const float4*>(src), b = *reinterpret_cast<const float4*>(src + 4);
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = p.w[(long)(g * 8 + k) * p.ldw + 16 * n + li];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float s_ = v[k] * TR::WSCALE;
            const T h = (T)s_;
            wh[n][k] = h;
            wl[n][k] = (T)(s_ - (float)h);
        }
    }
    f32x4 bv[2];
#pragma unroll
    for (int n = 0; n < 2; ++n)
        bv[n] = (MODE == 0 && p.bias) ? *reinterpret_cast<const f32x4*>(p.bias + 4 * (g & 1)) : f32x4{0.f, 0.f, 0.f, 0.f};
    const long groups = (p.pixels + 15) >> 4;
    const long nwaves = (long)gridDim.x * 4;
    for (long grp = (long)blockIdx.x * 4 + (threadIdx.x >> 6); grp < groups; grp += nwaves) {
        const long pix_raw = grp * 16 + li;
        const bool live = pix_raw < p.pixels;
        const long pix = live ? pix_raw : p.pixels - 1;
        const int j = (int)(pix % p.W);
        const long t = pix / p.W;
        const int i = (int)(t % p.H);
        const long img = t / p.H;
        // this lane's 8 values of the pixel operand
        const float* src = MODE == 0 ? p.X + pix * p.ldx + 8 * g
                                     : p.X + ((img * 2 * p.H + 2 * i + (g >> 1)) * (2L * p.W) + 2 * j + (g & 1)) * p.ldx;
        const float4 a = *reinterpret_cast<const float4*>(src), b = *reinterpret_cast<const float4*>(src + 4);
        uint2 h0, l0, h1, l1;
        split4<TR>(a, h0, l0);
        split4<TR>(b, h1, l1);
        const V8 xh = __builtin_bit_cast(V8, make_uint4(h0.x, h0.y, h1.x, h1.y));
        const V8 xl = __builtin_bit_cast(V8, make_uint4(l0.x, l0.y, l1.x, l1.y));
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
            acc = TR::mfma(wl[n], xh, acc);
            acc = TR::mfma(wh[n], xl, acc);
            acc = TR::mfma(wh[n], xh, acc);
            f32x4 v = acc * TR::OUTSCALE + bv[n];
            if (!live) continue;
            if (MODE == 0) {
                // rows 16 n + 4 g .. + 3 = tap 2 n + (g >> 1), channels 4 (g & 1) .. + 3: output pixel (2 i + n, 2 j + (g >> 1))
#pragma unroll
                for (int c = 0; c < 4; ++c) v[c] = apply_act(v[c], p.act);
                float* dst = p.Y + ((img * 2 * p.H + 2 * i + n) * (2L * p.W) + 2 * j + (g >> 1)) * p.ldy + 4 * (g & 1);
                *reinterpret_cast<f32x4*>(dst) = v;
            } else {
                if (p.mask) {
                    const f32x4 m = *reinterpret_cast<const f32x4*>(p.mask + pix * p.ldmask + 16 * n + 4 * g);
#pragma unroll
                    for (int c = 0; c < 4; ++c) v[c] = m[c] > 0.f ? v[c] : 0.f;
                }
                *reinterpret_cast<f32x4*>(p.Y + pix * p.ldy + 16 * n + 4 * g) = v;
            }
        }
    }
}

// Weight gradient of the same layer: dW[(tap, k)][c] = sum over input pixels of gy[n][2i + r][2j + s][k] x[n][i][j][c], a
// 32 x 32 matrix reduced over all pixels.  The pixels are the MFMA's K axis here, so a lane's 8 k-values are the SAME
// element of 8 consecutive pixels: 4-byte loads (16 lanes cover 64 contiguous bytes of a pixel, the texture addresser
// coalesces them), bf16 hi / lo on the way, 12 MFMAs per 32 pixels, the 32 x 32 tile in 16 accumulators per wave; the
// workgroup's sixteen waves are added through LDS in wave order into one slab per workgroup (slab_reduce_wide_kernel adds
// those in slab order: deterministic).  No LDS staging, no transposing reads.  136 MB in 127 us before (gather GEMM).
struct Patch2WgradParams {
    const float* X; int ldx;       // x [N][H][W][32]
    const float* G; int ldg;       // gy [N][2H][2W][8]
    float* out; int ldo;           // slabs [gridDim.x][32][ldo]
    float* db_part;                // [gridDim.x][8] partial bias gradients (sum of gy per channel) or null
    int H, W; long pixels;
};

__global__ __launch_bounds__(1024) void patch2_wgrad_32x8_kernel(const Patch2WgradParams p) {
    typedef SplitBF16 TR;
    typedef TR::V8 V8;
    typedef TR::T T;
    constexpr int NW = 16;             // waves per workgroup: one workgroup per CU, few slabs for the reduce launch to walk
    __shared__ float red[NW][32][33];
    __shared__ float redb[NW][64];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, li = lane & 15, g = lane >> 4;
    float sdb = 0.f;               // this lane's share of the bias gradient: every gy value it loads has channel li & 7
    f32x4 acc[2][2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    // row m = 16 mt + li of gy' is (tap, k) = (2 mt + (li >> 3), li & 7): its offset from the pixel's top-left output position
    long offa[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) offa[mt] = ((long)mt * 2 * p.W + (li >> 3)) * p.ldg + (li & 7);
    const long blocks = (p.pixels + 31) >> 5;
    const long nwaves = (long)gridDim.x * NW;
    for (long blk = (long)blockIdx.x * NW + wid; blk < blocks; blk += nwaves) {
        // this lane's eight pixels: blk * 32 + 8 g + t
        const long pix0 = blk * 32 + 8 * g;
        long pc = pix0 < p.pixels ? pix0 : p.pixels - 1;
        int j = (int)(pc % p.W);
        long t_ = pc / p.W;
        int i = (int)(t_ % p.H);
        long img = t_ / p.H;
        float av[2][8], bvv[2][8];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const bool live = pix0 + t < p.pixels;
            const float* gb = p.G + ((img * 2 * p.H + 2 * i) * (2L * p.W) + 2 * j) * p.ldg;
            const float* xb = p.X + ((img * p.H + i) * (long)p.W + j) * p.ldx;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) av[mt][t] = live ? gb[offa[mt]] : 0.f;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) bvv[nt][t] = live ? xb[16 * nt + li] : 0.f;
            if (live && pix0 + t + 1 < p.pixels) {       // the next pixel, by carry (no division)
                if (++j == p.W) {
                    j = 0;
                    if (++i == p.H) { i = 0; ++img; }
                }
            }
        }
        V8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                sdb += av[q][t];
                const T h = (T)av[q][t];
                ah[q][t] = h;
                al[q][t] = (T)(av[q][t] - (float)h);
                const T hb = (T)bvv[q][t];
                bh[q][t] = hb;
                bl[q][t] = (T)(bvv[q][t] - (float)hb);
            }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                acc[mt][nt] = TR::mfma(al[mt], bh[nt], acc[mt][nt]);
                acc[mt][nt] = TR::mfma(ah[mt], bl[nt], acc[mt][nt]);
                acc[mt][nt] = TR::mfma(ah[mt], bh[nt], acc[mt][nt]);
            }
    }
    // lane (li, g) of acc[mt][nt] holds rows 16 mt + 4 g .. + 3, column 16 nt + li
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[wid][16 * mt + 4 * g + r][16 * nt + li] = acc[mt][nt][r];
    redb[wid][lane] = sdb;
    __syncthreads();
    if (p.db_part && threadIdx.x < 8) {          // channel k: lanes k and k + 8 of every 16-lane row, every wave, in a fixed order
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w)
#pragma unroll
            for (int gg = 0; gg < 4; ++gg) t += redb[w][16 * gg + threadIdx.x] + redb[w][16 * gg + 8 + threadIdx.x];
        p.db_part[(long)blockIdx.x * 8 + threadIdx.x] = t;
    }
    float* slab = p.out + (long)blockIdx.x * 32 * p.ldo;
    {
        const int row = threadIdx.x >> 5, col = threadIdx.x & 31;      // 1024 threads = the 32 x 32 tile, waves added in order
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) t += red[w][row][col];
        slab[row * p.ldo + col] = t;
    }
}
static constexpr int PATCH2_WGRAD_WGS = 256;

static bool patch2_shape(const AcimgConvDesc* d) {
    return d->R == 2 && d->S == 2 && d->stride == 2 && d->C == 32 && d->K == 8 && d->OH == 2 * d->H && d->OW == 2 * d->W &&
           (long)d->N * d->H * d->W >= 65536 && g_cfg.wgrad_halo;
}
template <typename TR, int MODE>
static int launch_patch2(const Patch2Params& q, hipStream_t st) {
    const long groups = (q.pixels + 15) >> 4;
    long blocks = (groups + 3) / 4;
    if (blocks > 256 * 16) blocks = 256 * 16;          // sixteen 4-wave workgroups per CU, each wave walks its groups
    hipLaunchKernelGGL((patch2_32x8_kernel<TR, MODE>), dim3((unsigned)blocks), dim3(256), 0, st, q);
    return check_launch("patch2_32x8");
}
}  // extern "C++"

static int fwd_kiters(const AcimgConvDesc* d) {
    const bool rowrun = d->S > 1 && d->ldx == d->C;
    const int L = rowrun ? d->S * d->C : d->C;
    return (rowrun ? d->R : d->R * d->S) * cdiv(L, 32);
}

// rows of y covered by one statistics partial: the implicit-GEMM tile height, or 256 after split-K
static int stats_block_rows(const AcimgConvDesc* d) {
    const int M = d->N * d->OH * d->OW;
    TileCfg c = pick_cfg(M, d->K);
    return pick_splits(M, d->K, c, fwd_kiters(d)) > 1 ? 256 : c.bm;
}
int acimg_conv2d_stats_rows(const AcimgConvDesc* d) {
    if (few16_fwd_shape(d)) return FEW16_WGS;        // the few-channel MFMA kernel leaves one row per workgroup
    return cdiv((long)d->N * d->OH * d->OW, stats_block_rows(d));
}

int acimg_conv2d_fwd_tiling(const AcimgConvDesc* d, int* out) {
    if (!d || !out) return fail(ACIMG_EINVAL, "conv2d_fwd_tiling: null argument");
    const int M = d->N * d->OH * d->OW;
    TileCfg c = pick_cfg(M, d->K);
    out[0] = c.bm;
    out[1] = c.bn;
    out[2] = pick_splits(M, d->K, c, fwd_kiters(d));
    return ACIMG_OK;
}

int acimg_config_default(AcimgConfig* c) {
    if (!c) return fail(ACIMG_EINVAL, "config_default: null");
    *c = AcimgConfig{320, 768, 1, 128, 1, 0, 0, 1, 0, 1, 0, 0, 0, 1, 0, 0};
    return ACIMG_OK;
}

int acimg_configure(const AcimgConfig* c) {
    if (!c) return fail(ACIMG_EINVAL, "configure: null");
    if (c->splitk_cut < 0 || c->splitk_target < 1 || c->wgrad_minpix < 1 || c->tail_s < 0)
        return fail(ACIMG_EINVAL, "configure: negative / zero tuning value");
    if (c->trunk_persistent < 0 || c->trunk_persistent > 2) return fail(ACIMG_EINVAL, "configure: trunk_persistent is 0, 1 or 2");
    if (c->trunk_dma_pos < 0 || c->trunk_dma_pos > 1) return fail(ACIMG_EINVAL, "configure: trunk_dma_pos is 0 or 1");
    if (c->trunk_stagger < 0 || c->trunk_stagger > 100) return fail(ACIMG_EINVAL, "configure: trunk_stagger is a percentage");
    if (c->trunk_bk != 0 && c->trunk_bk != 32)
        return fail(ACIMG_EINVAL, "configure: trunk_bk must be 0 or 32 (the 64-deep K step was measured slower and removed)");
    if (c->trunk_ring < 0 || c->trunk_ring > 2) return fail(ACIMG_EINVAL, "configure: trunk_ring is 0, 1 or 2");
    if (c->trunk_ring_bm != 0 && c->trunk_ring_bm != 128 && c->trunk_ring_bm != 256)
        return fail(ACIMG_EINVAL, "configure: trunk_ring_bm must be 0 (per shape), 128 or 256");
    if (c->trunk_halo < 0 || c->trunk_halo > 2) return fail(ACIMG_EINVAL, "configure: trunk_halo is 0, 1 or 2");
    if (c->split3_tile_bm || c->split3_tile_bn) {
        const int bm = c->split3_tile_bm, bn = c->split3_tile_bn;
        if (!((bm == 128 && bn == 128) || (bm == 64 && bn == 128) || (bm == 128 && bn == 64)))
            return fail(ACIMG_EINVAL, "configure: split3 tile %dx%d is not an instantiated tile", bm, bn);
    }
    g_cfg = *c;
    return ACIMG_OK;
}

// a dense layer over at most 64 batch rows with a large weight matrix (csrc/skinny_kernel.hpp)
static bool skinny_shape(const AcimgConvDesc* d) {
    return d->R == 1 && d->S == 1 && d->H == 1 && d->W == 1 && d->OH == 1 && d->OW == 1 && d->stride == 1 &&
           d->pad_t == 0 && d->pad_l == 0 && d->N <= 64 && d->C >= 4096 && (d->C & 3) == 0 && (d->K & 3) == 0 &&
           (long)d->C * d->ldw * 4 < (1L << 31) && (long)d->N * d->ldx * 4 < (1L << 31);
}
static size_t skinny_fwd_ws_bytes(const AcimgConvDesc* d) {
    return skinny_shape(d) ? (size_t)cdiv(d->C, SKINNY_KS) * d->N * d->K * 4 : 0;
}

size_t acimg_conv2d_fwd_workspace(const AcimgConvDesc* d) {
    const size_t a = igemm_ws_bytes(d->N * d->OH * d->OW, d->K, fwd_kiters(d));
    const size_t b = d->C <= 16 && d->K <= 32 ? direct_ws_bytes(d->R, d->S, d->C, (d->K + 7) & ~7) : 0;
    const size_t c = skinny_fwd_ws_bytes(d);
    const size_t f = few16_fwd_shape(d) ? few16_ws_bytes(d->C, d->K) : 0;
    return std::max(std::max(a, f), std::max(b, c));
}

int acimg_conv2d_fwd(const AcimgConvDesc* d, const float* x, const float* w, const float* bias,
                     float* y, const float* in_scale, const float* in_shift, int in_relu,
                     float* stats, void* ws, size_t ws_bytes, void* tickets, void* stream) {
    int rc = check_desc(d, "conv2d_fwd");
    if (rc) return rc;
    if (d->ldw < d->K) return fail(ACIMG_EINVAL, "conv2d_fwd: ldw < K");
    if (skinny_shape(d) && !in_scale && !in_shift && !in_relu && !stats && ws && ws_bytes >= skinny_fwd_ws_bytes(d) && aligned16(x) &&
        aligned16(w) && aligned16(y) && aligned16(ws) && (!bias || aligned16(bias)) && (d->ldw & 3) == 0 && (d->ldx & 3) == 0 &&
        (d->ldy & 3) == 0) {
        // the VAE heads' dense layer: weight rows read once in whole lines, K slabs combined in slab order
        SkinnyParams q{};
        q.W = w; q.X = x; q.out = y; q.bias = bias; q.act = d->act; q.part = static_cast<float*>(ws);
        q.M = d->N; q.C = d->C; q.N = d->K; q.ldw = d->ldw; q.ldx = d->ldx; q.ldo = d->ldy;
        q.slabs = cdiv(d->C, SKINNY_KS);
        const dim3 grid(q.slabs, cdiv(d->K, 64));
        const int mb = cdiv(d->N, 16);
        if (mb == 1) hipLaunchKernelGGL(skinny_fwd_kernel<1>, grid, dim3(64), 0, (hipStream_t)stream, q);
        else if (mb == 2) hipLaunchKernelGGL(skinny_fwd_kernel<2>, grid, dim3(64), 0, (hipStream_t)stream, q);
        else if (mb == 3) hipLaunchKernelGGL(skinny_fwd_kernel<3>, grid, dim3(64), 0, (hipStream_t)stream, q);
        else hipLaunchKernelGGL(skinny_fwd_kernel<4>, grid, dim3(64), 0, (hipStream_t)stream, q);
        rc = check_launch("conv2d_fwd (skinny)");
        if (rc) return rc;
        hipLaunchKernelGGL(skinny_fwd_reduce_kernel, dim3(cdiv(d->N * (d->K / 4), 256)), dim3(256), 0, (hipStream_t)stream, q);
        return check_launch("conv2d_fwd (skinny reduce)");
    }
    if (few16_fwd_shape(d)) {
        // acimg_conv2d_stats_rows(d) promised one statistics row per workgroup of this kernel: no silent fallback
        if ((in_scale != nullptr) != (in_shift != nullptr) || (in_relu && !in_scale) || !aligned16(x) || !aligned16(y) || (d->ldy & 3) ||
            d->ldy < d->K || (bias && !aligned16(bias)) || (in_scale && (!aligned16(in_scale) || !aligned16(in_shift))))
            return fail(ACIMG_EINVAL, "conv2d_fwd: few-channel MFMA shape with half an input affine or unaligned operands");
        FewParams q{};
        q.a_scale = in_scale; q.a_shift = in_shift; q.a_relu = in_relu;
        q.X = x; q.H = d->H; q.W = d->W; q.ldx = d->ldx; q.Y = y; q.ldy = d->ldy; q.nout = d->K; q.bias = bias;
        q.Hin = q.SH = d->H; q.Win = q.SW = d->W; q.dil = 1; q.pad_t = 1; q.pad_l = 1;
        q.stats = stats; q.stats_ld = d->ldw;
        q.w = w; q.ldw = d->ldw; q.wrows = d->C; q.cin = d->C; q.mode = 0;
        return dispatch_few16<0>(q, d->N, d->C, d->K, ws, ws_bytes, (hipStream_t)stream);
    }
    if (direct_ok(d->C, d->K, d->ldy, 0, y, bias, nullptr, in_scale != nullptr, nullptr) &&
        (long)d->N * d->OH * d->OW >= 65536) {
        DirectParams q{};
        q.x = x; q.ldx = d->ldx; q.H = d->H; q.W = d->W; q.C = d->C;
        q.y = y; q.ldy = d->ldy; q.OH = d->OH; q.OW = d->OW; q.K = d->K;
        q.R = d->R; q.S = d->S; q.stride = d->stride; q.pad_t = d->pad_t; q.pad_l = d->pad_l;
        q.w = w; q.ldw = d->ldw; q.mode = 0; q.wrows = d->C; q.bias = bias; q.act = d->act;
        q.M = (long)d->N * d->OH * d->OW;
        const bool fuse_stats = stats && stats_block_rows(d) == 256 && d->act == ACIMG_ACT_NONE;
        if (fuse_stats) { q.stats = stats; q.stats_ld = d->ldw; }
        rc = launch_direct(q, ws, ws_bytes, (hipStream_t)stream);
        if (!rc && stats && !fuse_stats) {   // batch-norm partials of y = conv + bias, in acimg_conv2d_stats_rows(d) row blocks
            hipLaunchKernelGGL(partial_stats_kernel, dim3(acimg_conv2d_stats_rows(d)), dim3(256), 0, (hipStream_t)stream,
                               y, d->ldy, (int)q.M, d->K, stats, d->ldw, stats_block_rows(d));
            rc = check_launch("partial_stats");
        }
        return rc;
    }
    IgemmParams p{};
    p.A = x; p.H = d->H; p.W = d->W; p.C = d->C; p.lda = d->ldx;
    p.OH = d->OH; p.OW = d->OW; p.R = d->R; p.S = d->S; p.stride = d->stride;
    p.pad_t = d->pad_t; p.pad_l = d->pad_l;
    p.M = d->N * d->OH * d->OW;
    p.rowrun = (d->S > 1 && d->ldx == d->C) ? 1 : 0;
    p.a_scale = in_scale; p.a_shift = in_shift; p.a_relu = in_relu;
    p.B = w; p.ldb = d->ldw; p.Nld = d->ldw; p.Ngemm = d->K; p.tap_stride = 0; p.flip = 0;
    p.e.Y = y; p.e.ldy = d->ldy; p.e.M = p.M; p.e.Nstore = d->K; p.e.bias = bias; p.e.act = d->act;
    p.e.stats = stats; p.e.stats_ld = d->ldw;
    return launch_igemm(p, false, ws, ws_bytes, tickets, (hipStream_t)stream);
}

// the sub-pixel form pays from 16 output channels on: with 8 (the full-resolution layers) its scatter writes 16-byte
// pieces of 32-byte pixels and the zero-inserted copy + the direct few-channel kernel is faster (224x298 8->8 3x3/2 data
// gradient: 99 us against 144 us; profiles/r04/op_report_unet_rgb_bf16_r04g_subpixel.txt)
static bool subpixel_ok(int stride, int Ko, int Kin) { return stride == 2 && (Ko & 3) == 0 && (Kin & 3) == 0 && Ko >= 16; }
static size_t subpixel_ws_bytes(int N, int YH, int YW, int oy0, int ox0, int R, int S, int Kin, int Ko) {
    const int U = (R + 1) / 2, V = (S + 1) / 2;
    const int AH = (YH - 1 - oy0) / 2 + 1, AW = (YW - 1 - ox0) / 2 + 1;
    const size_t wc = ((size_t)U * V * Kin * 4 * Ko * 4 + 255) & ~(size_t)255;
    const size_t a = igemm_ws_bytes(N * AH * AW, 4 * Ko, U * V * cdiv(Kin, 32));
    const size_t b = igemm_ws_bytes(N * AH * AW, 4 * Ko, U * cdiv(V * Kin, 32));
    return wc + (a > b ? a : b);
}
// A: [N][aH][aW][Kin] (pixel stride lda); w[r][s][ko][kin] with row pitch ldw; Y: [N][YH][YW] pixels of ldy floats
static int launch_subpixel(const float* A, int N, int aH, int aW, int Kin, int lda, const float* w, int R, int S, int ldw,
                           int Ko, float* Y, int ldy, int YH, int YW, int oy0, int ox0, const float* bias, const float* res,
                           int ldres, const float* mask, int ldmask, int act, void* ws, size_t ws_bytes, void* tickets,
                           hipStream_t st, const char* what) {
    const int U = (R + 1) / 2, V = (S + 1) / 2;
    const int AH = (YH - 1 - oy0) / 2 + 1, AW = (YW - 1 - ox0) / 2 + 1;
    const int rows = U * V * Kin, ncol = 4 * Ko;
    const size_t wcb = ((size_t)rows * ncol * 4 + 255) & ~(size_t)255;
    if (!ws || ws_bytes < wcb || !aligned16(ws)) return fail(ACIMG_EWORKSPACE, "%s: workspace too small for the sub-pixel weights", what);
    if ((Ko & 3) || (Kin & 3)) return fail(ACIMG_EINVAL, "%s: channel counts must be multiples of 4", what);
    float* wc = static_cast<float*>(ws);
    hipLaunchKernelGGL(subpixel_weights_kernel, dim3(cdiv((long)rows * ncol, 256)), dim3(256), 0, st, w, R, S, Ko, Kin, ldw, U, V, wc,
                       rows * ncol);
    int rc = check_launch("subpixel_weights");
    if (rc) return rc;
    IgemmParams p{};
    p.A = A; p.H = aH; p.W = aW; p.C = Kin; p.lda = lda; p.OH = AH; p.OW = AW;
    p.R = U; p.S = V; p.stride = 1; p.pad_t = U - 1; p.pad_l = V - 1;
    p.M = N * AH * AW;
    p.rowrun = (V > 1 && lda == Kin) ? 1 : 0;
    p.B = wc; p.ldb = ncol; p.Nld = ncol; p.Ngemm = ncol; p.tap_stride = 0; p.flip = 0;
    p.e.Y = Y; p.e.ldy = ldy; p.e.M = p.M; p.e.Nstore = ncol; p.e.bias = bias; p.e.act = act;
    p.e.res = res; p.e.ldres = ldres; p.e.mask = mask; p.e.ldmask = ldmask;
    p.e.scatter = 2; p.e.Ko = Ko; p.e.Sq = 2; p.e.sc = 2; p.e.YH = YH; p.e.YW = YW; p.e.AH = AH; p.e.AW = AW;
    p.e.oy0 = oy0; p.e.ox0 = ox0;
    return launch_igemm(p, false, static_cast<char*>(ws) + wcb, ws_bytes - wcb, tickets, st);
}

static bool dgrad_is_patch(const AcimgConvDesc* d) {
    return d->stride > 1 && d->stride == d->R && d->stride == d->S && !d->pad_t && !d->pad_l &&
           d->OH * d->stride == d->H && d->OW * d->stride == d->W;
}
static size_t dilated_bytes(int N, int H, int W, int C, int s) {
    return ((size_t)N * ((H - 1) * s + 1) * ((W - 1) * s + 1) * C * 4 + 255) & ~(size_t)255;
}

// data gradient of a 3x3 / stride-1 / SAME layer with 32 output and 4 - 16 input channels (configs[1]: 112x149 8 -> 32) through
// the fp32 entry: a conv of the 32-channel gy on the 16-row instance of the halo kernel (defined with conv_halo16_kernel)
static bool dgrad_halo16_narrow_shape(const AcimgConvDesc* d) {
    return d->R == 3 && d->S == 3 && d->stride == 1 && d->pad_t == 1 && d->pad_l == 1 && d->OH == d->H && d->OW == d->W &&
           d->K == 32 && d->C <= 16 && (d->C & 3) == 0 && (long)d->N * d->H * d->W >= 65536 && g_cfg.wgrad_halo;
}
static constexpr size_t DGRAD_HALO16_NARROW_WS = 2 * 16 * 288 * 2;
static int dgrad_halo16_narrow(const AcimgConvDesc* d, const float* gy, int ldgy, const float* w, float* dx, int lddx,
                               const float* residual, int ldres, const float* mask, int ldmask, void* ws, hipStream_t st);

size_t acimg_conv2d_dgrad_workspace(const AcimgConvDesc* d) {
    if (dgrad_halo16_narrow_shape(d)) return DGRAD_HALO16_NARROW_WS + 256;
    if (dgrad_is_patch(d)) return igemm_ws_bytes(d->N * d->OH * d->OW, d->R * d->S * d->C, cdiv(up4(d->K), 32));
    const int ca = up4(d->K);
    if (subpixel_ok(d->stride, d->C, ca))   // sub-pixel form: combined weights + the GEMM's own split-K slabs
        return subpixel_ws_bytes(d->N, d->H, d->W, -d->pad_t, -d->pad_l, d->R, d->S, ca, d->C);
    if (d->stride > 1)   // zero-inserted copy of gy, then the stride-1 path
        return dilated_bytes(d->N, d->OH, d->OW, ca, d->stride) +
               igemm_ws_bytes(d->N * d->H * d->W, d->C, d->R * d->S * cdiv(ca, 32)) +
               igemm_ws_bytes(d->N * d->H * d->W, d->C, d->R * cdiv(d->S * ca, 32)) +
               (ca <= 16 && d->C <= 32 ? std::max(direct_ws_bytes(d->R, d->S, ca, (d->C + 7) & ~7), few16_ws_bytes(ca, d->C)) : 0);
    // rowrun depends on ldgy, unknown here: per-tap kiters is the larger bound for splits
    return igemm_ws_bytes(d->N * d->H * d->W, d->C, d->R * d->S * cdiv(ca, 32)) +
           igemm_ws_bytes(d->N * d->H * d->W, d->C, d->R * cdiv(d->S * ca, 32)) +
           (ca <= 16 && d->C <= 32 ? std::max(direct_ws_bytes(d->R, d->S, ca, (d->C + 7) & ~7), few16_ws_bytes(ca, d->C)) : 0);
}

int acimg_conv2d_dgrad(const AcimgConvDesc* d, const float* gy, int ldgy, const float* w,
                       float* dx, int lddx, const float* residual, int ldres, const float* mask,
                       int ldmask, void* ws, size_t ws_bytes, void* tickets, void* stream) {
    int rc = check_desc(d, "conv2d_dgrad");
    if (rc) return rc;
    const int ca = up4(d->K);
    if (ca > ldgy || ca > d->ldw || (ldgy & 3)) return fail(ACIMG_EINVAL, "conv2d_dgrad: padded K=%d exceeds ldgy=%d/ldw=%d", ca, ldgy, d->ldw);
    if (skinny_shape(d) && aligned16(gy) && aligned16(w) && (d->ldw & 3) == 0) {
        // a dense layer over a few batch rows (the 28 416 -> 300 VAE heads): the weight matrix is read once, in rows
        SkinnyParams q{};
        q.W = w; q.G = gy; q.out = dx; q.res = residual; q.mask = mask;
        q.M = d->N; q.C = d->C; q.N = ca;
        q.ldw = d->ldw; q.ldg = ldgy; q.ldo = lddx > 0 ? lddx : d->ldx; q.ldres = ldres; q.ldmask = ldmask;
        if (q.ldo < d->C) return fail(ACIMG_EINVAL, "conv2d_dgrad: lddx < C");
        const dim3 grid(cdiv(cdiv(d->C, 16), 4));
        const int mb = cdiv(d->N, 16);
        if (mb == 1) hipLaunchKernelGGL(skinny_dgrad_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, q);
        else if (mb == 2) hipLaunchKernelGGL(skinny_dgrad_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, q);
        else if (mb == 3) hipLaunchKernelGGL(skinny_dgrad_kernel<3>, grid, dim3(256), 0, (hipStream_t)stream, q);
        else hipLaunchKernelGGL(skinny_dgrad_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, q);
        return check_launch("conv2d_dgrad (skinny)");
    }
    IgemmParams p{};
    p.A = gy; p.C = ca; p.lda = ldgy;
    p.B = w; p.ldb = d->ldw;
    if (lddx <= 0) lddx = d->ldx;
    if (lddx < d->C) return fail(ACIMG_EINVAL, "conv2d_dgrad: lddx < C");
    p.e.Y = dx; p.e.ldy = lddx; p.e.res = residual; p.e.ldres = ldres; p.e.mask = mask; p.e.ldmask = ldmask;
    p.e.act = ACIMG_ACT_NONE;
    if (subpixel_ok(d->stride, d->C, ca) && !dgrad_is_patch(d))
        // dx[2 oy - pad_t + r][2 ox - pad_l + s][c] += gy[oy][ox][k] w[r][s][c][k]: the sub-pixel form over gy's own grid
        return launch_subpixel(gy, d->N, d->OH, d->OW, ca, ldgy, w, d->R, d->S, d->ldw, d->C, dx, lddx, d->H, d->W, -d->pad_t,
                               -d->pad_l, nullptr, residual, ldres, mask, ldmask, ACIMG_ACT_NONE, ws, ws_bytes, tickets,
                               (hipStream_t)stream, "conv2d_dgrad");
    if (dgrad_halo16_narrow_shape(d) && aligned16(gy) && aligned16(dx) && (lddx & 3) == 0 && ws && aligned16(ws) &&
        ws_bytes >= DGRAD_HALO16_NARROW_WS && (!residual || ((ldres & 3) == 0 && aligned16(residual))) &&
        (!mask || ((ldmask & 3) == 0 && aligned16(mask))))
        return dgrad_halo16_narrow(d, gy, ldgy, w, dx, lddx, residual, ldres, mask, ldmask, ws, (hipStream_t)stream);
    if (few16_dgrad_shape(d) && !subpixel_ok(d->stride, d->C, ca) && !mask && aligned16(gy) && aligned16(dx) && (lddx & 3) == 0 &&
        ws && aligned16(ws) && ws_bytes >= few16_ws_bytes(ca, d->C) && (!residual || ((ldres & 3) == 0 && aligned16(residual)))) {
        // dx[h][w][c] = sum gy1[h - (2 - pad_t) + r'][w - (2 - pad_l) + s'][k] W[2 - r'][2 - s'][c][k], gy1 = gy (stride 1) or
        // its zero-inserted view (stride 2), zero outside
        FewParams q{};
        q.X = gy; q.ldx = ldgy; q.SH = d->OH; q.SW = d->OW; q.dil = d->stride;
        q.Hin = (d->OH - 1) * d->stride + 1; q.Win = (d->OW - 1) * d->stride + 1;
        q.pad_t = 2 - d->pad_t; q.pad_l = 2 - d->pad_l;
        q.H = d->H; q.W = d->W; q.Y = dx; q.ldy = lddx; q.nout = d->C;
        q.res = residual; q.ldres = ldres;
        q.w = w; q.ldw = d->ldw; q.wrows = d->C; q.cin = d->K; q.mode = 1;
        return dispatch_few16<1>(q, d->N, ca, d->C, ws, ws_bytes, (hipStream_t)stream);
    }
    if (d->stride > 1 && !dgrad_is_patch(d)) {
        // general stride: the strided conv is a subsampled stride-1 conv, so its data gradient is the stride-1
        // data gradient of the zero-inserted gy
        const size_t db = dilated_bytes(d->N, d->OH, d->OW, ca, d->stride);
        if (ws_bytes < db || !ws) return fail(ACIMG_EWORKSPACE, "conv2d_dgrad: workspace too small for the dilated gradient");
        const int OH1 = (d->OH - 1) * d->stride + 1, OW1 = (d->OW - 1) * d->stride + 1;
        const long opix = (long)d->N * OH1 * OW1;
        if (opix * ca >= (1L << 31)) return fail(ACIMG_EINVAL, "conv2d_dgrad: dilated gradient exceeds 2^31 elements");
        hipLaunchKernelGGL(dilate2d_kernel, dim3(cdiv(opix * (ca / 4), 256)), dim3(256), 0, (hipStream_t)stream, gy, ldgy,
                           static_cast<float*>(ws), opix, d->OH, d->OW, OH1, OW1, ca, d->stride);
        rc = check_launch("dilate2d");
        if (rc) return rc;
        AcimgConvDesc d1 = *d;
        d1.stride = 1; d1.OH = OH1; d1.OW = OW1;
        return acimg_conv2d_dgrad(&d1, static_cast<const float*>(ws), ca, w, dx, lddx, residual, ldres, mask, ldmask,
                                  static_cast<char*>(ws) + db, ws_bytes - db, tickets, stream);
    }
    if (d->stride == 1 && direct_ok(ca, d->C, lddx, ldres, dx, nullptr, residual, false, mask) &&
        (long)d->N * d->H * d->W >= 65536) {
        DirectParams q{};
        q.x = gy; q.ldx = ldgy; q.H = d->OH; q.W = d->OW; q.C = ca;
        q.y = dx; q.ldy = lddx; q.OH = d->H; q.OW = d->W; q.K = d->C;
        q.R = d->R; q.S = d->S; q.stride = 1; q.pad_t = d->R - 1 - d->pad_t; q.pad_l = d->S - 1 - d->pad_l;
        q.w = w; q.ldw = d->ldw; q.mode = 1; q.wrows = d->C; q.act = ACIMG_ACT_NONE;
        q.res = residual; q.ldres = ldres;
        q.M = (long)d->N * d->H * d->W;
        return launch_direct(q, ws, ws_bytes, (hipStream_t)stream);
    }
    if (d->stride == 1) {
        // dx[h,w,c] = sum_{r',s',k} gy[h-(R-1-pt)+r', w-(S-1-pl)+s', k] * W[R-1-r'][S-1-s'][c][k]
        p.H = d->OH; p.W = d->OW; p.OH = d->H; p.OW = d->W;
        p.R = d->R; p.S = d->S; p.stride = 1;
        p.pad_t = d->R - 1 - d->pad_t; p.pad_l = d->S - 1 - d->pad_l;
        p.M = d->N * d->H * d->W;
        p.rowrun = (d->S > 1 && ldgy == ca) ? 1 : 0;
        p.tap_stride = (long)d->C * d->ldw; p.flip = 1;
        p.Ngemm = d->C;
        p.e.M = p.M; p.e.Nstore = d->C;
    } else {
        // patch scatter: rows = output pixels, columns = (tap, c)
        p.H = d->OH; p.W = d->OW; p.OH = d->OH; p.OW = d->OW;
        p.R = 1; p.S = 1; p.stride = 1; p.pad_t = 0; p.pad_l = 0;
        p.M = d->N * d->OH * d->OW;
        p.rowrun = 0; p.tap_stride = 0; p.flip = 0;
        p.Ngemm = d->R * d->S * d->C;
        p.e.M = p.M; p.e.Nstore = p.Ngemm;
        p.e.scatter = 1; p.e.Ko = d->C; p.e.Sq = d->S; p.e.sc = d->stride;
        p.e.YH = d->H; p.e.YW = d->W; p.e.AH = d->OH; p.e.AW = d->OW;
    }
    return launch_igemm(p, true, ws, ws_bytes, tickets, (hipStream_t)stream);
}

size_t acimg_conv2d_wgrad_workspace(const AcimgConvDesc* d) {
    return wgrad_ws_bytes(d->N * d->OH * d->OW, d->R * d->S * d->C, up4(d->K), d->ldw) + colsum_ws_bytes(up4(d->K));
}

int acimg_conv2d_wgrad(const AcimgConvDesc* d, const float* x, const float* gy, int ldgy,
                       float* dw, float* db, void* ws, size_t ws_bytes, void* stream) {
    int rc = check_desc(d, "conv2d_wgrad");
    if (rc) return rc;
    const int kp = up4(d->K);
    if (kp > ldgy || kp > d->ldw) return fail(ACIMG_EINVAL, "conv2d_wgrad: padded K exceeds ldgy/ldw");
    if (skinny_shape(d) && aligned16(x) && aligned16(gy) && aligned16(dw) && (!db || aligned16(db)) && (d->ldw & 3) == 0 &&
        (ldgy & 3) == 0 && (d->ldx & 3) == 0) {
        // the same dense layer's weight gradient: every weight row is written once, 256 contiguous bytes per wave
        SkinnyParams q{};
        q.X = x; q.G = gy; q.out = dw; q.db = db;
        q.M = d->N; q.C = d->C; q.N = kp;
        q.ldw = d->ldw; q.ldg = ldgy; q.ldx = d->ldx;
        const int ngroups = cdiv(kp, 64);
        const int tasks = cdiv(d->C, 64) * ngroups;
        hipLaunchKernelGGL(skinny_wgrad_kernel, dim3(cdiv(tasks, 4)), dim3(256), 0, (hipStream_t)stream, q, ngroups);
        return check_launch("conv2d_wgrad (skinny)");
    }
    WgradParams p{};
    p.X = x; p.H = d->H; p.W = d->W; p.C = d->C; p.ldx = d->ldx;
    p.OH = d->OH; p.OW = d->OW; p.R = d->R; p.S = d->S; p.stride = d->stride; p.pad_t = d->pad_t; p.pad_l = d->pad_l;
    p.M = d->N * d->OH * d->OW; p.KK = d->R * d->S * d->C;
    p.G = gy; p.ldg = ldgy; p.Ngemm = kp; p.Nld = kp; p.ldo = d->ldw;
    return launch_wgrad(p, dw, db, ws, ws_bytes, (hipStream_t)stream);
}

size_t acimg_deconv_workspace(const AcimgConvDesc* d) {
    size_t a = igemm_ws_bytes(d->N * d->H * d->W, d->R * d->S * d->K, cdiv(d->C, 32));
    size_t b = igemm_ws_bytes(d->N * d->H * d->W, d->C, d->R * d->S * cdiv(up4(d->K), 32));
    size_t c = wgrad_ws_bytes(d->N * d->H * d->W, d->R * d->S * up4(d->K), d->C, d->ldw) + colsum_ws_bytes(up4(d->K));
    size_t m = a > b ? a : b;
    m = m > c ? m : c;
    if (patch2_shape(d)) {                      // weight gradient of the pointwise form: one 32 x ldw slab per workgroup
        const size_t pw = (size_t)PATCH2_WGRAD_WGS * (32 * d->ldw + 8) * sizeof(float);
        m = m > pw ? m : pw;
    }
    if (up4(d->K) <= 16 && d->C <= 32) {        // data gradient on the direct few-channel kernel
        const size_t dd = direct_ws_bytes(d->R, d->S, up4(d->K), (d->C + 7) & ~7);
        m = m > dd ? m : dd;
    }
    if ((d->R > d->stride || d->S > d->stride) && subpixel_ok(d->stride, d->K, d->C)) {   // forward in the sub-pixel form
        const size_t sp = subpixel_ws_bytes(d->N, d->OH, d->OW, 0, 0, d->R, d->S, d->C, d->K);
        return m > sp ? m : sp;
    }
    if (d->R > d->stride || d->S > d->stride)   // forward goes through a zero-inserted copy of x
        m += dilated_bytes(d->N, d->H, d->W, d->C, d->stride) +
             igemm_ws_bytes(d->N * d->OH * d->OW, d->K, d->R * d->S * cdiv(d->C, 32)) +
             igemm_ws_bytes(d->N * d->OH * d->OW, d->K, d->R * cdiv(d->S * d->C, 32));
    return m;
}

int acimg_deconv_fwd(const AcimgConvDesc* d, const float* x, const float* w, const float* bias,
                     float* y, void* ws, size_t ws_bytes, void* tickets, void* stream) {
    int rc = check_desc(d, "deconv_fwd");
    if (rc) return rc;
    if (d->ldw < d->C || (d->K & 3)) return fail(ACIMG_EINVAL, "deconv_fwd: ldw<C or K%%4");
    if (d->R > d->stride || d->S > d->stride) {
        // overlapping patches (tf conv2d_transpose VALID: OH = (H-1)*stride + R): y = stride-1 "full" correlation
        // of the zero-inserted x with the flipped kernel = the data gradient of the stride-1 VALID conv
        // [OH,OW,K] -> [(H-1)s+1, (W-1)s+1, C] whose HWIO kernel is this layer's [R][S][K][C]
        if (d->OH != (d->H - 1) * d->stride + d->R || d->OW != (d->W - 1) * d->stride + d->S)
            return fail(ACIMG_EINVAL, "deconv_fwd: kernel>stride needs OH=(H-1)*stride+R");
        if (subpixel_ok(d->stride, d->K, d->C))
            // y[2 h + r][2 w + s][k] += x[h][w][c] W[r][s][k][c]: the sub-pixel form over x's own grid (every output
            // pixel belongs to exactly one parity class: written once, bias included)
            return launch_subpixel(x, d->N, d->H, d->W, d->C, d->ldx, w, d->R, d->S, d->ldw, d->K, y, d->ldy, d->OH, d->OW, 0, 0,
                                   bias, nullptr, 0, nullptr, 0, d->act, ws, ws_bytes, tickets, (hipStream_t)stream,
                                   "deconv_fwd");
        const size_t db = dilated_bytes(d->N, d->H, d->W, d->C, d->stride);
        if (ws_bytes < db || !ws) return fail(ACIMG_EWORKSPACE, "deconv_fwd: workspace too small for the dilated input");
        const int H1 = (d->H - 1) * d->stride + 1, W1 = (d->W - 1) * d->stride + 1;
        const long opix = (long)d->N * H1 * W1;
        hipLaunchKernelGGL(dilate2d_kernel, dim3(cdiv(opix * (d->C / 4), 256)), dim3(256), 0, (hipStream_t)stream, x, d->ldx,
                           static_cast<float*>(ws), opix, d->H, d->W, H1, W1, d->C, d->stride);
        rc = check_launch("dilate2d");
        if (rc) return rc;
        IgemmParams p{};
        p.A = static_cast<const float*>(ws); p.C = d->C; p.lda = d->C;
        p.B = w; p.ldb = d->ldw;
        p.H = H1; p.W = W1; p.OH = d->OH; p.OW = d->OW;
        p.R = d->R; p.S = d->S; p.stride = 1; p.pad_t = d->R - 1; p.pad_l = d->S - 1;
        p.M = d->N * d->OH * d->OW;
        p.rowrun = d->S > 1 ? 1 : 0;
        p.tap_stride = (long)d->K * d->ldw; p.flip = 1;
        p.Ngemm = d->K;
        p.e.Y = y; p.e.ldy = d->ldy; p.e.M = p.M; p.e.Nstore = d->K; p.e.bias = bias; p.e.act = d->act;
        return launch_igemm(p, true, static_cast<char*>(ws) + db, ws_bytes - db, tickets, (hipStream_t)stream);
    }
    if (d->OH != d->H * d->stride || d->OW != d->W * d->stride)
        return fail(ACIMG_EINVAL, "deconv_fwd: kernel<=stride needs OH=H*stride");
    if (patch2_shape(d) && aligned16(x) && aligned16(y) && aligned16(w) && (d->ldx & 3) == 0 && (d->ldy & 3) == 0 && (d->ldw & 3) == 0 &&
        (!bias || aligned16(bias))) {
        Patch2Params q{};
        q.X = x; q.ldx = d->ldx; q.Y = y; q.ldy = d->ldy; q.w = w; q.ldw = d->ldw; q.bias = bias; q.act = d->act;
        q.H = d->H; q.W = d->W; q.pixels = (long)d->N * d->H * d->W;
        return launch_patch2<SplitF16, 0>(q, (hipStream_t)stream);
    }
    IgemmParams p{};
    p.A = x; p.H = d->H; p.W = d->W; p.C = d->C; p.lda = d->ldx; p.OH = d->H; p.OW = d->W;
    p.R = 1; p.S = 1; p.stride = 1; p.M = d->N * d->H * d->W; p.rowrun = 0;
    p.B = w; p.ldb = d->ldw; p.tap_stride = 0; p.flip = 0; p.Ngemm = d->R * d->S * d->K;
    p.e.Y = y; p.e.ldy = d->ldy; p.e.M = p.M; p.e.Nstore = p.Ngemm; p.e.bias = bias; p.e.act = d->act;
    p.e.scatter = 1; p.e.Ko = d->K; p.e.Sq = d->S; p.e.sc = d->stride;
    p.e.YH = d->OH; p.e.YW = d->OW; p.e.AH = d->H; p.e.AW = d->W;
    rc = launch_igemm(p, true, ws, ws_bytes, tickets, (hipStream_t)stream);
    if (rc) return rc;
    if (d->R < d->stride || d->S < d->stride) {
        const long pixels = (long)d->N * d->OH * d->OW;
        hipLaunchKernelGGL(deconv_gap_fill_kernel, dim3(cdiv(pixels * (d->K / 4), 256)), dim3(256), 0,
                           (hipStream_t)stream, y, d->ldy, bias, pixels, d->OH, d->OW, d->K, d->R, d->S, d->stride);
        rc = check_launch("deconv_gap_fill");
    }
    return rc;
}

int acimg_deconv_dgrad(const AcimgConvDesc* d, const float* gy, int ldgy, const float* w,
                       float* dx, const float* mask, int ldmask, void* ws, size_t ws_bytes,
                       void* tickets, void* stream) {
    int rc = check_desc(d, "deconv_dgrad");
    if (rc) return rc;
    const int ca = up4(d->K);
    if (ca > ldgy || (ldgy & 3) || ca != d->K) return fail(ACIMG_EINVAL, "deconv_dgrad: K must be a multiple of 4 and <= ldgy");
    // dx[n,h,w,c] = sum_{r,s,k} gy[n, h*stride+r, w*stride+s, k] * W[r][s][k][c]  (a strided conv)
    if (patch2_shape(d) && aligned16(gy) && aligned16(dx) && aligned16(w) && (d->ldx & 3) == 0 && (d->ldw & 3) == 0 &&
        (!mask || (aligned16(mask) && (ldmask & 3) == 0))) {
        Patch2Params q{};
        q.X = gy; q.ldx = ldgy; q.Y = dx; q.ldy = d->ldx; q.w = w; q.ldw = d->ldw; q.mask = mask; q.ldmask = ldmask;
        q.H = d->H; q.W = d->W; q.pixels = (long)d->N * d->H * d->W;
        return launch_patch2<SplitBF16, 1>(q, (hipStream_t)stream);
    }
    if (!mask && direct_ok(ca, d->C, d->ldx, 0, dx, nullptr, nullptr, false, nullptr) &&
        (long)d->N * d->H * d->W >= 65536) {
        // few channels: the direct kernel, the [kh][kw][out][in] kernel read as the HWIO kernel of that conv
        DirectParams q{};
        q.x = gy; q.ldx = ldgy; q.H = d->OH; q.W = d->OW; q.C = ca;
        q.y = dx; q.ldy = d->ldx; q.OH = d->H; q.OW = d->W; q.K = d->C;
        q.R = d->R; q.S = d->S; q.stride = d->stride; q.pad_t = 0; q.pad_l = 0;
        q.w = w; q.ldw = d->ldw; q.mode = 0; q.wrows = d->K; q.act = ACIMG_ACT_NONE;
        q.M = (long)d->N * d->H * d->W;
        return launch_direct(q, ws, ws_bytes, (hipStream_t)stream);
    }
    IgemmParams p{};
    p.A = gy; p.H = d->OH; p.W = d->OW; p.C = ca; p.lda = ldgy; p.OH = d->H; p.OW = d->W;
    p.R = d->R; p.S = d->S; p.stride = d->stride; p.pad_t = 0; p.pad_l = 0;
    p.M = d->N * d->H * d->W;
    p.rowrun = (d->S > 1 && ldgy == ca) ? 1 : 0;
    p.B = w; p.ldb = d->ldw; p.Nld = d->ldw; p.Ngemm = d->C;
    p.e.Y = dx; p.e.ldy = d->ldx; p.e.M = p.M; p.e.Nstore = d->C; p.e.mask = mask; p.e.ldmask = ldmask;
    return launch_igemm(p, false, ws, ws_bytes, tickets, (hipStream_t)stream);
}

int acimg_deconv_wgrad(const AcimgConvDesc* d, const float* x, const float* gy, int ldgy,
                       float* dw, float* db, void* ws, size_t ws_bytes, void* stream) {
    int rc = check_desc(d, "deconv_wgrad");
    if (rc) return rc;
    const int ca = up4(d->K);
    if (ca > ldgy || (ldgy & 3) || ca != d->K) return fail(ACIMG_EINVAL, "deconv_wgrad: K must be a multiple of 4 and <= ldgy");
    // dW[(r,s,k)][c] = sum_{n,h,w} gy[n,h*stride+r,w*stride+s,k] * x[n,h,w,c]
    if (patch2_shape(d) && aligned16(dw) && (d->ldw & 3) == 0 && ws && aligned16(ws) &&
        ws_bytes >= (size_t)PATCH2_WGRAD_WGS * (32 * d->ldw + 8) * sizeof(float) && (!db || aligned16(db))) {
        Patch2WgradParams q{};
        q.X = x; q.ldx = d->ldx; q.G = gy; q.ldg = ldgy; q.out = static_cast<float*>(ws); q.ldo = d->ldw;
        q.db_part = db ? q.out + (size_t)PATCH2_WGRAD_WGS * 32 * d->ldw : nullptr;
        q.H = d->H; q.W = d->W; q.pixels = (long)d->N * d->H * d->W;
        hipLaunchKernelGGL(patch2_wgrad_32x8_kernel, dim3(PATCH2_WGRAD_WGS), dim3(1024), 0, (hipStream_t)stream, q);
        rc = check_launch("patch2_wgrad");
        if (rc) return rc;
        launch_slab_reduce_wide(q.out, PATCH2_WGRAD_WGS, 32L, 32, d->ldw, dw, nullptr, nullptr, (hipStream_t)stream);
        rc = check_launch("patch2_wgrad reduce");
        if (rc) return rc;
        // the transposed conv adds its bias at every output pixel, and every output pixel belongs to exactly one patch: the
        // bias gradient is the sum of all the gy values the kernel has just read
        if (db) {
            hipLaunchKernelGGL(colsum_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, q.db_part, PATCH2_WGRAD_WGS, 8, db);
            rc = check_launch("patch2_wgrad bias");
        }
        return rc;
    }
    WgradParams p{};
    p.X = gy; p.H = d->OH; p.W = d->OW; p.C = ca; p.ldx = ldgy;
    p.OH = d->H; p.OW = d->W; p.R = d->R; p.S = d->S; p.stride = d->stride; p.pad_t = 0; p.pad_l = 0;
    p.M = d->N * d->H * d->W; p.KK = d->R * d->S * ca;
    p.G = x; p.ldg = d->ldx; p.Ngemm = d->C; p.Nld = d->C; p.ldo = d->ldw;
    rc = launch_wgrad(p, dw, nullptr, ws, ws_bytes, (hipStream_t)stream);
    if (rc) return rc;
    // the transposed conv adds its bias at EVERY output pixel (gaps included): plain column sum of gy
    if (db) rc = launch_colsum(gy, (long)d->N * d->OH * d->OW, d->K, ldgy, db, ws, ws_bytes, (hipStream_t)stream);
    return rc;
}

// ------------------------------------------------------------------------------------------
// f16x3 (split fp16) forward convolution (frozen ResNet trunk)
// ------------------------------------------------------------------------------------------
struct Split3Cfg { int bm, bn; };
// Tile choice, measured per trunk conv shape at batch 32 (tools/tune_dma.py): the 8-wave 128x128 tile
// (2 workgroups/CU) wins on every shape with at least ~1 tile per CU; below that (the stride-2 3x3 conv into
// the 14x19 stage: 134 tiles) 64x128 fills more CUs; Cout = 64 uses 128x64.
static Split3Cfg pick_split3(int M, int K, bool allow32 = false) {
    if (K <= 32 && allow32) return {128, 32};     // on-the-fly kernel only (32-channel U-Net layers)
    if (K <= 64) return {128, 64};
    if (g_cfg.split3_tile_bm) {   // experiments only (acimg_configure validated the pair)
        return {g_cfg.split3_tile_bm, g_cfg.split3_tile_bn};
    }
    if ((long)cdiv(M, 128) * cdiv(K, 128) < 200) return {64, 128};
    return {128, 128};
}

extern "C++" {
// ------------------------------------------------------------------------------------------
// HALO form of the FORWARD conv / DATA GRADIENT of the same 3x3 / stride-1 / SAME layers with 32 or 64 channels on either
// side (round 4; configs[1]'s 112x149 and 56x74 stages).  As an implicit GEMM with a 32- or 64-column tile these layers
// gather x once per tap through L2 (112x149 64->32 forward: 121 us against ~50 us of bytes).  Here a workgroup stages a
// TH x 32 pixel tile of the input WITH ITS HALO once (fp32 -> 16-bit hi [, lo] planes, [pixel][channel] rows padded by 16
// bytes: conflict-light ds_read_b128 fragments at every tap shift with no swizzle) and keeps the layer's whole weight image
// in LDS (rows padded the same way); K walks (tap, 32-channel chunk); weights in the A slot, so a lane's 4 accumulators are
// 4 consecutive output channels of one pixel (16-byte stores).  MODE 0: forward - bias, raw fp32 output, batch-norm partials
// of conv + bias accumulated over the workgroup's tiles: ONE statistics row per workgroup.  MODE 1: data gradient - a
// forward conv of gy with the flipped / transposed image of acimg_conv2d_split3_prepare_dgrad; residual and ReLU mask in the
// epilogue.  One workgroup per CU, the next tile's loads held in registers while the current one is multiplied.
// ------------------------------------------------------------------------------------------
struct ConvHaloParams {
    const float* X; int H, W, ldx;
    const char* Wimg; unsigned w_lo_off;         // 16-bit image [rows][9 CIN], hi plane; lo plane w_lo_off bytes further
    // the producer's deferred batch norm on load: x' = relu(x * a_scale[c] + a_shift[c]) for pixels INSIDE the image (the
    // conv's zero padding applies after the affine); null = x as stored
    const float* a_scale; const float* a_shift; int a_relu;
    float* Y; int ldy, nout;                     // nout: channels written (0 = all NOUT; the image's further rows are zero)
    const float* bias; const float* res; int ldres; const float* mask; int ldmask;
    float* stats; int stats_ld;                  // [gridDim.x][2][stats_ld] or null
    int tiles_x, tiles_y; long tiles;
};

template <typename TR, int TERMS, int CIN, int NOUT, int MODE>
__global__ __launch_bounds__(512, 1) void conv_halo16_kernel(const ConvHaloParams p) {
    typedef typename TR::V8 V8;
    constexpr int TH = TERMS == 1 ? 8 : 4, TW = 32, XH = TH + 2, XWV = TW + 2, XW = 36;
    constexpr int PITCH = CIN * 2 + 16;               // bytes per pixel and plane (padded)
    constexpr int XPL = XH * XW * PITCH;
    constexpr int KTOT = 9 * CIN;
    constexpr int WROW = KTOT * 2 + 16;               // bytes per weight row and plane (padded)
    constexpr int WPL = NOUT * WROW;
    constexpr int NPL = TERMS == 3 ? 2 : 1;
    constexpr int MT = TERMS == 1 ? 2 : 1;            // 16-pixel tiles per wave: a whole tile row, or half of one
    constexpr int NT = NOUT / 16;
    constexpr int NXL = (XH * XWV * (CIN / 4) + 511) / 512;
    extern __shared__ __attribute__((aligned(16))) float ch16_smem[];
    char* const xl = reinterpret_cast<char*>(ch16_smem);
    char* const wl = xl + NPL * XPL;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int li = lane & 15, g = lane >> 4;
    const int jrow = TERMS == 1 ? wid : wid >> 1;     // tile row of this wave
    const int mcol0 = TERMS == 1 ? 0 : (wid & 1) * 16;

    // the weight image -> LDS (once per workgroup): NOUT rows of KTOT 16-bit values, 16-byte chunks
    for (int i = tid; i < NPL * NOUT * (KTOT / 8); i += 512) {
        const int ch = i % (KTOT / 8), r2 = i / (KTOT / 8);
        const int n = r2 % NOUT, pl = r2 / NOUT;
        const uint4 v = *reinterpret_cast<const uint4*>(p.Wimg + (size_t)pl * p.w_lo_off + ((size_t)n * KTOT + ch * 8) * 2);
        *reinterpret_cast<uint4*>(wl + pl * WPL + n * WROW + ch * 16) = v;
    }

    float4 rx[NXL];
    unsigned okm = 0;                          // which of rx[] came from inside the image (the affine applies to those only)
    // (a thread's items are always the same four channels: 512 is a multiple of CIN / 4)
    const float4 asc = p.a_scale ? *reinterpret_cast<const float4*>(p.a_scale + (tid % (CIN / 4)) * 4) : make_float4(1.f, 1.f, 1.f, 1.f);
    const float4 ash = p.a_scale ? *reinterpret_cast<const float4*>(p.a_shift + (tid % (CIN / 4)) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    auto load_tile = [&](long tile) {
        okm = 0;
        const int tx = (int)(tile % p.tiles_x);
        const long t2 = tile / p.tiles_x;
        const int ty = (int)(t2 % p.tiles_y);
        const long img = t2 / p.tiles_y;
        const float* xi = p.X + img * p.H * p.W * p.ldx;
#pragma unroll
        for (int k = 0; k < NXL; ++k) {
            const int i = tid + 512 * k;
            const int c4 = i % (CIN / 4), pix = i / (CIN / 4);
            const int row = pix / XWV, col = pix - row * XWV;
            const int iy = ty * TH + row - 1, ix = tx * TW + col - 1;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < XH && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) {
                v = *reinterpret_cast<const float4*>(xi + ((long)iy * p.W + ix) * p.ldx + c4 * 4);
                okm |= 1u << k;
            }
            rx[k] = v;
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int k = 0; k < NXL; ++k) {
            const int i = tid + 512 * k;
            const int c4 = i % (CIN / 4), pix = i / (CIN / 4);
            const int row = pix / XWV, col = pix - row * XWV;
            if (row < XH) {
                const int off = (row * XW + col) * PITCH + c4 * 8;
                uint2 hi, lo;
                split4<TR>(p.a_scale && ((okm >> k) & 1u) ? affine_relu4(rx[k], asc, ash, p.a_relu != 0) : rx[k], hi, lo);
                *reinterpret_cast<uint2*>(xl + off) = hi;
                if (TERMS == 3) *reinterpret_cast<uint2*>(xl + XPL + off) = lo;
            }
        }
    };

    f32x4 s1[NT], s2[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) s1[n] = s2[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 bv[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n)
        bv[n] = (MODE == 0 && p.bias) ? *reinterpret_cast<const f32x4*>(p.bias + n * 16 + 4 * g) : f32x4{0.f, 0.f, 0.f, 0.f};

    const int a_base = (jrow * XW + mcol0 + li) * PITCH + g * 16;       // + ((r * XW + s + 16 m) * PITCH + chunk * 64)
    const int b_base = li * WROW + g * 16;                              // + (nt * 16 * WROW + (tap * CIN + chunk * 32) * 2)

    long tile = blockIdx.x;
    if (tile < p.tiles) load_tile(tile);
    for (; tile < p.tiles; tile += gridDim.x) {
        __syncthreads();                               // everyone has finished reading the previous tile (and the weights landed)
        store_tile();
        __syncthreads();
        const int tx = (int)(tile % p.tiles_x);
        const long t2 = tile / p.tiles_x;
        const int ty = (int)(t2 % p.tiles_y);
        const long img = t2 / p.tiles_y;
        if (tile + gridDim.x < p.tiles) load_tile(tile + gridDim.x);     // in flight while this tile is multiplied
        f32x4 acc[MT][NT];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int s_ = 0; s_ < 3; ++s_)
#pragma unroll
                for (int ck = 0; ck < CIN / 32; ++ck) {
                    V8 ah[MT], al[MT], bh[NT], bl[NT];
#pragma unroll
                    for (int m = 0; m < MT; ++m) {
                        const int off = a_base + (r * XW + s_ + 16 * m) * PITCH + ck * 64;
                        ah[m] = *reinterpret_cast<const V8*>(xl + off);
                        if (TERMS == 3) al[m] = *reinterpret_cast<const V8*>(xl + XPL + off);
                    }
#pragma unroll
                    for (int n = 0; n < NT; ++n) {
                        const int off = b_base + n * 16 * WROW + ((r * 3 + s_) * CIN + ck * 32) * 2;
                        bh[n] = *reinterpret_cast<const V8*>(wl + off);
                        if (TERMS == 3) bl[n] = *reinterpret_cast<const V8*>(wl + WPL + off);
                    }
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int n = 0; n < NT; ++n) {
                            if (TERMS == 3) {
                                acc[m][n] = TR::mfma(bl[n], ah[m], acc[m][n]);
                                acc[m][n] = TR::mfma(bh[n], al[m], acc[m][n]);
                            }
                            acc[m][n] = TR::mfma(bh[n], ah[m], acc[m][n]);
                        }
                }
        // lane (li, g) of acc[m][n] holds output pixel (row jrow, column mcol0 + 16 m + li), channels 16 n + 4 g .. + 3
        const int oy = ty * TH + jrow;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int ox = tx * TW + mcol0 + 16 * m + li;
            if (oy < p.H && ox < p.W) {
                const long pix = (img * p.H + oy) * p.W + ox;
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    if (NOUT == 16 && n * 16 + 4 * g >= p.nout) continue;      // (the 16-row instance serves 4 - 16 channels)
                    f32x4 v = acc[m][n] * TR::OUTSCALE + bv[n];
                    if (MODE == 1) {
                        if (p.res) v += *reinterpret_cast<const f32x4*>(p.res + pix * p.ldres + n * 16 + 4 * g);
                        if (p.mask) {
                            const f32x4 k = *reinterpret_cast<const f32x4*>(p.mask + pix * p.ldmask + n * 16 + 4 * g);
#pragma unroll
                            for (int c = 0; c < 4; ++c) v[c] = k[c] > 0.f ? v[c] : 0.f;
                        }
                    } else {
                        s1[n] += v;
                        s2[n] += v * v;
                    }
                    *reinterpret_cast<f32x4*>(p.Y + pix * p.ldy + n * 16 + 4 * g) = v;
                }
            }
        }
    }
    if (MODE == 0 && p.stats) {
        // the workgroup's statistics row: 16 pixel lanes by DPP, 8 waves through LDS, in wave order
        __syncthreads();
        float* red = reinterpret_cast<float*>(xl);     // [8][2][NOUT]
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float a = row16_sum(s1[n][c]), b = row16_sum(s2[n][c]);
                if (li == 0) {
                    red[(wid * 2 + 0) * NOUT + n * 16 + 4 * g + c] = a;
                    red[(wid * 2 + 1) * NOUT + n * 16 + 4 * g + c] = b;
                }
            }
        __syncthreads();
        if (tid < 2 * NOUT) {
            const int which = tid / NOUT, n = tid % NOUT;
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) t += red[(w * 2 + which) * NOUT + n];
            p.stats[((long)blockIdx.x * 2 + which) * p.stats_ld + n] = t;
        }
    }
}

// shapes the halo forward / data-gradient kernel takes: 3x3, stride 1, one pixel of padding, (reduction, output) channels
// (32 | 64, 32) forward and (32, 32 | 64) backward, from 65536 pixels on
static bool conv_halo16_fwd_shape(const AcimgConvDesc* d) {
    return d->R == 3 && d->S == 3 && d->stride == 1 && d->pad_t == 1 && d->pad_l == 1 && d->OH == d->H && d->OW == d->W &&
           (d->C == 32 || d->C == 64) && d->K == 32 && (long)d->N * d->H * d->W >= 65536 && d->act == ACIMG_ACT_NONE &&
           g_cfg.wgrad_halo;
}
static bool conv_halo16_dgrad_shape(const AcimgConvDesc* d) {
    return d->R == 3 && d->S == 3 && d->stride == 1 && d->pad_t == 1 && d->pad_l == 1 && d->OH == d->H && d->OW == d->W &&
           (d->C == 32 || d->C == 64) && d->K == 32 && (long)d->N * d->H * d->W >= 65536 && g_cfg.wgrad_halo;
}
static constexpr int CONV_HALO16_WGS = 256;          // one workgroup per CU = statistics rows of the forward form

template <typename TR, int TERMS, int CIN, int NOUT, int MODE>
static int launch_conv_halo16(ConvHaloParams q, int N, hipStream_t st) {
    constexpr int TH = TERMS == 1 ? 8 : 4, NPL = TERMS == 3 ? 2 : 1;
    constexpr int lds = NPL * ((TH + 2) * 36 * (CIN * 2 + 16) + NOUT * (9 * CIN * 2 + 16));
    static_assert(lds <= 160 * 1024 && 8 * 2 * NOUT * 4 <= lds, "LDS budget");
    q.tiles_x = cdiv(q.W, 32); q.tiles_y = cdiv(q.H, TH);
    q.tiles = (long)N * q.tiles_x * q.tiles_y;
    if (!q.nout) q.nout = NOUT;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_halo16_kernel<TR, TERMS, CIN, NOUT, MODE>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_set = true;
    }
    hipLaunchKernelGGL((conv_halo16_kernel<TR, TERMS, CIN, NOUT, MODE>), dim3(CONV_HALO16_WGS), dim3(512), lds, st, q);
    return check_launch("conv_halo16");
}
}  // extern "C++"

static int dgrad_halo16_narrow(const AcimgConvDesc* d, const float* gy, int ldgy, const float* w, float* dx, int lddx,
                               const float* residual, int ldres, const float* mask, int ldmask, void* ws, hipStream_t st) {
    // the flipped / transposed image [16 rows = input channels, zero beyond C][k = tap * 32 + gy channel], bf16 hi / lo
    FewParams f{};
    f.w = w; f.ldw = d->ldw; f.wrows = d->C; f.cin = d->K; f.mode = 1; f.nout = d->C;
    hipLaunchKernelGGL((few16_prepare_kernel<SplitBF16>), dim3(cdiv(16 * 288, 256)), dim3(256), 0, st, f, 32, 16,
                       static_cast<__bf16*>(ws));
    ConvHaloParams q{};
    q.X = gy; q.H = d->H; q.W = d->W; q.ldx = ldgy; q.Wimg = static_cast<const char*>(ws); q.w_lo_off = 16 * 288 * 2;
    q.Y = dx; q.ldy = lddx; q.nout = d->C; q.res = residual; q.ldres = ldres; q.mask = mask; q.ldmask = ldmask;
    return launch_conv_halo16<SplitBF16, 3, 32, 16, 1>(q, d->N, st);
}

int acimg_conv2d_fwd_split3_stats_rows(const AcimgConvDesc* d) {
    const int M = d->N * d->OH * d->OW;
    if (conv_halo16_fwd_shape(d)) return CONV_HALO16_WGS;      // the halo form leaves one row per workgroup
    return cdiv(M, pick_split3(M, d->K).bm);
}

// Persistent kernel or one tile per workgroup for the pre-split (trunk) forward conv (tools/trunk_shapes.py,
// profiles/r02/trunk_shapes_*.txt): walking a tile list wins 2-9 % wherever there is at least one full round of whole
// 128x128 tiles (most on short-K multi-round layers); when every tile belongs to the split tail (fewer tiles than
// resident workgroups) the one-tile kernel's leaner code is 3-7 % faster.
static bool split3p_persistent(const Split3Cfg& c, long tiles) {
    if (c.bm != 128 || c.bn != 128) return false;
    return g_cfg.trunk_persistent == 2 || (g_cfg.trunk_persistent == 1 && tiles >= 512);
}

// Halo kernel (igemm_split3h_kernel.hpp) for a pre-split trunk conv: 3x3 / stride 1 / SAME on 128x128 tiles, image rows
// short enough for an 18-brick patch.  trunk_halo = 1 takes it where it was measured to pay, 2 wherever it applies.
static constexpr int HALO_NB = 18;
static bool halo_applies(const AcimgConvDesc* d, int terms) {
    if (terms != 3 || d->R != 3 || d->S != 3 || d->stride != 1 || d->pad_t != 1 || d->pad_l != 1 || d->OH != d->H ||
        d->OW != d->W || d->C % 32)
        return false;
    const int M = d->N * d->OH * d->OW;
    const Split3Cfg c = pick_split3(M, d->K);
    if (c.bm != 128 || c.bn != 128) return false;
    return ((d->W + 16) >> 4) + 8 + (d->W >> 4) + 1 <= HALO_NB;
}
static bool halo_on(const AcimgConvDesc* d, int terms) {
    // trunk_halo = 1 ("where it was measured to pay") selects nothing yet: at batch 32 the halo form is at parity with the
    // per-tap kernels on the 56x75 / 28x38 layers and behind the ring kernel on 14x19; 2 = wherever it applies
    return g_cfg.trunk_halo == 2 && halo_applies(d, terms);
}

// Ring kernel (igemm_split3r_kernel.hpp) for a pre-split trunk conv, and with how many tile rows: 0 = not on it.
// Measured per shape at batch 32 and 30 (tools/trunk_shapes.py, profiles/r03/trunk_shapes_r03*.txt): in its steady state
// the ring kernel's K loop is 8-14 % faster than the two-workgroups-per-CU kernels' (long-K layers whose tiles fill one
// round of the 256 CUs), but with ONE workgroup per CU it pays more for tile quantisation (a 1.04-round layer leaves
// 246 CUs idle for a round, where 512 slots leave a quarter of the chip) and for short-K tiles (the output tile's stores
// and the deferred stage at every unit boundary).  trunk_ring = 1 therefore takes it where it won: layers of at most
// ~1.2 rounds of 128x128 tiles over the CUs (the 14x19 stage) with at least 64 K steps, on 128-row tiles with the tail
// cut into K ranges; trunk_ring = 2 forces it (experiments, tests).
static int ring_rows(const AcimgConvDesc* d, int terms) {
    if (!g_cfg.trunk_ring || terms != 3 || halo_on(d, terms)) return 0;
    const int M = d->N * d->OH * d->OW;
    if ((long)M * d->ldy * 4 >= (1L << 31)) return 0;      // 32-bit output descriptor (see fwd_presplit)
    const Split3Cfg c = pick_split3(M, d->K);
    if (c.bm != 128 || c.bn != 128) return 0;
    const long t128 = (long)cdiv(M, 128) * cdiv(d->K, 128);
    const int kiters = d->R * d->S * (d->C / 32);
    if (g_cfg.trunk_ring == 1) return (t128 > 256 && t128 <= 300 && kiters >= 64) ? 128 : 0;
    if (g_cfg.trunk_ring_bm) return g_cfg.trunk_ring_bm;
    const long t256 = (long)cdiv(M, 256) * cdiv(d->K, 128);
    return t256 >= 500 ? 256 : 128;
}

int acimg_conv2d_fwd_split3p_stats_rows(const AcimgConvDesc* d) {
    const int M = d->N * d->OH * d->OW;
    const int rr = ring_rows(d, 3);
    return cdiv(M, rr ? rr : pick_split3(M, d->K).bm);
}

int acimg_conv2d_fwd_split3_tiling(const AcimgConvDesc* d, int* out) {
    if (!d || !out) return fail(ACIMG_EINVAL, "conv2d_fwd_split3_tiling: null argument");
    const int M = d->N * d->OH * d->OW;
    Split3Cfg c = pick_split3(M, d->K);
    out[0] = c.bm;
    out[1] = c.bn;
    out[2] = split3p_persistent(c, (long)cdiv(M, c.bm) * cdiv(d->K, c.bn)) ? 1 : 0;
    if (const int rr = ring_rows(d, 3)) {      // the ring kernel: its row tile, flag 2
        out[0] = rr;
        out[2] = 2;
    }
    if (halo_on(d, 3)) out[2] = 3;             // the halo kernel (128x128)
    return ACIMG_OK;
}

// Row-major planes [hi | lo][ldw][R*S*C], then (C % 32 == 0) the same weights once more in LDS-TILE ORDER: for every
// (128-column tile nt, K step q of 32) the two 8 KiB plane images exactly as the trunk kernels keep them in LDS
// (64-byte rows, logical chunk kc of row r at chunk kc ^ swz(r)), so a wave's LDS-DMA request for a weight piece is one
// contiguous KiB = eight whole 128-byte lines instead of sixteen 64-byte pieces of rows 2*R*S*C bytes apart.
static size_t split3_rowmajor_bytes(const AcimgConvDesc* d) { return (size_t)2 * d->ldw * d->R * d->S * d->C * 2; }
static size_t split3_brick_bytes(const AcimgConvDesc* d) {
    return d->C % 32 ? 0 : (size_t)cdiv(d->ldw, 128) * (d->R * d->S * d->C / 32) * 2 * 8192;
}
size_t acimg_conv2d_split3_weight_bytes(const AcimgConvDesc* d) {
    return split3_rowmajor_bytes(d) + split3_brick_bytes(d);
}

// one thread per 16-byte chunk of the tile-ordered image: brick (nt, q, plane), row, physical chunk
__global__ void split3_brick_kernel(const char* __restrict__ planes, char* __restrict__ bricks, const int Nrows,
                                    const int Ktot, const long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int pch = (int)(i & 3), row = (int)((i >> 2) & 127), plane = (int)((i >> 9) & 1);
    const long bq = i >> 10;
    const int kit = Ktot / 32;
    const int nt = (int)(bq / kit), q = (int)(bq - (long)nt * kit);
    const int n = nt * 128 + row;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (n < Nrows)
        v = *reinterpret_cast<const uint4*>(planes + (((long)plane * Nrows + n) * Ktot + q * 32 + ((pch ^ swz(row)) << 3)) * 2);
    *reinterpret_cast<uint4*>(bricks + i * 16) = v;
}

int acimg_conv2d_split3_prepare(const AcimgConvDesc* d, const float* w, void* wsplit, void* stream) {
    int rc = check_desc(d, "conv2d_split3_prepare", true);
    if (rc) return rc;
    const int Ktot = d->R * d->S * d->C;
    hipLaunchKernelGGL((split3_prepare_kernel<SplitF16, false>), dim3(cdiv(Ktot, 32), cdiv(d->ldw, 32)), dim3(256), 0,
                       (hipStream_t)stream, w, d->R * d->S, d->C, d->K, d->ldw, d->ldw, static_cast<_Float16*>(wsplit));
    rc = check_launch("split3_prepare");
    if (rc || !split3_brick_bytes(d)) return rc;
    const long total = (long)(split3_brick_bytes(d) / 16);
    hipLaunchKernelGGL(split3_brick_kernel, dim3((unsigned)cdiv(total, 256L)), dim3(256), 0, (hipStream_t)stream,
                       static_cast<const char*>(wsplit), static_cast<char*>(wsplit) + split3_rowmajor_bytes(d), d->ldw, Ktot,
                       total);
    return check_launch("split3_prepare (tile order)");
}

int acimg_conv2d_bf16_prepare(const AcimgConvDesc* d, const float* w, void* wsplit, void* stream) {
    int rc = check_desc(d, "conv2d_bf16_prepare", true);
    if (rc) return rc;
    const int Ktot = d->R * d->S * d->C;
    hipLaunchKernelGGL((split3_prepare_kernel<SplitBF16, false>), dim3(cdiv(Ktot, 32), cdiv(d->ldw, 32)), dim3(256), 0,
                       (hipStream_t)stream, w, d->R * d->S, d->C, d->K, d->ldw, d->ldw, static_cast<__bf16*>(wsplit));
    return check_launch("conv2d_bf16_prepare");
}

size_t acimg_conv2d_split3_dgrad_weight_bytes(const AcimgConvDesc* d) {
    return (size_t)2 * d->C * d->R * d->S * up4(d->K) * 2;
}

int acimg_conv2d_split3_prepare_dgrad(const AcimgConvDesc* d, const float* w, void* wsplit, void* stream) {
    int rc = check_desc(d, "conv2d_split3_prepare_dgrad");
    if (rc) return rc;
    if (d->K % 32 || d->K > d->ldw) return fail(ACIMG_EINVAL, "conv2d_split3_prepare_dgrad: K must be a multiple of 32");
    const int Ktot = d->R * d->S * d->K;
    hipLaunchKernelGGL((split3_prepare_kernel<SplitBF16, true>), dim3(cdiv(Ktot, 32), cdiv(d->C, 32)), dim3(256), 0,
                       (hipStream_t)stream, w, d->R * d->S, d->C, d->K, d->ldw, d->C, static_cast<__bf16*>(wsplit));
    return check_launch("split3_prepare_dgrad");
}

/* all of a model's trainable kernels in one launch: mode[i] = 0 forward image (acimg_conv2d_split3_prepare), 1 data-
 * gradient image (acimg_conv2d_split3_prepare_dgrad), 2 forward bf16 image (acimg_conv2d_bf16_prepare) */
int acimg_conv2d_split3_prepare_multi(int n, const AcimgConvDesc* const* descs, const float* const* w, void* const* out,
                                      const int* mode, void* stream) {
    if (n == 0) return ACIMG_OK;
    if (n < 0 || n > 16 || !descs || !w || !out || !mode)
        return fail(ACIMG_EINVAL, "conv2d_split3_prepare_multi: need 0..16 jobs and non-null tables");
    PrepJobs jobs{};
    jobs.n = n;
    int blocks = 0;
    for (int i = 0; i < n; ++i) {
        const AcimgConvDesc* d = descs[i];
        int rc = check_desc(d, "conv2d_split3_prepare_multi", true);
        if (rc) return rc;
        if (!w[i] || !out[i]) return fail(ACIMG_EINVAL, "conv2d_split3_prepare_multi: null pointer in job %d", i);
        PrepJob& J = jobs.j[i];
        J.w = w[i]; J.out = out[i]; J.ntaps = d->R * d->S; J.C = d->C; J.K = d->K; J.ldw = d->ldw;
        if (mode[i] < 0 || mode[i] > 2) return fail(ACIMG_EINVAL, "conv2d_split3_prepare_multi: job %d: mode is 0, 1 or 2", i);
        J.dgrad = mode[i];
        int tiles_n;
        if (J.dgrad == 1) {
            if (d->K % 32 || d->K > d->ldw)
                return fail(ACIMG_EINVAL, "conv2d_split3_prepare_multi: job %d: K must be a multiple of 32", i);
            J.Nrows = d->C;
            J.tiles_k = cdiv(d->R * d->S * d->K, 32);
            tiles_n = cdiv(d->C, 32);
        } else {
            J.Nrows = d->ldw;
            J.tiles_k = cdiv(d->R * d->S * d->C, 32);
            tiles_n = cdiv(d->ldw, 32);
        }
        J.block0 = blocks;
        blocks += J.tiles_k * tiles_n;
    }
    hipLaunchKernelGGL(split3_prepare_multi_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, jobs);
    return check_launch("split3_prepare_multi");
}

extern "C++" {
template <typename TR, int TERMS = 3>
static int launch_split3(IgemmParams& p, hipStream_t st) {
    Split3Cfg c = pick_split3(p.M, p.Ngemm, true);
    dim3 grid(cdiv(p.M, c.bm), cdiv(p.Ngemm, c.bn), 1);
    if (c.bm == 128 && c.bn == 32)
        hipLaunchKernelGGL((igemm_split3_kernel<128, 32, 4, 1, 256, TR, TERMS>), grid, dim3(256), 2 * (2 * 128 * 64 + 2 * 32 * 64), st, p);
    else if (c.bm == 128 && c.bn == 128)
        hipLaunchKernelGGL((igemm_split3_kernel<128, 128, 2, 4, 512, TR, TERMS>), grid, dim3(512), 65536, st, p);
    else if (c.bm == 64 && c.bn == 128)
        hipLaunchKernelGGL((igemm_split3_kernel<64, 128, 1, 4, 256, TR, TERMS>), grid, dim3(256), 2 * (2 * 64 * 64 + 2 * 128 * 64), st, p);
    else if (c.bm == 128 && c.bn == 64)
        hipLaunchKernelGGL((igemm_split3_kernel<128, 64, 2, 2, 256, TR, TERMS>), grid, dim3(256), 2 * (2 * 128 * 64 + 2 * 64 * 64), st, p);
    else
        return fail(ACIMG_EINVAL, "split3: unsupported tile %dx%d", c.bm, c.bn);
    return check_launch("igemm_split3");
}

static void epi_vec_flag(EpiParams& e) {
    e.vec = aligned16(e.Y) && (e.ldy & 3) == 0 && (!e.bias || aligned16(e.bias)) &&
            (!e.res || (aligned16(e.res) && (e.ldres & 3) == 0)) &&
            (!e.mask || (aligned16(e.mask) && (e.ldmask & 3) == 0)) && (!e.scatter || (e.Ko & 3) == 0);
}
}  // extern "C++"

static int fwd_split_onthefly(const AcimgConvDesc* d, const float* x, const void* wsplit, const float* bias, float* y,
                              const float* in_scale, const float* in_shift, int in_relu, float* stats, void* stream,
                              bool bf16);

int acimg_conv2d_fwd_split3(const AcimgConvDesc* d, const float* x, const void* wsplit, const float* bias,
                            float* y, const float* in_scale, const float* in_shift, int in_relu, float* stats,
                            void* stream) {
    return fwd_split_onthefly(d, x, wsplit, bias, y, in_scale, in_shift, in_relu, stats, stream, false);
}

/* bf16 operands (activations and weights rounded to bf16, one MFMA per product, fp32 accumulation); weights from
 * acimg_conv2d_bf16_prepare */
int acimg_conv2d_fwd_bf16(const AcimgConvDesc* d, const float* x, const void* wsplit, const float* bias,
                          float* y, const float* in_scale, const float* in_shift, int in_relu, float* stats,
                          void* stream) {
    return fwd_split_onthefly(d, x, wsplit, bias, y, in_scale, in_shift, in_relu, stats, stream, true);
}

static int fwd_split_onthefly(const AcimgConvDesc* d, const float* x, const void* wsplit, const float* bias, float* y,
                              const float* in_scale, const float* in_shift, int in_relu, float* stats, void* stream,
                              bool bf16) {
    int rc = check_desc(d, "conv2d_fwd_split3", true);
    if (rc) return rc;
    if (d->C % 32) return fail(ACIMG_EINVAL, "conv2d_fwd_split3: C=%d must be a multiple of 32", d->C);
    if (d->ldw < d->K || !aligned16(x) || !aligned16(wsplit))
        return fail(ACIMG_EINVAL, "conv2d_fwd_split3: ldw<K or unaligned operands");
    if (conv_halo16_fwd_shape(d)) {
        // (acimg_conv2d_fwd_split3_stats_rows already told the caller this shape leaves CONV_HALO16_WGS rows: no fallback)
        if ((in_scale != nullptr) != (in_shift != nullptr) || (in_relu && !in_scale) || !aligned16(y) || (d->ldy & 3) || (d->ldx & 3) ||
            (bias && !aligned16(bias)) || (in_scale && (!aligned16(in_scale) || !aligned16(in_shift))))
            return fail(ACIMG_EINVAL, "conv2d_fwd_split3: the halo form of this shape needs scale AND shift of an input affine, "
                                      "16-byte aligned y / bias / affine and ldx, ldy multiples of 4");
        ConvHaloParams q{};
        q.a_scale = in_scale; q.a_shift = in_shift; q.a_relu = in_relu;
        q.X = x; q.H = d->H; q.W = d->W; q.ldx = d->ldx; q.Wimg = static_cast<const char*>(wsplit);
        q.w_lo_off = (unsigned)((size_t)d->ldw * d->R * d->S * d->C * 2);
        q.Y = y; q.ldy = d->ldy; q.bias = bias; q.stats = stats; q.stats_ld = d->ldw;
        hipStream_t st = (hipStream_t)stream;
        if (bf16) return d->C == 64 ? launch_conv_halo16<SplitBF16, 1, 64, 32, 0>(q, d->N, st)
                                    : launch_conv_halo16<SplitBF16, 1, 32, 32, 0>(q, d->N, st);
        return d->C == 64 ? launch_conv_halo16<SplitF16, 3, 64, 32, 0>(q, d->N, st)
                          : launch_conv_halo16<SplitF16, 3, 32, 32, 0>(q, d->N, st);
    }
    IgemmParams p{};
    p.A = x; p.H = d->H; p.W = d->W; p.C = d->C; p.lda = d->ldx;
    p.OH = d->OH; p.OW = d->OW; p.R = d->R; p.S = d->S; p.stride = d->stride;
    p.pad_t = d->pad_t; p.pad_l = d->pad_l;
    p.M = d->N * d->OH * d->OW;
    p.a_scale = in_scale; p.a_shift = in_shift; p.a_relu = in_relu;
    p.B = static_cast<const float*>(wsplit); p.Nld = d->ldw; p.Ngemm = d->K;
    p.ntaps = d->R * d->S;
    p.kiters = p.ntaps * (d->C / 32);
    p.splits = 1;
    const long a_bytes = (((long)d->N * d->H * d->W - 1) * d->ldx + d->C) * 4;
    const long b_bytes = (long)acimg_conv2d_split3_weight_bytes(d);
    if (a_bytes >= (1L << 31) || b_bytes >= (1L << 31)) return fail(ACIMG_EINVAL, "conv2d_fwd_split3: operand >= 2 GiB");
    p.a_bytes = (unsigned)a_bytes; p.b_bytes = (unsigned)b_bytes;
    EpiParams& e = p.e;
    e.Y = y; e.ldy = d->ldy; e.M = p.M; e.Nstore = d->K; e.act = d->act; e.bias = bias;
    e.stats = stats; e.stats_ld = d->ldw;
    epi_vec_flag(e);
    if (bf16) return launch_split3<SplitBF16, 1>(p, (hipStream_t)stream);
    return launch_split3<SplitF16>(p, (hipStream_t)stream);
}

/* data gradient on the bf16x3 path (stride-1 convs): a forward conv of gy with the flipped/transposed kernel */
static int dgrad_split_onthefly(const AcimgConvDesc* d, const float* gy, int ldgy, const void* wsplit_t, float* dx,
                                int lddx, const float* residual, int ldres, const float* mask, int ldmask,
                                void* stream, int terms);

int acimg_conv2d_dgrad_split3(const AcimgConvDesc* d, const float* gy, int ldgy, const void* wsplit_t, float* dx,
                              int lddx, const float* residual, int ldres, const float* mask, int ldmask,
                              void* stream) {
    return dgrad_split_onthefly(d, gy, ldgy, wsplit_t, dx, lddx, residual, ldres, mask, ldmask, stream, 3);
}

/* the same with gy and the kernel rounded to bf16, one MFMA per product (weights: acimg_conv2d_split3_prepare_dgrad) */
int acimg_conv2d_dgrad_bf16(const AcimgConvDesc* d, const float* gy, int ldgy, const void* wsplit_t, float* dx,
                            int lddx, const float* residual, int ldres, const float* mask, int ldmask,
                            void* stream) {
    return dgrad_split_onthefly(d, gy, ldgy, wsplit_t, dx, lddx, residual, ldres, mask, ldmask, stream, 1);
}

static int dgrad_split_onthefly(const AcimgConvDesc* d, const float* gy, int ldgy, const void* wsplit_t, float* dx,
                                int lddx, const float* residual, int ldres, const float* mask, int ldmask,
                                void* stream, int terms) {
    int rc = check_desc(d, "conv2d_dgrad_split3");
    if (rc) return rc;
    if (d->stride != 1 || d->K % 32 || d->K > ldgy || (ldgy & 3))
        return fail(ACIMG_EINVAL, "conv2d_dgrad_split3: needs stride 1 and K %% 32 == 0 (K=%d)", d->K);
    if (lddx <= 0) lddx = d->ldx;
    if (conv_halo16_dgrad_shape(d) && aligned16(gy) && aligned16(wsplit_t) && aligned16(dx) && (lddx & 3) == 0 &&
        (!residual || (aligned16(residual) && (ldres & 3) == 0)) && (!mask || (aligned16(mask) && (ldmask & 3) == 0))) {
        // a forward SAME conv of gy (32 channels) with the flipped / transposed image: rows = the layer's input channels
        ConvHaloParams q{};
        q.X = gy; q.H = d->H; q.W = d->W; q.ldx = ldgy; q.Wimg = static_cast<const char*>(wsplit_t);
        q.w_lo_off = (unsigned)((size_t)d->C * d->R * d->S * d->K * 2);
        q.Y = dx; q.ldy = lddx; q.res = residual; q.ldres = ldres; q.mask = mask; q.ldmask = ldmask;
        hipStream_t st = (hipStream_t)stream;
        if (terms == 1) return d->C == 64 ? launch_conv_halo16<SplitBF16, 1, 32, 64, 1>(q, d->N, st)
                                          : launch_conv_halo16<SplitBF16, 1, 32, 32, 1>(q, d->N, st);
        return d->C == 64 ? launch_conv_halo16<SplitBF16, 3, 32, 64, 1>(q, d->N, st)
                          : launch_conv_halo16<SplitBF16, 3, 32, 32, 1>(q, d->N, st);
    }
    IgemmParams p{};
    p.A = gy; p.H = d->OH; p.W = d->OW; p.C = d->K; p.lda = ldgy;
    p.OH = d->H; p.OW = d->W; p.R = d->R; p.S = d->S; p.stride = 1;
    p.pad_t = d->R - 1 - d->pad_t; p.pad_l = d->S - 1 - d->pad_l;
    p.M = d->N * d->H * d->W;
    p.B = static_cast<const float*>(wsplit_t); p.Nld = d->C; p.Ngemm = d->C;
    p.ntaps = d->R * d->S;
    p.kiters = p.ntaps * (d->K / 32);
    p.splits = 1;
    const long a_bytes = (((long)d->N * d->OH * d->OW - 1) * ldgy + d->K) * 4;
    const long b_bytes = (long)acimg_conv2d_split3_dgrad_weight_bytes(d);
    if (a_bytes >= (1L << 31) || b_bytes >= (1L << 31)) return fail(ACIMG_EINVAL, "conv2d_dgrad_split3: operand >= 2 GiB");
    p.a_bytes = (unsigned)a_bytes; p.b_bytes = (unsigned)b_bytes;
    EpiParams& e = p.e;
    e.Y = dx; e.ldy = lddx; e.M = p.M; e.Nstore = d->C; e.act = ACIMG_ACT_NONE;
    e.res = residual; e.ldres = ldres; e.mask = mask; e.ldmask = ldmask;
    epi_vec_flag(e);
    if (terms == 1) return launch_split3<SplitBF16, 1>(p, (hipStream_t)stream);
    return launch_split3<SplitBF16>(p, (hipStream_t)stream);
}

// ---- tail split of the trunk kernel: which tiles to cut, and into how many K ranges ------------------------
struct TailPlan { int whole, s, rem; };
static int resident_slots(int which, const void* fn, int threads, size_t lds) {
    static int cache[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (!cache[which]) {
        int dev = 0, ncu = 0, per = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, fn, threads, lds) != hipSuccess || ncu <= 0 || per <= 0) {
            (void)hipGetLastError();
            ncu = 256;                    // MI355X; no device (CPU-side sizing queries): same answer
            per = which >= 7 ? 2 : which >= 4 ? 1 : (which == 0 || which == 3) ? 2 : 3;
        }
        cache[which] = ncu * per;
    }
    return cache[which];
}
static TailPlan pick_tail(int T, int P, int KI, int max_units) {
    const int rem = T % P;
    TailPlan best{T, 1, 0};
    if (rem == 0 || !g_cfg.tail_split) return best;
    if (g_cfg.tail_s) {   // experiments only: force the number of K ranges
        const int s = g_cfg.tail_s;
        if (s > 1 && KI / s >= 1 && (long)rem * s <= max_units) return TailPlan{T - rem, s, rem};
        return best;
    }
    // cost in K steps of the last round: whole tiles = KI; s ranges = rounds(rem*s) * ceil(KI/s) + hand-off
    double best_cost = KI;
    static const int cand[] = {2, 3, 4, 6, 8, 12, 16};
    for (int s : cand) {
        if (KI / s < 2 || (long)rem * s > max_units) break;
        // hand-off calibrated on the trunk shapes (tools/tune_dma.py): partial store + ticket + the last arriver's
        // s x 64 KiB of sc1 loads cost about 8 + s K steps, so 1x1 layers with few K steps are left whole
        const double cost = (double)cdiv((long)rem * s, P) * cdiv(KI, s) + 8.0 + 1.0 * s;
        if (cost < 0.9 * best_cost) {
            best_cost = cost;
            best = TailPlan{T - rem, s, rem};
        }
    }
    return best;
}
static constexpr int TS_MAX_UNITS = 1024;      // partial slots (64 KiB each for a 128x128 tile; a 256x128 tile takes two)
static constexpr size_t TS_COUNTER_BYTES = 4096;

#if defined(ACIMG_STAMP) || defined(ACIMG_ABLATE)
static float* g_stamp_buf = nullptr;      // diagnostic build only (tools/build_stamp.sh): never in libacimg.so
static int g_stamp_nostore = 0;           // ablation: the persistent kernel's output stores go out of range (dropped)
extern "C" int acimg_debug_stamp_buffer(void* buf) {
    g_stamp_buf = static_cast<float*>(buf);
    return 0;
}
extern "C" int acimg_debug_no_output_stores(int on) {
    g_stamp_nostore = on;
    return 0;
}
#endif

size_t acimg_conv2d_fwd_split3p_workspace(const AcimgConvDesc* d) {
    (void)d;
    return TS_COUNTER_BYTES + (size_t)TS_MAX_UNITS * 128 * 128 * sizeof(float);
}

// the two passes of a conv whose raw output never reaches memory (igemm_split3dp_kernel, EPI 1 / 2)
struct Split3pTail {
    int mode;                      // 1: statistics only; 2: relu(acc * scale + shift + shortcut) -> split planes;
                                   // 3: the same with a projection shortcut (raw fp32 + its own scale / shift)
    const float* scale;
    const float* shift;
    const void* sc_planes;         // mode 3: the fp32 [M][K] output of the shortcut conv
    size_t sc_lo_off;
    void* out_planes;
    size_t out_lo_off;
    const float* scale2;
    const float* shift2;
};

static int fwd_presplit(const AcimgConvDesc* d, const void* x_planes, size_t x_lo_off, const void* wsplit, float* y,
                        float* stats, void* ws, size_t ws_bytes, void* stream, const int terms,
                        const Split3pTail* tail = nullptr) {
    int rc = check_desc(d, "conv2d_fwd_split3p");
    if (rc) return rc;
    if (d->C % 32) return fail(ACIMG_EINVAL, "conv2d_fwd_split3p: C=%d must be a multiple of 32", d->C);
    if (d->ldw < d->K || !aligned16(x_planes) || !aligned16(wsplit) || !aligned16(y) || (d->ldy & 3) || d->ldx != d->C ||
        (x_lo_off & 15))
        return fail(ACIMG_EINVAL, "conv2d_fwd_split3p: ldw<K, ldx != C (split-format tensors are dense) or unaligned operands");
    IgemmParams p{};
    p.A = static_cast<const float*>(x_planes); p.H = d->H; p.W = d->W; p.C = d->C; p.lda = d->ldx;
    p.OH = d->OH; p.OW = d->OW; p.R = d->R; p.S = d->S; p.stride = d->stride;
    p.pad_t = d->pad_t; p.pad_l = d->pad_l;
    p.M = d->N * d->OH * d->OW;
    p.B = static_cast<const float*>(wsplit); p.Nld = d->ldw; p.Ngemm = d->K;
    p.ntaps = d->R * d->S;
    p.kiters = p.ntaps * (d->C / 32);
    p.splits = 1;
    const long plane = (long)acimg_split_plane_bytes((long)d->N * d->H * d->W, d->C);
    const long a_bytes = (long)x_lo_off + plane;
    const long b_bytes = (long)acimg_conv2d_split3_weight_bytes(d);
    if (a_bytes >= (1L << 31) || b_bytes >= (1L << 31) || (long)x_lo_off < plane)
        return fail(ACIMG_EINVAL, "conv2d_fwd_split3p: operand >= 2 GiB or overlapping planes");
    p.a_bytes = (unsigned)a_bytes; p.b_bytes = (unsigned)b_bytes; p.a_lo_off = (unsigned)x_lo_off;
    p.b_brick = (unsigned)split3_rowmajor_bytes(d);
    EpiParams& e = p.e;
    e.Y = y; e.ldy = d->ldy; e.M = p.M; e.Nstore = d->K; e.act = d->act;
    e.stats = stats; e.stats_ld = d->ldw; e.vec = 1;
    Split3Cfg c = pick_split3(p.M, d->K);
    hipStream_t st = (hipStream_t)stream;
    // XCD-aware rasterisation: the ~64 tiles resident on one XCD form a (64/gn) x gn rectangle of the tile grid
    p.ras_tiles_m = cdiv(p.M, c.bm);
    p.ras_tiles_n = cdiv(d->K, c.bn);
    p.ras_gn = std::min(p.ras_tiles_n, 8);
    p.ras_gm = std::max(1, 64 / p.ras_gn);
    // 2 stages x (hi, lo) x 64-byte rows; the epilogue restages the fp32 output tile there (+ 2 x WGM x BN floats
    // of statistics scratch behind it)
    const size_t lds_bytes = std::max((size_t)2 * 2 * (c.bm + c.bn) * 64, (size_t)c.bm * c.bn * 4) + 4 * 2 * c.bn * 4;
    const int T = p.ras_tiles_m * p.ras_tiles_n;
    const int which = (c.bm == 128 && c.bn == 128) ? 0 : (c.bm == 64 ? 1 : 2);
    const void* fn = which == 0 ? (const void*)igemm_split3d_kernel<128, 128, 2, 4, 512, 2, 2>
                   : which == 1 ? (const void*)igemm_split3d_kernel<64, 128, 1, 4, 256, 2, 2>
                                : (const void*)igemm_split3d_kernel<128, 64, 2, 2, 256, 2, 2>;
    // the persistent and the ring kernel address the output through a 32-bit buffer descriptor; a larger output (per-GPU
    // batches around 512 on the first trunk units) falls back to the one-tile kernel's 64-bit pointer stores
    const bool big_out = (long)p.M * d->ldy * 4 >= (1L << 31);
    if (tail) {
        // statistics-only / fused-tail passes: always the persistent kernel (whole tiles, then K ranges of the tail tiles)
        if (terms != 3 || c.bm != 128 || c.bn != 128 || big_out || d->K % 128 || d->ldy != d->K || d->ldw != d->K)
            return fail(ACIMG_EINVAL, "conv2d_fwd_split3p (two-pass): needs 128x128 tiles, K %% 128 == 0, dense output < 2 GiB");
        const size_t lds_t = (size_t)2 * 4 * 128 * 64 + 4 * 2 * 128 * 4;
        const void* ft = tail->mode == 1 ? (const void*)igemm_split3dp_kernel<32, 0, 3, 1>
                       : tail->mode == 2 ? (const void*)igemm_split3dp_kernel<32, 0, 3, 2>
                                         : (const void*)igemm_split3dp_kernel<32, 0, 3, 3>;
        const int Pt = resident_slots(7 + tail->mode, ft, 512, lds_t);
        TailPlan tt{T, 1, 0};
        if (ws && ws_bytes >= acimg_conv2d_fwd_split3p_workspace(d)) tt = pick_tail(T, Pt, p.kiters, TS_MAX_UNITS);
        p.ts_whole = tt.whole; p.ts_s = tt.s;
        p.ts_counters = static_cast<int*>(ws);
        p.ts_partial = ws ? reinterpret_cast<float*>(static_cast<char*>(ws) + TS_COUNTER_BYTES) : nullptr;
        const int units = tt.whole + tt.rem * tt.s;
        const int nwg = std::min(units, Pt);
        if (tail->mode == 1) {
            e.Y = nullptr;
            hipLaunchKernelGGL((igemm_split3dp_kernel<32, 0, 3, 1>), dim3(nwg), dim3(512), lds_t, st, p, units, nwg);
            return check_launch("conv2d_fwd_split3p_stats");
        }
        const long oplane = (long)acimg_split_plane_bytes((long)p.M, d->K);
        if (!aligned16(tail->sc_planes) || !aligned16(tail->out_planes) || (tail->out_lo_off & 15) ||
            (long)tail->out_lo_off < oplane || (long)tail->out_lo_off + oplane >= (1L << 31))
            return fail(ACIMG_EINVAL, "conv2d_fwd_split3p_tail: unaligned / overlapping / >= 2 GiB split-format operands");
        e.Y = nullptr; e.stats = nullptr;
        e.f_scale = tail->scale; e.f_shift = tail->shift;
        e.f_sc = static_cast<const char*>(tail->sc_planes);
        e.f_out = static_cast<char*>(tail->out_planes); e.f_out_lo = (unsigned)tail->out_lo_off;
        e.f_out_bytes = (unsigned)(tail->out_lo_off + oplane);
        if (tail->mode == 3) {
            e.f_scale2 = tail->scale2; e.f_shift2 = tail->shift2;
            e.f_sc_lo = 0; e.f_sc_bytes = (unsigned)((long)p.M * d->K * 4);      // < 2 GiB: big_out was refused above
            hipLaunchKernelGGL((igemm_split3dp_kernel<32, 0, 3, 3>), dim3(nwg), dim3(512), lds_t, st, p, units, nwg);
            return check_launch("conv2d_fwd_split3p_tail_proj");
        }
        if ((tail->sc_lo_off & 15) || (long)tail->sc_lo_off < oplane || (long)tail->sc_lo_off + oplane >= (1L << 31))
            return fail(ACIMG_EINVAL, "conv2d_fwd_split3p_tail: unaligned / overlapping / >= 2 GiB shortcut planes");
        e.f_sc_lo = (unsigned)tail->sc_lo_off;
        e.f_sc_bytes = (unsigned)(tail->sc_lo_off + oplane);
        hipLaunchKernelGGL((igemm_split3dp_kernel<32, 0, 3, 2>), dim3(nwg), dim3(512), lds_t, st, p, units, nwg);
        return check_launch("conv2d_fwd_split3p_tail");
    }
    if (halo_on(d, terms)) {
        // one tile per workgroup, K walked as (channel chunk, tap) over one staged patch per chunk; tail tiles in K ranges
        const size_t lds_h = (size_t)2 * (HALO_NB * 1024 + 64) + 2 * 2 * 128 * 64;
        const void* fh = (const void*)igemm_split3h_kernel<HALO_NB>;
        static bool attr_h = false;                  // > 64 KiB of dynamic LDS needs the opt-in once per process
        if (!attr_h) {
            (void)hipFuncSetAttribute(fh, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_h);
            attr_h = true;
        }
        const int Ph = resident_slots(7, fh, 512, lds_h);
        TailPlan th{T, 1, 0};
        if (ws && ws_bytes >= acimg_conv2d_fwd_split3p_workspace(d)) th = pick_tail(T, Ph, p.kiters, TS_MAX_UNITS);
        p.ts_whole = th.whole; p.ts_s = th.s;
        p.ts_counters = static_cast<int*>(ws);
        p.ts_partial = ws ? reinterpret_cast<float*>(static_cast<char*>(ws) + TS_COUNTER_BYTES) : nullptr;
        hipLaunchKernelGGL((igemm_split3h_kernel<HALO_NB>), dim3(th.whole + th.rem * th.s), dim3(512), lds_h, st, p);
        return check_launch("conv2d_fwd_split3p (halo)");
    }
    if (const int rr = big_out ? 0 : ring_rows(d, terms)) {
#if defined(ACIMG_STAMP) || defined(ACIMG_ABLATE)
        p.slab = g_stamp_buf;
        p.flip = g_stamp_nostore;
#endif
        // one workgroup per CU walks units blockIdx.x, blockIdx.x + P, ... (whole tiles, then K ranges of the tail tiles)
        p.ras_tiles_m = cdiv(p.M, rr);
        p.ras_gm = std::max(1, 32 / p.ras_gn);       // 32 resident tiles per XCD
        const int Tr = p.ras_tiles_m * p.ras_tiles_n;
        const size_t lds_r = (size_t)3 * 2 * (rr + 128) * 64 + 4 * 2 * 128 * 4;
        const void* fr = rr == 256 ? (const void*)igemm_split3r_kernel<4> : (const void*)igemm_split3r_kernel<2>;
        static bool attr[2] = {false, false};        // > 64 KiB of dynamic LDS needs the opt-in once per process
        if (!attr[rr == 256]) {
            (void)hipFuncSetAttribute(fr, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_r);
            attr[rr == 256] = true;
        }
        const int Pr = resident_slots(rr == 256 ? 5 : 6, fr, 512, lds_r);
        TailPlan tr{Tr, 1, 0};
        if (ws && ws_bytes >= acimg_conv2d_fwd_split3p_workspace(d))
            tr = pick_tail(Tr, Pr, p.kiters, TS_MAX_UNITS * 128 / rr);
        p.ts_whole = tr.whole; p.ts_s = tr.s;
        p.ts_counters = static_cast<int*>(ws);
        p.ts_partial = ws ? reinterpret_cast<float*>(static_cast<char*>(ws) + TS_COUNTER_BYTES) : nullptr;
        const int units = tr.whole + tr.rem * tr.s;
        const int nwg = std::min(units, Pr);
        p.splits = g_cfg.trunk_stagger > 0 ? 2 : 1;   // ring kernel: waves 4-7 do a step's scalar work before their first MFMA group
        if (rr == 256) hipLaunchKernelGGL((igemm_split3r_kernel<4>), dim3(nwg), dim3(512), lds_r, st, p, units, nwg);
        else hipLaunchKernelGGL((igemm_split3r_kernel<2>), dim3(nwg), dim3(512), lds_r, st, p, units, nwg);
        return check_launch("conv2d_fwd_split3p (ring)");
    }
    const bool persistent = !big_out && split3p_persistent(c, T);
    // (a 64-deep K step - whole 128-byte operand rows, one workgroup per CU - was measured slower on every trunk shape in
    //  round 2 and removed when the operands moved to LDS-tile order, which gives whole-line requests at two per CU)
    const size_t lds_p = (size_t)2 * 4 * 128 * 64 + 4 * 2 * 128 * 4;    // 2 stages + statistics scratch
    TailPlan tp{T, 1, 0};
    const int P = !persistent ? resident_slots(which, fn, which == 0 ? 512 : 256, lds_bytes)
                              : resident_slots(3, (const void*)igemm_split3dp_kernel<32, 0>, 512, lds_p);
    if (ws && ws_bytes >= acimg_conv2d_fwd_split3p_workspace(d) && which == 0)
        tp = pick_tail(T, P, p.kiters, TS_MAX_UNITS);
    p.ts_whole = tp.whole; p.ts_s = tp.s;
    p.ts_counters = static_cast<int*>(ws);
    p.ts_partial = ws ? reinterpret_cast<float*>(static_cast<char*>(ws) + TS_COUNTER_BYTES) : nullptr;
    const int n_units = tp.whole + tp.rem * tp.s;
    const dim3 grid(n_units);
#if defined(ACIMG_STAMP) || defined(ACIMG_ABLATE)
    p.slab = g_stamp_buf;
    p.flip = g_stamp_nostore;
#endif
    if (persistent) {
        p.splits = g_cfg.trunk_stagger > 0 ? g_cfg.trunk_stagger * (p.kiters * 2500 + 8000) / 100 : 1;
        // a workgroup per resident slot walks units blockIdx.x, blockIdx.x + P, ...: whole tiles first (with the
        // next tile's first operand stage and addresses prepared under the current tile's last K step and output
        // stores), then the K ranges of the tail tiles
        const int nwg = std::min(n_units, P);
        if (terms == 1) {
            hipLaunchKernelGGL((igemm_split3dp_kernel<32, 0, 1>), dim3(nwg), dim3(512), lds_p, st, p, n_units, nwg);
        } else if (g_cfg.trunk_dma_pos == 1) {
            hipLaunchKernelGGL((igemm_split3dp_kernel<32, 1>), dim3(nwg), dim3(512), lds_p, st, p, n_units, nwg);
        } else {
            hipLaunchKernelGGL((igemm_split3dp_kernel<32, 0>), dim3(nwg), dim3(512), lds_p, st, p, n_units, nwg);
        }
    } else if (terms == 1) {
        if (which == 0)
            hipLaunchKernelGGL((igemm_split3d_kernel<128, 128, 2, 4, 512, 2, 2, 1>), grid, dim3(512), lds_bytes, st, p);
        else if (which == 1)
            hipLaunchKernelGGL((igemm_split3d_kernel<64, 128, 1, 4, 256, 2, 2, 1>), grid, dim3(256), lds_bytes, st, p);
        else
            hipLaunchKernelGGL((igemm_split3d_kernel<128, 64, 2, 2, 256, 2, 2, 1>), grid, dim3(256), lds_bytes, st, p);
    } else if (which == 0)
        hipLaunchKernelGGL((igemm_split3d_kernel<128, 128, 2, 4, 512, 2, 2>), grid, dim3(512), lds_bytes, st, p);
    else if (which == 1)
        hipLaunchKernelGGL((igemm_split3d_kernel<64, 128, 1, 4, 256, 2, 2>), grid, dim3(256), lds_bytes, st, p);
    else
        hipLaunchKernelGGL((igemm_split3d_kernel<128, 64, 2, 2, 256, 2, 2>), grid, dim3(256), lds_bytes, st, p);
    return check_launch("conv2d_fwd_split3p");
}

int acimg_conv2d_fwd_split3p(const AcimgConvDesc* d, const void* x_planes, size_t x_lo_off, const void* wsplit,
                             float* y, float* stats, void* ws, size_t ws_bytes, void* stream) {
    return fwd_presplit(d, x_planes, x_lo_off, wsplit, y, stats, ws, ws_bytes, stream, 3);
}

int acimg_conv2d_fwd_split1p(const AcimgConvDesc* d, const void* x_planes, size_t x_lo_off, const void* wsplit,
                             float* y, float* stats, void* ws, size_t ws_bytes, void* stream) {
    return fwd_presplit(d, x_planes, x_lo_off, wsplit, y, stats, ws, ws_bytes, stream, 1);
}

int acimg_conv2d_fwd_split3p_stats(const AcimgConvDesc* d, const void* x_planes, size_t x_lo_off, const void* wsplit,
                                   float* stats, void* ws, size_t ws_bytes, void* stream) {
    if (!stats) return fail(ACIMG_EINVAL, "conv2d_fwd_split3p_stats: null statistics buffer");
    const Split3pTail t{1, nullptr, nullptr, nullptr, 0, nullptr, 0, nullptr, nullptr};
    return fwd_presplit(d, x_planes, x_lo_off, wsplit, nullptr, stats, ws, ws_bytes, stream, 3, &t);
}

int acimg_conv2d_fwd_split3p_tail(const AcimgConvDesc* d, const void* x_planes, size_t x_lo_off, const void* wsplit,
                                  const float* scale, const float* shift, const void* sc_planes, size_t sc_lo_off,
                                  void* out_planes, size_t out_lo_off, void* ws, size_t ws_bytes, void* stream) {
    if (!scale || !shift || !sc_planes || !out_planes || !aligned16(scale) || !aligned16(shift))
        return fail(ACIMG_EINVAL, "conv2d_fwd_split3p_tail: null / unaligned scale, shift, shortcut or output");
    const Split3pTail t{2, scale, shift, sc_planes, sc_lo_off, out_planes, out_lo_off, nullptr, nullptr};
    return fwd_presplit(d, x_planes, x_lo_off, wsplit, nullptr, nullptr, ws, ws_bytes, stream, 3, &t);
}

int acimg_conv2d_fwd_split3p_tail_proj(const AcimgConvDesc* d, const void* x_planes, size_t x_lo_off, const void* wsplit,
                                       const float* scale, const float* shift, const float* sc32, const float* sc_scale,
                                       const float* sc_shift, void* out_planes, size_t out_lo_off, void* ws, size_t ws_bytes,
                                       void* stream) {
    if (!scale || !shift || !sc32 || !sc_scale || !sc_shift || !out_planes || !aligned16(scale) || !aligned16(shift) ||
        !aligned16(sc_scale) || !aligned16(sc_shift))
        return fail(ACIMG_EINVAL, "conv2d_fwd_split3p_tail_proj: null / unaligned scale, shift, shortcut or output");
    const Split3pTail t{3, scale, shift, sc32, 0, out_planes, out_lo_off, sc_scale, sc_shift};
    return fwd_presplit(d, x_planes, x_lo_off, wsplit, nullptr, nullptr, ws, ws_bytes, stream, 3, &t);
}

/* ---- tap-GEMM helpers (see the kernels above) ---- */
static int tapconv_check(const AcimgConvDesc* d, const char* who) {
    int rc = check_desc(d, who);
    if (rc) return rc;
    if (d->stride != 1 || d->pad_t || d->pad_l || d->OH != d->H - d->R + 1 || d->OW != d->W - d->S + 1)
        return fail(ACIMG_EINVAL, "%s: stride-1 VALID convolutions only", who);
    if (d->K > 64) return fail(ACIMG_EINVAL, "%s: K=%d > 64 (this form is for few output channels)", who, d->K);
    return ACIMG_OK;
}

int acimg_tapconv_stats_rows(const AcimgConvDesc* d) { return cdiv(d->N * d->OH * d->OW, TG_PPB); }

int acimg_tapconv_pack(const AcimgConvDesc* d, const float* w, float* wt, int ldwt, void* stream) {
    int rc = tapconv_check(d, "tapconv_pack");
    if (rc) return rc;
    const int TK = d->R * d->S * d->K;
    if (!w || !wt || ldwt < TK) return fail(ACIMG_EINVAL, "tapconv_pack: null pointer or ldwt < R*S*K");
    hipLaunchKernelGGL(tapconv_pack_kernel, dim3(cdiv((long)d->C * TK, 256)), dim3(256), 0, (hipStream_t)stream, w,
                       d->R * d->S, d->C, d->K, d->ldw, wt, ldwt);
    return check_launch("tapconv_pack");
}

int acimg_tapconv_unpack(const AcimgConvDesc* d, const float* dwt, int ldwt, const float* w, float decay, float* dw,
                         void* stream) {
    int rc = tapconv_check(d, "tapconv_unpack");
    if (rc) return rc;
    const int TK = d->R * d->S * d->K;
    if (!dwt || !dw || ldwt < TK) return fail(ACIMG_EINVAL, "tapconv_unpack: null pointer or ldwt < R*S*K");
    hipLaunchKernelGGL(tapconv_unpack_kernel, dim3(cdiv((long)d->R * d->S * d->C * d->K, 256)), dim3(256), 0,
                       (hipStream_t)stream, dwt, ldwt, d->R * d->S, d->C, d->K, d->ldw, w, decay, dw);
    return check_launch("tapconv_unpack");
}

int acimg_tapconv_gather(const AcimgConvDesc* d, const float* z, int ldz, float* y, float* stats, void* stream) {
    int rc = tapconv_check(d, "tapconv_gather");
    if (rc) return rc;
    const int TK = d->R * d->S * d->K;
    if (!z || !y || ldz < TK || d->ldy < d->K) return fail(ACIMG_EINVAL, "tapconv_gather: null pointer or ldz < R*S*K");
    const long Mout = (long)d->N * d->OH * d->OW;
    hipLaunchKernelGGL(tapconv_gather_kernel, dim3(cdiv(Mout, TG_PPB)), dim3(256), (size_t)TG_PPB * d->K * 4,
                       (hipStream_t)stream, z, ldz, d->H, d->W, d->R, d->S, d->K, d->OH, d->OW, Mout, y, d->ldy, stats,
                       d->ldw);
    return check_launch("tapconv_gather");
}

int acimg_tapconv_scatter(const AcimgConvDesc* d, const float* gy, int ldgy, float* gz, int ldgz, void* stream) {
    int rc = tapconv_check(d, "tapconv_scatter");
    if (rc) return rc;
    const int TK = d->R * d->S * d->K;
    if (!gy || !gz || ldgz < TK || ldgy < d->K) return fail(ACIMG_EINVAL, "tapconv_scatter: null pointer or ldgz < R*S*K");
    const long Min = (long)d->N * d->H * d->W;
    hipLaunchKernelGGL(tapconv_scatter_kernel, dim3(cdiv(Min * TK, 256)), dim3(256), 0, (hipStream_t)stream, gy, ldgy,
                       d->H, d->W, d->R, d->S, d->K, d->OH, d->OW, Min, gz, ldgz);
    return check_launch("tapconv_scatter");
}

/* weight + bias gradient on the bf16x3 MFMA path (same contract as acimg_conv2d_wgrad) */
static int wgrad_split_onthefly(const AcimgConvDesc* d, const float* x, const float* gy, int ldgy, float* dw, float* db,
                                void* ws, size_t ws_bytes, void* stream, int terms);

int acimg_conv2d_wgrad_split3(const AcimgConvDesc* d, const float* x, const float* gy, int ldgy,
                              float* dw, float* db, void* ws, size_t ws_bytes, void* stream) {
    return wgrad_split_onthefly(d, x, gy, ldgy, dw, db, ws, ws_bytes, stream, 3);
}

/* the same with x and gy rounded to bf16, one MFMA per product */
int acimg_conv2d_wgrad_bf16(const AcimgConvDesc* d, const float* x, const float* gy, int ldgy,
                            float* dw, float* db, void* ws, size_t ws_bytes, void* stream) {
    return wgrad_split_onthefly(d, x, gy, ldgy, dw, db, ws, ws_bytes, stream, 1);
}

static int wgrad_split_onthefly(const AcimgConvDesc* d, const float* x, const float* gy, int ldgy, float* dw, float* db,
                                void* ws, size_t ws_bytes, void* stream, int terms) {
    int rc = check_desc(d, "conv2d_wgrad_split3");
    if (rc) return rc;
    const int kp = up4(d->K);
    if (kp > ldgy || kp > d->ldw) return fail(ACIMG_EINVAL, "conv2d_wgrad_split3: padded K exceeds ldgy/ldw");
    WgradParams p{};
    p.X = x; p.H = d->H; p.W = d->W; p.C = d->C; p.ldx = d->ldx;
    p.OH = d->OH; p.OW = d->OW; p.R = d->R; p.S = d->S; p.stride = d->stride; p.pad_t = d->pad_t; p.pad_l = d->pad_l;
    p.M = d->N * d->OH * d->OW; p.KK = d->R * d->S * d->C;
    p.G = gy; p.ldg = ldgy; p.Ngemm = kp; p.Nld = kp; p.ldo = d->ldw;
    return launch_wgrad(p, dw, db, ws, ws_bytes, (hipStream_t)stream, true, terms);
}

/* precision of a conv layer's entry points: 0 = acimg_conv2d_fwd / _wgrad (fp32-class), 1 = _split3, 2 = _bf16 */
int acimg_conv2d_affine_input_ok(const AcimgConvDesc* d, int precision) {
    if (!d || precision < 0 || precision > 2 || check_desc(d, "conv2d_affine_input_ok")) return 0;
    WgradParams p{};
    p.H = d->H; p.W = d->W; p.C = d->C; p.ldx = d->ldx;
    p.OH = d->OH; p.OW = d->OW; p.R = d->R; p.S = d->S; p.stride = d->stride; p.pad_t = d->pad_t; p.pad_l = d->pad_l;
    p.M = d->N * d->OH * d->OW; p.KK = d->R * d->S * d->C;
    p.Ngemm = up4(d->K); p.Nld = p.Ngemm; p.ldo = d->ldw;
    const bool fwd = precision == 0 ? few16_fwd_shape(d) : conv_halo16_fwd_shape(d);
    return fwd && wgrad_halo16_ok(p, precision != 0) ? 1 : 0;
}

int acimg_conv2d_wgrad_affine(const AcimgConvDesc* d, int precision, const float* x, const float* in_scale, const float* in_shift,
                              int in_relu, const float* gy, int ldgy, float* dw, float* db, void* ws, size_t ws_bytes,
                              void* stream) {
    int rc = check_desc(d, "conv2d_wgrad_affine");
    if (rc) return rc;
    if (precision < 0 || precision > 2 || !in_scale || !in_shift)
        return fail(ACIMG_EINVAL, "conv2d_wgrad_affine: precision 0 / 1 / 2 and both halves of the affine");
    const int kp = up4(d->K);
    if (kp > ldgy || kp > d->ldw) return fail(ACIMG_EINVAL, "conv2d_wgrad_affine: padded K exceeds ldgy/ldw");
    WgradParams p{};
    p.X = x; p.H = d->H; p.W = d->W; p.C = d->C; p.ldx = d->ldx;
    p.OH = d->OH; p.OW = d->OW; p.R = d->R; p.S = d->S; p.stride = d->stride; p.pad_t = d->pad_t; p.pad_l = d->pad_l;
    p.M = d->N * d->OH * d->OW; p.KK = d->R * d->S * d->C;
    p.G = gy; p.ldg = ldgy; p.Ngemm = kp; p.Nld = kp; p.ldo = d->ldw;
    p.a_scale = in_scale; p.a_shift = in_shift; p.a_relu = in_relu;
    return launch_wgrad(p, dw, db, ws, ws_bytes, (hipStream_t)stream, precision != 0, precision == 2 ? 1 : 3);
}

}  // extern "C"
