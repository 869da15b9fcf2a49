// bf16x3 weight-gradient kernel for gfx950:  dW[kk][n] = sum_m X(m, kk) * G[m][n]   (kk = (tap, c) gathered)
//
// The reduction index is the PIXEL m, while both operands are stored pixel-major in HBM (x[m][c], gy[m][n]),
// i.e. "k" is the slow dimension of both tiles.  Instead of transposing on the way into LDS (4x the LDS write
// instructions) the tiles are kept [pixel][column] in LDS and the MFMA fragments are fetched with gfx950's
// transposing read `ds_read_b64_tr_b16`: per 16-lane group it returns, column-major, a 4(pixel) x 16(column)
// block, exactly the 16x16x32 operand layout (lane group g owns pixels 8g..8g+7 -> two reads per fragment).
//
//   operands: fp32 in HBM, split on the fly into bf16 hi/lo (gradients ~1e-7 underflow fp16; bf16 keeps fp32's
//             range), three MFMAs per product, fp32 accumulate;
//   LDS image: 32 pixels x 128 columns per plane, 256-byte rows, 32-byte chunk index XOR (q | (g&1)<<2) so the
//             8 row segments a 32-lane half touches per tr-read land on distinct bank groups;
//   roles:    G^T in the A slot, X in the B slot -> a lane's 4 accumulators are 4 consecutive n of one dW row:
//             16-byte stores into the slab;
//   the bias gradient rides along as an implicit all-ones column of X (row KK of the output).
#pragma once
#include "igemm_split3_kernel.hpp"

namespace acimg {

typedef short s16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ b16x8 tr_frag(const char* tile, int row0, int colbyte, int lane) {
    // block rows row0..row0+3 and row0+4..row0+7 (row0 = 8g), 16 columns starting at byte `colbyte` of the row
    const int t = lane & 15, q = t >> 2, p = t & 3, g = lane >> 4;
    const int f = q | ((g & 1) << 2);
    const int c32 = colbyte >> 5;
    const int off = ((c32 ^ f) << 5) + p * 8;
    const s16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4*)(tile + (row0 + q) * 256 + off));
    const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4*)(tile + (row0 + 4 + q) * 256 + off));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 v = {lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
    return __builtin_bit_cast(b16x8, v);
}

// BN = 128 or 64 output columns per block; 128 dW rows per block; 32 pixels per K step
// TERMS = 1: both operands rounded to bf16 and multiplied once (`acimg_conv2d_wgrad_bf16`), lo planes unused
// NTHR = 256: 4 waves of 64 dW rows x BN/2 columns; NTHR = 512: 8 waves of 64 x BN/4 (half the load / split work per
// thread and twice the waves to hide it: the K step is paced by the register-path staging, not by the MFMAs)
template <int BN, int TERMS = 3, int NTHR = 256>
__global__ __launch_bounds__(NTHR) void wgrad_split3_kernel(const WgradParams p) {
    constexpr int BMO = 128, BKR = 32;
    constexpr int WN = NTHR / 128;            // wave columns (2 wave rows of 64 dW rows each)
    constexpr int XRS = NTHR / 32;            // X rows per pass
    constexpr int PLANE = BKR * 256;          // bytes of one 32 x 128 bf16 plane (G planes use the same pitch)
    constexpr int STAGE = 4 * PLANE;          // Xh | Xl | Gh | Gl
    constexpr int WTN = BN / WN;
    constexpr int TM = 4, TN = WTN / 16;      // per wave: 64 dW rows x WTN columns
    constexpr int GQ = BN / 4;                // float4 per G row
    constexpr int GRPP = NTHR / GQ;           // G rows per pass
    constexpr int NG = BKR / GRPP;            // G float4 per thread
    constexpr int NX = BKR / XRS;             // X float4 per thread (32 rows x 32 float4 / NTHR)
    static_assert(WTN % 16 == 0 && NG >= 1 && NX >= 1, "tile / thread mapping");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* const lds = reinterpret_cast<char*>(smem);

    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    const int li = lane & 15, g = lane >> 4;
    const int kk0 = blockIdx.x * BMO, n0 = blockIdx.y * BN;
    const int m_begin = blockIdx.z * p.rows_per_split;
    const int m_end = min(p.M, m_begin + p.rows_per_split);

    // ---- this thread's X column (4 consecutive kk) and pixel rows -----------------------------------
    const int xq = tid & 31;                  // float4 index inside the 128-wide row
    const int xrow0 = tid >> 5;               // rows xrow0 + XRS j
    const int kk = kk0 + xq * 4;
    const bool kk_ok = kk < p.KK;
    const bool kk_ones = p.db_out != nullptr && kk == p.KK;
    int tr = 0, ts = 0, tc = 0;
    if (kk_ok) {
        const int tap = kk / p.C;
        tc = kk - tap * p.C;
        tr = tap / p.S;
        ts = tap - tr * p.S;
    }
    int x_img[NX], x_oh[NX], x_ow[NX];
    {
        const int ohw = p.OH * p.OW;
#pragma unroll
        for (int j = 0; j < NX; ++j) {
            const int m = m_begin + xrow0 + XRS * j;
            x_img[j] = m / ohw;
            const int rem = m - x_img[j] * ohw;
            x_oh[j] = rem / p.OW;
            x_ow[j] = rem - x_oh[j] * p.OW;
        }
    }
    const int gq = tid % GQ;
    const int grow0 = tid / GQ;
    const int gn = n0 + gq * 4;
    const bool gn_ok = gn < p.Nld;

    float4 rx[NX], rg[NG];
    int mb_next = m_begin;                    // first pixel of the next tile to load

    auto load_tiles = [&]() {
#pragma unroll
        for (int j = 0; j < NX; ++j) {
            const int m = mb_next + xrow0 + XRS * j;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (m < m_end) {
                if (kk_ones) v.x = 1.f;
                if (kk_ok) {
                    const int ih = x_oh[j] * p.stride - p.pad_t + tr;
                    const int iw = x_ow[j] * p.stride - p.pad_l + ts;
                    if ((unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W)
                        v = *reinterpret_cast<const float4*>(p.X + ((long)(x_img[j] * p.H + ih) * p.W + iw) * p.ldx + tc);
                }
            }
            rx[j] = v;
            // advance this row slot by 32 pixels
            x_ow[j] += BKR;
            while (x_ow[j] >= p.OW) {
                x_ow[j] -= p.OW;
                if (++x_oh[j] == p.OH) {
                    x_oh[j] = 0;
                    ++x_img[j];
                }
            }
        }
#pragma unroll
        for (int j = 0; j < NG; ++j) {
            const int m = mb_next + grow0 + j * GRPP;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gn_ok && m < m_end) v = *reinterpret_cast<const float4*>(p.G + (long)m * p.ldg + gn);
            rg[j] = v;
        }
        mb_next += BKR;
    };

    auto store_tiles = [&](int buf) {
        char* st = lds + buf * STAGE;
#pragma unroll
        for (int j = 0; j < NX; ++j) {
            const int row = xrow0 + XRS * j;
            const int f = (row & 3) | (((row >> 3) & 1) << 2);
            const int off = row * 256 + (((xq >> 2) ^ f) << 5) + ((xq & 3) << 3);
            uint2 hi, lo;
            split4<SplitBF16>(rx[j], hi, lo);
            *reinterpret_cast<uint2*>(st + off) = hi;
            if (TERMS == 3) *reinterpret_cast<uint2*>(st + PLANE + off) = lo;
        }
#pragma unroll
        for (int j = 0; j < NG; ++j) {
            const int row = grow0 + j * GRPP;
            const int f = (row & 3) | (((row >> 3) & 1) << 2);
            const int off = row * 256 + (((gq >> 2) ^ f) << 5) + ((gq & 3) << 3);
            uint2 hi, lo;
            split4<SplitBF16>(rg[j], hi, lo);
            *reinterpret_cast<uint2*>(st + 2 * PLANE + off) = hi;
            if (TERMS == 3) *reinterpret_cast<uint2*>(st + 3 * PLANE + off) = lo;
        }
    };

    f32x4 acc[TN][TM];   // acc[tn][tm]: rows (regs) = 4 consecutive n, column (lane) = dW row kk
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (m_begin < m_end) {
        load_tiles();
        store_tiles(0);
        if (mb_next < m_end) load_tiles();
    }
    __syncthreads();

    int cur = 0;
    for (int mb = m_begin; mb < m_end; mb += BKR) {
        const char* st = lds + cur * STAGE;
        b16x8 xh[TM], xl[TM];
#pragma unroll
        for (int j = 0; j < TM; ++j) {
            const int colbyte = (wm * 64 + j * 16) * 2;
            xh[j] = tr_frag(st, 8 * g, colbyte, lane);
            if (TERMS == 3) xl[j] = tr_frag(st + PLANE, 8 * g, colbyte, lane);
        }
#pragma unroll
        for (int i = 0; i < TN; ++i) {
            const int colbyte = (wn * WTN + i * 16) * 2;
            const b16x8 gh = tr_frag(st + 2 * PLANE, 8 * g, colbyte, lane);
            b16x8 gl;
            if (TERMS == 3) gl = tr_frag(st + 3 * PLANE, 8 * g, colbyte, lane);
#pragma unroll
            for (int j = 0; j < TM; ++j) {
                if (TERMS == 3) {
                    acc[i][j] = SplitBF16::mfma(gl, xh[j], acc[i][j]);
                    acc[i][j] = SplitBF16::mfma(gh, xl[j], acc[i][j]);
                }
                acc[i][j] = SplitBF16::mfma(gh, xh[j], acc[i][j]);
            }
        }
        const bool more = mb + BKR < m_end;
        if (more) {
            store_tiles(cur ^ 1);
            if (mb_next < m_end) load_tiles();
        }
        __syncthreads();
        cur ^= 1;
    }

    // ---- store: lane = dW row (kk), 4 regs = 4 consecutive columns n ---------------------------------------
    float* out = p.out + (long)blockIdx.z * p.KK * p.ldo;
#pragma unroll
    for (int j = 0; j < TM; ++j) {
        const int row = kk0 + wm * 64 + j * 16 + li;
        if (row > p.KK || (row == p.KK && p.db_out == nullptr)) continue;
        float* dst = row < p.KK ? out + (long)row * p.ldo : p.db_out + (long)blockIdx.z * p.ldo;
#pragma unroll
        for (int i = 0; i < TN; ++i) {
            const int n = n0 + wn * WTN + i * 16 + 4 * g;
            if (n + 3 < p.Ngemm)
                *reinterpret_cast<float4*>(dst + n) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
            else
                for (int k = 0; k < 4; ++k)
                    if (n + k < p.Ngemm) dst[n + k] = acc[i][j][k];
        }
    }
}

}  // namespace acimg
