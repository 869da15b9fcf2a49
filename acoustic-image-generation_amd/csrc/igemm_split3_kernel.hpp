// f16x3 ("split fp16") implicit-GEMM forward convolution for gfx950.
//
// fp32 operands are split on the fly into hi = f16(x) and lo = f16(x - hi) (11 + 11 = 22 mantissa bits) and
// every 16x16x32 product is three fp16 MFMAs accumulated in fp32:  a*w ~= ah*wh + ah*wl + al*wh.  The dropped
// al*wl term and the split residuals are ~2^-22 relative per product, i.e. fp32-class results (the parity
// tests hold the same 1e-3 bar as the exact-f32 kernel; a bf16 split, 16 bits, measured 1.7e-3 at the end of
// the 53-layer trunk) at 3/16 of the fp32-MFMA cost per FLOP.  fp16's narrow range is handled with EXACT
// power-of-two scaling: weights are multiplied by 2^10 when they are split (|w| up to 63 stays finite, lo
// parts of typical weights stay normal), activations by 2^-2 (finite up to 2.6e5), and the epilogue
// multiplies the fp32 accumulators by 2^-8.
//
//   A (activations): fp32 NHWC in HBM, gathered like the fp32 kernel (branch-free buffer loads, deferred BN
//      scale/shift/ReLU applied while staging), split to hi/lo when written to LDS.
//   B (weights): pre-split + transposed by `split3_prepare_kernel` into [hi|lo][n][k] fp16 (k contiguous), so
//      both operands are read with one ds_read_b128 per 16x16x32 fragment.
//   LDS image: 64-byte rows (32 halves), 16-byte chunk index XOR ((row>>2)&3): conflict-free b128 reads for the
//      16 rows of a fragment without padding; 32 KiB per stage, double buffered.
//   MFMA roles swapped as in the fp32 kernel (weights in the A slot): a lane's 4 accumulators are 4 consecutive
//      output channels of one pixel -> shared 16-byte epilogue incl. the batch-norm statistics partials.
#pragma once
#include "igemm_kernel.hpp"

namespace acimg {

// SPLIT3_* constants and the h16 vector types live in common.hpp (shared with the elementwise producers)

// Element type of the split: fp16 (22 mantissa bits, needs the power-of-two range scaling; forward passes)
// or bf16 (16 mantissa bits, fp32's range, no scaling; backward passes, where gradients ~1e-7 would
// underflow fp16).
struct SplitF16 {
    typedef _Float16 T;
    typedef h16x8 V8;
    static constexpr float ASCALE = SPLIT3_ASCALE, WSCALE = SPLIT3_WSCALE, OUTSCALE = SPLIT3_OUTSCALE;
    static __device__ __forceinline__ f32x4 mfma(V8 a, V8 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
};
typedef __bf16 b16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 b16x4 __attribute__((ext_vector_type(4)));
struct SplitBF16 {
    typedef __bf16 T;
    typedef b16x8 V8;
    static constexpr float ASCALE = 1.f, WSCALE = 1.f, OUTSCALE = 1.f;
    static __device__ __forceinline__ f32x4 mfma(V8 a, V8 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
};

// LDS chunk swizzle: a tile row is 64 bytes = four 16-byte chunks; logical chunk kc of row `row` is stored at
// chunk kc ^ swz(row).  ds_read_b128 is serviced in the hardware lane groups {0-3,12-15,20-27},
// {4-11,16-19,28-31}, ... (not in natural 16-lane groups): each group sees all 16 fragment rows once, rows
// 0-3/12-15 with one k-chunk g and rows 4-11 with g^1.  swz = (0,3,2,1)[(row>>2)&3] is the assignment that
// makes the 16 lanes of every such group hit 16 distinct 16-byte slots of the 256-byte bank window
// (the plain (row>>2)&3 pattern measured 35 % of LDS cycles as bank conflicts).
__device__ __forceinline__ int swz(int row) { return (0 - (row >> 2)) & 3; }

template <typename TR>
__device__ __forceinline__ void split4(float4 v, uint2& hi, uint2& lo) {
    typedef typename TR::T T;
    typedef T T4 __attribute__((ext_vector_type(4)));
    if (TR::ASCALE != 1.f) { v.x *= TR::ASCALE; v.y *= TR::ASCALE; v.z *= TR::ASCALE; v.w *= TR::ASCALE; }
    const T4 h = {(T)v.x, (T)v.y, (T)v.z, (T)v.w};
    const T4 l = {(T)(v.x - (float)h[0]), (T)(v.y - (float)h[1]), (T)(v.z - (float)h[2]), (T)(v.w - (float)h[3])};
    hi = __builtin_bit_cast(uint2, h);
    lo = __builtin_bit_cast(uint2, l);
}

// Weight preparation: fp32 HWIO kernel w[tap][c][ldw] -> split + transposed planes out[2][Nrows][Ktot]
// (out[0] = hi, out[1] = lo of w * WSCALE, k contiguous).
//   FWD  : rows n = output channel,  k = tap*C + c            value W[tap][c][n]
//   DGRAD: rows n = input channel c, k = tap'*K + ko          value W[ntaps-1-tap'][n][ko]   (flipped taps)
template <typename TR, bool DGRAD>
__device__ __forceinline__ void split3_prepare_body(const float* w, int ntaps, int C, int K, int ldw, int Nrows,
                                                    typename TR::T* out, int bx, int by, float (*tile)[33]) {
    typedef typename TR::T T;
    const int Ktot = DGRAD ? ntaps * K : ntaps * C;
    const int k0 = bx * 32, n0 = by * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float v = 0.f;
        if constexpr (!DGRAD) {
            const int k = k0 + ty + 8 * i, n = n0 + tx;      // coalesced along n
            if (k < Ktot && n < ldw) v = w[(long)k * ldw + n];
            tile[ty + 8 * i][tx] = v;                        // tile[k][n]
        } else {
            const int n = n0 + ty + 8 * i, k = k0 + tx;      // coalesced along ko
            if (k < Ktot && n < C) {
                const int tp = k / K, ko = k - tp * K;
                v = w[((long)(ntaps - 1 - tp) * C + n) * ldw + ko];
            }
            tile[tx][ty + 8 * i] = v;                        // tile[k][n]
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int n = n0 + ty + 8 * i, k = k0 + tx;
        if (n < Nrows && k < Ktot) {
            const float v = tile[tx][ty + 8 * i] * TR::WSCALE;
            const T h = (T)v;
            out[(long)n * Ktot + k] = h;
            out[((long)Nrows + n) * Ktot + k] = (T)(v - (float)h);
        }
    }
}

template <typename TR, bool DGRAD>
__global__ __launch_bounds__(256) void split3_prepare_kernel(const float* w, int ntaps, int C, int K, int ldw,
                                                             int Nrows, typename TR::T* out) {
    __shared__ float tile[32][33];
    split3_prepare_body<TR, DGRAD>(w, ntaps, C, K, ldw, Nrows, out, blockIdx.x, blockIdx.y, tile);
}

// several kernels in ONE launch (the generator re-splits its trainable kernels every step: forward f16 images and
// flipped / transposed bf16 images for the data gradients): job table by value, workgroup -> job by block ranges
struct PrepJob {
    const float* w;
    void* out;
    int ntaps, C, K, ldw, Nrows, dgrad, tiles_k, block0;   // dgrad: 0 forward f16, 1 data-gradient bf16, 2 forward bf16
};
struct PrepJobs {
    PrepJob j[16];
    int n;
};
__global__ __launch_bounds__(256) void split3_prepare_multi_kernel(const PrepJobs jobs) {
    __shared__ float tile[32][33];
    int ji = 0;
#pragma unroll 1
    for (int i = 1; i < jobs.n; ++i)
        if ((int)blockIdx.x >= jobs.j[i].block0) ji = i;
    const PrepJob& J = jobs.j[ji];
    const int lb = blockIdx.x - J.block0;
    const int bx = lb % J.tiles_k, by = lb / J.tiles_k;
    if (J.dgrad == 1)
        split3_prepare_body<SplitBF16, true>(J.w, J.ntaps, J.C, J.K, J.ldw, J.Nrows, static_cast<__bf16*>(J.out), bx, by, tile);
    else if (J.dgrad == 2)
        split3_prepare_body<SplitBF16, false>(J.w, J.ntaps, J.C, J.K, J.ldw, J.Nrows, static_cast<__bf16*>(J.out), bx, by, tile);
    else
        split3_prepare_body<SplitF16, false>(J.w, J.ntaps, J.C, J.K, J.ldw, J.Nrows, static_cast<_Float16*>(J.out), bx, by, tile);
}

// TERMS = 3: the split product (fp32-class results).  TERMS = 1: 16-BIT OPERAND ARITHMETIC - activations and weights are
// rounded to TR::T (the hi halves) and multiplied once, fp32 accumulation: the arithmetic of a bf16 / fp16 mixed-
// precision network (BASELINE configs[1], `acimg_conv2d_*_bf16`); the lo planes are neither written nor read.
template <int BM, int BN, int WGM, int WGN, int NTHR, typename TR, int TERMS = 3>
__global__ __launch_bounds__(NTHR) void igemm_split3_kernel(const IgemmParams p) {
    static_assert(TERMS == 1 || TERMS == 3, "1 = rounded operands, 3 = split product");
    typedef typename TR::V8 V8;
    constexpr int BK = 32;
    constexpr int ROWB = BK * 2;                  // bytes per LDS row (32 halves)
    constexpr int A_BYTES = BM * ROWB;            // one of hi / lo
    constexpr int B_BYTES = BN * ROWB;
    constexpr int STAGE = 2 * A_BYTES + 2 * B_BYTES;
    constexpr int WTM = BM / WGM, WTN = BN / WGN;
    constexpr int TM = WTM / 16, TN = WTN / 16;
    constexpr int KQ = BK / 4;                    // float4 per A row
    constexpr int RPP = NTHR / KQ;
    constexpr int NA = BM / RPP;
    constexpr int BCH = 2 * BN * 4;               // 16-byte chunks of the B tile (hi + lo)
    constexpr int NB = BCH / NTHR;
    static_assert(WGM * WGN * 64 == NTHR && BM % RPP == 0 && BCH % NTHR == 0, "tile / thread mapping");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* const lds = reinterpret_cast<char*>(smem);

    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WGN, wn = wid % WGN;
    const int li = lane & 15, g = lane >> 4;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;

    const __amdgpu_buffer_rsrc_t rsA =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.A), 0, p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.B), 0, p.b_bytes, 0x00020000);

    const int it_end = p.kiters;
    const int Ktot = p.ntaps * p.C;               // fp16 elements per weight row

    // ---- A rows of this thread ---------------------------------------------------------------------
    const int kq = tid % KQ;
    const int arow0 = tid / KQ;
    int a_off[NA], a_ih0[NA], a_iw0[NA];
    {
        const int ohw = p.OH * p.OW;
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            const int m = m0 + arow0 + j * RPP;
            if (m < p.M) {
                const int img = m / ohw;
                const int rem = m - img * ohw;
                const int oh = rem / p.OW;
                const int ow = rem - oh * p.OW;
                a_ih0[j] = oh * p.stride - p.pad_t;
                a_iw0[j] = ow * p.stride - p.pad_l;
                a_off[j] = ((img * p.H + a_ih0[j]) * p.W + a_iw0[j]) * p.lda * 4;
            } else {
                a_ih0[j] = -(1 << 28);
                a_iw0[j] = -(1 << 28);
                a_off[j] = 0;
            }
        }
    }
    // ---- B chunks of this thread: chunk c -> (hi/lo, row, 16-byte k chunk) --------------------------------
    unsigned b_goff[NB];   // byte offset of (which, row) at k = 0, or OOB
    int b_lds[NB];         // byte offset inside the stage
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int c = tid + j * NTHR;
        const int which = c / (BN * 4);
        const int rem = c - which * (BN * 4);
        const int row = rem >> 2, kc = rem & 3;
        const int n = n0 + row;
        b_goff[j] = n < p.Nld ? (unsigned)((((long)which * p.Nld + n) * Ktot + kc * 8) * 2) : OOB;
        b_lds[j] = 2 * A_BYTES + which * B_BYTES + row * ROWB + ((kc ^ swz(row)) << 4);
    }

    int nit = 0, st_r = 0, st_s = 0, st_c0 = 0;
    // two prefetch register sets: tiles t+1 and t+2 are in flight while tile t is multiplied, so every
    // global load has TWO K steps to land (one step is only ~400 MFMA cycles here: with a single set the
    // kernel ran at the L2 round-trip latency, ~6900 cycles per step)
    float4 ra0[NA], ra1[NA];
    uint4 rb0[NB], rb1[NB];

    auto load_tiles = [&](float4 (&ra)[NA], uint4 (&rb)[NB]) {
        const int c = st_c0 + kq * 4;
        const bool affine = p.a_scale != nullptr;
        float4 sc4 = make_float4(1.f, 1.f, 1.f, 1.f), sh4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (affine) {
            sc4 = *reinterpret_cast<const float4*>(p.a_scale + c);
            sh4 = *reinterpret_cast<const float4*>(p.a_shift + c);
        }
        const int tapoff = ((st_r * p.W + st_s) * p.lda + c) * 4;
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            const int ih = a_ih0[j] + st_r, iw = a_iw0[j] + st_s;
            const bool ok = (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
            float4 v = buf_load4(rsA, ok ? (unsigned)(a_off[j] + tapoff) : OOB);
            if (affine) {
                v.x = v.x * sc4.x + sh4.x;
                v.y = v.y * sc4.y + sh4.y;
                v.z = v.z * sc4.z + sh4.z;
                v.w = v.w * sc4.w + sh4.w;
                if (p.a_relu) {
                    v.x = fmaxf(v.x, 0.f);
                    v.y = fmaxf(v.y, 0.f);
                    v.z = fmaxf(v.z, 0.f);
                    v.w = fmaxf(v.w, 0.f);
                }
                if (!ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            ra[j] = v;
        }
        const unsigned kbyte = (unsigned)(((st_r * p.S + st_s) * p.C + st_c0) * 2);
#pragma unroll
        for (int j = 0; j < NB; ++j)
            rb[j] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(
                                                  rsB, b_goff[j] == OOB ? OOB : b_goff[j] + kbyte, 0, 0));
        ++nit;
        st_c0 += BK;
        if (st_c0 == p.C) {
            st_c0 = 0;
            if (++st_s == p.S) {
                st_s = 0;
                ++st_r;
            }
        }
    };

    auto store_tiles = [&](int buf, const float4 (&ra)[NA], const uint4 (&rb)[NB]) {
        char* st = lds + buf * STAGE;
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            const int row = arow0 + j * RPP;
            uint2 hi, lo;
            split4<TR>(ra[j], hi, lo);
            const int off = row * ROWB + (((kq >> 1) ^ swz(row)) << 4) + ((kq & 1) << 3);
            *reinterpret_cast<uint2*>(st + off) = hi;
            if (TERMS == 3) *reinterpret_cast<uint2*>(st + A_BYTES + off) = lo;
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) *reinterpret_cast<uint4*>(st + b_lds[j]) = rb[j];
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto compute = [&](int buf) {
        const char* st = lds + buf * STAGE;
        V8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int row = wm * WTM + i * 16 + li;
            const int off = row * ROWB + ((g ^ swz(row)) << 4);
            ah[i] = *reinterpret_cast<const V8*>(st + off);
            if (TERMS == 3) al[i] = *reinterpret_cast<const V8*>(st + A_BYTES + off);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int row = wn * WTN + j * 16 + li;
            const int off = 2 * A_BYTES + row * ROWB + ((g ^ swz(row)) << 4);
            bh[j] = *reinterpret_cast<const V8*>(st + off);
            if (TERMS == 3) bl[j] = *reinterpret_cast<const V8*>(st + B_BYTES + off);
        }
        // rows of D = output channels (weights in the A slot), columns = pixels; small terms first.  All fragment
        // reads are issued before the first MFMA and the three terms of a product are a whole sweep apart, so no
        // MFMA waits on its predecessor's accumulator (per-accumulator order unchanged).
        __builtin_amdgcn_sched_barrier(0);
        if (TERMS == 3) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = TR::mfma(bl[j], ah[i], acc[i][j]);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = TR::mfma(bh[j], al[i], acc[i][j]);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = TR::mfma(bh[j], ah[i], acc[i][j]);
    };

    if (it_end > 0) {
        load_tiles(ra0, rb0);                       // tile 0
        store_tiles(0, ra0, rb0);
        if (nit < it_end) load_tiles(ra0, rb0);     // tile 1
        if (nit < it_end) load_tiles(ra1, rb1);     // tile 2
    }
    __syncthreads();

    // loop top (it even): LDS[0] = tile it, set 0 = tile it+1, set 1 = tile it+2
    for (int it = 0; it < it_end; it += 2) {
        compute(0);
        if (it + 1 < it_end) {
            store_tiles(1, ra0, rb0);
            if (nit < it_end) load_tiles(ra0, rb0);  // tile it+3
        }
        __syncthreads();
        if (it + 1 >= it_end) break;
        compute(1);
        if (it + 2 < it_end) {
            store_tiles(0, ra1, rb1);
            if (nit < it_end) load_tiles(ra1, rb1);  // tile it+4
        }
        __syncthreads();
    }

#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
            if (TR::OUTSCALE != 1.f) acc[i][j] *= TR::OUTSCALE;   // exact: undo the power-of-two operand scaling
    igemm_epilogue<BM, BN, WGM, WGN, NTHR, TM, TN>(p, acc, smem, m0, n0, wm, wn, li, g, tid);
}

}  // namespace acimg
