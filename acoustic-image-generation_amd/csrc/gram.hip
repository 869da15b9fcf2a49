// INPUT-SIDE batch-norm statistics of a 1x1 convolution (round 4): acimg_gram_stats.
//
// The expanding 1x1 conv of a bottleneck (conv3, models/resnet50.py:119-123: C -> 4C channels, slim batch_norm in
// batch-statistics mode) needs sum_p y_n(p) and sum_p y_n(p)^2 over all pixels BEFORE its fused tail can normalise
// (igemm_split3dp_kernel EPI 2 / 3).  Round 3 got them by running the conv's K loop twice (a statistics pass that stores
// nothing: 4 P C^2 MACs x 3 MFMAs per product).  They also follow from the conv's INPUT alone:
//
//     sum_p y_n   = w_n^T sx,            sx = sum_p x(p)                     (C numbers)
//     sum_p y_n^2 = w_n^T G w_n,         G  = sum_p x(p) x(p)^T              (C x C, P C^2 MACs: a quarter of the conv)
//
// and a quadratic form only sees the symmetric part of its matrix, so with x = xh + xl (the split planes)
//     w^T G w = w^T (Hh + 2 T) w,   Hh = sum xh xh^T,  T = sum xh xl^T      (xl xl^T ~ 2^-22 relative: dropped, as in the conv)
// i.e. TWO fp16 MFMAs per product instead of three, the second on the lo plane scaled by 2 (exact in fp16).
//
// Three launches:
//   gram_partial_kernel  V_s = sum over a pixel range of xh (xh + 2 xl)^T and the column sums, per (range, 128x128 block):
//                        K = pixels, both operands are the SAME brick planes [16 pixels][32 channels] copied to LDS by
//                        LDS-DMA (one contiguous KiB per request, the HBM image IS the LDS image) through a ring of 4
//                        stages (3 K steps in flight: the K loop of a workgroup is only ~16 steps long, so the latency of a
//                        step's bricks has to hide under the steps before it), fragments fetched with the transposing
//                        ds_read_b64_tr_b16 (pixel-major bricks -> k-contiguous MFMA operands; conflict free on the brick
//                        swizzle); diagonal blocks read their A fragments from the B image (a third fewer requests);
//                        the column sums ride along as two MFMAs against constant fragments (1.0 for hi, 0.5 for 2 lo).
//   gram_reduce_kernel   partials summed in double in split order (deterministic), scaled back (x16 / x4).
//   gram_quadfin_kernel  per 16 output channels: Y = V W[:, n0 .. n0+15] by exact-f32 MFMA (16x16x4, an fmaf chain; the
//                        j-th register of a 16-byte load feeds MFMA j on both operand sides), s2_n = sum_c W[c][n] Y[c][n] and
//                        s1_n = sum_c W[c][n] sx[c] in double, then exactly acimg_bn_finalize's arithmetic: scale / shift and
//                        the moving averages.
// fp32 accuracy is enough here (tests/test_ops_gpu.py::test_gram_statistics_match_fp64: mean and variance to 1e-6
// relative against fp64 statistics of the conv output; a numpy experiment at C = 512 put the fp32 quadratic form at 1e-6 of
// the variance, 10 x that of an fp64 one, three orders under the 1e-3 bar).
#include "common.hpp"

namespace acimg {

// (the few helpers shared with the trunk kernels are restated here: their headers define non-template kernels, which
//  may live in one translation unit only)
namespace {
typedef __attribute__((address_space(3))) void* lds_ptr_t;
constexpr unsigned OOB = 0x80000000u;      // >= num_records of every descriptor (tensors < 2 GiB): the DMA writes zeros
__device__ __forceinline__ int swz(int row) { return (0 - (row >> 2)) & 3; }   // brick chunk swizzle (igemm_split3_kernel.hpp)
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else static_assert(N == 2, "add the immediate");
}
}  // namespace

struct GramParams {
    const char* X;       // split-format planes (hi at X, lo at X + lo_off)
    unsigned x_bytes;    // extent for the buffer resource
    unsigned lo_off;
    int nblk16;          // 16-pixel blocks = ceil(rows / 16)
    int C;
    int nb;              // BC-channel blocks per side
    int S;               // pixel ranges
    int steps;           // K steps (4 pixel blocks = 64 pixels) per range
    float* part;         // [S][nb * nb][BC][BC]
    float* sxpart;       // [S][C]
};

typedef short g_s16x4 __attribute__((ext_vector_type(4)));
typedef short g_s16x8 __attribute__((ext_vector_type(8)));

// fragment of 16 channels (tile ct of the block side) x 32 pixels (half h of the 64-pixel K step) out of a region
// [4 pixel blocks][NCC bricks]: lane (li, g) gets channel 16 ct + li, pixels 32 h + 8 g .. + 7 (two transposing reads of 4
// pixel rows each)
template <int NCC>
__device__ __forceinline__ h16x8 gram_frag(const char* region, int ct, int h, int lane) {
    const int t = lane & 15, q = t >> 2, p = t & 3, g = lane >> 4;
    const int brick = ((2 * h + (g >> 1)) * NCC + (ct >> 1)) * 1024;
    const int kc = 2 * (ct & 1) + (p >> 1);
    const int row0 = 8 * (g & 1) + q, row1 = row0 + 4;
    const int a0 = brick + row0 * 64 + ((kc ^ swz(row0)) << 4) + 8 * (p & 1);
    const int a1 = brick + row1 * 64 + ((kc ^ swz(row1)) << 4) + 8 * (p & 1);
    const g_s16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) g_s16x4*)(region + a0));
    const g_s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) g_s16x4*)(region + a1));
    const g_s16x8 v = {lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
    return __builtin_bit_cast(h16x8, v);
}

template <int BC>
__global__ __launch_bounds__(512, 2) void gram_partial_kernel(const GramParams p) {
    constexpr int NCC = BC / 32;                 // 32-channel bricks per block side
    constexpr int KPB = 4;                       // pixel blocks per K step: 64 pixels = two 32-deep MFMA sweeps per barrier
    constexpr int REG = KPB * NCC * 1024;        // one region of a stage: KPB pixel blocks x NCC bricks
    // ring of stages [B hi | B lo | A hi] in NREG regions of LDS: an off-diagonal block has NSN stages of 3 regions, a
    // diagonal one (no A region: its A fragments are read from the B hi image) NSD stages of 2.  A workgroup's K loop is
    // short (~9 steps), each step a barrier phase: what was measured (profiles/r04/gram_*): 32-pixel steps took 1.2 us each
    // whatever the ring depth (4 or 6 stages) - one workgroup per CU has nobody to fill its barrier / request / read
    // phases - so a step carries two sweeps instead
    constexpr int NSD = BC == 128 ? 4 : 3, NSN = 3;
    constexpr int WGN = 4;
    constexpr int TM = BC / 2 / 16, TN = BC / WGN / 16;      // 128: 4 x 2 (wave tile 64 x 32); 64: 2 x 1
    constexpr int PWD = 2 * KPB * NCC / 8;       // DMA pieces per wave and step, diagonal block (B hi, B lo)
    constexpr int PWN = 3 * KPB * NCC / 8;       // ... off-diagonal (+ A hi); only BC = 128 has off-diagonal blocks
    static_assert(BC == 128 || BC == 64, "block side");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* const lds = reinterpret_cast<char*>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WGN, wn = wid % WGN;
    const int li = lane & 15, g = lane >> 4;

    // workgroup -> (pixel range s, block b): the nb^2 blocks of one pixel range sit on ONE XCD (workgroups L and L + 8
    // share an XCD), so they share the range's bricks through its L2
    const int nb2 = p.nb * p.nb;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int s = xcd + 8 * (slot / nb2), b = slot % nb2;
    if (s >= p.S) return;
    const int bi = b / p.nb, bj = b % p.nb;
    const bool diag = BC == 64 || bi == bj;
    const int pb0 = s * KPB * p.steps;
    const unsigned cpb = (unsigned)p.C >> 5;     // bricks per pixel block

    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.X), 0, p.x_bytes, 0x00020000);

    // this wave's pieces of a step: q = wid + 8 j -> region q / (2 NCC) (0 B hi, 1 B lo, 2 A hi), pixel block, brick
    unsigned pc_goff[PWN], pc_dst[PWN];
    int pc_pb[PWN];
#pragma unroll
    for (int j = 0; j < PWN; ++j) {
        const int q = wid + 8 * j;
        const int r = q / (KPB * NCC), rem = q % (KPB * NCC);
        const int pb = rem / NCC, cc = rem % NCC;
        pc_pb[j] = pb;
        pc_goff[j] = (r == 1 ? p.lo_off : 0u) + (unsigned)((r == 2 ? bi : bj) * NCC + cc) * 1024u + (unsigned)lane * 16u;
        pc_dst[j] = (unsigned)(r * REG + (pb * NCC + cc) * 1024);
    }
    const int NS = diag ? NSD : NSN;
    const int STAGE = diag ? 2 * REG : 3 * REG;
    int islot = 0;                               // stage the next request goes to
    auto issue = [&](int t) {
        char* st = lds + islot * STAGE;
        islot = islot + 1 == NS ? 0 : islot + 1;
#pragma unroll
        for (int j = 0; j < PWN; ++j) {
            if (j >= PWD && diag) break;
            const int p16 = pb0 + KPB * t + pc_pb[j];
            const unsigned goff = (t < p.steps && p16 < p.nblk16) ? pc_goff[j] + (unsigned)p16 * cpb * 1024u : OOB;
            char* dst = st + pc_dst[j];
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)dst, 16, goff, 0, 0, 0);
        }
    };

    f32x4 acc[TM][TN], accs[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        accs[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < TM; ++i) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const _Float16 one = (_Float16)1.f, half = (_Float16)0.5f;
    const h16x8 ones = {one, one, one, one, one, one, one, one};
    const h16x8 halves = {half, half, half, half, half, half, half, half};
    const bool sums = bi == 0 && wm == 0;        // the column sums: once per column block, by one wave row

    for (int t = 0; t < NS - 1; ++t) issue(t);
    int cslot = 0;                               // stage of the step being multiplied
    for (int t = 0; t < p.steps; ++t) {
        // step t has landed when at most the pieces of the NS - 2 younger steps are still in flight
        if (diag) wait_vmcnt<(NSD - 2) * PWD>();
        else wait_vmcnt<(NSN - 2) * PWN>();
        __builtin_amdgcn_s_barrier();            // everyone's pieces of step t landed; everyone left step t - 1
        issue(t + NS - 1);                       // -> the stage step t - 1 used
        const char* st = lds + cslot * STAGE;
        cslot = cslot + 1 == NS ? 0 : cslot + 1;
        const char* sta = diag ? st : st + 2 * REG;
#pragma unroll
        for (int h = 0; h < KPB / 2; ++h) {
            h16x8 ah[TM], bh[TN], bl2[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) ah[i] = gram_frag<NCC>(sta, wm * TM + i, h, lane);
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                bh[j] = gram_frag<NCC>(st, wn * TN + j, h, lane);
                bl2[j] = gram_frag<NCC>(st + REG, wn * TN + j, h, lane) * (_Float16)2.f;
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl2[j], ah[i], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[j], ah[i], acc[i][j], 0, 0, 0);
            if (sums) {
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    accs[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl2[j], halves, accs[j], 0, 0, 0);
                    accs[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[j], ones, accs[j], 0, 0, 0);
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the zero fills requested past the range: nothing in flight at exit
    // lane (li, g) of acc[i][j] holds V[c1 = 16 (wm TM + i) + li][c2 = 16 (wn TN + j) + 4 g .. + 3]
    float* const out = p.part + ((long)s * nb2 + b) * (BC * BC);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int c1 = (wm * TM + i) * 16 + li, c2 = (wn * TN + j) * 16 + 4 * g;
            *reinterpret_cast<f32x4*>(out + c1 * BC + c2) = acc[i][j];
        }
    if (sums && li == 0) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
            *reinterpret_cast<f32x4*>(p.sxpart + (long)s * p.C + bj * BC + (wn * TN + j) * 16 + 4 * g) = accs[j];
    }
}

// V[c1][c2] = 16 sum_s part[s][block][c1][c2] (the planes carry v / 4), sx[c] = 4 sum_s sxpart[s][c]; double sums in
// split order.  256 threads = 8 float4 elements x 32 split groups (the 64-channel layers have 512 ranges and only 1024
// float4 elements: the parallelism has to come from the ranges); blocks past the matrix take the column sums.
__global__ __launch_bounds__(256) void gram_reduce_kernel(const float* part, const float* sxpart, int S, int nb, int BC, int C,
                                                          float* V, float* sx) {
    constexpr int EPB = 8, SG = 32;
    __shared__ double red[SG][EPB][4];
    const int e = threadIdx.x & (EPB - 1), sg = threadIdx.x / EPB;
    const int nmat = C * C / 4 / EPB;            // blocks that own matrix elements
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    long dst = -1;
    float scale_back = 0.f;
    if ((int)blockIdx.x < nmat) {
        const int idx = blockIdx.x * EPB + e;                    // float4 index in [block][c1][c2 / 4] order
        const int per = BC * BC / 4;
        const int blk = idx / per, w = idx - blk * per;
        const int c1 = w / (BC / 4), c24 = w - c1 * (BC / 4);
        const long stride = (long)nb * nb * BC * BC;
        const float* src = part + (long)blk * BC * BC + (long)c1 * BC + c24 * 4;
        for (int s = sg; s < S; s += SG) {
            const float4 v = *reinterpret_cast<const float4*>(src + s * stride);
            a0 += v.x; a1 += v.y; a2 += v.z; a3 += v.w;
        }
        dst = (long)((blk / nb) * BC + c1) * C + (blk % nb) * BC + c24 * 4;
        scale_back = 1.f / (SPLIT3_ASCALE * SPLIT3_ASCALE);
    } else {
        const int c4 = ((int)blockIdx.x - nmat) * EPB + e;       // float4 index of the column sums
        if (c4 * 4 < C) {
            for (int s = sg; s < S; s += SG) {
                const float4 v = *reinterpret_cast<const float4*>(sxpart + (long)s * C + c4 * 4);
                a0 += v.x; a1 += v.y; a2 += v.z; a3 += v.w;
            }
            dst = c4 * 4;
        }
        scale_back = 1.f / SPLIT3_ASCALE;
    }
    red[sg][e][0] = a0; red[sg][e][1] = a1; red[sg][e][2] = a2; red[sg][e][3] = a3;
    __syncthreads();
    if (sg != 0 || dst < 0) return;
    double t[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int k = 0; k < SG; ++k)
#pragma unroll
        for (int c = 0; c < 4; ++c) t[c] += red[k][e][c];
    float* o = ((int)blockIdx.x < nmat ? V : sx) + dst;
    *reinterpret_cast<float4*>(o) = make_float4((float)(t[0] * scale_back), (float)(t[1] * scale_back),
                                                (float)(t[2] * scale_back), (float)(t[3] * scale_back));
}

// 16 output channels per workgroup: Y = V W_tile by exact-f32 MFMA, the two dot products in double, then
// acimg_bn_finalize's arithmetic.  One wave per 16-row tile of V (C / 16 waves, at most 16: C = 512 takes two tiles per
// wave); the K loop is unrolled per channel count so that a tile's C / 16 row loads are all in flight before its first
// MFMA (the first version walked them one by one: 26 us at C = 256).  LDS: the weight tile transposed, Wt[n][c] with rows
// padded by 4 floats (the 16 rows of a fragment read 16-byte pieces that land on distinct bank groups).
template <int C>
__global__ __launch_bounds__(C >= 256 ? 1024 : C * 4) void gram_quadfin_kernel(const float* V, const float* sx, const float* W,
                                                                               int ldw, int K, double count, const float* gamma,
                                                                               const float* beta, float* moving_mean,
                                                                               float* moving_var, float decay, float eps,
                                                                               float* scale, float* shift) {
    constexpr int NW = C >= 256 ? 16 : C / 16, NTHR = NW * 64;
    constexpr int ldt = C + 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const Wt = smem;                                              // [16][C + 4]
    double* const red = reinterpret_cast<double*>(smem + 16 * ldt);      // [NW * 4][16] (s2), then [NTHR / 16][16] (s1)
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int li = lane & 15, g = lane >> 4;
    const int n0 = blockIdx.x * 16;
    for (int idx = tid; idx < C * 16; idx += NTHR) {
        const int c = idx >> 4, n = idx & 15;
        Wt[n * ldt + c] = n0 + n < K ? W[(long)c * ldw + n0 + n] : 0.f;
    }
    __syncthreads();
    double part = 0.0;
#pragma unroll
    for (int t = wid; t < C / 16; t += NW) {
        const float* vrow = V + (long)(t * 16 + li) * C + 4 * g;
        const float* wrow = Wt + li * ldt + 4 * g;
        f32x4 a[C / 16];
#pragma unroll
        for (int k = 0; k < C / 16; ++k) a[k] = *reinterpret_cast<const f32x4*>(vrow + 16 * k);
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < C / 16; ++k) {
            const f32x4 bq = *reinterpret_cast<const f32x4*>(wrow + 16 * k);
#pragma unroll
            for (int j = 0; j < 4; ++j) {      // two accumulation chains: an MFMA does not wait for its predecessor
                if (k & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[k][j], bq[j], acc1, 0, 0, 0);
                else acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[k][j], bq[j], acc0, 0, 0, 0);
            }
        }
        // acc[r] = Y[c1 = 16 t + 4 g + r][n0 + li]
        const f32x4 wv = *reinterpret_cast<const f32x4*>(Wt + li * ldt + t * 16 + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r) part += ((double)acc0[r] + (double)acc1[r]) * (double)wv[r];
    }
    red[(wid * 4 + g) * 16 + li] = part;
    {
        double s1p = 0.0;
        const int n = tid & 15, pr = tid >> 4;
        for (int c = pr; c < C; c += NTHR / 16) s1p += (double)Wt[n * ldt + c] * (double)sx[c];
        red[NW * 64 + pr * 16 + n] = s1p;
    }
    __syncthreads();
    if (tid >= 16 || n0 + tid >= K) return;
    double s2 = 0.0, s1 = 0.0;
    for (int k = 0; k < NW * 4; ++k) s2 += red[k * 16 + tid];
    for (int k = 0; k < NTHR / 16; ++k) s1 += red[NW * 64 + k * 16 + tid];
    const int c = n0 + tid;
    const double m = s1 / count;
    double v = s2 / count - m * m;
    if (v < 0.0) v = 0.0;
    const float mean = (float)m, var = (float)v;
    if (moving_mean) {
        const double unbiased = count > 1.0 ? v * (count / (count - 1.0)) : v;
        moving_mean[c] = decay * moving_mean[c] + (1.f - decay) * mean;
        moving_var[c] = decay * moving_var[c] + (1.f - decay) * (float)unbiased;
    }
    const float invstd = 1.f / sqrtf(var + eps);
    const float sc = (gamma ? gamma[c] : 1.f) * invstd;
    scale[c] = sc;
    shift[c] = (beta ? beta[c] : 0.f) - mean * sc;
}

struct GramPlan {
    int BC, nb, S, steps, nblk16;
    size_t off_sx, off_V, off_sxr, bytes;
};
static GramPlan gram_plan(long rows, int C) {
    GramPlan g{};
    g.BC = C % 128 == 0 ? 128 : 64;
    g.nb = C / g.BC;
    g.nblk16 = (int)((rows + 15) / 16);
    // about one workgroup per CU for the 128-blocks (96 KiB of LDS each), two for the 64-block; S a multiple of 8 (the XCD
    // mapping), at least two K steps per range
    int target = g.BC == 128 ? 256 : 512;
    int S = target / (g.nb * g.nb);
    const int max_s = (g.nblk16 + 7) / 8;
    if (S > max_s) S = max_s;
    S = (S + 7) / 8 * 8;
    if (S < 8) S = 8;
    g.S = S;
    g.steps = (g.nblk16 + 4 * S - 1) / (4 * S);
    const size_t part = (size_t)S * g.nb * g.nb * g.BC * g.BC * 4;
    g.off_sx = part;
    g.off_V = g.off_sx + (size_t)S * C * 4;
    g.off_sxr = g.off_V + (size_t)C * C * 4;
    g.bytes = g.off_sxr + (size_t)C * 4;
    return g;
}

}  // namespace acimg

using namespace acimg;

extern "C" {

size_t acimg_gram_stats_workspace(long rows, int C) {
    if (rows <= 0 || C < 64 || (C != 64 && C % 128)) return 0;
    return gram_plan(rows, C).bytes;
}

int acimg_gram_stats(const void* x_planes, size_t x_lo_off, long rows, int C, const float* w, int ldw, int K,
                     const float* gamma, const float* beta, float* moving_mean, float* moving_var, float decay, float eps,
                     float* scale, float* shift, void* ws, size_t ws_bytes, void* stream) {
    if (!x_planes || !w || !scale || !shift || !ws || rows <= 0 || K <= 0 || ldw < K)
        return fail(ACIMG_EINVAL, "gram_stats: null argument or bad shape");
    if (C != 64 && (C % 128 || C > 512)) return fail(ACIMG_EINVAL, "gram_stats: C must be 64 or a multiple of 128 up to 512");
    const size_t plane = acimg_split_plane_bytes(rows, C);
    if (!aligned16(x_planes) || (x_lo_off & 15) || x_lo_off < plane || x_lo_off + plane >= (size_t)1 << 31)
        return fail(ACIMG_EINVAL, "gram_stats: unaligned / overlapping / >= 2 GiB split-format input");
    if (!aligned16(ws) || !aligned16(w) || (ldw & 3)) return fail(ACIMG_EINVAL, "gram_stats: unaligned workspace / weights");
    const GramPlan g = gram_plan(rows, C);
    if (ws_bytes < g.bytes) return fail(ACIMG_EINVAL, "gram_stats: workspace too small (%zu < %zu)", ws_bytes, g.bytes);
    char* base = static_cast<char*>(ws);
    GramParams p{};
    p.X = static_cast<const char*>(x_planes);
    p.x_bytes = (unsigned)(x_lo_off + plane);
    p.lo_off = (unsigned)x_lo_off;
    p.nblk16 = g.nblk16;
    p.C = C;
    p.nb = g.nb;
    p.S = g.S;
    p.steps = g.steps;
    p.part = reinterpret_cast<float*>(base);
    p.sxpart = reinterpret_cast<float*>(base + g.off_sx);
    float* V = reinterpret_cast<float*>(base + g.off_V);
    float* sxr = reinterpret_cast<float*>(base + g.off_sxr);
    hipStream_t st = (hipStream_t)stream;
    const int grid = g.S * g.nb * g.nb;      // S is a multiple of 8: (xcd, slot) covers every (range, block) exactly once
    if (g.BC == 128) {
        constexpr int lds = 9 * 4 * 4 * 1024;        // 9 regions (3 stages x 3, or 4 x 2) of 4 pixel blocks x 4 bricks
        static bool attr = false;
        if (!attr) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gram_partial_kernel<128>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            attr = true;
        }
        hipLaunchKernelGGL(gram_partial_kernel<128>, dim3(grid), dim3(512), lds, st, p);
    } else {
        constexpr int lds = 6 * 4 * 2 * 1024;        // 3 stages x 2 regions of 4 pixel blocks x 2 bricks
        hipLaunchKernelGGL(gram_partial_kernel<64>, dim3(grid), dim3(512), lds, st, p);
    }
    int rc = check_launch("gram_stats (partials)");
    if (rc) return rc;
    hipLaunchKernelGGL(gram_reduce_kernel, dim3(C * C / 32 + cdiv(C, 32)), dim3(256), 0, st, p.part, p.sxpart, g.S, g.nb, g.BC, C,
                       V, sxr);
    rc = check_launch("gram_stats (reduce)");
    if (rc) return rc;
    const int nw = C >= 256 ? 16 : C / 16;
    const int lds3 = 16 * (C + 4) * 4 + (nw * 64 + nw * 4 * 16) * 8;
#define ACIMG_QUADFIN(CC)                                                                                                  \
    hipLaunchKernelGGL(gram_quadfin_kernel<CC>, dim3(cdiv(K, 16)), dim3(nw * 64), lds3, st, V, sxr, w, ldw, K, (double)rows, \
                       gamma, beta, moving_mean, moving_var, decay, eps, scale, shift)
    if (C == 64) ACIMG_QUADFIN(64);
    else if (C == 128) ACIMG_QUADFIN(128);
    else if (C == 256) ACIMG_QUADFIN(256);
    else if (C == 384) ACIMG_QUADFIN(384);
    else ACIMG_QUADFIN(512);
#undef ACIMG_QUADFIN
    return check_launch("gram_stats (quadratic forms)");
}

}  // extern "C"
