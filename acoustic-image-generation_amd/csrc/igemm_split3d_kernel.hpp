// f16x3 forward convolution on PRE-SPLIT activations with LDS-DMA operand staging (gfx950): the trunk kernel.
//
// A is stored in HBM as two fp16 planes of 16-pixel x 32-channel bricks (hi at p.A, lo p.a_lo_off bytes further; values
// already carry the 2^-2 scale and any BN affine + ReLU, written by the elementwise producers in elementwise.hip), B is
// the tile-ordered weight image behind the row-major planes of `split3_prepare_kernel` (see "bricks" below).  Same GEMM core, LDS image (64-byte rows,
// chunk ^ swz(row)) and epilogue as `igemm_split3_kernel`, but neither operand tile passes through VGPRs:
// every wave copies 1-KiB pieces (16 tile rows of one fp16 plane) global -> LDS with
// `buffer_load_dwordx4 ... lds`.  That removes the ds_write_b128 pass (13 store-path cycles per
// wave-instruction), the staging registers and the s_waitcnt/VALU work between them (72 VGPRs in all).
//
//   * LDS destination of a DMA is wave-uniform base + 16*lane, so the XOR swizzle is applied on the SOURCE
//     side: lane l of a piece fills physical chunk (l&3) of tile row (l>>2) and therefore fetches logical k
//     chunk (l&3)^swz(row).
//   * im2col: per-lane buffer offsets; padding taps / M tail / N tail use an out-of-range offset, for which
//     the DMA writes zeros.
//   * rings of NSA A stages and NSB B stages (NSA = NSB or NSB + 1), ONE barrier per K step:
//         s_waitcnt vmcnt(INFLIGHT)          own pieces of tile `it` have landed
//         s_barrier                          everyone's have, and everyone is done reading tile it-1
//         request B(it+NSB-1), A(it+NSA-1)   into the slots tile it-1 just left
//         12 ds_read_b128, then 24 MFMAs back to back (term-major: no MFMA waits on its predecessor)
//   * measured (profiles/r01, tools/tune_dma.py): 2+2 stages at 2 workgroups/CU is the best point; a third A
//     stage (80 KiB, still 2/CU), 3+3 stages at 1/CU and a 256x128 tile are all equal or slower - the K step is
//     not bound by DMA latency.  In-round MFMA utilisation is ~53 %; the rest of the loss is tile quantisation.
//   * workgroup -> tile mapping is XCD-aware (raster_tile below).
#pragma once
#include "igemm_split3_kernel.hpp"

namespace acimg {

typedef __attribute__((address_space(3))) void* lds_ptr_t;

// ---- operands in LDS-TILE ORDER ("bricks") --------------------------------------------------------------------------
// A request of the LDS-DMA path is 64 lanes x 16 bytes = 16 tile rows of 64 bytes.  With row-major operands those are
// sixteen 64-byte pieces of sixteen different rows (half a cache line each, a row pitch apart), and the K loop waits at
// the CU's texture addresser for them (profiles/r02 ablation: no weight requests -29 %, no activation requests -21 % on a
// long-K layer; BK = 64 "whole lines" showed 1.7x less issue stall per byte but cost the second workgroup per CU).  Both
// operands are therefore kept in HBM the way the LDS holds them:
//   * activations: a split-format tensor [P pixels][C] (C % 32 == 0) is two planes of BRICKS; brick (pixel block
//     f >> 4, channel chunk c >> 5) is 1 KiB at ((f >> 4) * C / 32 + (c >> 5)) * 1024: 16 rows (f & 15) of 64 bytes, logical
//     16-byte chunk kc of a row at physical chunk kc ^ swz(f).  A 1x1 / stride-1 piece is one contiguous KiB (eight whole
//     128-byte lines); a shifted (3x3 tap) or strided piece is still a per-lane gather, now out of one or two bricks;
//   * weights: behind the row-major planes, for every (128-column tile n >> 7, K step q) the two 8 KiB plane images
//     [row n & 127][64 B] with the same chunk swizzle (acimg_conv2d_split3_prepare writes both forms).
__device__ __forceinline__ unsigned brick_a_off(unsigned f, unsigned c32, unsigned kc) {
    // byte offset of logical chunk kc of pixel f's row in channel chunk 0; c32 = C * 32 = bytes of one pixel block
    return (f >> 4) * c32 + ((f & 15u) << 6) + (((kc ^ (0u - (f >> 2))) & 3u) << 4);
}
__device__ __forceinline__ unsigned brick_b_off(unsigned n, unsigned ksteps, unsigned pch) {
    // byte offset (hi plane, K step 0) of physical chunk pch of weight row n; + 16384 per K step, + 8192 for the lo plane
    return (n >> 7) * ksteps * 16384u + ((n & 127u) << 6) + (pch << 4);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    static_assert(N >= 0 && N < 64, "vmcnt immediate");
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (N == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
    else if constexpr (N == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else static_assert(N == 0, "add the immediate");
}

// linear workgroup id -> (row tile, column tile), see IgemmParams::ras_*
__device__ __forceinline__ void raster_tile(const IgemmParams& p, int L, int& mt, int& nt) {
    const int T = p.ras_tiles_m * p.ras_tiles_n;
    const int xcd = L & 7, slot = L >> 3;
    const int base = T >> 3, extra = T & 7;
    const int ord = xcd * base + min(xcd, extra) + slot;
    const int band = p.ras_gm * p.ras_tiles_n;
    const int mband = ord / band;
    const int rem = ord - mband * band;
    const int gm = min(p.ras_gm, p.ras_tiles_m - mband * p.ras_gm);
    const int grp = gm * p.ras_gn;
    const int ngroup = rem / grp;
    const int r2 = rem - ngroup * grp;
    const int gn = min(p.ras_gn, p.ras_tiles_n - ngroup * p.ras_gn);
    const int mi = r2 / gn;
    mt = mband * p.ras_gm + mi;
    nt = ngroup * p.ras_gn + (r2 - mi * gn);
}

// TERMS = 3: the split product ah*wh + ah*wl + al*wh (fp32-class results).  TERMS = 1: fp16 OPERAND STORAGE - only the
// hi planes are fetched and multiplied (one MFMA sweep, half the operand bytes): the arithmetic of an fp16-storage /
// fp32-accumulate network (BASELINE configs[4]); activations and weights carry 11 significant bits.
template <int BM, int BN, int WGM, int WGN, int NTHR, int NSA, int NSB, int TERMS = 3>
__global__ __launch_bounds__(NTHR) void igemm_split3d_kernel(const IgemmParams p) {
    constexpr int BK = 32;
    constexpr int ROWB = BK * 2;
    constexpr int A_BYTES = BM * ROWB;            // one plane of one stage
    constexpr int B_BYTES = BN * ROWB;
    constexpr int B_BASE = NSA * 2 * A_BYTES;     // LDS: NSA x [A hi | A lo], then NSB x [B hi | B lo]
    constexpr int WTM = BM / WGM, WTN = BN / WGN;
    constexpr int TM = WTM / 16, TN = WTN / 16;
    constexpr int NW = NTHR / 64;
    constexpr int RA = BM / 16 / NW;              // 16-row blocks of A per wave (each: hi piece + lo piece)
    constexpr int RB = BN / 16 / NW;
    // pieces that may stay in flight at the top of a step (see the loop): whole younger tiles, plus the A
    // pieces of the extra A stage (A pieces are requested after the B pieces of the same step)
    constexpr int INFLIGHT = (NSB - 2) * 2 * (RA + RB) + (NSA - NSB) * 2 * RA;
    static_assert(WGM * WGN == NW && (BM / 16) % NW == 0 && (BN / 16) % NW == 0, "tile / wave mapping");
    static_assert(NSB >= 2 && (NSA == NSB || NSA == NSB + 1), "stage counts");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* const lds = reinterpret_cast<char*>(smem);

    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WGN, wn = wid % WGN;
    const int li = lane & 15, g = lane >> 4;
    // Tail split (p.ts_s > 1): workgroups [0, ts_whole) own whole tiles; the tiles of the last, partial round are
    // cut into ts_s K ranges each, so that round costs kiters / ts_s steps instead of idling most of the chip
    // for kiters steps.  Ranges of one tile meet in the workspace: see the end of the kernel.
    int vt = blockIdx.x, chunk = 0;
    if (p.ts_s > 1 && (int)blockIdx.x >= p.ts_whole) {
        const int u = blockIdx.x - p.ts_whole;
        vt = p.ts_whole + u / p.ts_s;
        chunk = u - (u / p.ts_s) * p.ts_s;
    }
    const bool split = p.ts_s > 1 && vt >= p.ts_whole;
    int mt, nt;
    raster_tile(p, vt, mt, nt);
    const int m0 = mt * BM, n0 = nt * BN;

    const __amdgpu_buffer_rsrc_t rsA =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.A), 0, p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.B), 0, p.b_bytes, 0x00020000);

    int it_begin = 0, it_end = p.kiters;
    if (split) {                                   // balanced K ranges: the first (kiters % s) get one step more
        const int base = p.kiters / p.ts_s, extra = p.kiters - base * p.ts_s;
        it_begin = chunk * base + min(chunk, extra);
        it_end = it_begin + base + (chunk < extra ? 1 : 0);
    }
    const int Ktot = p.ntaps * p.C;

    // ---- this lane's slot in a piece: tile row (lane>>2) of the 16-row block, physical chunk (lane&3) -------
    const int prow = lane >> 2, pch = lane & 3;
    const unsigned kc_sw = (unsigned)(pch ^ swz(prow));      // the logical k chunk this lane's LDS slot holds
    const unsigned c32 = (unsigned)p.C * 32u;
    int a_f0[RA], a_ih0[RA], a_iw0[RA];                      // pixel index / row / column of tap (0, 0)
    {
        const int ohw = p.OH * p.OW;
#pragma unroll
        for (int j = 0; j < RA; ++j) {
            const int row = (wid + j * NW) * 16 + prow;
            const int m = m0 + row;
            if (m < p.M) {
                const int img = m / ohw;
                const int r2 = m - img * ohw;
                const int oh = r2 / p.OW;
                const int ow = r2 - oh * p.OW;
                a_ih0[j] = oh * p.stride - p.pad_t;
                a_iw0[j] = ow * p.stride - p.pad_l;
                a_f0[j] = (img * p.H + a_ih0[j]) * p.W + a_iw0[j];
            } else {
                a_ih0[j] = -(1 << 28);
                a_iw0[j] = -(1 << 28);
                a_f0[j] = 0;
            }
        }
    }
    unsigned b_goff[RB];
#pragma unroll
    for (int j = 0; j < RB; ++j) {
        const int row = (wid + j * NW) * 16 + prow;
        b_goff[j] = p.b_brick + brick_b_off((unsigned)(n0 + row), (unsigned)(Ktot / BK), (unsigned)pch);
    }
    constexpr unsigned b_lo_off = 8192u;

    // request cursors: A walks (tap row, tap column, channel chunk); B's k offset is linear in the step
    int qa = it_begin, sa = 0, st_r = 0, st_s = 0, st_c0 = 0;     // next A tile to request, its LDS slot
    int qb = it_begin, sb = 0;
    if (it_begin > 0) {
        const int cpk = p.C / BK;
        const int tap = it_begin / cpk;
        st_c0 = (it_begin - tap * cpk) * BK;
        st_r = tap / p.S;
        st_s = tap - st_r * p.S;
    }

    auto issue_a = [&]() {
        char* st = lds + sa * (2 * A_BYTES);
        const int tapf = st_r * p.W + st_s;
        const unsigned cbyte = (unsigned)st_c0 * 32u;
#pragma unroll
        for (int j = 0; j < RA; ++j) {
            const int ih = a_ih0[j] + st_r, iw = a_iw0[j] + st_s;
            const bool ok = (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
            const unsigned off = ok ? brick_a_off((unsigned)(a_f0[j] + tapf), c32, kc_sw) + cbyte : OOB;
            char* dst = st + (wid + j * NW) * 1024;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr_t)dst, 16, off, 0, 0, 0);
            if (TERMS == 3)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr_t)(dst + A_BYTES), 16, off, (int)p.a_lo_off, 0, 0);
        }
        ++qa;
        sa = sa + 1 == NSA ? 0 : sa + 1;
        st_c0 += BK;
        if (st_c0 == p.C) {
            st_c0 = 0;
            if (++st_s == p.S) {
                st_s = 0;
                ++st_r;
            }
        }
    };
    auto issue_b = [&]() {
        char* st = lds + B_BASE + sb * (2 * B_BYTES);
        const unsigned kbyte = (unsigned)qb * 16384u;
#pragma unroll
        for (int j = 0; j < RB; ++j) {
            const unsigned off = b_goff[j] + kbyte;
            char* dst = st + (wid + j * NW) * 1024;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr_t)dst, 16, off, 0, 0, 0);
            if (TERMS == 3)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr_t)(dst + B_BYTES), 16, off, (int)b_lo_off, 0, 0);
        }
        ++qb;
        sb = sb + 1 == NSB ? 0 : sb + 1;
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto compute = [&](int ca, int cb) {
        const char* sta = lds + ca * (2 * A_BYTES);
        const char* stb = lds + B_BASE + cb * (2 * B_BYTES);
        h16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int row = wm * WTM + i * 16 + li;
            const int off = row * ROWB + ((g ^ swz(row)) << 4);
            ah[i] = *reinterpret_cast<const h16x8*>(sta + off);
            if (TERMS == 3) al[i] = *reinterpret_cast<const h16x8*>(sta + A_BYTES + off);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int row = wn * WTN + j * 16 + li;
            const int off = row * ROWB + ((g ^ swz(row)) << 4);
            bh[j] = *reinterpret_cast<const h16x8*>(stb + off);
            if (TERMS == 3) bl[j] = *reinterpret_cast<const h16x8*>(stb + B_BYTES + off);
        }
        // all 12 fragment reads are in flight before the first MFMA (one exposed LDS latency per step instead
        // of one per fragment pair), and the three terms of a product are issued a whole sweep apart so that no
        // MFMA waits on its predecessor's accumulator (the per-accumulator order lo*hi, hi*lo, hi*hi is kept)
        __builtin_amdgcn_sched_barrier(0);
        if (TERMS == 3) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl[j], ah[i], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[j], al[i], acc[i][j], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[j], ah[i], acc[i][j], 0, 0, 0);
    };

    // prologue = the requests the steps -(NSA-1) .. -1 would have made, in the same order (B before A)
#pragma unroll
    for (int v = -(NSA - 1); v < 0; ++v) {
        if (v + NSB - 1 >= 0 && qb < it_end) issue_b();
        if (qa < it_end) issue_a();
    }

    int ca = 0, cb = 0;                 // LDS slots of the tile being multiplied
    for (int it = it_begin; it < it_end; ++it) {
        // oldest-first the queue holds ... B(it) A(it+NSA-NSB) | B(it+1) A(..) ...: everything up to B(it) must
        // have landed, INFLIGHT younger pieces may stay in flight; near the tail fewer exist -> drain
        if (it + NSA - 1 <= it_end) wait_vmcnt<INFLIGHT>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();   // everyone's pieces of tile `it` landed; everyone left tile it-1
        if (qb < it_end) issue_b();     // -> slot of B(it-1)
        if (qa < it_end) issue_a();     // -> slot of A(it-1)
        __builtin_amdgcn_s_setprio(1);
        compute(ca, cb);
        __builtin_amdgcn_s_setprio(0);
        ca = ca + 1 == NSA ? 0 : ca + 1;
        cb = cb + 1 == NSB ? 0 : cb + 1;
    }
    __syncthreads();                    // the epilogue reuses the LDS
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] *= SPLIT3_OUTSCALE;
    if (split) {
        // every range parks its partial sums (write-through sc1 stores: visible to the other XCDs without an L2
        // write-back fence) and takes a ticket; the last arriver adds the ts_s partials IN RANGE ORDER (its own
        // included, from memory: the result does not depend on who came last) and runs the epilogue.  Nobody waits.
        const int tl = vt - p.ts_whole;
        float* const slot0 = p.ts_partial + (long)tl * p.ts_s * (BM * BN);
        const __amdgpu_buffer_rsrc_t rsP =
            __builtin_amdgcn_make_buffer_rsrc(slot0 + (long)chunk * (BM * BN), 0, BM * BN * 4, 0x00020000);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[i][j]), rsP,
                                                       ((i * TN + j) * NTHR + tid) * 16, 0, 16);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int* const flag = reinterpret_cast<int*>(smem);
        if (tid == 0) {
            const int ticket = __hip_atomic_fetch_add(p.ts_counters + tl, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = ticket == p.ts_s - 1;
            if (last) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_store(p.ts_counters + tl, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            *flag = last;
        }
        __syncthreads();
        if (!*flag) return;
        __syncthreads();                           // everyone has read the flag before the LDS is reused below
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int c = 0; c < p.ts_s; ++c) {
            const __amdgpu_buffer_rsrc_t rsQ =
                __builtin_amdgcn_make_buffer_rsrc(slot0 + (long)c * (BM * BN), 0, BM * BN * 4, 0x00020000);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)     // sc1 loads: never a stale L1 / L2 copy
                    acc[i][j] += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                               rsQ, ((i * TN + j) * NTHR + tid) * 16, 0, 16));
        }
    }
    const EpiParams& e = p.e;
    if (e.vec && !e.bias && !e.res && !e.mask && !e.scatter && e.act == ACIMG_ACT_NONE) {
        // Raw output through LDS: the accumulator layout gives every lane 4 channels of one pixel, i.e. 64-byte
        // pieces of 16 different output rows per store instruction.  Staging the tile in the (now free) stage
        // buffers and writing it back row-major turns that into full 512-byte row segments, two per wave
        // instruction (measured on the 1x1 layers with 512 outputs: 3.3 -> 5 TB/s of output traffic).
        // Tile image: [BM][BN] fp32, 16-byte chunk c of row r stored at chunk c ^ (r & (BN/4 - 1)): both the
        // accumulator-shaped writes and the row-shaped reads are bank-conflict free.
        constexpr int CH = BN / 4;
        f32x4* tile = reinterpret_cast<f32x4*>(smem);
        float* red = smem + BM * BN;          // [WGM][2][BN] statistics partials of the wave rows
        if (e.stats) {
            // batch-norm partials straight from the accumulators (rows past M hold zeros): a lane adds its TM row
            // blocks, a 16-lane DPP butterfly adds the 16 pixel rows of a fragment, the WGM wave rows meet in LDS -
            // instead of every thread re-reading a column of the staged tile (BM / PARTS dependent LDS reads)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                f32x4 s1 = acc[0][j], s2 = acc[0][j] * acc[0][j];
#pragma unroll
                for (int i = 1; i < TM; ++i) {
                    s1 += acc[i][j];
                    s2 += acc[i][j] * acc[i][j];
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    s1[k] = row16_sum(s1[k]);
                    s2[k] = row16_sum(s2[k]);
                }
                if (li == 0) {
                    const int n = wn * WTN + j * 16 + g * 4;
                    *reinterpret_cast<f32x4*>(red + (wm * 2 + 0) * BN + n) = s1;
                    *reinterpret_cast<f32x4*>(red + (wm * 2 + 1) * BN + n) = s2;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int row = wm * WTM + i * 16 + li;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int c = (wn * WTN + j * 16) / 4 + g;
                tile[row * CH + (c ^ (row & (CH - 1)))] = acc[i][j];
            }
        }
        __syncthreads();
        constexpr int NIT = BM * CH / NTHR;
        static_assert(BM * CH % NTHR == 0, "row store mapping");
#pragma unroll
        for (int t0 = 0; t0 < NIT; t0 += 4) {         // four row reads in flight before their stores
            f32x4 v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int t = tid + (t0 + k) * NTHR;
                const int row = t / CH, c = t - row * CH;
                v[k] = tile[row * CH + (c ^ (row & (CH - 1)))];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int t = tid + (t0 + k) * NTHR;
                const int row = t / CH, c = t - row * CH;
                const int m = m0 + row, n = n0 + 4 * c;
                if (m < e.M && n < e.Nstore) {
                    float* dst = e.Y + (long)m * e.ldy + n;
                    if (n + 3 < e.Nstore) *reinterpret_cast<f32x4*>(dst) = v[k];
                    else
                        for (int q = 0; q < 4 && n + q < e.Nstore; ++q) dst[q] = v[k][q];
                }
            }
        }
        if (e.stats) {
            for (int idx = tid; idx < 2 * BN; idx += NTHR) {
                const int which = idx / BN, c = idx - which * BN;
                const int n = n0 + c;
                if (n < e.stats_ld) {
                    float sum = 0.f;
#pragma unroll
                    for (int w = 0; w < WGM; ++w) sum += red[(w * 2 + which) * BN + c];
                    e.stats[((long)mt * 2 + which) * e.stats_ld + n] = sum;
                }
            }
        }
        return;
    }
    igemm_epilogue<BM, BN, WGM, WGN, NTHR, TM, TN>(p, acc, smem, m0, n0, wm, wn, li, g, tid, mt);
}

}  // namespace acimg
