// HALO form of the trunk kernel for the 3x3 / stride-1 / SAME layers (slim conv2d_same 3x3, models/resnet50.py:104-125).
//
// The other trunk kernels walk K as (tap, channel chunk) and request the 128-pixel x 32-channel activation tile once
// per tap: nine LDS-DMA fetches of (almost) the same pixels per channel chunk, 144 of the 288 KiB-pieces a workgroup
// requests per chunk.  Here K is walked as (channel chunk, tap): per chunk ONE patch of the hi / lo planes is staged -
// the flat pixel range [m0 - W - 1, m0 + 128 + W + 1) of the tile's 128 output pixels, rounded out to 16-pixel bricks
// (8 + 2 ceil((W + 1) / 16) bricks: 18 for W = 75, 14 for 38, 12 for 19) - and all nine taps are formed from it:
//   * brick storage is already the LDS image (igemm_split3d_kernel.hpp, "bricks"), so a patch piece is a straight copy
//     of one contiguous KiB: 2 x nb requests per chunk instead of 9 x 16 (36 + 144 pieces per chunk instead of 288);
//   * tap (r, s) of output pixel m is patch row m - base + (r - 1) W + (s - 1): ONE address per wave and tap (the
//     four 16-row fragments of a wave are 1 KiB apart: immediate offsets), the chunk swizzle follows the patch row;
//   * no zero border in memory: a lane whose tap falls off its image (or whose row is past M) reads a 64-byte ZERO ROW
//     kept behind each plane of the patch instead (per-lane flags top / bottom / left / right / invalid, one AND with
//     the tap's kill mask and one v_cndmask per fragment row);
//   * a tap shifts the 16 rows of a fragment by an arbitrary amount, and the XOR swizzle is conflict free for
//     ds_read_b128 only when the 16 rows start at a multiple of 16: the hardware's 16-lane groups {0-3, 12-15, 20-27}
//     pair 8 rows at k-chunk g with 8 rows at g ^ 1.  The fragment lanes therefore hold a PERMUTATION of the 16 pixels
//     (halo_pi): lanes 0-3 / 12-15 take the pixels with (pixel & 3) < 2, lanes 4-11 those with (pixel & 3) >= 2, so in
//     every lane group the four rows that share a bank quarter (row & 3) carry the same k-chunk and four consecutive
//     (row >> 2) - distinct 16-byte slots for EVERY shift.  The accumulators come out in the same permuted pixel order;
//     the epilogue undoes it when it stages the output tile (batch-norm partials are sums over the 16 lanes anyway);
//   * the patch is single buffered (2 x 18.06 KiB; with the two 16 KiB weight stages 68.1 KiB: two workgroups per CU as
//     before): in the LAST tap of a chunk every wave reads its fragments, the workgroup meets at a barrier, and the next
//     chunk's patch is requested UNDER that step's 24 MFMAs per wave; the weight tiles keep their two-stage ring.
// Accumulation order is (chunk, tap) instead of (tap, chunk): fp32-rounding-level differences from the other trunk
// kernels (same products, same 2^-22 operands); tests hold it to the split product's tolerance, not to their bits.
// Tail split, rasterisation, raw output + batch-norm partials: as igemm_split3d_kernel.
#pragma once
#include "igemm_split3d_kernel.hpp"

namespace acimg {

// pixel (of its 16-row fragment) that fragment lane li multiplies: {0,1,4,5 | 2,3,6,7,10,11,14,15 | 8,9,12,13}
__device__ __forceinline__ int halo_pi(int li) { return (int)((0xDC98FEBA76325410ull >> (4 * li)) & 15ull); }

template <int NBMAX>
__global__ __launch_bounds__(512, 4) void igemm_split3h_kernel(const IgemmParams p) {
    constexpr int BM = 128, BN = 128, WGM = 2, WGN = 4, NTHR = 512, NW = 8;
    constexpr int PATCH = NBMAX * 1024;           // brick rows of one plane of the patch
    constexpr int PLANE_A = PATCH + 64;           // ... + the zero row
    constexpr int B_BASE = 2 * PLANE_A;
    constexpr int B_BYTES = BN * 64;              // one plane of one weight stage
    constexpr int WTM = BM / WGM, WTN = BN / WGN, TM = WTM / 16, TN = WTN / 16;
    constexpr int NPA = (NBMAX + NW - 1) / NW;    // patch bricks per plane and wave
    constexpr int TAPS = 9;
    static_assert(B_BASE + 4 * B_BYTES >= BM * BN * 4 + 2 * WGM * BN * 4, "the output tile is staged in the operand buffers");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* const lds = reinterpret_cast<char*>(smem);

    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WGN, wn = wid % WGN;
    const int li = lane & 15, g = lane >> 4;
    int vt = blockIdx.x, krange = 0;
    if (p.ts_s > 1 && (int)blockIdx.x >= p.ts_whole) {
        const int u = blockIdx.x - p.ts_whole;
        vt = p.ts_whole + u / p.ts_s;
        krange = u - (u / p.ts_s) * p.ts_s;
    }
    const bool split = p.ts_s > 1 && vt >= p.ts_whole;
    int mt, nt;
    raster_tile(p, vt, mt, nt);
    const int m0 = mt * BM, n0 = nt * BN;

    const __amdgpu_buffer_rsrc_t rsA =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.A), 0, p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.B), 0, p.b_bytes, 0x00020000);

    int it_begin = 0, it_end = p.kiters;          // K steps in (chunk, tap) order: step = chunk * 9 + tap
    if (split) {
        const int base = p.kiters / p.ts_s, extra = p.kiters - base * p.ts_s;
        it_begin = krange * base + min(krange, extra);
        it_end = it_begin + base + (krange < extra ? 1 : 0);
    }
    const int cpk = p.C >> 5;                     // channel chunks; the weight image's K step of (chunk, tap) is tap * cpk + chunk
    const unsigned c32 = (unsigned)p.C * 32u;

    // ---- the patch: bricks [fb0, fb0 + nb) of the flat pixel index, one KiB per brick, plane and chunk ---------------
    const int before = (p.W + 16) >> 4;           // ceil((W + 1) / 16) bricks in front of the tile's own eight
    const int nb = before + 8 + (p.W >> 4) + 1;
    const int fb0 = (m0 >> 4) - before;
    const int nbricks = (p.M + 15) >> 4;          // stride 1, SAME: input pixels = output pixels
    unsigned a_goff[NPA];
#pragma unroll
    for (int j = 0; j < NPA; ++j) {
        const int gb = fb0 + wid + j * NW;
        a_goff[j] = (gb >= 0 && gb < nbricks) ? (unsigned)gb * c32 + (unsigned)lane * 16u : OOB;
    }
    auto issue_patch = [&](int chunk) {
        const unsigned cbyte = (unsigned)chunk * 1024u;
#pragma unroll
        for (int j = 0; j < NPA; ++j) {
            if (wid + j * NW < nb) {              // wave-uniform
                const unsigned off = a_goff[j] + cbyte;
                char* dst = lds + (wid + j * NW) * 1024;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr_t)dst, 16, off, 0, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr_t)(dst + PLANE_A), 16, off, (int)p.a_lo_off, 0, 0);
            }
        }
    };
    // ---- weight pieces: as the one-tile kernel (one 16-row piece per plane and wave) ---------------------------------
    const int prow = lane >> 2, pch = lane & 3;
    const unsigned b_goff = p.b_brick + brick_b_off((unsigned)(n0 + wid * 16 + prow), (unsigned)p.kiters, (unsigned)pch);
    auto issue_b = [&](int q, int slot) {
        const unsigned off = b_goff + (unsigned)q * 16384u;
        char* dst = lds + B_BASE + slot * (2 * B_BYTES) + wid * 1024;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr_t)dst, 16, off, 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr_t)(dst + B_BYTES), 16, off, 8192, 0, 0);
    };

    // ---- this lane's fragment rows: pixel m0 + wm 64 + i 16 + halo_pi(li), patch row rowrel0 + 16 i + tap shift ------
    const int pi = halo_pi(li);
    const int rowrel0 = before * 16 + wm * WTM + pi;
    unsigned fl = 0;                              // 5 flags per fragment row: top, bottom, left, right, invalid
    {
        const int ohw = p.OH * p.OW;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = m0 + wm * WTM + i * 16 + pi;
            unsigned f = 16u;
            if (m < p.M) {
                const int img = m / ohw;
                const int r2 = m - img * ohw;
                const int oh = r2 / p.OW;
                const int ow = r2 - oh * p.OW;
                f = (oh == 0 ? 1u : 0u) | (oh == p.OH - 1 ? 2u : 0u) | (ow == 0 ? 4u : 0u) | (ow == p.OW - 1 ? 8u : 0u);
            }
            fl |= f << (5 * i);
        }
    }
    if (tid < 8)                                  // the zero rows (never a DMA destination)
        *reinterpret_cast<uint4*>(lds + (tid >> 2) * PLANE_A + PATCH + (tid & 3) * 16) = make_uint4(0, 0, 0, 0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    int chunk = it_begin / TAPS;
    int tap = it_begin - chunk * TAPS;
    if (it_begin < it_end) {
        issue_patch(chunk);
        issue_b(tap * cpk + chunk, 0);
    }
    int sb = 0;
    for (int it = it_begin; it < it_end; ++it) {
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();             // this step's pieces landed; everyone left step it - 1
        const bool more = it + 1 < it_end;
        const bool last_tap = tap == TAPS - 1;
        const int ntap = last_tap ? 0 : tap + 1;
        const int nchunk = last_tap ? chunk + 1 : chunk;
        if (more) issue_b(ntap * cpk + nchunk, sb ^ 1);
        __builtin_amdgcn_s_setprio(1);
        {
            const int r = tap / 3, s = tap - 3 * r;
            const int d = (r - 1) * p.W + (s - 1);
            const unsigned tm = 16u | (r == 0 ? 1u : 0u) | (r == 2 ? 2u : 0u) | (s == 0 ? 4u : 0u) | (s == 2 ? 8u : 0u);
            const unsigned hit = fl & (tm * 0x8421u);
            const int p0 = rowrel0 + d;
            const int addr0 = (p0 << 6) + (((g ^ (0 - (p0 >> 2))) & 3) << 4);
            const char* stb = lds + B_BASE + sb * (2 * B_BYTES);
            h16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int a = (hit & (31u << (5 * i))) ? PATCH + g * 16 - i * 1024 : addr0;
                ah[i] = *reinterpret_cast<const h16x8*>(lds + a + i * 1024);
                al[i] = *reinterpret_cast<const h16x8*>(lds + a + i * 1024 + PLANE_A);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int row = wn * WTN + j * 16 + li;
                const int off = row * 64 + ((g ^ swz(row)) << 4);
                bh[j] = *reinterpret_cast<const h16x8*>(stb + off);
                bl[j] = *reinterpret_cast<const h16x8*>(stb + B_BYTES + off);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (last_tap && more) {
                // every wave holds its fragments: the patch is free, the next chunk's lands under this step's MFMAs
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                issue_patch(nchunk);
                __builtin_amdgcn_sched_barrier(0);
            }
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
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[j], ah[i], acc[i][j], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
        sb ^= 1;
        tap = ntap;
        chunk = nchunk;
    }
    __syncthreads();                              // the epilogue reuses the LDS
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] *= SPLIT3_OUTSCALE;
    if (split) {                                  // K ranges of a tail tile meet in the workspace (igemm_split3d_kernel)
        const int tl = vt - p.ts_whole;
        float* const slot0 = p.ts_partial + (long)tl * p.ts_s * (BM * BN);
        const __amdgpu_buffer_rsrc_t rsP =
            __builtin_amdgcn_make_buffer_rsrc(slot0 + (long)krange * (BM * BN), 0, BM * BN * 4, 0x00020000);
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
        __syncthreads();
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
                for (int j = 0; j < TN; ++j)
                    acc[i][j] += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                               rsQ, ((i * TN + j) * NTHR + tid) * 16, 0, 16));
        }
    }
    // raw output through LDS + batch-norm partials from the accumulators (igemm_split3d_kernel's raw path); the only
    // difference: accumulator lane li holds pixel halo_pi(li) of its fragment
    const EpiParams& e = p.e;
    constexpr int CH = BN / 4;
    f32x4* tile = reinterpret_cast<f32x4*>(smem);
    float* red = smem + BM * BN;
    if (e.stats) {
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
        const int row = wm * WTM + i * 16 + pi;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int c = (wn * WTN + j * 16) / 4 + g;
            tile[row * CH + (c ^ (row & (CH - 1)))] = acc[i][j];
        }
    }
    __syncthreads();
    constexpr int NIT = BM * CH / NTHR;
#pragma unroll
    for (int t0 = 0; t0 < NIT; t0 += 4) {
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
}

}  // namespace acimg
