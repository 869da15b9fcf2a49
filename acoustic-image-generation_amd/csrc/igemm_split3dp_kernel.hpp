// PERSISTENT form of the trunk kernel (igemm_split3d_kernel.hpp): same 128x128x32 tile, same LDS image, same MFMA
// schedule and the same bits out, but a workgroup walks a list of tiles instead of owning one, so that the parts of a
// tile that are not matrix work overlap the next tile instead of serialising with it.
//
// Why: the trunk's 1x1 layers have 2 ... 16 K steps per tile.  With one tile per workgroup, all ~512 resident
// workgroups load, multiply and store in lockstep: per round the chip sees operand-latency (~1.5 us), then MFMA work
// (~6 us for K = 256), then 33 MB of output stores (~6 us) one after the other (profiles/r01: 18 us per round for
// 28x38 256->1024, against 6 us of MFMA and 7 us of HBM time).  Here
//   * the FIRST operand stage of tile t+1 is requested (LDS-DMA) at the top of the LAST K step of tile t, into the
//     stage buffer that step t's second-to-last tile just left, so it lands while tile t multiplies and stores;
//   * the addresses of tile t+1 (integer divisions) are computed while tile t's loads are in flight;
//   * the output tile leaves through the ONE stage buffer the last K step used (two half-tiles of 32 KiB) while the
//     other stage already holds tile t+1; output stores are fire-and-forget buffer stores with out-of-range offsets
//     for tail rows (a fixed number of vector-memory instructions per wave), and the first K step of tile t+1 waits
//     with a COUNTED vmcnt that covers its DMA pieces but not the younger stores — the stores drain under the MFMAs;
//   * no workgroup launch / exit per tile.
// Tiles of the last partial round are still cut into K ranges (tail split, unchanged protocol) and run after a
// workgroup's whole tiles, without the overlap.
//
// LDS (68 KiB, 2 workgroups / CU): 2 stages x [A hi | A lo | B hi | B lo] x 8 KiB, then 4 KiB of statistics scratch.
#pragma once
#include "igemm_split3d_kernel.hpp"

namespace acimg {

struct TileAddrP {
    int mt, nt;
    int a_off, a_ih0, a_iw0;   // this lane's row of the A piece (one 16-row block per wave)
    unsigned b_goff;           // this lane's row of the B piece, byte offset at k = 0 (or OOB)
};

struct KCursorP {
    int r, s, c0, q;           // tap row, tap column, first channel of the chunk, linear K step
};
__device__ __forceinline__ KCursorP kcursor_at(int step, int C, int S) {
    KCursorP k;
    k.q = step;
    const int cpk = C / 32;
    const int tap = step / cpk;
    k.c0 = (step - tap * cpk) * 32;
    k.r = tap / S;
    k.s = tap - k.r * S;
    return k;
}
__device__ __forceinline__ KCursorP kcursor_next(KCursorP k, int C, int S) {
    KCursorP n;
    n.q = k.q + 1;
    const bool wrap_c = k.c0 + 32 == C;
    n.c0 = wrap_c ? 0 : k.c0 + 32;
    const bool wrap_s = wrap_c && k.s + 1 == S;
    n.s = wrap_c ? (wrap_s ? 0 : k.s + 1) : k.s;
    n.r = wrap_s ? k.r + 1 : k.r;
    return n;
}

__global__ __launch_bounds__(512, 4) void igemm_split3dp_kernel(const IgemmParams p, const int n_units, const int stride_units) {
    constexpr int BM = 128, BN = 128, WGN = 4, NTHR = 512, BK = 32, ROWB = BK * 2;
    constexpr int PLANE = BM * ROWB;                 // 8 KiB: one fp16 plane of one operand of one stage
    constexpr int STAGE = 4 * PLANE;                 // [A hi | A lo | B hi | B lo]
    constexpr int WTM = 64, WTN = 32, TM = 4, TN = 2;
    constexpr int CH = BN / 4;                       // 16-byte chunks per output row
    constexpr int NST = 2 * (BM / 2 * CH / NTHR) + 1;   // vector-memory stores a wave issues per tile epilogue (8 + 1)
    static_assert(BM / 2 * CH % NTHR == 0, "half-tile store mapping");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* const lds = reinterpret_cast<char*>(smem);
    float* const red = smem + 2 * STAGE / 4;         // [4][2][BN] statistics partials

    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WGN, wn = wid % WGN;
    const int li = lane & 15, g = lane >> 4;
    const int prow = lane >> 2, pch = lane & 3;
    const int prow_t = wid * 16 + prow;              // tile row this lane fetches (A and B pieces alike)
    const int kc_sw = pch ^ swz(prow_t);
    const int Ktot = p.ntaps * p.C;
    const int ohw = p.OH * p.OW;
    const unsigned b_lo_off = (unsigned)((long)p.Nld * Ktot * 2);

    const __amdgpu_buffer_rsrc_t rsA =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.A), 0, p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.B), 0, p.b_bytes, 0x00020000);
    const EpiParams& e = p.e;

    auto setup = [&](int vt) -> TileAddrP {
        TileAddrP t;
        int mt_, nt_;
        raster_tile(p, vt, mt_, nt_);
        t.mt = mt_;
        t.nt = nt_;
        const int m = t.mt * BM + prow_t;
        if (m < p.M) {
            const int img = m / ohw;
            const int r2 = m - img * ohw;
            const int oh = r2 / p.OW;
            const int ow = r2 - oh * p.OW;
            t.a_ih0 = oh * p.stride - p.pad_t;
            t.a_iw0 = ow * p.stride - p.pad_l;
            t.a_off = ((img * p.H + t.a_ih0) * p.W + t.a_iw0) * p.lda * 2 + kc_sw * 16;
        } else {
            t.a_ih0 = -(1 << 28);
            t.a_iw0 = -(1 << 28);
            t.a_off = 0;
        }
        const int n = t.nt * BN + prow_t;
        t.b_goff = n < p.Nld ? (unsigned)(((long)n * Ktot + kc_sw * 8) * 2) : OOB;
        return t;
    };

    // K cursor of the tile being requested: tap (row, column), channel chunk, linear step
    KCursorP kc{0, 0, 0, 0};
    // request K step `kc.q` of tile t into stage `slot` (B pieces before A pieces, 4 DMA instructions per wave)
    auto issue = [&](const TileAddrP& t, int slot) {
        char* st = lds + slot * STAGE + wid * 1024;
        const unsigned kbyte = (unsigned)(kc.q * (BK * 2));
        const unsigned boff = t.b_goff == OOB ? OOB : t.b_goff + kbyte;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr_t)(st + 2 * PLANE), 16, boff, 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr_t)(st + 3 * PLANE), 16, boff, (int)b_lo_off, 0, 0);
        const int tapoff = ((kc.r * p.W + kc.s) * p.lda + kc.c0) * 2;
        const int ih = t.a_ih0 + kc.r, iw = t.a_iw0 + kc.s;
        const bool ok = (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
        const unsigned aoff = ok ? (unsigned)(t.a_off + tapoff) : OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr_t)st, 16, aoff, 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr_t)(st + PLANE), 16, aoff, (int)p.a_lo_off, 0, 0);
        kc = kcursor_next(kc, p.C, p.S);
    };
    auto cursor_set = [&](int step) { kc = kcursor_at(step, p.C, p.S); };

    f32x4 acc[TM][TN];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    zero_acc();

    auto compute = [&](int slot) {
        const char* sta = lds + slot * STAGE;
        const char* stb = sta + 2 * PLANE;
        h16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int row = wm * WTM + i * 16 + li;
            const int off = row * ROWB + ((g ^ swz(row)) << 4);
            ah[i] = *reinterpret_cast<const h16x8*>(sta + off);
            al[i] = *reinterpret_cast<const h16x8*>(sta + PLANE + off);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int row = wn * WTN + j * 16 + li;
            const int off = row * ROWB + ((g ^ swz(row)) << 4);
            bh[j] = *reinterpret_cast<const h16x8*>(stb + off);
            bl[j] = *reinterpret_cast<const h16x8*>(stb + PLANE + off);
        }
        __builtin_amdgcn_sched_barrier(0);
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
    };

    // Output tile through ONE stage buffer (32 KiB = 64 rows x 128 fp32), two halves: half h holds the accumulator
    // rows i = 2h, 2h+1 of both wave rows, i.e. tile row (lr / 32) * 64 + h * 32 + lr % 32 at local row lr.  Image:
    // 16-byte chunk c of local row lr at chunk c ^ (lr & 31): accumulator-shaped writes and row-shaped reads are both
    // conflict free.  Exactly NST vector-memory instructions per wave (out-of-range offsets instead of branches).
    auto epilogue = [&](const int mt, const int nt, int slot) {
        const __amdgpu_buffer_rsrc_t rsY =
            __builtin_amdgcn_make_buffer_rsrc(e.Y, 0, (unsigned)((long)e.M * e.ldy * 4), 0x00020000);
        const __amdgpu_buffer_rsrc_t rsS = __builtin_amdgcn_make_buffer_rsrc(
            e.stats, 0, e.stats ? (unsigned)((long)p.ras_tiles_m * 2 * e.stats_ld * 4) : 0u, 0x00020000);
        f32x4* tile = reinterpret_cast<f32x4*>(lds + slot * STAGE);
        const int m0 = mt * BM, n0 = nt * BN;
        // every index below is re-derived from an opaque copy of the thread id: otherwise the compiler hoists this
        // block's ~30 loop-invariant addresses out of the tile loop and spills inside the K loop
        int te = tid;
        asm volatile("" : "+v"(te));
        const int e_lane = te & 63, e_wid = te >> 6;
        const int e_wm = e_wid / WGN, e_wn = e_wid % WGN, e_li = e_lane & 15, e_g = e_lane >> 4;
        const int col = te % BN, part = te / BN;        // statistics: 4 row groups per column
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int ii = 0; ii < 2; ++ii) {
                const int lr = e_wm * 32 + ii * 16 + e_li;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int c = (e_wn * WTN + j * 16) / 4 + e_g;
                    tile[lr * CH + (c ^ (lr & (CH - 1)))] = acc[2 * h + ii][j] * SPLIT3_OUTSCALE;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
#pragma unroll
            for (int k = 0; k < BM / 2 * CH / NTHR; ++k) {
                const int tt = te + k * NTHR;
                const int lr = tt / CH, c = tt - lr * CH;
                const int m = m0 + (lr >> 5) * 64 + h * 32 + (lr & 31), n = n0 + 4 * c;
                const f32x4 v = tile[lr * CH + (c ^ (lr & (CH - 1)))];
                const unsigned off = (m < e.M && n < e.Nstore) ? (unsigned)(((long)m * e.ldy + n) * 4) : OOB;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsY, off, 0, 0);
            }
            if (e.stats) {
                // rows past M hold exact zeros (their A pieces were out of range): they add nothing
                const float* tf = reinterpret_cast<const float*>(tile);
#pragma unroll 8
                for (int r = part * 16; r < part * 16 + 16; ++r) {
                    const float v = tf[(r * CH + ((col >> 2) ^ (r & (CH - 1)))) * 4 + (col & 3)];
                    s1 += v;
                    s2 += v * v;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
        float sum = 0.f;
        const int which = part & 1;
        if (e.stats) {
            red[(part * 2 + 0) * BN + col] = s1;
            red[(part * 2 + 1) * BN + col] = s2;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
#pragma unroll
            for (int w = 0; w < 4; ++w) sum += red[(w * 2 + which) * BN + col];
        }
        {
            const int n = n0 + col;
            const unsigned off = (e.stats && te < 2 * BN && n < e.stats_ld)
                                     ? (unsigned)((((long)mt * 2 + which) * e.stats_ld + n) * 4) : OOB;
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, sum), rsS, off, 0, 0);
        }
    };

    // ================= whole tiles: units [0, ts_whole), this workgroup takes blockIdx.x + k * stride ============
    const int n_whole = p.ts_s > 1 ? p.ts_whole : n_units;
    int unit = blockIdx.x;
    int slot = 0;
    if (unit < n_whole) {
        TileAddrP cur = setup(unit);
        kc = KCursorP{0, 0, 0, 0};
        issue(cur, 0);
        bool pend = false;              // epilogue stores of the previous tile are younger than this tile's first DMA
        for (;;) {
            const int next = unit + stride_units;
            const bool has_next = next < n_whole;
            TileAddrP nxt = cur;
            for (int it = 0; it < p.kiters; ++it) {
                // the next tile's addresses (integer divisions) are worked out right before the last K step, while
                // this wave would only be waiting for that step's pieces
                if (it + 1 == p.kiters && has_next) nxt = setup(next);
                if (it == 0 && pend) wait_vmcnt<NST>();
                else wait_vmcnt<0>();
                __builtin_amdgcn_s_barrier();       // everyone's pieces of step `it` landed; everyone left step it-1
                if (it + 1 < p.kiters) {
                    issue(cur, slot ^ 1);
                } else if (has_next) {
                    kc = KCursorP{0, 0, 0, 0};
                    issue(nxt, slot ^ 1);
                }
                __builtin_amdgcn_s_setprio(1);
                compute(slot);
                __builtin_amdgcn_s_setprio(0);
                slot ^= 1;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();           // everyone has read the last stage: it becomes the output buffer
            asm volatile("" ::: "memory");
            epilogue(cur.mt, cur.nt, slot ^ 1);
            zero_acc();
            pend = true;
            if (!has_next) break;
            cur = nxt;
            unit = next;
        }
        unit += stride_units;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    // ================= tail: K ranges of the tiles of the last partial round (no overlap) =========================
    if (p.ts_s <= 1) return;
    while (unit < n_whole) unit += stride_units;
    int* const flag = reinterpret_cast<int*>(red);
    for (; unit < n_units; unit += stride_units) {
        const int u = unit - p.ts_whole;
        const int vt = p.ts_whole + u / p.ts_s;
        const int chunk = u - (u / p.ts_s) * p.ts_s;
        const int base = p.kiters / p.ts_s, extra = p.kiters - base * p.ts_s;
        const int it_begin = chunk * base + min(chunk, extra);
        const int it_end = it_begin + base + (chunk < extra ? 1 : 0);
        const TileAddrP cur = setup(vt);
        __builtin_amdgcn_s_barrier();               // nobody still reads LDS from the previous unit
        cursor_set(it_begin);
        slot = 0;
        if (it_begin < it_end) issue(cur, 0);
        for (int it = it_begin; it < it_end; ++it) {
            wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            if (it + 1 < it_end) issue(cur, slot ^ 1);
            __builtin_amdgcn_s_setprio(1);
            compute(slot);
            __builtin_amdgcn_s_setprio(0);
            slot ^= 1;
        }
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
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
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
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const int last = *reinterpret_cast<volatile int*>(flag);
        zero_acc();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();               // everyone has read the flag before `red` is reused
        if (!last) continue;
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
        epilogue(cur.mt, cur.nt, 0);
        zero_acc();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
}

}  // namespace acimg
