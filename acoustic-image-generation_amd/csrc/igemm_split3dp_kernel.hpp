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
// Two K-step depths:
//   BK = 32: the one-tile kernel's image (64-byte rows, 16 rows per 1-KiB DMA piece); 68 KiB of LDS, 2 workgroups / CU.
//   BK = 64: FULL-LINE operand staging.  With 64-byte rows every DMA wave-instruction is sixteen 64-byte requests to
//            the L2 (PMC, profiles/r01/pmc/dma: TCP_TCC_READ_REQ x 64 B = the operand bytes), and 64-byte requests use
//            half of an L2 channel's 128 B/clk: at 2 x 32 KiB per K step and CU the K loop asks for ~22 TB/s of them,
//            above what the L2 can serve that way, which is why the MFMA pipe is only ~53 % busy inside full rounds.
//            Here a tile row holds 64 k-values = one whole 128-byte line per plane, a DMA piece is 8 rows x 128 B
//            (lane l: row l >> 3, physical chunk l & 7, fetching logical chunk (l & 7) ^ (row & 7): conflict-free
//            ds_read_b128 for the 16 rows of a fragment in the hardware's lane groups), a stage is 64 KiB
//            (132 KiB of LDS, 1 workgroup / CU) and a K step is 48 MFMAs per wave between barriers.
// LDS: 2 stages x [A hi | A lo | B hi | B lo], then 4 KiB of statistics scratch.
#pragma once
#include "igemm_split3d_kernel.hpp"

// Diagnostic build only (tools/build_stamp.sh, -DACIMG_STAMP): every wave accumulates the shader cycles it spends in
// each part of a K step and of the epilogue and leaves them in a debug buffer (cdna_hip_programming.md §7, in-kernel
// stamps).  The product build contains none of this.
#ifdef ACIMG_STAMP
#define ACIMG_ABLATE            // the ablation switches (IgemmParams::flip bits) come with the stamps, or alone
#define ACIMG_STAMP_DECL unsigned long long st_last = __builtin_amdgcn_s_memtime(); unsigned st_acc[14] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; const unsigned long long st_begin = st_last;
#define ACIMG_STAMP_AT(i)                                              \
    do {                                                               \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime();    \
        st_acc[i] += (unsigned)(t_ - st_last);                         \
        st_last = t_;                                                  \
    } while (0)
#else
#define ACIMG_STAMP_DECL
#define ACIMG_STAMP_AT(i)
#endif

namespace acimg {

template <int PP>
struct TileAddrP {
    int mt, nt;
    int a_f0[PP], a_ih0[PP], a_iw0[PP];    // this lane's row of each of the wave's A pieces: pixel index / row / column at tap (0, 0)
    unsigned b_goff[PP];                   // this lane's row of each B piece, byte offset at K step 0
};

struct KCursorP {
    int r, s, c0, q;           // tap row, tap column, first channel of the chunk, linear K step
};
template <int BK>
__device__ __forceinline__ KCursorP kcursor_at(int step, int C, int S) {
    KCursorP k;
    k.q = step;
    const int cpk = C / BK;
    const int tap = step / cpk;
    k.c0 = (step - tap * cpk) * BK;
    k.r = tap / S;
    k.s = tap - k.r * S;
    return k;
}
template <int BK>
__device__ __forceinline__ KCursorP kcursor_next(KCursorP k, int C, int S) {
    KCursorP n;
    n.q = k.q + 1;
    const bool wrap_c = k.c0 + BK == C;
    n.c0 = wrap_c ? 0 : k.c0 + BK;
    const bool wrap_s = wrap_c && k.s + 1 == S;
    n.s = wrap_c ? (wrap_s ? 0 : k.s + 1) : k.s;
    n.r = wrap_s ? k.r + 1 : k.r;
    return n;
}

// DPOS: where a K step's DMA requests sit.  0: right after the step barrier, in one burst (all 8 waves of the workgroup
// queue 4 KiB each at the CU's texture addresser at the same moment: the issuing waves stall ~200 cycles per request,
// 30 % of a long-K layer's wave time - tools/stamp_probe.py - before any of them has an MFMA ready).  1: spread
// under the MFMA block - the B pieces after the first of the three MFMA sweeps, the A pieces after the second - so a
// wave that waits at the addresser has already queued matrix work and the others keep the pipe busy.
// TERMS: 3 = split product, 1 = fp16 operand storage (hi planes only), as in igemm_split3d_kernel.
// EPI: what leaves a tile.  0: the raw fp32 tile + batch-norm partials (a BN pass reads it back).  The expanding 1x1
// conv of an identity bottleneck unit (conv3: short K, 4x wide output) can instead run TWICE (slim bottleneck,
// models/resnet50.py:104-125: out = relu(BN(conv3) + shortcut)):
//   1: statistics only - the K loop and the partials, no output at all (its 4 B / element never reach HBM);
//   2: the same tiles again, now with the layer's (scale, shift) known: relu(acc * scale + shift + shortcut), split into
//      the hi / lo fp16 planes and written straight in brick order (the next unit's operand format) - the shortcut comes
//      from the previous unit's planes.  Per wide element 4 B read + 4 B written instead of 4 written + (4 + 4 read,
//      4 written) by conv + bn_add_relu_split; the price is the conv's K loop twice.
//      Both passes run the same units in the same order: the statistics are those of exactly the values normalised.
//   3: as 2 for a unit with a PROJECTION shortcut: the shortcut is the raw fp32 output of the unit's 1x1 shortcut conv
//      with a batch norm of its own: relu(acc * scale + shift + (sc32 * scale2 + shift2)).
template <int BK, int DPOS, int TERMS = 3, int EPI = 0>
__global__ __launch_bounds__(512, BK == 32 ? 4 : 2) void igemm_split3dp_kernel(const IgemmParams p, const int n_units,
                                                                               const int stride_units) {
    constexpr int BM = 128, BN = 128, WGN = 4, NTHR = 512, NW = 8, ROWB = BK * 2;
    constexpr int PLANE = BM * ROWB;                 // one fp16 plane of one operand of one stage (8 / 16 KiB)
    constexpr int STAGE = 4 * PLANE;                 // [A hi | A lo | B hi | B lo]
    constexpr int WTM = 64, WTN = 32, TM = 4, TN = 2;
    constexpr int CH = BN / 4;                       // 16-byte chunks per output row
    constexpr int CPR = ROWB / 16;                   // 16-byte chunks per operand row (4 / 8)
    constexpr int RPI = 64 / CPR;                    // operand rows per DMA wave-instruction (16 / 8)
    constexpr int PP = BM / RPI / NW;                // pieces per plane and wave (1 / 2)
    constexpr int KH = BK / 32;                      // 32-deep MFMA sweeps per K step
    constexpr int HALVES = BM * BN * 4 / STAGE;      // the output tile leaves through one stage: in 2 / 1 passes
    constexpr int RH = BM / HALVES;                  // tile rows per pass
    constexpr int WH = WTM / HALVES;                 // ... of which per wave row
    // vector-memory instructions a wave issues per tile epilogue AFTER its last wait (8 + 1; statistics only: 1)
    constexpr int NST = EPI == 1 ? 1 : HALVES * (RH * CH / NTHR) + 1;
    static_assert(RH * CH % NTHR == 0 && (EPI == 1 || NST == 9), "output store mapping");
    static_assert(EPI == 0 || (BK == 32 && TERMS == 3), "the fused epilogues belong to the split-product BK = 32 form");
    typedef TileAddrP<PP> Tile;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* const lds = reinterpret_cast<char*>(smem);
    float* const red = smem + 2 * STAGE / 4;         // [4][2][BN] statistics partials

    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WGN, wn = wid % WGN;
    const int li = lane & 15, g = lane >> 4;
    const int prow = lane / CPR, pch = lane % CPR;  // this lane's row / physical chunk inside a DMA piece
    // logical k chunk this lane fetches: the swizzle of its tile row, which depends on prow only (pieces start at
    // multiples of 16 / 8 rows)
    const int kc_sw = pch ^ (BK == 32 ? swz(prow) : (prow & 7));
    const int Ktot = p.ntaps * p.C;
    const int ohw = p.OH * p.OW;
    // both operands come in LDS-tile order (igemm_split3d_kernel.hpp, "bricks"): a weight piece (nt, q, plane, wave) is one
    // contiguous KiB, an activation piece one (1x1) or parts of two (shifted taps) KiB bricks
    static_assert(BK == 32, "the brick layouts are the BK = 32 LDS image");
    constexpr unsigned b_lo_off = 8192u, b_kstep = 16384u;
    const unsigned c32 = (unsigned)p.C * 32u;

    const __amdgpu_buffer_rsrc_t rsA =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.A), 0, p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.B), 0, p.b_bytes, 0x00020000);
    const EpiParams& e = p.e;

    auto setup = [&](int vt) -> Tile {
        Tile t;
        int mt_, nt_;
        raster_tile(p, vt, mt_, nt_);
        t.mt = mt_;
        t.nt = nt_;
#pragma unroll
        for (int j = 0; j < PP; ++j) {
            const int row = (wid + j * NW) * RPI + prow;
            const int m = t.mt * BM + row;
            if (m < p.M) {
                const int img = m / ohw;
                const int r2 = m - img * ohw;
                const int oh = r2 / p.OW;
                const int ow = r2 - oh * p.OW;
                t.a_ih0[j] = oh * p.stride - p.pad_t;
                t.a_iw0[j] = ow * p.stride - p.pad_l;
                t.a_f0[j] = (img * p.H + t.a_ih0[j]) * p.W + t.a_iw0[j];
            } else {
                t.a_ih0[j] = -(1 << 28);
                t.a_iw0[j] = -(1 << 28);
                t.a_f0[j] = 0;
            }
            t.b_goff[j] = p.b_brick + brick_b_off((unsigned)(t.nt * BN + row), (unsigned)(Ktot / BK), (unsigned)pch);
        }
        return t;
    };

    // K cursor of the tile being requested: tap (row, column), channel chunk, linear step
    KCursorP kc{0, 0, 0, 0};
    // request K step `kc.q` of tile t into stage `slot` (B pieces before A pieces, 4 DMA instructions per wave)
    auto issue_b = [&](const Tile& t, int slot) {
#ifdef ACIMG_ABLATE
        if (p.flip & 4) return;                    // ablation: no weight-tile requests
#endif
        char* st = lds + slot * STAGE;
        const unsigned kbyte = (unsigned)kc.q * b_kstep;
#pragma unroll
        for (int j = 0; j < PP; ++j) {
            const unsigned boff = t.b_goff[j] + kbyte;
            char* dst = st + 2 * PLANE + (wid + j * NW) * 1024;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr_t)dst, 16, boff, 0, 0, 0);
            if (TERMS == 3)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr_t)(dst + PLANE), 16, boff, (int)b_lo_off, 0, 0);
        }
    };
    auto issue_a = [&](const Tile& t, int slot) {      // ... and moves the cursor on
        char* st = lds + slot * STAGE;
        const int tapf = kc.r * p.W + kc.s;
        const unsigned cbyte = (unsigned)kc.c0 * 32u;
#ifdef ACIMG_ABLATE
        if (p.flip & 2) {                          // ablation: no activation-tile requests
            kc = kcursor_next<BK>(kc, p.C, p.S);
            return;
        }
#endif
#pragma unroll
        for (int j = 0; j < PP; ++j) {
            const int ih = t.a_ih0[j] + kc.r, iw = t.a_iw0[j] + kc.s;
            const bool ok = (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
            const unsigned aoff = ok ? brick_a_off((unsigned)(t.a_f0[j] + tapf), c32, (unsigned)kc_sw) + cbyte : OOB;
            char* dst = st + (wid + j * NW) * 1024;
            // (fetching the activation tiles with the non-temporal policy, to keep the weight tiles in the XCD's L2,
            //  was measured: 25 % slower - the column tiles of an XCD share A through that L2;
            //  profiles/r02/trunk_shapes_r02_nt_experiment.txt)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr_t)dst, 16, aoff, 0, 0, 0);
            if (TERMS == 3)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr_t)(dst + PLANE), 16, aoff, (int)p.a_lo_off, 0, 0);
        }
        kc = kcursor_next<BK>(kc, p.C, p.S);
    };
    auto issue = [&](const Tile& t, int slot) {
        issue_b(t, slot);
        issue_a(t, slot);
    };
    auto cursor_set = [&](int step) { kc = kcursor_at<BK>(step, p.C, p.S); };

    ACIMG_STAMP_DECL
    f32x4 acc[TM][TN];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    zero_acc();

    // `req` (DPOS 1): whose next K step (cursor kc) this step requests into the other stage: 0 nobody, 1 `a`, 2 `b`
    auto compute = [&](int slot, int req, const Tile& ta, const Tile& tb) {
        const char* sta = lds + slot * STAGE;
        const char* stb = sta + 2 * PLANE;
#pragma unroll
        for (int kh = 0; kh < KH; ++kh) {
            h16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int row = wm * WTM + i * 16 + li;
                const int off = row * ROWB + (((kh * 4 + g) ^ (BK == 32 ? swz(row) : (row & 7))) << 4);
                ah[i] = *reinterpret_cast<const h16x8*>(sta + off);
                if (TERMS == 3) al[i] = *reinterpret_cast<const h16x8*>(sta + PLANE + off);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int row = wn * WTN + j * 16 + li;
                const int off = row * ROWB + (((kh * 4 + g) ^ (BK == 32 ? swz(row) : (row & 7))) << 4);
                bh[j] = *reinterpret_cast<const h16x8*>(stb + off);
                if (TERMS == 3) bl[j] = *reinterpret_cast<const h16x8*>(stb + PLANE + off);
            }
            if (BK == 32) __builtin_amdgcn_sched_barrier(0);
#ifdef ACIMG_STAMP
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            ACIMG_STAMP_AT(3);
#endif
#ifdef ACIMG_ABLATE
            const bool lo_terms = !(p.flip & 8);   // ablation: only the hi x hi sweep
#else
            constexpr bool lo_terms = true;
#endif
            if (TERMS == 3 && lo_terms) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl[j], ah[i], acc[i][j], 0, 0, 0);
            }
            if (DPOS == 1 && kh == 0) {
                __builtin_amdgcn_sched_barrier(0);
                if (req == 1) issue_b(ta, slot ^ 1);
                else if (req == 2) issue_b(tb, slot ^ 1);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (TERMS == 3 && lo_terms) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[j], al[i], acc[i][j], 0, 0, 0);
            }
            if (DPOS == 1 && kh == 0) {
                __builtin_amdgcn_sched_barrier(0);
                if (req == 1) issue_a(ta, slot ^ 1);
                else if (req == 2) issue_a(tb, slot ^ 1);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[j], ah[i], acc[i][j], 0, 0, 0);
            ACIMG_STAMP_AT(4);
        }
    };

    // Output tile through ONE stage buffer, in HALVES passes of RH tile rows (BK = 32: two passes of 64 rows through
    // 32 KiB; BK = 64: the whole tile through 64 KiB): pass h holds the accumulator rows i in [h TM / HALVES,
    // (h + 1) TM / HALVES) of both wave rows, i.e. tile row (lr / WH) * 64 + h * WH + lr % WH at local row lr.  Image:
    // 16-byte chunk c of local row lr at chunk c ^ (lr & 31): accumulator-shaped writes and row-shaped reads are both
    // conflict free.  Exactly NST vector-memory instructions per wave (out-of-range offsets instead of branches).
    // Batch-norm partials come straight from the accumulators: a lane adds its 4 row blocks, a 16-lane DPP butterfly
    // adds the 16 pixel rows of a fragment (the lanes of a DPP row ARE the pixel rows of the MFMA layout), and the two
    // wave rows meet through 2 KiB of LDS - instead of every thread re-reading a column of the staged tile (32
    // dependent-latency LDS reads per thread: 14-17 % of a short-K tile's time, tools/stamp_probe.py).
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
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] *= SPLIT3_OUTSCALE;
        // EPI 2: the shortcut chunks of this thread's items (2 passes x 2 bricks per wave x hi / lo), requested before
        // anything else of the epilogue; item (h, q): brick q of the wave in pass h = pixel group (wave >> 1) of the pass,
        // channel group 2 (wave & 1) + q; lane l = pixel row l >> 2, physical 16-byte chunk l & 3 of the brick
        constexpr bool FUSED = EPI == 2 || EPI == 3;
        u32x4 sc_hi[FUSED ? 4 : 1], sc_lo[FUSED ? 4 : 1];      // EPI 2: hi / lo chunks; EPI 3: the two fp32 chunks
        unsigned f_off[FUSED ? 4 : 1];
        if constexpr (FUSED) {
            const __amdgpu_buffer_rsrc_t rsC =
                __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(e.f_sc), 0, e.f_sc_bytes, 0x00020000);
            const unsigned CQ = (unsigned)e.Nstore >> 5;
            const int pg = e_wid >> 1;
            const unsigned kc_e = (((unsigned)e_lane & 3u) ^ (0u - ((unsigned)e_lane >> 4))) & 3u;
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int trow = (pg >> 1) * WTM + h * WH + (pg & 1) * 16;        // first tile row of the pixel group
                    const int m = m0 + trow + (e_lane >> 2);
                    const unsigned cgl = 2u * (unsigned)(e_wid & 1) + (unsigned)q;   // channel group inside the tile
                    const unsigned cg = (unsigned)(n0 >> 5) + cgl;
                    const unsigned off = (((unsigned)(m0 + trow) >> 4) * CQ + cg) * 1024u + (unsigned)e_lane * 16u;
                    f_off[h * 2 + q] = m < e.M ? off : OOB;
                    if constexpr (EPI == 2) {
                        sc_hi[h * 2 + q] = __builtin_amdgcn_raw_buffer_load_b128(rsC, f_off[h * 2 + q], 0, 0);
                        sc_lo[h * 2 + q] = __builtin_amdgcn_raw_buffer_load_b128(rsC, f_off[h * 2 + q], (int)e.f_sc_lo, 0);
                    } else {    // row-major fp32 [M][Nstore]: this lane's 8 channels n0 + cgl 32 + kc 8 .. + 7 of pixel m
                        const unsigned o32 = m < e.M ? ((unsigned)m * (unsigned)e.Nstore + (unsigned)n0 + cgl * 32u + kc_e * 8u) * 4u
                                                     : OOB;
                        sc_hi[h * 2 + q] = __builtin_amdgcn_raw_buffer_load_b128(rsC, o32, 0, 0);
                        sc_lo[h * 2 + q] = __builtin_amdgcn_raw_buffer_load_b128(rsC, o32 + 16u, 0, 0);
                    }
                }
            if (te < 64) {      // this column tile's scale | shift -> red[0 .. 255] (read after the staging barrier)
                const float* src = (te < 32 ? e.f_scale : e.f_shift) + n0 + (te & 31) * 4;
                *reinterpret_cast<f32x4*>(red + te * 4) = *reinterpret_cast<const f32x4*>(src);
            } else if (EPI == 3 && te < 128) {      // ... and the shortcut's -> red[256 .. 511]
                const float* src = (te < 96 ? e.f_scale2 : e.f_shift2) + n0 + (te & 31) * 4;
                *reinterpret_cast<f32x4*>(red + 2 * BN + (te - 64) * 4) = *reinterpret_cast<const f32x4*>(src);
            }
        }
        if (!FUSED && e.stats) {
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
                if (e_li == 0) {     // rows past M hold exact zeros (their A pieces were out of range): they add nothing
                    const int n = e_wn * WTN + j * 16 + e_g * 4;
                    *reinterpret_cast<f32x4*>(red + (e_wm * 2 + 0) * BN + n) = s1;
                    *reinterpret_cast<f32x4*>(red + (e_wm * 2 + 1) * BN + n) = s2;
                }
            }
        }
        ACIMG_STAMP_AT(12);                         // statistics from the accumulators
        if constexpr (EPI == 1) {                   // no output: the partials meet, nothing else leaves the tile
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
#pragma unroll
        for (int h = 0; h < (EPI == 1 ? 0 : HALVES); ++h) {
#pragma unroll
            for (int ii = 0; ii < TM / HALVES; ++ii) {
                const int lr = e_wm * WH + ii * 16 + e_li;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int c = (e_wn * WTN + j * 16) / 4 + e_g;
                    tile[lr * CH + (c ^ (lr & (CH - 1)))] = acc[h * (TM / HALVES) + ii][j];
                }
            }
            ACIMG_STAMP_AT(8);                      // tile -> LDS writes issued
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            ACIMG_STAMP_AT(9);                      // ... and completed
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            ACIMG_STAMP_AT(10);                     // barrier
            if constexpr (FUSED) {
                // a wave turns two bricks of the pass (16 pixels x 32 channels each) into their hi / lo KiB: lane l takes
                // the 8 channels of physical chunk l & 3 of pixel row l >> 2 (two 16-byte fp32 chunks of the staged
                // tile), normalises, adds the shortcut, ReLU, 2^-2, splits; 64 lanes x 16 bytes = one contiguous KiB
                const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc(e.f_out, 0, e.f_out_bytes, 0x00020000);
                const unsigned lr = (unsigned)(e_wid >> 1) * 16u + ((unsigned)e_lane >> 2);
                const unsigned kc = (((unsigned)e_lane & 3u) ^ (0u - ((unsigned)e_lane >> 4))) & 3u;
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const unsigned cl = (2u * (unsigned)(e_wid & 1) + (unsigned)q) * 32u + kc * 8u;
                    const unsigned c4 = cl >> 2;
                    const f32x4 v0 = tile[lr * CH + (c4 ^ (lr & (CH - 1)))];
                    const f32x4 v1 = tile[lr * CH + ((c4 + 1u) ^ (lr & (CH - 1)))];
                    const f32x4 s0 = *reinterpret_cast<const f32x4*>(red + cl);
                    const f32x4 s1 = *reinterpret_cast<const f32x4*>(red + cl + 4);
                    const f32x4 t0 = *reinterpret_cast<const f32x4*>(red + BN + cl);
                    const f32x4 t1 = *reinterpret_cast<const f32x4*>(red + BN + cl + 4);
                    const h16x8 bh = __builtin_bit_cast(h16x8, sc_hi[h * 2 + q]);
                    const h16x8 bl = __builtin_bit_cast(h16x8, sc_lo[h * 2 + q]);
                    const f32x4 r0 = __builtin_bit_cast(f32x4, sc_hi[h * 2 + q]);      // EPI 3: the same registers as fp32
                    const f32x4 r1 = __builtin_bit_cast(f32x4, sc_lo[h * 2 + q]);
                    f32x4 u0 = s0, u1 = s1, w0 = t0, w1 = t1;
                    if constexpr (EPI == 3) {
                        u0 = *reinterpret_cast<const f32x4*>(red + 2 * BN + cl);
                        u1 = *reinterpret_cast<const f32x4*>(red + 2 * BN + cl + 4);
                        w0 = *reinterpret_cast<const f32x4*>(red + 3 * BN + cl);
                        w1 = *reinterpret_cast<const f32x4*>(red + 3 * BN + cl + 4);
                    }
                    h16x8 oh, ol;
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const float raw = k < 4 ? v0[k & 3] : v1[k & 3];
                        const float sk = k < 4 ? s0[k & 3] : s1[k & 3];
                        const float tk = k < 4 ? t0[k & 3] : t1[k & 3];
                        const float scv = EPI == 3 ? __builtin_fmaf(k < 4 ? r0[k & 3] : r1[k & 3], k < 4 ? u0[k & 3] : u1[k & 3],
                                                                    k < 4 ? w0[k & 3] : w1[k & 3])
                                                   : ((float)bh[k] + (float)bl[k]) * (1.f / SPLIT3_ASCALE);
                        const float o = fmaxf(__builtin_fmaf(raw, sk, tk) + scv, 0.f) * SPLIT3_ASCALE;
                        const _Float16 hh = (_Float16)o;
                        oh[k] = hh;
                        ol[k] = (_Float16)(o - (float)hh);
                    }
                    // (the lo plane's distance goes into the VECTOR offset, not the scalar one: with a register in soffset the
                    //  compiler's hazard recognizer - following the ISA manual - leaves no wait state between a 16-byte store
                    //  and a VALU write of its data registers, and gfx950 was seen taking a register written one instruction
                    //  after `buffer_store_dwordx4 v[a:a+3], v, s[..], sN offen` for the lanes it reads last)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, oh), rsO, f_off[h * 2 + q], 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, ol), rsO, f_off[h * 2 + q] + e.f_out_lo, 0, 0);
                }
            } else {
            f32x4 v[RH * CH / NTHR];
            unsigned off[RH * CH / NTHR];
            // thread te reads 16-byte chunk c = te % CH of local rows lr0 + RSTEP k, lr0 = te / CH < RSTEP: the row of
            // read k is a compile-time distance from the row of read 0 (32-bit offsets: the output is < 2 GiB)
            constexpr int RSTEP = NTHR / CH;
            static_assert(WH % RSTEP == 0 && (CH & (CH - 1)) == 0, "row-read mapping");
            const unsigned lr0 = (unsigned)te / CH, c = (unsigned)te % CH;
            const unsigned cx = c ^ lr0;
            const int nn = n0 + 4 * (int)c;
            const int mb = m0 + h * WH + (int)lr0;
#ifdef ACIMG_ABLATE
            const bool n_ok = nn < e.Nstore && (p.flip & 1) == 0;
#else
            const bool n_ok = nn < e.Nstore;
#endif
            const unsigned ob = ((unsigned)mb * (unsigned)e.ldy + (unsigned)nn) * 4u;
#pragma unroll
            for (int k = 0; k < RH * CH / NTHR; ++k) {      // all row reads in flight before the first store
                const int rk = RSTEP * k;                                    // local row distance (compile time)
                const int mk = (rk / WH) * WTM + rk % WH;                    // ... as a tile row distance
                v[k] = tile[(lr0 + rk) * CH + (cx ^ (unsigned)(rk & (CH - 1)))];
                off[k] = (n_ok && mb + mk < e.M) ? ob + (unsigned)(mk * e.ldy) * 4u : OOB;
            }
#pragma unroll
            for (int k = 0; k < RH * CH / NTHR; ++k)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v[k]), rsY, off[k], 0, 0);
            }
            ACIMG_STAMP_AT(11);                     // row reads + output stores issued
            if (h + 1 < HALVES) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();       // everyone has read this pass before the next one overwrites it
                asm volatile("" ::: "memory");
                ACIMG_STAMP_AT(13);                 // barrier
            }
        }
        // the two wave rows' partials (written before the first barrier above) -> this row block's statistics row
        const int which = (int)(((unsigned)te / BN) & 1u), col = (int)((unsigned)te % BN);
        float sum = 0.f;
        if (!FUSED && e.stats) sum = red[(0 * 2 + which) * BN + col] + red[(1 * 2 + which) * BN + col];
        {
            const int n = n0 + col;
            const unsigned soff = (!FUSED && e.stats && te < 2 * BN && n < e.stats_ld)
                                      ? (unsigned)((((long)mt * 2 + which) * e.stats_ld + n) * 4) : OOB;
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, sum), rsS, soff, 0, 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };

    // ================= whole tiles: units [0, ts_whole), this workgroup takes blockIdx.x + k * stride ============
    const int n_whole = p.ts_s > 1 ? p.ts_whole : n_units;
    int unit = blockIdx.x;
    int slot = 0;
    if (p.splits > 1 && (int)blockIdx.x >= (stride_units >> 1)) {
        // Stagger (p.splits = shader cycles): the second half of the grid - the workgroups that share a CU with one
        // of the first half under in-order dispatch - starts late, so that the two co-resident workgroups are in
        // different phases of a tile (one multiplies while the other stores) instead of in lockstep.  Pure timing:
        // any placement gives the same results.
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)p.splits) __builtin_amdgcn_s_sleep(16);
    }
    if (unit < n_whole) {
        Tile cur = setup(unit);
        kc = KCursorP{0, 0, 0, 0};
        issue(cur, 0);
        bool pend = false;              // epilogue stores of the previous tile are younger than this tile's first DMA
        for (;;) {
            const int next = unit + stride_units;
            const bool has_next = next < n_whole;
            Tile nxt = cur;
            for (int it = 0; it < p.kiters; ++it) {
                // the next tile's addresses (integer divisions) are worked out right before the last K step, while
                // this wave would only be waiting for that step's pieces
                if (it + 1 == p.kiters && has_next) nxt = setup(next);
                ACIMG_STAMP_AT(7);                  // everything not itemised below (setup, loop control)
                if (it == 0 && pend) wait_vmcnt<NST>();
                else wait_vmcnt<0>();
                ACIMG_STAMP_AT(0);                  // own DMA pieces landed
                __builtin_amdgcn_s_barrier();       // everyone's pieces of step `it` landed; everyone left step it-1
                ACIMG_STAMP_AT(1);                  // barrier
                int req = 0;
                if (it + 1 < p.kiters) {
                    req = 1;
                } else if (has_next) {
                    kc = KCursorP{0, 0, 0, 0};
                    req = 2;
                }
                if (DPOS == 0) {
                    if (req == 1) issue(cur, slot ^ 1);
                    else if (req == 2) issue(nxt, slot ^ 1);
                }
                ACIMG_STAMP_AT(2);                  // DMA issue
                __builtin_amdgcn_s_setprio(1);
                compute(slot, DPOS == 0 ? 0 : req, cur, nxt);
                __builtin_amdgcn_s_setprio(0);
                slot ^= 1;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();           // everyone has read the last stage: it becomes the output buffer
            asm volatile("" ::: "memory");
            ACIMG_STAMP_AT(5);                      // end-of-tile barrier
            epilogue(cur.mt, cur.nt, slot ^ 1);
            zero_acc();
            ACIMG_STAMP_AT(6);                      // epilogue
            pend = true;
            if (!has_next) break;
            cur = nxt;
            unit = next;
        }
        unit += stride_units;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
#ifdef ACIMG_STAMP
    if (p.slab && lane == 0) {
        unsigned* dbg = reinterpret_cast<unsigned*>(p.slab) + ((long)blockIdx.x * NW + wid) * 16;
#pragma unroll
        for (int i = 0; i < 14; ++i) dbg[i] = st_acc[i];
        dbg[14] = (unsigned)(__builtin_amdgcn_s_memtime() - st_begin);
        dbg[15] = (unsigned)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));   // HW_REG_XCC_ID
    }
#endif
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
        const Tile cur = setup(vt);
        __builtin_amdgcn_s_barrier();               // nobody still reads LDS from the previous unit
        cursor_set(it_begin);
        slot = 0;
        if (it_begin < it_end) issue(cur, 0);
        for (int it = it_begin; it < it_end; ++it) {
            wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            if (it + 1 < it_end) issue(cur, slot ^ 1);
            __builtin_amdgcn_s_setprio(1);
            compute(slot, 0, cur, cur);
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
