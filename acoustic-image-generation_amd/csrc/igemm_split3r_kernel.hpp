// RING form of the trunk kernel (round 3): the same f16x3 product on pre-split activations as igemm_split3d /
// igemm_split3dp_kernel (same LDS rows, same per-accumulator term order, hence the same bits per output element), but
// the K loop is a software pipeline INSIDE every wave instead of barrier-separated phases.
//
// Why (profiles/r02/ablate_probe_r02.txt): in the two-stage kernels every wave of a CU reads its 12 fragments right
// behind the step barrier (768 LDS cycles for 16 waves), then multiplies (1536 MFMA cycles per SIMD), then queues the
// next stage's LDS-DMA requests at the CU's one texture addresser (64 KiB at 64 B/clk = 1024 cycles, every wave
// blocked on its own requests): the three phases ADD (a K step of two co-resident workgroups takes ~3000 cycles, the
// ablation's parts sum to the whole).  Here
//   * one workgroup per CU, 8 waves, 256 VGPRs: wave tile 64x64 (TM = 4: workgroup tile 256x128) or 32x64 (TM = 2:
//     128x128 for the layers with few row tiles): 48 KiB of operands and 128 KiB of fragment reads per 1536 MFMA cycles
//     instead of 64 and 192;
//   * the fragments of step g + 1 are read WHILE the 48 MFMAs of step g run: the hi planes (live through all three
//     sweeps) into a second register set, bl' under the second and third sweep and al' under the third in place - one
//     or two ds_read_b128 per group of four MFMAs;
//   * a ring of THREE LDS slots: step g multiplies D(g) from registers, reads D(g+1) from its slot, and requests D(g+3)
//     into the slot D(g) left - one request between groups of MFMAs, never a burst; a request has between one and two
//     whole steps to land (counted vmcnt at the end of a step: everything but the youngest stage);
//   * ONE barrier per step (everyone's pieces of D(g+1) landed; everyone's reads of D(g) done);
//   * the operand stream does not stop at tile boundaries: the loader cursor walks (unit, K step) pairs ahead of the
//     multiplier, so the first steps of the next tile are in LDS (and its first fragments in registers) when a tile's
//     last MFMA issues;
//   * the output tile leaves through WAVE-PRIVATE staging (16 rows x 256 B per wave in the slot D(g) left; the
//     request that would have refilled it is deferred by one step): no workgroup barrier in the epilogue, 256-byte row
//     segments, stores left in flight under the next tile's MFMAs;
//   * units = whole tiles, then the K ranges of the tail tiles (same hand-off protocol as the other trunk kernels:
//     sc1 partial slabs + ticket, last arriver adds in range order), all in one stream.
// LDS: 3 slots x [A hi | A lo | B hi | B lo] (48 / 32 KiB each), then [4][2][128] floats of statistics scratch.
#pragma once
#include "igemm_split3dp_kernel.hpp"

namespace acimg {

template <int RA>
struct TileAddrR {
    int a_f0[RA], a_ih0[RA], a_iw0[RA];    // this lane's row of each of the wave's A pieces: pixel index / row / column at tap (0, 0)
    unsigned b_goff;                       // this lane's row of the wave's B piece, byte offset at K step 0
};

// s_waitcnt through the builtin, so that the compiler's own counter bookkeeping sees it (an `asm volatile` wait is opaque:
// behind it the compiler still protects every fragment register loaded one loop iteration earlier with a conservative
// lgkmcnt wait in front of its first use - which, in a pipelined loop, waits for the reads issued just before)
__device__ __forceinline__ void wait_lgkm0() { __builtin_amdgcn_s_waitcnt(0xC07F); }
template <int N>
__device__ __forceinline__ void wait_vm() {
    static_assert(N >= 0 && N < 64, "vmcnt immediate");
    __builtin_amdgcn_s_waitcnt((N & 15) | ((N >> 4) << 14) | 0x0F70);
}

template <int TM>
__global__ __launch_bounds__(512, 2) void igemm_split3r_kernel(const IgemmParams p, const int n_units,
                                                               const int stride_units) {
    constexpr int BK = 32, BN = 128, NTHR = 512, NW = 8, WGM = 4, WGN = 2, TN = 4;
    constexpr int WTM = TM * 16, WTN = TN * 16, BM = WGM * WTM;
    constexpr int ROWB = BK * 2;
    constexpr int A_PLANE = BM * ROWB, B_PLANE = BN * ROWB;
    constexpr int SLOT = 2 * A_PLANE + 2 * B_PLANE;
    constexpr int RA = BM / 16 / NW;                 // A pieces (16 rows x 64 B of one plane) per wave and plane: 2 / 1
    constexpr int RPS = 2 * (RA + 1);                // LDS-DMA requests per stage and wave: 6 / 4
    constexpr int NG = 3 * TM;                       // groups of TN MFMAs per step: 12 / 6
    static_assert((TM + TN) % TM == 0 && TN % TM == 0 && WGM * WGN == NW && (BM / 16) % NW == 0 && BN / 16 == NW,
                  "tile / wave mapping");
    static_assert(NW * 4096 <= SLOT, "wave-private output staging lives in one ring slot");
    typedef TileAddrR<RA> Tile;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* const lds = reinterpret_cast<char*>(smem);
    float* const red = smem + 3 * SLOT / 4;          // [WGM][2][BN] statistics partials; red[0] doubles as the tail flag

    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave id: a scalar (LDS-DMA destinations)
    const int wm = wid / WGN, wn = wid % WGN;
    const int li = lane & 15, g4 = lane >> 4;
    const int prow = lane >> 2, pch = lane & 3;      // this lane's row / physical chunk inside a DMA piece
    const int kc_sw = pch ^ swz(prow);               // logical k chunk it fetches (pieces start at multiples of 16 rows)
    const int Ktot = p.ntaps * p.C;
    const int ohw = p.OH * p.OW;
    constexpr unsigned b_lo_off = 8192u;             // operands in LDS-tile order (igemm_split3d_kernel.hpp, "bricks")
    const unsigned c32 = (unsigned)p.C * 32u;
    const EpiParams& e = p.e;

    const __amdgpu_buffer_rsrc_t rsA =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.A), 0, p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.B), 0, p.b_bytes, 0x00020000);

    // fragment read offsets inside a slot (row * 64 + swizzled chunk): fragment i of A is 1024 i bytes further
    const int fa_off = (wm * WTM + li) * ROWB + ((g4 ^ swz(li)) << 4);
    const int fb_off = 2 * A_PLANE + (wn * WTN + li) * ROWB + ((g4 ^ swz(li)) << 4);

    // ---- units: [0, ts_whole) whole tiles, then ts_s K ranges per tail tile -------------------------------------
    auto unit_range = [&](int u, int& vt, int& kb, int& ke, int& chunk) __attribute__((always_inline)) {
        if (p.ts_s > 1 && u >= p.ts_whole) {
            const int uu = u - p.ts_whole;
            const int t = uu / p.ts_s;
            chunk = uu - t * p.ts_s;
            vt = p.ts_whole + t;
            const int base = p.kiters / p.ts_s, extra = p.kiters - base * p.ts_s;
            kb = chunk * base + min(chunk, extra);
            ke = kb + base + (chunk < extra ? 1 : 0);
        } else {
            vt = u;
            chunk = -1;
            kb = 0;
            ke = p.kiters;
        }
    };

    // ---- loader: walks (unit, K step) ahead of the multiplier ----------------------------------------------------
    Tile lt;
    KCursorP lkc{0, 0, 0, 0};
    int l_unit = blockIdx.x, l_k = 0, l_ke = 0, l_slot = 0, n_issued = 0;
    // the (up to) two stages a step requests: set 0 = the regular one (spread over the step), set 1 = the one a unit's
    // last step deferred (requested first): per-lane source offsets and the slot's LDS byte offset
    unsigned st_aoff[2][RA] = {}, st_boff[2] = {0, 0};
    int st_base[2] = {0, 0};

    auto setup = [&](int vt) __attribute__((always_inline)) {
        int mt_, nt_;
        raster_tile(p, vt, mt_, nt_);
#pragma unroll
        for (int j = 0; j < RA; ++j) {
            const int row = (wid + j * NW) * 16 + prow;
            const int m = mt_ * BM + row;
            if (m < p.M) {
                const int img = m / ohw;
                const int r2 = m - img * ohw;
                const int oh = r2 / p.OW;
                const int ow = r2 - oh * p.OW;
                lt.a_ih0[j] = oh * p.stride - p.pad_t;
                lt.a_iw0[j] = ow * p.stride - p.pad_l;
                lt.a_f0[j] = (img * p.H + lt.a_ih0[j]) * p.W + lt.a_iw0[j];
            } else {
                lt.a_ih0[j] = -(1 << 28);
                lt.a_iw0[j] = -(1 << 28);
                lt.a_f0[j] = 0;
            }
        }
        lt.b_goff = p.b_brick + brick_b_off((unsigned)(nt_ * BN + wid * 16 + prow), (unsigned)(Ktot / BK), (unsigned)pch);
    };
    // the loader moves to a unit: `first` = this workgroup's first one, else the next of its list (the caller checked
    // that there is one); the tile's addresses are integer divisions: once per unit, between two groups of MFMAs
    auto advance_unit = [&](const bool first) __attribute__((always_inline)) {
        if (!first) l_unit += stride_units;
        int vt, chunk;
        unit_range(l_unit, vt, l_k, l_ke, chunk);
        setup(vt);
        lkc = kcursor_at<BK>(l_k, p.C, p.S);
    };
    // the loader has a stage to request (its current unit is set up and not exhausted)
    auto loader_more = [&]() __attribute__((always_inline)) -> bool { return l_k < l_ke; };
    auto loader_wants_unit = [&]() __attribute__((always_inline)) -> bool {
        return l_k == l_ke && l_unit + stride_units < n_units;
    };
    // describe the next stage in descriptor set `ds` and move the cursor on (needs loader_more())
    auto begin_stage = [&](const int ds) __attribute__((always_inline)) {
        const int tapf = lkc.r * p.W + lkc.s;
        const unsigned cbyte = (unsigned)lkc.c0 * 32u;
#pragma unroll
        for (int j = 0; j < RA; ++j) {
            const int ih = lt.a_ih0[j] + lkc.r, iw = lt.a_iw0[j] + lkc.s;
            const bool ok = (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
            st_aoff[ds][j] = ok ? brick_a_off((unsigned)(lt.a_f0[j] + tapf), c32, (unsigned)kc_sw) + cbyte : OOB;
        }
        st_boff[ds] = lt.b_goff + (unsigned)lkc.q * 16384u;
        st_base[ds] = l_slot * SLOT;
        lkc = kcursor_next<BK>(lkc, p.C, p.S);
        ++l_k;
        l_slot = l_slot == 2 ? 0 : l_slot + 1;
        ++n_issued;
    };
    // The same work in pieces, for the steady-state step (descriptor set 0): a piece runs behind the group of MFMAs named
    // by its argument, right before the request that needs its result, so that these few instructions sit BETWEEN
    // matrix instructions (a wave's serial chain of scalar / vector instructions is what a K step of this kernel waits
    // for: ~125 of them in one block cost more than the 48 MFMAs)
    int fs_tapf = 0;
    unsigned fs_cbyte = 0;
    auto begin_stage_part = [&](const int G) __attribute__((always_inline)) {
        if (G == 0) {                       // B offset, slot (requests 0, 1)
            fs_tapf = lkc.r * p.W + lkc.s;
            fs_cbyte = (unsigned)lkc.c0 * 32u;
            st_boff[0] = lt.b_goff + (unsigned)lkc.q * 16384u;
            st_base[0] = l_slot * SLOT;
            l_slot = l_slot == 2 ? 0 : l_slot + 1;
        }
        const int GA = NG == 12 ? 3 : 1;    // A pieces: j behind group GA + 4 j (TM = 4) / GA (TM = 2)
#pragma unroll
        for (int j = 0; j < RA; ++j)
            if (G == GA + 4 * j) {
                const int ih = lt.a_ih0[j] + lkc.r, iw = lt.a_iw0[j] + lkc.s;
                const bool ok = (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
                st_aoff[0][j] = ok ? brick_a_off((unsigned)(lt.a_f0[j] + fs_tapf), c32, (unsigned)kc_sw) + fs_cbyte : OOB;
            }
        if (G == NG - 3) {                  // the cursor moves on once every offset of this stage is formed
            lkc = kcursor_next<BK>(lkc, p.C, p.S);
            ++l_k;
            ++n_issued;
        }
    };
    // request r of the stage described in set `ds`: r = 0, 1: the B piece (hi, lo); then the A pieces (hi, lo each); ds
    // and r, like every index parameter below, are literals at each call site and fold after inlining
    auto issue_req = [&](const int ds, const int r) __attribute__((always_inline)) {
#ifdef ACIMG_ABLATE
        if (r < 2 ? (p.flip & 4) : (p.flip & 2)) return;     // ablation: no weight- / activation-tile requests
#endif
        if (r < 2) {
            char* dst = lds + st_base[ds] + 2 * A_PLANE + r * B_PLANE + wid * 1024;
            const unsigned bo = st_boff[ds];
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr_t)dst, 16, bo, r ? (int)b_lo_off : 0, 0, 0);
        } else {
            const int j = (r - 2) >> 1, pl = (r - 2) & 1;
            char* dst = lds + st_base[ds] + pl * A_PLANE + (wid + j * NW) * 1024;
            const unsigned ao = st_aoff[ds][j];      // (a scalar copy: an array element as the builtin's offset operand
                                                     //  makes hipcc's host pass drop the kernel without a diagnostic)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr_t)dst, 16, ao, pl ? (int)p.a_lo_off : 0, 0, 0);
        }
    };
    auto issue_stage_now = [&]() __attribute__((always_inline)) {                   // a whole stage in one go (prologue)
        begin_stage(0);
#pragma unroll
        for (int r = 0; r < RPS; ++r) issue_req(0, r);
    };

    // ---- multiplier state ----------------------------------------------------------------------------------------
    f32x4 acc[TM][TN];
    auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    zero_acc();
    // fragments: the hi planes are live until a step's last sweep, so they are DOUBLE BUFFERED; bl is dead after the first
    // sweep and al after the second, so the next step's bl / al are read in place under the sweeps that follow
    h16x8 fah[2][TM], fbh[2][TN], fal[TM], fbl[TN];

    // fragment reads of the NEXT step, by the sweep they are issued under: 0: ah', bh' into the other set (q in
    // [0, TM + TN)); 1: bl' in place (q in [0, TN)); 2: al' in place (q in [0, TM))
    auto read_frag = [&](const int s, const int kind, const int q, const char* slot) __attribute__((always_inline)) {
#ifdef ACIMG_ABLATE
        if (p.flip & 16) return;                             // ablation: no fragment reads
#endif
        if (kind == 0) {
            if (q < TM) fah[s][q] = *reinterpret_cast<const h16x8*>(slot + fa_off + q * 1024);
            else fbh[s][q - TM] = *reinterpret_cast<const h16x8*>(slot + fb_off + (q - TM) * 1024);
        } else if (kind == 1) {
            fbl[q] = *reinterpret_cast<const h16x8*>(slot + fb_off + B_PLANE + q * 1024);
        } else {
            fal[q] = *reinterpret_cast<const h16x8*>(slot + fa_off + A_PLANE + q * 1024);
        }
    };

    ACIMG_STAMP_DECL
    int c_unit = blockIdx.x, c_k = 0, c_ke = 0, c_vt = 0, c_chunk = -1, c_slot = 0, g = 0;
    int pend_mt = -1, pend_n0 = 0;                   // statistics partials waiting in `red` for their flush
    int* const flag = reinterpret_cast<int*>(red);

    // statistics partials of the tile finished last: the WGM wave rows meet here, one step (one barrier) later
    auto flush_stats = [&]() __attribute__((always_inline)) {
        const __amdgpu_buffer_rsrc_t rsS = __builtin_amdgcn_make_buffer_rsrc(
            e.stats, 0, e.stats ? (unsigned)((long)p.ras_tiles_m * 2 * e.stats_ld * 4) : 0u, 0x00020000);
        const int which = (tid >> 7) & 1, col = tid & (BN - 1);
        float sum = 0.f;
#pragma unroll
        for (int w = 0; w < WGM; ++w) sum += red[(w * 2 + which) * BN + col];
        const int n = pend_n0 + col;
        const unsigned soff = (tid < 2 * BN && n < e.stats_ld)
                                  ? (unsigned)((((long)pend_mt * 2 + which) * e.stats_ld + n) * 4) : OOB;
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, sum), rsS, soff, 0, 0);
        pend_mt = -1;
    };

    // Output tile of unit-tile `vt` through this wave's 4 KiB of ring slot `slot`: 16 pixel rows x 64 channels per
    // pass; 16-byte chunk c of row r lives at chunk c ^ r (accumulator-shaped writes and row-shaped reads both conflict
    // free).  Only this wave touches its region: no barrier.  EST + 0 vector-memory stores per wave (out-of-range
    // offsets for tail rows).
    auto epilogue = [&](int vt, int slot) __attribute__((always_inline)) {
        int mt, nt;
        raster_tile(p, vt, mt, nt);
        const __amdgpu_buffer_rsrc_t rsY =
            __builtin_amdgcn_make_buffer_rsrc(e.Y, 0, (unsigned)((long)e.M * e.ldy * 4), 0x00020000);
        // every index below is re-derived from an opaque copy of the thread id: otherwise the compiler hoists this block's
        // loop-invariant addresses out of the unit loop and keeps them in registers through the K loop
        int te = tid;
        asm volatile("" : "+v"(te));
        const int e_lane = te & 63, e_wid = te >> 6;
        const int e_wm = e_wid / WGN, e_wn = e_wid % WGN, e_li = e_lane & 15, e_g = e_lane >> 4;
        const int m0 = mt * BM + e_wm * WTM, n0 = nt * BN + e_wn * WTN;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] *= SPLIT3_OUTSCALE;
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
                if (e_li == 0) {   // rows past M hold exact zeros (their A pieces were out of range): they add nothing
                    const int n = e_wn * WTN + j * 16 + e_g * 4;
                    *reinterpret_cast<f32x4*>(red + (e_wm * 2 + 0) * BN + n) = s1;
                    *reinterpret_cast<f32x4*>(red + (e_wm * 2 + 1) * BN + n) = s2;
                }
            }
            pend_mt = mt;
            pend_n0 = nt * BN;
        }
        f32x4* const stg = reinterpret_cast<f32x4*>(lds + slot * SLOT + e_wid * 4096);
        const int rr = e_lane >> 4, cc = e_lane & 15;    // row-shaped view: row rr + 4 k, chunk cc
        const int nn = n0 + 4 * cc;
#ifdef ACIMG_ABLATE
        const bool n_ok = nn < e.Nstore && (p.flip & 1) == 0;   // ablation: no output stores
#else
        const bool n_ok = nn < e.Nstore;
#endif
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int j = 0; j < TN; ++j) stg[e_li * 16 + ((j * 4 + e_g) ^ e_li)] = acc[i][j];
            asm volatile("" ::: "memory");
            f32x4 v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = stg[(rr + 4 * k) * 16 + (cc ^ (rr + 4 * k))];
            wait_lgkm0();
            asm volatile("" ::: "memory");
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int m = m0 + i * 16 + rr + 4 * k;
                const unsigned off = (n_ok && m < e.M) ? ((unsigned)m * (unsigned)e.ldy + (unsigned)nn) * 4u : OOB;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v[k]), rsY, off, 0, 0);
            }
        }
    };

    // K range of a tail tile: park the partial sums (sc1), take a ticket; the last arriver adds all ranges in range
    // order (-> true: the caller runs the epilogue)
    auto handoff = [&](int vt, int chunk) __attribute__((always_inline)) -> bool {
        const int tl = vt - p.ts_whole;
        float* const slot0 = p.ts_partial + (long)tl * p.ts_s * (BM * BN);
        const __amdgpu_buffer_rsrc_t rsP =
            __builtin_amdgcn_make_buffer_rsrc(slot0 + (long)chunk * (BM * BN), 0, BM * BN * 4, 0x00020000);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[i][j]), rsP,
                                                       ((i * TN + j) * NTHR + tid) * 16, 0, 16);     // (soffset 0: DESIGN 7d)
        // (`red` carries the flag below: a tail unit has at least two steps, so the statistics partials of the unit
        //  before it were flushed at the top of its first step)
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
        const int last = __builtin_amdgcn_readfirstlane(*reinterpret_cast<volatile int*>(flag));   // uniform
        zero_acc();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                // everyone has read the flag before `red` is reused
        if (!last) return false;
        // the s partial tiles of this unit's tile, added in range order.  The loads are what this takes (s x 128 KiB per
        // tile from L2 / HBM at ~65 GB/s per CU): TWO ranges x TN fragments (8 loads of 1 KiB per wave) are in flight at a
        // time - with 4 the reduction was latency-bound (one ~1.5 us round trip per 4 KiB and wave: 53 us of tail for
        // ten 4-range tiles, profiles/r03/trunk_shapes_r03i_ring_tail_sweep.txt)
        for (int c = 0; c < p.ts_s; c += 2) {
            const bool two = c + 1 < p.ts_s;
            const __amdgpu_buffer_rsrc_t rsQ0 =
                __builtin_amdgcn_make_buffer_rsrc(slot0 + (long)c * (BM * BN), 0, BM * BN * 4, 0x00020000);
            const __amdgpu_buffer_rsrc_t rsQ1 = __builtin_amdgcn_make_buffer_rsrc(
                slot0 + (long)(c + (two ? 1 : 0)) * (BM * BN), 0, two ? BM * BN * 4 : 0, 0x00020000);   // size 0: reads zeros
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                f32x4 v0[TN], v1[TN];
#pragma unroll
                for (int j = 0; j < TN; ++j)     // sc1 loads: never a stale L1 / L2 copy
                    v0[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                          rsQ0, tid * 16, (i * TN + j) * NTHR * 16, 16));
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    v1[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                          rsQ1, tid * 16, (i * TN + j) * NTHR * 16, 16));
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = (acc[i][j] + v0[j]) + v1[j];
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        return true;
    };

    // ---- one K step: MFMAs of D(g) from set P, reads of D(g+1), requests in between -------------------------------
    // returns false after the last step of this workgroup's stream.  Hot path: one barrier, NG groups of TN MFMAs with
    // one or two ds_read_b128 and at most one LDS-DMA request (a uniform branch) behind each, two counted waits.
    // FAST (a literal, like P): the steady state - not a unit's last step, exactly one stage owed, the loader has one,
    // no statistics waiting - with every decision below folded away: per-wave instruction issue is what a K step of a
    // one-workgroup-per-CU kernel has least of (two waves per SIMD: ~190 scalar and ~140 vector issue slots per wave
    // and step next to the 48 MFMAs), so the common step carries no control flow beyond the once-per-unit branch.
    auto step = [&](const int P, const bool FAST) __attribute__((always_inline)) -> bool {
        ACIMG_STAMP_AT(7);                           // everything not itemised (loop control, statistics flush, copies)
        __builtin_amdgcn_s_barrier();                // D(g+1) landed for everyone; everyone's reads of D(g) are done
        asm volatile("" ::: "memory");
        ACIMG_STAMP_AT(1);                           // step barrier
        if (!FAST && pend_mt >= 0) flush_stats();
        const bool unit_end = FAST ? false : c_k + 1 == c_ke;
        const bool has_next = FAST ? true : !(unit_end && c_unit + stride_units >= n_units);
        // stages to request now: up to D(g+3), or D(g+2) in a unit's last step (its slot stages the output tile); the
        // step behind a unit's last one therefore owes two: the deferred one goes first (`burst`, set 1).  Decided behind
        // the first group of MFMAs: the matrix pipe is busy while the scalar unit works this out.
#ifdef ACIMG_ABLATE
        const bool lo_terms = !(p.flip & 8);         // ablation: only the hi x hi sweep
#else
        constexpr bool lo_terms = true;
#endif
        // The two waves of a SIMD (w and w + 4) run the same program from the same barrier: left alone they alternate
        // their MFMAs and then both sit in the non-matrix instructions behind a group at the same time, the matrix pipe
        // idle.  Waves 4-7 therefore do the step's scalar work BEFORE their first group, waves 0-3 behind it: half a
        // group of offset that the in-order issue then keeps (one wave's reads and requests under the other's MFMAs).
        const bool late_half = !FAST && p.splits > 1 && wid >= NW / 2;     // (acimg_configure trunk_stagger > 0)
        bool burst = false, iss = FAST;
        auto decide = [&]() __attribute__((always_inline)) {
            if (FAST) return;                        // (the steady state: begin_stage_part, spread over the groups)
            const int want = (g + (unit_end ? 3 : 4)) - n_issued;
            if (want >= 2 && loader_more()) {
                begin_stage(1);
                burst = true;
                if (loader_wants_unit()) advance_unit(false);    // rare: the deferred stage was its unit's last
            }
            if (want - (burst ? 1 : 0) >= 1 && loader_more()) {
                begin_stage(0);
                iss = true;
            }
        };
        const int nslot = c_slot == 2 ? 0 : c_slot + 1;
        const char* const rd = lds + nslot * SLOT;
        // request slots: without a burst the regular stage is spread over the whole step; with one, the deferred stage
        // takes the first half of the step (it has less than a step to land) and the regular one the second
#define ACIMG_R_GROUP(G)                                                                                              \
        {                                                                                                             \
            constexpr int T_ = (G) / TM, i_ = (G) % TM;                                                               \
            if constexpr ((G) == 0) {                                                                                 \
                if (late_half) decide();                                                                              \
            }                                                                                                         \
            if (T_ == 2 || lo_terms)                                                                                  \
            _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                                          \
                if constexpr (T_ == 0)                                                                                \
                    acc[i_][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fbl[j], fah[P][i_], acc[i_][j], 0, 0, 0);     \
                else if constexpr (T_ == 1)                                                                           \
                    acc[i_][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fbh[P][j], fal[i_], acc[i_][j], 0, 0, 0);     \
                else                                                                                                  \
                    acc[i_][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fbh[P][j], fah[P][i_], acc[i_][j], 0, 0, 0);  \
            }                                                                                                         \
            __builtin_amdgcn_sched_barrier(0);                                                                        \
            if constexpr ((G) == 0) {                                                                                 \
                if (!late_half) decide();                                                                             \
            }                                                                                                         \
            /* the next step's fragments (read unconditionally: behind the stream's last step they are never used) */ \
            if constexpr (T_ == 0) {                                                                                  \
                constexpr int n_ = (TM + TN) / TM;                                                                    \
                read_frag(1 - P, 0, i_ * n_, rd);                                                                     \
                read_frag(1 - P, 0, i_ * n_ + 1, rd);                                                                 \
                if constexpr (n_ == 3) read_frag(1 - P, 0, i_ * n_ + 2, rd);                                          \
            } else if constexpr (T_ == 1) {                                                                           \
                constexpr int n_ = TN / TM;                                                                           \
                read_frag(0, 1, i_ * n_, rd);                                                                         \
                if constexpr (n_ == 2) read_frag(0, 1, i_ * n_ + 1, rd);                                              \
            } else if constexpr (i_ < (TM + 1) / 2) {                                                                 \
                /* al' in place, all of it behind the first half of the last sweep: the reads' latency is covered   */ \
                /* by the groups that follow, not exposed in front of the step's closing wait                       */ \
                read_frag(0, 2, 2 * i_, rd);                                                                          \
                if constexpr (2 * i_ + 1 < TM) read_frag(0, 2, 2 * i_ + 1, rd);                                       \
            }                                                                                                         \
            if (FAST) begin_stage_part(G);                                                                            \
            if constexpr (NG == 12) {                                                                                 \
                if (burst) {                                                                                          \
                    if constexpr ((G) < 6) issue_req(1, (G));                                                         \
                    else if (iss) issue_req(0, (G) - 6);                                                              \
                } else if (iss) {                                                                                     \
                    if constexpr ((G) % 2 == 0) issue_req(0, (G) / 2);                                                \
                }                                                                                                     \
            } else {                                                                                                  \
                if (burst) {                                                                                          \
                    if constexpr ((G) < 2) {                                                                          \
                        issue_req(1, 2 * (G));                                                                        \
                        issue_req(1, 2 * (G) + 1);                                                                    \
                    } else if (iss) issue_req(0, (G) - 2);                                                            \
                } else if (iss) {                                                                                     \
                    if constexpr ((G) < 4) issue_req(0, (G));                                                         \
                }                                                                                                     \
            }                                                                                                         \
            if constexpr ((G) == NG - 2) {                                                                            \
                if (!FAST && loader_wants_unit()) advance_unit(false);  /* once per unit: the next tile's addresses */ \
            }                                                                                                         \
            __builtin_amdgcn_sched_barrier(0);                                                                        \
        }
        ACIMG_R_GROUP(0) ACIMG_R_GROUP(1) ACIMG_R_GROUP(2) ACIMG_R_GROUP(3) ACIMG_R_GROUP(4) ACIMG_R_GROUP(5)
        if constexpr (NG == 12) {
            ACIMG_R_GROUP(6) ACIMG_R_GROUP(7) ACIMG_R_GROUP(8) ACIMG_R_GROUP(9) ACIMG_R_GROUP(10) ACIMG_R_GROUP(11)
        }
#undef ACIMG_R_GROUP
        // D(g+2) must have landed before the next barrier: everything but the youngest stage, if one was requested
        // beyond it in this step
        ACIMG_STAMP_AT(4);                           // MFMA / read / request issue (+ the reads' completion: the stamp waits)
        if (FAST || n_issued == g + 4) wait_vm<RPS>();
        else wait_vm<0>();
        wait_lgkm0();
        ACIMG_STAMP_AT(0);                           // own DMA pieces of D(g+2) landed
        ++g;
        ++c_k;
        const int slot_now = c_slot;                 // the slot D(g) left: free until the next barrier
        c_slot = nslot;
        if (unit_end) {
            bool whole = true;                       // this workgroup holds the finished tile
            if (c_chunk >= 0) whole = handoff(c_vt, c_chunk);
            ACIMG_STAMP_AT(5);                       // hand-off of a K range (tail units)
            if (whole) epilogue(c_vt, slot_now);
            zero_acc();
            wait_lgkm0();
            ACIMG_STAMP_AT(6);                       // epilogue
            if (!has_next) return false;
            c_unit += stride_units;
            unit_range(c_unit, c_vt, c_k, c_ke, c_chunk);
        }
        return true;
    };

    // ---- prologue -------------------------------------------------------------------------------------------------
    if (c_unit >= n_units) return;
    unit_range(c_unit, c_vt, c_k, c_ke, c_chunk);
    {
        // D(0), D(1) exist (a unit has at least two steps); D(2) if the stream has a third step
        advance_unit(true);
        issue_stage_now();
        issue_stage_now();
        if (loader_wants_unit()) advance_unit(false);
        const bool third = loader_more();
        if (third) issue_stage_now();
        if (loader_wants_unit()) advance_unit(false);
        if (third) wait_vm<2 * RPS>();
        else wait_vm<RPS>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const char* const rd0 = lds;
#pragma unroll
        for (int q = 0; q < TM + TN; ++q) read_frag(0, 0, q, rd0);
#pragma unroll
        for (int q = 0; q < TN; ++q) read_frag(0, 1, q, rd0);
#pragma unroll
        for (int q = 0; q < TM; ++q) read_frag(0, 2, q, rd0);
        if (third) wait_vm<RPS>();
        else wait_vm<0>();
        wait_lgkm0();
    }
    // Steady steps run in PAIRS (register sets 0 -> 1 -> 0), every other step alone from set 0 with the next step's hi
    // planes copied back from set 1 behind it (32 v_mov): the loop then has ONE register-set parity, and no merge point
    // keeps both sets alive (two parities x two step forms, merged, spilled > 200 registers).
    auto steady_pair = [&]() __attribute__((always_inline)) -> bool {
        return c_k + 2 < c_ke && n_issued == g + 3 && pend_mt < 0 && l_ke - l_k >= 2;   // the loader stays inside its unit
    };
    for (;;) {
        while (steady_pair()) {                      // the tight loop: nothing but steady steps
            step(0, true);
            step(1, true);
        }
        if (!step(0, false)) break;
#pragma unroll
        for (int i = 0; i < TM; ++i) fah[0][i] = fah[1][i];
#pragma unroll
        for (int j = 0; j < TN; ++j) fbh[0][j] = fbh[1][j];
    }
#ifdef ACIMG_STAMP
    if (p.slab && lane == 0) {
        unsigned* dbg = reinterpret_cast<unsigned*>(p.slab) + ((long)blockIdx.x * NW + wid) * 16;
#pragma unroll
        for (int i = 0; i < 14; ++i) dbg[i] = st_acc[i];
        dbg[14] = (unsigned)(__builtin_amdgcn_s_memtime() - st_begin);
        dbg[15] = (unsigned)g;
    }
#endif
    // the last tile's statistics partials
    __builtin_amdgcn_s_barrier();
    if (pend_mt >= 0) flush_stats();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

}  // namespace acimg
