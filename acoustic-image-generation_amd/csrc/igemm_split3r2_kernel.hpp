// TWO-SLOT form of the ring kernel (igemm_split3r_kernel.hpp) at the occupancy of the shipped trunk kernels: 128x128
// tile, 8 waves of 64x32, TWO workgroups per CU (128 VGPRs, 74 KiB of LDS), 512 resident slots - so tile quantisation
// and the tail split behave as in igemm_split3dp_kernel - but with the ring kernel's K loop: the fragments of step
// g + 1 are read WHILE the 24 MFMAs of step g run, the stage's four LDS-DMA requests sit between the first groups of
// MFMAs, the steady-state step is one basic block and every wait is visible to the compiler.
//
// What makes that fit 128 registers: the sweep order of a K step is  lo*hi (bl . ah),  hi*hi (bh . ah),  hi*lo (bh . al)
// - so bl is dead after the first sweep, ah after the second and al[i] after its own MFMAs of the third: the next step's
// bl', ah', al' are read IN PLACE under the sweeps that follow, and only bh (2 fragments = 8 registers) is double
// buffered: 32 accumulator + 56 fragment registers.  (The per-accumulator term order differs from the other trunk
// kernels': results agree to rounding, not bit for bit; a replay of this kernel is bit-identical to itself.)
//
// Ring of TWO LDS slots: step g multiplies D(g) from registers, reads D(g+1) from slot (g+1) % 2 and requests D(g+2)
// into slot g % 2 (D(g) left it during step g - 1: every wave's reads are complete before the step barrier); a request
// has one step to land (requests behind the first groups; vmcnt(0) at the end of the step), as in the shipped kernels.
// The output tile leaves through a DEDICATED 8 KiB of wave-private staging (8 pixel rows x 128 B per wave and pass), so
// the operand stream never stops at a tile boundary and there is no deferred stage: every step requests exactly one
// stage while the loader has one.  Units = whole tiles, then the K ranges of the tail tiles (hand-off as elsewhere).
// LDS: 2 slots x [A hi | A lo | B hi | B lo] (32 KiB each), 8 KiB of output staging, [2][2][128] floats of statistics.
#pragma once
#include "igemm_split3r_kernel.hpp"

namespace acimg {

__global__ __launch_bounds__(512, 4) void igemm_split3r2_kernel(const IgemmParams p, const int n_units,
                                                                const int stride_units) {
    constexpr int BK = 32, BM = 128, BN = 128, NTHR = 512, NW = 8, WGM = 2, WGN = 4, TM = 4, TN = 2;
    constexpr int WTM = TM * 16, WTN = TN * 16;
    constexpr int ROWB = BK * 2;
    constexpr int PLANE = BM * ROWB;                 // 8 KiB: one fp16 plane of one operand of one stage
    constexpr int SLOT = 4 * PLANE;                  // [A hi | A lo | B hi | B lo]
    constexpr int STG = 2 * SLOT;                    // output staging: 1 KiB per wave
    static_assert(WGM * WGN == NW && BM / 16 == NW && BN / 16 == NW, "one A piece and one B piece per wave and plane");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* const lds = reinterpret_cast<char*>(smem);
    float* const red = smem + (STG + NW * 1024) / 4;     // [WGM][2][BN] statistics partials; red[0] doubles as the tail flag

    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid / WGN, wn = wid % WGN;
    const int li = lane & 15, g4 = lane >> 4;
    const int prow = lane >> 2, pch = lane & 3;
    const int kc_sw = pch ^ swz(prow);
    const int Ktot = p.ntaps * p.C;
    const int ohw = p.OH * p.OW;
    const unsigned b_lo_off = (unsigned)((long)p.Nld * Ktot * 2);
    const EpiParams& e = p.e;

    const __amdgpu_buffer_rsrc_t rsA =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.A), 0, p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.B), 0, p.b_bytes, 0x00020000);

    const int fa_off = (wm * WTM + li) * ROWB + ((g4 ^ swz(li)) << 4);
    const int fb_off = 2 * PLANE + (wn * WTN + li) * ROWB + ((g4 ^ swz(li)) << 4);

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

    // ---- loader: walks (unit, K step) two stages ahead of the multiplier -------------------------------------------
    int lt_off = 0, lt_ih0 = 0, lt_iw0 = 0;          // this lane's row of the wave's A piece
    unsigned lt_boff = 0;                            // ... and of its B piece (byte offset at k = 0, or OOB)
    KCursorP lkc{0, 0, 0, 0};
    int l_unit = blockIdx.x, l_k = 0, l_ke = 0, l_slot = 0;
    unsigned st_aoff = 0, st_boff = 0;               // the stage being requested
    int st_base = 0;

    auto setup = [&](int vt) __attribute__((always_inline)) {
        int mt_, nt_;
        raster_tile(p, vt, mt_, nt_);
        const int m = mt_ * BM + wid * 16 + prow;
        if (m < p.M) {
            const int img = m / ohw;
            const int r2 = m - img * ohw;
            const int oh = r2 / p.OW;
            const int ow = r2 - oh * p.OW;
            lt_ih0 = oh * p.stride - p.pad_t;
            lt_iw0 = ow * p.stride - p.pad_l;
            lt_off = ((img * p.H + lt_ih0) * p.W + lt_iw0) * p.lda * 2 + kc_sw * 16;
        } else {
            lt_ih0 = -(1 << 28);
            lt_iw0 = -(1 << 28);
            lt_off = 0;
        }
        const int n = nt_ * BN + wid * 16 + prow;
        lt_boff = n < p.Nld ? (unsigned)(((long)n * Ktot + kc_sw * 8) * 2) : OOB;
    };
    auto advance_unit = [&](const bool first) __attribute__((always_inline)) {
        if (!first) l_unit += stride_units;
        int vt, chunk;
        unit_range(l_unit, vt, l_k, l_ke, chunk);
        setup(vt);
        lkc = kcursor_at<BK>(l_k, p.C, p.S);
    };
    auto loader_more = [&]() __attribute__((always_inline)) -> bool { return l_k < l_ke; };
    auto loader_wants_unit = [&]() __attribute__((always_inline)) -> bool {
        return l_k == l_ke && l_unit + stride_units < n_units;
    };
    // the next stage's source offsets and slot (cursor moves on)
    auto begin_stage = [&]() __attribute__((always_inline)) {
        const int tapoff = ((lkc.r * p.W + lkc.s) * p.lda + lkc.c0) * 2;
        const int ih = lt_ih0 + lkc.r, iw = lt_iw0 + lkc.s;
        const bool ok = (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
        st_aoff = ok ? (unsigned)(lt_off + tapoff) : OOB;
        st_boff = lt_boff == OOB ? OOB : lt_boff + (unsigned)(lkc.q * (BK * 2));
        st_base = l_slot * SLOT;
        lkc = kcursor_next<BK>(lkc, p.C, p.S);
        ++l_k;
        l_slot ^= 1;
    };
    // request r of the stage begun last: 0, 1 = the B piece (hi, lo), 2, 3 = the A piece (hi, lo)
    auto issue_req = [&](const int r) __attribute__((always_inline)) {
#ifdef ACIMG_ABLATE
        if (r < 2 ? (p.flip & 4) : (p.flip & 2)) return;
#endif
        if (r < 2) {
            char* dst = lds + st_base + (2 + r) * PLANE + wid * 1024;
            const unsigned bo = st_boff;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr_t)dst, 16, bo, r ? (int)b_lo_off : 0, 0, 0);
        } else {
            char* dst = lds + st_base + (r - 2) * PLANE + wid * 1024;
            const unsigned ao = st_aoff;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr_t)dst, 16, ao, r == 3 ? (int)p.a_lo_off : 0, 0, 0);
        }
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
    h16x8 fah[TM], fal[TM], fbl[TN], fbh[2][TN];     // only bh is double buffered (see the header)

    // kind 0: bh'[q] into set s; 1: bl'[q]; 2: ah'[q]; 3: al'[q] (in place)
    auto read_frag = [&](const int s, const int kind, const int q, const char* slot) __attribute__((always_inline)) {
#ifdef ACIMG_ABLATE
        if (p.flip & 16) return;
#endif
        if (kind == 0) fbh[s][q] = *reinterpret_cast<const h16x8*>(slot + fb_off + q * 1024);
        else if (kind == 1) fbl[q] = *reinterpret_cast<const h16x8*>(slot + fb_off + PLANE + q * 1024);
        else if (kind == 2) fah[q] = *reinterpret_cast<const h16x8*>(slot + fa_off + q * 1024);
        else fal[q] = *reinterpret_cast<const h16x8*>(slot + fa_off + PLANE + q * 1024);
    };

    int c_unit = blockIdx.x, c_k = 0, c_ke = 0, c_vt = 0, c_chunk = -1, c_slot = 0;
    int pend_mt = -1, pend_n0 = 0;
    int* const flag = reinterpret_cast<int*>(red);

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

    // Output tile through this wave's 1 KiB of staging: 8 pixel rows x 32 channels (128 B) per pass, two passes per
    // 16-row fragment; 16-byte chunk c of row r at chunk c ^ (r & 7).  Only this wave touches its region: no barrier.
    auto epilogue = [&](int vt) __attribute__((always_inline)) {
        int mt, nt;
        raster_tile(p, vt, mt, nt);
        const __amdgpu_buffer_rsrc_t rsY =
            __builtin_amdgcn_make_buffer_rsrc(e.Y, 0, (unsigned)((long)e.M * e.ldy * 4), 0x00020000);
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
                if (e_li == 0) {
                    const int n = e_wn * WTN + j * 16 + e_g * 4;
                    *reinterpret_cast<f32x4*>(red + (e_wm * 2 + 0) * BN + n) = s1;
                    *reinterpret_cast<f32x4*>(red + (e_wm * 2 + 1) * BN + n) = s2;
                }
            }
            pend_mt = mt;
            pend_n0 = nt * BN;
        }
        f32x4* const stg = reinterpret_cast<f32x4*>(lds + STG + e_wid * 1024);
        const int rr = e_lane >> 3, cc = e_lane & 7;     // row-shaped view: row rr (0..7), chunk cc (0..7)
        const int nn = n0 + 4 * cc;
#ifdef ACIMG_ABLATE
        const bool n_ok = nn < e.Nstore && (p.flip & 1) == 0;
#else
        const bool n_ok = nn < e.Nstore;
#endif
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {            // rows 8 h .. 8 h + 7 of the fragment: written by the lanes that hold them
                if ((e_li >> 3) == h) {
#pragma unroll
                    for (int j = 0; j < TN; ++j) stg[(e_li & 7) * 8 + ((j * 4 + e_g) ^ (e_li & 7))] = acc[i][j];
                }
                asm volatile("" ::: "memory");
                const f32x4 v = stg[rr * 8 + (cc ^ rr)];
                wait_lgkm0();
                asm volatile("" ::: "memory");
                const int m = m0 + i * 16 + h * 8 + rr;
                const unsigned off = (n_ok && m < e.M) ? ((unsigned)m * (unsigned)e.ldy + (unsigned)nn) * 4u : OOB;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsY, off, 0, 0);
            }
        }
    };

    auto handoff = [&](int vt, int chunk) __attribute__((always_inline)) -> bool {
        const int tl = vt - p.ts_whole;
        float* const slot0 = p.ts_partial + (long)tl * p.ts_s * (BM * BN);
        const __amdgpu_buffer_rsrc_t rsP =
            __builtin_amdgcn_make_buffer_rsrc(slot0 + (long)chunk * (BM * BN), 0, BM * BN * 4, 0x00020000);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[i][j]), rsP, tid * 16,
                                                       (i * TN + j) * NTHR * 16, 16);
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
        const int last = __builtin_amdgcn_readfirstlane(*reinterpret_cast<volatile int*>(flag));
        zero_acc();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (!last) return false;
        for (int c = 0; c < p.ts_s; ++c) {
            const __amdgpu_buffer_rsrc_t rsQ =
                __builtin_amdgcn_make_buffer_rsrc(slot0 + (long)c * (BM * BN), 0, BM * BN * 4, 0x00020000);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                               rsQ, tid * 16, (i * TN + j) * NTHR * 16, 16));
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        return true;
    };

    // ---- one K step.  P: bh set holding D(g).  FAST (a literal): not a unit's last step, the loader has a stage, no
    // statistics waiting: one basic block.  Returns false after the stream's last step. ------------------------------
    auto step = [&](const int P, const bool FAST) __attribute__((always_inline)) -> bool {
        __builtin_amdgcn_s_barrier();                // D(g+1) landed for everyone; everyone's reads of D(g) are done
        asm volatile("" ::: "memory");
        if (!FAST && pend_mt >= 0) flush_stats();
        // every step requests exactly one stage while the stream has one: a loader that has finished its unit moves to
        // the next one BEFORE this step's request (once per tile, on the slow path only)
        if (!FAST && loader_wants_unit()) advance_unit(false);
        const bool unit_end = FAST ? false : c_k + 1 == c_ke;
        const bool has_next = FAST ? true : !(unit_end && c_unit + stride_units >= n_units);
        const bool iss = FAST ? true : loader_more();
#ifdef ACIMG_ABLATE
        const bool lo_terms = !(p.flip & 8);
#else
        constexpr bool lo_terms = true;
#endif
        const char* const rd = lds + (c_slot ^ 1) * SLOT;
        // groups of two MFMAs (one fragment row i): sweep 0 = bl . ah, 1 = bh . ah, 2 = bh . al
#define ACIMG_R2_GROUP(G)                                                                                             \
        {                                                                                                             \
            constexpr int T_ = (G) / TM, i_ = (G) % TM;                                                               \
            if constexpr ((G) == 0) {                                                                                 \
                if (iss) begin_stage();              /* D(g+2): offsets + slot (a dozen instructions) */             \
            }                                                                                                         \
            if (T_ == 1 || lo_terms)                                                                                  \
            _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                                          \
                if constexpr (T_ == 0)                                                                                \
                    acc[i_][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fbl[j], fah[i_], acc[i_][j], 0, 0, 0);        \
                else if constexpr (T_ == 1)                                                                           \
                    acc[i_][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fbh[P][j], fah[i_], acc[i_][j], 0, 0, 0);     \
                else                                                                                                  \
                    acc[i_][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fbh[P][j], fal[i_], acc[i_][j], 0, 0, 0);     \
            }                                                                                                         \
            __builtin_amdgcn_sched_barrier(0);                                                                        \
            /* the next step's fragments: bh' (other set) under the first sweep; bl' once the first sweep has issued; */ \
            /* ah'[i] behind the second sweep's MFMAs of row i; al'[i] behind the third sweep's                       */ \
            if constexpr ((G) < TN) read_frag(1 - P, 0, (G), rd);                                                     \
            if constexpr ((G) >= TM && (G) < TM + TN) read_frag(0, 1, (G) - TM, rd);                                  \
            if constexpr (T_ == 1) read_frag(0, 2, i_, rd);                                                           \
            if constexpr (T_ == 2) read_frag(0, 3, i_, rd);                                                           \
            if constexpr ((G) < 4) {                                                                                  \
                if (iss) issue_req(G);                                                                                \
            }                                                                                                         \
            __builtin_amdgcn_sched_barrier(0);                                                                        \
        }
        ACIMG_R2_GROUP(0) ACIMG_R2_GROUP(1) ACIMG_R2_GROUP(2) ACIMG_R2_GROUP(3) ACIMG_R2_GROUP(4) ACIMG_R2_GROUP(5)
        ACIMG_R2_GROUP(6) ACIMG_R2_GROUP(7) ACIMG_R2_GROUP(8) ACIMG_R2_GROUP(9) ACIMG_R2_GROUP(10) ACIMG_R2_GROUP(11)
#undef ACIMG_R2_GROUP
        wait_vm<0>();                                // D(g+2) landed (own pieces): it is read from the next barrier on
        wait_lgkm0();
        ++c_k;
        c_slot ^= 1;
        if (unit_end) {
            bool whole = true;
            if (c_chunk >= 0) whole = handoff(c_vt, c_chunk);
            if (whole) epilogue(c_vt);
            zero_acc();
            wait_lgkm0();
            if (!has_next) return false;
            c_unit += stride_units;
            unit_range(c_unit, c_vt, c_k, c_ke, c_chunk);
        }
        return true;
    };

    // ---- prologue: D(0), D(1) requested; D(0) into registers -------------------------------------------------------
    if (c_unit >= n_units) return;
    unit_range(c_unit, c_vt, c_k, c_ke, c_chunk);
    advance_unit(true);
    begin_stage();
    issue_req(0); issue_req(1); issue_req(2); issue_req(3);
    begin_stage();                                   // a unit has at least two steps
    issue_req(0); issue_req(1); issue_req(2); issue_req(3);
    if (loader_wants_unit()) advance_unit(false);
    wait_vm<4>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    {
        const char* const rd0 = lds;
#pragma unroll
        for (int q = 0; q < TN; ++q) read_frag(0, 0, q, rd0);
#pragma unroll
        for (int q = 0; q < TN; ++q) read_frag(0, 1, q, rd0);
#pragma unroll
        for (int q = 0; q < TM; ++q) read_frag(0, 2, q, rd0);
#pragma unroll
        for (int q = 0; q < TM; ++q) read_frag(0, 3, q, rd0);
    }
    wait_vm<0>();
    wait_lgkm0();
    // steady steps in pairs (bh sets 0 -> 1 -> 0); every other step alone from set 0, bh' copied back behind it
    auto steady_pair = [&]() __attribute__((always_inline)) -> bool {
        return c_k + 2 < c_ke && pend_mt < 0 && l_ke - l_k >= 2;
    };
    for (;;) {
        while (steady_pair()) {
            step(0, true);
            step(1, true);
        }
        if (!step(0, false)) break;
#pragma unroll
        for (int j = 0; j < TN; ++j) fbh[0][j] = fbh[1][j];
    }
    __builtin_amdgcn_s_barrier();
    if (pend_mt >= 0) flush_stats();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

}  // namespace acimg
