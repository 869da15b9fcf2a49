// fp32 implicit-GEMM kernel, version 2 (gfx950).  See igemm.hip for the op mapping.
//
//   * global loads are branch-free `buffer_load_dwordx4` through a resource descriptor: padding taps,
//     rows past M and columns past N get an out-of-range offset and the hardware returns zeros;
//   * K steps over (tap, 32-channel chunk); when C % 32 == 0 (template CAL) the tap/chunk counters
//     are wave-uniform scalars advanced incrementally, no per-thread integer division in the loop;
//   * LDS is double buffered: tile t+1 is written to the other buffer after the MFMAs of tile t and
//     tile t+2's global loads are issued right after, so there is ONE barrier per K step and every
//     global load has a full MFMA phase to land;
//   * MFMA operand roles are swapped (weights in the A slot, activations in the B slot): the 4
//     accumulator registers of a lane are then 4 CONSECUTIVE OUTPUT CHANNELS of one pixel, so the
//     epilogue is one 16-byte store (and 16-byte bias / residual / mask loads) per 16x16 tile.
#pragma once
#include "igemm.hpp"

namespace acimg {

using u32x4 = __attribute__((__vector_size__(4 * sizeof(unsigned int)))) unsigned int;

__device__ __forceinline__ float4 buf_load4(__amdgpu_buffer_rsrc_t rsrc, unsigned voff) {
    return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 0));
}
constexpr unsigned OOB = 0x80000000u;  // >= num_records of every descriptor (tensors < 2 GiB)

// ---- epilogue -------------------------------------------------------------------------------
// row m -> linear output pixel; (oh, ow): where tap (0, 0) of the row lands (scatter modes)
__device__ __forceinline__ long epi_row_pix(const EpiParams& e, int m, int& oh, int& ow) {
    oh = 0;
    ow = 0;
    if (!e.scatter) return m;
    const int hw = e.AH * e.AW;
    const int img = m / hw;
    const int rem = m - img * hw;
    const int h = rem / e.AW;
    const int w = rem - h * e.AW;
    oh = e.sc * h + e.oy0;
    ow = e.sc * w + e.ox0;
    return ((long)img * e.YH + oh) * e.YW + ow;
}
__device__ __forceinline__ long epi_row_pix(const EpiParams& e, int m) {
    int oh, ow;
    return epi_row_pix(e, m, oh, ow);
}
// column n -> channel, pixel offset of its tap, and the tap (r, q) itself
__device__ __forceinline__ void epi_col(const EpiParams& e, int n, int& cn, int& pixoff, int& r, int& q) {
    r = 0;
    q = 0;
    if (!e.scatter) {
        cn = n;
        pixoff = 0;
    } else {
        const int t = n / e.Ko;
        cn = n - t * e.Ko;
        r = t / e.Sq;
        q = t - r * e.Sq;
        pixoff = r * e.YW + q;
    }
}
__device__ __forceinline__ void epi_col(const EpiParams& e, int n, int& cn, int& pixoff) {
    int r, q;
    epi_col(e, n, cn, pixoff, r, q);
}
// scatter = 2: does tap (r, q) of a row whose tap (0, 0) lands at (oh, ow) fall inside the output?
__device__ __forceinline__ bool epi_lands(const EpiParams& e, int oh, int ow, int r, int q) {
    return e.scatter != 2 || ((unsigned)(oh + r) < (unsigned)e.YH && (unsigned)(ow + q) < (unsigned)e.YW);
}
__device__ __forceinline__ float epi_act(float v, int act) {
    if (act == ACIMG_ACT_RELU) return fmaxf(v, 0.f);
    if (act == ACIMG_ACT_SIGMOID) return 1.f / (1.f + expf(-v));
    return v;
}
__device__ __forceinline__ void epi_store(const EpiParams& e, long opix, int cn, float v) {
    if (e.bias) v += e.bias[cn];
    if (e.res) v += e.res[opix * e.ldres + cn];
    v = epi_act(v, e.act);
    if (e.mask && !(e.mask[opix * e.ldmask + cn] > 0.f)) v = 0.f;
    e.Y[opix * e.ldy + cn] = v;
}
// 4 consecutive channels cn..cn+3 of one output pixel; `vec` = every pointer/stride is 16-byte friendly
__device__ __forceinline__ void epi_store4(const EpiParams& e, long opix, int cn, int nvalid, float4 v) {
    if (e.vec && nvalid == 4) {
        if (e.bias) {
            const float4 b = *reinterpret_cast<const float4*>(e.bias + cn);
            v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
        }
        if (e.res) {
            const float4 r = *reinterpret_cast<const float4*>(e.res + opix * e.ldres + cn);
            v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
        }
        if (e.act != ACIMG_ACT_NONE) {
            v.x = epi_act(v.x, e.act); v.y = epi_act(v.y, e.act);
            v.z = epi_act(v.z, e.act); v.w = epi_act(v.w, e.act);
        }
        if (e.mask) {
            const float4 k = *reinterpret_cast<const float4*>(e.mask + opix * e.ldmask + cn);
            v.x = k.x > 0.f ? v.x : 0.f; v.y = k.y > 0.f ? v.y : 0.f;
            v.z = k.z > 0.f ? v.z : 0.f; v.w = k.w > 0.f ? v.w : 0.f;
        }
        *reinterpret_cast<float4*>(e.Y + opix * e.ldy + cn) = v;
    } else {
        const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (k < nvalid) epi_store(e, opix, cn + k, vv[k]);
    }
}

// sum over the 16 lanes of a DPP row (= the 16 pixel rows of an MFMA accumulator fragment), result in every lane:
// quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror - four v_add_f32 with a DPP operand, no LDS
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
    return v;
}

// ---- shared epilogue: lane holds channels n4..n4+3 of pixel m for each (tm, tn) -------------------
template <int BM, int BN, int WGM, int WGN, int NTHR, int TM, int TN>
__device__ __forceinline__ void igemm_epilogue(const IgemmParams& p, f32x4 (&acc)[TM][TN], float* smem, int m0,
                                               int n0, int wm, int wn, int li, int g, int tid,
                                               int stat_row = -1, bool stats_only = false, bool merged = false) {
    constexpr int WTM = BM / WGM, WTN = BN / WGN;
    if (stat_row < 0) stat_row = blockIdx.x;
    if (p.splits > 1 && !merged) {
        float* slab = p.slab + (long)blockIdx.z * p.M * p.slab_ld;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = m0 + wm * WTM + i * 16 + li;
            if (m >= p.M) continue;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n4 = n0 + wn * WTN + j * 16 + 4 * g;
                if (n4 < p.slab_ld)
                    *reinterpret_cast<float4*>(slab + (long)m * p.slab_ld + n4) =
                        make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
            }
        }
        return;
    }

    const EpiParams& e = p.e;
    int cn[TN], pixoff[TN], tr[TN], tq[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) epi_col(e, n0 + wn * WTN + j * 16 + 4 * g, cn[j], pixoff[j], tr[j], tq[j]);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m0 + wm * WTM + i * 16 + li;
        if (m >= e.M || stats_only) continue;
        int oh, ow;
        const long rp = epi_row_pix(e, m, oh, ow);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n4 = n0 + wn * WTN + j * 16 + 4 * g;
            const int nvalid = e.Nstore - n4;
            if (nvalid > 0 && epi_lands(e, oh, ow, tr[j], tq[j]))
                epi_store4(e, rp + pixoff[j], cn[j], nvalid < 4 ? nvalid : 4,
                           make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]));
        }
    }

    if (e.stats) {
        // per-channel sum / sum of squares of the raw accumulators over this block's rows (rows past
        // M were zero-filled on load and add nothing): reduce over the 16 pixel lanes, then over waves.
        // With a bias (tf.layers.conv2d + batch_normalization: the normalised tensor is conv + bias) the
        // statistics are those of acc + bias over the valid rows.
        __syncthreads();
        float* red = smem;  // [WGM][2][BN]
        const bool with_bias = e.bias != nullptr;
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                float s1 = 0.f, s2 = 0.f;
                float bv = 0.f;
                if (with_bias) {
                    const int n = n0 + wn * WTN + j * 16 + 4 * g + rg;
                    bv = n < e.Nstore ? e.bias[n] : 0.f;
                }
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    float v = acc[i][j][rg];
                    if (with_bias) v = (m0 + wm * WTM + i * 16 + li < e.M) ? v + bv : 0.f;
                    s1 += v;
                    s2 += v * v;
                }
                s1 = row16_sum(s1);      // the 16 pixel lanes of a fragment are one DPP row: four v_add_f32 with a DPP
                s2 = row16_sum(s2);      // operand each (the __shfl_xor butterfly this replaces went through the LDS crossbar)
                if (li == 0) {
                    const int col = wn * WTN + j * 16 + 4 * g + rg;
                    red[(wm * 2 + 0) * BN + col] = s1;
                    red[(wm * 2 + 1) * BN + col] = s2;
                }
            }
        __syncthreads();
        for (int idx = tid; idx < 2 * BN; idx += NTHR) {
            const int which = idx / BN, col = idx - which * BN;
            const int n = n0 + col;
            if (n < e.stats_ld) {
                float s = 0.f;
#pragma unroll
                for (int w = 0; w < WGM; ++w) s += red[(w * 2 + which) * BN + col];
                e.stats[((long)stat_row * 2 + which) * e.stats_ld + n] = s;
            }
        }
    }
}

// ---- main kernel ------------------------------------------------------------------------------
template <int BM, int BN, int WGM, int WGN, int NTHR, bool B_NT, bool CAL>
__global__ __launch_bounds__(NTHR) void igemm_f32_kernel(const IgemmParams p) {
    constexpr int BK = 32;
    constexpr int LDA_S = BK + 4;
    constexpr int LDB_S = B_NT ? (BK + 4) : (BN + 4);
    constexpr int A_ELEMS = BM * LDA_S;
    constexpr int B_ELEMS = B_NT ? BN * LDB_S : BK * LDB_S;
    constexpr int STAGE = A_ELEMS + B_ELEMS;
    constexpr int WTM = BM / WGM, WTN = BN / WGN;
    constexpr int TM = WTM / 16, TN = WTN / 16;
    constexpr int KQ = BK / 4;
    constexpr int RPP = NTHR / KQ;
    constexpr int NA = (BM + RPP - 1) / RPP;
    constexpr int NBT = (BN + RPP - 1) / RPP;
    constexpr int NQ = BN / 4;
    constexpr int RPPB = NTHR / NQ;
    constexpr int NBN = (BK + RPPB - 1) / RPPB;
    constexpr int NB = B_NT ? NBT : NBN;
    static_assert(WGM * WGN * 64 == NTHR, "one wave per (wm, wn)");

    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WGN, wn = wid % WGN;
    const int li = lane & 15, g = lane >> 4;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;

    const __amdgpu_buffer_rsrc_t rsA =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.A), 0, p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.B), 0, p.b_bytes, 0x00020000);

    int it_begin = 0, it_end = p.kiters;
    if (p.splits > 1) {
        const int per = (p.kiters + p.splits - 1) / p.splits;
        it_begin = blockIdx.z * per;
        it_end = min(p.kiters, it_begin + per);
    }

    // ---- per-thread A rows ---------------------------------------------------------------------
    const int kq = tid % KQ;
    const int arow0 = tid / KQ;
    int a_off[NA], a_ih0[NA], a_iw0[NA];
    {
        // (img, oh, ow) of the first row by division, of the following rows by carrying RPP pixels forward:
        // few-channel layers run 3 K steps per workgroup, 2*NA integer divisions would outweigh them
        const int ohw = p.OH * p.OW;
        int img = (m0 + arow0) / ohw;
        int rem0 = (m0 + arow0) - img * ohw;
        int oh = rem0 / p.OW;
        int ow = rem0 - oh * p.OW;
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            const int row = arow0 + j * RPP;
            const int m = m0 + row;
            if (j > 0) {
                ow += RPP;
                while (ow >= p.OW) {
                    ow -= p.OW;
                    if (++oh == p.OH) {
                        oh = 0;
                        ++img;
                    }
                }
            }
            if (row < BM && m < p.M) {
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

    // K-step state: `nit` = next iteration to load
    int nit = it_begin;
    int st_r = 0, st_s = 0, st_c0 = 0;  // CAL: wave-uniform (tap, channel chunk)
    if constexpr (CAL) {
        const int cpt = p.C >> 5;
        const int tap = it_begin / cpt;
        st_c0 = (it_begin - tap * cpt) << 5;
        st_r = tap / p.S;
        st_s = tap - st_r * p.S;
    }

    float4 ra[NA], rb[NB];
    float4 ld_sc4 = make_float4(1.f, 1.f, 1.f, 1.f), ld_sh4 = make_float4(0.f, 0.f, 0.f, 0.f);
    unsigned ld_ok = 0;     // bit j: element j of the parked A tile is a real (not padding / tail) element

    auto load_tiles = [&]() {
        int r, s, c, seg = 0, chunk = 0;
        bool validk = true;
        if constexpr (CAL) {
            r = st_r;
            s = st_s;
            c = st_c0 + kq * 4;
        } else {
            seg = nit / p.cps;
            chunk = nit - seg * p.cps;
            const int pp = chunk * BK + kq * 4;
            validk = pp < p.L;
            if (p.rowrun) {
                r = seg;
                s = pp / p.C;
                c = pp - s * p.C;
            } else {
                r = seg / p.S;
                s = seg - r * p.S;
                c = pp;
            }
        }
        // The deferred-BN affine (+ ReLU) of the A operand is applied in store_tiles, NOT here: touching a loaded
        // value inside this loop makes the compiler wait for each load before the next is issued (a full memory
        // round trip per load); here the loads only leave, with the scale / shift of their channel chunk and the
        // padding flags parked beside them.
        if (p.a_scale != nullptr && validk) {
            ld_sc4 = *reinterpret_cast<const float4*>(p.a_scale + c);
            ld_sh4 = *reinterpret_cast<const float4*>(p.a_shift + c);
        }
        const int tapoff = ((r * p.W + s) * p.lda + c) * 4;
        ld_ok = 0;
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            const int ih = a_ih0[j] + r, iw = a_iw0[j] + s;
            const bool ok = validk && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
            ra[j] = buf_load4(rsA, ok ? (unsigned)(a_off[j] + tapoff) : OOB);      // out of range -> zeros
            ld_ok |= ok ? (1u << j) : 0u;
        }
        if constexpr (B_NT) {
            int tap;
            if constexpr (CAL) tap = r * p.S + s;
            else tap = p.rowrun ? (seg * p.S + s) : seg;
            const int tapb = p.flip ? (p.ntaps - 1 - tap) : tap;
            const int base = tapb * (int)p.tap_stride + c;
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int nrow = arow0 + j * RPP;
                const int n = n0 + nrow;
                const bool ok = validk && nrow < BN && n < p.Ngemm;
                rb[j] = buf_load4(rsB, ok ? (unsigned)((base + n * p.ldb) * 4) : OOB);
            }
        } else {
            const int nq = tid % NQ;
            const int krow0 = tid / NQ;
            const int n = n0 + nq * 4;
            int kb0, plim;
            if constexpr (CAL) {
                kb0 = (r * p.S + s) * p.C + st_c0;
                plim = BK;
            } else {
                kb0 = seg * p.L + chunk * BK;
                plim = p.L - chunk * BK;
            }
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int krow = krow0 + j * RPPB;
                const bool ok = krow < BK && krow < plim && n < p.Nld;
                rb[j] = buf_load4(rsB, ok ? (unsigned)(((kb0 + krow) * p.ldb + n) * 4) : OOB);
            }
        }
        // advance
        ++nit;
        if constexpr (CAL) {
            st_c0 += BK;
            if (st_c0 == p.C) {
                st_c0 = 0;
                if (++st_s == p.S) {
                    st_s = 0;
                    ++st_r;
                }
            }
        }
    };

    auto store_tiles = [&](int buf) {
        float* As = smem + buf * STAGE;
        float* Bs = As + A_ELEMS;
        if (p.a_scale != nullptr) {
#pragma unroll
            for (int j = 0; j < NA; ++j) {
                float4 v = ra[j];
                v.x = v.x * ld_sc4.x + ld_sh4.x;
                v.y = v.y * ld_sc4.y + ld_sh4.y;
                v.z = v.z * ld_sc4.z + ld_sh4.z;
                v.w = v.w * ld_sc4.w + ld_sh4.w;
                if (p.a_relu) {
                    v.x = fmaxf(v.x, 0.f);
                    v.y = fmaxf(v.y, 0.f);
                    v.z = fmaxf(v.z, 0.f);
                    v.w = fmaxf(v.w, 0.f);
                }
                if (!((ld_ok >> j) & 1u)) v = make_float4(0.f, 0.f, 0.f, 0.f);  // zero padding AFTER the affine
                ra[j] = v;
            }
        }
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            const int row = arow0 + j * RPP;
            if (row < BM) *reinterpret_cast<float4*>(&As[row * LDA_S + kq * 4]) = ra[j];
        }
        if constexpr (B_NT) {
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int nrow = arow0 + j * RPP;
                if (nrow < BN) *reinterpret_cast<float4*>(&Bs[nrow * LDB_S + kq * 4]) = rb[j];
            }
        } else {
            const int nq = tid % NQ;
            const int krow0 = tid / NQ;
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int krow = krow0 + j * RPPB;
                if (krow < BK) *reinterpret_cast<float4*>(&Bs[krow * LDB_S + nq * 4]) = rb[j];
            }
        }
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (it_begin < it_end) {
        load_tiles();
        store_tiles(0);
        if (nit < it_end) load_tiles();
    }
    __syncthreads();

    int cur = 0;
    for (int it = it_begin; it < it_end; ++it) {
        const float* As = smem + cur * STAGE;
        const float* Bs = As + A_ELEMS;
#pragma unroll
        for (int kk = 0; kk < BK / 16; ++kk) {
            float4 a4[TM];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                a4[i] = *reinterpret_cast<const float4*>(&As[(wm * WTM + i * 16 + li) * LDA_S + kk * 16 + 4 * g]);
            float bf[TN][4];
            if constexpr (B_NT) {
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const float4 b4 = *reinterpret_cast<const float4*>(
                        &Bs[(wn * WTN + j * 16 + li) * LDB_S + kk * 16 + 4 * g]);
                    bf[j][0] = b4.x;
                    bf[j][1] = b4.y;
                    bf[j][2] = b4.z;
                    bf[j][3] = b4.w;
                }
            } else {
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        bf[j][t] = Bs[(kk * 16 + 4 * g + t) * LDB_S + wn * WTN + j * 16 + li];
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const float av = t == 0 ? a4[i].x : (t == 1 ? a4[i].y : (t == 2 ? a4[i].z : a4[i].w));
#pragma unroll
                    for (int j = 0; j < TN; ++j)  // rows of D = output channels, columns = pixels
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[j][t], av, acc[i][j], 0, 0, 0);
                }
            }
        }
        if (it + 1 < it_end) {
            store_tiles(cur ^ 1);          // tile it+1 (loaded one step ago) -> the other buffer
            if (nit < it_end) load_tiles();  // tile it+2
        }
        __syncthreads();
        cur ^= 1;
    }

    if (p.splits > 1 && p.ts_counters != nullptr) {
        // split-K without a reduce launch (same hand-off as the trunk kernel's tail split): every K range parks its
        // partial tile (write-through sc1 stores: visible to the other XCDs), takes a ticket; the last arriver adds
        // the ranges IN RANGE ORDER (its own included, from memory: the result does not depend on who came last)
        // and runs the ordinary epilogue.  Nobody waits.
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        const int tl = blockIdx.y * gridDim.x + blockIdx.x;
        float* const slot0 = p.slab + (long)tl * p.splits * (BM * BN);
        const __amdgpu_buffer_rsrc_t rsP =
            __builtin_amdgcn_make_buffer_rsrc(slot0 + (long)blockIdx.z * (BM * BN), 0, BM * BN * 4, 0x00020000);
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
            const int last = ticket == p.splits - 1;
            if (last) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_store(p.ts_counters + tl, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            *flag = last;
        }
        __syncthreads();
        if (!*flag) return;
        __syncthreads();                           // everyone has read the flag before the epilogue reuses the LDS
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int c = 0; c < p.splits; ++c) {
            const __amdgpu_buffer_rsrc_t rsQ =
                __builtin_amdgcn_make_buffer_rsrc(slot0 + (long)c * (BM * BN), 0, BM * BN * 4, 0x00020000);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)     // sc1 loads: never a stale L1 / L2 copy
                    acc[i][j] += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                               rsQ, ((i * TN + j) * NTHR + tid) * 16, 0, 16));
        }
        igemm_epilogue<BM, BN, WGM, WGN, NTHR, TM, TN>(p, acc, smem, m0, n0, wm, wn, li, g, tid, -1, false, true);
        return;
    }
    igemm_epilogue<BM, BN, WGM, WGN, NTHR, TM, TN>(p, acc, smem, m0, n0, wm, wn, li, g, tid);
}

}  // namespace acimg
