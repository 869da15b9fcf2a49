// Internal descriptors of the fp32 implicit-GEMM kernels (igemm.hip).
#pragma once
#include "common.hpp"

namespace acimg {

// Epilogue: what happens to one accumulator element acc(m, n).
struct EpiParams {
    float* Y;
    int ldy;
    int M;       // valid rows
    int Nstore;  // valid columns
    const float* bias;
    const float* res;
    int ldres;
    const float* mask;
    int ldmask;
    int act;
    // scatter mode (kernel<=stride transposed conv / patch dgrad): row m is a pixel (img,h,w) of
    // an AH x AW grid, column n = (tap, ko); element lands at pixel (img, sc*h+r, sc*w+q), chan ko
    // scatter = 2 (round 4, sub-pixel form of a stride-2 transposed conv / data gradient with OVERLAPPING taps): as 1,
    // the landing pixel shifted by (oy0, ox0) and dropped when it falls outside YH x YW
    int scatter;
    int Ko, Sq, sc, YH, YW, AH, AW;
    int oy0, ox0;
    // per-row-block column statistics of the raw accumulator (batch-norm), [gridDim.x][2][stats_ld]
    float* stats;
    int stats_ld;
    int vec;  // every pointer / stride of the epilogue allows 16-byte accesses
    // fused tail of an identity bottleneck unit (igemm_split3dp_kernel, EPI 2): relu(acc * f_scale + f_shift + shortcut)
    // -> hi / lo fp16 planes in brick order; shortcut and output are split-format tensors of the output's shape
    const float* f_scale;
    const float* f_shift;
    const char* f_sc;
    char* f_out;
    unsigned f_sc_bytes, f_sc_lo, f_out_bytes, f_out_lo;
    // EPI 3 (projection shortcut): f_sc is the raw fp32 [M][Nstore] output of the shortcut conv, normalised by these
    const float* f_scale2;
    const float* f_shift2;
};

// C[m][n] = sum_k A(m,k) * B(k,n);  A gathered from an NHWC tensor (im2col on the fly).
struct IgemmParams {
    // A operand
    const float* A;
    int H, W, C, lda;
    int OH, OW;
    int R, S, stride, pad_t, pad_l;
    int M;
    int rowrun;  // 1: a K segment is a whole kernel row r (S*C contiguous floats), 0: one tap
    int L;       // segment length in floats
    int cps;     // BK-chunks per segment
    int kiters;  // total K iterations = nseg * cps
    const float* a_scale;
    const float* a_shift;
    int a_relu;
    // B operand
    const float* B;
    int ldb;
    long tap_stride;  // NT: distance between taps
    int flip;         // NT: use tap (ntaps-1-tap)
    int ntaps;
    int Ngemm;  // number of GEMM columns
    int Nld;    // NN: valid floats per B row (multiple of 4)
    unsigned a_bytes, b_bytes;  // extents for the buffer resource descriptors
    unsigned a_lo_off;          // split3p: byte distance from the hi plane to the lo plane of A
    unsigned b_brick;           // split3p: byte offset of the weight image in LDS-tile order (0 = row-major planes only)
    // XCD-aware tile rasterisation (igemm_split3d_kernel, 1-D grid of ras_tiles_m * ras_tiles_n workgroups):
    // workgroup L runs on XCD L % 8; every XCD walks its own contiguous share of the tile order
    // (bands of ras_gm row tiles) x (groups of ras_gn column tiles) x (rows) x (columns), so the ~64 tiles
    // resident on an XCD at a time form a ras_gm x ras_gn rectangle that shares A and B tiles through its L2
    int ras_tiles_m, ras_tiles_n, ras_gm, ras_gn;
    // tail split of igemm_split3d_kernel: workgroups >= ts_whole handle one of ts_s K ranges of a tile of the
    // last partial round; partial sums in ts_partial[tile - ts_whole][range][BM*BN], tickets in ts_counters
    int ts_whole, ts_s;
    float* ts_partial;
    int* ts_counters;
    // split-K
    int splits;
    float* slab;  // [splits][M][slab_ld]
    int slab_ld;
    EpiParams e;
};

// dW[kk][n] = sum_m A(m,kk) * G[m][n]   (kk = (tap, c) gathered as in IgemmParams)
struct WgradParams {
    const float* X;
    int H, W, C, ldx;
    int OH, OW;
    int R, S, stride, pad_t, pad_l;
    int M;      // reduction length N*OH*OW
    int KK;     // R*S*C rows of dW
    const float* G;
    int ldg;
    int Ngemm;  // columns
    int Nld;    // valid floats per G row (multiple of 4)
    int splits;
    int rows_per_split;  // multiple of BKR
    float* out;          // slab [splits][KK][ldo] when splits>1 else dW
    int ldo;
    // optional fused bias gradient: an implicit all-ones im2col column at row KK, i.e.
    // db[n] = sum_m G[m][n]; written to db_out[blockIdx.z * ldo + n] (slab when splits > 1)
    float* db_out;
    // x' = relu(x * a_scale[c] + a_shift[c]) on load (the producer's deferred batch norm; zero padding after the affine):
    // honoured by the halo weight-gradient kernel only - launch_wgrad refuses it on every other path
    const float* a_scale; const float* a_shift; int a_relu;
};

}  // namespace acimg
