/*
 * acimg.h — C ABI of libacimg.so: the MI355X (gfx950) kernels behind the acoustic-image
 * generation train step of IIT-PAVIS/Acoustic-Image-Generation.
 *
 * The reference has no FFI layer: every arithmetic op on its hot path is a TensorFlow-1.x op
 * call inside models/<name>.py and trainer/mfcctrainer.py.  Each entry point
 * below replaces one family of those op calls and cites the call site(s) it stands in for.
 *
 * Conventions (all entry points):
 *   - return 0 on success, a negative ACIMG_E* code otherwise; acimg_last_error() holds text;
 *     nothing is thrown across the boundary;
 *   - the CALLER owns every buffer (device pointers, 16-byte aligned), including workspace and the
 *     ticket words of the in-kernel reductions (explicit `tickets` / scratch arguments); the library
 *     allocates nothing.  Its only process-wide state is the tuning record of acimg_configure() (thirteen
 *     plain ints with compiled-in defaults, written by that call alone, never by a launch, and never
 *     read from the process environment) and the per-thread text of acimg_last_error();
 *   - all work is enqueued on `stream` (a hipStream_t passed as void*), no host sync;
 *   - tensors are NHWC float32; a tensor is (ptr, C, ld): C logical channels per pixel and
 *     ld >= C the pixel stride in floats, so producers can write straight into channel slices
 *     of a concat buffer (tf.concat never materialises).  Channel counts seen by the GEMM
 *     kernels are multiples of 4 (the host pads 3->4, 133->136, 145->148, 150->152, 266->268
 *     with zeros; padded weights rows/cols are zero and provably stay zero under Adam).
 *   - weights use TensorFlow's own layouts, padded as above:
 *       conv2d            HWIO  [R][S][Cin_p][Cout_p]
 *       conv2d_transpose        [R][S][Cout_p][Cin_p]
 *       dense                   [Kin_p][Nout_p]
 */
#ifndef ACIMG_H_
#define ACIMG_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ACIMG_VERSION 207

#define ACIMG_OK 0
#define ACIMG_EINVAL (-1)     /* bad descriptor / shape / alignment */
#define ACIMG_EWORKSPACE (-2) /* workspace too small */
#define ACIMG_ELAUNCH (-3)    /* hipLaunch / runtime error */

#define ACIMG_ACT_NONE 0
#define ACIMG_ACT_RELU 1
#define ACIMG_ACT_SIGMOID 2

int acimg_version(void);
/* copies the calling thread's last error text (NUL terminated) into buf */
int acimg_last_error(char* buf, size_t len);

/* ------------------------------------------------------------------------------------------
 * Convolution family.  One descriptor serves forward / data-gradient / weight-gradient.
 *   x : [N,H,W,C]   (pixel stride ldx)      y : [N,OH,OW,K] (pixel stride ldy)
 *   w : HWIO [R][S][C][ldw], ldw >= K
 * TF 'SAME' asymmetric padding is resolved by the host into pad_t/pad_l;
 * OH/OW are given explicitly.
 * ---------------------------------------------------------------------------------------- */
typedef struct AcimgConvDesc {
    int32_t N, H, W, C, ldx;
    int32_t K, ldy, OH, OW;
    int32_t R, S, stride, pad_t, pad_l;
    int32_t ldw;
    int32_t act; /* ACIMG_ACT_* applied by the forward epilogue */
} AcimgConvDesc;

/* Forward: y = act(conv(x', w) + bias), x' = relu(x*in_scale[c] + in_shift[c]) when in_scale
 * is given (deferred batch-norm of the producer, zero padding applied AFTER the affine), else x.
 * stats (optional, [grid_m][2][ldw] floats, grid_m = acimg_conv2d_stats_rows(d)) receives per
 * row-block partial sums / sums of squares of the raw conv output for batch-norm statistics.
 * The row count belongs to THIS descriptor, `act` included (the few-channel 3x3 / stride-1 layers from 65536
 * pixels on run on an MFMA kernel that leaves one row per workgroup - 512 - when act is ACIMG_ACT_NONE, and on
 * the direct kernel with one row per 256 pixels otherwise): size the buffer from the descriptor that is launched.
 * Replaces: tf.layers.conv2d   models/unet_acresnet.py:159-168,173-182,82,89-94
 *           slim layers.conv2d / resnet_utils.conv2d_same   models/resnet50.py:109-121,205-209 */
int acimg_conv2d_fwd(const AcimgConvDesc* d, const float* x, const float* w, const float* bias,
                     float* y, const float* in_scale, const float* in_shift, int in_relu,
                     float* stats, void* ws, size_t ws_bytes, void* tickets, void* stream);
int acimg_conv2d_stats_rows(const AcimgConvDesc* d);
/* out[3] = {BM, BN, split-K factor} the forward launch will use (profiling / roofline bookkeeping) */
int acimg_conv2d_fwd_tiling(const AcimgConvDesc* d, int* out);
/* `tickets` argument of acimg_conv2d_fwd / _dgrad and acimg_deconv_fwd / _dgrad: NULL, or ACIMG_TICKET_WORDS ints of
 * ZEROED device memory (16-byte aligned) owned by the caller and written by nothing else.  With it, a split-K launch
 * combines its K ranges inside the kernel — one ticket per output tile, the last arriver adds the ranges in range
 * order and runs the epilogue — instead of through a separate reduce launch; every launch leaves the words at zero and
 * results are bit-identical to the reduce-launch path.  Launches sharing one ticket block must be ordered against each
 * other (one stream, as the recorded plans are): use one block per stream. */
#define ACIMG_TICKET_WORDS 4096

/* Tuning record of the launch heuristics.  acimg_config_default() fills in the compiled-in values;
 * acimg_configure() installs a record (validate, copy).  Call it before the first launch, from one thread; launches
 * only read it.  The library never reads the process environment (the Python host maps ACIMG_* variables onto this
 * call once, at load time: acimg/_lib.py). */
typedef struct AcimgConfig {
    int32_t splitk_cut;      /* f32 implicit GEMM: no K split at or above this many output tiles (320) */
    int32_t splitk_target;   /* ... otherwise split towards this many workgroups (768) */
    int32_t splitk_handoff;  /* 1: combine K ranges in-kernel when `tickets` is given; 0: always the reduce launch */
    int32_t wgrad_minpix;    /* weight gradients: at least this many pixels per slab (128) */
    int32_t wgrad_halo;      /* 1: few-channel weight gradients on the halo kernel */
    int32_t split3_tile_bm;  /* 0 = per-shape choice; else force the split-MFMA tile (experiments): 128x128, 64x128, */
    int32_t split3_tile_bn;  /*     128x64 */
    int32_t tail_split;      /* 1: trunk kernel cuts the tiles of the last partial round into K ranges */
    int32_t tail_s;          /* 0 = cost model; else force that many K ranges (experiments) */
    int32_t trunk_persistent;/* 128x128 trunk convs on the persistent kernel (a workgroup walks a tile list; the next tile's
                                first loads overlap the current tile's last K step and output stores): 0 never, 1 where it
                                was measured to pay (short-K, multi-round layers), 2 always */
    int32_t trunk_bk;        /* K-step depth of the persistent kernel: 0 / 32 (a 64-deep step, one workgroup per CU, was
                                measured slower in round 2 and removed; the field keeps the record's layout) */
    int32_t trunk_stagger;   /* persistent kernel: start the second half of the grid this many percent of a tile's
                                estimated time late (0 = together) */
    int32_t trunk_dma_pos;   /* persistent kernel: a K step's operand requests 0 = in one burst after the step barrier,
                                1 = spread under the MFMA block (B after the first sweep, A after the second)
                                (requests TWO steps ahead - a second barrier after the fragment reads frees the stage
                                early - were measured in round 3: 2.5 % slower over the trunk, removed) */
    int32_t trunk_ring;      /* 128-column trunk convs on the RING kernel (one workgroup per CU, three LDS slots, fragments
                                double buffered in registers, requests spread between the MFMAs): 0 never, 1 where it was
                                measured to pay, 2 always */
    int32_t trunk_ring_bm;   /* ring kernel's tile rows: 0 = per shape, else 128 or 256 (experiments) */
    int32_t trunk_halo;      /* 3x3 / stride-1 trunk convs on the HALO kernel (one patch of the activation planes staged per
                                channel chunk, all nine taps formed from it in LDS): 0 never, 1 where it was measured to
                                pay, 2 wherever it applies */
} AcimgConfig;
int acimg_config_default(AcimgConfig* cfg);
int acimg_configure(const AcimgConfig* cfg);
size_t acimg_conv2d_fwd_workspace(const AcimgConvDesc* d);

/* f16x3 ("split fp16") forward convolution for frozen weights (the ResNet-50 trunk): x and w are fp32,
 * each is split into hi = f16(v), lo = f16(v - hi) (22 mantissa bits) and every product is
 * ah*wh + ah*wl + al*wh on the fp16 matrix cores with fp32 accumulation (~2^-22 relative per product:
 * fp32-class results, same 1e-3 parity bar) at 3/16 of the exact-f32 MFMA cost.  Exact power-of-two
 * scaling (weights x2^10, activations x2^-2, accumulators x2^-8) keeps everything in fp16's range for
 * |w| < 63 and |x| < 2.6e5.  Needs C % 32 == 0.  `wsplit` (caller-owned,
 * acimg_conv2d_split3_weight_bytes(d) bytes) is filled by acimg_conv2d_split3_prepare from the HWIO fp32
 * kernel: [hi|lo][ldw][R*S*C] fp16, followed - for the pre-split entry points acimg_conv2d_fwd_split3p / _split1p - by
 * the same weights in LDS-tile order (per 128 output channels and 32-deep K step the two 8 KiB plane images
 * [row][64 bytes], 16-byte groups swizzled as in the activation bricks below).  Semantics otherwise as acimg_conv2d_fwd (deferred BN on load, raw
 * output + statistics partials of acimg_conv2d_fwd_split3_stats_rows(d) rows; optional bias + d->act; no
 * split-K).
 * Row-run view (this entry only): with S == 1 and pad_l == 0, ldx < C is accepted when C % ldx == 0 — the C
 * "channels" of a tap are then the C / ldx consecutive pixels of an input row starting at ow*stride (windows of
 * neighbouring outputs overlap; (OW-1)*stride + C/ldx <= W).  The 7x7/2 stem on a zero-padded 4-channel frame is
 * such a conv: R = 7, S = 1, C = 32 = 7 pixels x 4 channels + one pixel of zero weights, ldx = 4.
 * Replaces: slim layers.conv2d / conv2d_same in the trunk, models/resnet50.py:109-121. */
size_t acimg_conv2d_split3_weight_bytes(const AcimgConvDesc* d);
int acimg_conv2d_split3_prepare(const AcimgConvDesc* d, const float* w, void* wsplit, void* stream);
/* Up to 16 kernels in ONE launch (a model that trains re-splits its kernels every step): mode[i] = 0 -> the forward
 * image of acimg_conv2d_split3_prepare, 1 -> the data-gradient image of acimg_conv2d_split3_prepare_dgrad. */
int acimg_conv2d_split3_prepare_multi(int n, const AcimgConvDesc* const* descs, const float* const* w, void* const* out,
                                      const int* mode, void* stream);
int acimg_conv2d_fwd_split3_stats_rows(const AcimgConvDesc* d);
/* out[3] = {BM, BN, kernel of acimg_conv2d_fwd_split3p for this shape: 0 one tile per workgroup, 1 persistent, 2 ring} */
int acimg_conv2d_fwd_split3_tiling(const AcimgConvDesc* d, int* out);
int acimg_conv2d_fwd_split3(const AcimgConvDesc* d, const float* x, const void* wsplit, const float* bias,
                            float* y, const float* in_scale, const float* in_shift, int in_relu, float* stats,
                            void* stream);
/* Data gradient of a stride-1 conv on the same structure with a bf16 hi/lo split (gradients ~1e-7 are
 * outside fp16's range; bf16 keeps fp32's range, 16 mantissa bits, no scaling): a forward conv of gy with
 * the flipped + transposed kernel prepared by acimg_conv2d_split3_prepare_dgrad ([hi|lo][C][R*S*K] bf16,
 * acimg_conv2d_split3_dgrad_weight_bytes(d) bytes).  Needs K % 32 == 0.  residual / mask / lddx as in
 * acimg_conv2d_dgrad. */
size_t acimg_conv2d_split3_dgrad_weight_bytes(const AcimgConvDesc* d);
int acimg_conv2d_split3_prepare_dgrad(const AcimgConvDesc* d, const float* w, void* wsplit, void* stream);
int acimg_conv2d_dgrad_split3(const AcimgConvDesc* d, const float* gy, int ldgy, const void* wsplit_t, float* dx,
                              int lddx, const float* residual, int ldres, const float* mask, int ldmask,
                              void* stream);

/* bf16 OPERAND ARITHMETIC (BASELINE configs[1], "bf16"): the same three convolutions with both GEMM operands ROUNDED to
 * bf16 and multiplied once per product (v_mfma_f32_16x16x32_bf16), fp32 accumulation; tensors in HBM, bias, batch
 * statistics, losses and the optimizer stay fp32 (the mixed-precision recipe of a bf16 network: what tf's
 * auto_mixed_precision / a bf16 cast at the conv inputs computes).  Same arguments, workspaces and epilogues as the
 * split3 entry points they mirror; 1/3 of the MFMAs, half the LDS traffic, results carry bf16's 8 mantissa bits per
 * operand (parity against the oracle WITH THE SAME ROUNDING: tests/test_unet_vae_gpu.py).
 *   acimg_conv2d_bf16_prepare : forward weight image (bf16 [hi|lo][ldw][R*S*C], acimg_conv2d_split3_weight_bytes)
 *   acimg_conv2d_fwd_bf16     : mirrors acimg_conv2d_fwd_split3
 *   acimg_conv2d_dgrad_bf16   : mirrors acimg_conv2d_dgrad_split3 (weights from acimg_conv2d_split3_prepare_dgrad)
 *   acimg_conv2d_wgrad_bf16   : mirrors acimg_conv2d_wgrad_split3
 * acimg_conv2d_split3_prepare_multi: mode 2 = the bf16 forward image.
 * Replaces: the conv2d calls of models/unet_architecture.py:159-213 under a bf16 policy. */
int acimg_conv2d_bf16_prepare(const AcimgConvDesc* d, const float* w, void* wsplit, void* stream);
int acimg_conv2d_fwd_bf16(const AcimgConvDesc* d, const float* x, const void* wsplit, const float* bias,
                          float* y, const float* in_scale, const float* in_shift, int in_relu, float* stats,
                          void* stream);
int acimg_conv2d_dgrad_bf16(const AcimgConvDesc* d, const float* gy, int ldgy, const void* wsplit_t, float* dx,
                            int lddx, const float* residual, int ldres, const float* mask, int ldmask,
                            void* stream);

/* Pre-split activation format: a tensor [rows][C] (C % 32 == 0, dense) is stored as TWO fp16 planes (hi at ptr, lo
 * `lo_off` bytes further, lo_off >= acimg_split_plane_bytes(rows, C), 16-byte aligned), hi = f16(v/4),
 * lo = f16(v/4 - hi): 22 mantissa bits in the same 4 bytes per element as fp32.  The elementwise producers below write
 * it (the BN affine + ReLU is already applied), and acimg_conv2d_fwd_split3p consumes it: both GEMM operands are then
 * plain 16-byte copies into LDS, so the K loop of the trunk convs carries no conversion / normalisation work at all.
 * A plane is laid out in LDS-TILE ORDER: ceil(rows / 16) x C / 32 bricks of 1 KiB, brick (row >> 4, c >> 5) at
 * ((row >> 4) * C / 32 + (c >> 5)) * 1024, inside it 16 rows (row & 15) of 64 bytes and the 16-byte group (c >> 3) & 3 of
 * a row at group ((c >> 3) ^ -(row >> 2)) & 3 - exactly the image the kernels keep in LDS, so an LDS-DMA request of the
 * K loop is one contiguous KiB (eight whole cache lines) instead of sixteen 64-byte pieces of sixteen rows.
 *   acimg_bn_relu_split:         planes = relu?(x*scale+shift)            (BN of a bottleneck conv, resnet50.py:109-121)
 *   acimg_bn_add_relu_split:     planes (+ optional fp32) = relu(a*sa+ta + shortcut); shortcut = b32*sb+tb
 *                                (projection) or the previous unit's planes (identity / subsample)   (resnet50.py:104-123)
 *   acimg_bn_relu_maxpool_split: planes = maxpool3x3/s2(relu(x*scale+shift))                        (resnet50.py:207-208) */
/* `ws` (optional, acimg_conv2d_fwd_split3p_workspace bytes, DEDICATED to these calls: its first 4 KiB are tile
 * tickets that must be zero before the first call and are left zero by every call) lets the kernel cut the tiles of
 * the last, partially filled round of workgroups into K ranges that meet in the workspace (deterministic: fixed
 * range order).  Without it every tile is computed by one workgroup. */
size_t acimg_conv2d_fwd_split3p_workspace(const AcimgConvDesc* d);
/* statistics rows acimg_conv2d_fwd_split3p writes for this shape under the current configuration (one per row tile of
 * the kernel it picks; acimg_conv2d_fwd_split1p keeps acimg_conv2d_fwd_split3_stats_rows) */
int acimg_conv2d_fwd_split3p_stats_rows(const AcimgConvDesc* d);
int acimg_conv2d_fwd_split3p(const AcimgConvDesc* d, const void* x_planes, size_t x_lo_off, const void* wsplit,
                             float* y, float* stats, void* ws, size_t ws_bytes, void* stream);
/* The same launch with fp16 OPERAND STORAGE (BASELINE configs[4]: "fp16 with fp32 loss accumulation"): only the hi
 * planes of activations and weights are fetched and multiplied - one fp16 MFMA per product instead of three, half the
 * operand bytes - with fp32 accumulation, fp32 batch-norm statistics and fp32 output.  Operands carry 11 significant
 * bits; everything else (arguments, tiling, statistics rows, workspace) is as for acimg_conv2d_fwd_split3p. */
int acimg_conv2d_fwd_split1p(const AcimgConvDesc* d, const void* x_planes, size_t x_lo_off, const void* wsplit,
                             float* y, float* stats, void* ws, size_t ws_bytes, void* stream);
/* The expanding 1x1 conv of an identity bottleneck unit in TWO PASSES, its raw output never stored (slim bottleneck,
 * models/resnet50.py:104-125: out = relu(BN(conv3(x)) + shortcut), batch statistics): the conv is short-K and 4x wide,
 * so its MFMA work is cheap and its output bytes are not.
 *   acimg_conv2d_fwd_split3p_stats: the K loop and the batch-norm partials only (rows as acimg_conv2d_fwd_split3p's);
 *   acimg_conv2d_fwd_split3p_tail:  the same tiles again with the layer's (scale, shift) from acimg_bn_finalize:
 *                                   relu(acc * scale + shift + shortcut) split into hi / lo planes in brick order;
 *                                   `sc_planes` is a split-format tensor of the output's shape (the unit's input),
 *                                   `out_planes` must not alias it.
 * Both need 128x128 tiles (>= 200 of them), K % 128 == 0, ldy == ldw == K, outputs < 2 GiB; same units in the same
 * order in both passes, so the statistics are those of exactly the values the second pass normalises.  Per output
 * element 4 B read + 4 B written instead of 4 written + 8 read + 4 written by conv + acimg_bn_add_relu_split. */
int acimg_conv2d_fwd_split3p_stats(const AcimgConvDesc* d, const void* x_planes, size_t x_lo_off, const void* wsplit,
                                   float* stats, void* ws, size_t ws_bytes, void* stream);
int acimg_conv2d_fwd_split3p_tail(const AcimgConvDesc* d, const void* x_planes, size_t x_lo_off, const void* wsplit,
                                  const float* scale, const float* shift, const void* sc_planes, size_t sc_lo_off,
                                  void* out_planes, size_t out_lo_off, void* ws, size_t ws_bytes, void* stream);
/* ... and of a unit with a PROJECTION shortcut (models/resnet50.py:112-118: shortcut = batch_norm(conv1x1(x))): `sc32` is the
 * raw fp32 [rows][K] output of the shortcut conv, (sc_scale, sc_shift) its affine from acimg_bn_finalize:
 * relu(acc * scale + shift + (sc32 * sc_scale + sc_shift)) -> split planes.  Same requirements as _tail. */
int acimg_conv2d_fwd_split3p_tail_proj(const AcimgConvDesc* d, const void* x_planes, size_t x_lo_off, const void* wsplit,
                                       const float* scale, const float* shift, const float* sc32, const float* sc_scale,
                                       const float* sc_shift, void* out_planes, size_t out_lo_off, void* ws, size_t ws_bytes,
                                       void* stream);
/* INPUT-SIDE batch-norm statistics of a 1x1 conv (round 4; replaces acimg_conv2d_fwd_split3p_stats + acimg_bn_finalize in
 * front of the fused tails above): for y = x w over `rows` pixels, sum_p y_n = w_n^T sum_p x and
 * sum_p y_n^2 = w_n^T (sum_p x x^T) w_n - the column sums and the C x C Gram matrix of the conv's INPUT (P C^2 MACs, a
 * quarter of the conv's, two fp16 MFMAs per product: a quadratic form only sees the symmetric part, hi hi^T + 2 hi lo^T).
 * Three launches on `stream` (csrc/gram.hip): Gram partials per pixel range (brick planes -> LDS by LDS-DMA, transposing
 * fragment reads), a deterministic reduce in double, and per 16 output channels the two forms by exact-f32 MFMA followed by
 * acimg_bn_finalize's arithmetic: `scale` / `shift` [K] for the conv's fused tail, the moving averages advanced (training
 * mode only; inference keeps acimg_bn_finalize with training = 0).
 *   x_planes / x_lo_off: the conv's input in split format [rows][C]; C = 64 or a multiple of 128 up to 512
 *   w: the conv's fp32 kernel [C][ldw] (TF layout [1, 1, C, K]); ws: acimg_gram_stats_workspace(rows, C) bytes, 16-byte aligned
 * Replaces: the moments slim batch_norm takes of conv3's output, models/resnet50.py:121-123 (is_training=True). */
size_t acimg_gram_stats_workspace(long rows, int C);
int acimg_gram_stats(const void* x_planes, size_t x_lo_off, long rows, int C, const float* w, int ldw, int K,
                     const float* gamma, const float* beta, float* moving_mean, float* moving_var, float decay, float eps,
                     float* scale, float* shift, void* ws, size_t ws_bytes, void* stream);
size_t acimg_split_plane_bytes(long rows, int C);
int acimg_bn_relu_split(const float* x, const float* scale, const float* shift, int relu, void* out,
                        size_t lo_off, long rows, int C, void* stream);
int acimg_bn_add_relu_split(const float* a, const float* sa, const float* ta, const float* b32,
                            const float* sb, const float* tb, const void* b_planes, size_t b_lo_off,
                            void* out_planes, size_t out_lo_off, float* out32, int N, int OH, int OW, int C,
                            int BH, int BW, int bstride, void* stream);
int acimg_bn_relu_maxpool_split(const float* x, const float* scale, const float* shift, void* out,
                                size_t lo_off, int N, int H, int W, int C, int OH, int OW, int pad_t, int pad_l,
                                void* stream);

/* Data gradient.  gy is the gradient w.r.t. the conv's PRE-activation output [N,OH,OW,K]
 * (pixel stride ldgy); dx = relu_mask(conv_T(gy, w) + residual): `residual` (optional, pixel
 * stride ldres) is another gradient flowing into x (fan-out), `mask` (optional, pixel stride
 * ldmask) is the saved post-ReLU activation x itself: dx is zeroed where mask <= 0, so dx is
 * the pre-activation gradient of the producing layer.  dx has pixel stride lddx (0 = d->ldx: the
 * gradient of a concat-slice input usually lives in its own, narrower buffer).  Geometries: stride 1 (direct),
 * stride == R == S with zero padding (non-overlapping patches, layer1/pool_2: scatter form), and any other
 * stride (the strided "pool" convs of models/unet_architecture.py:168-176, models/unet_sound.py:162-170) via a
 * zero-inserted copy of gy in the workspace followed by the stride-1 form.
 * Replaces: the Conv2DBackpropInput ops tf.gradients emits for the calls above
 *           (trainer/mfcctrainer.py:72-79). */
int acimg_conv2d_dgrad(const AcimgConvDesc* d, const float* gy, int ldgy, const float* w,
                       float* dx, int lddx, const float* residual, int ldres, const float* mask,
                       int ldmask, void* ws, size_t ws_bytes, void* tickets, void* stream);
size_t acimg_conv2d_dgrad_workspace(const AcimgConvDesc* d);

/* Weight + bias gradient: dw[R][S][C][ldw] = sum_pixels x (*) gy, db[k] = sum gy (db optional).
 * Replaces: Conv2DBackpropFilter / BiasAddGrad (trainer/mfcctrainer.py:72-79). */
int acimg_conv2d_wgrad(const AcimgConvDesc* d, const float* x, const float* gy, int ldgy,
                       float* dw, float* db, void* ws, size_t ws_bytes, void* stream);
size_t acimg_conv2d_wgrad_workspace(const AcimgConvDesc* d);
/* the same on the bf16x3 MFMA path: x and gy are split into bf16 hi/lo on the fly (gradients need fp32's
 * range), 3 MFMAs per product, fp32 accumulate; fragments come out of LDS through ds_read_b64_tr_b16
 * because the reduction runs over pixels.  Same workspace as acimg_conv2d_wgrad. */
int acimg_conv2d_wgrad_split3(const AcimgConvDesc* d, const float* x, const float* gy, int ldgy,
                              float* dw, float* db, void* ws, size_t ws_bytes, void* stream);
int acimg_conv2d_wgrad_bf16(const AcimgConvDesc* d, const float* x, const float* gy, int ldgy,
                            float* dw, float* db, void* ws, size_t ws_bytes, void* stream);
/* A batch-norm + ReLU between two convs without a pass of its own (round 4): the consumer applies the producer's
 * (scale, shift) and the ReLU while it stages its input tile - x' = relu(x * in_scale[c] + in_shift[c]) inside the
 * image, zero padding after the affine - so the normalised activation is never written.  The forward entries above
 * already carry in_scale / in_shift / in_relu; this is the weight gradient of the same layer with the same view of
 * its input.  precision: 0 = the acimg_conv2d_fwd / _wgrad arithmetic, 1 = _split3, 2 = _bf16.  Only the halo
 * kernels stage through registers and can do this: acimg_conv2d_affine_input_ok(d, precision) is 1 when BOTH the
 * forward and the weight gradient of this layer take the affine (3x3 / stride 1 / SAME, <= 32 output channels, from
 * 65536 pixels on); acimg_conv2d_wgrad_affine fails loudly on any other shape.  Workspace: acimg_conv2d_wgrad_workspace.
 * Replaces: the read side of tf.layers.batch_normalization + tf.nn.relu between two tf.layers.conv2d
 * (models/unet_architecture.py:55-60). */
int acimg_conv2d_affine_input_ok(const AcimgConvDesc* d, int precision);
int acimg_conv2d_wgrad_affine(const AcimgConvDesc* d, int precision, const float* x, const float* in_scale,
                              const float* in_shift, int in_relu, const float* gy, int ldgy, float* dw, float* db,
                              void* ws, size_t ws_bytes, void* stream);

/* Transposed convolution, VALID (TF output = in*stride + max(kernel - stride, 0)):
 *   x : [N,H,W,C] low-res input, y : [N,OH,OW,K], w : [R][S][K][ldw>=C].
 *   kernel <= stride in both directions (unet_acresnet.py:210-217): scatter form, gaps receive the bias;
 *   kernel >= stride with overlap (unet_architecture.py:72,78 kernel [2,3]; unet_sound.py:82,85 kernels [3,2],
 *   [3,3]): zero-inserted copy of x in the workspace, then a stride-1 full correlation with the flipped kernel.
 * Every input pixel writes a disjoint RxS patch; the remaining positions get the bias only.
 * Replaces: tf.layers.conv2d_transpose  models/unet_acresnet.py:210-217 (call :86). */
int acimg_deconv_fwd(const AcimgConvDesc* d, const float* x, const float* w, const float* bias,
                     float* y, void* ws, size_t ws_bytes, void* tickets, void* stream);
int acimg_deconv_dgrad(const AcimgConvDesc* d, const float* gy, int ldgy, const float* w,
                       float* dx, const float* mask, int ldmask, void* ws, size_t ws_bytes,
                       void* tickets, void* stream);
int acimg_deconv_wgrad(const AcimgConvDesc* d, const float* x, const float* gy, int ldgy,
                       float* dw, float* db, void* ws, size_t ws_bytes, void* stream);
size_t acimg_deconv_workspace(const AcimgConvDesc* d);

/* ------------------------------------------------------------------------------------------
 * Batch-norm pieces of the frozen ResNet-50 trunk (slim batch_norm, decay .997, eps 1e-5,
 * batch statistics while training: models/vision.py:55-56, trainer/mfcctrainer.py:348-349).
 * ---------------------------------------------------------------------------------------- */
/* Reduce the per-row-block partials of acimg_conv2d_fwd into scale/shift:
 *   mean = S1/count, var = S2/count - mean^2 (biased), scale = gamma/sqrt(var+eps),
 *   shift = beta - mean*scale; if training: moving_mean/var <- decay*old + (1-decay)*new with the
 *   UNBIASED variance (fused-BN semantics); save_mean/save_invstd optional.
 * If training == 0 the moving statistics are used instead of the partials. */
int acimg_bn_finalize(const float* stats, int rows, int C, int ldstats, double count,
                      const float* gamma, const float* beta, float* moving_mean,
                      float* moving_var, float decay, float eps, int training, float* scale,
                      float* shift, float* save_mean, float* save_invstd, void* stream);

/* ------------------------------------------------------------------------------------------
 * DualCamNet classifier head on generated acoustic images (models/dualcamnet.py:82-106,
 * trainer/trainer_reconstructed_class.py:44-56).  The convs / FCs are acimg_conv2d_* (the temporal
 * 12x1x1 conv3d is a 12x1 conv over a [clips, 12, 36*48, 12] view).
 * ---------------------------------------------------------------------------------------- */
/* tf.nn.max_pool k x k, stride k, VALID (models/base.py:34-38): y [N, H/k, W/k, C]. */
int acimg_maxpool_fwd(const float* x, int ldx, float* y, int ldy, int N, int H, int W, int C, int k, void* stream);
/* gx[N,H,W,C] = gradient w.r.t. the PRE-activation of the ReLU that produced x: gy of the window at its first
 * maximum if x > 0, else 0 (positions outside every window get 0). */
int acimg_maxpool_relu_bwd(const float* x, int ldx, const float* gy, int ldgy, float* gx, int ldgx, int N, int H,
                           int W, int C, int k, void* stream);
/* tf.reduce_sum(x, axis=[1,2]) (dualcamnet.py:96): y[n][c] = sum_p x[n][p][c]; and its backward through the ReLU
 * that produced x: gx[n][p][c] = x > 0 ? gy[n][c] : 0. */
int acimg_spatial_sum(const float* x, int ldx, float* y, int N, int P, int C, void* stream);
int acimg_spatial_sum_relu_bwd(const float* x, int ldx, const float* gy, float* gx, int ldgx, int N, int P, int C,
                               void* stream);
/* logits [clips*F, ldl] -> mean over the F frames of a clip -> tf.losses.softmax_cross_entropy (mean over clips)
 * and tf.argmax accuracy (trainer_reconstructed_class.py:49-56): out[0] += loss, out[1] += #correct (zero them
 * first); g_logits (optional) = d loss / d logits.  labels: int32 class per clip.  classes <= 64. */
int acimg_clip_softmax_ce(const float* logits, int ldl, int clips, int F, int K, const int* labels, float* out,
                          float* g_logits, int ldg, void* stream);

/* Training-mode batch-norm + ReLU backward for the conv-BN-ReLU stacks of the RGB / spectrogram U-Nets
 * (tf.layers.batch_normalization(training=True) + relu, models/unet_architecture.py:161-166,
 * models/unet_sound.py:155-160).  x = the pre-BN conv output (bias included), gy = gradient w.r.t. the ReLU
 * output; the ReLU mask is recomputed as x*scale+shift > 0 (scale/shift/save_* from acimg_bn_finalize):
 *   dbeta = sum g, dgamma = sum g*xhat, gx = gamma*invstd*(g - dbeta/n - xhat*dgamma/n).
 * gx may alias gy.  C % 4 == 0, C <= 1024.  Deterministic (ordered partial sums). */
size_t acimg_bn_bwd_workspace(long rows, int C);
int acimg_bn_bwd(const float* x, int ldx, const float* gy, int ldgy, const float* scale, const float* shift,
                 const float* save_mean, const float* save_invstd, const float* gamma, long rows, int C,
                 float* gx, int ldgx, float* dgamma, float* dbeta, void* ws, size_t ws_bytes, void* stream);

/* out = relu(a*sa[c]+ta[c] + shortcut), shortcut = b*sb[c]+tb[c] (sb given) or b (identity),
 * b read with spatial subsampling `bstride` (resnet_utils.subsample, models/resnet50.py:107).
 * a,out: [N,OH,OW,C]; b: [N,OH*bstride.. ,C]. Replaces models/resnet50.py:123. */
int acimg_bn_add_relu(const float* a, const float* sa, const float* ta, const float* b,
                      const float* sb, const float* tb, float* out, int N, int OH, int OW, int C,
                      int BH, int BW, int bstride, void* stream);

/* pool1: out = maxpool3x3/s2 'SAME' over relu(x*scale+shift) (models/resnet50.py:207-208). */
int acimg_bn_relu_maxpool(const float* x, const float* scale, const float* shift, float* out,
                          int N, int H, int W, int C, int OH, int OW, int pad_t, int pad_l,
                          void* stream);

/* y = relu(x*scale+shift) materialised (conv_map output feature, models/resnet50.py:208-209) */
int acimg_bn_relu(const float* x, const float* scale, const float* shift, float* y, long rows,
                  int C, int ldx, int ldy, void* stream);

/* Backward of y = relu(gamma*(x-mean)*invstd+beta) in batch-statistics mode for a small-C map
 * (conv_map, C=12): g_x, dgamma, dbeta from g_y. x,gy,y,gx: [rows,C] dense. */
int acimg_bn_relu_bwd(const float* x, const float* y, const float* gy, const float* gamma,
                      const float* save_mean, const float* save_invstd, float* gx, float* dgamma,
                      float* dbeta, long rows, int C, void* stream);

/* [N,H,W,3] -> [N,H,W,4] zero padded (stem input; lets the stem use 16-byte loads) */
int acimg_pad_channels(const float* x, float* y, long pixels, int C, int Cp, void* stream);

/* [N,H,W,C] -> the interior (at pad_t, pad_l) of a [N,Hp,Wp,Cp] frame the caller zeroed once: explicit zero
 * padding for the stem (resnet_utils.conv2d_same pads 3 + 3 before the VALID 7x7/2, models/resnet50.py:205),
 * so that the stem can run as a row-run convolution (acimg_conv2d_fwd_split3). */
int acimg_pad_image(const float* x, float* y, int N, int H, int W, int C, int Cp, int Hp, int Wp, int pad_t,
                    int pad_l, void* stream);

/* ------------------------------------------------------------------------------------------
 * Generator-side elementwise / reduction ops (models/unet_acresnet.py, trainer/mfcctrainer.py)
 * ---------------------------------------------------------------------------------------- */
/* tf.tile of the MFCC vector: out[n,h,w,c] = mfcc[n,c]   (trainer/mfcctrainer.py:38-40) */
int acimg_tile_mfcc(const float* mfcc, float* out, int N, int HW, int C, void* stream);

/* Per-sample min-max normalisation o = (x-min)/(max-min) over `cnt` = P*C logical elements of
 * sample n (x pixel stride ldx, out pixel stride ldo, written at out+0: pass an offset pointer to
 * write into a concat slice).  mm[n] = {min, max, #argmin, #argmax} saved for backward.
 * Every sample is cut into pixel chunks (one workgroup each); the per-chunk partials live in `ws`
 * (acimg_minmax_workspace(N, P, C) bytes, scratch: nothing is kept there between calls).
 * Replaces models/unet_acresnet.py:55-58, :70-71. */
size_t acimg_minmax_workspace(int N, int P, int C);
int acimg_minmax_fwd(const float* x, int ldx, float* out, int ldo, float* mm, int N, int P, int C, void* ws,
                     size_t ws_bytes, void* stream);
/* Gradient incl. the reduce_min / reduce_max paths with equal tie splitting (App. B.8).
 * gx = [accumulate? gx : 0] + d/dx; then zeroed where relu_mask (=x itself) <= 0 if mask_relu. */
int acimg_minmax_bwd(const float* x, int ldx, const float* go, int ldgo, const float* mm,
                     float* gx, int ldgx, int N, int P, int C, int accumulate, int mask_relu, void* ws,
                     size_t ws_bytes, void* stream);

/* heads [N,2*Z] = (mean | std_raw) -> sigma = softplus(std_raw), z = mean + sigma*eps,
 * kl[n] = 0.5*sum(mean^2+sigma^2-log(1e-8+sigma^2)-1).  z has row stride ldz (pads untouched).
 * Replaces models/unet_acresnet.py:73-78 and trainer/mfcctrainer.py:56-58. */
int acimg_latent_fwd(const float* heads, const float* eps, float* z, int ldz, float* sigma,
                     float* kl, int N, int Z, void* stream);
/* g_heads from gz (row stride ldgz) and the KL term weighted by kl_weight (= latent_loss / N) */
int acimg_latent_bwd(const float* heads, const float* eps, const float* sigma, const float* gz,
                     int ldgz, float kl_weight, float* g_heads, int N, int Z, void* stream);

/* Same without the softplus: the second head IS sigma (models/unet_architecture.py:62-69,
 * models/unet_sound.py:65-72: guessed_z = mean + variance * samples).  heads = [N][2Z] = [mean | sigma];
 * kl[n] = 0.5 * sum_j (mu^2 + s^2 - log(1e-8 + s^2) - 1)  (trainer/trainer.py:61-62 takes the mean over j:
 * fold 1/Z into the weights). */
int acimg_latent_linear_fwd(const float* heads, const float* eps, float* z, int ldz, float* kl, int N, int Z,
                            void* stream);
int acimg_latent_linear_bwd(const float* heads, const float* eps, const float* gz, int ldgz, float kl_weight,
                            float* g_heads, int N, int Z, void* stream);

/* y = softplus(x) and gx = gy * sigmoid(x) on [rows][C] tensors with row strides (the std tower of the latent
 * associators: tf.nn.softplus(tf.layers.dense(...)), models/multimodal.py:47-48,106-107). */
int acimg_softplus_fwd(const float* x, int ldx, float* y, int ldy, int rows, int C, void* stream);
int acimg_softplus_bwd(const float* x, int ldx, const float* gy, int ldgy, float* gx, int ldgx, int rows, int C,
                       void* stream);

/* "Tap GEMM" form of a stride-1 VALID convolution with few output channels (d->K <= 64; the trunk's conv_map,
 * models/resnet50.py:205-209 / models/vision.py:60-66: 3x4, 2048 -> 12).  With T = R*S*K:
 *   pack:    wt[c][tap*K + k] = w[tap][c][k]                 (then split / multiply it as a 1x1 conv C -> T)
 *   gather:  y[(n,oh,ow)][k] = sum_taps z[(n,oh+r,ow+s)][tap*K + k]  from z = x . wt over the INPUT pixels;
 *            optional batch-norm partials stats[acimg_tapconv_stats_rows(d)][2][d->ldw]
 *   scatter: gz[(n,ih,iw)][tap*K + k] = gy[(n,ih-r,iw-s)][k]  (0 outside), so that dwt = x^T . gz is a 1x1
 *            weight gradient
 *   unpack:  dw[tap][c][k] = dwt[c][tap*K + k] + decay * w[tap][c][k]   (w optional: slim's L2 term)
 * d is the descriptor of the ORIGINAL convolution. */
int acimg_tapconv_stats_rows(const AcimgConvDesc* d);
int acimg_tapconv_pack(const AcimgConvDesc* d, const float* w, float* wt, int ldwt, void* stream);
int acimg_tapconv_unpack(const AcimgConvDesc* d, const float* dwt, int ldwt, const float* w, float decay, float* dw,
                         void* stream);
int acimg_tapconv_gather(const AcimgConvDesc* d, const float* z, int ldz, float* y, float* stats, void* stream);
int acimg_tapconv_scatter(const AcimgConvDesc* d, const float* gy, int ldgy, float* gz, int ldgz, void* stream);

/* Cross-modal triplet losses over B embedding pairs e0[B][D], e1[B][D] with int32 labels / scenario per sample
 * (trainer/trainer_three.py): distances as `_pairwise_distances` :551-591 computes them (squared form,
 * D[i][j] = max(|e0_j|^2 - 2 <e0_i, e1_j> + |e1_i|^2, 0)); a pair (i, j) is "same video" when label and scenario
 * agree (:593-624); hard = 0: batch-all loss `mix_all` :685-732 (sum over valid triplets of max(D[a][p] - D[a][n] +
 * margin, 0) / (number of positive ones + 1e-16)); hard = 1: batch-hard loss `mix_data_hard` :648-683.
 * out[0] = loss, out[1] = fraction of positive triplets, out[2] = positive count, out[3] = valid count.
 * The forward leaves what the backward needs in ws (acimg_triplet_loss_workspace(B) bytes, B <= 2048);
 * bwd: g0 / g1 (either may be null) (+)= weight * d loss / d e0, e1, with TensorFlow's gradient conventions
 * (tf.maximum passes the gradient at equality; reduce_max / reduce_min split it evenly among ties). */
size_t acimg_triplet_loss_workspace(int B);
int acimg_triplet_loss_fwd(const float* e0, int lde0, const float* e1, int lde1, const int* labels,
                           const int* scenario, int B, int D, float margin, int hard, void* ws, size_t ws_bytes,
                           float* out, void* stream);
int acimg_triplet_loss_bwd(const float* e0, int lde0, const float* e1, int lde1, int B, int D, float weight,
                           const void* ws, size_t ws_bytes, float* g0, int ldg0, float* g1, int ldg1,
                           int accumulate, void* stream);

/* Reconstruction loss on yhat = sigmoid output and its gradient w.r.t. the PRE-sigmoid logits:
 *   sums[0] += sum (yhat-y)^2, sums[1] += sum huber_1(yhat-y)   (caller zeroes sums)
 *   g_logit = (w_mse*2e + w_huber*clip(e,-1,1)) / count * yhat*(1-yhat)
 * Replaces tf.losses.mean_squared_error / huber_loss, trainer/mfcctrainer.py:47,50. */
/* `scratch` (optional; acimg_loss_scratch_bytes() bytes, zeroed ONCE by the caller and dedicated to acimg_recon_loss /
 * acimg_sumsq calls of one stream) makes the two sums bit-reproducible: the workgroups' partials are combined in
 * workgroup order by the last arriver instead of by float atomics. */
size_t acimg_loss_scratch_bytes(void);
int acimg_recon_loss(const float* yhat, const float* target, float* g_logit, float* sums,
                     long count, float w_mse, float w_huber, void* scratch, size_t scratch_bytes, void* stream);

/* dst[p][c] = (accumulate ? dst : 0) + src[p][c] for c < C, zeroed where mask[p][c] <= 0 (mask
 * optional): routes the gradient of one channel slice of a tf.concat back to its producer
 * (models/unet_acresnet2skip.py:82, models/unet_acresnet.py:197-198). */
int acimg_grad_slice(const float* src, int ldsrc, float* dst, int lddst, const float* mask, int ldmask,
                     long pixels, int C, int accumulate, void* stream);

/* out[5] = {mse, huber, latent, reg, total}: mse = sums[0]/count, huber = sums[1]/count,
 * latent = latent_w * mean_n kl[n] (kl may be NULL: auto-encoder mode), reg = half_wd * sums[2],
 * total = latent + w_mse*mse + w_huber*huber + reg   (trainer/mfcctrainer.py:46-62). */
int acimg_loss_finalize(const float* sums, const float* kl, int N, double count, float latent_w,
                        float half_wd, float w_mse, float w_huber, float* out, void* stream);

/* out[i] ~ N(0,1), i < n: Philox4x32-10 keyed by `seed`, counter = offset + i/4, Box-Muller.
 * Stands where the reference samples tf.random_normal (models/unet_acresnet.py:77). */
int acimg_randn(float* out, long n, uint64_t seed, uint64_t offset, void* stream);

/* stream-ordered memset to zero (loss accumulators, gradient buffers) */
int acimg_zero(void* ptr, size_t bytes, void* stream);
/* A load for stream-concurrency probes: ONE wave keeps `stream` busy for `cycles` shader cycles (s_memtime; bounded by
 * 2^32) and then writes the elapsed cycles to out[0] (4 bytes, device memory).  The host library uses it to find HIP
 * streams that really run beside each other (the runtime maps streams onto a few hardware queues). */
int acimg_spin(uint64_t cycles, void* out, void* stream);

/* out[c] += sum over pixels of (a-b)^2 for channel c (dense [pixels][C], C<=64; caller zeroes out):
 * the per-3-channel test MSEs of trainer/mfcctrainer.py:105-117 are sums of 3 of these / count. */
int acimg_sqerr_channels(const float* a, const float* b, long pixels, int C, float* out, void* stream);

/* sum of squares of a flat buffer into *out (+=; `scratch` as for acimg_recon_loss) — slim l2_regularizer terms
 * (models/vision.py:54, tf.losses.get_total_loss trainer/mfcctrainer.py:60). */
int acimg_sumsq(const float* x, long n, float* out, void* scratch, size_t scratch_bytes, void* stream);
/* y += a*x */
int acimg_axpy(float a, const float* x, float* y, long n, void* stream);

/* TF-1 Adam (tf.train.AdamOptimizer, trainer/mfcctrainer.py:72-79):
 *   m=b1 m+(1-b1)g; v=b2 v+(1-b2)g^2; p -= lr_t * m/(sqrt(v)+eps), lr_t precomputed by host. */
int acimg_adam_step(float* p, const float* g, float* m, float* v, long n, float lr_t, float beta1,
                    float beta2, float eps, float grad_scale, void* stream);

/* ------------------------------------------------------------------------------------------
 * Audio front end (dataloader/outdoor_data_mfcc.py:796-876): one 1024-sample int32 frame ->
 * 12 MFCCs: Tukey(.75) window, rFFT-1024 (Nyquist dropped), power, 24 mel filters, floor 1e-3,
 * log, DCT(12)*sqrt(2/24), lifter(22).  window[1024], melfb[512*24] (row-major [bin][filter]),
 * dctl[24*12] (DCT*norm*lifter folded) are float64 tables built by the host exactly as the
 * reference does; the kernel computes in fp64 like NumPy and rounds to float32 at the end (:823).
 * normalize != 0 also applies the per-vector min-max of _normalize_mfcc (:696-703).
 * ---------------------------------------------------------------------------------------- */
int acimg_mfcc_frontend(const int32_t* frames, const double* window, const double* melfb,
                        const double* dctl, float* out, int nframes, int normalize, void* stream);
/* the same on float32 frames: the low-passed ("silence") waveform of butter_lowpass_filter, which the reference
 * feeds to the same function (dataloader/outdoor_data_mfcc.py:791-792) */
int acimg_mfcc_frontend_f32(const float* frames, const double* window, const double* melfb,
                            const double* dctl, float* out, int nframes, int normalize, void* stream);

/* STFT magnitude of the older audio path (dataloader/outdoor_data.py:844-851: tf.contrib.signal.stft(frame_length 246,
 * frame_step 122, fft_length 512) + tf.abs): wav [clips][nsamples] float32 -> out [clips][frames][257] float32,
 * frames = 1 + (nsamples - frame_len) / step (pad_end = False; 12288 samples -> 99 x 257).  window[frame_len]
 * (periodic Hann, float32) and twiddle[256][2] (exp(-2 pi i j / 512), float32 from float64) are host tables.  norm
 * (optional, [clips]) divides every sample first: the `wav / max |wav|` of _build_wav_py_function (:577-596); get it
 * from acimg_absmax.  fp32 arithmetic like TensorFlow's rfft. */
int acimg_stft_mag(const float* wav, const float* norm, const float* window, const float* twiddle, float* out,
                   int clips, int nsamples, int frame_len, int step, int fft_len, void* stream);
/* out[r] = max_i |x[r][i]|   (x: [rows][n]) */
int acimg_absmax(const float* x, int rows, int n, float* out, void* stream);

/* tf.image.resize_bilinear(x, [OH, OW], align_corners=False) with TF-1 sampling (src = dst * in/out, no half-pixel
 * centres), NHWC float32.  Replaces: trainer/trainer.py:367-369 (99x257 spectrogram -> 193x257). */
int acimg_resize_bilinear(const float* x, float* y, int N, int H, int W, int C, int OH, int OW, void* stream);

/* scipy.signal.filtfilt(b, a, x) along the last axis for 11-tap (order-10) IIR filters, default padding (odd
 * extension, padlen 33), float64 arithmetic without fused multiply-adds, float32 out:
 * dataloader/outdoor_data_mfcc.py:565-575 (butter_lowpass_filter, the "silence" variant of the MFCC input).
 * x: [rows][n] int32 (x_is_int32 != 0, the raw audio frames) or float32; ba: b[11] then a[11] (a[0] = 1);
 * zi: scipy.signal.lfilter_zi(b, a) [10]; ws: acimg_filtfilt_workspace(rows, n) bytes. */
int acimg_filtfilt(const void* x, int x_is_int32, int rows, int n, const double* ba, const double* zi, float* out,
                   void* ws, size_t ws_bytes, void* stream);
size_t acimg_filtfilt_workspace(int rows, int n);

/* find_logen energy map (iouenergythreshold.py:294-323): mfcc image [pixels,12] -> [pixels] */
int acimg_find_logen(const float* mfcc_img, const double* idct /*12x24*/, float* out, long pixels,
                     void* stream);

/* Mean-threshold IoU of two energy maps per sample (iouenergythreshold.py:213-229): m = map > mean(map),
 * iou[n] = |m_a & m_b| / |m_a | m_b|.  map_a, map_b: [N][P] float32 (acimg_find_logen outputs). */
int acimg_mask_iou(const float* map_a, const float* map_b, int N, int P, float* iou, void* stream);

/* ------------------------------------------------------------------------------------------
 * Dataset records (host side, no GPU work; caller-owned memory like everything else): the GZIP TFRecord files of
 * tf.train.SequenceExample written by convert_data.py:247-279 and read by dataloader/outdoor_data_mfcc.py:62,263-343.
 * ---------------------------------------------------------------------------------------- */
/* Inflate a whole .tfrecord file image (GZIP auto-detected by its magic; anything else is copied).  *produced =
 * the inflated size, also when `out` is NULL / too small (ACIMG_EWORKSPACE): call once to size, once to fill. */
int acimg_gzip_inflate(const uint8_t* src, size_t src_len, uint8_t* out, size_t cap, size_t* produced);
/* Walk the TFRecord framing [u64 len][u32 masked crc32c(len)][data][u32 masked crc32c(data)] of an inflated file
 * image: returns the number of records (>= 0) and fills the payload offset / length of the first `cap` of them;
 * verify != 0 checks both checksums of every record.  Negative = ACIMG_E* (truncated / corrupt). */
long acimg_tfrecord_index(const uint8_t* buf, size_t len, uint64_t* offsets, uint64_t* lengths, long cap, int verify);
/* What `_parse_sequence` (dataloader/outdoor_data_mfcc.py:263-343) extracts from one serialized SequenceExample. */
typedef struct AcimgSequenceDims {
    int64_t classes, location;                        /* context 'classes', 'location' */
    int64_t audio_height, audio_width, audio_depth;   /* 'audio_image/{height,width,depth}' (0 if absent) */
    int64_t mics, samples;                            /* 'audio_data/{mics,samples}' */
    int64_t video_height, video_width, video_depth;   /* 'video/{height,width,depth}' */
    int64_t audio_image_steps, audio_data_steps, video_steps;   /* feature-list lengths (12 = one second) */
    int64_t audio_data_values;                        /* int32 values over all 'audio/data' steps */
} AcimgSequenceDims;
/* Decode one record: context scalars + list lengths into *dims; the raw tensors (tf.decode_raw) into the caller's
 * buffers when given (NULL = sizes only; capacities in ELEMENTS): audio_images float32 [steps][H][W][D] already
 * flipped left-right and up-down as :314-315 do, audio_samples int32 [values], video uint8 [steps][H][W][D]. */
int acimg_sequence_example_decode(const uint8_t* rec, size_t len, AcimgSequenceDims* dims, float* audio_images,
                                  size_t audio_images_cap, int32_t* audio_samples, size_t audio_samples_cap,
                                  uint8_t* video, size_t video_cap);

/* ------------------------------------------------------------------------------------------
 * Host-side helper (no GPU): CRC-32C (Castagnoli) of a byte range, continuing from `crc` (0 to start) — the
 * checksum of TensorFlow checkpoint bundles and TFRecord files, used by the checkpoint / TFRecord readers
 * that stand where trainer/mfcctrainer.py:214-247 calls tf.train.Saver and
 * dataloader/outdoor_data_mfcc.py:558-575 reads TFRecords.
 * ---------------------------------------------------------------------------------------- */
uint32_t acimg_crc32c(const void* data, size_t n, uint32_t crc);

#ifdef __cplusplus
}
#endif
#endif /* ACIMG_H_ */
