// Cross-modal triplet losses over a batch of embedding pairs (trainer/trainer_three.py:551-732): the "distance"
// matrix of `_pairwise_distances` (squared form, quirks kept: D[i][j] = |e0_j|^2 - 2 <e0_i, e1_j> + |e1_i|^2,
// clamped at 0), the batch-all loss `mix_all` and the batch-hard loss `mix_data_hard`, and their gradients w.r.t.
// both embedding sets with TensorFlow's gradient conventions (tf.maximum passes the gradient at equality,
// reduce_max / reduce_min split it evenly among ties).
//
// B is a batch (tens to hundreds): one workgroup per anchor row, the row of distances and the same-video flags in
// LDS, integer triplet counts (exact), float sums combined in a fixed order -> bit-reproducible run to run.
// Workspace (floats): raw[B*B] | gun[B*B] | rowstats[B*4] | scal[4].
#include "common.hpp"

namespace acimg {

constexpr int TRI_THREADS = 256;

__global__ __launch_bounds__(TRI_THREADS) void triplet_dist_kernel(const float* e0, int ld0, const float* e1, int ld1,
                                                                   int B, int D, float* raw) {
    const int i = blockIdx.x;
    float n1 = 0.f;
    for (int d = 0; d < D; ++d) n1 = fmaf(e1[(long)i * ld1 + d], e1[(long)i * ld1 + d], n1);
    for (int j = threadIdx.x; j < B; j += TRI_THREADS) {
        float dot = 0.f, n0 = 0.f;
        for (int d = 0; d < D; ++d) {
            const float a = e0[(long)j * ld0 + d];
            n0 = fmaf(a, a, n0);
        }
        for (int d = 0; d < D; ++d) dot = fmaf(e0[(long)i * ld0 + d], e1[(long)j * ld1 + d], dot);
        raw[(long)i * B + j] = (n0 - 2.f * dot) + n1;
    }
}

template <typename T, typename Op>
__device__ __forceinline__ T block_reduce(T v, T* scratch, Op op) {
    const int tid = threadIdx.x;
    __syncthreads();
    scratch[tid] = v;
    __syncthreads();
    for (int s = TRI_THREADS / 2; s > 0; s >>= 1) {
        if (tid < s) scratch[tid] = op(scratch[tid], scratch[tid + s]);
        __syncthreads();
    }
    return scratch[0];
}

// LDS: Drow[B] floats | same[B] ints | (hard) an[B] floats
__global__ __launch_bounds__(TRI_THREADS) void triplet_all_kernel(const float* raw, const int* labels,
                                                                  const int* scenario, int B, float margin,
                                                                  float* gun, float* rowstats) {
    extern __shared__ __attribute__((aligned(16))) float tri_smem[];
    __shared__ float fred[TRI_THREADS];
    __shared__ int ired[TRI_THREADS];
    float* Drow = tri_smem;
    int* same = reinterpret_cast<int*>(tri_smem + B);
    const int i = blockIdx.x, tid = threadIdx.x;
    const int li = labels[i], si = scenario[i];
    for (int c = tid; c < B; c += TRI_THREADS) {
        Drow[c] = fmaxf(raw[(long)i * B + c], 0.f);
        same[c] = labels[c] == li && scenario[c] == si;
    }
    __syncthreads();
    float sum = 0.f;
    int npos = 0, np = 0;
    for (int c = tid; c < B; c += TRI_THREADS) {
        int cnt = 0;
        const float dc = Drow[c];
        if (same[c]) {                       // c is the positive: sweep the negatives
            ++np;
            for (int k = 0; k < B; ++k) {
                if (same[k]) continue;
                const float t = (dc - Drow[k]) + margin;
                cnt += t >= 0.f;
                const float v = fmaxf(t, 0.f);
                sum += v;
                npos += v > 1e-16f;
            }
            gun[(long)i * B + c] = (float)cnt;
        } else {                             // c is the negative: sweep the positives
            for (int j = 0; j < B; ++j) {
                if (!same[j]) continue;
                const float t = (Drow[j] - dc) + margin;
                cnt += t >= 0.f;
            }
            gun[(long)i * B + c] = -(float)cnt;
        }
    }
    const float S = block_reduce(sum, fred, [](float a, float b) { return a + b; });
    const int NP = block_reduce(npos, ired, [](int a, int b) { return a + b; });
    const int P = block_reduce(np, ired, [](int a, int b) { return a + b; });
    if (tid == 0) {
        rowstats[i * 4 + 0] = S;
        rowstats[i * 4 + 1] = (float)NP;
        rowstats[i * 4 + 2] = (float)P * (float)(B - P);
        rowstats[i * 4 + 3] = 0.f;
    }
}

__global__ __launch_bounds__(TRI_THREADS) void triplet_hard_kernel(const float* raw, const int* labels,
                                                                   const int* scenario, int B, float margin,
                                                                   float* gun, float* rowstats) {
    extern __shared__ __attribute__((aligned(16))) float tri_smem[];
    __shared__ float fred[TRI_THREADS];
    __shared__ int ired[TRI_THREADS];
    float* Drow = tri_smem;
    int* same = reinterpret_cast<int*>(tri_smem + B);
    float* an = tri_smem + 2 * B;
    const int i = blockIdx.x, tid = threadIdx.x;
    const int li = labels[i], si = scenario[i];
    float hp = 0.f, rm = -INFINITY;          // same * D >= 0 and same[i][i] holds: the max over the row is >= 0
    int np = 0;
    for (int c = tid; c < B; c += TRI_THREADS) {
        const float d = fmaxf(raw[(long)i * B + c], 0.f);
        const int s = labels[c] == li && scenario[c] == si;
        Drow[c] = d;
        same[c] = s;
        np += s;
        hp = fmaxf(hp, s ? d : 0.f);
        rm = fmaxf(rm, d);
    }
    auto fmaxop = [](float a, float b) { return fmaxf(a, b); };
    auto fminop = [](float a, float b) { return fminf(a, b); };
    auto iadd = [](int a, int b) { return a + b; };
    hp = block_reduce(hp, fred, fmaxop);
    rm = block_reduce(rm, fred, fmaxop);
    float hn = INFINITY;
    for (int c = tid; c < B; c += TRI_THREADS) {
        const float a = Drow[c] + rm * (same[c] ? 1.f : 0.f);     // pairwise_dist + max * (1 - mask_negative)
        an[c] = a;
        hn = fminf(hn, a);
    }
    hn = block_reduce(hn, fred, fminop);
    int chp = 0, crm = 0, chn = 0, chn_same = 0;
    for (int c = tid; c < B; c += TRI_THREADS) {
        chp += (same[c] ? Drow[c] : 0.f) == hp;
        crm += Drow[c] == rm;
        const int e = an[c] == hn;
        chn += e;
        chn_same += e && same[c];
    }
    chp = block_reduce(chp, ired, iadd);
    crm = block_reduce(crm, ired, iadd);
    chn = block_reduce(chn, ired, iadd);
    chn_same = block_reduce(chn_same, ired, iadd);
    const int P = block_reduce(np, ired, iadd);
    const float x = (hp - hn) + margin;
    const float tl = fmaxf(x, 0.f);
    const float pass = x >= 0.f ? 1.f : 0.f;
    const float A = (float)chn_same / (float)chn;
    for (int c = tid; c < B; c += TRI_THREADS) {
        float gsum = 0.f;
        if (same[c] && Drow[c] == hp) gsum += 1.f / (float)chp;
        if (an[c] == hn) gsum -= 1.f / (float)chn;
        if (Drow[c] == rm) gsum -= A / (float)crm;
        gun[(long)i * B + c] = pass * gsum;
    }
    if (tid == 0) {
        rowstats[i * 4 + 0] = tl;
        rowstats[i * 4 + 1] = tl > 1e-16f ? 1.f : 0.f;
        rowstats[i * 4 + 2] = (float)P * (float)(B - P);
        rowstats[i * 4 + 3] = 0.f;
    }
}

// out[0] = loss, out[1] = fraction of positive triplets, out[2] = positive count, out[3] = valid count; scal[0] =
// d loss / d (row sum)
__global__ __launch_bounds__(TRI_THREADS) void triplet_finalize_kernel(const float* rowstats, int B, int hard,
                                                                       float* scal, float* out) {
    __shared__ double dred[3][TRI_THREADS];
    const int tid = threadIdx.x;
    double s = 0.0, np = 0.0, nv = 0.0;
    for (int r = tid; r < B; r += TRI_THREADS) {
        s += rowstats[r * 4 + 0];
        np += rowstats[r * 4 + 1];
        nv += rowstats[r * 4 + 2];
    }
    dred[0][tid] = s;
    dred[1][tid] = np;
    dred[2][tid] = nv;
    __syncthreads();
    for (int st = TRI_THREADS / 2; st > 0; st >>= 1) {
        if (tid < st)
            for (int q = 0; q < 3; ++q) dred[q][tid] += dred[q][tid + st];
        __syncthreads();
    }
    if (tid == 0) {
        const float S = (float)dred[0][0], NP = (float)dred[1][0], NV = (float)dred[2][0];
        const float den = hard ? (float)B : NP + 1e-16f;
        out[0] = S / den;
        out[1] = NP / (NV + 1e-16f);
        out[2] = NP;
        out[3] = NV;
        scal[0] = 1.f / den;
    }
}

// grid (B, 2): y = 0 -> d loss / d e0[r], y = 1 -> d loss / d e1[r];  LDS: grow[B] | gcol[B]
__global__ __launch_bounds__(TRI_THREADS) void triplet_grad_kernel(const float* raw, const float* gun,
                                                                   const float* scal, float weight, const float* e0,
                                                                   int ld0, const float* e1, int ld1, int B, int D,
                                                                   float* g0, int ldg0, float* g1, int ldg1,
                                                                   int accumulate) {
    extern __shared__ __attribute__((aligned(16))) float tri_smem[];
    __shared__ float fred[TRI_THREADS];
    float* gv = tri_smem;
    const int r = blockIdx.x, which = blockIdx.y, tid = threadIdx.x;
    if ((which == 0 ? g0 : g1) == nullptr) return;
    float part = 0.f;
    for (int c = tid; c < B; c += TRI_THREADS) {
        // e0 side: column r of G weights the norms, row r the cross terms; e1 side: the other way round
        const long rowidx = (long)r * B + c, colidx = (long)c * B + r;
        const float grow = raw[rowidx] >= 0.f ? gun[rowidx] : 0.f;
        const float gcol = raw[colidx] >= 0.f ? gun[colidx] : 0.f;
        gv[c] = which == 0 ? grow : gcol;
        part += which == 0 ? gcol : grow;
    }
    const float nsum = block_reduce(part, fred, [](float a, float b) { return a + b; });
    const float gs = scal[0] * weight;
    const float* own = which == 0 ? e0 + (long)r * ld0 : e1 + (long)r * ld1;
    const float* oth = which == 0 ? e1 : e0;
    const int ldo = which == 0 ? ld1 : ld0;
    float* dst = which == 0 ? g0 + (long)r * ldg0 : g1 + (long)r * ldg1;
    for (int d = tid; d < D; d += TRI_THREADS) {
        float acc = 0.f;
        for (int c = 0; c < B; ++c) acc = fmaf(gv[c], oth[(long)c * ldo + d], acc);
        const float v = gs * (2.f * own[d] * nsum - 2.f * acc);
        dst[d] = accumulate ? dst[d] + v : v;
    }
}

}  // namespace acimg

using namespace acimg;

extern "C" {

size_t acimg_triplet_loss_workspace(int B) {
    return B > 0 ? ((size_t)2 * B * B + (size_t)4 * B + 4) * sizeof(float) : 0;
}

int acimg_triplet_loss_fwd(const float* e0, int lde0, const float* e1, int lde1, const int* labels,
                           const int* scenario, int B, int D, float margin, int hard, void* ws, size_t ws_bytes,
                           float* out, void* stream) {
    if (!e0 || !e1 || !labels || !scenario || !ws || !out) return fail(ACIMG_EINVAL, "triplet_loss_fwd: null argument");
    if (B <= 0 || B > 2048 || D <= 0 || lde0 < D || lde1 < D)
        return fail(ACIMG_EINVAL, "triplet_loss_fwd: need 0 < B <= 2048, 0 < D <= lde0, lde1 (B=%d D=%d)", B, D);
    if (ws_bytes < acimg_triplet_loss_workspace(B)) return fail(ACIMG_EINVAL, "triplet_loss_fwd: workspace too small");
    float* raw = static_cast<float*>(ws);
    float* gun = raw + (size_t)B * B;
    float* rowstats = gun + (size_t)B * B;
    float* scal = rowstats + (size_t)4 * B;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(triplet_dist_kernel, dim3(B), dim3(TRI_THREADS), 0, st, e0, lde0, e1, lde1, B, D, raw);
    if (hard)
        hipLaunchKernelGGL(triplet_hard_kernel, dim3(B), dim3(TRI_THREADS), (size_t)3 * B * 4, st, raw, labels,
                           scenario, B, margin, gun, rowstats);
    else
        hipLaunchKernelGGL(triplet_all_kernel, dim3(B), dim3(TRI_THREADS), (size_t)2 * B * 4, st, raw, labels,
                           scenario, B, margin, gun, rowstats);
    hipLaunchKernelGGL(triplet_finalize_kernel, dim3(1), dim3(TRI_THREADS), 0, st, rowstats, B, hard ? 1 : 0, scal, out);
    return check_launch("triplet_loss_fwd");
}

int acimg_triplet_loss_bwd(const float* e0, int lde0, const float* e1, int lde1, int B, int D, float weight,
                           const void* ws, size_t ws_bytes, float* g0, int ldg0, float* g1, int ldg1,
                           int accumulate, void* stream) {
    if (!e0 || !e1 || !ws || (!g0 && !g1)) return fail(ACIMG_EINVAL, "triplet_loss_bwd: null argument");
    if (B <= 0 || B > 2048 || D <= 0 || lde0 < D || lde1 < D || (g0 && ldg0 < D) || (g1 && ldg1 < D))
        return fail(ACIMG_EINVAL, "triplet_loss_bwd: bad shape (B=%d D=%d)", B, D);
    if (ws_bytes < acimg_triplet_loss_workspace(B)) return fail(ACIMG_EINVAL, "triplet_loss_bwd: workspace too small");
    const float* raw = static_cast<const float*>(ws);
    const float* gun = raw + (size_t)B * B;
    const float* scal = gun + (size_t)B * B + (size_t)4 * B;
    hipLaunchKernelGGL(triplet_grad_kernel, dim3(B, 2), dim3(TRI_THREADS), (size_t)B * 4, (hipStream_t)stream, raw,
                       gun, scal, weight, e0, lde0, e1, lde1, B, D, g0, ldg0, g1, ldg1, accumulate ? 1 : 0);
    return check_launch("triplet_loss_bwd");
}

}  // extern "C"
