// Audio front end on the GPU: 1024-sample frame -> 12 MFCCs, and the find_logen energy map.
// Follows dataloader/outdoor_data_mfcc.py:796-876 (NumPy, float64 inside, float32 out): the whole
// chain runs in fp64 here as well so the float32 results agree with NumPy to the last bits.
// One 256-thread block per frame: radix-2 FFT in LDS (16 KiB), mel filter bank as 8-lane partial
// dot products reduced with wave shuffles.
#include "common.hpp"

namespace acimg {

__global__ __launch_bounds__(256) void mfcc_frontend_kernel(const int32_t* frames, const double* window,
                                                            const double* melfb, const double* dctl,
                                                            float* out, int normalize) {
    __shared__ double re[1024];
    __shared__ double im[1024];
    __shared__ double logmel[24];
    __shared__ float coef[12];
    const int tid = threadIdx.x;
    const int32_t* x = frames + (long)blockIdx.x * 1024;

    for (int i = tid; i < 1024; i += 256) {
        const int rev = (int)(__brev((unsigned)i) >> 22);
        re[rev] = (double)x[i] * window[i];
        im[rev] = 0.0;
    }
    __syncthreads();
    for (int s = 1; s <= 10; ++s) {
        const int m = 1 << s, half = m >> 1;
        for (int b = tid; b < 512; b += 256) {
            const int grp = b / half, j = b - grp * half;
            const int i0 = grp * m + j, i1 = i0 + half;
            double sn, cs;
            sincospi(-2.0 * (double)j / (double)m, &sn, &cs);
            const double tr = cs * re[i1] - sn * im[i1];
            const double ti = cs * im[i1] + sn * re[i1];
            const double ur = re[i0], ui = im[i0];
            re[i0] = ur + tr;
            im[i0] = ui + ti;
            re[i1] = ur - tr;
            im[i1] = ui - ti;
        }
        __syncthreads();
    }
    // power spectrum of bins 0..511 (the Nyquist bin is dropped, :803) kept in re[]
    for (int k = tid; k < 512; k += 256) {
        const double a = sqrt(re[k] * re[k] + im[k] * im[k]);
        re[k] = a * a;
    }
    __syncthreads();
    {
        const int f = tid >> 3, part = tid & 7;
        double acc = 0.0;
        if (f < 24)
            for (int k = part; k < 512; k += 8) acc += re[k] * melfb[k * 24 + f];
        acc += __shfl_xor(acc, 1, 64);
        acc += __shfl_xor(acc, 2, 64);
        acc += __shfl_xor(acc, 4, 64);
        if (f < 24 && part == 0) logmel[f] = log(acc < 0.001 ? 0.001 : acc);
    }
    __syncthreads();
    if (tid < 12) {
        double c = 0.0;
        for (int f = 0; f < 24; ++f) c += logmel[f] * dctl[f * 12 + tid];
        if (isnan(c) || isinf(c)) c = 0.0;
        coef[tid] = (float)c;
    }
    __syncthreads();
    if (tid < 12) {
        float v = coef[tid];
        if (normalize) {
            float mn = coef[0], mx;
            for (int i = 1; i < 12; ++i) mn = fminf(mn, coef[i]);
            mx = coef[0] - mn;
            for (int i = 1; i < 12; ++i) mx = fmaxf(mx, coef[i] - mn);
            v = (v - mn) / mx;
        }
        out[(long)blockIdx.x * 12 + tid] = v;
    }
}

// iouenergythreshold.py:294-323 — lifter/norm folded into idct[12][24] by the host
__global__ __launch_bounds__(256) void find_logen_kernel(const float* mfcc, const double* idct, float* out,
                                                         long pixels) {
    __shared__ double w[12 * 24];
    for (int i = threadIdx.x; i < 12 * 24; i += 256) w[i] = idct[i];
    __syncthreads();
    const long p = (long)blockIdx.x * 256 + threadIdx.x;
    if (p >= pixels) return;
    double c[12];
#pragma unroll
    for (int j = 0; j < 12; ++j) c[j] = (double)mfcc[p * 12 + j];
    double s = 0.0;
    for (int f = 0; f < 24; ++f) {
        double a = 0.0;
#pragma unroll
        for (int j = 0; j < 12; ++j) a += c[j] * w[j * 24 + f];
        s += exp(a);
    }
    out[p] = (float)(1.0 / s);
}

// iouenergythreshold.py:213-229: per sample, mask = map > mean(map) for the real and the generated energy map,
// IoU = |m & m2| / |m | m2| (float64 means like NumPy's on float64 maps; 0/0 -> NaN as in the reference).
// One workgroup per sample.
__global__ __launch_bounds__(256) void mask_iou_kernel(const float* a, const float* b, int P, float* iou) {
    __shared__ double sm[8];
    __shared__ double means[2];
    const float* pa = a + (long)blockIdx.x * P;
    const float* pb = b + (long)blockIdx.x * P;
    double sa = 0.0, sb = 0.0;
    for (int i = threadIdx.x; i < P; i += 256) {
        sa += (double)pa[i];
        sb += (double)pb[i];
    }
    sa = wave_sum_d(sa);
    sb = wave_sum_d(sb);
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) {
        sm[wid] = sa;
        sm[4 + wid] = sb;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        means[0] = (sm[0] + sm[1] + sm[2] + sm[3]) / (double)P;
        means[1] = (sm[4] + sm[5] + sm[6] + sm[7]) / (double)P;
    }
    __syncthreads();
    double inter = 0.0, uni = 0.0;
    for (int i = threadIdx.x; i < P; i += 256) {
        const bool m1 = (double)pa[i] > means[0], m2 = (double)pb[i] > means[1];
        inter += (m1 && m2) ? 1.0 : 0.0;
        uni += (m1 || m2) ? 1.0 : 0.0;
    }
    inter = wave_sum_d(inter);
    uni = wave_sum_d(uni);
    __syncthreads();
    if (lane == 0) {
        sm[wid] = inter;
        sm[4 + wid] = uni;
    }
    __syncthreads();
    if (threadIdx.x == 0)
        iou[blockIdx.x] = (float)((sm[0] + sm[1] + sm[2] + sm[3]) / (sm[4] + sm[5] + sm[6] + sm[7]));
}

}  // namespace acimg

using namespace acimg;

extern "C" {

int acimg_mfcc_frontend(const int32_t* frames, const double* window, const double* melfb,
                        const double* dctl, float* out, int nframes, int normalize, void* stream) {
    if (nframes <= 0) return fail(ACIMG_EINVAL, "mfcc_frontend: nframes must be positive");
    hipLaunchKernelGGL(mfcc_frontend_kernel, dim3(nframes), dim3(256), 0, (hipStream_t)stream, frames,
                       window, melfb, dctl, out, normalize);
    return check_launch("mfcc_frontend");
}

int acimg_find_logen(const float* mfcc_img, const double* idct, float* out, long pixels, void* stream) {
    if (pixels <= 0) return fail(ACIMG_EINVAL, "find_logen: pixels must be positive");
    hipLaunchKernelGGL(find_logen_kernel, dim3(cdiv(pixels, 256)), dim3(256), 0, (hipStream_t)stream,
                       mfcc_img, idct, out, pixels);
    return check_launch("find_logen");
}

int acimg_mask_iou(const float* map_a, const float* map_b, int N, int P, float* iou, void* stream) {
    if (N <= 0 || P <= 0) return fail(ACIMG_EINVAL, "mask_iou: N and P must be positive");
    hipLaunchKernelGGL(mask_iou_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, map_a, map_b, P, iou);
    return check_launch("mask_iou");
}

}  // extern "C"
