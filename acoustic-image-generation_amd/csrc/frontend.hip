// Audio front end on the GPU: 1024-sample frame -> 12 MFCCs, and the find_logen energy map.
// Follows dataloader/outdoor_data_mfcc.py:796-876 (NumPy, float64 inside, float32 out): the whole
// chain runs in fp64 here as well so the float32 results agree with NumPy to the last bits.
// One 256-thread block per frame: radix-2 FFT in LDS (16 KiB), mel filter bank as 8-lane partial
// dot products reduced with wave shuffles.
#include <type_traits>

#include "common.hpp"

namespace acimg {

template <typename T>
__global__ __launch_bounds__(256) void mfcc_frontend_kernel(const T* frames, const double* window,
                                                            const double* melfb, const double* dctl,
                                                            float* out, int normalize) {
    __shared__ double re[1024];
    __shared__ double im[1024];
    __shared__ double logmel[24];
    __shared__ float coef[12];
    const int tid = threadIdx.x;
    const T* x = frames + (long)blockIdx.x * 1024;

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


// ---- STFT magnitude (dataloader/outdoor_data.py:844-851: tf.contrib.signal.stft + tf.abs) -----------------------
// One 256-thread block per frame: frame_len samples times the window table (periodic Hann, built by the host as
// tf.contrib.signal.hann_window does), zero-padded to NFFT = 512, radix-2 complex FFT in LDS (one butterfly per
// thread per stage; fp32 like TF's rfft, twiddles from a host table computed in float64), |X[k]| for k = 0..256.
// pad_end = False: frames = 1 + (nsamples - frame_len) / step (12288 samples -> 99 x 257).
// `norm` (optional, one float per clip) is the divisor of dataloader/outdoor_data.py:577-596
// (_build_wav_py_function: wav / max |wav|); every sample is divided by it BEFORE the window, as the reference
// normalises the waveform first.
constexpr int STFT_NFFT = 512;
__global__ __launch_bounds__(256) void stft_mag_kernel(const float* wav, const float* norm, const float* window,
                                                       const float2* twiddle, float* out, int nsamples, int frame_len,
                                                       int step, int frames) {
    __shared__ float2 b[STFT_NFFT];
    const int tid = threadIdx.x;
    const long f = blockIdx.x;
    const long clip = f / frames;
    const int fr = (int)(f - clip * frames);
    const float* x = wav + clip * nsamples + (long)fr * step;
    const float dv = norm ? norm[clip] : 1.f;
    for (int i = tid; i < STFT_NFFT; i += 256) {
        const int rev = (int)(__brev((unsigned)i) >> 23);
        float v = 0.f;
        if (i < frame_len) {
            v = x[i];
            if (norm) v = v / dv;
            v *= window[i];
        }
        b[rev] = make_float2(v, 0.f);
    }
    __syncthreads();
    for (int s = 1; s <= 9; ++s) {
        const int m = 1 << s, half = m >> 1, tstep = STFT_NFFT / m;
        const int grp = tid / half, j = tid - grp * half;
        const int i0 = grp * m + j, i1 = i0 + half;
        const float2 tw = twiddle[j * tstep];              // exp(-2 pi i j / m)
        const float2 u = b[i0], v = b[i1];
        const float tr = tw.x * v.x - tw.y * v.y;
        const float ti = tw.x * v.y + tw.y * v.x;
        b[i0] = make_float2(u.x + tr, u.y + ti);           // a butterfly owns its two slots: no hazard inside a stage
        b[i1] = make_float2(u.x - tr, u.y - ti);
        __syncthreads();
    }
    float* o = out + f * (STFT_NFFT / 2 + 1);
    for (int k = tid; k <= STFT_NFFT / 2; k += 256) o[k] = sqrtf(b[k].x * b[k].x + b[k].y * b[k].y);
}

// max |x| per row: the divisor `abs(max(min, max, key=abs))` of _build_wav_py_function
__global__ __launch_bounds__(256) void absmax_kernel(const float* x, int n, float* out) {
    __shared__ float sm[4];
    const float* p = x + (long)blockIdx.x * n;
    float m = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) m = fmaxf(m, fabsf(p[i]));
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
}

// ---- tf.image.resize_bilinear, align_corners = False, TF-1 (no half-pixel centres) (trainer/trainer.py:364-369) ---
// scale = in / out (float); src = dst * scale; lo = floor(src), hi = min(lo + 1, in - 1), lerp = src - lo;
// top = tl + (tr - tl) * xl; bottom = bl + (br - bl) * xl; out = top + (bottom - top) * yl   — all in fp32, the
// operation order of TensorFlow's resize_bilinear kernel.
__global__ __launch_bounds__(256) void resize_bilinear_kernel(const float* x, float* y, long total, int H, int W,
                                                              int C, int OH, int OW, float hs, float ws) {
#pragma clang fp contract(off)
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int c = (int)(idx % C);
    long t = idx / C;
    const int ow = (int)(t % OW);
    t /= OW;
    const int oh = (int)(t % OH);
    const long n = t / OH;
    const float sy = (float)oh * hs, sx = (float)ow * ws;
    const int y0 = (int)floorf(sy), x0 = (int)floorf(sx);
    const int y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);
    const float yl = sy - (float)y0, xl = sx - (float)x0;
    const float* base = x + n * (long)H * W * C + c;
    const float tl = base[((long)y0 * W + x0) * C], tr = base[((long)y0 * W + x1) * C];
    const float bl = base[((long)y1 * W + x0) * C], br = base[((long)y1 * W + x1) * C];
    const float top = tl + (tr - tl) * xl;
    const float bot = bl + (br - bl) * xl;
    y[idx] = top + (bot - top) * yl;
}

// ---- scipy.signal.filtfilt(b, a, x) along the last axis (dataloader/outdoor_data_mfcc.py:565-575) ------------------
// Order-10 Butterworth low-pass (11 taps each); method 'pad', padtype 'odd', padlen = 3 * 11 = 33: the signal is
// extended by 33 odd-reflected samples on both sides, filtered forward with the steady-state initial condition
// zi * ext[0], reversed, filtered again with zi * y[last], reversed, and the padding is dropped.  The recurrence is
// SciPy's lfilter (direct form II transposed, float64):  y = z0 + b0*x;  z_k = z_{k+1} + b_{k+1}*x - a_{k+1}*y.
// One lane per row; fp64 with NO fused multiply-add (SciPy's C loop rounds every product), so results follow the
// reference closely although the 125 Hz / 12288 Hz design is very ill-conditioned (b0 ~ 9e-16).
// `tmp` holds the forward pass, transposed ([sample][row]) so that lanes write and read it coalesced.
constexpr int FF_TAPS = 11, FF_PAD = 33;
template <typename T>
__global__ __launch_bounds__(64) void filtfilt_kernel(const T* x, int rows, int n, const double* ba, const double* zi,
                                                      double* tmp, float* out) {
#pragma clang fp contract(off)
    const int row = blockIdx.x * 64 + threadIdx.x;
    if (row >= rows) return;
    double b[FF_TAPS], a[FF_TAPS], z[FF_TAPS - 1];
#pragma unroll
    for (int k = 0; k < FF_TAPS; ++k) {
        b[k] = ba[k];
        a[k] = ba[FF_TAPS + k];
    }
    const T* xr = x + (long)row * n;
    const int next = n + 2 * FF_PAD;
    const double x0 = (double)xr[0], xe = (double)xr[n - 1];
    // odd extension (scipy.signal._arraytools.odd_ext): computed in the INPUT's dtype as NumPy does — exact for the
    // loader's int32 frames, one float32 rounding for float32 input — then promoted to float64 by lfilter
    auto edge = [](double v) -> double { return sizeof(T) == sizeof(float) && !std::is_integral<T>::value ? (double)(float)v : v; };
    auto ext = [&](int i) -> double {
        if (i < FF_PAD) return edge(2.0 * x0 - (double)xr[FF_PAD - i]);
        if (i < FF_PAD + n) return (double)xr[i - FF_PAD];
        return edge(2.0 * xe - (double)xr[n - 2 - (i - FF_PAD - n)]);
    };
    const double e0 = ext(0);
#pragma unroll
    for (int k = 0; k < FF_TAPS - 1; ++k) z[k] = zi[k] * e0;
    double last = 0.0;
    for (int i = 0; i < next; ++i) {
        const double v = ext(i);
        const double y = z[0] + b[0] * v;
#pragma unroll
        for (int k = 0; k < FF_TAPS - 2; ++k) z[k] = z[k + 1] + b[k + 1] * v - a[k + 1] * y;
        z[FF_TAPS - 2] = b[FF_TAPS - 1] * v - a[FF_TAPS - 1] * y;
        tmp[(long)i * rows + row] = y;
        last = y;
    }
#pragma unroll
    for (int k = 0; k < FF_TAPS - 1; ++k) z[k] = zi[k] * last;
    for (int i = next - 1; i >= 0; --i) {
        const double v = tmp[(long)i * rows + row];
        const double y = z[0] + b[0] * v;
#pragma unroll
        for (int k = 0; k < FF_TAPS - 2; ++k) z[k] = z[k + 1] + b[k + 1] * v - a[k + 1] * y;
        z[FF_TAPS - 2] = b[FF_TAPS - 1] * v - a[FF_TAPS - 1] * y;
        if (i >= FF_PAD && i < FF_PAD + n) out[(long)row * n + (i - FF_PAD)] = (float)y;
    }
}

}  // namespace acimg

using namespace acimg;

extern "C" {

int acimg_mfcc_frontend(const int32_t* frames, const double* window, const double* melfb,
                        const double* dctl, float* out, int nframes, int normalize, void* stream) {
    if (nframes <= 0) return fail(ACIMG_EINVAL, "mfcc_frontend: nframes must be positive");
    hipLaunchKernelGGL(mfcc_frontend_kernel<int32_t>, dim3(nframes), dim3(256), 0, (hipStream_t)stream, frames,
                       window, melfb, dctl, out, normalize);
    return check_launch("mfcc_frontend");
}

int acimg_mfcc_frontend_f32(const float* frames, const double* window, const double* melfb,
                            const double* dctl, float* out, int nframes, int normalize, void* stream) {
    if (nframes <= 0) return fail(ACIMG_EINVAL, "mfcc_frontend_f32: nframes must be positive");
    hipLaunchKernelGGL(mfcc_frontend_kernel<float>, dim3(nframes), dim3(256), 0, (hipStream_t)stream, frames,
                       window, melfb, dctl, out, normalize);
    return check_launch("mfcc_frontend_f32");
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

int acimg_stft_mag(const float* wav, const float* norm, const float* window, const float* twiddle, float* out,
                   int clips, int nsamples, int frame_len, int step, int fft_len, void* stream) {
    if (clips <= 0 || nsamples <= 0 || frame_len <= 0 || step <= 0)
        return fail(ACIMG_EINVAL, "stft_mag: clips, nsamples, frame_len and step must be positive");
    if (fft_len != STFT_NFFT) return fail(ACIMG_EINVAL, "stft_mag: fft_len must be %d", STFT_NFFT);
    if (frame_len > fft_len || frame_len > nsamples)
        return fail(ACIMG_EINVAL, "stft_mag: frame_len %d exceeds fft_len / nsamples", frame_len);
    const int frames = 1 + (nsamples - frame_len) / step;
    hipLaunchKernelGGL(stft_mag_kernel, dim3((unsigned)((long)clips * frames)), dim3(256), 0, (hipStream_t)stream, wav,
                       norm, window, reinterpret_cast<const float2*>(twiddle), out, nsamples, frame_len, step, frames);
    return check_launch("stft_mag");
}

int acimg_absmax(const float* x, int rows, int n, float* out, void* stream) {
    if (rows <= 0 || n <= 0) return fail(ACIMG_EINVAL, "absmax: rows and n must be positive");
    hipLaunchKernelGGL(absmax_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, x, n, out);
    return check_launch("absmax");
}

int acimg_resize_bilinear(const float* x, float* y, int N, int H, int W, int C, int OH, int OW, void* stream) {
    if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || OH <= 0 || OW <= 0)
        return fail(ACIMG_EINVAL, "resize_bilinear: every extent must be positive");
    const long total = (long)N * OH * OW * C;
    hipLaunchKernelGGL(resize_bilinear_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, x, y, total, H,
                       W, C, OH, OW, (float)H / (float)OH, (float)W / (float)OW);
    return check_launch("resize_bilinear");
}

size_t acimg_filtfilt_workspace(int rows, int n) {
    return rows > 0 && n > 0 ? (size_t)rows * (n + 2 * FF_PAD) * sizeof(double) : 0;
}

int acimg_filtfilt(const void* x, int x_is_int32, int rows, int n, const double* ba, const double* zi, float* out,
                   void* ws, size_t ws_bytes, void* stream) {
    if (rows <= 0) return fail(ACIMG_EINVAL, "filtfilt: rows must be positive");
    if (n <= FF_PAD)
        return fail(ACIMG_EINVAL, "filtfilt: the signal must be longer than padlen = %d (scipy's ValueError)", FF_PAD);
    if (!ws || ws_bytes < acimg_filtfilt_workspace(rows, n))
        return fail(ACIMG_EWORKSPACE, "filtfilt: workspace %zu < %zu", ws_bytes, acimg_filtfilt_workspace(rows, n));
    if (x_is_int32)
        hipLaunchKernelGGL(filtfilt_kernel<int32_t>, dim3(cdiv(rows, 64)), dim3(64), 0, (hipStream_t)stream,
                           static_cast<const int32_t*>(x), rows, n, ba, zi, static_cast<double*>(ws), out);
    else
        hipLaunchKernelGGL(filtfilt_kernel<float>, dim3(cdiv(rows, 64)), dim3(64), 0, (hipStream_t)stream,
                           static_cast<const float*>(x), rows, n, ba, zi, static_cast<double*>(ws), out);
    return check_launch("filtfilt");
}

}  // extern "C"
