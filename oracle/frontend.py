"""Oracle: the NumPy audio front end and the find_logen energy map.  TEST INFRASTRUCTURE.

Restates dataloader/outdoor_data_mfcc.py:
  :796-824  _build_spectrograms_function  (Tukey(.75) window, rFFT-1024 without the Nyquist bin,
            power, mel bank built for fft_len=512 / samp_freq=12800, 12 MFCCs, float32 out)
  :826-849  createfilters                 (24 triangular filters, 0-6400 Hz, mel scale 1127/700)
  :851-876  get_feats                     (floor 1e-3, log, DCT*sqrt(2/24), lifter 22, nan/inf -> 0)
  :696-703  _normalize_mfcc               (per-vector (x-min)/max(x-min), float32)
  :565-575  butter_lowpass / butter_lowpass_filter (order-10 125 Hz, filtfilt)
and iouenergythreshold.py:294-323 find_logen.
Pinned by tests/golden/frontend_golden.npz (made from the reference's own functions).

`stft_mag` / `resize_bilinear` restate the TensorFlow ops of the older spectrogram path
(dataloader/outdoor_data.py:844-851, trainer/trainer.py:364-369) from their published definitions
(tf.contrib.signal.stft: periodic Hann window, frames zero-padded to fft_length, rfft, pad_end=False;
tf.image.resize_bilinear TF-1, align_corners=False: src = dst * in/out, no half-pixel centres).  TensorFlow is not
installed here and the reference holds no fixture for them: PARITY UNPINNED for these two functions.
"""
import numpy as np
from scipy import signal

LIFTER_NUM = 22
FILTER_NUM = 24
MFCC_NUM = 12
FFT_LEN = 512
LO_FREQ, HI_FREQ = 0, 6400
SAMPLE_RATE = 12288


def tukey_window(n=1024, alpha=0.75):
    return signal.windows.tukey(n, alpha=alpha)


def createfilters(fft_len=FFT_LEN, filter_num=FILTER_NUM, lo_freq=LO_FREQ, hi_freq=HI_FREQ,
                  samp_freq=2 * HI_FREQ):
    filter_mat = np.zeros((fft_len, filter_num))
    mel2freq = lambda mel: 700.0 * (np.exp(mel / 1127.0) - 1)
    freq2mel = lambda freq: 1127 * (np.log(1 + (freq / 700.0)))
    mel_c = np.linspace(freq2mel(lo_freq), freq2mel(hi_freq), filter_num + 2)
    freq_c = mel2freq(mel_c)
    point_c = np.floor(freq_c / float(samp_freq) * (fft_len - 1) * 2).astype("int")
    for f in range(filter_num):
        d1 = point_c[f + 1] - point_c[f]
        d2 = point_c[f + 2] - point_c[f + 1]
        filter_mat[point_c[f]:point_c[f + 1] + 1, f] = np.linspace(0, 1, d1 + 1)
        filter_mat[point_c[f + 1]:point_c[f + 2] + 1, f] = np.linspace(1, 0, d2 + 1)
    return filter_mat


def dct_base():
    base = np.zeros((FILTER_NUM, MFCC_NUM))
    for m in range(MFCC_NUM):
        base[:, m] = np.cos((m + 1) * np.pi / FILTER_NUM * (np.arange(FILTER_NUM) + 0.5))
    return base


def lifter():
    return 1 + (LIFTER_NUM / 2) * np.sin(np.pi * (1 + np.arange(MFCC_NUM)) / LIFTER_NUM)


MFNORM = np.sqrt(2.0 / FILTER_NUM)


def mfcc(audio_data):
    """int32 [n,1024] -> float32 [n,12]  (_build_spectrograms_function + get_feats)"""
    audio_data = np.asarray(audio_data)
    raw = audio_data * tukey_window()[None, :]
    fftdata = np.abs(np.fft.rfft(raw, 1024, axis=1))[:, :-1] ** 2
    melspec = np.dot(fftdata, createfilters())
    melspec[melspec < 0.001] = 0.001
    melspec = np.log(melspec)
    c = np.dot(melspec, dct_base())
    c *= MFNORM
    c *= lifter()
    c[np.isnan(c)] = 0
    c[np.isinf(c)] = 0
    return np.float32(c)


def normalize_mfcc(v):
    """per-vector min-max in float32 (_normalize_mfcc)"""
    v = np.float32(v)
    v = v - v.min(axis=-1, keepdims=True)
    return v / v.max(axis=-1, keepdims=True)


def butter_lowpass_filter(data, cutoff=125, order=10):
    b, a = signal.butter(order, cutoff / (0.5 * SAMPLE_RATE), btype="low", analog=False)
    return np.float32(signal.filtfilt(b, a, data))


def find_logen(mfcc_img):
    """[...,12] MFCC image -> energy map (iouenergythreshold.py:294-323); float64 throughout when the
    input is float64."""
    m = np.array(mfcc_img, copy=True).reshape(-1, 12)
    m /= np.expand_dims(lifter(), 0)
    m *= MFNORM
    melspec = np.exp(np.dot(m, np.transpose(dct_base())))
    return 1 / np.sum(melspec, -1)


def mask_iou(real_img, gen_img):
    """iouenergythreshold.py:213-229 for one sample: [36,48,12] x 2 -> IoU of the mean-threshold masks"""
    m1 = find_logen(real_img)
    m2 = find_logen(gen_img)
    a = m1 > np.mean(m1)
    b = m2 > np.mean(m2)
    return np.sum(np.logical_and(a, b)) / np.sum(np.logical_or(a, b))


def build_wav(audio_samples):
    """dataloader/outdoor_data.py:577-596 `_build_wav_py_function`: float32, flatten, divide by max |.|"""
    w = np.asarray(audio_samples).astype(np.float32).flatten("C")
    return w / abs(max(w.min(), w.max(), key=abs))


def stft_mag(wav, frame_length=246, frame_step=122, fft_length=512):
    """float32 [..., nsamples] -> float64-accurate |STFT| rounded to float32, [..., frames, fft_length//2+1]"""
    wav = np.asarray(wav, np.float32)
    n = wav.shape[-1]
    frames = 1 + (n - frame_length) // frame_step
    win = (0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(frame_length) / frame_length)).astype(np.float32)
    idx = np.arange(frame_length)[None, :] + frame_step * np.arange(frames)[:, None]
    fr = (wav[..., idx] * win).astype(np.float32)                     # TF multiplies in float32
    return np.abs(np.fft.rfft(fr.astype(np.float64), fft_length, axis=-1)).astype(np.float32)


def resize_bilinear(x, oh, ow):
    """NHWC float32, TF-1 legacy sampling, float32 arithmetic in the order of TF's kernel"""
    x = np.asarray(x, np.float32)
    N, H, W, C = x.shape
    hs, ws = np.float32(H) / np.float32(oh), np.float32(W) / np.float32(ow)

    def axis(out, size, scale):
        src = (np.arange(out, dtype=np.float32) * scale).astype(np.float32)
        lo = np.floor(src).astype(np.int64)
        hi = np.minimum(lo + 1, size - 1)
        return lo, hi, (src - lo.astype(np.float32)).astype(np.float32)

    y0, y1, yl = axis(oh, H, hs)
    x0, x1, xl = axis(ow, W, ws)
    xl = xl[None, None, :, None]
    yl = yl[None, :, None, None]
    tl, tr = x[:, y0][:, :, x0], x[:, y0][:, :, x1]
    bl, br = x[:, y1][:, :, x0], x[:, y1][:, :, x1]
    top = (tl + ((tr - tl) * xl).astype(np.float32)).astype(np.float32)
    bot = (bl + ((br - bl) * xl).astype(np.float32)).astype(np.float32)
    return (top + ((bot - top) * yl).astype(np.float32)).astype(np.float32)
