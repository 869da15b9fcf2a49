"""Audio front end on the GPU: 1024-sample int32 frames -> 12 MFCCs (and the find_logen energy map), the Butterworth
`filtfilt` low-pass of the "silence" variant, and the STFT-magnitude spectrogram + bilinear resize of the older path.

Mirrors the reference's loader functions by name (dataloader/outdoor_data_mfcc.py):
`createfilters` (:826-849), `_build_spectrograms_function` (:796-824, + `get_feats` :851-876),
`_normalize_mfcc` (:696-703) and iouenergythreshold.py:294-323 `find_logen`.  The constant tables
(Tukey window, mel filter bank, DCT*norm*lifter) are built once on the host exactly as the reference
builds them per call; the per-frame arithmetic (window, FFT-1024, power, mel, log, DCT) is the HIP
kernel `acimg_mfcc_frontend` — fp64 inside like NumPy, float32 out.
"""
import numpy as np
import torch

from . import ops

LIFTER_NUM, FILTER_NUM, MFCC_NUM, FFT_LEN = 22, 24, 12, 512
LO_FREQ, HI_FREQ = 0, 6400


def tukey(n, alpha):
    """scipy.signal.tukey(n, alpha) (symmetric), restated so the host needs no SciPy"""
    if alpha <= 0:
        return np.ones(n)
    if alpha >= 1:
        return np.hanning(n)
    x = np.arange(0, n)
    width = int(np.floor(alpha * (n - 1) / 2.0))
    n1, n2, n3 = x[0:width + 1], x[width + 1:n - width - 1], x[n - width - 1:]
    w1 = 0.5 * (1 + np.cos(np.pi * (-1 + 2.0 * n1 / alpha / (n - 1))))
    w2 = np.ones(n2.shape)
    w3 = 0.5 * (1 + np.cos(np.pi * (-2.0 / alpha + 1 + 2.0 * n3 / alpha / (n - 1))))
    return np.concatenate((w1, w2, w3))


def createfilters(fft_len=FFT_LEN, filter_num=FILTER_NUM, lo_freq=LO_FREQ, hi_freq=HI_FREQ, samp_freq=2 * HI_FREQ):
    filter_mat = np.zeros((fft_len, filter_num))
    mel2freq = lambda mel: 700.0 * (np.exp(mel / 1127.0) - 1)  # noqa: E731
    freq2mel = lambda freq: 1127 * (np.log(1 + (freq / 700.0)))  # noqa: E731
    mel_c = np.linspace(freq2mel(lo_freq), freq2mel(hi_freq), filter_num + 2)
    point_c = np.floor(mel2freq(mel_c) / float(samp_freq) * (fft_len - 1) * 2).astype('int')
    for f in range(filter_num):
        d1 = point_c[f + 1] - point_c[f]
        d2 = point_c[f + 2] - point_c[f + 1]
        filter_mat[point_c[f]:point_c[f + 1] + 1, f] = np.linspace(0, 1, d1 + 1)
        filter_mat[point_c[f + 1]:point_c[f + 2] + 1, f] = np.linspace(1, 0, d2 + 1)
    return filter_mat


def butter_lowpass(cutoff=125, order=10, sample_rate=12288):
    """(b, a) of `signal.butter(order, cutoff / (0.5 * sample_rate), btype='low', analog=False)`
    (dataloader/outdoor_data_mfcc.py:565-569), restated with NumPy only, in SciPy's own order of operations:
    analog Butterworth prototype (buttap) -> frequency pre-warp + lp2lp_zpk -> bilinear_zpk (fs = 2) -> zpk2tf."""
    wn = cutoff / (0.5 * sample_rate)
    m = np.arange(-order + 1, order, 2)
    p = -np.exp(1j * np.pi * m / (2 * order))           # buttap: poles on the unit circle, k = 1
    k = 1.0
    fs = 2.0
    warped = 2 * fs * np.tan(np.pi * wn / fs)
    p = warped * p                                       # lp2lp_zpk (no zeros: degree = order)
    k = k * warped ** order
    fs2 = 2.0 * fs
    pz = (fs2 + p) / (fs2 - p)                           # bilinear_zpk
    zz = -np.ones(order)
    kz = k * np.real(1.0 / np.prod(fs2 - p))
    b = kz * np.poly(zz)                                 # zpk2tf
    a = np.real(np.poly(pz))
    return b, a


def lfilter_zi(b, a):
    """scipy.signal.lfilter_zi: the delay-line state of a unit step response's steady state,
    solve((I - companion(a)^T), b[1:] - a[1:] * b[0])"""
    b, a = np.atleast_1d(b) / a[0], np.atleast_1d(a) / a[0]
    n = max(len(a), len(b))
    comp = np.zeros((n - 1, n - 1))
    comp[0, :] = -a[1:]
    comp[1:, :-1] = np.eye(n - 2)
    return np.linalg.solve(np.eye(n - 1) - comp.T, b[1:] - a[1:] * b[0])


def hann_window_periodic(n):
    """tf.contrib.signal.hann_window(n, periodic=True) in float32: 0.5 - 0.5 cos(2 pi k / n)"""
    return (0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n) / float(n))).astype(np.float32)


STFT_FRAME, STFT_STEP, STFT_FFT = 246, 122, 512          # dataloader/outdoor_data.py:36-38


_TABLES = None


def tables():
    """float64 host tables: window[1024], melfb[512,24], dctl[24,12] (= dct_base*mfnorm*lifter),
    idct[12,24] (find_logen: mfnorm/lifter folded into dct_base^T)"""
    global _TABLES
    if _TABLES is None:
        dct_base = np.zeros((FILTER_NUM, MFCC_NUM))
        for m in range(MFCC_NUM):
            dct_base[:, m] = np.cos((m + 1) * np.pi / FILTER_NUM * (np.arange(FILTER_NUM) + 0.5))
        lifter = 1 + (LIFTER_NUM / 2) * np.sin(np.pi * (1 + np.arange(MFCC_NUM)) / LIFTER_NUM)
        mfnorm = np.sqrt(2.0 / FILTER_NUM)
        _TABLES = dict(window=tukey(1024, 0.75), melfb=createfilters(),
                       dctl=dct_base * mfnorm * lifter[None, :],
                       idct=(dct_base * (mfnorm / lifter)[None, :]).T.copy())
    return _TABLES


class FrontEnd(object):
    """Device-resident tables + launchers."""

    def __init__(self, device):
        self.device = torch.device(device)
        t = tables()
        self.window = torch.tensor(t["window"], dtype=torch.float64, device=self.device)
        self.melfb = torch.tensor(t["melfb"], dtype=torch.float64, device=self.device).contiguous()
        self.dctl = torch.tensor(t["dctl"], dtype=torch.float64, device=self.device).contiguous()
        self.idct = torch.tensor(t["idct"], dtype=torch.float64, device=self.device).contiguous()
        self.plan = ops.Plan(self.device, eager=True)

    def _build_spectrograms_function(self, audio_data, normalize=False, out=None):
        """audio_data: int32 (raw frames) or float32 (low-passed frames) [n,1024] on the device -> float32 [n,12]
        MFCCs; normalize=True also applies `_normalize_mfcc` (per-vector (x-min)/max)."""
        assert audio_data.dtype in (torch.int32, torch.float32) and audio_data.shape[-1] == 1024
        n = audio_data.numel() // 1024
        if out is None:
            out = torch.empty(n, MFCC_NUM, dtype=torch.float32, device=self.device)
        ops.mfcc_frontend(self.plan, audio_data.contiguous(), self.window, self.melfb, self.dctl, out, n, normalize)
        return out

    def find_logen(self, mfcc_img, out=None):
        """float32 [...,12] MFCC image -> energy map [...] (iouenergythreshold.py:294-323)"""
        x = mfcc_img.contiguous()
        pixels = x.numel() // 12
        if out is None:
            out = torch.empty(x.shape[:-1], dtype=torch.float32, device=self.device)
        ops.find_logen(self.plan, x, self.idct, out, pixels)
        return out

    # ---- "silence" variant: Butterworth-10 125 Hz zero-phase low-pass (outdoor_data_mfcc.py:558-575) ----------------
    def butter_lowpass_filter(self, data, cutoff=125, order=10, out=None):
        """data: [rows, n] int32 (raw audio frames, as the loader passes them) or float32, on the device ->
        float32 [rows, n] = np.float32(signal.filtfilt(b, a, data))"""
        key = (cutoff, order)
        if getattr(self, "_butter_key", None) != key:
            b, a = butter_lowpass(cutoff, order)
            self._ba = torch.tensor(np.concatenate([b, a]), dtype=torch.float64, device=self.device)
            self._zi = torch.tensor(lfilter_zi(b, a), dtype=torch.float64, device=self.device)
            self._butter_key = key
        x = data.contiguous()
        assert x.dtype in (torch.int32, torch.float32) and x.dim() == 2
        rows, n = x.shape
        if out is None:
            out = torch.empty(rows, n, dtype=torch.float32, device=self.device)
        ops.filtfilt(self.plan, x, rows, n, self._ba, self._zi, out)
        return out

    # ---- STFT-magnitude path (dataloader/outdoor_data.py:571-596,844-851; trainer/trainer.py:364-369) ---------------
    def _stft_tables(self):
        if not hasattr(self, "_hann"):
            self._hann = torch.tensor(hann_window_periodic(STFT_FRAME), device=self.device)
            j = np.arange(STFT_FFT // 2)
            tw = np.stack([np.cos(2 * np.pi * j / STFT_FFT), -np.sin(2 * np.pi * j / STFT_FFT)], 1)
            self._tw = torch.tensor(tw.astype(np.float32), device=self.device).contiguous()
        return self._hann, self._tw

    def build_wav(self, audio_samples):
        """`_build_wav_py_function`: [clips, ...] samples -> float32 [clips, nsamples]; returns (wav, max |wav| per
        clip).  The division itself happens inside the STFT kernel's loader (same operation, same order)."""
        clips = audio_samples.shape[0]
        wav = audio_samples.reshape(clips, -1).to(torch.float32).contiguous()
        norm = torch.empty(clips, dtype=torch.float32, device=self.device)
        ops.absmax(self.plan, wav, clips, wav.shape[1], norm)
        return wav, norm

    def stft_magnitude(self, wav, norm=None, out=None):
        """`_map_func_audio_samples_build_spectrogram`: float32 [clips, nsamples] -> [clips, frames, 257] magnitudes
        (nsamples = 12288 -> 99 frames)"""
        win, tw = self._stft_tables()
        clips, ns = wav.shape
        frames = 1 + (ns - STFT_FRAME) // STFT_STEP
        if out is None:
            out = torch.empty(clips, frames, STFT_FFT // 2 + 1, dtype=torch.float32, device=self.device)
        ops.stft_mag(self.plan, wav.contiguous(), norm, win, tw, out, clips, ns, STFT_FRAME, STFT_STEP, STFT_FFT)
        return out

    def resize_bilinear(self, x, size, out=None):
        """tf.image.resize_bilinear(x, size, align_corners=False): NHWC float32"""
        N, H, W, Cn = x.shape
        if out is None:
            out = torch.empty(N, size[0], size[1], Cn, dtype=torch.float32, device=self.device)
        ops.resize_bilinear(self.plan, x.contiguous(), out, N, H, W, Cn, size[0], size[1])
        return out
