"""Audio front end on the GPU: 1024-sample int32 frames -> 12 MFCCs (and the find_logen energy map).

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
        """audio_data: int32 [n,1024] (device) -> float32 [n,12] MFCCs; normalize=True also applies
        `_normalize_mfcc` (per-vector (x-min)/max)."""
        assert audio_data.dtype == torch.int32 and audio_data.shape[-1] == 1024
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
