"""GPU parity of the audio front end and find_logen against the golden vectors produced by the
REFERENCE's own NumPy code (tests/golden/frontend_golden.npz, see make_frontend_golden.py) and against
the oracle on random frames.  Tolerance: the kernel computes in fp64 like NumPy and rounds to float32
at the end, so results agree to 2e-6 relative + 2e-6 absolute (float32 rounding of the final cast)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "frontend_golden.npz"))


def test_mfcc_matches_reference_golden(device):
    from acimg.frontend import FrontEnd

    fe = FrontEnd(device)
    frames = torch.tensor(GOLD["frames"], dtype=torch.int32, device=device)
    got = fe._build_spectrograms_function(frames).cpu().numpy()
    np.testing.assert_allclose(got, GOLD["mfcc"], rtol=2e-6, atol=2e-6)
    # the low-pass ("silence") variant: the reference feeds float data; int32-rounded frames here
    lp = np.round(GOLD["lowpassed"]).astype(np.int32)
    from oracle import frontend as ofe
    got = fe._build_spectrograms_function(torch.tensor(lp, device=device)).cpu().numpy()
    np.testing.assert_allclose(got, ofe.mfcc(lp), rtol=2e-6, atol=2e-6)


def test_mfcc_random_frames_and_normalisation(device):
    from acimg.frontend import FrontEnd
    from oracle import frontend as ofe

    fe = FrontEnd(device)
    rng = np.random.RandomState(3)
    frames = (rng.randn(96, 1024) * rng.choice([1, 30, 1000, 30000], size=(96, 1))).astype(np.int32)
    frames[5] = 0                                   # all-zero frame: every mel energy at the 1e-3 floor
    t = torch.tensor(frames, device=device)
    got = fe._build_spectrograms_function(t).cpu().numpy()
    ref = ofe.mfcc(frames)
    np.testing.assert_allclose(got, ref, rtol=2e-6, atol=2e-6)
    gotn = fe._build_spectrograms_function(t, normalize=True).cpu().numpy()
    keep = np.arange(96) != 5                       # the zero frame normalises rounding noise by rounding noise
    np.testing.assert_allclose(gotn[keep], ofe.normalize_mfcc(ref)[keep], rtol=1e-5, atol=1e-6)


def test_find_logen_matches_reference_golden(device):
    from acimg.frontend import FrontEnd

    fe = FrontEnd(device)
    img = torch.tensor(GOLD["img32"], device=device)
    got = fe.find_logen(img).cpu().numpy()
    np.testing.assert_allclose(got, GOLD["logen32"], rtol=2e-6)
    img64 = torch.tensor(GOLD["img64"].astype(np.float32), device=device)
    np.testing.assert_allclose(fe.find_logen(img64).cpu().numpy(), GOLD["logen64"], rtol=5e-6)


def test_energy_iou_vs_oracle(device):
    """find_logen on real + generated image, mean-threshold masks, IoU (iouenergythreshold.py:213-229)"""
    from acimg import evaluate
    from oracle import frontend as ofe

    g = torch.Generator().manual_seed(31)
    N = 6
    real = torch.rand(N, 36, 48, 12, generator=g) * 4 - 2
    gen = real + 0.8 * torch.randn(N, 36, 48, 12, generator=g)
    gen[5] = real[5]                                  # identical maps -> IoU exactly 1
    ev = evaluate.EnergyIoU(device)
    iou = ev.iou(real.to(device), gen.to(device)).cpu().numpy()
    want = np.array([ofe.mask_iou(real[i].double().numpy(), gen[i].double().numpy()) for i in range(N)])
    assert iou[5] == 1.0
    # masks are thresholded at the mean: a pixel within float32 rounding of it may flip -> allow 2 pixels of 1728
    assert np.all(np.abs(iou - want) <= 2.5 / (36 * 48 * want.clip(0.05))), (iou, want)
    acc = evaluate.accuracy_curve(iou)
    assert acc[0] == 1.0 and acc[-1] == 0.0 and np.all(np.diff(acc) <= 0)


def test_lowpass_filtfilt_matches_reference_golden(device):
    """Device Butterworth-10 `filtfilt` (dataloader/outdoor_data_mfcc.py:565-575) against the REFERENCE's own output
    (`lowpassed`), and the MFCCs of the low-passed frames against the reference's `mfcc_lowpassed`: the "silence"
    variant of the audio input, end to end on the device."""
    from acimg.frontend import FrontEnd

    fe = FrontEnd(device)
    frames = torch.tensor(GOLD["frames"][:12], dtype=torch.int32, device=device)
    lp = fe.butter_lowpass_filter(frames)
    got = lp.cpu().numpy()
    # fp64 recurrence without fused multiply-adds, SciPy's operation order: bit-identical on the reference's vectors
    np.testing.assert_array_equal(got, GOLD["lowpassed"])
    mf = fe._build_spectrograms_function(lp).cpu().numpy()
    np.testing.assert_allclose(mf, GOLD["mfcc_lowpassed"], rtol=2e-6, atol=2e-6)
    # float32 input path, more rows than one workgroup, vs SciPy
    from scipy import signal
    rng = np.random.RandomState(9)
    x = (rng.randn(150, 1024) * 500).astype(np.float32)
    b, a = signal.butter(10, 125 / (0.5 * 12288), btype="low", analog=False)
    want = np.float32(signal.filtfilt(b, a, x))
    got = fe.butter_lowpass_filter(torch.tensor(x, device=device)).cpu().numpy()
    np.testing.assert_array_equal(got, want)     # NumPy extends float32 input in float32; so does the kernel
    # a signal no longer than the padding is refused, like scipy's ValueError
    from acimg import _lib
    with pytest.raises(_lib.AcimgError):
        fe.butter_lowpass_filter(torch.zeros(2, 33, dtype=torch.float32, device=device))


def test_stft_magnitude_and_resize_vs_oracle(device):
    """STFT-magnitude front end of the older audio path (dataloader/outdoor_data.py:571-596,844-851) and the bilinear
    resize of trainer/trainer.py:364-369 against the oracle's restatement of the published TensorFlow definitions
    (unpinned: no TensorFlow here).  1e-5 of the spectrogram's maximum."""
    from acimg.frontend import FrontEnd
    from oracle import frontend as ofe

    fe = FrontEnd(device)
    rng = np.random.RandomState(2)
    clips = 5
    samples = (rng.randn(clips, 12, 1024) * rng.choice([10, 1000, 20000], size=(clips, 1, 1))).astype(np.int32)
    wav, norm = fe.build_wav(torch.tensor(samples, device=device))
    want_wav = np.stack([ofe.build_wav(samples[i]) for i in range(clips)])
    spec = fe.stft_magnitude(wav, norm)
    assert tuple(spec.shape) == (clips, 99, 257)
    want = ofe.stft_mag(want_wav)
    got = spec.cpu().numpy()
    assert np.abs(got - want).max() <= 1e-5 * want.max(), np.abs(got - want).max() / want.max()
    # without the waveform normalisation (norm = None) the kernel windows the raw samples
    raw = fe.stft_magnitude(wav).cpu().numpy()
    want_raw = ofe.stft_mag(samples.reshape(clips, -1).astype(np.float32))
    assert np.abs(raw - want_raw).max() <= 1e-5 * want_raw.max()
    # a pure tone lands in its bin: 48 cycles per 512 samples -> bin 48
    t = np.arange(12288, dtype=np.float32)
    tone = torch.tensor(np.sin(2 * np.pi * 48 * t / 512)[None, :].astype(np.float32), device=device)
    assert int(fe.stft_magnitude(tone)[0, 10].argmax()) == 48
    # resize 99x257 -> 193x257 (trainer.py:367-369), and a case that also stretches the width
    x = spec.reshape(clips, 99, 257, 1)
    up = fe.resize_bilinear(x, (193, 257)).cpu().numpy()
    np.testing.assert_array_equal(up, ofe.resize_bilinear(got.reshape(clips, 99, 257, 1), 193, 257))
    y = torch.rand(2, 7, 5, 3, generator=torch.Generator().manual_seed(1))
    up2 = fe.resize_bilinear(y.to(device), (16, 12)).cpu().numpy()
    np.testing.assert_allclose(up2, ofe.resize_bilinear(y.numpy(), 16, 12), rtol=0, atol=1e-6)
    assert np.array_equal(up2[:, 0, 0], y.numpy()[:, 0, 0])          # dst (0,0) samples src (0,0) exactly


def test_unet_sound_trains_on_device_spectrograms(device):
    """`UNetSound` (models/unet_sound.py, 99x257x1) fed by the device STFT front end instead of synthetic maps: the
    train step's loss terms equal the oracle's on the oracle's own spectrogram of the same waveform (1e-3), and the
    loss falls over a few steps."""
    from acimg.frontend import FrontEnd
    from acimg.session import Session
    from acimg.trainer_vae import TrainerVAE
    from acimg.unet_vae import UNetSound
    from oracle import frontend as ofe
    from oracle import unet_vae as ouv

    N = 2
    fe = FrontEnd(device)
    rng = np.random.RandomState(4)
    samples = (rng.randn(N, 12, 1024) * 3000).astype(np.int32)
    wav, norm = fe.build_wav(torch.tensor(samples, device=device))
    spec = fe.stft_magnitude(wav, norm)                       # [N, 99, 257]
    spec = spec / spec.amax(dim=(1, 2), keepdim=True)         # per-sample max normalisation into [0, 1] (sigmoid output)
    sess = Session(device)
    tr = TrainerVAE(UNetSound(), learning_rate=1e-3, session=sess)
    tr._build_functions(batch_size=N)
    params = ouv.init_params("UNetSound", seed=7, dtype=torch.float64, bias_std=0.05, bn_jitter=0.1)
    tr.model.initialize(state={k: v.float() for k, v in params.items()})
    _, eps = ouv.synthetic_batch("UNetSound", N, seed=11, dtype=torch.float64)
    r = tr.train_step(spec.reshape(N, 99, 257, 1), eps.float().to(device), apply=False)
    want_spec = ofe.stft_mag(np.stack([ofe.build_wav(samples[i]) for i in range(N)]))
    want_spec = want_spec / want_spec.max(axis=(1, 2), keepdims=True)
    orc = ouv.Oracle("UNetSound", learning_rate=1e-3, dtype=torch.float64, params=params)
    ref = orc.train_step(torch.tensor(want_spec, dtype=torch.float64).reshape(N, 99, 257, 1), eps, apply=False)
    for k in ("mse", "huber", "latent", "loss"):
        assert abs(r[k] - ref["losses"][k]) <= 1e-3 * abs(ref["losses"][k]) + 1e-9, (k, r[k], ref["losses"][k])
    first = tr.train_step(None, eps.float().to(device), apply=True)
    for _ in range(5):
        last = tr.train_step(None, eps.float().to(device), apply=True)
    assert last["loss"] < first["loss"]
