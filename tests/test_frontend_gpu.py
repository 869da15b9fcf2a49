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
