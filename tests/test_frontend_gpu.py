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
