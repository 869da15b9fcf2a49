"""Generates tests/golden/frontend_golden.npz by running the REFERENCE's own NumPy front-end code
(dataloader/outdoor_data_mfcc.py, iouenergythreshold.py) in the build container.

TensorFlow / cv2 / librosa / torchfile are not installed, so they are stubbed with MagicMock module
objects; only pure NumPy/SciPy functions of the reference are executed.  scipy >= 1.13 moved
signal.tukey to signal.windows.tukey; the alias is restored before import.  Nothing from the
reference is copied: the fixture holds inputs and the outputs the reference computed from them.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_frontend_golden.py
"""
import importlib.abc
import importlib.machinery
import os
import sys
from unittest import mock

import numpy as np

REF = "/root/reference"
STUBS = ("tensorflow", "cv2", "torchfile", "librosa", "matplotlib", "sklearn")


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, name, path, target=None):
        if name.split(".")[0] in STUBS:
            return importlib.machinery.ModuleSpec(name, self, is_package=True)
        return None

    def create_module(self, spec):
        m = mock.MagicMock(name=spec.name)
        m.__path__ = []
        m.__name__ = spec.name
        m.__spec__ = spec
        return m

    def exec_module(self, module):
        pass


def main():
    sys.dont_write_bytecode = True
    sys.meta_path.insert(0, _StubFinder())
    import scipy.signal
    import scipy.signal.windows

    scipy.signal.tukey = scipy.signal.windows.tukey
    sys.path.insert(0, REF)
    from dataloader.outdoor_data_mfcc import ActionsDataLoader  # noqa: E402
    import iouenergythreshold  # noqa: E402

    loader = object.__new__(ActionsDataLoader)
    loader.sample_rate = 12288
    rng = np.random.RandomState(0)
    frames = (rng.randn(12, 1024) * 1000).astype(np.int32)
    quiet = np.zeros((2, 1024), np.int32)           # exercises the 1e-3 floor
    quiet[1, ::7] = 3
    frames_all = np.concatenate([frames, quiet], 0)
    mfcc = loader._build_spectrograms_function(frames_all)
    filt = loader.createfilters(512, 24, 0, 6400, 12800)
    lowpassed = loader.butter_lowpass_filter(frames.astype(np.float64))
    mfcc_lp = loader._build_spectrograms_function(lowpassed)
    img64 = rng.rand(36, 48, 12)
    logen64 = iouenergythreshold.find_logen(img64.copy())
    img32 = rng.rand(36, 48, 12).astype(np.float32)
    logen32 = iouenergythreshold.find_logen(img32.copy())
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "frontend_golden.npz")
    np.savez_compressed(out, frames=frames_all, mfcc=mfcc, filters=filt, lowpassed=lowpassed,
                        mfcc_lowpassed=mfcc_lp, img64=img64, logen64=logen64, img32=img32,
                        logen32=np.asarray(logen32))
    print("wrote", out, mfcc.shape, mfcc.dtype, float(mfcc[:12].sum()))


if __name__ == "__main__":
    main()
