import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG_PARENT = os.path.join(ROOT, "acoustic-image-generation_amd")
for p in (ROOT, PKG_PARENT):
    if p not in sys.path:
        sys.path.insert(0, p)


def _host_cores():
    """host cores this process can really use: affinity mask capped by the cgroup CPU quota (a one-GPU box gives the job
    e.g. 16 of 128 cores; PyTorch's default of one thread per visible core then runs the CPU oracle ~4x slower)"""
    n = len(os.sched_getaffinity(0))
    for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(f).read().split()
            if f.endswith("cpu.max"):
                quota, period = txt[0], float(txt[1])
            else:
                quota, period = txt[0], float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota not in ("max", "-1"):
                n = max(1, min(n, int(float(quota) / period + 0.5)))
            break
        except Exception:
            continue
    return n


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    import torch

    torch.set_num_threads(_host_cores())      # the CPU oracle is the long pole of the -m gpu suite


@pytest.fixture(scope="session")
def device():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")
