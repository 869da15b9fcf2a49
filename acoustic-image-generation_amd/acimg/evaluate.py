"""Energy-map localisation metric, the step right after the generator in every evaluation script of the
reference: `find_logen` on the real and the generated acoustic image, mean-threshold masks,
IoU, hit if IoU > tau (iouenergythreshold.py:213-236); accuracy over tau in {0, .1, ..., 1} integrated with the
trapezoid rule (areaundercurve.py:26-40, sklearn.metrics.auc).  The per-sample work (2 x 1728 inverse-DCT +
exp pixels, two means, two mask counts) runs on the GPU; the 11-point curve is host arithmetic."""
import numpy as np
import torch

from . import ops
from .frontend import FrontEnd

THRESHOLDS = [0.0, 0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9, 1.0]


class EnergyIoU(object):
    def __init__(self, device):
        self.device = torch.device(device)
        self.fe = FrontEnd(self.device)

    def iou(self, real, generated):
        """real, generated: float32 [N,36,48,12] device tensors -> IoU per sample [N] (device)"""
        N = real.shape[0]
        P = real.shape[1] * real.shape[2]
        a = self.fe.find_logen(real)
        b = self.fe.find_logen(generated)
        out = torch.empty(N, dtype=torch.float32, device=self.device)
        ops.mask_iou(self.fe.plan, a, b, N, P, out)
        return out


def accuracy_curve(ious, thresholds=THRESHOLDS):
    """fraction of samples with IoU > tau, per tau (iouenergythreshold.py:226-236)"""
    v = np.asarray(ious, dtype=np.float64)
    return np.array([float(np.mean(v > t)) for t in thresholds])


def area_under_curve(acc, thresholds=THRESHOLDS):
    """sklearn.metrics.auc on the reversed lists, as areaundercurve.py:36-40 calls it = trapezoid rule"""
    x = np.asarray(thresholds, dtype=np.float64)[::-1]
    y = np.asarray(acc, dtype=np.float64)[::-1]
    return float(abs(np.sum((x[1:] - x[:-1]) * (y[1:] + y[:-1]) * 0.5)))
