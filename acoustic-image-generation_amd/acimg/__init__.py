"""acimg — MI355X-native acoustic-image generation (hot path of IIT-PAVIS/Acoustic-Image-Generation).

Host side in Python on PyTorch-ROCm (allocation, streams, torch.distributed); arithmetic in
hand-written HIP kernels behind the C ABI of include/acimg.h (libacimg.so).
"""
__version__ = "0.1.0"
