"""osvos-pytorch_amd: MI355X (gfx950) native OSVOS forward/backward hot path.

Python host side (autograd + SGD bookkeeping on PyTorch-ROCm) over a C-ABI HIP library
(``libosvos_hip.so``, sources in ``csrc/``, ABI in ``include/osvos_hip.h``).  Import it as
``osvos_pytorch_amd``; the reference's own import paths (``networks.vgg_osvos``,
``layers.osvos_layers``, ``mypath``) are provided by thin shims at the repository root.
"""
__version__ = "0.1.0"
