"""Reference import path ``layers.osvos_layers`` -> the MI355X implementation."""
from osvos_pytorch_amd.layers.osvos_layers import (center_crop, class_balanced_cross_entropy_loss,  # noqa: F401
                                                   interp_surgery, logit, sigmoid_np, upsample_filt)
