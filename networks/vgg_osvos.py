"""Reference import path ``networks.vgg_osvos`` -> the MI355X implementation."""
from osvos_pytorch_amd.networks.vgg_osvos import *  # noqa: F401,F403
from osvos_pytorch_amd.networks.vgg_osvos import OSVOS, VGG, find_conv_layers, make_layers, make_layers_osvos  # noqa: F401
