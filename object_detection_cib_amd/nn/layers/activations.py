"""Activation callables of the reference's layer constructors (kod/nn/layers/activations.py:7): `SiLUInplace` is what
CSPBlock / CSPLayer / SPPFBottleneck / Yolov5Network take as their default `activation_layer`.  On the HIP path SiLU is fused
into the BatchNorm apply kernels (csrc/bn_act.hip); the callable only has to identify itself as SiLU (nn/graph_module.py
check_norm_act)."""
import functools

import torch.nn as nn

SiLUInplace = functools.partial(nn.SiLU, inplace=True)
