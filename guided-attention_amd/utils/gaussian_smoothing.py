"""`GaussianSmoothing` with the reference's constructor and call convention
(utils/gaussian_smoothing.py:8-71): a fixed, normalised Gaussian applied per channel; the caller
reflect-pads (pipeline_guided_attention.py:252-254).  The weights come from the library's host
helper `ga_gaussian_weights` (same fp32 recipe as the reference, including its -((x-mu)/(2 sigma))^2
exponent).  On the loss hot path the smoothing is fused inside `ga_smooth_loss_fwd`; this module is
the stand-alone operator for callers that use it directly."""
import numbers

import torch
from torch import nn
from torch.nn import functional as F

from .. import ops


class GaussianSmoothing(nn.Module):
    def __init__(self, channels, kernel_size, sigma, dim=2):
        super().__init__()
        if dim != 2:
            raise RuntimeError("Only 2 dimensions are supported on this path. Received {}.".format(dim))
        if not isinstance(kernel_size, numbers.Number) or not isinstance(sigma, numbers.Number):
            ks, sg = list(kernel_size), list(sigma)
            if len(set(ks)) != 1 or len(set(sg)) != 1:
                raise RuntimeError("anisotropic kernels are not supported on this path")
            kernel_size, sigma = ks[0], sg[0]
        kernel = ops.gaussian_weights(int(kernel_size), float(sigma))
        self.register_buffer("weight", kernel.view(1, 1, *kernel.shape).repeat(channels, 1, 1, 1))
        self.groups = channels
        self.conv = F.conv2d

    def forward(self, input):
        return self.conv(input, weight=self.weight.to(input.dtype), groups=self.groups)
