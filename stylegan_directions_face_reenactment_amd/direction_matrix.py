"""DirectionMatrix ``A``: linear map from the 3DMM parameter difference (15 = yaw, pitch, roll, jaw +
11 expression coefficients) to a shift in W / W+ -- drop-in for the reference's
``libs/models/direction_matrix.py`` (constructor signature :7-8, attributes ``shift_dim``, ``input_dim``,
``out_dim``, ``w_plus``, ``num_layers``, parameter container ``linear`` with keys ``linear.weight`` /
``linear.bias`` so saved ``A_matrix`` state_dicts load unchanged; forward semantics :41-48).

The matmul and all three of its gradients run on the HIP linear kernel (functional.affine), so the
direction-learning step can optimise ``A`` without leaving the library.  ``np.product`` (removed in
NumPy 2, direction_matrix.py:11-12) is not used.
"""
import math

import torch
from torch import nn

from . import functional as F_


def _numel(shape):
    return int(math.prod(shape)) if isinstance(shape, (tuple, list)) else int(shape)


class DirectionMatrix(nn.Module):
    def __init__(self, shift_dim, input_dim=None, out_dim=None, inner_dim=512, bias=True, w_plus=False,
                 num_layers=14, initialization='normal', verbose=True):
        super().__init__()
        self.shift_dim = shift_dim
        self.input_dim = input_dim if input_dim is not None else _numel(shift_dim)
        self.out_dim = out_dim if out_dim is not None else _numel(shift_dim)
        self.w_plus = w_plus
        self.num_layers = num_layers
        if verbose:   # the reference prints its configuration at construction (:16-21)
            print('Linear Direction matrix-A {}: input dimension {}, output dimension {}, shift dimension {} '.format(
                'in w+ space' if w_plus else 'type ', self.input_dim, self.out_dim, self.shift_dim))
        rows = self.out_dim * num_layers if w_plus else self.out_dim
        self.linear = nn.Linear(self.input_dim, rows, bias=bias)
        with torch.no_grad():
            self.linear.weight.zero_()
            if initialization == 'normal':
                self.linear.weight.normal_(mean=0.0, std=0.03)
            elif initialization == 'eye':
                n = int(min(self.input_dim, rows))
                blocks = range(num_layers) if w_plus else range(1)
                for b in blocks:
                    self.linear.weight[b * self.out_dim:b * self.out_dim + n, :n] = torch.eye(n)

    def forward(self, input):
        x = input.reshape(-1, self.input_dim)
        out = F_.affine(x, self.linear.weight, self.linear.bias)
        if self.w_plus:
            out = out.view(x.shape[0], self.num_layers, self.shift_dim)
        return out
