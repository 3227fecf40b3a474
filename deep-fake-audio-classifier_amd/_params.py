"""Parameter containers that reproduce the reference models' state_dict keys, shapes, parameter order and default
initialisation WITHOUT building torch compute layers: the containers hold tensors, the HIP kernels do the math.

Initialisation follows what torch.nn.{Conv2d,Conv1d,ConvTranspose2d,Linear,BatchNorm*} do at construction
(kaiming_uniform_(a=sqrt(5)) on the weight, U(-1/sqrt(fan_in), 1/sqrt(fan_in)) on the bias, BN = (1, 0, 0, 1, 0)),
drawing from the global torch RNG in the same order, so `torch.manual_seed(s); CNN2D()` yields the same initial
weights as the reference class under the same seed (src/model.py:12-31)."""
from __future__ import annotations

import math

import torch
from torch import nn


class ConvParams(nn.Module):
    """weight [Cout, Cin, *k] + bias [Cout]  (Conv1d / Conv2d).  transposed=True: weight [Cin, Cout, *k]."""

    def __init__(self, cin: int, cout: int, ksize: tuple, transposed: bool = False):
        super().__init__()
        shape = (cin, cout, *ksize) if transposed else (cout, cin, *ksize)
        self.weight = nn.Parameter(torch.empty(shape))
        self.bias = nn.Parameter(torch.empty(cout))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        fan_in = shape[1] * math.prod(ksize)          # torch's _calculate_fan_in_and_fan_out uses dim 1
        bound = 1.0 / math.sqrt(fan_in) if fan_in > 0 else 0.0
        nn.init.uniform_(self.bias, -bound, bound)


class LinearParams(nn.Module):
    def __init__(self, fin: int, fout: int):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(fout, fin))
        self.bias = nn.Parameter(torch.empty(fout))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        bound = 1.0 / math.sqrt(fin) if fin > 0 else 0.0
        nn.init.uniform_(self.bias, -bound, bound)


class BatchNormParams(nn.Module):
    eps = 1e-5
    momentum = 0.1

    def __init__(self, c: int):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))


class Slots(nn.Module):
    """A container whose children are named by their index in the reference's nn.Sequential ('0', '1', '5', ...);
    indices occupied there by parameter-free layers (ReLU, pooling, dropout) are simply absent."""

    def __init__(self, children: dict):
        super().__init__()
        for idx in sorted(children):
            self.add_module(str(idx), children[idx])

    def __getitem__(self, idx):
        return self._modules[str(idx)]


def tensors_signature(tensors):
    """(data_ptr, version) of every tensor: changes whenever a weight is replaced or modified in place."""
    return tuple((t.data_ptr(), t._version) for t in tensors)
