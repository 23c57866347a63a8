"""Shared closed-form test inputs (same formulas as tests/golden/make_goldens.py)."""
import numpy as np
import torch

from oracle import nets as onet


def hashed(idx, shape, scale=1.0):
    return (onet._hash_uniform(idx, int(np.prod(shape))) * scale).reshape(shape)


def spec_input(idx, B, T, F=256):
    u = hashed(idx, (B, 2, F, T))
    env = 2.0 / (1.0 + np.arange(F, dtype=np.float64) / 16.0)
    return torch.from_numpy((u * env[None, None, :, None]).astype(np.float32))


def silent_gate(x):
    T = x.shape[-1]
    g = ((np.arange(T) // 10) % 3 == 0).astype(np.float32)
    return x * torch.from_numpy(g)[None, None, None, :]


def _np(a):
    if isinstance(a, torch.Tensor):
        a = a.detach().cpu().numpy()
    return np.asarray(a, dtype=np.float64)


def rel_err(a, b):
    a = _np(a)
    b = _np(b)
    return float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-30))
