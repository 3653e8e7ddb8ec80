"""Oracle restatement of the z -> activation layers on the path (TEST INFRASTRUCTURE).

* ``mapping_network``: the StyleGAN2 ``Generator.style`` MLP called from
  ``/root/reference/models/wrappers.py:177,200``.  The defining submodule
  (``models/stylegan2/stylegan2-pytorch``, ``.gitmodules:5-8``) is absent from
  the reference tree; SURVEY.md §A.5 restates its published definition
  (rosinality/stylegan2-pytorch ``model.py``: ``PixelNorm`` + 8 x
  ``EqualLinear(512, 512, lr_mul=0.01, activation='fused_lrelu')``).  The
  in-tree analogue ``models/stylegan/model.py:190-216`` (``G_mapping``) computes
  the same function for zero biases (``sqrt(2)*lrelu(Wx) == lrelu(sqrt(2)*Wx)``)
  and is what ``tests/golden/mapping_gmapping_ref.npz`` pins this against.
* ``biggan_gen_z``: ``cond = cat(z, embed)`` -> ``Linear(256, 32768)``
  (``wrappers.py:627-636`` -> ``models/biggan/.../model.py:211-212``).
"""
from __future__ import annotations

import numpy as np


def pixel_norm(x: np.ndarray, eps: float = 1e-8) -> np.ndarray:
    """``x * rsqrt(mean(x^2, dim=1) + eps)`` (stylegan/model.py:138-143)."""
    return x / np.sqrt(np.mean(x * x, axis=1, keepdims=True) + eps)


def equal_linear_lrelu(x, weight, bias, lr_mul=0.01, slope=0.2, gain=np.sqrt(2.0)):
    """One ``EqualLinear(..., activation='fused_lrelu')`` layer.

    ``weight`` is the stored parameter ``randn(out, in) / lr_mul``; the
    effective matrix is ``weight * (lr_mul / sqrt(in))``; the bias enters as
    ``bias * lr_mul``; activation ``gain * leaky_relu(., slope)``.
    """
    scale = lr_mul / np.sqrt(weight.shape[1])
    y = x @ (weight * scale).T + bias * lr_mul
    return gain * np.where(y >= 0, y, slope * y)


def mapping_network(z, weights, biases, lr_mul=0.01, dtype=np.float64):
    """PixelNorm followed by ``len(weights)`` fused-lrelu EqualLinear layers."""
    x = pixel_norm(np.asarray(z, dtype=dtype))
    for w, b in zip(weights, biases):
        x = equal_linear_lrelu(x, np.asarray(w, dtype=dtype), np.asarray(b, dtype=dtype), lr_mul)
    return x


def linear(x, weight, bias=None, dtype=np.float64):
    """``torch.nn.functional.linear``: ``x @ weight.T + bias``."""
    y = np.asarray(x, dtype=dtype) @ np.asarray(weight, dtype=dtype).T
    if bias is not None:
        y = y + np.asarray(bias, dtype=dtype)
    return y


def biggan_gen_z(z, class_embedding, weight, bias, dtype=np.float64):
    """``gen_z(cat(z, embed))`` for one fixed class (wrappers.py:627-636).

    ``class_embedding`` is the single 128-vector ``embeddings(one_hot)``.
    Returns the flat ``[n, out]`` activation the hook on ``generator.gen_z``
    retains.
    """
    z = np.asarray(z, dtype=dtype)
    e = np.broadcast_to(np.asarray(class_embedding, dtype=dtype), (z.shape[0], len(class_embedding)))
    return linear(np.concatenate([z, e], axis=1), weight, bias, dtype)
