"""``InstrumentedModel``: observe and edit the outputs of named sub-modules of a generator.

Same protocol as the piece of GAN Dissect that GANSpace uses
(``/root/reference/netdissect/nethook.py:15-240``): ``retain_layer(s)`` / ``retained_features()`` /
``retained_layer()`` to harvest activations, ``edit_layer(layer, ablation, replacement, offset)`` /
``remove_edits()`` to steer them, ``close()``, and the wrapped network as ``.model``.  The component
discovery loop only needs the *retain* half: the hooked module's output is kept as a detached DEVICE
tensor and goes straight into the HIP Gram kernel - no host round trip.

Design: every instrumented layer gets one ``_Tap`` (PyTorch forward hook + its retain/edit state).
The reference instead swaps each layer's bound ``forward`` for a closure; hooks compose with other
PyTorch machinery and are removed by handle.
"""
from __future__ import annotations

from collections import OrderedDict

import numpy as np
import torch


def make_matching_tensor(valuedict, name, data):
    """``valuedict[name]`` as a tensor compatible with ``data`` (device, dtype; a lower-rank value is
    treated as per-channel and reshaped to ``[1, C, 1, ...]``).  The converted tensor is cached back."""
    value = valuedict.get(name)
    if value is None:
        return None
    if not torch.is_tensor(value):
        value = torch.as_tensor(np.asarray(value))
    if value.device != data.device or value.dtype != data.dtype:
        value = value.to(device=data.device, dtype=data.dtype)
    if 0 < value.dim() < data.dim():
        value = value.reshape((1,) + tuple(value.shape) + (1,) * (data.dim() - value.dim() - 1))
    valuedict[name] = value
    return value


class _Tap:
    """Forward hook on one module: optionally keeps the raw output, then applies the edits."""

    __slots__ = ("owner", "key", "layername", "handle")

    def __init__(self, owner, module, layername, key):
        self.owner, self.key, self.layername = owner, key, layername
        self.handle = module.register_forward_hook(self)

    def __call__(self, module, inputs, output):
        own, key = self.owner, self.key
        if key in own._retained:
            own._retained[key] = output.detach()          # retained BEFORE any edit
        alpha = make_matching_tensor(own._ablation, key, output)
        if alpha is not None:
            output = output * (1 - alpha)
            fill = make_matching_tensor(own._replacement, key, output)
            if fill is not None:
                output = output + fill * alpha
        shift = make_matching_tensor(own._offset, key, output)
        if shift is not None:
            output = output + shift
        return output

    def remove(self):
        self.handle.remove()


def _split_name(layername):
    """``'layer'`` or ``('layer', 'alias')`` -> (layer, alias)."""
    return (layername, layername) if isinstance(layername, str) else tuple(layername)


class InstrumentedModel(torch.nn.Module):
    def __init__(self, model):
        super().__init__()
        self.model = model
        self._taps = {}                 # alias -> _Tap
        self._retained = OrderedDict()  # alias -> last output (None until the model has run)
        self._ablation, self._replacement, self._offset = {}, {}, {}

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def forward(self, *inputs, **kwargs):
        return self.model(*inputs, **kwargs)

    # ---- observing ----------------------------------------------------------------------------
    def retain_layer(self, layername):
        self.retain_layers([layername])

    def retain_layers(self, layernames):
        self.add_hooks(layernames)
        for name in layernames:
            self._retained.setdefault(_split_name(name)[1], None)

    def retained_features(self):
        """Snapshot (ordered copy) of everything retained so far."""
        return OrderedDict(self._retained)

    def retained_layer(self, aka=None, clear=False):
        if aka is None:
            aka = next(iter(self._retained))
        value = self._retained[aka]
        if clear:
            self._retained[aka] = None
        return value

    # ---- editing --------------------------------------------------------------------------------
    def edit_layer(self, layername, ablation=None, replacement=None, offset=None):
        """``out = x * (1 - a) + r * a`` (``a`` defaults to 1 when only ``r`` is given), then ``+ offset``."""
        layer, aka = _split_name(layername)
        self.add_hooks([(layer, aka)])
        if ablation is None and replacement is not None:
            ablation = 1.0
        for store, value in ((self._ablation, ablation), (self._replacement, replacement), (self._offset, offset)):
            if value is not None:
                store[aka] = value

    def remove_edits(self, layername=None, remove_offset=True, remove_replacement=True):
        akas = None if layername is None else [_split_name(layername)[1]]
        stores = ([self._ablation, self._replacement] if remove_replacement else []) + \
                 ([self._offset] if remove_offset else [])
        for store in stores:
            if akas is None:
                store.clear()
            else:
                for aka in akas:
                    store.pop(aka, None)

    # ---- hook management ------------------------------------------------------------------------
    def add_hooks(self, layernames):
        wanted = {}
        for name in layernames:
            layer, aka = _split_name(name)
            tap = self._taps.get(aka)
            if tap is None:
                wanted[layer] = aka
            elif tap.layername != layer:
                raise ValueError("Layer %s already hooked" % aka)
        if not wanted:
            return
        hooked_layers = {t.layername for t in self._taps.values()}
        for name, module in self.model.named_modules():
            aka = wanted.pop(name, None)
            if aka is None:
                continue
            if name in hooked_layers:
                raise ValueError("Layer %s already hooked" % name)
            self._taps[aka] = _Tap(self, module, name, aka)
        for name in wanted:
            raise ValueError("Layer %s not found in model" % name)

    def close(self):
        """Detach every hook and forget all retained values and edits."""
        for tap in self._taps.values():
            tap.remove()
        self._taps.clear()
        for store in (self._retained, self._ablation, self._replacement, self._offset):
            store.clear()
