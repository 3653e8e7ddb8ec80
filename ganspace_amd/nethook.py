"""``InstrumentedModel``: retain / edit the output of named sub-modules.

Same protocol as the reference's ``netdissect/nethook.py:15-240`` (the piece of GAN Dissect
that GANSpace uses): ``retain_layer(s)``, ``retained_features()``, ``retained_layer()``,
``edit_layer(layer, ablation, replacement, offset)``, ``remove_edits()``, ``close()`` and the
``.model`` attribute.  The hot path only needs *retain*: the hooked module's output is kept as
a detached DEVICE tensor (nethook.py:211-217) that ``decomposition.compute`` hands straight to
the HIP Gram kernel - no host round trip.

Implementation differs from the reference (which monkey-patches ``layer.forward``): standard
``register_forward_hook`` handles are used, which also lets a hook replace the output.
"""
from collections import OrderedDict

import torch


def make_matching_tensor(valuedict, name, data):
    """Value for ``name`` as a tensor matching ``data`` (dtype/device, broadcastable channel dim);
    mirrors nethook.py:243-266."""
    v = valuedict.get(name, None)
    if v is None:
        return None
    if not isinstance(v, torch.Tensor):
        v = torch.from_numpy(__import__("numpy").asarray(v))
    if v.device != data.device or v.dtype != data.dtype:
        v = v.to(device=data.device, dtype=data.dtype)
        valuedict[name] = v
    if len(v.shape) < len(data.shape) and len(v.shape) > 0:
        v = v.view((1,) + tuple(v.shape) + (1,) * (len(data.shape) - len(v.shape) - 1))
        valuedict[name] = v
    return v


class InstrumentedModel(torch.nn.Module):
    def __init__(self, model):
        super().__init__()
        self.model = model
        self._retained = OrderedDict()
        self._ablation = {}
        self._replacement = {}
        self._offset = {}
        self._hooked_layer = {}     # aka -> layer name
        self._handles = {}          # aka -> hook handle

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def forward(self, *inputs, **kwargs):
        return self.model(*inputs, **kwargs)

    # ---- retain -------------------------------------------------------------------------
    def retain_layer(self, layername):
        self.retain_layers([layername])

    def retain_layers(self, layernames):
        self.add_hooks(layernames)
        for layername in layernames:
            aka = layername if isinstance(layername, str) else layername[1]
            if aka not in self._retained:
                self._retained[aka] = None

    def retained_features(self):
        return OrderedDict(self._retained)

    def retained_layer(self, aka=None, clear=False):
        if aka is None:
            aka = next(iter(self._retained.keys()))
        result = self._retained[aka]
        if clear:
            self._retained[aka] = None
        return result

    # ---- edit ----------------------------------------------------------------------------
    def edit_layer(self, layername, ablation=None, replacement=None, offset=None):
        layername, aka = (layername, layername) if isinstance(layername, str) else layername
        if ablation is None and replacement is not None:
            ablation = 1.0
        self.add_hooks([(layername, aka)])
        if ablation is not None:
            self._ablation[aka] = ablation
        if replacement is not None:
            self._replacement[aka] = replacement
        if offset is not None:
            self._offset[aka] = offset

    def remove_edits(self, layername=None, remove_offset=True, remove_replacement=True):
        if layername is None:
            if remove_replacement:
                self._ablation.clear()
                self._replacement.clear()
            if remove_offset:
                self._offset.clear()
            return
        aka = layername if isinstance(layername, str) else layername[1]
        if remove_replacement:
            self._ablation.pop(aka, None)
            self._replacement.pop(aka, None)
        if remove_offset:
            self._offset.pop(aka, None)

    # ---- hooks ---------------------------------------------------------------------------
    def add_hooks(self, layernames):
        needed, aka_map = set(), {}
        for name in layernames:
            aka = name
            if not isinstance(aka, str):
                name, aka = name
            if self._hooked_layer.get(aka, None) != name:
                aka_map[name] = aka
                needed.add(name)
        if not needed:
            return
        for name, layer in self.model.named_modules():
            if name in aka_map:
                needed.remove(name)
                self._hook_layer(layer, name, aka_map[name])
        for name in needed:
            raise ValueError("Layer %s not found in model" % name)

    def _hook_layer(self, layer, layername, aka):
        if aka in self._hooked_layer:
            raise ValueError("Layer %s already hooked" % aka)
        if layername in self._hooked_layer.values():
            raise ValueError("Layer %s already hooked" % layername)
        self._hooked_layer[aka] = layername

        def hook(module, inputs, output, aka=aka):
            return self._postprocess_forward(output, aka)

        self._handles[aka] = layer.register_forward_hook(hook)

    def _unhook_layer(self, aka):
        if aka not in self._hooked_layer:
            return
        self._handles.pop(aka).remove()
        del self._hooked_layer[aka]
        for d in (self._ablation, self._replacement, self._offset, self._retained):
            d.pop(aka, None)

    def _postprocess_forward(self, x, aka):
        if aka in self._retained:
            self._retained[aka] = x.detach()
        a = make_matching_tensor(self._ablation, aka, x)
        if a is not None:
            x = x * (1 - a)
            v = make_matching_tensor(self._replacement, aka, x)
            if v is not None:
                x = x + (v * a)
        b = make_matching_tensor(self._offset, aka, x)
        if b is not None:
            x = x + b
        return x

    def close(self):
        for aka in list(self._hooked_layer.keys()):
            self._unhook_layer(aka)
        assert len(self._handles) == 0
