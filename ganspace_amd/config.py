"""Run configuration with the reference's command-line surface.

Drop-in for the reference ``Config`` (``/root/reference/config.py:16-72``): the same option names,
flags and defaults (``--model --layer --class --est --sparsity --video --batch -b -c -n --use_w --sigma
--inputs --seed``), keyword overrides in the constructor, ``from_args`` / ``from_dict``, and a JSON
``str()`` that separates customised from default values.  The options live in ONE declarative table
below (flag, attribute, type, default, help); both the defaults and the argument parser are derived
from it, and knobs specific to this implementation can be appended without touching any call site.
"""
from __future__ import annotations

import argparse
import json
import sys
from collections import namedtuple

Option = namedtuple("Option", "flags dest kind default help")

OPTIONS = (
    Option(("--model",), "model", str, "StyleGAN", "generator to analyse (StyleGAN2, BigGAN-512, ...)"),
    Option(("--layer",), "layer", str, "g_mapping", "name of the layer whose activations are decomposed"),
    Option(("--class",), "output_class", str, None, "output class (BigGAN: ImageNet id, StyleGAN2: dataset)"),
    Option(("--est",), "estimator", str, "ipca", "estimator: ipca (sklearn-faithful) | ipca-exact"),
    Option(("--sparsity",), "sparsity", float, 1.0, "sparsity weight (SPCA of the reference; unused here)"),
    Option(("--video",), "make_video", bool, False, "render videos of the edits"),
    Option(("--batch",), "batch_mode", bool, False, "no windows: write results to files"),
    Option(("-b",), "batch_size", int, None, "generator mini-batch size (default: probe, at most 20)"),
    Option(("-c",), "components", int, 80, "number of principal components kept"),
    Option(("-n",), "n", int, 300_000, "number of latent samples fed to the PCA"),
    Option(("--use_w",), "use_w", bool, False, "decompose StyleGAN's W space instead of Z"),
    Option(("--sigma",), "sigma", float, 2.0, "edit strength in standard deviations (visualisation)"),
    Option(("--inputs",), "inputs", str, None, "directory with named, exported components"),
    Option(("--seed",), "seed", int, None, "seed of the latent stream (default 1)"),
)


def _parser() -> argparse.ArgumentParser:
    parser = argparse.ArgumentParser(description="GAN component analysis (MI355X)")
    for opt in OPTIONS:
        if opt.kind is bool:
            parser.add_argument(*opt.flags, dest=opt.dest, action="store_true", help=opt.help)
        else:
            parser.add_argument(*opt.flags, dest=opt.dest, type=opt.kind, default=opt.default, help=opt.help)
    return parser


class Config:
    """Attribute bag: table defaults, then keyword overrides."""

    def __init__(self, **overrides):
        self.default_args = {opt.dest: opt.default for opt in OPTIONS}
        self.from_dict(self.default_args)
        self.from_dict(overrides)

    def from_dict(self, values):
        for key, value in dict(values).items():
            setattr(self, key, value)
        return self

    def from_args(self, args=None):
        """Parse a command line (``sys.argv[1:]`` by default) on top of the current values."""
        parsed = _parser().parse_args(sys.argv[1:] if args is None else args)
        return self.from_dict(vars(parsed))

    def _split(self):
        custom, default = {}, {}
        for key, value in vars(self).items():
            if key == "default_args":
                continue
            untouched = key in self.default_args and self.default_args[key] == value
            (default if untouched else custom)[key] = value
        return custom, default

    def __str__(self):
        custom, default = self._split()
        return json.dumps({"custom": custom, "default": default}, indent=4)

    __repr__ = __str__
