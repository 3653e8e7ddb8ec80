"""``Config``: argparse-backed option bag, same flags/defaults/semantics as the reference
(``/root/reference/config.py:16-72``): defaults are obtained by parsing ``[]``, keyword
arguments override, ``from_args`` re-parses a command line, ``__str__`` prints the
custom-vs-default split as JSON.  New knobs of this implementation are plain attributes
with defaults so that reference call sites are unaffected.
"""
import argparse
import json
import sys
from copy import deepcopy


class Config:
    def __init__(self, **kwargs):
        self.from_args([])
        self.default_args = deepcopy(self.__dict__)
        self.from_dict(kwargs)

    def __str__(self):
        custom, default = {}, {}
        for k, v in self.__dict__.items():
            if k == "default_args":
                continue
            if k in self.default_args and self.default_args.get(k) == v:
                default[k] = v
            else:
                custom[k] = v
        return json.dumps({"custom": custom, "default": default}, indent=4)

    __repr__ = __str__

    def from_dict(self, dictionary):
        for k, v in dictionary.items():
            setattr(self, k, v)
        return self

    def from_args(self, args=None):
        if args is None:
            args = sys.argv[1:]
        p = argparse.ArgumentParser(description="GAN component analysis config")
        p.add_argument("--model", dest="model", type=str, default="StyleGAN", help="The network to analyze")
        p.add_argument("--layer", dest="layer", type=str, default="g_mapping", help="The layer to analyze")
        p.add_argument("--class", dest="output_class", type=str, default=None, help="Output class to generate")
        p.add_argument("--est", dest="estimator", type=str, default="ipca",
                       help="The algorithm to use [ipca, ipca-exact]")
        p.add_argument("--sparsity", type=float, default=1.0, help="Sparsity parameter of SPCA")
        p.add_argument("--video", dest="make_video", action="store_true", help="Generate output videos (MP4s)")
        p.add_argument("--batch", dest="batch_mode", action="store_true", help="Don't open windows")
        p.add_argument("-b", dest="batch_size", type=int, default=None, help="Minibatch size")
        p.add_argument("-c", dest="components", type=int, default=80, help="Number of components to keep")
        p.add_argument("-n", type=int, default=300_000, help="Number of examples to use in decomposition")
        p.add_argument("--use_w", action="store_true", help="Use W latent space (StyleGAN(2))")
        p.add_argument("--sigma", type=float, default=2.0, help="Number of stdevs to walk in visualize.py")
        p.add_argument("--inputs", type=str, default=None, help="Path to directory with named components")
        p.add_argument("--seed", type=int, default=None, help="Seed used in decomposition")
        return self.from_dict(p.parse_args(args).__dict__)
