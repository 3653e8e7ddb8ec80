#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the REAL reference.

Run in the build container only (needs /root/reference, which does not exist on
the GPU box):   python tests/golden/make_golden.py

What it imports from the reference (nothing is copied into the repo):
* ``/root/reference/estimators.py`` -> ``get_estimator('ipca', k, 1.0)``
  (the absent third-party ``fbpca`` module is stubbed); the
  arithmetic it delegates to is scikit-learn's ``IncrementalPCA`` installed in
  this image (1.7.2).
* ``/root/reference/models/stylegan/model.py`` -> ``G_mapping`` (the only
  in-tree 8 x (512->512) mapping MLP) with explicitly seeded weights.

Inputs are *not* stored: each fixture records the generator parameters and
``tests/golden/inputs.py`` rebuilds the exact arrays, so the fixtures stay
small (a few hundred KiB).
"""
import importlib.util
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import inputs as gin  # noqa: E402

REF = "/root/reference"


def load_reference_estimators():
    sys.modules.setdefault("fbpca", types.ModuleType("fbpca"))
    spec = importlib.util.spec_from_file_location("ref_estimators", os.path.join(REF, "estimators.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def load_reference_stylegan_model():
    spec = importlib.util.spec_from_file_location("ref_stylegan_model", os.path.join(REF, "models/stylegan/model.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def run_ipca_case(est_mod, case):
    est = est_mod.get_estimator("ipca", case["k"], 1.0)
    assert est.batch_support and est.get_param_str() == f"ipca_c{case['k']}"
    per_block_sv = []
    for X in gin.ipca_blocks(case):
        # the reference loop reuses one buffer for every block (decomposition.py:243,261)
        ok = est.fit_partial(X)
        assert ok
        per_block_sv.append(np.array(est.transformer.singular_values_, dtype=np.float64))
    comp, stdev, ratio = est.get_components()
    t = est.transformer
    return dict(
        components=np.asarray(comp, dtype=np.float64),
        stdev=np.asarray(stdev, dtype=np.float64),
        var_ratio=np.asarray(ratio, dtype=np.float64),
        singular_values=np.asarray(t.singular_values_, dtype=np.float64),
        mean=np.asarray(t.mean_, dtype=np.float64),
        var=np.asarray(t.var_, dtype=np.float64),
        explained_variance=np.asarray(t.explained_variance_, dtype=np.float64),
        n_samples_seen=np.int64(t.n_samples_seen_),
        per_block_singular_values=np.stack(per_block_sv),
        param_str=np.array(est.get_param_str()),
    )


def main():
    est_mod = load_reference_estimators()
    import sklearn
    for name, case in gin.IPCA_CASES.items():
        out = run_ipca_case(est_mod, case)
        out["sklearn_version"] = np.array(sklearn.__version__)
        path = os.path.join(HERE, f"ipca_ref_{name}.npz")
        np.savez_compressed(path, **out)
        print("wrote", path, {k: getattr(v, "shape", None) for k, v in out.items()})

    # error-path fixture: first block smaller than k -> fit_partial returns False
    est = est_mod.get_estimator("ipca", 20, 1.0)
    ok = est.fit_partial(np.zeros((10, 32), dtype=np.float32))
    assert ok is False

    # G_mapping golden (float32 torch CPU, as the reference would run it on CPU)
    import torch
    sg = load_reference_stylegan_model()
    torch.manual_seed(0)
    gm = sg.G_mapping().eval()
    case = gin.MAPPING_CASE
    W, b = gin.mapping_weights(case)
    with torch.no_grad():
        for i in range(8):
            lin = getattr(gm, f"dense{i}")
            # MyLinear stores randn * (1/lrmul): identical parameterisation to EqualLinear's weight
            lin.weight.copy_(torch.from_numpy(W[i]))
            lin.bias.copy_(torch.from_numpy(b[i]))
        z = torch.from_numpy(gin.mapping_z(case))
        w64 = gm.double()(z.double()).numpy()
        w32 = gm.float()(z).numpy()
    path = os.path.join(HERE, "mapping_gmapping_ref.npz")
    np.savez_compressed(path, w_f64=w64, w_f32=w32)
    print("wrote", path, w64.shape)


if __name__ == "__main__":
    main()
