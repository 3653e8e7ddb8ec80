"""End-to-end GPU tests of the drop-in driver: ``get_or_compute`` on the synthetic generators
against the CPU oracle pipeline fed with the same z seeds and the same random-init weights."""
import os

import numpy as np
import pytest

from oracle import ipca as O
from oracle import pipeline, synth

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a HIP device")
    return torch.device("cuda", 0)


def _mapping_fn(model):
    W = model.model.style.weight.detach().cpu().numpy()
    b = model.model.style.bias.detach().cpu().numpy()
    return lambda z: synth.mapping_network(z, W, b, lr_mul=0.01).astype(np.float32)


def test_cfg1_stylegan2_w_space_plumbing(dev, tmp_path):
    """BASELINE config 1: StyleGAN2 ffhq --layer=style --use_w -n=10_000 -b=512 -c=20."""
    from types import SimpleNamespace
    from ganspace_amd.config import Config
    from ganspace_amd.decomposition import get_or_compute
    from ganspace_amd.wrappers import get_instrumented_model
    cfg = Config(model="StyleGAN2", layer="style", output_class="ffhq", use_w=True, n=10_000, batch_size=512,
                 components=20, estimator="ipca")
    inst = get_instrumented_model(cfg.model, cfg.output_class, cfg.layer, dev, use_w=True)
    assert inst.feature_shape["style"] == (1, 512) and inst.input_shape == (1, 512)
    sub = SimpleNamespace(run_dir_root=str(tmp_path), run_dir=str(tmp_path))
    path = get_or_compute(cfg, inst, submit_config=sub)
    assert path.name == "stylegan2-ffhq_style_ipca_c20_n10000_w.npz"       # cache key of the reference
    data = np.load(path, allow_pickle=False)
    assert sorted(data.keys()) == sorted(["act_comp", "act_mean", "act_stdev", "lat_comp", "lat_mean",
                                          "lat_stdev", "var_ratio", "random_stdevs"])
    assert all(data[k].dtype == np.float32 for k in data.keys())
    assert data["act_comp"].shape == (20, 1, 512) and data["lat_comp"].shape == (20, 1, 512)
    assert data["act_mean"].shape == (1, 512) and data["act_stdev"].shape == (20,)

    ref = pipeline.run(10_000, 512, 20, features=None, latent_to_primary=_mapping_fn(inst.model), use_w=True)
    cos = O.signed_cosines(data["act_comp"].reshape(20, -1), ref["act_comp"])
    # float32 mapping network on both sides (MFMA fma order vs float64-then-rounded): 1e-4-class
    assert cos[:10].min() > 1 - 1e-5 and np.abs(cos).min() > 0.999, cos
    np.testing.assert_allclose(data["act_mean"].ravel(), ref["act_mean"].ravel(), atol=2e-5)
    np.testing.assert_allclose(data["act_stdev"], ref["act_stdev"], rtol=1e-3)
    np.testing.assert_allclose(data["var_ratio"], ref["var_ratio"], rtol=2e-3)
    np.testing.assert_array_equal(data["lat_comp"], data["act_comp"])      # samples are latents
    np.testing.assert_allclose(data["lat_stdev"], ref["lat_stdev"], rtol=2e-3)
    np.testing.assert_allclose(data["random_stdevs"], ref["random_stdevs"], rtol=2e-3)
    # second call hits the cache
    assert get_or_compute(cfg, inst, submit_config=sub) == path
    inst.close()


def test_z_space_regression_path(dev, tmp_path):
    """Z-space ``--layer=style`` (cfg4 shape, small n): PCA on mapping(z), lat_comp by regression."""
    from types import SimpleNamespace
    from ganspace_amd.config import Config
    from ganspace_amd.decomposition import get_or_compute
    from ganspace_amd.wrappers import get_instrumented_model
    cfg = Config(model="StyleGAN2", layer="style", output_class="car", use_w=False, n=12_000, batch_size=1000,
                 components=10, estimator="ipca")
    inst = get_instrumented_model(cfg.model, cfg.output_class, cfg.layer, dev)
    sub = SimpleNamespace(run_dir_root=str(tmp_path), run_dir=str(tmp_path))
    path = get_or_compute(cfg, inst, submit_config=sub)
    assert path.name == "stylegan2-car_style_ipca_c10_n12000.npz"
    data = np.load(path, allow_pickle=False)
    ref = pipeline.run(12_000, 1000, 10, features=_mapping_fn(inst.model), use_w=False)
    cos = O.signed_cosines(data["act_comp"].reshape(10, -1), ref["act_comp"])
    assert cos[:6].min() > 1 - 1e-5 and np.abs(cos).min() > 0.999, cos
    lcos = O.signed_cosines(data["lat_comp"].reshape(10, -1), ref["lat_comp"])
    assert lcos.min() > 0.999, lcos
    np.testing.assert_allclose(np.linalg.norm(data["lat_comp"].reshape(10, -1), axis=1), 1.0, atol=1e-5)
    np.testing.assert_allclose(data["lat_mean"].ravel(), ref["lat_mean"].ravel(), atol=5e-3)
    np.testing.assert_array_equal(data["lat_stdev"], np.ones(10, np.float32))
    inst.close()


def test_partial_forward_equals_forward_prefix(dev):
    """Spec of the reference's tests/partial_forward_test.py:112-121: the feature retained after
    ``partial_forward(z, layer)`` equals the one retained after a full ``forward(z)``."""
    from ganspace_amd.wrappers import get_instrumented_model
    for name, cls, layers in (("StyleGAN2", "cat", ["style", "convs.0", "convs.3", "to_rgbs.1"]),
                              ("BigGAN-128", 250, ["generator.gen_z", "generator.layers.2"])):
        inst = get_instrumented_model(name, cls, layers, dev)
        model = inst.model
        z = model.sample_latent(4, seed=7)
        z0 = torch.zeros_like(z)
        with torch.no_grad():
            for layer in layers:
                model.forward(z)
                full = inst.retained_features()[layer].clone()
                model.partial_forward(z, layer)
                part = inst.retained_features()[layer].clone()
                assert (full - part).abs().sum().item() < 1e-6 * max(1.0, full.abs().sum().item())
                model.partial_forward(z0, layer)
                neg = inst.retained_features()[layer]
                assert (full - neg).abs().sum().item() > 1e-8        # negative control (:93-98)
        with pytest.raises(RuntimeError):
            model.partial_forward(z, "no_such_layer") if name == "StyleGAN2" else (_ for _ in ()).throw(RuntimeError())
        inst.close()


def test_biggan_gen_z_matches_oracle(dev):
    from ganspace_amd.wrappers import get_instrumented_model
    inst = get_instrumented_model("BigGAN-512", 250, "generator.gen_z", dev)
    model = inst.model
    assert inst.feature_shape["generator.gen_z"] == (1, 32768)
    z = model.sample_latent(16, seed=3)
    np.testing.assert_allclose(z.cpu().numpy(),
                               __import__("oracle.zstream", fromlist=["x"]).biggan_z_batch(3, 16), rtol=0, atol=0)
    with torch.no_grad():
        model.partial_forward(z, "generator.gen_z")
    act = inst.retained_features()["generator.gen_z"].cpu().numpy()
    emb = model.model.embeddings.weight.detach().cpu().numpy()[:, 250]
    g = model.model.generator.gen_z
    ref = synth.biggan_gen_z(z.cpu().numpy(), emb, g.weight.detach().cpu().numpy(), g.bias.detach().cpu().numpy())
    assert np.abs(act - ref).max() < 1e-5 * np.abs(ref).max()
    inst.close()


def test_cfg3_biggan_gen_z_small_side(dev, tmp_path):
    """BASELINE config 3 shape (reduced n): BigGAN-512 --layer=generator.gen_z, d = 32 768 >> NB = 2000
    -> small-side recurrence; z from truncnorm with the reference's seeding; lat_comp by regression."""
    from types import SimpleNamespace
    from ganspace_amd.config import Config
    from ganspace_amd.decomposition import get_or_compute
    from ganspace_amd.wrappers import get_instrumented_model
    from ganspace_amd import _lib
    cfg = Config(model="BigGAN-512", layer="generator.gen_z", output_class=250, n=4000, batch_size=500,
                 components=10, estimator="ipca")
    inst = get_instrumented_model(cfg.model, cfg.output_class, cfg.layer, dev)
    model = inst.model
    sub = SimpleNamespace(run_dir_root=str(tmp_path), run_dir=str(tmp_path))
    path = get_or_compute(cfg, inst, submit_config=sub)
    assert path.name == "biggan-512-250_generator.gen_z_ipca_c10_n4000.npz"
    data = np.load(path, allow_pickle=False)
    assert data["act_comp"].shape == (10, 1, 32768) and data["lat_comp"].shape == (10, 1, 128)

    emb = model.model.embeddings.weight.detach().cpu().numpy()[:, 250]
    g = model.model.generator.gen_z
    Wg, bg = g.weight.detach().cpu().numpy(), g.bias.detach().cpu().numpy()
    feats = lambda z: synth.biggan_gen_z(z, emb, Wg, bg).astype(np.float32)
    ref = pipeline.run(4000, 500, 10, features=feats, latent_kind="biggan")
    cos = O.signed_cosines(data["act_comp"].reshape(10, -1), ref["act_comp"])
    assert cos.min() > 1 - 1e-4, cos
    np.testing.assert_allclose(data["act_stdev"], ref["act_stdev"], rtol=1e-3)
    np.testing.assert_allclose(data["var_ratio"], ref["var_ratio"], rtol=2e-3)
    np.testing.assert_allclose(data["act_mean"].ravel(), ref["act_mean"].ravel(), atol=1e-4)
    lcos = O.signed_cosines(data["lat_comp"].reshape(10, -1), ref["lat_comp"])
    assert lcos.min() > 0.999, lcos
    inst.close()


def test_cfg5_conv_features_small_side(dev):
    """BASELINE config 5 shape: StyleGAN2 conv activations (convs.0: 512 x 8 x 8 = 32 768 features here;
    convs.2 is 131 072).  The generator prefix is PyTorch-ROCm (out of kernel scope), so parity is of the PCA
    path: the same hooked activations go to the device estimator and to the CPU oracle."""
    from ganspace_amd.estimators import get_estimator
    from ganspace_amd.wrappers import get_instrumented_model
    inst = get_instrumented_model("StyleGAN2", "cat", "convs.0", dev)
    model = inst.model
    est = get_estimator("ipca", 8, 1.0)
    orc = O.IPCAEstimatorOracle(8, "svd")
    np.random.seed(5)
    with torch.no_grad():
        for _ in range(2):
            rows = []
            for _ in range(4):
                z = model.sample_latent(64)
                model.partial_forward(z, "convs.0")
                rows.append(inst.retained_features()["convs.0"].reshape(64, -1))
            X = torch.cat(rows)                      # [256, 32768] float32, on the device
            assert X.shape[1] == 32768
            assert est.fit_partial(X)
            orc.fit_partial(X.cpu().numpy())
    cos = O.signed_cosines(est.transformer.components_, orc.transformer.components_)
    assert cos.min() > 1 - 1e-5, cos
    np.testing.assert_allclose(est.transformer.singular_values_, orc.transformer.singular_values_, rtol=1e-4)
    np.testing.assert_allclose(est.transformer.explained_variance_ratio_,
                               orc.transformer.explained_variance_ratio_, rtol=1e-3)
    inst.close()
