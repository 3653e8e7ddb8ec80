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


def test_grouped_generator_calls_give_the_blocks_of_the_mini_batch_loop(dev, monkeypatch):
    """``_fit_blocks`` pushes several mini-batches of a narrow layer through one ``partial_forward`` (cfg4: 8 x 10 000
    rows, ``_forward_rows``) and hands the estimator slices of the hooked activation.  The blocks must be the ones the
    reference's mini-batch loop builds (decomposition.py:245-261: block gi = rows gi .. gi + NB - 1 of the latents, the
    tail mini-batch kept in part): the last block's rows must be bit-identical between the grouped loop and the
    one-mini-batch-per-call loop (ragged B / NB included), and the faithful estimator - sequential in the blocks, every
    block its own solve - must end in the same state up to the order of its float64 atomic additions (split-K products:
    components to the last float32 bit or one beside it)."""
    from ganspace_amd import decomposition as dec
    from ganspace_amd.estimators import get_estimator
    from ganspace_amd.wrappers import get_instrumented_model
    inst = get_instrumented_model("StyleGAN2", "car", "style", dev)
    model = inst.model
    inst.retain_layer("style")
    k, n, B = 12, 30_000, 768                      # NB = 2000: mini-batches of 768 straddle the block boundaries
    plan = dec._Plan.make(n, B, k)
    assert plan.NB == 2000 and plan.NB % plan.B != 0
    torch.manual_seed(dec.SEED_SAMPLING)
    np.random.seed(dec.SEED_SAMPLING)
    latents, _ = dec._presample(model, plan, model.get_latent_shape(), dev)
    results = []
    for rows_per_call in (dec.FORWARD_ROWS, B):    # grouped (81 920-row cap: all 15 blocks in one call) / one mini-batch
        monkeypatch.setattr(dec, "FORWARD_ROWS", rows_per_call)
        est = get_estimator("ipca", k, 1.0)
        last = dec._fit_blocks(est, inst, latents, plan, "style", 512, False, dev)
        comp, stdev, ratio = est.get_components()
        results.append((np.array(comp), np.array(stdev), np.array(est.transformer.mean_), last.clone()))
        assert int(est.transformer.n_samples_seen_) == len(list(plan.block_starts)) * plan.NB
    (c0, s0, m0, l0), (c1, s1, m1, l1) = results
    assert torch.equal(l0, l1)                     # the rows the random-direction statistic reads (last block): bit for bit
    np.testing.assert_allclose(c0, c1, rtol=0, atol=2e-8)
    assert O.signed_cosines(c0, c1).min() > 1 - 1e-12
    np.testing.assert_allclose(s0, s1, rtol=1e-9)
    np.testing.assert_allclose(m0, m1, rtol=0, atol=1e-12 * max(1.0, float(np.abs(m1).max())))
    # ... and they are the mapping network's rows of latents[gi : gi + NB]
    with torch.no_grad():
        ref = model.model.style(latents[plan.block_starts[-1]:plan.block_starts[-1] + plan.NB])
    assert torch.equal(ref, l0)
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
        if name == "StyleGAN2":
            with pytest.raises(RuntimeError, match="not encountered"):      # wrappers.py:259
                model.partial_forward(z, "no_such_layer")
        else:
            # the reference's BigGAN.partial_forward runs the whole generator for a name it does not know and
            # returns None without raising (wrappers.py:618-648)
            with torch.no_grad():
                assert model.partial_forward(z, "no_such_layer") is None
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


@pytest.mark.parametrize("precision", ["f32", "bf16x6"])
def test_cfg5_conv_features_d131072(dev, precision):
    """BASELINE config 5 at its own feature dimension: StyleGAN2 ``convs.2`` activations are 512 x 16 x 16 =
    131 072 features (r = k + rows + 1 rows of the stacked matrix, 1 GB per 2000-row block at full NB; small blocks
    here so that the float64 SVD oracle stays affordable).  Same hooked activations to the device estimator and to the
    CPU oracle of sklearn's stacked-SVD recurrence (_incremental_pca.py:347-378)."""
    from ganspace_amd.estimators import get_estimator
    from ganspace_amd.wrappers import get_instrumented_model
    from ganspace_amd.estimators import IPCAEstimator
    inst = get_instrumented_model("StyleGAN2", "cat", "convs.2", dev)
    model = inst.model
    # 'ipca' picks the small-side recurrence by itself for d > 8192; precision = contraction of T = M M^T
    est = get_estimator("ipca", 8, 1.0) if precision == "f32" else IPCAEstimator(8, "faithful", precision=precision)
    orc = O.IPCAEstimatorOracle(8, "svd")
    np.random.seed(6)
    with torch.no_grad():
        for _ in range(3):
            rows = []
            for _ in range(3):
                z = model.sample_latent(64)
                model.partial_forward(z, "convs.2")
                rows.append(inst.retained_features()["convs.2"].reshape(64, -1))
            X = torch.cat(rows)                      # [192, 131072] float32, on the device
            assert X.shape[1] == 131072
            assert est.fit_partial(X)
            orc.fit_partial(X.cpu().numpy())
    cos = O.signed_cosines(est.transformer.components_, orc.transformer.components_)
    assert cos.min() > 1 - 1e-5, cos
    np.testing.assert_allclose(est.transformer.singular_values_, orc.transformer.singular_values_, rtol=1e-4)
    np.testing.assert_allclose(est.transformer.explained_variance_ratio_,
                               orc.transformer.explained_variance_ratio_, rtol=1e-3)
    np.testing.assert_allclose(est.transformer.mean_, orc.transformer.mean_, atol=1e-5 * np.abs(orc.transformer.mean_).max())
    inst.close()


def test_cfg2_full_n_through_get_or_compute_matches_sklearn(dev, tmp_path):
    """BASELINE config 2 at its own size: StyleGAN2 ffhq --layer=style --use_w -n=1_000_000 -b=10_000 -c=80 through
    ``get_or_compute`` (reference z stream -> mapping kernel -> 100 IPCA blocks -> .npz), against scikit-learn's
    IncrementalPCA - configured as estimators.py:59 and driven as decomposition.py:263-264 - on the SAME 1e6 rows.
    ``ipca``: all 80 components, signs included.  ``ipca-exact``: the leading 20 (north_star tolerance: cos >= 0.999)."""
    from types import SimpleNamespace
    from oracle import reference_cpu
    from ganspace_amd import decomposition as dec
    from ganspace_amd.config import Config
    from ganspace_amd.wrappers import get_instrumented_model
    n, B, k = 1_000_000, 10_000, 80
    inst = get_instrumented_model("StyleGAN2", "ffhq", "style", dev, use_w=True)
    sub = SimpleNamespace(run_dir_root=str(tmp_path), run_dir=str(tmp_path))
    out = {}
    for est_name in ("ipca", "ipca-exact"):
        cfg = Config(model="StyleGAN2", layer="style", output_class="ffhq", use_w=True, n=n, batch_size=B,
                     components=k, estimator=est_name)
        path = dec.get_or_compute(cfg, inst, submit_config=sub)
        out[est_name] = np.load(path, allow_pickle=False)
    assert (tmp_path / "cache" / "components" / "stylegan2-ffhq_style_ipca_c80_n1000000_w.npz").is_file()

    # the same rows for the CPU reference: the stream compute() consumed (np.random.seed(1), one randint per batch)
    model = inst.model
    model.use_w()
    plan = dec._Plan.make(n, B, k)
    np.random.seed(dec.SEED_SAMPLING)
    latents, _ = dec._presample(model, plan, (1, 512), dev)
    ref = reference_cpu.make_reference_ipca(k)
    for gi in plan.block_starts:
        ref.partial_fit(latents[gi:gi + plan.NB].cpu().numpy())
        ref.n_samples_seen_ = np.int64(ref.n_samples_seen_)
    del latents

    got = out["ipca"]
    cos = O.signed_cosines(got["act_comp"].reshape(k, -1), ref.components_)
    assert cos.min() > 1 - 5e-6, cos                                     # all 80, signed
    np.testing.assert_allclose(got["act_stdev"], np.sqrt(ref.explained_variance_), rtol=1e-4)
    np.testing.assert_allclose(got["var_ratio"], ref.explained_variance_ratio_, rtol=1e-4)
    np.testing.assert_allclose(got["act_mean"].ravel(), ref.mean_, atol=1e-5)
    ex = out["ipca-exact"]
    cos_ex = O.signed_cosines(ex["act_comp"].reshape(k, -1), ref.components_)
    assert cos_ex[:20].min() > 0.999, cos_ex[:20]
    np.testing.assert_allclose(ex["act_stdev"][:20], np.sqrt(ref.explained_variance_)[:20], rtol=1e-3)
    inst.close()


def test_integration_stub_runs_against_the_c_abi(dev, golden_dir):
    """the ctypes class a maintainer of the reference would paste into ``estimators.py``
    (``examples/reference_binding.py``, binds ``include/ganspace_hip.h`` directly, no ``ganspace_amd`` import) driven
    the way ``decomposition.compute()`` drives an estimator, against the reference-generated fixture."""
    import importlib.util
    import inputs as gin
    from ganspace_amd import _build
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    os.environ["GANSPACE_HIP_LIB"] = _build.lib_path()
    spec = importlib.util.spec_from_file_location("reference_binding", os.path.join(root, "examples", "reference_binding.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    case = gin.IPCA_CASES["d512_k20"]
    est = mod.HipIPCAEstimator(case["k"])
    assert est.batch_support and est.get_param_str() == "ipca_c20"
    X = None
    for X in gin.ipca_blocks(case):
        assert est.fit_partial(X)                                      # host float32 block, like decomposition.py:264
    mean = est.transformer.mean_.reshape(1, -1)                        # decomposition.py:289
    comp, stdev, ratio = est.get_components()
    g = np.load(os.path.join(golden_dir, "ipca_ref_d512_k20.npz"), allow_pickle=False)
    cos = O.signed_cosines(comp, g["components"])
    assert cos[:case["ncheck"]].min() > 1 - 5e-6, cos
    np.testing.assert_allclose(stdev[:case["ncheck"]], g["stdev"][:case["ncheck"]], rtol=1e-4)
    np.testing.assert_allclose(mean.ravel(), g["mean"], atol=1e-5)


def test_whole_matrix_estimators_on_the_device(dev):
    """``get_estimator('pca' | 'fbpca')`` (reference estimators.py:84-160) as device estimators.  'pca' against
    sklearn ``PCA(svd_solver='full')`` post-processed as the reference does; 'fbpca' (randomized PCA of the uncentred
    matrix, raw=True, n_iter=2, l=2k) against the restatement of fbpca's algorithm on the same test matrix."""
    from sklearn.decomposition import PCA
    from ganspace_amd.estimators import get_estimator
    rs = np.random.RandomState(4)
    A = rs.standard_normal((24, 96)) * (1.3 ** -np.arange(24))[:, None] * 3.0
    X = (rs.standard_normal((5000, 24)) @ A + 0.05 * rs.standard_normal((5000, 96)) + 0.7).astype(np.float32)
    Xd = torch.from_numpy(X).to(dev)
    k = 12
    # --- pca
    est = get_estimator("pca", k, 1.0)
    est.fit(Xd)
    comp, stdev, ratio = est.get_components()
    ref = PCA(k, svd_solver="full").fit(X.astype(np.float64))
    ref_stdev = np.dot(ref.components_, X.astype(np.float64).T).std(axis=1)
    cos = O.signed_cosines(comp, ref.components_)
    assert cos.min() > 1 - 1e-6, cos
    np.testing.assert_allclose(stdev, ref_stdev, rtol=1e-5)
    np.testing.assert_allclose(ratio, ref_stdev ** 2 / X.astype(np.float64).var(axis=0).sum(), rtol=1e-5)
    np.testing.assert_allclose(np.asarray(est.transformer.mean_).ravel(), X.astype(np.float64).mean(0), atol=1e-5)
    # --- fbpca: the randomized range finder itself (n_iter = 2, l = 2k) against its NumPy restatement fed the same
    #     test matrix (same state of NumPy's global stream), and - leading components - against the exact SVD
    from oracle import fbpca_port
    est = get_estimator("fbpca", k, 1.0)
    np.random.seed(21)
    est.fit(Xd)
    comp, stdev, ratio = est.get_components()
    orc = fbpca_port.FacebookPCAEstimatorOracle(k)
    np.random.seed(21)
    orc.fit(X.astype(np.float64))
    ocomp, ostdev, oratio = orc.get_components()
    acos = np.abs(np.sum(comp.astype(np.float64) * ocomp, axis=1))
    assert acos.min() > 1 - 1e-5, acos
    np.testing.assert_allclose(stdev, ostdev, rtol=1e-4)
    np.testing.assert_allclose(ratio, oratio, rtol=2e-4)
    _, _, Vt = np.linalg.svd(X.astype(np.float64), full_matrices=False)
    assert np.abs(np.sum(comp[:4].astype(np.float64) * Vt[:4], axis=1)).min() > 0.999


def test_pca_estimator_through_get_or_compute(dev, tmp_path):
    """A non-batch estimator through the driver (decomposition.py:222-224, 266, 276-287): all blocks are collected
    into the ``[N + NB, d]`` matrix (zero rows included, like the reference), centred and fitted once."""
    from types import SimpleNamespace
    from sklearn.decomposition import PCA
    from ganspace_amd import decomposition as dec
    from ganspace_amd.config import Config
    from ganspace_amd.wrappers import get_instrumented_model
    n, B, k = 6000, 500, 10
    cfg = Config(model="StyleGAN2", layer="style", output_class="cat", use_w=True, n=n, batch_size=B, components=k,
                 estimator="pca")
    inst = get_instrumented_model(cfg.model, cfg.output_class, cfg.layer, dev, use_w=True)
    sub = SimpleNamespace(run_dir_root=str(tmp_path), run_dir=str(tmp_path))
    path = dec.get_or_compute(cfg, inst, submit_config=sub)
    assert path.name == "stylegan2-cat_style_pca-full_c10_n6000_w.npz"
    data = np.load(path, allow_pickle=False)
    # CPU restatement on the same rows
    model = inst.model
    model.use_w()
    plan = dec._Plan.make(n, B, k)
    np.random.seed(dec.SEED_SAMPLING)
    latents, _ = dec._presample(model, plan, (1, 512), dev)
    lat = latents.cpu().numpy()
    samples = np.zeros((plan.N + plan.NB, 512), np.float32)
    for gi in plan.block_starts:
        samples[gi:gi + plan.NB] = lat[gi:gi + plan.NB]
    mean = samples.mean(axis=0, keepdims=True, dtype=np.float32)
    samples -= mean
    ref = PCA(k, svd_solver="full").fit(samples)
    ref_stdev = np.dot(ref.components_, samples.T).std(axis=1)
    cos = O.signed_cosines(data["act_comp"].reshape(k, -1), ref.components_)
    assert np.abs(cos).min() > 1 - 1e-5, cos
    np.testing.assert_allclose(data["act_stdev"], ref_stdev, rtol=1e-3)
    np.testing.assert_allclose(data["act_mean"].ravel(), mean.ravel(), atol=1e-5)
    inst.close()


def test_harvesting_the_next_group_on_a_second_stream_changes_nothing(dev, monkeypatch):
    """Faithful Gram-side fits of a generator layer issue the NEXT group's ``partial_forward`` on a second stream before
    fitting the current one (and switch the long Linear launches to the per-tile kernel meanwhile,
    ``gs_linear_set_resident``).  Same rows to the same blocks in the same order: the last block's rows are bit-identical,
    the fitted state equal to the in-order loop's (``GANSPACE_HARVEST_AHEAD=0``) to float64 reduction order - for groups
    that alias the hooked activation and for groups assembled from several forward calls; and the switch is back on after."""
    from ganspace_amd import _lib
    from ganspace_amd import decomposition as dec
    from ganspace_amd.estimators import get_estimator
    from ganspace_amd.wrappers import get_instrumented_model
    inst = get_instrumented_model("StyleGAN2", "car", "style", dev, use_w=False)
    model = inst.model
    inst.retain_layer("style")
    lib = _lib.load()
    for n, B, rows_per_call in ((24_000, 2000, 4000), (18_000, 768, 768)):      # aliased groups of 2 blocks / 3 calls per block
        k = 12
        plan = dec._Plan.make(n, B, k)
        torch.manual_seed(dec.SEED_SAMPLING)
        np.random.seed(dec.SEED_SAMPLING)
        latents, _ = dec._presample(model, plan, model.get_latent_shape(), dev)
        monkeypatch.setattr(dec, "FORWARD_ROWS", rows_per_call)
        results = []
        for ahead in ("1", "0"):
            monkeypatch.setenv("GANSPACE_HARVEST_AHEAD", ahead)
            est = get_estimator("ipca", k, 1.0)
            last = dec._fit_blocks(est, inst, latents, plan, "style", 512, False, dev)
            comp, stdev, _ = est.get_components()
            results.append((np.array(comp), np.array(stdev), np.array(est.transformer.mean_), last.clone()))
            assert lib.gs_linear_set_resident(1) == 1             # restored by _fit_blocks
        (c0, s0, m0, l0), (c1, s1, m1, l1) = results
        assert torch.equal(l0, l1)
        np.testing.assert_allclose(c0, c1, rtol=0, atol=2e-8)
        np.testing.assert_allclose(s0, s1, rtol=1e-9)
        np.testing.assert_allclose(m0, m1, rtol=0, atol=1e-12 * max(1.0, float(np.abs(m1).max())))
