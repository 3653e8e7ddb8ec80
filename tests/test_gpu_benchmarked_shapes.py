"""Parity of the small-side recurrence AT THE SHAPES bench.py TIMES (BASELINE cfg3 / cfg5: NB = 2000, k = 80,
d = 32 768 / 131 072; r = 2081, deferred diagonalisation from the fifth block on, K-sliced T = M M^T, split-bf16
contraction) - not at a miniature of them.

Checker: ``oracle/smallside_torch.py`` (float64 restatement of sklearn's recurrence on the r x r side, pinned to the
SVD-form oracle in tests/test_oracle.py), and for d = 32 768 scikit-learn's ``IncrementalPCA`` itself, configured as
``/root/reference/estimators.py:59`` does, on the first six blocks."""
import numpy as np
import pytest

from oracle import ipca as O
from oracle.smallside_torch import SmallSideTorchOracle, lowrank_plus_noise_blocks

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

NB, K = 2000, 80


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a HIP device")
    return torch.device("cuda", 0)


@pytest.mark.parametrize("d,n_blocks", [(32768, 8), (131072, 6)])
def test_smallside_at_benchmarked_shape_matches_float64_oracle(dev, d, n_blocks):
    """f32 and bf16x6 contractions fed the same blocks as the float64 oracle; all 80 components, singular values,
    mean / variance and the explained-variance ratio.  Tolerances: signed cosine 1 - 3e-6 (f32) / 1 - 8e-6 (bf16x6),
    singular values 2e-4 relative - those of the miniature small-side tests (tests/test_gpu_parity.py)."""
    from ganspace_amd import _lib
    from ganspace_amd.estimators import IPCAEstimator
    lib = _lib.load()
    ests = {"f32": IPCAEstimator(K, "faithful", precision="f32"), "bf16x6": IPCAEstimator(K, "faithful", precision="bf16x6")}
    orc = SmallSideTorchOracle(K)
    sk = None
    if d == 32768:
        from oracle import reference_cpu
        sk = reference_cpu.make_reference_ipca(K)
    carried = {p: 0 for p in ests}
    for i, X in enumerate(lowrank_plus_noise_blocks(d, n_blocks, rows=NB, device=dev)):
        for p, e in ests.items():
            assert e.fit_partial(X) is True
            h = e.transformer._h
            carried[p] += int(i >= 4 and lib.gs_ipca_last_sweeps(h) == 0 and lib.gs_ipca_last_mults(h) > 0)
        orc.partial_fit(X)
        if sk is not None and i < 6:
            sk.partial_fit(X.cpu().numpy())
            if i == 5:
                # scikit-learn itself (the code the reference runs) after six blocks; reading in mid-stream folds the
                # pending rotation back into the state and the fit continues
                mid = ests["f32"].get_components()[0]
                cs = O.signed_cosines(mid, sk.components_)
                assert cs.min() > 1 - 3e-6, cs.min()
                np.testing.assert_allclose(ests["f32"].transformer.singular_values_, sk.singular_values_, rtol=2e-4)
        del X
    assert ests["f32"].transformer._mode == _lib.GS_MODE_SMALLSIDE
    for p, e in ests.items():
        assert carried[p] >= n_blocks - 5, (p, carried)      # the deferred path is what was exercised (and is timed)
        comp = e.get_components()[0]
        cos = O.signed_cosines(comp, orc.components_)
        assert cos.min() > 1 - (3e-6 if p == "f32" else 8e-6), (p, cos.min())
        t = e.transformer
        np.testing.assert_allclose(t.singular_values_, orc.singular_values_, rtol=2e-4)
        np.testing.assert_allclose(t.explained_variance_ratio_, orc.explained_variance_ratio_, rtol=4e-4)
        np.testing.assert_allclose(t.mean_, orc.mean_, atol=2e-6)
        np.testing.assert_allclose(t.var_, orc.var_, rtol=1e-4)
        assert int(t.n_samples_seen_) == n_blocks * NB


def test_smallside_fit_survives_a_last_batch_taller_than_the_others(dev):
    """sklearn's ``fit`` merges a tail shorter than k into the last batch (gen_batches(min_batch_size=k)), so the last
    block can be up to batch_size + k - 1 rows: taller than every block before it.  The small side must re-size its
    buffers without losing the deferred state (W = Q^T M) it carries from the fifth block on."""
    from ganspace_amd.estimators import IPCAEstimator
    d, k = 8448, 24
    rng = np.random.default_rng(19)
    A = rng.standard_normal((60, d)) * (1.12 ** -np.arange(60))[:, None]
    n = 7 * 100 + 10                        # batch_size = max(100, 2k) = 100: six blocks of 100, the last of 110
    X = (rng.standard_normal((n, 60)) @ A + 0.03 * rng.standard_normal((n, d)) + 0.2).astype(np.float32)
    est = IPCAEstimator(k, "faithful")
    est.fit(torch.from_numpy(X).to(dev))
    orc = O.SklearnRecurrenceOracle(k)
    for lo in range(0, 600, 100):
        orc.partial_fit(X[lo:lo + 100])
    orc.partial_fit(X[600:])
    assert int(est.transformer.n_samples_seen_) == n
    cos = O.signed_cosines(est.get_components()[0], orc.components_)
    assert cos.min() > 1 - 3e-6, cos.min()
    np.testing.assert_allclose(est.transformer.singular_values_, orc.singular_values_, rtol=2e-4)
    np.testing.assert_allclose(est.transformer.mean_, orc.mean_, atol=2e-6)
    # growing blocks fed by hand, with a read in between
    est2, orc2 = IPCAEstimator(k, "faithful"), O.SklearnRecurrenceOracle(k)
    sizes = [60, 60, 60, 60, 60, 60, 90, 90, 130]
    lo = 0
    for i, m in enumerate(sizes):
        assert est2.fit_partial(torch.from_numpy(X[lo:lo + m]).to(dev))
        orc2.partial_fit(X[lo:lo + m])
        lo += m
        if i == 6:
            assert O.signed_cosines(est2.get_components()[0], orc2.components_).min() > 1 - 3e-6
    assert O.signed_cosines(est2.get_components()[0], orc2.components_).min() > 1 - 3e-6


def test_smallside_block_taller_than_4096_rows(dev):
    """``-b 5000`` on a wide layer: r = k + rows + 1 = 5021 (the former cap was 4096)."""
    from ganspace_amd.estimators import IPCAEstimator
    d, k, rows = 9216, 20, 5000
    orc = SmallSideTorchOracle(k)
    est = IPCAEstimator(k, "faithful")
    for X in lowrank_plus_noise_blocks(d, 3, rows=rows, latent=48, decay=1.1, seed=5, device=dev):
        assert est.fit_partial(X) is True
        orc.partial_fit(X)
    cos = O.signed_cosines(est.get_components()[0], orc.components_)
    assert cos.min() > 1 - 3e-6, cos.min()
    np.testing.assert_allclose(est.transformer.singular_values_, orc.singular_values_, rtol=2e-4)


@pytest.mark.parametrize("k,rows", [(27, 100), (127, 128)])
def test_smallside_stack_height_is_a_multiple_of_the_panel(dev, k, rows):
    """r = k + rows + 1 a multiple of 128: the stacked matrix fills its last 128-row panel exactly, the coefficient
    buffer has no spare row behind row r - 1 (round-4 advisor: ``tn_gemm_kernel`` read one row past it)."""
    from ganspace_amd.estimators import IPCAEstimator
    assert (k + rows + 1) % 128 == 0
    d = 4128                                  # not a multiple of 128: a partial last column tile as well
    orc = SmallSideTorchOracle(k)
    est = IPCAEstimator(k, "smallside")
    for X in lowrank_plus_noise_blocks(d, 7, rows=rows, latent=max(48, k + 16), decay=1.06, seed=9, device=dev):
        assert est.fit_partial(X) is True
        orc.partial_fit(X)
    cos = O.signed_cosines(est.get_components()[0], orc.components_)
    # (the trailing components of the k = 127 case sit in the noise floor - neighbouring singular values 1e-3 apart: single
    #  directions are identifiable there to ~1e-2 only; the leading ones and every singular value are checked)
    lead = min(k, 40)
    assert cos[:lead].min() > 1 - 3e-6, cos[:lead].min()
    assert np.abs(cos).min() > 0.98, cos
    np.testing.assert_allclose(est.transformer.singular_values_, orc.singular_values_, rtol=2e-4)
    np.testing.assert_allclose(est.transformer.mean_, orc.mean_, atol=2e-6)
