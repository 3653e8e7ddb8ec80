"""CPU-side checks of the host shell that need no GPU: Config semantics, cache-file naming and the
validation errors of ``_compute`` (reference decomposition.py:370-394), nethook retain/edit protocol."""
from pathlib import Path
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from ganspace_amd.config import Config
from ganspace_amd.nethook import InstrumentedModel


def test_config_defaults_and_overrides_match_reference_flags():
    c = Config()
    assert (c.model, c.layer, c.estimator, c.components, c.n, c.use_w, c.batch_size, c.seed, c.sigma) == \
        ("StyleGAN", "g_mapping", "ipca", 80, 300_000, False, None, None, 2.0)
    c = Config(model="StyleGAN2", n=1_000_000, use_w=True)
    assert c.n == 1_000_000 and c.use_w and "custom" in str(c) and '"n": 1000000' in str(c)
    c2 = Config().from_args(["--model=BigGAN-512", "-c", "20", "-n=10_000", "-b", "512", "--use_w", "--class", "husky"])
    assert (c2.model, c2.components, c2.n, c2.batch_size, c2.use_w, c2.output_class) == \
        ("BigGAN-512", 20, 10000, 512, True, "husky")


def test_compute_validation_errors_and_cache_hit(tmp_path):
    from ganspace_amd import decomposition as D
    sub = SimpleNamespace(run_dir_root=str(tmp_path), run_dir=str(tmp_path))
    with pytest.raises(RuntimeError, match="Must specify number of samples"):
        D._compute(sub, Config(model="StyleGAN2", n=None))
    with pytest.raises(RuntimeError, match="InstrumentedModel"):
        D._compute(sub, Config(model="StyleGAN2", n=100), model=torch.nn.Linear(2, 2))
    with pytest.raises(RuntimeError, match="non-StyleGAN"):
        D._compute(sub, Config(model="BigGAN-512", n=100, use_w=True, output_class="husky"))
    # an existing cache file short-circuits compute(): same naming rule as the reference
    cfg = Config(model="StyleGAN2", layer="style", output_class="ffhq", use_w=True, n=10_000, components=20,
                 estimator="ipca", seed=7)
    p = Path(tmp_path) / "cache" / "components" / "stylegan2-ffhq_style_ipca_c20_n10000_w_seed7.npz"
    p.parent.mkdir(parents=True)
    np.savez_compressed(p, x=np.zeros(1))
    assert D._compute(sub, cfg) == p
    cfg.estimator = "ipca-exact"
    assert "ipca-exact_c20" in D.get_estimator(cfg.estimator, cfg.components).get_param_str()


def test_random_dirs_are_unit_and_seeded():
    from ganspace_amd.decomposition import get_random_dirs
    a, b = get_random_dirs(5, 64), get_random_dirs(5, 64)
    np.testing.assert_array_equal(a, b)
    np.testing.assert_allclose(np.linalg.norm(a, axis=1), 1.0, atol=1e-6)
    assert a.dtype == np.float32


class _Toy(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.a = torch.nn.Linear(4, 4, bias=False)
        self.b = torch.nn.Sequential(torch.nn.Linear(4, 3, bias=False))
        with torch.no_grad():
            self.a.weight.copy_(torch.eye(4) * 2)
            self.b[0].weight.fill_(1.0)

    def forward(self, x):
        return self.b(self.a(x))


def test_instrumented_model_retain_and_edit_protocol():
    inst = InstrumentedModel(_Toy())
    inst.retain_layers(["a", "b.0"])
    x = torch.ones(2, 4)
    y = inst(x)
    feats = inst.retained_features()
    assert list(feats.keys()) == ["a", "b.0"]
    torch.testing.assert_close(feats["a"], 2 * x)
    torch.testing.assert_close(y, torch.full((2, 3), 8.0))
    assert inst.retained_layer() is feats["a"] and not feats["a"].requires_grad
    # offset edit (what interactive.py / notebook_utils use to move along a component)
    inst.edit_layer("a", offset=torch.tensor([1.0, 0, 0, 0]))
    torch.testing.assert_close(inst(x), torch.full((2, 3), 9.0))
    torch.testing.assert_close(inst.retained_features()["a"], 2 * x)      # retained BEFORE the edit
    inst.edit_layer("a", ablation=1.0, replacement=torch.zeros(4))
    inst.remove_edits("a", remove_offset=True, remove_replacement=False)
    torch.testing.assert_close(inst(x), torch.zeros(2, 3))
    inst.remove_edits()
    torch.testing.assert_close(inst(x), torch.full((2, 3), 8.0))
    with pytest.raises(ValueError, match="not found"):
        inst.retain_layer("nope")
    inst.close()
    assert inst.retained_features() == {} and len(list(inst.model.a._forward_hooks)) == 0


def test_sample_latent_follows_the_reference_seed_stream(monkeypatch):
    """StyleGAN2.sample_latent in Z mode draws ONE randint from the global stream, then RandomState(seed)."""
    from ganspace_amd import wrappers
    from oracle import zstream
    m = wrappers.StyleGAN2.__new__(wrappers.StyleGAN2)
    torch.nn.Module.__init__(m)
    m.device, m.w_primary = torch.device("cpu"), False
    np.random.seed(1)
    z1 = m.sample_latent(4)
    z2 = m.sample_latent(3)
    seeds = zstream.batch_seeds(2)
    np.testing.assert_array_equal(z1.numpy(), zstream.stylegan_z_batch(seeds[0], 4))
    np.testing.assert_array_equal(z2.numpy(), zstream.stylegan_z_batch(seeds[1], 3))
    np.testing.assert_array_equal(m.sample_latent(2, seed=5).numpy(), zstream.stylegan_z_batch(5, 2))


@pytest.mark.parametrize("kind,dim", [("stylegan", 64), ("biggan", 16)])
def test_parallel_z_generation_is_bit_identical_and_ordered(monkeypatch, kind, dim):
    """Worker subprocesses + the parent's own share (it generates batches itself while workers start up) must give
    exactly the batches of the reference's serial protocol, in seed order (oracle/zstream.py)."""
    from ganspace_amd import _zgen
    from oracle import zstream
    monkeypatch.setenv("GANSPACE_ZGEN_WORKERS", "3")
    seeds = [11, 7, 123456, 42, 7, 99, 2_000_000_000, 5, 31337, 8]
    got = [np.array(z) for z in _zgen.generate(kind, seeds, 37, dim, 0.8)]
    assert len(got) == len(seeds)
    for s, z in zip(seeds, got):
        want = zstream.stylegan_z_batch(s, 37, dim) if kind == "stylegan" else zstream.biggan_z_batch(s, 37, dim, 0.8)
        np.testing.assert_array_equal(z, want)


def test_native_z_generator_is_bit_identical_to_numpy_randomstate():
    """csrc/gs_zgen.hip restates MT19937 + NumPy's legacy polar Gaussian (models/wrappers.py:167-174 draws z with
    ``RandomState(seed).standard_normal``): float32 rows identical bit for bit, for the seeds of the reference's
    stream (np.random.seed(1) -> 1791095845, ...), edge seeds and odd / even counts (the cached second value)."""
    import ctypes as C
    from ganspace_amd import _lib
    lib = _lib.load()
    for seed, count in [(1791095845, 4), (2135392491, 512 * 300 + 1), (0, 1), (1, 7), (2 ** 31 - 2, 100_000),
                        (2 ** 32 - 1, 12_345)]:
        out = np.empty(count, np.float32)
        assert lib.gs_zgen_fill(seed, count, out.ctypes.data_as(C.c_void_p)) == 0
        ref = np.random.RandomState(seed).standard_normal(count).astype(np.float32)
        assert out.tobytes() == ref.tobytes(), (seed, count)
    first = np.empty(4, np.float32)
    lib.gs_zgen_fill(1791095845, 4, first.ctypes.data_as(C.c_void_p))
    np.testing.assert_allclose(first, [0.76455638, -1.12429114, -0.13731647, 0.52814697], rtol=1e-6)


def test_native_z_generator_chunk_boundaries_and_short_counts():
    """The generator works on blocks of 624 MT19937 draws = 156 candidate pairs and evaluates the accepted ones in a
    separate pass (the latency chain log -> divide -> sqrt of one pair no longer serialises the next): every count
    from 0 to 40, counts around the number of values one block yields (~ 2 * 156 * pi / 4 = 245) and a few blocks
    yield, odd counts (the last pair gives one value) - all identical to ``RandomState(seed).standard_normal(count)``."""
    import ctypes as C
    from ganspace_amd import _lib
    lib = _lib.load()
    rs = np.random.RandomState(11)
    counts = list(range(41)) + list(range(225, 265)) + list(range(480, 500)) + [1607, 1608, 1609, 2411, 5119, 30_001]
    for count in counts:
        for seed in (3, int(rs.randint(0, 2 ** 31 - 1))):
            out = np.full(count + 2, 7.0, np.float32)
            assert lib.gs_zgen_fill(seed, count, out.ctypes.data_as(C.c_void_p)) == 0
            ref = np.random.RandomState(seed).standard_normal(count).astype(np.float32)
            assert out[:count].tobytes() == ref.tobytes(), (seed, count)
            assert out[count] == 7.0 and out[count + 1] == 7.0          # nothing written past the end


def test_native_z_stream_ring_backpressure_and_order():
    """More batches than ring slots, more threads than cores: batches come out in order, each equal to the NumPy
    stream of its seed, while slots are recycled only after release()."""
    from ganspace_amd import _zgen
    seeds = [int(s) for s in np.random.RandomState(5).randint(0, 2 ** 31 - 1, size=37)]
    stream = _zgen.NativeNormalStream(seeds, 300, 64, threads=6, pinned=False)
    assert stream._n_slots == 10
    got = []
    held = []
    for i, z in stream:
        held.append((i, z))
        if len(held) > 3:                       # keep three batches alive, like a consumer with copies in flight
            j, zj = held.pop(0)
            got.append(np.array(zj, copy=True))
            stream.release(j + 1)
    for j, zj in held:
        got.append(np.array(zj, copy=True))
    stream.close()
    assert len(got) == 37
    for s, z in zip(seeds, got):
        assert z.tobytes() == _zgen.stylegan_z(s, 300, 64).tobytes()
    # the generator front end used by decomposition._presample on hosts without a GPU
    out = list(_zgen.generate("stylegan", seeds[:5], 300, 64))
    assert all(o.tobytes() == _zgen.stylegan_z(s, 300, 64).tobytes() for s, o in zip(seeds, out))


def test_native_truncnorm_matches_scipy_float32_rows():
    """csrc/gs_zgen.hip restates ``scipy.stats.truncnorm.rvs(-2, 2, ..., random_state=RandomState(seed))`` (BigGAN's
    ``truncated_noise_sample``, models/biggan/pytorch_biggan/pytorch_pretrained_biggan/utils.py:21-33): one uniform per
    value through SciPy's logsumexp / ndtri_exp / Cephes-ndtri chain.  The float32 rows must equal SciPy's.  The
    only operation that is not the same code is NumPy's SIMD ``np.log`` on the uniform array (libm's ``log`` here: the two
    differ in ~0.4 % of the draws by one ulp of the float64 intermediate, which survives the float32 cast about once in
    1e9 values): budget = at most 2 float32-ulp-sized mismatches per million values, none seen on the seeds below."""
    import ctypes as C
    from scipy.stats import truncnorm
    from ganspace_amd import _lib, _zgen
    lib = _lib.load()
    la, lm = _zgen.TRUNCNORM_M2_P2
    from scipy import special as sc
    assert la == float(sc.log_ndtr(-2.0)) and lm == float(np.log1p(-sc.ndtr(-2.0) - sc.ndtr(-2.0)))
    total = bad = 0
    for seed, count, scale in [(1791095845, 2000 * 128, 1.0), (2135392491, 128 * 37 + 1, 0.8), (0, 1, 1.0), (1, 7, 0.5),
                               (2 ** 31 - 2, 311, 1.0), (2 ** 32 - 1, 312, 1.0), (5, 313, 1.0), (6, 624, 1.0), (7, 625, 1.0),
                               (8, 0, 1.0), (946286476, 100_000, 1.0)]:
        out = np.full(count + 2, 7.0, np.float32)
        assert lib.gs_zgen_fill_truncnorm(seed, count, la, lm, scale, out.ctypes.data_as(C.c_void_p)) == 0
        ref = truncnorm.rvs(-2, 2, size=max(count, 1), random_state=np.random.RandomState(seed)).astype(np.float32)[:count]
        ref = (np.float32(scale) * ref).astype(np.float32)
        assert out[count] == 7.0 and out[count + 1] == 7.0
        diff = out[:count] != ref
        bad += int(diff.sum())
        total += count
        if diff.any():                      # a mismatch may only be the float32 neighbour
            assert np.abs(out[:count][diff] - ref[diff]).max() <= np.spacing(np.abs(ref[diff]).max())
        assert np.abs(out[:count]).max(initial=0.0) <= 2.0 * scale
    assert bad <= max(2, 2 * total // 1_000_000), (bad, total)
    assert lib.gs_zgen_fill_truncnorm(1, 4, 0.5, lm, 1.0, out.ctypes.data_as(C.c_void_p)) == -1      # log-probabilities are < 0


def test_native_biggan_stream_equals_reference_protocol():
    """The thread-pool front end for BigGAN latents (``NativeNormalStream(kind="biggan")``, used by ``_presample`` and the
    regression): batches in seed order, equal to the reference's ``truncation * truncnorm.rvs(...)`` float32 batches."""
    from ganspace_amd import _zgen
    from oracle import zstream
    seeds = [int(s) for s in np.random.RandomState(9).randint(0, 2 ** 31 - 1, size=23)]
    got = list(_zgen.generate("biggan", seeds, 41, 128, 0.7))
    assert len(got) == 23
    for s, z in zip(seeds, got):
        want = zstream.biggan_z_batch(s, 41, 128, 0.7)
        assert z.shape == want.shape and (z != want).sum() <= 1 and np.abs(z - want).max() <= 1e-7


def test_two_live_native_streams_do_not_share_a_ring():
    """Two streams of the same shape alive at once (zip of two generators, a generator that was not exhausted) used to be
    handed the SAME cached ring and overwrote each other's slots (round-3 advisor finding).  A ring belongs to one
    stream at a time; it returns to the cache when its owner closes."""
    from ganspace_amd import _zgen
    seeds_a = [int(s) for s in np.random.RandomState(1).randint(0, 2 ** 31 - 1, size=20)]
    seeds_b = [int(s) for s in np.random.RandomState(2).randint(0, 2 ** 31 - 1, size=20)]
    ga = _zgen.generate("stylegan", seeds_a, 200, 32)
    gb = _zgen.generate("stylegan", seeds_b, 200, 32)
    for sa, sb, (za, zb) in zip(seeds_a, seeds_b, zip(ga, gb)):
        assert za.tobytes() == _zgen.stylegan_z(sa, 200, 32).tobytes()
        assert zb.tobytes() == _zgen.stylegan_z(sb, 200, 32).tobytes()
    ga.close()                              # (suspended behind their last yield: closing runs their `finally: stream.close()`)
    gb.close()
    assert all(e[2] is False for e in _zgen._RING_CACHE.values())
    s1 = _zgen.NativeNormalStream(seeds_a[:6], 200, 32, threads=2, pinned=False)
    s2 = _zgen.NativeNormalStream(seeds_b[:6], 200, 32, threads=2, pinned=False)
    assert s1._storage is not s2._storage
    s1.close()
    s2.close()
    s3 = _zgen.NativeNormalStream(seeds_a[:6], 200, 32, threads=2, pinned=False)
    assert not any(s3._storage is s._storage for s in (s1, s2)) or s3._ring[2]
    assert s3._ring[2] is True
    s3.close()
    assert all(e[2] is False for e in _zgen._RING_CACHE.values())


def test_regression_staging_is_bounded_in_rows_and_bytes():
    """``linreg_lstsq`` stages [A|Z] rows per Gram call: as many whole mini-batches as fit 65 536 rows / 256 MB, at least
    one mini-batch and at least one forward group."""
    from ganspace_amd.decomposition import _regression_stage_rows as stage
    assert stage(2000, 208, 2000) == 64000             # cfg3: 500 mini-batches -> 16 Gram calls (was 250)
    assert stage(10000, 592, 80000) == 80000           # cfg4: one forward group of 8 mini-batches per Gram call
    assert stage(10000, 592, 10000) == 60000
    assert stage(10000, 8192, 10000) == 10000          # a mini-batch wider than the byte budget still goes through whole
    assert stage(70000, 208, 70000) == 70000
    for B, wp in [(1, 8), (250, 208), (4096, 4096), (3, 7)]:
        n = stage(B, wp, B)
        assert n >= B and n % B == 0 and (n == B or (n <= 65536 and n * wp * 4 <= 256 << 20))


def test_forward_rows_groups_narrow_layers_only():
    """Rows per ``partial_forward`` call: several mini-batches for a narrow layer (Z-space ``style``: the mapping GEMM needs
    tens of thousands of rows to fill the chip), the configured mini-batch for wide ones; always whole mini-batches."""
    from ganspace_amd.decomposition import _forward_rows as fr
    assert fr(10_000, 512) == 80_000                   # cfg4: 8 mini-batches per call
    assert fr(512, 512) == 81_920 // 512 * 512         # cfg1
    assert fr(2000, 32_768) == 2000                    # cfg3: 262 MB per mini-batch
    assert fr(500, 131_072) == 500                     # cfg5
    assert fr(100_000, 512) == 100_000                 # a mini-batch above the cap stays whole
    for B, d in [(1, 4), (20, 8192), (333, 37), (4096, 2048)]:
        r = fr(B, d)
        assert r >= B and r % B == 0
    # a model has to vouch for the layer: BigGAN's 128-wide `embeddings` hook sits in front of the 32 768-wide gen_z that
    # partial_forward evaluates too - 80 000 rows of it would be a 10 GB temporary
    from types import SimpleNamespace
    from ganspace_amd.wrappers import BaseModel, StyleGAN2
    assert fr(2000, 128, SimpleNamespace(cheap_prefix=lambda name: False), "embeddings") == 2000
    assert fr(2000, 128, SimpleNamespace(), "embeddings") == 2000
    assert fr(10_000, 512, SimpleNamespace(cheap_prefix=lambda name: "style" in name), "style") == 80_000
    assert BaseModel.cheap_prefix(None, "anything") is False
    assert StyleGAN2.cheap_prefix(None, "style") and StyleGAN2.cheap_prefix(None, "strided_style")
    assert not StyleGAN2.cheap_prefix(None, "convs.2")


@pytest.mark.parametrize("k,demod,up", [(3, True, False), (3, True, True), (1, False, False)])
def test_modulated_conv_shared_weight_form_equals_the_grouped_form(k, demod, up):
    """``ModulatedConv2d.forward`` (input scaling -> one dense convolution -> per-(sample, channel) demodulation) is the
    same map as the published per-sample-weight / ``groups = B`` form (``forward_grouped``), float64 to roundoff."""
    import torch
    from ganspace_amd.wrappers import ModulatedConv2d
    torch.manual_seed(3)
    m = ModulatedConv2d(12, 7, k, 16, demodulate=demod, upsample=up).double()
    x = torch.randn(5, 12, 6, 6, dtype=torch.float64)
    style = torch.randn(5, 16, dtype=torch.float64)
    with torch.no_grad():
        a, b = m(x, style), m.forward_grouped(x, style)
    assert a.shape == b.shape == (5, 7, 12 if up else 6, 12 if up else 6)
    assert torch.allclose(a, b, rtol=1e-10, atol=1e-12)


@pytest.mark.parametrize("k,cin,cout", [(3, 12, 8), (1, 10, 3), (3, 5, 7)])
def test_modulated_conv_im2col_product_equals_conv2d(monkeypatch, k, cin, cout):
    """``ModulatedConv2d._conv_hip`` (device path: NHWC patches x weight matrix through ``gs_linear_forward``) indexes the
    same products as ``F.conv2d``: the GEMM call is replaced by a float64 matmul here, the patch / weight layout, the
    padding to multiples of 4 columns and the sub-batching are the product's own."""
    import torch
    import torch.nn.functional as F
    from ganspace_amd import wrappers
    torch.manual_seed(5)
    m = wrappers.ModulatedConv2d(cin, cout, k, 16).double()
    x = torch.randn(6, cin, 5, 7, dtype=torch.float64)
    calls = []

    def fake_linear(cols, w, b):
        assert cols.shape[1] % 4 == 0 and cols.shape[1] == w.shape[1] and b is None
        calls.append(cols.shape[0])
        return cols @ w.T
    monkeypatch.setattr(wrappers.ops, "linear_forward", fake_linear)
    with torch.no_grad():
        got = m._conv_hip(x)
        ref = F.conv2d(x, m.scale * m.weight[0], padding=k // 2)
    assert got.shape == ref.shape and torch.allclose(got, ref, rtol=1e-12, atol=1e-12)
    assert sum(calls) == 6 * 5 * 7
    # a weight update invalidates the cached matrix
    with torch.no_grad():
        m.weight.mul_(2.0)
        assert torch.allclose(m._conv_hip(x), 2 * ref, rtol=1e-12, atol=1e-12)


def test_device_z_generator_switch(monkeypatch):
    """``_zgen.device_generation_enabled``: HIP devices use the device generator unless ``GANSPACE_ZGEN=host`` asks for the
    host thread pool; CPU runs never do (no device code without a device)."""
    from ganspace_amd import _zgen
    monkeypatch.delenv("GANSPACE_ZGEN", raising=False)
    assert _zgen.device_generation_enabled("cuda:0") is True
    assert _zgen.device_generation_enabled("cpu") is False
    monkeypatch.setenv("GANSPACE_ZGEN", "host")
    assert _zgen.device_generation_enabled("cuda:0") is False


def test_bench_builds_the_launcher_line_for_multi_gpu_runs():
    """``bench.py --gpus N`` without WORLD_SIZE re-executes itself under the launcher the driver uses for N > 1
    (``python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py ...``)."""
    import importlib.util
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_for_test", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    cmd = bench.launcher_command(["--gpus", "8", "--steps", "20", "--warmup", "5"], 8, port=29511)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29511"
    i = cmd.index(os.path.join(root, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"]
    auto = bench.launcher_command([], 2)                       # a free port is picked when none is given
    assert 1024 <= int(auto[auto.index("--master-port") + 1]) <= 65535


def test_mt19937_jump_polynomials_reach_the_states_numpy_reaches():
    """ganspace_amd/data/mt19937_jump_L2048.npz (tools/make_mt_jump.py): polynomial i is x^(i * 2048 * 624) mod phi, and the
    state block of a stream at that offset is the XOR of the windows words[k : k + 624] of its first 33 blocks over the set
    bits k - checked here against the state ``RandomState(seed)`` has after that many draws.  Also the segment plan: enough
    blocks for the values asked for, with the margin the generator relies on."""
    from ganspace_amd import _zgen
    polys = _zgen.jump_polys_host()
    L = _zgen.JUMP_BLOCK_LEN
    assert polys.shape == (63, 624) and polys.dtype == np.uint32
    assert (polys[:, 623] >> 1 == 0).all()                       # degree < 19937: bits 19937 .. 19967 are clear
    seed = 424242
    rs = np.random.RandomState(seed)
    words = []
    for _ in range(33):
        rs.random_sample(312)                                    # 624 draws = one block of state words
        words.append(np.array(rs.get_state()[1], dtype=np.uint32))
    words = np.concatenate(words)
    for i in (1, 2, 17):
        ks = np.nonzero(np.unpackbits(polys[i - 1].view(np.uint8), bitorder="little")[:19937])[0]
        got = np.zeros(624, dtype=np.uint32)
        for lo in range(0, len(ks), 2048):
            got ^= np.bitwise_xor.reduce(words[ks[lo:lo + 2048, None] + np.arange(624)[None, :]], axis=0)
        ref = np.random.RandomState(seed)
        ref.random_sample((i * L + 1) * 312)
        np.testing.assert_array_equal(got, np.array(ref.get_state()[1], dtype=np.uint32))
    for count in (40_960, 512_000, 2_560_000, 5_120_000, 15_000_000):
        s = _zgen.plan_segments(count)
        assert s >= 1 and s * L * _zgen.VALUES_PER_BLOCK >= count * 1.004
        assert (s - 1) * L * _zgen.VALUES_PER_BLOCK < count * 1.004 + 3 * _zgen.VALUES_PER_BLOCK
    assert _zgen.plan_segments(40_960) == 1 and _zgen.plan_segments(5_120_000) == 11
    assert _zgen.plan_segments(64 * L * 250) == 0                # beyond the file: the serial kernel takes it


def test_parallel_npz_writer_is_a_drop_in_for_numpys(tmp_path):
    """``_npz.savez_compressed`` (the component file of decomposition.py:331-341 with its members deflated piecewise on a thread
    pool): a valid zip archive with the keys, dtypes, shapes and values ``np.savez_compressed`` would have written - members
    longer than one piece, shorter than one, empty and 0-d included -, the ``.npz`` suffix added when missing."""
    import zipfile
    from ganspace_amd import _npz
    rs = np.random.RandomState(5)
    rec = dict(act_comp=rs.standard_normal((40, 16384)).astype(np.float32),          # 2.6 MB: three pieces
               act_mean=rs.standard_normal((1, 16384)).astype(np.float32), act_stdev=rs.rand(40).astype(np.float32),
               lat_comp=rs.standard_normal((40, 1, 128)).astype(np.float32), empty=np.zeros((0, 3), np.float32),
               scalar=np.float32(3.5), counts=np.arange(7, dtype=np.int64))
    for threads in (1, 3):
        path = tmp_path / f"out{threads}"
        _npz.savez_compressed(path, threads=threads, **rec)
        path = tmp_path / f"out{threads}.npz"
        assert path.exists()
        with zipfile.ZipFile(path) as z:
            assert z.testzip() is None
            assert [i.filename for i in z.infolist()] == [k + ".npy" for k in rec]
            assert all(i.compress_type == zipfile.ZIP_DEFLATED for i in z.infolist())
        ref = tmp_path / "ref.npz"
        np.savez_compressed(ref, **rec)
        with np.load(path) as got, np.load(ref) as want:
            assert got.files == want.files
            for k in rec:
                assert got[k].dtype == want[k].dtype and got[k].shape == want[k].shape
                np.testing.assert_array_equal(got[k], want[k])
