"""The device latent generator (csrc/gs_zgen_device.hip, one workgroup of four waves per seed) against the generators the reference calls:
``np.random.RandomState(seed).standard_normal`` (models/wrappers.py:167-174) and BigGAN's
``truncnorm.rvs(-2, 2, random_state=RandomState(seed))`` (biggan/.../utils.py:21-33).

The integer part of the stream (MT19937, the 53-bit doubles, the polar method's accept / reject decisions, the order of
the accepted pairs) must agree exactly - one flipped decision would shift every later value.  The float64 ``log`` /
``exp`` behind the accepted values are the device libm's instead of glibc's: the float32 rows may differ in isolated
values by ONE float32 ulp (expected: about one value in 1e8); the bound asserted here is <= 1 ulp, <= 1e-5 of the values."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a HIP device")
    return torch.device("cuda", 0)


def _ulp_diff(a, b):
    ai = a.view(np.int32).astype(np.int64)
    bi = b.view(np.int32).astype(np.int64)
    ai = np.where(ai < 0, np.int64(-2147483648) - ai, ai)      # order-preserving map of the float32 bit patterns
    bi = np.where(bi < 0, np.int64(-2147483648) - bi, bi)
    return np.abs(ai - bi)


def _check_rows(got, want):
    assert got.shape == want.shape and got.dtype == want.dtype == np.float32
    d = _ulp_diff(got, want)
    assert d.max() <= 1, (d.max(), int((d > 0).sum()))
    assert (d > 0).mean() <= 1e-5, int((d > 0).sum())


@pytest.mark.parametrize("n,dim", [(37, 64), (3, 5), (1, 1), (1, 2), (61, 4), (1000, 512), (4001, 129)])
def test_device_normals_match_numpy_randomstate(dev, n, dim):
    from ganspace_amd import _zgen
    seeds = [0, 1, 12345, 2 ** 31 - 2, 987654321]
    got = {i: z.cpu().numpy() for i, z in _zgen.device_batches("stylegan", seeds, n, dim, dev)}
    assert sorted(got) == list(range(len(seeds)))
    for i, s in enumerate(seeds):
        want = np.random.RandomState(s).standard_normal(n * dim).astype(np.float32).reshape(n, dim)
        _check_rows(got[i], want)


@pytest.mark.parametrize("n,dim,trunc", [(20, 128, 1.0), (7, 128, 0.7), (1, 3, 1.0), (500, 128, 0.4)])
def test_device_truncated_normals_match_scipy(dev, n, dim, trunc):
    from scipy.stats import truncnorm
    from ganspace_amd import _zgen
    seeds = [3, 250, 2 ** 31 - 7]
    got = {i: z.cpu().numpy() for i, z in _zgen.device_batches("biggan", seeds, n, dim, dev, truncation=trunc)}
    for i, s in enumerate(seeds):
        vals = truncnorm.rvs(-2, 2, size=(n, dim), random_state=np.random.RandomState(s)).astype(np.float32)
        _check_rows(got[i], (np.float32(trunc) * vals).astype(np.float32))
        assert np.abs(got[i]).max() <= 2.0 * trunc + 1e-6


def test_device_generator_groups_and_many_seeds(dev):
    """More seeds than one launch group takes: the groups tile the seed list in order."""
    from ganspace_amd import _zgen
    seeds = list(range(100, 100 + 21))
    got = [z.cpu().numpy() for _, z in _zgen.device_batches("stylegan", seeds, 16, 8, dev, group=8)]
    assert len(got) == 21
    for s, g in zip(seeds, got):
        _check_rows(g, np.random.RandomState(s).standard_normal(128).astype(np.float32).reshape(16, 8))


def test_presample_device_and_host_generators_agree(dev, monkeypatch):
    """``decomposition._presample`` through the device generator and through the host thread pool (``GANSPACE_ZGEN=host``):
    the same latents (W space: the mapping network of equal z rows) and the same state of the global stream afterwards."""
    from ganspace_amd import decomposition as dec
    from ganspace_amd.wrappers import get_instrumented_model
    inst = get_instrumented_model("StyleGAN2", "ffhq", "style", dev)
    model = inst.model
    model.use_w()
    plan = dec._Plan.make(3000, 500, 20)
    outs = []
    for mode in ("device", "host"):
        monkeypatch.setenv("GANSPACE_ZGEN", mode)
        np.random.seed(dec.SEED_SAMPLING)
        lat, row0 = dec._presample(model, plan, model.get_latent_shape(), dev)
        outs.append((lat.cpu().numpy(), row0, np.random.randint(1 << 30)))
    (a, ra, sa), (b, rb, sb) = outs
    assert ra == rb == 0 and sa == sb and a.shape == b.shape
    # the mapping network amplifies a one-ulp difference of a z entry: allow isolated rows to differ at float32 roundoff
    rel = np.abs(a - b).max(axis=1) / np.abs(b).max()
    assert (rel > 1e-5).mean() <= 1e-3 and rel.max() < 1e-3, (rel.max(), (rel > 1e-5).mean())
    inst.close()


def test_long_sample_latent_streams_come_from_the_device_generator(dev, monkeypatch):
    """``StyleGAN2.sample_latent`` (models/wrappers.py:167-174) with a long stream - the 5 000 fresh latents behind
    ``lat_stdev`` - takes the device generator: same stream as NumPy's (isolated float32 values one ulp apart), and the host
    generator is not called; a short one (one latent) stays on NumPy, bit for bit."""
    from ganspace_amd import _zgen
    from ganspace_amd.wrappers import StyleGAN2
    model = StyleGAN2(dev, "car")                       # Z space: latent_from_z is the identity
    n = model.DEVICE_SAMPLE_VALUES // 512 + 3
    want = np.random.RandomState(77).standard_normal(n * 512).astype(np.float32).reshape(n, 512)
    short = model.sample_latent(4, seed=77).cpu().numpy()
    np.testing.assert_array_equal(short, np.random.RandomState(77).standard_normal(4 * 512).astype(np.float32).reshape(4, 512))
    monkeypatch.setattr(_zgen, "stylegan_z", lambda *a, **k: pytest.fail("host generator called for a long stream"))
    got = model.sample_latent(n, seed=77)
    assert got.is_cuda and got.shape == (n, 512)
    _check_rows(got.cpu().numpy(), want)


def test_streams_cut_into_segments_equal_the_serial_streams_bit_for_bit(dev, monkeypatch):
    """Few, long streams are cut into segments of 2 048 blocks that start from jumped-ahead MT19937 states
    (gs_zgen_device_segmented; polynomials from ganspace_amd/data, checked against NumPy in the CPU suite).  Same arithmetic
    per value as the one-workgroup-per-stream kernel: the rows are BIT-identical to it, whatever the number of segments, for
    even and odd lengths, into a caller's array as well - and equal to NumPy up to the usual isolated float32 ulp."""
    from ganspace_amd import _zgen
    seeds = [5, 77, 2 ** 31 - 3]
    for n, dim in [(2500, 512), (1000, 513), (10_000, 512)]:        # 3, 2 and 11 segments
        count = n * dim
        assert _zgen.plan_segments(count) >= 2
        monkeypatch.setenv("GANSPACE_ZGEN_SEGMENTS", "0")
        serial = [z.clone() for _, z in _zgen.device_batches("stylegan", seeds, n, dim, dev)]
        monkeypatch.setenv("GANSPACE_ZGEN_SEGMENTS", "1")
        out = torch.full((len(seeds), n, dim), float("nan"), device=dev)
        cut = [z for _, z in _zgen.device_groups("stylegan", seeds, n, dim, dev, out=out)]
        assert len(cut) == 1 and cut[0].data_ptr() == out.data_ptr()
        for i, s in enumerate(seeds):
            assert torch.equal(out[i], serial[i]), (n, dim, s)
        _check_rows(out[1].cpu().numpy(), np.random.RandomState(seeds[1]).standard_normal(count).astype(np.float32).reshape(n, dim))
    # an odd number of values per stream: the last pair is cut in two
    n, dim = 1501, 333
    assert (n * dim) % 2 == 1 and _zgen.plan_segments(n * dim) >= 2
    got = {i: z.cpu().numpy() for i, z in _zgen.device_batches("stylegan", seeds, n, dim, dev)}
    for i, s in enumerate(seeds):
        _check_rows(got[i], np.random.RandomState(s).standard_normal(n * dim).astype(np.float32).reshape(n, dim))


def test_segmented_generator_reports_a_stream_that_ran_short(dev):
    """Two segments of 2 048 blocks hold about a million normals: asked for 1.2 M, the call must say so (the caller then
    takes the serial kernel) instead of leaving the tail of the row unwritten."""
    import ctypes as C
    from ganspace_amd import _lib, _zgen
    lib = _lib.load()
    count = 1_200_000
    seeds = torch.tensor([3, 4], dtype=torch.int32, device=dev)
    out = torch.empty((2, count), device=dev)
    nbytes = C.c_int64(0)
    _lib.check(lib.gs_zgen_segmented_nbytes(2, 2, _zgen.JUMP_BLOCK_LEN, C.cast(C.byref(nbytes), C.c_void_p)))
    scratch = torch.empty(nbytes.value, dtype=torch.uint8, device=dev)
    polys = torch.from_numpy(_zgen.jump_polys_host().view(np.int32)).to(dev)
    short = C.c_int(0)
    _lib.check(lib.gs_zgen_device_segmented(C.c_void_p(seeds.data_ptr()), 2, count, C.c_void_p(out.data_ptr()), count,
                                            C.c_void_p(polys.data_ptr()), _zgen.JUMP_BLOCK_LEN, 2, C.c_void_p(scratch.data_ptr()),
                                            nbytes.value, C.cast(C.byref(short), C.c_void_p), _lib.current_stream_ptr()))
    assert short.value == 1
    short.value = 7
    _lib.check(lib.gs_zgen_device_segmented(C.c_void_p(seeds.data_ptr()), 2, 900_000, C.c_void_p(out.data_ptr()), count,
                                            C.c_void_p(polys.data_ptr()), _zgen.JUMP_BLOCK_LEN, 2, C.c_void_p(scratch.data_ptr()),
                                            nbytes.value, C.cast(C.byref(short), C.c_void_p), _lib.current_stream_ptr()))
    assert short.value == 0
    _check_rows(out[1, :900_000].cpu().numpy(), np.random.RandomState(4).standard_normal(900_000).astype(np.float32))
