"""``gs_ipca_allreduce`` with MORE THAN ONE rank, executed: the C entry resolves RCCL with dlsym, so an in-process stand-in
(``tests/shim/fake_rccl.cpp``: P host threads, one communicator object each, rendezvous + host arithmetic in rank order)
lets P estimator handles on the box's single GPU go through the entry's whole P > 1 path - the [n | n mean] header
all-reduce, the Chan re-centring of the local scatter about the global mean (also for a rank that saw no samples), the
scatter all-reduce and import; for the faithful estimators the all-gather layout and ``gs_ipca_lowrank_merge``.  RCCL
itself refuses two ranks on one device (the only N > 1 RCCL run is the driver's 8-GPU bench), so this is the one place
where that arithmetic runs before then.  Reference loop being sharded: ``/root/reference/decomposition.py:245-267``."""
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import ipca as O

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM_DIR = os.path.join(ROOT, "tests", "shim")


@pytest.fixture(scope="module")
def shim():
    so = os.path.join(SHIM_DIR, "libfake_rccl.so")
    src = os.path.join(SHIM_DIR, "fake_rccl.cpp")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.run(["/opt/rocm/bin/hipcc", "-w", "-O2", "-shared", "-fPIC", src, "-o", so], check=True)
    return so


def _run(mode, P, empty, tmp_path):
    out = str(tmp_path / f"{mode}_{P}_{empty}.npz")
    r = subprocess.run([sys.executable, os.path.join(SHIM_DIR, "allreduce_worker.py"), mode, str(P), str(empty), out],
                       cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "OK" in r.stdout, r.stderr[-4000:]
    return np.load(out)


@pytest.mark.parametrize("P,empty", [(2, -1), (3, 1), (2, 0)])
def test_exact_allreduce_entry_with_several_ranks(shim, tmp_path, P, empty):
    """Every rank ends with the statistics of ALL rows: components / singular values / mean equal the float64 exact PCA of
    the concatenated shards, identical on every rank (the stand-in hands all ranks the same bits)."""
    z = _run("exact", P, empty, tmp_path)
    X = z["all_rows"].astype(np.float64)
    ref = O.exact_pca([X], 8)
    assert int(z["allreduce_calls"]) == 2 and int(z["allgather_calls"]) == 0
    for r in range(P):
        assert int(z[f"n{r}"]) == len(X)
        cos = O.signed_cosines(z[f"comp{r}"], ref["components_"])
        assert cos.min() > 1 - 1e-9, (r, cos)
        # (the contraction is exact-float32 MFMA: singular values agree with the float64 PCA to f32-accumulation level)
        np.testing.assert_allclose(z[f"sv{r}"], ref["singular_values_"], rtol=2e-7)
        np.testing.assert_allclose(z[f"mean{r}"], X.mean(0), atol=1e-9)
        assert np.array_equal(z[f"comp{r}"], z["comp0"]) and np.array_equal(z[f"sv{r}"], z["sv0"])


@pytest.mark.parametrize("P,empty", [(2, -1), (3, 2)])
def test_faithful_allgather_entry_with_several_ranks(shim, tmp_path, P, empty):
    """``--est=ipca`` sharded: all-gather of the per-rank low-rank states, one more step of the recurrence on every rank.
    Ranks agree bit for bit; the merged leading components are those of the exact PCA of all rows (the data is
    rank-24 plus a little noise, k = 8: truncation order does not matter for the first components)."""
    z = _run("faithful", P, empty, tmp_path)
    X = z["all_rows"].astype(np.float64)
    ref = O.exact_pca([X], 8)
    assert int(z["allgather_calls"]) == 1 and int(z["allreduce_calls"]) == 0
    for r in range(P):
        assert int(z[f"n{r}"]) == len(X)
        assert np.array_equal(z[f"comp{r}"], z["comp0"])
        np.testing.assert_allclose(z[f"mean{r}"], X.mean(0), atol=1e-10)
        cos = np.abs(O.signed_cosines(z[f"comp{r}"], ref["components_"]))
        assert cos[:4].min() > 0.999, cos
