"""End-to-end BASELINE config 2 through the drop-in API (get_or_compute), wall-clock per phase."""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from types import SimpleNamespace
import numpy as np, torch
from ganspace_amd.config import Config
from ganspace_amd.decomposition import get_or_compute
from ganspace_amd.wrappers import get_instrumented_model

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dev = torch.device("cuda", 0)
inst = get_instrumented_model("StyleGAN2", "ffhq", "style", dev, use_w=True)
out = {}
for est in ("ipca-exact", "ipca"):
    cfg = Config(model="StyleGAN2", layer="style", output_class="ffhq", use_w=True, n=n, batch_size=10_000,
                 components=80, estimator=est)
    with tempfile.TemporaryDirectory() as td:
        sub = SimpleNamespace(run_dir_root=td, run_dir=td)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        path = get_or_compute(cfg, inst, submit_config=sub)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        d = np.load(path)
        out[est] = d["act_comp"].reshape(80, -1).copy()
        print(f"E2E {est}: n={n} wall {dt:.2f} s  ({n/dt:.0f} samples/s end to end)  file {path.name}", flush=True)
c = np.abs(np.sum(out["ipca-exact"][:20] * out["ipca"][:20], axis=1))
print("top-20 |cos| exact vs faithful:", c.min())
