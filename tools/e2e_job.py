"""One end-to-end get_or_compute job (BASELINE cfg3 / cfg5 shapes) with the phase times of decomposition.LAST_TIMINGS -
the command the rocprofv3 kernel traces under profiles/ are taken of.

    python tools/e2e_job.py cfg3 [n] [batch]      BigGAN-512 generator.gen_z (d = 32 768)
    python tools/e2e_job.py cfg5 [n] [batch]      StyleGAN2 convs.2          (d = 131 072)
    python tools/e2e_job.py cfg2 [n] [batch]      StyleGAN2 W space, ipca-exact (d = 512; cfg2f: the faithful `ipca`)
    python tools/e2e_job.py cfg4 [n] [batch]      StyleGAN2-car Z space --layer=style, n = 8e6, ipca-exact (cfg4f: `ipca`)
GS_E2E_PROFILE=0: wall time only (the phase timers synchronise the device between phases; without them the host runs ahead).
"""
import contextlib, json, os, shutil, sys, tempfile, time
from types import SimpleNamespace
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ganspace_amd import decomposition as dec
from ganspace_amd.config import Config
from ganspace_amd.wrappers import get_instrumented_model

which = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
if which == "cfg3":
    kw = dict(model="BigGAN-512", layer="generator.gen_z", output_class=250, n=1_000_000, batch_size=2000, components=80,
              estimator="ipca")
elif which in ("cfg2", "cfg2f"):
    kw = dict(model="StyleGAN2", layer="style", output_class="ffhq", use_w=True, n=1_000_000, batch_size=10_000, components=80,
              estimator="ipca-exact" if which == "cfg2" else "ipca")
elif which in ("cfg4", "cfg4f"):
    kw = dict(model="StyleGAN2", layer="style", output_class="car", n=8_000_000, batch_size=10_000, components=80,
              estimator="ipca-exact" if which == "cfg4" else "ipca")
else:
    kw = dict(model="StyleGAN2", layer="convs.2", output_class="ffhq", n=20_000, batch_size=250, components=80,
              estimator="ipca")
if len(sys.argv) > 2:
    kw["n"] = int(sys.argv[2])
if len(sys.argv) > 3:
    kw["batch_size"] = int(sys.argv[3])
dev = torch.device("cuda", 0)
cfg = Config(**kw)
run_dir = tempfile.mkdtemp(prefix="gs_e2e_")
try:
    inst = get_instrumented_model(cfg.model, cfg.output_class, cfg.layer, dev, **({"use_w": True} if kw.get("use_w") else {}))
    torch.cuda.synchronize()
    dec.PROFILE = os.environ.get("GS_E2E_PROFILE", "1") != "0"     # 0: no phase timers, i.e. no synchronisation between the phases
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(sys.stderr):
        dec.get_or_compute(cfg, inst, submit_config=SimpleNamespace(run_dir_root=run_dir, run_dir=run_dir))
    wall = time.perf_counter() - t0
    print(json.dumps({"job": which, "config": kw, "wall_s": round(wall, 3),
                      "phases": {k: (round(v, 4) if isinstance(v, float) else v) for k, v in dec.LAST_TIMINGS.items()}}))
finally:
    shutil.rmtree(run_dir, ignore_errors=True)
