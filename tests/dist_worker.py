"""Worker of tests/test_gpu_distributed.py: one rank of a world_size-N run of the product's sharded paths.

    RANK=r WORLD_SIZE=N MASTER_ADDR=127.0.0.1 MASTER_PORT=p python tests/dist_worker.py <case> <run_dir>

All ranks share the ONE GPU a test box has, so the process group uses the gloo backend (which accepts device
tensors for all_reduce / all_gather / broadcast); everything else - shard plan, per-rank z generation, estimator
state exchange, head broadcast, sharded regression, rank-0 write - is the code an 8-GPU RCCL run executes.
"""
import os
import sys
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

CASES = {
    # cfg2 in miniature: W-space, additive statistics, one all-reduce before the eigensolve
    "w_exact": dict(model="StyleGAN2", layer="style", output_class="ffhq", use_w=True, n=40_000, batch_size=2000,
                    components=20, estimator="ipca-exact"),
    # cfg4 in miniature: Z-space -> regression back to the latent space (sharded, all-reduced normal equations)
    "z_exact": dict(model="StyleGAN2", layer="style", output_class="car", use_w=False, n=12_000, batch_size=1000,
                    components=10, estimator="ipca-exact"),
    # the reference's default estimator, sharded: per-rank sklearn-faithful recurrence + low-rank merge
    "w_ipca": dict(model="StyleGAN2", layer="style", output_class="ffhq", use_w=True, n=40_000, batch_size=2000,
                   components=20, estimator="ipca"),
    # a wide layer (small-side recurrence per rank + low-rank merge), Z-space regression included
    "wide_ipca": dict(model="BigGAN-512", layer="generator.gen_z", output_class="husky", use_w=False, n=8_000,
                      batch_size=500, components=10, estimator="ipca"),
}


def main():
    case, run_dir = sys.argv[1], sys.argv[2]
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from ganspace_amd.config import Config
    from ganspace_amd.decomposition import get_or_compute
    from ganspace_amd.wrappers import get_instrumented_model
    kw = dict(CASES[case])
    if kw["output_class"] == "husky":
        kw["output_class"] = 250           # class names need nltk: use the ImageNet id
    cfg = Config(**kw)
    inst = get_instrumented_model(cfg.model, cfg.output_class, cfg.layer, dev, use_w=cfg.use_w)
    sub = SimpleNamespace(run_dir_root=run_dir, run_dir=run_dir)
    path = get_or_compute(cfg, inst, submit_config=sub)
    if rank == 0:
        print("WROTE", path, flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
