"""How long does pinning the z ring take, and does splitting the allocation over threads help?  """
import time, threading, torch
torch.cuda.init()
def pin(nbytes):
    t = torch.empty(nbytes, dtype=torch.uint8, pin_memory=True)
    return t
for total_mb, parts, threads in ((1400, 1, 1), (1400, 68, 1), (1400, 68, 8), (1400, 68, 16), (400, 20, 1), (400, 20, 8)):
    per = total_mb * (1 << 20) // parts
    keep = [None] * parts
    t0 = time.perf_counter()
    if threads == 1:
        for i in range(parts):
            keep[i] = pin(per)
    else:
        def work(lo):
            for i in range(lo, parts, threads):
                keep[i] = pin(per)
        ths = [threading.Thread(target=work, args=(j,)) for j in range(threads)]
        [t.start() for t in ths]; [t.join() for t in ths]
    dt = time.perf_counter() - t0
    print(f"pin {total_mb} MB as {parts} pieces on {threads} threads: {dt*1e3:.1f} ms", flush=True)
    del keep
    torch.cuda.empty_cache()
    import gc; gc.collect()
    torch._C._host_emptyCache() if hasattr(torch._C, "_host_emptyCache") else None
