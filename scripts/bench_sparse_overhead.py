"""Device + host time of shard.GradSync.finish() on ONE rank (RCCL group of one: the collectives move nothing) for a table-shaped gradient whose
rows are filled like the post-prune regime's (profiles/r06_touched_rows_post_prune_8_ranks.json): what the touched-rows exchange ADDS on top of its
collective - mask, bit-packing, slot map, gather, rewrite - against the dense exchange's in-place all-reduce.
usage: python scripts/bench_sparse_overhead.py"""
import json
import os
import socket
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from pagnerf_amd import shard

with socket.socket() as s:
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
shard.FORCE_COLLECTIVES = True
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
fills = [x["union_fill"] for x in json.load(open(os.path.join(root, "profiles", "r06_touched_rows_post_prune_8_ranks.json")))["levels"]]
L, T, F = 24, 1 << 18, 2
gen = torch.Generator(device=dev).manual_seed(0)
keep = torch.rand(L, T, device=dev, generator=gen) < torch.tensor(fills, device=dev)[:, None]
g0 = torch.randn(L, T, F, device=dev, generator=gen) * keep[..., None]
for name, kw in (("dense fp32 all-reduce", dict()), ("dense bf16 direct reduce", dict(comm_dtype=torch.bfloat16)),
                 ("touched rows, bounded, fp32", dict(sparse="bounded")), ("touched rows, bounded, bf16", dict(sparse="bounded", comm_dtype=torch.bfloat16)),
                 ("touched rows, exact, fp32", dict(sparse="exact"))):
    tab = torch.nn.Parameter(torch.zeros(L, T, F, device=dev))
    sy = shard.GradSync([tab], big=1 << 16, **kw)
    for _ in range(4):
        tab.grad = g0.clone()
        sy.finish()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    grads = [g0.clone() for _ in range(n)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    a.record()
    for i in range(n):
        tab.grad = grads[i]
        sy.finish()
    b.record()
    t_issue = (time.perf_counter() - t0) / n * 1e3
    torch.cuda.synchronize()
    st = sy.sparse_stats()
    print("%-30s device %.3f ms per table  host issue %.3f ms  %s" % (name, a.elapsed_time(b) / n, t_issue,
          ("exchanged %.1f of %.1f MB, %d whole levels" % (st[0]["exchanged_bytes"] / 1e6, st[0]["dense_bytes"] / 1e6, st[0]["whole_levels"])) if st else ""), flush=True)
    sy.remove()
dist.destroy_process_group()
