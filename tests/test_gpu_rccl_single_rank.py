"""RCCL under pagnerf_amd.shard on ONE GPU: a one-rank "nccl" process group (backend "nccl" is RCCL on ROCm) with
shard.FORCE_COLLECTIVES, so that every collective the N > 1 path issues - ReduceOp.AVG all-reduce and its first-use probe,
all_to_all_single + all_gather_into_tensor of the bf16 direct reduce, the flat all-reduce of the small gradients, the render
all_gather, the early all-reduce from the post-accumulate hook, also behind the SPLIT backward graphs, the touched-rows exchange
(uint8 mask all_gather + one collective over the compacted slots) - runs through the real
library on device buffers.  What a one-rank group cannot show is inter-GPU transport (xGMI) and scaling; what it does show: the
library initialises on this image, every call is well-formed (dtypes, contiguity, sizes, stream use) and leaves the values a
one-rank mean must leave.  Runs in a child process (its own process group; the parent's GPU state is untouched)."""
import os
import socket
import subprocess
import sys

import pytest

from conftest import REPO

pytestmark = pytest.mark.gpu

CHILD = r'''
import os, sys
sys.path.insert(0, %(repo)r); sys.path.insert(0, os.path.join(%(repo)r, "tests"))
import torch, torch.distributed as dist
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
import pagnerf_amd
from pagnerf_amd import shard
shard.FORCE_COLLECTIVES = True
print("avg_supported", shard._avg_supported())                      # a real RCCL all-reduce with ReduceOp.AVG + the MIN agreement
# in-place fp32 all-reduce of a table-sized gradient + ONE flat all-reduce of the small ones
big = torch.nn.Parameter(torch.zeros(24, 4096, 2, device=dev)); small = [torch.nn.Parameter(torch.zeros(64, 48, device=dev)), torch.nn.Parameter(torch.zeros(7, device=dev))]
gen = torch.Generator(device=dev).manual_seed(0)
for p in [big] + small:
    p.grad = torch.randn(p.shape, device=dev, generator=gen)
want = [p.grad.clone() for p in [big] + small]
shard.allreduce_grads([big] + small, average=True, big=1 << 16)
torch.cuda.synchronize()
for p, w in zip([big] + small, want):
    assert torch.equal(p.grad, w), "one-rank mean must be the identity"
# the bf16 direct reduce: all_to_all_single -> fp32 sum -> all_gather_into_tensor
g = torch.randn((1 << 20) + 3, device=dev, generator=gen); ref = g.clone()
d = shard._DirectReduce(g, torch.bfloat16, average=True); d.finish(); torch.cuda.synchronize()
assert torch.equal(g, ref.bfloat16().float()), float((g - ref).abs().max())        # every value rounded once to the message dtype
shard.allreduce_grads([big], average=True, big=1 << 16, comm_dtype=torch.bfloat16)
# render all_gather
rb = pagnerf_amd.RenderBuffer(rgb=torch.rand(37, 3, device=dev), hit=torch.rand(37, device=dev) > 0.5, inst=torch.rand(37, 200, device=dev))
out = shard.all_gather_render(rb, 37)
assert torch.equal(out.rgb, rb.rgb) and torch.equal(out.hit, rb.hit) and torch.equal(out.inst, rb.inst)
# a train step with GradSync: the delta table's all-reduce starts from its post-accumulate hook BETWEEN the two backward graphs
import test_gpu_parity as T
from test_gpu_train_step import ragged_scene, train_loss, hip_leaves
from test_gpu_graphs import _targets, _step, CH
nef, tracer, rays, occ, jitter = ragged_scene(dev, "bf16", N=96, S=32)
jit, targets = jitter.to(dev), _targets(96, dev)
_, _, g_eager = _step(nef, tracer, rays, jit, targets)
gt = pagnerf_amd.PanopticPackedRFTracer(raymarch_type="ray", num_steps=32, bg_color="white", use_graphs=True)      # world "> 1" is not true here: force the split
gt.graph_split = True
sync = shard.GradSync(list(nef.parameters()), early=[nef.delta_grid.tables])
assert len(sync._hooks) == 1
for it in range(4):
    for p in nef.parameters():
        p.grad = None
    rb = gt(nef, channels=CH, rays=rays, jitter=jit, stage="train")
    train_loss(rb.rgb, rb.semantics.float(), rb.inst_embedding.float(), *targets).backward()
    launched = len(sync._handles)
    sync.finish()
    torch.cuda.synchronize()
    assert launched == 1, launched                                  # the early all-reduce was issued from the hook
for name, p in hip_leaves(nef).items():
    assert T._rel_l2(p.grad.float(), g_eager[name].float()) < 1e-5, name
st = next(iter(gt._graphs.states.values())); graphed = next(iter(st.buckets.values()))
assert len(graphed.groups) == 2 and gt._graphs.replays >= 3
sync.remove()
# comm_dtype="auto" through RCCL: the decision's MAX all-reduce on a device scalar, then the chosen exchange.  On ONE rank the predicted exchange is 0 ms
# (2 (W-1)/W bytes), so the decision is always the fp32 all-reduce - what runs here is the collective plumbing of the decision, not the switch
# (tests/test_shard_gloo.py covers both outcomes on two ranks; the bf16 exchange itself ran through RCCL above)
tab = torch.nn.Parameter(torch.zeros(1 << 17, device=dev))
sy = shard.GradSync([tab], comm_dtype="auto", big=1 << 16, bus_gbs=1e-6)
for it in range(shard.AUTO_WARM + 3):
    tab.grad = torch.randn(tab.shape, device=dev, generator=gen); ref = tab.grad.clone()
    sy.finish(); torch.cuda.synchronize()
    assert torch.equal(tab.grad, ref), it
assert sy.auto_decision["comm_dtype"] == "fp32" and sy.auto_decision["predicted_fp32_exchange_ms"] == 0.0 and sy.auto_decision["exposed_bytes"] == (1 << 17) * 4, sy.auto_decision
sy.remove()
# the touched-rows exchange (shard.SparseRows) through RCCL: the bit-packed all_gather of the row masks (uint8), the slot compaction, the ONE collective over the
# slots (fp32 all-reduce, then the bf16 direct reduce), the rewrite of every row - in both modes; on one rank the result must be the input (fp32) / the input
# rounded once to bf16 on the touched rows and exact zeros elsewhere, and a sparse step must move fewer bytes than the table
Lv, Tv, Fv = 24, 1 << 14, 2
fills = [min(1.0, 0.0004 * 1.9 ** l) for l in range(Lv)]
for mode in ("exact", "bounded"):
    for comm in (None, torch.bfloat16):
        tab = torch.nn.Parameter(torch.zeros(Lv, Tv, Fv, device=dev))
        sy = shard.GradSync([tab], comm_dtype=comm, big=1 << 16, sparse=mode)
        for it in range(4):
            keep = torch.rand(Lv, Tv, device=dev, generator=gen) < torch.tensor(fills, device=dev)[:, None]
            g0 = torch.randn(Lv, Tv, Fv, device=dev, generator=gen) * keep[..., None]
            tab.grad = g0.clone()
            sy.finish(); torch.cuda.synchronize()
            want = g0 if comm is None else g0.bfloat16().float()
            assert torch.equal(tab.grad, want), (mode, comm, it, float((tab.grad - want).abs().max()))
            stt = sy.sparse_stats()[0]
            if mode == "exact" or it >= shard.SPARSE_LAG:
                assert stt["exchanged_bytes"] < 0.75 * Lv * Tv * Fv * (4 if comm is None else 2) and 0 < stt["whole_levels"] < Lv, stt
        assert sy.sparse_stats()[0]["dropped_rows"] == 0
        sy.remove()
# the kernel passes (csrc/sparse.hip, the default on GPU gradients) and the tensor-op form give the same table, also when rows do not fit their slots
for kern in (True, False):
    shard.SPARSE_KERNELS = kern
    tab = torch.nn.Parameter(torch.zeros(Lv, Tv, Fv, device=dev))
    sy = shard.GradSync([tab], big=1 << 16, sparse="bounded")
    g2 = torch.Generator(device=dev).manual_seed(77)
    outs = []
    for it, f in enumerate(([0.002] * Lv, [0.002] * Lv, [0.002] * 10 + [0.2] + [0.002] * (Lv - 11), [0.002] * 10 + [0.2] + [0.002] * (Lv - 11))):
        keep = torch.rand(Lv, Tv, device=dev, generator=g2) < torch.tensor(f, device=dev)[:, None]
        tab.grad = torch.randn(Lv, Tv, Fv, device=dev, generator=g2) * keep[..., None]
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            sy.finish()
        torch.cuda.synchronize()
        outs.append(tab.grad.clone())
    if kern:
        res_k = outs
    else:
        res_t = outs
    sy.remove()
shard.SPARSE_KERNELS = True
assert all(torch.equal(a, b) for a, b in zip(res_k, res_t)) and float((res_k[2] == 0).float().mean()) > 0.9
dist.barrier()
dist.destroy_process_group()
print("RCCL_SINGLE_RANK_OK")
'''


def test_shard_collectives_run_through_rccl_on_a_one_rank_group():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, "-c", CHILD % dict(repo=REPO)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RCCL_SINGLE_RANK_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
