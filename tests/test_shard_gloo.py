"""N > 1 path on CPU: world-size-2 gloo processes exercise the shard arithmetic and the two collectives
(render all_gather, gradient all_reduce) of pagnerf_amd/shard.py."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pagnerf_amd import Rays, RenderBuffer
from pagnerf_amd import shard


def test_shard_bounds_cover_exactly():
    for n in (0, 1, 7, 8, 4096, 4097, 24576):
        for world in (1, 2, 3, 8):
            spans = [shard.shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def _fake_render(rays):
    """Deterministic per-ray 'render' so gathered results can be checked against a single-process run."""
    o, d = rays.origins, rays.dirs          # exact elementwise arithmetic only (SIMD tails may round transcendentals differently)
    rgb = o * 3 + d
    alpha = o[:, :1] * 0.5 + d[:, 1:2] * 0.25
    inst = (o[:, :1] + d[:, :1]) * torch.arange(5.0)
    return RenderBuffer(rgb=rgb, alpha=alpha, hit=alpha[:, 0] > 0.4, inst_embedding=inst)


def _worker(rank, world, port, n, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(0)
        rays = Rays(torch.rand(n, 3, generator=g), torch.rand(n, 3, generator=g), 0.0, 2.0)
        local = shard.shard_rays(rays)
        lo, hi = shard.shard_bounds(n, rank, world)
        assert local.origins.shape[0] == hi - lo and torch.equal(local.origins, rays.origins[lo:hi])
        full = shard.all_gather_render(_fake_render(local), n)
        ref = _fake_render(rays)
        for ch in ("rgb", "alpha", "hit", "inst_embedding"):
            assert torch.equal(getattr(full, ch), getattr(ref, ch)), ch          # bitwise: gather moves data only
        # labels only: the instance channel travels as its arg-max (one column), everything else unchanged
        lab = shard.all_gather_render(_fake_render(local), n, label_channels=("inst_embedding", "not_there"))
        assert lab.inst_embedding.dtype == torch.int64 and torch.equal(lab.inst_embedding, ref.inst_embedding.argmax(-1))
        assert torch.equal(lab.rgb, ref.rgb) and torch.equal(lab.hit, ref.hit)
        # gradient all-reduce: mean over ranks of rank-dependent grads
        p1, p2 = torch.nn.Parameter(torch.zeros(5, 3)), torch.nn.Parameter(torch.zeros(7))
        p1.grad = torch.full((5, 3), float(rank + 1))
        p2.grad = torch.arange(7.0) * (rank + 1)
        p3 = torch.nn.Parameter(torch.zeros(2))                                  # no grad: skipped
        shard.allreduce_grads([p1, p2, p3])
        mean = sum(range(1, world + 1)) / world
        assert torch.allclose(p1.grad, torch.full((5, 3), mean)) and torch.allclose(p2.grad, torch.arange(7.0) * mean)
        assert p3.grad is None
        # large gradients (tables) go in place, one message each; small ones flat - same result either way
        p1.grad = torch.full((5, 3), float(rank + 1))
        p2.grad = torch.arange(7.0) * (rank + 1)
        p2_storage = p2.grad.data_ptr()
        shard.allreduce_grads([p1, p2, p3], big=7)
        assert torch.allclose(p1.grad, torch.full((5, 3), mean)) and torch.allclose(p2.grad, torch.arange(7.0) * mean)
        assert p2.grad.data_ptr() == p2_storage
        # GradSync: the early parameter is exchanged from its post-accumulate hook while the backward continues
        a, b, c = (torch.nn.Parameter(torch.ones(4, 2)), torch.nn.Parameter(torch.ones(3)), torch.nn.Parameter(torch.ones(2)))
        sync = shard.GradSync([a, b, c], early=[a])
        for step in range(2):
            for prm in (a, b, c):
                prm.grad = None
            loss = (a * (rank + 1)).sum() * (step + 1) + (b * (rank + 2)).sum()          # c gets no gradient
            loss.backward()
            sync.finish()
            mean_a = sum(r + 1 for r in range(world)) / world * (step + 1)
            mean_b = sum(r + 2 for r in range(world)) / world
            assert torch.allclose(a.grad, torch.full((4, 2), mean_a)) and torch.allclose(b.grad, torch.full((3,), mean_b)), step
            assert c.grad is None
        sync.remove()
        # reduced-precision exchange (GradSync(comm_dtype=bf16)): direct all_to_all + fp32 accumulate + all_gather.  Error bound per
        # element: each rank's value rounded to bf16 (8 significant bits: 2^-8 relative) and the mean rounded once more:
        # |err| <= 2^-7 * mean_r |g_r| (+ fp32 noise)
        gen = torch.Generator().manual_seed(100 + rank)
        n_big = 3 * 1000 + 7                                   # not a multiple of the world size: the last chunk is padded
        gtab = torch.randn(n_big, 2, generator=gen) * (10.0 ** torch.randint(-3, 3, (n_big, 1), generator=gen).float())
        all_g = [torch.empty_like(gtab) for _ in range(world)]
        dist.all_gather(all_g, gtab)
        exact = torch.stack(all_g).double().mean(0)
        bound = torch.stack(all_g).abs().double().mean(0) * 2.0 ** -7 + 1e-12
        for early in (True, False):
            tab, small = torch.nn.Parameter(torch.zeros(n_big, 2)), torch.nn.Parameter(torch.zeros(5))
            sync = shard.GradSync([tab, small], early=[tab] if early else [], comm_dtype=torch.bfloat16, big=1000)
            ((tab * gtab).sum() + (small * (rank + 1.0)).sum()).backward()
            sync.finish()
            err = (tab.grad.double() - exact).abs()
            assert bool((err <= bound).all()), (early, float((err / bound).max()))
            assert tab.grad.dtype == torch.float32 and torch.allclose(small.grad, torch.full((5,), sum(range(1, world + 1)) / world))
            sync.remove()
        # comm_dtype="auto": the regime decides.  A "slow interconnect" (tiny bus bandwidth: the predicted fp32 exchange dwarfs the step) switches to the bf16
        # direct reduce after AUTO_WARM steps, a fast one keeps the exact fp32 all-reduce; both ranks switch at the same step; results stay right across it
        for bus, want in ((1e-6, "bf16"), (1e9, "fp32")):
            tab, small = torch.nn.Parameter(torch.zeros(n_big, 2)), torch.nn.Parameter(torch.zeros(5))
            sync = shard.GradSync([tab, small], comm_dtype="auto", big=1000, bus_gbs=bus)
            for step in range(shard.AUTO_WARM + 4):
                tab.grad = small.grad = None
                ((tab * gtab).sum() + (small * (rank + 1.0)).sum()).backward()
                sync.finish()
                err = (tab.grad.double() - exact).abs()
                assert bool((err <= bound).all()), (bus, step)
                if sync.auto_decision is None:
                    assert bool((err <= 1e-6 * torch.stack(all_g).abs().double().mean(0) + 1e-12).all())     # before the decision: fp32 all-reduce
            assert sync.auto_decision is not None and sync.auto_decision["comm_dtype"] == want, sync.auto_decision
            assert sync.auto_decision["exposed_bytes"] == n_big * 2 * 4 and (sync.comm_dtype is torch.bfloat16) == (want == "bf16")
            sync.remove()
        assert abs(shard.predicted_exchange_ms(50.33e6, 8) - 2 * 7 / 8 * 50.33e6 / 250e9 * 1e3) < 1e-9
        # every rank ends with the SAME reduced tensor (the all_gather distributes one rounded value per element)
        chk = [torch.empty_like(tab.grad) for _ in range(world)]
        dist.all_gather(chk, tab.grad.detach())
        assert all(torch.equal(chk[0], c) for c in chk[1:])
        q.put((rank, "ok"))
    except Exception as e:          # surface the failure in the parent
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def test_world_size_2_gloo_gather_and_allreduce():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    world, n = 2, 1001                # uneven split: 501 + 500
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(results) == [(0, "ok"), (1, "ok")], results


def _touched_grad(rank, step, L, T, F, fills, seed=7):
    """A table gradient as the encode backward leaves it: exact zeros outside the rows this rank's samples touched; level l touches about
    fills[l] of its rows (rank- and step-dependent sets that overlap partly)."""
    g = torch.Generator().manual_seed(seed + 1000 * rank + 17 * step)
    grad = torch.zeros(L, T, F)
    for l in range(L):
        rows = torch.nonzero(torch.rand(T, generator=g) < fills[l]).reshape(-1)
        grad[l, rows] = torch.randn(rows.numel(), F, generator=g) * 10.0 ** float(torch.randint(-3, 3, (1,), generator=g))
    return grad


def _sparse_worker(rank, world, port, q):
    import warnings
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        L, T, F = 6, 4096, 2
        fills = [0.0005, 0.01, 0.08, 0.2, 0.45, 0.9]               # coarse -> fine: the last two exceed SPARSE_DENSE_FILL as a union and travel whole

        def dense_reference(g):
            allg = [torch.empty_like(g) for _ in range(world)]
            dist.all_gather(allg, g)
            st = torch.stack(allg)
            return st, st.mean(0)

        for mode in ("exact", "bounded"):
            for comm in (None, torch.bfloat16):
                for early in (False, True):
                    tab, small = torch.nn.Parameter(torch.zeros(L, T, F)), torch.nn.Parameter(torch.zeros(5))
                    sync = shard.GradSync([tab, small], early=[tab] if early else [], comm_dtype=comm, big=1000, sparse=mode)
                    for step in range(4):
                        g = _touched_grad(rank, step, L, T, F, fills)
                        st, mean = dense_reference(g)
                        tab.grad = small.grad = None
                        ((tab * g).sum() + (small * (rank + 1.0)).sum()).backward()
                        sync.finish()
                        got = tab.grad
                        union = (st != 0).any(-1).any(0)                                   # [L, T]
                        assert bool((got[~union] == 0).all()), "rows no rank touched must stay exact zeros"
                        if comm is None:
                            assert torch.allclose(got, mean, rtol=1e-6, atol=1e-12), (mode, early, step)   # the dense all-reduce's values on the touched rows
                        else:
                            bound = st.abs().double().mean(0) * 2.0 ** -7 + 1e-12
                            assert bool(((got.double() - mean.double()).abs() <= bound).all()), (mode, early, step)
                        assert torch.allclose(small.grad, torch.full((5,), sum(range(1, world + 1)) / world))
                        stats = sync.sparse_stats()[0]
                        elem = 2 if comm is not None else 4
                        assert stats["dense_bytes"] == L * T * F * 4 and stats["bitmap_bytes"] == L * T // 8
                        if mode == "bounded" and step < shard.SPARSE_LAG:
                            assert stats["exchanged_bytes"] == L * T * F * elem and stats["whole_levels"] == L      # no counts taken in yet (fixed lag): whole
                        else:
                            # levels 0 - 3 as slots (level 3: a 36 % union x 1.25 head-room), 4 - 5 whole: 2 / 6 + 0.2 of the dense message
                            assert stats["whole_levels"] == 2 and stats["exchanged_bytes"] < 0.6 * L * T * F * elem, stats
                        # every rank holds the SAME reduced tensor
                        chk = [torch.empty_like(got) for _ in range(world)]
                        dist.all_gather(chk, got.detach())
                        assert all(torch.equal(chk[0], c) for c in chk[1:])
                    assert sync.sparse_stats()[0]["dropped_rows"] == 0
                    sync.remove()
        # a row count that is no multiple of 8 (the bit-packing pads the last byte) with slots smaller than a level (granule 64 instead of 1024)
        gran, shard.SPARSE_GRANULE = shard.SPARSE_GRANULE, 64
        try:
            Lo, To = 3, 1001
            for mode in ("exact", "bounded"):
                tab = torch.nn.Parameter(torch.zeros(Lo, To, F))
                sync = shard.GradSync([tab], comm_dtype=None, big=1000, sparse=mode)
                for step in range(4):
                    g = _touched_grad(rank, step, Lo, To, F, [0.02, 0.1, 0.9], seed=3)
                    st, mean = dense_reference(g)
                    tab.grad = None
                    (tab * g).sum().backward()
                    sync.finish()
                    assert torch.allclose(tab.grad, mean, rtol=1e-6, atol=1e-12), (mode, step)
                    assert bool((tab.grad[~(st != 0).any(-1).any(0)] == 0).all())
                stats = sync.sparse_stats()[0]
                assert stats["bitmap_bytes"] == Lo * 126 and stats["whole_levels"] == 1 and stats["exchanged_bytes"] < stats["dense_bytes"] and stats["dropped_rows"] == 0, stats
                sync.remove()
        finally:
            shard.SPARSE_GRANULE = gran
        # bounded mode when the regime changes under it: a level's union jumps past its slots.  The rows that do not fit are zero on EVERY rank alike
        # (replicas stay identical), counted, warned about once; the slots grow; reset_sparse() sends the next step whole
        tab = torch.nn.Parameter(torch.zeros(L, T, F))
        sync = shard.GradSync([tab], comm_dtype=None, big=1000, sparse=True)
        low, high = [0.001] * L, [0.001, 0.001, 0.3, 0.001, 0.001, 0.001]
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            for step, f in enumerate((low, low, low, high, high, high, high)):
                g = _touched_grad(rank, step, L, T, F, f)
                st, mean = dense_reference(g)
                tab.grad = None
                (tab * g).sum().backward()
                sync.finish()
                chk = [torch.empty_like(tab.grad) for _ in range(world)]
                dist.all_gather(chk, tab.grad.detach())
                assert all(torch.equal(chk[0], c) for c in chk[1:]), step
                wrong = (tab.grad - mean).abs() > 1e-6 * mean.abs() + 1e-12
                if step in (3, 4):     # slots sized from the 0.1 % steps (fixed lag of two): most of level 2's union is dropped - as zeros, nothing else is disturbed
                    assert bool(wrong.any()) and bool((tab.grad[wrong] == 0).all()) and not bool(wrong[[0, 1, 3, 4, 5]].any())
                else:                  # before the jump; and from step 5 on the counts of step 3 have been taken in: level 2 fits (or travels whole)
                    assert not bool(wrong.any()), step
        assert sync.sparse_stats()[0]["dropped_rows"] > 0 and any("did not fit" in str(w.message) for w in caught)
        sync.reset_sparse()
        g = _touched_grad(rank, 9, L, T, F, low)
        tab.grad = None
        (tab * g).sum().backward()
        sync.finish()
        assert sync.sparse_stats()[0]["whole_levels"] == L
        sync.remove()
        q.put((rank, "ok"))
    except Exception as e:
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_world_size_2_gloo_sparse_table_exchange():
    """the touched-rows exchange (shard.SparseRows behind GradSync(sparse=...)): values equal the dense all-reduce on the union of
    touched rows and exact zeros elsewhere, for the fp32 all-reduce and the bf16 direct reduce, from the post-accumulate hook and from finish(),
    in the exact (host reads the counts) and the bounded (slots from earlier steps) mode; the message shrinks; an overflow of the bounded
    slots zeroes the same rows on every rank and is reported."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    world = 2
    procs = [ctx.Process(target=_sparse_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(results) == [(0, "ok"), (1, "ok")], results
