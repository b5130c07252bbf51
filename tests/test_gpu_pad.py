"""pag_pack_offsets_pad / pag_pad_packed on the GPU against a host restatement: the pack table and its clamped copy, the filler samples
[M, capacity) and filler entries [M / k, capacity / k) - written by 16 workgroups that share the range - the direction copy, and that nothing
before M is touched.  Capacities far beyond M (every workgroup has fillers to write), k = 1 and k > 1, an overflowing batch (M > capacity:
no filler written, clamped table)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N,k,cap_mult,fused", [(4096, 1, 1.02, True), (4096, 2, 1.5, True), (257, 4, 3.0, True), (4096, 1, 0.5, True),
                                                 (1000, 2, 2.0, False), (3, 1, 40.0, True)])
def test_padding_launch_matches_host_restatement(gpu_device, N, k, cap_mult, fused):
    from pagnerf_amd import _lib as L
    dev = gpu_device
    g = torch.Generator().manual_seed(N + k)
    counts = (torch.randint(0, 40, (N,), generator=g) * k).int()
    M = int(counts.sum())
    cap = max(k, int(M * cap_mult) // k * k)
    big = max(cap, M) + 64 * k
    mark = 7.0
    samples = torch.full((big, 3), mark, device=dev)
    depths, deltas = torch.full((big,), mark, device=dev), torch.full((big,), mark, device=dev)
    boundary = torch.full((big,), 9, device=dev, dtype=torch.uint8)
    ridx_sample = torch.full((big,), -5, device=dev, dtype=torch.int32)
    ne = big // k
    ridx_entry = torch.full((ne,), -5, device=dev, dtype=torch.int32)
    ridx64 = torch.full((ne,), -5, device=dev, dtype=torch.int64)
    pidx = torch.full((ne,), -5, device=dev, dtype=torch.int32)
    pack_start = torch.full((N + 1,), -1, device=dev, dtype=torch.int64)
    pack_c = torch.full((N + 1,), -1, device=dev, dtype=torch.int64)
    dirs = torch.randn(N, 3, generator=g).to(dev)
    dirs_out = torch.zeros(N, 3, device=dev)
    cd = counts.to(dev)
    lib = L.load()
    st = L.stream()
    if fused:
        L.check(lib.pag_pack_offsets_pad(L.ptr(cd), N, L.ptr(pack_start), None, cap, k, L.ptr(samples), L.ptr(depths), L.ptr(deltas),
                                         L.ptr(ridx_sample) if k > 1 else None, L.ptr(ridx_entry), L.ptr(ridx64), L.ptr(pidx), L.ptr(boundary),
                                         L.ptr(pack_c), L.ptr(dirs), L.ptr(dirs_out), st), "pag_pack_offsets_pad")
    else:
        L.check(lib.pag_pack_offsets(L.ptr(cd), N, L.ptr(pack_start), None, st), "pag_pack_offsets")
        L.check(lib.pag_pad_packed(L.ptr(pack_start), N, cap, k, L.ptr(samples), L.ptr(depths), L.ptr(deltas), L.ptr(ridx_sample) if k > 1 else None,
                                   L.ptr(ridx_entry), L.ptr(ridx64), L.ptr(pidx), L.ptr(boundary), L.ptr(pack_c), st), "pag_pad_packed")
    torch.cuda.synchronize()
    ref = np.concatenate([[0], np.cumsum(counts.numpy().astype(np.int64))])
    assert np.array_equal(pack_start.cpu().numpy(), ref)
    assert np.array_equal(pack_c.cpu().numpy(), np.minimum(ref, cap))
    if fused:
        assert torch.equal(dirs_out, dirs)
    lo, hi = (M, cap) if M <= cap else (0, 0)          # an overflowing batch gets no filler
    s, d, dl, b, rs = samples.cpu(), depths.cpu(), deltas.cpu(), boundary.cpu(), ridx_sample.cpu()
    assert bool((s[lo:hi] == 0).all()) and bool((d[lo:hi] == 0).all()) and bool((dl[lo:hi] == 0).all()) and bool((b[lo:hi] == 0).all())
    for t, fill in ((s, mark), (d, mark), (dl, mark), (b, 9)):
        assert bool((t[:lo] == fill).all()) and bool((t[hi:] == fill).all())          # nothing outside [M, capacity) is written
    if k > 1:
        assert bool((rs[lo:hi] == N - 1).all()) and bool((rs[:lo] == -5).all()) and bool((rs[hi:] == -5).all())
    else:
        assert bool((rs == -5).all())
    elo, ehi = (M // k, cap // k) if M <= cap else (0, 0)
    re_, r64, pi = ridx_entry.cpu(), ridx64.cpu(), pidx.cpu()
    assert bool((re_[elo:ehi] == N - 1).all()) and bool((r64[elo:ehi] == N - 1).all()) and bool((pi[elo:ehi] == 0).all())
    for t in (re_, r64, pi):
        assert bool((t[:elo] == -5).all()) and bool((t[ehi:] == -5).all())
