"""The four passes of the touched-rows exchange (csrc/sparse.hip, ABI 14: pag_sparse_rows_mask / _plan / _pack / _unpack) against the tensor-op form of
pagnerf_amd.shard.SparseRows (the form the world-size-2 gloo tests pin against the dense all-reduce), pass by pass and bit for bit: row masks, per-word
prefixes and counts (incl. whole levels, a row count that is no multiple of 32, rows that do not fit their slots), the packed buffer, the rewritten table."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _reference(grad, union, caps):
    """torch form: -> (word prefix per row rank, counts [L+1], buf [total, F], rewritten grad)."""
    L, T, F = grad.shape
    caps_t = torch.tensor(caps, device=grad.device)
    offs = torch.tensor([sum(caps[:l]) for l in range(L)], device=grad.device)
    whole = (caps_t >= T)[:, None]
    member = union | whole
    pos = torch.cumsum(member, 1)
    valid = member & (pos <= caps_t[:, None])
    total = int(sum(caps))
    buf = torch.zeros(total, F, device=grad.device)
    slot = (offs[:, None] + pos - 1)[valid]
    buf[slot] = grad[valid]
    counts = torch.cat([union.sum(1), (union & ~valid & ~whole).sum()[None]])
    return counts, buf, valid, slot


@pytest.mark.parametrize("T,F", [(1 << 12, 2), (1001, 2), (4096 + 17, 4), (64, 1)])
def test_sparse_rows_passes_equal_the_tensor_op_form(gpu_device, T, F):
    from pagnerf_amd import ops, _lib as L_
    dev = gpu_device
    gen = torch.Generator(device=dev).manual_seed(T + F)
    L = 5
    fills = [0.0, 0.01, 0.2, 0.6, 1.0]
    keep = torch.rand(L, T, device=dev, generator=gen) < torch.tensor(fills, device=dev)[:, None]
    grad = torch.randn(L, T, F, device=dev, generator=gen) * keep[..., None]
    grad[2, : min(T, 40)] = 0.0                                    # a run of untouched rows at the start of a word
    grad[2, 7, F - 1] = -0.0                                       # a negative zero is a zero
    W = (T + 31) // 32
    st = L_.stream()
    bits = torch.full((L * W,), -1, device=dev, dtype=torch.int32)
    ops._call("pag_sparse_rows_mask", grad.data_ptr(), L, T, F, bits.data_ptr(), st)
    want_mask = (grad != 0).any(-1)
    got_mask = ((bits.reshape(L, W)[:, :, None] >> torch.arange(32, device=dev, dtype=torch.int32)) & 1).bool().reshape(L, -1)
    assert torch.equal(got_mask[:, :T], want_mask) and not bool(got_mask[:, T:].any())
    # the union another rank contributes to: OR in a second mask
    other = torch.rand(L, T, device=dev, generator=gen) < torch.tensor([0.001, 0.02, 0.1, 0.1, 0.0], device=dev)[:, None]
    union_b = want_mask | other
    pad = (-T) % 32
    ub = torch.cat([union_b, union_b.new_zeros(L, pad)], 1).reshape(L, W, 32).to(torch.int64)
    union_bits = (ub << torch.arange(32, device=dev)).sum(-1)
    union_bits = torch.where(union_bits >= 2 ** 31, union_bits - 2 ** 32, union_bits).to(torch.int32).reshape(-1).contiguous()
    # slots: level 0 nothing to send (1 slot), 1 fits with room, 2 does NOT fit (rows dropped), 3 and 4 whole
    n_union = union_b.sum(1).tolist()
    caps = [1, int(n_union[1]) + 5, max(1, int(n_union[2]) // 2), T, T + 3]
    caps_dev = torch.tensor(caps, device=dev, dtype=torch.int32)
    offs_dev = torch.tensor([sum(caps[:l]) for l in range(L)], device=dev, dtype=torch.int64)
    prefix = torch.full((L * W,), -7, device=dev, dtype=torch.int32)
    counts = torch.full((L + 1,), -1, device=dev, dtype=torch.int64)
    ops._call("pag_sparse_rows_plan", union_bits.data_ptr(), L, T, caps_dev.data_ptr(), prefix.data_ptr(), counts.data_ptr(), st)
    ref_counts, ref_buf, valid, slot = _reference(grad, union_b, caps)
    assert torch.equal(counts, ref_counts), (counts.tolist(), ref_counts.tolist())
    pw = prefix.reshape(L, W)
    for l in range(L):
        if caps[l] >= T:
            assert torch.equal(pw[l], torch.arange(W, device=dev, dtype=torch.int32) * 32)
        else:
            per_word = torch.cat([union_b[l], union_b.new_zeros(pad)]).reshape(W, 32).sum(1)
            assert torch.equal(pw[l].long(), torch.cumsum(per_word, 0) - per_word)
    total = int(sum(caps))
    buf = torch.zeros(total, F, device=dev)
    ops._call("pag_sparse_rows_pack", grad.data_ptr(), L, T, F, union_bits.data_ptr(), prefix.data_ptr(), caps_dev.data_ptr(), offs_dev.data_ptr(), buf.data_ptr(), st)
    assert torch.equal(buf, ref_buf)
    reduced = buf * 0.5 + 1.0                                      # stands for the collective's result (every slot changed, also the empty ones)
    out = torch.full_like(grad, float("nan"))
    ops._call("pag_sparse_rows_unpack", reduced.data_ptr(), L, T, F, union_bits.data_ptr(), prefix.data_ptr(), caps_dev.data_ptr(), offs_dev.data_ptr(), out.data_ptr(), st)
    want = torch.zeros_like(grad)
    want[valid] = reduced[slot]
    assert torch.equal(out, want)
    dropped = sum(max(0, int(n_union[l]) - caps[l]) for l in range(L) if caps[l] < T)
    assert int(counts[L]) == dropped and dropped >= int(n_union[2]) - caps[2] > 0
    assert bool((out[2][union_b[2] & ~valid[2]] == 0).all())       # the rows that did not fit come back as zeros
