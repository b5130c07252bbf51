"""A TRAJECTORY of the oracle chain, not a single step (BASELINE.json metric: "... PSNR vs ref"): the same K Adam steps - learning
rates and epsilon of configs/bup20/best.yaml:103,108 / config_parser.py:672 through the parameter groups of pc_nerf/trainer.py:268-286 -
taken once by the HIP path and once by torch autograd over the CPU oracle (oracle.permuto_encode -> oracle.decoders.nef_forward ->
oracle.render.composite, the loss of trainer.py:443-467) from the same initial state on the ragged scene, and the render of the TRAINED
HIP parameters by both (PSNR of the HIP render against the oracle's render of the same parameters)."""
import numpy as np
import pytest
import torch

import test_gpu_parity as T
from test_gpu_train_step import ragged_scene, train_loss, hip_leaves

pytestmark = pytest.mark.gpu

LR, GRID_LR_WEIGHT, EPS = 1e-3, 100.0, 1e-15          # best.yaml:108,103 ; config_parser.py:672


class OracleChain:
    """The all-channel train step of test_gpu_train_step.oracle_step with persistent leaves: the lattice vertices and barycentric
    weights of the (fixed) samples are computed once by oracle.permuto_encode - the features are linear in the tables."""

    def __init__(self, nef, rays, occ, jitter, S, operand_round=None):
        from oracle import permuto_encode as op, render as orr
        self.o, self.d = rays.origins.cpu(), rays.dirs.cpu()
        self.N = self.o.shape[0]
        self.march = orr.raymarch_ray(self.o, self.d, rays.dist_min, rays.dist_max, S, jitter, occ, nef.grid.blas_level)
        xyz = self.march[2][:, 0].numpy()
        xyz = op.half_round(xyz) if nef.grid.half_coords else xyz
        self.leaves, self.enc = {}, {}
        for name, grid in (("grid.tables", nef.grid), ("delta_grid.tables", nef.delta_grid)):
            sf = grid.scale_factors(grid.resolutions).numpy()
            tab = grid.tables.detach().float().cpu().clone().requires_grad_(True)
            _, idx, bary = op.permuto_encode(xyz, tab.detach().numpy(), grid.random_shift_per_level.cpu().numpy(), sf)
            self.leaves[name] = tab
            self.enc[name] = (torch.from_numpy(idx.astype(np.int64)), torch.from_numpy(bary))
        self.params = {}
        for short in ("density", "color", "semantics", "inst"):
            W, b = getattr(nef, "decoder_" + short).weights()
            Wc = [w.detach().float().cpu().clone().requires_grad_(True) for w in W]
            bc = [v.detach().float().cpu().clone().requires_grad_(True) for v in b]
            self.params[short] = (Wc, bc)
            for i in range(len(Wc)):
                self.leaves["decoder_%s.W%d" % (short, i)] = Wc[i]
                self.leaves["decoder_%s.b%d" % (short, i)] = bc[i]
        self.lod_weights, self.operand_round = nef.lod_weights, operand_round

    def _feats(self, name):
        tab, (idx, bary) = self.leaves[name], self.enc[name]
        return torch.cat([(tab[l][idx[l]] * bary[l][..., None]).sum(1) for l in range(tab.shape[0])], -1)

    def render(self):
        from oracle import decoders as od, render as orr
        ridx, pidx, samples, depths, deltas, boundary = self.march
        out = od.nef_forward(self._feats("grid.tables"), self._feats("delta_grid.tables"), self.d[ridx], self.params,
                             {"rgb", "semantics", "inst_embedding"}, lod_weights=self.lod_weights, operand_round=self.operand_round)
        comp = orr.composite(self.N, ridx, boundary, out["density"], deltas, depths=depths, rgb=out["rgb"], bg_color="white")
        pan = orr.composite(self.N, ridx, boundary, out["density"].detach(), deltas, semantics=out["semantics"], inst=out["inst_embedding"],
                            bg_color="white")                    # tracer :148-155: weights from the detached optical thickness
        return comp["rgb"], pan["semantics"], pan["inst_embedding"]

    def load(self, hip):
        with torch.no_grad():
            for k, v in self.leaves.items():
                v.copy_(hip[k].detach().float().cpu())


def hip_vs_oracle_render(nef, tracer, rays, S, seed=5):
    """PSNR (dB) of the HIP render of `rays` against the oracle chain's render of the SAME parameters, samples and jitter, + both rgb
    buffers.  Used by the trajectory test below and by scripts/train_synthetic.py --oracle-psnr (the oracle stays under tests/)."""
    dev = rays.origins.device
    N = rays.origins.shape[0]
    jitter = torch.rand(N, S, generator=torch.Generator().manual_seed(seed))
    occ = None if nef.grid._all_occupied else nef.grid.occupancy_mask().reshape(3 * [2 ** nef.grid.blas_level])
    chain = OracleChain(nef, rays, occ, jitter, S)
    with torch.no_grad():
        rgb_o, sem_o, inst_o = chain.render()
        rb = tracer(nef, channels={"rgb", "depth", "semantics", "inst_embedding"}, rays=rays, jitter=jitter.to(dev), stage="val")
    return dict(psnr_hip_vs_oracle_db=round(_psnr(rb.rgb.float().cpu(), rgb_o), 2), rgb_hip=rb.rgb.float().cpu(), rgb_oracle=rgb_o,
                sem_max_abs_diff=float((rb.semantics.float().cpu() - sem_o).abs().max()),
                inst_max_abs_diff=float((rb.inst_embedding.float().cpu() - inst_o).abs().max()), samples=int(chain.march[0].shape[0]))


def _adam(leaves):
    grid = [v for k, v in leaves.items() if "grid" in k]
    rest = [v for k, v in leaves.items() if "grid" not in k]
    return torch.optim.Adam([dict(params=grid, lr=LR * GRID_LR_WEIGHT), dict(params=rest, lr=LR)], eps=EPS)


def _psnr(a, b):
    return float(-10.0 * torch.log10(torch.mean((a - b) ** 2) + 1e-20))


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_adam_trajectory_tracks_the_oracle_chain(gpu_device, precision):
    """30 Adam steps.  fp32 path: loss of every step within 1e-3 (relative) of the oracle's, final parameters within 1e-2 (relative L2 per
    leaf); bf16 path: loss within 3e-2, decoder parameters within 6e-2, tables within 0.15 (measured 0.10 - 0.11: Adam with eps = 1e-15 moves
    a table entry by ~lr = 0.1 per step whatever the size of its gradient, so wherever the gradient of an entry is of the order of the bf16
    rounding of the decoders its SIGN, and with it the whole step, differs).  Then the trained HIP parameters rendered by the HIP path and by the oracle: PSNR of one against
    the other > 60 dB (fp32) / > 38 dB (bf16: its decoders round features, weights and activations to 8 bits)."""
    dev = gpu_device
    N, S, K = 96, 32, 30
    nef, tracer, rays, occ, jitter = ragged_scene(dev, precision, N=N, S=S)
    gen = torch.Generator().manual_seed(9)
    gt, sem_gt, inst_gt = torch.rand(N, 3, generator=gen), torch.randint(0, 6, (N,), generator=gen), torch.randint(0, 200, (N,), generator=gen)
    chain = OracleChain(nef, rays, occ, jitter, S)
    opt_o = _adam(chain.leaves)
    opt_h = _adam(hip_leaves(nef))
    jit = jitter.to(dev)
    tg = (gt.to(dev), sem_gt.to(dev), inst_gt.to(dev))
    CH = {"rgb", "depth", "semantics", "inst_embedding"}
    lo, lh = [], []
    for it in range(K):
        opt_o.zero_grad(set_to_none=True)
        loss_o = train_loss(*chain.render(), gt, sem_gt, inst_gt)
        loss_o.backward()
        opt_o.step()
        lo.append(float(loss_o.detach()))
        opt_h.zero_grad(set_to_none=True)
        rb = tracer(nef, channels=CH, rays=rays, jitter=jit, stage="train")
        loss_h = train_loss(rb.rgb, rb.semantics.float(), rb.inst_embedding.float(), *tg)
        loss_h.backward()
        opt_h.step()
        lh.append(float(loss_h.detach()))
    assert lo[-1] < 0.9 * lo[0], lo                                   # the steps do train
    tol_loss, tol_par, tol_tab = (1e-3, 1e-2, 1e-2) if precision == "fp32" else (3e-2, 6e-2, 0.15)
    rel = [abs(a - b) / abs(b) for a, b in zip(lh, lo)]
    assert max(rel) < tol_loss, (max(rel), rel)
    worst = {k: round(T._rel_l2(v.detach().float().cpu(), chain.leaves[k].detach()), 5) for k, v in hip_leaves(nef).items()}
    bad = {k: e for k, e in worst.items() if not e < (tol_tab if "tables" in k else tol_par)}
    assert not bad, (bad, worst)
    # "PSNR vs ref" anchored on the oracle: the TRAINED HIP parameters rendered by both
    chain.load(hip_leaves(nef))
    with torch.no_grad():
        rgb_o, sem_o, inst_o = chain.render()
        rb = tracer(nef, channels=CH, rays=rays, jitter=jit, stage="val")
    psnr = _psnr(rb.rgb.float().cpu(), rgb_o)
    assert psnr > (60.0 if precision == "fp32" else 38.0), psnr
    assert float((rb.semantics.float().cpu() - sem_o).abs().max()) < (1e-4 if precision == "fp32" else 3e-2)
    assert float((rb.inst_embedding.float().cpu() - inst_o).abs().max()) < (1e-4 if precision == "fp32" else 3e-2)
