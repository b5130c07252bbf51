#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by running the REFERENCE's own Python
modules (from /root/reference, read-only) on CPU.

Run only in the build container (the reference does not travel to the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

The reference imports four third-party packages that are not installed (kaolin, wisp,
permutohedral_encoding, tinycudann).  They are replaced here by minimal stand-in modules
written for this build (restating the public semantics recorded in SURVEY.md Appendix A);
the reference files themselves are imported unmodified and nothing of them is copied.

Fixtures are DATA: inputs, seeds and the reference's outputs.
  g1_hash.npz       grids/hash_grid_torch.py  HashGridTorch.interpolate  (indices bit-exact, feats fp32)
  g2_resolutions.npz  HashEmbedder level resolutions (incl. the 2047 fp32 quirk)
  g3_nef.npz        pc_nerf/panoptic_delta_nef.py  PanopticDeltaNeF.rgb_semantics with HashGridTorch grids
  g4_tracer.npz     tracers/panoptic_packed_rf_tracer.py  trace() on a packed random scene, both bg colours
  g5_linassign.npz  loss/lin_assignment.py + loss/lin_assignment_things.py virtual labels (bit-exact)
  g6_reg.npz        loss/regularizers.py sigma_sparsity_loss, segment_consistency_regularizer (value + autograd gradient)
"""
import os
import sys
import types
import inspect

sys.dont_write_bytecode = True
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
import logging as log

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("PAGNERF_REFERENCE", "/root/reference")
sys.path.insert(0, REPO)
sys.path.insert(0, REF)

from oracle import render as orender  # restated kaolin semantics used as the stand-in ops


def table_from_seed(seed, shape, kind):
    """Deterministic tables shared by this generator and the tests."""
    rs = np.random.RandomState(seed)
    if kind == "uniform1e-4":
        return rs.uniform(-1e-4, 1e-4, size=shape).astype(np.float32)
    return rs.standard_normal(size=shape).astype(np.float32)


# ------------------------------------------------------------------ stand-in third-party modules
def _mod(name):
    m = types.ModuleType(name)
    sys.modules[name] = m
    return m


class PerfTimer:
    def __init__(self, *a, **k):
        pass

    def check(self, *a, **k):
        pass


class _BLAS:
    def __init__(self):
        self.octree = torch.zeros(1, dtype=torch.uint8)
        self.points = torch.zeros(1, 3, dtype=torch.int16)
        self.prefix = torch.zeros(1, dtype=torch.int32)
        self.pyramid = torch.zeros(2, 2, dtype=torch.int32)


class HashGrid(nn.Module):
    def __init__(self, feature_dim, interpolation_type='linear', multiscale_type='cat', feature_std=0.0,
                 feature_bias=0.0, codebook_bitwidth=16, blas_level=7, **kwargs):
        super().__init__()
        self.feature_dim = feature_dim
        self.interpolation_type = interpolation_type
        self.multiscale_type = multiscale_type
        self.codebook_bitwidth = codebook_bitwidth
        self.blas_level = blas_level
        self.blas = _BLAS()


class BasicDecoder(nn.Module):
    def __init__(self, input_dim, output_dim, activation, bias, layer=nn.Linear, num_layers=1, hidden_dim=128, skip=[]):
        super().__init__()
        self.activation = activation
        layers = []
        for i in range(num_layers):
            layers.append(layer(input_dim if i == 0 else hidden_dim, hidden_dim, bias=bias))
        self.layers = nn.ModuleList(layers)
        self.lout = layer(hidden_dim if num_layers > 0 else input_dim, output_dim, bias=bias)

    def forward(self, x):
        h = x
        for l in self.layers:
            h = self.activation(l(h))
        return self.lout(h)


class PositionalEmbedder(nn.Module):
    def __init__(self, num_freq, max_freq_log2):
        super().__init__()
        self.bands = nn.Parameter(2.0 ** torch.linspace(0.0, max_freq_log2, num_freq), requires_grad=False)

    def forward(self, x):
        n = x.shape[0]
        w = (x[:, None, :] * self.bands[None, :, None]).reshape(n, -1)
        return torch.cat([x, torch.sin(w), torch.cos(w)], dim=-1)


def get_positional_embedder(frequencies, active, input_dim=3):
    if not active:
        return nn.Identity(), input_dim
    return PositionalEmbedder(frequencies, frequencies - 1), input_dim + 2 * input_dim * frequencies


class BaseNeuralField(nn.Module):
    def __init__(self, grid_type='OctreeGrid', interpolation_type='linear', multiscale_type='none', as_type='octree',
                 raymarch_type='voxel', decoder_type='none', embedder_type='none', activation_type='relu',
                 layer_type='none', base_lod=2, num_lods=1, sample_tex=False, dilate=None, feature_dim=16,
                 hidden_dim=128, pos_multires=10, view_multires=4, num_layers=1, position_input=False, **kwargs):
        super().__init__()
        for k, v in list(locals().items()):
            if k not in ('self', 'kwargs', '__class__'):
                setattr(self, k, v)
        self.kwargs = kwargs
        self.grid = None
        self._fns = []
        self.init_grid()
        self.init_embedder()
        self.init_decoder()
        self.register_forward_functions()

    def _register_forward_function(self, fn, channels):
        self._fns.append((fn, set(channels)))

    def get_supported_channels(self):
        s = set()
        for _, c in self._fns:
            s |= c
        return s

    def forward(self, channels=None, **kwargs):
        if isinstance(channels, str):
            req = {channels}
        else:
            req = set(channels)
        out = {}
        for fn, chans in self._fns:
            if not (chans & req):
                continue
            sig = inspect.signature(fn)
            args = {k: kwargs[k] for k in sig.parameters if k in kwargs}
            res = fn(**args)
            for c in chans & req:
                out[c] = res[c]
        if isinstance(channels, str):
            return out[channels]
        if isinstance(channels, list):
            return [out[c] for c in channels]
        return out


class RenderBuffer:
    def __init__(self, **kw):
        self.__dict__.update(kw)


class PackedRFTracer(nn.Module):
    def __init__(self, raymarch_type='voxel', num_steps=64, step_size=1.0, bg_color='white', **kwargs):
        super().__init__()
        self.raymarch_type = raymarch_type
        self.num_steps = num_steps
        self.step_size = step_size
        self.bg_color = bg_color


def install_stubs():
    wisp = _mod("wisp")
    grids = _mod("wisp.models.grids")
    for n, v in dict(HashGrid=HashGrid, OctreeGrid=HashGrid, CodebookOctreeGrid=HashGrid, TriplanarGrid=HashGrid,
                     BLASGrid=HashGrid, BasicDecoder=BasicDecoder, PerfTimer=PerfTimer, np=np, F=F, log=log,
                     torch=torch, nn=nn, spc_ops=types.SimpleNamespace()).items():
        setattr(grids, n, v)
    _mod("wisp.models")
    _mod("wisp.models.nefs").BaseNeuralField = BaseNeuralField
    _mod("wisp.models.activations").get_activation_class = lambda name: {'relu': torch.relu, 'sin': torch.sin,
                                                                        'none': (lambda x: x)}[name]
    _mod("wisp.models.layers").get_layer_class = lambda name: nn.Linear
    _mod("wisp.models.embedders").get_positional_embedder = get_positional_embedder
    _mod("wisp.ops")
    _mod("wisp.ops.geometric").sample_unif_sphere = lambda n: np.tile(np.array([[0., 0., 1.]]), (n, 1))
    core = _mod("wisp.core")
    core.RenderBuffer = RenderBuffer
    core.Rays = object
    utils = _mod("wisp.utils")
    utils.PerfTimer = PerfTimer
    utils.PsDebugger = object
    _mod("wisp.tracers").PackedRFTracer = PackedRFTracer
    _mod("tinycudann")
    _mod("permutohedral_encoding").PermutoEncoding = object
    _mod("kaolin")
    _mod("kaolin.render")
    spc = _mod("kaolin.render.spc")
    spc.mark_pack_boundaries = orender.mark_pack_boundaries
    spc.sum_reduce = orender.sum_reduce
    spc.exponential_integration = lambda feats, tau, boundary, exclusive=True: (
        None, orender.exponential_integration_weights(tau, boundary))
    sys.modules["kaolin.render"].spc = spc
    _mod("kaolin.render.camera").Camera = object


class _CudaToCpu:
    """While active, torch.tensor(..., device='cuda') and Tensor.to('cuda') stay on CPU
    (grids/hash_grid_torch.py:10-11, loss/lin_assignment_things.py:20)."""

    def __enter__(self):
        self._tensor, self._to = torch.tensor, torch.Tensor.to

        def tensor(*a, **k):
            if k.get("device") == "cuda":
                k["device"] = "cpu"
            return self._tensor(*a, **k)

        def to(t, *a, **k):
            a = tuple("cpu" if (isinstance(x, str) and x == "cuda") else x for x in a)
            return self._to(t, *a, **k)

        torch.tensor, torch.Tensor.to = tensor, to

    def __exit__(self, *e):
        torch.tensor, torch.Tensor.to = self._tensor, self._to


# ------------------------------------------------------------------ fixtures
def sample_points(rs, n):
    x = rs.uniform(-1, 1, size=(n, 3)).astype(np.float32)
    edge = np.array([[1, 1, 1], [-1, -1, -1], [0, 0, 0], [1, -1, 0.5], [0.125, -0.25, 0.5], [-1, 0.999999, 1],
                     [1.25, -1.5, 0.3], [0.99999994, 0.5, -0.99999994]], dtype=np.float32)
    return np.concatenate([x, edge], 0)


def g1_g2(hgt):
    rs = np.random.RandomState(11)
    out = {}
    for tag, log2T, res0, res1, L, kind in (("a", 19, 16, 2048, 16, "uniform1e-4"), ("b", 12, 16, 512, 8, "normal")):
        grid = hgt.HashGridTorch(2, codebook_bitwidth=log2T)
        with torch.no_grad():
            grid.init_from_resolutions([res0] * (L - 1) + [res1])
            tab = table_from_seed(100 + log2T, (L, 2 ** log2T, 2), kind)
            for i in range(L):
                grid.embedder.embeddings[i].weight.copy_(torch.from_numpy(tab[i]))
            x = torch.from_numpy(sample_points(rs, 600))
            feats = grid.interpolate(x[None], L - 1)
            res = [float(torch.floor(grid.embedder.base_resolution * grid.embedder.b ** i)) for i in range(L)]
            idx = np.stack([hgt.get_voxel_vertices(x, torch.tensor(r), log2T)[2].numpy() for r in res])
        out.update({f"{tag}_x": x.numpy(), f"{tag}_feats": feats.numpy(), f"{tag}_idx": idx.astype(np.int32),
                    f"{tag}_res": np.array(res, np.float32), f"{tag}_log2T": log2T, f"{tag}_seed": 100 + log2T,
                    f"{tag}_kind": kind})
    np.savez_compressed(os.path.join(HERE, "g1_hash.npz"), **out)
    res = {}
    for (a, b, L) in ((16, 2048, 16), (16, 512, 16), (16, 2048, 14), (16, 1024, 16)):
        e = hgt.HashEmbedder(n_levels=L, log2_hashmap_size=4, base_resolution=a, finest_resolution=b)
        res[f"r_{a}_{b}_{L}"] = np.array([float(torch.floor(e.base_resolution * e.b ** i)) for i in range(L)], np.float32)
    np.savez_compressed(os.path.join(HERE, "g2_resolutions.npz"), **res)


def g3(delta_nef_mod):
    torch.manual_seed(3)
    L, log2T, C, I = 8, 12, 6, 200
    nef = delta_nef_mod.PanopticDeltaNeF(
        grid_type='HashGridTorch', interpolation_type='linear', multiscale_type='cat', feature_dim=2, num_lods=L,
        base_lod=2, hidden_dim=64, num_layers=1, activation_type='relu', layer_type='none', embedder_type='positional',
        view_multires=4, pos_multires=4, position_input=False, num_classes=C, num_instances=I, sem_num_layers=1,
        sem_hidden_dim=64, sem_softmax=True, inst_num_layers=2, inst_hidden_dim=64, inst_softmax=True,
        panoptic_features_type='delta', codebook_bitwidth=log2T, delta_capacity_log_2=log2T)
    res = [16] * (L - 1) + [256]
    save = {}
    with torch.no_grad():
        for gi, g in enumerate((nef.grid, nef.delta_grid)):
            g.init_from_resolutions(res)
            tab = table_from_seed(300 + gi, (L, 2 ** log2T, 2), "normal") * np.float32(0.5)
            for i in range(L):
                g.embedder.embeddings[i].weight.copy_(torch.from_numpy(tab[i]))
        rs = np.random.RandomState(5)
        M = 256
        coords = torch.from_numpy(rs.uniform(-1, 1, size=(M, 1, 3)).astype(np.float32))
        d = rs.standard_normal(size=(M, 3)).astype(np.float32)
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        ray_d = torch.from_numpy(d)
        out = nef(coords=coords, ray_d=ray_d, pidx=None, lod_idx=None,
                  channels={'density', 'rgb', 'semantics', 'inst_embedding'})
        feats = nef.grid.interpolate(coords, L - 1)
        dfeats = nef.delta_grid.interpolate(coords, L - 1)
        dens_only = nef(coords=coords, ray_d=ray_d, channels="density")
    for name in ("decoder_density", "decoder_color", "decoder_semantics", "decoder_inst"):
        dec = getattr(nef, name)
        lins = list(dec.layers) + [dec.lout]
        for li, lin in enumerate(lins):
            save[f"{name}_w{li}"] = lin.weight.detach().numpy()
            save[f"{name}_b{li}"] = lin.bias.detach().numpy()
        save[f"{name}_n"] = len(lins)
    save.update(coords=coords.numpy(), ray_d=ray_d.numpy(), res=np.array(res, np.float32), log2T=log2T, L=L,
                seed_main=300, seed_delta=301, feats=feats.numpy(), delta_feats=dfeats.numpy(),
                density=out['density'].numpy(), rgb=out['rgb'].numpy(), semantics=out['semantics'].numpy(),
                inst_embedding=out['inst_embedding'].numpy(), density_only=dens_only.numpy(),
                view_embed_dim=nef.view_embed_dim, bias0=float(nef.decoder_density.lout.bias[0]))
    np.savez_compressed(os.path.join(HERE, "g3_nef.npz"), **save)


def g4(tracer_mod):
    rs = np.random.RandomState(7)
    N, S, C, I = 48, 24, 6, 20
    origins = torch.from_numpy(rs.uniform(-0.3, 0.3, size=(N, 3)).astype(np.float32))
    d = rs.standard_normal(size=(N, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    dirs = torch.from_numpy(d)
    jitter = torch.from_numpy(rs.uniform(0, 1, size=(N, S)).astype(np.float32))
    occ = torch.from_numpy(rs.uniform(size=(8, 8, 8)) > 0.3)
    occ_dirs = dirs.clone()
    ridx, pidx, samples, depths, deltas, boundary = orender.raymarch_ray(origins, dirs, 0.0, 2.0, S, jitter, occ, 3)
    # make ray 5 empty
    keep = ridx != 5
    ridx, pidx, samples, depths, deltas = ridx[keep], pidx[keep], samples[keep], depths[keep], deltas[keep]
    boundary = orender.mark_pack_boundaries(ridx)
    M = ridx.shape[0]
    density = torch.from_numpy((rs.gamma(1.0, 8.0, size=(M, 1, 1)) * (rs.uniform(size=(M, 1, 1)) > 0.3)).astype(np.float32))
    rgb = torch.from_numpy(rs.uniform(size=(M, 1, 3)).astype(np.float32))
    sem = torch.softmax(torch.from_numpy(rs.standard_normal(size=(M, C)).astype(np.float32)), -1)
    inst = torch.softmax(torch.from_numpy(rs.standard_normal(size=(M, I)).astype(np.float32)), -1)

    class Grid:
        num_lods = 4
        active_lods = [0, 1, 2, 3]

        def raymarch(self, rays, level, num_samples, raymarch_type):
            return ridx, pidx, samples, depths, deltas, boundary

    class Nef:
        grid = Grid()
        device = 'cpu'

        def __call__(self, coords, ray_d, pidx, lod_idx, channels):
            full = {'density': density, 'rgb': rgb, 'semantics': sem, 'inst_embedding': inst}
            if isinstance(channels, str):
                return full[channels]
            return {c: full[c] for c in channels}

    rays = types.SimpleNamespace(origins=origins, dirs=dirs)
    save = dict(N=N, S=S, origins=origins.numpy(), dirs=dirs.numpy(), jitter=jitter.numpy(), occ=occ.numpy(),
                ridx=ridx.numpy(), pidx=pidx.numpy(), samples=samples.numpy(), depths=depths.numpy(),
                deltas=deltas.numpy(), boundary=boundary.numpy(), density=density.numpy(), rgb=rgb.numpy(),
                semantics=sem.numpy(), inst_embedding=inst.numpy(), empty_ray=5)
    for bg in ("white", "black"):
        tr = tracer_mod.PanopticPackedRFTracer(ray_sparcity_reg=0.01, ray_max_travel=6.0, raymarch_type='ray',
                                               num_steps=S, bg_color=bg)
        rb = tr.trace(Nef(), {'rgb', 'depth', 'semantics', 'inst_embedding'}, set(), rays, lod_idx=None,
                      raymarch_type='ray', num_steps=S, bg_color=bg, stage='train')
        for ch in ('rgb', 'alpha', 'hit', 'depth', 'semantics', 'inst_embedding', 'ray_sparcity_loss'):
            save[f"{bg}_{ch}"] = getattr(rb, ch).numpy()
    # voxel-mode travel filter case (tracer :88-108): k=2 samples per nugget
    Mv = 200
    vr = np.sort(rs.randint(0, 16, size=Mv))
    vdepth = np.sort(rs.uniform(0, 2, size=(Mv, 2, 1)).astype(np.float32), axis=1)
    order = np.lexsort((vdepth[:, 0, 0], vr))
    vr, vdepth = vr[order], vdepth[order]
    v_ridx = torch.from_numpy(vr).long()
    v_depths = torch.from_numpy(vdepth)
    v_samples = torch.from_numpy(rs.uniform(-1, 1, size=(Mv, 2, 3)).astype(np.float32))
    v_deltas = torch.from_numpy(rs.uniform(0.001, 0.01, size=(Mv * 2, 1)).astype(np.float32))
    v_boundary = orender.mark_pack_boundaries(v_ridx).repeat_interleave(2)
    v_boundary[1::2] = False
    v_density = torch.from_numpy(rs.gamma(1.0, 30.0, size=(Mv, 2, 1)).astype(np.float32))
    v_rgb = torch.from_numpy(rs.uniform(size=(Mv, 2, 3)).astype(np.float32))
    captured = {}

    class VGrid(Grid):
        def raymarch(self, rays, level, num_samples, raymarch_type):
            return v_ridx, v_ridx.clone(), v_samples, v_depths, v_deltas, v_boundary

    class VNef(Nef):
        grid = VGrid()

        def __call__(self, coords, ray_d, pidx, lod_idx, channels):
            captured['coords'] = coords
            m = captured['mask']
            full = {'density': v_density[m], 'rgb': v_rgb[m]}
            return {c: full[c] for c in channels}

    captured['mask'] = orender.voxel_travel_filter(v_ridx, v_depths, 0.35)
    tr = tracer_mod.PanopticPackedRFTracer(ray_sparcity_reg=0.0, ray_max_travel=0.35, raymarch_type='voxel', num_steps=2)
    rays16 = types.SimpleNamespace(origins=torch.zeros(16, 3), dirs=torch.ones(16, 3))
    rb = tr.trace(VNef(), {'rgb', 'depth'}, set(), rays16, raymarch_type='voxel', num_steps=2, bg_color='white')
    save.update(v_ridx=vr, v_depths=vdepth, v_deltas=v_deltas.numpy(), v_boundary=v_boundary.numpy(),
                v_density=v_density.numpy(), v_rgb=v_rgb.numpy(), v_max_travel=0.35,
                v_kept=captured['coords'].shape[0], v_out_rgb=rb.rgb.numpy(), v_out_alpha=rb.alpha.numpy(),
                v_out_depth=rb.depth.numpy())
    np.savez_compressed(os.path.join(HERE, "g4_tracer.npz"), **save)


def g5(la, lat):
    rs = np.random.RandomState(9)
    B, P, I = 2, 512, 200
    logits = rs.standard_normal(size=(B, P, I)).astype(np.float32) * 2
    gt = rs.choice([0, 0, 3, 7, 8, 15, 21, 40], size=(B, P)).astype(np.int64)
    logits[np.arange(B)[:, None], np.arange(P)[None], (gt * 3 + 1) % I] += 3.0   # some structure
    prob = torch.softmax(torch.from_numpy(logits), -1)
    stuff = torch.from_numpy(rs.uniform(size=(B, P)) > 0.5)
    pts = torch.from_numpy(rs.uniform(-1, 1, size=(B, P, 3)).astype(np.float32))
    gt_t = torch.from_numpy(gt)
    plain = la.LinAssignmentLoss()
    v_plain = np.stack([plain.create_virtual_gt_with_linear_assignment(gt_t[b], torch.from_numpy(logits[b])).numpy()
                        for b in range(B)])
    loss_plain = plain(prob, gt_t).numpy()
    save = dict(logits=logits, prob=prob.numpy(), gt=gt, stuff=stuff.numpy(), points_3d=pts.numpy(),
                virt_plain=v_plain, loss_plain=loss_plain)
    for tag, rej in (("things", False), ("things_rej", True)):
        with _CudaToCpu():
            lo = lat.LinAssignmentThingsLoss(outlier_rejection=rej)
        virt = []
        for b in range(B):
            vm = torch.logical_or(stuff[b], gt_t[b] > 0)
            virt.append(lo.create_virtual_gt_with_linear_assignment(prob[b][vm], gt_t[b][vm],
                                                                    pts[b][vm] if rej else None).numpy())
        save[f"virt_{tag}_0"], save[f"virt_{tag}_1"] = virt
        save[f"loss_{tag}"] = lo(prob, gt_t, stuff, pts if rej else None).numpy()
    np.savez_compressed(os.path.join(HERE, "g5_linassign.npz"), **save)


def g6(reg):
    rs = np.random.RandomState(13)
    s = rs.gamma(1.0, 5.0, size=(512,)).astype(np.float32)
    save = dict(sigma=s, sparsity=reg.sigma_sparsity_loss(torch.from_numpy(s)).numpy())
    # segment_consistency_regularizer (loss/regularizers.py:5-35) as pc_nerf/trainer.py:525-527 calls it: softmaxed instance probabilities
    # + 1e-27, [B, P, I], and the per-ray ground-truth ids [B, P].  The batch exercises every branch: id 0 is a segment of its own, a
    # segment whose rays all predict column 0 (skipped, :24-25), a segment where more than twice as many rays predict 0 as the best other
    # column (label forced to 0, :29-30), segments of one ray, images with different segment counts (the running total is divided by each
    # image's count in turn, :33), ids that are large and not contiguous.
    B, P, I = 3, 600, 12
    logits = (rs.standard_normal(size=(B, P, I)) * 2.0).astype(np.float32)
    labels = np.zeros((B, P), dtype=np.int64)
    labels[0] = rs.choice([0, 3, 4, 17, 1005], size=P, p=[0.4, 0.2, 0.2, 0.15, 0.05])
    labels[1] = rs.choice([0, 2, 9], size=P, p=[0.5, 0.3, 0.2])
    labels[2] = rs.choice([5, 6, 7, 8, 40, 41, 1000000007], size=P)
    labels[2, 0] = 99                                                  # a one-ray segment
    for b, lab, col, boost in ((0, 3, 5, 6.0), (0, 4, 0, 9.0), (1, 2, 7, 5.0), (2, 6, 2, 4.0)):
        logits[b, labels[b] == lab, col] += boost                      # (0, 4): every ray of segment 4 predicts column 0 -> skipped
    m = np.nonzero(labels[1] == 9)[0]                                  # segment 9 of image 1: ~75 % of the rays predict 0, the rest column 3
    logits[1, m[: (3 * len(m)) // 4], 0] += 9.0
    logits[1, m[(3 * len(m)) // 4:], 3] += 9.0
    prob = torch.softmax(torch.from_numpy(logits), -1)
    x = (prob + 1e-27).clone().requires_grad_(True)
    val = reg.segment_consistency_regularizer(x, torch.from_numpy(labels))
    val.backward()
    save.update(seg_prob=prob.numpy(), seg_labels=labels, seg_reg=val.detach().numpy(), seg_reg_grad=x.grad.numpy())
    np.savez_compressed(os.path.join(HERE, "g6_reg.npz"), **save)


def g7(hgt):
    """Autograd through the reference's HashGridTorch: d loss / d coords (what pose optimisation back-propagates,
    pc_nerf/ba_pipeline.py:85-92) and d loss / d tables, for a seeded upstream gradient."""
    rs = np.random.RandomState(17)
    log2T, L = 12, 8
    grid = hgt.HashGridTorch(2, codebook_bitwidth=log2T)
    with torch.no_grad():
        grid.init_from_resolutions([16] * (L - 1) + [512])
        tab = table_from_seed(100 + log2T, (L, 2 ** log2T, 2), "normal")
        for i in range(L):
            grid.embedder.embeddings[i].weight.copy_(torch.from_numpy(tab[i]))
    x = torch.from_numpy(sample_points(rs, 500)).requires_grad_(True)
    go = torch.from_numpy(rs.standard_normal(size=(x.shape[0], L * 2)).astype(np.float32))
    feats = grid.interpolate(x[None], L - 1)
    feats.backward(go)
    dtab = np.stack([grid.embedder.embeddings[i].weight.grad.numpy() for i in range(L)])
    res = [float(torch.floor(grid.embedder.base_resolution * grid.embedder.b ** i)) for i in range(L)]
    np.savez_compressed(os.path.join(HERE, "g7_hash_grad.npz"), x=x.detach().numpy(), go=go.numpy(), dx=x.grad.numpy(),
                        dtables=dtab, feats=feats.detach().numpy(), res=np.array(res, np.float32), log2T=log2T, seed=100 + log2T,
                        kind="normal")


def g8(dd_nef_mod, dd_tracer_mod):
    """Delta-density variant: pc_nerf/panoptic_dd_nef.py::PanopticDDensityNeF.rgb_semantics on HashGridTorch
    grids with fixed weights, and tracers/panoptic_dd_packed_rf_tracer.py::PanopticDDensityPackedRFTracer.trace() on a fixed
    packed scene (panoptic channels composited with the panoptic density's own weights)."""
    torch.manual_seed(8)
    L, log2T, C, I = 8, 12, 6, 20
    nef = dd_nef_mod.PanopticDDensityNeF(
        grid_type='HashGridTorch', interpolation_type='linear', multiscale_type='cat', feature_dim=2, num_lods=L,
        base_lod=2, hidden_dim=64, num_layers=1, activation_type='relu', layer_type='none', embedder_type='positional',
        view_multires=4, pos_multires=4, position_input=False, num_classes=C, num_instances=I, sem_num_layers=1,
        sem_hidden_dim=64, sem_softmax=True, inst_num_layers=2, inst_hidden_dim=64, inst_softmax=True,
        delta_num_layers=1, delta_hidden_dim=64, codebook_bitwidth=log2T, delta_capacity_log_2=log2T)
    res = [16] * (L - 1) + [256]
    save = {}
    with torch.no_grad():
        for gi, g in enumerate((nef.grid, nef.delta_grid)):
            g.init_from_resolutions(res)
            tab = table_from_seed(800 + gi, (L, 2 ** log2T, 2), "normal") * np.float32(0.5)
            for i in range(L):
                g.embedder.embeddings[i].weight.copy_(torch.from_numpy(tab[i]))
        rs = np.random.RandomState(8)
        M = 256
        coords = torch.from_numpy(rs.uniform(-1, 1, size=(M, 1, 3)).astype(np.float32))
        d = rs.standard_normal(size=(M, 3)).astype(np.float32)
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        ray_d = torch.from_numpy(d)
        chans = {'density', 'rgb', 'delta_density', 'panoptic_density', 'semantics', 'inst_embedding'}
        out = nef(coords=coords, ray_d=ray_d, pidx=None, lod_idx=None, channels=chans)
    for name in ("decoder_density", "decoder_color", "decoder_semantics", "decoder_inst", "decoder_delta_density"):
        dec = getattr(nef, name)
        lins = list(dec.layers) + [dec.lout]
        for li, lin in enumerate(lins):
            save[f"{name}_w{li}"] = lin.weight.detach().numpy()
            save[f"{name}_b{li}"] = lin.bias.detach().numpy()
        save[f"{name}_n"] = len(lins)
    save.update(coords=coords.numpy(), ray_d=ray_d.numpy(), res=np.array(res, np.float32), log2T=log2T, L=L,
                seed_main=800, seed_delta=801, **{"nef_" + k: out[k].numpy() for k in sorted(out)})      # sorted: the archive's member order must not follow set hashing
    # ---- tracer on a fixed packed scene
    N, S = 40, 20
    origins = torch.from_numpy(rs.uniform(-0.3, 0.3, size=(N, 3)).astype(np.float32))
    dd = rs.standard_normal(size=(N, 3)).astype(np.float32)
    dd /= np.linalg.norm(dd, axis=1, keepdims=True)
    dirs = torch.from_numpy(dd)
    jitter = torch.from_numpy(rs.uniform(0, 1, size=(N, S)).astype(np.float32))
    occ = torch.from_numpy(rs.uniform(size=(8, 8, 8)) > 0.3)
    ridx, pidx, samples, depths, deltas, boundary = orender.raymarch_ray(origins, dirs, 0.0, 2.0, S, jitter, occ, 3)
    keep = ridx != 3                                                     # ray 3 has no samples
    ridx, pidx, samples, depths, deltas = ridx[keep], pidx[keep], samples[keep], depths[keep], deltas[keep]
    boundary = orender.mark_pack_boundaries(ridx)
    Mt = ridx.shape[0]
    density = torch.from_numpy((rs.gamma(1.0, 8.0, size=(Mt, 1, 1)) * (rs.uniform(size=(Mt, 1, 1)) > 0.3)).astype(np.float32))
    pdens = torch.from_numpy((rs.gamma(1.0, 6.0, size=(Mt, 1, 1)) * (rs.uniform(size=(Mt, 1, 1)) > 0.2)).astype(np.float32))
    rgb = torch.from_numpy(rs.uniform(size=(Mt, 1, 3)).astype(np.float32))
    sem = torch.softmax(torch.from_numpy(rs.standard_normal(size=(Mt, C)).astype(np.float32)), -1)
    inst = torch.softmax(torch.from_numpy(rs.standard_normal(size=(Mt, I)).astype(np.float32)), -1)

    class Grid:
        num_lods = 4
        active_lods = [0, 1, 2, 3]

        def raymarch(self, rays, level, num_samples, raymarch_type):
            return ridx, pidx, samples, depths, deltas, boundary

    class Nef:
        grid = Grid()
        device = 'cpu'

        def __call__(self, coords, ray_d, pidx, lod_idx, channels):
            full = {'density': density, 'rgb': rgb, 'semantics': sem, 'inst_embedding': inst, 'panoptic_density': pdens}
            if isinstance(channels, str):
                return full[channels]
            return {c: full[c] for c in channels}

    rays = types.SimpleNamespace(origins=origins, dirs=dirs)
    save.update(t_N=N, t_S=S, t_ridx=ridx.numpy(), t_depths=depths.numpy(), t_deltas=deltas.numpy(), t_boundary=boundary.numpy(),
                t_density=density.numpy(), t_panoptic_density=pdens.numpy(), t_rgb=rgb.numpy(), t_semantics=sem.numpy(),
                t_inst_embedding=inst.numpy(), t_origins=origins.numpy(), t_dirs=dirs.numpy(), t_jitter=jitter.numpy(),
                t_occ=occ.numpy(), t_empty_ray=3)
    for bg in ("white", "black"):
        tr = dd_tracer_mod.PanopticDDensityPackedRFTracer(ray_sparcity_reg=0.01, raymarch_type='ray', num_steps=S, bg_color=bg)
        rb = tr.trace(Nef(), {'rgb', 'depth', 'semantics', 'inst_embedding'}, set(), rays, lod_idx=None,
                      raymarch_type='ray', num_steps=S, bg_color=bg, stage='train')
        for ch in ('rgb', 'alpha', 'hit', 'depth', 'semantics', 'inst_embedding', 'ray_sparcity_loss'):
            save[f"t_{bg}_{ch}"] = getattr(rb, ch).numpy()
    np.savez_compressed(os.path.join(HERE, "g8_dd.npz"), **save)


def g9(nef_mod):
    """The BASE field pc_nerf/panoptic_nef.py::PanopticNeF on a HashGridTorch grid: channels, and reference autograd of a fixed linear
    functional of them with sem_detach / inst_detach on and off (does the panoptic term reach the grid?), plus inst_direct_pos."""
    L, log2T, C, I = 8, 10, 6, 200
    res = [16] * (L - 1) + [256]
    rs = np.random.RandomState(19)
    M = 192
    coords = torch.from_numpy(rs.uniform(-1, 1, size=(M, 1, 3)).astype(np.float32))
    d = rs.standard_normal(size=(M, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    ray_d = torch.from_numpy(d)
    G = {"rgb": torch.from_numpy(rs.standard_normal(size=(M, 1, 3)).astype(np.float32)),
         "density": torch.from_numpy(rs.standard_normal(size=(M, 1, 1)).astype(np.float32)),
         "semantics": torch.from_numpy(rs.standard_normal(size=(M, C)).astype(np.float32)),
         "inst_embedding": torch.from_numpy(rs.standard_normal(size=(M, I)).astype(np.float32) * 30)}
    tab = table_from_seed(900, (L, 2 ** log2T, 2), "normal") * np.float32(0.5)
    save = dict(coords=coords.numpy(), ray_d=ray_d.numpy(), res=np.array(res, np.float32), log2T=log2T, L=L, seed_main=900,
                **{"G_" + k: v.numpy() for k, v in G.items()})
    for tag, sd, idt, direct in (("dd", True, True, False), ("ll", False, False, False), ("ld", False, True, False), ("pos", True, True, True)):
        torch.manual_seed(9)
        nef = nef_mod.PanopticNeF(
            grid_type='HashGridTorch', interpolation_type='linear', multiscale_type='cat', feature_dim=2, num_lods=L,
            base_lod=2, hidden_dim=64, num_layers=1, activation_type='relu', layer_type='none', embedder_type='positional',
            view_multires=4, pos_multires=4, position_input=False, num_classes=C, num_instances=I, sem_num_layers=2,
            sem_hidden_dim=64, sem_softmax=True, inst_num_layers=1, inst_hidden_dim=64, inst_softmax=True, sem_detach=sd, inst_detach=idt,
            panoptic_features_type='position' if direct else None, codebook_bitwidth=log2T)
        nef.inst_direct_pos = direct           # read at panoptic_nef.py:350, set by no constructor
        nef.grid.init_from_resolutions(res)
        with torch.no_grad():
            for i in range(L):
                nef.grid.embedder.embeddings[i].weight.copy_(torch.from_numpy(tab[i]))
        chans = {'density', 'rgb', 'inst_embedding'} if direct else {'density', 'rgb', 'semantics', 'inst_embedding'}
        out = nef(coords=coords, ray_d=ray_d, pidx=None, lod_idx=None, channels=chans)
        loss = sum((out[c] * G[c].reshape(out[c].shape)).sum() for c in sorted(chans))       # sorted: fixed summation and archive order
        loss.backward()
        for c in sorted(chans):
            save[f"{tag}_{c}"] = out[c].detach().numpy()
        save[f"{tag}_dtables"] = np.stack([nef.grid.embedder.embeddings[i].weight.grad.numpy() for i in range(L)])
        for name in ("decoder_density", "decoder_color", "decoder_semantics", "decoder_inst"):
            dec = getattr(nef, name)
            lins = list(dec.layers) + [dec.lout]
            for li, lin in enumerate(lins):
                wtag = tag if tag in ("dd", "pos") else "dd"       # same seed, same shapes: ll / ld carry dd's weights (checked, stored once)
                if wtag == tag:
                    save[f"{tag}_{name}_w{li}"] = lin.weight.detach().numpy()
                    save[f"{tag}_{name}_b{li}"] = lin.bias.detach().numpy()
                else:
                    assert np.array_equal(save[f"dd_{name}_w{li}"], lin.weight.detach().numpy())
                    assert np.array_equal(save[f"dd_{name}_b{li}"], lin.bias.detach().numpy())
                if lin.weight.grad is not None:
                    save[f"{tag}_{name}_dw{li}"] = lin.weight.grad.numpy()
                    save[f"{tag}_{name}_db{li}"] = lin.bias.grad.numpy()
            save[f"{tag}_{name}_n"] = len(lins)
        save[f"{tag}_nef_type"] = nef.get_nef_type()
    np.savez_compressed(os.path.join(HERE, "g9_base_nef.npz"), **save)


def g10():
    """SciPy's own answers (scipy.optimize.linear_sum_assignment - the call of loss/lin_assignment_things.py:45 / loss/lin_assignment.py:22 - of the SciPy installed
    here) on 48 small cost matrices with rows <= columns: random fp32-valued, tie-heavy integer, constant, half-integer and outlier-masked (10000) ones.  The
    oracle's sequential restatement (oracle/lin_assign.py::lsap_jv) and the device kernel (pag_assign_solve) must give these columns whatever SciPy a later
    image ships."""
    import scipy
    from scipy.optimize import linear_sum_assignment
    rs = np.random.RandomState(10)
    save = {"scipy_version": np.array(scipy.__version__)}
    for t in range(48):
        nr = int(rs.randint(1, 25))
        nc = int(rs.randint(nr, 64))
        kind = t % 6
        if kind == 0:
            c = rs.randn(nr, nc)
        elif kind == 1:
            c = rs.randint(0, 3, (nr, nc)).astype(np.float64)
        elif kind == 2:
            c = -rs.rand(nr, nc).astype(np.float32).astype(np.float64)
        elif kind == 3:
            c = np.full((nr, nc), 0.25)
        elif kind == 4:
            c = -rs.rand(nr, nc).astype(np.float32).astype(np.float64)
            c[rs.rand(nr, nc) < 0.6] = 10000
        else:
            c = np.round(rs.randn(nr, nc) * 2) / 2
        c = c.astype(np.float32)                                   # what pag_assign_cost hands over: fp32, widened to float64 by the solver
        rows, cols = linear_sum_assignment(c.astype(np.float64))
        assert np.array_equal(rows, np.arange(nr))
        save["cost_%d" % t], save["cols_%d" % t] = c, cols.astype(np.int64)
    np.savez_compressed(os.path.join(HERE, "g10_lsap.npz"), **save)


def main():
    if "--only-g10" in sys.argv:
        g10()
        return
    install_stubs()
    torch.set_num_threads(1)
    import importlib
    with _CudaToCpu():
        hgt = importlib.import_module("grids.hash_grid_torch")
    if "--only-g7" in sys.argv:
        g7(hgt)
        return
    if "--only-g6" in sys.argv:
        g6(importlib.import_module("loss.regularizers"))
        return
    if "--only-g9" in sys.argv:
        with _CudaToCpu():
            g9(importlib.import_module("pc_nerf.panoptic_nef"))
        return
    if "--only-g8" in sys.argv:
        with _CudaToCpu():
            dd_nef = importlib.import_module("pc_nerf.panoptic_dd_nef")
        g8(dd_nef, importlib.import_module("tracers.panoptic_dd_packed_rf_tracer"))
        return
    g1_g2(hgt)
    g7(hgt)
    with _CudaToCpu():
        delta_mod = importlib.import_module("pc_nerf.panoptic_delta_nef")
    g3(delta_mod)
    tracer_mod = importlib.import_module("tracers.panoptic_packed_rf_tracer")
    g4(tracer_mod)
    la = importlib.import_module("loss.lin_assignment")
    lat = importlib.import_module("loss.lin_assignment_things")
    g5(la, lat)
    g6(importlib.import_module("loss.regularizers"))
    with _CudaToCpu():
        dd_nef = importlib.import_module("pc_nerf.panoptic_dd_nef")
    g8(dd_nef, importlib.import_module("tracers.panoptic_dd_packed_rf_tracer"))
    with _CudaToCpu():
        g9(importlib.import_module("pc_nerf.panoptic_nef"))
    g10()
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
