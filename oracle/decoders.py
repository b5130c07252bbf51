"""Oracle: tiny-MLP decoders and the neural-field wiring, torch CPU fp32.

TEST INFRASTRUCTURE - see oracle/__init__.py.

In-tree logic followed:
  pc_nerf/panoptic_nef.py:108-164        decoder shapes (density ->16, colour (16+PE)->3,
                                         semantics ->C, instances ->I), density bias[0]=1
  pc_nerf/panoptic_nef.py:72-77          view embedder (positional, view_multires freqs)
  pc_nerf/panoptic_delta_nef.py:155-259  rgb_semantics(): interp -> *lod_weights -> density MLP
                                         -> relu(ch 0) -> colour MLP on cat(density_feats, PE(-d))
                                         -> sigmoid ; delta grid on detached coords ;
                                         panoptic feats = feats.detach() + delta ;
                                         sem / inst MLP -> [sigmoid] -> [normalize] -> [/T] -> [softmax]

Third-party pieces restated from their public semantics (wisp 0.1.1, SURVEY.md
Appendix A1/A2): BasicDecoder = num_layers x (Linear + activation) then Linear
"lout"; PositionalEmbedder = cat(x, sin(x*2^k), cos(x*2^k)), frequency-major.
"""
import torch
import torch.nn.functional as F


def positional_embed(x, num_freq):
    """x [N,3] -> [N, 3 + 6*num_freq]  (wisp PositionalEmbedder, max_freq_log2 = num_freq-1)."""
    bands = 2.0 ** torch.linspace(0.0, num_freq - 1, num_freq)
    w = (x[:, None, :] * bands[None, :, None]).reshape(x.shape[0], -1)
    return torch.cat([x, torch.sin(w), torch.cos(w)], dim=-1)


def bf16_operands(t):
    """Straight-through bf16 rounding: the value a bf16 matrix-core operand carries, the gradient of the identity.  Passed as
    `operand_round` it turns the fp32 chain below into the arithmetic of a bf16-operand / fp32-accumulate implementation (weights,
    layer inputs and stored activations rounded, sums and epilogues in fp32) - used by the tests to separate rounding noise from
    real differences; the default (None) is the reference's fp32 arithmetic."""
    return t + (t.bfloat16().float() - t).detach()


def mlp(x, weights, biases, act=torch.relu, operand_round=None):
    """weights[i] is [out,in] (nn.Linear layout); activation after every layer but the last."""
    r = operand_round if operand_round is not None else (lambda t: t)
    h = x
    for i, (W, b) in enumerate(zip(weights, biases)):
        h = F.linear(r(h), r(W), b)
        if i + 1 < len(weights):
            h = act(h)
    return h


def nef_forward(feats, delta_feats, ray_d, params, channels,
                view_multires=4, lod_weights=None,
                sem_softmax=True, inst_softmax=True, sem_sigmoid=False, inst_sigmoid=False,
                sem_normalize=False, inst_normalize=False, inst_soft_temperature=0.0,
                panoptic_features_type="delta", multiscale_sum_levels=0, coords=None, pos_multires=4, operand_round=None):
    """Everything of rgb_semantics() after the two grid interpolations (pc_nerf/panoptic_delta_nef.py:170-259).

    feats, delta_feats: [M, L*F] grid features (delta may be None when no panoptic channel).
    ray_d [M,3] per-sample view direction.  params: dict name -> (weights, biases).
    panoptic_features_type: what the panoptic heads read (:210-234) - 'delta' (None in the reference), 'separate', 'appearance',
    'pos_encoding' (the embedded sample position, coords [M,3] required) or 'position'.
    multiscale_sum_levels = L: multiscale_type 'sum', the L levels' F features are summed (:172-173, :221-222).
    Returns dict with density [M,1], rgb [M,3], semantics [M,C], inst_embedding [M,I].
    """
    out = {}
    rnd = operand_round if operand_round is not None else (lambda t: t)      # stored tensors of a reduced-precision implementation
    kw = dict(operand_round=operand_round)
    if lod_weights is not None:
        feats = feats * lod_weights
    if multiscale_sum_levels:
        feats = feats.reshape(-1, multiscale_sum_levels, feats.shape[-1] // multiscale_sum_levels).sum(-2)
    feats = rnd(feats)
    dfe = rnd(mlp(feats, *params["density"], **kw))
    out["density_feats"] = dfe
    out["density"] = torch.relu(dfe[..., 0:1])
    if "rgb" in channels:
        pe = positional_embed(-ray_d, view_multires)
        out["rgb"] = torch.sigmoid(mlp(torch.cat([dfe, pe], dim=-1), *params["color"], **kw))
    if "semantics" in channels or "inst_embedding" in channels:
        if panoptic_features_type in ("delta", "separate", None):
            d = delta_feats * lod_weights if lod_weights is not None else delta_feats
            if multiscale_sum_levels:
                d = d.reshape(-1, multiscale_sum_levels, d.shape[-1] // multiscale_sum_levels).sum(-2)
            d = rnd(d)
        if panoptic_features_type in ("delta", None):
            pan = rnd(feats.detach() + d)
        elif panoptic_features_type == "separate":
            pan = d
        elif panoptic_features_type == "appearance":
            pan = feats.detach()
        elif panoptic_features_type == "pos_encoding":
            pan = positional_embed(coords.reshape(-1, 3), pos_multires)
        elif panoptic_features_type == "position":
            pan = coords.reshape(-1, 3)
        else:
            raise ValueError(panoptic_features_type)
        if "semantics" in channels:
            s = mlp(pan, *params["semantics"], **kw)
            s = torch.sigmoid(s) if sem_sigmoid else s
            s = F.normalize(s, dim=-1) if sem_normalize else s
            s = F.softmax(s, dim=-1) if sem_softmax else s
            out["semantics"] = s
        if "inst_embedding" in channels:
            e = mlp(pan, *params["inst"], **kw)
            e = torch.sigmoid(e) if inst_sigmoid else e
            e = F.normalize(e, dim=-1) if inst_normalize else e
            e = e / inst_soft_temperature if inst_soft_temperature > 0.0 else e
            e = F.softmax(e, dim=-1) if inst_softmax else e
            out["inst_embedding"] = e
    return out


def nef_forward_dd(feats, delta_feats, ray_d, params, channels, view_multires=4, lod_weights=None, separate_sem_grid=False,
                   sem_softmax=True, inst_softmax=True, sem_sigmoid=False, inst_sigmoid=False,
                   sem_normalize=False, inst_normalize=False, inst_soft_temperature=0.0):
    """Delta-density variant: pc_nerf/panoptic_dd_nef.py:130-275.  As nef_forward() plus
      delta_density    = decoder_delta_density(panoptic feats)                      (:236-241)
      panoptic_density = relu(density_feats[...,0:1].detach() + delta_density)      (:243-247)
    (separate_sem_grid: panoptic feats = delta feats alone, panoptic density = relu(delta_density)).
    params additionally holds "delta_density"."""
    out = {}
    if lod_weights is not None:
        feats = feats * lod_weights
    dfe = mlp(feats, *params["density"])
    out["density_feats"] = dfe
    out["density"] = torch.relu(dfe[..., 0:1])
    if "rgb" in channels:
        pe = positional_embed(-ray_d, view_multires)
        out["rgb"] = torch.sigmoid(mlp(torch.cat([dfe, pe], dim=-1), *params["color"]))
    if any(c in channels for c in ("delta_density", "panoptic_density", "semantics", "inst_embedding")):
        d = delta_feats * lod_weights if lod_weights is not None else delta_feats
        pan = d if separate_sem_grid else feats.detach() + d
        if "delta_density" in channels or "panoptic_density" in channels:
            dd = mlp(pan, *params["delta_density"], act=lambda t: t)      # activation 'none' (panoptic_dd_nef.py:49-56)
            out["delta_density"] = dd
            if "panoptic_density" in channels:
                out["panoptic_density"] = torch.relu(dd if separate_sem_grid else dfe[..., 0:1].detach() + dd)
        if "semantics" in channels:
            s = mlp(pan, *params["semantics"])
            s = torch.sigmoid(s) if sem_sigmoid else s
            s = F.normalize(s, dim=-1) if sem_normalize else s
            s = F.softmax(s, dim=-1) if sem_softmax else s
            out["semantics"] = s
        if "inst_embedding" in channels:
            e = mlp(pan, *params["inst"])
            e = torch.sigmoid(e) if inst_sigmoid else e
            e = F.normalize(e, dim=-1) if inst_normalize else e
            e = e / inst_soft_temperature if inst_soft_temperature > 0.0 else e
            e = F.softmax(e, dim=-1) if inst_softmax else e
            out["inst_embedding"] = e
    return out


def nef_forward_base(feats, ray_d, params, channels, view_multires=4, lod_weights=None, sem_detach=True, inst_detach=True,
                     inst_direct_pos=False, coords=None, sem_softmax=True, inst_softmax=True, sem_sigmoid=False, inst_sigmoid=False,
                     sem_normalize=False, inst_normalize=False):
    """The base field, pc_nerf/panoptic_nef.py:253-363 (PanopticNeF.rgb_semantics): one grid; the semantic head reads
    `feats.detach() if sem_detach else feats` (:338), the instance head `coords` when inst_direct_pos (:350-351) else
    `feats.detach() if inst_detach else feats` (:353); with inst_softmax the instance output is softmax(decoder(x)) - :358 re-evaluates
    the decoder, so the sigmoid / normalize results of :355-356 are discarded."""
    out = {}
    if lod_weights is not None:
        feats = feats * lod_weights
    dfe = mlp(feats, *params["density"])
    out["density_feats"] = dfe
    out["density"] = torch.relu(dfe[..., 0:1])
    if "rgb" in channels:
        pe = positional_embed(-ray_d, view_multires)
        out["rgb"] = torch.sigmoid(mlp(torch.cat([dfe, pe], dim=-1), *params["color"]))
    if "semantics" in channels:
        s = mlp(feats.detach() if sem_detach else feats, *params["semantics"])
        s = torch.sigmoid(s) if sem_sigmoid else s
        s = F.normalize(s, dim=-1) if sem_normalize else s
        out["semantics"] = F.softmax(s, dim=-1) if sem_softmax else s
    if "inst_embedding" in channels:
        x = coords if inst_direct_pos else (feats.detach() if inst_detach else feats)
        e = mlp(x, *params["inst"])
        if inst_softmax:
            e = F.softmax(e, dim=-1)
        else:
            e = torch.sigmoid(e) if inst_sigmoid else e
            e = F.normalize(e, dim=-1) if inst_normalize else e
        out["inst_embedding"] = e
    return out
