"""CPU oracle for the Signal hot path -- TEST INFRASTRUCTURE ONLY.

A fresh fp32 restatement (plain PyTorch on CPU, written functionally over a flat
``state_dict`` whose keys are the reference's parameter names) of the algorithm
in maxingan2412/Signal that the HIP path in ``signal_amd/csrc`` implements.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this file; the product package never does.

Parity status: **pinned** -- the reference ships no tests or golden vectors for
this path, so the pin is the committed fixtures under
``tests/golden/`` that ``tests/golden/make_golden.py`` generated in the build
container by importing the reference's own leaf modules from /root/reference
and running them on PCG64-seeded weights and inputs; ``tests/test_oracle_golden.py``
checks every function here against those fixtures.

Every function cites the reference file:line it restates (paths relative to
/root/reference).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]

MODALITIES = ("RGB", "NI", "TI")


# --------------------------------------------------------------------------- #
# configuration (the hot-path-relevant yacs keys)
# --------------------------------------------------------------------------- #
@dataclass
class RefConfig:
    """Subset of config/defaults.py + configs/*/Signal.yml the path consumes."""
    size_train: Tuple[int, int] = (256, 128)      # INPUT.SIZE_TRAIN
    width: int = 768                              # CLIP ViT-B/16 vision width
    layers: int = 12
    heads: int = 12
    patch: int = 16                               # MODEL.STRIDE_SIZE == patch size
    out_dim: int = 512                            # visual.proj columns
    sie_camera: bool = True                       # MODEL.SIE_CAMERA
    sie_coe: float = 1.0                          # MODEL.SIE_COE
    direct: int = 1                               # MODEL.DIRECT
    use_a: bool = True                            # MODEL.USE_A  (SIM)
    use_b: bool = True                            # MODEL.USE_B  (GAM+LAM)
    topk: int = 80                                # MODEL.TOPK
    keep_ratio: Optional[float] = None            # MODEL.KEEP_RATIO when MODEL.FIXED_KEEP_RATIO (make_model.py:107), else None
    stage: str = "together_CLS_Patch"             # MODEL.stageName
    id_loss_weight: float = 0.25
    triplet_loss_weight: float = 1.0
    gram_loss_weight: float = 0.2
    pat_loss_weight: float = 0.2
    num_instance: int = 8
    sim_heads: int = 8                            # useA.py:450
    label_smooth_eps: float = 0.1                 # softmax_loss.py:16
    num_classes: int = 171
    camera_num: int = 4

    @property
    def grid(self) -> Tuple[int, int]:
        return self.size_train[0] // self.patch, self.size_train[1] // self.patch

    @property
    def tokens(self) -> int:
        h, w = self.grid
        return h * w + 1


def rgbnt201_config(**kw) -> RefConfig:
    """configs/RGBNT201/Signal.yml"""
    return RefConfig(**kw)


def rgbnt100_config(**kw) -> RefConfig:
    """configs/RGBNT100/Signal.yml (128x256, TOPK 112, DIRECT 0, alpha=beta=0.1)."""
    base = dict(size_train=(128, 256), topk=112, direct=0, gram_loss_weight=0.1,
                pat_loss_weight=0.1, num_instance=16, num_classes=50, camera_num=8)
    base.update(kw)
    return RefConfig(**base)


# --------------------------------------------------------------------------- #
# portable PRNG: weights and inputs come from NumPy PCG64 so that fixtures are
# seeds + small outputs (torch's generator is not relied on)
# --------------------------------------------------------------------------- #
def _rng(seed: int) -> np.random.Generator:
    return np.random.Generator(np.random.PCG64(seed))


def randn(gen: np.random.Generator, *shape, std: float = 1.0) -> Tensor:
    return torch.from_numpy((gen.standard_normal(shape) * std).astype(np.float32))


def uniform(gen: np.random.Generator, *shape, bound: float) -> Tensor:
    return torch.from_numpy(gen.uniform(-bound, bound, size=shape).astype(np.float32))


def init_state_dict(cfg: RefConfig, seed: int = 1234, head_scale: float = 1.0) -> SD:
    """Random-init parameters with the reference's names, shapes and scales
    (clip/model.py:436-445,212-221; meta_arch.py:25-31,85-86;
    make_model.py:77-81; useB.py:56). Values come from PCG64(seed), not torch."""
    g = _rng(seed)
    D, Dh, L = cfg.width, cfg.out_dim, cfg.tokens
    sd: SD = {}
    base = "clip_vision_encoder.base."
    sc = D ** -0.5
    sd[base + "class_embedding"] = randn(g, D, std=sc)
    sd[base + "positional_embedding"] = randn(g, L, D, std=sc)
    sd[base + "proj"] = randn(g, D, Dh, std=sc)
    fan_in = 3 * cfg.patch * cfg.patch
    sd[base + "conv1.weight"] = uniform(g, D, 3, cfg.patch, cfg.patch, bound=1.0 / math.sqrt(fan_in))
    for ln in ("ln_pre", "ln_post"):
        sd[base + ln + ".weight"] = 1.0 + randn(g, D, std=0.02)
        sd[base + ln + ".bias"] = randn(g, D, std=0.02)
    for i in range(cfg.layers):
        p = f"{base}transformer.resblocks.{i}."
        sd[p + "attn.in_proj_weight"] = uniform(g, 3 * D, D, bound=math.sqrt(6.0 / (4 * D)))
        sd[p + "attn.in_proj_bias"] = randn(g, 3 * D, std=0.02)
        sd[p + "attn.out_proj.weight"] = randn(g, D, D, std=0.02)
        sd[p + "attn.out_proj.bias"] = randn(g, D, std=0.02)
        sd[p + "ln_1.weight"] = 1.0 + randn(g, D, std=0.02)
        sd[p + "ln_1.bias"] = randn(g, D, std=0.02)
        sd[p + "ln_2.weight"] = 1.0 + randn(g, D, std=0.02)
        sd[p + "ln_2.bias"] = randn(g, D, std=0.02)
        sd[p + "mlp.c_fc.weight"] = randn(g, 4 * D, D, std=0.02)
        sd[p + "mlp.c_fc.bias"] = randn(g, 4 * D, std=0.02)
        sd[p + "mlp.c_proj.weight"] = randn(g, D, 4 * D, std=0.02)
        sd[p + "mlp.c_proj.bias"] = randn(g, D, std=0.02)
    if cfg.sie_camera:
        sd["clip_vision_encoder.cv_embed"] = randn(g, cfg.camera_num, 1, D, std=0.02)

    C = cfg.num_classes
    if cfg.direct:
        names = [("bottleneck", "classifier", 3 * Dh)]
    else:
        names = [(f"bottleneck_{m}", f"classifier_{m}", Dh) for m in "rnt"]
    if cfg.use_a:
        names.append(("bottleneck_var", "classifier_var", 3 * Dh))
    for bn, cl, n in names:
        sd[bn + ".weight"] = 1.0 + randn(g, n, std=0.02)
        sd[bn + ".bias"] = torch.zeros(n)
        sd[bn + ".running_mean"] = torch.zeros(n)
        sd[bn + ".running_var"] = torch.ones(n)
        sd[cl + ".weight"] = randn(g, C, n, std=0.001 * head_scale)

    if cfg.use_a:
        b = 1.0 / math.sqrt(Dh)
        for nm in ("W_q", "W_k", "W_v"):
            sd[f"SIM.token_selection.{nm}.weight"] = uniform(g, Dh, Dh, bound=b)
            sd[f"SIM.token_selection.{nm}.bias"] = uniform(g, Dh, bound=b)
        m = "SIM.modal_interactive."
        sd[m + "cross_attn.in_proj_weight"] = uniform(g, 3 * Dh, Dh, bound=math.sqrt(6.0 / (4 * Dh)))
        sd[m + "cross_attn.in_proj_bias"] = randn(g, 3 * Dh, std=0.02)
        sd[m + "cross_attn.out_proj.weight"] = uniform(g, Dh, Dh, bound=b)
        sd[m + "cross_attn.out_proj.bias"] = randn(g, Dh, std=0.02)
        sd[m + "ffn.0.weight"] = uniform(g, 2 * Dh, Dh, bound=b)
        sd[m + "ffn.0.bias"] = uniform(g, 2 * Dh, bound=b)
        sd[m + "ffn.2.weight"] = uniform(g, Dh, 2 * Dh, bound=1.0 / math.sqrt(2 * Dh))
        sd[m + "ffn.2.bias"] = uniform(g, Dh, bound=1.0 / math.sqrt(2 * Dh))
        for nm in ("norm1", "norm2"):
            sd[m + nm + ".weight"] = 1.0 + randn(g, Dh, std=0.02)
            sd[m + nm + ".bias"] = randn(g, Dh, std=0.02)
    if cfg.use_b:
        sd["AlignM.contra_temp"] = torch.tensor(0.07)
        for m in "rnt":
            p = f"AlignM.DAS_{m}."
            b = 1.0 / math.sqrt(Dh)
            sd[p + "conv_offset.0.weight"] = uniform(g, Dh, Dh, 1, 1, bound=b)
            sd[p + "conv_offset.0.bias"] = uniform(g, Dh, bound=b)
            sd[p + "conv_offset.2.weight"] = uniform(g, Dh, 1, 4, 4, bound=0.25)
            sd[p + "conv_offset.2.bias"] = uniform(g, Dh, bound=0.25)
            sd[p + "conv_offset.4.weight"] = uniform(g, 1, Dh, 1, 1, bound=b)
            sd[p + "proj_q.weight"] = uniform(g, Dh, Dh, 1, 1, bound=b)
            sd[p + "proj_q.bias"] = uniform(g, Dh, bound=b)
    return sd


def synthetic_batch(cfg: RefConfig, batch: int, seed: int = 1234, ids: Optional[int] = None):
    """SURVEY.md section 8(d) 'Synthetic inputs': independent N(0,1) fp32 images per
    modality, P x K identity blocks, cam ~ U{0..cams-1}; all from PCG64(seed)."""
    g = _rng(seed)
    H, W = cfg.size_train
    img = {m: randn(g, batch, 3, H, W) for m in MODALITIES}
    k = cfg.num_instance if ids is None else batch // ids
    k = max(1, min(k, batch))
    vid = torch.arange(batch, dtype=torch.int64) // k
    cam = torch.from_numpy(g.integers(0, cfg.camera_num, size=batch).astype(np.int64))
    return img, vid, cam


# --------------------------------------------------------------------------- #
# ViT (modeling/clip/model.py)
# --------------------------------------------------------------------------- #
def layer_norm(x: Tensor, w: Tensor, b: Tensor, eps: float = 1e-5) -> Tensor:
    """clip/model.py:154-160 (fp32 LayerNorm, eps 1e-5)."""
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) * torch.rsqrt(var + eps) * w + b


def quick_gelu(x: Tensor) -> Tensor:
    """clip/model.py:163-165."""
    return x * torch.sigmoid(1.702 * x)


def mha_self(x: Tensor, w_in: Tensor, b_in: Tensor, w_o: Tensor, b_o: Tensor, heads: int) -> Tensor:
    """nn.MultiheadAttention(x,x,x) as called at clip/model.py:223-225.
    x: [S, L, D] (sequence-major; the reference's [L,N,D] layout is a transpose)."""
    S, L, D = x.shape
    hd = D // heads
    qkv = x @ w_in.t() + b_in                                   # [S,L,3D]
    q, k, v = qkv.split(D, dim=-1)
    q = q.reshape(S, L, heads, hd).transpose(1, 2)              # [S,h,L,hd]
    k = k.reshape(S, L, heads, hd).transpose(1, 2)
    v = v.reshape(S, L, heads, hd).transpose(1, 2)
    att = torch.softmax((q @ k.transpose(-1, -2)) / math.sqrt(hd), dim=-1)
    o = (att @ v).transpose(1, 2).reshape(S, L, D)
    return o @ w_o.t() + b_o


def vit_block(sd: SD, pre: str, x: Tensor, heads: int) -> Tensor:
    """ResidualAttentionBlock.forward_ori, clip/model.py:227-231."""
    h = layer_norm(x, sd[pre + "ln_1.weight"], sd[pre + "ln_1.bias"])
    x = x + mha_self(h, sd[pre + "attn.in_proj_weight"], sd[pre + "attn.in_proj_bias"],
                     sd[pre + "attn.out_proj.weight"], sd[pre + "attn.out_proj.bias"], heads)
    h = layer_norm(x, sd[pre + "ln_2.weight"], sd[pre + "ln_2.bias"])
    u = quick_gelu(h @ sd[pre + "mlp.c_fc.weight"].t() + sd[pre + "mlp.c_fc.bias"])
    return x + u @ sd[pre + "mlp.c_proj.weight"].t() + sd[pre + "mlp.c_proj.bias"]


def vit_embed(sd: SD, cfg: RefConfig, img: Tensor, cv_emb: Optional[Tensor]) -> Tensor:
    """clip/model.py:448-459: patch conv, CLS, camera embedding on the CLS row
    (before the positional embedding), positional embedding, ln_pre."""
    base = "clip_vision_encoder.base."
    B = img.shape[0]
    p = cfg.patch
    h, w = cfg.grid
    # non-overlapping conv == per-patch linear map over (c, dy, dx)
    patches = img.reshape(B, 3, h, p, w, p).permute(0, 2, 4, 1, 3, 5).reshape(B, h * w, 3 * p * p)
    tok = patches @ sd[base + "conv1.weight"].reshape(cfg.width, -1).t()
    cls = sd[base + "class_embedding"].expand(B, 1, cfg.width)
    if cv_emb is not None:
        cls = cls + cv_emb.reshape(B, 1, cfg.width)
    x = torch.cat([cls, tok], dim=1) + sd[base + "positional_embedding"]
    return layer_norm(x, sd[base + "ln_pre.weight"], sd[base + "ln_pre.bias"])


def vit_forward(sd: SD, cfg: RefConfig, img: Tensor, cam_label: Optional[Tensor],
                return_hidden: bool = False):
    """build_transformer.forward (meta_arch.py:96-112) + VisionTransformer.forward
    (clip/model.py:447-488). Returns (patches [B,Lp,512], cls [B,512])."""
    base = "clip_vision_encoder.base."
    cv = None
    if cfg.sie_camera and cam_label is not None:
        cv = cfg.sie_coe * sd["clip_vision_encoder.cv_embed"][cam_label]
    x = vit_embed(sd, cfg, img, cv)
    hidden = [x]
    for i in range(cfg.layers):
        x = vit_block(sd, f"{base}transformer.resblocks.{i}.", x, cfg.heads)
        if return_hidden:
            hidden.append(x)
    x = layer_norm(x, sd[base + "ln_post.weight"], sd[base + "ln_post.bias"]) @ sd[base + "proj"]
    if return_hidden:
        return x[:, 1:], x[:, 0], hidden
    return x[:, 1:], x[:, 0]


# --------------------------------------------------------------------------- #
# SIM (modeling/AddModule/useA.py)
# --------------------------------------------------------------------------- #
def sim_scores_intra(patches: Tensor, cls: Tensor) -> Tensor:
    """useA.py:72-74. patches [3,B,Lp,d], cls [3,B,d] -> softmax scores [3,B,Lp]."""
    d = patches.shape[-1]
    s = torch.einsum("mbd,mbld->mbl", cls, patches) / math.sqrt(d)
    return torch.softmax(s, dim=-1)


def sim_scores_inter(sd: SD, patches: Tensor, cls: Tensor) -> Tensor:
    """useA.py:116-129. Returns softmax scores [B,3,3*Lp] over keys ordered [RGB|NI|TI]."""
    d = patches.shape[-1]
    pre = "SIM.token_selection."
    q = cls.transpose(0, 1) @ sd[pre + "W_q.weight"].t() + sd[pre + "W_q.bias"]          # [B,3,d]
    keys = torch.cat([patches[0], patches[1], patches[2]], dim=1)                       # [B,3Lp,d]
    k = keys @ sd[pre + "W_k.weight"].t() + sd[pre + "W_k.bias"]
    s = (q @ k.transpose(1, 2)) / math.sqrt(d)
    return torch.softmax(s, dim=2)


# candidate order of the two *other* modalities for query modality m (useA.py:136-151)
INTER_OTHERS = ((1, 2), (0, 2), (0, 1))


def _topk_set(scores: Tensor, k: int) -> Tuple[Tensor, Tensor]:
    """Set of the k largest entries per row, lowest index first on ties, plus a
    per-row ``tie_free`` flag (k-th value strictly greater than the (k+1)-th).
    torch.topk's CPU tie order is unspecified, so bit-exact index parity is only
    defined on tie-free rows."""
    n = scores.shape[-1]
    k = min(k, n)
    order = torch.sort(scores, dim=-1, descending=True, stable=True).indices
    mask = torch.zeros_like(scores, dtype=torch.bool)
    mask.scatter_(-1, order[..., :k], True)
    srt = torch.gather(scores, -1, order)
    if k < n:
        tie_free = srt[..., k - 1] > srt[..., k]
    else:
        tie_free = torch.ones(scores.shape[:-1], dtype=torch.bool)
    return mask, tie_free


def keep_ratio_trim(mask: Tensor, patches: Tensor, cls: Tensor, keep_ratio: float):
    """The exact keep-ratio branch of TokenSelection.forward (useA.py:253-316): every (sample, modality) keeps exactly
    max_keep = int(Lp * keep_ratio) tokens.  More selected than that: the max_keep of the SELECTED tokens with the largest
    RAW intra-modal dot product cls . patch (un-scaled, no softmax, :259-261) stay; fewer: the best un-selected ones are added.
    Returns (mask, tie_free [B])."""
    M, B, Lp, _ = patches.shape
    max_keep = int(Lp * keep_ratio)
    raw = torch.einsum("mbd,mbld->mbl", cls, patches)
    out = mask.clone()
    tie_free = torch.ones(B, dtype=torch.bool)
    for m in range(M):
        for b in range(B):
            cur = mask[m, b]
            cnt = int(cur.sum())
            if cnt > max_keep:
                idx = cur.nonzero().flatten()
                keep, tf = _topk_set(raw[m, b, idx], max_keep)
                out[m, b] = False
                out[m, b, idx[keep]] = True
                tie_free[b] &= tf
            elif cnt < max_keep:
                idx = (~cur).nonzero().flatten()
                add, tf = _topk_set(raw[m, b, idx], min(max_keep - cnt, len(idx)))
                out[m, b, idx[add]] = True
                tie_free[b] &= tf
    return out, tie_free


def sim_select(sd: SD, patches: Tensor, cls: Tensor, topk: int, keep_ratio: Optional[float] = None):
    """TokenSelection.forward (useA.py:223-316; keep_ratio = the optional exact-count branch :253-316).
    patches [3,B,Lp,d], cls [3,B,d]. Returns (mask [3,B,Lp] bool, tie_free [B] bool)."""
    M, B, Lp, _ = patches.shape
    k1, k2 = topk, 2 * topk
    intra = sim_scores_intra(patches, cls)
    m_intra, tf_i = _topk_set(intra, k1)                       # [3,B,Lp]
    inter = sim_scores_inter(sd, patches, cls)                  # [B,3,3Lp]
    m_inter = torch.zeros(M, B, Lp, dtype=torch.bool)
    tie_free = tf_i.all(dim=0)
    for m in range(3):
        a, b = INTER_OTHERS[m]
        cand = torch.cat([inter[:, m, a * Lp:(a + 1) * Lp], inter[:, m, b * Lp:(b + 1) * Lp]], dim=1)
        sel, tf = _topk_set(cand, k2)                           # [B,2Lp]
        m_inter[a] |= sel[:, :Lp]
        m_inter[b] |= sel[:, Lp:]
        tie_free &= tf
    mask = m_intra | m_inter
    if keep_ratio is not None:
        mask, tf = keep_ratio_trim(mask, patches, cls, keep_ratio)
        tie_free &= tf
    return mask, tie_free


def gelu_erf(x: Tensor) -> Tensor:
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


def sim_interact(sd: SD, patches: Tensor, cls: Tensor, mask: Tensor, heads: int = 8) -> Tensor:
    """ModalInteractive.forward (useA.py:364-411): 3 CLS queries attend over the 3*Lp
    masked (zeroed, not removed) tokens; LN; FFN(erf GELU); LN; concat -> [B,3d]."""
    pre = "SIM.modal_interactive."
    M, B, Lp, d = patches.shape
    hd = d // heads
    sel = patches * mask.unsqueeze(-1).to(patches.dtype)
    q_in = cls.transpose(0, 1)                                            # [B,3,d]
    kv = torch.cat([sel[0], sel[1], sel[2]], dim=1)                       # [B,3Lp,d]
    w, b = sd[pre + "cross_attn.in_proj_weight"], sd[pre + "cross_attn.in_proj_bias"]
    q = (q_in @ w[:d].t() + b[:d]).reshape(B, 3, heads, hd).transpose(1, 2)
    k = (kv @ w[d:2 * d].t() + b[d:2 * d]).reshape(B, 3 * Lp, heads, hd).transpose(1, 2)
    v = (kv @ w[2 * d:].t() + b[2 * d:]).reshape(B, 3 * Lp, heads, hd).transpose(1, 2)
    att = torch.softmax((q @ k.transpose(-1, -2)) / math.sqrt(hd), dim=-1)
    o = (att @ v).transpose(1, 2).reshape(B, 3, d)
    o = o @ sd[pre + "cross_attn.out_proj.weight"].t() + sd[pre + "cross_attn.out_proj.bias"]
    y = layer_norm(q_in + o, sd[pre + "norm1.weight"], sd[pre + "norm1.bias"])
    f = gelu_erf(y @ sd[pre + "ffn.0.weight"].t() + sd[pre + "ffn.0.bias"])
    f = f @ sd[pre + "ffn.2.weight"].t() + sd[pre + "ffn.2.bias"]
    z = layer_norm(y + f, sd[pre + "norm2.weight"], sd[pre + "norm2.bias"])
    return z.reshape(B, 3 * d)


def sim_forward(sd: SD, cfg: RefConfig, patches: Tensor, cls: Tensor, force_mask: Optional[Tensor] = None):
    """Select_Interactive_Module.forward (useA.py:454-476).
    force_mask (test aid, [3,B,Lp] bool): use this token selection instead of the module's own.  The selection is a
    discrete top-k on fp32 scores; a device under test whose 16-bit tokens flip a near-tie selects another token, and
    everything downstream (features, every gradient) is then compared under the device's selection, while the selection
    itself is compared separately."""
    with torch.no_grad():
        mask, tie_free = sim_select(sd, patches, cls, cfg.topk, cfg.keep_ratio)
    used = mask if force_mask is None else force_mask.to(torch.bool)
    return sim_interact(sd, patches, cls, used, cfg.sim_heads), mask, tie_free


# --------------------------------------------------------------------------- #
# GAM / LAM (modeling/AddModule/useB.py, DAS.py, utils/volume.py)
# --------------------------------------------------------------------------- #
def gram_volume3(r: Tensor, n: Tensor, t: Tensor) -> Tensor:
    """volume_computation3 (utils/volume.py:14-62) in closed form:
    V[i,j] = sqrt(|det Gram(r_i, n_j, t_j)|)."""
    ll = (r * r).sum(-1)[:, None]
    lv, la = r @ n.t(), r @ t.t()
    vv, va, aa = (n * n).sum(-1)[None], (n * t).sum(-1)[None], (t * t).sum(-1)[None]
    det = ll * (vv * aa - va * va) - lv * (lv * aa - va * la) + la * (lv * va - vv * la)
    return torch.sqrt(torch.abs(det))


def ce_label_smooth(logits: Tensor, target: Tensor, eps: float) -> Tensor:
    """F.cross_entropy(..., label_smoothing=eps), mean reduction (useB.py:122-123)."""
    logp = torch.log_softmax(logits, dim=-1)
    nll = -logp.gather(-1, target[:, None]).squeeze(-1)
    smooth = -logp.mean(-1)
    return ((1.0 - eps) * nll + eps * smooth).mean()


def gam_loss(sd: SD, patches: Tensor) -> Tensor:
    """AlignmentM.Cls_Align (useB.py:76-126). patches [3,B,Lp,d]."""
    feats = [F.normalize(patches[m].mean(dim=1), dim=-1) for m in range(3)]
    V = gram_volume3(*feats) / sd["AlignM.contra_temp"]
    tgt = torch.arange(V.shape[0])
    return 0.5 * (ce_label_smooth(-V, tgt, 0.1) + ce_label_smooth(-V.t(), tgt, 0.1))


def das_offsets(sd: SD, pre: str, fmap: Tensor) -> Tensor:
    """conv_offset(proj_q(x)) of DA_sample.forward (DAS.py:129-136).
    fmap [B,d,H,W] -> raw offsets o [B,Hk,Wk] (single channel)."""
    q = F.conv2d(fmap, sd[pre + "proj_q.weight"], sd[pre + "proj_q.bias"])
    a = gelu_erf(F.conv2d(q, sd[pre + "conv_offset.0.weight"], sd[pre + "conv_offset.0.bias"]))
    d = a.shape[1]
    a = gelu_erf(F.conv2d(a, sd[pre + "conv_offset.2.weight"], sd[pre + "conv_offset.2.bias"],
                          stride=4, groups=d))
    return F.conv2d(a, sd[pre + "conv_offset.4.weight"]).squeeze(1)


def das_positions(o: Tensor) -> Tuple[Tensor, Tensor]:
    """DAS.py:143-153: the same scalar o drives both axes;
    reference points ((i+0.5)/(n-1))*2-1; clamp to [-1,1]. Returns (p_y, p_x) [B,Hk,Wk]."""
    B, Hk, Wk = o.shape
    t = torch.tanh(o)
    ry = ((torch.arange(Hk, dtype=o.dtype) + 0.5) / (Hk - 1.0) * 2.0 - 1.0)[None, :, None]
    rx = ((torch.arange(Wk, dtype=o.dtype) + 0.5) / (Wk - 1.0) * 2.0 - 1.0)[None, None, :]
    py = (t * (1.0 / (Hk - 1.0)) * 2.0 + ry).clamp(-1.0, 1.0)
    px = (t * (1.0 / (Wk - 1.0)) * 2.0 + rx).clamp(-1.0, 1.0)
    return py, px


def bilinear_sample(fmap: Tensor, py: Tensor, px: Tensor) -> Tensor:
    """F.grid_sample(bilinear, zeros padding, align_corners=True) (DAS.py:158-163)
    written out: fmap [B,d,H,W], normalised coords [B,Hk,Wk] -> [B,d,Hk,Wk]."""
    B, d, H, W = fmap.shape
    fy = (py + 1.0) * 0.5 * (H - 1)
    fx = (px + 1.0) * 0.5 * (W - 1)
    y0, x0 = torch.floor(fy), torch.floor(fx)
    wy1, wx1 = fy - y0, fx - x0
    out = 0.0
    flat = fmap.reshape(B, d, H * W)
    for dy, wy in ((0, 1.0 - wy1), (1, wy1)):
        for dx, wx in ((0, 1.0 - wx1), (1, wx1)):
            yy, xx = y0 + dy, x0 + dx
            ok = ((yy >= 0) & (yy <= H - 1) & (xx >= 0) & (xx <= W - 1)).to(fmap.dtype)
            idx = (yy.clamp(0, H - 1) * W + xx.clamp(0, W - 1)).long().reshape(B, 1, -1).expand(B, d, -1)
            val = torch.gather(flat, 2, idx).reshape(B, d, *py.shape[1:])
            out = out + val * (wy * wx * ok)[:, None]
    return out


def das_sample(sd: SD, pre: str, tokens: Tensor, h: int, w: int):
    """DA_sample.forward on tokens [B,Lp,d] reshaped as useB.py:146-148. Returns
    (sampled [B,d,Hk,Wk], raw offsets [B,Hk,Wk])."""
    B, Lp, d = tokens.shape
    fmap = tokens.reshape(B, h, w, d).permute(0, 3, 1, 2)
    o = das_offsets(sd, pre, fmap)
    py, px = das_positions(o)
    return bilinear_sample(fmap, py, px), o


def lam_loss(sd: SD, cfg: RefConfig, patches: Tensor) -> Tensor:
    """AlignmentM.patch_Align (useB.py:128-167)."""
    h, w = cfg.grid
    s = [das_sample(sd, f"AlignM.DAS_{m}.", patches[i], h, w)[0] for i, m in enumerate("rnt")]
    mse = lambda a, b: ((a - b) ** 2).mean()
    return (mse(s[1], s[0]) + mse(s[2], s[0]) + mse(s[2], s[1])) / 3.0


# --------------------------------------------------------------------------- #
# BNNeck / classifier / ReID loss (make_model.py:194-219, layers/*)
# --------------------------------------------------------------------------- #
def bnneck_train(sd: SD, name: str, x: Tensor, eps: float = 1e-5) -> Tensor:
    """nn.BatchNorm1d in train mode on local batch statistics (make_model.py:77,114)."""
    mu = x.mean(0)
    var = ((x - mu) ** 2).mean(0)
    return (x - mu) * torch.rsqrt(var + eps) * sd[name + ".weight"] + sd[name + ".bias"]


def id_loss(score: Tensor, target: Tensor, eps: float) -> Tensor:
    """CrossEntropyLabelSmooth.forward (softmax_loss.py:23-34)."""
    C = score.shape[1]
    logp = torch.log_softmax(score, dim=1)
    t = torch.zeros_like(logp).scatter_(1, target[:, None], 1.0)
    t = (1.0 - eps) * t + eps / C
    return (-t * logp).mean(0).sum()


def pairwise_dist(feat: Tensor) -> Tensor:
    """euclidean_dist (triplet_loss.py:16-31)."""
    sq = (feat * feat).sum(1, keepdim=True)
    return (sq + sq.t() - 2.0 * feat @ feat.t()).clamp(min=1e-12).sqrt()


def batch_hard(dist: Tensor, labels: Tensor):
    """hard_example_mining (triplet_loss.py:51-104) as index sets plus, per anchor, the RELATIVE gap between the chosen
    distance and the runner-up: (pidx, nidx, pgap, ngap).  A gap below the accuracy of the features under test marks a
    near-tie on which a 16-bit device may legitimately mine the other candidate (test aid, see train_loss)."""
    same = labels[:, None] == labels[None, :]
    ap = torch.where(same, dist, torch.full_like(dist, -float("inf")))
    an = torch.where(same, torch.full_like(dist, float("inf")), dist)
    sp, sn = ap.sort(1, descending=True), an.sort(1)
    pgap = (sp.values[:, 0] - sp.values[:, 1]) / sp.values[:, 0] if dist.shape[1] > 1 else torch.ones(len(dist))
    ngap = (sn.values[:, 1] - sn.values[:, 0]) / sn.values[:, 0] if dist.shape[1] > 1 else torch.ones(len(dist))
    return sp.indices[:, 0], sn.indices[:, 0], pgap, ngap


def triplet_soft(feat: Tensor, labels: Tensor, force_mining=None) -> Tensor:
    """TripletLoss() soft-margin with batch-hard mining on un-normalised features
    (triplet_loss.py:16-31,51-104,121-135).  force_mining = (pidx, nidx) (test aid): differentiate under these mined
    indices instead of the function's own arg-max / arg-min."""
    dist = pairwise_dist(feat)
    if force_mining is not None:
        pidx, nidx = (t.to(torch.int64)[:, None] for t in force_mining)
        d_ap, d_an = dist.gather(1, pidx).squeeze(1), dist.gather(1, nidx).squeeze(1)
    else:
        same = labels[:, None] == labels[None, :]
        d_ap = torch.where(same, dist, torch.full_like(dist, -float("inf"))).max(1).values
        d_an = torch.where(same, torch.full_like(dist, float("inf")), dist).min(1).values
    return F.softplus(-(d_an - d_ap)).mean()


def triplet_margin(feat: Tensor, labels: Tensor, margin: float) -> Tensor:
    """TripletLoss(margin): nn.MarginRankingLoss(margin)(d_an, d_ap, 1) = mean(relu(d_ap - d_an + margin)) on the same
    batch-hard distances (triplet_loss.py:107-135, the MODEL.NO_MARGIN=False branch of make_loss.py:66-72)."""
    sq = (feat * feat).sum(1, keepdim=True)
    dist = (sq + sq.t() - 2.0 * feat @ feat.t()).clamp(min=1e-12).sqrt()
    same = labels[:, None] == labels[None, :]
    d_ap = torch.where(same, dist, torch.full_like(dist, -float("inf"))).max(1).values
    d_an = torch.where(same, torch.full_like(dist, float("inf")), dist).min(1).values
    return F.relu(d_ap - d_an + margin).mean()


def reid_loss(cfg: RefConfig, score: Tensor, feat: Tensor, target: Tensor, force_mining=None) -> Tensor:
    """loss_func of make_loss (make_loss.py:109-150), label-smooth on, soft triplet."""
    return (cfg.id_loss_weight * id_loss(score, target, cfg.label_smooth_eps)
            + cfg.triplet_loss_weight * triplet_soft(feat, target, force_mining))


# --------------------------------------------------------------------------- #
# Signal.forward + the train-step loss (make_model.py:148-290, processor.py:173-256)
# --------------------------------------------------------------------------- #
@dataclass
class SignalOut:
    pairs: list = field(default_factory=list)       # [(score, feat), ...] in the reference's tuple order
    loss_area: Optional[Tensor] = None
    patch_loss: Optional[Tensor] = None
    mask: Optional[Tensor] = None
    tie_free: Optional[Tensor] = None
    patches: Optional[Tensor] = None                # [3,B,Lp,d]
    cls: Optional[Tensor] = None                    # [3,B,d]


def backbone3(sd: SD, cfg: RefConfig, img: Dict[str, Tensor], cam_label: Optional[Tensor]):
    """make_model.py:181-183: the same ViT on each modality."""
    ps, cs = [], []
    for m in MODALITIES:
        p, c = vit_forward(sd, cfg, img[m], cam_label)
        ps.append(p)
        cs.append(c)
    return torch.stack(ps), torch.stack(cs)


def signal_forward_train(sd: SD, cfg: RefConfig, img, cam_label, force_mask: Optional[Tensor] = None) -> SignalOut:
    """Signal.forward(training=True) (make_model.py:170-255).  force_mask: see sim_forward."""
    out = SignalOut()
    patches, cls = backbone3(sd, cfg, img, cam_label)
    out.patches, out.cls = patches, cls
    if cfg.direct:
        ori = torch.cat([cls[0], cls[1], cls[2]], dim=-1)
        out.pairs.append((bnneck_train(sd, "bottleneck", ori) @ sd["classifier.weight"].t(), ori))
    else:
        for i, m in enumerate("rnt"):
            sc = bnneck_train(sd, f"bottleneck_{m}", cls[i]) @ sd[f"classifier_{m}.weight"].t()
            out.pairs.append((sc, cls[i]))
    if cfg.use_a:
        vt, out.mask, out.tie_free = sim_forward(sd, cfg, patches, cls, force_mask)
        out.pairs.append((bnneck_train(sd, "bottleneck_var", vt) @ sd["classifier_var.weight"].t(), vt))
    if cfg.use_b:
        out.loss_area = gam_loss(sd, patches)
        if cfg.stage != "CLS":
            out.patch_loss = lam_loss(sd, cfg, patches)
    return out


def signal_forward_infer(sd: SD, cfg: RefConfig, img, cam_label) -> Tensor:
    """Signal.forward(training=False) (make_model.py:257-290): cat(ori, vars_total)."""
    patches, cls = backbone3(sd, cfg, img, cam_label)
    ori = torch.cat([cls[0], cls[1], cls[2]], dim=-1)
    if not cfg.use_a:
        return ori
    vt, _, _ = sim_forward(sd, cfg, patches, cls)
    return torch.cat([ori, vt], dim=-1)


def train_loss(sd: SD, cfg: RefConfig, img, target: Tensor, cam_label: Tensor, force_mask: Optional[Tensor] = None,
               force_mining=None):
    """Total loss of one iteration (processor.py:173-256). Returns (loss, parts dict, SignalOut).  force_mask: see sim_forward.
    force_mining (test aid): per (score, feat) pair either None or the (pidx, nidx) to differentiate under -- the step has two
    kinds of DISCRETE decisions (SIM top-k, batch-hard mining); a device whose 16-bit features resolve a near-tie the other
    way has a different but equally valid gradient, so gradients are compared under the device's decisions and the decisions
    themselves are compared separately, with the size of the tie stated."""
    out = signal_forward_train(sd, cfg, img, cam_label, force_mask)
    parts = {}
    loss = 0.0
    for i, (score, feat) in enumerate(out.pairs):
        li = reid_loss(cfg, score, feat, target, None if force_mining is None else force_mining[i])
        parts[f"reid{i}"] = li.detach()
        loss = loss + li
    if out.loss_area is not None:
        parts["gam"] = out.loss_area.detach()
        loss = loss + cfg.gram_loss_weight * out.loss_area
    if out.patch_loss is not None:
        parts["lam"] = out.patch_loss.detach()
        loss = loss + cfg.pat_loss_weight * out.patch_loss
    return loss, parts, out
