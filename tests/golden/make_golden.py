#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE.

Runs only in the build container (needs /root/reference); never at test time and
never on the GPU box.  It imports the reference's own hot-path leaf modules
(modeling/clip/model.py, modeling/AddModule/{useA,useB,DAS}.py, utils/volume.py,
layers/{triplet_loss,softmax_loss}.py) through stub parent packages (their
__init__ files pull in timm/fvcore/yacs, which are absent; the leaves need only
torch/einops), loads PCG64-seeded weights into them and records inputs' seeds and
the reference's outputs.  Nothing of the reference's source is copied: fixtures
are numbers only.

    python tests/golden/make_golden.py            # rewrites tests/golden/*.npz
"""
import importlib
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("SIGNAL_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)

from oracle import signal_ref as O  # noqa: E402  (only for init_state_dict / PRNG helpers / configs)


def _import_reference():
    sys.path.insert(0, REF)
    for name, sub in [("modeling", "modeling"), ("modeling.clip", "modeling/clip"),
                      ("modeling.AddModule", "modeling/AddModule"), ("utils", "utils"),
                      ("layers", "layers")]:
        mod = types.ModuleType(name)
        mod.__path__ = [os.path.join(REF, sub)]
        sys.modules[name] = mod
    R = types.SimpleNamespace()
    R.clip = importlib.import_module("modeling.clip.model")
    R.useA = importlib.import_module("modeling.AddModule.useA")
    R.useB = importlib.import_module("modeling.AddModule.useB")
    R.triplet = importlib.import_module("layers.triplet_loss")
    R.softmax = importlib.import_module("layers.softmax_loss")
    return R


def _cfg_ns():
    return types.SimpleNamespace(MODEL=types.SimpleNamespace(PROMPT=False, ADAPTER=False))


class RefSignal(torch.nn.Module):
    """The reference leaf modules composed exactly as make_model.py:35-122,170-255 and
    meta_arch.py:84-112 compose them (the originals need timm/fvcore/a CLIP checkpoint
    and .to('cuda'))."""

    def __init__(self, R, cfg: O.RefConfig):
        super().__init__()
        self.cfg = cfg
        h, w = cfg.grid
        enc = torch.nn.Module()
        enc.base = R.clip.VisionTransformer(h, w, cfg.patch, cfg.patch, cfg.width, cfg.layers,
                                            cfg.heads, cfg.out_dim, _cfg_ns())
        if cfg.sie_camera:
            enc.cv_embed = torch.nn.Parameter(torch.zeros(cfg.camera_num, 1, cfg.width))
        self.clip_vision_encoder = enc
        d, C = cfg.out_dim, cfg.num_classes
        if cfg.direct:
            self.bottleneck = torch.nn.BatchNorm1d(3 * d)
            self.classifier = torch.nn.Linear(3 * d, C, bias=False)
        else:
            for m in "rnt":
                setattr(self, f"bottleneck_{m}", torch.nn.BatchNorm1d(d))
                setattr(self, f"classifier_{m}", torch.nn.Linear(d, C, bias=False))
        if cfg.use_a:
            self.SIM = R.useA.Select_Interactive_Module(d, k=cfg.topk)
            self.bottleneck_var = torch.nn.BatchNorm1d(3 * d)
            self.classifier_var = torch.nn.Linear(3 * d, C, bias=False)
        if cfg.use_b:
            self.AlignM = R.useB.AlignmentM(d, h, w)

    def encode(self, x, cam):
        enc = self.clip_vision_encoder
        cv = self.cfg.sie_coe * enc.cv_embed[cam] if self.cfg.sie_camera else None
        y = enc.base(x, cv, None)
        return y[:, 1:], y[:, 0]

    def forward(self, img, cam):
        cfg = self.cfg
        (rp, rg), (npp, ng), (tp, tg) = (self.encode(img[m], cam) for m in O.MODALITIES)
        out = []
        if cfg.direct:
            ori = torch.cat([rg, ng, tg], dim=-1)
            out.append((self.classifier(self.bottleneck(ori)), ori))
        else:
            for m, g in zip("rnt", (rg, ng, tg)):
                out.append((getattr(self, f"classifier_{m}")(getattr(self, f"bottleneck_{m}")(g)), g))
        if cfg.use_a:
            vt = self.SIM(rp, npp, tp, rg, ng, tg)
            out.append((self.classifier_var(self.bottleneck_var(vt)), vt))
        la, pl = self.AlignM(rp, npp, tp, stage="together_CLS_Patch")
        return out, la, pl, (rp, npp, tp), (rg, ng, tg)


def _load(mod, sd):
    missing, unexpected = mod.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    miss = [k for k in missing if "num_batches_tracked" not in k]
    assert not miss, miss


def head_features(cfg, B, seed, scale=1.0):
    g = O._rng(seed)
    Lp = cfg.tokens - 1
    patches = O.randn(g, 3, B, Lp, cfg.out_dim, std=scale)
    cls = O.randn(g, 3, B, cfg.out_dim, std=scale)
    return patches, cls


def save(name, **arrs):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **{k: (v.detach().numpy() if isinstance(v, torch.Tensor) else np.asarray(v))
                                 for k, v in arrs.items()})
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.0f} KiB)")


def main():
    torch.manual_seed(0)
    torch.set_grad_enabled(True)
    R = _import_reference()

    # ---- G1: ViT.  (a) reduced-width twin, full output; (b) real ViT-B/16, B=2, slices ----
    tw = O.RefConfig(width=64, heads=2, layers=2, out_dim=32, num_classes=5, camera_num=3,
                     use_a=False, use_b=False)
    sd = O.init_state_dict(tw, seed=11)
    ref = RefSignal(R, tw)
    _load(ref, sd)
    img, vid, cam = O.synthetic_batch(tw, 3, seed=12)
    with torch.no_grad():
        p, c = ref.encode(img["RGB"], cam)
    save("g1_vit_twin", seed_w=11, seed_x=12, patches=p, cls=c)

    full = O.rgbnt201_config(use_a=False, use_b=False)
    sd = O.init_state_dict(full, seed=21)
    ref = RefSignal(R, full)
    _load(ref, sd)
    img, vid, cam = O.synthetic_batch(full, 2, seed=22)
    with torch.no_grad():
        p, c = ref.encode(img["NI"], cam)
        # one block in isolation on the ln_pre output (reference layout is [L,N,D])
        x0 = O.vit_embed(sd, full, img["NI"], full.sie_coe * sd["clip_vision_encoder.cv_embed"][cam])
        blk = ref.clip_vision_encoder.base.transformer.resblocks[0]
        y0 = blk(x0.permute(1, 0, 2), None, 0, None, prompt_sign=False, adapter_sign=False).permute(1, 0, 2)
    save("g1_vit_b16", seed_w=21, seed_x=22, cls=c, patches_head=p[:, :4], patches_norm=p.norm(dim=-1),
         block0_in_norm=x0.norm(dim=-1), block0_out_rows=y0[:, :3], block0_out_norm=y0.norm(dim=-1))

    # ---- G2/G3: SIM select + interact, TOPK in {80,112,64}; moderate and saturated scales ----
    for topk, scale, tag in [(80, 1.0, "k80"), (112, 1.0, "k112"), (64, 1.0, "k64"), (80, 40.0, "k80_sat")]:
        cfg = O.rgbnt201_config(topk=topk)
        sd = O.init_state_dict(cfg, seed=31)
        sim = R.useA.Select_Interactive_Module(cfg.out_dim, k=topk)
        _load(sim, {k[len("SIM."):]: v for k, v in sd.items() if k.startswith("SIM.")})
        patches, cls = head_features(cfg, 8, seed=32, scale=scale)
        with torch.no_grad():
            out = sim(patches[0], patches[1], patches[2], cls[0], cls[1], cls[2])
        masks = torch.stack([sim.token_selection.last_masks[m][..., 0] for m in O.MODALITIES]).to(torch.int8)
        _, tie_free = O.sim_select(sd, patches, cls, topk)
        save(f"g2_sim_{tag}", seed_w=31, seed_x=32, scale=scale, topk=topk, masks=masks,
             tie_free=tie_free.to(torch.int8), interact=out)

    # ---- G2b: the exact keep-ratio branch (useA.py:253-316; MODEL.FIXED_KEEP_RATIO): TOPK 80 leaves more than
    # int(0.75 * 128) = 96 tokens selected (trim), TOPK 24 fewer (grow) ----
    for topk, keep, tag in [(80, 0.75, "k80_keep75"), (24, 0.75, "k24_keep75")]:
        cfg = O.rgbnt201_config(topk=topk, keep_ratio=keep)
        sd = O.init_state_dict(cfg, seed=31)
        sim = R.useA.Select_Interactive_Module(cfg.out_dim, k=topk, keep_ratio=keep)
        _load(sim, {k[len("SIM."):]: v for k, v in sd.items() if k.startswith("SIM.")})
        patches, cls = head_features(cfg, 8, seed=33)
        with torch.no_grad():
            out = sim(patches[0], patches[1], patches[2], cls[0], cls[1], cls[2])
        masks = torch.stack([sim.token_selection.last_masks[m][..., 0] for m in O.MODALITIES]).to(torch.int8)
        before, _ = O.sim_select(sd, patches, cls, topk)
        _, tie_free = O.sim_select(sd, patches, cls, topk, keep)
        save(f"g2_sim_{tag}", seed_w=31, seed_x=33, scale=1.0, topk=topk, keep_ratio=keep, masks=masks,
             count_before=before.sum(-1).to(torch.int32), tie_free=tie_free.to(torch.int8), interact=out)

    # ---- G4: GAM loss and input gradients (regular + near-degenerate batch) ----
    cfg = O.rgbnt201_config()
    sd = O.init_state_dict(cfg, seed=41)
    align = R.useB.AlignmentM(cfg.out_dim, *cfg.grid)
    _load(align, {k[len("AlignM."):]: v for k, v in sd.items() if k.startswith("AlignM.")})
    for tag, mix in [("regular", 0.0), ("aligned", 0.97)]:
        patches, _ = head_features(cfg, 8, seed=42)
        if mix:
            patches = torch.stack([patches[0], mix * patches[0] + (1 - mix) * patches[1],
                                   mix * patches[0] + (1 - mix) * patches[2]])
        pr = [patches[m].clone().requires_grad_(True) for m in range(3)]
        loss = align.Cls_Align(*pr)
        loss.backward()
        save(f"g4_gam_{tag}", seed_w=41, seed_x=42, mix=mix, loss=loss,
             grad_rows=torch.stack([p.grad[:, :2] for p in pr]),
             grad_norm=torch.stack([p.grad.norm() for p in pr]),
             temp_grad=align.contra_temp.grad)
        align.zero_grad()

    # ---- G5: DAS / LAM on both grids ----
    for tag, cfg in [("16x8", O.rgbnt201_config()), ("8x16", O.rgbnt100_config())]:
        sd = O.init_state_dict(cfg, seed=51)
        align = R.useB.AlignmentM(cfg.out_dim, *cfg.grid)
        _load(align, {k[len("AlignM."):]: v for k, v in sd.items() if k.startswith("AlignM.")})
        patches, _ = head_features(cfg, 4, seed=52)
        h, w = cfg.grid
        pr = [patches[m].clone().requires_grad_(True) for m in range(3)]
        fm = pr[0].reshape(4, h, w, -1).permute(0, 3, 1, 2)
        das = align.DAS_r
        q = das.proj_q(fm)
        off = das.conv_offset(q)                                            # [B,1,Hk,Wk]
        sampled = das(fm)
        loss = align.patch_Align(*pr)
        loss.backward()
        save(f"g5_lam_{tag}", seed_w=51, seed_x=52, offsets=off[:, 0], sampled=sampled, loss=loss,
             grad_rows=torch.stack([p.grad[:, :2] for p in pr]),
             grad_norm=torch.stack([p.grad.norm() for p in pr]),
             w4_grad=das.conv_offset[4].weight.grad.reshape(-1))

    # ---- G6: ID + soft-margin triplet on a P x K batch ----
    g = O._rng(61)
    score = O.randn(g, 16, 171, std=2.0).requires_grad_(True)
    feat = O.randn(g, 16, 1536, std=1.0).requires_grad_(True)
    target = torch.arange(16) // 4 + 7
    xent = R.softmax.CrossEntropyLabelSmooth(num_classes=171, use_gpu=False)
    tri = R.triplet.TripletLoss()
    idl = xent(score, target)
    trl = tri(feat, target)[0]
    (0.25 * idl + 1.0 * trl).backward()
    save("g6_reid", seed=61, id_loss=idl, tri_loss=trl, dscore_rows=score.grad[:2], dfeat_norm=feat.grad.norm(dim=1))

    # ---- G7: one full train-step loss at real size, B=8 (2 ids x 4), both dataset configs ----
    for tag, cfg in [("rgbnt201", O.rgbnt201_config(num_instance=4)),
                     ("rgbnt100", O.rgbnt100_config(num_instance=4))]:
        sd = O.init_state_dict(cfg, seed=1234, head_scale=30.0)
        ref = RefSignal(R, cfg)
        _load(ref, sd)
        ref.train()
        img, vid, cam = O.synthetic_batch(cfg, 8, seed=1234)
        out, la, pl, pats, glbs = ref(img, cam)
        xent = R.softmax.CrossEntropyLabelSmooth(num_classes=cfg.num_classes, use_gpu=False)
        tri = R.triplet.TripletLoss()
        parts, loss = {}, 0.0
        for i, (sc, ft) in enumerate(out):
            li = cfg.id_loss_weight * xent(sc, vid) + cfg.triplet_loss_weight * tri(ft, vid)[0]
            parts[f"reid{i}"] = li.detach()
            loss = loss + li
        loss = loss + cfg.gram_loss_weight * la + cfg.pat_loss_weight * pl
        loss.backward()
        gn = {}
        for k, p in ref.named_parameters():
            if p.grad is not None:
                gn[k] = float(p.grad.norm())
        keys = sorted(gn)
        masks = torch.stack([ref.SIM.token_selection.last_masks[m][..., 0] for m in O.MODALITIES]).to(torch.int8)
        save(f"g7_step_{tag}", seed=1234, loss=loss.detach(), gam=la.detach(), lam=pl.detach(),
             **{k: v for k, v in parts.items()},
             cls=torch.stack([g_.detach() for g_ in glbs]),
             patches_norm=torch.stack([p_.detach().norm(dim=-1) for p_ in pats]),
             vars_total=out[-1][1].detach(), masks=masks,
             grad_keys=np.array(keys), grad_norms=np.array([gn[k] for k in keys], dtype=np.float64))


if __name__ == "__main__":
    main()
