"""Signal model with the reference's construction and call contract
(modeling/make_model.py:22-319 of maxingan2412/Signal):

    model = make_frame(cfg, num_class, camera_num, view_num)
    out = model(x={'RGB','NI','TI'}, label=..., cam_label=..., view_label=..., training=True, sge=stage)

Same parameter names, same training tuples (`sign`, score/feature pairs, loss_area, patch_loss), same
inference features.  The three backbone calls of the reference (make_model.py:181-183) are batched into one
[3B,129,768] problem that runs in hand-written HIP (signal_amd/csrc); SIM, GAM and LAM run in HIP as well.
BNNeck + classifier (make_model.py:194-219) run in HIP too (csrc/reid.hip)."""
from __future__ import annotations

import torch
import torch.nn as nn

from . import params as P
from .hip_engine import BackboneFn, HipPath, SimFn
from .reid_head import bnneck_classifier


class Signal(nn.Module):
    def __init__(self, num_classes, cfg, camera_num, view_num, factory=None):
        super().__init__()
        if "ViT-B-16" not in cfg.MODEL.TRANSFORMER_TYPE:
            raise NotImplementedError(
                f"TRANSFORMER_TYPE={cfg.MODEL.TRANSFORMER_TYPE!r}: only the CLIP 'ViT-B-16' tower used by every "
                "shipped Signal config is on the HIP path (SURVEY.md section 2.1)")
        if cfg.MODEL.PROMPT or cfg.MODEL.ADAPTER or cfg.MODEL.FROZEN:
            raise NotImplementedError("MODEL.PROMPT / ADAPTER / FROZEN branches are outside the hot path")
        self.feat_dim = 512
        self.num_classes = num_classes
        self.cfg = cfg
        self.num_instance = cfg.DATALOADER.NUM_INSTANCE
        self.direct = cfg.MODEL.DIRECT
        self.camera = camera_num
        self.view = view_num
        self.use_A = cfg.MODEL.USE_A
        self.use_B = cfg.MODEL.USE_B
        self.ID_LOSS_TYPE = cfg.MODEL.ID_LOSS_TYPE
        self.image_size = cfg.INPUT.SIZE_TRAIN
        self.h, self.w = self.image_size[0] // 16, self.image_size[1] // 16

        self.clip_vision_encoder = P.BuildTransformerParams(cfg, camera_num, self.feat_dim)

        if self.direct:
            self.bottleneck = nn.BatchNorm1d(3 * self.feat_dim)
            self.bottleneck.bias.requires_grad_(False)
            self.bottleneck.apply(P.weights_init_kaiming)
            self.classifier = nn.Linear(3 * self.feat_dim, num_classes, bias=False)
            self.classifier.apply(P.weights_init_classifier)
        else:
            for m in "rnt":
                cl = nn.Linear(self.feat_dim, num_classes, bias=False)
                cl.apply(P.weights_init_classifier)
                setattr(self, f"classifier_{m}", cl)
                bn = nn.BatchNorm1d(self.feat_dim)
                bn.bias.requires_grad_(False)
                if m != "t":        # the reference leaves bottleneck_t at default init (make_model.py:99-100)
                    bn.apply(P.weights_init_kaiming)
                setattr(self, f"bottleneck_{m}", bn)
        if self.use_A:
            keep_ratio = cfg.MODEL.KEEP_RATIO if cfg.MODEL.FIXED_KEEP_RATIO else None       # make_model.py:107-108
            self.SIM = P.SimParams(self.feat_dim, k=int(cfg.MODEL.TOPK), keep_ratio=keep_ratio)
            self.classifier_var = nn.Linear(3 * self.feat_dim, num_classes, bias=False)
            self.classifier_var.apply(P.weights_init_classifier)
            self.bottleneck_var = nn.BatchNorm1d(3 * self.feat_dim)
            self.bottleneck_var.bias.requires_grad_(False)
            self.bottleneck_var.apply(P.weights_init_kaiming)
        if self.use_B:
            self.AlignM = P.AlignParams(self.feat_dim, self.h, self.w)
        self._hip = None

    # ------------------------------------------------------------------
    @property
    def hip(self) -> HipPath:
        if self._hip is None:
            object.__setattr__(self, "_hip", HipPath(self))
        return self._hip

    def load_param(self, trained_path):
        """make_model.py:125-130; weights_only load (a state_dict holds tensors only)."""
        state_dict = torch.load(trained_path, map_location="cpu", weights_only=True)
        state_dict = {k.replace("module.", "", 1) if k.startswith("module.") else k: v for k, v in state_dict.items()}
        print("Successfully load ckpt!")
        incompatible = self.load_state_dict(state_dict, strict=False)
        print(incompatible)

    def flops(self, shape=(3, 256, 128)):
        """Forward FLOPs of one RGB+NIR+TIR triplet (make_model.py:132-146 measured this with fvcore);
        closed form of SURVEY.md section 8(d)."""
        L, D, Fd, O, H = self.hip.L, self.hip.D, self.hip.F, self.hip.out_dim, self.hip.H
        per_tok = 2 * (3 * D * D + D * D + 2 * D * Fd) + 4 * L * D
        vit = self.hip.layers * L * per_tok + 2 * (L - 1) * D * 3 * self.hip.patch ** 2 + 2 * L * D * O
        total = 3 * vit
        if self.use_A:
            d, Lp = self.feat_dim, L - 1
            total += 2 * 3 * Lp * d * 2 * d + 2 * 3 * (2 * d * d + 2 * d * 2 * d) + 4 * 3 * 3 * Lp * d
        if self.use_B:
            total += 3 * 2 * (L - 1) * self.feat_dim * self.feat_dim * 2
        return float(total)

    # ------------------------------------------------------------------
    def _encode(self, x, cam_label, training=True):
        imgs = [x["RGB"], x["NI"], x["TI"]]
        dev = imgs[0].device
        B = imgs[0].shape[0]
        if B == 0 or any(im.dim() != 4 or im.shape[0] != B for im in imgs):
            raise ValueError(f"expected three [B,3,H,W] image tensors with the same B >= 1, got {[tuple(im.shape) for im in imgs]}")
        self.hip.prepare(dev)
        cam = None
        if self.clip_vision_encoder.cv_embed_sign:
            if cam_label is None:
                raise ValueError("SIE_CAMERA is on: cam_label is required (meta_arch.py:101-103)")
            cam = torch.as_tensor(cam_label, device=dev, dtype=torch.int64).reshape(-1)
            if cam.numel() == 1 and B > 1:
                cam = cam.expand(B)
            cam = cam.contiguous()
            if cam.numel() != B:
                raise ValueError(f"cam_label has {cam.numel()} entries for a batch of {B}")
        hip = self.hip
        vit_params = [hip.flat.byname[n] for n in hip.vit_param_names]
        # training=False never differentiates: take the (small, ping-pong) inference workspace even under grad mode
        hip.grad_mode = torch.is_grad_enabled() and bool(training)
        tokens, cls = BackboneFn.apply(hip, cam, 3, *[im.contiguous().float() for im in imgs], *vit_params)
        tok4 = tokens.view(3, B, hip.L, hip.out_dim)
        # cls is its own (small) output: the ReID heads' gradient comes back as [3B, 512] instead of a zero-filled token tensor
        return tokens, tok4[:, :, 1:], cls.view(3, B, hip.out_dim)

    def _sim(self, tokens, B, training=True):
        hip = self.hip
        hip.grad_mode = torch.is_grad_enabled() and bool(training)
        out, mask = SimFn.apply(hip, B, tokens, *[hip.flat.byname[n] for n in hip.sim_param_names])
        m = mask.unsqueeze(-1)
        self.SIM.token_selection.last_masks = {"RGB": m[0], "NI": m[1], "TI": m[2]}
        return out

    def _align(self, tokens, patches, B, sge):
        from .align import gam_loss, lam_loss
        loss_area = gam_loss(self, tokens, B)
        if sge == "CLS":
            return loss_area, None
        return loss_area, lam_loss(self, tokens, B)

    def forward(self, x, label=None, cam_label=None, view_label=None, return_pattern=1, training=True, sge="CLS"):
        if not training and "cam_label" in x:
            cam_label = x["cam_label"]
        B = x["RGB"].shape[0]
        tokens, patches, cls = self._encode(x, cam_label, training)
        RGB_global, NI_global, TI_global = cls[0], cls[1], cls[2]

        vars_total = self._sim(tokens, B, training) if self.use_A else None
        loss_area = patch_loss = None
        if self.use_B and training:
            # the reference also evaluates AlignM at inference and discards it (make_model.py:277-281);
            # the features returned are identical without that work
            loss_area, patch_loss = self._align(tokens, patches, B, sge)

        if not training:
            ori = torch.cat([RGB_global, NI_global, TI_global], dim=-1)
            return ori if not self.use_A else torch.cat([ori, vars_total], dim=-1)

        if self.use_A:
            vars_score = bnneck_classifier(self.bottleneck_var, self.classifier_var, vars_total, self.hip)
        if self.direct:
            ori = torch.cat([RGB_global, NI_global, TI_global], dim=-1)
            ori_score = bnneck_classifier(self.bottleneck, self.classifier, ori, self.hip)
            head = (ori_score, ori)
        else:
            head = (bnneck_classifier(self.bottleneck_r, self.classifier_r, RGB_global, self.hip), RGB_global,
                    bnneck_classifier(self.bottleneck_n, self.classifier_n, NI_global, self.hip), NI_global,
                    bnneck_classifier(self.bottleneck_t, self.classifier_t, TI_global, self.hip), TI_global)
        if not self.use_A:
            return (1, *head)
        if not self.use_B:
            return (2, *head, vars_score, vars_total)
        if sge == "CLS":
            return (3, *head, vars_score, vars_total, loss_area)
        return (3, *head, vars_score, vars_total, loss_area, patch_loss)


def make_frame(cfg, num_class, camera_num, view_num=0):
    """modeling/make_model.py:304-319."""
    model = Signal(num_class, cfg, camera_num, view_num, None)
    print("===========Building Signal===========")
    return model
