"""Parameter tree of the Signal model with the reference's state_dict keys (SURVEY.md section 8(b):
parameter names are API -- make_optimizer assigns learning rates by substring, checkpoints are loaded
by name).  The nn.Module classes below are CONTAINERS: they own Parameters/buffers under the reference's
names and initialise them the way the reference does when no checkpoint is loaded;
the arithmetic runs in the HIP path (hip_engine.py), never through these modules' forward()."""
from __future__ import annotations

import math

import torch
import torch.nn as nn


def _trunc_normal_(t: torch.Tensor, std: float):
    # reference: trunc_normal_(std=.02) with ABSOLUTE cut-offs +-2, i.e. effectively N(0, std^2)
    with torch.no_grad():
        t.normal_(0.0, std).clamp_(-2.0, 2.0)
    return t


class _Holder(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover - containers are never called
        raise RuntimeError("parameter container: the computation runs in signal_amd's HIP path")


class AttnParams(_Holder):
    """nn.MultiheadAttention's parameter names (in_proj_weight/in_proj_bias/out_proj.*)."""

    def __init__(self, dim: int):
        super().__init__()
        self.in_proj_weight = nn.Parameter(torch.empty(3 * dim, dim))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * dim))
        self.out_proj = nn.Linear(dim, dim)
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.zeros_(self.out_proj.bias)


class BlockParams(_Holder):
    """clip/model.py:168-221 ResidualAttentionBlock parameters."""

    def __init__(self, dim: int):
        super().__init__()
        self.attn = AttnParams(dim)
        self.ln_1 = nn.LayerNorm(dim)
        self.mlp = nn.Sequential()
        self.mlp.add_module("c_fc", nn.Linear(dim, 4 * dim))
        self.mlp.add_module("c_proj", nn.Linear(4 * dim, dim))
        self.ln_2 = nn.LayerNorm(dim)
        for lin in (self.attn.out_proj, self.mlp.c_fc, self.mlp.c_proj):
            _trunc_normal_(lin.weight, 0.02)
            nn.init.zeros_(lin.bias)


class TransformerParams(_Holder):
    def __init__(self, dim: int, layers: int):
        super().__init__()
        self.resblocks = nn.Sequential(*[BlockParams(dim) for _ in range(layers)])


class VisionTransformerParams(_Holder):
    """clip/model.py:419-445 VisionTransformer parameters."""

    def __init__(self, h_res: int, w_res: int, patch: int, width: int, layers: int, heads: int, out_dim: int):
        super().__init__()
        self.h_resolution, self.w_resolution, self.patch, self.width = h_res, w_res, patch, width
        self.layers, self.heads, self.output_dim = layers, heads, out_dim
        scale = width ** -0.5
        self.conv1 = nn.Conv2d(3, width, kernel_size=patch, stride=patch, bias=False)
        self.class_embedding = nn.Parameter(scale * torch.randn(width))
        self.positional_embedding = nn.Parameter(scale * torch.randn(h_res * w_res + 1, width))
        self.ln_pre = nn.LayerNorm(width)
        self.transformer = TransformerParams(width, layers)
        self.ln_post = nn.LayerNorm(width)
        self.proj = nn.Parameter(scale * torch.randn(width, out_dim))


class BuildTransformerParams(_Holder):
    """modeling/meta_arch.py:34-94: `base` (CLIP visual tower) + `cv_embed` camera embedding."""

    def __init__(self, cfg, camera_num: int, feat_dim: int):
        super().__init__()
        h = cfg.INPUT.SIZE_TRAIN[0] // cfg.MODEL.STRIDE_SIZE[0]
        w = cfg.INPUT.SIZE_TRAIN[1] // cfg.MODEL.STRIDE_SIZE[1]
        self.in_planes = feat_dim
        self.sie_xishu = cfg.MODEL.SIE_COE
        self.cv_embed_sign = bool(cfg.MODEL.SIE_CAMERA)
        self.camera_num = camera_num if cfg.MODEL.SIE_CAMERA else 0
        self.base = VisionTransformerParams(h, w, cfg.MODEL.STRIDE_SIZE[0], 768, 12, 12, feat_dim)
        if cfg.MODEL.SIE_CAMERA:
            self.cv_embed = nn.Parameter(_trunc_normal_(torch.zeros(camera_num, 1, 768), 0.02))

    def load_param(self, trained_path):
        """meta_arch.py:114-118 (strips DDP's 'module.' prefix)."""
        sd = torch.load(trained_path, map_location="cpu", weights_only=True)
        own = self.state_dict()
        for k, v in sd.items():
            own[k.replace("module.", "")].copy_(v)


class TokenSelectionParams(_Holder):
    """useA.py:33-48; W_v exists but is dead in the reference."""

    def __init__(self, dim: int, k: int, keep_ratio=None):
        super().__init__()
        self.dim, self.k1, self.k2, self.keep_ratio = dim, k, 2 * k, keep_ratio
        self.W_q = nn.Linear(dim, dim)
        self.W_k = nn.Linear(dim, dim)
        self.W_v = nn.Linear(dim, dim)
        self.last_masks = None  # {'RGB','NI','TI'} -> [B,L,1] float, set by every forward (useA.py:323)


class ModalInteractiveParams(_Holder):
    """useA.py:340-362."""

    def __init__(self, dim: int, num_heads: int = 8):
        super().__init__()
        self.dim, self.num_heads = dim, num_heads
        self.cross_attn = AttnParams(dim)
        # nn.MultiheadAttention leaves out_proj.weight at nn.Linear's default init
        nn.init.kaiming_uniform_(self.cross_attn.out_proj.weight, a=math.sqrt(5))
        self.ffn = nn.Sequential(nn.Linear(dim, 2 * dim), nn.GELU(), nn.Linear(2 * dim, dim))
        self.norm1 = nn.LayerNorm(dim)
        self.norm2 = nn.LayerNorm(dim)


class SimParams(_Holder):
    """useA.py:442-452 Select_Interactive_Module."""

    def __init__(self, dim: int, k: int, keep_ratio=None):
        super().__init__()
        self.token_selection = TokenSelectionParams(dim, k, keep_ratio)
        self.modal_interactive = ModalInteractiveParams(dim, 8)


class DASParams(_Holder):
    """DAS.py:30-72 DA_sample (n_heads=1, 512 channels, groups=1, stride 4, range factor 2, ksize 4)."""

    def __init__(self, channels: int = 512, stride: int = 4, ksize: int = 4, offset_range_factor: float = 2):
        super().__init__()
        self.nc, self.stride, self.kk, self.offset_range_factor = channels, stride, ksize, offset_range_factor
        self.conv_offset = nn.Sequential(
            nn.Conv2d(channels, channels, 1, 1, 0), nn.GELU(),
            nn.Conv2d(channels, channels, ksize, stride, 0, groups=channels), nn.GELU(),
            nn.Conv2d(channels, 1, 1, 1, 0, bias=False))
        self.proj_q = nn.Conv2d(channels, channels, 1, 1, 0)


class AlignParams(_Holder):
    """useB.py:44-74 AlignmentM."""

    def __init__(self, feat_dim: int, H: int, W: int):
        super().__init__()
        self.feat_dim, self.h, self.w = feat_dim, H, W
        self.contra_temp = nn.Parameter(torch.tensor(0.07))
        self.DAS_r = DASParams(feat_dim)
        self.DAS_n = DASParams(feat_dim)
        self.DAS_t = DASParams(feat_dim)


def weights_init_kaiming(m):
    """meta_arch.py:9-22 (only the BatchNorm branch is reached by Signal)."""
    if isinstance(m, nn.BatchNorm1d) and m.affine:
        nn.init.constant_(m.weight, 1.0)
        nn.init.constant_(m.bias, 0.0)


def weights_init_classifier(m):
    """meta_arch.py:25-31."""
    if isinstance(m, nn.Linear):
        nn.init.normal_(m.weight, std=0.001)
        if m.bias is not None:
            nn.init.constant_(m.bias, 0.0)
