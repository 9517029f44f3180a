"""Weights-only loader for the pretrained CLIP ViT-B/16 VISUAL tower (the reference's load_clip_to_cpu +
clip.build_model, modeling/make_model_clipreid.py:177-197 and modeling/clip/model.py:651-729).

The reference opens '../ViT-B-16.pt' with torch.jit.load (a TorchScript archive executes code from the file) and falls
back to an unrestricted torch.load.  Here the file is read with torch.load(weights_only=True) ONLY: it must be a plain
state_dict (the `model.state_dict()` of OpenAI CLIP, e.g. exported once with
`torch.save(torch.jit.load('ViT-B-16.pt').state_dict(), 'ViT-B-16.sd.pt')` on a machine the user trusts); a TorchScript
archive is refused with a message saying so.  Only the `visual.*` entries are used -- the text tower is dead in Signal.
The 197-token positional embedding is resized to the model's grid exactly as clip/model.py:712-729 does: CLS row kept,
the 14x14 grid bilinearly interpolated (align_corners=False) to h x w."""
from __future__ import annotations

import math
import zipfile

import torch
import torch.nn.functional as F

VISUAL = "visual."


def resize_pos_embed(posemb: torch.Tensor, height: int, width: int) -> torch.Tensor:
    """[1 + g*g, D] -> [1 + height*width, D] (clip/model.py:712-729)."""
    tok, grid = posemb[:1], posemb[1:]
    gs = int(math.sqrt(len(grid)))
    if gs * gs != len(grid):
        raise ValueError(f"positional embedding has {len(grid)} grid tokens: not a square grid")
    if (gs, gs) == (height, width):
        return posemb.clone()
    grid = grid.reshape(1, gs, gs, -1).permute(0, 3, 1, 2).float()
    grid = F.interpolate(grid, size=(height, width), mode="bilinear")
    grid = grid.permute(0, 2, 3, 1).reshape(height * width, -1)
    return torch.cat([tok.float(), grid], dim=0)


def _refuse_torchscript(path: str):
    try:
        with zipfile.ZipFile(path) as z:
            names = z.namelist()
    except (zipfile.BadZipFile, OSError):
        return
    if any(n.endswith("constants.pkl") or "/code/" in n for n in names):
        raise ValueError(
            f"{path} is a TorchScript archive (the released CLIP 'ViT-B-16.pt'): signal_amd never calls torch.jit.load. "
            "Export its state_dict once (torch.save(torch.jit.load(p).state_dict(), out)) on a machine you trust and pass that file.")


def load_clip_visual_state_dict(path: str) -> dict:
    _refuse_torchscript(path)
    sd = torch.load(path, map_location="cpu", weights_only=True)
    if not isinstance(sd, dict):
        raise ValueError(f"{path}: expected a state_dict, got {type(sd).__name__}")
    if "state_dict" in sd and isinstance(sd["state_dict"], dict):
        sd = sd["state_dict"]
    vis = {k[len(VISUAL):]: v for k, v in sd.items() if k.startswith(VISUAL) and torch.is_tensor(v)}
    if "proj" not in vis or "conv1.weight" not in vis:
        raise ValueError(f"{path}: no CLIP ViT visual tower ('visual.proj' / 'visual.conv1.weight') in this state_dict")
    return vis


def load_clip_visual(model, path: str, verbose: bool = True):
    """Copy the CLIP visual tower into model.clip_vision_encoder.base (shapes checked; positional embedding resized).
    Returns the (missing, unexpected) key lists like load_state_dict."""
    base = model.clip_vision_encoder.base
    vis = load_clip_visual_state_dict(path)
    h, w = base.h_resolution, base.w_resolution
    old = vis["positional_embedding"]
    vis["positional_embedding"] = resize_pos_embed(old, h, w)
    if verbose:
        print("Resized position embedding: %s to %s" % (tuple(old.shape), tuple(vis["positional_embedding"].shape)))
    own = base.state_dict()
    bad = [k for k, v in vis.items() if k in own and tuple(v.shape) != tuple(own[k].shape)]
    if bad:
        raise ValueError("CLIP checkpoint does not match ViT-B/16 at this geometry: " +
                         ", ".join(f"{k} {tuple(vis[k].shape)} vs {tuple(own[k].shape)}" for k in bad))
    out = base.load_state_dict({k: v.float() for k, v in vis.items()}, strict=False)
    if verbose:
        print("Successfully load ckpt!")
        print(out)
    return out
