"""CPU diagnostic (no GPU): where does the bf16 train step's per-parameter gradient error come from?

The oracle's ViT is re-run with the DEVICE's 16-bit rounding points emulated (operands of every GEMM, the 16-bit
backward signals dh / dqkv / dattn / du / dx_b, 16-bit weights), everything else f32 -- and its gradients are compared with
the plain fp32 oracle per parameter.  Then the same with a few LayerNorm outputs moved by ONE 16-bit ulp (the round-2
two-rows-per-wave LayerNorm differed from the shipped one by exactly that), to see which quantity moves: the SIM
selection, the batch-hard triplet indices, the smallest Gram determinant, or only rounding noise.

    python tests/probes/bf16_emulation.py [--dtype bf16|fp16] [--tag rgbnt201] [--ulp-frac 0.003]

Test infrastructure (imports oracle/); never imported by the product."""
import argparse
import math
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import signal_ref as O  # noqa: E402

DT = torch.bfloat16
PERTURB = {"frac": 0.0, "gen": None}


def _r(x):
    return x.to(DT).to(torch.float32)


class _RQ(torch.autograd.Function):          # 16-bit value forward, 16-bit gradient backward
    @staticmethod
    def forward(ctx, x, fwd, bwd):
        ctx.bwd = bwd
        return _r(x) if fwd else x.clone()

    @staticmethod
    def backward(ctx, g):
        return (_r(g) if ctx.bwd else g), None, None


def rq(x): return _RQ.apply(x, True, True)
def rf(x): return _RQ.apply(x, True, False)
def rb(x): return _RQ.apply(x, False, True)


def ln16(x, w, b):
    """LayerNorm in f32, output rounded to 16 bits; optionally a fraction of the outputs moved by one ulp."""
    y = O.layer_norm(x, w, b)
    if PERTURB["frac"] > 0:
        with torch.no_grad():
            y16 = y.to(DT)
            pick = torch.from_numpy(PERTURB["gen"].random(y.shape) < PERTURB["frac"])
            up = torch.from_numpy(PERTURB["gen"].random(y.shape) < 0.5)
            bits = y16.view(torch.int16).clone()
            bits[pick & up] += 1
            bits[pick & ~up] -= 1
            delta = bits.view(DT).to(torch.float32) - y16.to(torch.float32)
        return _RQ.apply(y, True, True) + delta
    return rq(y)


class _Gelu16(torch.autograd.Function):
    """g = r16(pre * sigmoid(1.702 pre)); backward du = r16(dg * r16(gelu'(pre))) -- the device saves the derivative in 16 bits."""
    @staticmethod
    def forward(ctx, pre):
        s = torch.sigmoid(1.702 * pre)
        ctx.save_for_backward(_r(s * (1 + 1.702 * pre * (1 - s))))
        return _r(pre * s)

    @staticmethod
    def backward(ctx, dg):
        (u,) = ctx.saved_tensors
        return _r(dg * u)


class _Attn16(torch.autograd.Function):
    """The device's attention core: P (f32, from the exact scores of the 16-bit q, k) is a 16-bit MFMA operand of P.V; the
    backward recomputes P and takes delta = rowsum(dO * O) from the SAVED 16-bit output; dS is a 16-bit operand of dQ / dK."""
    @staticmethod
    def forward(ctx, q, k, v):
        sc = 1.0 / math.sqrt(q.shape[-1])
        p = torch.softmax((q @ k.transpose(-1, -2)) * sc, dim=-1)
        o = _r(_r(p) @ v)
        ctx.save_for_backward(q, k, v, o)
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v, o = ctx.saved_tensors
        sc = 1.0 / math.sqrt(q.shape[-1])
        do = _r(do)
        p = torch.softmax((q @ k.transpose(-1, -2)) * sc, dim=-1)
        delta = (do * o).sum(-1, keepdim=True) if DELTA_FROM_O else (p * (do @ v.transpose(-1, -2))).sum(-1, keepdim=True)
        ds = _r(p * (do @ v.transpose(-1, -2) - delta))
        return _r(ds @ k * sc), _r(ds.transpose(-1, -2) @ q * sc), _r(_r(p).transpose(-1, -2) @ do)


DELTA_FROM_O = True
ATTN_DEVICE = False


def block16(sd, pre, x, heads):
    S, L, D = x.shape
    hd = D // heads
    h1 = ln16(x, sd[pre + "ln_1.weight"], sd[pre + "ln_1.bias"])
    qkv = rq(h1 @ rf(sd[pre + "attn.in_proj_weight"]).t() + sd[pre + "attn.in_proj_bias"])
    q, k, v = (t.reshape(S, L, heads, hd).transpose(1, 2) for t in qkv.split(D, dim=-1))
    if ATTN_DEVICE:
        o = _Attn16.apply(q, k, v).transpose(1, 2).reshape(S, L, D)
    else:
        s = rb((q @ k.transpose(-1, -2)) / math.sqrt(hd))          # dS is a 16-bit MFMA operand in the backward
        p = rf(torch.softmax(s, dim=-1))                            # P is a 16-bit MFMA operand of P.V
        o = rq((p @ v).transpose(1, 2).reshape(S, L, D))
    x = x + rb(o @ rf(sd[pre + "attn.out_proj.weight"]).t() + sd[pre + "attn.out_proj.bias"])
    h2 = ln16(x, sd[pre + "ln_2.weight"], sd[pre + "ln_2.bias"])
    g = _Gelu16.apply(h2 @ rf(sd[pre + "mlp.c_fc.weight"]).t() + sd[pre + "mlp.c_fc.bias"])
    return x + rb(g @ rf(sd[pre + "mlp.c_proj.weight"]).t() + sd[pre + "mlp.c_proj.bias"])


def vit16(sd, cfg, img, cam, hidden=None):
    base = "clip_vision_encoder.base."
    B, p = img.shape[0], cfg.patch
    h, w = cfg.grid
    cv = cfg.sie_coe * sd["clip_vision_encoder.cv_embed"][cam] if cfg.sie_camera else None
    patches = img.reshape(B, 3, h, p, w, p).permute(0, 2, 4, 1, 3, 5).reshape(B, h * w, 3 * p * p)
    tok = rb(rf(patches) @ rf(sd[base + "conv1.weight"].reshape(cfg.width, -1)).t())
    cls = sd[base + "class_embedding"].expand(B, 1, cfg.width)
    if cv is not None:
        cls = cls + cv.reshape(B, 1, cfg.width)
    x = torch.cat([cls, tok], dim=1) + sd[base + "positional_embedding"]
    x = O.layer_norm(x, sd[base + "ln_pre.weight"], sd[base + "ln_pre.bias"])
    for i in range(cfg.layers):
        if hidden is not None:
            x.retain_grad(); hidden.append(x)
        x = block16(sd, f"{base}transformer.resblocks.{i}.", x, cfg.heads)
    if hidden is not None:
        x.retain_grad(); hidden.append(x)
    hp = rq(O.layer_norm(x, sd[base + "ln_post.weight"], sd[base + "ln_post.bias"]))
    t = rb(hp @ rf(sd[base + "proj"]))
    return t[:, 1:], t[:, 0]


def run(sd0, cfg, img, vid, cam, emulate, hidden=None):
    sd = {k: v.clone() for k, v in sd0.items()}
    for k, v in sd.items():
        if v.is_floating_point() and "running_" not in k:
            v.requires_grad_(True)
    keep = O.vit_forward
    if emulate:
        O.vit_forward = lambda s, c, im, cl, return_hidden=False: vit16(s, c, im, cl, hidden)
    elif hidden is not None:
        def vf(s, c, im, cl, return_hidden=False):
            a, b, hs = keep(s, c, im, cl, True)
            for t in hs:
                t.retain_grad()
            hidden.extend(hs)
            return a, b
        O.vit_forward = vf
    try:
        loss, parts, out = O.train_loss(sd, cfg, img, vid, cam)
        out.patches.retain_grad(); out.cls.retain_grad()
        loss.backward()
    finally:
        O.vit_forward = keep
    grads = {k: v.grad for k, v in sd.items() if v.requires_grad and v.grad is not None}
    # discrete decisions of the step
    disc = {"mask": out.mask.clone()}
    for i, (score, feat) in enumerate(out.pairs):
        f = feat.detach()
        sq = (f * f).sum(1, keepdim=True)
        dist = (sq + sq.t() - 2 * f @ f.t()).clamp(min=1e-12).sqrt()
        same = vid[:, None] == vid[None, :]
        ap = torch.where(same, dist, torch.full_like(dist, -math.inf))
        an = torch.where(same, torch.full_like(dist, math.inf), dist)
        disc[f"pidx{i}"], disc[f"nidx{i}"] = ap.argmax(1), an.argmin(1)
        sp, sn = ap.sort(1, descending=True).values, an.sort(1).values
        disc[f"pgap{i}"] = ((sp[:, 0] - sp[:, 1]) / sp[:, 0]).min().item()
        disc[f"ngap{i}"] = ((sn[:, 1] - sn[:, 0]) / sn[:, 0]).min().item()
    with torch.no_grad():
        feats = [torch.nn.functional.normalize(out.patches[m].mean(1), dim=-1) for m in range(3)]
        disc["min_volume"] = O.gram_volume3(*feats).min().item()
    return loss.item(), grads, disc, out


def cos(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float(a @ b / (a.norm() * b.norm()).clamp_min(1e-300))


def report(tag, ga, gb, top=8):
    rows = []
    for k in gb:
        if k in ga and float(gb[k].norm()) > 1e-5:
            rows.append((cos(ga[k], gb[k]), float(ga[k].norm() / gb[k].norm()), k))
    rows.sort()
    whole = cos(torch.cat([ga[k].flatten() for _, _, k in rows]), torch.cat([gb[k].flatten() for _, _, k in rows]))
    print(f"--- {tag}: whole-gradient cos {whole:.6f}; worst parameters:")
    for c, r, k in rows[:top]:
        print(f"    cos {c:.5f} norm ratio {r:.4f}  {k}")
    return rows


def main():
    global DT, ATTN_DEVICE, DELTA_FROM_O
    ap = argparse.ArgumentParser()
    ap.add_argument("--attn", default="autograd", choices=["autograd", "device", "device-exact-delta"])
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--tag", default="rgbnt201")
    ap.add_argument("--ulp-frac", type=float, default=0.003)
    ap.add_argument("--seed", type=int, default=None)
    a = ap.parse_args()
    DT = torch.bfloat16 if a.dtype == "bf16" else torch.float16
    ATTN_DEVICE, DELTA_FROM_O = a.attn != "autograd", a.attn == "device"
    torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
    g = np.load(os.path.join(os.path.dirname(__file__), "..", "golden", f"g7_step_{a.tag}.npz"))
    seed = int(g["seed"]) if a.seed is None else a.seed
    cfg = O.rgbnt201_config(num_instance=4) if a.tag == "rgbnt201" else O.rgbnt100_config(num_instance=4)
    sd = O.init_state_dict(cfg, seed=seed, head_scale=30.0)
    img, vid, cam = O.synthetic_batch(cfg, 8, seed=seed)
    h32, h16, h16p = [], [], []
    l32, g32, d32, o32 = run(sd, cfg, img, vid, cam, False, h32)
    PERTURB["frac"] = 0.0
    l16, g16, d16, o16 = run(sd, cfg, img, vid, cam, True, h16)
    PERTURB["frac"], PERTURB["gen"] = a.ulp_frac, np.random.default_rng(1)
    l16p, g16p, d16p, o16p = run(sd, cfg, img, vid, cam, True, h16p)
    print(f"loss fp32 {l32:.6f}  emulated-{a.dtype} {l16:.6f} ({abs(l16 - l32) / l32:.2e})  +1ulp on {a.ulp_frac:.1%} of LN outputs "
          f"{l16p:.6f} ({abs(l16p - l32) / l32:.2e})")
    for name, d in (("emulated", d16), ("emulated+ulp", d16p)):
        flips = int((d["mask"] != d32["mask"]).sum())
        tri = {k: int((d[k] != d32[k]).sum()) for k in d if k.startswith(("pidx", "nidx"))}
        print(f"{name}: SIM tokens flipped vs fp32 {flips}; batch-hard indices changed {tri}; min Gram volume {d['min_volume']:.4f}")
    print("fp32 smallest relative gaps of the batch-hard choices:", {k: f"{v:.2e}" for k, v in d32.items() if "gap" in k})
    report(f"emulated {a.dtype} vs fp32", g16, g32)
    report(f"emulated {a.dtype} + 1-ulp LN perturbation vs fp32", g16p, g32)
    report("emulated vs emulated + 1-ulp", g16p, g16)
    # the token gradient at every block boundary (hidden[i] = input of block i, hidden[12] = ln_post input), three modalities
    n = cfg.layers + 1
    for name, hs in (("emulated", h16), ("emulated+ulp", h16p)):
        cs = []
        for i in range(n):
            a_ = torch.cat([hs[m * n + i].grad.flatten() for m in range(3)])
            b_ = torch.cat([h32[m * n + i].grad.flatten() for m in range(3)])
            cs.append((cos(a_, b_), float(a_.norm() / b_.norm())))
        print(f"{name}: cos / norm ratio of dL/dx at block boundaries 12..0 vs fp32: " +
              " ".join(f"{c:.5f}/{r:.4f}" for c, r in reversed(cs)))
    print("top-level token gradient (patches, cls) emulated vs fp32:", round(cos(o16.patches.grad, o32.patches.grad), 6),
          round(cos(o16.cls.grad, o32.cls.grad), 6))


if __name__ == "__main__":
    main()
