"""GPU diagnostic for the bf16 per-parameter gradient deviation of the B=8 train fixture.

One train step of fixture G7 on the device; the fp32 oracle's autograd on the CPU.  Prints, per block, the cosine / norm
ratio of every backward signal the device materialises (dx_out, du, dx_mid, dqkv, dh1, dx_in) against the oracle's, the
batch-hard indices, the SIM tokens flipped, the smallest Gram volume, and the worst parameters.

    python tests/probes/grad_probe_gpu.py [--dtype bf16] [--tag rgbnt201]

Test infrastructure (imports oracle/); never imported by the product."""
import argparse
import math
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import signal_ref as O  # noqa: E402
from tests.test_model_gpu import build  # noqa: E402


def cos(a, b):
    a, b = a.detach().double().cpu().flatten(), b.detach().double().cpu().flatten()
    return float(a @ b / (a.norm() * b.norm()).clamp_min(1e-300))


def ratio(a, b):
    return float(a.detach().double().cpu().norm() / b.detach().double().cpu().norm().clamp_min(1e-300))


REC = []   # per vit_block call of the oracle: dict of retained tensors


def block_rec(sd, pre, x, heads):
    S, L, D = x.shape
    hd = D // heads
    h1 = O.layer_norm(x, sd[pre + "ln_1.weight"], sd[pre + "ln_1.bias"])
    qkv = h1 @ sd[pre + "attn.in_proj_weight"].t() + sd[pre + "attn.in_proj_bias"]
    q, k, v = (t.reshape(S, L, heads, hd).transpose(1, 2) for t in qkv.split(D, dim=-1))
    att = torch.softmax((q @ k.transpose(-1, -2)) / math.sqrt(hd), dim=-1)
    o = (att @ v).transpose(1, 2).reshape(S, L, D)
    x_mid = x + o @ sd[pre + "attn.out_proj.weight"].t() + sd[pre + "attn.out_proj.bias"]
    h2 = O.layer_norm(x_mid, sd[pre + "ln_2.weight"], sd[pre + "ln_2.bias"])
    pre_act = h2 @ sd[pre + "mlp.c_fc.weight"].t() + sd[pre + "mlp.c_fc.bias"]
    g = O.quick_gelu(pre_act)
    out = x_mid + g @ sd[pre + "mlp.c_proj.weight"].t() + sd[pre + "mlp.c_proj.bias"]
    rec = dict(x_in=x, h1=h1, qkv=qkv, o=o, x_mid=x_mid, h2=h2, pre=pre_act, x_out=out)
    for t in rec.values():
        t.retain_grad()
    REC.append(rec)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--tag", default="rgbnt201")
    a = ap.parse_args()
    from signal_amd.layers.make_loss import make_loss, total_loss
    dev = torch.device("cuda:0")
    g = np.load(os.path.join(os.path.dirname(__file__), "..", "golden", f"g7_step_{a.tag}.npz"))
    ocfg = O.rgbnt201_config(num_instance=4) if a.tag == "rgbnt201" else O.rgbnt100_config(num_instance=4)
    sd = O.init_state_dict(ocfg, seed=int(g["seed"]), head_scale=30.0)
    img, vid, cam = O.synthetic_batch(ocfg, 8, seed=int(g["seed"]))
    B, L, NB = 8, ocfg.tokens, ocfg.layers

    # ---------------- device ----------------
    model = build(ocfg, sd, dev, a.dtype)
    model.train()
    hip = model.hip
    cap = {}
    orig_bwd = hip.vit_backward

    def vit_backward(ws):
        cap["ws"] = ws
        M = ws["M"]
        cap["dtokens"] = ws["dtokens"][:M].clone()
        hip.on_block_grads_ready = lambda i: cap.__setitem__(i, {k: ws[k][:M].float().clone() for k in ("du", "dh", "dqkv", "dx_mid", "dx")})
        orig_bwd(ws)
        hip.on_block_grads_ready = None
    hip.vit_backward = vit_backward
    orig_head = None
    loss_fn, _ = make_loss(model.cfg, ocfg.num_classes)
    out = model({k: v.to(dev) for k, v in img.items()}, label=vid.to(dev), cam_label=cam.to(dev), training=True, sge=ocfg.stage)
    loss = total_loss(model.cfg, out, loss_fn, vid.to(dev), cam.to(dev), ocfg.stage)
    scale = 1024.0 if a.dtype == "fp16" else 1.0
    # dx entering block 11 = after head_bwd: captured as the "dx" the first hook sees is block 11's dx_in; take head dx separately
    loss.backward(gradient=torch.tensor(scale, device=dev))
    torch.cuda.synchronize()
    hip_mask = torch.stack([model.SIM.token_selection.last_masks[m][..., 0] for m in O.MODALITIES]).cpu().numpy().astype(bool)
    flipped = int((hip_mask.astype(np.int8) != g["masks"]).sum())

    # ---------------- oracle ----------------
    torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
    sdo = {k: v.clone() for k, v in sd.items()}
    for k, v in sdo.items():
        if v.is_floating_point() and "running_" not in k:
            v.requires_grad_(True)
    keep = O.vit_block
    O.vit_block = block_rec
    try:
        oloss, parts, oout = O.train_loss(sdo, ocfg, img, vid, cam, force_mask=torch.from_numpy(hip_mask) if flipped else None)
        oout.patches.retain_grad(); oout.cls.retain_grad()
        oloss.backward()
    finally:
        O.vit_block = keep
    print(f"[{a.tag} {a.dtype}] loss device {loss.item():.6f} oracle {oloss.item():.6f}; SIM tokens flipped vs fixture {flipped} "
          f"(oracle differentiated under the device's selection)")
    # discrete decisions
    for i, (score, feat) in enumerate(oout.pairs):
        f = feat.detach()
        sq = (f * f).sum(1, keepdim=True)
        dist = (sq + sq.t() - 2 * f @ f.t()).clamp(min=1e-12).sqrt()
        same = vid[:, None] == vid[None, :]
        ap_, an_ = torch.where(same, dist, torch.full_like(dist, -math.inf)), torch.where(same, torch.full_like(dist, math.inf), dist)
        fd = out[2 + 2 * i].detach().float().cpu()
        sqd = (fd * fd).sum(1, keepdim=True)
        dd = (sqd + sqd.t() - 2 * fd @ fd.t()).clamp(min=1e-12).sqrt()
        apd, and_ = torch.where(same, dd, torch.full_like(dd, -math.inf)), torch.where(same, torch.full_like(dd, math.inf), dd)
        print(f"  pair {i}: features rel err {float((fd - f).norm() / f.norm()):.2e}; batch-hard positives changed "
              f"{int((apd.argmax(1) != ap_.argmax(1)).sum())}, negatives changed {int((and_.argmin(1) != an_.argmin(1)).sum())}")
    with torch.no_grad():
        feats = [torch.nn.functional.normalize(oout.patches[m].mean(1), dim=-1) for m in range(3)]
        print(f"  smallest Gram volume {O.gram_volume3(*feats).min().item():.4f}")
    # top-level token gradient
    dt = cap["dtokens"].view(3, B, L, -1).float().cpu() / scale
    print(f"  top-level token gradient: patches cos {cos(dt[:, :, 1:], oout.patches.grad):.6f} ratio {ratio(dt[:, :, 1:], oout.patches.grad):.4f}; "
          f"cls cos {cos(dt[:, :, 0], oout.cls.grad):.6f} ratio {ratio(dt[:, :, 0], oout.cls.grad):.4f}")

    def ora(i, key):      # oracle gradient of block i's tensor, rows ordered like the device (modality-major)
        return torch.cat([REC[m * NB + i][key].grad.reshape(B * L, -1) for m in range(3)])

    print("  block:  dx_out        du(dpre)      dx_mid        dqkv          dh1           dx_in      (cos/norm ratio vs oracle)")
    for i in reversed(range(NB)):
        c = cap[i]
        dx_out = cap[i + 1]["dx"] if i + 1 < NB else None
        cols = []
        for dev_t, key in ((dx_out, "x_out"), (c["du"], "pre"), (c["dx_mid"], "x_mid"), (c["dqkv"], "qkv"), (c["dh"], "h1"), (c["dx"], "x_in")):
            if dev_t is None:
                cols.append("     -      ")
                continue
            o_ = ora(i, key)
            cols.append(f"{cos(dev_t, o_):.5f}/{ratio(dev_t / scale, o_):.4f}")
        print(f"  {i:5d}:  " + "  ".join(cols))
    # parameters
    named = dict(model.named_parameters())
    rows = []
    for k, p in named.items():
        if p.grad is None or sdo[k].grad is None or float(sdo[k].grad.norm()) < 1e-5:
            continue
        rows.append((cos(p.grad, sdo[k].grad), ratio(p.grad / scale, sdo[k].grad), k))
    rows.sort()
    print("  worst parameters:")
    for c_, r_, k in rows[:24]:
        print(f"    cos {c_:.5f} norm ratio {r_:.4f}  {k}")


if __name__ == "__main__":
    main()
