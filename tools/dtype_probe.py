#!/usr/bin/env python3
"""Measure, for both MFMA operand types, what the parity tests bound: inference feature error against the fp32
oracle, the G7 loss terms against the reference fixture, and per-parameter gradient cosine / norm ratio.
    python tools/dtype_probe.py [rgbnt201|rgbnt100]   (GPU box)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import signal_ref as O  # noqa: E402
from tests.test_model_gpu import build, rel_err  # noqa: E402
from tests.test_train_gpu import cos  # noqa: E402


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "rgbnt201"
    dev = torch.device("cuda:0")
    from signal_amd.layers.make_loss import make_loss, total_loss
    g = np.load(os.path.join(ROOT, "tests", "golden", f"g7_step_{tag}.npz"))
    mk = O.rgbnt201_config if tag == "rgbnt201" else O.rgbnt100_config
    # ---- inference ----
    ocfg = mk()
    sd = O.init_state_dict(ocfg, seed=1234)
    img, vid, cam = O.synthetic_batch(ocfg, 4, seed=99)
    with torch.no_grad():
        ref = O.signal_forward_infer(sd, ocfg, img, cam)
        patches, cls = O.backbone3(sd, ocfg, img, cam)
        ref_mask, _ = O.sim_select(sd, patches, cls, ocfg.topk)
    for dt in ("bf16", "fp16"):
        model = build(ocfg, sd, dev, dt)
        x = {k: v.to(dev) for k, v in img.items()}
        with torch.no_grad():
            feat = model(x, cam_label=cam.to(dev), training=False)
            _, p_h, c_h = model._encode(x, cam.to(dev), False)
        hip_mask = torch.stack([model.SIM.token_selection.last_masks[m][..., 0] for m in O.MODALITIES]).bool().cpu()
        same = (hip_mask == ref_mask).all(dim=2).all(dim=0)
        print(f"[{tag} {dt}] infer: cls {rel_err(c_h, cls):.2e} patches {rel_err(p_h, patches):.2e} ori "
              f"{rel_err(feat[:, :1536], ref[:, :1536]):.2e} sim(all) {rel_err(feat[:, 1536:], ref[:, 1536:]):.2e} "
              f"sim(same masks, {int(same.sum())}/{len(same)}) "
              f"{rel_err(feat[same][:, 1536:], ref[same][:, 1536:]) if same.any() else float('nan'):.2e} "
              f"mask agree {(hip_mask == ref_mask).float().mean().item():.5f}", flush=True)
        del model
    # ---- train step ----
    ocfg = mk(num_instance=4)
    sd = O.init_state_dict(ocfg, seed=int(g["seed"]), head_scale=30.0)
    img, vid, cam = O.synthetic_batch(ocfg, 8, seed=int(g["seed"]))
    sdo = {k: v.clone() for k, v in sd.items()}
    for k, v in sdo.items():
        if v.is_floating_point() and "running_" not in k:
            v.requires_grad_(True)
    oloss, parts, oout = O.train_loss(sdo, ocfg, img, vid, cam)
    oloss.backward()
    ref_norm = dict(zip([str(k) for k in g["grad_keys"]], g["grad_norms"]))
    for dt in ("bf16", "fp16"):
        model = build(ocfg, sd, dev, dt)
        model.train()
        cfg = model.cfg
        loss_fn, _ = make_loss(cfg, ocfg.num_classes)
        out = model({k: v.to(dev) for k, v in img.items()}, label=vid.to(dev), cam_label=cam.to(dev), training=True, sge=ocfg.stage)
        loss = total_loss(cfg, out, loss_fn, vid.to(dev), cam.to(dev), ocfg.stage)
        scale = 1024.0 if dt == "fp16" else 1.0
        loss.backward(gradient=torch.tensor(scale, device=dev))
        named = dict(model.named_parameters())
        hip_mask = torch.stack([model.SIM.token_selection.last_masks[m][..., 0] for m in O.MODALITIES]).cpu().numpy()
        agree = (hip_mask.astype(np.int8) == g["masks"]).mean()
        rel = lambda a, b: abs(a - b) / abs(b)
        print(f"[{tag} {dt}] train: loss {rel(loss.item(), float(g['loss'])):.2e} gam {rel(out[-2].item(), float(g['gam'])):.2e} "
              f"lam {rel(out[-1].item(), float(g['lam'])):.2e} mask agree {agree:.5f}", flush=True)
        worst_c, worst_r = (1.0, None), (0.0, None)
        allg, allo = [], []
        for k, rn in ref_norm.items():
            p = named[k]
            if p.grad is None or rn < 1e-5:
                continue
            gh, go = p.grad / scale, sdo[k].grad
            c, ratio = cos(gh, go), float(gh.norm()) / rn
            if c < worst_c[0]:
                worst_c = (c, k)
            if abs(ratio - 1) > worst_r[0]:
                worst_r = (abs(ratio - 1), k)
            allg.append(gh.flatten().double().cpu())
            allo.append(go.flatten().double())
        G, Gr = torch.cat(allg), torch.cat(allo)
        print(f"[{tag} {dt}] grads: whole-vector cos {float(G @ Gr / (G.norm() * Gr.norm())):.6f} rel {float((G - Gr).norm() / Gr.norm()):.2e}; "
              f"worst cos {worst_c[0]:.5f} ({worst_c[1]}); worst |norm ratio - 1| {worst_r[0]:.2e} ({worst_r[1]})", flush=True)
        del model


if __name__ == "__main__":
    main()
