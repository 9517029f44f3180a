#!/usr/bin/env python3
"""Diagnostic (GPU box): run the same train step twice on freshly built identical models and report, per parameter, the
relative difference of the gradients (f32 atomics should give ~1e-6)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import signal_ref as O
from tests.test_model_gpu import build
from signal_amd.engine.trainer import TrainStep
dev = torch.device("cuda:0")
ocfg = O.rgbnt201_config(num_instance=2)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
def run():
    sd = O.init_state_dict(ocfg, seed=100, head_scale=30.0)
    model = build(ocfg, sd, dev)
    cfg = model.cfg
    cfg.SOLVER.OPTIMIZER_NAME = "Adam"; cfg.SOLVER.BASE_LR = 0.0
    ts = TrainStep(cfg, model, num_classes=ocfg.num_classes, world_size=1)
    img, vid, cam = O.synthetic_batch(ocfg, B, seed=500)
    loss = ts.step({k: v.to(dev) for k, v in img.items()}, vid.to(dev), cam.to(dev))
    torch.cuda.synchronize()
    fl = model.hip.flat
    with torch.no_grad():
        feat = model({k: v.to(dev) for k, v in img.items()}, cam_label=cam.to(dev), training=False).cpu()
    run.extra = (loss, feat)
    return fl.grad.cpu().clone(), fl.names, fl.offsets, {n: fl.byname[n].numel() for n in fl.names}
g0, names, off, num = run(); e0 = run.extra
g1, _, _, _ = run(); e1 = run.extra
print("loss:", e0[0], e1[0])
print("inference features equal:", torch.equal(e0[1], e1[1]), float((e0[1] - e1[1]).abs().max()))
def rd(n):
    a, b = g0[off[n]:off[n]+num[n]], g1[off[n]:off[n]+num[n]]
    return float((a-b).norm()/a.norm().clamp_min(1e-30))
for n in ["classifier.weight", "classifier_var.weight", "bottleneck.weight", "SIM.modal_interactive.ffn.0.weight", "SIM.modal_interactive.cross_attn.in_proj_weight",
          "AlignM.contra_temp", "AlignM.DAS_r.proj_q.weight", "clip_vision_encoder.base.proj", "clip_vision_encoder.base.ln_post.weight"] + \
         [f"clip_vision_encoder.base.transformer.resblocks.{i}.mlp.c_fc.weight" for i in (11, 10, 8, 6, 4, 2, 0)]:
    if n in off: print(f"  {rd(n):.3e}  {n}")
print(f"whole gradient: rel diff {float((g0-g1).norm()/g0.norm()):.3e}")
rows = []
for n in names:
    a, b = g0[off[n]:off[n]+num[n]], g1[off[n]:off[n]+num[n]]
    if float(a.norm()) > 0:
        rows.append((float((a-b).norm()/a.norm()), n, float(a.norm())))
rows.sort(reverse=True)
print("largest RELATIVE differences:")
for r, n, nm in rows[:8]: print(f"  {r:.3e}  |g|={nm:.3e}  {n}")
tot = float((g0 - g1).norm())
print("largest ABSOLUTE contributions to |g0 - g1| (share of the squared difference):")
ab = sorted(((float(((g0[off[n]:off[n]+num[n]] - g1[off[n]:off[n]+num[n]]) ** 2).sum()) / tot ** 2, n) for n in names), reverse=True)
for sh, n in ab[:12]:
    a, b = g0[off[n]:off[n]+num[n]], g1[off[n]:off[n]+num[n]]
    print(f"  {sh:6.3f}  rel {float((a-b).norm()/a.norm()):.3e}  {n}")
