#!/usr/bin/env python3
"""Diagnostic (GPU box): the one-rank RCCL train step against the plain step, several repetitions, with the parameters that
differ named (tests/test_ddp_gpu.py::test_one_rank_rccl_group_runs_the_whole_exchange_path failed intermittently)."""
import os, sys, socket, torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import signal_ref as O
from tests.test_model_gpu import build
from signal_amd.engine.trainer import TrainStep
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
ocfg = O.rgbnt201_config(num_instance=2)
img, vid, cam = O.synthetic_batch(ocfg, 4, seed=500)
batch = ({k: v.to(dev) for k, v in img.items()}, vid.to(dev), cam.to(dev))
STEPS = int(os.environ.get("PROBE_STEPS", "2"))
def run(force):
    sd = O.init_state_dict(ocfg, seed=100, head_scale=30.0)
    model = build(ocfg, sd, dev)
    cfg = model.cfg
    cfg.SOLVER.OPTIMIZER_NAME = "Adam"; cfg.SOLVER.BASE_LR = 3.5e-4
    ts = TrainStep(cfg, model, num_classes=ocfg.num_classes, world_size=1, force_reducer=force)
    grads = []
    for _ in range(STEPS):
        loss = ts.step(*batch)
        grads.append(model.hip.flat.grad.clone())
    torch.cuda.synchronize()
    return float(loss), grads, model.hip.flat
with socket.socket() as sk:
    sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
for rep in range(int(os.environ.get("PROBE_REPS", "4"))):
    res = {}
    for force in (False, True, False, True):
        l, grads, fl = run(force)
        res.setdefault(force, []).append((l, grads))
    def rel(a, b): return float((a - b).norm() / b.norm().clamp_min(1e-30))
    for name, (x, y) in {"plain vs plain": (res[False][0], res[False][1]), "rccl vs rccl": (res[True][0], res[True][1]),
                         "rccl vs plain": (res[True][0], res[False][0])}.items():
        for stp in range(STEPS):
            e = rel(x[1][stp], y[1][stp])
            line = f"rep {rep} {name:15s} step {stp}: whole-gradient rel diff {e:.3e}"
            if e > 1e-5:
                rows = []
                for n in fl.names:
                    o, k = fl.offsets[n], fl.byname[n].numel()
                    a, b = x[1][stp][o:o + k], y[1][stp][o:o + k]
                    if float(b.norm()) > 0: rows.append((rel(a, b), n))
                rows.sort(reverse=True)
                line += "  worst: " + ", ".join(f"{n} {r:.2e}" for r, n in rows[:6])
            print(line, flush=True)
dist.destroy_process_group()
