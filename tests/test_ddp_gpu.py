"""Two training processes on ONE GPU over gloo (RCCL needs one device per rank, so the collective backend is
the only thing swapped): exercises TrainStep's multi-process path end to end -- parameter broadcast, per-block
asynchronous bucket all-reduce from the backward hook, remainder buckets, 1/world scaling in the fused Adam --
and checks it against a single process that computes both ranks' gradients itself."""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from oracle import signal_ref as O
from tests.test_model_gpu import build
from signal_amd.engine.trainer import TrainStep
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:" + sys.argv[2], rank=rank, world_size=world)
dev = torch.device("cuda:0")
ocfg = O.rgbnt201_config(num_instance=2)
# different initial weights per rank on purpose: the reducer must broadcast rank 0's
sd = O.init_state_dict(ocfg, seed=100 + rank, head_scale=30.0)
model = build(ocfg, sd, dev)
cfg = model.cfg
cfg.SOLVER.OPTIMIZER_NAME = "Adam"; cfg.SOLVER.BASE_LR = 3.5e-4
ts = TrainStep(cfg, model, num_classes=ocfg.num_classes, world_size=world)
# SIGNAL_RESERVED_CUS (set by the test) sizes the backward's tiles and grids for 240 CUs even under gloo
assert ts.reserved_cus == int(os.environ.get("SIGNAL_RESERVED_CUS", "0")), ts.reserved_cus
if os.environ.get("SIGNAL_LATE_EVERYTHING") == "1":      # move the head-side remainder into the end-of-backward group
    ts.reducer.rest = sorted(ts.reducer.rest_early + ts.reducer.rest)
    ts.reducer.rest_early = []
# DDP's buffer broadcast: perturb this rank's BN running statistics, the step must restore rank 0's before the forward
if rank == 1:
    model.bottleneck.running_mean.add_(3.0)
img, vid, cam = O.synthetic_batch(ocfg, 4, seed=500 + rank)
img = {k: v.to(dev) for k, v in img.items()}
_fwd = model.forward
def spy(*a, **k):            # what the forward sees: the buffers after DDP's per-forward broadcast
    ts.bn_mean_seen = model.bottleneck.running_mean.clone()
    return _fwd(*a, **k)
model.forward = spy
loss = ts.step(img, vid.to(dev), cam.to(dev))
torch.cuda.synchronize()
out = {"grad": model.hip.flat.grad.cpu(), "param": model.hip.flat.data.cpu(), "loss": float(loss),
       "bn_mean_before": ts.bn_mean_seen.cpu()}
torch.save(out, sys.argv[3] + f"/rank{rank}.pt")
dist.destroy_process_group()
print("ok", rank)
'''


@pytest.mark.parametrize("reserved,late_all", [("16", "0"), ("0", "1")])
def test_two_rank_train_step_matches_single_process(tmp_path, reserved, late_all):
    """reserved = 16: the backward runs with the tile choice and grids sized for 240 CUs (what an nccl group gets by default);
    late_all = 1: every non-block bucket is sent at the end of the backward instead of right after the head stage -- the early
    issue must not change a single bit (it is only legal because every head-side gradient is final by then)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = str(sk.getsockname()[1])
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", SIGNAL_RESERVED_CUS=reserved,
                   SIGNAL_LATE_EVERYTHING=late_all)
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT, port, str(tmp_path)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=400)[0])
        except subprocess.TimeoutExpired:
            p.kill()
            outs.append("TIMEOUT " + p.communicate()[0])
    assert all(p.returncode == 0 for p in procs), "\n".join(o[-2000:] for o in outs)
    r0 = torch.load(tmp_path / "rank0.pt", weights_only=True)
    r1 = torch.load(tmp_path / "rank1.pt", weights_only=True)
    # both ranks hold the same summed gradient and the same updated parameters
    assert torch.equal(r0["grad"], r1["grad"])
    assert torch.equal(r0["param"], r1["param"])
    # rank 1's perturbed BN running mean was overwritten by rank 0's before its forward (DDP broadcast_buffers)
    assert torch.equal(r0["bn_mean_before"], r1["bn_mean_before"])

    # single-process reference: rank-0 weights, both ranks' batches, gradients summed by hand
    from oracle import signal_ref as O
    from tests.test_model_gpu import build
    from signal_amd.engine.trainer import TrainStep
    dev = torch.device("cuda:0")
    ocfg = O.rgbnt201_config(num_instance=2)
    grads = []
    for r in range(2):
        sd = O.init_state_dict(ocfg, seed=100, head_scale=30.0)
        model = build(ocfg, sd, dev)
        cfg = model.cfg
        cfg.SOLVER.OPTIMIZER_NAME = "Adam"
        cfg.SOLVER.BASE_LR = 0.0          # lr 0: the step leaves parameters alone, we only want the gradient
        ts = TrainStep(cfg, model, num_classes=ocfg.num_classes, world_size=1)
        img, vid, cam = O.synthetic_batch(ocfg, 4, seed=500 + r)
        ts.step({k: v.to(dev) for k, v in img.items()}, vid.to(dev), cam.to(dev))
        torch.cuda.synchronize()
        grads.append(model.hip.flat.grad.cpu().clone())
        names, offsets = model.hip.flat.names, model.hip.flat.offsets
    want = grads[0] + grads[1]
    # SIM.token_selection.* is never reduced (grad-less); everything else is the sum over ranks.  The token-gradient path
    # is order-deterministic (tests/test_train_gpu.py::test_backward_is_reproducible_run_to_run), so a rank's gradient is
    # the same in the 2-rank job and in the single process up to the f32-atomic sums of the bias / LayerNorm gradients.
    err = float((r0["grad"] - want).norm() / want.norm())
    assert err < 5e-6, err
    assert float((r0["param"] - grads[0] * 0).abs().sum()) > 0


def test_sig_comm_exports_run_over_rccl_with_one_rank():
    """K18 through the C ABI (include/signal_hip.h: sig_comm_*): RCCL is bound at run time, a one-rank communicator is
    created from a unique id, two gradient ranges are all-reduced on the communicator's side stream behind kernels that are
    still running on the compute stream, and the compute stream waits for them on the device.  One GPU = one rank (RCCL
    needs a device per rank), so the SUM over ranks is the identity here: what is checked is the binding, the stream
    ordering (the all-reduce must see the values written by the kernels enqueued before it) and that nothing else moves.
    More ranks: unmeasured (no multi-GPU box)."""
    import ctypes
    from signal_amd import _lib
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    dev = torch.device("cuda:0")
    _lib.load()
    uid = ctypes.create_string_buffer(128)
    _lib.call("sig_comm_unique_id", ctypes.cast(uid, ctypes.c_void_p))
    assert any(uid.raw)
    comm = ctypes.c_void_p()
    _lib.call("sig_comm_init", ctypes.cast(ctypes.byref(comm), ctypes.c_void_p), 0, 1, ctypes.cast(uid, ctypes.c_void_p))
    assert comm.value
    try:
        st = torch.cuda.current_stream().cuda_stream
        g = torch.zeros(1 << 22, device=dev)
        big = torch.randn(4096, 4096, device=dev)
        for _ in range(3):                       # keep the compute stream busy, then produce the "gradients"
            big = big @ big * 1e-3
        g[: 1 << 20] = 2.0
        _lib.call("sig_comm_allreduce_async", comm, g.data_ptr(), 1 << 20, st)
        g[1 << 20: 1 << 21] = 3.0
        _lib.call("sig_comm_allreduce_async", comm, g.data_ptr() + 4 * (1 << 20), 1 << 20, st)
        _lib.call("sig_comm_wait", comm, st)
        total = g.sum()                          # enqueued behind the wait on the compute stream
        torch.cuda.synchronize()
        assert float(total) == 2.0 * (1 << 20) + 3.0 * (1 << 20)
        assert float(g[1 << 21:].abs().max()) == 0.0
    finally:
        _lib.call("sig_comm_destroy", comm)


_ONE_RANK = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from oracle import signal_ref as O
from tests.test_model_gpu import build
from signal_amd.engine.trainer import TrainStep
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
ocfg = O.rgbnt201_config(num_instance=2)
img, vid, cam = O.synthetic_batch(ocfg, 4, seed=500)
batch = ({k: v.to(dev) for k, v in img.items()}, vid.to(dev), cam.to(dev))
def run(force):
    sd = O.init_state_dict(ocfg, seed=100, head_scale=30.0)
    model = build(ocfg, sd, dev)
    cfg = model.cfg
    cfg.SOLVER.OPTIMIZER_NAME = "Adam"; cfg.SOLVER.BASE_LR = 3.5e-4
    ts = TrainStep(cfg, model, num_classes=ocfg.num_classes, world_size=1, force_reducer=force)
    sent = []
    if force:
        assert ts.reducer is not None and ts.reducer.active and ts.reserved_cus == 16, (ts.reducer, ts.reserved_cus)
        orig = ts.reducer._reduce
        ts.reducer._reduce = lambda lo, hi: (sent.append((lo, hi)), orig(lo, hi))[1]
    # What is COMPARED is the first step (loss, gradients, updated parameters).  The second step only has to run the whole exchange
    # again: its gradients are not comparable between two runs of ANY configuration -- Adam's first update is lr * g / |g|, so the
    # elements whose gradient is zero up to the ~1e-8 reordering noise of the f32 atomics move by +-lr at random, and one batch-hard
    # or top-k near-tie of the second forward then falls the other way in about every third run (measured with
    # tools/ddp_one_rank_probe.py: plain vs plain, same binary, step 0 3e-8, step 1 either 5e-8 or exactly 5.27e-3).
    loss = ts.step(*batch)
    torch.cuda.synchronize()
    first = (float(loss), model.hip.flat.grad.clone(), model.hip.flat.data.clone())
    loss2 = ts.step(*batch)
    torch.cuda.synchronize()
    assert torch.isfinite(loss2) and float(loss2) < 1.5 * abs(first[0]), (first[0], float(loss2))   # (one Adam step at 3.5e-4 moves it 4.15 -> 2.6)
    return first[0], first[1], first[2], sent, ts
l0, g0, p0, _, _ = run(False)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:" + sys.argv[2], rank=0, world_size=1, device_id=dev)
l1, g1, p1, sent, ts = run(True)
# 2 steps x (head-side remainder + 12 block buckets + embedding group), every range through ncclAllReduce
per_step = len(ts.reducer.rest_early) + 12 + len(ts.reducer.rest)
assert len(sent) == 2 * per_step, (len(sent), per_step)
blocks = [r for r in sent[:per_step] if r in ts.reducer.blocks.values()]
assert blocks == [ts.reducer.blocks[i] for i in reversed(range(12))]
# sum over one rank = identity; the backward ran with the tiles / grids sized for 240 CUs (another summation order only)
err_g = float((g1 - g0).norm() / g0.norm()); err_p = float((p1 - p0).norm() / p0.norm())
assert abs(l1 - l0) < 1e-5 * abs(l0) and err_g < 5e-6 and err_p < 1e-6, (l0, l1, err_g, err_p)
dist.destroy_process_group()
print("ok one-rank RCCL", len(sent), err_g)
'''


def test_one_rank_rccl_group_runs_the_whole_exchange_path(tmp_path):
    """The RCCL path itself on the one GPU there is: an `nccl` process group with ONE rank, TrainStep(force_reducer=True).  Every
    bucket of two train steps goes through torch.distributed's asynchronous ncclAllReduce on RCCL's stream, issued from the
    backward hooks in reverse block order, waited for before the fused Adam; the nccl default of 16 reserved CUs sizes the
    backward GEMMs.  The sum over one rank is the identity, so losses, gradients and updated parameters must equal the plain
    single-GPU step's.  (More ranks: no multi-GPU box; unmeasured.)"""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    script = tmp_path / "one_rank.py"
    script.write_text(_ONE_RANK)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = str(sk.getsockname()[1])
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.pop("SIGNAL_RESERVED_CUS", None)
    p = subprocess.run([sys.executable, str(script), ROOT, port], env=env, capture_output=True, text=True, timeout=600)
    if p.returncode != 0:       # (pytest shortens long assertion messages: keep the child's full stderr where it can be read)
        try:
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            open(os.path.join(ROOT, "gpurun_out", "one_rank_stderr.txt"), "w").write(p.stdout + "\n---- stderr ----\n" + p.stderr)
        except OSError:
            pass
    assert p.returncode == 0 and "ok one-rank RCCL" in p.stdout, (p.stdout[-1500:], p.stderr[-3000:])
