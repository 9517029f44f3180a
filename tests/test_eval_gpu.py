"""The drop-in boundary and the evaluation row (N4) on a real MI355X: do_train / do_inference with the reference's
signatures over synthetic loaders, the evaluator's MFMA distance matrix against float64, checkpoint round trips."""
import logging
import os

import numpy as np
import pytest
import torch

from oracle import signal_ref as O
from tests.test_model_gpu import build, make_cfg, rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


class ValLoader:
    """The reference's val collate tuple (make_dataloader.py:164-183): (img dict, pids, camids, camids_batch, viewids, paths)."""

    def __init__(self, ocfg, n, bs, seed):
        self.items = []
        g = np.random.default_rng(seed)
        for lo in range(0, n, bs):
            b = min(bs, n - lo)
            img, _, cam = O.synthetic_batch(ocfg, b, seed=seed + lo)
            pids = g.integers(0, 6, size=b).tolist()
            self.items.append((img, pids, cam.tolist(), cam, torch.zeros(b, dtype=torch.int64), [f"im{lo + i}.jpg" for i in range(b)]))

    def __iter__(self):
        return iter(self.items)

    def __len__(self):
        return len(self.items)


def test_evaluator_distance_matrix_on_the_mfma_gemm_matches_float64(dev):
    from signal_amd.utils.metrics import R1_mAP_eval, eval_func, euclidean_distance
    g = torch.Generator().manual_seed(0)
    nq, ng, d = 37, 203, 3072                       # ragged on purpose: rows / gallery padded to the GEMM's tiles
    feats = torch.randn(nq + ng, d, generator=g)
    feats[nq + 5] = feats[3] + 1e-4 * torch.randn(d, generator=g)      # a near-duplicate: ranking needs f32-grade distances
    pids = np.random.default_rng(1).integers(0, 12, size=nq + ng)
    cams = np.random.default_rng(2).integers(0, 4, size=nq + ng)
    fn = torch.nn.functional.normalize(feats.double(), dim=1)
    want = (fn[:nq].pow(2).sum(1, keepdim=True) + fn[nq:].pow(2).sum(1, keepdim=True).t() - 2 * fn[:nq] @ fn[nq:].t()).numpy()
    got = euclidean_distance(torch.nn.functional.normalize(feats.to(dev), dim=1)[:nq], torch.nn.functional.normalize(feats.to(dev), dim=1)[nq:])
    assert got.is_cuda and got.shape == (nq, ng)
    assert np.abs(got.cpu().numpy() - want).max() < 5e-6          # f32-grade (a plain bf16 GEMM would be ~4e-3)
    ev = R1_mAP_eval(nq, max_rank=50, feat_norm="yes")
    for lo in range(0, nq + ng, 64):
        ev.update((feats[lo:lo + 64].to(dev), pids[lo:lo + 64], cams[lo:lo + 64]))
    cmc, mAP, distmat, *_ = ev.compute()
    ref_cmc, ref_map = eval_func(want, pids[:nq], pids[nq:], cams[:nq], cams[nq:], 50)
    np.testing.assert_allclose(cmc, ref_cmc, atol=1e-6)
    assert mAP == pytest.approx(ref_map, abs=1e-6)
    assert (np.argsort(distmat, axis=1)[:, :10] == np.argsort(want, axis=1)[:, :10]).mean() > 0.999


@pytest.mark.parametrize("tag", ["small", "reid201", "few_gallery", "many_cams"])
def test_evaluators_on_the_device_vs_reference_fixture(dev, golden, tag):
    """The evaluator classes end to end on the GPU (features in HBM, distance matrix on the MFMA GEMM, device argsort) against G10:
    the REFERENCE's eval_func / eval_func_msrv / euclidean_distance run on the same seeded features (tests/golden/make_golden_metrics.py).
    R1_mAP_eval = camera protocol, R1_mAP = MSVR310 scene protocol."""
    from signal_amd.utils.metrics import R1_mAP, R1_mAP_eval
    from tests.golden.make_golden_metrics import make_case
    g = golden("g10_metrics")
    seed, nq, ng, ids, cams, scenes, dim, max_rank = (int(v) for v in g[f"{tag}_case"])
    (qf, qp, qc, qs), (gf, gp, gc, gs) = make_case(seed, nq, ng, ids, cams, scenes, dim)
    feats = torch.from_numpy(np.concatenate([qf, gf])).to(dev)
    pids, camids, scn = np.concatenate([qp, gp]), np.concatenate([qc, gc]), np.concatenate([qs, gs])
    ev = R1_mAP_eval(nq, max_rank=max_rank, feat_norm="yes")
    ev2 = R1_mAP(nq, max_rank=max_rank, feat_norm="yes")
    for lo in range(0, nq + ng, 50):
        ev.update((feats[lo:lo + 50], pids[lo:lo + 50], camids[lo:lo + 50]))
        ev2.update((feats[lo:lo + 50], pids[lo:lo + 50], camids[lo:lo + 50], scn[lo:lo + 50]))
    cmc, mAP, distmat, *_ = ev.compute()
    np.testing.assert_allclose(distmat[:3], g[f"{tag}_dist_rows"], rtol=0, atol=3e-6)
    np.testing.assert_allclose(cmc, g[f"{tag}_cmc"], rtol=0, atol=1e-6)
    assert mAP == pytest.approx(float(g[f"{tag}_mAP"]), abs=1e-6)
    cmc_s, mAP_s = ev2.compute()[:2]
    np.testing.assert_allclose(cmc_s, g[f"{tag}_cmc_msrv"], rtol=0, atol=1e-6)
    assert mAP_s == pytest.approx(float(g[f"{tag}_mAP_msrv"]), abs=1e-6)


def test_checkpoint_round_trip_with_the_ddp_prefix(dev, tmp_path):
    """processor.py:310-321 saves model.state_dict(); under DDP the keys carry 'module.' and Signal.load_param strips it
    (make_model.py:125-130).  state_dict -> add prefix -> load_param into a fresh model -> bit-identical features."""
    ocfg = O.rgbnt201_config()
    sd = O.init_state_dict(ocfg, seed=11)
    img, _, cam = O.synthetic_batch(ocfg, 3, seed=12)
    x = {k: v.to(dev) for k, v in img.items()}
    model = build(ocfg, sd, dev)
    with torch.no_grad():
        want = model(x, cam_label=cam.to(dev), training=False)
    full = model.state_dict()
    assert set(sd) <= set(full)
    path = str(tmp_path / "Signal_1.pth")
    torch.save({"module." + k: v for k, v in full.items()}, path)
    fresh = build(ocfg, O.init_state_dict(ocfg, seed=999), dev)
    with torch.no_grad():
        other = fresh(x, cam_label=cam.to(dev), training=False)
    assert not torch.equal(other, want)
    fresh.load_param(path)
    with torch.no_grad():
        got = fresh(x, cam_label=cam.to(dev), training=False)
    assert torch.equal(got, want)
    # the backbone's own loader (meta_arch.py:114-118) with the same prefix handling
    bpath = str(tmp_path / "backbone.pth")
    torch.save({"module." + k: v for k, v in model.clip_vision_encoder.state_dict().items()}, bpath)
    third = build(ocfg, O.init_state_dict(ocfg, seed=5), dev)
    third.clip_vision_encoder.load_param(bpath)
    for k, v in model.clip_vision_encoder.state_dict().items():
        assert torch.equal(third.clip_vision_encoder.state_dict()[k], v), k
    assert third.flops() == model.flops() and 60e9 < model.flops() < 80e9     # 69.4 GFLOP per triplet


def test_do_train_and_do_inference_run_the_reference_loop(dev, tmp_path, caplog):
    """engine/processor.py:41-50,353-359 signatures, the reference's call order of train.py:72-109 (optimizer and scheduler
    are built while the model is still on the CPU), one epoch over synthetic triplets with checkpointing and the in-training
    evaluation, then do_inference on the saved checkpoint."""
    from signal_amd.data import SyntheticTriplets
    from signal_amd.engine.processor import do_inference, do_train
    from signal_amd.layers.make_loss import make_loss
    from signal_amd.modeling import make_frame
    from signal_amd.solver.make_optimizer import make_optimizer
    from signal_amd.solver.scheduler_factory import create_scheduler
    ocfg = O.rgbnt201_config(num_instance=2)
    cfg = make_cfg(ocfg)
    cfg.SOLVER.OPTIMIZER_NAME, cfg.SOLVER.BASE_LR = "Adam", 3.5e-4
    cfg.SOLVER.MAX_EPOCHS, cfg.SOLVER.LOG_PERIOD, cfg.SOLVER.CHECKPOINT_PERIOD, cfg.SOLVER.EVAL_PERIOD = 1, 1, 1, 1
    cfg.OUTPUT_DIR, cfg.ckpt_save_path = str(tmp_path), "run"
    torch.manual_seed(1234)
    model = make_frame(cfg, ocfg.num_classes, ocfg.camera_num, 0)
    sd0 = {k: v.clone() for k, v in model.state_dict().items()}
    loss_fn, center = make_loss(cfg, ocfg.num_classes)
    optimizer, optimizer_center = make_optimizer(cfg, model, center)        # CPU model: plain torch.optim.Adam
    scheduler = create_scheduler(cfg, optimizer)
    train_loader = SyntheticTriplets(batch=4, hw=tuple(ocfg.size_train), num_instances=2, cams=ocfg.camera_num, steps=3, seed=3)
    val_loader = ValLoader(ocfg, n=21, bs=6, seed=50)          # 5 queries + 16 gallery (Rank-10 is logged)
    caplog.set_level(logging.INFO)
    # do_train feeds the engine through DevicePrefetcher (H2D of batch i+1 on a side stream under step i): what the engine sees
    # must be the loader's batches, in order, already on the device
    seen = []
    do_train(cfg, model, center, train_loader, val_loader, optimizer, optimizer_center, scheduler, loss_fn, 5, 0, cfg.MODEL.stageName,
             batch_hook=lambda ep, it, img, vid, cam, view: seen.append(({k: v.clone() for k, v in img.items()}, vid.clone(), cam.clone())))
    host = list(train_loader)                                   # (SyntheticTriplets is seeded: the same three batches again)
    assert len(seen) == len(host) == 3
    for (dimg, dvid, dcam), (himg, hvid, hcam, _, _) in zip(seen, host):
        assert all(dimg[k].is_cuda and torch.equal(dimg[k].cpu(), himg[k]) for k in ("RGB", "NI", "TI"))
        assert torch.equal(dvid.cpu(), hvid) and torch.equal(dcam.cpu(), hcam)
    text = caplog.text
    assert "Epoch[1] Iteration[3/3] Loss:" in text and "Validation Results - Epoch: 1" in text and "Best mAP:" in text
    ck = os.path.join(str(tmp_path), "run", cfg.MODEL.NAME + "_1.pth")
    best = os.path.join(str(tmp_path), "run", cfg.MODEL.NAME + "best.pth")
    assert os.path.exists(ck) and os.path.exists(best)
    saved = torch.load(ck, map_location="cpu", weights_only=True)
    assert set(saved) == set(sd0) and not any(k.startswith("module.") for k in saved)
    w = "clip_vision_encoder.base.transformer.resblocks.5.attn.in_proj_weight"
    assert not torch.equal(saved[w], sd0[w])                                  # it trained
    for k in sd0:
        if k.startswith("SIM.token_selection."):
            assert torch.equal(saved[k], sd0[k]), k                           # grad-less parameters untouched
    # test.py: fresh model, load_param, do_inference
    model2 = make_frame(cfg, ocfg.num_classes, ocfg.camera_num, 0)
    model2.load_param(ck)
    r1, r5 = do_inference(cfg, model2, val_loader, 5, logging.getLogger("t"), cfg.MODEL.stageName, 0)
    assert 0.0 <= r1 <= r5 <= 1.0
    # the evaluation inside do_train and do_inference saw the same weights -> the same numbers
    import re
    maps = re.findall(r"mAP: ([0-9.]+)%", caplog.text)
    assert len(maps) >= 2 and maps[0] == maps[-1], maps
