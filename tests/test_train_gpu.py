"""Training-path parity on a real MI355X: backward of every HIP stage against the fp32 oracle's autograd on the
same weights/inputs, the reference's golden fixtures for GAM/LAM, a full train step, and the fused Adam."""
import numpy as np
import pytest
import torch

from oracle import signal_ref as O
from tests.test_model_gpu import FEAT_TOL, build, make_cfg, rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def cos(a, b):
    a, b = a.detach().double().cpu().flatten(), b.detach().double().cpu().flatten()
    return float(a @ b / (a.norm() * b.norm()).clamp_min(1e-300))


def tokens_from(patches, cls):
    """[3,B,Lp,d], [3,B,d] -> the backbone's output layout [3B, L, d]"""
    return torch.cat([cls.unsqueeze(2), patches], dim=2).reshape(-1, patches.shape[2] + 1, patches.shape[3]).contiguous()


def head_model(ocfg, sd, dev, dtype="bf16"):
    """A model whose ViT is never run (tests drive the head stages from given tokens)."""
    model = build(ocfg, sd, dev, dtype)
    model.hip.prepare(dev)
    return model


@pytest.mark.parametrize("tag", ["regular", "aligned"])
def test_gam_vs_reference_fixture(dev, golden, tag):
    from tests.golden.make_golden import head_features
    from signal_amd.modeling.hip_engine import GamFn
    g = golden(f"g4_gam_{tag}")
    ocfg = O.rgbnt201_config()
    sd = O.init_state_dict(ocfg, seed=int(g["seed_w"]))
    patches, cls = head_features(ocfg, 8, seed=int(g["seed_x"]))
    mix = float(g["mix"])
    if mix:
        patches = torch.stack([patches[0], mix * patches[0] + (1 - mix) * patches[1], mix * patches[0] + (1 - mix) * patches[2]])
    model = head_model(ocfg, sd, dev)
    tok = tokens_from(patches, cls).to(dev).requires_grad_(True)
    temp = model.AlignM.contra_temp
    loss = GamFn.apply(model.hip, 8, tok, temp)
    loss.backward()
    lo = tag == "aligned"
    np.testing.assert_allclose(loss.item(), float(g["loss"]), rtol=5e-4 if lo else 2e-5)
    dpat = tok.grad.view(3, 8, 129, 512)[:, :, 1:]
    assert float(tok.grad.view(3, 8, 129, 512)[:, :, 0].abs().max()) == 0.0      # GAM never touches the CLS rows
    np.testing.assert_allclose(dpat.flatten(1).norm(dim=1).cpu().numpy(), g["grad_norm"], rtol=5e-2 if lo else 1e-3)
    np.testing.assert_allclose(temp.grad.item(), float(g["temp_grad"]), rtol=5e-3 if lo else 1e-3)
    if not lo:
        np.testing.assert_allclose(dpat[:, :, :2].cpu().numpy(), g["grad_rows"], rtol=2e-3, atol=1e-8)


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("tag", ["16x8", "8x16"])
def test_lam_vs_reference_fixture(dev, golden, tag, dtype):
    from tests.golden.make_golden import head_features
    from signal_amd.modeling.hip_engine import LamFn
    g = golden(f"g5_lam_{tag}")
    ocfg = O.rgbnt201_config() if tag == "16x8" else O.rgbnt100_config()
    sd = O.init_state_dict(ocfg, seed=int(g["seed_w"]))
    patches, cls = head_features(ocfg, 4, seed=int(g["seed_x"]))
    model = head_model(ocfg, sd, dev, dtype)
    hip = model.hip
    tok = tokens_from(patches, cls).to(dev).requires_grad_(True)
    loss = LamFn.apply(hip, 4, tok, *[hip.flat.byname[n] for n in hip.das_param_names])
    loss.backward()
    # the two 1x1 convs run with 16-bit operands: offsets / samples / loss at that accuracy (T: fp16 rounds 8x finer)
    T = 1.0 if dtype == "bf16" else 0.15
    ws = hip._lam_ws[(4, True)][-1]
    P = 8
    offs = ws["t"]["offs"].view(3, 4, P, 3)[0, :, :, 0].cpu().numpy().reshape(g["offsets"].shape)
    dpat = tok.grad.view(3, 4, 129, 512)[:, :, 1:]
    w4g = model.AlignM.DAS_r.conv_offset[4].weight.grad.reshape(-1).cpu()
    sdo = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in sd.items()}
    pr = patches.clone().requires_grad_(True)
    O.lam_loss(sdo, ocfg, pr).backward()
    stats = [(n, cos(hip.flat.byname[n].grad, sdo[n].grad), abs(float(hip.flat.byname[n].grad.norm()) / float(sdo[n].grad.norm()) - 1))
             for n in hip.das_param_names]
    print(f"[lam {tag} {dtype}] offsets {np.abs(offs - g['offsets']).max() / np.abs(g['offsets']).max():.2e} loss "
          f"{abs(loss.item() - float(g['loss'])) / float(g['loss']):.2e} dpatch cos {cos(dpat, pr.grad):.6f} "
          f"param cos min {min(c for _, c, _ in stats):.6f} norm dev max {max(r for _, _, r in stats):.2e}")
    np.testing.assert_allclose(offs, g["offsets"], rtol=0, atol=1e-2 * T * np.abs(g["offsets"]).max())
    np.testing.assert_allclose(loss.item(), float(g["loss"]), rtol=5e-3 * T)
    np.testing.assert_allclose(dpat.flatten(1).norm(dim=1).cpu().numpy(), g["grad_norm"], rtol=2e-2 * T)
    assert cos(w4g, torch.from_numpy(g["w4_grad"])) > 1 - 2e-3 * T
    # against the oracle's autograd for every DAS parameter
    assert cos(dpat, pr.grad) > 1 - 2e-3 * T
    for n, c, r in stats:
        assert c > 1 - 3e-3 * T, (n, c)
        assert r < 3e-2 * T, (n, r)


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_sim_backward_vs_oracle(dev, dtype):
    from tests.golden.make_golden import head_features
    from signal_amd.modeling.hip_engine import SimFn
    ocfg = O.rgbnt201_config()
    sd = O.init_state_dict(ocfg, seed=31)
    patches, cls = head_features(ocfg, 8, seed=32)
    model = head_model(ocfg, sd, dev, dtype)
    hip = model.hip
    T = 1.0 if dtype == "bf16" else 0.15
    tok = tokens_from(patches, cls).to(dev).requires_grad_(True)
    out, mask = SimFn.apply(hip, 8, tok, *[hip.flat.byname[n] for n in hip.sim_param_names])
    w = torch.randn(8, 1536, generator=torch.Generator().manual_seed(3))
    (out * w.to(dev)).sum().backward()
    sdo = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in sd.items()}
    pr, cr = patches.clone().requires_grad_(True), cls.clone().requires_grad_(True)
    ref, rmask, _ = O.sim_forward(sdo, ocfg, pr, cr)
    (ref * w).sum().backward()
    assert torch.equal(mask.bool().cpu(), rmask)
    g4 = tok.grad.view(3, 8, 129, 512)
    stats = [(n, cos(hip.flat.byname[n].grad, sdo[n].grad), abs(float(hip.flat.byname[n].grad.norm()) / float(sdo[n].grad.norm()) - 1))
             for n in hip.sim_param_names]
    print(f"[sim bwd {dtype}] out {rel_err(out, ref):.2e} dpatch cos {cos(g4[:, :, 1:], pr.grad):.6f} dcls cos {cos(g4[:, :, 0], cr.grad):.6f} "
          f"param cos min {min(c for _, c, _ in stats):.6f} norm dev max {max(r for _, _, r in stats):.2e}")
    assert rel_err(out, ref) < 6e-3 * T
    assert cos(g4[:, :, 1:], pr.grad) > 1 - 2e-3 * T and cos(g4[:, :, 0], cr.grad) > 1 - 2e-3 * T
    assert abs(float(g4[:, :, 1:].norm()) / float(pr.grad.norm()) - 1) < 2e-2 * T
    for n, c, r in stats:
        assert c > 1 - 3e-3 * T, (n, c)
        assert r < 3e-2 * T, (n, r)
    for n in ("SIM.token_selection.W_q.weight", "SIM.token_selection.W_k.weight", "SIM.token_selection.W_v.weight"):
        assert hip.flat.byname[n].grad is None      # dead / selection-only parameters (useA.py:46-48)


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("tag", ["rgbnt201", "rgbnt100"])
def test_full_train_step_vs_oracle(dev, golden, tag, dtype):
    """Loss terms and every parameter gradient of one training iteration at real size (B=8: 2 ids x 4), both operand types.
    fp16 (the north_star bar): loss terms <= 1e-3 of the reference fixture, every parameter's gradient cos >= 0.9999;
    bf16: losses <= 2e-3, every parameter >= 0.9995 (measured 0.99985 = the operand-rounding floor)."""
    from signal_amd.layers.make_loss import make_loss, total_loss
    g = golden(f"g7_step_{tag}")
    ocfg = O.rgbnt201_config(num_instance=4) if tag == "rgbnt201" else O.rgbnt100_config(num_instance=4)
    sd = O.init_state_dict(ocfg, seed=int(g["seed"]), head_scale=30.0)
    img, vid, cam = O.synthetic_batch(ocfg, 8, seed=int(g["seed"]))
    model = build(ocfg, sd, dev, dtype)
    model.train()
    cfg = model.cfg
    loss_fn, _ = make_loss(cfg, ocfg.num_classes)
    out = model({k: v.to(dev) for k, v in img.items()}, label=vid.to(dev), cam_label=cam.to(dev), training=True,
                sge=ocfg.stage)
    assert out[0] == 3 and len(out) == (7 if ocfg.direct else 11)
    loss = total_loss(cfg, out, loss_fn, vid.to(dev), cam.to(dev), ocfg.stage)
    # fp16 backward signals need the loss scaled (the train engine uses a dynamic scale; a fixed one here)
    scale = 1024.0 if dtype == "fp16" else 1.0
    loss.backward(gradient=torch.tensor(scale, device=dev))
    # ---- forward quantities against the REFERENCE's fixture (fp32) ----
    ltol = 1e-3 if dtype == "fp16" else 2e-3
    np.testing.assert_allclose(out[-2].item(), float(g["gam"]), rtol=ltol)
    np.testing.assert_allclose(out[-1].item(), float(g["lam"]), rtol=ltol)
    np.testing.assert_allclose(loss.item(), float(g["loss"]), rtol=ltol)
    from tests.conftest import record_measure
    record_measure(f"{dtype}_loss_terms_rel", max(abs(out[-2].item() / float(g["gam"]) - 1), abs(out[-1].item() / float(g["lam"]) - 1),
                                                   abs(loss.item() / float(g["loss"]) - 1)))
    hip_mask = torch.stack([model.SIM.token_selection.last_masks[m][..., 0] for m in O.MODALITIES]).cpu().numpy()
    agree = (hip_mask.astype(np.int8) == g["masks"]).mean()
    assert agree > 0.995, agree
    # ---- gradients against the reference's per-parameter norms and the oracle's full gradients ----
    # The step has two kinds of DISCRETE decisions: the SIM top-k selection and the batch-hard triplet mining.  Where the
    # device's 16-bit features resolve a near-tie the other way, its gradient is a different, equally valid one: conv1 and
    # the attention weights drop to cos 0.997 with +1 % norm for ONE flipped hard negative (fixture gaps: 1.4e-4 / 2.8e-4
    # relative, below the 3.7e-3 bf16 feature error; tests/probes/grad_probe_gpu.py, round-2's unexplained 0.9974), and the
    # SIM cross-attention weights to 0.9995 for five flipped tokens.  So: (1) the decisions are compared on their own --
    # every deviation must be a near-tie in the fp32 oracle, and their number is bounded; (2) the gradients are compared
    # TIGHTLY under the device's decisions; (3) and loosely against the oracle's own decisions.
    def grads_of_oracle(force_mask, force_mining):
        sdo = {k: v.clone() for k, v in sd.items()}
        for k, v in sdo.items():
            if v.is_floating_point() and "running_" not in k:
                v.requires_grad_(True)
        oloss, parts, oout = O.train_loss(sdo, ocfg, img, vid, cam, force_mask=force_mask, force_mining=force_mining)
        oloss.backward()
        return sdo, oout

    flipped = int((hip_mask.astype(np.int8) != g["masks"]).sum())
    npairs = (len(out) - 3) // 2
    dev_mining = []
    for i in range(npairs):
        pi, ni, _, _ = O.batch_hard(O.pairwise_dist(out[2 + 2 * i].detach().float().cpu()), vid)
        dev_mining.append((pi, ni))
    sdo, oout = grads_of_oracle(torch.from_numpy(hip_mask.astype(bool)) if flipped else None, dev_mining)
    tie = 2 * FEAT_TOL[dtype]                       # a relative distance gap the 16-bit features cannot resolve
    mining_flips = 0
    for i, (_, feat) in enumerate(oout.pairs):
        pi, ni, pgap, ngap = O.batch_hard(O.pairwise_dist(feat.detach()), vid)
        for mine, own, gap in ((dev_mining[i][0], pi, pgap), (dev_mining[i][1], ni, ngap)):
            diff = (mine != own).nonzero().flatten().tolist()
            mining_flips += len(diff)
            assert all(float(gap[a_]) < tie for a_ in diff), (i, diff, gap)     # only genuine near-ties may be mined differently
    assert mining_flips <= 2, mining_flips
    print(f"[train step {tag} {dtype}] SIM tokens flipped {flipped} of {hip_mask.size}, batch-hard choices flipped {mining_flips}")
    ref_norm = dict(zip([str(k) for k in g["grad_keys"]], g["grad_norms"]))
    named = dict(model.named_parameters())
    # the fixture harness left the BNNeck biases trainable; the reference freezes them (make_model.py:78,88,...)
    frozen = {k for k in ref_norm if k.startswith("bottleneck") and k.endswith(".bias")}
    assert all(not named[k].requires_grad for k in frozen)
    assert {k for k, p in named.items() if p.grad is not None} == set(ref_norm) - frozen, "same set of parameters must receive gradients"

    def compare(sdo, pc, pw, check_norm):
        bad, all_h, all_o, worst = [], [], [], 1.0
        for k, rn in ref_norm.items():
            if k in frozen:
                continue
            gh, go = named[k].grad / scale, sdo[k].grad
            all_h.append(gh.detach().float().cpu().reshape(-1)); all_o.append(go.detach().float().reshape(-1))
            if rn < 1e-5:       # exactly-zero gradients in exact arithmetic (bias in front of a BatchNorm)
                continue
            c, ratio = cos(gh, go), float(gh.norm()) / float(go.norm())
            worst = min(worst, c)
            # (a scalar's "norm" is the value itself: GAM's temperature gradient is a cancelling sum over the B x B volumes,
            #  3.4 % off on bf16 features that are 3.8e-3 off; pinned at 1e-3 on exact features by the G4 fixture test)
            ntol = (5e-3 if dtype == "fp16" else 2e-2) * (3 if go.numel() == 1 else 1)
            if c < pc or (check_norm and abs(ratio - 1) > ntol):
                bad.append((k, round(c, 5), round(ratio, 4)))
        assert not bad, bad
        whole = cos(torch.cat(all_h), torch.cat(all_o))
        assert whole > pw, whole
        return worst, whole

    # (2) under the device's decisions.  Measured (shipped library): fp16 every parameter 1.00000, bf16 >= 0.99985 -- what the
    # CPU emulation of the 16-bit rounding points predicts (tests/probes/bf16_emulation.py: 0.99986), uniform over the blocks.
    worst, whole = compare(sdo, 0.9999 if dtype == "fp16" else 0.9995, 0.99999 if dtype == "fp16" else 0.9998, True)
    print(f"[train step {tag} {dtype}] under the device's decisions: worst parameter cos {worst:.6f}, whole gradient {whole:.7f}")
    from tests.conftest import record_measure
    record_measure(f"{dtype}_per_parameter_grad_cos", worst)
    if not flipped and not mining_flips:
        # the reference's per-parameter gradient norms (fixture G7) apply when no decision differs
        for k, rn in ref_norm.items():
            if k not in frozen and rn >= 1e-5:
                assert abs(float(named[k].grad.norm()) / scale / rn - 1) < (5e-3 if dtype == "fp16" else 2e-2) * (3 if named[k].numel() == 1 else 1), k
    else:
        # (3) against the oracle's OWN decisions: bounded by what flipped decisions cost (one hard negative: 0.997)
        sdo2, _ = grads_of_oracle(None, None)
        worst, whole = compare(sdo2, 0.995, 0.995, False)
        print(f"[train step {tag} {dtype}] against the oracle's own decisions: worst parameter cos {worst:.6f}, whole gradient {whole:.7f}")


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_full_train_step_B64_vs_oracle(dev, dtype):
    """configs[2] at FULL size against the oracle itself: B = 64 RGBNT201 (8 ids x 8), the benched kernels
    (persistent 192x256 / 320x256 / 256x256 NT tiles, grouped weight gradient), loss terms and every parameter's gradient
    against O.train_loss differentiated under the device's discrete decisions -- the same bounds as the B = 8 fixture test
    (0.9995 bf16 / 0.9999 fp16 per parameter).  With 64 anchors x 2 heads the batch-hard mining sees many more near-ties
    than at B = 8: every deviating choice must be a near-tie of the fp32 oracle (relative gap below twice the feature
    tolerance) and they are counted.  One oracle step at this size takes 15-40 s on the box's host cores."""
    from signal_amd.layers.make_loss import make_loss, total_loss
    B = 64
    ocfg = O.rgbnt201_config(num_instance=8)
    sd = O.init_state_dict(ocfg, seed=2024, head_scale=30.0)
    img, vid, cam = O.synthetic_batch(ocfg, B, seed=2024)
    assert len(set(vid.tolist())) == 8
    model = build(ocfg, sd, dev, dtype)
    model.train()
    cfg = model.cfg
    loss_fn, _ = make_loss(cfg, ocfg.num_classes)
    out = model({k: v.to(dev) for k, v in img.items()}, label=vid.to(dev), cam_label=cam.to(dev), training=True, sge=ocfg.stage)
    loss = total_loss(cfg, out, loss_fn, vid.to(dev), cam.to(dev), ocfg.stage)
    scale = 1024.0 if dtype == "fp16" else 1.0
    loss.backward(gradient=torch.tensor(scale, device=dev))
    hip_mask = torch.stack([model.SIM.token_selection.last_masks[m][..., 0] for m in O.MODALITIES]).cpu().numpy().astype(bool)
    npairs = (len(out) - 3) // 2
    dev_mining = []
    for i in range(npairs):
        pi, ni, _, _ = O.batch_hard(O.pairwise_dist(out[2 + 2 * i].detach().float().cpu()), vid)
        dev_mining.append((pi, ni))

    def oracle(force_mask, force_mining):
        sdo = {k: v.clone() for k, v in sd.items()}
        for k, v in sdo.items():
            if v.is_floating_point() and "running_" not in k:
                v.requires_grad_(True)
        oloss, parts, oout = O.train_loss(sdo, ocfg, img, vid, cam, force_mask=force_mask, force_mining=force_mining)
        oloss.backward()
        return sdo, oloss.detach(), parts, oout

    # ONE oracle pass (fp32, ~50 s at this size): differentiated under the device's decisions.  The SIM selection the oracle would
    # have made by itself comes back next to the forced one (sim_forward returns its own mask either way); GAM / LAM do not
    # depend on a decision; the batch-hard choices are re-derived from the oracle's features, with the size of every tie.
    sdo, oloss, parts, oout = oracle(torch.from_numpy(hip_mask), dev_mining)
    ltol = 1e-3 if dtype == "fp16" else 2e-3
    np.testing.assert_allclose(out[-2].item(), float(parts["gam"]), rtol=ltol)
    np.testing.assert_allclose(out[-1].item(), float(parts["lam"]), rtol=ltol)
    np.testing.assert_allclose(loss.item(), float(oloss), rtol=ltol)
    omask = oout.mask.reshape(hip_mask.shape).numpy().astype(bool)        # [3, B, Lp]: the oracle's own SIM selection
    agree = (hip_mask == omask).mean()
    assert agree > 0.995, agree
    tie = 2 * FEAT_TOL[dtype]
    mining_flips = 0
    for i, (_, feat) in enumerate(oout.pairs):
        pi, ni, pgap, ngap = O.batch_hard(O.pairwise_dist(feat.detach()), vid)
        for mine, own, gap in ((dev_mining[i][0], pi, pgap), (dev_mining[i][1], ni, ngap)):
            diff = (mine != own).nonzero().flatten().tolist()
            mining_flips += len(diff)
            assert all(float(gap[a_]) < tie for a_ in diff), (i, diff, [float(gap[a_]) for a_ in diff])
    assert mining_flips <= 16, mining_flips            # of 2 heads x 64 anchors x (positive, negative) = 256 choices
    named = dict(model.named_parameters())
    pc, pw = (0.9999, 0.99999) if dtype == "fp16" else (0.9995, 0.9998)
    bad, all_h, all_o, worst = [], [], [], 1.0
    for k, p in named.items():
        if p.grad is None:
            continue
        gh, go = p.grad / scale, sdo[k].grad
        assert go is not None, k
        all_h.append(gh.detach().float().cpu().reshape(-1)); all_o.append(go.detach().float().reshape(-1))
        if float(go.norm()) < 1e-5:
            continue        # exactly-zero gradients in exact arithmetic (a bias in front of a BatchNorm)
        c, ratio = cos(gh, go), float(gh.norm()) / float(go.norm())
        worst = min(worst, c)
        ntol = (5e-3 if dtype == "fp16" else 2e-2) * (3 if go.numel() == 1 else 1)
        if c < pc or abs(ratio - 1) > ntol:
            bad.append((k, round(c, 6), round(ratio, 4)))
    assert not bad, bad
    whole = cos(torch.cat(all_h), torch.cat(all_o))
    assert whole > pw, whole
    print(f"[B=64 train step {dtype}] batch-hard choices flipped {mining_flips} of 256 (all near-ties < {tie:g}); under the device's "
          f"decisions: worst parameter cos {worst:.6f}, whole gradient {whole:.7f}")
    from tests.conftest import record_measure
    record_measure(f"{dtype}_B64_per_parameter_grad_cos", worst)
    record_measure(f"{dtype}_B64_whole_grad_cos", whole)


def test_fused_adam_matches_torch(dev):
    from signal_amd.solver.make_optimizer import make_optimizer
    ocfg = O.RefConfig(use_a=True, use_b=True)
    sd = O.init_state_dict(ocfg, seed=5)
    model = build(ocfg, sd, dev)
    cfg = model.cfg
    cfg.SOLVER.OPTIMIZER_NAME = "Adam"
    cfg.SOLVER.BASE_LR = 3.5e-4
    opt, _ = make_optimizer(cfg, model, None)
    hip = model.hip
    ref_p = {n: p.detach().clone() for n, p in model.named_parameters()}
    groups = [{"params": [ref_p[n]], "lr": g["lr"], "weight_decay": g["weight_decay"]} for n, g in zip(opt._names, opt.param_groups)]
    topt = torch.optim.Adam(groups)
    gen = torch.Generator(device="cpu").manual_seed(0)
    live = [n for n in opt._names if hip.flat.byname[n].grad is not None]
    assert "SIM.token_selection.W_q.weight" in opt._names and "SIM.token_selection.W_q.weight" not in live
    for it in range(3):
        hip.flat.grad.zero_()
        for n in live:
            gr = torch.randn(ref_p[n].shape, generator=gen).to(dev) * 1e-2
            hip.flat.byname[n].grad.copy_(gr)
            ref_p[n].grad = gr.clone()
        opt.step()
        topt.step()
    for n in live:
        assert rel_err(hip.flat.byname[n].data, ref_p[n]) < 1e-5, n
    # grad-less parameters are untouched (torch skips .grad None; no weight decay either)
    assert torch.equal(hip.flat.byname["SIM.token_selection.W_q.weight"].data.cpu(), sd["SIM.token_selection.W_q.weight"])
    # the bf16 operand mirror follows the update
    n = "clip_vision_encoder.base.transformer.resblocks.3.mlp.c_fc.weight"
    assert torch.equal(hip._pk(n), hip.flat.byname[n].data.to(torch.bfloat16))


def test_reid_loss_hip_vs_reference_fixture(dev, golden):
    """sig_reid_loss (label-smoothed CE + batch-hard soft-margin triplet, fwd + bwd) against fixture G6."""
    from signal_amd.modeling.reid_head import reid_loss
    g = golden("g6_reid")
    gen = O._rng(int(g["seed"]))
    score = O.randn(gen, 16, 171, std=2.0).to(dev).requires_grad_(True)
    feat = O.randn(gen, 16, 1536, std=1.0).to(dev).requires_grad_(True)
    target = (torch.arange(16) // 4 + 7).to(dev)
    loss = reid_loss(score, feat, target, 0.1, 0.25, 1.0, None)
    (2.0 * loss).backward()                      # upstream gradient 2: exercises the device-scalar scaling
    np.testing.assert_allclose(loss.item(), 0.25 * float(g["id_loss"]) + float(g["tri_loss"]), rtol=1e-5)
    np.testing.assert_allclose(score.grad[:2].cpu().numpy() / 2, g["dscore_rows"], rtol=1e-4, atol=1e-8)
    np.testing.assert_allclose(feat.grad.norm(dim=1).cpu().numpy() / 2, g["dfeat_norm"], rtol=1e-4)
    # margin form (MODEL.NO_MARGIN = False) against the oracle
    f2 = feat.detach().clone().requires_grad_(True)
    l2 = reid_loss(score.detach(), f2, target, 0.0, 0.0, 1.0, 0.3)
    l2.backward()
    f3 = feat.detach().cpu().clone().requires_grad_(True)
    l3 = O.triplet_margin(f3, target.cpu(), 0.3)
    l3.backward()
    np.testing.assert_allclose(l2.item(), l3.item(), rtol=1e-5)
    assert rel_err(f2.grad, f3.grad) < 1e-4
    # plain cross entropy (DATALOADER.SAMPLER = 'softmax'): eps 0, no triplet term
    s2 = score.detach().clone().requires_grad_(True)
    l4 = reid_loss(s2, feat.detach(), target, 0.0, 1.0, 0.0, None)
    l4.backward()
    s3 = score.detach().cpu().clone().requires_grad_(True)
    l5 = torch.nn.functional.cross_entropy(s3, target.cpu())
    l5.backward()
    np.testing.assert_allclose(l4.item(), l5.item(), rtol=1e-5)
    assert rel_err(s2.grad, s3.grad) < 1e-4


@pytest.mark.parametrize("B,F,C", [(8, 1536, 171), (64, 512, 50)])
def test_bnneck_classifier_hip_vs_torch(dev, B, F, C):
    from signal_amd.modeling.reid_head import bnneck_classifier
    gen = torch.Generator().manual_seed(B + F)
    x = (torch.randn(B, F, generator=gen) * 2 + 0.3)
    bn_h, bn_t = torch.nn.BatchNorm1d(F), torch.nn.BatchNorm1d(F)
    cl_h, cl_t = torch.nn.Linear(F, C, bias=False), torch.nn.Linear(F, C, bias=False)
    with torch.no_grad():
        bn_h.weight.copy_(1 + 0.1 * torch.randn(F, generator=gen)); bn_t.weight.copy_(bn_h.weight)
        cl_t.weight.copy_(cl_h.weight)
    for bn in (bn_h, bn_t):
        bn.bias.requires_grad_(False)
    bn_h.to(dev); cl_h.to(dev)
    xh = x.to(dev).requires_grad_(True)
    xt = x.clone().requires_grad_(True)
    w = torch.randn(B, C, generator=gen)
    sh = bnneck_classifier(bn_h, cl_h, xh)
    (sh * w.to(dev)).sum().backward()
    st = cl_t(bn_t(xt))
    (st * w).sum().backward()
    assert rel_err(sh, st) < 1e-5
    assert rel_err(xh.grad, xt.grad) < 1e-4
    assert rel_err(bn_h.weight.grad, bn_t.weight.grad) < 1e-4
    assert rel_err(cl_h.weight.grad, cl_t.weight.grad) < 1e-5
    assert bn_h.bias.grad is None
    assert rel_err(bn_h.running_mean, bn_t.running_mean) < 1e-5 and rel_err(bn_h.running_var, bn_t.running_var) < 1e-5
    assert int(bn_h.num_batches_tracked) == 1


def _small_train_setup(dev, dtype="bf16", stage=None, on_cpu_optimizer=False, seed=41):
    from signal_amd.modeling import make_frame
    from signal_amd.solver.make_optimizer import make_optimizer
    from tests.test_model_gpu import make_cfg
    ocfg = O.rgbnt201_config(num_instance=2)
    sd = O.init_state_dict(ocfg, seed=seed, head_scale=30.0)
    cfg = make_cfg(ocfg, dtype)
    cfg.SOLVER.OPTIMIZER_NAME, cfg.SOLVER.BASE_LR = "Adam", 3.5e-4
    if stage is not None:
        cfg.MODEL.stageName = stage
    model = make_frame(cfg, ocfg.num_classes, ocfg.camera_num, 0)
    model.load_state_dict(sd, strict=False)
    opt = None
    if on_cpu_optimizer:                      # the reference's train.py order: optimizer first, model.to(device) later
        opt = make_optimizer(cfg, model, None)[0]
        assert type(opt) is torch.optim.Adam
    model.to(dev)
    img, vid, cam = O.synthetic_batch(ocfg, 4, seed=seed + 1)
    batch = ({k: v.to(dev) for k, v in img.items()}, vid.to(dev), cam.to(dev))
    return ocfg, sd, cfg, model, opt, batch


def test_train_step_adopts_an_optimizer_built_on_the_cpu_model(dev):
    """train.py calls make_optimizer before the model moves to the GPU, so do_train receives a plain
    torch.optim.Adam.  TrainStep must (a) run it as the fused kernel with the SAME param_groups (the scheduler keeps steering
    them), (b) leave the gradient-less selection weights W_q / W_k / W_v bit-unchanged (no weight decay on a zero gradient),
    (c) produce the same parameters as the path whose optimizer was fused from the start."""
    from signal_amd.engine.trainer import TrainStep
    from signal_amd.solver.make_optimizer import FusedAdam
    from signal_amd.solver.scheduler_factory import create_scheduler
    ocfg, sd, cfg, model, opt, batch = _small_train_setup(dev, on_cpu_optimizer=True)
    sched = create_scheduler(cfg, opt)
    ts = TrainStep(cfg, model, num_classes=ocfg.num_classes, optimizer=opt)
    assert isinstance(ts.optimizer, FusedAdam) and ts.optimizer.param_groups is opt.param_groups
    sched.step(1)
    lr_now = [g["lr"] for g in opt.param_groups]
    for _ in range(2):
        ts.step(*batch)
    torch.cuda.synchronize()
    assert [g["lr"] for g in ts.optimizer.param_groups] == lr_now
    for n in ("W_q", "W_k", "W_v"):
        for part in ("weight", "bias"):
            k = f"SIM.token_selection.{n}.{part}"
            assert model.hip.flat.byname[k].grad is None
            assert torch.equal(model.state_dict()[k].cpu(), sd[k]), k
    # the frozen BNNeck biases stay put as well
    assert torch.equal(model.bottleneck.bias.detach().cpu(), sd["bottleneck.bias"])
    # twin: optimizer built on the GPU model (fused from the start), same schedule
    ocfg2, _, cfg2, model2, _, batch2 = _small_train_setup(dev)
    ts2 = TrainStep(cfg2, model2, num_classes=ocfg2.num_classes)
    create_scheduler(cfg2, ts2.optimizer).step(1)
    for _ in range(2):
        ts2.step(*batch2)
    torch.cuda.synchronize()
    a, b = model.hip.flat.data, model2.hip.flat.data
    assert rel_err(a, b) < 1e-4          # (not bit-equal: a few bias / LayerNorm gradients are summed with f32 atomics)
    w = "clip_vision_encoder.base.transformer.resblocks.0.mlp.c_fc.weight"
    assert not torch.equal(model.state_dict()[w].cpu(), sd[w])


def test_stage_cls_leaves_the_sampler_parameters_untouched(dev):
    """In stage 'CLS' only GAM runs (useB.py:181-184): AlignM.DAS_* get no gradient in the reference (grad None), so Adam
    must not touch them -- not even with weight decay."""
    from signal_amd.engine.trainer import TrainStep
    ocfg, sd, cfg, model, _, batch = _small_train_setup(dev, stage="CLS")
    ts = TrainStep(cfg, model, num_classes=ocfg.num_classes)
    for _ in range(2):
        loss = ts.step(*batch)
    assert len(ts.last_output) == 6 and torch.isfinite(loss)
    for k, v in model.state_dict().items():
        if k.startswith("AlignM.DAS_"):
            assert torch.equal(v.cpu(), sd[k]), k
    assert not torch.equal(model.state_dict()["AlignM.contra_temp"].cpu(), sd["AlignM.contra_temp"])


def test_fp16_train_step_with_device_loss_scaler(dev):
    """fp16 operands + the device-resident GradScaler: clean steps apply and count, an overflowing step is skipped (parameters,
    moments, operand mirror bit-unchanged), the scale backs off, and growth doubles it after `growth_interval` clean steps."""
    from signal_amd.engine.trainer import TrainStep
    ocfg, sd, cfg, model, _, batch = _small_train_setup(dev, dtype="fp16")
    ts = TrainStep(cfg, model, num_classes=ocfg.num_classes)
    assert ts.scaler is not None and model.hip.flat.op16.dtype == torch.float16
    ts.scaler.growth_interval = 3
    l0 = float(ts.step(*batch))
    l1 = float(ts.step(*batch))
    st = ts.scaler.state.cpu().tolist()
    assert st[0] == 65536.0 and st[2] == 0.0 and st[3] == 2.0 and st[4] == 2.0, st
    assert l1 < l0                                   # it trains
    # gradient magnitudes are the unscaled ones: compare one step against the bf16 engine on the same weights / batch
    p_before = model.hip.flat.data.clone()
    m_before, mirror_before = ts.optimizer.m.clone(), model.hip.flat.op16.clone()
    # force an overflow: a scale so large that the f16 backward signals saturate
    ts.scaler.state[0] = 2.0 ** 60
    ts.scaler.state[1] = 2.0 ** -60
    ts.step(*batch)
    st = ts.scaler.state.cpu().tolist()
    assert st[0] == 2.0 ** 59 and st[3] == 0.0 and st[4] == 2.0, st        # backed off, tracker reset, no step counted
    assert torch.equal(model.hip.flat.data, p_before) and torch.equal(ts.optimizer.m, m_before)
    assert torch.equal(model.hip.flat.op16, mirror_before)
    # back to a sane scale: three clean steps -> one growth
    ts.scaler.state[0] = 1024.0
    ts.scaler.state[1] = 1.0 / 1024.0
    for _ in range(3):
        ts.step(*batch)
    st = ts.scaler.state.cpu().tolist()
    assert st[0] == 2048.0 and st[3] == 0.0 and st[4] == 5.0, st
    assert not torch.equal(model.hip.flat.data, p_before)
    assert torch.isfinite(model.hip.flat.data).all()


def test_fp16_and_bf16_engines_take_the_same_first_step(dev):
    """One Adam step from the same weights on the same batch: the fp16 engine (scaled backward, unscale in the optimizer)
    and the bf16 engine must move the parameters the same way (Adam's first step is lr * sign-like: compare directions)."""
    from signal_amd.engine.trainer import TrainStep
    res = {}
    for dtype in ("bf16", "fp16"):
        ocfg, sd, cfg, model, _, batch = _small_train_setup(dev, dtype=dtype)
        ts = TrainStep(cfg, model, num_classes=ocfg.num_classes)
        before = model.hip.flat.data.clone()
        ts.step(*batch)
        res[dtype] = (model.hip.flat.data - before).cpu()
    a, b = res["bf16"], res["fp16"]
    live = (a != 0) | (b != 0)
    assert cos(a[live], b[live]) > 0.97


def test_backward_is_reproducible_run_to_run(dev):
    """Two independent runs of the same train step (same weights, same batch).  Every kernel on the TOKEN-gradient path is
    order-deterministic (owner-computes / fixed-order reductions instead of float atomics), so the backward signal that the
    16-bit roundings see is bit-identical run to run.  What is still summed with f32 atomics -- bias / LayerNorm / embedding
    gradients and, at this test's small batch, the weight gradients of the 128x128 TN kernel (at B = 64 the large ones take
    the 256x256 kernel: per-chunk partial tiles + a fixed-order reduce) -- differs in the last bits only: the whole gradient
    agrees to ~1e-7.  Round 1: 1-3e-3, because the scatter order fed back into the 16-bit backward signal."""
    from signal_amd.engine.trainer import TrainStep
    grads, losses = [], []
    for _ in range(2):
        ocfg, sd, cfg, model, _, batch = _small_train_setup(dev, seed=77)
        cfg.SOLVER.BASE_LR = 0.0
        ts = TrainStep(cfg, model, num_classes=ocfg.num_classes)
        losses.append(ts.step(*batch).item())
        torch.cuda.synchronize()
        grads.append(model.hip.flat.grad.clone())
    assert losses[0] == losses[1]                                  # forward + loss: bit-identical
    err = rel_err(grads[0], grads[1])
    print(f"[reproducibility] whole-gradient relative difference between two runs: {err:.2e}")
    assert err < 1e-6


@pytest.mark.parametrize("B", [64, 8])
def test_overwritten_block_weight_gradients_equal_zeroed_then_accumulated(dev, B, monkeypatch):
    """TrainStep does not zero the transformer blocks' weight gradients (340 of flat.grad's 345 MB): the grouped weight-gradient
    launch of the backward overwrites them (sig_tune_tn_overwrite), one sig_zero_ranges launch zeroes everything else.  Three
    steps on three different batches (every learning rate 0: the parameters stay put, so every step's gradient is a function of its
    batch alone -- with moving parameters the f32-atomics noise of two runs is amplified by Adam's sign-like first steps) must give the same losses, bit for bit, and the same final gradient -- to the 1e-6 that two runs of one engine differ by,
    some bias / LayerNorm gradients being summed with f32 atomics -- as the engine that zeroes the whole buffer and accumulates
    (SIGNAL_WGRAD_OVERWRITE=0): a stale or double-counted gradient would be off by a factor.  The un-zeroed part is poisoned with
    NaN first.  B = 64 runs the grouped kernel + reduce, B = 8 the per-weight fallback (which zeroes its outputs itself)."""
    from signal_amd.engine.trainer import TrainStep
    from signal_amd.modeling import make_frame
    ocfg = O.rgbnt201_config() if B == 64 else O.rgbnt201_config(num_instance=4)      # (at least two identities per batch)
    sd = O.init_state_dict(ocfg, seed=31, head_scale=30.0)
    batches = []
    for k in range(3):
        img, vid, cam = O.synthetic_batch(ocfg, B, seed=90 + k)
        batches.append(({m: v.to(dev) for m, v in img.items()}, vid.to(dev), cam.to(dev)))
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("SIGNAL_WGRAD_OVERWRITE", mode)
        cfg = make_cfg(ocfg, "bf16")
        cfg.SOLVER.OPTIMIZER_NAME, cfg.SOLVER.BASE_LR = "Adam", 0.0
        model = make_frame(cfg, ocfg.num_classes, ocfg.camera_num, 0)
        model.load_state_dict(sd, strict=False)
        model.to(dev)
        ts = TrainStep(cfg, model, num_classes=ocfg.num_classes)
        for grp in ts.optimizer.param_groups:       # (BASE_LR = 0 leaves the backbone its fixed 5e-6, solver/make_optimizer.py:20)
            grp["lr"], grp["weight_decay"] = 0.0, 0.0
        assert model.hip.wgrad_overwrite == (mode == "1")
        if mode == "1":       # poison what the engine no longer zeroes: the backward must overwrite every element of it
            for n in model.hip.block_weight_names:
                model.hip._g(n).fill_(float("nan"))
        losses = [ts.step(*b).item() for b in batches]
        torch.cuda.synchronize()
        out[mode] = (losses, model.hip.flat.grad.clone(), model.hip.flat.data.clone())
        del ts, model
    assert out["1"][0] == out["0"][0], (out["1"][0], out["0"][0])
    assert len(set(out["1"][0])) == 3                                   # (three different batches)
    assert bool(torch.isfinite(out["1"][1]).all())
    assert rel_err(out["1"][1], out["0"][1]) < 1e-6, "gradients differ between the overwrite and the zero-and-accumulate engine"
    assert torch.equal(out["1"][2], out["0"][2])                        # (learning rate 0)


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_benched_batch_train_step_big_tiles_vs_small_tiles(dev, dtype):
    """The train step AT THE BENCHED SIZE (RGBNT201, both operand types, B = 64: 320x256 / 256x256 NT tiles with their training epilogues --
    c_fc + saved derivative, GELU' dgrad -- and the grouped weight-gradient kernel) against the same step on the kernels the
    B = 8 oracle comparisons pin (128-row NT tiles, 128x128 weight-gradient tiles).  Same operands, same rounding points,
    another summation order: loss equal to 1e-5, every parameter's gradient cos >= 0.9999, whole gradient >= 0.99999."""
    from signal_amd import _lib
    from signal_amd.engine.trainer import TrainStep
    from signal_amd.modeling import make_frame
    lib = _lib.load()
    ocfg = O.rgbnt201_config()
    sd = O.init_state_dict(ocfg, seed=71, head_scale=30.0)
    img, vid, cam = O.synthetic_batch(ocfg, 64, seed=72)
    batch = ({k: v.to(dev) for k, v in img.items()}, vid.to(dev), cam.to(dev))
    res = {}
    for tag, nt, tn in (("big", 0, 0), ("small", 128, 128)):
        cfg = make_cfg(ocfg, dtype)
        cfg.SOLVER.OPTIMIZER_NAME, cfg.SOLVER.BASE_LR = "Adam", 0.0
        model = make_frame(cfg, ocfg.num_classes, ocfg.camera_num, 0)
        model.load_state_dict(sd, strict=False)
        model.to(dev)
        p_nt, p_tn = lib.sig_tune_gemm_tile(nt), lib.sig_tune_tn_path(tn)
        try:
            ts = TrainStep(cfg, model, num_classes=ocfg.num_classes)
            loss = ts.step(*batch).item()
            torch.cuda.synchronize()
        finally:
            lib.sig_tune_gemm_tile(p_nt); lib.sig_tune_tn_path(p_tn)
        res[tag] = (loss, model.hip.flat.grad.clone(), {n: (model.hip.flat.offsets[n], p.numel()) for n, p in model.hip.flat.byname.items()
                                                        if p.grad is not None})
        del ts, model
    (l0, g0, names), (l1, g1, _) = res["big"], res["small"]
    assert abs(l0 - l1) / abs(l1) < 1e-5, (l0, l1)
    worst = min((cos(g0[o:o + n], g1[o:o + n]), k) for k, (o, n) in names.items() if float(g1[o:o + n].norm()) > 1e-6)
    whole = cos(g0, g1)
    print(f"[B=64 train step {dtype}, big tiles vs small tiles] loss {l0:.6f} / {l1:.6f}, worst parameter cos {worst[0]:.6f} ({worst[1]}), whole {whole:.7f}")
    assert worst[0] > 0.9999, worst
    assert whole > 0.99999, whole
