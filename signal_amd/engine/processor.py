"""Train / inference loops with the reference's signatures (engine/processor.py:41-50,353-359,454-540):

    do_train(cfg, model, center_criterion, train_loader, val_loader, optimizer, optimizer_center, scheduler,
             loss_fn, num_query, local_rank, stage)
    do_inference(cfg, model, val_loader, num_query, logger, sge, local_rank)

so the reference's train.py / test.py call them unchanged.  Differences, all on the host side: the GradScaler of
processor.py:119,259-261 lives on the device (fp16 operand mode; bf16 needs none); gradients are reduced by signal_amd.parallel.GradReducer instead of DistributedDataParallel; rank 0
evaluates/saves and the other ranks wait at a barrier (the reference lets them run ahead into the next
all-reduce); checkpoints never carry a 'module.' prefix."""
from __future__ import annotations

import logging
import os
import time

import torch
import torch.distributed as dist

from ..data.pipeline import DevicePrefetcher
from ..utils.metrics import R1_mAP, R1_mAP_eval, make_evaluator
from .trainer import TrainStep


class AverageMeter:
    def __init__(self):
        self.reset()

    def reset(self):
        self.sum, self.count, self.avg = 0.0, 0, 0.0

    def update(self, val, n=1):
        self.sum += float(val) * n
        self.count += n
        self.avg = self.sum / max(self.count, 1)


def _to_dev(img, device):
    return {k: img[k].to(device, non_blocking=True) for k in ("RGB", "NI", "TI")}


def do_train(cfg, model, center_criterion, train_loader, val_loader, optimizer, optimizer_center, scheduler, loss_fn,
             num_query, local_rank, stage, batch_hook=None):
    log_period, ckpt_period, eval_period = cfg.SOLVER.LOG_PERIOD, cfg.SOLVER.CHECKPOINT_PERIOD, cfg.SOLVER.EVAL_PERIOD
    if not torch.cuda.is_available():
        raise RuntimeError("signal_amd trains on an MI355X only (no CPU path)")
    device = torch.device(f"cuda:{local_rank}")
    logger = logging.getLogger("Signal.train")
    logger.info("start training")
    model.to(device)
    world = dist.get_world_size() if (cfg.MODEL.DIST_TRAIN and dist.is_initialized()) else 1
    rank = dist.get_rank() if world > 1 else 0
    # `optimizer` normally is the torch.optim.Adam that train.py built on the CPU model (train.py:85): TrainStep adopts its
    # param_groups (the scheduler keeps steering them) and applies the update with the fused kernel
    engine = TrainStep(cfg, model, num_classes=model.num_classes, world_size=world, loss_fn=loss_fn, optimizer=optimizer,
                       stage=stage)
    loss_meter, acc_meter = AverageMeter(), AverageMeter()
    evaluator = make_evaluator(cfg, num_query)          # MSVR310: the scene protocol (processor.py:112-116)
    best = {"mAP": 0.0, "Rank-1": 0.0, "Rank-5": 0.0, "Rank-10": 0.0}
    out_dir = os.path.join(cfg.OUTPUT_DIR, cfg.ckpt_save_path)
    os.makedirs(out_dir, exist_ok=True)
    for epoch in range(1, cfg.SOLVER.MAX_EPOCHS + 1):
        t0 = time.time()
        loss_meter.reset()
        acc_meter.reset()
        scheduler.step(epoch)
        model.train()
        n_iter = -1
        # The reference's loader is pin_memory=True (make_dataloader.py:224) and processor.py:155-162 copies each batch at the top
        # of the iteration, on the compute stream.  Here batch i+1 crosses PCIe on a side stream while step i computes (75.5 MB
        # per step at B = 64); the compute stream only waits on the copy's event.  `batch_hook` (tests) sees what the engine gets.
        for n_iter, (img, target, target_cam, target_view, _) in enumerate(DevicePrefetcher(train_loader, device)):
            if batch_hook is not None:
                batch_hook(epoch, n_iter, img, target, target_cam, target_view)
            loss = engine.step(img, target, target_cam, target_view)
            if (n_iter + 1) % log_period == 0:        # the only host sync: every LOG_PERIOD iterations
                score = engine.last_output[1]
                acc = (score.max(1)[1] == target).float().mean()
                loss_meter.update(loss.item(), img["RGB"].shape[0])
                acc_meter.update(acc.item(), 1)
                logger.info("Epoch[{}] Iteration[{}/{}] Loss: {:.3f}, Acc: {:.3f}, Base Lr: {:.2e}".format(
                    epoch, n_iter + 1, len(train_loader), loss_meter.avg, acc_meter.avg, scheduler._get_lr(epoch)[0]))
        torch.cuda.synchronize()
        per_batch = (time.time() - t0) / max(n_iter + 1, 1)
        if rank == 0:
            logger.info("Epoch {} done. Time per batch: {:.3f}[s] Speed: {:.1f}[samples/s]".format(
                epoch, per_batch, world * train_loader.batch_size / per_batch))
        if epoch % ckpt_period == 0 and rank == 0:
            torch.save(model.state_dict(), os.path.join(out_dir, cfg.MODEL.NAME + "_{}.pth".format(epoch)))
        if epoch % eval_period == 0:
            if rank == 0:
                mAP, cmc = training_neat_eval(cfg, model, val_loader, device, evaluator, epoch, logger, sge=stage)
                if mAP >= best["mAP"]:
                    best.update({"mAP": mAP, "Rank-1": cmc[0], "Rank-5": cmc[4], "Rank-10": cmc[9]})
                    torch.save(model.state_dict(), os.path.join(out_dir, cfg.MODEL.NAME + "best.pth"))
                logger.info("~" * 50)
                for k in ("mAP", "Rank-1", "Rank-5", "Rank-10"):
                    logger.info("Best {}: {:.1%}".format(k, best[k]))
                logger.info("~" * 50)
            if world > 1:
                dist.barrier()


def _run_eval(model, val_loader, device, evaluator, sge):
    evaluator.reset()
    model.eval()
    for batch in val_loader:
        img, pid, camid, camids, target_view = batch[0], batch[1], batch[2], batch[3], batch[4]
        with torch.no_grad():
            feat = model(_to_dev(img, device), cam_label=camids.to(device), view_label=target_view.to(device),
                         training=False, sge=sge)
        if isinstance(evaluator, R1_mAP):               # MSVR310: the view label carries the scene id (processor.py:421,430-433)
            evaluator.update((feat, pid, camid, target_view))
        else:
            evaluator.update((feat, pid, camid))
    cmc, mAP = evaluator.compute()[:2]
    return cmc, mAP


def training_neat_eval(cfg, model, val_loader, device, evaluator, epoch, logger, sge="CLS"):
    cmc, mAP = _run_eval(model, val_loader, device, evaluator, sge)
    logger.info("Validation Results - Epoch: {}".format(epoch))
    logger.info("mAP: {:.1%}".format(mAP))
    for r in (1, 5, 10):
        logger.info("CMC curve, Rank-{:<3}:{:.1%}".format(r, cmc[r - 1]))
    torch.cuda.empty_cache()
    return mAP, cmc


def do_inference(cfg, model, val_loader, num_query, logger, sge, local_rank):
    if not torch.cuda.is_available():
        raise RuntimeError("signal_amd runs inference on an MI355X only (no CPU path)")
    device = torch.device(f"cuda:{local_rank}")
    logger = logging.getLogger("Signal.test")
    logger.info("Enter inferencing")
    evaluator = make_evaluator(cfg, num_query)          # processor.py:385-389
    model.to(device)
    cmc, mAP = _run_eval(model, val_loader, device, evaluator, sge)
    logger.info("Validation Results ")
    logger.info("mAP: {:.1%}".format(mAP))
    for r in (1, 5, 10):
        logger.info("CMC curve, Rank-{:<3}:{:.1%}".format(r, cmc[r - 1]))
    return cmc[0], cmc[4]
