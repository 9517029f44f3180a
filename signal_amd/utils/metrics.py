"""Retrieval metrics of the evaluation loop (utils/metrics.py:111-170,222-300,494-500 of the reference):
market1501-style CMC / mAP where, per query, gallery images of the same identity AND camera are discarded.
Vectorised NumPy restatement; the t-SNE/KDE plotting side effects of the reference's compute() (hard-coded
home-directory paths, utils/metrics.py:289-297) are intentionally absent."""
from __future__ import annotations

import numpy as np
import torch


def euclidean_distance(qf: torch.Tensor, gf: torch.Tensor) -> np.ndarray:
    """squared Euclidean distance matrix (utils/metrics.py:494-500)"""
    d = qf.pow(2).sum(1, keepdim=True) + gf.pow(2).sum(1, keepdim=True).t() - 2.0 * qf @ gf.t()
    return d.cpu().numpy()


def eval_func(distmat, q_pids, g_pids, q_camids, g_camids, max_rank=50):
    num_q, num_g = distmat.shape
    max_rank = min(max_rank, num_g)
    order = np.argsort(distmat, axis=1)
    all_cmc, all_ap = [], []
    for q in range(num_q):
        o = order[q]
        keep = ~((g_pids[o] == q_pids[q]) & (g_camids[o] == q_camids[q]))
        m = (g_pids[o] == q_pids[q])[keep].astype(np.int32)
        if not m.any():
            continue
        c = m.cumsum()
        all_cmc.append(np.minimum(c, 1)[:max_rank])
        prec = c / np.arange(1, m.size + 1)
        all_ap.append((prec * m).sum() / m.sum())
    if not all_ap:
        raise AssertionError("Error: all query identities do not appear in gallery")
    cmc = np.zeros(max_rank, np.float32)
    for c in all_cmc:
        cmc[: c.size] += c
        cmc[c.size:] += c[-1]
    return cmc / len(all_ap), float(np.mean(all_ap))


class R1_mAP_eval:
    def __init__(self, num_query, max_rank=50, feat_norm="yes"):
        self.num_query, self.max_rank, self.feat_norm = num_query, max_rank, feat_norm
        self.reset()

    def reset(self):
        self.feats, self.pids, self.camids = [], [], []

    def update(self, output):
        feat, pid, camid = output[0], output[1], output[2]
        self.feats.append(feat.detach().float().cpu())
        self.pids.extend(np.asarray(pid).tolist())
        self.camids.extend(np.asarray(camid).tolist())

    def compute(self):
        feats = torch.cat(self.feats, dim=0)
        if self.feat_norm == "yes":
            feats = torch.nn.functional.normalize(feats, dim=1, p=2)
        qf, gf = feats[: self.num_query], feats[self.num_query:]
        q_pids, g_pids = np.asarray(self.pids[: self.num_query]), np.asarray(self.pids[self.num_query:])
        q_cam, g_cam = np.asarray(self.camids[: self.num_query]), np.asarray(self.camids[self.num_query:])
        distmat = euclidean_distance(qf, gf)
        cmc, mAP = eval_func(distmat, q_pids, g_pids, q_cam, g_cam, self.max_rank)
        return cmc, mAP, distmat, self.pids, self.camids, qf, gf
