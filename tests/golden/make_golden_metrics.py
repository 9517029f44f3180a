#!/usr/bin/env python3
"""G10: retrieval metrics.  The reference's utils/metrics.py cannot be IMPORTED here (its module top pulls in seaborn and
scipy.integrate.simps, neither installed -- they stay absent), but its rank statistics are plain numpy / torch functions.  This
script reads the reference file, takes the definitions of `eval_func`, `eval_func_msrv` and `euclidean_distance` out of its syntax tree
(ast; nothing else of the module is executed), runs them on PCG64-seeded synthetic cases and records inputs' seeds + outputs.
Output: tests/golden/g10_metrics.npz (numbers only, no code).  The evaluator classes' compute() (t-SNE / KDE plotting into
hard-coded home directories, metrics.py:289-297) is not run.

    python tests/golden/make_golden_metrics.py
"""
import ast
import os
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("SIGNAL_REFERENCE", "/root/reference")
WANTED = ("eval_func", "eval_func_msrv", "euclidean_distance")


def reference_functions():
    path = os.path.join(REF, "utils", "metrics.py")
    tree = ast.parse(open(path).read(), filename=path)
    keep = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in WANTED]
    assert sorted(n.name for n in keep) == sorted(WANTED)
    ns = {"np": np, "torch": torch}
    exec(compile(ast.Module(body=keep, type_ignores=[]), path, "exec"), ns)
    return {k: ns[k] for k in WANTED}


def cases():
    """(tag, seed, num_query, num_gallery, ids, cams, scenes, dim, max_rank)"""
    return [("small", 1, 12, 40, 6, 3, 4, 32, 10), ("reid201", 2, 60, 300, 30, 4, 5, 128, 50), ("few_gallery", 3, 8, 24, 4, 2, 3, 16, 8),
            ("many_cams", 4, 40, 200, 10, 8, 8, 64, 50)]


def make_case(seed, nq, ng, ids, cams, scenes, dim):
    g = np.random.Generator(np.random.PCG64(seed))
    centers = g.standard_normal((ids, dim)).astype(np.float32)
    def draw(n):
        pid = g.integers(0, ids, size=n)
        feat = centers[pid] + 3.0 * g.standard_normal((n, dim)).astype(np.float32)
        return feat.astype(np.float32), pid.astype(np.int64), g.integers(0, cams, size=n).astype(np.int64), g.integers(0, scenes, size=n).astype(np.int64)
    return draw(nq), draw(ng)


def main():
    R = reference_functions()
    out = {}
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)                      # eval_func_msrv writes ./re.txt (metrics.py:38-39)
        try:
            for tag, seed, nq, ng, ids, cams, scenes, dim, max_rank in cases():
                (qf, qp, qc, qs), (gf, gp, gc, gs) = make_case(seed, nq, ng, ids, cams, scenes, dim)
                qn = torch.nn.functional.normalize(torch.from_numpy(qf), dim=1, p=2)
                gn = torch.nn.functional.normalize(torch.from_numpy(gf), dim=1, p=2)
                dist = R["euclidean_distance"](qn, gn)
                cmc, mAP = R["eval_func"](dist, qp, gp, qc, gc, max_rank=max_rank)
                cmc_s, mAP_s = R["eval_func_msrv"](dist, qp, gp, qc, gc, qs, gs, max_rank=max_rank)
                out[f"{tag}_case"] = np.array([seed, nq, ng, ids, cams, scenes, dim, max_rank], dtype=np.int64)
                out[f"{tag}_dist_rows"] = dist[:3]
                out[f"{tag}_dist_sum"] = np.float64(dist.astype(np.float64).sum())
                out[f"{tag}_cmc"], out[f"{tag}_mAP"] = np.asarray(cmc, np.float32), np.float64(mAP)
                out[f"{tag}_cmc_msrv"], out[f"{tag}_mAP_msrv"] = np.asarray(cmc_s, np.float32), np.float64(mAP_s)
        finally:
            os.chdir(cwd)
    path = os.path.join(HERE, "g10_metrics.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.1f} KiB)")
    for tag, *_ in cases():
        print(tag, "mAP", float(out[f"{tag}_mAP"]), "R1", float(out[f"{tag}_cmc"][0]), "| msrv mAP", float(out[f"{tag}_mAP_msrv"]), "R1", float(out[f"{tag}_cmc_msrv"][0]))


if __name__ == "__main__":
    main()
