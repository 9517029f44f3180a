#!/usr/bin/env python3
"""G9: data-parallel identity sampler.  Runs the reference's own data/datasets/sampler_ddp.py (imported by path from
/root/reference; plain torch / numpy) over a synthetic identity table and records, for several (world size, batch size,
instances) settings and two epochs, the shared seed the reference drew and the index list every rank iterates.
World size 1 runs the file untouched inside a real single-process gloo group (the seed all-gather included); for world
sizes 2 and 8 `dist.get_world_size` / `dist.get_rank` are pointed at the emulated rank and `shared_random_seed` returns the
seed drawn on rank 0 -- what its all-gather returns on every rank; `sample_list` and `__fetch_current_node_idxs`
(sampler_ddp.py:154-199) run unmodified.  Output: tests/golden/g9_sampler.npz (numbers only).

    python tests/golden/make_golden_sampler.py
"""
import importlib.util
import os
import socket

import numpy as np
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("SIGNAL_REFERENCE", "/root/reference")


def identity_table(seed=0, n_ids=23):
    """pid per sample: 23 identities with 2..13 images each (some below NUM_INSTANCE: the replace=True branch)."""
    rng = np.random.default_rng(seed)
    counts = rng.integers(2, 14, size=n_ids)
    pids = np.concatenate([np.full(c, 100 + 3 * i) for i, c in enumerate(counts)])
    return rng.permutation(pids)


def main():
    spec = importlib.util.spec_from_file_location("ref_sampler_ddp", os.path.join(REF, "data", "datasets", "sampler_ddp.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    pids = identity_table()
    source = [(f"img{i}", int(p), 0, 0) for i, p in enumerate(pids)]
    out = {"pids": pids.astype(np.int64)}
    cases = [(1, 16, 4), (2, 32, 4), (8, 64, 2), (2, 24, 4), (1, 64, 8)]
    out["cases"] = np.array(cases, dtype=np.int64)
    real_ws, real_rank, real_seed = mod.dist.get_world_size, mod.dist.get_rank, mod.shared_random_seed
    for ci, (world, bs, k) in enumerate(cases):
        for epoch in range(2):
            np.random.seed(1000 * ci + epoch)             # state of rank 0's global numpy RNG before the epoch
            if world == 1:
                s = mod.RandomIdentitySampler_DDP(source, bs, k)
                lists = [np.array(list(iter(s)), dtype=np.int64)]
                seed = s._seed
            else:
                seed = int(np.random.randint(2 ** 31))    # the value shared_random_seed() returns on every rank (rank 0's draw)
                lists = []
                for r in range(world):
                    mod.dist.get_world_size, mod.dist.get_rank = (lambda w=world: w), (lambda r=r: r)
                    mod.shared_random_seed = lambda seed=seed: seed
                    try:
                        s = mod.RandomIdentitySampler_DDP(source, bs, k)
                        lists.append(np.array(list(iter(s)), dtype=np.int64))
                        assert s._seed == seed
                    finally:
                        mod.dist.get_world_size, mod.dist.get_rank, mod.shared_random_seed = real_ws, real_rank, real_seed
            out[f"c{ci}_e{epoch}_seed"] = np.int64(seed)
            assert len({len(l) for l in lists}) == 1
            out[f"c{ci}_e{epoch}_ranks"] = np.stack(lists)
    dist.destroy_process_group()
    path = os.path.join(HERE, "g9_sampler.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path} ({os.path.getsize(path)} bytes)")


if __name__ == "__main__":
    main()
