"""Input side of the train step: the identity-aware P x K batch sampler with the reference's
data-parallel sharding rule, a synthetic triplet source of the metric's shape, and a pinned-memory H2D prefetcher.

Reference behaviour mirrored here:
  * data/datasets/sampler_ddp.py:118-202 RandomIdentitySampler_DDP -- draw (mini_batch / NUM_INSTANCE) identities
    x NUM_INSTANCE images per mini-batch block from a seed shared by all ranks, then rank r takes the blocks
    [mb*r + mb*W*i, mb*r + mb*W*i + mb) of the global list (:165-175); BatchSampler(drop_last=True)
    (make_dataloader.py:215-218).
  * the shared seed: the reference all-gathers a pickled int over a gloo side group every epoch (:105-115); here the
    seed is a pure function of (base_seed, epoch), so no collective is needed at all.  Given the same seed the index lists
    are bit-identical to the reference's (tests/golden/g9_sampler.npz).
  * host -> device copies (engine/processor.py:155-162) are blocking .to(device) calls there; here batches are staged
    in pinned memory and copied on a side HIP stream one step ahead (75.5 MB per B=64 step of f32 triplets)."""
from __future__ import annotations

import copy
from collections import defaultdict
from typing import Dict, Iterable, Iterator, List, Sequence

import numpy as np
import torch


def shard_for_rank(indices: Sequence[int], mini_batch: int, rank: int, world: int) -> List[int]:
    """sampler_ddp.py:165-175: rank r keeps blocks r, r+W, r+2W, ... of size mini_batch; the ragged tail is dropped so
    every rank sees the same number of full blocks."""
    total = len(indices)
    length = -(-total // world)
    blocks = length // mini_batch
    out: List[int] = []
    for i in range(blocks):
        lo = mini_batch * rank + mini_batch * world * i
        out.extend(indices[lo:min(lo + mini_batch, total)])
    return out


def epoch_seed(base_seed: int, epoch: int) -> int:
    """The per-epoch seed every rank uses.  The reference draws np.random.randint(2**31) on each rank and all-gathers rank
    0's value over a gloo side group (sampler_ddp.py:105-115); a pure function of (base seed, epoch) gives every rank the
    same value with no collective."""
    return int(np.random.RandomState([int(base_seed) & 0xFFFFFFFF, int(epoch) & 0xFFFFFFFF]).randint(2 ** 31))


class PKSampler:
    """P identities x K instances per mini-batch, sharded over ranks; iterate to get this rank's sample indices
    (feed to torch BatchSampler(drop_last=True) or use batches()).

    For a given shared seed the index lists are the reference's (RandomIdentitySampler_DDP.sample_list and
    __fetch_current_node_idxs, sampler_ddp.py:154-199: same legacy-MT19937 stream, same call order -- pinned by
    tests/golden/g9_sampler.npz, which records that file's own output for world sizes 1, 2 and 8)."""

    def __init__(self, pids: Sequence[int], batch_size: int, num_instances: int, rank: int = 0, world: int = 1,
                 seed: int = 1234):
        if batch_size % world or (batch_size // world) % num_instances:
            raise ValueError("IMS_PER_BATCH must split into world_size mini-batches of whole identities")
        self.k, self.rank, self.world, self.seed = num_instances, rank, world, seed
        self.mini_batch = batch_size // world
        self.p = self.mini_batch // num_instances
        self.index_dic: Dict[int, List[int]] = defaultdict(list)
        for i, pid in enumerate(pids):
            self.index_dic[int(pid)].append(i)
        self.pids = list(self.index_dic)
        self.epoch = 0
        self.shared_seed = None

    def set_epoch(self, epoch: int, shared_seed: int = None):
        """shared_seed: use this value instead of epoch_seed(seed, epoch) (e.g. one exchanged the reference's way)."""
        self.epoch, self.shared_seed = epoch, shared_seed

    def global_list(self) -> List[int]:
        seed = self.shared_seed if self.shared_seed is not None else epoch_seed(self.seed, self.epoch)
        rs = np.random.RandomState(seed)                   # = np.random.seed(seed) + the module-level functions
        avail = copy.copy(self.pids)
        pools: Dict[int, List[int]] = {}
        out: List[int] = []
        while len(avail) >= self.p:
            for pid in rs.choice(avail, self.p, replace=False).tolist():
                if pid not in pools:
                    idxs = list(self.index_dic[pid])
                    if len(idxs) < self.k:
                        idxs = rs.choice(idxs, size=self.k, replace=True).tolist()
                    rs.shuffle(idxs)
                    pools[pid] = idxs
                pool = pools[pid]
                out.extend(pool[: self.k])
                del pool[: self.k]
                if len(pool) < self.k:
                    avail.remove(pid)
        return out

    def __iter__(self) -> Iterator[int]:
        return iter(shard_for_rank(self.global_list(), self.mini_batch, self.rank, self.world))

    def __len__(self) -> int:
        return len(shard_for_rank(self.global_list(), self.mini_batch, self.rank, self.world))

    def batches(self) -> Iterator[List[int]]:
        mine = list(self)
        for i in range(0, len(mine) - self.mini_batch + 1, self.mini_batch):
            yield mine[i:i + self.mini_batch]


class SyntheticTriplets:
    """Synthetic RGB+NIR+TIR triplets of the metric's shape: N(0,1) f32 [3,H,W] per modality, P x K
    identity blocks, camera ~ U{0..cams-1}; yields the reference's train collate tuple
    (img dict, vid, target_cam, target_view, paths) on the HOST (pinned when asked)."""

    def __init__(self, batch: int, hw=(256, 128), num_instances: int = 8, cams: int = 4, steps: int = 10, seed: int = 1234,
                 pin: bool = False):
        self.batch, self.hw, self.k, self.cams, self.steps, self.seed, self.pin = batch, hw, num_instances, cams, steps, seed, pin

    def __len__(self):
        return self.steps

    @property
    def batch_size(self):            # torch DataLoader's attribute; engine/processor.py:293-302 reads it for the speed log
        return self.batch

    def __iter__(self):
        g = torch.Generator().manual_seed(self.seed)
        for _ in range(self.steps):
            img = {m: torch.randn(self.batch, 3, *self.hw, generator=g) for m in ("RGB", "NI", "TI")}
            vid = torch.arange(self.batch) // self.k
            cam = torch.randint(0, self.cams, (self.batch,), generator=g)
            view = torch.zeros(self.batch, dtype=torch.int64)
            if self.pin:
                img = {k: v.pin_memory() for k, v in img.items()}
            yield img, vid, cam, view, None


class DevicePrefetcher:
    """Wraps any iterable of (img dict, vid, cam, view, extra) host batches: copies batch i+1 to the device on a side
    stream while step i computes.  Yields device batches; the consumer's stream waits on the copy event only."""

    def __init__(self, loader: Iterable, device: torch.device):
        self.loader, self.device = loader, device
        self.stream = torch.cuda.Stream(device=device)

    def _stage(self, batch):
        img, vid, cam, view, extra = batch
        with torch.cuda.stream(self.stream):
            dimg = {k: (v if v.is_pinned() else v.pin_memory()).to(self.device, non_blocking=True) for k, v in img.items()}
            out = (dimg, vid.to(self.device, non_blocking=True), cam.to(self.device, non_blocking=True),
                   view.to(self.device, non_blocking=True), extra)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        return out, ev

    def __iter__(self):
        it = iter(self.loader)
        try:
            nxt = self._stage(next(it))
        except StopIteration:
            return
        while nxt is not None:
            cur, ev = nxt
            try:
                nxt = self._stage(next(it))
            except StopIteration:
                nxt = None
            torch.cuda.current_stream(self.device).wait_event(ev)
            for t in list(cur[0].values()) + [cur[1], cur[2], cur[3]]:
                t.record_stream(torch.cuda.current_stream(self.device))
            yield cur

    def __len__(self):
        return len(self.loader)
