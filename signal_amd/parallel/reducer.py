"""Data-parallel gradient exchange: the reference wraps the model in
DistributedDataParallel(find_unused_parameters=True); here the gradients already live in ONE flat f32 buffer,
so the reducer is a bucket plan over that buffer -- one ~28 MB bucket per transformer block, issued as an
asynchronous RCCL all-reduce (torch.distributed 'nccl' backend = RCCL over xGMI) the moment the block's
backward has been enqueued, overlapping with the backward of the blocks below.  The remainder is split by WHEN its
gradients are complete: everything on the head side (ln_post, proj, SIM, AlignM, BNNeck, classifiers: ~22 MB) is final
once the head stage of the ViT backward has run, so it is reduced right then, under the twelve blocks; only the
embedding group (conv1, class / positional / camera embeddings, ln_pre: ~3 MB) waits for the end of the backward.
Sum all-reduce; the 1/world_size average is folded into the
optimizer's grad_scale.  Gradient-less parameters (SIM.token_selection.*) are excluded statically instead of
find_unused_parameters' per-step bitmap exchange."""
from __future__ import annotations

from typing import Callable, Dict, List, Sequence, Tuple

import torch
import torch.distributed as dist


def embedding_side(name: str) -> bool:
    """Parameters whose gradient is produced by the LAST stage of the backward (sig_embed_bwd)."""
    return name.endswith(("cv_embed", "base.class_embedding", "base.positional_embedding", "base.conv1.weight",
                          "base.ln_pre.weight", "base.ln_pre.bias"))


def split_rest(names: Sequence[str], offsets: Dict[str, int], sizes: Dict[str, int], skip: Callable[[str], bool] = lambda n: False,
               late_names=None):
    """-> (early, late): ranges of the non-block, non-skipped parameters.  late = the parameters whose gradient the LAST stage
    of the backward writes: `late_names` (the engine's own list of what sig_embed_bwd is bound to, HipPath.embed_param_names)
    or, without it, embedding_side()'s name rule."""
    pad = lambda n: (n + 63) // 64 * 64
    is_late = embedding_side if late_names is None else set(late_names).__contains__
    out: Dict[bool, List[Tuple[int, int]]] = {False: [], True: []}
    for n in names:
        if ".transformer.resblocks." in n or skip(n):
            continue
        lst, lo, hi = out[bool(is_late(n))], offsets[n], offsets[n] + pad(sizes[n])
        if lst and lst[-1][1] == lo:
            lst[-1] = (lst[-1][0], hi)
        else:
            lst.append((lo, hi))
    return out[False], out[True]


def plan_buckets(names: Sequence[str], offsets: Dict[str, int], sizes: Dict[str, int], total: int,
                 skip: Callable[[str], bool] = lambda n: False):
    """-> (block_buckets {layer: (lo, hi)}, rest [(lo, hi), ...]) element ranges of the flat buffer.
    A block bucket is the contiguous range of 'transformer.resblocks.<i>.' parameters; `rest` covers every other
    non-skipped parameter, merged into maximal contiguous ranges (split_rest() divides it into the part that is final
    when the ViT backward starts and the embedding part)."""
    pad = lambda n: (n + 63) // 64 * 64
    blocks: Dict[int, List[int]] = {}
    other: List[Tuple[int, int]] = []
    for n in names:
        lo, hi = offsets[n], offsets[n] + pad(sizes[n])
        if ".transformer.resblocks." in n:
            i = int(n.split(".transformer.resblocks.")[1].split(".")[0])
            b = blocks.setdefault(i, [lo, hi])
            if lo != b[1] and lo != b[0]:
                if lo < b[0] or lo > b[1]:
                    raise RuntimeError(f"block {i} parameters are not contiguous in the flat buffer")
            b[0], b[1] = min(b[0], lo), max(b[1], hi)
        elif not skip(n):
            if other and other[-1][1] == lo:
                other[-1] = (other[-1][0], hi)
            else:
                other.append((lo, hi))
    return {i: (b[0], b[1]) for i, b in blocks.items()}, other


class GradReducer:
    def __init__(self, flat_grad: torch.Tensor, block_buckets: Dict[int, Tuple[int, int]], rest: List[Tuple[int, int]],
                 group=None, rest_early: List[Tuple[int, int]] = None, force: bool = False):
        """rest = ranges reduced by finish(); rest_early (optional) = ranges reduced by on_head_ready().
        force: issue the collectives even in a one-rank group (the sum over one rank is the identity) -- lets a single GPU
        execute the whole exchange path (RCCL kernels on their side stream under the backward) for tests and measurements."""
        self.g, self.blocks, self.rest, self.group = flat_grad, block_buckets, rest, group
        self.rest_early = rest_early or []
        self._early_done = False
        self.pending = []
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.active = self.world > 1 or (force and dist.is_initialized())

    def _reduce(self, lo, hi):
        self.pending.append(dist.all_reduce(self.g[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def on_head_ready(self):
        """Call right after the head stage of the ViT backward was enqueued: every head-side gradient is final."""
        if not self.active or self._early_done:
            return
        for lo, hi in self.rest_early:
            self._reduce(lo, hi)
        self._early_done = True

    def on_block_ready(self, layer: int):
        """Call right after block `layer`'s backward was enqueued on the current stream."""
        if not self.active or layer not in self.blocks:
            return
        lo, hi = self.blocks[layer]
        self._reduce(lo, hi)

    def finish(self):
        """Reduce the remainder and make the current stream wait for every outstanding bucket."""
        if not self.active:
            return
        self.on_head_ready()                 # (a backward that never reached the head hook, e.g. no backbone gradient)
        for lo, hi in self.rest:
            self._reduce(lo, hi)
        for w in self.pending:
            w.wait()
        self.pending.clear()
        self._early_done = False

    def broadcast_params(self, flat_data: torch.Tensor, src: int = 0):
        """Initial parameter sync (DDP construction broadcast)."""
        if self.active:
            dist.broadcast(flat_data, src=src, group=self.group)

    def broadcast_buffers(self, buffers: Sequence[torch.Tensor], src: int = 0):
        """DistributedDataParallel(broadcast_buffers=True), the reference's wrapper (engine/processor.py:100-105): module
        buffers -- here the BNNecks' running_mean / running_var / num_batches_tracked -- are overwritten by rank `src`'s at
        construction and at the start of EVERY forward, so every rank carries rank 0's running statistics (they never enter
        the training loss; a checkpoint or an in-training evaluation on any rank then reads the same numbers).  The float
        buffers travel as one flat tensor, the integer counters as another: two small broadcasts per step."""
        if self.world == 1:
            return
        for kind in (True, False):
            grp = [b for b in buffers if b.is_floating_point() == kind and b.numel()]
            if not grp:
                continue
            flat = torch.cat([b.reshape(-1) for b in grp])
            dist.broadcast(flat, src=src, group=self.group)
            off = 0
            for b in grp:
                b.copy_(flat[off:off + b.numel()].view_as(b))
                off += b.numel()
