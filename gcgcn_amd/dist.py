"""Batch data-parallelism for the graph blocks: one process per GPU, documents sharded along the
batch axis, ONE fp32 all-reduce (RCCL over xGMI through torch.distributed backend "nccl") of a single
contiguous gradient bucket per step.

The reference has no distributed code at all; its "batch" is gradient accumulation
over documents followed by one backward of ``total_loss / batch_size`` (config/Config.py:366-373).
Summing per-rank gradients and dividing by the global document count reproduces exactly that.

Because every block already keeps its parameters (and therefore its gradients) in one flat buffer,
bucketing is a re-pointing of storage, not a copy: all ``.flat`` parameters become views of one
parameter arena and all ``.flat.grad`` views of one gradient arena.  The HIP backward kernels write
into fresh tensors that autograd accumulates into those views in place, so after ``backward()`` the
arena IS the all-reduce operand.  ``linears_k.*`` never receive gradients (reference quirk) and are
left out of the bucket on every rank alike.
"""
from __future__ import annotations

from typing import Iterable, List, Optional

import torch
import torch.distributed as dist
from torch import nn


class FlatGradBucket:
    """overlap=False: one all-reduce of the whole arena after backward (``all_reduce()``).
    overlap=True : each block's slice is all-reduced asynchronously the moment autograd has finished
    accumulating that block's gradient (post-accumulate hook), so the MAGGC gradients travel over xGMI while
    the MHA / CAGGC / GAT backward kernels still run; ``all_reduce()`` then only waits for the handles.
    Every rank issues the slices in the same (reverse-topological) order, as RCCL requires."""

    def __init__(self, module: nn.Module, process_group=None, overlap: bool = False):
        self.pg = process_group
        self.overlap = overlap
        self._pending = []
        self.params: List[nn.Parameter] = [p for n, p in module.named_parameters()
                                           if p.requires_grad and not n.endswith("flat_k")]
        # every parameter starts on a 256-byte boundary so the GEMM kernels keep their 16-byte vector loads
        pad = lambda n: (n + 63) // 64 * 64
        total = sum(pad(p.numel()) for p in self.params)
        dev = self.params[0].device
        self.param_arena = torch.empty(total, device=dev, dtype=torch.float32)
        self.grad_arena = torch.zeros(total, device=dev, dtype=torch.float32)
        off = 0
        with torch.no_grad():
            for p in self.params:
                n = p.numel()
                self.param_arena[off:off + n].copy_(p.reshape(-1))
                p.data = self.param_arena[off:off + n].view_as(p)
                p.grad = self.grad_arena[off:off + n].view_as(p)
                if overlap:
                    p.register_post_accumulate_grad_hook(self._make_hook(off, n))
                off += pad(n)
        self.numel = total

    def _distributed(self) -> bool:
        return dist.is_available() and dist.is_initialized() and dist.get_world_size(self.pg) > 1

    def _make_hook(self, off: int, n: int):
        def hook(param):
            if self._distributed():
                self._pending.append(dist.all_reduce(self.grad_arena[off:off + n], op=dist.ReduceOp.SUM,
                                                     group=self.pg, async_op=True))
        return hook

    def zero_grad(self):
        self.grad_arena.zero_()
        for p in self.params:          # keep the views attached (zero_grad(set_to_none=True) would detach them)
            if p.grad is None:
                raise RuntimeError("FlatGradBucket: a gradient view was detached; do not set grads to None")

    def all_reduce(self, global_docs: Optional[int] = None, async_op: bool = False):
        """Sum gradients over ranks (in place, one collective); optionally divide by the global document count
        (the reference's total_loss / batch_size)."""
        work = None
        if self.overlap:
            for w in self._pending:      # the slices were launched from the backward hooks
                w.wait()
            self._pending.clear()
        elif self._distributed():
            work = dist.all_reduce(self.grad_arena, op=dist.ReduceOp.SUM, group=self.pg, async_op=async_op)
        if global_docs is not None and not async_op:
            self.grad_arena.div_(float(global_docs))
        return work


def shard_batch(tensors: Iterable[torch.Tensor], rank: int, world: int) -> List[torch.Tensor]:
    """Contiguous shard of the document axis for this rank (global B must divide evenly)."""
    out = []
    for t in tensors:
        B = t.shape[0]
        if B % world != 0:
            raise ValueError(f"global batch {B} is not divisible by world size {world}")
        k = B // world
        out.append(t[rank * k:(rank + 1) * k])
    return out
