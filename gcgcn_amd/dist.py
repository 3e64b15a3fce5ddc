"""Batch data-parallelism for the graph blocks: one process per GPU, documents sharded along the
batch axis, gradients summed with RCCL over xGMI through torch.distributed (backend "nccl").

The reference has no distributed code at all; its "batch" is gradient accumulation
over documents followed by one backward of ``total_loss / batch_size`` (config/Config.py:366-373).
Summing per-rank gradients and dividing by the global document count reproduces exactly that.

Every block keeps its parameters -- and therefore its gradient -- in ONE flat fp32 tensor in kernel
layout (params.py), so the gradients ARE the buckets: no flattening copies, 4 tensors for the whole path
instead of the reference's 48 parameters.  The HIP backward writes a block's gradient into a fresh
contiguous tensor that autograd installs as ``.grad`` without a copy (``.grad`` is reset to None each
step), and

* overlap=False (default): ``all_reduce()`` sums the four tensors after backward in ONE coalesced,
  synchronous collective (one RCCL group launch on the compute stream's timeline).  Measured on MI355X
  with a 1-rank RCCL group (cfg 2, 0.68 ms step): +0.02 ms, against +0.09 ms for four synchronous
  collectives and +0.29 ms for four asynchronous ones -- every asynchronous collective pays a fork and a
  join between the compute stream and the collective stream, which costs more than overlapping an
  8.5 MB all-reduce over xGMI can win back at this step length.
* overlap=True: a post-accumulate hook launches the asynchronous all-reduce of a block's tensor the
  moment it exists (the MAGGC gradient travels while the MHA / CAGGC / GAT backward kernels still run);
  ``all_reduce()`` only waits for the handles.  Every rank issues the collectives in the same
  (reverse-topological) order, as RCCL requires.  For much longer steps (large N, large D).

``linears_k.*`` never receive gradients (reference quirk) and are skipped on every rank
alike.  Every OTHER bucketed parameter takes part in every step's collective on every rank: a rank whose shard
produced no gradient for a block contributes zeros (a per-rank skip would issue a different collective sequence
and hang RCCL).

Which mode at 8 GPUs (model, not yet measured -- DESIGN.md section 7): the cfg-2 bucket is 9 MB; a ring all-reduce
over xGMI moves 2 * 7/8 * 9 MB per link at ~153 GB/s/link = ~0.10 ms (+ ~0.03 ms launch/latency) against a 0.6 ms
step.  Exposed (overlap=False) that is a ~18 % scaling loss at most; overlap=True can hide the MAGGC block's share
(80 % of the bytes) behind the ~0.25 ms of MHA / CAGGC / GAT backward that follows it, but it needs tensor hooks,
which switch deferred weight gradients off for the hooked parameters (functional._pass_for_parking): +0.03 ms of
compute, and +0.29 ms of stream fork/join was measured for four async collectives on one rank.  So: overlap=False
with deferral stays the default; ``bench.py --overlap-grads`` flips it for the driver's scaling run to compare.
"""
from __future__ import annotations

import os
from typing import Iterable, List, Optional

import torch
import torch.distributed as dist
from torch import nn


class FlatGradBucket:
    def __init__(self, module: nn.Module, process_group=None, overlap: bool = False):
        self.pg = process_group
        self.overlap = overlap
        self.active = True          # False: no collectives at all (steps that only some ranks run, e.g. profiling samples)
        self._pending = []
        self.params: List[nn.Parameter] = [p for n, p in module.named_parameters()
                                           if p.requires_grad and not n.endswith("flat_k")]
        self.numel = sum(p.numel() for p in self.params)
        if overlap:
            for p in self.params:
                p.register_post_accumulate_grad_hook(self._hook)

    def _distributed(self) -> bool:
        # GCGCN_FORCE_DIST=1: issue the collectives even in a 1-rank group (exercises the RCCL path on one GPU)
        if not (self.active and dist.is_available() and dist.is_initialized()):
            return False
        return dist.get_world_size(self.pg) > 1 or os.environ.get("GCGCN_FORCE_DIST") == "1"

    def _hook(self, param):
        if self._distributed():
            self._pending.append(dist.all_reduce(param.grad, op=dist.ReduceOp.SUM, group=self.pg, async_op=True))

    def _all_reduce_coalesced(self, grads):
        if not grads:
            return
        cm = getattr(dist, "_coalescing_manager", None)
        if cm is not None and grads[0].is_cuda and len(grads) > 1 and dist.get_backend(self.pg) == "nccl":
            try:
                with cm(group=self.pg, device=grads[0].device, async_ops=False):
                    for g in grads:
                        dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.pg)
                return
            except (TypeError, RuntimeError, NotImplementedError):
                pass          # backend / version without coalescing: one collective per tensor
        for g in grads:
            dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.pg)

    def zero_grad(self):
        """Drop the gradients: the next backward installs its freshly written flat tensors without any
        accumulate kernel."""
        for p in self.params:
            p.grad = None

    def all_reduce(self, global_docs: Optional[int] = None):
        """Sum gradients over ranks in place; optionally divide by the global document count (the reference's
        total_loss / batch_size)."""
        if self._distributed():
            if not self.overlap:
                for p in self.params:               # same collective sequence on every rank, whatever its shard produced
                    if p.grad is None:
                        p.grad = torch.zeros_like(p)
                self._all_reduce_coalesced([p.grad for p in self.params])
            elif len(self._pending) != len(self.params):
                raise RuntimeError(f"FlatGradBucket(overlap=True): {len(self._pending)} of {len(self.params)} gradient "
                                   "hooks fired on this rank; every rank must produce every block's gradient each step")
            for w in self._pending:
                w.wait()
        self._pending.clear()
        if global_docs is not None:
            for p in self.params:
                if p.grad is not None:
                    p.grad.div_(float(global_docs))


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
