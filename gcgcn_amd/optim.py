"""The trainer's optimiser as ONE launch per step: ``FusedAdam`` is ``torch.optim.Adam`` (the reference's
``optim.Adam(filter(lambda p: p.requires_grad, model.parameters()), lr=...)``, config/Config.py:300) with its update run by
``gcgcn_adam_step`` (csrc/optim.hip) over every parameter tensor at once.  It IS a ``torch.optim.Optimizer``: ``param_groups``,
``state_dict()`` / ``load_state_dict()`` (keys ``step``, ``exp_avg``, ``exp_avg_sq`` like torch's Adam, so a checkpoint moves
between the two), ``zero_grad()``.  Parameters whose ``.grad`` is None are skipped and keep their own step count, as in torch
(the dead last hop of the model, ``linears_k.*``).  fp32 GPU parameters only; no weight decay / amsgrad / maximize (the reference
uses none of them): ``param_groups`` carry those keys with torch's defaults so that a checkpoint moves in either direction, and a
group that asks for one of them (e.g. loaded from a torch Adam checkpoint trained with weight decay) raises instead of silently
training with different arithmetic.  The update is torch's formula with ``1 / sqrt(1 - beta2^t)`` as a multiplier (parameters agree
with torch.optim.Adam to 1e-6 relative over five steps, not bit for bit).

``step()`` never waits for the GPU: the launch table travels through a small ring of pinned staging buffers, each with its own
event recorded right behind its host-to-device copy; a buffer's event is only waited for when the ring comes round to it again
(three steps later, long done).
"""
from __future__ import annotations

import math

import numpy as np
import torch

from ._lib import call
from .functional import _stream


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8):
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1 and 0 <= betas[1] < 1):
            raise ValueError("FusedAdam: bad hyper-parameters")
        # weight_decay / amsgrad / maximize: torch.optim.Adam's keys at their defaults (checkpoint interchange); anything else raises
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=0, amsgrad=False, maximize=False))
        self._ring = []            # pinned staging buffers of the launch table: [tensor, event or None]
        self._ring_next = 0

    _RING = 3

    def _staging(self, rows: int):
        """The next pinned buffer of the ring with room for `rows` table rows; waits (host side only) for the copy that last read
        it -- issued _RING steps ago -- and never for anything else on the stream."""
        if len(self._ring) < self._RING:
            self._ring.append([torch.empty(max(64, rows), 7, dtype=torch.int64).pin_memory(), None])
            slot = self._ring[-1]
        else:
            slot = self._ring[self._ring_next % self._RING]
        self._ring_next += 1
        if slot[1] is not None and not torch.cuda.is_current_stream_capturing():
            slot[1].synchronize()
            slot[1] = None
        if slot[0].shape[0] < rows:
            slot[0] = torch.empty(rows, 7, dtype=torch.int64).pin_memory()
        return slot

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            if group.get("weight_decay", 0) != 0 or group.get("amsgrad", False) or group.get("maximize", False):
                raise RuntimeError("FusedAdam: weight_decay / amsgrad / maximize are not implemented (the reference trainer uses "
                                   "plain Adam, config/Config.py:300); this parameter group asks for one of them")
            b1, b2 = group["betas"]
            rows, blocks, keep = [], 0, []
            for p in group["params"]:
                g = p.grad
                if g is None:
                    continue
                if not (p.is_cuda and p.dtype == torch.float32 and g.dtype == torch.float32 and p.is_contiguous()):
                    raise RuntimeError("FusedAdam: fp32 contiguous GPU parameters only (no CPU fallback)")
                if g.is_sparse:
                    raise RuntimeError("FusedAdam: sparse gradients are not supported")
                g = g.contiguous()
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st["step"] = int(st["step"]) + 1
                t = st["step"]
                ss = group["lr"] / (1.0 - b1 ** t)
                ib = 1.0 / math.sqrt(1.0 - b2 ** t)
                n = p.numel()
                if n == 0:
                    continue
                rows.append((p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), n, blocks, ss, ib))
                blocks += (n + 1023) // 1024
                keep.append(g)
            if not rows:
                continue
            dev = group["params"][0].device
            tab = np.zeros((len(rows), 7), dtype=np.int64)
            for i, r in enumerate(rows):
                tab[i, :6] = r[:6]
                tab[i, 6] = np.array([r[6], r[7]], dtype=np.float32).view(np.int64)[0]
            slot = self._staging(len(rows))
            slot[0][:len(rows)].copy_(torch.from_numpy(tab))
            dtab = slot[0][:len(rows)].to(dev, non_blocking=True)
            if not torch.cuda.is_current_stream_capturing():
                slot[1] = torch.cuda.Event()
                slot[1].record()                 # right behind the copy: what a later reuse of this pinned buffer waits for
            call("gcgcn_adam_step", len(rows), dtab.data_ptr(), blocks, float(b1), float(b2), float(group["eps"]), _stream())
            # dtab and the gradients are released to the caching allocator in stream order: nothing to wait for here
        return loss
