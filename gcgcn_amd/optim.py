"""The trainer's optimiser as ONE launch per step: ``FusedAdam`` is ``torch.optim.Adam`` (the reference's
``optim.Adam(filter(lambda p: p.requires_grad, model.parameters()), lr=...)``, config/Config.py:300) with its update run by
``gcgcn_adam_step`` (csrc/optim.hip) over every parameter tensor at once.  It IS a ``torch.optim.Optimizer``: ``param_groups``,
``state_dict()`` / ``load_state_dict()`` (keys ``step``, ``exp_avg``, ``exp_avg_sq`` like torch's Adam, so a checkpoint moves
between the two), ``zero_grad()``.  Parameters whose ``.grad`` is None are skipped and keep their own step count, as in torch
(the dead last hop of the model, ``linears_k.*``).  fp32 GPU parameters only; no weight decay / amsgrad (the reference uses neither).
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
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps))
        self._host = None          # pinned staging buffer of the launch table

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
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
            if self._host is None or self._host.shape[0] < len(rows):
                self._host = torch.empty(max(64, len(rows)), 7, dtype=torch.int64).pin_memory()
            self._host[:len(rows)].copy_(torch.from_numpy(tab))
            dtab = self._host[:len(rows)].to(dev, non_blocking=True)
            call("gcgcn_adam_step", len(rows), dtab.data_ptr(), blocks, float(b1), float(b2), float(group["eps"]), _stream())
            # the pinned buffer is rewritten by the next step(): wait for this copy (tiny) -- one event, no device sync
            ev = torch.cuda.Event()
            ev.record()
            self._pending = (ev, dtab, keep)
            ev.synchronize()
        return loss
