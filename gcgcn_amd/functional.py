"""torch.autograd bindings of the HIP blocks (C ABI in include/gcgcn.h).

PyTorch is used for device memory (caching allocator), the current HIP stream and autograd
bookkeeping only; all arithmetic happens in libgcgcn_hip.so.  Inputs must be fp32 tensors on a
GPU -- anything else raises (no CPU path).
"""
from __future__ import annotations

import ctypes
import os
import threading
import weakref
from typing import Optional

import torch

from . import _lib
from ._lib import SALT_GLUE, call

Tensor = torch.Tensor


def _p(t: Optional[Tensor]):
    return None if t is None else t.data_ptr()          # ctypes turns the int into the void* the ABI declares


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    """hipStream_t of torch's current stream.  The raw getter costs ~0.3 us; ``torch.cuda.current_stream()`` builds a
    Stream object (~15 us), which at ten C calls per step was a fifth of the host time of a step."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def _chk(t: Tensor, name: str, ndim: Optional[int] = None) -> Tensor:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a tensor, got {type(t).__name__}")
    if not t.is_cuda:
        raise RuntimeError(f"{name}: gcgcn_amd runs on MI355X only; got a {t.device} tensor (no CPU fallback)")
    if t.dtype != torch.float32:
        raise TypeError(f"{name}: expected float32, got {t.dtype}")
    if ndim is not None and t.dim() != ndim:
        raise ValueError(f"{name}: expected {ndim} dims, got shape {tuple(t.shape)}")
    return t.contiguous()


def _nv(n_valid: Optional[Tensor], B: int, N: int, dev) -> Optional[Tensor]:
    if n_valid is None:
        return None
    nv = n_valid.to(device=dev, dtype=torch.int32).contiguous()
    if nv.shape != (B,):
        raise ValueError(f"n_valid: expected shape ({B},), got {tuple(nv.shape)}")
    return nv


# ---- ragged batches: the entity rows that exist ------------------------------------------------------------------------
# With n_valid the node-phase products of a block (X WnX, Ebar We, the output projection, every data and weight gradient) run
# on the LIVE 16-row blocks of the [B N]-row tensors only (block r of document b is live iff 16 r < n_valid[b]); the tensors
# stay padded in memory, so nothing else changes.  The list is built on the device from n_valid (one tiny launch, no host
# read, captured with the step) once per hop loop and handed to every block call.  N must be a multiple of 16; the blocks fall
# back to the dense products by themselves where the column-strip chain kernels do not serve the shape.
row_block_launches = os.environ.get("GCGCN_ROW_BLOCKS", "1") != "0"      # GCGCN_ROW_BLOCKS=0: A/B knob (every padded row computed)


def row_blocks(n_valid: Optional[Tensor], B: int, N: int) -> Optional[Tensor]:
    """int32[gcgcn_row_blocks_ints(B, N)] for gcgcn_gcn_fwd / _bwd / gcgcn_mha_fwd / _bwd (``rowblk``), or None (dense products)."""
    # (N = 16 is one block per document: nothing to skip but whole empty documents, and the list's lookups cost cfg 1 10 %)
    if n_valid is None or not row_block_launches or N % 16 != 0 or N < 32 or not n_valid.is_cuda:
        return None
    nv = n_valid.to(dtype=torch.int32).contiguous()
    out = torch.empty(int(_lib.lib().gcgcn_row_blocks_ints(B, N)), dtype=torch.int32, device=nv.device)
    call("gcgcn_row_blocks", B, N, _p(nv), _p(out), _stream())
    return out


# ---- dropout RNG state -----------------------------------------------------------------------------
# One generator per DEVICE, like torch's own default CUDA generator behind nn.Dropout (which the reference uses): a
# device-resident {seed, counter} pair advanced by the kernels themselves.  This is the one piece of process-wide state of the
# product path; creation / re-seeding is guarded by a lock, the counter advances on the device in stream order.
_rng_state = {}
_rng_lock = threading.Lock()


def manual_seed(seed: int, device=None):
    """Seed the dropout generator of this library (per device): state = {seed, counter = 0}."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    st = torch.tensor([seed, 0], dtype=torch.int64, device=dev)
    with _rng_lock:
        _rng_state[dev.index or 0] = st


_tls = threading.local()   # .scope: [snaps tensor [count, 2], next index, pending] while a rng_scope is active on this
                           # thread; .handoff: the edge means parked on this thread (no process-global mutable state)


def _get_scope():
    return getattr(_tls, "scope", None)


def _state(dev: torch.device) -> Tensor:
    idx = dev.index or 0
    st = _rng_state.get(idx)
    if st is None:
        st = torch.tensor([torch.initial_seed() & 0x7FFFFFFFFFFFFFFF, 0], dtype=torch.int64, device=dev)
        with _rng_lock:
            st = _rng_state.setdefault(idx, st)     # two threads racing here agree on one state
    return st


def _new_snaps(dev: torch.device, count: int) -> Tensor:
    snaps = torch.empty(count, 2, dtype=torch.int64, device=dev)
    call("gcgcn_rng_next", _p(_state(dev)), _p(snaps), count, _stream())
    return snaps


class rng_scope:
    """Draw the {seed, counter} snapshots of up to `count` dropout-using forwards at once (GraphHops uses it for a whole
    hop loop); outside a scope every forward launches its own tiny kernel.  The draw itself is LAZY: the scope only
    allocates the snapshot buffer; GATAttention's forward -- the first kernel of a hop loop -- fills it inside its own
    first launch (gcgcn_gat_fwd's rng_state argument), and any other consumer that comes first launches gcgcn_rng_next."""

    def __init__(self, dev: torch.device, count: int, enabled: bool = True):
        self.dev, self.count, self.enabled = dev, count, enabled

    def __enter__(self):
        self.prev = _get_scope()
        if self.enabled:
            _tls.scope = [torch.empty(self.count, 2, dtype=torch.int64, device=self.dev), 0, True]
        return self

    def __exit__(self, *exc):
        _tls.scope = self.prev
        return False


def _take_pending_rng(dev: torch.device):
    """(state, snaps, count) of the active scope if its snapshots have not been drawn yet -- the caller promises to draw
    them in its next launch, before anything reads a snapshot -- else None."""
    sc = _get_scope()
    if sc is not None and sc[2] and sc[0].device == dev:
        sc[2] = False
        return _state(dev), sc[0], sc[0].shape[0]
    return None


def rng_snapshot(dev: torch.device, lazy: bool = False) -> Tensor:
    """Device-side {seed, counter} for one dropout-using forward; the counter advances on the GPU, so a
    captured hipGraph draws a fresh mask at every replay.  lazy=True: the caller will take the pending draw itself
    (_take_pending_rng) in the launch that reads this snapshot."""
    sc = _get_scope()
    if sc is not None and sc[1] < sc[0].shape[0] and sc[0].device == dev:
        if sc[2] and not lazy:                          # somebody other than GATAttention comes first: draw now
            sc[2] = False
            call("gcgcn_rng_next", _p(_state(dev)), _p(sc[0]), sc[0].shape[0], _stream())
        snap = sc[0][sc[1]]
        sc[1] += 1
        return snap
    return _new_snaps(dev, 1)[0]


def dropout_keep_mask(snap: Tensor, salt: int, p: float, numel: int) -> Tensor:
    """Boolean keep-mask of a dropout site (test aid: lets the CPU oracle replay the same mask)."""
    keep = torch.empty(numel, dtype=torch.uint8, device=snap.device)
    call("gcgcn_dropout_keep", _p(keep), numel, _p(snap), salt, float(p), _stream())
    return keep.bool()


# ---- GATAttention + single pass over E -------------------------------------------------------------
class GatFn(torch.autograd.Function):
    """(X[B,N,D], E[B,N,N,D], flat) -> (A[B,N,N], Ebar[B,N,D], X).  GCGCN_glove.py:154-168 (+ :40-41).
    The third output is X itself: a hop that hands THIS alias to its convolution lets the convolution's dX arrive
    here, where the attention's own dX kernel adds it -- instead of autograd summing the two with one more launch."""

    @staticmethod
    def forward(ctx, x, e, flat, n_valid, p, snap, pending, Dh, mask, uvc, uvc_valid):
        B, N, D = x.shape
        dev = x.device
        st, sn, cnt = pending if pending is not None else (None, None, 0)
        if uvc is None:
            uvc, uvc_valid = torch.empty(2 * D + 1, device=dev), False
        s = torch.empty(B, N, device=dev)
        P = torch.empty(B, N, N, device=dev)
        A = torch.empty(B, N, N, device=dev) if snap is not None else None
        ebar = torch.empty(B, N, D, device=dev)
        call("gcgcn_gat_fwd", B, N, D, Dh, _p(x), _p(e), _p(n_valid), _p(flat), _p(snap), float(p), _p(uvc), _p(s),
             _p(P), _p(A), _p(ebar), _p(st), _p(sn), cnt, _p(mask), 1 if uvc_valid else 0, _stream())
        ctx.save_for_backward(x, e, flat, uvc, P)
        ctx.n_valid, ctx.p, ctx.snap, ctx.Dh = n_valid, float(p), snap, Dh
        return (P if A is None else A), ebar, x.view_as(x)

    @staticmethod
    def backward(ctx, dA, dEbar, dXin):
        x, e, flat, uvc, P = ctx.saved_tensors
        B, N, D = x.shape
        dev = x.device
        dA = torch.zeros(B, N, N, device=dev) if dA is None else dA.contiguous()
        dEbar = None if dEbar is None else dEbar.contiguous()
        dXin = None if dXin is None else dXin.contiguous()
        dX = torch.empty_like(x)
        dE = torch.empty_like(e) if ctx.needs_input_grad[1] else None
        dflat = torch.empty_like(flat)
        dlogit = torch.empty(B, N, N, device=dev)
        ds = torch.empty(B, N, device=dev)
        dvpart = torch.empty(B * N, D, device=dev)
        duvc = torch.empty(2 * D + 1, device=dev)
        nscr = _lib.lib().gcgcn_gat_bwd_scratch(B, N, D)
        scratch = torch.empty(max(nscr, 1), device=dev)
        bp = _current_pass()                     # weight gradients parked earlier in this backward pass ride in the edge pass
        call("gcgcn_gat_bwd", B, N, D, ctx.Dh, _p(x), _p(e), _p(ctx.n_valid), _p(flat), _p(ctx.snap), ctx.p, _p(uvc), _p(P),
             _p(dA), _p(dEbar), _p(dXin), _p(dX), _p(dE), _p(dflat), _p(dlogit), _p(ds), _p(dvpart), _p(duvc),
             _p(scratch), None if bp is None else bp.queue, _stream())
        return dX, dE, dflat, None, None, None, None, None, None, None, None


class EdgeMeanFn(torch.autograd.Function):
    """E[B,N,N,D] -> Ebar[B,N,D] = mean_j E  (GraphConv edge term, GCGCN_glove.py:40-41 commuted)."""

    @staticmethod
    def forward(ctx, e, n_valid):
        B, N, _, D = e.shape
        ebar = torch.empty(B, N, D, device=e.device)
        call("gcgcn_edge_mean_fwd", B, N, D, _p(e), _p(n_valid), _p(ebar), _stream())
        ctx.n_valid, ctx.shape = n_valid, (B, N, D)
        return ebar

    @staticmethod
    def backward(ctx, dEbar):
        B, N, D = ctx.shape
        dEbar = dEbar.contiguous()
        dE = torch.empty(B, N, N, D, device=dEbar.device)
        call("gcgcn_edge_mean_bwd", B, N, D, _p(dEbar), _p(ctx.n_valid), _p(dE), _stream())
        return dE, None


class MhaFn(torch.autograd.Function):
    """(X[B,N,D], flat) -> (A[B,H,N,N], X).  MultiHeadAttention.forward, GCGCN_glove.py:133-142.  The second output
    is the alias of X described at GatFn."""

    @staticmethod
    def forward(ctx, x, flat, n_valid, H, p, snap, rowblk=None):
        B, N, D = x.shape
        dev = x.device
        Q = torch.empty(B, N, D, device=dev)
        P = torch.empty(B, H, N, N, device=dev)
        A = torch.empty(B, H, N, N, device=dev) if snap is not None else None
        scratch = torch.empty(max(_lib.lib().gcgcn_mha_scratch(B, N, D), 1), device=dev)
        call("gcgcn_mha_fwd", B, N, D, H, _p(x), _p(n_valid), _p(flat), _p(snap), float(p), _p(Q), _p(P), _p(A),
             _p(scratch), _p(rowblk), _stream())
        ctx.save_for_backward(x, flat, Q, P)
        ctx.H, ctx.p, ctx.snap, ctx.rowblk = H, float(p), snap, rowblk
        ctx.flat_leaf = flat if (flat.is_leaf and not _under_ddp()) else None
        return (P if A is None else A), x.view_as(x)

    @staticmethod
    def backward(ctx, dA, dXin):
        x, flat, Q, P = ctx.saved_tensors
        B, N, D = x.shape
        H, dev = ctx.H, x.device
        dA = torch.zeros(B, H, N, N, device=dev) if dA is None else dA.contiguous()
        dXin = None if dXin is None else dXin.contiguous()
        dX = torch.empty_like(x)
        dflat = torch.empty_like(flat)
        dS = torch.empty(B, H, N, N, device=dev)
        dQ = torch.empty(B, N, D, device=dev)
        scratch = torch.empty(max(_lib.lib().gcgcn_mha_scratch(B, N, D), 1), device=dev)
        # dWq could be parked too (defer_mha_weight_grads); measured neutral at cfg 2 -- the carrying edge pass is already the
        # longer side with the two convolutions' products (0.652 vs 0.648 ms) -- so it stays with its own group launch
        bp = _pass_for_parking(ctx, 1) if defer_mha_weight_grads else None
        call("gcgcn_mha_bwd", B, N, D, H, _p(x), _p(flat), _p(ctx.snap), ctx.p, _p(Q), _p(P), _p(dA), _p(dXin), _p(dX),
             _p(dflat), _p(dS), _p(dQ), _p(scratch), None if bp is None else bp.queue, 0, _p(ctx.rowblk), _stream())
        if bp is not None:
            bp.park(ctx.flat_leaf, dflat, (x, dQ, ctx.rowblk))
            dflat = None                                # installed as .grad by the pass's end-of-backward callback
        return dX, dflat, None, None, None, None, None


# ---- deferred weight gradients ---------------------------------------------------------------------------------
# A block's weight-gradient products (7 GFLOP for MAGGC at cfg 2) are needed by nobody before the end of backward, and the
# last big kernel of backward -- GATAttention's pass over E -- is HBM-bound with idle matrix pipes.  gcgcn_gcn_bwd parks
# them in the queue of the running backward pass (include/gcgcn.h); gcgcn_gat_bwd carries them as extra workgroups of its
# edge pass; the pass's end-of-backward callback launches whatever is still parked and hands the gradients over.
#
# Soundness.  A parked gradient is NOT returned to autograd (the function returns None for it): autograd would add or
# clone the tensor -- e.g. when one module is called twice inside one backward pass, the reference trainer's own pattern,
# config/Config.py:340-372 -- before the parked products have written it.  Instead the callback, which runs after every
# node of the pass, installs each finished buffer as ``.grad`` (or adds it to a ``.grad`` that exists by then: gradient
# accumulation, other consumers of the parameter).  Parking is refused -- the products then run inside the block's own
# backward and the gradient takes autograd's normal route -- unless this is a ``.backward()`` pass that will reach the
# parameter's AccumulateGrad node (not ``autograd.grad``, not ``inputs=[...]`` without it) and the parameter has no hooks.
# Node-level hooks on AccumulateGrad (torch DistributedDataParallel) cannot be seen from Python; a forward that runs inside
# DDP's own forward is recognised instead (_under_ddp) and its blocks do not park.  (gcgcn_amd.dist.FlatGradBucket needs
# nothing: it reduces after backward, or uses tensor hooks, which are seen.)
#
# State.  One _BackwardPass per autograd graph task, registered under the task's id and owned by the engine through the
# queued callback: a pass that raises half-way is destroyed with its graph task -- parked operands, queue and all --
# and leaves nothing behind; passes of other models, devices or threads never share a queue.
defer_weight_grads = os.environ.get("GCGCN_DEFER", "1") != "0"      # GCGCN_DEFER=0: A/B knob
defer_mha_weight_grads = False
defer_fused_mha_weight_grads = os.environ.get("GCGCN_DEFER_MHA", "1") != "0"   # the fused MAGGC hop's dWq + dbq second stage (A/B knob)
_warned_ddp = [False]


def _under_ddp() -> bool:
    """True while torch's DistributedDataParallel is running the forward this block is part of.  DDP reduces gradients from
    C++ hooks on the parameters' AccumulateGrad nodes, which fire when autograd delivers a gradient -- a parked gradient is
    installed after backward and would never be reduced.  Blocks whose forward ran under DDP therefore do not park."""
    ddp = getattr(torch.nn.parallel, "DistributedDataParallel", None)
    if ddp is not None and not hasattr(ddp, "_active_ddp_module"):
        # a torch build without the (private) marker this test relies on: fail CLOSED -- with several ranks a DDP wrapper may be
        # active and cannot be recognised, so nothing is parked (FlatGradBucket users: pass through, it reduces after backward)
        import torch.distributed as dist
        on = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    else:
        on = ddp is not None and getattr(ddp, "_active_ddp_module", None) is not None
    if on and defer_weight_grads and not _warned_ddp[0]:
        _warned_ddp[0] = True
        import warnings
        warnings.warn("gcgcn_amd: running under torch DistributedDataParallel -- deferred weight gradients are switched off "
                      "for these blocks (their AccumulateGrad hooks need the gradient during backward); "
                      "gcgcn_amd.dist.FlatGradBucket reduces the flat gradients without that cost")
    return on
_passes = {}            # graph task id -> weakref to its _BackwardPass
_passes_lock = threading.Lock()


class _BackwardPass:
    def __init__(self, task_id: int):
        self.task_id = task_id
        self.queue = _lib.lib().gcgcn_defer_create()
        if not self.queue:
            raise MemoryError("gcgcn_defer_create failed")
        self.keep = []          # operands of parked products
        self.installs = []      # (leaf parameter, its finished flat gradient)

    def park(self, leaf, dflat, operands):
        self.keep.append(operands)
        self.installs.append((leaf, dflat))

    def __call__(self):         # end of the backward pass (autograd final callback; runs on the caller's stream)
        with _passes_lock:
            _passes.pop(self.task_id, None)
        try:
            if _lib.lib().gcgcn_defer_count(self.queue) > 0:
                call("gcgcn_defer_flush", self.queue, _stream())
            with torch.no_grad():
                for leaf, dflat in self.installs:
                    if leaf.grad is None:
                        leaf.grad = dflat
                    else:
                        leaf.grad.add_(dflat)
        finally:
            self.keep.clear()
            self.installs.clear()

    def __del__(self):          # also the only thing that happens to the pass of a backward that raised
        q, self.queue = self.queue, None
        if q:
            try:
                _lib.lib().gcgcn_defer_destroy(q)
            except Exception:   # interpreter shutdown
                pass


def _current_pass() -> Optional[_BackwardPass]:
    """The pass object of the graph task this thread is executing, if a block has parked something in it."""
    tid = torch._C._current_graph_task_id()
    if tid < 0:
        return None
    ref = _passes.get(tid)
    return None if ref is None else ref()


def _pass_for_parking(ctx, idx: int, leaf=None) -> Optional[_BackwardPass]:
    """The running backward pass if input ``idx`` of this function (a flat parameter: ``leaf``, default ``ctx.flat_leaf``) may be
    parked, else None."""
    if not (defer_weight_grads and ctx.needs_input_grad[idx]):
        return None
    leaf = ctx.flat_leaf if leaf is None else leaf
    if leaf is None or leaf._post_accumulate_grad_hooks or leaf._backward_hooks:
        return None
    tid = torch._C._current_graph_task_id()
    if tid < 0:
        return None
    try:
        acc = ctx.next_functions[idx][0]         # the parameter's AccumulateGrad node
        if acc is None or getattr(acc, "variable", None) is not leaf or torch._C._will_engine_execute_node(acc) is not True:
            return None
    except RuntimeError:                          # autograd.grad(): the gradient is captured, not accumulated
        return None
    with _passes_lock:
        ref = _passes.get(tid)
        bp = None if ref is None else ref()
        if bp is None:
            # (a dead entry under this id belongs to an earlier process-lifetime task that failed: ids are not reused
            # while a task lives)
            bp = _BackwardPass(tid)
            _passes[tid] = weakref.ref(bp)
            torch.autograd.Variable._execution_engine.queue_callback(bp)     # the engine owns the pass from here on
            for k in [k for k, r in _passes.items() if r() is None]:
                del _passes[k]
    return bp


def _ride(inp, n_valid, out, B, N, D):
    """ctypes gcgcn_edge_ride for (inp -> out); the caller keeps the struct alive across the call."""
    r = _lib.EdgeRide(B, N, D, inp.data_ptr(), None if n_valid is None else n_valid.data_ptr(), out.data_ptr())
    return r, ctypes.cast(ctypes.pointer(r), ctypes.c_void_p)


# The sum over heads of the output projection (gcgcn_gcn_fwd's wsum) rides in the forward call and is kept for backward;
# False: backward sums it itself in a small launch (what a caller of the C ABI that passes NULL gets).  A/B and test switch.
head_sum_in_forward = True


class GcnFn(torch.autograd.Function):
    """(X[B,N,D], Ebar[B,N,D], A[B,H,N,N], flat[, E_next[B,N,N,D]]) -> out[B,N,D][, mean_j E_next].
    GraphConvolution.forward (H = 1) / MultiGraphConvolution.forward, GCGCN_glove.py:63-80 / 97-120.  The optional
    E_next is the NEXT hop's edge tensor: its mean (all that hop needs of it, glove:40-41) rides inside this block's
    latency-bound chain launch, and so does its backward."""

    @staticmethod
    def forward(ctx, x, ebar, adj, flat, n_valid, L, H, p, snap, e_next, out_p, out_snap, rowblk=None):
        B, N, D = x.shape
        dev = x.device
        HD = H * D
        out = torch.empty(B, N, D, device=dev)
        Pn = torch.empty(B, N, HD, device=dev)
        Y = torch.empty(B, N, HD, device=dev)
        HO = torch.empty(B, N, HD, device=dev)
        rinv = torch.empty(B, H, N, device=dev)
        G = torch.empty(B, N, HD, device=dev)
        # sum over heads of the output projection's column blocks: a by-product of the first launch, used by backward
        wsum = torch.empty(D, D, device=dev) if (H > 1 and head_sum_in_forward and any(ctx.needs_input_grad[:4])) else None
        scratch = torch.empty(max(_lib.lib().gcgcn_gcn_scratch(B, N, D, H), 1), device=dev)
        ebar_next, ride, ride_p = None, None, None
        if e_next is not None:
            ebar_next = torch.empty(e_next.shape[0], e_next.shape[1], e_next.shape[3], device=dev)
            ride, ride_p = _ride(e_next, n_valid, ebar_next, *ebar_next.shape)
        call("gcgcn_gcn_fwd", B, N, D, L, H, _p(x), _p(ebar), _p(adj), _p(n_valid), _p(flat), _p(snap), float(p),
             _p(out_snap), float(out_p), _p(out), _p(Pn), _p(Y), _p(HO), _p(rinv), _p(G), _p(wsum), _p(scratch), ride_p,
             None, _p(rowblk), _stream())
        del ride
        ctx.save_for_backward(x, ebar, adj, flat, Pn, Y, HO, rinv)
        ctx.wsum, ctx.rowblk = wsum, rowblk
        ctx.n_valid, ctx.L, ctx.H, ctx.p, ctx.snap = n_valid, L, H, float(p), snap
        ctx.out_p, ctx.out_snap = float(out_p), out_snap
        # to see in backward whether .grad will be installed or added to (never parked under DDP: see _under_ddp)
        ctx.flat_leaf = flat if (flat.is_leaf and not _under_ddp()) else None
        ctx.next_shape = None if e_next is None else tuple(e_next.shape)
        return out, ebar_next

    @staticmethod
    def backward(ctx, dout, debar_next):
        x, ebar, adj, flat, Pn, Y, HO, rinv = ctx.saved_tensors
        B, N, D = x.shape
        L, H, dev = ctx.L, ctx.H, x.device
        HD = H * D
        dout = dout.contiguous()
        dX = torch.empty_like(x)
        dEbar = torch.empty_like(ebar)
        dA = torch.empty_like(adj)
        dflat = torch.empty_like(flat)
        W1 = torch.empty(B, N, HD, device=dev)
        W2 = torch.empty(B, N, HD, device=dev)
        W3 = torch.empty(B, N, HD, device=dev)
        drow = torch.empty(B, H, N, device=dev)
        dXres = torch.empty(B, N, D, device=dev)
        dout_m = torch.empty(B, N, D, device=dev) if (ctx.n_valid is not None or ctx.out_snap is not None) else None
        scratch = torch.empty(max(_lib.lib().gcgcn_gcn_scratch(B, N, D, H), 1), device=dev)
        dE_next, ride, ride_p = None, None, None
        if ctx.next_shape is not None and ctx.needs_input_grad[9] and debar_next is not None:
            debar_next = debar_next.contiguous()
            dE_next = torch.empty(ctx.next_shape, device=dev)
            ride, ride_p = _ride(debar_next, ctx.n_valid, dE_next, *debar_next.shape)
        bp = _pass_for_parking(ctx, 3)
        call("gcgcn_gcn_bwd", B, N, D, L, H, _p(x), _p(ebar), _p(adj), _p(ctx.n_valid), _p(flat), _p(ctx.snap),
             ctx.p, _p(ctx.out_snap), ctx.out_p, _p(Pn), _p(Y), _p(HO), _p(rinv), _p(ctx.wsum), _p(dout), _p(dX), _p(dEbar), _p(dA),
             _p(dflat), _p(W1), _p(W2), _p(W3), _p(drow), _p(dXres), _p(dout_m), _p(scratch), ride_p, None,
             None if bp is None else bp.queue, _p(ctx.rowblk), _stream())
        del ride
        if bp is not None:
            bp.park(ctx.flat_leaf, dflat, (x, ebar, Y, HO, dout, dout_m, W2, W3, ctx.rowblk))
            dflat = None                                # installed as .grad by the pass's end-of-backward callback
        return dX, dEbar, dA, dflat, None, None, None, None, None, dE_next, None, None, None


class MaggcFn(torch.autograd.Function):
    """One MAGGC hop as ONE autograd node: MultiHeadAttention.forward (glove:133-142) + MultiGraphConvolution.forward
    (glove:97-120) on its adjacencies, (X[B,N,D], Ebar[B,N,D], flat_mha, flat_gcn[, E_next]) -> out[B,N,D][, mean_j E_next].
    The attention's launches ride inside the convolution's (gcgcn_mha_hook, include/gcgcn.h): the query projection is one more
    problem of the convolution's first group launch, the attention core's backward runs as passenger workgroups of its last
    one -- two launches fewer per step than MhaFn + GcnFn, the same arithmetic (same kernels' bodies, same dropout draws)."""

    @staticmethod
    def forward(ctx, x, ebar, flat_mha, flat, n_valid, L, H, p_mha, snap_mha, p, snap, e_next, out_p, out_snap, rowblk=None):
        B, N, D = x.shape
        dev = x.device
        HD = H * D
        Q = torch.empty(B, N, D, device=dev)
        P = torch.empty(B, H, N, N, device=dev)
        A = torch.empty(B, H, N, N, device=dev) if snap_mha is not None else None
        out = torch.empty(B, N, D, device=dev)
        Pn, Y, HO, G = (torch.empty(B, N, HD, device=dev) for _ in range(4))
        rinv = torch.empty(B, H, N, device=dev)
        wsum = torch.empty(D, D, device=dev) if (H > 1 and head_sum_in_forward and any(ctx.needs_input_grad[:4])) else None
        scratch = torch.empty(max(_lib.lib().gcgcn_gcn_scratch(B, N, D, H), 1), device=dev)
        ebar_next, ride, ride_p = None, None, None
        if e_next is not None:
            ebar_next = torch.empty(e_next.shape[0], e_next.shape[1], e_next.shape[3], device=dev)
            ride, ride_p = _ride(e_next, n_valid, ebar_next, *ebar_next.shape)
        hook = _lib.MhaHook(flat_mha.data_ptr(), Q.data_ptr(), P.data_ptr(), _p(A), None, _p(snap_mha), float(p_mha))
        call("gcgcn_gcn_fwd", B, N, D, L, H, _p(x), _p(ebar), None, _p(n_valid), _p(flat), _p(snap), float(p), _p(out_snap), float(out_p),
             _p(out), _p(Pn), _p(Y), _p(HO), _p(rinv), _p(G), _p(wsum), _p(scratch), ride_p,
             ctypes.cast(ctypes.pointer(hook), ctypes.c_void_p), _p(rowblk), _stream())
        del ride, hook
        adj = P if A is None else A
        ctx.save_for_backward(x, ebar, adj, flat, Pn, Y, HO, rinv, flat_mha, Q, P)
        ctx.wsum, ctx.rowblk = wsum, rowblk
        ctx.n_valid, ctx.L, ctx.H, ctx.p, ctx.snap = n_valid, L, H, float(p), snap
        ctx.p_mha, ctx.snap_mha = float(p_mha), snap_mha
        ctx.out_p, ctx.out_snap = float(out_p), out_snap
        ctx.flat_leaf = flat if (flat.is_leaf and not _under_ddp()) else None
        ctx.flat_mha_leaf = flat_mha if (flat_mha.is_leaf and not _under_ddp()) else None
        ctx.next_shape = None if e_next is None else tuple(e_next.shape)
        return out, ebar_next

    @staticmethod
    def backward(ctx, dout, debar_next):
        x, ebar, adj, flat, Pn, Y, HO, rinv, flat_mha, Q, P = ctx.saved_tensors
        B, N, D = x.shape
        L, H, dev = ctx.L, ctx.H, x.device
        HD = H * D
        dout = dout.contiguous()
        dXc, dEbar = torch.empty_like(x), torch.empty_like(ebar)        # dXc: the convolution's share of dX
        dA = torch.empty_like(adj)
        dflat = torch.empty_like(flat)
        W1, W2, W3 = (torch.empty(B, N, HD, device=dev) for _ in range(3))
        drow = torch.empty(B, H, N, device=dev)
        dXres = torch.empty(B, N, D, device=dev)
        dout_m = torch.empty(B, N, D, device=dev) if (ctx.n_valid is not None or ctx.out_snap is not None) else None
        scratch = torch.empty(max(_lib.lib().gcgcn_gcn_scratch(B, N, D, H), 1), device=dev)
        dQ = torch.empty(B, N, D, device=dev)
        dE_next, ride, ride_p = None, None, None
        if ctx.next_shape is not None and ctx.needs_input_grad[11] and debar_next is not None:
            debar_next = debar_next.contiguous()
            dE_next = torch.empty(ctx.next_shape, device=dev)
            ride, ride_p = _ride(debar_next, ctx.n_valid, dE_next, *debar_next.shape)
        bp = _pass_for_parking(ctx, 3)
        hook = _lib.MhaHook(None, Q.data_ptr(), P.data_ptr(), None, dQ.data_ptr(), _p(ctx.snap_mha), ctx.p_mha)
        call("gcgcn_gcn_bwd", B, N, D, L, H, _p(x), _p(ebar), _p(adj), _p(ctx.n_valid), _p(flat), _p(ctx.snap), ctx.p, _p(ctx.out_snap),
             ctx.out_p, _p(Pn), _p(Y), _p(HO), _p(rinv), _p(ctx.wsum), _p(dout), _p(dXc), _p(dEbar), _p(dA), _p(dflat), _p(W1), _p(W2),
             _p(W3), _p(drow), _p(dXres), _p(dout_m), _p(scratch), ride_p, ctypes.cast(ctypes.pointer(hook), ctypes.c_void_p),
             None if bp is None else bp.queue, _p(ctx.rowblk), _stream())
        del ride, hook
        if bp is not None:
            bp.park(ctx.flat_leaf, dflat, (x, ebar, Y, HO, dout, dout_m, W2, W3, ctx.rowblk))
            dflat = None
        # the rest of the attention's backward: dX = dQ Wq + dXc, dWq = dQ^T X, dbq (the core already ran, as passengers).
        # dWq and the second stage of dbq's column sums are needed by nobody before the end of backward: parked like the
        # convolution's weight gradients (round 5: the launch then has nothing to reduce -- one launch less, a shorter group)
        dX = torch.empty_like(x)
        dflat_mha = torch.empty_like(flat_mha)
        dS = torch.empty(1, device=dev)
        scratch2 = torch.empty(max(_lib.lib().gcgcn_mha_scratch(B, N, D), 1), device=dev)
        bq = _pass_for_parking(ctx, 2, ctx.flat_mha_leaf) if (defer_fused_mha_weight_grads and ctx.flat_mha_leaf is not None) else None
        call("gcgcn_mha_bwd", B, N, D, H, _p(x), _p(flat_mha), _p(ctx.snap_mha), ctx.p_mha, _p(Q), _p(P), _p(dA), _p(dXc), _p(dX),
             _p(dflat_mha), _p(dS), _p(dQ), _p(scratch2), None if bq is None else bq.queue, 1, _p(ctx.rowblk), _stream())
        if bq is not None:
            bq.park(ctx.flat_mha_leaf, dflat_mha, (x, dQ, scratch2, ctx.rowblk))
            dflat_mha = None
        return dX, dEbar, dflat_mha, dflat, None, None, None, None, None, None, None, dE_next, None, None, None


class GraphConvFn(torch.autograd.Function):
    """(X[B,N,Din], Ebar[B,N,De], A[B,N,N], We, Wn, bias|None) -> out[B,N,Dout].  GraphConv.forward, glove:36-50."""

    @staticmethod
    def forward(ctx, x, ebar, adj, we, wn, bias):
        B, N, Din = x.shape
        De, Dout = we.shape
        dev = x.device
        out = torch.empty(B, N, Dout, device=dev)
        T = torch.empty(B, N, Dout, device=dev)
        rinv = torch.empty(B, N, device=dev)
        scratch = torch.empty(max(_lib.lib().gcgcn_gcn_scratch(B, N, max(Din, Dout), 1), 1), device=dev)
        call("gcgcn_graphconv_fwd", B, N, Din, De, Dout, _p(x), _p(ebar), _p(adj), _p(we), _p(wn), _p(bias), _p(out),
             _p(T), _p(rinv), _p(scratch), _stream())
        ctx.save_for_backward(x, ebar, adj, we, wn, out, T, rinv)
        ctx.has_bias = bias is not None
        return out

    @staticmethod
    def backward(ctx, dout):
        x, ebar, adj, we, wn, out, T, rinv = ctx.saved_tensors
        B, N, Din = x.shape
        De, Dout = we.shape
        dev = x.device
        dout = dout.contiguous()
        dX, dEbar, dA = torch.empty_like(x), torch.empty_like(ebar), torch.empty_like(adj)
        dWe, dWn = torch.empty_like(we), torch.empty_like(wn)
        dbias = torch.empty(Dout, device=dev) if ctx.has_bias else None
        dS, dT = torch.empty(B, N, Dout, device=dev), torch.empty(B, N, Dout, device=dev)
        drow = torch.empty(B, N, device=dev)
        scratch = torch.empty(max(_lib.lib().gcgcn_gcn_scratch(B, N, max(Din, Dout), 1), 1), device=dev)
        call("gcgcn_graphconv_bwd", B, N, Din, De, Dout, _p(x), _p(ebar), _p(adj), _p(we), _p(wn), _p(out), _p(T),
             _p(rinv), _p(dout), _p(dX), _p(dEbar), _p(dA), _p(dWe), _p(dWn), _p(dbias), _p(dS), _p(dT), _p(drow),
             _p(scratch), _stream())
        return dX, dEbar, dA, dWe, dWn, dbias


class DropoutFn(torch.autograd.Function):
    """Elementwise dropout of the hop glue (GCGCN_glove.py:341); backward replays the same mask."""

    @staticmethod
    def forward(ctx, x, p, snap, salt):
        y = torch.empty_like(x)
        call("gcgcn_dropout", _p(x), _p(y), x.numel(), _p(snap), salt, float(p), _stream())
        ctx.p, ctx.snap, ctx.salt = float(p), snap, salt
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        dx = torch.empty_like(dy)
        call("gcgcn_dropout", _p(dy), _p(dx), dy.numel(), _p(ctx.snap), ctx.salt, ctx.p, _stream())
        return dx, None, None, None


class PairBceFn(torch.autograd.Function):
    """(logits[B,N,N,R], labels[B,N,N,R]) -> loss[B].  The trainer's per-document loss, config/Config.py:355-366."""

    @staticmethod
    def forward(ctx, logits, labels, n_valid):
        B, N, _, R = logits.shape
        loss = torch.empty(B, device=logits.device)
        part = torch.empty(B * N, device=logits.device)
        call("gcgcn_pair_bce_fwd", B, N, R, _p(logits), _p(labels), _p(n_valid), _p(loss), _p(part), _stream())
        ctx.save_for_backward(logits, labels)
        ctx.n_valid = n_valid
        return loss

    @staticmethod
    def backward(ctx, dloss):
        logits, labels = ctx.saved_tensors
        B, N, _, R = logits.shape
        dlogits = torch.empty_like(logits)
        call("gcgcn_pair_bce_bwd", B, N, R, _p(logits), _p(labels), _p(ctx.n_valid), _p(dloss.contiguous()), _p(dlogits),
             _stream())
        return dlogits, None, None


class ProducerFn(torch.autograd.Function):
    """(ctx[B,T,Hd], node[B,N,Hd], dis_table[ND,P], flat; sen, pos_h, pos_t) -> E[B,N,N,Hd].  The edge-feature producer of
    one hop: WordAttention x2 + linear_word_att + SentenceAttention x2 + linear_sentence_att, GCGCN_glove.py:171-214 as
    called at :300-330.  ``compact``: E is never written; the outputs are the compact rows Ec[rows, Hd] (one per entity pair with
    a live sentence slot) and the pair -> row index, for the consumers in csrc/compact.hip (CompactEdges)."""

    @staticmethod
    def forward(ctx_, tok, node, dis_table, flat, sen, pos_h, pos_t, n_valid, cap_rows, cap_pairs, compact):
        B, T, Hd = tok.shape
        N, S = sen.shape[1], sen.shape[3]
        ND, P = dis_table.shape
        dev = tok.device
        if compact:
            cap_pairs = max(int(cap_pairs), 1)
        sizes = (ctypes.c_int64 * 7)()
        call("gcgcn_producer_sizes", B, N, S, T, Hd, P, ND, cap_rows, cap_pairs, ctypes.cast(sizes, ctypes.c_void_p))
        ibuf = torch.empty(sizes[0], dtype=torch.int32, device=dev)
        fbuf = torch.empty(sizes[1], device=dev)
        E = None if compact else torch.empty(B, N, N, Hd, device=dev)
        pb = pos_h.element_size()
        call("gcgcn_producer_fwd", B, N, S, T, Hd, P, ND, _p(tok), _p(sen), _p(pos_h), _p(pos_t), pb, _p(node), _p(dis_table),
             _p(n_valid), _p(flat), cap_rows, cap_pairs, _p(ibuf), _p(fbuf), None, 0, _p(E), _stream())
        ctx_.save_for_backward(tok, node, dis_table, flat, sen, pos_h, pos_t, ibuf, fbuf)
        ctx_.n_valid, ctx_.caps, ctx_.nbwd, ctx_.compact = n_valid, (cap_rows, cap_pairs), int(sizes[2]), bool(compact)
        counts = ibuf[int(sizes[3]):int(sizes[3]) + 4]      # {live rows, live pairs, over capacity, 0}, on the device
        if not compact:
            ctx_.mark_non_differentiable(counts)
            return E, counts
        rows = int(sizes[6])
        Ec = fbuf[int(sizes[5]):int(sizes[5]) + rows * Hd].view(rows, Hd)
        prow = ibuf[int(sizes[4]):int(sizes[4]) + B * N * N].view(B, N, N)
        ctx_.mark_non_differentiable(counts, prow)
        return Ec, counts, prow

    @staticmethod
    def backward(ctx_, dE, _dcounts=None, _dprow=None):
        tok, node, dis_table, flat, sen, pos_h, pos_t, ibuf, fbuf = ctx_.saved_tensors
        B, T, Hd = tok.shape
        N, S = sen.shape[1], sen.shape[3]
        ND, P = dis_table.shape
        dev = tok.device
        dE = dE.contiguous()                       # compact: the gradient of Ec (rows beyond the live pairs are zero)
        bbuf = torch.empty(ctx_.nbwd, device=dev)
        dtok, dnode = torch.empty_like(tok), torch.empty_like(node)
        dtab, dflat = torch.empty_like(dis_table), torch.zeros_like(flat)       # (zeros: the layout's alignment gaps)
        call("gcgcn_producer_bwd", B, N, S, T, Hd, P, ND, _p(tok), _p(sen), _p(pos_h), _p(pos_t), pos_h.element_size(), _p(node),
             _p(dis_table), _p(ctx_.n_valid), _p(flat), ctx_.caps[0], ctx_.caps[1], _p(ibuf), _p(fbuf), _p(bbuf),
             None if ctx_.compact else _p(dE), _p(dE) if ctx_.compact else None, _p(dtok), _p(dnode), _p(dtab), _p(dflat), _stream())
        return dtok, dnode, dtab, dflat, None, None, None, None, None, None, None


class CompactEdges:
    """A hop's edge tensor without the tensor: ``Ec[rows, Hd]`` (one row per entity pair with a live sentence slot), ``prow[B,N,N]``
    (row of a pair, -1 = none) and ``bias[Hd]`` -- every other real pair of ``context_sent_att`` equals the bias of
    ``linear_sentence_att`` (EdgeFeatureProducer).  GATAttention / GraphConvolution / MultiGraphConvolution (and GraphHops,
    GraphModelTail) accept it wherever they accept the dense ``edge_feat``; ``dense()`` expands it (tests, other consumers)."""

    def __init__(self, Ec: Tensor, prow: Tensor, bias: Tensor, n_valid: Optional[Tensor], batched: bool = True):
        self.Ec, self.prow, self.bias, self.n_valid, self.batched = Ec, prow, bias, n_valid, batched

    @property
    def _version(self):                       # (the edge-mean hand-off keys on it, like on a tensor's version counter)
        return self.Ec._version

    @property
    def shape(self):
        B, N, _ = self.prow.shape
        return (B, N, N, self.Ec.shape[1]) if self.batched else (N, N, self.Ec.shape[1])

    def dense(self) -> Tensor:
        B, N, _ = self.prow.shape
        live = (self.prow >= 0).unsqueeze(-1)
        e = torch.where(live, self.Ec[self.prow.clamp_min(0).long()], self.bias.expand(B, N, N, -1))
        if self.n_valid is not None:
            ok = torch.arange(N, device=e.device)[None, :] < self.n_valid[:, None]
            e = e * (ok[:, :, None] & ok[:, None, :]).unsqueeze(-1).to(e.dtype)
        return e if self.batched else e[0]


class GatCompactFn(torch.autograd.Function):
    """GatFn on compact rows: (X[B,N,D], Ec[rows,D], bias[D], flat) -> (A[B,N,N], Ebar[B,N,D], X)."""

    @staticmethod
    def forward(ctx, x, Ec, bias, flat, prow, n_valid, p, snap, pending, Dh, uvc, uvc_valid):
        B, N, D = x.shape
        dev = x.device
        st, sn, cnt = pending if pending is not None else (None, None, 0)
        if uvc is None:
            uvc, uvc_valid = torch.empty(2 * D + 1, device=dev), False
        s = torch.empty(B, N, device=dev)
        P = torch.empty(B, N, N, device=dev)
        A = torch.empty(B, N, N, device=dev) if snap is not None else None
        ebar = torch.empty(B, N, D, device=dev)
        call("gcgcn_gat_fwd_compact", B, N, D, Dh, _p(x), _p(Ec), _p(prow), _p(bias), _p(n_valid), _p(flat), _p(snap), float(p),
             _p(uvc), _p(s), _p(P), _p(A), _p(ebar), _p(st), _p(sn), cnt, 1 if uvc_valid else 0, _stream())
        ctx.save_for_backward(x, Ec, bias, flat, uvc, P, prow)
        ctx.n_valid, ctx.p, ctx.snap, ctx.Dh = n_valid, float(p), snap, Dh
        return (P if A is None else A), ebar, x.view_as(x)

    @staticmethod
    def backward(ctx, dA, dEbar, dXin):
        x, Ec, bias, flat, uvc, P, prow = ctx.saved_tensors
        B, N, D = x.shape
        dev = x.device
        dA = torch.zeros(B, N, N, device=dev) if dA is None else dA.contiguous()
        dEbar = None if dEbar is None else dEbar.contiguous()
        dXin = None if dXin is None else dXin.contiguous()
        dX = torch.empty_like(x)
        dEc = torch.zeros_like(Ec)                       # rows beyond the live pairs stay zero (the producer's GEMMs read them)
        dbias = torch.empty_like(bias)
        dflat = torch.empty_like(flat)
        dlogit = torch.empty(B, N, N, device=dev)
        ds = torch.empty(B, N, device=dev)
        dvpart = torch.empty(B * N, D, device=dev)
        duvc = torch.empty(2 * D + 1, device=dev)
        scratch = torch.empty(max(_lib.lib().gcgcn_gat_bwd_compact_scratch(B, N, D), 1), device=dev)
        call("gcgcn_gat_bwd_compact", B, N, D, ctx.Dh, _p(x), _p(Ec), _p(prow), _p(bias), _p(ctx.n_valid), _p(flat), _p(ctx.snap), ctx.p,
             _p(uvc), _p(P), _p(dA), _p(dEbar), _p(dXin), _p(dX), _p(dEc), _p(dbias), _p(dflat), _p(dlogit), _p(ds), _p(dvpart), _p(duvc),
             _p(scratch), _stream())
        return dX, dEc, dbias, dflat, None, None, None, None, None, None, None, None


class EdgeMeanCompactFn(torch.autograd.Function):
    """EdgeMeanFn on compact rows: (Ec[rows,D], bias[D]; prow[B,N,N]) -> Ebar[B,N,D]."""

    @staticmethod
    def forward(ctx, Ec, bias, prow, n_valid):
        B, N, _ = prow.shape
        D = Ec.shape[1]
        ebar = torch.empty(B, N, D, device=Ec.device)
        call("gcgcn_edge_mean_fwd_compact", B, N, D, _p(Ec), _p(prow), _p(bias), _p(n_valid), _p(ebar), _stream())
        ctx.save_for_backward(prow)
        ctx.n_valid, ctx.rows = n_valid, Ec.shape[0]
        return ebar

    @staticmethod
    def backward(ctx, dEbar):
        prow, = ctx.saved_tensors
        B, N, _ = prow.shape
        dEbar = dEbar.contiguous()
        D = dEbar.shape[-1]
        dEc = torch.zeros(ctx.rows, D, device=dEbar.device)
        dbias = torch.empty(D, device=dEbar.device)
        rowbuf = torch.empty(2 * B * N, device=dEbar.device)
        call("gcgcn_edge_mean_bwd_compact", B, N, D, _p(prow), _p(ctx.n_valid), _p(dEbar), _p(dEc), _p(dbias), _p(rowbuf), _stream())
        return dEc, dbias, None, None


class HeadFn(torch.autograd.Function):
    """(flat, ner_emb[7,Pt], dis_table[ND,Pr], feats_0 .. feats_{nf-1} [B,N,Hd]; node_type, node_relative_pos) ->
    logits[B,N,N,R].  The classifier head, GCGCN_glove.py:306-307, 344-358."""

    @staticmethod
    def forward(ctx, flat, ner_emb, dis_table, node_type, rel, n_valid, R, dis_plus, *feats):
        B, N, Hd = feats[0].shape
        nf, Pt, (ND, Pr) = len(feats), ner_emb.shape[1], dis_table.shape
        dev = flat.device
        sizes = (ctypes.c_int64 * 3)()
        call("gcgcn_head_sizes", B, N, R, ND, ctypes.cast(sizes, ctypes.c_void_p))
        fbuf = torch.empty(sizes[0], device=dev)
        # ragged batch: the pair passes run on the pairs that exist (compacted rows; the index lives in ibuf, on the device)
        ibuf = torch.empty(sizes[2], dtype=torch.int32, device=dev) if n_valid is not None else None
        logits = torch.empty(B, N, N, R, device=dev)
        fp = (ctypes.c_void_p * nf)(*[f.data_ptr() for f in feats])
        call("gcgcn_head_fwd", B, N, Hd, nf, Pt, Pr, R, ND, dis_plus, ctypes.cast(fp, ctypes.c_void_p), _p(node_type), _p(rel),
             _p(ner_emb), _p(dis_table), _p(n_valid), _p(flat), _p(fbuf), _p(ibuf), _p(logits), _stream())
        ctx.save_for_backward(flat, ner_emb, dis_table, node_type, rel, fbuf, *feats)
        ctx.n_valid, ctx.R, ctx.dis_plus, ctx.nbwd, ctx.ibuf = n_valid, R, dis_plus, int(sizes[1]), ibuf
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        flat, ner_emb, dis_table, node_type, rel, fbuf, *feats = ctx.saved_tensors
        B, N, Hd = feats[0].shape
        nf, Pt, (ND, Pr) = len(feats), ner_emb.shape[1], dis_table.shape
        dev = flat.device
        dlogits = dlogits.contiguous()
        bbuf = torch.empty(ctx.nbwd, device=dev)
        dfeats = [torch.empty_like(f) for f in feats]
        dner, dtab, dflat = torch.empty_like(ner_emb), torch.empty_like(dis_table), torch.zeros_like(flat)   # (gaps stay zero)
        fp = (ctypes.c_void_p * nf)(*[f.data_ptr() for f in feats])
        dp = (ctypes.c_void_p * nf)(*[f.data_ptr() for f in dfeats])
        call("gcgcn_head_bwd", B, N, Hd, nf, Pt, Pr, ctx.R, ND, ctx.dis_plus, ctypes.cast(fp, ctypes.c_void_p), _p(node_type),
             _p(rel), _p(ner_emb), _p(dis_table), _p(ctx.n_valid), _p(flat), _p(fbuf), _p(ctx.ibuf), _p(bbuf), _p(dlogits),
             ctypes.cast(dp, ctypes.c_void_p), _p(dner), _p(dtab), _p(dflat), _stream())
        return (dflat, dner, dtab, None, None, None, None, None, *dfeats)


# Range-check the integer ids handed to the head / the producer before the kernels index with them.  The reference's
# embedding lookups raise IndexError on an id outside the table; the kernels clamp instead (an out-of-range id must never
# become an out-of-bounds read), which would turn corrupt inputs into plausible logits.  The check is one fused reduction and
# ONE host read per call; it is skipped inside a hipGraph capture, and a training loop that has validated its data once can
# switch it off (``gcgcn_amd.functional.check_ids = False``).
check_ids = True
NER_ROWS = 7        # nn.Embedding(7, entity_type_size, padding_idx=0), GCGCN_glove.py:241 -- the head's kernels index 7 rows


def _check_id_range(checks):
    """checks: [(name, tensor, lo, hi)] -- raise IndexError naming the first tensor with an id outside [lo, hi]."""
    if not check_ids or torch.cuda.is_current_stream_capturing():
        return
    bad = torch.stack([((t < lo) | (t > hi)).any() for _, t, lo, hi in checks])
    if bool(bad.any().item()):
        flags = bad.tolist()
        name, t, lo, hi = next(c for c, f in zip(checks, flags) if f)
        raise IndexError(f"{name}: ids must lie in [{lo}, {hi}], got min {int(t.min())} / max {int(t.max())}")


def classifier_head(feats, node_type, node_relative_pos, ner_emb, dis_table, flat, relation_num, n_valid=None, dis_plus=10):
    """logits[B,N,N,R] from the model's node_feats list (each [B,N,Hd]), node_type int64[B,N], node_relative_pos
    int64[B,N,N], the two embedding tables and the head's flat parameters."""
    feats = [_chk(f, f"node_feats[{i}]", 3) for i, f in enumerate(feats)]
    B, N, Hd = feats[0].shape
    if any(tuple(f.shape) != (B, N, Hd) for f in feats):
        raise ValueError("node_feats: every entry must have the same [B,N,H] shape")
    for nm, t, shp in (("node_type", node_type, (B, N)), ("node_relative_pos", node_relative_pos, (B, N, N))):
        if not t.is_cuda or t.dtype != torch.int64 or tuple(t.shape) != shp:
            raise ValueError(f"{nm}: expected a GPU int64 tensor of shape {shp}, got {t.dtype} {tuple(t.shape)} on {t.device}")
    if ner_emb.dim() != 2 or ner_emb.shape[0] != NER_ROWS:
        raise ValueError(f"ner_emb.weight: the head indexes {NER_ROWS} rows (nn.Embedding(7, .), glove:241), got {tuple(ner_emb.shape)}")
    if dis_table.dim() != 2 or dis_table.shape[0] < 2 * int(dis_plus) + 1:
        raise ValueError(f"dis_embed.weight: needs at least 2 * dis_plus + 1 = {2 * int(dis_plus) + 1} rows, got {tuple(dis_table.shape)}")
    _check_id_range([("node_type", node_type, 0, NER_ROWS - 1),
                     ("node_relative_pos", node_relative_pos, -int(dis_plus), int(dis_plus))])
    return HeadFn.apply(_chk(flat, "flat"), _chk(ner_emb, "ner_emb.weight", 2), _chk(dis_table, "dis_embed.weight", 2),
                        node_type.contiguous(), node_relative_pos.contiguous(), _nv(n_valid, B, N, flat.device),
                        int(relation_num), int(dis_plus), *feats)


def producer_live_counts(sen: Tensor, n_valid=None):
    """(live sentence slots, entity pairs with a live slot) of a batch -- one small kernel and a device-to-host read.
    A slot is live iff token 0 belongs to it: the reference's own padding test (``~sen_matrix[..., 0:1]``, glove:305)."""
    B, N, _, S, T = sen.shape
    counts = torch.empty(2, dtype=torch.int32, device=sen.device)
    call("gcgcn_producer_count", B, N, S, T, _p(sen), _p(n_valid), _p(counts), _stream())
    r, q = counts.tolist()
    return r, q


# The position matrices are [B,N,N,S,T] (77 MB per DocRED document as int64): a range check reads them once more, which the
# producer's own kernels do not (they touch live slots only).  Off by default; the head's ids (a few KB) are always checked.
check_position_ids = False


class ProducerCapacityError(RuntimeError):
    """max_live_slots / max_live_pairs were smaller than what the batch holds."""


def edge_features(tok, sen, pos_h, pos_t, node, dis_table, flat, n_valid=None, max_live_slots=None, max_live_pairs=None,
                  check_capacity=False, return_counts=False, compact=False):
    """E[B,N,N,Hd] of one hop from token states tok[B,T,Hd], sentence masks sen[B,N,N,S,T] (bool / uint8), distance ids
    pos_h / pos_t [B,N,N,S,T] (int64 as the reference passes them, or int32 / uint8), entity features node[B,N,Hd] and
    the distance-embedding table dis_table[ND,P].  Without ``max_live_*`` the live slots are counted first (one host
    synchronisation per call, as cheap as the reference's own ``.cuda()`` copies); with them nothing synchronises and the
    call can be captured in a hipGraph.  Capacities that turn out too small are never silent: the kernels compute nothing
    and write NaN into every real pair of E (so a captured graph fails loudly downstream); ``check_capacity=True`` reads the
    device-side flag back (one synchronisation) and raises :class:`ProducerCapacityError`; ``return_counts=True`` also returns
    the device tensor ``int32[4] = {live slots, live pairs, over capacity, 0}`` for a check of the caller's own timing.
    ``compact=True``: returns ``(Ec[rows,Hd], prow[B,N,N])`` instead of E -- E is never written (see CompactEdges)."""
    tok, node, dis_table = _chk(tok, "context_output", 3), _chk(node, "node_feat", 3), _chk(dis_table, "dis_embed.weight", 2)
    B, T, Hd = tok.shape
    if sen.dim() != 5 or sen.shape[0] != B or sen.shape[4] != T or sen.shape[1] != sen.shape[2]:
        raise ValueError(f"sen_matrix: expected [B={B}, N, N, S, T={T}], got {tuple(sen.shape)}")
    N, S = sen.shape[1], sen.shape[3]
    if node.shape != (B, N, Hd):
        raise ValueError(f"node_feat: expected {(B, N, Hd)}, got {tuple(node.shape)}")
    if not sen.is_cuda:
        raise RuntimeError("sen_matrix: gcgcn_amd runs on MI355X only (no CPU fallback)")
    if sen.dtype == torch.bool:
        sen = sen.view(torch.uint8)
    elif sen.dtype != torch.uint8:
        sen = (sen != 0).to(torch.uint8)
    sen = sen.contiguous()
    for nm, p in (("pos_matrix_h", pos_h), ("pos_matrix_t", pos_t)):
        if tuple(p.shape) != tuple(sen.shape) or p.dtype not in (torch.int64, torch.int32, torch.uint8) or not p.is_cuda:
            raise ValueError(f"{nm}: expected a GPU int64 / int32 / uint8 tensor of shape {tuple(sen.shape)}")
    if pos_t.dtype != pos_h.dtype:
        raise ValueError("pos_matrix_h and pos_matrix_t must have the same dtype")
    if check_position_ids:
        _check_id_range([("pos_matrix_h", pos_h, 0, dis_table.shape[0] - 1), ("pos_matrix_t", pos_t, 0, dis_table.shape[0] - 1)])
    nv = _nv(n_valid, B, N, tok.device)
    given = max_live_slots is not None and max_live_pairs is not None
    if not given:
        r, q = producer_live_counts(sen, nv)
        max_live_slots = r if max_live_slots is None else max_live_slots
        max_live_pairs = q if max_live_pairs is None else max_live_pairs
    out = ProducerFn.apply(tok, node, dis_table, _chk(flat, "flat"), sen, pos_h.contiguous(), pos_t.contiguous(), nv,
                           int(max_live_slots), int(max_live_pairs), bool(compact))
    if compact:
        E, counts = (out[0], out[2]), out[1]          # (Ec, prow): the caller adds the bias and wraps them in CompactEdges
    else:
        E, counts = out
    if check_capacity and not torch.cuda.is_current_stream_capturing():
        if int(counts[2].item()) != 0:
            raise ProducerCapacityError(f"edge_features: max_live_slots={max_live_slots} / max_live_pairs={max_live_pairs} are "
                                        "smaller than the batch's live sentence slots / entity pairs (producer_live_counts "
                                        "reports both); nothing was computed")
    return (E, counts) if return_counts else E


# ---- functional entry points -------------------------------------------------------------------------
def _snap_for(training: bool, p: float, dev) -> Optional[Tensor]:
    return rng_snapshot(dev) if (training and p > 0.0) else None


def gat_attention(x, e, flat, n_valid=None, p=0.1, training=False, hidden_dim=None, mask=None, uvc=None, uvc_valid=False):
    """``mask`` (bool/uint8 ``[B,N,N]``, True = fill with -100000): the paper-faithful opt-in; None = the reference.
    ``uvc``: caller-owned buffer ``[2D+1]`` for the folded projection; ``uvc_valid``: it already holds the fold of THESE
    parameter values (kept from an earlier call) and the fold kernel is skipped."""
    x, e = _chk(x, "node_feat", 3), _chk(e, "edge_feat", 4)
    B, N, D = x.shape
    if e.shape != (B, N, N, D):
        raise ValueError(f"edge_feat: expected {(B, N, N, D)}, got {tuple(e.shape)}")
    nv = _nv(n_valid, B, N, x.device)
    snap = rng_snapshot(x.device, lazy=True) if (training and p > 0.0) else None
    if mask is not None:
        if not mask.is_cuda or tuple(mask.shape) != (B, N, N):
            raise ValueError(f"mask: expected a GPU tensor of shape {(B, N, N)}, got {tuple(mask.shape)} on {mask.device}")
        mask = (mask != 0).to(torch.uint8).contiguous()
    return GatFn.apply(x, e, _chk(flat, "flat"), nv, p, snap, _take_pending_rng(x.device),
                       D if hidden_dim is None else int(hidden_dim), mask, uvc, bool(uvc_valid))   # (A, Ebar, alias of x)


def gat_attention_compact(x, ce: CompactEdges, flat, n_valid=None, p=0.1, training=False, hidden_dim=None, uvc=None, uvc_valid=False):
    x = _chk(x, "node_feat", 3)
    B, N, D = x.shape
    if tuple(ce.prow.shape) != (B, N, N) or ce.Ec.shape[1] != D:
        raise ValueError(f"compact edge_feat: pairs {tuple(ce.prow.shape)} / width {ce.Ec.shape[1]} do not match node_feat {tuple(x.shape)}")
    nv = _nv(n_valid, B, N, x.device)
    snap = rng_snapshot(x.device, lazy=True) if (training and p > 0.0) else None
    return GatCompactFn.apply(x, _chk(ce.Ec, "Ec", 2), _chk(ce.bias, "bias", 1), _chk(flat, "flat"), ce.prow, nv, p, snap,
                              _take_pending_rng(x.device), D if hidden_dim is None else int(hidden_dim), uvc, bool(uvc_valid))


def edge_mean_compact(ce: CompactEdges, n_valid=None):
    B, N, _ = ce.prow.shape
    return EdgeMeanCompactFn.apply(_chk(ce.Ec, "Ec", 2), _chk(ce.bias, "bias", 1), ce.prow, _nv(n_valid, B, N, ce.Ec.device))


def edge_mean(e, n_valid=None):
    if isinstance(e, CompactEdges):
        return edge_mean_compact(e, n_valid)
    e = _chk(e, "edge_feat", 4)
    B, N = e.shape[0], e.shape[1]
    return EdgeMeanFn.apply(e, _nv(n_valid, B, N, e.device))


def multi_head_adjacency(x, flat, H, n_valid=None, p=0.1, training=False, rowblk=None):
    x = _chk(x, "node_feat", 3)
    B, N, D = x.shape
    return MhaFn.apply(x, _chk(flat, "flat"), _nv(n_valid, B, N, x.device), H, p, _snap_for(training, p, x.device), rowblk)


def gcn_stack(x, ebar, adj, flat, L, H, n_valid=None, p=0.2, training=False, e_next=None, out_dropout=0.0, rowblk=None):
    """Returns ``out`` -- or ``(out, mean_j e_next)`` when the next hop's edge tensor ``e_next[B,N,N,D']`` is given.
    ``out_dropout`` > 0 (training only): the hop's ``x <- dropout(out)`` (glove:341) applied inside the block."""
    x, ebar, adj = _chk(x, "node_feat", 3), _chk(ebar, "edge_mean", 3), _chk(adj, "adjacency", 4)
    B, N, D = x.shape
    if ebar.shape != (B, N, D) or adj.shape != (B, H, N, N):
        raise ValueError(f"gcn_stack: shapes x{tuple(x.shape)} ebar{tuple(ebar.shape)} adj{tuple(adj.shape)} H={H}")
    nv = _nv(n_valid, B, N, x.device)
    if e_next is not None:
        e_next = _chk(e_next, "next edge_feat", 4)
        if e_next.shape[:3] != (B, N, N):
            raise ValueError(f"gcn_stack: next edge tensor {tuple(e_next.shape)} does not match B={B} N={N}")
    snap = _snap_for(training, p, x.device)                       # draw order: block, then the hop's output dropout
    out_snap = _snap_for(training, out_dropout, x.device)
    if rowblk is None:                                            # a block called on its own: its own list
        rowblk = row_blocks(nv, B, N)
    out, ebar_next = GcnFn.apply(x, ebar, adj, _chk(flat, "flat"), nv, L, H, p, snap, e_next,
                                 out_dropout if out_snap is not None else 0.0, out_snap, rowblk)
    return out if e_next is None else (out, ebar_next)


def maggc_fusable(x: Tensor, H: int) -> bool:
    """Can MultiHeadAttention + MultiGraphConvolution of one hop run as one fused call pair on this input?  (graph of at most 64
    entities, narrow heads; gcgcn_maggc_fusable)"""
    return x.is_cuda and x.dim() == 3 and bool(_lib.lib().gcgcn_maggc_fusable(x.shape[1], x.shape[2], int(H)))


def maggc_hop(x, ebar, flat_mha, flat, L, H, n_valid=None, p_mha=0.1, p=0.2, training=False, e_next=None, out_dropout=0.0, rowblk=None):
    """MultiHeadAttention (its own draw first, like the separate modules) + MultiGraphConvolution of one hop, fused (MaggcFn)."""
    x, ebar = _chk(x, "node_feat", 3), _chk(ebar, "edge_mean", 3)
    B, N, D = x.shape
    if ebar.shape != (B, N, D):
        raise ValueError(f"maggc_hop: shapes x{tuple(x.shape)} ebar{tuple(ebar.shape)}")
    nv = _nv(n_valid, B, N, x.device)
    if e_next is not None:
        e_next = _chk(e_next, "next edge_feat", 4)
    snap_mha = _snap_for(training, p_mha, x.device)                 # draw order of the separate modules: attention, block, glue
    snap = _snap_for(training, p, x.device)
    out_snap = _snap_for(training, out_dropout, x.device)
    if rowblk is None:
        rowblk = row_blocks(nv, B, N)
    out, ebar_next = MaggcFn.apply(x, ebar, _chk(flat_mha, "flat"), _chk(flat, "flat"), nv, L, H, p_mha, snap_mha, p, snap, e_next,
                                   out_dropout if out_snap is not None else 0.0, out_snap, rowblk)
    return out if e_next is None else (out, ebar_next)


def graph_conv(x, ebar, adj, we, wn, bias=None):
    x, ebar, adj = _chk(x, "inputs", 3), _chk(ebar, "edge_mean", 3), _chk(adj, "adjacency_matrix", 3)
    return GraphConvFn.apply(x, ebar, adj, _chk(we, "weights_edge", 2), _chk(wn, "weights_node", 2),
                             None if bias is None else _chk(bias, "bias", 1))


def dropout(x, p=0.2, training=False, salt=SALT_GLUE):
    x = _chk(x, "x")
    if not training or p <= 0.0:
        return x
    return DropoutFn.apply(x, p, rng_snapshot(x.device), salt)


# ---- hand-off of the edge mean from GATAttention to the GraphConvolution that follows ---------------
# The reference calls gat(X, E, mask) and then graphcnn[0](X, E, A) with the SAME E (glove:332-333).
# GATAttention's single pass over E also yields mean_j E; it is parked here so the convolution does
# not stream E from HBM a second time.  Per thread, keyed by the edge tensor's identity (a few entries: interleaved
# models keep theirs), consumed on use, dropped when the tensor dies or changes (version counter).
_HANDOFF_SLOTS = 4


def _handoffs() -> dict:
    h = getattr(_tls, "handoff", None)
    if h is None:
        h = _tls.handoff = {}
    return h


def park_edge_mean(e: Tensor, n_valid, ebar: Tensor):
    h = _handoffs()
    for k in [k for k, v in h.items() if v[0]() is None]:
        del h[k]
    h.pop(id(e), None)
    while len(h) >= _HANDOFF_SLOTS:
        del h[next(iter(h))]                       # oldest first
    h[id(e)] = (weakref.ref(e), e._version, n_valid, ebar)


def take_edge_mean(e: Tensor, n_valid) -> Optional[Tensor]:
    ent = _handoffs().pop(id(e), None)
    if ent is None:
        return None
    ref, ver, nvp, ebar = ent
    if ref() is e and e._version == ver and nvp is n_valid:
        return ebar
    return None


def pair_bce_loss(logits, labels, n_valid=None):
    """Per-document loss of the reference trainer (config/Config.py:355-366): sigmoid, BCE averaged over the relation
    axis, summed over ordered entity pairs h != t and divided by n^2 - n.  ``logits``/``labels``: ``[N,N,R]`` (returns
    a scalar, like the trainer's ``temp_loss``) or ``[B,N,N,R]`` (returns ``[B]``; ``n_valid[B]`` for ragged batches).
    The reference accumulates ``total_loss`` over documents and divides by ``batch_size`` (:366-369): that is
    ``pair_bce_loss(...).sum() / batch_size``."""
    single = logits.dim() == 3
    lg = _chk(logits.unsqueeze(0) if single else logits, "logits", 4)
    lb = _chk((labels.unsqueeze(0) if single else labels).to(torch.float32), "labels", 4)
    if lb.shape != lg.shape or lg.shape[1] != lg.shape[2]:
        raise ValueError(f"pair_bce_loss: logits {tuple(lg.shape)} vs labels {tuple(lb.shape)}")
    out = PairBceFn.apply(lg, lb, _nv(n_valid, lg.shape[0], lg.shape[1], lg.device))
    return out[0] if single else out
