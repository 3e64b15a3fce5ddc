"""SURVEY 8(e) check on the real kernels: the gradients of a global batch sharded over two ranks and summed by
``FlatGradBucket.all_reduce()`` equal the gradients one process computes on the whole batch (dropout off).  Two processes
share the one GPU of the test box; the collective travels over gloo on the DEVICE tensors (RCCL needs one GPU per rank), so
shard_batch, the flat per-block gradients written by the HIP backward, the deferred-weight-gradient hand-over and the
bucket's collective code are all the ones the N-GPU bench uses.  Modes: one collective after backward (default, weight
gradients deferred) and per-block collectives from tensor hooks (overlap=True; parking must then be refused)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        import gcgcn_amd
        from gcgcn_amd import functional as F_
        from gcgcn_amd.dist import FlatGradBucket, shard_batch
        from oracle import gcgcn_oracle as O
        dev = torch.device("cuda:0")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        B, N, D, L, H = 4, 64, 128, 2, 4
        sd = O.init_stack_params(D, L, H, seed=7)
        x, e1, e2, _ = O.synth_docs(B, N, D, seed=8)
        cot = torch.randn(B, N, D, generator=torch.Generator().manual_seed(9))
        assert F_.defer_weight_grads                                            # GCGCN_DEFER=1, the default
        msgs = []
        for overlap in (False, True):
            hops = gcgcn_amd.GraphHops(D, L, H).to(dev).eval()
            hops.load_state_dict(sd, strict=True)
            plain = FlatGradBucket(hops)                                        # no hooks: for the single-process reference
            parked = []
            orig = F_._BackwardPass.park

            def spy(self, leaf, dflat, operands, _orig=orig, _parked=parked):
                _parked.append(leaf)
                return _orig(self, leaf, dflat, operands)
            F_._BackwardPass.park = spy

            def grads(bucket, xs, a, b, c):
                bucket.zero_grad()
                o = hops(xs.to(dev), [a.to(dev), b.to(dev)])[-1]
                torch.autograd.backward(o, c.to(dev))

            grads(plain, x, e1, e2, cot)                                        # one process, the global batch
            whole = [p.grad.clone() for p in plain.params]
            assert len(parked) == 3                                             # both convolutions and the fused hop's attention projection parked their products
            bucket = FlatGradBucket(hops, overlap=overlap)
            del parked[:]
            xs, a, b, c = shard_batch([x, e1, e2, cot], rank, world)            # this rank's documents
            grads(bucket, xs, a, b, c)
            if overlap:
                assert not parked, "hooked parameters must not be parked"       # _pass_for_parking refuses: hooks present
                assert len(bucket._pending) == len(bucket.params)
            else:
                assert len(parked) == 3
            bucket.all_reduce()                                                 # the collective itself, on device tensors
            F_._BackwardPass.park = orig
            err = max(((p.grad - w).abs().max() / w.abs().max().clamp_min(1e-12)).item()
                      for p, w in zip(bucket.params, whole))
            msgs.append("ok" if err < 1e-5 else f"overlap={overlap}: relative gradient error {err:.3e}")
        # a rank without a gradient for some block still joins the collective (zeros), instead of hanging the others
        hops = gcgcn_amd.GraphHops(D, L, H).to(dev).eval()
        bucket = FlatGradBucket(hops)
        bucket.zero_grad()
        if rank == 0:
            for p in bucket.params:
                p.grad = torch.ones_like(p)
        bucket.all_reduce()
        ok = all(torch.equal(p.grad, torch.ones_like(p)) for p in bucket.params)
        msgs.append("ok" if ok else "zero-fill collective gave a wrong sum")
        out.put((rank, "ok" if all(m == "ok" for m in msgs) else "; ".join(msgs)))
    except Exception as e:  # noqa: BLE001
        import traceback
        out.put((rank, repr(e) + traceback.format_exc()[-600:]))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_sharded_gradients_equal_single_process(gpu_device):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=280) for _ in procs]
    for p in procs:
        p.join(timeout=30)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res
