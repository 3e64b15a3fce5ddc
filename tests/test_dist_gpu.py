"""SURVEY 8(e) check on the real kernels: the gradients of a global batch sharded over two ranks and summed by
FlatGradBucket equal the gradients one process computes on the whole batch (dropout off).  Two processes share the one
GPU of the test box; the collective travels over gloo (RCCL needs one GPU per rank), which exercises the same
FlatGradBucket / shard_batch code the N-GPU bench uses."""
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
        from gcgcn_amd.dist import FlatGradBucket, shard_batch
        from oracle import gcgcn_oracle as O
        dev = torch.device("cuda:0")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        B, N, D, L, H = 4, 64, 128, 2, 4
        sd = O.init_stack_params(D, L, H, seed=7)
        x, e1, e2, _ = O.synth_docs(B, N, D, seed=8)
        cot = torch.randn(B, N, D, generator=torch.Generator().manual_seed(9))
        hops = gcgcn_amd.GraphHops(D, L, H).to(dev).eval()
        hops.load_state_dict(sd, strict=True)
        bucket = FlatGradBucket(hops)

        def grads(xs, a, b, c):
            bucket.zero_grad()
            out = hops(xs.to(dev), [a.to(dev), b.to(dev)])[-1]
            torch.autograd.backward(out, c.to(dev))
            return [p.grad for p in bucket.params]

        whole = [g.clone().cpu() for g in grads(x, e1, e2, cot)]              # one process, the global batch
        xs, a, b, c = shard_batch([x, e1, e2, cot], rank, world)              # this rank's documents
        grads(xs, a, b, c)
        for p in bucket.params:                                               # gloo sums host copies of the flat buffers
            host = p.grad.cpu()
            dist.all_reduce(host)
            p.grad.copy_(host)
        err = max(((p.grad.cpu() - w).abs().max() / w.abs().max().clamp_min(1e-12)).item() for p, w in zip(bucket.params, whole))
        out.put((rank, "ok" if err < 1e-5 else f"relative gradient error {err:.3e}"))
    except Exception as e:  # noqa: BLE001
        out.put((rank, repr(e)))
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
