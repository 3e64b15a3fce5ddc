"""Batch-DP plumbing on CPU with the gloo backend, world_size = 2: the flat gradient arena is ONE
all-reduce operand, parameters/gradients are views of the arenas, documents shard along the batch axis.
(The kernels themselves need a GPU; here gradients are synthetic.)"""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import gcgcn_amd
from gcgcn_amd.dist import FlatGradBucket, shard_batch


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker_overlap(rank, world, port, out):
    """overlap=True: slices are all-reduced from post-accumulate hooks during backward."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(1337)
        hops = gcgcn_amd.GraphHops(16, 2, 4)
        bucket = FlatGradBucket(hops, overlap=True)
        bucket.zero_grad()
        # a stand-in backward: a loss that touches every bucketed parameter through autograd
        loss = sum((p * float((rank + 1) * (i + 1))).sum() for i, p in enumerate(bucket.params))
        loss.backward()
        assert len(bucket._pending) == len(bucket.params)        # one async slice per block, launched by the hooks
        bucket.all_reduce(global_docs=4)
        assert not bucket._pending
        for i, p in enumerate(bucket.params):
            want = (1 + 2) * (i + 1) / 4.0
            assert torch.allclose(p.grad, torch.full_like(p, want)), (i, p.grad.flatten()[:3])
        # second step: zero_grad drops the tensors, the hooks stay
        bucket.zero_grad()
        loss = sum((p * 2.0).sum() for p in bucket.params)
        loss.backward()
        bucket.all_reduce()
        for p in bucket.params:
            assert torch.allclose(p.grad, torch.full_like(p, 4.0))
        out.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        out.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def _worker(rank, world, port, out):
    """overlap=False: the per-block flat gradients are all-reduced after backward."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(1337)                                   # identical replicas
        hops = gcgcn_amd.GraphHops(16, 2, 4)
        bucket = FlatGradBucket(hops)
        assert bucket.numel == sum(p.numel() for n, p in hops.named_parameters() if not n.endswith("flat_k"))
        assert len(bucket.params) == 4                            # gat, caggc conv, mha, maggc conv
        bucket.zero_grad()
        assert all(p.grad is None for p in bucket.params)
        loss = sum((p * float((rank + 1) * (i + 1))).sum() for i, p in enumerate(bucket.params))
        loss.backward()
        bucket.all_reduce(global_docs=4)
        for i, p in enumerate(bucket.params):
            want = (1 + 2) * (i + 1) / 4.0                          # sum over ranks / global batch
            assert torch.allclose(p.grad, torch.full_like(p, want)), (i, p.grad.flatten()[:3])
        assert hops.get_adj_matrix[0].flat_k.grad is None         # linears_k stay out of the bucket
        # batch sharding
        x = torch.arange(8.0).view(8, 1)
        (xs,) = shard_batch([x], rank, world)
        assert xs.flatten().tolist() == [4.0 * rank + k for k in range(4)]
        out.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        out.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def _worker_c4(rank, world, port, out):
    """Config c4's partitioning (BASELINE.json configs[3]): 256 documents over 8 ranks = 32 per rank, one coalesced all-reduce of
    the four flat gradients, every rank ends with sum over ranks / global batch."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(1337)
        hops = gcgcn_amd.GraphHops(16, 2, 4)
        bucket = FlatGradBucket(hops)
        docs = torch.arange(256.0).view(256, 1)
        n_valid = torch.arange(256, dtype=torch.int32)
        xs, ns = shard_batch([docs, n_valid], rank, world)
        assert xs.shape[0] == 32 and xs.flatten().tolist() == [32.0 * rank + k for k in range(32)]
        assert ns.tolist() == list(range(32 * rank, 32 * rank + 32))
        bucket.zero_grad()
        # a rank's gradient = the sum over ITS documents; the reduced gradient must be the mean over all 256
        loss = sum((p * xs.sum() * float(i + 1)).sum() for i, p in enumerate(bucket.params))
        loss.backward()
        bucket.all_reduce(global_docs=256)
        for i, p in enumerate(bucket.params):
            want = float(i + 1) * (255.0 * 256.0 / 2.0) / 256.0
            assert torch.allclose(p.grad, torch.full_like(p, want)), (i, p.grad.flatten()[:3], want)
        out.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        out.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(240)
def test_c4_partitioning_gloo_world8():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_c4, args=(r, 8, port, q)) for r in range(8)]
    for p in procs:
        p.start()
    res = [q.get(timeout=200) for _ in procs]
    for p in procs:
        p.join(timeout=30)
    assert sorted(res) == [(r, "ok") for r in range(8)], res


@pytest.mark.timeout(120)
@pytest.mark.parametrize("target", [_worker, _worker_overlap], ids=["single_allreduce", "overlapped_slices"])
def test_flat_bucket_allreduce_gloo_world2(target):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=100) for _ in procs]
    for p in procs:
        p.join(timeout=30)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res


def test_shard_batch_rejects_uneven():
    with pytest.raises(ValueError):
        shard_batch([torch.zeros(5, 2)], 0, 2)
