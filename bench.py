#!/usr/bin/env python3
"""bench.py -- docs/sec forward+backward through the CAGGC+MAGGC stack on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c2|c3|c5|c1] [--mode graph|eager] [--global-batch G]

One step = one forward + backward of GATAttention -> GraphConvolution -> MultiHeadAttention ->
MultiGraphConvolution (the hop loop of GCGCN_glove.py:329-341) over one batch of B synthetic
DocRED-shaped documents per GPU, train mode (all four dropout sites on), X / E1 / E2 and every
parameter requiring grad, loss = sum of the MAGGC output.  For N > 1 there is one rank per
GPU: either the driver launches them (torch.distributed.run sets RANK / WORLD_SIZE), or -- when
``--gpus N`` is given without that environment -- this script starts the N ranks itself as child
processes (before anything here touches a GPU) and relays rank 0's JSON line.  Documents are sharded
along the batch axis -- weak scaling by default (B per GPU fixed), strong scaling with ``--global-batch G``
(G / N documents per GPU; SURVEY 8d: G = 256) -- and the step ends with ONE RCCL all-reduce of the flat
gradient bucket.

Prints ONE JSON line on rank 0 (contract in the task statement), with two extra objects:
  roofline     -- the dominant kernel's algorithmic HBM bytes / its HIP-event-timed duration
  cpu_baseline -- the CPU oracle (a port of the reference's op sequence) timed on this host's cores
"""
import argparse
import ctypes
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {  # BASELINE.json configs; per-GPU batch
    "c1": dict(B=8, N=16, D=128, L=2, H=8),
    "c2": dict(B=32, N=64, D=256, L=2, H=8),
    "c3": dict(B=32, N=64, D=768, L=4, H=4),
    "c5": dict(B=32, N=256, D=512, L=2, H=8),
}
HBM_PEAK = 8.0e12        # B/s, MI355X_MICROARCH.md "HBM3E peak BW" (spec)
MFMA_F32_PEAK = 157.3e12  # flop/s, exact-f32 MFMA (spec)


def algorithmic_flops_per_doc(N, D, L, H):
    """F = 3 * F_fwd."""
    gh, dh = D // L, D // H
    S = sum(2 * N * (D + l * gh) * gh + 2 * N * N * gh + 2 * N * D * gh for l in range(L))
    f = (2 * N * N * D + 2 * N * D) + 2 * N * N * D + S + 2 * N * D * D
    f += H * (2 * N * D * dh + 2 * N * N * dh) + H * S + 2 * N * H * D * D
    return 3 * f


def synth(cfg, seed, dev):
    """Same generator as oracle.synth_docs, restated here so that the product path
    never imports the oracle: X~U(-1,1), E~N(0,0.25), adj~Bernoulli(0.3) zero-diag, E1 masked by adj."""
    B, N, D = cfg["B"], cfg["N"], cfg["D"]
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(B, N, D, generator=g) * 2 - 1
    adj = (torch.rand(B, N, N, generator=g) < 0.3).float() * (1 - torch.eye(N)).unsqueeze(0)
    e1 = torch.randn(B, N, N, D, generator=g) * 0.5 * adj.unsqueeze(-1)
    e2 = torch.randn(B, N, N, D, generator=g) * 0.5
    return [t.to(dev) for t in (x, e1, e2, adj)]


def cpu_baseline(cfg, budget_s=20.0):
    """The CPU oracle (reference op sequence, one document per call, PyTorch CPU kernels) on this host."""
    from oracle import gcgcn_oracle as O
    N, D, L, H = cfg["N"], cfg["D"], cfg["L"], cfg["H"]
    # the GPU box exposes every host core but a 1-GPU job owns a 16-core share; more threads than that
    # only add oversubscription on these small per-document tensors
    ncpu = os.cpu_count() or 1
    cores = min(ncpu, 16)
    torch.set_num_threads(cores)
    sd = {k: v.requires_grad_() for k, v in O.init_stack_params(D, L, H, seed=1337).items()}
    nd = 8
    x, e1, e2, adj = O.synth_docs(nd, N, D, seed=1337)
    g = torch.Generator().manual_seed(0)

    def one(b):
        xb, a, c = x[b].clone().requires_grad_(), e1[b].clone().requires_grad_(), e2[b].clone().requires_grad_()
        keeps = {"gat": torch.rand(N, N, generator=g) > 0.1, "cag": [torch.rand(N, D // L, generator=g) > 0.2] * L,
                 "glue.0": torch.rand(N, D, generator=g) > 0.2, "mha.1": [torch.rand(N, N, generator=g) > 0.1] * H,
                 "mag.1": [[torch.rand(N, D // L, generator=g) > 0.2] * L] * H, "glue.1": torch.rand(N, D, generator=g) > 0.2}
        out = O.hop_stack(xb, [a, c], adj[b], sd, L, H, keeps=keeps)[-1]
        for v in sd.values():
            v.grad = None
        out.sum().backward()

    t0 = time.perf_counter()
    one(0)
    first = time.perf_counter() - t0
    warm = 2 if first < budget_s / 8 else 0
    for b in range(warm):
        one(1 + b)
    times, t0 = [], time.perf_counter()
    while len(times) < 10 and (time.perf_counter() - t0) < budget_s:       # 10 timed documents, median
        t1 = time.perf_counter()
        one(len(times) % nd)
        times.append(time.perf_counter() - t1)
    if not times:
        times = [first]
    times.sort()
    n = len(times)
    med = times[n // 2] if n % 2 else 0.5 * (times[n // 2 - 1] + times[n // 2])
    return {"value": round(1.0 / med, 4), "unit": "docs/s", "cores": cores, "kind": "port",
            "sample": f"median of {n} documents (N={N}, D={D}, L={L}, H={H}) forward+backward, one per call, train mode, "
                      f"after {1 + warm} warm-up (BASELINE.md 3's protocol; mean over the same documents: {n / sum(times):.3f} docs/s); "
                      f"PyTorch CPU kernels with {cores} threads (the host reports {ncpu} CPUs; a "
                      "1-GPU lease of this pool owns a 16-core share, and more threads than that only oversubscribe these small "
                      "per-document tensors -- BASELINE.md 3 says os.cpu_count(), this is the deviation)"}


def self_launch(ngpus):
    """``python bench.py --gpus N`` without a torch.distributed environment: start the N ranks as children of this process
    (which has not touched a GPU: no HIP call, no torch.cuda.is_available()), wait, and exit with their status.  Rank 0 of the
    children inherits this process's stdout and prints the JSON line there."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ngpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print("bench.py: starting", ngpus, "ranks:", " ".join(cmd), file=sys.stderr)
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--mode", default=os.environ.get("GCGCN_BENCH_MODE", "graph"), choices=["eager", "graph"],
                    help="graph (default): the step is captured once in a hipGraph and replayed; the steps whose kernels are "
                         "bracketed by HIP events for the roofline (every 10th) run eagerly -- same kernels, same data. "
                         "eager: every step is issued from Python (the GPU time is the same; the host's jitter is not)")
    ap.add_argument("--prof-kernel", default="auto",
                    help="kernel family timed with HIP events inside the timed region (auto = the one with the largest time share)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--eval-mode", action="store_true", help="A/B only: dropout off (the metric is defined in train mode)")
    ap.add_argument("--input-sets", type=int, default=3,
                    help="resident synthetic batches rotated through the steps (default 3: 3 x (E1 + E2) = 805 MB at cfg 2, "
                         "more than the 256 MiB Infinity Cache, so no step finds its inputs cached by the step before -- as in "
                         "a training loop that feeds new documents every step).  1 = replay one batch (cache-warm)")
    ap.add_argument("--ragged", action="store_true",
                    help="secondary run (SURVEY 8d): DocRED-like entity counts n_valid ~ clip(round(N(19.5, 6^2)), 2, 42) padded to N")
    ap.add_argument("--global-batch", type=int, default=None,
                    help="strong scaling: this many documents per step over ALL GPUs (per GPU: global / N; SURVEY 8d uses 256). "
                         "Default: weak scaling, the config's B per GPU")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl (default) = RCCL over xGMI, one GPU per rank.  gloo: TEST ONLY -- the ranks may share one card "
                         "(rank r uses GPU r mod device_count) and the collectives travel over gloo on the device tensors; "
                         "exercises the whole N > 1 control flow of this script on a 1-GPU box")
    ap.add_argument("--overlap-grads", action="store_true",
                    help="all-reduce each block's gradient asynchronously from a backward hook (A/B; default: one "
                         "coalesced collective after backward)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus))

    # stdout carries exactly ONE line (the JSON).  This image's RCCL writes a version banner (and, at
    # NCCL_DEBUG=WARN, warnings) to fd 1 when the communicator is created, so fd 1 is pointed at stderr for
    # the whole run and the JSON goes to a saved copy of the real stdout.
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE=1")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; there is no CPU fallback for the product path")
    dev_index = local_rank
    if args.dist_backend == "gloo":                              # test mode: ranks may share a card
        dev_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)

    import torch.distributed as dist
    force_dist = os.environ.get("GCGCN_FORCE_DIST") == "1"      # 1-rank RCCL group: exercises the N > 1 code path
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    import gcgcn_amd
    from gcgcn_amd import _lib
    from gcgcn_amd.dist import FlatGradBucket

    cfg = dict(CONFIGS[args.config])
    if args.global_batch is not None:                 # strong scaling: a fixed global batch split over the ranks
        if args.global_batch % world != 0 or args.global_batch <= 0:
            raise SystemExit(f"--global-batch {args.global_batch} is not divisible by {world} GPUs")
        cfg["B"] = args.global_batch // world
    B, N, D, L, H = (cfg[k] for k in "BNDLH")
    torch.manual_seed(1337)                       # identical parameters on every rank
    hops = gcgcn_amd.GraphHops(D, L, H).to(dev).train()
    if args.eval_mode:
        hops.eval()
    gcgcn_amd.manual_seed(1337 + rank, dev)
    bucket = FlatGradBucket(hops, overlap=args.overlap_grads)   # default: one coalesced all-reduce after backward
    n_valid = None
    if args.ragged:
        g = torch.Generator().manual_seed(4242 + rank)
        n_valid = torch.clamp(torch.round(torch.randn(B, generator=g) * 6.0 + 19.5), 2, min(42, N)).to(torch.int32).to(dev)
        if os.environ.get("GCGCN_BENCH_NVALID"):      # experiment switch: every document with the same number of real entities
            n_valid = torch.full((B,), int(os.environ["GCGCN_BENCH_NVALID"]), dtype=torch.int32, device=dev)
    nsets = max(1, args.input_sets)
    sets = []                                     # resident batches, rotated: step i runs on sets[i % nsets]
    for k in range(nsets):
        x, e1, e2, adj = synth(cfg, 1337 + rank + 1000 * k, dev)
        if n_valid is not None:
            with torch.no_grad():                  # padding rows of X must be zero (include/gcgcn.h)
                x.mul_((torch.arange(N, device=dev)[None, :] < n_valid[:, None]).unsqueeze(-1).float())
        for t in (x, e1, e2):
            t.requires_grad_()
        sets.append((x, e1, e2, adj))
    cot = torch.ones(B, N, D, device=dev)         # d(sum(out))/d(out)

    def fwd_bwd(k):
        x, e1, e2, adj = sets[k]
        x.grad = e1.grad = e2.grad = None
        bucket.zero_grad()
        out = hops(x, [e1, e2], adj, n_valid=n_valid)[-1]
        torch.autograd.backward(out, cot)

    graphs, graph_grads = [], []
    if args.mode == "graph":                      # capture the launch-bound step in one hipGraph per resident batch
        try:
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                for k in range(nsets):
                    fwd_bwd(k)
            torch.cuda.current_stream().wait_stream(s)
            for k in range(nsets):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    fwd_bwd(k)
                graphs.append(g)
                graph_grads.append([(p, p.grad) for p in bucket.params])   # the tensors that graph's replays write the gradients into
        except Exception as ex:                   # never lose the measurement to a capture problem: issue the steps eagerly
            print(f"bench.py: hipGraph capture failed ({ex!r}); running --mode eager", file=sys.stderr)
            graphs, graph_grads = [], []
            args.mode = "eager"
            torch.cuda.synchronize()
    counter = [0]

    def step(eager=False, k=None, collective=True):
        """One forward + backward (+ the gradient all-reduce when there is more than one rank).  collective=False: the
        profiling / sampling steps outside the timed region -- they may run on a subset of the ranks, or a different number
        of times per rank, so they must not issue collectives (every rank has to issue the same collective sequence)."""
        if k is None:
            k = counter[0] % nsets
            counter[0] += 1
        bucket.active = collective                             # (overlap mode all-reduces from backward hooks)
        if graphs and not eager:
            graphs[k].replay()
            for p, g in graph_grads[k]:                        # .grad = what this replay wrote
                p.grad = g
        else:
            fwd_bwd(k)
        if collective and (world > 1 or force_dist):
            bucket.all_reduce()

    def sync():
        if world > 1 or force_dist:
            dist.barrier()
        torch.cuda.synchronize()

    FAMILIES = {  # timer tag (= one kernel) -> (bound, peak)
        "gemm_group": ("mfma", MFMA_F32_PEAK), "gemm_single": ("mfma", MFMA_F32_PEAK),
        "gcn_chain_fwd": ("mfma", MFMA_F32_PEAK), "gcn_chain_bwd": ("mfma", MFMA_F32_PEAK),
        "edge_bwd": ("hbm", HBM_PEAK), "edge_fwd_att": ("hbm", HBM_PEAK), "edge_fwd_mean": ("hbm", HBM_PEAK),
        "edge_bcast": ("hbm", HBM_PEAK),
    }
    SMALL = ("gemm_splitk_reduce", "softmax", "mha_core", "head_sum", "dropout", "gat_fold", "gat_dlogit", "node_score",
             "colsum", "mask_rows", "rowsum", "relu_norm")

    KERNEL_RE = {"gemm_group": r"gc::gemm_group(_pass)?_kernel", "gemm_single": r"gc::gemm_kernel<", "gcn_chain_fwd": r"gc::gcn_chain\w*_fwd_kernel",
                 "gcn_chain_bwd": r"gc::gcn_chain\w*_bwd_kernel", "edge_bwd": r"gc::edge_bwd(_carry)?_kernel",
                 "edge_fwd_att": r"gc::edge_fwd_kernel<\d+, true", "edge_fwd_mean": r"gc::edge_fwd_kernel<\d+, false",
                 "edge_bcast": r"gc::edge_bcast"}
    KERNEL_NAMES = {"gemm_group": "gc::gemm_group_kernel / gc::gemm_group_pass_kernel", "gemm_single": "gc::gemm_kernel<...>",
                    "gcn_chain_fwd": "gc::gcn_chain_fwd_kernel", "gcn_chain_bwd": "gc::gcn_chain_bwd_kernel",
                    "edge_bwd": "gc::edge_bwd_carry_kernel / gc::edge_bwd_kernel", "edge_fwd_att": "gc::edge_fwd_kernel<4,true,*>"}

    def profile(prefix, nsteps):
        """HIP-event time of every launch whose kernel name starts with `prefix` over nsteps steps."""
        _lib.call("gcgcn_prof_start", prefix.encode(), nsteps * 64 + 64)
        for _ in range(nsteps):
            step(eager=True, collective=False)                 # events cannot be recorded inside a graph replay; no collective
        torch.cuda.synchronize()
        ms, n, w = ctypes.c_double(0), ctypes.c_int(0), ctypes.c_double(0)
        _lib.call("gcgcn_prof_stop", ctypes.byref(ms), ctypes.byref(n), ctypes.byref(w))
        return ms.value, n.value, w.value

    for _ in range(2):                                         # first touches (allocator, code objects) before the pre-pass
        step()
    sync()
    use_prof = True
    # which kernel family dominates the step?  (untimed pre-pass, 3 steps per family)
    shares = {}
    dominant = args.prof_kernel
    if use_prof:
        for fam in list(FAMILIES) + list(SMALL):
            ms, n, _ = profile(fam, 3)
            shares[fam] = {"ms_per_step": round(ms / 3, 4), "launches_per_step": round(n / 3, 2)}
        if dominant == "auto":
            dominant = max(FAMILIES, key=lambda f: shares[f]["ms_per_step"])
        sync()

    # ---- the timed region: exactly K steps, nothing else ---------------------------------------------------------------
    # graph mode: K replays of the captured hipGraphs (rotating inputs); eager mode: K steps issued from Python.  No HIP
    # events, no profiling hooks inside the region: the dominant kernel's launches are timed on extra steps AFTER it (same
    # kernels, same rotating data) -- two events per launch stall the queue, and in graph mode such steps have to be issued
    # eagerly, which would mix two kinds of step into `value`.
    import gc
    gc.collect()
    gc.disable()                                               # no collector pauses inside the timed region
    # The W warm-up steps come last, directly in front of the timed region: everything host-side that takes milliseconds
    # (the pre-pass above, event creation, the collector) is done, so the GPU is idle only for the mandated barrier +
    # synchronize between warm-up and t0 -- after a longer idle gap the first ~20 steps run 3-5 % slower (clock ramp:
    # 0.585 -> 0.555 ms over 20 steps, GCGCN_BENCH_TRACE=1).
    for _ in range(args.warmup):
        step()
    sync()
    counter[0] = 0
    t0 = time.perf_counter()
    _evs = []
    for i in range(args.steps):
        step()
        if os.environ.get("GCGCN_BENCH_TRACE"):                # diagnosis only: per-step GPU and host-issue times on stderr
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            _evs.append((e, time.perf_counter() - t0))
    sync()
    dt = time.perf_counter() - t0
    if _evs:
        print("per-step GPU ms:", [round(_evs[i][0].elapsed_time(_evs[i + 1][0]), 3) for i in range(len(_evs) - 1)], file=sys.stderr)
        print("host issue ms:", [round(t * 1e3, 2) for _, t in _evs], "total", round(dt * 1e3, 2), file=sys.stderr)
    gc.enable()
    # the dominant kernel family, HIP events on its launches (hipExtLaunchKernelGGL start/stop events on the launch stream),
    # on `nsamp` eagerly issued steps right after the region
    nsamp = max(1, min(5, args.steps))
    kms, kn, kw = profile(dominant, nsamp) if use_prof else (0.0, 0, 0.0)

    tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1 or force_dist:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = tmax.item()

    # ragged batches: the host-side work counters of the launches assume full documents (they cannot see n_valid, which
    # lives on the device); scale them to the rows / pairs that exist -- edge streams touch n_b^2 of N^2 pairs per document,
    # the node-phase products n_b of N rows
    rag_hbm = rag_mfma = 1.0
    if n_valid is not None:
        nvf = n_valid.double()
        rag_hbm = float((nvf * nvf).sum().item()) / (B * N * N)
        rag_mfma = float(nvf.sum().item()) / (B * N)

    def roof(fam, ms, n, work, samp):
        bound, peak = FAMILIES[fam]
        if n == 0 or ms <= 0:
            return None
        if n_valid is not None:
            work = work * (rag_hbm if bound == "hbm" else rag_mfma)
        ach = work / (ms * 1e-3)                           # work/s over the time those launches were running
        r = {"bound": bound, "kernel": KERNEL_NAMES.get(fam, fam + "*"), "achieved": round(ach / (1e9 if bound == "hbm" else 1e12), 2),
             "peak": peak / (1e9 if bound == "hbm" else 1e12), "unit": "GB/s" if bound == "hbm" else "TFLOP/s",
             "frac": round(ach / peak, 4), "traffic": None, "avg_launch_us": round(ms / n * 1e3, 2),
             "launches": n, "work_per_launch": work / n, "sampled_steps": samp,
             "work": "executed fp32 flops (2MNK)" if bound == "mfma" else "algorithmic HBM bytes"}
        if n_valid is not None:
            r["work"] += (f" of the real entities only (x {rag_hbm:.4f} = sum n_b^2 / (B N^2))" if bound == "hbm" else
                          f" on real rows only (x {rag_mfma:.4f} = sum n_b / (B N))")
        import glob
        import re
        # PMC summaries of this config (offline: separate rocprofv3 --pmc passes, tools/profile_config.sh), latest round
        pmcs = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_{args.config}_pmc.json")) or
                      (glob.glob(os.path.join(ROOT, "profiles", "r*_c2_pmc_hbm.json")) if args.config == "c2" else []))
        if pmcs and not args.ragged:
            data = json.load(open(pmcs[-1]))["kernels"]
            want = KERNEL_RE.get(fam)
            hit = [v for k, v in data.items() if want and re.search(want, k)]
            if hit:
                r["pmc_source"] = os.path.relpath(pmcs[-1], ROOT)
                tot = sum(v.get("launches", 1) for v in hit)
                avg = lambda key: (sum(v[key] * v.get("launches", 1) for v in hit if key in v) / tot) if any(key in v for v in hit) else None
                r["traffic"] = avg("hbm_bytes_per_launch_corrected")     # HBM bytes per launch (FETCH_SIZE x 2 + WRITE_SIZE)
                r["traffic_unit"] = "HBM bytes per launch, rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes"
                if avg("mfma_busy") is not None:
                    r["mfma_busy"] = round(avg("mfma_busy"), 4)
        return r

    roofline = roof(dominant, kms, kn, kw, nsamp)
    if roofline is not None:
        roofline["sampled_on"] = (f"{nsamp} eagerly issued steps directly after the {args.steps} timed steps (same kernels, same "
                                  "rotating inputs; the timed region itself is " + ("hipGraph replays only)" if graphs else "eager steps without events)"))
    # the cache-warm figure (one batch replayed back to back, E1 partly served from the Infinity Cache) for comparison
    warm = None
    if nsets > 1 and rank == 0 and world == 1 and not force_dist:
        for _ in range(3):
            step(k=0, collective=False)
        torch.cuda.synchronize()
        tw = time.perf_counter()
        for _ in range(args.steps):
            step(k=0, collective=False)
        torch.cuda.synchronize()
        warm = (time.perf_counter() - tw) / args.steps
    roofline_hbm = None
    if use_prof and rank == 0 and dominant != "edge_bwd":
        roofline_hbm = roof("edge_bwd", *profile("edge_bwd", 5), 5)
        if roofline_hbm:
            roofline_hbm["sampled_on"] = "5 eager steps after the timed region (rotating inputs)"
        from gcgcn_amd import functional as F_
        if roofline_hbm and F_.defer_weight_grads and N % 64 == 0 and D % 64 == 0 and (D // L) % 64 == 0:
            # The launch also carries the convolutions' parked weight-gradient products: their flops run on
            # the matrix pipes under the HBM stream, so the launch is judged against both roofs.  (With a short CAGGC chain,
            # cfg 1/2, all of them ride here; a long one, cfg 3, takes a share itself.)
            M, gh = B * N, D // L
            carried = sum(3 * 2.0 * D * (h * D) * M + sum(2.0 * (l * gh) * gh * M * h for l in range(1, L)) for h in (1, H))
            us = roofline_hbm["avg_launch_us"]
            roofline_hbm["carried_mfma"] = {"flops_per_launch": carried, "TFLOP/s": round(carried / us / 1e6, 2),
                                            "frac_of_f32_mfma_peak": round(carried / (us * 1e-6) / MFMA_F32_PEAK, 4),
                                            "note": "upper bound when a long CAGGC chain launch takes some of them; the PMC "
                                                    "traffic of this launch includes the carried products' operand reads "
                                                    "(edge stream alone: 1.02 x algorithmic, profiles/r01_c2_eager_kernel_stats_v5 era)"}

    if rank == 0:
        docs = B * world * args.steps
        value = docs / dt
        flops = algorithmic_flops_per_doc(N, D, L, H)
        bytes_doc = 20.0 * N * N * D
        if n_valid is not None:                                # per-document work of the entities that exist
            nl = [int(v) for v in n_valid.tolist()]
            flops = sum(algorithmic_flops_per_doc(v, D, L, H) for v in nl) / len(nl)
            bytes_doc = sum(20.0 * v * v * D for v in nl) / len(nl)
        line = {
            "metric": "docs/sec fwd+bwd through CAGGC+MAGGC", "value": round(value, 2), "unit": "docs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "strong" if args.global_batch is not None else "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.config}: GraphHops fwd+bwd, B={B}/GPU N={N} D={D} L={L} H={H}, {'EVAL mode (A/B run, not the metric)' if args.eval_mode else 'train mode'}, "
                                   f"E1/E2/X/params require grad" + (", ragged n_valid (mean %.1f)" % n_valid.float().mean().item()
                                                                      if n_valid is not None else "")
                                   + (f", rotating inputs: {nsets} resident batches ({nsets * 8 * N * N * D * B / 1e6:.0f} MB of E "
                                      "per GPU), step i runs on batch i mod " + str(nsets) if nsets > 1 else ", one batch replayed (cache-warm)"),
                       "global_batch": B * world, "mode": args.mode, "input_sets": nsets,
                       "parallelism": f"dp{world}", "grad_allreduce": ("per-block, overlapped with backward (tensor hooks)"
                                                                       if args.overlap_grads else "one coalesced collective after backward")},
            "roofline": roofline,
            "roofline_hbm": roofline_hbm,
            "whole_step": {"hbm_frac": round(value / world * bytes_doc / HBM_PEAK, 4),
                           "mfma_frac": round(value / world * flops / MFMA_F32_PEAK, 4),
                           "algorithmic_bytes_per_doc": bytes_doc, "algorithmic_flops_per_doc": flops},
            "time_shares_ms_per_step": {k: v for k, v in shares.items() if v["launches_per_step"]},
        }
        if warm is not None:
            line["extra"] = {"warm_replay_ms_per_step": round(warm * 1e3, 4), "warm_replay_docs_per_s": round(B / warm, 2),
                             "note": "one resident batch replayed back to back (the round-1 bench): its E1 is partly served from "
                                     "the 256 MiB Infinity Cache; `value` above is the rotating-input (cold) figure"}
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(cfg)
        real_stdout.write(json.dumps(line) + "\n")
        real_stdout.flush()
    if world > 1 or force_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
