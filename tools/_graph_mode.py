"""Shared by the row benches (tail / head / producer / train step): time a step the way bench.py times the hot path.

    run = capture(step)          # one hipGraph of the whole step (forward + backward [+ optimiser]); None if capture is refused
    timed(step, run, steps)      # {"eager": ms, "graph": ms} -- host-issued launches vs replays of the captured graph

A step must not read anything back to the host (the library's kernels keep every count on the device) and must leave its
results in tensors that the replays rewrite in place: `.grad = None` before backward inside the step is fine, the tensors
backward then allocates belong to the graph's private pool and every replay writes them again.
"""
import sys
import time

import torch


def capture(step, warmup=3):
    try:
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(warmup):
                step()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            step()
        torch.cuda.synchronize()
        return g
    except Exception as ex:   # report and carry on eagerly: a bench must not lose its line to a capture problem
        print(f"hipGraph capture failed ({ex!r}); eager timing only", file=sys.stderr)
        torch.cuda.synchronize()
        return None


def wall_ms(fn, steps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def timed(step, graph, steps, mode="both"):
    out = {}
    if mode in ("eager", "both") or graph is None:
        for _ in range(3):
            step()
        out["eager"] = wall_ms(step, steps)
    if graph is not None and mode in ("graph", "both"):
        for _ in range(3):
            graph.replay()
        out["graph"] = wall_ms(graph.replay, steps)
    return out
