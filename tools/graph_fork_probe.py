#!/usr/bin/env python3
"""Does a forked branch of a hipGraph (or a side stream in eager mode) run concurrently with a chain of small
latency-bound kernels on MI355X?  One 69 GFLOP matmul beside 30 tiny elementwise kernels, serial vs forked.
Measured (ROCm 7.0 runtime of torch 2.10): serial 566 / 575 us (eager / graph), forked 633 / 645 us -- the fork costs more
than it overlaps, in a graph as well as across streams.  Hence every overlap in this repository is INSIDE a launch
(passenger workgroups), never across streams."""
import torch, time
dev = torch.device("cuda:0")
a = torch.randn(4096, 2048, device=dev); b = torch.randn(2048, 4096, device=dev)
small = torch.randn(64, 64, device=dev)
side = torch.cuda.Stream()

def main_chain(n=30):
    y = small
    for _ in range(n):
        y = y * 1.0001 + 0.1
    return y

def serial():
    c = a @ b
    y = main_chain()
    return c, y

def forked():
    cur = torch.cuda.current_stream()
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        c = a @ b
    y = main_chain()
    cur.wait_stream(side)
    return c, y

def timeit(fn, n=200):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6

def graphed(fn):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3): fn()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = fn()
    return g

print("eager serial   %.1f us" % timeit(serial))
print("eager forked   %.1f us" % timeit(forked))
gs = graphed(serial); gf = graphed(forked)
print("graph serial   %.1f us" % timeit(gs.replay))
print("graph forked   %.1f us" % timeit(gf.replay))
print("gemm alone     %.1f us" % timeit(lambda: a @ b))
print("chain alone    %.1f us" % timeit(graphed(main_chain).replay))
