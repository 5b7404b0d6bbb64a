#!/usr/bin/env python3
"""Reference points for the headline size: per-launch time (hipGraph replay, dependent launches on
one stream) of (a) a near-empty kernel, (b) a plain device copy moving the same number of bytes
as one step of 65 536 envs x 8 traffic (23.66 MB read+write), (c) the same at larger sizes.
Tells how much of a step launch is launch/boundary + memory-system floor that no kernel body can
remove."""
import torch
dev = "cuda:0"


def bench(fn, n=100, reps=20):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (n * reps)


tiny = torch.zeros(64, device=dev)
print("near-empty kernel (64 floats add_): %.2f us/launch" % bench(lambda: tiny.add_(1.0)))
for mb, label in ((23.66 / 2, "23.66 MB r+w (headline step)"), (23.66, "47 MB r+w (2x)"),
                  (23.66 * 8, "378 MB r+w (16x)"), (23.66 * 32, "1.5 GB r+w (64x)")):
    n = int(mb * 1e6 / 4)
    a, b = torch.randn(n, device=dev), torch.empty(n, device=dev)
    state = {"f": True}

    def cp():
        if state["f"]:
            b.copy_(a)
        else:
            a.copy_(b)
        state["f"] = not state["f"]
    us = bench(cp, n=50 if mb > 100 else 100, reps=10)
    print("copy %s: %.2f us/launch = %.0f GB/s" % (label, us, 2 * n * 4 / us / 1e3))
n = int(23.66e6 / 2 / 4)
a = torch.randn(n, device=dev)
b = torch.empty(n, device=dev)
print("elementwise sin (same bytes as the headline step, more VALU): %.2f us/launch" % bench(lambda: torch.sin(a, out=b)))
