"""Interleaved A/B timing of kernel variants and a plain HBM copy on one GPU (one process, rule 24).
usage: python tools/exp_bench.py [variant ...]   (variants = values for TfftPlan(..., variant=))"""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import debuglib  # noqa: E402,F401  (libtfft_debug.so: timing-only variants and env knobs)
import __graft_entry__ as g
g.build()
import tensor_fft_amd as tf

N, B = 4096, int(os.environ.get("EXP_BATCH", "65536"))
variants = [int(v) for v in sys.argv[1:]] or [0]
x = ((torch.rand(B * 2 * N, device="cuda") * 2 - 1)).to(torch.float16)
y = torch.empty_like(x)
plans = {v: tf.TfftPlan(N, B, 0, variant=v) for v in variants}
ref = None
for v in variants:
    y.zero_(); plans[v].exec(x, x[N:], y, y[N:]); torch.cuda.synchronize()
    if ref is None: ref = y.clone(); print(f"variant {v}: reference")
    else: print(f"variant {v}: identical to variant {variants[0]}: {bool((y == ref).all())}")
def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
res = {("copy",): []}
for v in variants: res[(v,)] = []
for rnd in range(7):
    res[("copy",)].append(timed(lambda: y.copy_(x)))
    for v in variants:
        res[(v,)].append(timed(lambda: plans[v].exec(x, x[N:], y, y[N:])))
gb = 2 * x.numel() * 2 / 1e9
for k, ts in res.items():
    print(f"{str(k[0]):>6}: median {statistics.median(ts)*1e3:8.1f} us  min {min(ts)*1e3:8.1f} us  -> {gb/statistics.median(ts)*1e3:7.1f} GB/s (min-time {gb/min(ts)*1e3:7.1f})")
