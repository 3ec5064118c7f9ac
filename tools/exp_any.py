"""Interleaved A/B of plan variants at arbitrary (N, batch, inner) in one process.
usage: python tools/exp_any.py N:batch[:inner] variantA variantB ..."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import debuglib  # noqa: E402,F401  (libtfft_debug.so: timing-only variants and env knobs)
import __graft_entry__ as g
g.build()
import tensor_fft_amd as tf
spec = sys.argv[1].split(":"); n, b = int(spec[0]), int(spec[1]); inner = int(spec[2]) if len(spec) > 2 else 1
variants = [int(v) for v in sys.argv[2:]] or [0]
x = ((torch.rand(b * 2 * n * inner, device="cuda") * 2 - 1)).half(); y = torch.empty_like(x)
plans = {}
for v in variants:
    p = tf.TfftPlan(n, b, 0, inner=inner, variant=v, preserve_input=True)
    ws = torch.empty(max(1, p.workspace_bytes // 2), dtype=torch.float16, device="cuda")
    if p.workspace_bytes: p.set_workspace(ws)
    plans[v] = (p, ws)
ref = None
for v in variants:
    y.zero_(); plans[v][0].exec(x, x[n * inner:], y, y[n * inner:]); torch.cuda.synchronize()
    if ref is None: ref = y.clone()
    else: print(f"variant {v} identical to {variants[0]}: {bool((y == ref).all())}")
def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
res = {v: [] for v in variants}
for rnd in range(5):
    for v in variants:
        res[v].append(timed(lambda: plans[v][0].exec(x, x[n * inner:], y, y[n * inner:])))
for v, ts in res.items():
    med = statistics.median(ts)
    print(f"N={n} batch={b} inner={inner} variant {v:5d}: median {med*1e3:9.1f} us  min {min(ts)*1e3:9.1f} us  {n*b*inner/med/1e6:7.1f} Gsamples/s  launches {plans[v][0].num_launches}")
