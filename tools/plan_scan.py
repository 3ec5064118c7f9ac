"""Times every column-pass split (radices 256 / 512 / 1024, every order) with the fewest passes for N = 2^lg, in one process,
through the experiment knob TFFT_PLAN_COLS (needs TFFT_DEBUG_VARIANTS=1, set here). The planner's defaults in
tfft.hip (plan_passes) follow the winners of this scan (profiles/r2_plan_scan.txt).

    python tools/plan_scan.py 21 22 23 24 [--total-log2 30]"""
import argparse
import itertools
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import debuglib  # noqa: E402,F401  (libtfft_debug.so: timing-only variants and env knobs)
import torch
import __graft_entry__ as g

g.build()
import tensor_fft_amd as tf

ap = argparse.ArgumentParser()
ap.add_argument("lgs", nargs="+", type=int)
ap.add_argument("--total-log2", type=int, default=30)
args = ap.parse_args()
BITS = {256: 8, 512: 9, 1024: 10}


def candidates(lg):
    out = []
    for k in (1, 2, 3):
        for cols in itertools.product((256, 512, 1024), repeat=k):
            bits = sum(BITS[c] for c in cols)
            t = lg - bits
            if 0 <= t <= 6:
                out.append((k + (1 if t else 0), cols))
    if not out:
        return []
    best = min(c[0] for c in out)
    return [c[1] for c in out if c[0] == best]


for lg in args.lgs:
    n = 1 << lg
    b = max(1, (1 << args.total_log2) // n)
    x = torch.empty(b * 2 * n, dtype=torch.float16, device="cuda")
    tf.synth_uniform(x, x[n:], n, b)
    y = torch.empty_like(x)
    os.environ.pop("TFFT_PLAN_COLS", None)
    plans = {"default": tf.TfftPlan(n, b, 0, preserve_input=True)}
    descr = {"default": tf.plan_describe(n, 1, 0)}
    for cols in candidates(lg):
        key = ",".join(str(c) for c in cols)
        os.environ["TFFT_PLAN_COLS"] = key
        plans[key] = tf.TfftPlan(n, b, 0, preserve_input=True)
        descr[key] = tf.plan_describe(n, 1, 0)
    os.environ.pop("TFFT_PLAN_COLS", None)
    ws_bytes = max(p.workspace_bytes for p in plans.values())
    ws = torch.empty(max(1, ws_bytes // 2), dtype=torch.float16, device="cuda")
    ref = None
    for key, p in plans.items():
        if p.workspace_bytes:
            p.set_workspace(ws)
        y.zero_()
        p.exec(x, x[n:], y, y[n:])
        torch.cuda.synchronize()
        if ref is None:
            ref = y.float()
        else:
            err = float((y.float() - ref).abs().max() / ref.abs().max())
            assert err < 2e-2, (lg, key, err)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.1:
        plans["default"].exec(x, x[n:], y, y[n:])
        torch.cuda.synchronize()
    res = {k: [] for k in plans}
    for rnd in range(4):
        for key, p in plans.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                p.exec(x, x[n:], y, y[n:])
            e1.record()
            torch.cuda.synchronize()
            res[key].append(e0.elapsed_time(e1) / 5)
    rows = sorted((statistics.median(v), k) for k, v in res.items())
    for med, key in rows:
        print(f"2^{lg} x {b}: {key:>16s}  {med * 1e3:8.1f} us  {n * b / med / 1e6:6.1f} Gsamples/s   {descr[key]}", flush=True)
    del plans, x, y, ws
