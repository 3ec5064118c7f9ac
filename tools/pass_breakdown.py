"""Per-pass time and algorithmic HBM rate of a plan: runs the chain truncated after 1, 2, ... passes (variant bits
8-11, a debugging aid of the library) and differences the timings. python tools/pass_breakdown.py N:batch[:inner] ..."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import debuglib  # noqa: E402,F401  (libtfft_debug.so: timing-only variants and env knobs)
import __graft_entry__ as g
g.build()
import tensor_fft_amd as tf

def timed(plan, x, y, n, inner, reps=8):
    plan.exec(x, x[n * inner:], y, y[n * inner:]); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): plan.exec(x, x[n * inner:], y, y[n * inner:])
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / reps)
    return statistics.median(ts)

for spec in sys.argv[1:]:
    f = spec.split(":"); n, b = int(f[0]), int(f[1]); inner = int(f[2]) if len(f) > 2 else 1
    extra = int(f[3]) if len(f) > 3 else 0
    x = ((torch.rand(b * 2 * n * inner, device="cuda") * 2 - 1)).half(); y = torch.empty_like(x)
    full = tf.TfftPlan(n, b, 0, inner=inner, preserve_input=True, variant=extra)
    ws = torch.empty(max(1, full.workspace_bytes // 2), dtype=torch.float16, device="cuda")
    if full.workspace_bytes: full.set_workspace(ws)
    np_ = full.num_launches
    prev = 0.0
    bytes_pass = 8.0 * n * b * inner
    print(f"N={n} batch={b} inner={inner} variant={extra}: {np_} passes, {bytes_pass/2**30:.2f} GiB algorithmic per pass")
    for p in range(1, np_ + 1):
        plan = tf.TfftPlan(n, b, 0, inner=inner, preserve_input=True, variant=extra | (p << 8)) if p < np_ else full
        if plan is not full and plan.workspace_bytes: plan.set_workspace(ws)
        t = timed(plan, x, y, n, inner)
        print(f"   pass {p}: {1e3*(t-prev):9.1f} us  {bytes_pass/(t-prev)/1e6:8.1f} GB/s   (cumulative {1e3*t:9.1f} us)")
        prev = t
