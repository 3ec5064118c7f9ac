"""Times tfft_exec for arbitrary (N, batch) on one GPU: python tools/bench_any.py N:batch [N:batch ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as g
g.build()
import tensor_fft_amd as tf
for spec in sys.argv[1:]:
    n, b = (int(v) for v in spec.split(":"))
    x = ((torch.rand(b * 2 * n, device="cuda") * 2 - 1)).to(torch.float16)
    y = torch.empty_like(x)
    plan = tf.TfftPlan(n, b, 0)
    ws = torch.empty(max(1, plan.workspace_bytes // 2), dtype=torch.float16, device="cuda")
    if plan.workspace_bytes: plan.set_workspace(ws)
    import time
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.06:          # clock ramp (the GPU idles between entries)
        for _ in range(4): plan.exec(x, x[n:], y, y[n:])
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    e0.record()
    for _ in range(reps): plan.exec(x, x[n:], y, y[n:])
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"N={n:>9} batch={b:>7} launches={plan.num_launches}  {ms*1e3:9.1f} us  {n*b/ms/1e6:8.1f} Gsamples/s  "
          f"{8*n*b/ms/1e6:8.1f} GB/s min-traffic  {plan.algorithmic_bytes/ms/1e6:8.1f} GB/s incl. passes")
