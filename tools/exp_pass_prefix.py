"""What each dependent launch of a small plan costs: device time per execution (16 executions per HIP graph) of the first 1, 2, ...
passes of the plan alone (measurement build: variant bits 8-11 = "run only the first p passes", WRONG output), differences = one
pass with the gap in front of it.
    python tools/exp_pass_prefix.py [lg[:batch] ...]"""
import os, sys
os.environ["TFFT_DEBUG_VARIANTS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import debuglib  # noqa: E402,F401
import tensor_fft_amd as tf

cases = sys.argv[1:] or ["16", "17", "18", "19", "20", "21", "22"]
for c in cases:
    f = c.split(":")
    lg, b = int(f[0]), int(f[1]) if len(f) > 1 else 1
    n = 1 << lg
    var = tf.plan_default_variant(n, 1, b)
    x = ((torch.rand(b * 2 * n, device="cuda") * 2 - 1)).half()
    y = torch.empty_like(x)
    full = tf.TfftPlan(n, b, 0, preserve_input=True, variant=var)
    np_ = full.num_launches
    desc = tf.plan_describe(n, 1, var)
    full.close()
    line, prev = f"N=2^{lg} x {b} [{desc}]:", 0.0
    for k in list(range(1, np_)) + [0]:
        plan = tf.TfftPlan(n, b, 0, preserve_input=True, variant=var | (k << 8))
        ws = torch.empty(max(1, plan.workspace_bytes // 2), dtype=torch.float16, device="cuda")
        if plan.workspace_bytes:
            plan.set_workspace(ws)
        plan.exec(x, x[n:], y, y[n:])
        torch.cuda.synchronize()
        s = torch.cuda.Stream()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.stream(s):
            with torch.cuda.graph(gr, stream=s):
                for _ in range(16):
                    plan.exec(x, x[n:], y, y[n:], stream=s.cuda_stream)
        torch.cuda.synchronize()
        ts = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(8):
                gr.replay()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 8 / 16 * 1e3)
        t = sorted(ts)[3]
        line += f"   first {k if k else np_}: {t:6.2f} us (+{t - prev:5.2f})"
        prev = t
        plan.close()
        del gr
    print(line, flush=True)
