"""Time of ONE transform per length (the reference's FFTBenchSinlge.cu protocol: a single transform, executions back to back), three
ways: eager tfft_exec calls from Python (host launch cost included), the same execution captured once in a HIP graph and replayed,
and a graph of 16 executions (device time per execution with the host out of the way).
    python tools/latency_single.py [--min-log2 8] [--max-log2 26] [--batch 1]"""
import argparse, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as g
g.build()
import tensor_fft_amd as tf

ap = argparse.ArgumentParser()
ap.add_argument("--min-log2", type=int, default=8)
ap.add_argument("--max-log2", type=int, default=26)
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--reps", type=int, default=50)
ap.add_argument("--rounds", type=int, default=5)
args = ap.parse_args()


def timed(fn, reps, rounds, per):
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / reps / per * 1e3)
    return statistics.median(ts)


b = args.batch
for lg in range(args.min_log2, args.max_log2 + 1):
    n = 1 << lg
    x = torch.empty(b * 2 * n, dtype=torch.float16, device="cuda")
    tf.synth_uniform(x, x[n:], n, b)
    y = torch.empty_like(x)
    plan = tf.TfftPlan(n, b, 0, preserve_input=True)
    ws = torch.empty(max(1, plan.workspace_bytes // 2), dtype=torch.float16, device="cuda")
    if plan.workspace_bytes:
        plan.set_workspace(ws)
    for _ in range(3):
        plan.exec(x, x[n:], y, y[n:])
    torch.cuda.synchronize()
    eager = timed(lambda: plan.exec(x, x[n:], y, y[n:]), args.reps, args.rounds, 1)
    s = torch.cuda.Stream()
    graphs = {}
    for k in (1, 16):
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.stream(s):
            with torch.cuda.graph(gr, stream=s):
                for _ in range(k):
                    plan.exec(x, x[n:], y, y[n:], stream=s.cuda_stream)
        torch.cuda.synchronize()
        gr.replay()
        torch.cuda.synchronize()
        graphs[k] = gr
    g1 = timed(graphs[1].replay, args.reps, args.rounds, 1)
    g16 = timed(graphs[16].replay, max(4, args.reps // 8), args.rounds, 16)
    desc = tf.plan_describe(n, 1, tf.plan_default_variant(n, 1, b))
    print(f"N=2^{lg:2d} x {b}: eager {eager:7.1f} us   graph of 1 {g1:7.1f} us   graph of 16 {g16:7.1f} us per transform "
          f"({n * b / g16 / 1e3:7.2f} Gsamples/s)   [{desc}]", flush=True)
    plan.close()
    del graphs
