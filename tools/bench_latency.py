"""Latency of ONE transform (batch 1, the reference's FFTBenchSinlge.cu protocol: N = 2^8 .. 2^28): eager tfft_exec calls
back to back, the same launches replayed from a HIP graph, and the GPU-side time of one transform (events around a
burst of 50). usage: python tools/bench_latency.py [lg ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as g
g.build()
import tensor_fft_amd as tf
lgs = [int(v) for v in sys.argv[1:]] or [8, 10, 12, 13, 14, 16, 18, 20, 22, 24, 26]
print("#  N      passes  eager_us  graph_us   (per transform, GPU events over 50 back-to-back executions)")
for lg in lgs:
    n = 1 << lg
    x = ((torch.rand(2 * n, device="cuda") * 2 - 1)).half(); y = torch.empty_like(x)
    plan = tf.TfftPlan(n, 1, 0, preserve_input=True)
    ws = torch.empty(max(1, plan.workspace_bytes // 2), dtype=torch.float16, device="cuda")
    if plan.workspace_bytes: plan.set_workspace(ws)
    s = torch.cuda.Stream()
    def burst(fn, reps=50):
        with torch.cuda.stream(s):
            fn(); s.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s)
            for _ in range(reps): fn()
            e1.record(s); s.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3
    eager = min(burst(lambda: plan.exec(x, x[n:], y, y[n:], s.cuda_stream)) for _ in range(3))
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=s):
        plan.exec(x, x[n:], y, y[n:], s.cuda_stream)
    rep = min(burst(lambda: graph.replay()) for _ in range(3))
    print(f"2^{lg:<3d}  {plan.num_launches:6d}  {eager:8.1f}  {rep:8.1f}")
