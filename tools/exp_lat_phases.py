"""Where does a single transform's time go INSIDE its column passes? Measurement build: wave 0 of every workgroup of the radix-256
workgroup kernel stamps the wall clock (100 MHz) and the shader clock at entry / loads issued / barrier A (block in LDS) / barrier B
(stage 1 done) / stage 2 done / stores issued / stores acknowledged (colfft.hpp TFFT_WG_STAMP; TFFT_WG_TIMES_PTR, one block of
16 x 8192 words per pass). The transform runs 16 x per HIP graph, the stamps are those of the last execution.
    python tools/exp_lat_phases.py [lg[:batch[:variant]] ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
dbg = torch.zeros(4 * 16 * 8192, dtype=torch.int64, device="cuda")
os.environ["TFFT_WG_TIMES_PTR"] = str(dbg.data_ptr())
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import debuglib  # noqa: E402,F401
import tensor_fft_amd as tf

cases = sys.argv[1:] or ["16", "18", "20", "20:4"]
for c in cases:
    f = c.split(":")
    n, b = 1 << int(f[0]), int(f[1]) if len(f) > 1 else 1
    kw = {"variant": int(f[2])} if len(f) > 2 else {}
    x = torch.empty(b * 2 * n, dtype=torch.float16, device="cuda")
    tf.synth_uniform(x, x[n:], n, b)
    y = torch.empty_like(x)
    plan = tf.TfftPlan(n, b, 0, preserve_input=True, **kw)
    ws = torch.empty(max(1, plan.workspace_bytes // 2), dtype=torch.float16, device="cuda")
    if plan.workspace_bytes:
        plan.set_workspace(ws)
    for _ in range(3):
        plan.exec(x, x[n:], y, y[n:])
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        with torch.cuda.graph(gr, stream=s):
            for _ in range(16):
                plan.exec(x, x[n:], y, y[n:], stream=s.cuda_stream)
    torch.cuda.synchronize()
    dbg.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        gr.replay()
    e0.record()
    for _ in range(8):
        gr.replay()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 8 / 16 * 1e3
    t = dbg.cpu().numpy().astype(np.float64).reshape(4, 16, 8192)
    desc = tf.plan_describe(n, 1, kw.get("variant", tf.plan_default_variant(n, 1, b)))
    print(f"== N=2^{f[0]} x {b}  [{desc}]  {us:.2f} us per transform (graph of 16)")
    prev_end = None
    for p in range(4):
        live = t[p, 0] > 0
        if not live.any():
            continue
        w = t[p][:, live]
        t0 = w[0].min()
        rel = (w[:7] - t0) / 100.0                       # us since the first workgroup's entry
        clk = (w[8 + 6] - w[8 + 0]) / np.maximum(w[6] - w[0], 1) * 100.0      # MHz
        names = ["entry", "loads issued", "A: block in LDS", "B: stage 1 done", "stage 2 done", "stores issued", "stores done"]
        med = np.median(rel, axis=1)
        print(f"  pass {p}: {int(live.sum())} workgroups, span {rel[6].max():.2f} us (last entry at {rel[0].max():.2f}); shader clock {np.median(clk):.0f} MHz"
              + (f"; gap since the previous pass's last exit {(t0 - prev_end) / 100:.2f} us" if prev_end else ""))
        print("     median us since the kernel's first entry: " + ", ".join(f"{nm} {m:.2f}" for nm, m in zip(names, med)))
        prev_end = w[6].max()
    plan.close()
