"""Workgroup shapes of the latency column kernel (collat.hpp: columns per workgroup x waves per column group), in ONE process on
the measurement build (TFFT_LAT_SHAPE = digits CG HH [PP]: 42 / 22 / 14 / 222 / 142, read at every launch; 0 = the throughput kernels via variant bit
1073741824): error against numpy's fp64 FFT and device time per transform (16 executions per HIP graph).
    python tools/exp_lat_shapes.py [lg[:batch] ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import debuglib  # noqa: E402,F401
import tensor_fft_amd as tf

SPLIT_256 = 8388608 | 33554432
NO_LAT = 1073741824
cases = sys.argv[1:] or ["14", "16", "17", "18", "19", "20", "21", "20:4", "18:16", "16:64"]
for c in cases:
    f = c.split(":")
    lg, b = int(f[0]), int(f[1]) if len(f) > 1 else 1
    n = 1 << lg
    rng = np.random.default_rng(lg)
    h = rng.uniform(-1, 1, (b, 2, n)).astype(np.float16)
    ref = np.fft.fft(h[:, 0].astype(np.float64) + 1j * h[:, 1].astype(np.float64), axis=1) / n
    x = torch.from_numpy(h).cuda().reshape(-1)
    y = torch.empty_like(x)
    line = f"N=2^{lg} x {b}:"
    for shape in (0, 42, 22, 14, 222, 142):
        os.environ["TFFT_LAT_SHAPE"] = str(shape) if shape else "0"
        var = SPLIT_256 | (16777216 if lg < 16 else 0) | (NO_LAT if shape == 0 else 0)
        plan = tf.TfftPlan(n, b, 0, preserve_input=True, variant=var)
        ws = torch.empty(max(1, plan.workspace_bytes // 2), dtype=torch.float16, device="cuda")
        if plan.workspace_bytes:
            plan.set_workspace(ws)
        y.fill_(float("nan"))
        plan.exec(x, x[n:], y, y[n:])
        torch.cuda.synchronize()
        got = y.cpu().numpy().reshape(b, 2, n).astype(np.float64)
        err = np.linalg.norm((got[:, 0] + 1j * got[:, 1]) - ref) / np.linalg.norm(ref)
        s = torch.cuda.Stream()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.stream(s):
            with torch.cuda.graph(gr, stream=s):
                for _ in range(16):
                    plan.exec(x, x[n:], y, y[n:], stream=s.cuda_stream)
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(8):
                gr.replay()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 8 / 16 * 1e3)
        line += f"   shape {shape:3d}: {sorted(ts)[2]:6.2f} us (err {err:.1e})"
        plan.close()
        del gr
    print(line, flush=True)
