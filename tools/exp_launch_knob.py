"""Launch-shape knobs of the measurement build, one child process per value, all on one box, over the multi-pass BASELINE
workloads and their neighbours:
    python tools/exp_launch_knob.py                                   TFFT_GENS = 1 2 3 4 6 8 12 16: generations of workgroups per
                                                                      resident capacity (tfft.hip gens_grid; 1 = static partition)
    KNOB=TFFT_NUM_CUS SCAN="256 224 192 512" python tools/exp_launch_knob.py    grids sized as if the chip had that many CUs"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) < 2:
    for cus in os.environ.get("SCAN", "1 2 3 4 6 8 12 16").split():
        subprocess.check_call([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, **{os.environ.get("KNOB", "TFFT_GENS"): cus}))
    sys.exit(0)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import time

import torch
import debuglib  # noqa: F401
import tensor_fft_amd as tf


def timed(fn):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.2:
        fn()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20


knob = os.environ.get("KNOB", "TFFT_GENS")
out = [f"{knob}={os.environ[knob]:>3s}:"]
for n, b in ((1 << 20, 1024), (1 << 16, 16384), (1 << 18, 4096), (1 << 15, 8192), (1 << 22, 256), (1 << 24, 64), (1 << 26, 1)):
    x = torch.empty(b * 2 * n, dtype=torch.float16, device="cuda")
    tf.synth_uniform(x, x[n:], n, b)
    y = torch.empty_like(x)
    p = tf.TfftPlan(n, b, 0, preserve_input=True)
    if p.workspace_bytes:
        p.set_workspace(torch.empty(p.workspace_bytes // 2, dtype=torch.float16, device="cuda"))
    ms = timed(lambda: p.exec(x, x[n:], y, y[n:]))
    out.append(f"2^{n.bit_length() - 1} x {b}: {n * b / ms / 1e6:6.1f}")
    del p, x, y
    torch.cuda.empty_cache()
half = 64 * 4096 * 4096
x = torch.empty(2 * half, dtype=torch.float16, device="cuda")
tf.synth_uniform(x[:half], x[half:], 4096 * 4096, 64, batch_stride=4096 * 4096)
y = torch.empty_like(x)
p2 = tf.TfftPlan2D(4096, 4096, 64, 0)
p2.set_workspace(torch.empty(p2.workspace_bytes // 2, dtype=torch.float16, device="cuda"))
ms = timed(lambda: p2.exec(x[:half], x[half:], y[:half], y[half:]))
out.append(f"2D 4096^2 x 64: {half / ms / 1e6:6.1f}")
print("  ".join(out) + "  Gsamples/s", flush=True)
