"""What does one wave per SIMD cost a column pass? (a radix-1024 pass with 128-column tiles = 256-byte row
segments needs 192 VGPRs of partial results per lane, i.e. 512-register waves, one per SIMD.) The cooperative radix-256 kernel with
4-wave workgroups normally runs two workgroups per CU (two waves per SIMD); the measurement build can launch it with so much dynamic
LDS that only one fits (TFFT_WG4_ONE_PER_CU=1): same kernel, same 128-byte segments, one wave per SIMD. Two processes, one box:
    python tools/exp_one_wave_per_simd.py            (prints both; re-executes itself with the knob set, before any GPU call)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) < 2:
    for knob in ("0", "1"):
        env = dict(os.environ, TFFT_WG4_ONE_PER_CU=knob)
        subprocess.check_call([sys.executable, os.path.abspath(__file__), "child"], env=env)
    sys.exit(0)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import time

import torch
import debuglib  # noqa: F401
import tensor_fft_amd as tf

one = os.environ.get("TFFT_WG4_ONE_PER_CU") == "1"
for n, inner, batch in ((256, 4096, 1024), (256, 1024, 4096), (256, 16384, 256)):
    nf = n * inner
    x = ((torch.rand(batch * 2 * nf, device="cuda") * 2 - 1)).half()
    y = torch.empty_like(x)
    p = tf.TfftPlan(n, batch, 0, inner=inner, variant=524288, preserve_input=True)      # 524288: 4-wave cooperative workgroups
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.2:
        p.exec(x, x[nf:], y, y[nf:])
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        p.exec(x, x[nf:], y, y[nf:])
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"radix-256 column pass, {inner:5d} columns x batch {batch:4d}, 4-wave workgroups, {'ONE per CU (1 wave / SIMD)' if one else 'two per CU (2 waves / SIMD)'}: "
          f"{ms * 1e3:8.1f} us  {8.0 * nf * batch / ms / 1e6:6.0f} GB/s", flush=True)
