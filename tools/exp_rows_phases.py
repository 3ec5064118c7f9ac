"""Where does the fused 2D row pass (k4096r::fft4096r_kernel<8, true>) spend its time? The measurement build sums, per wave,
the s_memtime cycles of each phase of its loop over all iterations into the buffer named by TFFT_ROWS_STAMPS_PTR:
    0 wait for this iteration's input (vmcnt)      1 radix-8 front end (MFMA + twiddles + LDS writes)   2 barrier B
    3 issue of the next iteration's 16 loads       4 stages 1-3 + staging to LDS                        5 read-back + 16 stores
    6 barrier D
python tools/exp_rows_phases.py [images]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import numpy as np
dbg = torch.zeros(256 * 8 * 8, dtype=torch.int64, device="cuda")
os.environ["TFFT_ROWS_STAMPS_PTR"] = str(dbg.data_ptr())
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import debuglib  # noqa: E402,F401
import tensor_fft_amd as tf
n = 4096
b = int(sys.argv[1]) if len(sys.argv) > 1 else 64
half = b * n * n
x = ((torch.rand(2 * half, device="cuda") * 2 - 1)).half()
y = torch.empty_like(x)
plan = tf.TfftPlan2D(n, n, b, 0)
plan.set_workspace(torch.empty(plan.workspace_bytes // 2, dtype=torch.float16, device="cuda"))
for _ in range(30):
    plan.exec(x[:half], x[half:], y[:half], y[half:])
torch.cuda.synchronize()
t = dbg.cpu().numpy().reshape(256, 8, 8).astype(np.float64)
iters = b * 512 / 256
names = ["wait input", "front end", "barrier B", "issue loads", "stages 1-3", "read-back+stores", "barrier D", "-"]
tot = t.sum(axis=2)
print(f"cycles per iteration and wave (mean over 256 workgroups x 8 waves), {iters:.0f} iterations per workgroup; total {tot.mean()/iters:.0f}")
for i in range(7):
    v = t[:, :, i] / iters
    print(f"  {names[i]:18s} mean {v.mean():8.0f}   waves 0-3 {v[:, :4].mean():8.0f}   waves 4-7 {v[:, 4:].mean():8.0f}   min {v.min():8.0f} max {v.max():8.0f}")
