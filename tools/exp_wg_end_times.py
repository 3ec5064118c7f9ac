"""When and where do the workgroups of a radix-1024 column pass run? The kernel records wall_clock64() at entry / exit and the
XCC id per workgroup into the buffer named by TFFT_WG_TIMES_PTR (honoured only with TFFT_DEBUG_VARIANTS=1; both set here).
    [TFFT_COLWG_ITERS=4] [TFFT_VARIANT=256] python tools/exp_wg_end_times.py        (256 = first pass only)
profiles/r2_wg_end_times.txt holds the runs that found the slow residue class."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import numpy as np
dbg = torch.zeros(4 * 16 * 8192, dtype=torch.int64, device="cuda")     # one block of 16 x 8192 words per pass of the plan
os.environ["TFFT_WG_TIMES_PTR"] = str(dbg.data_ptr())
os.environ["TFFT_NO_SPLIT"] = "1"
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import debuglib  # noqa: E402,F401  (libtfft_debug.so: timing-only variants and env knobs)
import tensor_fft_amd as tf
n, b = 1 << 20, 1024
x = torch.empty(b * 2 * n, dtype=torch.float16, device="cuda"); tf.synth_uniform(x, x[n:], n, b)
y = torch.empty_like(x)
p = tf.TfftPlan(n, b, 0, preserve_input=True, variant=int(os.environ.get("TFFT_VARIANT", "0")))
ws = torch.empty(p.workspace_bytes // 2, dtype=torch.float16, device="cuda"); p.set_workspace(ws)
for _ in range(20): p.exec(x, x[n:], y, y[n:])
torch.cuda.synchronize()
t = dbg.cpu().numpy()[int(os.environ.get("TFFT_PASS", "1")) * 16 * 8192:]           # the stamps of pass TFFT_PASS (default: the second) of the last execution
iters = int(os.environ.get("TFFT_COLWG_ITERS", "1000000"))
grid = min(8192, max(256, (16384 + iters - 1) // iters)) if iters < 1000000 else 256
st, en, xcc = t[:grid].astype(np.float64), t[8192:8192 + grid].astype(np.float64), (t[16384:16384 + grid] & 15)
t0 = st.min()
e = (en - t0) / 100             # us (100 MHz counter)
print(f"grid {grid}: kernel span {e.max():.1f} us; WG durations us: min {((en-st)/100).min():.1f} median {np.median((en-st)/100):.1f} max {((en-st)/100).max():.1f}")
print("blockIdx % 8 == XCC id for", int((xcc == (np.arange(grid) % 8)).sum()), "of", grid, "workgroups")
for x8 in range(8):
    sel = xcc == x8
    print(f"  XCC {x8}: {int(sel.sum()):5d} workgroups, busy time {((en-st)[sel]).sum()/100/32:9.1f} us per CU, last end {e[sel].max():8.1f} us")
print("idle fraction if all wait for the last: %.3f" % (1 - sum(((en-st)/100)) / 256 / e.max()))
