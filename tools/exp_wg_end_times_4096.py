"""Per-XCC busy time of the N = 4096 kernel (debug build libtfft_dbg.so, not part of the product)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import numpy as np
dbg = torch.zeros(3 * 8192, dtype=torch.int64, device="cuda")
os.environ["TFFT_DBG_PTR"] = str(dbg.data_ptr())
import tensor_fft_amd as tf
from tensor_fft_amd import capi
capi._LIB_NAME = "libtfft_dbg.so"
n, b = 4096, 65536
x = torch.empty(b * 2 * n, dtype=torch.float16, device="cuda"); tf.synth_uniform(x, x[n:], n, b)
y = torch.empty_like(x)
p = tf.TfftPlan(n, b, 0)
for _ in range(200): p.exec(x, x[n:], y, y[n:])
torch.cuda.synchronize()
t = dbg.cpu().numpy()
grid = 4096
st, en, xcc = t[:grid].astype(np.float64), t[8192:8192 + grid].astype(np.float64), (t[16384:16384 + grid] & 15)
t0 = st.min()
e = (en - t0) / 100
print(f"grid {grid}: kernel span {e.max():.1f} us; WG durations us: min {((en-st)/100).min():.1f} median {np.median((en-st)/100):.1f} max {((en-st)/100).max():.1f}")
print("blockIdx % 8 == XCC id for", int((xcc == (np.arange(grid) % 8)).sum()), "of", grid)
for x8 in range(8):
    sel = xcc == x8
    print(f"  XCC {x8}: {int(sel.sum()):5d} workgroups, busy time {((en-st)[sel]).sum()/100/32:9.1f} us per CU (2 WGs resident), last end {e[sel].max():8.1f} us")
