"""Where does ONE transform of 2^13 .. 2^15 (k4096r::fft4096r_kernel<R>, one workgroup) spend its time? Measurement build: s_memtime
cycles per phase and wave (TFFT_ROWS_STAMPS_PTR): [7] entry -> loop start (tables, first loads issued), 0 wait for the input,
1 radix-R front end, 2 barrier B, 3 issue of the next loads, 4 stages 1-3 + staging, 6 barrier C, 5 read-out + stores acknowledged.
    python tools/exp_k4096r_phases.py [lg ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
dbg = torch.zeros(256 * 8 * 8, dtype=torch.int64, device="cuda")
os.environ["TFFT_ROWS_STAMPS_PTR"] = str(dbg.data_ptr())
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import debuglib  # noqa: E402,F401
import tensor_fft_amd as tf
for lg in [int(a) for a in sys.argv[1:]] or [13, 14, 15]:
    n = 1 << lg
    x = ((torch.rand(2 * n, device="cuda") * 2 - 1)).half()
    y = torch.empty_like(x)
    plan = tf.TfftPlan(n, 1, 0, preserve_input=True)
    for _ in range(20):
        plan.exec(x, x[n:], y, y[n:])
    torch.cuda.synchronize()
    dbg.zero_()
    plan.exec(x, x[n:], y, y[n:])
    torch.cuda.synchronize()
    t = dbg.cpu().numpy().reshape(256, 8, 8)[0].astype(np.float64)      # workgroup 0
    order = [7, 0, 1, 2, 3, 4, 6, 5]
    names = {7: "entry->loop", 0: "wait input", 1: "front end", 2: "barrier B", 3: "issue next", 4: "stages 1-3", 6: "barrier C", 5: "read-out+stores"}
    live = n // 4096
    print(f"N=2^{lg}: cycles per phase, mean over the {live} waves of the transform (total {t[:live].sum(axis=1).mean():.0f})")
    print("   " + ", ".join(f"{names[i]} {t[:live, i].mean():.0f}" for i in order))
    plan.close()
