"""A/B of two builds of the library in separate processes on the same box: python tools/ab_lib.py LIBNAME N:batch ...
(LIBNAME = a library file, e.g. build/libtfft_old.so made by tools/build_ab.py from another revision). 100 ms clock ramp,
median of 5 x 20 launches."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
import tensor_fft_amd as tf
from tensor_fft_amd import capi
capi._LIB_NAME = os.path.abspath(sys.argv[1])
for spec in sys.argv[2:]:
    n, b = (int(v) for v in spec.split(":"))
    x = torch.empty(b * 2 * n, dtype=torch.float16, device="cuda"); tf.synth_uniform(x, x[n:], n, b)
    y = torch.empty_like(x)
    plan = tf.TfftPlan(n, b, 0)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.1:
        for _ in range(4): plan.exec(x, x[n:], y, y[n:])
        torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): plan.exec(x, x[n:], y, y[n:])
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 20)
    ts.sort()
    print(f"{sys.argv[1]:16s} N={n:6d}: median {ts[2]*1e3:7.1f} us  min {ts[0]*1e3:7.1f} us  {n*b/ts[2]/1e6:6.1f} Gsamples/s")
