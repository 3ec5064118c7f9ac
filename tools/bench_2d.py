"""BASELINE config 4: 2D 4096 x 4096 fp16 C2C, batch 64 (row pass + column pass). python tools/bench_2d.py [batch]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as g
g.build()
import tensor_fft_amd as tf
if os.environ.get("TFFT_AB_LIB"):            # A/B of another build of the library (path, e.g. build/libtfft_NAME.so)
    from tensor_fft_amd import capi
    capi._LIB_NAME = os.path.abspath(os.environ["TFFT_AB_LIB"])
    capi._lib = None                            # (g.build() above has already loaded the default build)
n = 4096
b = int(sys.argv[1]) if len(sys.argv) > 1 else 64
re = ((torch.rand(b * n * n, device="cuda") * 2 - 1)).half(); im = ((torch.rand(b * n * n, device="cuda") * 2 - 1)).half()
o_re, o_im = torch.empty_like(re), torch.empty_like(im)
plan = tf.TfftPlan2D(n, n, b, 0)
ws = torch.empty(plan.workspace_bytes // 2, dtype=torch.float16, device="cuda"); plan.set_workspace(ws)
import time
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.06:          # clock ramp
    plan.exec(re, im, o_re, o_im); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): plan.exec(re, im, o_re, o_im)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
print(f"{os.environ.get('TFFT_AB_LIB', 'libtfft.so')}: 2D {n}x{n} batch {b}: {ms:.2f} ms, launches {plan.num_launches}, {b*n*n/ms/1e6:.1f} Gsamples/s, {16*b*n*n/ms/1e6:.0f} GB/s vs 2-pass minimum")
