"""Property check of very long single transforms (the reference benches up to 2^29): a plane wave must land in one bin."""
import os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as g
g.build()
import tensor_fft_amd as tf
for lg in [int(v) for v in sys.argv[1:]] or [27, 28]:
    n = 1 << lg
    f0 = 123456789 % n
    idx = torch.arange(n, device="cuda", dtype=torch.int64)
    ph = ((idx * f0) % n).double() * (2 * math.pi / n)
    x = torch.empty(2 * n, dtype=torch.float16, device="cuda")
    x[:n] = torch.cos(ph).half(); x[n:] = torch.sin(ph).half()
    del idx, ph
    y = torch.empty_like(x)
    plan = tf.TfftPlan(n, 1, 0, preserve_input=True)
    plan.exec(x, x[n:], y, y[n:]); torch.cuda.synchronize()
    mag = torch.sqrt(y[:n].float() ** 2 + y[n:].float() ** 2)
    peak = int(mag.argmax()); pv = float(mag[peak]); mag[peak] = 0
    print(f"N=2^{lg}: launches {plan.num_launches}, peak at {peak} (expected {f0}) value {pv:.4f}, largest other bin {float(mag.max()):.2e}, "
          f"Parseval out/in = {float((y.float()**2).sum() / ((x.float()**2).sum() / n)):.4f}")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); plan.exec(x, x[n:], y, y[n:]); e1.record(); torch.cuda.synchronize()
    print(f"   {e0.elapsed_time(e1):.3f} ms, {n / e0.elapsed_time(e1) / 1e6:.1f} Gsamples/s")
    del x, y, mag, plan
    torch.cuda.empty_cache()
