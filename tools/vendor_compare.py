"""Throughput of the vendor FFT (hipFFT / rocFFT through torch.fft) on the same shapes, for context: complex32 (half,
interleaved) and complex64. Not a like-for-like replacement: planar fp16 in/out and 1/N scaling are this library's
contract. usage: python tools/vendor_compare.py [N:batch ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as g
g.build()
import tensor_fft_amd as tf

def timed(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

specs = sys.argv[1:] or ["256:1048576", "4096:65536", "8192:32768", "65536:4096", "1048576:256"]
print("#        N    batch   this library   vendor complex32   vendor complex64   (Gsamples/s)")
for spec in specs:
    n, b = (int(v) for v in spec.split(":"))
    x = ((torch.rand(b * 2 * n, device="cuda") * 2 - 1)).half(); y = torch.empty_like(x)
    plan = tf.TfftPlan(n, b, 0, preserve_input=True)
    ws = torch.empty(max(1, plan.workspace_bytes // 2), dtype=torch.float16, device="cuda")
    if plan.workspace_bytes: plan.set_workspace(ws)
    ours = n * b / timed(lambda: plan.exec(x, x[n:], y, y[n:])) / 1e6
    del plan, y, ws
    xc = torch.view_as_complex(x.reshape(b, n, 2).float().contiguous())
    v64 = n * b / timed(lambda: torch.fft.fft(xc, dim=1)) / 1e6
    try:
        xh = xc.to(torch.complex32)
        v32 = n * b / timed(lambda: torch.fft.fft(xh, dim=1)) / 1e6
        v32s = f"{v32:16.1f}"
    except Exception as e:   # noqa: BLE001
        v32s = f"{'n/a: ' + type(e).__name__:>16}"
    print(f"{n:10d} {b:8d} {ours:14.1f} {v32s} {v64:18.1f}")
    del x, xc
    torch.cuda.empty_cache()

# 2D 4096 x 4096 x 16 (BASELINE configs[3] at a quarter of its batch: the vendor path needs complex64 copies of the planar data)
n, b = 4096, 16
half = b * n * n
x = ((torch.rand(2 * half, device="cuda") * 2 - 1)).half()
y = torch.empty_like(x)
p2 = tf.TfftPlan2D(n, n, b, 0)
p2.set_workspace(torch.empty(p2.workspace_bytes // 2, dtype=torch.float16, device="cuda"))
ours = half / timed(lambda: p2.exec(x[:half], x[half:], y[:half], y[half:])) / 1e6
xc = torch.complex(x[:half].float(), x[half:].float()).reshape(b, n, n)
v64 = half / timed(lambda: torch.fft.fft2(xc), reps=5) / 1e6
print(f"2D {n}x{n} x {b}: this library {ours:8.1f}   vendor complex64 {v64:8.1f}   Gsamples/s")
