"""Plane distance, continued: (a) N = 4096 x 65536 planar (planes 512 MiB apart) against [RE | IM] blocks, (b) the radix-512
column pass of the 2D plan (n = 512 along a strided axis of 4096 columns, 512 entries) with planes 2 GiB apart against
[RE | IM] per entry (4 MiB apart), (c) 2^20 x 1024 likewise (planar 2 GiB apart / blocks 2 MiB apart)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as g
g.build()
import tensor_fft_amd as tf


def timed(fn, reps=10):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.08:
        fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / reps)
    return sorted(ts)[1]


def case(name, n, b, inner=1, **kw):
    nf = n * inner
    tot = nf * b
    x = torch.empty(2 * tot, dtype=torch.float16, device="cuda").uniform_(-1, 1)
    y = torch.empty_like(x)
    p = tf.TfftPlan(n, b, 0, inner=inner, **kw)
    ms = timed(lambda: p.exec(x, x[nf:], y, y[nf:]))
    print(f"{name}: [RE|IM] blocks  {ms*1e3:8.1f} us  {tot/ms/1e6:6.1f} Gsamples/s  (launches {p.num_launches})")
    p2 = tf.TfftPlan(n, b, 0, inner=inner, in_batch_stride=nf, out_batch_stride=nf, **kw)
    ms = timed(lambda: p2.exec(x, x[tot:], y, y[tot:]))
    print(f"{name}: planar          {ms*1e3:8.1f} us  {tot/ms/1e6:6.1f} Gsamples/s")
    p3 = tf.TfftPlan(n, b, 0, inner=inner, in_batch_stride=nf, out_batch_stride=2 * nf, **kw)
    ms = timed(lambda: p3.exec(x, x[tot:], y, y[nf:]))
    print(f"{name}: planar in, blocks out  {ms*1e3:8.1f} us  {tot/ms/1e6:6.1f} Gsamples/s")
    p4 = tf.TfftPlan(n, b, 0, inner=inner, in_batch_stride=2 * nf, out_batch_stride=nf, **kw)
    ms = timed(lambda: p4.exec(x, x[nf:], y, y[tot:]))
    print(f"{name}: blocks in, planar out  {ms*1e3:8.1f} us  {tot/ms/1e6:6.1f} Gsamples/s")


case("N=4096 x 65536", 4096, 65536)
case("N=32768 x 32768", 32768, 32768)
case("n=512 inner=4096 x 512 (2D column pass)", 512, 512, inner=4096, variant=67108864)
case("N=2^20 x 1024", 1 << 20, 1024)
