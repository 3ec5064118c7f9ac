"""Would overlapping consecutive chunks of the chunked 2D plan on TWO streams (each with its own intermediate) hide the ramp and
tail of the ~95-us launches? python tools/exp_2d_two_streams.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as g
g.build()
import tensor_fft_amd as tf

n, images, chunk = 4096, 64, 4
half = images * n * n
x = torch.empty(2 * half, dtype=torch.float16, device="cuda")
tf.synth_uniform(x[:half], x[half:], n * n, images, batch_stride=n * n)
y = torch.empty_like(x)
whole = tf.TfftPlan2D(n, n, images, 0)
whole.set_workspace(torch.empty(whole.workspace_bytes // 2, dtype=torch.float16, device="cuda"))
plans, streams = [], []
for i in range(3):
    p = tf.TfftPlan2D(n, n, chunk, 0)
    p.set_workspace(torch.empty(p.workspace_bytes // 2, dtype=torch.float16, device="cuda"))
    plans.append(p)
    streams.append(torch.cuda.Stream())
c = chunk * n * n


def one_stream():
    whole.exec(x[:half], x[half:], y[:half], y[half:])


def multi(k):
    def run():
        for j, i in enumerate(range(0, images, chunk)):
            s = streams[j % k]
            plans[j % k].exec(x[i * n * n:i * n * n + c], x[half + i * n * n:half + i * n * n + c], y[i * n * n:i * n * n + c],
                              y[half + i * n * n:half + i * n * n + c], stream=s.cuda_stream)
    return run


for name, fn in (("library plan, one stream", one_stream), ("chunks round-robin over 2 streams", multi(2)), ("over 3 streams", multi(3)),
                 ("library plan, one stream", one_stream)):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.2:
        fn()
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 10 * 1e3
    print(f"{name:36s}: {ms:7.3f} ms  {half / ms / 1e6:6.1f} Gsamples/s", flush=True)
