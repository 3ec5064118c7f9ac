"""2D 4096 x 4096 x 64 as the library runs it (two launches over the whole batch) against image-by-image chains (both passes of a
chunk of images back to back, so that the 64-MiB intermediate image is still in the 256-MiB Infinity Cache when the column pass
reads it; tools/mall_probe.hip: a producer / consumer copy chain gains 17 % that way).
    python tools/exp_2d_chunked.py [n = 4096] [images = 64] [chunk ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as g
g.build()
import tensor_fft_amd as tf
if os.environ.get("TFFT_AB_LIB"):            # another build of the library (path, e.g. build/libtfft_NAME.so)
    from tensor_fft_amd import capi
    capi._LIB_NAME = os.path.abspath(os.environ["TFFT_AB_LIB"])
    capi._lib = None

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
images = int(sys.argv[2]) if len(sys.argv) > 2 else 64
chunks = [int(v) for v in sys.argv[3:]] or [64, 1, 2, 4, 8]
half = images * n * n
x = torch.empty(2 * half, dtype=torch.float16, device="cuda")
tf.synth_uniform(x[:half], x[half:], n * n, images, batch_stride=n * n)
y = torch.empty_like(x)
ref = None
for chunk in chunks:
    plan = tf.TfftPlan2D(n, n, chunk, 0)
    ws = torch.empty(plan.workspace_bytes // 2, dtype=torch.float16, device="cuda")
    plan.set_workspace(ws)
    c = chunk * n * n

    def run():
        for i in range(0, images, chunk):
            plan.exec(x[i * n * n:i * n * n + c], x[half + i * n * n:half + i * n * n + c], y[i * n * n:i * n * n + c], y[half + i * n * n:half + i * n * n + c])

    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.2:
        run()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    same = ""
    if ref is None:
        ref = y.clone()
    else:
        same = "  output bit-identical" if bool((ref.view(torch.int16) == y.view(torch.int16)).all()) else "  OUTPUT DIFFERS"
    print(f"chunks of {chunk:2d} image(s): {ms:7.3f} ms  {half / ms / 1e6:6.1f} Gsamples/s{same}", flush=True)
    del plan, ws
