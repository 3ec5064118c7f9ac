"""(Historical: the gain this shows, +3..+8 % for two half-batches on two streams, came from workgroups idling behind the slow
residue class of the plain grid stride; with the rotated work distribution, k4096::Rotor, it is gone and the library does
not split batches.)
Sub-batches on several HIP streams: does overlapping the passes of different sub-batches (tails filled, intermediate of a
sub-batch possibly still in the 256-MiB Infinity Cache for its next pass) beat one plan over the whole batch?
usage: [TFFT_AB_LIB=libtfft_x.so] python tools/exp_streams_chunked.py N batch chunk:streams [chunk:streams ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as g
g.build()
import tensor_fft_amd as tf
if os.environ.get("TFFT_AB_LIB"):
    from tensor_fft_amd import capi
    capi._LIB_NAME = os.environ["TFFT_AB_LIB"]
    capi._lib = None
n, batch = int(sys.argv[1]), int(sys.argv[2])
x = torch.empty(batch * 2 * n, dtype=torch.float16, device="cuda"); tf.synth_uniform(x, x[n:], n, batch)
y = torch.empty_like(x)


def timed(fn, reps=5):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.08:
        fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


whole = tf.TfftPlan(n, batch, 0, preserve_input=True)
ws = torch.empty(max(1, whole.workspace_bytes // 2), dtype=torch.float16, device="cuda"); whole.set_workspace(ws)
ms = timed(lambda: whole.exec(x, x[n:], y, y[n:]))
ref = y.clone()
print(f"N={n} batch={batch} one plan: {ms*1e3:9.1f} us  {n*batch/ms/1e6:7.1f} Gsamples/s")
for spec in sys.argv[3:]:
    c, ns = (int(v) for v in spec.split(":"))
    streams = [torch.cuda.Stream() for _ in range(ns)]
    plans = []
    for _ in range(ns):
        p = tf.TfftPlan(n, c, 0, preserve_input=True)
        w = torch.empty(max(1, p.workspace_bytes // 2), dtype=torch.float16, device="cuda"); p.set_workspace(w)
        plans.append((p, w))
    main = torch.cuda.current_stream()

    def run():
        for s in streams: s.wait_stream(main)
        for i, s0 in enumerate(range(0, batch, c)):
            st = streams[i % ns]
            o = s0 * 2 * n
            plans[i % ns][0].exec(x[o:], x[o + n:], y[o:], y[o + n:], stream=st.cuda_stream)
        for s in streams: main.wait_stream(s)
    y.zero_(); run(); torch.cuda.synchronize()
    ok = bool((y == ref).all())
    ms = timed(run)
    print(f"   sub-batches of {c:5d} ({c*n*4/2**20:6.1f} MiB in) on {ns} streams: {ms*1e3:9.1f} us  {n*batch/ms/1e6:7.1f} Gsamples/s  identical {ok}")
