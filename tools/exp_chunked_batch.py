"""Does running a long-transform batch in cache-sized sub-batches (all passes of one sub-batch back to back) beat one
plan over the whole batch? The 256-MiB Infinity Cache could keep a sub-batch's intermediate between passes.
usage: [TFFT_VARIANT=262144] [ORDER=transposed | IN_ORDER=transposed] [TFFT_AB_LIB=build/libtfft_X.so] python tools/exp_chunked_batch.py N batch chunk [chunk ...]
(262144 = no non-temporal accesses in the radix-256 column kernels)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as g
g.build()
import tensor_fft_amd as tf
if os.environ.get("TFFT_AB_LIB"):            # another build of the library (file name under tensor-fft_amd/)
    from tensor_fft_amd import capi
    capi._LIB_NAME = os.path.abspath(os.environ["TFFT_AB_LIB"])
    capi._lib = None                            # (g.build() above has already loaded the default build)
n, batch = int(sys.argv[1]), int(sys.argv[2])
chunks = [int(v) for v in sys.argv[3:]]
x = ((torch.rand(batch * 2 * n, device="cuda") * 2 - 1)).half(); y = torch.empty_like(x)
def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for c in [batch] + chunks:
    plan = tf.TfftPlan(n, c, 0, preserve_input=True, variant=int(os.environ.get("TFFT_VARIANT", "0")),
                       output_order=os.environ.get("ORDER", "natural"), input_order=os.environ.get("IN_ORDER", "natural"))
    ws = torch.empty(max(1, plan.workspace_bytes // 2), dtype=torch.float16, device="cuda")
    if plan.workspace_bytes: plan.set_workspace(ws)
    def run():
        for s in range(0, batch, c):
            o = s * 2 * n
            plan.exec(x[o:], x[o + n:], y[o:], y[o + n:])
    ms = min(timed(run) for _ in range(3))
    print(f"N={n} batch={batch} in sub-batches of {c:5d} ({c*n*4/2**20:7.1f} MiB in, same out, workspace {plan.workspace_bytes/2**20:7.1f} MiB): "
          f"{ms*1e3:9.1f} us  {n*batch/ms/1e6:7.1f} Gsamples/s")
