"""Single N = 2^26 transform (BASELINE configs[4b]) on whatever ranks are present: one rank = the local passes of the
distributed driver without the exchange, compared with the plain single-GPU plan of the same length.
    python tools/bench_dist.py [log2N=26]
    python -m torch.distributed.run --nproc-per-node 8 ... tools/bench_dist.py 26     (8-GPU node)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as g
g.build()
import tensor_fft_amd as tf
from tensor_fft_amd.distributed import DistributedFFT1D, HipEngine

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 26
n = 1 << lg
local_rank = int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(local_rank)
dist = None
if "RANK" in os.environ:
    import torch.distributed as dist
    dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
world = dist.get_world_size() if dist else 1
rank = dist.get_rank() if dist else 0
f = DistributedFFT1D(n, engine=HipEngine(local_rank), input_layout="columns", output_layout="transposed")
loc = n // world
re = (torch.rand(loc, device="cuda") * 2 - 1).half(); im = (torch.rand(loc, device="cuda") * 2 - 1).half()
def fence():
    torch.cuda.synchronize()
    if dist: dist.barrier(); torch.cuda.synchronize()
for _ in range(2): f.forward(re, im)
fence()
t0 = time.perf_counter()
reps = 5
for _ in range(reps): f.forward(re, im)
fence()
dt = (time.perf_counter() - t0) / reps
if rank == 0:
    print(f"distributed driver: N=2^{lg} on {world} rank(s): {dt*1e3:.3f} ms per transform, {n/dt/1e9:.2f} Gsamples/s "
          f"(N1={f.n1}, N2={f.n2}, {f.c} columns / {f.k} rows per rank)")
if world == 1:
    x = torch.cat([re, im]); y = torch.empty_like(x)
    plan = tf.TfftPlan(n, 1, 0, preserve_input=True)
    ws = torch.empty(plan.workspace_bytes // 2, dtype=torch.float16, device="cuda"); plan.set_workspace(ws)
    for _ in range(2): plan.exec(x, x[n:], y, y[n:])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): plan.exec(x, x[n:], y, y[n:])
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    print(f"single-GPU plan:    N=2^{lg}: {dt*1e3:.3f} ms, {n/dt/1e9:.2f} Gsamples/s, {plan.num_launches} launches")
if dist: dist.destroy_process_group()
