"""A stand-in for bench.py's rank body, used by tests/test_bench_launch.py to drive bench.self_launch on CPU: N ranks over
gloo run the distributed transform's index logic with the numpy engine of tests/test_distributed_cpu.py, rank 0 prints ONE JSON
line. `--fail-rank R` makes rank R exit non-zero after the collective (the launcher must report that)."""
import argparse
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(HERE))

ap = argparse.ArgumentParser()
ap.add_argument("--gpus", type=int, default=1)
ap.add_argument("--fail-rank", type=int, default=-1)
args = ap.parse_args()

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
assert world == args.gpus == int(os.environ["WORLD_SIZE"])
import tensor_fft_amd  # noqa: E402,F401
from tensor_fft_amd.distributed import DistributedFFT1D  # noqa: E402
from test_distributed_cpu import NumpyEngine  # noqa: E402

n = 1 << 12
rng = np.random.default_rng(3)
xr, xi = rng.uniform(-1, 1, n).astype(np.float16), rng.uniform(-1, 1, n).astype(np.float16)
f = DistributedFFT1D(n, engine=NumpyEngine(), fused=False)
idx = f.input_indices()
re, im = f.forward(torch.from_numpy(xr[idx].copy()), torch.from_numpy(xi[idx].copy()))
exact = np.fft.fft(xr.astype(np.float64) + 1j * xi.astype(np.float64)) / n
got = re.numpy().astype(np.float64) + 1j * im.numpy().astype(np.float64)
want = exact[f.output_indices()]
err = torch.tensor([float(np.linalg.norm(got - want) / np.linalg.norm(want))], dtype=torch.float64)
dist.all_reduce(err, op=dist.ReduceOp.MAX)
if rank == 0:
    print(json.dumps({"metric": "launch test", "n_gpus": world, "rel_l2": float(err[0])}), flush=True)
dist.barrier()
dist.destroy_process_group()
if rank == args.fail_rank:
    sys.exit(3)
