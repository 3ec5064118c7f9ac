#!/usr/bin/env python3
"""bench.py — headline benchmark of the hot path: batched 1D N=4096 fp16 C2C FFT (BASELINE.json configs[1]).

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one tfft_exec over one resident batch of 65536 transforms (2^28 complex samples, 1 GiB in +
1 GiB out in HBM). With N GPUs every rank owns its own batch of the same size (the batch shards with no
data-path collective: SURVEY 8e), so scaling is weak and `value` is the aggregate over all ranks.

Rank 0 prints ONE JSON line. `roofline` is the HBM roofline of the dominant kernel (algorithmic bytes =
8 B per complex sample per launch: 4 read + 4 written) with the launch duration measured here
by HIP events on the launch stream. `cpu_baseline` is the CPU oracle's fp64 FFT (oracle/, a port: the
reference has no CPU path, its oracle is cuFFT on the GPU) timed on this host's cores on a bounded sample.
At N=1 GPU a further object `other_configs` carries short measurements of the other BASELINE configs (2^20 x 1024,
2D 4096^2 x 64, single 2^26) and neighbouring lengths, taken AFTER the timed region; they do not enter `value`.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N = 4096
BATCH = 65536
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
MFMA_PEAK_TFLOPS = 2500.0      # dense fp16 MFMA


RAMP = 100          # untimed launches before the warmup steps (GPU clock ramp, ~35 ms)


def cpu_baseline(seconds_target=12.0):
    """fp64 radix-2 FFT/N of the oracle over OpenMP threads on a bounded sample of the same workload."""
    import numpy as np
    from oracle import orc

    threads = orc.num_threads()
    rng = np.random.default_rng(0)
    probe = 64 * threads
    re = rng.uniform(-1, 1, (probe, N)).astype(np.float16)
    im = rng.uniform(-1, 1, (probe, N)).astype(np.float16)
    orc.dft64(re[:threads], im[:threads])                       # warm
    t0 = time.perf_counter()
    orc.dft64(re, im)
    rate = probe / (time.perf_counter() - t0)                    # FFTs / s
    reps = max(1, int(rate * seconds_target / probe))           # bounded: about seconds_target of CPU work
    t0 = time.perf_counter()
    for _ in range(reps):
        orc.dft64(re, im)
    dt = time.perf_counter() - t0
    done = reps * probe
    return {
        "value": done * N / dt / 1e9,
        "unit": "Gsamples/s",
        "cores": threads,
        "kind": "port",
        "sample": f"{done} FFTs of N={N} ({done / BATCH:.2f} x the GPU batch of {BATCH}; the same {probe} random "
                  f"transforms repeated), oracle fp64 radix-2 FFT/N, OpenMP over the batch, {dt:.1f} s",
    }


def other_configs(torch, tf, device):
    """Short measurements of the other BASELINE configs and neighbouring lengths on the same GPU, after the headline
    timing (not part of `value`): Gsamples/s over 10 back-to-back executions each, inputs resident, workspace preset."""
    out = {}

    def timed(fn, reps=10):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    for name, n, b in (("n256_x_1048576", 256, 1 << 20), ("n1024_x_262144", 1024, 1 << 18),
                       ("n8192_x_32768", 8192, 1 << 15), ("n65536_x_4096", 1 << 16, 1 << 12),
                       ("configs[2]_n2^20_x_1024", 1 << 20, 1024), ("n2^24_x_16", 1 << 24, 16),
                       ("configs[4b]_single_gpu_n2^26_x_1", 1 << 26, 1)):
        x = (torch.rand(b * 2 * n, device="cuda") * 2 - 1).to(torch.float16)
        y = torch.empty_like(x)
        plan = tf.TfftPlan(n, b, device, preserve_input=True)
        ws = torch.empty(max(1, plan.workspace_bytes // 2), dtype=torch.float16, device="cuda")
        if plan.workspace_bytes:
            plan.set_workspace(ws)
        ms = timed(lambda: plan.exec(x, x[n:], y, y[n:]))
        out[name] = {"gsamples_per_s": n * b / ms / 1e6, "ms": ms, "passes": plan.num_launches}
        del plan, x, y, ws
        torch.cuda.empty_cache()
    rows = cols = 4096
    images = 64
    x = (torch.rand(images * 2 * rows * cols, device="cuda") * 2 - 1).to(torch.float16)
    y = torch.empty_like(x)
    plan2 = tf.TfftPlan2D(rows, cols, images, device)
    half = images * rows * cols                      # fully planar: all RE images, then all IM images
    ms = timed(lambda: plan2.exec(x[:half], x[half:], y[:half], y[half:]), reps=5)
    out["configs[3]_2d_4096x4096_x_64"] = {"gsamples_per_s": half / ms / 1e6, "ms": ms, "passes": plan2.num_launches}
    return out


def shard(rank, world, total):
    """Contiguous slice [lo, hi) of `total` independent transforms owned by `rank`: the whole multi-GPU story of
    the batched path (FFTs are independent, no data-path collective). bench.py itself runs weak
    scaling (every rank a full BASELINE batch); this helper documents and tests the strong-scaling split."""
    per = total // world
    extra = total % world
    lo = rank * per + min(rank, extra)
    return lo, lo + per + (1 if rank < extra else 0)


def measured_traffic(kernel_name):
    """HBM bytes per launch from the newest committed PMC summary (tools/summarize_pmc.py; FETCH_SIZE x2 +
    WRITE_SIZE from separate rocprofv3 --pmc passes of this same command), or None."""
    import glob

    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_summary.json"))):
        try:
            d = json.load(open(path))
        except (OSError, ValueError):
            continue
        if d.get("kernel") == kernel_name and "hbm_bytes_per_dispatch" in d:
            best = (d["hbm_bytes_per_dispatch"]["total"], os.path.relpath(path, ROOT))
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=BATCH, help="transforms per GPU (default: the BASELINE config)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the short extra measurements of the other BASELINE configs (reported under 'other_configs')")
    args = ap.parse_args()

    import torch
    import __graft_entry__ as g

    g.build()
    import tensor_fft_amd as tf

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or "RANK" in os.environ:      # under torch.distributed.run even a single rank takes the RCCL path
        import torch.distributed as dist

        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))   # RCCL: barrier + max only

    tf.device_check(local_rank)
    batch = args.batch
    gen = torch.Generator(device="cuda").manual_seed(42 + rank)
    # synthetic planar fp16, uniform(-1,1), DataBatchHandler layout [fft_i RE | fft_i IM], resident in HBM
    x = torch.empty(batch * 2 * N, dtype=torch.float16, device="cuda")
    chunk = 1 << 26
    for s in range(0, x.numel(), chunk):
        e = min(x.numel(), s + chunk)
        x[s:e] = (torch.rand(e - s, device="cuda", generator=gen) * 2 - 1).to(torch.float16)
    y = torch.empty_like(x)
    plan = tf.TfftPlan(N, batch, local_rank, preserve_input=True)

    def step():
        plan.exec(x, x[N:], y, y[N:])

    token = torch.zeros(1, device="cuda") if dist is not None else None

    def fence():
        """barrier + synchronize. The barrier is a one-element all-reduce on a preallocated tensor (what dist.barrier()
        does, minus its extra device synchronisations, which cost ~1 ms per call and would be charged to the K steps)."""
        torch.cuda.synchronize()
        if dist is not None:
            dist.all_reduce(token)
            torch.cuda.synchronize()

    # The GPU leaves its idle clock state only after a few milliseconds of work (20 steps timed cold read 0.382 ms
    # per step, steady state 0.354 ms): RAMP untimed launches first, reported in the JSON line, then the contract's
    # W warmup steps and K timed steps.
    for _ in range(RAMP):
        step()
    fence()
    for _ in range(args.warmup):
        step()
    fence()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        step()
    ev1.record()
    fence()
    wall = time.perf_counter() - t0
    kernel_ms = ev0.elapsed_time(ev1) / args.steps            # launches are back to back on one stream
    if dist is not None:
        t = torch.tensor([wall, kernel_ms], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall, kernel_ms = float(t[0]), float(t[1])

    # per-launch spread (SURVEY 8d asks for mean and sigma; the true mean, not the reference's sum / (n - 1),
    # BenchUtil.h:41-48): 20 individually timed launches after the timed region, not part of `value`
    singles = []
    for _ in range(20):
        a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a0.record()
        step()
        a1.record()
        torch.cuda.synchronize()
        singles.append(a0.elapsed_time(a1))
    single_mean = sum(singles) / len(singles)
    single_sigma = (sum((v - single_mean) ** 2 for v in singles) / (len(singles) - 1)) ** 0.5

    # light self-check so a broken kernel cannot post a number: Parseval on a slice
    xs = x[: 64 * 2 * N].float().reshape(64, 2 * N)
    ys = y[: 64 * 2 * N].float().reshape(64, 2 * N)
    par = float((((ys ** 2).sum(1) - (xs ** 2).sum(1) / N).abs() / ((xs ** 2).sum(1) / N)).max())
    if not par < 5e-3:
        raise SystemExit(f"self-check failed: Parseval mismatch {par:.3e}")

    if rank == 0:
        samples_per_step = float(N) * batch * world
        value = samples_per_step * args.steps / wall / 1e9
        alg_bytes = plan.algorithmic_bytes                      # per launch, this rank
        achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
        mfma_tflops = plan.mfma_flops / (kernel_ms * 1e-3) / 1e12
        traffic = measured_traffic(plan.kernel_name) if batch == BATCH else None
        line = {
            "metric": "Gsamples/s + %fp16-MFMA-peak, batched N=4096 fp16 C2C FFT",
            "value": value,
            "unit": "Gsamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": wall / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f16",
            "data": "synthetic",
            "config": {
                "workload": "BASELINE configs[1]: batched 1D N=4096 fp16 C2C FFT, batch=65536 per GPU, "
                            "planar [RE|IM] blocks resident in HBM, result = DFT(x)/N",
                "n": N,
                "batch_per_gpu": batch,
                "clock_ramp_launches": RAMP,
                "parallelism": f"batch sharded over {world} GPU(s), no data-path collective",
            },
            "roofline": {
                "bound": "hbm",
                "kernel": plan.kernel_name,
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic[0] if traffic else None,
                "traffic_source": traffic[1] if traffic else None,
                "kernel_ms": kernel_ms,
                "kernel_ms_single_launches": {"mean": single_mean, "sigma": single_sigma, "n": len(singles)},
                "algorithmic_bytes_per_launch": alg_bytes,
            },
            "mfma": {"tflops": mfma_tflops, "peak": MFMA_PEAK_TFLOPS, "frac": mfma_tflops / MFMA_PEAK_TFLOPS,
                     "flop_per_sample": 384},
        }
        if world == 1 and not args.no_other_configs and batch == BATCH:
            del x, y
            torch.cuda.empty_cache()
            line["other_configs"] = other_configs(torch, tf, local_rank)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
