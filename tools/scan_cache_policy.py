"""Cache policy of the column passes against the plan's footprint: streaming (non-temporal, variant bit 536870912) against
plain accesses (variant bit 262144) and the library's own choice (variant 0), in ONE process per (N, batch), median of rounds.
Two situations per shape: "hot" = the same buffers again and again (what a caller that iterates on one data set sees: a footprint
below the 256-MiB Infinity Cache can stay in it), "cold" = a ring of buffer sets of > 1 GiB in total (every execution finds its
input in HBM).
    [ORDER=transposed | IN_ORDER=transposed] [POLICIES=name=variant,...] python tools/scan_cache_policy.py N:batch [N:batch ...]"""
import os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as g
g.build()
import tensor_fft_amd as tf

POL = (("stream", 536870912), ("plain", 262144), ("default", 0))
if os.environ.get("POLICIES"):        # e.g. POLICIES=stream=10,plain=2,default=0 for the N = 4096 kernel (its non-temporal bit is 8)
    POL = tuple((kv.split("=")[0], int(kv.split("=")[1])) for kv in os.environ["POLICIES"].split(","))
KW = dict(output_order=os.environ.get("ORDER", "natural"), input_order=os.environ.get("IN_ORDER", "natural"))
for spec in sys.argv[1:]:
    n, b = (int(v) for v in spec.split(":"))
    per = b * 2 * n                                                  # halves per buffer
    sets = max(2, -(-(1 << 30) // (2 * per * 2)))                    # ring of (in, out) pairs, > 1 GiB in total
    if sets * 2 * per * 2 > (48 << 30):
        sets = 2
    ring = [(torch.empty(per, dtype=torch.float16, device="cuda"), torch.empty(per, dtype=torch.float16, device="cuda")) for _ in range(sets)]
    for x, _ in ring:
        tf.synth_uniform(x, x[n:], n, b)
    plans = []
    for name, v in POL:
        p = tf.TfftPlan(n, b, 0, variant=v, preserve_input=True, **KW)
        ws = torch.empty(max(1, p.workspace_bytes // 2), dtype=torch.float16, device="cuda")
        if p.workspace_bytes:
            p.set_workspace(ws)
        plans.append((name, p, ws))
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.1:
        plans[0][1].exec(ring[0][0], ring[0][0][n:], ring[0][1], ring[0][1][n:])
        torch.cuda.synchronize()
    res = {(name, mode): [] for name, _, _ in plans for mode in ("hot", "cold")}
    reps = 24
    for _ in range(7):
        for name, p, _ in plans:
            for mode in ("hot", "cold"):
                bufs = ring[:1] if mode == "hot" else ring
                for i in range(3):
                    x, y = bufs[i % len(bufs)]
                    p.exec(x, x[n:], y, y[n:])
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for i in range(reps):
                    x, y = bufs[i % len(bufs)]
                    p.exec(x, x[n:], y, y[n:])
                e1.record()
                torch.cuda.synchronize()
                res[(name, mode)].append(e0.elapsed_time(e1) / reps * 1e3)
    foot = 3 * 4 * n * b / 2 ** 20
    med = {k: statistics.median(v) for k, v in res.items()}
    print(f"N=2^{n.bit_length() - 1:2d} batch={b:5d} footprint {foot:7.0f} MiB launches={plans[0][1].num_launches} ring={sets:4d}: " +
          "  ".join(f"{mode} " + " / ".join(f"{med[(name, mode)]:7.1f}" for name, _, _ in plans) for mode in ("hot", "cold")) +
          "  us (" + " / ".join(name for name, _ in POL) + ")", flush=True)
    del plans, ring
    torch.cuda.empty_cache()
