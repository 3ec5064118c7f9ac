"""Does the time of a multi-pass plan depend on WHERE its three streams (input, workspace, output) lie relative to each other?
Separate processes time 2^17 x 8192 at either ~351 or ~368 Gsamples/s (2^26 x 4: 193 or 201.5) with the same library and
arguments. Part 1: fresh allocations in one process (addresses printed). Part 2: one pool, the three buffers at controlled offsets.
    python tools/exp_buffer_offsets.py [N=131072] [batch=8192]"""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as g
g.build()
import tensor_fft_amd as tf

n = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
b = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
per = b * 2 * n                                   # halves per buffer
plan = tf.TfftPlan(n, b, 0, preserve_input=True)
wsh = max(1, plan.workspace_bytes // 2)


def timed(x, y, ws, reps=10, rounds=5):
    plan.set_workspace(ws)
    for _ in range(5):
        plan.exec(x, x[n:], y, y[n:])
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            plan.exec(x, x[n:], y, y[n:])
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / reps * 1e3)
    return statistics.median(ts)


print(f"N={n} batch={b}: {per * 2 / 2**20:.0f} MiB per buffer, workspace {wsh * 2 / 2**20:.0f} MiB, launches {plan.num_launches}")
print("part 1: fresh allocations")
keep = []
for i in range(6):
    if i % 2 == 1:
        keep.append(torch.empty((3 + i) * (1 << 20) + 4096 * i, dtype=torch.uint8, device="cuda"))     # shifts what follows
    x = torch.empty(per, dtype=torch.float16, device="cuda")
    tf.synth_uniform(x, x[n:], n, b)
    y = torch.empty(per, dtype=torch.float16, device="cuda")
    ws = torch.empty(wsh, dtype=torch.float16, device="cuda")
    us = timed(x, y, ws)
    print(f"  x {x.data_ptr():#x}  y {y.data_ptr():#x}  ws {ws.data_ptr():#x}   (y-x) mod 2^21 {(y.data_ptr() - x.data_ptr()) % (1 << 21):8d}  "
          f"(ws-x) mod 2^21 {(ws.data_ptr() - x.data_ptr()) % (1 << 21):8d}: {us:8.1f} us  {n * b / us / 1e3:6.1f} Gsamples/s", flush=True)
    del x, y, ws
    torch.cuda.empty_cache()
del keep
torch.cuda.empty_cache()

print("part 2: one pool, buffers at pool + k * (buffer + gap) + skew")
slack = 64 << 20
pool = torch.empty(3 * per + 3 * slack // 2 + 4096, dtype=torch.float16, device="cuda")
base = (-(pool.data_ptr()) % (1 << 21)) // 2                  # in halves: align the pool's start to 2 MiB
for skew_y, skew_ws in ((0, 0), (4096, 0), (0, 4096), (4096, 8192), (65536, 131072), (1 << 20, 1 << 19), (2 << 20, 4 << 20),
                        (4352, 8704), (1 << 16, 0), (0, 1 << 16), (3 << 12, 5 << 12), (1 << 13, 1 << 14), (1 << 17, 1 << 18)):
    ox = base
    oy = base + per + skew_y // 2
    ow = base + 2 * per + (8 << 20) + skew_ws // 2
    x, y, ws = pool[ox:ox + per], pool[oy:oy + per], pool[ow:ow + wsh]
    tf.synth_uniform(x, x[n:], n, b)
    us = timed(x, y, ws)
    print(f"  skew of y {skew_y:8d} B, of the workspace {skew_ws:8d} B (+ 8 MiB): {us:8.1f} us  {n * b / us / 1e3:6.1f} Gsamples/s", flush=True)
