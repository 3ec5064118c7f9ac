"""One named workload, a few back-to-back executions: the command rocprofv3 wraps when collecting kernel traces and
PMC counters for a kernel other than the headline one (python3 tools/prof_workload.py NAME [reps [warmup]]).

STEADY STATE: the `reps` counted executions follow `warmup` untimed ones (default 120: >= 200 ms of work for every workload here,
what bench.py's other_configs run before they time, bench.py:171-193). A GPU that has idled needs that long to reach its clocks
(profiles/r3_c2_per_dispatch.txt: 1.63, 2.11, 1.91, 1.79 ... ms before 1.56 for good); round 3's summaries averaged over that
transient. tools/steady_stats.py and tools/summarize_pmc.py --skip drop the warm-up dispatches (the first warmup / (warmup + reps)
of every kernel's dispatches) from what they fold.

    c1      BASELINE configs[1]   4096 x 65536                    (fft4096_kernel)
    c2      BASELINE configs[2]   2^20 x 1024                     (colfft256_wg_kernel x2 + tail)
    c2t     configs[2] with the transposed-output order (two passes); c2ti: with the transposed-input order
    c3      BASELINE configs[3]   2D 4096 x 4096 x 64             (fft4096r_kernel<8,true> + colfft512_wg_kernel)
    n8192 / n16384 / n32768       2^28 samples                    (fft4096r_kernel<R>)
    n65536, n262144, n2^26 ...    any "nLEN[:batch[:variant]]"
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as g

# runs under rocprofv3: load the prebuilt library only, never start a compiler here (tools/profile_pmc.sh builds first)
if not g.is_current():
    raise SystemExit("libtfft.so is missing or stale: run `python3 -c 'import __graft_entry__ as g; g.build()'` first")
import tensor_fft_amd as tf

name = sys.argv[1]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
warmup = int(sys.argv[3]) if len(sys.argv) > 3 else 120


def run1d(n, b, **kw):
    x = ((torch.rand(b * 2 * n, device="cuda") * 2 - 1)).half()
    y = torch.empty_like(x)
    plan = tf.TfftPlan(n, b, 0, preserve_input=True, **kw)
    ws = torch.empty(max(1, plan.workspace_bytes // 2), dtype=torch.float16, device="cuda")
    if plan.workspace_bytes:
        plan.set_workspace(ws)
    for i in range(warmup + reps):
        plan.exec(x, x[n:], y, y[n:])
        if i % 16 == 15:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    print(f"{name}: N={n} batch={b} passes={plan.num_launches} warmup={warmup} reps={reps}")


if name == "c1":
    run1d(4096, 65536)
elif name == "c2":
    run1d(1 << 20, 1024)
elif name == "c2t":
    run1d(1 << 20, 1024, output_order="transposed")
elif name == "c2ti":
    run1d(1 << 20, 1024, input_order="transposed")
elif name == "c3":
    n, b = 4096, 64
    re = ((torch.rand(b * n * n, device="cuda") * 2 - 1)).half()
    im = ((torch.rand(b * n * n, device="cuda") * 2 - 1)).half()
    o_re, o_im = torch.empty_like(re), torch.empty_like(im)
    plan = tf.TfftPlan2D(n, n, b, 0)
    plan.set_workspace(torch.empty(plan.workspace_bytes // 2, dtype=torch.float16, device="cuda"))
    for i in range(warmup + reps):
        plan.exec(re, im, o_re, o_im)
        if i % 16 == 15:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    print(f"{name}: 2D 4096x4096 x{b} passes={plan.num_launches} warmup={warmup} reps={reps}")
elif name.startswith("n"):
    f = name[1:].split(":")
    n = (1 << int(f[0][2:])) if f[0].startswith("2^") else int(f[0])
    b = int(f[1]) if len(f) > 1 else max(1, (1 << 28) // n)
    run1d(n, b, **({"variant": int(f[2])} if len(f) > 2 else {}))
else:
    raise SystemExit(f"unknown workload {name}")
