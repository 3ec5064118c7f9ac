"""Which split is fastest when the work does not fill the chip? The default splits come from a scan at 2^30 samples per launch
(tools/plan_scan.py); a single 2^20-point transform is 16 workgroups of the radix-1024 kernel on 256 CUs. Times every planner
variant per (N, batch): device time per execution, 8 executions per HIP graph (round 5: with passes of 4-6 us an eager loop
measures the host's launch rate, not the plan), median of rounds.
    python tools/scan_small_batch.py [--min-log2 16] [--max-log2 24] [--max-total-log2 25] [--variants 0,32,...]"""
import argparse, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as g
g.build()
import tensor_fft_amd as tf

ap = argparse.ArgumentParser()
ap.add_argument("--min-log2", type=int, default=16)
ap.add_argument("--max-log2", type=int, default=24)
ap.add_argument("--max-total-log2", type=int, default=25)
ap.add_argument("--variants", default="0,1073741824,524288,8388608,33554432,41943040,1115684864,268435456,268959744,42467328")
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--rounds", type=int, default=5)
args = ap.parse_args()
variants = [int(v) for v in args.variants.split(",")]

for lg in range(args.min_log2, args.max_log2 + 1):
    n = 1 << lg
    for tl in range(lg, max(lg, args.max_total_log2) + 1):
        b = 1 << (tl - lg)
        x = torch.empty(b * 2 * n, dtype=torch.float16, device="cuda")
        tf.synth_uniform(x, x[n:], n, b)
        y = torch.empty_like(x)
        row = {}
        for v in variants:
            try:
                p = tf.TfftPlan(n, b, 0, variant=v, preserve_input=True)
            except tf.TfftError:
                continue
            ws = torch.empty(max(1, p.workspace_bytes // 2), dtype=torch.float16, device="cuda")
            if p.workspace_bytes:
                p.set_workspace(ws)
            for _ in range(3):
                p.exec(x, x[n:], y, y[n:])
            torch.cuda.synchronize()
            st = torch.cuda.Stream()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.stream(st):
                with torch.cuda.graph(gr, stream=st):
                    for _ in range(8):
                        p.exec(x, x[n:], y, y[n:], stream=st.cuda_stream)
            torch.cuda.synchronize()
            gr.replay()
            ts = []
            reps = max(1, args.reps // 8)
            for _ in range(args.rounds):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    gr.replay()
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / reps / 8 * 1e3)
            del gr
            row[v] = (statistics.median(ts), p.num_launches, tf.plan_describe(n, 1, v or tf.plan_default_variant(n, 1, b)))
            p.close()
        best = min(row, key=lambda v: row[v][0])
        print(f"N=2^{lg} batch={b:5d} (2^{tl} samples): default {row[0][0]:7.1f} us [{row[0][2]}]  best {row[best][0]:7.1f} us variant {best} "
              f"[{row[best][2]}]  " + "  ".join(f"{v}:{row[v][0]:.1f}" for v in sorted(row)), flush=True)
        del x, y
