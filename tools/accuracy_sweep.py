"""Accuracy sweeps after the reference's protocol (src/testing/benchmarks/AccuracyTest.cu:17-86,
AccuracyTestBandwidth.cu:17-87, AccuracyTestCuFFT.cu, AccuracyTestBandwidthCuFFT.cu):

  (1) error vs N:          N = 2^8 .. 2^max, 256 harmonics (or N/2 if smaller), weights seeds 42 / 1764
  (2) error vs bandwidth:  N = 2^20, frequency cutoff 1, 2, 4, ..., N/2
  each line: `x max avg sigma` of |delta| against the fp64 DFT(x)/N of the same fp16 input (the reference uses
  cuFFT Z2Z / N; here the CPU oracle), for three columns of transforms:
     ours       - this library on the MI355X
     reference  - the oracle's fp16 restatement of the reference CUDA kernels (N <= 2^20)
     vendor     - hipFFT in fp16 and fp32 through torch.fft (the reference compares with cuFFT half / float)

Writes gnuplot-style .dat files (the reference's FileWriter format, FileWriter.h:206-225) into --outdir.
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--outdir", default="profiles")
    ap.add_argument("--max-log2", type=int, default=28)
    ap.add_argument("--tag", default="r3")
    args = ap.parse_args()
    import torch
    import __graft_entry__ as g

    g.build()
    import tensor_fft_amd as tf
    from oracle import orc
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import accuracy_protocol as proto         # the protocol itself lives with the tests (it uses the CPU oracle)

    w_re, w_im = proto.weights(orc)

    def run_all(n, cutoff):
        return proto.run_point(torch, tf, orc, n, cutoff, w_re, w_im)

    os.makedirs(args.outdir, exist_ok=True)
    cols = ["ours", "reference", "vendor_fp16", "vendor_fp32"]
    with open(os.path.join(args.outdir, f"{args.tag}_accuracy_vs_n.dat"), "w") as f:
        f.write("# N  then (max avg sigma) of |delta| vs fp64 DFT/N for: " + " | ".join(cols) + "   (nan = not run)\n")
        for lg in range(8, args.max_log2 + 1):
            n = 1 << lg
            res = run_all(n, min(256, n // 2))
            row = [str(n)]
            for c in cols:
                mx, avg, sig = res.get(c, (float("nan"),) * 3)
                row += [f"{mx:.4e}", f"{avg:.4e}", f"{sig:.4e}"]
            f.write(" ".join(row) + "\n")
            print(" ".join(row), flush=True)
    n = 1 << 20
    with open(os.path.join(args.outdir, f"{args.tag}_accuracy_vs_bandwidth.dat"), "w") as f:
        f.write("# cutoff (N = 2^20)  then (max avg sigma) for: " + " | ".join(cols) + "\n")
        cutoff = 1
        while cutoff <= n // 2:          # 1, 4, 16, ..., N/4 (the signal generator costs N * cutoff sines)
            res = run_all(n, cutoff)
            row = [str(cutoff)]
            for c in cols:
                mx, avg, sig = res.get(c, (float("nan"),) * 3)
                row += [f"{mx:.4e}", f"{avg:.4e}", f"{sig:.4e}"]
            f.write(" ".join(row) + "\n")
            print(" ".join(row), flush=True)
            cutoff *= 4
    print("done")


if __name__ == "__main__":
    main()
