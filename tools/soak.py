"""Randomised soak of the plan space through the C ABI against numpy's fp64 FFT: lengths 2 .. 2^22, batches, padded and
planar strides, in place, preserve_input, strided axes, scale modes, transposed output and transposed input order (with
batches that end in a tail chunk of the chunked execution), the four-step twiddle, every tuner variant. Each case runs twice (bit-identical results required).

    python tools/soak.py [--seconds 120] [--seed 1]

Prints one line per failure and a summary; exit code 1 on any failure. (The fixed-seed subsets that run in CI are
tests/test_gpu_parity.py::test_randomized_* ; this tool is for longer runs on a GPU box.)"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
REL_L2_TOL = 1.5e-3


def _c(re, im):
    return np.asarray(re, dtype=np.float64) + 1j * np.asarray(im, dtype=np.float64)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120.0)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    import torch
    import __graft_entry__ as g

    g.build()
    import tensor_fft_amd as tf
    import tuner

    rng = np.random.default_rng(args.seed)
    t_end = time.time() + args.seconds
    cases = fails = 0
    t_last = time.time()
    kinds = {}
    while time.time() < t_end:
        kind = str(rng.choice(["plain", "plain", "strided", "transposed", "transposed_in", "fourstep", "scale", "variant", "launch"]))
        kw = {}
        inner = 1
        if kind == "strided":
            lg = int(rng.integers(1, 15))
            inner = int(rng.choice([8, 16, 32, 64, 128, 256, 1024]))
        elif kind == "transposed":
            lg = int(rng.integers(16, 23))
            kw["output_order"] = "transposed"
            kw["scale"] = str(rng.choice(["sequential", "sequential", "none", "once"]))
        elif kind == "transposed_in":               # round 4: the [N1][N2] layout as INPUT, natural order out
            lg = int(rng.integers(16, 23))
            kw["input_order"] = "transposed"
            kw["scale"] = str(rng.choice(["sequential", "sequential", "none"]))
        elif kind == "fourstep":
            lg = int(rng.choice([8, 9]))
            inner = int(rng.choice([64, 128, 512, 2048]))
            m = (1 << lg) * inner * int(rng.choice([1, 2, 8]))
            kw["fourstep_n"] = m
            kw["fourstep_col0"] = int(rng.integers(0, m // (1 << lg) - inner + 1)) // 8 * 8
            kw["scale"] = str(rng.choice(["sequential", "sequential", "none", "once"]))
        elif kind == "scale":
            lg = int(rng.integers(1, 22))
            kw["scale"] = str(rng.choice(["none", "once"]))
        elif kind == "variant":
            lg = int(rng.integers(8, 23))
            kw["variant"] = int(rng.choice(tuner.candidates(1 << lg)))
        elif kind == "launch":                      # launch shapes (round 3): any rounds-per-workgroup value, any length
            lg = int(rng.integers(8, 23))
            kw["launch_iters"] = int(rng.choice([1, 2, 3, 4, 7, 65535]))
            if rng.integers(0, 2):
                kw["variant"] = int(rng.choice(tuner.candidates(1 << lg)))
        else:
            lg = int(rng.integers(1, 23))
        n = 1 << lg
        nf = n * inner
        batch = int(rng.integers(1, max(2, min(40, (1 << 22) // nf))))
        if kind in ("transposed", "transposed_in") and lg <= 18 and rng.integers(0, 3) == 0:
            batch = (1 << 27) // n + int(rng.integers(1, 9))      # one full chunk of the chunked execution plus a tail chunk
        pad = int(rng.integers(0, 3)) * 8 if nf >= 8 and kind in ("plain", "scale", "variant") else 0
        in_place = bool(rng.integers(0, 2)) and pad == 0
        preserve = bool(rng.integers(0, 2)) and not in_place
        amp = 1.0
        if kw.get("scale") == "none":
            amp = min(1.0, 8192.0 / n)
        re = (rng.uniform(-1, 1, (batch, n, inner)) * amp).astype(np.float16)
        im = (rng.uniform(-1, 1, (batch, n, inner)) * amp).astype(np.float16)
        sig = _c(re, im)
        if kind == "transposed_in":                 # the block handed over is the [N1][N2] matrix: x[k1 + N1 k2] = in[k1 N2 + k2]
            n2 = tf.transposed_n2(n)
            sig = sig.reshape(batch, n // n2, n2).transpose(0, 2, 1).reshape(batch, n, 1)
        exact = np.fft.fft(sig, axis=1) / n
        if kw.get("scale") == "none":
            exact = exact * n
        if kind == "fourstep":
            k = np.arange(n, dtype=np.int64)[:, None]
            c = (kw["fourstep_col0"] + np.arange(inner, dtype=np.int64))[None, :]
            exact = exact * np.exp(-2j * np.pi * ((k * c) % kw["fourstep_n"]) / kw["fourstep_n"])[None]
        if kind == "transposed":
            n2 = tf.transposed_n2(n)
            exact = exact.reshape(batch, n2, n // n2).transpose(0, 2, 1).reshape(batch, n, 1)
        stride = 2 * nf + pad
        blk = torch.zeros(batch, stride, dtype=torch.float16, device="cuda")
        blk[:, :nf] = torch.from_numpy(re.reshape(batch, nf)).cuda()
        blk[:, nf:2 * nf] = torch.from_numpy(im.reshape(batch, nf)).cuda()
        keep = blk.clone()
        try:
            plan = tf.TfftPlan(n, batch, 0, inner=inner, in_batch_stride=stride, out_batch_stride=stride,
                               preserve_input=preserve, **kw)
            outs = []
            for rep in range(2):
                src = keep.clone()
                out = src if in_place else torch.full_like(src, float("nan"))
                fi, fo = src.reshape(-1), out.reshape(-1)
                plan.exec(fi, fi[nf:], fo, fo[nf:])
                torch.cuda.synchronize()
                if preserve and not bool((src == keep).all()):
                    raise AssertionError("input clobbered despite preserve_input")
                outs.append(out)
            if not bool((outs[0][:, :2 * nf] == outs[1][:, :2 * nf]).all()):
                raise AssertionError("two runs differ")
            got = _c(outs[0][:, :nf].cpu().numpy(), outs[0][:, nf:2 * nf].cpu().numpy()).reshape(exact.shape)
            rel = np.linalg.norm(got - exact) / np.linalg.norm(exact)
            if not (np.isfinite(got).all() and rel <= REL_L2_TOL):
                raise AssertionError(f"rel-L2 {rel:.3e}")
        except Exception as e:      # noqa: BLE001 - report and go on
            fails += 1
            print(f"FAIL {kind} n=2^{lg} inner={inner} batch={batch} pad={pad} in_place={in_place} preserve={preserve} {kw}: {e}", flush=True)
        cases += 1
        kinds[kind] = kinds.get(kind, 0) + 1
        if time.time() - t_last > 60:             # (a long silent run looks hung to a job watchdog)
            t_last = time.time()
            print(f"... {cases} cases, {fails} failure(s)", flush=True)
        del blk, keep
    print(f"soak: {cases} cases ({kinds}), {fails} failure(s), seed {args.seed}")
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
