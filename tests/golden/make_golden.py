"""Regenerates tests/golden/*.npz from the CPU oracle (oracle/tfft_oracle.cpp).

The reference stores no golden vectors and cannot run here (CUDA only), so these
fixtures are NOT reference outputs: they freeze (a) GetRandomWeights values, which
are libstdc++-defined and therefore identical to what the reference's harness
draws (TestingDataCreation.h:15-27; the seed-42 / seed-4242 values also appear in
SURVEY.md section 4), and (b) the oracle's own outputs on the reference benchmark
signal (Bench.h:84-87), so that a change in the oracle is noticed.

Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import orc  # noqa: E402

out = os.path.dirname(os.path.abspath(__file__))

# (a) weights: UnitTest.cu:22-24 uses seeds 42*i and 42*42*i, Bench.h:84-85 uses 42 / 4242,
# AccuracyTest.cu uses 42 / 1764.
seeds = [0, 42, 84, 1764, 3528, 4242]
np.savez(os.path.join(out, "weights.npz"), **{f"seed_{s}": orc.random_weights(20, s) for s in seeds})

# (b) benchmark signal (10 harmonics, seeds 42 / 4242) and the oracle's outputs on it.
w_re, w_im = orc.random_weights(10, 42), orc.random_weights(10, 4242)
blobs = {}
for n, modes in ((256, (0,)), (4096, (0, 1)), (8192, (0, 1))):
    re, im = orc.sine_superposition(n, w_re, w_im, 10)
    blobs[f"in_re_{n}"] = re.view(np.uint16)
    blobs[f"in_im_{n}"] = im.view(np.uint16)
    for m in modes:
        rr, ri = orc.ref_fft(re, im, m)
        blobs[f"ref_re_{n}_mode{m}"] = rr.view(np.uint16)[0]
        blobs[f"ref_im_{n}_mode{m}"] = ri.view(np.uint16)[0]
np.savez_compressed(os.path.join(out, "bench_signal.npz"), **blobs)
print("wrote", sorted(os.listdir(out)))
