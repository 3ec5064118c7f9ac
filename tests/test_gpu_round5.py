"""Round 5 on the GPU: the latency kernels of single transforms (the reference's own benchmark protocol,
FFTBenchSinlge.cu:11-15: one transform per length) against the CPU oracle at batch 1, 2, 3, the plans tfft_plan_default_variant
picks for small work, and the wisdom file a caller loads through the C ABI."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REL_L2_TOL = 1.5e-3          # the library's stated tolerance against the fp64 DFT / N
NO_LAT = 1073741824          # tfft_plan_opts.variant: column passes of small work by the throughput kernels
SPLIT_256 = 8388608 | 33554432   # no radix-512 / radix-1024 passes: N = 256 x 256 x R


@pytest.fixture(scope="module")
def tf():
    import __graft_entry__ as g

    g.build()
    import tensor_fft_amd as t

    t.device_check(0)
    return t


def _run(tf, torch, n, batch, seed, **kw):
    x = torch.empty(batch * 2 * n, dtype=torch.float16, device="cuda")
    tf.synth_uniform(x, x[n:], n, batch, seed=seed)
    y = torch.full_like(x, float("nan"))
    plan = tf.TfftPlan(n, batch, 0, preserve_input=True, **kw)
    plan.exec(x, x[n:], y, y[n:])
    torch.cuda.synchronize()
    plan.close()
    return y


def _check(orc, y, n, batch, seed, ids=None):
    worst = 0.0
    for b in (range(batch) if ids is None else ids):
        re, im = orc.synth_uniform(n, 1, b, seed)
        e_re, e_im = orc.dft64(re, im)
        o = y[b * 2 * n:(b + 1) * 2 * n].cpu().numpy().astype(np.float64)
        got, exact = o[:n] + 1j * o[n:], e_re[0] + 1j * e_im[0]
        assert np.isfinite(got).all()
        worst = max(worst, float(np.linalg.norm(got - exact) / np.linalg.norm(exact)))
    return worst


@pytest.mark.parametrize("lg", [14, 15, 16, 17, 18, 19, 20, 21])
@pytest.mark.parametrize("batch", [1, 2, 3])
def test_latency_column_kernel_against_the_oracle(tf, orc, lg, batch):
    """N = 256 x 256 x R (256 x 64 / 256 x 128 for 2^14 / 2^15) with the column passes on collat256_kernel (at most two blocks per
    CU: every batch here), both output forms, with and without the next pass's twiddles: every transform of the batch against
    orc.dft64; and the same plan on the throughput kernels (variant bit 1073741824) agrees with it to well inside the tolerance."""
    import torch

    n = 1 << lg
    var = SPLIT_256 | (16777216 if lg < 16 else 0)
    assert "col:256" in tf.plan_describe(n, 1, var)
    y = _run(tf, torch, n, batch, 50 + lg, variant=var)
    err = _check(orc, y, n, batch, 50 + lg)
    assert err <= REL_L2_TOL, err
    y_thr = _run(tf, torch, n, batch, 50 + lg, variant=var | NO_LAT)
    d = (y.float() - y_thr.float()).double()
    rel = float(d.norm() / y_thr.double().norm())
    assert rel <= 4e-4, rel          # two roundings to binary16 apart at most (hardware sin / cos against table twiddles)


@pytest.mark.parametrize("lg", list(range(13, 23)))
def test_default_plan_of_a_single_transform_against_the_oracle(tf, orc, lg):
    """Whatever tfft_plan_create picks for ONE transform (variant 0: tfft_plan_default_variant, possibly a loaded wisdom line)."""
    import torch

    n = 1 << lg
    for batch in (1, 3):
        y = _run(tf, torch, n, batch, 70 + lg)
        assert _check(orc, y, n, batch, 70 + lg) <= REL_L2_TOL


def test_latency_kernel_many_blocks_and_strided_axis(tf, orc):
    """More than one block per workgroup slot (batch 7 of 2^20: 448 blocks on 256 CUs), the columns-in-registers form along a
    strided axis (inner = 64 columns: 256-point transforms of a [256][64] matrix), in place."""
    import torch

    n, batch = 1 << 20, 7
    y = _run(tf, torch, n, batch, 91, variant=SPLIT_256)
    assert _check(orc, y, n, batch, 91, ids=(0, 3, 6)) <= REL_L2_TOL
    # strided axis: data [batch][256][64], transform along the 256 axis
    inner, nn, b = 64, 256, 5
    rng = np.random.default_rng(3)
    h = rng.uniform(-1, 1, (b, 2, nn, inner)).astype(np.float16)
    x = torch.from_numpy(h).cuda().reshape(-1)
    out = torch.full_like(x, float("nan"))
    plan = tf.TfftPlan(nn, b, 0, inner=inner, preserve_input=True)
    plan.exec(x, x[nn * inner:], out, out[nn * inner:])
    torch.cuda.synchronize()
    got = out.cpu().numpy().reshape(b, 2, nn, inner).astype(np.float64)
    ref = np.fft.fft(h[:, 0].astype(np.float64) + 1j * h[:, 1].astype(np.float64), axis=1) / nn
    err = np.linalg.norm((got[:, 0] + 1j * got[:, 1]) - ref) / np.linalg.norm(ref)
    assert err <= REL_L2_TOL, err
    plan.close()
