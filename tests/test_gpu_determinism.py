"""Run-to-run determinism: every transform is a fixed sequence of fp operations, so repeated executions must be
bit-identical. This is a regression guard for an intermittent wrong twiddle product that packed fp32 VALU
sequences (v_pk_*_f32 emitted by the SLP vectoriser) produced next to MFMAs on gfx950: it showed up as a few
outliers at one output index of the radix-256 column pass, invisible to rel-L2 checks. Needs an MI355X: `-m gpu`."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,batch,inner", [(4096, 2048, 1), (1 << 13, 512, 1), (1 << 16, 128, 1), (1 << 20, 8, 1),
                                            (256, 16, 1024), (4096, 2, 2048),
                                            (256, 40000, 1), (512, 9000, 1), (1024, 7000, 1), (2048, 3000, 1),   # single-pass kernels
                                            (1 << 13, 515, 1), (1 << 14, 255, 1),    # narrow column pass + ragged remainder, radix-32/64 tails
                                            (1 << 15, 64, 1), (1 << 21, 4, 1), (1 << 17, 32, 1)])
def test_repeated_runs_are_bit_identical(n, batch, inner):
    _repeat(n, batch, inner, 0)


@pytest.mark.parametrize("n,batch", [(1 << 13, 515), (1 << 14, 255), (1 << 15, 64)])
def test_repeated_runs_of_the_column_plan_are_bit_identical(n, batch):
    """2^13..2^15 default to the single-pass kernel; variant bit 16777216 selects the column-pass plan they used before."""
    _repeat(n, batch, 1, 16777216)


@pytest.mark.parametrize("n,batch", [(1 << 16, 64), (1 << 20, 8), (1 << 21, 4), (1 << 24, 1)])
def test_repeated_runs_of_the_transposed_order_plan_are_bit_identical(n, batch):
    """Column pass with the four-step twiddle (v_sin / v_cos recurrences) + grouped single-kernel pass."""
    _repeat(n, batch, 1, 0, output_order="transposed")


def test_repeated_runs_of_the_fused_2d_plan_are_bit_identical():
    import torch
    import __graft_entry__ as g

    g.build()
    import tensor_fft_amd as tf

    n, batch = 4096, 4
    gen = torch.Generator(device="cuda").manual_seed(11)
    re = (torch.rand(batch * n * n, device="cuda", generator=gen) * 2 - 1).half()
    im = (torch.rand(batch * n * n, device="cuda", generator=gen) * 2 - 1).half()
    plan = tf.TfftPlan2D(n, n, batch, 0)
    ref = None
    for _ in range(10):
        o_re, o_im = torch.zeros_like(re), torch.zeros_like(im)
        plan.exec(re, im, o_re, o_im)
        torch.cuda.synchronize()
        if ref is None:
            ref = (o_re, o_im)
        else:
            assert bool((o_re == ref[0]).all()) and bool((o_im == ref[1]).all())


def _repeat(n, batch, inner, variant, **plan_kw):
    import torch
    import __graft_entry__ as g

    g.build()
    import tensor_fft_amd as tf

    gen = torch.Generator(device="cuda").manual_seed(n + inner)
    x = (torch.rand(batch * 2 * n * inner, device="cuda", generator=gen) * 2 - 1).half()
    plan = tf.TfftPlan(n, batch, 0, inner=inner, preserve_input=True, variant=variant, **plan_kw)
    ref = None
    for _ in range(25):
        y = torch.zeros_like(x)
        plan.exec(x, x[n * inner:], y, y[n * inner:])
        torch.cuda.synchronize()
        if ref is None:
            ref = y
        else:
            assert bool((y == ref).all()), f"{int((y != ref).sum())} elements differ between two runs"


def test_tone_signal_has_no_outliers(orc):
    """The signal that exposed the problem: 16384 integer-frequency tones at N = 2^20 (AccuracyTestBandwidth.cu
    protocol). Max |delta| must stay at the few-fp16-ulp level, not just the rel-L2 norm."""
    import torch
    import tensor_fft_amd as tf

    n = 1 << 20
    re, im = orc.sine_superposition(n, orc.random_weights(16384, 42), orc.random_weights(16384, 1764), 16384)
    ex_re, ex_im = orc.dft64(re, im)
    dev = torch.from_numpy(np.concatenate([re, im])).cuda()
    out = torch.empty_like(dev)
    plan = tf.TfftPlan(n, 1, 0, preserve_input=True)
    for _ in range(5):
        plan.exec(dev, dev[n:], out, out[n:])
        torch.cuda.synchronize()
        o = out.cpu().numpy().astype(np.float64)
        err = max(np.abs(o[:n] - ex_re[0]).max(), np.abs(o[n:] - ex_im[0]).max())
        assert err < 1.5e-3, err              # lines of magnitude 0.5: 3 fp16 ulps
