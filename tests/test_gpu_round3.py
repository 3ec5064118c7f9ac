"""Round 3: launch shapes as plan options (tfft_plan_opts.launch_iters, the tuner's per-(N, batch) knob)."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
REL_L2_TOL = 1.5e-3


@pytest.fixture(scope="module")
def tf():
    import __graft_entry__ as g

    g.build()
    import tensor_fft_amd as t

    t.device_check(0)
    return t


def _cases():
    """Every kernel family at a batch that fills the chip several times over and at a small one, every launch_iters value
    tools/tuner.py can emit (plus an odd one)."""
    import tuner

    out = []
    for n, batches in ((256, (5, 20000)), (1024, (3, 9000)), (4096, (7, 5000)), (8192, (2, 700)), (1 << 15, (1, 130)),
                       (1 << 16, (1, 70)), (1 << 18, (1, 33)), (1 << 20, (1, 17)), (1 << 22, (1, 3))):
        for b in batches:
            for it in sorted(set(tuner.iters_candidates()[1:] + [3])):
                out.append((n, b, it))
    return out


@pytest.mark.parametrize("n,batch,iters", _cases())
def test_launch_shape_never_changes_results(tf, orc, n, batch, iters):
    """launch_iters re-shapes the grid only: spectra are bit-identical to the default shape's, whose first and last transforms
    are checked against the CPU oracle."""
    import torch

    x = torch.empty(batch * 2 * n, dtype=torch.float16, device="cuda")
    tf.synth_uniform(x, x[n:], n, batch, seed=n + batch)
    outs = []
    for it in (0, iters):
        y = torch.full_like(x, float("nan"))
        plan = tf.TfftPlan(n, batch, 0, preserve_input=True, launch_iters=it)
        plan.exec(x, x[n:], y, y[n:])
        torch.cuda.synchronize()
        outs.append(y)
    assert bool((outs[0].view(torch.int16) == outs[1].view(torch.int16)).all()), (n, batch, iters)
    for b in sorted({0, batch - 1}):
        re, im = orc.synth_uniform(n, 1, b, n + batch)
        e_re, e_im = orc.dft64(re, im)
        o = outs[0][b * 2 * n:(b + 1) * 2 * n].cpu().numpy().astype(np.float64)
        got, exact = o[:n] + 1j * o[n:], e_re[0] + 1j * e_im[0]
        assert np.linalg.norm(got - exact) / np.linalg.norm(exact) <= REL_L2_TOL


def test_launch_shape_reaches_strided_axes_and_transposed_plans(tf):
    import torch

    rng = np.random.default_rng(9)
    for kw, n, batch in (({"inner": 256}, 512, 3), ({"output_order": "transposed"}, 1 << 18, 5), ({"inner": 64}, 4096, 2)):
        nf = n * kw.get("inner", 1)
        x = torch.from_numpy(rng.uniform(-1, 1, batch * 2 * nf).astype(np.float16)).cuda()
        want = None
        for it in (0, 1, 2, 65535):
            y = torch.full_like(x, float("nan"))
            tf.TfftPlan(n, batch, 0, preserve_input=True, launch_iters=it, **kw).exec(x, x[nf:], y, y[nf:])
            torch.cuda.synchronize()
            if want is None:
                want = y
            assert bool((y.view(torch.int16) == want.view(torch.int16)).all()), (kw, it)


def test_reference_api_uses_the_tuner_line_of_the_nearest_batch(tf, tmp_path):
    """CreatePlan(N, tuner_file) + ComputeFFT(plan, DataBatchHandler): columns 7 / 8 of the file reach tfft_plan_opts."""
    f = tmp_path / "TunerResults.dat"
    f.write_text("4096 4096 16 1 256 16 65535 1\n4096 4096 16 1 256 10 1 64\n4096 4096 16 1 256 10 2 65536\n")
    plan = tf.CreatePlan(4096, str(f))
    rng = np.random.default_rng(4)
    for batch in (1, 50, 3000):
        host = rng.uniform(-1, 1, batch * 2 * 4096).astype(np.float16)
        h = tf.DataBatchHandler(4096, batch)
        assert h.CopyDataHostToDevice(host) is None
        assert tf.ComputeFFT(plan, h) is None
        out = np.empty_like(host)
        assert h.CopyResultsDeviceToHost(out, plan.results_in_results_) is None
        z = host.reshape(batch, 2, 4096).astype(np.float64)
        exact = np.fft.fft(z[:, 0] + 1j * z[:, 1], axis=1) / 4096
        o = out.reshape(batch, 2, 4096).astype(np.float64)
        got = o[:, 0] + 1j * o[:, 1]
        assert np.linalg.norm(got - exact) / np.linalg.norm(exact) <= REL_L2_TOL


@pytest.mark.parametrize("inner,batch", [(64, 3), (128, 2), (256, 5), (4096, 2)])
@pytest.mark.parametrize("extra", [0, 524288])
def test_two_round_radix512_kernel_strided_axis(tf, inner, batch, extra):
    """colfft512r.hpp (variant bit 268435456): 128-column tiles / 8 waves where the geometry allows, 64-column tiles / 4 waves
    otherwise or with bit 524288; against numpy's fp64 FFT and within an ulp or so of the 8-wave single-round kernel."""
    import torch

    n = 512
    rng = np.random.default_rng(inner + batch)
    re = rng.uniform(-1, 1, (batch, n, inner)).astype(np.float16)
    im = rng.uniform(-1, 1, (batch, n, inner)).astype(np.float16)
    dev = torch.from_numpy(np.ascontiguousarray(np.stack([re, im], axis=1))).cuda().reshape(-1)
    exact = np.fft.fft(re.astype(np.float64) + 1j * im.astype(np.float64), axis=1) / n
    got = {}
    for v in (67108864, 67108864 | 268435456 | extra):
        for scale in ("sequential", "once"):
            out = torch.full_like(dev, float("nan"))
            plan = tf.TfftPlan(n, batch, 0, inner=inner, variant=v, scale=scale)
            assert plan.num_launches == 1
            plan.exec(dev, dev[n * inner:], out, out[n * inner:])
            torch.cuda.synchronize()
            o = out.cpu().numpy().reshape(batch, 2, n, inner).astype(np.float64)
            z = o[:, 0] + 1j * o[:, 1]
            assert np.linalg.norm(z - exact) / np.linalg.norm(exact) <= REL_L2_TOL, (v, scale)
            got[(v, scale)] = z
    a, b = got[(67108864, "sequential")], got[(67108864 | 268435456 | extra, "sequential")]
    assert np.abs(a - b).max() <= 2.0 ** -10 * np.abs(exact).max()       # one rounding apart, not two algorithms apart


@pytest.mark.parametrize("variant", [268435456, 268435456 | 524288])
def test_two_round_radix512_kernel_as_last_pass_of_2pow18(tf, orc, variant):
    import torch

    n, batch = 1 << 18, 6
    assert tf.plan_describe(n) == "col:512+tw col:512"
    x = torch.empty(batch * 2 * n, dtype=torch.float16, device="cuda")
    tf.synth_uniform(x, x[n:], n, batch, seed=18)
    for scale in ("sequential", "none", "once"):
        y = torch.full_like(x, float("nan"))
        xin = x if scale != "none" else (x * (1.0 / 64)).half()
        tf.TfftPlan(n, batch, 0, preserve_input=True, variant=variant, scale=scale).exec(xin, xin[n:], y, y[n:])
        torch.cuda.synchronize()
        for b in (0, batch - 1):
            re, im = orc.synth_uniform(n, 1, b, 18)
            if scale == "none":
                re, im = (re * np.float16(1.0 / 64)).astype(np.float16), (im * np.float16(1.0 / 64)).astype(np.float16)
            e_re, e_im = orc.dft64(re, im)
            exact = (e_re[0] + 1j * e_im[0]) * (n if scale == "none" else 1)
            o = y[b * 2 * n:(b + 1) * 2 * n].cpu().numpy().astype(np.float64)
            got = o[:n] + 1j * o[n:]
            assert np.linalg.norm(got - exact) / np.linalg.norm(exact) <= REL_L2_TOL, (variant, scale, b)


def test_cxx_unit_test_of_the_reference_protocol(tf):
    """examples/unit_test.cpp: the reference's UnitTest.cu (2^8 .. 2^20, 10 signals per length, 20 harmonics, thresholds
    1e-3 / 1e-2 / 0.5) written against include/tensor_fft.hpp, comparison data from hipFFT Z2Z as the reference's comes
    from cuFFT Z2Z."""
    import subprocess

    exe = os.path.join(ROOT, "examples", "unit_test")
    r = subprocess.run(["timeout", "-k", "10", "600", exe, "20"], capture_output=True, text=True)
    print(r.stdout[-3000:])
    assert r.returncode == 0 and "All tests passed!" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.count("Testing fft_length") == 13
