"""Parity of the HIP path (through the C ABI) against the CPU oracle. Needs an MI355X: `-m gpu`.

Stated tolerance (fp16 storage, fp16 MFMA operands, fp32 accumulation), against the fp64 DFT(x)/N:
    rel-L2 error <= 1.5e-3  and  max|delta| <= 1.5 x the max|delta| of the oracle's fp16 restatement
    of the reference kernels on the same input (+ one fp16 ulp of the largest output),
and the reference's own acceptance thresholds (UnitTest.cu:14-16) on the reference's test signal.
Bit-exactness with the CUDA kernels is not defined: they accumulate in fp16 inside HMMA, gfx950 MFMA
accumulates in fp32.
"""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REL_L2_TOL = 1.5e-3
AVG_THR, SIGMA_THR, MAX_THR = 1e-3, 1e-2, 0.5      # UnitTest.cu:14-16


@pytest.fixture(scope="module")
def tf():
    import __graft_entry__ as g

    g.build()
    import torch

    assert torch.cuda.is_available()
    import tensor_fft_amd as t

    t.device_check(0)
    return t


@pytest.fixture(scope="module")
def torch():
    import torch as t

    return t


def _c(re, im):
    return np.asarray(re, dtype=np.float64) + 1j * np.asarray(im, dtype=np.float64)


def _run(tf, torch, re, im, **plan_kw):
    """re/im: float16 (batch, n) on the host -> float16 (batch, n) pair via tfft_exec, block layout."""
    batch, n = re.shape
    host = np.stack([re, im], axis=1)                  # (batch, 2, n) == [fft_i RE | fft_i IM]
    dev = torch.from_numpy(np.ascontiguousarray(host)).cuda().reshape(-1)
    out = torch.full_like(dev, float("nan"))
    plan = tf.TfftPlan(n, batch, 0, **plan_kw)
    plan.exec(dev, dev[n:], out, out[n:])
    torch.cuda.synchronize()
    o = out.cpu().numpy().reshape(batch, 2, n)
    return o[:, 0], o[:, 1]


def _check_against_oracle(orc, re, im, got_re, got_im, mode):
    exact = _c(*orc.dft64(re, im))
    got = _c(got_re, got_im)
    assert np.isfinite(got).all()
    rel = np.linalg.norm(got - exact) / np.linalg.norm(exact)
    assert rel <= REL_L2_TOL, rel
    ref = _c(*orc.ref_fft(re, im, mode))
    err_got = max(np.abs((got - exact).real).max(), np.abs((got - exact).imag).max())
    err_ref = max(np.abs((ref - exact).real).max(), np.abs((ref - exact).imag).max())
    ulp = 2.0 ** (np.floor(np.log2(np.abs(exact).max())) - 10)
    assert err_got <= 1.5 * err_ref + ulp, (err_got, err_ref)
    return rel


def test_instruction_semantics_probe():
    exe = os.path.join(ROOT, "tools", "probe_gfx950")
    if not os.path.exists(exe):
        subprocess.check_call(["hipcc", "-O2", "--offload-arch=gfx950", "-o", exe, exe + ".hip"])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    print(r.stdout)
    assert r.returncode == 0, r.stdout + r.stderr


@pytest.mark.parametrize("batch", [1, 8, 37, 300])
def test_n4096_random_uniform(tf, torch, orc, batch):
    rng = np.random.default_rng(100 + batch)
    re = rng.uniform(-1, 1, (batch, 4096)).astype(np.float16)
    im = rng.uniform(-1, 1, (batch, 4096)).astype(np.float16)
    g = _run(tf, torch, re, im)
    _check_against_oracle(orc, re, im, *g, mode=orc.MODE_4096)


@pytest.mark.parametrize("batch", [1, 16, 37, 1000])
def test_n256_dedicated_kernel(tf, torch, orc, batch):
    """N = 256 (the reference's TensorFFT256 base case): 16 transforms per wave, ragged last group."""
    rng = np.random.default_rng(256 + batch)
    re = rng.uniform(-1, 1, (batch, 256)).astype(np.float16)
    im = rng.uniform(-1, 1, (batch, 256)).astype(np.float16)
    plan = tf.TfftPlan(256, batch, 0)
    assert plan.kernel_name == "fft256_kernel" and plan.num_launches == 1
    gr, gi = _run(tf, torch, re, im)
    _check_against_oracle(orc, re, im, gr, gi, mode=orc.MODE_256)
    # in place and fully planar strides give the same bits
    dev = torch.from_numpy(np.ascontiguousarray(np.stack([re, im], axis=1))).cuda().reshape(-1)
    plan.exec(dev, dev[256:], dev, dev[256:])
    torch.cuda.synchronize()
    o = dev.cpu().numpy().reshape(batch, 2, 256)
    assert np.array_equal(o[:, 0].view(np.uint16), gr.view(np.uint16)) and np.array_equal(o[:, 1].view(np.uint16), gi.view(np.uint16))
    d_re, d_im = torch.from_numpy(re).cuda().reshape(-1), torch.from_numpy(im).cuda().reshape(-1)
    o_re, o_im = torch.empty_like(d_re), torch.empty_like(d_im)
    tf.TfftPlan(256, batch, 0, in_batch_stride=256, out_batch_stride=256).exec(d_re, d_im, o_re, o_im)
    torch.cuda.synchronize()
    assert np.array_equal(o_re.cpu().numpy().view(np.uint16).reshape(batch, 256), gr.view(np.uint16))


def test_n4096_known_answers(tf, torch):
    n = 4096
    z = np.zeros((4, n), dtype=np.float16)
    re = z.copy()
    im = z.copy()
    re[0, 0] = 1.0                                   # impulse -> 1/N everywhere
    re[1, :] = 1.0                                   # constant -> X[0] = 1
    t = 2 * np.pi * 5 * np.arange(n) / n
    re[2], im[2] = np.cos(t), np.sin(t)              # exp(+i 2 pi 5 n/N) -> X[5] = 1
    re[3, 1] = 1.0                                   # shifted impulse -> exp(-2 pi i k/N)/N
    gr, gi = _run(tf, torch, re, im)
    np.testing.assert_allclose(gr[0].astype(np.float64), 1.0 / n, rtol=2e-3)
    assert np.abs(gi[0].astype(np.float64)).max() < 1e-6
    assert abs(float(gr[1, 0]) - 1.0) < 2e-3 and np.abs(_c(gr[1], gi[1])[1:]).max() < 2e-3
    assert abs(float(gr[2, 5]) - 1.0) < 3e-3
    m = np.abs(_c(gr[2], gi[2]))
    m[5] = 0
    assert m.max() < 2e-3
    k = np.arange(n)
    np.testing.assert_allclose(_c(gr[3], gi[3]), np.exp(-2j * np.pi * k / n) / n, atol=2e-6)


def test_n4096_reference_thresholds_on_reference_signal(tf, torch, orc):
    """UnitTest.cu:8-24: 20 harmonics, seeds 42*i / 42*42*i, thresholds on max/avg/sigma of |delta|."""
    n = 4096
    sigs = [orc.sine_superposition(n, orc.random_weights(20, 42 * i), orc.random_weights(20, 42 * 42 * i), 20)
            for i in range(10)]
    re = np.stack([s[0] for s in sigs])
    im = np.stack([s[1] for s in sigs])
    gr, gi = _run(tf, torch, re, im)
    ex_re, ex_im = orc.dft64(re, im)
    for i in range(10):
        mx, avg, sig = orc.deviation_stats(gr[i].astype(np.float64), gi[i].astype(np.float64), ex_re[i], ex_im[i])
        assert mx <= MAX_THR and avg <= AVG_THR and sig <= SIGMA_THR
        assert mx < 1e-3 and avg < 5e-5
    _check_against_oracle(orc, re, im, gr, gi, mode=orc.MODE_4096)


def test_n4096_golden_benchmark_signal(tf, torch, orc, golden_dir):
    g = np.load(os.path.join(golden_dir, "bench_signal.npz"))
    re = g["in_re_4096"].view(np.float16)[None, :]
    im = g["in_im_4096"].view(np.float16)[None, :]
    gr, gi = _run(tf, torch, re, im)
    ref = _c(g["ref_re_4096_mode1"].view(np.float16), g["ref_im_4096_mode1"].view(np.float16))
    # both are within a few fp16 ulps of DFT/N, hence of each other
    assert np.abs(_c(gr[0], gi[0]) - ref).max() < 1e-3
    _check_against_oracle(orc, re, im, gr, gi, mode=orc.MODE_4096)


def test_n4096_large_and_tiny_magnitudes(tf, torch, orc):
    """Per-stage 1/16 scaling keeps |intermediates| <= max|x|: no overflow at fp16 max, and small inputs
    are not flushed the way an up-front x/4096 (TensorFFT4096.cu:169-173) pushes them into subnormals."""
    rng = np.random.default_rng(9)
    base = rng.uniform(-1, 1, (2, 2, 4096))
    re = np.stack([base[0, 0] * 60000, base[1, 0] * 2e-3]).astype(np.float16)
    im = np.stack([base[0, 1] * 60000, base[1, 1] * 2e-3]).astype(np.float16)
    gr, gi = _run(tf, torch, re, im)
    exact = _c(*orc.dft64(re, im))
    got = _c(gr, gi)
    assert np.isfinite(got).all()
    assert np.linalg.norm(got[0] - exact[0]) / np.linalg.norm(exact[0]) < REL_L2_TOL
    # row 1: outputs ~3e-5 are fp16-subnormal (quantum 6e-8): absolute bound
    assert np.abs(got[1] - exact[1]).max() < 2 * 2.0 ** -24 + 1e-3 * np.abs(exact[1]).max()


@pytest.mark.parametrize("n", [256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536])
def test_full_scale_inputs_do_not_overflow(tf, torch, orc, n):
    """Every kernel family scales by its radix per stage, so |intermediates| <= max|x|: full-scale constant and
    alternating inputs (fp16 max = 65504) give the exact DC / Nyquist line, full-scale noise stays finite."""
    rng = np.random.default_rng(n)
    re = np.empty((3, n), np.float16)
    im = np.empty((3, n), np.float16)
    re[0], im[0] = 65504.0, -65504.0
    re[1] = 65504.0 * (1 - 2 * (np.arange(n) & 1))
    im[1] = 0.0
    re[2] = (rng.uniform(-1, 1, n) * 65504).astype(np.float16)
    im[2] = (rng.uniform(-1, 1, n) * 65504).astype(np.float16)
    gr, gi = _run(tf, torch, re, im)
    got = _c(gr, gi)
    assert np.isfinite(got).all()
    want0 = np.zeros(n, complex); want0[0] = 65504.0 - 65504.0j
    want1 = np.zeros(n, complex); want1[n // 2] = 65504.0
    assert np.abs(got[0] - want0).max() <= 65504 * 2.0 ** -10
    assert np.abs(got[1] - want1).max() <= 65504 * 2.0 ** -10
    exact = _c(*orc.dft64(re[2:], im[2:]))
    assert np.linalg.norm(got[2] - exact[0]) / np.linalg.norm(exact[0]) < REL_L2_TOL


@pytest.mark.parametrize("n", [256, 512, 2048, 8192, 32768, 65536])
def test_tiny_magnitudes(tf, torch, orc, n):
    """Inputs of amplitude 2e-3: outputs land in binary16's subnormal range (quantum 6e-8); no stage may flush them
    (the reference's up-front x / N does: TensorFFT4096.cu:169-173). Absolute bound: two subnormal quanta + 1e-3 relative."""
    rng = np.random.default_rng(n + 5)
    re = (rng.uniform(-1, 1, (2, n)) * 2e-3).astype(np.float16)
    im = (rng.uniform(-1, 1, (2, n)) * 2e-3).astype(np.float16)
    gr, gi = _run(tf, torch, re, im)
    exact = _c(*orc.dft64(re, im))
    got = _c(gr, gi)
    assert np.isfinite(got).all()
    assert np.abs(got - exact).max() < 2 * 2.0 ** -24 + 1e-3 * np.abs(exact).max()
    assert np.linalg.norm(got) > 0.5 * np.linalg.norm(exact)          # not flushed to zero


def test_n4096_in_place_and_strides(tf, torch, orc):
    n, batch = 4096, 19
    rng = np.random.default_rng(21)
    re = rng.uniform(-1, 1, (batch, n)).astype(np.float16)
    im = rng.uniform(-1, 1, (batch, n)).astype(np.float16)
    want = _run(tf, torch, re, im)
    # (a) in place on the [RE|IM] block
    dev = torch.from_numpy(np.ascontiguousarray(np.stack([re, im], axis=1))).cuda().reshape(-1)
    plan = tf.TfftPlan(n, batch, 0)
    plan.exec(dev, dev[n:], dev, dev[n:])
    torch.cuda.synchronize()
    o = dev.cpu().numpy().reshape(batch, 2, n)
    assert np.array_equal(o[:, 0].view(np.uint16), want[0].view(np.uint16))
    assert np.array_equal(o[:, 1].view(np.uint16), want[1].view(np.uint16))
    # (b) fully planar: all RE planes, then all IM planes (stride N), padded output stride
    d_re, d_im = torch.from_numpy(re).cuda(), torch.from_numpy(im).cuda()
    o_re = torch.zeros(batch, n + 64, dtype=torch.float16, device="cuda")
    o_im = torch.zeros_like(o_re)
    plan2 = tf.TfftPlan(n, batch, 0, in_batch_stride=n, out_batch_stride=n + 64)
    plan2.exec(d_re.reshape(-1), d_im.reshape(-1), o_re.reshape(-1), o_im.reshape(-1))
    torch.cuda.synchronize()
    assert np.array_equal(o_re[:, :n].cpu().numpy().view(np.uint16), want[0].view(np.uint16))
    assert np.array_equal(o_im[:, :n].cpu().numpy().view(np.uint16), want[1].view(np.uint16))
    assert float(o_re[:, n:].abs().max()) == 0.0          # padding untouched


def test_n4096_full_size_config(tf, torch, orc):
    """BASELINE config 2 at full size (batch 65536, 1 GiB in + 1 GiB out): size-independent properties.
    (1) every wave of every workgroup computes the same thing: one signal replicated over the batch gives
    bit-identical spectra; (2) Parseval per FFT; (3) linearity spot checks against the oracle on sampled FFTs."""
    n, batch = 4096, 65536
    gen = torch.Generator(device="cuda").manual_seed(42)
    x = (torch.rand(batch, 2, n, device="cuda", generator=gen) * 2 - 1).to(torch.float16)
    x[1::2] = x[0]                                       # odd FFTs: replicas of FFT 0
    flat = x.reshape(-1)
    out = torch.empty_like(flat)
    plan = tf.TfftPlan(n, batch, 0)
    plan.exec(flat, flat[n:], out, out[n:])
    torch.cuda.synchronize()
    y = out.reshape(batch, 2, n)
    assert bool((y[1::2] == y[1]).all())
    e_in = (x.float() ** 2).sum(dim=(1, 2)) / n
    e_out = (y.float() ** 2).sum(dim=(1, 2))
    assert float(((e_out - e_in).abs() / e_in).max()) < 5e-3
    idx = [0, 2, 4094, 32768, 65534, 12346]
    re = x[idx, 0].cpu().numpy()
    im = x[idx, 1].cpu().numpy()
    g = y[idx].cpu().numpy()
    _check_against_oracle(orc, re, im, g[:, 0], g[:, 1], mode=orc.MODE_4096)


@pytest.mark.parametrize("n", [512, 1024, 2048])
@pytest.mark.parametrize("batch", [1, 3, 9, 130, 1000])
def test_n256r_single_pass_kernel(tf, torch, orc, n, batch):
    """N = 512 / 1024 / 2048 (the reference's TensorFFT256 + radix-2 steps) run as ONE kernel: 16 * 256 / N transforms
    per wave, ragged last group, strides, in place; an impulse at every position exercises each (n0, n1, r) path."""
    rng = np.random.default_rng(n + batch)
    re = rng.uniform(-1, 1, (batch, n)).astype(np.float16)
    im = rng.uniform(-1, 1, (batch, n)).astype(np.float16)
    plan = tf.TfftPlan(n, batch, 0)
    assert plan.kernel_name == "fft256r_kernel" and plan.num_launches == 1 and plan.workspace_bytes == 0
    gr, gi = _run(tf, torch, re, im)
    _check_against_oracle(orc, re, im, gr, gi, mode=orc.MODE_256)
    dev = torch.from_numpy(np.ascontiguousarray(np.stack([re, im], axis=1))).cuda().reshape(-1)
    plan.exec(dev, dev[n:], dev, dev[n:])
    torch.cuda.synchronize()
    o = dev.cpu().numpy().reshape(batch, 2, n)
    assert np.array_equal(o[:, 0].view(np.uint16), gr.view(np.uint16)) and np.array_equal(o[:, 1].view(np.uint16), gi.view(np.uint16))
    d_re, d_im = torch.from_numpy(re).cuda().reshape(-1), torch.from_numpy(im).cuda().reshape(-1)
    o_re, o_im = torch.empty_like(d_re), torch.empty_like(d_im)
    tf.TfftPlan(n, batch, 0, in_batch_stride=n, out_batch_stride=n).exec(d_re, d_im, o_re, o_im)
    torch.cuda.synchronize()
    assert np.array_equal(o_re.cpu().numpy().reshape(batch, n).view(np.uint16), gr.view(np.uint16))
    assert np.array_equal(o_im.cpu().numpy().reshape(batch, n).view(np.uint16), gi.view(np.uint16))


@pytest.mark.parametrize("n", [512, 1024, 2048])
def test_n256r_impulse_at_every_position(tf, torch, n):
    """x = N * delta[n - p] for every p: X[k] = exp(-2 pi i p k / N) exactly representable inputs, every sample path."""
    re = (np.eye(n, dtype=np.float64) * 1024.0).astype(np.float16)      # batch = n transforms, impulse p in transform p
    im = np.zeros_like(re)
    gr, gi = _run(tf, torch, re, im)
    k = np.arange(n)
    want = np.exp(-2j * np.pi * np.outer(k, k) / n) * (1024.0 / n)
    got = gr.astype(np.float64) + 1j * gi.astype(np.float64)
    assert np.abs(got - want).max() <= 2.5e-3 * (1024.0 / n) + 2.0 ** -11 * (1024.0 / n)


@pytest.mark.parametrize("lg", [1, 2, 3, 4, 5, 8, 9, 10, 11, 13, 14, 15, 16])
def test_generic_lengths(tf, torch, orc, lg):
    n = 1 << lg
    batch = 5
    rng = np.random.default_rng(lg)
    re = rng.uniform(-1, 1, (batch, n)).astype(np.float16)
    im = rng.uniform(-1, 1, (batch, n)).astype(np.float16)
    gr, gi = _run(tf, torch, re, im)
    exact = _c(*orc.dft64(re, im))
    got = _c(gr, gi)
    assert np.linalg.norm(got - exact) / np.linalg.norm(exact) <= REL_L2_TOL
    if lg >= 8:
        _check_against_oracle(orc, re, im, gr, gi, mode=orc.MODE_256)
    # preserve_input and in-place variants give the same bits
    gr2, gi2 = _run(tf, torch, re, im, preserve_input=True)
    assert np.array_equal(gr2.view(np.uint16), gr.view(np.uint16)) and np.array_equal(gi2.view(np.uint16), gi.view(np.uint16))
    dev = torch.from_numpy(np.ascontiguousarray(np.stack([re, im], axis=1))).cuda().reshape(-1)
    keep = dev.clone()
    out = torch.empty_like(dev)
    tf.TfftPlan(n, batch, 0, preserve_input=True).exec(dev, dev[n:], out, out[n:])
    torch.cuda.synchronize()
    assert bool((dev == keep).all())
    tf.TfftPlan(n, batch, 0).exec(dev, dev[n:], dev, dev[n:])
    torch.cuda.synchronize()
    o = dev.cpu().numpy().reshape(batch, 2, n)
    assert np.array_equal(o[:, 0].view(np.uint16), gr.view(np.uint16))


@pytest.mark.parametrize("n,batch", [(8192, 4), (8192, 5), (8192, 7), (8192, 64), (16384, 2), (16384, 3), (16384, 9)])
def test_narrow_pitch_cooperative_column_pass(tf, torch, orc, n, batch):
    """N = 2^13 / 2^14: the radix-256 column pass has only 32 / 64 columns; a workgroup takes 4 / 2 whole
    transforms (contiguous rows); a ragged remainder goes to a second launch. Must equal the per-wave kernel (variant bit 131072)
    bit for bit, and the oracle within tolerance."""
    rng = np.random.default_rng(n + batch)
    re = rng.uniform(-1, 1, (batch, n)).astype(np.float16)
    im = rng.uniform(-1, 1, (batch, n)).astype(np.float16)
    # the column plan (default is the single-pass kernel), on the throughput kernels (1073741824: a small batch of 2^14 would take
    # the latency column kernel since round 5, whose hardware sin / cos twiddles round differently)
    gr, gi = _run(tf, torch, re, im, variant=16777216 | 1073741824)
    pr, pi = _run(tf, torch, re, im, variant=16777216 | 1073741824 | 131072)
    assert np.array_equal(gr.view(np.uint16), pr.view(np.uint16)) and np.array_equal(gi.view(np.uint16), pi.view(np.uint16))
    _check_against_oracle(orc, re, im, gr, gi, mode=orc.MODE_256)


@pytest.mark.parametrize("n", [8192, 16384, 32768])
@pytest.mark.parametrize("batch", [1, 3, 4, 9, 130])
def test_n4096r_single_pass_kernel(tf, torch, orc, n, batch):
    """N = 8192 / 16384 / 32768 as ONE kernel (R = N / 4096 waves share a transform: radix-R front end in fp32, the
    4096 kernel's three MFMA stages, interleaved read-out); ragged last workgroup, in place, strides."""
    rng = np.random.default_rng(n + batch)
    re = rng.uniform(-1, 1, (batch, n)).astype(np.float16)
    im = rng.uniform(-1, 1, (batch, n)).astype(np.float16)
    # (round 5: up to 8 transforms of 2^15 default to the two-launch plan 256 x 128, tfft_plan_default_variant; a caller that names
    # any variant bit gets exactly that variant, and 1073741824 alone changes nothing about a single-pass plan)
    kw = {"variant": 1073741824} if tf.plan_default_variant(n, 1, batch) else {}
    plan = tf.TfftPlan(n, batch, 0, **kw)
    assert plan.kernel_name == "fft4096r_kernel" and plan.num_launches == 1 and plan.workspace_bytes == 0
    gr, gi = _run(tf, torch, re, im, **kw)
    _check_against_oracle(orc, re, im, gr, gi, mode=orc.MODE_4096)
    dev = torch.from_numpy(np.ascontiguousarray(np.stack([re, im], axis=1))).cuda().reshape(-1)
    plan.exec(dev, dev[n:], dev, dev[n:])
    torch.cuda.synchronize()
    o = dev.cpu().numpy().reshape(batch, 2, n)
    assert np.array_equal(o[:, 0].view(np.uint16), gr.view(np.uint16)) and np.array_equal(o[:, 1].view(np.uint16), gi.view(np.uint16))
    d_re, d_im = torch.from_numpy(re).cuda().reshape(-1), torch.from_numpy(im).cuda().reshape(-1)
    o_re, o_im = torch.empty_like(d_re), torch.empty_like(d_im)
    tf.TfftPlan(n, batch, 0, in_batch_stride=n, out_batch_stride=n, **kw).exec(d_re, d_im, o_re, o_im)
    torch.cuda.synchronize()
    assert np.array_equal(o_re.cpu().numpy().reshape(batch, n).view(np.uint16), gr.view(np.uint16))
    assert np.array_equal(o_im.cpu().numpy().reshape(batch, n).view(np.uint16), gi.view(np.uint16))


@pytest.mark.parametrize("lg,batch", [(15, 3), (15, 64), (17, 1), (17, 5), (18, 2), (23, 1), (25, 1)])
def test_radix512_column_pass(tf, torch, orc, lg, batch):
    """2^15 = 512 x 64, 2^17 = 256 x 512, 2^18 = 512 x 512, 2^23 = 256 x 512 x 64, 2^25 = 256 x 256 x 512: the radix-512 column pass (two decimated radix-256
    halves + radix-2 combine at read-out, with and without the next pass's twiddles) against the oracle, and
    against the chain without it (variant bit 8388608), which must agree to fp16 rounding."""
    n = 1 << lg
    rng = np.random.default_rng(lg + batch)
    re = rng.uniform(-1, 1, (batch, n)).astype(np.float16)
    im = rng.uniform(-1, 1, (batch, n)).astype(np.float16)
    # 2^13..2^15 default to the single-pass kernel: ask for the column plan; from 2^16 on name a (neutral) cache-policy bit, so that
    # the large-batch split is what runs (a variant-0 plan of a few transforms takes the small-work split, tfft_plan_default_variant)
    multi = 16777216 if lg <= 15 else 536870912
    plan = tf.TfftPlan(n, batch, 0, variant=multi)
    no512 = multi | 8388608 | 33554432           # (without the radix-1024 pass either, which would stand in at 2^18)
    other = tf.TfftPlan(n, batch, 0, variant=no512)
    # (2^15 without the radix-512 pass is 256 x 128 with the cooperative radix-128 pass since round 5: two launches as well)
    assert plan.num_launches == other.num_launches - (0 if lg == 15 else 1)
    gr, gi = _run(tf, torch, re, im, variant=multi)
    exact = _c(*orc.dft64(re, im))
    got = _c(gr, gi)
    assert np.isfinite(got).all()
    assert np.linalg.norm(got - exact) / np.linalg.norm(exact) <= REL_L2_TOL
    pr, pi = _run(tf, torch, re, im, variant=no512)
    ref = _c(pr, pi)
    assert np.abs(got - exact).max() <= 1.5 * np.abs(ref - exact).max() + 2.0 ** -11 * np.abs(exact).max()


@pytest.mark.parametrize("lg,batch", [(19, 1), (19, 3), (20, 1), (20, 5), (28, 1)])
def test_radix1024_column_pass(tf, torch, orc, lg, batch):
    """2^19 = 512 x 1024, 2^20 = 1024 x 1024 (two passes in natural order), 2^28 = 512 x 512 x 1024: the radix-1024 column
    pass (four decimated radix-256 problems in two rounds, first-round results held in registers, radix-4 combine in two
    read-outs; both output forms, with and without the next pass's twiddles) against the oracle's fp64 DFT / N (2^28: against
    hipFFT complex64 on the device), and against the plan without it (variant bit 33554432), to fp16 rounding."""
    n = 1 << lg
    # a few 2^19 / 2^20-point transforms default to the split with more workgroups (tfft_plan_default_variant, round 4): naming
    # the cache-policy bit (536870912 = streaming, 262144 = plain accesses) asks for the large-batch split, the radix-1024 one
    wide = 536870912 if lg <= 20 else 0
    plan = tf.TfftPlan(n, batch, 0, variant=wide)
    other = tf.TfftPlan(n, batch, 0, variant=33554432)
    assert plan.num_launches == other.num_launches - 1
    assert "col:1024" in tf.plan_describe(n, 1, 0) and "col:1024" not in tf.plan_describe(n, 1, 33554432)
    if lg > 24:
        gen = torch.Generator(device="cuda").manual_seed(lg)
        x = (torch.rand(batch, 2, n, device="cuda", generator=gen) * 2 - 1).half()
        flat = x.reshape(-1)
        y = torch.empty_like(flat)
        plan = tf.TfftPlan(n, batch, 0, preserve_input=True)
        plan.exec(flat, flat[n:], y, y[n:])
        want = torch.fft.fft(torch.complex(x[:, 0].float(), x[:, 1].float()), dim=1) / n
        got = y.reshape(batch, 2, n)
        got = torch.complex(got[:, 0].float(), got[:, 1].float())
        rel = float(torch.linalg.vector_norm(got - want) / torch.linalg.vector_norm(want))
        assert rel <= REL_L2_TOL, rel
        assert float((got - want).abs().max()) <= 8 * 2.0 ** -11 * float(want.abs().max())
        return
    rng = np.random.default_rng(lg + batch)
    re = rng.uniform(-1, 1, (batch, n)).astype(np.float16)
    im = rng.uniform(-1, 1, (batch, n)).astype(np.float16)
    gr, gi = _run(tf, torch, re, im, variant=wide)
    exact = _c(*orc.dft64(re, im))
    got = _c(gr, gi)
    assert np.isfinite(got).all()
    assert np.linalg.norm(got - exact) / np.linalg.norm(exact) <= REL_L2_TOL
    qr, qi = _run(tf, torch, re, im, variant=262144)                  # the same kernels with plain accesses: the same bits
    assert np.array_equal(qr.view(np.uint16), gr.view(np.uint16)) and np.array_equal(qi.view(np.uint16), gi.view(np.uint16))
    pr, pi = _run(tf, torch, re, im, variant=33554432)
    ref = _c(pr, pi)
    assert np.abs(got - exact).max() <= 1.5 * np.abs(ref - exact).max() + 2.0 ** -11 * np.abs(exact).max()
    # known answer: a tone at an arbitrary bin, amplitude chosen so that the line is 1.0
    f0 = (5 * n) // 7
    t = np.arange(n)
    tone = np.exp(2j * np.pi * f0 * t / n)
    tr, ti = _run(tf, torch, tone.real.astype(np.float16)[None], tone.imag.astype(np.float16)[None], variant=wide)
    spec = _c(tr, ti)[0]
    assert abs(spec[f0] - 1.0) <= 2e-3
    spec[f0] = 0
    assert np.abs(spec).max() <= 2e-3


@pytest.mark.parametrize("lg", [13, 16, 17])
def test_plain_autosort_chain_still_correct(tf, torch, orc, lg):
    """variant bit 32 forces the radix-2/4/8/16 autosort chain (no radix-256 column kernel)."""
    n = 1 << lg
    rng = np.random.default_rng(300 + lg)
    re = rng.uniform(-1, 1, (3, n)).astype(np.float16)
    im = rng.uniform(-1, 1, (3, n)).astype(np.float16)
    gr, gi = _run(tf, torch, re, im, variant=32)
    _check_against_oracle(orc, re, im, gr, gi, mode=orc.MODE_256)


@pytest.mark.parametrize("n,inner", [(256, 16), (256, 64), (4096, 16), (4096, 128), (8192, 32), (1 << 16, 16), (512, 8), (64, 16)])
def test_transform_along_strided_axis(tf, torch, orc, n, inner):
    """opts.inner = C: data [batch][n][C], C independent columns innermost (2D column pass, distributed local passes)."""
    batch = 2
    rng = np.random.default_rng(n + inner)
    re = rng.uniform(-1, 1, (batch, n, inner)).astype(np.float16)
    im = rng.uniform(-1, 1, (batch, n, inner)).astype(np.float16)
    host = np.stack([re, im], axis=1)                                   # (batch, 2, n, C)
    dev = torch.from_numpy(np.ascontiguousarray(host)).cuda().reshape(-1)
    out = torch.full_like(dev, float("nan"))
    plan = tf.TfftPlan(n, batch, 0, inner=inner)
    plan.exec(dev, dev[n * inner:], out, out[n * inner:])
    torch.cuda.synchronize()
    o = out.cpu().numpy().reshape(batch, 2, n, inner)
    # oracle works on contiguous transforms: move the column axis out
    cre = np.ascontiguousarray(re.transpose(0, 2, 1)).reshape(-1, n)
    cim = np.ascontiguousarray(im.transpose(0, 2, 1)).reshape(-1, n)
    exact = _c(*orc.dft64(cre, cim))
    got = _c(o[:, 0].transpose(0, 2, 1).reshape(-1, n), o[:, 1].transpose(0, 2, 1).reshape(-1, n))
    assert np.isfinite(got).all()
    rel = np.linalg.norm(got - exact) / np.linalg.norm(exact)
    assert rel <= REL_L2_TOL, rel
    assert np.abs(got - exact).max() < 16 * 2.0 ** -11 * np.abs(exact).max()


@pytest.mark.parametrize("rows,cols,batch", [(256, 512, 3), (4096, 256, 1), (512, 4096, 2)])
def test_2d_transform(tf, torch, rows, cols, batch):
    """BASELINE config 4 shape (row FFT, then column FFT along the strided axis), small instances vs numpy fft2."""
    rng = np.random.default_rng(rows + cols)
    re = rng.uniform(-1, 1, (batch, rows, cols)).astype(np.float16)
    im = rng.uniform(-1, 1, (batch, rows, cols)).astype(np.float16)
    d_re, d_im = torch.from_numpy(re).cuda().reshape(-1), torch.from_numpy(im).cuda().reshape(-1)
    o_re, o_im = torch.empty_like(d_re), torch.empty_like(d_im)
    plan = tf.TfftPlan2D(rows, cols, batch, 0)
    plan.exec(d_re, d_im, o_re, o_im)
    torch.cuda.synchronize()
    got = _c(o_re.cpu().numpy(), o_im.cpu().numpy()).reshape(batch, rows, cols)
    exact = np.fft.fft2(_c(re, im), axes=(1, 2)) / (rows * cols)
    rel = np.linalg.norm(got - exact) / np.linalg.norm(exact)
    assert rel <= REL_L2_TOL, rel


def test_2d_4096_square_properties(tf, torch):
    """One 4096 x 4096 image (BASELINE config 4 has 64 of them): impulse -> flat, plane wave -> single bin, Parseval."""
    n = 4096
    re = torch.zeros(3, n, n, dtype=torch.float16, device="cuda")
    im = torch.zeros_like(re)
    re[0, 3, 5] = 4096.0                   # amplitude 2^12 keeps |X| = 2^-12 clear of fp16 subnormals
    yy, xx = torch.meshgrid(torch.arange(n, device="cuda"), torch.arange(n, device="cuda"), indexing="ij")
    ph = 2 * np.pi * ((7 * yy + 11 * xx) % n).double() / n
    re[1], im[1] = torch.cos(ph).half(), torch.sin(ph).half()
    gen = torch.Generator(device="cuda").manual_seed(1)
    re[2] = (torch.rand(n, n, device="cuda", generator=gen) * 2 - 1).half()
    im[2] = (torch.rand(n, n, device="cuda", generator=gen) * 2 - 1).half()
    o_re, o_im = torch.empty_like(re), torch.empty_like(im)
    plan = tf.TfftPlan2D(n, n, 3, 0)
    plan.exec(re.reshape(-1), im.reshape(-1), o_re.reshape(-1), o_im.reshape(-1))
    torch.cuda.synchronize()
    mag0 = torch.sqrt(o_re[0].float() ** 2 + o_im[0].float() ** 2)
    assert float((mag0 * n - 1).abs().max()) < 5e-3              # |X| = 4096 / N^2 everywhere
    assert abs(float(o_re[1, 7, 11]) - 1.0) < 3e-3
    m1 = torch.sqrt(o_re[1].float() ** 2 + o_im[1].float() ** 2)
    m1[7, 11] = 0
    assert float(m1.max()) < 2e-3
    e_in = float((re[2].float() ** 2 + im[2].float() ** 2).sum()) / (n * n)
    e_out = float((o_re[2].float() ** 2 + o_im[2].float() ** 2).sum())
    assert abs(e_out - e_in) / e_in < 5e-3
    # the random image against hipFFT (complex64) on the same fp16 input; this shape runs the fused two-pass plan
    assert plan.num_launches == 2
    want = torch.fft.fft2(torch.complex(re[2].float(), im[2].float())) / (n * n)
    got = torch.complex(o_re[2].float(), o_im[2].float())
    assert float(torch.linalg.vector_norm(got - want) / torch.linalg.vector_norm(want)) <= REL_L2_TOL
    # a full-scale image stays finite (headroom of the fused radix-8 butterfly)
    re[0] = (torch.rand(n, n, device="cuda", generator=gen) * 2 - 1).half() * 65504
    im[0] = (torch.rand(n, n, device="cuda", generator=gen) * 2 - 1).half() * 65504
    plan.exec(re.reshape(-1), im.reshape(-1), o_re.reshape(-1), o_im.reshape(-1))
    torch.cuda.synchronize()
    assert bool(torch.isfinite(o_re[0]).all()) and bool(torch.isfinite(o_im[0]).all())
    want = torch.fft.fft2(torch.complex(re[0].float(), im[0].float())) / (n * n)
    got = torch.complex(o_re[0].float(), o_im[0].float())
    assert float(torch.linalg.vector_norm(got - want) / torch.linalg.vector_norm(want)) <= REL_L2_TOL


def test_n_2pow20(tf, torch, orc):
    n, batch = 1 << 20, 3
    rng = np.random.default_rng(20)
    re = rng.uniform(-1, 1, (batch, n)).astype(np.float16)
    im = rng.uniform(-1, 1, (batch, n)).astype(np.float16)
    gr, gi = _run(tf, torch, re, im)
    exact = _c(*orc.dft64(re, im))
    rel = np.linalg.norm(_c(gr, gi) - exact) / np.linalg.norm(exact)
    assert rel <= REL_L2_TOL, rel


@pytest.mark.parametrize("lg", list(range(8, 21, 2)) + [9, 13])
def test_reference_api_unit_test_protocol(tf, torch, orc, lg):
    """The reference's UnitTest.cu flow through its own interface names, both base modes."""
    n = 1 << lg
    dev_id = torch.cuda.current_device()
    for i in (0, 3):
        w_re, w_im = orc.random_weights(20, 42 * i), orc.random_weights(20, 42 * 42 * i)
        re, im = orc.sine_superposition(n, w_re, w_im, 20)
        exact = orc.dft64(re, im)
        for mode in ((tf.Mode_256,) if n < 4096 else (tf.Mode_256, tf.Mode_4096)):
            plan = tf.CreatePlan(n, mode, 16 if mode == tf.Mode_4096 else 1, 1, 256)
            assert plan is not None and tf.PlanWorksOnDevice(plan, dev_id)
            handler = tf.DataHandler(n)
            assert handler.PeakAtLastError() is None
            data = np.concatenate([re, im])
            assert handler.CopyDataHostToDevice(data) is None
            assert tf.ComputeFFT(plan, handler, tf.GetMaxNoOptInSharedMem(dev_id)) is None
            res = np.empty(2 * n, dtype=np.float16)
            assert handler.CopyResultsDeviceToHost(res, plan.results_in_results_) is None
            torch.cuda.synchronize()
            mx, avg, sig = orc.deviation_stats(res[:n].astype(np.float64), res[n:].astype(np.float64), exact[0][0], exact[1][0])
            assert mx <= MAX_THR and avg <= AVG_THR and sig <= SIGMA_THR, (n, mode, mx, avg, sig)
            assert mx < 1e-3


def test_reference_api_batch_handler(tf, torch, orc):
    """ExampleBatchFFT.cu / FFTBenchBatch.cu: 20 FFTs through DataBatchHandler, results region layout."""
    n, b = 4096, 20
    rng = np.random.default_rng(33)
    host = rng.uniform(-1, 1, (b, 2, n)).astype(np.float16)
    plan = tf.CreatePlan(n, tf.Mode_4096, 16, 8, 256)
    bh = tf.DataBatchHandler(n, b)
    assert bh.PeakAtLastError() is None
    assert bh.CopyDataHostToDevice(host.reshape(-1)) is None
    assert tf.ComputeFFT(plan, bh, 32768) is None
    res = np.empty(b * 2 * n, dtype=np.float16)
    assert bh.CopyResultsDeviceToHost(res, plan.results_in_results_) is None
    res = res.reshape(b, 2, n)
    _check_against_oracle(orc, host[:, 0], host[:, 1], res[:, 0], res[:, 1], mode=orc.MODE_4096)
    # per-FFT pointer views address the same memory as the block (DataHandler.h:105-114)
    assert bh.dptr_results_RE_[3].data_ptr() == bh.dptr_results_RE_[0].data_ptr() + 3 * 2 * n * 2
    assert bh.dptr_input_IM_[0].data_ptr() == bh.dptr_input_RE_[0].data_ptr() + 2 * n


@pytest.mark.parametrize("n", [4096, 1 << 14])
def test_inverse_is_forward_on_swapped_planes(tf, torch, n):
    batch = 4
    rng = np.random.default_rng(n)
    re = rng.uniform(-1, 1, (batch, n)).astype(np.float16)
    im = rng.uniform(-1, 1, (batch, n)).astype(np.float16)
    dev = torch.from_numpy(np.ascontiguousarray(np.stack([re, im], axis=1))).cuda().reshape(-1)
    out = torch.empty_like(dev)
    plan = tf.TfftPlan(n, batch, 0, preserve_input=True)
    plan.exec_inverse(dev, dev[n:], out, out[n:])
    torch.cuda.synchronize()
    o = out.cpu().numpy().reshape(batch, 2, n)
    want = np.fft.ifft(_c(re, im), axis=1)            # numpy's ifft carries the 1/N
    got = _c(o[:, 0], o[:, 1])
    assert np.linalg.norm(got - want) / np.linalg.norm(want) <= REL_L2_TOL
    # round trip: inverse(forward(x)) = x / N  (each direction scales by 1/N)
    mid = torch.empty_like(dev)
    plan.exec(dev, dev[n:], mid, mid[n:])
    big = (mid.float() * 64).half()                   # keep the second pass out of the subnormal range
    plan.exec_inverse(big, big[n:], out, out[n:])
    torch.cuda.synchronize()
    back = out.cpu().numpy().reshape(batch, 2, n).astype(np.float64) * n / 64
    assert np.abs(back[:, 0] - re).max() < 0.05 and np.abs(back[:, 1] - im).max() < 0.05


def test_layout_adapters_roundtrip(tf, torch):
    from tensor_fft_amd import capi

    n = 1 << 16
    z = (torch.rand(n, 2, device="cuda") * 2 - 1).half()            # interleaved (re, im)
    re, im = torch.empty(n, dtype=torch.float16, device="cuda"), torch.empty(n, dtype=torch.float16, device="cuda")
    capi.deinterleave(z, re, im)
    assert bool((re == z[:, 0]).all()) and bool((im == z[:, 1]).all())
    back = torch.empty_like(z)
    capi.interleave(re, im, back)
    assert bool((back == z).all())
    # an interleaved caller end to end: hipFFT-style buffer -> planar -> FFT -> interleaved, vs torch.fft in fp32
    plan = tf.TfftPlan(n, 1, 0, in_batch_stride=n, out_batch_stride=n)
    o_re, o_im = torch.empty_like(re), torch.empty_like(im)
    plan.exec(re, im, o_re, o_im)
    capi.interleave(o_re, o_im, back)
    want = torch.fft.fft(torch.complex(z[:, 0].float(), z[:, 1].float())) / n
    got = torch.complex(back[:, 0].float(), back[:, 1].float())
    assert float((got - want).abs().pow(2).sum().sqrt() / want.abs().pow(2).sum().sqrt()) < REL_L2_TOL


@pytest.mark.parametrize("n,batch", [(1 << 16, 8), (1 << 13, 6), (1 << 17, 4), (1 << 21, 1)])
def test_exec_is_graph_capturable(tf, torch, n, batch):
    """tfft_exec allocates nothing once the workspace is set, so a pass chain can be captured in a HIP graph
    (column passes, the single-pass 8192 kernel, radix-512 passes, fused tails)."""
    x = (torch.rand(batch * 2 * n, device="cuda") * 2 - 1).half()
    y = torch.zeros_like(x)
    plan = tf.TfftPlan(n, batch, 0, preserve_input=True)
    ws = torch.empty(max(1, plan.workspace_bytes // 2), dtype=torch.float16, device="cuda")
    if plan.workspace_bytes:
        plan.set_workspace(ws)
    plan.exec(x, x[n:], y, y[n:])                     # warm-up (sets function attributes)
    torch.cuda.synchronize()
    ref = y.clone()
    y.zero_()
    s = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        with torch.cuda.graph(graph, stream=s):
            plan.exec(x, x[n:], y, y[n:], stream=s.cuda_stream)
    torch.cuda.synchronize()
    graph.replay()
    torch.cuda.synchronize()
    assert bool((y == ref).all())


def test_2d_inverse_round_trip(tf, torch):
    """inverse(forward(x)) = x / (rows cols): amplitudes chosen so that the round trip stays above the subnormal range."""
    rows, cols, batch = 256, 512, 2
    rng = np.random.default_rng(3)
    re = (rng.uniform(-1, 1, (batch, rows, cols)) * 30000).astype(np.float16)
    im = (rng.uniform(-1, 1, (batch, rows, cols)) * 30000).astype(np.float16)
    d_re, d_im = torch.from_numpy(re).cuda().reshape(-1), torch.from_numpy(im).cuda().reshape(-1)
    f_re, f_im = torch.empty_like(d_re), torch.empty_like(d_im)
    b_re, b_im = torch.empty_like(d_re), torch.empty_like(d_im)
    plan = tf.TfftPlan2D(rows, cols, batch, 0)
    plan.exec(d_re, d_im, f_re, f_im)
    want = np.fft.ifft2(np.fft.fft2(_c(re, im), axes=(1, 2)) / (rows * cols), axes=(1, 2))    # = x / (rows cols)
    big_re, big_im = (f_re.float() * 256).half(), (f_im.float() * 256).half()                  # keep pass 2 out of the subnormals
    plan.exec_inverse(big_re, big_im, b_re, b_im)
    torch.cuda.synchronize()
    got = _c(b_re.cpu().numpy(), b_im.cpu().numpy()).reshape(batch, rows, cols) / 256
    assert np.linalg.norm(got - want) / np.linalg.norm(want) < 3e-3


def test_2d_plan_is_graph_capturable(tf, torch):
    n, batch = 4096, 2
    re = (torch.rand(batch * n * n, device="cuda") * 2 - 1).half()
    im = (torch.rand(batch * n * n, device="cuda") * 2 - 1).half()
    o_re, o_im = torch.zeros_like(re), torch.zeros_like(im)
    plan = tf.TfftPlan2D(n, n, batch, 0)
    plan.exec(re, im, o_re, o_im)                     # warm-up: workspace, function attributes
    torch.cuda.synchronize()
    ref_re, ref_im = o_re.clone(), o_im.clone()
    o_re.zero_(); o_im.zero_()
    s = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        with torch.cuda.graph(graph, stream=s):
            plan.exec(re, im, o_re, o_im, stream=s.cuda_stream)
    torch.cuda.synchronize()
    graph.replay()
    torch.cuda.synchronize()
    assert bool((o_re == ref_re).all()) and bool((o_im == ref_im).all())


def test_argument_errors(tf, torch):
    n = 4096
    plan = tf.TfftPlan(n, 2, 0)
    buf = torch.zeros(4 * n + 8, dtype=torch.float16, device="cuda")
    with pytest.raises(tf.TfftError):
        plan.exec_ptr(buf.data_ptr() + 2, buf.data_ptr() + 2 * n, buf.data_ptr(), buf.data_ptr())   # misaligned
    with pytest.raises(tf.TfftError):
        plan.exec_ptr(0, buf.data_ptr(), buf.data_ptr(), buf.data_ptr())                            # null
    with pytest.raises(tf.TfftError):
        plan.exec(buf[:n], buf[n:], buf, buf[n:])                                                   # plane too short
    with pytest.raises(tf.TfftError):
        tf.TfftPlan(3000, 1, 0)
    with pytest.raises(tf.TfftError):
        tf.TfftPlan(4096, 1, 0, in_batch_stride=4100)


def test_cxx_shim_example():
    exe = os.path.join(ROOT, "examples", "example_single_fft")
    for args in (["12", "5"], ["8", "3"], ["13", "2"], ["16", "2"]):
        r = subprocess.run([exe] + args, capture_output=True, text=True, timeout=300)
        print(r.stdout)
        assert r.returncode == 0, r.stdout + r.stderr


def test_cxx_batched_benchmark_driver():
    """examples/bench_batched.cpp: the C++ counterpart of the reference's FFTBenchBatch.cu over the shim + C ABI."""
    exe = os.path.join(ROOT, "examples", "bench_batched")
    for args in (["12", "512", "5", "2"], ["10", "1000", "5", "2"], ["14", "64", "5", "2"]):
        r = subprocess.run([exe] + args, capture_output=True, text=True, timeout=300)
        print(r.stdout)
        assert r.returncode == 0 and "Gsamples/s" in r.stdout, r.stdout + r.stderr


def test_one_plan_from_several_host_threads(tf, torch):
    """the plan is immutable after creation, so host threads may share it (each on its own stream and
    buffers). N = 4096 is a single-pass plan: no workspace is shared between the threads."""
    import threading
    n, batch = 4096, 64
    plan = tf.TfftPlan(n, batch, 0)
    rng = np.random.default_rng(77)
    xs = [torch.from_numpy(rng.uniform(-1, 1, batch * 2 * n).astype(np.float16)).cuda() for _ in range(4)]
    want = []
    for x in xs:
        y = torch.empty_like(x)
        plan.exec(x, x[n:], y, y[n:])
        torch.cuda.synchronize()
        want.append(y.clone())
    got = [torch.zeros_like(x) for x in xs]
    errs = []

    def work(i):
        try:
            s = torch.cuda.Stream()
            for _ in range(20):
                plan.exec(xs[i], xs[i][n:], got[i], got[i][n:], s.cuda_stream)
            s.synchronize()
        except Exception as e:   # noqa: BLE001
            errs.append(e)

    ts = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    for t in ts: t.start()
    for t in ts: t.join()
    assert not errs, errs
    for g_, w_ in zip(got, want):
        assert bool((g_ == w_).all())


@pytest.mark.parametrize("lg", [8, 9, 11, 12, 13, 15, 16, 17, 18, 19, 20, 21, 23, 25, 26, 27])
def test_against_vendor_fft_on_device(tf, torch, lg):
    """Independent cross-check that needs no CPU oracle: hipFFT (through torch.fft, complex64) on the same fp16 input,
    the role cuFFT plays in the reference's tests (CuFFTTest.h:193-261). Large N run here in seconds."""
    n = 1 << lg
    batch = max(1, (1 << 22) // n)
    gen = torch.Generator(device="cuda").manual_seed(lg)
    x = (torch.rand(batch, 2, n, device="cuda", generator=gen) * 2 - 1).half()
    flat = x.reshape(-1)
    y = torch.empty_like(flat)
    tf.TfftPlan(n, batch, 0, preserve_input=True).exec(flat, flat[n:], y, y[n:])
    want = torch.fft.fft(torch.complex(x[:, 0].float(), x[:, 1].float()), dim=1) / n
    got = y.reshape(batch, 2, n)
    got = torch.complex(got[:, 0].float(), got[:, 1].float())
    rel = float(torch.linalg.vector_norm(got - want) / torch.linalg.vector_norm(want))
    assert rel <= REL_L2_TOL, rel
    assert float((got - want).abs().max()) <= 8 * 2.0 ** -11 * float(want.abs().max())


def test_randomized_plan_space(tf, torch):
    """60 random points of the plan space (length 2 .. 2^19, batch, padded / planar strides, preserve_input, in place)
    against numpy's fp64 FFT: every kernel family and launch split gets exercised with shapes nobody hand-picked."""
    rng = np.random.default_rng(20261004)
    for case in range(60):
        lg = int(rng.integers(1, 20))
        n = 1 << lg
        batch = int(rng.integers(1, max(2, min(48, (1 << 21) // n))))
        planar = bool(rng.integers(0, 2)) and n >= 8
        pad = int(rng.integers(0, 3)) * 8 if n >= 8 else 0
        in_place = bool(rng.integers(0, 2)) and not planar and pad == 0
        preserve = bool(rng.integers(0, 2)) and not in_place
        re = rng.uniform(-1, 1, (batch, n)).astype(np.float16)
        im = rng.uniform(-1, 1, (batch, n)).astype(np.float16)
        exact = np.fft.fft(_c(re, im), axis=1) / n
        if planar:                      # all RE planes, then all IM planes, each with its own batch stride
            stride = n + pad
            d_re = torch.zeros(batch, stride, dtype=torch.float16, device="cuda"); d_re[:, :n] = torch.from_numpy(re).cuda()
            d_im = torch.zeros(batch, stride, dtype=torch.float16, device="cuda"); d_im[:, :n] = torch.from_numpy(im).cuda()
            o_re, o_im = torch.zeros_like(d_re), torch.zeros_like(d_im)
            plan = tf.TfftPlan(n, batch, 0, in_batch_stride=stride, out_batch_stride=stride, preserve_input=preserve)
            keep = d_re.clone()
            plan.exec(d_re.reshape(-1), d_im.reshape(-1), o_re.reshape(-1), o_im.reshape(-1))
            torch.cuda.synchronize()
            got = _c(o_re[:, :n].cpu().numpy(), o_im[:, :n].cpu().numpy())
            if preserve:
                assert bool((d_re == keep).all()), (case, n, batch)
        else:                           # [fft RE | fft IM] blocks, optionally padded
            stride = 2 * n + pad
            blk = torch.zeros(batch, stride, dtype=torch.float16, device="cuda")
            blk[:, :n] = torch.from_numpy(re).cuda(); blk[:, n:2 * n] = torch.from_numpy(im).cuda()
            out = blk if in_place else torch.zeros_like(blk)
            plan = tf.TfftPlan(n, batch, 0, in_batch_stride=stride, out_batch_stride=stride, preserve_input=preserve)
            keep = blk.clone()
            flat_in, flat_out = blk.reshape(-1), out.reshape(-1)
            plan.exec(flat_in, flat_in[n:], flat_out, flat_out[n:])
            torch.cuda.synchronize()
            got = _c(out[:, :n].cpu().numpy(), out[:, n:2 * n].cpu().numpy())
            if preserve:
                assert bool((blk == keep).all()), (case, n, batch)
        rel = np.linalg.norm(got - exact) / np.linalg.norm(exact)
        assert np.isfinite(got).all() and rel <= REL_L2_TOL, (case, n, batch, planar, pad, in_place, preserve, rel)


def test_randomized_strided_axis_and_2d(tf, torch):
    """Random transforms along a strided axis (data [batch][n][C]) and random 2D shapes against numpy."""
    rng = np.random.default_rng(4242)
    for case in range(30):
        lg = int(rng.integers(1, 14))
        n = 1 << lg
        inner = int(2 ** rng.integers(3, 11))
        batch = int(rng.integers(1, 4))
        if n * inner * batch > (1 << 23):
            continue
        re = rng.uniform(-1, 1, (batch, n, inner)).astype(np.float16)
        im = rng.uniform(-1, 1, (batch, n, inner)).astype(np.float16)
        host = np.ascontiguousarray(np.stack([re, im], axis=1))           # [batch][RE | IM][n][C]
        dev = torch.from_numpy(host).cuda().reshape(-1)
        out = torch.zeros_like(dev)
        tf.TfftPlan(n, batch, 0, inner=inner).exec(dev, dev[n * inner:], out, out[n * inner:])
        torch.cuda.synchronize()
        o = out.cpu().numpy().reshape(batch, 2, n, inner)
        got = _c(o[:, 0], o[:, 1])
        exact = np.fft.fft(_c(re, im), axis=1) / n
        rel = np.linalg.norm(got - exact) / np.linalg.norm(exact)
        assert np.isfinite(got).all() and rel <= REL_L2_TOL, (case, n, inner, batch, rel)
    for case in range(8):
        rows, cols = int(2 ** rng.integers(1, 11)), int(2 ** rng.integers(3, 12))
        batch = int(rng.integers(1, 4))
        re = rng.uniform(-1, 1, (batch, rows, cols)).astype(np.float16)
        im = rng.uniform(-1, 1, (batch, rows, cols)).astype(np.float16)
        d_re, d_im = torch.from_numpy(re).cuda().reshape(-1), torch.from_numpy(im).cuda().reshape(-1)
        o_re, o_im = torch.empty_like(d_re), torch.empty_like(d_im)
        tf.TfftPlan2D(rows, cols, batch, 0).exec(d_re, d_im, o_re, o_im)
        torch.cuda.synchronize()
        got = _c(o_re.cpu().numpy(), o_im.cpu().numpy()).reshape(batch, rows, cols)
        exact = np.fft.fft2(_c(re, im), axes=(1, 2)) / (rows * cols)
        rel = np.linalg.norm(got - exact) / np.linalg.norm(exact)
        assert rel <= REL_L2_TOL, (case, rows, cols, batch, rel)

