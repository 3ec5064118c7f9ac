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


@pytest.mark.parametrize("lg,batch", [(16, 1), (18, 1), (20, 1), (17, 3)])
def test_latency_kernel_is_deterministic(tf, lg, batch):
    """25 executions of the small-work plan on the same input: the same bits every time (no race between the waves that share a
    column group's stage 2, the staging image and the read-out)."""
    import torch

    n = 1 << lg
    x = torch.empty(batch * 2 * n, dtype=torch.float16, device="cuda")
    tf.synth_uniform(x, x[n:], n, batch, seed=33)
    plan = tf.TfftPlan(n, batch, 0, preserve_input=True)
    first = None
    for _ in range(25):
        y = torch.full_like(x, float("nan"))
        plan.exec(x, x[n:], y, y[n:])
        torch.cuda.synchronize()
        if first is None:
            first = y
        assert bool((y.view(torch.int16) == first.view(torch.int16)).all())
    plan.close()


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


# ---------------------------------------------------------------------------------------------------------------------
# the exchange of a distributed transform overlapped with its column pass, slab by slab (TFFT_DIST_SLABS_*)
# ---------------------------------------------------------------------------------------------------------------------
def _all_ranks(torch, capi, xr, xi, world, slabs):
    """All `world` ranks of one transform as plans in this process (comm = NULL), the exchange emulated by device copies of whole
    chunks (a peer's chunk is one contiguous block for every number of slabs): returns every rank's output planes."""
    n = xr.size
    plans = [capi.DistPlan(n, world, r, 0, slabs=slabs) for r in range(world)]
    g = plans[0].geometry
    assert int(g.slabs) == slabs
    n1, n2, c, chunk = int(g.n1), int(g.n2), int(g.cols), int(g.chunk)
    loc = n // world
    mk = lambda: torch.full((loc,), float("nan"), dtype=torch.float16, device="cuda")      # noqa: E731
    bufs = []
    for p in plans:
        s_re, s_im = mk(), mk()
        b = (s_re, s_im, mk(), mk()) if world > 1 else (s_re, s_im, s_re, s_im)
        p.set_buffers(*b)
        bufs.append(b)
    x2r, x2i = xr.reshape(n1, n2), xi.reshape(n1, n2)
    for r, p in enumerate(plans):
        p.pre(torch.from_numpy(np.ascontiguousarray(x2r[:, r * c:(r + 1) * c]).reshape(-1)).cuda(),
              torch.from_numpy(np.ascontiguousarray(x2i[:, r * c:(r + 1) * c]).reshape(-1)).cuda())
    torch.cuda.synchronize()
    if world > 1:
        for q in range(world):
            for pp in range(world):
                bufs[q][2][pp * chunk:(pp + 1) * chunk].copy_(bufs[pp][0][q * chunk:(q + 1) * chunk])
                bufs[q][3][pp * chunk:(pp + 1) * chunk].copy_(bufs[pp][1][q * chunk:(q + 1) * chunk])
    outs = []
    for p in plans:
        o_re, o_im = mk(), mk()
        p.post(o_re, o_im)
        torch.cuda.synchronize()
        outs.append((o_re.clone(), o_im.clone()))
    for p in plans:
        p.close()
    return outs, g


@pytest.mark.parametrize("lg,world", [(25, 2), (25, 4), (26, 8), (25, 1)])
def test_column_slabs_give_the_same_bits_as_one_slab(tf, orc, lg, world):
    """S = 2 and S = 4 slabs against S = 1, all ranks of the transform in one process: every rank's output bit for bit (the slabs
    only cut the column pass's launch and re-lay the exchange buffers), and rank 0 / the last rank against the fp64 oracle at the
    smaller sizes."""
    import torch
    from tensor_fft_amd import capi

    n = 1 << lg
    xr, xi = orc.synth_uniform(n, 1, 0, 900 + lg)
    base, g = _all_ranks(torch, capi, xr[0], xi[0], world, 1)
    assert (int(g.n1), int(g.reorder)) == (256, 0)
    for slabs in (2, 4):
        outs, _ = _all_ranks(torch, capi, xr[0], xi[0], world, slabs)
        for r in range(world):
            assert bool((outs[r][0].view(torch.int16) == base[r][0].view(torch.int16)).all()), (slabs, r)
            assert bool((outs[r][1].view(torch.int16) == base[r][1].view(torch.int16)).all()), (slabs, r)
    if lg <= 25 and world > 1:
        exact = np.fft.fft(xr[0].astype(np.float64) + 1j * xi[0].astype(np.float64)) / n
        n1, n2, k = int(g.n1), int(g.n2), int(g.rows)
        for r in (0, world - 1):
            got = base[r][0].cpu().numpy().astype(np.float64) + 1j * base[r][1].cpu().numpy().astype(np.float64)
            k1 = r * k + np.arange(k)[:, None]
            want = exact[(k1 + n1 * np.arange(n2)[None, :]).reshape(-1)]
            assert np.linalg.norm(got - want) / np.linalg.norm(want) < REL_L2_TOL


def test_column_slabs_are_refused_where_the_geometry_cannot_overlap(tf):
    from tensor_fft_amd import capi

    with pytest.raises(tf.TfftError, match="SLABS"):
        capi.DistPlan(1 << 20, 2, 0, 0, slabs=2)          # 256 x 4096: the row transforms are single kernels behind a re-order pass
    with pytest.raises(tf.TfftError, match="SLABS"):
        capi.DistPlan(1 << 21, 8, 0, 0, slabs=4)          # 512 x 4096: the column pass is radix 512


def test_overlapped_exchange_through_a_real_communicator(tf):
    """World 1 with the own chunk through ncclSend / ncclRecv (TFFT_DIST_SELF_VIA_COMM), S = 1, 2, 4 through tfft_dist_exec: with
    S > 1 the exchange runs on the plan's second stream behind per-slab events. Same bits for every S, correct against numpy, and
    the overlapped transform is not slower than column pass + exchange one after the other."""
    import subprocess

    code = r'''
import numpy as np, torch, time
import __graft_entry__ as g
g.build()
from tensor_fft_amd.distributed import DistributedFFT1D, HipEngine
n = 1 << 26
rng = np.random.default_rng(5)
xr, xi = rng.uniform(-1, 1, n).astype(np.float16), rng.uniform(-1, 1, n).astype(np.float16)
exact = np.fft.fft(xr.astype(np.float64) + 1j * xi.astype(np.float64)) / n
ref = None
for slabs in (1, 2, 4):
    f = DistributedFFT1D(n, engine=HipEngine(0), transport="rccl", self_via_comm=True, slabs=slabs)
    assert f.geometry.slabs == slabs
    idx = f.input_indices()
    a, b = torch.from_numpy(xr[idx].copy()).cuda(), torch.from_numpy(xi[idx].copy()).cuda()
    for _ in range(3):
        re, im = f.forward(a, b)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        re, im = f.forward(a, b)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 20 * 1e3
    ph = f.phase_times(a, b, 10)
    got = re.cpu().numpy().astype(np.float64) + 1j * im.cpu().numpy().astype(np.float64)
    want = exact[f.output_indices()]
    rel = float(np.linalg.norm(got - want) / np.linalg.norm(want))
    assert rel < 1.5e-3, rel
    if ref is None:
        ref = (re.clone(), im.clone())
    else:
        assert bool((re.view(torch.int16) == ref[0].view(torch.int16)).all()) and bool((im.view(torch.int16) == ref[1].view(torch.int16)).all())
    print("slabs %d: %.3f ms per transform; phases one after the other: pre %.3f + exchange %.3f + post %.3f = %.3f ms; rel-L2 %.2e"
          % (slabs, ms, ph["pre_ms"], ph["exchange_ms"], ph["post_ms"], ph["pre_ms"] + ph["exchange_ms"] + ph["post_ms"], rel))
    # (on ONE GPU the "exchange" is a device-local copy through RCCL that competes with the column pass for the same HBM, and S
    # launches + S send / receive groups cost their launch overheads: the overlap cannot pay here, it must only not cost more than
    # those overheads; what it is FOR - xGMI busy while the next slab computes - needs a node)
    if slabs > 1:
        assert ms < (ph["pre_ms"] + ph["exchange_ms"] + ph["post_ms"]) * 1.3, "the overlapped transform is much slower than its phases in a row"   # (measured 1.02 - 1.08; a wall-clock bound in a test must leave room for a noisy box)
    f.close()
print("SLABS-OK")
'''
    r = subprocess.run(["timeout", "-k", "10", "300", "python3", "-c", code], cwd=ROOT, capture_output=True, text=True)
    print(r.stdout[-4000:], r.stderr[-4000:])
    assert r.returncode == 0 and "SLABS-OK" in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("lg,order", [(20, "transposed"), (21, "transposed")])
def test_mfma_flops_are_linear_in_the_batch_across_the_chunk_boundary(tf, lg, order):
    """tfft_plan_mfma_flops of a chunked plan (sub-plans built for ONE chunk, run batch / chunk times plus a tail pair): the same
    flops per transform below one chunk, at whole multiples and with a ragged tail, and equal to the unchunked natural-order plan's
    count where both run the same kernels' MFMA stages."""
    per = []
    for batch in (1, 64, 128, 256, 300, 513):
        plan = tf.TfftPlan(1 << lg, batch, 0, output_order=order)
        f = plan.mfma_flops
        assert f > 0
        per.append(f / batch)
        plan.close()
    assert max(per) - min(per) <= 1e-9 * max(per), per


@pytest.mark.parametrize("lg", [13, 14])
@pytest.mark.parametrize("batch", [1, 3, 4])
def test_cooperative_radix64_pass(tf, orc, lg, batch):
    """2^14 = 256 x 64 (2^13 = 256 x 32): the latency column kernel on 64 (32) columns, then stockham::tail_coop_kernel<64> (<32>)
    (4 x 4 x 4, 4 x 8 through LDS): the default for up to 4 transforms. Against orc.dft64 and against the single-pass kernel; with
    variant bit 4194304 the final pass runs one butterfly per thread instead: same values to two roundings."""
    import torch

    n = 1 << lg
    v = tf.plan_default_variant(n, 1, batch)
    assert tf.plan_describe(n, 1, v) == "col:256+tw autosort:%d-tw" % (n // 256)
    y = _run(tf, torch, n, batch, 150 + batch)
    assert _check(orc, y, n, batch, 150 + batch) <= REL_L2_TOL
    for other in (NO_LAT, v | 4194304):          # the single-pass kernel; the butterfly-per-thread radix-64 pass
        y1 = _run(tf, torch, n, batch, 150 + batch, variant=other)
        assert _check(orc, y1, n, batch, 150 + batch) <= REL_L2_TOL
        d = (y.float() - y1.float()).double()
        assert float(d.norm() / y1.double().norm()) <= 8e-4


@pytest.mark.parametrize("n,batch", [(256, 100), (256, 3000), (1024, 40), (2048, 300), (4096, 9), (4096, 700), (8192, 300), (16384, 200)])
def test_small_batches_spread_over_the_cus(tf, orc, n, batch):
    """A batch that does not fill the chip runs with fewer working waves per workgroup on more CUs (tfft.hip live_waves, the
    one-group-per-workgroup form of k4096r): bit-identical to the packed launch shape (variant bit 4194304), every transform
    written, first / last / a middle one against the oracle."""
    import torch

    y = _run(tf, torch, n, batch, 170)
    y_packed = _run(tf, torch, n, batch, 170, variant=4194304)
    assert bool((y.view(torch.int16) == y_packed.view(torch.int16)).all())
    assert _check(orc, y, n, batch, 170, ids=(0, batch // 2, batch - 1)) <= REL_L2_TOL


@pytest.mark.parametrize("batch", [1, 5, 8])
def test_cooperative_radix128_pass(tf, orc, batch):
    """2^15 = 256 x 128: the latency column kernel, then stockham::tail_coop_kernel<128> (8 columns per workgroup, 4 x 4 x 8 through
    LDS in fp32): the default for up to 8 transforms. Against orc.dft64, against the single-pass kernel (two roundings apart at
    most), in place, and with padded batch strides on both sides."""
    import torch

    n = 1 << 15
    v = tf.plan_default_variant(n, 1, batch)
    assert tf.plan_describe(n, 1, v) == "col:256+tw autosort:128-tw"
    plan = tf.TfftPlan(n, batch, 0)
    assert plan.num_launches == 2
    y = _run(tf, torch, n, batch, 140 + batch)
    assert _check(orc, y, n, batch, 140 + batch) <= REL_L2_TOL
    y1 = _run(tf, torch, n, batch, 140 + batch, variant=NO_LAT)          # any explicit bit: exactly that variant = the single-pass kernel
    d = (y.float() - y1.float()).double()
    assert float(d.norm() / y1.double().norm()) <= 8e-4          # (two results, each within 5e-4 of the exact one)
    x = torch.empty(batch * 2 * n, dtype=torch.float16, device="cuda")
    tf.synth_uniform(x, x[n:], n, batch, seed=140 + batch)
    work = x.clone()
    plan.exec(work, work[n:], work, work[n:])
    torch.cuda.synchronize()
    assert bool((work.view(torch.int16) == y.view(torch.int16)).all())
    plan.close()
    pad_in, pad_out = 2 * n + 512, 2 * n + 1024
    xp = torch.zeros(batch * pad_in, dtype=torch.float16, device="cuda")
    yp = torch.full((batch * pad_out,), float("nan"), dtype=torch.float16, device="cuda")
    for b in range(batch):
        xp[b * pad_in:b * pad_in + 2 * n] = x[b * 2 * n:(b + 1) * 2 * n]
    plan = tf.TfftPlan(n, batch, 0, in_batch_stride=pad_in, out_batch_stride=pad_out, preserve_input=True)
    plan.exec(xp, xp[n:], yp, yp[n:])
    torch.cuda.synchronize()
    for b in range(batch):
        assert bool((yp[b * pad_out:b * pad_out + 2 * n].view(torch.int16) == y[b * 2 * n:(b + 1) * 2 * n].view(torch.int16)).all())
        assert bool(torch.isnan(yp[b * pad_out + 2 * n:(b + 1) * pad_out]).all())
    plan.close()


@pytest.mark.parametrize("lg,batch", [(18, 1), (18, 3), (21, 1), (17, 2)])
def test_in_place_with_two_workspace_blocks(tf, orc, lg, batch):
    """In place, odd number of passes (three): with a workspace of twice tfft_plan_workspace_bytes the chain runs IN -> A -> B -> IN
    without the leading copy; with the plain workspace it starts from a copy. Both give the out-of-place bits."""
    import torch

    n = 1 << lg
    x = torch.empty(batch * 2 * n, dtype=torch.float16, device="cuda")
    tf.synth_uniform(x, x[n:], n, batch, seed=160 + lg)
    plan = tf.TfftPlan(n, batch, 0, preserve_input=True)      # (otherwise the out-of-place run below may use x as its scratch)
    assert plan.num_launches == 3
    ref = torch.empty_like(x)
    plan.exec(x, x[n:], ref, ref[n:])
    torch.cuda.synchronize()
    assert _check(orc, ref, n, batch, 160 + lg, ids=(0, batch - 1)) <= REL_L2_TOL
    for blocks in (1, 2):
        ws = torch.empty(blocks * plan.workspace_bytes // 2, dtype=torch.float16, device="cuda")
        plan.set_workspace(ws)
        work = x.clone()
        plan.exec(work, work[n:], work, work[n:])
        torch.cuda.synchronize()
        assert bool((work.view(torch.int16) == ref.view(torch.int16)).all()), blocks
    plan.close()
