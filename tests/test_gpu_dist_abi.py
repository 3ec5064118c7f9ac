"""GPU side of the distributed-transform ABI (tfft_dist_*): BASELINE configs[4b].

One MI355X is all a test box has, RCCL wants one GPU per rank, so the pieces are covered like this:
  * all P ranks of a transform as P plans in ONE process (comm = NULL), the exchange emulated by device copies between the plans'
    buffers: the column pass with each rank's four-step twiddle, the chunk layout, the segmented (re-order-free) row pass and
    the output layout run exactly as on P GPUs, up to N = 2^26 over 8 ranks, against the CPU oracle;
  * a REAL RCCL communicator with one rank whose own chunk is routed through ncclSend / ncclRecv (TFFT_DIST_SELF_VIA_COMM): kernel ->
    collective -> kernel on one stream, through the same tfft_dist_exec the 8-GPU run uses;
  * the C++ host (include/tensor_fft.hpp: DataHandlerMultiGPU / ComputeFFTMultiGPU) over the same entry points.
Several processes on this one GPU with a host-staged exchange: tests/test_gpu_distributed.py."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def tf():
    import __graft_entry__ as g

    g.build()
    import tensor_fft_amd as t

    t.device_check(0)
    return t


@pytest.fixture(scope="module")
def signal26(orc):
    """One N = 2^26 signal and its fp64 oracle spectrum, shared by the tests of this module (the host FFT takes ~30 s)."""
    n = 1 << 26
    xr, xi = orc.synth_uniform(n, 1, 0, 2026)
    e_re, e_im = orc.dft64(xr, xi)
    return xr[0], xi[0], e_re[0] + 1j * e_im[0]


def _emulate(torch, capi, xr, xi, exact, world):
    """Runs all `world` ranks of one transform in this process; returns the worst rel-L2 error over the ranks."""
    n = xr.size
    plans = [capi.DistPlan(n, world, r, 0) for r in range(world)]
    g = plans[0].geometry
    n1, n2, c, k, chunk = int(g.n1), int(g.n2), int(g.cols), int(g.rows), int(g.chunk)
    loc = n // world
    mk = lambda: torch.full((loc,), float("nan"), dtype=torch.float16, device="cuda")      # noqa: E731
    bufs = []
    for p in plans:
        s_re, s_im = mk(), mk()
        b = (s_re, s_im, mk(), mk()) if world > 1 else (s_re, s_im, s_re, s_im)
        p.set_buffers(*b)
        bufs.append(b)
    x2r, x2i = xr.reshape(n1, n2), xi.reshape(n1, n2)
    for r, p in enumerate(plans):       # input layout "columns": rank r holds columns [r C, (r + 1) C) of the [N1][N2] view
        p.pre(torch.from_numpy(np.ascontiguousarray(x2r[:, r * c:(r + 1) * c]).reshape(-1)).cuda(),
              torch.from_numpy(np.ascontiguousarray(x2i[:, r * c:(r + 1) * c]).reshape(-1)).cuda())
    torch.cuda.synchronize()
    for q in range(world):              # the exchange: chunk q of rank p' -> slot p' of rank q
        for pp in range(world):
            if world > 1:
                bufs[q][2][pp * chunk:(pp + 1) * chunk].copy_(bufs[pp][0][q * chunk:(q + 1) * chunk])
                bufs[q][3][pp * chunk:(pp + 1) * chunk].copy_(bufs[pp][1][q * chunk:(q + 1) * chunk])
    worst = 0.0
    for r, p in enumerate(plans):
        o_re, o_im = mk(), mk()
        p.post(o_re, o_im)
        torch.cuda.synchronize()
        got = o_re.cpu().numpy().astype(np.float64) + 1j * o_im.cpu().numpy().astype(np.float64)
        k1 = r * k + np.arange(k)[:, None]
        want = exact[(k1 + n1 * np.arange(n2)[None, :]).reshape(-1)]          # output layout "transposed": [K][N2]
        assert np.isfinite(got).all()
        worst = max(worst, float(np.linalg.norm(got - want) / np.linalg.norm(want)))
    return worst, g


@pytest.mark.parametrize("lg,world", [(16, 2), (20, 2), (20, 8), (21, 4), (24, 4), (25, 4), (25, 8), (27, 2)])
def test_all_ranks_in_one_process(tf, orc, lg, world):
    import torch
    from tensor_fft_amd import capi

    n = 1 << lg
    xr, xi = orc.synth_uniform(n, 1, 0, lg * 16 + world)
    e_re, e_im = orc.dft64(xr, xi)
    worst, g = _emulate(torch, capi, xr[0], xi[0], e_re[0] + 1j * e_im[0], world)
    assert worst < 1.5e-3, (lg, world, worst)
    if lg >= 25:
        assert g.reorder == 0       # the row pass read the received chunks in place


@pytest.mark.parametrize("world", [8, 2])
def test_configs_4b_all_ranks_in_one_process(tf, signal26, world):
    """BASELINE configs[4b]: N = 2^26, here all 8 (2) ranks on one GPU: 256 x 2^18, 2^15 columns and 32 rows per rank, chunks of
    1 Mi samples, three local passes per rank, the full spectrum against the fp64 oracle."""
    import torch
    from tensor_fft_amd import capi

    xr, xi, exact = signal26
    worst, g = _emulate(torch, capi, xr, xi, exact, world)
    assert (g.n1, g.n2, g.reorder, g.local_passes) == (256, 1 << 18, 0, 3)
    assert worst < 1.5e-3, worst


def test_exchange_needs_a_communicator(tf):
    import torch
    from tensor_fft_amd import capi

    p = capi.DistPlan(1 << 20, 2, 0, 0)
    with pytest.raises(tf.TfftError) as e:
        p.exchange()
    assert e.value.code == 9 and "communicator" in e.value.message
    x = torch.zeros(1 << 19, dtype=torch.float16, device="cuda")
    with pytest.raises(tf.TfftError):
        p.exec(x, x, x.clone(), x.clone())
    with pytest.raises(tf.TfftError):
        capi.DistPlan(1 << 14, 2, 0, 0)               # too small for two ranks


def _run_isolated(code, timeout=300):
    """A collective that never completes must not take the test process (and the GPU box) with it: run it in a child with a
    hard time limit."""
    r = subprocess.run(["timeout", "-k", "10", str(timeout), "python3", "-c", code], cwd=ROOT, capture_output=True, text=True)
    print(r.stdout[-4000:], r.stderr[-4000:])
    return r


def test_rccl_communicator_with_own_chunk_through_the_collective(tf):
    """A real RCCL communicator (ncclCommInitRank through tfft_dist_comm_create, one rank) and tfft_dist_exec with the own chunk
    sent through ncclSend / ncclRecv: the stream order kernel -> collective -> kernel of the 8-GPU path, checked at 2^20 and 2^26
    against numpy's fp64 FFT, twice (second call: same buffers, same bits)."""
    code = r'''
import numpy as np, torch
import __graft_entry__ as g
g.build()
from tensor_fft_amd.distributed import DistributedFFT1D, HipEngine
for lg in (20, 26):
    n = 1 << lg
    rng = np.random.default_rng(lg)
    xr, xi = rng.uniform(-1, 1, n).astype(np.float16), rng.uniform(-1, 1, n).astype(np.float16)
    f = DistributedFFT1D(n, engine=HipEngine(0), transport="rccl", self_via_comm=True)
    assert f.transport == "rccl" and f._comm is not None and f.geometry.world == 1
    idx = f.input_indices()
    a, b = torch.from_numpy(xr[idx].copy()).cuda(), torch.from_numpy(xi[idx].copy()).cuda()
    re, im = f.forward(a, b)
    torch.cuda.synchronize()
    exact = np.fft.fft(xr.astype(np.float64) + 1j * xi.astype(np.float64)) / n
    got = re.cpu().numpy().astype(np.float64) + 1j * im.cpu().numpy().astype(np.float64)
    want = exact[f.output_indices()]
    rel = float(np.linalg.norm(got - want) / np.linalg.norm(want))
    assert rel < 1.5e-3, rel
    keep = re.clone()
    re2, _ = f.forward(a, b)
    torch.cuda.synchronize()
    assert bool((re2 == keep).all())
    print("N=2^%d through RCCL: rel-L2 %.2e" % (lg, rel))
print("RCCL-SELF-OK")
'''
    r = _run_isolated(code)
    assert r.returncode == 0 and "RCCL-SELF-OK" in r.stdout, r.stdout + r.stderr


def test_cxx_multi_gpu_host(tf):
    """include/tensor_fft.hpp: DataHandlerMultiGPU / ComputeFFTMultiGPU (one transform over the visible devices; here one,
    with and without the own chunk through a real RCCL communicator) and DataBatchHandlerMultiGPU / ComputeFFTsMultiGPU."""
    exe = os.path.join(ROOT, "examples", "example_multi_gpu_fft")
    for args in (["20", "1", "0"], ["20", "1", "1"], ["26", "1", "1"], ["16", "1", "0"]):
        r = subprocess.run(["timeout", "-k", "10", "300", exe] + args, capture_output=True, text=True)
        print(r.stdout)
        assert r.returncode == 0 and "OK" in r.stdout, r.stdout + r.stderr
