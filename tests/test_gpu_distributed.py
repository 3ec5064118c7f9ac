"""GPU side of the distributed transform: the fused re-order + twiddle kernel and the driver with the real
HipEngine at world_size 1 (the multi-rank collectives are covered on CPU over gloo; an 8-GPU run is the
driver's job). Needs an MI355X: `-m gpu`."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tf():
    import __graft_entry__ as g

    g.build()
    import tensor_fft_amd as t

    t.device_check(0)
    return t


def _c(re, im):
    return np.asarray(re, dtype=np.float64) + 1j * np.asarray(im, dtype=np.float64)


@pytest.mark.parametrize("a,b,c,n_tw,e0", [(4, 8, 16, 0, 0), (8, 32, 64, 1 << 14, 96), (2, 128, 8, 1 << 11, 0),
                                            (8, 1024, 1024, 1 << 26, 7 * 1024)])
def test_permute_twiddle_kernel(tf, a, b, c, n_tw, e0):
    import torch
    from tensor_fft_amd import capi

    rng = np.random.default_rng(a * b + c)
    re = rng.uniform(-1, 1, (a, b, c)).astype(np.float16)
    im = rng.uniform(-1, 1, (a, b, c)).astype(np.float16)
    d_re, d_im = torch.from_numpy(re).cuda().reshape(-1), torch.from_numpy(im).cuda().reshape(-1)
    o_re, o_im = torch.empty_like(d_re), torch.empty_like(d_im)
    capi.permute_twiddle(d_re, d_im, o_re, o_im, a, b, c, n_tw, e0)
    torch.cuda.synchronize()
    got = _c(o_re.cpu().numpy(), o_im.cpu().numpy()).reshape(b, a, c)
    want = _c(re, im).transpose(1, 0, 2)
    if n_tw:
        row = (e0 + np.arange(b, dtype=np.int64))[:, None, None]
        col = np.arange(a, dtype=np.int64)[None, :, None] * c + np.arange(c, dtype=np.int64)[None, None, :]
        want = want * np.exp(-2j * np.pi * ((row * col) % n_tw) / n_tw)
        assert np.abs(got - want).max() < 2.5e-3          # one fp16 rounding of |x w| <= 1.42
    else:
        assert np.array_equal(got, want)
    with pytest.raises(tf.TfftError):
        capi.permute_twiddle(d_re, d_im, d_re, d_im, a, b, c)       # in place is refused
    with pytest.raises(tf.TfftError):
        capi.load_library()  # keep the import used
        capi._check(capi.load_library().tfft_permute_twiddle(d_re.data_ptr(), d_im.data_ptr(), o_re.data_ptr(),
                                                             o_im.data_ptr(), a, b, 12, 0, 0, 0))


@pytest.mark.parametrize("lg,out_layout,fused", [(12, "transposed", False), (16, "transposed", False), (20, "transposed", False),
                                                 (20, "natural", False), (22, "natural", False), (23, "transposed", False),
                                                 (14, "transposed", True), (16, "transposed", True), (20, "transposed", True),
                                                 (21, "transposed", True), (21, "natural", True), (24, "transposed", True)])
def test_driver_with_hip_engine_single_rank(tf, orc, lg, out_layout, fused):
    """General form (balanced split, fused re-order + twiddle kernel) and fused form (one radix-256 / 512 column pass
    that applies the four-step twiddle itself) against the CPU oracle."""
    import torch
    from tensor_fft_amd.distributed import DistributedFFT1D, HipEngine

    n = 1 << lg
    rng = np.random.default_rng(lg)
    xr, xi = rng.uniform(-1, 1, n).astype(np.float16), rng.uniform(-1, 1, n).astype(np.float16)
    f = DistributedFFT1D(n, engine=HipEngine(0), input_layout="natural", output_layout=out_layout, fused=fused)
    assert f.fused == fused and (not fused or f.n1 in (256, 512))
    re, im = f.forward(torch.from_numpy(xr).cuda(), torch.from_numpy(xi).cuda())
    torch.cuda.synchronize()
    exact = _c(*orc.dft64(xr, xi))[0]
    got = _c(re.cpu().numpy(), im.cpu().numpy())
    want = exact[f.output_indices()]
    rel = np.linalg.norm(got - want) / np.linalg.norm(want)
    assert rel < 1.5e-3, rel
    # second call reuses every buffer (nothing is allocated per step) and gives the same bits
    keep_re = re.clone()
    ptr = re.data_ptr()
    re2, _ = f.forward(torch.from_numpy(xr).cuda(), torch.from_numpy(xi).cuda())
    torch.cuda.synchronize()
    assert re2.data_ptr() == ptr and bool((re2 == keep_re).all())


def test_single_gpu_2pow26_against_oracle(tf, orc):
    """BASELINE configs[4b] length on one GPU (world size 1): N = 2^26 = 256 x 2^18 through the fused four-step path
    (what each of 8 ranks would run around the exchange), full spectrum against the CPU oracle's fp64 FFT."""
    import torch
    from tensor_fft_amd.distributed import DistributedFFT1D, HipEngine

    n = 1 << 26
    xr, xi = orc.synth_uniform(n, 1, 0, 26)
    f = DistributedFFT1D(n, engine=HipEngine(0), input_layout="columns", output_layout="transposed")
    assert f.fused and f.n1 * f.n2 == n
    re, im = f.forward(torch.from_numpy(xr[0]).cuda(), torch.from_numpy(xi[0]).cuda())
    torch.cuda.synchronize()
    e_re, e_im = orc.dft64(xr, xi)
    want = (e_re[0] + 1j * e_im[0]).reshape(f.n2, f.n1).T.reshape(-1)          # X[k1 + N1 k2] stored [k1][k2]
    got = _c(re.cpu().numpy(), im.cpu().numpy())
    rel = np.linalg.norm(got - want) / np.linalg.norm(want)
    assert rel < 1.5e-3, rel
    # the library's own single-GPU plan of the same length agrees
    dev = torch.from_numpy(np.concatenate([xr[0], xi[0]])).cuda()
    out = torch.empty_like(dev)
    tf.TfftPlan(n, 1, 0, preserve_input=True).exec(dev, dev[n:], out, out[n:])
    torch.cuda.synchronize()
    o = out.cpu().numpy().astype(np.float64)
    nat = o[:n] + 1j * o[n:]
    assert np.linalg.norm(nat - (e_re[0] + 1j * e_im[0])) / np.linalg.norm(want) < 1.5e-3


# ---------------------------------------------------------------------------------------------------------------
# world size 2 / 4 with the REAL engine: every rank is a process of its own on this box's one GPU. RCCL needs one GPU per
# rank and gloo cannot send device tensors, so the exchange (and only the exchange) is staged through the host here; the
# column pass with its rank-dependent four-step twiddle, the re-order kernel with P > 1 peers, the row transforms and
# the index logic are the product's, on the device.
# ---------------------------------------------------------------------------------------------------------------
def _free_port():
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _rank_on_one_gpu(rank, world, port, n, in_layout, out_layout, fused, ret):
    import os

    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import __graft_entry__ as g

        g.build()
        from tensor_fft_amd.distributed import DistributedFFT1D, HipEngine

        class HostStagedExchange(DistributedFFT1D):
            def _exchange(self, re, im, role, out=None):
                p = self.world
                h_re, h_im = re.cpu(), im.cpu()
                o_re, o_im = torch.empty_like(h_re), torch.empty_like(h_im)
                chunk = h_re.numel() // p
                ops = []
                for q in range(p):
                    sl = slice(q * chunk, (q + 1) * chunk)
                    if q == self.rank:
                        o_re[sl].copy_(h_re[sl])
                        o_im[sl].copy_(h_im[sl])
                        continue
                    ops += [dist.P2POp(dist.isend, h_re[sl], q), dist.P2POp(dist.isend, h_im[sl], q),
                            dist.P2POp(dist.irecv, o_re[sl], q), dist.P2POp(dist.irecv, o_im[sl], q)]
                for req in dist.batch_isend_irecv(ops):
                    req.wait()
                if out is not None:                    # the tfft_dist_plan's own receive tensors
                    out[0].copy_(o_re)
                    out[1].copy_(o_im)
                    return out
                return o_re.to(re.device), o_im.to(im.device)

        rng = np.random.default_rng(99)               # the same signal on every rank
        xr, xi = rng.uniform(-1, 1, n).astype(np.float16), rng.uniform(-1, 1, n).astype(np.float16)
        f = HostStagedExchange(n, engine=HipEngine(0), input_layout=in_layout, output_layout=out_layout, fused=fused)
        assert f.fused == fused and f.world == world
        idx = f.input_indices()
        re, im = f.forward(torch.from_numpy(xr[idx].copy()).cuda(), torch.from_numpy(xi[idx].copy()).cuda())
        torch.cuda.synchronize()
        exact = np.fft.fft(xr.astype(np.float64) + 1j * xi.astype(np.float64)) / n
        got = _c(re.cpu().numpy(), im.cpu().numpy())
        want = exact[f.output_indices()]
        ret[rank] = float(np.linalg.norm(got - want) / np.linalg.norm(want))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,lg,in_layout,out_layout,fused", [
    (2, 20, "columns", "transposed", True),      # 256 x 4096, C = 2048 columns per rank
    (2, 22, "natural", "natural", True),
    (4, 21, "columns", "transposed", True),
    (2, 20, "columns", "transposed", False),
    (4, 22, "natural", "natural", False),
])
def test_driver_with_hip_engine_multi_rank_on_one_gpu(tf, world, lg, in_layout, out_layout, fused):
    import torch.multiprocessing as mp

    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_rank_on_one_gpu, args=(world, _free_port(), 1 << lg, in_layout, out_layout, fused, ret), nprocs=world, join=True)
    assert len(ret) == world
    for rank in range(world):
        assert ret[rank] < 1.5e-3, (rank, ret[rank])
