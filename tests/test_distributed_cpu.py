"""The N > 1 paths on CPU: world_size 2 / 4 over gloo.

(1) the single distributed transform (tensor-fft_amd/distributed.py): its index logic and its collectives run
    unchanged; only the local arithmetic is swapped for a numpy engine defined HERE (test infrastructure), so
    what is verified is the decomposition, the layouts and the all-to-all plumbing;
(2) the batch-sharded benchmark contract: every rank owns an equal, disjoint slice and no collective is needed.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


class NumpyEngine:
    """fp64 numpy stand-in for HipEngine with the same interface (results rounded to fp16 like the GPU's)."""

    @staticmethod
    def _c(re, im):
        return re.numpy().astype(np.float64) + 1j * im.numpy().astype(np.float64)

    @staticmethod
    def _split(z):
        return (torch.from_numpy(np.ascontiguousarray(z.real).astype(np.float16).reshape(-1)),
                torch.from_numpy(np.ascontiguousarray(z.imag).astype(np.float16).reshape(-1)))

    def fft_strided(self, re, im, n, inner):
        z = self._c(re, im).reshape(n, inner)
        return self._split(np.fft.fft(z, axis=0) / n)

    def fft_rows(self, re, im, n, batch):
        z = self._c(re, im).reshape(batch, n)
        return self._split(np.fft.fft(z, axis=1) / n)

    def supports_fourstep(self, n1, inner):
        return n1 in (256, 512) and inner % 8 == 0          # (the GPU engine needs >= 64 columns; the logic is the same)

    def fft_strided_fourstep(self, re, im, n1, inner, n_total, col0):
        z = np.fft.fft(self._c(re, im).reshape(n1, inner), axis=0) / n1
        k = np.arange(n1)[:, None]
        col = col0 + np.arange(inner)[None, :]
        return self._split(z * np.exp(-2j * np.pi * ((k * col) % n_total) / n_total))

    def permute_twiddle(self, re, im, a, b, c, n_tw=0, e0=0, role=None):
        z = self._c(re, im).reshape(a, b, c).transpose(1, 0, 2)
        if n_tw:
            row = (e0 + np.arange(b))[:, None, None]
            col = (np.arange(a)[None, :, None] * c + np.arange(c)[None, None, :])
            z = z * np.exp(-2j * np.pi * ((row * col) % n_tw) / n_tw)
        return self._split(z)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, n, in_layout, out_layout, fused, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import tensor_fft_amd  # noqa: F401
        from tensor_fft_amd.distributed import DistributedFFT1D

        rng = np.random.default_rng(1234)            # same signal on every rank
        x = (rng.uniform(-1, 1, n) + 1j * rng.uniform(-1, 1, n))
        xr, xi = x.real.astype(np.float16), x.imag.astype(np.float16)
        f = DistributedFFT1D(n, engine=NumpyEngine(), input_layout=in_layout, output_layout=out_layout, fused=fused)
        assert f.fused == fused
        idx = f.input_indices()
        re, im = f.forward(torch.from_numpy(xr[idx].copy()), torch.from_numpy(xi[idx].copy()))
        exact = np.fft.fft(xr.astype(np.float64) + 1j * xi.astype(np.float64)) / n
        got = re.numpy().astype(np.float64) + 1j * im.numpy().astype(np.float64)
        want = exact[f.output_indices()]
        err = np.linalg.norm(got - want) / np.linalg.norm(want)
        # every output index is owned exactly once
        owned = torch.zeros(n, dtype=torch.int32)
        owned[torch.from_numpy(f.output_indices())] += 1
        dist.all_reduce(owned)
        ret[rank] = (float(err), bool((owned == 1).all()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n,in_layout,out_layout,fused", [
    (2, 1 << 10, "columns", "transposed", False),
    (2, 1 << 13, "columns", "transposed", False),
    (4, 1 << 12, "columns", "transposed", False),
    (2, 1 << 12, "natural", "transposed", False),
    (2, 1 << 12, "columns", "natural", False),
    (4, 1 << 14, "natural", "natural", False),
    # fused form: N1 = 256, the four-step twiddle applied by the column pass, the unpack a pure re-order
    (2, 1 << 13, "columns", "transposed", True),
    (4, 1 << 14, "columns", "transposed", True),
    (4, 1 << 16, "natural", "natural", True),
    (2, 1 << 15, "natural", "transposed", True),
])
def test_distributed_fft_over_gloo(world, n, in_layout, out_layout, fused):
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), n, in_layout, out_layout, fused, ret), nprocs=world, join=True)
    assert len(ret) == world
    for rank in range(world):
        err, partition_ok = ret[rank]
        assert partition_ok
        assert err < 1.5e-3, (rank, err)      # three fp16 roundings of intermediates


def test_single_rank_layouts():
    """world_size 1 (no process group): the driver degenerates to the plain four-step transform."""
    import tensor_fft_amd  # noqa: F401
    from tensor_fft_amd.distributed import DistributedFFT1D

    n = 1 << 11
    rng = np.random.default_rng(5)
    xr, xi = rng.uniform(-1, 1, n).astype(np.float16), rng.uniform(-1, 1, n).astype(np.float16)
    exact = np.fft.fft(xr.astype(np.float64) + 1j * xi.astype(np.float64)) / n
    for out_layout, fused in (("transposed", False), ("natural", False), ("transposed", True), ("natural", True)):
        f = DistributedFFT1D(n, engine=NumpyEngine(), input_layout="natural", output_layout=out_layout, fused=fused)
        assert np.array_equal(f.input_indices(), np.arange(n)) and f.fused == fused and (not fused or f.n1 == 256)
        re, im = f.forward(torch.from_numpy(xr.copy()), torch.from_numpy(xi.copy()))
        got = re.numpy().astype(np.float64) + 1j * im.numpy().astype(np.float64)
        want = exact[f.output_indices()]
        assert np.linalg.norm(got - want) / np.linalg.norm(want) < 1.5e-3


def _shard_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import bench

        lo, hi = bench.shard(rank, world, 1000)
        t = torch.zeros(1000, dtype=torch.int32)
        t[lo:hi] = 1
        dist.all_reduce(t)
        # the timing reduction bench.py uses: max over ranks
        w = torch.tensor([0.5 + rank], dtype=torch.float64)
        dist.all_reduce(w, op=dist.ReduceOp.MAX)
        ret[rank] = (bool((t == 1).all()), float(w[0]), hi - lo)
    finally:
        dist.destroy_process_group()


def test_batch_sharding_contract_over_gloo():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_shard_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    for rank in range(world):
        covered, tmax, count = ret[rank]
        assert covered and tmax == 0.5 + (world - 1) and count == 500
