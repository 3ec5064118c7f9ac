"""Host-side checks of the distributed-transform ABI (tfft_dist_*, include/tfft.h) that need no GPU: the four-step split the
library chooses, its agreement with the Python driver's own index logic, and the re-order-free row pass of BASELINE
configs[4b] (single N = 2^26 over 2 / 4 / 8 ranks)."""
import numpy as np
import pytest

import tensor_fft_amd as tf
from tensor_fft_amd import capi
from tensor_fft_amd.distributed import DistributedFFT1D


@pytest.fixture(scope="module", autouse=True)
def built():
    import __graft_entry__ as g

    g.build()


def test_geometry_of_configs_4b():
    for world in (1, 2, 4, 8):
        g = capi.dist_geometry(1 << 26, world, world - 1)
        assert (g.n, g.n1, g.n2) == (1 << 26, 256, 1 << 18) and g.world == world and g.rank == world - 1
        assert g.cols * world == g.n2 and g.rows * world == g.n1 and g.chunk == g.rows * g.cols
        # no re-order pass between the exchange and the row transforms: column pass + two row passes
        assert g.reorder == 0 and g.local_passes == 3 and g.fused == 1
    assert capi.dist_geometry(1 << 26, 8).chunk * 2 == 2 * 1024 * 1024        # 2 MiB per plane and peer pair... in halves: 1 Mi
    # lengths whose row transform is one LDS-resident kernel keep the re-order pass (documented in tfft.h)
    g = capi.dist_geometry(1 << 20, 8)
    assert (g.n1, g.n2, g.reorder, g.local_passes) == (256, 4096, 1, 3)
    assert capi.dist_geometry(1 << 20, 1).reorder == 0


def test_geometry_rejections():
    for n, world, rank in ((1 << 14, 2, 0), (1 << 16, 8, 0), (3000, 2, 0), (1 << 20, 3, 0), (1 << 20, 2, 2), (1 << 20, 0, 0)):
        with pytest.raises(tf.TfftError):
            capi.dist_geometry(n, world, rank)


class _GeometryOnlyEngine:
    """What DistributedFFT1D asks an engine about when it picks the fused split (same rule as HipEngine)."""

    def supports_fourstep(self, n1, inner):
        return n1 in (256, 512) and inner >= 64 and inner % 64 == 0


def test_python_driver_and_c_abi_choose_the_same_split(monkeypatch):
    import torch.distributed as dist

    for world in (1, 2, 4, 8):
        monkeypatch.setattr(dist, "is_initialized", lambda: True)
        monkeypatch.setattr(dist, "get_world_size", lambda group=None, w=world: w)
        monkeypatch.setattr(dist, "get_rank", lambda group=None: 0)
        for lg in range(14, 31):
            n = 1 << lg
            try:
                g = capi.dist_geometry(n, world, 0)
            except tf.TfftError:
                g = None
            f = DistributedFFT1D(n, engine=_GeometryOnlyEngine())
            if g is None:
                assert not f.fused, (lg, world)
                continue
            assert f.fused and (f.n1, f.n2, f.c, f.k) == (g.n1, g.n2, g.cols, g.rows), (lg, world)
            # the chunk rank q receives from rank p' holds rows k of ITS block, columns of p': what the row pass reads as
            # segment p' of row k (sample index p' C + c), i.e. the layouts of the two sides agree
            if lg <= 20:
                idx = f.output_indices(0)
                assert idx[0] == 0 and idx[1] == g.n1 and idx[g.n2] == 1


def test_exec_entry_points_fail_cleanly_without_a_plan():
    L = capi.load_library()
    assert L.tfft_dist_exec_pre(None, None, None, None) == 5
    assert L.tfft_dist_exec(None, None, None, None, None, None) == 5
    assert "null plan" in capi.last_error()
    assert L.tfft_dist_comm_destroy(None) == 0
