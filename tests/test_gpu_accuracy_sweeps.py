"""SURVEY 8f-2 as a gated test: the reference's two accuracy sweeps on a thinned grid, at the
reference's range (N up to 2^28; cutoff up to N/4 at N = 2^20). Per point: the reference's own acceptance thresholds
(UnitTest.cu:14-16) and, where the restatement of the reference kernels runs (N <= 2^20), max|error| of this library no larger
than the restatement's. The full tables: tools/accuracy_sweep.py -> profiles/r3_accuracy_vs_*.dat."""
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env(orc):
    import torch

    import __graft_entry__ as g

    g.build()
    import tensor_fft_amd as tf

    tf.device_check(0)
    import accuracy_protocol as ap

    return torch, tf, ap, ap.weights(orc)


def _assert_point(ap, res, tag):
    mx, mean, sigma = res["ours"]
    assert mean <= ap.MEAN_MAX and sigma <= ap.SIGMA_MAX and mx <= ap.MAX_MAX, (tag, res["ours"])
    if "reference" in res:
        assert mx <= 1.05 * res["reference"][0] + 1e-7, (tag, mx, res["reference"][0])


@pytest.mark.parametrize("lg", [8, 9, 11, 12, 13, 16, 17, 20, 21, 22, 24, 26, 28])
def test_error_vs_n(env, orc, lg):
    """AccuracyTest.cu:17-86: N = 2^8 .. 2^28, 256 harmonics (N/2 below 512), seeds 42 / 1764."""
    torch, tf, ap, (w_re, w_im) = env
    n = 1 << lg
    res = ap.run_point(torch, tf, orc, n, min(256, n // 2), w_re, w_im, vendor=False)
    _assert_point(ap, res, f"N=2^{lg}")


@pytest.mark.parametrize("cutoff", [1, 2, 16, 256, 4096, 65536, 1 << 18])
def test_error_vs_bandwidth(env, orc, cutoff):
    """AccuracyTestBandwidth.cu:17-87: N = 2^20, frequency cutoff from 1 towards N/2."""
    torch, tf, ap, (w_re, w_im) = env
    res = ap.run_point(torch, tf, orc, 1 << 20, cutoff, w_re, w_im, vendor=False)
    _assert_point(ap, res, f"cutoff={cutoff}")
