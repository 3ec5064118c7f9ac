"""The reference's two accuracy sweeps (src/testing/benchmarks/AccuracyTest.cu:17-86: error vs N, N = 2^8 .. 2^28, 256
harmonics, weight seeds 42 / 42^2 = 1764; AccuracyTestBandwidth.cu:17-87: error vs signal bandwidth at N = 2^20, frequency
cutoff 1 .. N/2) as importable pieces: tests/test_gpu_accuracy_sweeps.py asserts them on a thinned grid, tools/accuracy_sweep.py
writes the full tables under profiles/. Test infrastructure: uses the CPU oracle.

Per point: the reference's test signal (sine superposition, TestingDataCreation.h:89-117) as binary16 planes, its exact spectrum
DFT(x)/N in fp64, and (max, mean, sigma) of |delta| over the 2N reals (AccuracyCalculator.h:86-148) for
   ours        this library on the MI355X
   reference   the oracle's fp16 restatement of the reference CUDA kernels (N <= 2^20)
   vendor      hipFFT in fp32 / fp16 through torch.fft (the reference compares with cuFFT float / half)

Exact spectrum: the CPU oracle's fp64 FFT up to 2^24; beyond that hipFFT Z2Z on the device (what the reference itself uses as
its oracle: cuFFT Z2Z / N, CuFFTTest.h:218-261), cross-checked here against direct fp64 DFT sums at sampled bins.
Signal: the CPU oracle's generator while N * cutoff <= 2^32, the same formula on the device beyond (fp64 phase, cast to fp32,
fp32 sine, float weight times float sine accumulated in fp64: TestingDataCreation.h:97-115)."""
import numpy as np

# thresholds of the reference's unit test on DFT(x)/N (src/testing/unitTesting/UnitTest.cu:14-16)
MEAN_MAX, SIGMA_MAX, MAX_MAX = 1e-3, 1e-2, 0.5


def weights(orc, count=1 << 20):
    return orc.random_weights(count, 42), orc.random_weights(count, 42 * 42)


def make_signal(torch, orc, n, cutoff, w_re, w_im):
    """binary16 (re, im) numpy planes of the reference test signal."""
    if n * cutoff <= (1 << 31):
        return orc.sine_superposition(n, w_re, w_im, cutoff)
    wr = torch.from_numpy(np.ascontiguousarray(w_re[:cutoff], dtype=np.float32)).cuda()
    wi = torch.from_numpy(np.ascontiguousarray(w_im[:cutoff], dtype=np.float32)).cuda()
    acc_re = torch.zeros(n, dtype=torch.float64, device="cuda")
    acc_im = torch.zeros(n, dtype=torch.float64, device="cuda")
    step = max(1, (1 << 27) // min(n, 1 << 27))                 # harmonics per block: <= 2^27 phases at a time
    rows = min(n, 1 << 27)
    for t0 in range(0, n, rows):
        t = torch.arange(t0, t0 + rows, dtype=torch.float64, device="cuda")
        for f0 in range(0, cutoff, step):
            f = torch.arange(f0, min(cutoff, f0 + step), dtype=torch.float64, device="cuda")
            s = torch.sin(((2 * np.pi) * f[:, None] * t[None, :] / n).float())          # [harmonics][samples], fp32 sine of an fp32 phase
            acc_re[t0:t0 + rows] += (wr[f0:f0 + f.numel(), None] * s).double().sum(0)
            acc_im[t0:t0 + rows] += (wi[f0:f0 + f.numel(), None] * s).double().sum(0)
    return acc_re.half().cpu().numpy(), acc_im.half().cpu().numpy()


def direct_bins(torch, d_re, d_im, bins):
    """DFT(x)/N at the given bins by direct fp64 summation on the device (d_re, d_im: fp64 CUDA planes)."""
    n = d_re.numel()
    t = torch.arange(n, device="cuda", dtype=torch.float64)
    out = []
    for k in bins:
        ph = ((t * float(k)) % n) * (-2.0 * np.pi / n)
        c, s = torch.cos(ph), torch.sin(ph)
        out.append(complex(float((d_re * c - d_im * s).sum()) / n, float((d_re * s + d_im * c).sum()) / n))
    return np.array(out)


def exact_spectrum(torch, orc, re, im):
    """fp64 DFT(x)/N of binary16 planes -> (re, im) float64 numpy arrays."""
    n = re.size
    if n <= (1 << 24):
        e_re, e_im = orc.dft64(re, im)
        return e_re[0], e_im[0]
    d_re, d_im = torch.from_numpy(re).cuda().double(), torch.from_numpy(im).cuda().double()
    z = torch.fft.fft(torch.complex(d_re, d_im)) / n
    rng = np.random.default_rng(n)
    bins = [1, 2, 255, n - 1, n - 255, n // 2] + [int(b) for b in rng.integers(0, n, 10)]
    want = direct_bins(torch, d_re, d_im, bins)
    got = z[torch.tensor(bins, device="cuda")].cpu().numpy()
    assert np.abs(got - want).max() < 1e-9, "hipFFT Z2Z disagrees with the direct fp64 sums"
    z = z.cpu().numpy()
    return np.ascontiguousarray(z.real), np.ascontiguousarray(z.imag)


def run_point(torch, tf, orc, n, cutoff, w_re, w_im, vendor=True):
    """{column: (max, mean, sigma)} for one (N, cutoff)."""
    re, im = make_signal(torch, orc, n, cutoff, w_re, w_im)
    ex_re, ex_im = exact_spectrum(torch, orc, re, im)

    def stats(g_re, g_im):
        return orc.deviation_stats(np.asarray(g_re, dtype=np.float64), np.asarray(g_im, dtype=np.float64), ex_re, ex_im)

    dev = torch.from_numpy(np.concatenate([re, im])).cuda()
    out = torch.empty_like(dev)
    tf.TfftPlan(n, 1, 0, preserve_input=True).exec(dev, dev[n:], out, out[n:])
    torch.cuda.synchronize()
    o = out.cpu().numpy()
    res = {"ours": stats(o[:n], o[n:])}
    del out
    if n <= (1 << 20):
        r = orc.ref_fft(re, im, orc.MODE_4096 if n >= 4096 else orc.MODE_256)
        res["reference"] = stats(r[0][0], r[1][0])
    if vendor:
        z = torch.complex(dev[:n].float(), dev[n:].float())
        f32 = (torch.fft.fft(z) / n).cpu().numpy()
        res["vendor_fp32"] = stats(f32.real, f32.imag)
        del z
        try:
            f16 = torch.fft.fft(torch.complex(dev[:n], dev[n:]))          # complex32
            f16 = (torch.view_as_real(f16).float() / n).cpu().numpy()
            res["vendor_fp16"] = stats(f16[:, 0], f16[:, 1])
        except Exception:                                   # noqa: BLE001  (half FFT unavailable in this build)
            pass
    torch.cuda.empty_cache()
    return res
