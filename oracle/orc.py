"""ctypes view of oracle/libtfft_oracle.so.

TEST INFRASTRUCTURE ONLY (see tfft_oracle.cpp): importable from tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg, never from the
tensor-fft_amd package.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libtfft_oracle.so")

MODE_256 = 0
MODE_4096 = 1


def build(force=False):
    src = os.path.join(_HERE, "tfft_oracle.cpp")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        u16p = ctypes.POINTER(ctypes.c_uint16)
        f64p = ctypes.POINTER(ctypes.c_double)
        f32p = ctypes.POINTER(ctypes.c_float)
        ip = ctypes.POINTER(ctypes.c_int)
        u64 = ctypes.c_uint64
        L.orc_f64_to_f16.restype = ctypes.c_uint16
        L.orc_f64_to_f16.argtypes = [ctypes.c_double]
        L.orc_f16_to_f64.restype = ctypes.c_double
        L.orc_f16_to_f64.argtypes = [ctypes.c_uint16]
        L.orc_num_threads.restype = ctypes.c_int
        L.orc_ref_plan.restype = ctypes.c_int
        L.orc_ref_plan.argtypes = [u64, ctypes.c_int, ip, ip, ip]
        L.orc_ref_gather_index.restype = u64
        L.orc_ref_gather_index.argtypes = [u64, ctypes.c_int, ctypes.c_int]
        L.orc_ref_compute_fft.restype = ctypes.c_int
        L.orc_ref_compute_fft.argtypes = [u64, ctypes.c_int, u16p, ip]
        L.orc_ref_fft.restype = ctypes.c_int
        L.orc_ref_fft.argtypes = [u64, u64, ctypes.c_int, u16p, u16p, u64, u16p, u16p, u64]
        L.orc_dft64.restype = ctypes.c_int
        L.orc_dft64.argtypes = [u64, u64, u16p, u16p, u64, f64p, f64p, u64, ctypes.c_int, ctypes.c_int]
        L.orc_fft64_rows.restype = ctypes.c_int
        L.orc_fft64_rows.argtypes = [u64, u64, f64p, f64p, u64]
        L.orc_synth_uniform.restype = None
        L.orc_synth_uniform.argtypes = [u64, u64, u64, u64, u16p]
        L.orc_random_weights.restype = None
        L.orc_random_weights.argtypes = [ctypes.c_int, ctypes.c_int, f32p]
        L.orc_sine_superposition.restype = None
        L.orc_sine_superposition.argtypes = [u64, f32p, f32p, ctypes.c_int, u16p]
        L.orc_deviation_stats.restype = None
        L.orc_deviation_stats.argtypes = [f64p, f64p, u64, f64p, f64p, f64p]
        _lib = L
    return _lib


def _u16(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint16))


def _f64(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def _f32(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _as_bits(x):
    """float16 ndarray (or uint16 bits) -> contiguous uint16 bits, shape (batch, n)."""
    x = np.asarray(x)
    if x.dtype == np.float16:
        x = x.view(np.uint16)
    assert x.dtype == np.uint16, x.dtype
    if x.ndim == 1:
        x = x[None, :]
    return np.ascontiguousarray(x)


def num_threads():
    return lib().orc_num_threads()


def ref_plan(n, mode=MODE_256):
    """(r16_steps, r2_steps, results_in_results) of the reference's CreatePlan, or None."""
    r16, r2, rir = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    rc = lib().orc_ref_plan(n, mode, ctypes.byref(r16), ctypes.byref(r2), ctypes.byref(rir))
    if rc:
        return None
    return r16.value, r2.value, bool(rir.value)


def ref_gather_index(o, r16, r2):
    return lib().orc_ref_gather_index(o, r16, r2)


def ref_fft(re, im, mode=MODE_4096):
    """fp16 restatement of the reference kernels. re/im: float16 (batch, n). Returns float16 pair."""
    rb, ib = _as_bits(re), _as_bits(im)
    batch, n = rb.shape
    orr = np.empty_like(rb)
    oi = np.empty_like(ib)
    rc = lib().orc_ref_fft(n, batch, mode, _u16(rb), _u16(ib), n, _u16(orr), _u16(oi), n)
    if rc:
        raise ValueError(f"orc_ref_fft rc={rc} (n={n}, mode={mode})")
    return orr.view(np.float16), oi.view(np.float16)


def dft64(re, im, algo=1, threads=0, out=None):
    """fp64 DFT(x)/N of fp16 planar input. algo 0 naive, 1 radix-2 FFT. out: optional preallocated (re, im) float64
    pair of shape (batch, n) (timing loops reuse it, so that page-faulting fresh output arrays is not what they time)."""
    rb, ib = _as_bits(re), _as_bits(im)
    batch, n = rb.shape
    if out is not None:
        orr, oi = out
        assert orr.shape == (batch, n) and oi.shape == (batch, n) and orr.dtype == np.float64 and oi.dtype == np.float64
    else:
        orr = np.empty((batch, n), dtype=np.float64)
        oi = np.empty((batch, n), dtype=np.float64)
    rc = lib().orc_dft64(n, batch, _u16(rb), _u16(ib), n, _f64(orr), _f64(oi), n, algo, threads)
    if rc:
        raise ValueError(f"orc_dft64 rc={rc}")
    return orr, oi


def fft64_rows(z):
    """fp64 DFT/N along the last axis of a complex128 array (oracle radix-2 FFT); returns a new array."""
    z = np.asarray(z, dtype=np.complex128)
    n = z.shape[-1]
    re = np.ascontiguousarray(z.real).reshape(-1, n)
    im = np.ascontiguousarray(z.imag).reshape(-1, n)
    rc = lib().orc_fft64_rows(n, re.shape[0], _f64(re), _f64(im), n)
    if rc:
        raise ValueError(f"orc_fft64_rows rc={rc}")
    return (re + 1j * im).reshape(z.shape)


def synth_uniform(n, batch, first_fft=0, seed=42):
    """CPU twin of tfft_synth_uniform: float16 (re, im), each (batch, n), of transforms first_fft .. first_fft + batch - 1."""
    out = np.empty((batch, 2, n), dtype=np.uint16)
    lib().orc_synth_uniform(n, batch, first_fft, seed, _u16(out))
    h = out.view(np.float16)
    return np.ascontiguousarray(h[:, 0]), np.ascontiguousarray(h[:, 1])


def random_weights(count, seed):
    out = np.empty(count, dtype=np.float32)
    lib().orc_random_weights(count, seed, _f32(out))
    return out


def sine_superposition(n, w_re, w_im, cutoff=None):
    """The reference test signal; returns float16 (re, im) each of length n."""
    w_re = np.ascontiguousarray(w_re, dtype=np.float32)
    w_im = np.ascontiguousarray(w_im, dtype=np.float32)
    if cutoff is None:
        cutoff = len(w_re)
    assert cutoff <= len(w_re) and cutoff <= len(w_im)
    out = np.empty(2 * n, dtype=np.uint16)
    lib().orc_sine_superposition(n, _f32(w_re), _f32(w_im), cutoff, _u16(out))
    h = out.view(np.float16)
    return h[:n].copy(), h[n:].copy()


def deviation_stats(a_re, a_im, b_re, b_im):
    """(max, mean, sigma) of |a-b| over the 2N reals of one FFT, as AccuracyCalculator.h."""
    a = np.ascontiguousarray(np.concatenate([np.ravel(a_re), np.ravel(a_im)]), dtype=np.float64)
    b = np.ascontiguousarray(np.concatenate([np.ravel(b_re), np.ravel(b_im)]), dtype=np.float64)
    mx, av, sg = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
    lib().orc_deviation_stats(_f64(a), _f64(b), a.size, ctypes.byref(mx), ctypes.byref(av), ctypes.byref(sg))
    return mx.value, av.value, sg.value
