"""A/B of plan options in ONE process, interleaved (box-to-box spread is +-4 %, so variants are only comparable inside a process):
    python tools/ab_variants.py N:batch[:inner] [lib=NAME:]VARIANT[,launch_iters] ... [--order transposed] [--reps 10] [--rounds 5]
(lib=NAME: the build build/libtfft_NAME.so made by tools/build_ab.py instead of the shipped library)
Prints per variant the median over rounds of the mean launch time, Gsamples/s and GB/s per pass."""
import argparse
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import __graft_entry__ as g  # noqa: E402

g.build()
import tensor_fft_amd as tf  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("shape")
ap.add_argument("variants", nargs="+")
ap.add_argument("--order", default="natural")
ap.add_argument("--in-order", default="natural")
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--rounds", type=int, default=5)
args = ap.parse_args()
if args.shape.startswith("2d:"):
    # 2D 4096 x 4096 x batch through tfft_plan2d_* of each named build: python tools/ab_variants.py 2d:64 0 lib=base:0
    images, n = int(args.shape[3:]), 4096
    half = images * n * n
    x = ((torch.rand(2 * half, device="cuda") * 2 - 1)).half()
    y = torch.empty_like(x)
    plans2 = []
    for spec in args.variants:
        mod = tf
        if spec.startswith("lib="):
            mod = None
    import importlib.util

    def capi2(libname):
        path = os.path.join(ROOT, "tensor-fft_amd", "capi.py")
        sp = importlib.util.spec_from_file_location("capi_" + libname, path)
        m = importlib.util.module_from_spec(sp)
        sp.loader.exec_module(m)
        m._LIB_NAME = os.path.join(ROOT, "build", f"libtfft_{libname}.so")     # an absolute path wins over the package directory
        return m

    for spec in args.variants:
        mod = capi2(spec[4:].split(":")[0]) if spec.startswith("lib=") else tf
        p = mod.TfftPlan2D(n, n, images, 0)
        p.set_workspace(torch.empty(p.workspace_bytes // 2, dtype=torch.float16, device="cuda"))
        plans2.append((spec, p))
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.08:
        plans2[0][1].exec(x[:half], x[half:], y[:half], y[half:])
        torch.cuda.synchronize()
    res = {spec: [] for spec, _ in plans2}
    for _ in range(args.rounds):
        for spec, p in plans2:
            p.exec(x[:half], x[half:], y[:half], y[half:])
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.reps):
                p.exec(x[:half], x[half:], y[:half], y[half:])
            e1.record()
            torch.cuda.synchronize()
            res[spec].append(e0.elapsed_time(e1) / args.reps)
    for spec, p in plans2:
        ms = statistics.median(res[spec])
        print(f"2D 4096x4096 x {images} {spec:>12s}: {ms:.3f} ms (min {min(res[spec]):.3f}, max {max(res[spec]):.3f})  {half/ms/1e6:7.1f} Gsamples/s", flush=True)
    # builds that only differ in scheduling must agree bit for bit
    ref = None
    for spec, p in plans2:
        y.fill_(float("nan"))
        p.exec(x[:half], x[half:], y[:half], y[half:])
        torch.cuda.synchronize()
        if ref is None:
            ref = y.clone()
        else:
            same = bool((ref.view(torch.int16) == y.view(torch.int16)).all())
            print(f"   {spec}: output {'bit-identical to' if same else 'DIFFERS from'} {plans2[0][0]}", flush=True)
    sys.exit(0)
f = [int(v) for v in args.shape.split(":")]
n, b, inner = f[0], f[1], (f[2] if len(f) > 2 else 1)
nf = n * inner
x = ((torch.rand(b * 2 * nf, device="cuda") * 2 - 1)).half()
y = torch.empty_like(x)
plans = []
_capis = {}


def capi_for(libname):
    """A second instance of the ctypes binding bound to another build of the library (tools/build_ab.py)."""
    import importlib.util

    if libname not in _capis:
        path = os.path.join(ROOT, "tensor-fft_amd", "capi.py")
        sp = importlib.util.spec_from_file_location("capi_" + libname, path)
        m = importlib.util.module_from_spec(sp)
        sp.loader.exec_module(m)
        m._LIB_NAME = os.path.join(ROOT, "build", f"libtfft_{libname}.so")     # an absolute path wins over the package directory
        _capis[libname] = m
    return _capis[libname]


for spec in args.variants:
    mod, rest = tf, spec
    if spec.startswith("lib="):
        libname, rest = spec[4:].split(":")
        mod = capi_for(libname)
    t = rest.split(",")
    v, it = int(t[0]), (int(t[1]) if len(t) > 1 else 0)
    p = mod.TfftPlan(n, b, 0, inner=inner, variant=v, launch_iters=it, preserve_input=True, output_order=args.order,
                     input_order=args.in_order)
    ws = torch.empty(max(1, p.workspace_bytes // 2), dtype=torch.float16, device="cuda")
    if p.workspace_bytes:
        p.set_workspace(ws)
    plans.append((spec, p, ws))
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.08:          # clock ramp
    plans[0][1].exec(x, x[nf:], y, y[nf:])
    torch.cuda.synchronize()
res = {spec: [] for spec, _, _ in plans}
for _ in range(args.rounds):
    for spec, p, _ in plans:
        p.exec(x, x[nf:], y, y[nf:])
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps):
            p.exec(x, x[nf:], y, y[nf:])
        e1.record()
        torch.cuda.synchronize()
        res[spec].append(e0.elapsed_time(e1) / args.reps)
for spec, p, _ in plans:
    ms = statistics.median(res[spec])
    print(f"N={n} batch={b} inner={inner} variant={spec:>12s} launches={p.num_launches}: {ms*1e3:9.1f} us (min {min(res[spec])*1e3:.1f}, max {max(res[spec])*1e3:.1f})  "
          f"{nf*b/ms/1e6:7.1f} Gsamples/s  {p.algorithmic_bytes/ms/1e6/p.num_launches:7.0f} GB/s per pass", flush=True)
