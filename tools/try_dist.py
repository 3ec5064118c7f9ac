"""First contact of the tfft_dist_* path with a GPU: emulated ranks in one process, then the RCCL self-exchange."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g

g.build()
import tensor_fft_amd as tf
from tensor_fft_amd import capi

if os.environ.get("TFFT_AB_LIB"):            # another build of the library (tools/build_ab.py), for same-box comparisons
    capi._LIB_NAME = os.path.abspath(os.environ["TFFT_AB_LIB"])
    capi._lib = None                            # (g.build() above has already loaded the default build)
what = sys.argv[1] if len(sys.argv) > 1 else "emu"


def emulate(lg, world):
    n = 1 << lg
    rng = np.random.default_rng(lg * 10 + world)
    xr, xi = rng.uniform(-1, 1, n).astype(np.float16), rng.uniform(-1, 1, n).astype(np.float16)
    plans = [capi.DistPlan(n, world, r, 0) for r in range(world)]
    geo = plans[0].geometry
    n1, n2, c, k, chunk = int(geo.n1), int(geo.n2), int(geo.cols), int(geo.rows), int(geo.chunk)
    loc = n // world
    mk = lambda: torch.empty(loc, dtype=torch.float16, device="cuda")
    bufs = []
    for r, p in enumerate(plans):
        b = (mk(), mk(), mk(), mk()) if world > 1 else None
        if b is None:
            s0, s1 = mk(), mk()
            b = (s0, s1, s0, s1)
        p.set_buffers(*b)
        bufs.append(b)
    outs = []
    x2r, x2i = xr.reshape(n1, n2), xi.reshape(n1, n2)
    for r, p in enumerate(plans):
        ir = torch.from_numpy(np.ascontiguousarray(x2r[:, r * c:(r + 1) * c]).reshape(-1)).cuda()
        ii = torch.from_numpy(np.ascontiguousarray(x2i[:, r * c:(r + 1) * c]).reshape(-1)).cuda()
        p.pre(ir, ii)
    torch.cuda.synchronize()
    if world > 1:
        for q in range(world):
            for pp in range(world):
                bufs[q][2][pp * chunk:(pp + 1) * chunk].copy_(bufs[pp][0][q * chunk:(q + 1) * chunk])
                bufs[q][3][pp * chunk:(pp + 1) * chunk].copy_(bufs[pp][1][q * chunk:(q + 1) * chunk])
    for r, p in enumerate(plans):
        o = (mk(), mk())
        p.post(*o)
        outs.append(o)
    torch.cuda.synchronize()
    exact = np.fft.fft(xr.astype(np.float64) + 1j * xi.astype(np.float64)) / n
    worst = 0.0
    for r in range(world):
        got = outs[r][0].cpu().numpy().astype(np.float64) + 1j * outs[r][1].cpu().numpy().astype(np.float64)
        k1 = r * k + np.arange(k)[:, None]
        k2 = np.arange(n2)[None, :]
        want = exact[(k1 + n1 * k2).reshape(-1)]
        worst = max(worst, float(np.linalg.norm(got - want) / np.linalg.norm(want)))
    print(f"emulated N=2^{lg} world={world}: n1={n1} n2={n2} C={c} K={k} reorder={geo.reorder} passes={geo.local_passes} rel-L2 {worst:.2e}",
          flush=True)
    assert worst < 1.5e-3
    # local time of rank 0 (pre + post, no exchange)
    p = plans[0]
    ir, ii = mk(), mk()
    o = outs[0]
    for _ in range(5):
        p.pre(ir, ii); p.post(*o)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        p.pre(ir, ii); p.post(*o)
    e1.record()
    torch.cuda.synchronize()
    print(f"    rank 0 local work: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us per transform", flush=True)


def legacy(lg, world):
    """Round 2's local work of rank 0 for the same geometry: column pass, re-order pass, row transforms (four passes at 2^26)."""
    n = 1 << lg
    geo = capi.dist_geometry(n, world, 0)
    n1, n2, c, k = int(geo.n1), int(geo.n2), int(geo.cols), int(geo.rows)
    loc = n // world
    mk = lambda: torch.empty(loc, dtype=torch.float16, device="cuda")
    a = [mk() for _ in range(8)]
    col = capi.TfftPlan(n1, 1, 0, inner=c, in_batch_stride=n1 * c, out_batch_stride=n1 * c, preserve_input=True, fourstep_n=n, fourstep_col0=0)
    row = capi.TfftPlan(n2, k, 0, in_batch_stride=n2, out_batch_stride=n2, preserve_input=True)
    ws = torch.empty(max(1, row.workspace_bytes // 2), dtype=torch.float16, device="cuda")
    if row.workspace_bytes:
        row.set_workspace(ws)

    def step():
        col.exec(a[0], a[1], a[2], a[3])
        capi.permute_twiddle(a[2], a[3], a[4], a[5], world, k, c)
        row.exec(a[4], a[5], a[6], a[7])

    for _ in range(5):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        step()
    e1.record()
    torch.cuda.synchronize()
    print(f"legacy (column pass + re-order + rows) N=2^{lg} world={world}: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us per transform", flush=True)


if what == "legacy":
    for lg, w in ((26, 8), (26, 2), (25, 4), (28, 8), (30, 8)):
        legacy(lg, w)
        emulate_time_only = True
elif what == "emu":
    cases = ((20, 1), (20, 2), (21, 4), (24, 4), (24, 2), (25, 4), (26, 2), (26, 8), (16, 2))
    if len(sys.argv) > 2:                    # python tools/try_dist.py emu 26:8 25:4 ...
        cases = tuple(tuple(int(v) for v in a.split(":")) for a in sys.argv[2:])
    for lg, w in cases:
        emulate(lg, w)
elif what == "self":
    from tensor_fft_amd.distributed import DistributedFFT1D, HipEngine

    for lg in (20, 26):
        n = 1 << lg
        rng = np.random.default_rng(lg)
        xr, xi = rng.uniform(-1, 1, n).astype(np.float16), rng.uniform(-1, 1, n).astype(np.float16)
        f = DistributedFFT1D(n, engine=HipEngine(0), transport="rccl", self_via_comm=True)
        idx = f.input_indices()
        t0 = time.time()
        re, im = f.forward(torch.from_numpy(xr[idx].copy()).cuda(), torch.from_numpy(xi[idx].copy()).cuda())
        torch.cuda.synchronize()
        print(f"self-via-RCCL N=2^{lg}: first forward {time.time() - t0:.2f} s", flush=True)
        exact = np.fft.fft(xr.astype(np.float64) + 1j * xi.astype(np.float64)) / n
        got = re.cpu().numpy().astype(np.float64) + 1j * im.cpu().numpy().astype(np.float64)
        want = exact[f.output_indices()]
        rel = float(np.linalg.norm(got - want) / np.linalg.norm(want))
        print(f"    rel-L2 {rel:.2e}", flush=True)
        assert rel < 1.5e-3
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a, b = torch.from_numpy(xr[idx].copy()).cuda(), torch.from_numpy(xi[idx].copy()).cuda()
        for _ in range(3):
            f.forward(a, b)
        e0.record()
        for _ in range(20):
            f.forward(a, b)
        e1.record()
        torch.cuda.synchronize()
        print(f"    {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per transform incl. the self exchange", flush=True)
