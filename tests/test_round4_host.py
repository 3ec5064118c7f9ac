"""Round 4, host side (no GPU): the versioned option structs of the C ABI, the setup agreement of the
distributed transform (a rank that fails alone must not strand its peers), bench.py's one failure policy."""
import ctypes
import os
import sys

import pytest

import tensor_fft_amd as tf
from tensor_fft_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module", autouse=True)
def built():
    import __graft_entry__ as g

    g.build()


def test_plan_opts_sizes_the_library_knows():
    L = capi.load_library()
    assert ctypes.sizeof(capi.PlanOpts) == 72                      # today's layout
    for size, known in ((0, 0), (8, 0), (48, 1), (56, 0), (64, 1), (72, 1), (80, 0), (4096, 0)):
        assert L.tfft_plan_opts_known_size(size) == known, size
    buf = (ctypes.c_uint8 * 80)(*([0xAB] * 80))
    assert L.tfft_plan_opts_init(buf, 64) == 0
    assert bytes(buf[:4]) == (64).to_bytes(4, "little") and not any(buf[4:64]) and buf[64] == 0xAB    # zeroed exactly `bytes`
    assert L.tfft_plan_opts_init(buf, 60) == 5 and "not the size" in capi.last_error()
    assert L.tfft_plan_opts_init(None, 72) == 5


def _create_rc(opts_bytes):
    """tfft_plan_create with a hand-built option block; returns (rc, message). Without a GPU every accepted block ends in the
    device check (TFFT_ERR_DEVICE / TFFT_ERR_HIP), every refused one in TFFT_ERR_ARG before any device call."""
    L = capi.load_library()
    buf = ctypes.create_string_buffer(bytes(opts_bytes), len(opts_bytes))
    h = ctypes.c_void_p()
    create = L.tfft_plan_create
    rc = create(4096, 8, 0, ctypes.cast(buf, ctypes.POINTER(capi.PlanOpts)), ctypes.byref(h))
    msg = capi.last_error()
    if rc == 0:
        L.tfft_plan_destroy(h)
    return rc, msg


def test_plan_create_reads_only_what_struct_size_covers():
    import torch

    gpu = torch.cuda.is_available()
    accepted = (0,) if gpu else (6, 7)
    # the 64-byte layout (no launch_iters / input_order) followed by garbage the library must never look at
    old = (64).to_bytes(4, "little") + bytes(60) + b"\xff" * 64
    rc, msg = _create_rc(old)
    assert rc in accepted, (rc, msg)
    # the oldest layout, 48 bytes, garbage behind it
    rc, msg = _create_rc((48).to_bytes(4, "little") + bytes(44) + b"\xff" * 64)
    assert rc in accepted, (rc, msg)
    # unknown sizes: refused before anything else happens, and the message says what to do
    for size in (0, 44, 52, 68, 76, 1 << 20):
        rc, msg = _create_rc(size.to_bytes(4, "little") + bytes(124))
        assert rc == 5 and "struct_size" in msg and "TFFT_PLAN_OPTS_INIT" in msg, (size, rc, msg)
    # reserved bytes must be zero
    rc, msg = _create_rc((72).to_bytes(4, "little") + (1).to_bytes(4, "little") + bytes(64))
    assert rc == 5 and "reserved" in msg


def test_dist_geometry_is_filled_up_to_the_callers_size():
    L = capi.load_library()
    g = capi.dist_geometry(1 << 26, 8, 3)
    assert (g.n1, g.n2, g.cols, g.rows, g.world, g.rank) == (256, 1 << 18, 1 << 15, 32, 8, 3)
    assert g.struct_size == ctypes.sizeof(capi.DistGeometry)
    # a caller with a LONGER struct (a future header): its extra bytes are left alone
    n = ctypes.sizeof(capi.DistGeometry)
    buf = (ctypes.c_uint8 * (n + 16))(*([0xCD] * (n + 16)))
    ctypes.memmove(buf, ctypes.byref(ctypes.c_uint32(n + 16)), 4)
    fn = L.tfft_dist_geometry_query
    assert fn(1 << 26, 8, 3, ctypes.cast(buf, ctypes.POINTER(capi.DistGeometry))) == 0
    assert bytes(buf[n:]) == b"\xcd" * 16 and int.from_bytes(bytes(buf[:4]), "little") == n + 16
    # no size / a size shorter than the struct the library fills: refused, nothing written
    for bad in (0, 8, n - 8):
        buf2 = (ctypes.c_uint8 * n)(*([0xEE] * n))
        ctypes.memmove(buf2, ctypes.byref(ctypes.c_uint32(bad)), 4)
        assert fn(1 << 26, 8, 3, ctypes.cast(buf2, ctypes.POINTER(capi.DistGeometry))) == 5
        assert "struct_size" in capi.last_error() and bytes(buf2[4:]) == b"\xee" * (n - 4)


def test_cache_policy_follows_the_footprint():
    """tfft_plan_cache_policy (host only): plain accesses where the measured scan (profiles/r4_cache_policy.txt) found them
    faster, streaming everywhere else. Footprint = 12 bytes per sample."""
    P = tf.plan_cache_policy
    # two passes of radix 512 / 1024: every footprint up to 512 MiB
    assert P(1 << 18, 1, 1) and P(1 << 20, 1, 1) and P(1 << 20, 1, 32) and P(1 << 19, 1, 64)
    assert not P(1 << 20, 1, 64) and not P(1 << 20, 1, 1024) and not P(1 << 18, 1, 256)
    assert P(1 << 20, 1, 42) and not P(1 << 20, 1, 43)               # 504 MiB / 516 MiB
    # three passes: 128..512 MiB
    assert not P(1 << 21, 1, 1) and not P(1 << 22, 1, 2) and P(1 << 22, 1, 4) and P(1 << 24, 1, 1) and P(1 << 25, 1, 1)
    assert not P(1 << 24, 1, 4) and not P(1 << 26, 1, 1) and not P(1 << 30, 1, 1)
    # radix-256 passes, single-kernel lengths, strided axes: streaming
    assert not P(1 << 16, 1, 256) and not P(1 << 17, 1, 128) and not P(4096, 1, 4096) and not P(256, 1, 1)
    assert not P(512, 4096, 8) and not P(1 << 20, 64, 1)
    # nonsense in, 0 out (no error channel)
    assert not P(3 << 18, 1, 1) and not P(1 << 20, 0, 1) and not P(1 << 20, 1, 0) and not P(1 << 20, 1, 1 << 62)


def test_small_work_gets_the_split_with_more_workgroups():
    """tfft_plan_default_variant (host only): what variant 0 means for a natural-order plan that does not fill the chip. Round 5:
    256 x 256 x R on the latency column kernel for 2^17 ... 2^21 up to the measured limits (profiles/r5_small_scan.txt); from the
    limits on, and for every other length, it is 0; the two within-noise rules of round 4 (2^24 x 2, 2^25 x 1) live in
    profiles/r5_TunerResults.dat now (tests/test_round5_host.py)."""
    V, D = tf.plan_default_variant, tf.plan_describe
    S = 8388608 | 33554432
    assert V(1 << 20, 1, 1) == S and V(1 << 20, 1, 4) == S and V(1 << 20, 1, 8) == 0 and V(1 << 20, 1, 1024) == 0
    assert D(1 << 20, 1, V(1 << 20, 1, 1)) == "col:256+tw col:256+tw autosort:16-tw" and D(1 << 20, 1, V(1 << 20, 1, 1024)) == "col:1024+tw col:1024"
    assert V(1 << 19, 1, 8) == S and V(1 << 19, 1, 16) == 0
    assert D(1 << 19, 1, V(1 << 19, 1, 1)) == "col:256+tw col:256+tw autosort:8-tw"
    assert V(1 << 18, 1, 1) == S and V(1 << 18, 1, 4) == S and V(1 << 18, 1, 8) == 268435456 and V(1 << 18, 1, 16) == 268435456 and V(1 << 18, 1, 32) == 0
    assert D(1 << 18, 1, V(1 << 18, 1, 1)) == "col:256+tw col:256+tw autosort:4-tw" and D(1 << 18, 1, V(1 << 18, 1, 16)) == "col:512+tw col:512"
    assert V(1 << 21, 1, 2) == S and V(1 << 21, 1, 4) == 0
    assert D(1 << 21, 1, V(1 << 21, 1, 1)) == "col:256+tw col:256+tw autosort:32-tw"
    assert V(1 << 17, 1, 1) == S and V(1 << 17, 1, 8) == S and V(1 << 17, 1, 16) == 0 and V(1 << 17, 1, 64) == 0
    assert D(1 << 17, 1, V(1 << 17, 1, 1)) == "col:256+tw col:256+tw autosort:2-tw"
    # 2^15 up to 8 transforms: 256 x 128 (latency column kernel + the workgroup-cooperative radix-128 pass) instead of the single-pass kernel
    assert V(1 << 15, 1, 1) == (S | 16777216) and V(1 << 15, 1, 8) == (S | 16777216) and V(1 << 15, 1, 16) == 0 and V(1 << 15, 1, 8192) == 0
    assert D(1 << 15, 1, V(1 << 15, 1, 1)) == "col:256+tw autosort:128-tw" and D(1 << 15, 1, 0) == "k4096r:8"
    # 2^14 / 2^13 up to 4 transforms: 256 x 64 / 256 x 32 the same way (cooperative radix-64 / 32 pass)
    assert V(1 << 14, 1, 1) == (S | 16777216) and V(1 << 14, 1, 4) == (S | 16777216) and V(1 << 14, 1, 8) == 0
    assert D(1 << 14, 1, V(1 << 14, 1, 1)) == "col:256+tw autosort:64-tw"
    assert V(1 << 13, 1, 1) == (S | 16777216) and V(1 << 13, 1, 4) == (S | 16777216) and V(1 << 13, 1, 8) == 0
    assert D(1 << 13, 1, V(1 << 13, 1, 1)) == "col:256+tw autosort:32-tw" and D(1 << 13, 1, 0) == "k4096r:2"
    for lg in (8, 12, 16, 22, 24, 25, 26):
        assert V(1 << lg, 1, 1) == 0 and V(1 << lg, 1, 2) == 0
    assert V(1 << 20, 64, 1) == 0 and V(3 << 19, 1, 1) == 0 and V(1 << 20, 1, 0) == 0
    for lg, b in ((15, 1), (17, 1), (18, 1), (18, 16), (19, 4), (20, 4), (21, 1)):                          # every value it returns is a variant the library accepts
        capi.variant_check(1 << lg, 1, V(1 << lg, 1, b))


def test_header_and_binding_agree_on_the_layouts():
    """ctypes mirrors of the two versioned structs against the header, compiled by the host compiler."""
    import subprocess
    import tempfile

    src = r'''
#include <stdio.h>
#include <stddef.h>
#include "tfft.h"
int main(void) {
  tfft_plan_opts o = TFFT_PLAN_OPTS_INIT;
  tfft_dist_geometry g = TFFT_DIST_GEOMETRY_INIT;
  printf("%zu %u %zu %zu %zu %zu %zu %u %zu\n", sizeof(o), o.struct_size, offsetof(tfft_plan_opts, in_batch_stride),
         offsetof(tfft_plan_opts, fourstep_n), offsetof(tfft_plan_opts, launch_iters), offsetof(tfft_plan_opts, input_order),
         sizeof(g), g.struct_size, offsetof(tfft_dist_geometry, local_passes));
  return (o.variant | o.input_order | (int)o.inner) != 0;
}
'''
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "t.c")
        open(c, "w").write(src)
        exe = os.path.join(d, "t")
        subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-o", exe, c])
        out = subprocess.check_output([exe], text=True).split()
    P, G = capi.PlanOpts, capi.DistGeometry
    want = [ctypes.sizeof(P), ctypes.sizeof(P), P.in_batch_stride.offset, P.fourstep_n.offset, P.launch_iters.offset,
            P.input_order.offset, ctypes.sizeof(G), ctypes.sizeof(G), G.local_passes.offset]
    assert [int(v) for v in out] == want, (out, want)


# ---- distributed setup: all ranks leave the constructor together (gloo, two ranks, an engine whose native part fails on ONE rank)
_WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
import torch, torch.distributed as dist
dist.init_process_group("gloo")
rank = dist.get_rank()
from tensor_fft_amd.distributed import DistributedFFT1D, DistSetupError

class Geo:                      # what capi.dist_geometry returns, as far as the constructor looks at it
    n1, n2, cols, rows, world = 256, 1 << 12, 1 << 11, 128, 2
class FakeCapi:
    class TfftError(RuntimeError): pass
    @staticmethod
    def dist_unique_id():
        if %(fail_id)d: raise RuntimeError("dlopen(librccl.so.1) failed")
        return b"x" * 128
    class DistComm:
        def __init__(self, world, rank, uid, device):
            if %(fail_comm_rank)d == rank: raise RuntimeError("ncclCommInitRank: unhandled system error")
        def close(self): pass
class Plan:
    geometry = Geo()
    def close(self): pass
class Engine:
    capi, device = FakeCapi, 0
    def supports_fourstep(self, n1, inner): return True
    def dist_geometry(self, n, world, rank): return Geo()
    def dist_plan(self, n, world, rank, comm=None, self_via_comm=False, slabs=1):
        if %(fail_plan_rank)d == rank: raise RuntimeError("hipMalloc(distributed plan buffers): out of memory")
        return (Plan(), None, None, None, None, None, None)
try:
    f = DistributedFFT1D(1 << 20, engine=Engine(), transport="rccl")
    print("rank", rank, "CREATED", flush=True)
except DistSetupError as e:
    print("rank", rank, "SETUP-ERROR:", e, flush=True)
# whatever happened, both ranks are here and the process group is still usable: no one is stuck in a collective
t = torch.ones(1)
dist.all_reduce(t)
assert float(t[0]) == 2.0
print("rank", rank, "IN-STEP", flush=True)
dist.destroy_process_group()
'''


@pytest.mark.parametrize("fail_id,fail_comm_rank,fail_plan_rank,expect", [
    (0, -1, -1, "CREATED"),
    (1, -1, -1, "could not create the RCCL unique id"),          # rank 0 fails before the broadcast: the error travels instead of the id
    (0, 1, -1, "creating the RCCL communicator"),                # rank 1 alone fails in ncclCommInitRank
    (0, -1, 0, "creating the distributed plan"),                 # rank 0 alone fails allocating its plan
])
def test_all_ranks_agree_on_a_failed_setup(tmp_path, fail_id, fail_comm_rank, fail_plan_rank, expect):
    import subprocess

    script = tmp_path / "w.py"
    script.write_text(_WORKER % {"root": ROOT, "fail_id": fail_id, "fail_comm_rank": fail_comm_rank, "fail_plan_rank": fail_plan_rank})
    import bench

    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(bench.free_port()), str(script)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out
    assert out.count("IN-STEP") == 2, out
    if expect == "CREATED":
        assert out.count("CREATED") == 2
    else:
        assert out.count("SETUP-ERROR") == 2 and expect in out, out


def test_bench_side_entries_share_one_failure_policy(monkeypatch, capsys):
    """other_configs(): a failing entry becomes {"error": ...}; the others are still measured (here: all entries stubbed)."""
    import bench

    class FakeTorch:
        class cuda:
            @staticmethod
            def empty_cache():
                pass

    calls = []

    def boom(*a, **k):
        calls.append(1)
        raise bench.CheckFailed("self-check failed: stub")

    class FakeTf:
        TfftPlan = staticmethod(boom)
        TfftPlan2D = staticmethod(boom)

        @staticmethod
        def synth_uniform(*a, **k):
            pass

    class T:
        float16 = None

        @staticmethod
        def empty(*a, **k):
            class X:
                def __getitem__(self, k):
                    return self
            return X()
        empty_like = empty
        cuda = FakeTorch.cuda

    out = bench.other_configs(T, FakeTf, None, 0)
    assert len(out) == 14 and all("error" in v and "stub" in v["error"] for v in out.values()), out
    assert len(calls) == 14                                        # every entry was attempted (round 5: + reference_protocol_single)
    assert "reference_protocol_single" in out
    assert "other_configs[configs[3]_2d_4096x4096_x_64]" in capsys.readouterr().err
