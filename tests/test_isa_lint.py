"""CPU-side checks of the gfx950 ISA inside libtfft.so (tools/isa_lint.py): no packed fp32 arithmetic in MFMA kernels
and every MFMA -> consumer wait state of the gfx950 tables present. Round 1 met an intermittent wrong twiddle product that
disappeared with clang's SLP vectoriser off; the GPU-side guards are tests/test_gpu_determinism.py, this
is the guard that needs no GPU and fails the moment a build (other flags, another hipcc, explicit ext-vector maths)
puts v_pk_{mul,fma,add}_f32 next to MFMAs again."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.fixture(scope="module")
def report():
    import __graft_entry__ as g
    import isa_lint

    g.build()
    text = isa_lint.disassemble(os.path.join(ROOT, "tensor-fft_amd", "libtfft.so"))
    return isa_lint.lint_text(text)


def test_code_object_holds_the_mfma_kernels(report):
    names = "\n".join(report)
    for needle in ("fft4096_kernel", "fft256_kernel", "fft256r_kernel", "fft4096r_kernel", "colfft256_wg_kernel",
                   "colfft512_wg_kernel", "colfft256_kernel", "pass_kernel"):
        assert needle in names, needle
    assert sum(r["mfma"] for r in report.values()) > 4000


def test_no_packed_fp32_in_mfma_kernels(report):
    bad = {k: r["pk_f32"] for k, r in report.items() if r["mfma"] and r["pk_f32"]}
    assert not bad, bad


def test_mfma_consumers_keep_the_gfx950_wait_states(report):
    bad = {k: r["findings"] for k, r in report.items() if r["findings"]}
    assert not bad, bad


def test_the_lint_sees_what_it_should():
    """Self-test on hand-written streams: a consumer 7 wait states behind a 4-pass MFMA is a finding, 8 is not; a packed
    multiply in an MFMA kernel is a finding, in a kernel without MFMAs it is not."""
    import isa_lint

    def kernel(body):
        return "_Z1kv:\n" + "\n".join("\t" + l for l in body) + "\n\ts_endpgm\n"

    mfma = "v_mfma_f32_16x16x32_f16 v[0:3], v[4:7], v[8:11], 0"
    short = kernel([mfma, "s_nop 6", "v_mul_f32_e32 v20, v1, v21"])
    ok = kernel([mfma, "s_nop 7", "v_mul_f32_e32 v20, v1, v21"])
    pk = kernel([mfma, "s_nop 15", "v_pk_mul_f32 v[20:21], v[22:23], v[24:25]"])
    pk_only = kernel(["v_pk_mul_f32 v[20:21], v[22:23], v[24:25]"])
    chain = kernel([mfma, "v_mfma_f32_16x16x32_f16 v[0:3], v[4:7], v[8:11], v[0:3]"])            # accumulate chain: fine
    srcb = kernel([mfma, "s_nop 5", "v_mfma_f32_16x16x32_f16 v[12:15], v[4:7], v[0:3], 0"])       # result as B after 6
    assert len(isa_lint.lint_text(short)["_Z1kv"]["findings"]) == 1
    assert not isa_lint.lint_text(ok)["_Z1kv"]["findings"]
    assert len(isa_lint.lint_text(pk)["_Z1kv"]["findings"]) == 1
    assert not isa_lint.lint_text(pk_only)["_Z1kv"]["findings"]
    assert not isa_lint.lint_text(chain)["_Z1kv"]["findings"]
    assert len(isa_lint.lint_text(srcb)["_Z1kv"]["findings"]) == 1


def test_shipped_library_holds_no_timing_only_kernels(report):
    """the wrong-result template instantiations (fake stores = bit 4, no compute = bit 64) and the
    per-workgroup clock hook exist only in the -DTFFT_DEBUG_KERNELS build (libtfft_debug.so), never in libtfft.so."""
    import isa_lint

    names = list(report)
    k4096 = [k for k in names if "fft4096_kernelILi" in k]
    assert k4096
    for k in k4096:
        v = int(k.split("fft4096_kernelILi")[1].split("E")[0])
        assert not (v & (4 | 64)), k
    text = isa_lint.disassemble(os.path.join(ROOT, "tensor-fft_amd", "libtfft.so"))
    assert "s_memrealtime" not in text and "MSG_RTN_GET_REALTIME" not in text and "HW_REG_XCC_ID" not in text
    so = open(os.path.join(ROOT, "tensor-fft_amd", "libtfft.so"), "rb").read()
    for needle in (b"TFFT_WG_TIMES_PTR", b"TFFT_PLAN_COLS", b"TFFT_K4096_ITERS", b"TFFT_COLWG_ITERS", b"TFFT_2D_NO_FUSE",
                   b"TFFT_DEBUG_VARIANTS=1 to allow"):
        assert needle not in so, needle


def test_lds_dma_is_drained_before_every_program_end(report):
    """Pins the fix of commit 6ed7571 (a workgroup must not retire with global_load_lds still in flight): the lint's
    control-flow check finds no s_endpgm reachable behind an LDS-DMA without an s_waitcnt vmcnt(0) in between, in any kernel
    that issues one."""
    dma_kernels = [k for k, r in report.items() if r["lds_dma"]]
    assert len(dma_kernels) >= 10
    bad = {k: r["findings"] for k, r in report.items() if any("LDS-DMA" in f for f in r["findings"])}
    assert not bad, bad


def test_the_dma_drain_lint_sees_what_it_should():
    import isa_lint

    def listing(body):
        lines = ["0000000000001000 <_Z1kv>:"]
        addr = 0x1000
        for ins in body:
            lines.append(f"\t{ins:58s} // {addr:012X}: BF800000")
            addr += 4
        return "\n".join(lines) + "\n"

    def findings(body):
        return isa_lint.lint_text(listing(body))["_Z1kv"]["findings"]

    dma = "global_load_lds_dwordx4 v[0:1], off"
    assert len(findings([dma, "s_endpgm"])) == 1
    assert len(findings([dma, "s_waitcnt vmcnt(2)", "s_endpgm"])) == 1                      # a partial wait is not a drain
    assert not findings([dma, "s_waitcnt vmcnt(0)", "s_endpgm"])
    assert not findings([dma, "s_waitcnt vmcnt(0) lgkmcnt(0)", "s_endpgm"])
    assert not findings(["v_mov_b32_e32 v0, v1", "s_endpgm"])                              # no LDS-DMA at all
    # loop whose body waits at the top: top (0x1000): wait; dma; s_cbranch_scc1 top; s_endpgm -> the exit edge leaves with a
    # copy in flight
    assert len(findings(["s_waitcnt vmcnt(0)", dma, "s_cbranch_scc1 65533", "s_endpgm"])) == 1
    # the structuriser's latch: exit path sets the flag to -1, the look-ahead path to 0, both meet at the latch
    latch = ["s_waitcnt vmcnt(0)",                       # 0x1000  loop top
             "s_mov_b64 s[8:9], -1",                     # 0x1004
             "s_cbranch_scc1 3",                         # 0x1008  -> latch (0x1018)
             dma,                                        # 0x100c
             "s_mov_b64 s[8:9], 0",                      # 0x1010
             "s_nop 0",                                  # 0x1014
             "s_andn2_b64 vcc, exec, s[8:9]",            # 0x1018  latch
             "s_cbranch_vccz 1",                         # 0x101c  -> 0x1024 (exit)
             "s_branch 65527",                           # 0x1020  -> 0x1000
             "s_endpgm"]                                 # 0x1024
    assert not findings(latch)
    broken = list(latch)
    broken[4] = "s_mov_b64 s[8:9], -1"                   # look-ahead path leaves too: a real finding
    assert len(findings(broken)) == 1
    clobbered = list(latch)
    clobbered[5] = "s_lshl_b64 s[8:9], s[8:9], 1"        # flag overwritten by something the lint does not follow: conservative
    assert len(findings(clobbered)) == 1


def test_shipped_column_kernels_are_exactly_the_dispatch_table(report):
    """The column passes are launched through ONE table (tfft.hip, TFFT_COL_* lists; tfft_kernel_list reports it). Its rows and the
    column-kernel instantiations in the gfx950 code object must be the same set: a kernel the launch path cannot reach would be dead
    weight in libtfft.so, a row without code cannot link."""
    import subprocess

    import tensor_fft_amd as tf
    from tensor_fft_amd import capi

    table = set(capi.kernel_list())
    assert len(table) == len(capi.kernel_list()) and len(table) >= 60
    mangled = [k for k in report if "colfft" in k]
    demangled = subprocess.run(["c++filt"], input="\n".join(mangled), capture_output=True, text=True, check=True).stdout.split("\n")
    shipped = set()
    for d in demangled:
        d = d.strip()
        if not d:
            continue
        d = d[len("void "):] if d.startswith("void ") else d
        shipped.add(d.split("(")[0])
    assert shipped == table, sorted(shipped ^ table)
