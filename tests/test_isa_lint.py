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
