"""Host-side checks of the C ABI that need no GPU: the library loads, exports every symbol
include/tfft.h declares, and its CreatePlan arithmetic equals the reference's (src/base/Plan.h:77-194)."""
import os
import re

import pytest

import tensor_fft_amd as tf
from tensor_fft_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module", autouse=True)
def built():
    import __graft_entry__ as g

    g.build()


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "tfft.h")).read()
    declared = set(re.findall(r"\b(tfft_[a-z0-9_]+)\s*\(", header))
    assert declared == set(capi.SYMBOLS), declared ^ set(capi.SYMBOLS)
    lib = capi.load_library()
    for name in declared:
        assert hasattr(lib, name), name
    assert b"gfx950" in lib.tfft_version()


def test_library_contains_gfx950_code_object():
    blob = open(capi.lib_path(), "rb").read()
    assert b"gfx950" in blob and b"fft4096_kernel" in blob


def test_create_plan_matches_reference_arithmetic(orc):
    for lg in range(8, 30):
        n = 1 << lg
        for mode in (tf.Mode_256, tf.Mode_4096):
            p = tf.CreatePlan(n, mode, 16 if mode == tf.Mode_4096 else 1, 1, 256)
            o = orc.ref_plan(n, mode)
            if o is None:
                assert p is None
                continue
            assert (p.amount_of_r16_steps_, p.amount_of_r2_steps_, p.results_in_results_) == o
            assert p.fft_length_ == n and p.base_fft_mode_ == mode
            assert p.base_fft_blocksize_ == 32 * p.base_fft_warps_per_block_
            assert p.base_fft_gridsize_ * p.base_fft_warps_per_block_ == n // 256
            assert p.base_fft_shared_mem_in_bytes_ == 2048 * p.base_fft_warps_per_block_
            assert p.r16_shared_mem_in_bytes_ == 1536 * p.r16_warps_per_block_


def test_create_plan_defaults_and_overrides(capsys):
    p = tf.CreatePlan(4096)                       # Mode_256, 8, 8, 256
    assert (p.base_fft_warps_per_block_, p.base_fft_gridsize_, p.r16_warps_per_block_, p.r2_blocksize_) == (8, 2, 8, 256)
    assert p.results_in_results_ is False         # one radix-16 pass after the 256 base: lands in the input half
    p = tf.CreatePlan(256)                        # 1 warp in total: both wpb clamped (Plan.h:119-126,153-160)
    assert (p.base_fft_warps_per_block_, p.r16_warps_per_block_) == (1, 1)
    assert "Warning" in capsys.readouterr().out
    p = tf.CreatePlan(8192, tf.Mode_4096, 8, 8, 256)   # forced to 16 warps (Plan.h:135-141)
    assert p.base_fft_warps_per_block_ == 16 and p.base_fft_blocksize_ == 512


def test_create_plan_rejections(capsys):
    assert tf.CreatePlan(3000) is None
    assert "power of 2" in capsys.readouterr().out
    assert tf.CreatePlan(128) is None
    assert tf.CreatePlan(1024, tf.Mode_4096) is None
    assert tf.CreatePlan(1 << 13, tf.Mode_256, 3, 8, 256) is None        # 32 warps not divisible by 3
    assert tf.CreatePlan(1 << 9, tf.Mode_256, 1, 1, 512) is None         # smallest r2 sub-FFT 256 % 512 != 0
    rc, _, msg = capi.ref_create_plan(3000)
    assert rc == 1 and "power of 2" in msg


def test_create_plan_from_tuner_file(tmp_path, capsys):
    f = tmp_path / "TunerResults.dat"
    f.write_text("4096 4096 16 4 128 10\n8192 256 8 16 256\n")
    p = tf.CreatePlan(4096, str(f))
    assert p.base_fft_mode_ == tf.Mode_4096 and p.r16_warps_per_block_ == 4 and p.r2_blocksize_ == 128
    assert p._variant == 10                                  # sixth column written by tools/tuner.py
    p = tf.CreatePlan(8192, str(f))
    assert p.base_fft_mode_ == tf.Mode_256 and p.r16_warps_per_block_ == 16
    assert tf.CreatePlan(1 << 20, str(f)) is None
    assert tf.CreatePlan(4096, str(tmp_path / "missing.dat")) is None
    assert "tuner file" in capsys.readouterr().out.lower()


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(capi, "_lib", None)
    monkeypatch.setattr(capi, "_LIB_NAME", "libtfft_not_built.so")
    with pytest.raises(ImportError):
        capi.load_library()


def test_no_gpu_means_errors_not_fallbacks():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(tf.TfftError):
        tf.TfftPlan(4096, 8, 0)
    with pytest.raises(RuntimeError):
        tf.DataHandler(4096)


def test_product_never_touches_the_oracle():
    pkg = os.path.join(ROOT, "tensor-fft_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                text = open(os.path.join(dirpath, fn)).read()
                assert "oracle" not in text.lower(), os.path.join(dirpath, fn)


def _radices(desc):
    return [int(tok.split(":")[1].split("+")[0].split("-")[0]) for tok in desc.split()]


def test_planner_decompositions_multiply_to_n():
    """tfft_plan_describe (host only): for every power of two and a range of strided-axis widths and planner variants the
    pass radices multiply to N, column passes come first, and only a pass directly behind a column pass skips its twiddles."""
    import tensor_fft_amd as tf

    for lg in range(1, 31):
        n = 1 << lg
        for inner in (1, 8, 16, 64, 4096):
            for variant in (0, 32, 8388608, 33554432, 8388608 | 33554432, 134217728, 16777216, 2097152, 16777216 | 8388608):
                d = tf.plan_describe(n, inner, variant)
                toks = d.split()
                rad = _radices(d)
                prod = 1
                for r in rad:
                    prod *= r
                if toks[0].startswith(("k256r", "k4096r")):
                    prod *= 256 if toks[0].startswith("k256r") else 4096
                assert prod == n, (n, inner, variant, d)
                kinds = [t.split(":")[0] for t in toks]
                if "col" in kinds:
                    last_col = max(i for i, k in enumerate(kinds) if k == "col")
                    assert all(k == "col" for k in kinds[: last_col + 1]), d
                for i, t in enumerate(toks):
                    if t.endswith("-tw"):
                        assert i > 0 and kinds[i - 1] == "col" and toks[i - 1].endswith("+tw"), d
                    if t.endswith("+tw"):
                        assert i + 1 < len(toks), d


def test_planner_pass_counts():
    """The pass counts DESIGN.md quotes: one pass up to 2^15, two up to 2^20, three up to 2^30 (contiguous axis)."""
    import tensor_fft_amd as tf

    want = {8: 1, 9: 1, 10: 1, 11: 1, 12: 1, 13: 1, 14: 1, 15: 1, 16: 2, 17: 2, 18: 2, 19: 2, 20: 2, 21: 3, 22: 3, 23: 3,
            24: 3, 25: 3, 26: 3, 27: 3, 28: 3, 29: 3, 30: 3}
    for lg, passes in want.items():
        assert len(tf.plan_describe(1 << lg).split()) == passes, (lg, tf.plan_describe(1 << lg))
    assert tf.plan_describe(1 << 12) == "k4096:4096"
    assert tf.plan_describe(1 << 15) == "k4096r:8"
    assert tf.plan_describe(1 << 17) == "col:512+tw col:256"
    assert tf.plan_describe(1 << 16) == "col:256+tw col:256"
    assert tf.plan_describe(1 << 19) == "col:512+tw col:1024"
    assert tf.plan_describe(1 << 20) == "col:1024+tw col:1024"
    assert tf.plan_describe(1 << 20, 1, 33554432) == "col:256+tw col:256+tw autosort:16-tw"
    assert tf.plan_describe(1 << 28) == "col:256+tw col:1024+tw col:1024"
    assert tf.plan_describe(1 << 28, 1, 134217728) == "col:1024+tw col:1024+tw col:256"
    assert tf.plan_describe(1 << 27, 1, 33554432) == "col:512+tw col:512+tw col:512"
    assert tf.plan_describe(1 << 26) == "col:512+tw col:512+tw col:256"
    assert tf.plan_describe(1 << 21) == "col:512+tw col:512+tw autosort:8-tw"
    assert tf.plan_describe(4096, 4096) == "col:256+tw autosort:16-tw"          # 2D column pass (general shapes)
    assert tf.plan_describe(512, 4096, 67108864) == "col:512"                    # second pass of the fused 4096^2 plan
    assert tf.plan_describe(1 << 13, 1, 16777216) == "col:256+tw autosort:32-tw"
    assert tf.plan_describe(1 << 16, 1, 32) == "autosort:16 autosort:16 autosort:16 autosort:16"
    with pytest.raises(tf.TfftError):
        tf.plan_describe(3000)



# ---- tfft_plan_opts.variant: only documented, CORRECT-result values pass (a stale tuner file must not
# ---- select a timing-only kernel silently)
DEBUG_VARIANTS = [4, 64, 128, 65536, 1 << 8, 3 << 8, 15 << 8, 4 | 8, 64 | 8, 524288 | 128]
TUNER_VARIANTS = [0, 1, 2, 8, 9, 10, 16, 32, 4096, 8192, 131072, 262144, 524288, 1048576, 2097152, 4194304, 8388608,
                  16777216, 16777216 | 8388608, 33554432, 8388608 | 33554432, 134217728, 268435456, 536870912, 1073741824,
                  1073741824 | 8388608 | 33554432]


def test_variant_check_refuses_debug_and_unknown_bits(monkeypatch):
    monkeypatch.delenv("TFFT_DEBUG_VARIANTS", raising=False)
    for n in (256, 4096, 8192, 1 << 16, 1 << 20):
        for v in TUNER_VARIANTS:
            capi.variant_check(n, 1, v)
        for v in DEBUG_VARIANTS:
            with pytest.raises(tf.TfftError) as e:
                capi.variant_check(n, 1, v)
            assert e.value.code == 5 and "TFFT_DEBUG_VARIANTS" in e.value.message
            with pytest.raises(tf.TfftError):
                tf.plan_describe(n, 1, v)
        for v in (-1, -(1 << 30), -(1 << 12)):                # (bits 0 .. 30 all have a meaning since round 5: only the sign bit is left)
            with pytest.raises(tf.TfftError) as e:
                capi.variant_check(n, 1, v)
            assert "unknown" in e.value.message
    for v in (3, 11, 16 | 2, 16 | 8):                       # N = 4096 kernel: combinations without a compiled kernel
        with pytest.raises(tf.TfftError):
            capi.variant_check(4096, 1, v)
    # the shipped library holds no timing-only kernels at all (they are compiled only into libtfft_debug.so, which the
    # drivers under tools/ build and load): the environment variable alone opens nothing
    monkeypatch.setenv("TFFT_DEBUG_VARIANTS", "1")
    for v in DEBUG_VARIANTS:
        with pytest.raises(tf.TfftError):
            capi.variant_check(1 << 20, 1, v)


def test_tuner_file_with_unusable_variant_is_refused(tmp_path, capsys, monkeypatch):
    monkeypatch.delenv("TFFT_DEBUG_VARIANTS", raising=False)
    f = tmp_path / "TunerResults.dat"
    f.write_text("4096 4096 16 1 256 64\n65536 4096 16 1 256 65536\n1048576 4096 16 1 256 524288\n"
                 "262144 4096 16 1 256 notanumber\n131072 4096 16 1 256 -5\n")
    assert tf.CreatePlan(4096, str(f)) is None                      # 64 = no-compute timing kernel
    assert tf.CreatePlan(65536, str(f)) is None                     # 65536 = copy-only column pass
    assert tf.CreatePlan(262144, str(f)) is None
    assert tf.CreatePlan(131072, str(f)) is None                    # not a variant at all
    assert "unusable kernel variant" in capsys.readouterr().out
    assert tf.CreatePlan(1 << 20, str(f))._variant == 524288


def test_every_committed_tuner_file_holds_only_accepted_variants(monkeypatch):
    import glob

    monkeypatch.delenv("TFFT_DEBUG_VARIANTS", raising=False)
    files = glob.glob(os.path.join(ROOT, "profiles", "*TunerResults.dat"))
    assert files
    for path in files:
        for line in open(path):
            tok = line.split()
            if len(tok) >= 6:
                capi.variant_check(int(tok[0]), 1, int(tok[5]))


def _hipcc_host(src, exe, *extra):
    import subprocess

    pkg = os.path.join(ROOT, "tensor-fft_amd")
    subprocess.check_call(["hipcc", "-O1", "-std=c++17", "--offload-arch=gfx950", "-I", os.path.join(ROOT, "include"),
                           "-o", exe, src, "-L", pkg, "-ltfft", "-Wl,-rpath," + pkg, *extra])


def test_cxx_shim_tuner_file_overload_keeps_column_six(tmp_path, monkeypatch):
    """include/tensor_fft.hpp CreatePlan(N, file): same plan as the Python binding from the same file, and
    the same refusals. Host only: CreatePlan touches no device."""
    import subprocess

    monkeypatch.delenv("TFFT_DEBUG_VARIANTS", raising=False)
    exe = str(tmp_path / "tuner_file_host")
    _hipcc_host(os.path.join(ROOT, "tests", "cxx", "tuner_file_host.cpp"), exe)
    f = tmp_path / "TunerResults.dat"
    f.write_text("4096 4096 16 4 128 10\n8192 256 8 16 256\n65536 4096 16 1 256 65536\n1048576 4096 16 1 256 524288\n")

    def run(n, *more):
        return subprocess.run([exe, str(f), str(n), *more], capture_output=True, text=True, check=True).stdout.split("\n")[-2].split()

    assert run(4096)[:5] == ["ok", "10", "2", "0", "1"]
    assert run(8192)[:5] == ["ok", "0", "2", "1", "0"]
    assert run(1 << 20)[:5] == ["ok", "524288", "4", "0", "1"]
    assert run(65536) == ["refused"]
    assert run(1 << 22) == ["refused"]
    for n in (4096, 8192, 1 << 20):
        p = tf.CreatePlan(n, str(f))
        assert [str(p._variant), str(p.amount_of_r16_steps_), str(p.amount_of_r2_steps_), str(p.base_fft_mode_)] == run(n)[1:5]
    # columns 7 / 8 (launch_iters, batch): one entry per line of the length, the nearest batch wins
    f.write_text("4096 4096 16 1 256 10 65535 1\n4096 4096 16 1 256 16 1 64\n4096 4096 16 1 256 10 2 65536\n8192 4096 16 1 256 0 70000 4\n")
    assert " ".join(run(4096, "100")[5:]) == "[1 10 65535] [64 16 1] [65536 10 2] pick 16 1"
    assert run(4096, "1")[-3:] == ["pick", "10", "65535"] and run(4096, "2000000")[-3:] == ["pick", "10", "2"]
    assert run(8192) == ["refused"]
    p = tf.CreatePlan(4096, str(f))
    assert p._tuned == [(1, 10, 65535), (64, 16, 1), (65536, 10, 2)]


def test_rotor_covers_every_item_once(tmp_path):
    """k4096::Rotor, the rotated work distribution of the persistent kernels, on the host: every item exactly
    once for ragged grids and totals, look-ahead = next item, all residues mod 8 per workgroup."""
    import subprocess

    exe = str(tmp_path / "rotor_host")
    _hipcc_host(os.path.join(ROOT, "tests", "cxx", "rotor_host.cpp"), exe)
    assert subprocess.run([exe], capture_output=True, text=True, check=True).stdout.strip() == "ok"


def test_reference_style_mains_compile_against_the_shim(tmp_path):
    """A main() that makes the reference's calls in the reference's order (ExampleBatchFFT.cu:20-85, ExampleSingleFFT.cu)
    builds against include/tensor_fft.hpp with hipcc. (Running it needs a GPU: tests/test_gpu_parity.py.)"""
    for name in ("example_batch_fft", "example_single_fft"):
        _hipcc_host(os.path.join(ROOT, "examples", name + ".cpp"), str(tmp_path / name))
        assert os.path.getsize(tmp_path / name) > 0


def test_tuner_file_with_batch_lines(tmp_path):
    """Columns 7 and 8 of the tuner file (launch_iters, the batch a line was tuned at; tools/tuner.py --batches): CreatePlan
    reads every line of the length, ComputeFFT takes the line nearest to its batch on a log scale."""
    from tensor_fft_amd import reference_api as ra

    f = tmp_path / "TunerResults.dat"
    f.write_text("4096 4096 16 1 256 10 65535 1\n4096 4096 16 1 256 16 1 64\n4096 4096 16 1 256 10 2 65536\n"
                 "8192 4096 16 1 256 0\n65536 4096 16 1 256 524288 70000 4\n")
    p = tf.CreatePlan(4096, str(f))
    assert p._tuned == [(1, 10, 65535), (64, 16, 1), (65536, 10, 2)] and p._variant == 10
    assert ra.tuned_for_batch(p._tuned, 1) == (10, 65535)
    assert ra.tuned_for_batch(p._tuned, 100) == (16, 1)
    assert ra.tuned_for_batch(p._tuned, 20000) == (10, 2)
    assert ra.tuned_for_batch(p._tuned, 1 << 21) == (10, 2)
    q = tf.CreatePlan(8192, str(f))
    assert q._tuned == [(0, 0, 0)] and ra.tuned_for_batch(q._tuned, 77) == (0, 0)
    assert tf.CreatePlan(65536, str(f)) is None                      # launch_iters outside 0 .. 65535
    assert ra.tuned_for_batch([], 5, default_variant=8) == (8, 0)

