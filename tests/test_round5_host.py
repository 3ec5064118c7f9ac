"""Round 5, host side (no GPU): tuner results as plan wisdom through the C ABI (tfft_tuning_load in the
reference's own file format, Plan.h:197-255 / FileWriter.h:250-269), the dispatch table (item 6) and the C++ mains that mirror
the reference's tuner and single-transform benchmark (TunerSingleFFT.cu, FFTBenchSinlge.cu)."""
import os

import pytest

import __graft_entry__ as g

g.build()
import tensor_fft_amd as tf  # noqa: E402
from tensor_fft_amd import capi  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(autouse=True)
def clean_wisdom():
    tf.tuning_clear()
    yield
    tf.tuning_clear()


def test_a_loaded_line_changes_what_variant_zero_means(tmp_path):
    """A wisdom file changes tfft_plan_describe-visible choices: the decomposition a variant-0 plan of (N, batch) gets is
    plan_describe(n, 1, plan_default_variant(n, 1, batch)), and that follows the loaded line."""
    n = 1 << 24
    before = tf.plan_describe(n, 1, tf.plan_default_variant(n, 1, 2))
    assert tf.tuning_query(n, 2) is None
    f = tmp_path / "TunerResults.dat"
    f.write_text("4096 4096 16 1 256\n"                      # a plain reference line: nothing for this library
                 "16777216 4096 16 1 256 33554432 0 2\n"     # N mode base_wpb r16_wpb r2_blocksize | variant launch_iters batch
                 "65536 4096 16 1 256 0 4 64\n"
                 "\n")
    assert tf.tuning_load(str(f)) == 2
    assert tf.tuning_query(n, 2) == (33554432, 0)
    after = tf.plan_describe(n, 1, tf.plan_default_variant(n, 1, 2))
    assert before == "col:512+tw col:1024+tw autosort:32-tw" and after == "col:256+tw col:256+tw col:256" and before != after
    # the line reaches three octaves either way on the batch axis, not further; other lengths are untouched
    assert tf.tuning_query(n, 16) == (33554432, 0) and tf.tuning_query(n, 32) is None and tf.tuning_query(n, 1) == (33554432, 0)
    assert tf.tuning_query(1 << 23, 2) is None and tf.tuning_query(4096, 1) is None
    # a line with variant 0 carries only its launch shape and keeps the library's default split
    assert tf.tuning_query(1 << 16, 64) == (0, 4) and tf.plan_default_variant(1 << 16, 1, 64) == 0
    tf.tuning_clear()
    assert tf.tuning_query(n, 2) is None and tf.plan_describe(n, 1, tf.plan_default_variant(n, 1, 2)) == before


def test_nearest_batch_wins_and_later_lines_replace_earlier_ones():
    n = 1 << 18
    tf.tuning_add(n, 1, 524288)
    tf.tuning_add(n, 64, 268435456, 2)
    tf.tuning_add(n, 4096, 0, 8)
    assert tf.tuning_query(n, 1) == (524288, 0) and tf.tuning_query(n, 4) == (524288, 0)
    assert tf.tuning_query(n, 16) == (268435456, 2) and tf.tuning_query(n, 256) == (268435456, 2)
    assert tf.tuning_query(n, 1024) == (0, 8) and tf.tuning_query(n, 32768) == (0, 8) and tf.tuning_query(n, 65536) is None
    tf.tuning_add(n, 64, 32)                                  # same (N, batch): replaced
    assert tf.tuning_query(n, 64) == (32, 0)
    tf.tuning_add(n, 0, 8388608)                              # batch 0 fits any batch, but a nearer line still wins
    assert tf.tuning_query(n, 65536) == (8388608, 0) and tf.tuning_query(n, 64) == (32, 0)


def test_bad_lines_load_nothing(tmp_path):
    f = tmp_path / "bad.dat"
    f.write_text("65536 4096 16 1 256 0 0 1\n65536 4096 16 1 256 64 0 1\n")        # 64: a timing-only kernel (wrong results)
    with pytest.raises(tf.TfftError, match="bad.dat:2"):
        tf.tuning_load(str(f))
    assert tf.tuning_query(1 << 16, 1) is None
    f.write_text("1000 256 1 1 256 0 0 1\n")                  # not a power of two
    with pytest.raises(tf.TfftError):
        tf.tuning_load(str(f))
    with pytest.raises(tf.TfftError, match="Failed to open"):
        tf.tuning_load(str(tmp_path / "missing.dat"))
    with pytest.raises(tf.TfftError):
        tf.tuning_add(1 << 16, 1, 4)                          # debugging bit
    with pytest.raises(tf.TfftError):
        tf.tuning_add(1 << 16, 1, 0, 1 << 20)


def test_committed_tuner_files_load_as_wisdom():
    """Every tuner file under profiles/ is something tfft_tuning_load accepts (the rule of round 2: every emitted value has a
    parity test and passes tfft_variant_check), and this round's file carries the choices that left the library's source."""
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_TunerResults.dat")))
    assert any(f.endswith("r5_TunerResults.dat") for f in files)
    for f in files:
        tf.tuning_clear()
        assert tf.tuning_load(f) >= 1, f
    tf.tuning_clear()
    tf.tuning_load(os.path.join(ROOT, "profiles", "r5_TunerResults.dat"))
    assert tf.plan_default_variant(1 << 24, 1, 2) == 33554432 and tf.plan_default_variant(1 << 25, 1, 1) == 33554432
    tf.tuning_clear()
    assert tf.plan_default_variant(1 << 24, 1, 2) == 0 and tf.plan_default_variant(1 << 25, 1, 1) == 0   # no longer rules of the source


def test_kernel_list_is_the_dispatch_table():
    names = tf.kernel_list()
    assert len(names) == len(set(names))
    for fam in ("colfft256_kernel<", "colfft256_wg_kernel<", "colfft512_wg_kernel<", "colfft512r_wg_kernel<", "colfft1024_wg_kernel<",
                "collat256_kernel<"):
        assert any(fam in k for k in names), fam
    assert "colfft::collat256_kernel<1, 1, 2, 2, 1>" in names and "colfft::collat256_kernel<0, 1, 1, 4, 2>" in names and "colfft::colfft512_wg_kernel<1, 2, false, true>" in names


def test_reference_style_mains_exist_and_build():
    """examples/bench_single (FFTBenchSinlge.cu protocol) and examples/tuner_single_fft (TunerSingleFFT.cu protocol) are built by
    build() from sources that include only the shim."""
    for name in ("bench_single", "tuner_single_fft"):
        assert os.access(os.path.join(ROOT, "examples", name), os.X_OK), name
        src = open(os.path.join(ROOT, "examples", name + ".cpp")).read()
        assert '#include "tensor_fft.hpp"' in src and "torch" not in src
