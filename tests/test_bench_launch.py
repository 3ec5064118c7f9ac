"""bench.py --gpus N without a launcher starts its ranks itself: the spawn logic on CPU, two ranks
over gloo with the numpy engine from tests/; and the failure modes: a rank that dies, a box with too few GPUs."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SCRIPT = os.path.join(ROOT, "tests", "helpers", "launch_rank.py")


def test_self_launch_runs_two_ranks_and_relays_rank_zero(capfd):
    import bench

    rc = bench.self_launch(2, SCRIPT, ["--gpus", "2"])
    out = capfd.readouterr().out
    assert rc == 0, out
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1                         # ONE JSON line, rank 0's
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["rel_l2"] < 1.5e-3


def test_self_launch_reports_a_failed_rank(capfd):
    import bench

    rc = bench.self_launch(2, SCRIPT, ["--gpus", "2", "--fail-rank", "1"])
    captured = capfd.readouterr()
    assert rc != 0
    assert "2-rank run failed" in captured.err


def test_self_launch_refuses_more_ranks_than_gpus(capfd):
    import bench

    assert bench.self_launch(2, SCRIPT, ["--gpus", "2"], nproc_visible=1) == 2
    captured = capfd.readouterr()
    assert "needs 2 GPUs" in captured.err and "{" not in captured.out          # nothing was started


def test_bench_gpus_2_on_a_box_without_two_gpus_fails_clearly():
    """The driver's own command line, no launcher: on this box (no GPU, or one) it must fail with a clear message and a
    non-zero exit code before any rank starts."""
    import torch

    if torch.cuda.device_count() >= 2:
        import pytest

        pytest.skip("two GPUs present")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    assert "needs 2 GPUs" in r.stderr and not r.stdout.strip().startswith("{")
