"""bench.py itself, in its shortest form, so that a change which breaks the bench breaks the suite: one rank, and two ranks started by
bench.py's own launcher (they share the box's one GPU; the gather then runs over gloo). The line's contract fields are checked, not
its numbers."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config")


def run_bench(*args, env=None):
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=900, cwd=ROOT,
                         env=dict(os.environ, **(env or {})))
    assert res.returncode == 0, res.stdout[-1500:] + res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout[-1500:]                              # ONE JSON line
    return json.loads(lines[0])


def test_bench_one_rank_short():
    d = run_bench("--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-tree", "--headline-only")
    for k in CONTRACT:
        assert k in d, k
    assert d["metric"] == "Wormhole proofs/sec" and d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1
    assert d["value"] > 0 and d["higher_is_better"] is True and d["scaling"] == "weak" and d["dtype"] == "u64" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "roofline" in d and d["roofline"]["bound"] in ("valu", "hbm") and 0 < d["roofline"]["frac"] < 1


def test_bench_two_ranks_short():
    # (bench.py refuses more ranks than visible GPUs unless told it is a rehearsal)
    d = run_bench("--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-tree", "--headline-only", "--no-ntt", env={"QPGPU_BENCH_BACKEND": "gloo"})
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["scaling"] == "weak"
