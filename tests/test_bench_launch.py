"""`python bench.py --gpus N` as the driver invokes it: the process is a parent that starts the N ranks itself
(bench.launch_ranks), relays rank 0's JSON line and the exit code.  The reference is single-device
(/root/reference/train.py:209-212); the data-parallel launch is this repo's own design."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, timeout=900):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=env,
                          timeout=timeout, cwd=ROOT)


def test_parent_relays_child_failure():
    """No HIP device here: every rank stops with bench.py's own message; the parent must hand the failure on (rc != 0, no
    result line) instead of swallowing it -- and must get that far without a GPU call of its own."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("the failure leg needs a box without a GPU")
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "1", "--no-cpu-baseline"], timeout=600)
    assert r.returncode != 0
    assert "needs a HIP device" in r.stderr
    assert '"metric"' not in r.stdout


@pytest.mark.gpu
def test_bench_two_ranks_end_to_end():
    """Two ranks on the one GPU over gloo (the rehearsal hooks of bench.py): the parent starts them, one JSON line comes
    back with the whole-job figures of a 2-rank run."""
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "2", "--no-cpu-baseline"],
             {"SELD_BENCH_BACKEND": "gloo", "SELD_BENCH_SINGLE_DEVICE": "1"})
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2 and out["dist_backend"] == "gloo"
    assert out["config"]["global_batch"] == 64 and out["config"]["parallelism"] == "dp2"
    assert out["steps"] == 3 and out["value"] > 0
    import math
    assert math.isfinite(out["loss"]) and 0 < out["loss"] < 10
    assert "cpu_baseline" not in out          # N > 1: no CPU leg
