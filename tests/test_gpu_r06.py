"""Round 6: bench.py launched the way the driver launches it for N > 1 (plain `python bench.py --gpus N`), `--check`."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _clean_env():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR", "TORCHELASTIC_RUN_ID", "GROUP_RANK"):
        env.pop(k, None)
    return env


def test_bench_py_plain_command_launches_itself_for_n_gt_1():
    """VERDICT r05 "what's weak" 7: `python bench.py --gpus 2 ...` with NO launcher around it (WORLD_SIZE unset) re-executes
    itself under torch.distributed.run (two gloo ranks sharing this GPU here) and prints exactly one JSON line, n_gpus 2."""
    from test_gpu_r03 import _bench_line
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--share-device",
                        "--steps", "2", "--warmup", "1", "--batch", "512", "--no-cpu-baseline"],
                       env=_clean_env(), capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    line = _bench_line(r.stdout)
    assert line["n_gpus"] == 2 and line["config"]["global_batch"] == 1024 and line["config"]["parallelism"] == "dp2"
    assert line["steps"] == 2 and np.isfinite(line["value"]) and line["value"] > 0


def test_bench_py_check_flag_small_batch():
    """`bench.py --check` (untimed value check of the workload as a graph replay) reports per-quantity errors in the line."""
    from test_gpu_r03 import _bench_line
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--batch", "1024",
                        "--no-cpu-baseline", "--check"], env=_clean_env(), capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    line = _bench_line(r.stdout)
    assert line["check"]["passed"] is True, line["check"]
    assert any(k.startswith("dW.real (+ KL accumulate)") for k in line["check"]["max_err_over_max_ref"])
