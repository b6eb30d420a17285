"""bench.py on the GPU box: the one-JSON-line contract at N = 1 on a short run, and a REHEARSAL of the N = 2 layer split on one GPU
(both ranks on device 0, the hop through gloo + host memory): per-rank decode plans, the boundary activation h = ffn_inp + ffn_down,
the completion message from the last stage to the first.  The real N > 1 run (one GPU per rank, RCCL) is the driver's."""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _json_line(out):
    lines = [ln for ln in out.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


def test_bench_contract_single_gpu():
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--steps", "8", "--warmup", "2", "--no-cpu-baseline", "--no-pp"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    j = _json_line(r.stdout)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in j, key
    assert j["n_gpus"] == 1 and j["steps"] == 8 and j["warmup"] == 2 and j["value"] > 50 and j["unit"] == "tok/s"
    rf = j["roofline"]
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and 0.05 < rf["frac"] < 1.0 and rf["kernel"].startswith("k_plan")
    assert abs(rf["achieved"] - rf["algorithmic_bytes_per_launch"] / rf["avg_launch_us"] / 1e3) < 0.01 * rf["achieved"]      # GB/s = bytes / us / 1e3
    assert "workload" in j["config"] and "model" not in j["config"]


def test_two_rank_layer_split_rehearsal():
    env = dict(os.environ, MI355Q_BENCH_REHEARSE="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(ROOT / "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--no-cpu-baseline", "--no-pp"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, (r.stdout[-1500:] + r.stderr[-3000:])
    j = _json_line(r.stdout)
    assert j["n_gpus"] == 2 and j["steps"] == 6 and j["value"] > 10 and j["scaling"] == "strong"
    assert "layer split over 2 GPUs" in j["config"]["parallelism"]
