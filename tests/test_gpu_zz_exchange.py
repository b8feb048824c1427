"""bench.py's multi-GPU step on the one GPU a test box has: a one-rank RCCL process group.  Kept in a file of its own
that sorts behind the other GPU tests: it starts a second process on the GPU and opens a rendezvous port."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_bench_exchange_step_with_a_one_rank_rccl_group():
    """The N > 1 step of bench.py (decode, then the all-gather of the label records through torch.distributed over
    RCCL) with a one-rank process group: the records that come back are the rank's own output and the line reports
    the exchange.  Runs in a process of its own (a process group belongs to its process)."""
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    run = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--utts", "96", "--steps", "3", "--warmup", "1",
                          "--no-cpu-baseline", "--no-h2d", "--no-other-configs", "--exchange-at-one"],
                         capture_output=True, text=True, timeout=300, env=env)
    if run.returncode != 0 and "line" not in run.stdout and any(
            w in run.stderr for w in ("init_process_group", "Address already in use", "ncclCommInitRank", "ProcessGroupNCCL")):
        pytest.skip("the one-rank process group could not be formed on this box: " + run.stderr.strip().splitlines()[-1][:200])
    assert run.returncode == 0, run.stderr[-2000:]
    line = json.loads(run.stdout.strip().splitlines()[-1])
    assert "all-gather of label records" in line["config"]["parallelism"]
    assert line["seq_accuracy_vs_truth"] == 1.0 and line["n_gpus"] == 1 and line["value"] > 0
