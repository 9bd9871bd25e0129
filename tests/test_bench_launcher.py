"""`python bench.py --gpus N` must start its N ranks itself (a plain --gpus 8 invocation used to exit).
CPU: the launcher starts fresh rank processes under torch.distributed.run, they rendezvous on 127.0.0.1 over gloo and
rank 0's single line is relayed.  GPU (one-GPU box): the same path with the real decode step, two ranks sharing the
card over gloo -- the rehearsal the 8-GPU run cannot get here."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, timeout):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, BENCH, *args], capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout  # exactly ONE JSON line, from rank 0
    return json.loads(lines[0])


def test_launcher_command_is_the_drivers_launch_line():
    sys.path.insert(0, ROOT)
    import bench
    cmd = bench.launcher_command(4, ["--gpus", "4", "--steps", "3"], 12345)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert "--nproc-per-node=4" in cmd and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-4:] == ["--gpus", "4", "--steps", "3"] and cmd[-5] == BENCH


def test_plain_gpus_2_starts_two_ranks_and_relays_one_line():
    out = _run(["--gpus", "2", "--rehearse-launch"], timeout=300)
    assert out["n_gpus"] == 2 and out["ranks_seen"] == 2 and out["rehearsal"] is True


def test_plain_gpus_1_needs_no_launcher():
    out = _run(["--gpus", "1", "--rehearse-launch"], timeout=120)
    assert out["n_gpus"] == 1 and out["ranks_seen"] == 1


@pytest.mark.gpu
def test_two_ranks_share_the_gpu_over_gloo():
    out = _run(["--gpus", "2", "--backend", "gloo", "--steps", "3", "--warmup", "1", "--workload", "c3",
                "--no-cpu-baseline", "--no-roofline"], timeout=600)
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["value"] > 0
    assert out["config"]["rows_per_gpu"] == 256 and out["scaling"] == "weak"
    # the N > 1 line proves what the collective saw
    coll = out["collective"]
    assert coll["backend"].startswith("gloo") and coll["ranks_seen"] == 2
    assert coll["gathered_token_ids"] == 2 * 256 and coll["gathered_token_ids_valid"] == 2 * 256
    assert coll["gather_us_per_step"] > 0 and coll["ms_per_step_without_gather"] > 0
    assert out["repeat"]["regions"] >= 1 and out["config"]["rows_total"] == 512
