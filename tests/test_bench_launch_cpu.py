"""bench.py's N-rank launcher (CPU): `--gpus N` outside torch.distributed.run must start N ranks as a CHILD process of
`python -m torch.distributed.run`, relay rank 0's JSON line and exit with the child's code -- the reference's counterpart is
the Ray fan-out of evaluation/eval_vicuna.py:39-68.  `--dry-launch` prints the command; `--launch-selftest` runs the
spawn / rendezvous (gloo) / reduce / relay plumbing with made-up numbers, no GPU."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(*args, env=None, timeout=240):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH, *args], capture_output=True, text=True, timeout=timeout, env=e)


def test_dry_launch_prints_the_torchrun_command():
    r = _run("--gpus", "8", "--steps", "7", "--warmup", "3", "--dry-launch")
    assert r.returncode == 0, r.stderr
    d = json.loads(r.stdout.strip().splitlines()[-1])
    cmd = d["launch"]
    assert d["n_gpus"] == 8 and cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and os.path.samefile(cmd[cmd.index("--master-port") + 2], BENCH)
    tail = cmd[cmd.index("--master-port") + 3:]
    assert tail == ["--gpus", "8", "--steps", "7", "--warmup", "3"]          # the ranks get the flags, not --dry-launch


def test_single_gpu_or_inside_a_launcher_does_not_spawn():
    assert json.loads(_run("--gpus", "1", "--dry-launch").stdout.strip().splitlines()[-1])["launch"] is None
    r = _run("--gpus", "4", "--dry-launch", env={"RANK": "0", "WORLD_SIZE": "4", "LOCAL_RANK": "0"})
    assert r.returncode == 0 and json.loads(r.stdout.strip().splitlines()[-1])["launch"] is None


def test_two_ranks_are_spawned_reduced_and_relayed():
    r = _run("--gpus", "2", "--launch-selftest")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                                          # only rank 0 prints
    d = lines[0]
    assert d["n_gpus"] == 2 and d["gpus_flag"] == 2
    assert d["per_rank"] == [{"rank": 0, "tokens": 100, "seconds": 1.0, "static_sam_distribution_ms": 5.0},
                             {"rank": 1, "tokens": 200, "seconds": 2.0, "static_sam_distribution_ms": 6.0}]    # every rank's own figures
    assert abs(d["value"] - 300 / 2.0) < 1e-9                                 # SUM of tokens / MAX of time
    assert d["rccl_ranks_seen"] == 2                                          # the size of the all_gather behind `value`


def test_more_ranks_than_gpus_is_one_clear_line_and_nothing_is_spawned():
    """`--gpus N` on a node with fewer GPUs must not start N ranks that die one by one (this container has none)."""
    import torch
    if torch.cuda.device_count() >= 8:
        import pytest
        pytest.skip("this node has 8 GPUs")
    r = _run("--gpus", "8", "--steps", "2", "--warmup", "1")
    assert r.returncode == 2 and "needs 8 visible GPUs" in r.stderr and "nothing was launched" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_child_failure_becomes_the_exit_code():
    r = _run("--gpus", "2", "--launch-selftest", env={"SAMD_SELFTEST_FAIL_RANK": "1"})   # rank 1 exits with 3
    assert r.returncode != 0 and not [l for l in r.stdout.splitlines() if l.startswith("{")]
