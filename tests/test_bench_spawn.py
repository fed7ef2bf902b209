"""``python bench.py --gpus N`` starts its own N ranks (dagl_amd/launch.py) -- the driver issues exactly that command.

CPU: the spawner with 2 gloo workers and a stub step (``--stub``: same rendezvous / barrier / MAX-over-ranks / one-JSON-line
protocol).  GPU (one MI355X): the spawn path at ``--gpus 1`` through RCCL, and the refusal of ``--gpus 2`` on a 1-GPU box."""
import json
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(REPO, "bench.py")


def _clean_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "DAGL_SPAWNED",
                                                             "DAGL_BENCH_FORCE_SPAWN", "DAGL_BENCH_FORCE_DIST")}
    env.update(extra)
    return env


def _last_json(stdout):
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, stdout          # ONE line, from rank 0 only
    return json.loads(lines[0])


@pytest.mark.parametrize("n", [2, 3])
def test_spawner_runs_n_gloo_ranks_with_a_stub_step(n):
    r = subprocess.run([sys.executable, BENCH, "--gpus", str(n), "--stub", "--steps", "4", "--warmup", "1"], env=_clean_env(),
                       capture_output=True, text=True, timeout=300, cwd=REPO)
    assert r.returncode == 0, r.stderr[-2000:]
    line = _last_json(r.stdout)
    assert line["n_gpus"] == n and line["n_ranks_seen"] == n and line["steps"] == 4
    assert line["devices"] == [f"cpu:{i}" for i in range(n)]
    assert line["value"] > 0 and line["stub"] is True


def test_forced_spawn_at_one_rank():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--stub", "--steps", "2", "--warmup", "0"],
                       env=_clean_env(DAGL_BENCH_FORCE_SPAWN="1"), capture_output=True, text=True, timeout=300, cwd=REPO)
    assert r.returncode == 0, r.stderr[-2000:]
    assert _last_json(r.stdout)["n_ranks_seen"] == 1


def test_torchrun_launch_still_works_and_a_wrong_rank_count_is_refused():
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", "29647", BENCH, "--gpus", "2", "--stub", "--steps", "2", "--warmup", "0"],
                       env=_clean_env(), capture_output=True, text=True, timeout=300, cwd=REPO)
    assert r.returncode == 0, r.stderr[-2000:]
    assert _last_json(r.stdout)["n_ranks_seen"] == 2
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--stub"], env=_clean_env(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1"),
                       capture_output=True, text=True, timeout=120, cwd=REPO)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr


def test_a_failing_rank_fails_the_job():
    from dagl_amd.launch import spawn_ranks
    script = os.path.join(REPO, "tests", "_spawn_probe.py")
    with open(script, "w") as f:
        f.write("import os, sys, time\nr = int(os.environ['RANK'])\nif r == 1:\n    sys.exit(7)\ntime.sleep(30)\n")
    try:
        assert spawn_ranks(script, [], 2, timeout=60) == 7       # rank 1 fails at once, rank 0 is stopped (not waited for)
    finally:
        os.remove(script)


def test_more_gpus_than_visible_is_refused_with_a_clear_message():
    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    want = have + 1 if have else 2
    r = subprocess.run([sys.executable, BENCH, "--gpus", str(want)], env=_clean_env(), capture_output=True, text=True,
                       timeout=300, cwd=REPO)
    assert r.returncode != 0
    assert f"{want} GPUs requested, {have} visible" in r.stderr, r.stderr[-500:]


@pytest.mark.gpu
@pytest.mark.parametrize("train", [False, True])
def test_gpu_spawn_path_through_rccl_at_one_rank(train):
    cmd = [sys.executable, BENCH, "--gpus", "1", "--steps", "2", "--warmup", "1"]
    cmd += ["--train", "--batch", "2", "--crop", "64"] if train else ["--size", "64", "--no-quality", "--no-cpu-baseline", "--no-extra"]
    r = subprocess.run(cmd, env=_clean_env(DAGL_BENCH_FORCE_SPAWN="1", HSA_ENABLE_IPC_MODE_LEGACY="0"), capture_output=True,
                       text=True, timeout=900, cwd=REPO)
    assert r.returncode == 0, r.stderr[-2000:]
    line = _last_json(r.stdout)
    assert line["n_gpus"] == 1 and line["n_ranks_seen"] == 1 and len(line["devices"]) == 1 and line["value"] > 0
    if train:
        assert line["allreduce_ms"] is not None            # the process group (RCCL) was really initialised
