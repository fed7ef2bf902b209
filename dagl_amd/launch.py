"""One process per GPU without an external launcher.

``python bench.py --gpus N`` must start its own N ranks (the driver's N = 1 command line has no
``torch.distributed.run`` in front of it, and the reference's own multi-GPU mode -- ``nn.DataParallel``,
DN_Gray/model/__init__.py:101-103 -- needs no launcher either).  ``spawn_ranks`` re-executes a script N times
with the rendezvous variables ``torch.distributed`` reads (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR /
MASTER_PORT), passes rank 0's stdout through (its ONE JSON line) and turns the first failing rank into the
job's exit code, stopping the others.  The ``torch.distributed.run`` path keeps working: a process that
already finds WORLD_SIZE in its environment never spawns.
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys
import time
from typing import Optional, Sequence


def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launched_by_torchrun() -> bool:
    """True inside a rank that a launcher (torch.distributed.run, or ``spawn_ranks`` below) started."""
    return "WORLD_SIZE" in os.environ and "RANK" in os.environ


def rank_env(rank: int, world: int, port: int, base: Optional[dict] = None) -> dict:
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), DAGL_SPAWNED="1")
    # the host driver only supports dmabuf IPC: RCCL across processes needs this (the image exports it; keep it)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


def spawn_ranks(script: str, argv: Sequence[str], world: int, timeout: Optional[float] = None, poll: float = 0.1) -> int:
    """Run ``python script *argv`` as ``world`` ranks on this node; returns the job's exit code (0 = every rank ended
    cleanly).  Rank 0 inherits stdout; the other ranks' stdout is dropped (they print nothing by contract); stderr of
    every rank is inherited.  A rank that fails (or the timeout) ends the others -- by their own PIDs."""
    if world < 1:
        raise ValueError("spawn_ranks: world must be >= 1")
    port = free_port()
    procs = []
    try:
        for r in range(world):
            procs.append(subprocess.Popen([sys.executable, script, *argv], env=rank_env(r, world, port),
                                          stdout=None if r == 0 else subprocess.DEVNULL))
        t0 = time.monotonic()
        code = 0
        while True:
            alive = 0
            for p in procs:
                rc = p.poll()
                if rc is None:
                    alive += 1
                elif rc != 0 and code == 0:
                    code = rc
            if code != 0 or alive == 0:
                break
            if timeout is not None and time.monotonic() - t0 > timeout:
                code = 124
                break
            time.sleep(poll)
        return code
    finally:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
