"""One process per GPU: the launcher bench.py (and the sharding tests) use when nobody else started the ranks.

`spawn_ranks` starts `world` copies of a command with the torch.distributed environment of one node
(RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT) and waits for them.  The parent never
imports torch and never touches a GPU: it only starts child processes (no exec of a process that has
initialised the GPU), forwards their output, and ends the others when one of them fails.
"""
from __future__ import annotations

import os
import signal
import socket
import subprocess
import sys
import time


def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def rank_env(rank: int, world: int, port: int, base=None) -> dict:
    env = dict(os.environ if base is None else base)
    env.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_WORLD_SIZE": str(world),
                "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL needs on this driver
    return env


def launched_by_a_launcher(world: int) -> bool:
    """True when torch.distributed.run (or spawn_ranks) already set this process up as one of `world` ranks."""
    return "RANK" in os.environ and int(os.environ.get("WORLD_SIZE", "1")) == world


def spawn_ranks(argv, world: int, env_extra=None, timeout=None) -> int:
    """Run `argv` once per rank; returns 0 iff every rank exited 0.  Rank 0 inherits stdout (its one JSON line is
    the job's output); every rank inherits stderr."""
    port = free_port()
    procs = []
    for rank in range(world):
        env = rank_env(rank, world, port)
        if env_extra:
            env.update(env_extra)
        procs.append(subprocess.Popen(list(argv), env=env, stdout=None if rank == 0 else subprocess.DEVNULL,
                                      start_new_session=True))
    deadline = None if timeout is None else time.time() + timeout
    rc = 0
    pending = list(procs)
    while pending:
        for p in list(pending):
            code = p.poll()
            if code is None:
                continue
            pending.remove(p)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
        if rc != 0 or (deadline is not None and time.time() > deadline):
            if rc == 0:
                rc = 124
            for p in pending:            # exactly the process groups started above
                try:
                    os.killpg(p.pid, signal.SIGTERM)
                except ProcessLookupError:
                    pass
            t_end = time.time() + 10
            for p in pending:
                try:
                    p.wait(timeout=max(0.1, t_end - time.time()))
                except subprocess.TimeoutExpired:
                    try:
                        os.killpg(p.pid, signal.SIGKILL)
                    except ProcessLookupError:
                        pass
                    p.wait()
            pending = []
            break
        time.sleep(0.05)
    return rc


def self_launch(script: str, args, world: int) -> int:
    """`python script args...` once per rank (used by `bench.py --gpus N` started without a launcher)."""
    return spawn_ranks([sys.executable, script] + list(args), world)
