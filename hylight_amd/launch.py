"""One process per GPU, started by the product itself.

The reference fans every stage out on its own (`xargs -i -P threads`, script/utils.py:65); the counterpart here is
`spawn_ranks`: the entry points (`bench.py --gpus N`, `python -m hylight_amd.driver --gpus N`) start N copies of their
own command line with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set, BEFORE the parent has made a
single GPU call (the parent never touches the GPU at all: it waits and relays the exit status).  A run that was started
by an external launcher (torchrun sets the same variables) is recognised by WORLD_SIZE being present and is left alone.
"""
from __future__ import annotations

import os
import signal
import socket
import subprocess
import sys
import time


def launched():
    """True inside a rank process (ours or an external launcher's)."""
    return "WORLD_SIZE" in os.environ and "RANK" in os.environ


def rank_env():
    """(rank, world, local_rank) of this process; (0, 1, 0) outside a multi-process run."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", os.environ.get("RANK", "0"))))


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(n, cmd, env_extra=None, poll_s=0.2):
    """Run `cmd` (argv list) as n rank processes on this node and wait for them.  Every rank inherits stdout / stderr
    (rank 0 prints the result).  Returns 0 when every rank exited with 0; otherwise the first non-zero status seen -
    the remaining ranks (which would wait for the dead one in their next collective) are ended by PID."""
    if n < 1:
        raise ValueError("need at least one rank")
    base = dict(os.environ)
    base.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                HL_LAUNCHER="self")
    # (HSA_ENABLE_IPC_MODE_LEGACY and the other runtime switches are INHERITED, never set here: a host whose driver only
    #  supports dmabuf IPC exports HSA_ENABLE_IPC_MODE_LEGACY=0 for RCCL itself - as this pool does - and the ranks see it)
    if env_extra:
        base.update(env_extra)
    procs = []
    try:
        for r in range(n):
            env = dict(base, RANK=str(r), LOCAL_RANK=str(r))
            procs.append(subprocess.Popen(list(cmd), env=env))
        rc = 0
        live = set(range(n))
        while live:
            for r in sorted(live):
                code = procs[r].poll()
                if code is None:
                    continue
                live.discard(r)
                if code != 0 and rc == 0:
                    rc = code
                    sys.stderr.write(f"[launch] rank {r} exited with status {code}: ending the other ranks\n")
                    for o in sorted(live):
                        procs[o].terminate()
            if live:
                time.sleep(poll_s)
        return rc if rc >= 0 else 128 - rc               # killed by a signal: the shell's convention
    finally:
        for p in procs:                                   # (an exception in here, e.g. KeyboardInterrupt)
            if p.poll() is None:
                p.send_signal(signal.SIGTERM)
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()


def init_process_group(device_index=None, backend=None):
    """torch.distributed for this rank (world > 1 only).  backend "nccl" = RCCL over xGMI, one GPU per rank;
    HL_BACKEND=gloo lets several ranks share a card (the one-GPU rehearsal of the N > 1 flow) or run on the CPU (tests).
    Returns (rank, world, backend)."""
    import torch.distributed as dist
    rank, world, _ = rank_env()
    backend = backend or os.environ.get("HL_BACKEND") or os.environ.get("HL_BENCH_BACKEND") or "nccl"
    if world == 1:
        return rank, world, None
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend == "nccl":
        import torch
        dist.init_process_group("nccl", device_id=torch.device("cuda", device_index or 0))
    else:
        dist.init_process_group(backend)
    assert dist.get_world_size() == world and dist.get_rank() == rank
    return rank, world, backend
