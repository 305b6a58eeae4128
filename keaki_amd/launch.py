"""Self-launch of the one-process-per-GPU jobs: `python3 bench.py --gpus N` / `python3 laconic_ot.py --gpus N` typed WITHOUT
torch.distributed.run must still run N ranks.

The parent -- this process, before it has imported torch or made any HIP call -- starts
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P <script> <same args>
as a CHILD process, relays what the child prints (rank 0's one JSON line goes to stdout as the child's stdout is inherited) and returns
the child's exit code. Nothing here replaces a process image (no exec of anything from a process that has initialised the GPU: on this
pool that takes the machine down), and the parent never touches the GPU at all.

Imports nothing beyond the standard library.
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys


def under_launcher() -> bool:
    """True inside a rank started by torch.distributed.run (or any launcher that exports the rendezvous variables)"""
    return "WORLD_SIZE" in os.environ and "RANK" in os.environ


def free_port() -> int:
    s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    try:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]
    finally:
        s.close()


def launch_command(script: str, argv: list[str], nproc: int, port: int | None = None) -> list[str]:
    """the driver's own launch line for `script argv` on `nproc` ranks of this node"""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
            "--master-port", str(port if port is not None else free_port()), script] + list(argv)


def self_launch(script: str, argv: list[str], nproc: int) -> int:
    """run `script argv` as `nproc` ranks in a child torch.distributed.run; returns the child's exit code. stdout / stderr are inherited, so
    rank 0's JSON line appears on this process's stdout exactly once."""
    if "torch" in sys.modules:
        raise RuntimeError("self_launch must run before torch is imported (the parent must never initialise the GPU)")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # the host driver supports dmabuf IPC only (RCCL between processes)
    env.setdefault("OMP_NUM_THREADS", "1")                     # torch.distributed.run would set it (with a warning) anyway
    env["KEAKI_SELF_LAUNCHED"] = "1"
    cmd = launch_command(os.path.abspath(script), argv, nproc)
    print("[launch] %s" % " ".join(cmd), file=sys.stderr, flush=True)
    proc = subprocess.Popen(cmd, env=env, cwd=os.getcwd())
    try:
        return proc.wait()
    except KeyboardInterrupt:
        proc.terminate()
        try:
            return proc.wait(timeout=30)
        except subprocess.TimeoutExpired:
            proc.kill()
            return proc.wait()
