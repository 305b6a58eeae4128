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
import signal
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
    # Whatever ends this process early must end the job: SIGTERM (a driver's timeout), Ctrl-C and a closed terminal are passed on to the child
    # launcher, which shuts its ranks down; should this process be killed outright (SIGKILL cannot be caught), the kernel sends the child SIGTERM
    # on our death (PR_SET_PDEATHSIG). The child stays in OUR process group, so a group-wide kill reaches it as well. No rank is left holding a GPU.
    import ctypes
    libc = ctypes.CDLL(None, use_errno=True)

    def die_with_parent():                     # runs in the child between fork and exec
        libc.prctl(1, int(signal.SIGTERM), 0, 0, 0)          # PR_SET_PDEATHSIG

    proc = subprocess.Popen(cmd, env=env, cwd=os.getcwd(), preexec_fn=die_with_parent)

    def stop_group(sig):
        try:
            proc.send_signal(sig)
        except (ProcessLookupError, PermissionError):
            pass

    def on_signal(signum, _frame):
        stop_group(signal.SIGTERM)
        try:
            proc.wait(timeout=30)
        except subprocess.TimeoutExpired:
            stop_group(signal.SIGKILL)
        raise SystemExit(128 + signum)

    previous = {}
    for sg in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP):
        try:
            previous[sg] = signal.signal(sg, on_signal)
        except (ValueError, OSError):          # not the main thread: the caller keeps its own handling
            pass
    try:
        return proc.wait()
    finally:
        for sg, h in previous.items():
            signal.signal(sg, h)
        if proc.poll() is None:                # leaving by an exception: take the job down with us
            stop_group(signal.SIGTERM)
