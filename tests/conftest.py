import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _run(cmd, timeout):
    import subprocess
    try:
        p = subprocess.run(cmd, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout, text=True)
        return p.returncode, p.stdout
    except subprocess.TimeoutExpired as e:
        return -9, "TIMEOUT after %ds\n%s" % (timeout, (e.stdout or b"").decode(errors="replace") if isinstance(e.stdout, bytes) else (e.stdout or ""))


@pytest.fixture(scope="session", autouse=True)
def multirank_runs(request, tmp_path_factory):
    """The multi-process jobs of tests/test_gpu_world2.py. Session-scoped and autouse so that they are launched FIRST, while this
    process has not touched the GPU yet (a process that has initialised the GPU must not fork + exec on this pool); every rank is a
    fresh interpreter. Does nothing unless a test of that module is selected (i.e. `-m gpu` on the GPU box)."""
    if not any("test_gpu_world2" in it.nodeid for it in request.session.items):
        yield {}
        return
    import subprocess
    py_exe = sys.executable
    out = {}
    port = 29400 + os.getpid() % 500
    # (1) two ranks of the MSM composition test on GPU 0, exchange over gloo
    d = str(tmp_path_factory.mktemp("world2_msm"))
    n_total = (1 << 16) + 3
    procs = [subprocess.Popen([py_exe, os.path.join(ROOT, "tests", "multirank", "msm_rank.py"), str(q), "2", str(port), str(n_total), d],
                              cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for q in range(2)]
    logs, rcs = [], []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=900)
        except subprocess.TimeoutExpired:
            p.kill()
            o, _ = p.communicate()
            o = "TIMEOUT\n" + (o or "")
        logs.append(o); rcs.append(p.returncode)
    out["msm"] = {"rc": rcs, "log": "\n---- rank ----\n".join(logs), "dir": d, "n_total": n_total}
    launch = [py_exe, "-m", "torch.distributed.run", "--nnodes=1", "--master-addr", "127.0.0.1"]
    # (2) bench.py at world 2 (gloo, both ranks on GPU 0): weak line + strong block (2^19 total) + KEM block + the sharded Laconic OT block,
    # small sizes, through torch.distributed.run ...
    rc, log = _run(launch + ["--nproc-per-node", "2", "--master-port", str(port + 1), "bench.py", "--gpus", "2", "--backend", "gloo", "--log2n", "18",
                             "--kem-log2n", "10", "--steps", "3", "--warmup", "1", "--cpu-log2n", "14", "--strong-log2n", "19", "--laconic-log2n", "12"], 1500)
    out["bench"] = {"rc": rc, "log": log}
    # (2b) ... and typed WITHOUT a launcher, the shape of the driver's recorded line (`python3 bench.py --gpus N --steps K --warmup W`,
    # BENCH_r05.json:cmd): bench.py starts its two ranks itself as a child job (keaki_amd/launch.py) and relays rank 0's line
    rc, log = _run([py_exe, "bench.py", "--gpus", "2", "--backend", "gloo", "--log2n", "18", "--kem-log2n", "10", "--steps", "3", "--warmup", "1",
                    "--cpu-log2n", "14", "--strong-log2n", "19", "--laconic-log2n", "12"], 1500)
    out["bench_self"] = {"rc": rc, "log": log}
    rc, log = _run([py_exe, "laconic_ot.py", "--gpus", "2", "--backend", "gloo", "--log2n", "10", "--check-single"], 1500)
    out["laconic_self"] = {"rc": rc, "log": log}
    # (3) laconic_ot.py at world 2
    rc, log = _run(launch + ["--nproc-per-node", "2", "--master-port", str(port + 2), "laconic_ot.py", "--gpus", "2", "--backend", "gloo", "--log2n", "15",
                             "--check-single"], 1500)
    out["laconic"] = {"rc": rc, "log": log}
    # (3b) four ranks (two rank bits in the sharded FK23), and three (not a power of two: the openings fall back to replicated)
    rc, log = _run(launch + ["--nproc-per-node", "4", "--master-port", str(port + 5), "laconic_ot.py", "--gpus", "4", "--backend", "gloo", "--log2n", "15",
                             "--check-single"], 1500)
    out["laconic4"] = {"rc": rc, "log": log}
    rc, log = _run(launch + ["--nproc-per-node", "3", "--master-port", str(port + 6), "laconic_ot.py", "--gpus", "3", "--backend", "gloo", "--log2n", "8",
                             "--check-single"], 1500)
    out["laconic3"] = {"rc": rc, "log": log}
    # (4) ONE rank under torch.distributed.run (the form the contract gives for N > 1, at N = 1; the driver's own N = 1 line is the plain
    # `python3 bench.py --gpus 1 --steps K --warmup W`, which the bench-line tests and smoke() run in-process)
    rc, log = _run(launch + ["--nproc-per-node", "1", "--master-port", str(port + 3), "bench.py", "--gpus", "1", "--log2n", "18", "--kem-log2n", "10",
                             "--steps", "2", "--warmup", "1", "--cpu-log2n", "14"], 1500)
    out["bench1"] = {"rc": rc, "log": log}
    rc, log = _run(launch + ["--nproc-per-node", "1", "--master-port", str(port + 4), "laconic_ot.py", "--gpus", "1", "--log2n", "9"], 1500)
    out["laconic1"] = {"rc": rc, "log": log}
    # (5) RCCL itself, with the one rank a one-GPU box allows: the calls of the N > 1 path (init "nccl", all-gather / all-to-all on device
    # buffers on the kernels' stream, barrier) and bench.py's whole distributed branch through --force-collectives
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port + 7), HSA_ENABLE_IPC_MODE_LEGACY="0")
    try:
        p = subprocess.run([py_exe, os.path.join(ROOT, "tests", "multirank", "rccl_world1.py")], cwd=ROOT, env=env, stdout=subprocess.PIPE,
                           stderr=subprocess.STDOUT, timeout=900, text=True)
        out["rccl1"] = {"rc": p.returncode, "log": p.stdout}
    except subprocess.TimeoutExpired as e:
        out["rccl1"] = {"rc": -9, "log": "TIMEOUT\n%s" % (e.stdout or "")}
    rc, log = _run(launch + ["--nproc-per-node", "1", "--master-port", str(port + 8), "bench.py", "--gpus", "1", "--force-collectives", "--backend", "nccl",
                             "--log2n", "18", "--kem-log2n", "10", "--steps", "3", "--warmup", "1", "--cpu-log2n", "14", "--strong-log2n", "18",
                             "--fk-log2d", "0", "--laconic-log2n", "0"], 1500)
    out["bench_rccl1"] = {"rc": rc, "log": log}
    # (6) EIGHT ranks on the one GPU (gloo): the driver's 8-GPU launch lines with everything but the transport real --
    # BASELINE config 4 at its size (bench.py's strong block: 2^26 points in total = 2^23 per rank, each rank with the window tables of its
    # chunk) and BASELINE config 5 at its size (laconic_ot.py: 2^20 receiver bits, FK23 openings sharded 8 ways at d = 2^21, the commit MSM
    # by point range, 2^21 encapsulations and 2^20 decapsulations by item range; rank 0 re-runs the un-sharded calls and compares bytes)
    rc, log = _run(launch + ["--nproc-per-node", "8", "--master-port", str(port + 9), "bench.py", "--gpus", "8", "--backend", "gloo", "--log2n", "20",
                             "--kem-log2n", "10", "--steps", "2", "--warmup", "1", "--cpu-log2n", "14", "--strong-log2n", "26",
                             "--fk-log2d", "0", "--laconic-log2n", "10"], 1500)
    out["bench8"] = {"rc": rc, "log": log}
    rc, log = _run(launch + ["--nproc-per-node", "8", "--master-port", str(port + 10), "laconic_ot.py", "--gpus", "8", "--backend", "gloo", "--log2n", "20",
                             "--check-single"], 1500)
    out["laconic8"] = {"rc": rc, "log": log}
    yield out


@pytest.fixture(scope="session")
def oc():
    """C oracle (ctypes) -- the checker, never the thing under test."""
    import oracle as _oc
    _oc.build()
    return _oc


@pytest.fixture(scope="session")
def py():
    import bn254_py
    return bn254_py


@pytest.fixture(scope="session")
def hip():
    from keaki_amd.hip import KeakiHip
    h = KeakiHip(0)  # raises KeakiHipError when the extension or the GPU is missing: no fallback
    yield h
    h.close()


class SplitMix64:
    """Deterministic u64 stream (SURVEY.md section 8d synthetic inputs)."""

    def __init__(self, seed):
        self.s = seed & 0xFFFFFFFFFFFFFFFF

    def next(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        return z ^ (z >> 31)


R_MOD = 21888242871839275222246405745257275088548364400416034343698204186575808495617


def rand_fr_ints(n, seed):
    g = SplitMix64(seed)
    out = []
    while len(out) < n:
        limbs = [g.next() for _ in range(4)]
        limbs[3] &= 0xFFFFFFFFFFFFFFFF >> 2
        v = sum(l << (64 * i) for i, l in enumerate(limbs))
        if v < R_MOD:
            out.append(v)
    return out


@pytest.fixture(scope="session")
def rand_fr():
    return rand_fr_ints
