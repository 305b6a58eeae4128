"""keaki_amd/launch.py on CPU: `python3 bench.py --gpus N` / `laconic_ot.py --gpus N` typed without torch.distributed.run start their ranks as
a CHILD job from a parent that never imports torch (so never initialises the GPU), relay rank 0's line and return the child's exit code.
The GPU form of this test is tests/test_gpu_world2.py::test_world2_bench_self_launched_carries_the_whole_metric."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable] + args, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)
    return p.returncode, p.stdout, p.stderr


def test_launch_command_is_the_contracts_line():
    from keaki_amd.launch import launch_command, under_launcher
    cmd = launch_command("bench.py", ["--gpus", "4", "--steps", "20"], 4, port=29555)
    assert cmd[1:] == ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr", "127.0.0.1", "--master-port", "29555",
                       "bench.py", "--gpus", "4", "--steps", "20"]
    assert under_launcher() is ("WORLD_SIZE" in os.environ and "RANK" in os.environ)


def test_self_launch_brings_up_the_ranks_and_relays_one_line():
    rc, out, err = _run([os.path.join("tests", "multirank", "echo_rank.py"), "--gpus", "2"])
    assert rc == 0, err[-2000:]
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j == {"world": 2, "sum": 3, "self_launched": True}
    assert "torch imported in parent: False" in err and "[launch]" in err


def test_self_launch_returns_the_childs_failure():
    rc, out, err = _run([os.path.join("tests", "multirank", "echo_rank.py"), "--gpus", "2", "--fail-rank", "1"])
    assert rc != 0, "a failing rank must fail the parent"


def test_bench_and_laconic_self_launch_then_fail_loudly_without_a_gpu():
    """no GPU here: the child ranks must refuse (there is no CPU fallback), and the parent must hand the failure back -- not SystemExit
    with 'must be launched through torch.distributed.run' as until round 5"""
    import torch
    if torch.cuda.device_count() > 0:
        import pytest
        pytest.skip("a GPU is present: the GPU suite runs the real thing")
    for script, needle in (("bench.py", "needs an MI355X"), ("laconic_ot.py", "needs an MI355X")):
        rc, out, err = _run([script, "--gpus", "2", "--backend", "gloo"])
        assert rc != 0
        assert "[launch]" in err and "torch.distributed.run" in err and "--nproc-per-node 2" in err
        assert "must be launched through" not in err + out
        assert needle in err + out


def test_terminating_the_parent_takes_the_ranks_down():
    """a driver's timeout sends SIGTERM to `python3 bench.py --gpus N`: the child launcher and its ranks (a process group of their own) must go too"""
    import signal
    import time
    script = os.path.join(ROOT, "tests", "multirank", "sleepy_rank.py")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    marker = os.path.join(ROOT, "tests", "multirank", ".sleepy_%d" % os.getpid())
    p = subprocess.Popen([sys.executable, script, "--gpus", "2", "--marker", marker], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    try:
        deadline = time.time() + 120
        while time.time() < deadline and not (os.path.exists(marker + ".0") and os.path.exists(marker + ".1")):
            time.sleep(0.2)
        assert os.path.exists(marker + ".0") and os.path.exists(marker + ".1"), "the ranks never came up"
        pids = [int(open(marker + ".%d" % q).read()) for q in range(2)]
        p.send_signal(signal.SIGTERM)
        rc = p.wait(timeout=60)
        assert rc == 128 + signal.SIGTERM
        time.sleep(1.0)
        for pid in pids:
            alive = True
            try:
                os.kill(pid, 0)
            except ProcessLookupError:
                alive = False
            assert not alive, "rank process %d survived its parent" % pid
    finally:
        if p.poll() is None:
            p.kill()
        for q in range(2):
            try:
                os.remove(marker + ".%d" % q)
            except OSError:
                pass


def test_killing_the_parent_outright_takes_the_ranks_down_too():
    """SIGKILL cannot be caught: the child launcher dies with its parent through PR_SET_PDEATHSIG and takes its ranks along"""
    import signal
    import time
    script = os.path.join(ROOT, "tests", "multirank", "sleepy_rank.py")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    marker = os.path.join(ROOT, "tests", "multirank", ".sleepyk_%d" % os.getpid())
    p = subprocess.Popen([sys.executable, script, "--gpus", "2", "--marker", marker], cwd=ROOT, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    try:
        deadline = time.time() + 120
        while time.time() < deadline and not (os.path.exists(marker + ".0") and os.path.exists(marker + ".1")):
            time.sleep(0.2)
        assert os.path.exists(marker + ".0") and os.path.exists(marker + ".1"), "the ranks never came up"
        pids = [int(open(marker + ".%d" % q).read()) for q in range(2)]
        p.send_signal(signal.SIGKILL)
        p.wait(timeout=30)
        gone = False
        for _ in range(150):                  # the launcher needs a moment to stop its workers
            gone = True
            for pid in pids:
                try:
                    os.kill(pid, 0)
                    gone = False
                except ProcessLookupError:
                    pass
            if gone:
                break
            time.sleep(0.2)
        assert gone, "rank processes survived a killed parent"
    finally:
        if p.poll() is None:
            p.kill()
        for q in range(2):
            try:
                os.remove(marker + ".%d" % q)
            except OSError:
                pass
