"""The randomised cross-checks of tests/fuzz/ (HIP path vs the oracle over random sizes, scalar shapes and CALL SEQUENCES on one
context -- the state-leak fuzzers for the stateful encapsulation policy) collected by pytest with fixed seeds and bounded rounds, so
that the driver's `pytest -m gpu` runs them. The scripts stay runnable by hand with more rounds: python tests/fuzz/fuzz_mixed.py 200 5"""
import os
import runpy
import sys

import pytest

pytestmark = pytest.mark.gpu
FUZZ = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fuzz")


def _run(script, *argv):
    old = sys.argv
    env = {k: v for k, v in os.environ.items() if k.startswith("KEAKI_")}      # the scripts flip KEAKI_* switches
    sys.argv = [script] + [str(a) for a in argv]
    try:
        with pytest.raises(SystemExit) as e:
            runpy.run_path(os.path.join(FUZZ, script), run_name="__main__")
        assert e.value.code in (0, None), "%s %s reported mismatches" % (script, argv)
    finally:
        sys.argv = old
        for k in [k for k in os.environ if k.startswith("KEAKI_")]:
            del os.environ[k]
        os.environ.update(env)


@pytest.mark.parametrize("seed", [2024, 3])
def test_fuzz_msm(seed):
    _run("fuzz_msm.py", 4, seed)


@pytest.mark.parametrize("seed", [7])
def test_fuzz_kem(seed):
    _run("fuzz_kem.py", 6, seed)


@pytest.mark.parametrize("seed", [1, 2])
def test_fuzz_mixed_call_sequences(seed):
    _run("fuzz_mixed.py", 40, seed)


@pytest.mark.parametrize("seed,members", [(1, 4), (2, 3), (3, 8)])
def test_fuzz_device_group_call_sequences(seed, members):
    _run("fuzz_group.py", 30, seed, members)
