"""CPU suite: the C-ABI library builds, loads and exports every symbol include/keaki_hip.h declares;
the device constants agree with the oracle's independent derivation; bench input generation is sane.
No compute call is made here (no GPU on the CPU box)."""
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "keaki_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(keaki_hip_[a-z0-9_]+)\s*\(", hdr)))


def test_header_symbols_are_exported_and_bound():
    from keaki_amd import hip
    lib = hip.load_library()      # raises if libkeaki_hip.so is not built: no fallback
    syms = declared_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(lib, s), "libkeaki_hip.so does not export %s" % s
    assert sorted(hip.EXPORTS) == syms, "keaki_amd/hip.py EXPORTS and include/keaki_hip.h disagree"
    assert b"gfx950" in lib.keaki_hip_version()


def test_library_carries_gfx950_code_only():
    path = os.path.join(ROOT, "keaki_amd", "libkeaki_hip.so")
    blob = open(path, "rb").read()
    targets = set(re.findall(rb"amdgcn-amd-amdhsa--(gfx[0-9a-z]+)", blob))
    assert targets == {b"gfx950"}, targets


def test_no_device_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from keaki_amd.hip import KeakiHip, KeakiHipError
    with pytest.raises(KeakiHipError):
        KeakiHip(0)


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under keaki_amd/ may reference it."""
    bad = []
    for dp, _, fs in os.walk(os.path.join(ROOT, "keaki_amd")):
        for f in fs:
            if f.endswith((".py", ".hip", ".cuh", ".h", ".hpp", ".cpp", "Makefile")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                if re.search(r"(import\s+oracle|from\s+oracle|bn254_py|bn254_ref|libbn254_oracle|oracle/)", txt):
                    if f == "gen_constants.py" and "oracle/" in txt and "import" not in txt.split("oracle/")[0][-20:]:
                        continue
                    bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_device_constants_match_oracle(py):
    """keaki_amd/csrc/gen_constants.py derives the device constants on its own; compare with the oracle's."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("devconst", os.path.join(ROOT, "keaki_amd", "csrc", "gen_constants.py"))
    g = importlib.util.module_from_spec(spec); spec.loader.exec_module(g)
    assert (g.P, g.R, g.Z) == (py.P, py.R, py.Z)
    assert g.G2_GEN == py.G2_GEN
    assert g.f2_mul(g.f2_inv(g.XI), (3, 0)) == py.B2
    assert g.f2_pow(g.XI, (g.P - 1) // 3) == py.TWIST_MUL_BY_Q_X and g.f2_pow(g.XI, (g.P - 1) // 2) == py.TWIST_MUL_BY_Q_Y
    for k in range(4):
        for i in range(6):
            assert g.f2_pow(g.XI, i * (g.P**k - 1) // 6) == py._FROB_W[k][i]
    naf = g.naf_6z2()
    assert sum(d << i for i, d in enumerate(naf)) == 6 * py.Z + 2
    assert all(not (naf[i] and naf[i + 1]) for i in range(len(naf) - 1))
    # the generated header on disk is up to date
    import io, contextlib
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        g.main()
    assert buf.getvalue().strip() == open(os.path.join(ROOT, "keaki_amd", "csrc", "bn254_constants.cuh")).read().strip()


def test_bench_input_generation():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
    from conftest import SplitMix64
    s = SplitMix64(b.SEED)
    ref = [s.next() for _ in range(16)]
    assert b.splitmix64_stream(b.SEED, 16).tolist() == ref
    x = b.random_fr_limbs(5000, 7)
    assert x.shape == (5000, 4) and x.dtype == np.uint64
    vals = [int.from_bytes(x[i].tobytes(), "little") for i in range(5000)]
    assert max(vals) < b.R_MOD and len(set(vals)) == 5000
    assert np.array_equal(x, b.random_fr_limbs(5000, 7))
