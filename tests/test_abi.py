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


def test_rccl_companion_library_exports_its_header():
    """include/keaki_hip_rccl.h (the optional RCCL collectives of the one-process-per-GPU form) against libkeaki_hip_rccl.so, its ctypes
    binding and its Rust declarations; the MAIN library must not depend on RCCL."""
    import subprocess
    from keaki_amd import rccl
    lib = rccl.load_rccl_library()
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "keaki_hip_rccl.h")).read(), flags=re.S)
    syms = sorted(set(re.findall(r"\b(keaki_hip_rccl_[a-z0-9_]+)\s*\(", hdr)))
    assert syms == sorted(rccl.EXPORTS) and len(syms) == 8
    for s in syms:
        assert hasattr(lib, s), s
    rs = open(os.path.join(ROOT, "rust", "keaki-hip-sys", "src", "rccl.rs")).read()
    assert sorted(set(re.findall(r"pub fn (keaki_hip_rccl_\w+)", rs))) == syms
    needed = subprocess.run(["readelf", "-d", os.path.join(ROOT, "keaki_amd", "libkeaki_hip.so")], stdout=subprocess.PIPE, text=True).stdout
    assert "librccl" not in needed, "libkeaki_hip.so must not link RCCL"
    needed = subprocess.run(["readelf", "-d", os.path.join(ROOT, "keaki_amd", "libkeaki_hip_rccl.so")], stdout=subprocess.PIPE, text=True).stdout
    assert "librccl" in needed and "libkeaki_hip.so" in needed


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
            if f.endswith((".py", ".hip", ".hip.h", ".h", ".hpp", ".cpp", "Makefile")):
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
    assert buf.getvalue().strip() == open(os.path.join(ROOT, "keaki_amd", "csrc", "bn254_constants.hip.h")).read().strip()


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


def test_host_mirror_library_loads():
    """libkeaki_host.so (C++ mirror of keaki's kzg/kem/enc/vec API) links against the C ABI and exports its front-end."""
    import ctypes
    from keaki_amd import hip
    hip.load_library()
    lib = ctypes.CDLL(os.path.join(ROOT, "keaki_amd", "libkeaki_host.so"))
    for s in ["keaki_host_setup", "keaki_host_commit", "keaki_host_open", "keaki_host_verify", "keaki_host_open_fk", "keaki_host_encapsulate",
              "keaki_host_decapsulate", "keaki_host_encrypt", "keaki_host_decrypt", "keaki_host_vec_commit", "keaki_host_vec_encrypt",
              "keaki_host_vec_decrypt", "keaki_host_setup_from_file", "keaki_host_ptau_parse", "keaki_host_ptau_section_info"]:
        assert hasattr(lib, s), s


def test_host_scalar_field_and_domain(py):
    """Host-side Fr arithmetic / Radix-2 domain of the mirror (no GPU needed) against the big-int oracle."""
    from keaki_amd import keaki as K
    to_int = lambda a: (int.from_bytes(np.asarray(a, np.uint64).tobytes(), "little") * py.FR_RINV) % py.R
    assert to_int(K.fr(-24)) == (-24) % py.R and to_int(K.fr(7)) == 7 and to_int(K.fr(0)) == 0
    a, b = K.fr(123456789), K.fr(-987654321)
    assert to_int(K.fr_mul(a, b)) == (123456789 * -987654321) % py.R
    assert to_int(K.fr_add(a, b)) == (123456789 - 987654321) % py.R
    assert to_int(K.fr_sub(a, b)) == (123456789 + 987654321) % py.R
    coeffs = np.stack([K.fr(c) for c in [-24, -25, -5, 9, 7]])
    assert to_int(K.poly_evaluate(coeffs, K.fr(3))) == py.poly_eval([c % py.R for c in [-24, -25, -5, 9, 7]], 3)
    # Radix2EvaluationDomain: ark-bn254's two-adic root 5^((r-1)/2^28), size = next power of two
    el = [to_int(e) for e in K.domain_elements(9)]
    w = pow(5, (py.R - 1) >> 28, py.R)
    g = pow(w, 1 << (28 - 4), py.R)
    assert len(el) == 16 and el == [pow(g, i, py.R) for i in range(16)]
    # ifft then fft is the identity; ifft gives the interpolating polynomial
    ev = np.stack([K.fr(i * i + 1) for i in range(9)])
    co = K.ifft(ev, 9)
    back = K.fft(co, 9)
    assert [to_int(x) for x in back[:9]] == [i * i + 1 for i in range(9)] and all(to_int(x) == 0 for x in back[9:])
    ci = [to_int(c) for c in co]
    assert all(py.poly_eval(ci, el[i]) == (i * i + 1 if i < 9 else 0) for i in range(16))
    # Fr::rand semantics: SplitMix64 stream -> limbs with the top two bits cleared, < r
    r = K.Rng(5).fr_rand()
    assert int.from_bytes(r.tobytes(), "little") < py.R


def test_committed_isa_counts_belong_to_this_tree():
    """profiles/r06_accumulate_isa.json and r05_pairing_isa.json (the pairing kernels did not change in round 6) (bench.py's `alu` / `kem.alu` diagnostics read them) must have been made from
    the kernel sources of this tree: re-run `python bench_tools/count_isa.py` (`--pairing`) after touching the MSM (pairing) kernels."""
    import json
    from bench_tools.srchash import source_hash, MSM_KERNEL_SOURCES
    from bench_tools.srchash import PAIRING_KERNEL_SOURCES
    j = json.load(open(os.path.join(ROOT, "profiles", "r06_accumulate_isa.json")))
    assert j["kernel_source_sha256"] == source_hash(MSM_KERNEL_SOURCES), "stale: run python bench_tools/count_isa.py"
    pj = json.load(open(os.path.join(ROOT, "profiles", "r05_pairing_isa.json")))
    assert pj["kernel_source_sha256"] == source_hash(PAIRING_KERNEL_SOURCES), "stale: run python bench_tools/count_isa.py --pairing"
    assert 0.4 < pj["mad_fraction_static"] < 0.8
    assert j["registers"].get("private_seg_size", 0) == 0, "the bucket kernel must not spill"
    assert 1500 < j["loop_instructions"] < 4000 and j["loop_v_mad_u64_u32"] > 1000
