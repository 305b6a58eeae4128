"""The .ptau parser (keaki_amd/host/ptau.cpp, mirror of the reference's src/kzg/ptau.rs:230-358; scope row f-3: SRS ingest) under
AddressSanitizer + UBSan on truncated and corrupted copies of the reference's fixture: whatever the bytes, the parser answers with a value
or an error and never reads outside the file. CPU only (the sanitizer build holds the parser alone: no GPU library behind it)."""
import os
import random
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "keaki_amd", "host")
PTAU = os.path.join(ROOT, "tests", "golden", "ppot_0080_01.ptau.test")


@pytest.fixture(scope="module")
def harness():
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    r = subprocess.run(["make", "-C", HOST, "ptau_fuzz_asan"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    return os.path.join(HOST, "ptau_fuzz_asan")


def run(harness, paths):
    env = dict(os.environ, ASAN_OPTIONS="exitcode=99:detect_leaks=1", UBSAN_OPTIONS="halt_on_error=1:exitcode=98")
    r = subprocess.run([harness] + paths, capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, "sanitizer report or crash (rc %d):\n%s" % (r.returncode, r.stderr[-3000:])
    lines = r.stdout.strip().splitlines()
    assert len(lines) == len(paths)
    return lines


def test_fixture_parses_under_the_sanitizers(harness):
    assert run(harness, [PTAU]) == ["ok 3 2 1"]


def test_truncated_and_corrupted_files_never_read_out_of_bounds(harness, tmp_path):
    data = open(PTAU, "rb").read()
    rnd = random.Random(20260)
    variants = []
    # every length around the structural boundaries, and random ones
    cuts = sorted(set(list(range(0, 80)) + [len(data) - k for k in range(1, 40)] + [rnd.randrange(len(data)) for _ in range(60)]))
    for c in cuts:
        variants.append(data[:c])
    # byte flips: dense in the header and the section table (first 256 bytes hold magic, version, section count, the first headers),
    # in every section header (12 bytes in front of each section), sparse elsewhere
    for _ in range(250):
        b = bytearray(data)
        for _ in range(rnd.randrange(1, 4)):
            pos = rnd.randrange(256) if rnd.random() < 0.7 else rnd.randrange(len(data))
            b[pos] = rnd.randrange(256)
        variants.append(bytes(b))
    # section sizes and counts blown up: 32- and 64-bit little-endian fields overwritten with extreme values
    for off in range(8, 140, 4):
        for v in (0, 1, 0x7FFFFFFF, 0xFFFFFFFF):
            b = bytearray(data); b[off:off + 4] = v.to_bytes(4, "little"); variants.append(bytes(b))
    paths = []
    for i, v in enumerate(variants):
        p = tmp_path / ("v%04d.ptau" % i)
        p.write_bytes(v)
        paths.append(str(p))
    out = []
    for i in range(0, len(paths), 100):
        out += run(harness, paths[i:i + 100])
    assert all(l.startswith("ok ") or l.startswith("err ") for l in out)
    n_err = sum(l.startswith("err ") for l in out)
    assert n_err >= 200 and n_err < len(out)       # every truncation and most header corruptions are errors; flips inside unused payload still parse


def test_host_mirror_cpu_arithmetic_under_the_sanitizers():
    """keaki_amd/host/host_cpu_asan_main.cpp: the mirror's scalar field and radix-2 domain (what keaki takes from ark-ff / ark-poly on the
    host) -- algebraic identities and fft / ifft round trips for sizes 1 .. 4096, with AddressSanitizer and UBSan watching the limb
    arithmetic. Links libkeaki_hip.so but calls nothing in it."""
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    if not os.path.exists(os.path.join(ROOT, "keaki_amd", "libkeaki_hip.so")):
        pytest.skip("libkeaki_hip.so not built")
    r = subprocess.run(["make", "-C", HOST, "host_cpu_asan"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    env = dict(os.environ, ASAN_OPTIONS="exitcode=99:detect_leaks=1", UBSAN_OPTIONS="halt_on_error=1:exitcode=98")
    r = subprocess.run([os.path.join(HOST, "host_cpu_asan")], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "all checks passed" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
