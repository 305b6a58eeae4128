"""The reference's nine .ptau parser tests (src/kzg/ptau.rs:392-514) restated against the host mirror's parser
(keaki_amd/host/ptau.cpp), on the reference's own fixture (tests/golden/ppot_0080_01.ptau.test = ptau/ppot_0080_01.ptau.test),
plus what the reference does not check: the parsed limbs are the Montgomery residues of on-curve points (vs the oracle's parser)
and malformed files give errors instead of panics. CPU only; the GPU half (new_from_file + device curve check) is in
test_gpu_keaki_api.py."""
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
PTAU = os.path.join(GOLDEN, "ppot_0080_01.ptau.test")
TEST_FILE_LEN, TEST_CEREMONY_POWER, TEST_FILE_POWER = 95_634, 28, 1
N_SECTIONS, METADATA_LEN, SECTION_HEADER_LEN = 11, 12, 12


@pytest.fixture(scope="module")
def K():
    from keaki_amd import keaki
    return keaki


def test_section_id(K):
    """src/kzg/ptau.rs:392-398"""
    assert K.ptau_section_index(1) == 0 and K.ptau_section_index(2) == 1 and K.ptau_section_index(3) == 2
    assert K.ptau_section_index(15) == 10 and K.ptau_section_index(12) == 7
    assert K.ptau_section_index(99) == -1 and K.ptau_section_index(8) == -1


def test_file_loader_and_metadata(K):
    """:400-415 -- length 95,634, magic 'ptau', 11 sections"""
    d = K.ptau_parse(PTAU)
    assert d["file_len"] == TEST_FILE_LEN


def test_section_info(K):
    """:417-425"""
    assert K.ptau_section_info(bytes([1, 0, 0, 0, 44, 0, 0, 0, 0, 0, 0, 0]), 0) == (1, 44, SECTION_HEADER_LEN)
    with pytest.raises(K.SetupFileError) as e:
        K.ptau_section_info(bytes([99] + [0] * 11), 0)
    assert e.value.kind == "UnknownSection" and e.value.a == 99


def test_file_sections(K):
    """:427-459 -- slot i holds the section whose index is i; sizes positive; chained positions"""
    d = K.ptau_parse(PTAU)
    secs = d["sections"]
    assert len(secs) == N_SECTIONS
    for i, (sid, size, pos) in enumerate(secs):
        assert K.ptau_section_index(sid) == i
        assert size > 0 and pos + size <= d["file_len"]
    assert secs[0][2] == METADATA_LEN + SECTION_HEADER_LEN
    for (sid, size, pos), nxt in zip(secs, secs[1:]):
        assert pos + size + SECTION_HEADER_LEN == nxt[2]


def test_header_section(K, py):
    """:461-475 -- modulus == BN254 Fq modulus (LE bytes), power 1, ceremony power 28"""
    d = K.ptau_parse(PTAU)
    assert d["field_modulus"] == py.P.to_bytes(32, "little")
    assert d["power"] == TEST_FILE_POWER and d["ceremony_power"] == TEST_CEREMONY_POWER


def test_tau_sections_and_powers_of_tau(K, py, oc):
    """:477-514 -- 2 * 2^power - 1 G1 powers, 2^power G2 powers; and (beyond the reference) their values"""
    d = K.ptau_parse(PTAU)
    assert d["tau_g1"].shape == (2 ** TEST_FILE_POWER * 2 - 1, 8) and d["tau_g2"].shape == (2 ** TEST_FILE_POWER, 16)
    pts = py.ptau_points(open(PTAU, "rb").read())
    assert oc.g1_to_ints(d["tau_g1"]) == pts["tau_g1"]
    assert oc.g2_to_ints(d["tau_g2"]) == pts["tau_g2"]
    assert pts["tau_g1"][0] == py.G1_GEN and pts["tau_g2"][0] == py.G2_GEN          # tau^0
    assert all(py.g1_is_on_curve(p) for p in pts["tau_g1"]) and all(py.g2_is_on_curve(p) for p in pts["tau_g2"])


def test_malformed_files_are_errors(K, tmp_path):
    blob = bytearray(open(PTAU, "rb").read())

    def parse(b, name):
        p = tmp_path / name
        p.write_bytes(bytes(b))
        return K.ptau_parse(str(p))

    with pytest.raises(K.SetupFileError) as e:
        K.ptau_parse(str(tmp_path / "missing.ptau"))
    assert e.value.kind == "FileError"
    bad = bytearray(blob); bad[0:4] = b"zkey"
    with pytest.raises(K.SetupFileError) as e:
        parse(bad, "magic.ptau")
    assert e.value.kind == "InvalidFileType"
    bad = bytearray(blob); bad[8] = 10
    with pytest.raises(K.SetupFileError) as e:
        parse(bad, "nsec.ptau")
    assert e.value == K.SetupFileError(4, 10, 0, "")
    bad = bytearray(blob); bad[METADATA_LEN] = 8                      # first section id -> unknown
    with pytest.raises(K.SetupFileError) as e:
        parse(bad, "sec.ptau")
    assert e.value.kind == "UnknownSection" and e.value.a == 8
    with pytest.raises(K.SetupFileError) as e:                          # cut inside the section table: the reference would panic on the slice
        parse(blob[:5000], "cut.ptau")
    assert e.value.kind == "Truncated"
    # header claims power 2: section 2 would need 7 points, holds 3 -> ElementSizeMismatch(expected-by-power, actual) as in :245-251
    d = K.ptau_parse(PTAU)
    hdr_pos = d["sections"][0][2]
    bad = bytearray(blob); bad[hdr_pos + 4 + 32] = 2
    with pytest.raises(K.SetupFileError) as e:
        parse(bad, "power.ptau")
    assert e.value.kind == "ElementSizeMismatch" and (e.value.a, e.value.b) == (64 * 7, 64 * 3)
