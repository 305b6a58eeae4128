"""Stamp for measurements that are committed under profiles/ and read back by bench.py (PMC traffic, ISA instruction counts):
the SHA-256 of the sources that define the measured kernel. bench.py refuses a committed figure whose stamp differs from the tree it
runs in, so a stale number can never ride along with a changed kernel. (The GPU box has no .git, hence no commit id.)"""
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MSM_KERNEL_SOURCES = ["msm.hip.h", "msm_host.hip.h", "msm_g1.hip", "fq29.hip.h", "fq29_core.hip.h", "fq29_asm.hip.h", "xyzz29.hip.h", "jac29.hip.h",
                      "bn254_field.hip.h", "bn254_field_asm.hip.h", "bn254_curve.hip.h"]
FK_KERNEL_SOURCES = ["fft_g1.hip", "jac29.hip.h", "fq29.hip.h", "fq29_core.hip.h", "fq29_asm.hip.h", "bn254_field.hip.h", "bn254_field_asm.hip.h", "bn254_curve.hip.h"]
PAIRING_KERNEL_SOURCES = ["pairing.hip.h", "pairing_wide.hip.h", "pairing.hip", "pair261.hip.h", "pair261_constants.hip.h", "fq29.hip.h", "fq29_core.hip.h", "fq29_asm.hip.h",
                          "fq29_dot_asm.hip.h", "bn254_field.hip.h", "bn254_field_asm.hip.h", "bn254_curve.hip.h"]


def source_hash(files=MSM_KERNEL_SOURCES) -> str:
    """Hash of the TREE's sources. A listed file that is missing is an error (a renamed source must be renamed here too)."""
    h = hashlib.sha256()
    for f in files:
        path = os.path.join(ROOT, "keaki_amd", "csrc", f)
        if not os.path.exists(path):
            raise FileNotFoundError("srchash: listed kernel source %s does not exist" % path)
        h.update(f.encode())
        h.update(open(path, "rb").read())
    return h.hexdigest()[:16]


def define_string() -> str:
    """What csrc/Makefile embeds in keaki_hip_version(): 'msm:<hash>,pairing:<hash>'."""
    return "msm:%s,pairing:%s,fk:%s" % (source_hash(MSM_KERNEL_SOURCES), source_hash(PAIRING_KERNEL_SOURCES), source_hash(FK_KERNEL_SOURCES))


def library_hashes(version: str) -> dict:
    """{'msm': ..., 'pairing': ...} parsed from keaki_hip_version() of the LOADED library: the sources that binary was built from."""
    tail = version.split("src=", 1)[1] if "src=" in version else ""
    out = {}
    for part in tail.split(","):
        if ":" in part:
            k, v = part.split(":", 1)
            out[k.strip()] = v.strip()
    return out


def built_hash(hip_lib, which="msm") -> str:
    """Hash embedded in the loaded libkeaki_hip.so; raises when the library was built without one or from other sources than the tree's
    (a stamp must describe the binary that was measured AND the tree that is committed)."""
    import ctypes
    hip_lib.keaki_hip_version.restype = ctypes.c_char_p
    got = library_hashes(hip_lib.keaki_hip_version().decode()).get(which)
    want = source_hash({"msm": MSM_KERNEL_SOURCES, "pairing": PAIRING_KERNEL_SOURCES, "fk": FK_KERNEL_SOURCES}[which])
    if got != want:
        raise RuntimeError("libkeaki_hip.so was built from other %s kernel sources (%s) than this tree holds (%s): rebuild" % (which, got, want))
    return got


if __name__ == "__main__":
    import sys
    if "--define" in sys.argv:
        print(define_string())
    else:
        print("msm", source_hash(MSM_KERNEL_SOURCES))
        print("pairing", source_hash(PAIRING_KERNEL_SOURCES))
        print("fk", source_hash(FK_KERNEL_SOURCES))
