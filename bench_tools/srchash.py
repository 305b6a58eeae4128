"""Stamp for measurements that are committed under profiles/ and read back by bench.py (PMC traffic, ISA instruction counts):
the SHA-256 of the sources that define the measured kernel. bench.py refuses a committed figure whose stamp differs from the tree it
runs in, so a stale number can never ride along with a changed kernel. (The GPU box has no .git, hence no commit id.)"""
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MSM_KERNEL_SOURCES = ["msm.cuh", "msm_host.cuh", "msm_g1.hip", "fq29.cuh", "fq29_core.cuh", "fq29_asm.cuh", "xyzz29.cuh", "jac29.cuh",
                      "bn254_field.cuh", "bn254_field_asm.cuh", "bn254_curve.cuh"]
PAIRING_KERNEL_SOURCES = ["pairing.cuh", "pairing.hip", "pair261.cuh", "pair261_constants.cuh", "fq29.cuh", "fq29_core.cuh", "fq29_asm.cuh",
                          "fq29_dot_asm.cuh", "bn254_field.cuh", "bn254_field_asm.cuh", "bn254_curve.cuh"]


def source_hash(files=MSM_KERNEL_SOURCES) -> str:
    h = hashlib.sha256()
    for f in files:
        path = os.path.join(ROOT, "keaki_amd", "csrc", f)
        if os.path.exists(path):
            h.update(f.encode())
            h.update(open(path, "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print("msm", source_hash(MSM_KERNEL_SOURCES))
    print("pairing", source_hash(PAIRING_KERNEL_SOURCES))
