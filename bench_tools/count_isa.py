#!/usr/bin/env python3
"""Instruction counts of the bucket kernel's loop body from the ISA hipcc emits for THIS tree (runs on the CPU box: hipcc cross-compiles).

    python bench_tools/count_isa.py            # writes profiles/r05_accumulate_isa.json
    python bench_tools/count_isa.py --pairing  # writes profiles/r05_pairing_isa.json (static VALU mix of k_pairing and its out-of-line products)

Compiles keaki_amd/csrc/msm_g1.hip to gfx950 assembly (the flags of the Makefile), cuts k_msm_accumulate_g1_u29 into basic blocks and
reports, per block, the number of instructions and of v_mad_u64_u32. The loop body of one mixed addition = the two consecutive blocks
with the largest combined v_mad_u64_u32 count among blocks of < 2000 instructions (the common path: products U2, S2 and the zero
filter; then PP, PPP, Q, R^2, the dual product, ZZ, ZZZ). The block of > 3000 instructions is the exact-zero / doubling path, entered
18 times in 2^29 additions; the first-point block runs once per bucket. The whole table is kept in the JSON so the choice can be audited.
bench.py reads `loop_instructions` for its `alu` diagnostic and refuses the file when the kernel sources changed (kernel_source_sha256)."""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench_tools.srchash import source_hash, MSM_KERNEL_SOURCES, PAIRING_KERNEL_SOURCES  # noqa: E402

KERNEL = "k_msm_accumulate_g1_u29"


def main():
    src = os.path.join(ROOT, "keaki_amd", "csrc", "msm_g1.hip")
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, "msm_g1.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-S", src, "-o", asm],
                              cwd=os.path.dirname(src), stderr=subprocess.DEVNULL)
        lines = open(asm).read().split("\n")
    # the one-pass form <NT = 0, MODE = ACC_WHOLE> (the chunked host-pointer calls run <0, 1..3>: the same loop body, other prologue / epilogue)
    start = [i for i, l in enumerate(lines) if re.match(r"^_ZN5bn254L\d+%sILi0ELi0E\w*:" % KERNEL, l)][0]
    end = [i for i, l in enumerate(lines) if i > start and l.startswith(".Lfunc_end")][0]
    blocks, cur = [], ["entry", []]
    for l in lines[start + 1:end]:
        t = l.strip()
        m = re.match(r"^(\.LBB\d+_\d+):", t)
        if m:
            blocks.append(cur); cur = [m.group(1), []]
            continue
        if not t or t[0] in ";./":
            continue
        cur[1].append(t.split(";")[0].strip())
    blocks.append(cur)
    table = [{"block": lab, "instructions": len(ins), "v_mad_u64_u32": sum(x.startswith("v_mad_u64_u32") for x in ins),
              "v_mul_lo_u32": sum(x.startswith("v_mul_lo_u32") for x in ins), "global_load": sum(x.startswith("global_load") for x in ins)}
             for lab, ins in blocks]
    best, pair = -1, None
    for a, b in zip(table, table[1:]):
        if a["instructions"] < 2000 and b["instructions"] < 2000 and a["v_mad_u64_u32"] + b["v_mad_u64_u32"] > best:
            best, pair = a["v_mad_u64_u32"] + b["v_mad_u64_u32"], (a, b)
    meta = {}
    for l in lines:
        m = re.match(r"\s*\.set\s+_ZN5bn254L\d+%sILi0ELi0E\w*\.(num_vgpr|num_agpr|numbered_sgpr|private_seg_size),\s*(\d+)" % KERNEL, l)
        if m:
            meta[m.group(1)] = int(m.group(2))
    out = {"kernel": KERNEL, "kernel_source_sha256": source_hash(MSM_KERNEL_SOURCES), "compiler": "hipcc -O3 -std=c++17 --offload-arch=gfx950 (ROCm 7.2)",
           "loop_blocks": [pair[0]["block"], pair[1]["block"]], "loop_instructions": pair[0]["instructions"] + pair[1]["instructions"],
           "loop_v_mad_u64_u32": best, "registers": meta, "blocks": table,
           "rule": "two consecutive basic blocks with the largest combined v_mad_u64_u32 count among blocks of < 2000 instructions"}
    dst = os.path.join(ROOT, "profiles", "r05_accumulate_isa.json")
    json.dump(out, open(dst, "w"), indent=1)
    print("loop: %s = %d instructions, %d v_mad_u64_u32; registers %s -> %s" % (out["loop_blocks"], out["loop_instructions"], best, meta, dst))


def pairing():
    """Static instruction mix of k_pairing (the lane-pair throughput kernel, csrc/pairing.hip.h) and of the products it calls out of line
    (p261::fq2d_mul / fq2d_sqr): VALU instructions and how many of them are v_mad_u64_u32. bench.py's kem.alu prices the measured
    SQ_INSTS_VALU per wave with the two issue rates of profiles/r01_ubench_u29_gfx950.txt weighted by this fraction."""
    src = os.path.join(ROOT, "keaki_amd", "csrc", "pairing.hip")
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, "pairing.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-S", src, "-o", asm],
                              cwd=os.path.dirname(src), stderr=subprocess.DEVNULL)
        lines = open(asm).read().split("\n")
    out = {"kernel_source_sha256": source_hash(PAIRING_KERNEL_SOURCES), "compiler": "hipcc -O3 -std=c++17 --offload-arch=gfx950 (ROCm 7.2)", "functions": {}}
    for i, l in enumerate(lines):
        m = re.match(r"^(_ZN5bn254\w*(k_pairingENS|fq2d_mul|fq2d_sqr)\w*):", l)
        if not m:
            continue
        end = [j for j in range(i, len(lines)) if lines[j].startswith(".Lfunc_end")][0]
        ins = [x.strip().split(";")[0].strip() for x in lines[i + 1:end] if x.strip() and x.strip()[0] not in ";./" and not x.strip().startswith(".LBB")]
        valu = [x for x in ins if x.startswith("v_")]
        out["functions"][m.group(2).replace("ENS", "")] = {"symbol": m.group(1), "instructions": len(ins), "valu": len(valu),
                                                           "v_mad_u64_u32": sum(x.startswith("v_mad_u64_u32") for x in valu)}
    k = out["functions"]["k_pairing"]
    out["mad_fraction_static"] = k["v_mad_u64_u32"] / k["valu"]
    out["note"] = ("static counts of the emitted ISA (loops counted once): the kernel body and the out-of-line Fq2 products have the same mix within "
                   "a few percent, which is what makes the static fraction usable as the dynamic one")
    dst = os.path.join(ROOT, "profiles", "r05_pairing_isa.json")
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps(out["functions"]), "mad fraction %.3f -> %s" % (out["mad_fraction_static"], dst))


if __name__ == "__main__":
    pairing() if "--pairing" in sys.argv else main()
