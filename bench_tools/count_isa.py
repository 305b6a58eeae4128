#!/usr/bin/env python3
"""Instruction counts of the bucket kernel's loop body from the ISA hipcc emits for THIS tree (runs on the CPU box: hipcc cross-compiles).

    python bench_tools/count_isa.py            # writes profiles/r06_accumulate_isa.json
    python bench_tools/count_isa.py --pairing  # writes profiles/r05_pairing_isa.json (static VALU mix of k_pairing and its out-of-line products)

Compiles keaki_amd/csrc/msm_g1.hip to gfx950 assembly (the flags of the Makefile), cuts k_msm_accumulate_g1_u29 into basic blocks and
reports, per block, the number of instructions and of v_mad_u64_u32. The loop body of one mixed addition = the blocks of the loop minus the
rare ones (round 5: until then the rule took two labelled blocks and with them the exact-zero test of P that hangs off the filter -- 241
instructions and one product stream that run 18 times in 2^29 additions). The whole table is kept in the JSON so the choice can be audited.
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
    # the one-pass form <NT = 0, MODE = ACC_WHOLE, PF = 2> (the chunked host-pointer calls run <0, 1..3>: the same loop body, other prologue / epilogue)
    start = [i for i, l in enumerate(lines) if re.match(r"^_ZN5bn254L\d+%sILi0ELi0ELi2E\w*:" % KERNEL, l)][0]
    end = [i for i, l in enumerate(lines) if i > start and l.startswith(".Lfunc_end")][0]
    # basic blocks: a label `.LBBn_m:` or the fall-through marker `; %bb.N:` starts one; the trailing comment says whether it lies in the loop
    blocks, cur = [], ["entry", [], False]
    for l in lines[start + 1:end]:
        t = l.strip()
        m = re.match(r"^(\.LBB\d+_\d+):", t) or re.match(r"^; %(bb\.\d+):", t)
        if m:
            blocks.append(cur); cur = [m.group(1), [], "Loop" in t]
            continue
        if not t or t[0] in ";./":
            continue
        cur[1].append(t.split(";")[0].strip())
    blocks.append(cur)
    table = [{"block": lab, "in_loop": inl, "instructions": len(ins), "valu": sum(x.startswith("v_") for x in ins),
              "v_mad_u64_u32": sum(x.startswith("v_mad_u64_u32") for x in ins), "branches": sum(x.startswith("s_cbranch") or x.startswith("s_branch") for x in ins),
              "global_load": sum(x.startswith("global_load") for x in ins)} for lab, ins, inl in blocks]
    # The COMMON path of one mixed addition = the blocks of the loop minus the rare ones, which are recognisable by their product streams:
    #   exactly one product (162 multiply-adds): an exact zero test (u29_is_zero: P after the filter hit, R after P was zero);
    #   exactly two products and a branch back (324): the first point of a bucket;   > 2,500 instructions: the doubling path.
    # What is left holds U2, S2 (324) and PP, PPP, Q, R^2, the dual product, ZZ, ZZZ (1,143): 1,467 multiply-adds = 9.06 product streams.
    def rare(b):
        return b["v_mad_u64_u32"] == 162 or (b["v_mad_u64_u32"] == 324 and b["instructions"] < 450) or b["instructions"] > 2500
    common = [b for b in table if b["in_loop"] and not rare(b) and b["block"] != "entry"]
    # seg_next's segment switch (a small inner loop entered at every segment boundary, ~1 in 16 iterations) is counted: it is short
    loop_instructions = sum(b["instructions"] for b in common)
    best = sum(b["v_mad_u64_u32"] for b in common)
    meta = {}
    for l in lines:
        m = re.match(r"\s*\.set\s+_ZN5bn254L\d+%sILi0ELi0ELi2E\w*\.(num_vgpr|num_agpr|numbered_sgpr|private_seg_size),\s*(\d+)" % KERNEL, l)
        if m:
            meta[m.group(1)] = int(m.group(2))
    out = {"kernel": KERNEL, "kernel_source_sha256": source_hash(MSM_KERNEL_SOURCES), "compiler": "hipcc -O3 -std=c++17 --offload-arch=gfx950 (ROCm 7.2)",
           "loop_blocks": [b["block"] for b in common], "loop_instructions": loop_instructions, "loop_valu": sum(b["valu"] for b in common),
           "loop_v_mad_u64_u32": best, "loop_branches": sum(b["branches"] for b in common), "registers": meta, "blocks": table,
           "rule": "blocks inside the loop minus the rare ones (one-product blocks = exact zero tests, the two-product first-point block, the doubling path)"}
    dst = os.path.join(ROOT, "profiles", "r06_accumulate_isa.json")
    json.dump(out, open(dst, "w"), indent=1)
    print("common path of one mixed addition: %d instructions (%d VALU, %d v_mad_u64_u32, %d branches) in %d blocks; registers %s -> %s"
          % (loop_instructions, out["loop_valu"], best, out["loop_branches"], len(common), meta, dst))


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
