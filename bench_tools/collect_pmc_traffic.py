#!/usr/bin/env python3
"""Folds two rocprofv3 counter passes (FETCH_SIZE, WRITE_SIZE; separate runs, CSV output) of `bench.py --log2n 24 --steps 1
--warmup 0 --no-cpu-baseline` into profiles/<name>.json: bytes per launch for every kernel (the LAST launch of each kernel
= the timed 2^24 step). Units: rocprofv3 reports both counters in KB -> x 1024. No gfx950 x2 correction is applied to
FETCH_SIZE for the bucket kernel: MI355X_MICROARCH.md calibrates that factor only for 16-B-per-lane coalesced streams and
says other shapes are uncalibrated; the kernel gathers 64-B rows at per-lane addresses.

    python bench_tools/collect_pmc_traffic.py gpurun_out/pmc_f gpurun_out/pmc_w profiles/r03_msm_2p24_hbm_traffic_pmc.json

The output is stamped with the SHA-256 of the kernel sources (bench_tools/srchash.py): bench.py refuses it when the tree differs.
"""
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
from bench_tools.srchash import built_hash  # noqa: E402


def last_per_kernel(d, counter):
    out = {}
    for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(path)):
            if row.get("Counter_Name") != counter:
                continue
            k = row["Kernel_Name"][:70]
            did = int(row["Dispatch_Id"])
            if k not in out or did >= out[k][0]:
                out[k] = (did, float(row["Counter_Value"]) * 1024.0)
    return {k: v[1] for k, v in out.items()}


def _stamp():
    """hash embedded in the libkeaki_hip.so that was measured (raises if that build and the tree disagree)"""
    import ctypes
    return built_hash(ctypes.CDLL(os.path.join(ROOT, "keaki_amd", "libkeaki_hip.so")), "msm")


def main():
    fdir, wdir, dst = sys.argv[1:4]
    f, w = last_per_kernel(fdir, "FETCH_SIZE"), last_per_kernel(wdir, "WRITE_SIZE")
    kernels = {k: {"fetch_bytes": f.get(k), "write_bytes": w.get(k)} for k in sorted(set(f) | set(w))}
    json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes; counters in KB x 1024), python3 bench.py --log2n 24 "
                         "--steps 1 --warmup 0 --no-cpu-baseline --no-extras --kem-log2n 0, MI355X; last launch of each kernel",
               "unit": "bytes per launch", "log2n": 24, "precompute": True, "kernel_source_sha256": _stamp(),
               "note": "FETCH_SIZE as measured (no x2: the gfx950 correction is calibrated for 16-B-per-lane coalesced streams only; "
                       "k_msm_accumulate_g1_u29 gathers 64-B table rows at per-lane addresses). Expected reads of that kernel: 12 windows "
                       "x 2^24 x 64 B = 12.9 GB of table rows + 0.8 GB of sorted indices; writes = 2^21 buckets x 128 B.",
               "kernels": kernels}, open(dst, "w"), indent=1)
    for k, v in kernels.items():
        if "part_" in k or "_sort" in k or "accumulate" in k:
            print(k[:60], v)


if __name__ == "__main__":
    main()
