#!/usr/bin/env python3
"""Folds two rocprofv3 counter passes (FETCH_SIZE, WRITE_SIZE; separate runs, CSV output) of `bench.py --log2n 24 --steps 1
--warmup 0 --no-cpu-baseline` into profiles/<name>.json: bytes per launch for every kernel (the LAST launch of each kernel
= the timed 2^24 step). Units: rocprofv3 reports both counters in KB -> x 1024.

FETCH_SIZE x 2 (round 6): the counter is TCC_EA0_RDREQ x 64 B, and EVERY fabric read request moves a whole 128-byte L2 line --
calibrated on this tree's own access shapes (bench_tools/ubench_fetch_calib.hip -> profiles/r06_fetch_size_calibration.txt: a
workgroup that reads the even 64-byte rows of a table and then the odd ones makes ONE request per line and hits L2 in the second
phase). `fetch_bytes` below is therefore bytes MOVED over the fabric (Infinity-Cache hits included); `fetch_size_counter` keeps
the raw figure. Rounds 1-5 committed the raw figure. WRITE_SIZE is exact (guide).

    python bench_tools/collect_pmc_traffic.py gpurun_out/pmc_f gpurun_out/pmc_w profiles/r06_msm_2p24_hbm_traffic_pmc.json

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
    kernels = {k: {"fetch_bytes": (2.0 * f[k]) if k in f else None, "fetch_size_counter": f.get(k), "write_bytes": w.get(k)} for k in sorted(set(f) | set(w))}
    json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes; counters in KB x 1024), python3 bench.py --log2n 24 "
                         "--steps 1 --warmup 0 --no-cpu-baseline --no-extras --kem-log2n 0, MI355X; last launch of each kernel",
               "unit": "bytes per launch", "log2n": 24, "precompute": True, "kernel_source_sha256": _stamp(), "fetch_factor": 2.0,
               "note": "fetch_bytes = FETCH_SIZE x 2 = TCC_EA0_RDREQ x 128 B: every fabric read request moves a whole 128-byte L2 line, in every access "
                       "shape of this tree (profiles/r06_fetch_size_calibration.txt). The bucket kernel's rows are 64 B: each costs a line (1.0 request "
                       "per row, half of it unwanted); expected: 12 windows x 2^24 rows x 128 B = 25.8 GB of lines for 12.9 GB of rows + the index stream "
                       "(0.8 GB useful; 1.15 requests per row in all, profiles/r06_bucket_clock_diagnosis.txt); writes = 2^21 buckets x 128 B.",
               "kernels": kernels}, open(dst, "w"), indent=1)
    for k, v in kernels.items():
        if "part_" in k or "_sort" in k or "accumulate" in k:
            print(k[:60], v)


if __name__ == "__main__":
    main()
