#!/bin/bash
# Fabric traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of the FK23 stage kernels at d = 2^21 and of the pairing kernel at 2^14 pairings,
# and the SQ instruction counters of the FK23 kernels (per wave).
# Usage (repo root): bench_tools/collect_pmc_fk_pairing.sh <tag>   ->  gpurun_out/<tag>/r05_fk_pairing_hbm_traffic_pmc.json
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/$1; mkdir -p $O
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/fk_$c -o p -- python3 $R/bench_tools/fk_calls.py 21 2 > $O/fk_$c.log 2>&1
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/fk1_$c -o p -- python3 $R/bench_tools/fk_calls.py 21 1 > $O/fk1_$c.log 2>&1
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pair_$c -o p -- python3 $R/bench_tools/profile_pairing_split.py > $O/pair_$c.log 2>&1
done
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/fk_SQ -o p -- python3 $R/bench_tools/fk_calls.py 21 1 > $O/fk_SQ.log 2>&1
cd $R
python3 - $O <<'PY'
import csv, glob, json, os, sys, ctypes
O = sys.argv[1]
sys.path.insert(0, os.getcwd())
from bench_tools.srchash import library_hashes
lib = ctypes.CDLL(os.path.join(os.getcwd(), "keaki_amd", "libkeaki_hip.so")); lib.keaki_hip_version.restype = ctypes.c_char_p
def per_kernel(d, counter):
    out = {}
    for path in glob.glob("%s/%s/**/*counter_collection.csv" % (O, d), recursive=True):
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == counter:
                k = r["Kernel_Name"].split("(")[0][-48:]
                out.setdefault(k, []).append(float(r["Counter_Value"]) * 1024.0)
    return out
res = {"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes; KB x 1024; no gfx950 x2 correction: scattered 96-B / 64-B accesses, uncalibrated) of bench_tools/bench_fk.py 21 "
                 "-> bench_tools/fk_calls.py 21 2 (setup + two calls; `fk_one_call` = that run minus a run with one call) and bench_tools/profile_pairing_split.py (2^14 pairings, three launches per kernel); MI355X",
       "library": lib.keaki_hip_version().decode(), "hashes": library_hashes(lib.keaki_hip_version().decode()), "fk_d": 1 << 21, "pairings": 1 << 14, "kernels": {}}
for tag in ("fk", "pair"):
    f, w = per_kernel(tag + "_FETCH_SIZE", "FETCH_SIZE"), per_kernel(tag + "_WRITE_SIZE", "WRITE_SIZE")
    for k in sorted(set(f) | set(w)):
        if any(x in k for x in ("fft_stage", "pointwise", "mul_jac", "fk_finish", "k_pairing", "k_miller", "k_final")):
            res["kernels"][k] = {"launches": len(f.get(k, [])), "fetch_bytes_total": sum(f.get(k, [])), "write_bytes_total": sum(w.get(k, [])),
                                 "fetch_bytes_per_launch": sum(f.get(k, [])) / max(1, len(f.get(k, []))), "write_bytes_per_launch": sum(w.get(k, [])) / max(1, len(w.get(k, [])))}
# instruction counters of the FK23 kernels, per wave (sums over all launches / SQ_WAVES)
sq = {}
for path in glob.glob("%s/fk_SQ/**/*counter_collection.csv" % O, recursive=True):
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0][-48:]
        if any(x in k for x in ("fft_stage", "pointwise", "mul_jac")):
            sq.setdefault(k, {}).setdefault(r["Counter_Name"], 0.0)
            sq[k][r["Counter_Name"]] += float(r["Counter_Value"])
res["fk_instructions_per_wave"] = {k: {c.lower()[3:]: v[c] / v["SQ_WAVES"] for c in v if c != "SQ_WAVES"} for k, v in sq.items() if v.get("SQ_WAVES")}
for k, v in res["fk_instructions_per_wave"].items():
    print("%-50s per wave: %s" % (k, {c: int(x) for c, x in v.items()}))
# one call = (setup + 2 calls) - (setup + 1 call), summed over the FK23 kernels
def total(tag, counter):
    return sum(sum(v) for k, v in per_kernel(tag + "_" + counter, counter).items() if any(x in k for x in ("fft_stage", "pointwise", "mul_jac", "fk_finish", "k_fr_", "k_fk_")))
res["fk_one_call"] = {"fetch_bytes": total("fk", "FETCH_SIZE") - total("fk1", "FETCH_SIZE"), "write_bytes": total("fk", "WRITE_SIZE") - total("fk1", "WRITE_SIZE"), "log2d": 21}
print("one call of open_fk at d = 2^21: fetch %.1f GB, write %.1f GB" % (res["fk_one_call"]["fetch_bytes"] / 1e9, res["fk_one_call"]["write_bytes"] / 1e9))
json.dump(res, open(O + "/r05_fk_pairing_hbm_traffic_pmc.json", "w"), indent=1)
for k, v in res["kernels"].items():
    print("%-50s launches %3d  fetch %.3f GB  write %.3f GB (totals)" % (k, v["launches"], v["fetch_bytes_total"] / 1e9, v["write_bytes_total"] / 1e9))
PY
find $O -name '*.csv' -size +1M -delete; find $O -name '*.db' -delete
