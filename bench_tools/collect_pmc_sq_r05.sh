#!/bin/bash
# SQ instruction counters behind bench.py's `kem.alu` and `fk.alu` (round 5): VALU wave-instructions per wave of the pairing kernels
# (2^14 pairings through bench_tools/profile_pairing_split.py) and of ONE call of open_fk at d = 2^21 (a run with two calls minus a run with
# one), counter passes only (--pmc with --kernel-trace, nothing else).
# Usage (repo root): bench_tools/collect_pmc_sq_r05.sh <tag>  ->  gpurun_out/<tag>/r05_pairing_pmc_sq_insts.json, r05_fk_sq_insts.json
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/$1; mkdir -p $O
cd /tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES --kernel-trace --output-format csv -d $O/pair_SQ -o p -- python3 $R/bench_tools/profile_pairing_split.py > $O/pair_SQ.log 2>&1; echo "pair rc=$?" >> $O/rc.txt
if [ "$2" != "nofk" ]; then
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES --kernel-trace --output-format csv -d $O/fk2_SQ -o p -- python3 $R/bench_tools/fk_calls.py 21 2 > $O/fk2_SQ.log 2>&1; echo "fk2 rc=$?" >> $O/rc.txt
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES --kernel-trace --output-format csv -d $O/fk1_SQ -o p -- python3 $R/bench_tools/fk_calls.py 21 1 > $O/fk1_SQ.log 2>&1; echo "fk1 rc=$?" >> $O/rc.txt
fi
cd $R
python3 - $O <<'PY'
import csv, glob, json, os, sys, ctypes
O = sys.argv[1]
sys.path.insert(0, os.getcwd())
from bench_tools.srchash import library_hashes
lib = ctypes.CDLL(os.path.join(os.getcwd(), "keaki_amd", "libkeaki_hip.so")); lib.keaki_hip_version.restype = ctypes.c_char_p
ver = lib.keaki_hip_version().decode()
def sums(d):
    out = {}
    for path in glob.glob("%s/%s/**/*counter_collection.csv" % (O, d), recursive=True):
        for r in csv.DictReader(open(path)):
            k = r["Kernel_Name"].split("(")[0]
            e = out.setdefault(k, {"SQ_INSTS_VALU": 0.0, "SQ_WAVES": 0.0, "launches": 0})
            e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            if r["Counter_Name"] == "SQ_WAVES":
                e["launches"] += 1
    return out
# per DISPATCH: the script launches the fused pairing, the Miller loop alone and the final exponentiation alone through the same kernel symbol
disp = {}
for path in glob.glob("%s/pair_SQ/**/*counter_collection.csv" % O, recursive=True):
    for r in csv.DictReader(open(path)):
        if r["Kernel_Name"].split("(")[0].endswith("::k_pairing"):
            disp.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
per = sorted((v["SQ_INSTS_VALU"] / v["SQ_WAVES"], v["SQ_WAVES"]) for v in disp.values() if v.get("SQ_WAVES"))
res = {"source": "rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES --kernel-trace of bench_tools/profile_pairing_split.py (2^14 pairings; three rounds of: fused pairing, Miller loop "
                 "alone, final exponentiation alone -- all dispatches of the symbol k_pairing, told apart by their instruction count); MI355X",
       "library": ver, "hashes": library_hashes(ver), "pairings_per_launch": 1 << 14, "dispatches_valu_per_wave": [round(x) for x, _ in per]}
if len(per) >= 3:
    third = len(per) // 3
    res["k_final_exp_only"] = {"valu_per_wave": sum(x for x, _ in per[:third]) / third}
    res["k_miller_only"] = {"valu_per_wave": sum(x for x, _ in per[third:2 * third]) / third}
    res["k_pairing"] = {"launches": third, "waves_per_launch": per[-1][1], "valu_per_wave": sum(x for x, _ in per[2 * third:]) / (len(per) - 2 * third)}
    print("k_pairing (fused):", res["k_pairing"], "Miller alone:", res["k_miller_only"], "final exponentiation alone:", res["k_final_exp_only"])
json.dump(res, open(O + "/r05_pairing_pmc_sq_insts.json", "w"), indent=1)
if glob.glob(O + "/fk2_SQ"):
    f2, f1 = sums("fk2_SQ"), sums("fk1_SQ")
    one = {}
    for k in f2:
        if "fft_stage" in k or "pointwise" in k or "mul_jac" in k or "fk_" in k:
            a, b = f2[k], f1.get(k, {"SQ_INSTS_VALU": 0.0, "SQ_WAVES": 0.0, "launches": 0})
            one[k[-64:]] = {"launches": a["launches"] - b["launches"], "waves": a["SQ_WAVES"] - b["SQ_WAVES"], "valu_wave_instructions": a["SQ_INSTS_VALU"] - b["SQ_INSTS_VALU"]}
    stage = sum(v["valu_wave_instructions"] for k, v in one.items() if "k_g1_fft_stage" in k)
    allk = sum(v["valu_wave_instructions"] for v in one.values())
    out = {"source": "rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES --kernel-trace of bench_tools/fk_calls.py 21 2 minus ... 21 1 (setup + N calls of keaki_hip_open_fk_poly): ONE call at d = 2^21; MI355X",
           "library": ver, "hashes": library_hashes(ver), "log2d": 21, "kernels_one_call": one, "stage_kernels_valu_wave_instructions_one_call": stage,
           "all_fk_kernels_valu_wave_instructions_one_call": allk}
    json.dump(out, open(O + "/r05_fk_sq_insts.json", "w"), indent=1)
    print("FK23 one call: stage kernels %.4g VALU wave-instructions, all FK kernels %.4g" % (stage, allk))
PY
find $O -name '*.csv' -size +1M -delete; find $O -name '*.db' -delete
cat $O/rc.txt
