"""What lowers the clock under the G1 bucket kernel? One box, one process, the 2^24 MSM with tables; for every configuration:
kernel time of k_msm_accumulate_g1_u29 (HIP events on the launch stream), the shader clock inside it (bench_tools/clock_probe.hip: s_memtime ticks over an
8 ms window of the constant 100 MHz clock) and the package power / sclk rocm-smi reports while the MSM runs back to back.

Uses the DIAGNOSTIC build of the library (make -C keaki_amd/csrc diag -> bench_tools/diag/libkeaki_hip.so, -DKEAKI_DIAG): option diag_row_mask ANDs the
table-row index of every entry of the bucket-ordered stream with a mask before the bucket kernel runs -- the kernel binary, its instruction stream and its
trip counts are the shipped ones, only the ADDRESSES of the gathers change (the result is wrong by construction: parity is off for these rows).
   mask 2^21 - 1 -> 128 MiB of table (fits the 256 MiB Infinity Cache: no HBM traffic, full L2 / fabric request traffic)
   mask 2^15 - 1 ->   2 MiB (fits one XCD's 4 MiB L2: no fabric traffic)
   mask 2^9  - 1 ->  32 KiB (fits the CU's L1)
`--product` runs the rows that need no mask on keaki_amd/libkeaki_hip.so itself; `--opt=name:value[,name:value]` appends a configuration.
"""
import os, sys, time, subprocess, threading, re, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import keaki_amd.hip as KH
DIAG = os.path.join(ROOT, "bench_tools", "diag", "libkeaki_hip.so")
PRODUCT = "--product" in sys.argv
if not PRODUCT:
    KH.lib_path = lambda: DIAG
from keaki_amd.hip import KeakiHip
from bench import random_fr_limbs, mont_words, SEED
dev = torch.device("cuda", 0)
hip = KeakiHip(0)
hip.lib.keaki_hip_version.restype = C.c_char_p
print("# library: %s" % hip.lib.keaki_hip_version().decode())
pr = C.CDLL(os.path.join(ROOT, "bench_tools", "libclock_probe.so"))
pr.probe_start.argtypes = [C.c_uint64]; pr.probe_wait.restype = C.c_double
log2n = 24
n = 1 << log2n
d_gen = torch.from_numpy(np.array(mont_words(1) + mont_words(2), np.uint64).view(np.int64)).to(dev)
d_k = torch.from_numpy(random_fr_limbs(n, SEED + 1).view(np.int64)).to(dev)
d_pts = torch.empty((n, 8), dtype=torch.int64, device=dev)
torch.cuda.synchronize()
hip.g1_mul_batch_dev(d_gen.data_ptr(), 0, d_k.data_ptr(), n, d_pts.data_ptr()); hip.synchronize()
srs = hip.srs_g1_wrap_dev(d_pts.data_ptr(), n)
hip.srs_g1_precompute(srs)
d_s = torch.from_numpy(random_fr_limbs(n, SEED + 2).view(np.int64)).to(dev)
d_out = torch.zeros(12, dtype=torch.int64, device=dev)
torch.cuda.synchronize()
DEFAULTS = {"acc_nt": 0, "acc_prefetch": 1, "acc_idxq": 1, "diag_row_mask": 0, "msm_c_shared": 0}


def msm(reps=1):
    for _ in range(reps):
        hip.msm_g1_dev(srs, d_s.data_ptr(), n, d_out.data_ptr())


def smi():
    try:
        out = subprocess.run(["/opt/rocm/bin/rocm-smi", "--showpower", "--showclocks", "-d", "0"], capture_output=True, text=True, timeout=20).stdout
    except Exception:
        return None
    pw = re.search(r"Power \(W\):\s*([0-9.]+)", out)
    sc = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", out)
    return (float(pw.group(1)) if pw else None, float(sc.group(1)) if sc else None)


def measure(name, opts):
    for k, v in opts.items():
        hip.set_option(k, v)
    msm(2); hip.synchronize()
    hip.set_timing(True)
    ms, tot = [], []
    for _ in range(6):
        msm(); hip.synchronize()
        st = hip.last_msm_stats(); ms.append(st["bucket_ms"]); tot.append(st["total_ms"])
    hip.set_timing(False)
    clk = []
    for _ in range(4):
        msm(3)                                  # enqueued; the probe starts beside them, inside the first bucket kernel
        time.sleep(0.0027)
        assert pr.probe_start(8000 if np.mean(ms) > 11 else 5000) == 0
        clk.append(pr.probe_wait()); hip.synchronize()
    stop = [False]; samples = []

    def sampler():
        time.sleep(0.8)
        while not stop[0]:
            s = smi()
            if s:
                samples.append(s)
            time.sleep(0.25)
    th = threading.Thread(target=sampler); th.start()
    t0 = time.time()
    while time.time() - t0 < 4.0:
        msm(8); hip.synchronize()
    stop[0] = True; th.join()
    pw = [s[0] for s in samples if s[0]]; sc = [s[1] for s in samples if s[1]]
    print("%-58s bucket kernel %7.3f ms (min %7.3f)  MSM %7.3f ms | probe clock %s MHz | rocm-smi: %s W, sclk %s MHz"
          % (name, np.mean(ms), np.min(ms), np.mean(tot), "/".join("%.0f" % c for c in clk),
             ("%.0f" % np.mean(pw)) if pw else "n/a", ("%.0f" % np.mean(sc)) if sc else "n/a"), flush=True)
    for k in opts:
        hip.set_option(k, DEFAULTS.get(k, 0))


configs = [("shipped (12.9 GB of tables, rows anywhere)", {})]
if not PRODUCT:
    configs += [("rows masked to 128 MiB (Infinity-Cache resident)", {"diag_row_mask": (1 << 21) - 1}),
                ("rows masked to 2 MiB (L2 resident)", {"diag_row_mask": (1 << 15) - 1}),
                ("rows masked to 32 KiB (L1 resident)", {"diag_row_mask": (1 << 9) - 1})]
configs += [("acc_idxq = 0 (index stream one 4-byte entry per load, rounds 1-5)", {"acc_idxq": 0}),
            ("acc_nt = 1 (non-temporal row loads)", {"acc_nt": 1}),
            ("acc_prefetch = 0 (loads at the top of the iteration)", {"acc_prefetch": 0})]
for extra in sys.argv[1:]:
    if extra.startswith("--opt="):
        kv = dict((a.split(":")[0], int(a.split(":")[1])) for a in extra[6:].split(","))
        configs.append((extra[6:], kv))
configs.append(("shipped again (drift check)", {}))
for name, opts in configs:
    measure(name, opts)
