"""Which HIP runtimes a process ends up with, for both import orders (debugging aid):  python3 bench_tools/dbg_torch_after_keaki.py [torch_first]"""
import os, sys
sys.path.insert(0, os.getcwd())
order = sys.argv[1] if len(sys.argv) > 1 else "keaki_first"
def maps():
    return sorted({l.split()[-1] for l in open("/proc/self/maps") if "amdhip64" in l or "hsa-runtime" in l})
if order == "torch_first":
    import torch
    print("torch available:", torch.cuda.is_available())
from keaki_amd.hip import KeakiHip
h = KeakiHip(0)
print("selftest", h.selftest_field(4, 2, 1))
print("loaded after keaki:", maps())
import torch
print("device_count", torch.cuda.device_count(), "available", torch.cuda.is_available())
print("loaded after torch:", maps())
try:
    x = torch.zeros(4, device="cuda")
    print("torch tensor ok", x.sum().item())
except Exception as e:
    print("torch failed:", e)
h.close()
