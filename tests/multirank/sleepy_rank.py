"""A rank that writes its pid and sleeps: tests/test_launch_cpu.py terminates the self-launching parent and checks that no rank is left behind."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from keaki_amd.launch import self_launch, under_launcher  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--gpus", type=int, default=1)
ap.add_argument("--marker", required=True)
args = ap.parse_args()
if args.gpus > 1 and not under_launcher():
    raise SystemExit(self_launch(__file__, sys.argv[1:], args.gpus))
open("%s.%s" % (args.marker, os.environ["RANK"]), "w").write(str(os.getpid()))
time.sleep(600)
