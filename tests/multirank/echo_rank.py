"""A rank of the CPU launcher test (tests/test_launch_cpu.py): self-launches like bench.py does, then every rank joins a gloo group, sums
its rank and prints one line. `--fail-rank q` makes rank q exit 3 (the parent must hand that code back)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from keaki_amd.launch import self_launch, under_launcher  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--gpus", type=int, default=1)
ap.add_argument("--fail-rank", type=int, default=-1)
args = ap.parse_args()
if args.gpus > 1 and not under_launcher():
    assert "torch" not in sys.modules, "the parent must not have imported torch"
    rc = self_launch(__file__, sys.argv[1:], args.gpus)
    assert "torch" not in sys.modules, "the parent imported torch while waiting"
    print("parent: child rc %d, torch imported in parent: %s" % (rc, "torch" in sys.modules), file=sys.stderr, flush=True)
    raise SystemExit(rc)
import torch
import torch.distributed as dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
t = torch.tensor([rank + 1])
dist.all_reduce(t)
if rank == args.fail_rank:
    dist.barrier()
    raise SystemExit(3)
dist.barrier()
if rank == 0:
    print('{"world": %d, "sum": %d, "self_launched": %s}' % (world, int(t.item()), "true" if os.environ.get("KEAKI_SELF_LAUNCHED") == "1" else "false"), flush=True)
dist.destroy_process_group()
