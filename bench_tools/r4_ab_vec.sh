#!/bin/bash
# A/B on one box: vec_encrypt / vec_decrypt / the host-array KEM calls, previous commit's libraries (ab_old/) against the tree, alternating.
R=$PWD
for side in old new old new; do
  D=$R; [ $side = old ] && D=$R/ab_old
  echo -n "$side: "; python3 $D/bench_tools/time_vec_encrypt.py 2>&1 | tail -1
done
