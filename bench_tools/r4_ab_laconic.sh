#!/bin/bash
# A/B on one box: the Laconic OT block of the bench line, an earlier build (ab_old/: see bench_tools/README.md) against the tree, alternating.
R=$PWD
for side in old new old new; do
  D=$R; [ $side = old ] && D=$R/ab_old
  python3 $D/bench.py --log2n 20 --steps 2 --warmup 1 --no-cpu-baseline --kem-log2n 0 --fk-log2d 0 > /tmp/l_$side.json 2>/dev/null
  python3 -c "
import json; d=json.loads(open('/tmp/l_$side.json').read().strip().split(chr(10))[-1]); l=d['laconic']; print('$side', 'new', l['receiver_new_s'], 'send', l['sender_send_s'], 'receive', l['receiver_receive_s'], l['all_messages_recovered'])"
done
