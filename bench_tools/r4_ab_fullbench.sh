#!/bin/bash
R=$PWD
for side in new old new old new; do
  D=$R; [ $side = old ] && D=$R/ab_old
  python3 $D/bench.py --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); l=d['laconic']; print('$side', d['ms_per_step'], 'setup', l['setup_s'], 'new', l['receiver_new_s'], 'send', l['sender_send_s'], 'receive', l['receiver_receive_s'], 'pairings', round(d['kem']['pairings_per_s']/1e6,2))"
done
cat /proc/loadavg
