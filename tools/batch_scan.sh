#!/bin/bash
# per-launch fixed cost vs per-tile cost: C5 at 1, 2, 4, 8, 13 batches per sweep
for k in 51 26 13 7 4; do
  b=$((196608 * k))
  timeout -k 10 120 python bench.py --no-cpu --no-extra --batch $b "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('batch',$b,'ms/sweep',round(d['ms_per_step'],4),'kernel ms',round(r['kernel_ms_per_sweep'],4),'launches',r['launches_per_sweep'],'threads',r['threads'])" || exit 1
done
