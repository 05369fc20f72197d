#!/bin/bash
# A/B of the lane=row LDS-DMA kernel's tile shapes (LAB_BCAST_TC x LAB_BCAST_DEPTH) on the full bench step
for cfg in "2 3" "4 2" "2 2" "4 3"; do set -- $cfg
  for b in "$@"; do :; done
  LAB_BCAST_TC=$1 LAB_BCAST_DEPTH=$2 python bench.py --no-cpu-baseline --no-extras --batch ${BATCH:-8} --steps 5 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('TC $1 DEPTH $2 batch', d['config']['proofs_per_step_per_gpu'], 'proofs/s %.1f  ms/proof %.3f  kernel GB/s %.0f' % (d['value'], d['ms_per_proof'], d['roofline']['achieved']))"
done
