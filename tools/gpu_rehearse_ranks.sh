#!/bin/bash
# Multi-rank REHEARSAL of bench.py on a one-GPU box: 2 ranks sharing cuda:0, bookkeeping collectives over gloo, weak and strong
# scaling at a reduced scale.  Exercises the launch contract (torch.distributed.run, RANK / LOCAL_RANK / WORLD_SIZE), the
# sharding, the barriers and the max-over-ranks clock; the figures are not scaling results.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for MODE in weak strong; do
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
    bench.py --gpus 2 --steps 3 --warmup 1 --scale 0.2 --scaling $MODE --rehearse-on-one-gpu 2>&1 | grep '^{' | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$MODE', d['n_gpus'], round(d['value'], 1), round(d['ms_per_step'], 2), d['config']['units_per_gpu'], d['config']['parallelism'], d.get('units_verified'), d.get('rehearsal'), d.get('ranks'))"
done
