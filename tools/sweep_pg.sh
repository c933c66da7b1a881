#!/bin/bash
# Scratch: step time under a 1-rank process group (collective path included) and without, for each march placement.
cd "$(dirname "$0")/.."
for at in top hashgrid_fwd mlp_fwd mlp_bwd; do
  for rep in 1 2; do
    a=$(NGP_MARCH_AT=$at timeout 120 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 20 --warmup 5 --timed-only 2>/dev/null | grep "^{" | python -c "import sys,json; print('%.4f' % json.loads(sys.stdin.read())['ms_per_step'])")
    b=$(NGP_MARCH_AT=$at timeout 120 python bench.py --steps 20 --warmup 5 --timed-only 2>/dev/null | grep "^{" | python -c "import sys,json; print('%.4f' % json.loads(sys.stdin.read())['ms_per_step'])")
    echo "at=$at rep=$rep  pg1=$a  plain=$b"
  done
done
