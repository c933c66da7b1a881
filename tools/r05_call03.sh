#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
OUT=gpurun_out
echo "== one marching stream per Trainer (round 4 behaviour)" > $OUT/r05_c03_variance.txt
NGP_SHARED_SIDE=0 timeout 300 python tools/loop_variance.py lego_hard 8 >> $OUT/r05_c03_variance.txt 2>$OUT/r05_c03_a.err
echo "== one marching stream per process" >> $OUT/r05_c03_variance.txt
NGP_SHARED_SIDE=1 timeout 300 python tools/loop_variance.py lego_hard 8 >> $OUT/r05_c03_variance.txt 2>$OUT/r05_c03_b.err
cat $OUT/r05_c03_variance.txt; tail -3 $OUT/r05_c03_a.err
