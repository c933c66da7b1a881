mkdir -p gpurun_out/r04m
O=gpurun_out/r04m
run() { name=$1; shift; env "$@" timeout 200 python tools/probe_16k.py > $O/$name.txt 2> $O/$name.err; echo "$name $(cat $O/$name.txt)"; }
run default_a A=1
run default_b A=1
run default_c A=1
run tail0_a NGP_FUSED_TAIL=0
run tail0_b NGP_FUSED_TAIL=0
run merge0_a NGP_MERGE_IN_ADAM=0
run merge0_b NGP_MERGE_IN_ADAM=0
run two_off_a NGP_TWO_ROUND=off
run lego8k_a WORKLOAD=lego
run lego8k_b WORKLOAD=lego
