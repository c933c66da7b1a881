mkdir -p gpurun_out/r04n
O=gpurun_out/r04n
run() { name=$1; shift; env "$@" timeout 200 python tools/probe_16k.py > $O/$name.txt 2> $O/$name.err; echo "$name $(cat $O/$name.txt | cut -c1-400) $(tail -1 $O/$name.err | cut -c1-200)"; }
run k16_a A=1
run k16_b A=1
run lego_a WORKLOAD=lego
(timeout 600 python -m pytest "tests/test_train_gpu.py::test_training_is_reproducible_run_to_run" "tests/test_train_gpu.py::test_merge_folded_into_adam_is_bit_identical" tests/test_field_gpu.py -q -p no:cacheprovider -k "reproducible or merge_folded or hashgrid_backward" 2>&1 | tail -4)
timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-render --no-full-run --no-api > $O/bench.json 2> $O/bench.err; python -c "
import json; d=json.load(open('$O/bench.json')); print(d['value'], d['ms_per_step'], d['config']['samples_per_ray_composited'])"
