mkdir -p gpurun_out/r04k
O=gpurun_out/r04k
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt -o bench -- python $REPO/bench.py --steps 20 --warmup 5 --no-render --no-cpu-baseline --no-api --no-full-run --timed-only > $REPO/$O/bench_under_trace.json 2> $REPO/$O/trace.err
DB=$(find /tmp/kt -name "*.db" | head -1)
python $REPO/tools/rocprof_summary.py "$DB" 200 > $REPO/$O/kernel_trace_summary.txt 2>> $REPO/$O/trace.err
cd $REPO
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-full-run --no-api --secondary > $O/secondary.json 2> $O/secondary.err
head -24 $O/kernel_trace_summary.txt; python -c "
import json; d=json.load(open('$O/secondary.json')); print(json.dumps(d.get('secondary'))[:1500])"
