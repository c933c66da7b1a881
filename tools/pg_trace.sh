cd /tmp && export TMPDIR=/tmp
export RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533
rm -rf /tmp/kt2; timeout 300 rocprofv3 --kernel-trace -d /tmp/kt2 -o bench -- python /root/repo/bench.py --steps 20 --warmup 5 --timed-only > /root/repo/gpurun_out/pg_trace_bench.json 2> /root/repo/gpurun_out/pg_trace.err
DB=$(find /tmp/kt2 -name "*.db" | head -1)
python /root/repo/tools/step_timeline.py $DB 2 > /root/repo/gpurun_out/pg_timeline.txt 2>&1
python /root/repo/tools/rocprof_summary.py $DB 100 > /root/repo/gpurun_out/pg_trace_summary.txt 2>&1
