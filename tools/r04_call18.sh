mkdir -p gpurun_out/r04r
O=gpurun_out/r04r
for i in 1 2 3; do timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-render --no-full-run > $O/bench$i.json 2> $O/bench$i.err; done
python -c "
import json
for f in ('bench1','bench2','bench3'):
    b=json.load(open('$O/%s.json'%f)); print(f, round(b['ms_per_step'],4), 'api', round(b['api_path']['ms_per_step'],4), 'plain', round(b['api_path_plain']['ms_per_step'],4), 'ratio %.3f %.3f' % (b['ms_per_step']/b['api_path']['ms_per_step'], b['ms_per_step']/b['api_path_plain']['ms_per_step']))"
