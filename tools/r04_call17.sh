mkdir -p gpurun_out/r04q
O=gpurun_out/r04q
(time timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -25) > $O/pytest.txt 2>&1
timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-render --no-full-run > $O/bench1.json 2> $O/bench1.err
timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-render --no-full-run > $O/bench2.json 2> $O/bench2.err
timeout 200 python tools/api_host_breakdown.py > $O/api_host.json 2> $O/api_host.err
tail -5 $O/pytest.txt
python -c "
import json
for f in ('bench1','bench2'):
    b=json.load(open('$O/%s.json'%f)); print(f, b['ms_per_step'], 'api', b['api_path']['ms_per_step'], 'plain', b['api_path_plain']['ms_per_step'], 'ratio %.3f %.3f' % (b['ms_per_step']/b['api_path']['ms_per_step'], b['ms_per_step']/b['api_path_plain']['ms_per_step']))
d=json.load(open('$O/api_host.json')); print({k:(round(v['ms_per_step'],4),v['host_us']) for k,v in d.items()})"
