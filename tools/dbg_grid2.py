import sys, time, torch
sys.path.insert(0, "/root/repo")
import argparse
import bench
args = argparse.Namespace(rays=0, res=800, images=100, setup_steps=320)
dev = torch.device("cuda", 0)
loop = bench.Loop("lego", args, dev, 0, 1, None)
r = loop.run(320, 5, 20)
print('ms/step', r['ms_per_step'])
tr = loop.trainer
tr.events = []
for i in range(20):
    loop.steps(1)
    st = dict(tr.stage_times_ms())
    na = int(tr.last['n_active'].item())
    print(tr.global_step, {k: round(v, 3) for k, v in st.items() if k in ("grid_update", "adam", "hashgrid_bwd", "march_count(side stream)")})
tr.events = None
