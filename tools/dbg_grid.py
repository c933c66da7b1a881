import sys, time, torch
sys.path.insert(0, "/root/repo")
import argparse
import bench
args = argparse.Namespace(rays=0, res=800, images=int(sys.argv[1]) if len(sys.argv) > 1 else 20, setup_steps=320)
dev = torch.device("cuda", 0)
loop = bench.Loop("lego", args, dev, 0, 1, None)
r = loop.run(320, 5, 20)
print('ms/step', r['ms_per_step'])
tr = loop.trainer
tr.events = []
for i in range(40):
    t = time.perf_counter()
    loop.steps(1)
    st = dict(tr.stage_times_ms())
    dt = time.perf_counter() - t
    if "grid_update" in st or i < 2:
        print(tr.global_step, "host %.2f ms" % (dt * 1e3), {k: round(v, 3) for k, v in st.items()})
tr.events = None
print("--- bench.kernel_roofline")
r = bench.kernel_roofline(loop)
print({s["stage"]: s["ms"] for s in r["stages"]})
