"""Where a training step's wall time goes that no stage accounts for (VERDICT r04 "What's weak" 9: ms_per_step 0.388 vs a
main-stream stage sum of 0.341).  Measured IN PROCESS, no tracer attached (under rocprofv3 every launch costs the host ~10x more and
the loop turns host-bound: the traced walls are 2.4 ms per step, useless for this question):
  * one HIP event per step on the main stream over 640 steps -> device time of a step by its position in the 16-step occupancy cycle
    (the update runs at the head of step k = 0 mod 16, the march of that batch can only start behind it);
  * then 64 steps with the stepper's stage marks -> stage sum of the same steps, and the march's own span on its stream.
Prints the budget: wall = stage sum + per-step launch gaps + amortised update + amortised exposed march.
Usage (GPU box): python tools/step_gaps.py > profiles/r05_step_gaps.txt"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402

argv, sys.argv = sys.argv, sys.argv[:1]
try:
    args = bench.parse()
finally:
    sys.argv = argv
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
loop = bench.Loop("lego", args, dev, 0, 1, None)
loop.steps(528)                                  # the bench's operating point (320 setup + warm-up + the timed windows' first steps)
n = 640
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
t0 = time.perf_counter()
ev[0].record()
for i in range(n):
    loop.steps(1)
    ev[i + 1].record()
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / n * 1e3
gs0 = loop.trainer.global_step - n
dt = [ev[i].elapsed_time(ev[i + 1]) for i in range(n)]
by_pos = {}
for i in range(n):
    by_pos.setdefault((gs0 + i) % 16, []).append(dt[i])
mean = lambda v: sum(v) / max(len(v), 1)         # noqa: E731
plain = mean([x for p, v in by_pos.items() if p not in (0, 15) for x in v])
print("# lego, 8192 rays, steps %d..%d, no tracer.  wall %.4f ms per step (host clock around the loop); device time per step by position in the" % (gs0, gs0 + n, wall))
print("# 16-step occupancy cycle (the update is enqueued at the END of step 15's call, in front of step 0's march):")
print("position   " + " ".join("%6d" % p for p in range(16)))
print("device ms  " + " ".join("%6.3f" % mean(by_pos[p]) for p in range(16)))
cycle = sum(mean(by_pos[p]) for p in range(16))
print("# mean over the cycle %.4f ms; plain steps (positions 1..14) %.4f ms -> the update cycle costs %.1f us per step amortised" % (
    cycle / 16, plain, (cycle / 16 - plain) * 1e3))
# stage marks over 64 more steps
tr = loop.trainer
tr.events = []
acc, cnt = {}, 0
for i in range(64):
    loop.steps(1)
    if (tr.global_step - 1) % 16 in (0, 15):
        continue                                 # plain steps only
    cnt += 1
    for name, ms in tr.stage_times_ms():
        acc[name] = acc.get(name, 0.0) + ms
tr.events = None
stages = {k: v / cnt for k, v in acc.items()}
main_sum = sum(v for k, v in stages.items() if not k.startswith("march_count"))
print("# stage marks (plain steps, %d of them): %s" % (cnt, ", ".join("%s %.4f" % (k, v) for k, v in stages.items())))
print("# main-stream stage sum %.4f ms; plain-step device time %.4f ms -> %.1f us per step between / around the stages (event marks themselves, launch gaps," % (
    main_sum, plain, (plain - main_sum) * 1e3))
print("#   the wait for the march's count at the top of the step when the march finishes after the previous step's Adam)")
print("# budget per step: stage sum %.1f + gaps %.1f + update cycle %.1f = %.1f us; measured wall %.1f us" % (
    main_sum * 1e3, (plain - main_sum) * 1e3, (cycle / 16 - plain) * 1e3, cycle / 16 * 1e3, wall * 1e3))
