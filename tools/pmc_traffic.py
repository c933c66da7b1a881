"""HBM bytes per launch of every stage of the training step from the two PMC passes of tools/profile_bench.sh
(pmc_fetch.txt / pmc_write.txt: per-kernel means over the launches of bench.py's stage-timed steps) plus the operating
point those launches ran at (bench_pmc_fetch.json: marched / active samples per step).  bench.py reads the result from
profiles/r*_pmc_traffic.json and reports `roofline.traffic` only when its own operating point is within 5 % of it.

Units and corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE reports
half of the bytes of wide coalesced streaming reads (128-byte requests tallied at 64 B), so it is doubled; for gathers
narrower than 16 B per lane the doubling is not calibrated -- both figures are kept, `hbm_bytes_per_launch` uses the
prescribed x2.  Usage: pmc_traffic.py <dir with pmc_fetch.txt, pmc_write.txt, bench_pmc_fetch.json>"""
import json
import os
import re
import sys

STAGES = {   # stage of bench.py's roofline block -> kernels of that library call
    "hashgrid_bwd": ["bin_kernel", "apply_kernel", "merge_kernel"],
    "hashgrid_fwd": ["hashgrid_fwd_kernel"],
    "mlp_fwd": ["field_fwd_kernel"],
    "mlp_bwd": ["mlp_bwd_kernel"],
    "adam": ["adam_field"],
    "composite_fw+loss": ["composite_train_fw_kernel", "composite_fw_tail_kernel"],
    "composite_bw": ["composite_train_bw_kernel"],
    "march_write": ["march_train_write_kernel"],
}


def matches(name, kernels):
    """Does the (possibly mangled) kernel name belong to one of `kernels`?  The name must START with it, or follow the length prefix of
    an Itanium-mangled name (..._112merge_kernelE...): `merge_kernel` must not claim `adam_field_merge_kernel`."""
    return any(re.search(r"(?:^|\d)%s" % re.escape(k), name) for k in kernels)


def table(path):
    out = {}
    with open(path) as f:
        head = f.readline().split()
        cols = head[3:]
        for line in f:
            m = re.match(r"(\S.*?)\s+(\d+)\s+([\d.]+)\s+(.*)$", line.rstrip())
            if not m:
                continue
            vals = [float(v) for v in m.group(4).split()]
            out[m.group(1).strip()] = dict(calls=int(m.group(2)), avg_us=float(m.group(3)), **dict(zip(cols, vals)))
    return out


def trace_table(path):
    """kernel_trace_summary.txt (tools/rocprof_summary.py): kernel -> (calls, avg_us)."""
    out = {}
    if not os.path.exists(path):
        return out
    with open(path) as f:
        for line in f:
            m = re.match(r"(\S+)\s+(\d+)\s+([\d.]+)\s+[\d.]+\s+[\d.]+\s", line)
            if m:
                out[m.group(1)] = (int(m.group(2)), float(m.group(3)))
    return out


N_SIMD, CLOCK_MHZ = 1024, 2400.0          # 256 CUs x 4 SIMDs; MI355X peak engine clock (MI355X_MICROARCH.md)


def main():
    d = sys.argv[1]
    fetch, write = table(os.path.join(d, "pmc_fetch.txt")), table(os.path.join(d, "pmc_write.txt"))
    sq = table(os.path.join(d, "pmc_sqwait.txt")) if os.path.exists(os.path.join(d, "pmc_sqwait.txt")) else {}
    trace = trace_table(os.path.join(d, "kernel_trace_summary.txt"))
    with open(os.path.join(d, "bench_pmc_fetch.json")) as f:
        bench = json.loads([ln for ln in f.read().splitlines() if ln.startswith("{")][-1])
    roof = bench["roofline"]
    out = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over `bench.py --steps 20 --warmup 5 "
                     "--no-render --no-cpu-baseline --no-secondary --no-api`; means over the launches of the last 20 (stage-timed) steps",
           "samples_marched_per_step": roof["samples_marched_per_launch"], "samples_active_per_step": roof["samples_active_per_launch"],
           "stages": {}, "kernel_sum_ms": {}, "issue_bound": {},
           "kernel_sum_ms_how": "sum over the stage's kernels of the average duration in kernel_trace_summary.txt (rocprofv3 --kernel-trace of "
                                "`bench.py --timed-only`, the 200 timed steps, marching stream running next to them)",
           "issue_bound_how": "SQ_ACTIVE_INST_ANY (quad-cycles: x4) / (1024 SIMDs x kernel duration in the same PMC pass x 2400 MHz), the "
                              "stage's longest kernel"}
    for stage, kernels in STAGES.items():
        fb = wb = 0.0
        found = []
        for name, rec in fetch.items():
            if matches(name, kernels):
                per_step = max(1, round(rec["calls"] / max(fetch[next(n for n in fetch if "adam_field" in n)]["calls"], 1)))
                fb += rec.get("FETCH_SIZE", 0.0) * 1024 * per_step
                found.append(name)
        for name, rec in write.items():
            if matches(name, kernels):
                per_step = max(1, round(rec["calls"] / max(write[next(n for n in write if "adam_field" in n)]["calls"], 1)))
                wb += rec.get("WRITE_SIZE", 0.0) * 1024 * per_step
        ks = [(name, rec) for name, rec in trace.items() if matches(name, [k.split("_kernel")[0] for k in kernels])]
        if ks:
            adam_calls = max([c for n, (c, a) in trace.items() if "adam_field" in n] + [1])
            out["kernel_sum_ms"][stage] = round(sum(a * max(1, round(c / adam_calls)) for _n, (c, a) in ks) / 1e3, 4)
        cand = [(rec["avg_us"], name, rec) for name, rec in sq.items() if matches(name, kernels) and rec.get("SQ_ACTIVE_INST_ANY")]
        if cand:
            avg_us, name, rec = max(cand)
            out["issue_bound"][stage] = {"kernel": name, "avg_us_in_pmc_pass": avg_us, "SQ_ACTIVE_INST_ANY": rec["SQ_ACTIVE_INST_ANY"],
                                         "frac": round(rec["SQ_ACTIVE_INST_ANY"] * 4.0 / (N_SIMD * avg_us * CLOCK_MHZ), 3)}
        if found:
            out["stages"][stage] = {"kernels": found, "fetch_bytes_raw": fb, "write_bytes": wb, "hbm_bytes_per_launch": 2 * fb + wb,
                                    "how": "2 x FETCH_SIZE (gfx950 halving, calibrated for 16 B/lane streams only) + WRITE_SIZE, KB x 1024"}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
