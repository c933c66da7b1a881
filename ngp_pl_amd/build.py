"""Builds the native pieces in-tree (no JIT cache, so the .so files travel with the repo snapshot).

  libngp_hip.so   -- the product: hand-written HIP kernels for gfx950 behind the C ABI of
                     include/ngp_hip.h.  hipcc cross-compiles without a GPU.
Run as `python -m ngp_pl_amd.build` or through `__graft_entry__.build()`.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libngp_hip.so")
SOURCES = ["march.hip", "composite.hip", "hashgrid.hip", "mlp.hip", "optim.hip", "occupancy.hip", "hashgrid_bwd_binned.hip", "stepper.hip", "comm.hip"]
HEADERS = ["ngp_common.h", "hashgrid_common.h", "loss_common.h", "comm.h", "adam_common.h", os.path.join("..", "..", "include", "ngp_hip.h")]
ARCH = "gfx950"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
CFLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden",
          "-fhip-fp32-correctly-rounded-divide-sqrt", "-Wno-unused-result"]
# per-source extras.  mlp.hip: MFMA results land in VGPRs -- every result of forward/dgrad is consumed by VALU code (convert to
# f16, ReLU), and with the accumulator-register form the compiler chose under this register pressure each of those 16-register
# results cost 16 v_accvgpr_read (208 of the 980 instructions of a backward tile)
EXTRA = {"mlp.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("command failed: %s\n%s" % (" ".join(cmd), r.stdout))
    return r.stdout


def build(force=False, verbose=False):
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    objs, jobs = [], []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s, os.path.abspath(__file__)] + hdrs):
            jobs.append([HIPCC] + CFLAGS + EXTRA.get(src, []) + ["-c", s, "-o", o])
    if jobs:
        if verbose:
            print("[ngp_pl_amd.build] compiling %d HIP sources for %s" % (len(jobs), ARCH))
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 1)) as ex:
            list(ex.map(_run, jobs))
    if force or jobs or _stale(LIB, objs):
        _run([HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC"] + objs + ["-o", LIB])
        if verbose:
            print("[ngp_pl_amd.build] linked", LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
