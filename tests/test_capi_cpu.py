"""CPU: the C-ABI library builds, loads and exports every symbol include/ngp_hip.h (the drop-in boundary) and
ngp_pl_amd/csrc/ngp_internal.h (library-internal launchers and white-box test hooks) declare, and nothing else; host
side logic (argument validation, level table) works without a GPU.  No compute calls here."""
import ctypes as C
import math
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    """The entry points a C COMPILER sees in the header (comments stripped first: round 3's header had one declaration inside a
    comment, which a regular expression over the raw text happily matched)."""
    from ngp_pl_amd import _abi
    return sorted(_abi.parse_all())


def test_the_boundary_stays_small():
    """include/ngp_hip.h is the drop-in boundary: the reference's 12 vren functions, the tiny-cuda-nn module operations, the
    optimizer, the native stepper, the frame renderer, the communicator -- under 100 entry points (128 in round 4, when every
    measured alternative was compiled into the ABI).  What the library's own translation units call in each other is declared
    apart."""
    from ngp_pl_amd import _abi
    pub, internal = _abi.parse(_abi.HEADER), _abi.parse(_abi.INTERNAL_HEADER)
    assert 80 <= len(pub) <= 100 and len(internal) <= 32, (len(pub), len(internal))
    for name in ("ngp_ray_aabb_intersect", "ngp_ray_sphere_intersect", "ngp_packbits", "ngp_morton3D", "ngp_morton3D_invert",
                 "ngp_raymarching_train_count", "ngp_raymarching_train_write", "ngp_raymarching_test", "ngp_composite_train_fw",
                 "ngp_composite_train_bw", "ngp_composite_test_fw", "ngp_distortion_loss_fw", "ngp_distortion_loss_bw"):
        assert name in pub, name                                    # binding.cpp:234-250, function by function


def test_header_compiles_as_c99():
    """The boundary is a C ABI: a C99 translation unit that includes nothing but the header must compile without a warning."""
    for src in ('#include "ngp_hip.h"\nint main(void) { return 0; }\n',
                '#include "ngp_hip.h"\n#include "../ngp_pl_amd/csrc/ngp_internal.h"\nint main(void) { return 0; }\n'):
        r = subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), "-x", "c", "-"],
                           input=src, text=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        assert r.returncode == 0, r.stdout


def test_c_program_links_every_declared_symbol(tmp_path):
    """A generated C program takes the address of EVERY declared entry point through a function pointer of the header's own
    prototype (a mismatch between declaration and pointer type is a compile error under -Werror), links against libngp_hip.so and
    runs: ngp_abi_version() / ngp_build_arch() through the C ABI, no Python in between."""
    from ngp_pl_amd import _abi, _lib
    protos = _abi.parse_all()
    lines = ['#include "ngp_hip.h"', '#include "../ngp_pl_amd/csrc/ngp_internal.h"', "#include <stdio.h>", "#include <string.h>", "int main(void) {", "    int n = 0;"]
    for i, (name, pr) in enumerate(sorted(protos.items())):
        lines.append("    { %s = &%s; n += (p%d != 0); }" % (pr.c_pointer_decl("p%d" % i), name, i))
    lines += ['    printf("%d %d %s\\n", n, ngp_abi_version(), ngp_build_arch());', "    return 0;", "}"]
    c = tmp_path / "link_all.c"
    c.write_text("\n".join(lines) + "\n")
    exe = tmp_path / "link_all"
    libdir = os.path.dirname(_lib.LIB_PATH)
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"), str(c), "-o", str(exe),
                        "-L", libdir, "-lngp_hip", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-Wl,--allow-shlib-undefined"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-3000:]
    r = subprocess.run([str(exe)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
    assert r.returncode == 0, r.stdout
    assert r.stdout.split() == [str(len(protos)), "6", "gfx950"], r.stdout


def test_ctypes_table_agrees_with_the_header():
    """Every hand-written argtypes list of ngp_pl_amd/_lib.py against the header's prototype: same arity, pointer where the header
    has a pointer, a scalar of the same size and kind (integer / floating) where it has a scalar."""
    from ngp_pl_amd import _abi, _lib
    protos = _abi.parse_all()
    problems = [m for m in (_abi.ctypes_agrees(a, protos[n]) for n, a in _lib._PROTOS.items() if n in protos) if m]
    assert not problems, "\n".join(problems)
    lib = _lib.lib()
    for name, pr in protos.items():                      # ... and what lib() set on the loaded functions, return types included
        f = getattr(lib, name)
        assert f.argtypes is not None and _abi.ctypes_agrees(list(f.argtypes), pr) is None, name
        want = {"int": C.c_int, "size_t": C.c_size_t, "const char*": C.c_char_p}[pr.ret]
        assert f.restype is want, (name, pr.ret, f.restype)


def test_a_declaration_inside_a_comment_is_not_a_declaration(tmp_path):
    """Round 3's defect, re-created: the prototype of ngp_nerf_loss_terms_fw moved back inside the preceding comment.  The reader
    must lose the symbol (so that test_library_exports_every_declared_symbol fails) and the C link test's translation unit must
    not compile."""
    from ngp_pl_amd import _abi
    hdr = open(_abi.HEADER).read()
    m = re.search(r"(hands back as a stride-0 view\)\. \*/\n)(int ngp_nerf_loss_terms_fw\(.*?\);\n)", hdr, re.S)
    assert m, "the header no longer has the passage this test re-breaks"
    broken = hdr[:m.start()] + "hands back as a stride-0 view).\n" + m.group(2) + " */\n" + hdr[m.end():]
    bad = tmp_path / "ngp_hip.h"
    bad.write_text(broken)
    assert "ngp_nerf_loss_terms_fw" in _abi.parse() and "ngp_nerf_loss_terms_fw" not in _abi.parse(str(bad))
    src = '#include "ngp_hip.h"\nint main(void) { return ngp_nerf_loss_terms_fw(0, 0, 0, 0.f, 0, 0, 0, 0); }\n'
    r = subprocess.run(["gcc", "-std=c99", "-Werror", "-fsyntax-only", "-I", str(tmp_path), "-x", "c", "-"], input=src, text=True,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert r.returncode != 0 and "implicit declaration" in r.stdout


def test_library_exports_every_declared_symbol():
    from ngp_pl_amd import _lib
    lib = _lib.lib()
    names = header_symbols()
    assert len(names) >= 40
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], stdout=subprocess.PIPE, text=True, check=True).stdout
    exported = set(re.findall(r" T (ngp_\w+)", out))
    assert set(names) <= exported, sorted(set(names) - exported)
    assert exported <= set(names), "exported but undeclared: %s" % sorted(exported - set(names))
    assert set(_lib.exported_symbols()) == set(names)          # the ctypes table covers the whole header
    assert lib.ngp_abi_version() == 6 == _lib.ABI_VERSION and lib.ngp_build_arch() == b"gfx950"


def test_code_object_is_gfx950_only():
    from ngp_pl_amd import _lib
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", _lib.LIB_PATH], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True).stdout
    blob = open(_lib.LIB_PATH, "rb").read()
    assert b"gfx950" in blob and b"gfx942" not in blob and b"sm_" not in blob, out[:200]


def test_argument_validation_needs_no_gpu():
    from ngp_pl_amd import _lib
    # empty inputs are a no-op (the reference handles N = 0 by launching zero blocks)
    assert _lib.call("ngp_morton3D", None, 0, None, None) == 0
    assert _lib.call("ngp_composite_train_fw", None, None, None, None, None, 1e-4, 0, 0, None, None, None, None, None, None, None) == 0
    with pytest.raises(_lib.NgpError, match="NGP_EINVAL"):
        _lib.call("ngp_morton3D", None, 5, None, None)                      # null pointers with n > 0
    with pytest.raises(_lib.NgpError, match="NGP_EINVAL"):
        _lib.call("ngp_raymarching_test", None, None, None, None, None, 1, 0.5, 0.0, 128, 1024, 0, 4, None, None, None, None, None, None)
    with pytest.raises(_lib.NgpError, match="NGP_EUNSUP"):
        fake = C.c_void_p(4096)     # never dereferenced: the configuration is rejected before any launch
        _lib.call("ngp_mlp_fwd", fake, fake, 48, 1, 3, 0, 10, fake, None)   # width 48 is not a supported input size
    with pytest.raises(_lib.NgpError, match="NGP_EINVAL"):
        _lib.call("ngp_adam_step", None, None, None, 0, None, None, 10, 1e-2, 0.9, 0.999, 1e-15, 0.0, 0, 1.0, None, None)   # step is 1-based


def test_vren_rejects_cpu_tensors():
    """CHECK_CUDA / CHECK_CONTIGUOUS of the reference (include/utils.h:4-6): no silent CPU path."""
    import torch
    import ngp_pl_amd.vren as vren
    with pytest.raises(RuntimeError, match="CUDA"):
        vren.morton3D(torch.zeros(4, 3, dtype=torch.int32))
    with pytest.raises(RuntimeError, match="CUDA"):
        vren.composite_train_fw(torch.zeros(3), torch.zeros(3, 3), torch.zeros(3), torch.zeros(3), torch.zeros(1, 3, dtype=torch.int64), 1e-4)


@pytest.mark.parametrize("scale", [0.5, 2.0, 16.0])
def test_level_table_matches_oracle(scale):
    """ngp_grid_meta_init is pure host code: compare with the oracle's float32 restatement of
    tiny-cuda-nn's level table for the reference's per_level_scale (networks.py:33)."""
    from ngp_pl_amd import _lib
    from oracle.tcnn_oracle import GridMeta
    b = math.exp(math.log(2048 * scale / 16) / 15)
    m = _lib.GridMeta()
    _lib.call("ngp_grid_meta_init", C.byref(m), 16, 2, 19, 16, float(b))
    o = GridMeta(16, 2, 19, 16, b)
    assert [m.resolution[i] for i in range(16)] == o.resolution
    assert [m.offset[i] for i in range(17)] == o.offset
    assert [float(m.scale[i]) for i in range(16)] == o.scale
    assert m.resolution[0] == 16 and all(m.offset[i + 1] - m.offset[i] <= 1 << 19 for i in range(16))
    assert all((m.offset[i + 1] - m.offset[i]) % 8 == 0 for i in range(16))


def test_missing_library_is_loud(monkeypatch, tmp_path):
    from ngp_pl_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU/eager fallback"):
        _lib.lib()


def test_workspace_queries_and_new_entry_points_validate_on_host():
    """The size queries of the workspace-taking entry points are pure host functions, and the entry
    points reject bad arguments before touching the device."""
    import ctypes as C
    import math
    from ngp_pl_amd import _lib
    lib = _lib.lib()
    meta = _lib.GridMeta()
    _lib.call("ngp_grid_meta_init", C.byref(meta), 16, 2, 19, 16, float(math.exp(math.log(2048 * 0.5 / 16) / 15)))
    # test-time frame loop: grows with rays and chunk_scale; min_samples = 4 when exp_step_factor > 0 (rendering.py:60)
    a = lib.ngp_render_test_workspace_bytes(640000, 1, 0.0)
    b = lib.ngp_render_test_workspace_bytes(640000, 4, 0.0)
    c = lib.ngp_render_test_workspace_bytes(640000, 1, 1 / 256.)
    assert 0 < a < b and c == b and lib.ngp_render_test_workspace_bytes(0, 1, 0.0) == 0
    assert a > 640000 * (12 + 12 + 4 + 4 + 64 + 4 + 12)          # per-slot sample buffers
    # occupancy update: tmp grid + cell buffers, power-of-two grids only
    o = lib.ngp_occupancy_update_workspace_bytes(1, 128)
    assert o > 128 ** 3 * (4 + 4 + 12 + 64) and lib.ngp_occupancy_update_workspace_bytes(6, 128) > o
    assert lib.ngp_occupancy_update_workspace_bytes(1, 100) == 0
    # binned backward: chunk slots of 8 entries per sample and level; refuses batches beyond 1024 chunks
    w = lib.ngp_hashgrid_bwd_binned_workspace_bytes(C.byref(meta), 300000)
    assert 16 * 300000 * 8 * 4 <= w < 16 * 300000 * 8 * 4 * 1.2
    assert lib.ngp_hashgrid_bwd_binned_workspace_bytes(C.byref(meta), 2_000_000) > 0 and lib.ngp_hashgrid_bwd_binned_workspace_bytes(C.byref(meta), 4608 * 1024 + 1) == 0
    with pytest.raises(_lib.NgpError, match="NGP_EINVAL"):
        _lib.call("ngp_render_test_frame", None, None, None, None, 1, 0.5, 0.0, 128, 1024, 1e-4, None, None, None, C.byref(meta), None, None,
                  100, 1, 0, None, None, 0, None, None, None, None, None, None)
    with pytest.raises(_lib.NgpError, match="NGP_EINVAL"):
        _lib.call("ngp_render_test_frame", None, None, None, None, 1, 0.5, 0.0, 128, 1024, 1e-4, None, None, None, C.byref(meta), None, None,
                  100, 0, 0, None, None, 0, None, None, None, None, None, None)         # chunk_scale < 1
    assert _lib.call("ngp_render_test_frame", None, None, None, None, 1, 0.5, 0.0, 128, 1024, 1e-4, None, None, None, C.byref(meta), None, None,
                     0, 1, 0, None, None, 0, None, None, None, None, None, None) == 0   # no rays: no-op
    with pytest.raises(_lib.NgpError, match="NGP_EINVAL"):
        _lib.call("ngp_occupancy_update", None, None, 1, 128, 0.5, 5.9, 0.95, None, 0, 1, None, None, None, C.byref(meta), None, None, 0, None)
    with pytest.raises(_lib.NgpError, match="NGP_EINVAL"):
        _lib.call("ngp_hashgrid_bwd_binned", None, None, None, None, C.byref(meta), 10, None, None, None, 0, None, None)
    with pytest.raises(_lib.NgpError, match="NGP_EINVAL"):
        _lib.call("ngp_hashgrid_bwd_input", None, None, None, None, None, C.byref(meta), 10, 1.0, None, None)
    assert _lib.call("ngp_sh4_bwd", None, None, 0, 1.0, None, None) == 0
    assert _lib.call("ngp_density_fwd_scatter", None, None, 0, None, None, None) == 0
    # fused composite + loss: two float rows padded to whole float4s; needs rays, every pointer and 16-byte alignment
    assert lib.ngp_composite_train_fw_loss_workspace_bytes(8192) == 8 * 8192
    assert lib.ngp_composite_train_fw_loss_workspace_bytes(777) == 8 * 780 and lib.ngp_composite_train_fw_loss_workspace_bytes(-1) == 0
    fw_loss_null = [None] * 5 + [1e-4, 16, 0] + [None] * 9 + [1e-3, 1.0] + [None] * 5 + [0, None]
    with pytest.raises(_lib.NgpError, match="NGP_EINVAL"):
        _lib.call("ngp_composite_train_fw_loss", *fw_loss_null)
    with pytest.raises(_lib.NgpError, match="NGP_EINVAL"):
        _lib.call("ngp_composite_train_fw_loss", *(fw_loss_null[:6] + [0, 0] + fw_loss_null[8:]))       # no rays: nothing to average
    # composite backward: the position copy comes with the active list only
    bw = [0x1000] * 13 + [1e-4, 4, 8, 0x1000, 0x1000]
    with pytest.raises(_lib.NgpError, match="NGP_EINVAL"):
        _lib.call("ngp_composite_train_bw", *(bw + [None, None, 0x1000, 0x1000, None]))
    with pytest.raises(_lib.NgpError, match="NGP_EINVAL"):
        _lib.call("ngp_composite_train_bw", *(bw + [0x1000, 0x1000, 0x1000, None, None]))
    # whole-field Adam: all three blocks must exist, step is 1-based
    field = [0x1000] * 5 + [100] + [0x1000] * 5 + [64] + [0x1000] * 5 + [64, 4, 1e-2, 0.9, 0.999, 1e-15, 0.0]
    with pytest.raises(_lib.NgpError, match="NGP_EINVAL"):
        _lib.call("ngp_adam_step_field", *(field + [0, 1.0, 1, None, None, None]))
    with pytest.raises(_lib.NgpError, match="NGP_EINVAL"):
        _lib.call("ngp_adam_step_field", *(field[:5] + [0] + field[6:] + [1, 1.0, 1, None, None, None]))
    with pytest.raises(_lib.NgpError, match="NGP_EINVAL"):
        _lib.call("ngp_adam_step_field", *([None] + field[1:] + [1, 1.0, 1, None, None, None]))


def test_mirrored_records_have_the_librarys_layout():
    """The ctypes mirrors of ngp_stepper_config / ngp_step_buffers are checked against sizeof() as the library was compiled (the
    loader refuses a mismatch: a stale .so would otherwise read pointers from the wrong offsets)."""
    import ctypes as C
    from ngp_pl_amd import _lib
    h = _lib.lib()
    assert h.ngp_stepper_record_bytes(0) == C.sizeof(_lib.StepperConfig)
    assert h.ngp_stepper_record_bytes(1) == C.sizeof(_lib.StepBuffersC)
    assert h.ngp_stepper_record_bytes(2) == C.sizeof(_lib.ExchangeConfig)
    assert h.ngp_stepper_record_bytes(3) < 0


@pytest.mark.parametrize("scale", [0.5, 16.0])
def test_forward_workgroup_map_covers_every_chunk_of_every_level_once(scale):
    """The hash forward's cost-balanced workgroup map (csrc/hashgrid.hip, FwdMap), walked on the CPU by
    ngp_debug_hashgrid_fwd_map: for any number of chunks every (level, chunk) is handed to exactly one workgroup, a level with a
    table above 1 MiB is served by a single XCD, and the XCDs' workgroup counts are level within the granularity of the split."""
    import ctypes as C
    import math
    import numpy as np
    from ngp_pl_amd import _lib
    meta = _lib.GridMeta()
    _lib.call("ngp_grid_meta_init", C.byref(meta), 16, 2, 19, 16, float(math.exp(math.log(2048 * scale / 16) / 15)))
    for n_chunks in list(range(1, 40)) + [255, 256, 257, 1192, 1259, 5078]:
        buf = np.full((16 * n_chunks + 8, 3), -1, np.int32)
        n = _lib.call("ngp_debug_hashgrid_fwd_map", C.byref(meta), n_chunks, buf.ctypes.data, buf.shape[0])
        assert n == 16 * n_chunks, (n_chunks, n)
        got = buf[:n]
        keys = got[:, 1].astype(np.int64) * (1 << 20) + got[:, 2]
        assert len(np.unique(keys)) == n and got[:, 2].max() == n_chunks - 1 and got[:, 1].min() == 0 and got[:, 1].max() == 15
        for l in range(16):
            size = meta.offset[l + 1] - meta.offset[l]
            xcds = np.unique(got[got[:, 1] == l, 0])
            if size * 4 > (1 << 20):
                assert len(xcds) == 1, (l, xcds)
        if n_chunks >= 256:
            per_xcd = np.bincount(got[:, 0], minlength=8)
            assert per_xcd.min() > 0 and per_xcd.max() <= 3 * n_chunks                    # at most two or three whole levels (cost-, not count-balanced)
