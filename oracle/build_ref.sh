#!/bin/bash
# TEST INFRASTRUCTURE.  Compiles the reference's OWN kernel sources -- in place, from
# /root/reference/models/csrc, never copied into this repo -- for the host CPU, producing
#   oracle/_ref/libvren_ref_nofma.so   (g++ -ffp-contract=off)
#   oracle/_ref/libvren_ref_fma.so     (g++ -mfma -ffp-contract=fast, i.e. nvcc-like contraction)
# They pin oracle/ngp_oracle.c (tests/test_oracle_vs_ref.py) and are the "reference" CPU
# baseline of bench.py.  The stream edits below are the three things g++/torch-2.10 cannot
# digest; none changes arithmetic:
#   k<<<grid, block>>>(args)               -> ref_launch(grid, block, k, args)  (sequential loop)
#   AT_DISPATCH_*(x.type(), ...)           -> x.scalar_type()   (removed torch API)
#   AT_DISPATCH_FLOATING_TYPES_AND_HALF    -> AT_DISPATCH_FLOATING_TYPES (c10::Half has no CPU
#                                             __expf/thrust path; the hot path is f32 only)
# The reference's own build system (setup.py / nvcc) is not used.  Skips silently when
# /root/reference is absent (GPU box): the prebuilt .so files travel with the repo snapshot.
#
# It also STAGES the reference's own Python hot path, unmodified, under oracle/_ref/py/ (git-ignored like the
# rest of oracle/_ref, shipped to the GPU box with the snapshot): models/{__init__,rendering,networks,
# custom_functions}.py and losses.py.  tests/test_reference_files_gpu.py and bench.py's
# `api_path_reference_files` leg execute THOSE files on the GPU over this package's `vren` / `tinycudann`
# bindings (oracle/ref_on_binding.py) -- INTEGRATION.md "Option A" run for real.  Round 6 adds train.py, opt.py, utils.py and
# datasets/{base,ray_utils}.py: tests/test_reference_train_gpu.py executes the reference's train.py itself
# (oracle/ref_train_harness.py supplies the packages this image lacks).  Nothing is copied into history.
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
REF="${NGP_REFERENCE_DIR:-/root/reference}/models/csrc"
OUT="$HERE/_ref"
if [ ! -d "$REF" ]; then echo "[build_ref] $REF not present; keeping prebuilt $OUT"; exit 0; fi
mkdir -p "$OUT"
PYREF="$(dirname "$(dirname "$REF")")"
mkdir -p "$OUT/py/models"
for f in __init__ rendering networks custom_functions; do cp -pf "$PYREF/models/$f.py" "$OUT/py/models/$f.py"; done
cp -pf "$PYREF/losses.py" "$OUT/py/losses.py"
# ... and what tests/test_reference_train_gpu.py needs to execute the reference's train.py itself (oracle/ref_train_harness.py)
mkdir -p "$OUT/py/datasets"
for f in train opt utils; do cp -pf "$PYREF/$f.py" "$OUT/py/$f.py"; done
for f in base ray_utils; do cp -pf "$PYREF/datasets/$f.py" "$OUT/py/datasets/$f.py"; done
echo "[build_ref] staged the reference's models/*.py + losses.py + train.py, opt.py, utils.py, datasets/{base,ray_utils}.py (unmodified) under $OUT/py"
PY="${PYTHON:-python3}"
TORCH_INC=$($PY -c "import torch.utils.cpp_extension as c; print(' '.join('-I'+p for p in c.include_paths()))")
TORCH_LIB=$($PY -c "import torch, os; print(os.path.join(os.path.dirname(torch.__file__), 'lib'))")
CXX="${CXX:-g++}"
COMMON="-x c++ -std=c++17 -O1 -fPIC -w -D_GLIBCXX_USE_CXX11_ABI=$($PY -c 'import torch; print(int(torch._C._GLIBCXX_USE_CXX11_ABI))') -I$HERE/ref_shim -I$REF/include $TORCH_INC"
EDIT=(-e 's/\([A-Za-z_0-9]\+\(<scalar_t>\)\?\)<<<\([^>]*\)>>>(/ref_launch(\3, \1, /'
      -e 's/\.type()/.scalar_type()/g'
      -e 's/AT_DISPATCH_FLOATING_TYPES_AND_HALF/AT_DISPATCH_FLOATING_TYPES/')
build_variant() {
    local name="$1"; shift
    local fp="$*"
    local objs=()
    for f in intersection raymarching volumerendering losses; do
        local newest="$OUT/${f}_${name}.o"
        if [ ! -f "$newest" ] || [ "$REF/$f.cu" -nt "$newest" ] || [ "$HERE/ref_shim/cuda_runtime.h" -nt "$newest" ]; then
            sed "${EDIT[@]}" "$REF/$f.cu" | $CXX $COMMON $fp -c - -o "$newest" &
        fi
        objs+=("$newest")
    done
    local b="$OUT/ref_binding_${name}.o"
    if [ ! -f "$b" ] || [ "$HERE/ref_shim/ref_binding.cpp" -nt "$b" ] || [ "$HERE/ref_shim/cuda_runtime.h" -nt "$b" ]; then
        $CXX $COMMON $fp -c "$HERE/ref_shim/ref_binding.cpp" -o "$b" &
    fi
    wait
    $CXX -shared -o "$OUT/libvren_ref_${name}.so" "${objs[@]}" "$b" -L"$TORCH_LIB" -Wl,-rpath,"$TORCH_LIB" -ltorch -ltorch_cpu -lc10
    echo "[build_ref] built $OUT/libvren_ref_${name}.so"
}
build_variant nofma -ffp-contract=off
build_variant fma -O2 -mfma -ffp-contract=fast   # -O2: gcc only forms FMAs with -fexpensive-optimizations
