// TEST INFRASTRUCTURE.  The reference's utils.h includes <torch/extension.h> (pybind + Python);
// the CPU pin only needs the tensor library.
#pragma once
#include <torch/all.h>
#include "cuda_runtime.h"
// torch 2.10 only defines RestrictPtrTraits under a device compiler (torch/headeronly/core/
// TensorAccessor.h:24); same definition here for the host build
namespace torch { template <typename T> struct RestrictPtrTraits { typedef T* __restrict__ PtrType; }; }
