// TEST INFRASTRUCTURE.  Forwards to the reference's helper_math.h with __CUDACC__ defined, so
// that it does not replace fminf/fmaxf with its NaN-unsafe host versions (helper_math.h:51-90):
// the device code path being pinned uses the IEEE fminf/fmaxf.
#pragma once
#define __CUDACC__ 1
#include_next "helper_math.h"
#undef __CUDACC__
