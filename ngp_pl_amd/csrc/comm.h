// Internal: what csrc/stepper.hip needs of the communicator (csrc/comm.hip).  Not part of the C ABI (include/ngp_hip.h declares
// ngp_comm as an opaque type and the ngp_comm_* entry points).
#pragma once
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

struct ngp_comm {
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;       // the communicator's own (high-priority) stream
    int world = 1, rank = 0, device = 0, version = 0;
};

int ngp_comm_check(ncclResult_t r, const char* what);      // 0 or NGP_ECOMM (message kept for ngp_comm_last_error)
int ngp_comm_group_begin();
int ngp_comm_group_end();
