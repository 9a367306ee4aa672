// icp_comm.h -- run-time bound RCCL wrapper for the loop's single all-reduce (see icp_comm.cpp)
#pragma once
#include <hip/hip_runtime.h>

#include <string>

namespace icp {
int comm_unique_id(void* out_128_bytes, std::string& err);
int comm_init(const void* id_128_bytes, int rank, int world, void** comm_out, std::string& err);
void comm_destroy(void* comm);
int comm_allreduce_sum_f64(void* comm, double* dev_buf, int count, hipStream_t stream, std::string& err);
}  // namespace icp
