// icp_lcomm.h -- host-memory communicator for the ranks of ONE node (see icp_lcomm.cpp)
#pragma once
#include <string>

namespace icp {
struct LocalComm;
int lcomm_create(const void* id_128_bytes, int rank, int world, LocalComm** out, std::string& err);
void lcomm_destroy(LocalComm* c);
// v[0..count) <- sum over ranks, added in rank order on every rank (bit-identical results); count <= 32
int lcomm_allreduce_sum_f64(LocalComm* c, double* v, int count, std::string& err);
}  // namespace icp
