// icp_host_math.h -- host-side dense solves of the ICP loop (see icp_host_math.cpp)
#pragma once
#include <stdint.h>

namespace icp {
void svd3_rows(const double A[9], double U[9], double S[3], double Vt[9]);
int solve_point_to_point(const double* mom, double* R9, double* t3);
int solve_point_to_plane(const double* mom, double* R9, double* t3, double* x6);
void eigh3(const double A[9], double w[3], double Z[9]);
int shard_range(int64_t n, int rank, int world, int64_t* begin, int64_t* count);
}  // namespace icp
