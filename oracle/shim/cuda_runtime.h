/* Stand-in for <cuda_runtime.h> so that the reference's dual host/device sources parse as plain C++.
 * Used ONLY by oracle/Makefile when it compiles /root/reference into oracle/_ref/ (test infrastructure). */
#pragma once
#include <algorithm>
#include <stdio.h>
#include <string.h>
#define __host__
#define __device__
#define __global__
typedef int cudaError_t;
