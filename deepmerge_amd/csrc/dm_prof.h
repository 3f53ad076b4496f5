// Optional in-library kernel timing with hipEvents on the launch stream (bench.py's roofline leg).
#pragma once
#include <hip/hip_runtime.h>

// RAII scope: records a start event now and a stop event at destruction when profiling is on.
struct DmProfScope {
  DmProfScope(const char *name, hipStream_t stream, double flops, double bytes);
  ~DmProfScope();
  int slot;
  hipStream_t stream;
};
