// hip_touch.hip -- a throw-away first GPU process: hipInit, one empty launch, exit.  tools/first_process_probe.sh
// runs it ahead of bench.py to see whether the "first process of a fresh box" effect belongs to the
// first GPU process or to the first torch-hosted one.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void nothing() {}
int main() {
  if (hipInit(0) != hipSuccess) return 1;
  hipLaunchKernelGGL(nothing, dim3(1), dim3(64), 0, 0);
  if (hipDeviceSynchronize() != hipSuccess) return 2;
  puts("touched");
  return 0;
}
