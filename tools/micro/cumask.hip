// tools/micro/cumask.hip -- which CUs does a CU-masked HIP stream reach?  For the ranges given on the command line
// (first n pairs; default: a few), 8192 one-wavefront workgroups on hipExtStreamCreateWithCUMask streams record
// (XCC_ID, SE, SH, CU) of the CU they ran on; prints the number of distinct CUs per XCD.  Backs the statement in
// include/linefront.h that consecutive mask bits are dealt round-robin over the XCDs.
//   hipcc --offload-arch=gfx950 -O2 tools/micro/cumask.hip -o tools/micro/cumask && tools/micro/cumask 0 32 32 224
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <set>
#include <map>
__global__ void k_where(unsigned *out) {
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  for (volatile int i = 0; i < 2000; i++) { }          // stay resident long enough for the grid to spread
  if (threadIdx.x == 0) out[blockIdx.x] = ((xcc & 0xf) << 16) | (hw & 0xffff);
}
int main(int argc, char **argv) {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int total = p.multiProcessorCount;
  printf("%s: %d CUs\n", p.name, total);
  std::vector<int> ranges;
  for (int i = 1; i + 1 < argc; i += 2) { ranges.push_back(atoi(argv[i])); ranges.push_back(atoi(argv[i + 1])); }
  if (ranges.empty()) ranges = {0, total, 0, 8, 0, 32, 32, total - 32, 0, 40};
  const int N = 8192;
  unsigned *d;
  hipMalloc(&d, N * sizeof(unsigned));
  std::vector<unsigned> h(N);
  for (size_t r = 0; r + 1 < ranges.size(); r += 2) {
    const int first = ranges[r], n = ranges[r + 1];
    std::vector<uint32_t> mask((total + 31) / 32, 0u);
    for (int i = first; i < first + n && i < total; i++) mask[i / 32] |= 1u << (i % 32);
    hipStream_t st;
    if (hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data()) != hipSuccess) { printf("mask [%d, %d): stream creation failed\n", first, first + n); continue; }
    hipLaunchKernelGGL(k_where, dim3(N), dim3(64), 0, st, d);
    hipStreamSynchronize(st);
    hipMemcpy(h.data(), d, N * sizeof(unsigned), hipMemcpyDeviceToHost);
    std::map<int, std::set<unsigned>> per;
    for (int i = 0; i < N; i++) per[(int)(h[i] >> 16)].insert((h[i] >> 8) & 0xff);      // CU_ID | SH | SE bits of HW_ID
    printf("mask bits [%3d, %3d): ", first, first + n);
    int tot = 0;
    for (auto &kv : per) { printf("xcc%d:%zu ", kv.first, kv.second.size()); tot += (int)kv.second.size(); }
    printf("-> %d distinct CUs\n", tot);
    hipStreamDestroy(st);
  }
  return 0;
}
