// tools/micro/clock.hip -- what the shader clock and the fp64 VALU rate are while a kernel runs (calibrates the SQ-counter ratios):
//   hipcc --offload-arch=gfx950 -O3 tools/micro/clock.hip -o gpurun_out/clock && gpurun_out/clock
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k_chain(double *o, int n, unsigned long long *cyc) {
  double a = o[threadIdx.x], b = 1.0000001, c = 1e-9;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < n; i++) { a = __builtin_fma(a, b, c); a = __builtin_fma(a, b, c); a = __builtin_fma(a, b, c); a = __builtin_fma(a, b, c); }
  unsigned long long t1 = __builtin_readcyclecounter();
  o[threadIdx.x + blockIdx.x * blockDim.x] = a;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
__global__ void k_tput(double *o, int n) {
  double a0 = o[threadIdx.x], a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  const double b = 1.0000001, c = 1e-9;
  for (int i = 0; i < n; i++) {
    a0 = __builtin_fma(a0, b, c); a1 = __builtin_fma(a1, b, c); a2 = __builtin_fma(a2, b, c); a3 = __builtin_fma(a3, b, c);
    a4 = __builtin_fma(a4, b, c); a5 = __builtin_fma(a5, b, c); a6 = __builtin_fma(a6, b, c); a7 = __builtin_fma(a7, b, c);
  }
  o[threadIdx.x + blockIdx.x * blockDim.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
int main(int argc, char **argv) {
  const bool brief = argc > 1 && argv[1][0] == '-' && argv[1][1] == '-' && argv[1][2] == 's';   // --short: one JSON line for bench.py
  double *d; unsigned long long *dc, hc;
  hipMalloc(&d, sizeof(double) * 256 * 4096); hipMemset(d, 0, sizeof(double) * 256 * 4096); hipMalloc(&dc, 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms;
  double chain_ns = 0, chain_ms = 0, chain_mhz = 0, tput_ms = 0, tput_tf = 0;
  for (int rep = 0; rep < (brief ? 2 : 3); rep++) {      // (--short: the first repetition warms the clocks up, the second is reported)
    const int n = 2000000;
    hipEventRecord(e0); hipLaunchKernelGGL(k_chain, dim3(1), dim3(64), 0, 0, d, n, dc); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1); hipMemcpy(&hc, dc, 8, hipMemcpyDeviceToHost);
    chain_ms = ms; chain_ns = ms * 1e6 / (4.0 * n); chain_mhz = hc / (ms * 1e3);
    if (!brief)
      printf("one wave, dependent fma chain: %.1f ms for %d fma -> %.2f ns per fma; s_memtime-style counter %.0f ticks -> %.1f MHz; fma latency if 2.4 GHz: %.1f cycles\n",
             ms, 4 * n, ms * 1e6 / (4.0 * n), (double)hc, hc / (ms * 1e3), ms * 1e6 / (4.0 * n) * 2.4);
  }
  for (int rep = 0; rep < (brief ? 2 : 3); rep++) {
    const int n = 200000, blocks = 256 * 8;      // 8 waves of 4 per CU ... 2 waves per SIMD x 4 SIMDs x 256 CUs
    hipEventRecord(e0); hipLaunchKernelGGL(k_tput, dim3(blocks), dim3(256), 0, 0, d, n); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    const double fma = (double)blocks * 256 * 8.0 * n;
    tput_ms = ms; tput_tf = 2 * fma / (ms * 1e-3) / 1e12;
    if (!brief)
      printf("full chip, independent fma: %.1f ms -> %.2f TFLOP/s fp64 (2 flops per fma); per SIMD %.2f wave-fma per ns (16 lanes per cycle at f GHz = f / 4)\n",
             ms, 2 * fma / (ms * 1e-3) / 1e12, fma / 64 / 1024 / (ms * 1e6));
  }
  if (brief)
    printf("{\"dependent_fma_chain_ms\": %.3f, \"ns_per_dependent_fma\": %.4f, \"cycle_counter_mhz\": %.1f, \"full_chip_fma_ms\": %.3f, \"fp64_tflops\": %.2f}\n",
           chain_ms, chain_ns, chain_mhz, tput_ms, tput_tf);
  return 0;
}
