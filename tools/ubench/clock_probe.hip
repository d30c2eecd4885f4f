// clock_probe.hip -- development tool: what clock does the shader core hold while every SIMD issues VALU work, and how
// many cycles does a fast-class wave64 instruction (v_add_u16 / v_max_i16) occupy a SIMD then?  Each wave reads the shader
// clock counter (s_memtime) and the constant 100 MHz counter (s_memrealtime) around its loop; the ratio is the clock.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

__global__ __launch_bounds__(256) void probe(int* out, unsigned long long* clk, int iters, int x, int y) {
  int r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7;
  const unsigned long long c0 = clock64(), w0 = wall_clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
      asm volatile("v_add_u16 %0, %0, %8\n v_max_i16 %1, %1, %9\n v_add_u16 %2, %2, %8\n v_max_i16 %3, %3, %9\n"
                   "v_add_u16 %4, %4, %8\n v_max_i16 %5, %5, %9\n v_add_u16 %6, %6, %8\n v_max_i16 %7, %7, %9\n"
                   "v_add_u16 %0, %0, %8\n v_max_i16 %1, %1, %9\n v_add_u16 %2, %2, %8\n v_max_i16 %3, %3, %9\n"
                   "v_add_u16 %4, %4, %8\n v_max_i16 %5, %5, %9\n v_add_u16 %6, %6, %8\n v_max_i16 %7, %7, %9\n"
                   : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(x), "v"(y));
  }
  const unsigned long long c1 = clock64(), w1 = wall_clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;
  if ((threadIdx.x & 63) == 0) {
    const unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    clk[2 * wave] = c1 - c0;
    clk[2 * wave + 1] = w1 - w0;
  }
}

int main() {
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  for (int wps = 1; wps <= 4; wps += 3) {  // waves per SIMD: 1 and 4
    const int blocks = cus * wps;           // 256-thread blocks: 4 waves each = one per SIMD
    const int waves = blocks * 4;
    int* out;
    unsigned long long* clk;
    hipMalloc(&out, (size_t)blocks * 256 * sizeof(int));
    hipMalloc(&clk, (size_t)waves * 2 * sizeof(unsigned long long));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int iters : {2000, 60000, 400000}) {
      hipLaunchKernelGGL(probe, dim3(blocks), dim3(256), 0, 0, out, clk, 100, 3, 5);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      hipLaunchKernelGGL(probe, dim3(blocks), dim3(256), 0, 0, out, clk, iters, 3, 5);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms = 0;
      hipEventElapsedTime(&ms, e0, e1);
      std::vector<unsigned long long> h((size_t)waves * 2);
      hipMemcpy(h.data(), clk, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
      std::vector<double> ratio, cyc;
      const double instr = (double)iters * 64;
      for (int wv = 0; wv < waves; ++wv) {
        ratio.push_back((double)h[2 * wv] / (double)h[2 * wv + 1]);
        cyc.push_back((double)h[2 * wv] / instr);
      }
      std::sort(ratio.begin(), ratio.end());
      std::sort(cyc.begin(), cyc.end());
      printf("%d wave(s)/SIMD, %7.2f ms: s_memtime/s_memrealtime median %.3f (x 100 MHz), s_memtime ticks per wave-instr median %.3f, "
             "wall ns per wave-instr per SIMD %.3f\n",
             wps, ms, ratio[ratio.size() / 2], cyc[cyc.size() / 2] / wps, ms * 1e6 / (instr * wps));
    }
    hipFree(out);
    hipFree(clk);
  }
  return 0;
}
