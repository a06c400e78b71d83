// Microbenchmark: cycles per v_mfma_f32_32x32x16_f16 for one wave per SIMD, with the A operand in AGPRs or in
// VGPRs, 4 accumulators round-robin (the dense search kernel's inner pattern).  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));

template <bool AGPR, bool WITH_LDS>
__global__ __launch_bounds__(256) void k(const half8 *src, float *out, long long *cyc, int iters) {
  __shared__ half8 lds[1024];
  const int lane = threadIdx.x & 63;
  half8 a0 = src[lane], a1 = src[64 + lane], a2 = src[128 + lane], a3 = src[192 + lane], b = src[256 + lane];
  lds[threadIdx.x] = src[threadIdx.x];
  lds[256 + threadIdx.x] = src[threadIdx.x];
  lds[512 + threadIdx.x] = src[threadIdx.x];
  lds[768 + threadIdx.x] = src[threadIdx.x];
  __syncthreads();
  float16v c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
  long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
    if (WITH_LDS) b = lds[(i * 64 + lane) & 1023];
    if (AGPR) {
      asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c0) : "a"(a0), "v"(b));
      asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c1) : "a"(a1), "v"(b));
      asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c2) : "a"(a2), "v"(b));
      asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c3) : "a"(a3), "v"(b));
    } else {
      asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c0) : "v"(a0), "v"(b));
      asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c1) : "v"(a1), "v"(b));
      asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c2) : "v"(a2), "v"(b));
      asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c3) : "v"(a3), "v"(b));
    }
  }
  asm volatile("s_nop 15\n\ts_nop 15" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3));
  long long t1 = clock64();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}

int main() {
  half8 *src; float *out; long long *cyc;
  hipMalloc(&src, 1024 * 16); hipMemset(src, 0x3c, 1024 * 16);
  hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
  const int iters = 20000;
  for (int v = 0; v < 4; ++v) {
    for (int rep = 0; rep < 3; ++rep) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0);
      if (v == 0) hipLaunchKernelGGL((k<true, false>), dim3(256), dim3(256), 0, 0, src, out, cyc, iters);
      if (v == 1) hipLaunchKernelGGL((k<false, false>), dim3(256), dim3(256), 0, 0, src, out, cyc, iters);
      if (v == 2) hipLaunchKernelGGL((k<true, true>), dim3(256), dim3(256), 0, 0, src, out, cyc, iters);
      if (v == 3) hipLaunchKernelGGL((k<false, true>), dim3(256), dim3(256), 0, 0, src, out, cyc, iters);
      hipEventRecord(e1); hipDeviceSynchronize();
      float ms; hipEventElapsedTime(&ms, e0, e1);
      long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
      double avg = 0; for (int i = 0; i < 256; ++i) avg += (double)h[i];
      avg /= 256;
      const double flop = 256.0 * 4 * iters * 4 * 32768.0;
      printf("A in %s, B from %s: %.1f cycles per MFMA, %.0f TFLOP/s, %.2f ms (clock %.2f GHz)\n", (v & 1) ? "VGPR" : "AGPR",
             v >= 2 ? "LDS " : "regs", avg / (iters * 4.0), flop / (ms * 1e-3) / 1e12, ms, avg / (ms * 1e-3) / 1e9);
    }
  }
  return 0;
}
