// Microbenchmark lab for the dense-search inner loop: one wave per SIMD, A fragments in AGPRs, 4 accumulators,
// 16 k-steps per "query block".  Features are switched on one by one to price them (cycles per MFMA):
//   bit 0: B fragments from LDS through a 3-deep register ring (else from a register)
//   bit 1: 8 v_max3 per k-step on unrelated registers (the previous block's epilogue folds)
//   bit 2: s_barrier once per block (between k-steps 0 and 1)
//   bit 3: 4 global_load_lds_dwordx4 per block + s_waitcnt vmcnt(0) before the barrier
//   bit 4: s_nop 15 + s_nop 3 after the last k-step of the block
//   bit 5: an LDS read whose value is needed at once in the middle of the block (threshold fetch)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));

template <int F>
__global__ __launch_bounds__(256) void k(const half8 *src, float *out, long long *cyc, int blocks) {
  __shared__ half8 lds[3 * 1024];
  __shared__ float thr[64];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  half8 a[4][16];
#pragma unroll
  for (int v = 0; v < 4; ++v)
#pragma unroll
    for (int s = 0; s < 16; ++s) a[v][s] = src[(v * 16 + s) * 64 % 1024 + lane];
  for (int i = t; i < 3 * 1024; i += 256) lds[i] = src[i & 1023];
  if (t < 64) thr[t] = 1e30f;
  __syncthreads();
  float16v c[4];
#pragma unroll
  for (int v = 0; v < 4; ++v) c[v] = float16v{0};
  float m0 = 0, m1 = 1, m2 = 2, m3 = 3, acc = 0;
  half8 bq[4];
  bq[0] = lds[lane]; bq[1] = lds[64 + lane]; bq[2] = lds[128 + lane];
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) half8 *)lds;
  long long t0 = clock64();
  int slot = 0;
  for (int blk = 0; blk < blocks; ++blk) {
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      if (s == 1) {
        if (F & 8) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (F & 4) __builtin_amdgcn_s_barrier();
        if (F & 8) {
          const int ns = slot == 0 ? 2 : slot - 1;
#pragma unroll
          for (int cc = 0; cc < 4; ++cc) {
            uint32_t keep;
            const half8 *g = src + ((cc * 4 + w) * 64 + lane);
            const uint32_t l = lds0 + (uint32_t)((ns * 1024 + (cc * 4 + w) * 64) * 16);
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(g), "s"(__builtin_amdgcn_readfirstlane(l)) : "memory");
          }
        }
      }
      if (F & 1) {
        if (s + 3 < 16) bq[(s + 3) & 3] = lds[slot * 1024 + (s + 3) * 64 + lane];
        else bq[(s + 3) & 3] = lds[(slot == 2 ? 0 : slot + 1) * 1024 + (s + 3 - 16) * 64 + lane];
      }
      if (s == 15 && (F & 16)) {
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %4, %8, %0\n\tv_mfma_f32_32x32x16_f16 %1, %5, %8, %1\n\t"
                     "v_mfma_f32_32x32x16_f16 %2, %6, %8, %2\n\tv_mfma_f32_32x32x16_f16 %3, %7, %8, %3\n\ts_nop 15\n\ts_nop 3"
                     : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3])
                     : "a"(a[0][15]), "a"(a[1][15]), "a"(a[2][15]), "a"(a[3][15]), "v"(bq[s & 3]));
      } else {
#pragma unroll
        for (int v = 0; v < 4; ++v)
          asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c[v]) : "a"(a[v][s]), "v"(bq[(F & 1) ? (s & 3) : 0]));
      }
      if (F & 2) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          m0 = fmaxf(fmaxf(m0, m1), m2);
          m1 = fmaxf(fmaxf(m1, m2), m3);
          m2 = fmaxf(fmaxf(m2, m3), m0);
          m3 = fmaxf(fmaxf(m3, m0), m1);
        }
      }
      if (s == 8 && (F & 32)) {
        const float th = thr[lane & 31];
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(m0 >= th) != 0, 0)) acc += 1.0f;
      }
    }
    slot = slot == 2 ? 0 : slot + 1;
  }
  asm volatile("s_nop 15\n\ts_nop 15" : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]));
  long long t1 = clock64();
  if (t == 0) cyc[blockIdx.x] = t1 - t0;
  out[blockIdx.x * 256 + t] = c[0][0] + c[1][1] + c[2][2] + c[3][3] + m0 + m1 + m2 + m3 + acc;
}

template <int F>
void run(const half8 *src, float *out, long long *cyc, const char *what) {
  const int blocks = 512;
  float best = 1e9f;
  double cy = 0;
  for (int rep = 0; rep < 3; ++rep) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<F>), dim3(256), dim3(256), 0, 0, src, out, cyc, blocks);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double avg = 0; for (int i = 0; i < 256; ++i) avg += (double)h[i];
    if (ms < best) { best = ms; cy = avg / 256 / (blocks * 64.0); }
  }
  const double flop = 256.0 * 4 * blocks * 64 * 32768.0;
  printf("F=%2d %-58s %.1f cycles per MFMA, %.0f TFLOP/s\n", F, what, cy, flop / (best * 1e-3) / 1e12);
}

int main() {
  half8 *src; float *out; long long *cyc;
  (void)hipMalloc(&src, 1024 * 16); (void)hipMemset(src, 0x3c, 1024 * 16);
  (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&cyc, 256 * 8);
  run<0>(src, out, cyc, "bare");
  run<1>(src, out, cyc, "+ B ring from LDS");
  run<3>(src, out, cyc, "+ 8 v_max3 per k-step");
  run<7>(src, out, cyc, "+ barrier per block");
  run<15>(src, out, cyc, "+ 4 LDS-DMA per block, vmcnt(0)");
  run<31>(src, out, cyc, "+ s_nop 20 after the last k-step");
  run<63>(src, out, cyc, "+ threshold read used at once");
  run<1 | 2 | 16 | 32>(src, out, cyc, "ring + folds + nops + threshold, no barrier / DMA");
  return 0;
}
