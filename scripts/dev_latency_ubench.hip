// Dependent-issue latency of the instructions the integrators chain, measured on one wavefront per SIMD:
//   hipcc --offload-arch=gfx950 -O3 scripts/dev_latency_ubench.hip -o /tmp/ubench && /tmp/ubench
// Each test runs NCH independent chains of N dependent operations per lane and reports cycles per operation of a chain
// (s_memtime around the loop; 64 lanes, 1 workgroup per CU).
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int NCH, int KIND>
__global__ void __launch_bounds__(64) chain(double* out, long long* cyc, double a, double b, int n) {
  double x[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) x[c] = out[threadIdx.x + 64 * c];
  const long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        if (KIND == 0) x[c] = fma(a, x[c], b);                               // v_fma_f64
        else if (KIND == 1) x[c] = x[c] * a;                                 // v_mul_f64
        else if (KIND == 2) x[c] = __builtin_amdgcn_rcp(x[c]);               // v_rcp_f64
        else if (KIND == 3) {                                                // fp32 fma
          float f = (float)x[c];
          asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f) : "v"((float)a), "v"((float)b));
          x[c] = (double)f;
        } else if (KIND == 4) {                                              // DPP move of both halves + add
          const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x[c]), 0x111, 0xf, 0xf, true);
          const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x[c]), 0x111, 0xf, 0xf, true);
          x[c] = __hiloint2double(hi, lo) + b;
        }
      }
    }
  }
  const long long t1 = __builtin_readcyclecounter();
#pragma unroll
  for (int c = 0; c < NCH; ++c) out[threadIdx.x + 64 * c] = x[c];
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

__global__ void __launch_bounds__(64) lds_roundtrip(double* out, long long* cyc, int n) {
  __shared__ double sh[128];
  double x = out[threadIdx.x];
  sh[threadIdx.x] = x;
  const long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      sh[threadIdx.x] = x;
      __atomic_signal_fence(__ATOMIC_SEQ_CST);
      x = sh[(threadIdx.x + 1) & 63] + 1.0;
      __atomic_signal_fence(__ATOMIC_SEQ_CST);
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = x;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NCH, int KIND>
static void run(const char* what, double* out, long long* cyc, int n) {
  hipLaunchKernelGGL((chain<NCH, KIND>), dim3(1), dim3(64), 0, 0, out, cyc, 0.999999, 1e-9, n);
  hipDeviceSynchronize();
  hipLaunchKernelGGL((chain<NCH, KIND>), dim3(1), dim3(64), 0, 0, out, cyc, 0.999999, 1e-9, n);
  hipDeviceSynchronize();
  long long c = 0;
  hipMemcpy(&c, cyc, sizeof(c), hipMemcpyDeviceToHost);
  printf("%-28s %d chain(s): %7.2f cycles per dependent op, %6.2f per instruction\n", what, NCH, (double)c / (8.0 * n),
         (double)c / (8.0 * n * NCH));
}

int main() {
  double* out;
  long long* cyc;
  hipMalloc(&out, 64 * 16 * sizeof(double));
  hipMalloc(&cyc, 1024);
  double h[64 * 16];
  for (int i = 0; i < 64 * 16; ++i) h[i] = 1.0 + 1e-3 * i;
  hipMemcpy(out, h, sizeof(h), hipMemcpyHostToDevice);
  const int n = 4096;
  run<1, 0>("v_fma_f64", out, cyc, n);
  run<2, 0>("v_fma_f64", out, cyc, n);
  run<4, 0>("v_fma_f64", out, cyc, n);
  run<8, 0>("v_fma_f64", out, cyc, n);
  run<12, 0>("v_fma_f64", out, cyc, n);
  run<1, 1>("v_mul_f64", out, cyc, n);
  run<4, 1>("v_mul_f64", out, cyc, n);
  run<1, 2>("v_rcp_f64", out, cyc, n);
  run<4, 2>("v_rcp_f64", out, cyc, n);
  run<1, 3>("v_fma_f32 (+2 cvt)", out, cyc, n);
  run<1, 4>("dpp row_shr:1 x2 + v_add_f64", out, cyc, n);
  run<4, 4>("dpp row_shr:1 x2 + v_add_f64", out, cyc, n);
  hipLaunchKernelGGL(lds_roundtrip, dim3(1), dim3(64), 0, 0, out, cyc, n);
  hipDeviceSynchronize();
  long long c = 0;
  hipMemcpy(&c, cyc, sizeof(c), hipMemcpyDeviceToHost);
  printf("LDS write -> read other lane + v_add_f64: %7.2f cycles per round trip\n", (double)c / (8.0 * n));
  return 0;
}
