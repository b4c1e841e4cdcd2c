// tools/fp64_peak.hip -- measures the FP64 VALU issue ceilings that bound the WENO5 kernel
// (development aid).  Build: hipcc --offload-arch=gfx950 -O3 tools/fp64_peak.hip -o tools/fp64_peak.bin
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int OP>
__global__ __launch_bounds__(256) void k(double *out, int iters, double a, double b) {
  double x[8];
#pragma unroll
  for (int i = 0; i < 8; i++) x[i] = a + threadIdx.x * 1e-9 + i;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      if (OP == 0) x[i] = __builtin_fma(x[i], a, b);
      if (OP == 1) x[i] = x[i] * a;
      if (OP == 2) x[i] = x[i] + b;
      if (OP == 3) x[i] = __builtin_amdgcn_rcp(x[i]);
      if (OP == 4) { x[i] = __builtin_fma(x[i], a, b); asm volatile("v_mov_b32 %0, %0" : "+v"(it)); }
      if (OP == 5) x[i] = (double)__builtin_amdgcn_rcpf((float)x[i]);                 // cvt + rcp_f32 + cvt: 3 instructions
      if (OP == 6) x[i] = (double)((float)x[i] + 1.0f);                               // cvt + add_f32 + cvt
      if (OP == 7) {                                                                    // 16 fma + 1 rcp_f64 (the WENO mix)
        x[i] = __builtin_amdgcn_rcp(x[i]);
#pragma unroll
        for (int k = 0; k < 16; k++) x[i] = __builtin_fma(x[i], a, b);
      }
      if (OP == 9 && (i & 1) == 0) {   // two reciprocals the plain way: 2 x (rcp + Newton) = 2 rcp + 4 fma (+ 16 fma of filler per pair)
        double r0 = __builtin_amdgcn_rcp(x[i]), r1 = __builtin_amdgcn_rcp(x[i + 1]);
        r0 = __builtin_fma(__builtin_fma(-x[i], r0, 1.0), r0, r0);
        r1 = __builtin_fma(__builtin_fma(-x[i + 1], r1, 1.0), r1, r1);
        x[i] = r0; x[i + 1] = r1;
#pragma unroll
        for (int k = 0; k < 8; k++) { x[i] = __builtin_fma(x[i], a, b); x[i + 1] = __builtin_fma(x[i + 1], a, b); }
      }
      if (OP == 10 && (i & 1) == 0) {  // ... from ONE reciprocal of the product: 1 mul + rcp + 2 fma + 2 mul
        const double pr = x[i] * x[i + 1];
        double r = __builtin_amdgcn_rcp(pr);
        r = __builtin_fma(__builtin_fma(-pr, r, 1.0), r, r);
        const double r0 = r * x[i + 1], r1 = r * x[i];
        x[i] = r0; x[i + 1] = r1;
#pragma unroll
        for (int k = 0; k < 8; k++) { x[i] = __builtin_fma(x[i], a, b); x[i + 1] = __builtin_fma(x[i + 1], a, b); }
      }
      if (OP == 8) {                                                                    // 16 fma + (cvt, rcp_f32, cvt)
        x[i] = (double)__builtin_amdgcn_rcpf((float)x[i]);
#pragma unroll
        for (int k = 0; k < 16; k++) x[i] = __builtin_fma(x[i], a, b);
      }
    }
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) s += x[i];
  if (s == 12345.678) out[0] = s;
}

template <int OP>
static void run(const char *name, int wgs) {
  double *d;
  hipMalloc(&d, 8);
  const int iters = 4096;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  k<OP><<<wgs, 256>>>(d, 16, 1.0000001, 1e-9);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<OP><<<wgs, 256>>>(d, iters, 1.0000001, 1e-9);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  double inst = (double)wgs * 256 * iters * 8;  // lane-instructions
  double cyc = ms * 1e-3 * 2.4e9 / ((double)wgs * 4 / 1024.0 * iters * 8);  // cycles per wave-instr per SIMD at 2.4 GHz
  printf("%-10s wgs=%5d  %.3f ms  %.2f T lane-instr/s  (%.2f cyc/wave-instr/SIMD @2.4GHz)\n", name, wgs, ms, inst / ms / 1e9,
         cyc);
  hipFree(d);
}

// accuracy of v_rcp_f64 and of one / two Newton steps on it (decides how many the WENO weights need)
__global__ void k_rcp_acc(double *err) {
  double e0 = 0, e1 = 0, e2 = 0;
  for (int i = 0; i < 4096; i++) {
    const double d = 1.0 + (threadIdx.x * 4096 + i) * (1.0 / (256.0 * 4096.0)) * 0.999;  // [1,2)
    const double exact = 1.0 / d;
    double r = __builtin_amdgcn_rcp(d);
    e0 = fmax(e0, fabs(r - exact) / exact);
    r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
    e1 = fmax(e1, fabs(r - exact) / exact);
    r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
    e2 = fmax(e2, fabs(r - exact) / exact);
  }
  err[threadIdx.x * 3 + 0] = e0; err[threadIdx.x * 3 + 1] = e1; err[threadIdx.x * 3 + 2] = e2;
}
static void rcp_accuracy() {
  double *d, h[768];
  hipMalloc(&d, sizeof h);
  k_rcp_acc<<<1, 256>>>(d);
  hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  double m[3] = {0, 0, 0};
  for (int i = 0; i < 256; i++) for (int j = 0; j < 3; j++) if (h[3 * i + j] > m[j]) m[j] = h[3 * i + j];
  printf("v_rcp_f64 max rel err: raw %.3e  +1 Newton %.3e  +2 Newton %.3e\n", m[0], m[1], m[2]);
  hipFree(d);
}

// issue rate of v_mfma_f64_16x16x4_f64 (the block-Jacobi jobs of the fused sweeps): four independent accumulators per wave
typedef double v4d __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_mfma(double *out, int iters, double a) {
  v4d acc[4];
  for (int i = 0; i < 4; i++) acc[i] = (v4d){0.0, 0.0, 0.0, 0.0};
  const double x = a + threadIdx.x * 1e-9, y = 1.0 - threadIdx.x * 1e-9;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 4; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < 4; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 12345.678) out[0] = s;
}
static void mfma_rate(int wgs, int waves_per_simd) {
  double *d;
  hipMalloc(&d, 8);
  const int iters = 4096;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  k_mfma<<<wgs, 256>>>(d, 16, 1.0000001);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k_mfma<<<wgs, 256>>>(d, iters, 1.0000001);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double mfma_per_simd = (double)wgs * 4 * iters * 4 / 1024.0;  // wave-MFMAs per SIMD (256 CUs x 4)
  printf("mfma_f64_16x16x4  wgs=%5d (%d waves/SIMD)  %.3f ms  %.1f cycles per MFMA per SIMD @2.4GHz  %.2f TFLOP/s\n", wgs, waves_per_simd, ms,
         ms * 1e-3 * 2.4e9 / mfma_per_simd, (double)wgs * 4 * iters * 4 * 2048.0 / (ms * 1e-3) / 1e12);
  hipFree(d);
}

int main() {
  rcp_accuracy();
  mfma_rate(256, 1);
  mfma_rate(512, 2);
  mfma_rate(1024, 4);
  for (int wgs : {2048}) {
    run<0>("fma_f64", wgs);
    run<1>("mul_f64", wgs);
    run<2>("add_f64", wgs);
    run<3>("rcp_f64", wgs);
    run<4>("fma+mov", wgs);
    run<5>("cvt,rcp32,cvt (x3)", wgs);
    run<6>("cvt,add32,cvt (x3)", wgs);
    run<7>("16fma+rcp64 (x17)", wgs);
    run<8>("16fma+3 (x19)", wgs);
    run<9>("2x(rcp+N)+16fma /pair", wgs);
    run<10>("rcp(prod)+N+3mul+16fma", wgs);
  }
  return 0;
}
