#include <hip/hip_runtime.h>
#include <cstdio>
// out[lane] = tab[sel] (sel wave-uniform 0..3) via VGPR index mode: v_mov_b32 with SRC0 relative
__global__ void k(const float* in, const int* sel, float* out) {
  float t0 = in[threadIdx.x], t1 = in[64 + threadIdx.x], t2 = in[128 + threadIdx.x], t3 = in[192 + threadIdx.x];
  int s = __builtin_amdgcn_readfirstlane(sel[0]);
  float r;
  asm volatile(
      "v_mov_b32 v200, %1\n\t"
      "v_mov_b32 v201, %2\n\t"
      "v_mov_b32 v202, %3\n\t"
      "v_mov_b32 v203, %4\n\t"
      "s_set_gpr_idx_on %5, 0x1\n\t"
      "v_mov_b32 %0, v200\n\t"
      "s_set_gpr_idx_off\n\t"
      : "=v"(r) : "v"(t0), "v"(t1), "v"(t2), "v"(t3), "s"(s) : "v200", "v201", "v202", "v203", "m0");
  out[threadIdx.x] = r;
}
__global__ void k2(const double* in, const int* sel, double* out) {
  double t0 = in[threadIdx.x], t1 = in[64 + threadIdx.x], t2 = in[128 + threadIdx.x], t3 = in[192 + threadIdx.x];
  int s = __builtin_amdgcn_readfirstlane(sel[0]) * 2;
  double acc = 1000.0;
  asm volatile(
      "v_mov_b64 v[200:201], %1\n\t"
      "v_mov_b64 v[202:203], %2\n\t"
      "v_mov_b64 v[204:205], %3\n\t"
      "v_mov_b64 v[206:207], %4\n\t"
      "s_set_gpr_idx_on %5, 0x1\n\t"
      "v_add_f64 %0, v[200:201], %0\n\t"
      "s_set_gpr_idx_off\n\t"
      : "+v"(acc) : "v"(t0), "v"(t1), "v"(t2), "v"(t3), "s"(s) : "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "m0");
  out[threadIdx.x] = acc;
}
int main() {
  float *in, *out; int* sel; double *din, *dout;
  hipMallocManaged(&in, 256 * 4); hipMallocManaged(&out, 64 * 4); hipMallocManaged(&sel, 4); hipMallocManaged(&din, 256 * 8); hipMallocManaged(&dout, 64 * 8);
  for (int i = 0; i < 256; ++i) { in[i] = (float)(i / 64 * 1000 + i % 64); din[i] = (double)(i / 64 * 100 + i % 64); }
  int bad = 0;
  for (int s = 0; s < 4; ++s) {
    sel[0] = s;
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, in, sel, out); hipDeviceSynchronize();
    hipLaunchKernelGGL(k2, dim3(1), dim3(64), 0, 0, din, sel, dout); hipDeviceSynchronize();
    for (int l = 0; l < 64; ++l) { if (out[l] != in[s * 64 + l]) ++bad; if (dout[l] != 1000.0 + din[s * 64 + l]) ++bad; }
    printf("sel %d: out[5] = %g (want %g), dout[5] = %g (want %g)\n", s, out[5], in[s * 64 + 5], dout[5], 1000.0 + din[s * 64 + 5]);
  }
  printf("bad %d\n", bad);
  return bad != 0;
}
