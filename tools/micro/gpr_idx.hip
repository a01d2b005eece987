// VGPR index mode on gfx950 (s_set_gpr_idx_on / _idx / _off): does it work, and what does a change of the index cost?
//   hipcc -O2 -w --offload-arch=gfx950 tools/micro/gpr_idx.hip -o /tmp/gpr_idx && /tmp/gpr_idx
// (1) correctness: out[lane] = T[sel][lane] through v_mov_b32 / v_add_f64 with SRC0 relative to the index, sel = 0..3;
// (2) timing: per wavefront, N groups of {index change, two dependent-free v_add_f64} against N groups of two v_add_f64 alone, at 1, 2
//     and 4 wavefronts per SIMD: cycles per group = what k_doublet_cls's uniform-j form pays per sample j.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ __launch_bounds__(64) void k_sel32(const float* in, const int* sel, float* out) {
  float t0 = in[threadIdx.x], t1 = in[64 + threadIdx.x], t2 = in[128 + threadIdx.x], t3 = in[192 + threadIdx.x];
  int s = __builtin_amdgcn_readfirstlane(sel[0]);
  float r;
  asm volatile("v_mov_b32 v200, %1\n\tv_mov_b32 v201, %2\n\tv_mov_b32 v202, %3\n\tv_mov_b32 v203, %4\n\t"
               "s_set_gpr_idx_on %5, 0x1\n\tv_mov_b32 %0, v200\n\ts_set_gpr_idx_off\n\t"
               : "=v"(r) : "v"(t0), "v"(t1), "v"(t2), "v"(t3), "s"(s) : "v200", "v201", "v202", "v203", "m0");
  out[threadIdx.x] = r;
}
__global__ __launch_bounds__(64) void k_sel64(const double* in, const int* sel, double* out) {
  double t0 = in[threadIdx.x], t1 = in[64 + threadIdx.x], t2 = in[128 + threadIdx.x], t3 = in[192 + threadIdx.x];
  int s = __builtin_amdgcn_readfirstlane(sel[0]) * 2;
  double acc = 1000.0;
  asm volatile("v_mov_b64 v[200:201], %1\n\tv_mov_b64 v[202:203], %2\n\tv_mov_b64 v[204:205], %3\n\tv_mov_b64 v[206:207], %4\n\t"
               "s_set_gpr_idx_on %5, 0x1\n\tv_add_f64 %0, v[200:201], %0\n\ts_set_gpr_idx_off\n\t"
               : "+v"(acc) : "v"(t0), "v"(t1), "v"(t2), "v"(t3), "s"(s) : "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "m0");
  out[threadIdx.x] = acc;
}

// MODE 0: adds only; 1: s_bfe + s_set_gpr_idx_on per group; 2: s_bfe + s_set_gpr_idx_idx per group (mode switched on once)
template <int MODE>
__global__ __launch_bounds__(256) void k_time(const double* in, double* out, int n, unsigned ids) {
  double t0 = in[threadIdx.x & 63], t1 = t0 + 1.0;
  double a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0, a7 = 0;
  unsigned w = __builtin_amdgcn_readfirstlane(ids), tmp;
  for (int i = 0; i < n; ++i) {
    if (MODE == 0)
      asm volatile("v_add_f64 %0, v[200:201], %0\n\tv_add_f64 %1, v[202:203], %1\n\tv_add_f64 %2, v[200:201], %2\n\tv_add_f64 %3, v[202:203], %3\n\t"
                   "v_add_f64 %4, v[200:201], %4\n\tv_add_f64 %5, v[202:203], %5\n\tv_add_f64 %6, v[200:201], %6\n\tv_add_f64 %7, v[202:203], %7\n\t"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(t0), "v"(t1) : "v200", "v201", "v202", "v203");
    else if (MODE == 1)
      asm volatile("s_bfe_u32 %[t], %[w], 0x20000\n\ts_set_gpr_idx_on %[t], 0x1\n\tv_add_f64 %0, v[200:201], %0\n\tv_add_f64 %1, v[202:203], %1\n\t"
                   "s_bfe_u32 %[t], %[w], 0x20002\n\ts_set_gpr_idx_on %[t], 0x1\n\tv_add_f64 %2, v[200:201], %2\n\tv_add_f64 %3, v[202:203], %3\n\t"
                   "s_bfe_u32 %[t], %[w], 0x20004\n\ts_set_gpr_idx_on %[t], 0x1\n\tv_add_f64 %4, v[200:201], %4\n\tv_add_f64 %5, v[202:203], %5\n\t"
                   "s_bfe_u32 %[t], %[w], 0x20006\n\ts_set_gpr_idx_on %[t], 0x1\n\tv_add_f64 %6, v[200:201], %6\n\tv_add_f64 %7, v[202:203], %7\n\t"
                   "s_set_gpr_idx_off\n\t"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), [t] "=&s"(tmp) : [w] "s"(w), "v"(t0), "v"(t1)
                   : "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", "m0", "scc");
    else if (MODE == 3)   // the index written into M0 directly (its bits 15:12 carry the mode: the images come with 0x1000 set)
      asm volatile("s_set_gpr_idx_on %[w], 0x1\n\t"
                   "s_or_b32 m0, %[w], 0x1000\n\tv_add_f64 %0, v[200:201], %0\n\tv_add_f64 %1, v[202:203], %1\n\t"
                   "s_or_b32 m0, %[w], 0x1000\n\tv_add_f64 %2, v[200:201], %2\n\tv_add_f64 %3, v[202:203], %3\n\t"
                   "s_or_b32 m0, %[w], 0x1000\n\tv_add_f64 %4, v[200:201], %4\n\tv_add_f64 %5, v[202:203], %5\n\t"
                   "s_or_b32 m0, %[w], 0x1000\n\tv_add_f64 %6, v[200:201], %6\n\tv_add_f64 %7, v[202:203], %7\n\t"
                   "s_set_gpr_idx_off\n\t"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), [t] "=&s"(tmp) : [w] "s"(w), "v"(t0), "v"(t1)
                   : "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", "m0", "scc");
    else if (MODE == 4)   // round 4: the index image shifted into M0 (s_lshr_b32, a 32-bit encoding), ONE addition per change (FAST's phase 2)
      asm volatile("s_set_gpr_idx_on %[w], 0x1\n\t"
                   "s_lshr_b32 m0, %[w], 0\n\tv_add_f64 %0, v[200:201], %0\n\ts_lshr_b32 m0, %[w], 0\n\tv_add_f64 %1, v[202:203], %1\n\t"
                   "s_lshr_b32 m0, %[w], 0\n\tv_add_f64 %2, v[200:201], %2\n\ts_lshr_b32 m0, %[w], 0\n\tv_add_f64 %3, v[202:203], %3\n\t"
                   "s_lshr_b32 m0, %[w], 0\n\tv_add_f64 %4, v[200:201], %4\n\ts_lshr_b32 m0, %[w], 0\n\tv_add_f64 %5, v[202:203], %5\n\t"
                   "s_lshr_b32 m0, %[w], 0\n\tv_add_f64 %6, v[200:201], %6\n\ts_lshr_b32 m0, %[w], 0\n\tv_add_f64 %7, v[202:203], %7\n\t"
                   "s_set_gpr_idx_off\n\t"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), [t] "=&s"(tmp) : [w] "s"(w | 0x1000u), "v"(t0), "v"(t1)
                   : "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", "m0", "scc");
    else if (MODE == 5)   // ... TWO additions per change (STRICT's phase 2 as built)
      asm volatile("s_set_gpr_idx_on %[w], 0x1\n\t"
                   "s_lshr_b32 m0, %[w], 0\n\tv_add_f64 %0, v[200:201], %0\n\tv_add_f64 %1, v[202:203], %1\n\t"
                   "s_lshr_b32 m0, %[w], 0\n\tv_add_f64 %2, v[200:201], %2\n\tv_add_f64 %3, v[202:203], %3\n\t"
                   "s_lshr_b32 m0, %[w], 0\n\tv_add_f64 %4, v[200:201], %4\n\tv_add_f64 %5, v[202:203], %5\n\t"
                   "s_lshr_b32 m0, %[w], 0\n\tv_add_f64 %6, v[200:201], %6\n\tv_add_f64 %7, v[202:203], %7\n\t"
                   "s_set_gpr_idx_off\n\t"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), [t] "=&s"(tmp) : [w] "s"(w | 0x1000u), "v"(t0), "v"(t1)
                   : "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", "m0", "scc");
    else if (MODE == 6)   // ... FOUR additions per change
      asm volatile("s_set_gpr_idx_on %[w], 0x1\n\t"
                   "s_lshr_b32 m0, %[w], 0\n\tv_add_f64 %0, v[200:201], %0\n\tv_add_f64 %1, v[202:203], %1\n\tv_add_f64 %2, v[200:201], %2\n\tv_add_f64 %3, v[202:203], %3\n\t"
                   "s_lshr_b32 m0, %[w], 0\n\tv_add_f64 %4, v[200:201], %4\n\tv_add_f64 %5, v[202:203], %5\n\tv_add_f64 %6, v[200:201], %6\n\tv_add_f64 %7, v[202:203], %7\n\t"
                   "s_set_gpr_idx_off\n\t"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), [t] "=&s"(tmp) : [w] "s"(w | 0x1000u), "v"(t0), "v"(t1)
                   : "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", "m0", "scc");
    else
      asm volatile("s_set_gpr_idx_on %[w], 0x1\n\t"
                   "s_bfe_u32 %[t], %[w], 0x20000\n\ts_set_gpr_idx_idx %[t]\n\tv_add_f64 %0, v[200:201], %0\n\tv_add_f64 %1, v[202:203], %1\n\t"
                   "s_bfe_u32 %[t], %[w], 0x20002\n\ts_set_gpr_idx_idx %[t]\n\tv_add_f64 %2, v[200:201], %2\n\tv_add_f64 %3, v[202:203], %3\n\t"
                   "s_bfe_u32 %[t], %[w], 0x20004\n\ts_set_gpr_idx_idx %[t]\n\tv_add_f64 %4, v[200:201], %4\n\tv_add_f64 %5, v[202:203], %5\n\t"
                   "s_bfe_u32 %[t], %[w], 0x20006\n\ts_set_gpr_idx_idx %[t]\n\tv_add_f64 %6, v[200:201], %6\n\tv_add_f64 %7, v[202:203], %7\n\t"
                   "s_set_gpr_idx_off\n\t"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), [t] "=&s"(tmp) : [w] "s"(w), "v"(t0), "v"(t1)
                   : "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", "m0", "scc");
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int MODE>
static double run(int waves_per_simd, const double* in, double* out, int n) {
  // 256 CUs x 4 SIMDs: waves_per_simd workgroups of 256 threads per CU (the registers named in the asm exist for 256-thread workgroups)
  const int threads = 256;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k_time<MODE>, dim3(256 * waves_per_simd), dim3(threads), 0, 0, in, out, 16, 0u);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k_time<MODE>, dim3(256 * waves_per_simd), dim3(threads), 0, 0, in, out, n, 0u);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return (double)ms * 1e-3 * 2.4e9 / ((double)n * 4.0 * waves_per_simd);   // SIMD cycles per group of (index change + 2 adds), per wavefront slot
}

int main() {
  float *in, *out; int* sel; double *din, *dout;
  hipMallocManaged(&in, 256 * 4); hipMallocManaged(&out, 64 * 4); hipMallocManaged(&sel, 4); hipMallocManaged(&din, 256 * 8); hipMallocManaged(&dout, 256 * 1024 * 8);
  for (int i = 0; i < 256; ++i) { in[i] = (float)(i / 64 * 1000 + i % 64); din[i] = (double)(i / 64 * 100 + i % 64); }
  int bad = 0;
  for (int s = 0; s < 4; ++s) {
    sel[0] = s;
    hipLaunchKernelGGL(k_sel32, dim3(1), dim3(64), 0, 0, in, sel, out); hipDeviceSynchronize();
    hipLaunchKernelGGL(k_sel64, dim3(1), dim3(64), 0, 0, din, sel, dout); hipDeviceSynchronize();
    for (int l = 0; l < 64; ++l) { if (out[l] != in[s * 64 + l]) ++bad; if (dout[l] != 1000.0 + din[s * 64 + l]) ++bad; }
  }
  printf("index mode selects the right registers: %s\n", bad ? "NO" : "yes");
  const int n = 200000;
  // round 4: cycles of SIMD time per EIGHT additions (one statement of the loop above) by the number of index changes in it
  for (int w : {1, 2, 3, 4}) {
    const double c0 = run<0>(w, din, dout, n) * 4, c4 = run<4>(w, din, dout, n) * 4, c5 = run<5>(w, din, dout, n) * 4, c6 = run<6>(w, din, dout, n) * 4;
    printf("%d wave(s)/SIMD: SIMD cycles per 8 v_add_f64: no index change %.1f, 2 changes (4 adds each) %.1f, 4 changes (2 adds each, s_lshr_b32 m0) %.1f, 8 changes (1 add each) %.1f\n",
           w, c0, c6, c5, c4);
  }
  for (int w : {1, 2, 4}) {
    const double c0 = run<0>(w, din, dout, n), c1 = run<1>(w, din, dout, n), c2 = run<2>(w, din, dout, n), c3 = run<3>(w, din, dout, n);
    printf("%d wave(s)/SIMD: cycles per group {index change, two v_add_f64} — SIMD throughput / one wavefront's pace: adds alone %.1f / %.1f, with s_bfe + s_set_gpr_idx_on %.1f / %.1f, "
           "with s_bfe + s_set_gpr_idx_idx %.1f / %.1f, with one s_or_b32 into M0 %.1f / %.1f\n", w, c0, c0 * w, c1, c1 * w, c2, c2 * w, c3, c3 * w);
  }
  return bad != 0;
}
