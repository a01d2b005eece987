// Issue cost of single VALU instructions on gfx950, wave64: cycles per instruction per SIMD with 1, 2 and 4 waves per SIMD.
// Build: hipcc -O3 --offload-arch=gfx950 tools/micro/valu_rate.hip -o tools/micro/valu_rate ; run on the GPU box.
// Each kernel is a loop of 64 independent instructions of ONE kind on 16 registers (register r is rewritten every 16
// instructions: latency is hidden from 2 waves per SIMD on); s_memtime around the loop of one wave per workgroup gives cycles.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))

#define KERNEL(NAME, DECL, BODY, SINK)                                                         \
  __global__ __launch_bounds__(256) void NAME(int iters, unsigned long long* cyc, double* out) { \
    DECL;                                                                                      \
    __syncthreads();                                                                           \
    const unsigned long long t0 = __builtin_readcyclecounter();                                \
    for (int i = 0; i < iters; ++i) { BODY }                                                   \
    const unsigned long long t1 = __builtin_readcyclecounter();                                \
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0; \
    SINK;                                                                                      \
  }

#define D4 double a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, b = 1.0000001, c = 0.5
#define U4 unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, b = 0x55aa, c = 3
#define F4 float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, b = 1.0001f, c = 0.5f
#define SINKD out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3
#define SINKU out[blockIdx.x * blockDim.x + threadIdx.x] = (double)(a0 + a1 + a2 + a3)
#define ASM4(ins, cons) asm volatile(ins " %0, %0, %4\n" ins " %1, %1, %4\n" ins " %2, %2, %4\n" ins " %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : cons(b));
#define ASM4_3(ins) asm volatile(ins " %0, %0, %4, %5\n" ins " %1, %1, %4, %5\n" ins " %2, %2, %4, %5\n" ins " %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));

KERNEL(k_add_f64, D4, REP16(ASM4("v_add_f64", "v")), SINKD)
KERNEL(k_mul_f64, D4, REP16(ASM4("v_mul_f64", "v")), SINKD)
KERNEL(k_fma_f64, D4, REP16(ASM4_3("v_fma_f64")), SINKD)
KERNEL(k_max_f64, D4, REP16(ASM4("v_max_f64", "v")), SINKD)
KERNEL(k_add_u32, U4, REP16(ASM4("v_add_u32", "v")), SINKU)
KERNEL(k_and_b32, U4, REP16(ASM4("v_and_b32", "v")), SINKU)
KERNEL(k_lshrrev_b32, U4, REP16(asm volatile("v_lshrrev_b32 %0, 1, %0\nv_lshrrev_b32 %1, 1, %1\nv_lshrrev_b32 %2, 1, %2\nv_lshrrev_b32 %3, 1, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));), SINKU)
KERNEL(k_and_or_b32, U4, REP16(ASM4_3("v_and_or_b32")), SINKU)
KERNEL(k_lshl_add_u32, U4, REP16(asm volatile("v_lshl_add_u32 %0, %0, 1, %4\nv_lshl_add_u32 %1, %1, 1, %4\nv_lshl_add_u32 %2, %2, 1, %4\nv_lshl_add_u32 %3, %3, 1, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));), SINKU)
KERNEL(k_cndmask_b32, U4, REP16(asm volatile("v_cndmask_b32 %0, %0, %4, vcc\nv_cndmask_b32 %1, %1, %4, vcc\nv_cndmask_b32 %2, %2, %4, vcc\nv_cndmask_b32 %3, %3, %4, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "vcc");), SINKU)
KERNEL(k_mov_b32, U4, REP16(asm volatile("v_mov_b32 %0, %4\nv_mov_b32 %1, %4\nv_mov_b32 %2, %4\nv_mov_b32 %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));), SINKU)
KERNEL(k_mov_b64, D4, REP16(asm volatile("v_mov_b64 %0, %4\nv_mov_b64 %1, %4\nv_mov_b64 %2, %4\nv_mov_b64 %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));), SINKD)
KERNEL(k_fma_f32, F4, REP16(ASM4_3("v_fma_f32")), SINKU)
KERNEL(k_add_f32, F4, REP16(ASM4("v_add_f32", "v")), SINKU)
KERNEL(k_cvt_f64_i32, D4; int s0 = threadIdx.x, REP16(asm volatile("v_cvt_f64_i32 %0, %4\nv_cvt_f64_i32 %1, %4\nv_cvt_f64_i32 %2, %4\nv_cvt_f64_i32 %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(s0));), SINKD)
KERNEL(k_cvt_f64_f32, D4; float s0 = threadIdx.x, REP16(asm volatile("v_cvt_f64_f32 %0, %4\nv_cvt_f64_f32 %1, %4\nv_cvt_f64_f32 %2, %4\nv_cvt_f64_f32 %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(s0));), SINKD)
KERNEL(k_cmp_class_f64, D4, REP16(asm volatile("v_cmp_class_f64 vcc, %0, %4\nv_cmp_class_f64 vcc, %1, %4\nv_cmp_class_f64 vcc, %2, %4\nv_cmp_class_f64 vcc, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(0x100) : "vcc");), SINKD)
KERNEL(k_rcp_f64, D4, REP16(asm volatile("v_rcp_f64 %0, %0\nv_rcp_f64 %1, %1\nv_rcp_f64 %2, %2\nv_rcp_f64 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));), SINKD)
KERNEL(k_mov_dpp, U4, REP16(asm volatile("v_mov_b32_dpp %0, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %1, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %2, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %3, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));), SINKU)
KERNEL(k_or_sdwa_b1, U4, REP16(asm volatile("v_or_b32_sdwa %0, %0, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\nv_or_b32_sdwa %1, %1, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\nv_or_b32_sdwa %2, %2, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3\nv_or_b32_sdwa %3, %3, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));), SINKU)
KERNEL(k_or_b32, U4, REP16(ASM4("v_or_b32", "v")), SINKU)
KERNEL(k_bfe_u32, U4, REP16(asm volatile("v_bfe_u32 %0, %0, 3, 8\nv_bfe_u32 %1, %1, 3, 8\nv_bfe_u32 %2, %2, 3, 8\nv_bfe_u32 %3, %3, 3, 8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));), SINKU)
KERNEL(k_alignbit, U4, REP16(ASM4_3("v_alignbit_b32")), SINKU)
KERNEL(k_add3_u32, U4, REP16(ASM4_3("v_add3_u32")), SINKU)
KERNEL(k_lshl_or_b32, U4, REP16(asm volatile("v_lshl_or_b32 %0, %0, 3, %4\nv_lshl_or_b32 %1, %1, 3, %4\nv_lshl_or_b32 %2, %2, 3, %4\nv_lshl_or_b32 %3, %3, 3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));), SINKU)
KERNEL(k_perm_b32, U4, REP16(ASM4_3("v_perm_b32")), SINKU)
KERNEL(k_cndmask_sgpr, U4; unsigned long long m = 0x5555aaaa5555aaaaull ^ blockIdx.x, REP16(asm volatile("v_cndmask_b32 %0, %0, %4, %5\nv_cndmask_b32 %1, %1, %4, %5\nv_cndmask_b32 %2, %2, %4, %5\nv_cndmask_b32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "s"(m));), SINKU)
KERNEL(k_and_b32_e64lit, U4, REP16(asm volatile("v_and_b32 %0, 24, %0\nv_and_b32 %1, 24, %1\nv_and_b32 %2, 24, %2\nv_and_b32 %3, 24, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));), SINKU)
KERNEL(k_mad_u32_u24, U4, REP16(ASM4_3("v_mad_u32_u24")), SINKU)
KERNEL(k_cmp_ne_u32, U4, REP16(asm volatile("v_cmp_ne_u32 vcc, %0, %4\nv_cmp_ne_u32 vcc, %1, %4\nv_cmp_ne_u32 vcc, %2, %4\nv_cmp_ne_u32 vcc, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "vcc");), SINKU)
KERNEL(k_add_f64_dpp_free, D4, REP16(asm volatile("v_add_f64 %0, %0, %4\nv_lshrrev_b32 %5, 1, %5\nv_add_f64 %1, %1, %4\nv_lshrrev_b32 %6, 1, %6\nv_add_f64 %2, %2, %4\nv_lshrrev_b32 %7, 1, %7\nv_add_f64 %3, %3, %4\nv_lshrrev_b32 %8, 1, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(0), "v"(1), "v"(2), "v"(3));), SINKD)
// a mix like the FAST doublet kernels' inner loop: 15 FP64 + 11 other per evaluation -> 4 fma + 3 int per group here
KERNEL(k_mix_4f64_3int, D4; unsigned u0 = threadIdx.x; unsigned u1 = u0 + 1; unsigned u2 = u0 + 2; unsigned ub = 0x33,
       REP16(asm volatile("v_fma_f64 %0, %0, %7, %8\nv_add_u32 %4, %4, %9\nv_fma_f64 %1, %1, %7, %8\nv_and_b32 %5, %5, %9\nv_fma_f64 %2, %2, %7, %8\nv_lshrrev_b32 %6, 1, %6\nv_fma_f64 %3, %3, %7, %8"
                          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(u0), "+v"(u1), "+v"(u2) : "v"(b), "v"(c), "v"(ub));), out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + u0 + u1 + u2)

struct K { const char* name; void (*fn)(int, unsigned long long*, double*); int per_iter; };

int main() {
  const K ks[] = {{"v_add_f64", k_add_f64, 64}, {"v_mul_f64", k_mul_f64, 64}, {"v_fma_f64", k_fma_f64, 64}, {"v_max_f64", k_max_f64, 64},
                  {"v_rcp_f64", k_rcp_f64, 64}, {"v_cvt_f64_i32", k_cvt_f64_i32, 64}, {"v_cvt_f64_f32", k_cvt_f64_f32, 64},
                  {"v_cmp_class_f64", k_cmp_class_f64, 64}, {"v_mov_b64", k_mov_b64, 64},
                  {"v_add_u32", k_add_u32, 64}, {"v_and_b32", k_and_b32, 64}, {"v_lshrrev_b32", k_lshrrev_b32, 64},
                  {"v_and_or_b32", k_and_or_b32, 64}, {"v_lshl_add_u32", k_lshl_add_u32, 64}, {"v_cndmask_b32", k_cndmask_b32, 64},
                  {"v_mov_b32", k_mov_b32, 64}, {"v_mov_b32_dpp", k_mov_dpp, 64}, {"v_fma_f32", k_fma_f32, 64}, {"v_add_f32", k_add_f32, 64},
                  {"v_or_b32_sdwa (byte select)", k_or_sdwa_b1, 64}, {"v_or_b32", k_or_b32, 64}, {"v_bfe_u32", k_bfe_u32, 64}, {"v_alignbit_b32", k_alignbit, 64},
                  {"v_add3_u32", k_add3_u32, 64}, {"v_lshl_or_b32", k_lshl_or_b32, 64}, {"v_perm_b32", k_perm_b32, 64},
                  {"v_cndmask_b32 (sgpr mask)", k_cndmask_sgpr, 64}, {"v_and_b32 (inline const)", k_and_b32_e64lit, 64}, {"v_mad_u32_u24", k_mad_u32_u24, 64},
                  {"v_cmp_ne_u32", k_cmp_ne_u32, 64}, {"pair: v_add_f64 + v_lshrrev_b32", k_add_f64_dpp_free, 128},
                  {"mix 4 v_fma_f64 + 3 int32", k_mix_4f64_3int, 16 * 7}};
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  unsigned long long* cyc; double* out;
  hipMalloc(&cyc, sizeof(unsigned long long) * cus * 16 * 4);
  hipMalloc(&out, sizeof(double) * cus * 16 * 256);
  const int iters = 2000;
  printf("{\"device\": \"%s\", \"cus\": %d, \"unit\": \"shader cycles per wave64 instruction per SIMD (s_memtime around a loop of independent instructions)\", \"rows\": [\n", prop.gcnArchName, cus);
  bool first = true;
  for (const K& k : ks) {
    double res[3], ns[3];
    int wi = 0;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wps : {1, 2, 4}) {                     // waves per SIMD: workgroups of 256 threads = one wave per SIMD each
      const int blocks = cus * wps;
      k.fn<<<blocks, 256>>>(iters, cyc, out);       // warm-up
      hipEventRecord(e0, 0);
      k.fn<<<blocks, 256>>>(iters, cyc, out);
      hipEventRecord(e1, 0);
      hipDeviceSynchronize();
      float ms = 0.f;
      hipEventElapsedTime(&ms, e0, e1);
      ns[wi] = (double)ms * 1e6 / ((double)iters * k.per_iter) / wps;   // wall nanoseconds per instruction per SIMD (incl. launch)
      std::vector<unsigned long long> h(blocks * 4);
      hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost);
      double mean = 0;
      for (auto v : h) mean += (double)v;
      mean /= h.size();
      // s_memtime counts at 100 MHz-independent shader clock on gfx9 (REFCLK on some parts): also report wall via events? keep cycles
      res[wi++] = mean / ((double)iters * k.per_iter) / wps;   // cycles per instruction per SIMD (all resident waves issue the same count)
    }
    printf("%s  {\"inst\": \"%s\", \"cycles_1wave\": %.3f, \"cycles_2waves\": %.3f, \"cycles_4waves\": %.3f, \"wall_ns_4waves\": %.4f}", first ? "" : ",\n", k.name, res[0], res[1], res[2], ns[2]);
    first = false;
  }
  printf("\n]}\n");
  return 0;
}
