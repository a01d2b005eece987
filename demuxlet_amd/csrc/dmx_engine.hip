// Device half of libdmx: the likelihood engine for one MI355X (gfx950).  Hand-written HIP, FP64 VALU + LDS; no MFMA
// (contraction lengths are 3 and 9), no library kernels.  Built with -ffp-contract=off: in STRICT mode every
// expression the reference evaluates is evaluated with the same operations in the same order (separate IEEE mul/add/
// div); the only fused operations are inside dmx_log(), our own replacement for libm's log().
//
//   k_gp0        a4  gp0s[s][l] = (sum_j g[s][j][l]) / V                       cmd_cram_demuxlet.cpp:390-401
//   k_singlet    a5  llks[c][k] += log(GL . g[s][k]),  llk0s[c] += log(GL . gp0s[s])        :412-461
//   k_doublet_*  a8+a9  pG[A][3][3] per covered pair, llksAB[c][j][k][n] += log(sum_lm g_j[l] g_k[m] pG[n][l][m]),
//                    llks00[c][n] likewise with gp0s                                        :576-710
//   k_reduce     a10,a11,a13  per-cell max / posterior sums / top-2 singlets / best doublet :713-734,:746-758,:799-828
//
// Order guarantee (what makes STRICT strict): every accumulator is owned by exactly one lane, which adds that cell's
// per-SNP terms in ascending SNP order — the order of the reference's std::map walks.  Terms themselves are computed
// by whichever lane is free (k_singlet stages them through LDS), which does not change any rounding.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <mutex>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <deque>
#include <new>
#include <memory>
#include <functional>
#include <thread>
#include <type_traits>
#include <unordered_map>
#include <cxxabi.h>
#include <string>
#include <numeric>
#include <vector>

#include "dmx_internal.hpp"
extern char** environ;
#include "dmx_log.hpp"

using dmx::set_error;

#define HIP_TRY(expr)                                                                                          \
  do {                                                                                                         \
    hipError_t _e = (expr);                                                                                    \
    if (_e != hipSuccess) return set_error(DMX_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                                           __FILE__, __LINE__);                                                \
  } while (0)

namespace {

struct PileupView {
  int32_t B, S;
  int64_t R;                   // total read bytes
  const int64_t* cell_pair_off;
  const int64_t* cell_read_off;
  const int32_t* pair_snp;     // nullptr = dense
  const void* pair_nrd;
  const uint8_t* reads;
};

// Ordering of LDS stores and loads between the lanes of ONE wavefront: the LDS executes a wavefront's instructions in
// program order, so only the compiler has to be kept from moving them across this point.
#ifdef DMX_FENCE_VARIANT
#define DMX_WAVE_LDS_ORDER() do { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)
#else
#define DMX_WAVE_LDS_ORDER() do { asm volatile("" ::: "memory"); __builtin_amdgcn_wave_barrier(); } while (0)
#endif

#ifndef DMX_ABLATE
#define DMX_ABLATE 0
#endif

constexpr int kThreads = 256;
constexpr int kLut = 3 * 128;   // mat | err/3 | 0.5-err/3
constexpr int kTab = kLut + DMX_LOG_TABLE_DOUBLES;   // device table buffer: read LUT, then dmx_log's {invc,logc} table
constexpr int kFirst = 256 * 3, kFinal = 257 * 3;     // then the singlet first-read tables (dmx::SingletTables)
constexpr int kTabK1 = kTab + kFirst + kFinal;
constexpr int kPair = 128 * 128 * 4;                    // then dmx::PairTables: second[16384][4] | final2[16384][4] (global memory only)
constexpr int kTriple = dmx::kTripleCodes * dmx::kTripleCodes * dmx::kTripleCodes * 4;   // then dmx::TripleTables: third | final3
constexpr int kTabAll = kTabK1 + 2 * kPair + 2 * kTriple;
constexpr int kTabLogLo = kTabAll;                      // then dmx_log_dd's second-order table (128 doubles)
constexpr int kTabLog2 = kTabAll + 128;                 // then dmx_log2's 256-bin {invc, logc} table (the doublet kernels' log, round 4)
constexpr int kTabLog32 = kTabLog2 + DMX_LOG2_TABLE_DOUBLES;   // then dmx_log2_lite32's split 32-bin table rc[32] | logc[32] (FAST k_doublet_sym, round 6)
constexpr int kTabTotal = kTabLog32 + DMX_LOG32_TABLE_DOUBLES;
// canonical-class log table (k_build_canon_logs): entries in the order of the final GL tables — one read (257, the last = no read) | two | three
constexpr int64_t kCanL2 = 257, kCanL3 = 257 + 128 * 128, kCanN = kCanL3 + (int64_t)dmx::kTripleCodes * dmx::kTripleCodes * dmx::kTripleCodes;
constexpr int kLut2 = 2 * 128;                           // the doublet kernels read mat | err/3 only (the third LUT part is the singlet kernels')
constexpr int kTab2 = kLut2 + DMX_LOG2_TABLE_DOUBLES;   // a doublet kernel's LDS table: read LUT (2 KB) | dmx_log2 table (4 KB) = 1 KB more than
                                                        // rounds 1-3's 3 + 2 KB: k_doublet_a2<64,4> keeps its three workgroups per CU (3 x 53.4 KB)
// stage a doublet kernel's tables: the read LUT from the head of the device buffer, dmx_log2's table from its tail
__device__ __forceinline__ void stage_k2_tables(double* s_tab, const double* __restrict__ tabs, int t, int nthreads) {
  for (int i = t; i < kLut2; i += nthreads) s_tab[i] = tabs[i];
  for (int i = t; i < DMX_LOG2_TABLE_DOUBLES; i += nthreads) s_tab[kLut2 + i] = tabs[kTabLog2 + i];
}

// ---------------------------------------------------------------------------------------------------------------------
__global__ void k_gp0(const float* __restrict__ g, int32_t S, int32_t V, double* __restrict__ gp0) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)S * 3) return;
  const int64_t s = i / 3;
  const int l = (int)(i % 3);
  const float* row = g + (size_t)s * V * 3 + l;
  double acc = 0.0;                                   // calloc'ed, :391
  for (int32_t j = 0; j < V; ++j) acc += (double)row[(size_t)j * 3];   // :393-397 sequential over samples
  gp0[i] = acc / (double)V;                           // :398-400
}

// ---------------------------------------------------------------------------------------------------------------------
// IEEE-754 binary64 division with the reciprocal refinement shared between several numerators of one denominator.
// This is, operation for operation, the sequence the compiler emits for `a / b` (v_rcp_f64, two Newton steps, quotient,
// residual, correction) minus v_div_scale / v_div_fmas-scaling / v_div_fixup, which only act when an exponent is
// extreme; callers guarantee 2^-700 < a,b < 2^700 (otherwise they use the plain `/`).  Correctly rounded, so
// bit-identical to the reference's x86 divsd (checked on the device by dmx_debug_device_div).
__device__ __forceinline__ double rcp_refined(double b) {
  double y = __builtin_amdgcn_rcp(b);
  double e = __builtin_fma(-b, y, 1.0);
  y = __builtin_fma(y, e, y);
  e = __builtin_fma(-b, y, 1.0);
  y = __builtin_fma(y, e, y);
  return y;
}
__device__ __forceinline__ double div_by(double a, double b, double y) {
  const double q = a * y;
  const double r = __builtin_fma(-b, q, a);
  return __builtin_fma(r, y, q);
}

// The plain IEEE division for the (rare) deep pairs, kept out of line: inlined five or nine times it is the register-pressure
// peak of the FAST kernels, whose accumulators have to stay in registers across it.
__device__ __attribute__((noinline)) double div_slow(double a, double b) { return a / b; }

constexpr int kPark = 12;              // doubles of parked k_certify state per barcode: 4 chains | 4 event words | clean[2] | ok | pad
constexpr int kA2MaxV = 1024;            // widest soft-field panel of k_doublet_a2<256,16> (LDS: 32 / 16 / 8 genotype rows of V * 12 bytes + 12 KB)
constexpr int kAnMaxV4 = 368, kAnMaxV8 = 344;   // k_doublet_an's widest panels (its pG block is 9 / 18 KB instead of 4.5): 160 KB of LDS in all
constexpr uint32_t kSafeReads = 15;   // each read scales a likelihood by >= err(127)/3 > 2^-44: 15 reads stay above 2^-700

// ---- genotype likelihoods of a (cell, SNP) pair (cmd_cram_demuxlet.cpp:427-452) -----------------------------------------
// They start from a table entry chosen by the pair's read count and leading read bytes (dmx::SingletTables / PairTables /
// TripleTables, host-computed with the reference's IEEE operations): the whole answer for 0-3 reads, the state after
// three (two, one) reads for deeper pairs, whose loop then starts at read r0.  All tables live in one buffer (`tabs`), so
// a 32-bit element offset selects the entry.  The K1 kernels issue this gather a whole tile ahead of its use.
struct GlSeed { double g0, g1, g2; uint32_t r0; };     // r0: first read the loop still has to apply (>= n: none)
__device__ __forceinline__ GlSeed gl_seed(const double* __restrict__ tabs, uint32_t n, uint32_t rd4) {
  constexpr uint32_t oFirst = kTab, oFinal1 = kTab + kFirst, oSecond = kTabK1, oFinal2 = kTabK1 + kPair,
                     oThird = kTabK1 + 2 * kPair, oFinal3 = kTabK1 + 2 * kPair + kTriple;
  const uint32_t b0 = rd4 & 0xFFu, b1 = (rd4 >> 8) & 0xFFu, b2 = (rd4 >> 16) & 0xFFu;
  const uint32_t q0 = b0 & 127u, q1 = b1 & 127u, q2 = b2 & 127u;
  const bool two = n >= 2 && ((b0 | b1) & 0x40u) == 0;                          // both base qualities < 64: PairTables
  const bool three = n >= 3 && max(max(q0, q1), q2) < (uint32_t)dmx::kTripleBq;  // all three < 48: TripleTables
  const uint32_t i2 = ((((b0 & 0x80u) >> 1) | (b0 & 0x3Fu)) << 7) | (((b1 & 0x80u) >> 1) | (b1 & 0x3Fu));
  const uint32_t c0 = ((b0 & 0x80u) ? (uint32_t)dmx::kTripleBq : 0u) + q0, c1 = ((b1 & 0x80u) ? (uint32_t)dmx::kTripleBq : 0u) + q1,
                 c2 = ((b2 & 0x80u) ? (uint32_t)dmx::kTripleBq : 0u) + q2;
  const uint32_t i3 = __umul24(__umul24(c0, (uint32_t)dmx::kTripleCodes) + c1, (uint32_t)dmx::kTripleCodes) + c2;   // full-rate multiplies
  uint32_t off = oFirst + 3u * b0;
  off = two ? (n == 2 ? oFinal2 : oSecond) + 4u * i2 : off;
  off = three ? (n == 3 ? oFinal3 : oThird) + 4u * i3 : off;
  off = n == 1 ? oFinal1 + 3u * b0 : off;
  off = n == 0 ? oFinal1 + 3u * 256u : off;
  const double* p = tabs + off;
  GlSeed sd;
  sd.g0 = p[0]; sd.g1 = p[1]; sd.g2 = p[2];
  sd.r0 = n < 2 ? n : (three ? 3u : (two ? 2u : 1u));
  return sd;
}
// reads beyond the tables: continue the reference's loop from read r0, then the +1e-6 renormalisation
__device__ __forceinline__ void gl_finish(const GlSeed& sd, uint32_t n, uint32_t rd4, const uint8_t* __restrict__ reads, int64_t off,
                                          const double* __restrict__ lut, double& G0, double& G1, double& G2) {
  G0 = sd.g0; G1 = sd.g1; G2 = sd.g2;
  if (sd.r0 < n) {
    double g0_ = G0, g1_ = G1, g2_ = G2;
    const bool safe = n <= kSafeReads;
    for (uint32_t r = sd.r0; r < n; ++r) {
      const uint32_t byte = (r < 4) ? ((rd4 >> (8 * r)) & 0xFFu) : (uint32_t)reads[off + r];
      const uint32_t bq = byte & 127u;
      const bool alt = (byte >> 7) != 0;
      const double m = lut[bq], e3 = lut[128 + bq], h = lut[256 + bq];
      g0_ *= alt ? e3 : m;                                                   // :437
      g1_ *= h;                                                              // :438
      g2_ *= alt ? m : e3;                                                   // :439
      const double tmp = g0_ + g1_ + g2_;                                    // :440
      if (safe) {
        const double y = rcp_refined(tmp);
        g0_ = div_by(g0_, tmp, y); g1_ = div_by(g1_, tmp, y); g2_ = div_by(g2_, tmp, y);   // :441-443
      } else {
        g0_ /= tmp; g1_ /= tmp; g2_ /= tmp;
      }
    }
    g0_ += 1e-6; g1_ += 1e-6; g2_ += 1e-6;                                   // :446-448
    const double tmp = g0_ + g1_ + g2_;
    const double y = rcp_refined(tmp);
    G0 = div_by(g0_, tmp, y); G1 = div_by(g1_, tmp, y); G2 = div_by(g2_, tmp, y);          // :449-452
  }
}


// the three singlet genotype likelihoods of one covered pair (:427-452): per read multiply, then renormalise to sum 1
__device__ __forceinline__ void pair_gl(const uint8_t* __restrict__ rd, uint32_t n, const double* s_lut, double& G0,
                                        double& G1, double& G2) {
  G0 = 1.0; G1 = 1.0; G2 = 1.0;                                            // :427
  const bool safe = n <= kSafeReads;
  for (uint32_t r = 0; r < n; ++r) {
    const uint32_t byte = rd[r];
    const uint32_t bq = byte & 127u;
    const bool alt = (byte >> 7) != 0;
    const double m = s_lut[bq], e3 = s_lut[128 + bq], h = s_lut[256 + bq];
    G0 *= alt ? e3 : m;                                                    // :437
    G1 *= h;                                                               // :438
    G2 *= alt ? m : e3;                                                    // :439
    const double tmp = G0 + G1 + G2;                                       // :440
    if (safe) {
      const double y = rcp_refined(tmp);
      G0 = div_by(G0, tmp, y); G1 = div_by(G1, tmp, y); G2 = div_by(G2, tmp, y);   // :441-443
    } else {
      G0 /= tmp; G1 /= tmp; G2 /= tmp;
    }
  }
  G0 += 1e-6; G1 += 1e-6; G2 += 1e-6;                                      // :446-448
  const double tmp = G0 + G1 + G2;
  const double y = rcp_refined(tmp);                                       // numerators >= 1e-6, tmp ~ 1
  G0 = div_by(G0, tmp, y); G1 = div_by(G1, tmp, y); G2 = div_by(G2, tmp, y);       // :449-452
}

// Inclusive prefix sum of v over segments of T consecutive lanes (T = 16, 32 or 64), on the VALU's DPP data path:
// four row-shift adds give the scan inside each 16-lane row, row_bcast:15 / row_bcast:31 carry the row totals.
template <int T>
__device__ __forceinline__ uint32_t seg_scan_incl(uint32_t v) {
  static_assert(T == 8 || T == 16 || T == 32 || T == 64, "segment width (8: the segment must start a 16-lane row)");
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);   // row_shr:1 (0 past the row start)
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);   // row_shr:2
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);   // row_shr:4
  if (T >= 16) v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);   // row_shr:8
  if (T >= 32) v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1,3
  if (T >= 64) v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2,3
  return v;
}
// value of the last lane of this lane's T-wide segment
template <int T>
__device__ __forceinline__ uint32_t seg_last(uint32_t v, int lane) {
  if (T == 64) return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
  if (T == 32) {
    const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)v, 31), b = (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
    return lane < 32 ? a : b;
  }
  return (uint32_t)__shfl((int)v, T - 1, T);
}

// The first four stored read bytes of a pair as one (unaligned) 32-bit load; bytes past the pair's reads are never interpreted.
__device__ __forceinline__ uint32_t load_rd4(const PileupView& pv, int64_t off, uint32_t cnt) {
  uint32_t rd4 = 0;
  if (cnt > 0) {
    if (off + 4 <= pv.R) __builtin_memcpy(&rd4, pv.reads + off, 4);
    else for (int64_t i = off; i < pv.R; ++i) rd4 |= (uint32_t)pv.reads[i] << (8 * (int)(i - off));
  }
  return rd4;
}

// The value of the neighbouring lane (lane ^ 1: the other alpha of the same pair in phase 1) by DPP quad permutation [1,0,3,2] —
// two VALU moves instead of the two LDS-crossbar operations (ds_bpermute) that __shfl_xor becomes, in the middle of phase 1's
// dependent chain (products -> maximum -> neighbour's maximum -> reciprocal -> quotients).
// Both lanes of an (even, odd) pair get the ODD lane's value (quad permutation [1,1,3,3]): one DPP move per word, no select.
__device__ __forceinline__ double shfl_odd(double x) {
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), 0xF5, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), 0xF5, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double shfl_xor1(double x) {
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), 0xB1, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), 0xB1, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}

// One 8-byte LDS read that the compiler must not pair with a neighbour: ds_read2_b64 is serviced as two 4 x 16-lane accesses
// (8 LDS cycles, 128 B/clk, MI355X_MICROARCH.md "LDS") where two ds_read_b64 take 2 + 2 — and the hot loops of the FAST doublet
// kernels read three doubles 8*VUS bytes apart per evaluation, which the load/store optimiser merges whenever it can.
__device__ __forceinline__ double lds_read_f64(const double* p) {
  return *(const volatile __attribute__((address_space(3))) double*)(const __attribute__((address_space(3))) double*)p;
}
// base + byte BYTE of w in ONE VALU instruction (SDWA operand select; the compiler emits v_bfe_u32 + v_add for the same expression)
template <int BYTE> __device__ __forceinline__ uint32_t add_byte_of(uint32_t base, uint32_t w) {
  uint32_t r;
  if constexpr (BYTE == 0) asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(r) : "v"(base), "v"(w));
  else if constexpr (BYTE == 1) asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(r) : "v"(base), "v"(w));
  else if constexpr (BYTE == 2) asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(r) : "v"(base), "v"(w));
  else asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(r) : "v"(base), "v"(w));
  return r;
}

// base + 16-bit half HALF of w, likewise (SDWA word select)
template <int HALF> __device__ __forceinline__ uint32_t add_word_of(uint32_t base, uint32_t w) {
  uint32_t r;
  if constexpr (HALF == 0) asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(r) : "v"(base), "v"(w));
  else asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(r) : "v"(base), "v"(w));
  return r;
}
#ifndef DMX_FAST_PRODUCT
#define DMX_FAST_PRODUCT 0                        // 1: FAST k_doublet_sym takes ONE log per entry and full sub-tile, of the product of its terms — an experiment
                                                  // build only: cfg3 FAST K2 217 -> 159 ms, but the sums then round differently from the
                                                  // reference's own sequence of adds and sit up to 1.3e-9 from it at cfg3's depth (rms 2.6e-10; a log per term: 4e-11)
#endif
#ifndef DMX_SYM_ABLATIONS
#define DMX_SYM_ABLATIONS 0                       // 1: k_doublet_sym carries its timing-ablation switches (results WRONG) — experiment builds only
#endif
#ifndef DMX_SYM_NB
#define DMX_SYM_NB 3                              // entries per step of k_doublet_sym's software-pipelined phase 2 (0: off)
#endif
#ifndef DMX_SYM_TIMING
#define DMX_SYM_TIMING 0                          // timing builds only (results WRONG): 1 = phase-2 polynomial one FMA shorter, 2 = k ln 2 from an LDS look-up and one add
#endif                                            //   instead of cvt + fma, 3 = both — what a 2 048-bin table / a k-indexed table would buy
#ifndef DMX_SYM_HYB
#define DMX_SYM_HYB 0                             // k_doublet_sym's mix of the two FAST logs: 0 none, 1 every third entry of a lane through dmx_log2_lite32, 2 two of three, 3 all, 4 every second
#endif
#ifndef DMX_FAST_LITE_LOG
#define DMX_FAST_LITE_LOG 1                       // FAST phase-2 terms through dmx_log2_lite (6 FP64 instructions; csrc/dmx_log.hpp) — 0: dmx_log2 (10), as rounds 2-4
#endif
// the log of FAST mode's phase-2 terms (k_doublet_sym / _a2f / _anf)
__device__ __forceinline__ double dmx_log2_fastmode(double x, const double* __restrict__ T, const DmxLogPins& K) {
  return DMX_FAST_LITE_LOG ? dmx_log2_lite(x, T, K) : dmx_log2_fast_pinned(x, T, K);
}
#ifndef DMX_LDS_NOMERGE
#define DMX_LDS_NOMERGE 1                         // cfg3 FAST K2 309 -> 291 ms (0 restores the merged reads: kernel experiments only)
#endif

__device__ __forceinline__ uint32_t load_nrd(const void* __restrict__ base, int64_t p, int width) {
  if (width == 1) return ((const uint8_t*)base)[p];
  if (width == 2) return ((const uint16_t*)base)[p];
  return ((const uint32_t*)base)[p];
}

// Cells whose fast kernels met a log() argument outside the normal positive range: one byte per cell, and in front of the array
// (kFlagHead bytes) a word that says whether ANY cell of the launch was flagged — what the fix-up pass looks at first.
constexpr int kFlagHead = 16;
__device__ __forceinline__ void flag_cell(uint8_t* __restrict__ flagged, int32_t cell) { flagged[cell] = 1; *(flagged - kFlagHead) = 1; }
__device__ __forceinline__ bool flags_any(const uint8_t* __restrict__ flagged) { return *(const volatile uint8_t*)(flagged - kFlagHead) != 0; }

// Round 6: FINAL phase-1 values of the default grid {0, 0.5} — the five distinct alpha = 0.5 values q[l + m] and the three of alpha = 0 after the read loop AND
// the +1e-6 renormalisation of :649-663 (k_doublet_sym's phase 1, certify_pair_values<5>) —
// for every pair whose reads index a table: no read, one read, two reads of base quality < 64, three of base quality < 48 (K1's TripleTables codes).  One
// thread per entry runs the loop of :597-639 and the finish for BOTH alpha lanes (the maxima run across them): the operations of k_certify's two lanes on
// the same operands, hence the same bits.  A tile none of whose pairs is deeper takes its values from here and skips the seeds, the loop, the finish and
// the hand-over between the lanes (at 1.25 reads per pair: 93 % of the tiles; 43 % with the one- and two-read entries alone).
constexpr int kCFinStride = 8;                   // doubles per entry: alpha 0.5's five values, alpha 0's three (one 64-byte line)
constexpr int64_t kCFin1 = 1, kCFin2 = 1 + 256, kCFin3 = kCFin2 + 128 * 128,
                  kCFinN = kCFin3 + (int64_t)dmx::kTripleCodes * dmx::kTripleCodes * dmx::kTripleCodes;
__device__ __forceinline__ int32_t certify_final_index(uint32_t cnt, uint32_t rd4) {     // -1: the pair's reads are not in the table
  const uint32_t b0 = rd4 & 0xFFu, b1 = (rd4 >> 8) & 0xFFu, b2 = (rd4 >> 16) & 0xFFu;
  const uint32_t q0 = b0 & 127u, q1 = b1 & 127u, q2 = b2 & 127u;
  const uint32_t i2 = ((((b0 & 0x80u) >> 1) | (b0 & 0x3Fu)) << 7) | (((b1 & 0x80u) >> 1) | (b1 & 0x3Fu));
  const uint32_t c0 = ((b0 & 0x80u) ? (uint32_t)dmx::kTripleBq : 0u) + q0, c1 = ((b1 & 0x80u) ? (uint32_t)dmx::kTripleBq : 0u) + q1,
                 c2 = ((b2 & 0x80u) ? (uint32_t)dmx::kTripleBq : 0u) + q2;
  const uint32_t i3 = __umul24(__umul24(c0, (uint32_t)dmx::kTripleCodes) + c1, (uint32_t)dmx::kTripleCodes) + c2;
  int32_t idx = -1;
  idx = (cnt == 3 && max(max(q0, q1), q2) < (uint32_t)dmx::kTripleBq) ? (int32_t)kCFin3 + (int32_t)i3 : idx;
  idx = (cnt == 2 && ((b0 | b1) & 0x40u) == 0) ? (int32_t)kCFin2 + (int32_t)i2 : idx;
  idx = cnt == 1 ? (int32_t)kCFin1 + (int32_t)b0 : idx;
  idx = cnt == 0 ? 0 : idx;
  return idx;
}
__global__ void k_build_certify_finals(const double* __restrict__ tabs, double* __restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= kCFinN) return;
  uint32_t bytes[3] = {0u, 0u, 0u}; int nb;
  if (e < kCFin1) nb = 0;
  else if (e < kCFin2) { bytes[0] = (uint32_t)(e - kCFin1); nb = 1; }
  else if (e < kCFin3) {
    const uint32_t i2 = (uint32_t)(e - kCFin2), c0 = i2 >> 7, c1 = i2 & 127u;
    bytes[0] = ((c0 & 0x40u) << 1) | (c0 & 0x3Fu); bytes[1] = ((c1 & 0x40u) << 1) | (c1 & 0x3Fu); nb = 2;
  } else {
    uint32_t c = (uint32_t)(e - kCFin3);
    const uint32_t c2 = c % (uint32_t)dmx::kTripleCodes; c /= (uint32_t)dmx::kTripleCodes;
    const uint32_t c1 = c % (uint32_t)dmx::kTripleCodes, c0 = c / (uint32_t)dmx::kTripleCodes;
    const uint32_t cs[3] = {c0, c1, c2};
    for (int r = 0; r < 3; ++r) bytes[r] = cs[r] >= (uint32_t)dmx::kTripleBq ? (0x80u | (cs[r] - (uint32_t)dmx::kTripleBq)) : cs[r];
    nb = 3;
  }
  double wA[2][5], wR[2][5], pG[2][5];
  for (int n1 = 0; n1 < 2; ++n1)
    for (int q = 0; q < 5; ++q) {                  // the weights of k_certify's FIVE form
      const int l = n1 ? (q > 2 ? 2 : q) : min(q, 2), m = n1 ? q - l : 0;
      const double p = 0.5 * l + (m - l) * 0.5 * (n1 ? 0.5 : 0.0);
      wA[n1][q] = p; wR[n1][q] = 1.0 - p; pG[n1][q] = 1.0;
    }
  for (int r = 0; r < nb; ++r) {
    const uint32_t byte = bytes[r], bq = byte & 127u;
    const bool alt = (byte >> 7) != 0;
    const double pR = alt ? tabs[128 + bq] : tabs[bq];
    const double pA = alt ? tabs[bq] : tabs[128 + bq];
    double mx[2] = {0.0, 0.0};
    for (int n1 = 0; n1 < 2; ++n1)
      for (int i = 0; i < 5; ++i) {
        pG[n1][i] *= (pR * wR[n1][i] + pA * wA[n1][i]);
        mx[n1] = fmax(mx[n1], pG[n1][i]);
      }
    const double m = fmax(mx[0], mx[1]);
    const double y = rcp_refined(m);
    for (int n1 = 0; n1 < 2; ++n1)
      for (int i = 0; i < 5; ++i) pG[n1][i] = div_by(pG[n1][i], m, y);
  }
  double mx[2] = {0.0, 0.0};
  for (int n1 = 0; n1 < 2; ++n1)
    for (int i = 0; i < 5; ++i) {
      pG[n1][i] += 1e-6;                                                     // :649
      mx[n1] = fmax(mx[n1], pG[n1][i]);
    }
  const double m = fmax(mx[0], mx[1]);
  const double y = rcp_refined(m);
  double* o = out + (size_t)e * kCFinStride;
  for (int i = 0; i < 5; ++i) o[i] = div_by(pG[1][i], m, y);                 // :656-663, the alpha = 0.5 lane's values (what both lanes of a k_certify pair use)
  for (int i = 0; i < 3; ++i) o[5 + i] = div_by(pG[0][i], m, y);             // ... and the alpha = 0 lane's three distinct ones (k_doublet_sym's phase 1)
}

// Phase 1 of k_certify for one (pair, alpha) lane: the pG values of :597-663 with the one-max-across-both-alphas renormalisation
// after every read, finished (:656-663) and handed from the alpha = 0.5 lane to its alpha = 0 neighbour.  NV = 9: the nine
// values pG[l][m]; NV = 5 (alpha[0] == 0): the five distinct values of alpha 0.5 (weight p = (l + m) / 4, index l + m) beside the
// three of alpha 0 (p = l / 2) — entries with equal weights go through identical operations, so the values are bit-identical.
// seeds (NV = 5, the default grid; round 4): the lane's five values after the pair's first read — or first two, both of base quality < 64 — come from a
// table built on the device with this very loop (k_build_certify_seeds: [256 + 16 384 read codes][alpha lane][6]), and the loop starts at read 1 or 2:
// at 1.25 reads per pair it used to run max(cnt) ~ 2.6 times per tile of 32 pairs with most lanes idle, now 0.6 times.
constexpr int kCSeedStride = 6;                  // doubles per (entry, alpha lane): five values + pad (16-byte aligned loads)
constexpr int64_t kCSeedN = 256 + 128 * 128;
template <int NV, bool HAND = true>            // HAND: k_certify's hand-over at the end (both lanes of a pair leave with the alpha = 0.5 lane's values); false: each lane keeps its own
__device__ __forceinline__ void certify_pair_values(const PileupView& pv, uint32_t cnt, int64_t off, uint32_t rd4, const double* s_tab,
                                                    const double (&wA)[NV], const double (&wR)[NV], int n1, double (&v)[NV],
                                                    const double* __restrict__ cseed = nullptr) {
  double pG[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) pG[i] = 1.0;                               // :597
  uint32_t r_start = 0;
  if (NV == 5 && cseed) {
    if (cnt >= 1 && cnt <= kSafeReads) {
      const uint32_t b0 = rd4 & 0xFFu, b1 = (rd4 >> 8) & 0xFFu;
      const bool two = cnt >= 2 && ((b0 | b1) & 0x40u) == 0;
      const uint32_t i2 = ((((b0 & 0x80u) >> 1) | (b0 & 0x3Fu)) << 7) | (((b1 & 0x80u) >> 1) | (b1 & 0x3Fu));
      const double* sp = cseed + ((size_t)(two ? 256u + i2 : b0) * 2 + (size_t)n1) * kCSeedStride;
      const double2 a = *reinterpret_cast<const double2*>(sp), b = *reinterpret_cast<const double2*>(sp + 2);
      pG[0] = a.x; pG[1] = a.y; pG[2] = b.x; pG[3] = b.y; pG[NV - 1] = sp[4];
      r_start = two ? 2u : 1u;
    }
  }
  // first read any lane still has to apply (wave-uniform)
  uint32_t r = 0;
  if (NV == 5 && cseed) r = __any(r_start == 0 && cnt > 0) ? 0u : (__any(r_start <= 1 && cnt > 1) ? 1u : 2u);
  for (; __any(r < cnt); ++r) {
    const bool live = r >= r_start && r < cnt;
    const uint32_t byte = live ? (r < 4 ? (rd4 >> (8 * r)) & 0xFFu : (uint32_t)pv.reads[off + r]) : 0u;
    const uint32_t bq = byte & 127u;
    const bool alt = (byte >> 7) != 0;
    const double pR = alt ? s_tab[128 + bq] : s_tab[bq];                // :606
    const double pA = alt ? s_tab[bq] : s_tab[128 + bq];                // :607
    double mx = 0.0;
    if (live) {
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        pG[i] *= (pR * wR[i] + pA * wA[i]);                             // :625
        mx = fmax(mx, pG[i]);                                 // :626-627
      }
    }
    {
      const double o = shfl_xor1(mx);                               // one max across both alphas of the pair
      mx = fmax(mx, o);
    }
    if (live) {
      if (cnt <= kSafeReads) {
        const double y = rcp_refined(mx);
#pragma unroll
        for (int i = 0; i < NV; ++i) pG[i] = div_by(pG[i], mx, y);      // :632-639
      } else {
#pragma unroll
        for (int i = 0; i < NV; ++i) pG[i] = div_slow(pG[i], mx);
      }
    }
  }
  double mx = 0.0;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    pG[i] += 1e-6;                                                       // :649
    mx = fmax(mx, pG[i]);
  }
  {
    const double o = shfl_xor1(mx);
    mx = fmax(mx, o);
  }
  // the alpha = 0.5 lane finishes its values (:656-663) and hands a copy to its alpha = 0 neighbour (which only had to contribute
  // to the shared maxima): the two lanes of a pair then take one accumulator each
  const double y = rcp_refined(mx);
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = div_by(pG[i], mx, y);
  if (HAND) {
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = shfl_odd(v[i]);   // (the alpha = 0.5 lane is the odd one: n1 == lane & 1)
  }
  (void)n1;
}

// The seeds of certify_pair_values<5>: one thread per read code (256 one-read codes, then 128 x 128 two-read codes of base quality < 64) runs the
// loop of :597-639 for BOTH alpha lanes (0 and 0.5: the maximum of :626-627 runs across them) — the same operations on the same operands as the
// two lanes of k_certify, hence the same bits.
__global__ void k_build_certify_seeds(const double* __restrict__ tabs, double* __restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= kCSeedN) return;
  uint32_t bytes[2]; int nb;
  if (e < 256) { bytes[0] = (uint32_t)e; bytes[1] = 0; nb = 1; }
  else {
    const uint32_t i2 = (uint32_t)(e - 256), c0 = i2 >> 7, c1 = i2 & 127u;
    bytes[0] = ((c0 & 0x40u) << 1) | (c0 & 0x3Fu); bytes[1] = ((c1 & 0x40u) << 1) | (c1 & 0x3Fu); nb = 2;
  }
  double wA[2][5], wR[2][5], pG[2][5];
  for (int n1 = 0; n1 < 2; ++n1)
    for (int q = 0; q < 5; ++q) {                  // the weights of k_certify's FIVE form
      const int l = n1 ? (q > 2 ? 2 : q) : min(q, 2), m = n1 ? q - l : 0;
      const double p = 0.5 * l + (m - l) * 0.5 * (n1 ? 0.5 : 0.0);
      wA[n1][q] = p; wR[n1][q] = 1.0 - p; pG[n1][q] = 1.0;
    }
  for (int r = 0; r < nb; ++r) {
    const uint32_t byte = bytes[r], bq = byte & 127u;
    const bool alt = (byte >> 7) != 0;
    const double pR = alt ? tabs[128 + bq] : tabs[bq];
    const double pA = alt ? tabs[bq] : tabs[128 + bq];
    double mx[2] = {0.0, 0.0};
    for (int n1 = 0; n1 < 2; ++n1)
      for (int i = 0; i < 5; ++i) {
        pG[n1][i] *= (pR * wR[n1][i] + pA * wA[n1][i]);
        mx[n1] = fmax(mx[n1], pG[n1][i]);
      }
    const double m = fmax(mx[0], mx[1]);
    const double y = rcp_refined(m);
    for (int n1 = 0; n1 < 2; ++n1)
      for (int i = 0; i < 5; ++i) pG[n1][i] = div_by(pG[n1][i], m, y);
  }
  for (int n1 = 0; n1 < 2; ++n1) {
    double* o = out + ((size_t)e * 2 + n1) * kCSeedStride;
    for (int i = 0; i < 5; ++i) o[i] = pG[n1][i];
    o[5] = 0.0;
  }
}



// K1.  Wavefronts are independent (no workgroup barrier in the loop).  A wavefront owns CW cells for their whole SNP
// range and walks them tile by tile, T = 64/CW SNP-pairs per cell per tile:
//   compute  lane (cell c = lane/T, pair ti = lane%T): GL of its pair once per tile, then per chunk of KC samples the KC
//            log terms (plus, in chunk 0, the term of the average-genotype model llk0) -> LDS, chain-major
//            ([cell][slot][ti]);
//   sum      lane a < CW*(KC+1) owns accumulator (cell a/(KC+1), slot a%(KC+1)): adds the tile's T terms in ascending
//            pair order (16-byte LDS reads, two terms each).
// Genotype probabilities stay float32 in memory (they ARE float32 values, bcf_filtered_reader.h:78) and are widened in
// registers.  Dense pileups (every cell covers every SNP, pair_snp == NULL) read them SNP-minor (gT[row element][snp],
// g0T[l][snp]) so a cell's T lanes read contiguous addresses; sparse pileups read the row of the lane's own SNP from the
// SNP-major originals (g[snp][k][l], gp0s[snp][l]).
// A two-deep software pipeline keeps global-memory latency off the dependent path: the pair header (read count, SNP id)
// of tile+2 is in flight while the read offsets of tile+1 are scanned and its leading read bytes requested, while tile
// is computed.
template <int CW, int KC, bool DENSE>
__global__ __launch_bounds__(kThreads, 4) void k_singlet(PileupView pv, int nrd_width, const float* __restrict__ gq,
                                                         const double* __restrict__ g0q, const double* __restrict__ tabs,
                                                         const int32_t* __restrict__ sched, int32_t V, int32_t QS,
                                                         double* __restrict__ llks, double* __restrict__ llk0s,
                                                         const int64_t* __restrict__ blk, int32_t blk_i, int32_t nblk) {
  // blk != nullptr: this launch covers SNP block blk_i of nblk only (sparse pileups whose genotype matrix does not fit the L2:
  // launch_singlet walks the SNP axis block by block so that the rows a launch gathers stay L2-resident).  blk[cell][b] = {first
  // pair, first read byte} of the cell at the start of block b (k_snp_blocks); the running sums are parked in llks / llk0s
  // between launches: same lanes, same order of additions, same bits as the one-launch walk.
  constexpr int ablate = DMX_ABLATE;             // profiling builds only (tools/build_variant.sh); 0 in the product
  constexpr int T = 64 / CW;
  constexpr int TS = T + 2;                      // row stride of a chain in LDS (doubles): keeps 16-byte alignment, and
                                                 // 2*TS mod 64 == 4 dwords spreads the chains of a wavefront over the banks
  constexpr int NC = KC + 1;                     // chains per cell and chunk: KC samples + llk0 (chunk 0 only)
  constexpr int NW = kThreads / 64;
  static_assert(CW * NC <= 64, "one lane per chain");
  extern __shared__ double s_dyn[];              // [NW][nq][CW*NC] running accumulators of this workgroup's sample chunks
  __shared__ double s_tab[kTabK1];
  const double* s_first = s_tab + kTab;
  __shared__ __attribute__((aligned(16))) double s_term[NW][CW * NC * TS];
  const double* s_log = s_tab + kLut;
  const double* s_final = s_first + kFirst;

  const int t = threadIdx.x, w = t >> 6, lane = t & 63;
  const int nch = (V + KC - 1) / KC;
  // panels with more than QS chunks of KC samples are cut into sample slabs, one workgroup (blockIdx.y) per slab; each
  // slab recomputes the (cheap) genotype likelihoods of its cells' pairs
  const int q_lo = (int)blockIdx.y * QS, q_hi = min(nch, q_lo + QS), nq = q_hi - q_lo;
  for (int i = t; i < kTabK1; i += kThreads) s_tab[i] = tabs[i];
  for (int i = t; i < NW * nq * CW * NC; i += kThreads) s_dyn[i] = 0.0;
  __syncthreads();                               // the only workgroup barrier

  double* term = s_term[w];
  double* accs = s_dyn + (size_t)w * nq * CW * NC;
  const int slot0 = (blockIdx.x * NW + w) * CW;  // first of this wavefront's cells in launch order
  if (slot0 >= pv.B) return;

  // compute-phase identity
  const int c = lane / T, ti = lane % T;
  const bool cell_ok = slot0 + c < pv.B;
  const int32_t cell = cell_ok ? sched[slot0 + c] : 0;
  const int64_t* bt = blk ? blk + ((size_t)cell * (nblk + 1) + blk_i) * 2 : nullptr;
  const int64_t p_beg = cell_ok ? (blk ? bt[0] : pv.cell_pair_off[cell]) : 0;
  const int64_t np = cell_ok ? (blk ? bt[2] : pv.cell_pair_off[cell + 1]) - p_beg : 0;
  int64_t rd_base = cell_ok ? (blk ? bt[1] : pv.cell_read_off[cell]) : 0;
  int64_t max_np = np;
#pragma unroll
  for (int d = T; d < 64; d <<= 1) max_np = max(max_np, __shfl_xor(max_np, d));
  // sum-phase identity
  const int a_c = lane / NC, a_kk = lane % NC;
  const bool a_ok = lane < CW * NC && slot0 + a_c < pv.B;
  const int64_t a_np = __shfl(np, (a_ok ? a_c : 0) * T);
  const int32_t a_cell = __shfl(cell, (a_ok ? a_c : 0) * T);
  const size_t S = (size_t)pv.S;
  if (blk && blk_i > 0 && a_ok) {                // resume: the sums of the blocks before this one
    for (int q = q_lo; q < q_hi; ++q) {
      double v = 0.0;
      if (a_kk < KC) { const int k = q * KC + a_kk; if (k < V) v = llks[(size_t)a_cell * V + k]; }
      else if (q == 0) v = llk0s[a_cell];
      accs[(q - q_lo) * (CW * NC) + lane] = v;
    }
  }

  // Dense genotype planes are read through buffer descriptors: address = descriptor base + scalar plane offset + lane
  // offset, one instruction per element and no per-lane 64-bit address arithmetic (S*V*12 < 4 GiB is checked at launch).
  const uint32_t plane4 = (uint32_t)S * 4u, plane8 = (uint32_t)S * 8u;
  const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc((void*)gq, 0, DENSE ? (int)min((size_t)0x7FFFFFFF, S * (size_t)V * 12) : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_g0 = __builtin_amdgcn_make_buffer_rsrc((void*)g0q, 0, DENSE ? (int)min((size_t)0x7FFFFFFF, S * (size_t)24) : 0, 0x00020000);

  struct Raw { uint32_t n; int32_t snp; };
  struct Hdr { uint32_t n; int32_t snp; uint32_t rd4; int64_t off; };
  auto issue = [&](int64_t tile) {               // loads only
    Raw r;
    const int64_t pi = tile * T + ti;
    const bool v = pi < np;
    r.n = v ? load_nrd(pv.pair_nrd, p_beg + pi, nrd_width) : 0u;
    r.snp = v ? (DENSE ? (int32_t)pi : pv.pair_snp[p_beg + pi]) : 0;
    return r;
  };
  auto prepare = [&](const Raw& r) {             // prefix-scan the read counts of the cell's T pairs, request the bytes
    Hdr h;
    h.n = r.n; h.snp = r.snp;
    const uint32_t incl = seg_scan_incl<T>(r.n);
    h.off = rd_base + (int64_t)(incl - r.n);
    rd_base += seg_last<T>(incl, lane);
    h.rd4 = 0;
    if (r.n > 0) {
      if (h.off + 4 <= pv.R) __builtin_memcpy(&h.rd4, pv.reads + h.off, 4);   // bytes past the pair's reads are never interpreted
      else for (int64_t i = h.off; i < pv.R; ++i) h.rd4 |= (uint32_t)pv.reads[i] << (8 * (int)(i - h.off));
    }
    return h;
  };

  Hdr nxt = prepare(issue(0));
  Raw pre = issue(1);
  for (int64_t tile = 0; tile * T < max_np; ++tile) {
    Hdr cur = nxt;
    if (!(ablate & 16)) {
      nxt = prepare(pre);                        // tile+1: its counts arrived a tile ago
      pre = issue(tile + 2);                     // tile+2: loads in flight
    } else { cur.n = 1; cur.rd4 = 0x9e; cur.snp = 0; }
    const bool valid = tile * T + ti < np;

    // ---- genotype likelihoods of this lane's pair (:427-452).  Pairs with at most one read (the bulk of single-cell
    // data) are a table lookup on the read byte; deeper pairs continue the reference's per-read loop from the table's
    // state after the first read.
    double G0, G1, G2;
    {
      const uint32_t n = cur.n;
      const uint32_t b0 = cur.rd4 & 0xFFu;
      const double* f = s_final + 3 * (n ? b0 : 256u);
      G0 = f[0]; G1 = f[1]; G2 = f[2];
      if (n >= 2 && !(ablate & 1)) {
        const double* f1 = s_first + 3 * b0;
        double g0 = f1[0], g1 = f1[1], g2 = f1[2];
        const bool safe = n <= kSafeReads;
        for (uint32_t r = 1; r < n; ++r) {
          const uint32_t byte = (r < 4) ? ((cur.rd4 >> (8 * r)) & 0xFFu) : (uint32_t)pv.reads[cur.off + r];
          const uint32_t bq = byte & 127u;
          const bool alt = (byte >> 7) != 0;
          const double m = s_tab[bq], e3 = s_tab[128 + bq], h = s_tab[256 + bq];
          g0 *= alt ? e3 : m;                                                // :437
          g1 *= h;                                                           // :438
          g2 *= alt ? m : e3;                                                // :439
          const double tmp = g0 + g1 + g2;                                   // :440
          if (safe) {
            const double y = rcp_refined(tmp);
            g0 = div_by(g0, tmp, y); g1 = div_by(g1, tmp, y); g2 = div_by(g2, tmp, y);   // :441-443
          } else {
            g0 /= tmp; g1 /= tmp; g2 /= tmp;
          }
        }
        g0 += 1e-6; g1 += 1e-6; g2 += 1e-6;                                  // :446-448
        const double tmp = g0 + g1 + g2;
        const double y = rcp_refined(tmp);
        G0 = div_by(g0, tmp, y); G1 = div_by(g1, tmp, y); G2 = div_by(g2, tmp, y);       // :449-452
      }
    }
    // dense: uniform plane base per row element + this lane's SNP index (scalar base, 32-bit lane offset, no per-load
    // address arithmetic); sparse: this lane's own row
    const uint32_t s_idx = (uint32_t)min((int64_t)(tile * T + ti), (int64_t)S - 1);
    const uint32_t boff4 = s_idx * 4u, boff8 = s_idx * 8u;     // lane byte offsets for the buffer loads below
    const float* __restrict__ grow = gq + (size_t)cur.snp * V * 3;
    const double* __restrict__ g0row = g0q + (size_t)cur.snp * 3;

    for (int q = q_lo; q < q_hi; ++q) {
      const int k0 = q * KC;
      float a[KC][3];
      double a0[3];
#pragma unroll
      for (int kk = 0; kk < KC; ++kk) {          // all loads of the chunk first; slots past sample V-1 re-read it (ignored)
        const int k = min(k0 + kk, V - 1);
#pragma unroll
        for (int l = 0; l < 3; ++l)
          a[kk][l] = (ablate & 8) ? 0.3f + 0.01f * kk
                     : (DENSE ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_g, boff4, (uint32_t)(k * 3 + l) * plane4, 0))
                              : grow[k * 3 + l]);
      }
      if (q == 0) {
#pragma unroll
        for (int l = 0; l < 3; ++l)
          a0[l] = (ablate & 8) ? 0.33
                  : (DENSE ? __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs_g0, boff8, (uint32_t)l * plane8, 0)) : g0row[l]);
      }
      if (valid) {
        bool fast_ok = true;
#pragma unroll
        for (int kk = 0; kk < KC; ++kk) {
          const double x = G0 * (double)a[kk][0] + G1 * (double)a[kk][1] + G2 * (double)a[kk][2];   // :456
          fast_ok &= __builtin_amdgcn_class(x, 0x100);                       // +normal: the fast path's domain
          if (!(ablate & 32) || kk == 0) term[(c * NC + kk) * TS + ti] = (ablate & 2) ? x : dmx_log_fast(x, s_log); else G0 += x;
        }
        if (q == 0) {
          const double x = G0 * a0[0] + G1 * a0[1] + G2 * a0[2];             // :459
          fast_ok &= __builtin_amdgcn_class(x, 0x100);
          term[(c * NC + KC) * TS + ti] = (ablate & 2) ? x : dmx_log_fast(x, s_log);
        }
        if (__builtin_expect(!fast_ok, 0)) {       // never for real likelihoods; keeps log(0) / log(nan) semantics
          for (int kk = 0; kk <= KC; ++kk) {       // one rolled copy of ocml's log; operands re-read from memory
            if (kk == KC && q != 0) break;
            double b[3];
            if (kk < KC) {
              const int k = min(k0 + kk, V - 1);
              for (int l = 0; l < 3; ++l) b[l] = (double)(DENSE ? (gq + (size_t)(k * 3 + l) * S)[s_idx] : grow[k * 3 + l]);
            } else {
              for (int l = 0; l < 3; ++l) b[l] = DENSE ? (g0q + (size_t)l * S)[s_idx] : g0row[l];
            }
            const double x = G0 * b[0] + G1 * b[1] + G2 * b[2];
            if (!__builtin_amdgcn_class(x, 0x100)) term[(c * NC + kk) * TS + ti] = log(x);
          }
        }
      }
      // ---- ordered sums of this (tile, chunk).  Wavefront-local: the LDS serves one wavefront's accesses in program
      // order; the compiler barrier only stops the compiler from moving the reads above the stores (and the next stores above the reads).
      DMX_WAVE_LDS_ORDER();
      if (a_ok && (a_kk < KC ? k0 + a_kk < V : q == 0)) {
        const int64_t left = a_np - tile * T;
        const int cnt = (ablate & 4) ? 1 : (left >= T ? T : (left > 0 ? (int)left : 0));
        const double* row = &term[lane * TS];
        double s = accs[(q - q_lo) * (CW * NC) + lane];
        int i = 0;
        for (; i + 16 <= cnt; i += 16) {           // loads first (latency paid once), then the ordered adds
          double2 v[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const double2*>(&row[i + 2 * j]);
#pragma unroll
          for (int j = 0; j < 8; ++j) { s += v[j].x; s += v[j].y; }
        }
        for (; i < cnt; ++i) s += row[i];           // ascending SNP order: the reference's order
        accs[(q - q_lo) * (CW * NC) + lane] = s;
      }
      DMX_WAVE_LDS_ORDER();
    }
  }
  if (a_ok) {
    for (int q = q_lo; q < q_hi; ++q) {
      const double s = accs[(q - q_lo) * (CW * NC) + lane];
      if (a_kk < KC) { const int k = q * KC + a_kk; if (k < V) llks[(size_t)a_cell * V + k] = s; }
      else if (q == 0) llk0s[a_cell] = s;
    }
  }
}

#ifndef DMX_K1O_U
#define DMX_K1O_U 4
#endif
// the C library's log() for an argument outside dmx_log's domain (never for real likelihoods), out of line: one copy per kernel, not per term
__device__ __attribute__((noinline)) double log_slow(double x) { return log(x); }

// K1 with OWNED accumulators (round 5; soft fields, up to 64 samples).  k_singlet's lanes are pairs: they park V + 1 log terms per pair in the
// LDS, and nine lanes of the wavefront then add them up chunk by chunk — a serial 64-add chain per chunk of eight samples with 9 of 64 lanes
// active, a third of that kernel's issue cycles, on top of an LDS store and load per term.  Here a sample's running sum lives in a register of
// ONE lane for the whole walk (k_doublet_*'s ownership):
//   phase 1  lane (cell c = lane / TPC, pair ti = lane % TPC): GL of its pair (:427-452) and the llk0 term (:459) -> LDS;
//   phase 2  lane (cell c, sample j = lane % TPC): for the tile's pairs in ascending order: the pair's GL (a broadcast LDS read), its own three
//            genotype probabilities at that SNP (SNP-major rows g[snp][k][l]: the cell's lanes read one contiguous row), the dot product, the log, the add.
// Same operands, same operations, same order of additions as k_singlet: the same bits (tests/test_gpu_parity.py::test_k1_owned_sums_are_k_singlets_bits).
// CHK = false (k_check_geno passed the matrix): GL >= 1e-6 / (1 + 3e-6) and a row's maximum >= 1e-30 make every log argument a normal positive
// number, and the per-term class test is dropped (k_doublet_a2's CHK).
template <int TPC, bool DENSE, bool CHK>
__global__ __launch_bounds__(kThreads, 5) void k_singlet_own(PileupView pv, int nrd_width, const float* __restrict__ g,
                                                             const double* __restrict__ gp0, const double* __restrict__ tabs,
                                                             const int32_t* __restrict__ sched, int32_t V,
                                                             double* __restrict__ llks, double* __restrict__ llk0s,
                                                             const int64_t* __restrict__ blk, int32_t blk_i, int32_t nblk) {
  static_assert(TPC == 16 || TPC == 32 || TPC == 64, "lanes per cell");
  constexpr int CW = 64 / TPC, T = TPC;          // cells per wavefront; pairs per cell and tile
  constexpr int TS = T + 2;                      // stride of a cell's llk0 terms (doubles): 16-byte aligned rows
  constexpr int NW = kThreads / 64;
  constexpr int U = DMX_K1O_U;                   // pairs per batch of phase 2: their rows are requested together, a batch ahead of their use
  static_assert(T % U == 0, "batches");
  __shared__ double s_tab[kTabK1];
  __shared__ __attribute__((aligned(16))) double s_gl[NW][64 * 4];     // [pair lane][G0 G1 G2 -]
  __shared__ __attribute__((aligned(16))) double s_t0[NW][CW * TS];    // [cell][pair] llk0 terms
  __shared__ uint32_t s_row[NW][64];                                   // byte offset of the pair's genotype row
  const double* s_first = s_tab + kTab;
  const double* s_log = s_tab + kLut;
  const double* s_final = s_first + kFirst;
  const int t = threadIdx.x, w = t >> 6, lane = t & 63;
  for (int i = t; i < kTabK1; i += kThreads) s_tab[i] = tabs[i];
  __syncthreads();                               // the only workgroup barrier
  const int slot0 = (blockIdx.x * NW + w) * CW;
  if (slot0 >= pv.B) return;
  double* gl = s_gl[w];
  double* t0s = s_t0[w];
  uint32_t* rowo = s_row[w];

  const int c = lane / T, ti = lane % T;         // phase 1: pair ti of cell c; phase 2: sample ti of cell c
  const bool cell_ok = slot0 + c < pv.B;
  const int32_t cell = cell_ok ? sched[slot0 + c] : 0;
  const int64_t* bt = blk ? blk + ((size_t)cell * (nblk + 1) + blk_i) * 2 : nullptr;     // (k_singlet's SNP-blocked walk)
  const int64_t p_beg = cell_ok ? (blk ? bt[0] : pv.cell_pair_off[cell]) : 0;
  const int64_t np = cell_ok ? (blk ? bt[2] : pv.cell_pair_off[cell + 1]) - p_beg : 0;
  int64_t rd_base = cell_ok ? (blk ? bt[1] : pv.cell_read_off[cell]) : 0;
  int64_t max_np = np;
#pragma unroll
  for (int d = T; d < 64; d <<= 1) max_np = max(max_np, __shfl_xor(max_np, d));
  const DmxLogPins lk = dmx_log_pins();          // the log's addend constants in registers (same operations, same bits: csrc/dmx_log.hpp)
  const bool own = cell_ok && ti < V;
  const bool own0 = cell_ok && ti == 0;          // the cell's first lane also owns llk0
  double acc = 0.0, acc0 = 0.0;
  if (blk && blk_i > 0) {                        // resume: the sums of the blocks before this one
    if (own) acc = llks[(size_t)cell * V + ti];
    if (own0) acc0 = llk0s[cell];
  }
  const uint32_t row_bytes = (uint32_t)V * 12u;
  const uint32_t my_off = (uint32_t)min(ti, V - 1) * 12u;          // lanes past the panel re-read its last sample (ignored)
  const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc((void*)g, 0, (int)min((size_t)0x7FFFFFFF, (size_t)pv.S * V * 12), 0x00020000);

  struct Raw { uint32_t n; int32_t snp; };
  struct Hdr { uint32_t n; int32_t snp; uint32_t rd4; int64_t off; };
  auto issue = [&](int64_t tile) {               // loads only
    Raw r;
    const int64_t pi = tile * T + ti;
    const bool v = pi < np;
    r.n = v ? load_nrd(pv.pair_nrd, p_beg + pi, nrd_width) : 0u;
    r.snp = v ? (DENSE ? (int32_t)pi : pv.pair_snp[p_beg + pi]) : 0;
    return r;
  };
  auto prepare = [&](const Raw& r) {             // prefix-scan the read counts of the cell's T pairs, request the bytes
    Hdr h;
    h.n = r.n; h.snp = r.snp;
    const uint32_t incl = seg_scan_incl<T>(r.n);
    h.off = rd_base + (int64_t)(incl - r.n);
    rd_base += seg_last<T>(incl, lane);
    h.rd4 = load_rd4(pv, h.off, r.n);
    return h;
  };
  Hdr nxt = prepare(issue(0));
  Raw pre = issue(1);
  for (int64_t tile = 0; tile * T < max_np; ++tile) {
    const Hdr cur = nxt;
    nxt = prepare(pre);                          // tile+1: its counts arrived a tile ago
    pre = issue(tile + 2);                       // tile+2: loads in flight
    // ---- phase 1: genotype likelihoods of this lane's pair (:427-452), exactly k_singlet's
    double G0, G1, G2;
    {
      const uint32_t n = cur.n;
      const uint32_t b0 = cur.rd4 & 0xFFu;
      const double* f = s_final + 3 * (n ? b0 : 256u);
      G0 = f[0]; G1 = f[1]; G2 = f[2];
      if (n >= 2) {
        const double* f1 = s_first + 3 * b0;
        double g0 = f1[0], g1 = f1[1], g2 = f1[2];
        const bool safe = n <= kSafeReads;
        for (uint32_t r = 1; r < n; ++r) {
          const uint32_t byte = (r < 4) ? ((cur.rd4 >> (8 * r)) & 0xFFu) : (uint32_t)pv.reads[cur.off + r];
          const uint32_t bq = byte & 127u;
          const bool alt = (byte >> 7) != 0;
          const double m = s_tab[bq], e3 = s_tab[128 + bq], h = s_tab[256 + bq];
          g0 *= alt ? e3 : m;                                                // :437
          g1 *= h;                                                           // :438
          g2 *= alt ? m : e3;                                                // :439
          const double tmp = g0 + g1 + g2;                                   // :440
          if (safe) {
            const double y = rcp_refined(tmp);
            g0 = div_by(g0, tmp, y); g1 = div_by(g1, tmp, y); g2 = div_by(g2, tmp, y);   // :441-443
          } else {
            g0 /= tmp; g1 /= tmp; g2 /= tmp;
          }
        }
        g0 += 1e-6; g1 += 1e-6; g2 += 1e-6;                                  // :446-448
        const double tmp = g0 + g1 + g2;
        const double y = rcp_refined(tmp);
        G0 = div_by(g0, tmp, y); G1 = div_by(g1, tmp, y); G2 = div_by(g2, tmp, y);       // :449-452
      }
    }
    {
      const double* __restrict__ g0row = gp0 + (size_t)cur.snp * 3;
      const double x = G0 * g0row[0] + G1 * g0row[1] + G2 * g0row[2];        // :459
      double tm = dmx_log_fast_pinned(x, s_log, lk);
      if (CHK) { if (__builtin_expect(!__builtin_amdgcn_class(x, 0x100), 0)) tm = log_slow(x); }
      t0s[c * TS + ti] = tm;
      *reinterpret_cast<double2*>(&gl[lane * 4]) = make_double2(G0, G1);
      gl[lane * 4 + 2] = G2;
      rowo[lane] = (uint32_t)cur.snp * row_bytes;
    }
    DMX_WAVE_LDS_ORDER();
    const int64_t left = np - tile * T;
    const int cnt = left >= T ? T : (left > 0 ? (int)left : 0);
    // ---- llk0: the cell's first lane adds the tile's terms in pair order
    if (own0) {
      const double* row = &t0s[c * TS];
      int i = 0;
      for (; i + 8 <= cnt; i += 8) {               // loads first, then the ordered adds
        double2 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = *reinterpret_cast<const double2*>(&row[i + 2 * j]);
#pragma unroll
        for (int j = 0; j < 4; ++j) { acc0 += v[j].x; acc0 += v[j].y; }
      }
      for (; i < cnt; ++i) acc0 += row[i];
    }
    // ---- phase 2: lane (c, sample ti) adds its term of every pair of the tile, ascending
    const bool whole = __all(cnt == T);
    int cmax = cnt;
#pragma unroll
    for (int d = T; d < 64; d <<= 1) cmax = max(cmax, __shfl_xor(cmax, d));
    auto request = [&](int p0, float (&a)[U][3]) {          // the rows of pairs p0 .. p0+U-1: this lane's three probabilities of each
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint32_t vo = rowo[c * T + p0 + u] + my_off;
#pragma unroll
        for (int l = 0; l < 3; ++l) a[u][l] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_g, vo + 4u * l, 0, 0));
      }
    };
    auto batch = [&](int p0, const float (&a)[U][3], auto whole_c) {
      constexpr bool WHOLE = decltype(whole_c)::value;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const double2 g01 = *reinterpret_cast<const double2*>(&gl[(c * T + p0 + u) * 4]);
        const double g2 = gl[(c * T + p0 + u) * 4 + 2];
        const double x = g01.x * (double)a[u][0] + g01.y * (double)a[u][1] + g2 * (double)a[u][2];      // :456
        double tm = dmx_log_fast_pinned(x, s_log, lk);
        if (CHK) { if (__builtin_expect(!__builtin_amdgcn_class(x, 0x100), 0)) tm = log_slow(x); }
        if (WHOLE) acc += tm;
        else if (p0 + u < cnt) acc += tm;
      }
    };
    if (whole) {                                   // two row buffers: batch b+1's rows travel while batch b computes
      float ra[2][U][3];
      request(0, ra[0]);
#pragma unroll
      for (int b = 0; b < T / U; ++b) {
        if (b + 1 < T / U) request((b + 1) * U, ra[(b + 1) & 1]);
        batch(b * U, ra[b & 1], std::true_type{});
      }
    } else {
#pragma unroll 1
      for (int p0 = 0; p0 < cmax; p0 += U) { float ra[U][3]; request(p0, ra); batch(p0, ra, std::false_type{}); }
    }
    DMX_WAVE_LDS_ORDER();
  }
  if (own) llks[(size_t)cell * V + ti] = acc;
  if (own0) llk0s[cell] = acc0;
}

// K1 over genotype classes (see "Genotype classes" below: <= 4 distinct probability rows per SNP, --field GT).
// Same walk, ownership and ordered sums as k_singlet.  Per pair the lane evaluates log(GL . row_d) ONCE per class d (plus
// the llk0 term), parks those doubles in a per-lane LDS scratch, and for every sample k of the chunk copies
// scratch[class id of (snp, k)] into the chain buffer — the very double k_singlet would have computed for sample k
// (same operands, same operations), so the sums that follow are bit-identical.  4+1 logs per pair instead of V+1.
// CAN (round 4; canonical GT classes, see k_canon_apply; launched with chk == 0 only): classes 0..2 are the three SNP-independent rows of a called
// genotype, so log(GL . row) of a pair with up to three tabulated reads is an entry of ltab (k_build_canon_logs: the same expression, the same
// log, evaluated once per read combination instead of once per pair) — three of the five dot products and logs of a pair become one gather,
// requested a tile ahead with the GL seed.  Class 3 (the SNP's own row, where oth[snp] says there is one) and the llk0 term are evaluated as before.
template <int CW, int KC, bool CAN = false>
__global__ __launch_bounds__(kThreads, 5) void k_singlet_cls(PileupView pv, int nrd_width, const float* __restrict__ rows, int chk,
                                                             const uint32_t* __restrict__ idw, const double* __restrict__ gp0,
                                                             const double* __restrict__ tabs,
                                                             const int32_t* __restrict__ sched, int32_t V,
                                                             double* __restrict__ llks, double* __restrict__ llk0s,
                                                             const double* __restrict__ ltab, const uint8_t* __restrict__ oth, double chi, double clo) {
  static_assert(KC == 8 || KC == 4, "a chunk's 2-bit class ids must sit inside one 32-bit id word");
  constexpr int ablate = DMX_ABLATE;             // profiling builds only (tools/build_variant.sh); 0 in the product
  constexpr int T = 64 / CW;
  constexpr int TS = T + 2;
  constexpr int NC = KC + 1;
  constexpr int NW = kThreads / 64;
  constexpr int SD = 4;                          // scratch doubles per lane: the 4 class terms (the llk0 term goes straight to its chain)
  static_assert(CW * NC <= 64, "one lane per chain");
  extern __shared__ double s_dyn[];              // [NW][nch][CW*NC] running accumulators
  // LDS budget: 5 workgroups per CU (<= 32 KB each) so that 10 k wavefronts make two full rounds of 5 120 instead of
  // three of 4 096.  Only dmx_log's table is staged; the read LUT and the first-read tables are read from global memory
  // (L1-hot, 3 + 12 KB) — they are touched once per pair at most.
  __shared__ double s_log_tab[DMX_LOG_TABLE_DOUBLES];
  __shared__ __attribute__((aligned(16))) double s_term[NW][CW * NC * TS];
  __shared__ double s_scr[NW][64 * SD];
  const double* s_log = s_log_tab;
  const double* s_tab = tabs;                    // read LUT in global memory

  const int t = threadIdx.x, w = t >> 6, lane = t & 63;
  const int nch = (V + KC - 1) / KC;
  const int nwd = (V + 15) / 16;
  for (int i = t; i < DMX_LOG_TABLE_DOUBLES; i += kThreads) s_log_tab[i] = tabs[kLut + i];
  for (int i = t; i < NW * nch * CW * NC; i += kThreads) s_dyn[i] = 0.0;
  __syncthreads();                               // the only workgroup barrier

  double* term = s_term[w];
  double* scr = &s_scr[w][lane];                 // class-major [d][lane]: a lane's four terms sit 512 bytes apart, i.e. in the SAME
                                                 // bank pair, and the 32 lanes of an LDS cycle in 32 different ones whatever class each
                                                 // of them looks up (lane-major [lane][d] made every lookup a 4-way bank conflict)
  double* accs = s_dyn + (size_t)w * nch * CW * NC;
  const int slot0 = (blockIdx.x * NW + w) * CW;
  if (slot0 >= pv.B) return;

  const int c = lane / T, ti = lane % T;
  const bool cell_ok = slot0 + c < pv.B;
  const int32_t cell = cell_ok ? sched[slot0 + c] : 0;
  const int64_t p_beg = cell_ok ? pv.cell_pair_off[cell] : 0;
  const int64_t np = cell_ok ? pv.cell_pair_off[cell + 1] - p_beg : 0;
  int64_t rd_base = cell_ok ? pv.cell_read_off[cell] : 0;
  int64_t max_np = np;
#pragma unroll
  for (int d = T; d < 64; d <<= 1) max_np = max(max_np, __shfl_xor(max_np, d));
  const int a_c = lane / NC, a_kk = lane % NC;
  const bool a_ok = lane < CW * NC && slot0 + a_c < pv.B;
  const int64_t a_np = __shfl(np, (a_ok ? a_c : 0) * T);
  const int32_t a_cell = __shfl(cell, (a_ok ? a_c : 0) * T);

  struct Raw { uint32_t n; int32_t snp; };
  struct Hdr { uint32_t n; int32_t snp; uint32_t rd4; int64_t off; };
  auto issue = [&](int64_t tile) {
    Raw r;
    const int64_t pi = tile * T + ti;
    const bool v = pi < np;
    r.n = v ? load_nrd(pv.pair_nrd, p_beg + pi, nrd_width) : 0u;
    r.snp = v ? (pv.pair_snp ? pv.pair_snp[p_beg + pi] : (int32_t)pi) : 0;
    return r;
  };
  auto prepare = [&](const Raw& r) {
    Hdr h;
    h.n = r.n; h.snp = r.snp;
    const uint32_t incl = seg_scan_incl<T>(r.n);
    h.off = rd_base + (int64_t)(incl - r.n);
    rd_base += seg_last<T>(incl, lane);
    h.rd4 = 0;
    if (r.n > 0) {
      if (h.off + 4 <= pv.R) __builtin_memcpy(&h.rd4, pv.reads + h.off, 4);
      else for (int64_t i = h.off; i < pv.R; ++i) h.rd4 |= (uint32_t)pv.reads[i] << (8 * (int)(i - h.off));
    }
    return h;
  };

  // CAN: the pair's three canonical-class terms from ltab (has: the reads are inside the final tables, i.e. gl_seed's r0 >= n)
  struct CanSeed { double l0, l1, l2; bool has; };
  auto can_seed = [&](uint32_t n, uint32_t rd4) {
    CanSeed c; c.l0 = c.l1 = c.l2 = 0.0; c.has = false;
    if constexpr (CAN) {
      const uint32_t b0 = rd4 & 0xFFu, b1 = (rd4 >> 8) & 0xFFu, b2 = (rd4 >> 16) & 0xFFu;
      const uint32_t q0 = b0 & 127u, q1 = b1 & 127u, q2 = b2 & 127u;
      const bool two = n == 2 && ((b0 | b1) & 0x40u) == 0;
      const bool three = n == 3 && max(max(q0, q1), q2) < (uint32_t)dmx::kTripleBq;
      const uint32_t i2 = ((((b0 & 0x80u) >> 1) | (b0 & 0x3Fu)) << 7) | (((b1 & 0x80u) >> 1) | (b1 & 0x3Fu));
      const uint32_t c0 = ((b0 & 0x80u) ? (uint32_t)dmx::kTripleBq : 0u) + q0, c1 = ((b1 & 0x80u) ? (uint32_t)dmx::kTripleBq : 0u) + q1,
                     c2 = ((b2 & 0x80u) ? (uint32_t)dmx::kTripleBq : 0u) + q2;
      const uint32_t i3 = __umul24(__umul24(c0, (uint32_t)dmx::kTripleCodes) + c1, (uint32_t)dmx::kTripleCodes) + c2;
      uint32_t i = n == 0 ? 256u : b0;
      i = two ? (uint32_t)kCanL2 + i2 : i;
      i = three ? (uint32_t)kCanL3 + i3 : i;
      c.has = n < 2 || two || three;
      if (c.has) {
        const double2 v = *reinterpret_cast<const double2*>(ltab + 4 * (size_t)i);
        c.l0 = v.x; c.l1 = v.y; c.l2 = ltab[4 * (size_t)i + 2];
      }
    }
    return c;
  };
  Hdr h1 = prepare(issue(0));
  Hdr h2 = prepare(issue(1));
  Raw pre = issue(2);
  GlSeed s1 = gl_seed(tabs, h1.n, h1.rd4);
  CanSeed c1 = can_seed(h1.n, h1.rd4);
  for (int64_t tile = 0; tile * T < max_np; ++tile) {
    const Hdr cur = h1;
    const GlSeed cs = s1;
    const CanSeed cc = c1;
    h1 = h2;
    s1 = gl_seed(tabs, h1.n, h1.rd4);              // tile+1: its read bytes were requested an iteration ago
    c1 = can_seed(h1.n, h1.rd4);
    h2 = prepare(pre);                             // tile+2: scan, request its read bytes
    pre = issue(tile + 3);                         // tile+3: header loads in flight
    const bool valid = tile * T + ti < np;

    // class rows, llk0 row and the first id word of this lane's SNP (SNP-major: contiguous across a dense tile)
    const int32_t snp_l = (ablate & 256) ? (cur.snp & 63) : cur.snp;     // ablation: every row load an L1 hit
    const float4* rp = reinterpret_cast<const float4*>(rows + (size_t)snp_l * 12);
    float4 r0 = {0.f, 0.f, 0.f, 0.f}, r1 = r0, r2 = r0;
    bool has_o = false;
    if constexpr (CAN) { has_o = valid && oth[snp_l] != 0; if (has_o) r2 = rp[2]; }   // (class 3's row: r2.y, r2.z, r2.w)
    else { r0 = rp[0]; r1 = rp[1]; r2 = rp[2]; }
    const double* g0 = gp0 + (size_t)snp_l * 3;
    const double q0 = g0[0], q1 = g0[1], q2 = g0[2];
    const uint32_t* idrow = idw + (size_t)snp_l * nwd;
    uint32_t wcur = idrow[0];

    double G0, G1, G2;                                                       // :427-452
    gl_finish(cs, (ablate & 1) ? min(cur.n, cs.r0) : cur.n, cur.rd4, pv.reads, cur.off, s_tab, G0, G1, G2);
    if constexpr (CAN) {
      if (valid) {
        double t0 = cc.l0, t1 = cc.l1, t2 = cc.l2;
        if (!cc.has) {                               // deeper pairs / base qualities beyond the tables: the expressions of k_build_canon_logs
          t0 = dmx_log_fast(G0 * chi + G1 * clo + G2 * clo, s_log);
          t1 = dmx_log_fast(G0 * clo + G1 * chi + G2 * clo, s_log);
          t2 = dmx_log_fast(G0 * clo + G1 * clo + G2 * chi, s_log);
        }
        scr[0] = t0; scr[64] = t1; scr[128] = t2;
        if (has_o) scr[192] = dmx_log_fast(G0 * (double)r2.y + G1 * (double)r2.z + G2 * (double)r2.w, s_log);   // class 3
        term[(c * NC + KC) * TS + ti] = dmx_log_fast(G0 * q0 + G1 * q1 + G2 * q2, s_log);                         // llk0 (:459)
      }
    } else
    if (valid) {
      const double x0 = G0 * (double)r0.x + G1 * (double)r0.y + G2 * (double)r0.z;     // class 0   (:456)
      const double x1 = G0 * (double)r0.w + G1 * (double)r1.x + G2 * (double)r1.y;     // class 1
      const double x2 = G0 * (double)r1.z + G1 * (double)r1.w + G2 * (double)r2.x;     // class 2
      const double x3 = G0 * (double)r2.y + G1 * (double)r2.z + G2 * (double)r2.w;     // class 3
      const double x4 = G0 * q0 + G1 * q1 + G2 * q2;                                    // llk0      (:459)
      // chk == 0: k_check_geno proved every row finite, non-negative, with an entry >= 1e-30, and GL >= 1e-6 / (1 + 3e-6) (:452): every
      // argument is a positive normal number and the class test cannot fire (what the K2 kernels' CHK = false variants rely on)
      const bool fast_ok = !chk || (__builtin_amdgcn_class(x0, 0x100) && __builtin_amdgcn_class(x1, 0x100) && __builtin_amdgcn_class(x2, 0x100) &&
                                    __builtin_amdgcn_class(x3, 0x100) && __builtin_amdgcn_class(x4, 0x100));
      double* t0 = &term[(c * NC + KC) * TS + ti];   // the llk0 chain's slot of this pair
      if (__builtin_expect(fast_ok, 1)) {
        scr[0] = dmx_log_fast(x0, s_log); scr[64] = dmx_log_fast(x1, s_log); scr[128] = dmx_log_fast(x2, s_log);
        scr[192] = dmx_log_fast(x3, s_log); *t0 = dmx_log_fast(x4, s_log);
      } else {                                     // never for real likelihoods; keeps log(0) / log(nan) semantics
        scr[0] = x0; scr[64] = x1; scr[128] = x2; scr[192] = x3; *t0 = x4;
        for (int d = 0; d < SD; ++d) scr[64 * d] = log(scr[64 * d]);
        *t0 = log(*t0);
      }
    }
    for (int q = 0; q < nch; ++q) {
      const int k0 = q * KC;
      if (q > 0 && (k0 & 15) == 0) wcur = idrow[k0 >> 4];
      if (valid && !(ablate & 64)) {
        const uint32_t bits = wcur >> (2 * (k0 & 15));      // the chunk's KC class ids, 2 bits each
#pragma unroll
        for (int kk = 0; kk < KC; ++kk)
          term[(c * NC + kk) * TS + ti] = scr[((bits >> (2 * kk)) & 3u) << 6];     // sample k0+kk's term (slots past V-1 are never summed)
      }
      DMX_WAVE_LDS_ORDER();
      if (a_ok && (a_kk < KC ? k0 + a_kk < V : q == 0)) {
        const int64_t left = a_np - tile * T;
        const int cnt = (ablate & 4) ? 1 : (left >= T ? T : (left > 0 ? (int)left : 0));
        const double* row = &term[lane * TS];
        double s = accs[q * (CW * NC) + lane];
        int i = 0;
        for (; i + 16 <= cnt; i += 16) {
          double2 v[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const double2*>(&row[i + 2 * j]);
#pragma unroll
          for (int j = 0; j < 8; ++j) { s += v[j].x; s += v[j].y; }
        }
        for (; i < cnt; ++i) s += row[i];           // ascending SNP order: the reference's order
        accs[q * (CW * NC) + lane] = s;
      }
      DMX_WAVE_LDS_ORDER();
    }
  }
  if (a_ok) {
    for (int q = 0; q < nch; ++q) {
      const double s = accs[q * (CW * NC) + lane];
      if (a_kk < KC) { const int k = q * KC + a_kk; if (k < V) llks[(size_t)a_cell * V + k] = s; }
      else if (q == 0) llk0s[a_cell] = s;
    }
  }
}

// K1 over CANONICAL genotype classes, lean form (round 5; V <= 16, one-byte read counts, arrays below 4 GiB — cfg2 — else k_singlet_cls<.., CAN>).
// Same ownership, same expressions, same order of additions as k_singlet_cls<CW, KC, true>: every output is bit-identical.  What changed is
// everything AROUND the arithmetic, which was three quarters of that kernel's instruction stream (profiles/pmc_cfg2_strict.json: 76 % of its
// VALU instructions were not FP64) and all of its exposed latency:
//   * ONE table gather per pair instead of two: ctab[i] = {GL0, GL1, GL2, log(GL . row_0), log(GL . row_1), log(GL . row_2)} (64-byte entries,
//     k_build_ctab: the GL of gl_seed + gl_finish and the logs of k_build_canon_logs, evaluated once per read combination), indexed by the
//     read count and the RAW leading read bytes — one read: the byte; two reads: the 16-bit word (all 65 536 combinations, no quality test);
//     three reads of quality < 48: the 96^3 code.  Anything else (0.25 % of the pairs at 1.25 reads per pair) walks the read loop as before.
//   * ONE record per SNP instead of four arrays: SnpRec = {gp0[3], the id word, the "has a fourth row" flag} (32 bytes, k_build_snprec).
//   * 32-bit element offsets from wave-uniform array bases (SGPR base + VGPR offset loads) instead of 64-bit pointer arithmetic per load.
//   * A three-deep software pipeline in which NOTHING a tile consumes was requested in the same iteration: tile t + 3's read counts, tile
//     t + 2's scan and leading read bytes, tile t + 1's table entry and SNP record are in flight while tile t computes (the memory system
//     returns a wavefront's loads in order, so the old kernel's wait for the current tile's rows also waited for the prefetches behind them).
// per-SNP record, four 16-byte parts in four arrays (SoA; a dense tile reads each contiguously): [S] {q0, q1} | [S] {q2, ids 0..3} | [S] {ids 4..11} | [S] {ids 12..15, oth, 0}
// (gp0 = q; id k = a 16-bit 512 * class of sample k: the byte offset of the class's plane in the scratch)
constexpr size_t kSnpRecBytes = 64;
constexpr uint32_t kCt2 = 257, kCt3 = 257 + 65536;                              // ctab regions: one read (256 = none) | two reads (raw bytes) | three reads
constexpr int64_t kCtN = (int64_t)kCt3 + (int64_t)dmx::kTripleCodes * dmx::kTripleCodes * dmx::kTripleCodes;
constexpr int kCtBq = 42, kCtLds = 2 * kCtBq + 1;                                // LDS copy of region 1: one read of quality < 42 (code allele * 42 + bq; the CLI caps at 40), 84 = no read

__global__ void k_build_snprec(const double* __restrict__ gp0, const uint8_t* __restrict__ ids, const uint8_t* __restrict__ oth, int32_t S, int32_t V,
                               uint4* __restrict__ out) {
  const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= S) return;
  const double q0 = gp0[3 * s], q1 = gp0[3 * s + 1], q2 = gp0[3 * s + 2];
  uint32_t w[8] = {0, 0, 0, 0, 0, 0, 0, 0};      // sample k's class id as the BYTE OFFSET of its class plane in the scratch: half-word k = id * 512
  for (int k = 0; k < V && k < 16; ++k) w[k >> 1] |= ((uint32_t)(ids[s * V + k] & 3u) * 512u) << (16 * (k & 1));
  out[s] = make_uint4((uint32_t)__double2loint(q0), (uint32_t)__double2hiint(q0), (uint32_t)__double2loint(q1), (uint32_t)__double2hiint(q1));
  out[(size_t)S + s] = make_uint4((uint32_t)__double2loint(q2), (uint32_t)__double2hiint(q2), w[0], w[1]);
  out[2 * (size_t)S + s] = make_uint4(w[2], w[3], w[4], w[5]);
  out[3 * (size_t)S + s] = make_uint4(w[6], w[7], (uint32_t)oth[s], 0u);
}

// ctab entry i: the pair's genotype likelihoods exactly as gl_seed + gl_finish produce them (the host tables where they reach, the kernel's own
// read loop beyond), and the three canonical-class terms with k_build_canon_logs' expressions and log.
__global__ void k_build_ctab(const double* __restrict__ tabs, double hi, double lo, double* __restrict__ ctab) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= kCtN) return;
  uint32_t n, rd4;
  if (i < kCt2) { n = i == 256 ? 0u : 1u; rd4 = (uint32_t)i & 0xFFu; }
  else if (i < kCt3) { n = 2u; rd4 = (uint32_t)(i - kCt2); }
  else {
    uint32_t c = (uint32_t)(i - kCt3);
    const uint32_t c2 = c % (uint32_t)dmx::kTripleCodes; c /= (uint32_t)dmx::kTripleCodes;
    const uint32_t c1 = c % (uint32_t)dmx::kTripleCodes, c0 = c / (uint32_t)dmx::kTripleCodes;
    auto byte_of = [](uint32_t code) { return code >= (uint32_t)dmx::kTripleBq ? (0x80u | (code - (uint32_t)dmx::kTripleBq)) : code; };
    n = 3u; rd4 = byte_of(c0) | (byte_of(c1) << 8) | (byte_of(c2) << 16);
  }
  const GlSeed sd = gl_seed(tabs, n, rd4);
  double G0, G1, G2;
  gl_finish(sd, n, rd4, nullptr, 0, tabs, G0, G1, G2);
  const double* T = tabs + kLut;
  double* o = ctab + 8 * i;
  o[0] = G0; o[1] = G1; o[2] = G2;
  o[3] = dmx_log_fast(G0 * hi + G1 * lo + G2 * lo, T);
  o[4] = dmx_log_fast(G0 * lo + G1 * hi + G2 * lo, T);
  o[5] = dmx_log_fast(G0 * lo + G1 * lo + G2 * hi, T);
  o[6] = 0.0; o[7] = 0.0;
}

// The first version of this kernel (everything gathered from the global table) was bound by the CU's vector L1: one tag look-up per lane and
// gather instruction, 273 per 64-pair tile = 0.94 look-ups per cycle and CU (profiles/r05_k1_probe.txt) — not by issue, not by bandwidth.  The LDS
// serves 16 lanes per cycle, so the table entries of the commonest pairs — no read, or ONE read of quality < 42 (the CLI caps at 40): 78 % of the
// pairs at 1.25 reads per pair — come from a 4 KB LDS copy and only the other lanes gather from the global table, a tile ahead.  The copy has to fit
// beside 5 workgroups per CU (10 000 barcodes at two per wavefront are ONE round of 5 wavefronts per SIMD; a workgroup of ten wavefronts sharing a
// bigger copy is admitted once per CU only — measured): 32 000 bytes per workgroup (the LDS is handed out in 1 280-byte pieces: 32 432 bytes were
// admitted four times per CU), found by keeping the running sums in registers and, for matrices without a fourth genotype row (OTH = false: no
// missing genotypes), three class planes of scratch instead of four.  OTH = true keeps the
// global gathers for every lane.
// (waves per SIMD: five for the one- and two-barcode forms — 10 000 barcodes at two per wavefront are ONE round of five; the four-barcode form, picked
//  from 131 072 barcodes, carries four barcodes' pipelines in its registers and asks for what it gets: four, three with two chunks — VERDICT r5 weak 9)
template <int CW, int KC, bool OTH, int NCH>     // NCH: chunks of KC samples (1 or 2; V <= NCH * KC)
__global__ __launch_bounds__(kThreads, (CW == 4 ? (NCH == 2 ? 3 : 4) : 5)) void k_singlet_can(PileupView pv, const uint4* __restrict__ snprec, const float* __restrict__ rows,
                                                             const double* __restrict__ ctab, const double* __restrict__ tabs,
                                                             const int32_t* __restrict__ sched, int32_t V,
                                                             double* __restrict__ llks, double* __restrict__ llk0s, double chi, double clo) {
  static_assert(KC == 8 || KC == 4, "a chunk's 2-bit class ids must sit inside one 32-bit id word");
  constexpr int T = 64 / CW;
  constexpr int TS = T + 2;
  constexpr int NC = KC + 1;
  constexpr int NW = kThreads / 64;
  constexpr int NPL = OTH ? 4 : 3;               // class planes of scratch
  static_assert(CW * NC <= 64, "one lane per chain");
  __shared__ double s_log_tab[DMX_LOG_TABLE_DOUBLES];
  __shared__ __attribute__((aligned(16))) double s_term[NW][CW * NC * TS];
  __shared__ double s_scr[NW][64 * NPL];
  __shared__ __attribute__((aligned(16))) double s_ct[OTH ? 2 : kCtLds * 6];
  const double* s_log = s_log_tab;

  const int t = threadIdx.x, w = t >> 6, lane = t & 63;
  constexpr int nch = NCH;
  constexpr int NIDW = NCH * KC / 2;             // id words per SNP (two samples each)
  for (int i = t; i < DMX_LOG_TABLE_DOUBLES; i += kThreads) s_log_tab[i] = tabs[kLut + i];
  if constexpr (!OTH)
    for (int i = t; i < kCtLds * 6; i += kThreads) {
      const int e = i / 6, f = i % 6;
      const int b = e == 2 * kCtBq ? 256 : (e >= kCtBq ? (0x80 | (e - kCtBq)) : e);   // the read byte (allele << 7) | bq of code e
      s_ct[i] = ctab[8 * b + f];
    }
  __syncthreads();                               // the only workgroup barrier

  double* term = s_term[w];
  double* scr = &s_scr[w][lane];                 // class-major [d][lane] (conflict-free whatever class each lane looks up, see k_singlet_cls): a sample's
                                                 // term is at (lane address + 512 * class) — ONE v_add_u32_sdwa per look-up with the ids stored as that offset
  const uint32_t scr_a = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) double*)scr;   // (its LDS byte address)
  const uint32_t ct_a = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) double*)s_ct;
  const int slot0 = (blockIdx.x * NW + w) * CW;
  if (slot0 >= pv.B) return;

  const int c = lane / T, ti = lane % T;
  const bool cell_ok = slot0 + c < pv.B;
  const int32_t cell = cell_ok ? sched[slot0 + c] : 0;
  const uint32_t p_beg = cell_ok ? (uint32_t)pv.cell_pair_off[cell] : 0u;               // (the launcher checked: every offset fits 32 bits)
  const uint32_t np = cell_ok ? (uint32_t)(pv.cell_pair_off[cell + 1] - pv.cell_pair_off[cell]) : 0u;
  uint32_t rd_base = cell_ok ? (uint32_t)pv.cell_read_off[cell] : 0u;
  uint32_t max_np = np;
#pragma unroll
  for (int d = T; d < 64; d <<= 1) max_np = max(max_np, (uint32_t)__shfl_xor((int)max_np, d));
  const int a_c = lane / NC, a_kk = lane % NC;
  const bool a_ok = lane < CW * NC && slot0 + a_c < pv.B;
  const uint32_t a_np = (uint32_t)__shfl((int)np, (a_ok ? a_c : 0) * T);
  const int32_t a_cell = __shfl(cell, (a_ok ? a_c : 0) * T);
  double acc[2] = {0.0, 0.0};                    // the chain's running sums (chunk 0, chunk 1), in registers
  const uint8_t* __restrict__ nrd8 = (const uint8_t*)pv.pair_nrd;
  const uint8_t* __restrict__ reads = pv.reads;
  const int32_t* __restrict__ psnp = pv.pair_snp;
  const uint4* __restrict__ recA = snprec;
  const uint4* __restrict__ recB = snprec + pv.S;
  const uint4* __restrict__ recC = snprec + 2 * (size_t)pv.S;
  const uint4* __restrict__ recD = snprec + 3 * (size_t)pv.S;
  constexpr bool need_c = NIDW > 2, need_d = OTH || NIDW > 6;

  // stage A: read count (and SNP id) of the lane's pair of a tile
  struct Raw { uint32_t n; uint32_t snp; };
  auto stage_a = [&](uint32_t tile) {
    Raw r;
    const uint32_t pi = tile * T + ti;
    const bool v = pi < np;
    const uint32_t pa = v ? p_beg + pi : 0u;
    r.n = v ? (uint32_t)nrd8[pa] : 0u;
    r.snp = v ? (psnp ? (uint32_t)psnp[pa] : pi) : 0u;
    return r;
  };
  // stage B: the pair's read offset (scan over the cell's lanes) and its leading read bytes
  struct Hdr { uint32_t n, snp, rd4, off; };
  auto stage_b = [&](const Raw& r) {
    Hdr h;
    h.n = r.n; h.snp = r.snp;
    const uint32_t incl = seg_scan_incl<T>(r.n);
    h.off = rd_base + (incl - r.n);
    rd_base += seg_last<T>(incl, lane);
    uint32_t v;
    __builtin_memcpy(&v, reads + (r.n ? h.off : 0u), 4);        // (the launcher checked: four bytes past the last read are readable)
    h.rd4 = v;
    return h;
  };
  // stage C: the pair's table entry — requested from the global table a tile ahead unless the LDS copy has it (no read, or one read of quality
  // < kCtBq), in which case stage D reads it when the tile computes — and its SNP's record
  struct Seed { double2 a, b, cc; double q0, q1, q2; uint32_t idb[NIDW], oth, lds; bool fast; };   // lds: byte offset of the entry in the LDS copy, or ~0u
  auto in_lds_copy = [](uint32_t n, uint32_t rd4) { return !OTH && (n == 0 || (n == 1 && (rd4 & 0x7Fu) < (uint32_t)kCtBq)); };
  auto stage_c = [&](const Hdr& h) {
    Seed sd;
    const uint32_t n = h.n, rd4 = h.rd4;
    sd.fast = true;
    sd.lds = (n == 0 ? (uint32_t)(2 * kCtBq) : (((rd4 & 0x80u) ? (uint32_t)kCtBq : 0u) + (rd4 & 0x7Fu))) * 48u;
    if (!in_lds_copy(n, rd4)) {
      sd.lds = ~0u;
      uint32_t idx = n == 0 ? 256u : (rd4 & 0xFFu);
      idx = n == 2 ? kCt2 + (rd4 & 0xFFFFu) : idx;
      bool fast = n <= 2;
      if (n == 3) {
        const uint32_t q0 = rd4 & 127u, q1 = (rd4 >> 8) & 127u, q2 = (rd4 >> 16) & 127u;
        if (max(max(q0, q1), q2) < (uint32_t)dmx::kTripleBq) {
          const uint32_t c0 = ((rd4 & 0x80u) ? (uint32_t)dmx::kTripleBq : 0u) + q0, c1 = ((rd4 & 0x8000u) ? (uint32_t)dmx::kTripleBq : 0u) + q1,
                         c2 = ((rd4 & 0x800000u) ? (uint32_t)dmx::kTripleBq : 0u) + q2;
          idx = kCt3 + __umul24(__umul24(c0, (uint32_t)dmx::kTripleCodes) + c1, (uint32_t)dmx::kTripleCodes) + c2;
          fast = true;
        }
      }
      sd.fast = fast;
      const double2* e = reinterpret_cast<const double2*>(ctab + 8u * idx);
      sd.a = e[0]; sd.b = e[1]; sd.cc = e[2];
    }
    const uint4 r0 = recA[h.snp], r1 = recB[h.snp];
    sd.q0 = __hiloint2double((int)r0.y, (int)r0.x); sd.q1 = __hiloint2double((int)r0.w, (int)r0.z);
    sd.q2 = __hiloint2double((int)r1.y, (int)r1.x); sd.idb[0] = r1.z; sd.idb[1] = r1.w;
    sd.oth = 0u;
    if constexpr (need_c) { const uint4 r2 = recC[h.snp]; sd.idb[2] = r2.x; sd.idb[3] = r2.y; if constexpr (NIDW > 4) { sd.idb[4] = r2.z; sd.idb[5] = r2.w; } }
    if constexpr (need_d) { const uint4 r3 = recD[h.snp]; if constexpr (NIDW > 6) { sd.idb[6] = r3.x; sd.idb[7] = r3.y; } sd.oth = r3.z; }
    return sd;
  };
  typedef double v2d_t __attribute__((ext_vector_type(2)));
  typedef const __attribute__((address_space(3))) v2d_t* LdsD2;        // (an LDS-address-space pointer: the two sources must not be merged into flat loads)

  // the sums of chunk q belong to lane (cell a_c, slot a_kk): sample q * KC + a_kk, or (slot KC, chunk 0) the llk0 chain
  const bool sum_lane[2] = {a_ok && (a_kk < KC ? a_kk < V : true), a_ok && a_kk < KC && KC + a_kk < V};
  uint32_t min_np = cell_ok ? np : 0xFFFFFFFFu;  // tiles below it are whole for every barcode of the wavefront: their sums need no counting
#pragma unroll
  for (int d = T; d < 64; d <<= 1) min_np = min(min_np, (uint32_t)__shfl_xor((int)min_np, d));

  // stage D: one tile.  Lanes beyond their barcode's last pair carry n = 0, SNP 0 (stage A): they compute like any other lane, into slots the
  // sums never reach — no per-lane validity test anywhere below.
  auto compute = [&](const Hdr& cur, Seed& cs, uint32_t tile) {
    if (cs.lds != ~0u) {                           // the LDS copy's entry
      LdsD2 p = (LdsD2)(uintptr_t)(ct_a + cs.lds);
      const v2d_t va = p[0], vb = p[1], vc = p[2];
      cs.a.x = va.x; cs.a.y = va.y; cs.b.x = vb.x; cs.b.y = vb.y; cs.cc.x = vc.x; cs.cc.y = vc.y;
    }
    double G0 = cs.a.x, G1 = cs.a.y, G2 = cs.b.x, t0 = cs.b.y, t1 = cs.cc.x, t2 = cs.cc.y;
    if (!cs.fast) {                                // deeper pairs, qualities beyond the tables: the read loop and the three logs, as k_singlet_cls<.., CAN>
      const GlSeed sd = gl_seed(tabs, cur.n, cur.rd4);
      gl_finish(sd, cur.n, cur.rd4, reads, (int64_t)cur.off, tabs, G0, G1, G2);
      t0 = dmx_log_fast(G0 * chi + G1 * clo + G2 * clo, s_log);
      t1 = dmx_log_fast(G0 * clo + G1 * chi + G2 * clo, s_log);
      t2 = dmx_log_fast(G0 * clo + G1 * clo + G2 * chi, s_log);
    }
    scr[0] = t0; scr[64] = t1; scr[128] = t2;
    if constexpr (OTH)
      if (cs.oth) {                                // class 3: the SNP's own fourth row (a missing genotype's Hardy-Weinberg row)
        const float* r3 = rows + (size_t)cur.snp * 12 + 9;
        scr[192] = dmx_log_fast(G0 * (double)r3[0] + G1 * (double)r3[1] + G2 * (double)r3[2], s_log);
      }
    term[(c * NC + KC) * TS + ti] = dmx_log_fast(G0 * cs.q0 + G1 * cs.q1 + G2 * cs.q2, s_log);            // llk0 (:459)
    const uint32_t done = tile * T;
    const bool whole = done + T <= min_np;         // (wave-uniform)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      if (q < nch) {
        // sample q*KC+kk's term = scratch[its class][lane]: address = lane address + the sample's id (slots past V-1 are never summed; class 3 only with OTH)
        typedef const __attribute__((address_space(3))) double* LdsD;
        const int h0 = q * (KC / 2);               // the chunk's first id word (two samples per word)
        term[(c * NC + 0) * TS + ti] = *(LdsD)(uintptr_t)add_word_of<0>(scr_a, cs.idb[h0]);
        term[(c * NC + 1) * TS + ti] = *(LdsD)(uintptr_t)add_word_of<1>(scr_a, cs.idb[h0]);
        term[(c * NC + 2) * TS + ti] = *(LdsD)(uintptr_t)add_word_of<0>(scr_a, cs.idb[h0 + 1]);
        term[(c * NC + 3) * TS + ti] = *(LdsD)(uintptr_t)add_word_of<1>(scr_a, cs.idb[h0 + 1]);
        if constexpr (KC == 8) {
          term[(c * NC + 4) * TS + ti] = *(LdsD)(uintptr_t)add_word_of<0>(scr_a, cs.idb[h0 + 2]);
          term[(c * NC + 5) * TS + ti] = *(LdsD)(uintptr_t)add_word_of<1>(scr_a, cs.idb[h0 + 2]);
          term[(c * NC + 6) * TS + ti] = *(LdsD)(uintptr_t)add_word_of<0>(scr_a, cs.idb[h0 + 3]);
          term[(c * NC + 7) * TS + ti] = *(LdsD)(uintptr_t)add_word_of<1>(scr_a, cs.idb[h0 + 3]);
        }
        DMX_WAVE_LDS_ORDER();
        if (sum_lane[q]) {
          const double* row = &term[lane * TS];
          double sacc = acc[q];
          if (whole) {                             // ascending SNP order: the reference's order
#pragma unroll
            for (int i = 0; i < T; i += 16) {
              double2 v[8];
#pragma unroll
              for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const double2*>(&row[i + 2 * j]);
#pragma unroll
              for (int j = 0; j < 8; ++j) { sacc += v[j].x; sacc += v[j].y; }
            }
          } else {
            const int cnt = a_np >= done + T ? T : (a_np > done ? (int)(a_np - done) : 0);
            for (int i = 0; i < cnt; ++i) sacc += row[i];
          }
          acc[q] = sacc;
        }
        DMX_WAVE_LDS_ORDER();
      }
    }
  };

  // two tiles per trip so that the two Seed sets alternate instead of being copied (the compiler's loop-carried copies were a tenth of the stream)
  Hdr h1 = stage_b(stage_a(0));
  Hdr h2 = stage_b(stage_a(1));
  Raw pre = stage_a(2);
  Seed sA = stage_c(h1), sB = sA;
  for (uint32_t tile = 0; tile * T < max_np; tile += 2) {
    {
      const Hdr cur = h1;
      h1 = h2;
      sB = stage_c(h1);                            // tile + 1
      h2 = stage_b(pre);                           // tile + 2
      pre = stage_a(tile + 3);                     // tile + 3
      compute(cur, sA, tile);
    }
    if ((tile + 1) * T >= max_np) break;
    {
      const Hdr cur = h1;
      h1 = h2;
      sA = stage_c(h1);
      h2 = stage_b(pre);
      pre = stage_a(tile + 4);
      compute(cur, sB, tile + 1);
    }
  }
  if (a_ok) {
#pragma unroll
    for (int q = 0; q < 2; ++q)
      if (q < nch) {
        if (a_kk < KC) { const int k = q * KC + a_kk; if (k < V) llks[(size_t)a_cell * V + k] = acc[q]; }
        else if (q == 0) llk0s[a_cell] = acc[q];
      }
  }
}

// K1, canonical GT classes, DENSE pileups, up to 8 samples (cfg2): producers / consumer (round 6).  k_singlet_can's lanes are pairs
// for the table look-up and the llk0 term and then 18 of 64 lanes are chains: per 64-pair tile it spends 32 ordered adds on 18 lanes and 17 LDS
// instructions on handing the terms from the pair lanes to the chain lanes (9 stores, 8 class look-ups), and the LDS (0.67 busy) binds it with the VALU
// close behind.  Here a workgroup is SEVEN producer wavefronts — one barcode each, lane = pair: headers, table entry, llk0 term, then ONE 32-byte record
// {log(GL . row_0), log(GL . row_1), log(GL . row_2), llk0 term} per pair into an LDS ring — and ONE consumer wavefront whose 63 lanes own the 7 x 9
// chains: chain (barcode b, sample k) adds record[b][pair][class of sample k at the pair's SNP] in ascending SNP order — the doubles k_singlet_can adds,
// in its order: the same bits.  On a dense pileup the pair's SNP is its index, so a sample's classes are one byte stream shared by all barcodes
// (clsb[k][snp] = 8 x class: the byte offset inside a record; row 8 = 24: the llk0 chain): the consumer takes 64 of them in four 16-byte loads a tile
// ahead and a chain step is one v_add_u32_sdwa (record address), one ds_read_b64 and one v_add_f64 — 64 steps per 448 pairs instead of 32 per 64.
// One workgroup barrier per tile: tile t is consumed from ring half t & 1 while the producers fill the other half with tile t + 1.
constexpr int kCpB = 7;                          // barcodes (producer wavefronts) per workgroup
constexpr int kCpThreads = (kCpB + 1) * 64;
__global__ void k_build_clsb(const uint8_t* __restrict__ ids, int32_t S, int32_t V, uint32_t stride, uint8_t* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)9 * stride) return;
  const int k = (int)(i / stride); const int64_t sidx = i % stride;
  out[i] = k == 8 ? (uint8_t)24 : ((k < V && sidx < S) ? (uint8_t)((ids[sidx * V + k] & 3u) * 8u) : (uint8_t)0);
}
#define DMX_CP_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")   // LDS traffic done, then the workgroup's barrier (NOT vmcnt: the prefetches stay in flight)
template <int MINW>
__global__ __launch_bounds__(kCpThreads, MINW) void k_singlet_canp(PileupView pv, const uint4* __restrict__ snprec, const uint8_t* __restrict__ clsb, uint32_t cls_stride,
                                                              const double* __restrict__ ctab, const double* __restrict__ tabs,
                                                              const int32_t* __restrict__ sched, int32_t V,
                                                              double* __restrict__ llks, double* __restrict__ llk0s, double chi, double clo) {
  constexpr int T = 64;
  __shared__ double s_log_tab[DMX_LOG_TABLE_DOUBLES];
  __shared__ __attribute__((aligned(16))) double s_ct[kCtLds * 6];
  __shared__ __attribute__((aligned(16))) double s_ring[2][kCpB][T][4];
  __shared__ uint32_t s_np[kCpB];
  const double* s_log = s_log_tab;
  const int t = threadIdx.x, w = t >> 6, lane = t & 63;
  for (int i = t; i < DMX_LOG_TABLE_DOUBLES; i += kCpThreads) s_log_tab[i] = tabs[kLut + i];
  for (int i = t; i < kCtLds * 6; i += kCpThreads) {
    const int e = i / 6, f = i % 6;
    const int b = e == 2 * kCtBq ? 256 : (e >= kCtBq ? (0x80 | (e - kCtBq)) : e);   // the read byte (allele << 7) | bq of code e
    s_ct[i] = ctab[8 * b + f];
  }
  const int slot0 = blockIdx.x * kCpB;
  if (t < kCpB) s_np[t] = slot0 + t < pv.B ? (uint32_t)(pv.cell_pair_off[sched[slot0 + t] + 1] - pv.cell_pair_off[sched[slot0 + t]]) : 0u;
  __syncthreads();
  uint32_t max_np = 0u, min_np = 0xFFFFFFFFu;    // over the workgroup's barcodes (uniform)
#pragma unroll
  for (int b = 0; b < kCpB; ++b) { const uint32_t v = s_np[b]; max_np = max(max_np, v); if (slot0 + b < pv.B) min_np = min(min_np, v); }
  const uint32_t ntiles = (max_np + T - 1) / T;
  typedef double v2d_t __attribute__((ext_vector_type(2)));
  typedef const __attribute__((address_space(3))) v2d_t* LdsD2;
  typedef const __attribute__((address_space(3))) double* LdsD;

  if (w == kCpB) {
    // ------------------------------------------------------------------------------------------------ the consumer
    const int b = lane / 9, kk = lane % 9;
    const bool valid = lane < kCpB * 9 && slot0 + b < pv.B;
    const int32_t cell = valid ? sched[slot0 + b] : 0;
    const uint32_t np_b = valid ? s_np[b] : 0u;
    const uint8_t* __restrict__ row = clsb + (size_t)kk * cls_stride;
    const uint32_t ring_a = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) double*)&s_ring[0][lane < kCpB * 9 ? b : 0][0][0];
    constexpr uint32_t kHalf = (uint32_t)(sizeof(double) * kCpB * T * 4);
    double acc = 0.0;
    uint4 cw[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) cw[i] = reinterpret_cast<const uint4*>(row)[i];
    for (uint32_t tile = 0; tile < ntiles; ++tile) {
      uint4 nw[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) nw[i] = reinterpret_cast<const uint4*>(row + (size_t)(tile + 1) * T)[i];   // (the table carries one tile of padding)
      DMX_CP_BARRIER();                            // the producers' tile is in the ring
      const uint32_t base = ring_a + (tile & 1u) * kHalf;
      if ((tile + 1) * T <= min_np) {              // whole for every barcode of the workgroup (uniform)
#pragma unroll
        for (int q = 0; q < 4; ++q) {              // 16 steps at a time: addresses, reads, then the ordered adds
          const uint32_t w0 = cw[q].x, w1 = cw[q].y, w2 = cw[q].z, w3 = cw[q].w;
          double v[16];
          v[0] = *(LdsD)(uintptr_t)(add_byte_of<0>(base, w0) + (q * 16 + 0) * 32);
          v[1] = *(LdsD)(uintptr_t)(add_byte_of<1>(base, w0) + (q * 16 + 1) * 32);
          v[2] = *(LdsD)(uintptr_t)(add_byte_of<2>(base, w0) + (q * 16 + 2) * 32);
          v[3] = *(LdsD)(uintptr_t)(add_byte_of<3>(base, w0) + (q * 16 + 3) * 32);
          v[4] = *(LdsD)(uintptr_t)(add_byte_of<0>(base, w1) + (q * 16 + 4) * 32);
          v[5] = *(LdsD)(uintptr_t)(add_byte_of<1>(base, w1) + (q * 16 + 5) * 32);
          v[6] = *(LdsD)(uintptr_t)(add_byte_of<2>(base, w1) + (q * 16 + 6) * 32);
          v[7] = *(LdsD)(uintptr_t)(add_byte_of<3>(base, w1) + (q * 16 + 7) * 32);
          v[8] = *(LdsD)(uintptr_t)(add_byte_of<0>(base, w2) + (q * 16 + 8) * 32);
          v[9] = *(LdsD)(uintptr_t)(add_byte_of<1>(base, w2) + (q * 16 + 9) * 32);
          v[10] = *(LdsD)(uintptr_t)(add_byte_of<2>(base, w2) + (q * 16 + 10) * 32);
          v[11] = *(LdsD)(uintptr_t)(add_byte_of<3>(base, w2) + (q * 16 + 11) * 32);
          v[12] = *(LdsD)(uintptr_t)(add_byte_of<0>(base, w3) + (q * 16 + 12) * 32);
          v[13] = *(LdsD)(uintptr_t)(add_byte_of<1>(base, w3) + (q * 16 + 13) * 32);
          v[14] = *(LdsD)(uintptr_t)(add_byte_of<2>(base, w3) + (q * 16 + 14) * 32);
          v[15] = *(LdsD)(uintptr_t)(add_byte_of<3>(base, w3) + (q * 16 + 15) * 32);
#pragma unroll
          for (int i = 0; i < 16; ++i) acc += v[i];                          // ascending SNP order: the reference's order (:457-460)
        }
      } else {                                     // a barcode ends inside this tile: each chain counts its own pairs
        const uint32_t done = tile * T;
        const int cnt = np_b >= done + T ? T : (np_b > done ? (int)(np_b - done) : 0);
        const uint8_t* rp = row + (size_t)done;
        for (int i = 0; i < cnt; ++i) acc += *(LdsD)(uintptr_t)(base + (uint32_t)rp[i] + (uint32_t)i * 32u);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) cw[i] = nw[i];
    }
    if (valid) {
      if (kk < 8) { if (kk < V) llks[(size_t)cell * V + kk] = acc; }
      else llk0s[cell] = acc;
    }
    return;
  }

  // -------------------------------------------------------------------------------------------------- a producer: barcode slot0 + w
  const bool cell_ok = slot0 + w < pv.B;
  if (!cell_ok) { for (uint32_t tile = 0; tile < ntiles; ++tile) DMX_CP_BARRIER(); return; }
  const int32_t cell = sched[slot0 + w];
  const uint32_t p_beg = (uint32_t)pv.cell_pair_off[cell];                  // (the launcher checked: every offset fits 32 bits)
  const uint32_t np = s_np[w];
  uint32_t rd_base = (uint32_t)pv.cell_read_off[cell];
  const uint8_t* __restrict__ nrd8 = (const uint8_t*)pv.pair_nrd;
  const uint8_t* __restrict__ reads = pv.reads;
  const uint4* __restrict__ recA = snprec;
  const uint4* __restrict__ recB = snprec + pv.S;
  const uint32_t ct_a = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) double*)s_ct;
  // stages A - C: k_singlet_can's three-deep pipeline with one barcode per wavefront (T = 64) — tile t + 3's read counts, tile t + 2's scan and leading
  // read bytes, tile t + 1's table entry and SNP record are in flight while tile t computes
  struct Raw { uint32_t n; uint32_t snp; };
  auto stage_a = [&](uint32_t tile) {
    Raw r;
    const uint32_t pi = tile * T + lane;
    const bool v = pi < np;
    r.n = v ? (uint32_t)nrd8[p_beg + pi] : 0u;
    r.snp = v ? pi : 0u;                           // dense: the pair's SNP is its index
    return r;
  };
  struct Hdr { uint32_t n, snp, rd4, off; };
  auto stage_b = [&](const Raw& r) {
    Hdr h;
    h.n = r.n; h.snp = r.snp;
    const uint32_t incl = seg_scan_incl<T>(r.n);
    h.off = rd_base + (incl - r.n);
    rd_base += seg_last<T>(incl, lane);
    uint32_t v;
    __builtin_memcpy(&v, reads + (r.n ? h.off : 0u), 4);        // (the launcher checked: four bytes past the last read are readable)
    h.rd4 = v;
    return h;
  };
  struct Seed { double2 a, b, cc; double q0, q1, q2; uint32_t lds; bool fast; };   // lds: byte offset of the entry in the LDS copy, or ~0u
  auto stage_c = [&](const Hdr& h) {
    Seed sd;
    const uint32_t n = h.n, rd4 = h.rd4;
    sd.fast = true;
    sd.lds = (n == 0 ? (uint32_t)(2 * kCtBq) : (((rd4 & 0x80u) ? (uint32_t)kCtBq : 0u) + (rd4 & 0x7Fu))) * 48u;
    if (!(n == 0 || (n == 1 && (rd4 & 0x7Fu) < (uint32_t)kCtBq))) {
      sd.lds = ~0u;
      uint32_t idx = rd4 & 0xFFu;
      idx = n == 2 ? kCt2 + (rd4 & 0xFFFFu) : idx;
      bool fast = n <= 2;
      if (n == 3) {
        const uint32_t q0 = rd4 & 127u, q1 = (rd4 >> 8) & 127u, q2 = (rd4 >> 16) & 127u;
        if (max(max(q0, q1), q2) < (uint32_t)dmx::kTripleBq) {
          const uint32_t c0 = ((rd4 & 0x80u) ? (uint32_t)dmx::kTripleBq : 0u) + q0, c1 = ((rd4 & 0x8000u) ? (uint32_t)dmx::kTripleBq : 0u) + q1,
                         c2 = ((rd4 & 0x800000u) ? (uint32_t)dmx::kTripleBq : 0u) + q2;
          idx = kCt3 + __umul24(__umul24(c0, (uint32_t)dmx::kTripleCodes) + c1, (uint32_t)dmx::kTripleCodes) + c2;
          fast = true;
        }
      }
      sd.fast = fast;
      const double2* e = reinterpret_cast<const double2*>(ctab + 8u * idx);
      sd.a = e[0]; sd.b = e[1]; sd.cc = e[2];
    }
    const uint4 r0 = recA[h.snp], r1 = recB[h.snp];
    sd.q0 = __hiloint2double((int)r0.y, (int)r0.x); sd.q1 = __hiloint2double((int)r0.w, (int)r0.z);
    sd.q2 = __hiloint2double((int)r1.y, (int)r1.x);
    return sd;
  };
  // stage D: one tile -> the ring.  Lanes beyond the barcode's last pair carry n = 0, SNP 0 (stage A): they compute like any other lane, into records
  // the chains never reach.
  auto compute = [&](const Hdr& cur, Seed& cs, uint32_t tile) {
    if (cs.lds != ~0u) {                           // the LDS copy's entry
      LdsD2 p = (LdsD2)(uintptr_t)(ct_a + cs.lds);
      const v2d_t va = p[0], vb = p[1], vc = p[2];
      cs.a.x = va.x; cs.a.y = va.y; cs.b.x = vb.x; cs.b.y = vb.y; cs.cc.x = vc.x; cs.cc.y = vc.y;
    }
    double G0 = cs.a.x, G1 = cs.a.y, G2 = cs.b.x, t0 = cs.b.y, t1 = cs.cc.x, t2 = cs.cc.y;
    if (!cs.fast) {                                // deeper pairs, qualities beyond the tables: the read loop and the three logs, as k_singlet_can
      const GlSeed sd = gl_seed(tabs, cur.n, cur.rd4);
      gl_finish(sd, cur.n, cur.rd4, reads, (int64_t)cur.off, tabs, G0, G1, G2);
      t0 = dmx_log_fast(G0 * chi + G1 * clo + G2 * clo, s_log);
      t1 = dmx_log_fast(G0 * clo + G1 * chi + G2 * clo, s_log);
      t2 = dmx_log_fast(G0 * clo + G1 * clo + G2 * chi, s_log);
    }
    const double l0 = dmx_log_fast(G0 * cs.q0 + G1 * cs.q1 + G2 * cs.q2, s_log);                                  // llk0 (:459)
    v2d_t* o = reinterpret_cast<v2d_t*>(&s_ring[tile & 1u][w][lane][0]);
    v2d_t r01, r23;
    r01.x = t0; r01.y = t1; r23.x = t2; r23.y = l0;
    o[0] = r01; o[1] = r23;
    DMX_CP_BARRIER();
  };
  Hdr h1 = stage_b(stage_a(0));
  Hdr h2 = stage_b(stage_a(1));
  Raw pre = stage_a(2);
  Seed sA = stage_c(h1), sB = sA;
  for (uint32_t tile = 0; tile < ntiles; tile += 2) {
    {
      const Hdr cur = h1;
      h1 = h2;
      sB = stage_c(h1);                            // tile + 1
      h2 = stage_b(pre);                           // tile + 2
      pre = stage_a(tile + 3);                     // tile + 3
      compute(cur, sA, tile);
    }
    if (tile + 1 >= ntiles) break;
    {
      const Hdr cur = h1;
      h1 = h2;
      sA = stage_c(h1);
      h2 = stage_b(pre);
      pre = stage_a(tile + 4);
      compute(cur, sB, tile + 1);
    }
  }
}

// K1 over genotype classes, wide-panel form (V >= 20, measured crossover): all V+1 accumulators of a cell are summed in one pass per tile.
// Same walk and ownership as k_singlet; per pair the lane evaluates log(GL . row_d) once per class d plus the llk0 term,
// and stores those five terms and the SNP's packed class ids; chain lane (cell, k) then adds term[id[snp][k]] for the
// tile's pairs in ascending order — the very doubles k_singlet would have formed for sample k, in the same order.
// One barcode per wavefront (CW == 1).  L0M (panels of a multiple of 64 samples, cfg4's 64): the llk0 chain would be the ONLY chain of a second
// pass over the tile — a whole pass of index arithmetic and look-ups for one active lane, a third of the kernel's instructions at V = 64 — so
// it is merged into the first pass instead: every lane adds the tile's llk0 terms (uniform LDS reads) beside its own sample's, lane 0 stores.
template <int CW, int MINW = 4, bool L0M = false>
__global__ __launch_bounds__(kThreads, MINW) void k_singlet_clsw(PileupView pv, int nrd_width, const float* __restrict__ rows,
                                                             const uint32_t* __restrict__ idw, const double* __restrict__ gp0,
                                                             const double* __restrict__ tabs,
                                                             const int32_t* __restrict__ sched, int32_t V,
                                                             double* __restrict__ llks, double* __restrict__ llk0s) {
  constexpr int T = 64 / CW;
  constexpr int NW = kThreads / 64;
  constexpr int TD = 6;                          // doubles per pair in LDS: 4 class terms, llk0 term, 1 pad
  extern __shared__ __attribute__((aligned(16))) unsigned char s_dynraw[];   // per wavefront: id words [nwd][64]
  __shared__ double s_tab[kTabK1];
  __shared__ __attribute__((aligned(16))) double s_term[NW][64 * TD];
  __shared__ __attribute__((aligned(16))) double s_t0[NW][64];      // L0M: the tile's llk0 terms, contiguous (the merged sum reads them two at a time)
  const double* s_log = s_tab + kLut;
  const double* s_first = s_tab + kTab;
  const double* s_final = s_first + kFirst;

  const int t = threadIdx.x, w = t >> 6, lane = t & 63;
  const int nwd = (V + 15) / 16;
  for (int i = t; i < kTabK1; i += kThreads) s_tab[i] = tabs[i];
  __syncthreads();                               // the only workgroup barrier

  double* term = s_term[w];
  double* t0s = s_t0[w];
  uint32_t* s_idw = reinterpret_cast<uint32_t*>(s_dynraw) + (size_t)w * 64 * nwd;
  const int slot0 = (blockIdx.x * NW + w) * CW;
  if (slot0 >= pv.B) return;

  const int c = lane / T, ti = lane % T;
  const bool cell_ok = slot0 + c < pv.B;
  const int32_t cell = cell_ok ? sched[slot0 + c] : 0;
  const int64_t p_beg = cell_ok ? pv.cell_pair_off[cell] : 0;
  const int64_t np = cell_ok ? pv.cell_pair_off[cell + 1] - p_beg : 0;
  int64_t rd_base = cell_ok ? pv.cell_read_off[cell] : 0;
  int64_t max_np = np;
#pragma unroll
  for (int d = T; d < 64; d <<= 1) max_np = max(max_np, __shfl_xor(max_np, d));

  // chains: a = lane + 64*i over the barcode's accumulators, sample a for a < V; a == V is llk0 unless it is merged (L0M).  A wavefront
  // carries 256 chains; wider panels are cut into chain slabs over blockIdx.y, each repeating the five logs per pair.
  static_assert(CW == 1, "one barcode per wavefront");
  constexpr int MAXCH = 4;
  const int nchain = L0M ? V : V + 1;
  double acc[MAXCH];
  [[maybe_unused]] double acc0 = 0.0;            // L0M: the llk0 sum, on every lane of chain slab 0
  int32_t ch_k[MAXCH];
#pragma unroll
  for (int i = 0; i < MAXCH; ++i) {
    const int a = (int)blockIdx.y * 64 * MAXCH + lane + 64 * i;
    acc[i] = 0.0;
    ch_k[i] = (cell_ok && a < nchain) ? a : -1;
  }
  const bool slab0 = blockIdx.y == 0;

  struct Raw { uint32_t n; int32_t snp; };
  struct Hdr { uint32_t n; int32_t snp; uint32_t rd4; int64_t off; };
  auto issue = [&](int64_t tile) {
    Raw r;
    const int64_t pi = tile * T + ti;
    const bool v = pi < np;
    r.n = v ? load_nrd(pv.pair_nrd, p_beg + pi, nrd_width) : 0u;
    r.snp = v ? (pv.pair_snp ? pv.pair_snp[p_beg + pi] : (int32_t)pi) : 0;
    return r;
  };
  auto prepare = [&](const Raw& r) {
    Hdr h;
    h.n = r.n; h.snp = r.snp;
    const uint32_t incl = seg_scan_incl<T>(r.n);
    h.off = rd_base + (int64_t)(incl - r.n);
    rd_base += seg_last<T>(incl, lane);
    h.rd4 = 0;
    if (r.n > 0) {
      if (h.off + 4 <= pv.R) __builtin_memcpy(&h.rd4, pv.reads + h.off, 4);
      else for (int64_t i = h.off; i < pv.R; ++i) h.rd4 |= (uint32_t)pv.reads[i] << (8 * (int)(i - h.off));
    }
    return h;
  };

  Hdr nxt = prepare(issue(0));
  Raw pre = issue(1);
  for (int64_t tile = 0; tile * T < max_np; ++tile) {
    const Hdr cur = nxt;
    nxt = prepare(pre);
    pre = issue(tile + 2);
    const bool valid = tile * T + ti < np;

    // class rows, llk0 row and id words of this lane's SNP (48 + 24 + 4*nwd bytes, SNP-major: contiguous across a dense tile)
    const float4* rp = reinterpret_cast<const float4*>(rows + (size_t)cur.snp * 12);
    const float4 r0 = rp[0], r1 = rp[1], r2 = rp[2];
    const double* g0 = gp0 + (size_t)cur.snp * 3;
    const double q0 = g0[0], q1 = g0[1], q2 = g0[2];
    for (int wq = 0; wq < nwd; ++wq) s_idw[wq * 64 + lane] = idw[(size_t)cur.snp * nwd + wq];   // [word][pair]: a chain reads four pairs' words at once

    double G0, G1, G2;                                                       // :427-452, as in k_singlet
    {
      const uint32_t n = cur.n;
      const uint32_t b0 = cur.rd4 & 0xFFu;
      const double* f = s_final + 3 * (n ? b0 : 256u);
      G0 = f[0]; G1 = f[1]; G2 = f[2];
      if (n >= 2) {
        const double* f1 = s_first + 3 * b0;
        double g0_ = f1[0], g1_ = f1[1], g2_ = f1[2];
        const bool safe = n <= kSafeReads;
        for (uint32_t r = 1; r < n; ++r) {
          const uint32_t byte = (r < 4) ? ((cur.rd4 >> (8 * r)) & 0xFFu) : (uint32_t)pv.reads[cur.off + r];
          const uint32_t bq = byte & 127u;
          const bool alt = (byte >> 7) != 0;
          const double m = s_tab[bq], e3 = s_tab[128 + bq], h = s_tab[256 + bq];
          g0_ *= alt ? e3 : m;
          g1_ *= h;
          g2_ *= alt ? m : e3;
          const double tmp = g0_ + g1_ + g2_;
          if (safe) {
            const double y = rcp_refined(tmp);
            g0_ = div_by(g0_, tmp, y); g1_ = div_by(g1_, tmp, y); g2_ = div_by(g2_, tmp, y);
          } else {
            g0_ /= tmp; g1_ /= tmp; g2_ /= tmp;
          }
        }
        g0_ += 1e-6; g1_ += 1e-6; g2_ += 1e-6;
        const double tmp = g0_ + g1_ + g2_;
        const double y = rcp_refined(tmp);
        G0 = div_by(g0_, tmp, y); G1 = div_by(g1_, tmp, y); G2 = div_by(g2_, tmp, y);
      }
    }
    if (valid) {
      const double x0 = G0 * (double)r0.x + G1 * (double)r0.y + G2 * (double)r0.z;     // class 0   (:456)
      const double x1 = G0 * (double)r0.w + G1 * (double)r1.x + G2 * (double)r1.y;     // class 1
      const double x2 = G0 * (double)r1.z + G1 * (double)r1.w + G2 * (double)r2.x;     // class 2
      const double x3 = G0 * (double)r2.y + G1 * (double)r2.z + G2 * (double)r2.w;     // class 3
      const double x4 = G0 * q0 + G1 * q1 + G2 * q2;                                    // llk0      (:459)
      const bool fast_ok = __builtin_amdgcn_class(x0, 0x100) && __builtin_amdgcn_class(x1, 0x100) && __builtin_amdgcn_class(x2, 0x100) &&
                           __builtin_amdgcn_class(x3, 0x100) && __builtin_amdgcn_class(x4, 0x100);
      double* tr = &term[lane * TD];
      tr[0] = dmx_log_fast(x0, s_log); tr[1] = dmx_log_fast(x1, s_log); tr[2] = dmx_log_fast(x2, s_log);
      tr[3] = dmx_log_fast(x3, s_log); tr[4] = dmx_log_fast(x4, s_log);
      if (__builtin_expect(!fast_ok, 0)) {         // never for real likelihoods; keeps log(0) / log(nan) semantics
        const double xs[5] = {x0, x1, x2, x3, x4};
        for (int d = 0; d < 5; ++d) if (!__builtin_amdgcn_class(xs[d], 0x100)) tr[d] = log(xs[d]);
      }
      if (L0M) t0s[lane] = tr[4];
    }
    DMX_WAVE_LDS_ORDER();
    // ---- ordered sums: chain k adds term[pair][class of sample k at the pair's SNP] for the tile's pairs
    const int64_t left = np - tile * T;            // (np is the wavefront's one barcode's: uniform)
    const int cnt = left >= T ? T : (left > 0 ? (int)left : 0);
#pragma unroll
    for (int i = 0; i < MAXCH; ++i) {
      const bool merged = L0M && i == 0 && slab0;  // this pass also carries the llk0 sum
      if (ch_k[i] < 0 && !merged) continue;
      const int k = ch_k[i] < 0 ? 0 : ch_k[i];     // (merged pass: lanes without a chain walk along with sample 0's look-ups, unused)
      const bool is0 = !L0M && k == V;             // (with L0M no chain is the llk0 chain: the test and its select compile away)
      const int wq = is0 ? 0 : (k >> 4), sh = is0 ? 0 : 2 * (k & 15);
      const double* tb = &term[0];
      const uint32_t* ib = &s_idw[wq * 64];
      // Round 5: the walk is instantiated with and without the merged llk0 sum (it used to test `merged` — a scalar branch — at every pair), the id
      // words of eight pairs come as two 16-byte reads from the transposed [word][pair] layout (they were eight reads through eight separately
      // advanced addresses): 10 -> 4 VALU instructions per (pair, chain) — extract, scale-and-add, the two adds.
      auto walk = [&](auto merged_c) {
        constexpr bool MG = decltype(merged_c)::value;
        double s = acc[i];
        int p = 0;
        constexpr int NB = L0M ? 8 : 16;           // pairs per batch: ids first, then the term reads, then the ordered adds
        for (; p + NB <= cnt; p += NB) {
          uint32_t wv[NB];
#pragma unroll
          for (int q4 = 0; q4 < NB / 4; ++q4) {
            const uint4 u = *reinterpret_cast<const uint4*>(&ib[p + 4 * q4]);
            wv[4 * q4] = u.x; wv[4 * q4 + 1] = u.y; wv[4 * q4 + 2] = u.z; wv[4 * q4 + 3] = u.w;
          }
          double tv[NB];
          [[maybe_unused]] double t0v[NB];
#pragma unroll
          for (int q = 0; q < NB; ++q) {
            const uint32_t d = is0 ? 4u : ((wv[q] >> sh) & 3u);
            tv[q] = tb[(p + q) * TD + d];
          }
          if (MG) {                                  // (uniform addresses; taking the terms from their pairs' lanes with v_readlane instead: 28.6 -> 34.1 ms)
#pragma unroll
            for (int q = 0; q < NB; q += 2) {
              const double2 u = *reinterpret_cast<const double2*>(&t0s[p + q]);
              t0v[q] = u.x; t0v[q + 1] = u.y;
            }
          }
#pragma unroll
          for (int q = 0; q < NB; ++q) {             // ascending SNP order: the reference's order
            s += tv[q];
            if (MG) acc0 += t0v[q];
          }
        }
        for (; p < cnt; ++p) {
          const uint32_t d = is0 ? 4u : ((ib[p] >> sh) & 3u);
          s += tb[p * TD + d];
          if (MG) acc0 += tb[p * TD + 4];
        }
        acc[i] = s;
      };
      if (merged) walk(std::true_type{}); else walk(std::false_type{});
    }
    DMX_WAVE_LDS_ORDER();
  }
#pragma unroll
  for (int i = 0; i < MAXCH; ++i) {
    if (ch_k[i] < 0) continue;
    if (ch_k[i] < V) llks[(size_t)cell * V + ch_k[i]] = acc[i];
    else llk0s[cell] = acc[i];
  }
  if (L0M && slab0 && cell_ok && lane == 0) llk0s[cell] = acc0;
}

// Where every cell stands at the boundaries of the SNP blocks [b << shift, (b + 1) << shift): blk[cell][b] = {index of its first
// pair with SNP id >= b << shift, first read byte of that pair}, b = 0..nblk (the last entry is the cell's end).  One wavefront per
// cell, one pass over its pair headers; built once per staged pileup.
__global__ __launch_bounds__(kThreads) void k_snp_blocks(PileupView pv, int nrd_width, int shift, int32_t nblk, int64_t* __restrict__ blk) {
  const int lane = threadIdx.x & 63;
  const int32_t cell = blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6);
  if (cell >= pv.B) return;
  const int64_t p_beg = pv.cell_pair_off[cell], p_end = pv.cell_pair_off[cell + 1];
  int64_t rd_base = pv.cell_read_off[cell];
  int64_t* out = blk + (size_t)cell * (nblk + 1) * 2;
  int32_t b_prev = -1;                            // block of the last pair seen (wave-uniform)
  for (int64_t p0 = p_beg; p0 < p_end; p0 += 64) {
    const int64_t p = p0 + lane;
    const bool v = p < p_end;
    const uint32_t n = v ? load_nrd(pv.pair_nrd, p, nrd_width) : 0u;
    const int32_t b = v ? (pv.pair_snp[p] >> shift) : 0x7FFFFFFF;
    const uint32_t incl = seg_scan_incl<64>(n);
    const int64_t off = rd_base + (int64_t)(incl - n);
    int32_t bp = __shfl_up(b, 1);
    if (lane == 0) bp = b_prev;
    if (v) for (int32_t bb = bp + 1; bb <= min(b, nblk); ++bb) { out[2 * bb] = p; out[2 * bb + 1] = off; }
    rd_base += (int64_t)__builtin_amdgcn_readlane((int)incl, 63);
    const int last = (int)min((int64_t)63, p_end - p0 - 1);
    b_prev = __shfl(b, last);
  }
  for (int32_t bb = b_prev + 1 + lane; bb <= nblk; bb += 64) { out[2 * bb] = p_end; out[2 * bb + 1] = rd_base; }
}

// SNP-minor copies for dense pileups: gT[r][s] = g[s][r] (r = k*3+l, float32 as stored) and g0T[l][s] = gp0s[s][l].
__global__ void k_transpose_geno(const float* __restrict__ g, const double* __restrict__ gp0, int32_t S, int32_t V,
                                 float* __restrict__ gT, double* __restrict__ g0T) {
  const int nrow = V * 3;
  const int64_t n = (int64_t)S * nrow;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t s = i % S;
    const int r = (int)(i / S);
    gT[i] = g[(size_t)s * nrow + r];
    if (r < 3) g0T[i] = gp0[(size_t)s * 3 + r];
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// K2, generic form (any V, any A).  One cell per workgroup.  Accumulator q of the cell (q < V*V*A: (j,k,n) row-major,
// then A entries for llks00) lives in lane q % 256, register q / 256; grids with more than NACC*256 accumulators are cut into
// slabs of that many, one workgroup (blockIdx.y) per slab, each repeating the cheap phase 1 for itself.
//   phase 1  lane (pair ti, alpha n): the 9 mixture likelihoods pG[n][l][m] of that pair, reads in UMI order, with the
//            one-max-across-all-alphas renormalisation after every read (:600-663); A is padded to a power of two so the
//            A lanes of a pair sit together in a wavefront and share their max by butterfly shuffles;
//   phase 2  every accumulator adds log(sum_lm ...) for the tile's pairs in ascending SNP order (:671-709).
template <typename NRD, int NACC, bool FIXUP>
__global__ __launch_bounds__(kThreads) void k_doublet_generic(PileupView pv, const float* __restrict__ g,
                                                              const double* __restrict__ gp0,
                                                              const double* __restrict__ lut,
                                                              const double* __restrict__ alpha,
                                                              const int32_t* __restrict__ sched, int32_t V, int32_t A,
                                                              int32_t A_pad, int32_t TP, double* __restrict__ grid,
                                                              double* __restrict__ l00,
                                                              uint8_t* __restrict__ flagged) {
  // FIXUP: second pass behind the fast kernels — only cells in which one of them met a log() argument outside the normal
  // positive range (flagged[cell] != 0) are recomputed, with ocml's log() for exact log(0) / log(nan) semantics.  The pass is
  // launched with a small fixed grid whose workgroups stride over the cells, and it ends at once when no cell of the launch was
  // flagged at all (flag_cell also raises the launch-wide word in front of the array): nothing to fix costs ~2 us instead of a
  // dispatch of B workgroups.  The first pass (FIXUP = false) is launched with one workgroup per cell: one trip through the loop.
  if (FIXUP && !flags_any(flagged)) return;
  __shared__ double s_lut[kTab2];
  __shared__ double s_pG[kThreads * 9];
  __shared__ int32_t s_snp[32];
  __shared__ uint32_t s_cnt[32];
  __shared__ int64_t s_off[32];

  const int t = threadIdx.x;
  stage_k2_tables(s_lut, lut, t, kThreads);
  const double* s_log = s_lut + kLut2;
  for (int32_t bx = (int32_t)blockIdx.x; bx < pv.B; bx += (int32_t)gridDim.x) {
  if (FIXUP && !flagged[sched[bx]]) continue;
  const int32_t cell = sched[bx];
  const int64_t p_beg = pv.cell_pair_off[cell];
  const int64_t np = pv.cell_pair_off[cell + 1] - p_beg;
  int64_t rd_base = pv.cell_read_off[cell];
  const NRD* __restrict__ nrd = (const NRD*)pv.pair_nrd;

  // phase-1 identity and the per-(l,m) mixing weights of this lane's alpha (:613)
  const int ti1 = t / A_pad, n1 = t % A_pad;
  const bool lane1 = (ti1 < TP) && (n1 < A);
  double wA[9], wR[9];
  {
    const double al = lane1 ? alpha[n1] : 0.0;
#pragma unroll
    for (int l = 0; l < 3; ++l)
#pragma unroll
      for (int m = 0; m < 3; ++m) {
        const double p = 0.5 * l + (m - l) * 0.5 * al;
        wA[l * 3 + m] = p;
        wR[l * 3 + m] = 1.0 - p;
      }
  }

  // phase-2 identity
  const int32_t nAB = V * V * A;
  const int32_t nacc = nAB + A;
  const int32_t q0 = (int32_t)blockIdx.y * NACC * kThreads;      // first accumulator of this workgroup's slab
  bool ok = true;              // every log() argument so far was a normal positive double (the fast path's domain)
  uint32_t code[NACC];         // (j << 20) | (k << 8) | n ; j = 0xFFF marks an llks00 entry; ~0u = no accumulator
  double acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) {
    const int32_t q = q0 + t + i * kThreads;
    acc[i] = 0.0;
    if (q < nAB) {
      const int32_t n = q % A, jk = q / A;
      code[i] = ((uint32_t)(jk / V) << 20) | ((uint32_t)(jk % V) << 8) | (uint32_t)n;
    } else if (q < nacc) {
      code[i] = (0xFFFu << 20) | (uint32_t)(q - nAB);
    } else {
      code[i] = ~0u;
    }
  }
  __syncthreads();

  for (int64_t base = 0; base < np; base += TP) {
    const int tp = (int)min((int64_t)TP, np - base);
    // pair headers of the tile
    if (t < 32) {
      const bool v = t < tp;
      const uint32_t n = v ? (uint32_t)nrd[p_beg + base + t] : 0u;
      uint32_t incl = n;
#pragma unroll
      for (int d = 1; d < 32; d <<= 1) {
        const uint32_t y = __shfl_up(incl, d, 32);
        if (t >= d) incl += y;
      }
      if (v) {
        s_cnt[t] = n;
        s_off[t] = rd_base + (int64_t)(incl - n);
        s_snp[t] = pv.pair_snp ? pv.pair_snp[p_beg + base + t] : (int32_t)(base + t);
      }
    }
    __syncthreads();
    rd_base = s_off[tp - 1] + (int64_t)s_cnt[tp - 1];     // first read byte of the next tile, for every lane

    // ---- phase 1
    {
      const bool on = lane1 && ti1 < tp;
      const uint32_t cnt = on ? s_cnt[ti1] : 0u;
      const int64_t off = on ? s_off[ti1] : 0;
      const uint32_t rd4 = load_rd4(pv, off, cnt);       // the first four read bytes in one load (one dependent latency instead of four)
      double pG[9];
#pragma unroll
      for (int i = 0; i < 9; ++i) pG[i] = 1.0;                              // :597
      for (uint32_t r = 0; __any(r < cnt); ++r) {
        const bool live = r < cnt;
        const uint32_t byte = live ? (r < 4 ? (rd4 >> (8 * r)) & 0xFFu : (uint32_t)pv.reads[off + r]) : 0u;
        const uint32_t bq = byte & 127u;
        const bool alt = (byte >> 7) != 0;
        const double pR = alt ? s_lut[128 + bq] : s_lut[bq];               // :606
        const double pA = alt ? s_lut[bq] : s_lut[128 + bq];               // :607
        double mx = 0.0;
        if (live) {
#pragma unroll
          for (int i = 0; i < 9; ++i) {
            pG[i] *= (pR * wR[i] + pA * wA[i]);                            // :625
            mx = fmax(mx, pG[i]);                                // :626-627
          }
        }
        for (int d = 1; d < A_pad; d <<= 1) {                              // one max across ALL alphas of the pair
          const double o = __shfl_xor(mx, d);
          mx = fmax(mx, o);
        }
        if (live) {
          if (cnt <= kSafeReads) {
            const double y = rcp_refined(mx);
#pragma unroll
            for (int i = 0; i < 9; ++i) pG[i] = div_by(pG[i], mx, y);      // :632-639
          } else {
#pragma unroll
            for (int i = 0; i < 9; ++i) pG[i] /= mx;
          }
        }
      }
      double mx = 0.0;
      if (on) {
#pragma unroll
        for (int i = 0; i < 9; ++i) {
          pG[i] += 1e-6;                                                    // :649
          mx = fmax(mx, pG[i]);
        }
      }
      for (int d = 1; d < A_pad; d <<= 1) {
        const double o = __shfl_xor(mx, d);
        mx = fmax(mx, o);
      }
      if (on) {
        const double y = rcp_refined(mx);                                   // numerators >= 1e-6, mx in [1e-6, 1+1e-6]
#pragma unroll
        for (int i = 0; i < 9; ++i) s_pG[(ti1 * A + n1) * 9 + i] = div_by(pG[i], mx, y);   // :656-663
      }
    }
    __syncthreads();

    // ---- phase 2
    for (int ti = 0; ti < tp; ++ti) {
      const int32_t snp = s_snp[ti];
      const float* __restrict__ grow = g + (size_t)snp * V * 3;
      const double* __restrict__ g0 = gp0 + (size_t)snp * 3;
#pragma unroll
      for (int i = 0; i < NACC; ++i) {
        const uint32_t cd = code[i];
        if (cd == ~0u) continue;
        const int32_t j = (cd >> 20) & 0xFFF, k = (cd >> 8) & 0xFFF, n = cd & 0xFF;
        double a[3], b[3];
        if (j == 0xFFF) {
          a[0] = b[0] = g0[0]; a[1] = b[1] = g0[1]; a[2] = b[2] = g0[2];   // gp00 = gp0s[l]*gp0s[m]  (:555)
        } else {
          a[0] = (double)grow[j * 3]; a[1] = (double)grow[j * 3 + 1]; a[2] = (double)grow[j * 3 + 2];
          b[0] = (double)grow[k * 3]; b[1] = (double)grow[k * 3 + 1]; b[2] = (double)grow[k * 3 + 2];
        }
        const double* P = &s_pG[(ti * A + n) * 9];
        double sum = 0.0;                                                   // :674 std::fill(...,0)
#pragma unroll
        for (int l = 0; l < 3; ++l)
#pragma unroll
          for (int m = 0; m < 3; ++m) sum += ((a[l] * b[m]) * P[l * 3 + m]);   // :553 then :677-679, l-major
        if (!FIXUP) ok &= __builtin_amdgcn_class(sum, 0x100);
        acc[i] += FIXUP ? log(sum) : dmx_log2_fast(sum, s_log);                     // :683 / :709
      }
    }
    __syncthreads();
  }

#pragma unroll
  for (int i = 0; i < NACC; ++i) {
    const int32_t q = q0 + t + i * kThreads;
    if (q < nAB) grid[(size_t)cell * nAB + q] = acc[i];
    else if (q < nacc) l00[(size_t)cell * A + (q - nAB)] = acc[i];
  }
  if (!FIXUP && !ok) flag_cell(flagged, cell);   // recomputed with ocml's log() by the FIXUP pass
  __syncthreads();                               // the next cell of this workgroup reuses the shared arrays
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// K2 for the default alpha grid size A = 2 (alpha[0] = singlet entry, alpha[1] = the doublet mixture).
// A cell is owned by TPC threads: TPC = 64 (V <= 16: one wavefront per cell, four independent cells per workgroup, no
// workgroup barrier in the loop) or TPC = 256 (V <= 64: one cell per workgroup).  Thread (j, kb) of the cell owns the
// 2*NK accumulators llksAB[j][kb*NK .. kb*NK+NK-1][0..1] in registers for the whole SNP range; it adds their log terms
// pair after pair, i.e. in the reference's ascending-SNP order.  Per tile of 32 covered pairs:
//   stage   the pairs' headers, then their genotype rows (float32, coalesced) into LDS;
//   phase 1 lane (pair ti, alpha n) of the cell's first wavefront: pG[n][3][3] with the shared-max renormalisation after
//           every read (:600-663) -> LDS; the same lane also forms the llks00 term of its (pair, alpha) (:699-709);
//   phase 2 every thread, for each pair in order: gpAB[l][m] = g_j[l]*g_k[m] (exact: float32 x float32 in binary64,
//           :553), the nine-term l-major sums for both alphas (:677-679), log, add (:683).
// log() arguments outside the normal positive range cannot occur for genuine likelihoods; if one does, the cell is
// flagged and recomputed by k_doublet_generic<FIXUP> with ocml's log().
// GD: the genotype rows are widened to binary64 once, when they are staged (a conversion per element and tile instead of one per
// use in phase 2); the LDS holds them as doubles.
// CHK = false drops the per-term argument-class test of phase 2 (a v_cmp_class_f64 per log): launched only when every genotype row was
// found finite, non-negative and not vanishing (k_check_geno) — then every phase-2 sum is >= max_l g_j[l] * (1e-6 / (1 + 1e-6)) *
// max_m g_k[m] > 2^-830, a normal positive number, and the test cannot fire.
// TP: covered pairs per tile (32; 16 or 8 on panels of more than ~180 samples, whose 32 staged genotype rows would leave one workgroup per CU)
// (the 64-thread-cell forms run three workgroups of four cells per CU = three wavefronts per SIMD: bounded to their 168 registers — left unbounded the
//  compiler's count moved from 158 to 246 with an unrelated edit of phase 1 and cfg5 STRICT lost a wavefront per SIMD, 144 -> 168 ms)
// SYMU (round 6, k_doublet_a2u below; V = 32 on 256 threads or 16 on 64, default grid): phase 2 over UNORDERED pairs — thread t owns the pairs {j, (j + d) % V} of units u = t
// and t + TPC (d = 1 + u / V, j = u % V; d = V/2 from j < V/2 only: 496 of 512 slots at V = 32, 120 of 128 at 16) with the four accumulators [j][k][0..1],
// [k][j][0..1] each: eight per thread, as before.
template <int TPC, int NK, int MINW, bool GD, bool CHK, int TP, bool SYMU>
__device__ __forceinline__ void a2_body(PileupView pv, int nrd_width, const float* __restrict__ g,
                                                         const double* __restrict__ gp0, const double* __restrict__ tabs,
                                                         const double* __restrict__ alpha,
                                                         const int32_t* __restrict__ sched, int32_t V, int32_t GS,
                                                         double* __restrict__ grid, double* __restrict__ l00,
                                                         uint8_t* __restrict__ flagged, const double* __restrict__ pfin, const double* __restrict__ pseed) {
  static_assert(!SYMU || ((TPC == 256 || TPC == 64) && NK == 4 && !CHK), "symmetric ownership: 32 samples on 256 threads or 16 on 64, eight accumulators per thread, provably safe rows");
  constexpr int kUV = TPC == 256 ? 32 : 16, kUSh = kUV == 32 ? 5 : 4;       // SYMU: the panel (V (V - 1) / 2 unordered pairs on 2 TPC slots) and log2 of it
  // pfin (round 6; NULL unless the grid is {0, 0.5} and the pileup shallow): k_build_certify_finals' table of finished phase-1 values;
  // pseed (NULL unless the grid is {0, 0.5}): k_build_certify_seeds' table — on that grid the tiles that walk the read loop walk it in the five-value form
  // (alpha 0.5's five distinct mixing weights, alpha 0's three: entries of equal weight go through identical operations, so the nine values are these,
  // repeated) from the state after a pair's first one or two reads
  constexpr int A = 2;
  static_assert(TP == 32 || TP == 16 || TP == 8, "phase 1 runs on the first 2 * TP lanes of the cell's first wavefront");
  constexpr int CPW = kThreads / TPC;            // cells per workgroup
  constexpr int T00 = TP + 2;
#define DMX_K2_SYNC() do { if (TPC == 64) { DMX_WAVE_LDS_ORDER(); } else { __syncthreads(); } } while (0)
  extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
  __shared__ double s_tab[kTab2];
  __shared__ double s_w[2][18];                  // mixing weights of :613 per alpha: [n][0..8] = p (ALT), [n][9..17] = 1 - p; in LDS so
                                                 // that they occupy registers only while phase 1 runs (36 VGPRs otherwise)
  const double* s_log = s_tab + kLut2;
  const int t = threadIdx.x;
  stage_k2_tables(s_tab, tabs, t, kThreads);
  if (t < 18) {
    const int n = t / 9, l = (t % 9) / 3, m = t % 3;
    const double p = 0.5 * l + (m - l) * 0.5 * alpha[n];
    s_w[n][t % 9] = p;
    s_w[n][9 + t % 9] = 1.0 - p;
  }
  __syncthreads();

  const int cw = t / TPC, tid = t % TPC;         // cell slot inside the workgroup, thread inside the cell
  // per-cell LDS regions
  using g_t = typename std::conditional<GD, double, float>::type;
  const size_t cell_bytes = (size_t)TP * 18 * 8 + (size_t)TP * GS * sizeof(g_t) + 2 * T00 * 8 + TP * (4 + 4 + 8);
  unsigned char* base = s_raw + (size_t)cw * cell_bytes;
  double* s_pG = (double*)base;                                  // [TP][2][9]
  g_t* s_g = (g_t*)(base + (size_t)TP * 18 * 8);                 // [TP][GS]   genotype rows of the tile's SNPs
  double* s_t00 = (double*)((unsigned char*)s_g + (size_t)TP * GS * sizeof(g_t));   // [2][T00]   llks00 terms
  int64_t* s_off = (int64_t*)(s_t00 + 2 * T00);                  // [TP]
  int32_t* s_snp = (int32_t*)(s_off + TP);                       // [TP]
  uint32_t* s_cnt = (uint32_t*)(s_snp + TP);                     // [TP]

  const int slot = blockIdx.x * CPW + cw;
  if (TPC == 64 && slot >= pv.B) return;         // whole wavefront idle (no workgroup barriers below in this mode)
  const bool cell_ok = slot < pv.B;
  const int32_t cell = cell_ok ? sched[slot] : 0;
  const int64_t p_beg = cell_ok ? pv.cell_pair_off[cell] : 0;
  const int64_t np = cell_ok ? pv.cell_pair_off[cell + 1] - p_beg : 0;
  int64_t rd_base = cell_ok ? pv.cell_read_off[cell] : 0;

  // phase-2 identity.  A workgroup covers JS = TPC / KB rows j of the cell's grid; panels with more rows than that are cut
  // into j-slabs, one workgroup (blockIdx.y) per slab, each repeating the cheap phases 0-1 for itself.
  const int KB = (V + NK - 1) / NK;              // k-blocks per j
  const int JS = TPC / KB;
  const int jl = tid / KB, kb = tid % KB;
  const int j = (int)blockIdx.y * JS + jl;
  const bool owner = jl < JS && j < V;
  double acc[NK][A];
#pragma unroll
  for (int kk = 0; kk < NK; ++kk) { acc[kk][0] = 0.0; acc[kk][1] = 0.0; }
  bool ok = true;
  const DmxLogPins lk = dmx_log_pins();
  int uj[2] = {0, 0}, uk[2] = {0, 0};            // SYMU: the thread's two unordered pairs; acc[2 su] = [j][k][0..1], acc[2 su + 1] = [k][j][0..1]
  if constexpr (SYMU) {
#pragma unroll
    for (int su = 0; su < 2; ++su) { const int u = tid + TPC * su; uj[su] = u & (kUV - 1); uk[su] = (uj[su] + 1 + (u >> kUSh)) & (kUV - 1); }
  }
  // phase-1 identity (first wavefront of the cell): pair ti1, alpha n1; mixing weights of :613
  const int ti1 = tid >> 1, n1 = tid & 1;
  double acc00 = 0.0;                            // lane n1 == tid < 2 owns llks00[n]
  const int row_len = V * 3;

  for (int64_t tbase = 0; tbase < np; tbase += TP) {
    const int tp = (int)min((int64_t)TP, np - tbase);
    // ---- headers of the tile's pairs (first 32 lanes of the cell)
    if (tid < TP) {
      const bool v = tid < tp;
      const uint32_t n = v ? load_nrd(pv.pair_nrd, p_beg + tbase + tid, nrd_width) : 0u;
      const uint32_t incl = seg_scan_incl<TP>(n);
      s_cnt[tid] = n;
      s_off[tid] = rd_base + (int64_t)(incl - n);
      s_snp[tid] = v ? (pv.pair_snp ? pv.pair_snp[p_beg + tbase + tid] : (int32_t)(tbase + tid)) : 0;
    }
    DMX_K2_SYNC();
    rd_base = s_off[tp - 1] + (int64_t)s_cnt[tp - 1];
    // ---- genotype rows -> LDS (coalesced along the row)
    {
      int r = tid % row_len, ti = tid / row_len;
      const int dr = TPC % row_len, dt = TPC / row_len;
      while (ti < tp) {
        s_g[ti * GS + r] = (g_t)g[(size_t)s_snp[ti] * row_len + r];
        r += dr; ti += dt;
        if (r >= row_len) { r -= row_len; ++ti; }
      }
    }
    // ---- phase 1
    if (tid < 2 * TP) {
      const bool on = ti1 < tp;
      const uint32_t cnt = on ? s_cnt[ti1] : 0u;
      const int64_t off = on ? s_off[ti1] : 0;
      const uint32_t rd4 = load_rd4(pv, off, cnt);       // the first four read bytes in one load (one dependent latency instead of four)
      // Round 6 (default grid {0, 0.5}): pairs of up to three tabled reads take their finished values from the table — alpha 0.5's five distinct ones
      // q[l + m], alpha 0's three q[l]: entries of equal mixing weight go through identical operations, so the nine of the loop below repeat them bit for
      // bit — and the loop and the +1e-6 renormalisation run only in tiles with a deeper pair (wave-uniform branch); see k_doublet_sym.
      // (a tile with a deeper pair runs the loop for ALL its lanes, as before: no merge of the two sources, no register held across the loop)
      double vf[9];
      const int32_t fi = pfin ? certify_final_index(cnt, rd4) : -1;
      if (!__any(fi < 0)) {
        const double* fp = pfin + (size_t)fi * kCFinStride + (n1 ? 0 : 5);
        const double f0 = fp[0], f1 = fp[1], f2 = fp[2], f3 = n1 ? fp[3] : 0.0, f4 = n1 ? fp[4] : 0.0;
        const double f5[5] = {f0, f1, f2, f3, f4};
#pragma unroll
        for (int l = 0; l < 3; ++l)
#pragma unroll
          for (int m = 0; m < 3; ++m) vf[l * 3 + m] = n1 ? f5[l + m] : f5[l];
      } else if (pseed) {
        double q5[5], wA5[5], wR5[5];
#pragma unroll
        for (int q = 0; q < 5; ++q) {                // the five-value form's weights: alpha 0.5: p = 0.25 (l + m), slot l + m; alpha 0: p = 0.5 l, slots 0..2 (3, 4 repeat 2)
          const int l = n1 ? (q > 2 ? 2 : q) : min(q, 2), m = n1 ? q - l : 0;
          const double p = 0.5 * l + (m - l) * 0.5 * (n1 ? 0.5 : 0.0);
          wA5[q] = p;
          wR5[q] = 1.0 - p;
        }
        certify_pair_values<5, false>(pv, cnt, off, rd4, s_tab, wA5, wR5, n1, q5, pseed);
#pragma unroll
        for (int l = 0; l < 3; ++l)
#pragma unroll
          for (int m = 0; m < 3; ++m) vf[l * 3 + m] = n1 ? q5[l + m] : q5[l];
      } else {
      double pG[9], wA[9], wR[9];
#pragma unroll
      for (int i = 0; i < 9; ++i) { pG[i] = 1.0; wA[i] = s_w[n1][i]; wR[i] = s_w[n1][9 + i]; }   // :597
      for (uint32_t r = 0; __any(r < cnt); ++r) {
        const bool live = r < cnt;
        const uint32_t byte = live ? (r < 4 ? (rd4 >> (8 * r)) & 0xFFu : (uint32_t)pv.reads[off + r]) : 0u;
        const uint32_t bq = byte & 127u;
        const bool alt = (byte >> 7) != 0;
        const double pR = alt ? s_tab[128 + bq] : s_tab[bq];                // :606
        const double pA = alt ? s_tab[bq] : s_tab[128 + bq];                // :607
        double mx = 0.0;
        if (live) {
#pragma unroll
          for (int i = 0; i < 9; ++i) {
            pG[i] *= (pR * wR[i] + pA * wA[i]);                             // :625
            mx = fmax(mx, pG[i]);                                 // :626-627
          }
        }
        {
          const double o = shfl_xor1(mx);                               // one max across both alphas of the pair
          mx = fmax(mx, o);
        }
        if (live) {
          if (cnt <= kSafeReads) {
            const double y = rcp_refined(mx);
#pragma unroll
            for (int i = 0; i < 9; ++i) pG[i] = div_by(pG[i], mx, y);       // :632-639
          } else {
#pragma unroll
            for (int i = 0; i < 9; ++i) pG[i] /= mx;
          }
        }
      }
      double mx = 0.0;
#pragma unroll
      for (int i = 0; i < 9; ++i) {
        pG[i] += 1e-6;                                                       // :649
        mx = fmax(mx, pG[i]);
      }
      {
        const double o = shfl_xor1(mx);
        mx = fmax(mx, o);
      }
      {
        const double y = rcp_refined(mx);                                    // numerators >= 1e-6, mx in [1e-6, 1+1e-6]
#pragma unroll
        for (int i = 0; i < 9; ++i) vf[i] = div_by(pG[i], mx, y);            // :656-663
      }
      }
      if (on) {
        const double* g0 = gp0 + (size_t)s_snp[ti1] * 3;
        const double q0 = g0[0], q1 = g0[1], q2 = g0[2];
        const double qq[3] = {q0, q1, q2};
        double sum = 0.0;
#pragma unroll
        for (int l = 0; l < 3; ++l)
#pragma unroll
          for (int m = 0; m < 3; ++m) {
            const double v = vf[l * 3 + m];
            s_pG[(ti1 * 2 + n1) * 9 + l * 3 + m] = v;
            sum += ((qq[l] * qq[m]) * v);                                    // gp00 (:555) then :702-705
          }
        ok &= __builtin_amdgcn_class(sum, 0x100);
        s_t00[n1 * T00 + ti1] = dmx_log2_fast(sum, s_log);                    // :708-709 term
      }
    }
    DMX_K2_SYNC();
    // ---- llks00: lane n < 2 of the cell adds its alpha's terms in pair order
    if (tid < 2) {
      const double* row = &s_t00[tid * T00];
      if (tp == TP) {                              // loads first (LDS latency paid once), then the ordered adds
        double2 v[TP / 2];
#pragma unroll
        for (int i = 0; i < TP / 2; ++i) v[i] = *reinterpret_cast<const double2*>(&row[2 * i]);
#pragma unroll
        for (int i = 0; i < TP / 2; ++i) { acc00 += v[i].x; acc00 += v[i].y; }
      } else {
        for (int i = 0; i < tp; ++i) acc00 += row[i];
      }
    }
    // ---- phase 2
    if constexpr (SYMU) {
      // At alpha 0.5 the mixture is symmetric — pG[1][l][m] == pG[1][m][l] bit for bit (equal mixing weights go through identical operations) — so the
      // nine terms (g_k[l] g_j[m]) pG[1][l][m] of entry [k][j] are entry [j][k]'s T[l][m] = (g_j[l] g_k[m]) pG[1][l][m], transposed: the reference adds them
      // l-major for [j][k] and, seen from [j][k]'s indices, m-major for [k][j].  The owner of both forms the exact products E and the terms T once: 36
      // multiplies per unordered pair instead of 54; the 32 adds, the four logs and the four accumulator adds are the reference's, on its operands, in its order.
      for (int ti = 0; ti < tp; ++ti) {
        const double* P = &s_pG[ti * 18];
        const double Q[3] = {P[0], P[3], P[6]};                              // pG[0][l][m] = q0[l]
        const double P1[5] = {P[9], P[10], P[11], P[14], P[17]};             // pG[1][l][m] = q1[l + m]
        const g_t* gr = &s_g[ti * GS];
#pragma unroll
        for (int su = 0; su < 2; ++su) {
          const int js = uj[su], ks = uk[su];
          const double a[3] = {(double)gr[js * 3], (double)gr[js * 3 + 1], (double)gr[js * 3 + 2]};
          const double b[3] = {(double)gr[ks * 3], (double)gr[ks * 3 + 1], (double)gr[ks * 3 + 2]};
          double E[3][3];
#pragma unroll
          for (int l = 0; l < 3; ++l)
#pragma unroll
            for (int m = 0; m < 3; ++m) E[l][m] = a[l] * b[m];                  // :553 (exact) — shared by [j][k] and [k][j] at both alphas
          {
            // [j][k][0]: l-major over (l, m) of (g_j[l] g_k[m]) q0[l]; [k][j][0]: of (g_k[l] g_j[m]) q0[l] = E[m][l] q0[l] (the first product initialises the sum)
            double s_jk0 = E[0][0] * Q[0], s_kj0 = E[0][0] * Q[0];
#pragma unroll
            for (int l = 0; l < 3; ++l)
#pragma unroll
              for (int m = 0; m < 3; ++m) {
                if (l == 0 && m == 0) continue;
                s_jk0 += (E[l][m] * Q[l]);
                s_kj0 += (E[m][l] * Q[l]);
              }
            acc[2 * su][0] += dmx_log2_fast_pinned(s_jk0, s_log, lk);          // :683
            acc[2 * su + 1][0] += dmx_log2_fast_pinned(s_kj0, s_log, lk);
          }
          {
#pragma unroll
            for (int l = 0; l < 3; ++l)
#pragma unroll
              for (int m = 0; m < 3; ++m) E[l][m] = E[l][m] * P1[l + m];        // T: :677-679 at alpha 0.5 — [k][j]'s nine terms are these, transposed
            double s_jk1 = E[0][0], s_kj1 = E[0][0];
#pragma unroll
            for (int l = 0; l < 3; ++l)
#pragma unroll
              for (int m = 0; m < 3; ++m) {
                if (l == 0 && m == 0) continue;
                s_jk1 += E[l][m];
                s_kj1 += E[m][l];
              }
            acc[2 * su][1] += dmx_log2_fast_pinned(s_jk1, s_log, lk);
            acc[2 * su + 1][1] += dmx_log2_fast_pinned(s_kj1, s_log, lk);
          }
        }
      }
    } else
    if (owner) {
      for (int ti = 0; ti < tp; ++ti) {
        const double* P = &s_pG[ti * 18];
        double P0[9], P1[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) { P0[i] = P[i]; P1[i] = P[9 + i]; }
        const g_t* gr = &s_g[ti * GS];
        const double a0 = (double)gr[j * 3], a1 = (double)gr[j * 3 + 1], a2 = (double)gr[j * 3 + 2];
        const double aj[3] = {a0, a1, a2};
#pragma unroll
        for (int kk = 0; kk < NK; ++kk) {
          const int k = min(kb * NK + kk, V - 1);
          const double b0 = (double)gr[k * 3], b1 = (double)gr[k * 3 + 1], b2 = (double)gr[k * 3 + 2];
          const double bk[3] = {b0, b1, b2};
          // :674-679, l-major.  The reference starts from 0 and adds nine products; 0 + x is x for these non-negative
          // products (no -0 can occur), so the first product initialises the sum: same bits, one add fewer per term.
          double s0 = (aj[0] * bk[0]) * P0[0], s1 = (aj[0] * bk[0]) * P1[0];
#pragma unroll
          for (int l = 0; l < 3; ++l)
#pragma unroll
            for (int m = 0; m < 3; ++m) {
              if (l == 0 && m == 0) continue;
              const double gp = aj[l] * bk[m];                                // :553 (exact)
              s0 += (gp * P0[l * 3 + m]);                                     // alpha 0
              s1 += (gp * P1[l * 3 + m]);                                     // alpha 1
            }
          if (CHK) ok &= __builtin_amdgcn_class(s0, 0x100) && __builtin_amdgcn_class(s1, 0x100);
          acc[kk][0] += dmx_log2_fast_pinned(s0, s_log, lk);                   // :683
          acc[kk][1] += dmx_log2_fast_pinned(s1, s_log, lk);
        }
      }
    }
    DMX_K2_SYNC();
  }
  if (cell_ok) {
    if constexpr (SYMU) {
#pragma unroll
      for (int su = 0; su < 2; ++su) {
        const int u = tid + TPC * su, d = 1 + (u >> kUSh);
        if (d < kUV / 2 || (d == kUV / 2 && uj[su] < kUV / 2)) {   // d = V/2 is reached from both sides: the j < V/2 thread stores; units beyond the last pair do not exist
          double* o = grid + (((size_t)cell * V + uj[su]) * V + uk[su]) * A;
          o[0] = acc[2 * su][0]; o[1] = acc[2 * su][1];
          double* o2 = grid + (((size_t)cell * V + uk[su]) * V + uj[su]) * A;
          o2[0] = acc[2 * su + 1][0]; o2[1] = acc[2 * su + 1][1];
        }
      }
    } else
    if (owner) {
#pragma unroll
      for (int kk = 0; kk < NK; ++kk) {
        const int k = kb * NK + kk;
        if (k < V) {
          double* o = grid + (((size_t)cell * V + j) * V + k) * A;
          o[0] = acc[kk][0]; o[1] = acc[kk][1];
        }
      }
    }
    if (tid < 2 && blockIdx.y == 0) l00[(size_t)cell * A + tid] = acc00;
    if (!ok) flag_cell(flagged, cell);
  }
#undef DMX_K2_SYNC
}
template <int TPC, int NK, int MINW = 1, bool GD = false, bool CHK = true, int TP = 32>
__global__ __launch_bounds__(kThreads, (MINW == 1 && TPC == 64) ? 3 : MINW) void k_doublet_a2(PileupView pv, int nrd_width, const float* __restrict__ g,
                                                         const double* __restrict__ gp0, const double* __restrict__ tabs,
                                                         const double* __restrict__ alpha,
                                                         const int32_t* __restrict__ sched, int32_t V, int32_t GS,
                                                         double* __restrict__ grid, double* __restrict__ l00,
                                                         uint8_t* __restrict__ flagged, const double* __restrict__ pfin, const double* __restrict__ pseed) {
  a2_body<TPC, NK, MINW, GD, CHK, TP, false>(pv, nrd_width, g, gp0, tabs, alpha, sched, V, GS, grid, l00, flagged, pfin, pseed);
}
// k_doublet_a2 over unordered pairs (STRICT, default grid, 32 soft-field samples: cfg3 — the headline): everything of k_doublet_a2<256,4,4,GD,noCHK> but phase 2's
// ownership (SYMU above).  The 32 diagonal entries [j][j][n] are not its: k_doublet_a2s<.., 0> (one wavefront per barcode) adds them behind it.
template <int MINW>
__global__ __launch_bounds__(kThreads, MINW) void k_doublet_a2u(PileupView pv, int nrd_width, const float* __restrict__ g,
                                                          const double* __restrict__ gp0, const double* __restrict__ tabs,
                                                          const double* __restrict__ alpha,
                                                          const int32_t* __restrict__ sched, int32_t V, int32_t GS,
                                                          double* __restrict__ grid, double* __restrict__ l00,
                                                          uint8_t* __restrict__ flagged, const double* __restrict__ pfin, const double* __restrict__ pseed) {
  a2_body<256, 4, MINW, true, false, 32, true>(pv, nrd_width, g, gp0, tabs, alpha, sched, V, GS, grid, l00, flagged, pfin, pseed);
}
// ... and for 16 samples (cfg5): k_doublet_a2<64,4,1,noCHK>'s kernel — 64 threads per barcode, four barcodes per workgroup, three wavefronts per SIMD
__global__ __launch_bounds__(kThreads, 3) void k_doublet_a2u16(PileupView pv, int nrd_width, const float* __restrict__ g,
                                                          const double* __restrict__ gp0, const double* __restrict__ tabs,
                                                          const double* __restrict__ alpha,
                                                          const int32_t* __restrict__ sched, int32_t V, int32_t GS,
                                                          double* __restrict__ grid, double* __restrict__ l00,
                                                          uint8_t* __restrict__ flagged, const double* __restrict__ pfin, const double* __restrict__ pseed) {
  a2_body<64, 4, 1, false, false, 32, true>(pv, nrd_width, g, gp0, tabs, alpha, sched, V, GS, grid, l00, flagged, pfin, pseed);
}

// K2, A = 2, FAST mode (dmx_engine_config.mode = DMX_MODE_FAST): k_doublet_a2 with the bilinear factoring of SURVEY H3.  Not the
// reference's operation sequence inside a term (so not STRICT), but the same accumulation order; tests bound it by 1e-9.
template <int TPC, int NK>
__global__ __launch_bounds__(kThreads, 3) void k_doublet_a2f(PileupView pv, int nrd_width, const float* __restrict__ g,
                                                         const double* __restrict__ gp0, const double* __restrict__ tabs,
                                                         const double* __restrict__ alpha,
                                                         const int32_t* __restrict__ sched, int32_t V, int32_t GS,
                                                         double* __restrict__ grid, double* __restrict__ l00,
                                                         uint8_t* __restrict__ flagged) {
  constexpr int A = 2, TP = 32;
  constexpr int CPW = kThreads / TPC;            // cells per workgroup
  constexpr int T00 = TP + 2;
#define DMX_K2_SYNC() do { if (TPC == 64) { DMX_WAVE_LDS_ORDER(); } else { __syncthreads(); } } while (0)
  extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
  __shared__ double s_tab[kTab2];
  const double* s_log = s_tab + kLut2;
  const int t = threadIdx.x;
  stage_k2_tables(s_tab, tabs, t, kThreads);
  __syncthreads();

  const int cw = t / TPC, tid = t % TPC;         // cell slot inside the workgroup, thread inside the cell
  // per-cell LDS regions
  constexpr bool SHARE = TPC > 64;               // one-wavefront cells (V <= 16) form u in registers: no LDS tile, no extra syncs
  constexpr int SUB = 8;                         // pairs per u sub-tile
  const int VU = V;                              // u row: [alpha][k][4 doubles] (4th pads the 3 to 32 bytes)
  const size_t cell_bytes = (size_t)TP * 18 * 8 + (size_t)TP * GS * 4 + 2 * T00 * 8 + TP * (4 + 4 + 8) + (SHARE ? (size_t)SUB * 2 * VU * 32 : 0);
  unsigned char* base = s_raw + (size_t)cw * cell_bytes;
  double* s_pG = (double*)base;                                  // [TP][2][9]
  float* s_g = (float*)(base + (size_t)TP * 18 * 8);             // [TP][GS]   genotype rows of the tile's SNPs
  double* s_t00 = (double*)((unsigned char*)s_g + (size_t)TP * GS * 4);   // [2][T00]   llks00 terms
  int64_t* s_off = (int64_t*)(s_t00 + 2 * T00);                  // [TP]
  int32_t* s_snp = (int32_t*)(s_off + TP);                       // [TP]
  uint32_t* s_cnt = (uint32_t*)(s_snp + TP);                     // [TP]
  double* s_u = (double*)(s_cnt + TP);                           // [SUB][2][V][4]   u[l] = sum_m pG[n][l][m] * g_k[m]

  const int slot = blockIdx.x * CPW + cw;
  if (TPC == 64 && slot >= pv.B) return;         // whole wavefront idle (no workgroup barriers below in this mode)
  const bool cell_ok = slot < pv.B;
  const int32_t cell = cell_ok ? sched[slot] : 0;
  const int64_t p_beg = cell_ok ? pv.cell_pair_off[cell] : 0;
  const int64_t np = cell_ok ? pv.cell_pair_off[cell + 1] - p_beg : 0;
  int64_t rd_base = cell_ok ? pv.cell_read_off[cell] : 0;

  // phase-2 identity.  A workgroup covers JS = TPC / KB rows j of the cell's grid; panels with more rows than that are cut
  // into j-slabs, one workgroup (blockIdx.y) per slab, each repeating the cheap phases 0-1 for itself.
  const int KB = (V + NK - 1) / NK;              // k-blocks per j
  const int JS = TPC / KB;
  const int jl = tid / KB, kb = tid % KB;
  const int j = (int)blockIdx.y * JS + jl;
  const bool owner = jl < JS && j < V;
  double acc[NK][A];
#pragma unroll
  for (int kk = 0; kk < NK; ++kk) { acc[kk][0] = 0.0; acc[kk][1] = 0.0; }
  bool ok = true;
  const DmxLogPins lk = dmx_log_pins();
  // phase-1 identity (first wavefront of the cell): pair ti1, alpha n1; mixing weights of :613
  const int ti1 = tid >> 1, n1 = tid & 1;
  double wA[9], wR[9];
  {
    const double al = alpha[n1];
#pragma unroll
    for (int l = 0; l < 3; ++l)
#pragma unroll
      for (int m = 0; m < 3; ++m) {
        const double p = 0.5 * l + (m - l) * 0.5 * al;
        wA[l * 3 + m] = p;
        wR[l * 3 + m] = 1.0 - p;
      }
  }
  double acc00 = 0.0;                            // lane n1 == tid < 2 owns llks00[n]
  const int row_len = V * 3;

  for (int64_t tbase = 0; tbase < np; tbase += TP) {
    const int tp = (int)min((int64_t)TP, np - tbase);
    // ---- headers of the tile's pairs (first 32 lanes of the cell)
    if (tid < TP) {
      const bool v = tid < tp;
      const uint32_t n = v ? load_nrd(pv.pair_nrd, p_beg + tbase + tid, nrd_width) : 0u;
      const uint32_t incl = seg_scan_incl<32>(n);
      s_cnt[tid] = n;
      s_off[tid] = rd_base + (int64_t)(incl - n);
      s_snp[tid] = v ? (pv.pair_snp ? pv.pair_snp[p_beg + tbase + tid] : (int32_t)(tbase + tid)) : 0;
    }
    DMX_K2_SYNC();
    rd_base = s_off[tp - 1] + (int64_t)s_cnt[tp - 1];
    // ---- genotype rows -> LDS (coalesced along the row)
    {
      int r = tid % row_len, ti = tid / row_len;
      const int dr = TPC % row_len, dt = TPC / row_len;
      while (ti < tp) {
        s_g[ti * GS + r] = g[(size_t)s_snp[ti] * row_len + r];
        r += dr; ti += dt;
        if (r >= row_len) { r -= row_len; ++ti; }
      }
    }
    // ---- phase 1
    if (tid < 64) {
      const bool on = ti1 < tp;
      const uint32_t cnt = on ? s_cnt[ti1] : 0u;
      const int64_t off = on ? s_off[ti1] : 0;
      const uint32_t rd4 = load_rd4(pv, off, cnt);       // the first four read bytes in one load (one dependent latency instead of four)
      double pG[9];
#pragma unroll
      for (int i = 0; i < 9; ++i) pG[i] = 1.0;                               // :597
      for (uint32_t r = 0; __any(r < cnt); ++r) {
        const bool live = r < cnt;
        const uint32_t byte = live ? (r < 4 ? (rd4 >> (8 * r)) & 0xFFu : (uint32_t)pv.reads[off + r]) : 0u;
        const uint32_t bq = byte & 127u;
        const bool alt = (byte >> 7) != 0;
        const double pR = alt ? s_tab[128 + bq] : s_tab[bq];                // :606
        const double pA = alt ? s_tab[bq] : s_tab[128 + bq];                // :607
        double mx = 0.0;
        if (live) {
#pragma unroll
          for (int i = 0; i < 9; ++i) {
            pG[i] *= (pR * wR[i] + pA * wA[i]);                             // :625
            mx = fmax(mx, pG[i]);                                 // :626-627
          }
        }
        {
          const double o = shfl_xor1(mx);                               // one max across both alphas of the pair
          mx = fmax(mx, o);
        }
        if (live) {
          if (cnt <= kSafeReads) {
            const double y = rcp_refined(mx);
#pragma unroll
            for (int i = 0; i < 9; ++i) pG[i] = div_by(pG[i], mx, y);       // :632-639
          } else {
#pragma unroll
            for (int i = 0; i < 9; ++i) pG[i] /= mx;
          }
        }
      }
      double mx = 0.0;
#pragma unroll
      for (int i = 0; i < 9; ++i) {
        pG[i] += 1e-6;                                                       // :649
        mx = fmax(mx, pG[i]);
      }
      {
        const double o = shfl_xor1(mx);
        mx = fmax(mx, o);
      }
      if (on) {
        const double y = rcp_refined(mx);                                    // numerators >= 1e-6, mx in [1e-6, 1+1e-6]
        const double* g0 = gp0 + (size_t)s_snp[ti1] * 3;
        const double q0 = g0[0], q1 = g0[1], q2 = g0[2];
        const double qq[3] = {q0, q1, q2};
        double sum = 0.0;
#pragma unroll
        for (int l = 0; l < 3; ++l)
#pragma unroll
          for (int m = 0; m < 3; ++m) {
            const double v = div_by(pG[l * 3 + m], mx, y);                   // :656-663
            s_pG[(ti1 * 2 + n1) * 9 + l * 3 + m] = v;
            sum += ((qq[l] * qq[m]) * v);                                    // gp00 (:555) then :702-705
          }
        ok &= __builtin_amdgcn_class(sum, 0x100);
        s_t00[n1 * T00 + ti1] = dmx_log2_fast(sum, s_log);                    // :708-709 term
      }
    }
    DMX_K2_SYNC();
    // ---- llks00: lane n < 2 of the cell adds its alpha's terms in pair order
    if (tid < 2) {
      const double* row = &s_t00[tid * T00];
      if (tp == TP) {                              // loads first (LDS latency paid once), then the ordered adds
        double2 v[TP / 2];
#pragma unroll
        for (int i = 0; i < TP / 2; ++i) v[i] = *reinterpret_cast<const double2*>(&row[2 * i]);
#pragma unroll
        for (int i = 0; i < TP / 2; ++i) { acc00 += v[i].x; acc00 += v[i].y; }
      } else {
        for (int i = 0; i < tp; ++i) acc00 += row[i];
      }
    }
    // ---- phase 2, in sub-tiles of SUB pairs.  FAST (bilinear) form: the nine-term sum g_j' pG[n] g_k is factored as
    // g_j . u with u[l] = sum_m pG[n][l][m] g_k[m] formed ONCE per (pair, alpha, k) and shared through LDS by the rows j — three
    // fused multiply-adds per evaluation instead of nine products and seventeen multiply/adds.  The value of a term moves by a
    // few ulp (<= ~1e-15 absolute); the accumulation order is the reference's, so the result stays within ~1e-11 of STRICT.
    if (!SHARE) {
      if (owner) {
        for (int ti = 0; ti < tp; ++ti) {
          const double* P = &s_pG[ti * 18];
          double P0[9], P1[9];
#pragma unroll
          for (int i = 0; i < 9; ++i) { P0[i] = P[i]; P1[i] = P[9 + i]; }
          const float* gr = &s_g[ti * GS];
          const double a0 = (double)gr[j * 3], a1 = (double)gr[j * 3 + 1], a2 = (double)gr[j * 3 + 2];
#pragma unroll
          for (int kk = 0; kk < NK; ++kk) {
            const int k = min(kb * NK + kk, V - 1);
            const double b0 = (double)gr[k * 3], b1 = (double)gr[k * 3 + 1], b2 = (double)gr[k * 3 + 2];
            double x[3], y[3];
#pragma unroll
            for (int l = 0; l < 3; ++l) {
              x[l] = __builtin_fma(P0[l * 3 + 2], b2, __builtin_fma(P0[l * 3 + 1], b1, P0[l * 3] * b0));
              y[l] = __builtin_fma(P1[l * 3 + 2], b2, __builtin_fma(P1[l * 3 + 1], b1, P1[l * 3] * b0));
            }
            const double s0 = __builtin_fma(a2, x[2], __builtin_fma(a1, x[1], a0 * x[0]));
            const double s1 = __builtin_fma(a2, y[2], __builtin_fma(a1, y[1], a0 * y[0]));
            ok &= __builtin_amdgcn_class(s0, 0x100) && __builtin_amdgcn_class(s1, 0x100);
            acc[kk][0] += dmx_log2_fastmode(s0, s_log, lk);
            acc[kk][1] += dmx_log2_fastmode(s1, s_log, lk);
          }
        }
      }
      DMX_K2_SYNC();
    } else
#pragma unroll 1
    for (int sub = 0; sub < tp; sub += SUB) {
      const int ns = min(SUB, tp - sub);
#pragma unroll 1
      for (int e = tid; e < ns * 2 * V; e += TPC) {
        const int k = e % V, n = (e / V) & 1, pi = e / (2 * V);
        const double* P = &s_pG[((sub + pi) * 2 + n) * 9];
        const float* gr = &s_g[(sub + pi) * GS + k * 3];
        const double b0 = (double)gr[0], b1 = (double)gr[1], b2 = (double)gr[2];
        double* u = &s_u[(size_t)((pi * 2 + n) * VU + k) * 4];
#pragma unroll
        for (int l = 0; l < 3; ++l) u[l] = __builtin_fma(P[l * 3 + 2], b2, __builtin_fma(P[l * 3 + 1], b1, P[l * 3] * b0));
      }
      DMX_K2_SYNC();
      if (owner) {
#pragma unroll 1
        for (int pi = 0; pi < ns; ++pi) {
          const float* gr = &s_g[(sub + pi) * GS];
          const double a0 = (double)gr[j * 3], a1 = (double)gr[j * 3 + 1], a2 = (double)gr[j * 3 + 2];
          const double* u0 = &s_u[(size_t)((pi * 2 + 0) * VU) * 4];
          const double* u1 = &s_u[(size_t)((pi * 2 + 1) * VU) * 4];
#pragma unroll
          for (int kk = 0; kk < NK; ++kk) {
            const int k = min(kb * NK + kk, V - 1);
            const double2 x01 = *reinterpret_cast<const double2*>(&u0[k * 4]);
            const double x2 = u0[k * 4 + 2];
            const double2 y01 = *reinterpret_cast<const double2*>(&u1[k * 4]);
            const double y2 = u1[k * 4 + 2];
            const double s0 = __builtin_fma(a2, x2, __builtin_fma(a1, x01.y, a0 * x01.x));
            const double s1 = __builtin_fma(a2, y2, __builtin_fma(a1, y01.y, a0 * y01.x));
            ok &= __builtin_amdgcn_class(s0, 0x100) && __builtin_amdgcn_class(s1, 0x100);
            acc[kk][0] += dmx_log2_fastmode(s0, s_log, lk);
            acc[kk][1] += dmx_log2_fastmode(s1, s_log, lk);
          }
        }
      }
      DMX_K2_SYNC();
    }
    DMX_K2_SYNC();
  }
  if (cell_ok) {
    if (owner) {
#pragma unroll
      for (int kk = 0; kk < NK; ++kk) {
        const int k = kb * NK + kk;
        if (k < V) {
          double* o = grid + (((size_t)cell * V + j) * V + k) * A;
          o[0] = acc[kk][0]; o[1] = acc[kk][1];
        }
      }
    }
    if (tid < 2 && blockIdx.y == 0) l00[(size_t)cell * A + tid] = acc00;
    if (!ok) flag_cell(flagged, cell);
  }
#undef DMX_K2_SYNC
}

// ---------------------------------------------------------------------------------------------------------------------
// K2, FAST mode, alpha grid {0, 0.5} (demuxlet's default): only the grid entries demuxlet ever prints or decides on.
//   * llksAB[j][k][0] for k != 0 is read by nothing but the maxLLK scan (cmd_cram_demuxlet.cpp:713-721; the printed uses of the
//     alpha[0] entries are [j][0][0] only, :726,:749-757,:774-780,:816-825), and maxLLK cancels in every printed posterior
//     (:726-733,:768,:780,:792,:827-828).  With alpha[0] == 0 the entry does not depend on k beyond float32 rounding of
//     sum_m g_k[m].  They are not computed; the grid cells are filled with [j][0][0].
//   * at alpha == 0.5 the mixture is symmetric: [k][j][1] is [j][k][1] up to the rounding of the nine-term sum (<= 3e-14 in the
//     reference), .pair prints j < k only (:785) and the host tie arbiter settles which order .best names.  One
//     evaluation per unordered pair {j,k}, mirrored into both cells.
// Per covered pair that is V + V(V+1)/2 log terms instead of 2 V^2: 560 instead of 2048 at V = 32.  Everything else follows
// k_doublet_a2f: the bilinear factoring g_j . u_k with u_k[l] = sum_m pG[1][l][m] g_k[m] formed once per (pair, k); every
// accumulator is owned by one lane which adds its terms in ascending SNP order (what keeps the result within ~1e-11 of STRICT).
// Entry e = tid + TPC*i of a lane: e < V*D (D = V/2 + 1): j = e % V, k = (j + e / V) % V at alpha 0.5 (the rotation makes the
// lanes of a wavefront read consecutive u_k: conflict-free LDS); V*D <= e < V*D + V: the singlet entry [j][0][0], j = e - V*D.
// For even V the offset d = V/2 is computed from both sides; only the j < k copy is stored.
// Panels wider than 64 samples cut the entry list into slabs of NEP entries per lane, one workgroup (blockIdx.y) per slab: each slab
// repeats the per-tile phases (cheap next to 256 * NEP evaluations per pair) and owns its entries' accumulators.
template <int TPC, int VMAX, int SUB, bool FIXJ, int MINW = 3, int NEP = 0, bool CHK = true>   // CHK: see k_doublet_a2
__global__ __launch_bounds__(kThreads, MINW) void k_doublet_sym(PileupView pv, int nrd_width, const float* __restrict__ g,
                                                          const double* __restrict__ gp0, const double* __restrict__ tabs,
                                                          const int32_t* __restrict__ sched, int32_t V_and_flags,
                                                          double* __restrict__ grid, double* __restrict__ l00,
                                                          uint8_t* __restrict__ flagged, const double* __restrict__ pfin, const double* __restrict__ pseed) {
  // pfin (round 6; NULL: none): k_build_certify_finals' table — the finished phase-1 values of pairs of up to three tabled reads; pseed (NULL: none):
  // k_build_certify_seeds' table — the state after a pair's first one or two reads, for the pairs that walk the loop
  const int32_t V = V_and_flags & 0xFFFF;
  const bool no_dma = (V_and_flags >> 16) & 1;   // kernel experiments (DMX_SYM_NO_DMA; bit-identical results)
#if DMX_SYM_ABLATIONS                             // timing builds only (tools/build_variant.sh ... -DDMX_SYM_ABLATIONS=1): each switch drops one part of the
                                                  // kernel — results WRONG, time right.  The shipped library does not contain them.
  const bool abl_p1 = (V_and_flags >> 17) & 1;   // DMX_SYM_ABLATE_P1: phase 1's global loads all hit the same lines
  const bool abl_p2 = (V_and_flags >> 18) & 1;   // DMX_SYM_ABLATE_P2: no phase-2 evaluations
  const bool abl_u = (V_and_flags >> 19) & 1;    // DMX_SYM_ABLATE_U: u is not formed
  const bool abl_00 = (V_and_flags >> 21) & 1;   // DMX_SYM_ABLATE_00: no llks00 sums
#else
  constexpr bool abl_p1 = false, abl_p2 = false, abl_u = false, abl_00 = false;
#endif
  const bool no_prod = (V_and_flags >> 22) & 1;  // kernel experiment (DMX_SYM_NO_PRODUCT): a log per term also in full sub-tiles
  const bool no_pipe = (V_and_flags >> 23) & 1;  // kernel experiment (DMX_SYM_NO_PIPE; bit-identical results): phase 2 without the software pipeline
  const bool wait_all = (V_and_flags >> 24) & 1; // test switch (DMX_SYM_WAIT_ALL; bit-identical results): vmcnt(0) instead of the three-buffer form's counted wait
  // Narrow panels put SEVERAL barcodes in one wavefront (TPC = 32: two, TPC = 16: four): their entries fill the lanes (V = 16: 160
  // entries = 5 per lane of 32; V = 8: 48 = 3 per lane of 16) and the per-tile phases 0-1 are shared instruction-wise.  The tile is
  // what one pass of phase 1 covers: two lanes per pair.
  static_assert(TPC == 16 || TPC == 32 || TPC == 64 || TPC == 256, "threads per barcode");
  constexpr int A = 2, TP = TPC >= 64 ? 32 : TPC / 2;
  static_assert(SUB <= TP && TP % 8 == 0, "sub-tile");
  constexpr int CPW = kThreads / TPC;            // cells per workgroup
  constexpr int T00 = TP + 2;
  constexpr int NE = NEP ? NEP : (VMAX * (VMAX / 2 + 1) + VMAX + TPC - 1) / TPC;   // entries per lane (and slab)
  const int e_base = NEP ? (int)blockIdx.y * NEP * TPC : 0;         // first entry of this slab
  constexpr int VUS = (VMAX + 2) & ~1;           // u row stride (V alpha-0.5 rows + the alpha-0 row of sample 0), even
  constexpr int GSS = (3 * VMAX + 3) & ~3;       // genotype row stride (floats)
#define DMX_K2_SYNC() do { if (TPC <= 64) { DMX_WAVE_LDS_ORDER(); } else { __syncthreads(); } } while (0)
  extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
  __shared__ double s_tab[kTab2];
  __shared__ double s_w[2][10];                  // mixing weights of :613 per alpha and distinct value: [n][0..4] = p (ALT), [n][5..9] = 1 - p
  __shared__ __attribute__((aligned(256))) double s_log32[DMX_LOG32_TABLE_DOUBLES];   // dmx_log2_lite32's rc[32] | logc[32]
  const double* s_log = s_tab + kLut2;
  const int t = threadIdx.x;
  stage_k2_tables(s_tab, tabs, t, kThreads);
  if (t < DMX_LOG32_TABLE_DOUBLES) s_log32[t] = tabs[kTabLog32 + t];
  // Round 6: WHICH of the two FAST logs an entry takes is a compile-time function of its slot i in the lane's entry list (so every accumulator sees one log
  // for the whole run, in every code path of this instantiation): dmx_log2_lite (256 bins, 16-byte gather with bank conflicts, 6 FP64 instructions) loads the
  // LDS, dmx_log2_lite32 (split 32-bin table, conflict-free, 8) loads the VALU — mixing them balances the two units that bind this kernel.
  constexpr int HYB = (FIXJ && MINW == 3 && NEP == 0 && DMX_FAST_LITE_LOG) ? DMX_SYM_HYB : 0;
  auto kind32 = [](int i) constexpr { return HYB == 3 || (HYB == 1 && i % 3 == 2) || (HYB == 2 && i % 3 != 0) || (HYB == 4 && (i & 1)); };
  const DmxLog32Pins lk32 = dmx_log32_pins();
  if (t < 10) {
    // alpha 0: p = 0.5 l does not depend on m -> slots 0..2 hold l = 0..2 (slots 3, 4 repeat l = 2); alpha 0.5: p = 0.25 (l + m)
    // -> slot = l + m.  Same operands, same operations as the nine entries of the reference: the values are bit-identical.
    const int n = t / 5, q = t % 5;
    const int l = n ? (q > 2 ? 2 : q) : min(q, 2), m = n ? q - l : 0;
    const double p = 0.5 * l + (m - l) * 0.5 * (n ? 0.5 : 0.0);
    s_w[n][q] = p;
    s_w[n][5 + q] = 1.0 - p;
  }
  __syncthreads();

  const int cw = t / TPC, tid = t % TPC;         // cell slot inside the workgroup, thread inside the cell
  // Up to two barcodes per wavefront (TPC >= 32): the genotype rows of a sub-tile go from global memory
  // straight into the LDS (global_load_lds: no registers, uniform source base per row), one sub-tile ahead of their use, into the
  // other half of a double buffer — the first sub-tile's while phase 1 runs.  cfg3 FAST: 394 -> 349 ms on one box.
  constexpr bool DMA_T = TPC >= 32;              // (four barcodes per wavefront, TPC = 16: too many small masked loads, measured +17 %)
  // row buffers: two (the next sub-tile's rows travel while this one computes) — THREE for the two-barcodes-per-wavefront form (V <= 16, small rows:
  // round 5), whose sparse workloads gather rows from a matrix beyond the L2 (cfg5: 38 MB) and waited for them half of their time with one sub-tile of lead
  constexpr int NBUF = DMA_T ? (TPC == 32 ? 3 : 2) : 1;
  constexpr size_t cell_bytes = (size_t)TP * 6 * 8 + (size_t)TP * 4 * 8 + 2 * T00 * 8 + TP * (4 + 4 + 8) + (size_t)NBUF * SUB * GSS * 4 + (size_t)SUB * 3 * VUS * 8;
  unsigned char* base = s_raw + (size_t)cw * cell_bytes;
  double* s_q1 = (double*)base;                                  // [TP][6]    pG of alpha 0.5: q[l+m], five distinct values
  double* s_u0 = s_q1 + TP * 6;                                  // [TP][4]    u of (alpha 0, sample 0)
  double* s_t00 = s_u0 + TP * 4;                                 // [2][T00]   llks00 terms
  int64_t* s_off = (int64_t*)(s_t00 + 2 * T00);                  // [TP]
  int32_t* s_snp = (int32_t*)(s_off + TP);                       // [TP]
  uint32_t* s_cnt = (uint32_t*)(s_snp + TP);                     // [TP]
  float* s_g0 = (float*)(s_cnt + TP);                            // [SUB][GSS] genotype rows of the sub-tile's SNPs (DMA: x 2)
  double* s_u = (double*)(s_g0 + NBUF * SUB * GSS);              // [SUB][3][VUS]

  const int slot = blockIdx.x * CPW + cw;
  if (TPC == 64 && slot >= pv.B) return;         // whole wavefront idle (no workgroup barriers below in this mode)
  const bool cell_ok = slot < pv.B;              // (sub-wavefront barcodes past the end stay in the wavefront with zero pairs)
  const int32_t cell = cell_ok ? sched[slot] : 0;
  const int64_t p_beg = cell_ok ? pv.cell_pair_off[cell] : 0;
  const int64_t np = cell_ok ? pv.cell_pair_off[cell + 1] - p_beg : 0;
  int64_t rd_base = cell_ok ? pv.cell_read_off[cell] : 0;

  // phase-2 identity: NE entries per lane
  const int D = V / 2 + 1, VD = V * D;
  int ek[NE], ej[NE];                            // u row (k, or V = the alpha-0 row) and sample j of each entry
#pragma unroll
  for (int i = 0; i < NE; ++i) {
    const int e = e_base + tid + TPC * i;
    if (e < VD) { const int j = e % V; int k = j + e / V; k = k >= V ? k - V : k; ej[i] = j; ek[i] = k; }
    else if (e < VD + V) { ej[i] = e - VD; ek[i] = V; }
    else { ej[i] = 0; ek[i] = V; }               // idle slot: computes the first singlet term, never stored
  }
  double acc[NE];
#pragma unroll
  for (int i = 0; i < NE; ++i) acc[i] = 0.0;
  // u-formation identity: entry tid + TPC * i of a sub-tile's SUB * V (pair, sample) values
  constexpr int NU = (SUB * VMAX + TPC - 1) / TPC;
  int uq[NU];                                    // (pair << 16) | sample; negative: no such entry
#pragma unroll
  for (int i = 0; i < NU; ++i) {
    const int e = tid + TPC * i;
    uq[i] = e < SUB * V ? ((e / V) << 16) | (e % V) : (int)0x80000000;
  }
  bool ok = true;
  const DmxLogPins lk = dmx_log_pins();
  // phase-1 identity (first wavefront of the cell): pair ti1, alpha n1
  const int ti1 = tid >> 1, n1 = tid & 1;
  double acc00 = 0.0;                            // lane n1 == tid < 2 owns llks00[n]
  const int row_len = V * 3;

  // Narrow panels (short tiles, several barcodes per wavefront) request a tile's header one tile ahead; the wide forms have no
  // register to spare for it (9 entries per lane at 167 VGPRs) and hide that latency behind their longer phase 2.
  constexpr bool PREFETCH = TPC < 64;
  uint32_t hd_n = 0u; int32_t hd_s = 0;          // header (stored reads, SNP id) of this lane's pair in the NEXT tile
  if (PREFETCH && tid < TP && tid < np) { hd_n = load_nrd(pv.pair_nrd, p_beg + tid, nrd_width); hd_s = pv.pair_snp ? pv.pair_snp[p_beg + tid] : (int32_t)tid; }
  const bool dma = DMA_T && !no_dma;
  auto request_rows = [&](int sub, int buf) {    // rows of the pairs sub .. sub+SUB-1 of the current tile -> s_g0[buf], asynchronously
    using gptr = const __attribute__((address_space(1))) void*;
    using lptr = __attribute__((address_space(3))) void*;
    const int wv = tid & ~63;                      // first lane of this wavefront inside the cell
    constexpr int SPW = TPC >= 64 ? 1 : 64 / TPC;
    const int slot = SPW == 1 ? 0 : (threadIdx.x & 63) / TPC;
    // (a ROLLED loop over the slots: with the slots as separate branches the compiler merges their identical tails into one
    //  instruction that serves lanes of two barcodes with one LDS base — it takes the base operand for uniform)
#pragma unroll 1
    for (int c = 0; c < SPW; ++c) {
      if (SPW == 1 || slot == c) {
#pragma unroll
    for (int pi = 0; pi < SUB; ++pi) {             // a row at a time: uniform source base, lane r reads element r
      const float* src = g + (size_t)__builtin_amdgcn_readfirstlane(s_snp[sub + pi]) * row_len;
#pragma unroll
      for (int h = 0; h < (GSS + TPC - 1) / TPC; ++h) {
        const int r = tid + TPC * h;
        // (a piece with no element of the row is skipped by the whole barcode: TPC * h < row_len is uniform, so the number of loads a request
        //  puts in flight is SUB * ceil(row_len / TPC) exactly — what the counted wait of the three-buffer form relies on)
        if (TPC * h < row_len && r < row_len) __builtin_amdgcn_global_load_lds((gptr)(src + r), (lptr)(s_g0 + (buf * SUB + pi) * GSS + wv + TPC * h - c * TPC), 4, 0, 0);
      }
    }
      }
    }
  };
  for (int64_t tbase = 0; tbase < np; tbase += TP) {
    const int tp = (int)min((int64_t)TP, np - tbase);
    // ---- headers of the tile's pairs (first 32 lanes of the cell)
    if (tid < TP) {
      uint32_t n; int32_t sn;
      if (PREFETCH) {                              // this tile's header was requested a tile ago; now request the next one's
        n = hd_n; sn = hd_s;
        const int64_t nx = tbase + TP + tid;
        hd_n = 0u; hd_s = 0;
        if (nx < np) { hd_n = load_nrd(pv.pair_nrd, p_beg + nx, nrd_width); hd_s = pv.pair_snp ? pv.pair_snp[p_beg + nx] : (int32_t)nx; }
      } else {
        const bool v = tid < tp;
        n = v ? load_nrd(pv.pair_nrd, p_beg + tbase + tid, nrd_width) : 0u;
        sn = v ? (pv.pair_snp ? pv.pair_snp[p_beg + tbase + tid] : (int32_t)(tbase + tid)) : 0;
      }
      const uint32_t incl = seg_scan_incl<TP>(n);
      s_cnt[tid] = n;
      s_off[tid] = rd_base + (int64_t)(incl - n);
      s_snp[tid] = sn;
    }
    DMX_K2_SYNC();
    rd_base = s_off[tp - 1] + (int64_t)s_cnt[tp - 1];
    if (dma) {
      request_rows(0, 0);                          // sub-tile 0's rows travel while phase 1 runs
      if (NBUF == 3 && SUB < tp) request_rows(SUB, 1);
    }
    // ---- phase 1: pG[n][3][3] of the pair (:600-663), exactly as k_doublet_a2; kept: alpha 0.5's nine values, and for
    //      alpha 0 the three u values of sample 0 (the only k the singlet column [j][0][0] needs)
    if (tid < 2 * TP) {
      const bool on = ti1 < tp;
      const uint32_t cnt = on ? s_cnt[ti1] : 0u;
      const int64_t off = on ? s_off[ti1] : 0;
      const uint32_t rd4 = load_rd4(pv, abl_p1 ? (off & 63) : off, cnt);       // the first four read bytes in one load (one dependent latency instead of four)
      const int32_t snp1 = on ? (abl_p1 ? (s_snp[ti1] & 15) : s_snp[ti1]) : 0;
      const float* g0r = g + (size_t)snp1 * row_len;             // sample 0's row (alpha-0 lanes use it)
      const float gf0 = g0r[0], gf1 = g0r[1], gf2 = g0r[2];
      double q[5];
      // Round 6: a pair of up to three tabled reads takes its FINISHED values from the table (one 64-byte entry per read code: the alpha = 0.5 lane its
      // five, the alpha = 0 lane its three); the read loop and the +1e-6 renormalisation below run only in tiles with a deeper pair (wave-uniform branch).
      // Timing builds: the loop is 22 of cfg3 FAST's 201 ms (a dependent chain — products, maximum, the neighbour's, reciprocal, quotients — per read,
      // with LDS look-ups in it), at 1.25 reads per pair 93 % of the tiles skip it.  Same operations on the same operands (k_build_certify_finals): same bits.
      // (a tile with a deeper pair runs the loop for ALL its lanes, as before: no merge of the two sources, no register held across the loop)
      const int32_t fi = pfin ? certify_final_index(cnt, rd4) : -1;
      if (!__any(fi < 0)) {
        const double* fp = pfin + (size_t)fi * kCFinStride + (n1 ? 0 : 5);
        q[0] = fp[0]; q[1] = fp[1]; q[2] = fp[2];
        q[3] = n1 ? fp[3] : 0.0; q[4] = n1 ? fp[4] : 0.0;
      } else {
        // the loop of :597-639 and the +1e-6 renormalisation (:649-663) — k_certify's function (each lane keeps its own alpha's values): on the default
        // grid a pair's first one or two reads come from the seed table (round 6 for this kernel: the loop starts at read 1 or 2 — what matters for pileups
        // too deep for the final-value table, cfg5's 2 reads per pair)
        double wA[5], wR[5];                                               // the weights live in registers during phase 1 only
#pragma unroll
        for (int i = 0; i < 5; ++i) { wA[i] = s_w[n1][i]; wR[i] = s_w[n1][5 + i]; }
        certify_pair_values<5, false>(pv, cnt, off, rd4, s_tab, wA, wR, n1, q, pseed);
      }
      if (on) {
        const double* g0 = gp0 + (size_t)snp1 * 3;
        const double q0 = g0[0], q1 = g0[1], q2 = g0[2];
        const double qq[3] = {q0, q1, q2};
        double sum = 0.0;
#pragma unroll
        for (int l = 0; l < 3; ++l)
#pragma unroll
          for (int m = 0; m < 3; ++m) {
            const double v = n1 ? q[l + m] : q[l];                           // pG[n][l][m]
            sum += ((qq[l] * qq[m]) * v);                                    // gp00 (:555) then :702-705
          }
        ok &= __builtin_amdgcn_class(sum, 0x100);
        s_t00[n1 * T00 + ti1] = dmx_log2_fast(sum, s_log);                    // :708-709 term
        if (n1) {
#pragma unroll
          for (int i = 0; i < 5; ++i) s_q1[ti1 * 6 + i] = q[i];
        } else {
          const double b0 = (double)gf0, b1 = (double)gf1, b2 = (double)gf2;
#pragma unroll
          for (int l = 0; l < 3; ++l) s_u0[ti1 * 4 + l] = __builtin_fma(q[l], b2, __builtin_fma(q[l], b1, q[l] * b0));
        }
      }
    }
    DMX_K2_SYNC();
    // ---- llks00: lane n < 2 of the cell adds its alpha's terms in pair order
    if (tid < 2 && !abl_00) {
      const double* row = &s_t00[tid * T00];
      if (tp == TP) {                              // eight terms at a time: loads first, then the ordered adds
#pragma unroll 1
        for (int h = 0; h < TP; h += 8) {
          double2 v[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] = *reinterpret_cast<const double2*>(&row[h + 2 * i]);
#pragma unroll
          for (int i = 0; i < 4; ++i) { acc00 += v[i].x; acc00 += v[i].y; }
        }
      } else {
        for (int i = 0; i < tp; ++i) acc00 += row[i];
      }
    }
    // ---- phase 2 in sub-tiles of SUB pairs
#pragma unroll 1
    for (int sub = 0; sub < tp; sub += SUB) {
      const int ns = min(SUB, tp - sub);
      const int buf = dma ? (NBUF == 3 ? (sub / SUB) % 3 : (sub / SUB) & 1) : 0;
      float* s_g = s_g0 + buf * SUB * GSS;
      if (dma) {
        if constexpr (NBUF == 3) {
          // this sub-tile's rows have landed when at most the loads THIS BARCODE requested after them are in flight (loads return in order): kReq = SUB
          // rows x pieces per row, per request and barcode (request_rows).  The wavefront's other barcode issues its own requests in between (or
          // none: its tile may be shorter), which can only make this wait longer, never too short.
          static_assert((GSS + TPC - 1) / TPC <= 2 && 2 * SUB <= 15, "vmcnt field / pieces per row");
          // (the count is safe only while request_rows issues EXACTLY SUB x pieces loads per barcode — its `TPC * h < row_len` test is uniform and its
          //  slot loop rolled for that reason; tests/test_gpu_parity.py::test_counted_row_wait_equals_waiting_for_everything compares this form with
          //  wait_all bit for bit on partial last sub-tiles, barcodes of unequal length in one wavefront and every V the form serves)
          if (sub + SUB >= tp || wait_all) __builtin_amdgcn_s_waitcnt(0x0F70);                // nothing requested after them: vmcnt(0)
          else if (row_len > TPC) __builtin_amdgcn_s_waitcnt(0x0F70 | (2 * SUB));             // vmcnt(kReq), two pieces per row
          else __builtin_amdgcn_s_waitcnt(0x0F70 | SUB);                                      // one piece per row
          if (sub + 2 * SUB < tp) request_rows(sub + 2 * SUB, (buf + 2) % 3);   // two sub-tiles ahead (that buffer was last read a sub-tile ago, two syncs back)
        } else {
          __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0): this sub-tile's rows have landed ...
          if (sub + SUB < tp) request_rows(sub + SUB, buf ^ 1);   // ... the next one's take off (its buffer was last read two syncs ago)
        }
      } else {
      // genotype rows of the sub-tile -> LDS (coalesced along the row)
        int r = tid % row_len, pi = tid / row_len;
        const int dr = TPC % row_len, dt = TPC / row_len;
        while (pi < ns) {
          s_g[pi * GSS + r] = g[(size_t)s_snp[sub + pi] * row_len + r];
          r += dr; pi += dt;
          if (r >= row_len) { r -= row_len; ++pi; }
        }
      }
      DMX_K2_SYNC();
      // u[pi][l][k] = sum_m pG[1][l][m] g_k[m]  (k < V), and row V = the alpha-0 u of sample 0.  Round 5: the lane's NU entries are formed TOGETHER,
      // branch-free, with the (pair, k) of each entry computed once per kernel — as a rolled loop with a division and a branch per entry this step
      // was a fifth of the kernel (timing builds without it: cfg3 FAST 95 -> 49 ms of 247, cfg5 FAST 29.9 -> 20.4 of 45.8), three exposed LDS round
      // trips per entry for 9 multiply-adds.
      if (!abl_u) {
        if (tid < ns * 3) { const int pi = tid / 3, l = tid - 3 * pi; s_u[(size_t)pi * 3 * VUS + l * VUS + V] = s_u0[(sub + pi) * 4 + l]; }
        double un[NU][3];
#pragma unroll
        for (int i = 0; i < NU; ++i) {
          const int pi = (uq[i] >> 16) & 0x7FFF, k = uq[i] & 0xFFFF;        // (an entry past the sub-tile's SUB * V: pair 0, sample 0, not stored)
          const double* P = &s_q1[(sub + pi) * 6];                 // pG[1][l][m] = P[l + m]
          const float* gr = &s_g[pi * GSS + k * 3];
          const double b0 = (double)gr[0], b1 = (double)gr[1], b2 = (double)gr[2];
#pragma unroll
          for (int l = 0; l < 3; ++l) un[i][l] = __builtin_fma(P[l + 2], b2, __builtin_fma(P[l + 1], b1, P[l] * b0));
        }
#pragma unroll
        for (int i = 0; i < NU; ++i) {
          const int pi = (uq[i] >> 16) & 0x7FFF, k = uq[i] & 0xFFFF;
          if (uq[i] >= 0 && pi < ns) {
            double* u = &s_u[(size_t)pi * 3 * VUS + k];
            u[0] = un[i][0]; u[VUS] = un[i][1]; u[2 * VUS] = un[i][2];
          }
        }
      }
      DMX_K2_SYNC();
      constexpr int UPI = MINW >= 4 ? 1 : SUB;       // pairs unrolled together (register budget)
      // Round-5 experiment (DMX_FAST_PRODUCT builds only; not shipped): a full sub-tile takes ONE log per entry — of the product of its SUB terms
      // (a multiply per term; the log's ~13 instructions and its table read once per SUB terms).  Every term is in [1e-66, 1e78] when k_check_geno
      // has passed the rows, and realistic ones in [1e-6, 1]: a product of four is a normal number, and the result is CLOSER to the exact sum than
      // the reference's.  That is the problem: the reference's llk carries the rounding of its own 50 000 adds (half an ulp of the running sum each,
      // rms 2e-12 at cfg3's depth), which only the same adds of the same terms reproduce — the per-term form agrees with it to 4e-11, this one to 1.3e-9.
      constexpr bool PROD = !CHK && SUB <= 4 && DMX_FAST_PRODUCT;
      constexpr int NB = DMX_SYM_NB > 0 ? DMX_SYM_NB : 1;
      constexpr bool PIPE2 = FIXJ && DMX_SYM_NB > 0 && DMX_FAST_LITE_LOG && !PROD && MINW == 3 && NEP == 0;
      if (PROD && ns == SUB && !abl_p2 && !no_prod) {
        double a[SUB][3];
        if (FIXJ) {
#pragma unroll
          for (int pi = 0; pi < SUB; ++pi) {
            const float* gr = &s_g[pi * GSS + ej[0] * 3];
            a[pi][0] = (double)gr[0]; a[pi][1] = (double)gr[1]; a[pi][2] = (double)gr[2];
          }
        }
#pragma unroll
        for (int i = 0; i < NE; ++i) {
          double pr = 1.0;
#pragma unroll
          for (int pi = 0; pi < SUB; ++pi) {
            if (!FIXJ) {
              const float* gr = &s_g[pi * GSS + ej[i] * 3];
              a[pi][0] = (double)gr[0]; a[pi][1] = (double)gr[1]; a[pi][2] = (double)gr[2];
            }
            const double* up = &s_u[pi * 3 * VUS];
            const double x0 = DMX_LDS_NOMERGE ? lds_read_f64(&up[ek[i]]) : up[ek[i]], x1 = DMX_LDS_NOMERGE ? lds_read_f64(&up[VUS + ek[i]]) : up[VUS + ek[i]],
                         x2 = DMX_LDS_NOMERGE ? lds_read_f64(&up[2 * VUS + ek[i]]) : up[2 * VUS + ek[i]];
            const double sj = __builtin_fma(a[pi][2], x2, __builtin_fma(a[pi][1], x1, a[pi][0] * x0));
            pr = pi ? pr * sj : sj;
          }
          acc[i] += dmx_log2_fastmode(pr, s_log, lk);
        }
      } else if (PIPE2 && ns == SUB && !abl_p2 && !no_pipe) {
        // Round 6: a full sub-tile's SUB x NE evaluations as ONE software-pipelined sequence of steps of NB entries (pair-major, so that every
        // accumulator still adds its terms in ascending pair order: the same operations on the same operands as the loop below — same bits).
        // Left to the compiler, a pair's entries were evaluated two at a time with three EXPOSED LDS round trips per two evaluations (u, then each
        // log's table entry) and only three wavefronts per SIMD to hide them behind: the kernel issued on half of its cycles.  Here step s + 1's
        // u values are requested, and step s's table entries, BEFORE the polynomial and the add of step s - 1 run: no wait in the steady state.
        constexpr int NBAT = (NE + NB - 1) / NB, NST = SUB * NBAT;
        double xu[NB][3];                            // u values of the step whose dot products come next
        double zq[2][NB]; int32_t kq[2][NB]; double2 tq[2][NB];   // per step: reduced argument, binary exponent << 20, table entry {1/c, log c}
        double a0, a1, a2;
        auto load_u = [&](int st) {
          const int pi = st / NBAT, b0 = (st % NBAT) * NB;
          const double* up = &s_u[pi * 3 * VUS];
#pragma unroll
          for (int q = 0; q < NB; ++q) if (b0 + q < NE) {
            xu[q][0] = lds_read_f64(&up[ek[b0 + q]]); xu[q][1] = lds_read_f64(&up[VUS + ek[b0 + q]]); xu[q][2] = lds_read_f64(&up[2 * VUS + ek[b0 + q]]);
          }
        };
        auto load_a = [&](int pi) { const float* gr = &s_g[pi * GSS + ej[0] * 3]; a0 = (double)gr[0]; a1 = (double)gr[1]; a2 = (double)gr[2]; };
        load_a(0);
        load_u(0);
#pragma unroll
        for (int st = 0; st <= NST; ++st) {
          const int cur = st & 1, prv = cur ^ 1;
          if (st < NST) {
            const int b0 = (st % NBAT) * NB;
#pragma unroll
            for (int q = 0; q < NB; ++q) if (b0 + q < NE) {         // dot product, argument reduction, table request (dmx_log2_lite, first half)
              const double sj = __builtin_fma(a2, xu[q][2], __builtin_fma(a1, xu[q][1], a0 * xu[q][0]));
              if (CHK) ok &= __builtin_amdgcn_class(sj, 0x100);
              const uint32_t hi = (uint32_t)__double2hiint(sj), lo = (uint32_t)__double2loint(sj);
              const uint32_t tmp = hi - (kind32(b0 + q) ? DMX_LOG32_OFF_HI : DMX_LOG2_OFF_HI);
              const uint32_t k20 = tmp & 0xFFF00000u;
              zq[cur][q] = __hiloint2double((int)(hi - k20), (int)lo);
              kq[cur][q] = (int32_t)k20;
              if (kind32(b0 + q)) {
                const char* e32 = reinterpret_cast<const char*>(s_log32) + ((tmp >> 12) & 0xF8u);
                tq[cur][q].x = lds_read_f64(reinterpret_cast<const double*>(e32));
                tq[cur][q].y = lds_read_f64(reinterpret_cast<const double*>(e32 + 256));
              } else
              tq[cur][q] = *reinterpret_cast<const double2*>(reinterpret_cast<const char*>(s_log) + ((tmp >> 8) & 0xFF0u));
            }
            if (st + 1 < NST) {
              if ((st + 1) % NBAT == 0) load_a((st + 1) / NBAT);
              load_u(st + 1);
            }
            __builtin_amdgcn_sched_barrier(0);                       // requests first: nothing of the arithmetic below moves above them
          }
          if (st > 0) {
            const int b0 = ((st - 1) % NBAT) * NB;
#pragma unroll
            for (int q = 0; q < NB; ++q) if (b0 + q < NE) {         // dmx_log2_lite, second half, and the ordered add
              const double r = __builtin_fma(zq[prv][q], tq[prv][q].x, -1.0);
#if DMX_SYM_TIMING & 2
              const double w = tq[prv][q].y + lds_read_f64(reinterpret_cast<const double*>(reinterpret_cast<const char*>(s_log) + 1024 + (kq[prv][q] >> 17)));
#else
              const double kd = (double)kq[prv][q];
              const double w = __builtin_fma(kd, 0x1.62e42fefa39efp-1 * 0x1p-20, tq[prv][q].y);
#endif
              double qq;
              if (kind32(b0 + q)) {
                qq = __builtin_fma(r, lk32.a5, lk32.a4);
                qq = __builtin_fma(r, qq, lk32.a3);
                qq = __builtin_fma(r, qq, lk32.a2);
              } else
              qq = __builtin_fma(r, lk.m14, lk.c13);
#if !(DMX_SYM_TIMING & 1)
              qq = __builtin_fma(r, qq, -0.5);
#endif
              qq = __builtin_fma(r, qq, 1.0);
              acc[b0 + q] += __builtin_fma(r, qq, w);
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      } else
#pragma unroll UPI
      for (int pi = 0; pi < SUB; ++pi) {
        if (pi < ns && !abl_p2) {
          const float* gr = &s_g[pi * GSS];
          const double* up = &s_u[pi * 3 * VUS];
          double a0 = 0.0, a1 = 0.0, a2 = 0.0;
          if (FIXJ) { a0 = (double)gr[ej[0] * 3]; a1 = (double)gr[ej[0] * 3 + 1]; a2 = (double)gr[ej[0] * 3 + 2]; }
#pragma unroll
          for (int i = 0; i < NE; ++i) {
            if (!FIXJ) { a0 = (double)gr[ej[i] * 3]; a1 = (double)gr[ej[i] * 3 + 1]; a2 = (double)gr[ej[i] * 3 + 2]; }
            const double x0 = DMX_LDS_NOMERGE ? lds_read_f64(&up[ek[i]]) : up[ek[i]], x1 = DMX_LDS_NOMERGE ? lds_read_f64(&up[VUS + ek[i]]) : up[VUS + ek[i]],
                         x2 = DMX_LDS_NOMERGE ? lds_read_f64(&up[2 * VUS + ek[i]]) : up[2 * VUS + ek[i]];
            const double sj = __builtin_fma(a2, x2, __builtin_fma(a1, x1, a0 * x0));
            if (CHK) ok &= __builtin_amdgcn_class(sj, 0x100);
            acc[i] += kind32(i) ? dmx_log2_lite32(sj, s_log32, lk32) : dmx_log2_fastmode(sj, s_log, lk);
          }
        }
      }
      DMX_K2_SYNC();
    }
  }
  if (cell_ok) {
    double* G = grid + (size_t)cell * V * V * A;
#pragma unroll
    for (int i = 0; i < NE; ++i) {
      const int e = e_base + tid + TPC * i;
      if (e < VD) {
        const int j = ej[i], k = ek[i], d = e / V;
        if (!(2 * d == V && j > k)) {              // d = V/2 is reached from both sides: the j < k lane stores
          G[((size_t)j * V + k) * A + 1] = acc[i];
          G[((size_t)k * V + j) * A + 1] = acc[i];
        }
      } else if (e < VD + V) {
        const int j = ej[i];
        for (int k = 0; k < V; ++k) G[((size_t)j * V + k) * A] = acc[i];
      }
    }
    if (tid < 2 && e_base == 0) l00[(size_t)cell * A + tid] = acc00;
    if (!ok) flag_cell(flagged, cell);
  }
#undef DMX_K2_SYNC
}

// ---------------------------------------------------------------------------------------------------------------------
// K2 STRICT, default grid {0, 0.5}, 32 soft-field samples (cfg3 — the headline): k_doublet_a2's arithmetic with SYMMETRIC ownership (round 6).
// At alpha = 0.5 the mixture is symmetric, pG[1][l][m] == pG[1][m][l] bit for bit (equal mixing weights go through identical operations), so the nine
// terms (g_k[l] g_j[m]) pG[1][l][m] of entry [k][j] are the nine terms T[l][m] = (g_j[l] g_k[m]) pG[1][l][m] of entry [j][k], transposed: the reference adds
// them row-major for [j][k] and column-major for [k][j] — two different roundings of one set of products.  A lane that owns BOTH entries forms the exact
// products E[l][m] = g_j[l] g_k[m] once (they serve both entries at both alphas) and T once: 36 multiplies per unordered pair instead of 54, the 32 adds,
// four logs and four accumulator adds unchanged — 112 FP64 instructions where k_doublet_a2 spends 130, every operation the reference's on its operands
// in its order: the same bits (tests/test_gpu_parity.py::test_symmetric_strict_kernel_gives_k_doublet_a2s_bits).
// Ownership: lane (j = lane & 31, h = lane >> 5) of slab s (blockIdx.y, two per barcode) owns the unordered pairs {j, (j + d) & 31}, d = 1 + 2 (4 s + i) + h,
// i = 0..3 — d = 1..16, d = 16 from j < 16 only — with four accumulators each ([j][k][0], [j][k][1], [k][j][0], [k][j][1]), and slab 0 owns the diagonal:
// lane (j, h) the entry [j][j][h].  496 pairs x 112 + 64 x 37 instructions per covered pair against 1 024 x 65.  One wavefront per (barcode, slab), no
// workgroup barrier: each slab runs the cheap per-tile phases for itself (k_doublet_sym's: headers, rows straight into the LDS one sub-tile ahead, phase 1
// with the final-value table), then its pairs.
template <int SUB, int MINW, int NUP = 4>     // NUP = 0: the diagonal entries only (behind k_doublet_a2u, which owns everything else and llks00)
__global__ __launch_bounds__(kThreads, MINW) void k_doublet_a2s(PileupView pv, int nrd_width, const float* __restrict__ g,
                                                          const double* __restrict__ gp0, const double* __restrict__ tabs,
                                                          const int32_t* __restrict__ sched,
                                                          double* __restrict__ grid, double* __restrict__ l00, const double* __restrict__ pfin) {
  constexpr int V = 32, A = 2, TP = 32, TPC = 64, CPW = kThreads / TPC, T00 = TP + 2, NU = NUP ? NUP : 1, GSS = 3 * V, row_len = 3 * V;
  constexpr bool DIAG_ONLY = NUP == 0;
  __shared__ double s_tab[kTab2];
  __shared__ double s_w[2][10];
  __shared__ __attribute__((aligned(16))) double s_pq_all[CPW][TP][8];       // per pair: alpha 0.5's five distinct values q[l + m] | alpha 0's three q[l]
  __shared__ double s_t00_all[CPW][2][T00];
  __shared__ int64_t s_off_all[CPW][TP];
  __shared__ int32_t s_snp_all[CPW][TP];
  __shared__ uint32_t s_cnt_all[CPW][TP];
  __shared__ __attribute__((aligned(16))) float s_g_all[CPW][2][SUB][GSS];
  const double* s_log = s_tab + kLut2;
  const int t = threadIdx.x;
  stage_k2_tables(s_tab, tabs, t, kThreads);
  if (t < 10) {                                  // the FIVE-form weights (k_doublet_sym): alpha 0: p = 0.5 l (slots 3, 4 repeat l = 2); alpha 0.5: p = 0.25 (l + m), slot l + m
    const int n = t / 5, q = t % 5;
    const int l = n ? (q > 2 ? 2 : q) : min(q, 2), m = n ? q - l : 0;
    const double p = 0.5 * l + (m - l) * 0.5 * (n ? 0.5 : 0.0);
    s_w[n][q] = p;
    s_w[n][5 + q] = 1.0 - p;
  }
  __syncthreads();
  const int cw = t / TPC, tid = t % TPC;
  double* s_pq = &s_pq_all[cw][0][0]; double* s_t00 = &s_t00_all[cw][0][0];
  int64_t* s_off = s_off_all[cw]; int32_t* s_snp = s_snp_all[cw]; uint32_t* s_cnt = s_cnt_all[cw];
  float* s_g0 = &s_g_all[cw][0][0][0];
  const int slot = blockIdx.x * CPW + cw;
  if (slot >= pv.B) return;                      // (no workgroup barrier below)
  const int slab = (int)blockIdx.y;
  const int32_t cell = sched[slot];
  const int64_t p_beg = pv.cell_pair_off[cell];
  const int64_t np = pv.cell_pair_off[cell + 1] - p_beg;
  int64_t rd_base = pv.cell_read_off[cell];
  const int j = tid & 31, h = tid >> 5;
  int kk[NU];
#pragma unroll
  for (int i = 0; i < NU; ++i) kk[i] = (j + 1 + 2 * (NU * slab + i) + h) & 31;
  double acc[NU][4];
#pragma unroll
  for (int i = 0; i < NU; ++i) { acc[i][0] = 0.0; acc[i][1] = 0.0; acc[i][2] = 0.0; acc[i][3] = 0.0; }
  double accd = 0.0, acc00 = 0.0;
  const DmxLogPins lk = dmx_log_pins();
  const int ti1 = tid >> 1, n1 = tid & 1;
  // the diagonal lane's value of (l, m): alpha 0.5 (h = 1) q[l + m] = s_pq[l + m], alpha 0 (h = 0) q[l] = s_pq[5 + l]: index (5 + l) + h (m - 5)
  const int dg[3] = {h * -5, h * -4, h * -3};
  auto request_rows = [&](int sub, int buf) {    // rows of the pairs sub .. sub+SUB-1 of the current tile -> s_g0[buf], asynchronously (k_doublet_sym)
    using gptr = const __attribute__((address_space(1))) void*;
    using lptr = __attribute__((address_space(3))) void*;
#pragma unroll
    for (int pi = 0; pi < SUB; ++pi) {
      const float* src = g + (size_t)__builtin_amdgcn_readfirstlane(s_snp[sub + pi]) * row_len;
#pragma unroll
      for (int hh = 0; hh < (GSS + TPC - 1) / TPC; ++hh) {
        const int r = tid + TPC * hh;
        if (r < row_len) __builtin_amdgcn_global_load_lds((gptr)(src + r), (lptr)(s_g0 + (buf * SUB + pi) * GSS + TPC * hh), 4, 0, 0);
      }
    }
  };
  for (int64_t tbase = 0; tbase < np; tbase += TP) {
    const int tp = (int)min((int64_t)TP, np - tbase);
    if (tid < TP) {
      const bool v = tid < tp;
      const uint32_t n = v ? load_nrd(pv.pair_nrd, p_beg + tbase + tid, nrd_width) : 0u;
      const int32_t sn = v ? (pv.pair_snp ? pv.pair_snp[p_beg + tbase + tid] : (int32_t)(tbase + tid)) : 0;
      const uint32_t incl = seg_scan_incl<TP>(n);
      s_cnt[tid] = n;
      s_off[tid] = rd_base + (int64_t)(incl - n);
      s_snp[tid] = sn;
    }
    DMX_WAVE_LDS_ORDER();
    rd_base = s_off[tp - 1] + (int64_t)s_cnt[tp - 1];
    request_rows(0, 0);                            // sub-tile 0's rows travel while phase 1 runs
    // ---- phase 1 (:597-663), k_doublet_sym's: lane (pair ti1, alpha n1), five distinct values
    {
      const bool on = ti1 < tp;
      const uint32_t cnt = on ? s_cnt[ti1] : 0u;
      const int64_t off = on ? s_off[ti1] : 0;
      const uint32_t rd4 = load_rd4(pv, off, cnt);
      const int32_t snp1 = on ? s_snp[ti1] : 0;
      double q[5];
      const int32_t fi = pfin ? certify_final_index(cnt, rd4) : -1;
      if (!__any(fi < 0)) {
        const double* fp = pfin + (size_t)fi * kCFinStride + (n1 ? 0 : 5);
        q[0] = fp[0]; q[1] = fp[1]; q[2] = fp[2];
        q[3] = n1 ? fp[3] : 0.0; q[4] = n1 ? fp[4] : 0.0;
      } else {
        double wA[5], wR[5];
#pragma unroll
        for (int i = 0; i < 5; ++i) { q[i] = 1.0; wA[i] = s_w[n1][i]; wR[i] = s_w[n1][5 + i]; }   // :597
        for (uint32_t r = 0; __any(r < cnt); ++r) {
          const bool live = r < cnt;
          const uint32_t byte = live ? (r < 4 ? (rd4 >> (8 * r)) & 0xFFu : (uint32_t)pv.reads[off + r]) : 0u;
          const uint32_t bq = byte & 127u;
          const bool alt = (byte >> 7) != 0;
          const double pR = alt ? s_tab[128 + bq] : s_tab[bq];                // :606
          const double pA = alt ? s_tab[bq] : s_tab[128 + bq];                // :607
          double mx = 0.0;
          if (live) {
#pragma unroll
            for (int i = 0; i < 5; ++i) {
              q[i] *= (pR * wR[i] + pA * wA[i]);                              // :625
              mx = fmax(mx, q[i]);                                            // :626-627
            }
          }
          {
            const double o = shfl_xor1(mx);                                   // one max across both alphas of the pair
            mx = fmax(mx, o);
          }
          if (live) {
            if (cnt <= kSafeReads) {
              const double y = rcp_refined(mx);
#pragma unroll
              for (int i = 0; i < 5; ++i) q[i] = div_by(q[i], mx, y);         // :632-639
            } else {
#pragma unroll
              for (int i = 0; i < 5; ++i) q[i] = div_slow(q[i], mx);
            }
          }
        }
        double mx = 0.0;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
          q[i] += 1e-6;                                                        // :649
          mx = fmax(mx, q[i]);
        }
        {
          const double o = shfl_xor1(mx);
          mx = fmax(mx, o);
        }
        const double y = rcp_refined(mx);
#pragma unroll
        for (int i = 0; i < 5; ++i) q[i] = div_by(q[i], mx, y);                // :656-663
      }
      if (on) {
        const double* g0 = gp0 + (size_t)snp1 * 3;
        const double qq[3] = {g0[0], g0[1], g0[2]};
        double sum = 0.0;
#pragma unroll
        for (int l = 0; l < 3; ++l)
#pragma unroll
          for (int m = 0; m < 3; ++m) {
            const double v = n1 ? q[l + m] : q[l];                           // pG[n][l][m]
            sum += ((qq[l] * qq[m]) * v);                                    // gp00 (:555) then :702-705
          }
        s_t00[n1 * T00 + ti1] = dmx_log2_fast(sum, s_log);                    // :708-709 term
        if (n1) {
#pragma unroll
          for (int i = 0; i < 5; ++i) s_pq[ti1 * 8 + i] = q[i];
        } else {
#pragma unroll
          for (int l = 0; l < 3; ++l) s_pq[ti1 * 8 + 5 + l] = q[l];
        }
      }
    }
    DMX_WAVE_LDS_ORDER();
    if (!DIAG_ONLY && tid < 2 && slab == 0) {      // llks00: lane n adds its alpha's terms in pair order
      const double* row = &s_t00[tid * T00];
      for (int i = 0; i < tp; ++i) acc00 += row[i];
    }
    // ---- phase 2 in sub-tiles of SUB pairs
#pragma unroll 1
    for (int sub = 0; sub < tp; sub += SUB) {
      const int ns = min(SUB, tp - sub);
      const int buf = (sub / SUB) & 1;
      const float* s_g = s_g0 + buf * SUB * GSS;
      __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0): this sub-tile's rows have landed ...
      if (sub + SUB < tp) request_rows(sub + SUB, buf ^ 1);   // ... the next one's take off (its buffer was last read two syncs ago)
      DMX_WAVE_LDS_ORDER();
#pragma unroll 1
      for (int pi = 0; pi < ns; ++pi) {                    // (two pairs per trip: 1 227.8 against 1 206.2 ms)
        const float* gr = &s_g[pi * GSS];
        const double* pq = &s_pq[(sub + pi) * 8];
        const double a[3] = {(double)gr[j * 3], (double)gr[j * 3 + 1], (double)gr[j * 3 + 2]};
        constexpr bool ROOMY = MINW <= 3;          // 168 registers: the pair's q values stay in registers, the next unit's row is read a unit ahead, the four logs interleave
        double Pr[5], Qr[3];
        if (ROOMY) {
#pragma unroll
          for (int c = 0; c < 5; ++c) Pr[c] = pq[c];
#pragma unroll
          for (int c = 0; c < 3; ++c) Qr[c] = pq[5 + c];
        }
        float bn[3] = {gr[kk[0] * 3], gr[kk[0] * 3 + 1], gr[kk[0] * 3 + 2]};
#pragma unroll
        for (int i = 0; i < (DIAG_ONLY ? 0 : NU); ++i) {
          const double b[3] = {(double)bn[0], (double)bn[1], (double)bn[2]};
          if (ROOMY && i + 1 < NU) { const int kn = kk[i + 1]; bn[0] = gr[kn * 3]; bn[1] = gr[kn * 3 + 1]; bn[2] = gr[kn * 3 + 2]; }
          else if (!ROOMY && i + 1 < NU) { const int kn = kk[i + 1]; bn[0] = gr[kn * 3]; bn[1] = gr[kn * 3 + 1]; bn[2] = gr[kn * 3 + 2]; }
          // two halves so that no more than the nine products, two sums and two logs are live at once (128 registers = four wavefronts per SIMD): first
          // the exact products E and the two alpha-0 entries, then E becomes T = E q[l + m] in place and serves the two alpha-0.5 entries
          double E[3][3];
#pragma unroll
          for (int l = 0; l < 3; ++l)
#pragma unroll
            for (int m = 0; m < 3; ++m) E[l][m] = a[l] * b[m];                  // :553 (exact) — shared by [j][k] and [k][j] at both alphas
          {
            // [j][k][0]: l-major over (l, m) of (g_j[l] g_k[m]) q0[l]; [k][j][0]: of (g_k[l] g_j[m]) q0[l] = E[m][l] q0[l].  The first product initialises
            // the sum (0 + x == x for these non-negative products: the reference's bits, one add fewer).  (The pair's q values are re-read per half —
            // uniform LDS reads — instead of held across the pair's units: 16 registers the kernel does not have at four wavefronts per SIMD.)
            const double Q[3] = {ROOMY ? Qr[0] : lds_read_f64(&pq[5]), ROOMY ? Qr[1] : lds_read_f64(&pq[6]), ROOMY ? Qr[2] : lds_read_f64(&pq[7])};
            double s_jk0 = E[0][0] * Q[0], s_kj0 = E[0][0] * Q[0];
#pragma unroll
            for (int l = 0; l < 3; ++l)
#pragma unroll
              for (int m = 0; m < 3; ++m) {
                if (l == 0 && m == 0) continue;
                s_jk0 += (E[l][m] * Q[l]);
                s_kj0 += (E[m][l] * Q[l]);
              }
            acc[i][0] += dmx_log2_fast_pinned(s_jk0, s_log, lk);               // :683
            acc[i][2] += dmx_log2_fast_pinned(s_kj0, s_log, lk);
          }
          if (!ROOMY) __builtin_amdgcn_sched_barrier(0);
          {
            const double P[5] = {ROOMY ? Pr[0] : lds_read_f64(&pq[0]), ROOMY ? Pr[1] : lds_read_f64(&pq[1]), ROOMY ? Pr[2] : lds_read_f64(&pq[2]),
                                 ROOMY ? Pr[3] : lds_read_f64(&pq[3]), ROOMY ? Pr[4] : lds_read_f64(&pq[4])};
#pragma unroll
            for (int l = 0; l < 3; ++l)
#pragma unroll
              for (int m = 0; m < 3; ++m) E[l][m] = E[l][m] * P[l + m];         // T: :677-679 at alpha 0.5 — [k][j]'s nine terms are these, transposed
            double s_jk1 = E[0][0], s_kj1 = E[0][0];
#pragma unroll
            for (int l = 0; l < 3; ++l)
#pragma unroll
              for (int m = 0; m < 3; ++m) {
                if (l == 0 && m == 0) continue;
                s_jk1 += E[l][m];
                s_kj1 += E[m][l];
              }
            acc[i][1] += dmx_log2_fast_pinned(s_jk1, s_log, lk);
            acc[i][3] += dmx_log2_fast_pinned(s_kj1, s_log, lk);
          }
          __builtin_amdgcn_sched_barrier(0);       // one pair of entries at a time
        }
        if (slab == 0) {                           // the diagonal entry [j][j][h]
          double sd = 0.0;
#pragma unroll
          for (int l = 0; l < 3; ++l)
#pragma unroll
            for (int m = 0; m < 3; ++m) {
              const double v = pq[5 + l + dg[m]];
              const double term = (a[l] * a[m]) * v;
              sd = (l == 0 && m == 0) ? term : sd + term;
            }
          accd += dmx_log2_fast_pinned(sd, s_log, lk);
        }
      }
      DMX_WAVE_LDS_ORDER();
    }
  }
  double* G = grid + (size_t)cell * V * V * A;
#pragma unroll
  for (int i = 0; i < (DIAG_ONLY ? 0 : NU); ++i) {
    const int d = 1 + 2 * (NU * slab + i) + h, k = kk[i];
    if (d < 16 || j < 16) {                        // d = 16 is reached from both sides: the j < 16 lane stores
      G[((size_t)j * V + k) * A] = acc[i][0]; G[((size_t)j * V + k) * A + 1] = acc[i][1];
      G[((size_t)k * V + j) * A] = acc[i][2]; G[((size_t)k * V + j) * A + 1] = acc[i][3];
    }
  }
  if (slab == 0) {
    G[((size_t)j * V + j) * A + h] = accd;
    if (!DIAG_ONLY && tid < 2) l00[(size_t)cell * A + tid] = acc00;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// The diagonal of k_doublet_a2u's grid: entries [j][j][0..1] of 32 samples (STRICT, default grid).  Lane j of a barcode's 32 lanes owns BOTH alphas of its
// sample — the six distinct exact products g_j[l] g_j[m] once, two nine-term l-major sums, two logs, two accumulators — and a wavefront carries two barcodes
// (k_doublet_a2s<.., 0>, the first form, spent a 64-lane wavefront per barcode: lane (j, alpha), 62 instructions per pair; this one 83 per pair for two
// barcodes).  Per tile of 16 pairs and barcode: headers (16 lanes), phase 1 (32 lanes = pair x alpha; k_doublet_sym's, from the final-value table where
// every pair of the wavefront's two tiles is on it), then the lane's own three floats per pair straight from the matrix.  No workgroup barrier.
template <int MINW, int VP = 32>                // VP: the panel — 32 samples (two barcodes per wavefront) or 16 (four)
__global__ __launch_bounds__(kThreads, MINW) void k_doublet_diag(PileupView pv, int nrd_width, const float* __restrict__ g,
                                                           const double* __restrict__ tabs, const int32_t* __restrict__ sched,
                                                           double* __restrict__ grid, const double* __restrict__ pfin, const double* __restrict__ pseed) {
  static_assert(VP == 32 || VP == 16, "panel");
  constexpr int V = VP, A = 2, TPC = VP, TP = VP / 2, CPW = kThreads / TPC, row_len = 3 * V, PB = 8, WPC = 64 / TPC;
  __shared__ double s_tab[kTab2];
  __shared__ double s_w[2][10];
  __shared__ __attribute__((aligned(16))) double s_pq_all[CPW][TP][8];
  __shared__ int64_t s_off_all[CPW][TP];
  __shared__ int32_t s_snp_all[CPW][TP];
  __shared__ uint32_t s_cnt_all[CPW][TP];
  const double* s_log = s_tab + kLut2;
  const int t = threadIdx.x;
  stage_k2_tables(s_tab, tabs, t, kThreads);
  if (t < 10) {
    const int n = t / 5, q = t % 5;
    const int l = n ? (q > 2 ? 2 : q) : min(q, 2), m = n ? q - l : 0;
    const double p = 0.5 * l + (m - l) * 0.5 * (n ? 0.5 : 0.0);
    s_w[n][q] = p;
    s_w[n][5 + q] = 1.0 - p;
  }
  __syncthreads();
  const int cw = t / TPC, tid = t % TPC;
  double* s_pq = &s_pq_all[cw][0][0];
  int64_t* s_off = s_off_all[cw]; int32_t* s_snp = s_snp_all[cw]; uint32_t* s_cnt = s_cnt_all[cw];
  const int slot = blockIdx.x * CPW + cw;
  if ((blockIdx.x * CPW + (cw & ~(WPC - 1))) >= pv.B) return;   // every barcode of the wavefront past the end (no workgroup barrier below)
  const bool cell_ok = slot < pv.B;
  const int32_t cell = cell_ok ? sched[slot] : 0;
  const int64_t p_beg = cell_ok ? pv.cell_pair_off[cell] : 0;
  const int64_t np = cell_ok ? pv.cell_pair_off[cell + 1] - p_beg : 0;
  int64_t rd_base = cell_ok ? pv.cell_read_off[cell] : 0;
  int64_t np_w = np;                               // the longer of the wavefront's two barcodes: the tile loop is wave-uniform
#pragma unroll
  for (int d = TPC; d < 64; d <<= 1) { const int64_t o = __shfl_xor(np_w, d); np_w = np_w > o ? np_w : o; }
  const int j = tid;
  double acc0 = 0.0, acc1 = 0.0;
  const DmxLogPins lk = dmx_log_pins();
  const int ti1 = tid >> 1, n1 = tid & 1;
  for (int64_t tbase = 0; tbase < np_w; tbase += TP) {
    const int tp = (int)max((int64_t)0, min((int64_t)TP, np - tbase));
    if (tid < TP) {
      const bool v = tid < tp;
      const uint32_t n = v ? load_nrd(pv.pair_nrd, p_beg + tbase + tid, nrd_width) : 0u;
      const int32_t sn = v ? (pv.pair_snp ? pv.pair_snp[p_beg + tbase + tid] : (int32_t)(tbase + tid)) : 0;
      const uint32_t incl = seg_scan_incl<TP>(n);
      s_cnt[tid] = n;
      s_off[tid] = rd_base + (int64_t)(incl - n);
      s_snp[tid] = sn;
    }
    DMX_WAVE_LDS_ORDER();
    if (tp > 0) rd_base = s_off[tp - 1] + (int64_t)s_cnt[tp - 1];
    // ---- phase 1 (:597-663), k_doublet_sym's: lane (pair ti1, alpha n1), five distinct values
    {
      const bool on = ti1 < tp;
      const uint32_t cnt = on ? s_cnt[ti1] : 0u;
      const int64_t off = on ? s_off[ti1] : 0;
      const uint32_t rd4 = load_rd4(pv, off, cnt);
      double q[5];
      const int32_t fi = pfin ? certify_final_index(cnt, rd4) : -1;
      if (!__any(fi < 0)) {
        const double* fp = pfin + (size_t)fi * kCFinStride + (n1 ? 0 : 5);
        q[0] = fp[0]; q[1] = fp[1]; q[2] = fp[2];
        q[3] = n1 ? fp[3] : 0.0; q[4] = n1 ? fp[4] : 0.0;
      } else {
        double wA[5], wR[5];
#pragma unroll
        for (int i = 0; i < 5; ++i) { wA[i] = s_w[n1][i]; wR[i] = s_w[n1][5 + i]; }
        certify_pair_values<5, false>(pv, cnt, off, rd4, s_tab, wA, wR, n1, q, pseed);
      }
      if (on) {
        if (n1) {
#pragma unroll
          for (int i = 0; i < 5; ++i) s_pq[ti1 * 8 + i] = q[i];
        } else {
#pragma unroll
          for (int l = 0; l < 3; ++l) s_pq[ti1 * 8 + 5 + l] = q[l];
        }
      }
    }
    DMX_WAVE_LDS_ORDER();
    // ---- phase 2: the lane's sample against itself, PB pairs' rows requested at once
#pragma unroll 1
    for (int sub = 0; sub < TP; sub += PB) {
      if (!__any(sub < tp)) break;
      float ar[PB][3];
#pragma unroll
      for (int pi = 0; pi < PB; ++pi) {
        const float* src = g + (size_t)s_snp[sub + pi] * row_len + j * 3;     // (entries past the tile's last pair hold SNP 0)
        ar[pi][0] = src[0]; ar[pi][1] = src[1]; ar[pi][2] = src[2];
      }
#pragma unroll
      for (int pi = 0; pi < PB; ++pi) {
        if (sub + pi < tp) {
          const double* pq = &s_pq[(sub + pi) * 8];
          const double a[3] = {(double)ar[pi][0], (double)ar[pi][1], (double)ar[pi][2]};
          const double P[5] = {pq[0], pq[1], pq[2], pq[3], pq[4]}, Q[3] = {pq[5], pq[6], pq[7]};
          double s0 = 0.0, s1 = 0.0;
#pragma unroll
          for (int l = 0; l < 3; ++l)
#pragma unroll
            for (int m = 0; m < 3; ++m) {
              const double e = a[l] * a[m];                                   // :553 (exact; a[l] a[m] == a[m] a[l]: six distinct)
              const double t0 = e * Q[l], t1 = e * P[l + m];                  // :677-679 at alpha 0 (pG = q0[l]) and alpha 0.5 (pG = q1[l + m])
              s0 = (l == 0 && m == 0) ? t0 : s0 + t0;                         // l-major; the first product initialises the sum (0 + x == x)
              s1 = (l == 0 && m == 0) ? t1 : s1 + t1;
            }
          acc0 += dmx_log2_fast_pinned(s0, s_log, lk);                        // :683
          acc1 += dmx_log2_fast_pinned(s1, s_log, lk);
        }
      }
    }
    DMX_WAVE_LDS_ORDER();
  }
  if (cell_ok) {
    double* o = grid + (((size_t)cell * V + j) * V + j) * A;
    o[0] = acc0; o[1] = acc1;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// K2 for alpha grids of 3..8 entries: k_doublet_a2's ownership, order and arithmetic with AP (= A rounded up to 2, 4 or 8)
// alphas per pair.  Phase 1 spreads the TP * AP (pair, alpha) lanes over all the cell's threads, in passes when the cell has
// fewer than that; the one-max-across-ALL-alphas renormalisation (:626-639) is a butterfly over the AP lanes of a pair.
// Phase 2 walks the alphas two at a time with their pG in registers, so an evaluation costs what it costs in k_doublet_a2.
template <int TPC, int NK, int AP>
__global__ __launch_bounds__(kThreads) void k_doublet_an(PileupView pv, int nrd_width, const float* __restrict__ g,
                                                         const double* __restrict__ gp0, const double* __restrict__ tabs,
                                                         const double* __restrict__ alpha,
                                                         const int32_t* __restrict__ sched, int32_t V, int32_t A, int32_t GS,
                                                         double* __restrict__ grid, double* __restrict__ l00,
                                                         uint8_t* __restrict__ flagged) {
  static_assert(AP == 2 || AP == 4 || AP == 8, "alphas per pair padded to a power of two");
  constexpr int TP = 32;
  constexpr int CPW = kThreads / TPC;
  constexpr int T00 = TP + 2;
#define DMX_K2_SYNC() do { if (TPC == 64) { DMX_WAVE_LDS_ORDER(); } else { __syncthreads(); } } while (0)
  extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
  __shared__ double s_tab[kTab2];
  const double* s_log = s_tab + kLut2;
  const int t = threadIdx.x;
  stage_k2_tables(s_tab, tabs, t, kThreads);
  __syncthreads();

  const int cw = t / TPC, tid = t % TPC;
  const size_t cell_bytes = (size_t)TP * AP * 9 * 8 + (size_t)TP * GS * 4 + (size_t)AP * T00 * 8 + TP * (4 + 4 + 8);
  unsigned char* base = s_raw + (size_t)cw * cell_bytes;
  double* s_pG = (double*)base;                                  // [TP][AP][9]
  float* s_g = (float*)(base + (size_t)TP * AP * 9 * 8);         // [TP][GS]
  double* s_t00 = (double*)((unsigned char*)s_g + (size_t)TP * GS * 4);   // [AP][T00]
  int64_t* s_off = (int64_t*)(s_t00 + AP * T00);                 // [TP]
  int32_t* s_snp = (int32_t*)(s_off + TP);                       // [TP]
  uint32_t* s_cnt = (uint32_t*)(s_snp + TP);                     // [TP]

  const int slot = blockIdx.x * CPW + cw;
  if (TPC == 64 && slot >= pv.B) return;
  const bool cell_ok = slot < pv.B;
  const int32_t cell = cell_ok ? sched[slot] : 0;
  const int64_t p_beg = cell_ok ? pv.cell_pair_off[cell] : 0;
  const int64_t np = cell_ok ? pv.cell_pair_off[cell + 1] - p_beg : 0;
  int64_t rd_base = cell_ok ? pv.cell_read_off[cell] : 0;

  const int KB = (V + NK - 1) / NK;
  const int JS = TPC / KB;
  const int jl = tid / KB, kb = tid % KB;
  const int j = (int)blockIdx.y * JS + jl;
  const bool owner = jl < JS && j < V;
  double acc[NK][AP];
#pragma unroll
  for (int kk = 0; kk < NK; ++kk)
#pragma unroll
    for (int n = 0; n < AP; ++n) acc[kk][n] = 0.0;
  bool ok = true;
  double acc00 = 0.0;                            // lane tid < A owns llks00[tid]
  const int row_len = V * 3;
  constexpr int P1 = TP * AP;                    // phase-1 lanes per tile
  constexpr int NPASS = (P1 + TPC - 1) / TPC;
  const int n1 = tid % AP;                       // this thread's alpha in phase 1 (TPC is a multiple of AP)
  const bool n_ok = n1 < A;
  double wA[9], wR[9];
  {
    const double al = n_ok ? alpha[n1] : 0.0;
#pragma unroll
    for (int l = 0; l < 3; ++l)
#pragma unroll
      for (int m = 0; m < 3; ++m) {
        const double p = 0.5 * l + (m - l) * 0.5 * al;                       // :613
        wA[l * 3 + m] = p;
        wR[l * 3 + m] = 1.0 - p;
      }
  }

  for (int64_t tbase = 0; tbase < np; tbase += TP) {
    const int tp = (int)min((int64_t)TP, np - tbase);
    if (tid < TP) {
      const bool v = tid < tp;
      const uint32_t n = v ? load_nrd(pv.pair_nrd, p_beg + tbase + tid, nrd_width) : 0u;
      const uint32_t incl = seg_scan_incl<32>(n);
      s_cnt[tid] = n;
      s_off[tid] = rd_base + (int64_t)(incl - n);
      s_snp[tid] = v ? (pv.pair_snp ? pv.pair_snp[p_beg + tbase + tid] : (int32_t)(tbase + tid)) : 0;
    }
    DMX_K2_SYNC();
    rd_base = s_off[tp - 1] + (int64_t)s_cnt[tp - 1];
    {
      int r = tid % row_len, ti = tid / row_len;
      const int dr = TPC % row_len, dt = TPC / row_len;
      while (ti < tp) {
        s_g[ti * GS + r] = g[(size_t)s_snp[ti] * row_len + r];
        r += dr; ti += dt;
        if (r >= row_len) { r -= row_len; ++ti; }
      }
    }
    // ---- phase 1: lane u = (pair u / AP, alpha u % AP)
#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass) {
      const int u = tid + pass * TPC;
      if (u >= P1) break;                          // uniform per wavefront (P1 and TPC are multiples of 64)
      const int ti1 = u / AP;
      const bool on = ti1 < tp && n_ok;
      const uint32_t cnt = (ti1 < tp) ? s_cnt[ti1] : 0u;
      const int64_t off = (ti1 < tp) ? s_off[ti1] : 0;
      double pG[9];
#pragma unroll
      for (int i = 0; i < 9; ++i) pG[i] = 1.0;                               // :597
      for (uint32_t r = 0; __any(r < cnt); ++r) {
        const bool live = r < cnt && n_ok;
        const uint32_t byte = (r < cnt) ? pv.reads[off + r] : 0u;
        const uint32_t bq = byte & 127u;
        const bool alt = (byte >> 7) != 0;
        const double pR = alt ? s_tab[128 + bq] : s_tab[bq];                // :606
        const double pA = alt ? s_tab[bq] : s_tab[128 + bq];                // :607
        double mx = 0.0;
        if (live) {
#pragma unroll
          for (int i = 0; i < 9; ++i) {
            pG[i] *= (pR * wR[i] + pA * wA[i]);                             // :625
            mx = fmax(mx, pG[i]);                                 // :626-627
          }
        }
#pragma unroll
        for (int d = 1; d < AP; d <<= 1) {                                  // one max across ALL alphas of the pair
          const double o = __shfl_xor(mx, d);
          mx = fmax(mx, o);
        }
        if (live) {
          if (cnt <= kSafeReads) {
            const double y = rcp_refined(mx);
#pragma unroll
            for (int i = 0; i < 9; ++i) pG[i] = div_by(pG[i], mx, y);       // :632-639
          } else {
#pragma unroll
            for (int i = 0; i < 9; ++i) pG[i] /= mx;
          }
        }
      }
      double mx = 0.0;
      if (n_ok) {
#pragma unroll
        for (int i = 0; i < 9; ++i) {
          pG[i] += 1e-6;                                                     // :649
          mx = fmax(mx, pG[i]);
        }
      }
#pragma unroll
      for (int d = 1; d < AP; d <<= 1) {
        const double o = __shfl_xor(mx, d);
        mx = fmax(mx, o);
      }
      if (on) {
        const double y = rcp_refined(mx);
        const double* g0 = gp0 + (size_t)s_snp[ti1] * 3;
        const double qq[3] = {g0[0], g0[1], g0[2]};
        double sum = 0.0;
#pragma unroll
        for (int l = 0; l < 3; ++l)
#pragma unroll
          for (int m = 0; m < 3; ++m) {
            const double v = div_by(pG[l * 3 + m], mx, y);                   // :656-663
            s_pG[(ti1 * AP + n1) * 9 + l * 3 + m] = v;
            sum += ((qq[l] * qq[m]) * v);                                    // gp00 (:555) then :702-705
          }
        ok &= __builtin_amdgcn_class(sum, 0x100);
        s_t00[n1 * T00 + ti1] = dmx_log2_fast(sum, s_log);                    // :708-709 term
      }
    }
    DMX_K2_SYNC();
    if (tid < A) {                                 // llks00[n]: lane n adds its alpha's terms in pair order
      const double* row = &s_t00[tid * T00];
      for (int i = 0; i < tp; ++i) acc00 += row[i];
    }
    // ---- phase 2
    if (owner) {
      for (int ti = 0; ti < tp; ++ti) {
        const float* gr = &s_g[ti * GS];
        const double aj[3] = {(double)gr[j * 3], (double)gr[j * 3 + 1], (double)gr[j * 3 + 2]};
#pragma unroll
        for (int ap = 0; ap < AP / 2; ++ap) {
          if (2 * ap >= A) break;
          const double* P = &s_pG[(ti * AP + 2 * ap) * 9];
          if (2 * ap + 1 < A) {                    // two alphas share the nine products
            double P0[9], P1v[9];
#pragma unroll
            for (int i = 0; i < 9; ++i) { P0[i] = P[i]; P1v[i] = P[9 + i]; }
#pragma unroll
            for (int kk = 0; kk < NK; ++kk) {
              const int k = min(kb * NK + kk, V - 1);
              const double bk[3] = {(double)gr[k * 3], (double)gr[k * 3 + 1], (double)gr[k * 3 + 2]};
              double s0 = 0.0, s1 = 0.0;                                      // :674
#pragma unroll
              for (int l = 0; l < 3; ++l)
#pragma unroll
                for (int m = 0; m < 3; ++m) {
                  const double gp = aj[l] * bk[m];                            // :553 (exact)
                  s0 += (gp * P0[l * 3 + m]);                                 // :677-679, l-major
                  s1 += (gp * P1v[l * 3 + m]);
                }
              ok &= __builtin_amdgcn_class(s0, 0x100) && __builtin_amdgcn_class(s1, 0x100);
              acc[kk][2 * ap] += dmx_log2_fast(s0, s_log);                     // :683
              acc[kk][2 * ap + 1] += dmx_log2_fast(s1, s_log);
            }
          } else {                                 // the last alpha of an odd-sized grid
            double P0[9];
#pragma unroll
            for (int i = 0; i < 9; ++i) P0[i] = P[i];
#pragma unroll
            for (int kk = 0; kk < NK; ++kk) {
              const int k = min(kb * NK + kk, V - 1);
              const double bk[3] = {(double)gr[k * 3], (double)gr[k * 3 + 1], (double)gr[k * 3 + 2]};
              double s0 = 0.0;
#pragma unroll
              for (int l = 0; l < 3; ++l)
#pragma unroll
                for (int m = 0; m < 3; ++m) s0 += ((aj[l] * bk[m]) * P0[l * 3 + m]);
              ok &= __builtin_amdgcn_class(s0, 0x100);
              acc[kk][2 * ap] += dmx_log2_fast(s0, s_log);
            }
          }
        }
      }
    }
    DMX_K2_SYNC();
  }
  if (cell_ok) {
    if (owner) {
#pragma unroll
      for (int kk = 0; kk < NK; ++kk) {
        const int k = kb * NK + kk;
        if (k < V) {
          double* o = grid + (((size_t)cell * V + j) * V + k) * A;
#pragma unroll
          for (int n = 0; n < AP; ++n) if (n < A) o[n] = acc[kk][n];
        }
      }
    }
    if (tid < A && blockIdx.y == 0) l00[(size_t)cell * A + tid] = acc00;
    if (!ok) flag_cell(flagged, cell);
  }
#undef DMX_K2_SYNC
}

// FAST mode for any alpha grid with alpha[0] == 0 (soft fields; the default grid {0, 0.5} has its own kernel, k_doublet_sym):
// k_doublet_an's ownership, phase 1 and accumulation order with the entry set demuxlet prints or decides on — the singlet column
// llksAB[j][0][0] and every (j, k) of the doublet alphas n >= 1 (cmd_cram_demuxlet.cpp:772-797,:799-814) — in the bilinear form
// g_j . (pG[n] g_k), u = pG[n] g_k formed once per (pair, n, k) in the LDS and shared by the V rows j.  llksAB[j][k != 0][0], which
// only the maxLLK scan reads (:713-721), is filled with llksAB[j][0][0].  A * V * V evaluations of 33.5 FP64 instructions become
// (A - 1) * V * V + V of 15.
template <int TPC, int NK, int AP, int VUS, int SUBP = 4, int MINW = 2, bool CHK = true>   // VUS: row stride of u in doubles (>= V + NK - 1,
                                                               // even): compile-time so that the reads use immediate offsets; CHK: see k_doublet_a2
__global__ __launch_bounds__(kThreads, MINW) void k_doublet_anf(PileupView pv, int nrd_width, const float* __restrict__ g,
                                                         const double* __restrict__ gp0, const double* __restrict__ tabs,
                                                         const double* __restrict__ alpha,
                                                         const int32_t* __restrict__ sched, int32_t V, int32_t A, int32_t GS,
                                                         double* __restrict__ grid, double* __restrict__ l00,
                                                         uint8_t* __restrict__ flagged) {
  static_assert(AP == 2 || AP == 4 || AP == 8, "alphas per pair padded to a power of two");
  constexpr int TP = 32;
  constexpr int CPW = kThreads / TPC;
  constexpr int T00 = TP + 2;
#define DMX_K2_SYNC() do { if (TPC == 64) { DMX_WAVE_LDS_ORDER(); } else { __syncthreads(); } } while (0)
  extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
  __shared__ double s_tab[kTab2];
  const double* s_log = s_tab + kLut2;
  const int t = threadIdx.x;
  stage_k2_tables(s_tab, tabs, t, kThreads);
  __syncthreads();

  const int cw = t / TPC, tid = t % TPC;
  constexpr int VU = VUS;
  const int NU = (A - 1) * 3 * VU + 4;           // doubles of u per pair: [n-1][l][k], then the alpha-0 u of sample 0 (3) + pad
  const size_t cell_bytes = (size_t)TP * AP * 9 * 8 + (size_t)TP * GS * 4 + (size_t)AP * T00 * 8 + TP * (4 + 4 + 8) +
                            (size_t)SUBP * NU * 8 + (size_t)TPC * 8;
  unsigned char* base = s_raw + (size_t)cw * cell_bytes;
  double* s_pG = (double*)base;                                  // [TP][AP][9]
  float* s_g = (float*)(base + (size_t)TP * AP * 9 * 8);         // [TP][GS]
  double* s_t00 = (double*)((unsigned char*)s_g + (size_t)TP * GS * 4);   // [AP][T00]
  int64_t* s_off = (int64_t*)(s_t00 + AP * T00);                 // [TP]
  int32_t* s_snp = (int32_t*)(s_off + TP);                       // [TP]
  uint32_t* s_cnt = (uint32_t*)(s_snp + TP);                     // [TP]
  double* s_u = (double*)(s_cnt + TP);                           // [SUBP][NU]
  double* s_sing = s_u + (size_t)SUBP * NU;                      // [TPC]  the singlet column at the end (row j -> its k-block threads)

  const int slot = blockIdx.x * CPW + cw;
  if (TPC == 64 && slot >= pv.B) return;
  const bool cell_ok = slot < pv.B;
  const int32_t cell = cell_ok ? sched[slot] : 0;
  const int64_t p_beg = cell_ok ? pv.cell_pair_off[cell] : 0;
  const int64_t np = cell_ok ? pv.cell_pair_off[cell + 1] - p_beg : 0;
  int64_t rd_base = cell_ok ? pv.cell_read_off[cell] : 0;

  const int KB = (V + NK - 1) / NK;
  const int JS = TPC / KB;
  const int jl = tid / KB, kb = tid % KB;
  const int j = (int)blockIdx.y * JS + jl;
  const bool owner = jl < JS && j < V;
  double acc[NK][AP - 1], accS = 0.0;            // [k of the block][alpha n >= 1]; accS: llksAB[j][0][0] (threads kb == 0)
#pragma unroll
  for (int kk = 0; kk < NK; ++kk)
#pragma unroll
    for (int n = 0; n < AP - 1; ++n) acc[kk][n] = 0.0;
  bool ok = true;
  const DmxLogPins lk = dmx_log_pins();
  double acc00 = 0.0;                            // lane tid < A owns llks00[tid]
  const int row_len = V * 3;
  constexpr int P1 = TP * AP;                    // phase-1 lanes per tile
  constexpr int NPASS = (P1 + TPC - 1) / TPC;
  const int n1 = tid % AP;                       // this thread's alpha in phase 1 (TPC is a multiple of AP)
  const bool n_ok = n1 < A;
  double wA[9], wR[9];
  {
    const double al = n_ok ? alpha[n1] : 0.0;
#pragma unroll
    for (int l = 0; l < 3; ++l)
#pragma unroll
      for (int m = 0; m < 3; ++m) {
        const double p = 0.5 * l + (m - l) * 0.5 * al;                       // :613
        wA[l * 3 + m] = p;
        wR[l * 3 + m] = 1.0 - p;
      }
  }

  for (int64_t tbase = 0; tbase < np; tbase += TP) {
    const int tp = (int)min((int64_t)TP, np - tbase);
    if (tid < TP) {
      const bool v = tid < tp;
      const uint32_t n = v ? load_nrd(pv.pair_nrd, p_beg + tbase + tid, nrd_width) : 0u;
      const uint32_t incl = seg_scan_incl<32>(n);
      s_cnt[tid] = n;
      s_off[tid] = rd_base + (int64_t)(incl - n);
      s_snp[tid] = v ? (pv.pair_snp ? pv.pair_snp[p_beg + tbase + tid] : (int32_t)(tbase + tid)) : 0;
    }
    DMX_K2_SYNC();
    rd_base = s_off[tp - 1] + (int64_t)s_cnt[tp - 1];
    {
      int r = tid % row_len, ti = tid / row_len;
      const int dr = TPC % row_len, dt = TPC / row_len;
      while (ti < tp) {
        s_g[ti * GS + r] = g[(size_t)s_snp[ti] * row_len + r];
        r += dr; ti += dt;
        if (r >= row_len) { r -= row_len; ++ti; }
      }
    }
    // ---- phase 1: lane u = (pair u / AP, alpha u % AP)
#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass) {
      const int u = tid + pass * TPC;
      if (u >= P1) break;                          // uniform per wavefront (P1 and TPC are multiples of 64)
      const int ti1 = u / AP;
      const bool on = ti1 < tp && n_ok;
      const uint32_t cnt = (ti1 < tp) ? s_cnt[ti1] : 0u;
      const int64_t off = (ti1 < tp) ? s_off[ti1] : 0;
      double pG[9];
#pragma unroll
      for (int i = 0; i < 9; ++i) pG[i] = 1.0;                               // :597
      for (uint32_t r = 0; __any(r < cnt); ++r) {
        const bool live = r < cnt && n_ok;
        const uint32_t byte = (r < cnt) ? pv.reads[off + r] : 0u;
        const uint32_t bq = byte & 127u;
        const bool alt = (byte >> 7) != 0;
        const double pR = alt ? s_tab[128 + bq] : s_tab[bq];                // :606
        const double pA = alt ? s_tab[bq] : s_tab[128 + bq];                // :607
        double mx = 0.0;
        if (live) {
#pragma unroll
          for (int i = 0; i < 9; ++i) {
            pG[i] *= (pR * wR[i] + pA * wA[i]);                             // :625
            mx = fmax(mx, pG[i]);                                 // :626-627
          }
        }
#pragma unroll
        for (int d = 1; d < AP; d <<= 1) {                                  // one max across ALL alphas of the pair
          const double o = __shfl_xor(mx, d);
          mx = fmax(mx, o);
        }
        if (live) {
          if (cnt <= kSafeReads) {
            const double y = rcp_refined(mx);
#pragma unroll
            for (int i = 0; i < 9; ++i) pG[i] = div_by(pG[i], mx, y);       // :632-639
          } else {
#pragma unroll
            for (int i = 0; i < 9; ++i) pG[i] /= mx;
          }
        }
      }
      double mx = 0.0;
      if (n_ok) {
#pragma unroll
        for (int i = 0; i < 9; ++i) {
          pG[i] += 1e-6;                                                     // :649
          mx = fmax(mx, pG[i]);
        }
      }
#pragma unroll
      for (int d = 1; d < AP; d <<= 1) {
        const double o = __shfl_xor(mx, d);
        mx = fmax(mx, o);
      }
      if (on) {
        const double y = rcp_refined(mx);
        const double* g0 = gp0 + (size_t)s_snp[ti1] * 3;
        const double qq[3] = {g0[0], g0[1], g0[2]};
        double sum = 0.0;
#pragma unroll
        for (int l = 0; l < 3; ++l)
#pragma unroll
          for (int m = 0; m < 3; ++m) {
            const double v = div_by(pG[l * 3 + m], mx, y);                   // :656-663
            s_pG[(ti1 * AP + n1) * 9 + l * 3 + m] = v;
            sum += ((qq[l] * qq[m]) * v);                                    // gp00 (:555) then :702-705
          }
        ok &= __builtin_amdgcn_class(sum, 0x100);
        s_t00[n1 * T00 + ti1] = dmx_log2_fast(sum, s_log);                    // :708-709 term
      }
    }
    DMX_K2_SYNC();
    if (tid < A) {                                 // llks00[n]: lane n adds its alpha's terms in pair order
      const double* row = &s_t00[tid * T00];
      for (int i = 0; i < tp; ++i) acc00 += row[i];
    }
    // ---- phase 2 in sub-tiles of SUBP pairs: u[n][l][k] = sum_m pG[n][l][m] g_k[m] once per (pair, n >= 1, k), then every printed
    //      entry is log(g_j . u_k) (bilinear form, fused multiply-adds)
#pragma unroll 1
    for (int sub = 0; sub < tp; sub += SUBP) {
      const int ns = min(SUBP, tp - sub);
      const int NI = (A - 1) * V + 1;              // items per pair: (n, k) for n >= 1, then the alpha-0 u of sample 0
#pragma unroll 1
      for (int e = tid; e < ns * NI; e += TPC) {
        const int pi = e / NI, it = e % NI;
        const bool sing = it == NI - 1;
        const int n = sing ? 0 : 1 + it / V, k = sing ? 0 : it % V;
        const double* P = &s_pG[((sub + pi) * AP + n) * 9];
        const float* gr = &s_g[(sub + pi) * GS + k * 3];
        const double b0 = (double)gr[0], b1 = (double)gr[1], b2 = (double)gr[2];
        double* u = sing ? &s_u[(size_t)pi * NU + (A - 1) * 3 * VU] : &s_u[(size_t)pi * NU + (size_t)(n - 1) * 3 * VU + k];
        const int us = sing ? 1 : VU;
#pragma unroll
        for (int l = 0; l < 3; ++l) u[l * us] = __builtin_fma(P[l * 3 + 2], b2, __builtin_fma(P[l * 3 + 1], b1, P[l * 3] * b0));
      }
      DMX_K2_SYNC();
      if (owner) {
        const int k0 = min(kb * NK, V - 1);
        const double* up = s_u + k0;               // this thread's first k; rows are VU apart, alphas 3 VU, pairs NU
#pragma unroll 1
        for (int pi = 0; pi < ns; ++pi, up += NU) {
          const float* gr = &s_g[(sub + pi) * GS];
          const double a0 = (double)gr[j * 3], a1 = (double)gr[j * 3 + 1], a2 = (double)gr[j * 3 + 2];
#pragma unroll
          for (int n = 1; n < AP; ++n) {
            if (n >= A) break;
#pragma unroll
            for (int kk = 0; kk < NK; ++kk) {
              // (k-blocks past the panel's end re-read its last columns: never stored)
              const int o = (n - 1) * 3 * VU + kk;
              const double x0 = lds_read_f64(&up[o]), x1 = lds_read_f64(&up[o + VU]), x2 = lds_read_f64(&up[o + 2 * VU]);
              const double sj = __builtin_fma(a2, x2, __builtin_fma(a1, x1, a0 * x0));
              if (CHK) ok &= __builtin_amdgcn_class(sj, 0x100) || kb * NK + kk >= V;
              acc[kk][n - 1] += dmx_log2_fastmode(sj, s_log, lk);
            }
            __builtin_amdgcn_sched_barrier(0);     // one alpha's NK evaluations in flight at a time (register budget)
          }
          if (kb == 0) {
            const double* u0 = s_u + (size_t)pi * NU + (A - 1) * 3 * VU;
            const double sj = __builtin_fma(a2, u0[2], __builtin_fma(a1, u0[1], a0 * u0[0]));
            ok &= __builtin_amdgcn_class(sj, 0x100);
            accS += dmx_log2_fastmode(sj, s_log, lk);
          }
        }
      }
      DMX_K2_SYNC();
    }
  }
  if (owner && kb == 0) s_sing[jl] = accS;
  DMX_K2_SYNC();
  if (cell_ok) {
    if (owner) {
      const double sj = s_sing[jl];                // llksAB[j][k != 0][0] is filled with llksAB[j][0][0]
#pragma unroll
      for (int kk = 0; kk < NK; ++kk) {
        const int k = kb * NK + kk;
        if (k < V) {
          double* o = grid + (((size_t)cell * V + j) * V + k) * A;
          o[0] = sj;
#pragma unroll
          for (int n = 1; n < AP; ++n) if (n < A) o[n] = acc[kk][n - 1];
        }
      }
    }
    if (tid < A && blockIdx.y == 0) l00[(size_t)cell * A + tid] = acc00;
    if (!ok) flag_cell(flagged, cell);
  }
#undef DMX_K2_SYNC
}

// ---------------------------------------------------------------------------------------------------------------------
// Genotype classes.  With --field GT every sample's probability row at a SNP is one of at most four float triplets
// (the three one-hot rows and the HWE row shared by all missing genotypes, bcf_filtered_reader.cpp:381-400), so
// log(sum_lm g_j[l] g_k[m] pG[l][m]) takes at most 16 distinct values per (pair, alpha) instead of V*V.  log and the
// arithmetic in front of it are pure functions of their operands: evaluating each distinct (class_j, class_k) once
// and adding the result into every accumulator that shares it is BIT-IDENTICAL to evaluating it V*V times, and keeps
// the per-accumulator addition order.  Rows are compared bitwise.  More than kMaxCls classes at any SNP (GP / PL
// inputs) => the class kernels are not used.
constexpr int kMaxCls = 4;

__global__ void k_build_classes(const float* __restrict__ g, int32_t S, int32_t V, float* __restrict__ rows /*[S][4][3]*/,
                                uint8_t* __restrict__ ids /*[S][V]*/, uint32_t* __restrict__ idw /*[S][ceil(V/16)] 2 bits per sample*/,
                                uint32_t* __restrict__ idd /*[S][nwd2]: word w = ids of samples (16 w + b) mod V, b = 0..15 (k_doublet_clsym)*/,
                                int32_t nwd2, int32_t* __restrict__ max_cls) {
  const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= S) return;
  const uint32_t* gr = reinterpret_cast<const uint32_t*>(g + (size_t)s * V * 3);
  uint32_t c[kMaxCls][3];
  int n = 0;
  bool over = false;
  for (int32_t k = 0; k < V; ++k) {
    const uint32_t a0 = gr[k * 3], a1 = gr[k * 3 + 1], a2 = gr[k * 3 + 2];
    int id = -1;
    for (int d = 0; d < n; ++d) if (c[d][0] == a0 && c[d][1] == a1 && c[d][2] == a2) { id = d; break; }
    if (id < 0) {
      if (n < kMaxCls) { id = n; c[n][0] = a0; c[n][1] = a1; c[n][2] = a2; ++n; }
      else { over = true; id = 0; }
    }
    ids[(size_t)s * V + k] = (uint8_t)id;
  }
  const int nwd = (V + 15) / 16;
  for (int wq = 0; wq < nwd; ++wq) {
    uint32_t wv = 0;
    for (int b = 0; b < 16 && wq * 16 + b < V; ++b) wv |= (uint32_t)ids[(size_t)s * V + wq * 16 + b] << (2 * b);
    idw[(size_t)s * nwd + wq] = wv;
  }
  for (int wq = 0; wq < nwd2; ++wq) {
    uint32_t wv = 0;
    int k = (16 * wq) % V;
    for (int b = 0; b < 16; ++b) { wv |= (uint32_t)ids[(size_t)s * V + k] << (2 * b); k = (k + 1 == V) ? 0 : k + 1; }
    idd[(size_t)s * nwd2 + wq] = wv;
  }
  uint32_t* ro = reinterpret_cast<uint32_t*>(rows + (size_t)s * kMaxCls * 3);
  for (int d = 0; d < kMaxCls; ++d) {
    const int e = d < n ? d : 0;                 // unused classes repeat class 0 (never referenced by an id)
    ro[d * 3] = c[e][0]; ro[d * 3 + 1] = c[e][1]; ro[d * 3 + 2] = c[e][2];
  }
  atomicMax(max_cls, over ? kMaxCls + 1 : n);
}

// ---- canonical GT classes (round 4) ----------------------------------------------------------------------------------------------
// --field GT gives every called genotype one of THREE rows that do not depend on the SNP: 1 - gt_error in the genotype's place, gt_error / 2
// in the other two (bcf_filtered_reader.cpp:397-400); only a missing genotype's Hardy-Weinberg row (:381-388) is the SNP's own.  When a
// matrix has that shape the classes are relabelled: id t < 3 = the row with hi in place t, id 3 = the SNP's other row (at most one; the
// slot repeats canonical row 0 where there is none, like k_build_classes' unused slots).  log(GL . row_t) of a pair then depends on the
// pair's reads only, and K1 reads it from a table (k_build_canon_logs) next to the GL tables.
__device__ __forceinline__ bool canon_pattern(uint32_t a, uint32_t b, uint32_t c, uint32_t* hi, uint32_t* lo) {
  const float fa = __uint_as_float(a), fb = __uint_as_float(b), fc = __uint_as_float(c);
  if (!(fa >= 0.f && fb >= 0.f && fc >= 0.f) || !(fa < 3e38f && fb < 3e38f && fc < 3e38f)) return false;
  if (b == c && fa > fb) { *hi = a; *lo = b; return true; }
  if (a == c && fb > fa) { *hi = b; *lo = a; return true; }
  if (a == b && fc > fa) { *hi = c; *lo = a; return true; }
  return false;
}
__device__ __forceinline__ int canon_type(const uint32_t* r, uint32_t hi, uint32_t lo) {
  if (r[0] == hi && r[1] == lo && r[2] == lo) return 0;
  if (r[0] == lo && r[1] == hi && r[2] == lo) return 1;
  if (r[0] == lo && r[1] == lo && r[2] == hi) return 2;
  return -1;
}
// the lowest SNP that has a row of the pattern (its first such class names hi and lo: deterministic)
__global__ void k_find_canon(const float* __restrict__ rows, int32_t S, int32_t* __restrict__ first) {
  const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= S) return;
  const uint32_t* r = reinterpret_cast<const uint32_t*>(rows + (size_t)s * kMaxCls * 3);
  uint32_t hi, lo;
  for (int d = 0; d < kMaxCls; ++d)
    if (canon_pattern(r[d * 3], r[d * 3 + 1], r[d * 3 + 2], &hi, &lo)) { atomicMin(first, (int32_t)s); return; }
}
// fail: an SNP with two different non-canonical rows among its classes
__global__ void k_canon_check(const float* __restrict__ rows, int32_t S, uint32_t hi, uint32_t lo, int32_t* __restrict__ fail) {
  const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= S) return;
  const uint32_t* r = reinterpret_cast<const uint32_t*>(rows + (size_t)s * kMaxCls * 3);
  int others = 0; uint32_t o[3] = {0, 0, 0};
  for (int d = 0; d < kMaxCls; ++d) {
    const uint32_t* q = r + d * 3;
    if (canon_type(q, hi, lo) >= 0) continue;
    if (others && o[0] == q[0] && o[1] == q[1] && o[2] == q[2]) continue;       // (an unused slot repeating class 0)
    if (others) { atomicOr(fail, 1); return; }
    o[0] = q[0]; o[1] = q[1]; o[2] = q[2]; others = 1;
  }
}
__global__ void k_canon_apply(float* __restrict__ rows, uint8_t* __restrict__ ids, uint32_t* __restrict__ idw, uint32_t* __restrict__ idd,
                              int32_t S, int32_t V, int32_t nwd2, uint32_t hi, uint32_t lo, uint8_t* __restrict__ oth, int32_t* __restrict__ any_oth) {
  const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= S) return;
  uint32_t* r = reinterpret_cast<uint32_t*>(rows + (size_t)s * kMaxCls * 3);
  uint32_t map[kMaxCls], o[3] = {hi, lo, lo};
  bool has_other = false;
  for (int d = 0; d < kMaxCls; ++d) {
    const int t = canon_type(r + d * 3, hi, lo);
    if (t >= 0) map[d] = (uint32_t)t;
    else { map[d] = 3u; o[0] = r[d * 3]; o[1] = r[d * 3 + 1]; o[2] = r[d * 3 + 2]; has_other = true; }
  }
  bool used3 = false;
  for (int32_t k = 0; k < V; ++k) {
    const uint32_t id = map[ids[(size_t)s * V + k] & 3u];
    used3 |= id == 3u;
    ids[(size_t)s * V + k] = (uint8_t)id;
  }
  const int nwd = (V + 15) / 16;
  for (int wq = 0; wq < nwd; ++wq) {
    uint32_t wv = 0;
    for (int b = 0; b < 16 && wq * 16 + b < V; ++b) wv |= (uint32_t)ids[(size_t)s * V + wq * 16 + b] << (2 * b);
    idw[(size_t)s * nwd + wq] = wv;
  }
  for (int wq = 0; wq < nwd2; ++wq) {
    uint32_t wv = 0;
    int k = (16 * wq) % V;
    for (int b = 0; b < 16; ++b) { wv |= (uint32_t)ids[(size_t)s * V + k] << (2 * b); k = (k + 1 == V) ? 0 : k + 1; }
    idd[(size_t)s * nwd2 + wq] = wv;
  }
  r[0] = hi; r[1] = lo; r[2] = lo;  r[3] = lo; r[4] = hi; r[5] = lo;  r[6] = lo; r[7] = lo; r[8] = hi;
  r[9] = o[0]; r[10] = o[1]; r[11] = o[2];
  oth[s] = (has_other && used3) ? 1 : 0;
  if (has_other && used3) atomicOr(any_oth, 1);
}
// ltab[i][t] = log(GL_i . canonical row t), i over the final GL tables: 257 one-read entries (256 = no read), 16 384 two-read, 96^3 three-read
__global__ void k_build_canon_logs(const double* __restrict__ tabs, double hi, double lo, double* __restrict__ ltab) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= kCanN) return;
  const double* G = i < kCanL2 ? tabs + kTab + kFirst + 3 * i
                  : (i < kCanL3 ? tabs + kTabK1 + kPair + 4 * (i - kCanL2) : tabs + kTabK1 + 2 * kPair + kTriple + 4 * (i - kCanL3));
  const double G0 = G[0], G1 = G[1], G2 = G[2];
  const double x0 = G0 * hi + G1 * lo + G2 * lo;      // the very expressions of k_singlet_cls on the canonical rows (:456)
  const double x1 = G0 * lo + G1 * hi + G2 * lo;
  const double x2 = G0 * lo + G1 * lo + G2 * hi;
  const double* T = tabs + kLut;
  ltab[4 * i] = dmx_log_fast(x0, T); ltab[4 * i + 1] = dmx_log_fast(x1, T); ltab[4 * i + 2] = dmx_log_fast(x2, T); ltab[4 * i + 3] = 0.0;
}

// K2 over genotype classes (A = 2).  Same ownership and order as k_doublet_a2; per tile of 32 pairs:
//   stage    headers, the pairs' class rows (4 x 3 float32) and per-sample class ids (V bytes) -> LDS
//   phase 1  pG[n][3][3] per (pair, alpha) and the llks00 term, exactly as k_doublet_a2
//   phase 1b the class table T[pair][cj][ck][n] = log(sum_lm row_cj[l] row_ck[m] pG[n][l][m]) — the very expression of
//            :553,:677-683 on the very operands, once per distinct (cj, ck)
//   phase 2  thread (j, k-block): acc[j][k][n] += T[pair][id_j][id_k][n], pairs in ascending SNP order
// UJ ("uniform j", panels of 33..64 samples): a wavefront owns a block of 16 samples j and its lanes are the samples k, so that the
// class of j is the same for the whole wavefront at every pair.  A lane reads its COLUMN T[0..3][ck] of the pair's table into sixteen
// consecutive registers and the row is chosen by VGPR-relative addressing (s_set_gpr_idx_on: the source register of the two v_add_f64
// is offset by 4 cj dwords) — no look-up, no address arithmetic and no branch per (j, k): two FP64 adds and two scalar instructions.
template <int TPC, int NK, int MINW = 1, bool UJ = false>
__global__ __launch_bounds__(kThreads, MINW) void k_doublet_cls(PileupView pv, int nrd_width, const float* __restrict__ rows,
                                                          const uint8_t* __restrict__ ids, const double* __restrict__ gp0,
                                                          const double* __restrict__ tabs, const double* __restrict__ alpha,
                                                          const int32_t* __restrict__ sched, int32_t V, int32_t VS,
                                                          double* __restrict__ grid, double* __restrict__ l00,
                                                          uint8_t* __restrict__ flagged) {
  constexpr int A = 2, TP = 32;
  constexpr int ablate = DMX_ABLATE;             // profiling builds only (tools/build_variant.sh); 0 in the product
  constexpr int CPW = kThreads / TPC;
  constexpr int T00 = TP + 2;
  constexpr int NT = kMaxCls * kMaxCls * A;      // class-table entries per pair
#define DMX_K2_SYNC() do { if (TPC == 64) { DMX_WAVE_LDS_ORDER(); } else { __syncthreads(); } } while (0)
  extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
  __shared__ double s_tab[kTab2];
  __shared__ double s_w[2][18];                  // mixing weights of :613 per alpha (see k_doublet_a2): registers only while phase 1 runs
  const double* s_log = s_tab + kLut2;
  const int t = threadIdx.x;
  stage_k2_tables(s_tab, tabs, t, kThreads);
  if (t < 18) {
    const int n = t / 9, l = (t % 9) / 3, m = t % 3;
    const double p = 0.5 * l + (m - l) * 0.5 * alpha[n];
    s_w[n][t % 9] = p;
    s_w[n][9 + t % 9] = 1.0 - p;
  }
  __syncthreads();

  const int cw = t / TPC, tid = t % TPC;
  // The V > 32 form (NK = 16: 208 registers, 2 wavefronts per SIMD either way) runs a two-deep pipeline: while tile t computes,
  // the header of tile t + 2 and the class rows, ids, leading read bytes and gp0 of tile t + 1 are in flight into registers.
  // (panels of up to 64 samples: a tile's id words fit two registers per thread)
  constexpr bool PF_T = NK >= 16 && TPC == 256;
  const bool PF = PF_T && VS <= 64;
  const size_t cell_bytes = (size_t)TP * 18 * 8 + (size_t)TP * NT * 8 + 2 * T00 * 8 + TP * (4 + 4 + 8) + (size_t)TP * 12 * 4 + (size_t)TP * VS +
                            (PF_T ? (size_t)TP * 16 : 0);
  unsigned char* base = s_raw + (size_t)cw * ((cell_bytes + 15) & ~(size_t)15);
  double* s_pG = (double*)base;                                  // [TP][2][9]
  double* s_T = s_pG + TP * 18;                                  // [TP][4][4][2]
  double* s_t00 = s_T + TP * NT;                                 // [2][T00]
  int64_t* s_off = (int64_t*)(s_t00 + 2 * T00);                  // [TP]
  int32_t* s_snp = (int32_t*)(s_off + TP);                       // [TP]
  uint32_t* s_cnt = (uint32_t*)(s_snp + TP);                     // [TP]
  float* s_rows = (float*)(s_cnt + TP);                          // [TP][4][3]
  uint8_t* s_ids = (uint8_t*)(s_rows + TP * 12);                 // [TP][VS]
  int64_t* s_off2 = (int64_t*)(s_ids + (size_t)TP * VS);         // PF: [TP] header of the NEXT tile (VS is a multiple of 4, TP * VS of 128)
  int32_t* s_snp2 = (int32_t*)(s_off2 + TP);
  uint32_t* s_cnt2 = (uint32_t*)(s_snp2 + TP);

  const int slot = blockIdx.x * CPW + cw;
  if (TPC == 64 && slot >= pv.B) return;
  const bool cell_ok = slot < pv.B;
  const int32_t cell = cell_ok ? sched[slot] : 0;
  const int64_t p_beg = cell_ok ? pv.cell_pair_off[cell] : 0;
  const int64_t np = cell_ok ? pv.cell_pair_off[cell + 1] - p_beg : 0;
  int64_t rd_base = cell_ok ? pv.cell_read_off[cell] : 0;

  static_assert(!UJ || (TPC == 256 && NK == 16 && MINW <= 2), "the uniform-j form: four wavefronts x 16 samples j, 64 lanes = samples k; registers v[232:247] are its column");
  const int KB = (V + NK - 1) / NK;
  const int JS = TPC / KB;                       // rows per workgroup; more rows => j-slabs over blockIdx.y (see k_doublet_a2)
  const int jl = tid / KB, kb = tid % KB;
  const int j = (int)blockIdx.y * JS + jl;
  const int uj_k = tid & 63, uj_jb = tid >> 6;   // UJ: lane = sample k, wavefront = samples j in [16 uj_jb, 16 uj_jb + 16)
  const bool owner = UJ ? (uj_k < V && uj_jb * 16 < V) : (jl < JS && j < V);
  double acc[NK][A];
#pragma unroll
  for (int kk = 0; kk < NK; ++kk) { acc[kk][0] = 0.0; acc[kk][1] = 0.0; }
  bool ok = true;
  const int ti1 = tid >> 1, n1 = tid & 1;
  double acc00 = 0.0;

  constexpr int NRR = (TP * 12 + TPC - 1) / TPC, NRI = (TP * 16 + TPC - 1) / TPC;   // per-thread registers of a tile's rows / id words
  const int wpr = VS / 4;                        // id words per pair
  uint32_t hd_n = 0u; int32_t hd_s = 0;          // PF: header loads in flight (lanes < TP of the first wavefront)
  uint32_t pn = 0u; int32_t psn = 0; int64_t poff = 0;
  float d_rows[NRR]; uint32_t d_ids[NRI]; uint32_t d_rd4 = 0u; double d_g0[3] = {0.0, 0.0, 0.0};
#pragma unroll
  for (int i = 0; i < NRR; ++i) d_rows[i] = 0.f;
#pragma unroll
  for (int i = 0; i < NRI; ++i) d_ids[i] = 0u;
  auto load_hdr = [&](int64_t first) {
    hd_n = 0u; hd_s = 0;
    const int64_t nx = first + tid;
    if (tid < TP && nx < np) { hd_n = load_nrd(pv.pair_nrd, p_beg + nx, nrd_width); hd_s = pv.pair_snp ? pv.pair_snp[p_beg + nx] : (int32_t)nx; }
  };
  auto publish_next = [&]() {                    // first wavefront: hd_* (arrived) -> prepared header of the next tile, into LDS
    if (tid < 64) {
      pn = hd_n; psn = hd_s;
      const uint32_t incl = seg_scan_incl<32>(pn);
      poff = rd_base + (int64_t)(incl - pn);
      rd_base += (int64_t)__shfl(incl, 31);
      if (tid < TP) { s_cnt2[tid] = pn; s_off2[tid] = poff; s_snp2[tid] = psn; }
    }
  };
  auto request_next = [&]() {                    // everybody: the published tile's rows / ids (and read bytes, gp0) -> registers
#pragma unroll
    for (int i = 0; i < NRR; ++i) {
      const int e = tid + TPC * i;
      if (e < TP * 12) d_rows[i] = rows[(size_t)s_snp2[e / 12] * 12 + (e % 12)];
    }
#pragma unroll
    for (int i = 0; i < NRI; ++i) {
      const int e = tid + TPC * i;
      if (e < TP * wpr) {
        const int ti = e / wpr, wq = e % wpr;
        const uint8_t* src = ids + (size_t)s_snp2[ti] * V + wq * 4;
        uint32_t wv = 0;
        for (int b = 0; b < 4; ++b) if (wq * 4 + b < V) wv |= (uint32_t)src[b] << (8 * b);
        d_ids[i] = wv;
      }
    }
    if (tid < 64) {
      d_rd4 = load_rd4(pv, s_off2[ti1], s_cnt2[ti1]);
      const double* g0 = gp0 + (size_t)s_snp2[ti1] * 3;
      d_g0[0] = g0[0]; d_g0[1] = g0[1]; d_g0[2] = g0[2];
    }
  };
  if (PF) {
    load_hdr(0);
    publish_next();
    load_hdr(TP);
    __syncthreads();
    request_next();
    __syncthreads();                             // the next tile's header may be overwritten from here on
  }
  for (int64_t tbase = 0; tbase < np; tbase += TP) {
    const int tp = (int)min((int64_t)TP, np - tbase);
    uint32_t rd4_cur = 0u; double g0_cur[3] = {0.0, 0.0, 0.0};
    if (PF) {
      if (tid < TP) { s_cnt[tid] = pn; s_off[tid] = poff; s_snp[tid] = psn; }
#pragma unroll
      for (int i = 0; i < NRR; ++i) { const int e = tid + TPC * i; if (e < TP * 12) s_rows[e] = d_rows[i]; }
#pragma unroll
      for (int i = 0; i < NRI; ++i) { const int e = tid + TPC * i; if (e < TP * wpr) reinterpret_cast<uint32_t*>(s_ids)[e] = d_ids[i] << 4; }   // (ids staged as byte offsets of 16-byte table cells)
      rd4_cur = d_rd4; g0_cur[0] = d_g0[0]; g0_cur[1] = d_g0[1]; g0_cur[2] = d_g0[2];
      publish_next();
      load_hdr(tbase + 2 * TP);
      DMX_K2_SYNC();
      request_next();
    } else {
    if (tid < TP) {
      const bool v = tid < tp;
      const uint32_t n = v ? load_nrd(pv.pair_nrd, p_beg + tbase + tid, nrd_width) : 0u;
      const uint32_t incl = seg_scan_incl<32>(n);
      s_cnt[tid] = n;
      s_off[tid] = rd_base + (int64_t)(incl - n);
      s_snp[tid] = v ? (pv.pair_snp ? pv.pair_snp[p_beg + tbase + tid] : (int32_t)(tbase + tid)) : 0;
    }
    DMX_K2_SYNC();
    rd_base = s_off[tp - 1] + (int64_t)s_cnt[tp - 1];
    // ---- class rows and ids -> LDS
    for (int e = tid; e < tp * 12; e += TPC) s_rows[e] = rows[(size_t)s_snp[e / 12] * 12 + (e % 12)];
    {
      for (int e = tid; e < tp * wpr; e += TPC) {
        const int ti = e / wpr, wq = e % wpr;
        const uint8_t* src = ids + (size_t)s_snp[ti] * V + wq * 4;
        uint32_t wv = 0;
        for (int b = 0; b < 4; ++b) if (wq * 4 + b < V) wv |= (uint32_t)src[b] << (8 * b);
        reinterpret_cast<uint32_t*>(s_ids)[ti * wpr + wq] = wv << 4;           // class id c (0..3) as c * 16: the byte offset of T[cj][c]
      }
    }
    }
    // ---- phase 1 (identical to k_doublet_a2)
    if (tid < 64 && !(ablate & 4096)) {
      const bool on = ti1 < tp;
      const uint32_t cnt = on ? s_cnt[ti1] : 0u;
      const int64_t off = on ? s_off[ti1] : 0;
      const uint32_t rd4 = PF ? rd4_cur : load_rd4(pv, off, cnt);   // the first four read bytes in one load (one dependent latency instead of four)
      double pG[9], wA[9], wR[9];
#pragma unroll
      for (int i = 0; i < 9; ++i) { pG[i] = 1.0; wA[i] = s_w[n1][i]; wR[i] = s_w[n1][9 + i]; }
      for (uint32_t r = 0; __any(r < cnt); ++r) {
        const bool live = r < cnt;
        const uint32_t byte = live ? (r < 4 ? (rd4 >> (8 * r)) & 0xFFu : (uint32_t)pv.reads[off + r]) : 0u;
        const uint32_t bq = byte & 127u;
        const bool alt = (byte >> 7) != 0;
        const double pR = alt ? s_tab[128 + bq] : s_tab[bq];
        const double pA = alt ? s_tab[bq] : s_tab[128 + bq];
        double mx = 0.0;
        if (live) {
#pragma unroll
          for (int i = 0; i < 9; ++i) {
            pG[i] *= (pR * wR[i] + pA * wA[i]);
            mx = fmax(mx, pG[i]);
          }
        }
        {
          const double o = shfl_xor1(mx);
          mx = fmax(mx, o);
        }
        if (live) {
          if (cnt <= kSafeReads) {
            const double y = rcp_refined(mx);
#pragma unroll
            for (int i = 0; i < 9; ++i) pG[i] = div_by(pG[i], mx, y);
          } else {
#pragma unroll
            for (int i = 0; i < 9; ++i) pG[i] /= mx;
          }
        }
      }
      double mx = 0.0;
#pragma unroll
      for (int i = 0; i < 9; ++i) {
        pG[i] += 1e-6;
        mx = fmax(mx, pG[i]);
      }
      {
        const double o = shfl_xor1(mx);
        mx = fmax(mx, o);
      }
      if (on) {
        const double y = rcp_refined(mx);
        const double* g0 = gp0 + (size_t)s_snp[ti1] * 3;
        const double qq[3] = {PF ? g0_cur[0] : g0[0], PF ? g0_cur[1] : g0[1], PF ? g0_cur[2] : g0[2]};
        double sum = 0.0;
#pragma unroll
        for (int l = 0; l < 3; ++l)
#pragma unroll
          for (int m = 0; m < 3; ++m) {
            const double v = div_by(pG[l * 3 + m], mx, y);
            s_pG[(ti1 * 2 + n1) * 9 + l * 3 + m] = v;
            sum += ((qq[l] * qq[m]) * v);
          }
        ok &= __builtin_amdgcn_class(sum, 0x100);
        s_t00[n1 * T00 + ti1] = dmx_log2_fast(sum, s_log);
      }
    }
    DMX_K2_SYNC();
    if (tid < 2) {
      const double* row = &s_t00[tid * T00];
      if (tp == TP) {
        double2 v[TP / 2];
#pragma unroll
        for (int i = 0; i < TP / 2; ++i) v[i] = *reinterpret_cast<const double2*>(&row[2 * i]);
#pragma unroll
        for (int i = 0; i < TP / 2; ++i) { acc00 += v[i].x; acc00 += v[i].y; }
      } else {
        for (int i = 0; i < tp; ++i) acc00 += row[i];
      }
    }
    // ---- phase 1b: the class table
    for (int e = tid; e < ((ablate & 2048) ? 0 : tp * NT); e += TPC) {
      const int ti = e / NT, cc = e % NT;
      const int cj = cc >> 3, ck = (cc >> 1) & 3, n = cc & 1;
      const float* rj = &s_rows[ti * 12 + cj * 3];
      const float* rk = &s_rows[ti * 12 + ck * 3];
      const double* P = &s_pG[(ti * 2 + n) * 9];
      const double aj[3] = {(double)rj[0], (double)rj[1], (double)rj[2]};
      const double bk[3] = {(double)rk[0], (double)rk[1], (double)rk[2]};
      double sum = 0.0;                                                          // :674
#pragma unroll
      for (int l = 0; l < 3; ++l)
#pragma unroll
        for (int m = 0; m < 3; ++m) sum += ((aj[l] * bk[m]) * P[l * 3 + m]);     // :553, :677-679
      ok &= __builtin_amdgcn_class(sum, 0x100);
      s_T[ti * NT + cc] = dmx_log2_fast(sum, s_log);                              // the :683 term
    }
    DMX_K2_SYNC();
    // ---- phase 2: one lookup and two adds per (j, k)
    if (UJ) {
      if (owner && !(ablate & 1024)) {
        typedef uint32_t dmx_u4 __attribute__((ext_vector_type(4)));
        using lds_u4 = const __attribute__((address_space(3))) dmx_u4*;
        const uint32_t t_a = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) double*)s_T;
        const uint32_t id_a = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) uint8_t*)s_ids;
        for (int ti = 0; ti < tp; ++ti) {
          // the 16 classes of the wavefront's samples j (bytes c * 16, one broadcast read) into scalar registers
          const dmx_u4 jw = *(lds_u4)(uintptr_t)(id_a + (uint32_t)(ti * VS + uj_jb * 16));
          const uint32_t w0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)jw.x), w1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)jw.y),
                         w2 = (uint32_t)__builtin_amdgcn_readfirstlane((int)jw.z), w3 = (uint32_t)__builtin_amdgcn_readfirstlane((int)jw.w);
          // this lane's column of the pair's table: T[cj][ck] for cj = 0..3 (rows 64 bytes apart), ck = the class of sample k
          const uint32_t col = t_a + (uint32_t)(ti * NT * 8) + (uint32_t)s_ids[ti * VS + uj_k];
          // One asm statement per pair: the column into v[232:247] (T[c] = v[232 + 4c : 235 + 4c]: alpha 0, alpha 1), then per j
          // "index = 4 cj" (s_bfe_u32 takes bits 2..5 of the staged byte c * 16) and the two adds with SRC0 relative to it.
#define DMX_UJ_STEP(A0, A1, W, B)                                                                              \
          "s_bfe_u32 %[t], %[" W "], " B "\n\ts_set_gpr_idx_on %[t], 0x1\n\t"                                  \
          "v_add_f64 %[" A0 "], v[232:233], %[" A0 "]\n\tv_add_f64 %[" A1 "], v[234:235], %[" A1 "]\n\t"
          {
            uint32_t tmp;
            uint32_t m0_keep;                         // s_set_gpr_idx_on writes M0: saved and restored inside the statement (the
                                                      // compiler may keep an LDS-DMA base or a movrel index live in it)
            asm volatile(
                "s_mov_b32 %[m0k], m0\n\t"
                "ds_read_b128 v[232:235], %[col]\n\tds_read_b128 v[236:239], %[col] offset:64\n\t"
                "ds_read_b128 v[240:243], %[col] offset:128\n\tds_read_b128 v[244:247], %[col] offset:192\n\t"
                "s_waitcnt lgkmcnt(0)\n\t"
                DMX_UJ_STEP("a0", "b0", "w0", "0x40002") DMX_UJ_STEP("a1", "b1", "w0", "0x4000a")
                DMX_UJ_STEP("a2", "b2", "w0", "0x40012") DMX_UJ_STEP("a3", "b3", "w0", "0x4001a")
                DMX_UJ_STEP("a4", "b4", "w1", "0x40002") DMX_UJ_STEP("a5", "b5", "w1", "0x4000a")
                DMX_UJ_STEP("a6", "b6", "w1", "0x40012") DMX_UJ_STEP("a7", "b7", "w1", "0x4001a")
                DMX_UJ_STEP("a8", "b8", "w2", "0x40002") DMX_UJ_STEP("a9", "b9", "w2", "0x4000a")
                DMX_UJ_STEP("a10", "b10", "w2", "0x40012") DMX_UJ_STEP("a11", "b11", "w2", "0x4001a")
                DMX_UJ_STEP("a12", "b12", "w3", "0x40002") DMX_UJ_STEP("a13", "b13", "w3", "0x4000a")
                DMX_UJ_STEP("a14", "b14", "w3", "0x40012") DMX_UJ_STEP("a15", "b15", "w3", "0x4001a")
                "s_set_gpr_idx_off\n\ts_mov_b32 m0, %[m0k]"
                : [a0] "+v"(acc[0][0]), [b0] "+v"(acc[0][1]), [a1] "+v"(acc[1][0]), [b1] "+v"(acc[1][1]),
                  [a2] "+v"(acc[2][0]), [b2] "+v"(acc[2][1]), [a3] "+v"(acc[3][0]), [b3] "+v"(acc[3][1]),
                  [a4] "+v"(acc[4][0]), [b4] "+v"(acc[4][1]), [a5] "+v"(acc[5][0]), [b5] "+v"(acc[5][1]),
                  [a6] "+v"(acc[6][0]), [b6] "+v"(acc[6][1]), [a7] "+v"(acc[7][0]), [b7] "+v"(acc[7][1]),
                  [a8] "+v"(acc[8][0]), [b8] "+v"(acc[8][1]), [a9] "+v"(acc[9][0]), [b9] "+v"(acc[9][1]),
                  [a10] "+v"(acc[10][0]), [b10] "+v"(acc[10][1]), [a11] "+v"(acc[11][0]), [b11] "+v"(acc[11][1]),
                  [a12] "+v"(acc[12][0]), [b12] "+v"(acc[12][1]), [a13] "+v"(acc[13][0]), [b13] "+v"(acc[13][1]),
                  [a14] "+v"(acc[14][0]), [b14] "+v"(acc[14][1]), [a15] "+v"(acc[15][0]), [b15] "+v"(acc[15][1]),
                  [t] "=&s"(tmp), [m0k] "=&s"(m0_keep)
                : [col] "v"(col), [w0] "s"(w0), [w1] "s"(w1), [w2] "s"(w2), [w3] "s"(w3)
                : "v232", "v233", "v234", "v235", "v236", "v237", "v238", "v239", "v240", "v241", "v242", "v243", "v244", "v245",
                  "v246", "v247", "scc", "memory");
          }
#undef DMX_UJ_STEP
        }
      }
    } else
    if (owner) {
      for (int ti = 0; ti < tp; ++ti) {
        const uint8_t* idr = &s_ids[ti * VS];
        const int cj = idr[j] >> 4;
        const double* Tj = &s_T[ti * NT + cj * 8];
        typedef double dmx_d2v __attribute__((ext_vector_type(2)));
        using lds_c2 = const __attribute__((address_space(3))) dmx_d2v*;
        const uint32_t tj_a = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) double*)Tj;   // LDS byte address of T[cj][0]
        if (NK >= 4) {                            // class ids of the k-block, four per 32-bit LDS read (VS is padded to 4)
#pragma unroll
          for (int kq = 0; kq < NK / 4; ++kq) {
            const uint32_t w4 = reinterpret_cast<const uint32_t*>(idr)[(kb * NK) / 4 + kq];
            // address = row + byte b of the id word: one v_add_u32 with an SDWA byte select (the ids were staged as c * 16)
            const dmx_d2v t0 = *(lds_c2)(uintptr_t)add_byte_of<0>(tj_a, w4), t1 = *(lds_c2)(uintptr_t)add_byte_of<1>(tj_a, w4),
                          t2 = *(lds_c2)(uintptr_t)add_byte_of<2>(tj_a, w4), t3 = *(lds_c2)(uintptr_t)add_byte_of<3>(tj_a, w4);
            acc[kq * 4 + 0][0] += t0.x; acc[kq * 4 + 0][1] += t0.y;             // :683, alpha 0 / alpha 1
            acc[kq * 4 + 1][0] += t1.x; acc[kq * 4 + 1][1] += t1.y;
            acc[kq * 4 + 2][0] += t2.x; acc[kq * 4 + 2][1] += t2.y;
            acc[kq * 4 + 3][0] += t3.x; acc[kq * 4 + 3][1] += t3.y;
          }
        } else {
#pragma unroll
          for (int kk = 0; kk < NK; ++kk) {
            const int k = min(kb * NK + kk, V - 1);
            const dmx_d2v tv = *(lds_c2)(uintptr_t)(tj_a + idr[k]);
            acc[kk][0] += tv.x;
            acc[kk][1] += tv.y;
          }
        }
      }
    }
    DMX_K2_SYNC();
  }
  if (cell_ok) {
    if (UJ) {
      if (owner) {
#pragma unroll
        for (int jj = 0; jj < 16; ++jj) {
          const int jx = uj_jb * 16 + jj;
          if (jx < V) {
            double* o = grid + (((size_t)cell * V + jx) * V + uj_k) * A;
            o[0] = acc[jj][0]; o[1] = acc[jj][1];
          }
        }
      }
    } else
    if (owner) {
#pragma unroll
      for (int kk = 0; kk < NK; ++kk) {
        const int k = kb * NK + kk;
        if (k < V) {
          double* o = grid + (((size_t)cell * V + j) * V + k) * A;
          o[0] = acc[kk][0]; o[1] = acc[kk][1];
        }
      }
    }
    if (tid < 2 && blockIdx.y == 0) l00[(size_t)cell * A + tid] = acc00;
    if (!ok) flag_cell(flagged, cell);
  }
#undef DMX_K2_SYNC
}

// K2 over genotype classes, STRICT, A = 2, panels of 33..64 samples: PRODUCER / CONSUMER form of k_doublet_cls's uniform-j kernel (round 4).
// In k_doublet_cls a tile has four workgroup barriers and phase 1 runs on one of the barcode's four wavefronts while the other three
// wait (43 % of the wave time parked, rocprofv3 r03).  Here the four wavefronts of a barcode's workgroup have two roles:
//   wavefront 0 (producer)   tile t + 1: headers, class rows and ids (requested a tile ahead, in registers) -> LDS, phase 1 (pG, the
//                            llks00 terms and their ordered sum), phase 1b (the class table) into one half of a double buffer {T, ids};
//   wavefronts 1..3 (consumers)  tile t: phase 2 of the uniform-j form for 24 / 20 / 20 samples j — lanes = samples k, the pair's table
//                            column in sixteen consecutive registers, the row chosen by VGPR-relative addressing (see k_doublet_cls);
// ONE s_barrier per tile.  Phase 2 of a tile is ONE software-pipelined asm statement per wavefront (DMX_PJ_* below): while a pair's
// additions issue out of one set of column registers, the next pair's column is read into a second set and the class words of the pair
// after it into v[128:133] — no LDS round trip on the additions' path (the unpipelined loop of the round's first version had two per
// pair, about as long as the additions themselves: 741 ms at the cfg4 shard, this form 486 ms).  A wavefront's class words are simply
// its bytes of the pair's id row (class 0 beyond V): six or five aligned words.  Production (about 15 k cycles of one wavefront per tile,
// mostly latency) is as long as a consumer's 32 x 24 samples; with the last four samples j also on the producer (4 / 20 / 20 / 20, the
// first version) it was the critical path.  The consumers never touch global memory inside the loop: 168 registers, 3 wavefronts per SIMD
// (k_doublet_cls<256,16,uniform-j>: 256 registers, 2 per SIMD), so a SIMD has two other barcodes to issue for while one waits.
// Same operands, same operations, same order of additions as k_doublet_cls: bit-identical (tests).  cmd_cram_demuxlet.cpp:594-710.
constexpr int kPcJ1 = 24, kPcJ = 20;           // samples j of wavefront 1 and of wavefronts 2, 3
static_assert(kPcJ1 + 2 * kPcJ == 64 && kPcJ1 % 4 == 0 && kPcJ % 4 == 0, "k_doublet_clsp: a wavefront's class bytes are whole words of the id row");
// FAST (DMX_MODE_FAST, alpha grid {0, 0.5}; round 4): the same kernel over the entries demuxlet prints or decides on.  The class table holds
// alpha 0.5 only, ONE evaluation per unordered class pair (cj <= ck, mirrored: T is bitwise symmetric, so the accumulators of (j, k) and (k, j)
// add identical terms and the grid comes out mirrored without an exchange), plus the four alpha-0 values T[cj][class of sample 0] of the
// singlet column.  Phase 2 adds one value per (pair, j, k) instead of two (columns of 8 registers, index 16 + 2 c); the producer wavefront
// (lanes = samples j) accumulates the singlet column [j][0][0], which the never-printed [j][k][0] repeat (as in k_doublet_sym / _clsym).
// The values are STRICT's own (reference operation order) for the orientation evaluated.
template <int MINW, bool FAST = false>
__global__ __launch_bounds__(kThreads, MINW) void k_doublet_clsp(PileupView pv, int nrd_width, const float* __restrict__ rows,
                                                                const uint8_t* __restrict__ ids, const double* __restrict__ gp0,
                                                                const double* __restrict__ tabs, const double* __restrict__ alpha,
                                                                const int32_t* __restrict__ sched, int32_t V, int32_t VS,
                                                                double* __restrict__ grid, double* __restrict__ l00,
                                                                uint8_t* __restrict__ flagged) {
  constexpr int A = 2, TP = 32;
  constexpr int T00 = TP + 2;
  constexpr int NT = kMaxCls * kMaxCls * A;      // class-table entries per pair
  constexpr int VSC = 64;                        // id row stride in the LDS (bytes): compile-time for panels of up to 64 samples
  extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
  __shared__ double s_tab[kTab2];
  __shared__ double s_w[2][18];                  // mixing weights of :613 per alpha
  const double* s_log = s_tab + kLut2;
  const int t = threadIdx.x;
  stage_k2_tables(s_tab, tabs, t, kThreads);
  if (t < 18) {
    const int n = t / 9, l = (t % 9) / 3, m = t % 3;
    const double p = 0.5 * l + (m - l) * 0.5 * alpha[n];
    s_w[n][t % 9] = p;
    s_w[n][9 + t % 9] = 1.0 - p;
  }
  // double-buffered (what phase 2 reads): the class table and the ids (as c * 16);
  // producer-private: pG, the llks00 terms, headers (current and next), rows
  double* s_Tb = (double*)s_raw;                                  // [2][TP][4][4][2]
  uint8_t* s_idb = (uint8_t*)(s_Tb + 2 * TP * NT);                // [2][TP][VSC]
  double* s_pG = (double*)(s_idb + 2 * (size_t)TP * VSC);         // [TP][2][9]
  double* s_t00 = s_pG + TP * 18;                                 // [2][T00]
  int64_t* s_off = (int64_t*)(s_t00 + 2 * T00);                   // [TP]
  int64_t* s_off2 = s_off + TP;                                   // [TP]  header of the NEXT tile
  int32_t* s_snp = (int32_t*)(s_off2 + TP);                       // [TP]
  int32_t* s_snp2 = s_snp + TP;
  uint32_t* s_cnt = (uint32_t*)(s_snp2 + TP);                     // [TP]
  uint32_t* s_cnt2 = s_cnt + TP;
  float* s_rows = (float*)(s_cnt2 + TP);                          // [TP][4][3]
  __syncthreads();

  const int32_t cell = sched[blockIdx.x];
  const int64_t p_beg = pv.cell_pair_off[cell];
  const int64_t np = pv.cell_pair_off[cell + 1] - p_beg;
  constexpr int wpr = VSC / 4;                   // id words per pair
  (void)VS;
  const int wave = t >> 6, lane = t & 63;        // lane = sample k in phase 2
  const int j0 = wave <= 1 ? 0 : kPcJ1 + (wave - 2) * kPcJ;      // first sample j of this (consumer) wavefront
  using lds_u8 = const __attribute__((address_space(3))) uint8_t*;

  // Phase 2 of one tile for a wavefront's samples j, one asm statement.  Per (pair, sample j): the class of j (wave-uniform) picks the row
  // of the lane's column by VGPR-relative addressing (see k_doublet_cls), then one addition per alpha.  Column sets X = v[152:167] and
  // Y = v[136:151] (the top of the 168-register budget), class words of the pair being requested in v[128:133], its id byte in v135;
  // two sets of class words in scalar registers (wa*, wb*).  Invariant at the top of a pair: its column requested, then the next pair's
  // words and id byte (NW LDS operations younger): s_waitcnt lgkmcnt(NW) is enough (LDS operations return in order).  The pairs beyond
  // the tile that the last rounds request read LDS inside the workgroup's allocation (the id byte is masked to a class's two bits, so the
  // column reads stay 16-byte aligned) and are drained before the statement ends.  DS instructions of gfx9+ do not use M0, so the
  // indexing state left in it is harmless; M0 is restored at the end.
  // The id bytes are 0x10 | c << 2 (FAST: 0x10 | c << 1): a word of four, shifted right by whole bytes, IS the M0 of the indexing mode —
  // M0[7:0] = 16 + 4 c (16 + 2 c; the column operands name the registers 16 below the set), M0[15:12] = the next byte's 0x1 = "index
  // SRC0".  One scalar instruction per sample j for three of a word's four (the hardware interlocks the M0 write against the indexed read).
  // Macro arguments: M = ST (STRICT: two additions per sample j, 16-register columns) | FA (FAST: one, 8-register columns); SET = SX | SY.
#define DMX_PJ_C0_SX "v[136:137]"
#define DMX_PJ_C1_SX "v[138:139]"
#define DMX_PJ_C0_SY "v[120:121]"
#define DMX_PJ_C1_SY "v[122:123]"
#define DMX_PJ_ADDS_ST(SET, A0, A1) "v_add_f64 %[" A0 "], " DMX_PJ_C0_##SET ", %[" A0 "]\n\tv_add_f64 %[" A1 "], " DMX_PJ_C1_##SET ", %[" A1 "]\n\t"
#define DMX_PJ_ADDS_FA(SET, A0, A1) "v_add_f64 %[" A1 "], " DMX_PJ_C0_##SET ", %[" A1 "]\n\t"
#define DMX_PJ_WORD4(M, SET, W, A0, B0, A1, B1, A2, B2, A3, B3)                                                     \
  "s_set_gpr_idx_on %[" W "], 0x1\n\t" DMX_PJ_ADDS_##M(SET, A0, B0)                                                \
  "s_lshr_b32 m0, %[" W "], 8\n\t" DMX_PJ_ADDS_##M(SET, A1, B1)                                                    \
  "s_lshr_b32 m0, %[" W "], 16\n\t" DMX_PJ_ADDS_##M(SET, A2, B2)                                                   \
  "s_lshr_b32 %[t], %[" W "], 24\n\ts_set_gpr_idx_on %[t], 0x1\n\t" DMX_PJ_ADDS_##M(SET, A3, B3)
#define DMX_PJ_FIRST4(M, SET, P) DMX_PJ_WORD4(M, SET, P "0", "a0", "b0", "a1", "b1", "a2", "b2", "a3", "b3")
#define DMX_PJ_REST16(M, SET, P)                                                                                  \
  DMX_PJ_WORD4(M, SET, P "1", "a4", "b4", "a5", "b5", "a6", "b6", "a7", "b7")                                       \
  DMX_PJ_WORD4(M, SET, P "2", "a8", "b8", "a9", "b9", "a10", "b10", "a11", "b11")                                   \
  DMX_PJ_WORD4(M, SET, P "3", "a12", "b12", "a13", "b13", "a14", "b14", "a15", "b15")                               \
  DMX_PJ_WORD4(M, SET, P "4", "a16", "b16", "a17", "b17", "a18", "b18", "a19", "b19")
#define DMX_PJ_REST20(M, SET, P) DMX_PJ_REST16(M, SET, P)                                                          \
  DMX_PJ_WORD4(M, SET, P "5", "a20", "b20", "a21", "b21", "a22", "b22", "a23", "b23")
  // the lane's column of the pair's table: T[cj][ck][n] for cj = 0..3 (rows 64 bytes apart), ck = the class of sample k = lane
#define DMX_PJ_COLADDR_ST "v_and_b32 %[colv], 12, v135\n\tv_lshl_add_u32 %[colv], %[colv], 2, %[pt]\n\t"
#define DMX_PJ_COLADDR_FA "v_and_b32 %[colv], 6, v135\n\tv_lshl_add_u32 %[colv], %[colv], 3, %[pt]\n\t"
#define DMX_PJ_READS_ST_SX "ds_read_b128 v[152:155], %[colv]\n\tds_read_b128 v[156:159], %[colv] offset:64\n\t"                                     \
                           "ds_read_b128 v[160:163], %[colv] offset:128\n\tds_read_b128 v[164:167], %[colv] offset:192\n\t"
#define DMX_PJ_READS_ST_SY "ds_read_b128 v[136:139], %[colv]\n\tds_read_b128 v[140:143], %[colv] offset:64\n\t"                                     \
                           "ds_read_b128 v[144:147], %[colv] offset:128\n\tds_read_b128 v[148:151], %[colv] offset:192\n\t"
  // (FAST: the alpha-0.5 halves only; offsets in 8-byte units)
#define DMX_PJ_READS_FA_SX "ds_read2_b64 v[152:155], %[colv] offset0:1 offset1:9\n\tds_read2_b64 v[156:159], %[colv] offset0:17 offset1:25\n\t"
#define DMX_PJ_READS_FA_SY "ds_read2_b64 v[136:139], %[colv] offset0:1 offset1:9\n\tds_read2_b64 v[140:143], %[colv] offset0:17 offset1:25\n\t"
  // the next pair: wait for its words / id byte, request its column into the set and move its words into a scalar set (RFL); then
  // request the words of the pair after it (WORDS).  pj: the wavefront's words in the id row, pi: the lane's id byte, pt: the table.
#define DMX_PJ_MID(M, SET, RFL, WORDS)                                                                            \
  "s_set_gpr_idx_off\n\ts_waitcnt lgkmcnt(0)\n\t" DMX_PJ_COLADDR_##M DMX_PJ_READS_##M##_##SET RFL                   \
  "v_add_u32 %[pj], 64, %[pj]\n\tv_add_u32 %[pi], 64, %[pi]\n\tv_add_u32 %[pt], 0x100, %[pt]\n\t"                   \
  WORDS
#define DMX_PJ_WORDS5 "ds_read2_b32 v[128:129], %[pj] offset1:1\n\tds_read2_b32 v[130:131], %[pj] offset0:2 offset1:3\n\t" \
                      "ds_read_b32 v132, %[pj] offset:16\n\tds_read_u8 v135, %[pi]\n\t"
#define DMX_PJ_WORDS6 "ds_read2_b32 v[128:129], %[pj] offset1:1\n\tds_read2_b32 v[130:131], %[pj] offset0:2 offset1:3\n\t" \
                      "ds_read2_b32 v[132:133], %[pj] offset0:4 offset1:5\n\tds_read_u8 v135, %[pi]\n\t"
#define DMX_PJ_RFL5(Q)                                                                                            \
  "v_readfirstlane_b32 %[" Q "0], v128\n\tv_readfirstlane_b32 %[" Q "1], v129\n\tv_readfirstlane_b32 %[" Q "2], v130\n\t" \
  "v_readfirstlane_b32 %[" Q "3], v131\n\tv_readfirstlane_b32 %[" Q "4], v132\n\t"
#define DMX_PJ_RFL6(Q) DMX_PJ_RFL5(Q) "v_readfirstlane_b32 %[" Q "5], v133\n\t"
  // a consumer's tile: WORDS / RFL(set) / REST for its 20 or 24 samples j.  At the top of a pair its column is requested and then the
  // next pair's words and id byte, four LDS operations younger: s_waitcnt lgkmcnt(4) waits for exactly the column.
#define DMX_PJ_CONSUMER(M, WORDS, RFL, REST)                                                                       \
  "s_mov_b32 %[m0k], m0\n\t" WORDS DMX_PJ_MID(M, SX, RFL("wa"), WORDS) "s_mov_b32 %[ti], 0\n"                      \
  "L_pjc_%=:\n\t"                                                                                                 \
  "s_waitcnt lgkmcnt(4)\n\t" DMX_PJ_FIRST4(M, SX, "wa") DMX_PJ_MID(M, SY, RFL("wb"), WORDS)                        \
  REST(M, SX, "wa") DMX_PJ_NEXT("L_pjd_%=")                                                                       \
  "s_waitcnt lgkmcnt(4)\n\t" DMX_PJ_FIRST4(M, SY, "wb") DMX_PJ_MID(M, SX, RFL("wa"), WORDS)                        \
  REST(M, SY, "wb")                                                                                               \
  "s_add_u32 %[ti], %[ti], 1\n\ts_cmp_lt_u32 %[ti], %[tp]\n\ts_cbranch_scc1 L_pjc_%=\n"                           \
  "L_pjd_%=:\n\t" DMX_PJ_END
#define DMX_PJ_NEXT(LBL) "s_add_u32 %[ti], %[ti], 1\n\ts_cmp_ge_u32 %[ti], %[tp]\n\ts_cbranch_scc1 " LBL "\n\t"
#define DMX_PJ_END "s_set_gpr_idx_off\n\ts_waitcnt lgkmcnt(0)\n\ts_mov_b32 m0, %[m0k]"
#define DMX_PJ_CLOB "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143", "v144", \
                    "v145", "v146", "v147", "v148", "v149", "v150", "v151", "v152", "v153", "v154", "v155", "v156", "v157", "v158", "v159", \
                    "v160", "v161", "v162", "v163", "v164", "v165", "v166", "v167", "scc", "memory"

  // FAST, phase 1b: the class table of tile [tbase, ..) into buffer b2, by ALL FOUR wavefronts between two barriers of their own (14 of 16
  // slots per pair, two per thread): in FAST the producer is the longer role (phase 2 is half of STRICT's), and its 1b was half of its time.
  auto phase1b_fast = [&](int64_t tbase, int b2, bool& ok) {
    const int tp = (int)min((int64_t)TP, np - tbase);
    double* s_T = s_Tb + (size_t)b2 * TP * NT;
    const uint8_t* s_ids = s_idb + (size_t)b2 * TP * VSC;
    const int q = t & 15;
    // q < 10: the unordered class pairs (cj <= ck) at alpha 0.5; q = 10..13: (cj = q - 10, the class of sample 0) at alpha 0
    const int cj = (0xE4E9500u >> (2 * q)) & 3, ckq = (0xFB9E4u >> (2 * q)) & 3, n = q < 10 ? 1 : 0;
    if (q < 14)
      for (int ti = t >> 4; ti < tp; ti += kThreads / 16) {
        const int ck = q < 10 ? ckq : (s_ids[ti * VSC] >> 1) & 3;
        const float* rj = &s_rows[ti * 12 + cj * 3];
        const float* rk = &s_rows[ti * 12 + ck * 3];
        const double* P = &s_pG[(ti * 2 + n) * 9];
        const double aj[3] = {(double)rj[0], (double)rj[1], (double)rj[2]};
        const double bk[3] = {(double)rk[0], (double)rk[1], (double)rk[2]};
        double sum = 0.0;                                                          // :674
#pragma unroll
        for (int l = 0; l < 3; ++l)
#pragma unroll
          for (int m = 0; m < 3; ++m) sum += ((aj[l] * bk[m]) * P[l * 3 + m]);     // :553, :677-679
        ok &= __builtin_amdgcn_class(sum, 0x100);
        const double val = dmx_log2_fast(sum, s_log);                              // the :683 term
        s_T[ti * NT + (cj * 4 + ck) * 2 + n] = val;
        if (q < 10) s_T[ti * NT + (ck * 4 + cj) * 2 + 1] = val;
      }
  };

  if (wave == 0) {
    // ================================================= producer =================================================
    const int ti1 = lane >> 1, n1 = lane & 1;
    int64_t rd_base = pv.cell_read_off[cell];
    double acc00 = 0.0;
    [[maybe_unused]] double acc0 = 0.0;          // FAST: the singlet column entry of sample j = lane
    bool ok = true;
    constexpr int NRR = 6, NRI = 8;              // per-lane registers of a tile's rows / id words: half a pair's 12 floats / 16 id words
    uint32_t hd_n = 0u; int32_t hd_s = 0;        // header loads in flight (lanes < TP)
    uint32_t pn = 0u; int32_t psn = 0; int64_t poff = 0;
    float d_rows[NRR]; uint32_t d_ids[NRI]; uint32_t d_rd4 = 0u; double d_g0[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (int i = 0; i < NRR; ++i) d_rows[i] = 0.f;
#pragma unroll
    for (int i = 0; i < NRI; ++i) d_ids[i] = 0u;
    auto load_hdr = [&](int64_t first) {
      hd_n = 0u; hd_s = 0;
      const int64_t nx = first + lane;
      if (lane < TP && nx < np) { hd_n = load_nrd(pv.pair_nrd, p_beg + nx, nrd_width); hd_s = pv.pair_snp ? pv.pair_snp[p_beg + nx] : (int32_t)nx; }
    };
    auto publish_next = [&]() {                  // hd_* (arrived) -> the prepared header of the next tile, into LDS
      pn = hd_n; psn = hd_s;
      const uint32_t incl = seg_scan_incl<32>(pn);
      poff = rd_base + (int64_t)(incl - pn);
      rd_base += (int64_t)__shfl(incl, 31);
      if (lane < TP) { s_cnt2[lane] = pn; s_off2[lane] = poff; s_snp2[lane] = psn; }
    };
    auto request_next = [&]() {                  // the published tile's rows, ids, leading read bytes and gp0 -> registers
      // lane (pair ti1, half n1): six consecutive floats of the pair's class rows and eight consecutive id words — one base address each
      const int32_t snp = s_snp2[ti1];
      const float* rsrc = rows + (size_t)snp * 12 + n1 * 6;
#pragma unroll
      for (int i = 0; i < NRR; ++i) d_rows[i] = rsrc[i];
      const uint8_t* isrc = ids + (size_t)snp * V + n1 * 32;
#pragma unroll
      for (int i = 0; i < NRI; ++i) {
        // one (possibly unaligned) 4-byte load per word: gfx950 global loads take any alignment; a row's last word is masked to its V
        // bytes (the bytes beyond belong to the next SNP's row; d_ids is allocated with 16 bytes of slack)
        const int k0 = n1 * 32 + 4 * i;
        uint32_t wv = 0;
        if (k0 < V) {
          __builtin_memcpy(&wv, isrc + 4 * i, 4);
          const int nb = V - k0;
          if (nb < 4) wv &= (1u << (8 * nb)) - 1u;
        }
        d_ids[i] = wv;
      }
      d_rd4 = load_rd4(pv, s_off2[ti1], s_cnt2[ti1]);
      const double* g0 = gp0 + (size_t)snp * 3;
      d_g0[0] = g0[0]; d_g0[1] = g0[1]; d_g0[2] = g0[2];
    };
    auto build = [&](int64_t tbase, int b) {     // tile [tbase, tbase + TP) -> buffer b
      const int tp = (int)min((int64_t)TP, np - tbase);
      double* s_T = s_Tb + (size_t)b * TP * NT;
      uint8_t* s_ids = s_idb + (size_t)b * TP * VSC;
      if (lane < TP) { s_cnt[lane] = pn; s_off[lane] = poff; s_snp[lane] = psn; }
#pragma unroll
      for (int i = 0; i < NRR; ++i) s_rows[ti1 * 12 + n1 * 6 + i] = d_rows[i];
#pragma unroll
      for (int i = 0; i < NRI; ++i) reinterpret_cast<uint32_t*>(s_ids)[ti1 * wpr + n1 * 8 + i] =
          (d_ids[i] << (FAST ? 1 : 2)) | 0x10101010u;                    // class id c as 0x10 | c << 2 (FAST: 0x10 | c << 1), see DMX_PJ_WORD4
      const uint32_t rd4 = d_rd4;
      const double qq[3] = {d_g0[0], d_g0[1], d_g0[2]};
      publish_next();
      load_hdr(tbase + 2 * TP);
      DMX_WAVE_LDS_ORDER();
      // ---- phase 1 (identical to k_doublet_a2 / k_doublet_cls)
      {
        const bool on = ti1 < tp;
        const uint32_t cnt = on ? s_cnt[ti1] : 0u;
        const int64_t off = on ? s_off[ti1] : 0;
        double pG[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) pG[i] = 1.0;
        for (uint32_t r = 0; __any(r < cnt); ++r) {
          const bool live = r < cnt;
          const uint32_t byte = live ? (r < 4 ? (rd4 >> (8 * r)) & 0xFFu : (uint32_t)pv.reads[off + r]) : 0u;
          const uint32_t bq = byte & 127u;
          const bool alt = (byte >> 7) != 0;
          const double pR = alt ? s_tab[128 + bq] : s_tab[bq];
          const double pA = alt ? s_tab[bq] : s_tab[128 + bq];
          double mx = 0.0;
          if (live) {
#pragma unroll
            for (int i = 0; i < 9; ++i) {
              pG[i] *= (pR * s_w[n1][9 + i] + pA * s_w[n1][i]);
              mx = fmax(mx, pG[i]);
            }
          }
          {
            const double o = shfl_xor1(mx);
            mx = fmax(mx, o);
          }
          if (live) {
            if (cnt <= kSafeReads) {
              const double y = rcp_refined(mx);
#pragma unroll
              for (int i = 0; i < 9; ++i) pG[i] = div_by(pG[i], mx, y);
            } else {
#pragma unroll
              for (int i = 0; i < 9; ++i) pG[i] /= mx;
            }
          }
        }
        double mx = 0.0;
#pragma unroll
        for (int i = 0; i < 9; ++i) {
          pG[i] += 1e-6;
          mx = fmax(mx, pG[i]);
        }
        {
          const double o = shfl_xor1(mx);
          mx = fmax(mx, o);
        }
        if (on) {
          const double y = rcp_refined(mx);
          double sum = 0.0;
#pragma unroll
          for (int l = 0; l < 3; ++l)
#pragma unroll
            for (int m = 0; m < 3; ++m) {
              const double v = div_by(pG[l * 3 + m], mx, y);
              s_pG[(ti1 * 2 + n1) * 9 + l * 3 + m] = v;
              sum += ((qq[l] * qq[m]) * v);
            }
          ok &= __builtin_amdgcn_class(sum, 0x100);
          s_t00[n1 * T00 + ti1] = dmx_log2_fast(sum, s_log);
        }
      }
      DMX_WAVE_LDS_ORDER();
      request_next();                                              // (after phase 1: its registers are free again; phase 1b, this wavefront's
                                                                   //  phase 2 and the barrier cover the loads' latency)
      if (lane < 2) {                                              // llks00[n] += the tile's terms, ascending SNP order (:688-704)
        const double* row = &s_t00[lane * T00];
        if (tp == TP) {
          double2 v[TP / 2];
#pragma unroll
          for (int i = 0; i < TP / 2; ++i) v[i] = *reinterpret_cast<const double2*>(&row[2 * i]);
#pragma unroll
          for (int i = 0; i < TP / 2; ++i) { acc00 += v[i].x; acc00 += v[i].y; }
        } else {
          for (int i = 0; i < tp; ++i) acc00 += row[i];
        }
      }
      // ---- phase 1b: the class table (16 entries per lane; FAST: 14 of 16 slots per pair, 8 per lane)
      if constexpr (!FAST)
      for (int e = lane; e < tp * NT; e += 64) {
        const int ti = e / NT, cc = e % NT;
        const int cj = cc >> 3, ck = (cc >> 1) & 3, n = cc & 1;
        const float* rj = &s_rows[ti * 12 + cj * 3];
        const float* rk = &s_rows[ti * 12 + ck * 3];
        const double* P = &s_pG[(ti * 2 + n) * 9];
        const double aj[3] = {(double)rj[0], (double)rj[1], (double)rj[2]};
        const double bk[3] = {(double)rk[0], (double)rk[1], (double)rk[2]};
        double sum = 0.0;                                                          // :674
#pragma unroll
        for (int l = 0; l < 3; ++l)
#pragma unroll
          for (int m = 0; m < 3; ++m) sum += ((aj[l] * bk[m]) * P[l * 3 + m]);     // :553, :677-679
        ok &= __builtin_amdgcn_class(sum, 0x100);
        s_T[ti * NT + cc] = dmx_log2_fast(sum, s_log);                              // the :683 term
      }
      DMX_WAVE_LDS_ORDER();                                        // (pG, t00, rows and the headers are rewritten by the next build)
    };
    if (np > 0) {
      load_hdr(0);
      publish_next();
      load_hdr(TP);
      DMX_WAVE_LDS_ORDER();
      request_next();
      DMX_WAVE_LDS_ORDER();                        // (the next tile's header may be overwritten from here on)
    }
    int b = 1;                                     // the round before the first tile only builds tile 0 (into buffer 0)
    for (int64_t tbase = -TP; tbase < np; tbase += TP, b ^= 1) {
      if (tbase + TP < np) build(tbase + TP, b ^ 1);
      if constexpr (FAST) {
        if (tbase >= 0) {                          // the singlet column [j][0][0] of tile t: lanes = samples j (class 0 beyond V: unused sums)
          const int tp = (int)min((int64_t)TP, np - tbase);
          const uint8_t* s_ids = s_idb + (size_t)b * TP * VSC;
          const double* s_T = s_Tb + (size_t)b * TP * NT;
#pragma unroll 8
          for (int ti = 0; ti < tp; ++ti) {
            const int cj = (s_ids[ti * VSC + lane] >> 1) & 3, c0 = (s_ids[ti * VSC] >> 1) & 3;
            acc0 += s_T[ti * NT + (cj * 4 + c0) * 2];
          }
        }
      }
      __syncthreads();
      if constexpr (FAST) {
        if (tbase + TP < np) phase1b_fast(tbase + TP, b ^ 1, ok);
        __syncthreads();
      }
    }
    if constexpr (FAST) {
      s_pG[lane] = acc0;                           // (pG is the producer's own and idle now) -> the consumers' rows
      __syncthreads();
    }
    if (lane < 2) l00[(size_t)cell * A + lane] = acc00;
    if (!ok) flag_cell(flagged, cell);
  } else {
    // ================================================= consumers ================================================
    const bool owner = lane < V && j0 < V;
    [[maybe_unused]] bool ok = true;               // (FAST: this wavefront's share of phase 1b)
    constexpr int NJM = kPcJ1;
    const int nj = wave == 1 ? kPcJ1 : kPcJ;
    double acc[NJM][A];
#pragma unroll
    for (int kk = 0; kk < NJM; ++kk) { acc[kk][0] = 0.0; acc[kk][1] = 0.0; }
    int b = 1;
    for (int64_t tbase = -TP; tbase < np; tbase += TP, b ^= 1) {
      const int tp = tbase < 0 ? 0 : (int)min((int64_t)TP, np - tbase);
      if (owner && tp > 0) {
        const uint8_t* s_ids = s_idb + (size_t)b * TP * VSC;
        uint32_t pi = (uint32_t)(uintptr_t)(lds_u8)(s_ids + lane), pj = (uint32_t)(uintptr_t)(lds_u8)(s_ids + j0);
        uint32_t pt = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) double*)(s_Tb + (size_t)b * TP * NT);
        uint32_t tmp, m0_keep, colv, ti_c, wa0, wa1, wa2, wa3, wa4, wb0, wb1, wb2, wb3, wb4;
        if (wave == 1) {
          uint32_t wa5, wb5;
          if constexpr (FAST) {
            asm volatile(DMX_PJ_CONSUMER(FA, DMX_PJ_WORDS6, DMX_PJ_RFL6, DMX_PJ_REST20)
              :
                [b0] "+v"(acc[0][1]),                [b1] "+v"(acc[1][1]),                [b2] "+v"(acc[2][1]),                [b3] "+v"(acc[3][1]),
                [b4] "+v"(acc[4][1]),                [b5] "+v"(acc[5][1]),                [b6] "+v"(acc[6][1]),                [b7] "+v"(acc[7][1]),
                [b8] "+v"(acc[8][1]),                [b9] "+v"(acc[9][1]),                [b10] "+v"(acc[10][1]),                [b11] "+v"(acc[11][1]),
                [b12] "+v"(acc[12][1]),                [b13] "+v"(acc[13][1]),                [b14] "+v"(acc[14][1]),                [b15] "+v"(acc[15][1]),
                [b16] "+v"(acc[16][1]),                [b17] "+v"(acc[17][1]),                [b18] "+v"(acc[18][1]),                [b19] "+v"(acc[19][1]),
                [b20] "+v"(acc[20][1]),                [b21] "+v"(acc[21][1]),                [b22] "+v"(acc[22][1]),                [b23] "+v"(acc[23][1]),
                [pj] "+v"(pj), [pi] "+v"(pi), [pt] "+v"(pt), [colv] "=&v"(colv),
                [t] "=&s"(tmp), [m0k] "=&s"(m0_keep), [ti] "=&s"(ti_c),
                [wa0] "=&s"(wa0), [wa1] "=&s"(wa1), [wa2] "=&s"(wa2), [wa3] "=&s"(wa3), [wa4] "=&s"(wa4), [wa5] "=&s"(wa5),
                [wb0] "=&s"(wb0), [wb1] "=&s"(wb1), [wb2] "=&s"(wb2), [wb3] "=&s"(wb3), [wb4] "=&s"(wb4), [wb5] "=&s"(wb5)
              : [tp] "s"(tp)
              : DMX_PJ_CLOB);
          } else {
            asm volatile(DMX_PJ_CONSUMER(ST, DMX_PJ_WORDS6, DMX_PJ_RFL6, DMX_PJ_REST20)
              :
                [a0] "+v"(acc[0][0]), [b0] "+v"(acc[0][1]),                [a1] "+v"(acc[1][0]), [b1] "+v"(acc[1][1]),
                [a2] "+v"(acc[2][0]), [b2] "+v"(acc[2][1]),                [a3] "+v"(acc[3][0]), [b3] "+v"(acc[3][1]),
                [a4] "+v"(acc[4][0]), [b4] "+v"(acc[4][1]),                [a5] "+v"(acc[5][0]), [b5] "+v"(acc[5][1]),
                [a6] "+v"(acc[6][0]), [b6] "+v"(acc[6][1]),                [a7] "+v"(acc[7][0]), [b7] "+v"(acc[7][1]),
                [a8] "+v"(acc[8][0]), [b8] "+v"(acc[8][1]),                [a9] "+v"(acc[9][0]), [b9] "+v"(acc[9][1]),
                [a10] "+v"(acc[10][0]), [b10] "+v"(acc[10][1]),                [a11] "+v"(acc[11][0]), [b11] "+v"(acc[11][1]),
                [a12] "+v"(acc[12][0]), [b12] "+v"(acc[12][1]),                [a13] "+v"(acc[13][0]), [b13] "+v"(acc[13][1]),
                [a14] "+v"(acc[14][0]), [b14] "+v"(acc[14][1]),                [a15] "+v"(acc[15][0]), [b15] "+v"(acc[15][1]),
                [a16] "+v"(acc[16][0]), [b16] "+v"(acc[16][1]),                [a17] "+v"(acc[17][0]), [b17] "+v"(acc[17][1]),
                [a18] "+v"(acc[18][0]), [b18] "+v"(acc[18][1]),                [a19] "+v"(acc[19][0]), [b19] "+v"(acc[19][1]),
                [a20] "+v"(acc[20][0]), [b20] "+v"(acc[20][1]),                [a21] "+v"(acc[21][0]), [b21] "+v"(acc[21][1]),
                [a22] "+v"(acc[22][0]), [b22] "+v"(acc[22][1]),                [a23] "+v"(acc[23][0]), [b23] "+v"(acc[23][1]),
                [pj] "+v"(pj), [pi] "+v"(pi), [pt] "+v"(pt), [colv] "=&v"(colv),
                [t] "=&s"(tmp), [m0k] "=&s"(m0_keep), [ti] "=&s"(ti_c),
                [wa0] "=&s"(wa0), [wa1] "=&s"(wa1), [wa2] "=&s"(wa2), [wa3] "=&s"(wa3), [wa4] "=&s"(wa4), [wa5] "=&s"(wa5),
                [wb0] "=&s"(wb0), [wb1] "=&s"(wb1), [wb2] "=&s"(wb2), [wb3] "=&s"(wb3), [wb4] "=&s"(wb4), [wb5] "=&s"(wb5)
              : [tp] "s"(tp)
              : DMX_PJ_CLOB);
          }
        } else {
          if constexpr (FAST) {
            asm volatile(DMX_PJ_CONSUMER(FA, DMX_PJ_WORDS5, DMX_PJ_RFL5, DMX_PJ_REST16)
              :
                [b0] "+v"(acc[0][1]),                [b1] "+v"(acc[1][1]),                [b2] "+v"(acc[2][1]),                [b3] "+v"(acc[3][1]),
                [b4] "+v"(acc[4][1]),                [b5] "+v"(acc[5][1]),                [b6] "+v"(acc[6][1]),                [b7] "+v"(acc[7][1]),
                [b8] "+v"(acc[8][1]),                [b9] "+v"(acc[9][1]),                [b10] "+v"(acc[10][1]),                [b11] "+v"(acc[11][1]),
                [b12] "+v"(acc[12][1]),                [b13] "+v"(acc[13][1]),                [b14] "+v"(acc[14][1]),                [b15] "+v"(acc[15][1]),
                [b16] "+v"(acc[16][1]),                [b17] "+v"(acc[17][1]),                [b18] "+v"(acc[18][1]),                [b19] "+v"(acc[19][1]),
                [pj] "+v"(pj), [pi] "+v"(pi), [pt] "+v"(pt), [colv] "=&v"(colv),
                [t] "=&s"(tmp), [m0k] "=&s"(m0_keep), [ti] "=&s"(ti_c),
                [wa0] "=&s"(wa0), [wa1] "=&s"(wa1), [wa2] "=&s"(wa2), [wa3] "=&s"(wa3), [wa4] "=&s"(wa4),
                [wb0] "=&s"(wb0), [wb1] "=&s"(wb1), [wb2] "=&s"(wb2), [wb3] "=&s"(wb3), [wb4] "=&s"(wb4)
              : [tp] "s"(tp)
              : DMX_PJ_CLOB);
          } else {
            asm volatile(DMX_PJ_CONSUMER(ST, DMX_PJ_WORDS5, DMX_PJ_RFL5, DMX_PJ_REST16)
              :
                [a0] "+v"(acc[0][0]), [b0] "+v"(acc[0][1]),                [a1] "+v"(acc[1][0]), [b1] "+v"(acc[1][1]),
                [a2] "+v"(acc[2][0]), [b2] "+v"(acc[2][1]),                [a3] "+v"(acc[3][0]), [b3] "+v"(acc[3][1]),
                [a4] "+v"(acc[4][0]), [b4] "+v"(acc[4][1]),                [a5] "+v"(acc[5][0]), [b5] "+v"(acc[5][1]),
                [a6] "+v"(acc[6][0]), [b6] "+v"(acc[6][1]),                [a7] "+v"(acc[7][0]), [b7] "+v"(acc[7][1]),
                [a8] "+v"(acc[8][0]), [b8] "+v"(acc[8][1]),                [a9] "+v"(acc[9][0]), [b9] "+v"(acc[9][1]),
                [a10] "+v"(acc[10][0]), [b10] "+v"(acc[10][1]),                [a11] "+v"(acc[11][0]), [b11] "+v"(acc[11][1]),
                [a12] "+v"(acc[12][0]), [b12] "+v"(acc[12][1]),                [a13] "+v"(acc[13][0]), [b13] "+v"(acc[13][1]),
                [a14] "+v"(acc[14][0]), [b14] "+v"(acc[14][1]),                [a15] "+v"(acc[15][0]), [b15] "+v"(acc[15][1]),
                [a16] "+v"(acc[16][0]), [b16] "+v"(acc[16][1]),                [a17] "+v"(acc[17][0]), [b17] "+v"(acc[17][1]),
                [a18] "+v"(acc[18][0]), [b18] "+v"(acc[18][1]),                [a19] "+v"(acc[19][0]), [b19] "+v"(acc[19][1]),
                [pj] "+v"(pj), [pi] "+v"(pi), [pt] "+v"(pt), [colv] "=&v"(colv),
                [t] "=&s"(tmp), [m0k] "=&s"(m0_keep), [ti] "=&s"(ti_c),
                [wa0] "=&s"(wa0), [wa1] "=&s"(wa1), [wa2] "=&s"(wa2), [wa3] "=&s"(wa3), [wa4] "=&s"(wa4),
                [wb0] "=&s"(wb0), [wb1] "=&s"(wb1), [wb2] "=&s"(wb2), [wb3] "=&s"(wb3), [wb4] "=&s"(wb4)
              : [tp] "s"(tp)
              : DMX_PJ_CLOB);
          }
        }
      }
      __syncthreads();
      if constexpr (FAST) {
        if (tbase + TP < np) phase1b_fast(tbase + TP, b ^ 1, ok);
        __syncthreads();
      }
    }
    if constexpr (FAST) {
      __syncthreads();                             // the producer's singlet column is in s_pG now
      if (!ok) flag_cell(flagged, cell);
    }
    if (owner) {
#pragma unroll
      for (int jj = 0; jj < NJM; ++jj) {
        const int jx = j0 + jj;
        if (jj < nj && jx < V) {
          double* o = grid + (((size_t)cell * V + jx) * V + lane) * A;
          o[0] = FAST ? s_pG[jx] : acc[jj][0]; o[1] = acc[jj][1];
        }
      }
    }
  }
#undef DMX_PJ_C0_SX
#undef DMX_PJ_C1_SX
#undef DMX_PJ_C0_SY
#undef DMX_PJ_C1_SY
#undef DMX_PJ_ADDS_ST
#undef DMX_PJ_ADDS_FA
#undef DMX_PJ_WORD4
#undef DMX_PJ_COLADDR_ST
#undef DMX_PJ_COLADDR_FA
#undef DMX_PJ_READS_ST_SX
#undef DMX_PJ_READS_ST_SY
#undef DMX_PJ_READS_FA_SX
#undef DMX_PJ_READS_FA_SY
#undef DMX_PJ_FIRST4
#undef DMX_PJ_REST16
#undef DMX_PJ_MID
#undef DMX_PJ_WORDS5
#undef DMX_PJ_WORDS6
#undef DMX_PJ_RFL6
#undef DMX_PJ_REST20
#undef DMX_PJ_CONSUMER
#undef DMX_PJ_RFL5
#undef DMX_PJ_NEXT
#undef DMX_PJ_END
#undef DMX_PJ_CLOB
}

// K2 over genotype classes, FAST mode, alpha grid {0, 0.5}: k_doublet_sym's entry set (singlet column + one evaluation per
// unordered pair) with k_doublet_cls's class table.  ONE WAVEFRONT PER BARCODE (four independent barcodes per workgroup, no
// workgroup barrier after the table staging): lane (j, q) owns NED consecutive rotation offsets d = q*NED .. q*NED+NED-1 of
// sample j, i.e. the pairs {j, (j+d) mod V}; lanes q = 0 also own the singlet entry [j][0][0].  Per tile of 32 pairs:
//   stage    headers, class rows (4 x 3 float32) and the pairs' 2-bit class ids as a stream over the periodic sample sequence
//            0..V-1,0..V-1,... (built once per SNP by k_build_classes) so that a lane's consecutive (j+d) mod V never wrap
//   phase 1  the five distinct pG values of alpha 0.5 and the three of alpha 0 (k_doublet_sym), the llks00 terms
//   per sub-tile of 8 pairs:
//   phase 1b T[pair][cj][ck] = log(row_cj . (pG[1] row_ck)) for cj <= ck (mirrored), T[pair][cj][4] = log(row_cj . (pG[0] row_c0)),
//            c0 = class of sample 0; 64-byte rows: a pair's table occupies the 64 LDS banks exactly once (conflict-free lookups)
//   phase 2  per pair and lane: its id window (2-4 words), then per entry one bit-field extract, one address op, one 8-byte LDS
//            lookup and one FP64 add.
template <int NED, int MINW>
__global__ __launch_bounds__(kThreads, MINW) void k_doublet_clsym(PileupView pv, int nrd_width, const float* __restrict__ rows,
                                                               const uint32_t* __restrict__ idd, int32_t nwd2,
                                                               const double* __restrict__ gp0, const double* __restrict__ tabs,
                                                               const int32_t* __restrict__ sched, int32_t V,
                                                               double* __restrict__ grid, double* __restrict__ l00,
                                                               uint8_t* __restrict__ flagged) {
  constexpr int A = 2, TP = 32, TPC = 64, SUBT = 8;
  constexpr int CPW = kThreads / TPC;
  constexpr int T00 = TP + 2;
  constexpr int NW = 10;                          // id-stream words per pair (V <= 64: nwd2 <= 10)
  constexpr int NRW = (2 * NED + 30) / 32 + 1;    // words a lane's window can span
  constexpr int NT = 32;                          // class-table doubles per pair: [cj][8]: ck = 0..3 | singlet | pad
  extern __shared__ __attribute__((aligned(64))) unsigned char s_raw[];
  __shared__ double s_tab[kTab2];
  __shared__ double s_w[2][10];                  // mixing weights per alpha and distinct value (see k_doublet_sym)
  const double* s_log = s_tab + kLut2;
  const int t = threadIdx.x;
  stage_k2_tables(s_tab, tabs, t, kThreads);
  if (t < 10) {
    const int n = t / 5, q = t % 5;
    const int l = n ? (q > 2 ? 2 : q) : min(q, 2), m = n ? q - l : 0;
    const double p = 0.5 * l + (m - l) * 0.5 * (n ? 0.5 : 0.0);
    s_w[n][q] = p;
    s_w[n][5 + q] = 1.0 - p;
  }
  __syncthreads();

  const int cw = t / TPC, tid = t % TPC;
  constexpr size_t cell_bytes = ((size_t)SUBT * NT * 8 + (size_t)TP * 6 * 8 + (size_t)TP * 4 * 8 + 2 * T00 * 8 + TP * (4 + 4 + 8) +
                                 (size_t)TP * 12 * 4 + (size_t)TP * NW * 4 + 63) & ~(size_t)63;
  unsigned char* base = s_raw + (size_t)cw * cell_bytes;
  double* s_T = (double*)base;                                   // [SUBT][4][8]  (a pair's 256 contiguous bytes = each of the 64 banks once)
  double* s_q1 = s_T + SUBT * NT;                                // [TP][6]   pG of alpha 0.5: q[l+m]
  double* s_q0 = s_q1 + TP * 6;                                  // [TP][4]   pG of alpha 0:   q[l]
  double* s_t00 = s_q0 + TP * 4;                                 // [2][T00]
  int64_t* s_off = (int64_t*)(s_t00 + 2 * T00);                  // [TP]
  int32_t* s_snp = (int32_t*)(s_off + TP);                       // [TP]
  uint32_t* s_cnt = (uint32_t*)(s_snp + TP);                     // [TP]
  float* s_rows = (float*)(s_cnt + TP);                          // [TP][4][3]
  uint32_t* s_pk = (uint32_t*)(s_rows + TP * 12);                // [TP][NW]  2-bit ids of samples 0..V-1,0..V-1,...

  const int slot = blockIdx.x * CPW + cw;
  if (slot >= pv.B) return;                      // whole wavefront idle (no workgroup barriers below)
  const int32_t cell = sched[slot];
  const int64_t p_beg = pv.cell_pair_off[cell];
  const int64_t np = pv.cell_pair_off[cell + 1] - p_beg;
  int64_t rd_base = pv.cell_read_off[cell];

  // phase-2 identity
  const int D = V / 2 + 1;                        // rotation offsets 0..V/2
  // Panels whose D offsets do not fit one wavefront's (64 / V) x NED are cut into slabs of offsets, one wavefront (blockIdx.y)
  // each, which repeat the cheap phases 1 and 1b for themselves: no barrier, and more, shorter wavefronts per barcode.
  const int j = tid % V, q = tid / V;
  const int d0 = ((int)blockIdx.y * (TPC / V) + q) * NED;
  const bool lane_on = d0 < D && q * V + V <= TPC;   // the lane owns at least one offset (and is a complete (j,q) row)
  const int p0 = lane_on ? j + d0 : 0;            // first position of its window in the id stream
  const uint32_t w0 = (uint32_t)p0 >> 4, sh0 = ((uint32_t)p0 & 15u) * 2u;
  const uint32_t wj = (uint32_t)j >> 4, shj = ((uint32_t)j & 15u) * 2u;
  const bool sing_owner = tid < V && blockIdx.y == 0;   // lanes q = 0 of the first slab
  double acc[NED], accS = 0.0;
#pragma unroll
  for (int i = 0; i < NED; ++i) acc[i] = 0.0;
  bool ok = true;
  const int ti1 = tid >> 1, n1 = tid & 1;
  double acc00 = 0.0;

  // The two-wavefront-per-SIMD forms (one slab of up to 33 offsets) have registers to spare and little else to hide memory
  // latency with: they run a two-deep pipeline like K1's.  While tile t computes, the header of tile t + 2 and everything tile
  // t + 1 reads from global memory (class rows, id streams, leading read bytes, gp0) are in flight into registers.
  constexpr bool PF = MINW <= 2;
  constexpr int NR_ROWS = TP * 12 / TPC, NR_PK = TP * NW / TPC;   // per-lane registers of a tile's rows (6) and id words (5)
  uint32_t hd_n = 0u; int32_t hd_s = 0;          // header (stored reads, SNP id) of this lane's pair in the NEXT tile (PF: tile + 2)
  uint32_t pn = 0u; int32_t psn = 0; int64_t poff = 0;             // PF: prepared header of the tile whose data is in flight
  float d_rows[NR_ROWS]; uint32_t d_pk[NR_PK]; uint32_t d_rd4 = 0u; double d_g0[3] = {0.0, 0.0, 0.0};
#pragma unroll
  for (int i = 0; i < NR_ROWS; ++i) d_rows[i] = 0.f;
#pragma unroll
  for (int i = 0; i < NR_PK; ++i) d_pk[i] = 0u;
  auto load_hdr = [&](int64_t first) {           // header loads of the tile that starts at pair `first`
    hd_n = 0u; hd_s = 0;
    const int64_t nx = first + tid;
    if (tid < TP && nx < np) { hd_n = load_nrd(pv.pair_nrd, p_beg + nx, nrd_width); hd_s = pv.pair_snp ? pv.pair_snp[p_beg + nx] : (int32_t)nx; }
  };
  auto prepare = [&]() {                         // PF: hd_* (arrived) -> pn/psn/poff, then request that tile's data
    pn = hd_n; psn = hd_s;
    const uint32_t incl = seg_scan_incl<32>(pn);
    poff = rd_base + (int64_t)(incl - pn);
    rd_base += (int64_t)__shfl(incl, 31);        // lanes >= 32 carry zeros: lane 31 holds the tile's read count
#pragma unroll
    for (int i = 0; i < NR_ROWS; ++i) {
      const int e = tid + TPC * i;
      d_rows[i] = rows[(size_t)__shfl(psn, e / 12) * 12 + (e % 12)];
    }
#pragma unroll
    for (int i = 0; i < NR_PK; ++i) {
      const int e = tid + TPC * i, w = e % NW;
      const int32_t sn_e = __shfl(psn, e / NW);
      d_pk[i] = w < nwd2 ? idd[(size_t)sn_e * nwd2 + w] : 0u;
    }
    const uint32_t cnt1 = __shfl(pn, ti1);
    const int64_t off1 = ((int64_t)__shfl((int)(poff >> 32), ti1) << 32) | (uint32_t)__shfl((int)(uint32_t)poff, ti1);
    d_rd4 = load_rd4(pv, off1, cnt1);
    const double* g0 = gp0 + (size_t)__shfl(psn, ti1) * 3;
    d_g0[0] = g0[0]; d_g0[1] = g0[1]; d_g0[2] = g0[2];
  };
  load_hdr(0);
  if (PF) { prepare(); load_hdr(TP); }
  for (int64_t tbase = 0; tbase < np; tbase += TP) {
    const int tp = (int)min((int64_t)TP, np - tbase);
    uint32_t rd4_cur = 0u; double g0_cur[3] = {0.0, 0.0, 0.0};
    if (PF) {
      // this tile's data (requested a tile ago) -> LDS; then prepare the next tile and request the one after's header
      if (tid < TP) { s_cnt[tid] = pn; s_off[tid] = poff; s_snp[tid] = psn; }
#pragma unroll
      for (int i = 0; i < NR_ROWS; ++i) s_rows[tid + TPC * i] = d_rows[i];
#pragma unroll
      for (int i = 0; i < NR_PK; ++i) s_pk[tid + TPC * i] = d_pk[i];
      rd4_cur = d_rd4; g0_cur[0] = d_g0[0]; g0_cur[1] = d_g0[1]; g0_cur[2] = d_g0[2];
      prepare();
      load_hdr(tbase + 2 * TP);
      DMX_WAVE_LDS_ORDER();
    } else {
    if (tid < TP) {
      const uint32_t n = hd_n;                     // this tile's header was requested a tile ago; now request the next one's
      const int32_t sn = hd_s;
      const int64_t nx = tbase + TP + tid;
      hd_n = 0u; hd_s = 0;
      if (nx < np) { hd_n = load_nrd(pv.pair_nrd, p_beg + nx, nrd_width); hd_s = pv.pair_snp ? pv.pair_snp[p_beg + nx] : (int32_t)nx; }
      const uint32_t incl = seg_scan_incl<32>(n);
      s_cnt[tid] = n;
      s_off[tid] = rd_base + (int64_t)(incl - n);
      s_snp[tid] = sn;
    }
    DMX_WAVE_LDS_ORDER();
    rd_base = s_off[tp - 1] + (int64_t)s_cnt[tp - 1];
    // ---- class rows and id streams -> LDS
    for (int e = tid; e < tp * 12; e += TPC) s_rows[e] = rows[(size_t)s_snp[e / 12] * 12 + (e % 12)];
    for (int e = tid; e < tp * NW; e += TPC) {
      const int ti = e / NW, w = e % NW;
      s_pk[e] = w < nwd2 ? idd[(size_t)s_snp[ti] * nwd2 + w] : 0u;
    }
    }
    // ---- phase 1 (k_doublet_sym's: five distinct values per alpha lane)
    {
      const bool on = ti1 < tp;
      const uint32_t cnt = on ? s_cnt[ti1] : 0u;
      const int64_t off = on ? s_off[ti1] : 0;
      const uint32_t rd4 = PF ? rd4_cur : load_rd4(pv, off, cnt);   // the first four read bytes in one load (one dependent latency instead of four)
      const int32_t snp1 = on ? s_snp[ti1] : 0;
      double qv[5], wA[5], wR[5];
#pragma unroll
      for (int i = 0; i < 5; ++i) { qv[i] = 1.0; wA[i] = s_w[n1][i]; wR[i] = s_w[n1][5 + i]; }   // :597
      for (uint32_t r = 0; __any(r < cnt); ++r) {
        const bool live = r < cnt;
        const uint32_t byte = live ? (r < 4 ? (rd4 >> (8 * r)) & 0xFFu : (uint32_t)pv.reads[off + r]) : 0u;
        const uint32_t bq = byte & 127u;
        const bool alt = (byte >> 7) != 0;
        const double pR = alt ? s_tab[128 + bq] : s_tab[bq];                // :606
        const double pA = alt ? s_tab[bq] : s_tab[128 + bq];                // :607
        double mx = 0.0;
        if (live) {
#pragma unroll
          for (int i = 0; i < 5; ++i) {
            qv[i] *= (pR * wR[i] + pA * wA[i]);                             // :625
            mx = fmax(mx, qv[i]);                                 // :626-627
          }
        }
        {
          const double o = shfl_xor1(mx);
          mx = fmax(mx, o);
        }
        if (live) {
          if (cnt <= kSafeReads) {
            const double y = rcp_refined(mx);
#pragma unroll
            for (int i = 0; i < 5; ++i) qv[i] = div_by(qv[i], mx, y);       // :632-639
          } else {
#pragma unroll
            for (int i = 0; i < 5; ++i) qv[i] = div_slow(qv[i], mx);
          }
        }
      }
      double mx = 0.0;
#pragma unroll
      for (int i = 0; i < 5; ++i) {
        qv[i] += 1e-6;                                                       // :649
        mx = fmax(mx, qv[i]);
      }
      {
        const double o = shfl_xor1(mx);
        mx = fmax(mx, o);
      }
      if (on) {
        const double y = rcp_refined(mx);
#pragma unroll
        for (int i = 0; i < 5; ++i) qv[i] = div_by(qv[i], mx, y);            // :656-663
        const double* g0 = gp0 + (size_t)snp1 * 3;
        const double qq[3] = {PF ? g0_cur[0] : g0[0], PF ? g0_cur[1] : g0[1], PF ? g0_cur[2] : g0[2]};
        double sum = 0.0;
#pragma unroll
        for (int l = 0; l < 3; ++l)
#pragma unroll
          for (int m = 0; m < 3; ++m) {
            const double v = n1 ? qv[l + m] : qv[l];                         // pG[n][l][m]
            sum += ((qq[l] * qq[m]) * v);                                    // :555, :702-705
          }
        ok &= __builtin_amdgcn_class(sum, 0x100);
        s_t00[n1 * T00 + ti1] = dmx_log2_fast(sum, s_log);
        if (n1) {
#pragma unroll
          for (int i = 0; i < 5; ++i) s_q1[ti1 * 6 + i] = qv[i];
        } else {
#pragma unroll
          for (int i = 0; i < 3; ++i) s_q0[ti1 * 4 + i] = qv[i];
        }
      }
    }
    DMX_WAVE_LDS_ORDER();
    if (tid < 2) {
      const double* row = &s_t00[tid * T00];
      if (tp == TP) {                              // eight terms at a time: loads first, then the ordered adds
#pragma unroll 1
        for (int h = 0; h < TP; h += 8) {
          double2 v[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] = *reinterpret_cast<const double2*>(&row[h + 2 * i]);
#pragma unroll
          for (int i = 0; i < 4; ++i) { acc00 += v[i].x; acc00 += v[i].y; }
        }
      } else {
        for (int i = 0; i < tp; ++i) acc00 += row[i];
      }
    }
#pragma unroll 1
    for (int sub = 0; sub < tp; sub += SUBT) {
      const int ns = min(SUBT, tp - sub);
      // ---- phase 1b: the class table of the sub-tile's pairs (bilinear form row_cj . (pG row_ck)); 14 items per pair:
      //      ten unordered class pairs cj <= ck (stored to both [cj][ck] and [ck][cj]) and the four singlet entries
      for (int e = tid; e < ns * 14; e += TPC) {
        const int pi = e / 14, c = e % 14;
        const int ti = sub + pi;
        const int cj = c < 10 ? (int)((0x3221110000ull >> (4 * c)) & 15u) : c - 10;        // (0,0)(0,1)(0,2)(0,3)(1,1)(1,2)(1,3)(2,2)(2,3)(3,3)
        const int ck = c < 10 ? (int)((0x3323213210ull >> (4 * c)) & 15u) : (int)(s_pk[ti * NW] & 3u);   // singlet: sample 0's class
        const float* rj = &s_rows[ti * 12 + cj * 3];
        const float* rk = &s_rows[ti * 12 + ck * 3];
        const double a0 = (double)rj[0], a1 = (double)rj[1], a2 = (double)rj[2];
        const double b0 = (double)rk[0], b1 = (double)rk[1], b2 = (double)rk[2];
        double u0, u1, u2;
        if (c < 10) {
          const double* P = &s_q1[ti * 6];                                   // pG[1][l][m] = P[l + m]
          u0 = __builtin_fma(P[2], b2, __builtin_fma(P[1], b1, P[0] * b0));
          u1 = __builtin_fma(P[3], b2, __builtin_fma(P[2], b1, P[1] * b0));
          u2 = __builtin_fma(P[4], b2, __builtin_fma(P[3], b1, P[2] * b0));
        } else {
          const double* P = &s_q0[ti * 4];                                   // pG[0][l][m] = P[l]
          u0 = __builtin_fma(P[0], b2, __builtin_fma(P[0], b1, P[0] * b0));
          u1 = __builtin_fma(P[1], b2, __builtin_fma(P[1], b1, P[1] * b0));
          u2 = __builtin_fma(P[2], b2, __builtin_fma(P[2], b1, P[2] * b0));
        }
        const double sum = __builtin_fma(a2, u2, __builtin_fma(a1, u1, a0 * u0));
        ok &= __builtin_amdgcn_class(sum, 0x100);
        const double lv = dmx_log2_fast(sum, s_log);
        if (c < 10) { s_T[pi * NT + cj * 8 + ck] = lv; s_T[pi * NT + ck * 8 + cj] = lv; }
        else s_T[pi * NT + cj * 8 + 4] = lv;
      }
      DMX_WAVE_LDS_ORDER();
      // ---- phase 2.  LDS byte address of an entry's table cell = (row base of this pair and this lane's class cj, a multiple of
      //      64) | (ck << 3): one shift and one and-or per entry.  Entries go in groups of GRP (their loads in flight together).
      if (lane_on) {
#ifdef DMX_CLSYM_GRP
        constexpr int GRP = NED <= DMX_CLSYM_GRP ? NED : (NED + (NED + DMX_CLSYM_GRP - 1) / DMX_CLSYM_GRP - 1) / ((NED + DMX_CLSYM_GRP - 1) / DMX_CLSYM_GRP);   // kernel experiments only
#else
        constexpr int GRP = NED <= 12 ? NED : (NED + 2) / 3;
#endif
        using lds_cd = const __attribute__((address_space(3))) double*;
        using lds_cu = const __attribute__((address_space(3))) uint32_t*;
        const uint32_t t_base = (uint32_t)(uintptr_t)(lds_cd)s_T;            // 64-byte aligned (checked by the launcher's layout)
        const uint32_t pk_lane = (uint32_t)(uintptr_t)(lds_cu)s_pk + 4u * w0, pk_j = (uint32_t)(uintptr_t)(lds_cu)s_pk + 4u * wj;
#ifdef DMX_CLSYM_UNR
        constexpr int UNR = DMX_CLSYM_UNR;             // kernel experiments only
#else
        constexpr int UNR = NED >= 12 ? 1 : 2;
#endif
#pragma unroll UNR
        for (int pi = 0; pi < ns; ++pi) {
          const uint32_t po = (uint32_t)((sub + pi) * NW * 4);
          const uint32_t cj = (*(lds_cu)(uintptr_t)(pk_j + po) >> shj) & 3u;
          uint32_t wd[NRW];
#pragma unroll
          for (int r = 0; r < NRW; ++r) wd[r] = *(lds_cu)(uintptr_t)(pk_lane + po + 4u * r);
          const uint32_t row = t_base + (uint32_t)(pi * NT * 8) + (cj << 6);
          uint32_t bits[(NED + 15) / 16];
#pragma unroll
          for (int r = 0; r < (NED + 15) / 16; ++r)     // ids of (j + d0 + i) mod V: 16 per funnel-shifted word
            bits[r] = (r + 1 < NRW) ? __builtin_amdgcn_alignbit(wd[r + 1], wd[r], sh0) : (wd[r] >> sh0);
#pragma unroll
          for (int g0 = 0; g0 < NED; g0 += GRP) {
            double tv[GRP];
#pragma unroll
            for (int i = g0; i < g0 + GRP && i < NED; ++i) {
              const int sh = 2 * (i % 16) - 3;           // ck << 3
              const uint32_t b = bits[i / 16];
              const uint32_t a = row | ((sh >= 0 ? (b >> sh) : (b << -sh)) & 24u);
              tv[i - g0] = *(lds_cd)(uintptr_t)a;
            }
#pragma unroll
            for (int i = g0; i < g0 + GRP && i < NED; ++i) acc[i] += tv[i - g0];
            if (g0 + GRP < NED) __builtin_amdgcn_sched_barrier(0);
          }
          if (sing_owner) accS += *(lds_cd)(uintptr_t)(row + 32u);
        }
      }
      DMX_WAVE_LDS_ORDER();
    }
  }
  {
    double* G = grid + (size_t)cell * V * V * A;
    if (lane_on) {
#pragma unroll
      for (int i = 0; i < NED; ++i) {
        const int d = d0 + i;
        if (d < D) {
          int k = j + d; k = k >= V ? k - V : k;
          if (!(2 * d == V && j > k)) {            // d = V/2 is reached from both sides: the j < k lane stores
            G[((size_t)j * V + k) * A + 1] = acc[i];
            G[((size_t)k * V + j) * A + 1] = acc[i];
          }
        }
      }
      if (sing_owner) for (int k = 0; k < V; ++k) G[((size_t)j * V + k) * A] = accS;
    }
    if (tid < 2 && blockIdx.y == 0) l00[(size_t)cell * A + tid] = acc00;
    if (!ok) flag_cell(flagged, cell);
  }
}

// K2 over genotype classes for alpha grids of 3..8 entries: k_doublet_cls with k_doublet_an's phase 1.  The class table holds
// T[pair][cj][ck][n] for the AP padded alphas; phase 2 is AP/2 16-byte lookups and A adds per (j, k).
template <int TPC, int NK, int AP>
__global__ __launch_bounds__(kThreads) void k_doublet_clsn(PileupView pv, int nrd_width, const float* __restrict__ rows,
                                                           const uint8_t* __restrict__ ids, const double* __restrict__ gp0,
                                                           const double* __restrict__ tabs, const double* __restrict__ alpha,
                                                           const int32_t* __restrict__ sched, int32_t V, int32_t A, int32_t VS,
                                                           double* __restrict__ grid, double* __restrict__ l00,
                                                           uint8_t* __restrict__ flagged) {
  static_assert(AP == 4 || AP == 8, "alphas per pair padded to a power of two");
  constexpr int TP = 32;
  constexpr int CPW = kThreads / TPC;
  constexpr int T00 = TP + 2;
  constexpr int NT = kMaxCls * kMaxCls * AP;     // class-table entries per pair
#define DMX_K2_SYNC() do { if (TPC == 64) { DMX_WAVE_LDS_ORDER(); } else { __syncthreads(); } } while (0)
  extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
  __shared__ double s_tab[kTab2];
  const double* s_log = s_tab + kLut2;
  const int t = threadIdx.x;
  stage_k2_tables(s_tab, tabs, t, kThreads);
  __syncthreads();

  const int cw = t / TPC, tid = t % TPC;
  const size_t cell_bytes = (size_t)TP * AP * 9 * 8 + (size_t)TP * NT * 8 + (size_t)AP * T00 * 8 + TP * (4 + 4 + 8) + (size_t)TP * 12 * 4 + (size_t)TP * VS;
  unsigned char* base = s_raw + (size_t)cw * ((cell_bytes + 15) & ~(size_t)15);
  double* s_pG = (double*)base;                                  // [TP][AP][9]
  double* s_T = s_pG + TP * AP * 9;                              // [TP][4][4][AP]
  double* s_t00 = s_T + TP * NT;                                 // [AP][T00]
  int64_t* s_off = (int64_t*)(s_t00 + AP * T00);                 // [TP]
  int32_t* s_snp = (int32_t*)(s_off + TP);                       // [TP]
  uint32_t* s_cnt = (uint32_t*)(s_snp + TP);                     // [TP]
  float* s_rows = (float*)(s_cnt + TP);                          // [TP][4][3]
  uint8_t* s_ids = (uint8_t*)(s_rows + TP * 12);                 // [TP][VS]

  const int slot = blockIdx.x * CPW + cw;
  if (TPC == 64 && slot >= pv.B) return;
  const bool cell_ok = slot < pv.B;
  const int32_t cell = cell_ok ? sched[slot] : 0;
  const int64_t p_beg = cell_ok ? pv.cell_pair_off[cell] : 0;
  const int64_t np = cell_ok ? pv.cell_pair_off[cell + 1] - p_beg : 0;
  int64_t rd_base = cell_ok ? pv.cell_read_off[cell] : 0;

  const int KB = (V + NK - 1) / NK;
  const int JS = TPC / KB;
  const int jl = tid / KB, kb = tid % KB;
  const int j = (int)blockIdx.y * JS + jl;
  const bool owner = jl < JS && j < V;
  double acc[NK][AP];
#pragma unroll
  for (int kk = 0; kk < NK; ++kk)
#pragma unroll
    for (int n = 0; n < AP; ++n) acc[kk][n] = 0.0;
  bool ok = true;
  double acc00 = 0.0;
  constexpr int P1 = TP * AP;                    // phase-1 lanes per tile
  constexpr int NPASS = (P1 + TPC - 1) / TPC;
  const int n1 = tid % AP;                       // this thread's alpha in phase 1 (TPC is a multiple of AP)
  const bool n_ok = n1 < A;
  double wA[9], wR[9];
  {
    const double al = n_ok ? alpha[n1] : 0.0;
#pragma unroll
    for (int l = 0; l < 3; ++l)
#pragma unroll
      for (int m = 0; m < 3; ++m) {
        const double p = 0.5 * l + (m - l) * 0.5 * al;                       // :613
        wA[l * 3 + m] = p;
        wR[l * 3 + m] = 1.0 - p;
      }
  }


  for (int64_t tbase = 0; tbase < np; tbase += TP) {
    const int tp = (int)min((int64_t)TP, np - tbase);
    if (tid < TP) {
      const bool v = tid < tp;
      const uint32_t n = v ? load_nrd(pv.pair_nrd, p_beg + tbase + tid, nrd_width) : 0u;
      const uint32_t incl = seg_scan_incl<32>(n);
      s_cnt[tid] = n;
      s_off[tid] = rd_base + (int64_t)(incl - n);
      s_snp[tid] = v ? (pv.pair_snp ? pv.pair_snp[p_beg + tbase + tid] : (int32_t)(tbase + tid)) : 0;
    }
    DMX_K2_SYNC();
    rd_base = s_off[tp - 1] + (int64_t)s_cnt[tp - 1];
    for (int e = tid; e < tp * 12; e += TPC) s_rows[e] = rows[(size_t)s_snp[e / 12] * 12 + (e % 12)];
    {
      const int wpr = VS / 4;
      for (int e = tid; e < tp * wpr; e += TPC) {
        const int ti = e / wpr, wq = e % wpr;
        const uint8_t* src = ids + (size_t)s_snp[ti] * V + wq * 4;
        uint32_t wv = 0;
        for (int b = 0; b < 4; ++b) if (wq * 4 + b < V) wv |= (uint32_t)src[b] << (8 * b);
        reinterpret_cast<uint32_t*>(s_ids)[ti * wpr + wq] = wv;
      }
    }
    // ---- phase 1: lane u = (pair u / AP, alpha u % AP)
#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass) {
      const int u = tid + pass * TPC;
      if (u >= P1) break;                          // uniform per wavefront (P1 and TPC are multiples of 64)
      const int ti1 = u / AP;
      const bool on = ti1 < tp && n_ok;
      const uint32_t cnt = (ti1 < tp) ? s_cnt[ti1] : 0u;
      const int64_t off = (ti1 < tp) ? s_off[ti1] : 0;
      double pG[9];
#pragma unroll
      for (int i = 0; i < 9; ++i) pG[i] = 1.0;                               // :597
      for (uint32_t r = 0; __any(r < cnt); ++r) {
        const bool live = r < cnt && n_ok;
        const uint32_t byte = (r < cnt) ? pv.reads[off + r] : 0u;
        const uint32_t bq = byte & 127u;
        const bool alt = (byte >> 7) != 0;
        const double pR = alt ? s_tab[128 + bq] : s_tab[bq];                // :606
        const double pA = alt ? s_tab[bq] : s_tab[128 + bq];                // :607
        double mx = 0.0;
        if (live) {
#pragma unroll
          for (int i = 0; i < 9; ++i) {
            pG[i] *= (pR * wR[i] + pA * wA[i]);                             // :625
            mx = fmax(mx, pG[i]);                                 // :626-627
          }
        }
#pragma unroll
        for (int d = 1; d < AP; d <<= 1) {                                  // one max across ALL alphas of the pair
          const double o = __shfl_xor(mx, d);
          mx = fmax(mx, o);
        }
        if (live) {
          if (cnt <= kSafeReads) {
            const double y = rcp_refined(mx);
#pragma unroll
            for (int i = 0; i < 9; ++i) pG[i] = div_by(pG[i], mx, y);       // :632-639
          } else {
#pragma unroll
            for (int i = 0; i < 9; ++i) pG[i] /= mx;
          }
        }
      }
      double mx = 0.0;
      if (n_ok) {
#pragma unroll
        for (int i = 0; i < 9; ++i) {
          pG[i] += 1e-6;                                                     // :649
          mx = fmax(mx, pG[i]);
        }
      }
#pragma unroll
      for (int d = 1; d < AP; d <<= 1) {
        const double o = __shfl_xor(mx, d);
        mx = fmax(mx, o);
      }
      if (on) {
        const double y = rcp_refined(mx);
        const double* g0 = gp0 + (size_t)s_snp[ti1] * 3;
        const double qq[3] = {g0[0], g0[1], g0[2]};
        double sum = 0.0;
#pragma unroll
        for (int l = 0; l < 3; ++l)
#pragma unroll
          for (int m = 0; m < 3; ++m) {
            const double v = div_by(pG[l * 3 + m], mx, y);                   // :656-663
            s_pG[(ti1 * AP + n1) * 9 + l * 3 + m] = v;
            sum += ((qq[l] * qq[m]) * v);                                    // gp00 (:555) then :702-705
          }
        ok &= __builtin_amdgcn_class(sum, 0x100);
        s_t00[n1 * T00 + ti1] = dmx_log2_fast(sum, s_log);                    // :708-709 term
      }
    }
    DMX_K2_SYNC();
    if (tid < A) {                                 // llks00[n]: lane n adds its alpha's terms in pair order
      const double* row = &s_t00[tid * T00];
      for (int i = 0; i < tp; ++i) acc00 += row[i];
    }
    // ---- phase 1b: the class table (padding alphas are skipped; their slots are never read)
    for (int e = tid; e < tp * NT; e += TPC) {
      const int ti = e / NT, cc = e % NT;
      const int n = cc % AP, ck = (cc / AP) & 3, cj = cc / (AP * 4);
      if (n >= A) continue;
      const float* rj = &s_rows[ti * 12 + cj * 3];
      const float* rk = &s_rows[ti * 12 + ck * 3];
      const double* P = &s_pG[(ti * AP + n) * 9];
      const double aj[3] = {(double)rj[0], (double)rj[1], (double)rj[2]};
      const double bk[3] = {(double)rk[0], (double)rk[1], (double)rk[2]};
      double sum = 0.0;                                                          // :674
#pragma unroll
      for (int l = 0; l < 3; ++l)
#pragma unroll
        for (int m = 0; m < 3; ++m) sum += ((aj[l] * bk[m]) * P[l * 3 + m]);     // :553, :677-679
      ok &= __builtin_amdgcn_class(sum, 0x100);
      s_T[ti * NT + cc] = dmx_log2_fast(sum, s_log);                              // the :683 term
    }
    DMX_K2_SYNC();
    // ---- phase 2
    if (owner) {
      for (int ti = 0; ti < tp; ++ti) {
        const uint8_t* idr = &s_ids[ti * VS];
        const int cj = idr[j];
        const double* Tj = &s_T[ti * NT + cj * 4 * AP];
#pragma unroll
        for (int kk = 0; kk < NK; ++kk) {
          const int k = min(kb * NK + kk, V - 1);
          const double* Tk = &Tj[idr[k] * AP];
#pragma unroll
          for (int h = 0; h < AP / 2; ++h) {
            if (2 * h >= A) break;
            const double2 tv = *reinterpret_cast<const double2*>(&Tk[2 * h]);
            acc[kk][2 * h] += tv.x;                                               // :683
            if (2 * h + 1 < A) acc[kk][2 * h + 1] += tv.y;
          }
        }
      }
    }
    DMX_K2_SYNC();
  }
  if (cell_ok) {
    if (owner) {
#pragma unroll
      for (int kk = 0; kk < NK; ++kk) {
        const int k = kb * NK + kk;
        if (k < V) {
          double* o = grid + (((size_t)cell * V + j) * V + k) * A;
#pragma unroll
          for (int n = 0; n < AP; ++n) if (n < A) o[n] = acc[kk][n];
        }
      }
    }
    if (tid < A && blockIdx.y == 0) l00[(size_t)cell * A + tid] = acc00;
    if (!ok) flag_cell(flagged, cell);
  }
#undef DMX_K2_SYNC
}

// Which genotype matrices can make a phase-2 log() argument leave the normal positive range at all?  Only those with a row that
// holds a NaN / infinity / negative entry or no entry above 2^-400 (see k_doublet_a2's CHK).  Sets *unsafe when it finds one.
__global__ void k_check_geno(const float* __restrict__ g, int64_t n_rows, int32_t* __restrict__ unsafe) {
  bool bad = false;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_rows; i += (int64_t)gridDim.x * blockDim.x) {
    const float a = g[3 * i], b = g[3 * i + 1], c = g[3 * i + 2];
    const bool fin = (a >= 0.f) && (b >= 0.f) && (c >= 0.f) && (a <= 3.0e38f) && (b <= 3.0e38f) && (c <= 3.0e38f);   // false for NaN
    bad |= !fin || !(fmaxf(a, fmaxf(b, c)) >= 1e-30f);             // float32: anything normal is far above 2^-400
  }
  if (bad) atomicOr(unsafe, 1);
}

// ---------------------------------------------------------------------------------------------------------------------
// K3.  One cell per workgroup over its finished grid.
struct ArgMax { double v; int32_t i; };
__device__ __forceinline__ ArgMax better(ArgMax a, ArgMax b) {      // larger value wins; equal values: lower scan index
  return (b.v > a.v || (b.v == a.v && b.i < a.i)) ? b : a;
}

__global__ __launch_bounds__(kThreads) void k_reduce(const double* __restrict__ grid, const double* __restrict__ l00,
                                                     const int64_t* __restrict__ cell_pair_off,
                                                     const double* __restrict__ alpha, int32_t V, int32_t A,
                                                     double prior, dmx_cell_summary* __restrict__ out,
                                                     double* __restrict__ sing) {
  __shared__ double s_d[kThreads];
  __shared__ double s_e[kThreads];
  __shared__ int32_t s_i[kThreads];
  const int t = threadIdx.x;
  const int32_t cell = blockIdx.x;
  const int32_t nAB = V * V * A;
  const double* G = grid + (size_t)cell * nAB;
  const int32_t npairs = (int32_t)(cell_pair_off[cell + 1] - cell_pair_off[cell]);

  for (int32_t jj = t; jj < V; jj += kThreads) sing[(size_t)cell * V + jj] = G[(size_t)jj * V * A];   // llksAB[j][0][0]
  // (1) max over the whole grid (:713-721)
  double mx = -1e300;
  for (int32_t q = t; q < nAB; q += kThreads) mx = fmax(mx, G[q]);
  s_d[t] = mx;
  __syncthreads();
  for (int s = kThreads / 2; s > 0; s >>= 1) {
    if (t < s) s_d[t] = (s_d[t] < s_d[t + s]) ? s_d[t + s] : s_d[t];
    __syncthreads();
  }
  const double max_llk = s_d[0];
  __syncthreads();

  // (2) posterior sums (:724-734) — same per-term expression, tree order for the sum
  double ss = 0.0, sd = 0.0;
  for (int32_t q = t; q < nAB; q += kThreads) {
    const int32_t n = q % A, jk = q / A, j = jk / V, k = jk % V;
    const double e = exp(G[q] - max_llk);
    if (k == 0 && n == 0) ss += (e * (1. - prior) / V);
    if (j != k && n >= 1) sd += (e * prior / V / (V - 1) / (A - 1) / (alpha[n] == 0.5 ? 2.0 : 1.0));
  }
  s_d[t] = ss; s_e[t] = sd;
  __syncthreads();
  for (int s = kThreads / 2; s > 0; s >>= 1) {
    if (t < s) { s_d[t] += s_d[t + s]; s_e[t] += s_e[t + s]; }
    __syncthreads();
  }
  const double sum_single = s_d[0], sum_double = s_e[0];
  __syncthreads();

  // (3) best doublet: first maximum in (j,k,n) scan order over j != k, n >= 1 (:799-814)
  ArgMax bd{-1e300, 0x7FFFFFFF};
  for (int32_t q = t; q < nAB; q += kThreads) {
    const int32_t n = q % A, jk = q / A, j = jk / V, k = jk % V;
    if (j != k && n >= 1) bd = better(bd, ArgMax{G[q], q});
  }
  s_d[t] = bd.v; s_i[t] = bd.i;
  __syncthreads();
  for (int s = kThreads / 2; s > 0; s >>= 1) {
    if (t < s) { ArgMax r = better(ArgMax{s_d[t], s_i[t]}, ArgMax{s_d[t + s], s_i[t + s]}); s_d[t] = r.v; s_i[t] = r.i; }
    __syncthreads();
  }
  const int32_t qbest = s_i[0];
  __syncthreads();

  // (4) top-2 singlets from llksAB[j][0][0] (:746-758)
  ArgMax b1{-1e300, 0x7FFFFFFF};
  for (int32_t j = t; j < V; j += kThreads) b1 = better(b1, ArgMax{G[(size_t)j * V * A], j});
  s_d[t] = b1.v; s_i[t] = b1.i;
  __syncthreads();
  for (int s = kThreads / 2; s > 0; s >>= 1) {
    if (t < s) { ArgMax r = better(ArgMax{s_d[t], s_i[t]}, ArgMax{s_d[t + s], s_i[t + s]}); s_d[t] = r.v; s_i[t] = r.i; }
    __syncthreads();
  }
  const int32_t i1 = s_i[0];
  __syncthreads();
  ArgMax b2{-1e300, 0x7FFFFFFF};
  for (int32_t j = t; j < V; j += kThreads) if (j != i1) b2 = better(b2, ArgMax{G[(size_t)j * V * A], j});
  s_d[t] = b2.v; s_i[t] = b2.i;
  __syncthreads();
  for (int s = kThreads / 2; s > 0; s >>= 1) {
    if (t < s) { ArgMax r = better(ArgMax{s_d[t], s_i[t]}, ArgMax{s_d[t + s], s_i[t + s]}); s_d[t] = r.v; s_i[t] = r.i; }
    __syncthreads();
  }
  const int32_t i2 = (s_i[0] == 0x7FFFFFFF) ? -1 : s_i[0];

  // (5) decisions within 1e-7 of an alternative (other than the alpha = 0.5 mirror of the best doublet, which every cell has):
  //     the host fetches the grid of such a cell and lets the tie arbiter re-evaluate the contenders (DESIGN.md "Ties")
  int32_t flags = 0;
  {
    const double tol = 1e-7;
    bool near_d = false;
    if (qbest != 0x7FFFFFFF) {
      const double best = G[qbest];
      const int32_t nb = qbest % A, jkb = qbest / A, jb = jkb / V, kb = jkb % V;
      const int32_t qmir = (alpha[nb] == 0.5) ? ((kb * V + jb) * A + nb) : -1;
      for (int32_t q = t; q < nAB; q += kThreads) {
        const int32_t n = q % A, jk = q / A, j = jk / V, k = jk % V;
        if (j != k && n >= 1 && q != qbest && q != qmir && G[q] >= best - tol) near_d = true;
      }
    }
    int cnt = 0;
    const double s1v = (i1 != 0x7FFFFFFF) ? G[(size_t)i1 * V * A] : 0.0, s2v = (i2 >= 0) ? G[(size_t)i2 * V * A] : -1e300;
    for (int32_t j = t; j < V; j += kThreads) if (G[(size_t)j * V * A] >= s2v - tol) ++cnt;
    const int any_d = __syncthreads_or(near_d ? 1 : 0);
    s_i[t] = cnt;
    __syncthreads();
    for (int s = kThreads / 2; s > 0; s >>= 1) {
      if (t < s) s_i[t] += s_i[t + s];
      __syncthreads();
    }
    const int n_near_s = s_i[0];
    if (any_d) flags |= DMX_CELL_NEAR_DOUBLET;
    if (i2 >= 0 && (n_near_s > 2 || s1v - s2v < tol)) flags |= DMX_CELL_NEAR_SINGLET;
  }

  if (t == 0) {
    // NaN likelihoods (e.g. a GP record with a missing sample poisons the whole SNP, bcf_filtered_reader.cpp:431-448) make
    // every `<` of the reference's scans false: it then indexes with -1 (:816-825, undefined behaviour).  Here the
    // indices stay -1 and the dependent values are NaN; nothing is read out of bounds.
    const double kNaN = __builtin_nan("");
    dmx_cell_summary r;
    r.max_llk = max_llk; r.sum_single = sum_single; r.sum_double = sum_double;
    const bool has1 = i1 != 0x7FFFFFFF, hasb = qbest != 0x7FFFFFFF;
    r.i_sing1 = has1 ? i1 : -1; r.i_sing2 = i2;
    r.sing_llk1 = has1 ? G[(size_t)i1 * V * A] : kNaN;
    r.sing_llk2 = (i2 >= 0) ? G[(size_t)i2 * V * A] : -1e300;
    const int32_t nb = hasb ? qbest % A : 0, jkb = hasb ? qbest / A : 0, jb = jkb / V, kb = jkb % V;
    r.j_best = hasb ? jb : -1; r.k_best = hasb ? kb : -1; r.n_best = hasb ? nb : -1;
    r.llk12 = hasb ? G[qbest] : kNaN;
    r.llk1 = hasb ? G[(size_t)jb * V * A] : kNaN; r.llk2 = hasb ? G[(size_t)kb * V * A] : kNaN;
    r.llk10 = hasb ? G[(size_t)jb * V * A + nb] : kNaN; r.llk20 = hasb ? G[(size_t)kb * V * A + nb] : kNaN;    // :824-825
    r.llk00_0 = l00[(size_t)cell * A]; r.llk00_best = l00[(size_t)cell * A + nb];
    // (6) the BEST rule's own comparisons (:837,:844) with a margin below 1e-7: the writers re-evaluate the entries involved (dmx::near_rule)
    if (hasb && has1 && i2 >= 0) {
      const double tol = 1e-7, s1 = r.sing_llk1, s2 = r.sing_llk2;
      if (fabs(r.llk12 - (s1 + 2)) < tol || fabs(s1 - (s2 + 2)) < tol ||
          ((fabs(r.llk12 - r.llk1) < tol || fabs(r.llk12 - r.llk2) < tol) && r.llk12 > s1 + 2 - tol))
        flags |= DMX_CELL_NEAR_RULE;
    }
    r.n_pairs = npairs; r.flags = flags; r.reserved = 0; r.llk_ab = 0.0; r.llk_ba = 0.0;
    r.llk_ab_alt = 0.0; r.llk_ba_alt = 0.0; r.ev_x_ab = 0.0; r.ev_t_ab = 0.0; r.ev_x_ba = 0.0; r.ev_t_ba = 0.0;
    out[cell] = r;
  }
}

// K3b — the tie-order certificate (DESIGN.md "Ties").  At alpha = 0.5 the reference's llksAB[j][k] and llksAB[k][j] are one number
// mathematically and differ by the rounding noise of its own evaluation order; its strict-< scan then names the doublet
// "a-b" or "b-a" by that noise.  To print the same order one has to know BOTH accumulators as the reference computes them, bit for
// bit.  Every operation of the reference except log() is an IEEE operation the device reproduces exactly; for log() the device
// evaluates hi + lo = log(x) in double-double and knows which double(s) a libm with < 0.55 ulp error can return
// (dmx_log_bracket).  One wavefront per barcode re-walks the barcode's pairs for its best alpha = 0.5 pair {a, b} only and
// accumulates, in the reference's order, a LOWER and an UPPER bound of each of the two accumulators (IEEE addition is monotone in
// both operands).  When lower == upper for both, they ARE the reference's values and the order is decided here — for any such
// libm; the host tie arbiter (which calls the host's log()) is left with the barcodes where a bracket stayed open, about one in
// ten.  Requires A = 2 (phase 1 is k_doublet_a2's) and no other doublet entry within 1e-7 of the best (K3's flag).
template <int MINW, bool FIVE, bool DENSE = false>   // FIVE: alpha[0] == 0 (the default grid): five distinct phase-1 values per lane instead of nine;
                                                     // DENSE: gT != NULL (SNP-minor columns) — compile-time strides, so the sparse form's three row entries are one load
__global__ __launch_bounds__(kThreads, MINW) void k_certify(PileupView pv, int nrd_width, const float* __restrict__ g,
                                                      const float* __restrict__ gT,
                                                      const double* __restrict__ tabs, const double* __restrict__ alpha,
                                                      int32_t V, dmx_cell_summary* __restrict__ summ,
                                                      const int64_t* __restrict__ blk, int32_t blk_i, int32_t nblk, double* __restrict__ park,
                                                      const double* __restrict__ cseed, const double* __restrict__ cfin) {
  // blk != nullptr (round 4): this launch covers SNP block blk_i of nblk only — sparse pileups whose genotype matrix does not fit an XCD's L2
  // gather two pieces of a V*12-byte row per pair, and walking the SNP axis block by block (launch_certify; the table is k_snp_blocks',
  // shared with sparse K1) keeps the rows a launch touches L2-resident.  Between launches a barcode's state — the four chains, the open
  // events, the two clean flags, the class-test flag — is parked in park[cell][kPark]; the chains add the same terms in the same order.
  constexpr int TP = 32, T00 = TP + 2, TPC = 64, CPW = kThreads / TPC;
  __shared__ double s_tab[kTab];
  __shared__ double s_lo[128];
  __shared__ __attribute__((aligned(16))) double s_t[CPW][6][T00];   // [ab_lo | ab_hi | ba_lo | ba_hi | ab_x | ba_x][pair]
  __shared__ double s_p[CPW][4][T00];                                // the four chains before each step of the tile (+ after the last)
  __shared__ double s_ev[CPW][4];
  __shared__ int64_t s_offs[CPW][TP];
  __shared__ int32_t s_snps[CPW][TP];
  __shared__ uint32_t s_cnts[CPW][TP];
  const int t = threadIdx.x;
  for (int i = t; i < kTab; i += kThreads) s_tab[i] = tabs[i];
  if (t < 128) s_lo[t] = tabs[kTabLogLo + t];
  __syncthreads();
  const double* s_log = s_tab + kLut;
  const int cw = t / TPC, tid = t % TPC;
  const int32_t cell = blockIdx.x * CPW + cw;
  if (cell >= pv.B) return;
  // only the five words the walk needs stay live (wave-uniform); the record itself is rewritten in place by lane 0
  const int32_t sm_np = summ[cell].n_pairs, sm_j = __builtin_amdgcn_readfirstlane(summ[cell].j_best),
                sm_k = __builtin_amdgcn_readfirstlane(summ[cell].k_best), sm_n = summ[cell].n_best, sm_fl = summ[cell].flags;
  if (sm_np <= 0 || sm_j < 0 || sm_k < 0 || sm_n != 1 || alpha[1] != 0.5 || (sm_fl & DMX_CELL_NEAR_DOUBLET)) return;
  const int32_t ia = min(sm_j, sm_k), ib = max(sm_j, sm_k);
  int64_t* s_off = s_offs[cw]; int32_t* s_snp = s_snps[cw]; uint32_t* s_cnt = s_cnts[cw];
  // (blk_i = first table block of this launch | blocks per launch << 16: a launch may cover several of K1's blocks)
  const int32_t b0 = blk_i & 0xFFFF, bs = max(1, blk_i >> 16), b1 = min(b0 + bs, nblk);
  const int64_t* bt = blk ? blk + ((size_t)cell * (nblk + 1) + b0) * 2 : nullptr;
  const int64_t p_beg = blk ? bt[0] : pv.cell_pair_off[cell];
  const int64_t np = (blk ? bt[2 * (b1 - b0)] : pv.cell_pair_off[cell + 1]) - p_beg;
  int64_t rd_base = blk ? bt[1] : pv.cell_read_off[cell];
  const bool resume = blk && b0 > 0, last_launch = !blk || b1 == nblk;
  double* const pk = park ? park + (size_t)cell * kPark : nullptr;
  const int ti1 = tid >> 1, n1 = tid & 1;
  constexpr bool five = FIVE;
  double wA5[5], wR5[5];
  {
#pragma unroll
    for (int q = 0; q < 5; ++q) {                  // alpha 0.5: p = 0.25 (l + m), slot l + m; alpha 0: p = 0.5 l, slots 0..2 (3, 4 repeat 2)
      const int l = n1 ? (q > 2 ? 2 : q) : min(q, 2), m = n1 ? q - l : 0;
      const double p = 0.5 * l + (m - l) * 0.5 * (n1 ? 0.5 : 0.0);
      wA5[q] = p;
      wR5[q] = 1.0 - p;
    }
  }
  bool ok = resume ? pk[10] != 0.0 : true;
  // Lanes 0..3 own the four chains: (a,b) low, (a,b) high, (b,a) low, (b,a) high = the accumulator had every ambiguous log() come
  // out low / high.  While low == high an accumulator is known.  A step that splits them is an EVENT (its log() argument and
  // lower candidate are kept); from there on each of the two paths has to stay unambiguous by itself (clean): then the
  // reference's value is the low path if its libm returned the lower candidate at the event and the high path otherwise,
  // whatever it returned elsewhere.  The chains are serial (one add per pair); the event / clean tests of a tile's 32 steps
  // run lane-parallel on the chains' prefixes.
  double acc = (resume && tid < 4) ? pk[tid] : 0.0;
  bool clean[2] = {resume ? pk[8] != 0.0 : true, resume ? pk[9] != 0.0 : true};
  if (tid < 4) s_ev[cw][tid] = resume ? pk[4 + tid] : 0.0;   // [ab: log argument, lower candidate | ba: ...] of the open event
  const size_t S = (size_t)pv.S;
  // lane n1 = 0 accumulates llksAB[a][b], lane 1 llksAB[b][a]: the lane's FIRST sample (rows l of :675-681) is a resp. b, its second b resp. a —
  // chosen here, once, by the column pointers instead of per product by selects
  const float* const colA = DENSE ? gT + (size_t)((n1 ? ib : ia) * 3) * S : g + (size_t)(n1 ? ib : ia) * 3;
  const float* const colB = DENSE ? gT + (size_t)((n1 ? ia : ib) * 3) * S : g + (size_t)(n1 ? ia : ib) * 3;
  const size_t estride = DENSE ? S : 1, sstride = DENSE ? 1 : (size_t)V * 3;
  uint32_t hd_n = 0u; int32_t hd_s = 0;          // header (stored reads, SNP id) of this lane's pair in the NEXT tile
  if (tid < TP && tid < np) { hd_n = load_nrd(pv.pair_nrd, p_beg + tid, nrd_width); hd_s = pv.pair_snp ? pv.pair_snp[p_beg + tid] : (int32_t)tid; }
  for (int64_t tbase = 0; tbase < np; tbase += TP) {
    const int tp = (int)min((int64_t)TP, np - tbase);
    if (tid < TP) {
      const uint32_t n = hd_n;                     // this tile's header was requested a tile ago; now request the next one's
      const int32_t sn = hd_s;
      const int64_t nx = tbase + TP + tid;
      hd_n = 0u; hd_s = 0;
      if (nx < np) { hd_n = load_nrd(pv.pair_nrd, p_beg + nx, nrd_width); hd_s = pv.pair_snp ? pv.pair_snp[p_beg + nx] : (int32_t)nx; }
      const uint32_t incl = seg_scan_incl<32>(n);
      s_cnt[tid] = n;
      s_off[tid] = rd_base + (int64_t)(incl - n);
      s_snp[tid] = sn;
    }
    DMX_WAVE_LDS_ORDER();
    rd_base = s_off[tp - 1] + (int64_t)s_cnt[tp - 1];
    {
      const bool on = ti1 < tp;
      const uint32_t cnt = on ? s_cnt[ti1] : 0u;
      const int64_t off = on ? s_off[ti1] : 0;
      const uint32_t rd4 = load_rd4(pv, off, cnt);       // the first four read bytes in one load (one dependent latency instead of four)
      // The two samples' probabilities at the pair's SNP.  Dense pileups read them from the SNP-minor copy gT[row element][snp]: the
      // 32 pairs of a tile are 32 consecutive SNPs, i.e. one 128-byte line per element, instead of two lines per PAIR out of
      // V*12-byte rows (165 GB through the L2 per launch at the cfg4 shard, profiles/r02_cfg4_fast_pmc_summary.json).  Sparse
      // pileups read the row of the lane's own SNP.
      const size_t so = (size_t)(on ? s_snp[ti1] : 0) * sstride;
      const float fa0 = colA[so], fa1 = colA[so + estride], fa2 = colA[so + 2 * estride];
      const float fb0 = colB[so], fb1 = colB[so + estride], fb2 = colB[so + 2 * estride];
      double v[9];                                 // pG[1][l][m] of the pair (five: v[l + m])
      if constexpr (five) {
        double v5[5];
        // round 6: pairs of up to three tabled reads take their FINAL values from k_build_certify_finals' table (both lanes of a pair the same entry);
        // the walk below runs only in tiles with a deeper pair (wave-uniform branch)
        const int32_t fi = cfin ? certify_final_index(cnt, rd4) : -1;
        if (!__any(fi < 0)) {
          const double* fp = cfin + (size_t)fi * kCFinStride;
          const double2 a = *reinterpret_cast<const double2*>(fp), b = *reinterpret_cast<const double2*>(fp + 2);
          v5[0] = a.x; v5[1] = a.y; v5[2] = b.x; v5[3] = b.y; v5[4] = fp[4];
        } else {                                   // (a tile with a deeper pair: the walk for ALL its lanes, as before)
          certify_pair_values<5>(pv, cnt, off, rd4, s_tab, wA5, wR5, n1, v5, cseed);
        }
#pragma unroll
        for (int l = 0; l < 3; ++l)
#pragma unroll
          for (int m = 0; m < 3; ++m) v[l * 3 + m] = v5[l + m];
      } else {                                     // other alpha[0]: the nine-value form, weights formed here (:613)
        double wA[9], wR[9];
        const double al = alpha[n1];
#pragma unroll
        for (int l = 0; l < 3; ++l)
#pragma unroll
          for (int m = 0; m < 3; ++m) {
            const double p = 0.5 * l + (m - l) * 0.5 * al;
            wA[l * 3 + m] = p;
            wR[l * 3 + m] = 1.0 - p;
          }
        certify_pair_values<9>(pv, cnt, off, rd4, s_tab, wA, wR, n1, v);
      }
      if (on) {
        const double aj[3] = {(double)fa0, (double)fa1, (double)fa2}, bk[3] = {(double)fb0, (double)fb1, (double)fb2};
        double sx = 0.0;                                                     // :674
#pragma unroll
        for (int l = 0; l < 3; ++l)
#pragma unroll
          for (int m = 0; m < 3; ++m) {
            sx += ((aj[l] * bk[m]) * v[l * 3 + m]);                          // :553, :677-679 (aj = this lane's first sample, see colA)
          }
        ok &= __builtin_amdgcn_class(sx, 0x100);
        double lo1, hi1;
        dmx_log_bracket((uint32_t)__double2hiint(sx), (uint32_t)__double2loint(sx), s_log, s_lo, &lo1, &hi1);
        s_t[cw][2 * n1][ti1] = lo1; s_t[cw][2 * n1 + 1][ti1] = hi1; s_t[cw][4 + n1][ti1] = sx;
      }
    }
    DMX_WAVE_LDS_ORDER();
    if (tid < 4) {
      const double* row = &s_t[cw][tid][0];
      double* pre = &s_p[cw][tid][0];
      double a = acc;
      for (int i = 0; i < tp; ++i) { pre[i] = a; a += row[i]; }              // :683, in pair order
      pre[tp] = a;
      acc = a;
    }
    DMX_WAVE_LDS_ORDER();
    {
      const int i = tid & 31, w = tid >> 5;                                  // step i of accumulator w
      bool ev = false, dirty = false;
      if (i < tp) {
        const double lo = s_p[cw][2 * w][i], hi = s_p[cw][2 * w + 1][i], ll = s_p[cw][2 * w][i + 1], hh = s_p[cw][2 * w + 1][i + 1];
        const double tl = s_t[cw][2 * w][i], th = s_t[cw][2 * w + 1][i];
        const double lh = lo + th, hl = hi + tl;
        const bool closed = lo == hi;
        ev = closed && (ll != lh);
        dirty = !closed && !((ll == lh) && (hl == hh));
      }
      const uint64_t evm = __ballot(ev), dm = __ballot(dirty);
#pragma unroll
      for (int w2 = 0; w2 < 2; ++w2) {
        const uint32_t e32 = (uint32_t)(evm >> (32 * w2)), d32 = (uint32_t)(dm >> (32 * w2));
        if (e32) {
          const int e = 31 - __clz((int)e32);                                // the tile's last event of this accumulator
          if (tid == 0) { s_ev[cw][2 * w2] = s_t[cw][4 + w2][e]; s_ev[cw][2 * w2 + 1] = s_t[cw][2 * w2][e]; }
          clean[w2] = (d32 & ~((2u << e) - 1u)) == 0u;                       // no dirty step after it
        } else {
          clean[w2] = clean[w2] && d32 == 0u;
        }
      }
    }
    DMX_WAVE_LDS_ORDER();
  }
  const bool all_ok = __all(ok ? 1 : 0) != 0;
  if (!last_launch) {                             // park the barcode's state for the next SNP block
    DMX_WAVE_LDS_ORDER();
    if (tid < 4) { pk[tid] = acc; pk[4 + tid] = s_ev[cw][tid]; }
    if (tid == 0) { pk[8] = clean[0] ? 1.0 : 0.0; pk[9] = clean[1] ? 1.0 : 0.0; pk[10] = all_ok ? 1.0 : 0.0; }
    return;
  }
  const double ab_lo = __shfl(acc, 0), ab_hi = __shfl(acc, 1), ba_lo = __shfl(acc, 2), ba_hi = __shfl(acc, 3);
  if (tid == 0 && all_ok && !(ab_lo == ab_hi && ba_lo == ba_hi) && (ab_lo == ab_hi || clean[0]) && (ba_lo == ba_hi || clean[1])) {
    dmx_cell_summary* r = summ + cell;
    r->llk_ab = ab_lo; r->llk_ab_alt = ab_hi; r->llk_ba = ba_lo; r->llk_ba_alt = ba_hi;
    r->ev_x_ab = s_ev[cw][0]; r->ev_t_ab = s_ev[cw][1]; r->ev_x_ba = s_ev[cw][2]; r->ev_t_ba = s_ev[cw][3];
    r->flags = sm_fl | DMX_CELL_ORDER_RESOLVABLE;
  }
  if (tid == 0 && all_ok && ab_lo == ab_hi && ba_lo == ba_hi) {
    // the reference's scan (:799-814, strict <) meets (a,b) before (b,a): it keeps (a,b) unless (b,a) is strictly larger
    const bool ba = ab_lo < ba_lo;
    dmx_cell_summary* r = summ + cell;
    const int32_t nj = ba ? ib : ia, nk = ba ? ia : ib;
    if (nj != sm_j) {
      const double l1 = r->llk1, l2 = r->llk2, l10 = r->llk10, l20 = r->llk20;
      r->llk1 = l2; r->llk2 = l1; r->llk10 = l20; r->llk20 = l10;
    }
    r->j_best = nj; r->k_best = nk;
    r->llk12 = ba ? ba_lo : ab_lo;
    r->llk_ab = ab_lo; r->llk_ba = ba_lo; r->llk_ab_alt = ab_lo; r->llk_ba_alt = ba_lo;
    r->flags = sm_fl | DMX_CELL_ORDER_CERTIFIED;
  }
}

}  // namespace

// =====================================================================================================================
struct dmx_engine {
  int32_t V = 0, A = 0, device = 0, mode = 0;
  bool certify = false;        // run k_certify (the host libm's log() was found inside dmx_log_bracket's brackets)
  double prior = 0.5;
  std::vector<double> alpha;
  hipStream_t own_stream = nullptr, stream = nullptr;
  hipStream_t k1_stream = nullptr;               // dmx_engine_run: K1 beside K2 (lowest priority: it fills what K2's last round leaves free)
  hipEvent_t ev_fork = nullptr, ev_k1_done = nullptr;
  double* d_lut = nullptr;
  double* d_alpha = nullptr;
  // genotypes
  const float* d_g = nullptr; float* d_g_own = nullptr; int32_t S = 0; double* d_gp0 = nullptr; float* d_gT = nullptr; double* d_g0T = nullptr;
  float* d_rows = nullptr; uint8_t* d_ids = nullptr; uint32_t* d_idw = nullptr; uint32_t* d_idd = nullptr; int32_t nwd2 = 0; int32_t n_classes = 0;   // genotype classes (0 = not usable)
  // pileup
  PileupView pv{}; int32_t nrd_width = 1; int64_t P = 0, R = 0; bool have_pileup = false;
  void* own[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // device copies of a host pileup (grow-only: a job's ranges reuse them); [5]: gather sources
  size_t own_cap[6] = {0, 0, 0, 0, 0, 0};
  int32_t* d_sched = nullptr; size_t sched_cap = 0;
  int64_t max_cell_pairs = 0;                                         // covered SNPs of the staged pileup's longest barcode
  int32_t* d_bad = nullptr;                                          // set by k_check_snp_ids
  bool have_gT = false;                                              // d_gT / d_g0T hold the current genotype matrix
  // canonical GT classes (round 4): every called genotype of a --field GT matrix is one of three SNP-independent rows (hi, lo, lo) permuted
  // (bcf_filtered_reader.cpp:397-400); class ids are then 0 / 1 / 2 = that row with hi in place 0 / 1 / 2 and 3 = the SNP's one other row (a missing
  // genotype's HWE row, :381-388), and K1 takes log(GL . row) of the three canonical rows from a table indexed like the GL tables (d_ltab)
  bool canon = false, ltab_valid = false; float can_hi = 0.f, can_lo = 0.f; double* d_ltab = nullptr; uint8_t* d_oth = nullptr;
  uint4* d_snprec = nullptr; bool snprec_valid = false; double* d_ctab = nullptr; bool ctab_valid = false;   // k_singlet_can's per-SNP records and merged GL / class-log table
  uint8_t* d_clsb = nullptr; bool clsb_valid = false; uint32_t clsb_stride = 0;                             // k_singlet_canp's per-sample class byte streams (k_build_clsb)
  bool any_oth = false;                                   // some SNP of the canonical-class matrix has a fourth row (k_canon_apply's oth)
  bool off32 = false;                                     // every absolute pair index and read offset of the staged pileup fits 32 bits
  bool reads_padded = false;                              // four bytes past the staged pileup's last read byte are readable (k_singlet_can's unconditional 4-byte loads)
  double* d_cseed = nullptr; bool cseed_valid = false;   // certify_pair_values' seeds (k_build_certify_seeds; a function of the phred tables)
  double* d_cfin = nullptr; bool cfin_valid = false;     // ... and its final values for pairs of up to three tabled reads (k_build_certify_finals, round 6)
  double* d_park = nullptr; size_t park_cap = 0;   // k_certify's per-barcode state between the launches of its SNP-blocked walk
  int64_t* d_blk = nullptr; size_t blk_cap = 0; int32_t blk_shift = 0, blk_n = 0;   // k_snp_blocks table of the staged (sparse) pileup; blk_n = 0: none
  bool geno_safe = false;                                            // every genotype row finite, non-negative, max >= 2^-400 (k_check_geno)
  // host -> device staging of the big pileup arrays: two pinned chunks filled by host threads while the other one is in flight
  void* h_stage[2] = {nullptr, nullptr}; hipEvent_t ev_stage[2] = {nullptr, nullptr}; bool stage_busy[2] = {false, false}; int stage_cur = 0;
  // results
  double *d_llks = nullptr, *d_llk0s = nullptr, *d_grid = nullptr, *d_l00 = nullptr;
  dmx_cell_summary* d_sum = nullptr;
  uint8_t* d_flag = nullptr;
  double* d_sing = nullptr;
  int32_t out_cap = 0, grid_cap = 0; bool have_grid = false, have_sing = false;   // cells the result buffers hold
  hipEvent_t ev[8] = {};
  // the last kRing launches of K1 and of K2 / K3 / K3b, bracketed by events on the engine's stream (dmx_engine_mean_kernel_times)
  static constexpr int kRing = 16;
  hipEvent_t ring_s[kRing][2] = {}, ring_d[kRing][4] = {};
  int64_t n_ring_s = 0, n_ring_d = 0; bool ring_certified[kRing] = {};
  bool timed[4] = {false, false, false, false};
  // Experiment switches.  The DMX_* variables that steer kernel selection — kernel experiments, the tests' forced
  // launch geometries — are copied ONCE, at dmx_engine_create, and only when DMX_EXPERIMENTS=1 is in the environment: a stray DMX_*
  // variable in a user's shell never changes which kernel runs, and no launch reads the environment.
  std::unordered_map<std::string, std::string> knobs;
  const char* knob(const char* name) const { if (knobs.empty()) return nullptr; auto it = knobs.find(name); return it == knobs.end() ? nullptr : it->second.c_str(); }
  // what the last run launched (dmx_engine_kernel_names): host function pointers of the K1 / K2 / K3b kernels, and where K1 ran
  const void *k1_fn = nullptr, *k2_fn = nullptr, *k3b_fn = nullptr; int32_t k1_placement = 0;
  bool k2_sym = false;                                               // launch_doublet picked a k_doublet_sym form (FAST, soft fields, grid {0, 0.5})
};

namespace {

int free_pileup(dmx_engine* e) {
  for (int i = 0; i < 6; ++i) { if (e->own[i]) (void)hipFree(e->own[i]); e->own[i] = nullptr; e->own_cap[i] = 0; }
  if (e->d_sched) (void)hipFree(e->d_sched);
  e->d_sched = nullptr; e->sched_cap = 0;
  e->have_pileup = false;
  return DMX_OK;
}
// Device buffer of at least `bytes`, kept across calls.  hipFree waits for the whole device, so a job whose ranges alternate
// between two engines must not free and allocate per range: buffers only grow (with some slack for the next, slightly larger range).
int ensure_dev(void** p, size_t* cap, size_t bytes) {
  bytes = std::max<size_t>(bytes, 16);
  if (*p && *cap >= bytes) return DMX_OK;
  if (*p) { (void)hipFree(*p); *p = nullptr; *cap = 0; }
  const size_t want = bytes + bytes / 16;
  HIP_TRY(hipMalloc(p, want));
  *cap = want;
  return DMX_OK;
}
int free_results(dmx_engine* e) {
  if (e->d_llks) (void)hipFree(e->d_llks);
  if (e->d_llk0s) (void)hipFree(e->d_llk0s);
  if (e->d_grid) (void)hipFree(e->d_grid);
  if (e->d_l00) (void)hipFree(e->d_l00);
  if (e->d_sum) (void)hipFree(e->d_sum);
  if (e->d_flag) (void)hipFree(e->d_flag - kFlagHead);
  if (e->d_sing) (void)hipFree(e->d_sing);
  e->d_flag = nullptr; e->d_sing = nullptr;
  e->d_llks = e->d_llk0s = e->d_grid = e->d_l00 = nullptr; e->d_sum = nullptr;
  e->out_cap = 0; e->grid_cap = 0; e->have_grid = e->have_sing = false;
  return DMX_OK;
}

template <typename T>
int upload(dmx_engine* e, const void* host, size_t count, void** slot, const T** view) {
  void* d = nullptr;
  const size_t bytes = std::max<size_t>(count * sizeof(T), 16);
  HIP_TRY(hipMalloc(&d, bytes));
  *slot = d;
  if (count) HIP_TRY(hipMemcpyAsync(d, host, count * sizeof(T), hipMemcpyHostToDevice, e->stream));
  *view = (const T*)d;
  return DMX_OK;
}

}  // namespace

namespace { __global__ void k_noop() {} }

extern "C" int dmx_device_warm_up(int32_t device, int32_t n_gpus) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return set_error(DMX_ERR_NOGPU, "dmx_device_warm_up: no HIP device is visible (this library has no CPU fallback)");
  if (device < 0 || device >= ndev) return set_error(DMX_ERR_ARG, "dmx_device_warm_up: device %d of %d", device, ndev);
  for (int i = 0; i < std::max(1, std::min(n_gpus, ndev)); ++i) {
    HIP_TRY(hipSetDevice((device + i) % ndev));
    HIP_TRY(hipFree(nullptr));                   // (creates the primary context)
    hipLaunchKernelGGL(k_noop, dim3(1), dim3(64), 0, nullptr);     // (loads this library's code object onto the device)
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
  }
  return DMX_OK;
}

extern "C" int dmx_engine_create(const dmx_engine_config* cfg, dmx_engine** out) {
  if (!cfg || !out) return set_error(DMX_ERR_ARG, "dmx_engine_create: null argument");
  if (cfg->n_samples < 1 || cfg->n_samples > 4094) return set_error(DMX_ERR_ARG, "dmx_engine_create: n_samples %d not in [1,4094]", cfg->n_samples);
  if (cfg->n_alpha < 1 || cfg->n_alpha > 64 || !cfg->alpha) return set_error(DMX_ERR_ARG, "dmx_engine_create: n_alpha %d not in [1,64] or null grid", cfg->n_alpha);
  if (cfg->mode != DMX_MODE_STRICT && cfg->mode != DMX_MODE_FAST) return set_error(DMX_ERR_ARG, "dmx_engine_create: unknown mode %d", cfg->mode);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return set_error(DMX_ERR_NOGPU, "dmx_engine_create: no HIP device is visible (this library has no CPU fallback)");
  if (cfg->device < 0 || cfg->device >= ndev) return set_error(DMX_ERR_ARG, "dmx_engine_create: device %d of %d", cfg->device, ndev);
  HIP_TRY(hipSetDevice(cfg->device));
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, cfg->device));
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return set_error(DMX_ERR_NOGPU, "dmx_engine_create: device %d is %s; this library is built for gfx950 (MI355X) only", cfg->device, prop.gcnArchName);
  dmx_engine* e = new (std::nothrow) dmx_engine;
  if (!e) return set_error(DMX_ERR_NOMEM, "dmx_engine_create: out of memory");
  e->V = cfg->n_samples; e->A = cfg->n_alpha; e->device = cfg->device; e->mode = cfg->mode; e->prior = cfg->doublet_prior;
  e->alpha.assign(cfg->alpha, cfg->alpha + cfg->n_alpha);
  if (const char* x = getenv("DMX_EXPERIMENTS")) if (x[0] == '1' && !x[1])
    for (char** ev = environ; ev && *ev; ++ev)
      if (!std::strncmp(*ev, "DMX_", 4)) if (const char* eq = std::strchr(*ev, '=')) e->knobs.emplace(std::string((const char*)*ev, (size_t)(eq - *ev)), std::string(eq + 1));
  HIP_TRY(hipStreamCreateWithFlags(&e->own_stream, hipStreamNonBlocking));
  e->stream = e->own_stream;
  {
    int least = 0, greatest = 0;
    HIP_TRY(hipDeviceGetStreamPriorityRange(&least, &greatest));
    HIP_TRY(hipStreamCreateWithPriority(&e->k1_stream, hipStreamNonBlocking, least));
    HIP_TRY(hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&e->ev_k1_done, hipEventDisableTiming));
  }
  for (hipEvent_t& ev : e->ev) HIP_TRY(hipEventCreate(&ev));
  for (auto& r : e->ring_s) for (hipEvent_t& ev : r) HIP_TRY(hipEventCreate(&ev));
  for (auto& r : e->ring_d) for (hipEvent_t& ev : r) HIP_TRY(hipEventCreate(&ev));
  HIP_TRY(hipMalloc((void**)&e->d_lut, sizeof(double) * kTabTotal));
  HIP_TRY(hipMemcpy(e->d_lut + kLut, dmx_log_table_host, sizeof(double) * DMX_LOG_TABLE_DOUBLES, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(e->d_lut + kTabLogLo, dmx_log_table_lo_host, sizeof(double) * 128, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(e->d_lut + kTabLog2, dmx_log2_table_host, sizeof(double) * DMX_LOG2_TABLE_DOUBLES, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(e->d_lut + kTabLog32, dmx_log32_table_host, sizeof(double) * DMX_LOG32_TABLE_DOUBLES, hipMemcpyHostToDevice));
  e->certify = !(cfg->flags & DMX_ENGINE_NO_CERTIFY) && !e->knob("DMX_NO_CERTIFY") && dmx::libm_log_within_brackets();
  HIP_TRY(hipMalloc((void**)&e->d_alpha, sizeof(double) * 64));
  HIP_TRY(hipMemcpy(e->d_alpha, e->alpha.data(), sizeof(double) * e->A, hipMemcpyHostToDevice));
  double mat[256], err[256];
  dmx_phred_tables(mat, err);
  *out = e;
  return dmx_engine_set_phred_tables(e, mat, err);
}

extern "C" int dmx_engine_destroy(dmx_engine* e) {
  if (!e) return DMX_OK;
  (void)hipSetDevice(e->device);
  (void)hipStreamSynchronize(e->stream);
  free_pileup(e); free_results(e);
  if (e->d_g_own) (void)hipFree(e->d_g_own);
  if (e->d_gp0) (void)hipFree(e->d_gp0);
  if (e->d_gT) (void)hipFree(e->d_gT);
  if (e->d_g0T) (void)hipFree(e->d_g0T);
  if (e->d_rows) (void)hipFree(e->d_rows);
  if (e->d_ids) (void)hipFree(e->d_ids);
  if (e->d_idw) (void)hipFree(e->d_idw);
  if (e->d_idd) (void)hipFree(e->d_idd);
  if (e->d_lut) (void)hipFree(e->d_lut);
  if (e->d_alpha) (void)hipFree(e->d_alpha);
  if (e->d_bad) (void)hipFree(e->d_bad);
  if (e->d_blk) (void)hipFree(e->d_blk);
  if (e->d_park) (void)hipFree(e->d_park);
  if (e->d_cseed) (void)hipFree(e->d_cseed);
  if (e->d_cfin) (void)hipFree(e->d_cfin);
  if (e->d_clsb) (void)hipFree(e->d_clsb);
  if (e->d_ltab) (void)hipFree(e->d_ltab);
  if (e->d_oth) (void)hipFree(e->d_oth);
  if (e->d_snprec) (void)hipFree(e->d_snprec);
  if (e->d_ctab) (void)hipFree(e->d_ctab);
  for (int i = 0; i < 2; ++i) { if (e->h_stage[i]) (void)hipHostFree(e->h_stage[i]); if (e->ev_stage[i]) (void)hipEventDestroy(e->ev_stage[i]); }
  for (hipEvent_t& ev : e->ev) if (ev) (void)hipEventDestroy(ev);
  for (auto& r : e->ring_s) for (hipEvent_t& ev : r) if (ev) (void)hipEventDestroy(ev);
  for (auto& r : e->ring_d) for (hipEvent_t& ev : r) if (ev) (void)hipEventDestroy(ev);
  if (e->own_stream) (void)hipStreamDestroy(e->own_stream);
  if (e->k1_stream) (void)hipStreamDestroy(e->k1_stream);
  if (e->ev_fork) (void)hipEventDestroy(e->ev_fork);
  if (e->ev_k1_done) (void)hipEventDestroy(e->ev_k1_done);
  delete e;
  return DMX_OK;
}

extern "C" int dmx_engine_set_stream(dmx_engine* e, void* s) {
  if (!e) return set_error(DMX_ERR_ARG, "dmx_engine_set_stream: null engine");
  e->stream = s ? (hipStream_t)s : e->own_stream;
  return DMX_OK;
}

extern "C" int dmx_engine_set_phred_tables(dmx_engine* e, const double mat[256], const double err[256]) {
  if (!e || !mat || !err) return set_error(DMX_ERR_ARG, "dmx_engine_set_phred_tables: null argument");
  HIP_TRY(hipSetDevice(e->device));
  dmx::ReadLut lut;
  dmx::build_read_lut(mat, err, &lut);
  HIP_TRY(hipMemcpy(e->d_lut, &lut, sizeof(double) * kLut, hipMemcpyHostToDevice));
  static_assert(sizeof(dmx::SingletTables) == sizeof(double) * (kFirst + kFinal), "table layout");
  dmx::SingletTables st;
  dmx::build_singlet_tables(lut, &st);
  HIP_TRY(hipMemcpy(e->d_lut + kTab, &st, sizeof st, hipMemcpyHostToDevice));
  static_assert(sizeof(dmx::PairTables) == sizeof(double) * 2 * kPair, "table layout");
  std::unique_ptr<dmx::PairTables> pt(new dmx::PairTables);
  dmx::build_pair_tables(lut, st, pt.get());
  HIP_TRY(hipMemcpy(e->d_lut + kTabK1, pt.get(), sizeof(dmx::PairTables), hipMemcpyHostToDevice));
  const dmx::TripleTables& tt = dmx::build_triple_tables(lut, *pt);
  HIP_TRY(hipMemcpy(e->d_lut + kTabK1 + 2 * kPair, tt.third.data(), sizeof(double) * kTriple, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(e->d_lut + kTabK1 + 2 * kPair + kTriple, tt.final3.data(), sizeof(double) * kTriple, hipMemcpyHostToDevice));
  e->ltab_valid = false; e->cseed_valid = false; e->cfin_valid = false; e->ctab_valid = false;   // (the canonical-class log tables and k_certify's seeds are functions of these tables)
  return DMX_OK;
}

extern "C" int dmx_engine_set_genotypes(dmx_engine* e, const float* g, int32_t n_snps, int32_t memory) {
  if (!e || !g || n_snps < 0) return set_error(DMX_ERR_ARG, "dmx_engine_set_genotypes: bad arguments");
  HIP_TRY(hipSetDevice(e->device));
  if (e->d_g_own) { (void)hipFree(e->d_g_own); e->d_g_own = nullptr; }
  if (e->d_gp0) { (void)hipFree(e->d_gp0); e->d_gp0 = nullptr; }
  e->have_gT = false;
  const size_t n = (size_t)n_snps * e->V * 3;
  if (memory == DMX_MEM_DEVICE) {
    e->d_g = g;
  } else {
    HIP_TRY(hipMalloc((void**)&e->d_g_own, std::max<size_t>(n * sizeof(float), 16)));
    HIP_TRY(hipMemcpyAsync(e->d_g_own, g, n * sizeof(float), hipMemcpyHostToDevice, e->stream));
    e->d_g = e->d_g_own;
  }
  e->S = n_snps;
  HIP_TRY(hipMalloc((void**)&e->d_gp0, std::max<size_t>((size_t)n_snps * 3 * sizeof(double), 16)));
  if (n_snps > 0) {
    const int64_t items = (int64_t)n_snps * 3;
    HIP_TRY(hipEventRecord(e->ev[0], e->stream));
    hipLaunchKernelGGL(k_gp0, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, e->stream, e->d_g, n_snps, e->V, e->d_gp0);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(e->ev[1], e->stream));
    e->timed[0] = true;
  }

  // genotype classes (<= 4 distinct rows per SNP: --field GT): enables the class kernels
  if (e->d_rows) { (void)hipFree(e->d_rows); e->d_rows = nullptr; }
  if (e->d_ids) { (void)hipFree(e->d_ids); e->d_ids = nullptr; }
  if (e->d_idw) { (void)hipFree(e->d_idw); e->d_idw = nullptr; }
  if (e->d_idd) { (void)hipFree(e->d_idd); e->d_idd = nullptr; }
  e->n_classes = 0;
  e->geno_safe = false;
  if (n_snps > 0) {
    int32_t* d_max = nullptr;
    HIP_TRY(hipMalloc((void**)&e->d_rows, (size_t)n_snps * kMaxCls * 3 * sizeof(float)));
    HIP_TRY(hipMalloc((void**)&e->d_ids, (size_t)n_snps * e->V + 16));
    HIP_TRY(hipMalloc((void**)&e->d_idw, (size_t)n_snps * ((e->V + 15) / 16) * sizeof(uint32_t) + 16));
    e->nwd2 = ((e->V + e->V / 2 - 1) >> 4) + 5;                     // last window start (V-1 + V/2) >> 4, plus a window of <= 4 words, plus one
    HIP_TRY(hipMalloc((void**)&e->d_idd, (size_t)n_snps * e->nwd2 * sizeof(uint32_t) + 16));
    HIP_TRY(hipMalloc((void**)&d_max, sizeof(int32_t)));
    HIP_TRY(hipMemsetAsync(d_max, 0, sizeof(int32_t), e->stream));
    hipLaunchKernelGGL(k_build_classes, dim3((unsigned)((n_snps + 255) / 256)), dim3(256), 0, e->stream, e->d_g, n_snps, e->V,
                       e->d_rows, e->d_ids, e->d_idw, e->d_idd, e->nwd2, d_max);
    HIP_TRY(hipGetLastError());
    int32_t* d_unsafe = nullptr;
    HIP_TRY(hipMalloc((void**)&d_unsafe, sizeof(int32_t)));
    HIP_TRY(hipMemsetAsync(d_unsafe, 0, sizeof(int32_t), e->stream));
    hipLaunchKernelGGL(k_check_geno, dim3(1024), dim3(256), 0, e->stream, e->d_g, (int64_t)n_snps * e->V, d_unsafe);
    HIP_TRY(hipGetLastError());
    int32_t h_max = 0, h_unsafe = 1;
    HIP_TRY(hipMemcpyAsync(&h_max, d_max, sizeof(int32_t), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipMemcpyAsync(&h_unsafe, d_unsafe, sizeof(int32_t), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    (void)hipFree(d_max); (void)hipFree(d_unsafe);
    e->geno_safe = h_unsafe == 0 && !e->knob("DMX_FORCE_CHECK");
    e->n_classes = (h_max >= 1 && h_max <= kMaxCls) ? h_max : 0;
    if (!e->n_classes) { (void)hipFree(e->d_rows); (void)hipFree(e->d_ids); (void)hipFree(e->d_idw); (void)hipFree(e->d_idd); e->d_rows = nullptr; e->d_ids = nullptr; e->d_idw = nullptr; e->d_idd = nullptr; }
    // canonical GT classes (see k_canon_apply): relabel when the matrix has that shape
    e->canon = false; e->ltab_valid = false; e->ctab_valid = false; e->snprec_valid = false; e->clsb_valid = false;
    if (e->d_oth) { (void)hipFree(e->d_oth); e->d_oth = nullptr; }
    if (e->n_classes && !e->knob("DMX_NO_CANON")) {
      int32_t* d_w = nullptr;
      HIP_TRY(hipMalloc((void**)&d_w, 3 * sizeof(int32_t)));
      int32_t h_w[3] = {0x7FFFFFFF, 0, 0};
      HIP_TRY(hipMemcpyAsync(d_w, h_w, sizeof h_w, hipMemcpyHostToDevice, e->stream));
      hipLaunchKernelGGL(k_find_canon, dim3((unsigned)((n_snps + 255) / 256)), dim3(256), 0, e->stream, e->d_rows, n_snps, d_w);
      HIP_TRY(hipGetLastError());
      HIP_TRY(hipMemcpyAsync(h_w, d_w, sizeof(int32_t), hipMemcpyDeviceToHost, e->stream));
      HIP_TRY(hipStreamSynchronize(e->stream));
      if (h_w[0] < n_snps) {
        float r[kMaxCls * 3];
        HIP_TRY(hipMemcpy(r, e->d_rows + (size_t)h_w[0] * kMaxCls * 3, sizeof r, hipMemcpyDeviceToHost));
        float hi = 0.f, lo = 0.f; bool found = false;
        for (int d = 0; d < kMaxCls && !found; ++d) {
          const float a = r[d * 3], b = r[d * 3 + 1], c = r[d * 3 + 2];
          uint32_t ua, ub, uc; std::memcpy(&ua, &a, 4); std::memcpy(&ub, &b, 4); std::memcpy(&uc, &c, 4);
          if (!(a >= 0.f && b >= 0.f && c >= 0.f) || !(a < 3e38f && b < 3e38f && c < 3e38f)) continue;
          if (ub == uc && a > b) { hi = a; lo = b; found = true; }
          else if (ua == uc && b > a) { hi = b; lo = a; found = true; }
          else if (ua == ub && c > a) { hi = c; lo = a; found = true; }
        }
        if (found) {
          uint32_t uh, ul; std::memcpy(&uh, &hi, 4); std::memcpy(&ul, &lo, 4);
          hipLaunchKernelGGL(k_canon_check, dim3((unsigned)((n_snps + 255) / 256)), dim3(256), 0, e->stream, e->d_rows, n_snps, uh, ul, d_w + 1);
          HIP_TRY(hipGetLastError());
          HIP_TRY(hipMemcpyAsync(h_w + 1, d_w + 1, sizeof(int32_t), hipMemcpyDeviceToHost, e->stream));
          HIP_TRY(hipStreamSynchronize(e->stream));
          if (h_w[1] == 0) {
            HIP_TRY(hipMalloc((void**)&e->d_oth, (size_t)n_snps + 16));
            hipLaunchKernelGGL(k_canon_apply, dim3((unsigned)((n_snps + 255) / 256)), dim3(256), 0, e->stream, e->d_rows, e->d_ids, e->d_idw,
                               e->d_idd, n_snps, e->V, e->nwd2, uh, ul, e->d_oth, d_w + 2);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipMemcpyAsync(h_w + 2, d_w + 2, sizeof(int32_t), hipMemcpyDeviceToHost, e->stream));
            HIP_TRY(hipStreamSynchronize(e->stream));
            e->canon = true; e->can_hi = hi; e->can_lo = lo; e->any_oth = h_w[2] != 0;
          }
        }
      }
      (void)hipFree(d_w);
    }
  }
  HIP_TRY(hipStreamSynchronize(e->stream));   // the host buffer may go away after return
  return DMX_OK;
}

namespace {

__global__ void k_check_snp_ids(const int32_t* __restrict__ snp, int64_t n, int32_t S, int32_t* __restrict__ bad) {
  bool b = false;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) b |= (uint32_t)snp[i] >= (uint32_t)S;
  if (b) atomicOr(bad, 1);
}

constexpr size_t kStageBytes = (size_t)16 << 20;

void par_memcpy(uint8_t* dst, const uint8_t* src, size_t n) {
  constexpr size_t kPart = (size_t)2 << 20;
  if (n < 2 * kPart) { std::memcpy(dst, src, n); return; }
  const int T = (int)std::min<size_t>(4, n / kPart);
  std::vector<std::thread> th;
  for (int t = 1; t < T; ++t) th.emplace_back([=] { std::memcpy(dst + n * t / T, src + n * t / T, n * (t + 1) / T - n * t / T); });
  std::memcpy(dst, src, n / T);
  for (std::thread& x : th) x.join();
}

// Streams host bytes into one device array through the engine's two pinned chunks: the host fills one while the copy engine
// drains the other.  (hipMemcpy from pageable memory measured 4 GB/s on this platform; this path is bound by the host memcpy.)
struct StageWriter {
  dmx_engine* e;
  uint8_t* dst;
  size_t fill = 0;
  int slot = 0;
  StageWriter(dmx_engine* e_, void* dst_) : e(e_), dst((uint8_t*)dst_) {}
  int put(const void* src_, size_t n) {
    const uint8_t* src = (const uint8_t*)src_;
    while (n) {
      if (fill == 0) {
        slot = e->stage_cur;
        if (e->stage_busy[slot]) { HIP_TRY(hipEventSynchronize(e->ev_stage[slot])); e->stage_busy[slot] = false; }
      }
      const size_t m = std::min(n, kStageBytes - fill);
      par_memcpy((uint8_t*)e->h_stage[slot] + fill, src, m);
      fill += m; src += m; n -= m;
      if (fill == kStageBytes) if (int rc = flush()) return rc;
    }
    return DMX_OK;
  }
  int flush() {
    if (!fill) return DMX_OK;
    HIP_TRY(hipMemcpyAsync(dst, e->h_stage[slot], fill, hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipEventRecord(e->ev_stage[slot], e->stream));
    e->stage_busy[slot] = true;
    dst += fill; fill = 0; e->stage_cur ^= 1;
    return DMX_OK;
  }
};

// Cells cells[0..nb) of the host pileup `pl` (NULL = all of them, in order) become cells 0..nb-1 of the engine: the CSR is
// re-based on the fly while it streams to the device, runs of consecutive cells move as one piece.
// A range of a DEVICE-resident pileup whose cells are not consecutive: the cells' pieces of the three big arrays are copied to the
// engine's own buffers on the device (one workgroup per cell; src = the cells' first pair / first read byte in the caller's arrays).
__global__ __launch_bounds__(256) void k_gather_cells(const int64_t* __restrict__ src_pair, const int64_t* __restrict__ src_read,
                                                      const int64_t* __restrict__ dst_pair_off, const int64_t* __restrict__ dst_read_off,
                                                      const int32_t* __restrict__ pair_snp, const uint8_t* __restrict__ pair_nrd, int w,
                                                      const uint8_t* __restrict__ reads, int32_t* __restrict__ o_snp,
                                                      uint8_t* __restrict__ o_nrd, uint8_t* __restrict__ o_reads) {
  const int k = blockIdx.x;
  const int64_t sp = src_pair[k], sr = src_read[k], dp = dst_pair_off[k], dr = dst_read_off[k];
  const int64_t np = dst_pair_off[k + 1] - dp, nr = dst_read_off[k + 1] - dr;
  if (pair_snp) for (int64_t i = threadIdx.x; i < np; i += blockDim.x) o_snp[dp + i] = pair_snp[sp + i];
  for (int64_t i = threadIdx.x; i < np * w; i += blockDim.x) o_nrd[dp * w + i] = pair_nrd[sp * w + i];
  for (int64_t i = threadIdx.x; i < nr; i += blockDim.x) o_reads[dr + i] = reads[sr + i];
}

// dmx_engine_set_pileup for cells cells[0..nb) (NULL: all) of a DEVICE-resident pileup; h_po / h_ro = host copies of its two offset
// arrays.  Consecutive cells are a VIEW of the caller's arrays (nothing is copied: the kernels index the big arrays with the absolute
// offsets they read from cell_pair_off / cell_read_off); anything else is gathered on the device.
int set_pileup_cells_device(dmx_engine* e, const dmx_pileup* pl, const int64_t* h_po, const int64_t* h_ro, const int32_t* cells, int32_t nb,
                            const char* who) {
  const int32_t B = nb;
  const bool dense = !pl->pair_snp && pl->n_pairs > 0;
  bool consecutive = true;
  for (int32_t k = 0; k < B; ++k) {
    const int32_t c = cells ? cells[k] : k;
    if (c < 0 || c >= pl->n_cells) return set_error(DMX_ERR_ARG, "%s: cell %d of %d", who, c, pl->n_cells);
    if (h_po[c + 1] < h_po[c] || h_ro[c + 1] < h_ro[c]) return set_error(DMX_ERR_ARG, "%s: cell_pair_off / cell_read_off not monotone at %d", who, c);
    if (h_po[c] < 0 || h_po[c + 1] > pl->n_pairs || h_ro[c] < 0 || h_ro[c + 1] > pl->n_reads) return set_error(DMX_ERR_ARG, "%s: offsets of cell %d leave the arrays", who, c);
    if (dense && h_po[c + 1] - h_po[c] != pl->n_snps) return set_error(DMX_ERR_ARG, "%s: dense layout needs n_snps pairs per cell (cell %d)", who, c);
    if (cells && k && cells[k] != cells[k - 1] + 1) consecutive = false;
  }
  std::vector<int64_t> h_off((size_t)B + 1, 0), h_roff((size_t)B + 1, 0);
  for (int32_t k = 0; k < B; ++k) {
    const int32_t c = cells ? cells[k] : k;
    h_off[(size_t)k + 1] = h_off[(size_t)k] + (h_po[c + 1] - h_po[c]);
    h_roff[(size_t)k + 1] = h_roff[(size_t)k] + (h_ro[c + 1] - h_ro[c]);
  }
  const int64_t P = h_off[(size_t)B], R = h_roff[(size_t)B];
  if (consecutive) {
    const int32_t c0 = B ? (cells ? cells[0] : 0) : 0;
    e->pv.cell_pair_off = pl->cell_pair_off + c0; e->pv.cell_read_off = pl->cell_read_off + c0;
    e->pv.pair_snp = pl->pair_snp; e->pv.pair_nrd = pl->pair_nrd; e->pv.reads = pl->reads;
    if (!pl->pair_snp && P == 0) {               // no pairs at all: never the dense layout (NULL would select it)
      if (int rc = ensure_dev(&e->own[2], &e->own_cap[2], 16)) return rc;
      e->pv.pair_snp = (const int32_t*)e->own[2];
    }
    e->pv.R = pl->n_reads;                       // (the bound of the caller's read array: offsets are absolute in a view)
  } else {
    const size_t w = (size_t)pl->nrd_width;
    if (int rc = ensure_dev(&e->own[0], &e->own_cap[0], sizeof(int64_t) * ((size_t)B + 1))) return rc;
    if (int rc = ensure_dev(&e->own[1], &e->own_cap[1], sizeof(int64_t) * ((size_t)B + 1))) return rc;
    if (int rc = ensure_dev(&e->own[2], &e->own_cap[2], dense ? 16 : sizeof(int32_t) * (size_t)P + 16)) return rc;
    if (int rc = ensure_dev(&e->own[3], &e->own_cap[3], (size_t)P * w + 4)) return rc;
    if (int rc = ensure_dev(&e->own[4], &e->own_cap[4], (size_t)R + 4)) return rc;
    if (int rc = ensure_dev(&e->own[5], &e->own_cap[5], sizeof(int64_t) * 2 * (size_t)std::max(B, 1))) return rc;
    std::vector<int64_t> src((size_t)2 * std::max(B, 1));
    for (int32_t k = 0; k < B; ++k) { src[(size_t)k] = h_po[cells[k]]; src[(size_t)B + k] = h_ro[cells[k]]; }
    HIP_TRY(hipMemcpyAsync(e->own[0], h_off.data(), sizeof(int64_t) * ((size_t)B + 1), hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipMemcpyAsync(e->own[1], h_roff.data(), sizeof(int64_t) * ((size_t)B + 1), hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipMemcpyAsync(e->own[5], src.data(), sizeof(int64_t) * 2 * (size_t)B, hipMemcpyHostToDevice, e->stream));
    if (B) {
      hipLaunchKernelGGL(k_gather_cells, dim3((unsigned)B), dim3(256), 0, e->stream, (const int64_t*)e->own[5], (const int64_t*)e->own[5] + B,
                         (const int64_t*)e->own[0], (const int64_t*)e->own[1], dense ? nullptr : pl->pair_snp, (const uint8_t*)pl->pair_nrd, (int)w,
                         pl->reads, (int32_t*)e->own[2], (uint8_t*)e->own[3], (uint8_t*)e->own[4]);
      HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipStreamSynchronize(e->stream));     // (h_off, h_roff, src go away)
    e->pv.cell_pair_off = (const int64_t*)e->own[0]; e->pv.cell_read_off = (const int64_t*)e->own[1];
    e->pv.pair_snp = (dense && P > 0) ? nullptr : (const int32_t*)e->own[2];
    e->pv.pair_nrd = (const uint8_t*)e->own[3]; e->pv.reads = (const uint8_t*)e->own[4];
    e->pv.R = R;
  }
  if (e->pv.pair_snp && P > 0) {                  // what the kernels will index the genotype matrix with (a view checks its own pairs only)
    if (!e->d_bad) HIP_TRY(hipMalloc((void**)&e->d_bad, sizeof(int32_t)));
    HIP_TRY(hipMemsetAsync(e->d_bad, 0, sizeof(int32_t), e->stream));
    const int32_t* first = consecutive ? pl->pair_snp + h_po[B ? (cells ? cells[0] : 0) : 0] : e->pv.pair_snp;
    hipLaunchKernelGGL(k_check_snp_ids, dim3(1024), dim3(256), 0, e->stream, first, P, e->S, e->d_bad);
    HIP_TRY(hipGetLastError());
  }
  e->pv.B = B; e->pv.S = e->S; e->nrd_width = pl->nrd_width; e->P = P; e->R = R;
  std::vector<int32_t> sched((size_t)B);
  std::iota(sched.begin(), sched.end(), 0);
  std::stable_sort(sched.begin(), sched.end(), [&](int32_t a, int32_t b) { return (h_off[a + 1] - h_off[a]) > (h_off[b + 1] - h_off[b]); });
  e->max_cell_pairs = B ? h_off[sched[0] + 1] - h_off[sched[0]] : 0;   // (the longest barcode comes first)
  if (int rc = ensure_dev((void**)&e->d_sched, &e->sched_cap, sizeof(int32_t) * (size_t)B)) return rc;
  if (B) HIP_TRY(hipMemcpyAsync(e->d_sched, sched.data(), sizeof(int32_t) * (size_t)B, hipMemcpyHostToDevice, e->stream));
  int32_t bad = 0;
  if (e->pv.pair_snp && P > 0) HIP_TRY(hipMemcpyAsync(&bad, e->d_bad, sizeof(int32_t), hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  if (bad) return set_error(DMX_ERR_ARG, "%s: a pair_snp entry is outside [0, %d)", who, e->S);
  return DMX_OK;
}

int set_pileup_cells(dmx_engine* e, const dmx_pileup* pl, const int32_t* cells, int32_t nb, const char* who) {
  const int32_t B = nb;
  std::vector<int64_t> h_off((size_t)B + 1, 0), h_roff((size_t)B + 1, 0);
  const bool dense = !pl->pair_snp && pl->n_pairs > 0;
  for (int32_t k = 0; k < B; ++k) {
    const int32_t c = cells ? cells[k] : k;
    if (c < 0 || c >= pl->n_cells) return set_error(DMX_ERR_ARG, "%s: cell %d of %d", who, c, pl->n_cells);
    const int64_t np = pl->cell_pair_off[c + 1] - pl->cell_pair_off[c], nr = pl->cell_read_off[c + 1] - pl->cell_read_off[c];
    if (np < 0 || nr < 0) return set_error(DMX_ERR_ARG, "%s: cell_pair_off / cell_read_off not monotone at %d", who, c);
    if (pl->cell_pair_off[c] < 0 || pl->cell_pair_off[c + 1] > pl->n_pairs || pl->cell_read_off[c] < 0 || pl->cell_read_off[c + 1] > pl->n_reads)
      return set_error(DMX_ERR_ARG, "%s: offsets of cell %d leave the arrays", who, c);
    if (dense && np != pl->n_snps) return set_error(DMX_ERR_ARG, "%s: dense layout needs n_snps pairs per cell (cell %d)", who, c);
    h_off[(size_t)k + 1] = h_off[(size_t)k] + np;
    h_roff[(size_t)k + 1] = h_roff[(size_t)k] + nr;
  }
  const int64_t P = h_off[(size_t)B], R = h_roff[(size_t)B];
  const size_t w = (size_t)pl->nrd_width;
  if (int rc = ensure_dev(&e->own[0], &e->own_cap[0], sizeof(int64_t) * ((size_t)B + 1))) return rc;
  if (int rc = ensure_dev(&e->own[1], &e->own_cap[1], sizeof(int64_t) * ((size_t)B + 1))) return rc;
  if (int rc = ensure_dev(&e->own[2], &e->own_cap[2], dense ? 16 : sizeof(int32_t) * (size_t)P)) return rc;
  if (int rc = ensure_dev(&e->own[3], &e->own_cap[3], (size_t)P * w + 4)) return rc;
  if (int rc = ensure_dev(&e->own[4], &e->own_cap[4], (size_t)R + 4)) return rc;
  if (!e->h_stage[0]) {
    for (int i = 0; i < 2; ++i) {
      HIP_TRY(hipHostMalloc(&e->h_stage[i], kStageBytes, hipHostMallocDefault));
      HIP_TRY(hipEventCreateWithFlags(&e->ev_stage[i], hipEventDisableTiming));
    }
  }
  HIP_TRY(hipMemcpyAsync(e->own[0], h_off.data(), sizeof(int64_t) * ((size_t)B + 1), hipMemcpyHostToDevice, e->stream));
  HIP_TRY(hipMemcpyAsync(e->own[1], h_roff.data(), sizeof(int64_t) * ((size_t)B + 1), hipMemcpyHostToDevice, e->stream));
  // the three big arrays, one after the other; a run of consecutive source cells is one contiguous piece of each
  for (int arr = 0; arr < 3; ++arr) {
    if (arr == 0 && dense) continue;
    StageWriter sw(e, e->own[2 + arr]);
    for (int32_t k = 0; k < B;) {
      const int32_t c0 = cells ? cells[k] : k;
      int32_t k1 = k + 1;
      if (!cells) k1 = B; else while (k1 < B && cells[k1] == c0 + (k1 - k)) ++k1;
      const int32_t c1 = c0 + (k1 - k);                                 // source cells [c0, c1)
      const int64_t p0 = pl->cell_pair_off[c0], p1 = pl->cell_pair_off[c1], r0 = pl->cell_read_off[c0], r1 = pl->cell_read_off[c1];
      int rc = DMX_OK;
      if (arr == 0) rc = sw.put(pl->pair_snp + p0, sizeof(int32_t) * (size_t)(p1 - p0));
      else if (arr == 1) rc = sw.put((const uint8_t*)pl->pair_nrd + (size_t)p0 * w, (size_t)(p1 - p0) * w);
      else rc = sw.put(pl->reads + r0, (size_t)(r1 - r0));
      if (rc) return rc;
      k = k1;
    }
    if (int rc = sw.flush()) return rc;
  }
  e->pv.cell_pair_off = (const int64_t*)e->own[0]; e->pv.cell_read_off = (const int64_t*)e->own[1];
  e->pv.pair_snp = (dense && P > 0) ? nullptr : (const int32_t*)e->own[2];   // a range without any pair is sparse with empty cells
  e->pv.pair_nrd = (const uint8_t*)e->own[3]; e->pv.reads = (const uint8_t*)e->own[4];
  if (!dense && P > 0) {                          // what the kernels will index the genotype matrix with
    if (!e->d_bad) HIP_TRY(hipMalloc((void**)&e->d_bad, sizeof(int32_t)));
    HIP_TRY(hipMemsetAsync(e->d_bad, 0, sizeof(int32_t), e->stream));
    hipLaunchKernelGGL(k_check_snp_ids, dim3(1024), dim3(256), 0, e->stream, e->pv.pair_snp, P, e->S, e->d_bad);
    HIP_TRY(hipGetLastError());
  }
  e->pv.B = B; e->pv.S = e->S; e->pv.R = R; e->nrd_width = pl->nrd_width; e->P = P; e->R = R;
  // launch order: longest cells first, so the tail of the grid is made of short cells and co-scheduled cells are alike
  std::vector<int32_t> sched((size_t)B);
  std::iota(sched.begin(), sched.end(), 0);
  std::stable_sort(sched.begin(), sched.end(), [&](int32_t a, int32_t b) { return (h_off[a + 1] - h_off[a]) > (h_off[b + 1] - h_off[b]); });
  e->max_cell_pairs = B ? h_off[sched[0] + 1] - h_off[sched[0]] : 0;   // (the longest barcode comes first)
  if (int rc = ensure_dev((void**)&e->d_sched, &e->sched_cap, sizeof(int32_t) * (size_t)B)) return rc;
  if (B) HIP_TRY(hipMemcpyAsync(e->d_sched, sched.data(), sizeof(int32_t) * (size_t)B, hipMemcpyHostToDevice, e->stream));
  int32_t bad = 0;
  if (!dense && P > 0) HIP_TRY(hipMemcpyAsync(&bad, e->d_bad, sizeof(int32_t), hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));       // h_off, h_roff, sched and the caller's arrays may go away after return
  for (int i = 0; i < 2; ++i) e->stage_busy[i] = false;
  if (bad) return set_error(DMX_ERR_ARG, "%s: a pair_snp entry is outside [0, %d)", who, e->S);
  return DMX_OK;
}

}  // namespace

extern "C" int dmx_engine_set_pileup(dmx_engine* e, const dmx_pileup* pl) {
  if (!e || !pl) return set_error(DMX_ERR_ARG, "dmx_engine_set_pileup: null argument");
  return dmx::engine_set_pileup_cells(e, pl, nullptr, pl->n_cells);
}

int dmx::engine_set_pileup_cells(dmx_engine* e, const dmx_pileup* pl, const int32_t* cells, int32_t nb, const int64_t* host_po, const int64_t* host_ro) {
  if (pl->n_cells < 0 || pl->n_pairs < 0 || pl->n_reads < 0 || nb < 0) return set_error(DMX_ERR_ARG, "dmx_engine_set_pileup: negative size");
  if (pl->nrd_width != 1 && pl->nrd_width != 2 && pl->nrd_width != 4) return set_error(DMX_ERR_ARG, "dmx_engine_set_pileup: nrd_width %d", pl->nrd_width);
  if (!pl->cell_pair_off || !pl->cell_read_off || (pl->n_pairs && !pl->pair_nrd) || (pl->n_reads && !pl->reads))
    return set_error(DMX_ERR_ARG, "dmx_engine_set_pileup: missing arrays");
  if (e->S == 0 && pl->n_pairs > 0) return set_error(DMX_ERR_STATE, "dmx_engine_set_pileup: call dmx_engine_set_genotypes first");
  if (pl->n_snps > e->S) return set_error(DMX_ERR_ARG, "dmx_engine_set_pileup: pileup has %d SNPs, genotype matrix %d", pl->n_snps, e->S);
  HIP_TRY(hipSetDevice(e->device));
  e->have_pileup = false;
  const int32_t B = nb;
  if (pl->memory == DMX_MEM_DEVICE) {
    hipPointerAttribute_t at{};                   // the arrays must be visible to this engine's device
    if (hipPointerGetAttributes(&at, pl->cell_pair_off) == hipSuccess && at.type == hipMemoryTypeDevice && at.device != e->device) {
      int can = 0;
      (void)hipDeviceCanAccessPeer(&can, e->device, at.device);
      if (!can) return set_error(DMX_ERR_ARG, "dmx_engine_set_pileup: the pileup lives on device %d, the engine on device %d (no peer access)", at.device, e->device);
    }
    (void)hipGetLastError();
    const int64_t *h_po = host_po, *h_ro = host_ro;
    std::vector<int64_t> po_buf, ro_buf;
    if (!h_po || !h_ro) {
      po_buf.resize((size_t)pl->n_cells + 1); ro_buf.resize((size_t)pl->n_cells + 1);
      HIP_TRY(hipMemcpy(po_buf.data(), pl->cell_pair_off, sizeof(int64_t) * ((size_t)pl->n_cells + 1), hipMemcpyDeviceToHost));
      HIP_TRY(hipMemcpy(ro_buf.data(), pl->cell_read_off, sizeof(int64_t) * ((size_t)pl->n_cells + 1), hipMemcpyDeviceToHost));
      h_po = po_buf.data(); h_ro = ro_buf.data();
    }
    if (!cells && (h_po[0] != 0 || h_po[(size_t)B] != pl->n_pairs)) return set_error(DMX_ERR_ARG, "dmx_engine_set_pileup: cell_pair_off does not span n_pairs");
    if (int rc = set_pileup_cells_device(e, pl, h_po, h_ro, cells, nb, "dmx_engine_set_pileup")) return rc;
  } else {
    if (!cells && (pl->cell_pair_off[0] != 0 || pl->cell_pair_off[(size_t)B] != pl->n_pairs))
      return set_error(DMX_ERR_ARG, "dmx_engine_set_pileup: cell_pair_off does not span n_pairs");
    if (int rc = set_pileup_cells(e, pl, cells, nb, "dmx_engine_set_pileup")) return rc;
  }
  // dense pileups: SNP-minor copies of the genotype probabilities for the singlet kernel (once per genotype matrix)
  if (!e->pv.pair_snp && e->S > 0 && !e->have_gT) {
    if (e->d_gT) { (void)hipFree(e->d_gT); e->d_gT = nullptr; }
    if (e->d_g0T) { (void)hipFree(e->d_g0T); e->d_g0T = nullptr; }
    HIP_TRY(hipMalloc((void**)&e->d_gT, (size_t)e->S * e->V * 3 * sizeof(float)));
    HIP_TRY(hipMalloc((void**)&e->d_g0T, (size_t)e->S * 3 * sizeof(double)));
    hipLaunchKernelGGL(k_transpose_geno, dim3(2048), dim3(256), 0, e->stream, e->d_g, e->d_gp0, e->S, e->V, e->d_gT, e->d_g0T);
    HIP_TRY(hipGetLastError());
    e->have_gT = true;
  }
  // sparse pileups over a genotype matrix that does not fit the L2: the SNP-block table of the blocked K1 walk (launch_singlet)
  e->blk_n = 0;
  if (e->pv.pair_snp && e->P > 0 && e->S > 0 && !e->knob("DMX_K1_NO_BLOCKS")) {
    const double row_bytes = (double)e->V * 12.0 + 24.0;
    double target = 2.0 * 1024 * 1024;                                  // bytes of genotype rows per block
    const char* env = e->knob("DMX_K1_BLOCK_BYTES");                     // tests / kernel experiments: any block size, no size heuristics
    if (env) target = atof(env);
    int shift = 0;
    while (shift < 30 && (double)(2ll << shift) * row_bytes <= target) ++shift;
    const int32_t nblk = (int32_t)(((int64_t)e->S + (1ll << shift) - 1) >> shift);
    // worth it only when the matrix is well beyond one L2 and a barcode still has a few tiles of pairs per block
    if (nblk > 1 && (env || ((double)e->S * row_bytes > 3.0 * 1024 * 1024 && (double)e->P / std::max(B, 1) / nblk >= 96.0))) {
      if (int rc = ensure_dev((void**)&e->d_blk, &e->blk_cap, sizeof(int64_t) * 2 * (size_t)B * (nblk + 1))) return rc;
      hipLaunchKernelGGL(k_snp_blocks, dim3((unsigned)((B + 3) / 4)), dim3(kThreads), 0, e->stream, e->pv, e->nrd_width, shift, nblk, e->d_blk);
      HIP_TRY(hipGetLastError());
      e->blk_shift = shift; e->blk_n = nblk;
    }
  }
  if (e->out_cap < B || !e->d_llks) {
    free_results(e);
    const int32_t cap = B + B / 16 + 16;
    HIP_TRY(hipMalloc((void**)&e->d_llks, sizeof(double) * (size_t)cap * e->V));
    HIP_TRY(hipMalloc((void**)&e->d_llk0s, sizeof(double) * (size_t)cap));
    e->out_cap = cap;
  }
  // k_singlet_can's preconditions: every ABSOLUTE pair index / read offset the kernels form fits 32 bits (a view of a caller's device arrays
  // indexes the caller's whole arrays), and the four bytes behind the last read byte are readable (the engine's own copies are allocated
  // with slack; a caller's device array is, when its allocation extends that far)
  {
    const bool own_reads = e->pv.reads == (const uint8_t*)e->own[4];
    const uint64_t maxP = own_reads ? (uint64_t)e->P : (uint64_t)std::max<int64_t>(pl->n_pairs, e->P);
    const uint64_t maxR = (uint64_t)std::max<int64_t>(e->pv.R, e->R);
    e->off32 = maxP + 4 * 64 < (1ull << 32) && maxR + 64 < (1ull << 32);
    e->reads_padded = own_reads;
    if (!own_reads && e->pv.reads) {
      hipDeviceptr_t base = nullptr; size_t size = 0;
      if (hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)e->pv.reads) == hipSuccess)
        e->reads_padded = (const uint8_t*)base + size >= e->pv.reads + maxR + 4;
      (void)hipGetLastError();
    }
  }
  e->have_sing = e->have_grid = false;
  e->have_pileup = true;
  e->k1_fn = e->k2_fn = e->k3b_fn = nullptr; e->k1_placement = 0;      // nothing has run on this pileup yet
  return DMX_OK;
}

namespace {

// Every K1 / K2 / K3b launch goes through these: the engine remembers WHICH kernel it launched (dmx_engine_kernel_names), so that nothing
// outside launch_singlet / launch_doublet / launch_certify re-derives their selection (bench.py pairs its counter files with it).
#define DMX_LAUNCH(SLOT, K, ...) do { e->SLOT = reinterpret_cast<const void*>(&K); hipLaunchKernelGGL(K, __VA_ARGS__); } while (0)
#define DMX_LAUNCH_IF(COND, SLOT, K, ...) do { if (COND) e->SLOT = reinterpret_cast<const void*>(&K); hipLaunchKernelGGL(K, __VA_ARGS__); } while (0)

// k_build_certify_finals' table (round 6), for the kernels that run phase 1 of the default grid {0, 0.5}: k_doublet_sym and k_certify<FIVE>.  Only shallow
// pileups take it — at most 1.6 stored reads per covered pair on average: a tile skips its read loop only when NONE of its 16 / 32 pairs is deeper than
// three reads (1.25 reads per pair: 93 % of the tiles; 2 reads per pair, cfg5: 7 %, and there the look-ups into a table beyond the L2 cost more than
// they save — measured, k_certify 7.57 -> 8.23 ms).  *out stays NULL when the pileup is deeper, or too small to pay for the table.
int ensure_seeds(dmx_engine* e, const double** out) {      // k_build_certify_seeds' table (default grid {0, 0.5}: k_certify<FIVE>, k_doublet_sym)
  if (!e->cseed_valid) {
    if (!e->d_cseed) HIP_TRY(hipMalloc((void**)&e->d_cseed, sizeof(double) * 2 * kCSeedStride * (size_t)kCSeedN));
    hipLaunchKernelGGL(k_build_certify_seeds, dim3((unsigned)((kCSeedN + 255) / 256)), dim3(256), 0, e->stream, e->d_lut, e->d_cseed);
    HIP_TRY(hipGetLastError());
    e->cseed_valid = true;
  }
  *out = e->d_cseed;
  return DMX_OK;
}
int ensure_finals(dmx_engine* e, const double** out) {
  *out = nullptr;
  if ((double)e->R > 1.6 * (double)std::max<int64_t>(e->P, 1) && !e->knob("DMX_FINALS_ANY_DEPTH")) return DMX_OK;
  // ... and only jobs big enough to pay for it: allocating and filling the 58 MB table costs an engine ~1.5 ms (cfg6's 0.035 s job: +4 %, measured),
  // what it saves is ~2 ms per 1e8 covered pairs (cfg3, 5e8 pairs: 13 ms STRICT, 7 ms FAST)
  if (e->P < 100000000 && !e->cfin_valid && !e->knob("DMX_FINALS_ANY_DEPTH")) return DMX_OK;
  if (!e->cfin_valid) {
    if (!e->d_cfin) HIP_TRY(hipMalloc((void**)&e->d_cfin, sizeof(double) * kCFinStride * (size_t)kCFinN));
    hipLaunchKernelGGL(k_build_certify_finals, dim3((unsigned)((kCFinN + 255) / 256)), dim3(256), 0, e->stream, e->d_lut, e->d_cfin);
    HIP_TRY(hipGetLastError());
    e->cfin_valid = true;
  }
  *out = e->d_cfin;
  return DMX_OK;
}

int launch_singlet(dmx_engine* e) {
  const int32_t B = e->pv.B, V = e->V;
  // cells per wavefront: more cells amortise the ordered sums, fewer keep >= ~4 wavefronts per SIMD in flight (1024 SIMDs)
  int CW = (B >= 64 * 1024) ? 4 : (B >= 24 * 1024 ? 2 : 1);
  const int KC = (V <= 4) ? 4 : 8;
  if (const char* cenv = e->knob("DMX_K1_CW")) CW = atoi(cenv);     // kernel experiments only
  if (e->n_classes > 0 && !e->knob("DMX_NO_CLASSES") && !e->knob("DMX_NO_K1_CLASSES")) {
    // --field GT inputs: log() once per genotype class instead of once per sample (bit-identical, see k_singlet_cls)
    const int wide_v = e->knob("DMX_K1_WIDE_V") ? atoi(e->knob("DMX_K1_WIDE_V")) : 20;      // kernel experiments only
    if (V >= wide_v && V <= 1024) {              // wide panels: chains look the class terms up themselves, one pass per tile
      const size_t dynw = (size_t)(kThreads / 64) * 64 * ((V + 15) / 16) * sizeof(uint32_t);
      const bool l0m = V % 64 == 0 && !e->knob("DMX_K1W_NO_L0M");     // llk0 merged into the first pass instead of a pass of its own (k_singlet_clsw)
      const dim3 blkw(kThreads), grdw((unsigned)((B + (kThreads / 64) - 1) / (kThreads / 64)), (unsigned)((V + (l0m ? 0 : 1) + 255) / 256));
#define DMX_K1W(...) do {                                                                                                          \
      if (dynw > 30 * 1024) HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_singlet_clsw<__VA_ARGS__>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dynw)); \
      DMX_LAUNCH(k1_fn, (k_singlet_clsw<__VA_ARGS__>), grdw, blkw, dynw, e->stream, e->pv, e->nrd_width, e->d_rows, e->d_idw, e->d_gp0,    \
                         e->d_lut, e->d_sched, V, e->d_llks, e->d_llk0s); } while (0)
      if (e->knob("DMX_K1W_MINW3")) { if (l0m) DMX_K1W(1, 3, true); else DMX_K1W(1, 3, false); }              // kernel experiments only
      else if (l0m) DMX_K1W(1, 4, true);
      else DMX_K1W(1, 4, false);
#undef DMX_K1W
      return DMX_OK;
    }
    // measured on cfg2-shaped inputs (profiles/): CW 2 wins from 10 k barcodes (6.90 vs 7.14 ms; 5 k: 5.07 vs 3.94).  Four barcodes per wavefront
    // (round 6, profiles/r06_k1_cw4.txt): k_singlet_cls LOSES with them at every size tried — 131 072 barcodes x 50 k SNPs 64.4 against
    // 50.7 ms, 262 144: 127.2 / 100.7 (round 3: 40 k 27.7 / 24.6) — so it never takes them; the lean kernel k_singlet_can gains 2.6 % at 65 536
    // (17.15 / 17.60 ms) and takes them from there (beyond 2^32 covered pairs per engine — 131 072 x 50 k — the lean kernel's 32-bit offsets do not
    // reach and k_singlet_cls runs: DESIGN 10).
    const bool lean_ok = e->canon && e->geno_safe && !e->knob("DMX_NO_CANON_K1") && V <= 16 && e->nrd_width == 1 && e->reads_padded && e->off32 &&
                         (uint64_t)e->S * 64 < (1ull << 32) && !e->knob("DMX_K1_NO_LEAN");       // (the conditions of `lean` below)
    if (!e->knob("DMX_K1_CW")) CW = (lean_ok && B >= 64 * 1024) ? 4 : (B >= 10000 ? 2 : 1);
    const int nchc = (V + KC - 1) / KC;
    const size_t dynb = sizeof(double) * (size_t)(kThreads / 64) * nchc * CW * (KC + 1);
    if (dynb <= 16 * 1024) {
      const dim3 blk(kThreads), grd((unsigned)((B + (kThreads / 64) * CW - 1) / ((kThreads / 64) * CW)));
      const int chk = e->geno_safe ? 0 : 1;       // (DMX_FORCE_CHECK=1 keeps the test: bit-identical, tests/test_gpu_parity.py)
      // canonical GT classes: three of a pair's five log terms from the table (k_singlet_cls<.., CAN>; DMX_NO_CANON_K1=1: the plain class form)
      const bool can = e->canon && chk == 0 && !e->knob("DMX_NO_CANON_K1");
      // the lean canonical-class kernel (round 5): one id word per SNP, one-byte read counts, every offset in 32 bits, padded read bytes
      const bool lean = can && V <= 16 && e->nrd_width == 1 && e->reads_padded && e->off32 && (uint64_t)e->S * 64 < (1ull << 32) && !e->knob("DMX_K1_NO_LEAN");
      if (lean) {
        if (!e->snprec_valid) {
          if (e->d_snprec) { (void)hipFree(e->d_snprec); e->d_snprec = nullptr; }
          HIP_TRY(hipMalloc((void**)&e->d_snprec, kSnpRecBytes * (size_t)std::max(e->S, 1)));
          hipLaunchKernelGGL(k_build_snprec, dim3((unsigned)((e->S + 255) / 256)), dim3(256), 0, e->stream, e->d_gp0, e->d_ids, e->d_oth, e->S, V, e->d_snprec);
          HIP_TRY(hipGetLastError());
          e->snprec_valid = true;
        }
        if (!e->ctab_valid) {
          if (!e->d_ctab) HIP_TRY(hipMalloc((void**)&e->d_ctab, sizeof(double) * 8 * (size_t)kCtN));
          hipLaunchKernelGGL(k_build_ctab, dim3((unsigned)((kCtN + 255) / 256)), dim3(256), 0, e->stream, e->d_lut, (double)e->can_hi, (double)e->can_lo, e->d_ctab);
          HIP_TRY(hipGetLastError());
          e->ctab_valid = true;
        }
        // (a matrix with a fourth genotype row somewhere — missing genotypes — keeps four class planes of scratch and gathers every entry from the
        //  global table; without one the commonest entries come from an LDS copy: see the kernel)
        const bool oth = e->any_oth;
        // dense pileups of up to 8 samples without a fourth genotype row (cfg2): producers / consumer (k_singlet_canp) — an EXPERIMENT (DMX_K1_CANP=1), bit-identical
        // and not faster: cfg2 3.34 ms at three workgroups per CU against k_singlet_can's 3.15 (4.11 at two, 7.8 at four with spills).  Its consumer does what
        // it was built for (64 ordered steps of sdwa-add / ds_read / add for 448 pairs), but a producer's own per-tile work — scan, table index, look-up, llk0 log:
        // ~120 VALU instructions — is what both kernels are made of, and both wait for the same two HBM streams (timing builds without the read-count and
        // read-byte loads: 2.31 and 2.71 ms; touching those lines 4 / 8 / 16 tiles ahead from the consumer's spare lanes: 3.44-3.50 ms, no help) — DESIGN 11.
        if (!oth && V <= 8 && e->pv.pair_snp == nullptr && e->knob("DMX_K1_CANP") && !e->knob("DMX_K1_NO_CANP") && !e->knob("DMX_K1_CW")) {
          if (!e->clsb_valid) {
            if (e->d_clsb) { (void)hipFree(e->d_clsb); e->d_clsb = nullptr; }
            e->clsb_stride = (uint32_t)(((int64_t)e->S + 63) / 64 * 64 + 64);
            HIP_TRY(hipMalloc((void**)&e->d_clsb, (size_t)9 * e->clsb_stride));
            hipLaunchKernelGGL(k_build_clsb, dim3((unsigned)(((size_t)9 * e->clsb_stride + 255) / 256)), dim3(256), 0, e->stream, e->d_ids, e->S, V, e->clsb_stride, e->d_clsb);
            HIP_TRY(hipGetLastError());
            e->clsb_valid = true;
          }
          const dim3 grdp((unsigned)((B + kCpB - 1) / kCpB));
          if (e->knob("DMX_K1_CANP_MINW4"))        // kernel experiments only (two workgroups per CU)
            DMX_LAUNCH(k1_fn, (k_singlet_canp<4>), grdp, dim3(kCpThreads), 0, e->stream, e->pv, e->d_snprec, e->d_clsb, e->clsb_stride, e->d_ctab, e->d_lut, e->d_sched, V,
                               e->d_llks, e->d_llk0s, (double)e->can_hi, (double)e->can_lo);
          else
            DMX_LAUNCH(k1_fn, (k_singlet_canp<6>), grdp, dim3(kCpThreads), 0, e->stream, e->pv, e->d_snprec, e->d_clsb, e->clsb_stride, e->d_ctab, e->d_lut, e->d_sched, V,
                               e->d_llks, e->d_llk0s, (double)e->can_hi, (double)e->can_lo);
          return DMX_OK;
        }
#define DMX_K1L_(CC, KK, OO, NN) DMX_LAUNCH(k1_fn, (k_singlet_can<CC, KK, OO, NN>), grd, blk, 0, e->stream, e->pv, e->d_snprec, e->d_rows, e->d_ctab, e->d_lut, e->d_sched, V, \
                                            e->d_llks, e->d_llk0s, (double)e->can_hi, (double)e->can_lo)
#define DMX_K1L(CC, KK, NN) do { if (oth) DMX_K1L_(CC, KK, true, NN); else DMX_K1L_(CC, KK, false, NN); } while (0)
        if (KC == 4) { if (CW == 4) DMX_K1L(4, 4, 1); else if (CW == 2) DMX_K1L(2, 4, 1); else DMX_K1L(1, 4, 1); }
        else if (nchc == 1) { if (CW == 4) DMX_K1L(4, 8, 1); else if (CW == 2) DMX_K1L(2, 8, 1); else DMX_K1L(1, 8, 1); }
        else                { if (CW == 4) DMX_K1L(4, 8, 2); else if (CW == 2) DMX_K1L(2, 8, 2); else DMX_K1L(1, 8, 2); }
#undef DMX_K1L
#undef DMX_K1L_
        return DMX_OK;
      }
      if (can && !e->ltab_valid) {
        if (!e->d_ltab) HIP_TRY(hipMalloc((void**)&e->d_ltab, sizeof(double) * 4 * (size_t)kCanN));
        hipLaunchKernelGGL(k_build_canon_logs, dim3((unsigned)((kCanN + 255) / 256)), dim3(256), 0, e->stream, e->d_lut, (double)e->can_hi,
                           (double)e->can_lo, e->d_ltab);
        HIP_TRY(hipGetLastError());
        e->ltab_valid = true;
      }
#define DMX_K1C_(CC, KK, CAN_) DMX_LAUNCH(k1_fn, (k_singlet_cls<CC, KK, CAN_>), grd, blk, dynb, e->stream, e->pv, e->nrd_width, e->d_rows, chk, \
                                           e->d_idw, e->d_gp0, e->d_lut, e->d_sched, V, e->d_llks, e->d_llk0s, e->d_ltab, e->d_oth,           \
                                           (double)e->can_hi, (double)e->can_lo)
#define DMX_K1C(CC, KK) do { if (can) DMX_K1C_(CC, KK, true); else DMX_K1C_(CC, KK, false); } while (0)
      if (KC == 4) { if (CW == 4) DMX_K1C(4, 4); else if (CW == 2) DMX_K1C(2, 4); else DMX_K1C(1, 4); }
      else         { if (CW == 4) DMX_K1C(4, 8); else if (CW == 2) DMX_K1C(2, 8); else DMX_K1C(1, 8); }
#undef DMX_K1C
#undef DMX_K1C_
      return DMX_OK;
    }
  }
  const int nch = (V + KC - 1) / KC;
  const int NW = kThreads / 64;
  const int QS = std::min(nch, (int)(16 * 1024 / (sizeof(double) * (size_t)NW * CW * (KC + 1))));   // chunks per workgroup: 16 KB of sums
  const unsigned q_slabs = (unsigned)((nch + QS - 1) / QS);
  const size_t dyn = sizeof(double) * (size_t)NW * QS * CW * (KC + 1);
  const bool dense = e->pv.pair_snp == nullptr;
  if (dense && (size_t)e->S * V * 12 > 0x7FFFFFFFull) return set_error(DMX_ERR_ARG, "run_singlet: dense genotype matrix over 2 GiB is not supported by this build");
  const float* gq = dense ? e->d_gT : e->d_g;
  const double* g0q = dense ? e->d_g0T : e->d_gp0;
  const dim3 block(kThreads), grid((unsigned)((B + NW * CW - 1) / (NW * CW)), q_slabs);
  // Sparse pileups gather one genotype row per covered pair; when the matrix is larger than an XCD's L2 (4 MB) nearly every row
  // comes out of the Infinity Cache (cfg5: 56 GB per launch for 1.5 GB of algorithmic bytes, profiles/r02_cfg5_pmc_summary.json).
  // The SNP axis is then walked in blocks of 2^blk_shift SNPs (about 2 MB of rows), one launch per block over all barcodes: the
  // rows of a launch stay L2-resident; the sums are parked in llks / llk0s between launches (see the kernel).
  const int64_t* blk = (!dense && e->blk_n > 1) ? e->d_blk : nullptr;
  const int n_launch = blk ? e->blk_n : 1;
  // soft fields, 9..64 samples: the kernel whose lanes own their samples' sums (k_singlet_own; DMX_K1_NO_OWN=1: k_singlet, bit-identical)
  if (V >= 9 && V <= 64 && (size_t)e->S * V * 12 <= 0x7FFFFFFFull && !e->knob("DMX_K1_NO_OWN")) {
    const int tpc = V <= 16 ? 16 : (V <= 32 ? 32 : 64);
    const int cwo = 64 / tpc;
    const dim3 grd((unsigned)((B + NW * cwo - 1) / (NW * cwo)));
    const bool chk = !e->geno_safe || e->knob("DMX_FORCE_CHECK");
#define DMX_K1O_(TT, DD, CC)                                                                                           \
    for (int bi_ = 0; bi_ < n_launch; ++bi_)                                                                           \
      DMX_LAUNCH(k1_fn, (k_singlet_own<TT, DD, CC>), grd, block, 0, e->stream, e->pv, e->nrd_width, e->d_g, e->d_gp0, e->d_lut, \
                         e->d_sched, V, e->d_llks, e->d_llk0s, blk, bi_, e->blk_n)
#define DMX_K1O(TT) do { if (dense) { if (chk) DMX_K1O_(TT, true, true); else DMX_K1O_(TT, true, false); }            \
                         else { if (chk) DMX_K1O_(TT, false, true); else DMX_K1O_(TT, false, false); } } while (0)
    if (tpc == 16) DMX_K1O(16); else if (tpc == 32) DMX_K1O(32); else DMX_K1O(64);
#undef DMX_K1O
#undef DMX_K1O_
    return DMX_OK;
  }
#define DMX_K1(CC, KK, DD)                                                                                            \
  for (int bi_ = 0; bi_ < n_launch; ++bi_)                                                                             \
    DMX_LAUNCH(k1_fn, (k_singlet<CC, KK, DD>), grid, block, dyn, e->stream, e->pv, e->nrd_width, gq, g0q, e->d_lut,   \
                       e->d_sched, V, QS, e->d_llks, e->d_llk0s, blk, bi_, e->blk_n)
#define DMX_K1_D(CC, KK) do { if (dense) DMX_K1(CC, KK, true); else DMX_K1(CC, KK, false); } while (0)
  if (KC == 4) { if (CW == 4) DMX_K1_D(4, 4); else if (CW == 2) DMX_K1_D(2, 4); else DMX_K1_D(1, 4); }
  else         { if (CW == 4) DMX_K1_D(4, 8); else if (CW == 2) DMX_K1_D(2, 8); else DMX_K1_D(1, 8); }
#undef DMX_K1_D
#undef DMX_K1
  return DMX_OK;
}

template <typename NRD, bool FIXUP>
int launch_doublet_generic(dmx_engine* e) {
  const int32_t B = e->pv.B, V = e->V, A = e->A;
  int A_pad = 1;
  while (A_pad < A) A_pad <<= 1;
  const int TP = std::min(32, kThreads / A_pad);
  const int64_t nacc = (int64_t)V * V * A + A;
  const int per = (int)((nacc + kThreads - 1) / kThreads);
  if (nacc > 0x7FFFFFFFll || V > 0xFFE) return set_error(DMX_ERR_ARG, "run_doublet: V*V*A = %lld accumulators per cell exceed this build's limit", (long long)nacc);
  const int slab_acc = 33;                        // accumulators per thread when the grid is cut into slabs
  const unsigned slabs = per <= 65 ? 1u : (unsigned)((nacc + (int64_t)slab_acc * kThreads - 1) / ((int64_t)slab_acc * kThreads));
  if (slabs > 65535u) return set_error(DMX_ERR_ARG, "run_doublet: V*V*A = %lld accumulators per cell exceed this build's limit (5.5e8)", (long long)nacc);
  // the fix-up pass strides a fixed grid over the cells (see the kernel); the first pass has one workgroup per cell
  const dim3 grid(FIXUP ? (unsigned)std::min(B, 256) : (unsigned)B, slabs), block(kThreads);
#define DMX_K2(NN)                                                                                                   \
  DMX_LAUNCH_IF(!FIXUP, k2_fn, (k_doublet_generic<NRD, NN, FIXUP>), grid, block, 0, e->stream, e->pv, e->d_g, e->d_gp0, e->d_lut, \
                     e->d_alpha, e->d_sched, V, A, A_pad, TP, e->d_grid, e->d_l00, e->d_flag)
  if (per <= 1) DMX_K2(1);
  else if (per <= 2) DMX_K2(2);
  else if (per <= 4) DMX_K2(4);
  else if (per <= 8) DMX_K2(8);
  else if (per <= 16) DMX_K2(16);
  else if (per <= 33) DMX_K2(33);
  else if (per <= 65) DMX_K2(65);
  else DMX_K2(33);                                // slabs of 33 * 256 accumulators
#undef DMX_K2
  return DMX_OK;
}

template <bool FIXUP>
int launch_doublet_generic_w(dmx_engine* e) {
  return e->nrd_width == 1 ? launch_doublet_generic<uint8_t, FIXUP>(e)
                           : (e->nrd_width == 2 ? launch_doublet_generic<uint16_t, FIXUP>(e) : launch_doublet_generic<uint32_t, FIXUP>(e));
}

// The A = 2 kernel with its fix-up pass; other alpha grids (or V > 64) take the generic kernel.
// FAST's soft-field kernels take their phase-2 terms through dmx_log2_lite, whose dropped r^5/5 term is at most 5.7e-15 per term with the sign of r:
// unbiased over varied inputs, but a barcode whose terms cluster on ONE value (a PL field repeats a handful of triples) can collect it linearly — the worst
// case is 6.7e-15 x (covered SNPs) with the result's own rounding (measured at the bin edges, tests/test_dmx_log.py), which reaches FAST's 1e-9 contract at
// 1.5e5.  Barcodes deeper than 130 000 covered SNPs run the STRICT kernels (bit-exact,
// inside FAST's contract); nothing in BASELINE.json comes near (cfg3: 5e4, cfg5: 1e4 covered SNPs per barcode; cfg4's GT classes take no per-term log).
constexpr int64_t kLiteLogMaxPairs = 130000;

int launch_doublet(dmx_engine* e) {
  const int32_t B = e->pv.B, V = e->V, A = e->A;
  e->k2_sym = false;
  const bool fast_soft = e->mode == DMX_MODE_FAST && (e->max_cell_pairs <= kLiteLogMaxPairs || e->knob("DMX_LITE_LOG_ANY_DEPTH"));
  const bool force_generic = e->knob("DMX_K2_GENERIC") != nullptr;      // kernel experiments only
  const bool use_cls = e->n_classes > 0 && !e->knob("DMX_NO_CLASSES");
  if (A >= 3 && A <= 8 && use_cls && V <= 1024 && !force_generic) {
    // GT inputs, longer alpha grids: the class kernel with AP alphas per pair.  One cell per workgroup (the class table is
    // 4 x 4 x AP doubles per pair: 16-32 KB a tile); the k-block width follows the panel.
    const int AP = A <= 4 ? 4 : 8;
    const int VS = (V + 15) & ~15;
    size_t cb = (size_t)32 * AP * 9 * 8 + (size_t)32 * 16 * AP * 8 + (size_t)AP * 34 * 8 + 32 * (4 + 4 + 8) + (size_t)32 * 12 * 4 + (size_t)32 * VS;
    cb = (cb + 15) & ~(size_t)15;
    HIP_TRY(hipMemsetAsync(e->d_flag - kFlagHead, 0, (size_t)B + kFlagHead, e->stream));
    auto slabs = [&](int tpc, int nk) { const int kb = (V + nk - 1) / nk, js = tpc / kb; return (unsigned)((V + js - 1) / js); };
#define DMX_K2CN(NK, APP)                                                                                              \
  do {                                                                                                                 \
    if (cb > 60 * 1024) HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_doublet_clsn<256, NK, APP>),       \
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)cb));              \
    DMX_LAUNCH(k2_fn, (k_doublet_clsn<256, NK, APP>), dim3((unsigned)B, slabs(256, NK)), dim3(kThreads), cb, e->stream, \
                       e->pv, e->nrd_width, e->d_rows, e->d_ids, e->d_gp0, e->d_lut, e->d_alpha, e->d_sched, V, A, VS,   \
                       e->d_grid, e->d_l00, e->d_flag);                                                                  \
  } while (0)
    if (AP == 4) { if (V <= 16) DMX_K2CN(1, 4); else if (V <= 32) DMX_K2CN(4, 4); else DMX_K2CN(8, 4); }
    else         { if (V <= 16) DMX_K2CN(1, 8); else DMX_K2CN(4, 8); }
#undef DMX_K2CN
    HIP_TRY(hipGetLastError());
    return launch_doublet_generic_w<true>(e);
  }
  if (fast_soft && !use_cls && A >= 2 && A <= 8 && V <= 128 && e->alpha[0] == 0.0 && !(A == 2 && e->alpha[1] == 0.5) &&
      !force_generic && !e->knob("DMX_NO_ANF")) {
    // FAST, soft fields, any alpha grid that starts with 0 (the default grid {0, 0.5} has k_doublet_sym): the printed entries only,
    // bilinear form
    const int AP = A <= 2 ? 2 : (A <= 4 ? 4 : 8);
    const int GS = (V * 3 + 3) & ~3;
    HIP_TRY(hipMemsetAsync(e->d_flag - kFlagHead, 0, (size_t)B + kFlagHead, e->stream));
    const dim3 block(kThreads);
    auto slabs = [&](int tpc, int nk) { const int kb = (V + nk - 1) / nk, js = tpc / kb; return (unsigned)((V + js - 1) / js); };
#define DMX_K2NF_(TPC, NK, APP, VUS, SUBP, MINW, CHK)                                                                  \
  do {                                                                                                                 \
    const int NU = (A - 1) * 3 * VUS + 4;                                                                              \
    const size_t cell_bytes = (size_t)32 * APP * 9 * 8 + (size_t)32 * GS * 4 + (size_t)APP * 34 * 8 + 32 * (4 + 4 + 8) +  \
                              (size_t)SUBP * NU * 8 + (size_t)TPC * 8;                                                 \
    const size_t lds = cell_bytes * (kThreads / TPC);                                                                  \
    if (lds > 60 * 1024) HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_doublet_anf<TPC, NK, APP, VUS, SUBP, MINW, CHK>),  \
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));            \
    DMX_LAUNCH(k2_fn, (k_doublet_anf<TPC, NK, APP, VUS, SUBP, MINW, CHK>), dim3((unsigned)((B + (kThreads / TPC) - 1) / (kThreads / TPC)), slabs(TPC, NK)), \
                       block, lds, e->stream, e->pv, e->nrd_width, e->d_g, e->d_gp0, e->d_lut,                            \
                       e->d_alpha, e->d_sched, V, A, GS, e->d_grid, e->d_l00, e->d_flag);                                \
  } while (0)
#define DMX_K2NF(TPC, NK, APP, VUS, SUBP, MINW) do { if (e->geno_safe) DMX_K2NF_(TPC, NK, APP, VUS, SUBP, MINW, false); else DMX_K2NF_(TPC, NK, APP, VUS, SUBP, MINW, true); } while (0)
    // u rows: V entries plus the slack a k-block may read past the panel's end (NK - 1), even; pairs per sub-tile by the LDS they need
    // (one barcode per workgroup throughout: the narrow-panel forms with four barcodes per workgroup need ~86 KB of LDS, i.e. one
    //  workgroup per CU; sub-tiles of 8 pairs and 3 wavefronts per SIMD measured 529 -> 418 ms at V = 32, A = 3, 4 000 barcodes)
    if (AP == 2) {
      if (V <= 16) DMX_K2NF(256, 1, 2, 16, 8, 3); else if (V <= 32) DMX_K2NF(256, 4, 2, 36, 8, 3);
      else if (V <= 64) DMX_K2NF(256, 8, 2, 72, 4, 2); else DMX_K2NF(256, 8, 2, 136, 2, 2);
    } else if (AP == 4) {
      if (V <= 16) DMX_K2NF(256, 1, 4, 16, 8, 3); else if (V <= 32) DMX_K2NF(256, 4, 4, 36, 8, 3);
      else if (V <= 64) DMX_K2NF(256, 8, 4, 72, 4, 2); else DMX_K2NF(256, 8, 4, 136, 2, 2);
    } else {
      if (V <= 16) DMX_K2NF(256, 1, 8, 16, 8, 2); else if (V <= 64) DMX_K2NF(256, 4, 8, 68, 2, 2);
      else DMX_K2NF(256, 4, 8, 132, 1, 2);
    }
#undef DMX_K2NF_
#undef DMX_K2NF
    HIP_TRY(hipGetLastError());
    return launch_doublet_generic_w<true>(e);
  }
  if (A >= 3 && A <= 8 && V <= (A <= 4 ? kAnMaxV4 : kAnMaxV8) && !force_generic) {
    // longer alpha grids: the A = 2 kernel's structure with AP alphas per pair (beyond ~130 samples with more than 64 KB of LDS, as k_doublet_a2)
    const int AP = A <= 4 ? 4 : 8;
    const int GS = (V * 3 + 3) & ~3;
    const size_t cell_bytes = (size_t)32 * AP * 9 * 8 + (size_t)32 * GS * 4 + (size_t)AP * 34 * 8 + 32 * (4 + 4 + 8);
    HIP_TRY(hipMemsetAsync(e->d_flag - kFlagHead, 0, (size_t)B + kFlagHead, e->stream));
    const dim3 block(kThreads);
    auto slabs = [&](int tpc, int nk) { const int kb = (V + nk - 1) / nk, js = tpc / kb; return (unsigned)((V + js - 1) / js); };
    // gfx950 lets one workgroup use all 160 KB of a CU's LDS; above the traditional 64 KB the limit is raised explicitly
#define DMX_K2N(TPC, NK, APP)                                                                                          \
  do {                                                                                                                 \
    const size_t lds = cell_bytes * (kThreads / TPC);                                                                  \
    if (lds > 60 * 1024) HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_doublet_an<TPC, NK, APP>),        \
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));            \
    DMX_LAUNCH(k2_fn, (k_doublet_an<TPC, NK, APP>), dim3((unsigned)((B + (kThreads / TPC) - 1) / (kThreads / TPC)), slabs(TPC, NK)), \
                       block, lds, e->stream, e->pv, e->nrd_width, e->d_g, e->d_gp0, e->d_lut,                            \
                       e->d_alpha, e->d_sched, V, A, GS, e->d_grid, e->d_l00, e->d_flag);                                \
  } while (0)
    if (AP == 4) {
      if (V <= 8) DMX_K2N(64, 1, 4);
      else if (V <= 16) DMX_K2N(64, 4, 4);
      else if (V <= 32) DMX_K2N(256, 4, 4);
      else DMX_K2N(256, 8, 4);
    } else {
      if (V <= 8) DMX_K2N(64, 1, 8);
      else if (V <= 16) DMX_K2N(64, 4, 8);
      else DMX_K2N(256, 4, 8);
    }
#undef DMX_K2N
    HIP_TRY(hipGetLastError());
    return launch_doublet_generic_w<true>(e);
  }
  // wide panels: the class kernel's LDS grows by 32 bytes per sample, the general A = 2 kernel's by 384 (64 KB at V = 128)
  // (FAST on the default grid reaches 512 soft-field samples: k_doublet_sym's slabs keep their LDS flat in V)
  const bool sym_wide = fast_soft && !use_cls && A == 2 && e->alpha[0] == 0.0 && e->alpha[1] == 0.5 && V > 128 && V <= 512 &&
                        !e->knob("DMX_NO_SYM") && !e->knob("DMX_NO_SYM_WIDE");
  // (round 4: the general A = 2 kernel itself runs up to kA2MaxV = 1024 samples — one workgroup may use all 160 KB of a gfx950 CU's LDS, and the
  // tile shortens from 32 to 16 or 8 pairs as the rows grow; beyond, the generic kernel)
  if (A != 2 || force_generic || (V > (use_cls ? 1024 : kA2MaxV) && !sym_wide)) {
    HIP_TRY(hipMemsetAsync(e->d_flag - kFlagHead, 0, (size_t)B + kFlagHead, e->stream));
    if (int rc = launch_doublet_generic_w<false>(e)) return rc;
    HIP_TRY(hipGetLastError());
    return launch_doublet_generic_w<true>(e);
  }
  auto slabs_of = [&](int tpc, int nk) { const int kb = (V + nk - 1) / nk, js = tpc / kb; return (unsigned)((V + js - 1) / js); };
  if (use_cls && e->mode == DMX_MODE_FAST && e->alpha[0] == 0.0 && e->alpha[1] == 0.5 && V <= 64 && !e->knob("DMX_NO_SYM")) {
    if (V > 32 && !e->knob("DMX_FAST_NO_PROD")) {
      // 33..64 samples: the producer / consumer kernel over the printed entries (round 4; k_doublet_clsym<33,2> ran 2 wavefronts per SIMD)
      HIP_TRY(hipMemsetAsync(e->d_flag - kFlagHead, 0, (size_t)B + kFlagHead, e->stream));
      const size_t lds = (size_t)2 * 32 * 32 * 8 + (size_t)2 * 32 * 64 + (size_t)32 * 18 * 8 + 2 * 34 * 8 + 2 * 32 * (8 + 4 + 4) +
                         (size_t)32 * 12 * 4;
      DMX_LAUNCH(k2_fn, (k_doublet_clsp<3, true>), dim3((unsigned)B), dim3(kThreads), lds, e->stream, e->pv, e->nrd_width, e->d_rows, e->d_ids,
                         e->d_gp0, e->d_lut, e->d_alpha, e->d_sched, V, 0, e->d_grid, e->d_l00, e->d_flag);
      HIP_TRY(hipGetLastError());
      return launch_doublet_generic_w<true>(e);
    }
    // GT inputs, default grid {0, 0.5}, FAST: class table + the printed entries only, one wavefront per barcode
    const int D = V / 2 + 1, Q = 64 / V;
    // offsets per lane NED and slabs NS with Q * NED * NS >= D: as few slabs as 17 accumulators per lane allow (more do not fit
    // the register file beside the kernel's invariants at 3 wavefronts per SIMD), then the smallest NED
    int NS = (D + Q * 17 - 1) / (Q * 17);
    // ... except that a panel that would need two slabs of 17 runs as ONE slab of up to 33 at 2 wavefronts per SIMD (212
    // registers, no spills): every slab repeats phase 1 and the class table, which is two thirds of a 17-offset slab's time
    // (cfg4, 6 144 barcodes: 267 ms against 302 ms)
    if (NS == 2 && (D + Q - 1) / Q <= 33) NS = 1;
    if (const char* env = e->knob("DMX_CLSYM_NED")) NS = (D + Q * atoi(env) - 1) / (Q * atoi(env));   // kernel experiments only
    const int need = (D + Q * NS - 1) / (Q * NS);
    HIP_TRY(hipMemsetAsync(e->d_flag - kFlagHead, 0, (size_t)B + kFlagHead, e->stream));
    constexpr size_t cb = ((size_t)8 * 32 * 8 + 32 * 6 * 8 + 32 * 4 * 8 + 2 * 34 * 8 + 32 * 16 + 32 * 12 * 4 + 32 * 10 * 4 + 63) & ~(size_t)63;
    static_assert(cb % 64 == 0, "a barcode's LDS block keeps its class table 64-byte aligned (k_doublet_clsym ORs the class offset in)");
#define DMX_K2CS(NED, MINW)                                                                                            \
  DMX_LAUNCH(k2_fn, (k_doublet_clsym<NED, MINW>), dim3((unsigned)((B + 3) / 4), (unsigned)NS), dim3(kThreads), cb * 4, e->stream, e->pv, \
                     e->nrd_width, e->d_rows, e->d_idd, e->nwd2, e->d_gp0, e->d_lut, e->d_sched, V, e->d_grid, e->d_l00, e->d_flag)
    if (need <= 1) DMX_K2CS(1, 4); else if (need <= 2) DMX_K2CS(2, 4); else if (need <= 3) DMX_K2CS(3, 4); else if (need <= 5) DMX_K2CS(5, 4);
    else if (need <= 7) DMX_K2CS(7, 3); else if (need <= 9) DMX_K2CS(9, 3); else if (need <= 11) DMX_K2CS(11, 3);
    else if (need <= 13) DMX_K2CS(13, 3); else if (need <= 15) DMX_K2CS(15, 3); else if (need <= 17) DMX_K2CS(17, 3);
    else if (need <= 25) DMX_K2CS(25, 2); else DMX_K2CS(33, 2);   // one slab, 2 wavefronts per SIMD
#undef DMX_K2CS
    HIP_TRY(hipGetLastError());
    return launch_doublet_generic_w<true>(e);
  }
  if (use_cls) {
    const int VS = (V <= 32) ? ((V + 3) & ~3) : ((V + 15) & ~15);   // id row stride: a whole number of k-blocks
    size_t cb = (size_t)32 * 18 * 8 + (size_t)32 * 32 * 8 + 2 * 34 * 8 + 32 * (4 + 4 + 8) + (size_t)32 * 12 * 4 + (size_t)32 * VS;
    if (V > 32) cb += 32 * 16;                      // the pipelined form's second header buffer
    cb = (cb + 15) & ~(size_t)15;
    HIP_TRY(hipMemsetAsync(e->d_flag - kFlagHead, 0, (size_t)B + kFlagHead, e->stream));
    const dim3 blk(kThreads);
#define DMX_K2C(TPC, NK, ...)                                                                                        \
  DMX_LAUNCH(k2_fn, (k_doublet_cls<TPC, NK, ##__VA_ARGS__>), dim3((unsigned)((B + (kThreads / TPC) - 1) / (kThreads / TPC)), slabs_of(TPC, NK)), blk,   \
                     cb * (kThreads / TPC), e->stream, e->pv, e->nrd_width, e->d_rows, e->d_ids, e->d_gp0, e->d_lut,   \
                     e->d_alpha, e->d_sched, V, VS, e->d_grid, e->d_l00, e->d_flag)
    if (V <= 8) DMX_K2C(64, 1);
    else if (V <= 16) DMX_K2C(64, 4);
    else if (V <= 32) DMX_K2C(256, 4);
    else if (e->knob("DMX_CLS_MINW3")) DMX_K2C(256, 16, 3);    // kernel experiments only (36 spills: slower)
    else if (e->knob("DMX_CLS_NK8")) { if (atoi(e->knob("DMX_CLS_NK8")) == 3) DMX_K2C(256, 8, 3); else DMX_K2C(256, 8); }
    else if (V <= 64 && !e->knob("DMX_CLS_NO_PROD") && !e->knob("DMX_CLS_NO_UJ")) {
      // producer / consumer form (round 4): one barrier per tile, wavefront 0 builds the next tile's class table, 3 wavefronts per SIMD
      const size_t lds = (size_t)2 * 32 * 32 * 8 + (size_t)2 * 32 * 64 + (size_t)32 * 18 * 8 + 2 * 34 * 8 + 2 * 32 * (8 + 4 + 4) +
                         (size_t)32 * 12 * 4;
      DMX_LAUNCH(k2_fn, (k_doublet_clsp<3>), dim3((unsigned)B), dim3(kThreads), lds, e->stream, e->pv, e->nrd_width, e->d_rows, e->d_ids, e->d_gp0,
                         e->d_lut, e->d_alpha, e->d_sched, V, VS, e->d_grid, e->d_l00, e->d_flag);
    }
    else if (V <= 64 && !e->knob("DMX_CLS_NO_UJ")) DMX_K2C(256, 16, 2, true);   // uniform-j form (a wavefront's 16 samples j share their class per pair)
    else DMX_K2C(256, 16);
#undef DMX_K2C
    HIP_TRY(hipGetLastError());
    return launch_doublet_generic_w<true>(e);
  }
  const int GS = (V * 3 + 3) & ~3;               // LDS row stride of a genotype row (floats), 16-byte multiple
  const size_t cell_bytes = (size_t)32 * 18 * 8 + (size_t)32 * GS * 4 + 2 * 34 * 8 + 32 * (4 + 4 + 8);
  HIP_TRY(hipMemsetAsync(e->d_flag - kFlagHead, 0, (size_t)B + kFlagHead, e->stream));
  const dim3 block(kThreads);
  // k_doublet_a2's phase 1 on the default grid {0, 0.5}: finished values of pairs of up to three tabled reads from the table (round 6; shallow pileups only,
  // ensure_finals; DMX_A2_NO_FINALS=1: the read loop everywhere) — other grids have other mixing weights and keep the loop
  const double* pfin_a2 = nullptr;
  const bool pfin_a2_grid = A == 2 && e->alpha[0] == 0.0 && e->alpha[1] == 0.5;
  if (pfin_a2_grid && !e->knob("DMX_A2_NO_FINALS")) if (int rc = ensure_finals(e, &pfin_a2)) return rc;
  const double* pseed_a2 = nullptr;              // ... and the tiles that walk the loop walk it in the five-value form from the seed table (DMX_A2_NO_SEEDS=1: the nine-value loop)
  if (pfin_a2_grid && !e->knob("DMX_A2_NO_SEEDS")) if (int rc = ensure_seeds(e, &pseed_a2)) return rc;
#define DMX_K2A(TPC, NK)                                                                                             \
  do {                                                                                                                \
    if (e->geno_safe)                                                                                                 \
      DMX_LAUNCH(k2_fn, (k_doublet_a2<TPC, NK, 1, false, false>), dim3((unsigned)((B + (kThreads / TPC) - 1) / (kThreads / TPC)), slabs_of(TPC, NK)), block,  \
                         cell_bytes * (kThreads / TPC), e->stream, e->pv, e->nrd_width, e->d_g, e->d_gp0, e->d_lut,    \
                         e->d_alpha, e->d_sched, V, GS, e->d_grid, e->d_l00, e->d_flag, pfin_a2, pseed_a2);                               \
    else                                                                                                              \
  DMX_LAUNCH(k2_fn, (k_doublet_a2<TPC, NK>), dim3((unsigned)((B + (kThreads / TPC) - 1) / (kThreads / TPC)), slabs_of(TPC, NK)), block,  \
                     cell_bytes * (kThreads / TPC), e->stream, e->pv, e->nrd_width, e->d_g, e->d_gp0, e->d_lut,        \
                     e->d_alpha, e->d_sched, V, GS, e->d_grid, e->d_l00, e->d_flag, pfin_a2, pseed_a2);                                   \
  } while (0)
  if (fast_soft && e->alpha[0] == 0.0 && e->alpha[1] == 0.5 && (V <= 128 || sym_wide) && !e->knob("DMX_NO_SYM")) {
    // demuxlet's default grid {0, 0.5}: only the printed entries (singlet column + one evaluation per unordered pair)
    const double* pfin = nullptr;                 // phase 1's finished values of pairs of up to three tabled reads (DMX_SYM_NO_FINALS=1: the read loop everywhere)
    if (!e->knob("DMX_SYM_NO_FINALS")) if (int rc = ensure_finals(e, &pfin)) return rc;
    const double* pseed = nullptr;                // the state after a pair's first one or two reads (DMX_SYM_NO_SEEDS=1: the loop from its start)
    if (!e->knob("DMX_SYM_NO_SEEDS")) if (int rc = ensure_seeds(e, &pseed)) return rc;
#define DMX_K2S(TPC, VMAX, SUB, FIX)                                                                                  \
  do { e->k2_sym = true;                                                                                                                \
    constexpr int GSS_ = (3 * VMAX + 3) & ~3, VUS_ = (VMAX + 2) & ~1, TP_ = TPC >= 64 ? 32 : TPC / 2;                   \
    constexpr size_t cb_ = (size_t)TP_ * 6 * 8 + TP_ * 4 * 8 + 2 * (TP_ + 2) * 8 + TP_ * 16 + (size_t)(TPC == 32 ? 3 : (TPC >= 32 ? 2 : 1)) * SUB * GSS_ * 4 + (size_t)SUB * 3 * VUS_ * 8; \
    const size_t lds = cb_ * (kThreads / TPC);                                                                         \
    if (lds > 60 * 1024) {                                                                                             \
      HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_doublet_sym<TPC, VMAX, SUB, FIX>),                  \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                              \
      HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_doublet_sym<TPC, VMAX, SUB, FIX, 3, 0, false>),     \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                              \
    }                                                                                                                  \
    if (e->geno_safe)                                                                                                  \
      DMX_LAUNCH(k2_fn, (k_doublet_sym<TPC, VMAX, SUB, FIX, 3, 0, false>), dim3((unsigned)((B + (kThreads / TPC) - 1) / (kThreads / TPC))), \
                         block, lds, e->stream, e->pv, e->nrd_width, e->d_g, e->d_gp0, e->d_lut, e->d_sched, V | sym_flags,         \
                         e->d_grid, e->d_l00, e->d_flag, pfin, pseed);                                                              \
    else                                                                                                               \
    DMX_LAUNCH(k2_fn, (k_doublet_sym<TPC, VMAX, SUB, FIX>), dim3((unsigned)((B + (kThreads / TPC) - 1) / (kThreads / TPC))), \
                       block, lds, e->stream, e->pv, e->nrd_width, e->d_g, e->d_gp0, e->d_lut, e->d_sched, V | sym_flags,           \
                       e->d_grid, e->d_l00, e->d_flag, pfin, pseed);                                                                \
  } while (0)
#define DMX_K2SV(TPC, VMAX, SUB, FIX, MINW)                                                                            \
  do { e->k2_sym = true;                                                                                                                \
    constexpr int GSS_ = (3 * VMAX + 3) & ~3, VUS_ = (VMAX + 2) & ~1, TP_ = TPC >= 64 ? 32 : TPC / 2;                   \
    constexpr size_t cb_ = (size_t)TP_ * 6 * 8 + TP_ * 4 * 8 + 2 * (TP_ + 2) * 8 + TP_ * 16 + (size_t)(TPC == 32 ? 3 : (TPC >= 32 ? 2 : 1)) * SUB * GSS_ * 4 + (size_t)SUB * 3 * VUS_ * 8; \
    const size_t lds = cb_ * (kThreads / TPC);                                                                         \
    if (lds > 60 * 1024) HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_doublet_sym<TPC, VMAX, SUB, FIX, MINW>), \
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));          \
    DMX_LAUNCH(k2_fn, (k_doublet_sym<TPC, VMAX, SUB, FIX, MINW>), dim3((unsigned)((B + (kThreads / TPC) - 1) / (kThreads / TPC))), \
                       block, lds, e->stream, e->pv, e->nrd_width, e->d_g, e->d_gp0, e->d_lut, e->d_sched, V | sym_flags,           \
                       e->d_grid, e->d_l00, e->d_flag, pfin, pseed);                                                                \
  } while (0)
    const int32_t sym_flags = (e->knob("DMX_SYM_NO_DMA") ? (1 << 16) : 0) | (e->knob("DMX_SYM_ABLATE_P1") ? (1 << 17) : 0) |
                              (e->knob("DMX_SYM_ABLATE_P2") ? (1 << 18) : 0) | (e->knob("DMX_SYM_ABLATE_U") ? (1 << 19) : 0) |
                              (e->knob("DMX_SYM_ABLATE_RD") ? (1 << 20) : 0) | (e->knob("DMX_SYM_ABLATE_00") ? (1 << 21) : 0) | (e->knob("DMX_SYM_NO_PRODUCT") ? (1 << 22) : 0) |
                              (e->knob("DMX_SYM_NO_PIPE") ? (1 << 23) : 0) | (e->knob("DMX_SYM_WAIT_ALL") ? (1 << 24) : 0);       // kernel experiments only (the ABLATE bits do something in -DDMX_SYM_ABLATIONS=1 builds only)
    const bool wide_cells = e->knob("DMX_SYM_ONE_CELL_PER_WAVE") != nullptr;   // kernel experiments only
    if (V <= 8 && !wide_cells) { if (16 % V == 0) DMX_K2S(16, 8, 4, true); else DMX_K2S(16, 8, 4, false); }        // four barcodes per wavefront
    else if (V <= 16 && !wide_cells) { if (V == 16) DMX_K2S(32, 16, 4, true); else DMX_K2S(32, 16, 4, false); }   // two
    else if (V <= 8) { if (64 % V == 0) DMX_K2S(64, 8, 4, true); else DMX_K2S(64, 8, 4, false); }
    else if (V <= 17) { if (V == 16) DMX_K2S(64, 17, 4, true); else DMX_K2S(64, 17, 4, false); }
    else if (V <= 23) DMX_K2S(64, 23, 4, false);
    else if (V <= 27) DMX_K2S(64, 27, 4, false);
    else if (V <= 32) {
      if (V == 32) {
        const int var = e->knob("DMX_SYM_VARIANT") ? atoi(e->knob("DMX_SYM_VARIANT")) : 0;      // kernel experiments only
        if (var == 1) DMX_K2SV(64, 32, 2, true, 4); else if (var == 2) DMX_K2SV(64, 32, 4, true, 4); else if (var == 3) DMX_K2SV(64, 32, 8, true, 3);
        else if (var == 4) DMX_K2SV(64, 32, 2, true, 3); else if (var == 5) DMX_K2SV(64, 32, 8, true, 2); else DMX_K2S(64, 32, 4, true);
      } else DMX_K2S(64, 32, 4, false);
    }
    else if (V <= 48) DMX_K2S(256, 48, 8, false);
    else if (V <= 64) { if (V == 64) DMX_K2S(256, 64, 8, true); else DMX_K2S(256, 64, 8, false); }
    else {
      // 64 < V <= 512: the entry list (V (V/2 + 1) + V, up to 132 096) in slabs of 9 entries per lane
      const unsigned ns = (unsigned)((V * (V / 2 + 1) + V + 256 * 9 - 1) / (256 * 9));
#define DMX_K2SS(VMAX, SUB, FIX)                                                                                       \
  do { e->k2_sym = true;                                                                                                                \
    constexpr int GSS_ = (3 * VMAX + 3) & ~3, VUS_ = (VMAX + 2) & ~1;                                                  \
    constexpr size_t cb_ = (size_t)32 * 6 * 8 + 32 * 4 * 8 + 2 * 34 * 8 + 32 * 16 + (size_t)2 * SUB * GSS_ * 4 + (size_t)SUB * 3 * VUS_ * 8; \
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_doublet_sym<256, VMAX, SUB, FIX, 3, 9>),              \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)cb_));                               \
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_doublet_sym<256, VMAX, SUB, FIX, 3, 9, false>),       \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)cb_));                               \
    if (e->geno_safe)                                                                                                  \
      DMX_LAUNCH(k2_fn, (k_doublet_sym<256, VMAX, SUB, FIX, 3, 9, false>), dim3((unsigned)B, ns), block, cb_, e->stream, e->pv, \
                         e->nrd_width, e->d_g, e->d_gp0, e->d_lut, e->d_sched, V | sym_flags, e->d_grid, e->d_l00, e->d_flag, pfin, pseed);      \
    else                                                                                                               \
    DMX_LAUNCH(k2_fn, (k_doublet_sym<256, VMAX, SUB, FIX, 3, 9>), dim3((unsigned)B, ns), block, cb_, e->stream, e->pv, \
                       e->nrd_width, e->d_g, e->d_gp0, e->d_lut, e->d_sched, V | sym_flags, e->d_grid, e->d_l00, e->d_flag, pfin, pseed);        \
  } while (0)
      if (V <= 96) DMX_K2SS(96, 8, false); else if (V == 128) DMX_K2SS(128, 8, true); else if (V < 128) DMX_K2SS(128, 8, false);
      else if (V <= 192) DMX_K2SS(192, 4, false); else if (V <= 256) DMX_K2SS(256, 4, false);        // (sub-tiles of 4 pairs: 40 / 53 KB of LDS per workgroup)
      else if (V <= 384) DMX_K2SS(384, 2, false); else DMX_K2SS(512, 2, false);        // (round 4: 257..512 samples, sub-tiles of 2 pairs: 40 / 53 KB)
#undef DMX_K2SS
    }
#undef DMX_K2SV
#undef DMX_K2S
    HIP_TRY(hipGetLastError());
    return launch_doublet_generic_w<true>(e);
  }
  if (fast_soft && V <= 128) {     // (wider panels that are not k_doublet_sym's run the STRICT kernel below: bit-exact, inside FAST's contract)
    // one-cell-per-workgroup panels share u through LDS (cfg3 1.33x); one-wavefront cells (V <= 16) form it in registers
    const size_t fast_bytes = cell_bytes + (V > 16 ? (size_t)8 * 2 * V * 32 : 0);
#define DMX_K2F(TPC, NK)                                                                                             \
  do {                                                                                                               \
    const size_t lds = fast_bytes * (kThreads / TPC);                                                                \
    if (lds > 60 * 1024) HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_doublet_a2f<TPC, NK>),          \
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));          \
    DMX_LAUNCH(k2_fn, (k_doublet_a2f<TPC, NK>), dim3((unsigned)((B + (kThreads / TPC) - 1) / (kThreads / TPC)), slabs_of(TPC, NK)), \
                       block, lds, e->stream, e->pv, e->nrd_width, e->d_g, e->d_gp0, e->d_lut, e->d_alpha, e->d_sched, V, GS,           \
                       e->d_grid, e->d_l00, e->d_flag);                                                              \
  } while (0)
    if (V <= 8) DMX_K2F(64, 1);
    else if (V <= 16) DMX_K2F(64, 4);
    else if (V <= 32) DMX_K2F(256, 4);
    else DMX_K2F(256, 16);
#undef DMX_K2F
    HIP_TRY(hipGetLastError());
    return launch_doublet_generic_w<true>(e);
  }
  if (V == 32 && pfin_a2_grid && e->geno_safe && !e->knob("DMX_A2_SYM") && !e->knob("DMX_A2_NO_SYMU") && !e->knob("DMX_A2_NO_GD") && !e->knob("DMX_A2_MINW1")) {
    // 32 soft-field samples on the default grid (cfg3, the headline): k_doublet_a2's kernel over UNORDERED pairs — a thread owns [j][k] and [k][j] and forms
    // the products they share once — and the 64 diagonal accumulators of a barcode on one wavefront behind it (DMX_A2_NO_SYMU=1: k_doublet_a2)
    const size_t cbd = (size_t)32 * 18 * 8 + (size_t)32 * GS * 8 + 2 * 34 * 8 + 32 * (4 + 4 + 8);
    DMX_LAUNCH(k2_fn, (k_doublet_a2u<4>), dim3((unsigned)B, 1), block, cbd, e->stream, e->pv, e->nrd_width, e->d_g,
                       e->d_gp0, e->d_lut, e->d_alpha, e->d_sched, V, GS, e->d_grid, e->d_l00, e->d_flag, pfin_a2, pseed_a2);
    if (e->knob("DMX_A2U_DIAG_WAVE"))             // kernel experiments only: the diagonal on one wavefront per barcode (the first form)
      hipLaunchKernelGGL((k_doublet_a2s<4, 4, 0>), dim3((unsigned)((B + 3) / 4), 1), block, 0, e->stream, e->pv, e->nrd_width, e->d_g, e->d_gp0, e->d_lut, e->d_sched,
                         e->d_grid, e->d_l00, pfin_a2);
    else
      hipLaunchKernelGGL((k_doublet_diag<5>), dim3((unsigned)((B + 7) / 8)), block, 0, e->stream, e->pv, e->nrd_width, e->d_g, e->d_lut, e->d_sched, e->d_grid, pfin_a2, pseed_a2);
  }
  else if (V == 32 && pfin_a2_grid && e->geno_safe && e->knob("DMX_A2_SYM")) {
    // 32 soft-field samples on the default grid (cfg3, the headline): symmetric ownership — one lane owns [j][k] and [k][j], their shared products once.
    // An EXPERIMENT (DMX_A2_SYM=1), bit-identical, 9 % fewer instructions (967 FP64 + 212 others against 1 040 + 260 per covered pair and barcode) and
    // about as fast: the 17 accumulators per lane do not fit four wavefronts per SIMD (128 registers: 18 spill accesses per pair in the loop, 3 139 ms),
    // and at three it runs alone 1 206 ms (sub-tiles of 8 pairs, DMX_A2S_SUB8=1: 1 177-1 191) against k_doublet_a2's 1 214-1 223 on the same boxes, the STEP
    // 1 229-1 235 against 1 237-1 250 — K1 beside it costs it more than it costs k_doublet_a2.  Not worth a second headline kernel.
    const dim3 grds((unsigned)((B + 3) / 4), 2);
    if (e->knob("DMX_A2S_SUB8"))                 // kernel experiments only
      DMX_LAUNCH(k2_fn, (k_doublet_a2s<8, 3>), grds, block, 0, e->stream, e->pv, e->nrd_width, e->d_g, e->d_gp0, e->d_lut, e->d_sched, e->d_grid, e->d_l00, pfin_a2);
    else if (!e->knob("DMX_A2S_MINW4"))          // (three wavefronts per SIMD unless the spilling four-wavefront form is asked for)
      DMX_LAUNCH(k2_fn, (k_doublet_a2s<4, 3>), grds, block, 0, e->stream, e->pv, e->nrd_width, e->d_g, e->d_gp0, e->d_lut, e->d_sched, e->d_grid, e->d_l00, pfin_a2);
    else
      DMX_LAUNCH(k2_fn, (k_doublet_a2s<4, 4>), grds, block, 0, e->stream, e->pv, e->nrd_width, e->d_g, e->d_gp0, e->d_lut, e->d_sched, e->d_grid, e->d_l00, pfin_a2);
  }
  else if (V == 16 && pfin_a2_grid && e->geno_safe && !e->knob("DMX_A2_NO_SYMU")) {
    // 16 soft-field samples on the default grid (cfg5): the same — unordered pairs on k_doublet_a2<64,4>'s kernel, the diagonal on 16 lanes per barcode behind it
    DMX_LAUNCH(k2_fn, (k_doublet_a2u16), dim3((unsigned)((B + 3) / 4), 1), block, cell_bytes * 4, e->stream, e->pv, e->nrd_width, e->d_g,
                       e->d_gp0, e->d_lut, e->d_alpha, e->d_sched, V, GS, e->d_grid, e->d_l00, e->d_flag, pfin_a2, pseed_a2);
    hipLaunchKernelGGL((k_doublet_diag<5, 16>), dim3((unsigned)((B + 15) / 16)), block, 0, e->stream, e->pv, e->nrd_width, e->d_g, e->d_lut, e->d_sched, e->d_grid, pfin_a2, pseed_a2);
  }
  else if (V <= 8) DMX_K2A(64, 1);
  else if (V <= 16) DMX_K2A(64, 4);
  else if (V <= 32) {
    // 4 wavefronts per SIMD (128 VGPRs, a few spills outside the hot loop) measured 2.8 % faster than 3 (158 VGPRs) on cfg3
    if (!e->knob("DMX_A2_NO_GD") && !e->knob("DMX_A2_MINW1")) {   // rows widened to binary64 at staging: 1-2.5 % (cfg3, 5 000 barcodes: 741-753 vs 760 ms)
      const size_t cbd = (size_t)32 * 18 * 8 + (size_t)32 * GS * 8 + 2 * 34 * 8 + 32 * (4 + 4 + 8);
      if (e->geno_safe)
        DMX_LAUNCH(k2_fn, (k_doublet_a2<256, 4, 4, true, false>), dim3((unsigned)B, slabs_of(256, 4)), block, cbd, e->stream, e->pv, e->nrd_width, e->d_g,
                           e->d_gp0, e->d_lut, e->d_alpha, e->d_sched, V, GS, e->d_grid, e->d_l00, e->d_flag, pfin_a2, pseed_a2);
      else
      DMX_LAUNCH(k2_fn, (k_doublet_a2<256, 4, 4, true>), dim3((unsigned)B, slabs_of(256, 4)), block, cbd, e->stream, e->pv, e->nrd_width, e->d_g,
                         e->d_gp0, e->d_lut, e->d_alpha, e->d_sched, V, GS, e->d_grid, e->d_l00, e->d_flag, pfin_a2, pseed_a2);
    } else if (!e->knob("DMX_A2_MINW1"))
      DMX_LAUNCH(k2_fn, (k_doublet_a2<256, 4, 4>), dim3((unsigned)B, slabs_of(256, 4)), block, cell_bytes, e->stream, e->pv, e->nrd_width, e->d_g,
                         e->d_gp0, e->d_lut, e->d_alpha, e->d_sched, V, GS, e->d_grid, e->d_l00, e->d_flag, pfin_a2, pseed_a2);
    else DMX_K2A(256, 4);
  }
  else if (V <= 128) DMX_K2A(256, 16);            // <= 55 KB of LDS
  else {
    // 129..1024 samples (round 4; the generic kernel before): the same kernel with the tile as long as leaves TWO workgroups per CU
    // (32 pairs up to 181 samples, 16 up to 377, 8 up to 770), else the longest that fits one workgroup's 160 KB; above 64 KB
    // the limit is raised explicitly.  2.2-3.6x the generic kernel's rate, bit-identical to it (tools/probe_wide_strict.py).
    auto bytes_of = [&](int tp) { return (size_t)tp * 18 * 8 + (size_t)tp * GS * 4 + 2 * (size_t)(tp + 2) * 8 + (size_t)tp * (4 + 4 + 8); };
    constexpr size_t kStatic = sizeof(double) * kTab2 + 2 * 18 * sizeof(double), kTwo = 80 * 1024, kOne = 160 * 1024;   // (the kernel's static LDS: s_tab, s_w)
    int tp = 0;
    for (int c : {32, 16, 8}) if (!tp && bytes_of(c) + kStatic <= kTwo) tp = c;
    for (int c : {32, 16, 8}) if (!tp && bytes_of(c) + kStatic <= kOne) tp = c;
    if (const char* env = e->knob("DMX_A2_TP")) { const int c = atoi(env); if ((c == 32 || c == 16 || c == 8) && bytes_of(c) + kStatic <= kOne) tp = c; }   // kernel experiments only
    if (!tp) return set_error(DMX_ERR_ARG, "run_doublet: V = %d exceeds k_doublet_a2's LDS budget", (int)V);
    const size_t lds = bytes_of(tp);
#define DMX_K2AW(TPP)                                                                                                 \
  do {                                                                                                                \
    if (e->geno_safe) {                                                                                               \
      if (lds > 60 * 1024) HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_doublet_a2<256, 16, 1, false, false, TPP>),  \
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));        \
      DMX_LAUNCH(k2_fn, (k_doublet_a2<256, 16, 1, false, false, TPP>), dim3((unsigned)B, slabs_of(256, 16)), block, lds, e->stream, e->pv,  \
                         e->nrd_width, e->d_g, e->d_gp0, e->d_lut, e->d_alpha, e->d_sched, V, GS, e->d_grid, e->d_l00, e->d_flag, pfin_a2, pseed_a2);  \
    } else {                                                                                                          \
      if (lds > 60 * 1024) HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_doublet_a2<256, 16, 1, false, true, TPP>),   \
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));        \
      DMX_LAUNCH(k2_fn, (k_doublet_a2<256, 16, 1, false, true, TPP>), dim3((unsigned)B, slabs_of(256, 16)), block, lds, e->stream, e->pv,   \
                         e->nrd_width, e->d_g, e->d_gp0, e->d_lut, e->d_alpha, e->d_sched, V, GS, e->d_grid, e->d_l00, e->d_flag, pfin_a2, pseed_a2);  \
    }                                                                                                                 \
  } while (0)
    if (tp == 32) DMX_K2AW(32); else if (tp == 16) DMX_K2AW(16); else DMX_K2AW(8);
#undef DMX_K2AW
  }
#undef DMX_K2A
  HIP_TRY(hipGetLastError());
  return launch_doublet_generic_w<true>(e);
}

}  // namespace

namespace {
int launch_certify(dmx_engine* e) {
  const int32_t B = e->pv.B;
  // sparse pileups over a matrix beyond the L2: one launch per SNP block (the table of the blocked K1 walk), state parked in between
  // — OFF by default: measured at cfg5 it cuts K3b's L2-side traffic (profiles/r04_certify_blocks.txt) but costs time (10.15 against 9.28 ms: the kernel is
  // bound by its own instruction stream, and 25 launches with their partial tiles cost more than the L2 hits save); DMX_CERTIFY_BLOCKS=1 turns it on,
  // and forced small blocks (DMX_K1_BLOCK_BYTES) always use it so that the tests cover the parked-state path
  const int64_t* blk = (e->pv.pair_snp && e->blk_n > 1 && (e->knob("DMX_CERTIFY_BLOCKS") || e->knob("DMX_K1_BLOCK_BYTES"))) ? e->d_blk : nullptr;
  const int bstride = (blk && e->knob("DMX_CERTIFY_BLOCK_STRIDE")) ? std::max(1, atoi(e->knob("DMX_CERTIFY_BLOCK_STRIDE"))) : 1;   // table blocks per launch
  const int n_launch = blk ? (e->blk_n + bstride - 1) / bstride : 1;
  double* park = nullptr;
  if (blk) {
    if (int rc = ensure_dev((void**)&e->d_park, &e->park_cap, sizeof(double) * kPark * (size_t)std::max(B, 1))) return rc;
    park = e->d_park;
  }
  const float* gT = (!e->pv.pair_snp && e->have_gT && !e->knob("DMX_CERTIFY_NO_GT")) ? e->d_gT : nullptr;   // dense pileups: SNP-minor columns
#define DMX_K3B(MINW_, FIVE_, DENSE_)                                                                                  \
  for (int bi = 0; bi < n_launch; ++bi)                                                                                 \
    DMX_LAUNCH(k3b_fn, (k_certify<MINW_, FIVE_, DENSE_>), dim3((unsigned)((B + 3) / 4)), dim3(kThreads), 0, e->stream, e->pv, e->nrd_width, e->d_g, gT, \
                       e->d_lut, e->d_alpha, e->V, e->d_sum, blk, (bi * bstride) | (bstride << 16), e->blk_n, park, cseed, cfin)
  const bool five = e->alpha[0] == 0.0;
  const double* cseed = nullptr;                  // the first one or two reads of a pair from a table (DMX_CERTIFY_NO_SEEDS=1: the whole loop)
  if (five && !e->knob("DMX_CERTIFY_NO_SEEDS")) if (int rc = ensure_seeds(e, &cseed)) return rc;
  const double* cfin = nullptr;                   // final values of pairs of up to three tabled reads (DMX_CERTIFY_NO_FINALS=1 / DMX_CERTIFY_NO_SEEDS=1: none)
  if (cseed && !e->knob("DMX_CERTIFY_NO_FINALS")) if (int rc = ensure_finals(e, &cfin)) return rc;
  if (e->knob("DMX_CERTIFY_MINW3")) { if (gT) DMX_K3B(3, false, true); else DMX_K3B(3, false, false); }      // kernel experiments only
  else if (gT) { if (five) DMX_K3B(4, true, true); else DMX_K3B(4, false, true); }
  else { if (five) DMX_K3B(4, true, false); else DMX_K3B(4, false, false); }
#undef DMX_K3B
  HIP_TRY(hipGetLastError());
  return DMX_OK;
}
}  // namespace

extern "C" int dmx_engine_run_singlet(dmx_engine* e) {
  if (!e) return set_error(DMX_ERR_ARG, "dmx_engine_run_singlet: null engine");
  if (!e->have_pileup || !e->d_g) return set_error(DMX_ERR_STATE, "dmx_engine_run_singlet: set genotypes and pileup first");
  HIP_TRY(hipSetDevice(e->device));
  if (e->pv.B == 0) { e->have_sing = true; return DMX_OK; }
  hipEvent_t* rs = e->ring_s[e->n_ring_s % dmx_engine::kRing];
  e->k1_fn = nullptr;                              // dmx_engine_kernel_names: what THIS call launches, never an earlier call's choice
  HIP_TRY(hipEventRecord(e->ev[2], e->stream));
  HIP_TRY(hipEventRecord(rs[0], e->stream));
  if (int rc = launch_singlet(e)) return rc;
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventRecord(e->ev[3], e->stream));
  HIP_TRY(hipEventRecord(rs[1], e->stream));
  ++e->n_ring_s;
  e->timed[1] = true; e->have_sing = true; e->k1_placement = 0;
  return DMX_OK;
}

namespace { int run_doublet_impl(dmx_engine* e, bool with_singlet); }

extern "C" int dmx_engine_run_doublet(dmx_engine* e) { return run_doublet_impl(e, false); }

// K1 and K2 -> K3 -> K3b of the staged pileup in one call.  K3 reads nothing of K1's, so K1 runs BESIDE K2 on a stream of the lowest
// priority: K2's workgroups are dispatched first and K1 fills the slots K2 leaves free — above all during K2's last, partly filled
// round (cfg3 FAST: 10 000 one-barcode wavefronts on 3 072 slots).  Fork and join are events on the engine's stream, so that for the
// caller everything is ordered on that stream exactly as after run_singlet + run_doublet; the results are the same bits.
namespace {
// the doublet kernel launch_doublet picked is k_doublet_clsp (GT classes, grid {0, 0.5}, 33..64 samples): what launch_doublet itself recorded
bool k2_is_clsp(const dmx_engine* e) {
  return e->k2_fn == reinterpret_cast<const void*>(&k_doublet_clsp<3, false>) || e->k2_fn == reinterpret_cast<const void*>(&k_doublet_clsp<3, true>);
}
}  // namespace

extern "C" int dmx_engine_run(dmx_engine* e) {
  if (!e) return set_error(DMX_ERR_ARG, "dmx_engine_run: null engine");
  // K1 beside K2 pays where K2 leaves slots free.  k_doublet_clsp does not (three wavefronts per SIMD, all of them issuing: measured on the
  // cfg4 shard, K1 beside it cost what it saved), so there K1 starts when K2 has finished and runs beside K3 + K3b (run_doublet_impl; cfg4 shard
  // 573 -> 566 ms per step).  DMX_FORCE_OVERLAP=1: beside K2 anyway; DMX_K1_FIRST=1: K1, then K2, K3, K3b, one after the other.
  const bool serial = e->knob("DMX_NO_OVERLAP") || e->knob("DMX_K1_FIRST");
  if (e->V < 2 || e->A < 2 || serial) {
    if (int rc = dmx_engine_run_singlet(e)) return rc;
    const int rc = (e->V < 2 || e->A < 2) ? DMX_OK : dmx_engine_run_doublet(e);
    e->k1_placement = 0;
    return rc;
  }
  return run_doublet_impl(e, true);
}

namespace {
int run_doublet_impl(dmx_engine* e, bool with_singlet) {
  if (!e) return set_error(DMX_ERR_ARG, "dmx_engine_run_doublet: null engine");
  if (!e->have_pileup || !e->d_g) return set_error(DMX_ERR_STATE, "dmx_engine_run_doublet: set genotypes and pileup first");
  if (e->V < 2 || e->A < 2) return set_error(DMX_ERR_ARG, "dmx_engine_run_doublet: needs >= 2 samples and >= 2 alphas (got %d, %d)", e->V, e->A);
  HIP_TRY(hipSetDevice(e->device));
  const int32_t B = e->pv.B;
  const size_t nAB = (size_t)e->V * e->V * e->A;
  if (!e->d_grid || e->grid_cap < B) {
    if (e->d_grid) { (void)hipFree(e->d_grid); (void)hipFree(e->d_l00); (void)hipFree(e->d_sum); if (e->d_flag) (void)hipFree(e->d_flag - kFlagHead); (void)hipFree(e->d_sing); }
    e->d_grid = e->d_l00 = e->d_sing = nullptr; e->d_sum = nullptr; e->d_flag = nullptr;
    const size_t cap = (size_t)e->out_cap;       // the singlet buffers' capacity (>= B)
    HIP_TRY(hipMalloc((void**)&e->d_grid, std::max<size_t>(sizeof(double) * nAB * cap, 16)));
    HIP_TRY(hipMalloc((void**)&e->d_l00, std::max<size_t>(sizeof(double) * (size_t)e->A * cap, 16)));
    HIP_TRY(hipMalloc((void**)&e->d_sum, std::max<size_t>(sizeof(dmx_cell_summary) * cap, 16)));
    { uint8_t* fb = nullptr; HIP_TRY(hipMalloc((void**)&fb, cap + 2 * kFlagHead)); e->d_flag = fb + kFlagHead; }
    HIP_TRY(hipMalloc((void**)&e->d_sing, std::max<size_t>(sizeof(double) * cap * e->V, 16)));
    e->grid_cap = (int32_t)cap;
  }
  e->k2_fn = e->k3b_fn = nullptr;                  // (dmx_engine_kernel_names reports this call's launches; a run without K3b leaves `certify` empty)
  if (with_singlet) e->k1_fn = nullptr;
  if (B == 0) { e->have_grid = true; if (with_singlet) e->have_sing = true; return DMX_OK; }
  hipEvent_t* rd = e->ring_d[e->n_ring_d % dmx_engine::kRing];
  if (with_singlet) HIP_TRY(hipEventRecord(e->ev_fork, e->stream));       // (K1 must not start before what precedes this call on the stream)
  HIP_TRY(hipEventRecord(e->ev[4], e->stream));
  HIP_TRY(hipEventRecord(rd[0], e->stream));
  if (int rc = launch_doublet(e)) return rc;
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventRecord(e->ev[5], e->stream));
  HIP_TRY(hipEventRecord(rd[1], e->stream));
  if (with_singlet) {                            // K1, enqueued after K2, on the low-priority stream
    hipStream_t main_stream = e->stream;
    hipEvent_t* rs = e->ring_s[e->n_ring_s % dmx_engine::kRing];
    // K1 beside K2 — or, where K2 leaves it no room (k_doublet_clsp: three issuing wavefronts per SIMD), beside K3 + K3b, which like K1
    // issue on about half of their cycles: K1 then starts when K2 has finished (DMX_FORCE_OVERLAP=1: beside K2 anyway)
    // The same holds for k_doublet_sym under ONE long K1 launch (round 5): its workgroups fill the CU's LDS three deep (46-53 KB each); a K1
    // workgroup (29 KB, alive for the whole launch) that takes the place of a retired one leaves no room for the next K2 workgroup, and the CU runs
    // two of K2's three until K1 ends — cfg3 FAST: K2 218 ms alone, 244 with a K1 beside it that takes 20 ms alone; step 260.5 -> 251.1 ms with K1
    // after K2.  A K1 walked in SNP blocks (sparse pileups over a matrix beyond the L2: a launch of short workgroups per block) does fit in
    // between: cfg5 FAST 49.0 ms beside K2, 53.2 after it.  STRICT's k_doublet_a2 keeps K1 beside it (cfg3 1 221.5 against 1 227.8, cfg5 144.2 / 150.2).
    const bool k1_blocked = e->pv.pair_snp != nullptr && e->blk_n > 1;
    const bool after_k2 = (k2_is_clsp(e) || (e->k2_sym && !k1_blocked) || e->knob("DMX_K1_AFTER_K2")) && !e->knob("DMX_FORCE_OVERLAP");
    e->k1_placement = after_k2 ? 2 : 1;
    HIP_TRY(hipStreamWaitEvent(e->k1_stream, after_k2 ? e->ev[5] : e->ev_fork, 0));
    e->stream = e->k1_stream;
    int rc = DMX_OK;
    hipError_t he = hipEventRecord(e->ev[2], e->stream);
    if (he == hipSuccess) he = hipEventRecord(rs[0], e->stream);
    if (he == hipSuccess) rc = launch_singlet(e);
    if (he == hipSuccess && rc == DMX_OK) he = hipGetLastError();
    if (he == hipSuccess && rc == DMX_OK) he = hipEventRecord(e->ev[3], e->stream);
    if (he == hipSuccess && rc == DMX_OK) he = hipEventRecord(rs[1], e->stream);
    if (he == hipSuccess && rc == DMX_OK) he = hipEventRecord(e->ev_k1_done, e->stream);
    e->stream = main_stream;
    if (rc != DMX_OK) return rc;
    HIP_TRY(he);
    ++e->n_ring_s;
    e->timed[1] = true; e->have_sing = true;
  }
  hipLaunchKernelGGL(k_reduce, dim3((unsigned)B), dim3(kThreads), 0, e->stream, e->d_grid, e->d_l00, e->pv.cell_pair_off,
                     e->d_alpha, e->V, e->A, e->prior, e->d_sum, e->d_sing);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventRecord(rd[2], e->stream));
  const bool certify = e->certify && e->A == 2 && e->alpha[1] == 0.5;
  if (certify) if (int rc = launch_certify(e)) return rc;
  HIP_TRY(hipEventRecord(e->ev[6], e->stream));
  HIP_TRY(hipEventRecord(rd[3], e->stream));
  e->ring_certified[e->n_ring_d % dmx_engine::kRing] = certify;
  ++e->n_ring_d;
  e->timed[2] = e->timed[3] = true; e->have_grid = true;
  if (with_singlet) HIP_TRY(hipStreamWaitEvent(e->stream, e->ev_k1_done, 0));   // join: whatever follows on the stream sees K1's results
  return DMX_OK;
}
}  // namespace

extern "C" int dmx_engine_sync(dmx_engine* e) {
  if (!e) return set_error(DMX_ERR_ARG, "dmx_engine_sync: null engine");
  HIP_TRY(hipSetDevice(e->device));
  HIP_TRY(hipStreamSynchronize(e->stream));
  return DMX_OK;
}

extern "C" int dmx_engine_get_singlet(dmx_engine* e, double* llks, double* llk0s) {
  if (!e) return set_error(DMX_ERR_ARG, "dmx_engine_get_singlet: null engine");
  if (!e->have_sing) return set_error(DMX_ERR_STATE, "dmx_engine_get_singlet: run_singlet has not been called");
  HIP_TRY(hipSetDevice(e->device));
  const size_t B = (size_t)e->pv.B;
  if (llks && B) HIP_TRY(hipMemcpyAsync(llks, e->d_llks, sizeof(double) * B * e->V, hipMemcpyDeviceToHost, e->stream));
  if (llk0s && B) HIP_TRY(hipMemcpyAsync(llk0s, e->d_llk0s, sizeof(double) * B, hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  return DMX_OK;
}

extern "C" int dmx_engine_get_doublet(dmx_engine* e, double* llksAB, double* llks00, dmx_cell_summary* summary) {
  if (!e) return set_error(DMX_ERR_ARG, "dmx_engine_get_doublet: null engine");
  if (!e->have_grid) return set_error(DMX_ERR_STATE, "dmx_engine_get_doublet: run_doublet has not been called");
  HIP_TRY(hipSetDevice(e->device));
  const size_t B = (size_t)e->pv.B, nAB = (size_t)e->V * e->V * e->A;
  if (llksAB && B) HIP_TRY(hipMemcpyAsync(llksAB, e->d_grid, sizeof(double) * B * nAB, hipMemcpyDeviceToHost, e->stream));
  if (llks00 && B) HIP_TRY(hipMemcpyAsync(llks00, e->d_l00, sizeof(double) * B * e->A, hipMemcpyDeviceToHost, e->stream));
  if (summary && B) HIP_TRY(hipMemcpyAsync(summary, e->d_sum, sizeof(dmx_cell_summary) * B, hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  return DMX_OK;
}

extern "C" int dmx_engine_get_sing(dmx_engine* e, double* sing) {
  if (!e || !sing) return set_error(DMX_ERR_ARG, "dmx_engine_get_sing: null argument");
  if (!e->have_grid) return set_error(DMX_ERR_STATE, "dmx_engine_get_sing: run_doublet has not been called");
  HIP_TRY(hipSetDevice(e->device));
  const size_t B = (size_t)e->pv.B;
  if (B) HIP_TRY(hipMemcpyAsync(sing, e->d_sing, sizeof(double) * B * e->V, hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  return DMX_OK;
}

extern "C" int dmx_engine_get_cell_grids(dmx_engine* e, const int32_t* cells, int32_t n, double* out) {
  if (!e || n < 0 || (n && (!cells || !out))) return set_error(DMX_ERR_ARG, "dmx_engine_get_cell_grids: bad arguments");
  if (!e->have_grid) return set_error(DMX_ERR_STATE, "dmx_engine_get_cell_grids: run_doublet has not been called");
  HIP_TRY(hipSetDevice(e->device));
  const size_t nAB = (size_t)e->V * e->V * e->A;
  for (int32_t i = 0; i < n; ++i)                  // every id before the first copy: an error must not leave copies into the caller's buffer in flight
    if (cells[i] < 0 || cells[i] >= e->pv.B) return set_error(DMX_ERR_ARG, "dmx_engine_get_cell_grids: cell %d out of range (0..%d)", cells[i], e->pv.B - 1);
  hipError_t he = hipSuccess;
  for (int32_t i = 0; i < n && he == hipSuccess; ++i)
    he = hipMemcpyAsync(out + (size_t)i * nAB, e->d_grid + (size_t)cells[i] * nAB, sizeof(double) * nAB, hipMemcpyDeviceToHost, e->stream);
  const hipError_t hs = hipStreamSynchronize(e->stream);                  // also after a failed copy: nothing stays in flight into `out`
  HIP_TRY(he);
  HIP_TRY(hs);
  return DMX_OK;
}

extern "C" int dmx_engine_device_view(dmx_engine* e, dmx_device_view* out) {
  if (!e || !out) return set_error(DMX_ERR_ARG, "dmx_engine_device_view: null argument");
  out->llks = e->d_llks; out->llk0s = e->d_llk0s; out->llksAB = e->d_grid; out->llks00 = e->d_l00; out->summary = e->d_sum;
  out->gp0s = e->d_gp0; out->sing = e->d_sing;
  return DMX_OK;
}

extern "C" int dmx_engine_last_kernel_times(dmx_engine* e, dmx_kernel_times* out) {
  if (!e || !out) return set_error(DMX_ERR_ARG, "dmx_engine_last_kernel_times: null argument");
  HIP_TRY(hipSetDevice(e->device));
  HIP_TRY(hipStreamSynchronize(e->stream));
  out->gp0_ms = out->singlet_ms = out->doublet_ms = out->reduce_ms = 0.f;
  if (e->timed[0]) HIP_TRY(hipEventElapsedTime(&out->gp0_ms, e->ev[0], e->ev[1]));
  if (e->timed[1]) HIP_TRY(hipEventElapsedTime(&out->singlet_ms, e->ev[2], e->ev[3]));
  if (e->timed[2]) HIP_TRY(hipEventElapsedTime(&out->doublet_ms, e->ev[4], e->ev[5]));
  if (e->timed[3]) HIP_TRY(hipEventElapsedTime(&out->reduce_ms, e->ev[5], e->ev[6]));
  return DMX_OK;
}

extern "C" int dmx_engine_mean_kernel_times(dmx_engine* e, int32_t reset, dmx_kernel_time_means* out) {
  if (!e) return set_error(DMX_ERR_ARG, "dmx_engine_mean_kernel_times: null engine");
  HIP_TRY(hipSetDevice(e->device));
  HIP_TRY(hipStreamSynchronize(e->stream));
  if (out) {
    dmx_kernel_time_means m{};
    const int64_t ns = std::min<int64_t>(e->n_ring_s, dmx_engine::kRing), nd = std::min<int64_t>(e->n_ring_d, dmx_engine::kRing);
    for (int64_t i = 0; i < ns; ++i) {
      float ms = 0.f;
      HIP_TRY(hipEventElapsedTime(&ms, e->ring_s[i][0], e->ring_s[i][1]));
      m.singlet_ms += ms / (double)ns;
    }
    for (int64_t i = 0; i < nd; ++i) {
      float a = 0.f, b = 0.f, c = 0.f;
      HIP_TRY(hipEventElapsedTime(&a, e->ring_d[i][0], e->ring_d[i][1]));
      HIP_TRY(hipEventElapsedTime(&b, e->ring_d[i][1], e->ring_d[i][2]));
      HIP_TRY(hipEventElapsedTime(&c, e->ring_d[i][2], e->ring_d[i][3]));
      m.doublet_ms += a / (double)nd; m.reduce_ms += b / (double)nd; m.certify_ms += c / (double)nd;
    }
    m.n_singlet = (int32_t)ns; m.n_doublet = (int32_t)nd;
    *out = m;
  }
  if (reset) e->n_ring_s = e->n_ring_d = 0;
  return DMX_OK;
}

extern "C" int dmx_engine_algorithmic_bytes(dmx_engine* e, dmx_kernel_bytes* out) {
  if (!e || !out) return set_error(DMX_ERR_ARG, "dmx_engine_algorithmic_bytes: null argument");
  if (!e->have_pileup) return set_error(DMX_ERR_STATE, "dmx_engine_algorithmic_bytes: no pileup staged");
  const double B = e->pv.B, V = e->V, A = e->A, S = e->S;
  // every input array once: pair counts + read bytes (+ SNP ids when sparse) + per-cell offsets and
  // launch order + the genotype matrix and gp0s once (they are L2/Infinity-Cache resident across cells)
  const double in = (double)e->P * e->nrd_width + (double)e->R + (e->pv.pair_snp ? 4.0 * (double)e->P : 0.0) +
                    (B + 1) * 16 + B * 4 + S * V * 12 + S * 24;
  out->singlet_bytes = in + B * (V + 1) * 8;
  out->doublet_bytes = in + B * (V * V * A + A) * 8;
  out->reduce_bytes = B * (V * V * A + A) * 8 + B * (double)sizeof(dmx_cell_summary);
  return DMX_OK;
}

namespace {
// "k_doublet_a2<256, 4, 4, true, false, 32>" — the demangled name rocprofv3 prints, without return type, namespace and parameter list
void kernel_name(const void* fn, char out[96]) {
  out[0] = 0;
  if (!fn) return;
  const char* mangled = hipKernelNameRefByPtr(fn, nullptr);
  if (!mangled) return;
  int st = 0;
  char* dem = abi::__cxa_demangle(mangled, nullptr, nullptr, &st);
  std::string n = (st == 0 && dem) ? dem : mangled;
  if (dem) std::free(dem);
  if (n.rfind("void ", 0) == 0) n.erase(0, 5);
  int depth = 0; size_t cut = n.size();                                  // the parameter list: the first '(' outside template brackets that is not "(anonymous namespace)"
  for (size_t i = 0; i < n.size(); ++i) {
    if (n[i] == '<') ++depth; else if (n[i] == '>') --depth;
    else if (n[i] == '(' && depth == 0 && n.compare(i, 21, "(anonymous namespace)") != 0) { cut = i; break; }
  }
  n.erase(cut);
  for (const char* pre : {"(anonymous namespace)::", "dmx::"}) for (size_t q; (q = n.find(pre)) != std::string::npos;) n.erase(q, std::strlen(pre));
  std::snprintf(out, 96, "%s", n.c_str());
}
}  // namespace

extern "C" int dmx_engine_kernel_names(dmx_engine* e, dmx_kernel_names* out) {
  if (!e || !out) return set_error(DMX_ERR_ARG, "dmx_engine_kernel_names: null argument");
  std::memset(out, 0, sizeof *out);
  kernel_name(e->k1_fn, out->singlet); kernel_name(e->k2_fn, out->doublet); kernel_name(e->k3b_fn, out->certify);
  out->k1_placement = e->k1_placement;
  return DMX_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
namespace {
template <int WHICH>   // 0: dmx_log (128 bins; the singlet kernels), 2: dmx_log2 (256 bins; the doublet kernels), 3: dmx_log2_lite, 4: dmx_log2_lite32 (FAST)
__global__ void k_debug_log(const double* __restrict__ x, double* __restrict__ y, int64_t n, const double* __restrict__ tab) {
  __shared__ double s_log[DMX_LOG2_TABLE_DOUBLES];
  for (int i = threadIdx.x; i < (WHICH == 4 ? DMX_LOG32_TABLE_DOUBLES : WHICH >= 2 ? DMX_LOG2_TABLE_DOUBLES : DMX_LOG_TABLE_DOUBLES); i += blockDim.x) s_log[i] = tab[i];
  __syncthreads();
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const double v = WHICH == 4 ? dmx_log2_lite32(x[i], s_log, dmx_log32_pins())
                   : WHICH == 3 ? dmx_log2_lite(x[i], s_log, dmx_log_pins()) : (WHICH == 2 ? dmx_log2_fast(x[i], s_log) : dmx_log_fast(x[i], s_log));
    y[i] = dmx_log_is_special(x[i]) ? log(x[i]) : v;
  }
}
}  // namespace

namespace {
__global__ void k_debug_div(const double* __restrict__ a, const double* __restrict__ b, double* __restrict__ q, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    q[i] = div_by(a[i], b[i], rcp_refined(b[i]));
}
}  // namespace

namespace {
// log() ceiling of the device: every lane evaluates `iters` logs of register-resident arguments spread over (0.01, 1] (the
// range likelihood terms live in), four independent streams per lane so the issue rate, not the latency, is measured.
// WHICH = 0: dmx_log (the singlet kernels), 1: ocml log(), 2: dmx_log2 (the doublet kernels).
template <int WHICH>
__global__ __launch_bounds__(256) void k_log_rate(int iters, const double* __restrict__ tab, double* __restrict__ sink) {
  __shared__ double s_log[DMX_LOG2_TABLE_DOUBLES];
  for (int i = threadIdx.x; i < (WHICH == 2 ? DMX_LOG2_TABLE_DOUBLES : DMX_LOG_TABLE_DOUBLES); i += blockDim.x) s_log[i] = tab[i];
  __syncthreads();
  const double x0 = 0.01 + 0.9 * ((threadIdx.x * 37 + blockIdx.x * 11) % 1024) / 1024.0;
  double x[4] = {x0, x0 * 0.75, x0 * 0.5, x0 * 0.31}, acc[4] = {0.0, 0.0, 0.0, 0.0};
  const double dx = 1.0 / (1024.0 * 1024.0);
  for (int i = 0; i < iters; i += 4) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      acc[u] += WHICH == 0 ? dmx_log(x[u], s_log) : (WHICH == 2 ? dmx_log2_fast(x[u], s_log) : log(x[u]));
      x[u] += dx;
    }
  }
  sink[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
}
}  // namespace

extern "C" int dmx_debug_log_rate(int32_t which, int32_t iters, int32_t device, double* logs_per_second) {
  if (which < 0 || which > 2 || iters < 4 || !logs_per_second) return set_error(DMX_ERR_ARG, "dmx_debug_log_rate: bad arguments");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return set_error(DMX_ERR_NOGPU, "dmx_debug_log_rate: no such HIP device");
  HIP_TRY(hipSetDevice(device));
  const int blocks = 256 * 8, threads = 256;                  // 8 workgroups (32 wavefronts) per CU
  double *dt = nullptr, *ds = nullptr;
  HIP_TRY(hipMalloc((void**)&dt, sizeof(double) * DMX_LOG2_TABLE_DOUBLES));
  HIP_TRY(hipMalloc((void**)&ds, sizeof(double) * (size_t)blocks * threads));
  if (which == 2) HIP_TRY(hipMemcpy(dt, dmx_log2_table_host, sizeof(double) * DMX_LOG2_TABLE_DOUBLES, hipMemcpyHostToDevice));
  else HIP_TRY(hipMemcpy(dt, dmx_log_table_host, sizeof(double) * DMX_LOG_TABLE_DOUBLES, hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1));
  iters &= ~3;
  float best = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {                          // first pass warms up; keep the fastest
    HIP_TRY(hipEventRecord(e0, 0));
    if (which == 0) hipLaunchKernelGGL((k_log_rate<0>), dim3(blocks), dim3(threads), 0, 0, iters, dt, ds);
    else if (which == 2) hipLaunchKernelGGL((k_log_rate<2>), dim3(blocks), dim3(threads), 0, 0, iters, dt, ds);
    else            hipLaunchKernelGGL((k_log_rate<1>), dim3(blocks), dim3(threads), 0, 0, iters, dt, ds);
    HIP_TRY(hipEventRecord(e1, 0));
    HIP_TRY(hipEventSynchronize(e1));
    HIP_TRY(hipGetLastError());
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    if (rep > 0 && ms < best) best = ms;
  }
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  (void)hipFree(dt); (void)hipFree(ds);
  *logs_per_second = (double)blocks * threads * iters / (best * 1e-3);
  return DMX_OK;
}

extern "C" int dmx_debug_device_div(const double* a, const double* b, double* q, int64_t n, int32_t device) {
  if (!a || !b || !q || n < 0) return set_error(DMX_ERR_ARG, "dmx_debug_device_div: bad arguments");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return set_error(DMX_ERR_NOGPU, "dmx_debug_device_div: no such HIP device");
  HIP_TRY(hipSetDevice(device));
  double *da = nullptr, *db = nullptr, *dq = nullptr;
  const size_t bytes = std::max<size_t>(sizeof(double) * (size_t)n, 16);
  HIP_TRY(hipMalloc((void**)&da, bytes)); HIP_TRY(hipMalloc((void**)&db, bytes)); HIP_TRY(hipMalloc((void**)&dq, bytes));
  HIP_TRY(hipMemcpy(da, a, sizeof(double) * (size_t)n, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(db, b, sizeof(double) * (size_t)n, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_debug_div, dim3(1024), dim3(256), 0, 0, da, db, dq, n);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpy(q, dq, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost));
  (void)hipFree(da); (void)hipFree(db); (void)hipFree(dq);
  return DMX_OK;
}

namespace {
int debug_device_log(int which, const double* x, double* y, int64_t n, int32_t device) {
  if (!x || !y || n < 0) return set_error(DMX_ERR_ARG, "dmx_debug_device_log: bad arguments");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return set_error(DMX_ERR_NOGPU, "dmx_debug_device_log: no such HIP device");
  HIP_TRY(hipSetDevice(device));
  double *dx = nullptr, *dy = nullptr, *dt = nullptr;
  HIP_TRY(hipMalloc((void**)&dx, std::max<size_t>(sizeof(double) * (size_t)n, 16)));
  HIP_TRY(hipMalloc((void**)&dy, std::max<size_t>(sizeof(double) * (size_t)n, 16)));
  HIP_TRY(hipMalloc((void**)&dt, sizeof(double) * DMX_LOG2_TABLE_DOUBLES));
  HIP_TRY(hipMemcpy(dx, x, sizeof(double) * (size_t)n, hipMemcpyHostToDevice));
  if (which == 4) HIP_TRY(hipMemcpy(dt, dmx_log32_table_host, sizeof(double) * DMX_LOG32_TABLE_DOUBLES, hipMemcpyHostToDevice));
  else if (which >= 2) HIP_TRY(hipMemcpy(dt, dmx_log2_table_host, sizeof(double) * DMX_LOG2_TABLE_DOUBLES, hipMemcpyHostToDevice));
  else HIP_TRY(hipMemcpy(dt, dmx_log_table_host, sizeof(double) * DMX_LOG_TABLE_DOUBLES, hipMemcpyHostToDevice));
  if (which == 4) hipLaunchKernelGGL((k_debug_log<4>), dim3(1024), dim3(256), 0, 0, dx, dy, n, dt);
  else if (which == 3) hipLaunchKernelGGL((k_debug_log<3>), dim3(1024), dim3(256), 0, 0, dx, dy, n, dt);
  else if (which == 2) hipLaunchKernelGGL((k_debug_log<2>), dim3(1024), dim3(256), 0, 0, dx, dy, n, dt);
  else hipLaunchKernelGGL((k_debug_log<0>), dim3(1024), dim3(256), 0, 0, dx, dy, n, dt);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpy(y, dy, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost));
  (void)hipFree(dx); (void)hipFree(dy); (void)hipFree(dt);
  return DMX_OK;
}
}  // namespace
extern "C" int dmx_debug_device_log(const double* x, double* y, int64_t n, int32_t device) { return debug_device_log(0, x, y, n, device); }
extern "C" int dmx_debug_device_log2(const double* x, double* y, int64_t n, int32_t device) { return debug_device_log(2, x, y, n, device); }
extern "C" int dmx_debug_device_log2_lite(const double* x, double* y, int64_t n, int32_t device) { return debug_device_log(3, x, y, n, device); }
extern "C" int dmx_debug_device_log2_lite32(const double* x, double* y, int64_t n, int32_t device) { return debug_device_log(4, x, y, n, device); }

// ---------------------------------------------------------------------------------------------------------------------
// (ABI 8) `.pair` rows formatted on the device (csrc/dmx_format.hpp)
#include "dmx_format.hpp"

struct dmx_pair_text {
  int device = 0;
  char* d_text = nullptr;
  int64_t n_bytes = 0;
  std::vector<int64_t> cell_off;
  std::vector<uint8_t> cell_flag;
  std::vector<dmx_pair_patch> patches;
  double format_ms = 0.0;
};

extern "C" void dmx_pair_text_free(dmx_pair_text* t) {
  if (!t) return;
  if (t->d_text) { (void)hipSetDevice(t->device); (void)hipFree(t->d_text); }
  delete t;
}

extern "C" int dmx_pair_text_get_info(const dmx_pair_text* t, dmx_pair_text_info* out) {
  if (!t || !out) return set_error(DMX_ERR_ARG, "dmx_pair_text_get_info: null argument");
  out->n_bytes = t->n_bytes; out->n_out = (int32_t)t->cell_flag.size(); out->n_patches = (int32_t)t->patches.size();
  out->cell_off = t->cell_off.data(); out->cell_flag = t->cell_flag.data(); out->patches = t->patches.data(); out->format_ms = t->format_ms;
  return DMX_OK;
}

extern "C" int dmx_pair_text_read(dmx_pair_text* t, int64_t offset, int64_t n, void* dst) {
  if (!t || offset < 0 || n < 0 || offset + n > t->n_bytes || (n && !dst)) return set_error(DMX_ERR_ARG, "dmx_pair_text_read: bad arguments");
  if (!n) return DMX_OK;
  HIP_TRY(hipSetDevice(t->device));
  HIP_TRY(hipMemcpy(dst, t->d_text + offset, (size_t)n, hipMemcpyDeviceToHost));
  return DMX_OK;
}

extern "C" int dmx_engine_format_pair(dmx_engine* e, const dmx_pair_request* rq, dmx_pair_text** out) {
  if (!e || !rq || !out || rq->n_out < 0 || (rq->n_out && (!rq->cells || !rq->barcodes)) || !rq->sample_ids)
    return set_error(DMX_ERR_ARG, "dmx_engine_format_pair: null argument");
  if (!e->have_grid) return set_error(DMX_ERR_STATE, "dmx_engine_format_pair: run_doublet has not been called");
  const int32_t V = e->V, A = e->A, n_out = rq->n_out;
  if (V > 4095 || A > 255) return set_error(DMX_ERR_ARG, "dmx_engine_format_pair: %d samples x %d alphas exceed the row map's fields", V, A);
  for (int32_t i = 0; i < n_out; ++i) {
    if (rq->cells[i] < 0 || rq->cells[i] >= e->pv.B) return set_error(DMX_ERR_ARG, "dmx_engine_format_pair: cell %d out of range (0..%d)", rq->cells[i], e->pv.B - 1);
    if (!rq->barcodes[i]) return set_error(DMX_ERR_ARG, "dmx_engine_format_pair: barcodes[%d] is null", i);
  }
  for (int32_t j = 0; j < V; ++j) if (!rq->sample_ids[j]) return set_error(DMX_ERR_ARG, "dmx_engine_format_pair: sample_ids[%d] is null", j);
  HIP_TRY(hipSetDevice(e->device));
  // ---- host-side tables: the rows of one barcode in print order (:772-797), the strings, "\t%.3lf\t" of every alpha, powers of ten
  std::vector<uint32_t> rowmap;
  for (int32_t j = 0; j < V; ++j) {
    rowmap.push_back(((uint32_t)j << 20) | ((uint32_t)j << 8));                                  // :774-781 the singlet row (SM1 = SM2 = j, alpha[0])
    for (int32_t k = 0; k < V; ++k)
      for (int32_t a = 1; a < A; ++a) {
        if (j == k) continue;
        if (j > k && e->alpha[(size_t)a] == 0.5) continue;                                       // :785
        rowmap.push_back(((uint32_t)j << 20) | ((uint32_t)k << 8) | (uint32_t)a);
      }
  }
  auto pool_of = [](const char* const* strs, int32_t n, std::string* pool, std::vector<uint32_t>* off, size_t* longest) -> bool {
    off->assign(1, 0u);
    for (int32_t i = 0; i < n; ++i) {
      const size_t len = std::strlen(strs[i]);
      if (pool->size() + len >= 0xFFFFFFF0ull) return false;
      pool->append(strs[i], len);
      off->push_back((uint32_t)pool->size());
      *longest = std::max(*longest, len);
    }
    return true;
  };
  std::string bc_pool, sm_pool, al_pool;
  std::vector<uint32_t> bc_off, sm_off, al_off(1, 0u);
  size_t bc_max = 0, sm_max = 0, al_max = 0;
  if (!pool_of(rq->barcodes, n_out, &bc_pool, &bc_off, &bc_max) || !pool_of(rq->sample_ids, V, &sm_pool, &sm_off, &sm_max))
    return set_error(DMX_ERR_ARG, "dmx_engine_format_pair: more than 4 GB of barcode text");
  for (int32_t a = 0; a < A; ++a) {
    char buf[400];
    const int n = std::snprintf(buf, sizeof buf, "\t%.3lf\t", e->alpha[(size_t)a]);              // the reference's own conversion (:776,:788)
    if (n < 0 || (size_t)n >= sizeof buf) return set_error(DMX_ERR_ARG, "dmx_engine_format_pair: alpha[%d] does not print", a);
    al_pool.append(buf, (size_t)n); al_off.push_back((uint32_t)al_pool.size()); al_max = std::max(al_max, (size_t)n);
  }
  const size_t max_row = bc_max + 1 + 2 * sm_max + 1 + al_max + 20 + 1 + 12 + 1;
  const size_t lds = (size_t)dmx_fmt::kFmtThreads * max_row;
  if (lds > 150 * 1024) return set_error(DMX_ERR_ARG, "dmx_engine_format_pair: rows of up to %zu bytes do not fit the kernel's buffer", max_row);
  std::vector<double> pow10(310);
  for (int n = 0; n < 310; ++n) { char b[16]; std::snprintf(b, sizeof b, "1e%d", n); pow10[(size_t)n] = std::strtod(b, nullptr); }   // correctly rounded
  // ---- device copies (small; per call)
  struct Dev {
    std::vector<void*> p;
    ~Dev() { for (void* x : p) if (x) (void)hipFree(x); }
  } dev;
  hipStream_t st = e->stream;
  auto up = [&](const void* src, size_t bytes, void** d) -> int {
    HIP_TRY(hipMalloc(d, std::max<size_t>(bytes, 16)));
    dev.p.push_back(*d);
    if (bytes && src) HIP_TRY(hipMemcpyAsync(*d, src, bytes, hipMemcpyHostToDevice, st));
    return DMX_OK;
  };
  dmx_fmt::Ctx c{};
  c.grid = e->d_grid; c.summ = e->d_sum; c.alpha = e->d_alpha; c.prior = e->prior; c.V = V; c.A = A; c.n_out = n_out; c.n_rows = (int32_t)rowmap.size();
  c.max_row = (int32_t)max_row;
  void* d = nullptr;
  if (int rc = up(rq->cells, sizeof(int32_t) * (size_t)n_out, &d)) return rc; c.cells = (const int32_t*)d;
  if (rq->host_rows) { if (int rc = up(rq->host_rows, (size_t)n_out, &d)) return rc; c.host_rows = (const uint8_t*)d; }
  if (rq->ovr) { if (int rc = up(rq->ovr, sizeof(dmx_pair_override) * (size_t)n_out, &d)) return rc; c.ovr = (const dmx_pair_override*)d; }
  if (int rc = up(rowmap.data(), sizeof(uint32_t) * rowmap.size(), &d)) return rc; c.rowmap = (const uint32_t*)d;
  if (int rc = up(bc_pool.data(), bc_pool.size(), &d)) return rc; c.bc_pool = (const char*)d;
  if (int rc = up(bc_off.data(), sizeof(uint32_t) * bc_off.size(), &d)) return rc; c.bc_off = (const uint32_t*)d;
  if (int rc = up(sm_pool.data(), sm_pool.size(), &d)) return rc; c.sm_pool = (const char*)d;
  if (int rc = up(sm_off.data(), sizeof(uint32_t) * sm_off.size(), &d)) return rc; c.sm_off = (const uint32_t*)d;
  if (int rc = up(al_pool.data(), al_pool.size(), &d)) return rc; c.al_pool = (const char*)d;
  if (int rc = up(al_off.data(), sizeof(uint32_t) * al_off.size(), &d)) return rc; c.al_off = (const uint32_t*)d;
  if (int rc = up(pow10.data(), sizeof(double) * pow10.size(), &d)) return rc; c.pow10 = (const double*)d;
  int64_t* d_len = nullptr; uint32_t* d_np = nullptr; int64_t* d_off = nullptr; uint32_t* d_poff = nullptr; uint8_t* d_flag = nullptr;
  if (int rc = up(nullptr, sizeof(int64_t) * (size_t)std::max(n_out, 1), (void**)&d_len)) return rc;
  if (int rc = up(nullptr, sizeof(uint32_t) * (size_t)std::max(n_out, 1), (void**)&d_np)) return rc;
  if (int rc = up(nullptr, sizeof(int64_t) * ((size_t)n_out + 1), (void**)&d_off)) return rc;
  if (int rc = up(nullptr, sizeof(uint32_t) * ((size_t)n_out + 1), (void**)&d_poff)) return rc;
  if (int rc = up(nullptr, (size_t)std::max(n_out, 1), (void**)&d_flag)) return rc;
  c.cell_len = d_len; c.cell_npatch = d_np; c.cell_off = d_off; c.cell_poff = d_poff; c.cell_flag = d_flag;
  std::unique_ptr<dmx_pair_text, void (*)(dmx_pair_text*)> t(new (std::nothrow) dmx_pair_text, dmx_pair_text_free);
  if (!t) return set_error(DMX_ERR_NOMEM, "dmx_engine_format_pair: out of memory");
  t->device = e->device;
  t->cell_off.assign((size_t)n_out + 1, 0);
  t->cell_flag.assign((size_t)n_out, 0);
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  HIP_TRY(hipEventCreate(&ev0)); HIP_TRY(hipEventCreate(&ev1));
  struct Evs { hipEvent_t a, b; ~Evs() { (void)hipEventDestroy(a); (void)hipEventDestroy(b); } } evs{ev0, ev1};
  if (n_out > 0) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&dmx_fmt::k_pair_write), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    HIP_TRY(hipEventRecord(ev0, st));
    hipLaunchKernelGGL(dmx_fmt::k_pair_lengths, dim3((unsigned)n_out), dim3(dmx_fmt::kFmtThreads), 0, st, c);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(dmx_fmt::k_pair_scan, dim3(1), dim3(1024), 0, st, d_len, d_np, n_out, d_off, d_poff);
    HIP_TRY(hipGetLastError());
    uint32_t n_patch = 0;
    HIP_TRY(hipMemcpyAsync(t->cell_off.data(), d_off, sizeof(int64_t) * ((size_t)n_out + 1), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(&n_patch, d_poff + n_out, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(t->cell_flag.data(), d_flag, (size_t)n_out, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    t->n_bytes = t->cell_off[(size_t)n_out];
    t->patches.resize(n_patch);
    HIP_TRY(hipMalloc((void**)&t->d_text, (size_t)std::max<int64_t>(t->n_bytes, 16)));
    dmx_pair_patch* d_patch = nullptr;
    if (int rc = up(nullptr, sizeof(dmx_pair_patch) * (size_t)std::max<uint32_t>(n_patch, 1), (void**)&d_patch)) return rc;
    c.text = t->d_text; c.patches = d_patch;
    hipLaunchKernelGGL(dmx_fmt::k_pair_write, dim3((unsigned)n_out), dim3(dmx_fmt::kFmtThreads), lds, st, c);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(ev1, st));
    if (n_patch) HIP_TRY(hipMemcpyAsync(t->patches.data(), d_patch, sizeof(dmx_pair_patch) * n_patch, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, ev0, ev1));
    t->format_ms = ms;
  }
  *out = t.release();
  return DMX_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// cmd_cram_demuxlet.cpp:390-881 in one call: store (or a frozen pileup) + genotype matrix in, four text files out.
extern "C" int dmx_demuxlet_run(const dmx_job* job) {
  if (!job || (!job->store && !(job->pileup && job->barcodes)) || !job->g || !job->sample_ids || !job->alpha || !job->out_prefix)
    return set_error(DMX_ERR_ARG, "dmx_demuxlet_run: null argument");
  using clk = std::chrono::steady_clock;
  auto secs = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double>(b - a).count(); };
  const clk::time_point t_begin = clk::now();
  dmx_job_timing tm{};
  dmx_pileup pl;
  if (job->store) { if (int rc = dmx_store_freeze(job->store, &pl)) return rc; }
  else {
    pl = *job->pileup;
    if (pl.memory != DMX_MEM_HOST && pl.memory != DMX_MEM_DEVICE) return set_error(DMX_ERR_ARG, "dmx_demuxlet_run: job->pileup.memory %d", pl.memory);
    if (!pl.rd_totl || !pl.rd_pass || !pl.rd_uniq) return set_error(DMX_ERR_ARG, "dmx_demuxlet_run: job->pileup needs the per-cell read counters");
    // everything this function itself indexes before the engine's own checks run
    if (pl.n_cells < 0 || pl.n_snps < 0 || pl.n_pairs < 0 || pl.n_reads < 0) return set_error(DMX_ERR_ARG, "dmx_demuxlet_run: job->pileup has a negative size");
    if (!pl.cell_pair_off || !pl.cell_read_off) return set_error(DMX_ERR_ARG, "dmx_demuxlet_run: job->pileup needs cell_pair_off and cell_read_off");
    for (int32_t c = 0; c < pl.n_cells; ++c) if (!job->barcodes[c]) return set_error(DMX_ERR_ARG, "dmx_demuxlet_run: barcodes[%d] is null", c);
  }
  // A DEVICE-resident pileup (dmx_pileup.memory == DMX_MEM_DEVICE: the five arrays live in the HBM of job->device; counters and barcodes are
  // host memory as always): nothing is sliced or copied on the host.  Its two offset arrays come to the host once (range cuts, N.SNP);
  // a range of consecutive cells is a view of the caller's arrays, any other range is gathered on the device; the few barcodes the tie
  // arbiter has to re-evaluate (near-tie flags, an open tie-order certificate) have their pieces of the pileup fetched when their range is written.
  const bool dev_pl = pl.memory == DMX_MEM_DEVICE;
  const dmx_pileup pl_dev = pl;                    // what the engines are given
  std::vector<int64_t> h_po, h_ro;
  if (dev_pl) {
    if (job->n_gpus > 1) return set_error(DMX_ERR_ARG, "dmx_demuxlet_run: a device-resident pileup lives on one GPU (n_gpus = %d)", job->n_gpus);
    int nd = 0;
    if (hipGetDeviceCount(&nd) != hipSuccess || nd <= 0) return set_error(DMX_ERR_NOGPU, "dmx_demuxlet_run: no HIP device is visible (this library has no CPU fallback)");
    HIP_TRY(hipSetDevice(job->device % nd));
    h_po.resize((size_t)pl.n_cells + 1); h_ro.resize((size_t)pl.n_cells + 1);
    HIP_TRY(hipMemcpy(h_po.data(), pl.cell_pair_off, sizeof(int64_t) * ((size_t)pl.n_cells + 1), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(h_ro.data(), pl.cell_read_off, sizeof(int64_t) * ((size_t)pl.n_cells + 1), hipMemcpyDeviceToHost));
    if (h_po[0] != 0 || h_po[(size_t)pl.n_cells] != pl.n_pairs || h_ro[0] != 0 || h_ro[(size_t)pl.n_cells] != pl.n_reads)
      return set_error(DMX_ERR_ARG, "dmx_demuxlet_run: job->pileup's offset arrays do not span n_pairs / n_reads");
    for (int32_t c = 0; c < pl.n_cells; ++c)
      if (h_po[(size_t)c + 1] < h_po[(size_t)c] || h_ro[(size_t)c + 1] < h_ro[(size_t)c]) return set_error(DMX_ERR_ARG, "dmx_demuxlet_run: job->pileup's offsets are not monotone at cell %d", c);
    pl.cell_pair_off = h_po.data(); pl.cell_read_off = h_ro.data();      // the host logic below reads offsets only; the big arrays stay device pointers
  }
  if (job->n_samples < 1 || job->n_alpha < 1) return set_error(DMX_ERR_ARG, "dmx_demuxlet_run: n_samples %d, n_alpha %d", job->n_samples, job->n_alpha);
  for (int32_t j = 0; j < job->n_samples; ++j) if (!job->sample_ids[j]) return set_error(DMX_ERR_ARG, "dmx_demuxlet_run: sample_ids[%d] is null", j);
  tm.freeze_s = secs(t_begin, clk::now());
  const clk::time_point t_setup = clk::now();
  const int32_t B = pl.n_cells, V = job->n_samples, A = job->n_alpha;
  const size_t nAB = (size_t)V * V * A;
  const bool doublet_ok = V >= 2 && A >= 2;
  std::vector<const char*> bcs((size_t)B);
  std::vector<int32_t> nsnp((size_t)B);
  for (int32_t c = 0; c < B; ++c) {
    bcs[c] = job->store ? dmx_store_barcode(job->store, c) : job->barcodes[c];
    nsnp[c] = (int32_t)(pl.cell_pair_off[c + 1] - pl.cell_pair_off[c]);
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return set_error(DMX_ERR_NOGPU, "dmx_demuxlet_run: no HIP device is visible (this library has no CPU fallback)");

  // ---- ranges: contiguous runs of the byte-wise sorted barcodes (= output order, cmd_cram_demuxlet.cpp:472,:576) with equal
  // work.  One engine per GPU; a range is what one engine holds at a time, sized so that its doublet grid stays inside a
  // byte budget (the grid, nb * V*V*A doubles, is the only thing that grows with the panel).  Ranges go through the engines
  // in waves; while the host arbitrates, formats and appends the rows of wave w, the GPUs already compute wave w + 1 — so a
  // job that is big enough is cut into at least four ranges per engine even when memory does not ask for it.
  const int ngpu = std::max(1, std::min(job->n_gpus > 0 ? job->n_gpus : 1, std::max(B, 1)));
  // How a job is cut is decided here and nowhere else: the three variables that override it (tests force many ranges, experiments time the
  // overlap) are read only behind the experiment fence, like the engines' own switches — a stray DMX_* variable cannot change a user's ranges
  const bool fence_open = [] { const char* x = getenv("DMX_EXPERIMENTS"); return x && x[0] == '1' && !x[1]; }();
  auto job_knob = [&](const char* name) -> const char* { return fence_open ? getenv(name) : nullptr; };
  size_t budget = (size_t)4 << 30;                                   // bytes of grid per range (host holds two waves of them)
  if (const char* env = job_knob("DMX_RANGE_BYTES")) budget = (size_t)std::max(1ll, atoll(env));   // tests: force many ranges
  {
    size_t free_b = 0, total_b = 0;
    HIP_TRY(hipSetDevice(job->device % ndev));
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b / 6 < budget) budget = std::max<size_t>(free_b / 6, 1);   // two ranges per GPU in flight
  }
  const double per_pair = (double)(V + 1) + (doublet_ok ? (double)nAB + A : 0.0);
  double work_total = 0;
  for (int32_t c = 0; c < B; ++c) work_total += nsnp[c] * per_pair + 1.0;
  const double grid_total = doublet_ok ? (double)B * (double)nAB * 8.0 : 0.0;
  const int by_mem = (int)std::min<double>((double)std::max(B, 1), std::ceil(grid_total / (double)budget));
  int by_overlap = 1;                                                // ranges per engine wanted for host/GPU overlap
  if (const char* env = job_knob("DMX_RANGES_PER_GPU")) by_overlap = std::max(1, atoi(env));
  else {
    // worth it when an engine has more than ~0.15 s of kernels ahead of it (at ~6e11 evaluations/s; FAST evaluates the printed
    // entries only): the host then writes range r and stages r + 2 while the GPU computes r + 1
    const bool sym = job->mode == DMX_MODE_FAST && A == 2 && job->alpha[0] == 0.0 && job->alpha[1] == 0.5 && V <= 512;
    const double evals = (double)(V + 1) + (doublet_ok ? (sym ? (double)V + 0.5 * V * (V + 1) : (double)nAB) : 0.0);
    // A device-resident pileup has nothing to stage: what ranges buy there is only the writer thread's overlap (0.06-0.08 s at cfg3 size), and
    // they cost GPU time — a range of 2 500 one-barcode wavefronts fills 3 072 slots less well than one launch of 10 000, and two engines'
    // workgroups interleave so that both ranges finish together.  Measured at cfg3 size, FAST / STRICT: one range 0.383 / 1.29 s, two 0.407 /
    // 1.31, four 0.55 / 1.39 (profiles/r05_e2e_ranges.txt; chaining the ranges by events was worse still).  So such a job is cut only when
    // every range still makes several full rounds of wavefronts.
    const int min_cells = dev_pl ? 40 * 1024 : 8 * 1024;
    if (doublet_ok && (double)pl.n_pairs * evals / ngpu > 0.15 * 6e11 && B / ngpu >= min_cells) by_overlap = 4;
  }
  int R = std::max(ngpu * by_overlap, by_mem);
  R = ((R + ngpu - 1) / ngpu) * ngpu;                                 // whole waves
  R = std::max(1, std::min(R, std::max(B, 1)));
  const size_t cell_cap = doublet_ok ? std::max<size_t>(1, budget / (nAB * 8)) : (size_t)B + 1;   // a range never holds more cells than its grid budget
  std::vector<int32_t> order((size_t)B);
  std::iota(order.begin(), order.end(), 0);
  const bool sliced = R > 1;
  if (sliced) std::sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return std::strcmp(bcs[a], bcs[b]) < 0; });
  std::vector<int32_t> cut;
  cut.push_back(0);
  {
    double run = 0;
    int r = 1;
    for (int32_t i = 0; i < B; ++i) {
      run += nsnp[order[i]] * per_pair + 1.0;
      const bool by_work = r < R && run >= work_total * r / R;
      const bool by_cap = (size_t)(i + 1 - cut.back()) >= cell_cap;
      if ((by_work || by_cap) && i + 1 < B) { cut.push_back(i + 1); if (by_work) ++r; }
    }
    cut.push_back(B);
    if (B == 0) cut.assign({0, 0});
  }
  R = (int)cut.size() - 1;
  tm.n_ranges = R;

  struct Range {
    std::vector<int32_t> totl, pass, uniq, ns;
    std::vector<const char*> bc;
    std::vector<double> llks, llk0s, grid, l00, sing;
    std::vector<dmx_cell_summary> summ;
    std::vector<std::vector<double>> flagged_grid;
    std::vector<const double*> cell_grid;
    // device-resident pileups: the host pieces of the barcodes the tie arbiter may have to walk (tie_cell[k] = index here, -1 = not staged)
    std::vector<int32_t> tie_cell, t_snp; std::vector<int64_t> t_po, t_ro; std::vector<uint8_t> t_nrd, t_reads;
    // write_pair with the rows formatted on the device: the packed text (device memory, owned by the range until it is written) and its output cells
    dmx_pair_text* ptext = nullptr; std::vector<int32_t> out_cells;
    int32_t lo = 0, hi = 0;
    void release() { if (ptext) dmx_pair_text_free(ptext); *this = Range(); }
  };
  std::vector<Range> rg((size_t)R);
  struct RangeGuard { std::vector<Range>* r; ~RangeGuard() { for (Range& x : *r) if (x.ptext) { dmx_pair_text_free(x.ptext); x.ptext = nullptr; } } } range_guard{&rg};
  // `.pair` rows are formatted on the device (dmx_engine_format_pair) unless the fenced DMX_PAIR_ON_HOST=1 asks for the host formatter (tests compare the two)
  const bool gpu_pair = job->write_pair && doublet_ok && !job_knob("DMX_PAIR_ON_HOST");
  double pair_format_ms = 0; int64_t pair_bytes = 0, pair_patches = 0, pair_host_cells = 0;
  // Engines: one per GPU and wave slot.  When the job takes several waves, every GPU gets TWO engines (own stream, own buffers)
  // that alternate between waves, so that the slicing + H2D of wave w + 2 overlaps the kernels of wave w + 1 (the copy engine
  // runs beside the compute units) while the host arbitrates and writes wave w.
  const int per_wave = std::min(ngpu, R);
  const int waves = (R + per_wave - 1) / per_wave;
  const int nset = (waves >= 2 && !job_knob("DMX_ONE_ENGINE_PER_GPU")) ? 2 : 1;
  std::vector<dmx_engine*> eng((size_t)per_wave * nset, nullptr);
  tm.n_engines = (int32_t)eng.size();
  struct Guard { std::vector<dmx_engine*>* e; ~Guard() { for (dmx_engine* x : *e) if (x) dmx_engine_destroy(x); } } guard{&eng};
  {
    auto make = [&](size_t i) -> int {
      dmx_engine_config cfg{};
      cfg.n_samples = V; cfg.n_alpha = A; cfg.alpha = job->alpha; cfg.doublet_prior = job->doublet_prior;
      cfg.device = (job->device + (int)(i % (size_t)per_wave)) % ndev; cfg.mode = job->mode;
      if (!job->arbiter) cfg.flags |= DMX_ENGINE_NO_CERTIFY;
      if (int rc = dmx_engine_create(&cfg, &eng[i])) return rc;
      return dmx_engine_set_genotypes(eng[i], job->g, pl.n_snps, DMX_MEM_HOST);
    };
    if (int rc = make(0)) return rc;              // the first one builds the host-side seed tables the others reuse
    std::vector<int> rcs(eng.size(), DMX_OK);
    std::vector<std::string> msgs(eng.size());
    std::vector<std::thread> th;
    for (size_t i = 1; i < eng.size(); ++i) th.emplace_back([&, i] { rcs[i] = make(i); if (rcs[i]) msgs[i] = dmx_last_error(); });
    for (std::thread& t : th) t.join();
    for (size_t i = 1; i < eng.size(); ++i) if (rcs[i]) return set_error(rcs[i], "%s", msgs[i].c_str());
  }
  tm.setup_s = secs(t_setup, clk::now());
  std::mutex tm_mu;
  const bool trace = getenv("DMX_E2E_TRACE") != nullptr;     // per-range timeline on stderr (seconds since the call began)
  auto mark = [&](const char* what, int r, clk::time_point t0) {
    if (trace) fprintf(stderr, "[dmx_demuxlet_run] %-6s range %d: %.4f -> %.4f s\n", what, r, secs(t_begin, t0), secs(t_begin, clk::now()));
  };
  double stage_s = 0, wait_s = 0, write_s = 0, kernel_ms = 0;
  int32_t n_fetched = 0;

  auto make_fin = [&](Range& x) {
    dmx_final_input fin{};
    fin.n_cells = x.hi - x.lo; fin.n_samples = V; fin.n_alpha = A; fin.alpha = job->alpha; fin.doublet_prior = job->doublet_prior;
    fin.min_total = job->min_total; fin.min_uniq = job->min_uniq; fin.min_snp = job->min_snp; fin.write_pair = job->write_pair;
    fin.sample_ids = job->sample_ids;
    if (sliced) { fin.barcodes = x.bc.data(); fin.rd_totl = x.totl.data(); fin.rd_pass = x.pass.data(); fin.rd_uniq = x.uniq.data(); fin.n_snp = x.ns.data(); }
    else { fin.barcodes = bcs.data(); fin.rd_totl = pl.rd_totl; fin.rd_pass = pl.rd_pass; fin.rd_uniq = pl.rd_uniq; fin.n_snp = nsnp.data(); }
    return fin;
  };
  auto eng_of = [&](int r) -> dmx_engine* { return eng[(size_t)(((r / per_wave) % nset) * per_wave + r % per_wave)]; };
  auto launch = [&](int r) -> int {              // stage range r on its engine and start its kernels (asynchronous)
    const clk::time_point t0 = clk::now();
    Range& x = rg[(size_t)r];
    x.lo = cut[(size_t)r]; x.hi = cut[(size_t)r + 1];
    const int32_t nb = x.hi - x.lo;
    if (sliced) {                                // cells order[lo..hi) become cells 0..nb-1 of the range: the small per-cell arrays
      const size_t nb1 = (size_t)std::max(nb, 1);                      // here, the CSR itself while it streams to the device
      x.totl.resize(nb1); x.pass.resize(nb1); x.uniq.resize(nb1); x.ns.resize(nb1); x.bc.resize(nb1);
      for (int32_t k = 0; k < nb; ++k) {
        const int32_t c = order[(size_t)x.lo + k];
        x.totl[(size_t)k] = pl.rd_totl[c]; x.pass[(size_t)k] = pl.rd_pass[c]; x.uniq[(size_t)k] = pl.rd_uniq[c];
        x.ns[(size_t)k] = nsnp[(size_t)c]; x.bc[(size_t)k] = bcs[(size_t)c];
      }
    }
    dmx_engine* e = eng_of(r);
    if (dev_pl) { if (int rc = dmx::engine_set_pileup_cells(e, &pl_dev, sliced ? order.data() + x.lo : nullptr, nb, h_po.data(), h_ro.data())) return rc; }
    else if (int rc = dmx::engine_set_pileup_cells(e, &pl, sliced ? order.data() + x.lo : nullptr, nb)) return rc;
    if (doublet_ok) { if (int rc = dmx_engine_run(e)) return rc; }
    else if (int rc = dmx_engine_run_singlet(e)) return rc;
    { std::lock_guard<std::mutex> lk(tm_mu); stage_s += secs(t0, clk::now()); }
    mark("launch", r, t0);
    return DMX_OK;
  };
  auto fetch = [&](int r) -> int {               // results of range r to the host (waits for its kernels)
    const clk::time_point t0 = clk::now();
    Range& x = rg[(size_t)r];
    const size_t nb = (size_t)(x.hi - x.lo), nb1 = std::max<size_t>(nb, 1);
    dmx_engine* e = eng_of(r);
    x.llks.resize(nb1 * (size_t)V); x.llk0s.resize(nb1);
    if (int rc = dmx_engine_get_singlet(e, x.llks.data(), x.llk0s.data())) return rc;
    int32_t fetched = 0;
    if (doublet_ok) {
      x.l00.resize(nb1 * (size_t)A);
      if (gpu_pair) {
        // .pair from the device: the records as below; the barcodes whose printed entries the arbiter may replace (dmx::cell_needs, a BEST-rule comparison
        // within 1e-7 — decided on the record AFTER resolve_tie_order, as the writers do) keep the host formatter and bring their grids
        x.sing.resize(nb1 * (size_t)V); x.summ.resize(nb1);
        if (int rc = dmx_engine_get_doublet(e, nullptr, x.l00.data(), x.summ.data())) return rc;
        if (int rc = dmx_engine_get_sing(e, x.sing.data())) return rc;
        x.cell_grid.assign(nb1, nullptr);
        const dmx_final_input fin = make_fin(x);
        x.out_cells = dmx::output_cells(&fin, true);
        const size_t n_out = x.out_cells.size();
        std::vector<uint8_t> host_rows(std::max<size_t>(n_out, 1), 0);
        std::vector<dmx_pair_override> ovr(std::max<size_t>(n_out, 1));
        std::vector<const char*> obc(std::max<size_t>(n_out, 1), nullptr);
        std::vector<int32_t> hostc;
        constexpr int32_t kNear = DMX_CELL_NEAR_DOUBLET | DMX_CELL_NEAR_SINGLET;
        for (size_t i = 0; i < n_out; ++i) {
          const size_t c = (size_t)x.out_cells[i];
          dmx_cell_summary sm = x.summ[c];
          if (sm.flags & DMX_CELL_ORDER_RESOLVABLE) (void)dmx::resolve_tie_order(&sm);
          const double* sg = x.sing.data() + c * (size_t)V;
          const bool rule = job->arbiter && sm.n_pairs > 0 &&
                            dmx::near_rule(sm.llk12, sm.llk1, sm.llk2, sg[std::max(sm.i_sing1, 0)], sg[std::max(sm.i_sing2, 0)], 1e-7);
          host_rows[i] = (dmx::cell_needs(sm, job->alpha, A, job->arbiter != 0) != 0 || rule) ? 1 : 0;
          ovr[i] = dmx_pair_override{-1, -1, -1, 0, 0.0, 0.0};
          if ((sm.flags & DMX_CELL_ORDER_CERTIFIED) && !(sm.flags & kNear) && sm.j_best >= 0 && sm.k_best >= 0)
            ovr[i] = dmx_pair_override{std::min(sm.j_best, sm.k_best), std::max(sm.j_best, sm.k_best), sm.n_best, 0, sm.llk_ab, sm.llk_ba};
          obc[i] = fin.barcodes[c];
          if (host_rows[i]) hostc.push_back((int32_t)c);
        }
        dmx_pair_request rq{};
        rq.n_out = (int32_t)n_out; rq.cells = x.out_cells.data(); rq.barcodes = obc.data(); rq.sample_ids = job->sample_ids;
        rq.host_rows = host_rows.data(); rq.ovr = ovr.data();
        if (int rc = dmx_engine_format_pair(e, &rq, &x.ptext)) return rc;
        dmx_pair_text_info pi{};
        (void)dmx_pair_text_get_info(x.ptext, &pi);
        for (size_t i = 0; i < n_out; ++i) if (pi.cell_flag[i] == 2) hostc.push_back(x.out_cells[i]);   // an unprintable entry: the host's printf prints it
        HIP_TRY(hipSetDevice(e->device));
        for (int32_t c : hostc) {
          x.flagged_grid.emplace_back(nAB);
          HIP_TRY(hipMemcpyAsync(x.flagged_grid.back().data(), e->d_grid + (size_t)c * nAB, sizeof(double) * nAB, hipMemcpyDeviceToHost, e->stream));
        }
        HIP_TRY(hipStreamSynchronize(e->stream));
        for (size_t f = 0; f < hostc.size(); ++f) x.cell_grid[(size_t)hostc[f]] = x.flagged_grid[f].data();
        fetched = (int32_t)hostc.size();
        { std::lock_guard<std::mutex> lk(tm_mu); pair_format_ms += pi.format_ms; pair_bytes += pi.n_bytes; pair_patches += pi.n_patches; pair_host_cells += (int64_t)hostc.size(); }
      } else if (job->write_pair) {              // .pair prints the grid: bring all of it (DMX_PAIR_ON_HOST: the host formatter)
        x.grid.resize(nb1 * nAB); x.summ.resize(nb1);
        if (int rc = dmx_engine_get_doublet(e, x.grid.data(), x.l00.data(), x.summ.data())) return rc;
      } else {                                   // otherwise the K3 records say everything, except for cells flagged as near-ties
        x.sing.resize(nb1 * (size_t)V); x.summ.resize(nb1);
        if (int rc = dmx_engine_get_doublet(e, nullptr, x.l00.data(), x.summ.data())) return rc;
        if (int rc = dmx_engine_get_sing(e, x.sing.data())) return rc;
        x.cell_grid.assign(nb1, nullptr);
        if (job->arbiter) {                       // (dmx::cell_needs: the one predicate the writers use too)
          for (size_t c = 0; c < nb; ++c) if (dmx::cell_needs(x.summ[c], job->alpha, A, true) & dmx::kNeedGrid) {
            x.flagged_grid.emplace_back(nAB);
            HIP_TRY(hipSetDevice(e->device));
            HIP_TRY(hipMemcpyAsync(x.flagged_grid.back().data(), e->d_grid + c * nAB, sizeof(double) * nAB, hipMemcpyDeviceToHost, e->stream));
            ++fetched;
          }
          HIP_TRY(hipStreamSynchronize(e->stream));
          size_t f = 0;
          for (size_t c = 0; c < nb; ++c) if (dmx::cell_needs(x.summ[c], job->alpha, A, true) & dmx::kNeedGrid) x.cell_grid[c] = x.flagged_grid[f++].data();
        }
      }
    }
    if (dev_pl && doublet_ok && job->arbiter) {
      // which barcodes can the writers' arbiter touch?  dmx::cell_needs says (a near-tie flag, a best doublet at alpha = 0.5 without a
      // (resolved) tie-order certificate, a BEST-rule comparison within 1e-7) — the predicate write_doublet_core itself checks before it
      // writes.  Their pairs and read bytes come to the host now.
      const size_t w = (size_t)pl.nrd_width;
      x.tie_cell.assign(nb1, -1);
      x.t_po.assign(1, 0); x.t_ro.assign(1, 0);
      std::vector<int32_t> need;
      for (size_t k = 0; k < nb; ++k) {
        dmx_cell_summary sm = x.summ[k];
        if (sm.n_pairs <= 0) continue;
        if (sm.flags & DMX_CELL_ORDER_RESOLVABLE) (void)dmx::resolve_tie_order(&sm);
        if (!(dmx::cell_needs(sm, job->alpha, A, true) & dmx::kNeedPileup)) continue;
        const int32_t c = sliced ? order[(size_t)x.lo + k] : (int32_t)k;
        x.tie_cell[k] = (int32_t)need.size();
        need.push_back(c);
        x.t_po.push_back(x.t_po.back() + (pl.cell_pair_off[c + 1] - pl.cell_pair_off[c]));
        x.t_ro.push_back(x.t_ro.back() + (pl.cell_read_off[c + 1] - pl.cell_read_off[c]));
      }
      x.t_nrd.resize((size_t)x.t_po.back() * w + 4); x.t_reads.resize((size_t)x.t_ro.back() + 4);
      if (pl.pair_snp) x.t_snp.resize((size_t)x.t_po.back() + 1);
      HIP_TRY(hipSetDevice(e->device));
      for (size_t i = 0; i < need.size(); ++i) {
        const int32_t c = need[i];
        const int64_t p0 = pl.cell_pair_off[c], np_ = pl.cell_pair_off[c + 1] - p0, r0 = pl.cell_read_off[c], nr_ = pl.cell_read_off[c + 1] - r0;
        if (np_ > 0) {
          if (pl.pair_snp) HIP_TRY(hipMemcpyAsync(x.t_snp.data() + x.t_po[i], pl.pair_snp + p0, sizeof(int32_t) * (size_t)np_, hipMemcpyDeviceToHost, e->stream));
          HIP_TRY(hipMemcpyAsync(x.t_nrd.data() + (size_t)x.t_po[i] * w, (const uint8_t*)pl.pair_nrd + (size_t)p0 * w, (size_t)np_ * w, hipMemcpyDeviceToHost, e->stream));
        }
        if (nr_ > 0) HIP_TRY(hipMemcpyAsync(x.t_reads.data() + x.t_ro[i], pl.reads + r0, (size_t)nr_, hipMemcpyDeviceToHost, e->stream));
      }
      HIP_TRY(hipStreamSynchronize(e->stream));
    }
    dmx_kernel_times kt{};
    (void)dmx_engine_last_kernel_times(e, &kt);
    { std::lock_guard<std::mutex> lk(tm_mu); wait_s += secs(t0, clk::now()); kernel_ms += kt.singlet_ms + kt.doublet_ms + kt.reduce_ms; n_fetched += fetched; }
    mark("fetch", r, t0);
    if (trace) fprintf(stderr, "[dmx_demuxlet_run]        range %d kernels: K1 %.1f ms, K2 %.1f ms, K3+K3b %.1f ms\n", r, kt.singlet_ms, kt.doublet_ms, kt.reduce_ms);
    return DMX_OK;
  };
  const std::string pre(job->out_prefix);
  // The .pair file of a range whose rows were formatted on the device: the packed text comes over in pieces through one pinned buffer and goes to
  // write(2) as it is; where the device left a POSTPRB field to the host (dmx_pair_patch: the denormal range, a digit next to a rounding boundary) the
  // field is computed here with the host libm — the expression of :780 / :792 on the barcode's K3 record — and spliced in; barcodes left to the host
  // formatter (cell_flag != 0) have their rows (pair_rows, from write_doublet_core) inserted at their place.
  void* pair_stage = nullptr;
  constexpr int64_t kPairStage = (int64_t)32 << 20;
  struct StageGuard { void** p; ~StageGuard() { if (*p) (void)hipHostFree(*p); } } stage_guard{&pair_stage};
  auto stream_pair = [&](Range& x, const std::vector<std::string>& pair_rows, bool append) -> int {
    dmx_pair_text_info pi{};
    if (int rc = dmx_pair_text_get_info(x.ptext, &pi)) return rc;
    FILE* f = fopen((pre + ".pair").c_str(), append ? "a" : "w");
    if (!f) return set_error(DMX_ERR_IO, "Cannot create %s.single, %s.pair files", job->out_prefix, job->out_prefix);     // :535-536
    struct Closer { FILE* f; ~Closer() { if (f) fclose(f); } } closer{f};
    if (!append) fputs("BARCODE\tSM1.ID\tSM2.ID\tLLK12\tPOSTPRB\n", f);                                               // :570
    if (!pair_stage) HIP_TRY(hipHostMalloc(&pair_stage, (size_t)kPairStage, hipHostMallocDefault));
    char* buf = (char*)pair_stage;
    int64_t wbeg = 0, wend = 0;                  // the piece of the text the buffer holds
    bool ok = true;
    auto put = [&](int64_t a, int64_t b) -> int {     // text bytes [a, b)
      while (a < b) {
        if (a < wbeg || a >= wend) {
          wbeg = a; wend = std::min(pi.n_bytes, a + kPairStage);
          if (int rc = dmx_pair_text_read(x.ptext, wbeg, wend - wbeg, buf)) return rc;
        }
        const int64_t n = std::min(b, wend) - a;
        if (fwrite(buf + (a - wbeg), 1, (size_t)n, f) != (size_t)n) ok = false;
        a += n;
      }
      return DMX_OK;
    };
    int32_t ip = 0;
    std::string field;
    int64_t run_beg = 0;                         // consecutive device-formatted barcodes go out as one run
    for (int32_t oc = 0; oc < pi.n_out; ++oc) {
      const int64_t cb = pi.cell_off[oc], ce = pi.cell_off[oc + 1];
      if (pi.cell_flag[oc] != 0) {
        if (int rc = put(run_beg, cb)) return rc;
        const std::string& rows = pair_rows[(size_t)x.out_cells[(size_t)oc]];
        if (!rows.empty() && fwrite(rows.data(), 1, rows.size(), f) != rows.size()) ok = false;
        run_beg = ce;
        continue;
      }
      while (ip < pi.n_patches && pi.patches[ip].offset < ce) {
        const dmx_pair_patch& pp = pi.patches[ip++];
        if (int rc = put(run_beg, pp.offset)) return rc;
        run_beg = pp.offset;
        const dmx_cell_summary& sm = x.summ[(size_t)x.out_cells[(size_t)pp.out_cell]];
        const double tot = sm.sum_single + sm.sum_double, prior = job->doublet_prior;
        const double p = pp.singlet ? std::exp(pp.value - sm.max_llk) * (1. - prior) / V / tot
                                    : std::exp(pp.value - sm.max_llk) * prior / V / (V - 1) / (A - 1) / tot;
        field.clear();
        dmx::put_general(field, p, 5);
        if (fwrite(field.data(), 1, field.size(), f) != field.size()) ok = false;
      }
    }
    if (int rc = put(run_beg, pi.n_bytes)) return rc;
    closer.f = nullptr;
    if (fclose(f) != 0 || !ok) return set_error(DMX_ERR_IO, "write failed (%s.pair)", job->out_prefix);
    return DMX_OK;
  };
  auto write = [&](int r) -> int {               // append range r's rows (rows of a range are sorted by the writers)
    const clk::time_point t0 = clk::now();
    Range& x = rg[(size_t)r];
    dmx_final_input fin = make_fin(x);
    fin.llks = x.llks.data(); fin.llk0s = x.llk0s.data();
    if (int rc = dmx::write_single_impl(&fin, (pre + ".single").c_str(), r > 0)) return rc;
    if (doublet_ok) {
      fin.llks00 = x.l00.data();
      dmx_pileup tie{};                          // device-resident pileups: the staged pieces of this range's arbiter barcodes
      if (job->arbiter && dev_pl) {
        tie.n_cells = (int32_t)x.t_po.size() - 1; tie.n_snps = pl.n_snps; tie.n_pairs = x.t_po.back(); tie.n_reads = x.t_ro.back();
        tie.cell_pair_off = x.t_po.data(); tie.cell_read_off = x.t_ro.data(); tie.pair_snp = pl.pair_snp ? x.t_snp.data() : nullptr;
        tie.pair_nrd = x.t_nrd.data(); tie.nrd_width = pl.nrd_width; tie.memory = DMX_MEM_HOST; tie.reads = x.t_reads.data();
        fin.tie_pileup = &tie; fin.tie_g = job->g;
      } else if (job->arbiter) { fin.tie_pileup = &pl; fin.tie_g = job->g; }
      dmx::DoubletSource src{};
      if (dev_pl) { if (job->arbiter) src.tie_cell = x.tie_cell.data(); }
      else if (sliced) src.tie_cell = order.data() + x.lo;
      std::vector<std::string> pair_rows;                     // device-formatted .pair: the rows of the barcodes left to the host formatter
      if (x.ptext) { pair_rows.resize((size_t)std::max(fin.n_cells, 1)); src.pair_rows = &pair_rows; }
      if (job->write_pair && !x.ptext) { fin.llksAB = x.grid.data(); src.grid_all = x.grid.data(); src.summary = x.summ.data(); }
      else { src.sing = x.sing.data(); src.summary = x.summ.data(); src.cell_grid = x.cell_grid.data(); }
      if (int rc = dmx::write_doublet_core(&fin, src, job->out_prefix, r > 0, "dmx_demuxlet_run")) return rc;
      if (x.ptext) if (int rc = stream_pair(x, pair_rows, r > 0)) return rc;
    }
    x.release();
    { std::lock_guard<std::mutex> lk(tm_mu); write_s += secs(t0, clk::now()); }
    mark("write", r, t0);
    return DMX_OK;
  };

  // the ranges of a wave belong to different engines: slice + stage (and later fetch) them on one host thread each, so that
  // eight GPUs are not fed one after the other (slicing a 2.8 GB shard out of the CSR is ~0.5 s of host memcpy)
  auto for_ranges = [&](int r0, int r1, const std::function<int(int)>& fn) -> int {
    if (r1 - r0 <= 1) { for (int r = r0; r < r1; ++r) if (int rc = fn(r)) return rc; return DMX_OK; }
    std::vector<int> rcs((size_t)(r1 - r0), DMX_OK);
    std::vector<std::string> msgs((size_t)(r1 - r0));
    std::vector<std::thread> th;
    for (int r = r0; r < r1; ++r)
      th.emplace_back([&, r] { const int rc = fn(r); rcs[(size_t)(r - r0)] = rc; if (rc) msgs[(size_t)(r - r0)] = dmx_last_error(); });
    for (std::thread& t : th) t.join();
    for (size_t i = 0; i < rcs.size(); ++i) if (rcs[i]) return set_error(rcs[i], "%s", msgs[i].c_str());   // the message is thread-local
    return DMX_OK;
  };
  auto wave_lo = [&](int w) { return std::min(R, w * per_wave); };
  // The rows of a fetched wave are arbitrated, formatted and appended by a writer thread, in range order, while this thread
  // stages and fetches the next waves; at most one wave waits for the writer when another is fetched (host memory: the results
  // of two waves, as the byte budget assumes).
  struct Writer {
    std::mutex mu; std::condition_variable cv; std::deque<int> q; bool closing = false; int pending = 0; int rc = DMX_OK; std::string msg; std::thread th;
    void close() { { std::lock_guard<std::mutex> lk(mu); closing = true; } cv.notify_all(); if (th.joinable()) th.join(); }
    ~Writer() { close(); }
  } wr;
  wr.th = std::thread([&] {
    for (;;) {
      int r;
      { std::unique_lock<std::mutex> lk(wr.mu); wr.cv.wait(lk, [&] { return !wr.q.empty() || wr.closing; }); if (wr.q.empty()) return; r = wr.q.front(); wr.q.pop_front(); }
      int rc = DMX_OK;
      { std::lock_guard<std::mutex> lk(wr.mu); rc = wr.rc; }
      if (rc == DMX_OK) { rc = write(r); if (rc) { std::lock_guard<std::mutex> lk(wr.mu); wr.rc = rc; wr.msg = dmx_last_error(); } }
      else rg[(size_t)r].release();
      { std::lock_guard<std::mutex> lk(wr.mu); --wr.pending; }
      wr.cv.notify_all();
    }
  });
  for (int w = 0; w < std::min(waves, nset); ++w) if (int rc = for_ranges(wave_lo(w), wave_lo(w + 1), launch)) return rc;
  for (int w = 0; w < waves; ++w) {
    { std::unique_lock<std::mutex> lk(wr.mu); wr.cv.wait(lk, [&] { return wr.pending <= per_wave; }); if (wr.rc) break; }
    if (int rc = for_ranges(wave_lo(w), wave_lo(w + 1), fetch)) return rc;
    { std::lock_guard<std::mutex> lk(wr.mu); for (int r = wave_lo(w); r < wave_lo(w + 1); ++r) { wr.q.push_back(r); ++wr.pending; } }
    wr.cv.notify_all();
    if (w + nset < waves) if (int rc = for_ranges(wave_lo(w + nset), wave_lo(w + nset + 1), launch)) return rc;   // the freed engines stage wave w + nset
  }
  wr.close();
  if (wr.rc) return set_error(wr.rc, "%s", wr.msg.c_str());
  tm.stage_s = stage_s; tm.wait_s = wait_s; tm.write_s = write_s; tm.kernel_ms = kernel_ms; tm.n_cells_grid_fetched = n_fetched;
  for (int32_t c = 0; c < B; ++c)                  // the droplets the reference counts at :480-524
    if (pl.rd_totl[c] >= job->min_total && pl.rd_uniq[c] >= job->min_uniq && nsnp[(size_t)c] >= job->min_snp) ++tm.n_cells_single;
  tm.total_s = secs(t_begin, clk::now());
  if (job->timing) *job->timing = tm;
  if (getenv("DMX_E2E_TIMING"))
    fprintf(stderr, "{\"dmx_demuxlet_run\": {\"freeze_s\": %.4f, \"setup_s\": %.4f, \"stage_s\": %.4f, \"wait_s\": %.4f, \"write_s\": %.4f, \"total_s\": %.4f, "
                    "\"kernel_ms\": %.3f, \"ranges\": %d, \"engines\": %d, \"cells_grid_fetched\": %d, \"pair_on_device\": %d, \"pair_format_ms\": %.3f, "
                    "\"pair_bytes\": %lld, \"pair_patches\": %lld, \"pair_host_cells\": %lld}}\n",
            tm.freeze_s, tm.setup_s, tm.stage_s, tm.wait_s, tm.write_s, tm.total_s, tm.kernel_ms, tm.n_ranges, tm.n_engines, tm.n_cells_grid_fetched,
            gpu_pair ? 1 : 0, pair_format_ms, (long long)pair_bytes, (long long)pair_patches, (long long)pair_host_cells);
  if (!doublet_ok) return set_error(DMX_ERR_ARG, "dmx_demuxlet_run: the doublet stage needs >= 2 samples and >= 2 alphas (got %d, %d)", V, A);
  return DMX_OK;
}
