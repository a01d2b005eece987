// `.pair` rows (cmd_cram_demuxlet.cpp:772-797, "%s\t%s\t%s\t%.3lf\t%.5lf\t%.5lg\n") formatted on the device — included by dmx_engine.hip.
//
// Why: `--write-pair` prints V + V(V-1)(A-1) rows per barcode (V(V-1)/2 of them at alpha = 0.5, :785) — 2 080 at cfg4, 2.08e8 rows = 10 GB of text
// for the whole job.  Formatted by 16 host threads that is 0.5 s per 12 500-barcode shard beside 0.55 s of kernels (profiles/r06_*): on eight GPUs the
// job would wait for the host six times longer than for the GPUs.  The grid is already in HBM; turning it into text there takes milliseconds and
// the host is left with one D2H and write(2) (11 GB/s into the page cache of this box, profiles/r06_write_rate.txt).
//
// What the device prints and what it does not:
//   * LLK (`%.5lf`): the decimal expansion of a binary64 is finite, so m * 2^e * 10^5 rounded to an integer in exact 128-bit arithmetic (ties to
//     even) gives printf's digits — the algorithm of put_fixed (dmx_host.cpp), which tests/test_host_units.py pins against glibc.  |v| >= 2^43, inf
//     and nan are left to the host (the whole barcode: its rows come from the host formatter, flag 2).
//   * POSTPRB (`%.5lg`): the value is exp(v - maxLLK) * c / tot (:780,:792), and exp() is the host libm's in the reference.  The device evaluates
//     the same chain with its own exp (<= 1 ulp) and K3's sums, and prints the five significant digits ONLY when they cannot depend on the last bits:
//     the scaled value is farther than 4e-13 (relative) from a rounding boundary.  Posteriors that underflowed for certain (v - maxLLK < -800) are "0".
//     Everything else — the denormal range, where glibc's and the device's exp round differently, and the one-in-1e7 value next to a boundary — becomes a
//     PATCH: the row is written without its POSTPRB field and (byte offset, v, barcode, row kind) goes to a list; the host computes that field with
//     its own libm, exactly as the host formatter does, and splices it in while it writes the file.
//   * Barcodes whose grid entries the tie arbiter may replace (dmx::cell_needs != 0, a BEST-rule comparison within 1e-7) are not formatted here at
//     all: the caller marks them (host_rows) and their rows come from the host formatter, inserted at the barcode's offset.  The two certified
//     entries of an alpha = 0.5 best doublet (K3b; resolved on the host where one log() was left open) are passed in and printed as the host does.
// Two passes: lengths (and patch count) per barcode, a scan, then the bytes — the text is packed, in output order, ready for write(2).
#pragma once

namespace dmx_fmt {

struct Ctx {
  const double* grid; const dmx_cell_summary* summ; const double* alpha; double prior; int32_t V, A;
  int32_t n_out; const int32_t* cells; const uint8_t* host_rows; const dmx_pair_override* ovr;
  const uint32_t* rowmap; int32_t n_rows;          // (j << 20) | (k << 8) | a of every row of a barcode, in print order; singlet rows: k == j, a == 0
  const char* bc_pool; const uint32_t* bc_off;     // barcodes of the output cells
  const char* sm_pool; const uint32_t* sm_off;     // sample ids
  const char* al_pool; const uint32_t* al_off;     // "\t%.3lf\t" of every alpha
  const double* pow10;                             // 10^n, n = 0..309
  int64_t* cell_len; uint32_t* cell_npatch;        // pass L out
  const int64_t* cell_off; const uint32_t* cell_poff;   // pass W in (exclusive scans)
  char* text; dmx_pair_patch* patches;
  uint8_t* cell_flag;                              // out: 0 formatted here, 1 left to the caller (host_rows), 2 unprintable value (left to the caller too)
  int32_t max_row;                                 // upper bound of a row's bytes (LDS chunk buffer = 256 rows)
};

// decimal digits of x < 10^9 into dst[0..n): returns n (no leading zeros; "0" for 0)
__device__ __forceinline__ int put_u32(char* dst, uint32_t x, bool write) {
  int n = 1;
  for (uint32_t y = x; y >= 10u; y /= 10u) ++n;
  if (write) { uint32_t y = x; for (int i = n - 1; i >= 0; --i) { dst[i] = (char)('0' + y % 10u); y /= 10u; } }
  return n;
}

// "%.5lf": returns the length, -1 for |v| >= 2^43 / inf / nan
template <bool WRITE>
__device__ int fmt_fixed5(double v, char* dst) {
  const uint64_t bits = (uint64_t)__double_as_longlong(v);
  const uint64_t frac = bits & 0xFFFFFFFFFFFFFull;
  const int ex = (int)((bits >> 52) & 0x7FF);
  if (ex >= 1023 + 43) return -1;
  const uint64_t m = ex ? (frac | (1ull << 52)) : frac;
  const int sh = 1075 - (ex ? ex : 1);                     // |v| = m * 2^-sh, 10 <= sh <= 1074
  const uint64_t lo = m * 100000ull, hi = __umul64hi(m, 100000ull);   // N = m * 10^5 < 2^70
  uint64_t q;
  if (sh >= 72) q = 0;                                     // N < 2^70 <= half of 2^sh / 2: rounds to 0
  else if (sh >= 64) {
    const int s = sh - 64;
    q = hi >> s;
    const uint64_t rem_hi = s ? (hi & ((1ull << s) - 1)) : 0ull, rem_lo = lo;
    const uint64_t half_hi = s ? (1ull << (s - 1)) : 0ull, half_lo = s ? 0ull : (1ull << 63);
    const bool gt = rem_hi > half_hi || (rem_hi == half_hi && rem_lo > half_lo), eq = rem_hi == half_hi && rem_lo == half_lo;
    if (gt || (eq && (q & 1))) ++q;
  } else {
    q = (hi << (64 - sh)) | (lo >> sh);
    const uint64_t rem = lo & ((1ull << sh) - 1), half = 1ull << (sh - 1);
    if (rem > half || (rem == half && (q & 1))) ++q;
  }
  const uint64_t ip = q / 100000ull;
  const uint32_t fp = (uint32_t)(q - ip * 100000ull);
  const uint32_t ip_hi = (uint32_t)(ip / 1000000000ull), ip_lo = (uint32_t)(ip - (uint64_t)ip_hi * 1000000000ull);   // ip < 2^43 < 10^13
  int n = 0;
  if (bits >> 63) { if (WRITE) dst[0] = '-'; n = 1; }
  if (ip_hi) {
    n += put_u32(dst + n, ip_hi, WRITE);
    if (WRITE) { uint32_t y = ip_lo; for (int i = 8; i >= 0; --i) { dst[n + i] = (char)('0' + y % 10u); y /= 10u; } }
    n += 9;
  } else n += put_u32(dst + n, ip_lo, WRITE);
  if (WRITE) {
    dst[n] = '.';
    uint32_t y = fp;
    for (int i = 5; i >= 1; --i) { dst[n + i] = (char)('0' + y % 10u); y /= 10u; }
  }
  return n + 6;
}

// "%.5lg" of a posterior p in [0, 1.5]: the length, or -1 when the digits are not certain here (see the head of this file)
template <bool WRITE>
__device__ int fmt_general5(double p, const double* __restrict__ pow10, char* dst) {
  if (p == 0.0) { if (WRITE) dst[0] = '0'; return 1; }
  if (!(p >= 1e-290 && p <= 1.5)) return -1;
  const int e2 = (int)(((uint64_t)__double_as_longlong(p) >> 52) & 0x7FF) - 1023;
  int X = (int)floor((double)e2 * 0.30102999566398120);   // floor(log10 p) or one below
  double t = p * pow10[4 - X];
  if (t >= 100000.0) { ++X; t = p * pow10[4 - X]; }
  else if (t < 10000.0) { --X; t = p * pow10[4 - X]; }
  if (!(t >= 9999.0 && t < 100001.0)) return -1;         // (cannot happen; leave anything odd to the host)
  const double fl = floor(t), fr = t - fl;
  if (fabs(fr - 0.5) < t * 4e-13) return -1;              // a rounding boundary within the margin: the host's libm decides
  uint32_t D = (uint32_t)(fr > 0.5 ? fl + 1.0 : fl);
  if (D < 10000u) return -1;                               // t in [9999, 10000) rounded down: the scale was off by one ulp — leave it
  if (D >= 100000u) { D = 10000u; ++X; }
  int nd = 5;
  while (nd > 1 && D % 10u == 0u) { D /= 10u; --nd; }     // %g strips trailing zeros
  char dg[5];
  { uint32_t y = D; for (int i = nd - 1; i >= 0; --i) { dg[i] = (char)('0' + y % 10u); y /= 10u; } }
  int n = 0;
  if (X >= -4) {                                           // fixed notation (p <= 1.5: X <= 0)
    if (X == 0) {
      if (WRITE) dst[0] = dg[0];
      n = 1;
      if (nd > 1) { if (WRITE) { dst[1] = '.'; for (int i = 1; i < nd; ++i) dst[1 + i] = dg[i]; } n = 1 + nd; }
    } else {
      const int z = -X - 1;                                // zeros between the point and the first digit
      if (WRITE) { dst[0] = '0'; dst[1] = '.'; for (int i = 0; i < z; ++i) dst[2 + i] = '0'; for (int i = 0; i < nd; ++i) dst[2 + z + i] = dg[i]; }
      n = 2 + z + nd;
    }
  } else {                                                 // d[.ddd]e-XX
    if (WRITE) dst[0] = dg[0];
    n = 1;
    if (nd > 1) { if (WRITE) { dst[1] = '.'; for (int i = 1; i < nd; ++i) dst[1 + i] = dg[i]; } n = 1 + nd; }
    const uint32_t ax = (uint32_t)(-X);
    if (WRITE) { dst[n] = 'e'; dst[n + 1] = '-'; }
    n += 2;
    if (ax < 10u) { if (WRITE) { dst[n] = '0'; dst[n + 1] = (char)('0' + ax); } n += 2; }
    else n += put_u32(dst + n, ax, WRITE);
  }
  return n;
}

struct CellCtx { const double* G; double max_llk, tot; int32_t oa, ob, on; double v_ab, v_ba; const char* bc; uint32_t bc_len; };

template <bool WRITE>
__device__ __forceinline__ int copy_str(char* dst, const char* __restrict__ src, uint32_t n) {
  if (WRITE) for (uint32_t i = 0; i < n; ++i) dst[i] = src[i];
  return (int)n;
}

// One row.  Returns its length; *post_at = offset of the POSTPRB field when it is left to the host (a patch), else -1; *bad when LLK is unprintable here.
template <bool WRITE>
__device__ int format_row(const Ctx& c, const CellCtx& cc, uint32_t rm, char* dst, int* post_at, double* v_out, bool* bad) {
  const int j = (int)(rm >> 20), k = (int)((rm >> 8) & 0xFFFu), a = (int)(rm & 0xFFu);
  const bool singlet = a == 0;
  const int V = c.V, A = c.A;
  double v = cc.G[((size_t)j * V + (singlet ? 0 : k)) * A + a];
  if (!singlet && a == cc.on) {
    if (j == cc.oa && k == cc.ob) v = cc.v_ab;
    else if (j == cc.ob && k == cc.oa) v = cc.v_ba;
  }
  int n = 0;
  n += copy_str<WRITE>(dst + n, cc.bc, cc.bc_len);
  if (WRITE) dst[n] = '\t';
  ++n;
  n += copy_str<WRITE>(dst + n, c.sm_pool + c.sm_off[j], c.sm_off[j + 1] - c.sm_off[j]);
  if (WRITE) dst[n] = '\t';
  ++n;
  n += copy_str<WRITE>(dst + n, c.sm_pool + c.sm_off[k], c.sm_off[k + 1] - c.sm_off[k]);
  n += copy_str<WRITE>(dst + n, c.al_pool + c.al_off[a], c.al_off[a + 1] - c.al_off[a]);
  const int nf = fmt_fixed5<WRITE>(v, dst + n);
  if (nf < 0) { *bad = true; *post_at = -1; return 0; }
  n += nf;
  if (WRITE) dst[n] = '\t';
  ++n;
  // the posterior of :780 / :792, operation by operation
  const double x = v - cc.max_llk;
  int np = -1;
  if (x < -800.0) { if (WRITE) dst[n] = '0'; np = 1; }                       // exp underflows to 0 in any libm, and 0 stays 0 down the chain
  else if (x == x) {
    const double e = exp(x);
    const double p = singlet ? e * (1. - c.prior) / V / cc.tot : e * c.prior / V / (V - 1) / (A - 1) / cc.tot;
    np = fmt_general5<WRITE>(p, c.pow10, dst + n);
  }
  *post_at = -1;
  if (np < 0) { *post_at = n; *v_out = v; np = 0; }
  n += np;
  if (WRITE) dst[n] = '\n';
  return n + 1;
}

__device__ __forceinline__ bool load_cell(const Ctx& c, int32_t oc, CellCtx* cc) {
  const int32_t cell = c.cells[oc];
  const dmx_cell_summary& sm = c.summ[cell];
  cc->G = c.grid + (size_t)cell * c.V * c.V * c.A;
  cc->max_llk = sm.max_llk;
  cc->tot = sm.sum_single + sm.sum_double;
  cc->oa = cc->ob = cc->on = -1; cc->v_ab = cc->v_ba = 0.0;
  if (c.ovr && c.ovr[oc].n >= 0) { cc->oa = c.ovr[oc].a; cc->ob = c.ovr[oc].b; cc->on = c.ovr[oc].n; cc->v_ab = c.ovr[oc].llk_ab; cc->v_ba = c.ovr[oc].llk_ba; }
  cc->bc = c.bc_pool + c.bc_off[oc];
  cc->bc_len = c.bc_off[oc + 1] - c.bc_off[oc];
  return true;
}

constexpr int kFmtThreads = 256;

// Pass L: bytes and patches of every output barcode (one workgroup per barcode)
__global__ __launch_bounds__(kFmtThreads) void k_pair_lengths(Ctx c) {
  const int32_t oc = blockIdx.x;
  __shared__ unsigned long long s_len;
  __shared__ uint32_t s_np, s_bad;
  if (threadIdx.x == 0) { s_len = 0ull; s_np = 0u; s_bad = 0u; }
  __syncthreads();
  if (c.host_rows && c.host_rows[oc]) {
    if (threadIdx.x == 0) { c.cell_len[oc] = 0; c.cell_npatch[oc] = 0u; c.cell_flag[oc] = 1; }
    return;
  }
  CellCtx cc;
  load_cell(c, oc, &cc);
  unsigned long long len = 0ull; uint32_t np = 0u; bool bad = false;
  for (int32_t r = threadIdx.x; r < c.n_rows; r += kFmtThreads) {
    int post_at; double v;
    len += (unsigned long long)format_row<false>(c, cc, c.rowmap[r], nullptr, &post_at, &v, &bad);
    np += post_at >= 0 ? 1u : 0u;
  }
  atomicAdd(&s_len, len);
  atomicAdd(&s_np, np);
  if (bad) atomicOr(&s_bad, 1u);
  __syncthreads();
  if (threadIdx.x == 0) {
    const bool b = s_bad != 0u;
    c.cell_len[oc] = b ? 0 : (int64_t)s_len;
    c.cell_npatch[oc] = b ? 0u : s_np;
    c.cell_flag[oc] = b ? 2 : 0;
  }
}

// exclusive scans of the per-barcode lengths and patch counts (n_out <= a few 100 000: one workgroup)
__global__ __launch_bounds__(1024) void k_pair_scan(const int64_t* __restrict__ len, const uint32_t* __restrict__ np, int32_t n,
                                                    int64_t* __restrict__ off, uint32_t* __restrict__ poff) {
  __shared__ long long s_a[1024];
  __shared__ uint32_t s_b[1024];
  const int t = threadIdx.x;
  const int32_t per = (n + 1023) / 1024, lo = min(n, t * per), hi = min(n, lo + per);
  long long a = 0; uint32_t b = 0;
  for (int32_t i = lo; i < hi; ++i) { a += len[i]; b += np[i]; }
  s_a[t] = a; s_b[t] = b;
  __syncthreads();
  if (t == 0) {
    long long ra = 0; uint32_t rb = 0;
    for (int i = 0; i < 1024; ++i) { const long long xa = s_a[i]; const uint32_t xb = s_b[i]; s_a[i] = ra; s_b[i] = rb; ra += xa; rb += xb; }
    off[n] = ra; poff[n] = rb;
  }
  __syncthreads();
  a = s_a[t]; b = s_b[t];
  for (int32_t i = lo; i < hi; ++i) { off[i] = a; poff[i] = b; a += len[i]; b += np[i]; }
}

// Pass W: the bytes.  A workgroup walks its barcode's rows in chunks of 256: lengths, a scan, the rows into an LDS buffer at their offsets, the buffer
// to the packed text with coalesced stores.
__global__ __launch_bounds__(kFmtThreads) void k_pair_write(Ctx c) {
  extern __shared__ char s_chunk[];                // [256 * max_row]
  __shared__ uint32_t s_wsum[kFmtThreads / 64], s_psum[kFmtThreads / 64];
  const int32_t oc = blockIdx.x;
  if (c.cell_flag[oc] != 0) return;
  CellCtx cc;
  load_cell(c, oc, &cc);
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  int64_t base = c.cell_off[oc];
  uint32_t pbase = c.cell_poff[oc];
  for (int32_t r0 = 0; r0 < c.n_rows; r0 += kFmtThreads) {
    const int32_t r = r0 + t;
    const bool on = r < c.n_rows;
    const uint32_t rm = on ? c.rowmap[r] : 0u;
    int post_at = -1; double v = 0.0; bool bad = false;
    const uint32_t len = on ? (uint32_t)format_row<false>(c, cc, rm, nullptr, &post_at, &v, &bad) : 0u;
    const uint32_t isp = post_at >= 0 ? 1u : 0u;
    // exclusive scans over the chunk's 256 rows: bytes and patches
    const uint32_t incl = seg_scan_incl<64>(len), pincl = seg_scan_incl<64>(isp);
    if (lane == 63) { s_wsum[w] = incl; s_psum[w] = pincl; }
    __syncthreads();
    uint32_t off = incl - len, poff = pincl - isp, total = 0u, ptotal = 0u;
    for (int i = 0; i < kFmtThreads / 64; ++i) { if (i < w) { off += s_wsum[i]; poff += s_psum[i]; } total += s_wsum[i]; ptotal += s_psum[i]; }
    if (on) {
      format_row<true>(c, cc, rm, s_chunk + off, &post_at, &v, &bad);
      if (post_at >= 0) {
        dmx_pair_patch pp;
        pp.offset = base + (int64_t)off + post_at; pp.value = v; pp.out_cell = oc; pp.singlet = (rm & 0xFFu) == 0u ? 1 : 0;
        c.patches[pbase + poff] = pp;
      }
    }
    __syncthreads();
    for (uint32_t i = t; i < total; i += kFmtThreads) c.text[base + i] = s_chunk[i];
    base += total; pbase += ptotal;
    __syncthreads();
  }
}

}  // namespace dmx_fmt
