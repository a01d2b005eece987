// Host half of libdmx: error plumbing, phred LUT (a2), genotype-field transforms (a3), the UMI-deduplicated pileup
// store (a1), the finaliser/writers (a6, a10..a14) and the tie arbiter.  Plain C++17, no HIP in this file.
//
// Reference lines are cited per function (paths relative to statgen/demuxlet).  Nothing here includes, links or calls
// anything under oracle/.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdarg>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <memory>
#include <mutex>
#include <numeric>
#include <thread>
#include <unordered_map>

#include "dmx_internal.hpp"
#include "dmx_log.hpp"

namespace dmx {

static thread_local std::string g_last_error;

int set_error(int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_last_error = buf;
  return code;
}

void build_read_lut(const double mat[256], const double err[256], ReadLut* out) {
  for (int q = 0; q < 128; ++q) {
    out->mat[q] = mat[q];
    out->e3[q] = err[q] / 3.0;            // phredConv.phred2Err[bq]/3.0   (cmd_cram_demuxlet.cpp:437,:439,:606,:607)
    out->het[q] = 0.5 - err[q] / 3.0;     // (0.5 - phredConv.phred2Err[bq]/3.0)   (:438)
  }
}

void build_singlet_tables(const ReadLut& lut, SingletTables* out) {
  auto finish = [](const double in[3], double o[3]) {
    double a = in[0] + 1e-6, b = in[1] + 1e-6, c = in[2] + 1e-6;   // :446-448
    const double tmp = a + b + c;
    o[0] = a / tmp; o[1] = b / tmp; o[2] = c / tmp;                // :449-452
  };
  for (int byte = 0; byte < 256; ++byte) {
    const int bq = byte & 127;
    const bool alt = (byte >> 7) != 0;
    double G0 = 1.0, G1 = 1.0, G2 = 1.0;                           // :427
    G0 *= alt ? lut.e3[bq] : lut.mat[bq];                          // :437
    G1 *= lut.het[bq];                                             // :438
    G2 *= alt ? lut.mat[bq] : lut.e3[bq];                          // :439
    const double tmp = G0 + G1 + G2;                               // :440
    out->first[byte][0] = G0 / tmp; out->first[byte][1] = G1 / tmp; out->first[byte][2] = G2 / tmp;   // :441-443
    finish(out->first[byte], out->final1[byte]);
  }
  const double ones[3] = {1.0, 1.0, 1.0};
  finish(ones, out->final1[256]);
}

void build_pair_tables(const ReadLut& lut, const SingletTables& st, PairTables* out) {
  for (int c0 = 0; c0 < 128; ++c0)
    for (int c1 = 0; c1 < 128; ++c1) {
      const int byte0 = ((c0 >> 6) << 7) | (c0 & 63);
      const int bq = c1 & 63;
      const bool alt = (c1 >> 6) != 0;
      double G0 = st.first[byte0][0], G1 = st.first[byte0][1], G2 = st.first[byte0][2];   // state after the first read
      G0 *= alt ? lut.e3[bq] : lut.mat[bq];                          // :437
      G1 *= lut.het[bq];                                             // :438
      G2 *= alt ? lut.mat[bq] : lut.e3[bq];                          // :439
      const double tmp = G0 + G1 + G2;                               // :440
      double* s2 = out->second[c0 * 128 + c1];
      s2[0] = G0 / tmp; s2[1] = G1 / tmp; s2[2] = G2 / tmp; s2[3] = 0.0;   // :441-443
      const double a = s2[0] + 1e-6, b = s2[1] + 1e-6, c = s2[2] + 1e-6;   // :446-448
      const double t2 = a + b + c;
      double* f2 = out->final2[c0 * 128 + c1];
      f2[0] = a / t2; f2[1] = b / t2; f2[2] = c / t2; f2[3] = 0.0;         // :449-452
    }
}

const TripleTables& build_triple_tables(const ReadLut& lut, const PairTables& pt) {
  static std::mutex mu;
  static TripleTables cache;
  static ReadLut cache_key;
  static bool have = false;
  std::lock_guard<std::mutex> lk(mu);
  if (have && std::memcmp(&cache_key, &lut, sizeof lut) == 0) return cache;
  const size_t n = (size_t)kTripleCodes * kTripleCodes * kTripleCodes;
  cache.third.assign(n * 4, 0.0); cache.final3.assign(n * 4, 0.0);
  for (int c0 = 0; c0 < kTripleCodes; ++c0)
    for (int c1 = 0; c1 < kTripleCodes; ++c1) {
      const int p0 = ((c0 / kTripleBq) << 6) | (c0 % kTripleBq), p1 = ((c1 / kTripleBq) << 6) | (c1 % kTripleBq);   // PairTables codes
      const double* s2 = pt.second[p0 * 128 + p1];
      for (int c2 = 0; c2 < kTripleCodes; ++c2) {
        const int bq = c2 % kTripleBq;
        const bool alt = c2 >= kTripleBq;
        double G0 = s2[0], G1 = s2[1], G2 = s2[2];
        G0 *= alt ? lut.e3[bq] : lut.mat[bq];                        // :437
        G1 *= lut.het[bq];                                           // :438
        G2 *= alt ? lut.mat[bq] : lut.e3[bq];                        // :439
        const double tmp = G0 + G1 + G2;                             // :440
        const size_t i = (((size_t)c0 * kTripleCodes + c1) * kTripleCodes + c2) * 4;
        double* t3 = &cache.third[i];
        t3[0] = G0 / tmp; t3[1] = G1 / tmp; t3[2] = G2 / tmp;        // :441-443
        const double a = t3[0] + 1e-6, b = t3[1] + 1e-6, c = t3[2] + 1e-6;   // :446-448
        const double t = a + b + c;
        double* f3 = &cache.final3[i];
        f3[0] = a / t; f3[1] = b / t; f3[2] = c / t;                 // :449-452
      }
    }
  cache_key = lut; have = true;
  return cache;
}

}  // namespace dmx

using dmx::set_error;

extern "C" int dmx_abi_version(void) { return DMX_ABI_VERSION; }

// DMX_CELL_ORDER_RESOLVABLE (K3b): each accumulator of the best alpha = 0.5 pair is one of two doubles the device computed, and
// which one depends on what the reference's libm returns for one log().  The host's libm is that libm: ask it.  On success the
// record reads like a certified one (order, llk12, llk_ab, llk_ba are the reference's); false = the libm's answer is neither
// candidate (the tie arbiter decides).
bool dmx::resolve_tie_order(dmx_cell_summary* r) {
  double ab = r->llk_ab, ba = r->llk_ba;
  if (r->llk_ab_alt != r->llk_ab) {
    const double L = std::log(r->ev_x_ab);
    if (L == r->ev_t_ab) ab = r->llk_ab;
    else if (L == std::nextafter(r->ev_t_ab, HUGE_VAL)) ab = r->llk_ab_alt;
    else return false;
  }
  if (r->llk_ba_alt != r->llk_ba) {
    const double L = std::log(r->ev_x_ba);
    if (L == r->ev_t_ba) ba = r->llk_ba;
    else if (L == std::nextafter(r->ev_t_ba, HUGE_VAL)) ba = r->llk_ba_alt;
    else return false;
  }
  const int32_t ia = std::min(r->j_best, r->k_best), ib = std::max(r->j_best, r->k_best);
  const bool ba_wins = ab < ba;               // the reference's strict-< scan (:799-814) meets (a,b) first
  const int32_t nj = ba_wins ? ib : ia, nk = ba_wins ? ia : ib;
  if (nj != r->j_best) { std::swap(r->llk1, r->llk2); std::swap(r->llk10, r->llk20); }
  r->j_best = nj; r->k_best = nk;
  r->llk12 = ba_wins ? ba : ab;
  r->llk_ab = ab; r->llk_ba = ba; r->llk_ab_alt = ab; r->llk_ba_alt = ba;
  r->flags = (r->flags | DMX_CELL_ORDER_CERTIFIED) & ~DMX_CELL_ORDER_RESOLVABLE;
  return true;
}

extern "C" int dmx_resolve_tie_order(dmx_cell_summary* summary, int64_t n) {
  if (n < 0 || (n && !summary)) return set_error(DMX_ERR_ARG, "dmx_resolve_tie_order: bad arguments");
  int left = 0;
  for (int64_t i = 0; i < n; ++i)
    if ((summary[i].flags & DMX_CELL_ORDER_RESOLVABLE) && !dmx::resolve_tie_order(&summary[i])) ++left;
  return left;
}

bool dmx::libm_log_within_brackets() {
  static const bool ok = [] {
    uint64_t st = 0x243F6A8885A308D3ull;
    auto next = [&] { uint64_t z = (st += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); };
    for (int i = 0; i < 400000; ++i) {
      const double u = (next() >> 11) * (1.0 / 9007199254740992.0);
      double x;
      switch (i & 3) {                           // likelihood-like arguments: near 1, (0.3, 1), tiny, and across many binades
        case 0: x = 0.93 + 0.14 * u; break;
        case 1: x = 0.3 + 0.7 * u; break;
        case 2: x = std::ldexp(0.5 + 0.5 * u, -(int)(next() % 60)); break;
        default: x = 1e-6 + u * 1e-3; break;
      }
      uint64_t b;
      std::memcpy(&b, &x, sizeof b);
      double lo, hi;
      dmx_log_bracket((uint32_t)(b >> 32), (uint32_t)b, dmx_log_table_host, dmx_log_table_lo_host, &lo, &hi);
      const double y = std::log(x);
      if (!(y >= lo && y <= hi)) return false;
    }
    return true;
  }();
  return ok;
}

extern "C" int dmx_debug_log_dd(const double* x, double* hi, double* lo, double* t_lo, double* t_hi, int64_t n) {
  if (!x || !hi || !lo || !t_lo || !t_hi || n < 0) return set_error(DMX_ERR_ARG, "dmx_debug_log_dd: bad arguments");
  for (int64_t i = 0; i < n; ++i) {
    uint64_t b;
    std::memcpy(&b, &x[i], sizeof b);
    const uint32_t hw = (uint32_t)(b >> 32), lw = (uint32_t)b;
    if ((hw - 0x00100000u) >= 0x7FE00000u) return set_error(DMX_ERR_ARG, "dmx_debug_log_dd: x[%lld] is not a normal positive number", (long long)i);
    dmx_log_dd(hw, lw, dmx_log_table_host, dmx_log_table_lo_host, &hi[i], &lo[i]);
    dmx_log_bracket(hw, lw, dmx_log_table_host, dmx_log_table_lo_host, &t_lo[i], &t_hi[i]);
  }
  return DMX_OK;
}
extern "C" const char* dmx_last_error(void) { return dmx::g_last_error.c_str(); }

// ---------------------------------------------------------------------------------------------------------------------
// a2  (PhredHelper.cpp:24-40)
extern "C" int dmx_phred_tables(double mat[256], double err[256]) {
  if (!mat || !err) return set_error(DMX_ERR_ARG, "dmx_phred_tables: null output");
  for (int q = 0; q < 256; ++q) {
    err[q] = (q > 1) ? std::pow(0.1, q * 0.1) : 0.75;
    mat[q] = 1. - err[q];
  }
  return DMX_OK;
}
static double phred_to_prob(uint32_t phred) {     // phredConverter::toProb, PhredHelper.h:45 (phred2Prob has no 0.75 floor)
  static double tab[256];
  static bool ready = false;
  if (!ready) { for (int q = 0; q < 256; ++q) tab[q] = std::pow(0.1, q * 0.1); ready = true; }
  return phred > 255 ? tab[255] : tab[phred];
}

// ---------------------------------------------------------------------------------------------------------------------
// a3  genotype-field transforms; biallelic diploid records (vfilt.maxAlleles = 2, cmd_cram_demuxlet.cpp:26,:106).
// Everything lands in float32 exactly where the reference rounds to float (bcf_filtered_reader.h:78).
static constexpr int kGenos = 3, kAlleles = 2;

extern "C" int dmx_geno_from_gt(const int32_t* alleles, int32_t nv, double gt_error, float* out) {
  if (!alleles || !out || nv < 0) return set_error(DMX_ERR_ARG, "dmx_geno_from_gt: bad arguments");
  // allele counts over the selected samples: bcf_filtered_reader.cpp:230-240
  double ac[kAlleles] = {0, 0};
  int32_t an = 0;
  for (int32_t i = 0; i < 2 * nv; ++i) {
    const int32_t a = alleles[i];
    if (a >= kAlleles) return set_error(DMX_ERR_ARG, "dmx_geno_from_gt: allele index %d on a biallelic record", a);
    if (a >= 0) { ++an; ac[a] += 1; }
  }
  // HWE prior used for missing genotypes (:381-388); evaluated left to right as the reference's expression is
  float hwe[kGenos];
  {
    int l = 0;
    for (int j = 0; j < kAlleles; ++j)
      for (int k = 0; k <= j; ++k, ++l)
        hwe[l] = (float)((j == k ? 1.0 : 2.0) * (ac[j] + 1.0 / kAlleles) / (an + 1.0) * (ac[k] + 1.0 / kAlleles) / (an + 1.0));
  }
  const float hit = (float)(1.0 - gt_error), miss = (float)(gt_error / (kGenos - 1.0));   // :399
  for (int32_t i = 0; i < nv; ++i) {
    const int32_t a1 = alleles[2 * i], a2 = alleles[2 * i + 1];
    float* o = out + 3 * (size_t)i;
    if (a1 < 0 || a2 < 0) { o[0] = hwe[0]; o[1] = hwe[1]; o[2] = hwe[2]; continue; }     // .h:144-149 -> -1
    const int32_t gt = a1 + a2;          // bcf_alleles2gt for alleles in {0,1}: (0,0)->0 (0,1)->1 (1,1)->2
    for (int g = 0; g < kGenos; ++g) o[g] = (g == gt) ? hit : miss;
  }
  return DMX_OK;
}

extern "C" int dmx_geno_from_pl(const int32_t* pl, int32_t nv, float* out) {
  if (!pl || !out || nv < 0) return set_error(DMX_ERR_ARG, "dmx_geno_from_pl: bad arguments");
  // 10 EM rounds on the allele frequencies from a uniform start (bcf_filtered_reader.cpp:255-311); genotype order
  // l = 0:(0,0) 1:(1,0) 2:(1,1) with HWE weights 1,2,1; the posterior of the LAST round is what is kept (:305-308).
  double af[kAlleles] = {1.0 / kAlleles, 1.0 / kAlleles};
  std::vector<double> like((size_t)nv * kGenos);
  for (size_t i = 0; i < like.size(); ++i) like[i] = phred_to_prob((uint32_t)pl[i]);
  for (int it = 0; it < 10; ++it) {
    double next[kAlleles] = {0, 0};
    int32_t an = 0;
    const bool last = (it == 9);
    for (int32_t i = 0; i < nv; ++i) {
      const double* L = &like[(size_t)i * kGenos];
      double gp[kGenos];
      double sum = 0;
      sum += (gp[0] = 1 * af[0] * af[0] * L[0]);       // (j==k ? 1 : 2) * acs[j] * acs[k] * toProb(pl)   (:279)
      sum += (gp[1] = 2 * af[1] * af[0] * L[1]);
      sum += (gp[2] = 1 * af[1] * af[1] * L[2]);
      gp[0] /= sum; next[0] += gp[0]; next[0] += gp[0];   // newacs[j] += gp; newacs[k] += gp   (:284-286)
      gp[1] /= sum; next[1] += gp[1]; next[0] += gp[1];
      gp[2] /= sum; next[1] += gp[2]; next[1] += gp[2];
      an += 2;
      if (last) { float* o = out + 3 * (size_t)i; o[0] = (float)gp[0]; o[1] = (float)gp[1]; o[2] = (float)gp[2]; }
    }
    af[0] = next[0] / an; af[1] = next[1] / an;           // :310-311
  }
  return DMX_OK;
}

extern "C" int dmx_geno_from_gp(const float* gp, int32_t nv, double gt_error, float* out) {
  if (!gp || !out || nv < 0) return set_error(DMX_ERR_ARG, "dmx_geno_from_gp: bad arguments");
  // pseudo-sample: HWE at uniform allele frequency, ((i==j)?1:2)/nalleles^2 in float (bcf_filtered_reader.cpp:421-425)
  float mean[kGenos] = {(float)(1.0 / (float)(kAlleles * kAlleles)), (float)(2.0 / (float)(kAlleles * kAlleles)),
                        (float)(1.0 / (float)(kAlleles * kAlleles))};
  for (int32_t i = 0; i < nv; ++i) {       // per-sample float normalisation, accumulated into the mean (:428-439)
    const float* in = gp + 3 * (size_t)i;
    float* o = out + 3 * (size_t)i;
    float s = 0;
    s += in[0]; s += in[1]; s += in[2];
    for (int g = 0; g < kGenos; ++g) { o[g] = in[g] / s; mean[g] += o[g]; }
  }
  const int32_t denom = (int32_t)(nv + 1.0);            // gpSums[j] /= (int32_t)(sm_icols.size()+1.0)   (:442)
  for (int g = 0; g < kGenos; ++g) mean[g] /= denom;
  for (int32_t i = 0; i < nv; ++i) {                    // (1-e)*gp + e*mean in double, stored as float (:448)
    float* o = out + 3 * (size_t)i;
    for (int g = 0; g < kGenos; ++g) o[g] = (float)((1.0 - gt_error) * o[g] + gt_error * mean[g]);
  }
  return DMX_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// a1  pileup store.  The reference nests std::maps (snp -> cell -> umi -> packed word, plus a mirror cell -> snp);
// what is observable downstream is (i) first observation of a (snp,cell,umi) key wins, later ones only count
// (sc_drop_seq.cpp:44,53,57), (ii) the per-cell counters (:39,:75 and cmd_cram_demuxlet.cpp:295), (iii) iteration order.
// Here: an append-only observation log with an open-addressing index for the "seen before?" test, and one sort at
// freeze time that emits the GPU's CSR directly.
namespace { int host_threads(); }   // defined with the writers below

struct dmx_store {
  // umi: offset into umi_pool in the low 40 bits (the pool passes 4 GiB at ~3.5e8 observations with 12-byte UMIs), length above
  struct Obs {
    int32_t cell, snp; uint64_t umi; uint8_t allele, bq; uint32_t count;
    uint64_t umi_off() const { return umi & 0xFFFFFFFFFFull; }
    uint32_t umi_len() const { return (uint32_t)(umi >> 40); }
  };
  std::vector<std::string> barcodes;
  std::vector<uint64_t> bc_index;      // open addressing over barcodes: (hash's high 32 bits << 32) | cell id; looked up with
                                       // the caller's C string as it is (no temporary std::string per read)
  // Per-cell counters in blocks that never move: dmx_store_add_cell (new cells, RD.TOTL) may run on one host thread while
  // dmx_store_add_batch (RD.PASS / RD.UNIQ of cells that already existed when the batch was built) runs on others — the `demuxlet`
  // binary's scan overlaps the two for consecutive windows of reads.
  struct Counters {
    static constexpr size_t kBlock = 1 << 16, kBlocks = 1 << 15;   // 2^31 cells
    std::unique_ptr<std::unique_ptr<int32_t[]>[]> blk{new std::unique_ptr<int32_t[]>[kBlocks]};
    size_t n = 0;
    void push_back(int32_t v) { if (n % kBlock == 0) blk[n / kBlock].reset(new int32_t[kBlock]()); blk[n / kBlock][n % kBlock] = v; ++n; }
    int32_t& operator[](size_t i) { return blk[i / kBlock][i % kBlock]; }
    int32_t operator[](size_t i) const { return blk[i / kBlock][i % kBlock]; }
    void copy_to(std::vector<int32_t>& out) const { out.resize(n); for (size_t i = 0; i < n; ++i) out[i] = (*this)[i]; }
  };
  Counters totl, pass, uniq;
  std::vector<int32_t> f_totl, f_pass, f_uniq;   // contiguous copies made by dmx_store_freeze (what dmx_pileup points at)
  std::atomic<int32_t> n_cells_pub{0};           // barcodes.size(), readable while another thread adds cells
  std::atomic<int32_t> n_snps{0};                 // (added to by dmx_store_add_snp while a batch of the window before is being inserted)
  // The observations live in kShards sub-stores selected by the cell id: a (snp, cell, umi) key belongs to exactly one of them, so
  // "first observation wins" (sc_drop_seq.cpp:44,53,57) only needs the order WITHIN a shard — dmx_store_add_batch inserts a whole
  // batch with one host thread per group of shards and gives the results of the same calls made one by one.
  static constexpr int kShards = 64;
  struct Shard {
    std::vector<Obs> obs;
    std::string umi_pool;
    std::vector<uint64_t> index;       // open addressing over obs: (hash's high 32 bits << 32) | obs id; kEmpty = free.  The
                                       // fingerprint settles almost every probe without touching obs (one cache miss per call)
    void rehash(size_t cap) {
      index.assign(cap, kEmpty);
      for (size_t i = 0; i < obs.size(); ++i) {
        const Obs& o = obs[i];
        const uint64_t h = hash(o.cell, o.snp, umi_pool.data() + o.umi_off(), o.umi_len());
        size_t p = h & (cap - 1);
        while (index[p] != kEmpty) p = (p + 1) & (cap - 1);
        index[p] = (h & 0xFFFFFFFF00000000ull) | (uint64_t)i;
      }
    }
    // one observation: 1 = new key, 0 = duplicate (its count moves), -1 = a limit was hit
    int add(int32_t cell, int32_t snp, const char* umi, size_t len, int allele, int bq) {
      const size_t cap = index.size();
      const uint64_t h = hash(cell, snp, umi, len);
      size_t p = h & (cap - 1);
      for (; index[p] != kEmpty; p = (p + 1) & (cap - 1)) {
        if ((index[p] ^ h) >> 32) continue;                                        // another key's fingerprint
        Obs& o = obs[(size_t)(index[p] & 0xFFFFFFFFull)];
        if (o.cell == cell && o.snp == snp && o.umi_len() == len && std::memcmp(umi_pool.data() + o.umi_off(), umi, len) == 0) {
          ++o.count;                                                               // :57 duplicate: only the count moves
          return 0;
        }
      }
      if (obs.size() >= 0xFFFFFFF0ull || len >= (1u << 24) || umi_pool.size() + len >= (1ull << 40)) return -1;
      Obs o{cell, snp, (uint64_t)umi_pool.size() | ((uint64_t)len << 40), (uint8_t)allele, (uint8_t)bq, 1u};
      umi_pool.append(umi, len);
      index[p] = (h & 0xFFFFFFFF00000000ull) | (uint64_t)obs.size();
      obs.push_back(o);
      if (obs.size() * 2 > cap) rehash(cap * 2);
      return 1;
    }
  };
  Shard shard[kShards];
  static constexpr uint64_t kEmpty = ~0ull;
  // frozen CSR
  std::atomic<bool> frozen{false};
  std::vector<int64_t> cell_pair_off, cell_read_off;
  std::vector<int32_t> pair_snp;
  std::vector<uint8_t> pair_nrd_bytes;
  int32_t nrd_width = 1;
  std::vector<uint8_t> reads;

  static uint64_t hash(int32_t cell, int32_t snp, const char* umi, size_t len) {
    uint64_t h = 0x9E3779B97F4A7C15ULL ^ ((uint64_t)(uint32_t)cell << 32 | (uint32_t)snp);
    h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ULL;
    for (size_t i = 0; i < len; ++i) { h ^= (unsigned char)umi[i]; h *= 0x100000001B3ULL; }
    h ^= h >> 32;
    return h;
  }
  static uint64_t hash_str(const char* z, size_t* len) {
    uint64_t h = 0xCBF29CE484222325ULL;
    size_t n = 0;
    for (; z[n]; ++n) { h ^= (unsigned char)z[n]; h *= 0x100000001B3ULL; }
    *len = n;
    h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ULL; h ^= h >> 32;
    return h;
  }
  void bc_rehash(size_t cap) {
    bc_index.assign(cap, kEmpty);
    for (size_t i = 0; i < barcodes.size(); ++i) {
      size_t len;
      const uint64_t h = hash_str(barcodes[i].c_str(), &len);
      size_t p = h & (cap - 1);
      while (bc_index[p] != kEmpty) p = (p + 1) & (cap - 1);
      bc_index[p] = (h & 0xFFFFFFFF00000000ull) | (uint64_t)i;
    }
  }
};

extern "C" dmx_store* dmx_store_new(void) {
  dmx_store* s = new (std::nothrow) dmx_store;
  if (!s) { set_error(DMX_ERR_NOMEM, "dmx_store_new: out of memory"); return nullptr; }
  for (dmx_store::Shard& sh : s->shard) sh.rehash(1 << 8);
  s->bc_rehash(1 << 10);
  return s;
}
extern "C" void dmx_store_free(dmx_store* s) { delete s; }
extern "C" int32_t dmx_store_add_snp(dmx_store* s) {
  if (!s) return set_error(DMX_ERR_ARG, "dmx_store_add_snp: null store");
  s->frozen = false;
  return s->n_snps++;
}
extern "C" int32_t dmx_store_add_cell(dmx_store* s, const char* barcode) {       // sc_drop_seq.cpp:20-32
  if (!s || !barcode) return set_error(DMX_ERR_ARG, "dmx_store_add_cell: null argument");
  size_t len;
  const uint64_t h = dmx_store::hash_str(barcode, &len);
  const size_t cap = s->bc_index.size();
  size_t p = h & (cap - 1);
  for (; s->bc_index[p] != dmx_store::kEmpty; p = (p + 1) & (cap - 1)) {
    if ((s->bc_index[p] ^ h) >> 32) continue;
    const std::string& b = s->barcodes[(size_t)(s->bc_index[p] & 0xFFFFFFFFull)];
    if (b.size() == len && std::memcmp(b.data(), barcode, len) == 0) return (int32_t)(s->bc_index[p] & 0xFFFFFFFFull);
  }
  if (s->barcodes.size() >= 0x7FFFFFF0ull) return set_error(DMX_ERR_ARG, "dmx_store_add_cell: too many barcodes");
  const int32_t id = (int32_t)s->barcodes.size();
  s->barcodes.emplace_back(barcode, len);
  s->bc_index[p] = (h & 0xFFFFFFFF00000000ull) | (uint64_t)id;
  if (s->barcodes.size() * 2 > cap) s->bc_rehash(cap * 2);
  s->totl.push_back(0); s->pass.push_back(0); s->uniq.push_back(0);
  s->n_cells_pub.store((int32_t)s->barcodes.size(), std::memory_order_release);
  s->frozen = false;
  return id;
}
extern "C" int dmx_store_count_read(dmx_store* s, int32_t cell) {                // cmd_cram_demuxlet.cpp:295
  if (!s || cell < 0 || cell >= (int32_t)s->barcodes.size()) return set_error(DMX_ERR_ARG, "dmx_store_count_read: bad cell %d", cell);
  ++s->totl[cell];
  return DMX_OK;
}
extern "C" int dmx_store_add_read(dmx_store* s, int32_t snp, int32_t cell, const char* umi, int32_t allele, int32_t bq) {
  if (!s || !umi) return set_error(DMX_ERR_ARG, "dmx_store_add_read: null argument");
  if (snp < 0 || snp >= s->n_snps) return set_error(DMX_ERR_ARG, "dmx_store_add_read: snp %d out of range", snp);
  if (cell < 0 || cell >= (int32_t)s->barcodes.size()) return set_error(DMX_ERR_ARG, "dmx_store_add_read: cell %d out of range", cell);
  if (allele < 0 || allele > 2) return set_error(DMX_ERR_ARG, "dmx_store_add_read: allele %d not in {0,1,2}", allele);
  if (bq < 0 || bq > 127) return set_error(DMX_ERR_ARG, "dmx_store_add_read: base quality %d not in [0,127]", bq);
  ++s->pass[cell];                                                                 // sc_drop_seq.cpp:39
  const int r = s->shard[cell & (dmx_store::kShards - 1)].add(cell, snp, umi, std::strlen(umi), allele, bq);
  if (r < 0) return set_error(DMX_ERR_ARG, "dmx_store_add_read: more than 2^32 unique observations, a UMI over 16 MiB or a UMI pool over 1 TiB");
  if (r) ++s->uniq[cell];                                                          // :75
  s->frozen = false;
  return r;
}

// n calls of dmx_store_add_read in the given order, with the observations of different cell shards inserted on different host
// threads (the order inside a shard — all that "first observation wins" can see — is the given one).
extern "C" int dmx_store_add_batch(dmx_store* s, int64_t n, const int32_t* snp, const int32_t* cell, const char* umi_pool,
                                   const uint64_t* umi_off, const uint32_t* umi_len, const uint8_t* allele, const uint8_t* bq,
                                   uint8_t* is_new, int32_t n_threads) {
  if (!s || n < 0 || (n && (!snp || !cell || !umi_pool || !umi_off || !umi_len || !allele || !bq)))
    return set_error(DMX_ERR_ARG, "dmx_store_add_batch: null argument");
  const int32_t B = s->n_cells_pub.load(std::memory_order_acquire);
  for (int64_t i = 0; i < n; ++i) {
    if (snp[i] < 0 || snp[i] >= s->n_snps) return set_error(DMX_ERR_ARG, "dmx_store_add_batch: snp %d out of range (item %lld)", snp[i], (long long)i);
    if (cell[i] < 0 || cell[i] >= B) return set_error(DMX_ERR_ARG, "dmx_store_add_batch: cell %d out of range (item %lld)", cell[i], (long long)i);
    if (allele[i] > 2 || bq[i] > 127) return set_error(DMX_ERR_ARG, "dmx_store_add_batch: allele %d / base quality %d out of range (item %lld)", allele[i], bq[i], (long long)i);
  }
  constexpr int K = dmx_store::kShards;
  // items of each shard, in the given order (counting sort by shard)
  std::vector<int64_t> first(K + 1, 0);
  for (int64_t i = 0; i < n; ++i) ++first[(cell[i] & (K - 1)) + 1];
  for (int k = 0; k < K; ++k) first[k + 1] += first[k];
  std::vector<uint32_t> item((size_t)n);
  if (n > 0xFFFFFFFFll) return set_error(DMX_ERR_ARG, "dmx_store_add_batch: more than 2^32 items in one batch");
  {
    std::vector<int64_t> at(first.begin(), first.end() - 1);
    for (int64_t i = 0; i < n; ++i) item[(size_t)at[cell[i] & (K - 1)]++] = (uint32_t)i;
  }
  std::atomic<int> next{0};
  std::atomic<bool> failed{false};
  auto work = [&]() {
    for (int k; (k = next.fetch_add(1)) < K;) {
      dmx_store::Shard& sh = s->shard[k];
      for (int64_t q = first[k]; q < first[k + 1]; ++q) {
        const uint32_t i = item[(size_t)q];
        ++s->pass[cell[i]];                                                        // a cell belongs to one shard: no two threads share a counter
        const int r = sh.add(cell[i], snp[i], umi_pool + umi_off[i], umi_len[i], allele[i], bq[i]);
        if (r < 0) { failed = true; return; }
        if (r) ++s->uniq[cell[i]];
        if (is_new) is_new[i] = (uint8_t)r;
      }
    }
  };
  const int T = std::max(1, std::min(n_threads > 0 ? n_threads : host_threads(), K));
  std::vector<std::thread> th;
  for (int t = 1; t < T; ++t) th.emplace_back(work);
  work();
  for (std::thread& t : th) t.join();
  s->frozen = false;
  if (failed) return set_error(DMX_ERR_ARG, "dmx_store_add_batch: more than 2^32 unique observations in a shard, a UMI over 16 MiB or a UMI pool over 1 TiB");
  return DMX_OK;
}
extern "C" int32_t dmx_store_n_cells(const dmx_store* s) { return s ? (int32_t)s->barcodes.size() : 0; }
extern "C" int32_t dmx_store_n_snps(const dmx_store* s) { return s ? s->n_snps.load() : 0; }
extern "C" const char* dmx_store_barcode(const dmx_store* s, int32_t cell) {
  if (!s || cell < 0 || cell >= (int32_t)s->barcodes.size()) return nullptr;
  return s->barcodes[cell].c_str();
}

extern "C" int dmx_store_freeze(dmx_store* s, dmx_pileup* out) {
  if (!s || !out) return set_error(DMX_ERR_ARG, "dmx_store_freeze: null argument");
  const int32_t B = (int32_t)s->barcodes.size();
  if (!s->frozen) {
    // The shards' logs stay where they are: a cell's observations are in ONE shard (cell & (kShards - 1)), in arrival order, with
    // their UMI bytes in that shard's pool, so that every pass below runs shard- or cell-parallel on the host's threads and
    // nothing is concatenated (the 2.9 s this took for 1.6e7 observations on one thread were a copy, a count and a scatter).
    constexpr int K = dmx_store::kShards;
    const bool timing = getenv("DMX_FREEZE_TIMING") != nullptr;
    auto tnow = [] { return std::chrono::steady_clock::now(); };
    auto t_prev = tnow();
    auto lap = [&](const char* what) { if (timing) { const auto t = tnow(); fprintf(stderr, "dmx_store_freeze: %s %.3f s\n", what, std::chrono::duration<double>(t - t_prev).count()); t_prev = t; } };
    size_t n = 0;
    {
      size_t np = 0;
      for (const dmx_store::Shard& sh : s->shard) { n += sh.obs.size(); np += sh.umi_pool.size(); }
      if (n > 0xFFFFFFFFull) return set_error(DMX_ERR_ARG, "dmx_store_freeze: more than 2^32 unique observations");
      if (np >= (1ull << 40)) return set_error(DMX_ERR_ARG, "dmx_store_freeze: UMI pool over 1 TiB");
    }
    auto over_shards = [&](auto&& fn) {
      std::atomic<int> next{0};
      auto work = [&]() { for (int k; (k = next.fetch_add(1)) < K;) fn(k); };
      const int T = std::max(1, std::min(host_threads(), K));
      std::vector<std::thread> th;
      for (int t = 1; t < T; ++t) th.emplace_back(work);
      work();
      for (std::thread& t : th) t.join();
    };
    // (1) stable counting sort by cell id, shard by shard: a cell's observations become one contiguous segment
    //     (still in arrival order — which for a coordinate-sorted BAM is already nearly SNP order) ...
    //     Shard k counts its cells k, k + K, k + 2K, ... in an array of its own (neighbouring cells belong to different shards).
    const size_t per = ((size_t)B + K - 1) / K + 1;
    std::vector<int64_t> cnt((size_t)K * per, 0);
    over_shards([&](int k) {
      int64_t* c = cnt.data() + (size_t)k * per;
      for (const dmx_store::Obs& o : s->shard[k].obs) ++c[(size_t)o.cell / K];
    });
    lap("count");
    std::vector<int64_t> seg((size_t)B + 1, 0);
    for (int32_t c = 0; c < B; ++c) seg[(size_t)c + 1] = seg[(size_t)c] + cnt[(size_t)(c % K) * per + (size_t)c / K];
    // ... as 16-byte records (SNP, allele | quality, UMI reference): the passes below then read a cell's segment front to back and
    // never go back to the shard's log, whose entries for one cell lie a cache miss apart
    struct Rec { int32_t snp; uint8_t allele, bq; uint16_t pad; uint64_t umi; };
    static_assert(sizeof(Rec) == 16, "");
    std::unique_ptr<Rec[]> recs(new Rec[n ? n : 1]);
    over_shards([&](int k) {
      int64_t* at = cnt.data() + (size_t)k * per;                      // (reused: where the next observation of cell k + K q goes)
      for (size_t q = 0; (int64_t)(q * K + k) < (int64_t)B; ++q) at[q] = seg[q * K + (size_t)k];
      for (const dmx_store::Obs& o : s->shard[k].obs) recs[(size_t)at[(size_t)o.cell / K]++] = Rec{o.snp, o.allele, o.bq, 0, o.umi};
    });
    lap("scatter");
    // (2) per cell, on all host threads: SNP id, then UMI as unsigned bytes with the shorter string first on a common prefix
    //     (== std::string::operator<, the order of the reference's std::map<std::string,uint32_t>); then the cell's pair and
    //     stored-read counts.  A coordinate-sorted BAM delivers a cell's SNPs in ascending order already: then only the runs of
    //     one SNP (its UMIs) are sorted.
    s->cell_pair_off.assign((size_t)B + 1, 0);
    s->cell_read_off.assign((size_t)B + 1, 0);
    const int nthreads = std::max(1, std::min(host_threads(), B));
    auto over_cells = [&](auto&& fn) {
      std::atomic<int32_t> next{0};
      auto work = [&]() { for (int32_t c0; (c0 = next.fetch_add(64)) < B;) for (int32_t c = c0; c < std::min(B, c0 + 64); ++c) fn(c); };
      std::vector<std::thread> pool_t;
      for (int t = 1; t < nthreads; ++t) pool_t.emplace_back(work);
      work();
      for (std::thread& t : pool_t) t.join();
    };
    over_cells([&](int32_t c) {
      const char* pool = s->shard[c % K].umi_pool.data();
      auto umi_less = [pool](const Rec& x, const Rec& y) {
        const uint64_t xo = x.umi & 0xFFFFFFFFFFull, yo = y.umi & 0xFFFFFFFFFFull;
        const uint32_t xl = (uint32_t)(x.umi >> 40), yl = (uint32_t)(y.umi >> 40);
        const int d = std::memcmp(pool + xo, pool + yo, std::min(xl, yl));
        if (d != 0) return d < 0;
        return xl < yl;
      };
      Rec* b0 = recs.get() + seg[(size_t)c];
      Rec* b1 = recs.get() + seg[(size_t)c + 1];
      bool ascending = true;
      for (Rec* q = b0; q + 1 < b1; ++q) if (q[1].snp < q[0].snp) { ascending = false; break; }
      if (!ascending) std::stable_sort(b0, b1, [](const Rec& x, const Rec& y) { return x.snp < y.snp; });
      int64_t np = 0, nr = 0;
      for (Rec* q = b0; q < b1;) {
        Rec* e = q + 1;
        while (e < b1 && e->snp == q->snp) ++e;
        if (e - q > 1 && !std::is_sorted(q, e, umi_less)) std::sort(q, e, umi_less);
        ++np;
        for (Rec* r = q; r < e; ++r) if (r->allele != 2) ++nr;    // allele 2 never enters a likelihood (cmd_cram_demuxlet.cpp:435,:604)
        q = e;
      }
      s->cell_pair_off[(size_t)c + 1] = np; s->cell_read_off[(size_t)c + 1] = nr;
    });
    lap("sort + count per cell");
    for (int32_t c = 0; c < B; ++c) { s->cell_pair_off[c + 1] += s->cell_pair_off[c]; s->cell_read_off[c + 1] += s->cell_read_off[c]; }
    // (3) fill the CSR, again per cell
    const size_t P = (size_t)s->cell_pair_off[(size_t)B], R = (size_t)s->cell_read_off[(size_t)B];
    s->pair_snp.assign(P, 0); s->reads.assign(R, 0);
    std::vector<uint32_t> nrd(P, 0);
    over_cells([&](int32_t c) {
      const Rec* b0 = recs.get() + seg[(size_t)c];
      const Rec* b1 = recs.get() + seg[(size_t)c + 1];
      int64_t p = s->cell_pair_off[(size_t)c] - 1, r = s->cell_read_off[(size_t)c];
      for (const Rec* q = b0; q < b1; ++q) {
        if (q == b0 || q[-1].snp != q->snp) s->pair_snp[(size_t)++p] = q->snp;
        if (q->allele != 2) { s->reads[(size_t)r++] = (uint8_t)((q->allele << 7) | q->bq); ++nrd[(size_t)p]; }
      }
    });
    lap("fill");
    uint32_t max_nrd = 0;
    for (size_t p = 0; p < P; ++p) max_nrd = std::max(max_nrd, nrd[p]);
    s->nrd_width = max_nrd <= 0xFF ? 1 : (max_nrd <= 0xFFFF ? 2 : 4);
    s->pair_nrd_bytes.assign(nrd.size() * (size_t)s->nrd_width + 4, 0);
    for (size_t p = 0; p < nrd.size(); ++p) std::memcpy(&s->pair_nrd_bytes[p * (size_t)s->nrd_width], &nrd[p], (size_t)s->nrd_width); // little endian
    s->frozen = true;
    lap("count widths");
  }
  std::memset(out, 0, sizeof *out);
  out->n_cells = B; out->n_snps = s->n_snps;
  out->n_pairs = (int64_t)s->pair_snp.size(); out->n_reads = (int64_t)s->reads.size();
  out->cell_pair_off = s->cell_pair_off.data(); out->cell_read_off = s->cell_read_off.data();
  static const int32_t kNoPairs[1] = {0};       // a store without any pair still hands out the sparse layout (NULL = dense, dmx.h)
  out->pair_snp = s->pair_snp.empty() ? kNoPairs : s->pair_snp.data(); out->pair_nrd = s->pair_nrd_bytes.data(); out->nrd_width = s->nrd_width;
  out->memory = DMX_MEM_HOST; out->reads = s->reads.data();
  s->totl.copy_to(s->f_totl); s->pass.copy_to(s->f_pass); s->uniq.copy_to(s->f_uniq);     // (the counters move with every add_*: copied at every freeze)
  out->rd_totl = s->f_totl.data(); out->rd_pass = s->f_pass.data(); out->rd_uniq = s->f_uniq.data();
  return DMX_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// tie arbiter: exact host re-evaluation of a few grid entries of one cell (cmd_cram_demuxlet.cpp:595-684)
namespace dmx {

static inline uint32_t nrd_at(const dmx_pileup& pl, int64_t p) {
  const uint8_t* b = (const uint8_t*)pl.pair_nrd + (size_t)p * (size_t)pl.nrd_width;
  uint32_t v = 0;
  std::memcpy(&v, b, (size_t)pl.nrd_width);
  return v;
}

// pG[A][3][3] of one pair from its stored read bytes: cmd_cram_demuxlet.cpp:597-663, operation for operation
static inline void mix_pair(const uint8_t* rd, uint32_t nr, const ReadLut& lut, const double* mixR, const double* mixA, size_t n9, double* pG) {
  for (size_t q = 0; q < n9; ++q) pG[q] = 1.0;                                                                          // :597
  for (uint32_t r = 0; r < nr; ++r) {
    const uint8_t b = rd[r];
    const int al = b >> 7, bq = b & 127;
    const double pR = (al == 0) ? lut.mat[bq] : lut.e3[bq];      // :606
    const double pA = (al == 1) ? lut.mat[bq] : lut.e3[bq];      // :607
    double mx = 0;
    for (size_t q = 0; q < n9; ++q) { pG[q] *= (pR * mixR[q] + pA * mixA[q]); if (mx < pG[q]) mx = pG[q]; }             // :625-627
    for (size_t q = 0; q < n9; ++q) pG[q] /= mx;                                                                        // :632-639
  }
  double mx = 0;
  for (size_t q = 0; q < n9; ++q) { pG[q] += 1e-6; if (mx < pG[q]) mx = pG[q]; }                                        // :643-654
  for (size_t q = 0; q < n9; ++q) pG[q] /= mx;                                                                          // :656-663
}

static void mix_weights(int32_t A, const double* alpha, std::vector<double>& mixR, std::vector<double>& mixA) {
  mixR.assign((size_t)A * 9, 0.0); mixA.assign((size_t)A * 9, 0.0);
  for (int32_t n = 0; n < A; ++n)
    for (int l = 0; l < 3; ++l)
      for (int m = 0; m < 3; ++m) {
        const double p = 0.5 * l + (m - l) * 0.5 * alpha[n];     // :613 expected ALT fraction
        mixA[(size_t)n * 9 + l * 3 + m] = p;
        mixR[(size_t)n * 9 + l * 3 + m] = 1.0 - p;
      }
}

void build_mix_tables(const ReadLut& lut, int32_t A, const double* alpha, MixTables* out) {
  std::vector<double> mixR, mixA;
  mix_weights(A, alpha, mixR, mixA);
  const size_t n9 = (size_t)A * 9;
  out->A = A;
  out->none.assign(n9, 0.0);
  mix_pair(nullptr, 0, lut, mixR.data(), mixA.data(), n9, out->none.data());
  out->one.assign(256 * n9, 0.0);
  for (int b = 0; b < 256; ++b) { const uint8_t rd[1] = {(uint8_t)b}; mix_pair(rd, 1, lut, mixR.data(), mixA.data(), n9, &out->one[(size_t)b * n9]); }
  out->two.assign((size_t)128 * 128 * n9, 0.0);
  for (int c0 = 0; c0 < 128; ++c0)
    for (int c1 = 0; c1 < 128; ++c1) {
      const uint8_t rd[2] = {(uint8_t)(((c0 & 64) << 1) | (c0 & 63)), (uint8_t)(((c1 & 64) << 1) | (c1 & 63))};
      mix_pair(rd, 2, lut, mixR.data(), mixA.data(), n9, &out->two[((size_t)c0 * 128 + c1) * n9]);
    }
}

void exact_grid_entries(const dmx_pileup& pl, const float* g, int32_t V, int32_t A, const double* alpha,
                        const ReadLut& lut, const MixTables* mix, int32_t cell, std::vector<GridReq>& reqs) {
  for (GridReq& r : reqs) r.value = 0.0;
  const size_t n9 = (size_t)A * 9;
  std::vector<double> pG(n9), mixR, mixA;
  mix_weights(A, alpha, mixR, mixA);
  if (mix && mix->A != A) mix = nullptr;
  int64_t rd = pl.cell_read_off[cell];
  const int64_t p0 = pl.cell_pair_off[cell], p1 = pl.cell_pair_off[cell + 1];
  for (int64_t p = p0; p < p1; ++p) {
    const int32_t snp = pl.pair_snp ? pl.pair_snp[p] : (int32_t)(p - p0);
    const uint32_t nr = nrd_at(pl, p);
    const uint8_t* rb = pl.reads + rd;
    const double* P;
    if (mix && nr == 0) P = mix->none.data();
    else if (mix && nr == 1) P = &mix->one[(size_t)rb[0] * n9];
    else if (mix && nr == 2 && !((rb[0] | rb[1]) & 0x40))
      P = &mix->two[((size_t)(((rb[0] & 0x80) >> 1) | (rb[0] & 0x3F)) * 128 + (((rb[1] & 0x80) >> 1) | (rb[1] & 0x3F))) * n9];
    else { mix_pair(rb, nr, lut, mixR.data(), mixA.data(), n9, pG.data()); P = pG.data(); }
    rd += nr;
    const float* gs = g + (size_t)snp * V * 3;
    for (GridReq& r : reqs) {
      const float* gj = gs + 3 * r.j;
      const float* gk = gs + 3 * r.k;
      const double* Pn = P + (size_t)r.n * 9;
      double sum = 0;
      for (int l = 0; l < 3; ++l)
        for (int m = 0; m < 3; ++m) sum += ((double)gj[l] * (double)gk[m]) * Pn[l * 3 + m];                             // :553,:677-679
      r.value += std::log(sum);                                                                                         // :683
    }
  }
}

}  // namespace dmx

// ---------------------------------------------------------------------------------------------------------------------
// a6, a10..a14  finaliser
namespace {

struct CellCall {
  double max_llk, sum_single, sum_double;
  int32_t i_sing1, i_sing2, j_best, k_best, n_best;
};

// One cell's grid -> the scalars every row of .sing2/.pair/.best is built from (cmd_cram_demuxlet.cpp:713-734,746-758,799-814)
CellCall call_cell(const double* grid, int32_t V, int32_t A, const double* alpha, double prior) {
  CellCall c;
  const size_t n = (size_t)V * V * A;
  c.max_llk = -1e300;
  for (size_t q = 0; q < n; ++q) if (c.max_llk < grid[q]) c.max_llk = grid[q];
  c.sum_single = 0; c.sum_double = 0;
  for (int32_t j = 0; j < V; ++j) {
    c.sum_single += (std::exp(grid[(size_t)j * V * A] - c.max_llk) * (1. - prior) / V);
    for (int32_t k = 0; k < V; ++k) {
      if (j == k) continue;
      for (int32_t a = 1; a < A; ++a)
        c.sum_double += (std::exp(grid[((size_t)j * V + k) * A + a] - c.max_llk) * prior / V / (V - 1) / (A - 1) / (alpha[a] == 0.5 ? 2.0 : 1.0));
    }
  }
  c.i_sing1 = c.i_sing2 = -1;
  double m1 = -1e300, m2 = -1e300;
  for (int32_t j = 0; j < V; ++j) {
    const double v = grid[(size_t)j * V * A];
    if (m1 < v) { m2 = m1; c.i_sing2 = c.i_sing1; c.i_sing1 = j; m1 = v; }
    else if (m2 < v) { c.i_sing2 = j; m2 = v; }
  }
  c.j_best = c.k_best = c.n_best = -1;
  double mab = -1e300;
  for (int32_t j = 0; j < V; ++j)
    for (int32_t k = 0; k < V; ++k) {
      if (j == k) continue;
      for (int32_t a = 1; a < A; ++a) {
        const double v = grid[((size_t)j * V + k) * A + a];
        if (mab < v) { c.j_best = j; c.k_best = k; c.n_best = a; mab = v; }
      }
    }
  return c;
}

std::vector<int32_t> barcode_order(const dmx_final_input* in) {
  std::vector<int32_t> ord((size_t)in->n_cells);
  std::iota(ord.begin(), ord.end(), 0);
  std::sort(ord.begin(), ord.end(), [&](int32_t a, int32_t b) { return std::strcmp(in->barcodes[a], in->barcodes[b]) < 0; });
  return ord;
}

bool cell_filtered(const dmx_final_input* in, int32_t c) {      // :480,:581
  return (in->rd_totl[c] < in->min_total) || (in->rd_uniq[c] < in->min_uniq) || (in->n_snp[c] < in->min_snp);
}

int check_common(const dmx_final_input* in, const char* who) {
  if (!in) return set_error(DMX_ERR_ARG, "%s: null input", who);
  if (in->n_cells < 0 || in->n_samples < 1) return set_error(DMX_ERR_ARG, "%s: bad sizes", who);
  if (!in->barcodes || !in->sample_ids || !in->rd_totl || !in->rd_pass || !in->rd_uniq || !in->n_snp)
    return set_error(DMX_ERR_ARG, "%s: missing per-cell arrays", who);
  return DMX_OK;
}

struct File {
  FILE* f = nullptr;
  ~File() { if (f) fclose(f); }
  bool open(const std::string& path, bool append = false) { f = fopen(path.c_str(), append ? "a" : "w"); return f != nullptr; }
};

// ---- rows are formatted by several host threads and written in barcode order ----------------------------------------
// The reference prints every row with hprintf (= vsnprintf of glibc, hts_utils.cpp:1013-1034); the same conversions are
// used here (so the bytes are the same), into per-chunk buffers.  `--write-pair` at cfg4 is 2.1e8 rows / 10 GB of text
// (SURVEY 8a-a12): at ~3 us per row one thread would need ten minutes for what the GPUs compute in seconds.
void appendf(std::string& out, const char* fmt, ...) {
  char buf[768];
  va_list ap;
  va_start(ap, fmt);
  const int n = vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (n < 0) return;
  if ((size_t)n < sizeof buf) { out.append(buf, (size_t)n); return; }
  std::vector<char> big((size_t)n + 1);
  va_start(ap, fmt);
  vsnprintf(big.data(), big.size(), fmt, ap);
  va_end(ap);
  out.append(big.data(), (size_t)n);
}

// printf("%.<prec>lf") without printf: the decimal expansion of a binary64 is finite, so rounding m*2^e*10^prec to an
// integer in exact 128-bit arithmetic (ties to even, the default rounding mode glibc's printf honours) gives the same
// digits.  Anything outside the comfortable range goes to vsnprintf.  tests/test_host_units.py compares the two on
// millions of values.
void put_fixed(std::string& out, double v, int prec) {
  static const uint64_t kPow10[7] = {1ull, 10ull, 100ull, 1000ull, 10000ull, 100000ull, 1000000ull};
  uint64_t bits;
  std::memcpy(&bits, &v, sizeof bits);
  const uint64_t frac = bits & 0xFFFFFFFFFFFFFull;
  const int ex = (int)((bits >> 52) & 0x7FF);
  if (prec < 0 || prec > 6 || ex == 0x7FF || ex >= 1023 + 43) { appendf(out, "%.*lf", prec, v); return; }   // nan, inf, |v| >= 2^43
  const uint64_t m = ex ? (frac | (1ull << 52)) : frac;
  const int e = (ex ? ex : 1) - 1075;                                      // |v| = m * 2^e
  unsigned __int128 N = (unsigned __int128)m * kPow10[prec];               // < 2^73
  uint64_t q;
  if (e >= 0) q = (uint64_t)(N << e);                                      // unreachable for |v| < 2^43 with m >= 2^52, kept for clarity
  else {
    const int sh = -e;
    if (sh >= 127) q = 0;                                                  // N < 2^73 is far below half of 2^127
    else {
      const unsigned __int128 one = 1;
      const unsigned __int128 rem = N & ((one << sh) - 1), half = one << (sh - 1);
      q = (uint64_t)(N >> sh);
      if (rem > half || (rem == half && (q & 1))) ++q;
    }
  }
  char buf[40];
  int n = 0;
  const uint64_t ip = q / kPow10[prec];
  uint64_t fp = q % kPow10[prec];
  for (int i = 0; i < prec; ++i) { buf[n++] = (char)('0' + fp % 10); fp /= 10; }
  if (prec > 0) buf[n++] = '.';
  uint64_t t = ip;
  do { buf[n++] = (char)('0' + t % 10); t /= 10; } while (t);
  if (bits >> 63) buf[n++] = '-';
  std::reverse(buf, buf + n);
  out.append(buf, (size_t)n);
}

void put_int(std::string& out, int32_t v) {
  char buf[16];
  int n = 0;
  uint32_t u = v < 0 ? 0u - (uint32_t)v : (uint32_t)v;
  do { buf[n++] = (char)('0' + u % 10); u /= 10; } while (u);
  if (v < 0) buf[n++] = '-';
  std::reverse(buf, buf + n);
  out.append(buf, (size_t)n);
}

// printf("%.<prec>lg"): zero (every underflowed posterior) is "0"; the rest is left to vsnprintf.
void put_general(std::string& out, double v, int prec) {
  if (v == 0.0 && !std::signbit(v)) { out.push_back('0'); return; }
  appendf(out, "%.*lg", prec, v);
}
void put_general_impl(std::string& out, double v, int prec) { put_general(out, v, prec); }

// Host threads for the formatters: DMX_THREADS, else the smaller of the visible CPUs and the container's CPU quota.
int host_threads() {
  if (const char* e = getenv("DMX_THREADS")) { const int n = atoi(e); if (n >= 1) return std::min(n, 256); }
  int n = (int)std::thread::hardware_concurrency();
  if (n < 1) n = 1;
  std::ifstream q("/sys/fs/cgroup/cpu.max");
  std::string quota; long long period = 0;
  if (q >> quota >> period && quota != "max" && period > 0) n = std::max(1, std::min(n, (int)(atoll(quota.c_str()) / period)));
  return std::min(n, 64);
}

constexpr int kOutFiles = 3;
struct Chunk { std::string out[kOutFiles]; };

// fn(first, last, chunk) formats items [first, last) into chunk.out[i]; chunks reach files[i] in item order.
template <class Fn>
int format_in_order(size_t n_items, size_t per_chunk, FILE* const files[kOutFiles], Fn&& fn) {
  per_chunk = std::max<size_t>(per_chunk, 1);
  const size_t n_chunks = (n_items + per_chunk - 1) / per_chunk;
  const int n_threads = (int)std::min<size_t>((size_t)host_threads(), n_chunks);
  bool io_ok = true;
  auto flush = [&](Chunk& ck) {
    for (int i = 0; i < kOutFiles; ++i) {
      if (files[i] && !ck.out[i].empty() && fwrite(ck.out[i].data(), 1, ck.out[i].size(), files[i]) != ck.out[i].size()) io_ok = false;
      ck.out[i].clear();
    }
  };
  if (n_threads <= 1) {
    Chunk ck;
    for (size_t k = 0; k < n_chunks; ++k) { fn(k * per_chunk, std::min(n_items, (k + 1) * per_chunk), ck); flush(ck); }
    return io_ok ? DMX_OK : set_error(DMX_ERR_IO, "write failed");
  }
  const size_t W = (size_t)n_threads * 2;                 // chunks in flight (bounds the buffered text)
  std::vector<Chunk> slots(W);
  std::vector<char> ready(W, 0);
  std::mutex mu;
  std::condition_variable cv_ready, cv_free;
  size_t next = 0, written = 0;
  auto worker = [&]() {
    for (;;) {
      size_t k;
      {
        std::unique_lock<std::mutex> lk(mu);
        k = next++;
        if (k >= n_chunks) return;
        cv_free.wait(lk, [&] { return k < written + W; });      // slot k % W was chunk k - W's: wait until that is on disk
      }
      fn(k * per_chunk, std::min(n_items, (k + 1) * per_chunk), slots[k % W]);
      { std::lock_guard<std::mutex> lk(mu); ready[k % W] = 1; }
      cv_ready.notify_all();
    }
  };
  std::vector<std::thread> pool;
  for (int t = 0; t < n_threads; ++t) pool.emplace_back(worker);
  for (size_t k = 0; k < n_chunks; ++k) {
    { std::unique_lock<std::mutex> lk(mu); cv_ready.wait(lk, [&] { return ready[k % W] != 0; }); }
    flush(slots[k % W]);
    { std::lock_guard<std::mutex> lk(mu); ready[k % W] = 0; ++written; }
    cv_free.notify_all();
  }
  for (std::thread& t : pool) t.join();
  return io_ok ? DMX_OK : set_error(DMX_ERR_IO, "write failed");
}

// cells that produce rows, in output order
std::vector<int32_t> output_cells(const dmx_final_input* in, bool need_snps) {
  std::vector<int32_t> out;
  for (int32_t c : barcode_order(in)) {
    if (cell_filtered(in, c)) continue;
    if (need_snps && in->n_snp[c] == 0) continue;                          // :592 no covered SNP: no rows
    out.push_back(c);
  }
  return out;
}

std::vector<int32_t> output_cells_impl(const dmx_final_input* in, bool need_snps) { return output_cells(in, need_snps); }

}  // namespace

std::vector<int32_t> dmx::output_cells(const dmx_final_input* in, bool need_snps) { return output_cells_impl(in, need_snps); }
void dmx::put_general(std::string& out, double v, int prec) { put_general_impl(out, v, prec); }

extern "C" int dmx_write_single(const dmx_final_input* in, const char* path) { return dmx::write_single_impl(in, path, false); }

int dmx::write_single_impl(const dmx_final_input* in, const char* path, bool append) {
  if (int rc = check_common(in, "dmx_write_single")) return rc;
  if (!path || !in->llks || !in->llk0s) return set_error(DMX_ERR_ARG, "dmx_write_single: null llks/llk0s/path");
  File w;
  if (!w.open(path, append)) return set_error(DMX_ERR_IO, "Cannot create %s file", path);
  const int32_t V = in->n_samples;
  if (!append) fputs("BARCODE\tSM_ID\tRD.TOTL\tRD.PASS\tRD.UNIQ\tN.SNP\tLLK1\tLLK0\tPOSTPRB\n", w.f);    // :470
  const std::vector<int32_t> cells = output_cells(in, false);
  FILE* const files[kOutFiles] = {w.f, nullptr, nullptr};
  return format_in_order(cells.size(), std::max<size_t>(1, 8192 / (size_t)V), files, [&](size_t first, size_t last, Chunk& ck) {
    for (size_t q = first; q < last; ++q) {
      const int32_t c = cells[q];
      const double* row = in->llks + (size_t)c * V;
      double lse = -1e300;                          // running log-sum-exp exactly as :484-489
      for (int32_t j = 0; j < V; ++j) {
        const double cur = row[j];
        lse = (lse > cur) ? lse + std::log(1.0 + std::exp(cur - lse)) : cur + std::log(1.0 + std::exp(lse - cur));
      }
      std::string head, mid;                        // "%s\t%s\t%d\t%d\t%d\t%d\t%.5lf\t%.5lf\t%.3lg\n" (:506-516), constant parts once
      head.append(in->barcodes[c]).push_back('\t');
      mid.push_back('\t'); put_int(mid, in->rd_totl[c]); mid.push_back('\t'); put_int(mid, in->rd_pass[c]); mid.push_back('\t');
      put_int(mid, in->rd_uniq[c]); mid.push_back('\t'); put_int(mid, in->n_snp[c]); mid.push_back('\t');
      std::string l0; put_fixed(l0, in->llk0s[c], 5);
      std::string& o = ck.out[0];
      for (int32_t j = 0; j < V; ++j) {
        o.append(head).append(in->sample_ids[j]).append(mid);
        put_fixed(o, row[j], 5); o.push_back('\t'); o.append(l0); o.push_back('\t');
        put_general(o, std::exp(row[j] - lse), 3); o.push_back('\n');
      }
    }
  });
}

extern "C" int dmx_write_doublet(const dmx_final_input* in, const char* out_prefix) { return dmx::write_doublet_impl(in, out_prefix, false); }

int dmx::write_doublet_impl(const dmx_final_input* in, const char* out_prefix, bool append) {
  if (int rc = check_common(in, "dmx_write_doublet")) return rc;
  if (!out_prefix || !in->llksAB || !in->llks00 || !in->alpha) return set_error(DMX_ERR_ARG, "dmx_write_doublet: null grid/alpha/prefix");
  dmx::DoubletSource src{};
  src.grid_all = in->llksAB;
  return dmx::write_doublet_core(in, src, out_prefix, append, "dmx_write_doublet");
}

// .sing2 and .best from the per-cell records of the device reduction (K3) — the multi-GPU path gathers exactly these.
namespace {
int write_summary_impl(const dmx_final_input* in, const double* sing, const dmx_cell_summary* summary, const double* const* cell_grid,
                       const char* out_prefix, const char* who) {
  if (int rc = check_common(in, who)) return rc;
  if (!out_prefix || !sing || !summary || !in->llks00 || !in->alpha) return set_error(DMX_ERR_ARG, "%s: null sing/summary/llks00/alpha/prefix", who);
  if (in->write_pair) return set_error(DMX_ERR_ARG, "%s: .pair rows need the full grid (use dmx_write_doublet)", who);
  dmx::DoubletSource src{};
  src.sing = sing; src.summary = summary; src.cell_grid = cell_grid;
  return dmx::write_doublet_core(in, src, out_prefix, false, who);
}
}  // namespace

extern "C" int dmx_write_doublet_summary(const dmx_final_input* in, const double* sing, const dmx_cell_summary* summary, const char* out_prefix) {
  return write_summary_impl(in, sing, summary, nullptr, out_prefix, "dmx_write_doublet_summary");
}
// (ABI 7) the grids of the near-tie-flagged barcodes as an argument: dmx_final_input is passed by pointer without a size member and does not grow
extern "C" int dmx_write_doublet_summary_grids(const dmx_final_input* in, const double* sing, const dmx_cell_summary* summary,
                                               const double* const* cell_grid, const char* out_prefix) {
  return write_summary_impl(in, sing, summary, cell_grid, out_prefix, "dmx_write_doublet_summary_grids");
}

// The doublet-stage writers (.sing2, .best, optionally .pair).  A cell's rows come either from its grid — the whole
// llksAB array (src.grid_all) or a per-cell pointer (src.cell_grid[c], the cells K3 flagged as near-ties) — or, without one,
// from its K3 record (src.sing / src.summary).
int dmx::write_doublet_core(const dmx_final_input* in, const DoubletSource& src, const char* out_prefix, bool append, const char* who) {
  const int32_t V = in->n_samples, A = in->n_alpha;
  if (V < 2 || A < 2) return set_error(DMX_ERR_ARG, "%s: needs >= 2 samples and >= 2 alphas (got %d, %d)", who, V, A);
  if (in->write_pair && !src.grid_all && !src.pair_rows) return set_error(DMX_ERR_ARG, "%s: .pair rows need the full grid", who);
  const bool pair_here = in->write_pair && !src.pair_rows;       // this function writes the .pair file (else: rows formatted on the device, DoubletSource)
  const double tol = in->tie_tol > 0 ? in->tie_tol : 1e-7;
  const bool arbiter = in->tie_pileup && in->tie_g;
  if (arbiter && in->tie_pileup->memory != DMX_MEM_HOST) return set_error(DMX_ERR_ARG, "%s: the tie arbiter needs a HOST pileup", who);
  const std::vector<int32_t> cells = output_cells(in, true);
  // ---- before a byte is written: every barcode the arbiter will walk has its pileup staged, and the records-only fall-back — a
  // near-tie-flagged barcode without a grid has its WHOLE grid re-evaluated on the host — stays a fall-back (pairs x V x V x A host logs)
  if (arbiter && src.summary) {
    double fallback_logs = 0;
    int64_t n_fallback = 0;
    for (int32_t c : cells) {
      dmx_cell_summary sm = src.summary[c];
      if (sm.flags & DMX_CELL_ORDER_RESOLVABLE) (void)dmx::resolve_tie_order(&sm);
      const int need = dmx::cell_needs(sm, in->alpha, A, true);
      if ((need & dmx::kNeedPileup) && src.tie_cell && src.tie_cell[c] < 0)
        return set_error(DMX_ERR_STATE, "%s: the tie arbiter needs barcode %s, whose pileup was not staged (%s)", who, in->barcodes[c],
                         append ? "no row of this range was written; the rows of earlier ranges are in the files" : "nothing was written");
      if ((need & dmx::kNeedGrid) && !src.grid_all && !(src.cell_grid && src.cell_grid[c])) {
        fallback_logs += (double)sm.n_pairs * (double)V * V * A;
        ++n_fallback;
      }
    }
    if (fallback_logs > 2e9)
      return set_error(DMX_ERR_ARG, "%s: %lld near-tie-flagged barcodes came without their grids; re-evaluating them on the host would take %.3g log() calls "
                       "— pass the grids (dmx_engine_get_cell_grids, dmx_write_doublet_summary_grids; a caller built against ABI 6 set dmx_final_input.cell_grid "
                       "for this: that member was withdrawn in ABI 7 and is no longer read)", who, (long long)n_fallback, fallback_logs);
  }
  const std::string pre(out_prefix);
  File sing2, pairf, best;
  if (!sing2.open(pre + ".sing2", append) || !best.open(pre + ".best", append) || (pair_here && !pairf.open(pre + ".pair", append)))
    return set_error(DMX_ERR_IO, "Cannot create %s.single, %s.pair files", out_prefix, out_prefix);     // :535-536
  if (!append) fputs("BARCODE\tSM_ID\tRD.TOTL\tRD.PASS\tRD.UNIQ\tN.SNP\tLLK1\tLLK0\tPOSTPRB\n", sing2.f);             // :533
  if (pairf.f && !append) fputs("BARCODE\tSM1.ID\tSM2.ID\tLLK12\tPOSTPRB\n", pairf.f);                              // :570 (5 names for 6 fields: reference quirk)
  if (!append) fputs("BARCODE\tRD.TOTL\tRD.PASS\tRD.UNIQ\tN.SNP\tBEST\tSNG.1ST\tSNG.LLK1\tSNG.2ND\tSNG.LLK2\tSNG.LLK0\tDBL.1ST\tDBL.2ND\tALPHA\tLLK12\tLLK1\tLLK2\tLLK10\tLLK20\tLLK00\tPRB.DBL\tPRB.SNG1\n", best.f);  // :571

  const size_t ng = (size_t)V * V * A;
  const double prior = in->doublet_prior;
  dmx::ReadLut lut;
  std::shared_ptr<const dmx::MixTables> mix;
  if (arbiter) {
    double mat[256], err[256];
    dmx_phred_tables(mat, err);
    dmx::build_read_lut(mat, err, &lut);
    // 128*128*A*9 doubles (75 MB at A = 64): built once per alpha grid, not once per appended range of a job
    static std::mutex mix_mu;
    static std::vector<double> mix_alpha;
    static std::shared_ptr<const dmx::MixTables> mix_cached;
    std::lock_guard<std::mutex> lk(mix_mu);
    if (!mix_cached || mix_alpha.size() != (size_t)A || !std::equal(mix_alpha.begin(), mix_alpha.end(), in->alpha,
                                                                       [](double a, double b) { return std::memcmp(&a, &b, sizeof a) == 0; })) {
      std::shared_ptr<dmx::MixTables> m(new dmx::MixTables);
      dmx::build_mix_tables(lut, A, in->alpha, m.get());
      mix_cached = m;
      mix_alpha.assign(in->alpha, in->alpha + A);
    }
    mix = mix_cached;
  }
  FILE* const files[kOutFiles] = {sing2.f, pairf.f, best.f};
  std::vector<std::string> alpha_txt((size_t)A);                           // "\t%.3lf\t" of every alpha
  for (int32_t a = 0; a < A; ++a) { alpha_txt[(size_t)a].push_back('\t'); put_fixed(alpha_txt[(size_t)a], in->alpha[a], 3); alpha_txt[(size_t)a].push_back('\t'); }
  const size_t rows_per_cell = (size_t)V + (pairf.f ? (size_t)V * V * (A - 1) : 0);
  // the arbiter walks a cell's whole pileup: keep chunks small enough that every host thread gets work
  const size_t per_chunk = arbiter ? std::max<size_t>(1, std::min<size_t>(16384 / rows_per_cell, (cells.size() + 4 * (size_t)host_threads() - 1) / (4 * (size_t)host_threads())))
                                   : std::max<size_t>(1, 16384 / rows_per_cell);
  std::atomic<int32_t> tie_missing{-1};           // a barcode the arbiter needed although src.tie_cell says its pileup is not staged
  const int frc = format_in_order(cells.size(), per_chunk, files, [&](size_t first, size_t last, Chunk& ck) {
  std::vector<double> scratch, sg_fix;
  std::vector<dmx::GridReq> reqs;
  for (size_t q = first; q < last; ++q) {
    const int32_t c = cells[q];
    const int32_t tc = src.tie_cell ? src.tie_cell[c] : c;      // the cell's index in in->tie_pileup
    // (tc < 0: the caller did not stage this barcode's pileup — legitimate only if the arbiter has nothing to do for it; checked below)
    const double* grid = src.grid_all ? src.grid_all + (size_t)c * ng : (src.cell_grid ? src.cell_grid[c] : nullptr);
    const double* l00 = in->llks00 + (size_t)c * A;
    const char* bc = in->barcodes[c];
    const int32_t t = in->rd_totl[c], p = in->rd_pass[c], u = in->rd_uniq[c], ns = in->n_snp[c];
    std::string head, mid, l0s;                                            // constant parts of this cell's rows, once
    head.append(bc).push_back('\t');
    mid.push_back('\t'); put_int(mid, t); mid.push_back('\t'); put_int(mid, p); mid.push_back('\t'); put_int(mid, u);
    mid.push_back('\t'); put_int(mid, ns); mid.push_back('\t');
    put_fixed(l0s, l00[0], 4);

    // ---- the scalars every row is built from: from the grid with the reference's scans, or from the K3 record
    double max_llk, sum_single, sum_double, sing1, sing2v, l12, l1, l2, l10, l20;
    int32_t i_sing1, i_sing2, jb, kb, nb;
    const double* sg = nullptr;                                            // singlet column when there is no grid
    dmx_cell_summary resolved;
    const dmx_cell_summary* smp = src.summary ? &src.summary[c] : nullptr;
    if (smp && (smp->flags & DMX_CELL_ORDER_RESOLVABLE)) {                 // K3b left one log() per accumulator to the host's libm:
                                                                           // no pileup needed, so with or without the arbiter (dmx.h)
      resolved = *smp;
      if (dmx::resolve_tie_order(&resolved)) smp = &resolved;
    }
    constexpr int32_t kNear = DMX_CELL_NEAR_DOUBLET | DMX_CELL_NEAR_SINGLET;
    bool host_grid = false;
    if (!grid && smp && (smp->flags & kNear) && smp->n_pairs > 0 && arbiter && tc < 0) tie_missing = c;
    else if (!grid && smp && (smp->flags & kNear) && smp->n_pairs > 0 && arbiter) {
      // A near-tie beyond the (j,k)/(k,j) mirror — another sample pair (duplicate samples), another alpha, a third singlet — and
      // nothing but the record: which candidates sit within tol is not in the record, so the barcode's whole grid is evaluated
      // here, in the reference's operation order with the host libm (:595-684).  Flagged barcodes are the shallow ones (a handful
      // of covered SNPs) unless the panel itself is degenerate; a caller with the engine at hand passes the grids instead
      // (dmx_final_input.cell_grid, dmx_engine_get_cell_grids).
      reqs.clear();
      for (int32_t j = 0; j < V; ++j) for (int32_t k = 0; k < V; ++k) for (int32_t a = 0; a < A; ++a) reqs.push_back({j, k, a, 0.0});
      dmx::exact_grid_entries(*in->tie_pileup, in->tie_g, V, A, in->alpha, lut, mix.get(), tc, reqs);
      scratch.resize(ng);
      for (size_t q = 0; q < ng; ++q) scratch[q] = reqs[q].value;
      grid = scratch.data();
      host_grid = true;
    }
    if (grid) {
      if (host_grid) {
        // every entry is already the reference's
      } else if (smp && (smp->flags & DMX_CELL_ORDER_CERTIFIED) && !(smp->flags & kNear)) {
        // the device certified both accumulators of the best alpha = 0.5 pair (K3b) and K3 saw no other near-tie: the two
        // entries the arbiter would re-evaluate are known, bit for bit
        const dmx_cell_summary& sm = *smp;
        const int32_t a = std::min(sm.j_best, sm.k_best), b = std::max(sm.j_best, sm.k_best);
        scratch.assign(grid, grid + ng);
        scratch[((size_t)a * V + b) * A + sm.n_best] = sm.llk_ab;
        scratch[((size_t)b * V + a) * A + sm.n_best] = sm.llk_ba;
        grid = scratch.data();
      } else if (arbiter) {
        // Which entries sit within tol of a decision?  top-2 singlets (:746-758) and the best doublet (:799-814).
        reqs.clear();
        double s1 = -1e300, s2 = -1e300;
        for (int32_t j = 0; j < V; ++j) { const double v = grid[(size_t)j * V * A]; if (v > s1) { s2 = s1; s1 = v; } else if (v > s2) s2 = v; }
        int near2 = 0;
        for (int32_t j = 0; j < V; ++j) if (grid[(size_t)j * V * A] >= s2 - tol) ++near2;
        if (near2 > 2 || s1 - s2 < tol)
          for (int32_t j = 0; j < V; ++j) if (grid[(size_t)j * V * A] >= s2 - tol) reqs.push_back({j, 0, 0, 0.0});
        double mab = -1e300;
        for (int32_t j = 0; j < V; ++j) for (int32_t k = 0; k < V; ++k) if (j != k) for (int32_t a = 1; a < A; ++a) mab = std::max(mab, grid[((size_t)j * V + k) * A + a]);
        size_t nd0 = reqs.size();
        for (int32_t j = 0; j < V; ++j) for (int32_t k = 0; k < V; ++k) if (j != k) for (int32_t a = 1; a < A; ++a)
          if (grid[((size_t)j * V + k) * A + a] >= mab - tol) reqs.push_back({j, k, a, 0.0});
        if (reqs.size() - nd0 == 1) reqs.pop_back();
        if (!reqs.empty() && tc < 0) tie_missing = c;
        else if (!reqs.empty()) {
          dmx::exact_grid_entries(*in->tie_pileup, in->tie_g, V, A, in->alpha, lut, mix.get(), tc, reqs);
          scratch.assign(grid, grid + ng);
          for (const dmx::GridReq& r : reqs) scratch[((size_t)r.j * V + r.k) * A + r.n] = r.value;
          grid = scratch.data();
        }
      }
      CellCall cc = call_cell(grid, V, A, in->alpha, prior);
      // NaN likelihoods leave the reference's scans without a winner and it indexes with -1 (undefined behaviour, :816-825);
      // we fall back to index 0 so that the row is still printable (its numbers are NaN).
      if (cc.i_sing1 < 0) cc.i_sing1 = 0;
      if (cc.i_sing2 < 0) cc.i_sing2 = 0;
      if (cc.j_best < 0) { cc.j_best = 0; cc.k_best = 0; cc.n_best = 0; }
      max_llk = cc.max_llk; sum_single = cc.sum_single; sum_double = cc.sum_double;
      i_sing1 = cc.i_sing1; i_sing2 = cc.i_sing2; jb = cc.j_best; kb = cc.k_best; nb = cc.n_best;
      sing1 = grid[(size_t)i_sing1 * V * A]; sing2v = grid[(size_t)i_sing2 * V * A];
      l12 = grid[((size_t)jb * V + kb) * A + nb];
      l1 = grid[(size_t)jb * V * A]; l2 = grid[(size_t)kb * V * A];
      l10 = grid[(size_t)jb * V * A + nb];                                 // :824 pairs with sample 0 (reference quirk)
      l20 = grid[(size_t)kb * V * A + nb];                                 // :825
    } else {
      dmx_cell_summary sm = *smp;
      if (sm.i_sing1 < 0) sm.i_sing1 = 0;         // NaN likelihoods: see above
      if (sm.i_sing2 < 0) sm.i_sing2 = 0;
      if (sm.j_best < 0) { sm.j_best = 0; sm.k_best = 0; sm.n_best = 0; }
      sg = src.sing + (size_t)c * V;
      max_llk = sm.max_llk; sum_single = sm.sum_single; sum_double = sm.sum_double;
      i_sing1 = sm.i_sing1; i_sing2 = sm.i_sing2; jb = sm.j_best; kb = sm.k_best; nb = sm.n_best;
      sing1 = sg[i_sing1]; sing2v = sg[i_sing2];
      l12 = sm.llk12; l1 = sm.llk1; l2 = sm.llk2; l10 = sm.llk10; l20 = sm.llk20;
      if (arbiter && in->alpha[nb] == 0.5 && !(sm.flags & DMX_CELL_ORDER_CERTIFIED) && tc < 0) tie_missing = c;
      else if (arbiter && in->alpha[nb] == 0.5 && !(sm.flags & DMX_CELL_ORDER_CERTIFIED)) {
        // (j,k) and (k,j) are one doublet at alpha = 0.5 and differ only by rounding: re-evaluate both in the
        // reference's operation order and let its strict-< scan decide, which visits the smaller first index first.
        const int32_t a = std::min(jb, kb), b = std::max(jb, kb);
        reqs.assign({{a, b, nb, 0.0}, {b, a, nb, 0.0}});
        dmx::exact_grid_entries(*in->tie_pileup, in->tie_g, V, A, in->alpha, lut, mix.get(), tc, reqs);
        const bool swap_to_ba = reqs[0].value < reqs[1].value;
        const int32_t nj = swap_to_ba ? b : a, nk = swap_to_ba ? a : b;
        if (nj != jb) { std::swap(l1, l2); std::swap(l10, l20); }
        jb = nj; kb = nk;
        l12 = swap_to_ba ? reqs[1].value : reqs[0].value;
      }
    }

    // ---- the BEST rule's own comparisons (:837,:844).  They are made on accumulators whose device values differ from the reference's by the
    // last bit of a log here and there (<= 7e-12 STRICT on soft fields, <= 6e-11 FAST): a barcode whose margin to one of the four thresholds is
    // smaller than that would print SNG where the reference prints AMB or DBL.  K3 flags margins below 1e-7 (DMX_CELL_NEAR_RULE; the host repeats
    // the test on the values it is about to compare), and the (at most five) entries involved are then the reference's own: re-evaluated in its
    // operation order with the host libm.
    const bool rule_flag = smp && (smp->flags & DMX_CELL_NEAR_RULE) && smp->n_pairs > 0;
    if (arbiter && !host_grid && (rule_flag || dmx::near_rule(l12, l1, l2, sing1, sing2v, tol))) {
      if (tc < 0) { if (rule_flag) tie_missing = c; }                      // (unflagged: the margin is 1e-7 give or take the device's 1e-11 — nothing to decide)
      else {
        reqs.assign({{jb, kb, nb, 0.0}, {jb, 0, 0, 0.0}, {kb, 0, 0, 0.0}, {i_sing1, 0, 0, 0.0}, {i_sing2, 0, 0, 0.0}});
        dmx::exact_grid_entries(*in->tie_pileup, in->tie_g, V, A, in->alpha, lut, mix.get(), tc, reqs);
        l12 = reqs[0].value; l1 = reqs[1].value; l2 = reqs[2].value; sing1 = reqs[3].value; sing2v = reqs[4].value;
        // the rows below print these entries too: one value per entry in all files
        if (grid) {
          if (grid != scratch.data()) { scratch.assign(grid, grid + ng); grid = scratch.data(); }
          for (const dmx::GridReq& r : reqs) scratch[((size_t)r.j * V + r.k) * A + r.n] = r.value;
        } else {
          sg_fix.assign(sg, sg + V);
          for (size_t i = 1; i < reqs.size(); ++i) sg_fix[(size_t)reqs[i].j] = reqs[i].value;
          sg = sg_fix.data();
        }
      }
    }

    for (int32_t j = 0; j < V; ++j) {                                      // :746-770 (.sing2) "%s\t%s\t%d\t%d\t%d\t%d\t%.4lf\t%.4lf\t%.3lg\n"
      const double v = grid ? grid[(size_t)j * V * A] : sg[j];
      std::string& o = ck.out[0];
      o.append(head).append(in->sample_ids[j]).append(mid);
      put_fixed(o, v, 4); o.push_back('\t'); o.append(l0s); o.push_back('\t');
      put_general(o, std::exp(v - max_llk) * (1. - prior) / V / sum_single, 3); o.push_back('\n');
    }
    std::string* const pair_out = pairf.f ? &ck.out[1] : ((src.pair_rows && grid) ? &(*src.pair_rows)[(size_t)c] : nullptr);
    if (pair_out) {                                                        // :772-797 (.pair) "%s\t%s\t%s\t%.3lf\t%.5lf\t%.5lg\n"
      const double tot = sum_single + sum_double;
      std::string& o = *pair_out;
      for (int32_t j = 0; j < V; ++j) {
        const double vs = grid[(size_t)j * V * A];
        std::string hj;
        hj.append(head).append(in->sample_ids[j]).push_back('\t');
        o.append(hj).append(in->sample_ids[j]).append(alpha_txt[0]);
        put_fixed(o, vs, 5); o.push_back('\t');
        put_general(o, std::exp(vs - max_llk) * (1. - prior) / V / tot, 5); o.push_back('\n');
        for (int32_t k = 0; k < V; ++k)
          for (int32_t a = 1; a < A; ++a) {
            if (j == k) continue;
            if ((j > k) && (in->alpha[a] == 0.5)) continue;                // :785 symmetric half only
            const double v = grid[((size_t)j * V + k) * A + a];
            o.append(hj).append(in->sample_ids[k]).append(alpha_txt[(size_t)a]);
            put_fixed(o, v, 5); o.push_back('\t');
            put_general(o, std::exp(v - max_llk) * prior / V / (V - 1) / (A - 1) / tot, 5); o.push_back('\n');
          }
      }
    }
    // :816-874 (.best)
    const double sing0 = l00[0], l00b = l00[nb];
    const double post_dbl = sum_double / (sum_single + sum_double);
    const double post_sng = std::exp(sing1 - max_llk) * (1. - prior) / V / sum_single;
    appendf(ck.out[2], "%s\t%d\t%d\t%d\t%d\t", bc, t, p, u, ns);
    if ((l12 > l1) && (l12 > l2) && (l12 > sing1 + 2))                     // :837
      appendf(ck.out[2], "DBL-%s-%s-%.3lf", in->sample_ids[jb], in->sample_ids[kb], in->alpha[nb]);
    else if (sing1 > sing2v + 2)                                           // :844
      appendf(ck.out[2], "SNG-%s", in->sample_ids[i_sing1]);
    else
      appendf(ck.out[2], "AMB-%s-%s-%s/%s", in->sample_ids[i_sing1], in->sample_ids[i_sing2], in->sample_ids[jb], in->sample_ids[kb]);
    appendf(ck.out[2], "\t%s\t%.4lf", in->sample_ids[i_sing1], sing1);
    appendf(ck.out[2], "\t%s\t%.4lf\t%.4lf", in->sample_ids[i_sing2], sing2v, sing0);
    appendf(ck.out[2], "\t%s\t%s\t%.3lf\t%.4lf\t%.4lf\t%.4lf\t%.4lf\t%.4lf\t%.4lf\t%.3lg\t%.3lg\n", in->sample_ids[jb],
            in->sample_ids[kb], in->alpha[nb], l12, l1, l2, l10, l20, l00b, post_dbl, post_sng);
  }
  });
  if (frc == DMX_OK && tie_missing.load() >= 0)
    return set_error(DMX_ERR_STATE, "%s: the tie arbiter needed barcode %s, whose pileup was not staged", who, in->barcodes[tie_missing.load()]);
  return frc;
}
