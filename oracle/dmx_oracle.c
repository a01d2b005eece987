/* TEST INFRASTRUCTURE — "oracle": a plain-C CPU restatement of statgen/demuxlet's per-barcode genotype-likelihood
 * engine, written from the reference's behaviour (not copied).  It exists so the HIP path can be checked against the
 * reference's arithmetic on a box that has no /root/reference.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load it; the product never does.
 *
 * Pinning: the reference ships no tests or golden vectors for this path.  The
 * restatement is pinned against outputs of the reference's own code run in the dev container:
 *   - oracle/_ref/ref_slice_harness  = cmd_cram_demuxlet.cpp:390-881 + sc_drop_seq.cpp + PhredHelper.cpp + Error.cpp
 *     (byte-identical .single/.sing2/.best/.pair and bit-identical raw llks/llk0s/llksAB/llks00, tests/golden/),
 *   - oracle/_ref/libref_units.so    = sc_drop_seq.cpp + PhredHelper.cpp compiled alone (rows a1, a2).
 *   Row a3 (genotype-field transforms) needs htslib to run in the reference and is therefore PARITY-UNPINNED by any
 *   reference output; it is pinned only by hand-derived vectors (tests/test_host_units.py: test_geno_transforms_match_oracle,
 *   test_geno_transforms_hand_vectors).
 *
 * Build: gcc -O2 -ffp-contract=off -std=c11 (no FMA contraction: the reference is built -O2 for baseline x86-64,
 * Makefile.am:6, and FMA contraction changes .best rows — SURVEY.md F6).
 *
 * Every function cites the reference file:line it follows (paths relative to /root/reference). */
#define _POSIX_C_SOURCE 200809L
#include "dmx_oracle.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ------------------------------------------------------------------------------------------------------------ */
/* a2  PhredHelper.cpp:24-40 — only phred2Err/phred2Mat (used at cmd_cram_demuxlet.cpp:437-439,606-607) and
 *     phred2Prob (bcf_filtered_reader.cpp:279 via PhredHelper.h:45) matter to demuxlet.                         */
static double T_err[256], T_mat[256], T_prob[256];
static int T_ready = 0;
static void tables_init(void) {
  if (T_ready) return;
  for (int q = 0; q <= 255; ++q) {
    T_err[q]  = (q > 1) ? pow(0.1, q * 0.1) : 0.75;     /* PhredHelper.cpp:30 */
    T_prob[q] = pow(0.1, q * 0.1);                      /* :31 */
    T_mat[q]  = 1. - T_err[q];                          /* :32 */
  }
  T_ready = 1;
}
void orc_phred_tables(double mat[256], double err[256]) {
  tables_init();
  memcpy(mat, T_mat, sizeof T_mat); memcpy(err, T_err, sizeof T_err);
}
double orc_phred_prob(uint32_t phred) { tables_init(); return phred > 255 ? T_prob[255] : T_prob[phred]; } /* PhredHelper.h:45 */

/* ------------------------------------------------------------------------------------------------------------ */
/* a3  genotype-field transforms. Biallelic (vfilt.maxAlleles=2, cmd_cram_demuxlet.cpp:26,106), diploid
 *     (ploidies all 2, bcf_filtered_reader.cpp init_params).  Results are FLOAT32 (bcf_filtered_reader.h:78).     */

/* GT: parse_genotypes bcf_filtered_reader.cpp:230-240 (an/acs over the selected samples), get_genotype_at .h:144-149,
 *     parse_posteriors :368-406. */
void orc_geno_from_gt(const int32_t* alleles, int32_t nv, double gt_error, float* out) {
  const int32_t nalleles = 2, ngenos = 3;
  double acs[2] = {0, 0}; int32_t an = 0;
  for (int32_t i = 0; i < nv; ++i)
    for (int32_t h = 0; h < 2; ++h) {
      int32_t a = alleles[2 * i + h];
      if (a >= 0) { ++an; acs[a] += 1; }                                       /* :234-238 */
    }
  for (int32_t i = 0; i < nv; ++i) {
    int32_t a1 = alleles[2 * i], a2 = alleles[2 * i + 1];
    int32_t g = (a1 < 0 || a2 < 0) ? -1 : (a1 > a2 ? a1 * (a1 + 1) / 2 + a2 : a2 * (a2 + 1) / 2 + a1); /* .h:144-149 */
    float* o = out + 3 * i;
    if (g < 0) {                                                               /* :381-388 HWE from smoothed AF */
      int32_t l = 0;
      for (int32_t j = 0; j < nalleles; ++j)
        for (int32_t k = 0; k <= j; ++k, ++l)
          o[l] = (float)((j == k ? 1.0 : 2.0) * (acs[j] + 1.0 / nalleles) / (an + 1.0) * (acs[k] + 1.0 / nalleles) / (an + 1.0));
    } else {
      for (int32_t j = 0; j < ngenos; ++j)
        o[j] = (float)((g == j) ? 1.0 - gt_error : gt_error / (ngenos - 1.0)); /* :397-400 */
    }
  }
}

/* PL: parse_likelihoods bcf_filtered_reader.cpp:244-320 — 10 EM iterations on the allele frequencies, posterior stored
 *     on the last one.  gt_error is NOT used on this path. */
void orc_geno_from_pl(const int32_t* pl, int32_t nv, float* out) {
  tables_init();
  const int32_t niter = 10, nalleles = 2, ngenos = 3;
  double acs[2]; for (int32_t i = 0; i < nalleles; ++i) acs[i] = 1.0 / nalleles;   /* :258-259 */
  double gp[3], sumgp; int32_t an = 0;
  for (int32_t it = 0; it < niter; ++it) {
    double newacs[2] = {0, 0};
    an = 0;
    for (int32_t i = 0; i < nv; ++i) {
      const int32_t* p = pl + (size_t)i * ngenos;
      sumgp = 0;
      int32_t l = 0;
      for (int32_t j = 0; j < nalleles; ++j)
        for (int32_t k = 0; k <= j; ++k, ++l)
          sumgp += (gp[l] = (j == k ? 1 : 2) * acs[j] * acs[k] * orc_phred_prob((uint32_t)p[l]));   /* :279 */
      l = 0;
      for (int32_t j = 0; j < nalleles; ++j)
        for (int32_t k = 0; k <= j; ++k, ++l) { gp[l] /= sumgp; newacs[j] += gp[l]; newacs[k] += gp[l]; } /* :282-288 */
      an += 2;
      if (it + 1 == niter) for (l = 0; l < ngenos; ++l) out[(size_t)i * ngenos + l] = (float)gp[l];   /* :305-308 */
    }
    for (int32_t i = 0; i < nalleles; ++i) acs[i] = newacs[i] / an;              /* :310-311 */
  }
}

/* GP: parse_posteriors bcf_filtered_reader.cpp:410-453 — float normalisation, HWE-uniform pseudo-sample, then a
 *     (1-gt_error, gt_error) mix with the across-sample mean. */
void orc_geno_from_gp(const float* gp, int32_t nv, double gt_error, float* out) {
  const int32_t nalleles = 2, ngenos = 3;
  float gpSums[3];
  for (int32_t i = 0; i < nalleles; ++i)
    for (int32_t j = 0; j <= i; ++j)
      gpSums[(i + 1) * i / 2 + j] = (float)(((i == j) ? 1.0 : 2.0) / (float)(nalleles * nalleles));   /* :421-425 */
  for (int32_t i = 0; i < nv * ngenos; ++i) out[i] = gp[i];
  for (int32_t i = 0; i < nv; ++i) {
    float* o = out + (size_t)i * ngenos;
    float sumgp = 0;
    for (int32_t j = 0; j < ngenos; ++j) sumgp += o[j];                          /* :431-434 */
    for (int32_t j = 0; j < ngenos; ++j) { o[j] /= sumgp; gpSums[j] += o[j]; }   /* :435-438 */
  }
  for (int32_t j = 0; j < ngenos; ++j) gpSums[j] /= (int32_t)(nv + 1.0);         /* :441-442 */
  for (int32_t i = 0; i < nv; ++i) {
    float* o = out + (size_t)i * ngenos;
    for (int32_t j = 0; j < ngenos; ++j) o[j] = (float)((1.0 - gt_error) * o[j] + gt_error * gpSums[j]); /* :448 */
  }
}

/* ------------------------------------------------------------------------------------------------------------ */
/* a1  pileup store. The reference keeps nested std::maps (sc_drop_seq.h:40-50); only three things are observable
 *     downstream: (i) first read of a (snp,cell,umi) key fixes (allele,bq) and later ones only bump the 16-bit count
 *     (sc_drop_seq.cpp:44,53,57); (ii) the three per-cell counters (:39,:75, cmd_cram_demuxlet.cpp:295); (iii) iteration
 *     order: cells by id / snps ascending / UMIs in ascending std::string order.  Restated as an event log that is
 *     stably sorted on freeze. */
typedef struct { int32_t cell, snp; char* umi; uint32_t word; int64_t seq; } orc_event;
struct orc_store {
  char** bc; int32_t nbc, capbc;
  int32_t *totl, *pass, *uniq;
  orc_event* ev; int64_t nev, capev;
  /* open-addressing hash over (cell,snp,umi) -> event index, to answer "is this UMI new" at add time */
  int64_t* ht; int64_t htcap;
  /* barcode hash */
  int32_t* bht; int32_t bhtcap;
  /* frozen CSR */
  int frozen; int64_t npairs, nwords; int64_t* cell_off; int32_t* pair_snp; int64_t* pair_off; uint32_t* words;
};
static uint64_t hash_bytes(const char* s, uint64_t h) { while (*s) { h ^= (unsigned char)*s++; h *= 1099511628211ULL; } return h; }
static uint64_t hash_key(int32_t cell, int32_t snp, const char* umi) {
  uint64_t h = 1469598103934665603ULL; h ^= (uint32_t)cell; h *= 1099511628211ULL; h ^= (uint32_t)snp; h *= 1099511628211ULL;
  return hash_bytes(umi, h);
}
orc_store* orc_store_new(void) {
  orc_store* s = (orc_store*)calloc(1, sizeof *s);
  s->htcap = 1 << 12; s->ht = (int64_t*)malloc(sizeof(int64_t) * s->htcap); for (int64_t i = 0; i < s->htcap; ++i) s->ht[i] = -1;
  s->bhtcap = 1 << 10; s->bht = (int32_t*)malloc(sizeof(int32_t) * s->bhtcap); for (int32_t i = 0; i < s->bhtcap; ++i) s->bht[i] = -1;
  return s;
}
void orc_store_free(orc_store* s) {
  if (!s) return;
  for (int32_t i = 0; i < s->nbc; ++i) free(s->bc[i]);
  for (int64_t i = 0; i < s->nev; ++i) free(s->ev[i].umi);
  free(s->bc); free(s->totl); free(s->pass); free(s->uniq); free(s->ev); free(s->ht); free(s->bht);
  free(s->cell_off); free(s->pair_snp); free(s->pair_off); free(s->words); free(s);
}
int32_t orc_store_add_cell(orc_store* s, const char* barcode) {           /* sc_drop_seq.cpp:20-32 */
  uint64_t h = hash_bytes(barcode, 1469598103934665603ULL);
  for (uint32_t p = (uint32_t)(h & (uint32_t)(s->bhtcap - 1));; p = (p + 1) & (uint32_t)(s->bhtcap - 1)) {
    int32_t id = s->bht[p];
    if (id < 0) break;
    if (strcmp(s->bc[id], barcode) == 0) return id;
  }
  if (s->nbc == s->capbc) {
    s->capbc = s->capbc ? s->capbc * 2 : 256;
    s->bc = (char**)realloc(s->bc, sizeof(char*) * s->capbc);
    s->totl = (int32_t*)realloc(s->totl, sizeof(int32_t) * s->capbc);
    s->pass = (int32_t*)realloc(s->pass, sizeof(int32_t) * s->capbc);
    s->uniq = (int32_t*)realloc(s->uniq, sizeof(int32_t) * s->capbc);
  }
  int32_t id = s->nbc++;
  s->bc[id] = strdup(barcode); s->totl[id] = s->pass[id] = s->uniq[id] = 0;
  if ((int64_t)s->nbc * 2 > s->bhtcap) {
    s->bhtcap *= 2; s->bht = (int32_t*)realloc(s->bht, sizeof(int32_t) * s->bhtcap);
    for (int32_t i = 0; i < s->bhtcap; ++i) s->bht[i] = -1;
    for (int32_t c = 0; c < s->nbc; ++c) {
      uint64_t hh = hash_bytes(s->bc[c], 1469598103934665603ULL);
      uint32_t p = (uint32_t)(hh & (uint32_t)(s->bhtcap - 1)); while (s->bht[p] >= 0) p = (p + 1) & (uint32_t)(s->bhtcap - 1);
      s->bht[p] = c;
    }
  } else {
    uint32_t p = (uint32_t)(h & (uint32_t)(s->bhtcap - 1)); while (s->bht[p] >= 0) p = (p + 1) & (uint32_t)(s->bhtcap - 1);
    s->bht[p] = id;
  }
  return id;
}
void orc_store_count_read(orc_store* s, int32_t cell) { ++s->totl[cell]; }   /* cmd_cram_demuxlet.cpp:295 */
int32_t orc_store_add_read(orc_store* s, int32_t snp, int32_t cell, const char* umi, int32_t allele, int32_t bq) {
  ++s->pass[cell];                                                          /* sc_drop_seq.cpp:39 */
  uint64_t h = hash_key(cell, snp, umi);
  for (uint64_t p = h & (uint64_t)(s->htcap - 1);; p = (p + 1) & (uint64_t)(s->htcap - 1)) {
    int64_t e = s->ht[p];
    if (e < 0) break;
    if (s->ev[e].cell == cell && s->ev[e].snp == snp && strcmp(s->ev[e].umi, umi) == 0) {
      ++s->ev[e].word;                                                      /* :57  duplicate: only the count moves */
      return 0;
    }
  }
  if (s->nev == s->capev) { s->capev = s->capev ? s->capev * 2 : 1024; s->ev = (orc_event*)realloc(s->ev, sizeof(orc_event) * s->capev); }
  orc_event* e = &s->ev[s->nev];
  e->cell = cell; e->snp = snp; e->umi = strdup(umi); e->seq = s->nev;
  e->word = (uint32_t)(((int32_t)(char)allele << 24) | ((int32_t)(char)bq << 16) | 0x01);   /* :44,:53 */
  ++s->nev;
  if (s->nev * 2 > s->htcap) {
    s->htcap *= 2; s->ht = (int64_t*)realloc(s->ht, sizeof(int64_t) * s->htcap);
    for (int64_t i = 0; i < s->htcap; ++i) s->ht[i] = -1;
    for (int64_t i = 0; i < s->nev; ++i) {
      uint64_t p = hash_key(s->ev[i].cell, s->ev[i].snp, s->ev[i].umi) & (uint64_t)(s->htcap - 1);
      while (s->ht[p] >= 0) p = (p + 1) & (uint64_t)(s->htcap - 1);
      s->ht[p] = i;
    }
  } else {
    uint64_t p = h & (uint64_t)(s->htcap - 1); while (s->ht[p] >= 0) p = (p + 1) & (uint64_t)(s->htcap - 1);
    s->ht[p] = s->nev - 1;
  }
  ++s->uniq[cell];                                                          /* :75 */
  s->frozen = 0;
  return 1;
}
static int ev_cmp(const void* a, const void* b) {
  const orc_event* x = (const orc_event*)a; const orc_event* y = (const orc_event*)b;
  if (x->cell != y->cell) return x->cell < y->cell ? -1 : 1;
  if (x->snp != y->snp) return x->snp < y->snp ? -1 : 1;
  int c = strcmp(x->umi, y->umi);                       /* std::string operator< == unsigned byte-wise compare */
  return c;
}
void orc_store_freeze(orc_store* s) {
  if (s->frozen) return;
  orc_event* ev = (orc_event*)malloc(sizeof(orc_event) * (s->nev ? s->nev : 1));
  memcpy(ev, s->ev, sizeof(orc_event) * s->nev);
  qsort(ev, (size_t)s->nev, sizeof(orc_event), ev_cmp);  /* keys are unique, so stability is not needed */
  free(s->cell_off); free(s->pair_snp); free(s->pair_off); free(s->words);
  s->cell_off = (int64_t*)calloc((size_t)s->nbc + 1, sizeof(int64_t));
  s->pair_snp = (int32_t*)malloc(sizeof(int32_t) * (s->nev ? s->nev : 1));
  s->pair_off = (int64_t*)malloc(sizeof(int64_t) * (s->nev + 1));
  s->words = (uint32_t*)malloc(sizeof(uint32_t) * (s->nev ? s->nev : 1));
  int64_t np = 0;
  for (int64_t i = 0; i < s->nev; ++i) {
    if (i == 0 || ev[i].cell != ev[i - 1].cell || ev[i].snp != ev[i - 1].snp) {
      s->pair_snp[np] = ev[i].snp; s->pair_off[np] = i; ++np; ++s->cell_off[ev[i].cell + 1];
    }
    s->words[i] = ev[i].word;
  }
  s->pair_off[np] = s->nev;
  for (int32_t c = 0; c < s->nbc; ++c) s->cell_off[c + 1] += s->cell_off[c];
  s->npairs = np; s->nwords = s->nev; s->frozen = 1;
  free(ev);
}
int32_t orc_store_ncells(const orc_store* s) { return s->nbc; }
int64_t orc_store_npairs(const orc_store* s) { return s->npairs; }
int64_t orc_store_nwords(const orc_store* s) { return s->nwords; }
const int64_t*  orc_store_cell_off(const orc_store* s) { return s->cell_off; }
const int32_t*  orc_store_pair_snp(const orc_store* s) { return s->pair_snp; }
const int64_t*  orc_store_pair_off(const orc_store* s) { return s->pair_off; }
const uint32_t* orc_store_words(const orc_store* s) { return s->words; }
const int32_t*  orc_store_totl(const orc_store* s) { return s->totl; }
const int32_t*  orc_store_pass(const orc_store* s) { return s->pass; }
const int32_t*  orc_store_uniq(const orc_store* s) { return s->uniq; }
const char*     orc_store_barcode(const orc_store* s, int32_t c) { return s->bc[c]; }
static __thread char** g_sort_bc;            /* thread-local: orc_* calls may run concurrently (bench.py's all-cores leg) */
static int order_cmp(const void* a, const void* b) { return strcmp(g_sort_bc[*(const int32_t*)a], g_sort_bc[*(const int32_t*)b]); }
void orc_store_sorted_order(const orc_store* s, int32_t* order) {
  for (int32_t i = 0; i < s->nbc; ++i) order[i] = i;
  g_sort_bc = s->bc; qsort(order, (size_t)s->nbc, sizeof(int32_t), order_cmp);
}

/* ------------------------------------------------------------------------------------------------------------ */
/* a4..a14  the engine */
static __thread const char* const* g_sort_names;
static int name_cmp(const void* a, const void* b) { return strcmp(g_sort_names[*(const int32_t*)a], g_sort_names[*(const int32_t*)b]); }

double orc_wall_seconds(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

int orc_run(const orc_problem* P, orc_raw* raw, const char* out_prefix) {
  tables_init();
  const int32_t B = P->n_cells, S = P->n_snps, nv = P->n_samples, nAlpha = P->n_alpha;
  const double* gridAlpha = P->alpha;
  const double doublet_prior = P->doublet_prior;
  if (!P->singlet_only && (nv < 2 || nAlpha < 2)) return -1;    /* cmd_cram_demuxlet.cpp:731 (/(nv-1)/(nAlpha-1)), :821 */

  /* ascending barcode order == iteration order of std::map<std::string,int32_t> bc_map (:472,:576) */
  int32_t* order = (int32_t*)malloc(sizeof(int32_t) * (B ? B : 1));
  for (int32_t i = 0; i < B; ++i) order[i] = i;
  g_sort_names = P->barcodes; qsort(order, (size_t)B, sizeof(int32_t), name_cmp);

  /* a4 (:390-401): average genotype probability, sequential sum over samples then one division */
  double* gp0s = (double*)calloc((size_t)S * 3 + 1, sizeof(double));
  for (int32_t i = 0; i < S; ++i) {
    const float* gi = P->g + (size_t)i * nv * 3;
    for (int32_t j = 0; j < nv; ++j) { gp0s[i*3] += (double)gi[3*j]; gp0s[i*3+1] += (double)gi[3*j+1]; gp0s[i*3+2] += (double)gi[3*j+2]; }
    gp0s[i*3] /= nv; gp0s[i*3+1] /= nv; gp0s[i*3+2] /= nv;
  }

  /* a5 (:412-461).  The reference walks SNP-major; per accumulator (cell,k) the additions happen in ascending SNP
   * order, which is exactly the order of a cell's CSR row, so a cell-major walk is bit-identical. */
  double* llks  = (double*)calloc((size_t)B * nv + 1, sizeof(double));
  double* llk0s = (double*)calloc((size_t)B + 1, sizeof(double));
  for (int32_t c = 0; c < B; ++c) {
    for (int64_t p = P->cell_off[c]; p < P->cell_off[c + 1]; ++p) {
      const int32_t i = P->pair_snp[p];
      double GLs[3] = {1.0, 1.0, 1.0}, tmp;                                         /* :427 */
      for (int64_t w = P->pair_off[p]; w < P->pair_off[p + 1]; ++w) {
        uint8_t al = (P->words[w] >> 24) & 0x00ff, bq = (P->words[w] >> 16) & 0x00ff;  /* :429-430 */
        if (al == 2) continue;                                                      /* :435 */
        GLs[0] *= ((al == 0) ? T_mat[bq] : T_err[bq] / 3.0);                        /* :437 */
        GLs[1] *= (0.5 - T_err[bq] / 3.0);                                          /* :438 */
        GLs[2] *= ((al == 1) ? T_mat[bq] : T_err[bq] / 3.0);                        /* :439 */
        tmp = GLs[0] + GLs[1] + GLs[2];                                             /* :440 */
        GLs[0] /= tmp; GLs[1] /= tmp; GLs[2] /= tmp;                                /* :441-443 */
      }
      GLs[0] += 1e-6; GLs[1] += 1e-6; GLs[2] += 1e-6;                               /* :446-448 */
      tmp = GLs[0] + GLs[1] + GLs[2];
      GLs[0] /= tmp; GLs[1] /= tmp; GLs[2] /= tmp;                                  /* :449-452 */
      const float* gi = P->g + (size_t)i * nv * 3;
      for (int32_t k = 0; k < nv; ++k)
        llks[(size_t)c * nv + k] += log(GLs[0] * (double)gi[k*3] + GLs[1] * (double)gi[k*3+1] + GLs[2] * (double)gi[k*3+2]);  /* :456 */
      llk0s[c] += log(GLs[0] * gp0s[i*3] + GLs[1] * gp0s[i*3+1] + GLs[2] * gp0s[i*3+2]);                                     /* :459 */
    }
  }
  if (raw && raw->llks)  memcpy(raw->llks, llks, sizeof(double) * (size_t)B * nv);
  if (raw && raw->llk0s) memcpy(raw->llk0s, llk0s, sizeof(double) * (size_t)B);

  /* a6 (:465-527): .single */
  char path[4096];
  FILE* wsingle = NULL;
  if (out_prefix) { snprintf(path, sizeof path, "%s.single", out_prefix); wsingle = fopen(path, "w"); if (!wsingle) return -2; }
  if (wsingle) fprintf(wsingle, "BARCODE\tSM_ID\tRD.TOTL\tRD.PASS\tRD.UNIQ\tN.SNP\tLLK1\tLLK0\tPOSTPRB\n");           /* :470 */
  for (int32_t oi = 0; oi < B; ++oi) {
    const int32_t c = order[oi];
    const int32_t nsnp = (int32_t)(P->cell_off[c + 1] - P->cell_off[c]);
    double sumLLK = -1e300;                                                                                            /* :477 */
    if ((P->rd_totl[c] < P->min_total) || (P->rd_uniq[c] < P->min_uniq) || (nsnp < P->min_snp)) continue;              /* :480 */
    for (int32_t j = 0; j < nv; ++j) {
      double curLLK = llks[(size_t)c * nv + j];
      if (sumLLK > curLLK) sumLLK = sumLLK + log(1.0 + exp(curLLK - sumLLK));                                          /* :484-486 */
      else                 sumLLK = curLLK + log(1.0 + exp(sumLLK - curLLK));                                          /* :487-489 */
    }
    if (wsingle)
      for (int32_t j = 0; j < nv; ++j) {
        double curLLK = llks[(size_t)c * nv + j];
        fprintf(wsingle, "%s\t%s\t%d\t%d\t%d\t%d\t%.5lf\t%.5lf\t%.3lg\n", P->barcodes[c], P->sample_ids[j],
                P->rd_totl[c], P->rd_pass[c], P->rd_uniq[c], nsnp, curLLK, llk0s[c], exp(curLLK - sumLLK));           /* :506-516 */
      }
  }
  if (wsingle) fclose(wsingle);
  if (P->singlet_only) { free(order); free(gp0s); free(llks); free(llk0s); return 0; }

  FILE *wsing2 = NULL, *wpair = NULL, *wbest = NULL;
  if (out_prefix) {
    snprintf(path, sizeof path, "%s.sing2", out_prefix); wsing2 = fopen(path, "w");
    if (P->write_pair) { snprintf(path, sizeof path, "%s.pair", out_prefix); wpair = fopen(path, "w"); }
    snprintf(path, sizeof path, "%s.best", out_prefix); wbest = fopen(path, "w");
    if (!wsing2 || !wbest || (P->write_pair && !wpair)) return -2;
    fprintf(wsing2, "BARCODE\tSM_ID\tRD.TOTL\tRD.PASS\tRD.UNIQ\tN.SNP\tLLK1\tLLK0\tPOSTPRB\n");                        /* :533 */
    if (wpair) fprintf(wpair, "BARCODE\tSM1.ID\tSM2.ID\tLLK12\tPOSTPRB\n");                                            /* :570 (5 names, 6 fields per row) */
    fprintf(wbest, "BARCODE\tRD.TOTL\tRD.PASS\tRD.UNIQ\tN.SNP\tBEST\tSNG.1ST\tSNG.LLK1\tSNG.2ND\tSNG.LLK2\tSNG.LLK0\tDBL.1ST\tDBL.2ND\tALPHA\tLLK12\tLLK1\tLLK2\tLLK10\tLLK20\tLLK00\tPRB.DBL\tPRB.SNG1\n"); /* :571 */
  }

  /* a7 (:542-560) is NOT materialised: gpAB[l][m] = g_j[l]*g_k[m] is a product of two float32-valued doubles and hence
   * exact in binary64 (24+24 <= 53 bits); computing it at the point of use gives the very same double. gpA0/llksA0
   * (:686-696) are never read by any output and are skipped. */
  const size_t nAB = (size_t)nv * nv * nAlpha;
  double* llksAB = (double*)malloc(sizeof(double) * nAB);
  double* llks00 = (double*)malloc(sizeof(double) * nAlpha);
  double* pGs    = (double*)malloc(sizeof(double) * nAlpha * 9);
  double* sumPs  = (double*)malloc(sizeof(double) * nAlpha);
  for (int32_t oi = 0; oi < B; ++oi) {                                                                                  /* :576 */
    const int32_t i = order[oi];
    const int32_t nsnp = (int32_t)(P->cell_off[i + 1] - P->cell_off[i]);
    if ((P->rd_totl[i] < P->min_total) || (P->rd_uniq[i] < P->min_uniq) || (nsnp < P->min_snp)) continue;              /* :581 */
    memset(llksAB, 0, sizeof(double) * nAB); memset(llks00, 0, sizeof(double) * nAlpha);                               /* :583-586 */
    if (nsnp == 0) continue;                                                                                            /* :592 */
    for (int64_t p = P->cell_off[i]; p < P->cell_off[i + 1]; ++p) {                                                     /* :595 */
      const int32_t isnp = P->pair_snp[p];
      for (int32_t q = 0; q < nAlpha * 9; ++q) pGs[q] = 1.0;                                                            /* :597 */
      for (int64_t w = P->pair_off[p]; w < P->pair_off[p + 1]; ++w) {                                                   /* :600 */
        uint8_t al = (P->words[w] >> 24) & 0x00ff, bq = (P->words[w] >> 16) & 0x00ff;
        if (al == 2) continue;                                                                                          /* :604 */
        double pR = (al == 0) ? T_mat[bq] : T_err[bq] / 3.0;                                                            /* :606 */
        double pA = (al == 1) ? T_mat[bq] : T_err[bq] / 3.0;                                                            /* :607 */
        double maxpG = 0;
        for (int32_t k = 0; k < nAlpha; ++k)
          for (int32_t l = 0; l < 3; ++l)
            for (int32_t m = 0; m < 3; ++m) {
              double p_ = 0.5 * l + (m - l) * 0.5 * gridAlpha[k];                                                       /* :613 */
              double* pG = &pGs[k * 9 + l * 3 + m];
              *pG *= (pR * (1.0 - p_) + pA * p_);                                                                       /* :625 */
              if (maxpG < *pG) maxpG = *pG;                                                                             /* :626-627 */
            }
        for (int32_t q = 0; q < nAlpha * 9; ++q) pGs[q] /= maxpG;                                                       /* :632-639 */
      }
      double maxpG = 0;
      for (int32_t q = 0; q < nAlpha * 9; ++q) { pGs[q] += 1e-6; if (maxpG < pGs[q]) maxpG = pGs[q]; }                  /* :643-654 */
      for (int32_t q = 0; q < nAlpha * 9; ++q) pGs[q] /= maxpG;                                                         /* :656-663 */

      const float* gi = P->g + (size_t)isnp * nv * 3;
      for (int32_t j = 0; j < nv; ++j)                                                                                  /* :671 */
        for (int32_t k = 0; k < nv; ++k) {                                                                              /* :673 */
          for (int32_t n = 0; n < nAlpha; ++n) sumPs[n] = 0;
          for (int32_t l = 0; l < 3; ++l)
            for (int32_t m = 0; m < 3; ++m) {
              double pp = (double)gi[j*3+l] * (double)gi[k*3+m];                                                        /* :553 via :677 */
              for (int32_t n = 0; n < nAlpha; ++n) sumPs[n] += (pp * pGs[n*9 + l*3 + m]);                                /* :678-679 */
            }
          for (int32_t n = 0; n < nAlpha; ++n) llksAB[(size_t)j*nv*nAlpha + k*nAlpha + n] += log(sumPs[n]);             /* :682-683 */
        }
      for (int32_t n = 0; n < nAlpha; ++n) sumPs[n] = 0;                                                                /* :699 */
      for (int32_t l = 0; l < 3; ++l)
        for (int32_t m = 0; m < 3; ++m) {
          double pp = gp0s[isnp*3+l] * gp0s[isnp*3+m];                                                                  /* :555 via :702 */
          for (int32_t n = 0; n < nAlpha; ++n) sumPs[n] += (pp * pGs[n*9 + l*3 + m]);
        }
      for (int32_t n = 0; n < nAlpha; ++n) llks00[n] += log(sumPs[n]);                                                  /* :708-709 */
    }
    if (raw && raw->llksAB) memcpy(raw->llksAB + (size_t)i * nAB, llksAB, sizeof(double) * nAB);
    if (raw && raw->llks00) memcpy(raw->llks00 + (size_t)i * nAlpha, llks00, sizeof(double) * nAlpha);
    if (raw && raw->processed) raw->processed[i] = 1;

    /* a10 (:713-734) */
    double maxLLK = -1e300;
    for (size_t q = 0; q < nAB; ++q) if (maxLLK < llksAB[q]) maxLLK = llksAB[q];
    double sumSingle = 0, sumDouble = 0;
    for (int32_t j = 0; j < nv; ++j) {
      sumSingle += (exp(llksAB[(size_t)j*nv*nAlpha] - maxLLK) * (1. - doublet_prior) / nv);                             /* :726 */
      for (int32_t k = 0; k < nv; ++k) {
        if (j == k) continue;
        for (int32_t n = 1; n < nAlpha; ++n)
          sumDouble += (exp(llksAB[(size_t)j*nv*nAlpha + k*nAlpha + n] - maxLLK) * doublet_prior / nv / (nv - 1) / (nAlpha - 1) / (gridAlpha[n] == 0.5 ? 2.0 : 1.0)); /* :731 */
      }
    }
    /* a11 (:746-770) */
    int32_t iSing1 = -1, iSing2 = -1; double maxSing1 = -1e300, maxSing2 = -1e300;
    for (int32_t j = 0; j < nv; ++j) {
      double v = llksAB[(size_t)j*nv*nAlpha];
      if (maxSing1 < v) { maxSing2 = maxSing1; iSing2 = iSing1; iSing1 = j; maxSing1 = v; }
      else if (maxSing2 < v) { iSing2 = j; maxSing2 = v; }
      if (wsing2)
        fprintf(wsing2, "%s\t%s\t%d\t%d\t%d\t%d\t%.4lf\t%.4lf\t%.3lg\n", P->barcodes[i], P->sample_ids[j],
                P->rd_totl[i], P->rd_pass[i], P->rd_uniq[i], nsnp, v, llks00[0],
                exp(v - maxLLK) * (1. - doublet_prior) / nv / sumSingle);                                               /* :759-769 */
    }
    /* a12 (:772-797) */
    if (wpair)
      for (int32_t j = 0; j < nv; ++j) {
        fprintf(wpair, "%s\t%s\t%s\t%.3lf\t%.5lf\t%.5lg\n", P->barcodes[i], P->sample_ids[j], P->sample_ids[j], gridAlpha[0],
                llksAB[(size_t)j*nv*nAlpha], exp(llksAB[(size_t)j*nv*nAlpha] - maxLLK) * (1. - doublet_prior) / nv / (sumSingle + sumDouble));
        for (int32_t k = 0; k < nv; ++k)
          for (int32_t n = 0; n < nAlpha; ++n)
            if ((n > 0) && (j != k)) {
              if ((j > k) && (gridAlpha[n] == 0.5)) continue;                                                           /* :785 */
              double v = llksAB[(size_t)j*nv*nAlpha + k*nAlpha + n];
              fprintf(wpair, "%s\t%s\t%s\t%.3lf\t%.5lf\t%.5lg\n", P->barcodes[i], P->sample_ids[j], P->sample_ids[k], gridAlpha[n],
                      v, exp(v - maxLLK) * doublet_prior / nv / (nv - 1) / (nAlpha - 1) / (sumSingle + sumDouble));     /* :786-792 */
            }
      }
    /* a13 (:799-874) */
    int32_t jBest = -1, kBest = -1, alphaBest = -1; double maxAB = -1e300;
    for (int32_t j = 0; j < nv; ++j)
      for (int32_t k = 0; k < nv; ++k) {
        if (j == k) continue;
        for (int32_t n = 1; n < nAlpha; ++n) {
          double v = llksAB[(size_t)j*nv*nAlpha + k*nAlpha + n];
          if (maxAB < v) { jBest = j; kBest = k; alphaBest = n; maxAB = v; }                                            /* :806 strict < */
        }
      }
    double singLLK1 = llksAB[(size_t)iSing1*nv*nAlpha];
    double singLLK2 = llksAB[(size_t)iSing2*nv*nAlpha];
    double singLLK0 = llks00[0];
    double pairLLK12 = llksAB[(size_t)jBest*nv*nAlpha + kBest*nAlpha + alphaBest];
    double pairLLK1  = llksAB[(size_t)jBest*nv*nAlpha];
    double pairLLK2  = llksAB[(size_t)kBest*nv*nAlpha];
    double pairLLK10 = llksAB[(size_t)jBest*nv*nAlpha + alphaBest];     /* :824  (pair with sample 0 — reference quirk) */
    double pairLLK20 = llksAB[(size_t)kBest*nv*nAlpha + alphaBest];     /* :825 */
    double pairLLK00 = llks00[alphaBest];
    double postDoublet = sumDouble / (sumSingle + sumDouble);
    double postSinglet = exp(singLLK1 - maxLLK) * (1. - doublet_prior) / nv / sumSingle;
    if (wbest) {
      fprintf(wbest, "%s\t%d\t%d\t%d\t%d\t", P->barcodes[i], P->rd_totl[i], P->rd_pass[i], P->rd_uniq[i], nsnp);
      if ((pairLLK12 > pairLLK1) && (pairLLK12 > pairLLK2) && (pairLLK12 > singLLK1 + 2))                               /* :837 */
        fprintf(wbest, "DBL-%s-%s-%.3lf", P->sample_ids[jBest], P->sample_ids[kBest], gridAlpha[alphaBest]);
      else if (singLLK1 > singLLK2 + 2)                                                                                 /* :844 */
        fprintf(wbest, "SNG-%s", P->sample_ids[iSing1]);
      else
        fprintf(wbest, "AMB-%s-%s-%s/%s", P->sample_ids[iSing1], P->sample_ids[iSing2], P->sample_ids[jBest], P->sample_ids[kBest]);
      fprintf(wbest, "\t%s\t%.4lf", P->sample_ids[iSing1], singLLK1);
      fprintf(wbest, "\t%s\t%.4lf\t%.4lf", P->sample_ids[iSing2], singLLK2, singLLK0);
      fprintf(wbest, "\t%s\t%s\t%.3lf\t%.4lf\t%.4lf\t%.4lf\t%.4lf\t%.4lf\t%.4lf\t%.3lg\t%.3lg\n", P->sample_ids[jBest], P->sample_ids[kBest],
              gridAlpha[alphaBest], pairLLK12, pairLLK1, pairLLK2, pairLLK10, pairLLK20, pairLLK00, postDoublet, postSinglet); /* :862-873 */
    }
  }
  if (wpair) fclose(wpair);
  if (wbest) fclose(wbest);
  if (wsing2) fclose(wsing2);
  free(llksAB); free(llks00); free(pGs); free(sumPs); free(order); free(gp0s); free(llks); free(llk0s);
  return 0;
}
