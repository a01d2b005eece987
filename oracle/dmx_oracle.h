/* TEST INFRASTRUCTURE — CPU restatement ("oracle") of demuxlet's per-barcode genotype-likelihood engine.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this. The product (libdmx.so)
 * never includes, links or calls anything under oracle/.   See dmx_oracle.c for the per-function citations. */
#ifndef DMX_ORACLE_H
#define DMX_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---- a2: phred LUT (PhredHelper.cpp:24-40) ---- */
void orc_phred_tables(double mat[256], double err[256]);
double orc_phred_prob(uint32_t phred);           /* phredConverter::toProb, PhredHelper.h:45 */

/* ---- a3: genotype field -> float32 probabilities (bcf_filtered_reader.cpp:186-242,244-320,360-454) ---- */
/* alleles[2*i+h] = allele index of haplotype h of selected sample i, or -1 when missing ("."). biallelic, diploid. */
void orc_geno_from_gt(const int32_t* alleles, int32_t nv, double gt_error, float* out /*[nv*3]*/);
/* pl[3*i+g] = PL of selected sample i (INT32_MIN = missing) */
void orc_geno_from_pl(const int32_t* pl, int32_t nv, float* out /*[nv*3]*/);
/* gp[3*i+g] = raw GP floats of selected sample i */
void orc_geno_from_gp(const float* gp, int32_t nv, double gt_error, float* out /*[nv*3]*/);

/* ---- a1: UMI-deduplicated pileup store (sc_drop_seq.h:34-58, sc_drop_seq.cpp:3-77) ---- */
typedef struct orc_store orc_store;
orc_store* orc_store_new(void);
void   orc_store_free(orc_store*);
int32_t orc_store_add_cell(orc_store*, const char* barcode);                 /* sc_drop_seq.cpp:20-32 */
void   orc_store_count_read(orc_store*, int32_t cell);                       /* ++cell_totl_reads, cmd_cram_demuxlet.cpp:295 */
int32_t orc_store_add_read(orc_store*, int32_t snp, int32_t cell, const char* umi, int32_t allele, int32_t bq); /* :34-77; 1 = new UMI */
/* Freeze into CSR in the reference's iteration order: cells by id; pairs ascending snp (std::map<int32_t,..>);
 * reads ascending UMI string (std::map<std::string,..>). words = (allele<<24)|(bq<<16)|count. */
void   orc_store_freeze(orc_store*);
int32_t orc_store_ncells(const orc_store*);
int64_t orc_store_npairs(const orc_store*);
int64_t orc_store_nwords(const orc_store*);
const int64_t*  orc_store_cell_off(const orc_store*);   /* [ncells+1] into pairs */
const int32_t*  orc_store_pair_snp(const orc_store*);   /* [npairs] */
const int64_t*  orc_store_pair_off(const orc_store*);   /* [npairs+1] into words */
const uint32_t* orc_store_words(const orc_store*);      /* [nwords] */
const int32_t*  orc_store_totl(const orc_store*);
const int32_t*  orc_store_pass(const orc_store*);
const int32_t*  orc_store_uniq(const orc_store*);
const char*     orc_store_barcode(const orc_store*, int32_t cell);
/* order[i] = cell id of the i-th barcode in ascending byte-wise string order (std::map<std::string,int32_t>) */
void   orc_store_sorted_order(const orc_store*, int32_t* order);

/* ---- a4..a14: the engine and finaliser (cmd_cram_demuxlet.cpp:390-881) ---- */
typedef struct {
  int32_t n_cells, n_snps, n_samples, n_alpha;
  const double* alpha;            /* [n_alpha]; alpha[0] is treated as the singlet entry whatever its value */
  double  doublet_prior;
  int32_t min_total, min_uniq, min_snp;
  int32_t write_pair;
  /* pileup, CSR (as produced by orc_store_freeze) */
  const int64_t*  cell_off;  const int32_t* pair_snp;  const int64_t* pair_off;  const uint32_t* words;
  const int32_t  *rd_totl, *rd_pass, *rd_uniq;
  const float*    g;              /* [n_snps][n_samples][3] */
  const char* const* sample_ids;  /* [n_samples] */
  const char* const* barcodes;    /* [n_cells] by id */
  int32_t singlet_only;           /* 1: stop after the .single stage (BASELINE config 2) */
} orc_problem;

typedef struct {
  double* llks;      /* [n_cells][V]        cmd_cram_demuxlet.cpp:412,455-458 */
  double* llk0s;     /* [n_cells]           :413,:459 */
  double* llksAB;    /* [n_cells][V][V][A]  :548,:682-683 ; rows of skipped cells are left untouched (caller zeroes) */
  double* llks00;    /* [n_cells][A]        :550,:708-709 */
  uint8_t* processed;/* [n_cells] 1 when the cell went through the doublet loop (:581,:592) */
} orc_raw;

/* Runs the engine. Any of raw->* may be NULL. out_prefix may be NULL (no files). Returns 0, or <0 on a
 * precondition error (n_samples<2 or n_alpha<2 for the doublet stage: division by zero / index -1 at :731,:821). */
int orc_run(const orc_problem*, orc_raw* raw, const char* out_prefix);

/* bounded timing helper for bench.py's cpu_baseline leg: identical arithmetic, no files */
double orc_wall_seconds(void);

#ifdef __cplusplus
}
#endif
#endif
