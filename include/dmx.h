/* dmx.h — C-ABI of libdmx, the MI355X-native (gfx950, HIP) replacement for demuxlet's per-barcode
 * genotype-likelihood engine.
 *
 * The reference (statgen/demuxlet) has no plugin/FFI interface: the engine is inline in main()
 * (cmd_cram_demuxlet.cpp:390-881) between two in-memory cuts:
 *     B1  after the BAM x VCF scan:  sc_dropseq_lib_t scl (sc_drop_seq.h:34-58) + per-SNP genotype probabilities
 *     B2  before the text writers:   llks[B][V], llk0s[B], llksAB[V][V][A] per cell, llks00[A] per cell
 * This header IS that boundary.  Every entry point names the reference lines it replaces.  A maintainer keeps
 * cmd_cram_demuxlet.cpp:1-388 (options, htslib scan) and replaces :390-881 by the calls shown in INTEGRATION.md.
 *
 * Conventions: plain C, plain pointers and sizes, no C++/torch types.  Every function returns DMX_OK (0) or a negative
 * dmx_status; dmx_last_error() gives the message of the last failure on the calling thread.  The library never exits
 * or throws across the ABI (the reference's error() prints and throws, Error.cpp:27-41; the caller decides).
 * Handles are not thread-safe; one engine drives one GPU on one HIP stream.  There is NO CPU fallback: engine entry
 * points fail with DMX_ERR_NOGPU / DMX_ERR_HIP when no gfx950 device is usable.
 */
#ifndef DMX_H
#define DMX_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define DMX_ABI_VERSION 8   /* 4: dmx_store_add_batch, dmx_engine_mean_kernel_times; 5: dmx_device_warm_up, dmx_engine_run; 6: dmx_engine_get_cell_grids;
                               7: DMX_CELL_NEAR_RULE, dmx_write_doublet_summary_grids, dmx_engine_kernel_names, dmx_debug_device_log2_lite.  ABI 6 had grown dmx_final_input in place by a
                               trailing `cell_grid` member; a by-pointer input struct without a size member cannot grow (a caller compiled against ABI 5
                               passes a shorter object), so ABI 7 WITHDRAWS that member — the struct has its ABI 5 layout again and the grids travel as an
                               argument of the new entry point.  Additions only otherwise: callers of ABI <= 5 run unchanged.
                               8: dmx_engine_format_pair / dmx_pair_text_* (`.pair` rows formatted on the device).  Additions only. */

typedef enum {
  DMX_OK = 0,
  DMX_ERR_ARG = -1,      /* bad argument / precondition (e.g. n_samples < 2 for the doublet stage, cmd_cram_demuxlet.cpp:731,:821) */
  DMX_ERR_HIP = -2,      /* a HIP runtime call failed */
  DMX_ERR_STATE = -3,    /* call order (e.g. run before set_pileup) */
  DMX_ERR_IO = -4,       /* cannot create an output file (cmd_cram_demuxlet.cpp:409-410,:535-536) */
  DMX_ERR_NOGPU = -5,    /* no usable gfx950 device */
  DMX_ERR_NOMEM = -6
} dmx_status;

enum { DMX_MEM_HOST = 0, DMX_MEM_DEVICE = 1 };

int         dmx_abi_version(void);
const char* dmx_last_error(void);

/* ------------------------------------------------------------------------------------------------------------------
 * a2  phred LUT — replaces the global phredConv (PhredHelper.cpp:24-40): err[q] = q>1 ? pow(0.1, q*0.1) : 0.75,
 *     mat[q] = 1-err[q].  Computed with the HOST libm so that the device uses the very doubles the reference would. */
int dmx_phred_tables(double mat[256], double err[256]);

/* ------------------------------------------------------------------------------------------------------------------
 * a3  genotype FORMAT field -> float32 probability triplets (replaces BCFFilteredReader::parse_posteriors,
 *     bcf_filtered_reader.cpp:360-454, parse_genotypes :186-242, parse_likelihoods :244-320) for one biallelic,
 *     diploid record.  Inputs are the raw htslib arrays restricted to the selected samples:
 *       alleles[2*i+h]  allele index (bcf_gt_allele) of haplotype h, or -1 when missing
 *       pl[3*i+g]       PL integers (INT32_MIN = bcf_int32_missing)
 *       gp[3*i+g]       GP floats
 *     out[3*i+g] is what get_posterior_at(3*i+g) would return (bcf_filtered_reader.h:159-161). */
int dmx_geno_from_gt(const int32_t* alleles, int32_t n_samples, double gt_error, float* out);
int dmx_geno_from_pl(const int32_t* pl, int32_t n_samples, float* out);      /* --geno-error is not used on the PL path */
int dmx_geno_from_gp(const float* gp, int32_t n_samples, double gt_error, float* out);

/* ------------------------------------------------------------------------------------------------------------------
 * a1  UMI-deduplicated pileup store — replaces sc_dropseq_lib_t (sc_drop_seq.h:34-58, sc_drop_seq.cpp:3-77).
 *     Same call sequence as the reference's scan loop: add_snp per VCF record (:184,:231), add_cell per read (:262),
 *     count_read per read (:295 ++cell_totl_reads), add_read per (read, overlapping SNP) (:325).
 *     add_read returns 1 when the (snp,cell,umi) key is new (first observation wins), 0 for a duplicate
 *     (sc_drop_seq.cpp:44,53,57), <0 on error. */
typedef struct dmx_store dmx_store;
dmx_store* dmx_store_new(void);
void       dmx_store_free(dmx_store*);
int32_t    dmx_store_add_snp(dmx_store*);                              /* returns the new snp id */
int32_t    dmx_store_add_cell(dmx_store*, const char* barcode);        /* returns the (new or existing) cell id */
int        dmx_store_count_read(dmx_store*, int32_t cell);
int        dmx_store_add_read(dmx_store*, int32_t snp, int32_t cell, const char* umi, int32_t allele, int32_t bq);
/* n dmx_store_add_read calls in the order given (item i: snp[i], cell[i], the umi_len[i] bytes at umi_pool + umi_off[i], allele[i],
 * bq[i]); is_new[i] (optional) receives what the i-th call would have returned.  Observations of different cells are inserted on up to
 * n_threads host threads (0 = all): a (snp, cell, umi) key lives in one cell shard, and inside a shard the given order is kept, which is
 * all that "first observation wins" can see — the store ends up exactly as after the n single calls.  Used by the `demuxlet` binary's
 * scan, which overlaps reads and SNPs on all host cores and hands the observations over window by window, in BAM order. */
int        dmx_store_add_batch(dmx_store*, int64_t n, const int32_t* snp, const int32_t* cell, const char* umi_pool, const uint64_t* umi_off,
                               const uint32_t* umi_len, const uint8_t* allele, const uint8_t* bq, uint8_t* is_new, int32_t n_threads);
int32_t    dmx_store_n_cells(const dmx_store*);
int32_t    dmx_store_n_snps(const dmx_store*);
const char* dmx_store_barcode(const dmx_store*, int32_t cell);

/* The pileup in the layout the GPU consumes (SoA/CSR).  Cells are in id order; a cell's pairs are in ascending SNP id
 * (iteration order of std::map<int32_t,...>, cmd_cram_demuxlet.cpp:595); a pair's reads are in ascending UMI byte
 * order (std::map<std::string,...>, :428,:600) because the per-read renormalisation makes that order observable.
 * Reads with allele 2 ("neither REF nor ALT") are not stored: both likelihood loops skip them (:435,:604); they still
 * create their pair (N.SNP) and count in the read counters. */
typedef struct {
  int32_t  n_cells, n_snps;
  int64_t  n_pairs, n_reads;
  const int64_t* cell_pair_off;   /* [n_cells+1] first pair of each cell */
  const int64_t* cell_read_off;   /* [n_cells+1] first read byte of each cell */
  const int32_t* pair_snp;        /* [n_pairs] SNP id, or NULL = dense layout (every cell has n_snps pairs, pair t is SNP t) */
  const void*    pair_nrd;        /* [n_pairs] stored reads of the pair, nrd_width bytes each */
  int32_t        nrd_width;       /* 1, 2 or 4 */
  int32_t        memory;          /* DMX_MEM_HOST or DMX_MEM_DEVICE — where ALL the arrays above and below live */
  const uint8_t* reads;           /* [n_reads] (allele<<7)|bq, allele in {0,1}, bq <= 127 */
  /* per-cell counters of the pileup stage; host memory always; only the finaliser reads them (may be NULL for dmx_engine_*) */
  const int32_t *rd_totl, *rd_pass, *rd_uniq;   /* RD.TOTL (:295) RD.PASS (sc_drop_seq.cpp:39) RD.UNIQ (:75) */
} dmx_pileup;

/* Freeze the store into a dmx_pileup (host memory, owned by the store, valid until the next add_* or free). */
int dmx_store_freeze(dmx_store*, dmx_pileup* out);

/* ------------------------------------------------------------------------------------------------------------------
 * a4,a5,a7,a8,a9,a10  the likelihood engine on one MI355X — replaces cmd_cram_demuxlet.cpp:390-401 (gp0s), :412-461
 *     (singlet accumulation), :542-560 (pair tables, never materialised) and :576-734 (doublet grid + per-cell sums). */
typedef struct dmx_engine dmx_engine;

typedef struct {
  int32_t n_samples;           /* V  = vr.get_nsamples() */
  int32_t n_alpha;             /* A  = gridAlpha.size(); alpha[0] is the singlet entry whatever its value (:726,:730) */
  const double* alpha;         /* [A] */
  double  doublet_prior;       /* --doublet-prior */
  int32_t device;              /* HIP device ordinal */
  int32_t mode;                /* DMX_MODE_STRICT (0): reference operation order, no FMA contraction, IEEE division; or DMX_MODE_FAST */
  int32_t flags;               /* DMX_ENGINE_* */
  int32_t reserved[3];
} dmx_engine_config;
enum { DMX_MODE_STRICT = 0,
       /* Same ownership and accumulation order as STRICT.  Inside a doublet term the nine-term sum of cmd_cram_demuxlet.cpp:677-679
        * is factored as g_j . (pG[n] g_k) with fused multiply-adds, and for the default grid {0, 0.5} only the
        * entries demuxlet prints or decides on are evaluated: llksAB[j][0][0] and one of llksAB[j][k][1] / [k][j][1] (mirrored);
        * llksAB[j][k != 0][0], read by the maxLLK scan (:713-721) only, is filled with llksAB[j][0][0].
        * A printed log-likelihood moves by <= ~6e-11 (tests bound it by 1e-9 against the reference); .best stays identical.
        * Other alpha grids that start with 0 (`--alpha` is multi-valued, cmd_cram_demuxlet.cpp:57) evaluate the singlet column and every
        * (j, k) of the alphas n >= 1 in the same bilinear form (soft fields, up to 128 samples); GT inputs with such grids, grids with
        * alpha[0] != 0 and wider panels (soft fields: 512 samples on the default grid, 128 on the others; GT inputs: 64) run the STRICT kernels,
        * whose results FAST's contract includes. */
       DMX_MODE_FAST = 1 };
enum { DMX_ENGINE_NO_CERTIFY = 1   /* skip the device-side tie-order certificate (K3b): for callers that do not need the
                                      reference's DBL-a-b / DBL-b-a order (dmx_job.arbiter = 0 sets it) */ };

/* per-cell result of the device-side reduction (K3) — what :713-734 and :746-770,:799-828 derive from one cell's grid */
typedef struct {
  double  max_llk;             /* :713-721 */
  double  sum_single;          /* :726 */
  double  sum_double;          /* :728-733 */
  double  sing_llk1, sing_llk2;/* llksAB[iSing1][0][0], llksAB[iSing2][0][0]  (:816-817) */
  double  llk12, llk1, llk2, llk10, llk20;   /* :820-825 */
  double  llk00_0, llk00_best; /* llks00[0] (:819), llks00[alphaBest] (:826) */
  int32_t i_sing1, i_sing2;    /* :746-758 (first maximum wins; second = first maximum of the rest) */
  int32_t j_best, k_best, n_best;            /* :799-814 (strict <: lowest (j,k,n) scan index among equal maxima) */
  int32_t n_pairs;             /* N.SNP of the cell; 0 => the cell has no .best row (:592) */
  int32_t flags;               /* DMX_CELL_* : decisions that sit within 1e-7 of an alternative other than the (j,k)/(k,j) mirror */
  int32_t reserved;
  double  llk_ab, llk_ba;      /* with DMX_CELL_ORDER_CERTIFIED: llksAB[a][b][n_best] and llksAB[b][a][n_best], a = min(j_best, k_best),
                                  b = max, exactly as the reference computes them (what the host tie arbiter would re-evaluate) */
  /* with DMX_CELL_ORDER_RESOLVABLE: each accumulator is one of two known doubles, and which one hangs on what the reference's
   * libm returns for ONE log():  llksAB[a][b] = llk_ab if log(ev_x_ab) == ev_t_ab, llk_ab_alt if it is the next double above
   * ev_t_ab (anything else: no statement); likewise (b,a).  An accumulator with llk_*_alt == llk_* is already certain. */
  double  llk_ab_alt, llk_ba_alt;
  double  ev_x_ab, ev_t_ab, ev_x_ba, ev_t_ba;
} dmx_cell_summary;
enum { DMX_CELL_NEAR_DOUBLET = 1,   /* another doublet entry (not the alpha = 0.5 mirror of the best one) within 1e-7 of the best */
       DMX_CELL_NEAR_SINGLET = 2,   /* the best two singlets within 1e-7 of each other, or a third within 1e-7 of the second */
       DMX_CELL_ORDER_CERTIFIED = 4, /* the order (j_best, k_best) of an alpha = 0.5 best doublet and llk12 are the reference's, bit for
                                       bit (device certificate, DESIGN.md "Ties"): the host tie arbiter has nothing left to decide */
       DMX_CELL_ORDER_RESOLVABLE = 8,/* the certificate stayed open over a single log() per accumulator: one host log() call each
                                       (dmx_write_doublet* do it) yields the reference's order and llk12 without the pileup */
       DMX_CELL_NEAR_RULE = 16      /* (ABI 7) one of the four comparisons of the BEST rule (cmd_cram_demuxlet.cpp:837,:844) — LLK12 > LLK1, LLK12 > LLK2,
                                       LLK12 > SNG.LLK1 + 2, SNG.LLK1 > SNG.LLK2 + 2 — has a margin below 1e-7: the device's log differs from libm's
                                       in the last bit of ~1.5 % of its evaluations, so SNG / DBL / AMB of such a barcode is decided by the writers from
                                       the (at most six) entries involved re-evaluated in the reference's operation order with the host libm */ };

int dmx_engine_create(const dmx_engine_config*, dmx_engine** out);
int dmx_engine_destroy(dmx_engine*);
/* Run every later launch/copy on this hipStream_t (e.g. torch's current stream). NULL = the engine's own stream. */
int dmx_engine_set_stream(dmx_engine*, void* hip_stream);
/* The phred LUT (host doubles from dmx_phred_tables, or the caller's own). Optional: defaults to dmx_phred_tables. */
int dmx_engine_set_phred_tables(dmx_engine*, const double mat[256], const double err[256]);
/* g[n_snps][n_samples][3] float32 (HOST or DEVICE memory). Also computes gp0s[n_snps][3] on the device (:390-401). */
int dmx_engine_set_genotypes(dmx_engine*, const float* g, int32_t n_snps, int32_t memory);
/* Stage the pileup: HOST arrays are copied to HBM; DEVICE arrays are adopted (caller keeps them alive). */
int dmx_engine_set_pileup(dmx_engine*, const dmx_pileup*);
/* K1: llks[B][V], llk0s[B] (:412-461).  Asynchronous on the engine's stream. */
int dmx_engine_run_singlet(dmx_engine*);
/* K2 (+K3): llksAB[B][V][V][A], llks00[B][A] (:576-710) and the per-cell summaries (:713-734,:746-758,:799-828). */
int dmx_engine_run_doublet(dmx_engine*);
/* Both in one call (ABI 5): K1 runs beside K2 on a low-priority stream of the engine and fills the slots K2's last round leaves free; fork
 * and join are events on the engine's stream (dmx_engine_set_stream), so the call orders like run_singlet + run_doublet.  Same bits. */
int dmx_engine_run(dmx_engine*);
int dmx_engine_sync(dmx_engine*);
/* Device->host copies of the results (any pointer may be NULL). Synchronises. */
int dmx_engine_get_singlet(dmx_engine*, double* llks, double* llk0s);
int dmx_engine_get_doublet(dmx_engine*, double* llksAB, double* llks00, dmx_cell_summary* summary);
/* sing[B][V] = llksAB[c][j][0][0], the singlet column of the grid (what .sing2 prints, :746-770) — lets a caller skip
 * the V*V*A grid entirely when --write-pair is off. */
int dmx_engine_get_sing(dmx_engine*, double* sing);
/* llksAB[V][V][A] of the n cells cells[0..n) (ids of the staged pileup) -> out[n][V][V][A]: the grids of the barcodes whose K3 record
 * carries a near-tie flag are all that a records-only consumer (dmx_write_doublet_summary, a multi-GPU gather) needs besides the records. */
int dmx_engine_get_cell_grids(dmx_engine*, const int32_t* cells, int32_t n, double* out);

/* (ABI 8) The `.pair` rows of `--write-pair` (cmd_cram_demuxlet.cpp:772-797, "%s\t%s\t%s\t%.3lf\t%.5lf\t%.5lg\n") formatted ON THE DEVICE from the grid that
 * already lies in HBM, packed in output order, ready for write(2) — instead of V + V(V-1)(A-1) rows per barcode through host threads.
 *   request   n_out barcodes in output order: `cells` = their ids in the staged pileup, `barcodes` / `sample_ids` the strings to print;
 *             host_rows[i] != 0: the caller prints this barcode's rows itself (the barcodes whose grid entries the tie arbiter may replace —
 *             dmx::cell_needs, a BEST-rule comparison within 1e-7 — dmx_demuxlet_run decides); ovr[i].n >= 0: the two certified entries of an
 *             alpha = 0.5 best doublet, llksAB[a][b][n] = llk_ab and llksAB[b][a][n] = llk_ba, to print instead of the device's own.
 *   text      LLK (`%.5lf`) is exact 128-bit integer arithmetic (printf's digits); POSTPRB (`%.5lg`) is printed only where its five digits cannot depend on the
 *             last bits of exp() — elsewhere (the denormal range, a value within 4e-13 of a rounding boundary) the field is left EMPTY and listed in
 *             `patches`: the caller inserts, at byte `offset` of the text, "%.5lg" of exp(value - maxLLK) * c / (sumSingle + sumDouble) computed with ITS
 *             libm (c = (1 - prior) / V for a singlet row, prior / V / (V-1) / (A-1) else; the record of barcode cells[out_cell] has the three scalars).
 *             cell_flag[i]: 0 = rows in the text at [cell_off[i], cell_off[i+1]); 1 = left to the caller as requested; 2 = left to the caller because
 *             an entry is not printable here (nan, inf, |v| >= 2^43) — both with an empty range: the caller's rows go in at cell_off[i].
 * dmx_demuxlet_run uses this for `write_pair` jobs; the result is byte-identical to the host formatter's (tests/test_gpu_pair_text.py). */
typedef struct { int32_t a, b, n, reserved; double llk_ab, llk_ba; } dmx_pair_override;   /* n < 0: none */
typedef struct { int64_t offset; double value; int32_t out_cell, singlet; } dmx_pair_patch;
typedef struct {
  int32_t n_out;
  const int32_t* cells;             /* [n_out] */
  const char* const* barcodes;      /* [n_out] */
  const char* const* sample_ids;    /* [n_samples] */
  const uint8_t* host_rows;         /* [n_out] or NULL */
  const dmx_pair_override* ovr;     /* [n_out] or NULL */
} dmx_pair_request;
typedef struct dmx_pair_text dmx_pair_text;   /* owns the device text; host copies of the small arrays */
typedef struct {
  int64_t n_bytes; int32_t n_out, n_patches;
  const int64_t* cell_off;          /* [n_out + 1] */
  const uint8_t* cell_flag;         /* [n_out] */
  const dmx_pair_patch* patches;    /* [n_patches], ascending offset */
  double format_ms;                 /* HIP-event time of the three kernels */
} dmx_pair_text_info;
int  dmx_engine_format_pair(dmx_engine*, const dmx_pair_request*, dmx_pair_text** out);   /* needs run_doublet's results; synchronises */
int  dmx_pair_text_get_info(const dmx_pair_text*, dmx_pair_text_info* out);
int  dmx_pair_text_read(dmx_pair_text*, int64_t offset, int64_t n_bytes, void* dst);     /* device -> host copy of a piece of the text */
void dmx_pair_text_free(dmx_pair_text*);

/* Device views for zero-copy hand-off (torch tensors over them, RCCL gather of the per-cell records). */
typedef struct {
  double* llks;  double* llk0s;  double* llksAB;  double* llks00;  dmx_cell_summary* summary;  double* gp0s;
  double* sing;                /* [B][V] */
} dmx_device_view;
int dmx_engine_device_view(dmx_engine*, dmx_device_view* out);

/* HIP-event timing of the last launch of each kernel on the engine's stream, in milliseconds (0 when not run). */
typedef struct { float gp0_ms, singlet_ms, doublet_ms, reduce_ms; } dmx_kernel_times;
int dmx_engine_last_kernel_times(dmx_engine*, dmx_kernel_times* out);
/* The same, averaged over the launches since the last reset (at most the last 16 of each kernel): what a benchmark divides a
 * kernel's algorithmic bytes by.  doublet_ms is K2 alone, reduce_ms K3 alone, certify_ms K3b alone (0 when it does not run).
 * Synchronises the engine's stream.  reset != 0 forgets the launches seen so far; out may be NULL (reset only). */
typedef struct { double singlet_ms, doublet_ms, reduce_ms, certify_ms; int32_t n_singlet, n_doublet; } dmx_kernel_time_means;
int dmx_engine_mean_kernel_times(dmx_engine*, int32_t reset, dmx_kernel_time_means* out);
/* (ABI 7) Which kernels have run on the staged pileup, by the demangled names rocprofv3 prints ("k_doublet_a2<256, 4, 4, true, false, 32>";
 * empty = not run).  Each entry is what the LAST call that launches that kernel picked: run_singlet replaces `singlet`, run_doublet replaces
 * `doublet` and `certify` (empty when that run had no K3b), dmx_engine_run all three; staging a pileup clears them.  Where K1 ran: 0 = a launch of its own (run_singlet), 1 = beside K2 on the low-priority stream (dmx_engine_run), 2 = beside
 * K3 + K3b (dmx_engine_run when K2 leaves K1 no room).  A benchmark pairs its committed counter files with these names instead of guessing. */
typedef struct { char singlet[96], doublet[96], certify[96]; int32_t k1_placement; int32_t reserved[3]; } dmx_kernel_names;
int dmx_engine_kernel_names(dmx_engine*, dmx_kernel_names* out);
/* Algorithmic HBM bytes one launch of each kernel must move for the staged problem. */
typedef struct { double singlet_bytes, doublet_bytes, reduce_bytes; } dmx_kernel_bytes;
int dmx_engine_algorithmic_bytes(dmx_engine*, dmx_kernel_bytes* out);

/* ------------------------------------------------------------------------------------------------------------------
 * a6,a10..a14  finaliser and writers — replaces cmd_cram_demuxlet.cpp:465-527 (.single), :713-875 (.sing2/.pair/.best).
 *     Rows come out in ascending byte-wise barcode order (std::map<std::string,int32_t>, :472,:576); cells failing
 *     --min-total/--min-uniq/--min-snp are skipped (:480,:581); cells without any covered SNP get .single rows only
 *     (:592).  All arrays are HOST memory. */
typedef struct {
  int32_t n_cells, n_samples, n_alpha;
  const double* alpha;
  double  doublet_prior;
  int32_t min_total, min_uniq, min_snp;
  int32_t write_pair;
  const char* const* barcodes;     /* [n_cells] by cell id */
  const char* const* sample_ids;   /* [n_samples] */
  const int32_t *rd_totl, *rd_pass, *rd_uniq, *n_snp;   /* [n_cells] */
  const double* llks;              /* [n_cells][V]       (for .single) */
  const double* llk0s;             /* [n_cells] */
  const double* llksAB;            /* [n_cells][V][V][A] (for .sing2/.pair/.best) */
  const double* llks00;            /* [n_cells][A] */
  /* Optional tie arbiter: with the host pileup and genotype matrix present, grid entries within
   * tie_tol of a decision (top-2 singlets, best doublet) are re-evaluated on the host in the reference's exact
   * operation order with the host libm before the strict-< scans run, so DBL-a-b vs DBL-b-a follows the reference. */
  const dmx_pileup* tie_pileup;    /* NULL = no arbiter */
  const float*  tie_g;             /* [n_snps][V][3] */
  double  tie_tol;                 /* 0 = default 1e-7 */
} dmx_final_input;                 /* fixed layout since ABI 5: passed by pointer without a size member, it never grows */

int dmx_write_single(const dmx_final_input*, const char* path);                    /* <out>.single */
int dmx_write_doublet(const dmx_final_input*, const char* out_prefix);            /* <out>.sing2, <out>.best, [<out>.pair] */
/* The same <out>.sing2 and <out>.best from the per-cell records of the device-side reduction (dmx_engine_get_sing,
 * dmx_engine_get_doublet's summary and llks00) instead of the full grid: what a multi-GPU run gathers to rank 0.
 * in->llksAB is ignored; in->write_pair must be 0 (the .pair rows need the grid).  With in->tie_pileup the order of
 * the two samples of an alpha = 0.5 best doublet is arbitrated exactly.  Barcodes K3 flagged as near-ties
 * (another sample pair, another alpha or another singlet within 1e-7 of a decision: duplicate samples, a handful of covered SNPs)
 * need more than their record: pass their grids to dmx_write_doublet_summary_grids; without a grid but with in->tie_pileup the barcode's
 * WHOLE grid is re-evaluated on the host in the reference's operation order (exact, pairs x V x V x A host log() calls for that barcode;
 * refused with DMX_ERR_ARG beyond 2e9 such calls in one job — pass the grids); with neither, the device's own choice among the near-tied
 * candidates is printed (each within 1e-7 of the reference's best).  A barcode with DMX_CELL_NEAR_RULE has the entries of the BEST rule
 * re-evaluated when in->tie_pileup is there (no grid needed). */
int dmx_write_doublet_summary(const dmx_final_input*, const double* sing, const dmx_cell_summary* summary, const char* out_prefix);
/* (ABI 7) The same with cell_grid[n_cells]: pointers to single cells' llksAB[V][V][A] (NULL entries = none; a NULL array = none =
 * dmx_write_doublet_summary).  A barcode whose K3 record carries DMX_CELL_NEAR_DOUBLET / _NEAR_SINGLET is decided from its grid
 * (dmx_engine_get_cell_grids fetches exactly those), with the tie arbiter re-evaluating the contenders as dmx_write_doublet does. */
int dmx_write_doublet_summary_grids(const dmx_final_input*, const double* sing, const dmx_cell_summary* summary,
                                    const double* const* cell_grid, const char* out_prefix);
/* For consumers of the K3 records themselves: turn every DMX_CELL_ORDER_RESOLVABLE record among summary[0..n) into a certified one
 * by asking this host's libm for the one or two log() values the device left open (the writers above do the same internally).
 * Returns the number of records that stay unresolved (>= 0), or a negative dmx_status. */
int dmx_resolve_tie_order(dmx_cell_summary* summary, int64_t n);

/* Diagnostics: evaluate the device's log() replacement (dmx_log, csrc/dmx_log.hpp) on n host doubles. Used by the tests
 * to show the device function performs exactly the IEEE operation sequence whose accuracy is measured on the host. */
int dmx_debug_device_log(const double* x, double* y, int64_t n, int32_t device);
int dmx_debug_device_log2(const double* x, double* y, int64_t n, int32_t device);   /* dmx_log2: the doublet kernels' log (256 bins, ABI 6) */
int dmx_debug_device_log2_lite32(const double* x, double* y, int64_t n, int32_t device); /* (ABI 8, appended in round 6) FAST's second phase-2 log: split 32-bin table (conflict-free LDS reads), degree-5 near-minimax tail — 8 FP64 instructions, |error| <= 1.2e-15 absolute (csrc/dmx_log.hpp) */
int dmx_debug_device_log2_lite(const double* x, double* y, int64_t n, int32_t device);   /* (ABI 7) DMX_MODE_FAST's phase-2 log: dmx_log2's table, series cut after r^4/4 — 6 FP64 instructions, |error| <= 6e-15 absolute, unbiased (csrc/dmx_log.hpp) */
/* Diagnostics: the device's log() ceiling, measured by a register-resident microkernel (no memory traffic): which = 0 the
 * kernels' dmx_log, which = 1 ocml's log().  bench.py reports both next to the kernels' achieved log rate. */
int dmx_debug_log_rate(int32_t which, int32_t iters, int32_t device, double* logs_per_second);   /* which: 0 dmx_log, 1 ocml log(), 2 dmx_log2 */
/* Diagnostics: q[i] = a[i] / b[i] through the kernels' shared-reciprocal division (must equal IEEE division bit for bit
 * for 2^-700 < a,b < 2^700). */
int dmx_debug_device_div(const double* a, const double* b, double* q, int64_t n, int32_t device);

/* Diagnostics (host evaluation of the code the device runs): the double-double logarithm behind the tie-order certificate
 * (csrc/dmx_log.hpp, DESIGN.md "Ties").  hi[i] + lo[i] = log(x[i]) to ~2^-75; [t_lo[i], t_hi[i]] = the doubles a libm with
 * < 0.53 ulp error can return for log(x[i]) (one double, or the two neighbours of a rounding midpoint). */
int dmx_debug_log_dd(const double* x, double* hi, double* lo, double* t_lo, double* t_hi, int64_t n);

/* ------------------------------------------------------------------------------------------------------------------
 * One call for the whole of cmd_cram_demuxlet.cpp:390-881: store + genotype matrix + options in, four files out. */
typedef struct {
  dmx_store*   store;
  const float* g;                  /* [n_snps][V][3], HOST */
  int32_t      n_samples;
  const char* const* sample_ids;
  int32_t      n_alpha;  const double* alpha;
  double       doublet_prior;
  int32_t      min_total, min_uniq, min_snp, write_pair;
  const char*  out_prefix;
  int32_t      device;
  int32_t      arbiter;            /* 1 = run the tie arbiter (default in the CLI) */
  int32_t      n_gpus;             /* 0 or 1: one GPU (`device`).  N > 1: devices (device + i) mod (visible devices).  The byte-wise
                                      sorted barcodes are cut into contiguous ranges of equal work — at least one per GPU, four for jobs
                                      worth overlapping, more when a range's doublet grid would exceed the byte budget (4 GiB, or a sixth of
                                      the free device memory) — which go through the GPUs in waves.  With several waves every GPU runs two
                                      engines (own HIP stream each) that alternate: the H2D of wave w + 2 overlaps the kernels of wave w + 1
                                      while the host arbitrates, formats and appends the rows of wave w.  Barcodes are independent
                                      (cmd_cram_demuxlet.cpp:576): no collective. */
  int32_t      mode;               /* DMX_MODE_STRICT (0: the default of this struct, of the Python binding and of the `demuxlet`
                                      binary) or DMX_MODE_FAST (`demuxlet --fast`) */
  /* Optional (ABI 2): a pileup that is already frozen (sparse or dense layout) instead of `store` (then NULL), with its barcodes by
   * cell id — what a caller that builds the CSR itself hands over (tools/e2e_bench.cpp, the benchmarks).  memory = DMX_MEM_HOST, or
   * (ABI 6) DMX_MEM_DEVICE: the five arrays live in the HBM of `device` (n_gpus <= 1), the rd_* counters stay host memory.  Nothing is
   * then sliced or copied on the host: a range of consecutive cells is a view of the caller's arrays, any other range is gathered
   * on the device, and the barcodes the tie arbiter walks (near-tie flags, an open tie-order certificate) have their pieces fetched. */
  const dmx_pileup*  pileup;
  const char* const* barcodes;
  /* Optional (ABI 2): wall-clock seconds of the stages of this call, written on return (NULL = not wanted). */
  struct dmx_job_timing* timing;
} dmx_job;
typedef struct dmx_job_timing {
  double freeze_s;                 /* dmx_store_freeze */
  double setup_s;                  /* engine creation, genotype upload, class detection */
  double stage_s;                  /* slicing the CSR into ranges + H2D (host time, all ranges) */
  double wait_s;                   /* host blocked on the GPUs (kernels not hidden behind host work) + D2H of the results */
  double write_s;                  /* tie arbiter + row formatting + file writes (all ranges) */
  double total_s;
  double kernel_ms;                /* sum over ranges of the K1 + K2 + K3 HIP-event times (device time, all engines) */
  int32_t n_ranges, n_engines, n_cells_grid_fetched;
  int32_t n_cells_single;          /* barcodes that passed --min-total / --min-uniq / --min-snp: the droplets of cmd_cram_demuxlet.cpp:480-524 (ABI 6; was `reserved`) */
} dmx_job_timing;
int dmx_demuxlet_run(const dmx_job*);
/* Creates the HIP contexts of devices (device + i) mod (visible devices), i < n_gpus (n_gpus <= 0: one), and returns.  Optional: the first
 * job of a process otherwise pays this (0.12-0.13 s) inside dmx_demuxlet_run; a caller that still has host work to do (the `demuxlet`
 * binary: the BAM x VCF scan, cmd_cram_demuxlet.cpp:195-338) calls it on a thread of its own first.  Thread-safe. */
int dmx_device_warm_up(int32_t device, int32_t n_gpus);

#ifdef __cplusplus
}
#endif
#endif
