// Internal helpers shared by the host (dmx_host.cpp) and device (dmx_engine.hip) halves of libdmx.
#pragma once
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>
#include "dmx.h"

namespace dmx {

int set_error(int code, const char* fmt, ...);   // records the message for dmx_last_error(), returns code

// LUT in the form the kernels consume: for bq in [0,128): mat, err/3.0, 0.5-err/3.0 — the three per-read factors of
// cmd_cram_demuxlet.cpp:437-439 / :606-607, formed on the host with the same IEEE operations the reference performs
// per read (x/3.0 and 0.5-x are correctly rounded, so hoisting them is exact).
struct ReadLut { double mat[128], e3[128], het[128]; };
void build_read_lut(const double mat[256], const double err[256], ReadLut* out);

// Singlet genotype likelihoods of a pair as a function of its FIRST read byte b = (allele<<7)|bq (cmd_cram_demuxlet.cpp
// :427-452 evaluated on the host with the reference's IEEE operations):
//   first[b]  = GL after that one read, renormalised to sum 1            (:437-443) — start of the loop for deeper pairs
//   final1[b] = first[b] + 1e-6, renormalised                            (:446-452) — the whole answer for 1-read pairs
//   final1[256] = the answer for a pair without usable reads: (1,1,1) + 1e-6, renormalised
struct SingletTables { double first[256][3]; double final1[257][3]; };
void build_singlet_tables(const ReadLut& lut, SingletTables* out);

// The same for the first TWO read bytes (both with base quality < 64, code c = (allele << 6) | bq, index c0 * 128 + c1;
// 4 doubles per entry, the 4th is padding):
//   second[c0][c1] = GL after two reads, renormalised after each (:437-443 twice) — start of the loop for pairs of >= 3 reads
//   final2[c0][c1] = second + 1e-6, renormalised (:446-452)               — the whole answer for 2-read pairs
struct PairTables { double second[128 * 128][4]; double final2[128 * 128][4]; };
void build_pair_tables(const ReadLut& lut, const SingletTables& st, PairTables* out);

// And for the first THREE read bytes, all with base quality < 48 (the CLI caps at --cap-BQ 40): code c = allele * 48 + bq,
// index (c0 * 96 + c1) * 96 + c2, 4 doubles per entry:
//   third[i]  = GL after three reads, renormalised after each — start of the loop for pairs of >= 4 reads
//   final3[i] = third + 1e-6, renormalised                    — the whole answer for 3-read pairs
// 2 x 28 MB; built once per process for a given phred table (build_triple_tables caches its last result).
constexpr int kTripleCodes = 96, kTripleBq = 48;
struct TripleTables { std::vector<double> third, final3; };       // [96*96*96][4] each
const TripleTables& build_triple_tables(const ReadLut& lut, const PairTables& pt);

// Host-side exact re-evaluation of selected doublet-grid entries of ONE cell, in the reference's operation order with
// the host libm (tie arbiter; cmd_cram_demuxlet.cpp:595-684 restricted to the requested (j,k,n)).
struct GridReq { int32_t j, k, n; double value; };
// pG[A][3][3] of a pair (cmd_cram_demuxlet.cpp:597-663, every IEEE operation of the reference) as a function of its read bytes,
// tabulated for pairs of 0, 1 and 2 (base quality < 64) reads: what exact_grid_entries needs per pair without the per-read
// divisions.  97 % of the pairs of a 10x-like pileup have at most two reads.
struct MixTables {
  int32_t A = 0;
  std::vector<double> none;     // [A*9]
  std::vector<double> one;      // [256][A*9]        index = read byte (allele << 7) | bq
  std::vector<double> two;      // [128*128][A*9]    index = c0 * 128 + c1, c = (allele << 6) | bq, bq < 64
};
void build_mix_tables(const ReadLut& lut, int32_t A, const double* alpha, MixTables* out);
void exact_grid_entries(const dmx_pileup& pl, const float* g, int32_t V, int32_t A, const double* alpha,
                        const ReadLut& lut, const MixTables* mix, int32_t cell, std::vector<GridReq>& reqs);

// Where the doublet-stage writers take a cell's numbers from: the whole grid, a per-cell grid pointer, or the K3 records.
struct DoubletSource {
  const double* grid_all = nullptr;              // [n_cells][V][V][A]
  const double* const* cell_grid = nullptr;      // [n_cells] -> [V][V][A] or NULL
  const double* sing = nullptr;                  // [n_cells][V]
  const dmx_cell_summary* summary = nullptr;     // [n_cells]
  const int32_t* tie_cell = nullptr;             // [n_cells] -> the cell's index in in->tie_pileup (NULL: the same index; -1: not staged there)
  // (round 6) write_pair with the `.pair` rows formatted on the device (dmx_engine_format_pair): the writer does not touch the .pair file.  The
  // barcodes that come with a grid here (cell_grid[c] != NULL: the ones left to the host formatter) have their rows stored in (*pair_rows)[c];
  // the caller splices them into the device's text.
  std::vector<std::string>* pair_rows = nullptr; // [n_cells]
};
std::vector<int32_t> output_cells(const dmx_final_input* in, bool need_snps);   // the cells that get rows, in output order (barcode order; :480,:581,:592)
void put_general(std::string& out, double v, int prec);                        // printf("%.<prec>lg")
// What the doublet-stage writers need of a barcode BEYOND its K3 record — ONE predicate for the code that stages (dmx_demuxlet_run's fetch, a
// multi-GPU rank choosing what rides along in the gather) and the code that writes (write_doublet_core); ADVICE r4: the two used to be maintained
// apart.  `sm` = the record after resolve_tie_order (a DMX_CELL_ORDER_RESOLVABLE record that this host's libm settles counts as certified).
//   kNeedGrid    a near-tie flag: which candidates sit within 1e-7 of a decision is not in the record
//   kNeedPileup  the arbiter re-evaluates entries of this barcode in the reference's operation order: the near-tie flags, an alpha = 0.5 best doublet
//                without a tie-order certificate, or a comparison of the BEST rule (cmd_cram_demuxlet.cpp:837,:844) with a margin below 1e-7
enum { kNeedGrid = 1, kNeedPileup = 2 };
inline int cell_needs(const dmx_cell_summary& sm, const double* alpha, int32_t A, bool arbiter) {
  if (!arbiter || sm.n_pairs <= 0) return 0;
  int need = 0;
  if (sm.flags & (DMX_CELL_NEAR_DOUBLET | DMX_CELL_NEAR_SINGLET)) need |= kNeedGrid | kNeedPileup;
  if (sm.n_best >= 0 && sm.n_best < A && alpha[sm.n_best] == 0.5 && !(sm.flags & DMX_CELL_ORDER_CERTIFIED)) need |= kNeedPileup;
  if (sm.flags & DMX_CELL_NEAR_RULE) need |= kNeedPileup;
  return need;
}
// a comparison of the BEST rule within tol of flipping: LLK12 > LLK1, LLK12 > LLK2, LLK12 > SNG.LLK1 + 2 (:837), SNG.LLK1 > SNG.LLK2 + 2 (:844).
// The same expression on the device (k_reduce) and on the host (the writers, on the values they are about to compare).
// LLK1 and LLK2 are singlet entries, so they are <= SNG.LLK1 and the first two comparisons follow from the third wherever the values are
// consistent: they are tested only where the third does not already rule the doublet out (a panel with duplicated samples has LLK12 == LLK1
// to the last bit in most barcodes; flagging those would stage their pileups for a comparison that decides nothing).
inline bool near_rule(double l12, double l1, double l2, double s1, double s2, double tol) {
  auto close = [tol](double a, double b) { const double d = a - b; return d < tol && d > -tol; };
  return close(l12, s1 + 2) || close(s1, s2 + 2) || ((close(l12, l1) || close(l12, l2)) && l12 > s1 + 2 - tol);
}
// dmx_engine_set_pileup for cells cells[0..nb) of a HOST pileup (NULL: all, in order), re-based while it streams to the device
// (a DEVICE pileup: host_po / host_ro = host copies of its offset arrays when the caller has them, else they are fetched)
int engine_set_pileup_cells(dmx_engine* e, const dmx_pileup* pl, const int32_t* cells, int32_t nb, const int64_t* host_po = nullptr, const int64_t* host_ro = nullptr);
// true when the host libm's log() stays inside dmx_log_bracket()'s brackets on a fixed sample of arguments (checked once)
bool libm_log_within_brackets();
bool resolve_tie_order(dmx_cell_summary* r);   // DMX_CELL_ORDER_RESOLVABLE -> certified, by the host libm's log()
int write_doublet_core(const dmx_final_input* in, const DoubletSource& src, const char* out_prefix, bool append, const char* who);

// The writers behind dmx_write_single / dmx_write_doublet with an append mode: dmx_demuxlet_run streams contiguous ranges
// of the sorted barcodes through the GPUs and appends each range's rows (append = true: no header, files opened "a").
int write_single_impl(const dmx_final_input* in, const char* path, bool append);
int write_doublet_impl(const dmx_final_input* in, const char* out_prefix, bool append);

}  // namespace dmx
