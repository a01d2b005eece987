// TEST INFRASTRUCTURE — never linked into, imported by, or executed from the product.
//
// Harness that runs the REFERENCE's own likelihood engine on a neutral "pileup spec" file.
//
// What is reference code here and what is ours:
//   * /root/reference/{sc_drop_seq,PhredHelper,Error}.cpp are compiled as they lie (no edits, no stand-ins).
//   * lines 390..881 of /root/reference/cmd_cram_demuxlet.cpp (the whole engine + finaliser, SURVEY.md §8a rows
//     a4..a14) are extracted at BUILD time by oracle/Makefile into a temp file outside the repo and #included below
//     verbatim (DMX_REF_SLICE).  One line `DMX_DUMP_CELL();` is inserted by the same recipe in front of the
//     "// normalize by max likelihood" comment (cmd_cram_demuxlet.cpp:712) so the raw per-cell doublet grid can be
//     written as binary doubles; it does not touch any arithmetic.
//   * ours: this file. It only provides (a) the text-output plumbing the slice calls (hts_open/hprintf/hts_close of
//     hts_utils.cpp:1013-1034 write plain text when mode is "w"; we route them to FILE*/vfprintf — htslib's
//     kvsprintf is vsnprintf for every format used by the slice), (b) a `vr` object with the two members the slice
//     touches (verbose, get_sample_id_at), (c) the locals of main() the slice expects, filled from the spec through
//     the reference's own add_snp/add_cell/add_read exactly like cmd_cram_demuxlet.cpp:180-185,239-325 does.
//   The full `demuxlet` binary is unbuildable in this image (htslib absent) — see DESIGN.md §Oracle.
//
// Spec format (text, written by tests/golden/make_golden.py and demuxlet_amd/specio.py):
//   DMXSPEC1
//   <nv> <nsnps> <ncells> <nalpha> <nevents>
//   <alpha_0> ... <alpha_{nalpha-1}>                      (C99 hex floats)
//   <doublet_prior> <min_total> <min_uniq> <min_snp> <write_pair>
//   SM <sample id>                                        x nv
//   G <3*nv hex floats = (double)(float) genotype probabilities of that SNP>   x nsnps
//   BC <barcode>                                          x ncells (informational; ids are assigned by add_cell order)
//   R <barcode> <snp|-1> <umi> <allele> <bq> <newread>    x nevents, in BAM order. snp=-1: read overlapping no SNP.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdarg>
#include <cmath>
#include <string>
#include <vector>
#include <map>
#include <set>
#include <algorithm>
#include <stdint.h>
#include <time.h>

#include "sc_drop_seq.h"
#include "PhredHelper.h"
#include "Error.h"

struct htsFile { FILE* fp; };
static htsFile* hts_open(const char* fn, const char* mode) {
  FILE* f = fopen(fn, mode);
  if (!f) return NULL;
  htsFile* h = new htsFile; h->fp = f; return h;
}
static int hts_close(htsFile* h) { int r = fclose(h->fp); delete h; return r; }
static void hprintf(htsFile* h, const char* msg, ...) {
  va_list ap; va_start(ap, msg); vfprintf(h->fp, msg, ap); va_end(ap);
}

struct vr_shim_t {
  int32_t verbose;
  std::vector<std::string> ids;
  const char* get_sample_id_at(int32_t i) { return ids[i].c_str(); }
};

static FILE* g_dump_ab = NULL;   // raw llksAB per processed cell
static FILE* g_dump_00 = NULL;   // raw llks00 per processed cell
static FILE* g_dump_id = NULL;   // cell id per processed cell (int32)
#define DMX_DUMP_CELL() do { \
    fwrite(&i, sizeof(int32_t), 1, g_dump_id); \
    fwrite(llksAB, sizeof(double), (size_t)n1*nv*nAlpha, g_dump_ab); \
    fwrite(llks00, sizeof(double), (size_t)nAlpha, g_dump_00); } while(0)

static void die(const char* m) { fprintf(stderr, "ref_slice_harness: %s\n", m); exit(2); }

int main(int argc, char** argv) {
  if (argc < 3) die("usage: ref_slice_harness <spec> <outprefix> [--no-raw]");
  bool raw = !(argc > 3 && strcmp(argv[3], "--no-raw") == 0);
  FILE* fs = fopen(argv[1], "r");
  if (!fs) die("cannot open spec");
  std::string outPrefix(argv[2]);
  char magic[64];
  if (fscanf(fs, "%63s", magic) != 1 || strcmp(magic, "DMXSPEC1")) die("bad magic");
  int32_t nv, nsnps_spec, ncells_spec, nAlpha_spec; long nevents;
  if (fscanf(fs, "%d %d %d %d %ld", &nv, &nsnps_spec, &ncells_spec, &nAlpha_spec, &nevents) != 5) die("bad header");
  std::vector<double> gridAlpha(nAlpha_spec);
  for (int a = 0; a < nAlpha_spec; ++a) if (fscanf(fs, "%la", &gridAlpha[a]) != 1) die("bad alpha");
  double doublet_prior; int32_t minTotalReads, minUniqReads, minCoveredSNPs, wp;
  if (fscanf(fs, "%la %d %d %d %d", &doublet_prior, &minTotalReads, &minUniqReads, &minCoveredSNPs, &wp) != 5) die("bad params");
  bool writePair = wp != 0;

  vr_shim_t vr; vr.verbose = 10000;
  char tag[16], buf[4096];
  for (int i = 0; i < nv; ++i) { if (fscanf(fs, "%15s %4095s", tag, buf) != 2 || strcmp(tag, "SM")) die("bad SM"); vr.ids.push_back(buf); }

  sc_dropseq_lib_t scl;
  std::vector<int32_t> snpids;
  for (int s = 0; s < nsnps_spec; ++s) {
    if (fscanf(fs, "%15s", tag) != 1 || strcmp(tag, "G")) die("bad G");
    double* g = new double[nv * 3];                       // cmd_cram_demuxlet.cpp:181,227
    for (int j = 0; j < nv * 3; ++j) { double d; if (fscanf(fs, "%la", &d) != 1) die("bad g"); g[j] = (double)(float)d; }
    snpids.push_back(scl.add_snp(0, s, 'A', 'C', 0.5, g));  // :184,:231 (rid/pos/ref/alt/af unused by the slice)
  }
  for (int c = 0; c < ncells_spec; ++c) { if (fscanf(fs, "%15s %4095s", tag, buf) != 2 || strcmp(tag, "BC")) die("bad BC"); }
  char umi[4096];
  for (long e = 0; e < nevents; ++e) {
    int snp, al, bq, nr;
    if (fscanf(fs, "%15s %4095s %d %4095s %d %d %d", tag, buf, &snp, umi, &al, &bq, &nr) != 7 || strcmp(tag, "R")) die("bad R");
    int32_t ibcd = scl.add_cell(buf);                       // :262
    if (nr) ++scl.cell_totl_reads[ibcd];                    // :295
    if (snp >= 0) scl.add_read(snpids[snp], ibcd, umi, (char)al, (char)bq);   // :325
  }
  fclose(fs);

  int32_t nAlpha = (int32_t)gridAlpha.size();               // :345
  double* gps = NULL;                                       // :181 (re-used as a scratch pointer by the slice)

  if (raw) {
    g_dump_ab = fopen((outPrefix + ".raw.llksAB").c_str(), "wb");
    g_dump_00 = fopen((outPrefix + ".raw.llks00").c_str(), "wb");
    g_dump_id = fopen((outPrefix + ".raw.cellid").c_str(), "wb");
  } else {
    g_dump_ab = fopen("/dev/null", "wb"); g_dump_00 = fopen("/dev/null", "wb"); g_dump_id = fopen("/dev/null", "wb");
  }

  // wall-clock of the reference's lines alone (bench.py's cpu_baseline "reference_slice" leg reads this line)
  struct timespec ts0, ts1;
  clock_gettime(CLOCK_MONOTONIC, &ts0);
#include DMX_REF_SLICE
  clock_gettime(CLOCK_MONOTONIC, &ts1);
  fprintf(stderr, "SLICE_SECONDS %.6f\n", (double)(ts1.tv_sec - ts0.tv_sec) + 1e-9 * (double)(ts1.tv_nsec - ts0.tv_nsec));

  if (raw) {
    FILE* f = fopen((outPrefix + ".raw.llks").c_str(), "wb");
    fwrite(llks.data(), sizeof(double), llks.size(), f); fclose(f);
    f = fopen((outPrefix + ".raw.llk0s").c_str(), "wb");
    fwrite(llk0s.data(), sizeof(double), llk0s.size(), f); fclose(f);
    // barcode -> id map in id order, so the test side can align raw arrays
    f = fopen((outPrefix + ".raw.barcodes").c_str(), "w");
    std::vector<std::string> byid(scl.nbcs);
    for (std::map<std::string,int32_t>::iterator it = scl.bc_map.begin(); it != scl.bc_map.end(); ++it) byid[it->second] = it->first;
    for (int32_t c = 0; c < scl.nbcs; ++c) fprintf(f, "%s\t%d\t%d\t%d\t%d\n", byid[c].c_str(), scl.cell_totl_reads[c], scl.cell_pass_reads[c], scl.cell_uniq_reads[c], (int32_t)scl.cell_umis[c].size());
    fclose(f);
  }
  fclose(g_dump_ab); fclose(g_dump_00); fclose(g_dump_id);
  return 0;
}
