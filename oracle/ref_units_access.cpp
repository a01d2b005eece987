// TEST INFRASTRUCTURE. extern "C" accessors (ours) over two reference translation units that compile with plain
// g++ from /root/reference and need no stand-in of any kind:
//   PhredHelper.cpp  (global phredConv, ctor PhredHelper.cpp:24-40)            -> pins SURVEY §8a row a2
//   sc_drop_seq.cpp  (sc_dropseq_lib_t::add_snp/add_cell/add_read, :3-77)      -> pins SURVEY §8a row a1
// Used by tests/golden/make_golden.py to generate the fixture tests/golden/ref_units.npz (checked by tests/test_host_units.py and
// tests/test_oracle_golden.py).
#include <cstring>
#include <string>
#include <vector>
#include "sc_drop_seq.h"
#include "PhredHelper.h"

extern "C" {

void ref_phred_tables(double* mat256, double* err256) {
  for (int i = 0; i < 256; ++i) { mat256[i] = phredConv.phred2Mat[i]; err256[i] = phredConv.phred2Err[i]; }
}
double ref_phred_prob(int q) { return phredConv.toProb((uint32_t)q); }

void* ref_scl_new() { return new sc_dropseq_lib_t; }
void ref_scl_free(void* p) { delete (sc_dropseq_lib_t*)p; }
int ref_scl_add_snp(void* p) { static double dummy[3] = {0, 0, 0}; return ((sc_dropseq_lib_t*)p)->add_snp(0, 0, 'A', 'C', 0.5, dummy); }
int ref_scl_add_cell(void* p, const char* bc) { return ((sc_dropseq_lib_t*)p)->add_cell(bc); }
int ref_scl_add_read(void* p, int snp, int cell, const char* umi, int al, int bq) {
  return ((sc_dropseq_lib_t*)p)->add_read(snp, cell, umi, (char)al, (char)bq) ? 1 : 0;
}
int ref_scl_ncells(void* p) { return ((sc_dropseq_lib_t*)p)->nbcs; }
void ref_scl_counters(void* p, int cell, int* pass, int* uniq, int* nsnp) {
  sc_dropseq_lib_t* s = (sc_dropseq_lib_t*)p;
  *pass = s->cell_pass_reads[cell]; *uniq = s->cell_uniq_reads[cell]; *nsnp = (int)s->cell_umis[cell].size();
}
// Flatten cell -> (snp ascending) -> (umi ascending) -> packed word, the iteration order of cmd_cram_demuxlet.cpp:595,:600
long ref_scl_flatten_cell(void* p, int cell, int* snps, int* nper, unsigned* words, long cap_pairs, long cap_words) {
  sc_dropseq_lib_t* s = (sc_dropseq_lib_t*)p;
  long np = 0, nw = 0;
  for (std::map<int32_t, sc_snp_droplet_t*>::iterator it = s->cell_umis[cell].begin(); it != s->cell_umis[cell].end(); ++it) {
    if (np >= cap_pairs) return -1;
    snps[np] = it->first; nper[np] = (int)it->second->size(); ++np;
    for (sc_snp_droplet_it_t it2 = it->second->begin(); it2 != it->second->end(); ++it2) {
      if (nw >= cap_words) return -1;
      words[nw++] = it2->second;
    }
  }
  return np;
}

}
