// End-to-end timing of the C-ABI path a maintainer would call from cmd_cram_demuxlet.cpp:
//   dmx_store_add_* (BAM-ordered synthetic observations)  ->  dmx_demuxlet_run  ->  four text files.
// Build (repo root):  hipcc -O2 -std=c++17 -Iinclude tools/e2e_bench.cpp -o tools/e2e_bench -Ldemuxlet_amd -ldmx -Wl,-rpath,$PWD/demuxlet_amd
// Run on a GPU box:   tools/e2e_bench <barcodes> <snps> <samples> <density> <write_pair 0|1> [n_gpus]
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "dmx.h"

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static inline uint64_t rnd() { uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
static inline double unif() { return (rnd() >> 11) * (1.0 / 9007199254740992.0); }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
  if (argc < 6) { fprintf(stderr, "usage: %s barcodes snps samples density write_pair [n_gpus]\n", argv[0]); return 2; }
  const int B = atoi(argv[1]), S = atoi(argv[2]), V = atoi(argv[3]);
  const double delta = atof(argv[4]);
  const int write_pair = atoi(argv[5]), n_gpus = argc > 6 ? atoi(argv[6]) : 1;
  // genotypes: GT with error 0.01 through the library's own transform
  std::vector<float> G((size_t)S * V * 3);
  std::vector<int8_t> dos((size_t)S * V);
  std::vector<int32_t> gt((size_t)V * 2);
  for (int s = 0; s < S; ++s) {
    const double af = 0.05 + 0.9 * unif();
    for (int j = 0; j < V; ++j) { gt[2 * j] = unif() < af; gt[2 * j + 1] = unif() < af; dos[(size_t)s * V + j] = (int8_t)(gt[2 * j] + gt[2 * j + 1]); }
    if (dmx_geno_from_gt(gt.data(), V, 0.01, &G[(size_t)s * V * 3]) != DMX_OK) { fprintf(stderr, "%s\n", dmx_last_error()); return 1; }
  }
  std::vector<std::string> bc((size_t)B), sm((size_t)V);
  for (int c = 0; c < B; ++c) { char b[32]; uint64_t x = rnd(); for (int i = 0; i < 16; ++i) { b[i] = "ACGT"[x & 3]; x >>= 2; } b[16] = 0; bc[c] = std::string(b) + "-1"; }
  for (int j = 0; j < V; ++j) sm[j] = "SM" + std::to_string(j);
  std::vector<const char*> smp((size_t)V);
  for (int j = 0; j < V; ++j) smp[j] = sm[j].c_str();

  const bool dry = getenv("E2E_DRY") != nullptr;      // generator only (its cost is part of the "store" time below)
  dmx_store* st = dmx_store_new();
  for (int s = 0; s < S; ++s) dmx_store_add_snp(st);
  // BAM order = SNP-major: for every SNP the reads of the cells that cover it
  double t0 = now();
  long n_obs = 0;
  char umi[16];
  for (int s = 0; s < S; ++s)
    for (int c = 0; c < B; ++c) {
      if (unif() >= delta) continue;
      const int32_t ib = dry ? c : dmx_store_add_cell(st, bc[c].c_str());
      const int src = c % V;
      int nr = 1; while (unif() < 0.2 && nr < 6) ++nr;
      for (int r = 0; r < nr; ++r) {
        if (!dry) dmx_store_count_read(st, ib);
        const int bq = 13 + (int)(rnd() % 28);
        const bool alt = unif() < 0.5 * dos[(size_t)s * V + src];
        snprintf(umi, sizeof umi, "U%07llu", (unsigned long long)(rnd() % 10000000ull));
        if (!dry) dmx_store_add_read(st, s, ib, umi, alt ? 1 : 0, bq); else rng_state += (uint64_t)umi[3];
        ++n_obs;
      }
    }
  double t1 = now();
  if (dry) { fprintf(stderr, "generator alone: %ld observations in %.2f s\n", n_obs, t1 - t0); return 0; }
  fprintf(stderr, "store: %ld observations in %.2f s = %.3e add_read/s (%d cells, %d SNPs)\n", n_obs, t1 - t0, n_obs / (t1 - t0), dmx_store_n_cells(st), S);
  dmx_pileup pl;
  dmx_store_freeze(st, &pl);
  double t2 = now();
  fprintf(stderr, "freeze: %.2f s (%lld pairs, %lld reads)\n", t2 - t1, (long long)pl.n_pairs, (long long)pl.n_reads);
  const double alpha[2] = {0.0, 0.5};
  dmx_job job;
  memset(&job, 0, sizeof job);
  job.store = st; job.g = G.data(); job.n_samples = V; job.sample_ids = smp.data(); job.n_alpha = 2; job.alpha = alpha; job.doublet_prior = 0.5;
  job.write_pair = write_pair; job.out_prefix = "/tmp/e2e_bench_out"; job.device = 0; job.arbiter = 1; job.n_gpus = n_gpus;
  if (dmx_demuxlet_run(&job) != DMX_OK) { fprintf(stderr, "%s\n", dmx_last_error()); return 1; }
  double t3 = now();
  const double pe = (double)pl.n_pairs * V * V * 2;
  fprintf(stderr, "dmx_demuxlet_run: %.2f s (engine create + staging + kernels + four files) = %.3e pair-evals/s end to end, %.3e triples/s\n",
          t3 - t2, pe / (t3 - t2), (double)pl.n_pairs * V / (t3 - t2));
  dmx_store_free(st);
  return 0;
}
