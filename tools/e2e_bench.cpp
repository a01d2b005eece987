// End-to-end timing of the C-ABI path a maintainer would call from cmd_cram_demuxlet.cpp, stage by stage:
//   mode "store":   dmx_store_add_* (BAM-ordered synthetic observations) -> dmx_store_freeze -> dmx_demuxlet_run -> four files
//   mode "pileup":  a frozen synthetic pileup (dmx_job.pileup; BASELINE-size jobs without the minutes of single-threaded
//                   add_read calls the reference's scan loop implies) -> dmx_demuxlet_run -> four files
// Build (repo root):  hipcc -O2 -std=c++17 -Iinclude tools/e2e_bench.cpp -o tools/e2e_bench -Ldemuxlet_amd -ldmx -Wl,-rpath,$PWD/demuxlet_amd -lpthread
// Run on a GPU box:   tools/e2e_bench <store|pileup> <barcodes> <snps> <samples> <density> <rbar> <GT|GP> <write_pair 0|1> <fast 0|1> [n_gpus] [arbiter 0|1]
// Prints one JSON line with the stage seconds of dmx_job_timing (include/dmx.h).
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#include "dmx.h"

struct Rng {
  uint64_t s;
  explicit Rng(uint64_t seed) : s(seed) {}
  uint64_t next() { uint64_t z = (s += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
  double unif() { return (next() >> 11) * (1.0 / 9007199254740992.0); }
};
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
  if (argc < 10) { fprintf(stderr, "usage: %s store|pileup barcodes snps samples density rbar GT|GP write_pair fast [n_gpus] [arbiter]\n", argv[0]); return 2; }
  const bool from_store = !strcmp(argv[1], "store");
  const int B = atoi(argv[2]), S = atoi(argv[3]), V = atoi(argv[4]);
  const double delta = atof(argv[5]), rbar = atof(argv[6]);
  const bool soft = !strcmp(argv[7], "GP");
  const int write_pair = atoi(argv[8]), fast = atoi(argv[9]), n_gpus = argc > 10 ? atoi(argv[10]) : 1, arbiter = argc > 11 ? atoi(argv[11]) : 1;
  Rng rg(0xD3A0);
  // genotypes through the library's own transforms (GT with error 0.01, or a softened one-hot GP)
  std::vector<float> G((size_t)S * V * 3);
  std::vector<int8_t> dos((size_t)S * V);
  {
    std::vector<int32_t> gt((size_t)V * 2);
    std::vector<float> gp((size_t)V * 3);
    for (int s = 0; s < S; ++s) {
      const double af = 0.05 + 0.9 * rg.unif();
      for (int j = 0; j < V; ++j) { gt[2 * j] = rg.unif() < af; gt[2 * j + 1] = rg.unif() < af; dos[(size_t)s * V + j] = (int8_t)(gt[2 * j] + gt[2 * j + 1]); }
      int rc;
      if (soft) {
        for (int j = 0; j < V; ++j) { const double e = 0.02 + 0.2 * rg.unif(); for (int l = 0; l < 3; ++l) gp[3 * j + l] = (float)(l == dos[(size_t)s * V + j] ? 1.0 - e : e / 2); }
        rc = dmx_geno_from_gp(gp.data(), V, 0.01, &G[(size_t)s * V * 3]);
      } else rc = dmx_geno_from_gt(gt.data(), V, 0.01, &G[(size_t)s * V * 3]);
      if (rc != DMX_OK) { fprintf(stderr, "%s\n", dmx_last_error()); return 1; }
    }
  }
  std::vector<std::string> bc((size_t)B), sm((size_t)V);
  for (int c = 0; c < B; ++c) { char b[32]; uint64_t x = rg.next(); for (int i = 0; i < 16; ++i) { b[i] = "ACGT"[x & 3]; x >>= 2; } b[16] = 0; bc[c] = std::string(b) + "-1"; }
  for (int j = 0; j < V; ++j) sm[j] = "SM" + std::to_string(j);
  std::vector<const char*> smp((size_t)V), bcp((size_t)B);
  for (int j = 0; j < V; ++j) smp[j] = sm[j].c_str();
  for (int c = 0; c < B; ++c) bcp[c] = bc[c].c_str();
  const double p_more = rbar > 1.0 ? 1.0 - 1.0 / rbar : 0.0;      // reads per covered pair: geometric with mean rbar

  dmx_job job;
  memset(&job, 0, sizeof job);
  dmx_job_timing tm;
  memset(&tm, 0, sizeof tm);
  dmx_store* st = nullptr;
  dmx_pileup pl;
  memset(&pl, 0, sizeof pl);
  std::vector<int64_t> pair_off, read_off;
  std::vector<int32_t> pair_snp, totl, pass, uniq;
  std::vector<uint8_t> nrd, reads;
  double store_s = 0;
  long n_obs = 0;
  if (from_store) {
    st = dmx_store_new();
    for (int s = 0; s < S; ++s) dmx_store_add_snp(st);
    const double t0 = now();
    char umi[16];
    for (int s = 0; s < S; ++s)              // BAM order = SNP-major: for every SNP the reads of the cells that cover it
      for (int c = 0; c < B; ++c) {
        if (rg.unif() >= delta) continue;
        const int32_t ib = dmx_store_add_cell(st, bc[c].c_str());
        const int src = c % V;
        int nr = 1; while (rg.unif() < p_more && nr < 12) ++nr;
        for (int r = 0; r < nr; ++r) {
          dmx_store_count_read(st, ib);
          const int bq = 13 + (int)(rg.next() % 28);
          const bool alt = rg.unif() < 0.5 * dos[(size_t)s * V + src];
          snprintf(umi, sizeof umi, "U%07llu", (unsigned long long)(rg.next() % 10000000ull));
          dmx_store_add_read(st, s, ib, umi, alt ? 1 : 0, bq);
          ++n_obs;
        }
      }
    store_s = now() - t0;
    job.store = st;
  } else {
    // the frozen pileup directly: cells in parallel, two passes (count, fill)
    const double t0 = now();
    const bool dense = delta >= 1.0;
    const int nt = std::max(1u, std::min(64u, std::thread::hardware_concurrency()));
    pair_off.assign((size_t)B + 1, 0); read_off.assign((size_t)B + 1, 0);
    totl.assign((size_t)B, 0); pass.assign((size_t)B, 0); uniq.assign((size_t)B, 0);
    auto cell_walk = [&](int c, bool fill) {
      Rng r(0xC0FFEE00ull + (uint64_t)c * 7919);
      int64_t p = fill ? pair_off[c] : 0, q = fill ? read_off[c] : 0;
      const int src = c % V;
      for (int s = 0; s < S; ++s) {
        if (!dense && r.unif() >= delta) continue;
        int nr = 1; while (r.unif() < p_more && nr < 12) ++nr;
        if (fill) { if (!dense) pair_snp[(size_t)p] = s; nrd[(size_t)p] = (uint8_t)nr; }
        for (int k = 0; k < nr; ++k) {
          const int bq = 13 + (int)(r.next() % 28);
          const bool alt = r.unif() < 0.5 * dos[(size_t)s * V + src];
          if (fill) reads[(size_t)q] = (uint8_t)((alt ? 0x80 : 0) | bq);
          ++q;
        }
        ++p;
      }
      if (!fill) { pair_off[(size_t)c + 1] = p; read_off[(size_t)c + 1] = q; totl[c] = pass[c] = uniq[c] = (int32_t)q; }
    };
    auto par = [&](bool fill) {
      std::vector<std::thread> th;
      for (int t = 0; t < nt; ++t) th.emplace_back([&, t] { for (int c = t; c < B; c += nt) cell_walk(c, fill); });
      for (auto& x : th) x.join();
    };
    par(false);
    for (int c = 0; c < B; ++c) { pair_off[(size_t)c + 1] += pair_off[c]; read_off[(size_t)c + 1] += read_off[c]; }
    if (!dense) pair_snp.resize((size_t)pair_off[B] + 1);
    nrd.resize((size_t)pair_off[B] + 4); reads.resize((size_t)read_off[B] + 4);
    par(true);
    n_obs = (long)read_off[B];
    pl.n_cells = B; pl.n_snps = S; pl.n_pairs = pair_off[B]; pl.n_reads = read_off[B];
    pl.cell_pair_off = pair_off.data(); pl.cell_read_off = read_off.data(); pl.pair_snp = dense ? nullptr : pair_snp.data();
    pl.pair_nrd = nrd.data(); pl.nrd_width = 1; pl.memory = DMX_MEM_HOST; pl.reads = reads.data();
    pl.rd_totl = totl.data(); pl.rd_pass = pass.data(); pl.rd_uniq = uniq.data();
    store_s = now() - t0;
    job.pileup = &pl; job.barcodes = bcp.data();
  }
  const double alpha[2] = {0.0, 0.5};
  job.g = G.data(); job.n_samples = V; job.sample_ids = smp.data(); job.n_alpha = 2; job.alpha = alpha; job.doublet_prior = 0.5;
  job.write_pair = write_pair; job.out_prefix = "/tmp/e2e_bench_out"; job.device = 0; job.arbiter = arbiter; job.n_gpus = n_gpus;
  job.mode = fast ? DMX_MODE_FAST : DMX_MODE_STRICT; job.timing = &tm;
  const double t2 = now();
  if (dmx_demuxlet_run(&job) != DMX_OK) { fprintf(stderr, "%s\n", dmx_last_error()); return 1; }
  const double run_s = now() - t2;
  long long P = from_store ? 0 : (long long)pl.n_pairs;
  if (from_store) { dmx_pileup f; dmx_store_freeze(st, &f); P = (long long)f.n_pairs; }
  printf("{\"mode\": \"%s\", \"barcodes\": %d, \"snps\": %d, \"samples\": %d, \"density\": %g, \"rbar\": %g, \"field\": \"%s\", \"write_pair\": %d, "
         "\"engine_mode\": \"%s\", \"arbiter\": %d, \"n_gpus\": %d, \"observations\": %ld, \"covered_pairs\": %lld, \"%s\": %.3f, "
         "\"dmx_demuxlet_run_s\": %.3f, \"stages\": {\"freeze_s\": %.3f, \"setup_s\": %.3f, \"stage_h2d_s\": %.3f, \"gpu_wait_s\": %.3f, "
         "\"arbiter_format_write_s\": %.3f, \"kernel_ms_sum\": %.1f, \"ranges\": %d, \"engines\": %d, \"cells_grid_fetched\": %d}, "
         "\"pair_evals_per_s_end_to_end\": %.4g, \"triples_per_s_end_to_end\": %.4g}\n",
         argv[1], B, S, V, delta, rbar, argv[7], write_pair, fast ? "fast" : "strict", arbiter, n_gpus, n_obs, P,
         from_store ? "store_add_read_s" : "generate_pileup_s", store_s, run_s, tm.freeze_s, tm.setup_s, tm.stage_s, tm.wait_s, tm.write_s,
         tm.kernel_ms, tm.n_ranges, tm.n_engines, tm.n_cells_grid_fetched, (double)P * V * V * 2 / run_s, (double)P * V / run_s);
  if (st) dmx_store_free(st);
  return 0;
}
