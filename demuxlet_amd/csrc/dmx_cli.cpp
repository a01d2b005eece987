// demuxlet (MI355X) — command-line front end: demuxlet's option surface, built-in SAM/BAM and VCF readers (no htslib),
// the lock-step BAM x VCF scan that builds the UMI-deduplicated pileup, then libdmx for everything from
// cmd_cram_demuxlet.cpp:390 on.  Rows f1..f4 of SURVEY.md §8: parity here is UNPINNED by the reference (its L4 needs
// htslib, absent from this image); behaviour is restated from the cited lines and pinned by our own fixtures.
//
//   f4  options            cmd_cram_demuxlet.cpp:37-72 (names, types, defaults), params.cpp:114-185,449-486,552-574
//   f2  SAM/BAM reader     sam_filtered_reader.cpp:180-296 (MQ / flag filter), plain SAM text, gzip'd SAM, BAM (BGZF = gzip
//                          members, decoded with zlib); CRAM and BCF are not supported
//   f3  VCF reader+filter  bcf_filtered_reader.cpp:498-574 (n_allele, call rate, MAC), :98-141 (--sm / --sm-list: a std::set,
//                          so selected samples come in SORTED id order), :671-765 (buffering)
//   f1  scan               cmd_cram_demuxlet.cpp:142-338, CIGAR walk hts_utils.cpp:279-359 (M, D/N, S/I only)
#include <sys/resource.h>
#include <zlib.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <ctime>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <map>
#include <set>
#include <unordered_map>
#include <string>
#include <vector>

#include "dmx.h"
#include "dmx_inflate.hpp"

namespace {

// ---- messages in the reference's shape (Error.cpp:27-86)
void notice(const char* fmt, ...) {
  char buf[255];
  time_t now = time(nullptr);
  strftime(buf, 120, "%Y/%m/%d %H:%M:%S", localtime(&now));
  fprintf(stderr, "NOTICE [%s] - ", buf);
  va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap);
  fprintf(stderr, "\n");
}
void warning(const char* fmt, ...) {
  fprintf(stderr, "\n\aWARNING - \n");
  va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap);
  fprintf(stderr, "\n");
}
std::thread g_warm_up;       // dmx_device_warm_up beside the scan (main): exit() must not run the HIP runtime's handlers under its feet
[[noreturn]] void fatal(const char* fmt, ...) {
  static std::atomic<bool> dying{false};
  if (dying.exchange(true)) for (;;) std::this_thread::sleep_for(std::chrono::seconds(1));   // a second thread's error: the first one is on its way out
  fprintf(stderr, "\nFATAL ERROR - \n");
  va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap);
  fprintf(stderr, "\n\n");
  fflush(stderr);
  if (g_warm_up.joinable() && std::this_thread::get_id() != g_warm_up.get_id()) g_warm_up.join();
  // the reference throws an uncaught exception here (Error.cpp:39): abnormal termination either way.  _exit, not exit: pool workers, the
  // reader, the sink, the BGZF producer and the feed thread may all be running, and exit() would run static destructors and the HIP
  // runtime's atexit handlers under their feet
  fflush(stdout);
  _exit(EXIT_FAILURE);
}

// ---- f4: options --------------------------------------------------------------------------------------------------
struct Options {
  std::string sam, tag_group = "CB", tag_umi = "UB";
  std::string vcf, field = "GP";
  double geno_error = 0.01;
  int min_mac = 1;
  double min_callrate = 0.5;
  std::vector<std::string> sm;
  std::string sm_list;
  std::string out;
  std::vector<double> alpha;
  bool write_pair = false;
  double doublet_prior = 0.5;
  int sam_verbose = 1000000, vcf_verbose = 10000;
  int cap_bq = 40, min_bq = 13, min_mq = 20, min_td = 0, excl_flag = 0x0f04;
  std::string group_list;
  int min_total = 0, min_uniq = 0, min_snp = 0;
  // additions of this implementation (do not collide with any reference option)
  int gpu = 0, gpus = 1;
  bool pileup_only = false;      // stop after the scan and write <out>.pileup.txt (no GPU needed)
  bool no_arbiter = false;
  bool fast = false;       // opt-in: DMX_MODE_FAST (printed entries only; a printed number may differ from the reference in its last digit)
  bool strict = false;     // the default, spelled out (kept for round-2 command lines)
};

enum OptType { O_BOOL, O_INT, O_DOUBLE, O_STRING, O_MULTI_DOUBLE, O_MULTI_STRING };
struct OptDef { const char* name; OptType type; void* ptr; const char* help; const char* group; };

bool check_integer(const char* s) {
  if (!s || !*s) return false;
  char* end; strtol(s, &end, 0);
  return *end == 0;
}
bool check_double(const char* s) {
  if (!s || !*s) return false;
  char* end; strtod(s, &end);
  return *end == 0;
}

// ---- the reference's option printouts (params.cpp) --------------------------------------------------------------------
// paramList::Read prints the help (params.cpp:306-405,:527-550) on -h/--help and exits 1; paramList::Status (:188-303,:552-574)
// echoes every run's effective options.  Both are reproduced character for character for the reference's own options (the
// group and name column widths follow the longest group title and option name, :96-104; this build's extra group is shorter
// than either, so it does not move a column).
std::string opt_state(const OptDef& d, bool help) {
  char buf[64];
  std::string st;
  auto dbl = [&](double v) { if (v == 0.0 || v >= 0.01) snprintf(buf, sizeof buf, "%.*f", 2, v); else snprintf(buf, sizeof buf, "%.1e", v); return std::string(buf); };   // precision = 2 (:111)
  switch (d.type) {
    case O_BOOL: st = help ? (*(bool*)d.ptr ? "[FLG: ON]" : "[FLG: OFF]") : (*(bool*)d.ptr ? "[ON]" : ""); break;
    case O_INT: {
      const int v = *(int*)d.ptr;
      if (v == 0) st = help ? "[INT: 0]" : "";
      else { snprintf(buf, sizeof buf, help ? "[INT: %d]" : "[%d]", v); st = buf; }
      break;
    }
    case O_DOUBLE: {
      const double v = *(double*)d.ptr;
      if (v != v) st = help ? "[FLT: NaN]" : "";
      else st = std::string(help ? "[FLT: " : "[") + dbl(v) + "]";
      break;
    }
    case O_STRING: {
      const std::string& v = *(std::string*)d.ptr;
      st = v.empty() ? (help ? "[STR: ]" : "") : std::string(help ? "[STR: " : "[") + v + "]";
      break;
    }
    case O_MULTI_DOUBLE: {
      const std::vector<double>& v = *(std::vector<double>*)d.ptr;
      if (v.empty()) { st = help ? "[V_FLT: ]" : ""; break; }
      st = help ? "[V_FLT: " : "[";
      for (size_t i = 0; i < v.size(); ++i) { if (i) st += ", "; st += dbl(v[i]); }
      st += "]";
      break;
    }
    case O_MULTI_STRING: {
      const std::vector<std::string>& v = *(std::vector<std::string>*)d.ptr;
      if (v.empty()) { st = help ? "[V_STR: ]" : ""; break; }
      st = help ? "[V_STR: " : "[";
      for (size_t i = 0; i < v.size(); ++i) { if (i) st += ", "; st += v[i]; }
      st += "]";
      break;
    }
  }
  return st.empty() ? st : " " + st;
}

void column_widths(const std::vector<OptDef>& defs, int& group_len, int& name_len) {
  group_len = name_len = 0;
  for (const OptDef& d : defs) { group_len = std::max(group_len, (int)strlen(d.group)); name_len = std::max(name_len, (int)strlen(d.name)); }
}

void print_help(const std::vector<OptDef>& defs) {                                 // paramList::HelpMessage, longParams::HelpMessage
  int group_len, name_len;
  column_widths(defs, group_len, name_len);
  fprintf(stderr, "\nDetailed instructions of parameters are available. Ones with \"[]\" are in effect:\n");
  fprintf(stderr, "\nAvailable Options\n\n");
  const char* grp = nullptr;
  for (const OptDef& d : defs) {
    if (!grp || strcmp(grp, d.group)) { fprintf(stderr, "\n%s\n", d.group); grp = d.group; }
    fprintf(stderr, "  --%-*s%-*s%s%s\n", name_len, d.name, 20, opt_state(d, true).c_str(), " : ", d.help);          // helpCol = 20 (:31)
  }
  fprintf(stderr, "\n\n");
  fprintf(stderr, "NOTES:\nWhen --help was included in the argument. The program prints the help message but do not actually run\n");
}

void print_status(const std::vector<OptDef>& defs) {                               // paramList::Status, longParams::Status
  int group_len, name_len;
  column_widths(defs, group_len, name_len);
  fprintf(stderr, "\nAvailable Options\n\n");
  fprintf(stderr, "The following parameters are available. Ones with \"[]\" are in effect:\n");
  const int line_start = group_len ? group_len + 5 : 0;
  bool need_a_comma = false;
  int line_len = 0;
  const char* grp = nullptr;
  for (const OptDef& d : defs) {
    if (!grp || strcmp(grp, d.group)) {
      fprintf(stderr, "%s %*s :", need_a_comma ? "\n" : "", group_len + 2, d.group);
      need_a_comma = false;
      line_len = line_start;
      grp = d.group;
    }
    const std::string state = opt_state(d, false);
    int item_len = 3 + (int)strlen(d.name) + (need_a_comma ? 1 : 0) + (int)state.size();
    if (item_len + line_len > 78 && line_len > line_start) {
      line_len = line_start;
      fprintf(stderr, "%s\n%*s", need_a_comma ? "," : "", line_len, "");
      need_a_comma = false;
      item_len -= 1;
    }
    fprintf(stderr, "%s --%s%s", need_a_comma ? "," : "", d.name, state.c_str());
    need_a_comma = true;
    line_len += item_len;
  }
  fprintf(stderr, "\n");
  fprintf(stderr, "\nRun with --help for more detailed help messages of each argument.\n");
  fprintf(stderr, "\n");
}

void parse_options(int argc, char** argv, Options& o) {
  std::vector<OptDef> defs = {
      {"sam", O_STRING, &o.sam, "Input SAM/BAM/CRAM file. Must be sorted by coordinates and indexed", "Options for input SAM/BAM/CRAM"},
      {"tag-group", O_STRING, &o.tag_group, "Tag representing readgroup or cell barcodes, in the case to partition the BAM file into multiple groups. For 10x genomics, use CB", "Options for input SAM/BAM/CRAM"},
      {"tag-UMI", O_STRING, &o.tag_umi, "Tag representing UMIs. For 10x genomiucs, use UB", "Options for input SAM/BAM/CRAM"},
      {"vcf", O_STRING, &o.vcf, "Input VCF/BCF file, containing the individual genotypes (GT), posterior probability (GP), or genotype likelihood (PL)", "Options for input VCF/BCF"},
      {"field", O_STRING, &o.field, "FORMAT field to extract the genotype, likelihood, or posterior from", "Options for input VCF/BCF"},
      {"geno-error", O_DOUBLE, &o.geno_error, "Genotype error rate (must be used with --field GT)", "Options for input VCF/BCF"},
      {"min-mac", O_INT, &o.min_mac, "Minimum minor allele frequency", "Options for input VCF/BCF"},
      {"min-callrate", O_DOUBLE, &o.min_callrate, "Minimum call rate", "Options for input VCF/BCF"},
      {"sm", O_MULTI_STRING, &o.sm, "List of sample IDs to compare to (default: use all)", "Options for input VCF/BCF"},
      {"sm-list", O_STRING, &o.sm_list, "File containing the list of sample IDs to compare", "Options for input VCF/BCF"},
      {"out", O_STRING, &o.out, "Output file prefix", "Output Options"},
      {"alpha", O_MULTI_DOUBLE, &o.alpha, "Grid of alpha to search for (default is 0, 0.5)", "Output Options"},
      {"write-pair", O_BOOL, &o.write_pair, "Writing the (HUGE) pair file", "Output Options"},
      {"doublet-prior", O_DOUBLE, &o.doublet_prior, "Prior of doublet", "Output Options"},
      {"sam-verbose", O_INT, &o.sam_verbose, "Verbose message frequency for SAM/BAM/CRAM", "Output Options"},
      {"vcf-verbose", O_INT, &o.vcf_verbose, "Verbose message frequency for VCF/BCF", "Output Options"},
      {"cap-BQ", O_INT, &o.cap_bq, "Maximum base quality (higher BQ will be capped)", "Read filtering Options"},
      {"min-BQ", O_INT, &o.min_bq, "Minimum base quality to consider (lower BQ will be skipped)", "Read filtering Options"},
      {"min-MQ", O_INT, &o.min_mq, "Minimum mapping quality to consider (lower MQ will be ignored)", "Read filtering Options"},
      {"min-TD", O_INT, &o.min_td, "Minimum distance to the tail (lower will be ignored)", "Read filtering Options"},
      {"excl-flag", O_INT, &o.excl_flag, "SAM/BAM FLAGs to be excluded", "Read filtering Options"},
      {"group-list", O_STRING, &o.group_list, "List of tag readgroup/cell barcode to consider in this run. All other barcodes will be ignored. This is useful for parallelized run", "Cell/droplet filtering options"},
      {"min-total", O_INT, &o.min_total, "Minimum number of total reads for a droplet/cell to be considered", "Cell/droplet filtering options"},
      {"min-uniq", O_INT, &o.min_uniq, "Minimum number of unique reads (determined by UMI/SNP pair) for a droplet/cell to be considered", "Cell/droplet filtering options"},
      {"min-snp", O_INT, &o.min_snp, "Minimum number of SNPs with coverage for a droplet/cell to be considered", "Cell/droplet filtering options"},
      {"gpu", O_INT, &o.gpu, "[MI355X build] HIP device ordinal", "MI355X build"},
      {"gpus", O_INT, &o.gpus, "[MI355X build] number of GPUs to shard the barcodes over (starting at --gpu)", "MI355X build"},
      {"pileup-only", O_BOOL, &o.pileup_only, "[MI355X build] stop after the BAM x VCF scan and write <out>.pileup.txt", "MI355X build"},
      {"no-arbiter", O_BOOL, &o.no_arbiter, "[MI355X build] skip the host tie arbiter (DESIGN.md, Ties)", "MI355X build"},
      {"strict", O_BOOL, &o.strict, "[MI355X build] DMX_MODE_STRICT (the default): every grid entry in the reference's operation order", "MI355X build"},
      {"fast", O_BOOL, &o.fast, "[MI355X build] DMX_MODE_FAST: only the entries demuxlet prints or decides on, 2-4x faster; same calls, log-likelihoods within 1e-9 of the reference (a printed number may differ in its last digit)", "MI355X build"},
  };
  std::set<std::string> touched;
  std::string errors;
  char ebuf[512];
  for (int i = 1; i < argc; ++i) {
    const char* a = argv[i];
    if (a[0] == '-' && a[1]) {
      if (a[1] == 'h' || strcmp(a, "--help") == 0) { print_help(defs); exit(1); }            // params.cpp:457-463
      if (a[1] != '-') { snprintf(ebuf, sizeof ebuf, "Command line parameter %s (#%d) not recognized\n", a, i); errors += ebuf; continue; }
      const char* name = a + 2;
      const OptDef* d = nullptr;
      for (const OptDef& c : defs) if (strcmp(c.name, name) == 0) { d = &c; break; }
      if (!d) { snprintf(ebuf, sizeof ebuf, "Command line parameter %s (#%d) not recognized\n", a, i); errors += ebuf; continue; }
      const char* extra = (i + 1 < argc) ? argv[i + 1] : nullptr;
      // longParams::TranslateExtras (params.cpp:113-186): an invalid or repeated option does NOT stop the parse — param::error
      // appends the message (no newline) to the list that paramList::Status reports AFTER the status echo (:562-567); a repeated
      // option keeps its first value; the value argument is consumed either way.  The "[E:file:line function]" prefix of the
      // reference carries its __FILE__, i.e. the path it was compiled from: written here as a build in the source directory has it.
      auto bad = [&](int line, const char* what) { snprintf(ebuf, sizeof ebuf, "[E:params.cpp:%d TranslateExtras] Invalid argument --%s %s. %s was expected", line, name, extra ? extra : "(null)", what); errors += ebuf; };
      auto again = [&](int line, const char* tail) { snprintf(ebuf, sizeof ebuf, "[E:params.cpp:%d TranslateExtras] Redundant use of option --%s%s is not allowed", line, name, tail); errors += ebuf; };
      const bool seen = touched.count(name) != 0;
      switch (d->type) {
        case O_BOOL: if (seen) again(122, " or its exclusive neighbor"); *(bool*)d->ptr = true; touched.insert(name); break;
        case O_INT: if (!check_integer(extra)) bad(139, "Integer"); else if (seen) again(140, ""); else { *(int*)d->ptr = atoi(extra); touched.insert(name); } ++i; break;
        case O_DOUBLE: if (!check_double(extra)) bad(147, "Double"); else if (seen) again(148, ""); else { *(double*)d->ptr = atof(extra); touched.insert(name); } ++i; break;
        case O_STRING: if (!extra) bad(155, "String"); else if (seen) again(156, ""); else { *(std::string*)d->ptr = extra; touched.insert(name); } ++i; break;
        case O_MULTI_DOUBLE: if (!check_double(extra)) bad(168, "Double"); else ((std::vector<double>*)d->ptr)->push_back(atof(extra)); ++i; break;
        case O_MULTI_STRING: if (!extra) bad(174, "String"); else ((std::vector<std::string>*)d->ptr)->push_back(extra); ++i; break;
      }
    } else {
      snprintf(ebuf, sizeof ebuf, "Cannot correspond command line parameter %s (#%d) to any of the options\n", a, i); errors += ebuf;
    }
  }
  print_status(defs);                                                                                                // params.cpp:552-560
  if (!errors.empty()) fatal("[E:params.cpp:564 Status] Problems encountered parsing command line:\n\n%s", errors.c_str());   // params.cpp:562-567
}

// ---- line / byte input: plain files, gzip, and BGZF (bgzip'd VCF, BAM) ------------------------------------------------
// BGZF is a series of independent <= 64 KiB gzip members whose compressed size sits in a 'BC' extra field (SAM spec 4.1), so
// the members of a batch are inflated on several host threads while the parser consumes the previous batch; inflate is what
// bounds a single-threaded BAM scan (0.8 M reads/s here against 3 M reads/s for the same records as text).  Anything that
// is not BGZF goes through zlib's gzFile (which also passes plain text through).
int cli_threads() {
  if (const char* e = getenv("DMX_THREADS")) { const int n = atoi(e); if (n >= 1) return std::min(n, 64); }
  int n = (int)std::thread::hardware_concurrency();
  if (n < 1) n = 1;
  if (FILE* q = fopen("/sys/fs/cgroup/cpu.max", "r")) {
    char quota[32]; long long period = 0;
    if (fscanf(q, "%31s %lld", quota, &period) == 2 && strcmp(quota, "max") != 0 && period > 0) n = std::max(1, std::min(n, (int)(atoll(quota) / period)));
    fclose(q);
  }
  return std::min(n, 32);
}

// A run of decompressed bytes.  Readers hand out views into it (BAM records are parsed where the inflater wrote them), so it is
// shared: it lives until the last window of reads that points into it has been through the store.
struct Chunk {
  std::unique_ptr<uint8_t[]> p;
  size_t n = 0, cap = 0;
  const uint8_t* data() const { return p.get(); }
  uint8_t* data() { return p.get(); }
  size_t size() const { return n; }
  // Full-size BGZF batches (256 members x 64 KiB) come from and go back to a free list: a fresh 16 MiB allocation is 4 096 page
  // faults, 0.3 CPU-seconds over a 440 MB BAM.
  static constexpr size_t kPooled = (size_t)256 << 16;
  struct Pool { std::mutex mu; std::vector<std::unique_ptr<uint8_t[]>> free; };
  static Pool& pool() { static Pool* p = new Pool; return *p; }
  void alloc(size_t bytes) {
    if (bytes <= kPooled) {
      cap = kPooled;
      Pool& q = pool();
      { std::lock_guard<std::mutex> lk(q.mu); if (!q.free.empty()) { p = std::move(q.free.back()); q.free.pop_back(); } }
      if (!p) p.reset(new uint8_t[kPooled]);
    } else { cap = bytes; p.reset(new uint8_t[bytes]); }
  }
  ~Chunk() {
    if (p && cap == kPooled) { Pool& q = pool(); std::lock_guard<std::mutex> lk(q.mu); if (q.free.size() < 16) q.free.push_back(std::move(p)); }
  }
};
using ChunkP = std::shared_ptr<Chunk>;

// DMX_CLI_TIMING: CPU seconds (CLOCK_THREAD_CPUTIME_ID) spent inside the parallel sections, summed over their worker threads
std::atomic<int64_t> g_cpu_ns[4];                    // 0 inflate, 1 record parsing, 2 overlap
static const bool g_cpu_on = getenv("DMX_CLI_TIMING") != nullptr;
inline int64_t thread_cpu_ns() { timespec ts; clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts); return (int64_t)ts.tv_sec * 1000000000 + ts.tv_nsec; }
struct CpuScope { int i; int64_t t0; explicit CpuScope(int i_) : i(i_), t0(g_cpu_on ? thread_cpu_ns() : 0) {} ~CpuScope() { if (g_cpu_on) g_cpu_ns[i] += thread_cpu_ns() - t0; } };

// The host's worker threads, shared by every parallel section of the scan (BGZF inflate, record parsing, read x SNP overlap): a section
// posts a job, the posting thread works on it too, and idle workers join the oldest job that still wants hands.  Several sections
// run at the same time (the stages of the pipeline work on different windows); with one pool the process stays at n_threads busy
// threads instead of creating ~1 800 short-lived ones per 2e6 reads.
struct WorkerPool {
  struct Job { std::function<void(int)> fn; int want = 0, taken = 0, done = 0; bool closed = false; };
  std::mutex mu;
  std::condition_variable cv_work, cv_done;
  std::deque<std::shared_ptr<Job>> jobs;
  int n_workers = 0;
  static WorkerPool& get() { static WorkerPool* p = new WorkerPool(cli_threads() - 1); return *p; }   // (never destroyed: workers sleep until exit)
  explicit WorkerPool(int n) : n_workers(std::max(0, n)) {
    for (int i = 0; i < n_workers; ++i) std::thread([this] { loop(); }).detach();
  }
  void loop() {
    std::unique_lock<std::mutex> lk(mu);
    for (;;) {
      std::shared_ptr<Job> j;
      for (auto& q : jobs) if (!q->closed && q->taken < q->want) { j = q; break; }
      if (!j) { cv_work.wait(lk); continue; }
      const int slot = ++j->taken;               // the poster is slot 0
      lk.unlock();
      j->fn(slot);
      lk.lock();
      ++j->done;
      cv_done.notify_all();
    }
  }
  // fn(slot) on the calling thread (slot 0) and on up to `helpers` workers (slots 1..helpers); returns when all of them returned.
  // fn is expected to pull its work items off a shared counter: a worker that joins late simply finds nothing left.
  void run(int helpers, const std::function<void(int)>& fn) {
    helpers = std::min(helpers, n_workers);
    if (helpers <= 0) { fn(0); return; }
    auto j = std::make_shared<Job>();
    j->fn = fn; j->want = helpers;
    { std::lock_guard<std::mutex> lk(mu); jobs.push_back(j); }
    cv_work.notify_all();
    fn(0);
    std::unique_lock<std::mutex> lk(mu);
    j->closed = true;
    for (auto it = jobs.begin(); it != jobs.end(); ++it) if (*it == j) { jobs.erase(it); break; }
    cv_done.wait(lk, [&] { return j->done == j->taken; });
  }
};

struct BgzfPipe {
  FILE* fp = nullptr;
  std::thread producer;
  std::mutex mu;
  std::condition_variable cv_put, cv_get;
  std::deque<ChunkP> queue;                     // inflated batches, in file order
  bool done = false, stop = false;
  std::string error;
  static constexpr size_t kBatchBlocks = 256, kQueueDepth = 3;

  struct Block { std::vector<uint8_t> raw; size_t data_off = 0, data_len = 0; uint32_t crc = 0, isize = 0; std::string err; };

  // one BGZF member into b.raw; false at a clean EOF
  bool read_block(Block& b) {
    uint8_t h[12];
    const size_t got = fread(h, 1, 12, fp);
    if (got == 0) return false;
    if (got != 12 || h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4)) { error = "corrupt BGZF block header"; return false; }
    const size_t xlen = (size_t)h[10] | ((size_t)h[11] << 8);
    std::vector<uint8_t> extra(xlen);
    if (fread(extra.data(), 1, xlen, fp) != xlen) { error = "truncated BGZF block"; return false; }
    size_t bsize = 0;
    for (size_t o = 0; o + 4 <= xlen;) {
      const size_t sl = (size_t)extra[o + 2] | ((size_t)extra[o + 3] << 8);
      if (extra[o] == 'B' && extra[o + 1] == 'C' && sl == 2 && o + 6 <= xlen) bsize = ((size_t)extra[o + 4] | ((size_t)extra[o + 5] << 8)) + 1;
      o += 4 + sl;
    }
    if (bsize < 12 + xlen + 8) { error = "BGZF block without a BC field"; return false; }
    const size_t rest = bsize - 12 - xlen;          // deflate data + CRC32 + ISIZE
    b.raw.resize(rest + 56);                         // (dmxz::inflate_raw reads ahead: 64 readable bytes past the deflate data)
    if (fread(b.raw.data(), 1, rest, fp) != rest) { error = "truncated BGZF block"; return false; }
    memset(b.raw.data() + rest, 0, 56);
    b.data_off = 0; b.data_len = rest - 8;
    memcpy(&b.crc, &b.raw[rest - 8], 4); memcpy(&b.isize, &b.raw[rest - 4], 4);
    if (b.isize > 65536) { error = "BGZF block with ISIZE > 64 KiB"; return false; }   // SAM spec 4.1; ISIZE sizes the batch allocation
    return true;
  }
  static void inflate_block(Block& b, uint8_t* out) {            // into the block's place in its batch (ISIZE bytes)
    if (b.isize == 0) return;
    // our decoder first (dmx_inflate.hpp: 2-3x zlib 1.2.11 on BAM records); anything it does not accept goes to zlib, which decides
    static const bool zlib_only = getenv("DMX_ZLIB_INFLATE") != nullptr;
    if (!zlib_only && dmxz::inflate_raw(b.raw.data() + b.data_off, b.data_len, out, b.isize) && dmxz::crc32_of(out, b.isize) == b.crc) return;
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    if (inflateInit2(&zs, -15) != Z_OK) { b.err = "inflateInit2 failed"; return; }
    zs.next_in = b.raw.data() + b.data_off; zs.avail_in = (uInt)b.data_len;
    zs.next_out = out; zs.avail_out = (uInt)b.isize;
    const int rc = inflate(&zs, Z_FINISH);
    inflateEnd(&zs);
    if (rc != Z_STREAM_END || zs.avail_out != 0) { b.err = "BGZF block does not inflate to its ISIZE"; return; }
    if ((uint32_t)crc32(crc32(0L, Z_NULL, 0), out, (uInt)b.isize) != b.crc) b.err = "BGZF block CRC mismatch";
  }
  // Two threads: one reads the members of batch i + 1 from the file while this one has batch i inflated on the host's threads,
  // every member straight into its place in the batch (the ISIZE fields give the places before anything is inflated).
  void run(int nthreads) {
    struct RawBatch { std::vector<Block> blocks; size_t n = 0; std::string err; int state = 0; };   // state: 0 free, 1 read
    RawBatch rb[2];
    for (RawBatch& r : rb) r.blocks.resize(kBatchBlocks);
    std::mutex io_mu; std::condition_variable io_cv;
    bool io_stop = false;
    std::thread io([&] {
      for (int i = 0;; i ^= 1) {
        { std::unique_lock<std::mutex> lk(io_mu); io_cv.wait(lk, [&] { return rb[i].state == 0 || io_stop; }); if (io_stop) return; }
        size_t n = 0;
        while (n < kBatchBlocks && read_block(rb[i].blocks[n])) ++n;
        rb[i].n = n; rb[i].err = error;
        { std::lock_guard<std::mutex> lk(io_mu); rb[i].state = 1; }
        io_cv.notify_all();
        if (n < kBatchBlocks || !error.empty()) return;
      }
    });
    struct IoGuard { std::mutex& mu; std::condition_variable& cv; bool& stop; std::thread& th; ~IoGuard() { { std::lock_guard<std::mutex> lk(mu); stop = true; } cv.notify_all(); if (th.joinable()) th.join(); } } io_guard{io_mu, io_cv, io_stop, io};
    for (int bi = 0;; bi ^= 1) {
      { std::unique_lock<std::mutex> lk(io_mu); io_cv.wait(lk, [&] { return rb[bi].state == 1; }); }
      std::vector<Block>& blocks = rb[bi].blocks;
      const size_t n = rb[bi].n;
      std::string err = rb[bi].err;
      if (n) {
        std::vector<size_t> off(n + 1, 0);
        for (size_t i = 0; i < n; ++i) off[i + 1] = off[i] + blocks[i].isize;
        ChunkP batch = std::make_shared<Chunk>();
        batch->alloc(off[n]);
        batch->n = off[n];
        const int nt = (int)std::min<size_t>((size_t)nthreads, n);
        std::atomic<size_t> next{0};
        WorkerPool::get().run(nt - 1, [&](int) { CpuScope cs(0); for (size_t i; (i = next.fetch_add(1)) < n;) inflate_block(blocks[i], batch->p.get() + off[i]); });
        for (size_t i = 0; i < n; ++i) if (err.empty() && !blocks[i].err.empty()) { err = blocks[i].err; blocks[i].err.clear(); }
        { std::lock_guard<std::mutex> lk(io_mu); rb[bi].state = 0; }
        io_cv.notify_all();
        std::unique_lock<std::mutex> lk(mu);
        cv_put.wait(lk, [&] { return queue.size() < kQueueDepth || stop; });
        if (stop) return;
        queue.push_back(std::move(batch));
        cv_get.notify_one();
      }
      if (n < kBatchBlocks || !err.empty()) {
        std::lock_guard<std::mutex> lk(mu);
        error = err; done = true;
        cv_get.notify_one();
        return;
      }
    }
  }
  void start(FILE* f, int nthreads) { fp = f; producer = std::thread([this, nthreads] { run(nthreads); }); }
  // next inflated batch; false at EOF (or error: see `error`)
  bool next(ChunkP& out) {
    std::unique_lock<std::mutex> lk(mu);
    cv_get.wait(lk, [&] { return !queue.empty() || done; });
    if (queue.empty()) return false;
    out = std::move(queue.front());
    queue.pop_front();
    cv_put.notify_one();
    return true;
  }
  ~BgzfPipe() {
    { std::lock_guard<std::mutex> lk(mu); stop = true; }
    cv_put.notify_all();
    if (producer.joinable()) producer.join();
    if (fp) fclose(fp);
  }
};

struct GzIn {
  gzFile f = nullptr;                            // plain text / ordinary gzip
  std::unique_ptr<BgzfPipe> bgzf;                // BGZF
  std::string path;
  ChunkP cur_p = std::make_shared<Chunk>();      // current decompressed chunk (shared: BAM records are parsed in place)
  size_t pos = 0;
  bool open(const std::string& p) {
    path = p;
    FILE* fp = fopen(p.c_str(), "rb");
    if (!fp) return false;
    uint8_t h[18];
    const size_t got = fread(h, 1, sizeof h, fp);
    const bool is_bgzf = got == 18 && h[0] == 0x1f && h[1] == 0x8b && h[2] == 8 && (h[3] & 4) && h[12] == 'B' && h[13] == 'C';
    if (is_bgzf) {
      rewind(fp);
      bgzf.reset(new BgzfPipe);
      bgzf->start(fp, cli_threads());
      return true;
    }
    fclose(fp);
    f = gzopen(p.c_str(), "rb");
    if (f) gzbuffer(f, 1 << 20);
    return f != nullptr;
  }
  ~GzIn() { if (f) gzclose(f); }
  bool fill() {                                  // next chunk into cur; false at EOF
    pos = 0;
    if (bgzf) {
      if (bgzf->next(cur_p)) return true;
      if (!bgzf->error.empty()) fatal("[E:%s] %s: %s", __func__, path.c_str(), bgzf->error.c_str());
      cur_p = std::make_shared<Chunk>();
      return false;
    }
    if (cur_p.use_count() != 1 || !cur_p->p) { cur_p = std::make_shared<Chunk>(); cur_p->p.reset(new uint8_t[1 << 20]); }
    const int n = gzread(f, cur_p->data(), 1u << 20);
    if (n < 0) fatal("[E:%s] %s: read error", __func__, path.c_str());
    cur_p->n = (size_t)n;
    return n > 0;
  }
  bool getline(std::string& line) {
    line.clear();
    bool any = false;
    for (;;) {
      if (pos >= cur_p->size() && !fill()) break;
      any = true;
      const uint8_t* b = cur_p->data() + pos;
      const uint8_t* nl = (const uint8_t*)memchr(b, '\n', cur_p->size() - pos);
      if (nl) {
        line.append((const char*)b, (size_t)(nl - b));
        pos += (size_t)(nl - b) + 1;
        if (!line.empty() && line.back() == '\r') line.pop_back();
        return true;
      }
      line.append((const char*)b, cur_p->size() - pos);
      pos = cur_p->size();
    }
    return any && !line.empty();
  }
  bool read(void* dst, size_t n) {
    uint8_t* d = (uint8_t*)dst;
    while (n) {
      if (pos >= cur_p->size() && !fill()) return false;
      const size_t k = std::min(n, cur_p->size() - pos);
      memcpy(d, cur_p->data() + pos, k);
      d += k; pos += k; n -= k;
    }
    return true;
  }
  // a record's length prefix: 1 = read, 0 = clean end of file (not one byte left), -1 = the file ends inside the prefix
  int read_prefix(void* dst, size_t n) {
    uint8_t* d = (uint8_t*)dst;
    size_t got = 0;
    while (got < n) {
      if (pos >= cur_p->size() && !fill()) return got ? -1 : 0;
      const size_t k = std::min(n - got, cur_p->size() - pos);
      memcpy(d + got, cur_p->data() + pos, k);
      got += k; pos += k;
    }
    return 1;
  }
  void unread_all_to(size_t p) { pos = p; }      // rewind inside the first chunk (SAM text sniffing)
};

std::vector<std::string> split(const std::string& s, char sep) {
  std::vector<std::string> out;
  size_t b = 0;
  for (;;) {
    size_t e = s.find(sep, b);
    if (e == std::string::npos) { out.push_back(s.substr(b)); break; }
    out.push_back(s.substr(b, e - b));
    b = e + 1;
  }
  return out;
}

// ---- f3: VCF reader + variant filter ------------------------------------------------------------------------------------
struct Variant {
  int rid; int64_t pos;        // 0-based
  int rlen, n_allele;
  char ref0, alt0;
  std::string ref, alt;
  std::vector<float> gps;       // [nv*3] of the selected samples
};

struct VcfReader {
  GzIn in;
  std::vector<std::string> samples;            // all header samples
  std::vector<int> sm_icols;                   // selected columns
  std::map<std::string, int> contig_rid;
  std::string field;
  double gt_error = 0.01, min_callrate = 0.5;
  int min_mac = 1, max_alleles = 2;
  int64_t n_read = 0, n_skip = 0;
  int verbose = 10000;
  bool eof = false;

  int name2id(const std::string& c) const { auto it = contig_rid.find(c); return it == contig_rid.end() ? -1 : it->second; }

  void open(const std::string& path, const std::set<std::string>& sm_ids) {
    if (!in.open(path)) fatal("[E:%s] Cannot open VCF file %s", __func__, path.c_str());
    std::string line;
    bool have_header = false;
    dict.assign(1, "PASS");                      // the implicit first dictionary entry
    in.fill();
    is_bcf = in.cur_p->size() >= 5 && memcmp(in.cur_p->data(), "BCF\2\2", 5) == 0;
    if (is_bcf) {                                // BCF2: magic, l_text, the VCF header as text (NUL-terminated), then binary records
      uint8_t magic[5]; uint32_t l_text = 0;
      if (!in.read(magic, 5) || !in.read(&l_text, 4) || l_text > (1u << 30)) fatal("[E:%s] %s: truncated BCF header", __func__, path.c_str());
      std::string text((size_t)l_text, '\0');
      if (!in.read(&text[0], l_text)) fatal("[E:%s] %s: truncated BCF header", __func__, path.c_str());
      while (!text.empty() && text.back() == '\0') text.pop_back();
      for (std::string ln : split(text, '\n')) {
        if (!ln.empty() && ln.back() == '\r') ln.pop_back();
        parse_header_line(ln, have_header);
      }
    } else {
      while (in.getline(line)) {
        if (line.rfind("##", 0) != 0 && line.rfind("#CHROM", 0) != 0)
          fatal("[E:%s] %s does not look like a VCF or BCF file (CRAM and indexed access need htslib, which this build does not use)", __func__, path.c_str());
        parse_header_line(line, have_header);
        if (have_header) break;
      }
    }
    if (!have_header) fatal("[E:%s] No #CHROM header line in %s", __func__, path.c_str());
    if (!sm_ids.empty()) {                     // bcf_filtered_reader.cpp:107-124: iterate the std::set => sorted id order
      for (const std::string& id : sm_ids) {
        auto it = std::find(samples.begin(), samples.end(), id);
        if (it == samples.end()) fatal("[E:%s] Cannot find sample ID %s from the BCF file", __func__, id.c_str());
        sm_icols.push_back((int)(it - samples.begin()));
      }
    } else {
      for (size_t i = 0; i < samples.size(); ++i) sm_icols.push_back((int)i);
    }
    notice("Finished identifying %u samples to load from VCF/BCF", (unsigned)sm_icols.size());
    if (sm_icols.empty()) fatal("[E:%s] No sample to load from VCF/BCF", __func__);
  }
  int nsamples() const { return (int)sm_icols.size(); }
  const char* sample_id(int i) const { return samples[sm_icols[i]].c_str(); }

  // One record, decoded from VCF text or from BCF2 into the same shape; the filter and the a3 transforms see no difference.
  struct Rec {
    std::string chrom, ref; std::vector<std::string> alts;
    int64_t pos = 0;                             // 0-based
    bool has_gt = false, has_fld = false;
    std::vector<int32_t> alleles;                // [nv*2], -1 = missing
    std::vector<int32_t> pl;                     // [nv*3], INT32_MIN = missing   (field PL)
    std::vector<float> gp;                       // [nv*3], NaN = missing         (field GP)
  };

  // ---- BCF2 (the binary VCF; bgzip'd).  Typed values: descriptor byte = (length << 4) | type, length 15 = "a typed integer
  //      follows"; types 1/2/3 = int8/16/32, 5 = float, 7 = char.  Strings of FILTER/INFO/FORMAT keys and contig names are
  //      indices into dictionaries built from the header lines (implicit PASS = 0; IDX= overrides the running index).
  bool is_bcf = false;
  std::vector<std::string> dict, ctg_names;
  std::vector<uint8_t> rbuf;

  static int64_t typed_int(const uint8_t*& p, const uint8_t* end, int type) {
    if (type == 1) { if (p + 1 > end) return INT64_MIN; const int8_t x = (int8_t)*p; p += 1; return x; }
    if (type == 2) { if (p + 2 > end) return INT64_MIN; int16_t x; memcpy(&x, p, 2); p += 2; return x; }
    if (type == 3) { if (p + 4 > end) return INT64_MIN; int32_t x; memcpy(&x, p, 4); p += 4; return x; }
    return INT64_MIN;
  }
  // descriptor -> (type, length); false on truncation
  static bool typed_desc(const uint8_t*& p, const uint8_t* end, int& type, int64_t& len) {
    if (p >= end) return false;
    const uint8_t d = *p++;
    type = d & 15; len = d >> 4;
    if (len == 15) {
      if (p >= end) return false;
      const int t2 = *p++ & 15;
      len = typed_int(p, end, t2);
      if (len < 0) return false;
    }
    return true;
  }
  static int type_size(int type) { return type == 1 ? 1 : type == 2 ? 2 : type == 3 ? 4 : type == 5 ? 4 : type == 7 ? 1 : 0; }
  static bool typed_string(const uint8_t*& p, const uint8_t* end, std::string& out) {
    int type; int64_t len;
    if (!typed_desc(p, end, type, len)) return false;
    if (type == 0 && len == 0) { out.clear(); return true; }          // missing
    if (type != 7 || p + len > end) return false;
    out.assign((const char*)p, (size_t)len); p += len;
    return true;
  }
  static bool typed_skip(const uint8_t*& p, const uint8_t* end) {
    int type; int64_t len;
    if (!typed_desc(p, end, type, len)) return false;
    const int64_t n = len * type_size(type);
    if (p + n > end) return false;
    p += n;
    return true;
  }

  // IDX= of a BCF header dictionary line: a position in the string / contig dictionary.  A damaged header must not become an
  // allocation request: anything but a plain number below 2^24 is fatal.
  static size_t parse_idx(const std::string& idx, const std::string& line) {
    char* end = nullptr;
    errno = 0;
    const long long v = strtoll(idx.c_str(), &end, 10);
    if (errno || end == idx.c_str() || *end || v < 0 || v >= (1ll << 24)) fatal("[E:%s] bad IDX=%s in header line %.120s", __func__, idx.c_str(), line.c_str());
    return (size_t)v;
  }
  void parse_header_line(const std::string& line, bool& have_header) {
    auto attr = [&](const char* key) -> std::string {                // value of key= inside <...>
      const std::string k = std::string(key) + "=";
      size_t b = line.find("<" + k);
      if (b == std::string::npos) b = line.find("," + k);
      if (b == std::string::npos) return "";
      b += 1 + k.size();
      const size_t e = line.find_first_of(",>", b);
      return line.substr(b, e == std::string::npos ? std::string::npos : e - b);
    };
    if (line.rfind("##contig=<", 0) == 0) {
      const std::string id = attr("ID");
      if (!contig_rid.count(id)) { int r = (int)contig_rid.size(); contig_rid[id] = r; }
      const std::string idx = attr("IDX");
      const size_t at = idx.empty() ? ctg_names.size() : parse_idx(idx, line);
      if (ctg_names.size() <= at) ctg_names.resize(at + 1);
      ctg_names[at] = id;
    } else if (line.rfind("##FILTER=<", 0) == 0 || line.rfind("##INFO=<", 0) == 0 || line.rfind("##FORMAT=<", 0) == 0) {
      const std::string id = attr("ID");
      if (std::find(dict.begin(), dict.end(), id) != dict.end()) return;
      const std::string idx = attr("IDX");
      const size_t at = idx.empty() ? dict.size() : parse_idx(idx, line);
      if (dict.size() <= at) dict.resize(at + 1);
      dict[at] = id;
    } else if (line.rfind("#CHROM", 0) == 0) {
      auto f = split(line, '\t');
      for (size_t i = 9; i < f.size(); ++i) samples.push_back(f[i]);
      have_header = true;
    }
  }

  template <typename T> static T rd(const uint8_t* p) { T x; memcpy(&x, p, sizeof x); return x; }

  bool read_bcf(Rec& r) {
    uint32_t len[2];
    const int got = in.read_prefix(len, 8);
    if (got == 0) return false;
    if (got < 0) fatal("[E:%s] %s: truncated BCF record (the file ends inside a record's length prefix)", __func__, in.path.c_str());
    const size_t l_shared = len[0], l_indiv = len[1];
    if (l_shared < 24 || l_shared + l_indiv > ((size_t)1 << 31)) fatal("[E:%s] %s: corrupt BCF record (lengths %u, %u)", __func__, in.path.c_str(), len[0], len[1]);
    rbuf.resize(l_shared + l_indiv);
    if (!in.read(rbuf.data(), rbuf.size())) fatal("[E:%s] %s: truncated BCF record", __func__, in.path.c_str());
    const uint8_t* p = rbuf.data();
    const uint8_t* se = p + l_shared;
    const int32_t chrom = rd<int32_t>(p), pos = rd<int32_t>(p + 4);
    const uint32_t n_allele_info = rd<uint32_t>(p + 16), n_fmt_sample = rd<uint32_t>(p + 20);
    const int n_allele = (int)(n_allele_info >> 16), n_fmt = (int)(n_fmt_sample >> 24);
    const size_t n_sample = n_fmt_sample & 0xFFFFFFu;
    p += 24;
    if (chrom < 0 || (size_t)chrom >= ctg_names.size()) fatal("[E:%s] %s: BCF record names contig %d, the header has %u", __func__, in.path.c_str(), chrom, (unsigned)ctg_names.size());
    if (n_sample != samples.size()) fatal("[E:%s] %s: BCF record with %u samples, header with %u", __func__, in.path.c_str(), (unsigned)n_sample, (unsigned)samples.size());
    r.chrom = ctg_names[(size_t)chrom];
    r.pos = pos;
    std::string id;
    if (!typed_string(p, se, id)) fatal("[E:%s] %s: corrupt BCF record (ID)", __func__, in.path.c_str());
    r.alts.clear(); r.ref.clear();
    for (int a = 0; a < n_allele; ++a) {
      std::string al;
      if (!typed_string(p, se, al)) fatal("[E:%s] %s: corrupt BCF record (alleles)", __func__, in.path.c_str());
      if (a == 0) r.ref = al; else r.alts.push_back(al);
    }
    // FILTER and INFO are not used by the scan
    const int nv = nsamples();
    r.has_gt = r.has_fld = false;
    r.alleles.assign((size_t)nv * 2, -1);
    if (field == "PL") r.pl.assign((size_t)nv * 3, INT32_MIN);
    if (field == "GP") r.gp.assign((size_t)nv * 3, NAN);
    p = se;
    const uint8_t* ie = se + l_indiv;
    for (int k = 0; k < n_fmt; ++k) {
      int kt; int64_t kl;
      if (!typed_desc(p, ie, kt, kl) || kl != 1) fatal("[E:%s] %s: corrupt BCF record (FORMAT key)", __func__, in.path.c_str());
      const int64_t key = typed_int(p, ie, kt);
      int type; int64_t n;
      if (!typed_desc(p, ie, type, n)) fatal("[E:%s] %s: corrupt BCF record (FORMAT type)", __func__, in.path.c_str());
      const size_t esz = (size_t)type_size(type), stride = (size_t)n * esz;
      if (p + stride * n_sample > ie) fatal("[E:%s] %s: corrupt BCF record (FORMAT data)", __func__, in.path.c_str());
      const std::string& name = (key >= 0 && (size_t)key < dict.size()) ? dict[(size_t)key] : std::string();
      auto ival = [&](const uint8_t* q, bool& missing, bool& end_) -> int32_t {      // one integer element of this field
        missing = end_ = false;
        if (type == 1) { const int8_t x = (int8_t)*q; missing = x == INT8_MIN; end_ = x == INT8_MIN + 1; return x; }
        if (type == 2) { const int16_t x = rd<int16_t>(q); missing = x == INT16_MIN; end_ = x == INT16_MIN + 1; return x; }
        const int32_t x = rd<int32_t>(q); missing = x == INT32_MIN; end_ = x == INT32_MIN + 1; return x;
      };
      if (name == "GT" && type >= 1 && type <= 3) {
        r.has_gt = true;
        for (int i = 0; i < nv; ++i) {
          const uint8_t* q = p + stride * (size_t)sm_icols[i];
          for (int h = 0; h < 2 && h < n; ++h) {
            bool ms, en;
            const int32_t x = ival(q + esz * (size_t)h, ms, en);
            r.alleles[(size_t)2 * i + h] = (ms || en) ? -1 : (x >> 1) - 1;              // bcf_gt_allele
          }
        }
        if (field == "GT") r.has_fld = true;
      }
      if (name == field && field == "PL" && type >= 1 && type <= 3) {
        r.has_fld = true;
        for (int i = 0; i < nv; ++i) {
          const uint8_t* q = p + stride * (size_t)sm_icols[i];
          for (int g = 0; g < 3 && g < n; ++g) {
            bool ms, en;
            const int32_t x = ival(q + esz * (size_t)g, ms, en);
            if (en) break;
            r.pl[(size_t)i * 3 + g] = ms ? INT32_MIN : x;
          }
        }
      }
      if (name == field && field == "GP" && type == 5) {
        r.has_fld = true;
        for (int i = 0; i < nv; ++i) {
          const uint8_t* q = p + stride * (size_t)sm_icols[i];
          for (int g = 0; g < 3 && g < n; ++g) {
            const uint32_t bits = rd<uint32_t>(q + 4 * (size_t)g);
            if (bits == 0x7F800002u) break;                                          // end of vector
            r.gp[(size_t)i * 3 + g] = bits == 0x7F800001u ? NAN : rd<float>(q + 4 * (size_t)g);
          }
        }
      }
      p += stride * n_sample;
    }
    return true;
  }

  bool read_text(Rec& r) {
    std::string line;
    for (;;) {
      if (!in.getline(line)) return false;
      if (!line.empty() && line[0] != '#') break;
    }
    auto f = split(line, '\t');
    if (f.size() < 10) fatal("[E:%s] VCF record with %u columns at line starting %.40s", __func__, (unsigned)f.size(), line.c_str());
    r.chrom = f[0];
    r.pos = atoll(f[1].c_str()) - 1;
    r.ref = f[3];
    r.alts.clear();
    if (f[4] != ".") r.alts = split(f[4], ',');
    const auto keys = split(f[8], ':');
    int i_gt = -1, i_fld = -1;
    for (size_t k = 0; k < keys.size(); ++k) { if (keys[k] == "GT") i_gt = (int)k; if (keys[k] == field) i_fld = (int)k; }
    r.has_gt = i_gt >= 0; r.has_fld = i_fld >= 0;
    const int nv = nsamples();
    r.alleles.assign((size_t)nv * 2, -1);
    if (field == "PL") r.pl.assign((size_t)nv * 3, INT32_MIN);
    if (field == "GP") r.gp.assign((size_t)nv * 3, NAN);
    for (int i = 0; i < nv; ++i) {
      const auto sf = split(f[9 + sm_icols[i]], ':');
      static const std::string kMissingGt = ".";
      const std::string& gt = (i_gt >= 0 && i_gt < (int)sf.size()) ? sf[i_gt] : kMissingGt;
      // diploid GT "a/b" or "a|b"; '.' = missing allele (bcf_gt_allele < 0); haploid "a" leaves the second allele missing
      size_t sep = gt.find_first_of("/|");
      const std::string a1 = gt.substr(0, sep), a2 = sep == std::string::npos ? "." : gt.substr(sep + 1);
      auto al = [](const std::string& s) { return (s.empty() || s == ".") ? -1 : atoi(s.c_str()); };
      r.alleles[2 * i] = al(a1); r.alleles[2 * i + 1] = al(a2);
      if (i_fld >= 0 && i_fld < (int)sf.size() && field != "GT") {
        const auto pv = split(sf[i_fld], ',');
        for (int g = 0; g < 3 && g < (int)pv.size(); ++g) {
          if (field == "PL") r.pl[(size_t)i * 3 + g] = (pv[g] == "." ? INT32_MIN : atoi(pv[g].c_str()));
          else r.gp[(size_t)i * 3 + g] = (pv[g] == "." ? NAN : (float)atof(pv[g].c_str()));
        }
      }
    }
    return true;
  }

  // next variant that passes the filter (bcf_filtered_reader.cpp:751-764 + passed_vfilter :498-574); false at EOF
  bool read(Variant& v) {
    Rec r;
    while (is_bcf ? read_bcf(r) : read_text(r)) {
      ++n_read;
      if (verbose > 0 && n_read % verbose == 0) notice("Reading %lld variants at %s:%lld, Skipping %lld, Missing 0.", (long long)n_read, r.chrom.c_str(), (long long)r.pos + 1, (long long)n_skip);
      if (!contig_rid.count(r.chrom)) { int rr = (int)contig_rid.size(); contig_rid[r.chrom] = rr; }     // headers without ##contig lines
      v.rid = contig_rid[r.chrom];
      v.pos = r.pos;
      v.ref = r.ref;
      v.alt.clear();
      for (size_t a = 0; a < r.alts.size(); ++a) { if (a) v.alt.push_back(','); v.alt += r.alts[a]; }
      if (r.alts.empty()) v.alt = ".";
      v.rlen = (int)v.ref.size();
      v.n_allele = 1 + (int)r.alts.size();
      v.ref0 = v.ref.empty() ? 'N' : v.ref[0];
      v.alt0 = (v.n_allele > 1 && !r.alts[0].empty()) ? r.alts[0][0] : '.';
      // passed_vfilter returns true before ANY check when neither --min-mac nor --min-callrate asks for genotypes
      // (require_GT = false, bcf_filtered_reader.cpp:507; cmd_cram_demuxlet.cpp:104-105 sets it from those two options)
      const bool require_gt = (min_mac > 0) || (min_callrate > 0);
      // (multi-allelic records stay skipped in that case too: the reference lets them through and then reads its 6-genotype
      //  layout as if it had 3 per sample, cmd_cram_demuxlet.cpp:227-231 — garbage this build does not reproduce)
      if (v.n_allele > max_alleles) { ++n_skip; continue; }                                    // :534
      if (!r.has_gt && (require_gt || field == "GT")) fatal("[E:%s] Cannot find the field GT from the VCF file at position %s:%lld", __func__, r.chrom.c_str(), (long long)v.pos + 1);   // :548-549
      const int nv = nsamples();
      int an = 0; std::vector<int> acs((size_t)std::max(v.n_allele, 2), 0);
      for (int i = 0; i < nv; ++i)
        for (int h = 0; h < 2; ++h) { const int x = r.alleles[2 * i + h]; if (x >= 0 && x < (int)acs.size()) { ++an; ++acs[x]; } }   // :230-240
      if (require_gt && min_callrate > (double)an / (2.0 * (double)nv)) { ++n_skip; continue; }   // :554
      const int ac = an - acs[0];
      if (require_gt && ((ac < min_mac) || (an - ac < min_mac))) { ++n_skip; continue; }       // :565
      // parse_posteriors (:360-454) through the library's a3 transforms
      v.gps.assign((size_t)nv * 3, 0.f);
      if (field == "GT") {
        if (dmx_geno_from_gt(r.alleles.data(), nv, gt_error, v.gps.data()) != DMX_OK) fatal("[E:%s] %s", __func__, dmx_last_error());
      } else {
        if (!r.has_fld) fatal("[E:%s] Cannot parse posterior probability at %s:%lld", __func__, r.chrom.c_str(), (long long)v.pos + 1);   // :154, :212
        if (field == "PL") {
          if (dmx_geno_from_pl(r.pl.data(), nv, v.gps.data()) != DMX_OK) fatal("[E:%s] %s", __func__, dmx_last_error());
        } else {
          if (dmx_geno_from_gp(r.gp.data(), nv, gt_error, v.gps.data()) != DMX_OK) fatal("[E:%s] %s", __func__, dmx_last_error());
        }
      }
      return true;
    }
    eof = true;
    return false;
  }
};

// ---- f2: SAM / BAM reader -------------------------------------------------------------------------------------------------
struct Read {
  // One alignment.  The record's bytes stay where the reader put them (a reused buffer): sequence and qualities are
  // decoded per queried base (a read overlaps 0-2 SNPs), names and tags are views into the record.
  std::string cb, ub;           // group / UMI tag values of a SAM line (not NUL-terminated there: copied) ...
  const char* cb_p = ""; const char* ub_p = "";   // ... and where they are: in cb / ub (SAM) or inside the BAM record (NUL-terminated Z fields)
  size_t cb_n = 0, ub_n = 0;
  bool has_cb = false, has_ub = false;
  int64_t endpos_c = 0;         // SamReader::endpos(*this), and the barcode as (worker slot, id in that slot's dictionary): filled by the
  int32_t cb_slot = -1, cb_lid = -1;   // parallel record parsing of the windowed scan, so that the in-order stage need not hash strings
  const uint8_t* rec = nullptr; // the BAM record's bytes: where the inflater wrote them (SamReader::next_raw's `keep` holds that
  size_t rec_n = 0;             // chunk alive), or in `own` for a record that straddles two chunks / ...
  std::vector<uint8_t> own;
  std::string own_line;         // ... the SAM line the views below point into (a window of reads is parsed and overlapped with
                                // the SNPs on several host threads while the reader moves on)
  int flag = 0, tid = -1, mapq = 0;
  int64_t pos = 0;              // 0-based
  std::vector<std::pair<char, uint32_t>> cigar;
  int l_qseq = 0;
  const char* qname_p = ""; size_t qname_n = 0;
  const uint8_t* seq4 = nullptr;   // BAM: 4-bit packed bases
  const char* seq_txt = nullptr;   // SAM: text bases
  const uint8_t* qual_raw = nullptr;  // BAM: phred values; SAM: phred + 33; nullptr = "*" (0xff)
  bool qual_is_text = false;
  std::string qname() const { return std::string(qname_p, qname_n); }
  char base(size_t i) const {      // htslib stores 4-bit codes and prints "=ACMGRSVTWYHKDBN"
    if (seq4) return "=ACMGRSVTWYHKDBN"[(seq4[i >> 1] >> ((i & 1) ? 0 : 4)) & 0xf];
    const char ch = (char)toupper((unsigned char)seq_txt[i]);
    return strchr("=ACMGRSVTWYHKDBN", ch) && ch ? ch : 'N';
  }
  char qual33(size_t i) const {    // quality character as the reference sees it (phred + 33)
    if (!qual_raw) return (char)(0xff + 33);
    return qual_is_text ? (char)qual_raw[i] : (char)(qual_raw[i] + 33);
  }
};

struct SamReader {
  GzIn in;
  bool is_bam = false;
  std::vector<std::string> targets;
  std::map<std::string, int> target_id;
  std::string pending;          // first alignment line of a SAM text file
  bool have_pending = false;
  char gtag[3] = {0, 0, 0}, utag[3] = {0, 0, 0};
  int min_mq = 20, excl_flag = 0x0f04, verbose = 1000000;
  int64_t n_read = 0, n_skip = 0;

  void open(const std::string& path) {
    if (!in.open(path)) fatal("[E:%s] Cannot open SAM/BAM file %s", __func__, path.c_str());
    char magic[4] = {0, 0, 0, 0};
    const int got = in.read(magic, 4) ? 4 : 0;
    if (got == 4 && memcmp(magic, "BAM\1", 4) == 0) {
      is_bam = true;
      int32_t l_text = 0, n_ref = 0;
      if (!in.read(&l_text, 4)) fatal("[E:%s] truncated BAM header", __func__);
      std::string text((size_t)l_text, 0);
      if (l_text && !in.read(&text[0], (size_t)l_text)) fatal("[E:%s] truncated BAM header", __func__);
      if (!in.read(&n_ref, 4)) fatal("[E:%s] truncated BAM header", __func__);
      for (int i = 0; i < n_ref; ++i) {
        int32_t l_name = 0, l_ref = 0;
        in.read(&l_name, 4);
        std::string nm((size_t)l_name, 0);
        in.read(&nm[0], (size_t)l_name); in.read(&l_ref, 4);
        nm.resize(strlen(nm.c_str()));
        target_id[nm] = (int)targets.size(); targets.push_back(nm);
      }
    } else {
      if (got == 4 && memcmp(magic, "CRAM", 4) == 0) fatal("[E:%s] CRAM input needs htslib, which this build does not use", __func__);
      in.unread_all_to(0);                       // the 4 sniffed bytes are inside the first chunk
      std::string line;
      while (in.getline(line)) {
        if (!line.empty() && line[0] == '@') {
          if (line.rfind("@SQ", 0) == 0) {
            for (const std::string& fld : split(line, '\t'))
              if (fld.rfind("SN:", 0) == 0) { const std::string nm = fld.substr(3); target_id[nm] = (int)targets.size(); targets.push_back(nm); }
          }
        } else { pending = line; have_pending = true; break; }
      }
    }
  }

  static int64_t endpos(const Read& r) {        // bam_endpos: pos + reference length of the CIGAR (M, D, N, =, X), at least 1
    int64_t rl = 0;
    if (!(r.flag & 4)) for (auto& op : r.cigar) if (op.first == 'M' || op.first == 'D' || op.first == 'N' || op.first == '=' || op.first == 'X') rl += op.second;
    return r.pos + (rl > 0 ? rl : 1);
  }

  bool parse_sam_line(const std::string& line, Read& r) const {
    // fields in place: [b[i], b[i+1]-1) without copying
    const char* p = line.data();
    const char* const end = p + line.size();
    const char* fb[12]; size_t fn[12];
    int nf = 0;
    const char* q = p;
    while (nf < 11) {
      const char* t = (const char*)memchr(q, '\t', (size_t)(end - q));
      fb[nf] = q; fn[nf] = (size_t)((t ? t : end) - q); ++nf;
      if (!t) { q = end; break; }
      q = t + 1;
    }
    if (nf < 11) fatal("[E:%s] SAM record with %u fields", __func__, (unsigned)nf);
    auto eq = [&](int i, const char* lit) { return fn[i] == strlen(lit) && memcmp(fb[i], lit, fn[i]) == 0; };
    r.qname_p = fb[0]; r.qname_n = fn[0];
    r.flag = atoi(fb[1]);
    if (eq(2, "*")) r.tid = -1;
    else {
      static thread_local std::string last_rname;                                           // reads come sorted: one lookup per contig
      static thread_local int last_tid = -1;                                                // (and thread; there is one SamReader per process)
      if (fn[2] != last_rname.size() || memcmp(fb[2], last_rname.data(), fn[2]) != 0) {
        last_rname.assign(fb[2], fn[2]);
        auto it = target_id.find(last_rname);
        last_tid = it == target_id.end() ? -1 : it->second;
      }
      r.tid = last_tid;
    }
    r.pos = atoll(fb[3]) - 1; r.mapq = atoi(fb[4]);
    r.cigar.clear();
    if (!eq(5, "*")) {
      uint32_t n = 0;
      for (size_t i = 0; i < fn[5]; ++i) { const char ch = fb[5][i]; if (ch >= '0' && ch <= '9') n = n * 10 + (uint32_t)(ch - '0'); else { r.cigar.emplace_back(ch, n); n = 0; } }
    }
    r.seq4 = nullptr;
    if (eq(9, "*")) { r.seq_txt = ""; r.l_qseq = 0; } else { r.seq_txt = fb[9]; r.l_qseq = (int)fn[9]; }
    if (eq(10, "*")) r.qual_raw = nullptr; else { r.qual_raw = (const uint8_t*)fb[10]; r.qual_is_text = true; }
    r.has_cb = r.has_ub = false;
    while (q < end) {                             // optional fields TAG:TYPE:VALUE
      const char* t = (const char*)memchr(q, '\t', (size_t)(end - q));
      const char* fe = t ? t : end;
      if (fe - q >= 5 && q[2] == ':' && q[4] == ':' && q[3] == 'Z') {
        if (gtag[0] && q[0] == gtag[0] && q[1] == gtag[1]) { r.cb.assign(q + 5, (size_t)(fe - q - 5)); r.has_cb = true; }
        if (utag[0] && q[0] == utag[0] && q[1] == utag[1]) { r.ub.assign(q + 5, (size_t)(fe - q - 5)); r.has_ub = true; }
      }
      if (!t) break;
      q = t + 1;
    }
    r.cb_p = r.cb.c_str(); r.cb_n = r.cb.size(); r.ub_p = r.ub.c_str(); r.ub_n = r.ub.size();
    return true;
  }

  // the next record, unparsed: r.rec / r.rec_n (BAM) or r.own_line (SAM); false at EOF.  A BAM record that lies inside one
  // inflated chunk is not copied: r.rec points into the chunk, which stays alive until the reader's next call — or, when the
  // caller collects records (windowed scan), for as long as `keep` holds it.
  bool next_raw(Read& r, std::vector<ChunkP>* keep = nullptr) {
    if (is_bam) {
      int32_t block = 0;
      if (in.pos + 4 <= in.cur_p->size()) {
        memcpy(&block, in.cur_p->data() + in.pos, 4);
        if (block < 32) fatal("[E:%s] corrupt BAM record (block size %d)", __func__, block);
        if (in.pos + 4 + (size_t)block <= in.cur_p->size()) {
          r.rec = in.cur_p->data() + in.pos + 4; r.rec_n = (size_t)block;
          in.pos += 4 + (size_t)block;
          if (keep && (keep->empty() || keep->back() != in.cur_p)) keep->push_back(in.cur_p);
          return true;
        }
      }
      const int got = in.read_prefix(&block, 4);
      if (got == 0) return false;
      if (got < 0) fatal("[E:%s] truncated BAM record (the file ends inside a record's block size)", __func__);
      if (block < 32) fatal("[E:%s] corrupt BAM record (block size %d)", __func__, block);
      r.own.resize((size_t)block);
      if (!in.read(r.own.data(), (size_t)block)) fatal("[E:%s] truncated BAM record", __func__);
      r.rec = r.own.data(); r.rec_n = (size_t)block;
      return true;
    }
    for (;;) {
      if (have_pending) { r.own_line.swap(pending); have_pending = false; }
      else if (!in.getline(r.own_line)) return false;
      if (!r.own_line.empty()) return true;
    }
  }
  // the fields of the record next_raw left in r (no reader state is touched: callable from several threads for different reads)
  void parse_raw(Read& r) const { if (is_bam) parse_bam_record(r); else parse_sam_line(r.own_line, r); }

  void parse_bam_record(Read& r) const {
    struct Bytes { const uint8_t* p; size_t n; const uint8_t& operator[](size_t i) const { return p[i]; } size_t size() const { return n; } };
    const Bytes b{r.rec, r.rec_n};
    auto i32 = [&](size_t o) { int32_t v; memcpy(&v, &b[o], 4); return v; };
    auto u16 = [&](size_t o) { uint16_t v; memcpy(&v, &b[o], 2); return v; };
    r.tid = i32(0); r.pos = i32(4);
    const int l_read_name = b[8]; r.mapq = b[9];
    const int n_cigar = u16(12); r.flag = u16(14);
    r.l_qseq = i32(16);
    size_t o = 32;
    if (o + (size_t)l_read_name + 4 * (size_t)n_cigar + (size_t)(r.l_qseq + 1) / 2 + (size_t)r.l_qseq > b.size()) fatal("[E:%s] corrupt BAM record", __func__);
    r.qname_p = (const char*)&b[o]; r.qname_n = (size_t)std::max(0, l_read_name - 1); o += (size_t)l_read_name;
    r.cigar.clear();
    for (int i = 0; i < n_cigar; ++i) { uint32_t c; memcpy(&c, &b[o], 4); o += 4; r.cigar.emplace_back("MIDNSHP=XB"[std::min<uint32_t>(c & 0xf, 9)], c >> 4); }
    r.seq4 = &b[o]; r.seq_txt = nullptr;
    o += (size_t)(r.l_qseq + 1) / 2;
    r.qual_raw = &b[o]; r.qual_is_text = false;
    o += (size_t)r.l_qseq;
    r.has_cb = r.has_ub = false;
    while (o + 3 <= b.size()) {                 // aux fields
      const char t0 = (char)b[o], t1 = (char)b[o + 1], ty = (char)b[o + 2];
      o += 3;
      size_t len = 0;
      if (ty == 'Z' || ty == 'H') {
        const char* s = (const char*)&b[o];
        const void* nul = o < b.size() ? memchr(s, 0, b.size() - o) : nullptr;          // the string must end inside the record
        if (!nul) fatal("[E:%s] corrupt BAM record: unterminated %c%c:%c aux field", __func__, t0, t1, ty);
        len = (size_t)((const char*)nul - s) + 1;
        if (ty == 'Z') {
          if (gtag[0] && t0 == gtag[0] && t1 == gtag[1]) { r.cb_p = s; r.cb_n = len - 1; r.has_cb = true; }
          if (utag[0] && t0 == utag[0] && t1 == utag[1]) { r.ub_p = s; r.ub_n = len - 1; r.has_ub = true; }
        }
      } else if (ty == 'A' || ty == 'c' || ty == 'C') len = 1;
      else if (ty == 's' || ty == 'S') len = 2;
      else if (ty == 'i' || ty == 'I' || ty == 'f') len = 4;
      else if (ty == 'B') {
        if (o + 5 > b.size()) fatal("[E:%s] corrupt BAM record: truncated B aux field", __func__);
        const char sub = (char)b[o]; int32_t cnt; memcpy(&cnt, &b[o + 1], 4);
        if (cnt < 0) fatal("[E:%s] corrupt BAM record: negative B array length", __func__);
        const size_t es = (sub == 'c' || sub == 'C') ? 1 : ((sub == 's' || sub == 'S') ? 2 : 4);
        len = 5 + es * (size_t)cnt;
      } else fatal("[E:%s] unknown BAM aux type %c", __func__, ty);
      if (len > b.size() - o) fatal("[E:%s] corrupt BAM record: aux field %c%c runs past the record", __func__, t0, t1);
      o += len;
    }
  }

  // counts a parsed read and applies the read filter (sam_filtered_reader.cpp:233-258, passed_filter :284-296)
  bool count_and_filter(const Read& r) {
    ++n_read;
    if (verbose > 0 && n_read % verbose == 0) notice("Reading %lld reads at %s:%lld and skipping %lld", (long long)n_read, r.tid >= 0 ? targets[r.tid].c_str() : "*", (long long)r.pos + 1, (long long)n_skip);
    if (r.mapq < min_mq || (excl_flag & r.flag)) { ++n_skip; return false; }
    return true;
  }
  // next read passing the filter; false at EOF
  bool read(Read& r) {
    for (;;) {
      if (!next_raw(r)) return false;
      parse_raw(r);
      if (count_and_filter(r)) return true;
    }
  }
};

// CIGAR walk: base / quality / read offset of the read at reference position pos (hts_utils.cpp:279-359):
// only M, D/N and S/I move the cursors ('=', 'X', 'H', 'P' are ignored, as in the reference)
constexpr int kNA = -1;
void base_at(const Read& r, int64_t pos, char& base, char& qual, int& rpos) {
  const int rlen = r.l_qseq;
  int64_t cpos = r.pos;
  int64_t rp = 0;
  base = 'N'; qual = 0;
  if (!r.cigar.empty()) {
    for (auto& op : r.cigar) {
      const int64_t len = op.second;
      if (op.first == 'M') {
        if (pos >= cpos && pos <= cpos + len - 1) { rp += pos - cpos; break; }
        cpos += len; rp += len;
      } else if (op.first == 'D' || op.first == 'N') {
        if (pos >= cpos && pos <= cpos + len - 1) { rp = -1; break; }
        cpos += len;
      } else if (op.first == 'S' || op.first == 'I') {
        rp += len;
      }
    }
    if (rp >= 0 && rp <= rlen) {
      if (rp < rlen) { base = r.base((size_t)rp); qual = r.qual33((size_t)rp); } else { base = 0; qual = 0; }
    } else {
      rp = kNA;
    }
  }
  if (rp >= rlen) { rp = kNA; base = '.'; }
  rpos = (int)rp;
}

}  // namespace

namespace {
struct Stopwatch {             // DMX_CLI_TIMING=1: where the scan's wall-clock goes (reported as NOTICE lines)
  bool on = getenv("DMX_CLI_TIMING") != nullptr;
  double acc[6] = {0, 0, 0, 0, 0, 0};
  std::chrono::steady_clock::time_point t0;
  void start() { if (on) t0 = std::chrono::steady_clock::now(); }
  void stop(int i) { if (on) acc[i] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
};
}  // namespace

int main(int argc, char** argv) {
  Options o;
  parse_options(argc, argv, o);
  if (o.alpha.empty()) { o.alpha.push_back(0); o.alpha.push_back(0.5); }                                   // cmd_cram_demuxlet.cpp:78-90

  std::set<std::string> bcd_set;
  if (!o.group_list.empty()) {                                                                            // :92-99
    GzIn g;
    if (!g.open(o.group_list)) fatal("[E:%s] Cannot open %s", __func__, o.group_list.c_str());
    std::string line;
    while (g.getline(line)) if (!line.empty()) bcd_set.insert(split(line, '\t')[0].substr(0, line.find_first_of(" \t")));
    notice("Finished loading %u droplet/cell barcodes to consider", (unsigned)bcd_set.size());
  }
  std::set<std::string> sm_ids(o.sm.begin(), o.sm.end());                                                  // :101-103
  if (!o.sm_list.empty()) {
    GzIn g;
    if (!g.open(o.sm_list)) fatal("[E:%s] Cannot open %s", __func__, o.sm_list.c_str());
    std::string line;
    while (g.getline(line)) if (!line.empty()) sm_ids.insert(line.substr(0, line.find_first_of(" \t")));
    notice("Finished loading %u IDs from %s", (unsigned)sm_ids.size(), o.sm_list.c_str());
  }
  if (o.vcf.empty()) fatal("[%s] bcf_file_name is empty", __func__);
  if (o.sam.empty()) fatal("[%s] sam_file_name is empty", __func__);

  VcfReader vr;
  vr.field = o.field; vr.gt_error = o.geno_error; vr.min_callrate = o.min_callrate; vr.min_mac = o.min_mac; vr.verbose = o.vcf_verbose;
  vr.open(o.vcf, sm_ids);
  SamReader sr;
  sr.min_mq = o.min_mq; sr.excl_flag = o.excl_flag; sr.verbose = o.sam_verbose;
  if (o.out.empty()) fatal("[E:%s] --out parameter is missing", __func__);                                 // :116-117
  if (!o.tag_group.empty()) {                                                                              // :122-140
    if (o.tag_group.size() != 2) fatal("[E:%s] Cannot recognize group tag %s. It is suppose to be a length 2 string", __func__, o.tag_group.c_str());
    sr.gtag[0] = o.tag_group[0]; sr.gtag[1] = o.tag_group[1];
  }
  if (!o.tag_umi.empty()) {
    if (o.tag_umi.size() != 2) fatal("[E:%s] Cannot recognize UMI tag %s. It is suppose to be a length 2 string", __func__, o.tag_umi.c_str());
    sr.utag[0] = o.tag_umi[0]; sr.utag[1] = o.tag_umi[1];
  }
  sr.open(o.sam);

  dmx_store* scl = dmx_store_new();
  if (!scl) fatal("%s", dmx_last_error());
  struct Snp { int rid; int64_t pos; int rlen; char ref, alt; };
  std::vector<Snp> snps;
  std::vector<float> G;

  Variant cur;
  if (!vr.read(cur)) fatal("[E:%s Cannot read any single variant from %s]", __func__, o.vcf.c_str());      // :150-151
  // chromosome order must agree between the two files (:157-178)
  {
    int prevrid = -1, nmatch = 0;
    std::string prevchrom;
    for (const std::string& chrom : sr.targets) {
      const int rid = vr.name2id(chrom);
      if (rid >= 0) {
        if (prevrid >= rid) fatal("[E:%s] Your VCF/BCF files and SAM/BAM/CRAM files have different ordering of chromosomes. SAM/BAM/CRAM file has %s before %s, but VCF/BCF file has %s after %s", __func__, prevchrom.c_str(), chrom.c_str(), prevchrom.c_str(), chrom.c_str());
        prevrid = rid; prevchrom = chrom; ++nmatch;
      }
    }
    if (nmatch == 0 && !vr.contig_rid.empty() && vr.contig_rid.size() > 1)
      fatal("[E:%s] Your VCF/BCF files and SAM/BAM/CRAM files does not have any matching chromosomes, or some chromosome names are duplicated", __func__);
  }
  const int nv = vr.nsamples();
  auto add_snp = [&](const Variant& v) {                                                                   // :180-185, :227-232
    G.insert(G.end(), v.gps.begin(), v.gps.end());
    snps.push_back({v.rid, v.pos, v.rlen, v.ref0, v.alt0});
    return dmx_store_add_snp(scl);
  };
  add_snp(cur);
  int64_t ibeg = 0, nbuf = 1;
  bool veof = false;
  long nReadsMultiSNPs = 0, nReadsSkipBCD = 0, nReadsPass = 0, nReadsRedundant = 0, nReadsN = 0, nReadsLQ = 0, nReadsTMP = 0, nNonBiallelic = 0;
  int n_warn_g = 0, n_warn_u = 0;

  // ---- the scan (cmd_cram_demuxlet.cpp:195-338).  Per read, in BAM order: (S) the lock-step bookkeeping of the reference — SNP buffer
  // window, VCF reads, barcode and UMI, RD.TOTL — and (O) the overlap of the read with the buffered SNPs (hts_utils.cpp:279-359), whose
  // observations go into the store.  With one host thread both happen read by read.  With several, reads are taken in windows: their
  // records are parsed on all threads, (S) runs over the window in order, (O) runs on all threads, and the window's observations enter
  // the store in BAM order through dmx_store_add_batch (cell-sharded, so "first UMI wins" sees the same order); the VCF is parsed ahead
  // on its own thread.  Every counter, id and stored byte is the one-thread path's (tests/test_cli_cpu.py compares the dumps).
  std::map<std::string, int> contig_seen = vr.contig_rid;      // the VCF contigs known so far AS THE READS SEE THEM: a VCF without ##contig
  std::vector<int> tid_rid(sr.targets.size());                 // lines teaches its contigs record by record, and a read whose contig has not
  size_t tid_rid_for = (size_t)-1;                             // appeared yet is skipped (:198-200) — also when the VCF is parsed ahead
  const int n_threads = cli_threads();
  const bool windowed = n_threads > 1 && !getenv("DMX_SCAN_SEQUENTIAL");
  Stopwatch sw;
  // the HIP context(s) the job will run on come up while the scan runs (0.12-0.13 s that dmx_demuxlet_run would otherwise wait for);
  // a failure here is not reported: dmx_demuxlet_run meets the same condition and reports it
  if (!o.pileup_only && !getenv("DMX_NO_WARM_UP")) g_warm_up = std::thread([&o] { dmx_device_warm_up(o.gpu, o.gpus); });
  struct WarmGuard { ~WarmGuard() { if (g_warm_up.joinable()) g_warm_up.join(); } } warm_guard;
  const std::chrono::steady_clock::time_point scan_t0 = std::chrono::steady_clock::now();

  // variants in file order, parsed ahead by a producer thread in the windowed mode; each comes with the contig names its vr.read call registered
  struct Fed { Variant v; std::vector<std::pair<std::string, int>> new_contigs; bool eof = false; };
  std::mutex feed_mu; std::condition_variable feed_cv_put, feed_cv_get; std::deque<std::vector<Fed>> feed_q; bool feed_stop = false;   // (batches of 256: one lock per batch)
  std::vector<Fed> feed_cur; size_t feed_i = 0;
  std::thread feed_th;
  auto read_variant = [&](Fed& f) {                            // one vr.read call + what it did to the contig dictionary
    const size_t before = vr.contig_rid.size();
    f.eof = !vr.read(f.v);
    f.new_contigs.clear();
    if (vr.contig_rid.size() != before) for (const auto& kv : vr.contig_rid) if ((size_t)kv.second >= before) f.new_contigs.push_back(kv);
  };
  if (windowed) feed_th = std::thread([&] {
    for (;;) {
      std::vector<Fed> batch;
      batch.reserve(256);
      bool last = false;
      while (batch.size() < 256 && !last) { batch.emplace_back(); read_variant(batch.back()); last = batch.back().eof; }
      { std::unique_lock<std::mutex> lk(feed_mu); feed_cv_put.wait(lk, [&] { return feed_q.size() < 64 || feed_stop; }); if (feed_stop) return; feed_q.push_back(std::move(batch)); }
      feed_cv_get.notify_one();
      if (last) return;
    }
  });
  struct FeedGuard { std::mutex& mu; std::condition_variable& cv; bool& stop; std::thread& th; ~FeedGuard() { { std::lock_guard<std::mutex> lk(mu); stop = true; } cv.notify_all(); if (th.joinable()) th.join(); } } feed_guard{feed_mu, feed_cv_put, feed_stop, feed_th};
  auto next_variant = [&](Variant& v) -> bool {                // the reference's vr.read at :211
    Fed f;
    if (windowed) {
      if (feed_i >= feed_cur.size()) {
        { std::unique_lock<std::mutex> lk(feed_mu); feed_cv_get.wait(lk, [&] { return !feed_q.empty(); }); feed_cur = std::move(feed_q.front()); feed_q.pop_front(); }
        feed_cv_put.notify_one();
        feed_i = 0;
      }
      f = std::move(feed_cur[feed_i++]);
    } else read_variant(f);
    for (auto& kv : f.new_contigs) contig_seen.insert(kv);
    if (f.eof) return false;
    v = std::move(f.v);
    return true;
  };
  // (contig_seen was copied from vr.contig_rid above, after the first record was read and BEFORE the feed thread started: from here on
  // the feed thread owns vr — contigs reach this thread only through Fed::new_contigs)

  std::vector<std::vector<int32_t>>* slot_cell_p = nullptr;     // (windowed scan: see slot_cell below)
  struct Hit { int32_t snp; uint8_t allele, bq; };
  struct Staged { int64_t ibeg = 0, nbuf = 0; int32_t ibcd = 0; bool used = false, umi_in_read = false; std::string umi; std::vector<Hit> hits; int nv_valid = 0; };   // umi_in_read: the UMI is the read's own tag value (windowed scan: not copied)
  // (S) for one parsed read that passed the read filter; false = the read contributes nothing (:198-200, :263)
  auto stage = [&](const Read& rd, Staged& st) -> bool {
    st.used = false;
    if (tid_rid_for != contig_seen.size()) {
      for (size_t t = 0; t < sr.targets.size(); ++t) { auto it = contig_seen.find(sr.targets[t]); tid_rid[t] = it == contig_seen.end() ? -1 : it->second; }
      tid_rid_for = contig_seen.size();
    }
    const int64_t endpos = windowed ? rd.endpos_c : SamReader::endpos(rd);
    const int tid2rid = (rd.tid >= 0 && (size_t)rd.tid < tid_rid.size()) ? tid_rid[(size_t)rd.tid] : -1;
    if (tid2rid < 0) return false;                                                                         // :198-200
    {   // clear_buffer_before(chrom, read start) — bcf_filtered_reader.cpp:649-669
      int64_t n_rm = 0;
      for (int64_t i = 0; i < nbuf; ++i) {
        const Snp& v = snps[(size_t)(ibeg + i)];
        if (v.rid < tid2rid) ++n_rm;
        else if (v.rid == tid2rid && v.pos + v.rlen < rd.pos) ++n_rm;
        else break;
      }
      nbuf -= n_rm; ibeg += n_rm;
    }
    while (!veof && (snps.back().rid < tid2rid || (snps.back().rid == tid2rid && snps.back().pos < endpos))) {    // :209
      Variant v;
      sw.start();
      const bool got = next_variant(v);
      sw.stop(1);
      if (got) {
        if (v.rlen > 1 || v.n_allele != 2 || v.ref.size() > 1) {                                           // :215-225 (warn only)
          if (nNonBiallelic < 10) warning("VCF record must be biallelic SNPs. Ignoring non-SNPs and/or multi-allelic variants at %d:%lld", v.rid, (long long)v.pos + 1);
          ++nNonBiallelic;
          if (nNonBiallelic == 10) warning("Suppressing 10+ warnings of the same kind (non-SNP or multi-alleic variants)");
        }
        add_snp(v); ++nbuf;
      } else {
        veof = true;
      }
    }
    // barcode (:239-269)
    int32_t ibcd = 0;
    if (o.tag_group.empty()) {
      ibcd = dmx_store_add_cell(scl, ".");
    } else {
      const char* sbcd = ".";
      if (rd.has_cb) sbcd = rd.cb_p;
      else {
        if (n_warn_g < 10) notice("WARNING: Cannot find Droplet/Cell tag %s from %lld-th read %s at %s:%lld-%lld. Treating all of them as a single group", o.tag_group.c_str(), (long long)sr.n_read, rd.qname().c_str(), sr.targets[(size_t)rd.tid].c_str(), (long long)rd.pos, (long long)endpos);
        else if (n_warn_g == 10) notice("WARNING: Suppressing 10+ missing Droplet/Cell tag warnings...");
        ++n_warn_g;
      }
      int32_t* known = nullptr;                  // windowed scan: what this (worker slot, barcode) turned out to be at its first read
      if (rd.cb_slot >= 0 && slot_cell_p) {
        std::vector<int32_t>& sc = (*slot_cell_p)[(size_t)rd.cb_slot];
        if ((size_t)rd.cb_lid >= sc.size()) sc.resize((size_t)rd.cb_lid + 1 + sc.size() / 2, -1);
        known = &sc[(size_t)rd.cb_lid];
      }
      if (known && *known >= 0) ibcd = *known;
      else if (known && *known == -2) { ++nReadsSkipBCD; return false; }
      else if (bcd_set.empty() || bcd_set.count(sbcd)) {
        ibcd = dmx_store_add_cell(scl, sbcd);
        const int32_t nb = dmx_store_n_cells(scl);
        if (ibcd + 1 == nb && nb % 1000 == 0) notice("Observed %d droplets with unique cell barcode", nb);
        if (known) *known = ibcd;
      } else { if (known) *known = -2; ++nReadsSkipBCD; return false; }
    }
    ++nReadsTMP;
    // UMI (:272-293)
    st.umi_in_read = false;
    if (o.tag_umi.empty()) { char b[32]; snprintf(b, sizeof b, "%x", rand()); st.umi.assign("."); st.umi += b; }
    else if (rd.has_ub) { if (windowed) st.umi_in_read = true; else st.umi.assign(rd.ub_p, rd.ub_n); }
    else {
      st.umi.assign(".");
      if (n_warn_u < 10) notice("WARNING: Cannot find UMI tag %s from %lld-th read %s at %s:%lld-%lld. Treating all of them as a single UMI", o.tag_umi.c_str(), (long long)sr.n_read, rd.qname().c_str(), sr.targets[(size_t)rd.tid].c_str(), (long long)rd.pos, (long long)endpos);
      else if (n_warn_u == 10) notice("WARNING: Suppressing 10+ UMI warnings...");
      ++n_warn_u;
    }
    dmx_store_count_read(scl, ibcd);                                                                       // :295
    st.ibeg = ibeg; st.nbuf = nbuf; st.ibcd = ibcd; st.used = true;
    return true;
  };
  // (O) the read against the SNPs its buffer held (:306-329): what it would hand to add_read, and how many SNPs gave a base at all
  auto overlap = [&](const Read& rd, Staged& st, const Snp* snps) {     // snps[i] = SNP i (the list itself, or a window's snapshot of its span)
    st.hits.clear(); st.nv_valid = 0;
    for (int64_t i = st.ibeg; i < st.ibeg + st.nbuf; ++i) {                                                // :306
      char base, qual; int rpos;
      base_at(rd, snps[(size_t)i].pos, base, qual, rpos);
      if (rpos == kNA) continue;
      if (base == 'N') continue;
      ++st.nv_valid;
      if (qual - 33 < o.min_bq) continue;                                                                  // :316
      if (rpos < o.min_td - 1) continue;
      if (rpos + o.min_td > rd.l_qseq) continue;
      const int allele = (base == snps[(size_t)i].ref) ? 0 : ((base == snps[(size_t)i].alt) ? 1 : 2);     // :322
      const int bq = qual - 33 > o.cap_bq ? o.cap_bq : qual - 33;
      if (bq < 0 || bq > 127) fatal("dmx_store_add_read: base quality %d not in [0,127]", bq);
      st.hits.push_back({(int32_t)i, (uint8_t)allele, (uint8_t)bq});
    }
  };
  auto classify = [&](int nv_pass, int nv_red, int nv_valid) {                                             // :331-335
    if (nv_pass > 1) ++nReadsMultiSNPs;
    if (nv_pass > 0) ++nReadsPass; else if (nv_red > 0) ++nReadsRedundant; else if (nv_valid > 0) ++nReadsLQ; else ++nReadsN;
  };

  if (!windowed) {
    Read rd;
    Staged st;
    for (;;) {                                                                                             // :195
      sw.start();
      const bool more = sr.read(rd);
      sw.stop(0);
      if (!more) break;
      if (!stage(rd, st)) continue;
      sw.start();
      overlap(rd, st, snps.data());
      int nv_pass = 0, nv_red = 0;
      for (const Hit& h : st.hits) {
        const int ret = dmx_store_add_read(scl, h.snp, st.ibcd, st.umi.c_str(), h.allele, h.bq);         // :325
        if (ret < 0) fatal("%s", dmx_last_error());
        if (ret) ++nv_pass; else ++nv_red;
      }
      sw.stop(2);
      classify(nv_pass, nv_red, st.nv_valid);
    }
  } else {
    const size_t W = getenv("DMX_SCAN_WINDOW") ? (size_t)std::max(1, atoi(getenv("DMX_SCAN_WINDOW"))) : ((size_t)1 << 16);
    auto parallel_for = [&](size_t n, const std::function<void(size_t, size_t, int)>& fn) {               // fn(first, last, worker slot) on chunks of the range
      const size_t T = std::min<size_t>((size_t)n_threads, std::max<size_t>(1, n / 256));
      if (T <= 1) { fn(0, n, 0); return; }
      std::atomic<size_t> next{0};
      const size_t step = std::max<size_t>(256, n / (T * 8));
      WorkerPool::get().run((int)T - 1, [&](int slot) { for (size_t a; (a = next.fetch_add(step)) < n;) fn(a, std::min(n, a + step), slot); });
    };
    // barcode dictionaries of the parsing workers (slot -> barcode -> id in the slot) and, owned by the in-order stage, what each of
    // those ids is in the store: -1 not seen yet, -2 not in --group-list.  dmx_store_add_cell is then called once per (slot, barcode),
    // at the barcode's first read in BAM order as before.
    // (open addressing over the barcodes' bytes: one probe and one memcmp per read, no node to chase)
    struct BcDict {
      std::vector<uint32_t> slot;                // 0 = empty, else id + 1
      std::vector<uint64_t> key_off;             // id -> offset of its bytes in pool
      std::vector<uint32_t> key_len;
      std::vector<uint64_t> key_hash;
      std::string pool;
      size_t mask = 0;
      static uint64_t hash(const char* p, size_t n) {
        uint64_t h = 0x9E3779B97F4A7C15ull ^ (n * 0xFF51AFD7ED558CCDull);
        while (n >= 8) { uint64_t w; memcpy(&w, p, 8); h = (h ^ w) * 0xC4CEB9FE1A85EC53ull; h ^= h >> 29; p += 8; n -= 8; }
        uint64_t w = 0;
        memcpy(&w, p, n);
        h = (h ^ w) * 0xFF51AFD7ED558CCDull;
        return h ^ (h >> 32);
      }
      void grow() {
        const size_t cap = slot.empty() ? 1024 : slot.size() * 2;
        slot.assign(cap, 0); mask = cap - 1;
        for (size_t id = 0; id < key_off.size(); ++id) { size_t i = key_hash[id] & mask; while (slot[i]) i = (i + 1) & mask; slot[i] = (uint32_t)id + 1; }
      }
      int32_t intern(const char* p, size_t n) {
        if (key_off.size() * 2 >= slot.size()) grow();
        const uint64_t h = hash(p, n);
        for (size_t i = h & mask;; i = (i + 1) & mask) {
          const uint32_t v = slot[i];
          if (!v) {
            slot[i] = (uint32_t)key_off.size() + 1;
            key_off.push_back(pool.size()); key_len.push_back((uint32_t)n); key_hash.push_back(h);
            pool.append(p, n);
            return (int32_t)key_off.size() - 1;
          }
          const size_t id = v - 1;
          if (key_hash[id] == h && key_len[id] == n && memcmp(pool.data() + key_off[id], p, n) == 0) return (int32_t)id;
        }
      }
    };
    std::vector<BcDict> slot_dict((size_t)n_threads);
    std::vector<std::vector<int32_t>> slot_cell((size_t)n_threads);
    slot_cell_p = &slot_cell;
    // Three threads work on consecutive windows at the same time:
    //   reader   takes a window's records off the (already parallel) BGZF inflater and parses them on the worker threads;
    //   this one runs (S) over the parsed window in BAM order and snapshots the SNPs its reads' buffers span;
    //   sink     runs (O) on the worker threads and hands the window's observations to the store, in BAM order, then classifies the reads.
    // The reader touches nothing but the SamReader and its window; the sink reads the window, its SNP snapshot and the store's
    // observation shards and RD.PASS / RD.UNIQ counters of cells that existed when the window was staged (this thread only adds
    // cells and moves RD.TOTL meanwhile: dmx_store_add_batch is made for that).  A window's store batch is complete before the next
    // window's begins: "first UMI wins" sees the BAM order.
    constexpr int NW = 4;
    struct Window { std::vector<Read> rd; std::vector<Staged> st; std::vector<Snp> snps; std::vector<ChunkP> keep; int64_t snp_lo = 0; size_t n = 0; bool last = false; int state = 0; };
    // state: 0 free, 1 parsed, 2 staged.  (On the heap and torn down by a thread of its own after the scan: destroying 4 x W reads
    // with their strings and vectors is 50-70 ms that nothing has to wait for.)
    std::unique_ptr<Window[]> wins_owner(new Window[NW]);
    Window* const wins = wins_owner.get();
    // (a window's W records are constructed by the thread that first fills them: 4 x W x ~350 bytes is 50 ms of page faults in one go)
    std::mutex w_mu; std::condition_variable w_cv;
    bool w_stop = false;
    double px[8] = {0, 0, 0, 0, 0, 0, 0, 0};                      // DMX_CLI_TIMING: reader wait / slice / parse, sink wait / assemble / store / classify (one writer each)
    auto tnow = [] { return std::chrono::steady_clock::now(); };
    auto tsec = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
    std::thread reader([&] {
      for (int i = 0;; i = (i + 1) % NW) {
        const auto r0 = tnow();
        { std::unique_lock<std::mutex> lk(w_mu); w_cv.wait(lk, [&] { return wins[i].state == 0 || w_stop; }); if (w_stop) return; }
        Window& w = wins[i];
        w.n = 0; w.last = false;
        if (w.rd.size() < W) w.rd.resize(W);
        w.keep.clear();                                          // (the inflated chunks the window's previous records lay in)
        const auto r1 = tnow();
        while (w.n < W) { if (!sr.next_raw(w.rd[w.n], &w.keep)) { w.last = true; break; } ++w.n; }
        const auto r2 = tnow();
        parallel_for(w.n, [&](size_t a, size_t b, int slot) {
          CpuScope cs(1);
          BcDict& dict = slot_dict[(size_t)slot];
          for (size_t k = a; k < b; ++k) {
            Read& r = w.rd[k];
            sr.parse_raw(r);
            r.endpos_c = SamReader::endpos(r);
            r.cb_slot = -1;
            if (r.has_cb && !o.tag_group.empty()) {
              r.cb_slot = slot; r.cb_lid = dict.intern(r.cb_p, r.cb_n);
            }
          }
        });
        if (sw.on) { px[0] += tsec(r0, r1); px[1] += tsec(r1, r2); px[2] += tsec(r2, tnow()); }
        { std::lock_guard<std::mutex> lk(w_mu); w.state = 1; }
        w_cv.notify_all();
        if (w.last) return;
      }
    });
    std::thread sink([&] {
      std::vector<int32_t> b_snp, b_cell; std::vector<uint64_t> b_off; std::vector<uint32_t> b_len; std::vector<uint8_t> b_al, b_bq, b_new;
      std::string b_pool;
      for (int i = 0;; i = (i + 1) % NW) {
        const auto q0 = tnow();
        { std::unique_lock<std::mutex> lk(w_mu); w_cv.wait(lk, [&] { return wins[i].state == 2 || w_stop; }); if (w_stop) return; }
        Window& w = wins[i];
        const size_t n = w.n;
        const std::chrono::steady_clock::time_point k0 = std::chrono::steady_clock::now();
        if (sw.on) px[3] += tsec(q0, k0);
        parallel_for(n, [&](size_t a, size_t b, int) {
          CpuScope cs(2);
          for (size_t k = a; k < b; ++k) if (w.st[k].used) overlap(w.rd[k], w.st[k], w.snps.data() - w.snp_lo);
        });
        const std::chrono::steady_clock::time_point k1 = std::chrono::steady_clock::now();
        // the window's observations, in BAM order, into the store
        b_snp.clear(); b_cell.clear(); b_off.clear(); b_len.clear(); b_al.clear(); b_bq.clear(); b_pool.clear();
        for (size_t k = 0; k < n; ++k) {
          const Staged& st = w.st[k];
          if (!st.used || st.hits.empty()) continue;
          const uint64_t off = b_pool.size();
          const char* umi_p = st.umi_in_read ? w.rd[k].ub_p : st.umi.data();
          const size_t umi_n = st.umi_in_read ? w.rd[k].ub_n : st.umi.size();
          b_pool.append(umi_p, umi_n);
          for (const Hit& h : st.hits) { b_snp.push_back(h.snp); b_cell.push_back(st.ibcd); b_off.push_back(off); b_len.push_back((uint32_t)umi_n); b_al.push_back(h.allele); b_bq.push_back(h.bq); }
        }
        b_new.assign(b_snp.size(), 0);
        b_pool.push_back('\0');
        const auto k2 = tnow();
        if (dmx_store_add_batch(scl, (int64_t)b_snp.size(), b_snp.data(), b_cell.data(), b_pool.data(), b_off.data(), b_len.data(), b_al.data(), b_bq.data(),
                                b_new.data(), n_threads) != DMX_OK) fatal("%s", dmx_last_error());
        const auto k3 = tnow();
        size_t q = 0;
        for (size_t k = 0; k < n; ++k) {
          const Staged& st = w.st[k];
          if (!st.used) continue;
          int nv_pass = 0, nv_red = 0;
          for (size_t h = 0; h < st.hits.size(); ++h, ++q) { if (b_new[q]) ++nv_pass; else ++nv_red; }
          classify(nv_pass, nv_red, st.nv_valid);
        }
        if (sw.on) { px[4] += tsec(k1, k2); px[5] += tsec(k2, k3); px[6] += tsec(k3, tnow()); }
        if (sw.on) { sw.acc[2] += std::chrono::duration<double>(k1 - k0).count(); sw.acc[5] += std::chrono::duration<double>(std::chrono::steady_clock::now() - k1).count(); }
        const bool last = w.last;
        { std::lock_guard<std::mutex> lk(w_mu); w.state = 0; }
        w_cv.notify_all();
        if (last) return;
      }
    });
    struct PipeGuard { std::mutex& mu; std::condition_variable& cv; bool& stop; std::thread &a, &b; ~PipeGuard() { { std::lock_guard<std::mutex> lk(mu); stop = true; } cv.notify_all(); if (a.joinable()) a.join(); if (b.joinable()) b.join(); } } pipe_guard{w_mu, w_cv, w_stop, reader, sink};
    const auto loop_t0 = tnow();
    for (int wi = 0;; wi = (wi + 1) % NW) {
      sw.start();
      { std::unique_lock<std::mutex> lk(w_mu); w_cv.wait(lk, [&] { return wins[wi].state == 1; }); }
      sw.stop(0);                                                // (time this thread WAITED for parsed records)
      Window& w = wins[wi];
      const size_t n = w.n;
      const bool last_window = w.last;
      {
        const double vcf_before = sw.acc[1];
        const std::chrono::steady_clock::time_point s0 = std::chrono::steady_clock::now();
        int64_t lo = INT64_MAX, hi = 0;
        if (w.st.size() < W) w.st.resize(W);
        for (size_t k = 0; k < n; ++k) {
          w.st[k].used = false;
          if (sr.count_and_filter(w.rd[k]) && stage(w.rd[k], w.st[k])) { lo = std::min(lo, w.st[k].ibeg); hi = std::max(hi, w.st[k].ibeg + w.st[k].nbuf); }
        }
        if (lo > hi) { lo = 0; hi = 0; }
        w.snp_lo = lo;
        w.snps.assign(snps.begin() + lo, snps.begin() + hi);       // the sink reads this copy: `snps` keeps growing under this thread
        if (sw.on) sw.acc[4] += std::chrono::duration<double>(std::chrono::steady_clock::now() - s0).count() - (sw.acc[1] - vcf_before);
      }
      { std::lock_guard<std::mutex> lk(w_mu); w.state = 2; }
      w_cv.notify_all();
      if (last_window) break;
    }
    const auto loop_t1 = tnow();
    sink.join();                                                  // (the guard then finds both threads finished)
    reader.join();
    std::thread([](Window* w) { delete[] w; }, wins_owner.release()).detach();
    if (sw.on) notice("scan threads: reader waited %.3f s, sliced records %.3f s, parsed %.3f s; sink waited %.3f s, assembled %.3f s, stored %.3f s, classified %.3f s; set-up %.3f s, in-order loop %.3f s, drain %.3f s",
                      px[0], px[1], px[2], px[3], px[4], px[5], px[6], tsec(scan_t0, loop_t0), tsec(loop_t0, loop_t1), tsec(loop_t1, tnow()));
  }
  if (n_warn_u > 10) notice("WARNING: Suppressed a total of %d UMI warnings...", n_warn_u);
  if (n_warn_g > 10) notice("WARNING: Suppressed a total of %d droplet/cell barcode warnings...", n_warn_g);
  const double scan_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - scan_t0).count();
  if (sw.on) {
    struct rusage ru;
    getrusage(RUSAGE_SELF, &ru);
    notice("process CPU so far: %.3f s user + %.3f s system; inside the parallel sections: inflate %.3f s, record parsing %.3f s, overlap %.3f s", ru.ru_utime.tv_sec + 1e-6 * ru.ru_utime.tv_usec, ru.ru_stime.tv_sec + 1e-6 * ru.ru_stime.tv_usec,
           1e-9 * g_cpu_ns[0].load(), 1e-9 * g_cpu_ns[1].load(), 1e-9 * g_cpu_ns[2].load());
  }
  if (sw.on) notice("scan timing (%d threads, %s): total %.3f s = %.3g reads/s; alignment reader %.3f s, record parsing %.3f s, lock-step bookkeeping %.3f s, VCF reader (wait) %.3f s, overlap%s %.3f s, store batches %.3f s",
                    n_threads, windowed ? "windowed" : "read by read", scan_s, (double)sr.n_read / scan_s, sw.acc[0], sw.acc[3], sw.acc[4], sw.acc[1], windowed ? "" : " + store", sw.acc[2], sw.acc[5]);
  notice("Finished reading %d markers from the VCF file", (int)snps.size());
  notice("Total number input reads : %lld", (long long)sr.n_read);
  notice("Total number valid droplets observed : %d", dmx_store_n_cells(scl));
  notice("Total number valid SNPs observed     : %d", dmx_store_n_snps(scl));
  notice("Total number of read-QC-passed reads : %lld ", (long long)(sr.n_read - sr.n_skip));
  notice("Total number of skipped reads with ignored barcodes : %ld", nReadsSkipBCD);
  notice("Total number of non-skipped reads with considered barcodes : %ld", nReadsTMP);
  notice("Total number of gapped/noninformative reads : %ld", nReadsN);
  notice("Total number of base-QC-failed reads : %ld", nReadsLQ);
  notice("Total number of redundant reads : %ld", nReadsRedundant);
  notice("Total number of pass-filtered reads : %ld", nReadsPass);
  notice("Total number of pass-filtered reads overlapping with multiple SNPs : %ld", nReadsMultiSNPs);

  if (o.pileup_only) {          // scan result for inspection / CPU tests: store in CSR order + genotype matrix (hex floats)
    dmx_pileup pl;
    if (dmx_store_freeze(scl, &pl) != DMX_OK) fatal("%s", dmx_last_error());
    FILE* f = fopen((o.out + ".pileup.txt").c_str(), "w");
    if (!f) fatal("Cannot create %s.pileup.txt", o.out.c_str());
    fprintf(f, "NV\t%d\nNSNP\t%d\nNCELL\t%d\n", nv, pl.n_snps, pl.n_cells);
    for (int j = 0; j < nv; ++j) fprintf(f, "SM\t%s\n", vr.sample_id(j));
    for (int32_t s = 0; s < pl.n_snps; ++s) {
      fprintf(f, "SNP\t%d\t%d\t%lld\t%c\t%c", s, snps[(size_t)s].rid, (long long)snps[(size_t)s].pos, snps[(size_t)s].ref, snps[(size_t)s].alt);
      for (int q = 0; q < nv * 3; ++q) fprintf(f, "\t%a", (double)G[(size_t)s * nv * 3 + q]);
      fprintf(f, "\n");
    }
    for (int32_t c = 0; c < pl.n_cells; ++c) {
      fprintf(f, "CELL\t%d\t%s\t%d\t%d\t%d\n", c, dmx_store_barcode(scl, c), pl.rd_totl[c], pl.rd_pass[c], pl.rd_uniq[c]);
      int64_t r = pl.cell_read_off[c];
      for (int64_t p = pl.cell_pair_off[c]; p < pl.cell_pair_off[c + 1]; ++p) {
        uint32_t n = 0;
        memcpy(&n, (const uint8_t*)pl.pair_nrd + (size_t)p * (size_t)pl.nrd_width, (size_t)pl.nrd_width);
        fprintf(f, "PAIR\t%d\t%u", pl.pair_snp[p], n);
        for (uint32_t q = 0; q < n; ++q) fprintf(f, "\t%d:%d", pl.reads[r + q] >> 7, pl.reads[r + q] & 127);
        fprintf(f, "\n");
        r += n;
      }
    }
    fclose(f);
    dmx_store_free(scl);
    return 0;
  }

  notice("Starting to prune out cells with too few reads...");                                             // :367-383 (dead in the reference for options >= 0)
  notice("Finishing pruning out %d cells with too few reads...", 0);
  notice("Starting to identify best matching individual IDs");
  std::vector<const char*> sm((size_t)nv);
  for (int j = 0; j < nv; ++j) sm[(size_t)j] = vr.sample_id(j);
  dmx_job job;
  memset(&job, 0, sizeof job);
  job.store = scl; job.g = G.data(); job.n_samples = nv; job.sample_ids = sm.data();
  job.n_alpha = (int32_t)o.alpha.size(); job.alpha = o.alpha.data(); job.doublet_prior = o.doublet_prior;
  job.min_total = o.min_total; job.min_uniq = o.min_uniq; job.min_snp = o.min_snp; job.write_pair = o.write_pair;
  job.out_prefix = o.out.c_str(); job.device = o.gpu; job.arbiter = o.no_arbiter ? 0 : 1; job.n_gpus = o.gpus; job.mode = (o.fast && !o.strict) ? DMX_MODE_FAST : DMX_MODE_STRICT;
  dmx_job_timing jt;
  memset(&jt, 0, sizeof jt);
  job.timing = &jt;
  if (dmx_demuxlet_run(&job) != DMX_OK) fatal("[E:%s] %s", __func__, dmx_last_error());
  // the hot path's own notices (:468, :524; the per-1000-droplets / per-100-cells progress lines in between belong to loops that do not exist here)
  notice("Identifying best-matching individual..");
  notice("Finished processing %d droplets total", jt.n_cells_single);
  notice("Finished writing output files");                                                                 // :876
  dmx_store_free(scl);
  return 0;
}
