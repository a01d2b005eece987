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
#include <zlib.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <map>
#include <set>
#include <string>
#include <vector>

#include "dmx.h"

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
[[noreturn]] void fatal(const char* fmt, ...) {
  fprintf(stderr, "\nFATAL ERROR - \n");
  va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap);
  fprintf(stderr, "\n\n");
  exit(EXIT_FAILURE);      // the reference throws an uncaught exception here (Error.cpp:39): abnormal termination either way
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

void print_help(const std::vector<OptDef>& defs) {
  fprintf(stderr, "\nDetailed instructions of parameters are available. Ones with \"[]\" are in effect:\n\nAvailable Options:\n");
  const char* grp = nullptr;
  for (const OptDef& d : defs) {
    if (!grp || strcmp(grp, d.group)) { fprintf(stderr, "\n== %s ==\n", d.group); grp = d.group; }
    fprintf(stderr, "   --%-14s %s\n", d.name, d.help);
  }
  fprintf(stderr, "\nNOTES:\nWhen --help was included in the argument. The program prints the help message but do not actually run\n");
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
      const bool multi = d->type == O_MULTI_DOUBLE || d->type == O_MULTI_STRING;
      if (!multi && touched.count(name)) fatal("Redundant use of option --%s is not allowed", name);               // params.cpp:124,143...
      touched.insert(name);
      switch (d->type) {
        case O_BOOL: *(bool*)d->ptr = true; break;
        case O_INT: if (!check_integer(extra)) fatal("Invalid argument --%s %s. Integer was expected", name, extra ? extra : "(null)"); *(int*)d->ptr = atoi(extra); ++i; break;
        case O_DOUBLE: if (!check_double(extra)) fatal("Invalid argument --%s %s. Double was expected", name, extra ? extra : "(null)"); *(double*)d->ptr = atof(extra); ++i; break;
        case O_STRING: if (!extra) fatal("Invalid argument --%s (null). String was expected", name); *(std::string*)d->ptr = extra; ++i; break;
        case O_MULTI_DOUBLE: if (!check_double(extra)) fatal("Invalid argument --%s %s. Double was expected", name, extra ? extra : "(null)"); ((std::vector<double>*)d->ptr)->push_back(atof(extra)); ++i; break;
        case O_MULTI_STRING: if (!extra) fatal("Invalid argument --%s (null). String was expected", name); ((std::vector<std::string>*)d->ptr)->push_back(extra); ++i; break;
      }
    } else {
      snprintf(ebuf, sizeof ebuf, "Cannot correspond command line parameter %s (#%d) to any of the options\n", a, i); errors += ebuf;
    }
  }
  if (!errors.empty()) fatal("Problems encountered parsing command line:\n\n%s", errors.c_str());                   // params.cpp:562-567
}

// ---- line / byte input over zlib (plain files pass through) ----------------------------------------------------------
struct GzIn {
  gzFile f = nullptr;
  std::string path;
  bool open(const std::string& p) { path = p; f = gzopen(p.c_str(), "rb"); if (f) gzbuffer(f, 1 << 20); return f != nullptr; }
  ~GzIn() { if (f) gzclose(f); }
  bool getline(std::string& line) {
    line.clear();
    char buf[65536];
    while (gzgets(f, buf, sizeof buf)) {
      line += buf;
      if (!line.empty() && line.back() == '\n') { line.pop_back(); if (!line.empty() && line.back() == '\r') line.pop_back(); return true; }
    }
    return !line.empty();
  }
  bool read(void* dst, size_t n) { return gzread(f, dst, (unsigned)n) == (int)n; }
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
    while (in.getline(line)) {
      if (line.rfind("##contig=<ID=", 0) == 0) {
        size_t b = 13, e = line.find_first_of(",>", b);
        const std::string id = line.substr(b, e - b);
        if (!contig_rid.count(id)) { int r = (int)contig_rid.size(); contig_rid[id] = r; }
      } else if (line.rfind("#CHROM", 0) == 0) {
        auto f = split(line, '\t');
        for (size_t i = 9; i < f.size(); ++i) samples.push_back(f[i]);
        have_header = true;
        break;
      } else if (line.rfind("##", 0) != 0) {
        fatal("[E:%s] %s does not look like a VCF file (BCF/CRAM need htslib, which this build does not use)", __func__, path.c_str());
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

  // next variant that passes the filter (bcf_filtered_reader.cpp:751-764 + passed_vfilter :498-574); false at EOF
  bool read(Variant& v) {
    std::string line;
    while (in.getline(line)) {
      if (line.empty() || line[0] == '#') continue;
      ++n_read;
      auto f = split(line, '\t');
      if (f.size() < 10) fatal("[E:%s] VCF record with %u columns at line starting %.40s", __func__, (unsigned)f.size(), line.c_str());
      if (verbose > 0 && n_read % verbose == 0) notice("Reading %lld variants at %s:%s, Skipping %lld, Missing 0.", (long long)n_read, f[0].c_str(), f[1].c_str(), (long long)n_skip);
      if (!contig_rid.count(f[0])) { int r = (int)contig_rid.size(); contig_rid[f[0]] = r; }     // headers without ##contig lines
      v.rid = contig_rid[f[0]];
      v.pos = atoll(f[1].c_str()) - 1;
      v.ref = f[3]; v.alt = f[4];
      v.rlen = (int)v.ref.size();
      const auto alts = split(f[4], ',');
      v.n_allele = (f[4] == "." ? 1 : 1 + (int)alts.size());
      v.ref0 = v.ref.empty() ? 'N' : v.ref[0];
      v.alt0 = (v.n_allele > 1 && !alts[0].empty()) ? alts[0][0] : '.';
      if (v.n_allele > max_alleles) { ++n_skip; continue; }                                    // :534
      // FORMAT keys
      const auto keys = split(f[8], ':');
      int i_gt = -1, i_fld = -1;
      for (size_t k = 0; k < keys.size(); ++k) { if (keys[k] == "GT") i_gt = (int)k; if (keys[k] == field) i_fld = (int)k; }
      if (i_gt < 0) fatal("[E:%s] Cannot find the field GT from the VCF file at position %s:%lld", __func__, f[0].c_str(), (long long)v.pos + 1);   // :548-549
      const int nv = nsamples();
      std::vector<int32_t> alleles((size_t)nv * 2, -1);
      std::vector<std::vector<std::string>> sf((size_t)nv);
      int an = 0; std::vector<int> acs((size_t)std::max(v.n_allele, 2), 0);
      for (int i = 0; i < nv; ++i) {
        sf[i] = split(f[9 + sm_icols[i]], ':');
        const std::string& gt = sf[i][i_gt < (int)sf[i].size() ? i_gt : 0];
        // diploid GT "a/b" or "a|b"; '.' = missing allele (bcf_gt_allele < 0); haploid "a" leaves the second allele missing
        size_t sep = gt.find_first_of("/|");
        const std::string a1 = gt.substr(0, sep), a2 = sep == std::string::npos ? "." : gt.substr(sep + 1);
        auto al = [](const std::string& s) { return (s.empty() || s == ".") ? -1 : atoi(s.c_str()); };
        alleles[2 * i] = al(a1); alleles[2 * i + 1] = al(a2);
        for (int h = 0; h < 2; ++h) { const int x = alleles[2 * i + h]; if (x >= 0 && x < (int)acs.size()) { ++an; ++acs[x]; } }   // :230-240
      }
      if (min_callrate > (double)an / (2.0 * (double)nv)) { ++n_skip; continue; }              // :554
      const int ac = an - acs[0];
      if ((ac < min_mac) || (an - ac < min_mac)) { ++n_skip; continue; }                       // :565
      // parse_posteriors (:360-454) through the library's a3 transforms
      v.gps.assign((size_t)nv * 3, 0.f);
      if (field == "GT") {
        if (dmx_geno_from_gt(alleles.data(), nv, gt_error, v.gps.data()) != DMX_OK) fatal("[E:%s] %s", __func__, dmx_last_error());
      } else {
        if (i_fld < 0) fatal("[E:%s] Cannot parse posterior probability at %s:%lld", __func__, f[0].c_str(), (long long)v.pos + 1);   // :154, :212
        if (field == "PL") {
          std::vector<int32_t> pl((size_t)nv * 3, INT32_MIN);
          for (int i = 0; i < nv; ++i) {
            if (i_fld >= (int)sf[i].size()) continue;
            const auto p = split(sf[i][i_fld], ',');
            for (int g = 0; g < 3 && g < (int)p.size(); ++g) pl[(size_t)i * 3 + g] = (p[g] == "." ? INT32_MIN : atoi(p[g].c_str()));
          }
          if (dmx_geno_from_pl(pl.data(), nv, v.gps.data()) != DMX_OK) fatal("[E:%s] %s", __func__, dmx_last_error());
        } else {
          std::vector<float> gp((size_t)nv * 3, NAN);
          for (int i = 0; i < nv; ++i) {
            if (i_fld >= (int)sf[i].size()) continue;
            const auto p = split(sf[i][i_fld], ',');
            for (int g = 0; g < 3 && g < (int)p.size(); ++g) gp[(size_t)i * 3 + g] = (p[g] == "." ? NAN : (float)atof(p[g].c_str()));
          }
          if (dmx_geno_from_gp(gp.data(), nv, gt_error, v.gps.data()) != DMX_OK) fatal("[E:%s] %s", __func__, dmx_last_error());
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
  std::string qname, seq, qual, cb, ub;
  bool has_cb = false, has_ub = false;
  int flag = 0, tid = -1, mapq = 0;
  int64_t pos = 0;              // 0-based
  std::vector<std::pair<char, uint32_t>> cigar;
  int l_qseq = 0;
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
    const int got = gzread(in.f, magic, 4);
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
      gzrewind(in.f);
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

  bool parse_sam_line(const std::string& line, Read& r) {
    auto f = split(line, '\t');
    if (f.size() < 11) fatal("[E:%s] SAM record with %u fields", __func__, (unsigned)f.size());
    r.qname = f[0]; r.flag = atoi(f[1].c_str());
    auto it = target_id.find(f[2]);
    r.tid = (f[2] == "*" || it == target_id.end()) ? -1 : it->second;
    r.pos = atoll(f[3].c_str()) - 1; r.mapq = atoi(f[4].c_str());
    r.cigar.clear();
    if (f[5] != "*") {
      uint32_t n = 0;
      for (char ch : f[5]) { if (ch >= '0' && ch <= '9') n = n * 10 + (uint32_t)(ch - '0'); else { r.cigar.emplace_back(ch, n); n = 0; } }
    }
    if (f[9] == "*") { r.seq.clear(); } else { r.seq = f[9]; }
    r.l_qseq = (int)r.seq.size();
    for (char& ch : r.seq) {                    // htslib stores 4-bit codes and prints "=ACMGRSVTWYHKDBN"
      ch = (char)toupper((unsigned char)ch);
      if (!strchr("=ACMGRSVTWYHKDBN", ch)) ch = 'N';
    }
    if (f[10] == "*") r.qual.assign((size_t)r.l_qseq, (char)(0xff + 33)); else r.qual = f[10];
    r.has_cb = r.has_ub = false;
    for (size_t i = 11; i < f.size(); ++i) {
      const std::string& t = f[i];
      if (t.size() >= 5 && t[2] == ':' && t[4] == ':') {
        if (gtag[0] && t[0] == gtag[0] && t[1] == gtag[1] && t[3] == 'Z') { r.cb = t.substr(5); r.has_cb = true; }
        if (utag[0] && t[0] == utag[0] && t[1] == utag[1] && t[3] == 'Z') { r.ub = t.substr(5); r.has_ub = true; }
      }
    }
    return true;
  }

  bool parse_bam_record(Read& r) {
    int32_t block = 0;
    if (!in.read(&block, 4)) return false;
    std::vector<uint8_t> b((size_t)block);
    if (!in.read(b.data(), (size_t)block)) fatal("[E:%s] truncated BAM record", __func__);
    auto i32 = [&](size_t o) { int32_t v; memcpy(&v, &b[o], 4); return v; };
    auto u16 = [&](size_t o) { uint16_t v; memcpy(&v, &b[o], 2); return v; };
    r.tid = i32(0); r.pos = i32(4);
    const int l_read_name = b[8]; r.mapq = b[9];
    const int n_cigar = u16(12); r.flag = u16(14);
    r.l_qseq = i32(16);
    size_t o = 32;
    r.qname.assign((const char*)&b[o], (size_t)std::max(0, l_read_name - 1)); o += (size_t)l_read_name;
    r.cigar.clear();
    for (int i = 0; i < n_cigar; ++i) { uint32_t c; memcpy(&c, &b[o], 4); o += 4; r.cigar.emplace_back("MIDNSHP=XB"[std::min<uint32_t>(c & 0xf, 9)], c >> 4); }
    r.seq.resize((size_t)r.l_qseq);
    for (int i = 0; i < r.l_qseq; ++i) r.seq[i] = "=ACMGRSVTWYHKDBN"[(b[o + (size_t)i / 2] >> ((i & 1) ? 0 : 4)) & 0xf];
    o += (size_t)(r.l_qseq + 1) / 2;
    r.qual.resize((size_t)r.l_qseq);
    for (int i = 0; i < r.l_qseq; ++i) r.qual[i] = (char)(b[o + (size_t)i] + 33);
    o += (size_t)r.l_qseq;
    r.has_cb = r.has_ub = false;
    while (o + 3 <= b.size()) {                 // aux fields
      const char t0 = (char)b[o], t1 = (char)b[o + 1], ty = (char)b[o + 2];
      o += 3;
      size_t len = 0;
      if (ty == 'Z' || ty == 'H') {
        const char* s = (const char*)&b[o]; len = strlen(s) + 1;
        if (ty == 'Z') {
          if (gtag[0] && t0 == gtag[0] && t1 == gtag[1]) { r.cb = s; r.has_cb = true; }
          if (utag[0] && t0 == utag[0] && t1 == utag[1]) { r.ub = s; r.has_ub = true; }
        }
      } else if (ty == 'A' || ty == 'c' || ty == 'C') len = 1;
      else if (ty == 's' || ty == 'S') len = 2;
      else if (ty == 'i' || ty == 'I' || ty == 'f') len = 4;
      else if (ty == 'B') {
        const char sub = (char)b[o]; int32_t cnt; memcpy(&cnt, &b[o + 1], 4);
        const size_t es = (sub == 'c' || sub == 'C') ? 1 : ((sub == 's' || sub == 'S') ? 2 : 4);
        len = 5 + es * (size_t)cnt;
      } else fatal("[E:%s] unknown BAM aux type %c", __func__, ty);
      o += len;
    }
    return true;
  }

  // next read passing the filter (sam_filtered_reader.cpp:233-258, passed_filter :284-296); false at EOF
  bool read(Read& r) {
    for (;;) {
      if (is_bam) { if (!parse_bam_record(r)) return false; }
      else {
        std::string line;
        if (have_pending) { line.swap(pending); have_pending = false; }
        else if (!in.getline(line)) return false;
        if (line.empty()) continue;
        parse_sam_line(line, r);
      }
      ++n_read;
      if (verbose > 0 && n_read % verbose == 0) notice("Reading %lld reads at %s:%lld and skipping %lld", (long long)n_read, r.tid >= 0 ? targets[r.tid].c_str() : "*", (long long)r.pos + 1, (long long)n_skip);
      if (r.mapq < min_mq || (excl_flag & r.flag)) { ++n_skip; continue; }
      return true;
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
      if (rp < rlen) { base = r.seq[(size_t)rp]; qual = r.qual[(size_t)rp]; } else { base = 0; qual = 0; }
    } else {
      rp = kNA;
    }
  }
  if (rp >= rlen) { rp = kNA; base = '.'; }
  rpos = (int)rp;
}

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

  Read rd;
  while (sr.read(rd)) {                                                                                    // :195
    const int64_t endpos = SamReader::endpos(rd);
    const int tid2rid = rd.tid >= 0 ? vr.name2id(sr.targets[(size_t)rd.tid]) : -1;
    if (tid2rid < 0) continue;                                                                             // :198-200
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
      if (vr.read(v)) {
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
      if (rd.has_cb) sbcd = rd.cb.c_str();
      else {
        if (n_warn_g < 10) notice("WARNING: Cannot find Droplet/Cell tag %s from %lld-th read %s at %s:%lld-%lld. Treating all of them as a single group", o.tag_group.c_str(), (long long)sr.n_read, rd.qname.c_str(), sr.targets[(size_t)rd.tid].c_str(), (long long)rd.pos, (long long)endpos);
        else if (n_warn_g == 10) notice("WARNING: Suppressing 10+ missing Droplet/Cell tag warnings...");
        ++n_warn_g;
      }
      if (bcd_set.empty() || bcd_set.count(sbcd)) {
        ibcd = dmx_store_add_cell(scl, sbcd);
        const int32_t nb = dmx_store_n_cells(scl);
        if (ibcd + 1 == nb && nb % 1000 == 0) notice("Observed %d droplets with unique cell barcode", nb);
      } else { ++nReadsSkipBCD; continue; }
    }
    ++nReadsTMP;
    // UMI (:272-293)
    std::string sumi(".");
    if (o.tag_umi.empty()) { char b[32]; snprintf(b, sizeof b, "%x", rand()); sumi += b; }
    else if (rd.has_ub) sumi = rd.ub;
    else {
      if (n_warn_u < 10) notice("WARNING: Cannot find UMI tag %s from %lld-th read %s at %s:%lld-%lld. Treating all of them as a single UMI", o.tag_umi.c_str(), (long long)sr.n_read, rd.qname.c_str(), sr.targets[(size_t)rd.tid].c_str(), (long long)rd.pos, (long long)endpos);
      else if (n_warn_u == 10) notice("WARNING: Suppressing 10+ UMI warnings...");
      ++n_warn_u;
    }
    dmx_store_count_read(scl, ibcd);                                                                       // :295
    int nv_pass = 0, nv_red = 0, nv_valid = 0;
    for (int64_t i = ibeg; i < ibeg + nbuf; ++i) {                                                         // :306
      char base, qual; int rpos;
      base_at(rd, snps[(size_t)i].pos, base, qual, rpos);
      if (rpos == kNA) continue;
      if (base == 'N') continue;
      ++nv_valid;
      if (qual - 33 < o.min_bq) continue;                                                                  // :316
      if (rpos < o.min_td - 1) continue;
      if (rpos + o.min_td > rd.l_qseq) continue;
      const int allele = (base == snps[(size_t)i].ref) ? 0 : ((base == snps[(size_t)i].alt) ? 1 : 2);     // :322
      const int bq = qual - 33 > o.cap_bq ? o.cap_bq : qual - 33;
      const int ret = dmx_store_add_read(scl, (int32_t)i, ibcd, sumi.c_str(), allele, bq);                // :325
      if (ret < 0) fatal("%s", dmx_last_error());
      if (ret) ++nv_pass; else ++nv_red;
    }
    if (nv_pass > 1) ++nReadsMultiSNPs;
    if (nv_pass > 0) ++nReadsPass; else if (nv_red > 0) ++nReadsRedundant; else if (nv_valid > 0) ++nReadsLQ; else ++nReadsN;
  }
  if (n_warn_u > 10) notice("WARNING: Suppressed a total of %d UMI warnings...", n_warn_u);
  if (n_warn_g > 10) notice("WARNING: Suppressed a total of %d droplet/cell barcode warnings...", n_warn_g);
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
  job.out_prefix = o.out.c_str(); job.device = o.gpu; job.arbiter = o.no_arbiter ? 0 : 1; job.n_gpus = o.gpus;
  if (dmx_demuxlet_run(&job) != DMX_OK) fatal("[E:%s] %s", __func__, dmx_last_error());
  notice("Finished writing output files");                                                                 // :876
  dmx_store_free(scl);
  return 0;
}
