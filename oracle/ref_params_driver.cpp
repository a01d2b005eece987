// TEST INFRASTRUCTURE (row f4 pin).  The reference's OWN command-line machinery — params.cpp + Error.cpp compiled from where they
// lie in /root/reference, nothing replaced — driven by the reference's OWN option table: lines 9-76 of cmd_cram_demuxlet.cpp (the
// option variables, BEGIN_LONG_PARAMS ... END_LONG_PARAMS, pl.Read, pl.Status) are extracted at build time into a temp file
// (oracle/Makefile, never stored in the repo) and #included below.  The only things supplied here are the two objects those lines
// write option values into: `sr` and `vr` are htslib-dependent reader classes in the reference (sam_filtered_reader.h,
// bcf_filtered_reader.h); the option table touches nothing of them but the plain data members declared here.
// Prints what the reference prints for the given arguments: the --help text (exit status 1) or the parameter status echo.
#include <cstdint>
#include <set>
#include <string>
#include <vector>
#include "params.h"
#include "Error.h"

struct SamSide { std::string sam_file_name; int32_t verbose; struct { int32_t exclude_flag, minMQ; } filt; };
struct VcfSide { std::string bcf_file_name, sample_id_list; int32_t verbose; struct { int32_t minMAC, maxAlleles; double minCallRate; } vfilt; };

int32_t main(int32_t argc, char** argv) {
  SamSide sr;
  VcfSide vr;
#include DMX_REF_PARAMS
  return 0;
}
