// Test helper (CPU): the arithmetic of dmx_log() (demuxlet_amd/csrc/dmx_log.hpp) executed on the host, bit-for-bit the
// operations the device performs (fma via libm's correctly-rounded fma()).
#include "dmx_log.hpp"
extern "C" void dmx_log_emul_n(const double* x, double* y, long n) { for (long i = 0; i < n; ++i) y[i] = dmx_log_host_emul(x[i]); }
extern "C" void dmx_log_lite_emul_n(const double* x, double* y, long n) { for (long i = 0; i < n; ++i) y[i] = dmx_log_comp_host_emul(x[i]); }   // the compensated form kept for the record
extern "C" void dmx_log2_emul_n(const double* x, double* y, long n) { for (long i = 0; i < n; ++i) y[i] = dmx_log2_host_emul(x[i]); }   // the K2 kernels' log (256 bins)
extern "C" void dmx_log2_lite32_emul_n(const double* x, double* y, long n) { for (long i = 0; i < n; ++i) y[i] = dmx_log2_lite32_host_emul(x[i]); }   // FAST k_doublet_sym's second log (32 bins, round 6)
