// How fast can ONE output file take text on this box?  (The `.pair` file of cfg4's shard is 1.1 GB.)  Times, for a buffer of <MB> megabytes:
// fwrite from pageable memory in 1 MiB pieces, write(2) in 8 MiB pieces, pwrite from T threads at disjoint offsets of one file, the same into T files,
// and hipMemcpy D2H of the same bytes into pageable and into pinned host memory.   usage: write_rate <dir> [MB] [threads]
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
  const std::string dir = argc > 1 ? argv[1] : "/tmp";
  const size_t n = (size_t)(argc > 2 ? atoi(argv[2]) : 1100) << 20;
  const int T = argc > 3 ? atoi(argv[3]) : 8;
  std::vector<char> buf(n);
  for (size_t i = 0; i < n; ++i) buf[i] = (char)('0' + i % 10);
  auto report = [&](const char* what, double t) { printf("%-46s %.3f s = %.2f GB/s\n", what, t, n / t / 1e9); fflush(stdout); };
  for (int rep = 0; rep < 2; ++rep) {
    { const std::string p = dir + "/wr_fwrite"; FILE* f = fopen(p.c_str(), "w"); double t0 = now();
      for (size_t o = 0; o < n; o += 1 << 20) fwrite(buf.data() + o, 1, std::min<size_t>(1 << 20, n - o), f);
      fclose(f); report("fwrite 1 MiB pieces (pageable source)", now() - t0); unlink(p.c_str()); }
    { const std::string p = dir + "/wr_write"; int fd = open(p.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644); double t0 = now();
      for (size_t o = 0; o < n; o += 8 << 20) if (write(fd, buf.data() + o, std::min<size_t>(8 << 20, n - o)) < 0) return 1;
      close(fd); report("write(2) 8 MiB pieces", now() - t0); unlink(p.c_str()); }
    { const std::string p = dir + "/wr_pwrite"; int fd = open(p.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644); double t0 = now();
      if (ftruncate(fd, (off_t)n)) return 1;
      std::vector<std::thread> th; const size_t per = (n + T - 1) / T;
      for (int t = 0; t < T; ++t) th.emplace_back([&, t] { for (size_t o = t * per; o < std::min(n, (t + 1) * per); o += 8 << 20)
        if (pwrite(fd, buf.data() + o, std::min<size_t>(8 << 20, std::min(n, (t + 1) * per) - o), (off_t)o) < 0) abort(); });
      for (auto& x : th) x.join();
      close(fd); char w[96]; snprintf(w, sizeof w, "pwrite, %d threads, ONE file, disjoint ranges", T); report(w, now() - t0); unlink(p.c_str()); }
    { double t0 = now(); std::vector<std::thread> th; const size_t per = (n + T - 1) / T;
      for (int t = 0; t < T; ++t) th.emplace_back([&, t] { const std::string p = dir + "/wr_part" + std::to_string(t); int fd = open(p.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
        for (size_t o = t * per; o < std::min(n, (t + 1) * per); o += 8 << 20) if (write(fd, buf.data() + o, std::min<size_t>(8 << 20, std::min(n, (t + 1) * per) - o)) < 0) abort();
        close(fd); unlink(p.c_str()); });
      for (auto& x : th) x.join();
      char w[96]; snprintf(w, sizeof w, "write, %d threads, %d FILES", T, T); report(w, now() - t0); }
  }
  void* d = nullptr; if (hipMalloc(&d, n) != hipSuccess) { printf("no GPU\n"); return 0; }
  hipMemset(d, 49, n); hipDeviceSynchronize();
  for (int rep = 0; rep < 2; ++rep) {
    double t0 = now(); hipMemcpy(buf.data(), d, n, hipMemcpyDeviceToHost); report("hipMemcpy D2H -> pageable", now() - t0);
    void* h = nullptr; t0 = now(); hipHostMalloc(&h, n, hipHostMallocDefault); report("hipHostMalloc (pin)", now() - t0);
    t0 = now(); hipMemcpy(h, d, n, hipMemcpyDeviceToHost); report("hipMemcpy D2H -> pinned", now() - t0);
    { const std::string p = dir + "/wr_pinned"; int fd = open(p.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644); t0 = now();
      for (size_t o = 0; o < n; o += 8 << 20) if (write(fd, (char*)h + o, std::min<size_t>(8 << 20, n - o)) < 0) return 1;
      close(fd); report("write(2) 8 MiB pieces from the pinned buffer", now() - t0); unlink(p.c_str()); }
    hipHostFree(h);
  }
  return 0;
}
