// Test driver for demuxlet_amd/csrc/dmx_inflate.hpp (built by tests/test_inflate_cpu.py with -fsanitize=address,undefined):
// dmxz::inflate_raw against zlib's inflate on streams zlib's deflate made from several kinds of data at every level / strategy,
// on damaged streams (must never touch memory outside its buffers, and must never claim success with other bytes than zlib's),
// and dmxz::crc32_of against zlib's crc32.  Prints one line of counts; exit code 0 = all good.
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>
#include "dmx_inflate.hpp"

static uint32_t rng_state = 2463534242u;
static uint32_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 17; rng_state ^= rng_state << 5; return rng_state; }

static std::vector<uint8_t> make_data(int kind, size_t n) {
  std::vector<uint8_t> d(n);
  switch (kind) {
    case 0: for (auto& b : d) b = (uint8_t)rnd(); break;                                   // incompressible
    case 1: for (auto& b : d) b = "ACGT"[rnd() & 3]; break;                                // four symbols
    case 2: for (size_t i = 0; i < n; ++i) d[i] = (uint8_t)(i < 7 ? rnd() : d[i - 1 - rnd() % 7]); break;   // distances 1..7
    case 3: for (auto& b : d) b = 0; break;                                                // one long run (distance 1, length 258)
    case 4: {                                                                              // BAM-like: short repeats + noisy bytes
      size_t i = 0;
      while (i < n) {
        if (i > 300 && (rnd() & 3)) { const size_t len = 3 + rnd() % 12, dist = 1 + rnd() % 300; for (size_t k = 0; k < len && i < n; ++k, ++i) d[i] = d[i - dist]; }
        else d[i++] = (uint8_t)(33 + rnd() % 42);
      }
      break;
    }
    default: for (size_t i = 0; i < n; ++i) d[i] = (uint8_t)((i * 2654435761u) >> 13);     // structured, long matches far back
  }
  return d;
}

static std::vector<uint8_t> deflate_raw(const std::vector<uint8_t>& d, int level, int strategy) {
  z_stream zs;
  memset(&zs, 0, sizeof zs);
  deflateInit2(&zs, level, Z_DEFLATED, -15, 8, strategy);
  std::vector<uint8_t> out(deflateBound(&zs, (uLong)d.size()) + 64);
  zs.next_in = (Bytef*)d.data(); zs.avail_in = (uInt)d.size();
  zs.next_out = out.data(); zs.avail_out = (uInt)out.size();
  deflate(&zs, Z_FINISH);
  out.resize(zs.total_out);
  deflateEnd(&zs);
  return out;
}

static bool zlib_inflate(const uint8_t* in, size_t n, std::vector<uint8_t>& out, size_t want) {
  z_stream zs;
  memset(&zs, 0, sizeof zs);
  inflateInit2(&zs, -15);
  out.assign(want ? want : 1, 0);
  zs.next_in = (Bytef*)in; zs.avail_in = (uInt)n;
  zs.next_out = out.data(); zs.avail_out = (uInt)want;
  const int rc = inflate(&zs, Z_FINISH);
  inflateEnd(&zs);
  out.resize(want);
  return rc == Z_STREAM_END && zs.avail_out == 0;
}

int main() {
  size_t n_ok = 0, n_bad = 0, n_damaged = 0, n_damaged_accepted = 0, n_crc = 0;
  static const size_t sizes[] = {0, 1, 2, 7, 64, 287, 288, 289, 300, 1000, 4096, 65280, 65536};
  for (int kind = 0; kind < 6; ++kind)
    for (size_t n : sizes)
      for (int level : {0, 1, 4, 6, 9})
        for (int strategy : {Z_DEFAULT_STRATEGY, Z_FIXED, Z_HUFFMAN_ONLY, Z_RLE}) {
          const std::vector<uint8_t> d = make_data(kind, n);
          const std::vector<uint8_t> z = deflate_raw(d, level, strategy);
          // exact-size heap buffers: the sanitizer sees any byte outside them; the decoder may READ 64 bytes past the input
          std::vector<uint8_t> in(z.size() + 64, 0);
          memcpy(in.data(), z.data(), z.size());
          std::vector<uint8_t> out(n ? n : 1, 0xEE);
          const bool ok = dmxz::inflate_raw(in.data(), z.size(), out.data(), n);
          if (ok && (n == 0 || memcmp(out.data(), d.data(), n) == 0)) ++n_ok;
          else { ++n_bad; fprintf(stderr, "MISMATCH kind %d n %zu level %d strategy %d ok %d\n", kind, n, level, strategy, (int)ok); }
          // a wrong expected size must be refused (too short and too long)
          if (n > 0) {
            std::vector<uint8_t> o2(n + 8, 0);
            if (dmxz::inflate_raw(in.data(), z.size(), o2.data(), n - 1)) { ++n_bad; fprintf(stderr, "accepted n-1\n"); }
            if (dmxz::inflate_raw(in.data(), z.size(), o2.data(), n + 1)) { ++n_bad; fprintf(stderr, "accepted n+1\n"); }
          }
          // damage: flipped bits, truncation.  Whatever comes out, success is only allowed with zlib's bytes.
          if (z.size() > 4 && n > 0 && n <= 4096) {
            for (int rep = 0; rep < 24; ++rep) {
              std::vector<uint8_t> bad(in);
              size_t zl = z.size();
              if (rep % 3 == 0) zl = 1 + rnd() % (z.size() - 1);
              else for (int f = 0; f <= rep % 3; ++f) bad[rnd() % z.size()] ^= (uint8_t)(1u << (rnd() & 7));
              if (zl < z.size()) memset(bad.data() + zl, 0, bad.size() - zl);
              std::vector<uint8_t> o3(n, 0xEE), zo;
              const bool mine = dmxz::inflate_raw(bad.data(), zl, o3.data(), n);
              ++n_damaged;
              if (mine) {
                ++n_damaged_accepted;
                const bool theirs = zlib_inflate(bad.data(), zl, zo, n);
                if (!theirs || memcmp(zo.data(), o3.data(), n) != 0) { ++n_bad; fprintf(stderr, "damaged stream accepted with other bytes than zlib (kind %d n %zu)\n", kind, n); }
              }
            }
          }
        }
  // Hand-made dynamic blocks whose distance code is INCOMPLETE: one code of length L.  zlib (inftrees.c) takes L = 1 only; so must we
  // (the builder used to take any single code).  Stream: 'a', <length 3, distance 1>, end of block -> "aaaa".
  for (int L = 1; L <= 3; ++L) {
    std::vector<uint8_t> z(64 + 64, 0);
    size_t bitpos = 0;
    auto put = [&](uint32_t v, int nb) { for (int i = 0; i < nb; ++i, ++bitpos) if ((v >> i) & 1u) z[bitpos >> 3] |= (uint8_t)(1u << (bitpos & 7)); };   // LSB first (header fields, extra bits)
    auto code = [&](uint32_t c, int nb) { for (int i = nb - 1; i >= 0; --i, ++bitpos) if ((c >> i) & 1u) z[bitpos >> 3] |= (uint8_t)(1u << (bitpos & 7)); }; // Huffman codes: MSB first
    put(1, 1); put(2, 2);                 // BFINAL, BTYPE = dynamic
    put(258 - 257, 5); put(0, 5); put(18 - 4, 4);
    static const int order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    int cl_len[19] = {0};
    cl_len[18] = 1; cl_len[1] = 2; cl_len[2] = 2; cl_len[3] = 0;
    // code-length code: 18 -> 0, 1 -> 10, 2 -> 11; L = 3 needs symbol 3 as well: 18 -> 0, 1 -> 10, 2 -> 110, 3 -> 111
    if (L == 3) { cl_len[2] = 3; cl_len[3] = 3; }
    for (int i = 0; i < 18; ++i) put((uint32_t)cl_len[order[i]], 3);
    auto zeros = [&](int n) { code(0, 1); put((uint32_t)(n - 11), 7); };                       // symbol 18: 11..138 zeros
    auto lenv = [&](int v) { if (v == 1) code(2, 2); else if (v == 2) code(L == 3 ? 6 : 3, L == 3 ? 3 : 2); else code(7, 3); };
    zeros(97); lenv(1); zeros(138); zeros(20); lenv(2); lenv(2);                               // 'a': 1 bit, 256 and 257: 2 bits
    lenv(L);                                                                                     // the one distance code
    code(0, 1); code(3, 2); code(0, L); code(2, 2);                                             // 'a', length 3, distance 1, end of block
    const size_t zl = (bitpos + 7) / 8;
    std::vector<uint8_t> mine_out(4, 0xEE), zo;
    const bool mine = dmxz::inflate_raw(z.data(), zl, mine_out.data(), 4);
    const bool theirs = zlib_inflate(z.data(), zl, zo, 4);
    if (theirs != (L == 1)) { ++n_bad; fprintf(stderr, "hand-made incomplete distance code L=%d: zlib says %d (the test's stream is wrong)\n", L, (int)theirs); }
    if (mine != theirs || (mine && memcmp(mine_out.data(), "aaaa", 4) != 0)) { ++n_bad; fprintf(stderr, "incomplete distance code L=%d: ours %d, zlib %d\n", L, (int)mine, (int)theirs); }
    else ++n_ok;
  }
  // CRC
  for (size_t n : {(size_t)0, (size_t)1, (size_t)15, (size_t)63, (size_t)64, (size_t)65, (size_t)79, (size_t)80, (size_t)1000, (size_t)65536, (size_t)100003}) {
    const std::vector<uint8_t> d = make_data(0, n + 3);
    for (size_t o = 0; o < 3; ++o) {
      const uint32_t want = (uint32_t)crc32(crc32(0L, Z_NULL, 0), d.data() + o, (uInt)n);
      if (dmxz::crc32_of(d.data() + o, n) != want) { ++n_bad; fprintf(stderr, "crc mismatch n %zu\n", n); }
      ++n_crc;
    }
  }
  printf("streams ok %zu, failures %zu, damaged %zu (accepted with zlib's bytes %zu), crc checks %zu\n", n_ok, n_bad, n_damaged, n_damaged_accepted, n_crc);
  return n_bad ? 1 : 0;
}
