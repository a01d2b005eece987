// dmx_inflate.hpp — a raw-DEFLATE (RFC 1951) decoder and a CRC-32 for BGZF members (SAM spec 4.1), written for this scanner.
//
// Why: the `demuxlet` binary reads BAM / bgzipped VCF without htslib, and zlib 1.2.11's inflate + crc32 were two thirds of the
// host CPU time of the BAM x VCF scan (170 MB/s + 1 GB/s per core on the bench BAM: 3.0 of 4.7 CPU-seconds per 2e6 reads).
// A BGZF member is small (<= 64 KiB in, <= 64 KiB out), its output size is known before decoding (ISIZE) and its CRC-32 is
// checked afterwards, so the decoder can be simple about memory (one flat output window, no streaming state) and the caller can
// afford a safety net: whenever this decoder reports failure, or the CRC / ISIZE disagree, the member is handed to zlib, whose
// verdict is final.  Nothing here is taken from zlib or libdeflate sources; the techniques are the textbook ones (RFC 1951 §3.2,
// canonical codes into bit-reversed lookup tables with second-level tables for long codes, a 64-bit bit buffer refilled without
// branches, word-wise match copies; CRC by carry-less multiplication folding, Gopal et al., Intel 2009).
#pragma once
#include <cstdint>
#include <cstring>
#include <zlib.h>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace dmxz {

// ---- table entries ---------------------------------------------------------------------------------------------------------
// bits 0-7 bits to consume (a length / distance base: code AND extra bits, so that the next lookup waits for one shift only) |
// 8-12 extra-bit count (or second-level index width) | 13-15 kind | 16-31 payload
enum : uint32_t { K_BAD = 0, K_LIT = 1, K_BASE = 2, K_EOB = 3, K_SUB = 4 };
inline uint32_t mk(uint32_t kind, uint32_t nbits, uint32_t extra, uint32_t payload) { return nbits | (extra << 8) | (kind << 13) | (payload << 16); }
inline uint32_t e_bits(uint32_t e) { return e & 0xffu; }
inline uint32_t e_extra(uint32_t e) { return (e >> 8) & 31u; }
inline uint32_t e_kind(uint32_t e) { return (e >> 13) & 7u; }
inline uint32_t e_val(uint32_t e) { return e >> 16; }

constexpr int kLitRoot = 10, kDistRoot = 8;
constexpr int kLitCap = 2048, kDistCap = 1024;   // 1334 / 402 suffice for complete codes; the builder checks

struct Tables { uint32_t lit[kLitCap]; uint32_t dist[kDistCap]; };

inline uint32_t bitrev(uint32_t c, int n) {
  uint32_t r = 0;
  for (int i = 0; i < n; ++i) { r = (r << 1) | (c & 1u); c >>= 1; }
  return r;
}

// Canonical code of `n` symbols with lengths len[] (0 = unused) -> lookup table.  leaf(sym, codelen) makes a symbol's entry.
// Returns false for an over-subscribed or (unless it is the one-code distance case) incomplete code.
template <class Leaf>
inline bool build(const uint8_t* len, int n, int root, uint32_t* tab, int cap, bool allow_incomplete, Leaf leaf) {
  int count[16] = {0};
  for (int i = 0; i < n; ++i) ++count[len[i]];
  count[0] = 0;
  int left = 1, used = 0;
  for (int l = 1; l <= 15; ++l) { left = left * 2 - count[l]; if (left < 0) return false; used += count[l]; }
  // zlib (inftrees.c) accepts an incomplete set only with no code at all or ONE code of length 1; anything else goes to zlib, which rejects it
  if (left > 0 && !(allow_incomplete && (used == 0 || (used == 1 && count[1] == 1)))) return false;
  uint32_t next[16]; uint32_t code = 0;
  for (int l = 1; l <= 15; ++l) { code = (code + (uint32_t)count[l - 1]) << 1; next[l] = code; }
  const uint32_t rmask = (1u << root) - 1u;
  for (uint32_t i = 0; i <= rmask; ++i) tab[i] = 0;
  // pass 1: the longest code under every root prefix that has long codes
  uint8_t submax[1u << kLitRoot];
  bool any_long = false;
  for (int l = root + 1; l <= 15; ++l) any_long |= count[l] != 0;
  if (any_long) memset(submax, 0, (size_t)rmask + 1);
  uint32_t nx[16];
  memcpy(nx, next, sizeof nx);
  if (any_long)
    for (int s = 0; s < n; ++s) {
      const int l = len[s];
      if (!l) continue;
      const uint32_t c = nx[l]++;
      if (l > root) { const uint32_t p = bitrev(c, l) & rmask; if (submax[p] < l) submax[p] = (uint8_t)l; }
    }
  int top = (int)rmask + 1;
  for (int s = 0; s < n; ++s) {
    const int l = len[s];
    if (!l) continue;
    const uint32_t r = bitrev(next[l]++, l);
    if (l <= root) {
      const uint32_t e = leaf(s, l);
      for (uint32_t i = r; i <= rmask; i += 1u << l) tab[i] = e;
    } else {
      const uint32_t p = r & rmask;
      if (e_kind(tab[p]) != K_SUB) {
        const int sb = submax[p] - root;
        if (top + (1 << sb) > cap) return false;
        for (int i = 0; i < (1 << sb); ++i) tab[top + i] = 0;
        tab[p] = mk(K_SUB, (uint32_t)root, (uint32_t)sb, (uint32_t)top);
        top += 1 << sb;
      }
      const uint32_t sb = e_extra(tab[p]), base = e_val(tab[p]);
      const uint32_t e = leaf(s, l - root);
      for (uint32_t i = r >> root; i < (1u << sb); i += 1u << (l - root)) tab[base + i] = e;
    }
  }
  return true;
}

inline uint32_t lit_leaf(int s, int l) {
  static const uint16_t lbase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
  static const uint8_t lext[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
  if (s < 256) return mk(K_LIT, (uint32_t)l, 0, (uint32_t)s);
  if (s == 256) return mk(K_EOB, (uint32_t)l, 0, 0);
  if (s > 285) return mk(K_BAD, (uint32_t)l, 0, 0);            // 286, 287 take part in the fixed code but never occur
  return mk(K_BASE, (uint32_t)l + lext[s - 257], lext[s - 257], lbase[s - 257]);
}
inline uint32_t dist_leaf(int s, int l) {
  static const uint16_t dbase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
  static const uint8_t dext[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
  if (s > 29) return mk(K_BAD, (uint32_t)l, 0, 0);
  return mk(K_BASE, (uint32_t)l + dext[s], dext[s], dbase[s]);
}

inline uint64_t load64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }

// Inflates one raw DEFLATE stream of `in_len` bytes at `in` into exactly `out_len` bytes at `out`.  THE INPUT BUFFER MUST BE
// READABLE FOR 64 BYTES PAST in + in_len (a BGZF member carries 8 there; the reader pads 56 more).  Nothing is written outside
// [out, out + out_len).  Returns true when the stream ended with its final block exactly at out + out_len.
inline bool inflate_raw(const uint8_t* in, size_t in_len, uint8_t* out, size_t out_len) {
  const uint8_t* const in_end = in + in_len;
  const uint8_t* const in_lim = in_end + 8;          // the bit buffer may run up to 8 bytes ahead of the bytes really consumed
  uint8_t* const out0 = out;
  uint8_t* const out_end = out + out_len;
  uint8_t* const out_fast = out_len > 288 ? out_end - 288 : out;   // three literals + one longest match + a 16-byte copy's overshoot
  uint64_t bb = 0; unsigned bc = 0;
#define DMXZ_REFILL() do { bb |= load64(in) << bc; in += (63u - bc) >> 3; bc |= 56u; } while (0)
#define DMXZ_DROP(n) do { bb >>= (n); bc -= (unsigned)(n); } while (0)
  Tables T;
  static const Tables* fixed = [] {
    static Tables F;
    uint8_t l[288];
    for (int i = 0; i < 288; ++i) l[i] = i < 144 ? 8 : (i < 256 ? 9 : (i < 280 ? 7 : 8));
    build(l, 288, kLitRoot, F.lit, kLitCap, false, lit_leaf);
    uint8_t d[32];
    for (int i = 0; i < 32; ++i) d[i] = 5;
    build(d, 32, kDistRoot, F.dist, kDistCap, false, dist_leaf);
    return &F;
  }();
  for (bool last = false; !last;) {
    if (in > in_lim) return false;
    DMXZ_REFILL();
    last = bb & 1u;
    const unsigned type = (unsigned)(bb >> 1) & 3u;
    DMXZ_DROP(3);
    const uint32_t* lit; const uint32_t* dst;
    if (type == 0) {                                 // stored: to the byte boundary, LEN, ~LEN, bytes
      DMXZ_DROP(bc & 7u);
      in -= bc >> 3; bb = 0; bc = 0;
      if (in + 4 > in_end) return false;
      const unsigned n = in[0] | ((unsigned)in[1] << 8), nn = in[2] | ((unsigned)in[3] << 8);
      if ((n ^ nn) != 0xffffu) return false;
      in += 4;
      if (n > (size_t)(in_end - in) || n > (size_t)(out_end - out)) return false;
      memcpy(out, in, n);
      in += n; out += n;
      continue;
    } else if (type == 1) {
      lit = fixed->lit; dst = fixed->dist;
    } else if (type == 2) {
      const unsigned hlit = ((unsigned)bb & 31u) + 257u, hdist = ((unsigned)(bb >> 5) & 31u) + 1u, hclen = ((unsigned)(bb >> 10) & 15u) + 4u;
      DMXZ_DROP(14);
      if (hlit > 286 || hdist > 30) return false;
      static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
      uint8_t cl[19] = {0};
      for (unsigned i = 0; i < hclen; ++i) {
        if (bc < 3) { if (in > in_lim) return false; DMXZ_REFILL(); }
        cl[order[i]] = (uint8_t)(bb & 7u);
        DMXZ_DROP(3);
      }
      uint32_t ct[128];
      if (!build(cl, 19, 7, ct, 128, false, [](int s, int l) { return mk(K_LIT, (uint32_t)l, 0, (uint32_t)s); })) return false;
      uint8_t lens[286 + 30 + 138];
      unsigned i = 0;
      const unsigned total = hlit + hdist;
      while (i < total) {
        if (in > in_lim) return false;
        DMXZ_REFILL();
        const uint32_t e = ct[bb & 127u];
        if (e_kind(e) != K_LIT) return false;
        DMXZ_DROP(e_bits(e));
        const unsigned s = e_val(e);
        if (s < 16) { lens[i++] = (uint8_t)s; continue; }
        unsigned rep; uint8_t v = 0;
        if (s == 16) { if (i == 0) return false; v = lens[i - 1]; rep = 3 + ((unsigned)bb & 3u); DMXZ_DROP(2); }
        else if (s == 17) { rep = 3 + ((unsigned)bb & 7u); DMXZ_DROP(3); }
        else { rep = 11 + ((unsigned)bb & 127u); DMXZ_DROP(7); }
        if (i + rep > total) return false;
        memset(lens + i, v, rep);
        i += rep;
      }
      if (lens[256] == 0) return false;
      if (!build(lens, (int)hlit, kLitRoot, T.lit, kLitCap, false, lit_leaf)) return false;
      if (!build(lens + hlit, (int)hdist, kDistRoot, T.dist, kDistCap, true, dist_leaf)) return false;
      lit = T.lit; dst = T.dist;
    } else return false;

    // ---- symbols
    for (;;) {
      if (in > in_lim) return false;
      DMXZ_REFILL();                                 // >= 56 bits
      uint32_t e = lit[bb & ((1u << kLitRoot) - 1u)];
      if (out < out_fast) {
        // up to three literals per refill (3 x 15 bits), then anything else with >= 11 bits left gets a fresh buffer
        if (e_kind(e) == K_LIT) {
          DMXZ_DROP(e_bits(e)); *out++ = (uint8_t)e_val(e);
          e = lit[bb & ((1u << kLitRoot) - 1u)];
          if (e_kind(e) == K_LIT) {
            DMXZ_DROP(e_bits(e)); *out++ = (uint8_t)e_val(e);
            e = lit[bb & ((1u << kLitRoot) - 1u)];
            if (e_kind(e) == K_LIT) { DMXZ_DROP(e_bits(e)); *out++ = (uint8_t)e_val(e); continue; }
          }
          if (in > in_lim) return false;
          DMXZ_REFILL();
        }
        if (e_kind(e) == K_SUB) { DMXZ_DROP(kLitRoot); e = lit[e_val(e) + ((uint32_t)bb & ((1u << e_extra(e)) - 1u))]; }
        const uint32_t k = e_kind(e);
        if (k == K_LIT) { DMXZ_DROP(e_bits(e)); *out++ = (uint8_t)e_val(e); continue; }
        if (k == K_EOB) { DMXZ_DROP(e_bits(e)); break; }
        if (k != K_BASE) return false;
        const uint64_t sv = bb;
        DMXZ_DROP(e_bits(e));
        const unsigned len = e_val(e) + ((unsigned)(sv >> (e_bits(e) - e_extra(e))) & ((1u << e_extra(e)) - 1u));
        // (at most 15 + 5 = 20 bits gone since the last refill: 36 left, a distance needs up to 15 + 13)
        uint32_t d = dst[bb & ((1u << kDistRoot) - 1u)];
        if (e_kind(d) == K_SUB) { DMXZ_DROP(kDistRoot); d = dst[e_val(d) + ((uint32_t)bb & ((1u << e_extra(d)) - 1u))]; }
        if (e_kind(d) != K_BASE) return false;
        const uint64_t dv = bb;
        DMXZ_DROP(e_bits(d));
        const unsigned dist = e_val(d) + ((unsigned)(dv >> (e_bits(d) - e_extra(d))) & ((1u << e_extra(d)) - 1u));
        if (dist > (size_t)(out - out0)) return false;
        const uint8_t* s = out - dist;
        uint8_t* o = out;
        out += len;
        if (dist >= 8) {                             // (two words unconditionally: most matches of a BAM are 3-8 bytes long)
          memcpy(o, s, 8); memcpy(o + 8, s + 8, 8);
          if (len > 16) { o += 16; s += 16; do { memcpy(o, s, 8); o += 8; s += 8; } while (o < out); }
        } else if (dist == 1) {
          memset(o, *s, len);
        } else {
          do { *o++ = *s++; } while (o < out);
        }
        continue;
      }
      // ---- the last 288 bytes of the member: every store checked
      if (e_kind(e) == K_SUB) { DMXZ_DROP(kLitRoot); e = lit[e_val(e) + ((uint32_t)bb & ((1u << e_extra(e)) - 1u))]; }
      const uint32_t k = e_kind(e);
      if (k == K_LIT) { if (out >= out_end) return false; DMXZ_DROP(e_bits(e)); *out++ = (uint8_t)e_val(e); continue; }
      if (k == K_EOB) { DMXZ_DROP(e_bits(e)); break; }
      if (k != K_BASE) return false;
      const uint64_t sv = bb;
      DMXZ_DROP(e_bits(e));
      const unsigned len = e_val(e) + ((unsigned)(sv >> (e_bits(e) - e_extra(e))) & ((1u << e_extra(e)) - 1u));
      DMXZ_REFILL();
      uint32_t d = dst[bb & ((1u << kDistRoot) - 1u)];
      if (e_kind(d) == K_SUB) { DMXZ_DROP(kDistRoot); d = dst[e_val(d) + ((uint32_t)bb & ((1u << e_extra(d)) - 1u))]; }
      if (e_kind(d) != K_BASE) return false;
      const uint64_t dv = bb;
      DMXZ_DROP(e_bits(d));
      const unsigned dist = e_val(d) + ((unsigned)(dv >> (e_bits(d) - e_extra(d))) & ((1u << e_extra(d)) - 1u));
      if (dist > (size_t)(out - out0) || len > (size_t)(out_end - out)) return false;
      const uint8_t* s = out - dist;
      for (unsigned i = 0; i < len; ++i) out[i] = s[i];
      out += len;
    }
  }
#undef DMXZ_REFILL
#undef DMXZ_DROP
  return out == out_end && in - (bc >> 3) <= in_end;
}

// ---- CRC-32 (the gzip polynomial) -------------------------------------------------------------------------------------------
#if defined(__x86_64__)
// Folding by carry-less multiplication: 64 bytes per step, then 16, then Barrett reduction.  n >= 64, n % 16 == 0.  `crc` and the
// result are the register's contents (the complement of what zlib's crc32() takes and returns).
__attribute__((target("pclmul,sse4.1"))) inline uint32_t crc32_fold(const uint8_t* p, size_t n, uint32_t crc) {
  alignas(16) static const uint64_t k1k2[2] = {0x0154442bd4ull, 0x01c6e41596ull};
  alignas(16) static const uint64_t k3k4[2] = {0x01751997d0ull, 0x00ccaa009eull};
  alignas(16) static const uint64_t k5k0[2] = {0x0163cd6124ull, 0x0000000000ull};
  alignas(16) static const uint64_t poly[2] = {0x01db710641ull, 0x01f7011641ull};
  __m128i x0, x1, x2, x3, x4, x5, x6, x7, x8, y5, y6, y7, y8;
  x1 = _mm_loadu_si128((const __m128i*)(p + 0x00));
  x2 = _mm_loadu_si128((const __m128i*)(p + 0x10));
  x3 = _mm_loadu_si128((const __m128i*)(p + 0x20));
  x4 = _mm_loadu_si128((const __m128i*)(p + 0x30));
  x1 = _mm_xor_si128(x1, _mm_cvtsi32_si128((int)crc));
  x0 = _mm_load_si128((const __m128i*)k1k2);
  p += 64; n -= 64;
  while (n >= 64) {
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x6 = _mm_clmulepi64_si128(x2, x0, 0x00);
    x7 = _mm_clmulepi64_si128(x3, x0, 0x00); x8 = _mm_clmulepi64_si128(x4, x0, 0x00);
    x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x2 = _mm_clmulepi64_si128(x2, x0, 0x11);
    x3 = _mm_clmulepi64_si128(x3, x0, 0x11); x4 = _mm_clmulepi64_si128(x4, x0, 0x11);
    y5 = _mm_loadu_si128((const __m128i*)(p + 0x00)); y6 = _mm_loadu_si128((const __m128i*)(p + 0x10));
    y7 = _mm_loadu_si128((const __m128i*)(p + 0x20)); y8 = _mm_loadu_si128((const __m128i*)(p + 0x30));
    x1 = _mm_xor_si128(_mm_xor_si128(x1, x5), y5); x2 = _mm_xor_si128(_mm_xor_si128(x2, x6), y6);
    x3 = _mm_xor_si128(_mm_xor_si128(x3, x7), y7); x4 = _mm_xor_si128(_mm_xor_si128(x4, x8), y8);
    p += 64; n -= 64;
  }
  x0 = _mm_load_si128((const __m128i*)k3k4);
  x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
  x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x3), x5);
  x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x4), x5);
  while (n >= 16) {
    x2 = _mm_loadu_si128((const __m128i*)p);
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
    p += 16; n -= 16;
  }
  x2 = _mm_clmulepi64_si128(x1, x0, 0x10);
  x3 = _mm_setr_epi32(~0, 0, ~0, 0);
  x1 = _mm_srli_si128(x1, 8);
  x1 = _mm_xor_si128(x1, x2);
  x0 = _mm_loadl_epi64((const __m128i*)k5k0);
  x2 = _mm_srli_si128(x1, 4);
  x1 = _mm_and_si128(x1, x3);
  x1 = _mm_clmulepi64_si128(x1, x0, 0x00);
  x1 = _mm_xor_si128(x1, x2);
  x0 = _mm_load_si128((const __m128i*)poly);
  x2 = _mm_and_si128(x1, x3);
  x2 = _mm_clmulepi64_si128(x2, x0, 0x10);
  x2 = _mm_and_si128(x2, x3);
  x2 = _mm_clmulepi64_si128(x2, x0, 0x00);
  x1 = _mm_xor_si128(x1, x2);
  return (uint32_t)_mm_extract_epi32(x1, 1);
}
#endif

// zlib's crc32(0, p, n), by folding when the CPU can and when a self-test against zlib (once per process) agrees.
inline uint32_t crc32_of(const uint8_t* p, size_t n) {
#if defined(__x86_64__)
  static const bool use_fold = [] {
    if (!__builtin_cpu_supports("pclmul") || !__builtin_cpu_supports("sse4.1")) return false;
    uint8_t t[1024 + 13];
    uint32_t s = 12345u;
    for (size_t i = 0; i < sizeof t; ++i) { s = s * 1664525u + 1013904223u; t[i] = (uint8_t)(s >> 24); }
    for (size_t n16 : {(size_t)64, (size_t)80, (size_t)128, (size_t)496, (size_t)1024}) {
      const uint32_t want = (uint32_t)crc32(crc32(0L, Z_NULL, 0), t + 5, (uInt)n16);
      if (~crc32_fold(t + 5, n16, ~0u) != want) return false;
    }
    return true;
  }();
  if (use_fold && n >= 64) {
    const size_t n16 = n & ~(size_t)15;
    const uint32_t c = ~crc32_fold(p, n16, ~0u);
    return n16 == n ? c : (uint32_t)crc32(c, p + n16, (uInt)(n - n16));
  }
#endif
  return (uint32_t)crc32(crc32(0L, Z_NULL, 0), p, (uInt)n);
}

}  // namespace dmxz
