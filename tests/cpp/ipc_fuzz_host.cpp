// ipc_fuzz_host.cpp — TEST INFRASTRUCTURE: byte-mutation fuzz of the HOST-side decoders of the C ABI, run against the
// AddressSanitizer + UBSan build of the library (`make -C arrow-rs_amd/csrc SAN=1`) on a CPU-only box.
//
// The reference runs Miri over the crates that parse untrusted bytes (.github/workflows/miri.sh:12-45) and its IPC
// reader goes through the flatbuffers verifier (arrow-ipc/src/reader.rs:944-1010).  Here the flatbuffer reader is
// hand-written (csrc/ipc.hip FbView): every read must be bounds-checked, so a corrupt or hostile message may only ever
// produce an error status — never a crash, an out-of-bounds read, an overflow or a giant allocation.
//
// What is fed (ctx == NULL: these entry points are host-only by contract):
//   ah_ipc_message_info / ah_ipc_decode_schema   framed Schema messages built by ah_ipc_schema_message, then mutated
//   ah_ipc_decode_footer                         file footers built by ah_ipc_file_footer, then mutated
// Mutations: byte flips, 32-bit field overwrites with boundary values (0, -1, INT_MAX, lengths past the end),
// truncations at every prefix, random splices, and pure noise.  Statuses are tallied; the run fails if a decoder
// ACCEPTS a message and then hands back fields it cannot have read (names / formats outside the block).
//
// usage: ipc_fuzz_host <iterations> <seed>      (prints a tally; exit 0 = no sanitizer report, no inconsistency)
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/arrow_hip.h"

static uint64_t rng_state = 88172645463325252ull;
static uint64_t rnd() {  // xorshift64*
  rng_state ^= rng_state >> 12;
  rng_state ^= rng_state << 25;
  rng_state ^= rng_state >> 27;
  return rng_state * 2685821657736338717ull;
}
static uint32_t rnd_below(uint32_t n) { return n ? (uint32_t)(rnd() % n) : 0; }

static const int32_t BOUNDARY[] = {0, 1, -1, 2, 4, 7, 8, 16, 255, 256, 65535, 65536, 0x7FFFFFFF, (int32_t)0x80000000, 0x7FFFFFF0, -8, -16, 1 << 20, 1 << 30};

static void mutate(std::vector<uint8_t>& m) {
  if (m.empty()) return;
  switch (rnd_below(7)) {
    case 0:  // flip a few bytes
      for (int k = 1 + (int)rnd_below(4); k > 0; --k) m[rnd_below((uint32_t)m.size())] ^= (uint8_t)(1u << rnd_below(8));
      break;
    case 1: {  // overwrite an aligned 32-bit word with a boundary value (offsets, lengths, vtable entries)
      if (m.size() < 4) break;
      const size_t at = (size_t)rnd_below((uint32_t)(m.size() / 4)) * 4;
      const int32_t v = BOUNDARY[rnd_below(sizeof BOUNDARY / sizeof BOUNDARY[0])];
      memcpy(&m[at], &v, 4);
      break;
    }
    case 2: {  // overwrite a 16-bit word (vtable slots)
      if (m.size() < 2) break;
      const size_t at = (size_t)rnd_below((uint32_t)(m.size() / 2)) * 2;
      const uint16_t v = (uint16_t)BOUNDARY[rnd_below(sizeof BOUNDARY / sizeof BOUNDARY[0])];
      memcpy(&m[at], &v, 2);
      break;
    }
    case 3:  // truncate
      m.resize(rnd_below((uint32_t)m.size()));
      break;
    case 4: {  // splice a random run of bytes
      const size_t at = rnd_below((uint32_t)m.size()), n = 1 + rnd_below(16);
      for (size_t i = at; i < at + n && i < m.size(); ++i) m[i] = (uint8_t)rnd();
      break;
    }
    case 5: {  // a random unaligned 32-bit value anywhere
      if (m.size() < 4) break;
      const size_t at = rnd_below((uint32_t)(m.size() - 3));
      const uint32_t v = (uint32_t)rnd();
      memcpy(&m[at], &v, 4);
      break;
    }
    default: {  // duplicate a chunk over another place (self-referential offsets)
      if (m.size() < 16) break;
      const size_t n = 4 + rnd_below(12), src = rnd_below((uint32_t)(m.size() - n)), dst = rnd_below((uint32_t)(m.size() - n));
      memmove(&m[dst], &m[src], n);
    }
  }
}

int main(int argc, char** argv) {
  const long iters = argc > 1 ? atol(argv[1]) : 20000;
  if (argc > 2) rng_state ^= (uint64_t)atoll(argv[2]) * 0x9E3779B97F4A7C15ull;
  // seed messages: schemas of several shapes, and a footer with blocks
  const ah_ipc_field f1[] = {{"a", "l", 1}, {"b", "g", 0}, {"name with spaces", "u", 1}, {"ts", "tsu:UTC", 1}, {"d", "d:38,10", 1},
                             {"flag", "b", 1}, {"big", "U", 1}, {"t32", "ttm", 0}, {"i8", "c", 1}, {"dur", "tDn", 1}};
  const ah_ipc_field f2[] = {{"x", "i", 1}};
  std::vector<std::vector<uint8_t>> schemas, footers;
  for (int shape = 0; shape < 3; ++shape) {
    for (int32_t align : {8, 64}) {
      uint8_t* out = nullptr;
      int64_t n = 0;
      const ah_status st = shape == 0 ? ah_ipc_schema_message(nullptr, 10, f1, align, &out, &n)
                                      : shape == 1 ? ah_ipc_schema_message(nullptr, 1, f2, align, &out, &n)
                                                   : ah_ipc_schema_message(nullptr, 0, nullptr, align, &out, &n);
      if (st != AH_OK || !out) {
        fprintf(stderr, "could not build a seed schema message (status %d)\n", st);
        return 2;
      }
      schemas.emplace_back(out, out + n);
      ah_host_free(out);
    }
  }
  const ah_ipc_block blocks[] = {{8, 256, 0, 1024}, {1288, 256, 0, 4096}, {5640, 320, 0, 0}};
  for (int nb : {0, 3}) {
    uint8_t* out = nullptr;
    int64_t n = 0;
    if (ah_ipc_file_footer(nullptr, 10, f1, nb, blocks, &out, &n) != AH_OK || !out) {
      fprintf(stderr, "could not build a seed footer\n");
      return 2;
    }
    footers.emplace_back(out, out + n);
    ah_host_free(out);
  }
  std::map<int, long> tally;
  long accepted_schema = 0, accepted_footer = 0;
  for (long it = 0; it < iters; ++it) {
    const bool footer = (it % 3) == 2;
    std::vector<uint8_t> m = footer ? footers[rnd_below((uint32_t)footers.size())] : schemas[rnd_below((uint32_t)schemas.size())];
    if (it % 50 == 49) {  // pure noise of a random size now and then
      m.resize(rnd_below(300));
      for (auto& b : m) b = (uint8_t)rnd();
    } else if (it >= 8) {  // (the first iterations run the unmutated seeds: they must be accepted)
      for (int k = 1 + (int)rnd_below(3); k > 0; --k) mutate(m);
    }
    // an exact-size heap copy: ASan flags any read past the end
    uint8_t* buf = (uint8_t*)malloc(m.size() ? m.size() : 1);
    memcpy(buf, m.data(), m.size());
    if (!footer) {
      int32_t ht = 0;
      int64_t body = 0;
      tally[ah_ipc_message_info(nullptr, buf, (int64_t)m.size(), &ht, &body)]++;
      int32_t nf = 0;
      ah_ipc_field* fields = nullptr;
      const ah_status st = ah_ipc_decode_schema(nullptr, buf, (int64_t)m.size(), &nf, &fields);
      tally[1000 + st]++;
      if (st == AH_OK) {
        ++accepted_schema;
        if (nf < 0 || (nf > 0 && !fields)) {
          fprintf(stderr, "decode_schema accepted a message but returned %d fields at %p\n", nf, (void*)fields);
          return 1;
        }
        size_t total = 0;
        for (int i = 0; i < nf; ++i) total += strlen(fields[i].name) + strlen(fields[i].format);  // ASan checks the reads
        (void)total;
        ah_host_free(fields);
      } else if (it < 8) {
        fprintf(stderr, "an unmutated schema message was rejected (status %d)\n", st);
        return 1;
      }
    } else {
      int64_t flen = 0;
      int32_t nf = 0, nb = 0;
      ah_ipc_field* fields = nullptr;
      ah_ipc_block* bl = nullptr;
      const ah_status st = ah_ipc_decode_footer(nullptr, buf, (int64_t)m.size(), &flen, &nf, &fields, &nb, &bl);
      tally[2000 + st]++;
      if (st == AH_OK) {
        ++accepted_footer;
        uint64_t sum = 0;  // (touch every returned byte: ASan checks the reads)
        for (int i = 0; i < nb; ++i) sum += (uint64_t)bl[i].offset + (uint64_t)bl[i].body_length;
        for (int i = 0; i < nf; ++i) sum += strlen(fields[i].name);
        (void)sum;
        ah_host_free(fields);
        ah_host_free(bl);
      }
    }
    free(buf);
  }
  printf("ipc_fuzz_host: %ld iterations, %ld schemas and %ld footers still accepted after mutation\n", iters, accepted_schema, accepted_footer);
  for (auto& kv : tally)
    printf("  %s status %d: %ld\n", kv.first >= 2000 ? "decode_footer" : kv.first >= 1000 ? "decode_schema" : "message_info", kv.first % 1000, kv.second);
  printf("IPC_FUZZ_OK\n");
  return 0;
}
