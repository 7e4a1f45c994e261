// oracle/ref_dump.cc -- TEST INFRASTRUCTURE (not product code).
//
// Compiles against the reference's *data-only* headers where they lie under
// /root/reference (no stand-in headers are written: every header included here
// depends only on the C++ standard library) and dumps, as JSON on stdout:
//   * the gen-1 move / species / type-chart / boost / PP tables
//       (cpp/include/libpkmn/data/{moves,species,types,boosts}.h)
//   * the OU learnsets (cpp/include/format/ou/data.h)
//   * known-answer streams of the reference's two device RNGs and the engine LCG
//       (cpp/include/util/random.h:10-133, cpp/include/libpkmn/rng.h:9-11)
// The JSON is data (public game facts + RNG known answers); tools/gen_tables.py
// turns it into this repo's own table format.  Output binary goes to oracle/_ref/.
#include <cstdio>
#include <cstdint>
#include <format/ou/data.h>
#include <libpkmn/data/boosts.h>
#include <libpkmn/data/moves.h>
#include <libpkmn/data/species.h>
#include <libpkmn/data/types.h>
#include <libpkmn/rng.h>
#include <util/random.h>

int main() {
  using namespace PKMN::Data;
  std::printf("{\n\"moves\": [");
  for (int i = 0; i < 165; ++i) {
    const auto &m = MOVE_DATA[i];
    std::printf("%s[%d,%d,%d,%d,%d,%d]", i ? "," : "", (int)m.effect, (int)m.bp,
                (int)m.type, (int)m.accuracy, (int)m.target, (int)PP[i]);
  }
  std::printf("],\n\"max_pp\": [");
  for (int i = 1; i <= 165; ++i)
    std::printf("%s%d", i > 1 ? "," : "", (int)max_pp(static_cast<Move>(i)));
  std::printf("],\n\"species\": [");
  for (int i = 0; i < 151; ++i) {
    const auto &s = SPECIES_DATA[i];
    std::printf("%s[%d,%d,%d,%d,%d,%d,%d]", i ? "," : "", s.base_stats.hp,
                s.base_stats.atk, s.base_stats.def, s.base_stats.spe,
                s.base_stats.spc, (int)s.types[0], (int)s.types[1]);
  }
  std::printf("],\n\"type_chart\": [");
  for (int a = 0; a < 15; ++a) {
    std::printf("%s[", a ? "," : "");
    for (int d = 0; d < 15; ++d)
      std::printf("%s%d", d ? "," : "", (int)TYPE_CHART[a][d]);
    std::printf("]");
  }
  std::printf("],\n\"boosts\": [");
  for (int i = 0; i < 13; ++i)
    std::printf("%s[%d,%d]", i ? "," : "", boosts[i][0], boosts[i][1]);
  std::printf("],\n\"ou_legal_species\": [");
  {
    bool first = true;
    for (const auto s : Format::OU::legal_species) {
      std::printf("%s%d", first ? "" : ",", (int)s);
      first = false;
    }
  }
  std::printf("],\n\"ou_move_pools\": {");
  {
    bool first = true;
    for (int s = 1; s <= 151; ++s) {
      const int n = Format::OU::move_pool_size(static_cast<Species>(s));
      if (!n) continue;
      std::printf("%s\"%d\": [", first ? "" : ",", s);
      first = false;
      const auto &pool = Format::OU::move_pool(static_cast<Species>(s));
      for (int k = 0; k < n; ++k)
        std::printf("%s%d", k ? "," : "", (int)pool[k]);
      std::printf("]");
    }
  }
  std::printf("},\n");

  // RNG known answers -------------------------------------------------------
  const uint64_t seeds[] = {1111111ull, 0x123456ull, 0ull, 0xDEADBEEFCAFEF00Dull,
                            0x0A4B00000000ull, 0x0A4B00000001ull, 7ull};
  std::printf("\"mt19937_uniform_64\": {");
  for (size_t k = 0; k < 3; ++k) {
    mt19937 dev{static_cast<uint32_t>(seeds[k])};
    std::printf("%s\"%u\": [", k ? "," : "", (unsigned)seeds[k]);
    for (int i = 0; i < 16; ++i)
      std::printf("%s\"%llu\"", i ? "," : "", (unsigned long long)dev.uniform_64());
    std::printf("]");
  }
  std::printf("},\n\"mt19937_uniform\": {");
  for (size_t k = 0; k < 2; ++k) {
    mt19937 dev{static_cast<uint32_t>(seeds[k])};
    std::printf("%s\"%u\": [", k ? "," : "", (unsigned)seeds[k]);
    for (int i = 0; i < 8; ++i) std::printf("%s%.17g", i ? "," : "", dev.uniform());
    std::printf("]");
  }
  std::printf("},\n\"fast_prng\": {");
  for (size_t k = 0; k < sizeof(seeds) / sizeof(seeds[0]); ++k) {
    uint8_t buf[8];
    fast_prng::seed(buf, seeds[k]);
    std::printf("%s\"%llu\": {\"state\": [", k ? "," : "", (unsigned long long)seeds[k]);
    for (int i = 0; i < 8; ++i) std::printf("%s%d", i ? "," : "", buf[i]);
    fast_prng dev{buf};
    std::printf("], \"uniform_64\": [");
    for (int i = 0; i < 16; ++i)
      std::printf("%s\"%llu\"", i ? "," : "", (unsigned long long)dev.uniform_64());
    std::printf("], \"random_int_9\": [");
    for (int i = 0; i < 8; ++i) std::printf("%s%d", i ? "," : "", dev.random_int(9));
    std::printf("], \"uniform\": [");
    for (int i = 0; i < 4; ++i) std::printf("%s%.17g", i ? "," : "", dev.uniform());
    std::printf("]}");
  }
  std::printf("},\n\"lcg\": {");
  for (size_t k = 0; k < 4; ++k) {
    uint64_t s = seeds[k];
    std::printf("%s\"%llu\": [", k ? "," : "", (unsigned long long)seeds[k]);
    for (int i = 0; i < 8; ++i) {
      PKMN::RNG::next(s);
      std::printf("%s\"%llu\"", i ? "," : "", (unsigned long long)s);
    }
    std::printf("]");
  }
  std::printf("}\n}\n");
  return 0;
}
