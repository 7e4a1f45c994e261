// oracle/ref_oakside_dump.cc -- TEST INFRASTRUCTURE (fixture generator; runs only in the build container).
//
// Runs the reference's own header-only OAK-SIDE code of the path -- the code that sits between libpkmn and the network and
// never calls into libpkmn -- on states handed to it, and prints what it computed as JSON:
//   * Encode::Battle::Pokemon::write / ActivePokemon::write, sparse form   (cpp/include/encode/battle/battle.h:208-214,544-551)
//     called with the arguments NetworkImpl::write_battle_embedding and the two caches give them
//     (cpp/include/nn/battle/network.h:131-175, nn/battle/cache.h:94-101,187-199), and the hp fractions of network.h:146,164
//   * Encode::Battle::pokemon_key                                            (cpp/include/encode/battle/key.h:65-71)
//   * Encode::Battle::Policy::get_index                                      (cpp/include/encode/battle/policy.h:29-58)
//   * MCTS::randomize_hidden_variables                                       (cpp/include/search/durations.h:25-97)
//   * PokeEngine::evaluate_battle / Eval::evaluate                           (cpp/include/search/poke-engine-evaluate.h:184-204)
//   * PKMN::battle (Init::init_side / init_pokemon / compute_stat)           (cpp/include/libpkmn/pkmn.h:50-57, init.h:90-154)
//   * PKMN::string_to_species / string_to_move: the prefix matching parse_battle's words go through (libpkmn/strings.h:53-83,
//     313-331)
//   * PKMN::battle_data_to_string, pyoak's battle_string                    (libpkmn/strings.h:187-303)
//   * PKMN::result(battle), the request byte recomputed from a state        (cpp/include/libpkmn/pkmn.h:235-272)
//   * Train::Battle::CompressedFrames::write / Update::write / compress_probs (cpp/include/train/battle/compressed-frame.h:11-25,
//     48-57,84-118,181-214): the `.battle.data` record of a game, from search outputs handed in as doubles
//
// HOW IT IS COMPILED, stated plainly: every header above includes <pkmn.h>, libpkmn's GENERATED C header, which this checkout
// does not have.  The include path therefore names ../include, i.e. the PRODUCT's own boundary header include/pkmn.h -- the
// header a maintainer building Oak against liboakgpu.so would compile with.  It supplies POD typedefs and
// declarations only; this program calls NO pkmn_* function (none is linked), so nothing of the engine restatement can leak
// into these fixtures: they are outputs of the reference's own arithmetic on given bytes.  This is an integration build of the
// reference's headers against the drop-in header, not a build of libpkmn, and it pins nothing at the libpkmn boundary.
//
// NOT covered, because it cannot be: Parse::parse_battle (util/parse.h) includes util/strings.h -> search/mcts.h -> the
// network headers -> Eigen, which is absent; no stand-in is written for it (oak_amd/parse.py stays pinned only by the positions
// of search-test.cc / TUTORIAL.md it has to reproduce).
//
// usage: ref_oakside_dump states <file>   records of 400 B: battle[384] durations[8] seed_le[8]
//        ref_oakside_dump teams  <file>   records of 68 B: 2 x 6 x (species, move[4]) then seed_le[8]
//        ref_oakside_dump names  <file>   one token per line -> [species index or -1, move index or -1]
//        ref_oakside_dump frames <file>   games: battle[384] result[1] count_le[2] then count x { m n c1 c2 iterations_le[4]
//                                         double empirical_value nash_value p1_empirical[9] p1_nash[9] p2_empirical[9] p2_nash[9] }
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include <encode/battle/battle.h>
#include <encode/battle/key.h>
#include <encode/battle/policy.h>
#include <libpkmn/pkmn.h>
#include <search/durations.h>
#include <search/poke-engine-evaluate.h>
#include <train/battle/compressed-frame.h>

namespace {
std::vector<uint8_t> slurp(const char *path) {
  std::vector<uint8_t> v;
  FILE *f = fopen(path, "rb");
  if (!f) { perror(path); exit(2); }
  uint8_t buf[4096]; size_t n;
  while ((n = fread(buf, 1, sizeof buf, f)) > 0) v.insert(v.end(), buf, buf + n);
  fclose(f);
  return v;
}
void hex(const uint8_t *p, size_t n) { putchar('"'); for (size_t i = 0; i < n; ++i) printf("%02x", p[i]); putchar('"'); }
void sparse(const char *name, const float *val, const uint16_t *idx, size_t n) {
  printf("\"%s_idx\":[", name);
  for (size_t i = 0; i < n; ++i) printf("%s%u", i ? "," : "", idx[i]);
  printf("],\"%s_val\":[", name);
  for (size_t i = 0; i < n; ++i) printf("%s%.9g", i ? "," : "", val[i]);
  printf("]");
}

void dump_state(const uint8_t *rec, bool first) {
  pkmn_gen1_battle b; pkmn_gen1_chance_durations d; uint64_t seed;
  memcpy(&b, rec, 384); memcpy(&d, rec + 384, 8); memcpy(&seed, rec + 392, 8);
  const auto &battle = PKMN::view(b);
  const auto &durations = PKMN::view(d);
  printf("%s{", first ? "" : ",\n");
  printf("\"sides\":[");
  for (int s = 0; s < 2; ++s) {
    const auto &side = battle.sides[s];
    const auto &duration = durations.get(s);
    const auto &stored = side.stored();
    printf("%s{", s ? "," : "");
    // active slot: network.h:142-152, cache.h:187-199
    if (stored.hp == 0) printf("\"active\":null");
    else {
      std::array<uint16_t, Encode::Battle::ActivePokemon::n_dim> idx{};
      std::array<float, Encode::Battle::ActivePokemon::n_dim> val{};
      float *t = val.data(); uint16_t *ix = idx.data();
      Encode::Battle::ActivePokemon::write(stored, side.active, duration, t, ix);
      printf("\"active\":{\"hp\":%.9g,\"key\":%u,", (float)stored.hp / stored.stats.hp,
             (unsigned)Encode::Battle::pokemon_key(stored, duration.sleep(0)));
      sparse("e", val.data(), idx.data(), (size_t)(t - val.data()));
      printf("}");
    }
    // party slots 2..6: network.h:154-172, cache.h:94-101
    printf(",\"slots\":[");
    for (int slot = 2; slot <= 6; ++slot) {
      const auto id = side.order[slot - 1];
      printf("%s", slot > 2 ? "," : "");
      if (id == 0 || side.pokemon[id - 1].hp == 0) { printf("null"); continue; }
      const auto &pokemon = side.pokemon[id - 1];
      const auto sleep = duration.sleep(slot - 1);
      std::array<uint16_t, Encode::Battle::Pokemon::n_dim> idx{};
      std::array<float, Encode::Battle::Pokemon::n_dim> val{};
      float *t = val.data(); uint16_t *ix = idx.data();
      Encode::Battle::Pokemon::write(pokemon, sleep, t, ix);
      printf("{\"hp\":%.9g,\"key\":%u,", (float)pokemon.hp / pokemon.stats.hp, (unsigned)Encode::Battle::pokemon_key(pokemon, sleep));
      sparse("e", val.data(), idx.data(), (size_t)(t - val.data()));
      printf("}");
    }
    printf("]");
    // policy index of every well-formed choice byte: policy.h:29-58
    printf(",\"policy\":[");
    bool any = false;
    for (int m = 1; m <= 4; ++m) { const uint8_t c = (uint8_t)(1 | (m << 2)); printf("%s[%u,%u]", any ? "," : "", c, Encode::Battle::Policy::get_index(side, c)); any = true; }
    for (int sl = 2; sl <= 6; ++sl) {
      if (side.order[sl - 1] == 0) continue;
      const uint8_t c = (uint8_t)(2 | (sl << 2));
      printf(",[%u,%u]", c, Encode::Battle::Policy::get_index(side, c));
    }
    printf("]}");
  }
  printf("]");
  // hidden-variable resampling: durations.h:25-97, as run_root_iteration arms it (mcts.h:254-257)
  {
    pkmn_gen1_battle r = b;
    PKMN::view(r).rng = seed;
    MCTS::randomize_hidden_variables(r, d);
    printf(",\"randomized\":"); hex(reinterpret_cast<const uint8_t *>(&r), 384);
  }
  // PokeEngine: score and the value against this state's own root score (poke-engine-evaluate.h:184-204)
  {
    PokeEngine::Eval e{};
    e.get_root_score(b);
    printf(",\"result\":%u", (unsigned)PKMN::result(b));
    printf(",\"text\":\"");
    for (const char c : PKMN::battle_data_to_string(b, d)) {
      if (c == '\n') printf("\\n"); else if (c == '"' || c == '\\') printf("\\%c", c); else putchar(c);
    }
    printf("\"");
    printf(",\"pe_score\":%.9g,\"pe_value_at_root\":%.9g", PokeEngine::evaluate_battle(battle), e.evaluate(b));
  }
  printf("}");
}

void dump_team(const uint8_t *rec, bool first) {
  PKMN::Team t[2];
  for (int s = 0; s < 2; ++s)
    for (int i = 0; i < 6; ++i) {
      const uint8_t *p = rec + (s * 6 + i) * 5;
      PKMN::Set set{};
      set.species = static_cast<PKMN::Data::Species>(p[0]);
      for (int m = 0; m < 4; ++m) set.moves[m] = static_cast<PKMN::Data::Move>(p[1 + m]);
      t[s][i] = set;
    }
  uint64_t seed; memcpy(&seed, rec + 60, 8);
  const pkmn_gen1_battle b = PKMN::battle(t[0], t[1], seed);
  printf("%s", first ? "" : ",\n"); hex(reinterpret_cast<const uint8_t *>(&b), 384);
}

// What Update's constructor reads of an MCTS::Output (mcts.h:68-90: k, empirical, nash per side; iterations; the two values).
struct OutputSide { uint8_t k; std::array<double, 9> empirical; std::array<double, 9> nash; };
struct OutputLike { size_t iterations; double empirical_value; double nash_value; OutputSide p1; OutputSide p2; };

size_t dump_game(const uint8_t *p, size_t left, bool first) {
  if (left < 387) { fprintf(stderr, "short game header\n"); exit(2); }
  pkmn_gen1_battle b; memcpy(&b, p, 384);
  Train::Battle::CompressedFrames frames{b};
  frames.result = p[384];
  uint16_t count; memcpy(&count, p + 385, 2);
  size_t at = 387;
  for (unsigned u = 0; u < count; ++u) {
    if (left < at + 8 + 38 * 8) { fprintf(stderr, "short update\n"); exit(2); }
    OutputLike o{};
    o.p1.k = p[at]; o.p2.k = p[at + 1];
    const pkmn_choice c1 = p[at + 2], c2 = p[at + 3];
    uint32_t it; memcpy(&it, p + at + 4, 4); o.iterations = it;
    double v[38]; memcpy(v, p + at + 8, sizeof v);
    o.empirical_value = v[0]; o.nash_value = v[1];
    for (int i = 0; i < 9; ++i) { o.p1.empirical[i] = v[2 + i]; o.p1.nash[i] = v[11 + i]; o.p2.empirical[i] = v[20 + i]; o.p2.nash[i] = v[29 + i]; }
    frames.updates.emplace_back(o, c1, c2);
    at += 8 + 38 * 8;
  }
  std::vector<char> buf(frames.n_bytes());
  frames.write(buf.data());
  printf("%s", first ? "" : ",\n"); hex(reinterpret_cast<const uint8_t *>(buf.data()), buf.size());
  return at;
}
} // namespace

int main(int argc, char **argv) {
  if (argc != 3) { fprintf(stderr, "usage: %s states|teams <file>\n", argv[0]); return 2; }
  const auto in = slurp(argv[2]);
  if (!strcmp(argv[1], "frames")) {
    printf("[");
    for (size_t at = 0, i = 0; at < in.size(); ++i) at += dump_game(in.data() + at, in.size() - at, i == 0);
    printf("]\n");
    return 0;
  }
  if (!strcmp(argv[1], "names")) {
    std::string text(in.begin(), in.end());
    printf("[");
    size_t at = 0; bool first = true;
    while (at < text.size()) {
      size_t nl = text.find('\n', at);
      if (nl == std::string::npos) nl = text.size();
      const std::string tok = text.substr(at, nl - at);
      at = nl + 1;
      if (tok.empty()) continue;
      int sp = -1, mv = -1;
      try { sp = (int)PKMN::string_to_species(tok); } catch (const std::exception &) {}
      try { mv = (int)PKMN::string_to_move(tok); } catch (const std::exception &) {}
      printf("%s[%d,%d]", first ? "" : ",", sp, mv); first = false;
    }
    printf("]\n");
    return 0;
  }
  const bool states = !strcmp(argv[1], "states");
  const size_t rec = states ? 400 : 68;
  if (in.size() % rec) { fprintf(stderr, "input is not a whole number of %zu-byte records\n", rec); return 2; }
  printf("[");
  for (size_t i = 0; i < in.size() / rec; ++i) states ? dump_state(in.data() + i * rec, i == 0) : dump_team(in.data() + i * rec, i == 0);
  printf("]\n");
  return 0;
}
