/* oracle/oak_host.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.  See oracle/oracle.h.
 *
 * CPU restatement of the Oak-side integer code around the libpkmn calls: turn-0 battle
 * construction, hidden-variable resampling, the two device RNGs and the random-playout
 * loop.  Unlike gen1_engine.c these are restated from reference code that IS present and
 * are pinned by reference-generated known answers (tests/golden/rng_known_answers.json,
 * SURVEY.md Appendix B).
 */
#include "oracle.h"
#include "gen1_tables.h"
#include <pthread.h>
#include <string.h>

/* ---- PKMN::battle / Init::init_side  (cpp/include/libpkmn/init.h:35-40,90-154) --------- */
static uint16_t compute_stat(uint32_t base, int hp, uint32_t level) {
  uint32_t core = 2 * (base + 15) + 63; /* DVs 15, stat exp 255/4 */
  uint32_t factor = hp ? level + 10 : 5;
  return (uint16_t)(core * level / 100 + factor);
}
static void put16(uint8_t *p, uint16_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); }

void oracle_init_battle(uint8_t *battle, const uint8_t *teams, uint64_t seed) {
  memset(battle, 0, ORACLE_BATTLE_SIZE);
  for (int s = 0; s < 2; ++s) {
    uint8_t *side = battle + 184 * s;
    for (int i = 0; i < 6; ++i) {
      const uint8_t *set = teams + (s * 6 + i) * 5;
      uint8_t *pk = side + 24 * i;
      uint8_t sp = set[0];
      pk[21] = sp;
      if (sp == 0) continue; /* init.h:94-96: empty slot stays zero */
      const oracle_species_t *sd = &ORACLE_SPECIES[sp];
      const uint32_t level = 100;
      pk[23] = (uint8_t)level;
      put16(pk + 0, compute_stat(sd->hp, 1, level));
      put16(pk + 2, compute_stat(sd->atk, 0, level));
      put16(pk + 4, compute_stat(sd->def, 0, level));
      put16(pk + 6, compute_stat(sd->spe, 0, level));
      put16(pk + 8, compute_stat(sd->spc, 0, level));
      for (int m = 0; m < 4; ++m) {
        pk[10 + 2 * m] = set[1 + m];
        /* max_pp = min(PP/5*8, 61) (moves.h:1794-1796); move 0 has no PP row: the
         * reference indexes PP[-1]; teams here always carry 4 real moves or pad with 0 */
        pk[11 + 2 * m] = set[1 + m] ? ORACLE_MOVES[set[1 + m]].pp : 0;
      }
      put16(pk + 18, compute_stat(sd->hp, 1, level));
      pk[20] = 0;
      pk[22] = (uint8_t)(sd->type1 | (sd->type2 << 4));
      if (i == 0 || (pk[18] | pk[19])) side[176 + i] = (uint8_t)(i + 1); /* init.h:148-150 */
    }
  }
  for (int k = 0; k < 8; ++k) battle[376 + k] = (uint8_t)(seed >> (8 * k));
}

/* ---- MCTS::randomize_hidden_variables  (cpp/include/search/durations.h:25-97) ---------- */
static const uint8_t MULTI[4][40] = {
    {1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 3, 3, 3, 3, 3, 3, 4, 4, 4, 4},
    {1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 3, 3, 3, 3, 3, 3, 3, 3},
    {1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2},
    {1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1}};

void oracle_randomize_hidden_variables(uint8_t *battle, const uint8_t *durations8) {
  uint64_t rng;
  memcpy(&rng, battle + 376, 8);
  for (int s = 0; s < 2; ++s) {
    uint8_t *side = battle + 184 * s;
    uint32_t d;
    memcpy(&d, durations8 + 4 * s, 4);
    uint64_t vol;
    memcpy(&vol, side + 144 + 16, 8);
    uint32_t confusion = (d >> 18) & 7, disable = (d >> 21) & 15, attacking = (d >> 25) & 7, binding = (d >> 28) & 7;
    if (confusion) {
      uint8_t max = (uint8_t)(6 - (confusion + (confusion == 1)));
      uint64_t left = (uint8_t)((rng % max) + 1 + (confusion == 1));
      vol = (vol & ~(7ull << 18)) | ((left & 7) << 18);
    }
    if (disable) {
      uint8_t max = (uint8_t)(9 - disable);
      uint64_t left = (uint8_t)((rng % max) + 1);
      vol = (vol & ~(15ull << 52)) | ((left & 15) << 52);
    }
    if (attacking) {
      if (vol & 3) { /* bide or thrashing: same logic (durations.h:61-76) */
        uint64_t a = attacking == 3 ? 1 : (uint64_t)(4 - (attacking + (rng % 2)));
        vol = (vol & ~(7ull << 21)) | ((a & 7) << 21);
      }
    }
    if (binding) {
      uint64_t a = MULTI[binding - 1][rng % 40];
      vol = (vol & ~(7ull << 21)) | ((a & 7) << 21);
    }
    memcpy(side + 144 + 16, &vol, 8);
    for (int i = 0; i < 6; ++i) {
      uint32_t sleep = (d >> (3 * i)) & 7;
      if (!sleep) continue;
      uint8_t id = side[176 + i];
      uint8_t *status = side + 24 * (id - 1) + 20;
      if ((*status & 7) && !(*status & 0x80)) {
        uint8_t max = (uint8_t)(8 - sleep);
        *status = (uint8_t)((*status & 0xF8) | (uint8_t)((rng % max) + 1));
      }
    }
  }
}

/* ---- device RNGs  (cpp/include/util/random.h) ------------------------------------------ */
void oracle_mt19937_seed(oracle_mt19937 *g, uint32_t seed) {
  g->mt[0] = seed;
  for (int i = 1; i < 624; ++i) g->mt[i] = 1812433253u * (g->mt[i - 1] ^ (g->mt[i - 1] >> 30)) + (uint32_t)i;
  g->idx = 624;
}
uint32_t oracle_mt19937_next32(oracle_mt19937 *g) {
  if (g->idx >= 624) {
    for (int i = 0; i < 624; ++i) {
      uint32_t y = (g->mt[i] & 0x80000000u) | (g->mt[(i + 1) % 624] & 0x7fffffffu);
      g->mt[i] = g->mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1) ? 0x9908b0dfu : 0);
    }
    g->idx = 0;
  }
  uint32_t y = g->mt[g->idx++];
  y ^= y >> 11;
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= y >> 18;
  return y;
}
/* libstdc++ uniform_int_distribution<uint64_t> over a 32-bit URNG: (hi << 32) + lo */
uint64_t oracle_mt19937_uniform_64(oracle_mt19937 *g) {
  uint64_t hi = oracle_mt19937_next32(g);
  uint64_t lo = oracle_mt19937_next32(g);
  return (hi << 32) + lo;
}

/* std::seed_seq{lo32, hi32}.generate(2 words)  (random.h:99-105) */
void oracle_fast_prng_seed(uint8_t state8[8], uint64_t seed) {
  const uint32_t v[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
  uint32_t b[2] = {0x8b8b8b8bu, 0x8b8b8b8bu};
  const uint32_t n = 2, s = 2, t = 0, p = (n - t) / 2, q = p + t, m = 3;
  for (uint32_t k = 0; k < m; ++k) {
    uint32_t arg = b[k % n] ^ b[(k + p) % n] ^ b[(k + n - 1) % n];
    uint32_t r1 = 1664525u * (arg ^ (arg >> 27));
    uint32_t r2 = r1 + (k == 0 ? s : (k <= s ? (k % n) + v[k - 1] : k % n));
    b[(k + p) % n] += r1;
    b[(k + q) % n] += r2;
    b[k % n] = r2;
  }
  for (uint32_t k = m; k < m + n; ++k) {
    uint32_t arg = b[k % n] + b[(k + p) % n] + b[(k + n - 1) % n];
    uint32_t r3 = 1566083941u * (arg ^ (arg >> 27));
    uint32_t r4 = r3 - (k % n);
    b[(k + p) % n] ^= r3;
    b[(k + q) % n] ^= r4;
    b[k % n] = r4;
  }
  memcpy(state8, b, 8);
}
static inline uint32_t rotl32(uint32_t x, int k) { return (x << k) | (x >> (32 - k)); }
uint32_t oracle_fast_prng_next32(uint8_t state8[8]) {
  uint32_t s0, s1;
  memcpy(&s0, state8, 4);
  memcpy(&s1, state8 + 4, 4);
  uint32_t result = rotl32(s0 + s1, 9) + s0;
  s1 ^= s0;
  s0 = rotl32(s0, 13) ^ s1 ^ (s1 << 5);
  s1 = rotl32(s1, 28);
  memcpy(state8, &s0, 4);
  memcpy(state8 + 4, &s1, 4);
  return result;
}
/* batch helpers of the tests (plain loops over the two functions above) */
void oracle_fast_prng_seed_batch(uint8_t *states8, uint32_t n, uint64_t seed0) {
  for (uint32_t i = 0; i < n; ++i) oracle_fast_prng_seed(states8 + (size_t)i * 8, seed0 + i);
}
/* oakgpu_root_steps' stream rule: lane i's stream advances by ONE uniform_64 (hi, lo); that draw is the 8-byte state {s0 = hi, s1 = lo} of
 * the step's fresh playout's own stream (all-zero -- the generator's fixed point -- becomes s1 = 1). */
void oracle_fast_prng_spawn_batch(uint8_t *lane_states8, uint32_t n, uint8_t *playout_states8) {
  for (uint32_t i = 0; i < n; ++i) {
    uint32_t hi = oracle_fast_prng_next32(lane_states8 + (size_t)i * 8), lo = oracle_fast_prng_next32(lane_states8 + (size_t)i * 8);
    if (hi == 0 && lo == 0) lo = 1;
    memcpy(playout_states8 + (size_t)i * 8, &hi, 4);
    memcpy(playout_states8 + (size_t)i * 8 + 4, &lo, 4);
  }
}
uint64_t oracle_fast_prng_uniform_64(uint8_t state8[8]) {
  uint64_t hi = oracle_fast_prng_next32(state8);
  uint64_t lo = oracle_fast_prng_next32(state8);
  return (hi << 32) | lo;
}

/* ---- MCTS::Search::init_stats_and_rollout  (cpp/include/search/mcts.h:448-496) ---------- */
static inline uint8_t rollout_step(uint8_t *battle, oracle_options *opt, uint8_t result, uint64_t seed) {
  uint8_t ch[ORACLE_MAX_CHOICES];
  uint8_t m = oracle_choices(battle, 0, (result >> 4) & 3, ch, ORACLE_MAX_CHOICES);
  uint8_t c1 = ch[seed % m];
  uint8_t n = oracle_choices(battle, 1, (result >> 6) & 3, ch, ORACLE_MAX_CHOICES);
  seed >>= 32;
  uint8_t c2 = ch[seed % n];
  oracle_options_set(opt, 0, 0);
  return oracle_update(battle, c1, c2, opt);
}

uint8_t oracle_rollout_fast(uint8_t *battle, uint8_t *durations8, uint8_t result, uint8_t prng8[8],
                            uint32_t max_steps, uint32_t *steps) {
  oracle_options opt;
  memset(&opt, 0, sizeof opt);
  memcpy(opt.durations, durations8, 8);
  uint32_t k = 0;
  while (!(result & 15) && k < max_steps) {
    result = rollout_step(battle, &opt, result, oracle_fast_prng_uniform_64(prng8));
    ++k;
  }
  memcpy(durations8, opt.durations, 8);
  if (steps) *steps = k;
  return result;
}

uint8_t oracle_rollout_mt(uint8_t *battle, uint8_t *durations8, uint8_t result, oracle_mt19937 *dev,
                          uint32_t max_steps, uint32_t *steps) {
  oracle_options opt;
  memset(&opt, 0, sizeof opt);
  memcpy(opt.durations, durations8, 8);
  uint32_t k = 0;
  while (!(result & 15) && k < max_steps) {
    result = rollout_step(battle, &opt, result, oracle_mt19937_uniform_64(dev));
    ++k;
  }
  memcpy(durations8, opt.durations, 8);
  if (steps) *steps = k;
  return result;
}

typedef struct {
  uint8_t *battles, *durations, *prng, *results_out;
  const uint8_t *results_in;
  uint32_t *steps_out;
  uint32_t lo, hi, max_steps;
  int prep;
} batch_job;

static void *batch_worker(void *arg) {
  batch_job *j = (batch_job *)arg;
  for (uint32_t i = j->lo; i < j->hi; ++i) {
    uint8_t *b = j->battles + (size_t)i * 384, *d = j->durations + (size_t)i * 8, *p = j->prng + (size_t)i * 8;
    if (j->prep) { /* run_root_iteration prep, mcts.h:250-263 */
      uint64_t r = oracle_fast_prng_uniform_64(p);
      memcpy(b + 376, &r, 8);
      oracle_randomize_hidden_variables(b, d);
    }
    uint32_t steps = 0;
    j->results_out[i] = oracle_rollout_fast(b, d, j->results_in[i], p, j->max_steps, &steps);
    j->steps_out[i] = steps;
  }
  return 0;
}

void oracle_rollout_batch(uint8_t *battles, uint8_t *durations, const uint8_t *results_in, uint8_t *prng,
                          uint32_t n, uint32_t max_steps, int prep, uint8_t *results_out,
                          uint32_t *steps_out, int threads) {
  if (threads < 1) threads = 1;
  if (threads > 256) threads = 256;
  pthread_t tid[256];
  batch_job jobs[256];
  for (int t = 0; t < threads; ++t) {
    batch_job j = {battles, durations, prng, results_out, results_in, steps_out,
                   (uint32_t)((uint64_t)n * t / threads), (uint32_t)((uint64_t)n * (t + 1) / threads), max_steps, prep};
    jobs[t] = j;
    if (threads == 1) { batch_worker(&jobs[0]); return; }
    pthread_create(&tid[t], 0, batch_worker, &jobs[t]);
  }
  for (int t = 0; t < threads; ++t) pthread_join(tid[t], 0);
}

/* ---- random OU team pairs --------------------------------------- */
static uint8_t g_legal[152];
static int g_nlegal;
static uint8_t g_pool[152][48];
static uint8_t g_pool_n[152];

void oracle_set_ou_pools(const uint8_t *legal_species, int n_species, const uint8_t *pool_moves,
                         const uint8_t *pool_sizes) {
  g_nlegal = n_species;
  memcpy(g_legal, legal_species, (size_t)n_species);
  memcpy(g_pool, pool_moves, sizeof g_pool);
  memcpy(g_pool_n, pool_sizes, 152);
}

/* lane generator: fast_prng::seed(state, seed); per side 6 distinct legal species drawn as
 * legal[next32 % n] with rejection; per species min(4, pool) distinct moves drawn as
 * pool[next32 % size] with rejection; battle.rng = uniform_64(); then update(0, 0). */
uint8_t oracle_make_random_ou_battle(uint8_t *battle, uint8_t *durations8, uint8_t prng8[8], uint64_t seed) {
  uint8_t teams[60];
  memset(teams, 0, sizeof teams);
  oracle_fast_prng_seed(prng8, seed);
  for (int s = 0; s < 2; ++s) {
    for (int k = 0; k < 6; ++k) {
      uint8_t sp;
      for (;;) {
        sp = g_legal[oracle_fast_prng_next32(prng8) % (uint32_t)g_nlegal];
        int dup = 0;
        for (int j = 0; j < k; ++j) dup |= teams[(s * 6 + j) * 5] == sp;
        if (!dup) break;
      }
      uint8_t *set = teams + (s * 6 + k) * 5;
      set[0] = sp;
      int want = g_pool_n[sp] < 4 ? g_pool_n[sp] : 4;
      for (int m = 0; m < want; ++m) {
        for (;;) {
          uint8_t mv = g_pool[sp][oracle_fast_prng_next32(prng8) % g_pool_n[sp]];
          int dup = 0;
          for (int j = 0; j < m; ++j) dup |= set[1 + j] == mv;
          if (!dup) { set[1 + m] = mv; break; }
        }
      }
    }
  }
  oracle_init_battle(battle, teams, oracle_fast_prng_uniform_64(prng8));
  oracle_options opt;
  memset(&opt, 0, sizeof opt);
  uint8_t r = oracle_update(battle, 0, 0, &opt);
  memcpy(durations8, opt.durations, 8);
  return r;
}
