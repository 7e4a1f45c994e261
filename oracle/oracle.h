/* oracle/oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C) of the reference's hot path, used only as the checker by
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  Nothing under
 * oak_amd/ may include, link or call this.
 *
 * PARITY STATUS: "parity unpinned" at the libpkmn boundary.  Turn resolution in the
 * reference lives in the un-vendored third-party Zig library lab-oak/engine (fork of
 * pkmn/engine, version unpinned: /root/reference/.gitmodules:4-6, built by
 * /root/reference/dev/libpkmn:9 with -Dshowdown -Doption=ebc=false -Doption=miss=false
 * -Doption=advance=false -Doption=key=true -Dchance -Dcalc).  Its source is absent and the
 * reference's tests hold no post-update state bytes, so gen1_engine.c restates the
 * published gen-1 / Pokemon-Showdown mechanics on the reference's own state layout
 * (cpp/include/libpkmn/layout.h, data.h) and is pinned only by:
 *   - the 13 known-answer positions of cpp/src/search-test.cc:50-109 (tests/test_known_answers.py)
 *   - the turn-0 battle bytes / RNG streams of SURVEY.md Appendix B (tests/golden/)
 *   - the hidden-vs-public counter semantics of cpp/include/search/durations.h:25-97
 * Everything Oak-side (init, hidden-variable resampling, device RNGs, LCG, rollout loop)
 * is restated from headers that ARE present and is pinned by reference-generated goldens.
 */
#ifndef ORACLE_H
#define ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORACLE_BATTLE_SIZE 384
#define ORACLE_MAX_CHOICES 9

/* result byte: type | p1_request<<4 | p2_request<<6   (cpp/include/libpkmn/pkmn.h:214-233) */
enum { ORACLE_NONE = 0, ORACLE_WIN = 1, ORACLE_LOSE = 2, ORACLE_TIE = 3, ORACLE_ERROR = 4 };
/* choice byte: kind | data<<2, kind 0 pass / 1 move / 2 switch (pkmn.h:108-133) */
enum { ORACLE_PASS = 0, ORACLE_MOVE = 1, ORACLE_SWITCH = 2 };

typedef struct {
  uint8_t actions[16];   /* chance actions: 2 x 8 B, bit offsets layout.h:98-117 */
  uint8_t durations[8];  /* chance durations: 2 x u32, layout.h:119-125 */
  uint8_t overrides[16]; /* calc overrides: damage roll byte 0 (P1) / byte 8 (P2), mcts.h:575-588 */
} oracle_options;

/* pkmn_gen1_battle_options_set (pkmn.h:88-104): NULL durations keeps the tracked
 * durations and resets actions; NULL overrides clears the damage overrides. */
void oracle_options_set(oracle_options *o, const uint8_t *durations8, const uint8_t *overrides16);

/* pkmn_gen1_battle_update / pkmn_gen1_battle_choices restatement. */
uint8_t oracle_update(uint8_t *battle384, uint8_t c1, uint8_t c2, oracle_options *o);
uint8_t oracle_choices(const uint8_t *battle384, int player, int request, uint8_t *out, size_t len);

/* PKMN::battle / Init::init_side (pkmn.h:50-57, init.h:90-154).  teams: 2 x 6 x
 * {species, move0..move3} bytes (5 B per set, species 0 = empty slot), level 100. */
void oracle_init_battle(uint8_t *battle384, const uint8_t *teams60, uint64_t seed);

/* PKMN::result(battle): request/result byte reconstructed from state (pkmn.h:235-272). */
uint8_t oracle_result_from_state(const uint8_t *battle384);

/* MCTS::randomize_hidden_variables (search/durations.h:25-97). */
void oracle_randomize_hidden_variables(uint8_t *battle384, const uint8_t *durations8);

/* Device RNGs (util/random.h).  mt19937: libstdc++ uniform_int_distribution<uint64_t>
 * over std::mt19937 = (draw0 << 32) + draw1. */
typedef struct { uint32_t mt[624]; int idx; } oracle_mt19937;
void oracle_mt19937_seed(oracle_mt19937 *g, uint32_t seed);
uint32_t oracle_mt19937_next32(oracle_mt19937 *g);
uint64_t oracle_mt19937_uniform_64(oracle_mt19937 *g);
void oracle_fast_prng_seed(uint8_t state8[8], uint64_t seed); /* std::seed_seq{lo,hi} */
uint32_t oracle_fast_prng_next32(uint8_t state8[8]);
uint64_t oracle_fast_prng_uniform_64(uint8_t state8[8]);
void oracle_fast_prng_seed_batch(uint8_t *states8, uint32_t n, uint64_t seed0);               /* lane i seeded with seed0 + i */
void oracle_fast_prng_spawn_batch(uint8_t *lane_states8, uint32_t n, uint8_t *playout_states8); /* oakgpu_root_steps' stream rule */

/* MCTS::Search::init_stats_and_rollout (search/mcts.h:448-496) with a fast_prng device
 * whose 8-byte state is `prng8`; stops at a terminal result or after max_steps
 * turn-steps (returns ORACLE_NONE-typed result then).  *steps gets #updates done. */
uint8_t oracle_rollout_fast(uint8_t *battle384, uint8_t *durations8, uint8_t result,
                            uint8_t prng8[8], uint32_t max_steps, uint32_t *steps);
/* Same loop driven by a shared mt19937 device (benchmark.cc:23-31 style). */
uint8_t oracle_rollout_mt(uint8_t *battle384, uint8_t *durations8, uint8_t result,
                          oracle_mt19937 *dev, uint32_t max_steps, uint32_t *steps);

/* Multi-threaded batch helper used only by bench.py's cpu_baseline leg and tests:
 * for lane i: run_root_iteration prep (battle.rng = prng.uniform_64(),
 * randomize_hidden_variables) when `prep` != 0, then oracle_rollout_fast.  Buffers are
 * AoS: battles n x 384, durations n x 8, prng n x 8, results n, steps n. */
void oracle_rollout_batch(uint8_t *battles, uint8_t *durations, const uint8_t *results_in,
                          uint8_t *prng, uint32_t n, uint32_t max_steps, int prep,
                          uint8_t *results_out, uint32_t *steps_out, int threads);

/* SURVEY 8(d) config-2 synthetic input: random OU team pair for lane seed `seed`
 * (fast_prng stream), written as turn-0 battle + first update(0,0).  Leaves the
 * continuing fast_prng state in prng8.  pools: see oracle_set_ou_pools. */
void oracle_set_ou_pools(const uint8_t *legal_species, int n_species,
                         const uint8_t *pool_moves /*152 x 48*/, const uint8_t *pool_sizes /*152*/);
uint8_t oracle_make_random_ou_battle(uint8_t *battle384, uint8_t *durations8, uint8_t prng8[8],
                                     uint64_t seed);

uint64_t oracle_hash64(const uint8_t *p, size_t n); /* FNV-1a, for fixtures */

#ifdef __cplusplus
}
#endif
#endif
