/* include/oakgpu.h -- C ABI of liboakgpu.so (MI355X / gfx950).
 *
 * Drop-in boundary for ONE hot path of lab-oak/oak: batched random-playout turn stepping
 * and MLP leaf evaluation.  Every entry point is plain C (pointers + sizes, no torch / C++
 * types) so the reference's own FFI layers (C++ headers, pybind11) can bind it directly.
 * Each function cites the reference interface it replaces; INTEGRATION.md shows the
 * reference-side glue.
 *
 * Conventions
 *   - `*_dev` functions take DEVICE pointers and enqueue on the context's HIP stream
 *     without synchronising; the un-suffixed forms take HOST pointers, copy, run, copy
 *     back and synchronise (PCIe-inclusive; never the benchmarked path).
 *   - Battles are the reference's 384-byte POD (cpp/include/libpkmn/layout.h:5-31), AoS,
 *     n x 384 bytes; durations n x 8 bytes (data.h:270-311); choices / results one byte.
 *   - Return value 0 = success; otherwise a hipError_t-compatible code (or -1 for
 *     argument errors) and oakgpu_last_error() describes it.  There is NO CPU fallback:
 *     without a usable HIP device every call fails.
 */
#ifndef OAKGPU_H
#define OAKGPU_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OAKGPU_BATTLE_SIZE 384
#define OAKGPU_DURATIONS_SIZE 8
#define OAKGPU_ACTIONS_SIZE 16
#define OAKGPU_MAX_CHOICES 9
#define OAKGPU_TEAMS_SIZE 60 /* 2 sides x 6 sets x {species, move x4} */

typedef struct oakgpu_ctx oakgpu_ctx;
typedef struct oakgpu_net oakgpu_net; /* a loaded .battle.net (see oakgpu_net_load below) */

int oakgpu_create(oakgpu_ctx **out, int device);
void oakgpu_destroy(oakgpu_ctx *ctx);
const char *oakgpu_last_error(void);
/* Use an existing hipStream_t (e.g. torch's current stream) for all *_dev launches. */
int oakgpu_set_stream(oakgpu_ctx *ctx, void *hip_stream);
/* The hipStream_t all *_dev launches of this context go to (own stream unless oakgpu_set_stream was called). */
void *oakgpu_get_stream(oakgpu_ctx *ctx);
int oakgpu_synchronize(oakgpu_ctx *ctx);
/* Diagnostics: with timing on, every oakgpu_leaf_eval*_dev call records HIP events around its three kernels on the
 * context's stream; oakgpu_get_leaf_kernel_ms waits for the last call and returns {party-slot embedding pass, actives'
 * embedding pass, main net} in milliseconds.  Off by default (the events cost a few microseconds per call). */
int oakgpu_set_kernel_timing(oakgpu_ctx *ctx, int on);
int oakgpu_get_leaf_kernel_ms(oakgpu_ctx *ctx, float ms[3]);
/* Rollout scheduling (results never depend on it).  k = 1 launches one lane per playout; k > 1 (default 2)
 * launches n/k persistent lanes that refill from an atomic playout queue as playouts finish. */
int oakgpu_set_playouts_per_lane(oakgpu_ctx *ctx, int k);
/* Regrouping rounds of the queue schedule (off unless this is called): the rollout runs as `rounds` dispatches (1..8); in
 * every round but the last a wave whose queue is dry and that has fewer than `suspend_below` (0..64) playouts still running
 * parks them (bit-exact state image) for the next round, which packs them 64 to a wave again on 1/`shrink` (>= 1) of the
 * waves.  It paid off while many small launches ran side by side on many streams (round 1); a lone launch is faster as one
 * dispatch (measured at every size, round 3), which is the default.  rounds = 1 or suspend_below = 0 disables it again. */
int oakgpu_set_regroup(oakgpu_ctx *ctx, int rounds, int suspend_below, int shrink);
/* The tail of a launch that saturates the device (a group of batches): when the playout queue is dry, a wave with fewer
 * than `below` (1..64; 0 = off) playouts still running parks them, and ONE follow-up dispatch of `waves` waves (0 = one
 * per compute unit) finishes the parked playouts, `lanes` of them per wave at a time (0 = 64).  The survivors are the rare
 * playouts that run into the step cap; a wave that holds one of them alone advances it no faster than a wave that holds
 * a few, and hundreds of such waves slow each other down.  Results never depend on it. */
int oakgpu_set_tail_pack(oakgpu_ctx *ctx, int below, int waves, int lanes);
/* Queue order of a launch of >= 8,192 playouts (default on): playouts with a Ghost-type Pokemon or a Ditto on either team are
 * handed out first.  They hold practically all of the playouts that run into the step cap (Normal-type Rage / Struggle locks
 * against a Ghost), and the launch ends with its longest playout: started early, the 1,000-step chains are mostly done when
 * the queue runs dry.  Pure scheduling: results are indexed by playout and never depend on it. */
int oakgpu_set_queue_order(oakgpu_ctx *ctx, int on);
/* Launches that do not fill the device use its empty wave slots: every wave takes only `lanes` playouts at a time, on as many
 * more waves as that needs (never more than the device holds).  -1 (default) = automatic: as few lanes per wave as fill the
 * device, at least 4; 0 or 64 = off.  Results never depend on it. */
int oakgpu_set_spread(oakgpu_ctx *ctx, int lanes);
/* Long-playout migration inside a launch (mode 0 off, 1 = launches that saturate the device (default), 2 = every queue launch:
 * tests): a wave hands a playout that is still running after `long_steps` turn-steps (default 300; 99.5% end before 250) to
 * `adopters` dedicated waves (0 = one per two compute units), which from the first donation on hold only such playouts -- a dozen
 * per wave at the top priority of their SIMD -- so the 1,000-step chains that end a launch advance at a sparse wave's pace
 * long before the device drains.  State travels as the regrouping rounds' bit-exact image; results never depend on it.
 * Co-residency of the launch's waves is NOT required (other launches may share the device, e.g. a second context's group launch):
 * a bulk wave never waits for anybody, so it always runs to its end and counts itself out; an adopter waits only for bulk waves
 * (bounded: ~10 s of polls, then the sticky error word and a failed oakgpu_synchronize) and holds nothing a bulk wave needs --
 * tests/test_gpu_parity.py::test_two_contexts_migrate_concurrently_on_one_device. */
int oakgpu_set_migration(oakgpu_ctx *ctx, int mode, int long_steps, int adopters);
/* ... and, earlier than that, a playout whose two active Pokemon have stood still -- same slots, same hp -- for `window`
 * consecutive turn-steps (default 48; 0 = off; at most 255): the playouts that run into the 1,000-step cap are stalemates whose
 * hp stop changing at a median of turn-step 85, so they leave their full wave at ~step 135 instead of 300.  A heuristic about
 * WHO finishes a playout only; results never depend on it. */
int oakgpu_set_migration_window(oakgpu_ctx *ctx, int window);
/* The queue kernel takes a PROVEN frozen standstill -- both actives frozen (gen 1 never thaws by itself), neither side able to leave or
 * act, different speeds: a turn-step then draws nothing and changes nothing but the turn counter -- to its last turn-step in one go
 * (exact; tests/test_gpu_parity.py::test_frozen_standstill_skip_is_exact).  on = 0 plays every turn-step instead (A / B; also
 * OAKGPU_STANDSTILL_SKIP=0 at context creation).  Default 1.  The one-lane-per-playout kernels (k_rollout_regs, k_root_step) never skip. */
int oakgpu_set_standstill_skip(oakgpu_ctx *ctx, int on);
/* Diagnostic (synchronises the stream): the 64 control words of the last queue launch -- [32] / [33]
 * the queue order's two counters, [40] donations, [41] adoptions, [42] bulk waves that left, [63] error bits (0 = none:
 * 1 a ticket never arrived, 2 an adopter gave up waiting) -- STICKY: no launch clears them; oakgpu_synchronize reports a
 * non-zero word as a failed call and clears it, so the error of any launch since the last synchronize is seen, not only the
 * last launch's. */
int oakgpu_get_queue_counters(oakgpu_ctx *ctx, uint32_t *out64);
/* Rollout engine (results never depend on it; both are bit-identical): 2 = register-resident engine, one wave per
 * workgroup, queue refill (default); 1 = LDS-resident engine (first implementation, kept as a second opinion on the
 * transcription: it is what oakgpu_update* / _choices* / the tree step run, and tests/test_gpu_parity.py holds its
 * rollouts to the oracle too), one launch per batch.  (Round 2-4's engine 3, 256-lane workgroups that re-binned
 * their playouts by action class through LDS before each action slot, lost to engine 2 by 8-12 % and was removed in round 5:
 * DESIGN_HISTORY.md.) */
int oakgpu_set_rollout_engine(oakgpu_ctx *ctx, int engine);
int oakgpu_device_count(void);

/* ---- rollout: replaces MCTS::Search::init_stats_and_rollout (search/mcts.h:448-496) and,
 * with prep != 0, the per-iteration prep of run_root_iteration (mcts.h:250-263:
 * battle.rng = device.uniform_64(); randomize_hidden_variables (search/durations.h:25-97)).
 * Device RNG = the reference's fast_prng (util/random.h:67-133), 8 bytes of state per lane,
 * advanced in place.  Per lane: play uniformly random legal joint choices
 * (c1 = p1_choices[seed % m]; c2 = p2_choices[(seed >> 32) % n]) until the result type is
 * non-zero or max_steps turn-steps were made.
 *   results_out[i] : final pkmn_result byte (type 0 = hit the step cap)
 *   steps_out[i]   : number of pkmn_gen1_battle_update-equivalent turn-steps executed
 *   values_out[i]  : 1 / 0 / 0.5 for win / lose / (tie or capped)  (mcts.h:481-495)
 *   battles_out / durations_out (nullable): final state bytes (parity checks, stepping a resident batch).
 * Every output may alias its input (battles_out == battles, durations_out == durations, results_out ==
 * results_in): lane i only ever reads and writes row i. */
int oakgpu_rollout_dev(oakgpu_ctx *ctx, const uint8_t *battles, const uint8_t *durations,
                       const uint8_t *results_in, uint8_t *prng_state, uint32_t n, uint32_t max_steps,
                       int prep, uint8_t *results_out, uint32_t *steps_out, float *values_out,
                       uint8_t *battles_out, uint8_t *durations_out);
int oakgpu_rollout(oakgpu_ctx *ctx, const uint8_t *battles, const uint8_t *durations,
                   const uint8_t *results_in, uint8_t *prng_state, uint32_t n, uint32_t max_steps, int prep,
                   uint8_t *results_out, uint32_t *steps_out, float *values_out, uint8_t *battles_out,
                   uint8_t *durations_out);

/* Group launch: `count` (<= 64) independent batches drained by ONE launch through one playout queue, so the group has
 * a single tail instead of one per batch (root-parallel MCTS submits its roots' batches together).  Same per-batch
 * contract as oakgpu_rollout_dev (outputs may alias inputs; results never depend on the grouping); batches with
 * n == 0 are skipped.  oakgpu_rollout_dev is the group of one. */
typedef struct {
  const uint8_t *battles, *durations, *results_in;
  uint8_t *prng_state;
  uint32_t n;
  uint8_t *results_out;
  uint32_t *steps_out;
  float *values_out;
  uint8_t *battles_out, *durations_out; /* nullable */
} oakgpu_rollout_batch;
int oakgpu_rollout_group_dev(oakgpu_ctx *ctx, const oakgpu_rollout_batch *batches, uint32_t count, uint32_t max_steps,
                             int prep);
int oakgpu_rollout_group(oakgpu_ctx *ctx, const oakgpu_rollout_batch *batches, uint32_t count, uint32_t max_steps, int prep);

/* ---- rollouts driven by a caller-supplied draw stream (a SHARED sequential device generator).
 * The reference's benchmark / search drive every playout from ONE std::mt19937 (benchmark.cc:24, search.cc:150), so
 * playout i starts in the generator's output where playout i-1 stopped.  `draws` holds that output: one u64 per
 * device.uniform_64() call (mcts.h:255-257: battle.rng of the root iteration when prep != 0; mcts.h:452: one per
 * turn-step).
 *   oakgpu_mt19937_fill           : std::mt19937{seed} -> uniform_64() values (util/random.h:10-65), host side.
 *   oakgpu_rollout_draws_dev      : lane i plays from draws[offsets[i]] (offsets == NULL: from draws[i]); strides of 0
 *                                   make every lane start from the same root; every output is nullable; used_out[i] =
 *                                   draws consumed, 0xFFFFFFFF if the lane ran off the end of the stream.
 *   oakgpu_rollout_shared_device  : (host pointers) n playouts from ONE root in the sequential order of the reference:
 *                                   offsets resolved on the device by playing a playout from every start offset first.
 *                                   Fails (-1) if the stream is too short.  offsets_out / draws_consumed nullable. */
int oakgpu_mt19937_fill(uint32_t seed, uint64_t skip, uint64_t *out, size_t count);
int oakgpu_rollout_draws_dev(oakgpu_ctx *ctx, const uint8_t *battles, uint32_t battle_stride, const uint8_t *durations,
                             uint32_t durations_stride, const uint8_t *results_in, uint32_t results_stride,
                             const uint64_t *draws, uint32_t n_draws, const uint32_t *offsets, uint32_t n,
                             uint32_t max_steps, int prep, uint8_t *results_out, uint32_t *steps_out, float *values_out,
                             uint8_t *battles_out, uint8_t *durations_out, uint32_t *used_out);
int oakgpu_rollout_shared_device(oakgpu_ctx *ctx, const uint8_t *battle, const uint8_t *durations, uint8_t result,
                                 const uint64_t *draws, uint32_t n_draws, uint32_t n, uint32_t max_steps, int prep,
                                 uint8_t *results_out, uint32_t *steps_out, float *values_out, uint8_t *battles_out,
                                 uint8_t *durations_out, uint32_t *offsets_out, uint64_t *draws_consumed);

/* ---- batched pkmn_gen1_battle_update (call sites mcts.h:278,350,463,479; wrapper
 * libpkmn/pkmn.h:106-139).  In place on battles; durations in/out (chance options);
 * actions (nullable) receives the 16-byte chance-actions key of each update (mcts.h:93-98);
 * overrides (nullable, n x 16) are the calc damage-roll overrides (mcts.h:575-588). */
int oakgpu_update_dev(oakgpu_ctx *ctx, uint8_t *battles, const uint8_t *c1, const uint8_t *c2,
                      uint8_t *durations, uint8_t *actions, const uint8_t *overrides, uint32_t n,
                      uint8_t *results);
int oakgpu_update(oakgpu_ctx *ctx, uint8_t *battles, const uint8_t *c1, const uint8_t *c2, uint8_t *durations,
                  uint8_t *actions, const uint8_t *overrides, uint32_t n, uint8_t *results);

/* ---- batched pkmn_gen1_battle_choices (mcts.h:161-166,337-342; pkmn.h:141-156).
 * requests come from the per-lane result byte (p1: bits 4-5, p2: bits 6-7).
 * out: n x 9 bytes, counts: n bytes. */
int oakgpu_choices_dev(oakgpu_ctx *ctx, const uint8_t *battles, const uint8_t *results, int player,
                       uint8_t *out, uint8_t *counts, uint32_t n);
int oakgpu_choices(oakgpu_ctx *ctx, const uint8_t *battles, const uint8_t *results, int player, uint8_t *out,
                   uint8_t *counts, uint32_t n);

/* ---- PokeEngine::Eval (search/poke-engine-evaluate.h:9-204), batched: scores[i] = evaluate_battle(battle i)
 * (Eval::get_root_score for a batch of one), values[i] = scaled_sigmoid(scores[i] - root_score) = Eval::evaluate.
 * Either output may be null. */
int oakgpu_poke_engine_eval_dev(oakgpu_ctx *ctx, const uint8_t *battles, uint32_t n, float root_score, float *values,
                                float *scores);
int oakgpu_poke_engine_eval(oakgpu_ctx *ctx, const uint8_t *battles, uint32_t n, float root_score, float *values,
                            float *scores);

/* ---- one tree level for a batch of descents (MCTS::Search::run_iteration, search/mcts.h:304-389): apply
 * the joint action of every lane (c1[i] == 0xFF: lane i is finished, leave it untouched), in place, and
 * report result, the 16-byte observation key (pkmn_gen1_battle_options_chance_actions; tree edge key,
 * mcts.h:93-98,359) and BOTH players' legal choices in the new state (n x 9 + n counts each; counts are 0
 * for terminal results).  rolls in {1, 2, 3, 20, 39}: damage-roll clamping of battle_options_set
 * (mcts.h:569-604; 39 = off), override bytes derived from the last two bytes of battle.rng.  Since round 5 the level runs on the
 * register-resident engine (k_tree_step_staged; 16-byte aligned battles, else -- or with OAKGPU_TREE_STEP=lds in the environment -- on the
 * LDS-resident engine's k_tree_step); both are held to the oracle byte for byte (tests/tree_step_check.py, test_gpu_move_coverage.py). */
int oakgpu_tree_step_dev(oakgpu_ctx *ctx, uint8_t *battles, uint8_t *durations, uint8_t *results, const uint8_t *c1,
                         const uint8_t *c2, uint32_t n, uint32_t rolls, uint8_t *actions, uint8_t *p1_choices,
                         uint8_t *p1_counts, uint8_t *p2_choices, uint8_t *p2_counts);

/* ---- tree search with batched leaves: MCTS::Search::run (search/mcts.h:154-247) with a Node heap
 * (mcts.h:95-105), joint UCB / PUCB bandits (search/bandit/ucb.h:17-66, pucb.h:17-75) and the Monte-Carlo or
 * network evaluator.  `batch` descents walk the host-side tree together, one oakgpu_tree_step_dev launch per
 * level, all battle states resident on the device; lanes of a batch repel each other with a virtual loss.
 * The root matrices come back as in MCTS::Output (mcts.h:68-90), with process_output's exact Nash solve
 * (mcts.h:620-659) in nash_value / p1_nash / p2_nash. */
typedef struct {
  uint64_t iterations;   /* root iterations (mcts.h:231-235, integer budget) */
  uint32_t batch;        /* descents in flight (GPU lanes); 1 reproduces the reference's one-at-a-time order */
  float ucb_c;           /* Bandit::Params.c (UCB, PUCB, UCB1) or gamma (Exp3, PExp3) */
  int32_t bandit;        /* 0: UCB (bandit/ucb.h), 1: PUCB (pucb.h), 2: UCB1 (ucb1.h), 3: Exp3 (exp3.h), 4: PExp3 (pexp3.h);
                          * PUCB / PExp3 take priors from the policy heads and need eval = 1 */
  int32_t eval;          /* 0: MCTS::MonteCarlo rollouts (mcts.h:448-496), 1: network value (network.h:72-123),
                          * 2: PokeEngine::Eval (poke-engine-evaluate.h:186-202; root score taken at the root) */
  uint32_t max_depth;    /* descent depth at which a node is evaluated even if already expanded (0 = 100) */
  uint32_t root_rolls;   /* SearchOptions.root_rolls / other_rolls (mcts.h:107-131): 1, 2, 3, 20 or 39 (= no clamping) */
  uint32_t other_rolls;
  uint64_t seed;         /* seeds the per-lane fast_prng streams (root resampling draws + rollouts) */
  /* MatrixUCBParams (mcts.h:107-113, 263-302, 498-566): once `mucb_delay` iterations are done the ROOT joint action is no
   * longer picked by the bandits but sampled from the Nash strategies of the optimistic / pessimistic UCB matrices
   * (cells below `mucb_minimum` visits are forced first).  The matrices are re-solved once per batch (the reference's
   * `interval` = batch here).  matrix_ucb = 0 disables it. */
  int32_t matrix_ucb;
  uint32_t mucb_delay;
  uint32_t mucb_minimum;
  float mucb_c;
  float exp3_alpha;      /* Exp3 / PExp3 uniform mixing (search.cc:268-286); negative (OAKGPU_EXP3_ALPHA_DEFAULT) = the reference's
                          * default 0.05 (its value when the agent string has no third field); 0 is honoured as 0, as the
                          * reference honours it -- so a ZERO-INITIALISED struct asks for no mixing at all: start from
                          * OAKGPU_SEARCH_PARAMS_INIT (or set this field) when Exp3 / PExp3 is used.  With alpha = 0 an arm's
                          * probability can underflow to 0; the importance-weighted update then divides by the smallest
                          * normal float instead (the reference would divide by zero) */
  uint64_t duration_us;  /* time budget (search.cc:300-306): when non-zero, `iterations` is ignored and whole batches are
                          * started until this much time has elapsed; output.iterations tells how many ran */
} oakgpu_search_params;
/* Return code of oakgpu_search_heap / oakgpu_search_agent_heap (every other failure is -1 or a HIP error code): the heap's root was
 * initialised with other action counts than the position passed in -- Heap::update was not called with the move that was played, or
 * (--keep-node self-play) the kept child was expanded under other resampled hidden variables than the game really reached. */
#define OAKGPU_E_ROOT_MISMATCH (-2)
#define OAKGPU_EXP3_ALPHA_DEFAULT (-1.0f)
/* the reference's defaults where a zero is not one: UCB c = 2 is the caller's business (agent strings carry it), root / other
 * rolls = default_search {3, 1} (mcts.h:131), Exp3 mixing = the default of an absent field */
#define OAKGPU_SEARCH_PARAMS_INIT {0, 0, 0.0f, 0, 0, 0, 3, 1, 0, 0, 0, 0, 0.0f, OAKGPU_EXP3_ALPHA_DEFAULT, 0}
typedef struct {
  uint8_t m, n;                 /* legal choices per side at the root */
  uint8_t p1_choices[9], p2_choices[9];
  uint64_t visit_matrix[81];    /* [i * 9 + j] */
  double value_matrix[81];      /* cumulative P1 values */
  uint64_t iterations;
  double empirical_value, initial_value;
  double p1_empirical[9], p2_empirical[9];
  uint64_t nodes, total_depth;
  double duration_us;
  /* MCTS::Search::process_output (mcts.h:620-659): equilibrium of the empirical root matrix (x 256 as integers, solved
   * exactly, see oakgpu_solve_matrix) */
  double nash_value, p1_nash[9], p2_nash[9];
  /* Output::Side::logit / prior (mcts.h:70-77): the policy heads' logits of the root's legal choices and their softmax,
   * filled when a contextual bandit (PUCB / PExp3) initialises a FRESH root (mcts.h:196-209); otherwise as passed in */
  double p1_logit[9], p2_logit[9], p1_prior[9], p2_prior[9];
} oakgpu_search_output;
/* LRSNash::solve_fast as the reference calls it (mcts.h:643-649, pyoak solve_matrix pyoak.cc:394-426): exact Nash
 * equilibrium of the m x n (<= 9 x 9) zero-sum game whose ROW player maximises the integer payoffs[i * n + j]
 * (|payoff| <= 2^20; the reference passes value * discretize_factor).  No floating point in the solve (integer-pivoting
 * simplex on 512-bit integers; the reference's lrsnash + GMP is absent).  p1[m], p2[n] = equilibrium strategies, *value =
 * game value / discretize_factor.  Host code: no GPU involved. */
int oakgpu_solve_matrix(const int32_t *payoffs, int m, int n, int discretize_factor, double *p1, double *p2, double *value);
/* RuntimeSearch::run (util/search.h:17-66, search.cc:150-313): the search configured by the Agent's strings, as the
 * reference's binaries and pyoak.search configure it.  budget "4096" | "100ms" | "8s"; bandit "ucb-1.0" | "ucb1-2.0" |
 * "pucb-1.5" | "exp3-<gamma>[-<alpha>]" | "pexp3-..."; eval "" / "mc" | "fp" | <.battle.net path> (loaded once per device
 * and kept, like Agent::network_ptr); matrix_ucb "" | "<delay>-<interval>-<minimum>-<c>".  Unparsable strings fail with
 * the reference's error texts (where it throws std::runtime_error); `table` and `discrete` agents are refused: those two
 * components are not built.  batch = descents in flight, 0 = chosen from the budget. */
typedef struct {
  const char *budget, *bandit, *eval, *matrix_ucb;
  int discrete, table;
} oakgpu_agent;
int oakgpu_search_agent(oakgpu_ctx *ctx, const uint8_t *battle, const uint8_t *durations, uint8_t result, const oakgpu_agent *agent,
                        uint32_t batch, uint64_t seed, oakgpu_search_output *out);
void oakgpu_agent_networks_clear(oakgpu_ctx *ctx);

/* Diagnostic (no GPU involved): ONE player's bandit of the search above replayed for `steps` rounds -- select, then
 * update with values[t] -- so its arithmetic can be compared with the reference's bandit headers.  kind as in
 * oakgpu_search_params.bandit; c = Params.c (UCB / PUCB / UCB1) or gamma (Exp3 / PExp3); logits (k floats): PUCB /
 * PExp3 priors; uniforms: the device.uniform() draw of each sampled selection (Exp3 / PExp3, k > 1).  Outputs: selected
 * index (and probability) per round, final stats_out[0..8] = scores / gains, [9..17] = priors, visits_out[0..8]. */
int oakgpu_bandit_replay(int kind, float c, float alpha, uint32_t k, const float *logits, uint32_t steps,
                         const double *uniforms, const float *values, uint8_t *index_out, float *prob_out,
                         float *stats_out, uint32_t *visits_out);
/* Diagnostic (no GPU involved): `count` select + visit rounds of one bandit from the given state (scores[9], priors[9],
 * visits[9]), once through the register-resident run the batched search uses at the root and once as the plain loop: the two
 * index sequences (run_out, loop_out: count bytes each) and final visit counts must be identical. */
int oakgpu_bandit_select_run(int kind, float c, float alpha, uint32_t k, const float *scores, const float *priors,
                             const uint32_t *visits, uint32_t count, uint8_t *run_out, uint8_t *loop_out, uint32_t *visits_run,
                             uint32_t *visits_loop);
int oakgpu_search(oakgpu_ctx *ctx, oakgpu_net *net /* nullable for eval = 0 */, const uint8_t *battle /* 384 */,
                  const uint8_t *durations /* 8 */, uint8_t result, const oakgpu_search_params *params,
                  oakgpu_search_output *out);

/* ---- RuntimeSearch::Heap (util/search.h:17-32, search.cc:17-58) and the resumable form of Search::run.
 * A heap keeps the tree of a search -- every node's bandit statistics -- between searches:
 *   oakgpu_heap_empty   = Heap::empty(): 1 until a search has used the heap (std::monostate);
 *   oakgpu_heap_update  = Heap::update(i, j, obs) (search.cc:27-52) after the joint action (p1 index i, p2 index j) was PLAYED
 *                         and produced the 16-byte observation `obs` (pkmn_gen1_battle_options_chance_actions): the child
 *                         becomes the root with its whole subtree, the rest is dropped; returns 1.  Returns 0 -- and
 *                         leaves an uninitialised root, like `node = {}` -- when the searches never took that edge, when
 *                         the root was never initialised, or when the heap is empty;
 *   a heap holds one bandit type (the first search fixes it, like the variant, search.cc:205-213): searching it with another
 *   fails with "RuntimeSearch: Bad Heap access. Expecting ...".
 * oakgpu_search_heap = MCTS::Search::run(device, budget, params, heap, eval, input, output) (mcts.h:153-248):
 *   heap      nullable: NULL = a fresh tree for this call (oakgpu_search);
 *   previous  nullable: MCTS::Output is passed BY VALUE and added to -- visit / value matrices, `iterations` and `duration`
 *             accumulate (:231-247), process_output (:620-659) then runs over the sums; MatrixUCB's delay counts
 *             the accumulated iterations (:270).  `previous` and `out` may be the same object.
 *   params->iterations == 0 with duration_us == 0 is legal here: no iteration runs and the output carries the fresh root's
 *   initial_value / logits / priors (what the reference's cpp_inference reads, pyoak.cc:331-392); the empirical fields are
 *   then 0 / 0 as in the reference.  A time budget always runs at least one batch (mcts.h:219-226), and its clock -- like
 *   `duration_us` in the output -- covers the iteration loop only, not the set-up. */
typedef struct oakgpu_heap oakgpu_heap;
int oakgpu_heap_create(oakgpu_heap **out);
void oakgpu_heap_destroy(oakgpu_heap *heap);
int oakgpu_heap_empty(const oakgpu_heap *heap);
void oakgpu_heap_clear(oakgpu_heap *heap);                       /* back to std::monostate */
int oakgpu_heap_kind(const oakgpu_heap *heap);                   /* -1 empty, else oakgpu_search_params.bandit of its nodes */
uint64_t oakgpu_heap_nodes(const oakgpu_heap *heap);
int oakgpu_heap_update(oakgpu_heap *heap, uint8_t i, uint8_t j, const uint8_t *obs16);
/* diagnostics of the host tree's sharding (tests): edges whose child sits in another table's arena (0 in a consistent tree);
 * a host-only self-test (no GPU) that grows a random tree with the search's own threaded resolve phase, promotes a child,
 * checks the shards and grows on: 0 = every check held; out = {nodes before, nodes kept, nodes at the end, shard violations} */
uint64_t oakgpu_heap_check_shards(const oakgpu_heap *heap);
int oakgpu_heap_selftest(uint32_t rounds, uint32_t lanes, uint64_t seed, int threads, uint64_t out[4]);
/* the root's bandit of one player (0 / 1): scores[9] (Exp3: gains), priors[9], visits[9], k (0 = root not initialised) */
int oakgpu_heap_root_stats(const oakgpu_heap *heap, int player, float *scores, float *priors, uint32_t *visits, uint8_t *k);
/* the same view of the child that oakgpu_heap_update(i, j, obs16) would promote; k = 0: no such (initialised) child */
int oakgpu_heap_child_stats(const oakgpu_heap *heap, uint8_t i, uint8_t j, const uint8_t *obs16, int player, float *scores,
                            float *priors, uint32_t *visits, uint8_t *k);
int oakgpu_search_heap(oakgpu_ctx *ctx, oakgpu_net *net, oakgpu_heap *heap, const uint8_t *battle, const uint8_t *durations,
                       uint8_t result, const oakgpu_search_params *params, const oakgpu_search_output *previous,
                       oakgpu_search_output *out);
/* RuntimeSearch::run(device, input, heap, agent, output) (util/search.h:66): oakgpu_search_agent with the heap and the
 * output to resume. */
/* n independent searches at once on ONE GPU: one tree per root (the positions of n self-play games; the reference runs them as N worker
 * threads that never wait for each other, generate.cc:527-536), search i on ctxs[i] -- every search needs a context of its own (its
 * stream and batch slots) and, when heaps != NULL, a heap of its own -- with battles n x 384, durations n x 8, results n, params n
 * entries (seeds!), outs n entries.  Each search is exactly oakgpu_search_heap(ctxs[i], net, heaps[i], ..., NULL, &outs[i]): same output,
 * same heap, whatever runs beside it.  threads_per_search: host threads of each tree walk (1, 2, 4, 8, 16); 0 = the usable cores shared
 * evenly (OAKGPU_SEARCH_CORES overrides the affinity mask's count, e.g. under a cgroup quota). */
int oakgpu_search_many(oakgpu_ctx *const *ctxs, oakgpu_net *net, oakgpu_heap *const *heaps, const uint8_t *battles, const uint8_t *durations,
                       const uint8_t *results, const oakgpu_search_params *params, uint32_t n, int threads_per_search,
                       oakgpu_search_output *outs);
int oakgpu_search_agent_heap(oakgpu_ctx *ctx, oakgpu_heap *heap, const uint8_t *battle, const uint8_t *durations, uint8_t result,
                             const oakgpu_agent *agent, uint32_t batch, uint64_t seed, const oakgpu_search_output *previous,
                             oakgpu_search_output *out);

/* ---- the path's one exchange step: per-root pre-reduction on the device + RCCL all-gather over xGMI.
 * Root-parallel MCTS shards its roots contiguously over the GPUs of a node; each rank reduces its playouts' leaf values to
 * one mean per root (oakgpu_segment_mean_dev: out[s] = mean(values[s * per_segment .. (s + 1) * per_segment))) and ONE
 * ncclAllGather of `count` floats per rank (oakgpu_all_gather_dev, on the context's stream) gives every rank all means:
 * recv = world x count floats, rank-major.  One process per GPU: rank 0 makes the 128-byte id (oakgpu_comm_unique_id),
 * hands it to the others out of band, and every rank calls oakgpu_comm_create.  RCCL is bound at run time (dlopen);
 * without it these calls fail with a message.  The reference has no collective: its playouts are unrelated threads
 * (cpp/src/generate.cc:527-536). */
typedef struct oakgpu_comm oakgpu_comm;
int oakgpu_segment_mean_dev(oakgpu_ctx *ctx, const float *values, uint32_t segments, uint32_t per_segment, float *out);
int oakgpu_comm_unique_id(uint8_t *id128);
int oakgpu_comm_create(oakgpu_ctx *ctx, const uint8_t *id128, int rank, int world, oakgpu_comm **out);
void oakgpu_comm_destroy(oakgpu_comm *comm);
int oakgpu_all_gather_dev(oakgpu_ctx *ctx, oakgpu_comm *comm, const float *send, float *recv, size_t count);

/* ---- root-parallel search steps that do not wait for their longest playout (BASELINE configs[3]).
 * One search step of root-parallel MCTS = `reps` fresh playouts per root (run_root_iteration's prep, mcts.h:250-263, then the
 * rollout loop of mcts.h:448-496) -> one aggregate per root.  The reference's workers never wait for each other
 * (generate.cc:527-536); a step here does not wait for its stragglers either: every launch advances each playout in flight by at
 * most `slice` turn-steps (a power of two; 0 = run to terminal inside the step, the plain rollout's behaviour).  A playout of
 * `len` turn-steps started in step k is credited to step k + (len - 1) / slice (len = 0: step k) -- a function of its own length,
 * not of the schedule -- and travels between launches as a bit-exact state image on a carry list owned by this object.  Values
 * never change, only the step they are credited to.
 *   Streams: lane_prng (n_roots * reps x 8 bytes, device) holds one fast_prng stream per (root, replica), advanced by ONE
 * uniform_64 per step; that draw is the 8-byte state of the fresh playout's own stream (all-zero -> s1 = 1), which supplies
 * battle.rng for the prep and then the choices.
 *   report (device, n_roots + 2 u64, written by the launch): [r] = playouts credited to this step for root r: count | (sum of
 * 2 x value) << 32 -- integers, so independent of the order playouts finish in; [n_roots] = turn-steps this launch executed;
 * [n_roots + 1] = playouts carried into the next step | error word << 32 (bit 0: the carry list overflowed -- playouts were
 * lost; sticky).  fresh = 0 launches a DRAIN step: no new playouts, the carried ones advance one more slice (root_battles must still
 * be the roots' battles, unchanged since the playouts started: a carried playout reads its Pokemon's immutable data from there).
 * When the roots themselves change (the games move on to new positions), drain first -- fresh = 0 until `carried` is 0, at most
 * ceil(max_steps / slice) launches -- or the old roots' stragglers are credited to the new roots' steps.
 * Everything is asynchronous on the context's stream; consecutive launches need no host round trip. */
typedef struct oakgpu_root_steps oakgpu_root_steps;
int oakgpu_root_steps_create(oakgpu_ctx *ctx, uint32_t n_roots, uint32_t reps, uint32_t slice, uint32_t max_steps, oakgpu_root_steps **out);
void oakgpu_root_steps_destroy(oakgpu_root_steps *rs);
int oakgpu_root_steps_launch_dev(oakgpu_root_steps *rs, const uint8_t *root_battles, const uint8_t *root_durations,
                                 const uint8_t *root_results, uint8_t *lane_prng, int fresh, unsigned long long *report);
/* The carry lists hold 1 + 256 / slice steps' worth of playouts by default (3-4x what random OU roots keep in flight).  Roots whose
 * playouts mostly run into the step cap need up to ceil(max_steps / slice) - 1 steps' worth: a caller whose capacity is below
 * 2 x (carried + n_roots x reps) -- what the next launch may carry, doubled for the shards' imbalance -- reserves more BEFORE that launch
 * (synchronises the stream, keeps the playouts in flight; oak_amd.dist.RootSteps does).  An overflow is never
 * silent either way (sticky error word in the report). */
int oakgpu_root_steps_capacity(const oakgpu_root_steps *rs, uint32_t *capacity);
int oakgpu_root_steps_reserve(oakgpu_root_steps *rs, uint64_t playouts);

/* ---- `.battle.data` training frames + self-play on the GPU path.
 * oakgpu_frames_write / _read = Train::Battle::CompressedFrames::write / read (train/battle/compressed-frame.h:37-243):
 * one game = u32 record length, u16 frame count, the 384-byte battle after the opening update, the final result byte,
 * then per turn {(m-1) | (n-1) << 4, c1, c2, u32 iterations, u16 empirical value, u16 nash value, m + m + n + n u16
 * probabilities}; probabilities and values are stored as x * 65535 truncated to u16.  A `.battle.data` file is a plain
 * concatenation of such records.  oakgpu_frames_size = the record's byte length. */
typedef struct {
  uint8_t m, n;          /* legal choices per side at this turn */
  uint8_t c1, c2;        /* the pkmn_choices played */
  uint32_t iterations;
  double empirical_value, nash_value;
  double p1_empirical[9], p1_nash[9], p2_empirical[9], p2_nash[9];
} oakgpu_frame_update;
size_t oakgpu_frames_size(const oakgpu_frame_update *updates, uint32_t count);
int oakgpu_frames_write(const uint8_t *battle /* 384 */, uint8_t result, const oakgpu_frame_update *updates, uint32_t count,
                        uint8_t *buffer, size_t capacity, size_t *written);
int oakgpu_frames_read(const uint8_t *buffer, size_t size, uint8_t *battle /* 384, nullable */, uint8_t *result /* nullable */,
                       oakgpu_frame_update *updates /* nullable */, uint32_t capacity, uint32_t *count, size_t *consumed);
/* One self-play game, the per-game loop of the reference's data generator (cpp/src/generate.cc:238-322): PKMN::battle(teams,
 * battle_seed) + opening update, then per turn oakgpu_search -> RuntimePolicy::process_and_sample for both sides
 * (util/policy.h:22-106; mode words e / n / x / p with optional weights, e.g. "e0.9-x0.1"; p = prior + empirical, the
 * reference's fall-through, useful with the contextual bandits only) -> frame -> update, until the result
 * is terminal; the finished record is written to `buffer`.  Every battle operation runs on the GPU.  Fails (no record
 * written) when the game exceeds max_battle_length updates, like the reference (generate.cc:268-271); 0 = no limit, the reference's
 * default: the engine ends a game at turn 1,000 by itself (records of up to ~1,010 frames). */
typedef struct {
  oakgpu_search_params search; /* seed is replaced per turn from `seed` below */
  char policy_mode[16];
  double policy_temp, policy_min;
  uint32_t max_battle_length;
  uint64_t seed;
  int32_t keep_node;           /* generate's --keep-node (generate.cc:324-333): one heap for the game, Heap::update with the played
                                * indices and the observation after every turn (the subtree is searched on); 0 = a fresh heap per turn */
  uint32_t nodes_kept;         /* out: how many of the game's updates found their child (RuntimeData::update_with_node_counter) */
} oakgpu_selfplay_params;
/* endless_battle_check (cpp/src/generate.cc:127-152): 1 when every pairing of the two teams (2 x 6 x {species, 4 moves}) is a Ghost against a
 * Ghost with no move on either side that can hit a Ghost (type Normal / Fighting or base power 0) -- with ebc=false such a game runs to turn
 * 1,000; the generator draws other teams, and oakgpu_selfplay_game(s) refuse the pair with "EBC check failed". */
int oakgpu_endless_battle_check(const uint8_t *teams);
int oakgpu_selfplay_game(oakgpu_ctx *ctx, oakgpu_net *net /* nullable for eval != 1 */, const uint8_t *teams /* 60 */,
                         uint64_t battle_seed, oakgpu_selfplay_params *params, uint8_t *buffer, size_t capacity,
                         size_t *written, uint32_t *n_frames, uint8_t *result);

/* n self-play games at once on one GPU (the generator's N worker threads, generate.cc:527-536): game g on ctxs[g] -- a context of its own
 * each -- with teams + 60 g, battle_seeds[g], params[g]; its record goes to buffers + g * capacity_each, written[g] / n_frames[g] /
 * results[g] as oakgpu_selfplay_game returns them.  Every game is byte for byte the game it would be alone.  threads_per_game: host
 * threads of each game's tree walks (1, 2, 4, 8, 16); 0 = the usable cores shared evenly. */
int oakgpu_selfplay_games(oakgpu_ctx *const *ctxs, oakgpu_net *net, const uint8_t *teams, const uint64_t *battle_seeds, oakgpu_selfplay_params *params,
                          uint32_t n, int threads_per_game, uint8_t *buffers, size_t capacity_each, size_t *written, uint32_t *n_frames,
                          uint8_t *results);

/* ---- batched PKMN::battle(p1, p2, seed) (pkmn.h:50-57, init.h:90-154), level 100 sets.
 * teams: n x 60 bytes; seeds: n x u64; with first_update != 0 also performs the opening
 * update(battle, 0, 0) (benchmark.cc:29) and writes its result byte. */
int oakgpu_init_battles_dev(oakgpu_ctx *ctx, const uint8_t *teams, const uint64_t *seeds, uint32_t n,
                            int first_update, uint8_t *battles, uint8_t *durations, uint8_t *results);
int oakgpu_init_battles(oakgpu_ctx *ctx, const uint8_t *teams, const uint64_t *seeds, uint32_t n,
                        int first_update, uint8_t *battles, uint8_t *durations, uint8_t *results);

/* ---- SURVEY 8(d) config-2 synthetic input: n random OU team pairs generated ON DEVICE.
 * Lane i: fast_prng::seed(state, seed0 + i); 2 x 6 distinct legal species, min(4, pool)
 * distinct moves each (legal[next32 % 149], pool[next32 % size], rejection on duplicates);
 * battle.rng = uniform_64(); opening update(0,0).  prng_state receives the continuing
 * stream.  ou_legal: 149 species ids; ou_pools: 152 x 48 move ids; ou_sizes: 152. */
int oakgpu_set_ou_pools(oakgpu_ctx *ctx, const uint8_t *ou_legal, int n_legal, const uint8_t *ou_pools,
                        const uint8_t *ou_sizes);
int oakgpu_random_ou_battles_dev(oakgpu_ctx *ctx, uint64_t seed0, uint32_t n, uint8_t *battles,
                                 uint8_t *durations, uint8_t *prng_state, uint8_t *results);

/* ---- leaf evaluator: replaces NN::Battle::NetworkImpl (cpp/include/nn/battle/network.h:22-176).
 * oakgpu_net_load parses a `.battle.net` parameter file exactly as the reference does (8-byte
 * header, byte 0 = activation - 1: cpp/src/search.cc:127-131; then 12 Affine blocks
 * `u32 in, u32 out, f32 bias[out], f32 W[out][in]`: nn/affine.h:35-70; must end at EOF:
 * network.h:60-63) and uploads the weights.  Errors (unreadable file, malformed / trailing
 * bytes, unsupported widths) return non-zero with a message, where the reference throws
 * std::runtime_error (search.cc:81,92,103,138,146).
 * oakgpu_leaf_eval* = value_inference(battle, durations) for n leaves (network.h:72-79):
 * encode + both embedding nets + MainNet value path + sigmoid, fp32 results throughout (the embedding nets' dense layers on
 * fp32 MFMA, the main net's as exact bf16 triples on the bf16 matrix pipe unless oakgpu_net_set_main_precision says
 * otherwise).  embedding_out (nullable): the n x in_dim battle embeddings (network.h:131-175). */
int oakgpu_net_load(oakgpu_ctx *ctx, const char *path, oakgpu_net **out);
int oakgpu_net_load_memory(oakgpu_ctx *ctx, const void *bytes, size_t size, oakgpu_net **out);
void oakgpu_net_free(oakgpu_ctx *ctx, oakgpu_net *net);
/* MainNet::shape() (main-net.h:32-34): fc0.in, fc0.out, value_fc2.out, p1_policy_fc2.out */
int oakgpu_net_shape(const oakgpu_net *net, int *in_dim, int *hidden, int *value_hidden, int *policy_hidden);
/* How the main net's three dense layers are multiplied (results are fp32 either way, held to the same 1e-5 against the
 * oracle and to 1e-6 against a float64 evaluation): OAKGPU_MAIN_PAIR (default since round 5) = every fp32 value times an exact
 * power of two (one per weight row = output feature, one per batch row for the activations) as the sum of two round-to-nearest fp16
 * parts, three fp16 MFMAs per multiply-add block, fp32 accumulation (k_mainnet_pair); OAKGPU_MAIN_SPLIT = every fp32 value as
 * an exact sum of three bf16 parts, six bf16 MFMAs per block (k_mainnet_split: 190-200 us per 65,536 leaves); OAKGPU_MAIN_FP32 =
 * fp32 MFMA (k_mainnet_wave: 336 us).  The environment variable OAKGPU_MAIN_NET=fp32 / bf16x3 makes one of the latter the
 * default of networks loaded after it is set.  Returns the previous mode, -1 on a bad argument. */
#define OAKGPU_MAIN_FP32 0
#define OAKGPU_MAIN_SPLIT 1
#define OAKGPU_MAIN_PAIR 2
int oakgpu_net_set_main_precision(oakgpu_net *net, int mode);
/* Edges of the bf16-triple form.  (1) A parameter file with a NaN / inf anywhere is REFUSED by oakgpu_net_load* ("non-finite
 * parameter ... in <layer>"); the reference loads it and propagates NaN (nn/affine.h:72-85 is a plain fp32 W x + b).  (2) A
 * triple carries a value to fp32 accuracy only while its low parts are normal bf16 numbers (|x| >= ~2^-102); what a flushed
 * part loses is at most 2^-126 per factor, harmless unless later layers multiply it back up.  So a network with a main-net
 * weight (fc0 / fc1 / value_fc2) above 2^20 in magnitude runs its main net on fp32 MFMA, and a request for OAKGPU_MAIN_SPLIT
 * is not honoured for it (tests/test_gpu_leafnet.py: layers scaled by 2^-100 / 2^-120 / 2^+100, alone and compensated).  The
 * embedding nets' passes multiply as bf16 triples as well; a weight above 2^20 in their second layers or in the main net sends them
 * through the fp32-MFMA form (k_embed_lds) instead.  (3) The fp16 pairs are scaled per weight row and per batch row, so no absolute
 * magnitude matters to them; what must hold is that every weight ROW survives the pairing to fp32 accuracy under its own scale (the
 * sum of what its pairs miss within 2^-23 of the sum of its magnitudes: only rows beyond 2^+-100 fail) and
 * that no non-zero weight COLUMN lies more than 2^18 below the layer's largest weight (a network that compensates tiny weights
 * with huge inputs is the same function in fp32 but not in a 5-bit exponent) -- a network that fails either runs on the triples, or
 * on fp32 MFMA by (2).  Returns the mode in effect (-1: null net); *split_allowed
 * (nullable) = 0 for a network of case (2). */
int oakgpu_net_main_precision(const oakgpu_net *net, int *split_allowed);
int oakgpu_leaf_eval_dev(oakgpu_ctx *ctx, oakgpu_net *net, const uint8_t *battles, const uint8_t *durations,
                         uint32_t n, float *values, float *embedding_out);
int oakgpu_leaf_eval(oakgpu_ctx *ctx, oakgpu_net *net, const uint8_t *battles, const uint8_t *durations, uint32_t n,
                     float *values, float *embedding_out);

/* value_inference for a RESIDENT batch evaluated again and again (BASELINE configs[2]: every lane, every turn), with the
 * party-slot embeddings cached like NN::Battle::PokemonCache (nn/battle/cache.h:18-131, key encode/battle/key.h:65-71):
 * the caller keeps `embedding` (n x in_dim floats) and `slot_tags` (n x 10 x OAKGPU_SLOT_TAG_WORDS u32) between calls; a
 * bench slot is re-embedded only when what its embedding depends on changed -- the stored Pokemon's bytes with PP reduced
 * to has-PP bits and the status to its encoder index, compared exactly (no hashing).  Fill slot_tags with 0xFF bytes before
 * the first call; after that nothing needs resetting, not even when other battles are put into the lanes (a tag holds the
 * slot's whole identity).  Results are identical to oakgpu_leaf_eval_dev. */
#define OAKGPU_SLOT_TAG_WORDS 6
int oakgpu_leaf_eval_cached_dev(oakgpu_ctx *ctx, oakgpu_net *net, const uint8_t *battles, const uint8_t *durations, uint32_t n,
                                float *values, float *embedding, uint32_t *slot_tags);
/* Diagnostic (synchronises the context's stream): how many party slots the LAST oakgpu_leaf_eval_cached_dev call of this
 * context re-embedded (the cache misses of nn/battle/cache.h:81-126), of the n x 10 it looked at. */
int oakgpu_leaf_cache_last_count(oakgpu_ctx *ctx, uint32_t *slots_recomputed);

/* value_policy_inference (network.h:102-123): value plus, per side, the logits of the <= 9 legal choices
 * (choices n x 9 bytes + counts n bytes per side, as produced by oakgpu_choices*; logits n x 9 floats,
 * entries past the count are 0).  Choice -> policy row via Encode::Battle::Policy::get_index
 * (encode/battle/policy.h:29-58): move -> stored move id - 1, switch -> 164 + species - 1, pass -> 0. */
int oakgpu_leaf_eval_policy_dev(oakgpu_ctx *ctx, oakgpu_net *net, const uint8_t *battles, const uint8_t *durations,
                                uint32_t n, const uint8_t *p1_choices, const uint8_t *p1_counts,
                                const uint8_t *p2_choices, const uint8_t *p2_counts, float *values, float *p1_logits,
                                float *p2_logits);
int oakgpu_leaf_eval_policy(oakgpu_ctx *ctx, oakgpu_net *net, const uint8_t *battles, const uint8_t *durations, uint32_t n,
                            const uint8_t *p1_choices, const uint8_t *p1_counts, const uint8_t *p2_choices,
                            const uint8_t *p2_counts, float *values, float *p1_logits, float *p2_logits);

#ifdef __cplusplus
}
#endif
#endif
