// oak_amd/csrc/selfplay.hip -- `.battle.data` training frames and the self-play game loop over the GPU hot path (host code).
//
// the sink of the search results.  Two pieces of the reference are mirrored here:
//   * Train::Battle::CompressedFrames (cpp/include/train/battle/compressed-frame.h:37-223): the on-disk record of one
//     game -- u32 byte length, u16 frame count, the 384-byte battle after the opening update, the final result byte, then
//     per turn an Update {mn byte = (m-1) | (n-1) << 4, both chosen pkmn_choices, u32 iterations, u16 empirical value,
//     u16 nash value, m u16 empirical + m u16 nash probabilities of P1, n + n of P2}; probabilities and values are stored
//     as `x * 65535` truncated to u16 (compress_probs, :11-25).  oakgpu_frames_write / _read are its write() / read().
//   * the per-game loop of the data generator (cpp/src/generate.cc:238-322): PKMN::battle + opening update, then per turn
//     RuntimeSearch::run -> RuntimePolicy::process_and_sample for both sides (cpp/include/util/policy.h:22-106) ->
//     frame -> PKMN::update, until the result is terminal.  oakgpu_selfplay_game runs it with every battle operation on
//     the GPU: oakgpu_init_battles, oakgpu_search (batched leaves), oakgpu_update.
// The replay self-check of the reference (cpp/include/py/battle/frames.h:52-67: replaying the stored choices from the
// stored battle must reproduce the stored result) is tests/test_gpu_frames.py.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <cmath>
#include <string>
#include <thread>
#include <vector>

#include "../../include/oakgpu.h"
#include "oakgpu_internal.h"

namespace {

uint16_t compress_prob(double x) { // compress_probs<double, uint16_t> (compressed-frame.h:11-25)
  const double v = x * 65535.0;
  return v <= 0 ? 0 : v >= 65535.0 ? 65535 : (uint16_t)v;
}
size_t update_bytes(uint32_t m, uint32_t n) { return 1 + 2 + 4 + 2 * 2 + 2 * (m + n) * 2; } // Update::n_bytes_static (:77-82)

template <class T> void put(uint8_t *&p, T v) { memcpy(p, &v, sizeof v); p += sizeof v; }
template <class T> T get(const uint8_t *&p) { T v; memcpy(&v, p, sizeof v); p += sizeof v; return v; }

uint64_t splitmix64(uint64_t &x) {
  uint64_t z = (x += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
double uniform01(uint64_t &rng) { return (double)(splitmix64(rng) >> 11) * (1.0 / 9007199254740992.0); }

// RuntimePolicy::get_policy (util/policy.h:22-98) for the modes the search output can feed: words of the mode string are
// 'e' (empirical), 'n' (nash), 'x' (argmax of empirical), each optionally followed by a weight ("e0.9-x0.1"); then
// temperature, the minimum-probability cut and renormalisation.  'p' (prior, the contextual bandits' softmaxed logits) adds
// w * prior AND THEN w * empirical: the reference's `case Mode::prior` has no `break` and falls into the empirical case
// (policy.h:37-46), and a drop-in follows what the code does.  'b' (beta) throws in the reference (policy.h:59-60) and is
// refused here, like an unknown mode character.
bool get_policy(const double *prior, const double *empirical, const double *nash, int k, const char *mode, double temp, double minp, double *policy) {
  for (int i = 0; i < 9; ++i) policy[i] = 0;
  const std::string s(mode && *mode ? mode : "e");
  size_t pos = 0;
  while (pos <= s.size()) {
    size_t end = s.find('-', pos);
    if (end == std::string::npos) end = s.size();
    const std::string word = s.substr(pos, end - pos);
    pos = end + 1;
    if (word.empty()) continue;
    const double w = word.size() > 1 ? atof(word.c_str() + 1) : 1.0;
    if (word[0] == 'p') { for (int i = 0; i < k; ++i) policy[i] += w * prior[i] + w * empirical[i]; }
    else if (word[0] == 'e') { for (int i = 0; i < k; ++i) policy[i] += w * empirical[i]; }
    else if (word[0] == 'n') { for (int i = 0; i < k; ++i) policy[i] += w * nash[i]; }
    else if (word[0] == 'x') { policy[std::max_element(empirical, empirical + k) - empirical] += w; }
    else return false;
  }
  if (temp != 1) {
    double sum = 0;
    for (int i = 0; i < 9; ++i) { policy[i] = std::pow(policy[i], temp); sum += policy[i]; }
    for (int i = 0; i < 9; ++i) policy[i] /= sum;
  }
  double sum = 0;
  for (int i = 0; i < 9; ++i) { if (policy[i] < minp) policy[i] = 0; sum += policy[i]; }
  if (!(sum > 0)) return false;
  for (int i = 0; i < 9; ++i) policy[i] /= sum;
  return true;
}
int sample_pdf(const double *p, int k, uint64_t &rng) { // device.sample_pdf (util/random.h:40-49)
  double u = uniform01(rng);
  for (int i = 0; i < k; ++i) { u -= p[i]; if (u <= 0) return i; }
  return 0;
}

// endless_battle_check (generate.cc:127-152): with ebc=false a battle of Ghosts that cannot hurt each other never ends before turn 1,000.
// Moves that cannot hit a Ghost: type Normal or Fighting, or base power 0 (the reference's own test) -- one bit per move id, from
// oak_amd/data/gen1_data.json (tests/test_frames.py recomputes the mask from that file).
const uint64_t CANT_HIT_GHOST[3] = {0x8047f8ffffb5fc7full, 0x03ffffdf9442e67cull, 0x0000003fd6dd5bfeull};
bool set_is_ghost(const uint8_t *set) { return set[0] == 92 || set[0] == 93 || set[0] == 94; } // Gastly, Haunter, Gengar
bool set_cant_hit_ghosts(const uint8_t *set) {
  for (int k = 1; k <= 4; ++k) { const uint32_t m = set[k]; if (m > 165 || !((CANT_HIT_GHOST[m >> 6] >> (m & 63)) & 1)) return false; }
  return true;
}

} // namespace

extern "C" {

// 1 when EVERY pairing of the two teams is Ghost against Ghost with no move on either side that can hit a Ghost -- the match-ups the
// reference's generator rejects up front (generate.cc:127-152, 222-225): teams = 2 x 6 x {species, 4 moves}.
int oakgpu_endless_battle_check(const uint8_t *teams) {
  if (!teams) return 0;
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j < 6; ++j) {
      const uint8_t *a = teams + 5 * i, *b = teams + 30 + 5 * j;
      if (!(set_is_ghost(a) && set_cant_hit_ghosts(b) && set_is_ghost(b) && set_cant_hit_ghosts(a))) return 0;
    }
  return 1;
}

size_t oakgpu_frames_size(const oakgpu_frame_update *updates, uint32_t count) { // CompressedFrames::n_bytes (:180-187)
  size_t n = 4 + 2 + 384 + 1;
  for (uint32_t i = 0; i < count; ++i) n += update_bytes(updates[i].m, updates[i].n);
  return n;
}

int oakgpu_frames_write(const uint8_t *battle, uint8_t result, const oakgpu_frame_update *updates, uint32_t count, uint8_t *buffer,
                        size_t capacity, size_t *written) {
  if (!battle || (!updates && count) || !buffer) return oakgpu_fail_msg("oakgpu_frames_write: null argument");
  if (count > 0xFFFF) return oakgpu_fail_msg("oakgpu_frames_write: more than 65535 frames in one game (FrameCount is u16)");
  for (uint32_t i = 0; i < count; ++i)
    if (updates[i].m < 1 || updates[i].m > 9 || updates[i].n < 1 || updates[i].n > 9) return oakgpu_fail_msg("oakgpu_frames_write: m, n must be in 1..9");
  const size_t total = oakgpu_frames_size(updates, count);
  if (total > capacity) return oakgpu_fail_msg("oakgpu_frames_write: buffer too small");
  uint8_t *p = buffer;
  put<uint32_t>(p, (uint32_t)total);          // Offset: byte length of this game's record (:190-191)
  put<uint16_t>(p, (uint16_t)count);          // FrameCount
  memcpy(p, battle, 384); p += 384;
  *p++ = result;
  for (uint32_t i = 0; i < count; ++i) {      // Update::write (:89-113)
    const oakgpu_frame_update &u = updates[i];
    *p++ = (uint8_t)((u.m - 1) | ((u.n - 1) << 4));
    *p++ = u.c1;
    *p++ = u.c2;
    put<uint32_t>(p, u.iterations);
    put<uint16_t>(p, compress_prob(u.empirical_value));
    put<uint16_t>(p, compress_prob(u.nash_value));
    for (int k = 0; k < u.m; ++k) put<uint16_t>(p, compress_prob(u.p1_empirical[k]));
    for (int k = 0; k < u.m; ++k) put<uint16_t>(p, compress_prob(u.p1_nash[k]));
    for (int k = 0; k < u.n; ++k) put<uint16_t>(p, compress_prob(u.p2_empirical[k]));
    for (int k = 0; k < u.n; ++k) put<uint16_t>(p, compress_prob(u.p2_nash[k]));
  }
  if (written) *written = total;
  return 0;
}

int oakgpu_frames_read(const uint8_t *buffer, size_t size, uint8_t *battle, uint8_t *result, oakgpu_frame_update *updates,
                       uint32_t capacity, uint32_t *count, size_t *consumed) {
  if (!buffer || !count) return oakgpu_fail_msg("oakgpu_frames_read: null argument");
  if (size < 4 + 2 + 384 + 1) return oakgpu_fail_msg("oakgpu_frames_read: truncated record");
  const uint8_t *p = buffer;
  const uint32_t total = get<uint32_t>(p);
  const uint16_t frames = get<uint16_t>(p);
  if (total > size || total < 4 + 2 + 384 + 1) return oakgpu_fail_msg("oakgpu_frames_read: record length out of range");
  if (battle) memcpy(battle, p, 384);
  p += 384;
  if (result) *result = *p;
  ++p;
  uint32_t n_read = 0;
  while ((size_t)(p - buffer) < total) {      // CompressedFrames::read (:224-243)
    if ((size_t)(buffer + total - p) < 3) return oakgpu_fail_msg("oakgpu_frames_read: truncated update");
    const uint8_t mn = *p;
    const uint32_t m = (mn & 15) + 1, n = (mn >> 4) + 1;
    if (m > 9 || n > 9 || (size_t)(buffer + total - p) < update_bytes(m, n)) return oakgpu_fail_msg("oakgpu_frames_read: malformed update");
    ++p;
    oakgpu_frame_update u{};
    u.m = (uint8_t)m; u.n = (uint8_t)n;
    u.c1 = *p++; u.c2 = *p++;
    u.iterations = get<uint32_t>(p);
    u.empirical_value = get<uint16_t>(p) / 65535.0; // uncompress_probs (:27-35)
    u.nash_value = get<uint16_t>(p) / 65535.0;
    for (uint32_t k = 0; k < m; ++k) u.p1_empirical[k] = get<uint16_t>(p) / 65535.0;
    for (uint32_t k = 0; k < m; ++k) u.p1_nash[k] = get<uint16_t>(p) / 65535.0;
    for (uint32_t k = 0; k < n; ++k) u.p2_empirical[k] = get<uint16_t>(p) / 65535.0;
    for (uint32_t k = 0; k < n; ++k) u.p2_nash[k] = get<uint16_t>(p) / 65535.0;
    if (updates && n_read < capacity) updates[n_read] = u;
    ++n_read;
  }
  if (n_read != frames) return oakgpu_fail_msg("oakgpu_frames_read: frame count does not match the record");
  if (updates && n_read > capacity) return oakgpu_fail_msg("oakgpu_frames_read: more frames than the caller's capacity");
  *count = n_read;
  if (consumed) *consumed = total;
  return 0;
}

int oakgpu_selfplay_game(oakgpu_ctx *ctx, oakgpu_net *net, const uint8_t *teams, uint64_t battle_seed, oakgpu_selfplay_params *prm,
                         uint8_t *buffer, size_t capacity, size_t *written, uint32_t *n_frames, uint8_t *result_out) {
  if (!ctx || !teams || !prm || !buffer) return oakgpu_fail_msg("oakgpu_selfplay_game: null argument");
  // generate.cc:222-225: such a pair is not played (the reference prints "EBC check failed. Continuing." and draws other teams; the teams
  // are the caller's here, so the caller is told)
  if (oakgpu_endless_battle_check(teams)) return oakgpu_fail_msg("oakgpu_selfplay_game: EBC check failed (every match-up is Ghost against Ghost with no move that can hit a Ghost, generate.cc:127-152)");
  // PKMN::battle(p1, p2, seed) + the opening update(0, 0) (generate.cc:238-240), on the device
  uint8_t battle[384], durations[8] = {}, result = 0;
  if (int rc = oakgpu_init_battles(ctx, teams, &battle_seed, 1, 1, battle, durations, &result)) return rc;
  uint8_t first[384];
  memcpy(first, battle, 384); // CompressedFrames{battle_data.battle} (generate.cc:244)
  std::vector<oakgpu_frame_update> frames;
  uint64_t rng = prm->seed ^ 0x9FB21C651E98DF25ull;
  // 0 = no limit, the reference's default (--max-battle-length -1, generate.cc:49-52): the engine itself ends a game at turn 1,000,
  // and only the handful of forced-switch updates come on top of that (2,048 is a backstop against a bug, never a rule of the game)
  const uint32_t max_len = prm->max_battle_length ? prm->max_battle_length : 2048;
  oakgpu_search_params sp = prm->search;
  struct HeapOwner { oakgpu_heap *h = nullptr; ~HeapOwner() { oakgpu_heap_destroy(h); } } heap;
  if (prm->keep_node && oakgpu_heap_create(&heap.h)) return -1;
  prm->nodes_kept = 0;
  while ((result & 15) == 0) {
    if (frames.size() >= max_len) return oakgpu_fail_msg("oakgpu_selfplay_game: max battle length exceeded (generate.cc:268-271)");
    oakgpu_search_output out;
    sp.seed = splitmix64(rng);
    int src = oakgpu_search_heap(ctx, net, heap.h, battle, durations, result, &sp, nullptr, &out);
    if (src == OAKGPU_E_ROOT_MISMATCH && heap.h) {
      // --keep-node: the kept child was expanded by a playout whose RESAMPLED hidden variables (mcts.h:254-259) gave it other
      // legal choices than the position the game really reached (a thrash / bide counter that ran out in one and not in the
      // other).  Its statistics are about another decision: start this position's tree afresh, like an update that found no child.
      oakgpu_heap_clear(heap.h);
      if (prm->nodes_kept) --prm->nodes_kept;
      src = oakgpu_search_heap(ctx, net, heap.h, battle, durations, result, &sp, nullptr, &out);
    }
    if (src) return src;
    double pol1[9], pol2[9];
    if (!get_policy(out.p1_prior, out.p1_empirical, out.p1_nash, out.m, prm->policy_mode, prm->policy_temp > 0 ? prm->policy_temp : 1.0, prm->policy_min, pol1) ||
        !get_policy(out.p2_prior, out.p2_empirical, out.p2_nash, out.n, prm->policy_mode, prm->policy_temp > 0 ? prm->policy_temp : 1.0, prm->policy_min, pol2))
      return oakgpu_fail_msg("oakgpu_selfplay_game: policy mode must be built from e / n / x words and leave a non-zero policy (util/policy.h:22-98)");
    const int i = sample_pdf(pol1, out.m, rng), j = sample_pdf(pol2, out.n, rng);
    oakgpu_frame_update u{};
    u.m = out.m; u.n = out.n;
    u.c1 = out.p1_choices[i]; u.c2 = out.p2_choices[j];
    u.iterations = (uint32_t)out.iterations;
    u.empirical_value = out.empirical_value;
    u.nash_value = out.nash_value;
    for (int k = 0; k < 9; ++k) { u.p1_empirical[k] = out.p1_empirical[k]; u.p1_nash[k] = out.p1_nash[k]; u.p2_empirical[k] = out.p2_empirical[k]; u.p2_nash[k] = out.p2_nash[k]; }
    frames.push_back(u);
    // PKMN::update(battle, c1, c2, options); durations <- options (generate.cc:319-322)
    uint8_t obs[16];
    if (int rc = oakgpu_update(ctx, battle, &u.c1, &u.c2, durations, obs, nullptr, 1, &result)) return rc;
    if (heap.h) prm->nodes_kept += (uint32_t)oakgpu_heap_update(heap.h, (uint8_t)i, (uint8_t)j, obs); // generate.cc:324-329
  }
  if (result_out) *result_out = result;
  if (n_frames) *n_frames = (uint32_t)frames.size();
  return oakgpu_frames_write(first, result, frames.data(), (uint32_t)frames.size(), buffer, capacity, written);
}

// n self-play games at once on one GPU -- the generator's worker pool (generate.cc:527-536: N threads, one game each, never waiting
// for each other): game g runs oakgpu_selfplay_game on ctxs[g] (a context of its own each) with teams[g], battle_seeds[g], params[g],
// its record into buffers + g * capacity_each.  One game at a time leaves the card idle for most of every search (the tree walk is host
// work); n games keep n trees in flight.  Every game is the game it would be alone.  threads_per_game: host threads of each game's tree
// walks (1, 2, 4, 8); 0 = the usable cores shared evenly (OAKGPU_SEARCH_CORES overrides the affinity mask's count).
int oakgpu_selfplay_games(oakgpu_ctx *const *ctxs, oakgpu_net *net, const uint8_t *teams, const uint64_t *battle_seeds, oakgpu_selfplay_params *params,
                          uint32_t n, int threads_per_game, uint8_t *buffers, size_t capacity_each, size_t *written, uint32_t *n_frames, uint8_t *results) {
  if (n == 0) return 0;
  if (!ctxs || !teams || !battle_seeds || !params || !buffers || !written) return oakgpu_fail_msg("oakgpu_selfplay_games: null argument");
  for (uint32_t g = 0; g < n; ++g) {
    if (!ctxs[g]) return oakgpu_fail_msg("oakgpu_selfplay_games: null context");
    for (uint32_t k = 0; k < g; ++k) if (ctxs[k] == ctxs[g]) return oakgpu_fail_msg("oakgpu_selfplay_games: every game needs a context of its own");
  }
  int W = threads_per_game;
  if (W <= 0) W = (int)(oakgpu_usable_cores() / n);
  W = W >= 16 ? 16 : W >= 8 ? 8 : W >= 4 ? 4 : W >= 2 ? 2 : 1;
  std::vector<int> rc(n, 0);
  std::vector<std::string> err(n);
  std::vector<std::thread> th;
  th.reserve(n);
  for (uint32_t g = 0; g < n; ++g)
    th.emplace_back([&, g] {
      oakgpu_set_thread_search_threads(W);
      rc[g] = oakgpu_selfplay_game(ctxs[g], net, teams + (size_t)g * 60, battle_seeds[g], params + g, buffers + (size_t)g * capacity_each, capacity_each,
                                   written + g, n_frames ? n_frames + g : nullptr, results ? results + g : nullptr);
      if (rc[g]) err[g] = oakgpu_last_error();
    });
  for (auto &t : th) t.join();
  for (uint32_t g = 0; g < n; ++g)
    if (rc[g]) return oakgpu_fail_msg(("oakgpu_selfplay_games: game " + std::to_string(g) + ": " + err[g]).c_str());
  return 0;
}

} // extern "C"
