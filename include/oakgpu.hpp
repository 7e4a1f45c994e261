// include/oakgpu.hpp -- thin C++ host layer over the C ABI (include/oakgpu.h), shaped like the
// reference's search-node surface so Oak's tree code can use it as an `eval` (cpp/include/search/mcts.h:25-57).
//
//   OakGPU::Context            <- per-thread owner of device state (reference: per-thread Agent/Heap, util/search.h:17-64)
//   OakGPU::Network            <- NN::Battle::Network (nn/battle/network.h:22-176): shape(), value_inference(batch), and the
//                                 reference's own per-leaf eval signatures value_inference(battle, durations) /
//                                 value_policy_inference(b, d, m, n, c1, c2, p1, p2) that MCTS::Search::run calls (mcts.h:196-209,401-422)
//   OakGPU::BatchedMonteCarlo  <- MCTS::MonteCarlo (mcts.h:21-23) + init_stats_and_rollout (mcts.h:448-496), batched
//   OakGPU::TreeSearch         <- MCTS::Search::run (mcts.h:154-247): Node heap + the five joint bandits, leaves batched on the GPU
//   OakGPU::run                <- RuntimeSearch::run (util/search.h:66, search.cc:150-313): the Agent's strings pick everything
//   OakGPU::solve_matrix       <- LRSNash::solve_fast as called at mcts.h:643-649 / pyoak.cc:394-426 (exact)
//   OakGPU::SharedDeviceRollout<- benchmark.cc:23-31: n playouts from one root driven by ONE sequential std::mt19937
//   OakGPU::Frames             <- Train::Battle::CompressedFrames (train/battle/compressed-frame.h:37-243)
//   OakGPU::Exchange           <- the path's one collective: per-root means on the device + RCCL all-gather (no reference analogue)
// Errors surface as std::runtime_error, like the reference's loaders (cpp/src/search.cc:81-146).
#pragma once
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <tuple>
#include <vector>

#include "oakgpu.h"
#include "pkmn.h"

namespace OakGPU {

inline void check(int rc) {
  if (rc != 0) throw std::runtime_error{std::string{"oakgpu: "} + oakgpu_last_error()};
}

class Context {
public:
  explicit Context(int device = 0) { check(oakgpu_create(&ctx_, device)); }
  ~Context() { oakgpu_destroy(ctx_); }
  Context(const Context &) = delete;
  Context &operator=(const Context &) = delete;
  oakgpu_ctx *get() const noexcept { return ctx_; }
  void synchronize() { check(oakgpu_synchronize(ctx_)); }

private:
  oakgpu_ctx *ctx_{};
};

// One leaf of the batch: what MCTS::Input carries (mcts.h:62-66)
struct Leaf {
  uint8_t battle[OAKGPU_BATTLE_SIZE];
  uint8_t durations[OAKGPU_DURATIONS_SIZE];
  uint8_t result;
};

struct RolloutResult {
  std::vector<float> value;     // 1 / 0 / 0.5 (mcts.h:481-495)
  std::vector<uint32_t> steps;  // turn-steps played
  std::vector<uint8_t> result;  // final pkmn_result byte
};

class BatchedMonteCarlo {
public:
  explicit BatchedMonteCarlo(Context &ctx) : ctx_{ctx} {}
  // Several independent batches (e.g. the leaf batches of many roots) drained by ONE launch through one playout queue:
  // a group has one tail instead of one per batch.  Device pointers; asynchronous on the context's stream.
  void rollout_group_dev(const std::vector<oakgpu_rollout_batch> &batches, bool prep = false, uint32_t max_steps = 1000) {
    check(oakgpu_rollout_group_dev(ctx_.get(), batches.data(), static_cast<uint32_t>(batches.size()), max_steps, prep ? 1 : 0));
  }
  // device_rng: one fast_prng state (8 bytes, util/random.h:67-133) per leaf, advanced in place.
  // prep = true performs run_root_iteration's re-seed + randomize_hidden_variables per leaf (mcts.h:254-259).
  RolloutResult rollout(const std::vector<Leaf> &leaves, std::vector<uint64_t> &device_rng, bool prep = false,
                        uint32_t max_steps = 1000) {
    const uint32_t n = static_cast<uint32_t>(leaves.size());
    if (device_rng.size() != n) throw std::runtime_error{"oakgpu: one RNG state per leaf required"};
    std::vector<uint8_t> battles(size_t{n} * OAKGPU_BATTLE_SIZE), durations(size_t{n} * OAKGPU_DURATIONS_SIZE), results(n);
    for (uint32_t i = 0; i < n; ++i) {
      std::memcpy(&battles[size_t{i} * OAKGPU_BATTLE_SIZE], leaves[i].battle, OAKGPU_BATTLE_SIZE);
      std::memcpy(&durations[size_t{i} * OAKGPU_DURATIONS_SIZE], leaves[i].durations, OAKGPU_DURATIONS_SIZE);
      results[i] = leaves[i].result;
    }
    RolloutResult out{std::vector<float>(n), std::vector<uint32_t>(n), std::vector<uint8_t>(n)};
    check(oakgpu_rollout(ctx_.get(), battles.data(), durations.data(), results.data(), reinterpret_cast<uint8_t *>(device_rng.data()), n,
                         max_steps, prep ? 1 : 0, out.result.data(), out.steps.data(), out.value.data(), nullptr, nullptr));
    return out;
  }

private:
  Context &ctx_;
};

class Network {
public:
  Network(Context &ctx, const std::string &path) : ctx_{ctx} { check(oakgpu_net_load(ctx_.get(), path.c_str(), &net_)); }
  ~Network() { oakgpu_net_free(ctx_.get(), net_); }
  Network(const Network &) = delete;
  Network &operator=(const Network &) = delete;
  // MainNet::shape(): fc0.in, fc0.out, value_fc2.out, p1_policy_fc2.out (main-net.h:32-34)
  std::tuple<int, int, int, int> shape() const {
    int a = 0, b = 0, c = 0, d = 0;
    check(oakgpu_net_shape(net_, &a, &b, &c, &d));
    return {a, b, c, d};
  }
  // how the main net's dense layers are multiplied: OAKGPU_MAIN_PAIR (default: scaled fp16 pairs, fp32 results), OAKGPU_MAIN_SPLIT (bf16 triples) or OAKGPU_MAIN_FP32
  int set_main_precision(int mode) { return oakgpu_net_set_main_precision(net_, mode); }
  // value_inference(battle, durations) for every leaf (network.h:72-79)
  std::vector<float> value_inference(const std::vector<Leaf> &leaves) {
    const uint32_t n = static_cast<uint32_t>(leaves.size());
    std::vector<uint8_t> battles(size_t{n} * OAKGPU_BATTLE_SIZE), durations(size_t{n} * OAKGPU_DURATIONS_SIZE);
    for (uint32_t i = 0; i < n; ++i) {
      std::memcpy(&battles[size_t{i} * OAKGPU_BATTLE_SIZE], leaves[i].battle, OAKGPU_BATTLE_SIZE);
      std::memcpy(&durations[size_t{i} * OAKGPU_DURATIONS_SIZE], leaves[i].durations, OAKGPU_DURATIONS_SIZE);
    }
    std::vector<float> values(n);
    check(oakgpu_leaf_eval(ctx_.get(), net_, battles.data(), durations.data(), n, values.data(), nullptr));
    return values;
  }

  // ---- the reference's per-leaf `eval` signatures (nn/battle/network.h:72-79, 102-123), what an UNCHANGED MCTS::Search::run
  // calls at every new leaf (mcts.h:401-422) and at a fresh contextual root (mcts.h:196-209).  Each is a batch of ONE through the
  // same kernels as the batched forms (like include/pkmn.h's update / choices): correct and within the same 1e-5 / 2e-5 of
  // the oracle, but it costs three kernel launches and four small PCIe copies per leaf (~60-100 us against 15 ns per leaf in
  // a 65,536-leaf batch).  It exists so that the network drops into the reference's sequential search unchanged; throughput
  // comes from the batched forms above and from OakGPU::TreeSearch.
  float value_inference(const pkmn_gen1_battle &b, const pkmn_gen1_chance_durations &d) {
    float value = 0.0f;
    check(oakgpu_leaf_eval(ctx_.get(), net_, b.bytes, d.bytes, 1, &value, nullptr));
    return value;
  }
  // m / n legal choices of the two sides (as pkmn_gen1_battle_choices returned them); p1 / p2 receive their m / n logits
  template <class Count, class Choice>
  float value_policy_inference(const pkmn_gen1_battle &b, const pkmn_gen1_chance_durations &d, const Count m, const Count n,
                               const Choice *p1_choice, const Choice *p2_choice, float *p1, float *p2) {
    if (m > 9 || n > 9) throw std::runtime_error{"oakgpu: more than PKMN_GEN1_MAX_CHOICES choices"};
    uint8_t c1[9] = {}, c2[9] = {};
    const uint8_t k1 = static_cast<uint8_t>(m), k2 = static_cast<uint8_t>(n);
    for (uint8_t i = 0; i < k1; ++i) c1[i] = static_cast<uint8_t>(p1_choice[i]);
    for (uint8_t i = 0; i < k2; ++i) c2[i] = static_cast<uint8_t>(p2_choice[i]);
    float value = 0.0f, l1[9], l2[9];
    check(oakgpu_leaf_eval_policy(ctx_.get(), net_, b.bytes, d.bytes, 1, c1, &k1, c2, &k2, &value, l1, l2));
    for (uint8_t i = 0; i < k1; ++i) p1[i] = l1[i];
    for (uint8_t i = 0; i < k2; ++i) p2[i] = l2[i];
    return value;
  }
  // value_policy_inference for every leaf (network.h:102-123): choices / counts as pkmn_gen1_battle_choices fills them (n x 9
  // bytes, n bytes per side); returns the values, fills the n x 9 logit arrays (entries past a count are 0)
  std::vector<float> value_policy_inference(const std::vector<Leaf> &leaves, const std::vector<uint8_t> &p1_choices,
                                            const std::vector<uint8_t> &p1_counts, const std::vector<uint8_t> &p2_choices,
                                            const std::vector<uint8_t> &p2_counts, std::vector<float> &p1_logits, std::vector<float> &p2_logits) {
    const uint32_t n = static_cast<uint32_t>(leaves.size());
    if (p1_choices.size() != size_t{n} * 9 || p2_choices.size() != size_t{n} * 9 || p1_counts.size() != n || p2_counts.size() != n)
      throw std::runtime_error{"oakgpu: choices must be n x 9 bytes and counts n bytes per side"};
    std::vector<uint8_t> battles(size_t{n} * OAKGPU_BATTLE_SIZE), durations(size_t{n} * OAKGPU_DURATIONS_SIZE);
    for (uint32_t i = 0; i < n; ++i) {
      std::memcpy(&battles[size_t{i} * OAKGPU_BATTLE_SIZE], leaves[i].battle, OAKGPU_BATTLE_SIZE);
      std::memcpy(&durations[size_t{i} * OAKGPU_DURATIONS_SIZE], leaves[i].durations, OAKGPU_DURATIONS_SIZE);
    }
    std::vector<float> values(n);
    p1_logits.assign(size_t{n} * 9, 0.0f);
    p2_logits.assign(size_t{n} * 9, 0.0f);
    check(oakgpu_leaf_eval_policy(ctx_.get(), net_, battles.data(), durations.data(), n, p1_choices.data(), p1_counts.data(), p2_choices.data(),
                                  p2_counts.data(), values.data(), p1_logits.data(), p2_logits.data()));
    return values;
  }

  // A RESIDENT batch evaluated every turn (device pointers): party-slot embeddings cached by exact identity tags, the GPU
  // form of NN::Battle::PokemonCache (nn/battle/cache.h:18-131).  `embedding` (n x in_dim floats) and `slot_tags`
  // (n x 10 x OAKGPU_SLOT_TAG_WORDS u32, initialised to 0xFF bytes) are the caller's, kept between calls.
  void value_inference_cached_dev(const uint8_t *battles, const uint8_t *durations, uint32_t n, float *values, float *embedding,
                                  uint32_t *slot_tags) {
    check(oakgpu_leaf_eval_cached_dev(ctx_.get(), net_, battles, durations, n, values, embedding, slot_tags));
  }

  oakgpu_net *get() const noexcept { return net_; }

private:
  Context &ctx_;
  oakgpu_net *net_{};
};

// RuntimeSearch::Heap (util/search.h:17-32, search.cc:17-58): the tree of a search kept between searches.
class Heap {
public:
  Heap() { check(oakgpu_heap_create(&h_)); }
  ~Heap() { oakgpu_heap_destroy(h_); }
  Heap(const Heap &) = delete;
  Heap &operator=(const Heap &) = delete;
  bool empty() const noexcept { return oakgpu_heap_empty(h_) != 0; }
  // after the joint action (p1 index i, p2 index j) was played and produced `obs` (the 16-byte chance actions): the child
  // becomes the root (true), or nothing could be kept (false) -- search.cc:27-52
  bool update(uint8_t i, uint8_t j, const uint8_t *obs16) { return oakgpu_heap_update(h_, i, j, obs16) != 0; }
  uint64_t nodes() const noexcept { return oakgpu_heap_nodes(h_); }
  oakgpu_heap *get() const noexcept { return h_; }

private:
  oakgpu_heap *h_{};
};

// MCTS::Search::run(device, budget, params, heap, eval, input, output = {}) (mcts.h:154-155): `params.bandit` picks the
// joint bandit, eval = Monte-Carlo (net == nullptr) or the network; heap == nullptr is a fresh Node tree for this call,
// `previous` (nullable) is the Output to add to.  The result carries MCTS::Output's root matrices (mcts.h:68-90).
class TreeSearch {
public:
  explicit TreeSearch(Context &ctx) : ctx_{ctx} {}
  // Start from these, not from `oakgpu_search_params{}`: roll clamping {3, 1} (mcts.h:131) and the DEFAULT Exp3 mixing of an
  // absent agent-string field -- a zeroed struct asks for alpha = 0 (no uniform mixing), which the reference only does on request.
  static oakgpu_search_params default_params() { return oakgpu_search_params OAKGPU_SEARCH_PARAMS_INIT; }
  oakgpu_search_output run(const Leaf &input, oakgpu_search_params params, Network *net = nullptr, Heap *heap = nullptr,
                           const oakgpu_search_output *previous = nullptr) {
    params.eval = net ? 1 : 0;
    oakgpu_search_output out{};
    check(oakgpu_search_heap(ctx_.get(), net ? net->get() : nullptr, heap ? heap->get() : nullptr, input.battle, input.durations, input.result,
                             &params, previous, &out));
    return out;
  }

private:
  Context &ctx_;
};

// RuntimeSearch::Agent (util/search.h:34-64) and RuntimeSearch::run: budget "4096" | "100ms" | "8s", bandit "ucb-1.0" |
// "exp3-0.1-0.05" | ..., eval "mc" | "fp" | <.battle.net path>, matrix_ucb "" | "delay-interval-minimum-c".  Unparsable
// strings throw std::runtime_error with the reference's texts (search.cc:200-307).
struct Agent {
  std::string budget{"4096"}, bandit{"ucb-1.0"}, eval{"mc"}, matrix_ucb{};
  bool discrete{false}, table{false};
};
inline oakgpu_search_output run(Context &ctx, const Leaf &input, const Agent &agent, uint64_t seed, uint32_t batch = 0, Heap *heap = nullptr,
                                const oakgpu_search_output *previous = nullptr) {
  const oakgpu_agent a{agent.budget.c_str(), agent.bandit.c_str(), agent.eval.c_str(), agent.matrix_ucb.c_str(), agent.discrete, agent.table};
  oakgpu_search_output out{};
  check(oakgpu_search_agent_heap(ctx.get(), heap ? heap->get() : nullptr, input.battle, input.durations, input.result, &a, batch, seed, previous, &out));
  return out;
}

// Exact Nash equilibrium of an integer m x n (<= 9 x 9) matrix game, row player maximising: {p1, p2, value / discretize}.
inline std::tuple<std::vector<double>, std::vector<double>, double> solve_matrix(const std::vector<int32_t> &payoffs, int m, int n,
                                                                                 int discretize_factor = 256) {
  if (m < 1 || n < 1 || payoffs.size() != static_cast<size_t>(m) * static_cast<size_t>(n)) throw std::runtime_error{"oakgpu: payoff matrix shape"};
  std::vector<double> p1(static_cast<size_t>(m)), p2(static_cast<size_t>(n));
  double value = 0;
  check(oakgpu_solve_matrix(payoffs.data(), m, n, discretize_factor, p1.data(), p2.data(), &value));
  return {p1, p2, value};
}

// benchmark.cc:23-31 + mcts.h:250-263,448-496 on the device: `n` playouts from one root, every draw taken from ONE
// std::mt19937{seed} in the reference's sequential order (playout i starts where playout i - 1 stopped).
class SharedDeviceRollout {
public:
  explicit SharedDeviceRollout(Context &ctx) : ctx_{ctx} {}
  RolloutResult run(const Leaf &root, uint32_t seed, uint32_t n, uint32_t max_steps = 100000, std::vector<uint8_t> *final_battles = nullptr) {
    RolloutResult out{std::vector<float>(n), std::vector<uint32_t>(n), std::vector<uint8_t>(n)};
    if (final_battles) final_battles->assign(size_t{n} * OAKGPU_BATTLE_SIZE, 0);
    for (size_t draws = size_t{n} * 160 + 4096;; draws *= 2) { // grow the generator's output until n playouts fit
      if (draws > (size_t{1} << 30)) throw std::runtime_error{"oakgpu: the shared generator's stream would exceed 2^30 draws"}; // (before the cast below)
      std::vector<uint64_t> stream(draws);
      check(oakgpu_mt19937_fill(seed, 0, stream.data(), stream.size()));
      uint64_t used = 0;
      const int rc = oakgpu_rollout_shared_device(ctx_.get(), root.battle, root.durations, root.result, stream.data(), static_cast<uint32_t>(draws), n,
                                                  max_steps, 1, out.result.data(), out.steps.data(), out.value.data(),
                                                  final_battles ? final_battles->data() : nullptr, nullptr, nullptr, &used);
      if (rc == 0) return out;
      if (draws > (size_t{1} << 30) || std::string{oakgpu_last_error()}.find("too short") == std::string::npos) check(rc);
    }
  }

private:
  Context &ctx_;
};

// One game's `.battle.data` record under construction (CompressedFrames: battle after the opening update, one Update per
// turn, the final result byte); bytes() serialises it in the reference's on-disk layout.
class Frames {
public:
  explicit Frames(const uint8_t *first_battle) { std::memcpy(battle_, first_battle, OAKGPU_BATTLE_SIZE); }
  void push(const oakgpu_search_output &o, uint8_t c1, uint8_t c2) { // Update{search_output, c1, c2} (compressed-frame.h:66-75)
    oakgpu_frame_update u{};
    u.m = o.m; u.n = o.n; u.c1 = c1; u.c2 = c2;
    u.iterations = static_cast<uint32_t>(o.iterations);
    u.empirical_value = o.empirical_value;
    u.nash_value = o.nash_value;
    for (int k = 0; k < 9; ++k) { u.p1_empirical[k] = o.p1_empirical[k]; u.p1_nash[k] = o.p1_nash[k]; u.p2_empirical[k] = o.p2_empirical[k]; u.p2_nash[k] = o.p2_nash[k]; }
    updates_.push_back(u);
  }
  std::vector<uint8_t> bytes(uint8_t result) const {
    std::vector<uint8_t> out(oakgpu_frames_size(updates_.data(), static_cast<uint32_t>(updates_.size())));
    size_t written = 0;
    check(oakgpu_frames_write(battle_, result, updates_.data(), static_cast<uint32_t>(updates_.size()), out.data(), out.size(), &written));
    out.resize(written);
    return out;
  }

private:
  uint8_t battle_[OAKGPU_BATTLE_SIZE];
  std::vector<oakgpu_frame_update> updates_;
};

// The path's one exchange step for root-parallel search sharded over the GPUs of a node (one process per GPU): reduce the
// rank's leaf values to one mean per root on the device, then ONE ncclAllGather (RCCL over xGMI) of those means.
class Exchange {
public:
  // id128: made by rank 0 with Exchange::unique_id() and handed to the other ranks out of band
  Exchange(Context &ctx, const uint8_t *id128, int rank, int world) : ctx_{ctx} { check(oakgpu_comm_create(ctx_.get(), id128, rank, world, &comm_)); }
  ~Exchange() { oakgpu_comm_destroy(comm_); }
  Exchange(const Exchange &) = delete;
  Exchange &operator=(const Exchange &) = delete;
  static std::vector<uint8_t> unique_id() {
    std::vector<uint8_t> id(128);
    check(oakgpu_comm_unique_id(id.data()));
    return id;
  }
  // device pointers, asynchronous on the context's stream: means[r] = mean(values[r * per_root ...]); all = world x roots
  void root_means_all_gather_dev(const float *values, uint32_t roots, uint32_t per_root, float *means, float *all) {
    check(oakgpu_segment_mean_dev(ctx_.get(), values, roots, per_root, means));
    check(oakgpu_all_gather_dev(ctx_.get(), comm_, means, all, roots));
  }

  // the sliced search steps' exchange: `count` 64-bit aggregates (count | 2 x value sum << 32, OakGPU::RootSteps) per rank
  void aggregates_all_gather_dev(const unsigned long long *mine, unsigned long long *all, size_t count) {
    check(oakgpu_all_gather_dev(ctx_.get(), comm_, reinterpret_cast<const float *>(mine), reinterpret_cast<float *>(all), 2 * count));
  }

private:
  Context &ctx_;
  oakgpu_comm *comm_{};
};

// BASELINE configs[3] as search steps that do not wait for their longest playout (oakgpu_root_steps_*, include/oakgpu.h): every launch
// advances each playout in flight by at most `slice` turn-steps; a playout of len turn-steps started in step k is credited to step
// k + (len - 1) / slice of its root and travels between launches on a carry list.  The reference's analogue: the generator's workers never
// wait for each other (cpp/src/generate.cc:527-536); per-playout prep = mcts.h:250-263, the playout = mcts.h:448-496.
class RootSteps {
public:
  RootSteps(Context &ctx, uint32_t roots, uint32_t replicas, uint32_t slice = 16, uint32_t max_steps = 1000) : roots_{roots} {
    check(oakgpu_root_steps_create(ctx.get(), roots, replicas, slice, max_steps, &rs_));
  }
  ~RootSteps() { oakgpu_root_steps_destroy(rs_); }
  RootSteps(const RootSteps &) = delete;
  RootSteps &operator=(const RootSteps &) = delete;
  // device pointers, asynchronous on the context's stream.  report: roots + 2 u64 -- [r] count | (2 x value sum) << 32 of the playouts
  // credited to this step, [roots] turn-steps executed, [roots + 1] carried playouts | error word << 32.  fresh = false: a drain step.
  void launch_dev(const uint8_t *root_battles, const uint8_t *root_durations, const uint8_t *root_results, uint8_t *lane_prng,
                  unsigned long long *report, bool fresh = true) {
    check(oakgpu_root_steps_launch_dev(rs_, root_battles, root_durations, root_results, lane_prng, fresh ? 1 : 0, report));
  }
  static uint32_t credited(unsigned long long a) { return static_cast<uint32_t>(a); }
  static double mean_value(unsigned long long a) { return credited(a) ? static_cast<double>(a >> 32) / (2.0 * credited(a)) : 0.5; }
  uint32_t roots() const { return roots_; }

private:
  oakgpu_root_steps *rs_{};
  uint32_t roots_;
};

} // namespace OakGPU
