// include/oakgpu.hpp -- thin C++ host layer over the C ABI (include/oakgpu.h), shaped like the
// reference's search-node surface so Oak's tree code can use it as an `eval` (cpp/include/search/mcts.h:25-57).
//
//   OakGPU::Context            <- per-thread owner of device state (reference: per-thread Agent/Heap, util/search.h:17-64)
//   OakGPU::Network            <- NN::Battle::Network (nn/battle/network.h:22-176): shape(), value_inference(batch)
//   OakGPU::BatchedMonteCarlo  <- MCTS::MonteCarlo (mcts.h:21-23) + init_stats_and_rollout (mcts.h:448-496), batched
//   OakGPU::TreeSearch         <- MCTS::Search::run (mcts.h:154-247): Node heap + joint UCB / PUCB, leaves batched on the GPU
// Errors surface as std::runtime_error, like the reference's loaders (cpp/src/search.cc:81-146).
#pragma once
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <tuple>
#include <vector>

#include "oakgpu.h"

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

  oakgpu_net *get() const noexcept { return net_; }

private:
  Context &ctx_;
  oakgpu_net *net_{};
};

// MCTS::Search::run(device, budget, params, heap, eval, input) (mcts.h:154-155) with an integer budget: the heap is a
// fresh Node tree per call, `params.bandit` picks UCB::Bandit / PUCB::Bandit, eval = Monte-Carlo (net == nullptr)
// or the network.  Output carries MCTS::Output's root matrices (mcts.h:68-90).
class TreeSearch {
public:
  explicit TreeSearch(Context &ctx) : ctx_{ctx} {}
  oakgpu_search_output run(const Leaf &input, oakgpu_search_params params, Network *net = nullptr) {
    params.eval = net ? 1 : 0;
    oakgpu_search_output out{};
    check(oakgpu_search(ctx_.get(), net ? net->get() : nullptr, input.battle, input.durations, input.result, &params, &out));
    return out;
  }

private:
  Context &ctx_;
};

} // namespace OakGPU
