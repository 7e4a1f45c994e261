// Compile-and-link check of include/oakgpu.hpp + include/pkmn.h against liboakgpu.so (tests/test_abi.py).
// When run on a machine with a GPU it also exercises one rollout through the C++ layer.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <oakgpu.hpp>
#include <pkmn.h>

int main() {
  static_assert(sizeof(pkmn_gen1_battle) == 384 && sizeof(pkmn_gen1_chance_durations) == 8 && sizeof(pkmn_gen1_chance_actions) == 16);
  { // host-only pieces of the layer: the exact Nash solver and the .battle.data record writer
    auto [p1, p2, v] = OakGPU::solve_matrix({256, 0, 0, 256}, 2, 2);   // matching pennies x 256
    if (p1[0] != 0.5 || p2[1] != 0.5 || v != 0.5) { std::puts("solve_matrix wrong"); return 1; }
    uint8_t battle[384] = {};
    OakGPU::Frames frames{battle};
    oakgpu_search_output o{};
    o.m = 2; o.n = 1; o.iterations = 7; o.empirical_value = 0.25; o.nash_value = 1.0;
    o.p1_empirical[0] = 1.0; o.p1_nash[1] = 1.0; o.p2_empirical[0] = 1.0; o.p2_nash[0] = 1.0;
    frames.push(o, 5, 9);
    const auto rec = frames.bytes(2);
    if (rec.size() != 391 + 1 + 2 + 4 + 4 + 4 * 3) { std::puts("frame record size wrong"); return 1; }
    uint32_t count = 0;
    if (oakgpu_frames_read(rec.data(), rec.size(), nullptr, nullptr, nullptr, 0, &count, nullptr) != 0 || count != 1) { std::puts("frame record unreadable"); return 1; }
  }
  if (oakgpu_device_count() == 0) { std::puts("no gpu: link check only"); return 0; }
  OakGPU::Context ctx{0};
  OakGPU::BatchedMonteCarlo mc{ctx};
  std::vector<OakGPU::Leaf> leaves(4);          // all-zero battles: both sides empty -> immediate tie
  for (auto &l : leaves) { std::memset(&l, 0, sizeof l); l.result = 0x50; }
  std::vector<uint64_t> rng(4, 0x1234567ull);
  auto out = mc.rollout(leaves, rng);
  for (float v : out.value) if (v != 0.5f) { std::puts("unexpected value"); return 1; }
  OakGPU::TreeSearch search{ctx};                 // a terminal root is refused, loudly (std::runtime_error)
  bool threw = false;
  leaves[0].result = 0x03;                        // PKMN_RESULT_TIE
  try { (void)search.run(leaves[0], oakgpu_search_params{64, 64, 2.0f, 0, 0, 0, 3, 1, 1}); } catch (const std::runtime_error &) { threw = true; }
  if (!threw) { std::puts("terminal root accepted"); return 1; }
  threw = false;                                  // RuntimeSearch-style agent strings: the reference's error texts
  OakGPU::Agent agent;
  agent.bandit = "thompson-1.0";
  leaves[0].result = 0x50;
  try { (void)OakGPU::run(ctx, leaves[0], agent, 1); } catch (const std::runtime_error &e) { threw = std::string{e.what()}.find("Could not parse bandit string") != std::string::npos; }
  if (!threw) { std::puts("bad bandit string accepted"); return 1; }
  // ---- the reference's per-leaf eval signatures (nn/battle/network.h:72-79,102-123; called by mcts.h:196-209,401-422) through
  // the C++ layer, against the batched calls on the same leaves.  Battles: two level-100 teams, a few random turns apart.
  if (const char *net_path = std::getenv("OAKGPU_SMOKE_NET")) {
    const int N = 8;
    uint8_t teams[N][60];
    uint64_t seeds[N];
    static const uint8_t sets[12][5] = {{124, 59, 142, 94, 156}, {65, 94, 86, 105, 69}, {103, 79, 94, 153, 95}, {143, 34, 156, 89, 63}, {128, 34, 89, 63, 126}, {121, 59, 94, 85, 105},
                                        {113, 135, 86, 58, 85}, {94, 95, 101, 85, 153}, {112, 89, 157, 34, 63}, {145, 65, 85, 86, 97}, {80, 133, 94, 57, 156}, {91, 59, 153, 128, 62}};
    for (int i = 0; i < N; ++i) { std::memcpy(teams[i], sets, 60); seeds[i] = 0x9E3779B97F4A7C15ull * (i + 1); }
    std::vector<pkmn_gen1_battle> battles(N);
    std::vector<pkmn_gen1_chance_durations> durations(N);
    std::vector<uint8_t> results(N);
    OakGPU::check(oakgpu_init_battles(ctx.get(), &teams[0][0], seeds, N, 1, battles[0].bytes, durations[0].bytes, results.data()));
    std::vector<OakGPU::Leaf> pos(N);
    std::vector<uint8_t> c1(N * 9), c2(N * 9), n1(N), n2(N);
    for (int turn = 0; turn < 6; ++turn) { // a few turns through the libpkmn-named single-battle ABI, first legal choice + i mod count
      for (int i = 0; i < N; ++i) {
        pkmn_choice o1[9], o2[9];
        const uint8_t k1 = pkmn_gen1_battle_choices(&battles[i], PKMN_PLAYER_P1, pkmn_result_p1(results[i]), o1, 9);
        const uint8_t k2 = pkmn_gen1_battle_choices(&battles[i], PKMN_PLAYER_P2, pkmn_result_p2(results[i]), o2, 9);
        if (pkmn_result_type(results[i]) != PKMN_RESULT_NONE || !k1 || !k2) continue;
        pkmn_gen1_battle_options opt{};
        pkmn_gen1_chance_options chance{};
        chance.durations = durations[i];
        pkmn_gen1_battle_options_set(&opt, nullptr, &chance, nullptr);
        results[i] = pkmn_gen1_battle_update(&battles[i], o1[(i + turn) % k1], o2[(2 * i + turn) % k2], &opt);
        durations[i] = *pkmn_gen1_battle_options_chance_durations(&opt);
      }
    }
    for (int i = 0; i < N; ++i) {
      std::memcpy(pos[i].battle, battles[i].bytes, 384);
      std::memcpy(pos[i].durations, durations[i].bytes, 8);
      pos[i].result = results[i];
      pkmn_choice o1[9] = {}, o2[9] = {};
      n1[i] = pkmn_gen1_battle_choices(&battles[i], PKMN_PLAYER_P1, pkmn_result_p1(results[i]), o1, 9);
      n2[i] = pkmn_gen1_battle_choices(&battles[i], PKMN_PLAYER_P2, pkmn_result_p2(results[i]), o2, 9);
      std::memcpy(&c1[i * 9], o1, 9);
      std::memcpy(&c2[i * 9], o2, 9);
    }
    OakGPU::Network net{ctx, net_path};
    const std::vector<float> batched = net.value_inference(pos);
    std::vector<float> l1, l2;
    const std::vector<float> batched_vp = net.value_policy_inference(pos, c1, n1, c2, n2, l1, l2);
    for (int i = 0; i < N; ++i) {
      const float v = net.value_inference(battles[i], durations[i]);               // network.h:72-79
      float p1[9] = {}, p2[9] = {};
      const float vp = net.value_policy_inference(battles[i], durations[i], n1[i], n2[i], &c1[i * 9], &c2[i * 9], p1, p2); // network.h:102-123
      if (!(v > 0.0f && v < 1.0f) || v != batched[i] || vp != batched_vp[i] || std::fabs(v - vp) > 2e-6f) { std::printf("per-leaf value differs from the batched call (leaf %d: %g %g %g %g)\n", i, v, batched[i], vp, batched_vp[i]); return 1; }
      for (int k = 0; k < 9; ++k)
        if ((k < n1[i] ? p1[k] != l1[i * 9 + k] : p1[k] != 0.0f) || (k < n2[i] ? p2[k] != l2[i * 9 + k] : p2[k] != 0.0f)) { std::printf("per-leaf logits differ from the batched call (leaf %d, choice %d)\n", i, k); return 1; }
    }
    std::puts("per-leaf eval == batched eval");
  }
  std::puts("ok");
  return 0;
}
