// Compile-and-link check of include/oakgpu.hpp + include/pkmn.h against liboakgpu.so (tests/test_abi.py).
// When run on a machine with a GPU it also exercises one rollout through the C++ layer.
#include <cstdio>
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
  std::puts("ok");
  return 0;
}
