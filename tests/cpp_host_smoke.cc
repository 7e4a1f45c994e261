// Compile-and-link check of include/oakgpu.hpp + include/pkmn.h against liboakgpu.so (tests/test_abi.py).
// When run on a machine with a GPU it also exercises one rollout through the C++ layer.
#include <cstdio>
#include <oakgpu.hpp>
#include <pkmn.h>

int main() {
  static_assert(sizeof(pkmn_gen1_battle) == 384 && sizeof(pkmn_gen1_chance_durations) == 8 && sizeof(pkmn_gen1_chance_actions) == 16);
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
  std::puts("ok");
  return 0;
}
