// oak_amd/csrc/pkmn_shim.hip -- the libpkmn-named single-battle C ABI (include/pkmn.h) as batches of one
// through the HIP kernels behind oakgpu_update / oakgpu_choices.  No CPU implementation.
#include <stdlib.h>
#include <string.h>

#include "../../include/oakgpu.h"
#include "../../include/pkmn.h"

namespace {
// One context (= one HIP stream + its own staging buffers) PER CALLING THREAD: libpkmn is re-entrant and thread-safe per
// battle, and the reference runs one game per std::thread (generate.cc:527-536, vs.cc:534-542) -- behind one
// process-wide context and mutex (rounds 1-2) those threads stepped one battle at a time.  A thread's context lives until the
// thread ends.
struct ThreadCtx {
  oakgpu_ctx *ctx = nullptr;
  bool tried = false;
  ~ThreadCtx() { if (ctx) oakgpu_destroy(ctx); }
};
oakgpu_ctx *shared_ctx() {
  static thread_local ThreadCtx t;
  if (!t.tried) {
    t.tried = true;
    int dev = 0;
    if (const char *e = getenv("OAKGPU_DEVICE")) dev = atoi(e);
    if (oakgpu_create(&t.ctx, dev) != 0) t.ctx = nullptr;
  }
  return t.ctx;
}
} // namespace

extern "C" {

pkmn_result pkmn_gen1_battle_update(pkmn_gen1_battle *battle, pkmn_choice c1, pkmn_choice c2, pkmn_gen1_battle_options *o) {
  oakgpu_ctx *ctx = shared_ctx();
  if (!ctx || !battle || !o) return PKMN_RESULT_ERROR;
  uint8_t res = PKMN_RESULT_ERROR;
  const int rc = oakgpu_update(ctx, battle->bytes, &c1, &c2, o->durations.bytes, o->actions.bytes,
                               o->has_overrides ? o->overrides.bytes : nullptr, 1, &res);
  return rc == 0 ? res : (pkmn_result)PKMN_RESULT_ERROR;
}

uint8_t pkmn_gen1_battle_choices(const pkmn_gen1_battle *battle, pkmn_player player, pkmn_choice_kind request, pkmn_choice out[],
                                 size_t len) {
  oakgpu_ctx *ctx = shared_ctx();
  if (!ctx || !battle || !out || len < PKMN_GEN1_MAX_CHOICES) return 0;
  const uint8_t result = (uint8_t)(player == PKMN_PLAYER_P1 ? (request << 4) : (request << 6));
  uint8_t buf[OAKGPU_MAX_CHOICES], n = 0;
  if (oakgpu_choices(ctx, battle->bytes, &result, (int)player, buf, &n, 1) != 0) return 0;
  memcpy(out, buf, n);
  return n;
}

void pkmn_gen1_battle_options_set(pkmn_gen1_battle_options *o, const pkmn_gen1_log_options *log, const pkmn_gen1_chance_options *chance,
                                  const pkmn_gen1_calc_options *calc) {
  (void)log; // protocol logging is out of scope
  memset(o->actions.bytes, 0, sizeof o->actions.bytes);
  if (chance) o->durations = chance->durations;
  if (calc) { o->overrides = calc->overrides; o->has_overrides = 1; }
  else { memset(o->overrides.bytes, 0, sizeof o->overrides.bytes); o->has_overrides = 0; }
}

pkmn_gen1_chance_actions *pkmn_gen1_battle_options_chance_actions(const pkmn_gen1_battle_options *o) {
  return const_cast<pkmn_gen1_chance_actions *>(&o->actions);
}
pkmn_gen1_chance_durations *pkmn_gen1_battle_options_chance_durations(const pkmn_gen1_battle_options *o) {
  return const_cast<pkmn_gen1_chance_durations *>(&o->durations);
}
pkmn_result_kind pkmn_result_type(pkmn_result r) { return (pkmn_result_kind)(r & 15); }
pkmn_choice_kind pkmn_result_p1(pkmn_result r) { return (pkmn_choice_kind)((r >> 4) & 3); }
pkmn_choice_kind pkmn_result_p2(pkmn_result r) { return (pkmn_choice_kind)((r >> 6) & 3); }

} // extern "C"
