/* include/pkmn.h -- the libpkmn gen-1 C ABI subset Oak uses, served by liboakgpu.so.
 *
 * The reference includes a generated <pkmn.h> from the lab-oak/engine Zig build (absent from the
 * checkout, .gitignore:1).  This header declares exactly the names / shapes Oak's code uses
 * (inferred from every call site: cpp/include/search/mcts.h:161-166,260,278,337-350,453-479,571,588,617;
 * cpp/include/libpkmn/pkmn.h:67-156,214-233; cpp/include/libpkmn/data.h:315-330), so Oak's headers
 * compile against it unchanged.  Every function is a batch of ONE through the same HIP kernels as the
 * batched oakgpu_* entry points (no CPU implementation exists in this library): correct, but it costs a
 * kernel launch + two small copies per call -- tree descent should keep its CPU libpkmn, and use the
 * batched ABI (include/oakgpu.h) for playouts.  First use creates a process-wide context on device
 * OAKGPU_DEVICE (default 0); on failure update() returns PKMN_RESULT_ERROR and choices() returns 0.
 */
#ifndef OAKGPU_PKMN_H
#define OAKGPU_PKMN_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PKMN_GEN1_BATTLE_SIZE 384
#define PKMN_GEN1_MAX_CHOICES 9
#define PKMN_GEN1_CHANCE_ACTIONS_SIZE 16
#define PKMN_GEN1_CHANCE_DURATIONS_SIZE 8

typedef uint8_t pkmn_choice;
typedef uint8_t pkmn_result;
typedef enum { PKMN_PLAYER_P1 = 0, PKMN_PLAYER_P2 = 1 } pkmn_player;
typedef enum { PKMN_CHOICE_PASS = 0, PKMN_CHOICE_MOVE = 1, PKMN_CHOICE_SWITCH = 2 } pkmn_choice_kind;
typedef enum { PKMN_RESULT_NONE = 0, PKMN_RESULT_WIN = 1, PKMN_RESULT_LOSE = 2, PKMN_RESULT_TIE = 3, PKMN_RESULT_ERROR = 4 } pkmn_result_kind;

typedef struct { uint8_t bytes[PKMN_GEN1_BATTLE_SIZE]; } pkmn_gen1_battle;
typedef struct { uint8_t bytes[PKMN_GEN1_CHANCE_DURATIONS_SIZE]; } pkmn_gen1_chance_durations;
typedef struct { uint8_t bytes[PKMN_GEN1_CHANCE_ACTIONS_SIZE]; } pkmn_gen1_chance_actions;
typedef struct { pkmn_gen1_chance_durations durations; pkmn_gen1_chance_actions actions; } pkmn_gen1_chance_options;
typedef struct { pkmn_gen1_chance_actions overrides; } pkmn_gen1_calc_options;
typedef struct { uint8_t *buf; size_t len; } pkmn_gen1_log_options;
/* value-initialisable with {} (pkmn.h:71) */
typedef struct {
  pkmn_gen1_chance_actions actions;
  pkmn_gen1_chance_durations durations;
  pkmn_gen1_chance_actions overrides;
  uint8_t has_overrides;
} pkmn_gen1_battle_options;

pkmn_result pkmn_gen1_battle_update(pkmn_gen1_battle *battle, pkmn_choice c1, pkmn_choice c2, pkmn_gen1_battle_options *options);
uint8_t pkmn_gen1_battle_choices(const pkmn_gen1_battle *battle, pkmn_player player, pkmn_choice_kind request,
                                 pkmn_choice out[], size_t len);
/* NULL chance: keep the tracked durations, reset actions; NULL calc: no damage-roll overrides (pkmn.h:88-104) */
void pkmn_gen1_battle_options_set(pkmn_gen1_battle_options *options, const pkmn_gen1_log_options *log,
                                  const pkmn_gen1_chance_options *chance, const pkmn_gen1_calc_options *calc);
pkmn_gen1_chance_actions *pkmn_gen1_battle_options_chance_actions(const pkmn_gen1_battle_options *options);
pkmn_gen1_chance_durations *pkmn_gen1_battle_options_chance_durations(const pkmn_gen1_battle_options *options);
pkmn_result_kind pkmn_result_type(pkmn_result result);
pkmn_choice_kind pkmn_result_p1(pkmn_result result);
pkmn_choice_kind pkmn_result_p2(pkmn_result result);

#ifdef __cplusplus
}
#endif
#endif
