// oak_amd/csrc/gen1_device.hpp -- device-side gen-1 (RBY) turn resolution for gfx950.
//
// Replaces, inside the batched rollout, the libpkmn calls the reference makes once per
// turn-step (cpp/include/search/mcts.h:453-479): pkmn_gen1_battle_choices x2 and
// pkmn_gen1_battle_update, plus the chance-durations tracking Oak's encoder reads
// (cpp/include/libpkmn/data.h:270-311).  Build configuration mirrored:
// /root/reference/dev/libpkmn:9 (showdown, miss=false, advance=false, ebc=false, key=true,
// chance, calc).
//
// Layout: ONE LANE PER PLAYOUT.  A lane's 384-byte battle (byte layout identical to
// cpp/include/libpkmn/layout.h so AoS buffers round-trip unchanged) lives in LDS as 96
// dwords, lane-interleaved: dword w of lane t is lds[w * STRIDE + t].  Any per-lane
// (divergent) byte offset therefore hits bank (t mod 32): conflict-free by construction,
// and byte / halfword fields are read with native ds_read_u8 / ds_read_u16.
// The move / species / type-chart / stage tables are staged once per workgroup into LDS
// (struct Tables) from the packed images in gen1_tables.inc.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace oak {

#include "gen1_tables.inc"

// Roll-order choices that pkmn/engine's -Dshowdown path settles by following Pokemon Showdown's own code (the reference builds
// libpkmn with -Dshowdown: /root/reference/dev/libpkmn:9).  Same names and defaults in the CPU checker's restatement of the engine (the tests hold the three to each other); 0 = rounds 1-3:
//   OAK_MULTIHIT_ROLL_FIRST  multi-hit count rolled behind the accuracy check, before crit / damage (gen-1 tryMoveHit)
//   OAK_PSYWAVE_SHOWDOWN     Psywave = random(0, level * 3 / 2), a 0 fails (Desync Clause Mod); the action holds the roll + 1
#ifndef OAK_MULTIHIT_ROLL_FIRST
#define OAK_MULTIHIT_ROLL_FIRST 1
#endif
#ifndef OAK_PSYWAVE_SHOWDOWN
#define OAK_PSYWAVE_SHOWDOWN 1
#endif
//   OAK_COUNTER_SHOWDOWN     (round 5, default 0) Counter hits iff the target side's last USED and last SELECTED move are both counterable
//                            (bp > 0, Normal / Fighting, not Counter) and last_damage > 0 -- Showdown's gen-1 damageCallback; 0: the
//                            `counterable` byte of last_moves[]
//   OAK_ACCURACY_LAST        (round 5, default 0) cartridge roll order of ordinary damaging moves: crit, damage roll, THEN accuracy
#ifndef OAK_COUNTER_SHOWDOWN
#define OAK_COUNTER_SHOWDOWN 0
#endif
#ifndef OAK_ACCURACY_LAST
#define OAK_ACCURACY_LAST 0
#endif
#if OAK_ACCURACY_LAST && OAK_MULTIHIT_ROLL_FIRST
#error "OAK_ACCURACY_LAST needs -DOAK_MULTIHIT_ROLL_FIRST=0 (the count is rolled behind the accuracy check)"
#endif

// ---- result / choice encodings (cpp/include/libpkmn/pkmn.h:108-133,214-233) ------------
enum : uint32_t { R_NONE = 0, R_WIN = 1, R_LOSE = 2, R_TIE = 3, R_ERROR = 4 };
enum : uint32_t { C_PASS = 0, C_MOVE = 1, C_SWITCH = 2 };
__device__ __forceinline__ uint32_t mk_result(uint32_t t, uint32_t p1, uint32_t p2) { return t | (p1 << 4) | (p2 << 6); }

// ---- byte offsets (cpp/include/libpkmn/layout.h:15-50) ---------------------------------
enum : int {
  SIDE_SZ = 184, PK_SZ = 24,
  O_ACTIVE = 144, O_ORDER = 176, O_LAST_SEL = 182, O_LAST_USED = 183,
  P_HP_MAX = 0, P_ATK = 2, P_DEF = 4, P_SPE = 6, P_SPC = 8, P_MOVES = 10, P_HP = 18, P_STATUS = 20,
  P_SPECIES = 21, P_TYPES = 22, P_LEVEL = 23,
  A_STATS = 0, A_SPECIES = 10, A_TYPES = 11, A_BOOSTS = 12, A_VOL = 16, A_MOVES = 24,
  B_TURN = 368, B_LAST_DAMAGE = 370, B_LAST_MOVES = 372, B_RNG = 376,
};

// volatiles (layout.h:69-96), split over the two dwords at A_VOL / A_VOL+4
enum : uint32_t {
  V_BIDE = 1u << 0, V_THRASHING = 1u << 1, V_MULTIHIT = 1u << 2, V_FLINCH = 1u << 3, V_CHARGING = 1u << 4,
  V_BINDING = 1u << 5, V_INVULNERABLE = 1u << 6, V_CONFUSION = 1u << 7, V_MIST = 1u << 8,
  V_FOCUSENERGY = 1u << 9, V_SUBSTITUTE = 1u << 10, V_RECHARGING = 1u << 11, V_RAGE = 1u << 12,
  V_LEECHSEED = 1u << 13, V_TOXIC = 1u << 14, V_LIGHTSCREEN = 1u << 15, V_REFLECT = 1u << 16,
  V_TRANSFORM = 1u << 17,
};
enum : uint32_t { ST_SLP = 7, ST_PSN = 0x08, ST_BRN = 0x10, ST_FRZ = 0x20, ST_PAR = 0x40, ST_EXT = 0x80, ST_TOX = 0x88 };

// chance-action bit offsets (layout.h:98-117)
enum : int { AC_DAMAGE = 0, AC_HIT = 8, AC_CRIT = 10, AC_SECONDARY = 12, AC_SPEEDTIE = 14, AC_CONFUSED = 16,
             AC_PARALYZED = 18, AC_SLEEP = 24, AC_CONFUSION = 26, AC_DISABLE = 29, AC_ATTACKING = 31,
             AC_BINDING = 33, AC_MOVESLOT = 36, AC_MULTIHIT = 44, AC_PSYWAVE = 48, AC_METRONOME = 56 };
enum : uint32_t { OBS_STARTED = 1, OBS_CONTINUING = 2, OBS_ENDED = 3 };

// LDS-address-space pointer types: keeping the address space in the type makes hipcc emit
// ds_read/ds_write (lgkmcnt only) instead of flat_load/flat_store for every state access.
#define OAK_LDS __attribute__((address_space(3)))
typedef OAK_LDS uint32_t lds_u32;
typedef OAK_LDS uint16_t lds_u16;
typedef OAK_LDS uint8_t lds_u8;

struct Tables {
  const lds_u32 *mv;     // [166] effect | bp<<8 | type<<16 | acc<<24
  const lds_u8 *maxpp;   // [166]
  const lds_u32 *sp0;    // [152]
  const lds_u32 *sp1;    // [152]
  const lds_u8 *chart;   // [225]
  const lds_u16 *boost;  // [13] num | den<<8
  const lds_u32 *rcp;    // [256] floor((2^32-1)/d) + 1: exact x / d by one mul-hi for x < 2^24, d in 2..255
  const lds_u16 *fx;     // [68] per-effect descriptor (fx_desc below): gate / action class / secondary effect
};
static constexpr int TABLE_LDS_BYTES = 166 * 4 + 168 + 152 * 4 + 152 * 4 + 228 + 28 + 256 * 4 + 68 * 2;

// ---- per-effect descriptors: what gen1_regs.hpp's run_move needs to know about a move effect, as data.
// Lanes of a wave run different moves; a `switch (effect)` costs the wave every case any lane takes plus the
// exec-mask bookkeeping of all of them, a table lookup costs one LDS read and a few selects for everybody.
//   status moves (bp == 0):  bits 0-3 gate (can the move fail before / instead of its accuracy check),
//                            bits 4-6 action class, bits 7-11 its parameter
//   damaging moves:          bits 0-2 secondary-effect kind, 3-5 chance index, 6-8 parameter
enum : uint32_t { G_NONE = 0, G_SUB, G_INVUL, G_GRASS, G_HIT, G_PAR, G_PSN, G_TELE, G_SLEEP, G_DISABLE };
enum : uint32_t { A_NONE = 0, A_BOOST, A_UNBOOST, A_SVOL, A_FVOL, A_PAR, A_CONF, A_HEAVY };
enum : uint32_t { SEC_NONE = 0, SEC_STATUS = 1, SEC_FLINCH = 2, SEC_CONF = 3, SEC_UNBOOST = 4 };
// chance index -> x/256: 0: 26 (10%), 1: 77 (30%), 2: 52 (20%), 3: 103 (40%), 4: 25 (confusion), 5: 85 (stat drop)
static constexpr uint64_t FX_CHANCES = 26ull | (77ull << 8) | (52ull << 16) | (103ull << 24) | (25ull << 32) | (85ull << 40);
constexpr uint32_t fx_status(uint32_t gate, uint32_t cls, uint32_t par) { return gate | (cls << 4) | (par << 7); }
constexpr uint32_t fx_stage(uint32_t idx, uint32_t n) { return idx | ((n - 1) << 3); } // stat idx 0 atk 1 def 2 spe 3 spc 4 acc 5 eva
constexpr uint32_t fx_sec(uint32_t kind, uint32_t chance_idx, uint32_t par) { return kind | (chance_idx << 3) | (par << 6); }
constexpr uint32_t fx_desc(uint32_t e) {
  switch (e) {
  case E_Confusion: return fx_status(G_SUB, A_CONF, 0);
  case E_Conversion: case E_Transform: return fx_status(G_INVUL, A_HEAVY, 0);
  case E_FocusEnergy: return fx_status(G_NONE, A_SVOL, 9);   // V_FOCUSENERGY
  case E_LightScreen: return fx_status(G_NONE, A_SVOL, 15);  // V_LIGHTSCREEN
  case E_Reflect: return fx_status(G_NONE, A_SVOL, 16);      // V_REFLECT
  case E_Mist: return fx_status(G_NONE, A_SVOL, 8);          // V_MIST
  case E_LeechSeed: return fx_status(G_GRASS, A_FVOL, 13);   // V_LEECHSEED
  case E_Haze: case E_Heal: case E_Substitute: case E_Bide: return fx_status(G_NONE, A_HEAVY, 0);
  case E_Mimic: return fx_status(G_HIT, A_HEAVY, 0);
  case E_Paralyze: return fx_status(G_PAR, A_PAR, 0);
  case E_Poison: return fx_status(G_PSN, A_HEAVY, 0);
  case E_SwitchAndTeleport: return fx_status(G_TELE, A_NONE, 0);
  case E_Sleep: return fx_status(G_SLEEP, A_HEAVY, 0);
  case E_Disable: return fx_status(G_DISABLE, A_HEAVY, 0);
  case E_AccuracyDown1: return fx_status(G_SUB, A_UNBOOST, fx_stage(4, 1));
  case E_AttackDown1: return fx_status(G_SUB, A_UNBOOST, fx_stage(0, 1));
  case E_DefenseDown1: return fx_status(G_SUB, A_UNBOOST, fx_stage(1, 1));
  case E_DefenseDown2: return fx_status(G_SUB, A_UNBOOST, fx_stage(1, 2));
  case E_SpeedDown1: return fx_status(G_SUB, A_UNBOOST, fx_stage(2, 1));
  case E_AttackUp1: return fx_status(G_NONE, A_BOOST, fx_stage(0, 1));
  case E_AttackUp2: return fx_status(G_NONE, A_BOOST, fx_stage(0, 2));
  case E_DefenseUp1: return fx_status(G_NONE, A_BOOST, fx_stage(1, 1));
  case E_DefenseUp2: return fx_status(G_NONE, A_BOOST, fx_stage(1, 2));
  case E_SpeedUp2: return fx_status(G_NONE, A_BOOST, fx_stage(2, 2));
  case E_SpecialUp1: return fx_status(G_NONE, A_BOOST, fx_stage(3, 1));
  case E_SpecialUp2: return fx_status(G_NONE, A_BOOST, fx_stage(3, 2));
  case E_EvasionUp1: return fx_status(G_NONE, A_BOOST, fx_stage(5, 1));
  // damaging: secondary effects; the status parameter p means status byte 8 << p (PSN BRN FRZ PAR)
  case E_PoisonChance1: case E_Twineedle: return fx_sec(SEC_STATUS, 2, 0);
  case E_PoisonChance2: return fx_sec(SEC_STATUS, 3, 0);
  case E_BurnChance1: return fx_sec(SEC_STATUS, 0, 1);
  case E_BurnChance2: return fx_sec(SEC_STATUS, 1, 1);
  case E_FreezeChance: return fx_sec(SEC_STATUS, 0, 2);
  case E_ParalyzeChance1: return fx_sec(SEC_STATUS, 0, 3);
  case E_ParalyzeChance2: return fx_sec(SEC_STATUS, 1, 3);
  case E_FlinchChance1: return fx_sec(SEC_FLINCH, 0, 0);
  case E_FlinchChance2: return fx_sec(SEC_FLINCH, 1, 0);
  case E_ConfusionChance: return fx_sec(SEC_CONF, 4, 0);
  case E_AttackDownChance: return fx_sec(SEC_UNBOOST, 5, 0);
  case E_DefenseDownChance: return fx_sec(SEC_UNBOOST, 5, 1);
  case E_SpeedDownChance: return fx_sec(SEC_UNBOOST, 5, 2);
  case E_SpecialDownChance: return fx_sec(SEC_UNBOOST, 5, 3);
  default: return 0; // Splash, plain damage, effects handled outside run_move
  }
}
struct FxTable {
  uint16_t d[68];
  constexpr FxTable() : d{} { for (uint32_t e = 0; e < 68; ++e) d[e] = (uint16_t)fx_desc(e); }
};
static __device__ const FxTable OAK_FX{};

// where each table sits in the workgroup's LDS table area (TABLE_LDS_BYTES)
__device__ __forceinline__ Tables tables_at(lds_u8 *lds) {
  lds_u32 *mv = (lds_u32 *)lds;
  lds_u8 *pp = lds + 166 * 4;
  lds_u32 *sp0 = (lds_u32 *)(pp + 168);
  lds_u32 *sp1 = sp0 + 152;
  lds_u8 *chart = (lds_u8 *)(sp1 + 152);
  lds_u16 *boost = (lds_u16 *)(chart + 228);
  lds_u32 *rcp = (lds_u32 *)(chart + 228 + 28);
  lds_u16 *fx = (lds_u16 *)(rcp + 256);
  Tables t{mv, pp, sp0, sp1, chart, boost, rcp, fx};
  return t;
}

// builds the tables in LDS from the packed images; call with all threads of the workgroup, then barrier.  (Only the
// table-image builder does this: the rollout kernels copy the finished image, see stage_default_tables.)
__device__ inline Tables stage_tables(lds_u8 *lds, const uint32_t *g_mv, const uint8_t *g_pp, const uint32_t *g_sp0,
                                      const uint32_t *g_sp1, const uint8_t *g_chart, const uint16_t *g_boost) {
  lds_u32 *mv = (lds_u32 *)lds;
  lds_u8 *pp = lds + 166 * 4;
  lds_u32 *sp0 = (lds_u32 *)(pp + 168);
  lds_u32 *sp1 = sp0 + 152;
  lds_u8 *chart = (lds_u8 *)(sp1 + 152);
  lds_u16 *boost = (lds_u16 *)(chart + 228);
  lds_u32 *rcp = (lds_u32 *)(chart + 228 + 28);
  lds_u16 *fx = (lds_u16 *)(rcp + 256);
  for (int i = threadIdx.x; i < 256; i += blockDim.x) rcp[i] = i ? 0xFFFFFFFFu / (uint32_t)i + 1u : 0u;
  for (int i = threadIdx.x; i < 68; i += blockDim.x) fx[i] = OAK_FX.d[i];
  for (int i = threadIdx.x; i < 166; i += blockDim.x) { mv[i] = g_mv[i]; pp[i] = g_pp[i]; }
  for (int i = threadIdx.x; i < 152; i += blockDim.x) { sp0[i] = g_sp0[i]; sp1[i] = g_sp1[i]; }
  for (int i = threadIdx.x; i < 225; i += blockDim.x) chart[i] = g_chart[i];
  for (int i = threadIdx.x; i < 13; i += blockDim.x) boost[i] = g_boost[i];
  Tables t{mv, pp, sp0, sp1, chart, boost, rcp, fx};
  return t;
}

struct Move { // unpacked move word
  uint32_t w;
  __device__ __forceinline__ uint32_t effect() const { return w & 0xFF; }
  __device__ __forceinline__ uint32_t bp() const { return (w >> 8) & 0xFF; }
  __device__ __forceinline__ uint32_t type() const { return (w >> 16) & 0xFF; }
  __device__ __forceinline__ uint32_t acc() const { return w >> 24; }
};

// ---- one lane's engine -------------------------------------------------------------------
template <int STRIDE, bool TRACK_ACTIONS>
struct Engine {
  lds_u32 *m;         // this lane's column in the lane-interleaved LDS state
  Tables T;
  // Per-player registers are packed into scalars and indexed by shift (never as arrays: a
  // lane-divergent player index into a register array would be demoted to scratch memory).
  uint64_t dur64;     // chance durations (public counters): side 0 in bits 0-31, side 1 in 32-63
  uint64_t act0, act1; // chance actions (only maintained when TRACK_ACTIONS)
  uint32_t over16;    // calc damage-roll overrides: P1 byte 0, P2 byte 1 (0 = roll)

  // -- LDS accessors (byte offset -> lane-interleaved address) --
  __device__ __forceinline__ uint32_t r32(int off) const { return m[(off >> 2) * STRIDE]; }
  __device__ __forceinline__ void w32(int off, uint32_t v) { m[(off >> 2) * STRIDE] = v; }
  __device__ __forceinline__ uint32_t r16(int off) const {
    return ((const lds_u16 *)m)[(off >> 2) * (STRIDE * 2) + ((off >> 1) & 1)];
  }
  __device__ __forceinline__ void w16(int off, uint32_t v) {
    ((lds_u16 *)m)[(off >> 2) * (STRIDE * 2) + ((off >> 1) & 1)] = (uint16_t)v;
  }
  __device__ __forceinline__ uint32_t r8(int off) const { return ((const lds_u8 *)m)[(off >> 2) * (STRIDE * 4) + (off & 3)]; }
  __device__ __forceinline__ void w8(int off, uint32_t v) { ((lds_u8 *)m)[(off >> 2) * (STRIDE * 4) + (off & 3)] = (uint8_t)v; }

  // -- RNG: showdown PSRNG over the 64-bit LCG (cpp/include/libpkmn/rng.h:9-11) --
  __device__ __forceinline__ uint32_t rng_next() {
    uint64_t s = (uint64_t)r32(B_RNG) | ((uint64_t)r32(B_RNG + 4) << 32);
    s = 0x5D588B656C078965ull * s + 0x0000000000269EC3ull;
    w32(B_RNG, (uint32_t)s);
    w32(B_RNG + 4, (uint32_t)(s >> 32));
    return (uint32_t)(s >> 32);
  }
  __device__ __forceinline__ uint32_t rng_range(uint32_t from, uint32_t to) {
    return from + (uint32_t)(((uint64_t)rng_next() * (uint64_t)(to - from)) >> 32);
  }
  __device__ __forceinline__ bool rng_chance(uint32_t num) { return rng_range(0, 256) < num; }

  // -- chance bookkeeping --
  __device__ __forceinline__ void act_set(int p, int sh, int bits, uint32_t v) {
    if constexpr (TRACK_ACTIONS) {
      uint64_t mask = ((1ull << bits) - 1) << sh;
      uint64_t a = p ? act1 : act0;
      a = (a & ~mask) | (((uint64_t)v << sh) & mask);
      if (p) act1 = a; else act0 = a;
    }
  }
  __device__ __forceinline__ void act_bool(int p, int sh, bool v) { act_set(p, sh, 2, v ? 2u : 1u); }
  __device__ __forceinline__ uint32_t dur_of(int p) const { return (uint32_t)(dur64 >> (32 * p)); }
  __device__ __forceinline__ void set_dur_of(int p, uint32_t v) {
    dur64 = (dur64 & ~(0xFFFFFFFFull << (32 * p))) | ((uint64_t)v << (32 * p));
  }
  __device__ __forceinline__ uint32_t dget(int p, int sh, int bits) const { return (uint32_t)(dur64 >> (32 * p + sh)) & ((1u << bits) - 1); }
  __device__ __forceinline__ void dset(int p, int sh, int bits, uint32_t v) {
    uint64_t mask = (uint64_t)((1u << bits) - 1) << (32 * p + sh);
    dur64 = (dur64 & ~mask) | (((uint64_t)v << (32 * p + sh)) & mask);
  }

  // -- field helpers; `so` = byte offset of a side, `ao` = so + O_ACTIVE --
  __device__ __forceinline__ int stored_off(int so) const { return so + PK_SZ * ((int)r8(so + O_ORDER) - 1); }
  __device__ __forceinline__ uint32_t vlo(int so) const { return r32(so + O_ACTIVE + A_VOL); }
  __device__ __forceinline__ uint32_t vhi(int so) const { return r32(so + O_ACTIVE + A_VOL + 4); }
  __device__ __forceinline__ void set_vlo(int so, uint32_t v) { w32(so + O_ACTIVE + A_VOL, v); }
  __device__ __forceinline__ void set_vhi(int so, uint32_t v) { w32(so + O_ACTIVE + A_VOL + 4, v); }
  __device__ __forceinline__ void vflag_set(int so, uint32_t f) { set_vlo(so, vlo(so) | f); }
  __device__ __forceinline__ void vflag_clear(int so, uint32_t f) { set_vlo(so, vlo(so) & ~f); }
  // lo-dword fields
  __device__ __forceinline__ uint32_t conf_left(int so) const { return (vlo(so) >> 18) & 7; }
  __device__ __forceinline__ void set_conf_left(int so, uint32_t x) { set_vlo(so, (vlo(so) & ~(7u << 18)) | ((x & 7) << 18)); }
  __device__ __forceinline__ uint32_t attacks(int so) const { return (vlo(so) >> 21) & 7; }
  __device__ __forceinline__ void set_attacks(int so, uint32_t x) { set_vlo(so, (vlo(so) & ~(7u << 21)) | ((x & 7) << 21)); }
  // state u16 straddles the two dwords (bits 24..39)
  __device__ __forceinline__ uint32_t vstate(int so) const { return (vlo(so) >> 24) | ((vhi(so) & 0xFF) << 8); }
  __device__ __forceinline__ void set_vstate(int so, uint32_t x) {
    set_vlo(so, (vlo(so) & 0x00FFFFFFu) | ((x & 0xFF) << 24));
    set_vhi(so, (vhi(so) & ~0xFFu) | ((x >> 8) & 0xFF));
  }
  // hi-dword fields
  __device__ __forceinline__ uint32_t sub_hp(int so) const { return (vhi(so) >> 8) & 0xFF; }
  __device__ __forceinline__ void set_sub_hp(int so, uint32_t x) { set_vhi(so, (vhi(so) & ~(0xFFu << 8)) | ((x & 0xFF) << 8)); }
  __device__ __forceinline__ uint32_t transform_id(int so) const { return (vhi(so) >> 16) & 15; }
  __device__ __forceinline__ void set_transform_id(int so, uint32_t x) { set_vhi(so, (vhi(so) & ~(15u << 16)) | ((x & 15) << 16)); }
  __device__ __forceinline__ uint32_t disable_left(int so) const { return (vhi(so) >> 20) & 15; }
  __device__ __forceinline__ void set_disable_left(int so, uint32_t x) { set_vhi(so, (vhi(so) & ~(15u << 20)) | ((x & 15) << 20)); }
  __device__ __forceinline__ uint32_t disable_move(int so) const { return (vhi(so) >> 24) & 7; }
  __device__ __forceinline__ void set_disable_move(int so, uint32_t x) { set_vhi(so, (vhi(so) & ~(7u << 24)) | ((x & 7) << 24)); }
  __device__ __forceinline__ uint32_t toxic_ctr(int so) const { return vhi(so) >> 27; }
  __device__ __forceinline__ void set_toxic_ctr(int so, uint32_t x) { set_vhi(so, (vhi(so) & ~(31u << 27)) | ((x & 31) << 27)); }

  __device__ __forceinline__ Move move_data(uint32_t id) const { return Move{T.mv[id]}; }
  __device__ __forceinline__ uint32_t chart(uint32_t atk_type, uint32_t def_type) const { return T.chart[atk_type * 15 + def_type]; }
  __device__ __forceinline__ bool has_type(uint32_t types, uint32_t t) const { return (types & 15) == t || (types >> 4) == t; }
  __device__ __forceinline__ int boost_get(int so, int idx) const { // 0 atk 1 def 2 spe 3 spc 4 acc 5 eva
    uint32_t n = (r8(so + O_ACTIVE + A_BOOSTS + (idx >> 1)) >> ((idx & 1) * 4)) & 15;
    return (int)((n ^ 8) - 8);
  }
  __device__ __forceinline__ void boost_put(int so, int idx, int v) {
    int off = so + O_ACTIVE + A_BOOSTS + (idx >> 1), sh = (idx & 1) * 4;
    w8(off, (r8(off) & ~(15u << sh)) | (((uint32_t)v & 15) << sh));
  }
  __device__ __forceinline__ uint32_t scale_boost(uint32_t x, int stage) const {
    uint32_t b = T.boost[stage + 6];
    return x * (b & 0xFF) / (b >> 8);
  }
  __device__ __forceinline__ void status_modify(uint32_t status, int so) { // PAR quarters speed, BRN halves attack
    int ao = so + O_ACTIVE;
    if (status & ST_PAR) { uint32_t s = r16(ao + P_SPE) / 4; w16(ao + P_SPE, s < 1 ? 1 : s); }
    else if (status & ST_BRN) { uint32_t a = r16(ao + P_ATK) / 2; w16(ao + P_ATK, a < 1 ? 1 : a); }
  }
  // byte offset of the Pokemon whose unmodified stats the active one currently uses
  __device__ __forceinline__ int unmodified_off(int so) const {
    if (!(vlo(so) & V_TRANSFORM)) return stored_off(so);
    uint32_t id = transform_id(so);
    return (int)(id >> 3) * SIDE_SZ + PK_SZ * ((int)(id & 7) - 1);
  }
  __device__ __forceinline__ bool any_alive(int so) const {
    uint32_t a = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i) a |= r16(so + PK_SZ * i + P_HP);
    return a != 0;
  }
  __device__ __forceinline__ void clear_binding(int p) {
    vflag_clear(p * SIDE_SZ, V_BINDING);
    dset(p, 28, 3, 0);
  }

  // ---- switching ---------------------------------------------------------------------
  __device__ void switch_in(int p, int slot) {
    const int so = p * SIDE_SZ, fo = (p ^ 1) * SIDE_SZ;
    int out = stored_off(so);
    if (r8(out + P_STATUS) == ST_TOX) w8(out + P_STATUS, ST_PSN); // toxic reverts on leaving the field
    uint32_t t = r8(so + O_ORDER);
    uint32_t in_id = r8(so + O_ORDER + slot - 1);
    w8(so + O_ORDER, in_id);
    w8(so + O_ORDER + slot - 1, t);
    uint32_t d = dur_of(p);
    uint32_t s0 = d & 7, sk = (d >> (3 * (slot - 1))) & 7;
    d = (d & ~7u) | sk;
    if (slot != 1) d = (d & ~(7u << (3 * (slot - 1)))) | (s0 << (3 * (slot - 1)));
    set_dur_of(p, d & ((1u << 18) - 1));
    w8(so + O_LAST_USED, 0);
    w8(fo + O_LAST_USED, 0);
    const int in = so + PK_SZ * ((int)in_id - 1), ao = so + O_ACTIVE;
    // stats (10 B) + moves (8 B) are dword-copyable: pokemon dwords 0..4 hold stats+moves[0..1]...
    uint32_t w0 = r32(in + 0), w1 = r32(in + 4), w2 = r32(in + 8), w3 = r32(in + 12), w4 = r32(in + 16), w5 = r32(in + 20);
    // pokemon: [hp atk][def spe][spc m0][m1 m2][m3 hpcur][status species types level]
    w32(ao + 0, w0);
    w32(ao + 4, w1);
    // active dword 2: spc(16) | species(8) | types(8)
    w32(ao + 8, (w2 & 0xFFFF) | (((w5 >> 8) & 0xFF) << 16) | (((w5 >> 16) & 0xFF) << 24));
    w32(ao + 12, 0);          // boosts
    w32(ao + 16, 0);          // volatiles lo
    w32(ao + 20, 0);          // volatiles hi
    w32(ao + 24, (w2 >> 16) | (w3 << 16));            // moves 0,1
    w32(ao + 28, (w3 >> 16) | (w4 << 16));            // moves 2,3
    status_modify(w5 & 0xFF, so);
    clear_binding(p ^ 1);
  }

  // ---- move selection ------------------------------------------------------------------
  __device__ __forceinline__ void select_move(int p, uint32_t choice) {
    if ((choice & 3) == C_PASS) return;
    const int so = p * SIDE_SZ;
    uint32_t v = vlo(so);
    if (v & (V_RECHARGING | V_RAGE)) return;
    if (v & V_FLINCH) { v &= ~V_FLINCH; set_vlo(so, v); }
    if (v & (V_THRASHING | V_CHARGING)) return;
    if ((choice & 3) == C_SWITCH) return;
    if (v & (V_BIDE | V_BINDING)) return;
    uint32_t data = choice >> 2;
    w8(so + O_LAST_SEL, data == 0 ? (uint32_t)M_Struggle : r8(so + O_ACTIVE + A_MOVES + 2 * ((int)data - 1)));
    w8(B_LAST_MOVES + 2 * p, data);
  }

  __device__ __forceinline__ int turn_order(uint32_t c1, uint32_t c2) {
    uint32_t t1 = c1 & 3, t2 = c2 & 3;
    if (t1 == C_PASS) return 1;
    if (t2 == C_PASS) return 0;
    if ((t1 == C_SWITCH) != (t2 == C_SWITCH)) return t1 == C_SWITCH ? 0 : 1;
    if (t1 == C_MOVE) {
      uint32_t m1 = r8(O_LAST_SEL), m2 = r8(SIDE_SZ + O_LAST_SEL);
      if ((m1 == M_QuickAttack) != (m2 == M_QuickAttack)) return m1 == M_QuickAttack ? 0 : 1;
      if ((m1 == M_Counter) != (m2 == M_Counter)) return m1 == M_Counter ? 1 : 0;
    }
    uint32_t s1 = r16(O_ACTIVE + P_SPE), s2 = r16(SIDE_SZ + O_ACTIVE + P_SPE);
    if (s1 == s2) {
      bool p1 = rng_range(0, 2) == 0;
      act_set(0, AC_SPEEDTIE, 2, p1 ? 1 : 2);
      act_set(1, AC_SPEEDTIE, 2, p1 ? 1 : 2);
      return p1 ? 0 : 1;
    }
    return s1 > s2 ? 0 : 1;
  }

  // ---- damage ------------------------------------------------------------------------------
  __device__ __forceinline__ bool check_crit(int p, Move mv) {
    const int so = p * SIDE_SZ;
    uint32_t species = r8(stored_off(so) + P_SPECIES);
    uint32_t chance = (T.sp0[species] >> 24) / 2; // base speed / 2
    if (vlo(so) & V_FOCUSENERGY) chance = chance / 2;
    else { chance *= 2; if (chance > 255) chance = 255; }
    if (mv.effect() == E_HighCritical) { chance *= 4; if (chance > 255) chance = 255; }
    else chance = chance / 2;
    bool crit = rng_chance(chance);
    act_bool(p, AC_CRIT, crit);
    return crit;
  }

  // base damage -> battle.last_damage; `tp` = target player (== p for confusion self-hits)
  __device__ bool calc_damage(int p, int tp, uint32_t bp, uint32_t type, bool explode, bool crit) {
    const int so = p * SIDE_SZ, to = tp * SIDE_SZ;
    const bool special = type >= 8;
    uint32_t atk, def;
    if (crit) {
      atk = r16(unmodified_off(so) + (special ? P_SPC : P_ATK));
      def = r16(unmodified_off(to) + (special ? P_SPC : P_DEF));
    } else {
      atk = r16(so + O_ACTIVE + (special ? P_SPC : P_ATK));
      uint32_t tv = vlo(to);
      def = r16(to + O_ACTIVE + (special ? P_SPC : P_DEF)) * ((tv & (special ? V_LIGHTSCREEN : V_REFLECT)) ? 2u : 1u);
    }
    if (atk > 255 || def > 255) {
      atk = (atk / 4) & 255; if (atk < 1) atk = 1;
      def = (def / 4) & 255; if (def < 1) def = 1;
    }
    uint32_t lvl = r8(stored_off(so) + P_LEVEL) * (crit ? 2u : 1u);
    if (explode) { def = def / 2; if (def < 1) def = 1; }
    if (def == 0) return false;
    uint32_t d = (lvl * 2 / 5) + 2;
    d *= bp;
    d *= atk;
    d /= def;
    d /= 50;
    if (d > 997) d = 997;
    d += 2;
    w16(B_LAST_DAMAGE, d);
    return true;
  }

  __device__ __forceinline__ void adjust_damage(int p, Move mv) { // STAB then type1, type2
    const int so = p * SIDE_SZ, fo = (p ^ 1) * SIDE_SZ;
    uint32_t ft = r8(fo + O_ACTIVE + A_TYPES), t1 = ft & 15, t2 = ft >> 4;
    uint32_t d = r16(B_LAST_DAMAGE);
    if (has_type(r8(so + O_ACTIVE + A_TYPES), mv.type())) d = (d + d / 2) & 0xFFFF;
    uint32_t e1 = chart(mv.type(), t1), e2 = chart(mv.type(), t2);
    if (e1 != 2) d = (d * e1 / 2) & 0xFFFF;
    if (t1 != t2 && e2 != 2) d = (d * e2 / 2) & 0xFFFF;
    w16(B_LAST_DAMAGE, d);
  }

  __device__ __forceinline__ void randomize_damage(int p) {
    uint32_t d = r16(B_LAST_DAMAGE);
    if (d <= 1) return;
    uint32_t roll = (over16 >> (8 * p)) & 0xFF;
    if (roll == 0) roll = rng_range(217, 256);
    act_set(p, AC_DAMAGE, 8, roll);
    w16(B_LAST_DAMAGE, d * roll / 255);
  }

  // applies battle.last_damage to player tp (through sub_p's substitute when up).
  // returns true when a substitute absorbed the hit and broke; hit_sub: any substitute took it
  __device__ bool apply_damage(int tp, int sub_p, bool &hit_sub) {
    const int subo = sub_p * SIDE_SZ;
    hit_sub = false;
    uint32_t dmg = r16(B_LAST_DAMAGE);
    if (vlo(subo) & V_SUBSTITUTE) {
      hit_sub = true;
      uint32_t hp = sub_hp(subo);
      if (dmg >= hp) { set_sub_hp(subo, 0); vflag_clear(subo, V_SUBSTITUTE); return true; }
      set_sub_hp(subo, hp - dmg);
      return false;
    }
    int st = stored_off(tp * SIDE_SZ);
    uint32_t hp = r16(st + P_HP);
    if (dmg > hp) { dmg = hp; w16(B_LAST_DAMAGE, dmg); }
    w16(st + P_HP, hp - dmg);
    return false;
  }

  __device__ bool move_hit(int p, Move mv) {
    const int so = p * SIDE_SZ, fo = (p ^ 1) * SIDE_SZ;
    bool miss;
    const uint32_t eff = mv.effect();
    if (eff == E_Swift) return true;
    const uint32_t fv = vlo(fo);
    if (fv & V_INVULNERABLE) miss = true;
    else if ((eff == E_DrainHP || eff == E_DreamEater) && (fv & V_SUBSTITUTE)) miss = true;
    else if (eff >= E_AccuracyDown1 && eff <= E_SpeedDown1 && (fv & V_MIST)) miss = true;
    else {
      uint32_t acc = mv.acc();
      acc = scale_boost(acc, boost_get(so, 4));
      acc = scale_boost(acc, -boost_get(fo, 5));
      if (acc > 255) acc = 255;
      if (acc < 1) acc = 1;
      if (acc == 255) miss = false; // miss=false build: no 1/256 miss, no roll
      else { miss = !rng_chance(acc); act_bool(p, AC_HIT, !miss); }
    }
    if (!miss) return true;
    w16(B_LAST_DAMAGE, 0);
    clear_binding(p);
    return false;
  }

  // ---- stat stages ---------------------------------------------------------------------
  __device__ bool boost_self(int p, int idx, int n) {
    const int so = p * SIDE_SZ, fo = (p ^ 1) * SIDE_SZ;
    int cur = boost_get(so, idx);
    if (cur >= 6) return false;
    int nv = cur + n; if (nv > 6) nv = 6;
    if (idx < 4) {
      const int field = P_ATK + 2 * idx;
      if (r16(so + O_ACTIVE + field) == 999) return false;
      boost_put(so, idx, nv);
      uint32_t x = scale_boost(r16(unmodified_off(so) + field), nv);
      if (x > 999) x = 999;
      w16(so + O_ACTIVE + field, x);
    } else boost_put(so, idx, nv);
    status_modify(r8(stored_off(fo) + P_STATUS), fo); // stat modification glitch
    return true;
  }
  __device__ bool unboost_foe(int p, int idx, int n) {
    const int fo = (p ^ 1) * SIDE_SZ;
    int cur = boost_get(fo, idx);
    if (cur <= -6) return false;
    int nv = cur - n; if (nv < -6) nv = -6;
    if (idx < 4) {
      const int field = P_ATK + 2 * idx;
      if (r16(fo + O_ACTIVE + field) == 1) return false;
      boost_put(fo, idx, nv);
      uint32_t x = scale_boost(r16(unmodified_off(fo) + field), nv);
      if (x < 1) x = 1;
      w16(fo + O_ACTIVE + field, x);
    } else boost_put(fo, idx, nv);
    status_modify(r8(stored_off(fo) + P_STATUS), fo);
    return true;
  }

  // ---- effects that run instead of damage (Effect onBegin group, data/moves.h:202-218) ---
  __device__ void haze_clear(int p) {
    const int so = p * SIDE_SZ;
    set_disable_move(so, 0);
    set_disable_left(so, 0);
    dset(p, 21, 4, 0);
    uint32_t v = vlo(so);
    if (v & V_CONFUSION) { v &= ~(V_CONFUSION | (7u << 18)); dset(p, 18, 3, 0); }
    v &= ~(V_MIST | V_FOCUSENERGY | V_LEECHSEED | V_LIGHTSCREEN | V_REFLECT);
    bool tox = v & V_TOXIC;
    v &= ~V_TOXIC;
    set_vlo(so, v);
    if (tox) {
      set_toxic_ctr(so, 0);
      int st = stored_off(so);
      if (r8(st + P_STATUS) == ST_TOX) w8(st + P_STATUS, ST_PSN);
    }
  }
  __device__ void start_confusion(int tp) {
    const int to = tp * SIDE_SZ;
    vflag_set(to, V_CONFUSION);
    set_conf_left(to, rng_range(2, 6));
    dset(tp, 18, 3, 1);
    act_set(tp, AC_CONFUSION, 3, OBS_STARTED);
  }

  __device__ void on_begin(int p, Move mv, uint32_t move_id, uint32_t mslot) {
    const int so = p * SIDE_SZ, fo = (p ^ 1) * SIDE_SZ;
    const int sp = stored_off(so), fp = stored_off(fo);
    w16(B_LAST_DAMAGE, 0);
    switch (mv.effect()) {
    case E_Confusion:
      if (vlo(fo) & V_SUBSTITUTE) return;
      if (!move_hit(p, mv)) return;
      if (vlo(fo) & V_CONFUSION) return;
      start_confusion(p ^ 1);
      return;
    case E_Conversion:
      if (vlo(fo) & V_INVULNERABLE) return;
      w8(so + O_ACTIVE + A_TYPES, r8(fo + O_ACTIVE + A_TYPES));
      return;
    case E_FocusEnergy: vflag_set(so, V_FOCUSENERGY); return;
    case E_Haze: {
      w32(so + O_ACTIVE + A_BOOSTS, 0);
      w32(fo + O_ACTIVE + A_BOOSTS, 0);
      for (int q = 0; q < 2; ++q) {
        const int qo = q * SIDE_SZ, uo = unmodified_off(qo);
        uint32_t a = r32(uo), b = r32(uo + 4), c = r16(uo + 8);
        w32(qo + O_ACTIVE, a);
        w32(qo + O_ACTIVE + 4, b);
        w16(qo + O_ACTIVE + 8, c);
      }
      uint32_t fs = r8(fp + P_STATUS);
      if (fs) {
        if (fs & ST_SLP) dset(p ^ 1, 0, 3, 0);
        w8(fp + P_STATUS, 0);
      }
      if (r8(sp + P_STATUS) == ST_TOX) w8(sp + P_STATUS, ST_PSN);
      haze_clear(p);
      haze_clear(p ^ 1);
      return;
    }
    case E_Heal: {
      uint32_t maxhp = r16(sp + P_HP_MAX), hp = r16(sp + P_HP);
      uint32_t delta = maxhp - hp;
      if (delta == 0 || (delta & 255) == 255) return; // gen-1 recovery failure glitch
      if (move_id == M_Rest) {
        w8(sp + P_STATUS, ST_EXT | 2);
        dset(p, 0, 3, 0);
        w16(sp + P_HP, maxhp);
        vflag_clear(so, V_TOXIC);
        set_toxic_ctr(so, 0);
      } else {
        uint32_t h = hp + maxhp / 2;
        w16(sp + P_HP, h > maxhp ? maxhp : h);
      }
      return;
    }
    case E_LeechSeed:
      if (has_type(r8(fo + O_ACTIVE + A_TYPES), T_Grass)) return;
      if (!move_hit(p, mv)) return;
      if (vlo(fo) & V_LEECHSEED) return;
      vflag_set(fo, V_LEECHSEED);
      return;
    case E_LightScreen: vflag_set(so, V_LIGHTSCREEN); return;
    case E_Reflect: vflag_set(so, V_REFLECT); return;
    case E_Mist: vflag_set(so, V_MIST); return;
    case E_Mimic: {
      if (!move_hit(p, mv)) return;
      uint32_t n = 0;
      for (int i = 0; i < 4; ++i) n += r8(fo + O_ACTIVE + A_MOVES + 2 * i) != 0;
      if (n == 0 || mslot == 0) return;
      uint32_t r = rng_range(0, n);
      act_set(p, AC_MOVESLOT, 4, r + 1);
      w8(so + O_ACTIVE + A_MOVES + 2 * ((int)mslot - 1), r8(fo + O_ACTIVE + A_MOVES + 2 * (int)r));
      return;
    }
    case E_Paralyze: {
      if (r8(fp + P_STATUS)) return;
      uint32_t ft = r8(fo + O_ACTIVE + A_TYPES);
      if (chart(mv.type(), ft & 15) == 0 || chart(mv.type(), ft >> 4) == 0) return;
      if (!move_hit(p, mv)) return;
      w8(fp + P_STATUS, ST_PAR);
      uint32_t s = r16(fo + O_ACTIVE + P_SPE) / 4;
      w16(fo + O_ACTIVE + P_SPE, s < 1 ? 1 : s);
      return;
    }
    case E_Poison:
      if (r8(fp + P_STATUS)) return;
      if (has_type(r8(fo + O_ACTIVE + A_TYPES), T_Poison)) return;
      if (vlo(fo) & V_SUBSTITUTE) return;
      if (!move_hit(p, mv)) return;
      if (move_id == M_Toxic) { w8(fp + P_STATUS, ST_TOX); vflag_set(fo, V_TOXIC); set_toxic_ctr(fo, 0); }
      else w8(fp + P_STATUS, ST_PSN);
      return;
    case E_Splash: return;
    case E_Substitute: {
      if (vlo(so) & V_SUBSTITUTE) return;
      uint32_t cost = r16(sp + P_HP_MAX) / 4, hp = r16(sp + P_HP);
      if (hp < cost) return;
      w16(sp + P_HP, hp - cost); // exactly a quarter left: the user faints (gen-1 behaviour)
      set_sub_hp(so, cost + 1);
      vflag_set(so, V_SUBSTITUTE);
      return;
    }
    case E_SwitchAndTeleport:
      if (move_id != M_Teleport) (void)move_hit(p, mv);
      return;
    case E_Transform: {
      uint32_t fv = vlo(fo);
      if (fv & V_INVULNERABLE) return;
      uint32_t id = (fv & V_TRANSFORM) ? transform_id(fo) : (uint32_t)(((p ^ 1) << 3) | r8(fo + O_ORDER));
      vflag_set(so, V_TRANSFORM);
      set_transform_id(so, id);
      const int sa = so + O_ACTIVE, fa = fo + O_ACTIVE;
      w32(sa + 0, r32(fa + 0));
      w32(sa + 4, r32(fa + 4));
      w32(sa + 8, r32(fa + 8));   // spc, species, types
      w32(sa + 12, r32(fa + 12)); // boosts
      for (int i = 0; i < 4; ++i) {
        uint32_t mid = r8(fa + A_MOVES + 2 * i);
        w8(sa + A_MOVES + 2 * i, mid);
        w8(sa + A_MOVES + 2 * i + 1, mid ? 5 : 0);
      }
      return;
    }
    default: return;
    }
  }

  // ---- pre-move checks -------------------------------------------------------------------
  enum : int { BM_OK = 0, BM_DONE = 1, BM_SKIP_CAN = 2, BM_SKIP_PP = 3, BM_ERR = 4 };

  __device__ int before_move(int p) {
    const int so = p * SIDE_SZ, fo = (p ^ 1) * SIDE_SZ;
    const int sp = stored_off(so);
    bool dummy;
    uint32_t status = r8(sp + P_STATUS);
    if (status & ST_SLP) {
      status -= 1;
      uint32_t left = status & ST_SLP;
      if (!(status & ST_EXT)) {
        if (left == 0) { dset(p, 0, 3, 0); act_set(p, AC_SLEEP, 2, OBS_ENDED); }
        else { dset(p, 0, 3, dget(p, 0, 3) + 1); act_set(p, AC_SLEEP, 2, OBS_CONTINUING); }
      }
      if (left == 0) status = 0;
      w8(sp + P_STATUS, status);
      w8(so + O_LAST_USED, 0);
      return BM_DONE;
    }
    if (status & ST_FRZ) { w8(so + O_LAST_USED, 0); return BM_DONE; }
    if (vlo(fo) & V_BINDING) return BM_DONE;
    uint32_t v = vlo(so);
    if (v & V_FLINCH) { set_vlo(so, v & ~V_FLINCH); return BM_DONE; }
    if (v & V_RECHARGING) { set_vlo(so, v & ~V_RECHARGING); return BM_DONE; }
    uint32_t dl = disable_left(so);
    if (dl > 0) {
      dl -= 1;
      set_disable_left(so, dl);
      if (dl == 0) { set_disable_move(so, 0); dset(p, 21, 4, 0); act_set(p, AC_DISABLE, 2, OBS_ENDED); }
      else { dset(p, 21, 4, dget(p, 21, 4) + 1); act_set(p, AC_DISABLE, 2, OBS_CONTINUING); }
    }
    if (v & V_CONFUSION) {
      uint32_t left = conf_left(so) - 1;
      set_conf_left(so, left);
      if (left == 0) {
        vflag_clear(so, V_CONFUSION);
        dset(p, 18, 3, 0);
        act_set(p, AC_CONFUSION, 3, OBS_ENDED);
      } else {
        dset(p, 18, 3, dget(p, 18, 3) + 1);
        act_set(p, AC_CONFUSION, 3, OBS_CONTINUING);
        bool confused = !rng_chance(128);
        act_bool(p, AC_CONFUSED, confused);
        if (confused) {
          vflag_clear(so, V_BIDE | V_THRASHING | V_MULTIHIT | V_FLINCH | V_CHARGING | V_BINDING | V_INVULNERABLE);
          dset(p, 25, 3, 0);
          dset(p, 28, 3, 0);
          if (!calc_damage(p, p, 40, T_Normal, false, false)) return BM_ERR; // 40 bp typeless physical self-hit
          (void)apply_damage(p, p ^ 1, dummy);
          return BM_DONE;
        }
      }
    }
    uint32_t dm = disable_move(so);
    uint32_t sel = r8(so + O_LAST_SEL);
    if (dm != 0 && sel != M_Struggle && r8(so + O_ACTIVE + A_MOVES + 2 * ((int)dm - 1)) == sel) {
      vflag_clear(so, V_CHARGING);
      return BM_DONE;
    }
    if (status & ST_PAR) {
      bool par = rng_chance(63);
      act_bool(p, AC_PARALYZED, par);
      if (par) {
        vflag_clear(so, V_BIDE | V_THRASHING | V_CHARGING | V_BINDING | V_INVULNERABLE);
        dset(p, 25, 3, 0);
        dset(p, 28, 3, 0);
        return BM_DONE;
      }
    }
    v = vlo(so);
    if (v & V_BIDE) {
      uint32_t left = attacks(so) - 1;
      set_attacks(so, left);
      if (left != 0) { dset(p, 25, 3, dget(p, 25, 3) + 1); act_set(p, AC_ATTACKING, 2, OBS_CONTINUING); return BM_DONE; }
      dset(p, 25, 3, 0);
      act_set(p, AC_ATTACKING, 2, OBS_ENDED);
      vflag_clear(so, V_BIDE);
      uint32_t dmg = (vstate(so) * 2) & 0xFFFF;
      set_vstate(so, 0);
      w16(B_LAST_DAMAGE, dmg);
      if (dmg == 0) return BM_DONE;
      if (vlo(fo) & V_INVULNERABLE) return BM_DONE;
      (void)apply_damage(p ^ 1, p ^ 1, dummy);
      return BM_DONE;
    }
    if (v & V_THRASHING) {
      uint32_t left = attacks(so) - 1;
      set_attacks(so, left);
      if (left == 0) {
        vflag_clear(so, V_THRASHING);
        dset(p, 25, 3, 0);
        act_set(p, AC_ATTACKING, 2, OBS_ENDED);
        start_confusion(p);
      } else {
        dset(p, 25, 3, dget(p, 25, 3) + 1);
        act_set(p, AC_ATTACKING, 2, OBS_CONTINUING);
      }
      return BM_SKIP_CAN;
    }
    if (v & V_BINDING) {
      set_attacks(so, attacks(so) - 1);
      dset(p, 28, 3, dget(p, 28, 3) + 1);
      act_set(p, AC_BINDING, 3, OBS_CONTINUING);
      if (r16(B_LAST_DAMAGE) != 0) (void)apply_damage(p ^ 1, p ^ 1, dummy);
      return BM_DONE;
    }
    return (v & V_RAGE) ? BM_SKIP_PP : BM_OK;
  }

  __device__ __forceinline__ void decrement_pp(int so, uint32_t mslot) {
    if (mslot == 0) return;
    int a = so + O_ACTIVE + A_MOVES + 2 * ((int)mslot - 1) + 1;
    w8(a, (r8(a) - 1) & 63);
    if (vlo(so) & V_TRANSFORM) return;
    int s = stored_off(so) + P_MOVES + 2 * ((int)mslot - 1) + 1;
    w8(s, (r8(s) - 1) & 63);
  }

  // ---- secondary effects ---------------------------------------------------------------
  __device__ void secondary_status(int p, Move mv, uint32_t status, uint32_t num) {
    const int fo = (p ^ 1) * SIDE_SZ, fp = stored_off(fo);
    uint32_t fs = r8(fp + P_STATUS);
    if (status == ST_BRN && (fs & ST_FRZ)) { w8(fp + P_STATUS, 0); return; } // fire thaws
    if (fs) return;
    if (has_type(r8(fo + O_ACTIVE + A_TYPES), status == ST_PSN ? (uint32_t)T_Poison : mv.type())) return;
    bool proc = rng_chance(num);
    act_bool(p, AC_SECONDARY, proc);
    if (!proc) return;
    w8(fp + P_STATUS, status);
    if (status == ST_PAR) { uint32_t s = r16(fo + O_ACTIVE + P_SPE) / 4; w16(fo + O_ACTIVE + P_SPE, s < 1 ? 1 : s); }
    if (status == ST_BRN) { uint32_t a = r16(fo + O_ACTIVE + P_ATK) / 2; w16(fo + O_ACTIVE + P_ATK, a < 1 ? 1 : a); }
  }

  // ---- the move itself -----------------------------------------------------------------
  __device__ void do_move(int p) {
    const int so = p * SIDE_SZ, fo = (p ^ 1) * SIDE_SZ;
    const int sp = stored_off(so), fp = stored_off(fo);
    const uint32_t move_id = r8(so + O_LAST_SEL);
    const Move mv = move_data(move_id);
    const uint32_t eff = mv.effect();
    w8(B_LAST_MOVES + 2 * p + 1, 0); // counterable

    if (mv.bp() == 0) { // non-damaging moves resolved after the accuracy check
      w16(B_LAST_DAMAGE, 0);
      switch (eff) {
      case E_AttackUp1: boost_self(p, 0, 1); return;
      case E_AttackUp2: boost_self(p, 0, 2); return;
      case E_DefenseUp1: boost_self(p, 1, 1); return;
      case E_DefenseUp2: boost_self(p, 1, 2); return;
      case E_SpeedUp2: boost_self(p, 2, 2); return;
      case E_SpecialUp1: boost_self(p, 3, 1); return;
      case E_SpecialUp2: boost_self(p, 3, 2); return;
      case E_EvasionUp1: boost_self(p, 5, 1); return;
      case E_Bide:
        vflag_set(so, V_BIDE);
        set_vstate(so, 0);
        set_attacks(so, rng_range(2, 4));
        dset(p, 25, 3, 1);
        act_set(p, AC_ATTACKING, 2, OBS_STARTED);
        return;
      case E_AccuracyDown1: case E_AttackDown1: case E_DefenseDown1: case E_DefenseDown2: case E_SpeedDown1: {
        if (vlo(fo) & V_SUBSTITUTE) return;
        if (!move_hit(p, mv)) return;
        int idx = eff == E_AccuracyDown1 ? 4 : eff == E_AttackDown1 ? 0 : eff == E_SpeedDown1 ? 2 : 1;
        unboost_foe(p, idx, eff == E_DefenseDown2 ? 2 : 1);
        return;
      }
      case E_Sleep: {
        uint32_t fv = vlo(fo), fs = r8(fp + P_STATUS);
        if (fv & V_RECHARGING) {
          set_vlo(fo, fv & ~V_RECHARGING); // always lands on a recharging target
          if (fs & ST_SLP) return;
        } else {
          if (fs) return;
          if (!move_hit(p, mv)) return;
        }
        w8(fp + P_STATUS, rng_range(1, 8));
        dset(p ^ 1, 0, 3, 1);
        act_set(p ^ 1, AC_SLEEP, 2, OBS_STARTED);
        return;
      }
      case E_Disable: {
        if (disable_move(fo) != 0) return;
        if (!move_hit(p, mv)) return;
        uint32_t n = 0, packed = 0; // eligible slots packed 4 bits each
        for (int i = 0; i < 4; ++i) {
          uint32_t ms = r16(fo + O_ACTIVE + A_MOVES + 2 * i);
          if ((ms & 0xFF) && (ms >> 8)) { packed |= (uint32_t)(i + 1) << (4 * n); ++n; }
        }
        if (n == 0) return;
        uint32_t slot = (packed >> (4 * rng_range(0, n))) & 15;
        act_set(p, AC_MOVESLOT, 4, slot);
        set_disable_move(fo, slot);
        set_disable_left(fo, rng_range(1, 9));
        dset(p ^ 1, 21, 4, 1);
        act_set(p ^ 1, AC_DISABLE, 2, OBS_STARTED);
        return;
      }
      default: return;
      }
    }

    // damaging moves
    const bool fixed = eff == E_SpecialDamage || eff == E_SuperFang || move_id == M_Counter;
    const bool ohko = eff == E_OHKO;
    const uint32_t ft = r8(fo + O_ACTIVE + A_TYPES);
    bool immune = false;
    if (!fixed) immune = chart(mv.type(), ft & 15) == 0 || chart(mv.type(), ft >> 4) == 0;
    if (eff == E_DreamEater && !(r8(fp + P_STATUS) & ST_SLP)) immune = true;
    if (ohko && r16(so + O_ACTIVE + P_SPE) < r16(fo + O_ACTIVE + P_SPE)) immune = true;
#if OAK_COUNTER_SHOWDOWN
    if (move_id == M_Counter) {
      const uint32_t lu = r8(fo + O_LAST_USED), ls = r8(fo + O_LAST_SEL);
      const Move mu = move_data(lu), ms = move_data(ls);
      const bool cu = lu != 0 && lu != M_Counter && mu.bp() > 0 && (mu.type() == T_Normal || mu.type() == T_Fighting);
      const bool cs = ls != 0 && ls != M_Counter && ms.bp() > 0 && (ms.type() == T_Normal || ms.type() == T_Fighting);
      if (!(cu && cs) || r16(B_LAST_DAMAGE) == 0) immune = true;
    }
#else
    if (move_id == M_Counter && (!r8(B_LAST_MOVES + 2 * (p ^ 1) + 1) || r16(B_LAST_DAMAGE) == 0)) immune = true;
#endif
    bool hit = false;
    const bool late_hit = OAK_ACCURACY_LAST && !fixed && !ohko; // cartridge order: crit and damage roll in front of the accuracy roll
    if (!immune) hit = late_hit ? true : move_hit(p, mv);
    if (immune || !hit) {
      w16(B_LAST_DAMAGE, 0);
      clear_binding(p);
      if (eff == E_Explode) { w16(sp + P_HP, 0); w8(sp + P_STATUS, 0); }
      if (eff == E_JumpKick && !immune) { uint32_t hp = r16(sp + P_HP); if (hp > 0) w16(sp + P_HP, hp - 1); } // crash: 1 HP
      return;
    }

    uint32_t hits = 1;
    if (fixed) {
      uint32_t d;
      if (move_id == M_Counter) { d = r16(B_LAST_DAMAGE) * 2; if (d > 65535) d = 65535; }
      else if (eff == E_SuperFang) { d = r16(fp + P_HP) / 2; if (d < 1) d = 1; }
      else if (move_id == M_SonicBoom) d = 20;
      else if (move_id == M_DragonRage) d = 40;
      else if (move_id == M_Psywave) {
        uint32_t max = r8(sp + P_LEVEL) * 3 / 2;
#if OAK_PSYWAVE_SHOWDOWN
        d = rng_range(0, max ? max : 1); // Showdown: random(0, max), a 0 fails the move; the action holds the roll + 1
        act_set(p, AC_PSYWAVE, 8, d + 1);
        if (d == 0) { w16(B_LAST_DAMAGE, 0); clear_binding(p); return; }
#else
        d = max <= 1 ? 1 : rng_range(1, max);
        act_set(p, AC_PSYWAVE, 8, d);
#endif
      } else d = r8(sp + P_LEVEL); // SeismicToss, NightShade
      w16(B_LAST_DAMAGE, d);
    } else if (ohko) {
      w16(B_LAST_DAMAGE, 65535);
    } else {
#if OAK_MULTIHIT_ROLL_FIRST
      if (eff == E_MultiHit) { // Showdown order: the count behind the accuracy check, before crit / damage
        hits = (0x54333222u >> (4 * rng_range(0, 8))) & 15; // {2,2,2,3,3,3,4,5}
        act_set(p, AC_MULTIHIT, 4, hits);
      }
#endif
      bool crit = check_crit(p, mv);
      if (!calc_damage(p, p ^ 1, mv.bp(), mv.type(), eff == E_Explode, crit)) return;
      adjust_damage(p, mv);
      randomize_damage(p);
      if (r16(B_LAST_DAMAGE) == 0) { clear_binding(p); return; }
#if OAK_ACCURACY_LAST
      if (!move_hit(p, mv)) {
        w16(B_LAST_DAMAGE, 0);
        clear_binding(p);
        if (eff == E_Explode) { w16(sp + P_HP, 0); w8(sp + P_STATUS, 0); }
        if (eff == E_JumpKick) { uint32_t hp = r16(sp + P_HP); if (hp > 0) w16(sp + P_HP, hp - 1); }
        return;
      }
#endif
    }

    if (eff == E_DoubleHit || eff == E_Twineedle) hits = 2;
#if !OAK_MULTIHIT_ROLL_FIRST
    else if (eff == E_MultiHit) {
      hits = (0x54333222u >> (4 * rng_range(0, 8))) & 15; // {2,2,2,3,3,3,4,5}
      act_set(p, AC_MULTIHIT, 4, hits);
    }
#endif

    bool broke = false, hit_sub = false;
    uint32_t dealt = 0;
    const uint32_t per_hit = r16(B_LAST_DAMAGE);
    for (uint32_t h = 0; h < hits; ++h) {
      w16(B_LAST_DAMAGE, per_hit);
      broke = apply_damage(p ^ 1, p ^ 1, hit_sub);
      dealt = r16(B_LAST_DAMAGE);
      if (!hit_sub) {
        uint32_t fv = vlo(fo);
        if (fv & V_BIDE) set_vstate(fo, (vstate(fo) + dealt) & 0xFFFF);
        if ((fv & V_RAGE) && r16(fp + P_HP) > 0) (void)boost_self(p ^ 1, 0, 1); // rage builds
      }
      if (broke || r16(fp + P_HP) == 0) break;
    }
    w8(B_LAST_MOVES + 2 * p + 1, (mv.type() == T_Normal || mv.type() == T_Fighting) && move_id != M_Counter);

    // user-side consequences
    if (eff == E_Explode && !broke) { w16(sp + P_HP, 0); w8(sp + P_STATUS, 0); }
    if (eff == E_Recoil && !broke && dealt > 0) {
      uint32_t r = dealt / (move_id == M_Struggle ? 2u : 4u); if (r < 1) r = 1;
      uint32_t hp = r16(sp + P_HP);
      w16(sp + P_HP, r > hp ? 0 : hp - r);
    }
    if ((eff == E_DrainHP || eff == E_DreamEater) && dealt > 0) {
      uint32_t h = dealt / 2; if (h < 1) h = 1;
      h += r16(sp + P_HP);
      uint32_t maxhp = r16(sp + P_HP_MAX);
      w16(sp + P_HP, h > maxhp ? maxhp : h);
    }
    if (r16(fp + P_HP) == 0 || broke) return; // no secondary effects, no recharge, no binding
    if (eff == E_HyperBeam) { vflag_set(so, V_RECHARGING); return; }
    if (eff == E_Binding) {
      if (!(vlo(so) & V_BINDING)) {
        uint32_t n = (0x54333222u >> (4 * rng_range(0, 8))) & 15;
        vflag_set(so, V_BINDING);
        set_attacks(so, n - 1);
        dset(p, 28, 3, 1);
        act_set(p, AC_BINDING, 3, OBS_STARTED);
      }
      return;
    }
    if (hit_sub) return; // a standing substitute blocks every secondary effect
    switch (eff) {
    case E_BurnChance1: secondary_status(p, mv, ST_BRN, 26); break;
    case E_BurnChance2: secondary_status(p, mv, ST_BRN, 77); break;
    case E_FreezeChance: secondary_status(p, mv, ST_FRZ, 26); break;
    case E_ParalyzeChance1: secondary_status(p, mv, ST_PAR, 26); break;
    case E_ParalyzeChance2: secondary_status(p, mv, ST_PAR, 77); break;
    case E_PoisonChance1: secondary_status(p, mv, ST_PSN, 52); break;
    case E_PoisonChance2: secondary_status(p, mv, ST_PSN, 103); break;
    case E_Twineedle: secondary_status(p, mv, ST_PSN, 52); break;
    case E_FlinchChance1: case E_FlinchChance2: {
      bool proc = rng_chance(eff == E_FlinchChance1 ? 26 : 77);
      act_bool(p, AC_SECONDARY, proc);
      if (proc) vflag_set(fo, V_FLINCH);
      break;
    }
    case E_ConfusionChance: {
      if (vlo(fo) & V_CONFUSION) break;
      bool proc = rng_chance(25);
      act_bool(p, AC_SECONDARY, proc);
      if (proc) start_confusion(p ^ 1);
      break;
    }
    case E_AttackDownChance: case E_DefenseDownChance: case E_SpeedDownChance: case E_SpecialDownChance: {
      bool proc = rng_chance(85);
      act_bool(p, AC_SECONDARY, proc);
      if (proc) unboost_foe(p, (int)eff - (int)E_AttackDownChance, 1);
      break;
    }
    default: break;
    }
  }

  // canMove: charge turns, PP, Metronome / Mirror Move redirection, onBegin effects
  __device__ void execute_selected(int p, uint32_t mslot, bool skip_can, bool skip_pp) {
    const int so = p * SIDE_SZ, fo = (p ^ 1) * SIDE_SZ;
    if (!skip_can) {
#pragma unroll 1
      for (int depth = 0; depth < 4; ++depth) {
        const uint32_t move_id = r8(so + O_LAST_SEL);
        const Move mv = move_data(move_id);
        const uint32_t eff = mv.effect();
        uint32_t v = vlo(so);
        if (v & V_CHARGING) {
          set_vlo(so, v & ~(V_CHARGING | V_INVULNERABLE));
        } else if (eff == E_Charge) {
          v |= V_CHARGING;
          if (move_id == M_Fly || move_id == M_Dig) v |= V_INVULNERABLE;
          set_vlo(so, v);
          w8(so + O_LAST_USED, move_id);
          w8(B_LAST_MOVES + 2 * p + 1, 0);
          return;
        }
        w8(so + O_LAST_USED, move_id);
        w8(B_LAST_MOVES + 2 * p + 1, 0);
        if (!skip_pp) decrement_pp(so, mslot);
        skip_pp = true;
        if (eff == E_Metronome) {
          uint32_t r = rng_range(0, 163);
          uint32_t pick = (r + 1 >= M_Metronome) ? r + 2 : r + 1;
          act_set(p, AC_METRONOME, 8, pick);
          w8(so + O_LAST_SEL, pick);
          continue;
        }
        if (eff == E_MirrorMove) {
          uint32_t mm = r8(fo + O_LAST_USED);
          if (mm == 0 || mm == M_MirrorMove) { w16(B_LAST_DAMAGE, 0); return; }
          w8(so + O_LAST_SEL, mm);
          continue;
        }
        if (eff >= E_Confusion && eff <= E_Transform) { on_begin(p, mv, move_id, mslot); return; }
        if (eff == E_Thrashing) {
          vflag_set(so, V_THRASHING);
          set_attacks(so, rng_range(2, 4));
          dset(p, 25, 3, 1);
          act_set(p, AC_ATTACKING, 2, OBS_STARTED);
        } else if (eff == E_Rage) {
          vflag_set(so, V_RAGE);
        }
        break;
      }
    }
    do_move(p);
  }

  // returns true if residual damage applies afterwards; err set on division by zero
  __device__ bool execute_move(int p, uint32_t choice, bool &err) {
    const int so = p * SIDE_SZ;
    const uint32_t type = choice & 3;
    if (type == C_SWITCH) { switch_in(p, (int)(choice >> 2)); return false; }
    if (type == C_PASS) return false;
    uint32_t mslot = choice >> 2;
    const uint32_t sel = r8(so + O_LAST_SEL);
    if (sel == M_Struggle) mslot = 0;
    else if (mslot == 0) mslot = r8(B_LAST_MOVES + 2 * p);
    int r = before_move(p);
    if (r == BM_ERR) { err = true; return true; }
    if (r == BM_DONE) return true;
    execute_selected(p, mslot, r == BM_SKIP_CAN, r == BM_SKIP_PP);
    return true;
  }

  __device__ void handle_residual(int p) {
    const int so = p * SIDE_SZ, fo = (p ^ 1) * SIDE_SZ;
    const int sp = stored_off(so), fp = stored_off(fo);
    uint32_t hp = r16(sp + P_HP);
    if (hp == 0) return;
    const uint32_t maxhp = r16(sp + P_HP_MAX);
    const uint32_t status = r8(sp + P_STATUS);
    const uint32_t v = vlo(so);
    if (status & (ST_BRN | ST_PSN)) {
      uint32_t dmg = maxhp / 16; if (dmg < 1) dmg = 1;
      if (v & V_TOXIC) { uint32_t t = (toxic_ctr(so) + 1) & 31; set_toxic_ctr(so, t); dmg *= t; }
      hp = dmg > hp ? 0 : hp - dmg;
      w16(sp + P_HP, hp);
      if (hp == 0) return;
    }
    if (v & V_LEECHSEED) {
      uint32_t dmg = maxhp / 16; if (dmg < 1) dmg = 1;
      if (v & V_TOXIC) { uint32_t t = (toxic_ctr(so) + 1) & 31; set_toxic_ctr(so, t); dmg *= t; }
      hp = dmg > hp ? 0 : hp - dmg;
      w16(sp + P_HP, hp);
      uint32_t fhp = r16(fp + P_HP);
      if (fhp > 0) {
        uint32_t h = fhp + dmg, fmax = r16(fp + P_HP_MAX);
        w16(fp + P_HP, h > fmax ? fmax : h);
      }
    }
  }

  __device__ void faint(int p) {
    const int so = p * SIDE_SZ, fo = (p ^ 1) * SIDE_SZ;
    uint32_t fv = vlo(fo) & ~V_MULTIHIT;
    set_vlo(fo, fv);
    if (fv & V_BIDE) set_vstate(fo, 0);
    set_vlo(so, 0);
    set_vhi(so, 0);
    w8(so + O_LAST_USED, 0);
    w8(stored_off(so) + P_STATUS, 0);
    clear_binding(p ^ 1);
  }
  __device__ uint32_t check_faint(int p) {
    const int so = p * SIDE_SZ, fo = (p ^ 1) * SIDE_SZ;
    if (r16(stored_off(so) + P_HP) > 0) return 0;
    const bool foe_fainted = r16(stored_off(fo) + P_HP) == 0;
    faint(p);
    if (foe_fainted) faint(p ^ 1);
    const bool player_out = !any_alive(so), foe_out = !any_alive(fo);
    if (player_out && foe_out) return mk_result(R_TIE, 0, 0);
    if (player_out) return mk_result(p == 0 ? R_LOSE : R_WIN, 0, 0);
    if (foe_out) return mk_result(p == 0 ? R_WIN : R_LOSE, 0, 0);
    const uint32_t fc = foe_fainted ? C_SWITCH : C_PASS;
    return p == 0 ? mk_result(0, C_SWITCH, fc) : mk_result(0, fc, C_SWITCH);
  }
  __device__ __forceinline__ uint32_t end_turn() {
    uint32_t t = r16(B_TURN) + 1;
    w16(B_TURN, t);
    if (t >= 1000) return mk_result(R_TIE, 0, 0);
    return mk_result(0, C_MOVE, C_MOVE);
  }

  // ---- pkmn_gen1_battle_update ---------------------------------------------------------
  __device__ uint32_t update(uint32_t c1, uint32_t c2) {
    if constexpr (TRACK_ACTIONS) { act0 = 0; act1 = 0; }
    if (r16(B_TURN) == 0) {
      const bool a1 = any_alive(0), a2 = any_alive(SIDE_SZ);
      if (!a1) return mk_result(a2 ? R_LOSE : R_TIE, 0, 0);
      if (!a2) return mk_result(R_WIN, 0, 0);
      switch_in(0, 1);
      switch_in(1, 1);
      return end_turn();
    }
    select_move(0, c1);
    select_move(1, c2);
    int p = turn_order(c1, c2);
    uint32_t pc = p == 0 ? c1 : c2, qc = p == 0 ? c2 : c1;
    // first mover, then second mover: one loop body so the move code exists once in the binary
#pragma unroll 1
    for (int k = 0; k < 2; ++k) {
      const int q = p ^ 1;
      bool err = false;
      const bool replace = r16(stored_off(p * SIDE_SZ) + P_HP) == 0;
      const bool residual = execute_move(p, pc, err);
      if (err) return mk_result(R_ERROR, 0, 0);
      if (!replace) {
        uint32_t r;
        if ((pc & 3) != C_SWITCH) { r = check_faint(q); if (r) return r; }
        if (residual) handle_residual(p);
        r = check_faint(p);
        if (r) return r;
      }
      if ((qc & 3) == C_PASS) break;
      p = q;
      uint32_t t = pc; pc = qc; qc = t;
    }
    for (int s = 0; s < 2; ++s) {
      uint32_t v = vlo(s * SIDE_SZ);
      if ((v & V_BINDING) && ((v >> 21) & 7) == 0) clear_binding(s);
    }
    return end_turn();
  }

  // ---- pkmn_gen1_battle_choices: returns count, choices packed one byte each into out[] --
  // (out as 9 bytes in three dwords to stay in registers)
  struct Choices { uint32_t n; uint32_t w0, w1, w2;
    __device__ __forceinline__ void push(uint32_t c) {
      if (n < 4) w0 |= c << (8 * n); else if (n < 8) w1 |= c << (8 * (n - 4)); else w2 |= c;
      ++n;
    }
    __device__ __forceinline__ uint32_t get(uint32_t i) const {
      uint32_t w = i < 4 ? w0 : i < 8 ? w1 : w2;
      return (w >> (8 * (i & 3))) & 0xFF;
    }
  };
  __device__ __forceinline__ Choices choices(int p, uint32_t request) const {
    Choices c{0, 0, 0, 0};
    const int so = p * SIDE_SZ;
    if (request == C_PASS) { c.push(0); return c; }
    if (request == C_SWITCH) {
      for (int slot = 2; slot <= 6; ++slot) {
        uint32_t id = r8(so + O_ORDER + slot - 1);
        if (id == 0 || r16(so + PK_SZ * ((int)id - 1) + P_HP) == 0) continue;
        c.push((uint32_t)(slot << 2) | C_SWITCH);
      }
      if (c.n == 0) c.push(0);
      return c;
    }
    const uint32_t v = vlo(so);
    if (v & (V_RECHARGING | V_RAGE | V_THRASHING | V_CHARGING)) { c.push(C_MOVE); return c; }
    const uint32_t m01 = r32(so + O_ACTIVE + A_MOVES), m23 = r32(so + O_ACTIVE + A_MOVES + 4);
    if (v & (V_BIDE | V_BINDING)) {
      const uint32_t sel = r8(so + O_LAST_SEL);
      for (int i = 0; i < 4; ++i) {
        uint32_t ms = ((i < 2 ? m01 : m23) >> (16 * (i & 1))) & 0xFFFF;
        if ((ms & 0xFF) && (ms & 0xFF) == sel) { c.push((uint32_t)((i + 1) << 2) | C_MOVE); return c; }
      }
      c.push(C_MOVE);
      return c;
    }
    for (int slot = 2; slot <= 6; ++slot) {
      uint32_t id = r8(so + O_ORDER + slot - 1);
      if (id == 0 || r16(so + PK_SZ * ((int)id - 1) + P_HP) == 0) continue;
      c.push((uint32_t)(slot << 2) | C_SWITCH);
    }
    const uint32_t before = c.n;
    const uint32_t dm = (vhi(so) >> 24) & 7;
    for (int i = 0; i < 4; ++i) {
      uint32_t ms = ((i < 2 ? m01 : m23) >> (16 * (i & 1))) & 0xFFFF;
      if ((ms & 0xFF) == 0) break;
      if ((ms >> 8) == 0) continue;
      if (dm == (uint32_t)(i + 1)) continue;
      c.push((uint32_t)((i + 1) << 2) | C_MOVE);
    }
    if (c.n == before) c.push(C_MOVE); // Struggle
    return c;
  }
};

} // namespace oak
