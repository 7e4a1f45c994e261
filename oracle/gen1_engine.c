/* oracle/gen1_engine.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.  See oracle/oracle.h.
 *
 * Restatement of pkmn_gen1_battle_update / pkmn_gen1_battle_choices (the libpkmn calls
 * at /root/reference/cpp/include/search/mcts.h:161-166,337-350,453-479) for the build
 * configuration of /root/reference/dev/libpkmn:9:
 *   showdown       -> Pokemon-Showdown gen-1 semantics, 64-bit LCG of libpkmn/rng.h:9-11,
 *                     every roll = (top 32 bits of the new seed) scaled into a range
 *   miss=false     -> a move whose final accuracy byte is 255 never rolls / never misses
 *   advance=false  -> no RNG frame advances beyond the rolls themselves
 *   ebc=false      -> no endless-battle clause; turn >= 1000 is a tie
 *   key=true       -> hidden rolled durations are masked out of the chance actions
 *   chance, calc   -> durations / actions tracking, damage-roll overrides
 * The library source is absent from the reference checkout (parity unpinned, see
 * oracle.h); state layout follows cpp/include/libpkmn/layout.h and data.h exactly.
 *
 * RNG call order per executed damaging move (documented contract, mirrored by the HIP
 * kernel):  [speed tie] -> [confusion self-hit] -> [full paralysis] -> [thrash/bide
 * duration] -> [metronome] -> accuracy -> [multi-hit count] -> critical hit -> damage roll ->
 * [binding count] -> [secondary-effect chance] -> [secondary duration].  (Rounds 1-3 rolled the
 * multi-hit count behind the damage roll: OAK_MULTIHIT_ROLL_FIRST=0.)
 *
 * RESTATEMENT CHOICES THAT NOTHING IN THE REFERENCE CAN CONFIRM (libpkmn's source is absent; these follow the
 * author's reading of the published gen-1 / Pokemon-Showdown mechanics; each lists what the alternative under
 * pkmn/engine's published `-Dshowdown` code path would be, should a maintainer hold the real library against it):
 *  1. Multi-hit count (DoubleSlap, PinMissile, ... `EFF_MultiHit`).  SINCE ROUND 4 rolled right after the accuracy check and
 *     BEFORE crit / damage -- Pokemon Showdown's gen-1 tryMoveHit samples `hits` and only then calls moveHit, which rolls crit
 *     and the damage factor, and pkmn/engine's -Dshowdown path exists to reproduce Showdown's RNG stream.  Rounds 1-3 rolled
 *     it AFTER the damage roll (OAK_MULTIHIT_ROLL_FIRST=0): same number of LCG draws per move and the same distributions, but
 *     every multi-hit move's (hits, crit, damage) triple differs between the two orders.
 *  2. Counter deals `last_damage * 2`, gated by the foe's `last_moves[].counterable` byte (set for Normal / Fighting
 *     moves other than Counter) and a non-zero last_damage.  Alternative (Showdown's gen-1 Counter): gated on the foe's
 *     last SELECTED move's type and on `last_damage` of either side, with the Desync-Clause failures of the cartridge
 *     corner cases (Counter after a switch, after a multi-turn move's second turn); the alternative changes WHEN Counter
 *     fails, never its damage.
 *  3. Psywave.  SINCE ROUND 4 Showdown's rule: `random(0, level * 3 / 2)`, and a 0 FAILS the move (gen-1 moves.ts psywave:
 *     "Desync Clause Mod activated!"); the chance action holds the roll + 1 (0 = no Psywave this turn).  Rounds 1-3 drew
 *     `rng_range(1, max)` -- uniform on 1 .. max - 1, never failing (OAK_PSYWAVE_SHOWDOWN=0).  One draw either way, so the LCG
 *     stream stays aligned; 1 in `max` Psywaves (1 / 150 at level 100) now does nothing and the others map the draw differently.
 *  4. Accuracy (`move_hit`) is rolled BEFORE the critical-hit and damage rolls for every damaging move (Showdown order:
 *     immunity, accuracy, then damage).  Alternative (cartridge order, which pkmn/engine uses without -Dshowdown): crit and
 *     damage first, accuracy last -- a miss would then have consumed two more draws.
 * What constrains them today: the 13 known answers of cpp/src/search-test.cc:50-109 (sleep / confusion only) and the
 * TUTORIAL.md:46-70 search result (Body Slam / Psychic / Thunder Wave / Rest / Recover: paralysis, secondary chances,
 * crits, damage rolls, speed order; reproduced within tolerance by tests/test_gpu_search.py) -- neither exercises a
 * multi-hit move, Counter or Psywave, and a 1000-fold-averaged value does not see roll ORDER.  Parity at the libpkmn
 * boundary stays UNPINNED.
 */
#include "oracle.h"
#include "gen1_tables.h"
#include <string.h>

/* The two roll-order choices above that pkmn/engine's -Dshowdown path settles by following Pokemon Showdown's own code (the
 * build of /root/reference/dev/libpkmn:9 IS -Dshowdown), as compile-time switches shared -- by name and default -- with the two
 * HIP engines (oak_amd/csrc/gen1_device.hpp); 0 restores rounds 1-3's behaviour in all three:
 *   OAK_MULTIHIT_ROLL_FIRST  the multi-hit count is rolled behind the accuracy check, BEFORE crit / damage (Showdown
 *                            data/mods/gen1/scripts.ts tryMoveHit: `hits = this.battle.sample([2,2,2,3,3,3,4,5])`, then moveHit);
 *   OAK_PSYWAVE_SHOWDOWN     Psywave draws random(0, level * 3 / 2) and FAILS on 0 (data/mods/gen1/moves.ts psywave: "Desync
 *                            Clause Mod activated!"); the chance action holds the roll + 1. */
#ifndef OAK_MULTIHIT_ROLL_FIRST
#define OAK_MULTIHIT_ROLL_FIRST 1
#endif
#ifndef OAK_PSYWAVE_SHOWDOWN
#define OAK_PSYWAVE_SHOWDOWN 1
#endif
/* ... and the other two (round 5), default = what rounds 1-4 did, alternatives for a future libpkmn fixture to select:
 *   OAK_COUNTER_SHOWDOWN     Counter gated like Pokemon Showdown's gen-1 Counter (data/mods/gen1/moves.ts counter.damageCallback): it
 *                            hits iff the target side's last USED move and its last SELECTED move are both counterable -- base power
 *                            > 0, Normal or Fighting, not Counter -- and last_damage > 0 ("Desync Clause Mod" fails the mixed case);
 *                            0: gated by the `counterable` byte the target's last damaging HIT left in last_moves[];
 *   OAK_ACCURACY_LAST        the cartridge's roll order for ordinary damaging moves: critical hit, damage roll, THEN accuracy (a miss
 *                            has consumed two more draws), multi-hit count behind it; 0: accuracy first (Showdown's order).  Needs
 *                            OAK_MULTIHIT_ROLL_FIRST = 0. */
#ifndef OAK_COUNTER_SHOWDOWN
#define OAK_COUNTER_SHOWDOWN 0
#endif
#ifndef OAK_ACCURACY_LAST
#define OAK_ACCURACY_LAST 0
#endif
#if OAK_ACCURACY_LAST && OAK_MULTIHIT_ROLL_FIRST
#error "OAK_ACCURACY_LAST rolls the multi-hit count behind the accuracy check, i.e. behind crit / damage: build with -DOAK_MULTIHIT_ROLL_FIRST=0"
#endif

#pragma pack(push, 1)
typedef struct { uint16_t hp, atk, def, spe, spc; } Stats;
typedef struct { uint8_t id, pp; } MoveSlot;
typedef struct {
  Stats stats;
  MoveSlot moves[4];
  uint16_t hp;
  uint8_t status, species, types, level;
} Pokemon; /* 24 B, layout.h:33-41 */
typedef struct {
  Stats stats;
  uint8_t species, types;
  uint8_t boosts[4];
  uint64_t vol;
  MoveSlot moves[4];
} Active; /* 32 B, layout.h:43-50 */
typedef struct {
  Pokemon pokemon[6];
  Active active;
  uint8_t order[6];
  uint8_t last_selected_move, last_used_move;
} Side; /* 184 B */
typedef struct {
  Side sides[2];
  uint16_t turn, last_damage;
  struct { uint8_t index, counterable; } last_moves[2];
  uint64_t rng;
} Battle; /* 384 B */
#pragma pack(pop)
typedef char assert_battle_size[(sizeof(Battle) == 384) ? 1 : -1];

/* volatiles bits, layout.h:69-96 */
#define V_BIDE (1ull << 0)
#define V_THRASHING (1ull << 1)
#define V_MULTIHIT (1ull << 2)
#define V_FLINCH (1ull << 3)
#define V_CHARGING (1ull << 4)
#define V_BINDING (1ull << 5)
#define V_INVULNERABLE (1ull << 6)
#define V_CONFUSION (1ull << 7)
#define V_MIST (1ull << 8)
#define V_FOCUSENERGY (1ull << 9)
#define V_SUBSTITUTE (1ull << 10)
#define V_RECHARGING (1ull << 11)
#define V_RAGE (1ull << 12)
#define V_LEECHSEED (1ull << 13)
#define V_TOXIC (1ull << 14)
#define V_LIGHTSCREEN (1ull << 15)
#define V_REFLECT (1ull << 16)
#define V_TRANSFORM (1ull << 17)
#define VF_GET(v, sh, bits) ((uint32_t)(((v) >> (sh)) & ((1ull << (bits)) - 1)))
#define VF_SET(v, sh, bits, x) ((v) = ((v) & ~((((1ull << (bits)) - 1)) << (sh))) | (((uint64_t)(x) & ((1ull << (bits)) - 1)) << (sh)))
#define CONF_LEFT(v) VF_GET(v, 18, 3)
#define SET_CONF_LEFT(v, x) VF_SET(v, 18, 3, x)
#define ATTACKS(v) VF_GET(v, 21, 3)
#define SET_ATTACKS(v, x) VF_SET(v, 21, 3, x)
#define VSTATE(v) VF_GET(v, 24, 16)
#define SET_VSTATE(v, x) VF_SET(v, 24, 16, x)
#define SUB_HP(v) VF_GET(v, 40, 8)
#define SET_SUB_HP(v, x) VF_SET(v, 40, 8, x)
#define TRANSFORM_ID(v) VF_GET(v, 48, 4)
#define SET_TRANSFORM_ID(v, x) VF_SET(v, 48, 4, x)
#define DISABLE_LEFT(v) VF_GET(v, 52, 4)
#define SET_DISABLE_LEFT(v, x) VF_SET(v, 52, 4, x)
#define DISABLE_MOVE(v) VF_GET(v, 56, 3)
#define SET_DISABLE_MOVE(v, x) VF_SET(v, 56, 3, x)
#define TOXIC_CTR(v) VF_GET(v, 59, 5)
#define SET_TOXIC_CTR(v, x) VF_SET(v, 59, 5, x)

/* status byte, data/status.h:10-28 */
#define ST_SLP_MASK 7
#define ST_PSN 0x08
#define ST_BRN 0x10
#define ST_FRZ 0x20
#define ST_PAR 0x40
#define ST_EXT 0x80
#define ST_TOX 0x88

/* durations u32 per side, layout.h:119-125 */
#define D_SLEEP(d, slot) (((d) >> (3 * (slot))) & 7u)
#define D_SET_SLEEP(d, slot, x) ((d) = ((d) & ~(7u << (3 * (slot)))) | (((uint32_t)(x) & 7u) << (3 * (slot))))
#define D_GET(d, sh, bits) (((d) >> (sh)) & ((1u << (bits)) - 1))
#define D_SET(d, sh, bits, x) ((d) = ((d) & ~(((1u << (bits)) - 1) << (sh))) | (((uint32_t)(x) & ((1u << (bits)) - 1)) << (sh)))
#define D_CONFUSION 18, 3
#define D_DISABLE 21, 4
#define D_ATTACKING 25, 3
#define D_BINDING 28, 3

/* chance action bit offsets, layout.h:98-117 */
enum { A_DAMAGE = 0, A_HIT = 8, A_CRIT = 10, A_SECONDARY = 12, A_SPEEDTIE = 14, A_CONFUSED = 16,
       A_PARALYZED = 18, A_DURATION = 20, A_SLEEP = 24, A_CONFUSION = 26, A_DISABLE = 29,
       A_ATTACKING = 31, A_BINDING = 33, A_MOVESLOT = 36, A_PP = 40, A_MULTIHIT = 44,
       A_PSYWAVE = 48, A_METRONOME = 56 };
enum { OBS_NONE = 0, OBS_STARTED = 1, OBS_CONTINUING = 2, OBS_ENDED = 3 };

typedef struct {
  Battle *b;
  uint64_t act[2];
  uint32_t dur[2];
  const uint8_t *over; /* 16 bytes */
} Ctx;

static inline void act_set(Ctx *c, int p, int sh, int bits, uint32_t v) {
  c->act[p] = (c->act[p] & ~((((1ull << bits) - 1)) << sh)) | (((uint64_t)v & ((1ull << bits) - 1)) << sh);
}
static inline void act_bool(Ctx *c, int p, int sh, int v) { act_set(c, p, sh, 2, v ? 2 : 1); }

/* ---- RNG (showdown PSRNG over the gen5/6 LCG) ---------------------------------- */
static inline uint32_t rng_next(Battle *b) {
  b->rng = 0x5D588B656C078965ull * b->rng + 0x0000000000269EC3ull;
  return (uint32_t)(b->rng >> 32);
}
static inline uint32_t rng_range(Battle *b, uint32_t from, uint32_t to) {
  return from + (uint32_t)(((uint64_t)rng_next(b) * (uint64_t)(to - from)) >> 32);
}
static inline int rng_chance(Battle *b, uint32_t num, uint32_t den) { return rng_range(b, 0, den) < num; }

/* ---- small helpers ------------------------------------------------------------- */
static inline Pokemon *stored(Side *s) { return &s->pokemon[s->order[0] - 1]; }
static inline int has_type(uint8_t types, uint8_t t) { return (types & 15) == t || (types >> 4) == t; }
static inline int boost_get(const Active *a, int idx) { /* 0 atk 1 def 2 spe 3 spc 4 acc 5 eva */
  uint8_t n = (a->boosts[idx >> 1] >> ((idx & 1) * 4)) & 15;
  return (int)((n ^ 8) - 8);
}
static inline void boost_set(Active *a, int idx, int v) {
  int sh = (idx & 1) * 4;
  a->boosts[idx >> 1] = (uint8_t)((a->boosts[idx >> 1] & ~(15 << sh)) | (((uint8_t)v & 15) << sh));
}
static inline uint16_t *stat_ptr(Stats *s, int idx) { /* 0 atk 1 def 2 spe 3 spc */
  return idx == 0 ? &s->atk : idx == 1 ? &s->def : idx == 2 ? &s->spe : &s->spc;
}
static void status_modify(uint8_t status, Stats *st) {
  if (status & ST_PAR) { st->spe = st->spe / 4; if (st->spe < 1) st->spe = 1; }
  else if (status & ST_BRN) { st->atk = st->atk / 2; if (st->atk < 1) st->atk = 1; }
}
static const Stats *unmodified_stats(Battle *b, int player) {
  Side *s = &b->sides[player];
  if (!(s->active.vol & V_TRANSFORM)) return &stored(s)->stats;
  uint32_t id = TRANSFORM_ID(s->active.vol);
  return &b->sides[id >> 3].pokemon[(id & 7) - 1].stats;
}
static int find_first_alive(const Side *s) {
  for (int i = 0; i < 6; ++i)
    if (s->pokemon[i].hp > 0) return s->order[i] ? s->order[i] : (i + 1);
  return 0;
}
static int any_alive(const Side *s) {
  for (int i = 0; i < 6; ++i) if (s->pokemon[i].hp > 0) return 1;
  return 0;
}
static inline uint8_t mk_result(int type, int p1, int p2) { return (uint8_t)(type | (p1 << 4) | (p2 << 6)); }

static void clear_binding(Ctx *c, int player) {
  c->b->sides[player].active.vol &= ~V_BINDING;
  D_SET(c->dur[player], 28, 3, 0);
}

/* ---- switching ----------------------------------------------------------------- */
static void switch_in(Ctx *c, int player, int slot, int initial) {
  Battle *b = c->b;
  Side *side = &b->sides[player], *foe = &b->sides[player ^ 1];
  (void)initial;
  /* Toxic reverts to regular poison when the afflicted Pokemon leaves the field. */
  Pokemon *out = stored(side);
  if (out->status == ST_TOX) out->status = ST_PSN;
  uint8_t t = side->order[0];
  side->order[0] = side->order[slot - 1];
  side->order[slot - 1] = t;
  /* durations: sleeps follow the party slots, active-only counters reset */
  uint32_t d = c->dur[player];
  uint32_t s0 = D_SLEEP(d, 0), sk = D_SLEEP(d, slot - 1);
  D_SET_SLEEP(d, 0, sk);
  D_SET_SLEEP(d, slot - 1, s0);
  d &= (1u << 18) - 1;
  c->dur[player] = d;
  Pokemon *in = stored(side);
  side->last_used_move = 0;
  foe->last_used_move = 0;
  Active *a = &side->active;
  a->stats = in->stats;
  a->species = in->species;
  a->types = in->types;
  memset(a->boosts, 0, 4);
  a->vol = 0;
  memcpy(a->moves, in->moves, 8);
  status_modify(in->status, &a->stats);
  clear_binding(c, player ^ 1);
}

/* ---- move selection ------------------------------------------------------------ */
static void save_move(Battle *b, int player, uint8_t data) {
  Side *s = &b->sides[player];
  if (data == 0) {
    s->last_selected_move = MV_Struggle;
  } else {
    s->last_selected_move = s->active.moves[data - 1].id;
  }
  b->last_moves[player].index = data;
}
static void select_move(Battle *b, int player, uint8_t choice) {
  if ((choice & 3) == ORACLE_PASS) return;
  Side *s = &b->sides[player];
  uint64_t *v = &s->active.vol;
  if (*v & V_RECHARGING) return;
  if (*v & V_RAGE) return;
  *v &= ~V_FLINCH;
  if (*v & (V_THRASHING | V_CHARGING)) return;
  if ((choice & 3) == ORACLE_SWITCH) return;
  if (*v & (V_BIDE | V_BINDING)) return;
  /* asleep / frozen / trapped: Showdown still records the selection */
  save_move(b, player, choice >> 2);
}

static int turn_order(Ctx *c, uint8_t c1, uint8_t c2) {
  Battle *b = c->b;
  int t1 = c1 & 3, t2 = c2 & 3;
  if (t1 == ORACLE_PASS) return 1;
  if (t2 == ORACLE_PASS) return 0;
  if ((t1 == ORACLE_SWITCH) != (t2 == ORACLE_SWITCH)) return t1 == ORACLE_SWITCH ? 0 : 1;
  if (t1 == ORACLE_MOVE) {
    uint8_t m1 = b->sides[0].last_selected_move, m2 = b->sides[1].last_selected_move;
    if ((m1 == MV_QuickAttack) != (m2 == MV_QuickAttack)) return m1 == MV_QuickAttack ? 0 : 1;
    if ((m1 == MV_Counter) != (m2 == MV_Counter)) return m1 == MV_Counter ? 1 : 0;
  }
  uint16_t s1 = b->sides[0].active.stats.spe, s2 = b->sides[1].active.stats.spe;
  if (s1 == s2) {
    int p1 = rng_range(b, 0, 2) == 0;
    act_set(c, 0, A_SPEEDTIE, 2, p1 ? 1 : 2);
    act_set(c, 1, A_SPEEDTIE, 2, p1 ? 1 : 2);
    return p1 ? 0 : 1;
  }
  return s1 > s2 ? 0 : 1;
}

/* ---- damage -------------------------------------------------------------------- */
static int check_crit(Ctx *c, int player, const oracle_move_t *mv) {
  Side *s = &c->b->sides[player];
  uint32_t chance = ORACLE_SPECIES[stored(s)->species].spe / 2;
  if (s->active.vol & V_FOCUSENERGY) chance = chance / 2;
  else { chance *= 2; if (chance > 255) chance = 255; }
  if (mv->effect == EFF_HighCritical) { chance *= 4; if (chance > 255) chance = 255; }
  else chance = chance / 2;
  int crit = rng_chance(c->b, chance, 256);
  act_bool(c, player, A_CRIT, crit);
  return crit;
}

/* base damage into battle.last_damage; returns 0 on division by zero (error) */
static int calc_damage(Ctx *c, int player, int target_player, const oracle_move_t *mv, int crit) {
  Battle *b = c->b;
  Side *s = &b->sides[player], *t = &b->sides[target_player];
  int special = mv->type >= 8;
  uint32_t atk, def;
  if (crit) {
    const Stats *su = unmodified_stats(b, player), *tu = unmodified_stats(b, target_player);
    atk = special ? su->spc : su->atk;
    def = special ? tu->spc : tu->def;
  } else {
    atk = special ? s->active.stats.spc : s->active.stats.atk;
    def = special ? (uint32_t)t->active.stats.spc * ((t->active.vol & V_LIGHTSCREEN) ? 2 : 1)
                  : (uint32_t)t->active.stats.def * ((t->active.vol & V_REFLECT) ? 2 : 1);
  }
  if (atk > 255 || def > 255) {
    atk = (atk / 4) & 255; if (atk < 1) atk = 1;
    def = (def / 4) & 255; if (def < 1) def = 1;
  }
  uint32_t lvl = (uint32_t)stored(s)->level * (crit ? 2 : 1);
  if (mv->effect == EFF_Explode) { def = def / 2; if (def < 1) def = 1; }
  if (def == 0) return 0;
  uint32_t d = (lvl * 2 / 5) + 2;
  d *= mv->bp;
  d *= atk;
  d /= def;
  d /= 50;
  if (d > 997) d = 997;
  d += 2;
  b->last_damage = (uint16_t)d;
  return 1;
}

/* STAB + type chart (Showdown order: type1 then type2).  returns effectiveness product x4 */
static uint32_t adjust_damage(Battle *b, int player, const oracle_move_t *mv) {
  Side *s = &b->sides[player], *f = &b->sides[player ^ 1];
  uint8_t t1 = f->active.types & 15, t2 = f->active.types >> 4;
  uint32_t d = b->last_damage;
  if (has_type(s->active.types, mv->type)) d = (d + d / 2) & 0xFFFF;
  uint32_t e1 = ORACLE_TYPE_CHART[mv->type][t1], e2 = ORACLE_TYPE_CHART[mv->type][t2];
  if (e1 != 2) d = (d * e1 / 2) & 0xFFFF;
  if (t1 != t2 && e2 != 2) d = (d * e2 / 2) & 0xFFFF;
  b->last_damage = (uint16_t)d;
  return e1 * (t1 != t2 ? e2 : 2);
}

static void randomize_damage(Ctx *c, int player) {
  Battle *b = c->b;
  if (b->last_damage <= 1) return;
  uint32_t roll = c->over[player * 8];
  if (roll == 0) roll = rng_range(b, 217, 256);
  act_set(c, player, A_DAMAGE, 8, roll);
  b->last_damage = (uint16_t)((uint32_t)b->last_damage * roll / 255);
}

/* apply battle.last_damage to target (through sub_player's substitute if present).
 * returns 1 if a substitute absorbed it and broke. *hit_sub set if any sub took the hit */
static int apply_damage(Ctx *c, int target_player, int sub_player, int *hit_sub) {
  Battle *b = c->b;
  Side *sub = &b->sides[sub_player];
  if (hit_sub) *hit_sub = 0;
  if (sub->active.vol & V_SUBSTITUTE) {
    if (hit_sub) *hit_sub = 1;
    uint32_t hp = SUB_HP(sub->active.vol);
    if (b->last_damage >= hp) {
      SET_SUB_HP(sub->active.vol, 0);
      sub->active.vol &= ~V_SUBSTITUTE;
      return 1;
    }
    SET_SUB_HP(sub->active.vol, hp - b->last_damage);
    return 0;
  }
  Pokemon *p = stored(&b->sides[target_player]);
  if (b->last_damage > p->hp) b->last_damage = p->hp;
  p->hp -= b->last_damage;
  return 0;
}

/* accuracy check; sets *immune for "does not affect" style failures */
static int move_hit(Ctx *c, int player, const oracle_move_t *mv, uint8_t move_id) {
  Battle *b = c->b;
  Side *s = &b->sides[player], *f = &b->sides[player ^ 1];
  int miss;
  if (mv->effect == EFF_Swift) return 1;
  if (f->active.vol & V_INVULNERABLE) { miss = 1; goto done; }
  if ((mv->effect == EFF_DrainHP || mv->effect == EFF_DreamEater) && (f->active.vol & V_SUBSTITUTE)) { miss = 1; goto done; }
  if (mv->effect >= EFF_AccuracyDown1 && mv->effect <= EFF_SpeedDown1 && (f->active.vol & V_MIST)) { miss = 1; goto done; }
  {
    uint32_t acc = mv->accuracy;
    int ab = boost_get(&s->active, 4), eb = boost_get(&f->active, 5);
    acc = acc * ORACLE_BOOSTS[ab + 6][0] / ORACLE_BOOSTS[ab + 6][1];
    acc = acc * ORACLE_BOOSTS[-eb + 6][0] / ORACLE_BOOSTS[-eb + 6][1];
    if (acc > 255) acc = 255;
    if (acc < 1) acc = 1;
    if (acc == 255) { miss = 0; } /* miss=false: the 1/256 miss is patched out, no roll */
    else {
      miss = !rng_chance(b, acc, 256);
      act_bool(c, player, A_HIT, !miss);
    }
  }
  (void)move_id;
done:
  if (!miss) return 1;
  b->last_damage = 0;
  clear_binding(c, player);
  return 0;
}

/* ---- stat stages --------------------------------------------------------------- */
static int boost_self(Ctx *c, int player, int idx, int n) {
  Battle *b = c->b;
  Side *s = &b->sides[player], *f = &b->sides[player ^ 1];
  int cur = boost_get(&s->active, idx);
  if (cur >= 6) return 0;
  int nv = cur + n; if (nv > 6) nv = 6;
  if (idx < 4) {
    uint16_t *st = stat_ptr(&s->active.stats, idx);
    if (*st == 999) return 0; /* already capped: stage change rolled back */
    boost_set(&s->active, idx, nv);
    uint32_t base = *stat_ptr((Stats *)unmodified_stats(b, player), idx);
    uint32_t x = base * ORACLE_BOOSTS[nv + 6][0] / ORACLE_BOOSTS[nv + 6][1];
    if (x > 999) x = 999;
    *st = (uint16_t)x;
  } else {
    boost_set(&s->active, idx, nv);
  }
  /* stat modification glitch: the opponent's PAR/BRN penalty is re-applied */
  status_modify(stored(f)->status, &f->active.stats);
  return 1;
}
static int unboost_foe(Ctx *c, int player, int idx, int n) {
  Battle *b = c->b;
  Side *f = &b->sides[player ^ 1];
  int cur = boost_get(&f->active, idx);
  if (cur <= -6) return 0;
  int nv = cur - n; if (nv < -6) nv = -6;
  if (idx < 4) {
    uint16_t *st = stat_ptr(&f->active.stats, idx);
    if (*st == 1) return 0;
    boost_set(&f->active, idx, nv);
    uint32_t base = *stat_ptr((Stats *)unmodified_stats(b, player ^ 1), idx);
    uint32_t x = base * ORACLE_BOOSTS[nv + 6][0] / ORACLE_BOOSTS[nv + 6][1];
    if (x < 1) x = 1;
    *st = (uint16_t)x;
  } else {
    boost_set(&f->active, idx, nv);
  }
  status_modify(stored(f)->status, &f->active.stats);
  return 1;
}

/* ---- effects that run instead of damage (Effect onBegin group, moves.h:202-218) -- */
static void clear_volatiles_haze(Ctx *c, int player) {
  Side *s = &c->b->sides[player];
  uint64_t *v = &s->active.vol;
  SET_DISABLE_MOVE(*v, 0); SET_DISABLE_LEFT(*v, 0);
  D_SET(c->dur[player], 21, 4, 0);
  if (*v & V_CONFUSION) { *v &= ~V_CONFUSION; SET_CONF_LEFT(*v, 0); D_SET(c->dur[player], 18, 3, 0); }
  *v &= ~(V_MIST | V_FOCUSENERGY | V_LEECHSEED | V_LIGHTSCREEN | V_REFLECT);
  if (*v & V_TOXIC) { *v &= ~V_TOXIC; SET_TOXIC_CTR(*v, 0); if (stored(s)->status == ST_TOX) stored(s)->status = ST_PSN; }
}

static void on_begin(Ctx *c, int player, const oracle_move_t *mv, uint8_t move_id, uint8_t mslot) {
  Battle *b = c->b;
  Side *s = &b->sides[player], *f = &b->sides[player ^ 1];
  Pokemon *sp = stored(s), *fp = stored(f);
  uint64_t *v = &s->active.vol, *fv = &f->active.vol;
  b->last_damage = 0;
  switch (mv->effect) {
  case EFF_Confusion:
    if (*fv & V_SUBSTITUTE) return;
    if (!move_hit(c, player, mv, move_id)) return;
    if (*fv & V_CONFUSION) return;
    *fv |= V_CONFUSION;
    SET_CONF_LEFT(*fv, rng_range(b, 2, 6));
    D_SET(c->dur[player ^ 1], 18, 3, 1);
    act_set(c, player ^ 1, A_CONFUSION, 3, OBS_STARTED);
    return;
  case EFF_Conversion:
    if (*fv & V_INVULNERABLE) return;
    s->active.types = f->active.types;
    return;
  case EFF_FocusEnergy:
    *v |= V_FOCUSENERGY;
    return;
  case EFF_Haze: {
    memset(s->active.boosts, 0, 4);
    memset(f->active.boosts, 0, 4);
    s->active.stats = *unmodified_stats(b, player);
    f->active.stats = *unmodified_stats(b, player ^ 1);
    if (fp->status) {
      if (fp->status & ST_SLP_MASK) D_SET_SLEEP(c->dur[player ^ 1], 0, 0);
      fp->status = 0;
    }
    if (sp->status == ST_TOX) sp->status = ST_PSN;
    clear_volatiles_haze(c, player);
    clear_volatiles_haze(c, player ^ 1);
    return;
  }
  case EFF_Heal: {
    uint32_t delta = sp->stats.hp - sp->hp;
    if (delta == 0 || (delta & 255) == 255) return; /* gen-1 recovery failure glitch */
    if (move_id == MV_Rest) {
      sp->status = ST_EXT | 2;
      D_SET_SLEEP(c->dur[player], 0, 0);
      sp->hp = sp->stats.hp;
      *v &= ~V_TOXIC; SET_TOXIC_CTR(*v, 0);
    } else {
      uint32_t h = sp->hp + sp->stats.hp / 2;
      sp->hp = (uint16_t)(h > sp->stats.hp ? sp->stats.hp : h);
    }
    return;
  }
  case EFF_LeechSeed:
    if (has_type(f->active.types, TY_Grass)) return;
    if (!move_hit(c, player, mv, move_id)) return;
    if (*fv & V_LEECHSEED) return;
    *fv |= V_LEECHSEED;
    return;
  case EFF_LightScreen: *v |= V_LIGHTSCREEN; return;
  case EFF_Reflect: *v |= V_REFLECT; return;
  case EFF_Mist: *v |= V_MIST; return;
  case EFF_Mimic: {
    if (!move_hit(c, player, mv, move_id)) return;
    int n = 0;
    for (int i = 0; i < 4; ++i) if (f->active.moves[i].id) ++n;
    if (n == 0 || mslot == 0) return;
    uint32_t r = rng_range(b, 0, (uint32_t)n);
    act_set(c, player, A_MOVESLOT, 4, r + 1);
    s->active.moves[mslot - 1].id = f->active.moves[r].id;
    return;
  }
  case EFF_Paralyze:
    if (fp->status) return;
    if (ORACLE_TYPE_CHART[mv->type][f->active.types & 15] == 0 || ORACLE_TYPE_CHART[mv->type][f->active.types >> 4] == 0) return;
    if (!move_hit(c, player, mv, move_id)) return;
    fp->status = ST_PAR;
    f->active.stats.spe = f->active.stats.spe / 4; if (f->active.stats.spe < 1) f->active.stats.spe = 1;
    return;
  case EFF_Poison:
    if (fp->status) return;
    if (has_type(f->active.types, TY_Poison)) return;
    if (*fv & V_SUBSTITUTE) return;
    if (!move_hit(c, player, mv, move_id)) return;
    if (move_id == MV_Toxic) { fp->status = ST_TOX; *fv |= V_TOXIC; SET_TOXIC_CTR(*fv, 0); }
    else fp->status = ST_PSN;
    return;
  case EFF_Splash: return;
  case EFF_Substitute: {
    if (*v & V_SUBSTITUTE) return;
    uint32_t cost = sp->stats.hp / 4;
    if (sp->hp < cost) return;
    sp->hp -= (uint16_t)cost; /* exactly a quarter left: the user faints (gen-1 behaviour) */
    SET_SUB_HP(*v, cost + 1);
    *v |= V_SUBSTITUTE;
    return;
  }
  case EFF_SwitchAndTeleport:
    if (move_id != MV_Teleport) (void)move_hit(c, player, mv, move_id);
    return;
  case EFF_Transform: {
    if (*fv & V_INVULNERABLE) return;
    *v |= V_TRANSFORM;
    /* ident of the Pokemon copied: player bit << 3 | party id; chains resolve to the original */
    uint32_t id = (*fv & V_TRANSFORM) ? TRANSFORM_ID(*fv) : (uint32_t)(((player ^ 1) << 3) | f->order[0]);
    SET_TRANSFORM_ID(*v, id);
    s->active.species = f->active.species;
    s->active.types = f->active.types;
    s->active.stats = f->active.stats;
    memcpy(s->active.boosts, f->active.boosts, 4);
    for (int i = 0; i < 4; ++i) {
      s->active.moves[i].id = f->active.moves[i].id;
      s->active.moves[i].pp = f->active.moves[i].id ? 5 : 0;
    }
    return;
  }
  default: return;
  }
}

/* ---- pre-move checks ----------------------------------------------------------- */
enum { BM_OK = 0, BM_DONE = 1, BM_SKIP_CAN = 2, BM_SKIP_PP = 3, BM_ERR = 4 };

static int before_move(Ctx *c, int player) {
  Battle *b = c->b;
  Side *s = &b->sides[player], *f = &b->sides[player ^ 1];
  Pokemon *sp = stored(s);
  uint64_t *v = &s->active.vol;
  uint32_t *d = &c->dur[player];

  if (sp->status & ST_SLP_MASK) {
    sp->status -= 1;
    int left = sp->status & ST_SLP_MASK;
    if (!(sp->status & ST_EXT)) {
      if (left == 0) { D_SET_SLEEP(*d, 0, 0); act_set(c, player, A_SLEEP, 2, OBS_ENDED); }
      else { D_SET_SLEEP(*d, 0, D_SLEEP(*d, 0) + 1); act_set(c, player, A_SLEEP, 2, OBS_CONTINUING); }
    }
    if (left == 0) sp->status = 0;
    s->last_used_move = 0;
    return BM_DONE;
  }
  if (sp->status & ST_FRZ) { s->last_used_move = 0; return BM_DONE; }
  if (f->active.vol & V_BINDING) return BM_DONE;
  if (*v & V_FLINCH) { *v &= ~V_FLINCH; return BM_DONE; }
  if (*v & V_RECHARGING) { *v &= ~V_RECHARGING; return BM_DONE; }
  if (DISABLE_LEFT(*v) > 0) {
    uint32_t left = DISABLE_LEFT(*v) - 1;
    SET_DISABLE_LEFT(*v, left);
    if (left == 0) { SET_DISABLE_MOVE(*v, 0); D_SET(*d, 21, 4, 0); act_set(c, player, A_DISABLE, 2, OBS_ENDED); }
    else { D_SET(*d, 21, 4, D_GET(*d, 21, 4) + 1); act_set(c, player, A_DISABLE, 2, OBS_CONTINUING); }
  }
  if (*v & V_CONFUSION) {
    uint32_t left = CONF_LEFT(*v) - 1;
    SET_CONF_LEFT(*v, left);
    if (left == 0) {
      *v &= ~V_CONFUSION;
      D_SET(*d, 18, 3, 0);
      act_set(c, player, A_CONFUSION, 3, OBS_ENDED);
    } else {
      D_SET(*d, 18, 3, D_GET(*d, 18, 3) + 1);
      act_set(c, player, A_CONFUSION, 3, OBS_CONTINUING);
      int confused = !rng_chance(b, 128, 256);
      act_bool(c, player, A_CONFUSED, confused);
      if (confused) {
        *v &= ~(V_BIDE | V_THRASHING | V_MULTIHIT | V_FLINCH | V_CHARGING | V_BINDING | V_INVULNERABLE);
        D_SET(*d, 25, 3, 0); D_SET(*d, 28, 3, 0);
        static const oracle_move_t pound = {EFF_None, 40, TY_Normal, 255, 0, 0};
        if (!calc_damage(c, player, player, &pound, 0)) return BM_ERR;
        (void)apply_damage(c, player, player ^ 1, 0);
        return BM_DONE;
      }
    }
  }
  if (DISABLE_MOVE(*v) != 0 && s->last_selected_move != MV_Struggle &&
      s->active.moves[DISABLE_MOVE(*v) - 1].id == s->last_selected_move) {
    *v &= ~V_CHARGING;
    return BM_DONE;
  }
  if (sp->status & ST_PAR) {
    int par = rng_chance(b, 63, 256);
    act_bool(c, player, A_PARALYZED, par);
    if (par) {
      *v &= ~(V_BIDE | V_THRASHING | V_CHARGING | V_BINDING | V_INVULNERABLE);
      D_SET(*d, 25, 3, 0); D_SET(*d, 28, 3, 0);
      return BM_DONE;
    }
  }
  if (*v & V_BIDE) {
    uint32_t left = ATTACKS(*v) - 1;
    SET_ATTACKS(*v, left);
    if (left != 0) { D_SET(*d, 25, 3, D_GET(*d, 25, 3) + 1); act_set(c, player, A_ATTACKING, 2, OBS_CONTINUING); return BM_DONE; }
    D_SET(*d, 25, 3, 0); act_set(c, player, A_ATTACKING, 2, OBS_ENDED);
    *v &= ~V_BIDE;
    uint32_t dmg = (VSTATE(*v) * 2) & 0xFFFF;
    SET_VSTATE(*v, 0);
    b->last_damage = (uint16_t)dmg;
    if (dmg == 0) return BM_DONE;
    if (f->active.vol & V_INVULNERABLE) return BM_DONE;
    (void)apply_damage(c, player ^ 1, player ^ 1, 0);
    return BM_DONE;
  }
  if (*v & V_THRASHING) {
    uint32_t left = ATTACKS(*v) - 1;
    SET_ATTACKS(*v, left);
    if (left == 0) {
      *v &= ~V_THRASHING;
      D_SET(*d, 25, 3, 0); act_set(c, player, A_ATTACKING, 2, OBS_ENDED);
      *v |= V_CONFUSION;
      SET_CONF_LEFT(*v, rng_range(b, 2, 6));
      D_SET(*d, 18, 3, 1);
      act_set(c, player, A_CONFUSION, 3, OBS_STARTED);
    } else {
      D_SET(*d, 25, 3, D_GET(*d, 25, 3) + 1); act_set(c, player, A_ATTACKING, 2, OBS_CONTINUING);
    }
    return BM_SKIP_CAN;
  }
  if (*v & V_BINDING) {
    uint32_t left = ATTACKS(*v) - 1;
    SET_ATTACKS(*v, left);
    D_SET(*d, 28, 3, D_GET(*d, 28, 3) + 1); act_set(c, player, A_BINDING, 3, OBS_CONTINUING);
    if (b->last_damage != 0) (void)apply_damage(c, player ^ 1, player ^ 1, 0);
    return BM_DONE;
  }
  return (*v & V_RAGE) ? BM_SKIP_PP : BM_OK;
}

static void decrement_pp(Side *s, uint8_t mslot) {
  if (mslot == 0) return;
  Active *a = &s->active;
  a->moves[mslot - 1].pp = (uint8_t)((a->moves[mslot - 1].pp - 1) & 63);
  if (a->vol & V_TRANSFORM) return;
  Pokemon *p = stored(s);
  p->moves[mslot - 1].pp = (uint8_t)((p->moves[mslot - 1].pp - 1) & 63);
}

/* ---- the move itself ----------------------------------------------------------- */
static void secondary_status(Ctx *c, int player, const oracle_move_t *mv, uint8_t status, uint32_t num) {
  Battle *b = c->b;
  Side *f = &b->sides[player ^ 1];
  Pokemon *fp = stored(f);
  if (status == ST_BRN && (fp->status & ST_FRZ)) { fp->status = 0; return; } /* fire thaws */
  if (fp->status) return;
  if (has_type(f->active.types, status == ST_PSN ? TY_Poison : mv->type)) return;
  int proc = rng_chance(b, num, 256);
  act_bool(c, player, A_SECONDARY, proc);
  if (!proc) return;
  fp->status = status;
  if (status == ST_PAR) { f->active.stats.spe /= 4; if (f->active.stats.spe < 1) f->active.stats.spe = 1; }
  if (status == ST_BRN) { f->active.stats.atk /= 2; if (f->active.stats.atk < 1) f->active.stats.atk = 1; }
  if (status == ST_FRZ) { /* a frozen target stops whatever it was locked into */ }
}

static void rage_build(Ctx *c, int target_player) {
  Side *t = &c->b->sides[target_player];
  if ((t->active.vol & V_RAGE) && stored(t)->hp > 0) (void)boost_self(c, target_player, 0, 1);
}

static void do_move(Ctx *c, int player, uint8_t mslot) {
  Battle *b = c->b;
  Side *s = &b->sides[player], *f = &b->sides[player ^ 1];
  Pokemon *sp = stored(s), *fp = stored(f);
  uint64_t *v = &s->active.vol, *fv = &f->active.vol;
  uint8_t move_id = s->last_selected_move;
  const oracle_move_t *mv = &ORACLE_MOVES[move_id];
  (void)mslot;
  b->last_moves[player].counterable = 0;

  /* --- non-damaging moves resolved after the accuracy check (onEnd group + Disable) */
  if (mv->bp == 0) {
    b->last_damage = 0;
    switch (mv->effect) {
    case EFF_AttackUp1: boost_self(c, player, 0, 1); return;
    case EFF_AttackUp2: boost_self(c, player, 0, 2); return;
    case EFF_DefenseUp1: boost_self(c, player, 1, 1); return;
    case EFF_DefenseUp2: boost_self(c, player, 1, 2); return;
    case EFF_SpeedUp2: boost_self(c, player, 2, 2); return;
    case EFF_SpecialUp1: boost_self(c, player, 3, 1); return;
    case EFF_SpecialUp2: boost_self(c, player, 3, 2); return;
    case EFF_EvasionUp1: boost_self(c, player, 5, 1); return;
    case EFF_Bide:
      *v |= V_BIDE;
      SET_VSTATE(*v, 0);
      SET_ATTACKS(*v, rng_range(b, 2, 4));
      D_SET(c->dur[player], 25, 3, 1);
      act_set(c, player, A_ATTACKING, 2, OBS_STARTED);
      return;
    case EFF_AccuracyDown1: case EFF_AttackDown1: case EFF_DefenseDown1: case EFF_DefenseDown2: case EFF_SpeedDown1: {
      if (*fv & V_SUBSTITUTE) return;
      if (!move_hit(c, player, mv, move_id)) return;
      int idx = mv->effect == EFF_AccuracyDown1 ? 4 : mv->effect == EFF_AttackDown1 ? 0 : mv->effect == EFF_SpeedDown1 ? 2 : 1;
      unboost_foe(c, player, idx, mv->effect == EFF_DefenseDown2 ? 2 : 1);
      return;
    }
    case EFF_Sleep: {
      if (*fv & V_RECHARGING) {
        *fv &= ~V_RECHARGING; /* always lands on a recharging target */
        if (fp->status & ST_SLP_MASK) return;
      } else {
        if (fp->status) return;
        if (!move_hit(c, player, mv, move_id)) return;
      }
      uint32_t n = rng_range(b, 1, 8);
      fp->status = (uint8_t)n;
      D_SET_SLEEP(c->dur[player ^ 1], 0, 1);
      act_set(c, player ^ 1, A_SLEEP, 2, OBS_STARTED);
      return;
    }
    case EFF_Disable: {
      if (DISABLE_MOVE(*fv) != 0) return;
      if (!move_hit(c, player, mv, move_id)) return;
      int n = 0, slots[4];
      for (int i = 0; i < 4; ++i) if (f->active.moves[i].id && f->active.moves[i].pp > 0) slots[n++] = i + 1;
      if (n == 0) return;
      uint32_t r = rng_range(b, 0, (uint32_t)n);
      act_set(c, player, A_MOVESLOT, 4, (uint32_t)slots[r]);
      SET_DISABLE_MOVE(*fv, slots[r]);
      SET_DISABLE_LEFT(*fv, rng_range(b, 1, 9));
      D_SET(c->dur[player ^ 1], 21, 4, 1);
      act_set(c, player ^ 1, A_DISABLE, 2, OBS_STARTED);
      return;
    }
    default: return;
    }
  }

  /* --- damaging moves ---------------------------------------------------------- */
  int fixed = mv->effect == EFF_SpecialDamage || mv->effect == EFF_SuperFang || move_id == MV_Counter;
  int ohko = mv->effect == EFF_OHKO;
  uint8_t t1 = f->active.types & 15, t2 = f->active.types >> 4;
  int immune = 0;
  if (!fixed) immune = ORACLE_TYPE_CHART[mv->type][t1] == 0 || ORACLE_TYPE_CHART[mv->type][t2] == 0;
  if (mv->effect == EFF_DreamEater && !(fp->status & ST_SLP_MASK)) immune = 1;
  if (ohko && s->active.stats.spe < f->active.stats.spe) immune = 1;
  if (move_id == MV_Counter) {
#if OAK_COUNTER_SHOWDOWN
    const uint8_t lu = f->last_used_move, ls = f->last_selected_move;
    const oracle_move_t *mu = &ORACLE_MOVES[lu], *ms = &ORACLE_MOVES[ls];
    int cu = lu != 0 && lu != MV_Counter && mu->bp > 0 && (mu->type == TY_Normal || mu->type == TY_Fighting);
    int cs = ls != 0 && ls != MV_Counter && ms->bp > 0 && (ms->type == TY_Normal || ms->type == TY_Fighting);
    if (!(cu && cs) || b->last_damage == 0) immune = 1;
#else
    if (!b->last_moves[player ^ 1].counterable || b->last_damage == 0) immune = 1;
#endif
  }
  int hit = 0;
  const int late_hit = OAK_ACCURACY_LAST && !fixed && !ohko; /* cartridge order: crit and damage roll in front of the accuracy roll */
  if (!immune) hit = late_hit ? 1 : move_hit(c, player, mv, move_id);
  if (immune || !hit) {
    b->last_damage = 0;
    clear_binding(c, player);
    if (mv->effect == EFF_Explode) { sp->hp = 0; sp->status = 0; }
    if (mv->effect == EFF_JumpKick && !immune && sp->hp > 0) sp->hp -= 1; /* crash: 1 HP */
    return;
  }

  int crit = 0, hits = 1;
  if (fixed) {
    uint32_t d;
    if (move_id == MV_Counter) { d = (uint32_t)b->last_damage * 2; if (d > 65535) d = 65535; }
    else if (mv->effect == EFF_SuperFang) { d = fp->hp / 2; if (d < 1) d = 1; }
    else if (move_id == MV_SonicBoom) d = 20;
    else if (move_id == MV_DragonRage) d = 40;
    else if (move_id == MV_Psywave) {
      uint32_t max = (uint32_t)sp->level * 3 / 2;
#if OAK_PSYWAVE_SHOWDOWN
      /* Showdown: random(0, max); a 0 fails the move (Desync Clause Mod).  The action holds the roll + 1 (0 = no Psywave). */
      d = rng_range(b, 0, max ? max : 1);
      act_set(c, player, A_PSYWAVE, 8, d + 1);
      if (d == 0) { b->last_damage = 0; clear_binding(c, player); return; }
#else
      d = max <= 1 ? 1 : rng_range(b, 1, max);
      act_set(c, player, A_PSYWAVE, 8, d);
#endif
    } else d = sp->level; /* SeismicToss, NightShade */
    b->last_damage = (uint16_t)d;
  } else if (ohko) {
    b->last_damage = 65535;
  } else {
#if OAK_MULTIHIT_ROLL_FIRST
    if (mv->effect == EFF_MultiHit) { /* Showdown's gen-1 tryMoveHit samples the count right behind the accuracy check, before moveHit rolls crit / damage */
      static const uint8_t dist0[8] = {2, 2, 2, 3, 3, 3, 4, 5};
      hits = dist0[rng_range(b, 0, 8)];
      act_set(c, player, A_MULTIHIT, 4, (uint32_t)hits);
    }
#endif
    crit = check_crit(c, player, mv);
    if (!calc_damage(c, player, player ^ 1, mv, crit)) return;
    (void)adjust_damage(b, player, mv);
    randomize_damage(c, player);
    if (b->last_damage == 0) { clear_binding(c, player); return; } /* rounded down to nothing */
#if OAK_ACCURACY_LAST
    if (!move_hit(c, player, mv, move_id)) { /* (move_hit zeroes last_damage and clears the binding on a miss) */
      if (mv->effect == EFF_Explode) { sp->hp = 0; sp->status = 0; }
      if (mv->effect == EFF_JumpKick && sp->hp > 0) sp->hp -= 1;
      return;
    }
#endif
  }

  if (mv->effect == EFF_DoubleHit || mv->effect == EFF_Twineedle) hits = 2;
#if !OAK_MULTIHIT_ROLL_FIRST
  else if (mv->effect == EFF_MultiHit) {
    static const uint8_t dist[8] = {2, 2, 2, 3, 3, 3, 4, 5};
    hits = dist[rng_range(b, 0, 8)];
    act_set(c, player, A_MULTIHIT, 4, (uint32_t)hits);
  }
#endif

  int broke = 0, hit_sub = 0;
  uint32_t dealt = 0;
  uint16_t per_hit = b->last_damage;
  for (int h = 0; h < hits; ++h) {
    b->last_damage = per_hit;
    broke = apply_damage(c, player ^ 1, player ^ 1, &hit_sub);
    dealt = b->last_damage;
    if (!hit_sub) {
      if (*fv & V_BIDE) SET_VSTATE(*fv, (VSTATE(*fv) + dealt) & 0xFFFF);
      rage_build(c, player ^ 1);
    }
    if (broke || fp->hp == 0) break;
  }
  b->last_moves[player].counterable = (uint8_t)((mv->type == TY_Normal || mv->type == TY_Fighting) && move_id != MV_Counter);

  /* user-side consequences */
  if (mv->effect == EFF_Explode) { if (!broke) { sp->hp = 0; sp->status = 0; } }
  if (mv->effect == EFF_Recoil && !broke && dealt > 0) {
    uint32_t r = dealt / (move_id == MV_Struggle ? 2 : 4); if (r < 1) r = 1;
    sp->hp = (uint16_t)(r > sp->hp ? 0 : sp->hp - r);
  }
  if ((mv->effect == EFF_DrainHP || mv->effect == EFF_DreamEater) && dealt > 0) {
    uint32_t h = dealt / 2; if (h < 1) h = 1;
    h += sp->hp; if (h > sp->stats.hp) h = sp->stats.hp;
    sp->hp = (uint16_t)h;
  }
  if (fp->hp == 0 || broke) return; /* no secondary effects, no recharge, no binding */
  if (mv->effect == EFF_HyperBeam) { *v |= V_RECHARGING; return; }
  if (mv->effect == EFF_Binding) {
    if (!(*v & V_BINDING)) {
      static const uint8_t dist[8] = {2, 2, 2, 3, 3, 3, 4, 5};
      uint32_t n = dist[rng_range(b, 0, 8)];
      *v |= V_BINDING;
      SET_ATTACKS(*v, n - 1);
      D_SET(c->dur[player], 28, 3, 1);
      act_set(c, player, A_BINDING, 3, OBS_STARTED);
    }
    return;
  }
  if (hit_sub) return; /* a standing substitute blocks every secondary effect */
  switch (mv->effect) {
  case EFF_BurnChance1: secondary_status(c, player, mv, ST_BRN, 26); break;
  case EFF_BurnChance2: secondary_status(c, player, mv, ST_BRN, 77); break;
  case EFF_FreezeChance: secondary_status(c, player, mv, ST_FRZ, 26); break;
  case EFF_ParalyzeChance1: secondary_status(c, player, mv, ST_PAR, 26); break;
  case EFF_ParalyzeChance2: secondary_status(c, player, mv, ST_PAR, 77); break;
  case EFF_PoisonChance1: secondary_status(c, player, mv, ST_PSN, 52); break;
  case EFF_PoisonChance2: secondary_status(c, player, mv, ST_PSN, 103); break;
  case EFF_Twineedle: secondary_status(c, player, mv, ST_PSN, 52); break;
  case EFF_FlinchChance1: case EFF_FlinchChance2: {
    int proc = rng_chance(b, mv->effect == EFF_FlinchChance1 ? 26 : 77, 256);
    act_bool(c, player, A_SECONDARY, proc);
    if (proc) *fv |= V_FLINCH;
    break;
  }
  case EFF_ConfusionChance: {
    if (*fv & V_CONFUSION) break;
    int proc = rng_chance(b, 25, 256);
    act_bool(c, player, A_SECONDARY, proc);
    if (proc) {
      *fv |= V_CONFUSION;
      SET_CONF_LEFT(*fv, rng_range(b, 2, 6));
      D_SET(c->dur[player ^ 1], 18, 3, 1);
      act_set(c, player ^ 1, A_CONFUSION, 3, OBS_STARTED);
    }
    break;
  }
  case EFF_AttackDownChance: case EFF_DefenseDownChance: case EFF_SpeedDownChance: case EFF_SpecialDownChance: {
    int proc = rng_chance(b, 85, 256);
    act_bool(c, player, A_SECONDARY, proc);
    if (proc) unboost_foe(c, player, mv->effect - EFF_AttackDownChance, 1);
    break;
  }
  default: break;
  }
}

/* canMove: charge turns, PP, Metronome / Mirror Move redirection, onBegin effects. */
static void execute_selected(Ctx *c, int player, uint8_t mslot, int skip_can, int skip_pp) {
  Battle *b = c->b;
  Side *s = &b->sides[player], *f = &b->sides[player ^ 1];
  uint64_t *v = &s->active.vol;
  if (!skip_can) {
    for (int depth = 0; depth < 4; ++depth) {
      uint8_t move_id = s->last_selected_move;
      const oracle_move_t *mv = &ORACLE_MOVES[move_id];
      if (*v & V_CHARGING) {
        *v &= ~(V_CHARGING | V_INVULNERABLE);
      } else if (mv->effect == EFF_Charge) {
        *v |= V_CHARGING;
        if (move_id == MV_Fly || move_id == MV_Dig) *v |= V_INVULNERABLE;
        s->last_used_move = move_id;
        b->last_moves[player].counterable = 0;
        return;
      }
      s->last_used_move = move_id;
      b->last_moves[player].counterable = 0;
      if (!skip_pp) decrement_pp(s, mslot);
      skip_pp = 1;
      if (mv->effect == EFF_Metronome) {
        uint32_t r = rng_range(b, 0, 163);
        uint8_t pick = (uint8_t)(r + 1 >= MV_Metronome ? r + 2 : r + 1);
        act_set(c, player, A_METRONOME, 8, pick);
        s->last_selected_move = pick;
        continue;
      }
      if (mv->effect == EFF_MirrorMove) {
        uint8_t m = f->last_used_move;
        if (m == 0 || m == MV_MirrorMove) { b->last_damage = 0; return; }
        s->last_selected_move = m;
        continue;
      }
      if (mv->effect >= EFF_Confusion && mv->effect <= EFF_Transform) {
        on_begin(c, player, mv, move_id, mslot);
        return;
      }
      if (mv->effect == EFF_Thrashing) {
        *v |= V_THRASHING;
        SET_ATTACKS(*v, rng_range(b, 2, 4));
        D_SET(c->dur[player], 25, 3, 1);
        act_set(c, player, A_ATTACKING, 2, OBS_STARTED);
      } else if (mv->effect == EFF_Rage) {
        *v |= V_RAGE;
      }
      break;
    }
  }
  do_move(c, player, mslot);
}

/* returns 1 if `residual` handling applies afterwards */
static int execute_move(Ctx *c, int player, uint8_t choice, int *err) {
  Battle *b = c->b;
  Side *s = &b->sides[player];
  int type = choice & 3;
  if (type == ORACLE_SWITCH) { switch_in(c, player, choice >> 2, 0); return 0; }
  if (type == ORACLE_PASS) return 0;
  uint8_t mslot = choice >> 2;
  if (mslot == 0 && s->last_selected_move != MV_Struggle) mslot = b->last_moves[player].index;
  if (s->last_selected_move == MV_Struggle) mslot = 0;
  int r = before_move(c, player);
  if (r == BM_ERR) { *err = 1; return 1; }
  if (r == BM_DONE) return 1;
  execute_selected(c, player, mslot, r == BM_SKIP_CAN, r == BM_SKIP_PP);
  return 1;
}

static void handle_residual(Ctx *c, int player) {
  Battle *b = c->b;
  Side *s = &b->sides[player], *f = &b->sides[player ^ 1];
  Pokemon *sp = stored(s), *fp = stored(f);
  uint64_t *v = &s->active.vol;
  if (sp->hp == 0) return;
  if (sp->status & (ST_BRN | ST_PSN)) {
    uint32_t dmg = sp->stats.hp / 16; if (dmg < 1) dmg = 1;
    if (*v & V_TOXIC) { uint32_t t = (TOXIC_CTR(*v) + 1) & 31; SET_TOXIC_CTR(*v, t); dmg *= t; }
    sp->hp = (uint16_t)(dmg > sp->hp ? 0 : sp->hp - dmg);
    if (sp->hp == 0) return;
  }
  if (*v & V_LEECHSEED) {
    uint32_t dmg = sp->stats.hp / 16; if (dmg < 1) dmg = 1;
    if (*v & V_TOXIC) { uint32_t t = (TOXIC_CTR(*v) + 1) & 31; SET_TOXIC_CTR(*v, t); dmg *= t; }
    sp->hp = (uint16_t)(dmg > sp->hp ? 0 : sp->hp - dmg);
    if (fp->hp > 0) {
      uint32_t h = fp->hp + dmg; if (h > fp->stats.hp) h = fp->stats.hp;
      fp->hp = (uint16_t)h;
    }
  }
}

/* faint bookkeeping; returns a result byte or 0 when the turn goes on */
static void faint(Ctx *c, int player) {
  Side *s = &c->b->sides[player], *f = &c->b->sides[player ^ 1];
  f->active.vol &= ~V_MULTIHIT;
  if (f->active.vol & V_BIDE) SET_VSTATE(f->active.vol, 0);
  s->active.vol = 0;
  s->last_used_move = 0;
  stored(s)->status = 0;
  clear_binding(c, player ^ 1);
}
static uint8_t check_faint(Ctx *c, int player) {
  Battle *b = c->b;
  Side *s = &b->sides[player], *f = &b->sides[player ^ 1];
  if (stored(s)->hp > 0) return 0;
  int foe_fainted = stored(f)->hp == 0;
  faint(c, player);
  if (foe_fainted) faint(c, player ^ 1);
  int player_out = !any_alive(s), foe_out = !any_alive(f);
  if (player_out && foe_out) return mk_result(ORACLE_TIE, 0, 0);
  if (player_out) return mk_result(player == 0 ? ORACLE_LOSE : ORACLE_WIN, 0, 0);
  if (foe_out) return mk_result(player == 0 ? ORACLE_WIN : ORACLE_LOSE, 0, 0);
  int fc = foe_fainted ? ORACLE_SWITCH : ORACLE_PASS;
  return player == 0 ? mk_result(0, ORACLE_SWITCH, fc) : mk_result(0, fc, ORACLE_SWITCH);
}

static uint8_t end_turn(Battle *b) {
  b->turn += 1;
  if (b->turn >= 1000) return mk_result(ORACLE_TIE, 0, 0);
  return mk_result(0, ORACLE_MOVE, ORACLE_MOVE);
}

static uint8_t do_turn(Ctx *c, int p, uint8_t pc, int q, uint8_t qc) {
  Battle *b = c->b;
  int err = 0;
  uint8_t r;
  int replace = stored(&b->sides[p])->hp == 0;
  int residual = execute_move(c, p, pc, &err);
  if (err) return mk_result(ORACLE_ERROR, 0, 0);
  if (!replace) {
    if ((pc & 3) != ORACLE_SWITCH) { if ((r = check_faint(c, q))) return r; }
    if (residual) handle_residual(c, p);
    if ((r = check_faint(c, p))) return r;
  } else if ((qc & 3) == ORACLE_PASS) return 0;
  if ((qc & 3) == ORACLE_PASS) return 0;

  replace = stored(&b->sides[q])->hp == 0;
  residual = execute_move(c, q, qc, &err);
  if (err) return mk_result(ORACLE_ERROR, 0, 0);
  if (!replace) {
    if ((qc & 3) != ORACLE_SWITCH) { if ((r = check_faint(c, p))) return r; }
    if (residual) handle_residual(c, q);
    if ((r = check_faint(c, q))) return r;
  }
  return 0;
}

static uint8_t update_impl(Ctx *c, uint8_t c1, uint8_t c2) {
  Battle *b = c->b;
  if (b->turn == 0) {
    int s1 = find_first_alive(&b->sides[0]);
    if (s1 == 0) return mk_result(find_first_alive(&b->sides[1]) == 0 ? ORACLE_TIE : ORACLE_LOSE, 0, 0);
    int s2 = find_first_alive(&b->sides[1]);
    if (s2 == 0) return mk_result(ORACLE_WIN, 0, 0);
    switch_in(c, 0, 1, 1);
    switch_in(c, 1, 1, 1);
    return end_turn(b);
  }
  select_move(b, 0, c1);
  select_move(b, 1, c2);
  uint8_t r;
  if (turn_order(c, c1, c2) == 0) r = do_turn(c, 0, c1, 1, c2);
  else r = do_turn(c, 1, c2, 0, c1);
  if (r) return r;
  for (int p = 0; p < 2; ++p) {
    uint64_t v = b->sides[p].active.vol;
    if ((v & V_BINDING) && ATTACKS(v) == 0) clear_binding(c, p);
  }
  return end_turn(b);
}

/* ---- public API ---------------------------------------------------------------- */
void oracle_options_set(oracle_options *o, const uint8_t *durations8, const uint8_t *overrides16) {
  memset(o->actions, 0, 16);
  if (durations8) memcpy(o->durations, durations8, 8);
  if (overrides16) memcpy(o->overrides, overrides16, 16); else memset(o->overrides, 0, 16);
}

uint8_t oracle_update(uint8_t *battle384, uint8_t c1, uint8_t c2, oracle_options *o) {
  Ctx c;
  c.b = (Battle *)battle384;
  c.act[0] = c.act[1] = 0;
  memcpy(c.dur, o->durations, 8);
  c.over = o->overrides;
  uint8_t r = update_impl(&c, c1, c2);
  /* key=true: the rolled hidden durations never enter the key (we never record them) */
  memcpy(o->actions, c.act, 16);
  memcpy(o->durations, c.dur, 8);
  return r;
}

uint8_t oracle_choices(const uint8_t *battle384, int player, int request, uint8_t *out, size_t len) {
  const Battle *b = (const Battle *)battle384;
  const Side *s = &b->sides[player];
  uint8_t n = 0;
  if (len < ORACLE_MAX_CHOICES) return 0;
  if (request == ORACLE_PASS) { out[n++] = 0; return n; }
  if (request == ORACLE_SWITCH) {
    for (int slot = 2; slot <= 6; ++slot) {
      uint8_t id = s->order[slot - 1];
      if (id == 0 || s->pokemon[id - 1].hp == 0) continue;
      out[n++] = (uint8_t)((slot << 2) | ORACLE_SWITCH);
    }
    if (n == 0) out[n++] = 0;
    return n;
  }
  uint64_t v = s->active.vol;
  if (v & (V_RECHARGING | V_RAGE | V_THRASHING | V_CHARGING)) { out[n++] = ORACLE_MOVE; return n; }
  int limited = (v & (V_BIDE | V_BINDING)) != 0;
  if (!limited) {
    for (int slot = 2; slot <= 6; ++slot) {
      uint8_t id = s->order[slot - 1];
      if (id == 0 || s->pokemon[id - 1].hp == 0) continue;
      out[n++] = (uint8_t)((slot << 2) | ORACLE_SWITCH);
    }
  } else {
    for (int i = 0; i < 4; ++i)
      if (s->active.moves[i].id && s->active.moves[i].id == s->last_selected_move) {
        out[n++] = (uint8_t)(((i + 1) << 2) | ORACLE_MOVE);
        return n;
      }
    out[n++] = ORACLE_MOVE;
    return n;
  }
  uint8_t before = n;
  for (int i = 0; i < 4; ++i) {
    if (s->active.moves[i].id == 0) break;
    if (s->active.moves[i].pp == 0) continue;
    if (DISABLE_MOVE(v) == (uint32_t)(i + 1)) continue;
    out[n++] = (uint8_t)(((i + 1) << 2) | ORACLE_MOVE);
  }
  if (n == before) out[n++] = ORACLE_MOVE; /* Struggle */
  return n;
}

uint8_t oracle_result_from_state(const uint8_t *battle384) {
  const Battle *b = (const Battle *)battle384;
  int a1 = any_alive(&b->sides[0]), a2 = any_alive(&b->sides[1]);
  if (!a1) return mk_result(a2 ? ORACLE_LOSE : ORACLE_TIE, 0, 0);
  if (!a2) return mk_result(ORACLE_WIN, 0, 0);
  int f1 = b->sides[0].pokemon[b->sides[0].order[0] - 1].hp == 0;
  int f2 = b->sides[1].pokemon[b->sides[1].order[0] - 1].hp == 0;
  if (f1) return mk_result(0, ORACLE_SWITCH, f2 ? ORACLE_SWITCH : ORACLE_PASS);
  if (f2) return mk_result(0, ORACLE_PASS, ORACLE_SWITCH);
  return mk_result(0, ORACLE_MOVE, ORACLE_MOVE);
}

uint64_t oracle_hash64(const uint8_t *p, size_t n) {
  uint64_t h = 0xcbf29ce484222325ull;
  for (size_t i = 0; i < n; ++i) { h ^= p[i]; h *= 0x100000001b3ull; }
  return h;
}
