// oak_amd/csrc/gen1_regs.hpp -- register-resident gen-1 turn resolution for the rollout kernel.
//
// Same semantics as gen1_device.hpp (and bit-identical results), different data placement:
// with one wave per SIMD every LDS round trip (~64-128 cycles) is exposed, and the LDS-resident
// engine makes ~150 of them per turn-step.  Here everything a turn touches lives in VGPRs:
//   * per side (struct SideR): the 32-byte active block, a cached copy of the stored (party)
//     Pokemon behind it, the order bytes + last selected/used move, the chance durations word
//     and a 6-bit "alive" mask (so legal-choice enumeration and faint checks need no memory);
//   * per battle: RNG seed, turn, last_damage, last_moves.
// LDS (lane-interleaved) keeps only the MUTABLE part of the 12 party slots (PP, hp, status: 24 dwords
// per lane) and is touched only on switches; immutable party data is re-read from the input battle.
//
// Player indices are lane-divergent (who moves first differs per lane), and indexing registers by
// a divergent value would spill them to scratch.  So the code is written in a MOVER / TARGET frame:
// `S` is always the side acting, `F` its foe, and the two register sets are physically swapped
// (v_cndmask / v_swap) between the two halves of a turn.  One copy of the move code serves both.
//
// Reference call sites replaced: cpp/include/search/mcts.h:453-479 (choices x2 + update per
// turn-step); build configuration mirrored: /root/reference/dev/libpkmn:9.
#pragma once
#include "gen1_device.hpp"

namespace oak {

// ---- optional region profile (tools/site_profile.sh builds a separate library with -DOAKGPU_SITE_PROFILE).
// Per region: passes, active lanes summed over passes, wave cycles, lane-weighted cycles.  A "pass" is one
// trip of (part of) a wave through the region, so lanes / (64 * passes) is the region's SIMT efficiency and
// cycles its share of the wave's time.  Compiles to nothing in the product build.
#ifdef OAKGPU_SITE_PROFILE
static __device__ unsigned long long g_site_prof[32 * 4];
// one wave per block in the profiled kernel: accumulate in LDS (cheap), flush once per wave at exit
__device__ __forceinline__ unsigned long long *site_lds() {
  __shared__ unsigned long long s_prof[32 * 4];
  return s_prof;
}
__device__ __forceinline__ void site_zero() {
  unsigned long long *p = site_lds();
  for (int i = threadIdx.x; i < 32 * 4; i += blockDim.x) p[i] = 0;
  __syncthreads();
}
__device__ __forceinline__ void site_flush() {
  __syncthreads();
  unsigned long long *p = site_lds();
  for (int i = threadIdx.x; i < 32 * 4; i += blockDim.x) if (p[i]) atomicAdd(&g_site_prof[i], p[i]);
}
__device__ __forceinline__ void site_add(int id, long long dt) {
  const uint64_t ex = __ballot(1);
  const uint32_t lane = __lane_id();
  if ((ex & ((1ull << lane) - 1)) == 0) {
    unsigned long long *p = site_lds(); // ds_add_u64 without return: fire and forget
    __hip_atomic_fetch_add(&p[id * 4 + 0], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_add(&p[id * 4 + 1], (unsigned long long)__popcll(ex), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_add(&p[id * 4 + 2], (unsigned long long)dt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_add(&p[id * 4 + 3], (unsigned long long)dt * __popcll(ex), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
}
struct SiteScope {
  int id; long long t0;
  __device__ __forceinline__ SiteScope(int i) : id(i), t0(clock64()) {}
  __device__ __forceinline__ ~SiteScope() { site_add(id, clock64() - t0); }
};
#define OAK_SCOPE(id) SiteScope oak_scope_##id(id)
#define OAK_T0(v) long long v = clock64()
#define OAK_T1(id, v) site_add(id, clock64() - v)
#define OAK_PROF_ZERO() site_zero()
#define OAK_PROF_FLUSH() site_flush()
#else
#define OAK_SCOPE(id)
#define OAK_T0(v)
#define OAK_T1(id, v)
#define OAK_PROF_ZERO()
#define OAK_PROF_FLUSH()
#endif
// region ids
enum { PS_REFILL = 0, PS_STEP, PS_LEGAL_DRAW, PS_ORDER, PS_EXEC_MOVE, PS_SWITCH_IN, PS_BEFORE_MOVE, PS_EXEC_SELECTED_PRE,
       PS_RUN_MOVE, PS_GATES_HIT, PS_STATUS_BODIES, PS_DAMAGE, PS_SECONDARY_APPLY, PS_FAINT_RESIDUAL, PS_PUBLISH,
       PS_CALC_DAMAGE, PS_APPLY_HITS, PS_DAMAGE_TAIL, PS_HEAVY, PS_H_CONVERSION, PS_H_HAZE, PS_H_HEAL, PS_H_MIMIC,
       PS_H_POISON, PS_H_SUBSTITUTE, PS_H_TRANSFORM, PS_H_BIDE, PS_H_SLEEP, PS_H_DISABLE, PS_COUNT };

struct SideR {
  uint32_t a0, a1, a2; // active: hp|atk<<16, def|spe<<16, spc|species<<16|types<<24
  uint32_t bo;         // boosts nibbles: atk def | spe spc | acc eva
  uint32_t vlo, vhi;   // volatiles
  uint32_t m01, m23;   // active move slots (id | pp<<8) x4
  uint32_t p0, p1, p2, p3, p4, p5; // stored Pokemon dwords (layout.h:33-41)
  uint32_t o0, o1;     // order[0..3]; order[4], order[5], last_selected_move, last_used_move
  uint32_t dur;        // chance durations of this side
  uint32_t misc;       // bits 0-5 alive mask by order position, bit 8 absolute player, bits 16-23 damage override
};

#define OAK_FOR_SIDE_FIELDS(X) X(a0) X(a1) X(a2) X(bo) X(vlo) X(vhi) X(m01) X(m23) X(p0) X(p1) X(p2) X(p3) X(p4) X(p5) X(o0) X(o1) X(dur) X(misc)

__device__ __forceinline__ void swap_sides(SideR &a, SideR &b) {
  // v_swap_b32: one instruction per field (the compiler's own lowering is three moves through a temporary)
#define X(f) asm("v_swap_b32 %0, %1" : "+v"(a.f), "+v"(b.f));
  OAK_FOR_SIDE_FIELDS(X)
#undef X
}
__device__ __forceinline__ void cswap_sides(bool c, SideR &a, SideR &b) {
#define X(f) { uint32_t t = c ? b.f : a.f; b.f = c ? a.f : b.f; a.f = t; }
  OAK_FOR_SIDE_FIELDS(X)
#undef X
}

// GIN_LDS: the lane's input battle lives in LDS (k_rollout_staged's staged copy) -- `gin` is then an LDS pointer and the immutable party
// data is read with ds_read; as a generic pointer (flat loads that resolve to LDS) the staged kernel carried 44-54 spilled SGPRs.
template <int STRIDE, bool TRACK_ACTIONS, bool GIN_LDS = false>
struct EngineR {
  lds_u32 *m; // party storage (lane-interleaved LDS, same addressing as Engine)
  Tables T;
  SideR S, F; // mover / target frame
  uint64_t actS, actF; // chance actions (TRACK_ACTIONS only)
  uint64_t rng;
  uint32_t turn, last_damage;
  uint32_t lm; // last_moves: index0 | counterable0<<8 | index1<<16 | counterable1<<24 (absolute players)

  // ---- party storage.  Only what can CHANGE on a benched Pokemon lives in LDS: two dwords per party
  // member (w0 = the four PP bytes, w1 = hp | status << 16), 24 dwords per lane, lane-interleaved.
  // Everything immutable (stats, move ids, species, types, level) is re-read from the lane's INPUT
  // battle in global memory (`gin`, L2-resident) on the rare occasions it is needed (switch-in,
  // Transform bookkeeping, final write-back).  6 KB of LDS per wave leaves occupancy to the VGPR budget.
  static constexpr int PARTY_WORDS = 24;
  template <bool L, class Dummy = void> struct GinPtr { typedef const uint32_t *type; };
  template <class Dummy> struct GinPtr<true, Dummy> { typedef const lds_u32 *type; };
  typedef typename GinPtr<GIN_LDS>::type gin_t;
  gin_t gin; // this lane's input battle: 96 dwords
  __device__ __forceinline__ uint4 gin4(int off) const { // 16 aligned bytes of it
    typedef uint32_t v4 __attribute__((ext_vector_type(4)));
    if constexpr (GIN_LDS) { const v4 t = *(const OAK_LDS v4 *)(gin + off); return make_uint4(t.x, t.y, t.z, t.w); }
    else return *(const uint4 *)(gin + off);
  }
  __device__ __forceinline__ uint32_t pw(uint32_t side, uint32_t i, int k) const { return m[((side * 6 + i) * 2 + k) * STRIDE]; }
  __device__ __forceinline__ void set_pw(uint32_t side, uint32_t i, int k, uint32_t v) { m[((side * 6 + i) * 2 + k) * STRIDE] = v; }
  __device__ __forceinline__ uint32_t gdw(uint32_t side, uint32_t i, int k) const { return gin[side * 46 + i * 6 + k]; }
  static __device__ __forceinline__ uint32_t pack_pp(uint32_t d2, uint32_t d3, uint32_t d4) { // stored dwords 2..4 -> 4 PP bytes
    return (d2 >> 24) | (((d3 >> 8) & 0xFF) << 8) | ((d3 >> 24) << 16) | (((d4 >> 8) & 0xFF) << 24);
  }
  // ---- per-side field helpers ----
  static __device__ __forceinline__ uint32_t absp(const SideR &x) { return (x.misc >> 8) & 1; }
  static __device__ __forceinline__ uint32_t hp(const SideR &x) { return x.p4 >> 16; }
  static __device__ __forceinline__ void set_hp(SideR &x, uint32_t v) { x.p4 = (x.p4 & 0xFFFF) | (v << 16); }
  static __device__ __forceinline__ uint32_t maxhp(const SideR &x) { return x.p0 & 0xFFFF; }
  static __device__ __forceinline__ uint32_t status(const SideR &x) { return x.p5 & 0xFF; }
  static __device__ __forceinline__ void set_status(SideR &x, uint32_t v) { x.p5 = (x.p5 & ~0xFFu) | (v & 0xFF); }
  static __device__ __forceinline__ uint32_t level(const SideR &x) { return x.p5 >> 24; }
  static __device__ __forceinline__ uint32_t species_stored(const SideR &x) { return (x.p5 >> 8) & 0xFF; }
  static __device__ __forceinline__ uint32_t types(const SideR &x) { return x.a2 >> 24; }
  static __device__ __forceinline__ uint32_t spe(const SideR &x) { return x.a1 >> 16; }
  static __device__ __forceinline__ uint32_t last_sel(const SideR &x) { return (x.o1 >> 16) & 0xFF; }
  static __device__ __forceinline__ void set_last_sel(SideR &x, uint32_t v) { x.o1 = (x.o1 & ~(0xFFu << 16)) | ((v & 0xFF) << 16); }
  static __device__ __forceinline__ uint32_t last_used(const SideR &x) { return x.o1 >> 24; }
  static __device__ __forceinline__ void set_last_used(SideR &x, uint32_t v) { x.o1 = (x.o1 & 0x00FFFFFFu) | (v << 24); }
  static __device__ __forceinline__ uint32_t order0(const SideR &x) { return x.o0 & 0xFF; }
  // Packed accessors always read / write BOTH dwords through one 64-bit value: selecting one of
  // several struct fields by a lane-divergent slot makes the compiler demote the fields to scratch.
  static __device__ __forceinline__ uint64_t amoves(const SideR &x) { return (uint64_t)x.m01 | ((uint64_t)x.m23 << 32); }
  static __device__ __forceinline__ void set_amoves(SideR &x, uint64_t v) { x.m01 = (uint32_t)v; x.m23 = (uint32_t)(v >> 32); }
  static __device__ __forceinline__ uint32_t active_move(const SideR &x, uint32_t slot) { // id | pp<<8, slot 1..4
    return (uint32_t)(amoves(x) >> (16 * (slot - 1))) & 0xFFFF;
  }
  // stored move slots live in p2 (high half), p3, p4 (low half)
  static __device__ __forceinline__ uint64_t smoves(const SideR &x) {
    return (uint64_t)(x.p2 >> 16) | ((uint64_t)x.p3 << 16) | ((uint64_t)(x.p4 & 0xFFFF) << 48);
  }
  static __device__ __forceinline__ void set_smoves(SideR &x, uint64_t v) {
    x.p2 = (x.p2 & 0xFFFF) | ((uint32_t)(v & 0xFFFF) << 16);
    x.p3 = (uint32_t)(v >> 16);
    x.p4 = (x.p4 & 0xFFFF0000u) | (uint32_t)(v >> 48);
  }
  // active stat by index: 0 atk 1 def 2 spe 3 spc
  static __device__ __forceinline__ uint32_t astat(const SideR &x, int idx) {
    return idx == 0 ? x.a0 >> 16 : idx == 1 ? x.a1 & 0xFFFF : idx == 2 ? x.a1 >> 16 : x.a2 & 0xFFFF;
  }
  static __device__ __forceinline__ void set_astat(SideR &x, int idx, uint32_t v) {
    if (idx == 0) x.a0 = (x.a0 & 0xFFFF) | (v << 16);
    else if (idx == 1) x.a1 = (x.a1 & 0xFFFF0000u) | v;
    else if (idx == 2) x.a1 = (x.a1 & 0xFFFF) | (v << 16);
    else x.a2 = (x.a2 & 0xFFFF0000u) | v;
  }
  // volatile sub-fields
  static __device__ __forceinline__ uint32_t conf_left(const SideR &x) { return (x.vlo >> 18) & 7; }
  static __device__ __forceinline__ void set_conf_left(SideR &x, uint32_t v) { x.vlo = (x.vlo & ~(7u << 18)) | ((v & 7) << 18); }
  static __device__ __forceinline__ uint32_t attacks(const SideR &x) { return (x.vlo >> 21) & 7; }
  static __device__ __forceinline__ void set_attacks(SideR &x, uint32_t v) { x.vlo = (x.vlo & ~(7u << 21)) | ((v & 7) << 21); }
  static __device__ __forceinline__ uint32_t vstate(const SideR &x) { return (x.vlo >> 24) | ((x.vhi & 0xFF) << 8); }
  static __device__ __forceinline__ void set_vstate(SideR &x, uint32_t v) {
    x.vlo = (x.vlo & 0x00FFFFFFu) | ((v & 0xFF) << 24);
    x.vhi = (x.vhi & ~0xFFu) | ((v >> 8) & 0xFF);
  }
  static __device__ __forceinline__ uint32_t sub_hp(const SideR &x) { return (x.vhi >> 8) & 0xFF; }
  static __device__ __forceinline__ void set_sub_hp(SideR &x, uint32_t v) { x.vhi = (x.vhi & ~(0xFFu << 8)) | ((v & 0xFF) << 8); }
  static __device__ __forceinline__ uint32_t transform_id(const SideR &x) { return (x.vhi >> 16) & 15; }
  static __device__ __forceinline__ void set_transform_id(SideR &x, uint32_t v) { x.vhi = (x.vhi & ~(15u << 16)) | ((v & 15) << 16); }
  static __device__ __forceinline__ uint32_t disable_left(const SideR &x) { return (x.vhi >> 20) & 15; }
  static __device__ __forceinline__ void set_disable_left(SideR &x, uint32_t v) { x.vhi = (x.vhi & ~(15u << 20)) | ((v & 15) << 20); }
  static __device__ __forceinline__ uint32_t disable_move(const SideR &x) { return (x.vhi >> 24) & 7; }
  static __device__ __forceinline__ void set_disable_move(SideR &x, uint32_t v) { x.vhi = (x.vhi & ~(7u << 24)) | ((v & 7) << 24); }
  static __device__ __forceinline__ uint32_t toxic_ctr(const SideR &x) { return x.vhi >> 27; }
  static __device__ __forceinline__ void set_toxic_ctr(SideR &x, uint32_t v) { x.vhi = (x.vhi & ~(31u << 27)) | ((v & 31) << 27); }
  static __device__ __forceinline__ int boost_get(const SideR &x, int idx) { // 0 atk 1 def 2 spe 3 spc 4 acc 5 eva
    uint32_t n = (x.bo >> (4 * idx)) & 15;
    return (int)((n ^ 8) - 8);
  }
  static __device__ __forceinline__ void boost_put(SideR &x, int idx, int v) {
    x.bo = (x.bo & ~(15u << (4 * idx))) | (((uint32_t)v & 15) << (4 * idx));
  }
  static __device__ __forceinline__ uint32_t dget(const SideR &x, int sh, int bits) { return (x.dur >> sh) & ((1u << bits) - 1); }
  static __device__ __forceinline__ void dset(SideR &x, int sh, int bits, uint32_t v) {
    uint32_t mask = ((1u << bits) - 1) << sh;
    x.dur = (x.dur & ~mask) | ((v << sh) & mask);
  }
  static __device__ __forceinline__ void clear_binding(SideR &x) { x.vlo &= ~V_BINDING; dset(x, 28, 3, 0); }
  static __device__ __forceinline__ void status_modify(uint32_t st, SideR &x) {
    if (st & ST_PAR) { uint32_t s = (x.a1 >> 16) / 4; set_astat(x, 2, s < 1 ? 1 : s); }
    else if (st & ST_BRN) { uint32_t a = (x.a0 >> 16) / 2; set_astat(x, 0, a < 1 ? 1 : a); }
  }

  // ---- actions (mover frame: `self` selects actS / actF) ----
  __device__ __forceinline__ void act_set(bool self, int sh, int bits, uint32_t v) {
    if constexpr (TRACK_ACTIONS) {
      uint64_t mask = ((1ull << bits) - 1) << sh;
      uint64_t &a = self ? actS : actF;
      a = (a & ~mask) | (((uint64_t)v << sh) & mask);
    }
  }
  __device__ __forceinline__ void act_bool(bool self, int sh, bool v) { act_set(self, sh, 2, v ? 2u : 1u); }

  // ---- RNG ----
  __device__ __forceinline__ uint32_t rng_next() {
    rng = 0x5D588B656C078965ull * rng + 0x0000000000269EC3ull;
    return (uint32_t)(rng >> 32);
  }
  __device__ __forceinline__ uint32_t rng_range(uint32_t from, uint32_t to) {
    return from + __umulhi(rng_next(), to - from);
  }
  __device__ __forceinline__ bool rng_chance(uint32_t num) { return (rng_next() >> 24) < num; } // range(0,256) < num

  __device__ __forceinline__ Move move_data(uint32_t id) const { return Move{T.mv[id]}; }
  __device__ __forceinline__ uint32_t chart(uint32_t atk_type, uint32_t def_type) const { return T.chart[atk_type * 15 + def_type]; }
  static __device__ __forceinline__ bool has_type(uint32_t ty, uint32_t t) { return (ty & 15) == t || (ty >> 4) == t; }
  // Variable integer division costs ~40 VALU on gfx950; the engine's divisors are tiny, so use one mul-hi
  // against the LDS reciprocal table instead (exact for x < 2^24, d <= 255; see gen1_device.hpp).
  __device__ __forceinline__ uint32_t fast_div24(uint32_t x, uint32_t d) const { return d == 1 ? x : __umulhi(x, T.rcp[d]); }
  __device__ __forceinline__ uint32_t fast_mod(uint32_t x, uint32_t n) const { // any 32-bit x, n in 1..255
    if (n == 1) return 0;
    uint32_t r = x - __umulhi(x, T.rcp[n]) * n; // quotient estimate is exact or one too large
    return (int32_t)r < 0 ? r + n : r;
  }
  __device__ __forceinline__ uint32_t scale_boost(uint32_t x, int stage) const {
    if (stage == 0) return x;
    const uint32_t b = T.boost[stage + 6], v = x * (b & 0xFF), den = b >> 8; // den is 100, 10 or 1
    return den == 100 ? v / 100 : den == 10 ? v / 10 : v;
  }

  // unmodified (party) stat idx of side x; through Transform this is the copied Pokemon's stat (LDS)
  __device__ __forceinline__ uint32_t unmodified_stat(const SideR &x, int idx) { // 0 atk 1 def 2 spe 3 spc
    if (!(x.vlo & V_TRANSFORM))
      return idx == 0 ? x.p0 >> 16 : idx == 1 ? x.p1 & 0xFFFF : idx == 2 ? x.p1 >> 16 : x.p2 & 0xFFFF;
    const uint32_t id = transform_id(x), sd = id >> 3, pi = (id & 7) - 1; // immutable stats of the copied Pokemon
    const uint32_t d = gdw(sd, pi, idx == 0 ? 0 : idx == 3 ? 2 : 1);
    return (idx == 0 || idx == 2) ? d >> 16 : d & 0xFFFF;
  }

  // ---- register <-> LDS movement ----
  __device__ __forceinline__ void writeback_stored(const SideR &x) { // mutable part of the active's party slot
    const uint32_t sd = absp(x), pi = order0(x) - 1;
    set_pw(sd, pi, 0, pack_pp(x.p2, x.p3, x.p4));
    set_pw(sd, pi, 1, (x.p4 >> 16) | ((x.p5 & 0xFF) << 16));
  }
  // One lane's battle: global AoS (96 dwords, 16-byte aligned) -> registers + party LDS.
  // Done in small chunk groups (scheduling barriers in between) so the staging registers never
  // pile up on top of the engine's own ~45 live registers.
  __device__ __forceinline__ void alive_from_lds(SideR &x) {
    const uint32_t sd = absp(x);
    uint32_t hpmask = 0; // bit i: party member i+1 has hp > 0
#pragma unroll
    for (int i = 0; i < 6; ++i) hpmask |= ((pw(sd, i, 1) & 0xFFFF) != 0 ? 1u : 0u) << i;
    const uint64_t ord = (uint64_t)x.o0 | ((uint64_t)(x.o1 & 0xFFFF) << 32);
    uint32_t alive = 0;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const uint32_t id = (uint32_t)(ord >> (8 * k)) & 0xFF;
      alive |= (id != 0 ? (hpmask >> ((id - 1) & 7)) & 1u : 0u) << k;
    }
    x.misc = (x.misc & ~63u) | alive;
  }
  __device__ __forceinline__ void load_stored(SideR &x) { // branch-free: an empty side (order[0] == 0) reads slot 1, masked to 0
    const uint32_t id = order0(x);
    const uint32_t keep = id != 0 ? 0xFFFFFFFFu : 0u;
    const uint32_t sd = absp(x), pi = (id != 0 ? id : 1u) - 1;
    const uint32_t pp = pw(sd, pi, 0), hs = pw(sd, pi, 1);
    x.p0 = gdw(sd, pi, 0) & keep;
    x.p1 = gdw(sd, pi, 1) & keep;
    x.p2 = ((gdw(sd, pi, 2) & 0x00FFFFFFu) | ((pp & 0xFF) << 24)) & keep;
    x.p3 = ((gdw(sd, pi, 3) & 0x00FF00FFu) | (((pp >> 8) & 0xFF) << 8) | (((pp >> 16) & 0xFF) << 24)) & keep;
    x.p4 = ((gdw(sd, pi, 4) & 0x000000FFu) | ((pp >> 24) << 8) | ((hs & 0xFFFF) << 16)) & keep;
    x.p5 = ((gdw(sd, pi, 5) & 0xFFFFFF00u) | ((hs >> 16) & 0xFF)) & keep;
  }
  __device__ __forceinline__ void load_battle_global(const uint8_t *battle384, uint32_t dur0, uint32_t dur1) {
    gin = (gin_t)battle384;
#pragma unroll
    for (uint32_t sd = 0; sd < 2; ++sd)
#pragma unroll
      for (uint32_t i = 0; i < 6; ++i) {
        const uint32_t d2 = gdw(sd, i, 2), d3 = gdw(sd, i, 3), d4 = gdw(sd, i, 4), d5 = gdw(sd, i, 5);
        set_pw(sd, i, 0, pack_pp(d2, d3, d4));
        set_pw(sd, i, 1, (d4 >> 16) | ((d5 & 0xFF) << 16));
      }
    { const uint4 v = gin4(36); S.a0 = v.x; S.a1 = v.y; S.a2 = v.z; S.bo = v.w; }
    { const uint4 v = gin4(40); S.vlo = v.x; S.vhi = v.y; S.m01 = v.z; S.m23 = v.w; }
    S.o0 = gin[44]; S.o1 = gin[45];
    F.a0 = gin[82]; F.a1 = gin[83];
    { const uint4 v = gin4(84); F.a2 = v.x; F.bo = v.y; F.vlo = v.z; F.vhi = v.w; }
    { const uint4 v = gin4(88); F.m01 = v.x; F.m23 = v.y; F.o0 = v.z; F.o1 = v.w; }
    { const uint4 v = gin4(92); turn = v.x & 0xFFFF; last_damage = v.x >> 16; lm = v.y; rng = (uint64_t)v.z | ((uint64_t)v.w << 32); }
    S.dur = dur0; F.dur = dur1;
    S.misc = 0; F.misc = 1u << 8;
    actS = actF = 0;
    alive_from_lds(S); // same-lane LDS write -> read: ordered by the wave's own lgkmcnt
    alive_from_lds(F);
    load_stored(S);
    load_stored(F);
  }
  // registers + party LDS (+ immutable input fields) -> global AoS; frame must be normalised (S = P1, F = P2)
  __device__ __forceinline__ void store_battle_global(uint8_t *battle384) {
    if (order0(S) != 0) writeback_stored(S);
    if (order0(F) != 0) writeback_stored(F);
    uint32_t *g = (uint32_t *)battle384;
#pragma unroll
    for (uint32_t sd = 0; sd < 2; ++sd)
#pragma unroll
      for (uint32_t i = 0; i < 6; ++i) {
        const uint32_t pp = pw(sd, i, 0), hs = pw(sd, i, 1);
        uint32_t *d = g + sd * 46 + i * 6;
        d[0] = gdw(sd, i, 0);
        d[1] = gdw(sd, i, 1);
        d[2] = (gdw(sd, i, 2) & 0x00FFFFFFu) | ((pp & 0xFF) << 24);
        d[3] = (gdw(sd, i, 3) & 0x00FF00FFu) | (((pp >> 8) & 0xFF) << 8) | (((pp >> 16) & 0xFF) << 24);
        d[4] = (gdw(sd, i, 4) & 0x000000FFu) | ((pp >> 24) << 8) | ((hs & 0xFFFF) << 16);
        d[5] = (gdw(sd, i, 5) & 0xFFFFFF00u) | ((hs >> 16) & 0xFF);
      }
    *(uint4 *)(g + 36) = make_uint4(S.a0, S.a1, S.a2, S.bo);
    *(uint4 *)(g + 40) = make_uint4(S.vlo, S.vhi, S.m01, S.m23);
    g[44] = S.o0; g[45] = S.o1;
    g[82] = F.a0; g[83] = F.a1;
    *(uint4 *)(g + 84) = make_uint4(F.a2, F.bo, F.vlo, F.vhi);
    *(uint4 *)(g + 88) = make_uint4(F.m01, F.m23, F.o0, F.o1);
    *(uint4 *)(g + 92) = make_uint4(turn | (last_damage << 16), lm, (uint32_t)rng, (uint32_t)(rng >> 32));
  }
  // MCTS::randomize_hidden_variables (cpp/include/search/durations.h:25-97) on the register image;
  // every draw reuses the same un-advanced battle.rng value
  __device__ __forceinline__ void randomize_hidden_side(SideR &x) {
    const uint32_t hi = (uint32_t)(rng >> 32), lo = (uint32_t)rng;
    auto mod = [&](uint32_t mm) { uint32_t two32 = (0xFFFFFFFFu % mm + 1) % mm; return ((hi % mm) * two32 + (lo % mm)) % mm; };
    const uint32_t d = x.dur;
    const uint32_t confusion = (d >> 18) & 7, disable = (d >> 21) & 15, attacking = (d >> 25) & 7, binding = (d >> 28) & 7;
    if (confusion) {
      const uint32_t one = confusion == 1;
      set_conf_left(x, (mod((6 - (confusion + one)) & 0xFF) + 1 + one) & 0xFF);
    }
    if (disable) set_disable_left(x, (mod((9 - disable) & 0xFF) + 1) & 0xFF);
    if (attacking && (x.vlo & (V_BIDE | V_THRASHING))) set_attacks(x, attacking == 3 ? 1u : 4u - (attacking + mod(2)));
    if (binding) {
      const uint32_t idx = mod(40);
      uint32_t a; // rows of the reference's 4 x 40 table as run lengths {15,15,6,4} {24,8,8} {20,20} {40}
      if (binding == 1) a = idx < 15 ? 1 : idx < 30 ? 2 : idx < 36 ? 3 : 4;
      else if (binding == 2) a = idx < 24 ? 1 : idx < 32 ? 2 : 3;
      else if (binding == 3) a = idx < 20 ? 1 : 2;
      else a = 1;
      set_attacks(x, a);
    }
    const uint64_t ord = (uint64_t)x.o0 | ((uint64_t)(x.o1 & 0xFFFF) << 32);
    for (int i = 0; i < 6; ++i) {
      const uint32_t sleep = (d >> (3 * i)) & 7;
      if (!sleep) continue;
      const uint32_t draw = (mod((8 - sleep) & 0xFF) + 1) & 0xFF;
      if (i == 0) {
        const uint32_t st = status(x);
        if ((st & 7) && !(st & 0x80)) set_status(x, (st & 0xF8) | draw);
      } else {
        const uint32_t id = (uint32_t)(ord >> (8 * i)) & 0xFF;
        const uint32_t hs = pw(absp(x), id - 1, 1), st = (hs >> 16) & 0xFF;
        if ((st & 7) && !(st & 0x80)) set_pw(absp(x), id - 1, 1, (hs & 0xFFFF) | (((st & 0xF8) | draw) << 16));
      }
    }
  }
  __device__ __forceinline__ void randomize_hidden() { randomize_hidden_side(S); randomize_hidden_side(F); }
  // last_moves accessors by absolute player
  __device__ __forceinline__ uint32_t lm_index(uint32_t ap) const { return (lm >> (16 * ap)) & 0xFF; }
  __device__ __forceinline__ void set_lm_index(uint32_t ap, uint32_t v) { lm = (lm & ~(0xFFu << (16 * ap))) | ((v & 0xFF) << (16 * ap)); }
  __device__ __forceinline__ uint32_t lm_counterable(uint32_t ap) const { return (lm >> (16 * ap + 8)) & 0xFF; }
  __device__ __forceinline__ void set_lm_counterable(uint32_t ap, uint32_t v) { lm = (lm & ~(0xFFu << (16 * ap + 8))) | ((v & 0xFF) << (16 * ap + 8)); }

  // ---- switching: side x switches to party position `slot`; y is the other side --------------
  __device__ __forceinline__ void switch_in(SideR &x, SideR &y, uint32_t slot) {
    if (order0(x) != 0) {
      if (status(x) == ST_TOX) set_status(x, ST_PSN); // toxic reverts on leaving the field
      writeback_stored(x);
    }
    // swap order bytes 0 and slot-1, alive bits 0 and slot-1
    uint64_t o = (uint64_t)x.o0 | ((uint64_t)(x.o1 & 0xFFFF) << 32);
    const uint32_t sh = 8 * (slot - 1);
    const uint64_t b0 = o & 0xFF, bk = (o >> sh) & 0xFF;
    o = (o & ~(0xFFull << sh)) | (b0 << sh);
    o = (o & ~0xFFull) | bk;
    x.o0 = (uint32_t)o;
    x.o1 = (x.o1 & 0xFFFF0000u) | (uint32_t)(o >> 32);
    const uint32_t al = x.misc & 63, a0b = al & 1, akb = (al >> (slot - 1)) & 1;
    uint32_t nal = (al & ~(1u | (1u << (slot - 1)))) | akb | (a0b << (slot - 1));
    if (slot == 1) nal = al;
    x.misc = (x.misc & ~63u) | nal;
    uint32_t d = x.dur;
    const uint32_t s0 = d & 7, sk = (d >> (3 * (slot - 1))) & 7;
    d = (d & ~7u) | sk;
    if (slot != 1) d = (d & ~(7u << (3 * (slot - 1)))) | (s0 << (3 * (slot - 1)));
    x.dur = d & ((1u << 18) - 1);
    set_last_used(x, 0);
    set_last_used(y, 0);
    load_stored(x);
    x.a0 = x.p0;
    x.a1 = x.p1;
    x.a2 = (x.p2 & 0xFFFF) | (((x.p5 >> 8) & 0xFF) << 16) | (((x.p5 >> 16) & 0xFF) << 24);
    x.bo = 0;
    x.vlo = 0;
    x.vhi = 0;
    x.m01 = (x.p2 >> 16) | (x.p3 << 16);
    x.m23 = (x.p3 >> 16) | (x.p4 << 16);
    status_modify(x.p5 & 0xFF, x);
    clear_binding(y);
  }

  // ---- move selection (absolute player ap owns side x) ---------------------------------------
  __device__ __forceinline__ void select_move(SideR &x, uint32_t choice) {
    if ((choice & 3) == C_PASS) return;
    if (x.vlo & (V_RECHARGING | V_RAGE)) return;
    x.vlo &= ~V_FLINCH;
    if (x.vlo & (V_THRASHING | V_CHARGING)) return;
    if ((choice & 3) == C_SWITCH) return;
    if (x.vlo & (V_BIDE | V_BINDING)) return;
    const uint32_t data = choice >> 2;
    set_last_sel(x, data == 0 ? (uint32_t)M_Struggle : active_move(x, data) & 0xFF);
    set_lm_index(absp(x), data);
  }

  // ---- damage ------------------------------------------------------------------------------------
  __device__ __forceinline__ bool check_crit(Move mv) {
    uint32_t chance = (T.sp0[species_stored(S)] >> 24) / 2;
    if (S.vlo & V_FOCUSENERGY) chance = chance / 2;
    else { chance *= 2; if (chance > 255) chance = 255; }
    if (mv.effect() == E_HighCritical) { chance *= 4; if (chance > 255) chance = 255; }
    else chance = chance / 2;
    bool crit = rng_chance(chance);
    act_bool(true, AC_CRIT, crit);
    return crit;
  }

  // base damage of S attacking `tgt` (tgt = S for confusion self-hits)
  __device__ __forceinline__ bool calc_damage(const SideR &tgt, uint32_t bp, uint32_t type, bool explode, bool crit) {
    const bool special = type >= 8;
    uint32_t atk, def;
    if (crit) {
      atk = unmodified_stat(S, special ? 3 : 0);
      def = unmodified_stat(tgt, special ? 3 : 1);
    } else {
      atk = special ? S.a2 & 0xFFFF : S.a0 >> 16;
      def = (special ? tgt.a2 & 0xFFFF : tgt.a1 & 0xFFFF) * ((tgt.vlo & (special ? V_LIGHTSCREEN : V_REFLECT)) ? 2u : 1u);
    }
    if (atk > 255 || def > 255) {
      atk = (atk / 4) & 255; if (atk < 1) atk = 1;
      def = (def / 4) & 255; if (def < 1) def = 1;
    }
    uint32_t lvl = level(S) * (crit ? 2u : 1u);
    if (explode) { def = def / 2; if (def < 1) def = 1; }
    if (def == 0) return false;
    uint32_t d = (lvl * 2 / 5) + 2;
    d *= bp;
    d *= atk;                 // <= 206 * 255 * 255 < 2^24
    d = fast_div24(d, def);   // def in 1..255 here
    d /= 50;
    if (d > 997) d = 997;
    d += 2;
    last_damage = d;
    return true;
  }

  __device__ __forceinline__ void adjust_damage(Move mv) {
    const uint32_t ft = types(F), t1 = ft & 15, t2 = ft >> 4;
    uint32_t d = last_damage;
    if (has_type(types(S), mv.type())) d = (d + d / 2) & 0xFFFF;
    const uint32_t e1 = chart(mv.type(), t1), e2 = chart(mv.type(), t2);
    if (e1 != 2) d = (d * e1 / 2) & 0xFFFF;
    if (t1 != t2 && e2 != 2) d = (d * e2 / 2) & 0xFFFF;
    last_damage = d;
  }

  __device__ __forceinline__ void randomize_damage() {
    if (last_damage <= 1) return;
    uint32_t roll = (S.misc >> 16) & 0xFF;
    if (roll == 0) roll = rng_range(217, 256);
    act_set(true, AC_DAMAGE, 8, roll);
    last_damage = last_damage * roll / 255;
  }

  // damage to `tgt` through `sub`'s substitute when up; returns true when a substitute broke
  __device__ __forceinline__ bool apply_damage(SideR &tgt, SideR &sub, bool &hit_sub) {
    hit_sub = false;
    if (sub.vlo & V_SUBSTITUTE) {
      hit_sub = true;
      const uint32_t shp = sub_hp(sub);
      if (last_damage >= shp) { set_sub_hp(sub, 0); sub.vlo &= ~V_SUBSTITUTE; return true; }
      set_sub_hp(sub, shp - last_damage);
      return false;
    }
    const uint32_t h = hp(tgt);
    if (last_damage > h) last_damage = h;
    set_hp(tgt, h - last_damage);
    return false;
  }

  __device__ __forceinline__ bool move_hit(Move mv) {
    bool miss;
    const uint32_t eff = mv.effect();
    if (eff == E_Swift) return true;
    if (F.vlo & V_INVULNERABLE) miss = true;
    else if ((eff == E_DrainHP || eff == E_DreamEater) && (F.vlo & V_SUBSTITUTE)) miss = true;
    else if (eff >= E_AccuracyDown1 && eff <= E_SpeedDown1 && (F.vlo & V_MIST)) miss = true;
    else {
      uint32_t acc = mv.acc();
      acc = scale_boost(acc, boost_get(S, 4));
      acc = scale_boost(acc, -boost_get(F, 5));
      if (acc > 255) acc = 255;
      if (acc < 1) acc = 1;
      if (acc == 255) miss = false;
      else { miss = !rng_chance(acc); act_bool(true, AC_HIT, !miss); }
    }
    if (!miss) return true;
    last_damage = 0;
    clear_binding(S);
    return false;
  }

  // ---- stat stages: x gains / loses stages; `other` gets the status penalty re-applied ------------
  __device__ __forceinline__ bool boost_side(SideR &x, SideR &other, int idx, int n) {
    int cur = boost_get(x, idx);
    if (cur >= 6) return false;
    int nv = cur + n; if (nv > 6) nv = 6;
    if (idx < 4) {
      if (astat(x, idx) == 999) return false;
      boost_put(x, idx, nv);
      uint32_t v = scale_boost(unmodified_stat(x, idx), nv);
      if (v > 999) v = 999;
      set_astat(x, idx, v);
    } else boost_put(x, idx, nv);
    status_modify(status(other), other); // stat modification glitch
    return true;
  }
  __device__ __forceinline__ bool unboost_foe(int idx, int n) {
    int cur = boost_get(F, idx);
    if (cur <= -6) return false;
    int nv = cur - n; if (nv < -6) nv = -6;
    if (idx < 4) {
      if (astat(F, idx) == 1) return false;
      boost_put(F, idx, nv);
      uint32_t v = scale_boost(unmodified_stat(F, idx), nv);
      if (v < 1) v = 1;
      set_astat(F, idx, v);
    } else boost_put(F, idx, nv);
    status_modify(status(F), F);
    return true;
  }

  __device__ __forceinline__ void haze_clear(SideR &x) {
    set_disable_move(x, 0);
    set_disable_left(x, 0);
    dset(x, 21, 4, 0);
    if (x.vlo & V_CONFUSION) { x.vlo &= ~(V_CONFUSION | (7u << 18)); dset(x, 18, 3, 0); }
    x.vlo &= ~(V_MIST | V_FOCUSENERGY | V_LEECHSEED | V_LIGHTSCREEN | V_REFLECT);
    if (x.vlo & V_TOXIC) {
      x.vlo &= ~V_TOXIC;
      set_toxic_ctr(x, 0);
      if (status(x) == ST_TOX) set_status(x, ST_PSN);
    }
  }
  __device__ __forceinline__ void start_confusion(SideR &x, bool self) {
    x.vlo |= V_CONFUSION;
    set_conf_left(x, rng_range(2, 6));
    dset(x, 18, 3, 1);
    act_set(self, AC_CONFUSION, 3, OBS_STARTED);
  }
  __device__ __forceinline__ void unmodified_to_active(SideR &x) { // Haze: active stats <- unmodified stats
    const uint32_t hpmax = x.a0 & 0xFFFF;
    if (!(x.vlo & V_TRANSFORM)) {
      x.a0 = x.p0; x.a1 = x.p1; x.a2 = (x.a2 & 0xFFFF0000u) | (x.p2 & 0xFFFF);
    } else {
      const uint32_t id = transform_id(x), sd = id >> 3, pi = (id & 7) - 1;
      x.a0 = gdw(sd, pi, 0); x.a1 = gdw(sd, pi, 1); x.a2 = (x.a2 & 0xFFFF0000u) | (gdw(sd, pi, 2) & 0xFFFF);
    }
    (void)hpmax;
  }

  // ---- pre-move checks ---------------------------------------------------------------------
  enum : int { BM_OK = 0, BM_DONE = 1, BM_SKIP_CAN = 2, BM_SKIP_PP = 3, BM_ERR = 4 };

  __device__ int before_move() {
    bool dummy;
    uint32_t st = status(S);
    if (st & ST_SLP) {
      st -= 1;
      const uint32_t left = st & ST_SLP;
      if (!(st & ST_EXT)) {
        if (left == 0) { dset(S, 0, 3, 0); act_set(true, AC_SLEEP, 2, OBS_ENDED); }
        else { dset(S, 0, 3, dget(S, 0, 3) + 1); act_set(true, AC_SLEEP, 2, OBS_CONTINUING); }
      }
      if (left == 0) st = 0;
      set_status(S, st);
      set_last_used(S, 0);
      return BM_DONE;
    }
    if (st & ST_FRZ) { set_last_used(S, 0); return BM_DONE; }
    if (F.vlo & V_BINDING) return BM_DONE;
    if (S.vlo & V_FLINCH) { S.vlo &= ~V_FLINCH; return BM_DONE; }
    if (S.vlo & V_RECHARGING) { S.vlo &= ~V_RECHARGING; return BM_DONE; }
    uint32_t dl = disable_left(S);
    if (dl > 0) {
      dl -= 1;
      set_disable_left(S, dl);
      if (dl == 0) { set_disable_move(S, 0); dset(S, 21, 4, 0); act_set(true, AC_DISABLE, 2, OBS_ENDED); }
      else { dset(S, 21, 4, dget(S, 21, 4) + 1); act_set(true, AC_DISABLE, 2, OBS_CONTINUING); }
    }
    if (S.vlo & V_CONFUSION) {
      const uint32_t left = conf_left(S) - 1;
      set_conf_left(S, left);
      if (left == 0) {
        S.vlo &= ~V_CONFUSION;
        dset(S, 18, 3, 0);
        act_set(true, AC_CONFUSION, 3, OBS_ENDED);
      } else {
        dset(S, 18, 3, dget(S, 18, 3) + 1);
        act_set(true, AC_CONFUSION, 3, OBS_CONTINUING);
        const bool confused = !rng_chance(128);
        act_bool(true, AC_CONFUSED, confused);
        if (confused) {
          S.vlo &= ~(V_BIDE | V_THRASHING | V_MULTIHIT | V_FLINCH | V_CHARGING | V_BINDING | V_INVULNERABLE);
          dset(S, 25, 3, 0);
          dset(S, 28, 3, 0);
          if (!calc_damage(S, 40, T_Normal, false, false)) return BM_ERR;
          (void)apply_damage(S, F, dummy); // gen-1 quirk: the FOE's substitute absorbs the self-hit
          return BM_DONE;
        }
      }
    }
    const uint32_t dm = disable_move(S), sel = last_sel(S);
    if (dm != 0 && sel != M_Struggle && (active_move(S, dm) & 0xFF) == sel) {
      S.vlo &= ~V_CHARGING;
      return BM_DONE;
    }
    if (st & ST_PAR) {
      const bool par = rng_chance(63);
      act_bool(true, AC_PARALYZED, par);
      if (par) {
        S.vlo &= ~(V_BIDE | V_THRASHING | V_CHARGING | V_BINDING | V_INVULNERABLE);
        dset(S, 25, 3, 0);
        dset(S, 28, 3, 0);
        return BM_DONE;
      }
    }
    if (S.vlo & V_BIDE) {
      const uint32_t left = attacks(S) - 1;
      set_attacks(S, left);
      if (left != 0) { dset(S, 25, 3, dget(S, 25, 3) + 1); act_set(true, AC_ATTACKING, 2, OBS_CONTINUING); return BM_DONE; }
      dset(S, 25, 3, 0);
      act_set(true, AC_ATTACKING, 2, OBS_ENDED);
      S.vlo &= ~V_BIDE;
      const uint32_t dmg = (vstate(S) * 2) & 0xFFFF;
      set_vstate(S, 0);
      last_damage = dmg;
      if (dmg == 0) return BM_DONE;
      if (F.vlo & V_INVULNERABLE) return BM_DONE;
      (void)apply_damage(F, F, dummy);
      return BM_DONE;
    }
    if (S.vlo & V_THRASHING) {
      const uint32_t left = attacks(S) - 1;
      set_attacks(S, left);
      if (left == 0) {
        S.vlo &= ~V_THRASHING;
        dset(S, 25, 3, 0);
        act_set(true, AC_ATTACKING, 2, OBS_ENDED);
        start_confusion(S, true);
      } else {
        dset(S, 25, 3, dget(S, 25, 3) + 1);
        act_set(true, AC_ATTACKING, 2, OBS_CONTINUING);
      }
      return BM_SKIP_CAN;
    }
    if (S.vlo & V_BINDING) {
      set_attacks(S, attacks(S) - 1);
      dset(S, 28, 3, dget(S, 28, 3) + 1);
      act_set(true, AC_BINDING, 3, OBS_CONTINUING);
      if (last_damage != 0) (void)apply_damage(F, F, dummy);
      return BM_DONE;
    }
    return (S.vlo & V_RAGE) ? BM_SKIP_PP : BM_OK;
  }

  __device__ __forceinline__ void decrement_pp(uint32_t mslot) {
    if (mslot == 0) return;
    const uint32_t sh = 16 * (mslot - 1) + 8;
    const uint64_t am = amoves(S);
    set_amoves(S, (am & ~(0xFFull << sh)) | ((uint64_t)((((uint32_t)(am >> sh) & 0xFF) - 1) & 63) << sh));
    if (S.vlo & V_TRANSFORM) return;
    const uint64_t sm = smoves(S);
    set_smoves(S, (sm & ~(0xFFull << sh)) | ((uint64_t)((((uint32_t)(sm >> sh) & 0xFF) - 1) & 63) << sh));
  }

  // ---- the move itself, as ONE staged pipeline -------------------------------------------------------
  // Every lane of a wave is executing a different move, and SIMT only reconverges at common program
  // points.  So instead of ~60 effect bodies that each call the accuracy check / secondary roll / status /
  // stat-stage code (ten copies of each, executed one after another by whichever lanes sit in them), the
  // effect-specific part only computes small decisions (gates, parameters) and all lanes meet at ONE
  // accuracy site, ONE secondary-roll site, ONE status site, ONE confusion site, ONE stat-stage site.
  // Per-lane RNG order is unchanged (accuracy -> effect rolls -> crit -> damage -> counts -> secondary
  // chance -> secondary duration), so results stay bit-identical to gen1_device.hpp and the oracle.
  __device__ void run_move(uint32_t mslot) {
    OAK_SCOPE(PS_RUN_MOVE);
    OAK_T0(t_g);
    const uint32_t move_id = last_sel(S);
    const Move mv = move_data(move_id);
    const uint32_t eff = mv.effect(), mtype = mv.type();
    const uint32_t sap = absp(S), fap = sap ^ 1;
    const bool damaging = mv.bp() != 0;
    set_lm_counterable(sap, 0);
    const uint32_t fs = status(F), ft = types(F);
    const bool type_immune = chart(mtype, ft & 15) == 0 || chart(mtype, ft >> 4) == 0;
    const bool fixed = eff == E_SpecialDamage || eff == E_SuperFang || move_id == M_Counter;
    const bool ohko = eff == E_OHKO;
    // deferred, single-site actions
    uint32_t want_status = 0;
    bool want_conf = false;
    int ub_idx = -1, ub_n = 1, b_idx = -1, b_n = 1;
    uint32_t sec_kind = SEC_NONE, sec_chance = 0, sec_status = 0;
    int sec_idx = 0;

    // -- stage 1: gates ----------------------------------------------------------------------------
    bool go = true, need_hit = false;
    const uint32_t fx = T.fx[eff]; // per-effect descriptor (gen1_device.hpp, fx_desc)
    if (!damaging) { // branch-free: the gate kind selects among conditions every lane can evaluate
      last_damage = 0;
      const uint32_t gate = fx & 15;
      const bool sub = (F.vlo & V_SUBSTITUTE) != 0;
      const bool sure = gate == G_SLEEP && (F.vlo & V_RECHARGING); // sleep always lands on a recharging target
      F.vlo &= sure ? ~V_RECHARGING : ~0u;
      go = gate == G_SUB ? !sub
         : gate == G_INVUL ? !(F.vlo & V_INVULNERABLE)
         : gate == G_GRASS ? !has_type(ft, T_Grass)
         : gate == G_PAR ? (fs == 0 && !type_immune)
         : gate == G_PSN ? (fs == 0 && !has_type(ft, T_Poison) && !sub)
         : gate == G_SLEEP ? (sure ? !(fs & ST_SLP) : fs == 0)
         : gate == G_DISABLE ? disable_move(F) == 0
         : true;
      need_hit = gate == G_TELE ? move_id != M_Teleport : (gate != G_NONE && gate != G_INVUL && !sure);
    } else {
      bool immune = !fixed && type_immune;
      if (eff == E_DreamEater && !(fs & ST_SLP)) immune = true;
      if (ohko && spe(S) < spe(F)) immune = true;
#if OAK_COUNTER_SHOWDOWN
      if (move_id == M_Counter) {
        const uint32_t lu = last_used(F), ls = last_sel(F);
        const Move mu = move_data(lu), ms = move_data(ls);
        const bool cu = lu != 0 && lu != M_Counter && mu.bp() > 0 && (mu.type() == T_Normal || mu.type() == T_Fighting);
        const bool cs = ls != 0 && ls != M_Counter && ms.bp() > 0 && (ms.type() == T_Normal || ms.type() == T_Fighting);
        if (!(cu && cs) || last_damage == 0) immune = true;
      }
#else
      if (move_id == M_Counter && (!lm_counterable(fap) || last_damage == 0)) immune = true;
#endif
      go = !immune;
      need_hit = !(OAK_ACCURACY_LAST && !fixed && !ohko); // (cartridge order: the accuracy roll behind crit and damage roll, stage 4)
    }
    // -- stage 2: THE accuracy check ------------------------------------------------------------------
    bool hit = true;
    if (go && need_hit) hit = move_hit(mv);
#if OAK_MULTIHIT_ROLL_FIRST
    if (eff == E_MultiHit && go && hit) { // Showdown order: the count right behind the accuracy check, before crit / damage
      const uint32_t h = (0x54333222u >> (4 * rng_range(0, 8))) & 15; // {2,2,2,3,3,3,4,5}
      act_set(true, AC_MULTIHIT, 4, h);
      S.misc = (S.misc & 0x00FFFFFFu) | (h << 24); // parked in the spare top byte of the mover's side word until the hit loop
    }
#endif
    OAK_T1(PS_GATES_HIT, t_g);
    if (!(go && hit)) {
      if (damaging) {
        last_damage = 0;
        clear_binding(S);
        if (eff == E_Explode) { set_hp(S, 0); set_status(S, 0); }
        if (eff == E_JumpKick && go) { const uint32_t h = hp(S); if (h > 0) set_hp(S, h - 1); } // crash: 1 HP
      }
      return;
    }
    // -- stage 3: effect bodies (no accuracy checks, no shared machinery inside) ------------------------
    if (!damaging) {
      OAK_SCOPE(PS_STATUS_BODIES);
      const uint32_t cls = (fx >> 4) & 7, par = fx >> 7;
      // the light effects are pure data: a stage change, a volatile bit, paralysis, confusion
      if (cls == A_BOOST) { b_idx = (int)(par & 7); b_n = (int)(par >> 3) + 1; }
      if (cls == A_UNBOOST) { ub_idx = (int)(par & 7); ub_n = (int)(par >> 3) + 1; }
      S.vlo |= cls == A_SVOL ? 1u << par : 0u; // FocusEnergy, LightScreen, Reflect, Mist
      F.vlo |= cls == A_FVOL ? 1u << par : 0u; // LeechSeed (already seeded: no-op)
      want_status = cls == A_PAR ? (uint32_t)ST_PAR : 0u;
      want_conf = cls == A_CONF && !(F.vlo & V_CONFUSION);
      if (cls == A_HEAVY) { OAK_SCOPE(PS_HEAVY); switch (eff) {
      case E_Conversion: { OAK_SCOPE(PS_H_CONVERSION); } S.a2 = (S.a2 & 0x00FFFFFFu) | (F.a2 & 0xFF000000u); break;
      case E_Haze: {
        OAK_SCOPE(PS_H_HAZE);
        S.bo = 0;
        F.bo = 0;
        unmodified_to_active(S);
        unmodified_to_active(F);
        if (fs) {
          if (fs & ST_SLP) dset(F, 0, 3, 0);
          set_status(F, 0);
        }
        if (status(S) == ST_TOX) set_status(S, ST_PSN);
        haze_clear(S);
        haze_clear(F);
        break;
      }
      case E_Heal: {
        OAK_SCOPE(PS_H_HEAL);
        const uint32_t mx = maxhp(S), h = hp(S), delta = mx - h;
        if (delta == 0 || (delta & 255) == 255) break; // gen-1 recovery failure glitch
        if (move_id == M_Rest) {
          set_status(S, ST_EXT | 2);
          dset(S, 0, 3, 0);
          set_hp(S, mx);
          S.vlo &= ~V_TOXIC;
          set_toxic_ctr(S, 0);
        } else {
          const uint32_t nh = h + mx / 2;
          set_hp(S, nh > mx ? mx : nh);
        }
        break;
      }
      case E_Mimic: {
        OAK_SCOPE(PS_H_MIMIC);
        uint32_t n = 0;
        for (uint32_t i = 1; i <= 4; ++i) n += (active_move(F, i) & 0xFF) != 0;
        if (n == 0 || mslot == 0) break;
        const uint32_t r = rng_range(0, n);
        act_set(true, AC_MOVESLOT, 4, r + 1);
        const uint64_t nid = active_move(F, r + 1) & 0xFF;
        const uint32_t sh = 16 * (mslot - 1);
        set_amoves(S, (amoves(S) & ~(0xFFull << sh)) | (nid << sh));
        break;
      }
      case E_Poison: {
        OAK_SCOPE(PS_H_POISON);
        if (move_id == M_Toxic) { set_status(F, ST_TOX); F.vlo |= V_TOXIC; set_toxic_ctr(F, 0); }
        else set_status(F, ST_PSN);
        break;
      }
      case E_Substitute: {
        OAK_SCOPE(PS_H_SUBSTITUTE);
        if (S.vlo & V_SUBSTITUTE) break;
        const uint32_t cost = maxhp(S) / 4, h = hp(S);
        if (h < cost) break;
        set_hp(S, h - cost); // exactly a quarter left: the user faints (gen-1 behaviour)
        set_sub_hp(S, cost + 1);
        S.vlo |= V_SUBSTITUTE;
        break;
      }
      case E_Transform: {
        OAK_SCOPE(PS_H_TRANSFORM);
        const uint32_t id = (F.vlo & V_TRANSFORM) ? transform_id(F) : ((absp(F) << 3) | order0(F));
        S.vlo |= V_TRANSFORM;
        set_transform_id(S, id);
        S.a0 = F.a0; S.a1 = F.a1; S.a2 = F.a2; S.bo = F.bo;
        const uint64_t fm = amoves(F);
        uint64_t nm = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const uint64_t mid = (fm >> (16 * i)) & 0xFF;
          nm |= (mid | ((mid ? 5ull : 0ull) << 8)) << (16 * i);
        }
        set_amoves(S, nm);
        break;
      }
      case E_Bide: {
        OAK_SCOPE(PS_H_BIDE);
        S.vlo |= V_BIDE;
        set_vstate(S, 0);
        set_attacks(S, rng_range(2, 4));
        dset(S, 25, 3, 1);
        act_set(true, AC_ATTACKING, 2, OBS_STARTED);
        break;
      }
      case E_Sleep: {
        OAK_SCOPE(PS_H_SLEEP);
        set_status(F, rng_range(1, 8));
        dset(F, 0, 3, 1);
        act_set(false, AC_SLEEP, 2, OBS_STARTED);
        break;
      }
      case E_Disable: {
        OAK_SCOPE(PS_H_DISABLE);
        uint32_t n = 0, packed = 0;
        for (uint32_t i = 1; i <= 4; ++i) {
          const uint32_t ms = active_move(F, i);
          if ((ms & 0xFF) && (ms >> 8)) { packed |= i << (4 * n); ++n; }
        }
        if (n == 0) break;
        const uint32_t slot = (packed >> (4 * rng_range(0, n))) & 15;
        act_set(true, AC_MOVESLOT, 4, slot);
        set_disable_move(F, slot);
        set_disable_left(F, rng_range(1, 9));
        dset(F, 21, 4, 1);
        act_set(false, AC_DISABLE, 2, OBS_STARTED);
        break;
      }
      default: break;
      } }
    } else {
      // -- stage 4: damage ---------------------------------------------------------------------------
      OAK_SCOPE(PS_DAMAGE);
      OAK_T0(t_cd);
      if (fixed) {
        uint32_t dd;
        if (move_id == M_Counter) { dd = last_damage * 2; if (dd > 65535) dd = 65535; }
        else if (eff == E_SuperFang) { dd = hp(F) / 2; if (dd < 1) dd = 1; }
        else if (move_id == M_SonicBoom) dd = 20;
        else if (move_id == M_DragonRage) dd = 40;
        else if (move_id == M_Psywave) {
          const uint32_t max = level(S) * 3 / 2;
#if OAK_PSYWAVE_SHOWDOWN
          dd = rng_range(0, max ? max : 1); // Showdown: random(0, max), a 0 fails the move; the action holds the roll + 1
          act_set(true, AC_PSYWAVE, 8, dd + 1);
          if (dd == 0) { last_damage = 0; clear_binding(S); return; }
#else
          dd = max <= 1 ? 1 : rng_range(1, max);
          act_set(true, AC_PSYWAVE, 8, dd);
#endif
        } else dd = level(S); // SeismicToss, NightShade
        last_damage = dd;
      } else if (ohko) {
        last_damage = 65535;
      } else {
        const bool crit = check_crit(mv);
        if (!calc_damage(F, mv.bp(), mtype, eff == E_Explode, crit)) return;
        adjust_damage(mv);
        randomize_damage();
        if (last_damage == 0) { clear_binding(S); return; } // rounded down to nothing
#if OAK_ACCURACY_LAST
        if (!move_hit(mv)) {
          last_damage = 0;
          clear_binding(S);
          if (eff == E_Explode) { set_hp(S, 0); set_status(S, 0); }
          if (eff == E_JumpKick) { const uint32_t h = hp(S); if (h > 0) set_hp(S, h - 1); }
          return;
        }
#endif
      }
      uint32_t hits = 1;
      if (eff == E_DoubleHit || eff == E_Twineedle) hits = 2;
#if OAK_MULTIHIT_ROLL_FIRST
      else if (eff == E_MultiHit) hits = S.misc >> 24;
#else
      else if (eff == E_MultiHit) {
        hits = (0x54333222u >> (4 * rng_range(0, 8))) & 15; // {2,2,2,3,3,3,4,5}
        act_set(true, AC_MULTIHIT, 4, hits);
      }
#endif
      OAK_T1(PS_CALC_DAMAGE, t_cd);
      OAK_T0(t_ah);
      bool broke = false, hit_sub = false;
      uint32_t dealt = 0, rage_hits = 0;
      const uint32_t per_hit = last_damage;
      for (uint32_t h = 0; h < hits; ++h) {
        last_damage = per_hit;
        broke = apply_damage(F, F, hit_sub);
        dealt = last_damage;
        if (!hit_sub) {
          if (F.vlo & V_BIDE) set_vstate(F, (vstate(F) + dealt) & 0xFFFF);
          if ((F.vlo & V_RAGE) && hp(F) > 0) ++rage_hits;
        }
        if (broke || hp(F) == 0) break;
      }
      // rage builds once per hit taken; nothing between the hits reads the attacker's stats, so the stage
      // changes (and their stat-modification side effect on S) can be applied after the loop
      OAK_T1(PS_APPLY_HITS, t_ah);
      OAK_SCOPE(PS_DAMAGE_TAIL);
      for (uint32_t k = 0; k < rage_hits; ++k) (void)boost_side(F, S, 0, 1);
      set_lm_counterable(sap, (mtype == T_Normal || mtype == T_Fighting) && move_id != M_Counter);
      if (eff == E_Explode && !broke) { set_hp(S, 0); set_status(S, 0); }
      if (eff == E_Recoil && !broke && dealt > 0) {
        uint32_t r = dealt / (move_id == M_Struggle ? 2u : 4u); if (r < 1) r = 1;
        const uint32_t h = hp(S);
        set_hp(S, r > h ? 0 : h - r);
      }
      if ((eff == E_DrainHP || eff == E_DreamEater) && dealt > 0) {
        uint32_t h = dealt / 2; if (h < 1) h = 1;
        h += hp(S);
        const uint32_t mx = maxhp(S);
        set_hp(S, h > mx ? mx : h);
      }
      if (hp(F) == 0 || broke) return; // no secondary effects, no recharge, no binding
      if (eff == E_HyperBeam) { S.vlo |= V_RECHARGING; return; }
      if (eff == E_Binding) {
        if (!(S.vlo & V_BINDING)) {
          const uint32_t n = (0x54333222u >> (4 * rng_range(0, 8))) & 15;
          S.vlo |= V_BINDING;
          set_attacks(S, n - 1);
          dset(S, 28, 3, 1);
          act_set(true, AC_BINDING, 3, OBS_STARTED);
        }
        return;
      }
      if (hit_sub) return; // a standing substitute blocks every secondary effect
      // secondary-effect parameters from the descriptor; the roll itself happens at the shared site below
      const uint32_t sk = fx & 7, sp = (fx >> 6) & 7;
      sec_chance = (uint32_t)(FX_CHANCES >> (8 * ((fx >> 3) & 7))) & 0xFF;
      sec_idx = (int)sp;
      sec_status = 8u << sp; // PSN BRN FRZ PAR
      sec_kind = sk;
      if (sk == SEC_STATUS) {
        const uint32_t fs2 = status(F);
        const bool thaw = sec_status == ST_BRN && (fs2 & ST_FRZ); // fire thaws instead of burning
        if (thaw) set_status(F, 0);
        if (thaw || fs2 != 0 || has_type(types(F), sec_status == ST_PSN ? (uint32_t)T_Poison : mtype)) sec_kind = SEC_NONE;
      }
      if (sk == SEC_CONF && (F.vlo & V_CONFUSION)) sec_kind = SEC_NONE;
    }
    // -- stage 5: THE secondary-effect roll ------------------------------------------------------------
    OAK_SCOPE(PS_SECONDARY_APPLY);
    if (sec_kind != SEC_NONE) {
      const bool proc = rng_chance(sec_chance);
      act_bool(true, AC_SECONDARY, proc);
      if (proc) {
        if (sec_kind == SEC_STATUS) want_status = sec_status;
        else if (sec_kind == SEC_FLINCH) F.vlo |= V_FLINCH;
        else if (sec_kind == SEC_CONF) want_conf = true;
        else { ub_idx = sec_idx; ub_n = 1; }
      }
    }
    // -- stage 6: single sites for status / confusion / stat stages ---------------------------------------
    if (want_status) {
      set_status(F, want_status);
      if (want_status == ST_PAR) { const uint32_t sp2 = spe(F) / 4; set_astat(F, 2, sp2 < 1 ? 1 : sp2); }
      if (want_status == ST_BRN) { const uint32_t at2 = (F.a0 >> 16) / 2; set_astat(F, 0, at2 < 1 ? 1 : at2); }
    }
    if (want_conf) start_confusion(F, false);
    if (ub_idx >= 0) (void)unboost_foe(ub_idx, ub_n);
    if (b_idx >= 0) (void)boost_side(S, F, b_idx, b_n);
  }

  __device__ void execute_selected(uint32_t mslot, bool skip_can, bool skip_pp) {
    const uint32_t sap = absp(S);
    OAK_T0(t_pre);
    if (!skip_can) {
#pragma unroll 1
      for (int depth = 0; depth < 4; ++depth) {
        const uint32_t move_id = last_sel(S);
        const Move mv = move_data(move_id);
        const uint32_t eff = mv.effect();
        if (S.vlo & V_CHARGING) {
          S.vlo &= ~(V_CHARGING | V_INVULNERABLE);
        } else if (eff == E_Charge) {
          S.vlo |= V_CHARGING;
          if (move_id == M_Fly || move_id == M_Dig) S.vlo |= V_INVULNERABLE;
          set_last_used(S, move_id);
          set_lm_counterable(sap, 0);
          return;
        }
        set_last_used(S, move_id);
        set_lm_counterable(sap, 0);
        if (!skip_pp) decrement_pp(mslot);
        skip_pp = true;
        if (eff == E_Metronome) {
          const uint32_t r = rng_range(0, 163);
          const uint32_t pick = (r + 1 >= M_Metronome) ? r + 2 : r + 1;
          act_set(true, AC_METRONOME, 8, pick);
          set_last_sel(S, pick);
          continue;
        }
        if (eff == E_MirrorMove) {
          const uint32_t mm = last_used(F);
          if (mm == 0 || mm == M_MirrorMove) { last_damage = 0; return; }
          set_last_sel(S, mm);
          continue;
        }
        if (eff >= E_Confusion && eff <= E_Transform) break;
        if (eff == E_Thrashing) {
          S.vlo |= V_THRASHING;
          set_attacks(S, rng_range(2, 4));
          dset(S, 25, 3, 1);
          act_set(true, AC_ATTACKING, 2, OBS_STARTED);
        } else if (eff == E_Rage) {
          S.vlo |= V_RAGE;
        }
        break;
      }
    }
    OAK_T1(PS_EXEC_SELECTED_PRE, t_pre);
    run_move(mslot);
  }

  __device__ void handle_residual() {
    uint32_t h = hp(S);
    if (h == 0) return;
    const uint32_t mx = maxhp(S);
    if (status(S) & (ST_BRN | ST_PSN)) {
      uint32_t dmg = mx / 16; if (dmg < 1) dmg = 1;
      if (S.vlo & V_TOXIC) { const uint32_t t = (toxic_ctr(S) + 1) & 31; set_toxic_ctr(S, t); dmg *= t; }
      h = dmg > h ? 0 : h - dmg;
      set_hp(S, h);
      if (h == 0) return;
    }
    if (S.vlo & V_LEECHSEED) {
      uint32_t dmg = mx / 16; if (dmg < 1) dmg = 1;
      if (S.vlo & V_TOXIC) { const uint32_t t = (toxic_ctr(S) + 1) & 31; set_toxic_ctr(S, t); dmg *= t; }
      h = dmg > h ? 0 : h - dmg;
      set_hp(S, h);
      const uint32_t fh = hp(F);
      if (fh > 0) {
        const uint32_t nh = fh + dmg, fm = maxhp(F);
        set_hp(F, nh > fm ? fm : nh);
      }
    }
  }

  // side x fainted, y is its foe
  __device__ __forceinline__ void faint(SideR &x, SideR &y) {
    y.vlo &= ~V_MULTIHIT;
    if (y.vlo & V_BIDE) set_vstate(y, 0);
    x.vlo = 0;
    x.vhi = 0;
    set_last_used(x, 0);
    set_status(x, 0);
    x.misc &= ~1u; // the active (order position 0) is no longer alive
    clear_binding(y);
  }
  // result if side x (foe y) has fainted, 0 otherwise
  __device__ __forceinline__ uint32_t check_faint(SideR &x, SideR &y) {
    if (hp(x) > 0) return 0;
    const bool foe_fainted = hp(y) == 0;
    faint(x, y);
    if (foe_fainted) faint(y, x);
    const bool x_out = (x.misc & 63) == 0, y_out = (y.misc & 63) == 0;
    const uint32_t xp = absp(x);
    if (x_out && y_out) return mk_result(R_TIE, 0, 0);
    if (x_out) return mk_result(xp == 0 ? R_LOSE : R_WIN, 0, 0);
    if (y_out) return mk_result(xp == 0 ? R_WIN : R_LOSE, 0, 0);
    const uint32_t fc = foe_fainted ? C_SWITCH : C_PASS;
    return xp == 0 ? mk_result(0, C_SWITCH, fc) : mk_result(0, fc, C_SWITCH);
  }
  __device__ __forceinline__ uint32_t end_turn() {
    turn += 1;
    if (turn >= 1000) return mk_result(R_TIE, 0, 0);
    return mk_result(0, C_MOVE, C_MOVE);
  }

  // frame normalisation: afterwards S = P1, F = P2
  __device__ __forceinline__ void normalize() {
    const bool flipped = absp(S) != 0;
    cswap_sides(flipped, S, F);
    if constexpr (TRACK_ACTIONS) { uint64_t t = flipped ? actF : actS; actF = flipped ? actS : actF; actS = t; }
  }

  // ---- one player's action = pre-state, the action itself, faint / residual checks.  Returns the turn's
  // result if the action ended it (a faint -> forced switch or game over, or an engine error), else 0.
  // Switches and passes are the cheap actions; moves go through before_move / execute_selected / run_move.
  __device__ __forceinline__ uint32_t post_action(uint32_t choice, bool replace, bool residual) {
    OAK_SCOPE(PS_FAINT_RESIDUAL);
    if (replace) return 0;
    uint32_t r = 0;
    if ((choice & 3) != C_SWITCH) r = check_faint(F, S);
    if (r == 0) {
      if (residual) handle_residual();
      r = check_faint(S, F);
    }
    return r;
  }
  __device__ __forceinline__ uint32_t act_cheap(uint32_t choice) { // C_SWITCH or C_PASS
    const bool replace = hp(S) == 0;
    if ((choice & 3) == C_SWITCH) { OAK_SCOPE(PS_SWITCH_IN); switch_in(S, F, choice >> 2); }
    return post_action(choice, replace, false);
  }
  __device__ __forceinline__ uint32_t act_move(uint32_t choice) { // C_MOVE
    OAK_T0(t_em);
    const bool replace = hp(S) == 0;
    uint32_t mslot = choice >> 2;
    if (last_sel(S) == M_Struggle) mslot = 0;
    else if (mslot == 0) mslot = lm_index(absp(S));
    OAK_T0(t_bm);
    const int r = before_move();
    OAK_T1(PS_BEFORE_MOVE, t_bm);
    if (r == BM_ERR) return mk_result(R_ERROR, 0, 0);
    if (r != BM_DONE) execute_selected(mslot, r == BM_SKIP_CAN, r == BM_SKIP_PP);
    OAK_T1(PS_EXEC_MOVE, t_em);
    return post_action(choice, replace, true);
  }

  // ---- pkmn_gen1_battle_update in FRAME terms: cS / cF are the choices of whoever currently sits in S / F
  // (absp() tells which player that is).  The frame is left wherever the turn ends -- callers that need
  // S = P1 call normalize().  Skipping the per-turn normalisation saves a 54-instruction conditional swap.
  //
  // turn_prologue: selection + who acts first; leaves the first actor in S and returns its choice in `pc`,
  // the second actor's in `qc`.  A non-zero return value is the result of a turn that is already over
  // (turn 0: both leads are sent out).
  __device__ __forceinline__ uint32_t turn_prologue(uint32_t cS, uint32_t cF, uint32_t &pc, uint32_t &qc) {
    if constexpr (TRACK_ACTIONS) { actS = 0; actF = 0; }
    if (turn == 0) {
      const bool aS = (S.misc & 63) != 0, aF = (F.misc & 63) != 0;
      const bool s_p1 = absp(S) == 0;
      if (!aS) return mk_result(aF ? (s_p1 ? R_LOSE : R_WIN) : R_TIE, 0, 0);
      if (!aF) return mk_result(s_p1 ? R_WIN : R_LOSE, 0, 0);
#pragma unroll 1
      for (int k = 0; k < 2; ++k) { switch_in(S, F, 1); swap_sides(S, F); } // both leads, one copy of the code
      return end_turn();
    }
    OAK_T0(t_ord);
    select_move(S, cS);
    select_move(F, cF);
    bool f_first;
    {
      const uint32_t tS = cS & 3, tF = cF & 3;
      if (tS == C_PASS) f_first = true;
      else if (tF == C_PASS) f_first = false;
      else if ((tS == C_SWITCH) != (tF == C_SWITCH)) f_first = tS != C_SWITCH;
      else {
        const uint32_t mS = last_sel(S), mF = last_sel(F);
        bool decided = false;
        f_first = false;
        if (tS == C_MOVE) {
          if ((mS == M_QuickAttack) != (mF == M_QuickAttack)) { f_first = mS != M_QuickAttack; decided = true; }
          else if ((mS == M_Counter) != (mF == M_Counter)) { f_first = mS == M_Counter; decided = true; }
        }
        if (!decided) {
          const uint32_t sS = spe(S), sF = spe(F);
          if (sS == sF) {
            const bool p1 = rng_range(0, 2) == 0; // P1 moves first on 0
            act_set(true, AC_SPEEDTIE, 2, p1 ? 1 : 2);
            act_set(false, AC_SPEEDTIE, 2, p1 ? 1 : 2);
            f_first = p1 != (absp(S) == 0);
          } else f_first = sS < sF;
        }
      }
    }
    pc = f_first ? cF : cS;
    qc = f_first ? cS : cF;
    cswap_sides(f_first, S, F);
    if constexpr (TRACK_ACTIONS) { uint64_t t = f_first ? actF : actS; actF = f_first ? actS : actF; actS = t; }
    OAK_T1(PS_ORDER, t_ord);
    return 0;
  }
  __device__ __forceinline__ uint32_t turn_epilogue() {
    if ((S.vlo & V_BINDING) && attacks(S) == 0) clear_binding(S);
    if ((F.vlo & V_BINDING) && attacks(F) == 0) clear_binding(F);
    return end_turn();
  }
  __device__ __forceinline__ void next_actor() {
    swap_sides(S, F);
    if constexpr (TRACK_ACTIONS) { uint64_t t = actS; actS = actF; actF = t; }
  }

  __device__ uint32_t update_frame(uint32_t cS, uint32_t cF) {
    uint32_t pc = 0, qc = 0;
    uint32_t r = turn_prologue(cS, cF, pc, qc);
    if (r) return r;
    // Both actions run through ONE copy of the action code.  The sides are swapped after EVERY action, whether
    // or not a second one follows (the turn's end is frame-agnostic): an unconditional swap keeps the register
    // roles of S and F identical on every path through the loop, where a conditional one makes the compiler
    // copy the whole engine state at each merge point.
#pragma unroll 1
    for (int k = 0; k < 2; ++k) {
      r = (pc & 3) == C_MOVE ? act_move(pc) : act_cheap(pc);
      const bool last = r != 0 || (qc & 3) == C_PASS;
      next_actor();
      const uint32_t t = pc; pc = qc; qc = t;
      if (last) break;
    }
    return r ? r : turn_epilogue();
  }
  // normalised-frame form (S = P1, F = P2 before and after)
  __device__ __forceinline__ uint32_t update(uint32_t c1, uint32_t c2) {
    const uint32_t r = update_frame(c1, c2);
    normalize();
    return r;
  }

  // ---- pkmn_gen1_battle_choices from registers, as (count, k-th choice) instead of a materialised list:
  // the rollout only ever needs `choices[seed % n]`.  Order = the reference's: switches by slot, then moves.
  struct Legal { uint32_t n, sw, mv, forced; }; // sw: bit s-2 for party position s; mv: bit i-1 for move slot i
  __device__ __forceinline__ Legal legal(const SideR &x, uint32_t request) const {
    const uint32_t sw = (x.misc >> 1) & 31;
    if (request == C_PASS) return Legal{1, 0, 0, 0};
    if (request == C_SWITCH) {
      const uint32_t n = (uint32_t)__popc(sw);
      return n ? Legal{n, sw, 0, 0} : Legal{1, 0, 0, 0};
    }
    if (x.vlo & (V_RECHARGING | V_RAGE | V_THRASHING | V_CHARGING)) return Legal{1, 0, 0, C_MOVE};
    const uint64_t am = amoves(x);
    if (x.vlo & (V_BIDE | V_BINDING)) {
      const uint32_t sel = last_sel(x);
      uint32_t forced = C_MOVE;
#pragma unroll
      for (int i = 3; i >= 0; --i) {
        const uint32_t id = (uint32_t)(am >> (16 * i)) & 0xFF;
        if (id != 0 && id == sel) forced = ((uint32_t)(i + 1) << 2) | C_MOVE;
      }
      return Legal{1, 0, 0, forced};
    }
    const uint32_t dm = disable_move(x);
    uint32_t mv = 0;
    bool open = true;
#pragma unroll
    for (uint32_t i = 0; i < 4; ++i) {
      const uint32_t ms = (uint32_t)(am >> (16 * i)) & 0xFFFF;
      open = open && (ms & 0xFF) != 0;
      mv |= (open && (ms >> 8) != 0 && dm != i + 1 ? 1u : 0u) << i;
    }
    const uint32_t nm = (uint32_t)__popc(mv);
    return Legal{(uint32_t)__popc(sw) + (nm ? nm : 1u), sw, mv, C_MOVE}; // no usable move: Struggle (move, data 0)
  }
  static __device__ __forceinline__ uint32_t kth_bit(uint32_t mask, uint32_t k) { // index of the k-th set bit (mask < 32)
    uint32_t idx = 0, seen = 0;
#pragma unroll
    for (uint32_t b = 0; b < 5; ++b) {
      const uint32_t bit = (mask >> b) & 1;
      idx = (bit && seen == k) ? b : idx;
      seen += bit;
    }
    return idx;
  }
  __device__ __forceinline__ uint32_t nth_choice(const Legal &L, uint32_t r) const {
    if (L.n == 1 && L.sw == 0 && L.mv == 0) return L.forced;
    const uint32_t nsw = (uint32_t)__popc(L.sw);
    if (r < nsw) return ((kth_bit(L.sw, r) + 2) << 2) | C_SWITCH;
    if (L.mv == 0) return C_MOVE;
    return ((kth_bit(L.mv, r - nsw) + 1) << 2) | C_MOVE;
  }
  // the reference's draw (mcts.h:452-476): P1 takes seed % m on the 64-bit seed, P2 (seed >> 32) % n
  __device__ __forceinline__ uint32_t draw_index(bool is_p1, uint32_t hi, uint32_t lo, uint32_t n) const {
    if (n == 1) return 0;
    uint32_t a = fast_mod(hi, n);
    if (is_p1) {
      uint32_t two32 = fast_mod(0xFFFFFFFFu, n) + 1; // 2^32 mod n
      two32 = two32 == n ? 0 : two32;
      a = fast_mod(a * two32 + fast_mod(lo, n), n);
    }
    return a;
  }
  // Is this position a frozen-versus-frozen standstill whose every further turn-step is the same turn-step?  (k_rollout_queue
  // skips all but the last of them.)  Both sides: request Move, status exactly FRZ, hp > 0, no Leech Seed and no binding volatile
  // (the first drains a Pokemon that cannot move, the second is cleared by the turn's epilogue and would free the side's choices),
  // and NO WAY OUT: either the active is the side's last Pokemon, or it is locked into its move (recharging, Rage, thrashing,
  // charging, Bide: `legal` offers the forced move only, and none of these counters runs while before_move returns at the freeze
  // check in front of them, so the lock never ends).  And speeds that differ (no tie draw).  Then update_frame does: select_move
  // x 2 (a locked side: nothing; else clears the flinch bit and writes last selected move and last move index from the choice),
  // an order decision that draws nothing, before_move x 2 -> `status & FRZ`: last used move = 0, done; no residual damage (FRZ
  // excludes PSN / BRN, no Leech Seed), nobody faints; turn + 1, tie at 1,000.  Everything it writes is either the same every
  // time or overwritten by the next turn-step before it is read.
  __device__ __forceinline__ bool frozen_standstill(uint32_t result) const {
    constexpr uint32_t locked = V_RECHARGING | V_RAGE | V_THRASHING | V_CHARGING | V_BIDE;
    const bool stuckS = (S.misc & 63) == 1 || (S.vlo & locked) != 0, stuckF = (F.misc & 63) == 1 || (F.vlo & locked) != 0;
    return result == mk_result(0, C_MOVE, C_MOVE) && turn >= 1 && turn < 1000 &&
           status(S) == ST_FRZ && status(F) == ST_FRZ && stuckS && stuckF && (S.misc & 1) && (F.misc & 1) &&
           ((S.vlo | F.vlo) & (V_BINDING | V_LEECHSEED)) == 0 && spe(S) != spe(F) && hp(S) > 0 && hp(F) > 0;
  }
  // one random-policy turn-step of the rollout (choices x2 + update), frame-agnostic
  __device__ __forceinline__ uint32_t random_step(uint32_t result, uint32_t hi, uint32_t lo) {
    OAK_SCOPE(PS_STEP);
    OAK_T0(t_ld);
    const bool s_p1 = absp(S) == 0;
    const uint32_t req1 = (result >> 4) & 3, req2 = (result >> 6) & 3;
    const Legal LS = legal(S, s_p1 ? req1 : req2), LF = legal(F, s_p1 ? req2 : req1);
    const uint32_t cS = nth_choice(LS, draw_index(s_p1, hi, lo, LS.n));
    const uint32_t cF = nth_choice(LF, draw_index(!s_p1, hi, lo, LF.n));
    OAK_T1(PS_LEGAL_DRAW, t_ld);
    return update_frame(cS, cF);
  }

  // ---- pkmn_gen1_battle_choices from registers (side x) -------------------------------------------
  struct Choices { uint32_t n; uint64_t lo; uint32_t hi; // up to 9 choice bytes, shift-indexed (no field select)
    __device__ __forceinline__ void push(uint32_t c) {
      lo |= n < 8 ? (uint64_t)c << (8 * n) : 0ull;
      hi |= n >= 8 ? c : 0u;
      ++n;
    }
    __device__ __forceinline__ uint32_t get(uint32_t i) const {
      const uint32_t a = (uint32_t)(lo >> (8 * (i & 7))) & 0xFF;
      return i < 8 ? a : hi;
    }
  };
  __device__ __forceinline__ Choices choices(const SideR &x, uint32_t request) const {
    Choices c{0, 0, 0};
    if (request == C_PASS) { c.push(0); return c; }
    const uint32_t alive = x.misc & 63;
    if (request == C_SWITCH) {
#pragma unroll
      for (uint32_t slot = 2; slot <= 6; ++slot)
        if ((alive >> (slot - 1)) & 1) c.push((slot << 2) | C_SWITCH);
      if (c.n == 0) c.push(0);
      return c;
    }
    if (x.vlo & (V_RECHARGING | V_RAGE | V_THRASHING | V_CHARGING)) { c.push(C_MOVE); return c; }
    if (x.vlo & (V_BIDE | V_BINDING)) {
      const uint32_t sel = last_sel(x);
#pragma unroll
      for (uint32_t i = 1; i <= 4; ++i) {
        const uint32_t ms = active_move(x, i);
        if ((ms & 0xFF) && (ms & 0xFF) == sel) { c.push((i << 2) | C_MOVE); return c; }
      }
      c.push(C_MOVE);
      return c;
    }
#pragma unroll
    for (uint32_t slot = 2; slot <= 6; ++slot)
      if ((alive >> (slot - 1)) & 1) c.push((slot << 2) | C_SWITCH);
    const uint32_t before = c.n, dm = disable_move(x);
    bool open = true;
#pragma unroll
    for (uint32_t i = 1; i <= 4; ++i) {
      const uint32_t ms = active_move(x, i);
      if ((ms & 0xFF) == 0) open = false;
      if (open && (ms >> 8) != 0 && dm != i) c.push((i << 2) | C_MOVE);
    }
    if (c.n == before) c.push(C_MOVE);
    return c;
  }
};

} // namespace oak
