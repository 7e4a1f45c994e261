// oak_amd/csrc/oakgpu.hip -- HIP kernels (gfx950) + the C ABI of include/oakgpu.h.
//
// K1  k_rollout   : batched random playouts (replaces cpp/include/search/mcts.h:448-496 and
//                   the per-iteration prep of mcts.h:250-263): one lane per playout, the
//                   384-byte battle resident in LDS (lane-interleaved) for the whole playout,
//                   tables in LDS, choice RNG (fast_prng) and durations in registers.
//     k_update    : batched pkmn_gen1_battle_update (+ chance durations / actions / calc).
//     k_choices   : batched pkmn_gen1_battle_choices.
//     k_init      : batched PKMN::battle (cpp/include/libpkmn/init.h:90-154) + opening update.
//     k_random_ou : SURVEY 8(d) config-2 input generator (teams drawn on device).
// There is no CPU fallback in this library.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <random>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/oakgpu.h"
#include "gen1_device.hpp"
#include "gen1_regs.hpp"
#include "oakgpu_internal.h"

namespace oak {

constexpr int BLOCK = 256;      // 4 waves: one per SIMD of a CU
constexpr int STATE_WORDS = 96; // 384-byte battle
constexpr int STATE_LDS_BYTES = STATE_WORDS * BLOCK * 4;
constexpr int TABLE_LDS_PAD = (TABLE_LDS_BYTES + 15) & ~15;
constexpr int ENGINE_LDS_BYTES = STATE_LDS_BYTES + TABLE_LDS_PAD;

// The engine's tables (moves, species, type chart, boosts, reciprocals, effect descriptors: 3.4 KB) as ONE image in exactly
// their LDS layout, built once per device by k_build_table_image (oakgpu_create).  A kernel's prologue is then a straight
// copy with all of a thread's loads in flight at once -- built table by table it was a dozen load -> wait -> store round
// trips plus four integer divisions per thread in front of every launch (~10 us of the 61 us a one-turn launch takes).
constexpr int TABLE_IMAGE_WORDS = TABLE_LDS_PAD / 4;
__device__ uint32_t g_table_image[TABLE_IMAGE_WORDS];

__global__ __launch_bounds__(64) void k_build_table_image() {
  extern __shared__ __align__(16) uint8_t smem[];
  for (int i = threadIdx.x; i < TABLE_IMAGE_WORDS; i += 64) ((lds_u32 *)smem)[i] = 0;
  __syncthreads();
  (void)stage_tables((lds_u8 *)smem, OAK_MOVE_WORDS, OAK_MOVE_MAXPP, OAK_SPECIES_W0, OAK_SPECIES_W1, OAK_TYPE_CHART, OAK_BOOSTS);
  __syncthreads();
  for (int i = threadIdx.x; i < TABLE_IMAGE_WORDS; i += 64) g_table_image[i] = ((lds_u32 *)smem)[i];
}

// call with all threads of the workgroup, then barrier (the table area is 16-byte aligned in every kernel's LDS layout)
__device__ __forceinline__ Tables stage_default_tables(lds_u8 *lds) {
  constexpr int N4 = TABLE_IMAGE_WORDS / 4;
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  typedef OAK_LDS u32x4 lds_u128;
  const u32x4 *src = (const u32x4 *)g_table_image;
  lds_u128 *dst = (lds_u128 *)lds;
  for (int base = threadIdx.x; base < N4; base += 4 * (int)blockDim.x) {
    u32x4 t[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int i = base + u * (int)blockDim.x; t[u] = src[i < N4 ? i : N4 - 1]; }
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int i = base + u * (int)blockDim.x; if (i < N4) dst[i] = t[u]; }
  }
  return tables_at(lds);
}

// AoS (n x 384 B, 4-byte aligned) <-> lane-interleaved LDS, whole workgroup cooperating.
template <int BLK = BLOCK>
__device__ __forceinline__ void load_state(lds_u32 *state, const uint8_t *battles, uint32_t base, uint32_t count) {
  const uint32_t *src = (const uint32_t *)battles + (size_t)base * STATE_WORDS;
  for (uint32_t i = threadIdx.x; i < count * STATE_WORDS; i += BLK) {
    uint32_t b = i / STATE_WORDS, w = i - b * STATE_WORDS;
    state[w * BLK + b] = src[i];
  }
}
template <int BLK = BLOCK>
__device__ __forceinline__ void store_state(const lds_u32 *state, uint8_t *battles, uint32_t base, uint32_t count) {
  uint32_t *dst = (uint32_t *)battles + (size_t)base * STATE_WORDS;
  for (uint32_t i = threadIdx.x; i < count * STATE_WORDS; i += BLK) {
    uint32_t b = i / STATE_WORDS, w = i - b * STATE_WORDS;
    dst[i] = state[w * BLK + b];
  }
}

// ---- fast_prng (cpp/include/util/random.h:67-133): 2 x u32 of state per lane ---------------
struct FastPrng {
  uint32_t s0, s1;
  __device__ __forceinline__ static uint32_t rotl(uint32_t x, int k) { return (x << k) | (x >> (32 - k)); }
  __device__ __forceinline__ uint32_t next32() {
    uint32_t result = rotl(s0 + s1, 9) + s0;
    s1 ^= s0;
    s0 = rotl(s0, 13) ^ s1 ^ (s1 << 5);
    s1 = rotl(s1, 28);
    return result;
  }
  // std::seed_seq{lo32, hi32}.generate(2 words), random.h:99-105
  __device__ void seed(uint64_t seed) {
    const uint32_t v[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    uint32_t b0 = 0x8b8b8b8bu, b1 = 0x8b8b8b8bu;
    // n = 2, s = 2, t = 0, p = q = 1, m = 3; indices alternate between the two words
#pragma unroll
    for (uint32_t k = 0; k < 3; ++k) {
      uint32_t &bk = (k & 1) ? b1 : b0, &bo = (k & 1) ? b0 : b1; // bk = b[k%2], bo = b[(k+1)%2] = b[(k-1)%2]
      uint32_t arg = bk ^ bo ^ bo;
      uint32_t r1 = 1664525u * (arg ^ (arg >> 27));
      uint32_t r2 = r1 + (k == 0 ? 2u : (k & 1) + v[k - 1 < 2 ? k - 1 : 0]);
      bo += r1;
      bo += r2;
      bk = r2;
    }
#pragma unroll
    for (uint32_t k = 3; k < 5; ++k) {
      uint32_t &bk = (k & 1) ? b1 : b0, &bo = (k & 1) ? b0 : b1;
      uint32_t arg = bk + bo + bo;
      uint32_t r3 = 1566083941u * (arg ^ (arg >> 27));
      uint32_t r4 = r3 - (k & 1);
      bo ^= r3;
      bo ^= r4;
      bk = r4;
    }
    s0 = b0;
    s1 = b1;
  }
};

__device__ __forceinline__ uint32_t mod64_small(uint32_t hi, uint32_t lo, uint32_t m) {
  // (hi * 2^32 + lo) % m for small m, 32-bit ops only
  uint32_t two32 = (0xFFFFFFFFu % m + 1) % m;
  return ((hi % m) * two32 + (lo % m)) % m;
}

// ---- MCTS::randomize_hidden_variables (cpp/include/search/durations.h:25-97) ---------------
template <class E>
__device__ void randomize_hidden(E &e) {
  // rows of the reference's 4 x 40 binding table as run lengths: {15,15,6,4} {24,8,8} {20,20} {40}
  const uint32_t hi = e.r32(B_RNG + 4), lo = e.r32(B_RNG);
  for (int s = 0; s < 2; ++s) {
    const int so = s * SIDE_SZ;
    const uint32_t d = e.dur_of(s);
    const uint32_t confusion = (d >> 18) & 7, disable = (d >> 21) & 15, attacking = (d >> 25) & 7, binding = (d >> 28) & 7;
    if (confusion) {
      uint32_t one = confusion == 1;
      uint32_t max = (6 - (confusion + one)) & 0xFF;
      e.set_conf_left(so, (mod64_small(hi, lo, max) + 1 + one) & 0xFF);
    }
    if (disable) e.set_disable_left(so, (mod64_small(hi, lo, (9 - disable) & 0xFF) + 1) & 0xFF);
    if (attacking && (e.vlo(so) & (V_BIDE | V_THRASHING)))
      e.set_attacks(so, attacking == 3 ? 1u : 4u - (attacking + mod64_small(hi, lo, 2)));
    if (binding) {
      uint32_t idx = mod64_small(hi, lo, 40), a;
      if (binding == 1) a = idx < 15 ? 1 : idx < 30 ? 2 : idx < 36 ? 3 : 4;
      else if (binding == 2) a = idx < 24 ? 1 : idx < 32 ? 2 : 3;
      else if (binding == 3) a = idx < 20 ? 1 : 2;
      else a = 1;
      e.set_attacks(so, a);
    }
    for (int i = 0; i < 6; ++i) {
      uint32_t sleep = (d >> (3 * i)) & 7;
      if (!sleep) continue;
      int pk = so + PK_SZ * ((int)e.r8(so + O_ORDER + i) - 1);
      uint32_t st = e.r8(pk + P_STATUS);
      if ((st & 7) && !(st & 0x80)) e.w8(pk + P_STATUS, (st & 0xF8) | ((mod64_small(hi, lo, (8 - sleep) & 0xFF) + 1) & 0xFF));
    }
  }
}

// ---- K1: random playouts -------------------------------------------------------------------
struct RolloutArgs {
  const uint8_t *battles;
  const uint8_t *durations;
  const uint8_t *results_in;
  uint8_t *prng;
  uint32_t n, max_steps;
  int prep;
  uint8_t *results_out;
  uint32_t *steps_out;
  float *values_out;
  uint8_t *battles_out;
  uint8_t *durations_out;
};

template <int BLK>
__global__ __launch_bounds__(BLK) void k_rollout(RolloutArgs a) {
  extern __shared__ __align__(16) uint8_t smem[];
  lds_u32 *state = (lds_u32 *)smem;
  Tables T = stage_default_tables((lds_u8 *)smem + STATE_WORDS * BLK * 4);
  const uint32_t base = blockIdx.x * BLK;
  const uint32_t count = min((uint32_t)BLK, a.n - base);
  load_state<BLK>(state, a.battles, base, count);
  __syncthreads();
  const uint32_t tid = threadIdx.x, lane = base + tid;
  if (tid < count) {
    Engine<BLK, false> e;
    e.m = state + tid;
    e.T = T;
    const uint32_t *dsrc = (const uint32_t *)a.durations + 2 * (size_t)lane;
    e.dur64 = (uint64_t)dsrc[0] | ((uint64_t)dsrc[1] << 32);
    e.over16 = 0;
    FastPrng g;
    const uint32_t *psrc = (const uint32_t *)a.prng + 2 * (size_t)lane;
    g.s0 = psrc[0];
    g.s1 = psrc[1];
    if (a.prep) { // mcts.h:254-259
      uint32_t hi = g.next32(), lo = g.next32();
      e.w32(B_RNG, lo);
      e.w32(B_RNG + 4, hi);
      randomize_hidden(e);
    }
    uint32_t result = a.results_in[lane];
    uint32_t steps = 0;
    while ((result & 15) == 0 && steps < a.max_steps) {
      const uint32_t hi = g.next32(), lo = g.next32(); // uniform_64 = hi << 32 | lo
      auto c1s = e.choices(0, (result >> 4) & 3);
      const uint32_t c1 = c1s.get(mod64_small(hi, lo, c1s.n));
      auto c2s = e.choices(1, (result >> 6) & 3);
      const uint32_t c2 = c2s.get(hi % c2s.n);
      result = e.update(c1, c2);
      ++steps;
    }
    a.results_out[lane] = (uint8_t)result;
    a.steps_out[lane] = steps;
    const uint32_t t = result & 15;
    a.values_out[lane] = t == R_WIN ? 1.0f : t == R_LOSE ? 0.0f : 0.5f;
    uint32_t *pdst = (uint32_t *)a.prng + 2 * (size_t)lane;
    pdst[0] = g.s0;
    pdst[1] = g.s1;
    if (a.durations_out) {
      uint32_t *ddst = (uint32_t *)a.durations_out + 2 * (size_t)lane;
      ddst[0] = e.dur_of(0);
      ddst[1] = e.dur_of(1);
    }
  }
  if (a.battles_out) {
    __syncthreads();
    store_state<BLK>(state, a.battles_out, base, count);
  }
}

// ---- K1 (register-resident engine, gen1_regs.hpp): same contract as k_rollout.  LDS holds only
// the party slots (72 dwords per lane) + one table image per workgroup, so 8 waves fit on a CU.
template <int BLK>
__global__ __launch_bounds__(BLK, 2) void k_rollout_regs(RolloutArgs a) {
  extern __shared__ __align__(16) uint8_t smem[];
  lds_u32 *party = (lds_u32 *)smem;
  using ER = EngineR<BLK, false>;
  Tables T = stage_default_tables((lds_u8 *)smem + ER::PARTY_WORDS * BLK * 4);
  __syncthreads();
  const uint32_t tid = threadIdx.x, lane = blockIdx.x * BLK + tid;
  if (lane >= a.n) return;
  const uint32_t *dsrc = (const uint32_t *)a.durations + 2 * (size_t)lane;
  FastPrng g;
  const uint32_t *psrc = (const uint32_t *)a.prng + 2 * (size_t)lane;
  g.s0 = psrc[0];
  g.s1 = psrc[1];
  ER e;
  e.m = party + tid;
  e.T = T;
  e.load_battle_global(a.battles + (size_t)lane * 384, dsrc[0], dsrc[1]);
  if (a.prep) { // mcts.h:254-259
    const uint32_t hi = g.next32(), lo = g.next32();
    e.rng = ((uint64_t)hi << 32) | lo;
    e.randomize_hidden();
  }
  uint32_t result = a.results_in[lane];
  uint32_t steps = 0;
  while ((result & 15) == 0 && steps < a.max_steps) {
    const uint32_t hi = g.next32(), lo = g.next32(); // uniform_64 = hi << 32 | lo
    result = e.random_step(result, hi, lo);
    ++steps;
  }
  e.normalize();
  a.results_out[lane] = (uint8_t)result;
  a.steps_out[lane] = steps;
  const uint32_t t = result & 15;
  a.values_out[lane] = t == R_WIN ? 1.0f : t == R_LOSE ? 0.0f : 0.5f;
  uint32_t *pdst = (uint32_t *)a.prng + 2 * (size_t)lane;
  pdst[0] = g.s0;
  pdst[1] = g.s1;
  if (a.durations_out) {
    uint32_t *ddst = (uint32_t *)a.durations_out + 2 * (size_t)lane;
    ddst[0] = e.S.dur;
    ddst[1] = e.F.dur;
  }
  if (a.battles_out) e.store_battle_global(a.battles_out + (size_t)lane * 384);
}

// The same, for launches of a FEW turn-steps over a resident batch (BASELINE configs[2]: one turn-step, then a leaf
// evaluation, every turn).  There the launch is all load / decode / encode / store: a lane reading its own 384-byte
// battle touches a 16-byte piece of 64 different lines per instruction, and writes it back the same way (45 us for a
// launch with ZERO steps; the HBM traffic is worth 12).  Here the wave moves its 64 battles (24 KB, contiguous) between
// global memory and LDS with fully coalesced 1 KB accesses, and the engine reads / writes the LDS copy through `gin`
// (a generic pointer: the flat loads resolve to LDS).
template <class P>
__device__ __forceinline__ P cold_ptr_at(const lds_u32 *cold, size_t byte_off) { // a 64-bit pointer parked in LDS
  return (P)((uint64_t)cold[byte_off / 4] | ((uint64_t)cold[byte_off / 4 + 1] << 32));
}
constexpr int STAGE_STRIDE = 100; // words per staged battle: 96 + 4 (16-byte aligned rows that spread over the banks)
constexpr int STAGED_COLD_BYTES = 128; // the kernel arguments, parked (below)
constexpr int STAGED_LDS_BYTES = 24 * 64 * 4 + TABLE_LDS_PAD + 64 * STAGE_STRIDE * 4 + STAGED_COLD_BYTES;
__global__ __launch_bounds__(64, 2) void k_rollout_staged(RolloutArgs a) {
  extern __shared__ __align__(16) uint8_t smem[];
  lds_u32 *party = (lds_u32 *)smem;
  using ER = EngineR<64, false, true>;
  Tables T = stage_default_tables((lds_u8 *)smem + ER::PARTY_WORDS * 64 * 4);
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  typedef OAK_LDS u32x4 lds_u128;
  lds_u32 *stage = (lds_u32 *)((lds_u8 *)smem + ER::PARTY_WORDS * 64 * 4 + TABLE_LDS_PAD);
  // the output pointers are parked in LDS across the turn loop, like k_rollout_queue's cold arguments: as kernel arguments they held
  // ~20 SGPRs the loop's nested exec masks need (54 spilled SGPRs, a v_writelane / v_readlane pair each, in a kernel that runs ONE
  // turn-step per launch in BASELINE configs[2])
  static_assert(sizeof(RolloutArgs) <= STAGED_COLD_BYTES, "parked arguments fit");
  lds_u32 *cold = stage + 64 * STAGE_STRIDE;
  if (threadIdx.x < sizeof(RolloutArgs) / 4) cold[threadIdx.x] = ((const uint32_t *)&a)[threadIdx.x];
#define SA_PTR(field, type) cold_ptr_at<type>(cold, offsetof(RolloutArgs, field))
  const uint32_t tid = threadIdx.x, base = blockIdx.x * 64, lane = base + tid;
  const uint32_t cnt = a.n - base < 64 ? a.n - base : 64; // battles of this wave
  {
    const u32x4 *src = (const u32x4 *)(a.battles + (size_t)base * 384);
    u32x4 t[24];
#pragma unroll
    for (int k = 0; k < 24; ++k) { const uint32_t i = k * 64 + tid; t[k] = src[i < cnt * 24 ? i : 0]; } // float4 i = battle i / 24, piece i % 24
#pragma unroll
    for (int k = 0; k < 24; ++k) { const uint32_t i = k * 64 + tid, b = i / 24, w = i - b * 24; *(lds_u128 *)(stage + b * STAGE_STRIDE + 4 * w) = t[k]; }
  }
  __syncthreads();
  uint32_t result = 0, steps = 0;
  FastPrng g;
  g.s0 = g.s1 = 0;
  ER e;
  e.m = party + tid;
  e.T = T;
  const uint32_t n_all = (uint32_t)__builtin_amdgcn_readfirstlane((int)cold[offsetof(RolloutArgs, n) / 4]);
  const uint32_t max_steps = (uint32_t)__builtin_amdgcn_readfirstlane((int)cold[offsetof(RolloutArgs, max_steps) / 4]);
  if (lane < n_all) {
    const uint32_t *dsrc = SA_PTR(durations, const uint32_t *) + 2 * (size_t)lane;
    const uint32_t *psrc = SA_PTR(prng, const uint32_t *) + 2 * (size_t)lane;
    g.s0 = psrc[0];
    g.s1 = psrc[1];
    e.load_battle_global((const uint8_t *)(stage + tid * STAGE_STRIDE), dsrc[0], dsrc[1]);   // (the address space comes back inside: gin_t)
    if (cold[offsetof(RolloutArgs, prep) / 4]) { // mcts.h:254-259
      const uint32_t hi = g.next32(), lo = g.next32();
      e.rng = ((uint64_t)hi << 32) | lo;
      e.randomize_hidden();
    }
    result = SA_PTR(results_in, const uint8_t *)[lane];
    while ((result & 15) == 0 && steps < max_steps) {
      const uint32_t hi = g.next32(), lo = g.next32(); // uniform_64 = hi << 32 | lo
      result = e.random_step(result, hi, lo);
      ++steps;
    }
    e.normalize();
    SA_PTR(results_out, uint8_t *)[lane] = (uint8_t)result;
    SA_PTR(steps_out, uint32_t *)[lane] = steps;
    const uint32_t t = result & 15;
    SA_PTR(values_out, float *)[lane] = t == R_WIN ? 1.0f : t == R_LOSE ? 0.0f : 0.5f;
    uint32_t *pdst = SA_PTR(prng, uint32_t *) + 2 * (size_t)lane;
    pdst[0] = g.s0;
    pdst[1] = g.s1;
    uint32_t *dout = SA_PTR(durations_out, uint32_t *);
    if (dout) {
      uint32_t *ddst = dout + 2 * (size_t)lane;
      ddst[0] = e.S.dur;
      ddst[1] = e.F.dur;
    }
    if (SA_PTR(battles_out, uint8_t *)) e.store_battle_global((uint8_t *)(stage + tid * STAGE_STRIDE));
  }
  uint8_t *bout = SA_PTR(battles_out, uint8_t *);
  if (!bout) return;
  __syncthreads();
  u32x4 *dst = (u32x4 *)(bout + (size_t)base * 384);
  // (opaque: the 24 bounds tests here are the ones of the staging loop at the top, and the compiler kept their 24 lane masks -- 48
  // SGPRs -- alive across the whole turn loop to reuse them: 44 of the kernel's 46 spilled SGPRs)
  uint32_t cnt_out = cnt;
  asm volatile("" : "+s"(cnt_out));
#pragma unroll
  for (int k = 0; k < 24; ++k) {
    const uint32_t i = k * 64 + tid, b = i / 24, w = i - b * 24;
    if (i < cnt_out * 24) dst[i] = *(const lds_u128 *)(stage + b * STAGE_STRIDE + 4 * w);
  }
#undef SA_PTR
}

// ---- K1 driven by a caller-supplied DRAW STREAM instead of per-lane fast_prng: lane i consumes
// draws[offsets[i]], draws[offsets[i] + 1], ... -- one u64 per device.uniform_64() call of the reference loop (with
// prep the first goes to battle.rng, mcts.h:255-257, then one per turn-step, mcts.h:452).  This is how a SHARED
// sequential generator (benchmark.cc:24: one std::mt19937 for every playout) is replayed on the device: the host
// generates the generator's output once, and playout i starts where playout i-1 stopped (oakgpu_rollout_shared_device).
// stride 0 for battles / durations / results_in = every lane starts from the same root state.  A lane that would
// read past the end of the stream stops there and reports used = 0xFFFFFFFF.
struct DrawArgs {
  const uint8_t *battles, *durations, *results_in;
  uint32_t battle_stride, dur_stride, res_stride; // bytes (384 / 8 / 1) or 0
  const uint64_t *draws;
  uint32_t n_draws;
  const uint32_t *offsets; // nullable: lane i starts at draw i
  uint32_t n, max_steps;
  int prep;
  uint8_t *results_out;   // every output below is nullable
  uint32_t *steps_out;
  float *values_out;
  uint8_t *battles_out, *durations_out;
  uint32_t *used_out;     // draws consumed by lane i
};
template <int BLK>
__global__ __launch_bounds__(BLK, 2) void k_rollout_draws(DrawArgs a) {
  extern __shared__ __align__(16) uint8_t smem[];
  lds_u32 *party = (lds_u32 *)smem;
  using ER = EngineR<BLK, false>;
  Tables T = stage_default_tables((lds_u8 *)smem + ER::PARTY_WORDS * BLK * 4);
  __syncthreads();
  const uint32_t tid = threadIdx.x, lane = blockIdx.x * BLK + tid;
  if (lane >= a.n) return;
  const uint32_t *dsrc = (const uint32_t *)(a.durations + (size_t)lane * a.dur_stride);
  ER e;
  e.m = party + tid;
  e.T = T;
  e.load_battle_global(a.battles + (size_t)lane * a.battle_stride, dsrc[0], dsrc[1]);
  uint32_t pos = a.offsets ? a.offsets[lane] : lane;
  const uint32_t first = pos;
  bool over = false;
  if (a.prep) { // mcts.h:254-259
    if (pos < a.n_draws) { e.rng = a.draws[pos++]; e.randomize_hidden(); }
    else over = true;
  }
  uint32_t result = a.results_in[(size_t)lane * a.res_stride];
  uint32_t steps = 0;
  while (!over && (result & 15) == 0 && steps < a.max_steps) {
    if (pos >= a.n_draws) { over = true; break; }
    const uint64_t dr = a.draws[pos++];
    result = e.random_step(result, (uint32_t)(dr >> 32), (uint32_t)dr);
    ++steps;
  }
  e.normalize();
  if (a.used_out) a.used_out[lane] = over ? 0xFFFFFFFFu : pos - first;
  if (a.results_out) a.results_out[lane] = (uint8_t)result;
  if (a.steps_out) a.steps_out[lane] = steps;
  const uint32_t t = result & 15;
  if (a.values_out) a.values_out[lane] = t == R_WIN ? 1.0f : t == R_LOSE ? 0.0f : 0.5f;
  if (a.durations_out) {
    uint32_t *ddst = (uint32_t *)a.durations_out + 2 * (size_t)lane;
    ddst[0] = e.S.dur;
    ddst[1] = e.F.dur;
  }
  if (a.battles_out) e.store_battle_global(a.battles_out + (size_t)lane * 384);
}

// ---- K1 with lane refill: a persistent grid of `gridDim.x * BLK` lanes pulls playouts from an atomic
// queue.  A lane that finishes its playout immediately starts the next unassigned one (wave ballot ->
// one atomicAdd per wave -> prefix rank), so waves stay full instead of idling on their longest lane.
// Results are indexed by playout, each playout owns its RNG streams: output is identical to k_rollout*.
//
// Regrouping rounds.  Playout lengths have a long tail (mean ~100 turn-steps, 0.1% reach the 1000-step cap),
// so once the queue is dry every wave decays towards a single live lane that still pays a whole wave's issue
// slots.  With `suspend_below` > 0 a wave whose queue is dry and that is down to fewer live lanes than that
// SUSPENDS them: the full state goes to a per-playout scratch slot (the 384-byte battle image, durations,
// PRNG, result, step count), the playout indices are appended to `list_out`, and the wave exits.  The next
// launch (`list_in` = that list) packs the survivors 64 to a wave again.  The last round runs with
// suspend_below = 0.  Serialisation is the engine's own bit-exact load/store, so results do not change.
struct RoundArgs {
  const uint32_t *list_in;  // nullptr: round 0, playout k of the group
  const uint32_t *n_in;     // device count of list_in
  uint32_t *list_out;       // suspended playouts of this round
  uint32_t *count_out;
  uint8_t *sb;              // scratch: total x 384 battle images
  uint8_t *sd;              // scratch: total x 8 durations
  uint8_t *sres;            // scratch: total results
  uint32_t suspend_below;
  uint32_t *queue;          // this round's queue heads: QUEUE_HEADS counters, QUEUE_HEAD_STRIDE words apart
  uint32_t lanes;           // lanes of a wave that take playouts (0 / 64: all); a tail round runs a few playouts per wave
  const uint32_t *order;    // round 0, nullable: queue position -> playout (k_queue_order: the likely-long playouts first)
  // long-playout migration (below): control words {tail, head, bulk waves exited, error}, one list entry per donation
  uint32_t *adopt_ctl;
  uint32_t *adopt_list;
  uint32_t n_adopters;      // 0 = off; waves [0, n_adopters) adopt
  uint32_t long_steps;      // bits 0-15: a bulk wave donates a playout that is still running after this many turn-steps;
                            // bits 16-24: ... or whose actives' {slot, hp} have not changed for this many turn-steps (256: never)
  uint32_t no_skip;         // 1: a proven frozen standstill is played turn by turn like everything else (oakgpu_set_standstill_skip: A/B)
};

// One launch drains a GROUP of independent batches (oakgpu_rollout_group_dev): the queue hands out GLOBAL playout
// indices 0 .. total-1, batch b owns [start_b, start_b + n_b).  A 20-batch group has ONE tail instead of twenty.
struct BatchDesc {
  const uint8_t *battles, *durations, *results_in;
  uint8_t *prng;
  uint8_t *results_out;
  uint32_t *steps_out;
  float *values_out;
  uint8_t *battles_out, *durations_out;
  uint32_t start, n;
};
constexpr int MAX_GROUP = 64;
struct GroupArgs {
  const BatchDesc *table; // device memory, `count` entries
  uint32_t count, total, max_steps;
  int prep;
};

// The kernel's pointer arguments are needed only when a lane is (re)filled or retired, but as kernel arguments
// they would sit in ~40 SGPRs for the whole turn loop, whose nested divergent control flow needs those SGPRs
// for exec masks (the overflow is spilled to VGPR lanes: v_writelane / v_readlane in the hot path).  So the cold
// arguments are parked in LDS once (with the batches' start offsets) and the cold paths read them back.
struct ColdArgs {
  GroupArgs g;
  RoundArgs q;
  uint32_t starts[MAX_GROUP];
};
constexpr int COLD_LDS_BYTES = (sizeof(ColdArgs) + 15) & ~15;
template <class P>
__device__ __forceinline__ P cold_ptr(const lds_u32 *cold, size_t byte_off) {
  return (P)((uint64_t)cold[byte_off / 4] | ((uint64_t)cold[byte_off / 4 + 1] << 32));
}
#define COLD_Q(field, type) cold_ptr<type>(cold, offsetof(ColdArgs, q) + offsetof(RoundArgs, field))
// batch of global playout idx: largest b with starts[b] <= idx (binary search over the LDS copy)
__device__ __forceinline__ const BatchDesc *find_batch(const lds_u32 *cold, uint32_t idx) {
  const uint32_t count = cold[(offsetof(ColdArgs, g) + offsetof(GroupArgs, count)) / 4];
  const lds_u32 *starts = cold + offsetof(ColdArgs, starts) / 4;
  uint32_t lo = 0, hi = count; // invariant: starts[lo] <= idx < starts[hi] (starts[count] = +inf)
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (starts[mid] <= idx) lo = mid; else hi = mid;
  }
  return cold_ptr<const BatchDesc *>(cold, offsetof(ColdArgs, g) + offsetof(GroupArgs, table)) + lo;
}

#ifdef OAKGPU_TIMELINE
// profile build only (tools/timeline.py): per wave of the queue kernel {start, first dry refill, exit} on the
// 100 MHz wall clock + turn-steps executed
static __device__ unsigned long long g_timeline[5 * 16384];
#define OAK_TL(slot, v) do { const unsigned long long tl_v = (v); if (wl == 0 && blockIdx.x < 16384) g_timeline[5 * blockIdx.x + (slot)] = tl_v; } while (0)
#else
#define OAK_TL(slot, v)
#endif

// Queue order of a group launch: LONGEST-EXPECTED FIRST.  The launch ends with its longest playout, and the longest are the
// 0.1% that run into the step cap: stalemates -- a Rage-locked or PP-less (Struggle) attacker whose Normal-type hits cannot
// touch a Ghost, a frozen Ghost nobody can hurt.  Measured on the CPU oracle (3 x 65,536 random OU playouts, 183 capped):
// 99.5% of the capped playouts have a Ghost-type Pokemon on one of the two teams, against 22% of all playouts; the rest had a
// Ditto (with Ghosts and Dittos: 183 of 183, 29% of all playouts).  A playout
// that starts when the queue runs dry still has its 1,000 dependent turn-steps in front of it; one that started in the first
// fifth of the launch has most of them behind it.  So the playouts with a Ghost on either team go to the front of the queue
// (order[] filled from the front), the others behind them (filled from the back).  Pure scheduling: results are indexed by
// playout and never depend on it.
__global__ __launch_bounds__(1024) void k_queue_order(GroupArgs g, uint32_t *order, uint32_t *counters) {
  __shared__ uint32_t starts[MAX_GROUP + 1];
  __shared__ uint32_t wave_s[16], wave_o[16], base_s, base_o;
  for (uint32_t i = threadIdx.x; i < g.count; i += 1024) starts[i] = g.table[i].start;
  if (threadIdx.x == 0) starts[g.count] = 0xFFFFFFFFu;
  __syncthreads();
  const uint32_t idx = blockIdx.x * 1024 + threadIdx.x, wl = threadIdx.x & 63, wib = threadIdx.x >> 6;
  bool suspect = false;
  if (idx < g.total) {
    uint32_t lo = 0, hi = g.count;
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (starts[mid] <= idx) lo = mid; else hi = mid; }
    const BatchDesc &bd = g.table[lo];
    const uint32_t *b = (const uint32_t *)(bd.battles + (size_t)(idx - bd.start) * 384);
    uint32_t w[12]; // the dword {status, species, types, level} of each of the 12 Pokemon: all twelve loads in flight together
#pragma unroll
    for (int k = 0; k < 12; ++k) w[k] = b[((k / 6) * SIDE_SZ + (k % 6) * PK_SZ + P_STATUS) / 4];
#pragma unroll
    for (int k = 0; k < 12; ++k) {
      const uint32_t sp = (w[k] >> 8) & 0xFF, ty = (w[k] >> 16) & 0xFF;
      suspect |= sp != 0 && ((ty & 15) == T_Ghost || (ty >> 4) == T_Ghost || sp == 132 /* Ditto: Transform into a stalemate */);
    }
  }
  // one pair of atomics per 1,024 playouts (a pair per wave: 41,000 atomics on two addresses took longer than the loads)
  const uint64_t ms = __ballot(suspect), mo = __ballot(idx < g.total && !suspect);
  if (wl == 0) { wave_s[wib] = (uint32_t)__popcll(ms); wave_o[wib] = (uint32_t)__popcll(mo); }
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t ts = 0, to = 0;
    for (int q = 0; q < 16; ++q) { const uint32_t a = wave_s[q], c = wave_o[q]; wave_s[q] = ts; wave_o[q] = to; ts += a; to += c; }
    base_s = ts ? atomicAdd(counters + 0, ts) : 0;
    base_o = to ? atomicAdd(counters + 1, to) : 0;
  }
  __syncthreads();
  if (idx < g.total) {
    const uint64_t below = (1ull << wl) - 1;
    if (suspect) order[base_s + wave_s[wib] + (uint32_t)__popcll(ms & below)] = idx;
    else order[g.total - 1 - (base_o + wave_o[wib] + (uint32_t)__popcll(mo & below))] = idx;
  }
}

#ifndef OAK_LONG_STEPS
#define OAK_LONG_STEPS 200
#endif
constexpr uint32_t LONG_STEPS = OAK_LONG_STEPS;
#ifndef OAK_PRIO_STILL
#define OAK_PRIO_STILL 24
#endif
constexpr uint32_t PRIO_STILL = OAK_PRIO_STILL;
#ifndef OAK_REFILL_EVERY
#define OAK_REFILL_EVERY 8
#endif
#ifndef OAK_REFILL_LANES
#define OAK_REFILL_LANES 16
#endif
constexpr uint32_t REFILL_EVERY = OAK_REFILL_EVERY, REFILL_LANES = OAK_REFILL_LANES; // free lanes refill every n-th iteration (a power of two), or at once when this many are free
constexpr uint32_t QUEUE_HEADS = 8, QUEUE_HEAD_STRIDE = 64; // eight queue heads per round, 64 words (one 256-byte line) apart
template <int BLK, int WPS>
__global__ __launch_bounds__(BLK, WPS) void k_rollout_queue(GroupArgs g_in, RoundArgs q_in) {
  extern __shared__ __align__(16) uint8_t smem[];
  lds_u32 *party = (lds_u32 *)smem;
  using ER = EngineR<BLK, false>;
  Tables T = stage_default_tables((lds_u8 *)smem + ER::PARTY_WORDS * BLK * 4);
  lds_u32 *cold = (lds_u32 *)((lds_u8 *)smem + ER::PARTY_WORDS * BLK * 4 + TABLE_LDS_PAD);
  if (threadIdx.x == 0) {
    struct { GroupArgs g; RoundArgs q; } c{g_in, q_in};
    const uint32_t *src = (const uint32_t *)&c;
#pragma unroll
    for (uint32_t i = 0; i < offsetof(ColdArgs, starts) / 4; ++i) cold[i] = src[i];
  }
  for (uint32_t i = threadIdx.x; i < g_in.count; i += BLK) cold[offsetof(ColdArgs, starts) / 4 + i] = g_in.table[i].start;
  __syncthreads();
  const uint32_t tid = threadIdx.x, wl = tid & 63;
  constexpr uint32_t NONE = 0xFFFFFFFFu, DONE = 0xFFFFFFFEu;
  // the few scalars the turn loop itself needs stay in SGPRs -- each read back from the LDS copy as a 32-bit value of its own:
  // taken straight from the kernel arguments they stay sub-registers of the 16-dword argument load, and the register
  // allocator then spills and restores that WHOLE tuple around the turn loop (round 3: 39 SGPR spills, 16 v_readlane per
  // wave iteration, after the migration arguments joined RoundArgs)
#define COLD_U32(off) ((uint32_t)__builtin_amdgcn_readfirstlane((int)cold[(off) / 4]))
#define COLD_QU(field) COLD_U32(offsetof(ColdArgs, q) + offsetof(RoundArgs, field))
#define COLD_GU(field) COLD_U32(offsetof(ColdArgs, g) + offsetof(GroupArgs, field))
  // (wave-uniform booleans live as BITS of one scalar word: as `bool`s each is a 64-bit lane mask, two SGPRs, and the turn loop
  // has none to spare)
  constexpr uint32_t U_RESUME = 1, U_PREP = 2, U_ADOPTER = 4, U_DRY = 8, U_ADOPTING = 16, U_ANY_PLAYING = 32, U_NO_SKIP = 64;
  uint32_t ust = (COLD_QU(list_in) | COLD_U32(offsetof(ColdArgs, q) + offsetof(RoundArgs, list_in) + 4)) != 0 ? U_RESUME : 0u; // a later round: playouts come from the previous round's suspended list
#define IS_RESUME ((ust & U_RESUME) != 0)
#define IS_PREP ((ust & U_PREP) != 0)
  const uint32_t total = IS_RESUME ? (uint32_t)__builtin_amdgcn_readfirstlane((int)*COLD_Q(n_in, const uint32_t *)) : COLD_GU(total);
  const uint32_t max_steps = COLD_GU(max_steps), suspend_below = COLD_QU(suspend_below);
  if (!IS_RESUME && COLD_GU(prep) != 0) ust |= U_PREP;
  ust |= (blockIdx.x & 7u) << 8; // the queue head this wave starts at
  if (COLD_QU(no_skip) != 0) ust |= U_NO_SKIP;
  ER e;
  e.m = party + tid;
  e.T = T;
  FastPrng g;
  g.s0 = g.s1 = 0;
  uint32_t idx = NONE, result = 0, steps = 0;
  { const uint32_t lanes = COLD_QU(lanes); if (lanes && wl >= lanes) idx = DONE; } // (tail round: this lane stays empty)
#define IS_DRY ((ust & U_DRY) != 0) // wave-uniform: the queue has handed out its last playout
  // ---- long-playout migration.  The launch ends with its longest playouts -- 1,000-step chains -- and while the device is
  // full such a playout advances at the pace of a full, divergent wave (31 us per turn-step).  So a BULK wave hands a playout
  // that is still running after `long_steps` turn-steps (99.5% end before 250) to the ADOPTER waves (the first n_adopters
  // waves, about one per CU): bit-exact state image to the scratch slot (the regrouping rounds' suspend image), agent-scope
  // release, a ticket in the adoption list.  An adopter stops taking playouts from the main queue once the first donation
  // exists, lets its own finish, and from then on holds only long playouts -- a dozen per wave, at the top priority of its
  // SIMD -- so the chains run at a sparse wave's pace (4-8 us per turn-step) from their 200th step on instead of from the
  // moment the device drains.  Every wait is bounded (a ticket reserved but not yet written; an adopter with nothing to
  // adopt while bulk waves still run): on overflow the error word is set and the wave leaves.  Results are indexed by
  // playout: they do not depend on who finishes a playout.
  // Round 4: a STANDSTILL counter donates earlier.  Measured on the oracle (2 x 65,536 playouts): the playouts that run into the cap
  // stop changing any hp at a median of turn-step 85 (p90 160) -- from then on they are the stalemates of k_queue_order's comment --
  // while an ordinary playout almost never goes 40 turn-steps without its actives' {slot, hp} changing.  `stale` = a 24-bit
  // signature of both actives (hp xor, slot sum: symmetric, the frame may be swapped) | the turn-steps it has stood still << 24.
  const uint32_t n_adopt = COLD_QU(n_adopters), long_steps = COLD_QU(long_steps);
  uint32_t stale = 0;
  if (n_adopt != 0 && blockIdx.x < n_adopt) ust |= U_ADOPTER;
#define IS_ADOPTER ((ust & U_ADOPTER) != 0)
#define IS_ADOPTING ((ust & U_ADOPTING) != 0) // adopter: a donation has been seen, no more playouts from the main queue
#define ANY_PLAYING ((ust & U_ANY_PLAYING) != 0)
  uint32_t idle_polls = 0, poll_tick = 15;
  constexpr uint32_t SPIN_CAP = 1u << 22;
  constexpr int ERR_WORD = 23; // adopt_ctl + 23 = word 63 of the context's control block: STICKY (no launch clears it; oakgpu_synchronize reads and clears)
  OAK_PROF_ZERO();
  OAK_TL(0, wall_clock64());
#ifdef OAKGPU_TIMELINE
  bool tl_dry = false;
  unsigned long long tl_steps = 0;
#endif
  for (;;) {
    bool need = idx == NONE, load = false, from_scratch = IS_RESUME;
    uint64_t mask = __ballot(need);
    // (an adopter looks at the adoption list every 16th turn-step, or at once when it has nothing to play: 256 sparse waves
    // polling three words on every iteration saturate that L2 line's atomics and slow the whole launch down)
    if (mask && IS_ADOPTER && ((++poll_tick & 15u) == 0 || !ANY_PLAYING)) { // wave-uniform: an adopter's free lanes take donated playouts first
      uint32_t *ctl = COLD_Q(adopt_ctl, uint32_t *);
      uint32_t h = 0, k = 0, closed = 0;
      if (wl == 0) {
        for (int tries = 0; tries < 8; ++tries) {
          // HEAD first, then TAIL: the tail only grows and the head never passes it, so tail - head cannot underflow (read the
          // other way round, a head that moved in between claimed tickets beyond the list: a memory fault at full size)
          const uint32_t hh = __hip_atomic_load(ctl + 1, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
          const uint32_t t = __hip_atomic_load(ctl + 0, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
          if (t) closed |= 2; // (bit 1: a donation exists)
          const uint32_t kk = t > hh ? min(t - hh, (uint32_t)__popcll(mask)) : 0u;
          if (!kk) {
            // every bulk wave has left (a wave's last donation is complete before it counts itself out) and nothing is waiting
            if (__hip_atomic_load(ctl + 2, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - n_adopt &&
                __hip_atomic_load(ctl + 0, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == hh &&
                __hip_atomic_load(ctl + 1, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == hh) closed |= 1;
            break;
          }
          if (atomicCAS(ctl + 1, hh, hh + kk) == hh) { h = hh; k = kk; break; }
        }
      }
      h = (uint32_t)__builtin_amdgcn_readfirstlane((int)h); k = (uint32_t)__builtin_amdgcn_readfirstlane((int)k); closed = (uint32_t)__builtin_amdgcn_readfirstlane((int)closed);
      if (closed & 2) ust |= U_ADOPTING;
      const uint32_t rank = (uint32_t)__popcll(mask & ((1ull << wl) - 1));
      const bool take = need && rank < k;
      uint32_t got = 0;
      if (take && h + rank < COLD_GU(total)) { // (tickets live in [0, total): one per donation at most)
        const uint32_t *slot = COLD_Q(adopt_list, const uint32_t *) + h + rank;
        uint32_t spins = 0;
        while ((got = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0 && ++spins < SPIN_CAP) __builtin_amdgcn_s_sleep(2);
        if (!got) atomicOr(ctl + ERR_WORD, 1u); // (a ticket that never arrived: reported, the playout is lost -- the tests would see it)
      }
      if (k) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); // the donors' images, step counts and PRNG states
      if (take && got) { idx = got - 1; load = true; from_scratch = true; }
      need = idx == NONE;
      if ((closed & 1) && need) idx = DONE;              // adoption is over
      mask = (IS_ADOPTING || (closed & 1)) ? 0 : __ballot(need); // once donations exist an adopter takes no more bulk work
      need = need && mask != 0;
    }
    if (IS_ADOPTING) { mask = 0; need = false; } // (an adopter between two looks at the list: no bulk work either)
    // (an adopter's free lanes stay NONE after the queue has run dry -- they wait for donations -- and must not keep asking the
    // dry queue: ~150 waves adding to one L2 word on every iteration, and in a long launch the 32-bit head could wrap and hand
    // playouts out twice; round-3 advice)
    // (refills in batches, OAK_REFILL_EVERY > 1: a refill is a wave-wide stall -- a returning atomic, then the dependent loads of the
    // queue order, the batch descriptor and the 384-byte battle -- for the sake of the one or two lanes that finished in this
    // iteration; free lanes cost no issue slots, so they wait for company: every 8th iteration, or at once when 16 lanes are free.
    // Measured, tools/refill_variants.sh: driver command 8.53-8.59 -> 8.91-8.96 G, 160 steps 9.24-9.30 -> 9.76-9.79 G; every 4th / 16th
    // iteration and thresholds of 8 / 64 lanes are within 2 % of it)
    ust = (ust & ~0xFF0000u) | ((ust + 0x10000u) & 0xFF0000u);
    if (mask && !IS_DRY && (REFILL_EVERY <= 1 || ((ust >> 16) & (REFILL_EVERY - 1)) == 0 || (uint32_t)__popcll(mask) >= REFILL_LANES || !ANY_PLAYING)) { // wave-uniform
      OAK_SCOPE(PS_REFILL);
      // The queue has EIGHT heads, each on a 256-byte line of its own (round 5): one device-scope counter saturates at ~88 returning
      // atomics per microsecond on this chip (MI355X_MICROARCH.md, "dequeue"), and 4,096 waves refilling 0.5 times per ~28 us
      // iteration ask for 70-90 -- the single head of rounds 1-4 ran at its limit.  Head s hands out the queue positions s, s + 8,
      // s + 16, ... (so every head's sequence starts with the likely-long playouts of k_queue_order); a wave starts at head
      // blockIdx % 8 (workgroups are dealt round-robin over the XCDs) and moves to the next head when its own is exhausted; the
      // queue is dry for a wave when it has seen all eight exhausted.  `ust` bits 8-10: the current head, bits 12-15: heads seen dry.
      uint64_t rem = mask;
      bool got = false;
      uint32_t my = 0;
      for (;;) {
        const uint32_t shard = (ust >> 8) & 7u, need_n = (uint32_t)__popcll(rem);
        const uint32_t lim = total > shard ? (total - shard + 7u) >> 3 : 0u; // positions of this head
        uint32_t base = 0;
        if (wl == 0) base = atomicAdd(COLD_Q(queue, uint32_t *) + shard * QUEUE_HEAD_STRIDE, need_n);
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base); // (lane 0 is active: whole waves run this loop)
        const uint32_t avail = base < lim ? (lim - base < need_n ? lim - base : need_n) : 0u;
        const uint32_t rank = (uint32_t)__popcll(rem & ((1ull << wl) - 1));
        if (((rem >> wl) & 1) && rank < avail) { my = shard + ((base + rank) << 3); got = true; }
        if (avail == need_n) break;
        rem = __ballot(need && !got);
        ust = (ust & ~(7u << 8)) | (((shard + 1u) & 7u) << 8);
        ust += 1u << 12;
        if (((ust >> 12) & 15u) >= 8u) { ust |= U_DRY; break; }
      }
#ifdef OAKGPU_TIMELINE
      if (IS_DRY && !tl_dry) { tl_dry = true; OAK_TL(1, wall_clock64()); }
#endif
      if (need) {
        if (got) {
          const uint32_t *order = COLD_Q(order, const uint32_t *);
          idx = IS_RESUME ? COLD_Q(list_in, const uint32_t *)[my] : order ? order[my] : my;
          load = true;
        } else idx = IS_ADOPTER ? NONE : DONE; // (the queue is dry; an adopter's free lanes wait for donations until adoption is over)
      }
    } else if (mask && IS_DRY && !IS_ADOPTER) { // the queue is dry: a bulk wave's free lanes are done (an adopter's wait for donations)
      if (need) idx = DONE;
    }
    if (__ballot(load)) { // wave-uniform: (re)fill the lanes that got a playout -- from its batch, or from its parked image
      OAK_SCOPE(PS_REFILL);
      if (load) {
        const BatchDesc *bd = find_batch(cold, idx);
        const uint32_t k = idx - bd->start; // playout k of its batch
        const uint32_t *dsrc = from_scratch ? COLD_Q(sd, const uint32_t *) + 2 * (size_t)idx : (const uint32_t *)bd->durations + 2 * (size_t)k;
        const uint32_t *psrc = (const uint32_t *)bd->prng + 2 * (size_t)k;
        g.s0 = psrc[0];
        g.s1 = psrc[1];
        e.load_battle_global(from_scratch ? COLD_Q(sb, const uint8_t *) + (size_t)idx * 384 : bd->battles + (size_t)k * 384, dsrc[0], dsrc[1]);
        if (IS_PREP && !from_scratch) { // mcts.h:254-259
          const uint32_t hi = g.next32(), lo = g.next32();
          e.rng = ((uint64_t)hi << 32) | lo;
          e.randomize_hidden();
        }
        result = from_scratch ? COLD_Q(sres, const uint8_t *)[idx] : bd->results_in[k];
        steps = from_scratch ? bd->steps_out[k] : 0;
        stale = 0;
      }
    }
    if (__ballot(idx != DONE) == 0) break;
    bool playing = idx != DONE && idx != NONE && (result & 15) == 0 && steps < max_steps; // (NONE: an adopter's lane that waits)
#ifdef OAKGPU_TIMELINE
    tl_steps += (unsigned long long)__popcll(__ballot(playing));
#endif
    // A standstill that can be PROVEN: both actives FROZEN (gen 1 never thaws by itself), neither side able to leave (its last
    // Pokemon, or locked into a move), nothing that acts on a Pokemon that cannot move (Leech Seed, binding) and different speeds.  Such a turn-step draws
    // nothing from battle.rng (no speed tie), executes no move (before_move returns at the freeze check) and leaves every byte as
    // it was except the turn counter and the fields the NEXT turn-step overwrites unconditionally -- last selected move, last
    // move index, last used move = 0, the flinch bit -- see EngineR::frozen_standstill.  So all but the last of the remaining
    // turn-steps are taken at once: turn and step count advance, the choice stream advances by its two draws per turn-step, and
    // the last turn-step runs for real (it writes those fields and, at turn 1,000, the tie).  These are the stalemates the queue
    // order cannot see in the teams (round 4: 1-5 per 1.31 M playouts, started anywhere in the queue, each ~950 dependent
    // turn-steps at a lone lane's 4.7 us: the launches they ended took 16.4-16.6 ms instead of 14.7-14.9).  Exact, not a
    // heuristic: tests/test_gpu_parity.py holds it to the oracle, which plays every turn.
    if (playing && stale >= (8u << 24) && !(ust & U_NO_SKIP) && e.frozen_standstill(result)) {
      const uint32_t by_steps = max_steps - steps, by_turn = 1000u - e.turn;
      const uint32_t skip = (by_steps < by_turn ? by_steps : by_turn) - 1u; // (both >= 1 while playing)
      for (uint32_t k = 0; k < skip; ++k) { (void)g.next32(); (void)g.next32(); }
      e.turn += skip;
      steps += skip;
    }
    if (playing) {
      const uint32_t hi = g.next32(), lo = g.next32(); // uniform_64 = hi << 32 | lo
      result = e.random_step(result, hi, lo);
      ++steps;
      playing = (result & 15) == 0 && steps < max_steps;
      const uint32_t sg = ((e.S.p4 ^ e.F.p4) >> 16) | (((e.S.o0 + e.F.o0) & 0xFFu) << 16);
      stale = sg != (stale & 0xFFFFFFu) ? sg : stale + (stale < 0xFF000000u ? 0x01000000u : 0u);
    }
    // a wave that holds a playout far beyond the usual length (99.5% end before 250 turn-steps) is on the launch's critical
    // path -- a 1000-step chain: it goes first on its SIMD (an adopter as soon as it adopts)
    // (round 4: ... or one whose actives have stood still for PRIO_STILL turn-steps -- the stalemates that run into the cap stop changing
    // at a median of turn-step 85, so their wave goes first ~90 turn-steps earlier; a false alarm costs nothing but a while of priority)
    if (IS_ADOPTING || __ballot(playing && (steps > LONG_STEPS || stale >= (PRIO_STILL << 24)))) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(0);
    if (__ballot(playing) != 0) ust |= U_ANY_PLAYING; else ust &= ~U_ANY_PLAYING;
    if (IS_ADOPTER && !ANY_PLAYING) { // nothing to play: wait for donations (bounded) without burning issue slots
      __builtin_amdgcn_s_sleep(100);
      if (++idle_polls > SPIN_CAP) { if (wl == 0) atomicOr(COLD_Q(adopt_ctl, uint32_t *) + ERR_WORD, 2u); break; }
    }
    // wave-uniform: the queue is dry and too few lanes are still playing -> hand them to the next round
    const uint64_t still = __ballot(playing);
    const bool suspend = IS_DRY && still != 0 && (uint32_t)__popcll(still) < suspend_below;
    const bool lng = n_adopt != 0 && !IS_ADOPTER && playing && (steps >= (long_steps & 0xFFFFu) || (stale >> 24) >= (long_steps >> 16)); // a bulk wave's long (or standing-still) playout: to the adopters
    if (idx != DONE && idx != NONE && (!playing || suspend || lng)) { // retire the lane: publish a finished playout / park a suspended or donated one
      OAK_SCOPE(PS_PUBLISH);
      const bool fin = !playing;
      e.normalize();
      const BatchDesc *bd = find_batch(cold, idx);
      const uint32_t k = idx - bd->start;
      if (fin) bd->results_out[k] = (uint8_t)result; else COLD_Q(sres, uint8_t *)[idx] = (uint8_t)result;
      bd->steps_out[k] = steps;
      const uint32_t t = result & 15;
      if (fin) bd->values_out[k] = t == R_WIN ? 1.0f : t == R_LOSE ? 0.0f : 0.5f;
      uint32_t *pdst = (uint32_t *)bd->prng + 2 * (size_t)k;
      pdst[0] = g.s0;
      pdst[1] = g.s1;
      uint32_t *ddst = fin ? (bd->durations_out ? (uint32_t *)bd->durations_out + 2 * (size_t)k : nullptr) : COLD_Q(sd, uint32_t *) + 2 * (size_t)idx;
      if (ddst) {
        ddst[0] = e.S.dur;
        ddst[1] = e.F.dur;
      }
      uint8_t *bdst = fin ? (bd->battles_out ? bd->battles_out + (size_t)k * 384 : nullptr) : COLD_Q(sb, uint8_t *) + (size_t)idx * 384;
      if (bdst) e.store_battle_global(bdst);
      idx = fin ? NONE : idx;
    }
    const uint64_t dm = __ballot(lng);
    if (dm) { // wave-uniform, rare: release the images, take tickets, write them
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      uint32_t *ctl = COLD_Q(adopt_ctl, uint32_t *);
      const uint32_t leader = (uint32_t)__ffsll((unsigned long long)dm) - 1;
      uint32_t base = 0;
      if (wl == leader) base = atomicAdd(ctl + 0, (uint32_t)__popcll(dm));
      base = (uint32_t)__builtin_amdgcn_readlane((int)base, (int)leader);
      if (lng) {
        __hip_atomic_store(COLD_Q(adopt_list, uint32_t *) + base + (uint32_t)__popcll(dm & ((1ull << wl) - 1)), idx + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        idx = NONE; // the lane is free again
      }
    }
    if (suspend) {
      const uint32_t leader = (uint32_t)__ffsll((unsigned long long)still) - 1;
      uint32_t base = 0;
      if (wl == leader) base = atomicAdd(COLD_Q(count_out, uint32_t *), (uint32_t)__popcll(still));
      base = (uint32_t)__builtin_amdgcn_readlane((int)base, (int)leader);
      if (playing) COLD_Q(list_out, uint32_t *)[base + (uint32_t)__popcll(still & ((1ull << wl) - 1))] = idx;
      break;
    }
  }
  if (n_adopt != 0 && !IS_ADOPTER && wl == 0) atomicAdd(COLD_Q(adopt_ctl, uint32_t *) + 2, 1u); // a bulk wave has left: it donates no more
  OAK_TL(2, wall_clock64());
#ifdef OAKGPU_TIMELINE
  OAK_TL(4, tl_steps);
#endif
  OAK_PROF_FLUSH();
}
#undef COLD_Q
#undef COLD_U32
#undef COLD_QU
#undef COLD_GU
#undef IS_RESUME
#undef IS_PREP
#undef IS_DRY
#undef IS_ADOPTER
#undef IS_ADOPTING
#undef ANY_PLAYING

// ---- root-parallel search steps in SLICES (BASELINE configs[3]: 256 roots x 4,096 playouts per search step) -----------------
// A search step of root-parallel MCTS hands every root `reps` fresh playouts (run_root_iteration's prep, mcts.h:250-263, then the
// rollout loop of mcts.h:448-496) and consumes one aggregate per root.  Run to terminal inside the step, a step lasts as long
// as its LONGEST playout -- 98 % of the 4,096-playout batches hold one that runs into the 1,000-step cap: a chain of 1,000 dependent
// turn-steps (6 ms) however little work the step holds, which is what bound a rank's share at 8 GPUs (round 4: 6.5 ms, 1.8x
// projected).  The reference's workers never wait for each other (generate.cc:527-536), and neither does a step here: a launch
// advances every playout in flight by at most `slice` turn-steps (a power of two).  A playout that ends inside its slice is
// CREDITED TO THE STEP WHOSE LAUNCH FINISHED IT -- step k + (len - 1) / slice for a playout of len turn-steps started in step k,
// a function of the playout's own length only, never of the schedule; one that does not is CARRIED: its bit-exact state image
// (the 384-byte battle, durations, result, step count, its own choice stream, its root) goes to the carry list and the NEXT
// step's launch resumes it beside that step's fresh playouts.  Values never change, only the step they are credited to.
// The per-root aggregate is folded into the retire path: one 64-bit atomic per finished playout, count | (2 x value) << 32
// -- integers, so a root's aggregate does not depend on the order its playouts finish in (byte-identical to the oracle's).
//   Streams: lane (root r, replica i) owns one fast_prng stream that advances by exactly ONE uniform_64 per step; that draw IS
// the 8-byte state of the step's fresh playout's own stream (an all-zero draw -- the generator's fixed point -- becomes s1 = 1),
// from which the playout takes battle.rng (prep) and its choices.  So a carried playout and the lane's next fresh one never
// share draws, and nothing depends on which launch, wave or rank runs a playout.
struct RootStepArgs {
  const uint8_t *root_battles, *root_durations, *root_results; // n_roots x {384, 8, 1}
  uint8_t *lane_prng;              // n_fresh x 8
  const uint4 *cin_state;          // carried playouts: ROOT_SHARDS segments of `seg` records of CARRY_VEC uint4 (below)
  const uint32_t *cin_count;       // ROOT_SHARDS counters, CTL_STRIDE words apart (one 256-byte line each)
  uint4 *cout_state;
  uint32_t *cout_count;
  unsigned long long *acc;         // ROOT_SHARDS x acc_stride: finished playouts credited to this step per root, count | (2 x value) << 32, one copy per
                                   // shard (a rank's 32 roots are ONE 256-byte line: 131,072 credits per 1.4 ms on it ran at the ~90 atomics/us a line takes)
  unsigned long long *turn_steps;  // += the turn-steps this launch executed
  uint32_t *queue;                 // ROOT_SHARDS queue heads (CTL_STRIDE words apart), zeroed before the launch
  uint32_t *err;                   // sticky: bit 0 = a carry segment overflowed (playouts lost)
  uint32_t n_fresh, reps, seg, slice_mask, max_steps, fper; // fper: fresh lanes per shard (the last shard takes what is left)
  uint32_t acc_stride, pad;        // u64 entries between two shards' accumulators
};
// The queue and the carry list are cut into 8 SHARDS, each with a head and a counter on a 256-byte line of its own: one device-scope
// counter saturates at ~88 returning atomics per microsecond on this chip (MI355X_MICROARCH.md, "dequeue"), and a step in slices
// of 64 turn-steps needs ~110 dequeues + ~60 carry tickets per microsecond -- with ONE head and ONE ticket counter the kernel ran at
// exactly that rate whatever else it did (256 roots: 20 ms per step at slice 64, 30 ms at 32, against 11 ms of turn-steps).  Shard
// s = {segment s of the carry list, fresh lanes [s * fper, (s + 1) * fper)}; a wave starts at shard blockIdx % 8 (the dispatcher
// deals workgroups round-robin over the XCDs) and moves on to the next shard when its own is exhausted, so every playout is taken
// whoever is left; it carries into its home shard's segment of the outgoing list.
#ifndef OAK_ROOT_REFILL_EVERY
#define OAK_ROOT_REFILL_EVERY 8
#endif
#ifndef OAK_ROOT_REFILL_LANES
#define OAK_ROOT_REFILL_LANES 16
#endif
constexpr uint32_t ROOT_REFILL_EVERY = OAK_ROOT_REFILL_EVERY, ROOT_REFILL_LANES = OAK_ROOT_REFILL_LANES;
constexpr int ROOT_SHARDS = 8;
constexpr int CTL_STRIDE = 64; // words between two counters
// A carried playout travels as the engine's own MUTABLE state, raw: both sides' register sets (2 x 18 dwords, in whatever frame the
// turn left them), the battle scalars, the 24 party dwords of its LDS column, its choice stream, step count, root and result --
// 72 dwords = 18 uint4, stored and loaded as whole vectors with no load in the store path.  (First form: the 384-byte battle
// image through store_battle_global / load_battle_global, whose stores each wait for a load of the immutable fields behind the
// previous store -- stores count in vmcnt on gfx950 -- ~37 us of wave time per carried playout: 256 roots in slices of 64 ran at
// 5.2 G turn-steps/s against 9.4 G without slices.)  The immutable party data is the ROOT's (`gin`), shared by the root's playouts.
constexpr int CARRY_VEC = 18;
constexpr int ROOT_STEP_COLD_BYTES = ((sizeof(RootStepArgs) + 15) & ~15) + ROOT_SHARDS * 4;
constexpr int ROOT_STEP_LDS_BYTES = 24 * 64 * 4 + TABLE_LDS_PAD + ROOT_STEP_COLD_BYTES;
template <int WPS>
__global__ __launch_bounds__(64, WPS) void k_root_step(RootStepArgs a_in) {
  extern __shared__ __align__(16) uint8_t smem[];
  lds_u32 *party = (lds_u32 *)smem;
  using ER = EngineR<64, false>;
  Tables T = stage_default_tables((lds_u8 *)smem + ER::PARTY_WORDS * 64 * 4);
  // the cold arguments are parked in LDS, like k_rollout_queue's: as kernel arguments they would hold ~34 SGPRs across the turn loop
  lds_u32 *cold = (lds_u32 *)((lds_u8 *)smem + ER::PARTY_WORDS * 64 * 4 + TABLE_LDS_PAD);
  if (threadIdx.x == 0) {
    const uint32_t *src = (const uint32_t *)&a_in;
#pragma unroll
    for (uint32_t i = 0; i < sizeof(RootStepArgs) / 4; ++i) cold[i] = src[i];
  }
  lds_u32 *sh_carry = cold + ((sizeof(RootStepArgs) + 15) & ~15) / 4; // carried playouts per shard (clamped to the segment size)
  if (threadIdx.x < ROOT_SHARDS) {
    const uint32_t c = a_in.cin_count[threadIdx.x * CTL_STRIDE];
    sh_carry[threadIdx.x] = c < a_in.seg ? c : a_in.seg; // (an overflowing launch lost the playouts beyond the segment and said so in *err)
  }
  __syncthreads();
#define RS_PTR(field, type) cold_ptr<type>(cold, offsetof(RootStepArgs, field))
#define RS_U32(field) ((uint32_t)__builtin_amdgcn_readfirstlane((int)cold[offsetof(RootStepArgs, field) / 4]))
  const uint32_t wl = threadIdx.x;
  constexpr uint32_t NONE = 0xFFFFFFFFu, DONE = 0xFFFFFFFEu;
  const uint32_t seg = RS_U32(seg), max_steps = RS_U32(max_steps), slice_mask = RS_U32(slice_mask);
  const uint32_t home = blockIdx.x & (ROOT_SHARDS - 1);
  uint32_t shard = home, exhausted = 0, iter = 0xFFFFFFFFu;
  bool any_playing = false;
  ER e;
  e.m = party + wl;
  e.T = T;
  FastPrng g;
  g.s0 = g.s1 = 0;
  uint32_t root = NONE, result = 0, steps = 0, executed = 0;
  bool dry = false;
  for (;;) {
    bool load = false;
    uint32_t my = 0, my_carry = 0, my_shard = 0;
    const uint64_t mask = __ballot(root == NONE);
    ++iter;
    if (mask && !dry && (ROOT_REFILL_EVERY <= 1 || (iter & (ROOT_REFILL_EVERY - 1)) == 0 || (uint32_t)__popcll(mask) >= ROOT_REFILL_LANES || !any_playing)) { // wave-uniform: free lanes take the next playouts of the wave's current shard -- its carried ones first (the oldest)
      uint64_t rem = mask;
      while (rem) {
        const uint32_t need = (uint32_t)__popcll(rem);
        const uint32_t carry_s = (uint32_t)__builtin_amdgcn_readfirstlane((int)sh_carry[shard]);
        const uint32_t n_fresh = RS_U32(n_fresh), fper = RS_U32(fper), flo = shard * fper;
        const uint32_t fresh_s = flo < n_fresh ? (n_fresh - flo < fper ? n_fresh - flo : fper) : 0u;
        const uint32_t tot = carry_s + fresh_s;
        uint32_t base = 0;
        if (wl == 0) base = atomicAdd(RS_PTR(queue, uint32_t *) + shard * CTL_STRIDE, need);
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        const uint32_t avail = base < tot ? (tot - base < need ? tot - base : need) : 0u;
        const uint32_t rank = (uint32_t)__popcll(rem & ((1ull << wl) - 1));
        if (((rem >> wl) & 1) && rank < avail) { my = base + rank; my_carry = carry_s; my_shard = shard; load = true; }
        if (avail == need) break;
        rem = __ballot(root == NONE && !load);           // the shard is exhausted: the lanes it could not serve try the next one
        shard = (shard + 1) & (ROOT_SHARDS - 1);
        if (++exhausted >= (uint32_t)ROOT_SHARDS) { dry = true; break; }
      }
    }
    if (dry && root == NONE && !load) root = DONE;
    if (__ballot(load)) {
      if (load && my < my_carry) { // resume a carried playout from its record
        const uint4 *rec = RS_PTR(cin_state, const uint4 *) + (size_t)CARRY_VEC * ((size_t)my_shard * seg + my);
        uint4 v[CARRY_VEC];
#pragma unroll
        for (int q = 0; q < CARRY_VEC; ++q) v[q] = rec[q];
        // (component i of the record by CONSTANT index: no pointer into the register array, which would put it in scratch)
        auto w = [&](int i) -> uint32_t { const uint4 &q = v[i >> 2]; return (i & 3) == 0 ? q.x : (i & 3) == 1 ? q.y : (i & 3) == 2 ? q.z : q.w; };
        int o = 0;
#define X(f) e.S.f = w(o++);
        OAK_FOR_SIDE_FIELDS(X)
#undef X
#define X(f) e.F.f = w(o++);
        OAK_FOR_SIDE_FIELDS(X)
#undef X
        e.rng = (uint64_t)w(36) | ((uint64_t)w(37) << 32);
        e.turn = w(38) & 0xFFFF; e.last_damage = w(38) >> 16; e.lm = w(39);
#pragma unroll
        for (int q = 0; q < 24; ++q) e.m[q * 64] = w(40 + q);
        g.s0 = w(64); g.s1 = w(65); steps = w(66); root = w(67); result = w(68);
        e.actS = e.actF = 0;
        e.gin = (const uint32_t *)(RS_PTR(root_battles, const uint8_t *) + (size_t)root * 384);
      } else if (load) {          // a fresh playout of lane `ln` = (root, replica): mcts.h:250-263
        const uint32_t ln = my_shard * RS_U32(fper) + (my - my_carry);
        root = ln / RS_U32(reps);
        uint32_t *ps = (uint32_t *)RS_PTR(lane_prng, uint8_t *) + 2 * (size_t)ln;
        FastPrng lane;
        lane.s0 = ps[0]; lane.s1 = ps[1];
        const uint32_t hi = lane.next32(), lo = lane.next32(); // the lane's uniform_64 of this step = the playout's own stream
        ps[0] = lane.s0; ps[1] = lane.s1;
        g.s0 = hi; g.s1 = (hi | lo) ? lo : 1u;
        const uint32_t *dsrc = (const uint32_t *)RS_PTR(root_durations, const uint8_t *) + 2 * (size_t)root;
        e.load_battle_global(RS_PTR(root_battles, const uint8_t *) + (size_t)root * 384, dsrc[0], dsrc[1]);
        const uint32_t bh = g.next32(), bl = g.next32();
        e.rng = ((uint64_t)bh << 32) | bl;
        e.randomize_hidden();
        result = RS_PTR(root_results, const uint8_t *)[root];
        steps = 0;
      }
    }
    if (__ballot(root != DONE) == 0) break;
    bool playing = root < DONE && (result & 15) == 0 && steps < max_steps;
    bool sliced = false;
    executed += (uint32_t)__popcll(__ballot(playing));
    if (playing) {
      const uint32_t hi = g.next32(), lo = g.next32(); // uniform_64 = hi << 32 | lo
      result = e.random_step(result, hi, lo);
      ++steps;
      playing = (result & 15) == 0 && steps < max_steps;
      sliced = playing && (steps & slice_mask) == 0; // the slice is over: the playout is carried
    }
    any_playing = __ballot(playing && !sliced) != 0;
    // (retiring in batches as well -- finished and sliced lanes waiting for the iteration in front of a refill -- was measured and
    // dropped: nothing at 256 roots, 1.53 -> 1.60 ms for a rank's 32 roots; same for the queue kernel's publishing)
    const uint64_t cm = __ballot(sliced);
    uint32_t slot = 0;
    if (cm) { // wave-uniform: one ticket range per wave for the playouts whose slice is over
      const uint32_t leader = (uint32_t)__ffsll((unsigned long long)cm) - 1;
      uint32_t base = 0;
      if (wl == leader) base = atomicAdd(RS_PTR(cout_count, uint32_t *) + home * CTL_STRIDE, (uint32_t)__popcll(cm));
      slot = (uint32_t)__builtin_amdgcn_readlane((int)base, (int)leader) + (uint32_t)__popcll(cm & ((1ull << wl) - 1));
    }
    if (root < DONE && (!playing || sliced)) { // retire the lane: credit a finished playout to this step / carry an unfinished one
      if (!sliced) {
        const uint32_t t = result & 15;
        const unsigned long long v2 = t == R_WIN ? 2ull : t == R_LOSE ? 0ull : 1ull; // 2 x {1, 0, 0.5}: mcts.h:481-495
        atomicAdd(RS_PTR(acc, unsigned long long *) + (size_t)home * RS_U32(acc_stride) + root, 1ull | (v2 << 32));
      } else if (slot < seg) {
        uint32_t w[CARRY_VEC * 4];
        int o = 0;
#define X(f) w[o++] = e.S.f;
        OAK_FOR_SIDE_FIELDS(X)
#undef X
#define X(f) w[o++] = e.F.f;
        OAK_FOR_SIDE_FIELDS(X)
#undef X
        w[36] = (uint32_t)e.rng; w[37] = (uint32_t)(e.rng >> 32); w[38] = e.turn | (e.last_damage << 16); w[39] = e.lm;
#pragma unroll
        for (int q = 0; q < 24; ++q) w[40 + q] = e.m[q * 64];
        w[64] = g.s0; w[65] = g.s1; w[66] = steps; w[67] = root; w[68] = result; w[69] = w[70] = w[71] = 0;
        uint4 *rec = RS_PTR(cout_state, uint4 *) + (size_t)CARRY_VEC * ((size_t)home * seg + slot);
#pragma unroll
        for (int q = 0; q < CARRY_VEC; ++q) rec[q] = make_uint4(w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]);
      } else atomicOr(RS_PTR(err, uint32_t *), 1u);
      root = NONE;
    }
  }
  if (wl == 0 && executed) atomicAdd(RS_PTR(turn_steps, unsigned long long *), (unsigned long long)executed);
#undef RS_PTR
#undef RS_U32
}
// The launch's report: report[r] = the shards' accumulators of root r added up; report[n_roots + 1] = playouts carried into the next
// step (sum over the shards, each clamped to its segment) | error word << 32.  (report[n_roots], the turn-steps, is written by the waves.)
__global__ __launch_bounds__(256) void k_root_step_report(const unsigned long long *acc, uint32_t acc_stride, uint32_t n_roots, const uint32_t *cout_count,
                                                          const uint32_t *err, uint32_t seg, unsigned long long *report) {
  const uint32_t r = blockIdx.x * 256 + threadIdx.x;
  if (r < n_roots) {
    unsigned long long a = 0;
#pragma unroll
    for (int s = 0; s < ROOT_SHARDS; ++s) a += acc[(size_t)s * acc_stride + r];
    report[r] = a;
  }
  if (blockIdx.x == 0 && threadIdx.x < 64) {
    uint32_t c = threadIdx.x < ROOT_SHARDS ? cout_count[threadIdx.x * CTL_STRIDE] : 0u;
    if (c > seg) c = seg;
#pragma unroll
    for (int o = 4; o > 0; o >>= 1) c += __shfl_down(c, o);
    if (threadIdx.x == 0) report[n_roots + 1] = (unsigned long long)c | ((unsigned long long)*err << 32);
  }
}

// ---- batched single update -------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void k_update(uint8_t *battles, const uint8_t *c1, const uint8_t *c2,
                                                  uint8_t *durations, uint8_t *actions, const uint8_t *overrides,
                                                  uint32_t n, uint8_t *results) {
  extern __shared__ __align__(16) uint8_t smem[];
  lds_u32 *state = (lds_u32 *)smem;
  Tables T = stage_default_tables((lds_u8 *)smem + STATE_LDS_BYTES);
  const uint32_t base = blockIdx.x * BLOCK;
  const uint32_t count = min((uint32_t)BLOCK, n - base);
  load_state(state, battles, base, count);
  __syncthreads();
  const uint32_t tid = threadIdx.x, lane = base + tid;
  if (tid < count) {
    Engine<BLOCK, true> e;
    e.m = state + tid;
    e.T = T;
    uint32_t *d = (uint32_t *)durations + 2 * (size_t)lane;
    e.dur64 = (uint64_t)d[0] | ((uint64_t)d[1] << 32);
    e.over16 = overrides ? (uint32_t)overrides[(size_t)lane * 16] | ((uint32_t)overrides[(size_t)lane * 16 + 8] << 8) : 0;
    e.act0 = e.act1 = 0;
    results[lane] = (uint8_t)e.update(c1[lane], c2[lane]);
    d[0] = e.dur_of(0);
    d[1] = e.dur_of(1);
    if (actions) {
      uint64_t *ad = (uint64_t *)actions + 2 * (size_t)lane;
      ad[0] = e.act0;
      ad[1] = e.act1;
    }
  }
  __syncthreads();
  store_state(state, battles, base, count);
}

__global__ __launch_bounds__(BLOCK) void k_choices(const uint8_t *battles, const uint8_t *results, int player,
                                                   uint8_t *out, uint8_t *counts, uint32_t n) {
  extern __shared__ __align__(16) uint8_t smem[];
  lds_u32 *state = (lds_u32 *)smem;
  const uint32_t base = blockIdx.x * BLOCK;
  const uint32_t count = min((uint32_t)BLOCK, n - base);
  load_state(state, battles, base, count);
  __syncthreads();
  const uint32_t tid = threadIdx.x, lane = base + tid;
  if (tid < count) {
    Engine<BLOCK, false> e;
    e.m = state + tid;
    const uint32_t r = results[lane];
    auto c = e.choices(player, player == 0 ? (r >> 4) & 3 : (r >> 6) & 3);
    counts[lane] = (uint8_t)c.n;
    for (uint32_t i = 0; i < OAKGPU_MAX_CHOICES; ++i) out[(size_t)lane * OAKGPU_MAX_CHOICES + i] = i < c.n ? (uint8_t)c.get(i) : 0;
  }
}

// ---- one level of a batch of tree descents (MCTS::Search::run_iteration, mcts.h:304-389, per lane) --------
// Applies the joint action the host-side bandits selected, reports the result, the 16-byte observation key
// (pkmn_gen1_battle_options_chance_actions: the tree edge key, mcts.h:93-98,359) and the legal choices of BOTH
// players in the new state, so that one device round trip per tree level is enough.  c1 == 0xFF: the lane's
// descent has already ended (leaf or terminal) -- its state is left untouched.  `rolls` != 39 clamps the damage
// rolls like battle_options_set (mcts.h:569-604): override bytes from the last two bytes of battle.rng.
__device__ __forceinline__ uint32_t roll_byte(uint32_t rolls, uint32_t seed) {
  if (rolls == 1) return 236;
  return 217 + (38 / (rolls - 1)) * (seed % rolls);
}
__global__ __launch_bounds__(BLOCK) void k_tree_step(uint8_t *battles, uint8_t *durations, uint8_t *results, const uint8_t *c1,
                                                     const uint8_t *c2, uint32_t n, uint32_t rolls, uint8_t *actions,
                                                     uint8_t *ch1, uint8_t *cnt1, uint8_t *ch2, uint8_t *cnt2) {
  extern __shared__ __align__(16) uint8_t smem[];
  lds_u32 *state = (lds_u32 *)smem;
  Tables T = stage_default_tables((lds_u8 *)smem + STATE_LDS_BYTES);
  const uint32_t base = blockIdx.x * BLOCK;
  const uint32_t count = min((uint32_t)BLOCK, n - base);
  load_state(state, battles, base, count);
  __syncthreads();
  const uint32_t tid = threadIdx.x, lane = base + tid;
  if (tid < count && c1[lane] != 0xFF) {
    Engine<BLOCK, true> e;
    e.m = state + tid;
    e.T = T;
    uint32_t *d = (uint32_t *)durations + 2 * (size_t)lane;
    e.dur64 = (uint64_t)d[0] | ((uint64_t)d[1] << 32);
    e.over16 = 0;
    if (rolls != 39) {
      const uint32_t hi = e.r32(B_RNG + 4);
      e.over16 = roll_byte(rolls, (hi >> 16) & 0xFF) | (roll_byte(rolls, hi >> 24) << 8);
    }
    e.act0 = e.act1 = 0;
    const uint32_t r = e.update(c1[lane], c2[lane]);
    results[lane] = (uint8_t)r;
    d[0] = e.dur_of(0);
    d[1] = e.dur_of(1);
    uint64_t *ad = (uint64_t *)actions + 2 * (size_t)lane;
    ad[0] = e.act0;
    ad[1] = e.act1;
#pragma unroll 1
    for (int pl = 0; pl < 2; ++pl) {
      auto c = e.choices(pl, pl == 0 ? (r >> 4) & 3 : (r >> 6) & 3);
      uint8_t *out = (pl ? ch2 : ch1) + (size_t)lane * OAKGPU_MAX_CHOICES;
      (pl ? cnt2 : cnt1)[lane] = (r & 15) ? 0 : (uint8_t)c.n;
      for (uint32_t i = 0; i < OAKGPU_MAX_CHOICES; ++i) out[i] = i < c.n ? (uint8_t)c.get(i) : 0;
    }
  }
  __syncthreads();
  store_state(state, battles, base, count);
}

// Round 5: the same tree level on the REGISTER-resident engine, staged like k_rollout_staged -- one wave per workgroup, its 64
// battles moved between global memory and LDS with coalesced 1-KB accesses, 35 KB of LDS (four workgroups per CU) instead of
// k_tree_step's 100 KB workgroup of four waves whose every field access is an LDS round trip.  A search level is a launch of
// 4-32 k lanes that does not fill the device: its time is the latency of ONE wave's turn-step, which is what the register engine
// halves (6.4 us for a dense lone wave).  Chance actions (EngineR<.., TRACK_ACTIONS>) and the damage-roll clamp (the override byte
// in SideR::misc) are the register engine's own; outputs are byte for byte k_tree_step's (tests/test_gpu_parity.py:
// test_tree_step_levels_match_the_oracle runs both against the oracle).
struct TreeStepArgs {
  uint8_t *battles, *durations, *results;
  const uint8_t *c1, *c2;
  uint8_t *actions, *ch1, *cnt1, *ch2, *cnt2;
  uint32_t n, rolls;
};
__global__ __launch_bounds__(64, 2) void k_tree_step_staged(TreeStepArgs a) {
  extern __shared__ __align__(16) uint8_t smem[];
  lds_u32 *party = (lds_u32 *)smem;
  using ER = EngineR<64, true, true>;
  Tables T = stage_default_tables((lds_u8 *)smem + ER::PARTY_WORDS * 64 * 4);
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  typedef OAK_LDS u32x4 lds_u128;
  lds_u32 *stage = (lds_u32 *)((lds_u8 *)smem + ER::PARTY_WORDS * 64 * 4 + TABLE_LDS_PAD);
  static_assert(sizeof(TreeStepArgs) <= STAGED_COLD_BYTES, "parked arguments fit");
  lds_u32 *cold = stage + 64 * STAGE_STRIDE; // the pointers are parked in LDS across the turn-step (k_rollout_staged: the SGPRs they would hold are needed for exec masks)
  if (threadIdx.x < sizeof(TreeStepArgs) / 4) cold[threadIdx.x] = ((const uint32_t *)&a)[threadIdx.x];
#define TS_PTR(field, type) cold_ptr_at<type>(cold, offsetof(TreeStepArgs, field))
  const uint32_t tid = threadIdx.x, base = blockIdx.x * 64, lane = base + tid;
  const uint32_t cnt = a.n - base < 64 ? a.n - base : 64; // battles of this wave
  {
    const u32x4 *src = (const u32x4 *)(a.battles + (size_t)base * 384);
    u32x4 t[24];
#pragma unroll
    for (int k = 0; k < 24; ++k) { const uint32_t i = k * 64 + tid; t[k] = src[i < cnt * 24 ? i : 0]; }
#pragma unroll
    for (int k = 0; k < 24; ++k) { const uint32_t i = k * 64 + tid, b = i / 24, w = i - b * 24; *(lds_u128 *)(stage + b * STAGE_STRIDE + 4 * w) = t[k]; }
  }
  __syncthreads();
  const uint32_t n_all = (uint32_t)__builtin_amdgcn_readfirstlane((int)cold[offsetof(TreeStepArgs, n) / 4]);
  const uint32_t rolls = (uint32_t)__builtin_amdgcn_readfirstlane((int)cold[offsetof(TreeStepArgs, rolls) / 4]);
  const uint32_t pc1 = lane < n_all ? TS_PTR(c1, const uint8_t *)[lane] : 0xFFu;
  if (pc1 != 0xFF) {
    ER e;
    e.m = party + tid;
    e.T = T;
    uint32_t *d = TS_PTR(durations, uint32_t *) + 2 * (size_t)lane;
    e.load_battle_global((const uint8_t *)(stage + tid * STAGE_STRIDE), d[0], d[1]);
    if (rolls != 39) { // battle_options_set's clamp (mcts.h:569-604): override bytes from the last two bytes of battle.rng
      const uint32_t hi = (uint32_t)(e.rng >> 32);
      e.S.misc |= roll_byte(rolls, (hi >> 16) & 0xFF) << 16;
      e.F.misc |= roll_byte(rolls, hi >> 24) << 16;
    }
    e.actS = e.actF = 0;
    const uint32_t r = e.update(pc1, TS_PTR(c2, const uint8_t *)[lane]);
    TS_PTR(results, uint8_t *)[lane] = (uint8_t)r;
    d[0] = e.S.dur;
    d[1] = e.F.dur;
    uint64_t *ad = TS_PTR(actions, uint64_t *) + 2 * (size_t)lane;
    ad[0] = e.actS;
    ad[1] = e.actF;
#pragma unroll 1
    for (int pl = 0; pl < 2; ++pl) {
      const auto c = e.choices(pl ? e.F : e.S, pl == 0 ? (r >> 4) & 3 : (r >> 6) & 3);
      uint8_t *out = (pl ? TS_PTR(ch2, uint8_t *) : TS_PTR(ch1, uint8_t *)) + (size_t)lane * OAKGPU_MAX_CHOICES;
      (pl ? TS_PTR(cnt2, uint8_t *) : TS_PTR(cnt1, uint8_t *))[lane] = (r & 15) ? 0 : (uint8_t)c.n;
      for (uint32_t i = 0; i < OAKGPU_MAX_CHOICES; ++i) out[i] = i < c.n ? (uint8_t)c.get(i) : 0;
    }
    e.store_battle_global((uint8_t *)(stage + tid * STAGE_STRIDE));
  }
  __syncthreads();
  u32x4 *dst = (u32x4 *)(TS_PTR(battles, uint8_t *) + (size_t)base * 384);
  uint32_t cnt_out = cnt;
  asm volatile("" : "+s"(cnt_out)); // (opaque: k_rollout_staged -- the bounds tests' lane masks would otherwise live across the turn-step)
#pragma unroll
  for (int k = 0; k < 24; ++k) {
    const uint32_t i = k * 64 + tid, b = i / 24, w = i - b * 24;
    if (i < cnt_out * 24) dst[i] = *(const lds_u128 *)(stage + b * STAGE_STRIDE + 4 * w);
  }
#undef TS_PTR
}

// ---- PokeEngine::Eval (cpp/include/search/poke-engine-evaluate.h:9-204): the hand-written fp32 position score the
// reference uses as its default data-generation evaluator.  One lane per battle, straight from the AoS bytes.
__device__ __forceinline__ float pe_boost(uint32_t nib) { // get_boost_multiplier (:52-85) of a 4-bit two's-complement stage
  const int st = (int)((nib ^ 8) - 8);
  const int a = st < 0 ? -st : st;
  const float m = a == 0 ? 0.0f : a == 1 ? 1.0f : a == 2 ? 2.0f : a == 3 ? 2.5f : a == 4 ? 3.0f : a == 5 ? 3.15f : 3.3f;
  return st < 0 ? -m : m;
}
__device__ __forceinline__ float pe_pokemon(const uint8_t *pk) { // evaluate_pokemon (:131-140), pk = 24 stored bytes
  const uint32_t hp = pk[18] | (pk[19] << 8);
  if (hp == 0) return 0.0f;
  const uint32_t maxhp = pk[0] | (pk[1] << 8), status = pk[20];
  float score = (100.0f * (float)hp) / (float)maxhp;
  float st = 0.0f; // evaluate_status (:106-129)
  if (status == ST_BRN) { // evaluate_burned (:87-104)
    float mult = 0.0f;
    for (int m = 0; m < 4; ++m) {
      const uint32_t w = OAK_MOVE_WORDS[pk[10 + 2 * m]];
      if (((w >> 8) & 0xFF) > 0 && ((w >> 16) & 0xFF) < 8) mult += 1.0f;
    }
    const uint32_t atk = pk[2] | (pk[3] << 8), spc = pk[8] | (pk[9] << 8);
    if (spc > atk) mult *= 0.5f;
    st = mult * -25.0f;
  } else if (status == ST_FRZ) st = -40.0f;
  else if (status == ST_PAR) st = -25.0f;
  else if (status == ST_TOX) st = -30.0f;
  else if (status == ST_PSN) st = -10.0f;
  else if (status & 7) st = -25.0f;
  score += st;
  score = fmaxf(score, 0.0f);
  return score + 30.0f;
}
__device__ __forceinline__ float pe_side(const uint8_t *side) { // evaluate_side / evaluate_active (:142-184)
  float score = 0.0f;
  const uint32_t aid = side[176];
  const uint8_t *stored = side + 24 * (aid ? aid - 1 : 0); // Side::stored(): pokemon[order[0] - 1]
  const uint32_t shp = aid ? (stored[18] | (stored[19] << 8)) : 0;
  if (shp) {
    score += pe_pokemon(stored);
    const uint8_t *ac = side + 144;
    const uint32_t vlo = ac[16] | (ac[17] << 8) | (ac[18] << 16);
    if (vlo & V_LEECHSEED) score += -30.0f;
    if (vlo & V_SUBSTITUTE) score += 40.0f;
    if (vlo & V_CONFUSION) score += -20.0f;
    if (vlo & V_REFLECT) score += 20.0f;
    if (vlo & V_LIGHTSCREEN) score += 20.0f;
    score += 30.0f * pe_boost(ac[12] & 15);   // atk
    score += 15.0f * pe_boost(ac[12] >> 4);   // def
    score += 30.0f * pe_boost(ac[13] >> 4);   // spc
    score += 30.0f * pe_boost(ac[13] & 15);   // spe
  }
  for (int slot = 2; slot <= 6; ++slot) {
    const uint32_t id = side[176 + slot - 1];
    if (id != 0) score += pe_pokemon(side + 24 * (id - 1));
  }
  return score;
}
__global__ void k_poke_engine(const uint8_t *battles, uint32_t n, float root_score, float *values, float *scores) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint8_t *b = battles + (size_t)i * 384;
  const float score = pe_side(b) - pe_side(b + 184);
  if (scores) scores[i] = score;
  if (values) values[i] = 1.0f / (1.0f + expf(-0.0125f * (score - root_score))); // scaled_sigmoid (:11-13), Eval::evaluate (:195-201)
}

// ---- PKMN::battle / Init::init_side (cpp/include/libpkmn/init.h:35-40,90-154) ---------------
template <class E>
__device__ void init_from_teams(E &e, const uint8_t *teams /* 60 B, this lane */, uint32_t seed_lo, uint32_t seed_hi) {
  for (int w = 0; w < STATE_WORDS; ++w) e.w32(4 * w, 0);
  for (int s = 0; s < 2; ++s) {
    const int so = s * SIDE_SZ;
    for (int i = 0; i < 6; ++i) {
      const uint8_t *set = teams + (s * 6 + i) * 5;
      const int pk = so + PK_SZ * i;
      const uint32_t sp = set[0];
      e.w8(pk + P_SPECIES, sp);
      if (sp == 0) { if (i == 0) e.w8(so + O_ORDER, 1); continue; }
      const uint32_t w0 = e.T.sp0[sp], w1 = e.T.sp1[sp];
      const uint32_t level = 100;
      auto stat = [&](uint32_t base, bool hp) { return (2 * (base + 15) + 63) * level / 100 + (hp ? level + 10 : 5); };
      const uint32_t hp = stat(w0 & 0xFF, true);
      e.w16(pk + P_HP_MAX, hp);
      e.w16(pk + P_ATK, stat((w0 >> 8) & 0xFF, false));
      e.w16(pk + P_DEF, stat((w0 >> 16) & 0xFF, false));
      e.w16(pk + P_SPE, stat(w0 >> 24, false));
      e.w16(pk + P_SPC, stat(w1 & 0xFF, false));
      for (int mi = 0; mi < 4; ++mi) {
        const uint32_t id = set[1 + mi];
        e.w8(pk + P_MOVES + 2 * mi, id);
        e.w8(pk + P_MOVES + 2 * mi + 1, id ? e.T.maxpp[id] : 0);
      }
      e.w16(pk + P_HP, hp);
      e.w8(pk + P_TYPES, ((w1 >> 8) & 0xFF) | (((w1 >> 16) & 0xFF) << 4));
      e.w8(pk + P_LEVEL, level);
      e.w8(so + O_ORDER + i, i + 1);
    }
  }
  e.w32(B_RNG, seed_lo);
  e.w32(B_RNG + 4, seed_hi);
}

__global__ __launch_bounds__(BLOCK) void k_init(const uint8_t *teams, const uint64_t *seeds, uint32_t n, int first_update,
                                                uint8_t *battles, uint8_t *durations, uint8_t *results) {
  extern __shared__ __align__(16) uint8_t smem[];
  lds_u32 *state = (lds_u32 *)smem;
  Tables T = stage_default_tables((lds_u8 *)smem + STATE_LDS_BYTES);
  __syncthreads();
  const uint32_t base = blockIdx.x * BLOCK;
  const uint32_t count = min((uint32_t)BLOCK, n - base);
  const uint32_t tid = threadIdx.x, lane = base + tid;
  if (tid < count) {
    Engine<BLOCK, false> e;
    e.m = state + tid;
    e.T = T;
    e.dur64 = 0;
    e.over16 = 0;
    const uint64_t sd = seeds[lane];
    init_from_teams(e, teams + (size_t)lane * OAKGPU_TEAMS_SIZE, (uint32_t)sd, (uint32_t)(sd >> 32));
    uint32_t r = mk_result(0, C_MOVE, C_MOVE);
    if (first_update) r = e.update(0, 0);
    if (results) results[lane] = (uint8_t)r;
    if (durations) {
      uint32_t *d = (uint32_t *)durations + 2 * (size_t)lane;
      d[0] = e.dur_of(0);
      d[1] = e.dur_of(1);
    }
  }
  __syncthreads();
  store_state(state, battles, base, count);
}

__global__ __launch_bounds__(BLOCK) void k_random_ou(uint64_t seed0, uint32_t n, const uint8_t *legal, int n_legal,
                                                     const uint8_t *pools, const uint8_t *sizes, uint8_t *battles,
                                                     uint8_t *durations, uint8_t *prng, uint8_t *results) {
  extern __shared__ __align__(16) uint8_t smem[];
  lds_u32 *state = (lds_u32 *)smem;
  Tables T = stage_default_tables((lds_u8 *)smem + STATE_LDS_BYTES);
  __syncthreads();
  const uint32_t base = blockIdx.x * BLOCK;
  const uint32_t count = min((uint32_t)BLOCK, n - base);
  const uint32_t tid = threadIdx.x, lane = base + tid;
  if (tid < count) {
    Engine<BLOCK, false> e;
    e.m = state + tid;
    e.T = T;
    e.dur64 = 0;
    e.over16 = 0;
    FastPrng g;
    g.seed(seed0 + lane);
    uint8_t teams[OAKGPU_TEAMS_SIZE];
    for (int i = 0; i < OAKGPU_TEAMS_SIZE; ++i) teams[i] = 0;
    for (int s = 0; s < 2; ++s)
      for (int k = 0; k < 6; ++k) {
        uint32_t sp;
        for (;;) {
          sp = legal[g.next32() % (uint32_t)n_legal];
          bool dup = false;
          for (int j = 0; j < k; ++j) dup |= teams[(s * 6 + j) * 5] == sp;
          if (!dup) break;
        }
        uint8_t *set = teams + (s * 6 + k) * 5;
        set[0] = (uint8_t)sp;
        const uint32_t psz = sizes[sp];
        const uint32_t want = psz < 4 ? psz : 4;
        for (uint32_t mi = 0; mi < want; ++mi) {
          for (;;) {
            uint32_t mvid = pools[sp * 48 + g.next32() % psz];
            bool dup = false;
            for (uint32_t j = 0; j < mi; ++j) dup |= set[1 + j] == mvid;
            if (!dup) { set[1 + mi] = (uint8_t)mvid; break; }
          }
        }
      }
    const uint32_t hi = g.next32(), lo = g.next32();
    init_from_teams(e, teams, lo, hi);
    const uint32_t r = e.update(0, 0);
    results[lane] = (uint8_t)r;
    uint32_t *d = (uint32_t *)durations + 2 * (size_t)lane;
    d[0] = e.dur_of(0);
    d[1] = e.dur_of(1);
    uint32_t *pd = (uint32_t *)prng + 2 * (size_t)lane;
    pd[0] = g.s0;
    pd[1] = g.s1;
  }
  __syncthreads();
  store_state(state, battles, base, count);
}

} // namespace oak

// =================================== C ABI ==================================================
struct oakgpu_ctx {
  int device;
  hipStream_t stream;
  bool own_stream;
  uint8_t *d_legal, *d_pools, *d_sizes;
  int n_legal;
  int rollout_engine; // 2 = register-resident (default), 1 = LDS-resident (a second implementation of the same engine: A/B)
  int playouts_per_lane; // > 1: persistent grid of n / this lanes with queue refill (k_rollout_queue)
  uint32_t *d_queue;      // 64 counters: suspended-playout counts of the regrouping rounds, queue-order counters, migration control block
  uint32_t *d_heads;      // the queue kernel's heads: MAX_ROUNDS rounds x 8 heads, each on a 256-byte line of its own
  int rounds;             // regrouping rounds of the queue kernel (1 = none)
  int suspend_below;      // a dry wave with fewer live lanes than this hands them to the next round
  int round_shrink;       // each round launches 1/round_shrink of the previous round's waves
  uint8_t *d_scratch;     // suspended playout state: n x (384 + 8 + 1) bytes + two n-entry index lists
  size_t scratch_n;
  int rounds_auto;        // 1 (default): every launch is one dispatch; 0 after oakgpu_set_regroup: the rounds it asked for
  int concurrent_hint;    // set by callers that keep several contexts busy at once (the tree search): small launches run in rounds
  int tail_below;         // > 0: a SATURATED launch parks the lanes of dry waves with fewer live lanes than this and ONE follow-up
  int tail_waves;         //      dispatch of this many waves (0 = one per CU) finishes them, a few to a wave (the tail)
  int tail_lanes;         //      lanes per wave of that dispatch that take playouts (0 = all 64)
  int migrate;            // long-playout migration (k_rollout_queue): 0 off, 1 (default) for launches that saturate the device, 2 always
  int migrate_steps;      //   a bulk wave donates a playout still running after this many turn-steps (default 300)
  int migrate_adopters;   //   adopter waves (0 = one per two CUs)
  int migrate_window;     //   ... or whose actives' {slot, hp} stood still for this many turn-steps (0 = off, at most 255)
  int migrate_used;       //   the last queue launch ran with migration: oakgpu_synchronize reports its error word
  int spread_lanes;       // launches that do not fill the device: lanes per wave that take playouts (-1 automatic, 0 / 64 = all)
  int queue_order;        // 1 (default): a saturated launch hands its playouts out likely-longest first (k_queue_order)
  int standstill_skip;    // 1 (default): the queue kernel takes a PROVEN frozen standstill to its last turn-step in one go (exact); 0: plays every turn
  uint32_t *d_order;      // total entries
  size_t order_n;
  int n_cu;               // compute units of the device
  static constexpr int TABLE_SLOTS = 8;
  void *h_table, *d_table; // batch tables of the group launches: pinned host ring -> device ring
  hipEvent_t table_ev[TABLE_SLOTS];
  uint64_t table_next;
  struct Block { void *p; size_t cap; };
  std::vector<Block> stage; // staging buffers of the host-pointer entry points (oakgpu_internal.h)
  size_t stage_cursor;
  Block ws[3];              // leaf-evaluator workspaces: embeddings, policy activations, party work list
  int timing;               // oakgpu_set_kernel_timing
  hipEvent_t tev[4];
  bool tev_valid;
  void *attachment = nullptr;              // oakgpu_internal.h: the tree search's cached batch slots
  void (*attachment_dtor)(void *) = nullptr;
};

static thread_local std::string g_err;
static int fail(hipError_t e, const char *what) {
  g_err = std::string(what) + ": " + hipGetErrorString(e);
  return (int)e;
}
static int bad(const char *what) {
  g_err = what;
  return -1;
}
int oakgpu_fail_hip(int e, const char *what) { return fail((hipError_t)e, what); }
int oakgpu_fail_msg(const char *what) { return bad(what); }
int oakgpu_ctx_device(const oakgpu_ctx *c) { return c->device; }
int oakgpu_ctx_enter(oakgpu_ctx *c) {
  hipError_t e = hipSetDevice(c->device);
  return e == hipSuccess ? 0 : fail(e, "hipSetDevice");
}
static void *grow_block(oakgpu_ctx *c, oakgpu_ctx::Block &b, size_t bytes) {
  if (b.cap >= bytes && b.p) return b.p;
  if (b.p) { // a launch on this context's stream may still use the old block
    (void)hipStreamSynchronize(c->stream);
    (void)hipFree(b.p);
    b.p = nullptr;
    b.cap = 0;
  }
  const size_t cap = (bytes + (1u << 20) - 1) & ~(size_t)((1u << 20) - 1);
  hipError_t e = hipMalloc(&b.p, cap);
  if (e != hipSuccess) { b.p = nullptr; fail(e, "hipMalloc(context buffer)"); return nullptr; }
  b.cap = cap;
  return b.p;
}
void **oakgpu_ctx_timing_events(oakgpu_ctx *c) { return c->timing && c->tev_valid ? (void **)c->tev : nullptr; }
void *oakgpu_ctx_workspace(oakgpu_ctx *c, int slot, size_t bytes) { return grow_block(c, c->ws[slot % 3], bytes ? bytes : 1); }
void oakgpu_stage_begin(oakgpu_ctx *c) { c->stage_cursor = 0; }
void oakgpu_stage_end(oakgpu_ctx *c) { (void)hipStreamSynchronize(c->stream); }
void *oakgpu_stage_get(oakgpu_ctx *c, size_t bytes) {
  if (c->stage_cursor >= c->stage.size()) c->stage.push_back(oakgpu_ctx::Block{nullptr, 0});
  return grow_block(c, c->stage[c->stage_cursor++], bytes ? bytes : 1);
}
void *oakgpu_ctx_stream(const oakgpu_ctx *c) { return (void *)c->stream; }
int oakgpu_ctx_set_concurrent_hint(oakgpu_ctx *c, int on) { const int old = c->concurrent_hint; c->concurrent_hint = on; return old; }
void *oakgpu_ctx_attachment(const oakgpu_ctx *c) { return c->attachment; }
void oakgpu_ctx_set_attachment(oakgpu_ctx *c, void *p, void (*dtor)(void *)) {
  if (c->attachment && c->attachment_dtor) c->attachment_dtor(c->attachment);
  c->attachment = p;
  c->attachment_dtor = dtor;
}
#define HIPCHK(x) do { hipError_t _e = (x); if (_e != hipSuccess) return fail(_e, #x); } while (0)

// ---- host-buffer conveniences (PCIe-inclusive; never the benchmarked path) ------------------
namespace {
struct DevBuf { // one staging buffer of the current host call (owned by the context's grow-only cache)
  void *p = nullptr;
  size_t bytes = 0;
  hipError_t alloc(size_t b, oakgpu_ctx *c) { bytes = b; p = oakgpu_stage_get(c, b); return p ? hipSuccess : hipErrorOutOfMemory; }
};
} // namespace

extern "C" {

const char *oakgpu_last_error(void) { return g_err.c_str(); }

int oakgpu_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

static int set_lds_limits() {
  HIPCHK(hipFuncSetAttribute((const void *)oak::k_rollout<64>, hipFuncAttributeMaxDynamicSharedMemorySize, oak::STATE_WORDS * 64 * 4 + oak::TABLE_LDS_PAD));
  HIPCHK(hipFuncSetAttribute((const void *)oak::k_rollout_regs<64>, hipFuncAttributeMaxDynamicSharedMemorySize, 24 * 64 * 4 + oak::TABLE_LDS_PAD));
  HIPCHK(hipFuncSetAttribute((const void *)oak::k_rollout_staged, hipFuncAttributeMaxDynamicSharedMemorySize, oak::STAGED_LDS_BYTES));
  HIPCHK(hipFuncSetAttribute((const void *)oak::k_rollout_draws<64>, hipFuncAttributeMaxDynamicSharedMemorySize, 24 * 64 * 4 + oak::TABLE_LDS_PAD));
  HIPCHK(hipFuncSetAttribute((const void *)oak::k_rollout_queue<64, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 24 * 64 * 4 + oak::TABLE_LDS_PAD + oak::COLD_LDS_BYTES));
  HIPCHK(hipFuncSetAttribute((const void *)oak::k_update, hipFuncAttributeMaxDynamicSharedMemorySize, oak::ENGINE_LDS_BYTES));
  HIPCHK(hipFuncSetAttribute((const void *)oak::k_choices, hipFuncAttributeMaxDynamicSharedMemorySize, oak::ENGINE_LDS_BYTES));
  HIPCHK(hipFuncSetAttribute((const void *)oak::k_tree_step, hipFuncAttributeMaxDynamicSharedMemorySize, oak::ENGINE_LDS_BYTES));
  HIPCHK(hipFuncSetAttribute((const void *)oak::k_tree_step_staged, hipFuncAttributeMaxDynamicSharedMemorySize, oak::STAGED_LDS_BYTES));
  HIPCHK(hipFuncSetAttribute((const void *)oak::k_init, hipFuncAttributeMaxDynamicSharedMemorySize, oak::ENGINE_LDS_BYTES));
  HIPCHK(hipFuncSetAttribute((const void *)oak::k_random_ou, hipFuncAttributeMaxDynamicSharedMemorySize, oak::ENGINE_LDS_BYTES));
  return 0;
}

int oakgpu_create(oakgpu_ctx **out, int device) {
  if (!out) return bad("oakgpu_create: null out");
  int n = 0;
  HIPCHK(hipGetDeviceCount(&n));
  if (n <= 0) return bad("oakgpu_create: no HIP device (this library has no CPU fallback)");
  if (device < 0 || device >= n) return bad("oakgpu_create: device index out of range");
  HIPCHK(hipSetDevice(device));
  {
    // per-DEVICE initialisation, once per process and device (round-2 advice: every context repeated it, on the NULL stream):
    // kernel attributes, and the engine's table image built on a non-blocking stream of its own -- it neither serialises
    // with the caller's blocking streams nor rewrites the image under another context's running kernels
    static std::mutex init_mu;
    static bool init_done[64] = {};
    std::lock_guard<std::mutex> lock(init_mu);
    if (device >= 64 || !init_done[device]) {
      if (int r = set_lds_limits()) return r;
      if (int r = oakgpu_leaf_set_lds_limits()) return r;
      hipStream_t s0;
      HIPCHK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
      hipLaunchKernelGGL(oak::k_build_table_image, dim3(1), dim3(64), oak::TABLE_LDS_PAD, s0);
      hipError_t le = hipGetLastError();
      hipError_t se = hipStreamSynchronize(s0);
      (void)hipStreamDestroy(s0);
      if (le != hipSuccess) return fail(le, "k_build_table_image launch");
      if (se != hipSuccess) return fail(se, "k_build_table_image");
      if (device < 64) init_done[device] = true;
    }
  }
  oakgpu_ctx *c = new oakgpu_ctx();
  c->device = device;
  c->own_stream = true;
  c->d_legal = c->d_pools = c->d_sizes = nullptr;
  c->n_legal = 0;
  c->playouts_per_lane = 2;
  if (const char *env = getenv("OAKGPU_PLAYOUTS_PER_LANE")) c->playouts_per_lane = atoi(env) > 0 ? atoi(env) : 1;
  c->d_queue = nullptr;
  c->d_heads = nullptr;
  c->d_scratch = nullptr;
  c->scratch_n = 0;
  c->h_table = c->d_table = nullptr;
  c->table_next = 0;
  c->stage_cursor = 0;
  c->ws[0] = c->ws[1] = c->ws[2] = oakgpu_ctx::Block{nullptr, 0};
  c->timing = 0;
  c->tev_valid = false;
  c->rounds_auto = 1;
  c->concurrent_hint = 0;
  if (const char *env = getenv("OAKGPU_ROUNDS_AUTO")) c->rounds_auto = atoi(env) != 0;
  {
    hipDeviceProp_t prop;
    hipError_t pe = hipGetDeviceProperties(&prop, device);
    if (pe != hipSuccess) { delete c; return fail(pe, "hipGetDeviceProperties"); }
    c->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  c->rounds = 4;
  c->suspend_below = 32;
  c->round_shrink = 3;
  c->tail_below = 0;
  c->tail_waves = 0;
  if (const char *env = getenv("OAKGPU_TAIL_BELOW")) c->tail_below = atoi(env) < 0 ? 0 : atoi(env) > 64 ? 64 : atoi(env);
  if (const char *env = getenv("OAKGPU_TAIL_WAVES")) c->tail_waves = atoi(env) < 0 ? 0 : atoi(env);
  c->spread_lanes = -1; // automatic
  if (const char *env = getenv("OAKGPU_SPREAD_LANES")) c->spread_lanes = atoi(env) < -1 ? -1 : atoi(env) > 64 ? 64 : atoi(env);
  c->queue_order = 1;
  if (const char *env = getenv("OAKGPU_QUEUE_ORDER")) c->queue_order = atoi(env) != 0;
  c->standstill_skip = 1;
  if (const char *env = getenv("OAKGPU_STANDSTILL_SKIP")) c->standstill_skip = atoi(env) != 0;
  c->migrate = 1;
  c->migrate_steps = 300;
  c->migrate_adopters = 0;
  c->migrate_window = 48;
  c->migrate_used = 0;
  if (const char *env = getenv("OAKGPU_MIGRATE")) c->migrate = atoi(env) < 0 ? 0 : atoi(env) > 2 ? 2 : atoi(env);
  if (const char *env = getenv("OAKGPU_MIGRATE_STEPS")) c->migrate_steps = atoi(env) < 1 ? 1 : atoi(env);
  if (const char *env = getenv("OAKGPU_MIGRATE_ADOPTERS")) c->migrate_adopters = atoi(env) < 0 ? 0 : atoi(env);
  if (const char *env = getenv("OAKGPU_MIGRATE_WINDOW")) c->migrate_window = atoi(env) < 0 ? 0 : atoi(env) > 255 ? 255 : atoi(env);
  c->d_order = nullptr;
  c->order_n = 0;
  c->tail_lanes = 0;
  if (const char *env = getenv("OAKGPU_TAIL_LANES")) c->tail_lanes = atoi(env) < 0 ? 0 : atoi(env) > 64 ? 64 : atoi(env);
  if (const char *env = getenv("OAKGPU_ROUNDS")) c->rounds = atoi(env) < 1 ? 1 : atoi(env) > 8 ? 8 : atoi(env);
  if (const char *env = getenv("OAKGPU_SUSPEND_BELOW")) c->suspend_below = atoi(env) < 0 ? 0 : atoi(env) > 64 ? 64 : atoi(env);
  if (const char *env = getenv("OAKGPU_ROUND_SHRINK")) c->round_shrink = atoi(env) < 1 ? 1 : atoi(env);
  c->rollout_engine = 2;
  if (const char *env = getenv("OAKGPU_ROLLOUT_ENGINE")) c->rollout_engine = atoi(env) == 1 ? 1 : 2;
  hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
  if (e != hipSuccess) { delete c; return fail(e, "hipStreamCreate"); }
  *out = c;
  return 0;
}

void oakgpu_destroy(oakgpu_ctx *c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  if (c->attachment && c->attachment_dtor) { c->attachment_dtor(c->attachment); c->attachment = nullptr; }
  if (c->own_stream) (void)hipStreamDestroy(c->stream);
  if (c->d_legal) (void)hipFree(c->d_legal);
  if (c->d_pools) (void)hipFree(c->d_pools);
  if (c->d_sizes) (void)hipFree(c->d_sizes);
  if (c->d_queue) (void)hipFree(c->d_queue);
  if (c->d_heads) (void)hipFree(c->d_heads);
  if (c->d_scratch) (void)hipFree(c->d_scratch);
  if (c->d_order) (void)hipFree(c->d_order);
  for (auto &b : c->stage) if (b.p) (void)hipFree(b.p);
  for (auto &b : c->ws) if (b.p) (void)hipFree(b.p);
  if (c->tev_valid) for (auto &e : c->tev) (void)hipEventDestroy(e);
  if (c->h_table) {
    (void)hipHostFree(c->h_table);
    (void)hipFree(c->d_table);
    for (int i = 0; i < oakgpu_ctx::TABLE_SLOTS; ++i) (void)hipEventDestroy(c->table_ev[i]);
  }
  delete c;
}

int oakgpu_set_stream(oakgpu_ctx *c, void *hip_stream) {
  if (!c) return bad("null ctx");
  if (c->own_stream) { (void)hipStreamSynchronize(c->stream); (void)hipStreamDestroy(c->stream); }
  c->stream = (hipStream_t)hip_stream;
  c->own_stream = false;
  return 0;
}

int oakgpu_set_playouts_per_lane(oakgpu_ctx *c, int k) {
  if (!c || k < 1) return bad("oakgpu_set_playouts_per_lane: bad argument");
  c->playouts_per_lane = k;
  return 0;
}

int oakgpu_set_rollout_engine(oakgpu_ctx *c, int engine) {
  if (!c || engine < 1 || engine > 2) return bad("oakgpu_set_rollout_engine: engine must be 1 (LDS-resident) or 2 (register-resident)");
  c->rollout_engine = engine;
  return 0;
}

int oakgpu_set_regroup(oakgpu_ctx *c, int rounds, int suspend_below, int shrink) {
  if (!c || rounds < 1 || rounds > 8 || suspend_below < 0 || suspend_below > 64 || shrink < 1)
    return bad("oakgpu_set_regroup: bad argument");
  c->rounds = rounds;
  c->suspend_below = suspend_below;
  c->round_shrink = shrink;
  c->rounds_auto = 0; // an explicit setting applies to every launch, saturating or not
  return 0;
}

int oakgpu_set_migration(oakgpu_ctx *c, int mode, int long_steps, int adopters) {
  if (!c || mode < 0 || mode > 2 || long_steps < 1 || adopters < 0) return bad("oakgpu_set_migration: bad argument");
  c->migrate = mode;
  c->migrate_steps = long_steps;
  c->migrate_adopters = adopters;
  return 0;
}

int oakgpu_set_migration_window(oakgpu_ctx *c, int window) {
  if (!c || window < 0 || window > 255) return bad("oakgpu_set_migration_window: window must be 0 (off) .. 255 turn-steps");
  c->migrate_window = window;
  return 0;
}

int oakgpu_get_queue_counters(oakgpu_ctx *c, uint32_t *out64) { // diagnostic: synchronises the stream
  if (!c || !out64) return bad("oakgpu_get_queue_counters: null argument");
  if (!c->d_queue) { memset(out64, 0, 256); return 0; }
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipMemcpyAsync(out64, c->d_queue, 256, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return 0;
}

int oakgpu_set_spread(oakgpu_ctx *c, int lanes) {
  if (!c || lanes < -1 || lanes > 64) return bad("oakgpu_set_spread: lanes must be -1 (automatic) or 0..64");
  c->spread_lanes = lanes;
  return 0;
}

int oakgpu_set_standstill_skip(oakgpu_ctx *c, int on) {
  if (!c) return bad("null ctx");
  c->standstill_skip = on != 0;
  return 0;
}

int oakgpu_set_queue_order(oakgpu_ctx *c, int on) {
  if (!c) return bad("null ctx");
  c->queue_order = on != 0;
  return 0;
}

int oakgpu_set_tail_pack(oakgpu_ctx *c, int below, int waves, int lanes) {
  if (!c || below < 0 || below > 64 || waves < 0 || lanes < 0 || lanes > 64) return bad("oakgpu_set_tail_pack: bad argument");
  c->tail_below = below;
  c->tail_waves = waves;
  c->tail_lanes = lanes;
  return 0;
}

void *oakgpu_get_stream(oakgpu_ctx *c) { return c ? (void *)c->stream : nullptr; }

int oakgpu_set_kernel_timing(oakgpu_ctx *c, int on) {
  if (!c) return bad("null ctx");
  HIPCHK(hipSetDevice(c->device));
  if (on && !c->tev_valid) {
    for (auto &e : c->tev) HIPCHK(hipEventCreate(&e));
    c->tev_valid = true;
  }
  c->timing = on != 0;
  return 0;
}

int oakgpu_get_leaf_kernel_ms(oakgpu_ctx *c, float ms[3]) {
  if (!c || !ms) return bad("oakgpu_get_leaf_kernel_ms: null argument");
  if (!c->timing || !c->tev_valid) return bad("oakgpu_get_leaf_kernel_ms: kernel timing is off (oakgpu_set_kernel_timing)");
  HIPCHK(hipEventSynchronize(c->tev[3]));
  for (int i = 0; i < 3; ++i) HIPCHK(hipEventElapsedTime(&ms[i], c->tev[i], c->tev[i + 1]));
  return 0;
}

int oakgpu_synchronize(oakgpu_ctx *c) {
  if (!c) return bad("null ctx");
  HIPCHK(hipStreamSynchronize(c->stream));
  if (c->d_queue && c->migrate_used) { // a queue launch since the last check migrated playouts: did a bounded wait run out?
    uint32_t err = 0;
    HIPCHK(hipMemcpy(&err, c->d_queue + 63, 4, hipMemcpyDeviceToHost));
    c->migrate_used = 0;
    if (err) { HIPCHK(hipMemsetAsync(c->d_queue + 63, 0, 4, c->stream)); HIPCHK(hipStreamSynchronize(c->stream)); }
    if (err) return bad("rollout: a bounded wait of the long-playout migration ran out (playouts were lost) -- oakgpu_set_migration(ctx, 0, ...) turns it off");
  }
  return 0;
}

#ifdef OAKGPU_SITE_PROFILE
// profile build only (tools/site_profile.sh): read-and-reset the region counters of gen1_regs.hpp
extern "C" int oakgpu_site_profile(unsigned long long *out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(oak::g_site_prof), sizeof(unsigned long long) * 32 * 4) != hipSuccess) return 1;
  if (reset) {
    unsigned long long z[32 * 4] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(oak::g_site_prof), z, sizeof z) != hipSuccess) return 1;
  }
  return 0;
}
#endif

#ifdef OAKGPU_TIMELINE
extern "C" int oakgpu_timeline(unsigned long long *out, int waves) { // profile build only (tools/timeline.py)
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(oak::g_timeline), sizeof(unsigned long long) * 5 * (size_t)waves) != hipSuccess;
}
#endif

static inline uint32_t grid_for(uint32_t n) { return (n + oak::BLOCK - 1) / oak::BLOCK; }

// ---- group launch of the queue kernel --------------------------------------------------------------------------
// The batch table travels through a small ring of pinned host slots -> device slots (async copy on the context's
// stream; a slot is reused only after the event behind its previous copy has completed).
static int launch_group(oakgpu_ctx *c, const oak::BatchDesc *descs, uint32_t count, uint32_t total, uint32_t max_steps, int prep) {
  HIPCHK(hipSetDevice(c->device));
  // (the first clearing covers the sticky word too, ON THE CONTEXT'S STREAM: a hipMemset on the NULL stream is not ordered with a
  // non-blocking stream and could land in the middle of the first launch -- queue heads zeroed under a running kernel)
  if (!c->d_queue) { HIPCHK(hipMalloc((void **)&c->d_queue, 256)); HIPCHK(hipMemsetAsync(c->d_queue, 0, 256, c->stream)); }
  constexpr int MAX_ROUNDS = 16;
  constexpr size_t HEADS_BYTES = (size_t)MAX_ROUNDS * oak::QUEUE_HEADS * oak::QUEUE_HEAD_STRIDE * 4;
  if (!c->d_heads) HIPCHK(hipMalloc((void **)&c->d_heads, HEADS_BYTES));
  if (!c->h_table) {
    HIPCHK(hipHostMalloc((void **)&c->h_table, sizeof(oak::BatchDesc) * oak::MAX_GROUP * oakgpu_ctx::TABLE_SLOTS, hipHostMallocDefault));
    HIPCHK(hipMalloc((void **)&c->d_table, sizeof(oak::BatchDesc) * oak::MAX_GROUP * oakgpu_ctx::TABLE_SLOTS));
    for (int i = 0; i < oakgpu_ctx::TABLE_SLOTS; ++i) HIPCHK(hipEventCreateWithFlags(&c->table_ev[i], hipEventDisableTiming));
  }
  const int slot = (int)(c->table_next++ % oakgpu_ctx::TABLE_SLOTS);
  HIPCHK(hipEventSynchronize(c->table_ev[slot])); // never-recorded events are complete
  oak::BatchDesc *ht = (oak::BatchDesc *)c->h_table + (size_t)slot * oak::MAX_GROUP, *dt = (oak::BatchDesc *)c->d_table + (size_t)slot * oak::MAX_GROUP;
  memcpy(ht, descs, sizeof(oak::BatchDesc) * count);
  HIPCHK(hipMemcpyAsync(dt, ht, sizeof(oak::BatchDesc) * count, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipEventRecord(c->table_ev[slot], c->stream));
  HIPCHK(hipMemsetAsync(c->d_queue, 0, 252, c->stream)); // (word 63 is the sticky migration error word: never cleared here)
  HIPCHK(hipMemsetAsync(c->d_heads, 0, HEADS_BYTES, c->stream));
  // grid: at least `playouts_per_lane` playouts per lane, and never more waves than the device keeps resident
  const uint32_t n64 = (total + 63) / 64;
  uint32_t waves = (n64 + c->playouts_per_lane - 1) / c->playouts_per_lane;
  if (waves < 1) waves = 1;
  const uint32_t resident = (uint32_t)c->n_cu * 4u * 4u; // four waves per SIMD (k_rollout_queue<64, 4>: 128 VGPRs, measured best; 2 and 3 were removed in round 5)
  bool saturated = waves >= resident;
  if (saturated) waves = resident;
  // A launch that leaves wave slots empty uses them: the same playouts in flight on MORE waves, each taking only a few at a
  // time -- a wave of 8 lanes executes the union of 8 playouts' paths instead of 64's, and the device had the issue slots free
  // (round 3, tools/small_launch_sweep.py ... spread: 65,536 playouts 8.5 -> 7.7 ms at 8 lanes per wave, 16,384 playouts 7.8 ->
  // 6.2 ms at 4, a rank's share of configs[3] 9.3 -> 8.3 ms at 16).  Automatic: as few lanes per wave as fill the resident
  // waves, at least 4.  Not for callers that keep several contexts busy (the other context needs the slots).
  uint32_t spread = 0;
  if (!saturated && c->spread_lanes != 0 && !c->concurrent_hint && max_steps > 64) {
    const uint64_t slots = ((uint64_t)total + c->playouts_per_lane - 1) / c->playouts_per_lane; // lanes in flight
    uint32_t lanes = c->spread_lanes > 0 ? (uint32_t)c->spread_lanes : (uint32_t)((slots + resident - 1) / resident);
    if (lanes < 4) lanes = 4;
    if (lanes < 64) {
      uint32_t w = (uint32_t)((slots + lanes - 1) / lanes);
      if (w > resident) w = resident;
      if (w > waves) { waves = w; spread = lanes; }
    }
  }
  // regrouping rounds pay off while the launch leaves SIMDs idle in its tail; a launch that saturates the device
  // for most of its life (a group of batches) runs as a single dispatch (measured: DESIGN.md 3)
  // ... and so does a launch capped at a few steps (stepping a resident batch turn by turn): there is no tail to regroup
  const bool tail_pack = saturated && c->tail_below > 0 && max_steps > 64;
  const uint32_t adopters = (uint32_t)(c->migrate_adopters > 0 ? c->migrate_adopters : (c->n_cu + 1) / 2); // measured best: one per two CUs
  // long-playout migration (k_rollout_queue) replaces the regrouping rounds where it applies: a single dispatch
  const bool migrate = !tail_pack && max_steps > (uint32_t)c->migrate_steps && waves > adopters &&
                       (c->migrate == 2 || (c->migrate == 1 && saturated && max_steps >= 500));
  // Regrouping rounds only on request (oakgpu_set_regroup).  They were the default for launches that do not fill the device while
  // twenty such launches ran side by side on twenty streams (round 1: a parked wave's slots went to another batch); since group
  // launches replaced that, a lone launch is measured FASTER as one dispatch at every size (round 3, tools/small_launch_sweep.py:
  // 16,384 / 65,536 / 131,072 / 262,144 playouts: 10.3 / 11.7 / 11.9 / 12.5 ms with rounds 4 / 32 / 3 against 8.0 / 8.6 / 9.5 / 9.5 ms
  // as a single dispatch with the queue order below)
  // (the tree search keeps two batches in flight on two contexts and asks for the rounds -- oakgpu_ctx_set_concurrent_hint --:
  // behind a single 8 ms dispatch the other context's small per-level kernels queued up: 11 -> 69 ms of waiting per search)
  const int rounds = tail_pack ? 2 : migrate ? 1 : ((!c->rounds_auto || (c->concurrent_hint && !saturated && max_steps > 64)) && c->suspend_below > 0 && waves >= 8) ? c->rounds : 1;
  if ((rounds > 1 || migrate) && c->scratch_n < total) {
    if (c->d_scratch) { HIPCHK(hipStreamSynchronize(c->stream)); HIPCHK(hipFree(c->d_scratch)); c->d_scratch = nullptr; }
    HIPCHK(hipMalloc((void **)&c->d_scratch, (size_t)total * (384 + 8 + 4 + 4) + (((size_t)total + 15) & ~(size_t)15)));
    c->scratch_n = total;
  }
  const size_t lq = 24 * 64 * 4 + oak::TABLE_LDS_PAD + oak::COLD_LDS_BYTES;
  uint8_t *sb = c->d_scratch, *sd = sb ? sb + (size_t)c->scratch_n * 384 : nullptr;
  uint32_t *lists[2] = {sd ? (uint32_t *)(sd + (size_t)c->scratch_n * 8) : nullptr, nullptr};
  lists[1] = lists[0] ? lists[0] + c->scratch_n : nullptr;
  uint8_t *sres = lists[1] ? (uint8_t *)(lists[1] + c->scratch_n) : nullptr;
  const oak::GroupArgs g{dt, count, total, max_steps, prep};
  const uint32_t *order = nullptr;
  if (c->queue_order && max_steps > 250 && total >= 8192) { // (a few thousand playouts have no queue to speak of: they all start at once)
    if (c->order_n < total) {
      if (c->d_order) { HIPCHK(hipStreamSynchronize(c->stream)); HIPCHK(hipFree(c->d_order)); c->d_order = nullptr; }
      HIPCHK(hipMalloc((void **)&c->d_order, (size_t)total * 4));
      c->order_n = total;
    }
    hipLaunchKernelGGL(oak::k_queue_order, dim3((total + 1023) / 1024), dim3(1024), 0, c->stream, g, c->d_order, c->d_queue + 32);
    order = c->d_order;
  }
  if (migrate) HIPCHK(hipMemsetAsync(lists[0], 0, (size_t)total * 4, c->stream)); // the adoption tickets (one per donation at most)
  if (migrate) c->migrate_used = 1; // (sticky, like the device word: back-to-back launches keep an earlier launch's error)
  for (int r = 0; r < rounds; ++r) {
    oak::RoundArgs q{};
    q.order = r == 0 ? order : nullptr;
    q.long_steps = (uint32_t)(c->migrate_steps > 0xFFFF ? 0xFFFF : c->migrate_steps) | (uint32_t)(c->migrate_window > 0 ? c->migrate_window : 256) << 16;
    q.no_skip = c->standstill_skip ? 0u : 1u;
    if (migrate) { q.adopt_ctl = c->d_queue + 40; q.adopt_list = lists[0]; q.n_adopters = adopters; }
    q.list_in = r ? lists[(r - 1) & 1] : nullptr;
    q.n_in = r ? c->d_queue + 2 * r - 1 : nullptr; // = count_out of round r - 1
    q.list_out = lists[r & 1];
    q.count_out = c->d_queue + 2 * r + 1;
    q.sb = sb; q.sd = sd; q.sres = sres;
    q.suspend_below = r + 1 < rounds ? (uint32_t)(tail_pack ? c->tail_below : c->suspend_below) : 0u;
    q.queue = c->d_heads + (size_t)r * oak::QUEUE_HEADS * oak::QUEUE_HEAD_STRIDE;
    q.lanes = (tail_pack && r > 0) ? (uint32_t)c->tail_lanes : (r == 0 ? spread : 0u);
    hipLaunchKernelGGL((oak::k_rollout_queue<64, 4>), dim3(waves), dim3(64), lq, c->stream, g, q);
    waves = tail_pack ? (uint32_t)(c->tail_waves > 0 ? c->tail_waves : c->n_cu) : (waves + c->round_shrink - 1) / c->round_shrink;
    if (waves < 1) waves = 1;
  }
  HIPCHK(hipGetLastError());
  return 0;
}

static int launch_single(oakgpu_ctx *c, const oak::RolloutArgs &a) { // one lane per playout, no queue
  const uint32_t n = a.n;
  const size_t lds64 = oak::STATE_WORDS * 64 * 4 + oak::TABLE_LDS_PAD;
  if (c->rollout_engine == 1) { // LDS-resident engine (gen1_device.hpp), kept for A/B and as a second implementation
    hipLaunchKernelGGL(oak::k_rollout<64>, dim3((n + 63) / 64), dim3(64), lds64, c->stream, a);
  } else {                      // register-resident engine (gen1_regs.hpp)
    static const bool no_staged = getenv("OAKGPU_NO_STAGED") != nullptr; // (A/B)
    if (a.max_steps <= 16 && !no_staged && ((uintptr_t)a.battles & 15) == 0 && (!a.battles_out || ((uintptr_t)a.battles_out & 15) == 0))
      hipLaunchKernelGGL(oak::k_rollout_staged, dim3((n + 63) / 64), dim3(64), oak::STAGED_LDS_BYTES, c->stream, a);
    else
      hipLaunchKernelGGL(oak::k_rollout_regs<64>, dim3((n + 63) / 64), dim3(64), 24 * 64 * 4 + oak::TABLE_LDS_PAD, c->stream, a);
  }
  HIPCHK(hipGetLastError());
  return 0;
}

int oakgpu_rollout_group_dev(oakgpu_ctx *c, const oakgpu_rollout_batch *batches, uint32_t count, uint32_t max_steps, int prep) {
  if (!c) return bad("null ctx");
  if (count == 0) return 0;
  if (!batches) return bad("oakgpu_rollout_group_dev: null batches");
  if (count > (uint32_t)oak::MAX_GROUP) return bad("oakgpu_rollout_group_dev: more than 64 batches in one group");
  oak::BatchDesc descs[oak::MAX_GROUP];
  uint32_t used = 0;
  uint64_t total = 0;
  for (uint32_t i = 0; i < count; ++i) {
    const oakgpu_rollout_batch &b = batches[i];
    if (b.n == 0) continue;
    if (!b.battles || !b.durations || !b.results_in || !b.prng_state || !b.results_out || !b.steps_out || !b.values_out)
      return bad("oakgpu_rollout_group_dev: null required pointer");
    descs[used++] = oak::BatchDesc{b.battles, b.durations, b.results_in, b.prng_state, b.results_out, b.steps_out, b.values_out,
                                   b.battles_out, b.durations_out, (uint32_t)total, b.n};
    total += b.n;
  }
  if (used == 0) return 0;
  if (total >= 0xFFFFFFF0ull) return bad("oakgpu_rollout_group_dev: more than 2^32 playouts in one group");
  // no queue: one launch per batch -- also for launches capped at a few turn-steps (stepping a resident batch turn by turn,
  // BASELINE configs[2]): there is no tail for a queue to fill, and twice the waves hide twice the latency (measured, one
  // turn-step of 65,536 battles: 61 us against 99 us; tools/onestep_latency.py)
  if (c->rollout_engine == 1 || c->playouts_per_lane <= 1 || max_steps <= 16) {
    HIPCHK(hipSetDevice(c->device));
    for (uint32_t i = 0; i < used; ++i) {
      const oak::BatchDesc &d = descs[i];
      if (int r = launch_single(c, oak::RolloutArgs{d.battles, d.durations, d.results_in, d.prng, d.n, max_steps, prep, d.results_out,
                                                    d.steps_out, d.values_out, d.battles_out, d.durations_out})) return r;
    }
    return 0;
  }
  return launch_group(c, descs, used, (uint32_t)total, max_steps, prep);
}

int oakgpu_rollout_dev(oakgpu_ctx *c, const uint8_t *battles, const uint8_t *durations, const uint8_t *results_in,
                       uint8_t *prng_state, uint32_t n, uint32_t max_steps, int prep, uint8_t *results_out,
                       uint32_t *steps_out, float *values_out, uint8_t *battles_out, uint8_t *durations_out) {
  const oakgpu_rollout_batch b{battles, durations, results_in, prng_state, n, results_out, steps_out, values_out, battles_out, durations_out};
  return oakgpu_rollout_group_dev(c, &b, 1, max_steps, prep);
}

// ---- root-parallel search steps in slices (k_root_step; include/oakgpu.h) -----------------------------------------------------
struct oakgpu_root_steps {
  oakgpu_ctx *ctx;
  uint32_t n_roots, reps, slice, max_steps, cap;
  uint4 *state[2];     // carry lists, ping-pong: cap records of oak::CARRY_VEC uint4
  unsigned long long *acc; // ROOT_SHARDS x acc_stride per-root accumulators of the launch in flight
  uint32_t acc_stride;
  uint32_t *ctl;       // counters on 256-byte lines of their own: lines [0, 8) / [8, 16) the two lists' per-shard counts, [16, 24) the queue heads, 24 the sticky error word
  int cur;             // the list the NEXT launch reads
};

int oakgpu_root_steps_create(oakgpu_ctx *c, uint32_t n_roots, uint32_t reps, uint32_t slice, uint32_t max_steps, oakgpu_root_steps **out) {
  if (!c || !out) return bad("oakgpu_root_steps_create: null argument");
  if (n_roots == 0 || reps == 0 || max_steps == 0) return bad("oakgpu_root_steps_create: roots, replicas and max_steps must be positive");
  if (slice & (slice - 1)) return bad("oakgpu_root_steps_create: slice must be a power of two (0 = playouts run to terminal inside their step)");
  if ((uint64_t)n_roots * reps >= 0x40000000ull) return bad("oakgpu_root_steps_create: more than 2^30 playouts per step");
  HIPCHK(hipSetDevice(c->device));
  oakgpu_root_steps *rs = new oakgpu_root_steps{};
  rs->ctx = c; rs->n_roots = n_roots; rs->reps = reps; rs->slice = slice; rs->max_steps = max_steps;
  // every playout in flight is in at most one list: a step adds n_roots * reps and a playout lives ceil(max_steps / slice) launches
  const uint64_t lives = slice ? (max_steps + slice - 1) / slice : 1;
  const uint64_t worst = (uint64_t)n_roots * reps * (lives > 1 ? lives - 1 : 0);
  // ... but the length distribution decays fast (random OU playouts: mean ~100 turn-steps, 99.5 % end before 250): the steady population
  // is ~(mean length / slice - 0.4) steps' worth (measured: 0.17 / 1.14 / 2.66 steps' worth at slice 128 / 64 / 32); 1 + 256 / slice steps'
  // worth leaves 3-4x of that, shard imbalance included; an overflow is reported (sticky error), never silent
  const uint64_t want = (uint64_t)n_roots * reps * (1 + (slice ? 256 / slice : 0));
  rs->cap = (uint32_t)(worst < want ? worst : want);
  rs->cap = (rs->cap + oak::ROOT_SHARDS - 1) / oak::ROOT_SHARDS * oak::ROOT_SHARDS; // (segments of cap / 8 records)
  if (rs->cap == 0) rs->cap = oak::ROOT_SHARDS;
  hipError_t e = hipSuccess;
  for (int k = 0; k < 2 && e == hipSuccess; ++k) e = hipMalloc((void **)&rs->state[k], (size_t)rs->cap * oak::CARRY_VEC * 16);
  rs->acc_stride = (n_roots + 31u) & ~31u; // (whole 256-byte lines per shard)
  if (e == hipSuccess) e = hipMalloc((void **)&rs->acc, (size_t)oak::ROOT_SHARDS * rs->acc_stride * 8);
  constexpr size_t CTL_BYTES = (size_t)25 * oak::CTL_STRIDE * 4;
  if (e == hipSuccess) e = hipMalloc((void **)&rs->ctl, CTL_BYTES);
  if (e == hipSuccess) e = hipMemsetAsync(rs->ctl, 0, CTL_BYTES, c->stream);
  if (e != hipSuccess) { oakgpu_root_steps_destroy(rs); return fail(e, "oakgpu_root_steps_create"); }
  *out = rs;
  return 0;
}

void oakgpu_root_steps_destroy(oakgpu_root_steps *rs) {
  if (!rs) return;
  (void)hipSetDevice(rs->ctx->device);
  (void)hipStreamSynchronize(rs->ctx->stream);
  for (int k = 0; k < 2; ++k) if (rs->state[k]) (void)hipFree(rs->state[k]);
  if (rs->ctl) (void)hipFree(rs->ctl);
  if (rs->acc) (void)hipFree(rs->acc);
  delete rs;
}

// Grow the carry lists to hold at least `playouts` carried playouts (never shrinks).  Synchronises the context's stream; the playouts
// in flight keep their places (each shard's segment is copied to its new base).  The default capacity (1 + 256 / slice steps' worth)
// is 3-4x what random OU roots need; roots whose playouts mostly run into the step cap (stalemates) need up to
// ceil(max_steps / slice) - 1 steps' worth -- a caller that sees `carried` approach the capacity reserves more BEFORE the next launch.
int oakgpu_root_steps_reserve(oakgpu_root_steps *rs, uint64_t playouts) {
  if (!rs) return bad("oakgpu_root_steps_reserve: null argument");
  oakgpu_ctx *c = rs->ctx;
  HIPCHK(hipSetDevice(c->device));
  uint64_t want = (playouts + oak::ROOT_SHARDS - 1) / oak::ROOT_SHARDS * oak::ROOT_SHARDS;
  if (want <= rs->cap) return 0;
  if (want >= 0xFFFFFFF0ull) return bad("oakgpu_root_steps_reserve: more than 2^32 carried playouts");
  HIPCHK(hipStreamSynchronize(c->stream));
  constexpr size_t LINES8 = (size_t)oak::ROOT_SHARDS * oak::CTL_STRIDE;
  uint32_t counts[oak::ROOT_SHARDS];
  for (int sh = 0; sh < oak::ROOT_SHARDS; ++sh) HIPCHK(hipMemcpy(&counts[sh], rs->ctl + rs->cur * LINES8 + (size_t)sh * oak::CTL_STRIDE, 4, hipMemcpyDeviceToHost));
  const size_t old_seg = rs->cap / oak::ROOT_SHARDS, new_seg = want / oak::ROOT_SHARDS, rec = (size_t)oak::CARRY_VEC * 16;
  uint4 *fresh[2] = {nullptr, nullptr};
  for (int k = 0; k < 2; ++k) {
    hipError_t e = hipMalloc((void **)&fresh[k], (size_t)want * rec);
    if (e != hipSuccess) { if (fresh[0]) (void)hipFree(fresh[0]); return fail(e, "oakgpu_root_steps_reserve"); }
  }
  for (int sh = 0; sh < oak::ROOT_SHARDS; ++sh) { // the list the next launch reads; the other one is scratch
    const size_t n = counts[sh] < old_seg ? counts[sh] : old_seg;
    if (n) HIPCHK(hipMemcpy((uint8_t *)fresh[rs->cur] + (size_t)sh * new_seg * rec, (const uint8_t *)rs->state[rs->cur] + (size_t)sh * old_seg * rec, n * rec, hipMemcpyDeviceToDevice));
  }
  for (int k = 0; k < 2; ++k) { (void)hipFree(rs->state[k]); rs->state[k] = fresh[k]; }
  rs->cap = (uint32_t)want;
  return 0;
}
int oakgpu_root_steps_capacity(const oakgpu_root_steps *rs, uint32_t *capacity) {
  if (!rs || !capacity) return bad("oakgpu_root_steps_capacity: null argument");
  *capacity = rs->cap;
  return 0;
}

int oakgpu_root_steps_launch_dev(oakgpu_root_steps *rs, const uint8_t *root_battles, const uint8_t *root_durations,
                                 const uint8_t *root_results, uint8_t *lane_prng, int fresh, unsigned long long *report) {
  if (!rs || !report) return bad("oakgpu_root_steps_launch_dev: null argument");
  // (root_battles also in a drain step: a carried playout reads its Pokemon's immutable data -- stats, move ids, species, types -- from its ROOT's battle)
  if (!root_battles) return bad("oakgpu_root_steps_launch_dev: null root_battles (carried playouts read their root's immutable data: drain steps need it too)");
  if (fresh && (!root_durations || !root_results || !lane_prng)) return bad("oakgpu_root_steps_launch_dev: null root / stream pointer");
  oakgpu_ctx *c = rs->ctx;
  HIPCHK(hipSetDevice(c->device));
  const int in = rs->cur, outl = in ^ 1;
  HIPCHK(hipMemsetAsync(report, 0, ((size_t)rs->n_roots + 2) * 8, c->stream));
  HIPCHK(hipMemsetAsync(rs->acc, 0, (size_t)oak::ROOT_SHARDS * rs->acc_stride * 8, c->stream));
  constexpr size_t LINES8 = (size_t)oak::ROOT_SHARDS * oak::CTL_STRIDE; // words of eight counters
  uint32_t *cnt_in = rs->ctl + in * LINES8, *cnt_out = rs->ctl + outl * LINES8, *heads = rs->ctl + 2 * LINES8, *errw = rs->ctl + 3 * LINES8;
  HIPCHK(hipMemsetAsync(cnt_out, 0, LINES8 * 4, c->stream)); // the list this launch fills
  HIPCHK(hipMemsetAsync(heads, 0, LINES8 * 4, c->stream));   // queue heads
  oak::RootStepArgs a{};
  a.root_battles = root_battles; a.root_durations = root_durations; a.root_results = root_results; a.lane_prng = lane_prng;
  a.cin_state = rs->state[in]; a.cin_count = cnt_in;
  a.cout_state = rs->state[outl]; a.cout_count = cnt_out;
  a.acc = rs->acc; a.acc_stride = rs->acc_stride; a.turn_steps = report + rs->n_roots; a.queue = heads; a.err = errw;
  a.n_fresh = fresh ? rs->n_roots * rs->reps : 0u; a.reps = rs->reps; a.seg = rs->cap / oak::ROOT_SHARDS;
  a.fper = (a.n_fresh + oak::ROOT_SHARDS - 1) / oak::ROOT_SHARDS;
  a.slice_mask = rs->slice ? rs->slice - 1 : 0xFFFFFFFFu; // (0xFFFFFFFF: steps & mask is never 0 after a step -- no slicing)
  a.max_steps = rs->max_steps;
  // a persistent grid of the resident waves: the number of carried playouts is only known on the device
  const uint64_t upper = (uint64_t)a.n_fresh + rs->cap;
  const uint32_t resident = (uint32_t)c->n_cu * 4u * 4u;
  const uint32_t waves = (uint32_t)std::min<uint64_t>(resident, (upper + 63) / 64);
  hipLaunchKernelGGL(oak::k_root_step<4>, dim3(waves), dim3(64), oak::ROOT_STEP_LDS_BYTES, c->stream, a);
  HIPCHK(hipGetLastError());
  hipLaunchKernelGGL(oak::k_root_step_report, dim3((rs->n_roots + 255) / 256), dim3(256), 0, c->stream, rs->acc, rs->acc_stride, rs->n_roots, cnt_out, errw,
                     a.seg, report);
  HIPCHK(hipGetLastError());
  rs->cur = outl;
  return 0;
}

int oakgpu_rollout_draws_dev(oakgpu_ctx *c, const uint8_t *battles, uint32_t battle_stride, const uint8_t *durations,
                             uint32_t durations_stride, const uint8_t *results_in, uint32_t results_stride, const uint64_t *draws,
                             uint32_t n_draws, const uint32_t *offsets, uint32_t n, uint32_t max_steps, int prep,
                             uint8_t *results_out, uint32_t *steps_out, float *values_out, uint8_t *battles_out,
                             uint8_t *durations_out, uint32_t *used_out) {
  if (!c) return bad("null ctx");
  if (n == 0) return 0;
  if (!battles || !durations || !results_in || !draws) return bad("oakgpu_rollout_draws_dev: null required pointer");
  if ((battle_stride != 0 && battle_stride != 384) || (durations_stride != 0 && durations_stride != 8) || results_stride > 1)
    return bad("oakgpu_rollout_draws_dev: strides must be 0 (one shared root) or the element size");
  HIPCHK(hipSetDevice(c->device));
  const oak::DrawArgs a{battles, durations, results_in, battle_stride, durations_stride, results_stride, draws, n_draws, offsets, n,
                        max_steps, prep, results_out, steps_out, values_out, battles_out, durations_out, used_out};
  hipLaunchKernelGGL(oak::k_rollout_draws<64>, dim3((n + 63) / 64), dim3(64), 24 * 64 * 4 + oak::TABLE_LDS_PAD, c->stream, a);
  HIPCHK(hipGetLastError());
  return 0;
}

// std::mt19937 + the reference's uniform_64 (util/random.h:37: std::uniform_int_distribution<uint64_t> over the full
// range = two 32-bit outputs, high word first -- pinned by tests/golden/rng_known_answers.json)
int oakgpu_mt19937_fill(uint32_t seed, uint64_t skip, uint64_t *out, size_t count) {
  if (!out && count) return bad("oakgpu_mt19937_fill: null out");
  std::mt19937 g{seed};
  g.discard(2 * skip);
  for (size_t i = 0; i < count; ++i) {
    const uint64_t hi = g(), lo = g();
    out[i] = (hi << 32) | lo;
  }
  return 0;
}

// n playouts from ONE root driven by ONE sequential device generator (benchmark.cc:23-31 + mcts.h:250-263,448-496):
// playout i consumes the generator's output right where playout i-1 stopped, so its start offset depends on every
// earlier playout's length.  Resolved on the device without serialising the playouts: pass 1 plays a playout from
// EVERY possible start offset (one lane per draw of the stream; only the number of draws it consumes is kept), the host
// then follows the chain offset[i+1] = offset[i] + used[offset[i]], and pass 2 plays the n real playouts from their
// resolved offsets with full outputs.
int oakgpu_rollout_shared_device(oakgpu_ctx *c, const uint8_t *battle, const uint8_t *durations, uint8_t result,
                                 const uint64_t *draws, uint32_t n_draws, uint32_t n, uint32_t max_steps, int prep,
                                 uint8_t *results_out, uint32_t *steps_out, float *values_out, uint8_t *battles_out,
                                 uint8_t *durations_out, uint32_t *offsets_out, uint64_t *draws_consumed) {
  if (!c) return bad("null ctx");
  if (!battle || !durations || (!draws && n_draws)) return bad("oakgpu_rollout_shared_device: null required pointer");
  if (draws_consumed) *draws_consumed = 0;
  if (n == 0) return 0;
  HIPCHK(hipSetDevice(c->device));
  OakHostCall hc(c);
  DevBuf root, dr, used, off, ro, st, va, bo, dd;
  hipError_t e = hipSuccess;
  int rc = 0;
  std::vector<uint32_t> h_used(n_draws), h_off(n);
  uint8_t h_root[384 + 8 + 8] = {};
  memcpy(h_root, battle, 384);
  memcpy(h_root + 384, durations, 8);
  h_root[392] = result;
  auto ok = [&]() { return e == hipSuccess && rc == 0; };
  e = root.alloc(sizeof h_root, c);
  if (ok()) e = hipMemcpyAsync(root.p, h_root, sizeof h_root, hipMemcpyHostToDevice, c->stream);
  if (ok()) e = dr.alloc((size_t)n_draws * 8 + 8, c);
  if (ok() && n_draws) e = hipMemcpyAsync(dr.p, draws, (size_t)n_draws * 8, hipMemcpyHostToDevice, c->stream);
  if (ok()) e = used.alloc((size_t)n_draws * 4 + 4, c);
  const uint8_t *rb = (const uint8_t *)root.p;
  // pass 1: draws consumed by a playout started at every offset
  if (ok() && n_draws)
    rc = oakgpu_rollout_draws_dev(c, rb, 0, rb + 384, 0, rb + 392, 0, (const uint64_t *)dr.p, n_draws, nullptr, n_draws, max_steps, prep,
                                  nullptr, nullptr, nullptr, nullptr, nullptr, (uint32_t *)used.p);
  if (ok() && n_draws) e = hipMemcpyAsync(h_used.data(), used.p, (size_t)n_draws * 4, hipMemcpyDeviceToHost, c->stream);
  if (ok()) e = hipStreamSynchronize(c->stream);
  bool short_stream = false;
  if (ok()) { // the chain
    uint64_t pos = 0;
    for (uint32_t i = 0; i < n; ++i) {
      if (pos >= n_draws || h_used[pos] == 0xFFFFFFFFu) { short_stream = true; break; }
      h_off[i] = (uint32_t)pos;
      pos += h_used[pos];
    }
    if (!short_stream && draws_consumed) *draws_consumed = pos;
  }
  // pass 2: the n playouts themselves
  if (ok() && !short_stream) {
    e = off.alloc((size_t)n * 4, c);
    if (ok()) e = hipMemcpyAsync(off.p, h_off.data(), (size_t)n * 4, hipMemcpyHostToDevice, c->stream);
    if (ok()) e = ro.alloc(n, c);
    if (ok()) e = st.alloc((size_t)n * 4, c);
    if (ok()) e = va.alloc((size_t)n * 4, c);
    if (ok() && battles_out) e = bo.alloc((size_t)n * 384, c);
    if (ok() && durations_out) e = dd.alloc((size_t)n * 8, c);
    if (ok())
      rc = oakgpu_rollout_draws_dev(c, rb, 0, rb + 384, 0, rb + 392, 0, (const uint64_t *)dr.p, n_draws, (const uint32_t *)off.p, n, max_steps,
                                    prep, (uint8_t *)ro.p, (uint32_t *)st.p, (float *)va.p, (uint8_t *)bo.p, (uint8_t *)dd.p, nullptr);
    auto down = [&](void *host, DevBuf &buf) {
      if (ok() && host && buf.p) e = hipMemcpyAsync(host, buf.p, buf.bytes, hipMemcpyDeviceToHost, c->stream);
    };
    down(results_out, ro); down(steps_out, st); down(values_out, va); down(battles_out, bo); down(durations_out, dd);
    if (ok() && offsets_out) memcpy(offsets_out, h_off.data(), (size_t)n * 4);
  }
  const hipError_t se = hipStreamSynchronize(c->stream);
  if (e != hipSuccess) return fail(e, "oakgpu_rollout_shared_device");
  if (rc) return rc;
  if (se != hipSuccess) return fail(se, "hipStreamSynchronize");
  if (short_stream) return bad("oakgpu_rollout_shared_device: the draw stream is too short for n playouts");
  return 0;
}

int oakgpu_update_dev(oakgpu_ctx *c, uint8_t *battles, const uint8_t *c1, const uint8_t *c2, uint8_t *durations,
                      uint8_t *actions, const uint8_t *overrides, uint32_t n, uint8_t *results) {
  if (!c) return bad("null ctx");
  if (n == 0) return 0;
  if (!battles || !c1 || !c2 || !durations || !results) return bad("oakgpu_update_dev: null required pointer");
  HIPCHK(hipSetDevice(c->device));
  hipLaunchKernelGGL(oak::k_update, dim3(grid_for(n)), dim3(oak::BLOCK), oak::ENGINE_LDS_BYTES, c->stream, battles, c1, c2,
                     durations, actions, overrides, n, results);
  HIPCHK(hipGetLastError());
  return 0;
}

int oakgpu_choices_dev(oakgpu_ctx *c, const uint8_t *battles, const uint8_t *results, int player, uint8_t *out,
                       uint8_t *counts, uint32_t n) {
  if (!c) return bad("null ctx");
  if (n == 0) return 0;
  if (!battles || !results || !out || !counts || player < 0 || player > 1) return bad("oakgpu_choices_dev: bad argument");
  HIPCHK(hipSetDevice(c->device));
  hipLaunchKernelGGL(oak::k_choices, dim3(grid_for(n)), dim3(oak::BLOCK), oak::ENGINE_LDS_BYTES, c->stream, battles, results,
                     player, out, counts, n);
  HIPCHK(hipGetLastError());
  return 0;
}

int oakgpu_tree_step_dev(oakgpu_ctx *c, uint8_t *battles, uint8_t *durations, uint8_t *results, const uint8_t *c1,
                         const uint8_t *c2, uint32_t n, uint32_t rolls, uint8_t *actions, uint8_t *p1_choices,
                         uint8_t *p1_counts, uint8_t *p2_choices, uint8_t *p2_counts) {
  if (!c) return bad("null ctx");
  if (n == 0) return 0;
  if (!battles || !durations || !results || !c1 || !c2 || !actions || !p1_choices || !p1_counts || !p2_choices || !p2_counts)
    return bad("oakgpu_tree_step_dev: null required pointer");
  if (!(rolls == 1 || rolls == 2 || rolls == 3 || rolls == 20 || rolls == 39)) return bad("oakgpu_tree_step_dev: rolls must be 1, 2, 3, 20 or 39");
  HIPCHK(hipSetDevice(c->device));
  static const bool lds_engine = getenv("OAKGPU_TREE_STEP") && strcmp(getenv("OAKGPU_TREE_STEP"), "lds") == 0; // (A/B: the LDS-resident engine's kernel)
  if (!lds_engine && ((uintptr_t)battles & 15) == 0 && ((uintptr_t)durations & 3) == 0 && ((uintptr_t)actions & 7) == 0) {
    const oak::TreeStepArgs ta{battles, durations, results, c1, c2, actions, p1_choices, p1_counts, p2_choices, p2_counts, n, rolls};
    hipLaunchKernelGGL(oak::k_tree_step_staged, dim3((n + 63) / 64), dim3(64), oak::STAGED_LDS_BYTES, c->stream, ta);
  } else
    hipLaunchKernelGGL(oak::k_tree_step, dim3(grid_for(n)), dim3(oak::BLOCK), oak::ENGINE_LDS_BYTES, c->stream, battles, durations,
                       results, c1, c2, n, rolls, actions, p1_choices, p1_counts, p2_choices, p2_counts);
  HIPCHK(hipGetLastError());
  return 0;
}

int oakgpu_poke_engine_eval_dev(oakgpu_ctx *c, const uint8_t *battles, uint32_t n, float root_score, float *values, float *scores) {
  if (!c) return bad("null ctx");
  if (n == 0) return 0;
  if (!battles || (!values && !scores)) return bad("oakgpu_poke_engine_eval_dev: null required pointer");
  HIPCHK(hipSetDevice(c->device));
  hipLaunchKernelGGL(oak::k_poke_engine, dim3((n + 255) / 256), dim3(256), 0, c->stream, battles, n, root_score, values, scores);
  HIPCHK(hipGetLastError());
  return 0;
}

int oakgpu_poke_engine_eval(oakgpu_ctx *c, const uint8_t *battles, uint32_t n, float root_score, float *values, float *scores) {
  if (!c) return bad("null ctx");
  if (n == 0) return 0;
  if (!battles || (!values && !scores)) return bad("oakgpu_poke_engine_eval: null required pointer");
  HIPCHK(hipSetDevice(c->device));
  OakHostCall hc(c);
  DevBuf db, dv;
  HIPCHK(db.alloc((size_t)n * 384, c));
  HIPCHK(dv.alloc((size_t)n * 8, c));
  float *fv = (float *)dv.p, *fs = fv + n;
  HIPCHK(hipMemcpyAsync(db.p, battles, (size_t)n * 384, hipMemcpyHostToDevice, c->stream));
  if (int rc = oakgpu_poke_engine_eval_dev(c, (const uint8_t *)db.p, n, root_score, fv, fs)) return rc;
  if (values) HIPCHK(hipMemcpyAsync(values, fv, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
  if (scores) HIPCHK(hipMemcpyAsync(scores, fs, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return 0;
}

int oakgpu_init_battles_dev(oakgpu_ctx *c, const uint8_t *teams, const uint64_t *seeds, uint32_t n, int first_update,
                            uint8_t *battles, uint8_t *durations, uint8_t *results) {
  if (!c) return bad("null ctx");
  if (n == 0) return 0;
  if (!teams || !seeds || !battles) return bad("oakgpu_init_battles_dev: null required pointer");
  HIPCHK(hipSetDevice(c->device));
  hipLaunchKernelGGL(oak::k_init, dim3(grid_for(n)), dim3(oak::BLOCK), oak::ENGINE_LDS_BYTES, c->stream, teams, seeds, n,
                     first_update, battles, durations, results);
  HIPCHK(hipGetLastError());
  return 0;
}

int oakgpu_set_ou_pools(oakgpu_ctx *c, const uint8_t *legal, int n_legal, const uint8_t *pools, const uint8_t *sizes) {
  if (!c || !legal || !pools || !sizes || n_legal > 152) return bad("oakgpu_set_ou_pools: bad argument");
  // k_random_ou draws 6 DISTINCT species per side and min(4, pool) distinct moves per species by rejection: with fewer
  // than 6 distinct legal species, or a pool holding repeated / too few move ids, its loops would never terminate
  if (n_legal < 6) return bad("oakgpu_set_ou_pools: at least 6 legal species are required");
  {
    bool seen[256] = {};
    for (int i = 0; i < n_legal; ++i) {
      const uint8_t sp = legal[i];
      if (sp == 0 || sp >= 152 || seen[sp]) return bad("oakgpu_set_ou_pools: legal species must be distinct ids in 1..151");
      seen[sp] = true;
      const uint32_t psz = sizes[sp];
      if (psz == 0 || psz > 48) return bad("oakgpu_set_ou_pools: every legal species needs a move pool of 1..48 moves");
      bool mseen[256] = {};
      for (uint32_t k = 0; k < psz; ++k) {
        const uint8_t mv = pools[(size_t)sp * 48 + k];
        if (mv == 0 || mv > 165 || mseen[mv]) return bad("oakgpu_set_ou_pools: move pools must hold distinct move ids in 1..165");
        mseen[mv] = true;
      }
    }
  }
  HIPCHK(hipSetDevice(c->device));
  if (!c->d_legal) {
    HIPCHK(hipMalloc((void **)&c->d_legal, 152));
    HIPCHK(hipMalloc((void **)&c->d_pools, 152 * 48));
    HIPCHK(hipMalloc((void **)&c->d_sizes, 152));
  }
  HIPCHK(hipMemcpy(c->d_legal, legal, (size_t)n_legal, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(c->d_pools, pools, 152 * 48, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(c->d_sizes, sizes, 152, hipMemcpyHostToDevice));
  c->n_legal = n_legal;
  return 0;
}

int oakgpu_random_ou_battles_dev(oakgpu_ctx *c, uint64_t seed0, uint32_t n, uint8_t *battles, uint8_t *durations,
                                 uint8_t *prng_state, uint8_t *results) {
  if (!c) return bad("null ctx");
  if (!c->d_legal) return bad("oakgpu_random_ou_battles_dev: call oakgpu_set_ou_pools first");
  if (n == 0) return 0;
  if (!battles || !durations || !prng_state || !results) return bad("oakgpu_random_ou_battles_dev: null pointer");
  HIPCHK(hipSetDevice(c->device));
  hipLaunchKernelGGL(oak::k_random_ou, dim3(grid_for(n)), dim3(oak::BLOCK), oak::ENGINE_LDS_BYTES, c->stream, seed0, n,
                     c->d_legal, c->n_legal, c->d_pools, c->d_sizes, battles, durations, prng_state, results);
  HIPCHK(hipGetLastError());
  return 0;
}

// ---- host-buffer conveniences (PCIe-inclusive; never the benchmarked path) ------------------
#define UP(buf, host, nbytes) do { HIPCHK((buf).alloc(nbytes, c)); if (host) HIPCHK(hipMemcpyAsync((buf).p, host, nbytes, hipMemcpyHostToDevice, c->stream)); } while (0)
#define DOWN(host, buf) do { if (host) HIPCHK(hipMemcpyAsync(host, (buf).p, (buf).bytes, hipMemcpyDeviceToHost, c->stream)); } while (0)

int oakgpu_rollout_group(oakgpu_ctx *c, const oakgpu_rollout_batch *batches, uint32_t count, uint32_t max_steps, int prep) {
  if (!c) return bad("null ctx");
  if (count == 0) return 0;
  if (!batches) return bad("oakgpu_rollout_group: null batches");
  if (count > (uint32_t)oak::MAX_GROUP) return bad("oakgpu_rollout_group: more than 64 batches in one group");
  for (uint32_t i = 0; i < count; ++i) {
    const oakgpu_rollout_batch &h = batches[i];
    if (h.n && (!h.battles || !h.durations || !h.results_in || !h.prng_state || !h.results_out || !h.steps_out || !h.values_out))
      return bad("oakgpu_rollout_group: null required pointer");
  }
  HIPCHK(hipSetDevice(c->device));
  OakHostCall hc(c);
  struct Bufs { DevBuf b, d, ri, pr, ro, st, va, bo, dd; };
  std::vector<Bufs> bufs(count);
  std::vector<oakgpu_rollout_batch> dev(count);
  int rc = 0;
  hipError_t e = hipSuccess;
  auto up = [&](DevBuf &buf, const void *host, size_t nbytes) {
    if (e != hipSuccess) return;
    e = buf.alloc(nbytes, c);
    if (e == hipSuccess && host) e = hipMemcpyAsync(buf.p, host, nbytes, hipMemcpyHostToDevice, c->stream);
  };
  auto down = [&](void *host, DevBuf &buf) {
    if (e == hipSuccess && host && buf.p) e = hipMemcpyAsync(host, buf.p, buf.bytes, hipMemcpyDeviceToHost, c->stream);
  };
  for (uint32_t i = 0; i < count; ++i) {
    const oakgpu_rollout_batch &h = batches[i];
    Bufs &B = bufs[i];
    const size_t n = h.n;
    dev[i] = oakgpu_rollout_batch{};
    if (n == 0) continue;
    up(B.b, h.battles, n * 384); up(B.d, h.durations, n * 8); up(B.ri, h.results_in, n); up(B.pr, h.prng_state, n * 8);
    up(B.ro, nullptr, n); up(B.st, nullptr, n * 4); up(B.va, nullptr, n * 4);
    if (h.battles_out) up(B.bo, nullptr, n * 384);
    if (h.durations_out) up(B.dd, nullptr, n * 8);
    dev[i] = oakgpu_rollout_batch{(uint8_t *)B.b.p, (uint8_t *)B.d.p, (uint8_t *)B.ri.p, (uint8_t *)B.pr.p, h.n, (uint8_t *)B.ro.p,
                                  (uint32_t *)B.st.p, (float *)B.va.p, (uint8_t *)B.bo.p, (uint8_t *)B.dd.p};
  }
  if (e == hipSuccess) rc = oakgpu_rollout_group_dev(c, dev.data(), count, max_steps, prep);
  if (e == hipSuccess && !rc)
    for (uint32_t i = 0; i < count; ++i) {
      const oakgpu_rollout_batch &h = batches[i];
      Bufs &B = bufs[i];
      down(h.results_out, B.ro); down(h.steps_out, B.st); down(h.values_out, B.va); down(h.prng_state, B.pr);
      down(h.battles_out, B.bo); down(h.durations_out, B.dd);
    }
  const hipError_t se = hipStreamSynchronize(c->stream);
  if (e != hipSuccess) return fail(e, "oakgpu_rollout_group");
  if (rc) return rc;
  if (se != hipSuccess) return fail(se, "hipStreamSynchronize");
  return oakgpu_synchronize(c); // (reports a migration wait that ran out)
}

int oakgpu_rollout(oakgpu_ctx *c, const uint8_t *battles, const uint8_t *durations, const uint8_t *results_in,
                   uint8_t *prng_state, uint32_t n, uint32_t max_steps, int prep, uint8_t *results_out, uint32_t *steps_out,
                   float *values_out, uint8_t *battles_out, uint8_t *durations_out) {
  const oakgpu_rollout_batch b{battles, durations, results_in, prng_state, n, results_out, steps_out, values_out, battles_out, durations_out};
  return oakgpu_rollout_group(c, &b, 1, max_steps, prep);
}

int oakgpu_update(oakgpu_ctx *c, uint8_t *battles, const uint8_t *c1, const uint8_t *c2, uint8_t *durations, uint8_t *actions,
                  const uint8_t *overrides, uint32_t n, uint8_t *results) {
  if (!c) return bad("null ctx");
  if (n == 0) return 0;
  if (!battles || !c1 || !c2 || !durations || !results) return bad("oakgpu_update: null required pointer");
  HIPCHK(hipSetDevice(c->device));
  OakHostCall hc(c);
  DevBuf b, a1, a2, d, ac, ov, rs;
  UP(b, battles, (size_t)n * 384);
  UP(a1, c1, n);
  UP(a2, c2, n);
  UP(d, durations, (size_t)n * 8);
  if (actions) UP(ac, (const void *)nullptr, (size_t)n * 16);
  if (overrides) UP(ov, overrides, (size_t)n * 16);
  UP(rs, (const void *)nullptr, n);
  int r = oakgpu_update_dev(c, (uint8_t *)b.p, (uint8_t *)a1.p, (uint8_t *)a2.p, (uint8_t *)d.p, (uint8_t *)ac.p,
                            (uint8_t *)ov.p, n, (uint8_t *)rs.p);
  if (r) return r;
  DOWN(battles, b);
  DOWN(durations, d);
  DOWN(actions, ac);
  DOWN(results, rs);
  HIPCHK(hipStreamSynchronize(c->stream));
  return 0;
}

int oakgpu_choices(oakgpu_ctx *c, const uint8_t *battles, const uint8_t *results, int player, uint8_t *out, uint8_t *counts,
                   uint32_t n) {
  if (!c) return bad("null ctx");
  if (n == 0) return 0;
  if (!battles || !results || !out || !counts) return bad("oakgpu_choices: null required pointer");
  HIPCHK(hipSetDevice(c->device));
  OakHostCall hc(c);
  DevBuf b, rs, o, cn;
  UP(b, battles, (size_t)n * 384);
  UP(rs, results, n);
  UP(o, (const void *)nullptr, (size_t)n * 9);
  UP(cn, (const void *)nullptr, n);
  int r = oakgpu_choices_dev(c, (uint8_t *)b.p, (uint8_t *)rs.p, player, (uint8_t *)o.p, (uint8_t *)cn.p, n);
  if (r) return r;
  DOWN(out, o);
  DOWN(counts, cn);
  HIPCHK(hipStreamSynchronize(c->stream));
  return 0;
}

int oakgpu_init_battles(oakgpu_ctx *c, const uint8_t *teams, const uint64_t *seeds, uint32_t n, int first_update,
                        uint8_t *battles, uint8_t *durations, uint8_t *results) {
  if (!c) return bad("null ctx");
  if (n == 0) return 0;
  if (!teams || !seeds || !battles) return bad("oakgpu_init_battles: null required pointer");
  HIPCHK(hipSetDevice(c->device));
  OakHostCall hc(c);
  DevBuf t, s, b, d, rs;
  UP(t, teams, (size_t)n * 60);
  UP(s, seeds, (size_t)n * 8);
  UP(b, (const void *)nullptr, (size_t)n * 384);
  UP(d, (const void *)nullptr, (size_t)n * 8);
  UP(rs, (const void *)nullptr, n);
  int r = oakgpu_init_battles_dev(c, (uint8_t *)t.p, (uint64_t *)s.p, n, first_update, (uint8_t *)b.p, (uint8_t *)d.p,
                                  (uint8_t *)rs.p);
  if (r) return r;
  DOWN(battles, b);
  DOWN(durations, d);
  DOWN(results, rs);
  HIPCHK(hipStreamSynchronize(c->stream));
  return 0;
}

} // extern "C"
