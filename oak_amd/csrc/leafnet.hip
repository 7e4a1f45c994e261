// oak_amd/csrc/leafnet.hip -- fp32 leaf evaluator on gfx950 (K2 + K3) and its C ABI.
//
// Replaces NN::Battle::NetworkImpl::value_inference (cpp/include/nn/battle/network.h:72-79):
//   K2  k_embed_both (k_embed_prows + k_embed_arows; k_embed_lds for embedding widths those do not take):
//                   Encode::Battle::{Pokemon,ActivePokemon}::write (encode/battle/battle.h:208-214, 544-551) fused with
//                   EmbeddingNet::propagate (nn/ffn.h:47-51, affine.h:87-103) and write_battle_embedding
//                   (network.h:131-175): first-layer weights resident in LDS, dense parts on fp32 MFMA.  The reference's
//                   per-battle embedding cache (nn/battle/cache.h) is oakgpu_leaf_eval_cached_dev (k_party_tags).
//   K3  k_mainnet_wave : MainNet::propagate value path (nn/battle/main-net.h:57-64) + sigmoid, the three dense layers on
//                   v_mfma_f32_32x32x2_f32 (exact fp32); k_policy: the policy heads (main-net.h:67-107).
// Parameter file reader: nn/affine.h:35-70, network.h:52-70, main-net.h:36-55, search.cc:127-131.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <cmath>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/oakgpu.h"
#include "oakgpu_internal.h"

namespace oak {

struct NetDev {
  // embedding nets: W0t [in][hidden <= 128] (transposed: one row per input feature), b0, b1
  const float *p_w0t, *p_b0, *p_b1;
  const float *a_w0t, *a_b0, *a_b1;
  const float *p_w1, *a_w1; // second layers in file layout [out][hidden] (k_embed_lds)
  const float *a_w0d, *a_img; // k_embed_arows: padded W0^T (move rows), and the image of its LDS weights (arows_image)
  const float *p_img;         // k_embed_prows: the image of its LDS weights (prows_image)
  int p_hidden, p_out, a_hidden, a_out;
  int side_dim, emb_dim;
  int activation; // 1 relu, 2 clamp
  // main net (value path), rows padded to multiples of 32 with zeros
  const float *b0, *b1, *b2, *w3;
  const float *w0g, *w1g, *w2g; // the three weight matrices in k_mainnet_wave's MFMA-fragment order (frag_order_wave)
  float b3;
  int H, VH; // padded dims
  // k_mainnet_split: the three matrices as ONE stream of bf16 triples in its MFMA-fragment order (split_stream), the k-steps
  // of fc0 (ws_T0, a multiple of 4), the block count both widths are padded to (ws_NB: 1, 2, 4, 8)
  const uint16_t *ws;
  int ws_T0, ws_NB;
  // k_mainnet_pair: the same stream as scaled fp16 pairs (pair_stream), and 1 / the scale of each of the three layers
  const uint16_t *wp;
  const float *wp_inv;        // 1 / the scale of every output row: [fc0: H | fc1: H | value_fc2: VH] (padded rows: 0)
  // policy heads (main-net.h:67-107): fc2 [PHp][H] (rows padded to 32), fc3 [315][PHp] (+ biases)
  const float *q1a, *q1a_b, *q1b, *q1b_b, *q2a, *q2a_b, *q2b, *q2b_b;
  const float *q1a_f, *q2a_f; // fc2 of the two heads as k_policy_rows' A operand (policy_frag_order)
  const float *q1a_t, *q2a_t; // ... and as bf16 triples (policy_triple_order), or null: a weight above 2^20 (see split_safe)
  const float *q1b_img, *q2b_img; // fc3's rows + biases exactly as k_policy_rows lays them out in LDS (policy_rows_image), or null: PH > 64
  int PH; // padded policy hidden width
};

// bias + activation of 16 accumulator registers (or activation alone), the activation chosen by a wave-uniform branch: computed per
// value as `activation == 1 ? relu : clamp` it costs max + min + select for each of them
typedef __attribute__((ext_vector_type(16))) float act_f32x16;
template <int NB_>
__device__ __forceinline__ void act_blocks(act_f32x16 (&v)[NB_], int activation) {
  if (activation == 1) {
#pragma unroll
    for (int b = 0; b < NB_; ++b)
#pragma unroll
      for (int q = 0; q < 16; ++q) v[b][q] = fmaxf(v[b][q], 0.0f);
  } else {
#pragma unroll
    for (int b = 0; b < NB_; ++b)
#pragma unroll
      for (int q = 0; q < 16; ++q) v[b][q] = fminf(fmaxf(v[b][q], 0.0f), 1.0f);
  }
}
__device__ __forceinline__ float act_fn(float x, int activation) {
  x = fmaxf(x, 0.0f);
  return activation == 2 ? fminf(x, 1.0f) : x;
}

// ---- K2 ---------------------------------------------------------------------------------------

__device__ __forceinline__ uint32_t status_index(uint32_t status, uint32_t sleeps) { // battle.h:103-123
  if (!(status & 7)) return (uint32_t)__builtin_ctz(status) - 3;
  if (!(status & 0x80)) return 3 + sleeps;
  return 14 - (status & 7);
}

// Feature `j` (0..11) of Encode::Battle::Pokemon (battle.h:16-214) from the 6 dwords of a party slot:
// 0-4 stats, 5-8 move slots, 9 status, 10-11 types.  Returns false when the feature is absent.
__device__ __forceinline__ bool pokemon_feature(uint32_t j, uint32_t pk0, uint32_t pk1, uint32_t pk2, uint32_t pk3, uint32_t pk4,
                                                uint32_t pk5, uint32_t sleep, uint32_t &idx, float &val) {
  if (j < 5) {
    const uint32_t raw = j == 0 ? pk0 & 0xFFFF : j == 1 ? pk0 >> 16 : j == 2 ? pk1 & 0xFFFF : j == 3 ? pk1 >> 16 : pk2 & 0xFFFF;
    idx = j;
    val = (float)raw / (j == 0 ? 703.0f : 999.0f);
    return true;
  }
  if (j < 9) {
    const uint32_t ms = j == 5 ? pk2 >> 16 : j == 6 ? pk3 & 0xFFFF : j == 7 ? pk3 >> 16 : pk4 & 0xFFFF;
    const uint32_t id = ms & 0xFF, pp = ms >> 8;
    idx = 5 + id - 1;
    val = 1.0f;
    return id != 0 && id != 165 && pp != 0;
  }
  if (j == 9) {
    const uint32_t st = pk5 & 0xFF;
    idx = 169 + (st ? status_index(st, sleep) : 0);
    val = 1.0f;
    return st != 0;
  }
  const uint32_t ty = (pk5 >> 16) & 0xFF, t1 = ty & 15, t2 = ty >> 4;
  idx = 183 + (j == 10 ? t1 : t2);
  val = 1.0f;
  return j == 10 || t2 != t1;
}

// Feature `j` (0..39) of Encode::Battle::Active (battle.h:229-489, sparse form): 0-4 stats, 5-6 types,
// 7-12 boosts, 13-31 volatiles, 32-35 move slots, 36-39 durations.
__device__ __forceinline__ bool active_feature(uint32_t j, uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3, uint32_t vlo,
                                               uint32_t vhi, uint32_t m01, uint32_t m23, uint32_t dur, uint32_t &idx, float &val) {
  if (j < 5) {
    const uint32_t raw = j == 0 ? a0 & 0xFFFF : j == 1 ? a0 >> 16 : j == 2 ? a1 & 0xFFFF : j == 3 ? a1 >> 16 : a2 & 0xFFFF;
    idx = j;
    val = (float)raw / (j == 0 ? 703.0f : 999.0f);
    return true;
  }
  if (j < 7) {
    const uint32_t ty = a2 >> 24, t1 = ty & 15, t2 = ty >> 4;
    idx = 5 + (j == 5 ? t1 : t2);
    val = 1.0f;
    return j == 5 || t2 != t1;
  }
  if (j < 13) { // boosts (battle.h:271-285): stage ratio * 1/4 (accuracy / evasion: 1/3)
    const uint32_t i = j - 7;
    const int st = (int)((((a3 >> (4 * i)) & 15) ^ 8) - 8);
    // ratios of libpkmn/data/boosts.h:10-24 as float(num) / den
    const float num = st == -6 ? 25.f : st == -5 ? 28.f : st == -4 ? 33.f : st == -3 ? 40.f : st == -2 ? 50.f : st == -1 ? 66.f
                      : st == 0 ? 1.f : st == 1 ? 15.f : st == 2 ? 2.f : st == 3 ? 25.f : st == 4 ? 3.f : st == 5 ? 35.f : 4.f;
    const float den = st < 0 ? 100.f : (st == 1 || st == 3 || st == 5) ? 10.f : 1.f;
    idx = 20 + i;
    val = (num / den) * (i < 4 ? 0.25f : (float)(1 / 3.0));
    return true;
  }
  if (j < 32) { // volatiles (battle.h:322-352)
    const uint32_t i = j - 13;
    idx = 26 + i;
    if (i < 16) {
      const uint32_t bit = i < 2 ? i : i + 2; // bide, thrashing, then charging(4)..transform(17)
      val = 1.0f;
      return (vlo >> bit) & 1;
    }
    const uint32_t state = (vlo >> 24) | ((vhi & 0xFF) << 8), sub = (vhi >> 8) & 0xFF, tox = vhi >> 27;
    if (i == 16) { val = (float)state / 65535.0f; return state != 0; }
    if (i == 17) { val = (float)sub / 177.0f; return sub != 0; }
    val = (float)tox / 16.0f;
    return tox != 0;
  }
  if (j < 36) {
    const uint32_t i = j - 32;
    const uint32_t ms = (i < 2 ? m01 >> (16 * i) : m23 >> (16 * (i - 2))) & 0xFFFF;
    const uint32_t id = ms & 0xFF, pp = ms >> 8;
    idx = 45 + id - 1;
    val = 1.0f;
    return id != 0 && id != 165 && pp != 0;
  }
  const uint32_t i = j - 36;
  const uint32_t v = i == 0 ? (dur >> 18) & 7 : i == 1 ? (dur >> 21) & 15 : i == 2 ? (dur >> 25) & 7 : (dur >> 28) & 7;
  idx = (i == 0 ? 209 : i == 1 ? 214 : i == 2 ? 222 : 225) + v - 1;
  val = 1.0f;
  return v != 0;
}

using f32x16 = __attribute__((ext_vector_type(16))) float;

// ---- K2: the embedding passes.  Items = party slots (10 per leaf) or actives (2 per leaf); the dense second layer is ~80% of
// the embedding FLOPs and runs on the matrix pipe in every form below.
constexpr int ET = 64;          // items per tile (k_embed_lds)
constexpr int EHP = 129;        // padded LDS row (odd stride: conflict-free column reads)
struct EmbedTileArgs {
  NetDev net;
  const uint8_t *battles;
  const uint8_t *durations;
  uint32_t n;
  float *emb;
  int kind; // 0: party slots (10 per leaf), 1: actives (2 per leaf)
  // LIST mode of the party-slot pass (oakgpu_leaf_eval_cached*): only the slots k_party_tags found changed
  const struct PartyWork *work;
  const uint32_t *work_count;
};
struct PartyWork { uint32_t item, pk[6], sleep; }; // item = leaf * 10 + q; the slot's stored Pokemon; its public sleep turns

// ---- K2, LDS-resident first layer (the default).  k_embed_tile's sparse first layer gathers ~210 weight rows of
// 512 B per leaf from L2 (7 GB per 65,536-leaf batch) one item after the other, and that gather latency is
// what it spends its time on.  Here one 512-thread workgroup per CU keeps the first-layer weights in LDS:
//   party-slot kind : all 198 rows of W0^T (99 KB);
//   active kind     : the 99 rows that are NOT move one-hots (stats, types, boosts, volatiles, durations and
//                     the stored Pokemon's stats / status / types: 50 KB) -- they carry ~37 of an item's ~45
//                     non-zeros; the <= 8 move rows still come from L2, prefetched for four items at once so
//                     that their latency hides behind the LDS part.
// The dense second layer stays on fp32 MFMA, with its B operand (W1, 64 values per lane) held in registers
// instead of LDS.  Items per tile, tile layout and outputs are k_embed_tile's.
#ifdef OAKGPU_LEAF_PROFILE // tools/leaf_phase_profile.py: wave 0's cycles per phase of k_embed_lds, summed over tiles
static __device__ unsigned long long g_leaf_prof[16];
// accumulate in registers (a global atomic per mark would sit in vmcnt and distort the very waits being measured)
#define EL_T0() long long el_t = clock64(); unsigned long long el_acc[8] = {}
#define EL_MARK(id) do { const long long _t = clock64(); el_acc[id] += (unsigned long long)(_t - el_t); el_t = _t; } while (0)
#define EL_FLUSH() do { if (threadIdx.x == 0) for (int _i = 0; _i < 8; ++_i) atomicAdd(&g_leaf_prof[_i], el_acc[_i]); } while (0)
#define MN_T0() long long mn_t = clock64()
#define MN_MARK(id) do { const long long _t = clock64(); if (threadIdx.x == 0) atomicAdd(&g_leaf_prof[id], (unsigned long long)(_t - mn_t)); mn_t = _t; } while (0)
// k_mainnet_split counts its vector-memory operations by hand: its marks are kept in registers and added once per tile
#define MS_T0() long long ms_t = clock64(); long long ms_d[6] = {0, 0, 0, 0, 0, 0}
#define MS_MARK(i) do { const long long _t = clock64(); ms_d[i] = _t - ms_t; ms_t = _t; } while (0)
#define MS_FLUSH() do { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); if (threadIdx.x == 0) for (int i_ = 0; i_ < 6; ++i_) atomicAdd(&g_leaf_prof[10 + i_], (unsigned long long)ms_d[i_]); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); } while (0)
#else
#define MN_T0()
#define MN_MARK(id)
#define MS_T0()
#define MS_MARK(i)
#define MS_FLUSH()
#define EL_MARK(id)
#define EL_T0()
#define EL_FLUSH()
#endif
constexpr int EL_BLOCK = 512;
constexpr int EL_WAVES = EL_BLOCK / 64;
constexpr int EL_ITEMS = ET / EL_WAVES; // 8 items per wave
constexpr int EL_A_ROWS = 99;
template <bool ACT> struct ELayout {
  static constexpr int NROWS = ACT ? EL_A_ROWS : 198;
  static constexpr int LCAP = ACT ? 48 : 16; // LDS-row features per item (<= 44 / 12), padded to a multiple of 4
  static constexpr int NLEAF = ACT ? 34 : 9; // leaves a tile of 64 items can touch
  static constexpr size_t BYTES = (size_t)(NROWS * 128 + ET * EHP + 2 * ET * LCAP + ET * 8 + 3 * ET + NLEAF * 98) * 4;
};
// LDS slot of row r of the active net's W0^T (move rows excluded), and its inverse
__device__ __forceinline__ uint32_t active_lds_slot(uint32_t r) { return r < 45 ? r : r < 234 ? r - 164 : r - 328; }
__device__ __forceinline__ uint32_t active_lds_row(uint32_t s) { return s < 45 ? s : s < 70 ? s + 164 : s + 328; }

template <bool ACT, bool LIST = false>
__global__ __launch_bounds__(EL_BLOCK) void k_embed_lds(EmbedTileArgs a) {
  static_assert(!(ACT && LIST), "the work-list form exists for the party-slot pass only");
  extern __shared__ __align__(16) float lds_f[];
  using L = ELayout<ACT>;
  const NetDev &N = a.net;
  const int hidden = ACT ? N.a_hidden : N.p_hidden;
  const int out_dim = ACT ? N.a_out : N.p_out;
  const int NBo = (out_dim + 31) >> 5;
  float *W0s = lds_f;                                   // NROWS x 128 (zero-padded beyond `hidden`)
  float *Hs = W0s + L::NROWS * 128;                     // ET x EHP
  uint32_t *Lidx = (uint32_t *)(Hs + ET * EHP);         // per item: LCAP LDS word offsets of W0s rows
  float *Lval = (float *)(Lidx + ET * L::LCAP);         // ... and their feature values
  uint32_t *Gidx = (uint32_t *)(Lval + ET * L::LCAP);   // per item: 8 global W0^T rows (move one-hots, ACT only)
  uint32_t *meta = Gidx + ET * 8;                       // per item: countL | nG << 8
  uint32_t *dst_off = meta + ET;                        // per item: float offset of its block in emb, or ~0
  float *hp_ratio = (float *)(dst_off + ET);
  uint32_t *Bs = (uint32_t *)(hp_ratio + ET);           // staged battles: 96 dwords + 2 duration dwords per leaf
  const float *w0t = ACT ? N.a_w0t : N.p_w0t;
  const float *W1 = ACT ? N.a_w1 : N.p_w1;
  const float *b0 = ACT ? N.a_b0 : N.p_b0;
  const float *b1 = ACT ? N.a_b1 : N.p_b1;
  for (int i = threadIdx.x; i < L::NROWS * 128; i += EL_BLOCK) {
    const uint32_t sl = (uint32_t)i >> 7, c = (uint32_t)i & 127;
    const uint32_t r = ACT ? active_lds_row(sl) : sl;
    W0s[i] = (int)c < hidden ? w0t[(size_t)r * hidden + c] : 0.0f;
  }
  const uint32_t lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
  // a lane owns the ADJACENT hidden channels 2*lane, 2*lane + 1: one 8-byte LDS / global read per weight row
  const uint32_t c0 = 2 * lane, c1 = 2 * lane + 1;
  const bool on0 = (int)c0 < hidden, on1 = (int)c1 < hidden;
  const bool pair_ok = (hidden & 1) == 0; // rows of W0^T in global memory are 8-byte aligned only for even widths
  const uint32_t cc0 = on0 ? c0 : 0, cc1 = on1 ? c1 : 0;
  const float bias0 = on0 ? b0[c0] : 0.0f, bias1 = on1 ? b0[c1] : 0.0f;
  // party slots: the five stat features are present in EVERY item, so their weight rows (W0^T rows 0..4) stay in
  // registers for the whole kernel instead of being read from LDS 64 times per tile -- 5 of an item's ~11 row reads
  float wst0[5], wst1[5];
#pragma unroll
  for (int k = 0; k < 5; ++k) { wst0[k] = (!ACT && on0) ? w0t[(size_t)k * hidden + c0] : 0.0f; wst1[k] = (!ACT && on1) ? w0t[(size_t)k * hidden + c1] : 0.0f; }
  // MFMA role of this wave: output block (mi, nb) of the 64 x out_pad tile; B fragments stay in registers
  const int mi = wib & 1, nb = wib >> 1, r32 = lane & 31, hh = lane >> 5;
  const bool mfma_wave = nb < NBo;
  float bfrag[64];
  {
    const int o = nb * 32 + r32;
#pragma unroll
    for (int s2 = 0; s2 < 64; ++s2) {
      const int c = 2 * s2 + hh;
      bfrag[s2] = (mfma_wave && o < out_dim && c < hidden) ? W1[(size_t)o * hidden + c] : 0.0f;
    }
  }
  const float obias = (mfma_wave && nb * 32 + r32 < out_dim) ? b1[nb * 32 + r32] : 0.0f;
  const uint32_t per_leaf = ACT ? 2 : 10;
  const uint32_t items = LIST ? *a.work_count : a.n * per_leaf;
  const uint32_t ntiles = (items + ET - 1) / ET;
  // The battles a tile touches are staged in LDS; the NEXT tile's are already on their way in registers while
  // this one is processed (one workgroup per CU: nothing else would hide that latency).
  constexpr uint32_t PF = (L::NLEAF * 98 + EL_BLOCK - 1) / EL_BLOCK; // staged dwords per thread
  uint32_t pf[PF];
  auto tile_leaves = [&](uint32_t tile, uint32_t &first_leaf) {
    first_leaf = (tile * ET) / per_leaf;
    uint32_t last_leaf = (tile * ET + ET - 1) / per_leaf;
    if (last_leaf >= a.n) last_leaf = a.n - 1;
    return last_leaf - first_leaf + 1;
  };
  auto prefetch = [&](uint32_t tile) {
    if constexpr (LIST) { // the tile's 64 work records (8 dwords each) instead of whole battles
      const uint32_t g = tile * ET + (threadIdx.x >> 3);
      pf[0] = (tile < ntiles && g < items) ? ((const uint32_t *)a.work)[(size_t)g * 8 + (threadIdx.x & 7)] : 0;
      return;
    }
    uint32_t first_leaf;
    const uint32_t nl = tile < ntiles ? tile_leaves(tile, first_leaf) : 0;
#pragma unroll
    for (uint32_t u = 0; u < PF; ++u) {
      const uint32_t i = threadIdx.x + u * EL_BLOCK; // slot i of the staged image: leaf i / 98, dword i % 98
      const uint32_t l = i / 98, d = i - l * 98;
      pf[u] = 0;
      if (l < nl) pf[u] = d < 96 ? ((const uint32_t *)a.battles)[(size_t)(first_leaf + l) * 96 + d]
                                 : ((const uint32_t *)a.durations)[(size_t)(first_leaf + l) * 2 + (d - 96)];
    }
  };
  auto stage = [&]() { // prefetched registers -> LDS image of the tile's battles
    if constexpr (LIST) { Bs[threadIdx.x] = pf[0]; return; }
#pragma unroll
    for (uint32_t u = 0; u < PF; ++u) { const uint32_t i = threadIdx.x + u * EL_BLOCK; if (i < (uint32_t)L::NLEAF * 98) Bs[i] = pf[u]; }
  };
  prefetch(blockIdx.x);
  stage();
  prefetch(blockIdx.x + gridDim.x);
  EL_T0();
  for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    __syncthreads(); // this tile's battles are staged, the previous tile's H is consumed (and W0s is staged, first time)
    EL_MARK(0);
    uint32_t first_leaf;
    (void)tile_leaves(tile, first_leaf);
    // ---- phase 1a: sparse feature lists of this wave's 8 items, ONE LANE PER ITEM: lane t < 8 walks its item's 12 / 52
    // candidate features in order and appends the present ones to the item's lists.  The feature index is a
    // compile-time constant in the unrolled walk, so the encoders' branches on it fold away and all eight lanes run one
    // straight-line instruction stream (the earlier lane-per-FEATURE form paid the encoder's whole branch ladder once
    // per item).
    {
      for (uint32_t k = lane; k < (uint32_t)(EL_ITEMS * L::LCAP); k += 64) { // padding entries: row 0 with weight 0
        Lidx[wib * EL_ITEMS * L::LCAP + k] = 0;
        Lval[wib * EL_ITEMS * L::LCAP + k] = 0.0f;
      }
      Gidx[wib * EL_ITEMS * 8 + lane] = 0;
      const bool mine = lane < (uint32_t)EL_ITEMS;
      const uint32_t i = wib * EL_ITEMS + (mine ? lane : 0);
      const uint32_t g = tile * ET + i;
      uint32_t doff = 0xFFFFFFFFu, dead_off = 0xFFFFFFFFu, cntL = 0, cntG = 0;
      if (mine && g < items) {
        const uint32_t *rec = Bs + i * 8; // LIST: this item's work record
        const uint32_t gi = LIST ? rec[0] : g;
        const uint32_t leaf = gi / per_leaf, q = gi - leaf * per_leaf;
        const uint32_t side = ACT ? q : q / 5, slot = ACT ? 0 : 1 + (q - side * 5);
        const uint32_t *lb = Bs + (LIST ? 0 : (leaf - first_leaf) * 98);
        const uint32_t *sb = lb + side * 46;
        const uint32_t dur = LIST ? rec[7] << (3 * slot) : lb[96 + side]; // LIST: only the slot's sleep turns are needed
        const uint32_t o0 = LIST ? 0 : sb[44], o1 = LIST ? 0 : sb[45];
        const uint32_t id = LIST ? 1 : slot < 4 ? (o0 >> (8 * slot)) & 0xFF : (o1 >> (8 * (slot - 4))) & 0xFF;
        const uint32_t dd = leaf * N.emb_dim + side * N.side_dim + (ACT ? 0 : (1 + N.a_out) + (slot - 1) * (1 + N.p_out));
        uint32_t pk0 = 0, pk1 = 0, pk2 = 0, pk3 = 0, pk4 = 0, pk5 = 0, hp = 0;
        if (id != 0) {
          const uint32_t *pk = LIST ? rec + 1 : sb + 6 * (id - 1);
          pk0 = pk[0]; pk1 = pk[1]; pk2 = pk[2]; pk3 = pk[3]; pk4 = pk[4]; pk5 = pk[5];
          hp = pk4 >> 16;
        }
        if (hp == 0) dead_off = dd; // empty or fainted: zero block (network.h:142-143,153-160), kept out of phase 3
        else {
          doff = dd;
          uint32_t *li = Lidx + i * L::LCAP;
          float *lv = Lval + i * L::LCAP;
          if (!ACT) cntL = 8; // party slots: list positions 0..4 carry the stat VALUES (rows in registers), rows start at 8
          auto emit = [&](bool valid, uint32_t fidx, float fval, bool to_global) {
            if (!valid) return;
            if (to_global) { Gidx[i * 8 + cntG] = fidx; ++cntG; }
            else { li[cntL] = (ACT ? active_lds_slot(fidx) : fidx) * 128; lv[cntL] = fval; ++cntL; }
          };
          if (ACT) {
            const uint32_t *ac = sb + 36;
            const uint32_t a0 = ac[0], a1 = ac[1], a2 = ac[2], a3 = ac[3], a4 = ac[4], a5 = ac[5], a6 = ac[6], a7 = ac[7];
#pragma unroll
            for (uint32_t j = 0; j < 40; ++j) {
              uint32_t fidx = 0; float fval = 0.0f;
              const bool v = active_feature(j, a0, a1, a2, a3, a4, a5, a6, a7, dur, fidx, fval);
              emit(v, fidx, fval, j >= 32 && j < 36);
            }
#pragma unroll
            for (uint32_t j = 0; j < 12; ++j) {
              uint32_t fidx = 0; float fval = 0.0f;
              const bool v = pokemon_feature(j, pk0, pk1, pk2, pk3, pk4, pk5, dur & 7, fidx, fval);
              emit(v, fidx + 229, fval, j >= 5 && j < 9);
            }
          } else {
#pragma unroll
            for (uint32_t j = 0; j < 12; ++j) {
              uint32_t fidx = 0; float fval = 0.0f;
              const bool v = pokemon_feature(j, pk0, pk1, pk2, pk3, pk4, pk5, (dur >> (3 * slot)) & 7, fidx, fval);
              if (j < 5) lv[j] = fval; // always present
              else emit(v, fidx, fval, false);
            }
          }
          hp_ratio[i] = (float)hp / (float)(pk0 & 0xFFFF);
        }
      }
      if (mine) { meta[i] = cntL | (cntG << 8); dst_off[i] = doff; }
      uint64_t dead = __ballot(dead_off != 0xFFFFFFFFu); // the wave zeroes the blocks of its dead items together
      while (dead) {
        const int src = __ffsll((unsigned long long)dead) - 1;
        dead &= dead - 1;
        float *dst = a.emb + __shfl(dead_off, src, 64);
        for (uint32_t o = lane; o < (uint32_t)out_dim + 1; o += 64) dst[o] = 0.0f;
      }
    }
    EL_MARK(3);
    // ---- phase 1b/c: first layer, GI items at a time (their LDS reads interleave: common trip count, padded lists);
    // the active kind's move rows are prefetched from L2 first and added last (two items at a time: 32 registers) ----
    constexpr int GI = ACT ? 2 : 4;
#pragma unroll 1
    for (uint32_t half = 0; half < (uint32_t)(EL_ITEMS / GI); ++half) {
      const uint32_t ibase = wib * EL_ITEMS + half * GI;
      float gx0[GI][8], gx1[GI][8];
      if (ACT) {
#pragma unroll
        for (int t = 0; t < GI; ++t) {
          const uint4 ra = *(const uint4 *)(Gidx + (ibase + t) * 8), rb = *(const uint4 *)(Gidx + (ibase + t) * 8 + 4);
          const uint32_t rr[8] = {ra.x, ra.y, ra.z, ra.w, rb.x, rb.y, rb.z, rb.w};
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const float *row = w0t + (size_t)rr[u] * hidden;
            if (pair_ok && on1) { const float2 v = *(const float2 *)(row + c0); gx0[t][u] = v.x; gx1[t][u] = v.y; }
            else { gx0[t][u] = row[cc0]; gx1[t][u] = row[cc1]; }
          }
        }
      }
      float h0[GI], h1[GI];
#pragma unroll
      for (int t = 0; t < GI; ++t) { h0[t] = bias0; h1[t] = bias1; }
      uint32_t kmin = 0, kmax = 16; // party slots: stat values at 0..4 (below), at most 7 row features at 8..14
      if (!ACT) {
        kmin = 8;
#pragma unroll
        for (int t = 0; t < GI; ++t) {
          const float4 sv = *(const float4 *)(Lval + (ibase + t) * L::LCAP);
          const float s4 = Lval[(ibase + t) * L::LCAP + 4];
          h0[t] = fmaf(wst0[0], sv.x, h0[t]); h1[t] = fmaf(wst1[0], sv.x, h1[t]);
          h0[t] = fmaf(wst0[1], sv.y, h0[t]); h1[t] = fmaf(wst1[1], sv.y, h1[t]);
          h0[t] = fmaf(wst0[2], sv.z, h0[t]); h1[t] = fmaf(wst1[2], sv.z, h1[t]);
          h0[t] = fmaf(wst0[3], sv.w, h0[t]); h1[t] = fmaf(wst1[3], sv.w, h1[t]);
          h0[t] = fmaf(wst0[4], s4, h0[t]); h1[t] = fmaf(wst1[4], s4, h1[t]);
        }
      }
      if (ACT) {
        kmax = 0;
#pragma unroll
        for (int t = 0; t < GI; ++t) { const uint32_t c = __builtin_amdgcn_readfirstlane(meta[ibase + t]) & 0xFF; kmax = c > kmax ? c : kmax; }
      }
#pragma unroll 1
      for (uint32_t k = kmin; k < kmax; k += 4) {
#pragma unroll
        for (int t = 0; t < GI; ++t) {
          const uint4 iv = *(const uint4 *)(Lidx + (ibase + t) * L::LCAP + k);
          const float4 vv = *(const float4 *)(Lval + (ibase + t) * L::LCAP + k);
          const float2 wx = *(const float2 *)(W0s + iv.x + c0), wy = *(const float2 *)(W0s + iv.y + c0);
          const float2 wz = *(const float2 *)(W0s + iv.z + c0), ww = *(const float2 *)(W0s + iv.w + c0);
          h0[t] = fmaf(wx.x, vv.x, h0[t]); h1[t] = fmaf(wx.y, vv.x, h1[t]);
          h0[t] = fmaf(wy.x, vv.y, h0[t]); h1[t] = fmaf(wy.y, vv.y, h1[t]);
          h0[t] = fmaf(wz.x, vv.z, h0[t]); h1[t] = fmaf(wz.y, vv.z, h1[t]);
          h0[t] = fmaf(ww.x, vv.w, h0[t]); h1[t] = fmaf(ww.y, vv.w, h1[t]);
        }
      }
#pragma unroll
      for (int t = 0; t < GI; ++t) {
        const uint32_t i = ibase + t;
        if (ACT) {
          const uint32_t nG = __builtin_amdgcn_readfirstlane(meta[i]) >> 8;
#pragma unroll
          for (int u = 0; u < 8; ++u) { const float w = (uint32_t)u < nG ? 1.0f : 0.0f; h0[t] = fmaf(gx0[t][u], w, h0[t]); h1[t] = fmaf(gx1[t][u], w, h1[t]); }
        }
        const bool live = dst_off[i] != 0xFFFFFFFFu;
        float *hrow = Hs + i * EHP;
        hrow[c0] = (on0 && live) ? act_fn(h0[t], N.activation) : 0.0f;
        hrow[c1] = (on1 && live) ? act_fn(h1[t], N.activation) : 0.0f;
      }
    }
    EL_MARK(4);
    __syncthreads();
    EL_MARK(5);
    // Every wave is done with this tile's battles: stage the NEXT tile's (prefetched at least one whole phase 1 ago,
    // so the wait is short -- and it is a wait on loads only: placed after the scatter stores below, the same
    // s_waitcnt vmcnt(0) would also sit out this tile's stores) and start fetching the one after.
    stage();
    prefetch(tile + 2 * gridDim.x);
    EL_MARK(1);
    // ---- phase 2: OUT[64][out_pad] = H[64][128] . W1^T on fp32 MFMA (one 32x32 block per wave) ----
    if (mfma_wave) {
      f32x16 acc;
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[q] = 0.0f;
      const float *arow = Hs + (mi * 32 + r32) * EHP + hh;
#pragma unroll
      for (int s2 = 0; s2 < 64; ++s2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(arow[2 * s2], bfrag[s2], acc, 0, 0, 0);
      EL_MARK(6);
      // ---- phase 3: bias + activation + scatter ----
      const int o = nb * 32 + r32;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int row = (q & 3) + 8 * (q >> 2) + 4 * hh;
        const uint32_t doff = dst_off[mi * 32 + row];
        if (o < out_dim && doff != 0xFFFFFFFFu) a.emb[(size_t)doff + 1 + o] = act_fn(acc[q] + obias, N.activation);
      }
    }
    if (threadIdx.x < ET && dst_off[threadIdx.x] != 0xFFFFFFFFu) a.emb[dst_off[threadIdx.x]] = hp_ratio[threadIdx.x];
    EL_MARK(7);
  }
  EL_FLUSH();
}

constexpr int ER_ITEMS = 32, ER_RS = 128; // items per wave mini-tile; LDS weight row (floats) of k_embed_arows / k_embed_prows (a half-wave
                                          // reads a whole row per instruction: contiguous, conflict-free at any stride -- no padding)

// ---- the embedding nets' SECOND layers on the bf16 matrix pipe (round 4) ----
// Rounds 2-3 ran them on v_mfma_f32_32x32x2_f32: 128 (party) / 192 (actives) MFMAs of 64 cycles per 32-item mini-tile, 8.2 k of a
// party mini-tile's 13.1 k matrix-pipe cycles.  The bf16 pipe is 16x faster per FLOP, and an fp32 value is the exact sum of three
// bf16 parts (h, m, l), so, as in k_mainnet_split:
//   second layer : the 128 hidden activations of an item as triples against W1's triples (LDS, prebuilt), the six largest partial
//                  products per 32x32x16 block, fp32 accumulation: 8 k-steps x 6 x NBo MFMAs of 32 cycles (party 96: 3.1 k cycles
//                  against 8.2 k) -- fp32 results (dropped terms < 2^-24 of a product);
//   transposition: on the bf16 pipe too (SelTranspose below: the one-hot selector is exact in bf16; twelve MFMAs move the (h, m, l)
//                  parts of four k-steps' row sums into the item lanes).  A first attempt with the round-to-nearest split cost
//                  ~1,000 more vector instructions per mini-tile and lost; with the truncation split it is 352, and the fp32
//                  MFMAs it replaces turned out to hold the vector issue for their whole 64 cycles.
// ---- fp32 values as SCALED fp16 PAIRS (round 5; the full argument is at k_mainnet_pair): x s = h + l to 2^-24 with h = fp16(x s), l =
// fp16(x s - h), s a power of two that takes the largest value of the row (weights: of the layer) into [2^14, 2^15) ----
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
// the power of two that takes m (> 0: callers floor it) into [2^14, 2^15), and the exact ratio of two such scales
__device__ __forceinline__ float mp_scale_of(float m) { return __uint_as_float((268u - (__float_as_uint(m) >> 23)) << 23); }
__device__ __forceinline__ float mp_ratio(float to, float from) { // (a ratio below 2^-126 is 0: the sums it would multiply are then zeros or nothing)
  const int e = 127 + (int)(__float_as_uint(to) >> 23) - (int)(__float_as_uint(from) >> 23);
  return __uint_as_float((uint32_t)(e > 0 ? e : 0) << 23);
}
__device__ __forceinline__ float mp_inverse(float s) { return __uint_as_float((254u - (__float_as_uint(s) >> 23)) << 23); }
constexpr float MP_FLOOR = 0x1p-110f; // a row of zeros (or of values below this): scale 2^124, every part 0 (or a subnormal's worth)
// elements [e0, e1) of a k-step's 8 activation values, times the row's scale, as their pairs
#ifndef OAK_MP_SPLIT
#define OAK_MP_SPLIT 0 // 0: both parts rounded to nearest; 1 (experiment): the high part by truncation, two values per instruction (v_cvt_pkrtz_f16_f32); 2 (experiment, WRONG results): no split at all
#endif
#ifndef OAK_MP_VALU
#define OAK_MP_VALU 3  // vector instructions the scheduler may place behind each MFMA of a block
#endif
__device__ __forceinline__ void mp_split_part(const float (&v)[8], float scale, f16x8 (&B)[2], int e0, int e1) {
#if OAK_MP_SPLIT == 2
  if (e0 == 0) { B[0] = __builtin_bit_cast(f16x8, *(const float4 *)&v[0]); B[1] = __builtin_bit_cast(f16x8, *(const float4 *)&v[4]); }
  (void)scale; (void)e1;
#elif OAK_MP_SPLIT == 1
  typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
#pragma unroll
  for (int i = 0; i < 8; i += 2)
    if (i >= e0 && i < e1) { // (e0, e1 even or the whole k-step: NB <= 4 or one call)
      const float x0 = v[i] * scale, x1 = v[i + 1] * scale;
      const f16x2 hi = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(x0, x1));
      B[0][i] = hi[0]; B[0][i + 1] = hi[1];
      B[1][i] = (_Float16)(x0 - (float)hi[0]);
      B[1][i + 1] = (_Float16)(x1 - (float)hi[1]);
    }
#else
#pragma unroll
  for (int i = 0; i < 8; ++i)
    if (i >= e0 && i < e1) {
      const float x = v[i] * scale;
      const _Float16 hi = (_Float16)x;
      B[0][i] = hi;
      B[1][i] = (_Float16)(x - (float)hi);
    }
#endif
}
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
constexpr int E2_BLOCK_BYTES = 8 * 3 * 1024; // W1's triples of one 32-wide output block: [k-step T][h m l][lane] x 16 B
// v[0..7] as their bf16 triples, by TRUNCATION: h = the top 16 bits of x (its 8 leading significant bits), m = the top 16 bits of
// x - h (the next 8), l = x - h - m (the last 8: it has no more, so its low 16 bits are zero) -- x = h + m + l EXACTLY, every
// subtraction exact, four vector instructions per value (and, sub, and, sub) and one v_perm per pair and part to pack; the
// round-to-nearest split of k_mainnet_split costs 7.5 (these kernels are bound by the instructions a wave issues).
__device__ __forceinline__ void e_split(const float (&v)[8], bf16x8 (&P)[3]) {
  uint32_t hb[8], mb[8], lb[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    hb[i] = __float_as_uint(v[i]) & 0xFFFF0000u;
    const float r1 = v[i] - __uint_as_float(hb[i]);
    mb[i] = __float_as_uint(r1) & 0xFFFF0000u;
    lb[i] = __float_as_uint(r1 - __uint_as_float(mb[i]));
  }
  typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
  u32x4 ph, pm, pl;
#pragma unroll
  for (int i = 0; i < 4; ++i) { // element 2i in the low half, 2i + 1 in the high half: bytes {a.2, a.3, b.2, b.3}
    ph[i] = __builtin_amdgcn_perm(hb[2 * i + 1], hb[2 * i], 0x07060302u);
    pm[i] = __builtin_amdgcn_perm(mb[2 * i + 1], mb[2 * i], 0x07060302u);
    pl[i] = __builtin_amdgcn_perm(lb[2 * i + 1], lb[2 * i], 0x07060302u);
  }
  P[0] = __builtin_bit_cast(bf16x8, ph); P[1] = __builtin_bit_cast(bf16x8, pm); P[2] = __builtin_bit_cast(bf16x8, pl);
}
// First layer, the selector transposition on the bf16 pipe.  In k-step t the half-wave hh holds the row sums of item 2 t + hh,
// lane (i, hh) the channels 4 i .. 4 i + 3 (`sum`); the second layer wants "lane = item".  D[channel i][item n] += sum over k of
// A[i][k] S[k][n] with a one-hot selector S[k][n] = [item of k == n] does it, and S is exact in bf16 -- but the sums are fp32, so
// they go through as their (h, m, l) triples (truncation split, e_split): 16 + 6 vector instructions per k-step and 12
// v_mfma_f32_32x32x8_bf16 per FOUR k-steps (element j of lane (i, kh) = k-step 4 G + j, item 2 (4 G + j) + kh), against four
// v_mfma_f32_32x32x2_f32 per k-step.  The fp32 MFMA is the expensive one: tools/experiments/mfma_overlap_bench.hip shows that a
// 32x32x2 fp32 MFMA holds the SIMD's vector issue for ALL of its 64 cycles (time = pipe + issue: it runs on the vector ALUs),
// while the bf16 ones overlap with vector work (hold: 8 cycles).  Each (channel, item) receives exactly one non-zero term per
// part, l then m then h, so the result is the fp32 sum to within two roundings of the accumulator.
typedef short s16x4 __attribute__((ext_vector_type(4)));
struct SelTranspose {
  uint32_t pe[4][3];    // the even k-step's parts, waiting for the odd one
  uint32_t pa[4][3][2]; // four k-steps packed: [component][part][pair]
  __device__ __forceinline__ void step(const int t, const float4 &sum, f32x16 (&hb)[4], const uint32_t r32, const uint32_t hh) {
    const float v[4] = {sum.x, sum.y, sum.z, sum.w};
    uint32_t p[4][3];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      p[c][0] = __float_as_uint(v[c]) & 0xFFFF0000u;
      const float r1 = v[c] - __uint_as_float(p[c][0]);
      p[c][1] = __float_as_uint(r1) & 0xFFFF0000u;
      p[c][2] = __float_as_uint(r1 - __uint_as_float(p[c][1]));
    }
    if ((t & 1) == 0) {
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int q = 0; q < 3; ++q) pe[c][q] = p[c][q];
      return;
    }
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int q = 0; q < 3; ++q) pa[c][q][(t >> 1) & 1] = __builtin_amdgcn_perm(p[c][q], pe[c][q], 0x07060302u); // even step low half, odd step high half
    if ((t & 3) != 3) return;
    // the selector of this group of four k-steps: element j of lane (n, kh) is 1.0 iff n == 2 (4 G + j) + kh
    const int e = (int)r32 - (int)hh - 8 * (t >> 2);
    typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
    u32x2 sel;
    sel[0] = e == 0 ? 0x3F80u : e == 2 ? 0x3F800000u : 0u;
    sel[1] = e == 4 ? 0x3F80u : e == 6 ? 0x3F800000u : 0u;
    const s16x4 S = __builtin_bit_cast(s16x4, sel);
#pragma unroll
    for (int q = 2; q >= 0; --q) // l, m, h: small parts first
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        u32x2 a;
        a[0] = pa[c][q][0]; a[1] = pa[c][q][1];
        hb[c] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(__builtin_bit_cast(s16x4, a), S, hb[c], 0, 0, 0);
      }
  }
};
// First layer, the DENSE features (bias, stats, boosts, volatiles: the weight row is the same for every item, only the value
// differs) on the bf16 pipe as well: H^T = W0d^T . X^T with both sides as triples, v_mfma_f32_32x32x8_bf16, KT k-steps of eight
// features.  Lane (item r, kh) holds the values of features 8 T + 4 kh + j (xv[T][j]) -> their triples xp; the weights'
// triples are prebuilt in LDS, 8 bytes per lane: [T][channel block][h m l][lane], lane (i, kh) element j = W0[channel 4 i + blk]
// [row of feature 8 T + 4 kh + j] (feature 0 = the bias).  Six products per block and k-step as in embed_layer2.  On the fp32 pipe
// (v_mfma_f32_32x32x2_f32, 64 cycles each, vector issue held throughout) the actives' 36 features cost 4.6 k cycles per mini-tile.
template <int KT>
__device__ __forceinline__ void dense_split(const float (&xv)[KT][4], uint32_t (&xp)[KT][3][2]) {
#pragma unroll
  for (int T = 0; T < KT; ++T) {
    uint32_t p[4][3];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      p[j][0] = __float_as_uint(xv[T][j]) & 0xFFFF0000u;
      const float r1 = xv[T][j] - __uint_as_float(p[j][0]);
      p[j][1] = __float_as_uint(r1) & 0xFFFF0000u;
      p[j][2] = __float_as_uint(r1 - __uint_as_float(p[j][1]));
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      xp[T][q][0] = __builtin_amdgcn_perm(p[1][q], p[0][q], 0x07060302u);
      xp[T][q][1] = __builtin_amdgcn_perm(p[3][q], p[2][q], 0x07060302u);
    }
  }
}
template <int KT>
__device__ __forceinline__ void dense_layer_bf16(const uint8_t *wd_lane, const uint32_t (&xp)[KT][3][2], f32x16 (&hb)[4]) {
  typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
#pragma unroll
  for (int T = 0; T < KT; ++T) {
    s16x4 X[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      u32x2 v;
      v[0] = xp[T][q][0]; v[1] = xp[T][q][1];
      X[q] = __builtin_bit_cast(s16x4, v);
    }
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const uint8_t *p = wd_lane + (T * 4 + b) * 3 * 512;
      const s16x4 W0 = *(const s16x4 *)p, W1 = *(const s16x4 *)(p + 512), W2 = *(const s16x4 *)(p + 1024);
      hb[b] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(W0, X[2], hb[b], 0, 0, 0); // h . l  (small terms first)
      hb[b] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(W2, X[0], hb[b], 0, 0, 0); // l . h
      hb[b] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(W1, X[1], hb[b], 0, 0, 0); // m . m
      hb[b] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(W0, X[1], hb[b], 0, 0, 0); // h . m
      hb[b] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(W1, X[0], hb[b], 0, 0, 0); // m . h
      hb[b] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(W0, X[0], hb[b], 0, 0, 0); // h . h
    }
  }
}
// Second layer: hb = the item's 128 activated hidden channels (lane (item r, hh): register s of block b = channel ar_channel),
// w1t = this lane's 16 bytes of W1's triple image in LDS.  Orientation Out^T = W1 . H^T: the weights are the A operand (lane
// (o, h) element j = W1[32 nb + o][ar_channel(8 T + j, h)], prebuilt on the host: embed_pair_order), the activations the B
// operand (lane (item r, h) element j = that channel = its own registers 8 T .. 8 T + 7) -- so acc[nb] holds, on lane (item r,
// h), the item's outputs 32 nb + 8 (q >> 2) + 4 h + (q & 3): every lane scatters ITS OWN item, four consecutive outputs per
// 16-byte store (embed_scatter), with the destination offset in a register.  (Rounds 2-3 had the outputs on the lanes and the
// items in the registers: 16 four-byte stores per block, each behind an LDS read of the item's offset, a branch and -- stores
// count in vmcnt -- the completion of the store before it: 11.8 k of a party mini-tile's 35 k cycles.)
// Round 5: as scaled fp16 PAIRS instead of bf16 triples -- three MFMAs per block and k-step instead of six (the phase was bound by its
// chain of dependent MFMAs: 8 k-steps x 6 x NBo x 32 cycles against ~350 vector instructions).  The item's scale comes from the largest of
// its 128 hidden activations (64 registers of this lane, 64 of its partner); W1's pairs carry the layer's scale (embed_pair_order).
// Returns 1 / the item's scale: embed_scatter multiplies the sums back (x N.*_l2_inv, the layer's).
template <int NBMAX>
__device__ __forceinline__ float embed_layer2(const f32x16 (&hb)[4], const uint8_t *w1t, int NBo, f32x16 (&acc)[NBMAX]) {
  float m = MP_FLOOR;
#pragma unroll
  for (int b = 0; b < 4; ++b)
#pragma unroll
    for (int q = 0; q < 16; q += 2) m = fmaxf(m, fmaxf(fabsf(hb[b][q]), fabsf(hb[b][q + 1])));
  m = fmaxf(m, __shfl_xor(m, 32, 64));
  const float scale = mp_scale_of(m);
#pragma unroll
  for (int nb = 0; nb < NBMAX; ++nb)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[nb][q] = 0.0f;
#pragma unroll
  for (int T = 0; T < 8; ++T) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = hb[T >> 1][8 * (T & 1) + j];
    f16x8 A[2];
    mp_split_part(v, scale, A, 0, 8);
#pragma unroll
    for (int nb = 0; nb < NBMAX; ++nb)
      if (nb < NBo) { // wave-uniform
        const uint8_t *p = w1t + nb * E2_BLOCK_BYTES + T * 3072; // (the image keeps the triples' 3 x 1 KB per k-step; the third is unused)
        const f16x8 W0 = *(const f16x8 *)p, W1 = *(const f16x8 *)(p + 1024);
        acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(W1, A[0], acc[nb], 0, 0, 0); // l . h   (small terms first)
        acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(W0, A[1], acc[nb], 0, 0, 0); // h . l
        acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(W0, A[0], acc[nb], 0, 0, 0); // h . h
      }
  }
  return mp_inverse(scale);
}
// The item's outputs to its block of the battle embedding: bias + activation, four consecutive outputs per store (the block
// starts one float behind a 16-byte boundary -- the hp ratio comes first -- so the stores are 4-byte aligned dwordx4's), all
// stores of a lane back to back.  doff = the item's block (0xFFFFFFFF: no live item on this lane); bias padded to 32 NBo floats.
struct __attribute__((packed, aligned(4))) f4u { float x, y, z, w; };
template <int NBMAX>
__device__ __forceinline__ void embed_scatter(float *emb, uint32_t doff, float hpr, uint32_t hh, const f32x16 (&acc)[NBMAX], int NBo, int out_dim,
                                              const float *bias, int activation, float inv) { // inv: 1 / the item's scale; bias + 32 NBo: 1 / the scales of W1's rows (embed_pair_order)
  float4 b[NBMAX][4];
#pragma unroll
  for (int nb = 0; nb < NBMAX; ++nb)
#pragma unroll
    for (int g = 0; g < 4; ++g) b[nb][g] = nb < NBo ? *(const float4 *)(bias + 32 * nb + 8 * g + 4 * hh) : make_float4(0.f, 0.f, 0.f, 0.f);
  f32x16 o[NBMAX];
#pragma unroll
  for (int nb = 0; nb < NBMAX; ++nb)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 sc = nb < NBo ? *(const float4 *)(bias + 32 * NBo + 32 * nb + 8 * g + 4 * hh) : make_float4(0.f, 0.f, 0.f, 0.f);
      o[nb][4 * g + 0] = acc[nb][4 * g + 0] * inv * sc.x + b[nb][g].x; o[nb][4 * g + 1] = acc[nb][4 * g + 1] * inv * sc.y + b[nb][g].y;
      o[nb][4 * g + 2] = acc[nb][4 * g + 2] * inv * sc.z + b[nb][g].z; o[nb][4 * g + 3] = acc[nb][4 * g + 3] * inv * sc.w + b[nb][g].w;
    }
  act_blocks<NBMAX>(o, activation);
  // (opaque per call: the width tests below depend on the lane's half only, so the compiler hoisted all 48 of them -- 64-bit lane
  // masks, two SGPRs each -- out of the mini-tile loop and kept them alive across it: 93 of the actives pass's 93 spilled SGPRs,
  // a v_readlane pair per mask and mini-tile; recomputed here they are 48 compares per mini-tile)
  asm volatile("" : "+v"(out_dim));
  if (doff != 0xFFFFFFFFu) {
    float *dst = emb + (size_t)doff + 1;
#pragma unroll
    for (int nb = 0; nb < NBMAX; ++nb)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int o0 = 32 * nb + 8 * g + 4 * (int)hh;
        if (nb < NBo && o0 < out_dim) {
          f4u v;
          v.x = o[nb][4 * g + 0]; v.y = o[nb][4 * g + 1]; v.z = o[nb][4 * g + 2]; v.w = o[nb][4 * g + 3];
          if (o0 + 3 < out_dim) *(f4u *)(dst + o0) = v;
          else { dst[o0] = v.x; if (o0 + 1 < out_dim) dst[o0 + 1] = v.y; if (o0 + 2 < out_dim) dst[o0 + 2] = v.z; }
        }
      }
    if (hh == 0) emb[doff] = hpr;
  }
}

// Copy a prebuilt image of the kernel's LDS weights (built once at load time in exactly the LDS layout) by LDS-DMA
// (global_load_lds_dwordx4: a wave instruction lands 64 x 16 B at a wave-uniform LDS base, lane-linear; no staging registers).
// The pieces are issued at the top of the kernel; the wait (stage_image_wait: vmcnt(0) written out -- a bare __syncthreads() is
// not a wait for LDS-DMA -- and the workgroup barrier) comes after the wave's FIRST encode, which needs no weights.
// History: the obvious `lds[i] = cond ? global[f(i)] : 0` loop compiled to load -> wait -> store per iteration (~30 serialized L2
// round trips per workgroup, 20-50 us in front of every embedding kernel); rounds 2-4 staged through registers, every load of a
// thread in flight at once -- 76-80 registers across the first encode, which is where the kernels' last 40 spilled registers were.
typedef __attribute__((address_space(3))) void img_lds_void;
typedef __attribute__((address_space(1))) const void img_glb_void;
template <int BLOCK>
__device__ __forceinline__ void stage_image_dma(float *lds, const float *img, int words) {
  const int n16 = words >> 2; // 16-byte pieces (images are padded to a multiple of 4 floats)
  const int tid = (int)threadIdx.x, wave = tid >> 6;
  for (int base = 0; base < n16; base += BLOCK) { // wave-uniform trip count
    const int first = base + wave * 64;            // this wave's 1-KB piece
    if (first + (tid & 63) < n16)
      __builtin_amdgcn_global_load_lds((img_glb_void *)((const float4 *)img + first + (tid & 63)), (img_lds_void *)((float4 *)lds + first), 16, 0, 0);
  }
}
__device__ __forceinline__ void stage_image_wait() {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
}

// The embedding blocks of empty / fainted slots are all-zero (network.h:142-143,153-160).  The lanes that found one hand it to
// the whole wave: one coalesced store per dead item instead of `len` four-byte stores by a single lane.
__device__ __forceinline__ void zero_blocks(float *emb, uint32_t dead_off, int len) {
  const int lane = threadIdx.x & 63;
  unsigned long long m = __ballot(dead_off != 0xFFFFFFFFu);
  while (m) {
    const int l = __builtin_ctzll(m);
    m &= m - 1;
    const uint32_t off = (uint32_t)__builtin_amdgcn_readlane((int)dead_off, l);
    for (int o = lane; o < len; o += 64) emb[(size_t)off + o] = 0.0f;
  }
}

// ---- K2, the actives' pass (the default): row-per-lane like k_embed_rows, with the DENSE part of the first layer on the
// matrix pipe.  An active's ~53 candidate features fall into three kinds:
//   36 DENSE features (bias, the 5 active stats, 6 boosts, 19 volatile features, the stored Pokemon's 5 stats): the weight
//                  row is the same for every item, only the VALUE differs -- a 32-item x 36 x 128 GEMM.  It runs as
//                  H^T = W0d^T . X^T on v_mfma_f32_32x32x2_f32 (A = the weight fragment, B = the items' values), because in
//                  THAT orientation the result lands as "lane = item, registers = channels": lane (r, hh) receives channels
//                  32 blk + (q & 3) + 8 (q >> 2) + 4 hh of item r -- already the A operand of the second layer's MFMAs, if W1's
//                  fragments and the sparse rows use the same channel order (ar_channel).  18 k-steps x 4 blocks = 72 MFMAs
//                  replace 36 x 16 ds_read_b128 + 36 x 64 FMAs per lane;
//    9 ONE-HOT rows in LDS (2 types, 4 durations, status, 2 stored types; 64 possible rows): a 16-bit LDS offset per item,
//                  absent = the zero row; added to the MFMA result with FMAs as in k_embed_rows;
//    8 MOVE rows   (4 active + 4 stored move slots, 328 possible rows): read from a copy in L2 in the same channel order
//                  (`a_w0d`, 219 KB), absent = its zero row.
// Then the <= 4 output blocks one after the other with W1's fragments from LDS (256 B per k-step, conflict-free).  No
// activation tile, no workgroup barrier per tile, every wave on the MFMA; the kernel fits 128 registers so that a CU holds
// ONE 16-wave workgroup (4 waves per SIMD).  LDS: 65 x 528 B rows + 18 KB dense fragment + <= 64 KB W1 fragments + 1.4 KB
// per wave (row indices) = <= 141 KB, staged from one prebuilt image (stage_image_dma).
constexpr int AR_SPARSE = 64, AR_ZERO = 64;   // LDS-resident one-hot rows (ar_sparse_slot) + a zero row
constexpr int AR_FIXED = 36, AR_HOT = 9, AR_MOVES = 8, AR_KT = (AR_FIXED + 7) / 8; // (dense features in k-steps of eight)
constexpr int AR_ITEM_WORDS = 20;               // ready to use: 9 LDS byte offsets of the item's one-hot rows, 8 byte offsets of its move rows in L2 (+ 3 pad)
constexpr int AR_WAVE_WORDS = ER_ITEMS * AR_ITEM_WORDS;
#ifndef OAK_EMBED_WAVES
#define OAK_EMBED_WAVES 8
#endif
// 8 waves of <= 256 registers (2 per SIMD): no scratch.  Measured, round 4, with the second layer on the bf16 pipe: 16 waves of
// 128 registers spill 150-400 of them (party pass 633 us), 12 of 168 spill 60-130 (246 us), 8 of 256 none (140 us).
constexpr int AR_WAVES = OAK_EMBED_WAVES, AR_BLOCK = 64 * AR_WAVES;
constexpr int AR_DENSE_WORDS = AR_KT * 4 * 3 * 64 * 2; // the dense weights' triples: [k-step][channel block][h m l][lane] x 8 B (dense_layer_bf16)
constexpr int AR_COMBINED = 428; // first precombined (active + stored) move row of a_w0d: behind W0^T's 427 rows and the zero row
constexpr int AR_MAX_NBO = 3; // W1 as bf16 triples: 24 KB per 32-wide output block; four blocks (outputs above 96) do not fit beside the rest
constexpr size_t ar_bytes(int nbo) { return (size_t)((AR_SPARSE + 1) * ER_RS + AR_DENSE_WORDS + AR_WAVES * AR_WAVE_WORDS + 64 * nbo) * 4 + (size_t)nbo * E2_BLOCK_BYTES; }
// channel held by register s (0..63) of a lane in half hh: the C layout of four 32x32 MFMA blocks, block b's row i being
// channel 4 i + b (so that a lane's float4 of a weight row feeds the four blocks)
__host__ __device__ constexpr int ar_channel(int s, int hh) { return 4 * ((s & 3) + 8 * ((s & 15) >> 2) + 4 * hh) + (s >> 4); }
// W0^T row of dense feature d (1..35; 0 is the bias): active stats, boosts, volatiles, the stored Pokemon's stats
__host__ __device__ constexpr int ar_dense_row(int d) { return d < 6 ? d - 1 : d < 12 ? 20 + (d - 6) : d < 31 ? 26 + (d - 12) : 229 + (d - 31); }
// compact LDS slot of a one-hot row (types 5..19, rows 209..228, rows 398..426) and back
__device__ __forceinline__ uint32_t ar_sparse_slot(uint32_t row) { return row < 20 ? row - 5 : row < 229 ? row - 209 + 15 : row - 398 + 35; }
__host__ __device__ constexpr int ar_sparse_row(int slot) { return slot < 15 ? slot + 5 : slot < 35 ? slot - 15 + 209 : slot - 35 + 398; }
// bias (padded to 32 NBo floats) behind W1's triples in both images: embed_scatter reads it from LDS
constexpr int ar_img_words(int nbo) { return (AR_SPARSE + 1) * ER_RS + AR_DENSE_WORDS + nbo * (E2_BLOCK_BYTES / 4) + 64 * nbo; } // (... + b1 + 1 / W1's row scales)
template <int WAVES, int NBO>
__device__ __forceinline__ void embed_arows_body(const EmbedTileArgs &a, float *lds_f, const uint32_t bid, const uint32_t nblocks) {
  constexpr int BLOCK = WAVES * 64;
  EL_T0();
  const NetDev &N = a.net;
  const int out_dim = N.a_out;
  float *W0s = lds_f;                                   // sparse rows, channels in natural order, + a zero row
  float *Wd = W0s + (AR_SPARSE + 1) * ER_RS;            // dense fragment: [k-step][channel block][lane]
  const uint8_t *W1t = (const uint8_t *)(Wd + AR_DENSE_WORDS); // the second layer's bf16 triples: [block][k-step][h m l][lane] x 16 B
  const float *b1s = (const float *)(W1t + NBO * E2_BLOCK_BYTES);
  constexpr int img_words = ar_img_words(NBO);
  stage_image_dma<BLOCK>(lds_f, N.a_img, img_words);
  const uint32_t lane = threadIdx.x & 63, wib = threadIdx.x >> 6, r32 = lane & 31, hh = lane >> 5;
  uint32_t *wl = (uint32_t *)(lds_f + img_words) + wib * AR_WAVE_WORDS; // this wave's private LDS (the items' row indices)
  const uint32_t items = a.n * 2;
  const uint32_t nmt = (items + ER_ITEMS - 1) / ER_ITEMS;
  const uint32_t stride = nblocks * WAVES;
  EL_MARK(0);
  // ---- encode, in three parts.  LOAD 1 (at the top of the previous mini-tile's body): both lanes (r, 0) and (r, 1) ask for the
  // slot of item r's stored Pokemon and the durations word; LOAD 2 (behind that body's first layer, when LOAD 1 has long
  // arrived): the active block and the stored Pokemon; nothing waits for either.  COMPUTE (behind its second layer): the same
  // instructions for the whole wave; each lane keeps the dense values of its own k-half in registers (x[t] = value 2t + hh,
  // the B operand below) and lane (r, 0) writes the row indices to the wave's LDS.  The wave's first encode runs BEFORE the
  // weights are written to LDS (it needs none): the image's round trip hides under it ----
  uint32_t xp[AR_KT][3][2]; // the dense values of this lane's k-half as bf16 triples (dense_split)
  uint32_t my_doff = 0xFFFFFFFFu; // this lane's item: its block of the embedding (both lanes (r, 0), (r, 1) hold item r's) and hp ratio
  float my_hpr = 0.0f;
  uint4 in_av0 = make_uint4(0, 0, 0, 0), in_av1 = in_av0;
  uint32_t in_id = 0, in_dur = 0, in_pk[6] = {0, 0, 0, 0, 0, 0};
  auto encode_load1 = [&](uint32_t mt)
  {
    const uint32_t g = mt * ER_ITEMS + r32;
    if (g < items) {
      const uint32_t leaf = g >> 1, side = g & 1;
      const uint32_t *sb = (const uint32_t *)a.battles + (size_t)leaf * 96 + side * 46;
      in_id = sb[44] & 0xFF;
      in_dur = ((const uint32_t *)a.durations)[(size_t)leaf * 2 + side];
    }
  };
  auto encode_load2 = [&](uint32_t mt)
  {
    const uint32_t g = mt * ER_ITEMS + r32;
#pragma unroll
    for (int k = 0; k < 6; ++k) in_pk[k] = 0;
    if (g < items) { // (the active block with LOAD 2: eight registers less across the first layer, which has none to spare)
      const uint32_t *sb = (const uint32_t *)a.battles + (size_t)(g >> 1) * 96 + (g & 1) * 46;
      in_av0 = *(const uint4 *)(sb + 36); in_av1 = *(const uint4 *)(sb + 40);
    }
    if (g < items && in_id != 0) {
      const uint32_t leaf = g >> 1, side = g & 1;
      const uint32_t *pk = (const uint32_t *)a.battles + (size_t)leaf * 96 + side * 46 + 6 * (in_id - 1);
      const uint2 p01 = *(const uint2 *)pk, p23 = *(const uint2 *)(pk + 2), p45 = *(const uint2 *)(pk + 4);
      in_pk[0] = p01.x; in_pk[1] = p01.y; in_pk[2] = p23.x; in_pk[3] = p23.y; in_pk[4] = p45.x; in_pk[5] = p45.y;
    }
  };
  auto encode_compute = [&](uint32_t mt)
  {
    const uint32_t g = mt * ER_ITEMS + r32;
    float fv[AR_FIXED];
    uint32_t hot[AR_HOT], mv[AR_MOVES];
    uint32_t doff = 0xFFFFFFFFu, dead_off = 0xFFFFFFFFu;
    float hpr = 0.0f;
#pragma unroll
    for (int f = 0; f < AR_FIXED; ++f) fv[f] = 0.0f;
#pragma unroll
    for (int k = 0; k < AR_HOT; ++k) hot[k] = AR_ZERO * ER_RS / 4;
#pragma unroll
    for (int k = 0; k < AR_MOVES; ++k) mv[k] = 427;
    if (g < items) {
      const uint32_t leaf = g >> 1, side = g & 1;
      const uint4 av0 = in_av0, av1 = in_av1;
      const uint32_t dur = in_dur, id = in_id;
      const uint32_t dd = leaf * N.emb_dim + side * N.side_dim;
      const uint32_t pk0 = in_pk[0], pk1 = in_pk[1], pk2 = in_pk[2], pk3 = in_pk[3], pk4 = in_pk[4], pk5 = in_pk[5];
      const uint32_t hp = id != 0 ? pk4 >> 16 : 0u;
      if (hp == 0) dead_off = dd; // no active / fainted active: zero block (network.h:142-143)
      else {
        doff = dd;
        fv[0] = 1.0f; // the bias
        int nh = 0, nm = 0;
#pragma unroll
        for (uint32_t j = 0; j < 40; ++j) { // Encode::Battle::Active (battle.h:229-489)
          uint32_t fidx = 0; float fval = 0.0f;
          const bool v = active_feature(j, av0.x, av0.y, av0.z, av0.w, av1.x, av1.y, av1.z, av1.w, dur, fidx, fval);
          if (j < 5) fv[1 + j] = fval;                                   // stats: always present
          else if (j < 7) { hot[nh++] = v ? ar_sparse_slot(fidx) * ER_RS / 4 : AR_ZERO * ER_RS / 4; } // types
          else if (j < 13) fv[6 + (j - 7)] = fval;                       // boosts: always present
          else if (j < 32) fv[12 + (j - 13)] = v ? fval : 0.0f;          // volatiles: dense, value 0 when absent
          else if (j < 36) { mv[nm++] = v ? fidx : 427u; }               // move slots: rows in L2
          else { hot[nh++] = v ? ar_sparse_slot(fidx) * ER_RS / 4 : AR_ZERO * ER_RS / 4; } // durations
        }
#pragma unroll
        for (uint32_t j = 0; j < 12; ++j) { // Encode::Battle::Pokemon of the stored active (battle.h:197-214), rows + 229
          uint32_t fidx = 0; float fval = 0.0f;
          const bool v = pokemon_feature(j, pk0, pk1, pk2, pk3, pk4, pk5, dur & 7, fidx, fval);
          if (j < 5) fv[31 + j] = fval;
          else if (j < 9) { mv[nm++] = v ? fidx + 229 : 427u; }
          else { hot[nh++] = v ? ar_sparse_slot(fidx + 229) * ER_RS / 4 : AR_ZERO * ER_RS / 4; }
        }
        hpr = (float)hp / (float)(pk0 & 0xFFFF);
      }
    }
    float xv[AR_KT][4];
    // Slot k's active move row (45 + id - 1) and stored move row (229 + 5 + id - 1) are the same move unless Transform / Mimic
    // replaced the active one: the pair is then ONE load of the precombined row AR_COMBINED + id - 1 (arows_rows: the sum of the
    // two) and the stored half of the pair becomes the zero row; `extra` says whether any stored half is left
    uint32_t extra = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const bool same = mv[k] != 427u && mv[4 + k] == mv[k] + 189u;
      mv[k] = same ? mv[k] - 45u + AR_COMBINED : mv[k];
      mv[4 + k] = same ? 427u : mv[4 + k];
      extra |= mv[4 + k] != 427u ? 1u : 0u;
    }
#pragma unroll
    for (int T = 0; T < AR_KT; ++T)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int d0 = 8 * T + j, d1 = d0 + 4;
        xv[T][j] = hh ? (d1 < AR_FIXED ? fv[d1 < AR_FIXED ? d1 : 0] : 0.0f) : (d0 < AR_FIXED ? fv[d0 < AR_FIXED ? d0 : 0] : 0.0f);
      }
    dense_split<AR_KT>(xv, xp);
    if (hh == 0) { // (byte offsets: the readers add their own 16 bytes of the row and nothing else)
      uint32_t *it = wl + r32 * AR_ITEM_WORDS;
      *(uint4 *)it = make_uint4(hot[0] * 16, hot[1] * 16, hot[2] * 16, hot[3] * 16);
      *(uint4 *)(it + 4) = make_uint4(hot[4] * 16, hot[5] * 16, hot[6] * 16, hot[7] * 16);
      *(uint4 *)(it + 8) = make_uint4(hot[8] * 16, mv[0] * 512, mv[1] * 512, mv[2] * 512);
      *(uint4 *)(it + 12) = make_uint4(mv[3] * 512, mv[4] * 512, mv[5] * 512, mv[6] * 512);
      *(uint2 *)(it + 16) = make_uint2(mv[7] * 512, extra);
    }
    my_doff = doff;
    my_hpr = hpr;
    zero_blocks(a.emb, hh == 0 ? dead_off : 0xFFFFFFFFu, out_dim + 1);
  };
  uint32_t mt = bid * WAVES + wib;
  if (mt < nmt) { encode_load1(mt); encode_load2(mt); encode_compute(mt); } // needs no weights: the image's round trip runs under it
  stage_image_wait(); // weights staged (the only workgroup barrier of the kernel)
  while (mt < nmt) {
    __builtin_amdgcn_wave_barrier();
    EL_MARK(3);
    __builtin_amdgcn_s_setprio(2); // a wave in its MFMA phases goes before waves that encode or scatter (see the second layer)
    const uint32_t next = mt + stride;
    if (next < nmt) encode_load1(next);
    const uint8_t *w0lane = (const uint8_t *)W0s + 16 * r32, *mvlane = (const uint8_t *)N.a_w0d + 16 * r32;
    // ---- first layer, dense part on the matrix pipe: hb[blk] = this lane's 16 channels of block blk of item r32 ----
    f32x16 hb[4];
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int q = 0; q < 16; ++q) hb[b][q] = 0.0f;
    dense_layer_bf16<AR_KT>((const uint8_t *)Wd + lane * 8, xp, hb);
    // ---- the one-hot and move rows.  Read "lane = item" they cost a 16-byte piece of 64 different cache lines per load
    // instruction (8,192 line requests per mini-tile: the L1 was the bottleneck of the whole kernel).  So they are read
    // "lane = channels": in k-step t the half-wave hh sums the rows of item 2t + hh, lane (i, hh) taking channels
    // 4i .. 4i+3 (one float4: a half-wave reads a whole 512-byte row per instruction), and SelTranspose moves the sums into
    // the "lane = item" registers.
    // The move rows come from L2: four loads per k-step (the slots' COMBINED rows, see encode_compute) and four more only if
    // one of the step's two items has a slot whose active and stored move differ (Transform, Mimic).  Their round trip is not
    // what the step waits for: with register rings that keep one or two later k-steps' loads in flight the pass took 72 / 99 us
    // against 67.5 (round 4) -- what costs is the REQUESTS: with every row an L1 hit the pass took 56 us.
    SelTranspose tr;
    // the row offsets of k-step t + 1 are read while step t's rows are summed (a step's thirteen row reads wait for them)
    struct StepIdx { uint4 i0, i1, i2, i3; uint2 i4; };
    auto idx_of = [&](int t, StepIdx &I) {
      const uint32_t *ip = wl + (2 * t + hh) * AR_ITEM_WORDS;
      I.i0 = *(const uint4 *)ip; I.i1 = *(const uint4 *)(ip + 4); I.i2 = *(const uint4 *)(ip + 8); I.i3 = *(const uint4 *)(ip + 12);
      I.i4 = *(const uint2 *)(ip + 16);
    };
    StepIdx Ic, In;
    idx_of(0, Ic);
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      if (t + 1 < 16) idx_of(t + 1, In);
      const uint4 i0 = Ic.i0, i1 = Ic.i1, i2 = Ic.i2, i3 = Ic.i3;
      const uint2 i4 = Ic.i4;
      const uint32_t mo[AR_MOVES] = {i2.y, i2.z, i2.w, i3.x, i3.y, i3.z, i3.w, i4.x};
      float4 g[AR_MOVES];
#pragma unroll
      for (int k = 0; k < 4; ++k) g[k] = *(const float4 *)(mvlane + mo[k]); // rows in L2
      const bool extra = __builtin_amdgcn_ballot_w64(i4.y != 0) != 0; // wave-uniform
      if (extra) {
#pragma unroll
        for (int k = 4; k < AR_MOVES; ++k) g[k] = *(const float4 *)(mvlane + mo[k]);
      }
      const uint32_t ho[AR_HOT] = {i0.x, i0.y, i0.z, i0.w, i1.x, i1.y, i1.z, i1.w, i2.x};
      float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int k = 0; k < AR_HOT; ++k) {
        const float4 xr = *(const float4 *)(w0lane + ho[k]);
        sum.x += xr.x; sum.y += xr.y; sum.z += xr.z; sum.w += xr.w;
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) { sum.x += g[k].x; sum.y += g[k].y; sum.z += g[k].z; sum.w += g[k].w; }
      if (extra) {
#pragma unroll
        for (int k = 4; k < AR_MOVES; ++k) { sum.x += g[k].x; sum.y += g[k].y; sum.z += g[k].z; sum.w += g[k].w; }
      }
      uint32_t r32o = r32;
      asm volatile("" : "+v"(r32o)); // (opaque: unrolled, the selectors would otherwise be hoisted out of the mini-tile loop into registers)
      tr.step(t, sum, hb, r32o, hh);
      Ic = In;
    }
    act_blocks<4>(hb, N.activation);
    __builtin_amdgcn_sched_barrier(0);
    EL_MARK(4);
    if (next < nmt) encode_load2(next); // (the slot byte of LOAD 1 arrived during the first layer)
    // ---- second layer on the bf16 pipe (embed_layer2): W1's triples from LDS ----
    // wave priority: second layer 3 > first layer 2 > encode / scatter 0 -- whoever can feed the matrix pipe issues first
    __builtin_amdgcn_s_setprio(3);
    f32x16 acc[NBO];
    const float inv_item = embed_layer2<NBO>(hb, W1t + lane * 16, NBO, acc);
    EL_MARK(6);
    __builtin_amdgcn_s_setprio(0);
    const uint32_t cur_doff = my_doff;
    const float cur_hpr = my_hpr;
    __builtin_amdgcn_wave_barrier(); // the wave's row indices are rewritten by the next mini-tile's encode
    // the next encode BEFORE this mini-tile's stores: its loads are then the oldest vector-memory operations in flight and the
    // wait for them does not wait for the stores (vmcnt retires in order)
    if (next < nmt) encode_compute(next);
    EL_MARK(7);
    embed_scatter<NBO>(a.emb, cur_doff, cur_hpr, hh, acc, NBO, out_dim, b1s, N.activation, inv_item);
    EL_MARK(5);
    mt = next;
  }
  EL_FLUSH();
}

template <int WAVES = AR_WAVES>
__device__ __forceinline__ void embed_arows_dispatch(const EmbedTileArgs &a, float *lds_f, const uint32_t bid, const uint32_t nblocks) {
  const int NBo = (a.net.a_out + 31) >> 5; // wave-uniform
  if (NBo == 1) embed_arows_body<WAVES, 1>(a, lds_f, bid, nblocks);
  else if (NBo == 2) embed_arows_body<WAVES, 2>(a, lds_f, bid, nblocks);
  else embed_arows_body<WAVES, 3>(a, lds_f, bid, nblocks);
}

__global__ __launch_bounds__(AR_BLOCK) void k_embed_arows(EmbedTileArgs a) {
  extern __shared__ __align__(16) float lds_f[];
  embed_arows_dispatch(a, lds_f, blockIdx.x, gridDim.x);
}
// (Round 3 measured the fp32-MFMA form of this body as eight waves of 256 registers against sixteen of 128 with 49 spilled: 86.5
// vs 86.4 us.  Since round 4 -- second layer and transposition on the bf16 pipe, the next input prefetched across the second
// layer -- the kernels run as eight waves: the sixteen-wave form spills 150-400 registers.)

// ---- K2, the party-slot pass in the same form as k_embed_arows (the default).  A bench Pokemon has 6 DENSE features (bias,
// 5 stats: 3 k-steps x 4 blocks = 12 MFMAs in the "lane = item" orientation) and 7 ONE-HOT rows (4 move slots, status, 2
// types; 193 possible rows, all LDS-resident in natural channel order): in k-step t the half-wave hh sums the 7 rows of
// item 2t + hh, a whole 512-byte row per ds_read_b128 and so free of the bank conflicts of k_embed_rows (every lane another
// row at the same column), and selector MFMAs transpose the sums into the item lanes.  Second layer (<= 2 output
// blocks) with W1's bf16 triples from LDS.  The input is read straight from global memory (the encode of a Pokemon is 12
// features; no staging).  LDS: 194 x 512 B rows + 3 KB dense fragment + 48 KB W1 triples + bias + 512 B per wave.
constexpr int PR_SPARSE = 193, PR_ZERO = 193, PR_KT = 1, PR_HOT = 7; // (6 dense features: one k-step of eight)
constexpr int PR_ITEM_WORDS = 8;                 // 7 LDS byte offsets of the item's one-hot rows, ready to use (+ 1 pad: two 16-byte reads)
constexpr int PR_WAVE_WORDS = ER_ITEMS * PR_ITEM_WORDS;
constexpr int PR_WAVES = OAK_EMBED_WAVES, PR_BLOCK = 64 * PR_WAVES;
constexpr int PR_DENSE_WORDS = PR_KT * 4 * 3 * 64 * 2;
constexpr int PR_MAX_NBO = 2;
constexpr int pr_img_words(int nbo) { return (PR_SPARSE + 1) * ER_RS + PR_DENSE_WORDS + nbo * (E2_BLOCK_BYTES / 4) + 64 * nbo; } // (... + b1 + 1 / W1's row scales)
constexpr size_t PR_BYTES = (size_t)(pr_img_words(PR_MAX_NBO) + PR_WAVES * PR_WAVE_WORDS) * 4;
static_assert(PR_BYTES <= 160 * 1024, "the party pass's LDS image fits one CU");
template <bool LIST, int NBO>
__device__ __forceinline__ void embed_prows_body(const EmbedTileArgs &a, float *lds_f, const uint32_t bid, const uint32_t nblocks) {
  EL_T0();
  const NetDev &N = a.net;
  const int out_dim = N.p_out;
  const uint32_t items = LIST ? *a.work_count : a.n * 10;
  const uint32_t nmt = (items + ER_ITEMS - 1) / ER_ITEMS;
  if (bid >= nmt) return; // (a very short work list: no weights staged for nothing)
  float *W0s = lds_f;                                   // rows 5..197 of W0^T, channels in natural order, + a zero row
  float *Wd = W0s + (PR_SPARSE + 1) * ER_RS;            // dense fragment
  const uint8_t *W1t = (const uint8_t *)(Wd + PR_DENSE_WORDS); // second layer's bf16 triples: [block][k-step][h m l][lane] x 16 B
  const float *b1s = (const float *)(W1t + NBO * E2_BLOCK_BYTES);
  constexpr int img_words = pr_img_words(NBO);
  stage_image_dma<PR_BLOCK>(lds_f, N.p_img, img_words);
  const uint32_t lane = threadIdx.x & 63, wib = threadIdx.x >> 6, r32 = lane & 31, hh = lane >> 5;
  uint32_t *wl = (uint32_t *)(lds_f + img_words) + wib * PR_WAVE_WORDS; // this wave's private LDS (the items' row indices)
  const uint32_t stride = nblocks * PR_WAVES;
  EL_MARK(0);
  // ---- encode, in two halves (see k_embed_arows): LOAD asks for the item's raw input -- the side's whole party (36 dwords)
  // together with its order bytes, the right Pokemon selected afterwards: ONE round trip to memory instead of two dependent
  // ones (the five lanes of a side read the same lines); a work-list record in LIST mode -- and COMPUTE, a whole second layer
  // later, turns it into dense values (registers) and row indices (the wave's LDS).  Both lanes (r, 0) and (r, 1) encode item r.
  uint32_t xp[PR_KT][3][2]; // the dense values of this lane's k-half as bf16 triples (dense_split)
  uint32_t my_doff = 0xFFFFFFFFu; // this lane's item: its block of the embedding (both lanes (r, 0), (r, 1) hold item r's) and hp ratio
  float my_hpr = 0.0f;
  uint4 in_pw[LIST ? 2 : 9];
  uint2 in_ow = make_uint2(0, 0);
  uint32_t in_dur = 0;
  auto encode_load = [&](uint32_t mt)
  {
    const uint32_t g = mt * ER_ITEMS + r32;
    if (g < items) {
      if (LIST) {
        const uint32_t *rec = (const uint32_t *)a.work + (size_t)g * 8;
        in_pw[0] = *(const uint4 *)rec; in_pw[1] = *(const uint4 *)(rec + 4);
      } else {
        const uint32_t leaf = g / 10, q = g - leaf * 10, side = q / 5;
        const uint32_t *sb = (const uint32_t *)a.battles + (size_t)leaf * 96 + side * 46;
#pragma unroll
        for (int u = 0; u < (LIST ? 2 : 9); ++u) in_pw[u] = ((const uint4 *)sb)[u];
        in_ow = *(const uint2 *)(sb + 44);
        in_dur = ((const uint32_t *)a.durations)[(size_t)leaf * 2 + side];
      }
    }
  };
  auto encode_compute = [&](uint32_t mt)
  {
    const uint32_t g = mt * ER_ITEMS + r32;
    uint32_t gi = g, pk0 = 0, pk1 = 0, pk2 = 0, pk3 = 0, pk4 = 0, pk5 = 0, sleep = 0;
    if (g < items) {
      if constexpr (LIST) {
        const uint4 r0 = in_pw[0], r1 = in_pw[1];
        gi = r0.x; pk0 = r0.y; pk1 = r0.z; pk2 = r0.w; pk3 = r1.x; pk4 = r1.y; pk5 = r1.z; sleep = r1.w;
      } else {
        const uint32_t leaf = g / 10, q = g - leaf * 10, side = q / 5, slot = 1 + (q - side * 5);
        const uint32_t o0 = in_ow.x, o1 = in_ow.y;
        const uint32_t id = slot < 4 ? (o0 >> (8 * slot)) & 0xFF : (o1 >> (8 * (slot - 4))) & 0xFF;
        sleep = (in_dur >> (3 * slot)) & 7;
        const uint32_t w[36] = {in_pw[0].x, in_pw[0].y, in_pw[0].z, in_pw[0].w, in_pw[1].x, in_pw[1].y, in_pw[1].z, in_pw[1].w, in_pw[2].x, in_pw[2].y, in_pw[2].z, in_pw[2].w,
                                in_pw[3].x, in_pw[3].y, in_pw[3].z, in_pw[3].w, in_pw[4].x, in_pw[4].y, in_pw[4].z, in_pw[4].w, in_pw[5].x, in_pw[5].y, in_pw[5].z, in_pw[5].w,
                                in_pw[6].x, in_pw[6].y, in_pw[6].z, in_pw[6].w, in_pw[7].x, in_pw[7].y, in_pw[7].z, in_pw[7].w, in_pw[8].x, in_pw[8].y, in_pw[8].z, in_pw[8].w};
#pragma unroll
        for (uint32_t k = 0; k < 6; ++k) {
          const bool m = id == k + 1;
          pk0 = m ? w[6 * k + 0] : pk0; pk1 = m ? w[6 * k + 1] : pk1; pk2 = m ? w[6 * k + 2] : pk2;
          pk3 = m ? w[6 * k + 3] : pk3; pk4 = m ? w[6 * k + 4] : pk4; pk5 = m ? w[6 * k + 5] : pk5;
        }
      }
    }
    float fv[6];
    uint32_t hot[8];
    uint32_t doff = 0xFFFFFFFFu, dead_off = 0xFFFFFFFFu;
    float hpr = 0.0f;
#pragma unroll
    for (int f = 0; f < 6; ++f) fv[f] = 0.0f;
#pragma unroll
    for (int k = 0; k < 8; ++k) hot[k] = PR_ZERO * ER_RS / 4; // (row offsets in 16-byte units)
    if (g < items) {
      const uint32_t leaf = gi / 10, q = gi - leaf * 10, side = q / 5, slot = 1 + (q - side * 5);
      const uint32_t hp = pk4 >> 16;
      const uint32_t dd = leaf * N.emb_dim + side * N.side_dim + (1 + N.a_out) + (slot - 1) * (1 + N.p_out);
      if (hp == 0) dead_off = dd; // empty or fainted: zero block (network.h:153-160), kept out of the scatter
      else {
        doff = dd;
        fv[0] = 1.0f; // the bias
#pragma unroll
        for (uint32_t j = 0; j < 12; ++j) { // Encode::Battle::Pokemon (battle.h:197-214)
          uint32_t fidx = 0; float fval = 0.0f;
          const bool v = pokemon_feature(j, pk0, pk1, pk2, pk3, pk4, pk5, sleep, fidx, fval);
          if (j < 5) fv[1 + j] = fval;
          else hot[j - 5] = v ? (fidx - 5) * ER_RS / 4 : PR_ZERO * ER_RS / 4;
        }
        hpr = (float)hp / (float)(pk0 & 0xFFFF);
      }
    }
    float xv[PR_KT][4];
    xv[0][0] = hh ? fv[4] : fv[0]; xv[0][1] = hh ? fv[5] : fv[1]; xv[0][2] = hh ? 0.0f : fv[2]; xv[0][3] = hh ? 0.0f : fv[3];
    dense_split<PR_KT>(xv, xp);
    if (hh == 0) { // (byte offsets from W0s: the readers add their own 16 bytes of the row and nothing else)
      uint32_t *it = wl + r32 * PR_ITEM_WORDS;
      *(uint4 *)it = make_uint4(hot[0] * 16, hot[1] * 16, hot[2] * 16, hot[3] * 16);
      *(uint4 *)(it + 4) = make_uint4(hot[4] * 16, hot[5] * 16, hot[6] * 16, 0);
    }
    my_doff = doff;
    my_hpr = hpr;
    zero_blocks(a.emb, hh == 0 ? dead_off : 0xFFFFFFFFu, out_dim + 1);
  };
  // mini-tiles go round-robin over the WORKGROUPS first, so that a short work list still spreads over every CU
  uint32_t mt = wib * nblocks + bid;
  if (mt < nmt) { encode_load(mt); encode_compute(mt); } // needs no weights: the image's round trip runs under it
  stage_image_wait(); // weights staged (the only workgroup barrier of the kernel)
  while (mt < nmt) {
    __builtin_amdgcn_wave_barrier();
    EL_MARK(3);
    __builtin_amdgcn_s_setprio(2); // a wave in its MFMA phases goes before waves that encode or scatter (see the second layer)
    // ---- first layer: dense part, then the one-hot rows through the selector transposition (see k_embed_arows) ----
    f32x16 hb[4];
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int q = 0; q < 16; ++q) hb[b][q] = 0.0f;
    dense_layer_bf16<PR_KT>((const uint8_t *)Wd + lane * 8, xp, hb);
    // The transposition runs on the bf16 pipe (SelTranspose): the 64 identity MFMAs of the fp32 form held the SIMD's vector issue
    // for 64 cycles each.
    // Software pipeline over the 16 k-steps, fully unrolled: an MFMA holds the wave's in-order issue until the matrix pipe takes it
    // (64 cycles each on the fp32 pipe), so k-step t + 1's row reads are issued IN FRONT of k-step t's four MFMAs and their LDS
    // round trip runs under them; the row indices are read two k-steps ahead.
    const uint8_t *w0lane = (const uint8_t *)W0s + 16 * r32;
    auto rows_of = [&](const uint4 &ia, const uint4 &ib, float4 (&r)[PR_HOT]) {
      r[0] = *(const float4 *)(w0lane + ia.x); r[1] = *(const float4 *)(w0lane + ia.y); r[2] = *(const float4 *)(w0lane + ia.z);
      r[3] = *(const float4 *)(w0lane + ia.w); r[4] = *(const float4 *)(w0lane + ib.x); r[5] = *(const float4 *)(w0lane + ib.y);
      r[6] = *(const float4 *)(w0lane + ib.z);
    };
    {
      uint4 ian = *(const uint4 *)(wl + (2 * 1 + hh) * PR_ITEM_WORDS), ibn = *(const uint4 *)(wl + (2 * 1 + hh) * PR_ITEM_WORDS + 4);
      float4 rc[PR_HOT], rn[PR_HOT];
      SelTranspose tr;
      rows_of(*(const uint4 *)(wl + hh * PR_ITEM_WORDS), *(const uint4 *)(wl + hh * PR_ITEM_WORDS + 4), rc);
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        if (t + 1 < 16) rows_of(ian, ibn, rn);
        if (t + 2 < 16) { ian = *(const uint4 *)(wl + (2 * (t + 2) + hh) * PR_ITEM_WORDS); ibn = *(const uint4 *)(wl + (2 * (t + 2) + hh) * PR_ITEM_WORDS + 4); }
        float4 sum;
        sum.x = ((rc[0].x + rc[1].x) + (rc[2].x + rc[3].x)) + ((rc[4].x + rc[5].x) + rc[6].x);
        sum.y = ((rc[0].y + rc[1].y) + (rc[2].y + rc[3].y)) + ((rc[4].y + rc[5].y) + rc[6].y);
        sum.z = ((rc[0].z + rc[1].z) + (rc[2].z + rc[3].z)) + ((rc[4].z + rc[5].z) + rc[6].z);
        sum.w = ((rc[0].w + rc[1].w) + (rc[2].w + rc[3].w)) + ((rc[4].w + rc[5].w) + rc[6].w);
#ifdef OAK_EMBED_FP32_TRANSPOSE
        const float ident = (uint32_t)(2 * t) + hh == r32 ? 1.0f : 0.0f;
        hb[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(sum.x, ident, hb[0], 0, 0, 0);
        hb[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(sum.y, ident, hb[1], 0, 0, 0);
        hb[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(sum.z, ident, hb[2], 0, 0, 0);
        hb[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(sum.w, ident, hb[3], 0, 0, 0);
#else
        tr.step(t, sum, hb, r32, hh);
#endif
#pragma unroll
        for (int k = 0; k < PR_HOT; ++k) rc[k] = rn[k];
      }
    }
    act_blocks<4>(hb, N.activation);
    EL_MARK(4);
    const uint32_t next = mt + stride;
    if (next < nmt) encode_load(next); // the next mini-tile's raw input: asked for now, looked at after the second layer
    // ---- second layer on the bf16 pipe (embed_layer2) ----
    // wave priority: second layer 3 > first layer 2 > encode / scatter 0 -- whoever can feed the matrix pipe issues first
    __builtin_amdgcn_s_setprio(3);
    f32x16 acc[NBO];
    const float inv_item = embed_layer2<NBO>(hb, W1t + lane * 16, NBO, acc);
    EL_MARK(6);
    __builtin_amdgcn_s_setprio(0);
    const uint32_t cur_doff = my_doff;
    const float cur_hpr = my_hpr;
    __builtin_amdgcn_wave_barrier(); // the wave's row indices are rewritten by the next mini-tile's encode
    // the next encode BEFORE this mini-tile's stores: its loads are then the oldest vector-memory operations in flight and the
    // wait for them does not wait for the stores (vmcnt retires in order)
    if (next < nmt) encode_compute(next);
    EL_MARK(7);
    embed_scatter<NBO>(a.emb, cur_doff, cur_hpr, hh, acc, NBO, out_dim, b1s, N.activation, inv_item);
    EL_MARK(5);
    mt = next;
  }
  EL_FLUSH();
}

template <bool LIST>
__device__ __forceinline__ void embed_prows_dispatch(const EmbedTileArgs &a, float *lds_f, const uint32_t bid, const uint32_t nblocks) {
  if (a.net.p_out <= 32) embed_prows_body<LIST, 1>(a, lds_f, bid, nblocks); // wave-uniform
  else embed_prows_body<LIST, 2>(a, lds_f, bid, nblocks);
}

template <bool LIST>
__global__ __launch_bounds__(PR_BLOCK) void k_embed_prows(EmbedTileArgs a) {
  extern __shared__ __align__(16) float lds_f[];
  embed_prows_dispatch<LIST>(a, lds_f, blockIdx.x, gridDim.x);
}

// Both embedding passes in ONE launch (the default): workgroups [0, np) run the party-slot pass, the rest the actives'
// pass.  A CU holds one of these workgroups at a time (LDS), so the actives' workgroups start where the party pass's finish:
// one kernel boundary less, and the slow tail of the first pass (the last waves of 256 CUs finishing one by one) is filled
// by the second instead of standing between two launches.
static_assert(PR_BLOCK == AR_BLOCK, "the merged launch uses one workgroup size");
template <bool LIST>
__global__ __launch_bounds__(PR_BLOCK) void k_embed_both(EmbedTileArgs party, EmbedTileArgs actives, uint32_t np) {
  extern __shared__ __align__(16) float lds_f[];
  if (blockIdx.x < np) embed_prows_dispatch<LIST>(party, lds_f, blockIdx.x, np);
  else embed_arows_dispatch(actives, lds_f, blockIdx.x - np, gridDim.x - np);
}

// ---- party-slot embedding cache (the GPU form of NN::Battle::PokemonCache, cpp/include/nn/battle/cache.h:18-131) ------------
// A bench Pokemon's embedding depends only on its stored bytes minus hp, with the PP bytes reduced to "has PP" bits and the
// status reduced to its encoder index (Encode::Battle::pokemon_key, encode/battle/key.h:65-71) -- the reference fills 240
// embeddings per Pokemon once per search and looks them up.  For a RESIDENT batch that is evaluated again and again
// (BASELINE configs[2]: a leaf evaluation of every lane every turn) the analogue is: the caller keeps the batch's embedding
// buffer and a TAG per (leaf, bench slot) -- the canonicalised 24 stored bytes, exact, no hashing -- and only the slots
// whose tag changed since the last call are recomputed.  k_party_tags does the comparison for all n x 10 slots (one lane
// each), refreshes the slot's hp ratio (hp is not part of the embedding input, network.h:153-160), zeroes the block of a
// slot that became empty / fainted and appends the changed live slots -- with their Pokemon bytes -- to a work list that
// k_embed_lds<party, LIST> then embeds.
constexpr uint32_t TAG_DEAD = 0xFFFFFFFEu, TAG_WORDS = 6;
constexpr int TAG_R = 8; // slots per lane: ONE atomic on the work counter per 2048 slots (one per wave serialised 10,000 of them)
__global__ __launch_bounds__(256) void k_party_tags(NetDev N, const uint8_t *battles, const uint8_t *durations, uint32_t n, float *emb,
                                                    uint32_t *tags, PartyWork *work, uint32_t *work_count) {
  __shared__ uint32_t cnt[TAG_R * 4 + 1];
  const uint32_t lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
  uint32_t pk[TAG_R][6], sleep[TAG_R];
  uint32_t changed = 0; // bit r: the slot of round r is live and its tag changed
  // Three passes over the lane's TAG_R slots instead of one slot after the other: all order bytes / durations / old tags are
  // asked for together, then all the Pokemon they point to, then the comparisons -- two round trips to memory per lane
  // instead of ~3 per slot (the kernel was a chain of 24 dependent latencies: 40 us for 52 MB).
  const uint32_t n_items = n * 10;
  uint32_t o0[TAG_R], o1[TAG_R], dur[TAG_R], old[TAG_R][TAG_WORDS];
#pragma unroll
  for (int r = 0; r < TAG_R; ++r) {
    const uint32_t it = (blockIdx.x * TAG_R + r) * 256 + threadIdx.x, item = it < n_items ? it : n_items - 1; // (clamped: loads stay unconditional)
    const uint32_t leaf = item / 10, q = item - leaf * 10, side = q / 5;
    const uint32_t *sb = (const uint32_t *)battles + (size_t)leaf * 96 + side * 46;
    const uint2 ow = *(const uint2 *)(sb + 44);
    o0[r] = ow.x; o1[r] = ow.y;
    dur[r] = ((const uint32_t *)durations)[(size_t)leaf * 2 + side];
    const uint2 *t = (const uint2 *)(tags + (size_t)item * TAG_WORDS);
    const uint2 t0 = t[0], t1 = t[1], t2 = t[2];
    old[r][0] = t0.x; old[r][1] = t0.y; old[r][2] = t1.x; old[r][3] = t1.y; old[r][4] = t2.x; old[r][5] = t2.y;
  }
#pragma unroll
  for (int r = 0; r < TAG_R; ++r) {
    const uint32_t it = (blockIdx.x * TAG_R + r) * 256 + threadIdx.x, item = it < n_items ? it : n_items - 1;
    const uint32_t leaf = item / 10, q = item - leaf * 10, side = q / 5, slot = 1 + (q - side * 5);
    const uint32_t id = slot < 4 ? (o0[r] >> (8 * slot)) & 0xFF : (o1[r] >> (8 * (slot - 4))) & 0xFF;
    const uint2 *p = (const uint2 *)((const uint32_t *)battles + (size_t)leaf * 96 + side * 46 + 6 * (id ? id - 1 : 0));
    const uint2 p0 = p[0], p1 = p[1], p2 = p[2];
    pk[r][0] = id ? p0.x : 0; pk[r][1] = id ? p0.y : 0; pk[r][2] = id ? p1.x : 0; pk[r][3] = id ? p1.y : 0; pk[r][4] = id ? p2.x : 0; pk[r][5] = id ? p2.y : 0;
    sleep[r] = (dur[r] >> (3 * slot)) & 7;
  }
#pragma unroll
  for (int r = 0; r < TAG_R; ++r) {
    const uint32_t item = (blockIdx.x * TAG_R + r) * 256 + threadIdx.x;
    bool live_change = false;
    if (item < n_items) {
      const uint32_t leaf = item / 10, q = item - leaf * 10, side = q / 5, slot = 1 + (q - side * 5);
      const uint32_t doff = leaf * N.emb_dim + side * N.side_dim + (1 + N.a_out) + (slot - 1) * (1 + N.p_out);
      const uint32_t hp = pk[r][4] >> 16;
      uint32_t c[TAG_WORDS];
      if (hp == 0) {
#pragma unroll
        for (uint32_t k = 0; k < TAG_WORDS; ++k) c[k] = TAG_DEAD;
      } else { // canonical form: everything Encode::Battle::Pokemon::write reads (battle.h:197-214), nothing else
        const uint32_t st = pk[r][5] & 0xFF;
        const uint32_t skey = st ? status_index(st, sleep[r]) + 1 : 0;
        c[0] = pk[r][0];                                                                   // hp max | atk
        c[1] = pk[r][1];                                                                   // def | spe
        c[2] = (pk[r][2] & 0x00FFFFFFu) | ((pk[r][2] >> 24) ? 1u << 24 : 0u);                // spc | move 1 id | has-pp 1
        c[3] = (pk[r][3] & 0x00FF00FFu) | (((pk[r][3] >> 8) & 0xFF) ? 1u << 8 : 0u) | ((pk[r][3] >> 24) ? 1u << 24 : 0u); // ids 2, 3 | has-pp 2, 3
        c[4] = (pk[r][4] & 0xFFu) | (((pk[r][4] >> 8) & 0xFF) ? 1u << 8 : 0u) | (skey << 16); // id 4 | has-pp 4 | status index + 1
        c[5] = pk[r][5] & 0xFFFFFF00u;                                                     // species | types | level
      }
      uint32_t *t = tags + (size_t)item * TAG_WORDS;
      bool same = true;
#pragma unroll
      for (uint32_t k = 0; k < TAG_WORDS; ++k) same = same && old[r][k] == c[k];
      if (!same) {
#pragma unroll
        for (uint32_t k = 0; k < TAG_WORDS; ++k) t[k] = c[k];
        if (hp == 0) { for (int o = 0; o <= N.p_out; ++o) emb[(size_t)doff + o] = 0.0f; } // network.h:153-160: empty / fainted slot
        else live_change = true;
      }
      if (hp != 0) emb[doff] = (float)hp / (float)(pk[r][0] & 0xFFFF); // the hp ratio changes without the tag changing
    }
    const uint64_t mask = __ballot(live_change);
    if (lane == 0) cnt[r * 4 + wib] = (uint32_t)__popcll(mask);
    changed |= live_change ? 1u << r : 0u;
  }
  __syncthreads();
  if (threadIdx.x == 0) { // exclusive prefix over the block's (round, wave) counts + ONE atomic for the block
    uint32_t run = 0;
    for (int k = 0; k < TAG_R * 4; ++k) { const uint32_t c = cnt[k]; cnt[k] = run; run += c; }
    cnt[TAG_R * 4] = run ? atomicAdd(work_count, run) : 0u;
  }
  __syncthreads();
  const uint32_t base = cnt[TAG_R * 4];
#pragma unroll
  for (int r = 0; r < TAG_R; ++r) {
    const bool mine = (changed >> r) & 1;
    const uint64_t mask = __ballot(mine);
    if (mine) {
      const uint32_t item = (blockIdx.x * TAG_R + r) * 256 + threadIdx.x;
      PartyWork w{item, {pk[r][0], pk[r][1], pk[r][2], pk[r][3], pk[r][4], pk[r][5]}, sleep[r]};
      work[base + cnt[r * 4 + wib] + (uint32_t)__popcll(mask & ((1ull << lane) - 1))] = w;
    }
  }
}

// ---- K3: main net on fp32 MFMA ------------------------------------------------------------------
constexpr int MN_BLOCK = 256; // 4 waves
constexpr int MAXH = 256;

struct MainArgs {
  NetDev net;
  const float *emb; // n x emb_dim
  uint32_t n;
  float *values;
  float *h1_out; // nullable: n x H activated fc1 outputs, input of the policy heads (k_policy)
};

// ---- K3, wave-independent form (the default).  k_mainnet_direct runs one wave per SIMD (its double-buffered weight
// fragments take the whole register file), so nothing overlaps with anything: every workgroup barrier (fc0's three
// 256-column input pieces, the layer hand-overs), every piece's trip through registers into LDS and every wait for the
// slowest of the four waves is time the matrix pipe stands still (MFMA busy 0.71).  Here a WAVE owns 32 leaves and all (up
// to 8) 32-wide output blocks of a layer: fc0's A operands come straight from the embedding rows in global memory (lane
// (r, h) reads 32 B of row r per 8 k-steps, prefetched one sub-chunk ahead like the weights), the hand-over between layers
// goes through the wave's PRIVATE 32 x 257 LDS tile (C layout -> A layout) and needs no workgroup barrier at all.  Weight
// fragments as in k_mainnet_direct but ordered by sub-chunk (frag_order_wave), read 8 k-steps at a time: 2 x 64 registers
// of B fragments + 128 accumulators.  L2 traffic is unchanged (there, the two waves of a row-half pair read the same fragments).
constexpr int MW_LD = MAXH + 1;
constexpr size_t MW_BYTES = (size_t)(4 * 32 * MW_LD + 4 * MAXH) * 4; // four wave tiles + the bias / value_fc3 vectors

__device__ __forceinline__ float f4_pick(const float4 &v, int i) { return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w; }

template <int NBc, bool A_GLOBAL>
__device__ __forceinline__ void wave_layer(const float4 *Wf, int K, int NB, const float *arow, f32x16 (&acc)[8]) {
  const int lane = threadIdx.x & 63, h = lane >> 5;
  const int nch = (K + 63) / 64;
  float4 bA[NBc][2], bB[NBc][2]; // B fragments of one sub-chunk (8 k-steps), double-buffered
  float4 aC[8], aN[8];           // the lane's 32 A values of the current / the next chunk (fc0: one 128-byte line of its row)
#pragma unroll
  for (int j = 0; j < NBc; ++j)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[j][q] = 0.0f;
  // sub-chunk (c, u): k-steps 8u .. 8u+7 of chunk c = float4 q = 2u, 2u+1 of every n-block's fragment
  auto wptr = [&](int t) { return Wf + (size_t)t * NBc * 128; }; // sub-chunk t = 4c + u; wave-uniform: a scalar base
  auto load_a = [&](float4 (&av)[8], int c) {
    const int col = c * 64 + h * 32;
    if (A_GLOBAL) { // columns past K: the address is clamped instead of the load predicated (a predicated load cannot be hoisted)
#pragma unroll
      for (int q = 0; q < 8; ++q) av[q] = *(const float4 *)(arow + (col + 4 * q < K ? col + 4 * q : 0));
    } else {        // the LDS tile row is addressable up to column 256 whatever K is
      const float *p = arow + col;
#pragma unroll
      for (int q = 0; q < 8; ++q) av[q] = make_float4(p[4 * q], p[4 * q + 1], p[4 * q + 2], p[4 * q + 3]);
    }
  };
  // columns past K contribute nothing (K is a multiple of 4; stale tile columns may hold anything).  Applied where the
  // values are first USED, not where they are loaded: a select right behind the load makes the wave wait for it on the spot
  auto mask_a = [&](float4 (&av)[8], int c) {
    const int col = c * 64 + h * 32;
#pragma unroll
    for (int q = 0; q < 8; ++q)
      if (col + 4 * q >= K) av[q] = make_float4(0.f, 0.f, 0.f, 0.f);
  };
  // The MFMAs of one sub-chunk (flat index t, fragments in P; Q holds sub-chunk t + 1).  Every half of a fragment buffer is
  // refilled right after its last use: Q's second halves (k-steps 4..7 of t - 1) in front of step 0 with those of t + 1, P's
  // first halves (k-steps 0..3 of t) in front of step 4 with those of t + 2 -- each load is issued 12 k-steps (6 k cycles of
  // MFMA) before its first use, out of the same 128 registers as plain double-buffering (8 k-steps).  Measured with plain
  // double-buffering (main net, us): two loads per k-step 385, four 366, all sixteen in front of step 0 356-361.
  const int nsc = 4 * nch;
  auto compute = [&](float4 (&P)[NBc][2], float4 (&Q)[NBc][2], int u, int t, int ca) {
    const float4 *w1 = wptr(t + 1 < nsc ? t + 1 : nsc - 1) + 64 + lane; // past the end: the last sub-chunk again, dropped
    const float4 *w2 = wptr(t + 2 < nsc ? t + 2 : nsc - 1) + lane;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      if (s == 0) {
#pragma unroll
        for (int j = 0; j < NBc; ++j) Q[j][1] = w1[j * 128];
      }
      if (s == 4) {
#pragma unroll
        for (int j = 0; j < NBc; ++j) P[j][0] = w2[j * 128];
      }
      // the next chunk's A values, a whole chunk ahead -- issued BEHIND this sub-chunk's first weight loads: the vector-memory
      // counter retires in order, so the next waits for weights would otherwise cover these (HBM latency) too
      if (ca >= 0 && s == 1) load_a(aN, ca);
      const float a_s = f4_pick(aC[2 * u + (s >> 2)], s & 3);
#pragma unroll
      for (int j = 0; j < NBc; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_s, f4_pick(P[j][s >> 2], s & 3), acc[j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  {
    const float4 *w0 = wptr(0) + lane, *w1 = wptr(1) + lane; // (nsc >= 4)
#pragma unroll
    for (int j = 0; j < NBc; ++j) { bA[j][0] = w0[j * 128]; bA[j][1] = w0[j * 128 + 64]; bB[j][0] = w1[j * 128]; }
  }
  load_a(aC, 0);
  mask_a(aC, 0);
#pragma unroll 1
  for (int c = 0; c < nch; ++c) {
    const int cn = c + 1 < nch ? c + 1 : c; // the last chunk prefetches itself again: straight-line code, the loads are dropped
    compute(bA, bB, 0, 4 * c, cn);
    compute(bB, bA, 1, 4 * c + 1, -1);
    compute(bA, bB, 2, 4 * c + 2, -1);
    compute(bB, bA, 3, 4 * c + 3, -1);
    mask_a(aN, cn);
#pragma unroll
    for (int q = 0; q < 8; ++q) aC[q] = aN[q];
  }
}

template <bool A_GLOBAL>
__device__ __forceinline__ void wave_layer_nb(const float *Wf, int K, int NB, const float *arow, f32x16 (&acc)[8]) {
  if (NB > 4) wave_layer<8, A_GLOBAL>((const float4 *)Wf, K, NB, arow, acc);
  else if (NB > 2) wave_layer<4, A_GLOBAL>((const float4 *)Wf, K, NB, arow, acc);
  else if (NB == 2) wave_layer<2, A_GLOBAL>((const float4 *)Wf, K, NB, arow, acc);
  else wave_layer<1, A_GLOBAL>((const float4 *)Wf, K, NB, arow, acc);
}

// bias + activation, accumulators (C layout: lane = column, registers = rows) -> the wave's tile (row-major)
__device__ __forceinline__ void wave_store_act(const f32x16 (&acc)[8], const float *bias, int NB, int activation, float *tile) {
  const int lane = threadIdx.x & 63, col = lane & 31, hh = lane >> 5;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    if (j < NB) {
      const int nn = j * 32 + col;
      const float b = bias[nn];
#pragma unroll
      for (int q = 0; q < 16; ++q) tile[((q & 3) + 8 * (q >> 2) + 4 * hh) * MW_LD + nn] = act_fn(acc[j][q] + b, activation);
    }
  }
}

// value_fc2's bias + activation, value_fc3 and the sigmoid (network.h:14,75) straight from the accumulators: lane (col, hh)
// holds column j * 32 + col of 16 rows per block, so its share of a row's dot product is sum_j act(acc[j][q] + b2) * w3, and
// a butterfly over the 32 lanes of its half finishes the rows -- no trip through the tile, no 256-long serial chain
__device__ __forceinline__ void wave_value_head(const f32x16 (&acc)[8], const float *b2, const float *w3, float b3, int NB, int activation,
                                                float *values, uint32_t row0, uint32_t n_rows) {
  const int lane = threadIdx.x & 63, col = lane & 31, hh = lane >> 5;
  float part[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) part[q] = 0.0f;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    if (j < NB) {
      const float b = b2[j * 32 + col], w = w3[j * 32 + col];
#pragma unroll
      for (int q = 0; q < 16; ++q) part[q] = fmaf(act_fn(acc[j][q] + b, activation), w, part[q]);
    }
  }
#pragma unroll
  for (int off = 1; off < 32; off <<= 1)
#pragma unroll
    for (int q = 0; q < 16; ++q) part[q] += __shfl_xor(part[q], off, 64);
  float y = part[0];
#pragma unroll
  for (int q = 1; q < 16; ++q) y = (col & 15) == q ? part[q] : y;
  const uint32_t row = (uint32_t)((col & 3) + 8 * ((col & 15) >> 2) + 4 * hh);
  if (col < 16 && row < n_rows) values[row0 + row] = 1.0f / (1.0f + expf(-(y + b3)));
}

__global__ __launch_bounds__(MN_BLOCK) void k_mainnet_wave(MainArgs a) {
  extern __shared__ __align__(16) float lds_f[];
  const NetDev &N = a.net;
  const int H = N.H, VH = N.VH;
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
  float *tile = lds_f + wave * 32 * MW_LD;
  const float *trow = tile + r * MW_LD;
  // the three bias vectors and value_fc3's weights, once per workgroup: read from LDS at the layer hand-overs instead of from
  // global memory in front of them (a round trip the lone wave of a SIMD cannot hide)
  float *vec = lds_f + 4 * 32 * MW_LD; // [b0 | b1 | b2 | w3], MAXH each
  for (int i = threadIdx.x; i < MAXH; i += MN_BLOCK) {
    vec[i] = i < H ? N.b0[i] : 0.0f; vec[MAXH + i] = i < H ? N.b1[i] : 0.0f;
    vec[2 * MAXH + i] = i < VH ? N.b2[i] : 0.0f; vec[3 * MAXH + i] = i < VH ? N.w3[i] : 0.0f;
  }
  __syncthreads();
  const uint32_t ntiles = (a.n + 31) / 32;
  for (uint32_t wt = blockIdx.x * 4 + wave; wt < ntiles; wt += gridDim.x * 4) {
    const uint32_t row0 = wt * 32;
    const uint32_t n_rows = min(32u, a.n - row0);
    const uint32_t grow = row0 + (r < n_rows ? r : n_rows - 1); // rows past the batch repeat the last one and are dropped below
    f32x16 acc[8];
    MN_T0();
    wave_layer_nb<true>(N.w0g, N.emb_dim, H / 32, a.emb + (size_t)grow * N.emb_dim, acc);
    MN_MARK(10);
    __builtin_amdgcn_wave_barrier(); // (the previous tile's value_fc2 reads of the tile come first)
    wave_store_act(acc, vec, H / 32, N.activation, tile);
    __builtin_amdgcn_wave_barrier();
    MN_MARK(11);
    wave_layer_nb<false>(N.w1g, H, H / 32, trow, acc);
    MN_MARK(12);
    __builtin_amdgcn_wave_barrier();
    wave_store_act(acc, vec + MAXH, H / 32, N.activation, tile);
    __builtin_amdgcn_wave_barrier();
    if (a.h1_out) { // keep fc1's activations for the policy heads
      for (uint32_t row = 0; row < n_rows; ++row)
        for (uint32_t c = lane; c < (uint32_t)H; c += 64) a.h1_out[(size_t)(row0 + row) * H + c] = tile[row * MW_LD + c];
    }
    MN_MARK(13);
    wave_layer_nb<false>(N.w2g, H, VH / 32, trow, acc);
    MN_MARK(14);
    wave_value_head(acc, vec + 2 * MAXH, vec + 3 * MAXH, N.b3, VH / 32, N.activation, a.values, row0, n_rows);
    MN_MARK(15);
  }
}

// ---- K3': the main net on the bf16 matrix pipe, fp32 values carried as bf16 TRIPLES --------------------------------------
// An fp32 value x is x = h + m + l exactly, with h = bf16(x), m = bf16(x - h), l = bf16(x - h - m) (each remainder is exact in
// fp32 and the last one fits bf16's 8 significant bits).  A product x * w is then the sum of nine bf16 x bf16 products, each of
// which is EXACT in the matrix pipe's fp32 accumulator; the kernel issues the six largest (hh, hm, mh, mm, hl, lh) and drops
// ml + lm + ll <= 3 * 2^-25 |x w| -- less than the rounding error of the one fp32 multiply it replaces.  Sums accumulate in
// fp32 exactly as before.  Why: v_mfma_f32_32x32x16_bf16 retires 32 x 32 x 16 multiply-adds in 32 cycles, the fp32 form
// (32x32x2) needs 8 x 64 cycles for the same block -- six bf16 MFMAs are 192 cycles against 512 (MI355X_MICROARCH.md, cycle
// constants).  The result is an fp32 result: tests/test_gpu_leafnet.py holds it to the same 1e-5 against the fp32 oracle and
// measures its error against a float64 evaluation next to k_mainnet_wave's.
//
// Orientation: Out^T = W . In^T, i.e. the WEIGHTS are the A operand (rows = output features) and the batch rows sit on the
// MFMA's columns = lanes.  A layer's result then has its features in the 16 accumulator registers and the batch row on the
// lane -- which is exactly the B-operand layout of the next layer summing over those features (cdna_hip_programming.md, "an
// accumulator tile as the next MFMA's operand"): activations never leave the registers, there is no LDS tile and no
// transposition.  The k-order this implies (k-step t of fc1 / value_fc2 = registers 8 (t & 1) .. + 7 of block t >> 1 =
// features 32 (t >> 1) + (reg & 3) + 8 (reg >> 2) + 4 h) is baked into the weight stream on the host (split_stream).
//
// One workgroup = 4 waves = 4 x 32 batch rows sharing ONE pass over the weight stream through a two-buffer LDS ring (a phase
// = MS_G k-steps = NB x 6 KB): while the phase-p MFMAs run, phase p + 1 arrives in the other buffer by LDS-DMA (every wave
// issues its 1-KB pieces at the start of phase p), and the waves meet at one barrier per phase.  L2 -> CU weight traffic is
// 1 / 4 of k_mainnet_wave's per row although the triples are 1.5 x the bytes.
//
// Where it stands (profiles/r03_mainnet_split.json): 187 us per 65,536 leaves against k_mainnet_wave's 336; the six-fold MFMA
// count keeps the matrix pipe 60-62 % busy, the level the best bf16 GEMMs reach on this chip under its power limit.  Deeper
// rings, hand-counted vmcnt waits, interleaved accumulation chains, one LDS read per MFMA pair and a level-wise split all
// measured 186-189 us; only removing work moved it (no split: 132 us, no LDS reads: 169 us) -- so the simplest form is kept.
constexpr int MS_G = 2;
template <int NB> struct MSplit {
  static constexpr int PHASE_BYTES = MS_G * NB * 3072;            // [k-step][block][h m l][lane] x 16 B
  static constexpr int PT = (PHASE_BYTES + 4095) / 4096;          // 1-KB DMA pieces per wave per phase, at most
  static constexpr size_t LDS = 2 * (size_t)PHASE_BYTES + 4 * MAXH * 4;
};

typedef __attribute__((address_space(3))) void ms_lds_void;
typedef __attribute__((address_space(1))) const void ms_glb_void;
// One phase of the stream into one ring buffer by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write): a wave
// instruction lands 64 x 16 B = 1 KB at a wave-uniform LDS base, lane-linear -- the stream IS the LDS image.
template <int NB>
__device__ __forceinline__ void ms_dma_phase(const uint8_t *src, uint8_t *dst, int tid) {
  using M = MSplit<NB>;
  const int wave = tid >> 6;
#pragma unroll
  for (int i = 0; i < M::PT; ++i) {
    const int piece = (i * MN_BLOCK + wave * 64) * 16; // wave-uniform; PHASE_BYTES is a multiple of 1 KB
    if (M::PHASE_BYTES % 4096 == 0 || piece < M::PHASE_BYTES)
      __builtin_amdgcn_global_load_lds((ms_glb_void *)(src + (i * MN_BLOCK + tid) * 16), (ms_lds_void *)(dst + piece), 16, 0, 0);
  }
}
// The ring of one workgroup: `cur` = the buffer the phase being computed is read from, `phase` = its position in the stream
struct MSRing { int cur, phase, n_phases; const uint8_t *stream; uint8_t *lds; };
// End of a phase.  The next phase's bytes were put in flight when THIS phase started (into the other buffer, which every wave
// had finished reading at the barrier before that); the barrier's own vmcnt(0) retires this wave's pieces, the barrier the
// others'.  Then the phase after next goes in flight into the buffer just finished with.  (The wave's own LDS reads, which the
// compiler does not see, are retired by ms_wait_a<0> in front of every call.)
template <int NB>
__device__ __forceinline__ void ms_next_phase(MSRing &R, int tid) {
  using M = MSplit<NB>;
  // The wait is written out: __syncthreads() does NOT reliably wait for LDS-DMA in flight.  hipcc (ROCm 7.2) emitted
  // `s_waitcnt vmcnt(0)` in front of every second barrier of the k-loops only (k_mainnet_split<2>: three of six barriers were
  // a bare `s_waitcnt lgkmcnt(0); s_barrier`), so a phase could start on a buffer whose bytes were still on their way.  With
  // 8 blocks a phase is 192 MFMAs long and the DMA always won; with 1-2 blocks (64-wide nets) it is 24-48 MFMAs and about one
  // run of the leaf tests in twelve read stale weights (round 4: values off by 1e-7 .. 1e-2 in a few rows, never reproducible
  // in isolation).  This wave's pieces are retired here, the other waves' by the barrier.
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const int done = R.cur;
  R.cur ^= 1;
  R.phase = R.phase + 1 == R.n_phases ? 0 : R.phase + 1;
  const int nx = R.phase + 1 == R.n_phases ? 0 : R.phase + 1;
  ms_dma_phase<NB>(R.stream + (size_t)nx * M::PHASE_BYTES, R.lds + done * M::PHASE_BYTES, tid);
}
// The weight triple (h, m, l) of the block at LDS byte address `addr` + OFF (this lane's 16 bytes of each 1-KB piece).  Written
// as the instructions themselves because the waits are: hipcc's own lgkmcnt for these reads came out as lgkmcnt(0) in every other
// block -- right behind the NEXT block's reads, i.e. a full LDS round trip exposed per block on a SIMD with one wave (226 us
// against 189).  The compiler does not know these are in flight; ms_wait_a is the wait, and it names the registers so that no
// MFMA moves above it.
template <int OFF>
__device__ __forceinline__ void ms_load_a(bf16x8 (&A)[3], uint32_t addr) {
  static_assert(OFF >= 0 && OFF + 2048 < 65536, "ds_read offset field");
  asm volatile("ds_read_b128 %0, %3 offset:%4\n\tds_read_b128 %1, %3 offset:%5\n\tds_read_b128 %2, %3 offset:%6"
               : "=&v"(A[0]), "=&v"(A[1]), "=&v"(A[2])
               : "v"(addr), "n"(OFF), "n"(OFF + 1024), "n"(OFF + 2048));
}
template <int YOUNGER> // wait for a triple with YOUNGER LDS reads (0 or 3) issued behind it
__device__ __forceinline__ void ms_wait_a(bf16x8 (&A)[3]) {
  if (YOUNGER == 3) asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(A[0]), "+v"(A[1]), "+v"(A[2]));
  else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(A[0]), "+v"(A[1]), "+v"(A[2]));
}
__device__ __forceinline__ uint32_t ms_lds_addr(const uint8_t *p) { return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const uint8_t *)p; }
// elements [e0, e1) of a k-step's 8 activation values as their triples
__device__ __forceinline__ void ms_split_part(const float (&v)[8], bf16x8 (&B)[3], int e0, int e1) {
#pragma unroll
  for (int i = 0; i < 8; ++i)
    if (i >= e0 && i < e1) {
      // by truncation (round 4; see e_split): x = h + m + l EXACTLY, and two vector instructions fewer per value than rounding
      const uint32_t hb = __float_as_uint(v[i]) & 0xFFFF0000u;
      const float r1 = v[i] - __uint_as_float(hb);
      const uint32_t mb = __float_as_uint(r1) & 0xFFFF0000u;
      const uint32_t lb = __float_as_uint(r1 - __uint_as_float(mb));
      B[0][i] = __builtin_bit_cast(__bf16, (uint16_t)(hb >> 16));
      B[1][i] = __builtin_bit_cast(__bf16, (uint16_t)(mb >> 16));
      B[2][i] = __builtin_bit_cast(__bf16, (uint16_t)(lb >> 16));
    }
}
// One k-step: the activation triple B against the NB weight blocks at LDS address `base`.  Software pipeline, one wave per
// SIMD: block nb + 1's weight triple is read from LDS in front of block nb's six MFMAs (192 cycles cover the read), the NEXT
// k-step's activation values are split into their triple one share per block in the MFMAs' shadow, and the next k-step's
// first weight triple is read in front of the last block when that k-step lies in the same phase (SAME_PHASE: it starts
// NB x 3 KB further on; otherwise the caller reads it behind the phase barrier).  Afirst: block 0's weights, reads in flight;
// Anext: where the next k-step's block 0 goes when SAME_PHASE.  These are two buffers the callers alternate, never copied:
// a register copy of a triple whose LDS reads are still in flight would read stale registers (the compiler does not know).
template <int NB, bool SAME_PHASE>
__device__ __forceinline__ void ms_kstep(f32x16 (&acc)[NB], const bf16x8 (&B)[3], const float (&vn)[8], bf16x8 (&Bn)[3], bf16x8 (&Afirst)[3],
                                         bf16x8 (&Anext)[3], uint32_t base) {
  bf16x8 A[2][3];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    bf16x8 (&W)[3] = nb == 0 ? Afirst : A[nb & 1];
    // (constant offsets: the template argument must be a constant expression, hence the chain)
#define MS_LOAD_NEXT(NBV)                                                                   \
  if (nb == NBV) {                                                                          \
    if (NBV + 1 < NB) ms_load_a<(NBV + 1) * 3072>(A[(NBV + 1) & 1], base);                   \
    else if (SAME_PHASE) ms_load_a<NB * 3072>(Anext, base);                                  \
  }
    MS_LOAD_NEXT(0) MS_LOAD_NEXT(1) MS_LOAD_NEXT(2) MS_LOAD_NEXT(3) MS_LOAD_NEXT(4) MS_LOAD_NEXT(5) MS_LOAD_NEXT(6) MS_LOAD_NEXT(7)
#undef MS_LOAD_NEXT
    if (nb + 1 < NB || SAME_PHASE) ms_wait_a<3>(W); else ms_wait_a<0>(W);
    ms_split_part(vn, Bn, nb * 8 / NB, (nb + 1) * 8 / NB);
    acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W[2], B[0], acc[nb], 0, 0, 0); // l . h   (small terms first)
    acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W[0], B[2], acc[nb], 0, 0, 0); // h . l
    acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W[1], B[1], acc[nb], 0, 0, 0); // m . m
    acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W[1], B[0], acc[nb], 0, 0, 0); // m . h
    acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W[0], B[1], acc[nb], 0, 0, 0); // h . m
    acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W[0], B[0], acc[nb], 0, 0, 0); // h . h
    // inside the block: the MFMAs with the split's vector instructions in their shadows (an MFMA holds the vector issue for
    // 8 of its 32 cycles)
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}
// bias + activation in place: register q of block nb = feature 32 nb + (q & 3) + 8 (q >> 2) + 4 h
template <int NB>
__device__ __forceinline__ void ms_bias_act(f32x16 (&acc)[NB], const float *bias, int h, int activation) {
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 b = *(const float4 *)(bias + 32 * nb + 8 * g + 4 * h);
      acc[nb][4 * g + 0] = act_fn(acc[nb][4 * g + 0] + b.x, activation);
      acc[nb][4 * g + 1] = act_fn(acc[nb][4 * g + 1] + b.y, activation);
      acc[nb][4 * g + 2] = act_fn(acc[nb][4 * g + 2] + b.z, activation);
      acc[nb][4 * g + 3] = act_fn(acc[nb][4 * g + 3] + b.w, activation);
    }
}
// a layer whose input is the previous layer's accumulators (k-step t = registers 8 (t & 1) .. + 7 of block t >> 1)
template <int NB>
__device__ __forceinline__ void ms_reg_layer(f32x16 (&acc)[NB], const f32x16 (&in)[NB], MSRing &R, int tid) {
  using M = MSplit<NB>;
  const int lane16 = (tid & 63) * 16;
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[nb][q] = 0.0f;
  bf16x8 B[3], Bn[3], A0[3], A1[3]; // A0 / A1: block 0's weights of the first / second k-step of a phase
  {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = in[0][j];
    ms_split_part(v, B, 0, 8);
  }
  ms_load_a<0>(A0, ms_lds_addr(R.lds + R.cur * M::PHASE_BYTES + lane16));
  static_assert(MS_G == 2, "k-steps alternate: first of a phase, last of a phase");
#pragma unroll
  for (int t = 0; t < 2 * NB; ++t) {
    float vn[8];
    const int tn = t + 1 < 2 * NB ? t + 1 : t;
#pragma unroll
    for (int j = 0; j < 8; ++j) vn[j] = in[tn >> 1][8 * (tn & 1) + j];
    const uint32_t base = ms_lds_addr(R.lds + R.cur * M::PHASE_BYTES + (t % MS_G) * NB * 3072 + lane16);
    if (t % MS_G == 0) ms_kstep<NB, true>(acc, B, vn, Bn, A0, A1, base);
    else ms_kstep<NB, false>(acc, B, vn, Bn, A1, A0, base);
    B[0] = Bn[0]; B[1] = Bn[1]; B[2] = Bn[2];
    if (t % MS_G == MS_G - 1) {
      ms_next_phase<NB>(R, tid);
      ms_load_a<0>(A0, ms_lds_addr(R.lds + R.cur * M::PHASE_BYTES + lane16)); // (behind the last k-step: the next layer's first weights)
    }
  }
  ms_wait_a<0>(A0); // nothing of this layer's reads stays in flight past it (the next layer reads A0 again)
}

template <int NB>
__global__ __launch_bounds__(MN_BLOCK) void k_mainnet_split(MainArgs a) {
  extern __shared__ __align__(16) uint8_t lds_b[];
  using M = MSplit<NB>;
  const NetDev &N = a.net;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
  float *vec = (float *)(lds_b + 2 * M::PHASE_BYTES); // [b0 | b1 | b2 | w3], MAXH each
  for (int i = tid; i < MAXH; i += MN_BLOCK) {
    vec[i] = i < N.H ? N.b0[i] : 0.0f; vec[MAXH + i] = i < N.H ? N.b1[i] : 0.0f;
    vec[2 * MAXH + i] = i < N.VH ? N.b2[i] : 0.0f; vec[3 * MAXH + i] = i < N.VH ? N.w3[i] : 0.0f;
  }
  const int T0 = N.ws_T0, K = N.emb_dim;
  MSRing R{0, 0, (T0 + 4 * NB) / MS_G, (const uint8_t *)N.ws, lds_b}; // fc0: T0 k-steps, fc1 and value_fc2: 2 NB each
  ms_dma_phase<NB>(R.stream, lds_b, tid);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // (written out: see ms_next_phase)
  __syncthreads(); // (also publishes vec)
  ms_dma_phase<NB>(R.stream + (size_t)M::PHASE_BYTES, lds_b + M::PHASE_BYTES, tid); // (a stream has at least 4 phases)

  f32x16 X[NB], Y[NB];
  const uint32_t ngroups = (a.n + 127) / 128;
  for (uint32_t grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    const uint32_t row0 = grp * 128 + wave * 32;
    const uint32_t n_rows = row0 < a.n ? min(32u, a.n - row0) : 0u;
    const uint32_t grow = n_rows ? row0 + ((uint32_t)r < n_rows ? (uint32_t)r : n_rows - 1) : a.n - 1; // rows past the batch repeat a real one
    const float *arow = a.emb + (size_t)grow * K;
    // ---- fc0: the lane's row of the embedding, 64 columns (4 k-steps) at a time, one chunk ahead ----
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int q = 0; q < 16; ++q) X[nb][q] = 0.0f;
    float4 aC[8], aN[8];
    // k-step u of chunk c: columns 64 c + 16 u + 8 h .. + 7; past K: a clamped address (its weights are zero)
#define MS_LOAD_A(av, c)                                                                                         \
  _Pragma("unroll") for (int u_ = 0; u_ < 4; ++u_) _Pragma("unroll") for (int qq_ = 0; qq_ < 2; ++qq_) {          \
    const int col_ = 64 * (c) + 16 * u_ + 8 * h + 4 * qq_;                                                       \
    av[2 * u_ + qq_] = *(const float4 *)(arow + (col_ < K ? col_ : 0));                                          \
  }
    const int nch = T0 / 4;
    MS_T0();
    MS_LOAD_A(aC, 0);
    bf16x8 B[3], Bn[3], A0[3], A1[3];
    {
      const float v[8] = {aC[0].x, aC[0].y, aC[0].z, aC[0].w, aC[1].x, aC[1].y, aC[1].z, aC[1].w};
      ms_split_part(v, B, 0, 8);
    }
    ms_load_a<0>(A0, ms_lds_addr(R.lds + R.cur * M::PHASE_BYTES + lane * 16));
#pragma unroll 1
    for (int c = 0; c < nch; ++c) {
      const int cn = c + 1 < nch ? c + 1 : c;
      MS_LOAD_A(aN, cn);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        // the NEXT k-step's values: k-step u + 1 of this chunk, or k-step 0 of the next chunk (the last chunk: itself, unused)
        const float4 n0 = u < 3 ? aC[2 * (u < 3 ? u + 1 : 0)] : aN[0], n1 = u < 3 ? aC[2 * (u < 3 ? u + 1 : 0) + 1] : aN[1];
        const float vn[8] = {n0.x, n0.y, n0.z, n0.w, n1.x, n1.y, n1.z, n1.w};
        const uint32_t base = ms_lds_addr(R.lds + R.cur * M::PHASE_BYTES + (u % MS_G) * NB * 3072 + lane * 16);
        if (u % MS_G == 0) ms_kstep<NB, true>(X, B, vn, Bn, A0, A1, base);
        else ms_kstep<NB, false>(X, B, vn, Bn, A1, A0, base);
        B[0] = Bn[0]; B[1] = Bn[1]; B[2] = Bn[2];
        if (u % MS_G == MS_G - 1) {
          ms_next_phase<NB>(R, tid);
          ms_load_a<0>(A0, ms_lds_addr(R.lds + R.cur * M::PHASE_BYTES + lane * 16));
        }
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) aC[q] = aN[q];
    }
#undef MS_LOAD_A
    ms_wait_a<0>(A0);
    MS_MARK(0);
    ms_bias_act<NB>(X, vec, h, N.activation);
    MS_MARK(1);
    // ---- fc1 ----
    ms_reg_layer<NB>(Y, X, R, tid);
    MS_MARK(2);
    ms_bias_act<NB>(Y, vec + MAXH, h, N.activation);
    if (a.h1_out && n_rows && (uint32_t)r < n_rows) { // keep fc1's activations for the policy heads (row-major n x H)
      float *dst = a.h1_out + (size_t)(row0 + r) * N.H;
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          if (32 * nb + 8 * g + 4 * h < N.H) *(float4 *)(dst + 32 * nb + 8 * g + 4 * h) = make_float4(Y[nb][4 * g], Y[nb][4 * g + 1], Y[nb][4 * g + 2], Y[nb][4 * g + 3]);
    }
    // ---- value_fc2, then value_fc3 + sigmoid straight from the accumulators (network.h:14,75) ----
    MS_MARK(3);
    ms_reg_layer<NB>(X, Y, R, tid);
    MS_MARK(4);
    float part = 0.0f;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 b = *(const float4 *)(vec + 2 * MAXH + 32 * nb + 8 * g + 4 * h);
        const float4 w = *(const float4 *)(vec + 3 * MAXH + 32 * nb + 8 * g + 4 * h);
        part = fmaf(act_fn(X[nb][4 * g + 0] + b.x, N.activation), w.x, part);
        part = fmaf(act_fn(X[nb][4 * g + 1] + b.y, N.activation), w.y, part);
        part = fmaf(act_fn(X[nb][4 * g + 2] + b.z, N.activation), w.z, part);
        part = fmaf(act_fn(X[nb][4 * g + 3] + b.w, N.activation), w.w, part);
      }
    part += __shfl_xor(part, 32, 64);
    if (h == 0 && (uint32_t)r < n_rows) a.values[row0 + r] = 1.0f / (1.0f + expf(-(part + N.b3)));
    MS_MARK(5);
    MS_FLUSH();
  }
  // The ring always has one phase in flight.  It must have landed before the wave ends: LDS-DMA still outstanding at
  // s_endpgm would be written into LDS that by then belongs to another workgroup (seen as 1e-5 noise in the next launch).
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---- K3'': the main net on the fp16 matrix pipe, fp32 values carried as scaled fp16 PAIRS (round 5) ------------------------
// k_mainnet_split pays six bf16 MFMAs per fp32 multiply-add block because a bf16 part holds 8 significant bits.  An fp16 part
// holds 11, so TWO round-to-nearest parts carry an fp32 value to 2^-24: x s = h + l + e with h = fp16(x s), l = fp16(x s - h) (that
// remainder is exact in fp32) and |e| <= 2^-12 |l| <= 2^-24 |x s|.  A product is then h.h + h.l + l.h (three MFMAs of the same
// rate, each product exact in the fp32 accumulator) and drops l.l <= 2^-24 |x w| -- the error class of the triple form at half
// the matrix work.  What fp16 lacks is RANGE (5 exponent bits), so every operand is multiplied by an exact power of two first:
//   * weights: one scale per ROW (= output feature), chosen on the host so that the row's largest weight lands in [2^14, 2^15)
//     (pair_stream); the accumulators are scaled back per feature, next to the bias;
//   * activations: one scale PER BATCH ROW (= per lane: the row's values of a layer sit in that lane's registers and its partner
//     lane's), so that the row's largest value lands in [2^14, 2^15).  For fc1 / value_fc2 the row maximum is read off the
//     accumulators; for fc0 the row arrives from memory 64 columns at a time, the scale is set from the first chunk and LOWERED
//     when a later chunk holds a larger value -- the accumulators are then multiplied by the (exact) ratio, like an online softmax.
// Accumulators are scaled back (two exact power-of-two factors) in front of the bias.  A part below 2^-14 of the scaled unit is an
// fp16 SUBNORMAL, which both v_cvt_f16_f32 and the fp16 MFMA honour on gfx950 (tools/experiments/mfma_f16_denorm.hip, run on the
// chip), so a value's absolute error is at most max(2^-24 |x|, 2^-39 x the row's / the layer's largest): the product is accurate
// normwise like the fp32 multiply-add it replaces whenever a row's sum is not 2^13 times smaller than its largest term --
// oakgpu_net_load* CHECKS the weight side of that (pair_layer_ok: rows and columns) and keeps a network that fails it on the bf16 triples.
// Everything else -- orientation, the LDS-DMA ring, the software pipeline -- is k_mainnet_split's; a phase is MPair::G k-steps of
// NB x 2 KB.
#ifndef OAK_MP_G
#define OAK_MP_G 4
#endif
template <int NB> struct MPair {
  static constexpr int G = NB >= 2 ? OAK_MP_G : 2;                // k-steps per phase: even, divides 4 (a chunk of fc0) and 2 NB (a register layer)
  static_assert(G % 2 == 0 && 4 % G == 0 && (2 * NB) % G == 0, "k-steps alternate two weight buffers; chunks and layers are whole phases");
  static constexpr int PHASE_BYTES = G * NB * 2048;               // [k-step][block][h l][lane] x 16 B
  static constexpr int PT = (PHASE_BYTES + 4095) / 4096;          // 1-KB DMA pieces per wave per phase, at most
  static constexpr size_t LDS = 2 * (size_t)PHASE_BYTES + 7 * MAXH * 4; // + [b0 | b1 | b2 | w3 | 1 / the rows' scales of fc0, fc1, value_fc2]
};
template <int NB>
__device__ __forceinline__ void mp_dma_phase(const uint8_t *src, uint8_t *dst, int tid) {
  using M = MPair<NB>;
  const int wave = tid >> 6;
#pragma unroll
  for (int i = 0; i < M::PT; ++i) {
    const int piece = (i * MN_BLOCK + wave * 64) * 16; // wave-uniform; PHASE_BYTES is a multiple of 1 KB
    if (M::PHASE_BYTES % 4096 == 0 || piece < M::PHASE_BYTES)
      __builtin_amdgcn_global_load_lds((ms_glb_void *)(src + (i * MN_BLOCK + tid) * 16), (ms_lds_void *)(dst + piece), 16, 0, 0);
  }
}
#ifndef OAK_MP_EXP
#define OAK_MP_EXP 0 // experiments (WRONG results): 1 no DMA after the prologue, 2 no phase barrier, 4 no LDS reads of the weights, 8 no row loads after the first chunk
#endif
template <int NB>
__device__ __forceinline__ void mp_next_phase(MSRing &R, int tid) { // (see ms_next_phase for the written-out wait)
  using M = MPair<NB>;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (!(OAK_MP_EXP & 2)) __syncthreads();
  const int done = R.cur;
  R.cur ^= 1;
  R.phase = R.phase + 1 == R.n_phases ? 0 : R.phase + 1;
  const int nx = R.phase + 1 == R.n_phases ? 0 : R.phase + 1;
  if (!(OAK_MP_EXP & 1)) mp_dma_phase<NB>(R.stream + (size_t)nx * M::PHASE_BYTES, R.lds + done * M::PHASE_BYTES, tid);
}
template <int OFF>
__device__ __forceinline__ void mp_load_a(f16x8 (&A)[2], uint32_t addr) {
  static_assert(OFF >= 0 && OFF + 1024 < 65536, "ds_read offset field");
  if (OAK_MP_EXP & 4) { asm volatile("" : "=v"(A[0]), "=v"(A[1]) : "v"(addr)); return; }
  asm volatile("ds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%4" : "=&v"(A[0]), "=&v"(A[1]) : "v"(addr), "n"(OFF), "n"(OFF + 1024));
}
template <int YOUNGER> // wait for a pair with YOUNGER LDS reads (0 or 2) issued behind it
__device__ __forceinline__ void mp_wait_a(f16x8 (&A)[2]) {
  if (YOUNGER == 2) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(A[0]), "+v"(A[1]));
  else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(A[0]), "+v"(A[1]));
}
template <int NB, bool SAME_PHASE>
__device__ __forceinline__ void mp_kstep(f32x16 (&acc)[NB], const f16x8 (&B)[2], const float (&vn)[8], float scale_n, f16x8 (&Bn)[2], f16x8 (&Afirst)[2],
                                         f16x8 (&Anext)[2], uint32_t base) {
  f16x8 A[2][2];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    f16x8 (&W)[2] = nb == 0 ? Afirst : A[nb & 1];
#define MP_LOAD_NEXT(NBV)                                                                   \
  if (nb == NBV) {                                                                          \
    if (NBV + 1 < NB) mp_load_a<(NBV + 1) * 2048>(A[(NBV + 1) & 1], base);                   \
    else if (SAME_PHASE) mp_load_a<NB * 2048>(Anext, base);                                  \
  }
    MP_LOAD_NEXT(0) MP_LOAD_NEXT(1) MP_LOAD_NEXT(2) MP_LOAD_NEXT(3) MP_LOAD_NEXT(4) MP_LOAD_NEXT(5) MP_LOAD_NEXT(6) MP_LOAD_NEXT(7)
#undef MP_LOAD_NEXT
    if (nb + 1 < NB || SAME_PHASE) mp_wait_a<2>(W); else mp_wait_a<0>(W);
    if (OAK_MP_SPLIT == 1 && NB == 8) { if ((nb & 1) == 0) mp_split_part(vn, scale_n, Bn, nb, nb + 2); } // (two values per conversion)
    else mp_split_part(vn, scale_n, Bn, nb * 8 / NB, (nb + 1) * 8 / NB);
    acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(W[1], B[0], acc[nb], 0, 0, 0); // l . h   (small terms first)
    acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(W[0], B[1], acc[nb], 0, 0, 0); // h . l
    acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(W[0], B[0], acc[nb], 0, 0, 0); // h . h
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, OAK_MP_VALU, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}
// the accumulators back to the layer's real values (x 1 / the batch row's scale x 1 / the weight row's: two exact factors), bias, activation
template <int NB>
__device__ __forceinline__ void mp_bias_act(f32x16 (&acc)[NB], float inv_row, const float *inv_out, const float *bias, int h, int activation) {
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 b = *(const float4 *)(bias + 32 * nb + 8 * g + 4 * h);
      const float4 s = *(const float4 *)(inv_out + 32 * nb + 8 * g + 4 * h); // 1 / the scale of the weights' row = output feature
      acc[nb][4 * g + 0] = act_fn(acc[nb][4 * g + 0] * inv_row * s.x + b.x, activation);
      acc[nb][4 * g + 1] = act_fn(acc[nb][4 * g + 1] * inv_row * s.y + b.y, activation);
      acc[nb][4 * g + 2] = act_fn(acc[nb][4 * g + 2] * inv_row * s.z + b.z, activation);
      acc[nb][4 * g + 3] = act_fn(acc[nb][4 * g + 3] * inv_row * s.w + b.w, activation);
    }
}
// a layer whose input is the previous layer's activations in registers; returns the row's scale (the caller scales back)
template <int NB>
__device__ __forceinline__ float mp_reg_layer(f32x16 (&acc)[NB], const f32x16 (&in)[NB], MSRing &R, int tid) {
  using M = MPair<NB>;
  const int lane16 = (tid & 63) * 16;
  float m = MP_FLOOR;
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int q = 0; q < 16; ++q) { m = fmaxf(m, fabsf(in[nb][q])); acc[nb][q] = 0.0f; }
  m = fmaxf(m, __shfl_xor(m, 32, 64)); // the other half of the row's features
  const float scale = mp_scale_of(m);
  f16x8 B[2], Bn[2], A0[2], A1[2]; // A0 / A1: block 0's weights of the even / odd k-steps
  {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = in[0][j];
    mp_split_part(v, scale, B, 0, 8);
  }
  mp_load_a<0>(A0, ms_lds_addr(R.lds + R.cur * M::PHASE_BYTES + lane16));
#pragma unroll
  for (int t = 0; t < 2 * NB; ++t) {
    float vn[8];
    const int tn = t + 1 < 2 * NB ? t + 1 : t;
#pragma unroll
    for (int j = 0; j < 8; ++j) vn[j] = in[tn >> 1][8 * (tn & 1) + j];
    constexpr int G = M::G;
    const uint32_t base = ms_lds_addr(R.lds + R.cur * M::PHASE_BYTES + (t % G) * NB * 2048 + lane16);
    const bool last = t % G == G - 1;
    if ((t & 1) == 0) { if (!last) mp_kstep<NB, true>(acc, B, vn, scale, Bn, A0, A1, base); else mp_kstep<NB, false>(acc, B, vn, scale, Bn, A0, A1, base); }
    else { if (!last) mp_kstep<NB, true>(acc, B, vn, scale, Bn, A1, A0, base); else mp_kstep<NB, false>(acc, B, vn, scale, Bn, A1, A0, base); }
    B[0] = Bn[0]; B[1] = Bn[1];
    if (last) {
      mp_next_phase<NB>(R, tid);
      if ((t & 1) == 0) mp_load_a<0>(A1, ms_lds_addr(R.lds + R.cur * M::PHASE_BYTES + lane16));
      else mp_load_a<0>(A0, ms_lds_addr(R.lds + R.cur * M::PHASE_BYTES + lane16));
    }
  }
  mp_wait_a<0>(A0); // nothing of this layer's reads stays in flight past it (the next layer reads A0 again)
  return scale;
}

template <int NB>
__global__ __launch_bounds__(MN_BLOCK) void k_mainnet_pair(MainArgs a) {
  extern __shared__ __align__(16) uint8_t lds_b[];
  using M = MPair<NB>;
  const NetDev &N = a.net;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
  float *vec = (float *)(lds_b + 2 * M::PHASE_BYTES); // [b0 | b1 | b2 | w3 | 1 / row scales of fc0 | fc1 | value_fc2], MAXH each
  for (int i = tid; i < MAXH; i += MN_BLOCK) {
    vec[i] = i < N.H ? N.b0[i] : 0.0f; vec[MAXH + i] = i < N.H ? N.b1[i] : 0.0f;
    vec[2 * MAXH + i] = i < N.VH ? N.b2[i] : 0.0f; vec[3 * MAXH + i] = i < N.VH ? N.w3[i] : 0.0f;
    vec[4 * MAXH + i] = i < N.H ? N.wp_inv[i] : 0.0f; vec[5 * MAXH + i] = i < N.H ? N.wp_inv[N.H + i] : 0.0f;
    vec[6 * MAXH + i] = i < N.VH ? N.wp_inv[2 * N.H + i] : 0.0f;
  }
  const int T0 = N.ws_T0, K = N.emb_dim;
  MSRing R{0, 0, (T0 + 4 * NB) / M::G, (const uint8_t *)N.wp, lds_b}; // fc0: T0 k-steps, fc1 and value_fc2: 2 NB each
  mp_dma_phase<NB>(R.stream, lds_b, tid);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads(); // (also publishes vec)
  mp_dma_phase<NB>(R.stream + (size_t)M::PHASE_BYTES, lds_b + M::PHASE_BYTES, tid);

  f32x16 X[NB], Y[NB];
  const uint32_t ngroups = (a.n + 127) / 128;
  for (uint32_t grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    const uint32_t row0 = grp * 128 + wave * 32;
    const uint32_t n_rows = row0 < a.n ? min(32u, a.n - row0) : 0u;
    const uint32_t grow = n_rows ? row0 + ((uint32_t)r < n_rows ? (uint32_t)r : n_rows - 1) : a.n - 1;
    const float *arow = a.emb + (size_t)grow * K;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int q = 0; q < 16; ++q) X[nb][q] = 0.0f;
    float4 aC[8], aN[8];
#define MP_LOAD_ROW(av, c)                                                                                       \
  _Pragma("unroll") for (int u_ = 0; u_ < 4; ++u_) _Pragma("unroll") for (int qq_ = 0; qq_ < 2; ++qq_) {          \
    const int col_ = 64 * (c) + 16 * u_ + 8 * h + 4 * qq_;                                                       \
    av[2 * u_ + qq_] = *(const float4 *)(arow + (col_ < K ? col_ : 0));                                          \
  }
    // the largest magnitude of a chunk's 64 columns of this row (both half-waves)
    auto chunk_max = [](const float4 (&av)[8]) {
      float m = MP_FLOOR;
#pragma unroll
      for (int q = 0; q < 8; ++q) m = fmaxf(fmaxf(m, fmaxf(fabsf(av[q].x), fabsf(av[q].y))), fmaxf(fabsf(av[q].z), fabsf(av[q].w)));
      return fmaxf(m, __shfl_xor(m, 32, 64));
    };
    const int nch = T0 / 4;
    MS_T0();
    MP_LOAD_ROW(aC, 0);
    float scale = mp_scale_of(chunk_max(aC)); // the row's scale so far
    f16x8 B[2], Bn[2], A0[2], A1[2];
    {
      const float v[8] = {aC[0].x, aC[0].y, aC[0].z, aC[0].w, aC[1].x, aC[1].y, aC[1].z, aC[1].w};
      mp_split_part(v, scale, B, 0, 8);
    }
    mp_load_a<0>(A0, ms_lds_addr(R.lds + R.cur * M::PHASE_BYTES + lane * 16));
#pragma unroll 1
    for (int c = 0; c < nch; ++c) {
      const int cn = c + 1 < nch ? c + 1 : c;
      if (!(OAK_MP_EXP & 8)) { MP_LOAD_ROW(aN, cn); }
      float scale_n = scale;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float4 n0 = u < 3 ? aC[2 * (u < 3 ? u + 1 : 0)] : aN[0], n1 = u < 3 ? aC[2 * (u < 3 ? u + 1 : 0) + 1] : aN[1];
        const float vn[8] = {n0.x, n0.y, n0.z, n0.w, n1.x, n1.y, n1.z, n1.w};
        if (u == 3) scale_n = fminf(scale, mp_scale_of(chunk_max(aN))); // the next chunk may need a smaller scale: its first k-step is split with it
        const int ph = u % M::G; // (a chunk is one or two whole phases)
        const uint32_t base = ms_lds_addr(R.lds + R.cur * M::PHASE_BYTES + ph * NB * 2048 + lane * 16);
        const bool last = ph == M::G - 1;
        const float sn = u == 3 ? scale_n : scale;
        if ((u & 1) == 0) { if (!last) mp_kstep<NB, true>(X, B, vn, sn, Bn, A0, A1, base); else mp_kstep<NB, false>(X, B, vn, sn, Bn, A0, A1, base); }
        else { if (!last) mp_kstep<NB, true>(X, B, vn, sn, Bn, A1, A0, base); else mp_kstep<NB, false>(X, B, vn, sn, Bn, A1, A0, base); }
        B[0] = Bn[0]; B[1] = Bn[1];
        if (last) {
          mp_next_phase<NB>(R, tid);
          if ((u & 1) == 0) mp_load_a<0>(A1, ms_lds_addr(R.lds + R.cur * M::PHASE_BYTES + lane * 16));
          else mp_load_a<0>(A0, ms_lds_addr(R.lds + R.cur * M::PHASE_BYTES + lane * 16));
        }
      }
      if (scale_n != scale) { // (per lane, rare: a later chunk raised the row's maximum) the sums so far move to the new scale
        const float f = mp_ratio(scale_n, scale);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
          for (int q = 0; q < 16; ++q) X[nb][q] *= f;
        scale = scale_n;
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) aC[q] = aN[q];
    }
#undef MP_LOAD_ROW
    mp_wait_a<0>(A0); // (four k-steps per chunk: the buffer in flight behind the last one is A0)
    MS_MARK(0);
    mp_bias_act<NB>(X, mp_inverse(scale), vec + 4 * MAXH, vec, h, N.activation);
    MS_MARK(1);
    // ---- fc1 ----
    const float s1 = mp_reg_layer<NB>(Y, X, R, tid);
    MS_MARK(2);
    mp_bias_act<NB>(Y, mp_inverse(s1), vec + 5 * MAXH, vec + MAXH, h, N.activation);
    if (a.h1_out && n_rows && (uint32_t)r < n_rows) { // keep fc1's activations for the policy heads (row-major n x H)
      float *dst = a.h1_out + (size_t)(row0 + r) * N.H;
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          if (32 * nb + 8 * g + 4 * h < N.H) *(float4 *)(dst + 32 * nb + 8 * g + 4 * h) = make_float4(Y[nb][4 * g], Y[nb][4 * g + 1], Y[nb][4 * g + 2], Y[nb][4 * g + 3]);
    }
    // ---- value_fc2, then value_fc3 + sigmoid straight from the accumulators (network.h:14,75) ----
    MS_MARK(3);
    const float s2 = mp_reg_layer<NB>(X, Y, R, tid);
    MS_MARK(4);
    const float i2 = mp_inverse(s2);
    float part = 0.0f;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 b = *(const float4 *)(vec + 2 * MAXH + 32 * nb + 8 * g + 4 * h);
        const float4 w = *(const float4 *)(vec + 3 * MAXH + 32 * nb + 8 * g + 4 * h);
        const float4 s = *(const float4 *)(vec + 6 * MAXH + 32 * nb + 8 * g + 4 * h);
        part = fmaf(act_fn(X[nb][4 * g + 0] * i2 * s.x + b.x, N.activation), w.x, part);
        part = fmaf(act_fn(X[nb][4 * g + 1] * i2 * s.y + b.y, N.activation), w.y, part);
        part = fmaf(act_fn(X[nb][4 * g + 2] * i2 * s.z + b.z, N.activation), w.z, part);
        part = fmaf(act_fn(X[nb][4 * g + 3] * i2 * s.w + b.w, N.activation), w.w, part);
      }
    part += __shfl_xor(part, 32, 64);
    if (h == 0 && (uint32_t)r < n_rows) a.values[row0 + r] = 1.0f / (1.0f + expf(-(part + N.b3)));
    MS_MARK(5);
    MS_FLUSH();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // (the ring's last phase must have landed before the wave ends: k_mainnet_split)
}

// ---- policy heads: value_policy_inference's logits (network.h:102-123, main-net.h:67-107) ----------
// Encode::Battle::Policy::get_index (encode/battle/policy.h:29-58) on the raw battle bytes
__device__ __forceinline__ uint32_t policy_index(const uint8_t *side, uint32_t choice) {
  const uint32_t kind = choice & 3, data = choice >> 2;
  if (kind == 1) {
    if (data == 0) return 0; // Struggle / forced continue: only ever a sole option (policy.h:11-19)
    const uint32_t sid = side[176] - 1u;
    const uint32_t mid = side[24 * sid + 10 + 2 * (data - 1)]; // side.stored().moves[data - 1].id
    return mid == 0 ? 0 : mid - 1;
  }
  if (kind == 2) {
    const uint32_t pid = side[176 + data - 1];
    return 164 + side[24 * (pid - 1) + 21] - 1u;
  }
  return 0;
}

struct PolicyArgs {
  NetDev net;
  const float *h1;  // n x H
  const uint8_t *battles;
  const uint8_t *choices[2]; // n x 9 each
  const uint8_t *counts[2];  // n each
  float *logits[2];          // n x 9 each
  uint32_t n;
};

// ---- policy heads, row form (round 3).  Rounds 1-2's k_policy staged a 64-leaf tile and both weight matrices through LDS
// behind workgroup barriers and finished with one THREAD per (leaf, choice) walking a 256-byte weight row of its own through
// L1 (64 lanes = 64 cache lines per load): 300-450 us per 65,536 leaves for 10 % of the main net's arithmetic.  Here a wave
// owns 32 leaves, like the main net's kernels:
//   fc2 (H -> PH): P^T = Wa . H1^T with the WEIGHTS as the A operand -- as bf16 triples on the bf16 pipe since round 4 (TRIPLE:
//   the branch at the top of the tile loop), or on fp32 MFMA when the net runs in fp32 mode (fragments in policy_frag_order, four
//   k-steps per float4, straight from L2) and the leaf's fc1 row as the B operand (lane (b, hh) reads 16 bytes of its row per
//   four k-steps) -- the result has the policy-hidden features in the accumulator registers and the leaf on the lane;
//   fc3 (PH -> 315, <= 9 legal rows per side, policy.h:29-58): lane (b, hh) holds half of leaf b's features in registers and
//   dots them with the weight row of each legal choice -- rows in LDS (315 x (PH + 4) floats, staged once per head and
//   workgroup; the pad spreads the row starts over the banks) when they fit, else from global memory -- then adds its
//   partner's half (one cross-lane add).
template <int PB> struct PolicyRows {
  static constexpr int ROWS = 315;
  static constexpr int STRIDE = PB * 32 + 4;                       // floats per fc3 row in LDS
  static constexpr bool WB_LDS = (size_t)ROWS * STRIDE * 4 <= 128 * 1024;
  static constexpr int IMG_WORDS = ROWS * STRIDE + 320;             // the rows + fc3's biases (315, padded)
  static constexpr size_t LDS = WB_LDS ? (size_t)IMG_WORDS * 4 : 16;
  // One workgroup per CU (the rows fill its LDS).  With <= 64 policy-hidden features the kernel needs < 256 registers, so the
  // workgroup has EIGHT waves (two per SIMD) instead of four: the kernel waits -- for the weight fragments from L2, the leaf's
  // fc1 row, the 27-field gather behind the choices -- more than it computes (round 3: SQ_WAIT_ANY 63 % of its wave cycles).
  static constexpr int WAVES = PB <= 2 ? 8 : 4, BLOCK = 64 * WAVES;
};
template <int PB, bool TRIPLE>
__global__ __launch_bounds__(PolicyRows<PB>::BLOCK) void k_policy_rows(PolicyArgs a) {
  extern __shared__ __align__(16) float lds_f[];
  using PR = PolicyRows<PB>;
  const NetDev &N = a.net;
  const int H = N.H, PH = N.PH;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, hh = lane >> 5;
  const uint32_t ntiles = (a.n + 31) / 32;
  const int nu = H / 8; // float4 groups of four k-steps: k = 8 u + 4 hh + e
  for (int head = 0; head < 2; ++head) {
    const float4 *Wf = (const float4 *)(head ? N.q2a_f : N.q1a_f);
    const float *ba = head ? N.q2a_b : N.q1a_b;
    const float *Wb = head ? N.q2b : N.q1b, *bb = head ? N.q2b_b : N.q1b_b;
    if (PR::WB_LDS) {
      // the head's rows + biases: one prebuilt image, by LDS-DMA (round 4; as a `lds[..] = global[..]` loop it was ten dependent L2
      // round trips per head in front of every tile)
      __syncthreads(); // (the previous head's rows are no longer read)
      stage_image_dma<PR::BLOCK>(lds_f, head ? N.q2b_img : N.q1b_img, PR::IMG_WORDS);
      stage_image_wait();
    }
    for (uint32_t wt = blockIdx.x * PR::WAVES + wave; wt < ntiles; wt += gridDim.x * PR::WAVES) {
      const uint32_t row0 = wt * 32, n_rows = min(32u, a.n - row0);
      const uint32_t leaf = row0 + ((uint32_t)r < n_rows ? (uint32_t)r : n_rows - 1); // rows past the batch repeat the last one, dropped below
      const float *hrow = a.h1 + (size_t)leaf * H + 4 * hh;
      f32x16 acc[PB];
#pragma unroll
      for (int b = 0; b < PB; ++b)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[b][q] = 0.0f;
      if constexpr (TRIPLE) {
        // fc2 as bf16 triples (round 4): the fp32 MFMAs of the other branch hold the SIMD's vector issue for 64 cycles each
        // (profiles/r04_mfma_overlap.json) -- 256 of them per head and tile.  Here: k-step T = columns 16 T + 8 hh .. + 7 of the
        // leaf's fc1 row, split by truncation (e_split), against the weights' prebuilt triples (policy_triple_order, from L2),
        // the six largest partial products per block as in the main net; the next k-step's row piece and weights are asked for
        // in front of this one's MFMAs.
        const uint8_t *Wt = (const uint8_t *)(head ? N.q2a_t : N.q1a_t) + lane * 16;
        const float *hr = a.h1 + (size_t)leaf * H + 8 * hh;
        const int nT = H / 16;
        float4 x0 = *(const float4 *)hr, x1 = *(const float4 *)(hr + 4);
        bf16x8 Wc[PB][3], Wn[PB][3];
#pragma unroll
        for (int b = 0; b < PB; ++b)
#pragma unroll
          for (int q = 0; q < 3; ++q) Wc[b][q] = *(const bf16x8 *)(Wt + ((size_t)b * 3 + q) * 1024);
#pragma unroll 2
        for (int T = 0; T < nT; ++T) {
          const int Tn = T + 1 < nT ? T + 1 : T;
          const float4 n0 = *(const float4 *)(hr + 16 * Tn), n1 = *(const float4 *)(hr + 16 * Tn + 4);
#pragma unroll
          for (int b = 0; b < PB; ++b)
#pragma unroll
            for (int q = 0; q < 3; ++q) Wn[b][q] = *(const bf16x8 *)(Wt + (((size_t)Tn * PB + b) * 3 + q) * 1024);
          const float v[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
          bf16x8 X[3];
          e_split(v, X);
#pragma unroll
          for (int b = 0; b < PB; ++b) {
            acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Wc[b][2], X[0], acc[b], 0, 0, 0); // l . h   (small terms first)
            acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Wc[b][0], X[2], acc[b], 0, 0, 0); // h . l
            acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Wc[b][1], X[1], acc[b], 0, 0, 0); // m . m
            acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Wc[b][1], X[0], acc[b], 0, 0, 0); // m . h
            acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Wc[b][0], X[1], acc[b], 0, 0, 0); // h . m
            acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Wc[b][0], X[0], acc[b], 0, 0, 0); // h . h
          }
          x0 = n0; x1 = n1;
#pragma unroll
          for (int b = 0; b < PB; ++b)
#pragma unroll
            for (int q = 0; q < 3; ++q) Wc[b][q] = Wn[b][q];
        }
      } else {
      float4 xc = *(const float4 *)hrow, xn;
      float4 wc[PB], wn[PB];
#pragma unroll
      for (int b = 0; b < PB; ++b) wc[b] = Wf[(size_t)b * 64 + lane];
#pragma unroll 2
      for (int u = 0; u < nu; ++u) {
        const int un = u + 1 < nu ? u + 1 : u;
        xn = *(const float4 *)(hrow + 8 * un);
#pragma unroll
        for (int b = 0; b < PB; ++b) wn[b] = Wf[((size_t)un * PB + b) * 64 + lane];
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int b = 0; b < PB; ++b) acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(f4_pick(wc[b], e), f4_pick(xc, e), acc[b], 0, 0, 0);
        xc = xn;
#pragma unroll
        for (int b = 0; b < PB; ++b) wc[b] = wn[b];
      }
      }
      // bias + activation: register q of block b = feature 32 b + (q & 3) + 8 (q >> 2) + 4 hh of leaf r
#pragma unroll
      for (int b = 0; b < PB; ++b)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 bv = *(const float4 *)(ba + 32 * b + 8 * g + 4 * hh);
          acc[b][4 * g + 0] = act_fn(acc[b][4 * g + 0] + bv.x, N.activation);
          acc[b][4 * g + 1] = act_fn(acc[b][4 * g + 1] + bv.y, N.activation);
          acc[b][4 * g + 2] = act_fn(acc[b][4 * g + 2] + bv.z, N.activation);
          acc[b][4 * g + 3] = act_fn(acc[b][4 * g + 3] + bv.w, N.activation);
        }
      // the legal rows of fc3.  Policy::get_index (policy.h:29-58) of all nine choice slots from ONE round trip: the side's six
      // stored Pokemon (their move ids and species), its order bytes and the nine choice bytes are asked for together and the
      // right fields selected in registers -- per choice it is three DEPENDENT byte loads (choice -> order -> species / move),
      // 27 global-memory latencies in a row per tile and head with one wave per SIMD to hide them (144 us of the call).
      const uint32_t cnt = a.counts[head][leaf];
      const uint32_t *sb = (const uint32_t *)(a.battles + (size_t)leaf * 384 + head * 184);
      uint2 pw2[18];
#pragma unroll
      for (int u = 0; u < 18; ++u) pw2[u] = ((const uint2 *)sb)[u]; // 6 x 24 B of stored Pokemon (8-byte aligned: 384 and 184 are)
      const uint2 ow = *(const uint2 *)(sb + 44);                   // order[6] at bytes 176..181
      uint32_t cb[OAKGPU_MAX_CHOICES];
#pragma unroll
      for (int j = 0; j < OAKGPU_MAX_CHOICES; ++j) cb[j] = a.choices[head][(size_t)leaf * OAKGPU_MAX_CHOICES + j];
      const uint32_t pwd[36] = {pw2[0].x, pw2[0].y, pw2[1].x, pw2[1].y, pw2[2].x, pw2[2].y, pw2[3].x, pw2[3].y, pw2[4].x, pw2[4].y, pw2[5].x, pw2[5].y,
                                pw2[6].x, pw2[6].y, pw2[7].x, pw2[7].y, pw2[8].x, pw2[8].y, pw2[9].x, pw2[9].y, pw2[10].x, pw2[10].y, pw2[11].x, pw2[11].y,
                                pw2[12].x, pw2[12].y, pw2[13].x, pw2[13].y, pw2[14].x, pw2[14].y, pw2[15].x, pw2[15].y, pw2[16].x, pw2[16].y, pw2[17].x, pw2[17].y};
      // per stored Pokemon k: its four move ids (bytes 10, 12, 14, 16) packed in one word, and its species (byte 21)
      uint32_t mv4[6], spc[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        const uint32_t d2 = pwd[6 * k + 2], d3 = pwd[6 * k + 3], d4 = pwd[6 * k + 4], d5 = pwd[6 * k + 5];
        mv4[k] = ((d2 >> 16) & 0xFF) | ((d3 & 0xFF) << 8) | (((d3 >> 16) & 0xFF) << 16) | ((d4 & 0xFF) << 24);
        spc[k] = (d5 >> 8) & 0xFF;
      }
      const uint32_t ord[6] = {ow.x & 0xFF, (ow.x >> 8) & 0xFF, (ow.x >> 16) & 0xFF, ow.x >> 24, ow.y & 0xFF, (ow.y >> 8) & 0xFF};
      auto by_id = [&](const uint32_t (&v)[6], uint32_t id) { // v[id - 1], id in 1..6 (0 for anything else)
        uint32_t x = 0;
#pragma unroll
        for (int k = 0; k < 6; ++k) x = id == (uint32_t)(k + 1) ? v[k] : x;
        return x;
      };
      const uint32_t stored_moves = by_id(mv4, ord[0]);
#pragma unroll 1
      for (uint32_t j = 0; j < OAKGPU_MAX_CHOICES; ++j) {
        const bool live = j < cnt;
        uint32_t c = cb[0];
#pragma unroll
        for (int q = 1; q < OAKGPU_MAX_CHOICES; ++q) c = j == (uint32_t)q ? cb[q] : c;
        const uint32_t kind = c & 3, data = c >> 2;
        uint32_t idx = 0;
        if (kind == 1 && data >= 1 && data <= 4) { const uint32_t mid = (stored_moves >> (8 * (data - 1))) & 0xFF; idx = mid == 0 ? 0 : mid - 1; }
        if (kind == 2 && data >= 1 && data <= 6) {
          uint32_t pid = ord[0];
#pragma unroll
          for (int q = 1; q < 6; ++q) pid = data == (uint32_t)(q + 1) ? ord[q] : pid;
          idx = 164 + by_id(spc, pid) - 1u;
        }
        idx = live && idx < (uint32_t)PR::ROWS ? idx : 0u;
        const float *w = (PR::WB_LDS ? lds_f + idx * PR::STRIDE : Wb + (size_t)idx * PH) + 4 * hh;
        float part = 0.0f;
#pragma unroll
        for (int b = 0; b < PB; ++b)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const float4 wv = *(const float4 *)(w + 32 * b + 8 * g);
            part = fmaf(wv.x, acc[b][4 * g + 0], part);
            part = fmaf(wv.y, acc[b][4 * g + 1], part);
            part = fmaf(wv.z, acc[b][4 * g + 2], part);
            part = fmaf(wv.w, acc[b][4 * g + 3], part);
          }
        part += __shfl_xor(part, 32, 64);
        const float bias = PR::WB_LDS ? lds_f[PR::ROWS * PR::STRIDE + idx] : bb[idx];
        if (hh == 0 && (uint32_t)r < n_rows) a.logits[head][(size_t)leaf * OAKGPU_MAX_CHOICES + j] = live ? part + bias : 0.0f;
      }
    }
  }
}

} // namespace oak

// =================================== C ABI =====================================================
struct oakgpu_net {
  oak::NetDev dev;
  std::vector<void *> allocs;
  int in_dim, hidden, value_hidden, policy_hidden; // unpadded, as in the file
  int device;        // the device the weights live on
  int main_mode;     // which kernel runs the main net: 0 = k_mainnet_wave (fp32 MFMA), 1 = k_mainnet_split (bf16 triples), 2 = k_mainnet_pair (fp16 pairs)
  bool pair_safe;    // every main-net layer passes pair_layer_ok (rows keep fp32 accuracy as scaled fp16 pairs, no dwarfed column): k_mainnet_pair may run
  bool split_safe;   // no main-net weight above 2^20 in magnitude: what a flushed low bf16 part loses cannot be amplified back (else fp32 MFMA only)
  bool embed_safe;   // ... and none in the embedding nets' second layers either: the embedding passes' triples are safe (else k_embed_lds: fp32 MFMA)
};

namespace {
struct HostAffine { uint32_t in, out; std::vector<float> b, w; };

bool read_affine(const uint8_t *&p, const uint8_t *end, HostAffine &a) {
  if (end - p < 8) return false;
  memcpy(&a.in, p, 4);
  memcpy(&a.out, p + 4, 4);
  p += 8;
  if (a.in == 0 || a.out == 0 || a.in > (1u << 16) || a.out > (1u << 16)) return false;
  const size_t nb = (size_t)a.out * 4, nw = (size_t)a.out * a.in * 4;
  if ((size_t)(end - p) < nb + nw) return false;
  a.b.resize(a.out);
  a.w.resize((size_t)a.out * a.in);
  memcpy(a.b.data(), p, nb);
  p += nb;
  memcpy(a.w.data(), p, nw);
  p += nw;
  return true;
}

int upload(oakgpu_net *net, const std::vector<float> &h, const float **out) {
  void *d = nullptr;
  hipError_t e = hipMalloc(&d, h.size() * 4);
  if (e != hipSuccess) return oakgpu_fail_hip((int)e, "hipMalloc(net)");
  net->allocs.push_back(d);
  e = hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  if (e != hipSuccess) return oakgpu_fail_hip((int)e, "hipMemcpy(net)");
  *out = (const float *)d;
  return 0;
}

std::vector<float> transpose(const HostAffine &a) { // W[out][in] -> Wt[in][out]
  std::vector<float> t((size_t)a.in * a.out);
  for (uint32_t o = 0; o < a.out; ++o)
    for (uint32_t i = 0; i < a.in; ++i) t[(size_t)i * a.out + o] = a.w[(size_t)o * a.in + i];
  return t;
}
// pad rows (out) to a multiple of 32 and columns (in) to `in_pad` with zeros
std::vector<float> pad_rows(const HostAffine &a, uint32_t out_pad, uint32_t in_pad) {
  std::vector<float> t((size_t)out_pad * in_pad, 0.0f);
  for (uint32_t o = 0; o < a.out; ++o)
    for (uint32_t i = 0; i < a.in; ++i) t[(size_t)o * in_pad + i] = a.w[(size_t)o * a.in + i];
  return t;
}
// k_mainnet_wave's MFMA-fragment order of a weight matrix: the 2 float4 x NB n-blocks of one SUB-chunk (8 k-steps) are contiguous:
// float4 (((c*4 + u)*NB + nb)*2 + qq)*64 + lane -> W[nb*32 + r][c*64 + h*32 + 8u + 4qq .. +3], so the 16 loads of a sub-chunk
// are one scalar base + the lane's offset + small constants (no per-block vector address arithmetic).  NB is rounded up to
// the kernel's template width (1, 2, 4, 8) with all-zero blocks, so that every block index is a compile-time constant.
std::vector<float> frag_order_wave(const HostAffine &a, uint32_t out_pad) {
  const uint32_t nch = (a.in + 63) / 64, nb_real = out_pad / 32;
  const uint32_t NB = nb_real > 4 ? 8 : nb_real > 2 ? 4 : nb_real; // the kernel's block count (wave_layer_nb): absent blocks are zeros
  std::vector<float> f((size_t)nch * NB * 8 * 64 * 4, 0.0f);
  for (uint32_t c = 0; c < nch; ++c)
    for (uint32_t u = 0; u < 4; ++u)
      for (uint32_t nb = 0; nb < NB; ++nb)
        for (uint32_t qq = 0; qq < 2; ++qq)
          for (uint32_t lane = 0; lane < 64; ++lane)
            for (uint32_t e = 0; e < 4; ++e) {
              const uint32_t row = nb * 32 + (lane & 31), col = c * 64 + (lane >> 5) * 32 + 8 * u + 4 * qq + e;
              if (row < a.out && col < a.in) f[(((((size_t)c * 4 + u) * NB + nb) * 2 + qq) * 64 + lane) * 4 + e] = a.w[(size_t)row * a.in + col];
            }
  return f;
}

// k_mainnet_split's weight stream: fc0, fc1, value_fc2 back to back, every weight as the bf16 triple (h, m, l), in the order
// the kernel's LDS ring is read: byte (((t * NB + nb) * 3 + part) * 64 + lane) * 16 + 2 j = part of W[32 nb + (lane & 31)][k],
// k = 16 t + 8 h + j for fc0 (h = lane >> 5) and 32 (t >> 1) + (reg & 3) + 8 (reg >> 2) + 4 h, reg = 8 (t & 1) + j, for the two
// layers whose input is the previous layer's accumulators.  Absent rows / columns are zeros.
uint16_t f32_to_bf16(float f) { // round to nearest even (weights are finite; a NaN stays a NaN)
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (uint16_t)((u >> 16) | 0x40);
  return (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}
float bf16_to_f32(uint16_t b) { const uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); return f; }
std::vector<uint16_t> split_stream(const HostAffine &fc0, const HostAffine &fc1, const HostAffine &v2, uint32_t NB, uint32_t T0) {
  const uint32_t steps = T0 + 4 * NB;
  std::vector<uint16_t> w((size_t)steps * NB * 3 * 64 * 8, 0);
  auto put = [&](uint32_t t_flat, const HostAffine &a, uint32_t nb, uint32_t lane, uint32_t j, uint32_t k) {
    const uint32_t n = 32 * nb + (lane & 31);
    if (n >= a.out || k >= a.in) return;
    const float x = a.w[(size_t)n * a.in + k];
    const uint16_t hh = f32_to_bf16(x);
    const float r1 = x - bf16_to_f32(hh);
    const uint16_t mm = f32_to_bf16(r1);
    const uint16_t ll = f32_to_bf16(r1 - bf16_to_f32(mm));
    const size_t base = ((size_t)t_flat * NB + nb) * 3;
    w[((base + 0) * 64 + lane) * 8 + j] = hh;
    w[((base + 1) * 64 + lane) * 8 + j] = mm;
    w[((base + 2) * 64 + lane) * 8 + j] = ll;
  };
  for (uint32_t nb = 0; nb < NB; ++nb)
    for (uint32_t lane = 0; lane < 64; ++lane)
      for (uint32_t j = 0; j < 8; ++j) {
        const uint32_t hb = lane >> 5;
        for (uint32_t t = 0; t < T0; ++t) put(t, fc0, nb, lane, j, 16 * t + 8 * hb + j);
        for (uint32_t t = 0; t < 2 * NB; ++t) {
          const uint32_t reg = 8 * (t & 1) + j, k = 32 * (t >> 1) + (reg & 3) + 8 * (reg >> 2) + 4 * hb;
          put(T0 + t, fc1, nb, lane, j, k);
          put(T0 + 2 * NB + t, v2, nb, lane, j, k);
        }
      }
  return w;
}

// k_mainnet_pair's weight stream: the same order with TWO parts per weight, (h, l) = the fp16 pair of w x scale, scale = the power
// of two that takes the largest |w| of the weight's ROW (= output feature) into [2^14, 2^15) (1 for an all-zero row): byte
// (((t * NB + nb) * 2 + part) * 64 + lane) * 16 + 2 j.  inv = 1 / scale of every row: [fc0: Hp | fc1: Hp | value_fc2: VHp] (padded rows: 0).
uint16_t f16_bits(_Float16 v) { uint16_t b; memcpy(&b, &v, 2); return b; }
float pair_scale_of(float m) { // the power of two that takes m into [2^14, 2^15); 1 for m = 0
  if (!(m > 0.0f) || !std::isfinite(m)) return 1.0f;
  int e;
  std::frexp(m, &e); // m = f x 2^e, f in [0.5, 1): m in [2^(e-1), 2^e)
  e = 15 - e;        // (kept inside 2^+-100: the scale, its inverse and their products with a batch row's scale stay normal fp32 numbers;
  return std::ldexp(1.0f, e > 100 ? 100 : e < -100 ? -100 : e); // a row beyond that fails pair_layer_ok or saturates like fp32 does)
}
float row_pair_scale(const HostAffine &a, uint32_t n) {
  float m = 0.0f;
  for (uint32_t k = 0; k < a.in; ++k) m = std::fmax(m, std::fabs(a.w[(size_t)n * a.in + k]));
  return pair_scale_of(m);
}
std::vector<uint16_t> pair_stream(const HostAffine &fc0, const HostAffine &fc1, const HostAffine &v2, uint32_t NB, uint32_t T0, uint32_t Hp, uint32_t VHp,
                                  std::vector<float> &inv) {
  const uint32_t steps = T0 + 4 * NB;
  std::vector<uint16_t> w((size_t)steps * NB * 2 * 64 * 8, 0);
  inv.assign((size_t)2 * Hp + VHp, 0.0f);
  std::vector<float> sc0(fc0.out), sc1(fc1.out), sc2(v2.out);
  for (uint32_t n = 0; n < fc0.out; ++n) { sc0[n] = row_pair_scale(fc0, n); inv[n] = 1.0f / sc0[n]; }
  for (uint32_t n = 0; n < fc1.out; ++n) { sc1[n] = row_pair_scale(fc1, n); inv[Hp + n] = 1.0f / sc1[n]; }
  for (uint32_t n = 0; n < v2.out; ++n) { sc2[n] = row_pair_scale(v2, n); inv[2 * (size_t)Hp + n] = 1.0f / sc2[n]; }
  auto put = [&](uint32_t t_flat, const HostAffine &a, const std::vector<float> &sc, uint32_t nb, uint32_t lane, uint32_t j, uint32_t k) {
    const uint32_t n = 32 * nb + (lane & 31);
    if (n >= a.out || k >= a.in) return;
    const float x = a.w[(size_t)n * a.in + k] * sc[n];
    const _Float16 hi = (_Float16)x;
    const _Float16 lo = (_Float16)(x - (float)hi);
    const size_t base = ((size_t)t_flat * NB + nb) * 2;
    w[((base + 0) * 64 + lane) * 8 + j] = f16_bits(hi);
    w[((base + 1) * 64 + lane) * 8 + j] = f16_bits(lo);
  };
  for (uint32_t nb = 0; nb < NB; ++nb)
    for (uint32_t lane = 0; lane < 64; ++lane)
      for (uint32_t j = 0; j < 8; ++j) {
        const uint32_t hb = lane >> 5;
        for (uint32_t t = 0; t < T0; ++t) put(t, fc0, sc0, nb, lane, j, 16 * t + 8 * hb + j);
        for (uint32_t t = 0; t < 2 * NB; ++t) {
          const uint32_t reg = 8 * (t & 1) + j, k = 32 * (t >> 1) + (reg & 3) + 8 * (reg >> 2) + 4 * hb;
          put(T0 + t, fc1, sc1, nb, lane, j, k);
          put(T0 + 2 * NB + t, v2, sc2, nb, lane, j, k);
        }
      }
  return w;
}
// May this layer run as pairs?  Two checks on its weights:
//   rows    -- every row (= output feature) carries its own scale, so a row of small weights next to rows of large ones loses nothing.
//              Within a row the pair of w x scale misses w by at most 2^-24 |w| while its low part is a normal fp16 number and by up to
//              2^-25 / scale (= 2^-39 x the row's largest weight) when it is a subnormal; the row passes when the sum of what its pairs
//              miss stays within 2^-23 of the sum of its magnitudes -- the normwise error of ONE fp32 rounding per weight;
//   columns -- a column whose largest weight is more than 2^16 below the layer's largest is carried with fewer bits, which is
//              harmless while its input is no larger than the others' and wrong when the network compensates small weights with
//              large inputs (an embedding net scaled up by 2^60 in front of fc0 columns scaled down by 2^60 is the same function in
//              fp32).  The loader cannot see the inputs, so every non-zero column must reach 2^-18 of the layer's largest weight: a column
//              that small matters only through inputs 2^18 times the others', and is then still carried to 2^-21 (the bound a
//              nearly dead input unit of a trained network passes, a rescaled one does not).
// With both, a layer's outputs are accurate to ~2^-23 of (the batch row's largest input) x (the weight row's magnitudes) -- the fp32
// multiply-add's own normwise error -- for any input whose values of consequence lie within 2^15 of the row's largest.  A network that
// fails stays on the bf16 triples, which have fp32's exponent range.
bool pair_layer_ok(const HostAffine &a) {
  float layer_max = 0.0f;
  std::vector<float> col_max(a.in, 0.0f);
  for (uint32_t n = 0; n < a.out; ++n) {
    const float scale = row_pair_scale(a, n);
    double miss = 0.0, mag = 0.0;
    for (uint32_t k = 0; k < a.in; ++k) {
      const float wv = a.w[(size_t)n * a.in + k], x = wv * scale;
      const _Float16 hi = (_Float16)x;
      const _Float16 lo = (_Float16)(x - (float)hi);
      miss += std::fabs((double)x - ((double)(float)hi + (double)(float)lo));
      mag += std::fabs((double)x);
      col_max[k] = std::fmax(col_max[k], std::fabs(wv));
      layer_max = std::fmax(layer_max, std::fabs(wv));
    }
    if (miss > mag * 0x1p-23 || !std::isfinite(mag)) return false;
  }
  for (uint32_t k = 0; k < a.in; ++k)
    if (col_max[k] != 0.0f && col_max[k] < layer_max * 0x1p-18f) return false;
  return true;
}

// k_policy_rows' A operand of a policy head's fc2 (H -> PH): float4 ((u * PB + b) * 64 + lane) = W[32 b + (lane & 31)][8 u + 4 (lane >> 5) .. + 3]
// (four k-steps per load; the leaf's fc1 row supplies the same columns as the B operand).  Absent rows / columns are zeros.
// ... and as bf16 triples, k_policy_rows<PB, true>'s A operand: 16-bit word ((((T * PB + b) * 3 + part) * 64 + lane) * 8 + j) = part of
// W[32 b + (lane & 31)][16 T + 8 (lane >> 5) + j]; absent rows / columns are zeros.  Returned as floats (the upload's unit).
std::vector<float> policy_triple_order(const HostAffine &a, uint32_t H, uint32_t PB);
std::vector<float> policy_frag_order(const HostAffine &a, uint32_t H, uint32_t PB) {
  const uint32_t nu = H / 8;
  std::vector<float> f((size_t)nu * PB * 64 * 4, 0.0f);
  for (uint32_t u = 0; u < nu; ++u)
    for (uint32_t b = 0; b < PB; ++b)
      for (uint32_t lane = 0; lane < 64; ++lane)
        for (uint32_t e = 0; e < 4; ++e) {
          const uint32_t row = 32 * b + (lane & 31), col = 8 * u + 4 * (lane >> 5) + e;
          if (row < a.out && col < a.in) f[(((size_t)u * PB + b) * 64 + lane) * 4 + e] = a.w[(size_t)row * a.in + col];
        }
  return f;
}

// k_embed_arows: W0^T padded to 128 floats per row, + an all-zero last row (the move rows are read from this copy)
// + rows AR_COMBINED + i (i = move id - 1, 0..164): the active move row 45 + i PLUS the stored Pokemon's move row 229 + 5 + i --
// the two halves of a move slot hold the same move unless Transform / Mimic replaced the active one, so the pass loads one row
std::vector<float> arows_rows(const HostAffine &a) { // a.w is [out = hidden][in]; row r of W0^T = column r of W
  std::vector<float> d((size_t)(oak::AR_COMBINED + 165) * 128, 0.0f);
  for (uint32_t r = 0; r < a.in; ++r)
    for (uint32_t c = 0; c < a.out && c < 128; ++c) d[(size_t)r * 128 + c] = a.w[(size_t)c * a.in + r];
  for (uint32_t i = 0; i < 165; ++i)
    for (uint32_t c = 0; c < 128; ++c) d[(size_t)(oak::AR_COMBINED + i) * 128 + c] = d[(size_t)(45 + i) * 128 + c] + d[(size_t)(229 + 5 + i) * 128 + c];
  return d;
}
// ... the dense part of W0 as bf16 triples, the A operand of dense_layer_bf16: 16-bit word ((((T * 4 + blk) * 3 + part) * 64 + lane) * 4 + j)
// = part of W0[channel 4 (lane & 31) + blk][row of dense feature d = 8 T + 4 (lane >> 5) + j]  (d = 0: the bias; d >= F: zero)
template <class RowOf>
std::vector<float> dense_triple_frag(const HostAffine &a, uint32_t KT, uint32_t F, RowOf row_of) {
  std::vector<uint16_t> w((size_t)KT * 4 * 3 * 64 * 4, 0);
  for (uint32_t T = 0; T < KT; ++T)
    for (uint32_t blk = 0; blk < 4; ++blk)
      for (uint32_t lane = 0; lane < 64; ++lane)
        for (uint32_t j = 0; j < 4; ++j) {
          const uint32_t d = 8 * T + 4 * (lane >> 5) + j, c = 4 * (lane & 31) + blk;
          if (d >= F || c >= a.out) continue;
          const float x = d == 0 ? a.b[c] : a.w[(size_t)c * a.in + row_of(d)];
          const uint16_t hh = f32_to_bf16(x);
          const float r1 = x - bf16_to_f32(hh);
          const uint16_t mm = f32_to_bf16(r1);
          const uint16_t ll = f32_to_bf16(r1 - bf16_to_f32(mm));
          const size_t base = ((size_t)T * 4 + blk) * 3;
          w[((base + 0) * 64 + lane) * 4 + j] = hh;
          w[((base + 1) * 64 + lane) * 4 + j] = mm;
          w[((base + 2) * 64 + lane) * 4 + j] = ll;
        }
  std::vector<float> f(w.size() / 2);
  memcpy(f.data(), w.data(), w.size() * 2);
  return f;
}
std::vector<float> arows_dense_frag(const HostAffine &a) {
  return dense_triple_frag(a, (uint32_t)oak::AR_KT, (uint32_t)oak::AR_FIXED, [](uint32_t d) { return (uint32_t)oak::ar_dense_row((int)d); });
}
// the same for k_embed_prows: dense feature d = 0 is the bias, d = 1..5 are W0^T rows 0..4 (the stats)
std::vector<float> prows_dense_frag(const HostAffine &a) {
  return dense_triple_frag(a, (uint32_t)oak::PR_KT, 6u, [](uint32_t d) { return d - 1; });
}
// ... and W1 [out][hidden] in MFMA-fragment order: [n-block][k-step s][lane (r32, hh)] = W1[nb * 32 + r32][ar_channel(s, hh)]
std::vector<float> embed_frag_order(const HostAffine &a) {
  const uint32_t NB = (a.out + 31) / 32;
  std::vector<float> f((size_t)NB * 64 * 64, 0.0f);
  for (uint32_t nb = 0; nb < NB; ++nb)
    for (uint32_t s2 = 0; s2 < 64; ++s2)
      for (uint32_t lane = 0; lane < 64; ++lane) {
        const uint32_t o = nb * 32 + (lane & 31), c = (uint32_t)oak::ar_channel((int)s2, (int)(lane >> 5));
        if (o < a.out && c < a.in) f[((size_t)nb * 64 + s2) * 64 + lane] = a.w[(size_t)o * a.in + c];
      }
  return f;
}

// fc3 of a policy head as k_policy_rows keeps it in LDS: 315 rows of PH floats at a stride of PH + 4, then the 315 biases (padded to 320)
std::vector<float> policy_rows_image(const HostAffine &b, uint32_t PH) {
  const uint32_t stride = PH + 4;
  std::vector<float> img((size_t)315 * stride + 320, 0.0f);
  for (uint32_t r = 0; r < 315 && r < b.out; ++r) {
    for (uint32_t c = 0; c < b.in && c < PH; ++c) img[(size_t)r * stride + c] = b.w[(size_t)r * b.in + c];
    img[(size_t)315 * stride + r] = b.b[r];
  }
  return img;
}
std::vector<float> policy_triple_order(const HostAffine &a, uint32_t H, uint32_t PB) {
  const uint32_t nT = H / 16;
  std::vector<uint16_t> w((size_t)nT * PB * 3 * 64 * 8, 0);
  for (uint32_t T = 0; T < nT; ++T)
    for (uint32_t b = 0; b < PB; ++b)
      for (uint32_t lane = 0; lane < 64; ++lane)
        for (uint32_t j = 0; j < 8; ++j) {
          const uint32_t row = 32 * b + (lane & 31), col = 16 * T + 8 * (lane >> 5) + j;
          if (row >= a.out || col >= a.in) continue;
          const float x = a.w[(size_t)row * a.in + col];
          const uint16_t hh = f32_to_bf16(x);
          const float r1 = x - bf16_to_f32(hh);
          const uint16_t mm = f32_to_bf16(r1);
          const uint16_t ll = f32_to_bf16(r1 - bf16_to_f32(mm));
          const size_t base = ((size_t)T * PB + b) * 3;
          w[((base + 0) * 64 + lane) * 8 + j] = hh;
          w[((base + 1) * 64 + lane) * 8 + j] = mm;
          w[((base + 2) * 64 + lane) * 8 + j] = ll;
        }
  std::vector<float> f(w.size() / 2);
  memcpy(f.data(), w.data(), w.size() * 2);
  return f;
}

std::vector<float> pad_vec(const std::vector<float> &v, uint32_t n) {
  std::vector<float> t(n, 0.0f);
  for (size_t i = 0; i < v.size(); ++i) t[i] = v[i];
  return t;
}
uint32_t up32(uint32_t x) { return (x + 31) & ~31u; }

// W1 of an embedding net as scaled fp16 PAIRS in embed_layer2's order (round 5; rounds 3-4: bf16 triples -- the image keeps their three
// 1-KB parts per k-step, the third is zeros): 16-bit word ((((nb * 8 + T) * 3 + part) * 64 + lane) * 8 + j) = part (0: h, 1: l) of
// W1[32 nb + (lane & 31)][ar_channel(8 T + j, lane >> 5)] x row_pair_scale(W1, that row); absent rows / channels are zeros.  Returned as floats
// (the LDS image's unit).
std::vector<float> embed_pair_order(const HostAffine &a, uint32_t NB) {
  std::vector<uint16_t> w((size_t)NB * 8 * 3 * 64 * 8, 0);
  for (uint32_t nb = 0; nb < NB; ++nb)
    for (uint32_t T = 0; T < 8; ++T)
      for (uint32_t lane = 0; lane < 64; ++lane)
        for (uint32_t j = 0; j < 8; ++j) {
          const uint32_t o = nb * 32 + (lane & 31), c = (uint32_t)oak::ar_channel((int)(8 * T + j), (int)(lane >> 5));
          if (o >= a.out || c >= a.in) continue;
          const float x = a.w[(size_t)o * a.in + c] * row_pair_scale(a, o);
          const _Float16 hi = (_Float16)x;
          const _Float16 lo = (_Float16)(x - (float)hi);
          const size_t base = ((size_t)nb * 8 + T) * 3;
          w[((base + 0) * 64 + lane) * 8 + j] = f16_bits(hi);
          w[((base + 1) * 64 + lane) * 8 + j] = f16_bits(lo);
        }
  std::vector<float> f(w.size() / 2);
  memcpy(f.data(), w.data(), w.size() * 2);
  return f;
}
std::vector<float> row_scale_inverses(const HostAffine &a, uint32_t padded) { // 1 / row_pair_scale of every output row (padded rows: 0)
  std::vector<float> v(padded, 0.0f);
  for (uint32_t o = 0; o < a.out; ++o) v[o] = 1.0f / row_pair_scale(a, o);
  return v;
}

// the images of the two kernels' LDS weights: [one-hot rows (stride ER_RS) + a zero row | dense fragment | W1's fp16 pairs | b1 | 1 / W1's row scales]
std::vector<float> arows_image(const HostAffine &a0, const HostAffine &a1) {
  std::vector<float> img((size_t)(oak::AR_SPARSE + 1) * oak::ER_RS, 0.0f);
  for (int sl = 0; sl < oak::AR_SPARSE; ++sl)
    for (uint32_t c = 0; c < a0.out && c < 128; ++c) img[(size_t)sl * oak::ER_RS + c] = a0.w[(size_t)c * a0.in + (uint32_t)oak::ar_sparse_row(sl)];
  const std::vector<float> d = arows_dense_frag(a0), f = embed_pair_order(a1, (a1.out + 31) / 32), bp = pad_vec(a1.b, up32(a1.out)), iv = row_scale_inverses(a1, up32(a1.out));
  img.insert(img.end(), d.begin(), d.end());
  img.insert(img.end(), f.begin(), f.end());
  img.insert(img.end(), bp.begin(), bp.end());
  img.insert(img.end(), iv.begin(), iv.end());
  return img;
}
std::vector<float> prows_image(const HostAffine &p0, const HostAffine &p1) {
  std::vector<float> img((size_t)(oak::PR_SPARSE + 1) * oak::ER_RS, 0.0f);
  for (int sl = 0; sl < oak::PR_SPARSE; ++sl)
    for (uint32_t c = 0; c < p0.out && c < 128; ++c) img[(size_t)sl * oak::ER_RS + c] = p0.w[(size_t)c * p0.in + (uint32_t)(sl + 5)];
  const std::vector<float> d = prows_dense_frag(p0), f = embed_pair_order(p1, (p1.out + 31) / 32), bp = pad_vec(p1.b, up32(p1.out)), iv = row_scale_inverses(p1, up32(p1.out));
  img.insert(img.end(), d.begin(), d.end());
  img.insert(img.end(), f.begin(), f.end());
  img.insert(img.end(), bp.begin(), bp.end());
  img.insert(img.end(), iv.begin(), iv.end());
  return img;
}

} // namespace

extern "C" {

void oakgpu_net_free(oakgpu_ctx *ctx, oakgpu_net *net) {
  (void)ctx;
  if (!net) return;
  (void)hipSetDevice(net->device);
  for (void *p : net->allocs) (void)hipFree(p);
  delete net;
}

int oakgpu_net_load_memory(oakgpu_ctx *ctx, const void *bytes, size_t size, oakgpu_net **out) {
  if (!ctx || !bytes || !out) return oakgpu_fail_msg("oakgpu_net_load_memory: null argument");
  const uint8_t *p = (const uint8_t *)bytes, *end = p + size;
  if (size < 8) return oakgpu_fail_msg("network file: truncated header");
  const int activation = (int)p[0] + 1; // search.cc:127-131: byte0 = activation - 1
  if (activation != 1 && activation != 2) return oakgpu_fail_msg("network file: unknown activation byte");
  p += 8;
  HostAffine L[12]; // p0 p1 a0 a1 fc0 fc1 v2 v3 q1a q1b q2a q2b
  for (int i = 0; i < 12; ++i)
    if (!read_affine(p, end, L[i])) return oakgpu_fail_msg("network file: truncated or malformed layer");
  if (p != end) return oakgpu_fail_msg("network file: trailing bytes (network.h:60-63)");
  // Non-finite parameters are refused.  The reference would load them and propagate NaN / inf through every inference (its
  // value_inference asserts !isnan in debug builds, network.h:77); a file with such a parameter is a failed training run, and
  // here an infinite weight would also split into (inf, NaN, NaN) on the bf16 pipe -- a different wrong answer than fp32's.
  {
    static const char *names[12] = {"pokemon_net.fc0", "pokemon_net.fc1", "active_net.fc0", "active_net.fc1", "main_net.fc0", "main_net.fc1", "main_net.value_fc2",
                                    "main_net.value_fc3", "main_net.p1_policy_fc2", "main_net.p1_policy_fc3", "main_net.p2_policy_fc2", "main_net.p2_policy_fc3"};
    for (int i = 0; i < 12; ++i) {
      bool finite = true;
      for (float v : L[i].b) finite = finite && std::isfinite(v);
      for (float v : L[i].w) finite = finite && std::isfinite(v);
      if (!finite) return oakgpu_fail_msg((std::string("network file: non-finite parameter (NaN / inf) in ") + names[i]).c_str());
    }
  }
  const HostAffine &p0 = L[0], &p1 = L[1], &a0 = L[2], &a1 = L[3], &fc0 = L[4], &fc1 = L[5], &v2 = L[6], &v3 = L[7];
  if (p0.in != 198 || a0.in != 427) return oakgpu_fail_msg("network file: embedding input dims must be 198 / 427");
  if (p1.in != p0.out || a1.in != a0.out || fc1.in != fc0.out || v2.in != fc1.out || v3.in != v2.out || v3.out != 1)
    return oakgpu_fail_msg("network file: inconsistent layer dims");
  if (p0.out > 128 || a0.out > 128 || p1.out > 128 || a1.out > 128) return oakgpu_fail_msg("embedding widths above 128 unsupported");
  const uint32_t side_dim = (1 + a1.out) + 5 * (1 + p1.out);
  if (fc0.in != 2 * side_dim) return oakgpu_fail_msg("network file: fc0 input != 2 * side embedding");
  if (fc0.out != fc1.out) return oakgpu_fail_msg("network file: fc0/fc1 widths differ");
  const uint32_t H = up32(fc0.out), VH = up32(v2.out);
  if (H > (uint32_t)oak::MAXH || VH > (uint32_t)oak::MAXH) return oakgpu_fail_msg("main-net widths above 256 unsupported");
  if (fc0.in % 4) return oakgpu_fail_msg("embedding dim must be a multiple of 4");
  hipError_t he = hipSetDevice(oakgpu_ctx_device(ctx));
  if (he != hipSuccess) return oakgpu_fail_hip((int)he, "hipSetDevice");
  oakgpu_net *net = new oakgpu_net();
  net->device = oakgpu_ctx_device(ctx);
  net->in_dim = (int)fc0.in;
  net->hidden = (int)fc0.out;
  net->value_hidden = (int)v2.out;
  net->policy_hidden = (int)L[8].out;
  oak::NetDev &D = net->dev;
  D.activation = activation;
  D.p_hidden = (int)p0.out; D.p_out = (int)p1.out; D.a_hidden = (int)a0.out; D.a_out = (int)a1.out;
  D.side_dim = (int)side_dim; D.emb_dim = (int)fc0.in;
  D.H = (int)H; D.VH = (int)VH;
  D.b3 = v3.b[0];
  int rc = 0;
  rc = rc ? rc : upload(net, transpose(p0), &D.p_w0t);
  rc = rc ? rc : upload(net, p0.b, &D.p_b0);
  rc = rc ? rc : upload(net, p1.w, &D.p_w1);
  rc = rc ? rc : upload(net, a1.w, &D.a_w1);
  rc = rc ? rc : upload(net, pad_vec(p1.b, up32(p1.out)), &D.p_b1); // (padded: embed_scatter reads whole float4 groups)
  rc = rc ? rc : upload(net, transpose(a0), &D.a_w0t);
  rc = rc ? rc : upload(net, a0.b, &D.a_b0);
  rc = rc ? rc : upload(net, pad_vec(a1.b, up32(a1.out)), &D.a_b1);
  rc = rc ? rc : upload(net, arows_rows(a0), &D.a_w0d);
  rc = rc ? rc : upload(net, arows_image(a0, a1), &D.a_img);
  rc = rc ? rc : upload(net, prows_image(p0, p1), &D.p_img);
  rc = rc ? rc : upload(net, pad_vec(fc0.b, H), &D.b0);
  rc = rc ? rc : upload(net, pad_vec(fc1.b, H), &D.b1);
  rc = rc ? rc : upload(net, pad_vec(v2.b, VH), &D.b2);
  rc = rc ? rc : upload(net, pad_vec(v3.w, VH), &D.w3);
  rc = rc ? rc : upload(net, frag_order_wave(fc0, H), &D.w0g);
  rc = rc ? rc : upload(net, frag_order_wave(fc1, H), &D.w1g);
  rc = rc ? rc : upload(net, frag_order_wave(v2, VH), &D.w2g);
  {
    const uint32_t nbr = (H > VH ? H : VH) / 32, NB = nbr > 4 ? 8 : nbr > 2 ? 4 : nbr, T0 = 4 * ((fc0.in + 63) / 64);
    const std::vector<uint16_t> ws = split_stream(fc0, fc1, v2, NB, T0);
    void *dptr = nullptr;
    if (!rc) {
      he = hipMalloc(&dptr, ws.size() * 2);
      if (he != hipSuccess) rc = oakgpu_fail_hip((int)he, "hipMalloc(weight stream)");
      else {
        net->allocs.push_back(dptr);
        he = hipMemcpy(dptr, ws.data(), ws.size() * 2, hipMemcpyHostToDevice);
        if (he != hipSuccess) rc = oakgpu_fail_hip((int)he, "hipMemcpy(weight stream)");
      }
    }
    D.ws = (const uint16_t *)dptr; D.ws_T0 = (int)T0; D.ws_NB = (int)NB;
    {
      std::vector<float> wp_inv;
      const std::vector<uint16_t> wp = pair_stream(fc0, fc1, v2, NB, T0, (uint32_t)H, (uint32_t)VH, wp_inv);
      rc = rc ? rc : upload(net, wp_inv, &D.wp_inv);
      void *pptr = nullptr;
      if (!rc) {
        he = hipMalloc(&pptr, wp.size() * 2);
        if (he != hipSuccess) rc = oakgpu_fail_hip((int)he, "hipMalloc(pair stream)");
        else {
          net->allocs.push_back(pptr);
          he = hipMemcpy(pptr, wp.data(), wp.size() * 2, hipMemcpyHostToDevice);
          if (he != hipSuccess) rc = oakgpu_fail_hip((int)he, "hipMemcpy(pair stream)");
        }
      }
      D.wp = (const uint16_t *)pptr;
    }
    // The bf16 triple (h, m, l) of a value x is exact to 2^-24 |x| only while its low parts are normal bf16 numbers (|x| >= ~2^-102);
    // below that a part that flushes loses at most 2^-126 per factor -- an ABSOLUTE error of at most 2^-126 |other factor| per
    // product.  That is harmless unless later layers amplify it: a layer scaled by 2^-120 feeding one scaled by 2^+120 computes an
    // O(1) value in fp32, and there the lost parts would come back multiplied by 2^120.  So the triple form is used only while
    // no main-net weight exceeds 2^20 in magnitude (two layers of 256 such weights amplify by < 2^56: 2^-126 x 768 x 2^56 is
    // nothing); a network with a larger weight runs its main net on fp32 MFMA, whose products need no such care.
    bool safe = true;
    for (const HostAffine *a : {&fc0, &fc1, &v2})
      for (float v : a->w) safe = safe && std::fabs(v) <= 0x1p20f;
    net->split_safe = safe;
    // The embedding passes (k_embed_prows / k_embed_arows) multiply as bf16 triples too (round 4).  What a flushed part loses there is
    // amplified by whatever comes BEHIND it -- the embedding nets' own second layers (L[1], L[3]) and the main net -- so a network
    // with a weight above 2^20 in any of those runs its embedding nets through k_embed_lds (fp32 MFMA), whatever the main net's mode
    // (round-4 advice: a layer scaled by 2^-110 in front of one scaled by 2^+110 is the same function in fp32).
    bool esafe = safe;
    for (const HostAffine *a : {&L[1], &L[3]})
      for (float v : a->w) esafe = esafe && std::fabs(v) <= 0x1p20f;
    // (round 5) ... and their second layers multiply as scaled fp16 pairs, which both must survive (pair_layer_ok: rows and columns)
    esafe = esafe && pair_layer_ok(L[1]) && pair_layer_ok(L[3]);
    net->embed_safe = esafe;
    // fp16 pairs (k_mainnet_pair, the default since round 5): scaled per weight row and per batch row, so no absolute magnitude matters;
    // what must hold is that every weight row survives the pairing to fp32 accuracy and no column is dwarfed (pair_layer_ok)
    net->pair_safe = pair_layer_ok(fc0) && pair_layer_ok(fc1) && pair_layer_ok(v2);
    const char *env = getenv("OAKGPU_MAIN_NET");
    const int want = env ? (strcmp(env, "fp32") == 0 ? 0 : strcmp(env, "bf16x3") == 0 ? 1 : 2) : 2;
    net->main_mode = want == 2 && net->pair_safe ? 2 : want >= 1 && safe ? 1 : 0;
  }
  {
    const HostAffine &q1a = L[8], &q1b = L[9], &q2a = L[10], &q2b = L[11];
    if (q1a.in != fc1.out || q2a.in != fc1.out || q1b.in != q1a.out || q2b.in != q2a.out || q1a.out != q2a.out ||
        q1b.out != 315 || q2b.out != 315) { oakgpu_net_free(ctx, net); return oakgpu_fail_msg("network file: inconsistent policy-head dims"); }
    const uint32_t ph32 = up32(q1a.out);
    if (ph32 > (uint32_t)oak::MAXH) { oakgpu_net_free(ctx, net); return oakgpu_fail_msg("policy hidden width above 256 unsupported"); }
    const uint32_t PB = ph32 > 128 ? 8 : ph32 > 64 ? 4 : ph32 > 32 ? 2 : 1, PH = 32 * PB; // k_policy_rows' block count: absent features are zeros
    D.PH = (int)PH;
    rc = rc ? rc : upload(net, pad_rows(q1a, PH, H), &D.q1a);
    rc = rc ? rc : upload(net, pad_vec(q1a.b, PH), &D.q1a_b);
    rc = rc ? rc : upload(net, pad_rows(q1b, 315, PH), &D.q1b);
    rc = rc ? rc : upload(net, q1b.b, &D.q1b_b);
    rc = rc ? rc : upload(net, pad_rows(q2a, PH, H), &D.q2a);
    rc = rc ? rc : upload(net, pad_vec(q2a.b, PH), &D.q2a_b);
    rc = rc ? rc : upload(net, pad_rows(q2b, 315, PH), &D.q2b);
    rc = rc ? rc : upload(net, q2b.b, &D.q2b_b);
    rc = rc ? rc : upload(net, policy_frag_order(q1a, H, PB), &D.q1a_f);
    rc = rc ? rc : upload(net, policy_frag_order(q2a, H, PB), &D.q2a_f);
    if (PH <= 64) { // (PolicyRows<PB>::WB_LDS: the rows fit the CU's LDS)
      rc = rc ? rc : upload(net, policy_rows_image(q1b, PH), &D.q1b_img);
      rc = rc ? rc : upload(net, policy_rows_image(q2b, PH), &D.q2b_img);
    }
    // the triples only while no fc2 weight is above 2^20 in magnitude (as for the main net: split_safe)
    bool psafe = true;
    for (float v : q1a.w) psafe = psafe && std::fabs(v) <= 1048576.0f;
    for (float v : q2a.w) psafe = psafe && std::fabs(v) <= 1048576.0f;
    if (psafe) {
      rc = rc ? rc : upload(net, policy_triple_order(q1a, H, PB), &D.q1a_t);
      rc = rc ? rc : upload(net, policy_triple_order(q2a, H, PB), &D.q2a_t);
    }
  }
  if (rc) { oakgpu_net_free(ctx, net); return rc; }
  *out = net;
  return 0;
}

int oakgpu_net_load(oakgpu_ctx *ctx, const char *path, oakgpu_net **out) {
  if (!path) return oakgpu_fail_msg("oakgpu_net_load: null path");
  FILE *f = fopen(path, "rb");
  if (!f) return oakgpu_fail_msg((std::string("Cannot open network file: ") + path).c_str());
  std::vector<uint8_t> buf;
  uint8_t tmp[1 << 16];
  size_t r;
  while ((r = fread(tmp, 1, sizeof tmp, f)) > 0) buf.insert(buf.end(), tmp, tmp + r);
  fclose(f);
  return oakgpu_net_load_memory(ctx, buf.data(), buf.size(), out);
}

int oakgpu_net_set_main_precision(oakgpu_net *net, int mode) {
  if (!net || (mode != OAKGPU_MAIN_FP32 && mode != OAKGPU_MAIN_SPLIT && mode != OAKGPU_MAIN_PAIR)) { oakgpu_fail_msg("oakgpu_net_set_main_precision: bad argument"); return -1; }
  const int prev = net->main_mode;
  // (a request the network's weights do not allow is not honoured: fp16 pairs fall back to bf16 triples, those -- a main-net weight
  // above 2^20 in magnitude -- to fp32 MFMA; the returned previous mode and oakgpu_net_main_precision say what runs)
  if (mode == OAKGPU_MAIN_PAIR && !net->pair_safe) mode = OAKGPU_MAIN_SPLIT;
  net->main_mode = (mode == OAKGPU_MAIN_SPLIT && !net->split_safe) ? OAKGPU_MAIN_FP32 : mode;
  return prev;
}

int oakgpu_net_main_precision(const oakgpu_net *net, int *split_allowed) {
  if (!net) { oakgpu_fail_msg("oakgpu_net_main_precision: null net"); return -1; }
  if (split_allowed) *split_allowed = net->split_safe ? 1 : 0;
  return net->main_mode;
}

int oakgpu_net_shape(const oakgpu_net *net, int *in_dim, int *hidden, int *value_hidden, int *policy_hidden) {
  if (!net) return oakgpu_fail_msg("null net");
  if (in_dim) *in_dim = net->in_dim;
  if (hidden) *hidden = net->hidden;
  if (value_hidden) *value_hidden = net->value_hidden;
  if (policy_hidden) *policy_hidden = net->policy_hidden;
  return 0;
}

int oakgpu_leaf_set_lds_limits(void) { // per DEVICE (hipFuncSetAttribute applies to the current device): called by oakgpu_create
  hipError_t e = hipSuccess;
  e = hipFuncSetAttribute((const void *)oak::k_embed_lds<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)oak::ELayout<false>::BYTES);
  if (e != hipSuccess) return oakgpu_fail_hip((int)e, "hipFuncSetAttribute(k_embed_lds<party>)");
  e = hipFuncSetAttribute((const void *)oak::k_embed_lds<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)oak::ELayout<false>::BYTES);
  if (e != hipSuccess) return oakgpu_fail_hip((int)e, "hipFuncSetAttribute(k_embed_lds<party, list>)");
  e = hipFuncSetAttribute((const void *)oak::k_embed_arows, hipFuncAttributeMaxDynamicSharedMemorySize, (int)oak::ar_bytes(oak::AR_MAX_NBO));
  if (e != hipSuccess) return oakgpu_fail_hip((int)e, "hipFuncSetAttribute(k_embed_arows)");
  e = hipFuncSetAttribute((const void *)oak::k_embed_prows<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)oak::PR_BYTES);
  if (e != hipSuccess) return oakgpu_fail_hip((int)e, "hipFuncSetAttribute(k_embed_prows)");
  {
    const int both_bytes = (int)(oak::PR_BYTES > oak::ar_bytes(oak::AR_MAX_NBO) ? oak::PR_BYTES : oak::ar_bytes(oak::AR_MAX_NBO));
    e = hipFuncSetAttribute((const void *)oak::k_embed_both<false>, hipFuncAttributeMaxDynamicSharedMemorySize, both_bytes);
    if (e != hipSuccess) return oakgpu_fail_hip((int)e, "hipFuncSetAttribute(k_embed_both)");
    e = hipFuncSetAttribute((const void *)oak::k_embed_both<true>, hipFuncAttributeMaxDynamicSharedMemorySize, both_bytes);
    if (e != hipSuccess) return oakgpu_fail_hip((int)e, "hipFuncSetAttribute(k_embed_both<list>)");
  }
  e = hipFuncSetAttribute((const void *)oak::k_embed_prows<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)oak::PR_BYTES);
  if (e != hipSuccess) return oakgpu_fail_hip((int)e, "hipFuncSetAttribute(k_embed_prows<list>)");
  e = hipFuncSetAttribute((const void *)oak::k_embed_lds<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)oak::ELayout<true>::BYTES);
  if (e != hipSuccess) return oakgpu_fail_hip((int)e, "hipFuncSetAttribute(k_embed_lds<active>)");
#define OAK_POLICY_ATTR(PBV)                                                                                                                              \
  if (e == hipSuccess) e = hipFuncSetAttribute((const void *)oak::k_policy_rows<PBV, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)oak::PolicyRows<PBV>::LDS); \
  if (e == hipSuccess) e = hipFuncSetAttribute((const void *)oak::k_policy_rows<PBV, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)oak::PolicyRows<PBV>::LDS);
  e = hipSuccess;
  OAK_POLICY_ATTR(1) OAK_POLICY_ATTR(2) OAK_POLICY_ATTR(4) OAK_POLICY_ATTR(8)
#undef OAK_POLICY_ATTR
  if (e != hipSuccess) return oakgpu_fail_hip((int)e, "hipFuncSetAttribute(k_policy_rows)");
  e = hipFuncSetAttribute((const void *)oak::k_mainnet_wave, hipFuncAttributeMaxDynamicSharedMemorySize, (int)oak::MW_BYTES);
  if (e != hipSuccess) return oakgpu_fail_hip((int)e, "hipFuncSetAttribute(k_mainnet_wave)");
  e = hipFuncSetAttribute((const void *)oak::k_mainnet_split<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)oak::MSplit<8>::LDS);
  if (e == hipSuccess) e = hipFuncSetAttribute((const void *)oak::k_mainnet_split<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)oak::MSplit<4>::LDS);
  if (e == hipSuccess) e = hipFuncSetAttribute((const void *)oak::k_mainnet_split<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)oak::MSplit<2>::LDS);
  if (e == hipSuccess) e = hipFuncSetAttribute((const void *)oak::k_mainnet_split<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)oak::MSplit<1>::LDS);
  if (e != hipSuccess) return oakgpu_fail_hip((int)e, "hipFuncSetAttribute(k_mainnet_split)");
  e = hipFuncSetAttribute((const void *)oak::k_mainnet_pair<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)oak::MPair<8>::LDS);
  if (e == hipSuccess) e = hipFuncSetAttribute((const void *)oak::k_mainnet_pair<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)oak::MPair<4>::LDS);
  if (e == hipSuccess) e = hipFuncSetAttribute((const void *)oak::k_mainnet_pair<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)oak::MPair<2>::LDS);
  if (e == hipSuccess) e = hipFuncSetAttribute((const void *)oak::k_mainnet_pair<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)oak::MPair<1>::LDS);
  if (e != hipSuccess) return oakgpu_fail_hip((int)e, "hipFuncSetAttribute(k_mainnet_pair)");
  return 0;
}

static int leaf_eval_impl(oakgpu_ctx *ctx, oakgpu_net *net, const uint8_t *battles, const uint8_t *durations, uint32_t n,
                          float *values, float *embedding_out, const oak::PolicyArgs *pol, uint32_t *slot_tags = nullptr) {
  if (!ctx || !net) return oakgpu_fail_msg("oakgpu_leaf_eval_dev: null ctx/net");
  if (n == 0) return 0;
  if (!battles || !durations || !values) return oakgpu_fail_msg("oakgpu_leaf_eval_dev: null required pointer");
  if (net->device != oakgpu_ctx_device(ctx)) return oakgpu_fail_msg("oakgpu_leaf_eval_dev: the network was loaded on another device than the context's");
  if (int rc = oakgpu_ctx_enter(ctx)) return rc;
  hipStream_t stream = (hipStream_t)oakgpu_ctx_stream(ctx);
  float *emb = embedding_out;
  if (!emb) { // per-context (= per-stream) workspace: contexts sharing a network never share scratch memory
    emb = (float *)oakgpu_ctx_workspace(ctx, 0, (size_t)n * net->dev.emb_dim * 4);
    if (!emb) return -1;
  }
  const oak::NetDev &D = net->dev;
  // The row kernels (k_embed_prows / k_embed_arows) take embedding nets up to 128 hidden channels, party outputs up to 64
  // and active outputs up to 128; anything wider goes to k_embed_lds (the 64-item tile form).  OAKGPU_EMBED_TILE=1 forces
  // the tile form (A/B and a second implementation for the tests).
  static const bool force_tile = getenv("OAKGPU_EMBED_TILE") != nullptr;
  const bool rows = !force_tile && net->embed_safe; // (the row kernels multiply as bf16 triples: see embed_safe)
  const bool prow_ok = rows && D.p_hidden <= 128 && D.p_out <= 64, arow_ok = rows && D.a_hidden <= 128 && D.a_out <= 32 * oak::AR_MAX_NBO;
  hipEvent_t *tev = (hipEvent_t *)oakgpu_ctx_timing_events(ctx); // diagnostic only (oakgpu_set_kernel_timing)
  static const int kinds = getenv("OAKGPU_EMBED_KINDS") ? atoi(getenv("OAKGPU_EMBED_KINDS")) : 3; // diagnostics: 1 party, 2 actives
  static const bool split = getenv("OAKGPU_EMBED_SPLIT") != nullptr; // A/B: the two passes as two launches
  oak::EmbedTileArgs tp{D, battles, durations, n, emb, 0, nullptr, nullptr}, tact{D, battles, durations, n, emb, 1, nullptr, nullptr};
  if (slot_tags && (kinds & 1)) { // cached party-slot pass: tag comparison first, then only the changed slots (work list)
    uint8_t *ws = (uint8_t *)oakgpu_ctx_workspace(ctx, 2, (size_t)n * 10 * sizeof(oak::PartyWork) + 16);
    if (!ws) return -1;
    uint32_t *count = (uint32_t *)ws;
    oak::PartyWork *work = (oak::PartyWork *)(ws + 16);
    hipError_t me = hipMemsetAsync(count, 0, 4, stream);
    if (me != hipSuccess) return oakgpu_fail_hip((int)me, "hipMemsetAsync(work count)");
    if (tev) (void)hipEventRecord(tev[0], stream);
    hipLaunchKernelGGL(oak::k_party_tags, dim3((n * 10 + 256 * oak::TAG_R - 1) / (256 * oak::TAG_R)), dim3(256), 0, stream, D, battles, durations, n, emb, slot_tags, work, count);
    tp.work = work;
    tp.work_count = count;
  } else if (tev) (void)hipEventRecord(tev[0], stream);
  const uint32_t nmt_p = (n * 10 + oak::ER_ITEMS - 1) / oak::ER_ITEMS, wg_p0 = (nmt_p + oak::PR_WAVES - 1) / oak::PR_WAVES;
  const uint32_t wg_p = slot_tags ? 256u : (wg_p0 < 256 ? wg_p0 : 256); // (a work list's length is only known on the device)
  const uint32_t nmt_a = (n * 2 + oak::ER_ITEMS - 1) / oak::ER_ITEMS, wg_a0 = (nmt_a + oak::AR_WAVES - 1) / oak::AR_WAVES, wg_a = wg_a0 < 256 ? wg_a0 : 256;
  const size_t ar_lds = oak::ar_bytes((D.a_out + 31) / 32);
  // default: both embedding passes in one launch (k_embed_both).  Not while the per-kernel timing diagnostic is on (it
  // wants an event between the passes).
  if (prow_ok && arow_ok && kinds == 3 && !split && !tev) {
    const size_t lds = oak::PR_BYTES > ar_lds ? oak::PR_BYTES : ar_lds;
    if (slot_tags) hipLaunchKernelGGL(oak::k_embed_both<true>, dim3(wg_p + wg_a), dim3(oak::PR_BLOCK), lds, stream, tp, tact, wg_p);
    else hipLaunchKernelGGL(oak::k_embed_both<false>, dim3(wg_p + wg_a), dim3(oak::PR_BLOCK), lds, stream, tp, tact, wg_p);
  } else {
    if (kinds & 1) {
      const uint32_t ntiles = (n * 10u + oak::ET - 1) / oak::ET, grid = ntiles < 256 ? ntiles : 256;
      // (the work list goes through the same kernel as the plain pass, so that cached and plain embeddings are bit-identical)
      if (prow_ok && slot_tags) hipLaunchKernelGGL(oak::k_embed_prows<true>, dim3(wg_p), dim3(oak::PR_BLOCK), oak::PR_BYTES, stream, tp);
      else if (prow_ok) hipLaunchKernelGGL(oak::k_embed_prows<false>, dim3(wg_p), dim3(oak::PR_BLOCK), oak::PR_BYTES, stream, tp);
      else if (slot_tags) hipLaunchKernelGGL((oak::k_embed_lds<false, true>), dim3(grid), dim3(oak::EL_BLOCK), oak::ELayout<false>::BYTES, stream, tp);
      else hipLaunchKernelGGL(oak::k_embed_lds<false>, dim3(grid), dim3(oak::EL_BLOCK), oak::ELayout<false>::BYTES, stream, tp);
    }
    if (tev) (void)hipEventRecord(tev[1], stream);
    if (kinds & 2) {
      const uint32_t ntiles = (n * 2u + oak::ET - 1) / oak::ET, grid = ntiles < 256 ? ntiles : 256;
      if (arow_ok) hipLaunchKernelGGL(oak::k_embed_arows, dim3(wg_a), dim3(oak::AR_BLOCK), ar_lds, stream, tact);
      else hipLaunchKernelGGL(oak::k_embed_lds<true>, dim3(grid), dim3(oak::EL_BLOCK), oak::ELayout<true>::BYTES, stream, tact);
    }
  }
  float *h1 = nullptr;
  if (pol) {
    h1 = (float *)oakgpu_ctx_workspace(ctx, 1, (size_t)n * D.H * 4);
    if (!h1) return -1;
  }
  if (tev) (void)hipEventRecord(tev[2], stream);
  oak::MainArgs ma{D, emb, n, values, h1};
  {
    const uint32_t wgs = ((n + 31) / 32 + 3) / 4, grid = wgs < 256 ? wgs : 256;
    if (net->main_mode == 0) hipLaunchKernelGGL(oak::k_mainnet_wave, dim3(grid), dim3(oak::MN_BLOCK), oak::MW_BYTES, stream, ma);
    else if (net->main_mode == 2 && D.ws_NB == 8) hipLaunchKernelGGL(oak::k_mainnet_pair<8>, dim3(grid), dim3(oak::MN_BLOCK), oak::MPair<8>::LDS, stream, ma);
    else if (net->main_mode == 2 && D.ws_NB == 4) hipLaunchKernelGGL(oak::k_mainnet_pair<4>, dim3(grid), dim3(oak::MN_BLOCK), oak::MPair<4>::LDS, stream, ma);
    else if (net->main_mode == 2 && D.ws_NB == 2) hipLaunchKernelGGL(oak::k_mainnet_pair<2>, dim3(grid), dim3(oak::MN_BLOCK), oak::MPair<2>::LDS, stream, ma);
    else if (net->main_mode == 2) hipLaunchKernelGGL(oak::k_mainnet_pair<1>, dim3(grid), dim3(oak::MN_BLOCK), oak::MPair<1>::LDS, stream, ma);
    else if (D.ws_NB == 8) hipLaunchKernelGGL(oak::k_mainnet_split<8>, dim3(grid), dim3(oak::MN_BLOCK), oak::MSplit<8>::LDS, stream, ma);
    else if (D.ws_NB == 4) hipLaunchKernelGGL(oak::k_mainnet_split<4>, dim3(grid), dim3(oak::MN_BLOCK), oak::MSplit<4>::LDS, stream, ma);
    else if (D.ws_NB == 2) hipLaunchKernelGGL(oak::k_mainnet_split<2>, dim3(grid), dim3(oak::MN_BLOCK), oak::MSplit<2>::LDS, stream, ma);
    else hipLaunchKernelGGL(oak::k_mainnet_split<1>, dim3(grid), dim3(oak::MN_BLOCK), oak::MSplit<1>::LDS, stream, ma);
  }
  if (tev) (void)hipEventRecord(tev[3], stream);
  if (pol) {
    oak::PolicyArgs pa = *pol;
    pa.net = D;
    pa.h1 = h1;
    const uint32_t ptiles = (n + 31) / 32;
    auto pgrid = [&](uint32_t waves) { const uint32_t wgs = (ptiles + waves - 1) / waves; return dim3(wgs < 256 ? wgs : 256); };
    // fc2 as bf16 triples when the main net runs that way and no fc2 weight forbids it (q1a_t), else on fp32 MFMA
    const bool triple = net->main_mode != OAKGPU_MAIN_FP32 && D.q1a_t != nullptr && D.q2a_t != nullptr; // (the heads stay on bf16 triples beside a main net on fp16 pairs)
#define OAK_POLICY_LAUNCH(PBV)                                                                                                                            \
  do {                                                                                                                                                    \
    if (triple) hipLaunchKernelGGL((oak::k_policy_rows<PBV, true>), pgrid(oak::PolicyRows<PBV>::WAVES), dim3(oak::PolicyRows<PBV>::BLOCK), oak::PolicyRows<PBV>::LDS, stream, pa);  \
    else hipLaunchKernelGGL((oak::k_policy_rows<PBV, false>), pgrid(oak::PolicyRows<PBV>::WAVES), dim3(oak::PolicyRows<PBV>::BLOCK), oak::PolicyRows<PBV>::LDS, stream, pa);       \
  } while (0)
    if (D.PH == 256) OAK_POLICY_LAUNCH(8);
    else if (D.PH == 128) OAK_POLICY_LAUNCH(4);
    else if (D.PH == 64) OAK_POLICY_LAUNCH(2);
    else OAK_POLICY_LAUNCH(1);
#undef OAK_POLICY_LAUNCH
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return oakgpu_fail_hip((int)e, "leaf_eval launch");
  return 0;
}

#ifdef OAKGPU_LEAF_PROFILE
int oakgpu_leaf_profile(unsigned long long *out, int reset) { // profile build only
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(oak::g_leaf_prof), sizeof(unsigned long long) * 16) != hipSuccess) return 1;
  if (reset) { unsigned long long z[16] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(oak::g_leaf_prof), z, sizeof z) != hipSuccess) return 1; }
  return 0;
}
#endif

int oakgpu_leaf_eval_dev(oakgpu_ctx *ctx, oakgpu_net *net, const uint8_t *battles, const uint8_t *durations, uint32_t n,
                         float *values, float *embedding_out) {
  return leaf_eval_impl(ctx, net, battles, durations, n, values, embedding_out, nullptr);
}

int oakgpu_leaf_eval_cached_dev(oakgpu_ctx *ctx, oakgpu_net *net, const uint8_t *battles, const uint8_t *durations, uint32_t n,
                                float *values, float *embedding, uint32_t *slot_tags) {
  if (!embedding || !slot_tags) return oakgpu_fail_msg("oakgpu_leaf_eval_cached_dev: the persistent embedding and tag buffers are required");
  return leaf_eval_impl(ctx, net, battles, durations, n, values, embedding, nullptr, slot_tags);
}

int oakgpu_leaf_cache_last_count(oakgpu_ctx *ctx, uint32_t *slots_recomputed) { // diagnostic: synchronises the stream
  if (!ctx || !slots_recomputed) return oakgpu_fail_msg("oakgpu_leaf_cache_last_count: null pointer");
  if (int rc = oakgpu_ctx_enter(ctx)) return rc;
  hipStream_t stream = (hipStream_t)oakgpu_ctx_stream(ctx);
  const uint32_t *count = (const uint32_t *)oakgpu_ctx_workspace(ctx, 2, 16);
  if (!count) return -1;
  hipError_t e = hipMemcpyAsync(slots_recomputed, count, 4, hipMemcpyDeviceToHost, stream);
  if (e == hipSuccess) e = hipStreamSynchronize(stream);
  if (e != hipSuccess) return oakgpu_fail_hip((int)e, "oakgpu_leaf_cache_last_count");
  return 0;
}

int oakgpu_leaf_eval_policy_dev(oakgpu_ctx *ctx, oakgpu_net *net, const uint8_t *battles, const uint8_t *durations, uint32_t n,
                                const uint8_t *p1_choices, const uint8_t *p1_counts, const uint8_t *p2_choices,
                                const uint8_t *p2_counts, float *values, float *p1_logits, float *p2_logits) {
  if (!p1_choices || !p1_counts || !p2_choices || !p2_counts || !p1_logits || !p2_logits)
    return oakgpu_fail_msg("oakgpu_leaf_eval_policy_dev: null pointer");
  oak::PolicyArgs pa{};
  pa.battles = battles;
  pa.choices[0] = p1_choices; pa.choices[1] = p2_choices;
  pa.counts[0] = p1_counts; pa.counts[1] = p2_counts;
  pa.logits[0] = p1_logits; pa.logits[1] = p2_logits;
  pa.n = n;
  return leaf_eval_impl(ctx, net, battles, durations, n, values, nullptr, &pa);
}

#define TRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return oakgpu_fail_hip((int)e_, #x); } while (0)
#define STAGE(var, bytes) void *var = hc.get(bytes); if (!var) return -1

int oakgpu_leaf_eval(oakgpu_ctx *ctx, oakgpu_net *net, const uint8_t *battles, const uint8_t *durations, uint32_t n,
                     float *values, float *embedding_out) {
  if (!ctx || !net) return oakgpu_fail_msg("oakgpu_leaf_eval: null ctx/net");
  if (n == 0) return 0;
  if (!battles || !durations || !values) return oakgpu_fail_msg("oakgpu_leaf_eval: null required pointer");
  if (int rc = oakgpu_ctx_enter(ctx)) return rc;
  hipStream_t stream = (hipStream_t)oakgpu_ctx_stream(ctx);
  OakHostCall hc(ctx); // staging buffers from the context's cache; synchronises the stream on every exit
  STAGE(db, (size_t)n * 384);
  STAGE(dd, (size_t)n * 8);
  STAGE(dv, (size_t)n * 4);
  void *de = nullptr;
  if (embedding_out) { de = hc.get((size_t)n * net->dev.emb_dim * 4); if (!de) return -1; }
  TRY(hipMemcpyAsync(db, battles, (size_t)n * 384, hipMemcpyHostToDevice, stream));
  TRY(hipMemcpyAsync(dd, durations, (size_t)n * 8, hipMemcpyHostToDevice, stream));
  if (int rc = oakgpu_leaf_eval_dev(ctx, net, (const uint8_t *)db, (const uint8_t *)dd, n, (float *)dv, (float *)de)) return rc;
  TRY(hipMemcpyAsync(values, dv, (size_t)n * 4, hipMemcpyDeviceToHost, stream));
  if (embedding_out) TRY(hipMemcpyAsync(embedding_out, de, (size_t)n * net->dev.emb_dim * 4, hipMemcpyDeviceToHost, stream));
  TRY(hipStreamSynchronize(stream));
  return 0;
}

int oakgpu_leaf_eval_policy(oakgpu_ctx *ctx, oakgpu_net *net, const uint8_t *battles, const uint8_t *durations, uint32_t n,
                            const uint8_t *p1_choices, const uint8_t *p1_counts, const uint8_t *p2_choices, const uint8_t *p2_counts,
                            float *values, float *p1_logits, float *p2_logits) {
  if (!ctx || !net) return oakgpu_fail_msg("oakgpu_leaf_eval_policy: null ctx/net");
  if (n == 0) return 0;
  if (!battles || !durations || !values || !p1_choices || !p1_counts || !p2_choices || !p2_counts || !p1_logits || !p2_logits)
    return oakgpu_fail_msg("oakgpu_leaf_eval_policy: null required pointer");
  if (int rc = oakgpu_ctx_enter(ctx)) return rc;
  hipStream_t stream = (hipStream_t)oakgpu_ctx_stream(ctx);
  OakHostCall hc(ctx);
  STAGE(db, (size_t)n * 384); STAGE(dd, (size_t)n * 8); STAGE(dv, (size_t)n * 4);
  STAGE(dc1, (size_t)n * 9); STAGE(dc2, (size_t)n * 9); STAGE(dn1, n); STAGE(dn2, n);
  STAGE(dl1, (size_t)n * 36); STAGE(dl2, (size_t)n * 36);
  TRY(hipMemcpyAsync(db, battles, (size_t)n * 384, hipMemcpyHostToDevice, stream));
  TRY(hipMemcpyAsync(dd, durations, (size_t)n * 8, hipMemcpyHostToDevice, stream));
  TRY(hipMemcpyAsync(dc1, p1_choices, (size_t)n * 9, hipMemcpyHostToDevice, stream));
  TRY(hipMemcpyAsync(dc2, p2_choices, (size_t)n * 9, hipMemcpyHostToDevice, stream));
  TRY(hipMemcpyAsync(dn1, p1_counts, n, hipMemcpyHostToDevice, stream));
  TRY(hipMemcpyAsync(dn2, p2_counts, n, hipMemcpyHostToDevice, stream));
  if (int rc = oakgpu_leaf_eval_policy_dev(ctx, net, (const uint8_t *)db, (const uint8_t *)dd, n, (const uint8_t *)dc1, (const uint8_t *)dn1,
                                           (const uint8_t *)dc2, (const uint8_t *)dn2, (float *)dv, (float *)dl1, (float *)dl2)) return rc;
  TRY(hipMemcpyAsync(values, dv, (size_t)n * 4, hipMemcpyDeviceToHost, stream));
  TRY(hipMemcpyAsync(p1_logits, dl1, (size_t)n * 36, hipMemcpyDeviceToHost, stream));
  TRY(hipMemcpyAsync(p2_logits, dl2, (size_t)n * 36, hipMemcpyDeviceToHost, stream));
  TRY(hipStreamSynchronize(stream));
  return 0;
}
#undef TRY
#undef STAGE

} // extern "C"
