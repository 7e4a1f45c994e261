// oak_amd/csrc/search_host.hip -- batched-leaf tree search over the GPU hot path (host code of liboakgpu.so).
//
// the consumer of the leaf values.  Mirrors MCTS::Search::run / run_root_iteration /
// run_iteration (cpp/include/search/mcts.h:154-389) with joint UCB / PUCB bandits (search/bandit/ucb.h:17-66,
// pucb.h:17-75, search/joint.h:6-52) and a Node tree keyed (p1 index, p2 index, 16-byte observation)
// (mcts.h:93-105).  The reference runs ONE descent at a time and calls libpkmn on the CPU at every tree edge;
// here a batch of B descents walks the tree level by level:
//   * the tree and the bandit statistics live on the host (pointer-chasing, a few bytes per visit);
//   * every battle state lives on the GPU for the whole iteration: root prep (battle.rng = device draw,
//     randomize_hidden_variables; mcts.h:250-263) is the rollout kernel's `prep`, one k_tree_step launch per tree
//     level applies all lanes' joint actions and returns result + observation key + the child's legal choices
//     (one device round trip per level), and the leaves are evaluated in place by the rollout kernel
//     (MCTS::MonteCarlo, mcts.h:448-496) or the network kernels (value / value + policy logits);
//   * lanes of one batch see each other's selections through a virtual loss (a visit without a score), the
//     standard way to keep B simultaneous descents from all choosing the same path.
// No battle arithmetic happens on the host: every state transition and evaluation above is a kernel launch.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <array>
#include <chrono>
#include <cmath>
#include <vector>

#include "../../include/oakgpu.h"
#include "oakgpu_internal.h"
#include "bandit.hpp"
#include "nash.hpp"

namespace {

uint64_t splitmix64(uint64_t &x) {
  uint64_t z = (x += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
double uniform01(uint64_t &rng) { return (double)(splitmix64(rng) >> 11) * (1.0 / 9007199254740992.0); }
using namespace oak_search;

// The tree.  The reference's Node owns a std::map<(i, j, Obs), Node> (mcts.h:95-105); nearly every iteration adds a
// node (81 joint actions x the observation fan-out), so here nodes are indices into flat arrays and ALL edges live in
// one open-addressing hash table keyed (parent, i, j, 16-byte observation): no per-node allocation, one probe per edge.
struct Stats { Bandit p1, p2; bool is_init() const { return p1.is_init(); } };
struct Edge { uint64_t hash; uint32_t parent, child; uint8_t key[18]; uint8_t used; };
struct Tree {
  std::vector<Stats> nodes;
  std::vector<Edge> table;
  size_t count = 0;
  Tree() : table(1u << 16) { memset(table.data(), 0, table.size() * sizeof(Edge)); }
  static uint64_t hash_of(uint32_t parent, const uint8_t *key) {
    uint64_t a, b;
    uint16_t c;
    memcpy(&a, key, 8); memcpy(&b, key + 8, 8); memcpy(&c, key + 16, 2);
    uint64_t h = (a ^ (uint64_t)parent * 0x9E3779B97F4A7C15ull) * 0xBF58476D1CE4E5B9ull;
    h = (h ^ (h >> 29) ^ b) * 0x94D049BB133111EBull;
    h = (h ^ (h >> 32) ^ c) * 0x9E3779B97F4A7C15ull;
    return h ^ (h >> 31);
  }
  uint32_t new_node() { nodes.emplace_back(); return (uint32_t)nodes.size() - 1; }
  // start a new search in the memory of the previous one (a search adds a node per iteration: tens of MB whose first touch
  // -- page faults, vector regrowth, table rehashes -- cost a third of the host time of a 2^18-iteration search)
  void reset(size_t expected_nodes) {
    nodes.clear();
    if (nodes.capacity() < expected_nodes + 16) nodes.reserve(expected_nodes + 16);
    size_t want = 1u << 16;
    while (want * 6 < (expected_nodes + 16) * 10) want *= 2;
    if (table.size() < want) table.resize(want);
    memset(table.data(), 0, table.size() * sizeof(Edge));
    count = 0;
  }
  void grow() {
    std::vector<Edge> old;
    old.swap(table);
    table.resize(old.size() * 2);
    memset(table.data(), 0, table.size() * sizeof(Edge));
    const size_t mask = table.size() - 1;
    for (const Edge &e : old)
      if (e.used) { size_t i = e.hash & mask; while (table[i].used) i = (i + 1) & mask; table[i] = e; }
  }
  // the host loops over a batch's lanes are bound by cache misses on this table and on `nodes` (tens of MB per search):
  // they ask for a later lane's lines while working on the current one
  void prefetch_edge(uint32_t parent, const uint8_t *key) const { __builtin_prefetch(&table[hash_of(parent, key) & (table.size() - 1)]); }
  void prefetch_node(uint32_t node) const { if (node < nodes.size()) { __builtin_prefetch(&nodes[node]); __builtin_prefetch((const char *)&nodes[node] + 64); } }
  // child of `parent` along (i, j, obs) = key; created (uninitialised) when absent -- heap.children[{i, j, obs}] (mcts.h:359-361)
  uint32_t child(uint32_t parent, const uint8_t *key) {
    if ((count + 1) * 10 > table.size() * 6) grow();
    const uint64_t h = hash_of(parent, key);
    const size_t mask = table.size() - 1;
    size_t i = h & mask;
    for (;; i = (i + 1) & mask) {
      Edge &e = table[i];
      if (!e.used) {
        e.used = 1; e.hash = h; e.parent = parent; memcpy(e.key, key, 18);
        e.child = new_node();
        ++count;
        return e.child;
      }
      if (e.hash == h && e.parent == parent && memcmp(e.key, key, 18) == 0) return e.child;
    }
  }
};

// Equilibrium of an integer zero-sum matrix game (row player maximises): the exact solver of nash.hpp, as the reference
// solves its root matrices exactly with lrsnash (mcts.h:532-543, 643-649).
void solve_zero_sum(const int32_t *A, int m, int n, double *x, double *y) {
  double v;
  if (!oak_nash::solve(A, m, n, x, y, &v)) {
    for (int i = 0; i < m; ++i) x[i] = 1.0 / m;
    for (int j = 0; j < n; ++j) y[j] = 1.0 / n;
  }
}

struct Buffers { // device arrays + pinned host mirrors of what crosses PCIe every level
  std::vector<void *> dev, pinned;
  ~Buffers() {
    for (void *p : dev) (void)hipFree(p);
    for (void *p : pinned) (void)hipHostFree(p);
  }
  template <class T> int d(T **out, size_t count) {
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, count * sizeof(T));
    if (e != hipSuccess) return oakgpu_fail_hip((int)e, "hipMalloc(search buffers)");
    dev.push_back(p);
    *out = (T *)p;
    return 0;
  }
  template <class T> int h(T **out, size_t count) {
    void *p = nullptr;
    hipError_t e = hipHostMalloc(&p, count * sizeof(T), hipHostMallocDefault);
    if (e != hipSuccess) return oakgpu_fail_hip((int)e, "hipHostMalloc(search buffers)");
    pinned.push_back(p);
    *out = (T *)p;
    return 0;
  }
};

#define HIPRC(x) do { hipError_t _e = (x); if (_e != hipSuccess) return oakgpu_fail_hip((int)_e, #x); } while (0)
#define RC(x) do { int _r = (x); if (_r) return _r; } while (0)

} // namespace

// Diagnostic: one player's bandit replayed over a supplied outcome sequence (select -> visit -> update per step), so that
// the bandit arithmetic can be pinned against traces of the reference's own headers without a GPU.
extern "C" int oakgpu_bandit_replay(int kind, float c, float alpha, uint32_t k, const float *logits, uint32_t steps,
                                    const double *uniforms, const float *values, uint8_t *index_out, float *prob_out,
                                    float *stats_out, uint32_t *visits_out) {
  if (kind < 0 || kind > 4 || k < 1 || k > 9 || !values || !index_out) return oakgpu_fail_msg("oakgpu_bandit_replay: bad argument");
  if ((kind == B_PUCB || kind == B_PEXP3) && !logits) return oakgpu_fail_msg("oakgpu_bandit_replay: PUCB / PExp3 need logits");
  if (kind >= B_EXP3 && k > 1 && !uniforms) return oakgpu_fail_msg("oakgpu_bandit_replay: Exp3 / PExp3 need the uniform draws");
  const BanditParams P{kind, c, alpha};
  Bandit b;
  b.init((uint8_t)k, kind);
  if (logits) b.set_logits(P, logits);
  uint32_t u = 0;
  for (uint32_t t = 0; t < steps; ++t) {
    float prob;
    const uint8_t i = b.select(P, [&] { return uniforms[u++]; }, prob);
    b.visit(P, i);
    b.update(P, i, values[t], prob);
    index_out[t] = i;
    if (prob_out) prob_out[t] = prob;
  }
  if (stats_out) for (int i = 0; i < 9; ++i) { stats_out[i] = b.scores[i]; stats_out[9 + i] = b.priors[i]; }
  if (visits_out) for (int i = 0; i < 9; ++i) visits_out[i] = b.visits[i];
  return 0;
}

// pyoak solve_matrix / LRSNash::solve_fast (pyoak.cc:394-426, mcts.h:643-649): exact equilibrium of an integer matrix game
extern "C" int oakgpu_solve_matrix(const int32_t *payoffs, int m, int n, int discretize_factor, double *p1, double *p2, double *value) {
  if (!payoffs || !p1 || !p2 || !value) return oakgpu_fail_msg("oakgpu_solve_matrix: null argument");
  if (m < 1 || n < 1 || m > 9 || n > 9) return oakgpu_fail_msg("oakgpu_solve_matrix: payoff matrix must be between 1x1 and 9x9");
  if (discretize_factor < 1) return oakgpu_fail_msg("oakgpu_solve_matrix: discretize_factor must be positive");
  double v = 0;
  if (!oak_nash::solve(payoffs, m, n, p1, p2, &v)) return oakgpu_fail_msg("oakgpu_solve_matrix: payoffs out of range (|payoff| <= 2^20)");
  *value = v / (double)discretize_factor;
  return 0;
}

extern "C" int oakgpu_search(oakgpu_ctx *ctx, oakgpu_net *net, const uint8_t *battle, const uint8_t *durations, uint8_t result,
                             const oakgpu_search_params *prm, oakgpu_search_output *out) {
  if (!ctx || !battle || !durations || !prm || !out) return oakgpu_fail_msg("oakgpu_search: null argument");
  const bool pucb = prm->bandit == B_PUCB || prm->bandit == B_PEXP3; // the bandits that take priors from the policy heads
  const bool use_net = prm->eval == 1, use_pe = prm->eval == 2;
  const BanditParams BP{prm->bandit, prm->ucb_c, prm->exp3_alpha > 0 ? prm->exp3_alpha : 0.05f};
  if (prm->bandit < 0 || prm->bandit > 4 || prm->eval < 0 || prm->eval > 2) return oakgpu_fail_msg("oakgpu_search: unknown bandit / eval");
  if ((use_net || pucb) && !net) return oakgpu_fail_msg("oakgpu_search: network evaluation / PUCB / PExp3 priors need a network");
  if (pucb && !use_net) return oakgpu_fail_msg("oakgpu_search: PUCB / PExp3 take their priors from the network evaluator (eval = 1)");
  if (prm->batch == 0 || prm->batch > (1u << 20)) return oakgpu_fail_msg("oakgpu_search: batch must be in 1..2^20");
  if (prm->iterations == 0 && prm->duration_us == 0) return oakgpu_fail_msg("oakgpu_search: give an iteration or a time budget");
  auto rolls_ok = [](uint32_t r) { return r == 1 || r == 2 || r == 3 || r == 20 || r == 39; };
  if (!rolls_ok(prm->root_rolls) || !rolls_ok(prm->other_rolls)) return oakgpu_fail_msg("oakgpu_search: rolls must be 1, 2, 3, 20 or 39");
  HIPRC(hipSetDevice(oakgpu_ctx_device(ctx)));
  const uint32_t B = prm->batch;
  const uint32_t max_depth = prm->max_depth ? prm->max_depth : 100;
  memset(out, 0, sizeof *out);
  const auto t_start = std::chrono::high_resolution_clock::now();

  // root choices (mcts.h:160-166) through the batched choices kernel, batch of one
  uint8_t root_c1[9], root_c2[9], m = 0, n = 0;
  RC(oakgpu_choices(ctx, battle, &result, 0, root_c1, &m, 1));
  RC(oakgpu_choices(ctx, battle, &result, 1, root_c2, &n, 1));
  out->m = m;
  out->n = n;
  memcpy(out->p1_choices, root_c1, 9);
  memcpy(out->p2_choices, root_c2, 9);
  if ((result & 15) != 0 || m == 0 || n == 0) return oakgpu_fail_msg("oakgpu_search: the root position is terminal");

  static thread_local Tree tree_storage; // (kept between searches of a thread: see Tree::reset)
  Tree &tree = tree_storage;
  tree.reset(prm->duration_us != 0 ? (size_t)1 << 20 : (size_t)std::min<uint64_t>(prm->iterations, (uint64_t)1 << 24));
  const uint32_t root = tree.new_node();
  tree.nodes[root].p1.init(m, BP.kind);
  tree.nodes[root].p2.init(n, BP.kind);
  if (pucb) { // root priors from the policy heads (mcts.h:196-209)
    float v, l1[9], l2[9];
    RC(oakgpu_leaf_eval_policy(ctx, net, battle, durations, 1, root_c1, &m, root_c2, &n, &v, l1, l2));
    tree.nodes[root].p1.set_logits(BP, l1);
    tree.nodes[root].p2.set_logits(BP, l2);
    out->initial_value = v;
  }

  float pe_root = 0.0f; // PokeEngine::Eval::get_root_score (mcts.h:172-174)
  if (use_pe) RC(oakgpu_poke_engine_eval(ctx, battle, 1, 0.0f, nullptr, &pe_root));

  // Two batches are kept in flight ("slots", each with its own context = HIP stream and buffers): while the GPU
  // evaluates the leaves of one batch (rollouts: milliseconds), the host walks the tree for the other.  The schedule
  // is fixed (A descends, B descends, A finishes, A descends, B finishes, ...), so a search is reproducible.
  struct Step { uint32_t node; uint8_t i, j; float prob1, prob2; };
  constexpr uint32_t NO_NODE = 0xFFFFFFFFu;
  struct Slot {
    oakgpu_ctx *ctx = nullptr;
    bool own_ctx = false;
    hipStream_t stream{};
    Buffers buf;
    uint8_t *d_root_b, *d_root_d, *d_root_r, *d_b, *d_d, *d_r, *d_prng, *d_c1, *d_c2, *d_act, *d_ch1, *d_cnt1, *d_ch2, *d_cnt2, *d_rout;
    uint32_t *d_steps;
    float *d_values, *d_l1 = nullptr, *d_l2 = nullptr, *d_emb = nullptr;
    uint8_t *h_c1, *h_c2, *h_r, *h_act, *h_ch1, *h_cnt1, *h_ch2, *h_cnt2, *h_stage;
    float *h_values, *h_l1 = nullptr, *h_l2 = nullptr;
    std::vector<std::vector<Step>> path;
    std::vector<uint32_t> cur, leaf;
    std::vector<uint8_t> active;
    std::vector<uint8_t> forced;
    double nash1[9], nash2[9];
    uint32_t nb = 0;
    bool busy = false;
    ~Slot() { if (own_ctx && ctx) oakgpu_destroy(ctx); }
  };
  // one batch at a time only for batch = 1, which is the reference's strictly sequential iteration order (the evaluator's
  // workspaces belong to the context, so two slots -- two contexts -- can evaluate the same network concurrently)
  const bool timed = prm->duration_us != 0; // time budget (search.cc:300-306): batches are started until it has elapsed
  const int n_slots = (B > 1 && (timed || prm->iterations > B)) ? 2 : 1;
  Slot slots[2];
  for (int si = 0; si < n_slots; ++si) {
    Slot &S = slots[si];
    if (si == 0) S.ctx = ctx;
    else { RC(oakgpu_create(&S.ctx, oakgpu_ctx_device(ctx))); S.own_ctx = true; }
    S.stream = (hipStream_t)oakgpu_ctx_stream(S.ctx);
    Buffers &buf = S.buf;
    RC(buf.d(&S.d_root_b, (size_t)B * 384)); RC(buf.d(&S.d_root_d, (size_t)B * 8)); RC(buf.d(&S.d_root_r, (size_t)B));
    RC(buf.d(&S.d_b, (size_t)B * 384)); RC(buf.d(&S.d_d, (size_t)B * 8)); RC(buf.d(&S.d_r, (size_t)B)); RC(buf.d(&S.d_prng, (size_t)B * 8));
    RC(buf.d(&S.d_c1, (size_t)B)); RC(buf.d(&S.d_c2, (size_t)B)); RC(buf.d(&S.d_act, (size_t)B * 16));
    RC(buf.d(&S.d_ch1, (size_t)B * 9)); RC(buf.d(&S.d_cnt1, (size_t)B)); RC(buf.d(&S.d_ch2, (size_t)B * 9)); RC(buf.d(&S.d_cnt2, (size_t)B));
    RC(buf.d(&S.d_rout, (size_t)B)); RC(buf.d(&S.d_steps, (size_t)B)); RC(buf.d(&S.d_values, (size_t)B));
    if (pucb) { RC(buf.d(&S.d_l1, (size_t)B * 9)); RC(buf.d(&S.d_l2, (size_t)B * 9)); }
    if (use_net && n_slots > 1) { // per-slot embedding buffer (kept from round 1; the context's own workspace would do as well)
      int emb_dim = 0;
      RC(oakgpu_net_shape(net, &emb_dim, nullptr, nullptr, nullptr));
      RC(buf.d(&S.d_emb, (size_t)B * emb_dim));
    }
    RC(buf.h(&S.h_c1, (size_t)B)); RC(buf.h(&S.h_c2, (size_t)B)); RC(buf.h(&S.h_r, (size_t)B)); RC(buf.h(&S.h_act, (size_t)B * 16));
    RC(buf.h(&S.h_ch1, (size_t)B * 9)); RC(buf.h(&S.h_cnt1, (size_t)B)); RC(buf.h(&S.h_ch2, (size_t)B * 9)); RC(buf.h(&S.h_cnt2, (size_t)B));
    RC(buf.h(&S.h_values, (size_t)B)); RC(buf.h(&S.h_stage, (size_t)B * 384));
    if (pucb) { RC(buf.h(&S.h_l1, (size_t)B * 9)); RC(buf.h(&S.h_l2, (size_t)B * 9)); }
    // root template: B copies of the input; one fast_prng stream per lane (util/random.h:67-133), never all-zero
    for (uint32_t l = 0; l < B; ++l) memcpy(S.h_stage + (size_t)l * 384, battle, 384);
    HIPRC(hipMemcpyAsync(S.d_root_b, S.h_stage, (size_t)B * 384, hipMemcpyHostToDevice, S.stream));
    HIPRC(hipStreamSynchronize(S.stream));
    for (uint32_t l = 0; l < B; ++l) memcpy(S.h_stage + (size_t)l * 8, durations, 8);
    HIPRC(hipMemcpyAsync(S.d_root_d, S.h_stage, (size_t)B * 8, hipMemcpyHostToDevice, S.stream));
    HIPRC(hipStreamSynchronize(S.stream));
    uint64_t sm = prm->seed + 0x632BE59BD9B4E019ull * (uint64_t)si;
    for (uint32_t l = 0; l < B; ++l) { uint64_t x = splitmix64(sm) | 1; memcpy(S.h_stage + (size_t)l * 8, &x, 8); }
    HIPRC(hipMemcpyAsync(S.d_prng, S.h_stage, (size_t)B * 8, hipMemcpyHostToDevice, S.stream));
    HIPRC(hipMemsetAsync(S.d_root_r, result, B, S.stream));
    HIPRC(hipStreamSynchronize(S.stream));
    S.path.resize(B); S.cur.resize(B); S.leaf.resize(B); S.active.resize(B);
  }

  double total_value = 0;
  const bool timing = getenv("OAKGPU_SEARCH_TIMING") != nullptr;
  double t_sel = 0, t_gpu = 0, t_proc = 0, t_eval = 0, t_back = 0;
  auto now = [] { return std::chrono::high_resolution_clock::now(); };
  auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
  uint64_t done = 0, started = 0, total_depth = 0;
  uint64_t mucb_rng = prm->seed ^ 0xA0761D6478BD642Full, bandit_rng = prm->seed ^ 0xE7037ED1A0B428DBull;

  // one batch: root prep, level-synchronous descent (host selection <-> k_tree_step), then the leaf evaluation is
  // LAUNCHED (not awaited)
  auto descend = [&](Slot &S) -> int {
    const uint32_t nb = timed ? B : (uint32_t)std::min<uint64_t>(B, prm->iterations - started);
    S.nb = nb;
    started += nb;
    S.busy = true;
    // root prep on the device (mcts.h:254-259): rollout kernel with max_steps = 0
    RC(oakgpu_rollout_dev(S.ctx, S.d_root_b, S.d_root_d, S.d_root_r, S.d_prng, nb, 0, 1, S.d_rout, S.d_steps, S.d_values, S.d_b, S.d_d));
    HIPRC(hipMemcpyAsync(S.d_r, S.d_root_r, nb, hipMemcpyDeviceToDevice, S.stream));
    for (uint32_t l = 0; l < nb; ++l) { S.path[l].clear(); S.cur[l] = root; S.leaf[l] = NO_NODE; S.active[l] = 1; }
    uint32_t n_active = nb;
    // MatrixUCB (mcts.h:263-302): the root's joint actions of this batch come from the UCB matrices, not the bandits
    const bool mucb = prm->matrix_ucb && done >= prm->mucb_delay;
    if (mucb) {
      S.forced.clear();
      uint64_t planned[81];
      for (int q = 0; q < 81; ++q) planned[q] = out->visit_matrix[q];
      for (int i = 0; i < m && S.forced.size() < nb; ++i) // cells under the minimum visit count go first (mcts.h:510-512)
        for (int j = 0; j < n && S.forced.size() < nb; ++j)
          while (planned[i * 9 + j] < prm->mucb_minimum && S.forced.size() < nb) { S.forced.push_back((uint8_t)(i * 9 + j)); ++planned[i * 9 + j]; }
      if (S.forced.size() < nb) {
        int32_t up[81], dn[81];
        const double log_T = std::log((double)(done ? done : 1)), w = std::log(2.0 * m * n);
        for (int i = 0; i < m; ++i)
          for (int j = 0; j < n; ++j) {
            const uint64_t v = out->visit_matrix[i * 9 + j];
            const double mean = v ? out->value_matrix[i * 9 + j] / (double)v : 0.0;
            const double e = prm->mucb_c * std::sqrt(2.0 * (2.0 * log_T + w) / (double)(v + 1));
            up[i * n + j] = (int32_t)(((v ? mean : 0.0) + e) * 256.0); // integer matrices x 256, truncated like the reference's (mcts.h:527-528)
            dn[i * n + j] = (int32_t)(((v ? mean : 1.0) - e) * 256.0);
          }
        double dummy[9];
        solve_zero_sum(up, m, n, S.nash1, dummy);
        solve_zero_sum(dn, m, n, dummy, S.nash2);
      }
    }
    for (uint32_t depth = 0; n_active > 0; ++depth) {
      const auto ta = now();
      for (uint32_t l = 0; l < nb; ++l) { // bandit selection, sequential: each lane sees the virtual losses before it
        if (l + 12 < nb && S.active[l + 12]) tree.prefetch_node(S.cur[l + 12]);
        if (!S.active[l]) { S.h_c1[l] = 0xFF; S.h_c2[l] = 0xFF; continue; }
        uint8_t i, j;
        if (mucb && depth == 0) { // sampled / forced root action; the root bandits are neither consulted nor updated
          if (l < S.forced.size()) { i = S.forced[l] / 9; j = S.forced[l] % 9; }
          else {
            auto sample = [&](const double *p, int k) {
              double u = uniform01(mucb_rng);
              for (int q = 0; q < k; ++q) { u -= p[q]; if (u <= 0) return (uint8_t)q; }
              return (uint8_t)(k - 1);
            };
            i = sample(S.nash1, m);
            j = sample(S.nash2, n);
          }
          S.path[l].push_back({NO_NODE, i, j, 1.0f, 1.0f});
        } else {
          Stats &nd = tree.nodes[S.cur[l]];
          float pr1, pr2;
          auto draw = [&] { return uniform01(bandit_rng); }; // device.uniform() of sample_pdf (util/random.h:40-49)
          i = nd.p1.select(BP, draw, pr1);
          j = nd.p2.select(BP, draw, pr2);
          nd.p1.visit(BP, i);
          nd.p2.visit(BP, j);
          S.path[l].push_back({S.cur[l], i, j, pr1, pr2});
        }
        S.h_c1[l] = depth == 0 ? root_c1[i] : S.h_ch1[(size_t)l * 9 + i];
        S.h_c2[l] = depth == 0 ? root_c2[j] : S.h_ch2[(size_t)l * 9 + j];
      }
      const auto tb = now();
      HIPRC(hipMemcpyAsync(S.d_c1, S.h_c1, nb, hipMemcpyHostToDevice, S.stream));
      HIPRC(hipMemcpyAsync(S.d_c2, S.h_c2, nb, hipMemcpyHostToDevice, S.stream));
      RC(oakgpu_tree_step_dev(S.ctx, S.d_b, S.d_d, S.d_r, S.d_c1, S.d_c2, nb, depth == 0 ? prm->root_rolls : prm->other_rolls, S.d_act,
                              S.d_ch1, S.d_cnt1, S.d_ch2, S.d_cnt2));
      HIPRC(hipMemcpyAsync(S.h_r, S.d_r, nb, hipMemcpyDeviceToHost, S.stream));
      HIPRC(hipMemcpyAsync(S.h_act, S.d_act, (size_t)nb * 16, hipMemcpyDeviceToHost, S.stream));
      HIPRC(hipMemcpyAsync(S.h_ch1, S.d_ch1, (size_t)nb * 9, hipMemcpyDeviceToHost, S.stream));
      HIPRC(hipMemcpyAsync(S.h_cnt1, S.d_cnt1, nb, hipMemcpyDeviceToHost, S.stream));
      HIPRC(hipMemcpyAsync(S.h_ch2, S.d_ch2, (size_t)nb * 9, hipMemcpyDeviceToHost, S.stream));
      HIPRC(hipMemcpyAsync(S.h_cnt2, S.d_cnt2, nb, hipMemcpyDeviceToHost, S.stream));
      HIPRC(hipStreamSynchronize(S.stream));
      const auto tc = now();
      for (uint32_t l = 0; l < nb; ++l) {
        if (l + 16 < nb && S.active[l + 16] && (S.h_r[l + 16] & 15) == 0) {
          uint8_t pk[18];
          pk[0] = S.path[l + 16].back().i;
          pk[1] = S.path[l + 16].back().j;
          memcpy(pk + 2, S.h_act + (size_t)(l + 16) * 16, 16);
          tree.prefetch_edge(S.cur[l + 16], pk);
        }
        if (!S.active[l]) continue;
        if ((S.h_r[l] & 15) != 0) { // terminal edge: the value comes from the result byte (mcts.h:427-441)
          S.active[l] = 0; --n_active; total_depth += depth + 1;
          continue;
        }
        uint8_t key[18];
        key[0] = S.path[l].back().i;
        key[1] = S.path[l].back().j;
        memcpy(key + 2, S.h_act + (size_t)l * 16, 16);
        const uint32_t child = tree.child(S.cur[l], key);
        if (tree.nodes[child].is_init() && depth + 1 < max_depth) { S.cur[l] = child; continue; }
        S.leaf[l] = child; // first visit (or depth cap): evaluate here (mcts.h:391-426)
        S.active[l] = 0; --n_active; total_depth += depth + 1;
      }
      const auto td = now();
      t_sel += us(ta, tb); t_gpu += us(tb, tc); t_proc += us(tc, td);
    }
    // leaf evaluation, in place on the device; results are collected by finish()
    if (use_pe) {
      RC(oakgpu_poke_engine_eval_dev(S.ctx, S.d_b, nb, pe_root, S.d_values, nullptr));
    } else if (!use_net) {
      RC(oakgpu_rollout_dev(S.ctx, S.d_b, S.d_d, S.d_r, S.d_prng, nb, 1000, 0, S.d_rout, S.d_steps, S.d_values, nullptr, nullptr));
    } else if (pucb) {
      RC(oakgpu_leaf_eval_policy_dev(S.ctx, net, S.d_b, S.d_d, nb, S.d_ch1, S.d_cnt1, S.d_ch2, S.d_cnt2, S.d_values, S.d_l1, S.d_l2));
      HIPRC(hipMemcpyAsync(S.h_l1, S.d_l1, (size_t)nb * 9 * 4, hipMemcpyDeviceToHost, S.stream));
      HIPRC(hipMemcpyAsync(S.h_l2, S.d_l2, (size_t)nb * 9 * 4, hipMemcpyDeviceToHost, S.stream));
    } else {
      RC(oakgpu_leaf_eval_dev(S.ctx, net, S.d_b, S.d_d, nb, S.d_values, S.d_emb));
    }
    HIPRC(hipMemcpyAsync(S.h_values, S.d_values, (size_t)nb * 4, hipMemcpyDeviceToHost, S.stream));
    return 0;
  };
  // wait for the batch's leaf values, initialise its new leaves and back the values up its paths
  auto finish = [&](Slot &S) -> int {
    const auto te = now();
    HIPRC(hipStreamSynchronize(S.stream));
    const auto tf = now();
    const uint32_t nb = S.nb;
    for (uint32_t l = 0; l < nb; ++l) {
      if (l + 8 < nb) { for (const Step &st : S.path[l + 8]) if (st.node != NO_NODE) tree.prefetch_node(st.node); if (S.leaf[l + 8] != NO_NODE) tree.prefetch_node(S.leaf[l + 8]); }
      float v1;
      const uint32_t t = S.h_r[l] & 15;
      if (t != 0) v1 = t == 1 ? 1.0f : t == 2 ? 0.0f : 0.5f;
      else v1 = S.h_values[l];
      if (S.leaf[l] != NO_NODE && !tree.nodes[S.leaf[l]].is_init() && S.h_cnt1[l] && S.h_cnt2[l]) { // stats.init(m, n) (+ priors), first evaluation
        Stats &lf = tree.nodes[S.leaf[l]];
        lf.p1.init(S.h_cnt1[l], BP.kind);
        lf.p2.init(S.h_cnt2[l], BP.kind);
        if (pucb) { lf.p1.set_logits(BP, S.h_l1 + (size_t)l * 9); lf.p2.set_logits(BP, S.h_l2 + (size_t)l * 9); }
      }
      const float v2 = 1.0f - v1;
      for (const Step &st : S.path[l]) { // Bandit::update, the visit was already counted as the virtual loss
        if (st.node == NO_NODE) continue;  // MatrixUCB root step: only the root matrices below are updated
        tree.nodes[st.node].p1.update(BP, st.i, v1, st.prob1);
        tree.nodes[st.node].p2.update(BP, st.j, v2, st.prob2);
      }
      const Step &s0 = S.path[l].front();
      ++out->visit_matrix[s0.i * 9 + s0.j];
      out->value_matrix[s0.i * 9 + s0.j] += v1;
      total_value += v1;
    }
    t_eval += us(te, tf); t_back += us(tf, now());
    done += nb;
    S.busy = false;
    return 0;
  };
  auto more = [&] {
    if (!timed) return started < prm->iterations;
    return std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t_start).count() < (double)prm->duration_us;
  };
  for (int turn = 0; more() || slots[0].busy || slots[1].busy; turn = (turn + 1) % n_slots) {
    Slot &S = slots[turn];
    if (S.busy) RC(finish(S));
    if (more()) RC(descend(S));
  }
  if (timing) fprintf(stderr, "oakgpu_search timing (ms): select %.1f  gpu-step %.1f  process %.1f  eval %.1f  backprop %.1f\n", t_sel / 1e3, t_gpu / 1e3, t_proc / 1e3, t_eval / 1e3, t_back / 1e3);
  out->iterations = done;
  out->empirical_value = done ? total_value / (double)done : 0.0;
  for (int i = 0; i < m; ++i)
    for (int j = 0; j < n; ++j) {
      out->p1_empirical[i] += (double)out->visit_matrix[i * 9 + j] / (double)done;
      out->p2_empirical[j] += (double)out->visit_matrix[i * 9 + j] / (double)done;
    }
  { // MCTS::Search::process_output (mcts.h:620-659): empirical root matrix x 256 as integers, solved exactly
    int32_t M[81];
    for (int i = 0; i < m; ++i)
      for (int j = 0; j < n; ++j) {
        uint64_t v = out->visit_matrix[i * 9 + j];
        v += !v;
        M[i * n + j] = (int32_t)(out->value_matrix[i * 9 + j] / (double)v * 256.0);
      }
    double nv = 0;
    if (oak_nash::solve(M, m, n, out->p1_nash, out->p2_nash, &nv)) out->nash_value = nv / 256.0;
  }
  out->nodes = tree.nodes.size();
  out->total_depth = total_depth;
  out->duration_us = std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t_start).count();
  return 0;
}


// ---- RuntimeSearch::run (cpp/include/util/search.h:17-66, cpp/src/search.cc:150-313): the Agent's strings select budget,
// bandit, evaluator and MatrixUCB at run time.  Same mini-languages and the same error texts as the reference, where it
// throws std::runtime_error; what this build does not have (transposition-table heaps, the int8 "discrete" network) is
// refused by name instead of being silently replaced.
#include <map>
#include <mutex>
#include <string>
namespace {
std::vector<std::string> split(const std::string &s, char sep) {
  std::vector<std::string> out;
  size_t pos = 0;
  for (;;) {
    const size_t end = s.find(sep, pos);
    out.push_back(s.substr(pos, end == std::string::npos ? std::string::npos : end - pos));
    if (end == std::string::npos) break;
    pos = end + 1;
  }
  return out;
}
bool to_float(const std::string &s, float &v) { char *e = nullptr; v = strtof(s.c_str(), &e); return e && e != s.c_str() && *e == 0; }
bool to_u64(const std::string &s, uint64_t &v) { char *e = nullptr; v = strtoull(s.c_str(), &e, 10); return !s.empty() && e && *e == 0 && s[0] != '-'; }
std::mutex g_net_mu;
std::map<std::pair<int, std::string>, oakgpu_net *> g_nets; // Agent::network_ptr (search.cc:62-148), shared per (device, path)
} // namespace

extern "C" int oakgpu_search_agent(oakgpu_ctx *ctx, const uint8_t *battle, const uint8_t *durations, uint8_t result, const oakgpu_agent *agent,
                                   uint32_t batch, uint64_t seed, oakgpu_search_output *out) {
  if (!ctx || !battle || !durations || !agent || !out) return oakgpu_fail_msg("oakgpu_search_agent: null argument");
  oakgpu_search_params P{};
  P.root_rolls = 3; P.other_rolls = 1; // default_search (mcts.h:131)
  P.seed = seed;
  // budget: "4096" | "100ms" | "8s" (search.cc:292-310)
  const std::string budget = agent->budget ? agent->budget : "";
  const size_t pos = budget.find_first_not_of("0123456789");
  uint64_t number = 0;
  if (budget.empty() || pos == 0 || !to_u64(budget.substr(0, pos), number)) return oakgpu_fail_msg(("Invalid search duration specification: " + budget).c_str());
  const std::string unit = pos == std::string::npos ? "" : budget.substr(pos);
  if (unit.empty()) P.iterations = number;
  else if (unit == "ms" || unit == "millisec" || unit == "milliseconds") P.duration_us = number * 1000;
  else if (unit == "s" || unit == "sec" || unit == "seconds") P.duration_us = number * 1000000;
  else return oakgpu_fail_msg(("Invalid search duration specification: " + budget).c_str());
  if (P.iterations == 0 && P.duration_us == 0) return oakgpu_fail_msg(("Invalid search duration specification: " + budget).c_str());
  // evaluator: "" / "mc" / "montecarlo" / "monte-carlo" | "fp" | <network path> (util/search.h:56-61)
  const std::string eval = agent->eval ? agent->eval : "";
  const bool mc = eval.empty() || eval == "mc" || eval == "montecarlo" || eval == "monte-carlo", fp = eval == "fp";
  P.eval = mc ? 0 : fp ? 2 : 1;
  // bandit: "ucb-1.0" | "ucb1-2.0" | "pucb-1.5" | "exp3-<gamma>[-<alpha>]" | "pexp3-..." (search.cc:237-290)
  const std::string bandit = agent->bandit ? agent->bandit : "";
  const std::vector<std::string> bs = split(bandit, '-');
  if (bs.size() < 2 || !to_float(bs[1], P.ucb_c)) return oakgpu_fail_msg(("Could not parse bandit string: " + bandit).c_str());
  const std::string &name = bs[0];
  if (name == "ucb") P.bandit = 0;
  else if (name == "pucb") P.bandit = 1;
  else if (name == "ucb1") P.bandit = 2;
  else if (name == "exp3") P.bandit = 3;
  else if (name == "pexp3") P.bandit = 4;
  else return oakgpu_fail_msg(("Could not parse bandit string: " + name).c_str());
  if ((P.bandit == 1 || P.bandit == 4) && (mc || fp)) return oakgpu_fail_msg("Contextual bandit specified with eval that does not produce policy priors.");
  P.exp3_alpha = 0.05f;
  if (P.bandit >= 3 && bs.size() >= 3 && !to_float(bs[2], P.exp3_alpha)) return oakgpu_fail_msg(("Could not parse bandit string: " + bandit).c_str());
  // matrix_ucb: "" | "delay-interval-minimum-c" (search.cc:216-235); the interval is the batch here
  const std::string mu = agent->matrix_ucb ? agent->matrix_ucb : "";
  if (!mu.empty()) {
    const std::vector<std::string> ms = split(mu, '-');
    uint64_t delay = 0, interval = 0, minimum = 0;
    if (ms.size() != 4 || !to_u64(ms[0], delay) || !to_u64(ms[1], interval) || !to_u64(ms[2], minimum) || !to_float(ms[3], P.mucb_c))
      return oakgpu_fail_msg(("Could not parse MatrixUCB name: " + mu).c_str());
    P.matrix_ucb = 1;
    P.mucb_delay = (uint32_t)delay;
    P.mucb_minimum = (uint32_t)minimum;
  }
  if (agent->table) return oakgpu_fail_msg("RuntimeSearch: transposition-table heaps (search/hash.h) are not built in this library");
  if (agent->discrete) return oakgpu_fail_msg("RuntimeSearch: the int8 (discrete) network (nn/battle/quantized) is not built in this library; use the fp32 network");
  // descents in flight: the caller's choice, or a size that keeps the GPU busy without starving the tree of feedback
  if (batch) P.batch = batch;
  else if (P.duration_us) P.batch = 4096;
  else P.batch = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(16384, P.iterations / 16));
  oakgpu_net *net = nullptr;
  if (P.eval == 1) { // Agent::initialize_network (search.cc:62-148): read once, keep
    std::lock_guard<std::mutex> lock(g_net_mu);
    const auto key = std::make_pair(oakgpu_ctx_device(ctx), eval);
    auto it = g_nets.find(key);
    if (it == g_nets.end()) {
      oakgpu_net *n = nullptr;
      if (int rc = oakgpu_net_load(ctx, eval.c_str(), &n)) return rc;
      it = g_nets.emplace(key, n).first;
    }
    net = it->second;
  }
  return oakgpu_search(ctx, net, battle, durations, result, &P, out);
}

extern "C" void oakgpu_agent_networks_clear(oakgpu_ctx *ctx) { // drops the networks oakgpu_search_agent loaded on this context's device
  if (!ctx) return;
  std::lock_guard<std::mutex> lock(g_net_mu);
  for (auto it = g_nets.begin(); it != g_nets.end();)
    if (it->first.first == oakgpu_ctx_device(ctx)) { oakgpu_net_free(ctx, it->second); it = g_nets.erase(it); } else ++it;
}
