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
#include <sched.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
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
// counter-based draw: the uniform of (search seed, batch serial, lane, depth, player) -- independent of the order in which
// the worker threads reach the lanes
double uniform_at(uint64_t seed, uint64_t serial, uint32_t lane, uint32_t depth, uint32_t player) {
  uint64_t x = seed ^ (serial * 0xD1342543DE82EF95ull) ^ ((uint64_t)lane << 20) ^ ((uint64_t)depth << 1) ^ player;
  (void)splitmix64(x);
  return uniform01(x);
}
using namespace oak_search;

// The tree.  The reference's Node owns a std::map<(i, j, Obs), Node> (mcts.h:95-105); nearly every iteration adds a
// node (81 joint actions x the observation fan-out), so here nodes are indices into flat arenas and all edges live in
// open-addressing hash tables keyed (parent, i, j, 16-byte observation): no per-node allocation, one probe per edge.
//
// The tree is cut into SHARDS = 16 shards (8 until round 5: a GPU box's job has 16 cores) so that the host work of a batch (bandit
// selection, edge lookups, back-ups) can run on several threads WITHOUT changing any result: a node id is
// `local << 2 SHARD_BITS | creator << SHARD_BITS | owner`; every
// read-modify-write of a node's bandits is done by the thread that serves its OWNER shard, in a fixed service order; an
// edge (parent, i, j, obs) lives in the table picked by SHARD_BITS bits of ITS hash (so the edges of one hot parent -- the root
// first of all -- are looked up and created by all threads, not by one), and a new child (owner = SHARD_BITS other hash bits) is
// appended to the arena [owner][creator = the edge's table] -- only that table's thread appends there.  The number of
// shards is fixed, the number of threads (1, 2, 4, 8 or 16, each serving shards = thread mod threads) is not: any thread
// count walks the same tree bit for bit.
constexpr int SHARD_BITS = 4, SHARDS = 1 << SHARD_BITS, SHARD_MASK = SHARDS - 1;
struct NodeRec {
  Bandit p1, p2;
  bool is_init() const { return p1.is_init(); }
};
struct Edge { uint64_t hash; uint32_t parent, child; uint8_t key[18]; uint16_t gen; };
inline int owner_of(uint32_t id) { return (int)(id & SHARD_MASK); }
inline int creator_of(uint32_t id) { return (int)((id >> SHARD_BITS) & SHARD_MASK); }
inline uint32_t local_of(uint32_t id) { return id >> (2 * SHARD_BITS); }
struct Tree {
  std::vector<NodeRec> arena[SHARDS][SHARDS]; // [owner][creator]
  struct Table { Edge *e = nullptr; size_t cap = 0, count = 0; uint16_t gen = 1; } tab[SHARDS];
  Tree() = default;
  Tree(const Tree &) = delete;
  Tree &operator=(const Tree &) = delete;
  ~Tree() { for (auto &t : tab) free(t.e); }
  static uint64_t hash_of(uint32_t parent, const uint8_t *key) {
    uint64_t a, b;
    uint16_t c;
    memcpy(&a, key, 8); memcpy(&b, key + 8, 8); memcpy(&c, key + 16, 2);
    uint64_t h = (a ^ (uint64_t)parent * 0x9E3779B97F4A7C15ull) * 0xBF58476D1CE4E5B9ull;
    h = (h ^ (h >> 29) ^ b) * 0x94D049BB133111EBull;
    h = (h ^ (h >> 32) ^ c) * 0x9E3779B97F4A7C15ull;
    return h ^ (h >> 31);
  }
  NodeRec &node(uint32_t id) { return arena[owner_of(id)][creator_of(id)][local_of(id)]; }
  size_t size() const {
    size_t n = 0;
    for (auto &o : arena) for (auto &v : o) n += v.size();
    return n;
  }
  uint32_t new_node(int own, int cre) {
    auto &v = arena[own][cre];
    v.emplace_back();
    return (uint32_t)((v.size() - 1) << (2 * SHARD_BITS) | (size_t)cre << SHARD_BITS | (size_t)own);
  }
  // Start a new tree in the memory of the previous one.  Nothing is memset: an edge slot is live only when its generation
  // stamp equals the table's, so a new search costs O(1) however large the last one was (round-2 advice: every search
  // re-zeroed the whole table, 80 MB and more); tables come from calloc (pages are touched when used) and are given
  // back when the last search was more than 8x larger than this one expects to be.
  void reset(size_t expected_nodes) {
    const size_t per = expected_nodes / SHARDS + 64;
    for (auto &o : arena)
      for (auto &v : o) {
        v.clear();
        if (v.capacity() > 8 * (per / SHARDS + 64)) std::vector<NodeRec>().swap(v);
        if (v.capacity() < per / SHARDS + 16) v.reserve(per / SHARDS + 16);
      }
    size_t want = 1u << 12;
    while (want * 6 < per * 10) want *= 2;
    for (auto &t : tab) {
      if (t.cap < want || t.cap > 8 * want) {
        free(t.e);
        t.e = (Edge *)calloc(want, sizeof(Edge));
        t.cap = t.e ? want : 0;
        t.gen = 1;
      } else if (++t.gen == 0) { // the stamp wrapped (65,535 searches): one real clear
        memset(t.e, 0, t.cap * sizeof(Edge));
        t.gen = 1;
      }
      t.count = 0;
    }
  }
  bool ok() const { for (auto &t : tab) if (!t.e) return false; return true; }
  void grow(Table &t) {
    Edge *old = t.e;
    const size_t oc = t.cap;
    t.e = (Edge *)calloc(oc * 2, sizeof(Edge));
    if (!t.e) { t.e = old; return; } // (keeps working, slower, until it is full: 60% -> 100% never happens before reset in practice)
    t.cap = oc * 2;
    const size_t mask = t.cap - 1;
    for (size_t q = 0; q < oc; ++q)
      if (old[q].gen == t.gen) { size_t i = old[q].hash & mask; while (t.e[i].gen == t.gen) i = (i + 1) & mask; t.e[i] = old[q]; }
    free(old);
  }
  // the host loops over a batch's lanes are bound by cache misses on these tables and on the arenas (tens of MB per search):
  // they ask for a later lane's lines while working on the current one
  static int edge_shard(uint64_t h) { return (int)(h >> (64 - SHARD_BITS)); }
  static int child_owner(uint64_t h) { return (int)((h >> (64 - 2 * SHARD_BITS)) & SHARD_MASK); }
  void prefetch_edge(uint64_t h) const { const Table &t = tab[edge_shard(h)]; __builtin_prefetch(&t.e[h & (t.cap - 1)]); }
  void prefetch_node(uint32_t id) const {
    const auto &v = arena[owner_of(id)][creator_of(id)];
    const uint32_t q = local_of(id);
    if (q < v.size()) { __builtin_prefetch(&v[q]); __builtin_prefetch((const char *)&v[q] + 64); __builtin_prefetch((const char *)&v[q] + 128); }
  }
  // child of `parent` along (i, j, obs) = key, h = hash_of(parent, key); created (uninitialised) when absent --
  // heap.children[{i, j, obs}] (mcts.h:359-361).  Called by the thread that serves edge_shard(h).
  uint32_t child(uint32_t parent, const uint8_t *key, uint64_t h) {
    Table &t = tab[edge_shard(h)];
    if ((t.count + 1) * 10 > t.cap * 6) grow(t);
    const size_t mask = t.cap - 1;
    for (size_t i = h & mask;; i = (i + 1) & mask) {
      Edge &e = t.e[i];
      if (e.gen != t.gen) {
        e.gen = t.gen; e.hash = h; e.parent = parent; memcpy(e.key, key, 18);
        e.child = new_node(child_owner(h), edge_shard(h));
        ++t.count;
        return e.child;
      }
      if (e.hash == h && e.parent == parent && memcmp(e.key, key, 18) == 0) return e.child;
    }
  }
  bool find_child(uint32_t parent, const uint8_t *key, uint32_t *out) const { // lookup only (Heap::update, search.cc:38)
    const uint64_t h = hash_of(parent, key);
    const Table &t = tab[edge_shard(h)];
    if (!t.e) return false;
    const size_t mask = t.cap - 1;
    for (size_t i = h & mask;; i = (i + 1) & mask) {
      const Edge &e = t.e[i];
      if (e.gen != t.gen) return false;
      if (e.hash == h && e.parent == parent && memcmp(e.key, key, 18) == 0) { *out = e.child; return true; }
    }
  }
  // Heap::update's `std::swap(node, child->second)` (search.cc:42): the subtree under `new_root` becomes the tree, everything
  // else is dropped.  Ids are reassigned in breadth-first order.  A kept node keeps its OWNER shard, but its CREATOR becomes the
  // table its incoming edge re-hashes to under the new parent id: the resolve phase relies on "arena[owner][t] is touched only
  // by the thread of table t" (a lookup through table t reads node(child) while table t's thread may append to -- and so
  // reallocate -- exactly the arenas [*][t]); with the old creator kept, thread t would read an arena that another thread is
  // appending to (round-3 advice).  Returns the new root id.
  uint32_t keep_subtree(uint32_t new_root) {
    struct E { uint32_t parent, child; uint8_t key[18]; };
    std::vector<E> edges;
    for (auto &t : tab)
      for (size_t q = 0; q < t.cap; ++q)
        if (t.e[q].gen == t.gen) { E x; x.parent = t.e[q].parent; x.child = t.e[q].child; memcpy(x.key, t.e[q].key, 18); edges.push_back(x); }
    std::sort(edges.begin(), edges.end(), [](const E &a, const E &b) { return a.parent != b.parent ? a.parent < b.parent : memcmp(a.key, b.key, 18) < 0; });
    auto first_edge = [&](uint32_t parent) {
      return std::lower_bound(edges.begin(), edges.end(), parent, [](const E &a, uint32_t p) { return a.parent < p; }) - edges.begin();
    };
    Tree fresh;
    std::vector<std::pair<uint32_t, uint32_t>> queue; // (old id, new id)
    struct NE { uint32_t parent, child; const uint8_t *key; };
    std::vector<NE> kept;
    auto move_node = [&](uint32_t old_id, int creator) {
      const uint32_t nid = fresh.new_node(owner_of(old_id), creator);
      fresh.node(nid) = node(old_id);
      return nid;
    };
    const uint32_t root_new = move_node(new_root, creator_of(new_root)); // (no incoming edge: never reached through a table)
    queue.emplace_back(new_root, root_new);
    for (size_t head = 0; head < queue.size(); ++head) {
      const auto [oid, nid] = queue[head];
      for (size_t q = (size_t)first_edge(oid); q < edges.size() && edges[q].parent == oid; ++q) {
        const uint32_t cn = move_node(edges[q].child, edge_shard(hash_of(nid, edges[q].key)));
        kept.push_back({nid, cn, edges[q].key});
        queue.emplace_back(edges[q].child, cn);
      }
    }
    fresh.reset_tables_for(kept.size());
    for (const NE &x : kept) fresh.insert_edge(x.parent, x.key, x.child);
    for (int o = 0; o < SHARDS; ++o) for (int c = 0; c < SHARDS; ++c) arena[o][c].swap(fresh.arena[o][c]);
    for (int o = 0; o < SHARDS; ++o) std::swap(tab[o], fresh.tab[o]);
    return root_new;
  }
  void reset_tables_for(size_t n_edges) {
    size_t want = 1u << 12;
    while (want * 6 < (n_edges / SHARDS + 64) * 10 * 2) want *= 2;
    for (auto &t : tab) { free(t.e); t.e = (Edge *)calloc(want, sizeof(Edge)); t.cap = t.e ? want : 0; t.gen = 1; t.count = 0; }
  }
  void insert_edge(uint32_t parent, const uint8_t *key, uint32_t child_id) {
    const uint64_t h = hash_of(parent, key);
    Table &t = tab[edge_shard(h)];
    if (!t.e) return;
    if ((t.count + 1) * 10 > t.cap * 6) grow(t);
    const size_t mask = t.cap - 1;
    size_t i = h & mask;
    while (t.e[i].gen == t.gen) i = (i + 1) & mask;
    Edge &e = t.e[i];
    e.gen = t.gen; e.hash = h; e.parent = parent; e.child = child_id; memcpy(e.key, key, 18);
    ++t.count;
  }
};

// W threads (the caller's + W - 1 workers) run one phase function each and meet again: the phases of a batch (select /
// process / back-up) are separated by these joins, so which thread touches which shard when is fixed.
struct Pool {
  int W;
  std::vector<std::thread> th;
  std::mutex mu;
  std::condition_variable cv;
  std::function<void(int)> job;
  std::atomic<uint64_t> phase{0};
  std::atomic<int> remaining{0}, sleepers{0};
  std::atomic<bool> stop{false};
  explicit Pool(int w) : W(w) {
    for (int t = 1; t < W; ++t) th.emplace_back([this, t] { work(t); });
  }
  ~Pool() {
    stop.store(true);
    { std::lock_guard<std::mutex> l(mu); phase.fetch_add(1); }
    cv.notify_all();
    for (auto &t : th) t.join();
  }
  void work(int t) {
    uint64_t seen = 0;
    for (;;) {
      // phases follow each other within microseconds while the host walks a level, and ~100 us apart across a GPU step:
      // spin for about that long, then sleep (a search that waits for a long rollout must not burn W cores)
      int spins = 0;
      while (phase.load(std::memory_order_acquire) == seen) {
        if (++spins < 40000) { __builtin_ia32_pause(); continue; }
        std::unique_lock<std::mutex> l(mu);
        sleepers.fetch_add(1);
        cv.wait(l, [&] { return phase.load(std::memory_order_acquire) != seen; });
        sleepers.fetch_sub(1);
      }
      seen = phase.load(std::memory_order_acquire);
      if (stop.load()) return;
      job(t);
      remaining.fetch_sub(1, std::memory_order_acq_rel);
    }
  }
  template <class F> void run(F &&f) {
    if (W == 1) { f(0); return; }
    job = f;
    remaining.store(W - 1, std::memory_order_release);
    { std::lock_guard<std::mutex> l(mu); phase.fetch_add(1, std::memory_order_acq_rel); }
    if (sleepers.load() > 0) cv.notify_all();
    f(0);
    while (remaining.load(std::memory_order_acquire) != 0) __builtin_ia32_pause();
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
  ~Buffers() { release(); }
  void release() {
    for (void *p : dev) (void)hipFree(p);
    for (void *p : pinned) (void)hipHostFree(p);
    dev.clear();
    pinned.clear();
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

// Diagnostic (no GPU): `count` select + visit rounds of one bandit from a given state, once through Bandit::select_run (the
// register-resident form the batched search uses at the root) and once as the plain select(); visit() loop it must equal.
extern "C" int oakgpu_bandit_select_run(int kind, float c, float alpha, uint32_t k, const float *scores, const float *priors,
                                        const uint32_t *visits, uint32_t count, uint8_t *run_out, uint8_t *loop_out,
                                        uint32_t *visits_run, uint32_t *visits_loop) {
  if (kind < 0 || kind > 4 || k < 1 || k > 9 || !scores || !priors || !visits || !run_out || !loop_out) return oakgpu_fail_msg("oakgpu_bandit_select_run: bad argument");
  const BanditParams P{kind, c, alpha};
  Bandit a;
  a.init((uint8_t)k, kind);
  for (int i = 0; i < 9; ++i) { a.scores[i] = scores[i]; a.priors[i] = priors[i]; a.visits[i] = visits[i]; }
  Bandit b = a;
  uint64_t s1 = 0x1234, s2 = 0x1234;
  a.select_run(P, count, run_out, nullptr, [&](uint32_t) { return uniform01(s1); });
  for (uint32_t r = 0; r < count; ++r) {
    float pr;
    const uint8_t i = b.select(P, [&] { return uniform01(s2); }, pr);
    b.visit(P, i);
    loop_out[r] = i;
  }
  for (int i = 0; i < 9; ++i) { if (visits_run) visits_run[i] = a.visits[i]; if (visits_loop) visits_loop[i] = b.visits[i]; }
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

// RuntimeSearch::Heap (util/search.h:17-32) for Node heaps: the tree of a search, kept between searches.  kind = -1 is
// std::monostate (Heap::empty()); the first search fixes the bandit type, as the variant does (search.cc:205-213).
struct oakgpu_heap {
  Tree tree;
  int kind = -1;
  bool rooted = false; // false: `node = {}` (search.cc:40): a Node whose stats are not initialised
  uint32_t root = 0;
};

namespace {
// One selection of one lane at one tree level (what Bandit::update needs at back-up time), one lane waiting at a node, one
// lane's pair of choice bytes for the device.  All per-SHARD lists: a thread only ever appends to the lists of the shards it
// serves, so no two threads write the same cache line (lane-indexed arrays written by whoever owns the lane's node cost
// more in line ping-pong than the threads gained).
struct Sel { uint32_t lane, node; uint8_t i, j; float prob1, prob2; };
struct LaneNode { uint32_t lane, node; };
struct OutC { uint32_t lane; uint8_t c1, c2; };
struct Route { uint64_t hash; uint32_t lane, parent; uint8_t key[18]; }; // a lane's edge on its way to the thread of the edge's table
constexpr uint32_t NO_NODE = 0xFFFFFFFFu;
// One batch in flight: a context (= HIP stream), its device arrays and the pinned host mirrors of what crosses PCIe at
// every level.  The slots of a caller's context are kept between searches (round-2 advice: a second context and ~50
// allocations were made and destroyed by every search, i.e. once per turn of a self-play game).
struct Slot {
  oakgpu_ctx *ctx = nullptr;
  bool own_ctx = false;
  hipStream_t stream{};
  Buffers buf;
  uint32_t cap = 0;
  static constexpr size_t PACK = 16 + 9 + 9 + 1 + 1 + 1; // actions, both choice lists, result, both counts: bytes per lane per level
  bool has_logits = false;
  int emb_dim = 0;
  uint8_t *d_root_b, *d_root_d, *d_root_r, *d_b, *d_d, *d_r, *d_prng, *d_c1, *d_c2, *d_act, *d_ch1, *d_cnt1, *d_ch2, *d_cnt2, *d_rout;
  uint32_t *d_steps;
  float *d_values, *d_l1 = nullptr, *d_l2 = nullptr, *d_emb = nullptr;
  uint8_t *h_c1, *h_c2, *h_r, *h_act, *h_ch1, *h_cnt1, *h_ch2, *h_cnt2, *h_stage;
  float *h_values, *h_l1 = nullptr, *h_l2 = nullptr;
  std::vector<std::vector<Sel>> log[SHARDS];     // [shard][depth]: the selections made at nodes of that shard, in service order
  std::vector<Route> route[SHARDS][SHARDS];      // [shard of the parent][table of the edge]
  std::vector<Route> work[SHARDS];               // [table of the edge]: its route lists, concatenated
  std::vector<LaneNode> next[2][SHARDS][SHARDS]; // [parity][table of the edge][shard of the child]: lanes that go one level deeper
  std::vector<LaneNode> leafs[SHARDS][SHARDS];   // [table of the edge][shard of the leaf]: lanes that stop at a node to evaluate
  std::vector<OutC> outc[SHARDS][SHARDS];        // [shard][lane block]: choice bytes on their way to the pinned arrays
  std::vector<uint8_t> root_i, root_j;
  std::vector<float> root_p1, root_p2;
  uint32_t levels = 0;
  std::vector<uint8_t> forced;
  double nash1[9], nash2[9];
  uint32_t nb = 0;
  uint64_t serial = 0;
  bool busy = false;
  // the batch's descent as a state machine (round 5): the two slots' levels are interleaved, one slot's per-level tree-step kernel runs
  // while the host selects / resolves for the other
  enum Stage { IDLE, STEP_IN_FLIGHT, EVAL_IN_FLIGHT };
  Stage stage = IDLE;
  uint32_t depth = 0, n_active = 0, lane_block = 0;
  int parity = 0;
  bool mucb = false;
  std::chrono::high_resolution_clock::time_point t_a, t_b; // (OAKGPU_SEARCH_TIMING: when the level's selection began / its kernel was launched)
  ~Slot() { buf.release(); if (own_ctx && ctx) oakgpu_destroy(ctx); }
  uint32_t lay = 0; // the batch size the packed arrays are carved for (<= cap): what a level copies is sized by THIS, not by cap
  // the packed per-level arrays, carved from their blocks for a batch of B lanes (same order on either side)
  void carve(uint32_t B) {
    d_c2 = d_c1 + B;
    d_ch1 = d_act + (size_t)B * 16; d_ch2 = d_ch1 + (size_t)B * 9; d_r = d_ch2 + (size_t)B * 9; d_cnt1 = d_r + B; d_cnt2 = d_cnt1 + B;
    h_c2 = h_c1 + B;
    h_ch1 = h_act + (size_t)B * 16; h_ch2 = h_ch1 + (size_t)B * 9; h_r = h_ch2 + (size_t)B * 9; h_cnt1 = h_r + B; h_cnt2 = h_cnt1 + B;
    lay = B;
  }
  int allocate(oakgpu_ctx *primary, int index, uint32_t B, bool logits, int emb) {
    if (ctx) stream = (hipStream_t)oakgpu_ctx_stream(ctx); // (the caller may have given the context another stream since)
    // kept between searches, re-carved for this search's batch; given back when the last search was more than 8x larger (round-3
    // advice: after one 2^20-lane search every later 1,024-lane search moved 37 MB per tree level)
    if (ctx && cap >= B && cap <= 8 * (size_t)B + 4096 && (has_logits || !logits) && emb_dim >= emb) {
      if (lay != B) { (void)hipStreamSynchronize(stream); carve(B); }
      return 0;
    }
    if (ctx) (void)hipStreamSynchronize(stream);
    buf.release();
    if (!ctx) {
      if (index == 0) ctx = primary;
      else { RC(oakgpu_create(&ctx, oakgpu_ctx_device(primary))); own_ctx = true; }
    }
    stream = (hipStream_t)oakgpu_ctx_stream(ctx);
    cap = 0;
    RC(buf.d(&d_root_b, (size_t)B * 384)); RC(buf.d(&d_root_d, (size_t)B * 8)); RC(buf.d(&d_root_r, (size_t)B));
    RC(buf.d(&d_b, (size_t)B * 384)); RC(buf.d(&d_d, (size_t)B * 8)); RC(buf.d(&d_prng, (size_t)B * 8));
    // what a level sends down (two choice bytes per lane) and what comes back (37 bytes per lane) each travel as ONE copy:
    // the arrays are carved from one block on either side, in the same order
    RC(buf.d(&d_c1, (size_t)B * 2));
    RC(buf.d(&d_act, (size_t)B * PACK));
    RC(buf.d(&d_rout, (size_t)B)); RC(buf.d(&d_steps, (size_t)B)); RC(buf.d(&d_values, (size_t)B));
    d_l1 = d_l2 = d_emb = nullptr; h_l1 = h_l2 = nullptr;
    if (logits) { RC(buf.d(&d_l1, (size_t)B * 9)); RC(buf.d(&d_l2, (size_t)B * 9)); }
    if (emb) RC(buf.d(&d_emb, (size_t)B * emb));
    RC(buf.h(&h_c1, (size_t)B * 2));
    RC(buf.h(&h_act, (size_t)B * PACK));
    carve(B);
    RC(buf.h(&h_values, (size_t)B)); RC(buf.h(&h_stage, (size_t)B * 384));
    if (logits) { RC(buf.h(&h_l1, (size_t)B * 9)); RC(buf.h(&h_l2, (size_t)B * 9)); }
    root_i.resize(B); root_j.resize(B); root_p1.resize(B); root_p2.resize(B);
    cap = B; has_logits = logits; emb_dim = emb;
    return 0;
  }
};
struct SearchScratch { Slot slots[2]; };
thread_local int tl_search_threads = 0; // > 0: host threads of the searches started by THIS thread (set by oakgpu_search_many's workers)
void scratch_dtor(void *p) { delete (SearchScratch *)p; }
} // namespace

extern "C" {

int oakgpu_heap_create(oakgpu_heap **out) {
  if (!out) return oakgpu_fail_msg("oakgpu_heap_create: null out");
  *out = new oakgpu_heap();
  return 0;
}
void oakgpu_heap_destroy(oakgpu_heap *h) { delete h; }
int oakgpu_heap_empty(const oakgpu_heap *h) { return !h || h->kind < 0; }
uint64_t oakgpu_heap_nodes(const oakgpu_heap *h) { return h && h->rooted ? h->tree.size() : 0; }
int oakgpu_heap_kind(const oakgpu_heap *h) { return h ? h->kind : -1; }
void oakgpu_heap_clear(oakgpu_heap *h) { if (h) { h->tree.reset(0); h->rooted = false; h->kind = -1; } }

// Heap::update(i, j, obs) (search.cc:27-52): 1 = the child reached by (i, j, obs) is the root now (its whole subtree, bandit
// statistics included, is kept; everything else is dropped); 0 = nothing to keep: an empty heap, a root that was never
// initialised, or an edge the searches never took (the heap then holds an uninitialised node, like `node = {}`).
int oakgpu_heap_update(oakgpu_heap *h, uint8_t i, uint8_t j, const uint8_t *obs16) {
  if (!h || !obs16) return 0;
  if (h->kind < 0 || !h->rooted) return 0;
  if (!h->tree.node(h->root).is_init()) return 0;
  uint8_t key[18];
  key[0] = i; key[1] = j;
  memcpy(key + 2, obs16, 16);
  uint32_t child;
  if (!h->tree.find_child(h->root, key, &child)) {
    h->tree.reset(0);
    h->rooted = false;
    return 0;
  }
  h->root = h->tree.keep_subtree(child);
  return 1;
}

// Diagnostic: live edges whose child does NOT sit in an arena created by the edge's own table.  0 in a consistent tree; the
// resolve phase's threading relies on it (Tree::keep_subtree re-establishes it after a promotion).
uint64_t oakgpu_heap_check_shards(const oakgpu_heap *h) {
  if (!h) return 0;
  uint64_t bad = 0;
  for (int t = 0; t < SHARDS; ++t) {
    const Tree::Table &tb = h->tree.tab[t];
    for (size_t q = 0; q < tb.cap; ++q)
      if (tb.e[q].gen == tb.gen) {
        const Edge &e = tb.e[q];
        bad += creator_of(e.child) != t || Tree::edge_shard(e.hash) != t || Tree::hash_of(e.parent, e.key) != e.hash;
      }
  }
  return bad;
}

// Host-only self-test of the sharded tree (no GPU): grows a random tree with the search's own resolve phase -- edges routed to
// the table their hash picks, `threads` threads each serving tables t = thread mod threads, every thread creating children
// and reading whether the child it found is initialised -- promotes a random child of the root (Heap::update), checks the
// shard invariant, and grows the promoted tree again with the same threads.  Returns 0 when every check holds, else a code;
// out[0] = nodes before the promotion, out[1] = nodes kept, out[2] = nodes at the end, out[3] = shard violations seen.
int oakgpu_heap_selftest(uint32_t rounds, uint32_t lanes, uint64_t seed, int threads, uint64_t out[4]) {
  if (threads != 1 && threads != 2 && threads != 4 && threads != 8 && threads != 16) return oakgpu_fail_msg("oakgpu_heap_selftest: threads must be 1, 2, 4, 8 or 16");
  oakgpu_heap H;
  Tree &tree = H.tree;
  tree.reset((size_t)rounds * lanes);
  if (!tree.ok()) return oakgpu_fail_msg("oakgpu_heap_selftest: out of memory");
  H.kind = 0; H.rooted = true;
  H.root = tree.new_node(0, 0);
  tree.node(H.root).p1.init(3, 0); tree.node(H.root).p2.init(3, 0);
  Pool pool(threads);
  uint64_t rng = seed * 0x9E3779B97F4A7C15ull + 1;
  auto next = [&] { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return rng; };
  std::vector<uint32_t> nodes{H.root};
  std::vector<Route> route[SHARDS];
  std::vector<uint32_t> made[SHARDS];
  auto grow_rounds = [&](uint32_t n) {
    for (uint32_t r = 0; r < n; ++r) {
      for (auto &v : route) v.clear();
      for (uint32_t l = 0; l < lanes; ++l) { // a lane walks from a known initialised node along a random edge
        Route x{};
        x.lane = l;
        x.parent = nodes[next() % nodes.size()];
        x.key[0] = (uint8_t)(next() % 3); x.key[1] = (uint8_t)(next() % 3); x.key[2] = (uint8_t)(next() % 4);
        x.hash = Tree::hash_of(x.parent, x.key);
        route[Tree::edge_shard(x.hash)].push_back(x);
      }
      pool.run([&](int w) {
        for (int t = w; t < SHARDS; t += threads) {
          made[t].clear();
          for (const Route &x : route[t]) {
            const uint32_t c = tree.child(x.parent, x.key, x.hash);
            if (!tree.node(c).is_init()) made[t].push_back(c); // (the read the promotion used to race with)
          }
        }
      });
      pool.run([&](int w) { // initialise the new leaves: by the thread of the node's OWNER shard, as finish() does
        for (int own = w; own < SHARDS; own += threads)
          for (int t = 0; t < SHARDS; ++t)
            for (uint32_t c : made[t])
              if (owner_of(c) == own && !tree.node(c).is_init()) { tree.node(c).p1.init(3, 0); tree.node(c).p2.init(3, 0); }
      });
      for (int t = 0; t < SHARDS; ++t) for (uint32_t c : made[t]) nodes.push_back(c);
      std::sort(nodes.begin(), nodes.end());
      nodes.erase(std::unique(nodes.begin(), nodes.end()), nodes.end());
    }
  };
  grow_rounds(rounds);
  uint64_t bad = oakgpu_heap_check_shards(&H);
  out[0] = tree.size();
  if (out[0] != nodes.size()) return 2;
  // promote the root's most travelled kind of child: any existing edge of the root
  uint32_t child = 0;
  bool found = false;
  for (int a = 0; a < 3 && !found; ++a) for (int b = 0; b < 3 && !found; ++b) for (int c = 0; c < 4 && !found; ++c) {
    uint8_t key[18] = {(uint8_t)a, (uint8_t)b, (uint8_t)c};
    found = tree.find_child(H.root, key, &child);
  }
  if (!found) return 3;
  H.root = tree.keep_subtree(child);
  out[1] = tree.size();
  bad += oakgpu_heap_check_shards(&H);
  // every kept node is reachable from the new root through the re-hashed tables, initialised, and counted once
  nodes.assign(1, H.root);
  for (size_t head = 0; head < nodes.size(); ++head)
    for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) for (int c = 0; c < 4; ++c) {
      uint8_t key[18] = {(uint8_t)a, (uint8_t)b, (uint8_t)c};
      uint32_t ch;
      if (tree.find_child(nodes[head], key, &ch)) { if (!tree.node(ch).is_init()) return 4; nodes.push_back(ch); }
    }
  if (nodes.size() != out[1]) return 5;
  grow_rounds(rounds);
  out[2] = tree.size();
  bad += oakgpu_heap_check_shards(&H);
  out[3] = bad;
  if (out[2] != nodes.size()) return 6;
  return bad ? 7 : 0;
}

// Test / diagnostic view of the root's two bandits: scores[9], priors[9], visits[9], k per player (0 = not initialised).
int oakgpu_heap_root_stats(const oakgpu_heap *h, int player, float *scores, float *priors, uint32_t *visits, uint8_t *k) {
  if (!h || player < 0 || player > 1) return oakgpu_fail_msg("oakgpu_heap_root_stats: bad argument");
  if (h->kind < 0 || !h->rooted) { if (k) *k = 0; return 0; }
  const NodeRec &nd = const_cast<oakgpu_heap *>(h)->tree.node(h->root);
  const Bandit &b = player ? nd.p2 : nd.p1;
  if (k) *k = b.k;
  for (int q = 0; q < 9; ++q) { if (scores) scores[q] = b.scores[q]; if (priors) priors[q] = b.priors[q]; if (visits) visits[q] = b.visits[q]; }
  return 0;
}

// The same view of the child that oakgpu_heap_update(i, j, obs) WOULD promote (k = 0: no such child, or not initialised).
int oakgpu_heap_child_stats(const oakgpu_heap *h, uint8_t i, uint8_t j, const uint8_t *obs16, int player, float *scores, float *priors,
                            uint32_t *visits, uint8_t *k) {
  if (!h || !obs16 || player < 0 || player > 1) return oakgpu_fail_msg("oakgpu_heap_child_stats: bad argument");
  if (k) *k = 0;
  if (h->kind < 0 || !h->rooted) return 0;
  uint8_t key[18];
  key[0] = i; key[1] = j;
  memcpy(key + 2, obs16, 16);
  uint32_t child;
  if (!h->tree.find_child(h->root, key, &child)) return 0;
  const NodeRec &nd = const_cast<oakgpu_heap *>(h)->tree.node(child);
  const Bandit &b = player ? nd.p2 : nd.p1;
  if (k) *k = b.k;
  for (int q = 0; q < 9; ++q) { if (scores) scores[q] = b.scores[q]; if (priors) priors[q] = b.priors[q]; if (visits) visits[q] = b.visits[q]; }
  return 0;
}

int oakgpu_search_heap(oakgpu_ctx *ctx, oakgpu_net *net, oakgpu_heap *heap, const uint8_t *battle, const uint8_t *durations, uint8_t result,
                       const oakgpu_search_params *prm, const oakgpu_search_output *previous, oakgpu_search_output *out) {
  if (!ctx || !battle || !durations || !prm || !out) return oakgpu_fail_msg("oakgpu_search: null argument");
  const bool pucb = prm->bandit == B_PUCB || prm->bandit == B_PEXP3; // the bandits that take priors from the policy heads
  const bool use_net = prm->eval == 1, use_pe = prm->eval == 2;
  // exp3_alpha < 0 (or NaN) = the reference's default 0.05 (search.cc:268-270: used when the third field is absent); an
  // explicit 0 is honoured, as the reference honours it
  const BanditParams BP{prm->bandit, prm->ucb_c, prm->exp3_alpha >= 0 ? prm->exp3_alpha : 0.05f};
  if (prm->bandit < 0 || prm->bandit > 4 || prm->eval < 0 || prm->eval > 2) return oakgpu_fail_msg("oakgpu_search: unknown bandit / eval");
  if ((use_net || pucb) && !net) return oakgpu_fail_msg("oakgpu_search: network evaluation / PUCB / PExp3 priors need a network");
  if (pucb && !use_net) return oakgpu_fail_msg("oakgpu_search: PUCB / PExp3 take their priors from the network evaluator (eval = 1)");
  if (prm->batch == 0 || prm->batch > (1u << 20)) return oakgpu_fail_msg("oakgpu_search: batch must be in 1..2^20");
  auto rolls_ok = [](uint32_t r) { return r == 1 || r == 2 || r == 3 || r == 20 || r == 39; };
  if (!rolls_ok(prm->root_rolls) || !rolls_ok(prm->other_rolls)) return oakgpu_fail_msg("oakgpu_search: rolls must be 1, 2, 3, 20 or 39");
  HIPRC(hipSetDevice(oakgpu_ctx_device(ctx)));
  const uint32_t B = prm->batch;
  const uint32_t max_depth = prm->max_depth ? prm->max_depth : 100;
  // MCTS::Search::run takes `Output output = {}` BY VALUE and adds to it (mcts.h:153-155): the matrices, `iterations` and
  // `duration` of a previous search of the SAME position accumulate (:231-247); the logits / priors / initial value
  // stay as they came in unless the root is fresh (:177-210)
  oakgpu_search_output prev;
  if (previous) prev = *previous; else memset(&prev, 0, sizeof prev);
  memset(out, 0, sizeof *out);
  if (previous) {
    memcpy(out->visit_matrix, prev.visit_matrix, sizeof out->visit_matrix);
    memcpy(out->value_matrix, prev.value_matrix, sizeof out->value_matrix);
    out->initial_value = prev.initial_value;
    memcpy(out->p1_logit, prev.p1_logit, sizeof out->p1_logit); memcpy(out->p2_logit, prev.p2_logit, sizeof out->p2_logit);
    memcpy(out->p1_prior, prev.p1_prior, sizeof out->p1_prior); memcpy(out->p2_prior, prev.p2_prior, sizeof out->p2_prior);
  }
  const uint64_t base_iterations = prev.iterations;

  // root choices (mcts.h:160-166) through the batched choices kernel, batch of one
  uint8_t root_c1[9], root_c2[9], m = 0, n = 0;
  RC(oakgpu_choices(ctx, battle, &result, 0, root_c1, &m, 1));
  RC(oakgpu_choices(ctx, battle, &result, 1, root_c2, &n, 1));
  out->m = m;
  out->n = n;
  memcpy(out->p1_choices, root_c1, 9);
  memcpy(out->p2_choices, root_c2, 9);
  if ((result & 15) != 0 || m == 0 || n == 0) return oakgpu_fail_msg("oakgpu_search: the root position is terminal");

  const bool timed = prm->duration_us != 0; // time budget (search.cc:300-306): batches are started until it has elapsed
  const size_t expected = timed ? (size_t)1 << 20 : (size_t)std::min<uint64_t>(prm->iterations, (uint64_t)1 << 24);
  static thread_local oakgpu_heap scratch_heap; // heap == NULL: a fresh tree per search, in the memory of the thread's last one
  oakgpu_heap &H = heap ? *heap : scratch_heap;
  if (!heap) { H.kind = -1; H.rooted = false; }
  if (H.kind >= 0 && H.kind != BP.kind) {
    static const char *names[5] = {"UCB", "PUCB", "UCB1", "Exp3", "PExp3"};
    return oakgpu_fail_msg((std::string("RuntimeSearch: Bad Heap access. Expecting MCTS::Node<") + names[BP.kind] + "::JointBandit>").c_str());
  }
  H.kind = BP.kind;
  Tree &tree = H.tree;
  if (!H.rooted) {
    tree.reset(expected);
    if (!tree.ok()) return oakgpu_fail_msg("oakgpu_search: out of host memory (edge tables)");
    H.root = tree.new_node(0, 0);
    H.rooted = true;
  }
  const uint32_t root = H.root;
  if (tree.node(root).is_init() && (tree.node(root).p1.k != m || tree.node(root).p2.k != n))
  {
    (void)oakgpu_fail_msg("oakgpu_search: the heap's root has other action counts than this position (Heap::update was not called with the move that was played?)");
    return OAKGPU_E_ROOT_MISMATCH;
  }
  if (!tree.node(root).is_init()) { // stats.init(k1, k2) + priors (mcts.h:177-210)
    tree.node(root).p1.init(m, BP.kind);
    tree.node(root).p2.init(n, BP.kind);
    if (pucb) { // root priors from the policy heads (mcts.h:196-209)
      float v, l1[9], l2[9];
      RC(oakgpu_leaf_eval_policy(ctx, net, battle, durations, 1, root_c1, &m, root_c2, &n, &v, l1, l2));
      tree.node(root).p1.set_logits(BP, l1);
      tree.node(root).p2.set_logits(BP, l2);
      out->initial_value = v;
      auto soft = [](double *o, const float *l, int k) { // softmax(output.p1.prior, logits, k): search/util/softmax.h:5-15
        float sum = 0;
        for (int i = 0; i < k; ++i) { const float y = std::exp(l[i]); o[i] = y; sum += y; }
        for (int i = 0; i < k; ++i) o[i] /= sum;
      };
      memset(out->p1_logit, 0, sizeof out->p1_logit); memset(out->p2_logit, 0, sizeof out->p2_logit);
      memset(out->p1_prior, 0, sizeof out->p1_prior); memset(out->p2_prior, 0, sizeof out->p2_prior);
      for (int i = 0; i < m; ++i) out->p1_logit[i] = l1[i];
      for (int j = 0; j < n; ++j) out->p2_logit[j] = l2[j];
      soft(out->p1_prior, l1, m);
      soft(out->p2_prior, l2, n);
    }
  }

  float pe_root = 0.0f; // PokeEngine::Eval::get_root_score (mcts.h:172-174)
  if (use_pe) RC(oakgpu_poke_engine_eval(ctx, battle, 1, 0.0f, nullptr, &pe_root));

  // Two batches are kept in flight ("slots", each with its own context = HIP stream and buffers): while the GPU steps or
  // evaluates one batch, the host walks the tree for the other.  The schedule is a fixed function of the batches' depths
  // (see the loop at the end), so a search is reproducible.
  // One batch at a time only for batch = 1, which is the reference's strictly sequential iteration order (the evaluator's
  // workspaces belong to the context, so two slots -- two contexts -- can evaluate the same network concurrently).
  const int n_slots = (B > 1 && (timed || prm->iterations > B)) ? 2 : 1;
  SearchScratch *scratch = (SearchScratch *)oakgpu_ctx_attachment(ctx);
  if (!scratch) { scratch = new SearchScratch(); oakgpu_ctx_set_attachment(ctx, scratch, scratch_dtor); }
  Slot *slots = scratch->slots;
  int emb_dim = 0;
  if (use_net && n_slots > 1) RC(oakgpu_net_shape(net, &emb_dim, nullptr, nullptr, nullptr));
  struct HintGuard { // both slots' contexts run their small launches in rounds while two batches are in flight
    oakgpu_ctx *c[2] = {nullptr, nullptr};
    int old[2] = {0, 0};
    ~HintGuard() { for (int q = 0; q < 2; ++q) if (c[q]) oakgpu_ctx_set_concurrent_hint(c[q], old[q]); }
  } hint_guard;
  for (int si = 0; si < n_slots; ++si) {
    Slot &S = slots[si];
    RC(S.allocate(ctx, si, B, pucb, emb_dim));
    if (n_slots > 1) { hint_guard.c[si] = S.ctx; hint_guard.old[si] = oakgpu_ctx_set_concurrent_hint(S.ctx, 1); }
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
    S.busy = false;
  }

  // host threads of the tree walk: 1, 2, 4 or 8 (OAKGPU_SEARCH_THREADS; default 8 where the process may use >= 16 CPUs, else
  // half of them; 1 for small batches).  Results do not depend on the count (see Tree)
  // (round 5: 16 shards, and every usable core walks -- one tree on a 16-core job: 7.4 M iterations/s on 8 threads, 9.4 M on 16;
  // on a box with fewer than 16 usable cores half of them, as before: the caller's own threads need some)
  int W = 16;
  const char *wenv = getenv("OAKGPU_SEARCH_THREADS");
  if (tl_search_threads > 0) W = tl_search_threads; // (oakgpu_search_many: the cores are shared by the concurrent searches)
  else if (wenv) W = atoi(wenv);
  else {
    const unsigned hc = oakgpu_usable_cores();
    if (hc < 16) { W = 8; while (W > 1 && (unsigned)W * 2 > hc) W /= 2; }
  }
  if (B < 1024) W = 1;
  W = W >= 16 ? 16 : W >= 8 ? 8 : W >= 4 ? 4 : W >= 2 ? 2 : 1;
  Pool pool(W);

  double total_value = 0;
  const bool timing = getenv("OAKGPU_SEARCH_TIMING") != nullptr;
  double t_sel = 0, t_gpu = 0, t_proc = 0, t_eval = 0, t_back = 0, t_sel_d[4] = {}, t_proc_d[4] = {};
  uint64_t n_d[4] = {};
  auto now = [] { return std::chrono::high_resolution_clock::now(); };
  auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
  uint64_t done = 0, started = 0, total_depth = 0, serial = 0;
  uint64_t mucb_rng = prm->seed ^ 0xA0761D6478BD642Full;
  const uint64_t bandit_seed = prm->seed ^ 0xE7037ED1A0B428DBull;
  // the budget's clock starts HERE, after the set-up above (round-2 advice: with t_start in front of ~50 allocations a short
  // budget on a cold context was spent before the first batch); the reference times only its iteration loop (mcts.h:213-245)
  const auto t_start = now();

  // one batch: root prep, level-synchronous descent (host selection <-> k_tree_step), then the leaf evaluation is
  // LAUNCHED (not awaited).  Service order at a node = (shard of the lane's previous node, lane): fixed by the tree, not by
  // the number of threads.
  const int rs = owner_of(root);
  auto shards_of = [&](int w, auto &&fn) { for (int sh = w; sh < SHARDS; sh += W) fn(sh); };
  auto begin_batch = [&](Slot &S) -> int {
    const uint32_t nb = timed ? B : (uint32_t)std::min<uint64_t>(B, prm->iterations - started);
    S.nb = nb;
    S.serial = serial++;
    started += nb;
    S.busy = true;
    // root prep on the device (mcts.h:254-259): rollout kernel with max_steps = 0
    RC(oakgpu_rollout_dev(S.ctx, S.d_root_b, S.d_root_d, S.d_root_r, S.d_prng, nb, 0, 1, S.d_rout, S.d_steps, S.d_values, S.d_b, S.d_d));
    HIPRC(hipMemcpyAsync(S.d_r, S.d_root_r, nb, hipMemcpyDeviceToDevice, S.stream));
    for (auto &a : S.leafs) for (auto &v : a) v.clear();
    for (auto &par : S.next) for (auto &a : par) for (auto &v : a) v.clear();
    S.n_active = nb;
    const uint32_t block = (nb + SHARDS - 1) / SHARDS; // lane blocks of the scatter phase
    S.lane_block = ((block + 63) / 64) * 64;
    // MatrixUCB (mcts.h:263-302): the root's joint actions of this batch come from the UCB matrices, not the bandits
    const uint64_t so_far = base_iterations + done; // output.iterations at this point (mcts.h:270)
    const bool mucb = prm->matrix_ucb && so_far >= prm->mucb_delay;
    S.mucb = mucb;
    if (mucb) {
      S.forced.clear();
      uint64_t planned[81];
      for (int q = 0; q < 81; ++q) planned[q] = out->visit_matrix[q];
      for (int i = 0; i < m && S.forced.size() < nb; ++i) // cells under the minimum visit count go first (mcts.h:510-512)
        for (int j = 0; j < n && S.forced.size() < nb; ++j)
          while (planned[i * 9 + j] < prm->mucb_minimum && S.forced.size() < nb) { S.forced.push_back((uint8_t)(i * 9 + j)); ++planned[i * 9 + j]; }
      if (S.forced.size() < nb) {
        int32_t up[81], dn[81];
        const double log_T = std::log((double)(so_far ? so_far : 1)), w = std::log(2.0 * m * n);
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
    S.depth = 0;
    S.parity = 0;
    return 0;
  };
  // one level, first half: bandit selection on the host, then the level's tree-step kernel is LAUNCHED (copies included) -- not awaited
  auto select_launch = [&](Slot &S) -> int {
    const uint32_t nb = S.nb, depth = S.depth, lane_block = S.lane_block;
    const int parity = S.parity;
    const bool mucb = S.mucb;

      const auto ta = now();
      for (int sh = 0; sh < SHARDS; ++sh) {
        if (S.log[sh].size() <= depth) S.log[sh].resize(depth + 1);
      }
      if (depth == 0) {
        // Every lane is at the root.  Its two bandits are independent of each other (each sees only its own virtual losses),
        // so player 1's and player 2's selection sequences run on two threads, on private copies of the bandits.
        if (mucb) { // sampled / forced root actions; the root bandits are neither consulted nor updated
          for (uint32_t l = 0; l < nb; ++l) {
            uint8_t i, j;
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
            S.root_i[l] = i; S.root_j[l] = j; S.root_p1[l] = 1.0f; S.root_p2[l] = 1.0f;
          }
        } else {
          NodeRec &nd = tree.node(root);
          pool.run([&](int w) {
            if (w == 0) {
              Bandit b = nd.p1;
              b.select_run(BP, nb, S.root_i.data(), S.root_p1.data(), [&](uint32_t l) { return uniform_at(bandit_seed, S.serial, l, 0, 0); }); // device.uniform() of sample_pdf (util/random.h:40-49)
              nd.p1 = b;
            }
            if (w == (W > 1 ? 1 : 0)) {
              Bandit b = nd.p2;
              b.select_run(BP, nb, S.root_j.data(), S.root_p2.data(), [&](uint32_t l) { return uniform_at(bandit_seed, S.serial, l, 0, 1); });
              nd.p2 = b;
            }
          });
        }
        auto &lg = S.log[rs][0];
        for (int sh = 0; sh < SHARDS; ++sh) S.log[sh][0].clear();
        lg.resize(nb);
        for (uint32_t l = 0; l < nb; ++l) {
          lg[l] = Sel{l, mucb ? NO_NODE : root, S.root_i[l], S.root_j[l], S.root_p1[l], S.root_p2[l]};
          S.h_c1[l] = root_c1[S.root_i[l]];
          S.h_c2[l] = root_c2[S.root_j[l]];
        }
      } else {
        // bandit selection: the thread of a shard serves the lanes waiting at that shard's nodes -- all lanes of one node by
        // one thread, in service order, each seeing the virtual losses of those before it
        pool.run([&](int w) {
          shards_of(w, [&](int sh) {
            auto &lg = S.log[sh][depth];
            lg.clear();
            for (auto &v : S.outc[sh]) v.clear();
            for (int p = 0; p < SHARDS; ++p) {
              const auto &todo = S.next[parity][p][sh];
              const size_t cnt = todo.size();
              for (size_t q = 0; q < cnt; ++q) {
                if (q + 10 < cnt) tree.prefetch_node(todo[q + 10].node);
                const uint32_t l = todo[q].lane, id = todo[q].node;
                NodeRec &nd = tree.node(id);
                float pr1, pr2;
                const uint8_t i = nd.p1.select(BP, [&] { return uniform_at(bandit_seed, S.serial, l, depth, 0); }, pr1);
                const uint8_t j = nd.p2.select(BP, [&] { return uniform_at(bandit_seed, S.serial, l, depth, 1); }, pr2);
                nd.p1.visit(BP, i);
                nd.p2.visit(BP, j);
                lg.push_back(Sel{l, id, i, j, pr1, pr2});
                S.outc[sh][l / lane_block].push_back(OutC{l, S.h_ch1[(size_t)l * 9 + i], S.h_ch2[(size_t)l * 9 + j]});
              }
            }
          });
        });
        // the choice bytes go to the pinned lane-indexed arrays by lane BLOCK (64-lane aligned): one writer per cache line
        pool.run([&](int w) {
          shards_of(w, [&](int blk) {
            const uint32_t lo = (uint32_t)blk * lane_block, hi = std::min(nb, lo + lane_block);
            if (lo >= hi) return;
            memset(S.h_c1 + lo, 0xFF, hi - lo); // 0xFF: the lane is finished, leave it untouched
            memset(S.h_c2 + lo, 0xFF, hi - lo);
            for (int sh = 0; sh < SHARDS; ++sh)
              for (const OutC &o : S.outc[sh][blk]) { S.h_c1[o.lane] = o.c1; S.h_c2[o.lane] = o.c2; }
          });
        });
      }
      const auto tb = now();
      HIPRC(hipMemcpyAsync(S.d_c1, S.h_c1, (size_t)S.lay + nb, hipMemcpyHostToDevice, S.stream)); // c1[0, lay) + c2[0, nb)
      RC(oakgpu_tree_step_dev(S.ctx, S.d_b, S.d_d, S.d_r, S.d_c1, S.d_c2, nb, depth == 0 ? prm->root_rolls : prm->other_rolls, S.d_act,
                              S.d_ch1, S.d_cnt1, S.d_ch2, S.d_cnt2));
      HIPRC(hipMemcpyAsync(S.h_act, S.d_act, (size_t)S.lay * Slot::PACK, hipMemcpyDeviceToHost, S.stream));
    S.t_a = ta;
    S.t_b = tb;
    return 0;
  };
  // one level, second half: wait for the kernel, route and resolve the lanes' edges; the lanes that reached a new node stop there
  auto wait_process = [&](Slot &S) -> int {
    const uint32_t depth = S.depth;
    int parity = S.parity;
    uint32_t n_active = S.n_active;
    uint32_t shard_done[SHARDS];
    const auto ta = S.t_a, tb = S.t_b, tw = now();
      HIPRC(hipStreamSynchronize(S.stream));
      const auto tc = now();
      // edges, in two steps.  Route: the thread of the PARENT's shard hashes its lanes' edges (parent, i, j, observation) and
      // hands each to the table its hash picks.  Resolve: the thread of that table looks the child up, or creates it, in
      // the order (parent's shard, service order); the lane goes on to the child's shard or stops there for evaluation.
      for (int q = 0; q < SHARDS; ++q) shard_done[q] = 0;
      pool.run([&](int w) {
        shards_of(w, [&](int sh) {
          for (auto &v : S.route[sh]) v.clear();
          uint32_t fin = 0;
          const auto &lg = S.log[sh][depth];
          const size_t cnt = lg.size();
          for (size_t q = 0; q < cnt; ++q) {
            if (q + 12 < cnt) { __builtin_prefetch(S.h_act + (size_t)lg[q + 12].lane * 16); __builtin_prefetch(S.h_r + lg[q + 12].lane); } // (fresh from the DMA: in no cache)
            const Sel &e = lg[q];
            if ((S.h_r[e.lane] & 15) != 0) { ++fin; continue; } // terminal edge: the value comes from the result byte (mcts.h:427-441)
            Route r;
            r.key[0] = e.i; r.key[1] = e.j;
            memcpy(r.key + 2, S.h_act + (size_t)e.lane * 16, 16);
            r.lane = e.lane;
            r.parent = e.node == NO_NODE ? root : e.node;
            r.hash = Tree::hash_of(r.parent, r.key);
            S.route[sh][Tree::edge_shard(r.hash)].push_back(r);
          }
          shard_done[sh] = fin;
        });
      });
      for (int q = 0; q < SHARDS; ++q) { n_active -= shard_done[q]; total_depth += (uint64_t)shard_done[q] * (depth + 1); shard_done[q] = 0; }
      pool.run([&](int w) {
        shards_of(w, [&](int t) {
          for (auto &v : S.next[parity ^ 1][t]) v.clear();
          auto &wk = S.work[t];
          wk.clear();
          for (int sh = 0; sh < SHARDS; ++sh) wk.insert(wk.end(), S.route[sh][t].begin(), S.route[sh][t].end());
          const size_t cnt = wk.size();
          uint32_t fin = 0;
          constexpr size_t LOOK = 8; // two-stage pipeline: the edge of entry q + LOOK is resolved (child created if new, its node's
          uint32_t pend[LOOK];       // line requested) while entry q reads whether its child is initialised
          auto resolve = [&](size_t q) -> uint32_t {
            const Route &r = wk[q];
            const uint32_t child = tree.child(r.parent, r.key, r.hash);
            tree.prefetch_node(child);
            return child;
          };
          for (size_t q = 0; q < std::min(cnt, 2 * LOOK); ++q) tree.prefetch_edge(wk[q].hash);
          for (size_t q = 0; q < std::min(cnt, LOOK); ++q) pend[q] = resolve(q);
          for (size_t q = 0; q < cnt; ++q) {
            const uint32_t child = pend[q % LOOK];
            if (q + 2 * LOOK < cnt) tree.prefetch_edge(wk[q + 2 * LOOK].hash);
            if (q + LOOK < cnt) pend[q % LOOK] = resolve(q + LOOK);
            const uint32_t lane = wk[q].lane;
            if (tree.node(child).is_init() && depth + 1 < max_depth) { S.next[parity ^ 1][t][owner_of(child)].push_back({lane, child}); continue; }
            S.leafs[t][owner_of(child)].push_back({lane, child}); // first visit (or depth cap): evaluate here (mcts.h:391-426)
            ++fin;
          }
          shard_done[t] = fin;
        });
      });
      parity ^= 1;
      for (int q = 0; q < SHARDS; ++q) { n_active -= shard_done[q]; total_depth += (uint64_t)shard_done[q] * (depth + 1); }
      S.levels = depth + 1;
      const auto td = now();
      t_sel += us(ta, tb); t_gpu += us(tw, tc); t_proc += us(tc, td); // (gpu-step = what the host WAITED for the level's kernel)
      if (timing) { const int dd = depth < 3 ? (int)depth : 3; t_sel_d[dd] += us(ta, tb); t_proc_d[dd] += us(tc, td); for (int q = 0; q < SHARDS; ++q) n_d[dd] += S.log[q][depth].size(); }
        S.parity = parity;
    S.n_active = n_active;
    S.depth = depth + 1;
    return 0;
  };
  auto launch_eval = [&](Slot &S) -> int {
    const uint32_t nb = S.nb;
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
    auto value_of = [&](uint32_t l) {
      const uint32_t t = S.h_r[l] & 15;
      return t != 0 ? (t == 1 ? 1.0f : t == 2 ? 0.0f : 0.5f) : S.h_values[l];
    };
    pool.run([&](int w) { // every node's bandits are written by the thread of its owner shard, in service order
      shards_of(w, [&](int sh) {
        for (int p = 0; p < SHARDS; ++p)
          for (const LaneNode &e : S.leafs[p][sh]) { // stats.init(m, n) (+ priors) at the first evaluation
            NodeRec &lf = tree.node(e.node);
            const uint32_t l = e.lane;
            if (lf.is_init() || !S.h_cnt1[l] || !S.h_cnt2[l]) continue;
            lf.p1.init(S.h_cnt1[l], BP.kind);
            lf.p2.init(S.h_cnt2[l], BP.kind);
            if (pucb) { lf.p1.set_logits(BP, S.h_l1 + (size_t)l * 9); lf.p2.set_logits(BP, S.h_l2 + (size_t)l * 9); }
          }
        for (uint32_t d = 0; d < S.levels; ++d) {
          const auto &lg = S.log[sh][d];
          const size_t cnt = lg.size();
          for (size_t q = 0; q < cnt; ++q) { // Bandit::update, the visit was already counted as the virtual loss
            if (q + 10 < cnt && lg[q + 10].node != NO_NODE) tree.prefetch_node(lg[q + 10].node);
            const Sel &e = lg[q];
            if (e.node == NO_NODE) continue; // (MatrixUCB root step: only the root matrices below are updated)
            const float v1 = value_of(e.lane);
            NodeRec &nd = tree.node(e.node);
            nd.p1.update(BP, e.i, v1, e.prob1);
            nd.p2.update(BP, e.j, 1.0f - v1, e.prob2);
          }
        }
      });
    });
    for (const Sel &e : S.log[rs][0]) { // the root matrices, in lane order
      const float v1 = value_of(e.lane);
      ++out->visit_matrix[e.i * 9 + e.j];
      out->value_matrix[e.i * 9 + e.j] += v1;
      total_value += v1;
    }
    t_eval += us(te, tf); t_back += us(tf, now());
    done += nb;
    S.busy = false;
    return 0;
  };
  auto more = [&] {
    if (!timed) return started < prm->iterations;
    // `while (elapsed < duration)` with elapsed = 0 at first (mcts.h:219-226): a time budget always runs at least once
    return started == 0 || us(t_start, now()) < (double)prm->duration_us;
  };
  // The schedule (round 5): the slots take turns, each advancing by ONE stage per turn -- a level's second half + the next level's
  // first half, or the back-up of an evaluated batch + the start of the next -- so while the host selects / resolves for one slot
  // the other slot's tree-step kernel (or leaf evaluation) is in flight.  (Rounds 2-4: a slot descended through ALL its levels in
  // one go, waiting for every level's kernel: 5.6 of 37 ms per 2^18 iterations.)  The order depends on the batches' own depths
  // only, never on timing: a search is reproducible, and equal to the same search run beside others.
  for (int si = 0; si < n_slots; ++si) slots[si].stage = Slot::IDLE;
  for (;;) {
    bool any = false;
    for (int si = 0; si < n_slots; ++si) {
      Slot &S = slots[si];
      if (S.stage == Slot::STEP_IN_FLIGHT) {
        RC(wait_process(S));
        if (S.n_active > 0) RC(select_launch(S));
        else { RC(launch_eval(S)); S.stage = Slot::EVAL_IN_FLIGHT; }
        any = true;
        continue;
      }
      if (S.stage == Slot::EVAL_IN_FLIGHT) { RC(finish(S)); S.stage = Slot::IDLE; any = true; }
      if (S.stage == Slot::IDLE && more()) { RC(begin_batch(S)); RC(select_launch(S)); S.stage = Slot::STEP_IN_FLIGHT; any = true; }
    }
    if (!any) break;
  }
  if (timing) {
    fprintf(stderr, "oakgpu_search timing (ms, %d threads): select %.1f  gpu-step %.1f  process %.1f  eval %.1f  backprop %.1f\n", W, t_sel / 1e3, t_gpu / 1e3, t_proc / 1e3, t_eval / 1e3, t_back / 1e3);
    for (int q = 0; q < 4; ++q) fprintf(stderr, "  depth %d%s: %llu lane-steps, select %.2f ms, process %.2f ms\n", q, q == 3 ? "+" : "", (unsigned long long)n_d[q], t_sel_d[q] / 1e3, t_proc_d[q] / 1e3);
  }
  out->iterations = base_iterations + done;
  { // MCTS::Search::process_output (mcts.h:620-659) over the ACCUMULATED matrices: empirical_value = sum of the value matrix /
    // iterations, empirical strategies = row / column visit sums / iterations (0 iterations: 0 / 0, as the reference leaves
    // it), and the empirical root matrix x 256 as integers, solved exactly
    double tv = 0;
    int32_t M[81];
    for (int i = 0; i < m; ++i)
      for (int j = 0; j < n; ++j) {
        tv += out->value_matrix[i * 9 + j];
        uint64_t v = out->visit_matrix[i * 9 + j];
        out->p1_empirical[i] += (double)v;
        out->p2_empirical[j] += (double)v;
        v += !v;
        M[i * n + j] = (int32_t)(out->value_matrix[i * 9 + j] / (double)v * 256.0);
      }
    out->empirical_value = tv / (double)out->iterations;
    for (int i = 0; i < m; ++i) out->p1_empirical[i] /= (double)(float)out->iterations;
    for (int j = 0; j < n; ++j) out->p2_empirical[j] /= (double)(float)out->iterations;
    double nv = 0;
    if (oak_nash::solve(M, m, n, out->p1_nash, out->p2_nash, &nv)) out->nash_value = nv / 256.0;
  }
  (void)total_value;
  out->nodes = tree.size();
  out->total_depth = total_depth;
  out->duration_us = prev.duration_us + us(t_start, now()); // output.duration += ... (mcts.h:246-247): the iteration loop only
  return 0;
}

int oakgpu_search(oakgpu_ctx *ctx, oakgpu_net *net, const uint8_t *battle, const uint8_t *durations, uint8_t result,
                  const oakgpu_search_params *prm, oakgpu_search_output *out) {
  if (prm && prm->iterations == 0 && prm->duration_us == 0) return oakgpu_fail_msg("oakgpu_search: give an iteration or a time budget");
  return oakgpu_search_heap(ctx, net, nullptr, battle, durations, result, prm, nullptr, out);
}

} // extern "C"


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

void oakgpu_set_thread_search_threads(int threads) { tl_search_threads = threads > 0 ? threads : 0; }

// Cores this process may really use: the affinity mask's count, capped by the cgroup's CPU quota (a GPU box shows 256 CPUs in the mask
// and grants 16) -- cgroup v2 cpu.max, else v1 cfs_quota_us / cfs_period_us; OAKGPU_SEARCH_CORES overrides both.
unsigned oakgpu_usable_cores() {
  if (const char *e = getenv("OAKGPU_SEARCH_CORES")) { const int v = atoi(e); if (v > 0) return (unsigned)v; }
  unsigned hc = std::thread::hardware_concurrency();
  cpu_set_t set;
  if (sched_getaffinity(0, sizeof set, &set) == 0) hc = (unsigned)CPU_COUNT(&set);
  double quota = 0;
  if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
    char a[64] = {0};
    double period = 0;
    if (fscanf(f, "%63s %lf", a, &period) == 2 && strcmp(a, "max") != 0 && period > 0) quota = atof(a) / period;
    fclose(f);
  } else {
    double q = -1, per = 0;
    if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(g, "%lf", &q) != 1) q = -1; fclose(g); }
    if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(g, "%lf", &per) != 1) per = 0; fclose(g); }
    if (q > 0 && per > 0) quota = q / per;
  }
  if (quota >= 1 && quota < hc) hc = (unsigned)(quota + 0.5);
  return hc ? hc : 1;
}

// Several independent searches at once: one tree per root (the positions of n self-play games, the roots of a root-parallel search),
// each on ITS OWN context (stream, batch slots) and its own host threads, all on one GPU.  A single search leaves the card mostly
// idle -- per 2^18 iterations ~25 ms of host tree walk against ~6 ms of GPU work -- and eight host threads do not speed one tree
// up eightfold; n trees walked by n x (cores / n) threads keep both sides busy.  Every search is exactly the search
// oakgpu_search_heap would run with the same arguments (the tree walk does not depend on the thread count): same outputs, same
// heaps.  The reference's analogue is its worker pool -- N threads, one search each, never waiting for each other
// (cpp/src/generate.cc:527-536).  threads_per_search: 0 = the usable cores shared evenly (1, 2, 4 or 8 each).
extern "C" int oakgpu_search_many(oakgpu_ctx *const *ctxs, oakgpu_net *net, oakgpu_heap *const *heaps, const uint8_t *battles, const uint8_t *durations,
                                  const uint8_t *results, const oakgpu_search_params *params, uint32_t n, int threads_per_search,
                                  oakgpu_search_output *outs) {
  if (n == 0) return 0;
  if (!ctxs || !battles || !durations || !results || !params || !outs) return oakgpu_fail_msg("oakgpu_search_many: null argument");
  for (uint32_t i = 0; i < n; ++i) {
    if (!ctxs[i]) return oakgpu_fail_msg("oakgpu_search_many: null context");
    for (uint32_t k = 0; k < i; ++k)
      if (ctxs[k] == ctxs[i] || (heaps && heaps[i] && heaps[k] == heaps[i])) return oakgpu_fail_msg("oakgpu_search_many: every search needs a context (and heap) of its own");
  }
  int W = threads_per_search;
  if (W <= 0) W = (int)(oakgpu_usable_cores() / n);
  W = W >= 16 ? 16 : W >= 8 ? 8 : W >= 4 ? 4 : W >= 2 ? 2 : 1;
  std::vector<int> rc(n, 0);
  std::vector<std::string> err(n);
  std::vector<std::thread> th;
  th.reserve(n);
  static const bool serial = getenv("OAKGPU_SEARCH_MANY_SERIAL") != nullptr; // (diagnostic: the same threads, one after the other)
  for (uint32_t i = 0; i < n; ++i) {
    if (serial && !th.empty()) th.back().join();
    th.emplace_back([&, i] {
      tl_search_threads = W;
      rc[i] = oakgpu_search_heap(ctxs[i], net, heaps ? heaps[i] : nullptr, battles + (size_t)i * 384, durations + (size_t)i * 8, results[i], params + i, nullptr, outs + i);
      if (rc[i]) err[i] = oakgpu_last_error(); // (the error text is per thread: carry it to the caller's)
    });
  }
  for (auto &t : th) if (t.joinable()) t.join();
  for (uint32_t i = 0; i < n; ++i)
    if (rc[i]) return oakgpu_fail_msg(("oakgpu_search_many: search " + std::to_string(i) + ": " + err[i]).c_str());
  return 0;
}

extern "C" int oakgpu_search_agent_heap(oakgpu_ctx *ctx, oakgpu_heap *heap, const uint8_t *battle, const uint8_t *durations, uint8_t result,
                                        const oakgpu_agent *agent, uint32_t batch, uint64_t seed, const oakgpu_search_output *previous,
                                        oakgpu_search_output *out) {
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
  // ("0" is a legal budget: no iteration runs, the output carries the root's value / logits / priors -- what the
  // reference's cpp_inference asks for, pyoak.cc:331-392)
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
  P.exp3_alpha = -1.0f; // absent third field: the default 0.05 (search.cc:268-270); an explicit value, 0 included, goes through unchanged
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
  return oakgpu_search_heap(ctx, net, heap, battle, durations, result, &P, previous, out);
}

extern "C" int oakgpu_search_agent(oakgpu_ctx *ctx, const uint8_t *battle, const uint8_t *durations, uint8_t result, const oakgpu_agent *agent,
                                   uint32_t batch, uint64_t seed, oakgpu_search_output *out) {
  return oakgpu_search_agent_heap(ctx, nullptr, battle, durations, result, agent, batch, seed, nullptr, out);
}

extern "C" void oakgpu_agent_networks_clear(oakgpu_ctx *ctx) { // drops the networks oakgpu_search_agent loaded on this context's device
  if (!ctx) return;
  std::lock_guard<std::mutex> lock(g_net_mu);
  for (auto it = g_nets.begin(); it != g_nets.end();)
    if (it->first.first == oakgpu_ctx_device(ctx)) { oakgpu_net_free(ctx, it->second); it = g_nets.erase(it); } else ++it;
}
