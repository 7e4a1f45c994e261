// oracle/ref_bandit_dump.cc -- TEST INFRASTRUCTURE (not product code).
//
// Compiles against the reference's own bandit headers where they lie under /root/reference
// (cpp/include/search/bandit/{ucb,pucb,ucb1,exp3,pexp3}.h, search/joint.h, search/util/*.h, util/random.h: they
// depend only on the C++ standard library, so no stand-in header is written) and dumps select / update traces of
// every bandit as JSON on stdout -> tests/golden/bandit_traces.json.  A trace = one player's Bandit driven for
// `steps` rounds by the reference's mt19937 device: select, then update with a value from a fixed pseudo-random
// sequence.  Recorded per round: the uniform draw device.sample_pdf consumed (Exp3 / PExp3, k > 1 only), the selected
// index, the selection probability; at the end the bandit's statistics.  The product's bandit arithmetic
// (oak_amd/csrc/bandit.hpp) is replayed on the same inputs by tests/test_search_host.py and must agree bit for bit.
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <iostream>
#include <limits>
#include <vector>

#include <util/random.h>
#include <search/bandit/exp3.h>
#include <search/bandit/pexp3.h>
#include <search/bandit/pucb.h>
#include <search/bandit/ucb.h>
#include <search/bandit/ucb1.h>

static float value_of(uint32_t &s) { // fixed outcome sequence in [0, 1], exactly representable steps of 1/65535
  s = s * 1664525u + 1013904223u;
  return (float)((s >> 8) & 0xFFFF) / 65535.0f;
}
static void pf(float x) { // a float as JSON: shortest round-trip decimal, infinities as strings
  if (std::isinf(x)) std::printf(x < 0 ? "\"-inf\"" : "\"inf\"");
  else std::printf("%.9g", (double)x);
}
template <class A> static void parr(const char *name, const A &a, int n) {
  std::printf("\"%s\": [", name);
  for (int i = 0; i < n; ++i) { if (i) std::printf(","); pf((float)a[i]); }
  std::printf("]");
}
static bool first_trace = true;
static void open_trace(const char *kind, int k, float c, float alpha, uint32_t seed, int steps) {
  std::printf("%s\n{\"kind\": \"%s\", \"k\": %d, \"c\": %.9g, \"alpha\": %.9g, \"seed\": %u, \"steps\": %d, ", first_trace ? "" : ",", kind, k,
              (double)c, (double)alpha, seed, steps);
  first_trace = false;
}

template <class Bandit, class Params>
static void counting_trace(const char *kind, int k, Params params, uint32_t seed, int steps, const float *logits) {
  Bandit b{};
  b.init(k);
  if constexpr (requires { b.softmax_logits(params, logits); }) b.softmax_logits(params, logits);
  mt19937 device{seed};
  uint32_t vs = seed * 2654435761u + 12345u;
  std::vector<int> idx;
  std::vector<float> vals;
  for (int t = 0; t < steps; ++t) {
    typename Bandit::Outcome o{};
    b.select(device, params, o);
    o.value = value_of(vs);
    b.update(o);
    idx.push_back(o.index);
    vals.push_back(o.value);
  }
  open_trace(kind, k, params.c, 0.0f, seed, steps);
  if (logits) { parr("logits", logits, k); std::printf(", "); }
  parr("values", vals, steps);
  std::printf(", \"index\": [");
  for (int t = 0; t < steps; ++t) std::printf("%s%d", t ? "," : "", idx[t]);
  std::printf("], ");
  parr("scores", b.scores, k);
  std::printf(", \"visits\": [");
  for (int i = 0; i < k; ++i) std::printf("%s%u", i ? "," : "", (unsigned)(uint32_t)b.visits[i]);
  std::printf("]");
  if constexpr (requires { b.priors; }) { std::printf(", "); parr("priors", b.priors, k); }
  std::printf("}");
}

template <class Bandit>
static void exp3_trace(const char *kind, int k, float gamma, float alpha, uint32_t seed, int steps, const float *logits) {
  typename Bandit::Params params{.gamma = gamma, .one_minus_gamma = (1 - gamma), .alpha = alpha, .one_minus_alpha = (1 - alpha)}; // search.cc:268-286
  Bandit b{};
  b.init(k);
  if constexpr (requires { b.softmax_logits(params, logits); }) b.softmax_logits(params, logits);
  mt19937 device{seed}, twin{seed}; // the twin replays the device to log the uniform draws sample_pdf consumes
  uint32_t vs = seed * 2654435761u + 12345u;
  std::vector<int> idx;
  std::vector<float> vals, probs;
  std::vector<double> us;
  for (int t = 0; t < steps; ++t) {
    typename Bandit::Outcome o{};
    b.select(device, params, o);
    if (k > 1) us.push_back(twin.uniform());
    o.value = value_of(vs);
    b.update(o);
    idx.push_back(o.index);
    vals.push_back(o.value);
    probs.push_back(o.prob);
  }
  open_trace(kind, k, gamma, alpha, seed, steps);
  if (logits) { parr("logits", logits, k); std::printf(", "); }
  parr("values", vals, steps);
  std::printf(", \"uniforms\": [");
  for (size_t t = 0; t < us.size(); ++t) std::printf("%s%.17g", t ? "," : "", us[t]);
  std::printf("], \"index\": [");
  for (int t = 0; t < steps; ++t) std::printf("%s%d", t ? "," : "", idx[t]);
  std::printf("], ");
  parr("prob", probs, steps);
  std::printf(", ");
  parr("gains", b.gains, 9);
  std::printf("}");
}

int main() {
  const float logits4[9] = {0.25f, -1.5f, 0.75f, 0.0f}, logits9[9] = {0.1f, -0.2f, 0.3f, -0.4f, 0.5f, -0.6f, 0.7f, -0.8f, 0.9f};
  std::printf("{\"source\": \"oracle/ref_bandit_dump.cc over the reference's search/bandit/*.h\", \"traces\": [");
  const int ks[] = {1, 2, 4, 9};
  for (int k : ks) {
    const float *lg = k == 9 ? logits9 : logits4;
    for (uint32_t seed : {7u, 1111111u}) {
      const int steps = k == 9 ? 240 : 120;
      counting_trace<UCB::Bandit>("ucb", k, UCB::Bandit::Params{.c = 1.0f}, seed, steps, nullptr);
      counting_trace<UCB::Bandit>("ucb", k, UCB::Bandit::Params{.c = 0.35f}, seed + 1, steps, nullptr);
      counting_trace<UCB1::Bandit>("ucb1", k, UCB1::Bandit::Params{.c = 2.0f}, seed, steps, nullptr);
      counting_trace<PUCB::Bandit>("pucb", k, PUCB::Bandit::Params{.c = 1.5f}, seed, steps, lg);
      exp3_trace<Exp3::Bandit>("exp3", k, 1.0f, 0.1f, seed, steps, nullptr);   // search-test.cc:28 "exp3-1.0-0.1"
      exp3_trace<Exp3::Bandit>("exp3", k, 0.3f, 0.05f, seed + 2, steps, nullptr);
      exp3_trace<PExp3::Bandit>("pexp3", k, 0.5f, 0.05f, seed, steps, lg);
    }
  }
  std::printf("\n]}\n");
  return 0;
}
