// oak_amd/csrc/bandit.hpp -- one player's bandit at a tree node, host side of the batched-leaf tree search.
//
// The arithmetic of the reference's five bandits, statement for statement (same float / double / integer types at every
// step, so that select / update traces agree BIT FOR BIT with traces dumped from the reference's own headers:
// tests/golden/bandit_traces.json):
//   UCB::Bandit    cpp/include/search/bandit/ucb.h:17-66     PUCB::Bandit   pucb.h:17-75
//   UCB1::Bandit   ucb1.h:16-67                              Exp3::Bandit   exp3.h:17-79      PExp3::Bandit  pexp3.h:17-84
//   softmax        search/util/softmax.h:5-28                sample_pdf     util/random.h:40-49
// One struct serves all five (`scores` doubles as Exp3's gains); the kind comes with the parameters.  The only
// deliberate difference is WHEN the visit of the counting bandits is booked: at selection time (`visit`, the virtual
// loss that keeps a batch of simultaneous descents apart) instead of inside update(); with one descent at a time
// select -> visit -> update is exactly the reference's select -> update.
#pragma once
#include <stdint.h>

#include <cmath>
#if !defined(__HIP_DEVICE_COMPILE__) && defined(__SSE2__)
#include <emmintrin.h>
#define OAK_BANDIT_SSE2 1
#endif

namespace oak_search {

enum BanditKind { B_UCB = 0, B_PUCB = 1, B_UCB1 = 2, B_EXP3 = 3, B_PEXP3 = 4 };
struct BanditParams { int kind; float c; float alpha; }; // c: UCB c / Exp3 gamma; alpha: Exp3 uniform mixing (search.cc:268-286)

struct Bandit {
  float scores[9];
  float priors[9];
  uint32_t visits[9];
  uint8_t k = 0;
  void init(uint8_t kk, int kind) {
    k = kk;
    for (int i = 0; i < 9; ++i) {
      priors[i] = kk ? 1.0f / kk : 0.0f;
      if (kind == B_UCB1) { scores[i] = 0.0f; visits[i] = 0; }
      else if (kind >= B_EXP3) { scores[i] = i < kk ? 0.0f : -INFINITY; visits[i] = 0; }
      else { scores[i] = 0.5f; visits[i] = 1; }
    }
  }
  bool is_init() const { return k != 0; }
  // network logits -> priors (PUCB::softmax_logits = softmax.h:5-15) or initial gains (PExp3::softmax_logits: logit / eta)
  void set_logits(const BanditParams &P, const float *logits) {
    if (P.kind == B_PUCB) {
      float sum = 0;
      for (int i = 0; i < k; ++i) { const float y = std::exp(logits[i]); priors[i] = y; sum += y; }
      for (int i = 0; i < k; ++i) priors[i] /= sum;
    } else if (P.kind == B_PEXP3) {
      const float eta = P.c / k;
      for (int i = 0; i < k; ++i) scores[i] = logits[i] / eta;
    }
  }
  // `uniform` is called once per sampled selection (Exp3 / PExp3 with k > 1): the device's uniform() in (0, 1)
  template <class U> uint8_t select(const BanditParams &P, U &&uniform, float &prob) const {
    prob = 1.0f;
    if (k == 1) return 0;
    if (P.kind >= B_EXP3) {
      const float eta = P.c / k, delta = P.alpha / k, one_minus_alpha = 1 - P.alpha;
      float policy[9], sum = 0;
      for (int i = 0; i < 9; ++i) { const float y = std::exp(scores[i] * eta); policy[i] = y; sum += y; }
      for (int i = 0; i < 9; ++i) policy[i] /= sum;
      for (int i = 0; i < 9; ++i) policy[i] = one_minus_alpha * policy[i] + delta;
      double u = uniform();
      uint32_t idx = 0;
      for (uint32_t i = 0; i < 9; ++i) { u -= (double)policy[i]; if (u <= 0) { idx = i; break; } }
      const uint8_t sel = (uint8_t)idx < (uint8_t)(k - 1) ? (uint8_t)idx : (uint8_t)(k - 1);
      prob = policy[sel];
      return sel;
    }
    uint64_t N = 0;
    uint8_t idx = 0; // the reference leaves outcome.index at its initial 0 when no arm beats max = 0
    float best = 0;
    if (P.kind == B_UCB1) {
      float q[9] = {};
      for (int i = k - 1; i >= 0; --i) {
        if (visits[i] == 0) return (uint8_t)i;
        q[i] = scores[i] / visits[i];
        N += visits[i];
      }
      const float lnN = (float)std::log((double)N);
      for (int i = 0; i < k; ++i) {
        const float e = std::sqrt(P.c * lnN / visits[i]);
        const float a = e + q[i];
        if (a > best) { best = a; idx = (uint8_t)i; }
      }
      return idx;
    }
    for (int i = 0; i < k; ++i) N += visits[i];
    const float sqrtN = (float)std::sqrt((double)N);
#ifdef OAK_BANDIT_SSE2
    // the same IEEE operations, four arms per instruction (the root of a batched search runs this tens of thousands of
    // times in a row), and a branch-free argmax: the scalar scan `if (a > best)` from best = 0 picks the FIRST arm that attains
    // the maximum provided it is positive, else arm 0 -- so: lanes past k forced to 0, the maximum by two max instructions and
    // a shuffle tree, the first index that equals it from the compare masks
    const __m128 sq = _mm_set1_ps(sqrtN), cc = _mm_set1_ps(P.c), ucb_e = _mm_set1_ps(P.c * sqrtN / k);
    __m128 a[2];
    for (int b = 0; b < 2; ++b) {
      const __m128 e = P.kind == B_PUCB ? _mm_mul_ps(_mm_mul_ps(cc, _mm_loadu_ps(priors + 4 * b)), sq) : ucb_e;
      const __m128 v = _mm_cvtepi32_ps(_mm_loadu_si128((const __m128i *)(visits + 4 * b)));
      const __m128 q = _mm_div_ps(_mm_add_ps(e, _mm_loadu_ps(scores + 4 * b)), v);
      const __m128i lane = _mm_add_epi32(_mm_set_epi32(3, 2, 1, 0), _mm_set1_epi32(4 * b));
      a[b] = _mm_and_ps(q, _mm_castsi128_ps(_mm_cmplt_epi32(lane, _mm_set1_epi32(k)))); // arms >= k: +0.0f, never a strict maximum
    }
    float a8 = 0.0f;
    if (k == 9) { const float e = P.kind == B_PUCB ? P.c * priors[8] * sqrtN : P.c * sqrtN / k; a8 = (e + scores[8]) / visits[8]; }
    __m128 m = _mm_max_ps(a[0], a[1]);
    m = _mm_max_ps(m, _mm_shuffle_ps(m, m, _MM_SHUFFLE(1, 0, 3, 2)));
    m = _mm_max_ps(m, _mm_shuffle_ps(m, m, _MM_SHUFFLE(2, 3, 0, 1)));
    m = _mm_max_ss(m, _mm_set_ss(a8));
    const float mx = _mm_cvtss_f32(m);
    if (!(mx > 0.0f)) return 0; // no arm beats the initial best = 0 (also: a NaN maximum)
    const __m128 mm = _mm_set1_ps(mx);
    const unsigned bits = (unsigned)_mm_movemask_ps(_mm_cmpeq_ps(a[0], mm)) | ((unsigned)_mm_movemask_ps(_mm_cmpeq_ps(a[1], mm)) << 4) |
                          ((a8 == mx ? 1u : 0u) << 8);
    idx = (uint8_t)__builtin_ctz(bits | 0x200u);
    (void)best;
#else
    for (int i = 0; i < k; ++i) {
      const float e = P.kind == B_PUCB ? P.c * priors[i] * sqrtN : P.c * sqrtN / k;
      const float a = (e + scores[i]) / visits[i];
      if (a > best) { best = a; idx = (uint8_t)i; }
    }
#endif
    return idx;
  }
  // `count` consecutive select + visit rounds of ONE node (the root of a batched search: every lane of the batch is there and
  // each sees the virtual losses of those before it).  Exactly the sequence select(); visit() repeated, for UCB / PUCB with the
  // visit counts, their sum and the scores kept in registers across the rounds (the round-to-round dependency is
  // convert -> divide -> max -> compare -> add instead of a trip through memory); other kinds take the plain loop.
  template <class U> void select_run(const BanditParams &P, uint32_t count, uint8_t *idx_out, float *prob_out, U &&uniform_at_round) {
#ifdef OAK_BANDIT_SSE2
    if ((P.kind == B_UCB || P.kind == B_PUCB) && k > 1) {
      uint64_t N = 0;
      for (int i = 0; i < k; ++i) N += visits[i];
      __m128i v0 = _mm_loadu_si128((const __m128i *)visits), v1 = _mm_loadu_si128((const __m128i *)(visits + 4));
      uint32_t v8 = visits[8];
      const __m128 s0 = _mm_loadu_ps(scores), s1 = _mm_loadu_ps(scores + 4), p0 = _mm_loadu_ps(priors), p1 = _mm_loadu_ps(priors + 4);
      const __m128 cc = _mm_set1_ps(P.c);
      const __m128i kk = _mm_set1_epi32(k), lane0 = _mm_set_epi32(3, 2, 1, 0), lane1 = _mm_set_epi32(7, 6, 5, 4);
      const __m128 live0 = _mm_castsi128_ps(_mm_cmplt_epi32(lane0, kk)), live1 = _mm_castsi128_ps(_mm_cmplt_epi32(lane1, kk));
      for (uint32_t r = 0; r < count; ++r, ++N) {
        const float sqrtN = (float)std::sqrt((double)N);
        const __m128 sq = _mm_set1_ps(sqrtN), ucb_e = _mm_set1_ps(P.c * sqrtN / k);
        const __m128 e0 = P.kind == B_PUCB ? _mm_mul_ps(_mm_mul_ps(cc, p0), sq) : ucb_e;
        const __m128 e1 = P.kind == B_PUCB ? _mm_mul_ps(_mm_mul_ps(cc, p1), sq) : ucb_e;
        const __m128 a0 = _mm_and_ps(_mm_div_ps(_mm_add_ps(e0, s0), _mm_cvtepi32_ps(v0)), live0);
        const __m128 a1 = _mm_and_ps(_mm_div_ps(_mm_add_ps(e1, s1), _mm_cvtepi32_ps(v1)), live1);
        float a8 = 0.0f;
        if (k == 9) { const float e = P.kind == B_PUCB ? P.c * priors[8] * sqrtN : P.c * sqrtN / k; a8 = (e + scores[8]) / v8; }
        __m128 m = _mm_max_ps(a0, a1);
        m = _mm_max_ps(m, _mm_shuffle_ps(m, m, _MM_SHUFFLE(1, 0, 3, 2)));
        m = _mm_max_ps(m, _mm_shuffle_ps(m, m, _MM_SHUFFLE(2, 3, 0, 1)));
        m = _mm_max_ss(m, _mm_set_ss(a8));
        const float mx = _mm_cvtss_f32(m);
        unsigned sel = 0;
        if (mx > 0.0f) {
          const __m128 mm = _mm_set1_ps(mx);
          const unsigned bits = (unsigned)_mm_movemask_ps(_mm_cmpeq_ps(a0, mm)) | ((unsigned)_mm_movemask_ps(_mm_cmpeq_ps(a1, mm)) << 4) |
                                ((a8 == mx ? 1u : 0u) << 8);
          sel = (unsigned)__builtin_ctz(bits | 0x200u);
        }
        idx_out[r] = (uint8_t)sel;
        if (prob_out) prob_out[r] = 1.0f;
        const __m128i hit = _mm_set1_epi32((int)sel); // the visit: + 1 in the selected arm's lane
        v0 = _mm_sub_epi32(v0, _mm_cmpeq_epi32(lane0, hit));
        v1 = _mm_sub_epi32(v1, _mm_cmpeq_epi32(lane1, hit));
        v8 += sel == 8;
      }
      _mm_storeu_si128((__m128i *)visits, v0);
      _mm_storeu_si128((__m128i *)(visits + 4), v1);
      visits[8] = v8;
      return;
    }
#endif
    for (uint32_t r = 0; r < count; ++r) {
      float pr;
      const uint8_t i = select(P, [&] { return uniform_at_round(r); }, pr);
      visit(P, i);
      idx_out[r] = i;
      if (prob_out) prob_out[r] = pr;
    }
  }
  // the visit of the counting bandits is booked at selection time (virtual loss), their score at back-up time
  void visit(const BanditParams &P, uint8_t i) { if (P.kind < B_EXP3) ++visits[i]; }
  void update(const BanditParams &P, uint8_t i, float value, float prob) {
    if (P.kind < B_EXP3) { scores[i] += value; return; }
    // (prob == 0: only with alpha = 0 and an underflowed softmax, where the reference divides by zero -- the smallest normal float instead)
    if ((scores[i] += (value - 0.5) / (prob > 0 ? prob : 1.17549435e-38f)) > 0) { // Exp3::update (the reference's 0.5 is a double literal): keep the largest gain at 0
      const float mx = scores[i];
      for (int q = 0; q < 9; ++q) scores[q] -= mx;
    }
  }
};

} // namespace oak_search
