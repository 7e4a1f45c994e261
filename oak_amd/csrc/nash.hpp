// oak_amd/csrc/nash.hpp -- exact Nash equilibrium of a <= 9 x 9 zero-sum matrix game with integer payoffs (host code).
//
// Replaces LRSNash::solve_fast (lab-oak/lrsnash over GMP: absent from the reference checkout) at its three call sites:
// MCTS::Search::process_output (cpp/include/search/mcts.h:620-659), the MatrixUCB root solve (mcts.h:532-543) and
// pyoak's solve_matrix (cpp/src/pyoak.cc:394-426).  Like the reference's, the solve uses NO floating point: it is the
// simplex method with Bland's rule in integer ("fraction-free") pivoting -- the tableau holds integers over one common
// denominator, every division is exact -- on 512-bit integers (tableau entries are minors of the payoff matrix: below
// 2^203 for |payoff| <= 2^20, so products fit).  Only the final strategies / value are rounded to double.
// The reference holds no test for its solver; this one is checked by LP optimality (zero best-response gap in exact
// arithmetic, tests/test_search_host.py).
#pragma once
#include <stdint.h>

#include <cmath>

namespace oak_nash {

struct Wide { // 512-bit two's complement
  static constexpr int L = 8;
  uint64_t w[L];
  Wide() { for (int i = 0; i < L; ++i) w[i] = 0; }
  Wide(int64_t v) { w[0] = (uint64_t)v; for (int i = 1; i < L; ++i) w[i] = v < 0 ? ~0ull : 0ull; }
  bool neg() const { return (w[L - 1] >> 63) != 0; }
  bool zero() const { for (int i = 0; i < L; ++i) if (w[i]) return false; return true; }
  Wide operator-() const {
    Wide r;
    unsigned __int128 c = 1;
    for (int i = 0; i < L; ++i) { c += (unsigned __int128)(~w[i]); r.w[i] = (uint64_t)c; c >>= 64; }
    return r;
  }
  Wide operator+(const Wide &o) const {
    Wide r;
    unsigned __int128 c = 0;
    for (int i = 0; i < L; ++i) { c += (unsigned __int128)w[i] + o.w[i]; r.w[i] = (uint64_t)c; c >>= 64; }
    return r;
  }
  Wide operator-(const Wide &o) const { return *this + (-o); }
  Wide operator*(const Wide &o) const { // truncating; callers stay far below 2^511
    Wide r;
    for (int i = 0; i < L; ++i) {
      unsigned __int128 c = 0;
      for (int j = 0; i + j < L; ++j) {
        c += (unsigned __int128)w[i] * o.w[j] + r.w[i + j];
        r.w[i + j] = (uint64_t)c;
        c >>= 64;
      }
    }
    return r;
  }
  // magnitude helpers (operands non-negative)
  static int ucmp(const Wide &a, const Wide &b) {
    for (int i = L - 1; i >= 0; --i) if (a.w[i] != b.w[i]) return a.w[i] < b.w[i] ? -1 : 1;
    return 0;
  }
  static int cmp(const Wide &a, const Wide &b) { // signed
    if (a.neg() != b.neg()) return a.neg() ? -1 : 1;
    return ucmp(a, b); // same sign: two's complement orders like unsigned
  }
  int top_bit() const { // index of the highest set bit of a non-negative value, -1 for zero
    for (int i = L - 1; i >= 0; --i) if (w[i]) return 64 * i + 63 - __builtin_clzll(w[i]);
    return -1;
  }
  Wide shl1() const { Wide r; uint64_t c = 0; for (int i = 0; i < L; ++i) { r.w[i] = (w[i] << 1) | c; c = w[i] >> 63; } return r; }
  bool bit(int k) const { return (w[k >> 6] >> (k & 63)) & 1; }
  // exact quotient a / b (b != 0, b divides a): shift-subtract long division on magnitudes
  static Wide divexact(const Wide &a, const Wide &b) {
    const bool sn = a.neg() != b.neg();
    const Wide ua = a.neg() ? -a : a, ub = b.neg() ? -b : b;
    Wide q, rem;
    for (int k = ua.top_bit(); k >= 0; --k) {
      rem = rem.shl1();
      if (ua.bit(k)) rem.w[0] |= 1;
      if (ucmp(rem, ub) >= 0) { rem = rem - ub; q.w[k >> 6] |= 1ull << (k & 63); }
    }
    return sn ? -q : q;
  }
  long double to_ld() const {
    const Wide u = neg() ? -*this : *this;
    long double r = 0;
    for (int i = L - 1; i >= 0; --i) r = r * 18446744073709551616.0L + (long double)u.w[i];
    return neg() ? -r : r;
  }
};

// Equilibrium of the zero-sum game in which the ROW player maximises payoffs[i * n + j] (integers, |.| <= 2^20).
// p1[m], p2[n]: equilibrium strategies; *value: the game value in payoff units.  Returns false on bad arguments.
inline bool solve(const int32_t *payoffs, int m, int n, double *p1, double *p2, double *value) {
  if (!payoffs || m < 1 || n < 1 || m > 9 || n > 9) return false;
  int64_t lo = payoffs[0];
  for (int i = 0; i < m * n; ++i) {
    if (payoffs[i] > (1 << 20) || payoffs[i] < -(1 << 20)) return false;
    lo = payoffs[i] < lo ? payoffs[i] : lo;
  }
  const int64_t shift = 1 - lo; // every entry >= 1
  // column player: maximise sum z  s.t.  B z <= 1, z >= 0;  p2 = z / sum z, value + shift = 1 / sum z; the duals
  // (reduced costs of the slacks) give p1.  Tableau [B | I | 1] and objective row over the common denominator D.
  const int W = n + m + 1;
  Wide T[9][19], z[19], D(1);
  int basis[9];
  for (int i = 0; i < m; ++i) {
    for (int j = 0; j < n; ++j) T[i][j] = Wide(payoffs[i * n + j] + shift);
    for (int k = 0; k < m; ++k) T[i][n + k] = Wide(i == k ? 1 : 0);
    T[i][W - 1] = Wide(1);
    basis[i] = n + i;
  }
  for (int j = 0; j < W; ++j) z[j] = Wide(j < n ? -1 : 0);
  for (int iter = 0; iter < 100000; ++iter) {
    int col = -1;
    for (int j = 0; j < n + m; ++j) if (z[j].neg()) { col = j; break; } // Bland: lowest index with negative reduced cost
    if (col < 0) break;
    int row = -1;
    for (int i = 0; i < m; ++i) {
      if (T[i][col].neg() || T[i][col].zero()) continue;
      if (row < 0) { row = i; continue; }
      // ratio_i < ratio_row  <=>  T[i][rhs] * T[row][col] < T[row][rhs] * T[i][col]   (both pivots positive)
      const int c = Wide::cmp(T[i][W - 1] * T[row][col], T[row][W - 1] * T[i][col]);
      if (c < 0 || (c == 0 && basis[i] < basis[row])) row = i;
    }
    if (row < 0) return false; // unbounded: impossible for entries >= 1
    const Wide piv = T[row][col];
    for (int i = 0; i < m; ++i) {
      if (i == row) continue;
      const Wide f = T[i][col];
      for (int j = 0; j < W; ++j) T[i][j] = Wide::divexact(T[i][j] * piv - f * T[row][j], D);
    }
    const Wide f = z[col];
    for (int j = 0; j < W; ++j) z[j] = Wide::divexact(z[j] * piv - f * T[row][j], D);
    D = piv;
    basis[row] = col;
  }
  const long double tot = z[W - 1].to_ld(); // sum z = tot / D  (> 0)
  if (!(tot > 0)) return false;
  for (int j = 0; j < n; ++j) p2[j] = 0.0;
  for (int i = 0; i < m; ++i) if (basis[i] < n) p2[basis[i]] = (double)(T[i][W - 1].to_ld() / tot);
  for (int i = 0; i < m; ++i) p1[i] = (double)(z[n + i].to_ld() / tot);
  *value = (double)(D.to_ld() / tot - (long double)shift);
  return true;
}

} // namespace oak_nash
