// Block order of the one-launch steps of the main_batched chain (chain_step.hip: equally shaped batches;
// chain_ragged.hip: differently sized images): the grid holds the blocks of five stages -- F, V, P, L, R -- of five
// different batches.  F and V blocks first (the longest dependent chains start at once), then P, L and R blocks in the
// order build_interleave lays out, in units of 8 consecutive blocks ("octets") so that block % 8 keeps naming the XCD for
// the resample's XCD-aware order.
#pragma once
#include "common.hpp"

#include <cstring>

namespace attwarp {

constexpr int CHAIN_Q = 32;                 // octets per period of the P / L / R interleave
constexpr int CHAIN_ORDER_DEFAULT = 2;      // see build_interleave

struct ChainOrder {
  int nF, nV;                               // blocks of F and V
  int nF8, nV8;                             // octets (8 blocks) of the F and V ranges
  int nPfirst8;                             // octets of P laid out as one range right behind them (0: P is interleaved)
  int nPmid8;                               // octets of P laid out as one range between the two interleaved sections
  int nP, nL, nR;                           // blocks of P, L, R
  // interleave: `periods` periods of CHAIN_Q octets, each holding q[t] octets of type t (0 = P, 1 = L, 2 = R) at the
  // positions type[] says (rank[] = how many octets of the same type precede inside the period); then the leftovers of
  // P, of L and of R one after the other
  // two such sections one after the other (the second starts where the first stopped in every kind); sec[1] may be empty
  struct Section { int periods, q[3], base8[3]; unsigned char type[CHAIN_Q], rank[CHAIN_Q]; } sec[2];
  int left8[3], leftbase8[3];
};

enum { CHAIN_F = 0, CHAIN_V = 1, CHAIN_P = 2, CHAIN_L = 3, CHAIN_R = 4, CHAIN_PAD = -1 };

#ifdef __HIPCC__
// block `blk` of the grid -> its kind and its index j inside that kind's range (CHAIN_PAD: a padding block)
__device__ __forceinline__ int chain_order_decode(const ChainOrder& a, int blk, int& j) {
  const int l8 = blk & 7;
  int oct = blk >> 3;
  if (oct < a.nF8) { j = blk; return blk < a.nF ? CHAIN_F : CHAIN_PAD; }
  oct -= a.nF8;
  if (oct < a.nV8) { j = oct * 8 + l8; return j < a.nV ? CHAIN_V : CHAIN_PAD; }
  oct -= a.nV8;
  int t, idx8;
  const int interA = a.sec[0].periods * CHAIN_Q, interB = a.sec[1].periods * CHAIN_Q;
  if (oct < a.nPfirst8) {
    t = 0; idx8 = oct;
  } else if ((oct -= a.nPfirst8) < interA) {
    const int per = oct / CHAIN_Q, pos = oct - per * CHAIN_Q;
    t = a.sec[0].type[pos];
    idx8 = a.sec[0].base8[t] + per * a.sec[0].q[t] + a.sec[0].rank[pos];
  } else if ((oct -= interA) < a.nPmid8) {
    t = 0; idx8 = a.sec[1].base8[0] + oct;          // (sec[1].base8[0] = where section A stopped in P)
  } else if ((oct -= a.nPmid8) < interB) {
    const int per = oct / CHAIN_Q, pos = oct - per * CHAIN_Q;
    t = a.sec[1].type[pos];
    idx8 = a.sec[1].base8[t] + per * a.sec[1].q[t] + a.sec[1].rank[pos] + (t == 0 ? a.nPmid8 : 0);
  } else {
    int r = oct - interB;
    if (r < a.left8[0]) { t = 0; idx8 = a.leftbase8[0] + r; }
    else if ((r -= a.left8[0]) < a.left8[1]) { t = 1; idx8 = a.leftbase8[1] + r; }
    else { t = 2; idx8 = a.leftbase8[2] + (r - a.left8[1]); }
  }
  j = idx8 * 8 + l8;
  if (t == 0) return j < a.nP ? CHAIN_P : CHAIN_PAD;
  if (t == 1) return j < a.nL ? CHAIN_L : CHAIN_PAD;
  return j < a.nR ? CHAIN_R : CHAIN_PAD;
}
#endif

// proportional interleave of three block kinds inside a period of CHAIN_Q octets (largest-remainder rounding, then an
// even spread: position i goes to the kind that is furthest behind its share); fills one section with as many whole
// periods as n8[] allows and returns what it consumed
inline void fill_section(ChainOrder::Section& S, const int n8[3], const int base8[3], int used8[3]) {
  memset(&S, 0, sizeof(S));
  for (int t = 0; t < 3; ++t) { S.base8[t] = base8[t]; used8[t] = 0; }
  const long long tot = (long long)n8[0] + n8[1] + n8[2];
  if (tot <= 0) return;
  int q[3] = {0, 0, 0}, used = 0;
  double frac[3];
  for (int t = 0; t < 3; ++t) {
    const double share = (double)CHAIN_Q * n8[t] / (double)tot;
    q[t] = (int)share;
    if (n8[t] > 0 && q[t] == 0) q[t] = 1;
    frac[t] = share - (int)share;
    used += q[t];
  }
  while (used < CHAIN_Q) {
    int best = 0;
    for (int t = 1; t < 3; ++t) if (frac[t] > frac[best]) best = t;
    ++q[best]; frac[best] = -1.0; ++used;
  }
  while (used > CHAIN_Q) {
    int big = 0;
    for (int t = 1; t < 3; ++t) if (q[t] > q[big]) big = t;
    --q[big]; --used;
  }
  int periods = 0x7fffffff;
  for (int t = 0; t < 3; ++t) if (q[t] > 0) periods = std::min(periods, n8[t] / q[t]);
  S.periods = periods == 0x7fffffff ? 0 : periods;
  int placed[3] = {0, 0, 0};
  for (int i = 0; i < CHAIN_Q; ++i) {
    int best = -1;
    double lag = -1e30;
    for (int t = 0; t < 3; ++t) {
      if (placed[t] >= q[t]) continue;
      const double l = (double)(i + 1) * q[t] / CHAIN_Q - placed[t];
      if (l > lag) { lag = l; best = t; }
    }
    S.type[i] = (unsigned char)best;
    S.rank[i] = (unsigned char)placed[best];
    ++placed[best];
  }
  for (int t = 0; t < 3; ++t) { S.q[t] = q[t]; used8[t] = S.periods * q[t]; }
}
// order: 0 = P, L and R interleaved; 1 = P, L, R one after the other; 2 = all of P first, then L and R interleaved;
// 3 = P and R interleaved, then all of L; 4 = P and L interleaved, then all of R (measured, not the default);
// 10..99 = P spread over the first `order` per cent of L and R (interleaved with them), then the rest of L and R
// interleaved: the long marginals blocks all start early enough not to be the launch's tail, and the resample's memory
// traffic runs beside their arithmetic from the start
inline void build_interleave(ChainOrder& a, int order) {
  int n8[3] = {(a.nP + 7) / 8, (a.nL + 7) / 8, (a.nR + 7) / 8};
  const int zero3[3] = {0, 0, 0};
  int usedA[3] = {0, 0, 0}, usedB[3] = {0, 0, 0};
  a.nPfirst8 = a.nPmid8 = 0;
  memset(a.sec, 0, sizeof(a.sec));
  int baseB[3] = {0, 0, 0};
  if (order >= 10 && order <= 99) {
    const int nA[3] = {n8[0], (int)((long long)n8[1] * order / 100), (int)((long long)n8[2] * order / 100)};
    fill_section(a.sec[0], nA, zero3, usedA);
    a.nPmid8 = n8[0] - usedA[0];                               // what the whole periods left of P: one range behind section A
    for (int t = 0; t < 3; ++t) baseB[t] = usedA[t];
    const int nB[3] = {0, n8[1] - usedA[1], n8[2] - usedA[2]};
    fill_section(a.sec[1], nB, baseB, usedB);
    for (int t = 0; t < 3; ++t) { a.leftbase8[t] = usedA[t] + usedB[t] + (t == 0 ? a.nPmid8 : 0); a.left8[t] = n8[t] - a.leftbase8[t]; }
    return;
  }
  if (order == 1) {                                            // three ranges
    for (int t = 0; t < 3; ++t) { a.leftbase8[t] = 0; a.left8[t] = n8[t]; }
    return;
  }
  int nA[3] = {n8[0], n8[1], n8[2]};
  int base[3] = {0, 0, 0};
  if (order == 2) { a.nPfirst8 = n8[0]; nA[0] = 0; base[0] = n8[0]; }
  const int excl = order == 3 ? 1 : order == 4 ? 2 : -1;      // that kind follows the interleaved part as one range
  if (excl >= 0) nA[excl] = 0;
  fill_section(a.sec[0], nA, base, usedA);
  for (int t = 0; t < 3; ++t) {
    a.leftbase8[t] = base[t] + usedA[t];
    a.left8[t] = (t == 0 && order == 2) ? 0 : n8[t] - usedA[t];
  }
}

// octets of the whole grid once build_interleave has run
inline long long chain_order_octets(const ChainOrder& a) {
  return (long long)a.nF8 + a.nV8 + a.nPfirst8 + (long long)a.sec[0].periods * CHAIN_Q + a.nPmid8 +
         (long long)a.sec[1].periods * CHAIN_Q + a.left8[0] + a.left8[1] + a.left8[2];
}

}  // namespace attwarp
