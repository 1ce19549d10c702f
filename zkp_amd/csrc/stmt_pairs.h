// Which terms of a variable-time statement job share a chain of doublings (round 6; hot_tables.h: stmt_job::pair, comb_tables.h: term_ladder16_joint).
// Plain host / device code: the statement classifier reads the array on the device, the plan builds it on the host, tests/host/tr_host_lib.cpp checks the rule.
//
// A verifier constraint is  commitment = sum s_i P_i - c LHS  (verifier.rs:95-106).  A per-proof point with ONE use in the statement (LHS) is a ladder of its
// own: 252 doublings + 64 additions.  Any other per-proof term of the same constraint can ride on those doublings (Straus) for its eight multiples + 64
// additions -- what a comb walk costs -- and its point then needs no comb table if all its terms ride.
//   pair[k] = k' < T            term k (the host: a point of one use) carries term k' of its constraint
//   pair[k'] = STMT_ABSORBED | k   term k' is on no class list and counts as no use of its point; its partial sum is the identity
//   STMT_UNPAIRED               otherwise
#pragma once
#include <cstdint>
#include <vector>
#include "fe25519.h"   // ZKP_HD

namespace zkp {

constexpr uint32_t STMT_UNPAIRED = 0xffffffffu, STMT_ABSORBED = 0x80000000u;
ZKP_HD bool stmt_absorbed(const uint32_t* pair, uint32_t k) { return pair && pair[k] != STMT_UNPAIRED && (pair[k] & STMT_ABSORBED); }

// A per-proof point ALL of whose terms ride and that has at least STMT_RIDER_MIN of them gets a table of its multiples 1 .. 128 (in the place of a 16-teeth
// comb table: 129 entries) -- its riders then add one signed 8-bit digit per byte of the scalar, 32 additions instead of 64, and do not build multiples of
// their own.  `uses` = terms of the point that do NOT ride, `riders` = those that do; ok = the job's tables have 16 teeth.
constexpr uint32_t STMT_RIDER_MIN = 2;
ZKP_HD bool stmt_rider(uint32_t ok, uint32_t p, uint32_t ns, uint32_t uses, uint32_t riders) { return ok && p >= ns && uses == 0 && riders >= STMT_RIDER_MIN; }

// toff[nc + 1]: term offsets of the constraints, tpt[T1]: point id of every term; ids < ns are common to the batch (fixed-base tables or shared comb tables:
// never paired), the others one point per proof.  Per constraint, every term on a per-proof point with one use takes along one other per-proof term of the
// same constraint: first a term of a point with several uses, else another single-use term.  Empty when nothing pairs.
inline std::vector<uint32_t> pair_terms(const uint32_t* toff, const uint32_t* tpt, uint32_t T1, uint32_t nc, uint32_t ns, uint32_t np) {
  std::vector<uint32_t> u(np, 0), pair(T1, STMT_UNPAIRED);
  for (uint32_t i = 0; i < T1; ++i) ++u[tpt[i]];
  bool any = false;
  for (uint32_t k = 0; k < nc; ++k) {
    for (uint32_t h = toff[k]; h < toff[k + 1]; ++h) {
      if (tpt[h] < ns || u[tpt[h]] != 1 || pair[h] != STMT_UNPAIRED) continue;
      uint32_t partner = STMT_UNPAIRED;
      for (uint32_t q = toff[k]; q < toff[k + 1] && partner == STMT_UNPAIRED; ++q)
        if (q != h && tpt[q] >= ns && u[tpt[q]] >= 2 && pair[q] == STMT_UNPAIRED) partner = q;
      for (uint32_t q = toff[k]; q < toff[k + 1] && partner == STMT_UNPAIRED; ++q)
        if (q != h && tpt[q] >= ns && u[tpt[q]] == 1 && pair[q] == STMT_UNPAIRED) partner = q;
      if (partner == STMT_UNPAIRED) continue;
      pair[h] = partner;
      pair[partner] = STMT_ABSORBED | h;
      any = true;
    }
  }
  if (!any) pair.clear();
  return pair;
}

}  // namespace zkp
