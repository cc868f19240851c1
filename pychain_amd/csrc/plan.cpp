// plan.cpp - host-side compiler from the reference graph layout to the device plan.
// See plan_format.h for the format and include/pychain_hip.h for the contract.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <vector>

#include "../../include/pychain_hip.h"
#include "common.h"
#include "plan_format.h"

namespace {

struct Arc { uint32_t i0, i1; float p; };

struct BuiltTile {
  std::vector<WaveEntry> waves;
  std::vector<GroupEntry> groups;   // wave order
  std::vector<uint32_t> slots;      // 2 words per lane per slot-row
  int total_slot_rows = 0, max_wave = 0, nrows = 0;
  std::vector<int> row_order;       // sorted position -> original row id
};

// ---- LDS bank-conflict-aware slot assignment -------------------------------------------
// A wave64 ds_read_b32 is serviced as two 32-lane halves over 32 banks (bank = word address
// mod 32; both operand arrays start on a multiple of 32 words); a half costs as many LDS
// cycles as its most-loaded bank.  Within a group, row r (lane r) may issue its arcs in any
// slot order, so for every 32-row half the arcs are permuted inside their rows to minimise
// the number of same-bank pairs per slot-row, over both gathered operands.  Greedy
// placement followed by a deterministic local search (fixed-seed LCG: plans are reproducible).
// Measured on the C3 graph: 7.0 -> ~3.9 LDS cycles per half slot-row (2.0 = conflict-free).
struct HalfOpt {
  int nrows, nslots;
  std::vector<int> cell;                 // [row*nslots + slot] -> arc index in the row's list, -1 = padding
  std::vector<std::vector<Arc>> const* rows;
  std::vector<int> const* order;
  int pos0;
  std::vector<int> cnt;                  // [slot][operand][bank]
  const Arc* arc(int r, int j) const {
    const int a = cell[r * nslots + j];
    return a < 0 ? nullptr : &(*rows)[(*order)[pos0 + r]][a];
  }
  int& c(int slot, int op, int bank) { return cnt[(slot * 2 + op) * 32 + bank]; }
  void add(int slot, const Arc* a, int d) { if (a) { c(slot, 0, a->i0 & 31) += d; c(slot, 1, a->i1 & 31) += d; } }
  // colliding pairs an arc would have in `slot` (arc itself not counted)
  int cost_in(int slot, const Arc* a) { return a ? c(slot, 0, a->i0 & 31) + c(slot, 1, a->i1 & 31) : 0; }
};

void optimise_half(HalfOpt& h) {
  const int R = h.nrows, A = h.nslots;
  h.cell.assign((size_t)R * A, -1);
  h.cnt.assign((size_t)A * 64, 0);
  // greedy: rows with most arcs first; each arc goes to the free slot of its row with fewest collisions
  std::vector<int> rorder(R);
  std::iota(rorder.begin(), rorder.end(), 0);
  std::stable_sort(rorder.begin(), rorder.end(), [&](int a, int b) {
    return (*h.rows)[(*h.order)[h.pos0 + a]].size() > (*h.rows)[(*h.order)[h.pos0 + b]].size(); });
  for (int r : rorder) {
    const auto& arcs = (*h.rows)[(*h.order)[h.pos0 + r]];
    for (int a = 0; a < (int)arcs.size(); a++) {
      int best = -1, bc = 0;
      for (int j = 0; j < A; j++) {
        if (h.cell[r * A + j] >= 0) continue;
        const int cst = h.cost_in(j, &arcs[a]);
        if (best < 0 || cst < bc) { best = j; bc = cst; }
      }
      h.cell[r * A + best] = a;
      h.add(best, &arcs[a], +1);
    }
  }
  // local search: swap two slots of one row when that does not increase the collision count
  uint32_t rng = 0x9E3779B9u;
  auto next = [&]() { rng = rng * 1664525u + 1013904223u; return rng >> 8; };
  static const long iter_mult = getenv("PYCHAIN_PLAN_ITERS") ? atol(getenv("PYCHAIN_PLAN_ITERS")) : 160;   // tuning knob
  const long iters = (long)R * A * iter_mult;
  for (long it = 0; it < iters; it++) {
    const int r = next() % R, j1 = next() % A, j2 = next() % A;
    if (j1 == j2) continue;
    const Arc* a1 = h.arc(r, j1); const Arc* a2 = h.arc(r, j2);
    if (!a1 && !a2) continue;
    h.add(j1, a1, -1); h.add(j2, a2, -1);
    const int before = h.cost_in(j1, a1) + h.cost_in(j2, a2);
    const int after = h.cost_in(j2, a1) + h.cost_in(j1, a2);
    if (after <= before) {
      std::swap(h.cell[r * A + j1], h.cell[r * A + j2]);
      h.add(j2, a1, +1); h.add(j1, a2, +1);
    } else {
      h.add(j1, a1, +1); h.add(j2, a2, +1);
    }
  }
}

// rows[r] = arcs of original row r.  `order` = row ids sorted by descending degree
// (stable), rows beyond order.size() do not exist.  npos = number of row positions
// (multiple of 64) the output vector has.
BuiltTile build_tile(const std::vector<std::vector<Arc>>& rows, const std::vector<int>& order,
                     int npos, int nwaves) {
  BuiltTile t;
  t.row_order = order;
  t.nrows = (int)order.size();
  const int ngroups = npos / 64;
  std::vector<int> gsl(ngroups, 0);
  for (int g = 0; g < ngroups; g++)
    for (int l = 0; l < 64; l++) {
      const int pos = g * 64 + l;
      if (pos < (int)order.size()) gsl[g] = std::max(gsl[g], (int)rows[order[pos]].size());
    }
  static const int pad_rows = getenv("PYCHAIN_PLAN_PAD") ? atoi(getenv("PYCHAIN_PLAN_PAD")) : 0;   // tuning knob
  for (int g = 0; g < ngroups; g++) if (gsl[g] >= 4) gsl[g] += pad_rows;
  // longest-processing-time-first: groups are already in descending slot order
  std::vector<std::vector<int>> per_wave(nwaves);
  std::vector<int> load(nwaves, 0);
  for (int g = 0; g < ngroups; g++) {
    int w = -1;                     // a wave keeps its group table in one VGPR pair: <= 64 groups
    for (int i = 0; i < nwaves; i++)
      if (per_wave[i].size() < 64 && (w < 0 || load[i] < load[w])) w = i;
    per_wave[w].push_back(g);
    load[w] += gsl[g] + 1;        // +1: the per-group store/bookkeeping cost
  }
  // refine the LPT deal: move or swap single groups while that lowers the heavier of the two
  // waves involved (the frame time of a workgroup is set by its most loaded wave)
  auto cost = [&](int g) { return gsl[g] + 1; };
  for (int pass = 0; pass < 64; pass++) {
    int wmax = 0;
    for (int i = 1; i < nwaves; i++) if (load[i] > load[wmax]) wmax = i;
    bool improved = false;
    for (size_t ia = 0; ia < per_wave[wmax].size() && !improved; ia++) {
      const int ga = per_wave[wmax][ia];
      for (int w2 = 0; w2 < nwaves && !improved; w2++) {
        if (w2 == wmax) continue;
        // move
        if (per_wave[w2].size() < 64 && load[w2] + cost(ga) < load[wmax]) {
          per_wave[w2].push_back(ga); per_wave[wmax].erase(per_wave[wmax].begin() + ia);
          load[w2] += cost(ga); load[wmax] -= cost(ga); improved = true; break;
        }
        // swap
        for (size_t ib = 0; ib < per_wave[w2].size(); ib++) {
          const int gb = per_wave[w2][ib];
          const int d = cost(ga) - cost(gb);
          if (d > 0 && load[w2] + d < load[wmax]) {
            per_wave[wmax][ia] = gb; per_wave[w2][ib] = ga;
            load[wmax] -= d; load[w2] += d; improved = true; break;
          }
        }
      }
    }
    if (!improved) break;
  }
  for (auto& v : per_wave)   // descending slot counts inside a wave: groups without arcs come last
    std::stable_sort(v.begin(), v.end(), [&](int a, int b) { return gsl[a] > gsl[b]; });
  t.waves.resize(nwaves);
  int row_cursor = 0;
  for (int w = 0; w < nwaves; w++) {
    WaveEntry& we = t.waves[w];
    we.first_group = (int)t.groups.size();
    we.ngroups = (int)per_wave[w].size();
    we.slot_row_begin = row_cursor;
    int n = 0;
    for (int g : per_wave[w]) {
      t.groups.push_back(GroupEntry{g * 64, gsl[g]});
      const int A = gsl[g];
      HalfOpt half[2];
      for (int hh = 0; hh < 2 && A > 0; hh++) {
        const int p0 = g * 64 + hh * 32;
        half[hh].nrows = std::max(0, std::min(32, (int)order.size() - p0));
        half[hh].nslots = A; half[hh].rows = &rows; half[hh].order = &order; half[hh].pos0 = p0;
        if (half[hh].nrows > 0) optimise_half(half[hh]);
      }
      for (int j = 0; j < A; j++) {
        // padding lanes re-read the operands of a real arc of their half (an LDS broadcast: no conflict)
        uint32_t fill[2] = {0u, 0u};
        for (int hh = 0; hh < 2; hh++)
          for (int r = 0; r < half[hh].nrows; r++)
            if (const Arc* a = half[hh].arc(r, j)) { fill[hh] = a->i0 | (a->i1 << 16); break; }
        for (int l = 0; l < 64; l++) {
          const int hh = l >> 5, r = l & 31;
          uint32_t idx = fill[hh]; float p = 0.f;
          if (r < half[hh].nrows)
            if (const Arc* a = half[hh].arc(r, j)) { idx = a->i0 | (a->i1 << 16); p = a->p; }
          uint32_t pb; memcpy(&pb, &p, 4);
          t.slots.push_back(idx); t.slots.push_back(pb);
        }
      }
      n += gsl[g];
    }
    we.nslot_rows = n;
    row_cursor += n;
    t.max_wave = std::max(t.max_wave, n);
  }
  t.total_slot_rows = row_cursor;
  return t;
}

std::vector<int> sort_by_degree(const std::vector<int>& deg, const std::vector<int>& ids) {
  std::vector<int> o = ids;
  std::stable_sort(o.begin(), o.end(), [&](int a, int b) { return deg[a] > deg[b]; });
  return o;
}

size_t align16(size_t x) { return (x + 15) & ~size_t(15); }

}  // namespace

extern "C" int64_t pychain_hip_den_plan_build(
    const int32_t* ft, const int32_t* fi, const float* fp,
    const int32_t* bt, const int32_t* bi, const float* bp,
    const float* leaky, const float* initial, const float* final_,
    int H, int K, int D, void* blob, size_t blob_bytes) {
  if (!ft || !fi || !fp || !bt || !bi || !bp || !leaky || !initial || !final_)
    return pychain_hip::fail(PYCHAIN_HIP_EINVAL, "den_plan_build: null graph pointer");
  if (H <= 0 || K <= 0 || D <= 0)
    return pychain_hip::fail(PYCHAIN_HIP_EINVAL, "den_plan_build: empty graph (H=%d K=%d D=%d)", H, K, D);
  if (H > 65535 || D > 65535)
    return pychain_hip::fail(PYCHAIN_HIP_EUNSUPPORTED,
                             "den_plan_build: packed arc format needs num_states and num_pdfs <= 65535 "
                             "(got H=%d D=%d)", H, D);
  for (int h = 0; h < H; h++) {
    if (fi[2 * h] < 0 || fi[2 * h + 1] < fi[2 * h] || fi[2 * h + 1] > K ||
        bi[2 * h] < 0 || bi[2 * h + 1] < bi[2 * h] || bi[2 * h + 1] > K)
      return pychain_hip::fail(PYCHAIN_HIP_EINVAL, "den_plan_build: transition_indices of state %d out of range", h);
  }
  for (int k = 0; k < K; k++) {
    const int32_t* a = ft + 3 * k; const int32_t* b = bt + 3 * k;
    if (a[0] < 0 || a[0] >= H || a[1] < 0 || a[1] >= H || a[2] < 0 || a[2] >= D ||
        b[0] < 0 || b[0] >= H || b[1] < 0 || b[1] >= H || b[2] < 0 || b[2] >= D)
      return pychain_hip::fail(PYCHAIN_HIP_EINVAL, "den_plan_build: transition %d has a state or pdf out of range", k);
  }
  const int Hp = (H + 63) / 64 * 64;
  std::vector<int> indeg(H), outdeg(H), ids(H);
  std::iota(ids.begin(), ids.end(), 0);
  for (int h = 0; h < H; h++) { indeg[h] = bi[2 * h + 1] - bi[2 * h]; outdeg[h] = fi[2 * h + 1] - fi[2 * h]; }
  const std::vector<int> order_a = sort_by_degree(indeg, ids), order_b = sort_by_degree(outdeg, ids);
  std::vector<int> pa(H), pb(H);
  for (int i = 0; i < H; i++) { pa[order_a[i]] = i; pb[order_b[i]] = i; }

  // alpha rows: arcs entering h, in the reference's order (fstext.cc:63-76)
  std::vector<std::vector<Arc>> rows_a(H), rows_b(H), rows_g(D);
  for (int h = 0; h < H; h++) {
    for (int k = bi[2 * h]; k < bi[2 * h + 1]; k++) {
      if (bt[3 * k + 1] != h)
        return pychain_hip::fail(PYCHAIN_HIP_EINVAL, "den_plan_build: backward transition %d is not grouped under its destination", k);
      rows_a[h].push_back(Arc{(uint32_t)pa[bt[3 * k]], (uint32_t)bt[3 * k + 2], bp[k]});
    }
    for (int k = fi[2 * h]; k < fi[2 * h + 1]; k++) {
      if (ft[3 * k] != h)
        return pychain_hip::fail(PYCHAIN_HIP_EINVAL, "den_plan_build: forward transition %d is not grouped under its source", k);
      rows_b[h].push_back(Arc{(uint32_t)pb[ft[3 * k + 1]], (uint32_t)ft[3 * k + 2], fp[k]});
      // gamma rows keep the (state, arc) order the reference accumulates in (chain-computation.cc:293-305)
      rows_g[ft[3 * k + 2]].push_back(Arc{(uint32_t)pa[h], (uint32_t)pb[ft[3 * k + 1]], fp[k]});
    }
  }
  std::vector<int> gdeg(D), gids;
  for (int n = 0; n < D; n++) { gdeg[n] = (int)rows_g[n].size(); if (gdeg[n] > 0) gids.push_back(n); }
  const std::vector<int> order_g = sort_by_degree(gdeg, gids);
  const int gpos = ((int)order_g.size() + 63) / 64 * 64;

  BuiltTile ta = build_tile(rows_a, order_a, Hp, PLAN_REC_WAVES);
  BuiltTile tb = build_tile(rows_b, order_b, Hp, PLAN_REC_WAVES);
  BuiltTile tg = build_tile(rows_g, order_g, gpos, PLAN_GAM_WAVES);
  BuiltTile tg2 = build_tile(rows_g, order_g, gpos, PLAN_GAM2_WAVES);

  // ---- lay the blob out
  size_t off = align16(sizeof(PlanHeader));
  PlanHeader hd;
  memset(&hd, 0, sizeof(hd));
  hd.magic = PLAN_MAGIC; hd.version = PLAN_VERSION;
  hd.H = H; hd.K = K; hd.D = D; hd.Hp = Hp;
  auto place_tile = [&](TilePlan& tp, const BuiltTile& t) {
    tp.ngroups = (int)t.groups.size(); tp.nwaves = (int)t.waves.size();
    tp.total_slot_rows = t.total_slot_rows; tp.max_wave_slot_rows = t.max_wave; tp.nrows = t.nrows;
    tp.off_wave_tab = (int32_t)off; off = align16(off + t.waves.size() * sizeof(WaveEntry));
    tp.off_group_tab = (int32_t)off; off = align16(off + std::max<size_t>(1, t.groups.size()) * sizeof(GroupEntry));
    tp.off_slots = (int32_t)off; off = align16(off + std::max<size_t>(1, t.slots.size()) * 4);
  };
  place_tile(hd.alpha, ta); place_tile(hd.beta, tb); place_tile(hd.gamma, tg); place_tile(hd.gamma2, tg2);
  auto place_vec = [&](int32_t& o, size_t n) { o = (int32_t)off; off = align16(off + n * 4); };
  place_vec(hd.off_init_a, Hp); place_vec(hd.off_leaky_a, Hp); place_vec(hd.off_final_a, Hp);
  place_vec(hd.off_leaky_b, Hp); place_vec(hd.off_final_b, Hp);
  place_vec(hd.off_row_pdf, std::max(gpos, 64));
  if (off > (size_t)INT32_MAX)
    return pychain_hip::fail(PYCHAIN_HIP_EUNSUPPORTED, "den_plan_build: plan larger than 2 GiB");
  hd.total_bytes = (int32_t)off;
  if (!blob || blob_bytes < off) return (int64_t)off;

  char* base = (char*)blob;
  memset(base, 0, off);
  memcpy(base, &hd, sizeof(hd));
  auto write_tile = [&](const TilePlan& tp, const BuiltTile& t) {
    memcpy(base + tp.off_wave_tab, t.waves.data(), t.waves.size() * sizeof(WaveEntry));
    memcpy(base + tp.off_group_tab, t.groups.data(), t.groups.size() * sizeof(GroupEntry));
    memcpy(base + tp.off_slots, t.slots.data(), t.slots.size() * 4);
  };
  write_tile(hd.alpha, ta); write_tile(hd.beta, tb); write_tile(hd.gamma, tg); write_tile(hd.gamma2, tg2);
  float* init_a = (float*)(base + hd.off_init_a); float* leaky_a = (float*)(base + hd.off_leaky_a);
  float* final_a = (float*)(base + hd.off_final_a); float* leaky_b = (float*)(base + hd.off_leaky_b);
  float* final_b = (float*)(base + hd.off_final_b); int32_t* row_pdf = (int32_t*)(base + hd.off_row_pdf);
  for (int h = 0; h < H; h++) {
    init_a[pa[h]] = initial[h]; leaky_a[pa[h]] = leaky[h]; final_a[pa[h]] = final_[h];
    leaky_b[pb[h]] = leaky[h]; final_b[pb[h]] = final_[h];
  }
  for (int i = 0; i < std::max(gpos, 64); i++) row_pdf[i] = i < (int)order_g.size() ? order_g[i] : -1;
  return (int64_t)off;
}
