// plan.cpp - host-side compiler from the reference graph layout to the device plan.
// See plan_format.h for the format and include/pychain_hip.h for the contract.
//
// Most of this file decides WHICH arcs meet in one LDS gather instruction.  A wave64
// ds_read_b32 is serviced as two 32-lane halves over 32 banks (bank = dword address mod 32);
// a half costs as many LDS cycles as its most-loaded bank, and the recursion kernels are
// bound by exactly these cycles.  Three freedoms are used, none of which the
// kernels can see (the plan format does not change):
//   1. rows of equal arc count may trade places: which 32 rows form a half-group, and - the
//      position of a state in the LDS vector being its row position - in which bank every
//      state lives                                                    (Balancer, annealing);
//   2. every recursion group gets spare slot-rows, as many as the register-resident loop of the
//      kernel walks anyway: a half-group that is 97 % full cannot avoid a bank holding more
//      than its share of the arcs;
//   3. a row may issue its arcs in any slot order                     (SlotOrder, annealing).
// Modelled LDS cycles per half slot-row for the two gathers of an arc on the C3 graph
// (2.0 = conflict-free): natural order 7.0, slot order alone 3.8, all three 2.3 at 19 % more
// slot-rows.  Deterministic (fixed-seed LCG): the same graph always gives the same plan.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <thread>
#include <vector>

#include "../../include/pychain_hip.h"
#include "common.h"
#include "plan_format.h"

namespace {

long env_long(const char* name, long dflt) { const char* e = getenv(name); return e ? atol(e) : dflt; }

// An arc of a tile: the entities (state ids / pdf ids) of its two gathered operands and its probability.
struct Arc { int e0, e1; float p; };

// Where the entities of an operand vector live: dword position in LDS (bank = position mod 32).
enum { kLayA = 0, kLayB = 1, kLayX = 2 };   // alpha' states, beta states, nnet-output row (identity)

struct Tile {
  const std::vector<std::vector<Arc>>* rows = nullptr;   // arcs of every row, by row id
  std::vector<int> order;            // row position -> row id; positions are cut into groups of 64
  int npos = 0;                      // row positions (multiple of 64)
  int lay[2] = {0, 0};               // layout of operand 0 / 1
  int own_layout = -1;               // the layout whose positions ARE this tile's row positions (-1: none)
  int weight = 1;                    // weight of this tile's LDS cycles in the objective
  std::vector<int> gsl;              // slot-rows per group
  bool fitted = false;               // gsl was fitted to a 32-row loop (init_tile): little slack, worth a longer annealing
};

struct Layouts {
  std::vector<int> pos[3];           // entity -> position
  int bank(int l, int e) const { return pos[l][e] & 31; }
};

struct Lcg {
  uint32_t s;
  uint32_t next() { s = s * 1664525u + 1013904223u; return s >> 8; }
  double unit() { return (double)(next() & 0xffffff) * (1.0 / 16777216.0); }
};

// ---- freedom 1: row placement -------------------------------------------------------------------
// Objective: in every half-group (32 rows x A slot-rows) no bank holds more than A arcs, for either
// operand - the necessary condition for a conflict-free slot order (energy: overload, then squares).
// Move: two rows of equal arc count trade places; in the alpha / beta tiles that also swaps the two
// states' positions in the alpha' / beta vector, i.e. the banks of every arc that gathers them.
struct Balancer {
  std::vector<Tile>& tiles;
  Layouts& lay;
  std::vector<int> hg0;                              // first half-group id of a tile
  std::vector<int> hist, cap, wgt;                   // [hg][op][32]; capacity and weight of a half-group
  std::vector<std::vector<int>> pos_of_row;          // [tile][row id] -> row position
  struct Occ { int tile, row, op; };
  std::vector<std::vector<Occ>> occ[2];              // [layout A / B][state]: arcs that gather this state
  std::vector<std::vector<int>> gmax;                // [tile][group]: arc count of the group's longest row (fixed: it sets the slot-rows)
  static constexpr int same_lane_moves = 2;          // eighths of the moves that keep the row's lane (C3: -0.8 % against none)
  static constexpr bool free_moves = true;           // rows of a group permute freely; rows of different arc counts trade where no group grows
  Lcg rng{0x2545F491u};

  Balancer(std::vector<Tile>& t, Layouts& l) : tiles(t), lay(l) {
    int n = 0;
    for (auto& tl : tiles) {
      hg0.push_back(n);
      for (int g : tl.gsl) for (int hh = 0; hh < 2; hh++) { cap.push_back(g); wgt.push_back(tl.weight); }
      n += 2 * (int)tl.gsl.size();
    }
    hist.assign((size_t)n * 64, 0);
    occ[0].resize(lay.pos[kLayA].size()); occ[1].resize(lay.pos[kLayB].size());
    pos_of_row.resize(tiles.size());
    gmax.resize(tiles.size());
    for (size_t ti = 0; ti < tiles.size(); ti++) {
      const Tile& tl = tiles[ti];
      gmax[ti].assign(tl.npos / 64 + 1, 0);
      for (size_t i = 0; i < tl.order.size(); i++) gmax[ti][i / 64] = std::max(gmax[ti][i / 64], (int)(*tl.rows)[tl.order[i]].size());
      pos_of_row[ti].assign(tl.rows->size(), -1);
      for (size_t i = 0; i < tl.order.size(); i++) pos_of_row[ti][tl.order[i]] = (int)i;
      for (int row : tl.order)
        for (const Arc& a : (*tl.rows)[row]) {
          const int e[2] = {a.e0, a.e1};
          for (int op = 0; op < 2; op++) {
            if (tl.lay[op] != kLayX) occ[tl.lay[op]][e[op]].push_back(Occ{(int)ti, row, op});
            h(hg_of((int)ti, row), op)[lay.bank(tl.lay[op], e[op])]++;
          }
        }
    }
  }
  int* h(int hg, int op) { return &hist[((size_t)hg * 2 + op) * 32]; }
  int hg_of(int ti, int row) const { return hg0[ti] + pos_of_row[ti][row] / 32; }
  static long phi(int x, int c, int w) { return (long)w * ((x > c ? 64L * (x - c) : 0L) + (long)x * x); }
  long overload() const {
    long o = 0;
    for (size_t hg = 0; hg < cap.size(); hg++) for (int k = 0; k < 64; k++) o += std::max(0, hist[hg * 64 + k] - cap[hg]);
    return o;
  }
  long shift(int hg, int op, int ob, int nb) {       // one arc of (hg, op) from bank ob to bank nb
    int* q = h(hg, op);
    const long d = phi(q[ob] - 1, cap[hg], wgt[hg]) - phi(q[ob], cap[hg], wgt[hg]) +
                   phi(q[nb] + 1, cap[hg], wgt[hg]) - phi(q[nb], cap[hg], wgt[hg]);
    q[ob]--; q[nb]++;
    return d;
  }
  long rebank(int l, int e, int ob, int nb) {        // every arc that gathers state e of layout l changes bank
    long d = 0;
    if (ob != nb) for (const Occ& o : occ[l][e]) d += shift(hg_of(o.tile, o.row), o.op, ob, nb);
    return d;
  }
  long move_row(int ti, int row, int from, int to) {  // the arcs OF row change half-group
    long d = 0;
    const Tile& tl = tiles[ti];
    for (const Arc& a : (*tl.rows)[row]) {
      const int e[2] = {a.e0, a.e1};
      for (int op = 0; op < 2; op++) {
        const int b = lay.bank(tl.lay[op], e[op]);
        int* qf = h(from, op); int* qt = h(to, op);
        d += phi(qf[b] - 1, cap[from], wgt[from]) - phi(qf[b], cap[from], wgt[from]); qf[b]--;
        d += phi(qt[b] + 1, cap[to], wgt[to]) - phi(qt[b], cap[to], wgt[to]); qt[b]++;
      }
    }
    return d;
  }
  void run(long iters, double t0, double t1) {
    if (iters <= 0) return;
    const double cool = pow(t1 / t0, 1.0 / (double)iters);
    double T = t0;
    for (long it = 0; it < iters; it++, T *= cool) {
      const int ti = rng.next() % tiles.size();
      Tile& tl = tiles[ti];
      const int n = (int)tl.order.size();
      if (n <= 32) continue;
      const int p1 = rng.next() % n;
      // rows are sorted by arc count: equal counts are neighbours.  Half of the moves stay inside p1's group of 64
      // (every permutation of a group's rows is free: its slot-row count is that of its longest row, whichever lane owns it)
      int p2;
      const uint32_t kind = rng.next() & 7;
      if ((int)kind < same_lane_moves) {
        // the row keeps its lane - the bank its state is gathered from - and changes its half-group: only the bank
        // histograms of its own arcs move (the two coordinates of a position are separate freedoms)
        const int k = 1 + (int)(rng.next() % 16);
        p2 = p1 + ((rng.next() & 1) ? 32 * k : -32 * k);
      } else if (free_moves && (kind & 1)) {
        p2 = (p1 & ~63) + (int)(rng.next() & 63);
      } else {
        const int span = 1 + rng.next() % 512;
        p2 = p1 + ((rng.next() & 1) ? span : -span);
      }
      if (p2 < 0 || p2 >= n || p2 == p1) continue;
      const int r1 = tl.order[p1], r2 = tl.order[p2];
      const int d1 = (int)(*tl.rows)[r1].size(), d2 = (int)(*tl.rows)[r2].size();
      if (d1 != d2) {
        // rows of different arc counts may trade places where neither group's longest row grows
        if (!free_moves || d1 > gmax[ti][p2 / 64] || d2 > gmax[ti][p1 / 64]) continue;
      } else if (!free_moves && p2 / 32 == p1 / 32) continue;
      const int h1 = hg0[ti] + p1 / 32, h2 = hg0[ti] + p2 / 32;
      const int L = tl.own_layout;
      if (h1 == h2 && L < 0) continue;                  // nothing changes
      // apply, evaluate, undo on rejection (all updates are exact inverses of each other)
      long d = h1 == h2 ? 0 : move_row(ti, r1, h1, h2) + move_row(ti, r2, h2, h1);
      pos_of_row[ti][r1] = p2; pos_of_row[ti][r2] = p1;
      if (L >= 0) {
        d += rebank(L, r1, p1 & 31, p2 & 31); lay.pos[L][r1] = p2;
        d += rebank(L, r2, p2 & 31, p1 & 31); lay.pos[L][r2] = p1;
      }
      if (d <= 0 || rng.unit() < exp(-(double)d / T)) {
        tl.order[p1] = r2; tl.order[p2] = r1;
      } else {
        if (L >= 0) {
          rebank(L, r2, p1 & 31, p2 & 31); lay.pos[L][r2] = p2;
          rebank(L, r1, p2 & 31, p1 & 31); lay.pos[L][r1] = p1;
        }
        pos_of_row[ti][r1] = p1; pos_of_row[ti][r2] = p2;
        if (h1 != h2) { move_row(ti, r2, h1, h2); move_row(ti, r1, h2, h1); }
      }
    }
  }
};

// ---- freedom 3: slot order ------------------------------------------------------------------------
// cell[g][lane][slot] = index of the arc in its row's list, -1 = padding.  Per half-group: greedy
// placement, then simulated annealing on  sum over columns and operands of
// kLambda * (max bank load) + sum of squared bank loads;  a move swaps two slots of one row.
struct SlotOrder {
  const Tile& t;
  const Layouts& lay;
  std::vector<int> cell_off, cell;
  std::atomic<long> cycles{0}, columns{0};           // sum over half slot-rows and operands of the fullest bank / half slot-rows (statistics)
  std::atomic<long> cycles_op[2] = {{0}, {0}};       // ... per operand
  std::atomic<long> cost_tenths{0}, slot_rows{0};    // modelled extra LDS cycles (x 10) of the conflicts / slot-rows
  // What a gather with a fullest bank of L lanes (the larger of its two 32-lane halves: they are served side by side)
  // costs beyond a conflict-free one, in tenths of a cycle, measured on gfx950 (tools/ubench/ldsbanks.hip,
  // profiles/r03_ubench_ldsbanks.txt): ds_read_b32  L=2 +0.6, L=4 +2.8, L=8 +7.0;  ds_read_b64  L=2 +0.2, L=4 +2.8, L=8 +7.1.
  // (A conflict-free wave64 gather occupies the LDS for 2.0 / 2.3 cycles - an all-padding slot-row costs that too.)
  int wide_op[2] = {0, 0};                           // operand gathered with ds_read_b64 (state vectors of the lazy recursions)
  bool ignore_op1 = false;                           // "pdf by state" plans: the recursions do not gather the nnet-output operand per arc
  int extra_tenths(int op, int L) const {
    if (L <= 1) return 0;
    if (L == 2) return wide_op[op] ? 2 : 6;
    if (L == 3) return 18;
    return 10 * L - 12;
  }

  SlotOrder(const Tile& tile, const Layouts& l, bool op0_wide, bool op1_wide) : t(tile), lay(l) {
    wide_op[0] = op0_wide; wide_op[1] = op1_wide;
    int off = 0;
    for (int g : t.gsl) { cell_off.push_back(off); off += 64 * g; }
    cell.assign(off, -1);
  }
  // one slot-row (column) of a group: bank loads of both halves and both operands
  struct Col { int cnt[2][2][32]; int nm[2][2][34]; int mx[2][2]; };
  // Energy of a column and operand: w * (fullest bank of half 0 + of half 1), both operands alike, + the sum of squared bank
  // loads.  This is the model the frame follows in situ (C3, 32-row loops: 1313 -> 878 of these units = 3.27 -> 3.09 ms,
  // profiles/r03_h_split_arcs.txt); a model built on the ISOLATED gather costs of tools/ubench/ldsbanks.hip (halves side by
  // side, a two-way conflict nearly free) gives plans that look better and run slower.
  static constexpr int kW = 12;
  int col_energy_op(const Col& c, int op) const { return kW * (c.mx[0][op] + c.mx[1][op]); }
  int col_add(Col& c, int hh, int op, int b, int d) const {   // returns the energy change
    if (op == 1 && ignore_op1) return 0;
    int& x = c.cnt[hh][op][b];
    const int before = col_energy_op(c, op) + x * x;
    c.nm[hh][op][x]--; x += d; c.nm[hh][op][x]++;
    if (d > 0) { if (x > c.mx[hh][op]) c.mx[hh][op] = x; }
    else { while (c.mx[hh][op] > 0 && c.nm[hh][op][c.mx[hh][op]] == 0) c.mx[hh][op]--; }
    return col_energy_op(c, op) + x * x - before;
  }
  void group(int g, long moves_per_cell) {
    Lcg rng{0x9E3779B9u ^ (uint32_t)((g * 2) * 2654435761u)};   // per group: results do not depend on the thread count
    const int A = t.gsl[g];
    int nr = 0;
    for (int r = 0; r < 64; r++) if (g * 64 + r < (int)t.order.size()) nr = r + 1;
    if (A == 0 || nr == 0) return;
    int* cl = &cell[cell_off[g]];                    // [row][slot]
    std::vector<Col> cs(A);
    for (auto& c : cs) { memset(&c, 0, sizeof(c)); for (int hh = 0; hh < 2; hh++) c.nm[hh][0][0] = c.nm[hh][1][0] = 32; }
    std::vector<int> b0(nr * A, -1), b1(nr * A, -1);   // banks of a cell's arc
    // greedy: rows with most arcs first; each arc goes to the free slot of its row with fewest collisions in its half
    std::vector<int> rorder(nr);
    std::iota(rorder.begin(), rorder.end(), 0);
    auto arcs_of = [&](int r) -> const std::vector<Arc>& { return (*t.rows)[t.order[g * 64 + r]]; };
    std::stable_sort(rorder.begin(), rorder.end(), [&](int x, int y) { return arcs_of(x).size() > arcs_of(y).size(); });
    for (int r : rorder) {
      const auto& arcs = arcs_of(r);
      const int hh = r >> 5;
      for (int a = 0; a < (int)arcs.size(); a++) {
        const int x0 = lay.bank(t.lay[0], arcs[a].e0), x1 = lay.bank(t.lay[1], arcs[a].e1);
        int best = -1, bc = 0;
        for (int j = 0; j < A; j++) {
          if (cl[r * A + j] >= 0) continue;
          const int cst = cs[j].cnt[hh][0][x0] + cs[j].cnt[hh][1][x1];
          if (best < 0 || cst < bc) { best = j; bc = cst; }
        }
        cl[r * A + best] = a; b0[r * A + best] = x0; b1[r * A + best] = x1;
        col_add(cs[best], hh, 0, x0, +1); col_add(cs[best], hh, 1, x1, +1);
      }
    }
    if (A >= 2) {
      const long iters = moves_per_cell * nr * A;
      const double t0 = 3.0, t1 = 0.03;
      const double cool = iters > 1 ? pow(t1 / t0, 1.0 / (double)iters) : 1.0;
      double T = t0;
      for (long it = 0; it < iters; it++, T *= cool) {
        int r = rng.next() % nr;
        const int j1 = rng.next() % A;
        if (rng.next() & 1) {
          // half of the moves start from a lane that sits in the fullest bank of a conflicting half-column (a blind pick
          // mostly proposes to move arcs that collide with nobody)
          const int hh = rng.next() & 1, op = rng.next() & 1;
          const Col& c = cs[j1];
          if (c.mx[hh][op] >= 2) {
            int bmax = 0;
            for (int b = 1; b < 32; b++) if (c.cnt[hh][op][b] > c.cnt[hh][op][bmax]) bmax = b;
            const std::vector<int>& bank = op ? b1 : b0;
            int pick = -1, seen = 0;
            for (int rr = 32 * hh; rr < std::min(nr, 32 * hh + 32); rr++)
              if (bank[rr * A + j1] == bmax && (rng.next() % ++seen) == 0) pick = rr;
            if (pick >= 0) r = pick;
          }
        }
        int j2 = rng.next() % (A - 1); if (j2 >= j1) j2++;
        const int hh = r >> 5;
        // the energy change of lane r's cells j1 and j2 trading places, applied; `revert` undoes it
        auto apply = [&](int jj) {
          const int a1 = r * A + j1, a2 = r * A + jj;
          int d = 0;
          if (b0[a1] >= 0) d += col_add(cs[j1], hh, 0, b0[a1], -1) + col_add(cs[j1], hh, 1, b1[a1], -1);
          if (b0[a2] >= 0) d += col_add(cs[jj], hh, 0, b0[a2], -1) + col_add(cs[jj], hh, 1, b1[a2], -1);
          if (b0[a1] >= 0) d += col_add(cs[jj], hh, 0, b0[a1], +1) + col_add(cs[jj], hh, 1, b1[a1], +1);
          if (b0[a2] >= 0) d += col_add(cs[j1], hh, 0, b0[a2], +1) + col_add(cs[j1], hh, 1, b1[a2], +1);
          return d;
        };
        auto revert = [&](int jj) {
          const int a1 = r * A + j1, a2 = r * A + jj;
          if (b0[a2] >= 0) { col_add(cs[j1], hh, 0, b0[a2], -1); col_add(cs[j1], hh, 1, b1[a2], -1); }
          if (b0[a1] >= 0) { col_add(cs[jj], hh, 0, b0[a1], -1); col_add(cs[jj], hh, 1, b1[a1], -1); }
          if (b0[a2] >= 0) { col_add(cs[jj], hh, 0, b0[a2], +1); col_add(cs[jj], hh, 1, b1[a2], +1); }
          if (b0[a1] >= 0) { col_add(cs[j1], hh, 0, b0[a1], +1); col_add(cs[j1], hh, 1, b1[a1], +1); }
        };
        const int i1 = r * A + j1, i2 = r * A + j2;
        if (b0[i1] < 0 && b0[i2] < 0) continue;
        const int dE = apply(j2);
        if (dE <= 0 || rng.unit() < exp(-(double)dE / T)) {
          std::swap(cl[i1], cl[i2]); std::swap(b0[i1], b0[i2]); std::swap(b1[i1], b1[i2]);
        } else {
          revert(j2);
        }
      }
    }
    long cyc0 = 0, cyc1 = 0, extra = 0;
    for (int j = 0; j < A; j++) {
      for (int hh = 0; hh < 2; hh++) { cyc0 += std::max(cs[j].mx[hh][0], 1); cyc1 += std::max(cs[j].mx[hh][1], 1); }
      for (int op = 0; op < 2; op++) extra += extra_tenths(op, std::max(cs[j].mx[0][op], cs[j].mx[1][op]));
    }
    cycles += cyc0 + cyc1; columns += 2 * A;
    cycles_op[0] += cyc0; cycles_op[1] += cyc1;
    cost_tenths += extra; slot_rows += A;
  }
  // groups are independent: annealed on up to 32 host threads
  void run(long moves_per_cell) {
    const int n = (int)t.gsl.size();
    std::atomic<int> next{0};
    auto work = [&]() { for (int i; (i = next++) < n;) group(i, moves_per_cell); };
    const int nthreads = (int)std::min<long>(std::min<unsigned>(32u, std::max(1u, std::thread::hardware_concurrency())), (long)n);
    std::vector<std::thread> pool;
    for (int i = 1; i < nthreads; i++) pool.emplace_back(work);
    work();
    for (auto& th : pool) th.join();
  }
  // modelled LDS-busy cycles of the tile per frame: every slot-row issues one gather per operand
  double lds_cycles() const {
    const double base = (wide_op[0] ? 2.3 : 2.0) + (wide_op[1] ? 2.3 : 2.0);
    return base * (double)slot_rows.load() + 0.1 * (double)cost_tenths.load();
  }
};

struct BuiltTile {
  std::vector<WaveEntry> waves;
  std::vector<GroupEntry> groups;   // wave order
  std::vector<uint32_t> slots;      // 2 words per lane per slot-row
  int total_slot_rows = 0, max_wave = 0, nrows = 0;
};

// Deal groups (slot-row counts `gsl`) to `nwaves` waves: longest-processing-time-first, then single
// moves / swaps while they lower the heavier wave (the frame time of a workgroup is set by its most
// loaded wave).  Returns the groups of every wave, in descending slot count.
// `max_groups`: groups a wave may own - the lazy recursions keep the bases of four in registers (den_lazy.inc.h: kMaxGroups),
// so a recursion tile of at most 4 x nwaves groups is dealt under that limit (rec_group_limit); else a wave keeps its
// group table in one VGPR pair: <= 64.
int rec_group_limit(size_t ngroups, int nwaves) { return (int)ngroups <= 4 * nwaves ? 4 : 64; }
std::vector<std::vector<int>> deal_groups(const std::vector<int>& gsl, int nwaves, int max_groups = 64) {
  const int ngroups = (int)gsl.size();
  std::vector<int> by_size(ngroups);
  std::iota(by_size.begin(), by_size.end(), 0);
  std::stable_sort(by_size.begin(), by_size.end(), [&](int a, int b) { return gsl[a] > gsl[b]; });
  std::vector<std::vector<int>> per_wave(nwaves);
  std::vector<int> load(nwaves, 0);
  auto cost = [&](int g) { return gsl[g] + 1; };     // +1: the per-group store/bookkeeping cost
  for (int g : by_size) {
    int w = -1;
    for (int i = 0; i < nwaves; i++)
      if ((int)per_wave[i].size() < max_groups && (w < 0 || load[i] < load[w])) w = i;
    per_wave[w].push_back(g);
    load[w] += cost(g);
  }
  for (int pass = 0; pass < 64; pass++) {
    int wmax = 0;
    for (int i = 1; i < nwaves; i++) if (load[i] > load[wmax]) wmax = i;
    bool improved = false;
    for (size_t ia = 0; ia < per_wave[wmax].size() && !improved; ia++) {
      const int ga = per_wave[wmax][ia];
      for (int w2 = 0; w2 < nwaves && !improved; w2++) {
        if (w2 == wmax) continue;
        if ((int)per_wave[w2].size() < max_groups && load[w2] + cost(ga) < load[wmax]) {          // move
          per_wave[w2].push_back(ga); per_wave[wmax].erase(per_wave[wmax].begin() + ia);
          load[w2] += cost(ga); load[wmax] -= cost(ga); improved = true; break;
        }
        for (size_t ib = 0; ib < per_wave[w2].size(); ib++) {                          // swap
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
  return per_wave;
}
int max_wave_rows(const std::vector<int>& gsl, int nwaves, int max_groups = 64) {
  int mx = 0;
  for (const auto& w : deal_groups(gsl, nwaves, max_groups)) { int n = 0; for (int g : w) n += gsl[g]; mx = std::max(mx, n); }
  return mx;
}

// Lay the slot stream of a tile out in wave order for a given dealing of its groups to waves.
BuiltTile emit_tile(const Tile& t, const SlotOrder& so, const Layouts& lay, const std::vector<std::vector<int>>& per_wave) {
  BuiltTile o;
  o.nrows = (int)t.order.size();
  const std::vector<int>& gsl = t.gsl;
  const int nwaves = (int)per_wave.size();
  o.waves.resize(nwaves);
  int row_cursor = 0;
  for (int w = 0; w < nwaves; w++) {
    WaveEntry& we = o.waves[w];
    we.first_group = (int)o.groups.size();
    we.ngroups = (int)per_wave[w].size();
    we.slot_row_begin = row_cursor;
    int n = 0;
    for (int g : per_wave[w]) {
      o.groups.push_back(GroupEntry{g * 64, gsl[g]});
      const int A = gsl[g];
      for (int j = 0; j < A; j++) {
        // padding lanes re-read the operands of a real arc of their half (an LDS broadcast: no conflict)
        uint32_t words[64]; float probs[64]; bool real[64];
        uint32_t fill[2] = {0u, 0u};
        for (int l = 0; l < 64; l++) {
          real[l] = false; probs[l] = 0.f; words[l] = 0u;
          const int pos = g * 64 + l;
          if (pos >= (int)t.order.size()) continue;
          const int ac = so.cell[so.cell_off[g] + l * A + j];
          if (ac < 0) continue;
          const Arc& arc = (*t.rows)[t.order[pos]][ac];
          words[l] = (uint32_t)lay.pos[t.lay[0]][arc.e0] | ((uint32_t)lay.pos[t.lay[1]][arc.e1] << 16);
          probs[l] = arc.p; real[l] = true;
          if (!fill[l >> 5]) fill[l >> 5] = words[l];
        }
        // PYCHAIN_PLAN_LINEAR=1 (timing experiments only - WRONG RESULTS): every gather lane-linear, i.e. conflict-free
        const bool linear = env_long("PYCHAIN_PLAN_LINEAR", 0) != 0;
        for (int l = 0; l < 64; l++) {
          uint32_t pb; memcpy(&pb, &probs[l], 4);
          uint32_t wd = real[l] ? words[l] : fill[l >> 5];
          if (linear) wd = (uint32_t)(l + 64 * (j & 7)) | ((uint32_t)(l + 64 * (j & 7)) << 16);
          o.slots.push_back(wd); o.slots.push_back(pb);
        }
      }
      n += gsl[g];
    }
    we.nslot_rows = n;
    row_cursor += n;
    o.max_wave = std::max(o.max_wave, n);
  }
  o.total_slot_rows = row_cursor;
  return o;
}

// Spare slot-rows per group such that no wave exceeds `target` rows: start from the uniform slack that fills
// the budget, take a row back from the fullest wave's most padded group while a wave is over, then hand rows
// out again (smallest groups first) where they still fit.  Groups of fewer than 4 rows get none (as with_slack).
std::vector<int> fit_slack(const std::vector<int>& base, int nwaves, int target, int max_groups = 64) {
  const int ng = (int)base.size();
  long total = 0, elig = 0;
  for (int g = 0; g < ng; g++) { total += base[g]; if (base[g] >= 4) elig++; }
  std::vector<int> sl(ng, 0);
  if (elig == 0) return base;
  const int k0 = (int)std::max(0L, std::min(4L, ((long)target * nwaves - total) / elig));
  for (int g = 0; g < ng; g++) if (base[g] >= 4) sl[g] = k0;
  auto rows = [&]() { std::vector<int> v = base; for (int g = 0; g < ng; g++) v[g] += sl[g]; return v; };
  for (int iter = 0; iter < 4 * ng; iter++) {
    const std::vector<int> v = rows();
    const auto deal = deal_groups(v, nwaves, max_groups);
    int wmax = 0, mx = -1;
    for (int w = 0; w < nwaves; w++) { int n = 0; for (int g : deal[w]) n += v[g]; if (n > mx) { mx = n; wmax = w; } }
    if (mx <= target) break;
    int pick = -1;
    for (int g : deal[wmax]) if (sl[g] > 0 && (pick < 0 || sl[g] > sl[pick] || (sl[g] == sl[pick] && v[g] > v[pick]))) pick = g;
    if (pick < 0) break;                                   // nothing left to take back: the caller checks the result
    sl[pick]--;
  }
  if (max_wave_rows(rows(), nwaves, max_groups) > target) return base;
  std::vector<int> order(ng);
  std::iota(order.begin(), order.end(), 0);
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return base[a] + sl[a] < base[b] + sl[b]; });
  for (int g : order) {
    if (base[g] < 4 || sl[g] >= 4) continue;
    sl[g]++;
    if (max_wave_rows(rows(), nwaves, max_groups) > target) sl[g]--;
  }
  return rows();
}

std::vector<int> sort_by_degree(const std::vector<int>& deg, const std::vector<int>& ids) {
  std::vector<int> o = ids;
  std::stable_sort(o.begin(), o.end(), [&](int a, int b) { return deg[a] > deg[b]; });
  return o;
}

// slot-rows of every group of 64 row positions without slack: the arc count of the group's longest row
std::vector<int> group_rows(const std::vector<std::vector<Arc>>& rows, const std::vector<int>& order, int npos, bool recursion) {
  const int ng = npos / 64;
  std::vector<int> gsl(ng, 0);
  for (int g = 0; g < ng; g++)
    for (int l = 0; l < 64; l++) {
      const int pos = g * 64 + l;
      if (pos < (int)order.size()) gsl[g] = std::max(gsl[g], (int)rows[order[pos]].size());
    }
  // a recursion group without arcs still gets one (all-padding) slot-row: every group then has a group end
  // in every frame, which is where the lazy-normalisation kernel writes a row's value (zeros here)
  if (recursion)
    for (int g = 0; g < ng; g++) gsl[g] = std::max(gsl[g], 1);
  return gsl;
}
// a dealing to `nwaves` waves in which no wave owns more than `max_rows` slot-rows or `max_groups` groups?
bool dealing_fits(const std::vector<int>& gsl, int nwaves, int max_rows, int max_groups) {
  for (const auto& w : deal_groups(gsl, nwaves, rec_group_limit(gsl.size(), nwaves))) {
    int n = 0;
    for (int g : w) n += gsl[g];
    if (n > max_rows || (int)w.size() > max_groups) return false;
  }
  return true;
}

void init_tile(Tile& t, const std::vector<std::vector<Arc>>& rows, const std::vector<int>& order, int npos,
               int lay0, int lay1, int own_layout, int weight, int slack, int nwaves) {
  t.rows = &rows; t.order = order; t.npos = npos; t.lay[0] = lay0; t.lay[1] = lay1; t.own_layout = own_layout; t.weight = weight;
  const int ng = npos / 64;
  // (a recursion tile is dealt under the lazy kernels' limit of four groups per wave where its groups allow: rec_group_limit)
  const int lim = own_layout >= 0 ? rec_group_limit((size_t)ng, nwaves) : 64;
  t.gsl = group_rows(rows, order, npos, own_layout >= 0);
  // freedom 2: spare slot-rows per group.  The kernels keep 16, 32 or 40 slot-rows of a wave in
  // registers and walk all of them every frame, so slack is free up to the next of those sizes:
  // take the smallest size that leaves room for >= 2 spare rows per group, then as many (<= slack)
  // as fit.  slack < 0: choose automatically (<= 4).
  auto with_slack = [&](int k) { std::vector<int> v = t.gsl; for (int& x : v) if (x >= 4) x += k; return v; };
  if (slack >= 0) { t.gsl = with_slack(slack); return; }
  static const int kResident[3] = {PLAN_RESIDENT_0, PLAN_RESIDENT_1, PLAN_RESIDENT_2};
  const int base = max_wave_rows(t.gsl, nwaves, lim);
  if (base > kResident[2]) return;                                  // the tail is streamed anyway
  // PYCHAIN_PLAN_FIT=n (32 <= n < 40): per-group slack fitted to an n-row wave budget (fit_slack) instead of uniform
  // slack; -1: never.  Default (0): fitted to the 32-row loop where uniform slack >= 2 does not land there but about one
  // spare row per two groups does - the 32-row loops keep their arcs in 2.5 registers each (den_lazy.inc.h: LazyArcsSplit),
  // 2.5 VALU instructions per arc instead of 4, and then the frame follows the bank conflicts that are left:
  // C3 623 -> 512 slot-rows, recursion 3.33 -> 3.08 ms (profiles/r03_h_split_arcs.txt).  [Rounds 1-2, all arcs packed:
  // 36 and 32 fitted rows measured no faster than 40 with slack.]
  const long fit_target = env_long("PYCHAIN_PLAN_FIT", 0);
  if (fit_target >= kResident[1] && fit_target < kResident[2] && base <= fit_target) {
    t.gsl = fit_slack(t.gsl, nwaves, (int)fit_target, lim);
    t.fitted = true;
    return;
  }
  if (fit_target == 0 && own_layout >= 0 && base <= kResident[1] && base > kResident[0] &&
      max_wave_rows(with_slack(2), nwaves, lim) > kResident[1]) {
    const std::vector<int> fit = fit_slack(t.gsl, nwaves, kResident[1], lim);
    long spare = 0, elig = 0;
    for (int g = 0; g < ng; g++) { spare += fit[g] - t.gsl[g]; if (t.gsl[g] >= 4) elig++; }
    if (getenv("PYCHAIN_PLAN_STATS")) fprintf(stderr, "[plan] fit: base %d, fitted max %d, spare %ld, eligible %ld\n", base, max_wave_rows(fit, nwaves, lim), spare, elig);
    // (round 5: taken whatever it leaves spare - the alternative is the 40-row loop with packed arcs)
    if (max_wave_rows(fit, nwaves, lim) <= kResident[1]) { t.gsl = fit; t.fitted = true; return; }
  }
  // (measured on C2 in round 5 and not kept: the 16-row loop for four-wave workgroups with NO spare row - two states on two
  // lanes each, rows 21 -> 16 - leaves the recursion where it was, 0.180 ms: the small frame is not waiting for its chunks)
  int chosen = 0;
  for (int ri = 0; ri < 3 && chosen == 0; ri++) {
    if (base > kResident[ri]) continue;
    for (int k = 4; k >= (ri < 2 ? 2 : 1); k--)
      if (max_wave_rows(with_slack(k), nwaves, lim) <= kResident[ri]) { chosen = k; break; }
  }
  t.gsl = with_slack(chosen);
}


// ---- states on several lanes (round 5) ------------------------------------------------------------------------------
// A group of 64 rows is as long as its longest row, a group cannot be cut between waves, and a wave keeps at most
// PLAN_RESIDENT_2 slot-rows in registers: ONE state with many arcs entering it sets the loop length of the whole
// workgroup (graphs with skewed in-degrees: DESIGN.md §4 "A structured graph").  Such a state is put on several lanes
// without any kernel knowing: in the numbering of a side (alpha: "pa", beta: "pb") it becomes `parts` positions, each of
// which collects a share of its arcs; whoever gathers the state gathers every part - its arc is repeated once per part
// (the recursions are linear in the gathered vector).  What is NOT linear is written into the per-position vectors:
//   alpha'(t,i) = alpha(t,i) + tot(t) coef leaky(i)   the first part carries leaky(i), initial(i); the others 0;
//   beta'(t,i)  = beta(t,i) + c(t)                    only the first part takes the constant: the others are listed in
//                                                     the plan (PlanHeader::off_no_const) and the kernels mark them
//                                                     once, before the first frame (every part carries leaky(i):
//                                                     c(t) = coef sum_i leaky_i beta(t,i) is linear);
//   final(i) multiplies alpha'(L, i): every part carries it; beta(L,i) = final(i): the first part.
// parts are chosen per side by a cap on the row length: the largest cap (fewest repeated arcs) that reaches the
// smallest register-resident loop the tile can reach at all; a side that gains no loop class stays as it is.
struct SideArcs {                       // the arcs of one side's rows: row = the state that owns the sum, other = the state it gathers
  const int32_t* tr; const int32_t* idx; int own_col, other_col;
  int begin(int h) const { return idx[2 * h]; }
  int end(int h) const { return idx[2 * h + 1]; }
  int other(int k) const { return tr[3 * k + other_col]; }
};
// row length of state h when every state s is on parts[s] positions
long virtual_len(const SideArcs& sa, const std::vector<int>& parts, int h) {
  long n = 0;
  for (int k = sa.begin(h); k < sa.end(h); k++) n += parts[sa.other(k)];
  return n;
}
// parts such that no position collects more than `cap` arcs (a fixed point: a state on more positions lengthens the rows
// that gather it); false if it does not settle within sane bounds
bool settle_parts(const SideArcs& sa, int H, int cap, std::vector<int>& parts, long max_arcs) {
  parts.assign(H, 1);
  for (int iter = 0; iter < 16; iter++) {
    bool changed = false;
    long total = 0;
    for (int h = 0; h < H; h++) {
      const long n = virtual_len(sa, parts, h);
      total += n;
      const int need = (int)((n + cap - 1) / cap);
      if (need > parts[h]) { parts[h] = need; changed = true; }
      if (parts[h] > 64) return false;
    }
    if (total > max_arcs) return false;
    if (!changed) return true;
  }
  return false;
}
// lengths of the positions (virtual rows) of a side
std::vector<int> part_lengths(const SideArcs& sa, int H, const std::vector<int>& parts) {
  std::vector<int> len;
  for (int h = 0; h < H; h++) {
    const long n = virtual_len(sa, parts, h);
    for (int m = 0; m < parts[h]; m++) len.push_back((int)((n + parts[h] - 1 - m) / parts[h]));
  }
  return len;
}
// the loop class a recursion tile with these row lengths lands in when dealt to `nwaves` waves (as init_tile would lay it
// out): 0..3 = PLAN_RESIDENT_0 / 24 (four waves only) / _1 / _2 slot-rows in registers, 4 = none of them
int loop_class(const std::vector<int>& len, int nwaves, int npos, int slack) {
  if ((int)len.size() > npos) return 4;
  std::vector<std::vector<Arc>> rows(len.size());
  for (size_t i = 0; i < len.size(); i++) rows[i].resize(len[i]);
  std::vector<int> ids(len.size());
  std::iota(ids.begin(), ids.end(), 0);
  Tile t;
  init_tile(t, rows, sort_by_degree(len, ids), npos, kLayA, kLayX, kLayA, 2, slack, nwaves);
  if (!dealing_fits(t.gsl, nwaves, PLAN_RESIDENT_2, 4)) return 4;
  const int m = max_wave_rows(t.gsl, nwaves, rec_group_limit(t.gsl.size(), nwaves));
  if (getenv("PYCHAIN_PLAN_STATS")) {
    const std::vector<int> g0 = group_rows(rows, t.order, npos, true);
    long tot0 = 0, tot1 = 0; for (int x : g0) tot0 += x; for (int x : t.gsl) tot1 += x;
    fprintf(stderr, "[plan]   %d waves: groups %zu, rows without slack %ld (max wave %d), with %ld (max wave %d); first groups %d %d %d %d\n", nwaves, g0.size(), tot0,
            max_wave_rows(g0, nwaves, rec_group_limit(g0.size(), nwaves)), tot1, m, g0[0], g0.size() > 1 ? g0[1] : 0, g0.size() > 2 ? g0[2] : 0, g0.size() > 3 ? g0[3] : 0);
  }
  if (m <= PLAN_RESIDENT_0) return 0;
  if (m <= 24 && nwaves == PLAN_REC4_WAVES) return 1;
  return m <= PLAN_RESIDENT_1 ? 2 : 3;
}
struct PartsChoice { std::vector<int> parts; int cls = 4; int cap = 0; };
// candidates: caps from the longest row down; per cap the class at 4 and at 16 waves
struct SideSearch {
  std::vector<PartsChoice> c4, c16;          // per candidate cap (descending); index 0 = no state split
};
SideSearch search_side(const SideArcs& sa, int H, int K, int D, int slack) {
  SideSearch out;
  int maxlen = 0;
  for (int h = 0; h < H; h++) maxlen = std::max(maxlen, sa.end(h) - sa.begin(h));
  const int lo = std::max(4, (int)(((long)K + H - 1) / H));         // (the mean row length: below it everything would be on two lanes)
  std::vector<int> caps{maxlen};
  for (int c = maxlen - 1; c >= lo; c -= std::max(1, (maxlen - lo) / 32)) caps.push_back(c);
  for (int cap : caps) {
    PartsChoice pc;
    pc.cap = cap;
    if (!settle_parts(sa, H, cap, pc.parts, K + K / 4 + 64)) {
      if (getenv("PYCHAIN_PLAN_STATS")) fprintf(stderr, "[plan] cap %d: does not settle within 5/4 of the arcs\n", cap);
      break;
    }
    const std::vector<int> len = part_lengths(sa, H, pc.parts);
    const int Hv = (int)len.size(), Hp = (Hv + 63) / 64 * 64;
    if (!pychain_hip::plan_fits_fast_kernels(Hv, D)) break;
    PartsChoice a = pc, b = pc;
    a.cls = (Hp <= 64 * 4 * PLAN_REC4_WAVES && D <= 4096) ? loop_class(len, PLAN_REC4_WAVES, Hp, slack) : 4;
    b.cls = loop_class(len, PLAN_REC_WAVES, Hp, slack);
    out.c4.push_back(a); out.c16.push_back(b);
    if (cap == maxlen) {
      // nothing to gain where the rows as they are already reach the shortest loop K arcs can fit at all (C3: 30 000 arcs on
      // 16 waves are 29.3 slot-rows per wave - the 32-row loop it has): no candidates, 1.6 s of compile time less
      auto floor_class = [&](int nwaves) {
        const long per_wave = ((long)K + 64L * nwaves - 1) / (64L * nwaves);
        return per_wave <= PLAN_RESIDENT_0 ? 0 : (per_wave <= 24 && nwaves == PLAN_REC4_WAVES ? 1 : (per_wave <= PLAN_RESIDENT_1 ? 2 : (per_wave <= PLAN_RESIDENT_2 ? 3 : 4)));
      };
      const bool done16 = b.cls <= floor_class(PLAN_REC_WAVES);
      const bool done4 = a.cls == 4 ? floor_class(PLAN_REC4_WAVES) == 4 : a.cls <= floor_class(PLAN_REC4_WAVES);
      if (done16 && done4) break;
    }
    if (getenv("PYCHAIN_PLAN_STATS")) fprintf(stderr, "[plan] cap %d: %d positions, class at 4 waves %d, at 16 waves %d\n", cap, Hv, a.cls, b.cls);
  }
  return out;
}
// the largest cap (first candidate) whose class is <= target
const PartsChoice* first_within(const std::vector<PartsChoice>& v, int target) {
  for (const PartsChoice& c : v) if (c.cls <= target) return &c;
  return nullptr;
}
int best_class(const std::vector<PartsChoice>& v) { int b = 4; for (const PartsChoice& c : v) b = std::min(b, c.cls); return b; }

size_t align16(size_t x) { return (x + 15) & ~size_t(15); }

// The general format (plan_format.h: GeneralPlanHeader): the reference layout, plus the arcs grouped by pdf-id in the
// (state, arc) order the reference accumulates in (chain-computation.cc:293-305)
int64_t build_general(const int32_t* ft, const int32_t* fi, const float* fp, const int32_t* bt, const int32_t* bi, const float* bp,
                      const float* leaky, const float* initial, const float* final_, int H, int K, int D, int Hp,
                      void* blob, size_t blob_bytes, std::vector<char>* grow) {
  for (int h = 0; h < H; h++) {
    for (int k = bi[2 * h]; k < bi[2 * h + 1]; k++)
      if (bt[3 * k + 1] != h)
        return pychain_hip::fail(PYCHAIN_HIP_EINVAL, "den_plan_build: backward transition %d is not grouped under its destination", k);
    for (int k = fi[2 * h]; k < fi[2 * h + 1]; k++)
      if (ft[3 * k] != h)
        return pychain_hip::fail(PYCHAIN_HIP_EINVAL, "den_plan_build: forward transition %d is not grouped under its source", k);
  }
  GeneralPlanHeader hd;
  memset(&hd, 0, sizeof(hd));
  hd.magic = PLAN_MAGIC_GENERAL; hd.version = PLAN_VERSION; hd.H = H; hd.K = K; hd.D = D; hd.Hp = Hp;
  size_t off = align16(sizeof(hd));
  auto place = [&](int64_t& o, size_t bytes) { o = (int64_t)off; off = align16(off + bytes); };
  place(hd.off_a_idx, (size_t)H * 8); place(hd.off_a_arc, (size_t)K * 8); place(hd.off_a_p, (size_t)K * 4);
  place(hd.off_b_idx, (size_t)H * 8); place(hd.off_b_arc, (size_t)K * 8); place(hd.off_b_p, (size_t)K * 4);
  place(hd.off_g_idx, ((size_t)D + 1) * 4); place(hd.off_g_arc, (size_t)K * 8); place(hd.off_g_p, (size_t)K * 4);
  place(hd.off_leaky, (size_t)Hp * 4); place(hd.off_init, (size_t)Hp * 4); place(hd.off_final, (size_t)Hp * 4);
  hd.total_bytes = (int64_t)off;
  if (grow) { grow->resize(off); blob = grow->data(); blob_bytes = off; }
  if (!blob || blob_bytes < off) return (int64_t)off;
  char* base = (char*)blob;
  memset(base, 0, off);
  memcpy(base + hd.off_a_idx, bi, (size_t)H * 8); memcpy(base + hd.off_b_idx, fi, (size_t)H * 8);
  int32_t* a_arc = (int32_t*)(base + hd.off_a_arc); int32_t* b_arc = (int32_t*)(base + hd.off_b_arc);
  for (int k = 0; k < K; k++) {
    a_arc[2 * k] = bt[3 * k]; a_arc[2 * k + 1] = bt[3 * k + 2];           // {src, pdf}, grouped by destination
    b_arc[2 * k] = ft[3 * k + 1]; b_arc[2 * k + 1] = ft[3 * k + 2];       // {dst, pdf}, grouped by source
  }
  memcpy(base + hd.off_a_p, bp, (size_t)K * 4); memcpy(base + hd.off_b_p, fp, (size_t)K * 4);
  int32_t* g_idx = (int32_t*)(base + hd.off_g_idx); int32_t* g_arc = (int32_t*)(base + hd.off_g_arc); float* g_p = (float*)(base + hd.off_g_p);
  std::vector<int32_t> fill(D + 1, 0);
  for (int k = 0; k < K; k++) fill[ft[3 * k + 2] + 1]++;
  for (int n = 0; n < D; n++) fill[n + 1] += fill[n];
  memcpy(g_idx, fill.data(), ((size_t)D + 1) * 4);
  for (int k = 0; k < K; k++) {                                             // counting sort: stable in (state, arc) order
    const int32_t slot = fill[ft[3 * k + 2]]++;
    g_arc[2 * slot] = ft[3 * k]; g_arc[2 * slot + 1] = ft[3 * k + 1]; g_p[slot] = fp[k];
  }
  memcpy(base + hd.off_leaky, leaky, (size_t)H * 4); memcpy(base + hd.off_init, initial, (size_t)H * 4);
  memcpy(base + hd.off_final, final_, (size_t)H * 4);
  hd.payload_hash = (int32_t)pychain_hip::general_payload_hash(base, off);
  hd.reserved[0] = (int32_t)pychain_hip::general_header_hash(hd);
  memcpy(base, &hd, sizeof(hd));
  return (int64_t)off;
}

}  // namespace

namespace {
// (`grow`: the blob is written into this vector, sized as needed - one compile instead of size + fill)
int64_t plan_build_impl(const int32_t* ft, const int32_t* fi, const float* fp, const int32_t* bt, const int32_t* bi, const float* bp,
                        const float* leaky, const float* initial, const float* final_, int H, int K, int D, void* blob, size_t blob_bytes,
                        std::vector<char>* grow = nullptr);
// The contract is "call once to size, once to fill": both calls would compile the plan (seconds of annealing for a
// C3-size graph).  The sizing call keeps what it built, keyed by a hash of every input byte and the knobs; the fill call
// with the same inputs copies it.  One entry per host thread.
struct LastPlan { uint64_t key = 0; std::vector<char> blob; };
uint64_t fnv64(uint64_t h, const void* p, size_t n) {
  const unsigned char* c = (const unsigned char*)p;
  for (size_t i = 0; i < n; i++) { h ^= c[i]; h *= 1099511628211ull; }
  return h;
}
}  // namespace

extern "C" int64_t pychain_hip_den_plan_build(
    const int32_t* ft, const int32_t* fi, const float* fp,
    const int32_t* bt, const int32_t* bi, const float* bp,
    const float* leaky, const float* initial, const float* final_,
    int H, int K, int D, void* blob, size_t blob_bytes) {
  if (!ft || !fi || !fp || !bt || !bi || !bp || !leaky || !initial || !final_ || H <= 0 || K <= 0 || D <= 0)
    return plan_build_impl(ft, fi, fp, bt, bi, bp, leaky, initial, final_, H, K, D, blob, blob_bytes);   // (reports the error)
  static thread_local LastPlan last;
  uint64_t key = 14695981039346656037ull;
  const int dims[4] = {H, K, D, pychain_hip::call_knobs().plan_split};
  key = fnv64(key, dims, sizeof(dims));
  key = fnv64(key, ft, (size_t)K * 12); key = fnv64(key, fi, (size_t)H * 8); key = fnv64(key, fp, (size_t)K * 4);
  key = fnv64(key, bt, (size_t)K * 12); key = fnv64(key, bi, (size_t)H * 8); key = fnv64(key, bp, (size_t)K * 4);
  key = fnv64(key, leaky, (size_t)H * 4); key = fnv64(key, initial, (size_t)H * 4); key = fnv64(key, final_, (size_t)H * 4);
  for (const char* knob : {"PYCHAIN_PLAN_GENERAL", "PYCHAIN_PLAN_SLACK", "PYCHAIN_PLAN_BALANCE", "PYCHAIN_PLAN_ANNEAL", "PYCHAIN_PLAN_FIT",
                           "PYCHAIN_PLAN_LINEAR", "PYCHAIN_PLAN_SPLIT", "PYCHAIN_PLAN_GAMMA_BOUND", "PYCHAIN_PLAN_SG"}) {
    const char* v = getenv(knob);
    key = fnv64(key, knob, strlen(knob));
    if (v) key = fnv64(key, v, strlen(v));
  }
  if (last.key != key || last.blob.empty()) {
    std::vector<char> fresh;
    const int64_t rc = plan_build_impl(ft, fi, fp, bt, bi, bp, leaky, initial, final_, H, K, D, nullptr, 0, &fresh);
    if (rc < 0) return rc;
    last.key = key; last.blob.swap(fresh);
  }
  const int64_t need = (int64_t)last.blob.size();
  if (blob && blob_bytes >= (size_t)need) {
    memcpy(blob, last.blob.data(), (size_t)need);
    LastPlan().blob.swap(last.blob);                   // handed over: the (possibly large) copy is not kept
    last.key = 0;
  }
  return need;
}

namespace {
int64_t plan_build_impl(
    const int32_t* ft, const int32_t* fi, const float* fp,
    const int32_t* bt, const int32_t* bi, const float* bp,
    const float* leaky, const float* initial, const float* final_,
    int H, int K, int D, void* blob, size_t blob_bytes, std::vector<char>* grow) {
  if (!ft || !fi || !fp || !bt || !bi || !bp || !leaky || !initial || !final_)
    return pychain_hip::fail(PYCHAIN_HIP_EINVAL, "den_plan_build: null graph pointer");
  if (H <= 0 || K <= 0 || D <= 0)
    return pychain_hip::fail(PYCHAIN_HIP_EINVAL, "den_plan_build: empty graph (H=%d K=%d D=%d)", H, K, D);
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
  // graphs the tile-plan kernels do not take (or PYCHAIN_PLAN_GENERAL=1: the tests): the general format
  if (!pychain_hip::plan_fits_fast_kernels(H, D) || env_long("PYCHAIN_PLAN_GENERAL", 0) != 0)
    return build_general(ft, fi, fp, bt, bi, bp, leaky, initial, final_, H, K, D, (H + 63) / 64 * 64, blob, blob_bytes, grow);
  for (int h = 0; h < H; h++) {
    for (int k = bi[2 * h]; k < bi[2 * h + 1]; k++)
      if (bt[3 * k + 1] != h)
        return pychain_hip::fail(PYCHAIN_HIP_EINVAL, "den_plan_build: backward transition %d is not grouped under its destination", k);
    for (int k = fi[2 * h]; k < fi[2 * h + 1]; k++)
      if (ft[3 * k] != h)
        return pychain_hip::fail(PYCHAIN_HIP_EINVAL, "den_plan_build: forward transition %d is not grouped under its source", k);
  }
  // tuning knobs (environment): PYCHAIN_PLAN_SLACK spare slot-rows per recursion group (default: automatic), PYCHAIN_PLAN_BALANCE
  // row-placement moves per transition (0 = rows stay in degree order), PYCHAIN_PLAN_ANNEAL slot moves per cell,
  // PYCHAIN_PLAN_SPLIT=0: no state on more than one lane
  const int slack = (int)env_long("PYCHAIN_PLAN_SLACK", -1);
  const long balance_moves = env_long("PYCHAIN_PLAN_BALANCE", 170);
  const long anneal_knob = env_long("PYCHAIN_PLAN_ANNEAL", -1);        // < 0: 200, and 3000 for a recursion tile fitted to the 32-row loop
  const bool stats = env_long("PYCHAIN_PLAN_STATS", 0) != 0;

  // ---- states on several lanes ("states on several lanes" above): parts per state and side
  const SideArcs side_a{bt, bi, 1, 0}, side_b{ft, fi, 0, 1};
  std::vector<int> parts_a(H, 1), parts_b(H, 1);
  const int split_knob = pychain_hip::call_knobs().plan_split;      // (option plan_split, else the environment)
  if ((split_knob >= 0 ? split_knob : (int)env_long("PYCHAIN_PLAN_SPLIT", 1)) != 0) {
    const SideSearch sa = search_side(side_a, H, K, D, slack), sb = search_side(side_b, H, K, D, slack);
    if (!sa.c16.empty() && !sb.c16.empty()) {
      // four-wave workgroups if both sides can be made to fit them, else sixteen; the loop is as long as the longer side's
      const bool four = best_class(sa.c4) <= 3 && best_class(sb.c4) <= 3;
      const std::vector<PartsChoice>& va = four ? sa.c4 : sa.c16;
      const std::vector<PartsChoice>& vb = four ? sb.c4 : sb.c16;
      const int target = std::max(best_class(va), best_class(vb));
      const int unsplit = std::max(va[0].cls, vb[0].cls);
      if (target < unsplit) {
        const PartsChoice* ca = first_within(va, target);
        const PartsChoice* cb = first_within(vb, target);
        if (ca && cb) { parts_a = ca->parts; parts_b = cb->parts; }
        if (stats) fprintf(stderr, "[plan] states on several lanes: loop class %d -> %d (%s waves), cap alpha %d beta %d\n", unsplit, target,
                           four ? "4" : "16", ca ? ca->cap : -1, cb ? cb->cap : -1);
        // The occupancy tile gets an arc once per (alpha position of its source, beta position of its destination): the 5/4
        // bound of settle_parts holds per recursion side only, and an arc between two hub states - or a hub's self-loop - is
        // repeated parts_a x parts_b times there.  Bounded like the sides: at most 3/2 of the arcs; beyond that
        // the graph keeps every state on one lane.
        auto gamma_arcs = [&]() { long n = 0; for (int k = 0; k < K; k++) n += (long)parts_a[ft[3 * k]] * parts_b[ft[3 * k + 1]]; return n; };
        // (PYCHAIN_PLAN_GAMMA_BOUND: the bound in percent of the arcs, default 150 - the sides' own 5/4 bounds keep real graphs
        // under about 2 K, so the tests lower it to see the rule act)
        const long gbound = (long)K * env_long("PYCHAIN_PLAN_GAMMA_BOUND", 150) / 100 + 64;
        if (gamma_arcs() > gbound) {                          // (the loop is as long as the longer side's: one side alone gains nothing)
          parts_a.assign(H, 1); parts_b.assign(H, 1);
          if (stats) fprintf(stderr, "[plan] states on several lanes: the occupancy tile would repeat too many arcs - split given up\n");
        }
      }
    }
  }
  // positions ("entities") of a side: state h owns ea0[h] .. ea0[h + 1] - 1 in the alpha numbering, eb0 likewise
  std::vector<int> ea0(H + 1, 0), eb0(H + 1, 0);
  for (int h = 0; h < H; h++) { ea0[h + 1] = ea0[h] + parts_a[h]; eb0[h + 1] = eb0[h] + parts_b[h]; }
  const int HA = ea0[H], HB = eb0[H], HV = std::max(HA, HB);
  const int Hp = (HV + 63) / 64 * 64;
  std::vector<int> ids_a(HA), ids_b(HB);
  std::iota(ids_a.begin(), ids_a.end(), 0);
  std::iota(ids_b.begin(), ids_b.end(), 0);

  // rows by entity ids.  alpha rows: arcs entering h, in the reference's order (fstext.cc:63-76); an arc is repeated once per
  // part of the state it gathers, and the arcs of a state on several positions are dealt to them in turn
  std::vector<std::vector<Arc>> rows_a(HA), rows_b(HB), rows_g(D);
  for (int h = 0; h < H; h++) {
    int n = 0;
    for (int k = bi[2 * h]; k < bi[2 * h + 1]; k++)
      for (int m = 0; m < parts_a[bt[3 * k]]; m++, n++)
        rows_a[ea0[h] + n % parts_a[h]].push_back(Arc{ea0[bt[3 * k]] + m, bt[3 * k + 2], bp[k]});
    n = 0;
    for (int k = fi[2 * h]; k < fi[2 * h + 1]; k++) {
      const int dst = ft[3 * k + 1];
      for (int m = 0; m < parts_b[dst]; m++, n++)
        rows_b[eb0[h] + n % parts_b[h]].push_back(Arc{eb0[dst] + m, ft[3 * k + 2], fp[k]});
      // gamma rows keep the (state, arc) order the reference accumulates in (chain-computation.cc:293-305)
      for (int ma = 0; ma < parts_a[h]; ma++)
        for (int mb = 0; mb < parts_b[dst]; mb++)
          rows_g[ft[3 * k + 2]].push_back(Arc{ea0[h] + ma, eb0[dst] + mb, fp[k]});
    }
  }
  std::vector<int> indeg(HA), outdeg(HB);
  for (int i = 0; i < HA; i++) indeg[i] = (int)rows_a[i].size();
  for (int i = 0; i < HB; i++) outdeg[i] = (int)rows_b[i].size();
  std::vector<int> gdeg(D), gids;
  for (int n = 0; n < D; n++) { gdeg[n] = (int)rows_g[n].size(); if (gdeg[n] > 0) gids.push_back(n); }
  const int gpos = std::max(((int)gids.size() + 63) / 64 * 64, 64);


  // The occupancy tiles take no slack: their kernels are not bound by gather cycles and the two-frame
  // kernel keeps exactly 64 slot-rows per wave in registers.
  std::vector<Tile> tiles(3);
  // A graph whose recursion tiles fit FOUR waves (<= PLAN_RESIDENT_2 slot-rows and <= 4 groups per wave: a few hundred states,
  // a few thousand arcs) gets its slack fitted to that dealing and carries it as alpha4 / beta4: its recursions then run in
  // 256-thread workgroups (den_lazy.inc.h: LzSmall) instead of sixteen waves meeting at a barrier for four groups of work.
  const std::vector<int> order_a = sort_by_degree(indeg, ids_a), order_b = sort_by_degree(outdeg, ids_b);
  bool small = Hp <= 64 * 4 * PLAN_REC4_WAVES && D <= 4096 &&
               dealing_fits(group_rows(rows_a, order_a, Hp, true), PLAN_REC4_WAVES, PLAN_RESIDENT_2, 4) &&
               dealing_fits(group_rows(rows_b, order_b, Hp, true), PLAN_REC4_WAVES, PLAN_RESIDENT_2, 4);
  const int rec_waves = small ? PLAN_REC4_WAVES : PLAN_REC_WAVES;
  init_tile(tiles[0], rows_a, order_a, Hp, kLayA, kLayX, kLayA, 2, slack, rec_waves);   // the recursions are the critical path
  init_tile(tiles[1], rows_b, order_b, Hp, kLayB, kLayX, kLayB, 2, slack, rec_waves);
  small = small && dealing_fits(tiles[0].gsl, PLAN_REC4_WAVES, PLAN_RESIDENT_2, 4) && dealing_fits(tiles[1].gsl, PLAN_REC4_WAVES, PLAN_RESIDENT_2, 4);
  init_tile(tiles[2], rows_g, sort_by_degree(gdeg, gids), gpos, kLayA, kLayB, -1, 1, 0, PLAN_GAM_WAVES);
  Layouts lay;
  lay.pos[kLayA].assign(HA, 0); lay.pos[kLayB].assign(HB, 0); lay.pos[kLayX].resize(D);
  for (int i = 0; i < HA; i++) lay.pos[kLayA][tiles[0].order[i]] = i;
  for (int i = 0; i < HB; i++) lay.pos[kLayB][tiles[1].order[i]] = i;
  std::iota(lay.pos[kLayX].begin(), lay.pos[kLayX].end(), 0);
  {
    Balancer bal(tiles, lay);
    const long before = stats ? bal.overload() : 0;
    bal.run(balance_moves * K, 40.0, 0.5);
    if (stats) fprintf(stderr, "[plan] row placement: bank overload %ld -> %ld (of %ld arc operands)\n", before, bal.overload(), 6L * K);
  }
  // (the state vectors of the recursion tiles are float2 in the lazy kernels: ds_read_b64; the occupancy tiles are
  // gathered with one width for both operands)
  SlotOrder so_a(tiles[0], lay, true, false), so_b(tiles[1], lay, true, false), so_g(tiles[2], lay, true, true);
  // "pdf by state" (plan_format.h: PLAN_FLAG_PDF_BY_STATE): every arc entering a state carries one pdf.  PYCHAIN_PLAN_SG=0: not
  // marked (the tests compare the two kernel families on one graph).
  std::vector<int> in_pdf(H, 0);
  bool pdf_by_state = env_long("PYCHAIN_PLAN_SG", 1) != 0 && D <= 4096;
  for (int h = 0; h < H && pdf_by_state; h++)
    for (int k = bi[2 * h]; k < bi[2 * h + 1]; k++) {
      if (k == bi[2 * h]) in_pdf[h] = bt[3 * k + 2];
      else if (bt[3 * k + 2] != in_pdf[h]) { pdf_by_state = false; break; }
    }
  if (stats) fprintf(stderr, "[plan] pdf by state (one gather per arc): %s\n", pdf_by_state ? "yes" : "no");
  so_a.ignore_op1 = so_b.ignore_op1 = pdf_by_state;
  auto moves = [&](const Tile& t) { return anneal_knob >= 0 ? anneal_knob : (t.fitted ? 3000L : 200L); };
  {
    // the three tiles are independent (and every group has its own random stream: the result does not depend on the threads)
    std::thread tb([&]() { so_b.run(moves(tiles[1])); }), tg([&]() { so_g.run(moves(tiles[2])); });
    so_a.run(moves(tiles[0]));
    tb.join(); tg.join();
  }
  if (stats)
    fprintf(stderr, "[plan] modelled LDS cycles per half slot-row (2.0 = conflict-free): alpha %.3f beta %.3f gamma %.3f; "
                    "half slot-rows %ld %ld %ld\n", (double)so_a.cycles / std::max(1L, so_a.columns.load()),
            (double)so_b.cycles / std::max(1L, so_b.columns.load()), (double)so_g.cycles / std::max(1L, so_g.columns.load()),
            so_a.columns.load(), so_b.columns.load(), so_g.columns.load());
  if (stats)
    fprintf(stderr, "[plan] modelled LDS-busy cycles per frame (gfx950 gather costs): alpha %.0f (%ld slot-rows, conflicts %.0f) beta %.0f (%ld, %.0f)\n",
            so_a.lds_cycles(), so_a.slot_rows.load(), 0.1 * so_a.cost_tenths.load(), so_b.lds_cycles(), so_b.slot_rows.load(),
            0.1 * so_b.cost_tenths.load());
  if (stats)
    fprintf(stderr, "[plan] ... per operand (1.0 = conflict-free): alpha state %.3f nnet-output %.3f; beta state %.3f nnet-output %.3f\n",
            (double)so_a.cycles_op[0] / std::max(1L, so_a.columns.load()), (double)so_a.cycles_op[1] / std::max(1L, so_a.columns.load()),
            (double)so_b.cycles_op[0] / std::max(1L, so_b.columns.load()), (double)so_b.cycles_op[1] / std::max(1L, so_b.columns.load()));

  const auto deal_a = deal_groups(tiles[0].gsl, PLAN_REC_WAVES, rec_group_limit(tiles[0].gsl.size(), PLAN_REC_WAVES));
  const auto deal_b = deal_groups(tiles[1].gsl, PLAN_REC_WAVES, rec_group_limit(tiles[1].gsl.size(), PLAN_REC_WAVES));
  BuiltTile ta = emit_tile(tiles[0], so_a, lay, deal_a);
  BuiltTile tb = emit_tile(tiles[1], so_b, lay, deal_b);
  BuiltTile tg = emit_tile(tiles[2], so_g, lay, deal_groups(tiles[2].gsl, PLAN_GAM_WAVES));
  BuiltTile tg2 = emit_tile(tiles[2], so_g, lay, deal_groups(tiles[2].gsl, PLAN_GAM2_WAVES));
  // the recursion tiles once more for four-wave workgroups, where the whole graph fits them (a wave keeps at most
  // PLAN_RESIDENT_2 slot-rows in registers and four groups: den_lazy.inc.h, LzSmall)
  // "pdf by state": the occupancy tiles over states instead of arcs (plan_format.h: gamma_sg) - one pseudo-arc per alpha position
  // of a state that has arcs entering it: {that alpha position, the state's beta position, 1}, in the row of the state's pdf
  BuiltTile tgs, tgs2;
  std::vector<std::vector<Arc>> rows_gs(D);
  Tile tile_gs;
  if (pdf_by_state) {
    if (HB != H) pdf_by_state = false;                      // (a state on several BETA positions: the kernels' NC form has no one-gather variant)
    if (HA - H > PLAN_MAX_EXTRA_A) pdf_by_state = false;    // (the crossing keeps a state's further ALPHA positions in a fixed table: den_lazy.inc.h, LzCross)
    for (int h = 0; h < H && pdf_by_state; h++)
      if (bi[2 * h + 1] > bi[2 * h])
        for (int m = 0; m < parts_a[h]; m++) rows_gs[in_pdf[h]].push_back(Arc{ea0[h] + m, eb0[h], 1.0f});
  }
  if (pdf_by_state) {
    std::vector<int> gsdeg(D), gsids;
    for (int n = 0; n < D; n++) { gsdeg[n] = (int)rows_gs[n].size(); if (gsdeg[n] > 0) gsids.push_back(n); }
    const int gspos = std::max(((int)gsids.size() + 63) / 64 * 64, 64);
    init_tile(tile_gs, rows_gs, sort_by_degree(gsdeg, gsids), gspos, kLayA, kLayB, -1, 1, 0, PLAN_GAM_WAVES);
    SlotOrder so_gs(tile_gs, lay, false, false);
    so_gs.run(50);
    tgs = emit_tile(tile_gs, so_gs, lay, deal_groups(tile_gs.gsl, PLAN_GAM_WAVES));
    tgs2 = emit_tile(tile_gs, so_gs, lay, deal_groups(tile_gs.gsl, PLAN_GAM2_WAVES));
    if (tgs.max_wave > PLAN_RESIDENT_0 || tgs2.max_wave > PLAN_RESIDENT_0) pdf_by_state = false;   // (one pdf shared by hundreds of states: not the shape this is for)
    if (stats) fprintf(stderr, "[plan] occupancy tile over states: %d slot-rows (over arcs: %d), max per wave %d / %d\n", tgs.total_slot_rows, tg.total_slot_rows, tgs.max_wave, tgs2.max_wave);
  }
  BuiltTile ta4, tb4;
  if (small) {
    ta4 = emit_tile(tiles[0], so_a, lay, deal_groups(tiles[0].gsl, PLAN_REC4_WAVES, rec_group_limit(tiles[0].gsl.size(), PLAN_REC4_WAVES)));
    tb4 = emit_tile(tiles[1], so_b, lay, deal_groups(tiles[1].gsl, PLAN_REC4_WAVES, rec_group_limit(tiles[1].gsl.size(), PLAN_REC4_WAVES)));
  }

  // ---- lay the blob out
  size_t off = align16(sizeof(PlanHeader));
  PlanHeader hd;
  memset(&hd, 0, sizeof(hd));
  hd.magic = PLAN_MAGIC; hd.version = PLAN_VERSION;
  // H = positions of the longer side (what the caller sizes its workspace by); reserved: the graph's states, states on
  // several positions per side
  hd.H = HV; hd.K = K; hd.D = D; hd.Hp = Hp;
  hd.graph_states = H; hd.n_no_const = HB - H;
  for (const BuiltTile* t : {&ta, &tb})
    for (const WaveEntry& we : t->waves) hd.rec_max_wave_groups = std::max(hd.rec_max_wave_groups, we.ngroups);
  if (small)
    for (const BuiltTile* t : {&ta4, &tb4})
      for (const WaveEntry& we : t->waves) hd.rec4_max_wave_groups = std::max(hd.rec4_max_wave_groups, we.ngroups);
  auto place_tile = [&](TilePlan& tp, const BuiltTile& t) {
    tp.ngroups = (int)t.groups.size(); tp.nwaves = (int)t.waves.size();
    tp.total_slot_rows = t.total_slot_rows; tp.max_wave_slot_rows = t.max_wave; tp.nrows = t.nrows;
    tp.off_wave_tab = (int32_t)off; off = align16(off + t.waves.size() * sizeof(WaveEntry));
    tp.off_group_tab = (int32_t)off; off = align16(off + std::max<size_t>(1, t.groups.size()) * sizeof(GroupEntry));
    tp.off_slots = (int32_t)off; off = align16(off + std::max<size_t>(1, t.slots.size()) * 4);
  };
  place_tile(hd.alpha, ta); place_tile(hd.beta, tb); place_tile(hd.gamma, tg); place_tile(hd.gamma2, tg2);
  if (small) { place_tile(hd.alpha4, ta4); place_tile(hd.beta4, tb4); }
  auto place_vec = [&](int32_t& o, size_t n) { o = (int32_t)off; off = align16(off + n * 4); };
  place_vec(hd.off_init_a, Hp); place_vec(hd.off_leaky_a, Hp); place_vec(hd.off_final_a, Hp);
  place_vec(hd.off_leaky_b, Hp); place_vec(hd.off_final_b, Hp);
  place_vec(hd.off_row_pdf, gpos);
  place_vec(hd.off_no_const, std::max(1, HB - H));
  if (pdf_by_state) {
    hd.flags |= PLAN_FLAG_PDF_BY_STATE; place_vec(hd.off_pdf_a, Hp); place_vec(hd.off_pdf_b, Hp);
    place_tile(hd.gamma_sg, tgs); place_tile(hd.gamma2_sg, tgs2);
    place_vec(hd.off_row_pdf_sg, std::max((int)tgs.groups.size() * 64, 64));
    place_vec(hd.off_a2b, Hp); place_vec(hd.off_b2a, Hp);
    hd.n_extra_a = HA - H;
    place_vec(hd.off_extra_a, std::max(2, 2 * (HA - H)));
  }
  if (off > (size_t)INT32_MAX)
    return pychain_hip::fail(PYCHAIN_HIP_EUNSUPPORTED, "den_plan_build: plan larger than 2 GiB");
  hd.total_bytes = (int32_t)off;
  if (grow) { grow->resize(off); blob = grow->data(); blob_bytes = off; }
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
  if (small) { write_tile(hd.alpha4, ta4); write_tile(hd.beta4, tb4); }
  float* init_a = (float*)(base + hd.off_init_a); float* leaky_a = (float*)(base + hd.off_leaky_a);
  float* final_a = (float*)(base + hd.off_final_a); float* leaky_b = (float*)(base + hd.off_leaky_b);
  float* final_b = (float*)(base + hd.off_final_b); int32_t* row_pdf = (int32_t*)(base + hd.off_row_pdf);
  int32_t* no_const = (int32_t*)(base + hd.off_no_const);
  int n_nc = 0;
  for (int h = 0; h < H; h++) {
    for (int m = 0; m < parts_a[h]; m++) {
      const int pa = lay.pos[kLayA][ea0[h] + m];
      init_a[pa] = m == 0 ? initial[h] : 0.f; leaky_a[pa] = m == 0 ? leaky[h] : 0.f; final_a[pa] = final_[h];
    }
    for (int m = 0; m < parts_b[h]; m++) {
      const int pb = lay.pos[kLayB][eb0[h] + m];
      leaky_b[pb] = leaky[h]; final_b[pb] = m == 0 ? final_[h] : 0.f;
      if (m > 0) no_const[n_nc++] = pb;                  // (beta' = beta + c(t): the constant once per state)
    }
  }
  for (int i = 0; i < gpos; i++) row_pdf[i] = i < (int)tiles[2].order.size() ? tiles[2].order[i] : -1;
  if (pdf_by_state) {
    write_tile(hd.gamma_sg, tgs); write_tile(hd.gamma2_sg, tgs2);
    int32_t* row_pdf_sg = (int32_t*)(base + hd.off_row_pdf_sg);
    for (int i = 0; i < std::max((int)tgs.groups.size() * 64, 64); i++) row_pdf_sg[i] = i < (int)tile_gs.order.size() ? tile_gs.order[i] : -1;
    int32_t* a2b = (int32_t*)(base + hd.off_a2b); int32_t* b2a = (int32_t*)(base + hd.off_b2a); int32_t* extra = (int32_t*)(base + hd.off_extra_a);
    int n_ex = 0;
    for (int h = 0; h < H; h++) {                            // (HB == H: one beta position per state)
      const int pb = lay.pos[kLayB][eb0[h]];
      b2a[pb] = lay.pos[kLayA][ea0[h]];
      for (int m = 0; m < parts_a[h]; m++) {
        a2b[lay.pos[kLayA][ea0[h] + m]] = pb;
        if (m > 0) { extra[2 * n_ex] = pb; extra[2 * n_ex + 1] = lay.pos[kLayA][ea0[h] + m]; n_ex++; }
      }
    }
    int32_t* pdf_a = (int32_t*)(base + hd.off_pdf_a); int32_t* pdf_b = (int32_t*)(base + hd.off_pdf_b);
    for (int h = 0; h < H; h++) {
      for (int m = 0; m < parts_a[h]; m++) pdf_a[lay.pos[kLayA][ea0[h] + m]] = in_pdf[h];
      for (int m = 0; m < parts_b[h]; m++) pdf_b[lay.pos[kLayB][eb0[h] + m]] = in_pdf[h];
    }
  }
  // integrity of everything behind the header: the kernels follow the blob's offsets and packed LDS addresses
  // unchecked, so a plan that comes back from a cache file is verified first (pychain_hip_den_plan_info)
  PlanHeader* out_hd = reinterpret_cast<PlanHeader*>(base);
  out_hd->payload_hash = (int32_t)pychain_hip::plan_payload_hash(base, off);
  out_hd->header_hash = (int32_t)pychain_hip::plan_header_hash(*out_hd);
  return (int64_t)off;
}
}  // namespace
