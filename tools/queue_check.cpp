// queue_check.cpp -- plays the tile-queue protocol of the pooled kernel on the CPU.
//
// The kernel's ticket arithmetic lives in rt_device.hpp as __host__ __device__ functions (shard_of, shard_tile,
// shard_tickets, ticket_span, queue_draw); this program drives exactly those with W emulated waves in a random
// interleaving and checks, for many random launch shapes, that
//   * every pixel slot of every position (tile x frame) is handed out exactly once,
//   * positions map to tiles one-to-one (the strips partition the tile grid),
//   * every wave ends with all shards seen dry, having failed at most once per counter,
//   * no counter is touched for a shard whose tickets are all static.
// Host-only (no GPU needed); part of the CPU test suite (tests/test_host_logic.py).
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "rt_device.hpp"

using namespace rtk;

struct Cfg {
  int tiles_x, tiles_y, ns_log2, nframes, ds, tpt, deep_class, waves, static_first, interleave;
  int cap_log2 = 5;
};

static int run(const Cfg &c, std::mt19937 &rng, bool verbose) {
  const int ns = 1 << c.ns_log2;                 // counters
  const int gl2 = c.interleave ? 0 : c.ns_log2;  // log2 of the shards of the GEOMETRY (strips, order table)
  const int gs = 1 << gl2;
  const int ntiles = c.tiles_x * c.tiles_y;
  // a synthetic order table: the identity within each strip + class tables with random deep counts
  std::vector<int> order(static_cast<size_t>(order_table_ints(ntiles)), 0);
  std::vector<int> seen_tile(static_cast<size_t>(ntiles), 0);
  for (int s = 0; s < gs; ++s) {
    const Shard sh = shard_of(s, gl2, c.tiles_x, c.tiles_y);
    for (int k = 0; k < sh.ntiles; ++k) {
      const int tile = shard_tile(sh, k, c.tiles_x);
      if (tile < 0 || tile >= ntiles) { std::printf("tile out of range\n"); return 1; }
      order[static_cast<size_t>(sh.seg + k)] = tile;
      seen_tile[static_cast<size_t>(tile)]++;
    }
    int *tab = order.data() + ntiles + kOrderTableDw * s;
    int acc = 0;
    for (int cl = 0; cl < 8; ++cl) {          // non-decreasing class starts, the last one the tile count
      tab[cl] = acc;
      if (sh.ntiles > acc) acc += static_cast<int>(rng() % static_cast<unsigned>(sh.ntiles - acc + 1)) / (cl < 3 ? 8 : 2);
    }
    tab[0] = 0;
    tab[8] = sh.ntiles;
  }
  for (int t = 0; t < ntiles; ++t)
    if (seen_tile[static_cast<size_t>(t)] != 1) { std::printf("tile %d in %d strips\n", t, seen_tile[static_cast<size_t>(t)]); return 1; }

  QueueConst qc;
  qc.ns_log2 = c.ns_log2; qc.tiles_x = c.tiles_x; qc.tiles_y = c.tiles_y; qc.nframes = c.nframes;
  qc.ds = c.ds; qc.tpt = c.tpt; qc.cap_log2 = c.cap_log2; qc.ntiles = ntiles; qc.interleave = c.interleave;
  const bool deep_on = c.nframes == 1 && c.deep_class > 0;
  qc.order = deep_on ? order.data() : nullptr; qc.deep_class = c.deep_class;
  qc.px = nullptr;
  qc.home_waves = static_cast<unsigned>(c.waves >> c.ns_log2);
  qc.q_static = c.static_first ? qc.home_waves : 0u;

  std::vector<unsigned> counter(static_cast<size_t>(ns), 0u), draws(static_cast<size_t>(ns), 0u);
  std::vector<unsigned char> cover(static_cast<size_t>(ntiles) * c.nframes * 64, 0);
  // waves: workgroup b = w / wpw, wave-in-workgroup = w % wpw; wpw = 4, grid = waves / 4 (a multiple of ns)
  const int wpw = 4, grid = c.waves / wpw;
  struct Wave { unsigned state; bool done; unsigned rank; std::vector<int> fails; };
  std::vector<Wave> wv(static_cast<size_t>(c.waves));
  for (int w = 0; w < c.waves; ++w) {
    const int b = w / wpw, wi = w % wpw;
    wv[static_cast<size_t>(w)].state = queue_state_init(b & (ns - 1), c.static_first != 0);
    wv[static_cast<size_t>(w)].done = false;
    wv[static_cast<size_t>(w)].rank = static_cast<unsigned>(wi) * static_cast<unsigned>(grid >> c.ns_log2) + static_cast<unsigned>(b >> c.ns_log2);
    wv[static_cast<size_t>(w)].fails.assign(static_cast<size_t>(ns), 0);
  }
  std::vector<int> alive(static_cast<size_t>(c.waves));
  for (int w = 0; w < c.waves; ++w) alive[static_cast<size_t>(w)] = w;
  unsigned long long tickets_taken = 0;
  while (!alive.empty()) {
    const size_t pick = rng() % alive.size();
    Wave &W = wv[static_cast<size_t>(alive[pick])];
    TicketSpan sp;
    const bool got = queue_draw(W.state, qc, W.rank, [&](int shard) {
      draws[static_cast<size_t>(shard)]++;
      return counter[static_cast<size_t>(shard)]++;
    }, &sp);
    if (!got) {
      if (((W.state >> 8) & 0xffu) != ((1u << ns) - 1u)) { std::printf("wave left with live shards\n"); return 1; }
      alive[pick] = alive.back();
      alive.pop_back();
      continue;
    }
    tickets_taken++;
    if (sp.q_end <= sp.q_next || sp.q_end > cover.size()) { std::printf("bad span [%u, %u) of %zu\n", sp.q_next, sp.q_end, cover.size()); return 1; }
    // the span lies in the shard the wave is drawing from
    const Shard sh = shard_of(queue_geo_shard(qc, W.state), gl2, c.tiles_x, c.tiles_y);
    if ((sp.q_next >> 6) < static_cast<unsigned>(sh.seg) * 1u || ((sp.q_end - 1) >> 6) >= static_cast<unsigned>(sh.seg + sh.ntiles * c.nframes)) {
      std::printf("span outside its shard\n");
      return 1;
    }
    for (unsigned i = sp.q_next; i < sp.q_end; ++i) {
      if (cover[i]) { std::printf("pixel slot %u handed out twice\n", i); return 1; }
      cover[i] = 1;
    }
  }
  for (size_t i = 0; i < cover.size(); ++i)
    if (!cover[i]) { std::printf("pixel slot %zu never handed out (cfg %d x %d ns %d nf %d ds %d tpt %d dc %d W %d sf %d)\n", i, c.tiles_x, c.tiles_y, ns, c.nframes, c.ds, c.tpt, c.deep_class, c.waves, c.static_first); return 1; }
  // counter draws: successes + at most one failure per wave per shard; none on a shard without dynamic tickets
  for (int s = 0; s < ns; ++s) {
    const int geo = c.interleave ? 0 : s;
    const Shard sh = shard_of(geo, gl2, c.tiles_x, c.tiles_y);
    const int ndeep = queue_ndeep(qc, geo);
    const unsigned n_split = queue_nsplit(qc, ndeep);
    unsigned tk = shard_tickets(static_cast<unsigned>(sh.ntiles) * c.nframes, n_split, static_cast<unsigned>(ndeep), c.ds, c.tpt);
    if (c.interleave) tk = (tk + static_cast<unsigned>(ns) - 1u - static_cast<unsigned>(s)) >> c.ns_log2;
    const unsigned dyn = tk > qc.q_static ? tk - qc.q_static : 0u;
    if (dyn == 0 && draws[static_cast<size_t>(s)] != 0) { std::printf("counter of an all-static shard was drawn from\n"); return 1; }
    if (dyn != 0 && (draws[static_cast<size_t>(s)] < dyn || draws[static_cast<size_t>(s)] > dyn + static_cast<unsigned>(c.waves))) {
      std::printf("shard %d: %u draws for %u dynamic tickets and %d waves\n", s, draws[static_cast<size_t>(s)], dyn, c.waves);
      return 1;
    }
  }
  if (verbose) std::printf("ok: %dx%d tiles, %d counter(s)%s, %d frame(s), ds %d, tpt %d, %d waves, static %d: %llu tickets\n", c.tiles_x, c.tiles_y, ns,
                           c.interleave ? " taking turns" : (ns > 1 ? ", a strip each" : ""), c.nframes, c.ds, c.tpt, c.waves, c.static_first, tickets_taken);
  return 0;
}

// Pixel tickets (rt_device.hpp: px_make_header / px_ticket_span through queue_draw): a random list of `npix` pixels cut into the
// five classes at random positions; every list position is handed out exactly once, a ticket of class k covers at most
// 1 << px_log2(k) positions of class k's segment, and every ticket of a class but its last one is full.
static int run_px(std::mt19937 &rng, bool verbose) {
  const int npix = 1 + static_cast<int>(rng() % 20000);
  int pos[kPxClasses + 1];
  pos[0] = 0;
  pos[kPxClasses] = npix;
  for (int k = 1; k < kPxClasses; ++k) pos[k] = static_cast<int>(rng() % static_cast<unsigned>(npix + 1)) / ((rng() & 1) ? 1 : 16);
  for (int k = 1; k < kPxClasses; ++k) if (pos[k] < pos[k - 1]) pos[k] = pos[k - 1];
  int hdr[kPxHdrInts];
  px_make_header(pos, hdr);
  hdr[7] = static_cast<int>(rng() & 1);      // the bulk zipped or not
  QueueConst qc{};
  qc.ns_log2 = (rng() & 1) ? 3 : 0;
  qc.interleave = qc.ns_log2 == 3;
  qc.tiles_x = 1 + static_cast<int>(rng() % 40); qc.tiles_y = 1 + static_cast<int>(rng() % 40); qc.nframes = 1;   // (unused by pixel tickets)
  qc.ds = static_cast<int>(rng() % 7); qc.tpt = static_cast<int>(rng() % 3); qc.cap_log2 = 5; qc.ntiles = qc.tiles_x * qc.tiles_y;
  qc.order = nullptr; qc.deep_class = 0; qc.px = hdr;
  const int waves = 32 * (1 + static_cast<int>(rng() % 40)), ns = 1 << qc.ns_log2;
  const int static_first = static_cast<int>(rng() & 1);
  qc.home_waves = static_cast<unsigned>(waves >> qc.ns_log2);
  qc.q_static = static_first ? qc.home_waves : 0u;
  std::vector<unsigned> counter(static_cast<size_t>(ns), 0u);
  std::vector<unsigned char> cover(static_cast<size_t>(npix), 0);
  const int wpw = 4, grid = waves / wpw;
  struct Wave { unsigned state, rank; };
  std::vector<Wave> wv(static_cast<size_t>(waves));
  std::vector<int> alive(static_cast<size_t>(waves));
  for (int w = 0; w < waves; ++w) {
    const int b = w / wpw, wi = w % wpw;
    wv[static_cast<size_t>(w)].state = queue_state_init(b & (ns - 1), static_first != 0);
    wv[static_cast<size_t>(w)].rank = static_cast<unsigned>(wi) * static_cast<unsigned>(grid >> qc.ns_log2) + static_cast<unsigned>(b >> qc.ns_log2);
    alive[static_cast<size_t>(w)] = w;
  }
  unsigned long long taken = 0;
  while (!alive.empty()) {
    const size_t pick = rng() % alive.size();
    Wave &W = wv[static_cast<size_t>(alive[pick])];
    TicketSpan sp;
    if (!queue_draw(W.state, qc, W.rank, [&](int shard) { return counter[static_cast<size_t>(shard)]++; }, &sp)) {
      alive[pick] = alive.back();
      alive.pop_back();
      continue;
    }
    taken++;
    const int k = static_cast<int>(sp.cls);
    if (k < 0 || k >= kPxClasses || sp.q_end <= sp.q_next || sp.q_next < static_cast<unsigned>(pos[k]) || sp.q_end > static_cast<unsigned>(pos[k + 1]) ||
        sp.q_end - sp.q_next > (1u << px_log2(k)) || (sp.q_end - sp.q_next < (1u << px_log2(k)) && sp.q_end != static_cast<unsigned>(pos[k + 1]))) {
      std::printf("pixel ticket: bad span [%u, %u) of class %d (segment [%d, %d))\n", sp.q_next, sp.q_end, k, pos[k], pos[k + 1]);
      return 1;
    }
    for (unsigned i = sp.q_next; i < sp.q_end; ++i) {
      if (cover[i]) { std::printf("pixel ticket: list position %u handed out twice\n", i); return 1; }
      cover[i] = 1;
    }
  }
  for (int i = 0; i < npix; ++i)
    if (!cover[static_cast<size_t>(i)]) { std::printf("pixel ticket: list position %d never handed out\n", i); return 1; }
  if (taken != static_cast<unsigned long long>(hdr[8 + kPxClasses])) { std::printf("pixel tickets: %llu taken, header says %d\n", taken, hdr[8 + kPxClasses]); return 1; }
  if (verbose) std::printf("ok: pixel tickets, %d pixels, classes at %d %d %d %d, %d counter(s), %d waves, static %d: %llu tickets\n", npix, pos[1], pos[2], pos[3],
                           pos[4], ns, waves, static_first, taken);
  return 0;
}

int main(int argc, char **argv) {
  const int cases = argc > 1 ? std::atoi(argv[1]) : 2000;
  const unsigned seed = argc > 2 ? static_cast<unsigned>(std::atoi(argv[2])) : 1u;
  std::mt19937 rng(seed);
  // the production shapes first
  const Cfg fixed[] = {
      {125, 125, 0, 1, 2, 0, 3, 4096, 1, 0},  {125, 125, 3, 1, 2, 0, 3, 4096, 1, 0}, {125, 125, 0, 20, 2, 2, 3, 4096, 1, 0},
      {500, 500, 3, 1, 2, 2, 3, 4096, 1, 0},  {500, 63, 3, 1, 2, 2, 3, 2048, 1, 0},  {250, 250, 0, 1, 2, 2, 3, 4096, 0, 0},
      {1, 1, 0, 1, 2, 0, 3, 4096, 1, 0},      {3, 2, 3, 1, 2, 0, 3, 64, 1, 0},       {25, 25, 3, 1, 0, 0, 0, 1024, 1, 0},
      {125, 125, 3, 1, 2, 0, 3, 4096, 1, 1},  {500, 500, 3, 1, 2, 1, 3, 4096, 1, 1}, {3, 2, 3, 1, 2, 0, 3, 64, 0, 1},
  };
  for (const Cfg &c : fixed)
    if (run(c, rng, true)) return 1;
  for (int i = 0; i < cases; ++i) {
    Cfg c;
    c.tiles_x = 1 + static_cast<int>(rng() % 40);
    c.tiles_y = 1 + static_cast<int>(rng() % 40);
    c.ns_log2 = (rng() & 1) ? 3 : 0;
    c.nframes = (c.ns_log2 == 0 && (rng() % 3) == 0) ? 1 + static_cast<int>(rng() % 5) : 1;
    c.ds = static_cast<int>(rng() % 7);
    c.cap_log2 = static_cast<int>(rng() % 6);
    c.tpt = static_cast<int>(rng() % 5);
    c.deep_class = static_cast<int>(rng() % 5);
    c.waves = 32 * (1 + static_cast<int>(rng() % 40));   // workgroups of 4 waves, a multiple of 8 workgroups
    c.static_first = static_cast<int>(rng() & 1);
    c.interleave = (c.ns_log2 == 3 && (rng() & 1)) ? 1 : 0;
    if (run(c, rng, false)) {
      std::printf("FAILED case %d (seed %u)\n", i, seed);
      return 1;
    }
  }
  for (int i = 0; i < cases; ++i)
    if (run_px(rng, i < 3)) {
      std::printf("FAILED pixel-ticket case %d (seed %u)\n", i, seed);
      return 1;
    }
  std::printf("queue_check: %d random cases + %zu fixed ones passed\n", cases, sizeof(fixed) / sizeof(fixed[0]));
  return 0;
}
