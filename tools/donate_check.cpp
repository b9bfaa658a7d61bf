// donate_check.cpp -- the ray-donation protocol of the pooled kernel's DONATE instantiation, played on the CPU.
//
// render_kernels.hip (pooled_kernel, TAIL == 2): a wave that has left the pooled loop decrements the workgroup's count of
// active waves, sets its bit in the workgroup's word of waiting waves and sleeps, looking at the count and then at its own inbox
// flag whenever it wakes; a wave in the loop that cannot refill claims waiting waves (fetch-and on the word) and gives each one
// ray: the ray's dwords into the receiver's inbox, then the receiver's flag.  The receiver takes the ray, clears the flag, walks
// the chain, and offers itself again; it ends when the count is zero and its inbox is empty.
//
// This is a MODEL of that protocol -- every LDS operation of the kernel's code is one atomic step here, W emulated waves are
// stepped in a random interleaving -- not the kernel's code itself (the kernel has it inline).  Checked over many random
// workgroups:
//   * every ray is finished exactly once, by its own wave or by a receiver,
//   * an inbox is never written while its flag is set, nor read half-written, and no ray is given to a wave that has ended,
//   * every wave ends (no wave waits for ever), within a step bound.
// Host-only; part of the CPU test suite (tests/test_host_logic.py).
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

namespace {

enum Pc {
  LOOP,          // in the pooled loop: between operations
  DONOR_CLAIM,   // fetch-and on the word for the chosen wave
  DONOR_WRITE0, DONOR_WRITE1, DONOR_WRITE2,   // the ray's three 16-byte quarters into the inbox
  DONOR_FLAG,    // the receiver's flag
  LEAVE,         // count of active waves -= 1
  OFFER,         // word |= bit
  READ_COUNT, READ_FLAG, DECIDE,
  TAKE0, TAKE1, TAKE2,   // the three quarters out of the inbox
  CLEAR_FLAG,
  SOLO,          // walking the chain
  SLEEP,
  ENDED
};

struct Wave {
  Pc pc = LOOP;
  std::vector<int> rays;     // ray ids this wave holds in the pooled loop
  int can_refill = 0;        // operations left before the wave cannot refill any more
  unsigned idle_seen = 0;    // donor: the word as last seen
  int target = -1, giving = -1;
  unsigned count_seen = 0, flag_seen = 0;
  bool offered = false;
  int taken[3] = {-1, -1, -1};
  int solo_left = 0, solo_ray = -1;
};

struct Group {
  unsigned word = 0, count = 0;
  std::vector<unsigned> flag;
  std::vector<int> inbox;    // 3 entries per wave: the ray id, written quarter by quarter
};

// mutate (the checker's own test: each of these must be caught): 1 = the receiver looks at its flag BEFORE the count,
// 2 = the donor raises the flag before it has written the ray
int run(int W, int max_rays, int donate_max, std::mt19937 &rng, bool verbose, int mutate) {
  Group g;
  g.count = static_cast<unsigned>(W);
  g.flag.assign(static_cast<size_t>(W), 0u);
  g.inbox.assign(static_cast<size_t>(3 * W), -1);
  std::vector<Wave> wv(static_cast<size_t>(W));
  int nrays = 0;
  for (auto &w : wv) {
    const int n = static_cast<int>(rng() % static_cast<unsigned>(max_rays + 1));
    for (int k = 0; k < n; ++k) w.rays.push_back(nrays++);
    w.can_refill = static_cast<int>(rng() % 6u);
  }
  std::vector<int> finished(static_cast<size_t>(nrays), 0);
  auto fail = [&](const char *what, int wave) {
    std::printf("donate_check: %s (wave %d of %d)\n", what, wave, W);
    return 1;
  };
  const long bound = 200000L + 4000L * nrays;
  long steps = 0;
  int alive = W;
  while (alive > 0) {
    if (++steps > bound) return fail("step bound exceeded: some wave waits for ever", -1);
    int i = static_cast<int>(rng() % static_cast<unsigned>(W));
    while (wv[static_cast<size_t>(i)].pc == ENDED) i = (i + 1) % W;
    Wave &w = wv[static_cast<size_t>(i)];
    switch (w.pc) {
      case LOOP: {
        // one operation of the pooled loop: some rays end; a wave that cannot refill and stands at a bounce boundary donates
        if (!w.rays.empty() && rng() % 3u == 0u) {
          finished[static_cast<size_t>(w.rays.back())]++;
          w.rays.pop_back();
        }
        if (w.can_refill > 0) {
          w.can_refill--;
          break;
        }
        if (w.rays.empty()) {
          w.pc = LEAVE;
          break;
        }
        if (static_cast<int>(w.rays.size()) <= donate_max && rng() % 2u == 0u) {   // (a SHADE with both lists empty)
          w.idle_seen = g.word;                                                     // the look at the word
          if (w.idle_seen != 0u) w.pc = DONOR_CLAIM;
        }
        break;
      }
      case DONOR_CLAIM: {
        const int t = __builtin_ctz(w.idle_seen);
        const unsigned old = g.word;
        g.word &= ~(1u << t);
        w.idle_seen = old & ~(1u << t);
        if ((old >> t) & 1u) {
          w.target = t;
          w.giving = w.rays.back();
          w.rays.pop_back();
          w.pc = DONOR_WRITE0;
        } else {
          w.pc = (w.idle_seen != 0u && !w.rays.empty()) ? DONOR_CLAIM : LOOP;
        }
        break;
      }
      case DONOR_WRITE0: case DONOR_WRITE1: case DONOR_WRITE2: {
        const int q = w.pc - DONOR_WRITE0;
        if (wv[static_cast<size_t>(w.target)].pc == ENDED) return fail("a ray was given to a wave that has ended", i);
        if (g.flag[static_cast<size_t>(w.target)] != 0u) return fail("an inbox was written while its flag was set", i);
        if (mutate == 2 && q == 0) g.flag[static_cast<size_t>(w.target)] = 1u;
        g.inbox[static_cast<size_t>(3 * w.target + q)] = w.giving;
        w.pc = static_cast<Pc>(w.pc + 1);
        break;
      }
      case DONOR_FLAG:
        g.flag[static_cast<size_t>(w.target)] = 1u;
        w.pc = (w.idle_seen != 0u && !w.rays.empty()) ? DONOR_CLAIM : LOOP;
        break;
      case LEAVE:
        g.count -= 1u;
        w.offered = false;
        w.pc = OFFER;
        break;
      case OFFER:
        if (!w.offered) {
          if (g.flag[static_cast<size_t>(i)] != 0u) return fail("a wave offered itself with a full inbox", i);
          g.word |= 1u << i;
          w.offered = true;
        }
        w.flag_seen = 0u;
        w.pc = mutate == 1 ? READ_FLAG : READ_COUNT;
        break;
      case READ_COUNT:
        w.count_seen = g.count;
        w.pc = mutate == 1 ? DECIDE : READ_FLAG;
        break;
      case READ_FLAG:
        w.flag_seen = g.flag[static_cast<size_t>(i)];
        w.pc = mutate == 1 ? READ_COUNT : DECIDE;
        break;
      case DECIDE:
        if (w.flag_seen != 0u) w.pc = TAKE0;
        else if (w.count_seen == 0u) {
          w.pc = ENDED;
          alive--;
        } else w.pc = SLEEP;
        break;
      case TAKE0: case TAKE1: case TAKE2:
        w.taken[w.pc - TAKE0] = g.inbox[static_cast<size_t>(3 * i + (w.pc - TAKE0))];
        w.pc = static_cast<Pc>(w.pc + 1);
        break;
      case CLEAR_FLAG:
        if (g.flag[static_cast<size_t>(i)] == 0u) return fail("a receiver took a ray from an inbox whose flag was down", i);
        if (w.taken[0] < 0 || w.taken[0] != w.taken[1] || w.taken[1] != w.taken[2]) return fail("an inbox was read half-written", i);
        g.flag[static_cast<size_t>(i)] = 0u;
        w.offered = false;
        w.solo_ray = w.taken[0];
        w.solo_left = static_cast<int>(rng() % 8u);
        w.pc = SOLO;
        break;
      case SOLO:
        if (w.solo_left-- <= 0) {
          finished[static_cast<size_t>(w.solo_ray)]++;
          w.pc = OFFER;
        }
        break;
      case SLEEP:
        w.pc = OFFER;
        break;
      case ENDED:
        break;
    }
  }
  for (int r = 0; r < nrays; ++r)
    if (finished[static_cast<size_t>(r)] != 1) {
      std::printf("donate_check: ray %d finished %d times (W %d)\n", r, finished[static_cast<size_t>(r)], W);
      return 1;
    }
  for (int k = 0; k < W; ++k)
    if (g.flag[static_cast<size_t>(k)] != 0u) return fail("a wave ended with a full inbox", k);
  if (verbose) std::printf("  W %d rays %d: %ld steps\n", W, nrays, steps);
  return 0;
}

}  // namespace

int main(int argc, char **argv) {
  const int trials = argc > 1 ? std::atoi(argv[1]) : 20000;
  const unsigned seed = argc > 2 ? static_cast<unsigned>(std::atoi(argv[2])) : 1u;
  const int mutate = argc > 3 ? std::atoi(argv[3]) : 0;
  std::mt19937 rng(seed);
  for (int t = 0; t < trials; ++t) {
    const int W = 1 + static_cast<int>(rng() % 16u);
    const int max_rays = (rng() % 4u == 0u) ? 64 : static_cast<int>(rng() % 6u);
    const int donate_max = (rng() % 3u == 0u) ? 64 : 1 + static_cast<int>(rng() % 8u);
    if (run(W, max_rays, donate_max, rng, false, mutate)) {
      std::printf("donate_check: FAILED in trial %d (seed %u)\n", t, seed);
      return 1;
    }
  }
  std::printf("donate_check: %d random workgroups, protocol holds\n", trials);
  return 0;
}
