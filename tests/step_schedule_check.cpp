// Host-side replay of the whole-step kernels' static tile schedule (zonos_amd/csrc/zn_step_sched.h), compiled with g++ by
// tests/test_step_schedule.py.  It walks a block exactly as a compute wave of a streaming workgroup does (prefetch phase, then the slots
// in order, with the requests each slot raises at once or holds until its op's publish) and checks the data flow the kernel relies on:
//   * every slot reads the tile it is supposed to read: from an LDS park slot that holds it, or from a register buffer whose LAST request
//     was that tile and has not been overwritten since;
//   * no request overwrites a register buffer whose tile has not been consumed;
//   * every REG load is requested exactly once, every tile is consumed exactly once (op 0's tiles twice: op 1 re-reads them);
//   * requests for another op's tiles are held until that op's publish when the deferral mask says so, except the first EARLY of them.
// Exit status 0 = every instantiation passed; a failed check prints what and where.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "zn_step_sched.h"

static int fails = 0;
#define CHECK(cond, ...) do { if (!(cond)) { ++fails; std::printf("FAIL %s: ", name); std::printf(__VA_ARGS__); std::printf("\n"); } } while (0)

template <int T_OUT, int T_FC1, int T_FC2, int T_IN, int NB, int P, int NH, int MASK, int EARLY>
static void replay(const char* name) {
  using SC = StepSched<T_OUT, T_FC1, T_FC2, T_IN, NB, P, NH, MASK, EARLY>;
  constexpr int NS = SC::NS, NL = SC::NL, NREG = SC::NREG;
  std::vector<int> buf(NB, -1);              // load held (or in flight) in each register buffer, -1 = free
  std::vector<int> park(P > 0 ? P : 1, -1);  // load held in each LDS park slot
  std::vector<int> requested(NL, 0), consumed(NL, 0);
  std::vector<int> transit(NB, -1);          // prefetch phase: parked tile on its way through a register buffer
  auto request_reg = [&](int k, const char* when, int at) {
    CHECK(k >= 0 && k < NREG, "request of REG load %d out of range (%s %d)", k, when, at);
    const int l = SC::nth_reg(k), b = (k + P) % NB;
    CHECK(l >= 0 && SC::src_of(l) == 0, "nth_reg(%d) = %d is not a REG load", k, l);
    CHECK(buf[b] == -1, "REG load %d (tile %d) requested into buffer %d which still holds tile %d (%s %d)", k, l, b, buf[b], when, at);
    CHECK(transit[b] == -1, "REG load %d (tile %d) requested into buffer %d while parked tile %d is still in transit there (%s %d)", k, l, b, transit[b], when, at);
    buf[b] = l;
    ++requested[l];
  };
  // ---- prefetch phase: items 0 .. P-1 transit through the buffers into the park slots, then the first NB REG loads stay in the buffers
  for (int i = 0; i < NB; ++i) {
    if (i < P) { transit[i % NB] = i; ++requested[i]; }
    else request_reg(i - P, "prefetch start", i);
  }
  for (int l = 0; l < P; ++l) {
    CHECK(transit[l % NB] == l, "transit buffer %d holds %d, expected parked tile %d", l % NB, transit[l % NB], l);
    park[l] = l;
    transit[l % NB] = -1;
    const int i = l + NB;
    if (i < P) { CHECK(buf[i % NB] == -1, "transit into a busy buffer"); transit[i % NB] = i; ++requested[i]; }
    else if (i - P < NREG) request_reg(i - P, "prefetch", l);
  }
  for (int b = 0; b < NB; ++b) CHECK(transit[b] == -1, "a parked tile is still in transit at the end of the prefetch phase");
  // ---- the slots
  int early_seen[5] = {0, 0, 0, 0, 0};
  for (int s = 0; s < NS; ++s) {
    const int op = SC::op_of(s);
    if (s == SC::first_of(op) && op > 0) {
      // the op before has been published: its held requests go out (the kernel walks the previous op's slots in order)
      for (int q = SC::first_of(op - 1); q < SC::first_of(op); ++q)
        if (SC::raise_late(q)) request_reg(SC::raised_by(q), "publish of op", op - 1);
      // helper tiles are dropped into the slots op 1 has left (after A(1), i.e. before op 2's first slot)
      if (op == 2) for (int h = 0; h < NH; ++h) { park[h] = SC::L_F2 - NH + h; ++requested[SC::L_F2 - NH + h]; }
    }
    const int l = SC::load_of_slot(s);
    CHECK(l >= 0 && l < NL, "slot %d maps to load %d", s, l);
    CHECK(SC::slot_of_load(l) == (op == 1 ? s - T_OUT : s), "slot_of_load(load_of_slot(%d)) = %d", s, SC::slot_of_load(l));
    if (SC::src_of(l) != 0) {
      const int sl = SC::slot_of(l);
      CHECK(sl >= 0 && sl < P && park[sl] == l, "slot %d (op %d) wants tile %d from park slot %d, which holds %d", s, op, l, sl, sl >= 0 && sl < P ? park[sl] : -2);
    } else {
      const int k = SC::regk(l), b = (k + P) % NB;
      CHECK(SC::nth_reg(k) == l, "regk / nth_reg disagree at load %d", l);
      CHECK(buf[b] == l, "slot %d (op %d) wants tile %d from register buffer %d, which holds %d", s, op, l, b, buf[b]);
      buf[b] = -1;                                            // consumed: the buffer is free for the request this slot raises
    }
    ++consumed[l];
    const int k = SC::raised_by(s);
    if (k >= 0) {
      const bool cross = SC::target_op(k) != op;
      CHECK(SC::raise_now(s) != SC::raise_late(s), "slot %d raises request %d both now and late (or never)", s, k);
      if (SC::raise_now(s)) {
        if (cross && ((MASK >> op) & 1)) { ++early_seen[op]; CHECK(early_seen[op] <= EARLY, "more than EARLY = %d early requests in op %d", EARLY, op); }
        request_reg(k, "slot", s);
      }
    } else CHECK(!SC::raise_now(s) && !SC::raise_late(s), "slot %d raises nothing but is flagged", s);
  }
  for (int l = 0; l < NL; ++l) {
    CHECK(requested[l] == 1, "tile %d requested %d times", l, requested[l]);
    CHECK(consumed[l] == (l < T_OUT ? 2 : 1), "tile %d consumed %d times", l, consumed[l]);
  }
  for (int b = 0; b < NB; ++b) CHECK(buf[b] == -1, "register buffer %d still holds tile %d at the end of the block", b, buf[b]);
  std::printf("%s: %d slots, %d tiles (%d parked, %d helper, %d through %d register buffers)%s\n", name, NS, NL, P, NH, NREG, NB, fails ? "  <-- FAILED" : "");
}

int main() {
  replay<2, 10, 5, 6, 3, 4, 0, 0xF, 0>("step_kernel<4,2,10,5,6,6> (shipped: 1 - 6 key blocks)");
  replay<2, 11, 6, 7, 3, 4, 0, 0xF, 0>("step_kernel<4,2,11,6,7,8> (shipped: 7 - 8 key blocks)");
  replay<2, 13, 7, 8, 3, 4, 0, 0xF, 0>("step_kernel<4,2,13,7,8,12> (shipped: 9 - 12 key blocks)");
  replay<2, 10, 5, 6, 3, 4, 2, 0xF, 0>("helper waves (ZN_SK_HELP 2)");
  replay<2, 10, 5, 6, 4, 4, 0, 0xF, 0>("4 register buffers");
  replay<2, 10, 5, 6, 3, 3, 0, 0xF, 0>("3 parked tiles");
  replay<2, 10, 5, 6, 3, 2, 0, 0xF, 0>("2 parked tiles");
  replay<2, 10, 5, 6, 3, 4, 0, 0xF, 1>("1 early request per op");
  replay<2, 10, 5, 6, 3, 4, 0, 0xF, 2>("2 early requests per op");
  replay<2, 10, 5, 6, 3, 4, 0, 0xB, 0>("deferral off for fc1");
  replay<2, 10, 5, 6, 3, 4, 0, 0x0, 0>("no deferral");
  replay<2, 10, 5, 6, 2, 4, 0, 0xF, 0>("2 register buffers");
  replay<0, 13, 7, 8, 3, 4, 0, 0xF, 0>("no out_proj slots (round 3's three-role experiment, bulk role; retired)");
  return fails ? 1 : 0;
}
