// Plain C++ (no HIP): shared by the kernel (zn_step_kernel.h) and by the host-side schedule check
// (tests/test_step_schedule.py compiles tests/step_schedule_check.cpp with g++ and replays the schedule on the CPU).
#pragma once

// The static tile schedule of a streaming workgroup's compute wave, as compile-time arithmetic.  Slots s = 0 .. NS-1 in processing
// order (op 0's T_OUT tiles, op 1 re-reading them, fc1, fc2, in_proj / heads); "loads" l = 0 .. NL-1 = the distinct tiles; each load
// comes from PARK (streamed through the register buffers into LDS slots 0 .. P-1 while the attention runs), from HELP (a helper
// wave's registers -> LDS slots 0 .. NH-1 after op 1: the last NH fc1 tiles) or from REG (the NB rotating register buffers).
template <int T_OUT, int T_FC1, int T_FC2, int T_IN, int NB, int P, int NH, int MASK, int EARLY>
struct StepSched {
  static constexpr int S1 = T_OUT, S2 = 2 * T_OUT, S3 = S2 + T_FC1, S4 = S3 + T_FC2, NS = S4 + T_IN;
  static constexpr int NL = NS - T_OUT, L_F2 = T_OUT + T_FC1, NREG = NL - P - NH;
  static constexpr int op_of(int s) { return s < S1 ? 0 : s < S2 ? 1 : s < S3 ? 2 : s < S4 ? 3 : 4; }
  static constexpr int first_of(int op) { return op == 0 ? 0 : op == 1 ? S1 : op == 2 ? S2 : op == 3 ? S3 : S4; }
  static constexpr int slot_of_load(int l) { return l < S1 ? l : l + T_OUT; }
  static constexpr int load_of_slot(int s) { return s < S1 ? s : s - T_OUT; }
  static constexpr int src_of(int l) { return l < P ? 1 : (l >= L_F2 - NH && l < L_F2) ? 2 : 0; }                 // 0 REG, 1 PARK, 2 HELP
  static constexpr int slot_of(int l) { return l < P ? l : l - (L_F2 - NH); }                                     // LDS slot of a PARK / HELP load
  static constexpr int regk(int l) { int k = 0; for (int i = 0; i < l; ++i) k += src_of(i) == 0; return k; }      // REG loads before l
  static constexpr int nth_reg(int k) { int n = 0; for (int i = 0; i < NL; ++i) { if (src_of(i) == 0) { if (n == k) return i; ++n; } } return -1; }
  // REG request (its number) raised by the last use of slot s's register buffer, -1: none
  static constexpr int raised_by(int s) {
    const int l = load_of_slot(s);
    if (op_of(s) == 1 || src_of(l) != 0) return -1;         // op 1 re-reads op 0's (parked) tiles
    const int k = regk(l);
    return k + NB < NREG ? k + NB : -1;
  }
  static constexpr int target_op(int k) { return op_of(slot_of_load(nth_reg(k))); }
  // requests for ANOTHER op's tiles raised by the slots of s's op before s
  static constexpr int cross_idx(int s) {
    const int op = op_of(s);
    int n = 0;
    for (int q = first_of(op); q < s; ++q) { const int k = raised_by(q); if (k >= 0 && target_op(k) != op) ++n; }
    return n;
  }
  // does slot s raise its request at once (true) or hold it until its op's results are published?
  static constexpr bool raise_now(int s) {
    const int k = raised_by(s), op = op_of(s);
    return k >= 0 && (((MASK >> op) & 1) == 0 || target_op(k) == op || cross_idx(s) < EARLY);
  }
  static constexpr bool raise_late(int s) { return raised_by(s) >= 0 && !raise_now(s); }
};

