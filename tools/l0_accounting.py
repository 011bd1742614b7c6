#!/usr/bin/env python
"""Cycle accounting of one kernel from separate rocprofv3 PMC passes (tools/gpu_session.sh stage `pmc40`, summarised
by tools/rocpd_pmc.py):   python tools/l0_accounting.py profiles/r05_pmc_l0_accounting.csv
Prints where a wave's life and a SIMD's time go.  SQ wave/instruction-cycle counters count in units of 4 clocks
(one issue slot); everything is reported as a ratio, so the unit drops out."""
import csv
import sys

vals = {}
for row in csv.reader(open(sys.argv[1])):
    if len(row) == 6 and row[0] != "kernel":
        vals[row[2]] = float(row[4])
v = vals.get
wc = v("SQ_WAVE_CYCLES")
simd = 4.0 * v("SQ_BUSY_CU_CYCLES") / 4.0          # SIMD issue slots: CU-busy clocks * 4 SIMDs / 4 clocks per slot
print("wave life (fractions of SQ_WAVE_CYCLES = %.3g wave issue slots, %d waves):" % (wc, v("SQ_WAVES")))
for name, key in (("executing an instruction", "SQ_ACTIVE_INST_ANY"), ("  VALU (MFMA issue included)", "SQ_ACTIVE_INST_VALU"),
                  ("  scalar", "SQ_ACTIVE_INST_SCA"), ("  LDS", "SQ_ACTIVE_INST_LDS"), ("  misc (barrier, ...)", "SQ_ACTIVE_INST_MISC"),
                  ("ready, waiting for its turn to issue", "SQ_WAIT_INST_ANY"), ("  of which for the LDS port", "SQ_WAIT_INST_LDS"),
                  ("blocked on a dependency (s_waitcnt, barrier, MFMA result)", "SQ_WAIT_ANY")):
    print("  %-62s %5.1f %%" % (name, 100 * v(key) / wc))
n_valu, n_mfma = v("SQ_INSTS_VALU"), v("SQ_INSTS_MFMA")
print("instructions per MFMA: VALU %.2f (exp %.2f, fma %.2f, cvt %.2f, mul+add %.2f, int %.2f), LDS %.2f, SALU %.2f"
      % ((n_valu - n_mfma) / n_mfma, v("SQ_INSTS_VALU_TRANS_F32") / n_mfma, v("SQ_INSTS_VALU_FMA_F32") / n_mfma,
         v("SQ_INSTS_VALU_CVT") / n_mfma, (v("SQ_INSTS_VALU_MUL_F32") + v("SQ_INSTS_VALU_ADD_F32")) / n_mfma,
         v("SQ_INSTS_VALU_INT32") / n_mfma, v("SQ_INSTS_LDS") / n_mfma, v("SQ_INSTS_SALU") / n_mfma))
print("issue slots per VALU instruction: %.2f (a 64-lane fp32 op takes 1; v_exp_f32 therefore ~%.1f)"
      % (v("SQ_ACTIVE_INST_VALU") / n_valu,
         (v("SQ_ACTIVE_INST_VALU") - (n_valu - v("SQ_INSTS_VALU_TRANS_F32"))) / v("SQ_INSTS_VALU_TRANS_F32")))
mfma_busy, coexec = v("SQ_VALU_MFMA_BUSY_CYCLES") / 4.0, v("SQ_VALU_MFMA_COEXEC_CYCLES") / 4.0
valu_busy = v("SQ_ACTIVE_INST_VALU") - n_mfma          # VALU pipe slots without the MFMA issue slots
print("SIMD time (fractions of %.3g SIMD issue slots = SQ_BUSY_CU_CYCLES x 4 SIMDs / 4):" % simd)
print("  matrix pipe busy                     %5.1f %%" % (100 * mfma_busy / simd))
print("  VALU pipe busy (MFMA issue excluded) %5.1f %%" % (100 * valu_busy / simd))
print("  both at once                         %5.1f %%" % (100 * coexec / simd))
print("  at least one of them                 %5.1f %%   -> neither: %.1f %%"
      % (100 * (mfma_busy + valu_busy - coexec) / simd, 100 * (1 - (mfma_busy + valu_busy - coexec) / simd)))
print("LDS (one per CU): address unit active %.1f %% of the CU-busy clocks, bank-conflict clocks %.1f %% of those"
      % (100 * v("SQ_LDS_IDX_ACTIVE") / v("SQ_BUSY_CU_CYCLES"), 100 * v("SQ_LDS_BANK_CONFLICT") / v("SQ_LDS_IDX_ACTIVE")))
slots = (n_valu - n_mfma + v("SQ_INSTS_VALU_TRANS_F32")) / n_mfma + 1.0
print("VALU issue port: %.2f slots per MFMA (VALU %.2f + a second slot per v_exp %.2f + the MFMA's own 1) against the 8 slots "
      "an MFMA keeps the matrix pipe busy -> the port, not the pipe, is the floor: matrix pipe <= %.0f %%; measured port "
      "occupancy %.0f %%" % (slots, (n_valu - n_mfma) / n_mfma, v("SQ_INSTS_VALU_TRANS_F32") / n_mfma, 100 * 8 / slots,
                             100 * (valu_busy + n_mfma) / simd))
