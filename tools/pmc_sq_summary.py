#!/usr/bin/env python3
"""Per-kernel matrix-core / vector-ALU / LDS utilisation from rocprofv3 --pmc passes (counter_collection CSVs).
usage: python tools/pmc_sq_summary.py dir [dir ...]      (one directory per pass; means per dispatch are merged by kernel)

Units (gfx950): GRBM_GUI_ACTIVE is summed over the 8 XCDs (kernel cycles = /8); SQ_VALU_MFMA_BUSY_CYCLES counts cycles
summed over the SIMDs (32 per v_mfma_f32_32x32x16_bf16); SQ_ACTIVE_INST_*, SQ_WAVE_CYCLES, SQ_WAIT_* count quad-cycles
summed over the waves; SQ_INSTS_VALU_MFMA_MOPS_BF16 counts 512 FLOP units.  Derived columns:
  mfma_util  = SQ_VALU_MFMA_BUSY_CYCLES / (kernel cycles x 1024 SIMDs)
  valu_busy  = 4 x SQ_ACTIVE_INST_VALU / (kernel cycles x 1024)      (vector-ALU issue incl. MFMA issue slots)
  coexec     = SQ_VALU_MFMA_COEXEC_CYCLES / SQ_VALU_MFMA_BUSY_CYCLES (share of matrix time with vector work beside it)
  lds_busy   = SQ_LDS_IDX_ACTIVE / (kernel cycles x 256 CUs)
  wait/wave  = SQ_WAIT_ANY / SQ_WAVE_CYCLES, issue-stall/wave = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES
  valu/mfma  = (SQ_INSTS_VALU - SQ_INSTS_MFMA) / SQ_INSTS_MFMA       (vector instructions per matrix instruction)
"""
import collections
import csv
import glob
import re
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("scm::", "")
            agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
rows = []
for name, ctrs in agg.items():
    if not any(k in name for k in ("gemm", "wgrad", "favor", "ln_", "ffn", "head", "embed")):
        continue
    m = {c: sum(v) / len(v) for c, v in ctrs.items()}
    n = len(next(iter(ctrs.values())))
    cyc = m.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    rows.append((cyc * n, name, n, m, cyc))
rows.sort(reverse=True)
g = lambda m, k: m.get(k, float("nan"))
print(f"{'kernel':44s} {'n':>5s} {'kcycles':>8s} {'mfma_util':>9s} {'valu_busy':>9s} {'coexec':>7s} {'lds_busy':>8s} "
      f"{'wait/wave':>9s} {'stall/wave':>10s} {'valu/mfma':>9s} {'trans/mfma':>10s} {'bank_confl/lds':>14s} {'MFMA TFLOP':>10s}")
for tot, name, n, m, cyc in rows[:16]:
    simd_cyc = cyc * 1024 if cyc else float("nan")
    mf = g(m, "SQ_INSTS_MFMA")
    print(f"{name[:44]:44s} {n:5d} {cyc / 1e3:8.1f} {g(m, 'SQ_VALU_MFMA_BUSY_CYCLES') / simd_cyc:9.3f} "
          f"{4 * g(m, 'SQ_ACTIVE_INST_VALU') / simd_cyc:9.3f} "
          f"{g(m, 'SQ_VALU_MFMA_COEXEC_CYCLES') / max(g(m, 'SQ_VALU_MFMA_BUSY_CYCLES'), 1):7.3f} "
          f"{g(m, 'SQ_LDS_IDX_ACTIVE') / (cyc * 256 if cyc else float('nan')):8.3f} "
          f"{g(m, 'SQ_WAIT_ANY') / max(g(m, 'SQ_WAVE_CYCLES'), 1):9.3f} "
          f"{g(m, 'SQ_WAIT_INST_ANY') / max(g(m, 'SQ_WAVE_CYCLES'), 1):10.3f} "
          f"{(g(m, 'SQ_INSTS_VALU') - mf) / max(mf, 1):9.2f} {g(m, 'SQ_INSTS_VALU_TRANS_F32') / max(mf, 1):10.2f} "
          f"{g(m, 'SQ_LDS_BANK_CONFLICT') / max(g(m, 'SQ_LDS_IDX_ACTIVE'), 1):14.3f} "
          f"{g(m, 'SQ_INSTS_VALU_MFMA_MOPS_BF16') * 512 / 1e12:10.4f}")
print()
print("raw means per dispatch:")
for tot, name, n, m, cyc in rows[:16]:
    print(name[:70])
    print("   " + "  ".join(f"{c}={v:.4g}" for c, v in sorted(m.items())))
