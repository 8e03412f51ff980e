#!/usr/bin/env python3
"""MFMA utilisation per gather-GEMM kernel from the text tools/pmc_kernels.py prints for a run with
--pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVES.
  python tools/pmc_mfma_table.py pmc_kernels.txt > profiles/<round>_pmc_mfma.txt"""
import re
import subprocess
import sys

ker, cur = {}, None
for line in open(sys.argv[1]):
    if line.startswith("_Z") or line.startswith("k_"):
        cur = line.strip()
        ker[cur] = {}
    elif cur and line.strip():
        m = re.match(r"\s+(\S+)\s+([\d.]+)\s+\((\d+) dispatches\)", line)
        if m:
            ker[cur][m.group(1)] = (float(m.group(2)), int(m.group(3)))
print("# MFMA utilisation of the gather-GEMM kernels: rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY")
print("# SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace -- python3 bench.py")
print("# --steps 3 --warmup 1 --no-cpu-baseline --streams 1 --pairs 8, summarised by tools/pmc_kernels.py and this script.")
print("# Counters are per shader engine (32 SIMDs) and launch, averaged over the launches of a kernel.")
print("# mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (32 x SQ_BUSY_CYCLES): share of the launch during which a SIMD's matrix")
print("# pipe runs (a v_mfma_f32_32x32x2_f32 holds it 64 cycles); wait = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES;")
print("# weight = SQ_BUSY_CYCLES x launches (share of the listed kernels' time)")
rows = []
for k, v in ker.items():
    if "SQ_BUSY_CYCLES" not in v or "SQ_VALU_MFMA_BUSY_CYCLES" not in v:
        continue
    name = subprocess.run(["c++filt", k.replace(".kd", "")], capture_output=True, text=True).stdout.strip()
    name = name.replace("void ", "").split("(")[0]
    n = v["SQ_BUSY_CYCLES"][1]
    rows.append((v["SQ_BUSY_CYCLES"][0] * n, name, n, v["SQ_VALU_MFMA_BUSY_CYCLES"][0] / (32 * v["SQ_BUSY_CYCLES"][0]),
                 v["SQ_INSTS_VALU"][0] / max(1.0, v["SQ_INSTS_MFMA"][0]), v["SQ_WAIT_INST_ANY"][0] / v["SQ_WAVE_CYCLES"][0]))
tot = sum(r[0] for r in rows)
print("%-64s %8s %7s %10s %10s %6s" % ("kernel", "samples", "weight", "mfma_busy", "valu/mfma", "wait"))
acc = gtot = 0.0
for w, name, n, busy, vm, wait in sorted(rows, reverse=True):
    if "gather_gemm" in name:        # (the summary line is over the gather-GEMM kernels only, whatever else was profiled)
        acc += busy * w
        gtot += w
    print("%-64s %8d %6.1f%% %9.1f%% %10.1f %5.0f%%" % (name[:64], n, 100 * w / tot, 100 * busy, vm, 100 * wait))
print("# all gather-GEMM kernels, time-weighted: mfma_busy %.1f %%" % (100 * acc / max(gtot, 1e-30)))
