#!/usr/bin/env python3
"""Vector-issue utilisation of trace_kernel from rocprofv3 PMC passes.

rocprofv3's VALUBusy = SQ_ACTIVE_INST_VALU * 4 / (SIMDs * kernel cycles) is NOT a fraction on gfx950: the counter advances
by one "quad-cycle" per VALU instruction whatever the instruction costs, so loops that do nothing but issue VALU
instructions read 1.53 (v_fma_f32: 2.6 cycles per wave-instruction) or 0.91 (v_alignbit_b32: 4.4 cycles) --
tools/microbench/valu_issue_calib.hip, measured in the same session with the same counters.  What is a fraction is

    util = sum over instruction classes (instructions x issue cycles of the class) / (SIMDs x kernel cycles)

with the dynamic class counts from SQ_INSTS_VALU_* (tools/valu_mix.sh) and the two issue costs from the calibration loops:
  fast class (c_fast):  ADD_F32 + MUL_F32 + FMA_F32
  slow class (c_slow):  the sign capture (v_alignbit_b32, 8 per MFMA), INT64 (v_mad_u64_u32), CVT, MFMA issue slots
  2 x c_slow:           TRANS_F32 (v_sqrt / v_rcp / v_rsq: 8.2 cycles measured in round 1), the few F64 operations
  unknown:              INT32 and the unclassified rest (moves, bit operations, compares, selects): between c_fast and c_slow
which brackets the utilisation between `util_low` (unknown = fast) and `util_high` (unknown = slow).

usage: valu_issue.py <dir with calib/, bench/ (tools/valu_issue.sh) and mix1/, mix2/ (tools/valu_mix.sh)> <out.json>"""
import glob
import json
import os
import sqlite3
import sys

SIMDS, XCDS = 1024, 8


def counters(d):
    out = {}
    for db in glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True):
        c = sqlite3.connect(db)
        cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
        name_col = "kernel_name" if "kernel_name" in cols else "name"
        for name, counter, avg in c.execute("select %s, counter_name, avg(value) from counters_collection group by 1, 2" % name_col):
            out.setdefault(name, {})[counter] = avg
    return out


def valu_busy(v):
    return v["SQ_ACTIVE_INST_VALU"] * 4.0 / (SIMDS * v["GRBM_GUI_ACTIVE"] / XCDS)


def main():
    d, out_path = sys.argv[1], sys.argv[2]
    cal = counters(os.path.join(d, "calib"))
    calib = {k.split("(")[0]: valu_busy(v) for k, v in cal.items() if k.startswith("calib_")}
    c_fast, c_slow = 4.0 / calib["calib_fma"], 4.0 / calib["calib_alignbit"]
    v = {}
    for sub in ("bench", "mix1", "mix2", "mix3"):
        for k, vals in counters(os.path.join(d, sub)).items():
            if "trace_kernel" in k:
                v.update(vals)
                name = k
    g = lambda k: float(v.get(k, 0.0))
    simd_cycles = SIMDS * g("GRBM_GUI_ACTIVE") / XCDS
    fast = g("SQ_INSTS_VALU_ADD_F32") + g("SQ_INSTS_VALU_MUL_F32") + g("SQ_INSTS_VALU_FMA_F32")
    mfma = g("SQ_INSTS_VALU_MFMA_F16")
    alignbit = 8.0 * mfma                       # 32 v_alignbit_b32 per 4 MFMAs (64 rays x 32 spheres)
    slow = alignbit + g("SQ_INSTS_VALU_INT64") + g("SQ_INSTS_VALU_CVT") + mfma
    dbl = g("SQ_INSTS_VALU_ADD_F64") + g("SQ_INSTS_VALU_MUL_F64") + g("SQ_INSTS_VALU_FMA_F64")
    trans = g("SQ_INSTS_VALU_TRANS_F32")
    unknown = g("SQ_INSTS_VALU") - fast - slow - dbl - trans
    known_cycles = fast * c_fast + slow * c_slow + (trans + dbl) * 2.0 * c_slow
    lo = (known_cycles + unknown * c_fast) / simd_cycles
    hi = (known_cycles + unknown * c_slow) / simd_cycles
    res = {"kernel": name[:60],
           "valu_issue_util_low": round(lo, 4), "valu_issue_util_high": round(min(hi, 1.0), 4), "valu_issue_util_high_unclamped": round(hi, 4),
           "rocprof_VALUBusy": round(valu_busy(v), 4), "calibration_VALUBusy_of_pure_valu_loops": {k: round(x, 4) for k, x in calib.items()},
           "issue_cycles_per_wave_instruction": {"fast": round(c_fast, 3), "slow": round(c_slow, 3)},
           "lane_util": round(g("SQ_THREAD_CYCLES_VALU") / (64.0 * g("SQ_ACTIVE_INST_VALU")), 4),
           "mfma_busy": round(g("SQ_VALU_MFMA_BUSY_CYCLES") / simd_cycles, 4), "mfma_coexec_share": round(g("SQ_VALU_MFMA_COEXEC_CYCLES") / max(g("SQ_VALU_MFMA_BUSY_CYCLES"), 1.0), 4),
           "lds_bank_conflict_share": round(g("SQ_LDS_BANK_CONFLICT") / max(g("SQ_LDS_IDX_ACTIVE"), 1.0), 4),
           "per_launch": {"simd_cycles": simd_cycles, "valu": g("SQ_INSTS_VALU"), "fast_f32": fast, "sign_capture_alignbit": alignbit, "mfma_f16": mfma, "int64": g("SQ_INSTS_VALU_INT64"),
                          "int32": g("SQ_INSTS_VALU_INT32"), "cvt": g("SQ_INSTS_VALU_CVT"), "trans_f32": trans, "f64": dbl, "unclassified_and_int32": unknown, "salu": g("SQ_INSTS_SALU"), "lds": g("SQ_INSTS_LDS")},
           "note": "see tools/valu_issue.py; counters collected by tools/valu_issue.sh + tools/valu_mix.sh on the bench workload"}
    with open(out_path, "w") as f:
        json.dump(res, f, indent=1)
        f.write("\n")
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
