#!/usr/bin/env python3
"""k_sample's vector-ALU counters held against a kernel that saturates the vector ALU (scripts/valu_pass.sh).

The counters' units are not documented for gfx950 in the guides at hand, so nothing here trusts one: the yardstick is
scripts/ubench/philox_mul's compiler form at 8 waves a SIMD -- dependent Philox blocks, nothing but VALU work, every
lane active -- measured in the same pass.  For a kernel K:
  valu_busy      = (SQ_ACTIVE_INST_VALU / GRBM_GUI_ACTIVE)_K / (the same ratio)_saturated
  lanes_active   = (SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU)_K / (the same ratio)_saturated
SQ_ACTIVE_INST_VALU turns out to count about one per VALU instruction whatever the instruction, so a kernel's
valu_busy depends on how many of its instructions are the slow ones (v_mad_u64_u32): against the Philox yardstick (a
third of its instructions are such multiplies) it is an UPPER bound for a kernel with fewer of them, against the
same loop without multiplies (`valu_busy_vs_no_multiplies`) a LOWER bound.  `valu_busy_by_mix` prices k_sample's own
mix: 20 multiplies per Philox block (one block per wave and loop iteration; iterations = draws / 2 / active lanes)
at the cost the yardsticks give for a multiply and for any other instruction.
`k_sample_alone`: bench.py's 50 launches after the timed region (the library's default geometry, GPU to itself);
`k_sample_step`: the launches of the steps (--serial --sample-wg-per-cu 7, the in-step geometry)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def dispatches(dirname):
    per = defaultdict(dict)
    for f in glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                d = per[int(r["Dispatch_Id"])]
                d["name"] = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].strip()
                d["grid"] = int(r["Grid_Size"])
                d[r["Counter_Name"]] = float(r["Counter_Value"])
    return [per[k] for k in sorted(per)]


def mean(ds):
    keys = [k for k in ds[0] if k not in ("name", "grid")]
    return {k: sum(d[k] for d in ds) / len(ds) for k in keys} | {"launches": len(ds)}


def ratios(m):
    return m["SQ_ACTIVE_INST_VALU"] / m["GRBM_GUI_ACTIVE"], m["SQ_THREAD_CYCLES_VALU"] / m["SQ_ACTIVE_INST_VALU"]


def main(out):
    ub = dispatches(os.path.join(out, "valu_ubench"))
    # philox_mul launches form 0..3 for 1, 2, 4, 8 waves a SIMD, 5 times each: grid = 256 CUs x waves x 256 lanes
    sat = [d for d in ub if d["name"].startswith("k<0") and d["grid"] == 256 * 8 * 256]
    nomul = [d for d in ub if d["name"].startswith("k<3") and d["grid"] == 256 * 8 * 256]
    one = [d for d in ub if d["name"].startswith("k<0") and d["grid"] == 256 * 1 * 256]
    be = [d for d in dispatches(os.path.join(out, "valu_bench")) if d["name"].startswith("k_sample")]
    alone, step = be[-50:], be[:-55]
    res = {"saturated_philox_8_waves": mean(sat), "saturated_no_multiplies_8_waves": mean(nomul),
           "philox_1_wave": mean(one), "k_sample_alone": mean(alone), "k_sample_step": mean(step)}
    a0, l0 = ratios(res["saturated_philox_8_waves"])
    a1, _ = ratios(res["saturated_no_multiplies_8_waves"])
    # cycles of one SIMD per instruction, from the yardsticks (GRBM_GUI_ACTIVE adds the 8 XCDs up; 1024 SIMDs)
    simd_cycles = lambda m: m["GRBM_GUI_ACTIVE"] / 8 * 1024  # noqa: E731
    other = simd_cycles(res["saturated_no_multiplies_8_waves"]) / res["saturated_no_multiplies_8_waves"]["SQ_INSTS_VALU"]
    ph = res["saturated_philox_8_waves"]
    blocks = 256 * 8 * 4 * 2000                    # waves x ITER of philox_mul
    per_block = ph["SQ_INSTS_VALU"] / blocks
    mul = (simd_cycles(ph) / blocks - (per_block - 20) * other) / 20
    res["cycles_per_instruction"] = {"v_mad_u64_u32": mul, "other_valu": other, "valu_instructions_per_philox_block": per_block}
    draws = None
    try:
        line = json.loads(open(os.path.join(out, "valu_bench.json")).read().strip().splitlines()[-1])
        draws = line["draws_per_sample"] * line["config"]["samples_per_gpu"]
    except (OSError, ValueError, KeyError):
        pass
    for k in ("saturated_no_multiplies_8_waves", "philox_1_wave", "k_sample_alone", "k_sample_step"):
        a, lanes = ratios(res[k])
        res[k]["valu_busy"] = a / a0
        res[k]["valu_busy_vs_no_multiplies"] = a / a1
        res[k]["lanes_active"] = lanes / l0
        if k.startswith("k_sample") and draws:
            iters = draws / 2 / (64 * res[k]["lanes_active"])
            muls = 20 * iters
            res[k]["valu_busy_by_mix"] = (muls * mul + (res[k]["SQ_INSTS_VALU"] - muls) * other) / simd_cycles(res[k])
            res[k]["loop_iterations_per_launch"] = iters
            res[k]["valu_instructions_per_iteration"] = res[k]["SQ_INSTS_VALU"] / iters
            res[k]["salu_instructions_per_iteration"] = res[k]["SQ_INSTS_SALU"] / iters
    res["note"] = ("valu_busy and lanes_active are relative to scripts/ubench/philox_mul's compiler form at 8 waves a SIMD "
                   "(= 1.0), measured in the same rocprofv3 --pmc pass; see scripts/valu_summary.py")
    try:
        from montecarlo_amd.build import sources_digest
        res["csrc_sha256"] = sources_digest()
    except Exception as e:  # noqa: BLE001
        res["csrc_sha256"] = None
        res["digest_error"] = str(e)
    json.dump(res, sys.stdout, indent=1)


if __name__ == "__main__":
    main(sys.argv[1])
