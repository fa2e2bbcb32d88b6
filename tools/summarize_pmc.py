#!/usr/bin/env python3
"""Condense one rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES
SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE) into per-kernel fractions.   usage: summarize_pmc.py <pmc_dir> <out.json>
Units (MI355X_MICROARCH.md): SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over SIMDs (64 per v_mfma_f32_32x32x2_f32);
GRBM_GUI_ACTIVE sums the 8 XCDs; SQ_WAVE_CYCLES / SQ_WAIT_* count quad-cycles summed over waves."""
import collections, csv, glob, json, os, re, sys

d, out = sys.argv[1:3]
f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.Counter()
seen = set()
for r in csv.DictReader(open(f)):
    name = re.sub(r"<.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")).strip()
    acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (r.get("Dispatch_Id"), name)
    if key not in seen:
        seen.add(key)
        launches[name] += 1
SIMDS = 256 * 4
rep = {}
for name, c in sorted(acc.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0)):
    gui = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0                 # cycles of the dispatches, per XCD average
    if gui <= 0:
        continue
    wave = c.get("SQ_WAVE_CYCLES", 0.0)
    rep[name] = {
        "launches": launches[name],
        "mfma_busy_fraction": round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui * SIMDS), 4),
        "lds_bank_conflict_fraction_of_lds_cycles": round(c.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(c.get("SQ_LDS_IDX_ACTIVE", 0.0), 1.0), 4),
        "waves_per_simd_avg": round(wave * 4.0 / (gui * SIMDS), 2),
        "wave_time_issue_stalled": round(c.get("SQ_WAIT_INST_ANY", 0.0) / max(wave, 1.0), 3),
        "wave_time_parked": round(c.get("SQ_WAIT_ANY", 0.0) / max(wave, 1.0), 3),
        "share_of_gpu_cycles": 0.0,
    }
tot = sum(acc[n].get("GRBM_GUI_ACTIVE", 0.0) for n in rep)
for n in rep:
    rep[n]["share_of_gpu_cycles"] = round(acc[n].get("GRBM_GUI_ACTIVE", 0.0) / tot, 4)
top = dict(list(rep.items())[:12])
json.dump({"source": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES "
                     "SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE over `bench.py --eager --steps 2 --warmup 1` (all dispatches of "
                     "each kernel summed); see tools/summarize_pmc.py for the units", "kernels": top}, open(out, "w"), indent=1)
for n, v in top.items():
    print(f"{n[:44]:44s} launches {v['launches']:5d} mfma_busy {v['mfma_busy_fraction']:.4f} waves/simd {v['waves_per_simd_avg']:.2f} "
          f"issue-stalled {v['wave_time_issue_stalled']:.3f} parked {v['wave_time_parked']:.3f} share {v['share_of_gpu_cycles']:.3f}")
