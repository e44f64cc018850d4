"""One line per stage group from the counters.json files tools/stage_counters.sh leaves under gpurun_out/ctr_*."""
import json, sys, glob, os
for path in sorted(glob.glob(os.path.join(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out", "ctr_*", "counters.json"))):
    r = json.load(open(path))
    d = sum(r["duration_ms"]) / len(r["duration_ms"])
    B = float(r.get("batch", 4096))
    cyc = r["GRBM_GUI_ACTIVE"]  # busy clocks of one launch, summed over the 8 XCDs (the GHz column is 8 x the clock)
    print("%-22s %7.2f ms  %.2f GHz | per stream: VALU %6.0fk SALU %5.0fk LDS %5.0fk VMEM rd %5.1fk wr %5.1fk | wave time: issuing %2.0f%% waitcnt/barrier %2.0f%% issue-stall %2.0f%% | VALU pipe %2.0f%% | LDS conflicts %2.0f%%" % (
        os.path.basename(os.path.dirname(path)), d, cyc / (d * 1e6), r["SQ_INSTS_VALU"] / B / 1e3, r["SQ_INSTS_SALU"] / B / 1e3, r["SQ_INSTS_LDS"] / B / 1e3,
        r["SQ_INSTS_VMEM_RD"] / B / 1e3, r["SQ_INSTS_VMEM_WR"] / B / 1e3,
        100 * r["SQ_ACTIVE_INST_ANY"] / r["SQ_WAVE_CYCLES"], 100 * r["SQ_WAIT_ANY"] / r["SQ_WAVE_CYCLES"], 100 * r["SQ_WAIT_INST_ANY"] / r["SQ_WAVE_CYCLES"],
        100 * r["SQ_INSTS_VALU"] * 2 * 8 / (1024 * cyc)  # two cycles per wave64 instruction (tools/micro/valu_rate.hip), 100 * r["SQ_LDS_BANK_CONFLICT"] / max(1, r["SQ_LDS_IDX_ACTIVE"])))
