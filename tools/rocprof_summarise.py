"""Summarise the rocprofv3 rocpd databases written by tools/rocprof_collect.sh into a small
text/JSON report (kernel durations from --kernel-trace, FETCH_SIZE / WRITE_SIZE from the two
--pmc passes).  FETCH_SIZE on gfx950 counts wide coalesced reads at half their size
(MI355X_MICROARCH.md §HBM): both the raw and the x2-corrected read bytes are reported.
usage: python tools/rocprof_summarise.py gpurun_out/prof_<tag> [--last N]
"""
import sys, os, sqlite3, json, glob, collections, argparse

def load(dbpath):
    db = sqlite3.connect(dbpath)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    def tab(key):
        return [t for t in tabs if key in t][0]
    kd, ks, pe = tab("rocpd_kernel_dispatch"), tab("rocpd_info_kernel_symbol"), tab("rocpd_pmc_event")
    names = {r[0]: r[1] for r in cur.execute("select id, kernel_name from %s" % ks)}
    disp = list(cur.execute("select id, kernel_id, start, end, event_id, grid_size_x, workgroup_size_x from %s order by start" % kd))
    pmc = collections.defaultdict(float)
    for ev, val in cur.execute("select event_id, value from %s" % pe):
        pmc[ev] += val
    return names, disp, pmc

ap = argparse.ArgumentParser()
ap.add_argument("dir")
ap.add_argument("--kernel", default="sf_frame_kernel")
ap.add_argument("--last", type=int, default=3, help="timed launches = the last N dispatches of the kernel")
a = ap.parse_args()
out = {"kernel": a.kernel}
names, disp, _ = load(glob.glob(os.path.join(a.dir, "trace", "*.db"))[0])
mine = [d for d in disp if a.kernel in names[d[1]]]
durs = [(d[3] - d[2]) * 1e-6 for d in mine]
out["dispatches_total"] = len(mine)
out["grid_size_x"], out["workgroup_size_x"] = mine[-1][5], mine[-1][6]
out["timed_dispatches"] = a.last
out["duration_ms_timed"] = durs[-a.last:]
out["duration_ms_avg_timed"] = sum(durs[-a.last:]) / a.last
for key, sub in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
    f = glob.glob(os.path.join(a.dir, sub, "*.db"))
    if not f:
        continue
    names, disp, pmc = load(f[0])
    mine = [d for d in disp if a.kernel in names[d[1]]]
    vals = [pmc[d[4]] for d in mine][-a.last:]
    out[key + "_KB_per_dispatch"] = vals
    out[key + "_KB_avg"] = sum(vals) / len(vals)
if "FETCH_SIZE_KB_avg" in out and "WRITE_SIZE_KB_avg" in out:
    rd_raw = out["FETCH_SIZE_KB_avg"] * 1024.0
    wr = out["WRITE_SIZE_KB_avg"] * 1024.0
    out["hbm_bytes_per_launch_raw"] = rd_raw + wr
    out["hbm_bytes_per_launch"] = 2.0 * rd_raw + wr  # gfx950 FETCH_SIZE correction (x2 on reads)
    t = out["duration_ms_avg_timed"] * 1e-3
    out["hbm_GBps_corrected"] = out["hbm_bytes_per_launch"] / t / 1e9
    out["hbm_GBps_raw"] = out["hbm_bytes_per_launch_raw"] / t / 1e9
# (round 6) vector-ALU issue: rocprofv3 --pmc with several counters stores one event per (dispatch, counter)
def load_named(dbpath):
    db = sqlite3.connect(dbpath)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    tab = lambda key: [t for t in tabs if key in t][0]
    kd, ks, pe, pi = tab("rocpd_kernel_dispatch"), tab("rocpd_info_kernel_symbol"), tab("rocpd_pmc_event"), tab("rocpd_info_pmc")
    names = {r[0]: r[1] for r in cur.execute("select id, kernel_name from %s" % ks)}
    cname = {r[0]: r[1] for r in cur.execute("select id, name from %s" % pi)}
    disp = list(cur.execute("select id, kernel_id, start, end, event_id from %s order by start" % kd))
    vals = collections.defaultdict(lambda: collections.defaultdict(float))
    for ev, pid, val in cur.execute("select event_id, pmc_id, value from %s" % pe):
        vals[ev][cname.get(pid, str(pid))] += val
    return names, disp, vals

sq = glob.glob(os.path.join(a.dir, "sq", "*.db"))
gr = glob.glob(os.path.join(a.dir, "grbm", "*.db"))
if sq and gr:
    c = collections.defaultdict(float)
    for f in (sq[0], gr[0]):
        names, disp, vals = load_named(f)
        mine = [d for d in disp if a.kernel in names[d[1]]][-a.last:]
        for d in mine:
            for k, v in vals[d[4]].items():
                c[k] += v / len(mine)
    out["sq_counters_per_launch"] = dict(c)
    XCDS, SIMDS = 8, 1024
    cyc = c.get("GRBM_GUI_ACTIVE", 0.0) / XCDS  # the counter is summed over the XCDs
    if cyc > 0:
        out["gpu_active_cycles_per_launch"] = cyc
        out["valu_busy"] = 2.0 * c.get("SQ_ACTIVE_INST_VALU", 0.0) / (cyc * SIMDS)  # two cycles per wave64 instruction on gfx950 (tools/micro/valu_rate.hip; it was 4 until round 6)
        out["valu_insts_per_launch"] = c.get("SQ_INSTS_VALU", 0.0)
        out["valu_quad_cycles_per_inst"] = c.get("SQ_ACTIVE_INST_VALU", 0.0) / max(c.get("SQ_INSTS_VALU", 0.0), 1.0)
print(json.dumps(out, indent=1))
