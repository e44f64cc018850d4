#!/bin/bash
# usage (on the GPU box): tools/traffic_by_stage.sh <workload> <batch>
set -u
WL=$1; B=$2
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=/tmp/tbs_$WL
rm -rf $OUT; mkdir -p $OUT
CMD="python tools/traffic_by_stage.py --workload $WL --batch $B"
timeout -k 10 300 rocprofv3 --kernel-trace -d $OUT/trace -o t -- $CMD > $OUT/trace.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -o f -- $CMD > $OUT/fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d $OUT/write -o w -- $CMD > $OUT/write.log 2>&1
python - <<PY
import sqlite3, glob, collections
def load(d):
    db = sqlite3.connect(glob.glob(d + "/*.db")[0]); cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    tab = lambda k: [t for t in tabs if k in t][0]
    names = {r[0]: r[1] for r in cur.execute("select id, kernel_name from %s" % tab("rocpd_info_kernel_symbol"))}
    disp = [d for d in cur.execute("select id, kernel_id, start, end, event_id from %s order by start" % tab("rocpd_kernel_dispatch")) if "sf_frame_kernel" in names[d[1]]]
    pmc = collections.defaultdict(float)
    try:
        for ev, val in cur.execute("select event_id, value from %s" % tab("rocpd_pmc_event")):
            pmc[ev] += val
    except Exception:
        pass
    return disp, pmc
dt, _ = load("$OUT/trace"); df, pf = load("$OUT/fetch"); dw, pw = load("$OUT/write")
names = ["pyramid(old)", "pyramid(new)", "kmeans alone (static: pyramid(new) again)", "kmeans+solver (runSolver(false))", "residuals vs history", "segm image", "ring push"]
NS = len(names)
print("workload $WL batch $B: per-launch HBM bytes (reads = 2 x FETCH_SIZE, gfx950), MB per stream")
tot = 0
for k in range(NS):
    rd = 2 * pf[df[-NS + k][4]] * 1024; wr = pw[dw[-NS + k][4]] * 1024; ms = (dt[-NS + k][3] - dt[-NS + k][2]) * 1e-6
    tot += rd + wr
    print("  %-28s %8.3f ms  read %7.3f MB  write %7.3f MB  -> %6.2f TB/s" % (names[k], ms, rd / $B / 1e6, wr / $B / 1e6, (rd + wr) / ms / 1e9))
print("  total %.2f MB per stream (K-means counted twice with segmentation, pyramid(new) twice without)" % (tot / $B / 1e6))
PY
