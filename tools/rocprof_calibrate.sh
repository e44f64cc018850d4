#!/bin/bash
# Calibrate FETCH_SIZE on this access pattern: the "loads only" variant of the isolated IRLS pass
# reads exactly 29 B per pixel (7 x 8-byte pair loads + 2 label bytes per lane and pixel pair).
# Run ON THE GPU BOX. usage: tools/rocprof_calibrate.sh <tag>
set -u
TAG=$1
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/calib_$TAG
rm -rf $OUT; mkdir -p $OUT
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -o fetch -- python tools/pass_microbench.py --batch 512 --reps 10 > $OUT/fetch.log 2>&1
python - <<PY
import sqlite3, glob, collections
db = sqlite3.connect(glob.glob("$OUT/fetch/*.db")[0]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
tab = lambda k: [t for t in tabs if k in t][0]
names = {r[0]: r[1] for r in cur.execute("select id, kernel_name from %s" % tab("rocpd_info_kernel_symbol"))}
disp = list(cur.execute("select id, kernel_id, start, end, event_id from %s order by start" % tab("rocpd_kernel_dispatch")))
pmc = collections.defaultdict(float)
for ev, val in cur.execute("select event_id, value from %s" % tab("rocpd_pmc_event")):
    pmc[ev] += val
mine = [d for d in disp if "sf_irls_pass_kernel" in names[d[1]]]
# order of launches in pass_microbench.py: for pass in (1,2): variants (product, loads, noacc, [fp32]) x (2 reps warm, 10 reps timed)
labels = ["p1 product", "p1 loads", "p1 noacc", "p2 product", "p2 loads", "p2 noacc"]
px = lambda reps: 512 * reps * 76800
for k, d in enumerate(mine):
    reps = 2 if k % 2 == 0 else 10
    exp = 29.0 * px(reps)
    print("%-11s reps %2d  FETCH_SIZE raw %8.1f MB (KB units) expected %8.1f MB  ratio raw/expected %.3f  dur %.3f ms" % (
        labels[k // 2] if k // 2 < len(labels) else "?", reps, pmc[d[4]] * 1024 / 1e6, exp / 1e6, pmc[d[4]] * 1024 / exp, (d[3] - d[2]) * 1e-6))
PY
