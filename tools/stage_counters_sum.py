"""Sum the --pmc counters of the last N dispatches of the frame kernel in every pass directory (rocpd databases)."""
import sys, os, glob, sqlite3, collections, json
root, last = sys.argv[1], int(sys.argv[2])
res = {}
for d in sorted(glob.glob(os.path.join(root, "*"))):
    if not os.path.isdir(d):
        continue
    dbs = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)
    if not dbs:
        continue
    db = sqlite3.connect(dbs[0]); cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    tab = lambda key: [t for t in tabs if key in t][0]
    kd, ks = tab("rocpd_kernel_dispatch"), tab("rocpd_info_kernel_symbol")
    names = {r[0]: r[1] for r in cur.execute("select id, kernel_name from %s" % ks)}
    disp = [r for r in cur.execute("select id, kernel_id, start, end, event_id from %s order by start" % kd) if "sf_frame_kernel" in names[r[1]]]
    mine = disp[-last:]
    if os.path.basename(d) == "trace":
        res["duration_ms"] = [(r[3] - r[2]) * 1e-6 for r in mine]
        continue
    pe, pi = tab("rocpd_pmc_event"), tab("rocpd_info_pmc")
    pn = {r[0]: r[1] for r in cur.execute("select id, name from %s" % pi)}
    ev = {r[4] for r in mine}
    acc = collections.defaultdict(float)
    for e, pid, val in cur.execute("select event_id, pmc_id, value from %s" % pe):
        if e in ev:
            acc[pn[pid]] += val
    for k, v in acc.items():
        res[k] = v / last
print(json.dumps(res, indent=1))
json.dump(res, open(os.path.join(root, "counters.json"), "w"), indent=1)
