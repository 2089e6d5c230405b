#!/usr/bin/env python
"""Condense gpurun_out/<tag>/ (tools/profile_round.sh) into the tracked files under profiles/:
   <round>_kernel_stats.txt, <round>_pmc_hbm_traffic.txt, pmc_traffic.json, <round>_bench.json,
   <round>_other_configs.json.   usage: tools/make_profiles.py <tag> <round>"""
import glob, io, json, os, sqlite3, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import prof_summary

tag, rnd = sys.argv[1], sys.argv[2]
src = os.path.join(ROOT, "gpurun_out", tag)
dst = os.path.join(ROOT, "profiles")


def dbs(sub):
    return sorted(glob.glob(os.path.join(src, sub, "**", "*.db"), recursive=True))


def counter(db, name, kernel_sub):
    c = sqlite3.connect(db)
    q = """select s.kernel_name, avg(t.v) from (select e.event_id as ev, sum(e.value) as v from rocpd_pmc_event e
             join rocpd_info_pmc p on e.pmc_id = p.id where p.name = ? group by e.event_id) t
           join rocpd_kernel_dispatch d on t.ev = d.event_id
           join rocpd_info_kernel_symbol s on d.kernel_id = s.id group by s.kernel_name"""
    for k, v in c.execute(q, (name,)):
        if kernel_sub in k:
            return v
    return None


stats = dbs("stats")
if stats:
    buf = io.StringIO()
    buf.write("# rocprofv3 --kernel-trace --stats -- python bench.py --no-extras   (tools/profile_round.sh %s bench)\n" % tag)
    prof_summary.from_db(stats[0], buf)
    open(os.path.join(dst, rnd + "_kernel_stats.txt"), "w").write(buf.getvalue())
for name in ("bench.json", "other_configs.json"):
    p = os.path.join(src, name)
    if os.path.exists(p) and os.path.getsize(p):
        text = open(p).read()
        text = text[text.index("{"):]
        try:
            json.loads(text)                               # one (possibly pretty-printed) object
        except ValueError:
            text = "\n".join(l for l in text.splitlines() if l.startswith("{"))
            [json.loads(l) for l in text.splitlines()]     # or one object per line
        open(os.path.join(dst, rnd + "_" + name), "w").write(text.rstrip() + "\n")
f, w = dbs("pmc_FETCH_SIZE"), dbs("pmc_WRITE_SIZE")
if f and w:
    buf = io.StringIO()
    buf.write("# rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) -- python tools/pmc_workload.py\n"
              "# per-dispatch values summed over the counter's instances, KB.  Calibration in the same run:\n"
              "#   __amd_rocclr_copyBuffer copies 524288 KB with 16 B/lane loads: FETCH_SIZE reports 1/2 of the bytes (gfx950\n"
              "#   under-count, MI355X_MICROARCH.md sec. HBM); the 4 B/lane elementwise add reading 786432 KB reports ~786470 KB\n"
              "#   (factor 1); WRITE_SIZE is exact.  The tiled kernels stage with 4 B/lane loads -> factor 1 is used.\n")
    out = {"_comment": "HBM-side bytes per launch at BASELINE config 2 from rocprofv3 PMC passes (profiles/%s_pmc_hbm_traffic.txt): "
                       "FETCH_SIZE*1024 (factor 1, calibrated in the same run) + WRITE_SIZE*1024" % rnd}
    rows = []
    for key, sub in (("grid_pull", "pull2_tiled"), ("grid_push", "push_tiled"), ("copy_calibration_16B_per_lane", "copyBuffer"),
                     ("add_calibration_4B_per_lane", "CUDAFunctor_add")):
        fk, wk = counter(f[0], "FETCH_SIZE", sub), counter(w[0], "WRITE_SIZE", sub)
        rows.append((key, sub, fk, wk))
        if key.startswith("grid_") and fk is not None and wk is not None:
            out[key] = int((fk + wk) * 1024)
            out[key + "_detail"] = {"fetch_KB": round(fk, 1), "write_KB": round(wk, 1)}
    buf.write("%-32s %-20s %16s %16s\n" % ("what", "kernel contains", "FETCH_SIZE_KB", "WRITE_SIZE_KB"))
    for key, sub, fk, wk in rows:
        buf.write("%-32s %-20s %16s %16s\n" % (key, sub, "%.1f" % fk if fk is not None else "-", "%.1f" % wk if wk is not None else "-"))
    open(os.path.join(dst, rnd + "_pmc_hbm_traffic.txt"), "w").write(buf.getvalue())
    json.dump(out, open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1)
print(os.listdir(dst))
