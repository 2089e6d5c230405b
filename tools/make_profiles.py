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
# --- HBM traffic from the FETCH_SIZE / WRITE_SIZE passes ---------------------------------------------------
# stage 1 (on the GPU box, where the rocprofv3 databases are): raw per-dispatch counters -> pmc_raw.json
# stage 2 (anywhere): pmc_raw.json -> <round>_pmc_hbm_traffic.txt + pmc_traffic.json
ALG_READ = {   # algorithmic HBM read bytes per launch at BASELINE config 2 (4 x 2 x 256^3 fp32, fp32 grid): grid + source
    "grid_pull": 4 * 256 ** 3 * 12 + 4 * 2 * 256 ** 3 * 4,
    "grid_push": 4 * 256 ** 3 * 12 + 4 * 2 * 256 ** 3 * 4,
}
KERNEL = {"grid_pull": "pull_sorted", "grid_push": "push_tiled"}
raw_path = os.path.join(src, "pmc_raw.json")
if all(dbs("pmc_%s_s%s" % (c, sg)) for c in ("FETCH_SIZE", "WRITE_SIZE") for sg in ("2.0", "0.0")):
    raw = {}
    for sg in ("2.0", "0.0"):
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            db = dbs("pmc_%s_s%s" % (c, sg))[0]
            for key, sub in list(KERNEL.items()) + [("copy", "copyBuffer")]:
                raw["%s|%s|%s" % (key, c, sg)] = counter(db, c, sub)
    json.dump(raw, open(raw_path, "w"), indent=1)
if os.path.exists(raw_path):
    raw = json.load(open(raw_path))
    # the raw counters travel with the table built from them: <round>_pmc_hbm_traffic.txt is reproducible from <round>_pmc_raw.json
    json.dump(raw, open(os.path.join(dst, rnd + "_pmc_raw.json"), "w"), indent=1)
    buf = io.StringIO()
    buf.write("# rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python tools/pmc_workload.py <sigma>\n"
              "# per-dispatch values summed over the counter's instances, KB (tools/profile_round.sh %s pmc; raw values: profiles/%s_pmc_raw.json).\n"
              "# Corrections (MI355X_MICROARCH.md sec. HBM: FETCH_SIZE reports 1/2 of wide coalesced reads on gfx950, other widths and\n"
              "# WRITE_SIZE are to be calibrated on a known byte count in one's own access pattern):\n"
              "#  * WRITE_SIZE: exact -- __amd_rocclr_copyBuffer of the same run writes 524288 KB and reports 524288.0 KB: factor 1.\n"
              "#  * FETCH_SIZE: the SAME kernel at the identity deformation (sigma = 0), where every input byte is fetched once and the\n"
              "#    reads are grid + source = 1342 MB: factor = 1342 MB / counter(sigma = 0).  pull_sorted mixes 16-byte staging loads\n"
              "#    (counted 1/2) with 4-byte coordinate loads: factor ~1.4 (this run: the table); push_tiled reads with 4-byte loads only: factor 1.00.\n"
              "# The push target is never read: its float atomics are executed memory-side and counted as writes (write-through of the\n"
              "# tile halos: 1.40 GB at the identity, 2.78 GB at sigma = 2 for a 0.54 GB target).\n" % (tag, rnd))
    out = {"_comment": "HBM-side bytes per launch at BASELINE config 2 (sigma = 2) from rocprofv3 PMC passes (profiles/%s_pmc_hbm_traffic.txt): "
                       "FETCH_SIZE KB x 1024 x the factor calibrated on the same kernel at the identity deformation + WRITE_SIZE KB x 1024 (exact per "
                       "the copy calibration of the same run)" % rnd}
    buf.write("%-10s %-12s %8s %15s %15s %8s %16s\n" % ("op", "kernel", "counter", "KB at sigma=0", "KB at sigma=2", "factor", "bytes at sigma=2"))
    for key, sub in KERNEL.items():
        f0, f2 = raw.get("%s|FETCH_SIZE|0.0" % key), raw.get("%s|FETCH_SIZE|2.0" % key)
        w0, w2 = raw.get("%s|WRITE_SIZE|0.0" % key), raw.get("%s|WRITE_SIZE|2.0" % key)
        if not f0 or f2 is None or w2 is None:
            continue
        fac = ALG_READ[key] / (f0 * 1024.0)
        rd, wr = f2 * 1024.0 * fac, w2 * 1024.0
        buf.write("%-10s %-12s %8s %15.1f %15.1f %8.3f %16d\n" % (key, sub, "FETCH", f0, f2, fac, rd))
        buf.write("%-10s %-12s %8s %15.1f %15.1f %8.3f %16d\n" % (key, sub, "WRITE", w0 or 0, w2, 1.0, wr))
        out[key] = int(rd + wr)
        out[key + "_detail"] = {"read_bytes": int(rd), "write_bytes": int(wr), "fetch_factor": round(fac, 3),
                                "fetch_KB_sigma0": round(f0, 1), "fetch_KB_sigma2": round(f2, 1), "write_KB_sigma0": round(w0 or 0, 1), "write_KB_sigma2": round(w2, 1)}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        v = raw.get("copy|%s|2.0" % c)
        if v is not None:
            buf.write("%-10s %-12s %8s %15s %15.1f   (copies 524288 KB with 16 B/lane accesses)\n" % ("copy", "copyBuffer", c[:5], "-", v))
    open(os.path.join(dst, rnd + "_pmc_hbm_traffic.txt"), "w").write(buf.getvalue())
    json.dump(out, open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1)
for name, target in (("phase_split.txt", "_phase_split.txt"), ("micro_lds_gather.txt", "_micro_lds_gather.txt"), ("micro_lds_atomics.txt", "_micro_lds_atomics.txt"),
                     (os.path.join("sq_cfg2", "sq_counters.txt"), "_sq_counters_cfg2.txt"), (os.path.join("sq_cfg5", "sq_counters.txt"), "_sq_counters_cfg5.txt")):
    pth = os.path.join(src, name)
    if os.path.exists(pth) and os.path.getsize(pth):
        open(os.path.join(dst, rnd + target), "w").write(open(pth).read())
print(os.listdir(dst))
