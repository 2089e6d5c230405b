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
    """databases under gpurun_out/<tag>/<sub>, the most recent first (gpurun merges every run's files into the same directory)"""
    return sorted(glob.glob(os.path.join(src, sub, "**", "*.db"), recursive=True), key=os.path.getmtime, reverse=True)


def counter(db, name, kernel_sub):
    """(sum over the dispatches whose kernel name contains kernel_sub, number of such dispatches) of counter `name`, KB"""
    c = sqlite3.connect(db)
    q = """select s.kernel_name, sum(t.v), count(*) from (select e.event_id as ev, sum(e.value) as v from rocpd_pmc_event e
             join rocpd_info_pmc p on e.pmc_id = p.id where p.name = ? group by e.event_id) t
           join rocpd_kernel_dispatch d on t.ev = d.event_id
           join rocpd_info_kernel_symbol s on d.kernel_id = s.id group by s.kernel_name"""
    tot, n = 0.0, 0
    for k, v, cnt in c.execute(q, (name,)):
        if kernel_sub in k:
            tot += v; n += cnt
    return [tot, n]


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
ALG_READ = 4 * 256 ** 3 * 12 + 4 * 2 * 256 ** 3 * 4     # algorithmic HBM read bytes per launch at BASELINE config 2: grid + source
# kernels of one grid_pull / grid_push launch (substrings of the kernel names); the first one counts the launches
KERNELS = {"grid_pull": ["pull_sorted"],
           "grid_push": ["own_probe", "own_bin", "own_accumulate", "own_zero", "push_tiled", "fillBuffer", "zero_fill"]}   # (own_bin also runs,
           # nearly empty, in index mode behind every routed pull since round 4: the probe counts the push launches)
raw_path = os.path.join(src, "pmc_raw.json")
if all(dbs("pmc_%s_s%s" % (c, sg)) for c in ("FETCH_SIZE", "WRITE_SIZE") for sg in ("2.0", "0.0")):
    raw = {}
    for sg in ("2.0", "0.0"):
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            db = dbs("pmc_%s_s%s" % (c, sg))[0]
            for sub in sorted(set(sum(KERNELS.values(), []))) + ["copyBuffer"]:
                raw["%s|%s|%s" % (sub, c, sg)] = counter(db, c, sub)      # [KB summed over the dispatches, dispatches]
    json.dump(raw, open(raw_path, "w"), indent=1)
if os.path.exists(raw_path):
    raw = json.load(open(raw_path))
    # the raw counters travel with the table built from them: <round>_pmc_hbm_traffic.txt is reproducible from <round>_pmc_raw.json
    json.dump(raw, open(os.path.join(dst, rnd + "_pmc_raw.json"), "w"), indent=1)

    def per_launch(sub, c, sg, op):
        """KB of counter c per launch of `op` spent in the kernels named sub"""
        tot, _ = raw.get("%s|%s|%s" % (sub, c, sg), [0.0, 0])
        n = raw.get("%s|%s|%s" % (KERNELS[op][0] if raw.get("%s|%s|%s" % (KERNELS[op][0], c, sg), [0, 0])[1] else "push_tiled", c, sg), [0, 0])[1]
        if op == "grid_push":
            # (round 5: the routed pull launches own_zero / own_probe / own_bin as well -- their traffic is negligible, but they
            #  cannot count the push launches any more: a push is nine own_accumulate launches, or one push_tiled at the identity)
            na = raw.get("own_accumulate|%s|%s" % (c, sg), [0, 0])[1]
            n = na // 9 if na else raw.get("push_tiled|%s|%s" % (c, sg), [0, 0])[1]
        return tot / n if n else 0.0

    buf = io.StringIO()
    buf.write("# rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python tools/pmc_workload.py <sigma>\n"
              "# KB per grid_pull / grid_push LAUNCH, summed over the counter's instances and over the kernels of the launch\n"
              "# (tools/profile_round.sh %s pmc; raw sums and dispatch counts: profiles/%s_pmc_raw.json; this table: tools/make_profiles.py).\n"
              "# Corrections (MI355X_MICROARCH.md sec. HBM: FETCH_SIZE reports 1/2 of wide coalesced reads on gfx950, other widths and\n"
              "# WRITE_SIZE are to be calibrated on a known byte count in one's own access pattern):\n"
              "#  * WRITE_SIZE: exact -- __amd_rocclr_copyBuffer of the same run writes 524288 KB and reports 524288.0 KB: factor 1.\n"
              "#  * FETCH_SIZE of pull_sorted: the SAME kernel at the identity deformation (sigma = 0), where every input byte is fetched\n"
              "#    once and the reads are grid + source = 1342 MB: factor = 1342 MB / counter(sigma = 0) (16-byte staging loads, counted\n"
              "#    1/2, mixed with 4-byte coordinate loads).\n"
              "#  * FETCH_SIZE of the push: at sigma = 2 the launch runs the owner-computes kernels (csrc/push_owner.hip; push_tiled returns\n"
              "#    at its gate).  own_bin reads grid + source exactly once (1342 MB, 12- and 4-byte loads): its factor is 1342 MB / its\n"
              "#    counter; own_accumulate (16-byte record gathers, 2- / 4-byte loads) is given the same factor -- an assumption; were its\n"
              "#    reads counted 1/2 like wide streaming reads, its read bytes would double (the 'x2' column).\n"
              "# The push moves more than its algorithmic bytes by design: own_bin writes 22 bytes per sample of sorted records that\n"
              "# own_accumulate reads back, and the target is flushed brick by brick with loads + stores (shell bricks: float atomics,\n"
              "# executed memory-side and counted as writes); the zero-fill of the target (zero_fill, a kernel of the library; fillBuffer where a memset served it) is part of the launch.\n" % (tag, rnd))
    out = {"_comment": "HBM-side bytes per launch at BASELINE config 2 (sigma = 2) from rocprofv3 PMC passes (profiles/%s_pmc_hbm_traffic.txt): "
                       "FETCH_SIZE KB x 1024 x calibration factor + WRITE_SIZE KB x 1024, summed over the kernels of the launch" % rnd}
    buf.write("%-10s %-15s %14s %14s %8s %14s %14s\n" % ("op", "kernel", "FETCH KB/launch", "WRITE KB/launch", "f_fetch", "read bytes", "write bytes"))
    # pull
    f0, f2 = per_launch("pull_sorted", "FETCH_SIZE", "0.0", "grid_pull"), per_launch("pull_sorted", "FETCH_SIZE", "2.0", "grid_pull")
    w2 = per_launch("pull_sorted", "WRITE_SIZE", "2.0", "grid_pull")
    if f0 and f2:
        fac = ALG_READ / (f0 * 1024.0)
        rd, wr = f2 * 1024.0 * fac, w2 * 1024.0
        buf.write("%-10s %-15s %14.1f %14.1f %8.3f %14d %14d\n" % ("grid_pull", "pull_sorted", f2, w2, fac, rd, wr))
        out["grid_pull"] = int(rd + wr)
        out["grid_pull_detail"] = {"read_bytes": int(rd), "write_bytes": int(wr), "fetch_factor": round(fac, 3), "fetch_KB_sigma0": round(f0, 1)}
    # push: every kernel of the launch
    fb = per_launch("own_bin", "FETCH_SIZE", "2.0", "grid_push")
    fac_bin = ALG_READ / (fb * 1024.0) if fb else 1.0
    rd_tot = wr_tot = rd_hi = 0.0
    detail = {}
    for sub in KERNELS["grid_push"]:
        fk, wk = per_launch(sub, "FETCH_SIZE", "2.0", "grid_push"), per_launch(sub, "WRITE_SIZE", "2.0", "grid_push")
        if not fk and not wk:
            continue
        fac = fac_bin if sub.startswith("own_") else 1.0
        rd, wr = fk * 1024.0 * fac, wk * 1024.0
        buf.write("%-10s %-15s %14.1f %14.1f %8.3f %14d %14d\n" % ("grid_push", sub, fk, wk, fac, rd, wr))
        rd_tot += rd; wr_tot += wr; rd_hi += rd * (2.0 if sub == "own_accumulate" else 1.0)
        detail[sub] = {"read_bytes": int(rd), "write_bytes": int(wr)}
    if rd_tot + wr_tot:
        buf.write("%-10s %-15s %14s %14s %8s %14d %14d   (x2 on own_accumulate's reads: %d)\n" % ("grid_push", "launch", "", "", "", rd_tot, wr_tot, rd_hi + wr_tot))
        out["grid_push"] = int(rd_tot + wr_tot)
        out["grid_push_detail"] = dict(detail, fetch_factor_own=round(fac_bin, 3), read_bytes=int(rd_tot), write_bytes=int(wr_tot),
                                       upper_bound_if_own_accumulate_reads_count_half=int(rd_hi + wr_tot))
    # the routed push at the identity (push_tiled) for reference, and the copy calibration
    f0, w0 = per_launch("push_tiled", "FETCH_SIZE", "0.0", "grid_push"), per_launch("push_tiled", "WRITE_SIZE", "0.0", "grid_push")
    if f0:
        buf.write("%-10s %-15s %14.1f %14.1f   (sigma = 0: the probe keeps the tiles; 1342 MB of reads, target 524288 KB)\n" % ("grid_push", "push_tiled s=0", f0, w0))
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        v = raw.get("copyBuffer|%s|2.0" % c)
        if v and v[1]:
            buf.write("%-10s %-15s %14s %14.1f   (%s per dispatch; copies 524288 KB with 16 B/lane accesses)\n" % ("copy", "copyBuffer", "", v[0] / v[1], c))
    open(os.path.join(dst, rnd + "_pmc_hbm_traffic.txt"), "w").write(buf.getvalue())
    json.dump(out, open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1)
for name, target in (("phase_split.txt", "_phase_split.txt"), ("micro_lds_gather.txt", "_micro_lds_gather.txt"), ("micro_lds_atomics.txt", "_micro_lds_atomics.txt"),
                     (os.path.join("sq_cfg2", "sq_counters.txt"), "_sq_counters_cfg2.txt"), (os.path.join("sq_cfg5", "sq_counters.txt"), "_sq_counters_cfg5.txt")):
    pth = os.path.join(src, name)
    if os.path.exists(pth) and os.path.getsize(pth):
        open(os.path.join(dst, rnd + target), "w").write(open(pth).read())
print(os.listdir(dst))
