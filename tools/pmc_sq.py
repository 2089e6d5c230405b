#!/usr/bin/env python
"""SQ / TCC performance counters of the hot kernels (run on the GPU box, from the repo root):
       python tools/pmc_sq.py <tag> [sigma]
   One rocprofv3 --pmc pass per counter group (no trace domains besides --kernel-trace) over
   tools/pmc_workload.py; per-kernel averages go to gpurun_out/<tag>/sq_counters.json and a
   table to gpurun_out/<tag>/sq_counters.txt (copied to profiles/ by hand when it is evidence).
   PMC_GROUPS=0,2,6 selects counter groups, PMC_WORKLOAD=r6/gradc_workload.py another workload under tools/."""
import glob, json, os, sqlite3, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "pmc"
sigma = sys.argv[2] if len(sys.argv) > 2 else "2.0"
out = os.path.join(ROOT, "gpurun_out", tag)
os.makedirs(out, exist_ok=True)
GROUPS = [
    ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY"],
    ["SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS"],
    ["SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_INSTS_VMEM"],
    ["SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_ADDR_CONFLICT", "SQ_INST_CYCLES_VMEM"],
    ["GRBM_GUI_ACTIVE", "TCC_HIT_sum", "TCC_MISS_sum"],
    ["SQC_ICACHE_REQ", "SQC_ICACHE_HITS", "SQC_ICACHE_MISSES", "SQ_IFETCH"],
    ["SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INST_CYCLES_VMEM_RD", "SQ_INST_CYCLES_VMEM_WR"],
    ["SQ_INSTS_LDS_ATOMIC", "SQ_INSTS_LDS_LOAD", "SQ_INSTS_LDS_STORE", "SQ_INSTS_BRANCH"],
    ["TCP_UTCL1_TRANSLATION_MISS_sum", "TCP_UTCL1_TRANSLATION_HIT_sum"],
    ["TCP_TCC_READ_REQ_LATENCY_sum", "TCP_TCC_READ_REQ_sum"],
    ["TCP_PENDING_STALL_CYCLES_sum", "TCP_TOTAL_CACHE_ACCESSES_sum"],
    ["SQ_INST_LEVEL_VMEM", "SQ_INST_LEVEL_LDS", "SQ_LEVEL_WAVES"],
]
Q = """select s.kernel_name, p.name, avg(t.v), count(*) from
         (select e.event_id as ev, e.pmc_id as pid, sum(e.value) as v from rocpd_pmc_event e group by e.event_id, e.pmc_id) t
       join rocpd_info_pmc p on t.pid = p.id
       join rocpd_kernel_dispatch d on t.ev = d.event_id
       join rocpd_info_kernel_symbol s on d.kernel_id = s.id group by s.kernel_name, p.name"""
res = {}
env = dict(os.environ, TMPDIR="/tmp")
only = os.environ.get("PMC_GROUPS")
for i, g in enumerate(GROUPS):
    if only and str(i) not in only.split(","):
        continue
    d = os.path.join(out, "pass%d" % i)
    cmd = ["rocprofv3", "--kernel-trace", "--pmc"] + g + ["-d", d, "--", sys.executable, os.path.join(ROOT, "tools", os.environ.get("PMC_WORKLOAD", "pmc_workload.py")), sigma]
    r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=900)
    open(os.path.join(out, "pass%d.log" % i), "w").write(r.stdout[-4000:] + "\n" + r.stderr[-4000:])
    for db in glob.glob(os.path.join(d, "**", "*.db"), recursive=True):
        try:
            for k, name, v, n in sqlite3.connect(db).execute(Q):
                if any(s in k for s in ("sorted", "window", "tiled", "t2d", "binned", "bricks", "generic", "copyBuffer", "prefilter", "filter", "owner")):
                    short = k.split("(")[0].replace("void ip::", "")[:70]
                    res.setdefault(short, {})[name] = v
        except sqlite3.Error as e:
            print("db error", db, e)
    import shutil
    shutil.rmtree(d, ignore_errors=True)             # (the databases stay on the box: gpurun merges at most 64 MiB back)
json.dump(res, open(os.path.join(out, "sq_counters.json"), "w"), indent=1)
with open(os.path.join(out, "sq_counters.txt"), "w") as f:
    f.write("# rocprofv3 --kernel-trace --pmc <group> -- python tools/pmc_workload.py %s   (tools/pmc_sq.py; per-dispatch averages, summed over instances)\n" % sigma)
    for k, c in res.items():
        f.write("\n%s\n" % k)
        for name in sorted(c):
            f.write("    %-28s %18.0f\n" % (name, c[name]))
        if "SQ_LDS_IDX_ACTIVE" in c and c["SQ_LDS_IDX_ACTIVE"]:
            f.write("    %-28s %18.3f\n" % ("bank_conflict / idx_active", c.get("SQ_LDS_BANK_CONFLICT", 0) / c["SQ_LDS_IDX_ACTIVE"]))
        if "SQ_BUSY_CYCLES" in c and c.get("SQ_ACTIVE_INST_LDS"):
            f.write("    %-28s %18.3f\n" % ("active_inst_lds / busy", c["SQ_ACTIVE_INST_LDS"] / c["SQ_BUSY_CYCLES"]))
            f.write("    %-28s %18.3f\n" % ("active_inst_valu / busy", c.get("SQ_ACTIVE_INST_VALU", 0) / c["SQ_BUSY_CYCLES"]))
print(open(os.path.join(out, "sq_counters.txt")).read())
