#!/usr/bin/env python
"""Condense rocprofv3 output (rocpd sqlite .db or *_kernel_stats.csv / *_counter_collection.csv)
into the short per-kernel summaries committed under profiles/."""
import csv, glob, os, sqlite3, sys


def from_db(path, out):
    c = sqlite3.connect(path)
    q = """select s.kernel_name, count(*), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start), sum(d.end-d.start),
                  max(s.arch_vgpr_count), max(s.sgpr_count), max(d.group_segment_size), max(d.private_segment_size)
           from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id=s.id
           group by s.kernel_name order by 6 desc"""
    rows = list(c.execute(q))
    tot = sum(r[5] for r in rows) or 1
    out.write("%-120s %6s %10s %10s %10s %7s %5s %5s %7s %7s\n" % ("kernel", "calls", "avg_us", "min_us", "max_us", "pct", "vgpr", "sgpr", "lds_B", "scr_B"))
    for r in rows[:25]:
        out.write("%-120s %6d %10.1f %10.1f %10.1f %6.1f%% %5s %5s %7s %7s\n" % (
            r[0][:120], r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3, 100.0 * r[5] / tot, r[6], r[7], r[8], r[9]))
    # counters, if any
    try:
        q = """select s.kernel_name, p.name, avg(e.value), count(*) from rocpd_pmc_event e
               join rocpd_info_pmc p on e.pmc_id = p.id
               join rocpd_kernel_dispatch d on e.event_id = d.event_id
               join rocpd_info_kernel_symbol s on d.kernel_id = s.id
               group by s.kernel_name, p.name order by 1, 2"""
        rows = list(c.execute(q))
        if rows:
            out.write("\n%-120s %-24s %16s %6s\n" % ("kernel", "counter", "avg_per_dispatch", "n"))
            for r in rows:
                out.write("%-120s %-24s %16.1f %6d\n" % (r[0][:120], r[1], r[2], r[3]))
    except sqlite3.Error as e:
        out.write("\n(no counter tables: %s)\n" % e)


def dispatches(path, pattern, out):
    c = sqlite3.connect(path)
    q = """select s.kernel_name, d.start, d.end from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id=s.id
           where s.kernel_name like ? order by d.start"""
    out.write("# dispatches of *%s* in launch order (us)\n" % pattern)
    out.write(" ".join("%.0f" % ((r[2] - r[1]) / 1e3) for r in c.execute(q, ("%" + pattern + "%",))) + "\n")


if __name__ == "__main__":
    if len(sys.argv) > 3 and sys.argv[1] == "--dispatches":
        dispatches(sys.argv[3], sys.argv[2], sys.stdout)
        sys.exit(0)
    out = sys.stdout
    for path in sys.argv[1:]:
        out.write("# %s\n" % path)
        from_db(path, out)
        out.write("\n")
