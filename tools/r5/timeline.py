#!/usr/bin/env python
"""Timeline of the last `n` kernel dispatches of a rocprofv3 rocpd database: name, start offset, duration, gap to the previous end (us)."""
import sqlite3, sys
db, n = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 24
c = sqlite3.connect(db)
rows = list(c.execute("""select s.kernel_name, d.start, d.end from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s
                         on d.kernel_id = s.id order by d.start"""))[-n:]
t0, prev = rows[0][1], None
for name, a, b in rows:
    short = name.split("(")[0].replace("void ", "")
    for cut in ("ip5owner", "ip6sorted", "ip5tiled"):
        if cut in short:
            short = short[short.index(cut):]
    print("%-46s start %9.1f  dur %8.1f  gap %7.1f" % (short[:46], (a - t0) / 1e3, (b - a) / 1e3, ((a - prev) / 1e3) if prev else 0.0))
    prev = b
