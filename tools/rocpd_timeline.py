#!/usr/bin/env python3
"""Kernel sequence of the LAST `n` dispatches of a rocprofv3 rocpd database, with start offsets, durations (us) and,
where the view has them, the queue / stream ids (to see what overlaps with what).
usage: rocpd_timeline.py results.db [n]"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 80
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
extra = [c for c in ("queue_id", "stream_id") if c in cols]
rows = db.execute("select name, start, end %s from kernels order by start" % "".join(", " + c for c in extra)).fetchall()[-n:]
t0 = rows[0][1]
for row in rows:
    name, s, e = row[:3]
    name = re.sub(r"\(.*", "", name).replace("void ", "").replace("cg::", "")[:48]
    print("%9.1f  %8.1f  %9.1f  %-48s %s" % ((s - t0) / 1e3, (e - s) / 1e3, (e - t0) / 1e3, name, " ".join(str(x) for x in row[3:])))
print("span %.1f us; columns: start, duration, end, kernel, %s" % ((max(r[2] for r in rows) - t0) / 1e3, " ".join(extra)))
