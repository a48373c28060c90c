#!/usr/bin/env python3
"""Kernel sequence of the LAST `n` dispatches of a rocprofv3 rocpd database, with start offsets and durations (us).
usage: rocpd_timeline.py results.db [n]"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 80
rows = db.execute("select name, start, end from kernels order by start").fetchall()[-n:]
t0 = rows[0][1]
for name, s, e in rows:
    name = re.sub(r"\(.*", "", name).replace("void ", "").replace("rocprim::ROCPRIM_400200_NS::detail::", "rp::")
    name = re.sub(r"rp::trampoline_kernel<rp::wrapped_(\w+?)_config<.*", r"rocprim \1", name)[:60]
    print("%9.1f  %8.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, name))
print("span %.1f us" % ((rows[-1][2] - t0) / 1e3))
