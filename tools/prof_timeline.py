#!/usr/bin/env python3
"""Timeline of the last N kernel dispatches of a rocprofv3 kernel trace (rocpd sqlite): start (ms after the first listed), duration, stream, kernel.
usage: prof_timeline.py <results.db> [N]"""
import sqlite3, sys
sys.path.insert(0, __import__("os").path.dirname(__file__))
from prof_summary import short  # noqa
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
sid = "stream_id" if "stream_id" in cols else ("queue_id" if "queue_id" in cols else "0")
rows = cur.execute("select start, end, %s, name from kernels order by start desc limit %d" % (sid, n)).fetchall()[::-1]
t0 = rows[0][0]
for s, e, q, nm in rows:
    print("%9.3f ms  +%8.3f ms  stream %-4s %s" % ((s - t0) / 1e6, (e - s) / 1e6, q, short(nm)))
