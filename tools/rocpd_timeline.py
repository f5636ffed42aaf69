#!/usr/bin/env python
"""GPU timeline of a rocprofv3 kernel trace (rocpd sqlite): over the LAST `frac` of the trace (steady state) — wall span, union of kernel busy time,
idle share, the gap histogram between consecutive kernels and the kernels in front of the longest gaps; + memory copies if the table exists.
usage: tools/rocpd_timeline.py <results.db> [frac=0.5]"""
import sqlite3
import sys
from collections import Counter


def main():
    db = sqlite3.connect(sys.argv[1])
    frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)").fetchall()]
    q = "stream_id" if "stream_id" in cols else ("queue_id" if "queue_id" in cols else "0")
    rows = cur.execute(f"select start, end, name, {q} from kernels order by start").fetchall()
    t0, t1 = rows[0][0], rows[-1][1]
    lo = t1 - (t1 - t0) * frac
    rows = [r for r in rows if r[0] >= lo]
    span = rows[-1][1] - rows[0][0]
    busy, cur_end, overlap = 0, rows[0][0], 0
    gaps = []
    for s, e, name, qid in rows:
        if s >= cur_end:
            if cur_end > rows[0][0]:
                gaps.append((s - cur_end, prev, name))
            busy += e - s
            cur_end = e
        else:
            overlap += min(e, cur_end) - s
            if e > cur_end:
                busy += e - cur_end
                cur_end = e
        prev = name
    print(f"kernels {len(rows)}  span {span / 1e6:.3f} ms  busy(union) {busy / 1e6:.3f} ms  idle {100 * (1 - busy / span):.1f} %  "
          f"sum(durations) {sum(e - s for s, e, _, _ in rows) / 1e6:.3f} ms  overlapped {overlap / 1e6:.3f} ms  queues {len({r[3] for r in rows})}")
    hist = Counter()
    for g, _, _ in gaps:
        b = "<2us" if g < 2e3 else "2-5us" if g < 5e3 else "5-10us" if g < 1e4 else "10-50us" if g < 5e4 else "50-200us" if g < 2e5 else "0.2-1ms" if g < 1e6 else ">1ms"
        hist[b] += g
    for b in ("<2us", "2-5us", "5-10us", "10-50us", "50-200us", "0.2-1ms", ">1ms"):
        n = sum(1 for g, _, _ in gaps if (b == "<2us" and g < 2e3) or (b == "2-5us" and 2e3 <= g < 5e3) or (b == "5-10us" and 5e3 <= g < 1e4) or
                (b == "10-50us" and 1e4 <= g < 5e4) or (b == "50-200us" and 5e4 <= g < 2e5) or (b == "0.2-1ms" and 2e5 <= g < 1e6) or (b == ">1ms" and g >= 1e6))
        print(f"  gaps {b:9s}: {n:6d}  total {hist[b] / 1e6:8.3f} ms ({100 * hist[b] / span:5.1f} % of the span)")
    print("longest gaps (us, kernel before -> kernel after):")
    for g, a, b in sorted(gaps, reverse=True)[:12]:
        print(f"  {g / 1e3:9.1f}  {a[:60]} -> {b[:60]}")
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')").fetchall()]
    for t in tabs:
        if "memory_cop" in t.lower():
            try:
                c = cur.execute(f"select count(*), sum(end - start) from {t} where start >= ?", (lo,)).fetchone()
                print(f"{t}: {c[0]} copies, {0 if c[1] is None else c[1] / 1e6:.3f} ms")
            except sqlite3.Error as e:
                print(t, e)


if __name__ == "__main__":
    main()
