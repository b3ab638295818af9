"""the copies of at least 50 us of a rocprofv3 --memory-copy-trace CSV, in start order: start (ms), duration (us), the other columns"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"]) if rows else 0
for r in rows:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    if d > 50000:
        print(f"{(int(r['Start_Timestamp']) - t0) / 1e6:10.3f} ms {d / 1e3:9.1f} us ", {k: v for k, v in r.items() if k not in ("Start_Timestamp", "End_Timestamp")})
