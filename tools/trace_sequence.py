"""Dispatch sequence of ONE build from a rocprofv3 --kernel-trace CSV: every kernel launch between two k_byte_hist
launches with its duration, consecutive launches of the same kernel merged.  python tools/trace_sequence.py trace.csv [build#]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
which = int(sys.argv[2]) if len(sys.argv) > 2 else -2
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "k_byte_hist" in r["Kernel_Name"]]
b = starts[which]
e = starts[which + 1] if which + 1 < len(starts) and which != -1 else len(rows)
def short(n):
    n = re.sub(r"\(.*", "", n).replace("void sa::", "").replace("sa::", "")
    return n[:70]
seq, t0 = [], int(rows[b]["Start_Timestamp"])
for r in rows[b:e]:
    nm, d = short(r["Kernel_Name"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if seq and seq[-1][0] == nm:
        seq[-1][1] += d; seq[-1][2] += 1
    else:
        seq.append([nm, d, 1, (int(r["Start_Timestamp"]) - t0) / 1e6, int(r["Grid_Size_X"])])
for nm, d, c, at, grid in seq:
    print(f"{at:8.2f} ms  {nm:70s} x{c:<3d} {d:9.1f} us  grid {grid}")
print("total kernel time", sum(x[1] for x in seq) / 1e3, "ms; span", (int(rows[e - 1]["End_Timestamp"]) - t0) / 1e6, "ms")
