import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = len(rows); mid = n * 2 // 3
prev = None
for r in rows[mid:mid + int(sys.argv[2]) if len(sys.argv) > 2 else mid + 45]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev) / 1e3 if prev else 0
    print("%8.1f us gap %8.1f us  %s" % (gap, (e - s) / 1e3, r["Kernel_Name"][:60]))
    prev = e
