"""print one step of a rocprofv3 kernel trace around the n-th launch of a kernel: name, queue, start (us), duration (us)"""
import csv, sys
path, key = sys.argv[1], sys.argv[2]
nth = int(sys.argv[3]) if len(sys.argv) > 3 else 5
count = int(sys.argv[4]) if len(sys.argv) > 4 else 14
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if key in r["Kernel_Name"]]
i0 = max(idx[nth] - 3, 0)
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:i0 + count]:
    print(r["Kernel_Name"][:56].ljust(56), r["Queue_Id"], round((int(r["Start_Timestamp"]) - t0) / 1e3, 1), round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, 1))
