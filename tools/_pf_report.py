import csv, glob, sys
cfgs = sys.argv[2:]
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "stream" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), r["Kernel_Name"].split("::")[-1][:34], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
rows.sort()
for i in range(0, len(rows), 2):
    c = cfgs[i // 2] if i // 2 < len(cfgs) else "?"
    print(c, " | ".join(f"{n} {t:.3f}" for _, n, t in rows[i:i + 2]))
