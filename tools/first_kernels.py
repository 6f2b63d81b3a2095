import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "multi_tensor_apply" in r["Kernel_Name"]]
groups = []
for i in idx:
    if not groups or i - groups[-1][-1] > 50: groups.append([i])
    else: groups[-1].append(i)
a, b = groups[-2][-1], groups[-1][0]
seg = rows[a + 1:b]
for r in seg[:int(sys.argv[2])]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    print(f'{r["Kernel_Name"][:70]:70s} grid {r.get("Grid_Size_X", r.get("Grid_Size","?")):>8s} wg {r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?")):>5s} {d:8.1f} us')
