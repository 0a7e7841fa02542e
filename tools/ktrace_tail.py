"""List the kernels of the last graph replay (from the last weight_prep launch on) of a rocprofv3 kernel trace."""
import csv, sys, glob, os
files = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)
rows = list(csv.DictReader(open(files[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "prep" in r["Kernel_Name"]]
start = idx[-1] if idx else max(0, len(rows) - 150)
with open(sys.argv[2], "w") as out:
    for r in rows[start:]:
        out.write(f"{r['Kernel_Name'][:160]} grid={r['Grid_Size_X']},{r['Grid_Size_Y']},{r['Grid_Size_Z']} wg={r['Workgroup_Size_X']} "
                  f"lds={r.get('LDS_Block_Size','')} q={r.get('Queue_Id','')} t={int(r['Start_Timestamp']) - int(rows[start]['Start_Timestamp'])}+{int(r['End_Timestamp']) - int(r['Start_Timestamp'])}\n")
print(len(rows), start, files[0])
