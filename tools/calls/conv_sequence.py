import csv, glob, sys
out = sys.argv[1]
rows = []
for f in glob.glob(f"{out}/raw/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"]))
rows.sort()
# find the last conv_in_kernel (start of the last UNet forward) and print every conv kernel after it, with the gap in front
idx = [i for i, r in enumerate(rows) if "conv_in_kernel" in r[2]]
start = idx[-2] if len(idx) > 1 else idx[-1]
end = idx[-1]
prev_end = rows[start][1]
for r in rows[start:end]:
    name = r[2]
    short = "h32<2>" if "conv_h32_kernelILi2" in name or "conv_h32_kernel<2>" in name else "h32<1>" if "conv_h32_kernel" in name else "halo" if "igemm_halo" in name else None
    if short:
        print(f"{short:7s} grid {r[3]}x{r[4]}x{r[5]:3s} {(r[1]-r[0])/1e3:8.1f} us   gap before {(r[0]-prev_end)/1e3:6.1f} us")
    prev_end = r[1]
