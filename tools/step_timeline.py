"""Print the kernel timeline of the last train step found in a rocprofv3 --kernel-trace csv (development aid).
   python tools/step_timeline.py gpurun_out/<dir> [min_us]"""
import csv, glob, sys
d = sys.argv[1]; min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 14.0
f = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'adam_kernel' in r['Kernel_Name']]
a, b = idx[-2], idx[-1]
t0 = int(rows[a]['End_Timestamp'])
for r in rows[a + 1:b + 1]:
    s = int(r['Start_Timestamp']); e = int(r['End_Timestamp']); dur = (e - s) / 1e3
    n = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').replace('at::native::', '')[:50]
    if dur < min_us: continue
    print(f"{(s - t0) / 1e3:9.1f} {dur:8.1f} q{r['Queue_Id']} {n}")
print('step len', (int(rows[b]['End_Timestamp']) - t0) / 1e3)
