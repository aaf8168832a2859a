"""Dev tool: every launch of ONE step of a rocprofv3 kernel trace, in order (duration, grid, workgroup, kernel).   python tools/trace_list.py <trace dir>"""
import csv,sys,glob
f=glob.glob(sys.argv[1]+'/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
names=[r['Kernel_Name'] for r in rows]
# last step = last ~155 launches: find the last pack_input
idx=[i for i,n in enumerate(names) if 'pack_input' in n]
a=idx[-2]; b=idx[-1]
for r in rows[a:b]:
    d=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
    n=r['Kernel_Name'].replace('wx::','').replace('unsigned short','u16')[:70]
    print(f"{d:8.1f} us  grid {r['Grid_Size_X']:>8} wg {r['Workgroup_Size_X']:>4}  {n}")
