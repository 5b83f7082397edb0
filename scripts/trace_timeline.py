import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ks = [r for r in rows if 'cgv::' in r['Kernel_Name']]
# last search = kernels after the last prep_rows_kernel
last = max(i for i, r in enumerate(ks) if 'prep_rows' in r['Kernel_Name'])
sel = ks[last:]
t0 = int(sel[0]['Start_Timestamp'])
for r in sel:
    nm = r['Kernel_Name'].split('(')[0].replace('void ', '')[-52:]
    print(f"{nm:52s} start={(int(r['Start_Timestamp'])-t0)/1e3:9.1f}us dur={(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:8.1f}us "
          f"grid={r['Grid_Size_X']} wg={r['Workgroup_Size_X']} vgpr={r.get('VGPR_Count')}")
print(f"total device span {(int(sel[-1]['End_Timestamp'])-t0)/1e3:.1f} us")
