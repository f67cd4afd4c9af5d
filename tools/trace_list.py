"""list the launches of a rocprofv3 kernel-trace csv in start order: kernel (demangled prefix), workgroups, microseconds"""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
filt = sys.argv[2] if len(sys.argv) > 2 else ''
for r in rows:
    n = r['Kernel_Name']
    if filt in n:
        print('%-90s wgs=%-6d %8.1f us' % (n[:90], int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']),
                                            (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3))
