#!/usr/bin/env python3
"""Summarise a tools_prof.sh output directory: per-launch kernel time and PMC counters."""
import collections, csv, glob, sys
d = sys.argv[1]
for f in glob.glob(d + '/trace/*/*kernel_trace.csv'):
    rows = list(csv.DictReader(open(f)))
    t = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rows if 'render' in r['Kernel_Name']]
    print('kernel us:', [round(x) for x in t][-10:], 'VGPR', rows[-1].get('VGPR_Count'), 'SGPR', rows[-1].get('SGPR_Count'), 'grid', rows[-1].get('Grid_Size'))
for f in sorted(glob.glob(d + '/pmc_*/*/*counter_collection.csv')):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'render' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in agg.items():
        print(f'{k:28s} {v[-1]:16.0f}')
