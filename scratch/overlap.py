import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ev = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:40]) for r in rows if 'decode_' in r['Kernel_Name'] or 'greedy' in r['Kernel_Name']]
ev.sort()
# take the last 2000 decode kernels
ev = ev[-2000:]
tot = sum(e - s for s, e, _ in ev)
busy = 0; cur_s, cur_e = ev[0][0], ev[0][1]
for s, e, _ in ev[1:]:
    if s <= cur_e: cur_e = max(cur_e, e)
    else: busy += cur_e - cur_s; cur_s, cur_e = s, e
busy += cur_e - cur_s
span = ev[-1][1] - ev[0][0]
print(f"kernels={len(ev)} sum_dur={tot/1e3:.0f}us union_busy={busy/1e3:.0f}us span={span/1e3:.0f}us  overlap_factor={tot/busy:.2f} idle_frac={(span-busy)/span:.2f}")
for s, e, n in ev[1000:1016]:
    print(f"  {s - ev[1000][0]:8d} +{e-s:6d} {n}")
