"""Timeline of the last bench step from a rocprofv3 --kernel-trace csv: every launch longer than 0.3 ms and all
persistent chain launches (start offset, duration, workgroups, queue).
   python tools/dbg/chain_timeline.py '<dir>/**/*kernel_trace.csv'"""
import csv, glob, sys, collections
rows = list(csv.DictReader(open(glob.glob(sys.argv[1], recursive=True)[0])))
K = [(r['Kernel_Name'].split('(')[0][-30:], int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Queue_Id'],
      int(r['Grid_Size_X']) // 256) for r in rows]
K.sort(key=lambda k: k[1])
ini = [k for k in K if 'init_bound' in k[0]]
t0 = ini[-1][1]
step = [k for k in K if k[1] >= t0 and 'Fill' not in k[0] and 'copyBuffer' not in k[0]]
print('step span ms %.2f' % ((max(k[2] for k in step) - t0) / 1e6))
byq = collections.defaultdict(list)
for k in step:
    byq[k[3]].append(k)
for q, ks in byq.items():
    print('queue', q, 'n', len(ks), 'busy ms %.2f' % (sum(k[2] - k[1] for k in ks) / 1e6))
    for k in ks:
        if (k[2] - k[1]) > 3e5 or 'chain' in k[0]:
            print('    %-30s start %7.2f dur %7.3f wgs %d' % (k[0], (k[1] - t0) / 1e6, (k[2] - k[1]) / 1e6, k[4]))
