#!/bin/bash
# kernel-trace timeline of one C3 step: when does each stream finish?
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/ctl -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-cold > $R/gpurun_out/ctl.log 2>&1; tail -5 $R/gpurun_out/ctl.log; find $R/gpurun_out/ctl -type f | head
F=$(find $R/gpurun_out/ctl -name "*kernel_trace.csv" | head -1)
python3 - "$F" <<'PY'
import csv, sys, collections
rows=list(csv.DictReader(open(sys.argv[1])))
K=[(r['Kernel_Name'].split('(')[0][-90:], int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Queue_Id'], int(r['Grid_Size_X'])//256) for r in rows]
K.sort(key=lambda k:k[1])
fused=[i for i,k in enumerate(K) if 'k_shoot_accumulate' in k[0]]
prev_end=K[fused[-2]][2]
step=[k for k in K if k[1]>prev_end]
ini=[k for k in step if 'init_tiles' in k[0]]
t0=min(k[1] for k in ini) if ini else step[0][1]
step=[k for k in K if k[1]>=t0 and 'Fill' not in k[0]]
print('step span ms %.2f'%((max(k[2] for k in step)-t0)/1e6))
byq=collections.defaultdict(list)
for k in step: byq[k[3]].append(k)
for q,ks in sorted(byq.items()):
    names=collections.Counter(k[0].split('<')[0][-26:] for k in ks)
    print('queue',q,'n',len(ks),'first %.2f last end %.2f busy %.2f ms'%((ks[0][1]-t0)/1e6,(ks[-1][2]-t0)/1e6,sum(k[2]-k[1] for k in ks)/1e6), dict(names.most_common(4)))
    for k in ks:
        if k[2]-k[1]>0.8e6: print('    %-36s start %6.2f dur %6.2f wgs %d'%(k[0][-36:],(k[1]-t0)/1e6,(k[2]-k[1])/1e6,k[4]))
    acc=[k for k in ks if 'accumulate_round' in k[0]]
    for a in range(0,len(acc),max(len(acc)//10,1)):
        print('      round-kernel %3d start %6.2f dur %.1f us wgs %d'%(a,(acc[a][1]-t0)/1e6,(acc[a][2]-acc[a][1])/1e3,acc[a][4]))
PY
python3 - "$F" <<'PY'
import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
K=[(r['Kernel_Name'].split('(')[0][-90:], int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Queue_Id'], int(r['Grid_Size_X'])//256) for r in rows]
K.sort(key=lambda k:k[1])
fused=[i for i,k in enumerate(K) if 'k_shoot_accumulate' in k[0]]
prev_end=K[fused[-2]][2]
q2=[k for k in K if k[1]>prev_end and k[3]=='2']
import numpy as np
for a in (30, 150, 400, 500):
    print('--- kernels', a, 'of the top chain')
    for i in range(a, min(a+7, len(q2))):
        k=q2[i]; gap=(k[1]-q2[i-1][2])/1e3 if i else 0
        print('   %-30s gap %7.1f us  dur %6.1f us  wgs %d'%(k[0].split('<')[0][-28:], gap, (k[2]-k[1])/1e3, k[4]))
d=np.array([k[2]-k[1] for k in q2])/1e3; g=np.array([q2[i][1]-q2[i-1][2] for i in range(1,len(q2))])/1e3
print('top chain: kernels %d, sum of durations %.2f ms, sum of gaps %.2f ms'%(len(q2), d.sum()/1e3, g.sum()/1e3))
PY
rm -rf $R/gpurun_out/ctl
