#!/bin/bash
# kernel trace of C5 on 24 CCDs with joint top chains: per queue the busy time and idle gaps, the joint kernels' durations by round
R=$PWD
J=${1:-8}
N=${2:-24}
export R4N=$N
cd /tmp && export TMPDIR=/tmp
export IMS_FOCAL_JOINT=$J R4_SKIP_SINGLE=1 R4_CONC=4
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/jt -- python3 $R/tools/dbg/r4_c5.py $N > $R/gpurun_out/jt.log 2>&1; grep concurrent $R/gpurun_out/jt.log
F=$(find $R/gpurun_out/jt -name "*kernel_trace.csv" | head -1)
python3 - "$F" <<'PY'
import csv, sys, collections, os
import numpy as np
N=int(os.environ.get('R4N','24'))
rows=list(csv.DictReader(open(sys.argv[1])))
K=[(r['Kernel_Name'].split('(')[0][-90:], int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Queue_Id'], int(r['Grid_Size_X'])//256) for r in rows]
K.sort(key=lambda k:k[1])
# the last step: after the last long idle gap (> 50 ms: the warm-up's end and the bench's bookkeeping) -- take the final 24 fused launches
fused=[i for i,k in enumerate(K) if 'k_shoot_accumulate' in k[0]]
first=fused[-N]
# step starts at the first kernel after the end of the fused launch before those
t_prev=K[fused[-N-1]][2] if len(fused)>N else K[0][1]
step=[k for k in K if k[1]>t_prev+5e6]        # skip the copies right behind
t0=step[0][1]
print('step span ms %.2f, kernels %d'%((max(k[2] for k in step)-t0)/1e6, len(step)))
byq=collections.defaultdict(list)
for k in step: byq[k[3]].append(k)
for q,ks in sorted(byq.items()):
    names=collections.Counter(k[0].split('<')[0][-26:] for k in ks)
    busy=sum(k[2]-k[1] for k in ks)/1e6
    print('queue',q,'n',len(ks),'first %.2f last end %.2f busy %.2f ms'%((ks[0][1]-t0)/1e6,(ks[-1][2]-t0)/1e6,busy), dict(names.most_common(5)))
    gaps=[(ks[i][1]-ks[i-1][2])/1e6 for i in range(1,len(ks))]
    big=[(round((ks[i][1]-t0)/1e6,1), round(g,2)) for i,g in zip(range(1,len(ks)),gaps) if g>2.0]
    print('     idle gaps > 2 ms (at, length):', big[:30])
for name in ('accumulate_round_j','update_distortions_q3_j','refresh_changed_j','build_active_j','update_list_j','refresh_list_j','accumulate_round_c','k_update_distortions_q3<','k_refresh_changed<','k_shoot_photons','k_shoot_accumulate','k_init_tiles','copyBuffer','elementwise','k_fft'):
    ks=[k for k in step if name in k[0]]
    if not ks: continue
    d=np.array([k[2]-k[1] for k in ks])/1e3
    print('%-28s n %5d sum %8.2f ms avg %6.1f us  p50 %6.1f p90 %6.1f max %7.1f   wgs avg %d'%(name,len(ks),d.sum()/1e3,d.mean(),np.percentile(d,50),np.percentile(d,90),d.max(), np.mean([k[4] for k in ks])))
jq=[k for k in step if 'round_j' in k[0] and k[4] > 0]
import collections as _c
qc=_c.Counter(k[3] for k in jq)
jq=[k for k in jq if k[3]==qc.most_common()[-1][0]] if len(qc)>1 else jq
if jq:
    q=jq[0][3]
    ks=byq[q]
    g=np.array([ks[i][1]-ks[i-1][2] for i in range(1,len(ks))])/1e3
    print('joint queue', q, ': kernels', len(ks), 'sum dur %.1f ms'%(sum(k[2]-k[1] for k in ks)/1e6), 'gaps: sum %.1f ms, median %.1f us'%(g.sum()/1e3, np.median(g)))
    for a in range(0, len(ks), max(len(ks)//30,1)):
        k=ks[a]
        print('   %-26s at %7.2f ms dur %6.1f us wgs %6d  gap before %.1f us'%(k[0].split('<')[0][-26:], (k[1]-t0)/1e6, (k[2]-k[1])/1e3, k[4], (k[1]-ks[a-1][2])/1e3 if a else 0))
PY
rm -rf $R/gpurun_out/jt
