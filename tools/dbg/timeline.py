import csv, numpy as np, glob, sys, collections
rows=list(csv.DictReader(open(glob.glob(sys.argv[1])[0])))
K=[(r['Kernel_Name'].split('(')[0][-28:], int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Queue_Id'], int(r['Grid_Size_X'])//256) for r in rows]
K.sort(key=lambda k:k[1])
bulk=[i for i,k in enumerate(K) if 'k_shoot_accumulate' in k[0]]
# last step = kernels after end of previous bulk
prev_end=K[bulk[-2]][2]
step=[k for k in K if k[1]>prev_end and 'Fill' not in k[0] and 'copyBuffer' not in k[0]]
# the previous step's chain may extend beyond prev bulk end; use first init_boundaries after prev_end as step start
ini=[k for k in step if 'init_bound' in k[0]]
t0=ini[-1][1] if ini else step[0][1]
step=[k for k in K if k[1]>=t0]
print('step span ms', (max(k[2] for k in step)-t0)/1e6)
byq=collections.defaultdict(list)
for k in step: byq[k[3]].append(k)
for q,ks in byq.items():
    big=[k for k in ks if (k[2]-k[1])>1e6]
    print('queue',q,'n',len(ks),'first %.2f last end %.2f'%((ks[0][1]-t0)/1e6,(ks[-1][2]-t0)/1e6), 'busy ms %.2f'%(sum(k[2]-k[1] for k in ks)/1e6))
    for k in big: print('    %-28s start %7.2f dur %6.2f wgs %d'%(k[0],(k[1]-t0)/1e6,(k[2]-k[1])/1e6,k[4]))
    acc=[k for k in ks if 'accumulate_segments' in k[0]]
    if acc:
        st=np.array([k[1] for k in acc]); 
        for a in range(0,len(acc),max(len(acc)//12,1)):
            print('      round %3d start %7.2f acc %.1f us wgs %d'%(a,(acc[a][1]-t0)/1e6,(acc[a][2]-acc[a][1])/1e3,acc[a][4]))
