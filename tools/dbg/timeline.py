import csv, numpy as np, glob, sys
rows=list(csv.DictReader(open(glob.glob(sys.argv[1])[0])))
K=[(r['Kernel_Name'].split('(')[0][-30:], int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Queue_Id'], int(r['Grid_Size_X'])//256) for r in rows]
K.sort(key=lambda k:k[1])
sp=[i for i,k in enumerate(K) if 'k_shoot_photons' in k[0]]
first=sp[-6]; t0=K[first][1]
end=max(k[2] for k in K[first:])
print('step span ms', (end-t0)/1e6)
for i in range(first,len(K)):
    n,s,e,q,g=K[i]
    if 'k_shoot_photons' in n or 'k_shoot_accumulate' in n or 'refresh_bounds' in n or 'init_bound' in n:
        print('%-30s q%s start %8.3f dur %8.3f wgs %d'%(n,q,(s-t0)/1e6,(e-s)/1e6,g))
acc=[i for i in range(first,len(K)) if 'accumulate_segments' in K[i][0]]
for r in list(range(0,10,2))+list(range(10,len(acc),12)):
    i=acc[r]; nxt=acc[r+1] if r+1<len(acc) else len(K)
    parts=[(K[j][0][-14:], (K[j][2]-K[j][1])/1e3) for j in range(i,nxt) if K[j][3]==K[i][3]]
    per=(K[nxt][1]-K[i][1])/1e3 if nxt<len(K) else 0
    print(r, 'start %.2f ms period %.1f us'%((K[i][1]-t0)/1e6, per), ' | '.join('%s %.1f'%p for p in parts))
