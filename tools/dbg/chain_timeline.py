"""Round durations of the longest brighter-fatter chain over one C3 step, from a `rocprofv3 --kernel-trace --output-format csv`
run of bench.py: for every tenth round of the top chain, when it starts and how long its three kernels take, next to what
the bulk stream is running at that time.  Usage: python tools/dbg/chain_timeline.py '<dir>/**/*kernel_trace.csv'"""
import csv, glob, sys, collections
rows = list(csv.DictReader(open(glob.glob(sys.argv[1], recursive=True)[0])))
K = [(r["Kernel_Name"].split("(")[0].replace("void ", "")[:26], int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"],
      int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1)) for r in rows]
K.sort(key=lambda k: k[1])
fused = [k for k in K if k[0].startswith("k_shoot_accumulate")]
t_end = fused[-1][2]
t_beg = fused[-2][2]                     # the last step: everything that starts after the previous fused launch ended
step = [k for k in K if k[1] > t_beg and k[1] <= t_end and "Fill" not in k[0] and "elementwise" not in k[0]]
t0 = min(k[1] for k in step)
byq = collections.defaultdict(list)
for k in step:
    byq[k[3]].append(k)
# the top chain: the queue with the most narrow accumulate launches
q = max(byq, key=lambda q: sum(1 for k in byq[q] if k[0].startswith("k_accumulate_round") and k[4] <= 2048))
acc = [k for k in byq[q] if k[0].startswith("k_accumulate_round")]
upd = [k for k in byq[q] if k[0].startswith("k_update")]
ref = [k for k in byq[q] if k[0].startswith("k_refresh")]
big = sorted((k for k in step if k[2] - k[1] > 300e3), key=lambda k: k[1])
print(f"step {(max(k[2] for k in step) - t0) / 1e6:.2f} ms; top chain on queue {q}: {len(acc)} rounds, ends at {(acc[-1][2] - t0) / 1e6:.2f} ms")
for n in range(0, len(acc), 10):
    a = acc[n]
    u = upd[n] if n < len(upd) else None
    r = ref[n] if n < len(ref) else None
    nxt = acc[n + 1][1] if n + 1 < len(acc) else a[2]
    running = [b[0] for b in big if b[1] <= a[1] <= b[2]]
    print(f"  round {n:3d} at {(a[1] - t0) / 1e6:6.2f} ms: acc {(a[2] - a[1]) / 1e3:6.1f} us ({a[4]} wgs)"
          + (f"  upd {(u[2] - u[1]) / 1e3:6.1f}" if u else "") + (f"  ref {(r[2] - r[1]) / 1e3:6.1f}" if r else "")
          + f"  round total {(nxt - a[1]) / 1e3:6.1f} us   beside: {', '.join(sorted(set(running)))}")
