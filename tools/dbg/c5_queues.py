"""Which hardware queues the kernels of concurrent CCDs run on, and how much they overlap: reads the kernel-trace CSV of
   rocprofv3 --kernel-trace --output-format csv -- python3 tools/dbg/c5_profile.py N."""
import collections
import csv
import glob
import sys

rows = list(csv.DictReader(open(glob.glob(sys.argv[1])[0])))
K = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r.get("Stream_Id", "?"), r["Kernel_Name"].split("(")[0][-24:]) for r in rows)
t0, t1 = K[len(K) // 2][0], K[-1][1]                       # second half of the run (steady state)
K = [k for k in K if k[0] >= t0]
byq = collections.defaultdict(list)
for k in K:
    byq[k[2]].append(k)
print("span ms %.1f kernels %d" % ((t1 - t0) / 1e6, len(K)))
for q, ks in sorted(byq.items()):
    names = collections.Counter(k[4] for k in ks)
    print("queue %s: n %6d busy %.1f ms streams %d  %s" % (q, len(ks), sum(k[1] - k[0] for k in ks) / 1e6, len(set(k[3] for k in ks)),
                                                          dict(names.most_common(3))))
    busy = collections.Counter()
    for k in ks:
        busy[k[4]] += k[1] - k[0]
    print("      busy by kernel [ms]:", {n: round(v / 1e6, 1) for n, v in busy.most_common(6)})
# concurrency histogram: how many kernels run at a time, weighted by time
ev = sorted([(k[0], 1) for k in K] + [(k[1], -1) for k in K])
lvl, last, hist = 0, ev[0][0], collections.Counter()
for t, d in ev:
    hist[lvl] += t - last
    lvl += d
    last = t
tot = sum(hist.values())
print("time fraction by number of kernels in flight:", {k: round(v / tot, 3) for k, v in sorted(hist.items())})
