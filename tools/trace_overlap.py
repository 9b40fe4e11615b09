import csv, glob, sys, collections
fs = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)
rows = []
for f in fs:
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        if 'ovtk' not in n: continue
        short = n.replace('void ','').split('(')[0].replace('ovtk::','').split('<')[0]
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), short, r.get('Stream_Id', r.get('Queue_Id','?'))))
rows.sort()
# steady state: last 40% of kernels
rows = rows[int(len(rows)*0.5):]
t0 = rows[0][0]
for s,e,n,q in rows[:40]:
    print(f"{(s-t0)/1000:9.1f} {(e-t0)/1000:9.1f} {(e-s)/1000:7.1f} us  q{q} {n}")
# busy fractions
T0, T1 = rows[0][0], rows[-1][1]
ev = []
for s,e,n,q in rows:
    ev.append((s, 1, n)); ev.append((e, -1, n))
ev.sort()
cur = collections.Counter(); last = T0; acc = collections.Counter(); conc = collections.Counter()
for t, d, n in ev:
    dt = t - last
    k = sum(cur.values())
    conc[k] += dt
    for nn, c in cur.items():
        if c > 0: acc[nn] += dt
    cur[n] += d; last = t
tot = T1 - T0
print("span", tot/1000, "us for", len(rows), "kernels")
for n, v in acc.items(): print(f"  {n:24s} running {v/tot:.2%} of the time")
for k in sorted(conc): print(f"  {k} kernels concurrently: {conc[k]/tot:.2%}")
per = collections.defaultdict(list)
for s,e,n,q in rows: per[n].append((e-s)/1000)
for n,v in per.items(): print(f"  {n:24s} mean {sum(v)/len(v):.1f} us over {len(v)}")
