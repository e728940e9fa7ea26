import csv, sys, glob
f = glob.glob(sys.argv[1] + '/*/*kernel_stats.csv')[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    name = r['Name'].replace('eioku::(anonymous namespace)::','').replace('(anonymous namespace)::','').replace('void ','')
    name = name.split('(')[0][:60]
    print(f"{name:60s} calls {int(r['Calls']):5d} total_ms {float(r['TotalDurationNs'])/1e6:9.3f} avg_us {float(r['AverageNs'])/1e3:9.1f} per_step_ms {float(r['TotalDurationNs'])/1e6/steps:7.3f}")
