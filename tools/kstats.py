import csv,sys,glob,collections
d=sys.argv[1]
f=glob.glob(d+'/**/*kernel_stats.csv',recursive=True)
rows=list(csv.DictReader(open(f[0])))
for r in sorted(rows,key=lambda r:-float(r['TotalDurationNs']))[:int(sys.argv[2]) if len(sys.argv)>2 else 16]:
    print('%-95s calls %5s total %9.3f ms avg %9.3f us' % (r['Name'][:95], r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e3))
