import csv, glob, sys, collections
d = sys.argv[1]
files = glob.glob(d + '/**/*counter_collection.csv', recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in files:
    for row in csv.DictReader(open(f)):
        name = row['Kernel_Name'].split('(')[0][:60]
        agg[name][row['Counter_Name']].append(float(row['Counter_Value']))
for name, ctrs in sorted(agg.items()):
    if not any(k in name for k in ('k_force', 'k_step', 'k_build', 'k_kickdrift', 'k_finalize')):
        continue
    print(name)
    for c, v in sorted(ctrs.items()):
        print('   %-36s n=%4d  mean=%.4g' % (c, len(v), sum(v) / len(v)))
