"""Print the top kernels of a rocprofv3 --kernel-trace --stats --output-format csv directory (calls, average us, total ms)."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[: int(sys.argv[2]) if len(sys.argv) > 2 else 12]:
    print(f'{r["Name"][:64]:64s} {int(r["Calls"]):6d} {float(r["AverageNs"]) / 1e3:9.1f} us {float(r["TotalDurationNs"]) / 1e6:9.2f} ms')
