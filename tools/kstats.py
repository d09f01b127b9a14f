"""Top kernels of a rocprofv3 --kernel-trace --stats run: python tools/kstats.py <dir> [n]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:int(sys.argv[2]) if len(sys.argv) > 2 else 8]:
    print('%-44s calls %-5s avg %9.1f ns  %5s%%' % (r['Name'][:44], r['Calls'], float(r['AverageNs']), r['Percentage']))
