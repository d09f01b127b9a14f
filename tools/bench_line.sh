#!/bin/bash
# tools/bench_line.sh [bench.py args]: value, ms per sweep, kernel ms
python bench.py --cpu-rows 0 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.4g G/s  %.4f ms/sweep  kernel %.4f ms  %s' % (d['value']/1e9, d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['kernel']))"
