"""Per-step summary of a rocprofv3 kernel_stats.csv: python scripts/kstats.py <csv> <steps incl. warm-up> [top]"""
import csv
import re
import sys


def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    if 'rocprim' in n:
        m = re.findall(r'(radix_sort_onesweep_iteration|onesweep_global_offsets|scan_impl|radix_sort_block_sort|merge_sort\w*|transform_impl|lookback_scan_state)', n)
        b = re.search(r'>, (\d+)u, \(rocprim', n)
        return 'rocprim ' + (m[0] if m else '?') + (' %sbit' % b.group(1) if b else '') + (' u64' if 'unsigned long' in n.split('target_arch')[0] else ' u32')
    n = re.sub(r'^void ', '', n)
    return re.split(r'\(', n)[0][:60]


rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('total GPU kernel time per step: %.1f ms' % (tot / 1e6 / steps))
for r in rows[:top]:
    print('%-62s %5s calls %9.3f ms/step  avg %8.3f ms  %5.2f%%' % (short(r['Name']), r['Calls'], float(r['TotalDurationNs']) / 1e6 / steps, float(r['AverageNs']) / 1e6, float(r['Percentage'])))
