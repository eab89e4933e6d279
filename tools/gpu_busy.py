"""GPU busy fraction of the last traced steps: union of kernel intervals / wall time (rocprofv3 kernel trace csv)."""
import csv, sys
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
# last 40 % of the trace = steady-state steps
t0 = rows[int(len(rows) * 0.6)][0]
rows = [r for r in rows if r[0] >= t0]
busy, cur_s, cur_e = 0, rows[0][0], rows[0][1]
gaps = []
for s, e, _ in rows[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; gaps.append(s - cur_e); cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
wall = rows[-1][1] - rows[0][0]
gaps.sort()
print('kernels %d  wall %.2f ms  busy %.2f ms (%.1f %%)  gaps: n=%d median %.1f us  p90 %.1f us  total %.2f ms' % (
    len(rows), wall / 1e6, busy / 1e6, 100.0 * busy / wall, len(gaps), gaps[len(gaps) // 2] / 1e3 if gaps else 0,
    gaps[int(len(gaps) * 0.9)] / 1e3 if gaps else 0, sum(gaps) / 1e6))
