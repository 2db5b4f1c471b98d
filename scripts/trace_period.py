"""From a rocprofv3 --kernel-trace directory: per DP kernel name, the launches' average DURATION (what --stats reports) and the
average PERIOD between consecutive ENDS of its launches (gaps longer than 3 x the median left out: warm-up / settle / probe phases
of bench.py are separated by host work). Queued searches overlap on two streams (round 6): the period, not the duration, is what a
search costs -- this is the profiler-side check of bench.py's roofline.kernel_ms / launch_ms."""
import csv, glob, sys, collections
d = sys.argv[1]
rows = collections.defaultdict(list)
for f in glob.glob(f"{d}/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        n = r.get("Kernel_Name", "")
        if "opal_dp" in n or "opal_sw_stream" in n or "opal_gl_stream" in n:
            rows[n].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
for n, v in rows.items():
    v.sort(key=lambda t: t[1])
    dur = [e - s for s, e in v]
    per = [v[i + 1][1] - v[i][1] for i in range(len(v) - 1)]
    if not per:
        continue
    med = sorted(per)[len(per) // 2]
    steady = [p for p in per if p <= 3 * med]
    overl = sum(1 for i in range(len(v) - 1) if v[i + 1][0] < v[i][1])
    # ... and over the QUEUED phases alone: pairs of launches of which the second starts before the first ends (bench.py's settle,
    # warm-up, timed and clock-probe searches; the drop-in calls of its e2e leg are one search per call and never overlap)
    qper = [v[i + 1][1] - v[i][1] for i in range(len(v) - 1) if v[i + 1][0] < v[i][1]]
    qdur = [v[i + 1][1] - v[i + 1][0] for i in range(len(v) - 1) if v[i + 1][0] < v[i][1]]
    if qper:
        print(f"{d.split('/')[-2]:10s} {n[:60]:60s} queued launches only ({len(qper)}): avg duration {sum(qdur) / len(qdur) / 1e3:9.1f} us  avg period {sum(qper) / len(qper) / 1e3:9.1f} us")
    print(f"{d.split('/')[-2]:10s} {n[:60]:60s} launches {len(v):4d}  avg duration {sum(dur) / len(dur) / 1e3:9.1f} us  avg period (steady) {sum(steady) / len(steady) / 1e3:9.1f} us  "
          f"launches that start before the one before ends: {overl}")
