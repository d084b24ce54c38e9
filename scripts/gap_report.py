"""Where the device idles inside a step: reads the chrome trace `bench.py --torch-profile PATH` exports (PATH.trace.json),
orders the kernel / memcpy / memset events of the compute stream by start time and charges every idle gap between two of them
to the host operator that launched the kernel AFTER the gap (matched through the launch's correlation id; the operator is
the outermost cpu_op -- an autograd Function, an aten operator -- that contains the launch on its thread).

  python scripts/gap_report.py PATH.trace.json [steps]
"""
import bisect, collections, json, sys

tr = json.load(open(sys.argv[1]))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ev = tr["traceEvents"] if isinstance(tr, dict) else tr
dev, launches, ops = [], {}, collections.defaultdict(list)
for e in ev:
    if e.get("ph") != "X":
        continue
    cat = e.get("cat", "")
    a = e.get("args", {})
    if cat in ("kernel", "gpu_memcpy", "gpu_memset"):
        dev.append((e["ts"], e["ts"] + e["dur"], e["name"], a.get("correlation")))
    elif cat in ("cuda_runtime", "cuda_driver"):
        if a.get("correlation") is not None:
            launches[a["correlation"]] = (e["ts"], e["tid"], e["name"])
    elif cat in ("cpu_op", "user_annotation"):
        ops[e["tid"]].append((e["ts"], e["ts"] + e["dur"], e["name"]))
for t in ops:
    ops[t].sort()
starts = {t: [o[0] for o in v] for t, v in ops.items()}


def chain(tid, ts):
    """outermost ... innermost operators of thread tid that contain ts"""
    v, st = ops.get(tid, []), starts.get(tid, [])
    i = bisect.bisect_right(st, ts) - 1
    out = []
    lo = max(0, i - 400)
    for j in range(lo, i + 1):
        if v[j][0] <= ts <= v[j][1]:
            out.append(v[j][2])
    return out


dev.sort()
busy = sum(e[1] - e[0] for e in dev)
span = dev[-1][1] - dev[0][0]
by_op, by_pair, n_by_op = collections.Counter(), collections.Counter(), collections.Counter()
idle = 0.0
end = dev[0][1]
big = []
for k in range(1, len(dev)):
    s, e, name, corr = dev[k]
    gap = s - end
    if gap > 0:
        idle += gap
        l = launches.get(corr)
        ch = chain(l[1], l[0]) if l else []
        ch = [c for c in ch if not c.startswith("autograd::engine")]
        top = ch[0] if ch else "(no host operator: ctypes launch)"
        by_op[top] += gap
        n_by_op[top] += 1
        if gap > 100:
            big.append((gap, top, name[:60], dev[k - 1][2][:60]))
    end = max(end, e)
print(f"device events {len(dev)} ({len(dev) / steps:.0f} per step); span {span / steps / 1e3:.2f} ms per step; busy {busy / steps / 1e3:.2f} ms; "
      f"idle {idle / steps / 1e3:.2f} ms per step")
print("idle time by the host operator that launched the next kernel (ms per step, gaps per step, mean gap us):")
for op, g in by_op.most_common(30):
    print(f"  {g / steps / 1e3:7.3f}  {n_by_op[op] / steps:7.1f}  {g / n_by_op[op]:7.1f}  {op[:90]}")
print("gaps over 100 us (us, launching operator, kernel after, kernel before):")
for g in sorted(big, reverse=True)[:40]:
    print(f"  {g[0]:8.0f}  {g[1][:40]:40s} {g[2]:60s} <- {g[3]}")
