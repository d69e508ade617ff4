"""Wall-time attribution of ONE replayed training step from a rocprofv3 kernel trace (kernel_trace.csv).

Per-kernel duration sums (rocprofv3 --stats) double-count overlapped launches (the three SECOND3D branches, the weight-gradient side
stream, FPS on its own stream) and hide idle gaps.  This walks the step's timeline instead: at every instant the wall time is shared
equally among the kernels running then, gaps are booked as 'idle'.  Steps are delimited by `k_adamw_flat` (last kernel of a step).

  python tools/timeline.py <kernel_trace.csv> [out.txt]
"""
import collections
import csv
import sys


def short(name):
    n = name.split("(")[0]
    if n.startswith("void "):
        n = n[5:]
    return n[:90]


def main():
    path = sys.argv[1]
    out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
    rows = []
    full = {}
    for r in csv.DictReader(open(path)):
        full[(int(r["Start_Timestamp"]), int(r["End_Timestamp"]))] = r["Kernel_Name"]
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Grid_Size_X", r.get("Grid_Size", "?")),
                     r.get("Stream_Id", r.get("Queue_Id", "?"))))
    rows.sort()
    ends = [i for i, r in enumerate(rows) if r[2].startswith("k_adamw_flat")]
    if len(ends) < 3:
        print("fewer than 3 steps in the trace", file=out)
        return
    lo, hi = ends[-3] + 1, ends[-1] + 1                      # the last two full steps
    nsteps = 2
    seg = rows[lo:hi]
    t0, t1 = seg[0][0], max(r[1] for r in seg)
    ev = []
    for s, e, n, g, q in seg:
        ev.append((s, 1, n)); ev.append((e, -1, n))
    ev.sort(key=lambda x: (x[0], x[1]))
    running = collections.Counter()
    attributed = collections.Counter()
    conc = collections.Counter()
    idle, last = 0, t0
    for t, d, n in ev:
        dt = t - last
        if dt > 0:
            k = sum(running.values())
            if k == 0:
                idle += dt
            else:
                for name, c in running.items():
                    if c:
                        attributed[name] += dt * c / k
            conc[min(k, 4)] += dt
        last = t
        running[n] += d
    wall = (t1 - t0) / nsteps
    cnt = collections.Counter(n for _, _, n, _, _ in seg)
    raw = collections.Counter()
    for s, e, n, g, q in seg:
        raw[n] += e - s
    print(f"# step wall {wall / 1e6:.3f} ms ({nsteps} steps averaged), {len(seg) // nsteps} launches/step, idle (no kernel running) {idle / nsteps / 1e6:.3f} ms", file=out)
    print("# time with k kernels running: " + ", ".join(f"k={k}{'+' if k == 4 else ''}: {v / nsteps / 1e6:.3f} ms" for k, v in sorted(conc.items())), file=out)
    print(f"# {'attributed_ms':>13s} {'raw_sum_ms':>10s} {'launches':>8s}  kernel", file=out)
    for n, v in attributed.most_common(60):
        print(f"  {v / nsteps / 1e6:13.4f} {raw[n] / nsteps / 1e6:10.4f} {cnt[n] / nsteps:8.1f}  {n}", file=out)
    # launches that are not this library's kernels (ATen element-wise / copies / fills, rocBLAS): one step's list in launch order
    one = rows[ends[-2] + 1:ends[-1] + 1]
    other = [(i, r) for i, r in enumerate(one) if not r[2].startswith("k_")]
    print(f"# {len(other)} launches per step that are not k_* kernels, {sum(r[1] - r[0] for _, r in other) / 1e6:.3f} ms:", file=out)
    for i, (s, e, n, g, q) in other:
        prev_k = next((one[j][2] for j in range(i - 1, -1, -1) if one[j][2].startswith("k_")), "-")
        nm = full.get((s, e), n).replace("at::native::", "").replace("(anonymous namespace)::", "").replace("void ", "")
        print(f"    {(e - s) / 1e3:7.1f} us  grid {g:>9s}  after {prev_k[:28]:28s} {nm[:150]}", file=out)
    streams = collections.Counter()
    for s, e, n, g, q in seg:
        streams[q] += e - s
    print("# busy time per stream/queue id: " + ", ".join(f"{q}: {v / nsteps / 1e6:.2f} ms" for q, v in streams.most_common()), file=out)
    # gaps >= 3 us on the timeline: what ran before / after
    gaps = []
    cur_end, prev = seg[0][1], seg[0][2]
    for s, e, n, g, q in seg[1:]:
        if s - cur_end >= 3000:
            gaps.append((s - cur_end, prev, n))
        if e > cur_end:
            cur_end, prev = e, n
    gaps.sort(reverse=True)
    print(f"# {len(gaps)} gaps >= 3 us in {nsteps} steps, total {sum(g for g, _, _ in gaps) / nsteps / 1e6:.3f} ms/step; largest:", file=out)
    for g, a, b in gaps[:15]:
        print(f"    {g / 1e3:8.1f} us  after {a}  before {b}", file=out)
    # what the other queues do while a long side-stream kernel (k_fps) runs
    for s0, e0, n0, g0, q0 in seg:
        if not n0.startswith("k_fps"):
            continue
        inside = [(s, e, n, q) for s, e, n, g, q in seg if q != q0 and s < e0 and e > s0]
        before = [(s, e, n, q) for s, e, n, g, q in seg if q != q0 and e <= s0][-3:]
        after = [(s, e, n, q) for s, e, n, g, q in seg if q != q0 and s >= e0][:3]
        print(f"# {n0} on queue {q0}: {(e0 - s0) / 1e3:.1f} us; {len(inside)} kernels of other queues overlap it, covering "
              f"{sum(min(e, e0) - max(s, s0) for s, e, _, _ in inside) / 1e3:.1f} us", file=out)
        for s, e, n, q in before:
            print(f"      before: q{q} {n} ended {(s0 - e) / 1e3:.1f} us before its start", file=out)
        for s, e, n, q in inside[:6]:
            print(f"      inside: q{q} {n} start +{(s - s0) / 1e3:.1f} us, {(e - s) / 1e3:.1f} us long", file=out)
        for s, e, n, q in after:
            print(f"      after : q{q} {n} started {(s - e0) / 1e3:.1f} us after its end", file=out)


if __name__ == "__main__":
    main()
