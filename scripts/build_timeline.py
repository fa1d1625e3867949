#!/usr/bin/env python3
"""Where the layout-building call's wall time goes: reads a `rocprofv3 --kernel-trace --hip-trace --output-format csv` directory of a
bench.py run and prints, for the window between the first and the last layout-build kernel, the GPU-busy time, the idle gaps between
kernels (largest first, with the kernels either side) and the host API time inside the window per HIP function.
    python scripts/build_timeline.py <rocprof-output-dir>"""
import csv, glob, os, sys, collections

BUILD_KERNELS = ("k_hot_hist", "k_split_classify", "k_split_fill", "k_long_keys", "k_order_keys", "k_order_place", "k_order_coarse", "k_twin_",
                 "k_strip_", "k_tag_", "k_rtile_keys", "k_rtile_place", "k_rtile_heads", "k_rtile_table", "k_rtile_starts", "k_rtile_units",
                 "k_rtile_pad_tags", "k_vdict_collect", "k_ctile_keys", "k_ctile_place", "k_ctile_first", "k_hrec_init", "k_tile_table")


def col(row, *names):
    for n in names:
        if n in row:
            return row[n]
    raise KeyError(names)


def main(d):
    kfiles = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    afiles = glob.glob(os.path.join(d, "**", "*hip_api_trace.csv"), recursive=True)
    kern = []
    for f in kfiles:
        for r in csv.DictReader(open(f)):
            kern.append((int(col(r, "Start_Timestamp")), int(col(r, "End_Timestamp")), col(r, "Kernel_Name")))
    kern.sort()
    build = [k for k in kern if "grb::" in k[2] and any(b in k[2] for b in BUILD_KERNELS)]
    if not build:
        print("no layout-build kernels in the trace")
        return
    t0, t1 = build[0][0], max(k[1] for k in build)
    inside = [k for k in kern if k[0] >= t0 and k[1] <= t1]
    busy, cur_end, gaps = 0, t0, []
    prev = "(window start)"
    for s, e, n in inside:
        if s > cur_end:
            gaps.append((s - cur_end, prev, n))
        if e > cur_end:
            busy += e - max(s, cur_end)
            cur_end = e
            prev = n
    short = lambda n: n.replace("void ", "").split("(")[0][:60]
    print(f"window {(t1 - t0) / 1e6:.2f} ms (first to last layout-build kernel), {len(inside)} kernels, GPU busy {busy / 1e6:.2f} ms, idle {(t1 - t0 - busy) / 1e6:.2f} ms")
    by = collections.defaultdict(lambda: [0, 0])
    for s, e, n in inside:
        by[short(n)][0] += e - s
        by[short(n)][1] += 1
    print("kernels inside the window (ms, calls):")
    for n, (t, c) in sorted(by.items(), key=lambda x: -x[1][0])[:24]:
        print(f"   {t / 1e6:7.2f} {c:4d}  {n}")
    print("largest idle gaps (us: after -> before):")
    for g, a, b in sorted(gaps, reverse=True)[:16]:
        print(f"   {g / 1e3:8.1f}  {short(a)} -> {short(b)}")
    api = collections.defaultdict(lambda: [0, 0])
    calls = []
    for f in afiles:
        for r in csv.DictReader(open(f)):
            s, e = int(col(r, "Start_Timestamp")), int(col(r, "End_Timestamp"))
            if s >= t0 and e <= t1:
                fn = col(r, "Function")
                api[fn][0] += e - s
                api[fn][1] += 1
                calls.append((s, e, fn))
    calls.sort()
    # what the host did during the three largest gaps: the API calls that overlap each gap (us from the gap's start; duration)
    ends = {}
    cur_end = t0
    for s, e, n in inside:
        if s > cur_end:
            ends[(s - cur_end, n)] = (cur_end, s)
        cur_end = max(cur_end, e)
    for (g, n), (gs, ge) in sorted(ends.items(), reverse=True)[:3]:
        print(f"host side of the {g / 1e3:.0f} us gap before {short(n)}:")
        before = [short(k[2]) for k in inside if k[1] <= gs][-5:]
        after = [short(k[2]) for k in inside if k[0] >= ge][:5]
        print("   kernels before:", " | ".join(before))
        print("   kernels after: ", " | ".join(after))
        last = gs
        for s, e, fn in calls:
            if e >= gs and s <= ge and (e - s > 20000 or s - last > 200000):
                print(f"   +{(s - gs) / 1e3:9.1f} us  {(e - s) / 1e3:9.1f} us  {fn}   (host time outside the API before it: {max(0, s - last) / 1e3:.0f} us)")
            if e >= gs and s <= ge:
                last = max(last, e)
    if os.environ.get("TIMELINE_DUMP"):  # raw events around the largest gap: (us from the gap's start) kernel executions K and host API calls A
        (g, n), (gs, ge) = sorted(ends.items(), reverse=True)[0]
        ev = [(s, e, "K " + short(nm)) for s, e, nm in inside if e >= gs - 8_000_000 and s <= ge + 300_000]
        ev += [(s, e, "A " + fn) for s, e, fn in calls if e >= gs - 8_000_000 and s <= ge + 300_000 and e - s > 4000]
        print("events around the largest gap (start us, end us, what):")
        for s, e, w in sorted(ev):
            print(f"   {(s - gs) / 1e3:10.1f} {(e - gs) / 1e3:10.1f}  {w}")
    print("host HIP API time inside the window (ms, calls):")
    for fn, (t, c) in sorted(api.items(), key=lambda x: -x[1][0])[:14]:
        print(f"   {t / 1e6:7.2f} {c:5d}  {fn}")


if __name__ == "__main__":
    main(sys.argv[1])
