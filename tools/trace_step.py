"""Per-step kernel breakdown from a rocprofv3 --kernel-trace CSV (one hot-path step = the kernels
between two consecutive launches of the marker kernel: the first kernel of the loss section, InfoNCE prep_kernel).
python tools/trace_step.py <kernel_trace.csv> [step_index] [--timeline] [--marker=NAME]"""
import collections
import csv
import re
import sys


def main(path, which=12):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"]),
                         r.get("Queue_Id", r.get("Stream_Id", "?"))))
    rows.sort(key=lambda r: r[1])
    marker = ([a.split("=", 1)[1] for a in sys.argv if a.startswith("--marker=")] or ["::prep_kernel("])[0]
    idx = [i for i, r in enumerate(rows) if marker in r[0]]
    which = min(which, len(idx) - 2)
    step = rows[idx[which]:idx[which + 1]]
    tot = sum(e - s for _, s, e, _ in step) / 1e3
    wall = (step[-1][2] - step[0][1]) / 1e3
    print("step %d: %d kernels, sum of kernel time %.1f us, first-start..last-end %.1f us" % (which, len(step), tot, wall))
    agg = collections.OrderedDict()
    for n, s, e, _ in step:
        k = n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:70]
        if "at::native" in n:
            k = "torch:" + n.split("at::native::")[1].split("<")[0][:40] + ("/" + n.split("at::native::")[2].split("<")[0][:30] if n.count("at::native::") > 1 else "")
        a = agg.setdefault(k, [0, 0.0])
        a[0] += 1
        a[1] += (e - s) / 1e3
    print("%9s %5s  %s" % ("us", "calls", "kernel"))
    for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("%9.1f %5d  %s" % (t, c, k))


    if "--timeline" in sys.argv:
        t0 = step[0][1]
        qs = sorted({q for *_, q in step})
        print("\ntimeline (us from step start; one column per queue)")
        for n, s, e, q in step:
            k = re.sub(r"<.*", "", n.replace("void ", "").replace("(anonymous namespace)::", ""))
            k = (k.split("(")[0] or n)[-30:] + ("<%s>" % n.split("<")[1].split(">")[0][:6] if "mmssl" in n and "<" in n else "")
            print("%8.1f %7.1f  %s%s" % ((s - t0) / 1e3, (e - s) / 1e3, " " * (36 * qs.index(q)), k))


if __name__ == "__main__":
    main(sys.argv[1], int([a for a in sys.argv[2:] if a.isdigit()][0]) if [a for a in sys.argv[2:] if a.isdigit()] else 12)
