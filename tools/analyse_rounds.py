"""Per-CU timeline of one stamped planned-forward launch (tools/stamp_rounds.hip): which workgroup ran where, when it started,
how long its fill and its tasks took, and what a CU's SECOND workgroup waited for.

    python tools/analyse_rounds.py gpurun_out/rounds/S300_A20.txt [...]

Times in microseconds from the launch's first wave start (s_memrealtime, 10 ns ticks); phases inside a workgroup in shader
cycles (s_memtime) converted with the launch's own cycles-per-tick ratio."""
import sys
from collections import defaultdict

import numpy as np


def load(path):
    head = open(path).readline().split()
    meta = {head[i]: head[i + 1] for i in range(1, len(head) - 1, 2)}
    d = np.loadtxt(path, dtype=np.int64, comments="#")
    return meta, d


def analyse(path):
    meta, d = load(path)
    wpw = int(meta["waves_per_wg"])
    nwg = d.shape[0] // wpw
    d = d[: nwg * wpw].reshape(nwg, wpw, -1)
    t0 = d[:, :, 1].min()
    start = (d[:, :, 1].min(axis=1) - t0) * 0.01            # us: first wave of the workgroup
    start_last = (d[:, :, 1].max(axis=1) - t0) * 0.01       # its last wave
    end = (d[:, :, 2].max(axis=1) - t0) * 0.01
    # shader cycles per 10 ns tick, from the longest-lived wave
    dur_rt = (d[:, :, 2] - d[:, :, 1]).astype(float)
    dur_cy = (d[:, :, 6] - d[:, :, 3]).astype(float)
    k = np.unravel_index(dur_rt.argmax(), dur_rt.shape)
    cyc_per_us = dur_cy[k] / (dur_rt[k] * 0.01)
    fill_issue = (d[:, :, 4] - d[:, :, 3]).max(axis=1) / cyc_per_us          # start -> fill loads issued
    barrier = (d[:, :, 5] - d[:, :, 3]).max(axis=1) / cyc_per_us             # start -> barrier passed (the unit is staged)
    tasks = (d[:, :, 6] - d[:, :, 5])                                        # per wave: barrier -> end
    tasks_max = tasks.max(axis=1) / cyc_per_us
    tasks_min = tasks.min(axis=1) / cyc_per_us
    hw = d[:, 0, 7]
    xcc = d[:, 0, 8] & 0xF
    cu = (hw >> 8) & 0xF
    sh = (hw >> 12) & 0x1
    se = (hw >> 13) & 0x7
    phys = xcc * 1000 + se * 100 + sh * 10 + cu
    per_cu = defaultdict(list)
    for w in range(nwg):
        per_cu[int(phys[w])].append(w)
    for v in per_cu.values():
        v.sort(key=lambda w: start[w])
    span = end.max()
    cut = f"{meta['units']} units x {meta['wgs_per_unit']} per unit"
    if int(meta.get("units1", meta["units"])) < int(meta["units"]):
        u2 = int(meta.get("units2", meta["units"]))
        cut = (f"the first {meta['units1']} units x {meta['wgs_per_unit']} per unit, the next {u2 - int(meta['units1'])} x "
               f"{meta['wgs_per_unit2']}, the last {int(meta['units']) - u2} x {meta.get('wgs_per_unit3', meta['wgs_per_unit2'])}")
    print(f"== {path}: S {meta['S']} A {meta['A']}: {nwg} workgroups of {wpw} waves ({cut}, "
          f"{meta['ns']} slices per unit), {len(per_cu)} CUs used, {cyc_per_us:.0f} shader cycles per us")
    print(f"   launch span (first wave start -> last wave end) {span:.2f} us; stamped build, back to back: {meta['us_per_launch_stamped_build']} us per launch")
    depth = np.array([len(v) for v in per_cu.values()])
    print(f"   workgroups per CU: " + ", ".join(f"{n} on {int((depth == n).sum())} CUs" for n in sorted(set(depth))))
    xc = np.array([int((xcc == x).sum()) for x in range(8)])
    print(f"   workgroups per XCD: {xc.tolist()}")
    # rounds: position of a workgroup on its CU
    pos = np.zeros(nwg, dtype=int)
    prev_end = np.full(nwg, np.nan)
    for v in per_cu.values():
        for i, w in enumerate(v):
            pos[w] = i
            if i:
                prev_end[w] = end[v[i - 1]]
    for r in range(pos.max() + 1):
        m = pos == r
        line = (f"   round {r}: {int(m.sum()):4d} workgroups | start p0/p50/p100 {np.percentile(start[m], 0):6.2f} {np.percentile(start[m], 50):6.2f} {np.percentile(start[m], 100):6.2f}"
                f" | last wave of the workgroup starts +{np.median(start_last[m] - start[m]):.2f}"
                f" | start -> staged p50/p100 {np.median(barrier[m]):5.2f} {barrier[m].max():5.2f} (its rows written at +{np.median(fill_issue[m]):.2f})"
                f" | tasks: longest wave p50/p100 {np.median(tasks_max[m]):5.2f} {tasks_max[m].max():5.2f}, shortest wave p50 {np.median(tasks_min[m]):5.2f}"
                f" | end p50/p100 {np.median(end[m]):6.2f} {end[m].max():6.2f}")
        if r:
            gap = start[m] - prev_end[m]
            line += f" | gap after the CU's previous workgroup p0/p50/p100 {np.percentile(gap, 0):.2f} {np.percentile(gap, 50):.2f} {np.percentile(gap, 100):.2f}"
        print(line)
    # per wave of a workgroup (median over the workgroups): start after the workgroup's first wave, then its own phases
    wstart = (d[:, :, 1] - d[:, :, 1].min(axis=1, keepdims=True)) * 0.01
    cols = [("starts", np.median(wstart, axis=0))]
    cols.append(("rows written", np.median((d[:, :, 4] - d[:, :, 3]) / cyc_per_us, axis=0)))
    cols.append(("barrier passed", np.median((d[:, :, 5] - d[:, :, 3]) / cyc_per_us, axis=0)))
    cols.append(("end", np.median((d[:, :, 6] - d[:, :, 3]) / cyc_per_us, axis=0)))
    for name, v in cols:
        print(f"   per wave, {name:15s} (us{'' if name == 'starts' else ' after its own start'}): " + " ".join(f"{x:5.2f}" for x in v))
    busy = sum(end[w] - start[w] for w in range(nwg))
    print(f"   CU occupancy: {busy / (256 * span) * 100:.1f} % of 256 CUs x span held by a workgroup; "
          f"staged-and-gathering (barrier -> longest wave's end): {sum(tasks_max) / (256 * span) * 100:.1f} %; "
          f"all 16 waves gathering (barrier -> shortest wave's end): {sum(tasks_min) / (256 * span) * 100:.1f} %")
    idle_tail = np.array([span - max(end[w] for w in v) for v in per_cu.values()])
    print(f"   CU idle at the end of the launch (its last workgroup's end -> launch end): mean {idle_tail.mean():.2f} us, max {idle_tail.max():.2f} us; "
          f"idle CUs (never used): {256 - len(per_cu)}")


if __name__ == "__main__":
    for p in sys.argv[1:]:
        analyse(p)
