"""Offline model of k_sphere_trace's schedule from a recorded request history (tools/trace_rounds.py): how many dependent evaluation
rounds does the slowest workgroup need under a given speculation policy, and what does the kernel cost with measured per-round costs
(1 tile 52 us, 2 tiles 89 us, 3 tiles ~125 us, 4 tiles 150 us at one workgroup per CU)?"""
import sys
import numpy as np

COST = {0: 0.0, 1: 52.0, 2: 89.0, 3: 125.0, 4: 150.0}


def iterations(req, phase, k):
    """per ray: list of iterations, each = (sides stepping [2 bool], levels needed per side [2 int])"""
    T, R, _ = req.shape
    out = []
    for r in range(R):
        its = []
        for t in range(T):
            if not req[t, r].any():
                continue
            if phase[t, r] in (0, 1):
                its.append([req[t, r].astype(bool).copy(), np.zeros(2, int), phase[t, r] == 0])
            else:
                lv = k[t, r] + 1
                for s in range(2):
                    if req[t, r, s]:
                        its[-1][1][s] = max(its[-1][1][s], lv)
        out.append(its)
    return out


def simulate(its_all, NR, policy, max_tiles=1, depth=3):
    R = len(its_all)
    nwg = (R + NR - 1) // NR
    times, rounds_all, rows_all = [], [], 0
    for w in range(nwg):
        rays = its_all[w * NR:(w + 1) * NR]
        pos = [0] * len(rays)                       # current iteration index per ray
        done_lv = [np.zeros(2, int) for _ in rays]  # levels already evaluated in the current iteration (per side): -1 = step not yet evaluated
        for i in range(len(rays)):
            done_lv[i][:] = -1
        hard = [np.zeros(2, bool) for _ in rays]
        t_wg, nrounds = 0.0, 0
        while True:
            # mandatory requests
            reqs = []   # (ray, side, level)
            for i, its in enumerate(rays):
                if pos[i] >= len(its):
                    continue
                S1, L, init = its[pos[i]]
                for s in range(2):
                    if not S1[s]:
                        continue
                    if done_lv[i][s] < 0:
                        reqs.append((i, s, 0))
                    elif done_lv[i][s] < L[s]:
                        reqs.append((i, s, done_lv[i][s] + 1))
            # lockstep: a ray in line search requests only for sides still negative; a ray whose other side is done waits -- handled by levels
            if not reqs:
                break
            n = len(reqs)
            tiles = (n + 15) // 16
            spec = []
            if policy == 'none':
                pass
            elif policy == 'current':
                cand = [(i, s, 1) for (i, s, lv) in reqs if lv == 0 and not rays[i][pos[i]][2]]
                if cand and n + len(cand) <= tiles * 16:
                    spec = cand
            else:
                # prioritized: hard sides first, deeper levels after all first levels
                cand = []
                for d in range(1, depth + 1):
                    for (i, s, lv) in reqs:
                        if rays[i][pos[i]][2]:
                            continue
                        tgt = lv + d
                        if tgt <= 3:
                            pri = (0 if hard[i][s] else 1, d)
                            cand.append((pri, i, s, tgt))
                cand.sort(key=lambda c: c[0])
                free = tiles * 16 - n
                if policy == 'prio_extra' and tiles < max_tiles:
                    nhard = sum(1 for c in cand if c[0][0] == 0 and c[0][1] == 1)
                    if nhard > free:
                        tiles += 1
                        free += 16
                spec = [(i, s, tgt) for (_, i, s, tgt) in cand[:free]]
            rows_all += n
            t_wg += COST[(n + len(spec) + 15) // 16]
            nrounds += 1
            # apply
            have = {}
            for (i, s, lv) in reqs:
                have.setdefault((i, s), set()).add(lv)
            for (i, s, lv) in spec:
                have.setdefault((i, s), set()).add(lv)
            for (i, s), lvs in have.items():
                L = rays[i][pos[i]][1][s]
                cur = done_lv[i][s]
                while cur + 1 in lvs:
                    cur += 1
                    if cur >= L:
                        break
                done_lv[i][s] = cur
            for i, its in enumerate(rays):
                if pos[i] >= len(its):
                    continue
                S1, L, init = its[pos[i]]
                if all((not S1[s]) or done_lv[i][s] >= L[s] for s in range(2)):
                    for s in range(2):
                        hard[i][s] = S1[s] and L[s] > 0
                    pos[i] += 1
                    done_lv[i][:] = -1
        times.append(t_wg); rounds_all.append(nrounds)
    return np.array(times), np.array(rounds_all), rows_all


if __name__ == '__main__':
    d = np.load(sys.argv[1])
    its = iterations(d['req'], d['phase'], d['k'])
    NR = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    for pol, kw in (('none', {}), ('current', {}), ('prio', {}), ('prio', {'depth': 1}), ('prio_extra', {'max_tiles': 2})):
        t, r, rows = simulate(its, NR, pol, **kw)
        print('%-11s %-16s rounds max %2d mean %.1f | time max %.0f us mean %.0f | rows %d' % (pol, kw, r.max(), r.mean(), t.max(), t.mean(), rows))
