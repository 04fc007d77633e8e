"""Offline model (tools/trace_rounds.py history): what would a GLOBAL re-partition of the still-active rays over the workgroups buy
k_sphere_trace?  Rounds run in lock-step over all workgroups (every workgroup evaluates one tile per round while it has requests); at the
given rounds the active rays are dealt round-robin to the workgroups, so that every tile has free slots for the speculative line-search
levels.  Prints the number of rounds until the last workgroup is done, per policy.
    python tools/sphere_rebalance_model.py gpurun_out/trace_rounds_2048.npz [rays per workgroup]"""
import sys
import numpy as np
from sphere_schedule_model import iterations


def run(its_all, NR, rebalance_at=(), depth=3, nwg=None, order='rr'):
    R = len(its_all)
    nwg = nwg or (R + NR - 1) // NR
    groups = [list(range(w * NR, min(R, (w + 1) * NR))) for w in range(nwg)]
    pos = [0] * R
    done = [np.full(2, -1) for _ in range(R)]
    hard = [np.zeros(2, bool) for _ in range(R)]
    rnd, rows_per_round = 0, []
    wg_rounds = np.zeros(nwg, int)
    while True:
        if rnd in rebalance_at:
            act = [r for r in range(R) if pos[r] < len(its_all[r])]
            if order == 'work':                                   # rays with more iterations left first
                act.sort(key=lambda r: -(len(its_all[r]) - pos[r]))
            groups = [[] for _ in range(nwg)]
            for j, r in enumerate(act):
                groups[j % nwg].append(r)
        any_req, tot = False, 0
        for w, rays in enumerate(groups):
            reqs = []
            for r in rays:
                if pos[r] >= len(its_all[r]):
                    continue
                S1, L, init = its_all[r][pos[r]]
                for s in range(2):
                    if not S1[s]:
                        continue
                    if done[r][s] < 0:
                        reqs.append((r, s, 0))
                    elif done[r][s] < L[s]:
                        reqs.append((r, s, done[r][s] + 1))
            if not reqs:
                continue
            any_req = True
            wg_rounds[w] += 1
            n = len(reqs)
            tiles = (n + 15) // 16
            cand = []
            for d in range(1, depth + 1):
                for (r, s, lv) in reqs:
                    if its_all[r][pos[r]][2]:
                        continue
                    tgt = lv + d
                    if tgt <= 3:
                        cand.append(((0 if hard[r][s] else 1, d), r, s, tgt))
            cand.sort(key=lambda c: c[0])
            spec = [(r, s, tgt) for (_, r, s, tgt) in cand[:tiles * 16 - n]]
            tot += n
            have = {}
            for (r, s, lv) in reqs + spec:
                have.setdefault((r, s), set()).add(lv)
            for (r, s), lvs in have.items():
                L = its_all[r][pos[r]][1][s]
                cur = done[r][s]
                while cur + 1 in lvs:
                    cur += 1
                    if cur >= L:
                        break
                done[r][s] = cur
            for r in rays:
                if pos[r] >= len(its_all[r]):
                    continue
                S1, L, init = its_all[r][pos[r]]
                if all((not S1[s]) or done[r][s] >= L[s] for s in range(2)):
                    for s in range(2):
                        hard[r][s] = S1[s] and L[s] > 0
                    pos[r] += 1
                    done[r][:] = -1
        if not any_req:
            break
        rows_per_round.append(tot)
        rnd += 1
    return rnd, rows_per_round


if __name__ == '__main__':
    d = np.load(sys.argv[1])
    its = iterations(d['req'], d['phase'], d['k'])
    NR = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    for reb in ((), (1,), (2,), (3,), (4,), (2, 5), (1, 3, 6), (2, 4, 6, 8), tuple(range(1, 14))):
        for order in ('rr', 'work'):
            n, rows = run(its, NR, reb, order=order)
            print('re-partition before rounds %-28s %-4s -> %2d rounds; rows per round %s' % (reb, order, n, rows))
