#!/usr/bin/env python3
"""Discrete-event model of the fused factor + inverse launch (kernels_chol.hip: potrf_dataflow_kernel with potri_team): the chain,
W1 workers that own the factorisation's tiles, G2 workgroups that own the inverse's items -- STATIC ownership as built (round-robin
deals, first ready task in list order) against a DYNAMIC pool (any free workgroup takes the ready task that is first in the global
order; what a per-item lock + shared queue would give).  Durations are the measured ones (us, POTRF_BENCH_TRACE / potri trace at
N = 4096): a K = 128 tile task 19.6, a K = 256 chunk 33, panel solve 14 behind its diagonal block, diagonal block 19.5, the chain's
product 10.7, its solve's tail 3, T(j) 25, P(i) 25.  Answers ONE question before any kernel code: how much of the 2.15 ms at N = 4096
is the static ownership?   usage: potri_sched_sim.py NB [W1] [G2]"""
import heapq
import sys


def build(nb, cx, ck, nbo=1, escort=None):
    """items: list of dicts(kind, key, tasks=[(inputs, dur, outputs)]) in the kernel's global order; inputs/outputs are flag names.
    nbo > 1 (large N): the factorisation's updates run in chunks of nbo steps (single steps for the tiles next to the diagonal), and a
    chunk of n K = 128 units costs 4 + 14.8 n us (POTRF_BENCH_TRACE at N = 8192: 37.5 us for 2.37 units; 19.6 / 33 for one / two)."""
    fac, inv = [], []
    esc = []     # escort = (band, U): the last U updates (and the solve) of the tiles within `band` blocks of the diagonal are items of their own
    T_UPD, T_UPD2, T_SOLVE, T_T, T_P = 19.6, 33.0, 14.0, 25.0, 25.0

    def dur(n):
        return T_UPD if n == 1 else T_UPD2 if n == 2 else 4.0 + 14.8 * n
    for k in range(nb):
        for i in range(k, nb):
            if i == 0:
                continue
            target = k - 1 if i == k else k
            tasks = []
            step = 1 if i - k <= 2 else nbo
            split_at = target
            if escort and i - k <= escort[0] and target > 0:
                split_at = max(0, target - escort[1])
                step = nbo if split_at > 0 else 1     # what stays with the owner is no longer urgent: full chunks
            j = 0
            while j < split_at:
                e = min(j + step, split_at)
                outs = []
                if e == target and i <= k + 1:
                    outs = [("diag_ready", k - 1) if i == k else ("chain_ready", k)]
                if e == split_at and split_at < target:
                    outs = outs + [("pre", i, k)]
                ins = []
                for q in range(j, e):
                    ins += [("panel", i, q)] + ([("panel", k, q)] if i != k else [])
                tasks.append((ins, dur(e - j), outs))
                j = e
            etasks = []
            j = split_at
            while j < target:
                outs = []
                if j + 1 == target and i <= k + 1:
                    outs = [("diag_ready", k - 1) if i == k else ("chain_ready", k)]
                ins = ([("pre", i, k)] if (j == split_at and split_at > 0) else []) + [("panel", i, j)] + ([("panel", k, j)] if i != k else [])
                etasks.append((ins, dur(1), outs))
                j += 1
            if target == 0 and i <= k + 1:
                tasks.append(([], 0.0, [("diag_ready", k - 1) if i == k else ("chain_ready", k)]))
            if i > k + 1:
                (etasks if etasks else tasks).append(([("fact_start", k)], T_SOLVE, [("panel", i, k)], ("after_fact", k)))
            if tasks:
                fac.append(dict(key=("F", i, k), tasks=tasks))
            if etasks:
                esc.append(dict(key=("E", i, k), tasks=etasks))
    for r in range(nb):
        inv.append(dict(key=("T", r), tasks=[([("fact", r)], T_T, [("x", r, r)])]))
        if r > 0:
            inv.append(dict(key=("P", r), tasks=[([("x", r, r), ("panel", r, r - 1)], T_P, [("p", r)])]))
        for j in range(r):
            tasks = []
            nterms = r - j - 1
            d = 0
            while d < nterms:
                e = min(d + cx, nterms)
                ins = []
                for k in range(j + d, j + e):
                    ins += [("x", k, j), ("panel", r, k)]
                tasks.append((ins, dur(e - d), []))
                d = e
            if nterms > 0:
                tasks.append(([("x", r, r)], T_UPD, []))
            tasks.append(([("x", r - 1, j), ("p", r)], T_UPD, [("x", r, j)]))
            inv.append(dict(key=("X", r, j), tasks=tasks))
    for r in range(nb):
        for j in range(r + 1):
            tasks = []
            d = r
            while d < nb:
                e = min(d + ck, nb)
                ins = []
                for k in range(d, e):
                    ins += [("x", k, r), ("x", k, j)]
                tasks.append((ins, dur(e - d), []))
                d = e
            inv.append(dict(key=("K", r, j), tasks=tasks))
    return (fac, inv, esc) if escort else (fac, inv)


def simulate(nb, W1, G2, dynamic, cx=2, ck=2, verbose=True, t_over=0.0, nbo=1, with_inverse=True, escort=None, n_escort=0, order_c=0.0):
    """escort = (band, U), n_escort = E (static ownership only): E of the W1 workers own nothing but the last U updates (and the solves) of the
    tiles within `band` blocks of the diagonal, dealt round-robin."""
    esc = []
    if escort:
        fac, inv, esc = build(nb, cx, ck, nbo, escort)
        W1 -= n_escort
    else:
        fac, inv = build(nb, cx, ck, nbo)
    if not with_inverse:
        inv = []
    if order_c:
        # list order by k + c (i - k) instead of column by column: what is near the diagonal in a LATER column goes before the stragglers
        # far below the diagonal in an earlier one (stable: ties keep the column-major order)
        fac.sort(key=lambda it: it["key"][2] + order_c * (it["key"][1] - it["key"][2]))
    T_DIAG, T_SYRK, T_TAIL, T_STREAM = 19.5, 10.7, 3.0, 8.0
    t_flag = {}                     # flag -> time it was raised
    waiting = {}                    # flag -> list of callbacks
    INF = float("inf")

    def ready_time(ins):
        t = 0.0
        for f in ins:
            if f not in t_flag:
                return INF
            t = max(t, t_flag[f])
        return t

    # owners
    if dynamic == 1:
        pools = [list(range(len(fac) + len(inv)))]
        n_wg = [W1 + G2]
    elif dynamic == 2:                                      # one pool per team
        pools = [list(range(len(fac))), list(range(len(fac), len(fac) + len(inv)))]
        n_wg = [W1, G2]
    elif dynamic == 4:                                      # one pool per XCD: items dealt round-robin over 8 pools of (W1 + G2) / 8 servers
        pools = [[] for _ in range(8)]
        for n in range(len(fac) + len(inv)):
            pools[n % 8].append(n)
        tot = W1 + G2
        n_wg = [tot // 8 + (1 if x < tot % 8 else 0) for x in range(8)]
    elif dynamic == 3:                                      # static factorisation, pooled inverse
        pools = [[] for _ in range(W1)] + [list(range(len(fac), len(fac) + len(inv)))]
        for n in range(len(fac)):
            pools[n % W1].append(n)
        n_wg = [1] * W1 + [G2]
    else:
        pools = [[] for _ in range(W1 + G2 + n_escort)]
        for n in range(len(fac)):
            pools[n % W1].append(n)
        for n in range(len(inv)):
            pools[W1 + n % G2].append(len(fac) + n)
        for n in range(len(esc)):
            pools[W1 + G2 + n % n_escort].append(len(fac) + len(inv) + n)
        n_wg = [1] * (W1 + G2 + n_escort)
    items = fac + inv + esc
    pos = [0] * len(items)
    # chain state
    chain = dict(j=0, free=T_DIAG)
    t_flag[("fact_start", 0)] = 0.0
    t_flag[("fact", 0)] = T_DIAG
    t_flag[("after_fact", 0)] = T_DIAG

    def chain_advance():
        # block j+1 needs chain_ready[j] (tile (j+1, j)) and diag_ready[j]
        while chain["j"] <= nb - 2:
            j = chain["j"]
            u1, u2 = t_flag.get(("chain_ready", j)), t_flag.get(("diag_ready", j))
            if u1 is None or u2 is None:
                return
            solved = max(t_flag[("fact", j)], u1 + T_STREAM) + T_TAIL
            t_flag[("panel", j + 1, j)] = solved
            start = max(solved + T_SYRK, u2)
            t_flag[("fact_start", j + 1)] = start
            t_flag[("fact", j + 1)] = start + T_DIAG
            t_flag[("after_fact", j + 1)] = start + T_DIAG
            chain["j"] = j + 1

    # event loop: each pool has n_wg servers; pick earliest-free server, give it the first item (in list order) whose next task is
    # ready by then (else the earliest-ready one)
    servers = []
    for p, n in enumerate(n_wg):
        for _ in range(n):
            servers.append([0.0, p])
    heapq.heapify(servers)
    busy = [0.0] * len(pools)
    done_items = 0
    total = len(items)
    last = 0.0
    claimed = set()
    stall_guard = 0
    while done_items < total:
        chain_advance()
        t_free, p = heapq.heappop(servers)
        best, best_t = None, INF
        for n in pools[p]:
            if pos[n] >= len(items[n]["tasks"]) or n in claimed:
                continue
            task = items[n]["tasks"][pos[n]]
            rt = ready_time(task[0])
            if len(task) > 3 and rt < INF:                 # streamed solve: cannot end before its diagonal block has
                pass
            if rt <= t_free:
                best, best_t = n, rt
                break
            if rt < best_t:
                best, best_t = n, rt
        if best is None or best_t == INF:
            # nothing startable yet for this server: let time pass to the next flag (crude: retry a little later)
            heapq.heappush(servers, [t_free + 2.0, p])
            stall_guard += 1
            if stall_guard > 5_000_000:
                raise RuntimeError("stalled")
            continue
        n = best
        task = items[n]["tasks"][pos[n]]
        start = max(t_free, best_t)
        end = start + task[1] + (t_over if task[1] > 0 else 0.0)
        if len(task) > 3:
            af = t_flag.get(task[3])
            if af is None:                                  # its diagonal block's end is not known yet: wait
                heapq.heappush(servers, [t_free + 2.0, p])
                continue
            end = max(end, af + T_TAIL)
        busy[p] += end - start
        for f in task[2]:
            t_flag[f] = end
        pos[n] += 1
        if pos[n] == len(items[n]["tasks"]):
            done_items += 1
        last = max(last, end)
        heapq.heappush(servers, [end, p])
    chain_advance()
    chain_end = t_flag[("fact", nb - 1)]
    last = max(last, chain_end)
    if verbose:
        nserv = sum(n_wg)
        print(f"nb={nb} {['static ', 'one pool', 'pool per team', 'static fact + pooled inv', 'pool per XCD'][dynamic]} W1={W1} G2={G2}: chain ends {chain_end:7.0f} us ({chain_end / nb:5.1f} per step), "
              f"all done {last:7.0f} us, mean busy {sum(busy) / nserv:6.0f} us")
    return last, chain_end


if __name__ == "__main__":
    nb = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    W1 = int(sys.argv[2]) if len(sys.argv) > 2 else 96
    G2 = int(sys.argv[3]) if len(sys.argv) > 3 else 158
    for mode in (0, 1, 4):
        simulate(nb, W1, G2, mode)
