#!/usr/bin/env python3
"""Discrete-event model of the dataflow Cholesky schedule (kernels_chol.hip: potrf_dataflow_kernel): one chain workgroup +
W workers that own tiles cyclically and run the first ready task in column order.  Used to choose the owner grid and the
chunking rule before spending GPU time; durations are the measured ones (us): a 128^3 product on one CU 13.6, read-modify-write
of a tile 3.5, scheduling round + acquire 3.5, chain step 70 (diag 36 + products 34).
usage: potrf_dataflow_sim.py NB [PR] [NBO] [W]"""
import heapq
import sys


def chunks_of(i, k, nbo, near):
    """Update chunks (j0, j1) of tile (i, k): df_chunk_end of the kernel."""
    target = k - 1 if i == k else k
    out, d = [], 0
    while d < target:
        if d // nbo < k // nbo and not (near and d // nbo == k // nbo - 1 and k % nbo < near):
            j1 = (d // nbo + 1) * nbo
        else:
            j1 = d + 1
        j1 = min(j1, target)
        out.append((d, j1))
        d = j1
    return out


def simulate(nb, PR=12, nbo=8, W=255, chain_step=(13.0, 21.0, 36.0), t_gemm=13.6, t_rmw=3.5, t_sched=3.5, near=0, verbose=False):
    PC = max(1, W // PR)
    owner = lambda i, k: (i % PR) + PR * (k % PC)
    chunks = lambda i, k: chunks_of(i, k, nbo, near)
    tiles = {}
    for k in range(nb):
        for i in range(k, nb):
            if i == 0:
                continue
            tiles[(i, k)] = dict(ch=chunks(i, k), pos=0, fin=False)
    mine = [[] for _ in range(PR * PC)]
    for (i, k) in sorted(tiles, key=lambda t: (t[1], t[0])):
        mine[owner(i, k)].append((i, k))
    INF = float("inf")
    panel_t = {}          # (i, j) -> time L_ij available
    fact_t = {}           # j -> time
    chain_ready = {j: [] for j in range(nb)}
    # event-driven: workers pick tasks when free; chain progresses when its tiles are ready
    free_at = [0.0] * (PR * PC)
    busy = [0.0] * (PR * PC)
    first = [0] * (PR * PC)
    # chain state
    g1, g2, dg = chain_step
    fact_t[0] = dg
    chain_j, chain_free = 0, dg
    chain_wait = 0.0
    now = 0.0
    # simple time-stepped loop over events: process in global time order using a heap of worker wake-ups
    heap = [(0.0, w) for w in range(PR * PC) if mine[w]]
    heapq.heapify(heap)
    pending_chain = True
    def chain_try(t):
        nonlocal chain_j, chain_free, chain_wait
        while chain_j <= nb - 2 and len(chain_ready[chain_j]) == 2:
            start = max(chain_free, max(chain_ready[chain_j]))
            chain_wait += max(0.0, max(chain_ready[chain_j]) - chain_free)
            panel_t[(chain_j + 1, chain_j)] = start + g1
            fact_t[chain_j + 1] = start + g1 + g2 + dg
            chain_free = start + g1 + g2 + dg
            chain_j += 1
    guard = 0
    while heap:
        guard += 1
        t, w = heapq.heappop(heap)
        chain_try(t)
        lst = mine[w]
        while first[w] < len(lst) and tiles[lst[first[w]]]["fin"]:
            first[w] += 1
        if first[w] >= len(lst):
            continue
        # earliest-ready task in the window, ready at time <= t?
        best, best_ready = None, INF
        for (i, k) in lst[first[w]:first[w] + 16]:
            T = tiles[(i, k)]
            if T["fin"]:
                continue
            if T["pos"] < len(T["ch"]):
                j0, j1 = T["ch"][T["pos"]]
                r = 0.0
                for j in range(j0, j1):
                    r = max(r, panel_t.get((i, j), INF), panel_t.get((k, j), INF) if i != k else 0.0)
            elif i > k + 1:
                r = fact_t.get(k, INF)
            else:
                r = 0.0
            if r <= t:
                best, best_ready = (i, k), r
                break
            if r < best_ready:
                best_ready = r
                cand = (i, k)
        if best is None:
            # sleep until something may be ready (unknown producers: poll)
            nxt = best_ready if best_ready < INF else t + 5.0
            heapq.heappush(heap, (max(nxt, t + 1.0), w))
            continue
        i, k = best
        T = tiles[best]
        if T["pos"] < len(T["ch"]):
            j0, j1 = T["ch"][T["pos"]]
            dur = t_sched + (j1 - j0) * t_gemm + t_rmw
            T["pos"] += 1
            end = t + dur
            if T["pos"] == len(T["ch"]) and i <= k + 1:
                T["fin"] = True
                chain_ready[k - 1 if i == k else k].append(end)
        elif i > k + 1:
            dur = t_sched + t_gemm + t_rmw
            end = t + dur
            panel_t[(i, k)] = end
            T["fin"] = True
        else:
            dur, end = 0.5, t + 0.5
            T["fin"] = True
            chain_ready[k - 1 if i == k else k].append(end)
        busy[w] += dur
        heapq.heappush(heap, (end, w))
        chain_try(end)
    chain_try(INF)
    total = chain_free
    act = [b for b, m in zip(busy, mine) if m]
    unfinished = sum(1 for t in tiles.values() if not t["fin"])
    if chain_j <= nb - 2:
        total = float("inf")                      # the chain never got its tiles: a circular wait in the schedule
    return dict(total_us=total, chain_wait_us=chain_wait, busy_max=max(act), busy_mean=sum(act) / len(act), workers=len(act),
                steps_done=chain_j, tiles_unfinished=unfinished)


if __name__ == "__main__":
    nb = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    for PR in (6, 12, 16, 24):
        for nbo in (1, 2, 4, 8):
            for near in (0, 3):
                if near and nbo == 1:
                    continue
                r = simulate(nb, PR=PR, nbo=nbo, near=near)
                print(f"nb={nb} PR={PR:2d} nbo={nbo} near={near}: total {r['total_us']/1000:6.3f} ms  chain waits {r['chain_wait_us']/1000:6.3f} ms  worker busy max/mean {r['busy_max']/1000:5.2f}/{r['busy_mean']/1000:5.2f} ms")
