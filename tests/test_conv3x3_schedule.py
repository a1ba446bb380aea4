"""Loader / compute protocol of the 3x3 halo convolution on 32 x 32 MFMA tiles (csrc/conv3x3.hip), replayed on the CPU.

Four loader waves issue DMA (weight slab two ahead into a three-stage ring, two halo pieces of the next channel slab per
window, residual row chunks in the last channel slab) and wait with COUNTED vmcnt -- "everything but the youngest n issues
has landed" -- in front of the one barrier per slab; four compute waves read fragments and multiply.  What makes the
protocol legal is independent of timing:

  visibility  a fragment read of slab (cs, tap) happens behind a barrier that every loader passed AFTER a wait that
              retired its pieces of that weight slab and of channel slab cs's halo;
  reuse       a DMA into a ring stage / halo buffer is issued behind a barrier that the compute waves passed AFTER their
              last read of what lived there (they wait for their LDS reads in front of every barrier).

The event lists below restate the kernel's two loops (window(), the wait counts, the barrier placement between the third
and the fourth k-step); the checker replays them.  A model of the schedule -- the kernel itself is covered by the parity
and determinism tests in test_gpu_kernels.py / test_gpu_properties.py."""
import pytest

WQ, NSTG = 5, 3


def hcount(nq, tap):
    return 0 if tap > 6 else max(0, min(2, nq - 2 * tap))


def rcount(rq, res_on, tap, last):
    if not (last and res_on and 1 <= tap <= 7):
        return 0
    return max(0, min(3, rq - (tap - 1) * 3))


def loader_program(ncs, nq, rq, res_on, miscount=0):
    """one loader wave: nq valid halo slots (12 or 13 of its 13), rq residual chunks per lane (20 or 10)"""
    ev = []
    for q in range(nq):
        ev.append(("dma", "halo", 0, ("hbuf", 0)))
    for tap in (0, 1):
        ev += [("dma", "w", (0, tap), ("stage", tap % NSTG))] * WQ
    ev += [("wait_vm", WQ), ("bar",)]
    for cs in range(ncs):
        last = cs + 1 >= ncs
        for tap in range(9):
            # window (cs, tap)
            if tap + 2 < 9:
                tgt = (cs, tap + 2)
            elif not last:
                tgt = (cs + 1, tap + 2 - 9)
            else:
                tgt = None
            if tgt is not None:
                ev += [("dma", "w", tgt, ("stage", tgt[1] % NSTG))] * WQ
            if not last and tap <= 6:
                for q in (2 * tap, 2 * tap + 1):
                    if q < nq:
                        ev.append(("dma", "halo", cs + 1, ("hbuf", (cs + 1) & 1)))
            for _ in range(rcount(rq, res_on, tap, last)):
                ev.append(("dma", "res", None, ("vgpr",)))
            # the wait in front of the barrier that publishes the next slab
            n = (WQ if tgt is not None else 0) + (0 if last else hcount(nq, tap)) + rcount(rq, res_on, tap, last)
            if tap >= 1:
                n += (0 if last else hcount(nq, tap - 1)) + rcount(rq, res_on, tap - 1, last)
            ev += [("wait_vm", n + miscount), ("bar",)]
    ev.append(("wait_vm", 0))
    return ev


def compute_program(ncs):
    ev = [("bar",), ("read", (0, 0), 0)]
    for cs in range(ncs):
        for tap in range(9):
            ev += [("read", (cs, tap), 1), ("mma", (cs, tap), 0), ("read", (cs, tap), 2), ("mma", (cs, tap), 1),
                   ("read", (cs, tap), 3), ("mma", (cs, tap), 2), ("wait_reads",), ("bar",)]
            nxt = (cs, tap + 1) if tap < 8 else (cs + 1, 0)
            if nxt[0] < ncs:
                ev.append(("read", nxt, 0))
            ev.append(("mma", (cs, tap), 3))
    return ev


def intervals(ev):
    out, cur = [], []
    for e in ev:
        if e[0] == "bar":
            out.append(cur)
            cur = []
        else:
            cur.append(e)
    out.append(cur)
    return out


def check(loaders, compute, ncs):
    liv = [intervals(p) for p in loaders]
    civ = intervals(compute)
    assert {len(iv) for iv in liv} == {len(civ)}, "loaders and compute waves execute different barrier counts"
    # when does every DMA land?  vmcnt retires in issue order: a wait_vm(n) retires all but the youngest n issues
    landed = []                                   # per loader: {(kind, key): interval index of the retiring wait}
    issued_in = []                                # per loader: [(interval, kind, key, region)]
    for iv in liv:
        queue, last_land, log = [], {}, []
        for i, evs in enumerate(iv):
            for e in evs:
                if e[0] == "dma":
                    queue.append((e[1], e[2]))
                    log.append((i, e[1], e[2], e[3]))
                elif e[0] == "wait_vm":
                    while len(queue) > e[1]:
                        last_land[queue.pop(0)] = i      # (a slab issued as several pieces lands when its LAST piece does)
        assert not queue, "DMA still in flight at the end of the loop"
        landed.append(last_land)
        issued_in.append(log)
    # visibility and completeness of the compute program
    have, last_read = set(), {}
    for i, evs in enumerate(civ):
        for e in evs:
            if e[0] == "read":
                slab, j = e[1], e[2]
                for ld in landed:
                    assert ("w", slab) in ld and ld[("w", slab)] < i, f"W{slab} read in interval {i} before a loader retired it"
                    if ("halo", slab[0]) in ld:          # (a loader with no piece of this halo has nothing to retire)
                        assert ld[("halo", slab[0])] < i, f"halo {slab[0]} read in interval {i} before a loader retired it"
                assert (slab, j) not in have
                have.add((slab, j))
                last_read[("stage", slab[1] % NSTG, slab)] = i
                last_read[("hbuf", slab[0] & 1, slab[0])] = i
            elif e[0] == "mma":
                assert (e[1], e[2]) in have, f"{e[1]} k-step {e[2]} multiplied before it was read"
    assert have == {((cs, tap), j) for cs in range(ncs) for tap in range(9) for j in range(4)}
    for evs in civ[1:-1]:
        if any(e[0] == "read" for e in evs):
            assert ("wait_reads",) in evs, "reads in flight across a barrier"
    # reuse: the previous tenant of a region was read for the last time in an EARLIER interval than the DMA that replaces it
    for log in issued_in:
        for i, kind, key, region in log:
            if kind == "w":
                cs, tap = key
                k = cs * 9 + tap
                if k >= NSTG:
                    pk = k - NSTG
                    prev = (pk // 9, pk % 9)
                    assert last_read[("stage", prev[1] % NSTG, prev)] < i, f"W{key} overwrites the stage of {prev} while it is read"
            elif kind == "halo" and key >= 2:
                assert last_read[("hbuf", key & 1, key - 2)] < i, f"halo {key} overwrites channel slab {key - 2}'s while it is read"
    return len(civ)


@pytest.mark.parametrize("ncs", [1, 2, 5, 10, 20])
@pytest.mark.parametrize("res_on", [False, True])
def test_loader_compute_protocol_is_legal(ncs, res_on):
    # the four loaders of a 256-token tile at 64 x 64 (50 halo pieces: 13, 13, 12, 12 slots) and of a 128-token tile (33: 9, 8, 8, 8)
    for slots, rq in (((13, 13, 12, 12), 20), ((9, 8, 8, 8), 10)):
        loaders = [loader_program(ncs, nq, rq, res_on) for nq in slots]
        nb = check(loaders, compute_program(ncs), ncs)
        assert nb == 9 * ncs + 2                 # one barrier per slab + the one that publishes slab 0 (+ the tail interval)


def test_the_checker_catches_a_wrong_count():
    """a loader that lets one more issue fly than the protocol allows publishes a slab whose last piece may not have landed"""
    loaders = [loader_program(2, 13, 20, True, miscount=1)] + [loader_program(2, 12, 20, True)] * 3
    with pytest.raises(AssertionError, match="before a loader retired it"):
        check(loaders, compute_program(2), 2)


def test_the_checker_catches_an_early_refill():
    """a weight slab issued THREE ahead would land in the stage the compute waves are still reading"""
    def eager(ncs, nq):
        ev = loader_program(ncs, nq, 20, False)
        out = []
        for e in ev:
            if e[0] == "dma" and e[1] == "w":
                cs, tap = e[2]
                k = cs * 9 + tap + 1                      # one slab further ahead, same ring position
                if k < 9 * ncs:
                    e = ("dma", "w", (k // 9, k % 9), ("stage", (k % 9) % NSTG))
            out.append(e)
        return out
    with pytest.raises(AssertionError):
        check([eager(2, 13)] * 4, compute_program(2), 2)
