"""Barrier algebra of the 3x3 convolution's half-slab offset pipeline (csrc/gemm.hip, igemm_halo_kernel),
checked on the CPU.  The two halves of the workgroup run DIFFERENT instruction orders between the same barriers; what
makes that legal is two properties of the event order, independent of any timing:

  visibility  a half reads slab s (weights stage s & 1, halo buffer cs & 1) only behind a barrier that every wave
              passed AFTER waiting for its own DMA pieces of that slab;
  reuse       a DMA that overwrites a stage / halo buffer is issued only behind a barrier that every wave passed AFTER
              its last LDS read of the slab that lived there.

The programs below restate the kernel's two loop bodies as event lists (one entry per DMA issue, fragment read, wait
and barrier -- the MFMAs touch registers only); the checker replays them interval by interval.  Lock-step form included
as the reference point.  No GPU, no library: this is a model of the schedule, the kernel itself is covered by the
parity and determinism tests in test_gpu_kernels.py / test_gpu_properties.py."""
import pytest

TAPS = 9


def first_half(s_begin, s_end, cs_begin, cs_end):
    """waves 0 .. NW/2-1 (also the lock-step order of every wave)"""
    ev = [("dma_halo", cs_begin), ("dma_w", s_begin), ("wait_dma",), ("bar",)]
    ev.append(("read", s_begin, 0))
    if s_begin + 1 < s_end:
        ev.append(("dma_w", s_begin + 1))
    if cs_begin + 1 < cs_end:
        ev.append(("dma_halo", cs_begin + 1))
    for slab in range(s_begin, s_end):
        ev.append(("read", slab, 1))
        ev.append(("mma", slab, 0))
        if slab + 1 < s_end:
            ev += [("wait_dma",), ("wait_reads",), ("bar",)]
            ev += issue_next(slab, s_end, cs_end)
            ev.append(("read", slab + 1, 0))
        ev.append(("mma", slab, 1))
    return ev


def second_half(s_begin, s_end, cs_begin, cs_end):
    """waves NW/2 .. NW-1: arrive at BAR(slab+1) with both fragment sets of `slab` unmultiplied"""
    ev = [("dma_halo", cs_begin), ("dma_w", s_begin), ("wait_dma",), ("bar",)]
    if s_begin + 1 < s_end:
        ev.append(("dma_w", s_begin + 1))
    if cs_begin + 1 < cs_end:
        ev.append(("dma_halo", cs_begin + 1))
    ev += [("read", s_begin, 0), ("read", s_begin, 1)]
    for slab in range(s_begin, s_end):
        more = slab + 1 < s_end
        if more:
            ev += [("wait_dma",), ("wait_reads",), ("bar",)]
        ev.append(("mma", slab, 0))
        if more:
            ev += issue_next(slab, s_end, cs_end)
            ev.append(("read", slab + 1, 0))
        ev.append(("mma", slab, 1))
        if more:
            ev.append(("read", slab + 1, 1))
    return ev


def issue_next(slab, s_end, cs_end):
    """behind BAR(slab+1): W(slab+2) into the stage of `slab`; the next channel slab's halo when slab+1 is a first tap"""
    ev = []
    if slab + 2 < s_end:
        ev.append(("dma_w", slab + 2))
    cs = (slab + 1) // TAPS
    if (slab + 1) - cs * TAPS == 0 and cs + 1 < cs_end:
        ev.append(("dma_halo", cs + 1))
    return ev


def intervals(ev):
    """split a program at its barriers; interval i = the events between barrier i-1 and barrier i"""
    out, cur = [], []
    for e in ev:
        if e[0] == "bar":
            out.append(cur)
            cur = []
        else:
            cur.append(e)
    out.append(cur)
    return out


def check(programs, s_begin, s_end):
    ivs = [intervals(p) for p in programs]
    nbar = {len(iv) for iv in ivs}
    assert len(nbar) == 1, f"halves execute different barrier counts: {[len(iv) for iv in ivs]}"
    n = nbar.pop()
    # per program: interval index of every event, and whether a DMA was waited for before the barrier closing an interval
    landed_w, landed_h = {}, {}            # (program, slab) -> index of the first barrier behind which the pieces are in LDS
    for p, iv in enumerate(ivs):
        pending = []
        for i, evs in enumerate(iv):
            for e in evs:
                if e[0] in ("dma_w", "dma_halo"):
                    pending.append(e)
                elif e[0] == "wait_dma":
                    for d in pending:
                        (landed_w if d[0] == "dma_w" else landed_h)[(p, d[1])] = i     # landed before barrier i
                    pending = []
        assert not [d for d in pending if d[0] == "dma_w"], "a weight slab is never waited for"
    nprog = len(programs)
    # visibility + every fragment is read exactly once and multiplied after it was read
    for p, iv in enumerate(ivs):
        have = set()
        for i, evs in enumerate(iv):
            for e in evs:
                if e[0] == "read":
                    slab, cs = e[1], e[1] // TAPS
                    for q in range(nprog):
                        assert (q, slab) in landed_w and landed_w[(q, slab)] < i, f"program {p} reads W({slab}) in interval {i} before program {q}'s pieces are published"
                        assert (q, cs) in landed_h and landed_h[(q, cs)] < i, f"program {p} reads halo({cs}) in interval {i} too early"
                    assert (slab, e[2]) not in have
                    have.add((slab, e[2]))
                elif e[0] == "mma":
                    assert (e[1], e[2]) in have, f"program {p} multiplies ({e[1]}, {e[2]}) before reading it"
        assert have == {(s, k) for s in range(s_begin, s_end) for k in (0, 1)}
    # reuse: last read of the previous tenant of a stage / halo buffer, over ALL programs, lies in an earlier interval
    last_read_w, last_read_h = {}, {}
    for p, iv in enumerate(ivs):
        for i, evs in enumerate(iv):
            for e in evs:
                if e[0] == "read":
                    last_read_w[e[1]] = max(last_read_w.get(e[1], -1), i)
                    last_read_h[e[1] // TAPS] = max(last_read_h.get(e[1] // TAPS, -1), i)
    for p, iv in enumerate(ivs):
        for i, evs in enumerate(iv):
            for e in evs:
                if e[0] == "dma_w" and e[1] - 2 >= s_begin:
                    assert last_read_w[e[1] - 2] < i, f"program {p} overwrites the stage of slab {e[1] - 2} in interval {i} while it is still read"
                if e[0] == "dma_halo" and (e[1] - 2) * TAPS >= s_begin:
                    assert last_read_h[e[1] - 2] < i, f"program {p} overwrites the halo of channel slab {e[1] - 2} too early"
    # and the reads that precede a barrier are waited for in front of it (the kernel's s_waitcnt lgkmcnt(0))
    for p, iv in enumerate(ivs):
        for evs in iv[1:-1]:
            if any(e[0] == "read" for e in evs):
                assert ("wait_reads",) in evs
    return n


@pytest.mark.parametrize("cslabs", [1, 2, 5, 20])
@pytest.mark.parametrize("splitk", [1, 2, 3])
def test_offset_pipeline_is_a_legal_schedule(cslabs, splitk):
    if splitk > cslabs:
        pytest.skip("split-K covers whole channel slabs")
    for z in range(splitk):
        cs_begin, cs_end = cslabs * z // splitk, cslabs * (z + 1) // splitk
        s_begin, s_end = cs_begin * TAPS, cs_end * TAPS
        a = first_half(s_begin, s_end, cs_begin, cs_end)
        b = second_half(s_begin, s_end, cs_begin, cs_end)
        nb = check([a, b], s_begin, s_end)
        assert nb == check([a, a], s_begin, s_end)            # same barrier count as the lock-step form
        assert nb == (s_end - s_begin) + 1


def test_the_checker_catches_an_illegal_order():
    """a second half that fetches slab s+2 BEFORE the barrier (i.e. while the first half may still read slab s) must fail"""
    def bad_second(s_begin, s_end, cs_begin, cs_end):
        ev = second_half(s_begin, s_end, cs_begin, cs_end)
        for i in range(len(ev) - 2):
            if ev[i][0] == "bar" and i > 4 and ev[i + 2][0] == "dma_w":
                ev[i - 2:i + 3] = [ev[i + 2]] + ev[i - 2:i + 2]      # move one DMA in front of the waits and the barrier
                break
        return ev
    with pytest.raises(AssertionError, match="overwrites the stage"):
        check([first_half(0, 18, 0, 2), bad_second(0, 18, 0, 2)], 0, 18)
