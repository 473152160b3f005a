"""Ring / landing protocol of the experimental three-group main loop (mint_amd/csrc/gemm_big.hip big_mainloop3), replayed on
the CPU: twelve waves in three groups, group g running g barrier-slots behind group 0, every slot closed by one workgroup
barrier; per K step a wave issues its DMA pieces of stage j+2 (A), reads the fragments of stage j (B), multiplies (C).
A piece may land at ANY time between its issue and the counted wait of its own wave that retires it, so the checks are

  * landing: when a wave reads stage j, every wave has passed - at least one barrier earlier - a wait that retired its
    pieces of stage j;
  * re-use:  when a wave issues a piece into a ring slot, every read of the stage that occupied the slot finished in an
    earlier barrier-slot.

The schedule constants below are the ones the kernel uses; the test fails if any of them is off by one."""
import pytest

NST, DIST = 4, 2                     # ring slots, prefetch distance in K steps
WAIT = {0: ("A", 2), 1: ("C", 1), 2: ("B", 1)}   # group -> (phase after which it waits, stages allowed to stay in flight)


def replay(nk, wait=WAIT, dist=DIST, nst=NST):
    # program of one group: prologue stages 0..dist-1 retired before slot 0; then per step j: A(j), B(j), C(j)
    issue_slot, retire_slot, read_slot = {}, {}, {}          # (group, stage) -> global barrier-slot index
    for g in range(3):
        issued = list(range(dist))                            # prologue
        for s in issued:
            issue_slot[(g, s)] = -10
            retire_slot[(g, s)] = -10
        t = g                                                 # group g enters its loop g slots late
        for j in range(nk):
            for phase in "ABC":
                if phase == "A":
                    st = j + dist                             # stages past the end of K are still issued (into dead slots)
                    issued.append(st)
                    issue_slot[(g, st)] = t
                if phase == "B":
                    read_slot[(g, j)] = t
                ph, left = wait[g]
                if phase == ph:                               # counted vmcnt: all but the `left` youngest stages retired
                    for s in issued[:len(issued) - left]:
                        retire_slot.setdefault((g, s), t)
                t += 1
    return issue_slot, retire_slot, read_slot


def violations(nk, **kw):
    issue_slot, retire_slot, read_slot = replay(nk, **kw)
    nst = kw.get("nst", NST)
    bad = []
    for (g, j), t in read_slot.items():
        for g2 in range(3):                                   # landing: retired by everyone in an EARLIER slot
            r = retire_slot.get((g2, j))
            if r is None or r >= t:
                bad.append(("landing", g, j, t, g2, r))
    for (g, st), t in issue_slot.items():
        old = st - nst                                        # previous occupant of the ring slot
        if old < 0 or t < 0:
            continue
        for g2 in range(3):
            r = read_slot.get((g2, old))
            if r is not None and r >= t:
                bad.append(("reuse", g, st, t, g2, r))
    return bad


@pytest.mark.parametrize("nk", [1, 2, 3, 4, 5, 8, 25, 96])
def test_shipped_schedule_is_safe(nk):
    assert violations(nk) == []


def test_barrier_counts_match_for_all_groups():
    # group g: g skew barriers + 3 per K step + (2 - g) drain barriers
    for nk in (1, 7, 25):
        assert len({g + 3 * nk + (2 - g) for g in range(3)}) == 1


@pytest.mark.parametrize("change", [
    {"dist": 3},                                             # one more step of prefetch overwrites a slot group 2 still reads
    {"wait": {0: ("A", 3), 1: ("C", 1), 2: ("B", 1)}},      # group 0 leaving three stages in flight reads un-landed data
    {"wait": {0: ("A", 2), 1: ("C", 2), 2: ("B", 1)}},
    {"wait": {0: ("A", 2), 1: ("C", 1), 2: ("B", 2)}},
    {"wait": {0: ("B", 2), 1: ("C", 1), 2: ("B", 1)}},      # group 0 waiting one phase later is too late for its own read
])
def test_the_checker_catches_off_by_one_schedules(change):
    assert violations(25, **change), change
