#!/usr/bin/env python3
"""Model of the barrier schedule of igemm_bf3_ws.h (habitat-lab_amd/csrc/igemm_bf3_ws.h): replays the producer and the consumer
program of one workgroup as two instruction lists cut at their barriers, executes them phase by phase (everything between barrier n and
barrier n + 1 of BOTH roles is concurrent) and checks, for every ntk:
  * both roles execute the same number of barriers;
  * a consumer read of LDS image i sees k-tile kt complete, and no producer write to image i is concurrent with a consumer read of i;
  * a gather of register set s is never issued while set s still holds a k-tile that has not been staged, and stage(kt) finds kt in its set;
  * shared gather keys: fetch(kt) reads key buffer kt % 3 holding keys(kt), written in an EARLIER phase; no write to a buffer that is
    read in the same phase.
usage: python tools/ws_schedule_model.py   (prints OK)"""


def producer(ntk, ksh):
    ops = []
    B = lambda: ops.append(("bar",))
    if ksh:
        for q in range(min(3, ntk)):
            ops.append(("keys", q, q))
        B()
    if ntk > 0:
        ops.append(("fetch", 0, 0, 0))
        if ntk > 1:
            ops.append(("fetch", 1, 1, 1))
        ops.append(("stage", 0, 0))
    B()
    k3 = 0
    kt = 0

    def step(S, kt):
        nonlocal k3
        if ksh and kt + 3 < ntk:
            ops.append(("keys", kt + 3, k3))
        if kt + 2 < ntk:
            ops.append(("fetch", S, kt + 2, 2 if k3 == 0 else k3 - 1))
        if kt + 1 < ntk:
            ops.append(("stage", 1 - S, kt + 1))
        k3 = 0 if k3 == 2 else k3 + 1

    while kt + 3 < ntk:  # steady state
        if ksh:
            ops.append(("keys", kt + 3, k3))
        ops.append(("fetch", 0, kt + 2, 2 if k3 == 0 else k3 - 1))
        ops.append(("stage", 1, kt + 1))
        k3 = 0 if k3 == 2 else k3 + 1
        B()
        if ksh and kt + 4 < ntk:
            ops.append(("keys", kt + 4, k3))
        ops.append(("fetch", 1, kt + 3, 2 if k3 == 0 else k3 - 1))
        ops.append(("stage", 0, kt + 2))
        k3 = 0 if k3 == 2 else k3 + 1
        B()
        kt += 2
    while kt < ntk:
        step(0, kt)
        B()
        if kt + 1 < ntk:
            step(1, kt + 1)
        B()
        kt += 2
    return ops


def consumer(ntk, ksh):
    ops = []
    B = lambda: ops.append(("bar",))
    if ksh:
        B()
    B()
    kt = 0
    while kt + 3 < ntk:
        ops.append(("mfma", 0, kt)); B()
        ops.append(("mfma", 1, kt + 1)); B()
        kt += 2
    while kt < ntk:
        ops.append(("mfma", 0, kt)); B()
        if kt + 1 < ntk:
            ops.append(("mfma", 1, kt + 1))
        B()
        kt += 2
    return ops


def phases(ops):
    out, cur = [], []
    for o in ops:
        if o[0] == "bar":
            out.append(cur); cur = []
        else:
            cur.append(o)
    out.append(cur)
    return out


def check(ntk, ksh):
    pp, cp = phases(producer(ntk, ksh)), phases(consumer(ntk, ksh))
    assert len(pp) == len(cp), (ntk, ksh, len(pp), len(cp))
    image = {0: None, 1: None}      # k-tile held, complete at a barrier
    regset = {0: None, 1: None}     # k-tile gathered, not yet staged
    keys = {0: None, 1: None, 2: None}
    done = []
    for ph, (po, co) in enumerate(zip(pp, cp)):
        reads = {o[1] for o in co}
        key_reads, key_writes = set(), {}
        new_image, new_keys = dict(image), dict(keys)
        for o in po:
            if o[0] == "keys":
                _, kt, kb = o
                key_writes[kb] = kt
                new_keys[kb] = kt
            elif o[0] == "fetch":
                _, s, kt, kb = o
                assert regset[s] is None, ("gather into a full set", ntk, ksh, ph, o)
                if ksh:
                    assert keys[kb] == kt and kb == kt % 3, ("keys", ntk, ph, o, keys)
                    key_reads.add(kb)
                regset[s] = kt
            elif o[0] == "stage":
                _, s, kt = o
                assert regset[s] == kt and s == kt % 2, ("stage", ntk, ksh, ph, o, regset)
                assert s not in reads, ("write to an image being read", ntk, ksh, ph, o)
                regset[s] = None
                new_image[s] = kt
        assert not (key_reads & set(key_writes)), ("key buffer written while read", ntk, ph)
        for o in co:
            _, s, kt = o
            assert image[s] == kt, ("consumer reads", ntk, ksh, ph, o, image)
            done.append(kt)
        image, keys = new_image, new_keys
    assert done == list(range(ntk)), (ntk, ksh, done)


def check_obs(ntiles_of_wg):
    """obs_conv_bf3_ws.h: a workgroup's tiles j = 0 .. n-1, four k-tiles each; register set = k-tile index, image = kt & 1; gathers run
    one tile ahead (the last tile re-reads itself: those sets are never staged)."""
    n = ntiles_of_wg
    P, Cn = [], []
    Pb = lambda: P.append(("bar",))
    Cb = lambda: Cn.append(("bar",))
    for k in range(4):
        P.append(("fetch", k, (0, k)))
    P.append(("stage", 0, (0, 0)))
    Pb(); Cb()
    for j in range(n):
        more = j + 1 < n
        nxt = j + 1 if more else j
        for k in range(4):
            P.append(("fetch", k, (nxt, k)))
            if k < 3:
                P.append(("stage", k + 1, (j, k + 1)))
            elif more:
                P.append(("stage", 0, (j + 1, 0)))
            Pb()
            Cn.append(("mfma", k & 1, (j, k))); Cb()
        Cn.append(("store", j))
    pp, cp = phases(P), phases(Cn)
    assert len(pp) == len(cp)
    image, regset, done = {0: None, 1: None}, {k: None for k in range(4)}, []
    for po, co in zip(pp, cp):
        reads = {o[1] for o in co if o[0] == "mfma"}
        new_image = dict(image)
        for o in po:
            if o[0] == "fetch":
                _, s, what = o
                assert regset[s] is None or regset[s] == what, ("gather into a full set", n, o, regset)  # == : the last tile's re-read
                regset[s] = what
            else:
                _, s, what = o
                assert regset[s] == what, ("stage", n, o, regset)
                assert (s & 1) not in reads, ("write to an image being read", n, o)
                regset[s] = None
                new_image[s & 1] = what
        for o in co:
            if o[0] == "mfma":
                assert image[o[1]] == o[2], ("consumer reads", n, o, image)
                done.append(o[2])
        image = new_image
    assert done == [(j, k) for j in range(n) for k in range(4)]


if __name__ == "__main__":
    for n in range(1, 12):
        check_obs(n)
    for ksh in (False, True):
        for ntk in range(0, 40):
            check(ntk, ksh)
    print("OK")
