"""Exhaustive check of the tagged-record protocol of tile_kernel_snap (csrc/mspmv_kernels.hpp: rec_announce / rec_store / rec_take) on a
model: every interleaving of the atomic operations of one publisher and one consumer on a two-word slot -- and of the two-level chain
publisher -> group leader -> consumer -- under RELAXED ordering (operations of one thread on DIFFERENT words may take effect in either
order unless one depends on the other's value; operations on the same word keep program order; the exchanges are atomic), for every
poll budget, from clean slots and from slots holding another call's leftovers.

Invariant checked ("record protocol"):
  1. the consumer returns the publisher's payload, or `false` (it then recomputes the sum from the matrix);
  2. when both have finished the slot is (0, 0) -- so a captured launch replays with the same tags on clean slots;
  3. from every reachable state the end is reachable (no deadlock / no wait that nothing can end), and the consumer enters its unbounded
     wait only for a publisher that HAS STARTED (its announcement or its record's word 0 came back from the cancelling exchange);
  4. the same for a leader (takes a predecessor's record, bounded; publishes its group's record) between a publisher and a consumer.
The model is pinned to the source: the three functions' text is hashed, so an edit of the protocol without an edit here fails."""
import hashlib
import os
import re

import pytest

from conftest import ROOT

ZERO, STALE, PENDING, CANCEL, REC0, REC1 = "0", "stale", "pending", "cancel", "rec0", "rec1"
W0, W1 = 0, 1


def _source_functions():
    text = open(os.path.join(ROOT, "merge_spmv_amd", "csrc", "mspmv_kernels.hpp")).read()
    out = []
    for name in ("rec_announce", "rec_store", "rec_take"):
        m = re.search(r"__device__ __forceinline__ [a-z ]+ " + name + r"\(.*?\n}\n", text, flags=re.S)
        assert m, name
        body = re.sub(r"//[^\n]*", "", m.group(0))
        out.append(re.sub(r"\s+", " ", body).strip())
    return out


def test_the_model_is_of_the_code_as_it_stands():
    """The operations modelled below, read off the source: announce = exchange(word 0 <- PENDING); store = [answer was CANCEL: word 0 <- 0,
    word 1 <- 0] else [word 1 <- record, word 0 <- record]; take = polls of {load word 0, load word 1} -> clear both | exchange(word 0 <-
    CANCEL) -> false unless the answer is the record's word 0 or PENDING -> wait for both tags -> clear both -> payload (false when the
    budget is 0).  A change of those functions must come with a change of this model."""
    ann, store, take = _source_functions()
    assert "__hip_atomic_exchange(rec, rec_pending_word(tag_a)" in ann
    assert store.index("announced == rec_cancel_word(tag_a)") < store.index("__hip_atomic_store(rec, 0ull") < store.index("__hip_atomic_store(rec + 1, 0ull")
    assert store.index("__hip_atomic_store(rec + 1, ((unsigned long long) tag_b << 32) | p1") < store.index("__hip_atomic_store(rec, ((unsigned long long) tag_a << 32) | p0")
    assert take.index("polls < max_polls") < take.index("__hip_atomic_exchange(rec, rec_cancel_word(tag_a)") < take.index("return false") < take.index("for (;;)")
    assert "w0 != rec_pending_word(tag_a)) return false" in take and take.rstrip().endswith("return max_polls > 0; }")
    digest = hashlib.sha256("\n".join((ann, store, take)).encode()).hexdigest()[:16]
    assert digest == MODELLED_SOURCE_DIGEST, f"rec_announce / rec_store / rec_take changed (digest {digest}): re-derive the model, then update MODELLED_SOURCE_DIGEST"


MODELLED_SOURCE_DIGEST = "39088bc4add08093"


# ---- threads as small state machines.  A thread state is a tuple (kind, pc, locals...); step(thread, mem) yields every
# (new thread state, new mem) one atomic operation later; a thread whose pc is "done" yields nothing.
# mem = tuple of words, two per slot.

def _set(mem, i, v):
    return mem[:i] + (v,) + mem[i + 1:]


def _pair(first, second):
    """two independent operations of one thread on different words: either order (names of sub-steps)"""
    return ((first, second), (second, first))


def publisher_steps(t, mem, base):
    """rec_announce, then rec_store.  t = ("P", pc, answer, todo) ; todo = the remaining stores of the current unordered pair"""
    _, pc, ans, todo = t
    if pc == "announce":
        ans = mem[base + W0]
        stores = (("w0", ZERO), ("w1", ZERO)) if ans == CANCEL else (("w1", REC1), ("w0", REC0))
        yield ("P", "store", ans, stores), _set(mem, base + W0, PENDING)
    elif pc == "store":
        for k, (word, val) in enumerate(todo):          # (relaxed: the two stores hit different words: either may land first)
            rest = todo[:k] + todo[k + 1:]
            m2 = _set(mem, base + (W0 if word == "w0" else W1), val)
            yield ("P", "store" if rest else "done", ans, rest), m2


def _valid(w0, w1):
    return w0 == REC0 and w1 == REC1


def consumer_steps(t, mem, base, budget):
    """rec_take.  t = ("C", pc, polls_left, l0, l1, todo, result, have_w0); todo = the loads / stores still to do in the current round"""
    _, pc, left, l0, l1, todo, res, have = t
    if pc == "poll":
        if left == 0:
            x = mem[base + W0]                              # the cancelling exchange: word 0 <- CANCEL, the old word comes back
            m2 = _set(mem, base + W0, CANCEL)
            if x != REC0 and x != PENDING:
                yield ("C", "done", 0, None, None, (), False, False), m2     # not started: the publisher will find the cancellation
            else:
                yield ("C", "wait", 0, x if x == REC0 else None, None, (), None, x == REC0), m2
            return
        for order in _pair("w0", "w1"):                     # one poll = two loads of different words, either order
            yield ("C", "poll_loads", left, None, None, order, None, False), mem
    elif pc == "poll_loads":
        word, rest = todo[0], todo[1:]
        if word == "w0": l0 = mem[base + W0]
        else: l1 = mem[base + W1]
        if rest:
            yield ("C", "poll_loads", left, l0, l1, rest, None, False), mem
        elif _valid(l0, l1):
            yield ("C", "clear", left, l0, l1, ("w0", "w1"), True, False), mem
        else:
            yield ("C", "poll", left - 1, None, None, (), None, False), mem
    elif pc == "wait":
        # (unbounded) one round = the loads still needed, either order; with have_w0 the exchanged-out word 0 is kept
        for order in ((("w1",),) if have else _pair("w0", "w1")):
            yield ("C", "wait_loads", 0, l0 if have else None, None, order, None, have), mem
    elif pc == "wait_loads":
        word, rest = todo[0], todo[1:]
        if word == "w0": l0 = mem[base + W0]
        else: l1 = mem[base + W1]
        if rest:
            yield ("C", "wait_loads", 0, l0, l1, rest, None, have), mem
        elif _valid(l0, l1):
            yield ("C", "clear", 0, l0, l1, ("w0", "w1"), budget > 0, have), mem
        else:
            yield ("C", "wait", 0, l0 if have else None, None, (), None, have), mem
    elif pc == "clear":
        for k, word in enumerate(todo):                     # (two stores to different words: either may land first)
            rest = todo[:k] + todo[k + 1:]
            m2 = _set(mem, base + (W0 if word == "w0" else W1), ZERO)
            yield ("C", "clear" if rest else "done", left, l0, l1, rest, res, have), m2


def explore(initial_threads, initial_mem, stepper):
    """All reachable (threads, mem) states; returns (states, edges, terminals)."""
    start = (initial_threads, initial_mem)
    seen = {start}
    stack = [start]
    edges = {}
    terminals = []
    while stack:
        st = stack.pop()
        threads, mem = st
        nxt = []
        for i, t in enumerate(threads):
            for t2, m2 in stepper(i, t, mem, threads):
                nxt.append((threads[:i] + (t2,) + threads[i + 1:], m2))
        edges[st] = nxt
        if not nxt:
            terminals.append(st)
        for n in nxt:
            if n not in seen:
                seen.add(n); stack.append(n)
    return seen, edges, terminals


def can_reach_end(seen, edges, terminals):
    """states from which a terminal state is reachable"""
    rev = {}
    for s, ns in edges.items():
        for n in ns:
            rev.setdefault(n, []).append(s)
    good = set(terminals); stack = list(terminals)
    while stack:
        s = stack.pop()
        for p in rev.get(s, ()):
            if p not in good:
                good.add(p); stack.append(p)
    return good


def _done(t):
    return t[1] == "done"


@pytest.mark.parametrize("budget", [0, 1, 2, 3])
@pytest.mark.parametrize("w0_init,w1_init", [(ZERO, ZERO), (STALE, STALE), (STALE, ZERO), (ZERO, STALE)])
def test_one_publisher_one_consumer_every_interleaving(budget, w0_init, w1_init):
    def stepper(i, t, mem, threads):
        if t[0] == "P":
            return publisher_steps(t, mem, 0)
        return consumer_steps(t, mem, 0, budget)
    threads = (("P", "announce", None, ()), ("C", "poll", budget, None, None, (), None, False))
    seen, edges, terminals = explore(threads, (w0_init, w1_init), stepper)
    assert terminals and len(seen) > 20
    for (p, c), mem in terminals:
        assert _done(p) and _done(c), (p, c, mem)                   # nobody is stuck half way
        assert mem == (ZERO, ZERO), (p, c, mem)                     # 2. the slot is clean when both are through
        assert c[6] in (True, False)                                # 1. payload (True: both words were the publisher's) or recompute
        if budget == 0:
            assert c[6] is False                                    # (the "never look" aid: every such tile recomputes)
    good = can_reach_end(seen, edges, terminals)
    assert good == seen, f"{len(seen - good)} states from which nothing ends the wait"          # 3. no deadlock
    # 3b. the unbounded wait is entered only for a publisher that has started
    for (p, c), mem in seen:
        if c[1] in ("wait", "wait_loads"):
            assert p[1] != "announce", (p, c, mem)
    # ... and the consumer really does take the payload in some interleavings and recompute in others
    results = {c[6] for (p, c), mem in terminals}
    assert results == ({False} if budget == 0 else {True, False})


def test_the_model_finds_the_deadlock_the_invariant_excludes():
    """Negative control, and why "clean when the launch ends" matters: a slot that ALREADY holds this call's PENDING marker (a launch that
    died between announce and store, then a replay with the same tags) lets the consumer's cancel see a publisher that "has started",
    while the real publisher then finds the cancellation and wipes the slot: the consumer waits for ever.  The checker must see it."""
    def stepper(i, t, mem, threads):
        return publisher_steps(t, mem, 0) if t[0] == "P" else consumer_steps(t, mem, 0, 1)
    threads = (("P", "announce", None, ()), ("C", "poll", 1, None, None, (), None, False))
    seen, edges, terminals = explore(threads, (PENDING, ZERO), stepper)
    good = can_reach_end(seen, edges, terminals)
    assert seen - good, "the checker should have found the wait that nothing ends"


@pytest.mark.parametrize("budget", [0, 1, 2])
def test_publisher_leader_consumer_chain(budget):
    """Group records: a LEADER announces its group slot, takes its predecessor's record like any consumer (bounded; false -> it computes
    the predecessor's part from the matrix), then stores its group record (or wipes the slot if the end consumer had cancelled before the
    leader started); the end consumer takes the group slot.  Slot 0 = the predecessor's, slot 1 = the group's.  The leader is the one
    publisher that waits before it publishes -- for publishers that wait for nothing, or not at all: the chain has depth two and no cycle."""
    def stepper(i, t, mem, threads):
        if t[0] == "P":
            return publisher_steps(t, mem, 0)
        if t[0] == "C":
            return consumer_steps(t, mem, 2, budget)
        # leader: ("L", phase, inner) -- announce on slot 1, then a consumer on slot 0, then the store on slot 1
        _, phase, inner = t
        if phase == "announce":
            out = []
            for p2, m2 in publisher_steps(("P", "announce", None, ()), mem, 2):
                out.append((("L", "take", (p2, ("C", "poll", budget, None, None, (), None, False))), m2))
            return out
        if phase == "take":
            pub, cons = inner
            out = []
            for c2, m2 in consumer_steps(cons, mem, 0, budget):
                out.append((("L", "store" if _done(c2) else "take", (pub, c2)), m2))
            return out
        if phase == "store":
            pub, cons = inner
            out = []
            for p2, m2 in publisher_steps(pub, mem, 2):
                out.append((("L", "done" if _done(p2) else "store", (p2, cons)), m2))
            return out
        return []
    threads = (("P", "announce", None, ()), ("L", "announce", None), ("C", "poll", budget, None, None, (), None, False))
    seen, edges, terminals = explore(threads, (ZERO, ZERO, STALE, STALE), stepper)
    assert terminals
    for (p, l, c), mem in terminals:
        assert _done(p) and l[1] == "done" and _done(c), (p, l, c, mem)
        assert mem == (ZERO,) * 4, (p, l, c, mem)
    good = can_reach_end(seen, edges, terminals)
    assert good == seen, f"{len(seen - good)} states from which nothing ends a wait"
    for (p, l, c), mem in seen:
        if c[1] in ("wait", "wait_loads"):
            assert l[1] != "announce"                 # the end consumer waits without bound only for a leader that has started ...
        if l[1] == "take" and l[2][1][1] in ("wait", "wait_loads"):
            assert p[1] != "announce"                 # ... and the leader only for a predecessor that has
