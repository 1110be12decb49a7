"""PreemptSegTree (JobScheduler.h:867-980) twice: node for node (`LiteralTree`, the reference's recursion), and in the
COMPRESSED form the device uses (`CompactTree`, cranesched_amd/csrc/preempt_dev.inc) — test infrastructure: the fuzz test
in tests/test_seg_compact.py runs both on the same operation sequences and compares `satisfied` after every operation.

Why a compressed form exists at all: the reference's tree is split lazily down to quarter-nanosecond ticks, so the walk to
ONE range end visits ~45 levels, on each of which one child carries the end on and the other ("sibling") is a leaf that is
either covered completely or not touched.  All nodes of such a run ("chain") that carry the same set of range ends go through
the same operations in the same order, and so do all their siblings on one side: a chain is kept as ONE record — the head
node (which alone can be covered completely, and alone holds pending tags between operations), the resource every other
chain node holds, and one (resource, satisfied) pair per side for the siblings.  What the reference does that a cleaner tree
would not is reproduced, not repaired: a node covered completely recomputes `satisfied` from its OWN resource; children
inherit the parent's flag and resource when it is split; tags are handed down add first, then sub, as accumulated sums.

Res algebra: cranesched_amd/csrc/res_dev.h (cpu, mem exact; core ids and GRES slots as bit sets: add = or, sub = and-not;
`<=` compares cpu, mem and the GRES slots, not the core ids; a tag is "zero" when all four are)."""


class R:
    __slots__ = ("cpu", "mem", "cores", "gres")

    def __init__(self, cpu=0, mem=0, cores=0, gres=0):
        self.cpu, self.mem, self.cores, self.gres = cpu, mem, cores, gres

    def copy(self):
        return R(self.cpu, self.mem, self.cores, self.gres)

    def add(self, b):
        self.cpu += b.cpu; self.mem += b.mem; self.cores |= b.cores; self.gres |= b.gres

    def sub(self, b):
        self.cpu -= b.cpu; self.mem -= b.mem; self.cores &= ~b.cores; self.gres &= ~b.gres

    def is_zero(self):
        return self.cpu == 0 and self.mem == 0 and self.cores == 0 and self.gres == 0

    def key(self):
        return (self.cpu, self.mem, self.cores, self.gres)


def le(a, b):
    return a.cpu <= b.cpu and a.mem <= b.mem and (a.gres & ~b.gres) == 0


# ---------------------------------------------------------------------------------------------------------------------
class LiteralTree:
    """The reference's tree, node for node (tests/select_pyref.py SegTree on this file's Res)."""

    def __init__(self, st, ed, target):
        self.target = target
        self.root = self._node(st, ed, False, R())
        self.visits = 0

    @staticmethod
    def _node(st, ed, sat, res):
        return {"st": st, "ed": ed, "ls": None, "rs": None, "sat": sat, "res": res, "add": R(), "sub": R()}

    def _apply(self, n, r, plus):
        (n["res"].add if plus else n["res"].sub)(r)
        n["sat"] = le(self.target, n["res"])
        if n["ls"] is not None:
            n["add" if plus else "sub"].add(r)

    def _down(self, n):
        if n["ls"] is None:
            mid = n["st"] + (n["ed"] - n["st"]) // 2
            n["ls"] = self._node(n["st"], mid, n["sat"], n["res"].copy())
            n["rs"] = self._node(mid, n["ed"], n["sat"], n["res"].copy())
            return
        if not n["add"].is_zero():
            self._apply(n["ls"], n["add"], True); self._apply(n["rs"], n["add"], True)
            n["add"] = R()
        if not n["sub"].is_zero():
            self._apply(n["ls"], n["sub"], False); self._apply(n["rs"], n["sub"], False)
            n["sub"] = R()

    def _walk(self, n, st, ed, r, plus):
        self.visits += 1
        if n["ed"] <= st or ed <= n["st"]:
            return
        if st <= n["st"] and n["ed"] <= ed:
            self._apply(n, r, plus)
            return
        self._down(n)
        self._walk(n["ls"], st, ed, r, plus)
        self._walk(n["rs"], st, ed, r, plus)
        n["sat"] = n["ls"]["sat"] and n["rs"]["sat"]

    def op(self, st, ed, r, plus):
        self._walk(self.root, st, ed, r, plus)

    @property
    def satisfied(self):
        return self.root["sat"]


# ---------------------------------------------------------------------------------------------------------------------
class Leaf:
    __slots__ = ("res", "sat")

    def __init__(self, res, sat):
        self.res, self.sat = res, sat


class Chain:
    """Nodes n_0 (the head: [st, ed)) .. n_m, all with the same range ends strictly inside — every one of them within
    [xmin, xmax] —, n_m the first whose midpoint does not leave them all on one side.  Their other children are leaves:
    `nl` of them to the left of xmin, `nr` to the right of xmax (the children of n_m that hold no end are counted in)."""
    __slots__ = ("st", "ed", "xmin", "xmax", "res", "sat", "ta", "ts", "rc", "rl", "sl", "nl", "rr", "sr", "nr", "cl", "cr")


def _levels(st, ed, xmin, xmax):
    """The chain's nodes from the head down: (p, q, mid, side) with side -1 / +1 = the ends go on in the left / right child,
    0 = the bottom node."""
    p, q = st, ed
    while True:
        mid = p + (q - p) // 2
        if xmax < mid:
            yield p, q, mid, -1
            q = mid
        elif xmin > mid:
            yield p, q, mid, +1
            p = mid
        else:
            yield p, q, mid, 0
            return


class CompactTree:
    def __init__(self, st, ed, target):
        self.target = target
        self.st, self.ed = st, ed
        self.root = Leaf(R(), False)
        self.visits = 0

    # ---- a node (leaf or chain head) covered completely: add_res_ / sub_res_ ----------------------------------------
    def _apply(self, n, r, plus):
        (n.res.add if plus else n.res.sub)(r)
        n.sat = le(self.target, n.res)
        if isinstance(n, Chain):
            (n.ta if plus else n.ts).add(r)

    def _chain_of_leaf(self, leaf, p, q, ends):
        """The leaf [p, q) is split for the first time, by an operation with `ends` strictly inside it."""
        c = Chain()
        c.st, c.ed, c.xmin, c.xmax = p, q, min(ends), max(ends)
        c.res, c.sat, c.ta, c.ts = leaf.res, leaf.sat, R(), R()
        c.rc, c.rl, c.rr = leaf.res.copy(), leaf.res.copy(), leaf.res.copy()   # children inherit the parent's resource and flag
        c.sl = c.sr = leaf.sat
        c.nl = c.nr = 0
        c.cl = c.cr = None
        for lp, lq, mid, side in _levels(p, q, c.xmin, c.xmax):
            if side < 0:
                c.nr += 1
            elif side > 0:
                c.nl += 1
            else:   # the bottom node's children: a chain to be where an end lies strictly inside, else one more leaf of that side
                if any(lp < e < mid for e in ends):
                    c.cl = Leaf(leaf.res.copy(), leaf.sat)
                else:
                    c.nl += 1
                if any(mid < e < lq for e in ends):
                    c.cr = Leaf(leaf.res.copy(), leaf.sat)
                else:
                    c.nr += 1
        return c

    def _and_below(self, c):
        return (c.nl == 0 or c.sl) and (c.nr == 0 or c.sr) and (c.cl is None or c.cl.sat) and (c.cr is None or c.cr.sat)

    def _insert_end(self, c, e):
        """A range end e strictly inside the head of chain c that is not one of the chain's ends yet: the chain is cut where
        e leaves the others (or sits on a midpoint), no node changes its state by that."""
        if c.xmin <= e <= c.xmax and (e == c.xmin or e == c.xmax):
            return
        nl = nr = 0
        for p, q, mid, side in _levels(c.st, c.ed, c.xmin, c.xmax):
            if side == 0:
                # e goes with the others down to the bottom node: into one of its children (or onto its midpoint)
                if e < mid and c.cl is None:
                    c.cl = Leaf(c.rl.copy(), c.sl); c.nl -= 1      # that child was a leaf of the left group: now on its own
                elif e > mid and c.cr is None:
                    c.cr = Leaf(c.rr.copy(), c.sr); c.nr -= 1
                break
            inside_same = (e < mid) if side < 0 else (e > mid)
            if inside_same:
                nl += side > 0
                nr += side < 0
                continue
            # e == mid, or e strictly inside the sibling of this level: this node becomes the bottom of the upper part
            low = Chain()
            low.st, low.ed = (p, mid) if side < 0 else (mid, q)
            low.xmin, low.xmax = c.xmin, c.xmax
            low.res, low.ta, low.ts = c.rc.copy(), R(), R()
            low.rc, low.rl, low.rr, low.sl, low.sr = c.rc.copy(), c.rl.copy(), c.rr.copy(), c.sl, c.sr
            low.nl = c.nl - nl - (1 if side > 0 else 0)
            low.nr = c.nr - nr - (1 if side < 0 else 0)
            low.cl, low.cr = c.cl, c.cr
            low.sat = self._and_below(low)
            sib_in = e != mid
            if side < 0:     # the others go left: the sibling is the right child
                c.cl = low
                c.cr = Leaf(c.rr.copy(), c.sr) if sib_in else None
                c.nl, c.nr = nl, nr + (0 if sib_in else 1)
            else:
                c.cr = low
                c.cl = Leaf(c.rl.copy(), c.sl) if sib_in else None
                c.nl, c.nr = nl + (0 if sib_in else 1), nr
            break
        c.xmin, c.xmax = min(c.xmin, e), max(c.xmax, e)

    def _walk(self, n, p, q, a, b, r, plus):
        """-> the node that stands for [p, q) afterwards."""
        self.visits += 1
        if q <= a or b <= p:
            return n
        if a <= p and q <= b:
            self._apply(n, r, plus)
            return n
        ends = [e for e in (a, b) if p < e < q]
        if isinstance(n, Leaf):
            n = self._chain_of_leaf(n, p, q, ends)
        else:
            for e in ends:
                self._insert_end(n, e)
        c = n
        # push_down_ along the chain: the head's tags reach every node below it, add first, then sub
        for tag, plus_t in ((c.ta, True), (c.ts, False)):
            if tag.is_zero():
                continue
            (c.rc.add if plus_t else c.rc.sub)(tag)
            (c.rl.add if plus_t else c.rl.sub)(tag); c.sl = le(self.target, c.rl)
            (c.rr.add if plus_t else c.rr.sub)(tag); c.sr = le(self.target, c.rr)
            for ch in (c.cl, c.cr):
                if ch is not None:
                    self._apply(ch, tag, plus_t)
        c.ta, c.ts = R(), R()
        # the operation itself: the siblings of a side are covered all or none
        if a <= p:
            (c.rl.add if plus else c.rl.sub)(r); c.sl = le(self.target, c.rl)
        if b >= q:
            (c.rr.add if plus else c.rr.sub)(r); c.sr = le(self.target, c.rr)
        # ... and the children of the bottom node
        bp, bq, bmid = [(lp, lq, mid) for lp, lq, mid, side in _levels(c.st, c.ed, c.xmin, c.xmax) if side == 0][0]
        if c.cl is not None:
            c.cl = self._walk(c.cl, bp, bmid, a, b, r, plus)
        if c.cr is not None:
            c.cr = self._walk(c.cr, bmid, bq, a, b, r, plus)
        c.sat = self._and_below(c)     # push_up_, level by level: every chain node = its sibling and what is below
        return c

    def op(self, a, b, r, plus):
        assert a < b
        self.root = self._walk(self.root, self.st, self.ed, a, b, r, plus)

    @property
    def satisfied(self):
        return self.root.sat
