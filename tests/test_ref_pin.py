"""PIN OF THE ORACLE: the restated oracle (oracle/*.hpp) against THE REFERENCE'S OWN CODE.

oracle/_ref/libcrane_ref.so is built from slices of /root/reference's sources (JobScheduler.{h,cpp}: NodeState,
NodeSelector, LocalScheduler::*, EarliestStartSubsetSelector, PreemptSegTree, SchedulerAlgo::NodeSelect,
MultiFactorPriority; PublicHeader.{h,cpp}: the whole resource algebra) that oracle/ref_build/extract.py cuts out at
build time, compiled against stand-ins for abseil / fpm / the CraneCtld singletons (oracle/ref_build/shim/), and driven
through the same C entry points as the oracle (oracle/ref_build/ref_harness.cpp).  Nothing of the reference is
committed; on a box without /root/reference the prebuilt .so is used, and without either these tests are skipped.

Every comparison is exact: reasons, start times, per-node allocations (cpu, memory, core ids, GRES slots), task
counts, EVERY final fp64 node cost as a bit pattern, EVERY node's final time map, preempted lists, cancel list,
preempting set.

Two flavours: `ref` = canonical (hash-map iteration of GRES types pinned to ascending type, unstable sorts made stable:
two macro renames, no slice line edited — the order SURVEY.md §7 fixes, used by the oracle and the engine), `ref_hash` =
libstdc++'s real unordered_map / introsort.  They must agree wherever the unspecified orders cannot matter.
"""
import numpy as np
import pytest

from cranesched_amd import abi, synth
from oracle import pyoracle
from tests import helpers, kat, kat_preempt

pytestmark = pytest.mark.skipif(not pyoracle.ref_available(), reason="oracle/_ref is not built and /root/reference is absent")


def same_run(tag, c, a, b, preempt=False, jobs=None):
    d = b.placements.diff(a.placements)
    assert d is None, f"{tag}: placements of the reference build differ from the oracle at {d}"
    ca, cb = a.costs().view(np.uint64), b.costs().view(np.uint64)
    ne = np.nonzero(ca != cb)[0]
    assert len(ne) == 0, f"{tag}: {len(ne)} fp64 costs differ, first at part-slot {ne[0]}: {a.costs()[ne[0]]!r} (oracle) vs {b.costs()[ne[0]]!r} (reference)"
    # the reference builds a NodeState only for the nodes of partitions that have pending jobs (JobScheduler.cpp:6571-6573);
    # the oracle's debug getter also reports the untouched map of every other schedulable node
    live = np.zeros(c.num_nodes, bool)
    if jobs is not None:
        outside = jobs.reservation is None
        for p in range(c.num_partitions):
            sel = jobs.partition == p
            if jobs.reservation is not None:
                sel = sel & (jobs.reservation == abi.RESV_NONE)
            if sel.any():
                live[np.asarray(c.part_nodes[c.part_offsets[p]:c.part_offsets[p + 1]], np.int64)] = True
    for n in range(c.num_nodes):
        x, y = a.timeline(n), b.timeline(n)
        if jobs is not None and not live[n]:
            assert len(y["t"]) == 0, f"{tag}: the reference has a NodeState for node {n}, which no partition with pending jobs lists"
            continue
        for f in ("t", "cpu_raw", "mem", "core_lo", "core_hi", "gres", "core_w2", "core_w3"):
            assert np.array_equal(x[f], y[f]), f"{tag}: final time map of node {n} differs in {f}:\n{x[f]}\n{y[f]}"
    if preempt:
        assert a.preempt_out.lists() == b.preempt_out.lists(), f"{tag}: preempted_jobs lists differ"
        assert a.preempt_out.cancelled_ids() == b.preempt_out.cancelled_ids(), f"{tag}: EnqueuePreemptCancel differs"
        assert a.preempt_out.preempting_ids() == b.preempt_out.preempting_ids(), f"{tag}: m_preempting_set_ differs"


def both(tag, c, j, now, backend="ref", **kw):
    a = pyoracle.select(c, j, now, **kw)
    b = pyoracle.select(c, j, now, backend=backend, **kw)
    same_run(tag, c, a, b, preempt=kw.get("preempt") is not None, jobs=j)
    return a, b


# ---------------------------------------------------------------------------------------------------------------------
# hand-derived scenarios: the reference's own code must give the hand-derived answers
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("backend", ["ref", "ref_hash"])
@pytest.mark.parametrize("scn", kat.scenarios(), ids=lambda s: s[0])
def test_reference_code_on_hand_derived_scenarios(scn, backend):
    name, c, j, cfg, expect = scn
    if cfg:
        with pytest.raises(pyoracle.RefUnsupported):   # kAlgoMaxJobNumPerNode is a compile-time constant there
            pyoracle.select(c, j, kat.NOW, backend=backend, **cfg)
        return
    r = pyoracle.select(c, j, kat.NOW, backend=backend)
    kat.check(name, c, j, r.placements, expect, costs=r.costs(), timeline=r.timeline)


@pytest.mark.parametrize("scn", kat_preempt.scenarios(), ids=lambda s: s[0])
def test_reference_code_on_hand_derived_preempt_scenarios(scn):
    from tests.test_preempt import check
    name, c, j, r, pre, expect = scn
    run = pyoracle.select(c, j, kat_preempt.NOW, running=r, preempt=pre, backend="ref")
    check(run, j, c, expect, name)
    both(name, c, j, kat_preempt.NOW, running=r, preempt=pre)


def test_reference_code_on_hand_derived_reservation_scenarios():
    from tests.test_reservations import NOW, SCENARIOS, _check
    for name in sorted(SCENARIOS):
        c, j, rn, rv, exp = SCENARIOS[name]()
        r = pyoracle.select(c, j, NOW, running=rn, reservations=rv, backend="ref")
        _check(r.placements, exp)
        both(name, c, j, NOW, running=rn, reservations=rv)


# ---------------------------------------------------------------------------------------------------------------------
# random clusters: every feature of the slice
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("seed", range(200))
def test_selection_random_small(seed):
    c, j, now, run = helpers.random_case(300 + seed, N=20 + seed % 13, J=160, P=1 + seed % 3, running=8 + seed % 9)
    both(f"random {seed}", c, j, now, running=run)


@pytest.mark.parametrize("seed", range(40))
def test_selection_random_on_nodes_with_192_and_256_cores(seed):
    """Core ids 128..255 (ABI 3: the core_w2 / core_w3 planes): CpuSet::core_ids is an unbounded std::set<uint32_t>
    (PublicHeader.h:555-573); jobs of up to 16 cpus x several tasks fill the wide nodes far beyond core id 127."""
    c, j, now, run = helpers.random_case(500 + seed, N=20 + seed % 13, J=400, P=1 + seed % 3, running=8 + seed % 9)
    c = helpers.widen_cores(c, seed)
    a, b = both(f"wide cores {seed}", c, j, now, running=run)
    assert (a.placements.core_w2 != 0).any(), "case must allocate core ids above 127"
    if seed % 4 == 0:
        l = pyoracle.select(c, j, now, running=run, algebra=pyoracle.LITERAL)
        assert l.placements.diff(a.placements) is None


@pytest.mark.parametrize("seed", range(12))
def test_selection_random_larger(seed):
    c, j, now, run = helpers.random_case(7000 + seed, N=96 + 16 * seed, J=1500, P=1 + seed % 4, running=60 + 10 * seed)
    a, b = both(f"larger {seed}", c, j, now, running=run)
    r = a.placements.reason[:j.num_jobs]
    assert (r == abi.REASON_NONE).sum() > 100 and (r == abi.REASON_PRIORITY).sum() > 50   # starts now AND backfills


@pytest.mark.parametrize("seed,lay", [(s, lay) for s in (11, 12, 13, 14) for lay in ("all+subsets", "chain", "random")])
def test_partitions_that_share_nodes(seed, lay):
    from tests.test_overlap import overlap_case
    c, j, now, run = overlap_case(seed, N=32, J=200, layout=lay)
    both(f"overlap {seed} {lay}", c, j, now, running=run)


@pytest.mark.parametrize("seed", range(12))
def test_reservations_random(seed):
    from tests.test_reservations import random_resv_case
    c, j, now, run, rv = random_resv_case(seed, N=24 + 4 * seed, J=300)
    try:
        both(f"resv {seed}", c, j, now, running=run, reservations=rv)
    except pyoracle.RefAsserted as e:
        # node lists inside reservations can reach the reference's own CRANE_ASSERT_MSG
        # (JobScheduler.cpp:6313-6317): the oracle keeps that assertion, so it must fail on the same input
        with pytest.raises(Exception):
            pyoracle.select(c, j, now, running=run, reservations=rv)
        pytest.skip(f"the reference itself asserts on this input: {e}")


@pytest.mark.parametrize("seed", range(60))
def test_preemption_random(seed):
    from tests.test_preempt import random_preempt_case
    c, j, now, run, pre = random_preempt_case(500 + seed, N=6 + seed % 7, J=50 + seed % 40, P=1 + seed % 2, running=10 + seed % 11)
    both(f"preempt {seed}", c, j, now, running=run, preempt=pre)


@pytest.mark.parametrize("seed", range(6))
def test_preemption_larger(seed):
    from tests.test_preempt import random_preempt_case
    c, j, now, run, pre = random_preempt_case(900 + seed, N=48 + 8 * seed, J=500, P=2, running=120)
    a, _ = both(f"preempt larger {seed}", c, j, now, running=run, preempt=pre)
    assert sum(len(x) for x in a.preempt_out.lists()) > 0


@pytest.mark.parametrize("seed", range(8))
def test_preemption_with_reservations(seed):
    from tests.test_preempt import resv_preempt_case
    c, j, now, run, rv, pre = resv_preempt_case(seed)
    both(f"resv preempt {seed}", c, j, now, running=run, reservations=rv, preempt=pre)


@pytest.mark.parametrize("seed,lay", [(s, lay) for s in range(6) for lay in ("all+subsets", "chain", "random")])
def test_preemption_with_shared_nodes(seed, lay):
    from tests.test_preempt import overlap_preempt_case
    c, j, now, run, pre = overlap_preempt_case(seed, layout=lay)
    both(f"overlap preempt {seed} {lay}", c, j, now, running=run, preempt=pre)


def test_batch_limit_and_skips():
    c, j, now, run = helpers.random_case(4711, N=40, J=500, P=2, running=12)
    both("batch", c, j, now, running=run, scheduled_batch_size=217)


# ---------------------------------------------------------------------------------------------------------------------
# the synthetic benchmark configurations (scaled) and the contended full-run cases the tile digests use
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("cfg", ["C1", "C2", "C3", "C4", "C5"])
def test_benchmark_configs_scaled(cfg):
    J, N, P = {"C1": (1000, 128, 1), "C2": (5000, 256, 1), "C3": (4000, 96, 1), "C4": (6000, 512, 8), "C5": (6000, 512, 8)}[cfg]
    c, j, now = synth.make_config(cfg, J=J, N=N, P=P)
    a, _ = both(cfg, c, j, now)
    assert (a.placements.reason[:j.num_jobs] == abi.REASON_NONE).sum() > 0


# ---------------------------------------------------------------------------------------------------------------------
# the resource algebra by itself: GetFeasibleResourceInNode, Ckmin, +=, -=, <= of the reference vs both oracle algebras
# ---------------------------------------------------------------------------------------------------------------------
def _rand_res(rng, lay, max_cores=128):
    cores = int(rng.integers(0, max_cores + 1))
    m = (1 << cores) - 1
    keep = 0
    for sh in range(0, 256, 30):
        keep |= int(rng.integers(0, 1 << 30)) << sh
    m &= keep if rng.random() < 0.7 else m
    g = int(rng.integers(0, 1 << 16)) if rng.random() < 0.8 else 0
    cpu = int(bin(m).count("1")) * 256 + (128 if rng.random() < 0.2 else 0)
    w = 2**64 - 1
    return pyoracle.make_res(cpu, int(rng.integers(0, 64)) << 30, m & w, (m >> 64) & w, g, (m >> 128) & w, (m >> 192) & w)


@pytest.mark.parametrize("seed,max_cores", [(0, 128), (1, 128), (2, 128), (3, 128), (4, 256), (5, 256), (6, 192)])
def test_algebra_random_against_the_reference_code(seed, max_cores):
    rng = np.random.default_rng(99 + seed)
    lay = helpers.multi_type_layout()
    for _ in range(1500):
        a, b = _rand_res(rng, lay, max_cores), _rand_res(rng, lay, max_cores)
        for op in ("ckmin", "add", "le"):
            want = pyoracle.binop(lay, 0, op, a, b, backend="ref")
            for alg in (pyoracle.MASK, pyoracle.LITERAL):
                assert pyoracle.binop(lay, alg, op, a, b) == want, (op, a.tup(), b.tup())
        # -= of a subset (what every caller on the path does): b := a AND b
        sub = pyoracle.make_res(min(a.cpu, b.cpu), min(a.mem, b.mem), a.clo & b.clo, a.chi & b.chi, a.gres & b.gres, a.c2 & b.c2, a.c3 & b.c3)
        want = pyoracle.binop(lay, 0, "sub", a, sub, backend="ref")
        for alg in (pyoracle.MASK, pyoracle.LITERAL):
            assert pyoracle.binop(lay, alg, "sub", a, sub) == want
        # requests: whole / fractional cpus, untyped / typed / mixed GRES
        s = int(rng.integers(0, 6))
        gtot, gspec = [0, 0, 0, 0], [0] * 8
        if s == 1: gtot[0] = int(rng.integers(1, 6))
        elif s == 2: v = int(rng.integers(1, 4)); gtot[0] = v; gspec[0] = v
        elif s == 3: v = int(rng.integers(1, 4)); gtot[0] = v + 1; gspec[1] = v
        elif s == 4: gtot[1] = int(rng.integers(1, 9))
        elif s == 5: gspec[0] = 1; gspec[1] = 1; gtot[0] = int(rng.integers(2, 5))
        req = pyoracle.make_req(int(rng.choice([128, 256, 384, 512, 1024, 4096] + ([130 * 256, 150 * 256, 200 * 256] if max_cores > 128 else []))),
                                int(rng.integers(0, 32)) << 30, gtot, gspec)
        want = pyoracle.feasible(lay, 0, req, a, backend="ref")
        for alg in (pyoracle.MASK, pyoracle.LITERAL):
            assert pyoracle.feasible(lay, alg, req, a) == want, (req.cpu, gtot, gspec, a.tup())


# ---------------------------------------------------------------------------------------------------------------------
# MultiFactorPriority (JobScheduler.cpp:7606-7819): fp64 priorities as bit patterns + the order
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("seed", range(40))
def test_multifactor_priority_against_the_reference_code(seed):
    from cranesched_amd.priority import PriorityConfig, synth_priority_case
    rng = np.random.default_rng(seed)
    J, R, A = int(rng.integers(1, 400)), int(rng.integers(0, 120)), int(rng.integers(1, 12))
    pd, rn, now = synth_priority_case(J, R, A, seed=seed, cached_frac=0.15 if seed % 3 == 0 else 0.0)
    if seed % 5 == 0:       # degenerate bounds: one value per attribute
        pd.node_num[:] = 2; pd.total_mem[:] = 4 << 30
        if R:
            rn.node_num[:] = 2; rn.alloc_mem[:] = 4 << 30
    if seed % 7 == 0:       # "submitted in the future": the unsigned age wraps and is capped at MaxAge (:7664-7665)
        pd.submit_sec[0] = now + 50
    cfg = PriorityConfig(max_age_sec=int(rng.integers(100, 100000)), weight_age=int(rng.integers(0, 2000)),
                         weight_fair_share=int(rng.integers(0, 2000)), weight_job_size=int(rng.integers(0, 2000)),
                         weight_partition=int(rng.integers(0, 2000)), weight_qos=int(rng.integers(0, 2000)),
                         favor_small=bool(seed % 2))
    num_accounts, rn = A, (rn if R else None)
    o_a, p_a = pyoracle.priority_order(now, cfg, num_accounts, pd, rn)
    o_b, p_b = pyoracle.priority_order(now, cfg, num_accounts, pd, rn, backend="ref")
    assert np.array_equal(p_a.view(np.uint64), p_b.view(np.uint64)), "priorities differ as bit patterns"
    assert np.array_equal(o_a, o_b), "order differs"


# ---------------------------------------------------------------------------------------------------------------------
# the two flavours agree where hash order / unstable sorting cannot matter (one type per GRES name per node)
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("cfg", ["C3", "C4"])
def test_canonical_renames_change_nothing_else(cfg):
    c, j, now = synth.make_config(cfg, J=5000, N=256, P=2)
    a = pyoracle.select(c, j, now, backend="ref")
    b = pyoracle.select(c, j, now, backend="ref_hash")
    same_run(f"{cfg} ref vs ref_hash", c, a, b)


# ---------------------------------------------------------------------------------------------------------------------
# a whole contended queue: the reference's own code reproduces the COMMITTED full-run digest that the GPU engine is held
# to (tests/test_gpu_fullrun.py) — tile1 here (6 000 jobs x 380 nodes, 24 % backfilled, 6 s); tile3 / tile10 and whole
# C4 / C2 partitions: tools/ref_fullsize.py -> profiles/r03_ref_vs_oracle_fullsize.txt
# ---------------------------------------------------------------------------------------------------------------------
def test_reference_code_reproduces_the_committed_fullrun_digest():
    import os
    from tests import fullrun
    from tests.golden.make_fullrun import CASES
    name, J, N, P = CASES["tile1"]
    c, j, now = synth.make_config(name, J=J, N=N, P=P)
    r = pyoracle.select(c, j, now, backend="ref")
    d = fullrun.digest(r.placements, r.costs().view(np.uint64), r.timeline, c.num_nodes)
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "fullrun_tile1.npz"))
    assert fullrun.compare(d, g) is None


# ---------------------------------------------------------------------------------------------------------------------
# the input class on which the REFERENCE ITSELF asserts (CRANE_ASSERT_MSG, JobScheduler.cpp:6313-6317):
# get_max_tasks counts 0.5-cpu tasks one by one (no core ids asked), the distribution composes them into one whole-cpu
# request that needs core ids the node's leftover set does not have
# ---------------------------------------------------------------------------------------------------------------------
def _resv_case_with_node_lists(seed):
    import tests.test_reservations as tr
    orig = helpers.random_case
    try:
        helpers.random_case = lambda s, **kw: orig(s, **{**kw, "lists": True})
        return tr.random_resv_case(seed, N=24, J=300)
    finally:
        helpers.random_case = orig


@pytest.mark.parametrize("seed", [1005, 1102])
def test_the_reference_asserts_on_its_own_on_this_input_class(seed):
    import subprocess
    import sys
    c, j, now, run, rv = _resv_case_with_node_lists(seed)
    with pytest.raises(pyoracle.RefAsserted, match="CRANE_ASSERT_MSG"):
        pyoracle.select(c, j, now, running=run, reservations=rv, backend="ref")
    # the restatement keeps the assertion (assert() -> abort): run it in a child process
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from tests.test_ref_pin import _resv_case_with_node_lists\n"
            "from oracle import pyoracle\n"
            "c, j, now, run, rv = _resv_case_with_node_lists(%d)\n"
            "pyoracle.select(c, j, now, running=run, reservations=rv)\n") % (str(__import__('pathlib').Path(__file__).resolve().parents[1]), seed)
    p = subprocess.run([sys.executable, "-c", code], capture_output=True)
    assert p.returncode != 0 and b"ssert" in p.stderr, "the oracle must fail on the input the reference asserts on"


def test_reference_code_on_the_core_id_shortfall_case():
    from tests.test_reservations import NOW, core_id_shortfall_case
    c, j, rn, rv = core_id_shortfall_case()
    a, b = both("core id shortfall", c, j, NOW, running=rn, reservations=rv)
    assert b.placements.reason[0] == 0 and b.placements.start_sec[0] == NOW and b.placements.core_lo[0] == 0b111


# ---------------------------------------------------------------------------------------------------------------------
# The license pre-pass of NodeSelect (JobScheduler.cpp:6739): LicenseManager::CheckLicenseCountSufficient
# (LicenseManager.cpp:167-221) — the reference's own compiled function against the host pass of the product's adapter
# (cranesched_amd/host/NodeSelectionAlgo.cpp: GpuNodeSelectionAlgo::CheckLicenseCountSufficient), same tables, same requests
# ---------------------------------------------------------------------------------------------------------------------
def _license_case(seed):
    import random
    rng = random.Random(seed)
    L = rng.choice([1, 2, 3, 6])
    big = rng.random() < 0.15      # counts near 2^32: the reference adds in uint32
    total = [rng.choice([0, 1, 4, 10, 100, 2 ** 32 - 1 if big else 50]) for _ in range(L)]
    used = [rng.randrange(0, t + 1) if rng.random() < 0.5 else 0 for t in total]
    reserved = [rng.choice([0, 0, 1, 3]) for _ in range(L)]
    deficit = [rng.choice([0, 0, 0, 2, 2 ** 31 if big else 5]) for _ in range(L)]
    J = rng.randrange(1, 120)
    reqs, is_or = [], []
    for _ in range(J):
        n = rng.choice([0, 1, 1, 2, 3])
        reqs.append([(rng.randrange(0, L + (1 if rng.random() < 0.2 else 0)), rng.choice([1, 1, 2, 5, 2 ** 32 - 2 if big else 7])) for _ in range(n)])
        is_or.append(rng.random() < 0.4)
    return total, used, reserved, deficit, reqs, is_or


@pytest.mark.parametrize("seed", range(60))
def test_license_pre_pass_of_the_adapter_against_the_reference_code(seed, tmp_path):
    import os
    import subprocess
    import __graft_entry__ as g
    g.build_host()
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cranesched_amd", "host", "test_host_adapter")
    total, used, reserved, deficit, reqs, is_or = _license_case(seed)
    rej, actual = pyoracle.license_check(total, used, reserved, deficit, reqs, is_or)
    path = tmp_path / "case.txt"
    with open(path, "w") as f:
        f.write(f"{len(total)} {len(reqs)}\n")
        for l in range(len(total)):
            f.write(f"{total[l]} {used[l]} {reserved[l]} {deficit[l]}\n")
        for r, o in zip(reqs, is_or):
            f.write(f"{int(o)} {len(r)} " + " ".join(f"{a} {b}" for a, b in r) + "\n")
    out = subprocess.run([exe, "--license-file", str(path)], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = out.stdout.strip().splitlines()
    assert len(lines) == len(reqs)
    for j, ln in enumerate(lines):
        v = [int(x) for x in ln.split()]
        got = [(v[2 + 2 * i], v[3 + 2 * i]) for i in range(v[1])]
        assert bool(v[0]) == bool(rej[j]) and got == actual[j], (seed, j, reqs[j], is_or[j], ln, rej[j], actual[j])
    # the pass does something in these cases: some job is rejected, or some license is granted
    assert any(rej) or any(actual)


def test_license_pre_pass_known_answers_of_the_reference_code():
    """hand-derived from LicenseManager.cpp:183-219: AND takes all or nothing, OR the first alternative that fits, `used` accumulates
    over the ordered jobs, `reserved` and `last_deficit` count against the total, an unknown license fails AND and is skipped by OR"""
    rej, act = pyoracle.license_check([4, 1], [1, 0], [1, 0], [0, 0],
                                      [[(0, 2)], [(0, 1)], [(0, 1), (1, 1)], [(0, 1), (1, 1)], [(5, 1), (1, 1)], [(5, 1), (1, 1)], []],
                                      [False, False, True, False, True, False, False])
    # job 0: 2 + reserved 1 + used 1 = 4 <= 4; job 1: 1 + 1 + 3 > 4; job 2 (OR): lic0 full, lic1 fits; job 3 (AND): lic0 full; job 4 (OR): the
    # unknown license is skipped, lic1 is taken by now; job 5 (AND): an unknown license fails it; job 6: no request, not looked at
    assert list(rej) == [False, True, False, True, True, True, False]
    assert act == [[(0, 2)], [], [(1, 1)], [], [], [], []]
    rej, act = pyoracle.license_check([4, 1], [0, 0], [0, 0], [0, 0], [[(0, 4)], [(0, 1), (1, 1)], [(1, 1)]], [False, True, False])
    assert list(rej) == [False, False, True] and act == [[(0, 4)], [(1, 1)], []]
