"""-m gpu: the rank kernels (through the C ABI) against the oracle and the vectors the
reference produced.  Integer / index work -> bit-exact."""
import os

import numpy as np
import pytest
import torch

from oracle import rank_step as rs
from util import GOLDEN

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


def _run_rank(dev, score, pri, idle, runs, starv, period, tiebreak=None, ascending=False, use_pri=None):
    from vllm_ltr_amd.rank import RankWorkspace, rank_step
    ws = RankWorkspace(dev)
    t = lambda a, dt: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(dev, dt)
    s = t(score, torch.float32)
    p, i, r = t(pri, torch.int32), t(idle, torch.int32), t(runs, torch.int32)
    tb = t(tiebreak, torch.int32)
    perm = rank_step(s, p, i, r, starv, period, ws, tiebreak=tb, ascending=ascending, use_pri=use_pri)
    torch.cuda.synchronize()
    back = lambda x: None if x is None else x.cpu().numpy()
    return perm.cpu().numpy(), back(p), back(i), back(r)


def test_golden_order_cases(dev):
    z = np.load(os.path.join(GOLDEN, "rank_order.npz"))
    for ci in range(int(z["n_cases"])):
        g = lambda k: z[f"c{ci}_{k}"]
        score, where = g("score"), g("where")
        starv, period = int(g("starv")), int(g("period"))
        concat = np.concatenate([np.nonzero(where == q)[0] for q in (0, 1, 2)])
        perm, p, i, r = _run_rank(dev, score[concat], g("pri0")[concat], g("idle0")[concat], g("runs0")[concat],
                                  starv, period)
        assert concat[perm].tolist() == g("order").tolist(), f"case {ci}"
        assert (np.stack([p, i, r], 1) == g("post")[concat]).all(), f"case {ci} counters"
        if starv == -1:
            ids = [str(k) for k in concat]
            tb = rs.string_rank(ids)
            perm, *_ = _run_rank(dev, score[concat], None, None, None, -1, 0, tiebreak=tb)
            assert concat[perm].tolist() == g("tpt").tolist()
            perm, *_ = _run_rank(dev, score[concat], None, None, None, -1, 0, tiebreak=tb, ascending=True)
            assert concat[perm].tolist() == g("rtpt").tolist()
            perm, *_ = _run_rank(dev, score[concat], None, None, None, -1, 0, ascending=True)
            assert concat[perm].tolist() == g("ropt").tolist()


@pytest.mark.parametrize("n,starv,period", [(1, -1, 0), (2, 0, 1), (63, 3, 2), (64, -1, 0), (65, 5, 1),
                                            (1000, 200, 10), (1025, 2, 2), (8192, 200, 10), (8192, -1, 0),
                                            (20000, 7, 3), (65536, 200, 10), (100003, -1, 0)])
def test_rank_step_vs_oracle(dev, n, starv, period):
    r = np.random.RandomState(n + 7 * (starv + 1))
    score = r.standard_normal(n).astype(np.float32).astype(np.float16).astype(np.float32)   # many ties
    score[r.randint(0, n, max(1, n // 20))] = 0.0
    score[r.randint(0, n, max(1, n // 20))] = -0.0
    pri = -(r.rand(n) < 0.2).astype(np.int32)
    idle = r.randint(0, max(2, 2 * max(starv, 1)), n).astype(np.int32)
    runs = r.randint(-2, period + 2, n).astype(np.int32)
    perm, p, i, rr = _run_rank(dev, score, pri, idle, runs, starv, period)
    ep, ei, er = pri.copy(), idle.copy(), runs.copy()
    want = rs.rank_step_np(score, ep, ei, er, starv, period)
    assert (perm == want).all()
    assert (p == ep).all() and (i == ei).all() and (rr == er).all()
    assert sorted(perm.tolist()) == list(range(n))      # a permutation (size-independent property)


def test_rank_empty_and_inf(dev):
    perm, *_ = _run_rank(dev, np.zeros(0, np.float32), np.zeros(0, np.int32), np.zeros(0, np.int32),
                         np.zeros(0, np.int32), 3, 2)
    assert perm.shape == (0,)
    score = np.array([np.inf, -np.inf, 1.0, -1.0, 0.0, -0.0, np.inf], np.float32)
    perm, *_ = _run_rank(dev, score, None, None, None, -1, 0)
    assert perm.tolist() == rs.order_np(score, None, use_pri=False).tolist() == [0, 6, 2, 4, 5, 3, 1]


def test_literal_python_equivalence(dev):
    """Against the literal per-object Python (what the reference executes)."""
    r = np.random.RandomState(5)
    n, starv, period = 4096, 50, 7
    score = r.standard_normal(n).astype(np.float16).astype(np.float32)
    reqs = [rs.Req(str(k), float(score[k])) for k in range(n)]
    for q in reqs:
        q.pri, q.idle, q.runs = -int(r.rand() < 0.3), int(r.randint(0, 100)), int(r.randint(-1, 9))
    pri = np.array([q.pri for q in reqs], np.int32)
    idle = np.array([q.idle for q in reqs], np.int32)
    runs = np.array([q.runs for q in reqs], np.int32)
    lit = [int(q.request_id) for q in rs.opt_order(reqs, starv, period)]
    perm, p, i, rr = _run_rank(dev, score, pri, idle, runs, starv, period)
    assert perm.tolist() == lit
    assert [(q.pri, q.idle, q.runs) for q in reqs] == list(zip(p.tolist(), i.tolist(), rr.tolist()))


@pytest.mark.parametrize("n", [1, 64, 1000, 8192, 65536])
def test_age_update(dev, n):
    from vllm_ltr_amd.rank import age_update
    r = np.random.RandomState(n)
    pri = -(r.rand(n) < 0.3).astype(np.int32)
    idle = r.randint(0, 100, n).astype(np.int32)
    runs = r.randint(-3, 10, n).astype(np.int32)
    ran = (r.rand(n) < 0.25).astype(np.uint8)
    d = [torch.from_numpy(a.copy()).to(dev) for a in (pri, idle, runs)]
    age_update(torch.from_numpy(ran).to(dev), *d)
    rs.age_update_np(ran, pri, idle, runs)
    assert (d[0].cpu().numpy() == pri).all() and (d[1].cpu().numpy() == idle).all() and (d[2].cpu().numpy() == runs).all()


def test_multi_step_replay_of_reference_runs(dev):
    """The reference's own multi-step schedule() runs (tests/golden/rank_steps.npz):
    device-resident counters, kernels only."""
    from vllm_ltr_amd.rank import RankWorkspace, age_update, rank_step
    z = np.load(os.path.join(GOLDEN, "rank_steps.npz"))
    ws = RankWorkspace(dev)
    for fi in range(int(z["n_cases"])):
        g = lambda k: z[f"f{fi}_{k}"]
        score, starv, period = g("score"), int(g("starv")), int(g("period"))
        orders, ran, present, states = g("orders"), g("ran"), g("present"), g("states")
        n = len(score)
        pri = torch.zeros(n, dtype=torch.int32, device=dev)
        idle = torch.zeros_like(pri); runs = torch.zeros_like(pri)
        sc = torch.from_numpy(score).to(dev)
        concat = g("concat")
        for step in range(orders.shape[0]):
            want = orders[step][orders[step] >= 0]
            # slot order = list(waiting)+list(running)+list(swapped) as the reference built it (scheduler.py:985):
            # ties in (pri, score) are broken by it, so the request ids must match exactly
            # the state stays in per-request slots on the device for the whole run (slot = request id); each
            # step only hands over `members`
            members = torch.from_numpy(concat[step][concat[step] >= 0].astype(np.int32)).to(dev)
            if len(want):
                perm = rank_step(sc, pri, idle, runs, starv, period, ws, members=members)
                got = members[perm.long()].cpu().numpy()
                assert got.tolist() == want.tolist(), f"case {fi} step {step}"
            alive = torch.from_numpy(np.nonzero(present[step])[0].astype(np.int32)).to(dev)
            if alive.numel():
                if step % 2 == 0:     # ran as a mask by position ...
                    age_update(torch.from_numpy(ran[step]).to(dev)[alive.long()].contiguous(), pri, idle, runs, members=alive)
                else:                 # ... or as the ascending slot list of running_this_step
                    rs_ = torch.from_numpy(np.nonzero(ran[step])[0].astype(np.int32)).to(dev)
                    age_update(None, pri, idle, runs, members=alive, ran_slots=rs_)
                st = torch.stack([pri, idle, runs], 1).cpu().numpy()
                a = alive.cpu().numpy()
                assert (st[a] == states[step][a]).all(), f"case {fi} step {step}"


@pytest.mark.parametrize("n,budget,max_seqs", [(1, 10, 1), (100, 64, 4), (1000, 2048, 256), (8192, 4096, 256),
                                               (8192, 10**6, 10**6), (5000, 300, 10**6)])
def test_budget_prefix(dev, n, budget, max_seqs):
    from vllm_ltr_amd.rank import budget_prefix
    r = np.random.RandomState(n + budget)
    perm = r.permutation(n).astype(np.int32)
    need = r.randint(1, 64, n).astype(np.int32)
    if n > 50:
        need[r.randint(0, n, 2)] = 0
    seqs = np.ones(n, np.int32)
    chunk = None
    if n >= 100:                      # mixed groups: best_of > 1 prompts (chunkable, 2 seqs) and multi-sequence decodes
        seqs = r.randint(1, 3, n).astype(np.int32)
        chunk = ((seqs == 1) | (r.rand(n) < 0.5)).astype(np.uint8)
    nsel, ran, granted = budget_prefix(torch.from_numpy(perm).to(dev), torch.from_numpy(need).to(dev),
                                       torch.from_numpy(seqs).to(dev), budget, max_seqs,
                                       chunkable=None if chunk is None else torch.from_numpy(chunk).to(dev))
    want_n, want_g = rs.budget_walk(need[perm], seqs[perm], budget, max_seqs, None if chunk is None else chunk[perm])
    assert int(nsel.item()) == want_n
    ran = ran.cpu().numpy(); granted = granted.cpu().numpy()
    assert (ran[perm[:want_n]] == 1).all() and ran.sum() == want_n
    assert granted[perm[:want_n]].tolist() == want_g and granted.sum() == sum(want_g)


@pytest.mark.parametrize("n,starv,period,sparse", [(1, 3, 2, False), (777, 5, 3, True), (8192, 200, 10, False),
                                                    (8192, 7, 2, True), (12288, -1, 0, False), (30000, 9, 4, True)])
def test_queue_step_two_launches_vs_oracle(dev, n, starv, period, sparse):
    """ltr_queue_step (rank + budget-walk selection + promote/demote write-back + aging in two launches) over a
    few consecutive steps against the oracle's separate functions; `sparse`: the queue occupies a random subset
    of the slots and is visited in a random order (members)."""
    from vllm_ltr_amd.rank import DeviceQueue
    r = np.random.RandomState(n + starv)
    cap = n * 2 if sparse else n
    q = DeviceQueue(dev, starv=starv, period=period, capacity=cap)
    score = r.standard_normal(cap).astype(np.float16).astype(np.float32)
    q.append(torch.from_numpy(score))
    pri = -(r.rand(cap) < 0.2).astype(np.int32)
    idle = r.randint(0, max(2, 2 * max(starv, 1)), cap).astype(np.int32)
    runs = r.randint(-2, period + 2, cap).astype(np.int32)
    q._pri[:cap] = torch.from_numpy(pri).to(dev); q._idle[:cap] = torch.from_numpy(idle).to(dev)
    q._runs[:cap] = torch.from_numpy(runs).to(dev)
    members = r.permutation(cap)[:n].astype(np.int32) if sparse else None
    mem_d = None if members is None else torch.from_numpy(members).to(dev)
    sl = members.astype(np.int64) if sparse else np.arange(n)
    for step in range(3):
        need = r.randint(1, 64, n).astype(np.int32)
        seqs = r.randint(1, 3, n).astype(np.int32)
        chunk = ((seqs == 1) | (r.rand(n) < 0.5)).astype(np.uint8)
        perm, nsel, ran, granted = q.step(torch.from_numpy(need).to(dev), torch.from_numpy(seqs).to(dev), 2048, 256,
                                          members=mem_d, chunkable=torch.from_numpy(chunk).to(dev), want_granted=True)
        p, i_, r_ = pri[sl].copy(), idle[sl].copy(), runs[sl].copy()
        want = rs.rank_step_np(score[sl], p, i_, r_, starv, period)
        assert (perm.cpu().numpy() == want).all(), step
        wn, wg = rs.budget_walk(need[want], seqs[want], 2048, 256, chunk[want])
        assert int(nsel.item()) == wn
        wr = np.zeros(n, np.uint8); wr[want[:wn]] = 1
        assert (ran.cpu().numpy() == wr).all()
        g = granted.cpu().numpy()
        assert g[want[:wn]].tolist() == wg and g.sum() == sum(wg)
        rs.age_update_np(wr, p, i_, r_)
        pri[sl], idle[sl], runs[sl] = p, i_, r_
        assert (q._pri[:cap].cpu().numpy() == pri).all() and (q._idle[:cap].cpu().numpy() == idle).all()
        assert (q._runs[:cap].cpu().numpy() == runs).all()            # slots outside the queue are untouched


def test_budget_prefix_vs_reference_schedule(dev):
    """ltr_budget_prefix on the orders / needs the reference's schedule() saw: same selected set,
    same granted chunk sizes (tests/golden/rank_steps.npz)."""
    from vllm_ltr_amd.rank import budget_prefix
    z = np.load(os.path.join(GOLDEN, "rank_steps.npz"))
    for fi in range(int(z["n_cases"])):
        g = lambda k: z[f"f{fi}_{k}"]
        B, S = int(g("token_budget")), int(g("max_num_seqs"))
        for step in range(g("orders").shape[0]):
            o = g("orders")[step]
            o = o[o >= 0].astype(np.int32)
            if len(o) == 0:
                continue
            need = torch.from_numpy(g("need_tokens")[step].astype(np.int32)).to(dev)
            seqs = torch.from_numpy(g("need_seqs")[step].astype(np.int32)).to(dev)
            chunk = torch.from_numpy(g("chunkable")[step].astype(np.uint8)).to(dev)
            # perm indexes requests directly (ids are 0..n-1); pad-free: only queued ids appear in o
            nsel, ran, granted = budget_prefix(torch.from_numpy(o).to(dev), need, seqs, B, S, chunkable=chunk)
            n = int(nsel.item())
            want = np.nonzero(g("ran")[step])[0]
            assert sorted(o[:n].tolist()) == want.tolist(), (fi, step)
            gr = granted.cpu().numpy()
            assert gr[o[:n]].tolist() == g("granted")[step][o[:n]].tolist(), (fi, step)
            assert (ran.cpu().numpy()[o[:n]] == 1).all() and (ran.cpu().numpy()[o[n:]] == 0).all()


def _reserve_case(z, c):
    g = lambda k: z[f"c{c}_{k}"]
    return (g("perm").astype(np.int32), int(g("n_selected")), g("state").astype(np.uint8), g("phys").astype(np.int32),
            g("logical").astype(np.int32), g("nrun").astype(np.int32), g("nswap").astype(np.int32), int(g("need")),
            g("action"), int(g("n_exec")))


def test_reserve_select_vs_reference_calls(dev):
    """ltr_reserve_select on the recorded calls of the reference's reserve_free_blocks
    (tests/golden/reserve_calls.npz): same evicted set, same class per request, same n_exec."""
    from vllm_ltr_amd.rank import reserve_select
    z = np.load(os.path.join(GOLDEN, "reserve_calls.npz"))
    t = lambda a: torch.from_numpy(a).to(dev)
    seen = {1: 0, 2: 0, 3: 0}
    for c in range(int(z["n_calls"])):
        perm, nsel, state, phys, logical, nrun, nswap, need, want_a, want_n = _reserve_case(z, c)
        act, nexec, _ = reserve_select(t(perm), torch.tensor([nsel], dtype=torch.int32, device=dev), t(state), t(phys),
                                       t(logical), t(nrun), t(nswap), need)
        assert act.cpu().numpy().tolist() == want_a.tolist(), c
        assert int(nexec.item()) == want_n, c
        for k in seen:
            seen[k] += int((want_a == k).sum())
    assert all(v > 0 for v in seen.values()), seen       # every eviction class is exercised


@pytest.mark.parametrize("n,frac_sel,need", [(1, 1.0, 5), (64, 0.3, 40), (1000, 0.1, 300), (8192, 0.03, 2500),
                                             (8192, 0.03, 10**7), (5000, 0.5, 0), (3000, 0.0, 100)])
def test_reserve_select_vs_oracle(dev, n, frac_sel, need):
    from vllm_ltr_amd.rank import reserve_select
    r = np.random.RandomState(n + need % 1000)
    perm = r.permutation(n).astype(np.int32)
    nsel = int(n * frac_sel)
    state = r.randint(0, 3, n).astype(np.uint8)
    phys = r.randint(1, 20, n).astype(np.int32)
    logical = r.randint(1, 20, n).astype(np.int32)
    nrun = (state == 1).astype(np.int32)
    nswap = (state == 2).astype(np.int32)
    t = lambda a: torch.from_numpy(a).to(dev)
    act, nexec, _ = reserve_select(t(perm), torch.tensor([nsel], dtype=torch.int32, device=dev), t(state), t(phys),
                                   t(logical), t(nrun), t(nswap), need)
    want_a, want_n = rs.reserve_select(perm, nsel, state, phys, logical, nrun, nswap, need)
    assert act.cpu().numpy().tolist() == want_a.tolist()
    assert int(nexec.item()) == want_n
    # device-side accumulation of gpu_block_required (scheduler.py:1137-1211) + need = required - (free - watermark)
    seqs = r.randint(1, 3, n).astype(np.int32)
    sel = perm[:nsel]
    required = int(np.where(state[sel] == 1, seqs[sel], np.where(state[sel] == 2, phys[sel] + nswap[sel], logical[sel])).sum())
    free_minus_wm = required - need
    act2, nexec2, req2 = reserve_select(t(perm), torch.tensor([nsel], dtype=torch.int32, device=dev), t(state), t(phys),
                                        t(logical), t(nrun), t(nswap), free_minus_wm, new_seqs=t(seqs))
    assert int(req2.item()) == required
    assert act2.cpu().numpy().tolist() == want_a.tolist() and int(nexec2.item()) == want_n


def test_plan_step_vs_reference_calls(dev):
    """MI355XRanker.plan_step (budget walk + eviction choice on the device) replays the recorded
    reserve_free_blocks calls of the reference: same swap-out list in the reference's eviction
    order, same put-back list, same execute_pinned_requests."""
    from types import SimpleNamespace
    from vllm_ltr_amd.plugin import MI355XRanker
    ranker = MI355XRanker.__new__(MI355XRanker)              # plan_step needs the device only
    ranker.device = torch.device(dev)
    z = np.load(os.path.join(GOLDEN, "reserve_calls.npz"))
    for c in range(int(z["n_calls"])):
        perm, nsel, state, phys, logical, nrun, nswap, need, want_a, want_n = _reserve_case(z, c)
        ordered = [SimpleNamespace(request_id=int(r)) for r in perm]
        # a budget walk that selects exactly the recorded prefix: one token each, budget = nsel
        new_tokens = np.ones(len(perm), np.int32)
        new_seqs = np.ones(len(perm), np.int32)
        sel = perm[:nsel]
        required = int(np.where(state[sel] == 1, 1, np.where(state[sel] == 2, phys[sel] + nswap[sel], logical[sel])).sum())
        blocks = dict(state=state[perm], phys=phys[perm], logical=logical[perm], nrun=nrun[perm], nswap=nswap[perm],
                      free=required - need, watermark=0)
        if nsel == 0:
            continue
        plan = ranker.plan_step(ordered, new_tokens, new_seqs, token_budget=nsel, max_num_seqs=10**6, blocks=blocks)
        assert [o.request_id for o in plan["selected"]] == sel.tolist()
        assert [o.request_id for o in plan["swap_out"]] == [int(r) for r in perm[nsel:][::-1] if want_a[r] == 1], c
        assert [o.request_id for o in plan["put_back"]] == [int(r) for r in sel[::-1] if want_a[r] in (2, 3)], c
        assert len(plan["execute"]) == want_n, c


def test_scan_kernels_edge_inputs(dev):
    """Empty queues and error codes of the scan kernels through the C ABI."""
    from vllm_ltr_amd import _lib
    from vllm_ltr_amd.rank import budget_prefix, reserve_select
    e32 = torch.zeros(0, dtype=torch.int32, device=dev)
    e8 = torch.zeros(0, dtype=torch.uint8, device=dev)
    nsel, ran, granted = budget_prefix(e32, e32, e32, 100, 10)
    assert int(nsel.item()) == 0 and ran.numel() == 0 and granted.numel() == 0
    act, nexec, req = reserve_select(e32, nsel, e8, e32, e32, e32, e32, 5)
    assert act.numel() == 0 and int(nexec.item()) == 0
    lib = _lib.load()
    assert lib.ltr_reserve_select(None, None, None, None, None, None, None, None, 4, 1, None, None, None, None) == -22   # LTR_E_INVAL
    assert b"ltr_reserve_select" in lib.ltr_last_error()
    # nothing selected, pressure: only unselected running requests can be evicted
    n = 50
    perm = torch.arange(n, dtype=torch.int32, device=dev)
    state = torch.ones(n, dtype=torch.uint8, device=dev)
    ones = torch.ones(n, dtype=torch.int32, device=dev)
    act, nexec, _ = reserve_select(perm, torch.zeros(1, dtype=torch.int32, device=dev), state, 2 * ones, ones, ones, 0 * ones, 7)
    a = act.cpu().numpy()
    assert a[-4:].tolist() == [1, 1, 1, 1] and a[:-4].sum() == 0 and int(nexec.item()) == 0   # ceil(7 / 2) victims from the end
