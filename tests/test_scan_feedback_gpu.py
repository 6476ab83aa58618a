"""Form feedback (qk_ctx_set_form_feedback, qk_scan_plan.hip): repeated searches of one batch shape are answered by whichever form
of the partition scan -- 16 x 16 tiles, per-wave walk, mixed sequence -- the context's measurements say is fastest on that shape; the
answer is the same bits under every form, so the choice must be invisible: ids and distances of every call equal the oracle's
(query_coordinator.cpp:612-799), whatever the context is trying at the moment.

No assertion here depends on a wall clock: the RULE is driven with injected figures (qk_ctx_set_form_times: a harvested measurement
reads ms3[form] in place of the elapsed time of its event pair), so which form answers call i is a pure function of i, and the test
states that function.  With measured times (the product's default) only the result bits are asserted."""
import numpy as np
import pytest

import oracle as O
from helpers import make_ivf, make_queries

pytestmark = pytest.mark.gpu

TILE, WALK, MIXED = 0, 1, 2
NAME = {TILE: ("k_scan", "k_scan (query-sharing)"), WALK: ("k_scan_rl",), MIXED: ("k_scan_rl (mixed)",)}


def _stores(ctx, ivf):
    from quake_amd.capi import Store
    s = Store(ctx, ivf["d"])
    s.build_csr(ivf["offsets"], ivf["ids"], ivf["vecs"])
    parent = Store(ctx, ivf["d"])
    parent.build_csr(np.array([0, ivf["nlist"]], np.int64), np.arange(ivf["nlist"], dtype=np.int64), ivf["centroids"])
    return parent, s


class Rule:
    """qk_pick_form (qk_scan_plan.hip) restated for ONE shape whose three forms are admissible, every measurement harvested by the
    next call (the test synchronises after each call): the expected form of every call under injected figures."""

    def __init__(self, static):
        self.static, self.ms, self.n, self.calls, self.pending, self.rr = static, [0.0] * 3, [0] * 3, 0, -1, 0

    def call(self, times):
        self.calls += 1
        if self.pending >= 0:
            f, ms = self.pending, times[self.pending]
            if self.n[f] >= 2 and (ms > 1.5 * self.ms[f] or ms < 0.6 * self.ms[f]):
                self.n = [0] * 3
            self.ms[f] = ms if self.n[f] == 0 else min(self.ms[f], ms) if self.n[f] == 1 else np.float32(0.75) * np.float32(self.ms[f]) + np.float32(0.25) * np.float32(ms)
            self.n[f] += 1
            if self.n[f] == 1 and any(g != f and self.n[g] >= 2 and ms > 2.0 * self.ms[g] for g in range(3)):
                self.n[f] = 2
            self.pending = -1
        best = -1
        for f in range(3):
            if self.n[f] >= 2 and (best < 0 or self.ms[f] < self.ms[best]):
                best = f
        nxt = -1
        if self.n[self.static] < 2:
            nxt = self.static
        else:
            for f in range(3):
                if self.n[f] < 2:
                    nxt = f
                    break
        if nxt < 0 and best >= 0 and self.calls % 512 == 0:
            for t in range(3):
                f = (self.rr + t) % 3
                if f != best and not (self.n[f] >= 1 and self.ms[f] > 2.0 * self.ms[best]):
                    nxt = f
                    break
            if nxt >= 0:
                self.rr = (nxt + 1) % 3
        if nxt < 0 and best >= 0 and self.calls % 16 == 0:
            nxt = best
        if nxt >= 0:
            self.pending = nxt
            return nxt
        return best if best >= 0 else self.static


@pytest.mark.parametrize("concentrated", [True, False])
def test_every_call_is_the_oracles_answer_whatever_form_runs(concentrated):
    import torch
    from quake_amd.capi import Context
    ctx = Context(0)
    ctx.set_form_feedback(True)  # (whatever QK_FORM_FEEDBACK says: this module is about the feedback)
    ivf = make_ivf(200000, 64, 64, seed=21)  # 3125 rows per list: long lists, all three forms admissible at nprobe 4
    parent, s = _stores(ctx, ivf)
    rng = np.random.default_rng(22)
    if concentrated:  # every query next to one of three rows: the whole batch lands on a handful of lists
        q = (ivf["x"][rng.integers(0, 3, 1024)] + 0.05 * rng.standard_normal((1024, 64))).astype(np.float32)
    else:
        q = make_queries(1024, 64, seed=23, like=ivf["x"])
    oi, od = O.search(q, ivf["centroids"], ivf["vecs"], ivf["ids"], ivf["offsets"], 4, 10, "l2", batched_scan=True)
    qd = torch.from_numpy(q).cuda()
    # three regimes of injected figures: the mixed form wins, then (a change of regime: every figure moves by more than 1.5 x, the
    # comparison re-opens) the tile form wins, then the walk
    regimes = [(3.0, 2.0, 1.0)] * 40 + [(0.25, 9.0, 8.0)] * 40 + [(40.0, 1.0, 30.0)] * 48
    rule = Rule(MIXED)
    forms, want = [], []
    for rep, times in enumerate(regimes):
        ctx.set_form_times(times)
        gi, gd = ctx.search(parent, s, qd, 4, 10, "l2")
        torch.cuda.synchronize()  # (the measurement in flight is harvested by the next call)
        forms.append(ctx.last_scan_kernel())
        want.append(rule.call(times))
        np.testing.assert_array_equal(gi.cpu().numpy(), oi, err_msg=f"rep {rep} form {forms[-1]}")
        np.testing.assert_array_equal(gd.cpu().numpy().view(np.uint32), od.view(np.uint32), err_msg=f"rep {rep} form {forms[-1]}")
    for rep, (f, w) in enumerate(zip(forms, want)):
        assert f in NAME[w], (rep, f, w, forms)
    # the restated rule itself says what the claim is: static form first, every form tried, then the regime's winner
    assert want[0] == MIXED and set(want[:8]) == {TILE, WALK, MIXED}
    assert set(want[8:40]) == {MIXED} and set(want[56:80]) == {TILE} and set(want[104:]) == {WALK}, want
    # measured times again: the bits stay the oracle's (which form answers is the device's business)
    ctx.set_form_times(None)
    for rep in range(12):
        gi, gd = ctx.search(parent, s, qd, 4, 10, "l2")
        torch.cuda.synchronize()
        np.testing.assert_array_equal(gi.cpu().numpy(), oi)
        np.testing.assert_array_equal(gd.cpu().numpy().view(np.uint32), od.view(np.uint32))
    # feedback off: the static rule, always
    ctx.set_form_feedback(False)
    for _ in range(3):
        ctx.search(parent, s, qd, 4, 10, "l2")
        assert ctx.last_scan_kernel() == "k_scan_rl (mixed)"
    s.close()
    parent.close()
    ctx.close()


def test_form_times_argument_checks():
    from quake_amd._lib import QuakeHipError
    from quake_amd.capi import Context
    ctx = Context(0)
    with pytest.raises(QuakeHipError):
        ctx.set_form_times((1.0, 0.0, 1.0))
    ctx.set_form_times((1.0, 2.0, 3.0))
    ctx.set_form_times(None)
    ctx.close()
