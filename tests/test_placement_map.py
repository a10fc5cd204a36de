"""The region map of vvc_amd/placement.py without a GPU: how launch times become regions, and regions spans to allocate from."""
from vvc_amd.placement import RegionArena, GB


def _arena(labels):
    a = object.__new__(RegionArena)
    a.n_slots, a.labels = len(labels), list(labels)
    a._set_runs()
    return a


def test_split_needs_two_classes_of_times():
    assert RegionArena._split({0: 1.00, 1: 1.01, 2: 1.03}) is None
    slow, fast = RegionArena._split({0: 1.10, 1: 1.09, 2: 1.01, 3: 1.02, 4: 1.11})
    assert sorted(slow) == [0, 1, 4] and sorted(fast) == [2, 3]


def test_spans_are_the_longest_runs_less_their_edge_slots():
    a = _arena([0] * 12 + [1] * 12 + [2] * 9)
    S = RegionArena.SLOT
    assert a.spans == {0: [1 * S, 11 * S], 1: [13 * S, 23 * S], 2: [25 * S, 32 * S]}
    # a stray slot inside a region does not move it; short runs keep their ends
    b = _arena([0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 1, 1, 1, 2, 2])
    assert b.spans[0] == [5 * S, 9 * S] and b.spans[1] == [10 * S, 13 * S] and b.spans[2] == [13 * S, 15 * S]


def test_buffers_spill_into_other_runs_before_they_fail(monkeypatch):
    a = _arena([0] * 10 + [1] * 20)                    # no third region in this arena
    S = RegionArena.SLOT
    assert a.spans == {0: [1 * S, 9 * S], 1: [11 * S, 29 * S]} and a.runs[2] == [] and GB == 1 << 30
    got = []
    monkeypatch.setattr(RegionArena, "view", lambda self, off, shape, dtype: got.append(off) or off)
    import torch
    a.empty((4 * S,), torch.uint8, 0)                  # region 0: 8 slots
    a.empty((4 * S,), torch.uint8, 2)                  # no run of region 2: the run with most room, region 1's
    a.empty((6 * S,), torch.uint8, 0)                  # 4 slots left in region 0: spills behind the previous one
    assert got == [1 * S, 11 * S, 15 * S] and a.spilled == 2
    a.empty((3 * S,), torch.uint8, 0)                  # fits region 0 again
    assert got[-1] == 5 * S and a.spilled == 2
    import pytest
    with pytest.raises(RuntimeError):
        a.empty((9 * S,), torch.uint8, 1)              # 8 slots left in region 1's run, 1 in region 0's
    # a scrambled map: secondary runs of a region are used once its longest is full
    b = _arena([int(c) for c in "000000000000111122222222111111221"])
    assert b.spans == {0: [1 * S, 11 * S], 1: [25 * S, 29 * S], 2: [17 * S, 23 * S]}
    got.clear()
    b.empty((4 * S,), torch.uint8, 1)
    b.empty((2 * S,), torch.uint8, 1)
    assert got == [25 * S, 13 * S] and b.spilled == 0


def test_a_given_map_must_fit_the_arena():
    a = _arena([0] * 11 + [1] * 11 + [2] * 11)
    assert not a.set_map("0" * 12 + "1" * 12) and not a.set_map("1" * 33) and not a.set_map("0" * 32 + "3")
    assert a.labels == [0] * 11 + [1] * 11 + [2] * 11
    m = "0" * 12 + "1" * 12 + "2" * 9
    assert a.set_map(m) and a.map_string() == m and a.spans[2] == [25 * RegionArena.SLOT, 32 * RegionArena.SLOT]
