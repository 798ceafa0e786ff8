"""BatchProducer bookkeeping (gcc_amd/train_step.py): chunks of steps, lane assignment, one multi-view eigensolver call
per chunk, buffers released after the last step of a chunk.  Host logic only (fake sampler / eigensolver objects)."""
from gcc_amd.train_step import BatchProducer


class FakeSampler:
    def __init__(self, name, log):
        self.name, self.log = name, log

    def sample(self, first_id, prof=None):
        self.log.append(("sample", self.name, first_id))
        return (("q", first_id), ("k", first_id))


class FakeMulti:
    def __init__(self, name, log):
        self.name, self.log = name, log

    def multi(self, views, prof=None):
        self.log.append(("multi", self.name, [v[1] for v in views]))


class FakeSingle:                        # no .multi(): the placeholder path embeds one view at a time
    def __init__(self, log):
        self.log = log

    def __call__(self, g):
        self.log.append(("single", g))


def test_chunks_lanes_and_one_multi_call_per_chunk():
    log = []
    lanes = [(FakeSampler("A", log), FakeMulti("A", log)), (FakeSampler("B", log), FakeMulti("B", log))]
    bp = BatchProducer(lanes, lambda step: 1000 + 10 * step, "cpu", depth=2, chunk=4)
    got = [bp.get(s) for s in range(0, 10)]
    for s, (q, k) in enumerate(got):
        assert q == ("q", 1000 + 10 * s) and k == ("k", 1000 + 10 * s)
    # host-only mode produces a chunk when its first step is asked for: chunks 0, 1, 2 on lanes A, B, A
    multis = [e for e in log if e[0] == "multi"]
    assert [m[1] for m in multis] == ["A", "B", "A"]
    assert multis[0][2] == [1000, 1000, 1010, 1010, 1020, 1020, 1030, 1030]      # q and k of 4 steps, in step order
    samples = [e for e in log if e[0] == "sample"]
    assert [e[2] for e in samples] == [1000 + 10 * s for s in range(12)]            # chunk 2 is produced whole
    assert [e[1] for e in samples] == ["A"] * 4 + ["B"] * 4 + ["A"] * 4


def test_release_drops_a_chunk_after_its_last_step():
    log = []
    bp = BatchProducer([(FakeSampler("A", log), FakeMulti("A", log))], lambda s: s, "cpu", depth=2, chunk=3)
    for s in range(3):
        bp.get(s)
        assert 0 in bp.ready
        bp.release(s)
    assert 0 not in bp.ready                     # popped by the release of step 2
    bp.get(3)
    assert list(bp.ready) == [1]


def test_posemb_without_multi_is_called_per_view():
    log = []
    bp = BatchProducer([(FakeSampler("A", log), FakeSingle(log))], lambda s: s, "cpu", depth=1, chunk=2)
    bp.get(0)
    singles = [e[1] for e in log if e[0] == "single"]
    assert singles == [("q", 0), ("k", 0), ("q", 1), ("k", 1)]
