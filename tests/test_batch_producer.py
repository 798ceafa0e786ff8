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


class OverflowingSampler(FakeSampler):
    """fake of DeviceRWRSampler's overflow protocol: the first pass over ``bad`` sample ids sets a status bit until grown."""

    def __init__(self, name, log, bad, bits=1):
        super().__init__(name, log)
        self.bad, self.bits, self.size, self.pending, self.snaps, self._next = set(bad), bits, 1, 0, [], 0

    def sample(self, first_id, prof=None):
        self._next += 1
        if first_id in self.bad and self.size < 4:
            self.pending |= self.bits
        return super().sample(first_id, prof)

    def status_snapshot(self):
        self.snaps.append(self.pending)
        self.pending = 0
        return len(self.snaps) - 1

    def read_snapshot(self, token):
        return self.snaps[token]

    def grow(self, bits):
        self.log.append(("grow", self.name, bits))
        self.size *= 4


def test_overflowed_chunk_is_regrown_and_resampled_into_the_same_slots():
    """sampler overflow -> grow -> the chunk is sampled again (gcc_amd/train_step.py: BatchProducer._reproduce); the ring
    position is put back so that the re-issue lands in the chunk's own slots and later chunks keep theirs."""
    log = []
    smp = OverflowingSampler("A", log, bad={1020})
    bp = BatchProducer([(smp, FakeMulti("A", log))], lambda step: 1000 + 10 * step, "cpu", depth=2, chunk=4)
    got = [bp.get(s) for s in range(8)]
    assert [q for q, _ in got] == [("q", 1000 + 10 * s) for s in range(8)]
    assert bp.regrown == 1 and ("grow", "A", 1) in log
    samples = [e[2] for e in log if e[0] == "sample"]
    assert samples == [1000, 1010, 1020, 1030] * 2 + [1040, 1050, 1060, 1070]          # chunk 0 twice, chunk 1 once
    assert smp._next == 12 - 4          # the re-issue did not advance the ring: 8 slots used by 2 chunks
    assert bp.launched == 2             # a re-issue is not a new chunk
    assert not bp.snap


def test_overflow_that_growing_does_not_cure_raises():
    import pytest

    log = []

    class Hopeless(OverflowingSampler):
        def grow(self, bits):
            self.log.append(("grow", self.name, bits))        # never helps

    bp = BatchProducer([(Hopeless("A", log, bad={0}), FakeMulti("A", log))], lambda s: s, "cpu", depth=1, chunk=2)
    with pytest.raises(RuntimeError, match="overflow persists"):
        bp.get(0)
