"""Host logic of policy_gradient_asr_amd.streams: how tensors that cross streams are kept alive (no GPU needed)."""
from policy_gradient_asr_amd import streams


class _FakeTensor:
    def __init__(self):
        self.recorded = []

    def record_stream(self, s):
        self.recorded.append(s)


def test_hold_outside_a_managed_step_is_record_stream():
    streams.release()
    t = _FakeTensor()
    streams.hold(t, "side")
    assert t.recorded == ["side"] and streams._held == []


def test_hold_inside_a_managed_step_keeps_the_tensor_until_the_next_step_begins():
    streams.release()
    a, b = _FakeTensor(), _FakeTensor()
    with streams.managed_step():
        streams.hold(a, "side")
        with streams.managed_step():          # nested scopes (a trainer calling a trainer) do not release early
            streams.hold(b, "side")
        assert streams._held == [a, b]
    assert a.recorded == [] and b.recorded == []
    assert streams._held == [a, b]            # still alive after the step: the GPU may be milliseconds behind the host
    c = _FakeTensor()
    with streams.managed_step():              # the next step releases the previous step's tensors on entry
        assert streams._held == []
        streams.hold(c, "side")
    assert streams._held == [c]
    streams.release()
    assert streams._held == []


def test_a_failing_step_leaves_the_scope_balanced():
    streams.release()
    try:
        with streams.managed_step():
            raise RuntimeError("boom")
    except RuntimeError:
        pass
    t = _FakeTensor()
    streams.hold(t, "s")
    assert t.recorded == ["s"]
