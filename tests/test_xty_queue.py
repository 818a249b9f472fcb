"""Host logic of the parameter-gradient queue (`algorithm/fused_rows_linear.py`): under the learner's direct mode the `x^T y`
products of a backward pass are issued four at a time (`asac_xty_multi`) and the rest when the backward pass ends.  CPU: the
launches are replaced by recorders; what is checked is WHEN and in which groups they are issued."""
import pytest
import torch


@pytest.fixture
def recorder(monkeypatch):
    import asac_amd  # noqa: F401
    from asac_amd import native
    from algorithm import fused_rows_linear as frl
    calls = []
    monkeypatch.setattr(native, 'xty', lambda g, x, w, b, accumulate=False: calls.append(('one', [w], accumulate)))
    monkeypatch.setattr(native, 'xty_multi', lambda jobs, accumulate=False: calls.append(('multi', [j[2] for j in jobs], accumulate)))
    frl.reset_queue()
    yield frl, calls
    frl.reset_queue()


class _Enqueue(torch.autograd.Function):
    """identity whose backward queues `n` products (what a fused layer's backward does under direct mode)"""

    @staticmethod
    def forward(ctx, x, frl, grads):
        ctx.frl, ctx.grads = frl, grads
        return x * 1.0

    @staticmethod
    def backward(ctx, g):
        for w, b in ctx.grads:
            ctx.frl.queue_param_grads(torch.zeros(4, w.shape[0]), torch.zeros(4, w.shape[1]), w, b)
        return g, None, None


def _grads(n):
    return [(torch.zeros(3, 2), torch.zeros(3)) for _ in range(n)]


def test_three_products_leave_together_when_the_backward_ends(recorder):
    frl, calls = recorder
    x = torch.ones(2, requires_grad=True)
    y = _Enqueue.apply(x, frl, _grads(3))
    assert not calls
    y.sum().backward()
    assert [(k, len(o), acc) for k, o, acc in calls] == [('multi', 3, True)]
    assert not frl._pending and not frl._armed


def test_four_at_a_time_and_the_rest_at_the_end(recorder):
    frl, calls = recorder
    x = torch.ones(2, requires_grad=True)
    gs = _grads(5)
    _Enqueue.apply(x, frl, gs).sum().backward()
    assert [(k, len(o)) for k, o, _ in calls] == [('multi', 4), ('one', 1)]
    assert [id(w) for _, o, _ in calls for w in o] == [id(w) for w, _ in gs]          # in the order they were queued


def test_a_layer_applied_twice_is_not_added_twice_in_one_launch(recorder):
    frl, calls = recorder
    x = torch.ones(2, requires_grad=True)
    w, b = torch.zeros(3, 2), torch.zeros(3)
    _Enqueue.apply(x, frl, [(w, b), (torch.zeros(3, 2), torch.zeros(3)), (w, b)]).sum().backward()
    # the second product into `w` waits for the launch that holds the first
    assert [(k, len(o)) for k, o, _ in calls] == [('multi', 2), ('one', 1)]


def test_a_failed_backward_does_not_strand_the_next_one(recorder):
    frl, calls = recorder
    frl._pending.append(('stale',) * 4)          # what an exception between two enqueues leaves behind
    frl._armed = True
    from algorithm.fused_mlp import direct_param_grads
    with direct_param_grads():
        assert not frl._pending and not frl._armed
        x = torch.ones(2, requires_grad=True)
        _Enqueue.apply(x, frl, _grads(2)).sum().backward()
    assert [(k, len(o)) for k, o, _ in calls] == [('multi', 2)]


def test_queue_switched_off_issues_every_product_on_its_own(recorder, monkeypatch):
    frl, calls = recorder
    monkeypatch.setattr(frl, 'QUEUE', False)
    x = torch.ones(2, requires_grad=True)
    _Enqueue.apply(x, frl, _grads(3)).sum().backward()
    assert [(k, len(o)) for k, o, _ in calls] == [('one', 1)] * 3
