"""CPU: the `torch.cat` interception the learner wraps the representation calls in (`algorithm/adjacent_cat.py`:
`AdjacentCat`, a TorchFunctionMode that hands back a view for side-by-side operands and — for widths a fused Linear +
Tanh head takes in two blocks — a `DeferredCat` stand-in) is transparent to user plugin code.

(1) A battery of the concatenation idioms the reference's plugin files use (surveyed over envs/*/nn*.py and
    tests/nn*.py: cat -> module call, cat AS the returned state, cat -> rnn, three operands, dim -2 / -3, indexing,
    arithmetic, isinstance checks), with every two-block last-dim cat deferred: results and gradients bit for bit those
    of the plain call.
(2) Every reference plugin file that imports here: its `ModelRep.forward` over a [batch, L] window under the mode,
    deferring at the width of EVERY concatenation it performs, against the plain call, bit for bit (the observation
    shapes are read from the file's own `assert self.obs_shapes[i] == ...` lines; the reference checkout is only present
    in the build container, the test is skipped elsewhere)."""
import ast
import glob
import importlib.util
import inspect
from pathlib import Path

import pytest
import torch
from torch import nn

import algorithm.nn_models as m
from algorithm.adjacent_cat import AdjacentCat, DeferredCat

REF = Path('/root/reference')


def _mode(widths=None):
    return AdjacentCat(defer_width=1 << 20, widths=widths, require_cuda=False)


class _Idioms(nn.Module):
    def __init__(self):
        super().__init__()
        self.dense = nn.Sequential(nn.Linear(10, 8), nn.Tanh())
        self.rnn = nn.GRU(10, 8, batch_first=True)
        self.conv = nn.Conv2d(6, 4, 3)

    def forward(self, a, b, img):
        out = {}
        out['into_module'] = self.dense(torch.cat([a, b], dim=-1))
        out['is_the_state'] = torch.cat([a, b], dim=-1)
        out['into_rnn'] = self.rnn(torch.cat([a, b], dim=-1))[0]
        out['three'] = torch.concat([a, b, a], dim=-1)
        out['dim_m2'] = torch.cat([a, a], dim=-2)
        out['dim_m3'] = self.conv(torch.cat([img, img], dim=-3))
        c = torch.cat([a, b], dim=-1)
        out['index'] = c[..., 2:7] * 2 + c[..., :5]
        out['method'] = c.sum(-1, keepdim=True).expand(-1, -1, 3)
        out['isinstance'] = torch.ones(1) * float(isinstance(torch.cat([a, b], -1), torch.Tensor) or True)
        out['unsqueeze_cat'] = torch.cat([a.unsqueeze(-2), a.unsqueeze(-2)], dim=-2).flatten(-2)
        out['keyword'] = torch.cat(tensors=[a, b], dim=-1) if False else torch.cat((a, b), -1) + 1
        d = torch.cat([b, a], dim=-1)
        d = d + 0                      # an operator on the stand-in
        out['operator'] = d
        out['len_shape'] = torch.zeros(len(torch.cat([a, b], -1)), torch.cat([a, b], -1).shape[-1])
        return out


def test_concatenation_idioms_are_transparent():
    torch.manual_seed(0)
    net = _Idioms()
    a = torch.randn(3, 5, 6, requires_grad=True)
    b = torch.randn(3, 5, 4, requires_grad=True)
    img = torch.randn(3, 3, 8, 8)
    plain = net(a, b, img)
    g_plain = torch.autograd.grad(sum(v.sum() for v in plain.values()), [a, b, *net.parameters()], allow_unused=True)
    with _mode():
        wrapped = net(a, b, img)
    seen_deferred = isinstance(wrapped['is_the_state'], DeferredCat)
    assert seen_deferred, 'the battery must exercise the stand-in'
    from algorithm.sac_base import _real
    wrapped = {k: _real(v) for k, v in wrapped.items()}      # (what `get_l_states` does with the representation's output)
    g_wrapped = torch.autograd.grad(sum(v.sum() for v in wrapped.values()), [a, b, *net.parameters()], allow_unused=True)
    for k in plain:
        assert type(wrapped[k]) is torch.Tensor and torch.equal(wrapped[k], plain[k]), k
    for gp, gw in zip(g_plain, g_wrapped):
        assert (gp is None) == (gw is None) and (gp is None or torch.equal(gp, gw))


def test_side_by_side_operands_become_a_view():
    joint = torch.randn(4, 6, 9)
    obs, act = joint[..., :5], joint[..., 5:]
    with _mode(widths=set()):
        v = torch.cat([obs, act], dim=-1)
        w = torch.cat([act, obs], dim=-1)            # not in memory order: ATen's copy
    assert v.data_ptr() == joint.data_ptr() and torch.equal(v, joint)
    assert w.data_ptr() != joint.data_ptr() and torch.equal(w, torch.cat([act, obs], -1))


# ---------------------------------------------------------------------------------------------------------------------
def _load(path):
    spec = importlib.util.spec_from_file_location('ref_plugin_cat_' + str(abs(hash(path))), path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _obs_shapes(path, mod):
    """the observation shapes `ModelRep._build_model` asserts (first assert per index), evaluated in the file's globals"""
    tree = ast.parse(Path(path).read_text())
    shapes = {}
    for cls in tree.body:
        if isinstance(cls, ast.ClassDef) and cls.name == 'ModelRep':
            for node in ast.walk(cls):
                if (isinstance(node, ast.Assert) and isinstance(node.test, ast.Compare)
                        and isinstance(node.test.left, ast.Subscript)
                        and ast.unparse(node.test.left.value) == 'self.obs_shapes'):
                    i = ast.literal_eval(node.test.left.slice)
                    shapes.setdefault(i, eval(ast.unparse(node.test.comparators[0]), vars(mod)))
    return [tuple(shapes[i]) for i in range(len(shapes))] if shapes and sorted(shapes) == list(range(len(shapes))) else None


_FILES = sorted(f for p in ('envs/*/nn*.py', 'envs/*/*/nn*.py', 'tests/nn*.py') for f in glob.glob(str(REF / p)))
_KW = {'pe': 'ROPE', 'gate': 'RESIDUAL'}


def _build_and_call(mod, path):
    shapes = _obs_shapes(path, mod) or ([(10,), (3, 30, 30)] if '/tests/' in path else [(6,)])
    last = None
    for d_sizes, c_size in (([], 3), ([3], 0), ([2, 3], 2)):
        try:
            torch.manual_seed(0)
            sig = inspect.signature(mod.ModelRep._build_model)
            kw = {k: _KW[k] for k, p in sig.parameters.items() if k in _KW and p.default is inspect.Parameter.empty}
            rep = mod.ModelRep([f'obs{i}' for i in range(len(shapes))], shapes, d_sizes, c_size, False, **kw)
            B, L, A = 2, 4, sum(d_sizes) + c_size
            g = torch.Generator().manual_seed(1)
            obs = [torch.randn(B, L, *s, generator=g) for s in shapes]
            pre_action = torch.randn(B, L, A, generator=g)
            pad = torch.zeros(B, L, dtype=torch.bool)
            if isinstance(rep, m.ModelBaseAttentionRep):
                index = torch.arange(L).unsqueeze(0).repeat(B, 1)
                call = lambda: rep(L, index, obs, pre_action, None, padding_mask=pad)    # noqa: E731
            else:
                call = lambda: rep(obs, pre_action, None, padding_mask=pad)              # noqa: E731
            rep.eval()
            call()
            return rep, call
        except Exception as e:     # the next action layout
            last = e
    raise last


@pytest.mark.skipif(not REF.exists(), reason='reference checkout not present')
def test_reference_representations_under_the_cat_modes():
    ran, skipped = 0, []
    for path in _FILES:
        try:
            mod = _load(path)
        except (ImportError, ModuleNotFoundError):
            continue                       # torchvision / package-relative imports: not importable in this image
        if not hasattr(mod, 'ModelRep'):
            continue
        try:
            rep, call = _build_and_call(mod, path)
        except Exception as e:
            skipped.append((path[len(str(REF)) + 1:], repr(e)[:80]))
            continue
        with torch.no_grad():
            plain = call()
            with _mode():                  # every two-block last-dim concatenation of the file is deferred
                wrapped = call()
        from algorithm.sac_base import _real
        flat_p = [t for t in (plain if isinstance(plain, tuple) else (plain,)) if isinstance(t, torch.Tensor)]
        flat_w = [_real(t) for t in (wrapped if isinstance(wrapped, tuple) else (wrapped,))
                  if isinstance(t, (torch.Tensor, DeferredCat))]
        assert len(flat_p) == len(flat_w), path
        for tp, tw in zip(flat_p, flat_w):
            assert type(tw) is torch.Tensor and torch.equal(tp, tw), path
        ran += 1
    print(f'{ran} reference representations compared; not built: {skipped}')
    assert ran >= 30, (ran, skipped)
