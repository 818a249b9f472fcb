"""`torch.cat` of tensors that already lie side by side in memory is a view, not a launch.

The reference's recurrent representations start with `torch.cat([obs, pre_action], dim=-1)` (envs/*/nn*.py, user code
this framework runs unchanged).  The learner's static batch keeps the vector observations and the previous actions as
column blocks of one [B, L, *] tensor (`PrioritizedReplayBuffer.join_vector_obs_with_pre_action`), so inside
`with AdjacentCat():` that concatenation returns the joint tensor's columns — same values, no `CatArrayBatchedCopy`
launch (three per train step of a GRU representation: online pass, target pass, the pass after the update).

What is returned aliases the step's input buffers instead of being a fresh tensor: a module that writes INTO its
concatenation in place would change the observations the later passes of the step read.  `hip_config['adjacent_cat'] =
False` keeps ATen's copy.  Anything that is not exactly a last-dim concatenation of adjacent, equally strided column
blocks of one storage that need no gradient goes to ATen untouched.
"""
import torch
from torch.overrides import TorchFunctionMode

_CATS = (torch.cat, torch.concat, torch.concatenate)


def joined_view(tensors, dim):
    """-> the view that equals torch.cat(tensors, dim), or None when the operands do not lie side by side"""
    if not isinstance(tensors, (list, tuple)) or len(tensors) < 2:
        return None
    first = tensors[0]
    if not all(type(t) is torch.Tensor for t in tensors) or first.dim() < 2:
        return None
    nd = first.dim()
    if dim < 0:
        dim += nd
    if dim != nd - 1:
        return None
    storage = first.untyped_storage().data_ptr()
    offset, width = first.storage_offset(), 0
    for t in tensors:
        if (t.dim() != nd or t.dtype != first.dtype or t.device != first.device or t.requires_grad
                or t.shape[:-1] != first.shape[:-1] or t.stride() != first.stride() or t.stride(-1) != 1
                or t.shape[-1] == 0 or t.untyped_storage().data_ptr() != storage
                or t.storage_offset() != offset + width):
            return None
        width += t.shape[-1]
    if width > first.stride(-2):        # the joined row must fit inside the row pitch
        return None
    return first.as_strided((*first.shape[:-1], width), first.stride(), offset)


_materializing = False


class DeferredCat:
    """What `torch.cat([a, b], dim=-1)` WILL be, for a consumer that can read the two blocks where they are: the
    plugin's `self.dense(torch.cat([vec, self.conv(img)], dim=-1))` with `dense` a fused Linear + Tanh head
    (`fused_linear.LinearTanhHead`, two-input launch) never forms the concatenation — no `CatArrayBatchedCopy`
    forward, no slice + copy of its gradient backward.  Anything else that touches the object (a torch function, a
    method, an operator, an attribute) gets the real concatenation, formed once on first use."""
    __slots__ = ('parts', 'width', '_value')

    def __init__(self, parts):
        self.parts, self._value = tuple(parts), None
        self.width = sum(p.shape[-1] for p in parts)

    def materialize(self) -> torch.Tensor:
        global _materializing
        if self._value is None:
            _materializing = True
            try:
                self._value = torch.cat(self.parts, dim=-1)
            finally:
                _materializing = False
        return self._value

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        real = lambda a: a.materialize() if isinstance(a, DeferredCat) else a  # noqa: E731

        def walk(v):
            if isinstance(v, (list, tuple)):
                return type(v)(walk(i) for i in v)
            if isinstance(v, dict):
                return {k: walk(i) for k, i in v.items()}
            return real(v)
        return func(*walk(args), **walk(kwargs or {}))

    def __getattr__(self, name):
        return getattr(self.materialize(), name)


def _forward_operator(name):
    def op(self, *args, **kwargs):
        return getattr(self.materialize(), name)(*args, **kwargs)
    op.__name__ = name
    return op


for _name in ('__add__', '__radd__', '__sub__', '__rsub__', '__mul__', '__rmul__', '__truediv__', '__rtruediv__',
              '__matmul__', '__rmatmul__', '__pow__', '__neg__', '__abs__', '__getitem__', '__len__', '__iter__',
              '__eq__', '__ne__', '__lt__', '__le__', '__gt__', '__ge__', '__bool__', '__float__', '__int__',
              '__repr__', '__format__'):
    setattr(DeferredCat, _name, _forward_operator(_name))
DeferredCat.__hash__ = object.__hash__


def deferrable(tensors, dim, max_width, require_cuda=True) -> bool:
    """two float32 device tensors with the same leading shape, concatenated along the last dim, narrow enough for the
    fused head (`require_cuda=False`: the transparency tests run the same logic on host tensors)"""
    if not isinstance(tensors, (list, tuple)) or len(tensors) != 2:
        return False
    a, b = tensors
    if not (isinstance(a, torch.Tensor) and isinstance(b, torch.Tensor)) or a.dim() < 2 or a.dim() != b.dim():
        return False
    if dim < 0:
        dim += a.dim()
    return (dim == a.dim() - 1 and a.shape[:-1] == b.shape[:-1] and (a.is_cuda and b.is_cuda or not require_cuda)
            and a.device == b.device
            and a.dtype == torch.float32 and b.dtype == torch.float32 and 0 < a.shape[-1] and 0 < b.shape[-1]
            and a.shape[-1] + b.shape[-1] <= max_width)


class AdjacentCat(TorchFunctionMode):
    """`defer_width` > 0: a two-block last-dim concatenation up to that width that is not already a view comes back as a
    `DeferredCat`; `widths` (a set) restricts that to the total widths some consumer is known to take in two blocks
    (the input widths of the representation's fused Linear + Tanh heads)"""

    def __init__(self, defer_width: int = 0, widths=None, require_cuda: bool = True):
        super().__init__()
        self.defer_width, self.widths, self.require_cuda = defer_width, widths, require_cuda

    def __torch_function__(self, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if func in _CATS and args and 'out' not in kwargs and not _materializing:
            dim = kwargs.get('dim', kwargs.get('axis', args[1] if len(args) > 1 else 0))
            if isinstance(dim, int):
                view = joined_view(args[0], dim)
                if view is not None:
                    return view
                if (self.defer_width and deferrable(args[0], dim, self.defer_width, self.require_cuda)
                        and (self.widths is None or args[0][0].shape[-1] + args[0][1].shape[-1] in self.widths)):
                    return DeferredCat(args[0])
        return func(*args, **kwargs)
