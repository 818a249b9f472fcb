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


class AdjacentCat(TorchFunctionMode):
    def __torch_function__(self, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if func in _CATS and args and 'out' not in kwargs:
            dim = kwargs.get('dim', kwargs.get('axis', args[1] if len(args) > 1 else 0))
            if isinstance(dim, int):
                view = joined_view(args[0], dim)
                if view is not None:
                    return view
        return func(*args, **kwargs)
