"""Host-side mirror of modules/utils/detection.py:24-130 -- the sequence bookkeeping of the reference's training step
(modules/detection.py:113-177): `BackboneFeatureSelector` (label-sparse gather of the backbone features over the timesteps
of a sequence) and `RNNStates` (recurrent states kept per data-loader worker, detached between steps, reset per sample).

Same class and method names as the reference.  The data movement runs in libsast_hip.so: the gather of the selected samples
of all timesteps is ONE launch per feature map (`sast_gather_samples`, backward `sast_gather_samples_bwd`), the per-sample
state reset is `sast_zero_samples`.  Feature maps may be logical NCHW tensors in channels-last memory (what the modules of
this package return) or NHWC buffers; a sample is a contiguous chunk either way.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Union

import torch

from .. import functional as SF


class BackboneFeatureSelector:
    """modules/utils/detection.py:24-47"""

    def __init__(self):
        self.features = None
        self.reset()

    def reset(self):
        self.features: Dict[int, List] = dict()

    def add_backbone_features(self, backbone_features: Dict[int, torch.Tensor], selected_indices: Optional[List[int]] = None) -> None:
        if selected_indices is not None:
            assert len(selected_indices) > 0
        for k, v in backbone_features.items():
            idx = list(range(v.shape[0])) if selected_indices is None else [int(i) % v.shape[0] for i in selected_indices]
            self.features.setdefault(k, []).append((v, idx))

    def get_batched_backbone_features(self) -> Optional[Dict[int, torch.Tensor]]:
        if len(self.features) == 0:
            return None
        out = {}
        for k, items in self.features.items():
            xs, idx = [v for v, _ in items], [i for _, i in items]
            nchw = xs[0].dim() == 4 and not xs[0].is_contiguous() and xs[0].permute(0, 2, 3, 1).is_contiguous()
            if nchw:      # logical NCHW over channels-last storage: gather the NHWC buffers, hand back the same kind of view
                out[k] = SF.as_nchw_view(SF.gather_samples([SF.as_nhwc(x) for x in xs], idx))
            else:
                out[k] = SF.gather_samples(xs, idx)
        return out


def _map_leaves(tree, fn):
    """fn applied to every tensor of a nested list / tuple / dict of tensors; the containers are rebuilt with their own types"""
    if torch.is_tensor(tree):
        return fn(tree)
    if isinstance(tree, dict):
        return {key: _map_leaves(val, fn) for key, val in tree.items()}
    if isinstance(tree, (list, tuple)):
        return type(tree)(_map_leaves(val, fn) for val in tree)
    raise NotImplementedError(f"RNNStates: unsupported state container {type(tree).__name__}")


def _zero_selected(which):
    if which is not None and len(which) == 0:
        raise AssertionError("RNNStates.reset: an empty selection (pass None to reset every sample)")

    def zero(t: torch.Tensor) -> torch.Tensor:
        if t.requires_grad:
            raise AssertionError("RNNStates.reset: saved states are detached; got a tensor that requires grad")
        return SF.zero_samples(t, which)       # in place, one launch per state tensor (sast_zero_samples)
    return zero


class RNNStates:
    """Recurrent states per data-loader worker (the role of modules/utils/detection.py:76-130; same method names so that
    modules/detection.py:141-177 can drive it): a table worker_id -> nested [(h, c) per stage], stored detached, selected
    samples zeroed at sequence starts.  The tree walk is `_map_leaves`; the zeroing runs on the device."""

    def __init__(self):
        self.states = {}

    @classmethod
    def recursive_detach(cls, inp):
        return _map_leaves(inp, torch.Tensor.detach)

    @classmethod
    def recursive_reset(cls, inp, indices_or_bool_tensor: Optional[Union[List[int], torch.Tensor]] = None):
        return _map_leaves(inp, _zero_selected(indices_or_bool_tensor))

    def save_states_and_detach(self, worker_id: int, states) -> None:
        self.states[worker_id] = self.recursive_detach(states)

    def get_states(self, worker_id: int):
        return self.states.get(worker_id)

    def reset(self, worker_id: int, indices_or_bool_tensor: Optional[Union[List[int], torch.Tensor]] = None):
        held = self.states.get(worker_id)
        if held is not None:
            self.states[worker_id] = self.recursive_reset(held, indices_or_bool_tensor)
