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


class RNNStates:
    """modules/utils/detection.py:76-130: states[worker_id] = [(h, c)] per stage, detached when saved, zeroed per sample at
    sequence starts."""

    def __init__(self):
        self.states = {}

    def _has_states(self):
        return len(self.states) > 0

    @classmethod
    def recursive_detach(cls, inp):
        if isinstance(inp, torch.Tensor):
            return inp.detach()
        if isinstance(inp, list):
            return [cls.recursive_detach(x) for x in inp]
        if isinstance(inp, tuple):
            return tuple(cls.recursive_detach(x) for x in inp)
        if isinstance(inp, dict):
            return {k: cls.recursive_detach(v) for k, v in inp.items()}
        raise NotImplementedError

    @classmethod
    def recursive_reset(cls, inp, indices_or_bool_tensor: Optional[Union[List[int], torch.Tensor]] = None):
        if isinstance(inp, torch.Tensor):
            assert inp.requires_grad is False, 'Not assumed here but should be the case.'
            if indices_or_bool_tensor is not None:
                assert len(indices_or_bool_tensor) > 0
            return SF.zero_samples(inp, indices_or_bool_tensor)
        if isinstance(inp, list):
            return [cls.recursive_reset(x, indices_or_bool_tensor=indices_or_bool_tensor) for x in inp]
        if isinstance(inp, tuple):
            return tuple(cls.recursive_reset(x, indices_or_bool_tensor=indices_or_bool_tensor) for x in inp)
        if isinstance(inp, dict):
            return {k: cls.recursive_reset(v, indices_or_bool_tensor=indices_or_bool_tensor) for k, v in inp.items()}
        raise NotImplementedError

    def save_states_and_detach(self, worker_id: int, states) -> None:
        self.states[worker_id] = self.recursive_detach(states)

    def get_states(self, worker_id: int):
        if not self._has_states():
            return None
        if worker_id not in self.states:
            return None
        return self.states[worker_id]

    def reset(self, worker_id: int, indices_or_bool_tensor: Optional[Union[List[int], torch.Tensor]] = None):
        if not self._has_states():
            return
        if worker_id in self.states:
            self.states[worker_id] = self.recursive_reset(self.states[worker_id], indices_or_bool_tensor=indices_or_bool_tensor)
