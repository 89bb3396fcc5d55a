"""Training-loop helpers that have no counterpart in the reference (its Lightning modules step eagerly, pl_modules/varnet_module.py:97-113).

``GraphedTrainingStep``: one training step of a FIXED shape -- forward through the drop-in model, loss, ``loss.backward()`` through the
HIP gradient kernels, the optional gradient all-reduce and the optimiser step -- captured into ONE hipGraph and replayed.  Numerically it is
the eager step: the same kernels in the same order (``tests/test_hip_grad.py::test_training_step_captured_in_one_hipgraph_matches_the_eager_step``).

MEASURED (MI355X, ROCm 7.2, ``tools/train_graph_probe.py``): replaying the graph is NOT faster than launching eagerly on this runtime --
cfg 2: 51 ms against 33.6 eager, cfg 3: 78 against 42, cfg 4: 72 against 34.8, cfg 5: 35.7 against 34.5.  The eager steps are GPU-bound already (the union of
busy intervals of the two queues equals the step), and the captured weight-gradient side lane becomes fork / join edges that the graph
executor serialises; with the side lane off the graph is the one-stream step (cfg 2: 43 ms of kernel time).  The class stays as the way to take the
host out of a step should a later runtime execute multi-stream graphs well; the benchmarks and the tests' timing use the eager step.
"""
from typing import Callable, Optional, Sequence

import torch

from . import ops


class GraphedTrainingStep:
    """Capture ``loss = loss_fn(model(*inputs, **forward_kwargs), target); loss.backward(); [grad_sync();] optimizer.step()``.

    ``optimizer`` must be capturable (``torch.optim.Adam(..., capturable=True)``: its step counter lives on the device).  ``inputs`` /
    ``target`` give the shapes; ``step()`` copies new data into the static buffers and replays.  ``warmup`` eager steps run first on the
    capture stream (they are real training steps: ``warmup_losses``) -- they size every cached workspace, which must not be allocated
    during capture.  Nothing inside ``forward`` may read the device from the host (the models' own data-dependent read -- the ACS window of a
    sampling mask, varnet.py:64-68 -- is found on the device, ``ops.acs_window_dev``).
    """

    def __init__(self, model: torch.nn.Module, loss_fn: Callable, optimizer: torch.optim.Optimizer, inputs: Sequence[torch.Tensor],
                 target: torch.Tensor, forward_kwargs: Optional[dict] = None, warmup: int = 3, grad_sync: Optional[Callable] = None):
        if not all(g.get("capturable", False) for g in optimizer.param_groups):
            raise ValueError("GraphedTrainingStep: the optimiser must be constructed with capturable=True")
        self.model, self.loss_fn, self.opt, self.sync = model, loss_fn, optimizer, grad_sync
        self.kw = dict(forward_kwargs or {})
        self.inputs = [t.clone() for t in inputs]
        self.target = target.clone()
        self.stream = torch.cuda.Stream(device=self.target.device)
        self.warmup_losses = []
        self.stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.stream), torch.enable_grad():
            for _ in range(max(1, warmup)):
                self.warmup_losses.append(self._eager_step().detach().clone())
        torch.cuda.current_stream().wait_stream(self.stream)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        self.opt.zero_grad(set_to_none=True)
        # (ops.training_capture: the kernels that re-pack the changing weights are part of the step; the packs made here belong to the
        #  graph and every packed-weight cache is invalidated when the capture ends)
        with torch.cuda.graph(self.graph, stream=self.stream), ops.training_capture(), torch.enable_grad():
            self.loss = self._eager_step(zero=False).detach()
        torch.cuda.synchronize()

    def _eager_step(self, zero: bool = True) -> torch.Tensor:
        if zero:
            self.opt.zero_grad(set_to_none=True)
        out = self.model(*self.inputs, **self.kw)
        loss = self.loss_fn(out, self.target)
        loss.backward()
        if self.sync is not None:
            self.sync()
        self.opt.step()
        return loss

    def step(self, *inputs: torch.Tensor, target: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Copy the new example into the static buffers (skipped for tensors that ARE the buffers), replay, return the static loss tensor
        (valid until the next ``step``; stream-ordered on the current stream)."""
        if inputs and len(inputs) != len(self.inputs):
            raise ValueError("GraphedTrainingStep.step: as many inputs as the step was captured with")
        for dst, src in zip(self.inputs, inputs):
            if src is not dst:
                dst.copy_(src, non_blocking=True)
        if target is not None and target is not self.target:
            self.target.copy_(target, non_blocking=True)
        self.graph.replay()
        return self.loss
