"""Whole-step HIP graph for the composable model paths (SlowFast, ViViT, the 0D encoders, the fusion models): their training step
is several hundred small launches issued from Python, and the host, not the GPU, sets the step time (SlowFast cfg5: 9.2 ms of
kernels in a 13.7 ms step).  ``GraphedStep`` records forward + loss + backward once into a HIP graph (``torch.cuda.CUDAGraph``:
every kernel of this library is launched on torch's current stream, so stream capture sees them) and replays it per batch; the
optimizer step stays outside.  Requirements, as for any captured step: fixed input shapes, no host synchronisation inside
the step, random masks only from torch's device generator (capture-aware); the NoiseLayer of the 0D encoders, which draws from
the CPU generator as the reference does, is switched to a pinned staging buffer that is refilled before every replay.  Gradients live in static tensors that the graph overwrites on every replay.
Observed on this stack (ROCm 7.2, torch 2.10): ending the capture segfaults inside the runtime when an output or loss tensor that
still carries its grad_fn from an EARLIER EAGER step on the legacy default stream is alive during the capture (probed on the
full-size SlowFast: one such step is enough; dropping the reference or ``.detach()``-ing it, or having run the eager step on any
other stream, avoids it; reproducer: ``SF_GRAPH=1 SF_GRAPH_DEFAULT=1 python tools/slowfast_smoke.py 4``).  So: construct the
GraphedStep first, or keep eager work under ``torch.cuda.stream(side)``, or detach / drop what earlier steps returned.
"""
from typing import Callable, Sequence

import torch


class GraphedStep:
    def __init__(self, model: torch.nn.Module, loss_fn: Callable, example_inputs: Sequence[torch.Tensor], example_target: torch.Tensor,
                 warmup: int = 3):
        # NoiseLayers draw from the CPU generator: switched to one pinned staging buffer each, which the captured upload reads on
        # every replay and __call__ refills beforehand
        self.noise_layers = [mod for mod in model.modules() if type(mod).__name__ == "NoiseLayer"]
        for mod in self.noise_layers:
            mod.__dict__["_graph_mode"] = True
        self.model, self.loss_fn = model, loss_fn
        self._done = torch.cuda.Event()
        self.inputs = [t.detach().clone() for t in example_inputs]
        self.target = example_target.detach().clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                      # warm-up off the default stream: builds plans, fills the allocator
            for _ in range(warmup):
                model.zero_grad(set_to_none=True)
                self._eager()
        torch.cuda.current_stream().wait_stream(side)
        model.zero_grad(set_to_none=True)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.outputs, self.loss = self._eager()

    def _eager(self):
        out = self.model(*self.inputs)
        outs = out if isinstance(out, tuple) else (out,)
        loss = self.loss_fn(*outs, self.target)
        loss.backward()
        return out, loss

    def __call__(self, inputs: Sequence[torch.Tensor], target: torch.Tensor):
        """Copies the batch into the static buffers, replays the step; returns (outputs, loss) - static tensors, valid until
        the next call.  Parameter ``.grad`` tensors hold this batch's gradients afterwards."""
        if self.noise_layers:
            self._done.synchronize()                       # the previous replay no longer reads the staging buffers
            for mod in self.noise_layers:
                mod.refresh_static()
        for dst, src in zip(self.inputs, inputs):
            dst.copy_(src, non_blocking=True)
        self.target.copy_(target, non_blocking=True)
        self.graph.replay()
        if self.noise_layers:
            self._done.record()
        return self.outputs, self.loss
