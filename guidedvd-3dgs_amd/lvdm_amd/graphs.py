"""HIP-graph replay of the no-grad U-Net evaluation (opt-in).

A U-Net forward is ~3 400 kernel launches.  At the resolution train_guidedvd.py runs (320x448 -> 40x56 latents) most
of them are a few microseconds long, so the step is bounded by the host's launch rate, not by the GPU.  The shapes
never change inside a DDIM run, so the whole forward is captured once into a hipGraph (torch.cuda.CUDAGraph on ROCm:
the hand-written kernels launch on torch's current stream, the capture stream during capture) and replayed with the
inputs copied into static buffers.  Same kernels in the same order as the eager call.
"""
import torch


def _sig(v):
    if torch.is_tensor(v):
        return ("T", tuple(v.shape), str(v.dtype))
    if isinstance(v, (list, tuple)):
        return tuple(_sig(u) for u in v)
    if isinstance(v, dict):
        return tuple((k, _sig(u)) for k, u in sorted(v.items()))
    return ("C", repr(v))


def _clone_static(v):
    if torch.is_tensor(v):
        return v.detach().clone()
    if isinstance(v, (list, tuple)):
        return type(v)(_clone_static(u) for u in v)
    if isinstance(v, dict):
        return {k: _clone_static(u) for k, u in v.items()}
    return v


def _copy_into(dst, src):
    if torch.is_tensor(dst):
        dst.copy_(src, non_blocking=True)
    elif isinstance(dst, (list, tuple)):
        for d, s in zip(dst, src):
            _copy_into(d, s)
    elif isinstance(dst, dict):
        for k in dst:
            _copy_into(dst[k], src[k])


class GraphedApplyModel:
    """`apply(x, t, cond, **kw)` == `model.apply_model(x, t, cond, **kw)` under no_grad, replayed from a hipGraph.
    One graph per input signature (shapes / dtypes / constant kwargs); `warmup` eager calls run first so that library
    autotuning (MIOpen find, hipBLASLt selection) and parameter caches are settled before the capture."""

    def __init__(self, model, warmup=2):
        self.model, self.warmup = model, warmup
        self._graphs = {}
        self._state = None       # what the captures saw: (id, data_ptr, version) of every parameter and buffer of the model

    def _model_state(self):
        """A graph holds the ADDRESSES of the packed weight images of its capture, and those are rebuilt on the next eager call when a
        parameter changes.  Three things move them without telling anyone: an in-place checkpoint load (a version bump per parameter), a
        `.data` re-assignment (`module.half()`, `.to()`, a re-layout: new data_ptr, version may not move), and a replaced Parameter / buffer
        object (new id).  All three are in this signature, for parameters AND buffers; ~1 400 attribute reads per call, on a host that runs
        ahead of the replayed graph anyway."""
        m = self.model
        if not hasattr(m, "parameters"):
            return ()
        ts = list(m.parameters()) + (list(m.buffers()) if hasattr(m, "buffers") else [])
        return tuple((id(t), t.data_ptr(), t._version) for t in ts)

    def _weights_changed(self):
        return self._state is not None and self._state != self._model_state()

    @torch.no_grad()
    def apply(self, x, t, cond, **kw):
        if self._weights_changed():
            self._graphs.clear()
            self._state = None
        key = (_sig(x), _sig(t), _sig(cond), _sig(kw))
        ent = self._graphs.get(key)
        if ent is None:
            ent = self._graphs[key] = self._capture(x, t, cond, kw)
        sx, st, sc, skw, graph, out = ent
        sx.copy_(x, non_blocking=True)
        st.copy_(t, non_blocking=True)
        _copy_into(sc, cond)
        _copy_into(skw, kw)
        graph.replay()
        return out.clone()

    def _capture(self, x, t, cond, kw):
        sx, st, sc, skw = _clone_static(x), _clone_static(t), _clone_static(cond), _clone_static(kw)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(self.warmup):
                self.model.apply_model(sx, st, sc, **skw)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            out = self.model.apply_model(sx, st, sc, **skw)
        self._state = self._model_state()   # (a change between two captures would have cleared the cache in apply() first)
        return sx, st, sc, skw, graph, out
