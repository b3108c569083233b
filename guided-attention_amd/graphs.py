"""hipGraph capture of the UNet passes of the guided-attention loop.

A batch-1 UNet pass is ~1000 kernel launches of 5-15 us each; issued eagerly the host is the
bottleneck.  The three passes of the loop are therefore captured ONCE per (shapes, prompt plan) into
HIP graphs and replayed:
    g_eval : latents -> UNet forward with the capture kernels -> aggregate -> smoothed box loss   (autograd on)
    g_grad : autograd backward of that loss to the latents (replays against g_eval's saved activations)
    g_cfg  : the no-grad CFG forward on [latents, latents]
    g_joint: steps on which no latent update can follow the guidance evaluation (its loss is only logged): the
             guidance forward (cond) and the CFG pair (uncond, cond) are three independent UNet evaluations of the
             SAME latents — one no-grad batch-3 pass, sample 0's maps feeding aggregate + loss
Host control flow (thresholds, refinement, recurse) stays in Python between replays; scalars that change
per step (timestep) live in static device tensors, per-step kernel scalars (step size, alphas) stay in
the eager one-launch kernels around the graphs.  The kernels inside the graphs are exactly the eager
ones (same C-ABI calls on the capture stream).
"""
import torch

from . import ops
from .utils.ptp_utils import pin_context_projections, refresh_context_projections


class PendingLossTable:
    """The packed loss table of ONE captured evaluation on its way to the host: copied device -> pinned host memory right
    behind the replay (asynchronously, on the replay's stream) with an event behind the copy.  `.cpu()` — what
    GuidedAttention._loss_host calls on the table — waits for THAT event only, not for whatever the host has enqueued behind it
    in the meantime (the speculative backward / update / next evaluation of the refinement loop)."""

    def __init__(self, row, event):
        self.row, self.event = row, event

    def cpu(self):
        self.event.synchronize()
        return self.row.clone()


class GraphRunner:
    HOST_SLOTS = 4    # evaluations whose loss tables may be in flight to the host at once (the refinement loop keeps two)

    @classmethod
    def for_run(cls, pipe, store, prompt_embeds, latents, attention_res, smooth, sigma, ksize, normalize_eot):
        from .utils import shared_state as state
        plan = pipe._loss_plan(smooth, sigma, ksize)
        custom = getattr(state.config, "custom_loss", None) or {}
        key = (tuple((name, id(fn), str(args)) for name, (fn, args) in sorted(custom.items())), tuple(latents.shape), latents.dtype, tuple(prompt_embeds.shape), pipe._plan_key, attention_res,
               pipe.guidance_forward, normalize_eot, str(pipe.prompt) if normalize_eot else None,
               getattr(store, "capture", None), bool(getattr(pipe, "batch_loss_only_guidance", False)),
               bool(pipe.fused_aggregate_loss))
        runner = pipe._graph_cache.get(key)
        if runner is None:
            for old in pipe._graph_cache.values():
                old.release()
            pipe._graph_cache.clear()  # one live configuration: the pools hold every activation of both passes
            runner = cls(pipe, store, prompt_embeds, latents, attention_res, smooth, sigma, ksize, normalize_eot)
            pipe._graph_cache[key] = runner
        if not torch.equal(runner.embeds, prompt_embeds):
            runner.embeds.copy_(prompt_embeds)
            runner._fill_embeds3()
            refresh_context_projections(pipe.unet)  # the captured graphs read the cached text K/V tensors
        return runner

    def __init__(self, pipe, store, prompt_embeds, latents, attention_res, smooth, sigma, ksize, normalize_eot):
        self.pipe = pipe
        dev = latents.device
        self.t_dev = torch.zeros(1, dtype=torch.int64, device=dev)
        self.embeds = prompt_embeds.detach().clone()
        self.lat_g = torch.zeros_like(latents).requires_grad_(True)
        self.lat2 = torch.zeros((2,) + tuple(latents.shape[1:]), dtype=latents.dtype, device=dev)
        self.joint = bool(getattr(pipe, "batch_loss_only_guidance", False)) and prompt_embeds.shape[0] == 2
        if self.joint:
            self.lat3 = torch.zeros((3,) + tuple(latents.shape[1:]), dtype=latents.dtype, device=dev)
            self.embeds3 = torch.empty((3,) + tuple(prompt_embeds.shape[1:]), dtype=prompt_embeds.dtype, device=dev)
            self._fill_embeds3()
        self.loss_args = (smooth, sigma, ksize, normalize_eot)
        self.res = attention_res
        # static copies of unet.time_projection(t, batch): the graphs read these instead of recomputing the
        # timestep-only part of the UNet in every pass; `_set_t` refreshes them before a replay
        self.tp = {n: torch.empty_like(pipe.unet.time_projection(981, n)) for n in ((1, 2, 3) if self.joint else (1, 2))}
        self._capture(store)
        # pinned landing slots for the loss tables + one event each (PendingLossTable)
        self._host_rows = torch.empty((self.HOST_SLOTS, self.parts[4].numel()), dtype=self.parts[4].dtype).pin_memory()
        self._host_events = [torch.cuda.Event() for _ in range(self.HOST_SLOTS)]
        self._host_turn = 0

    def _set_t(self, t, *batches):
        self.t_dev.fill_(int(t))
        for n in batches:
            self.tp[n].copy_(self.pipe.unet.time_projection(int(t), n))

    def _fill_embeds3(self):
        if self.joint:  # [guidance: cond | CFG: uncond, cond]
            self.embeds3[0].copy_(self.embeds[1])
            self.embeds3[1].copy_(self.embeds[0])
            self.embeds3[2].copy_(self.embeds[1])

    # -- bodies (run eagerly for warm-up, then once more under capture)
    def _eval_body(self, store):
        pipe = self.pipe
        with torch.enable_grad():
            pipe._guidance_forward(self.lat_g, self.t_dev, self.embeds[1:2], time_projection=self.tp[1])
            parts = pipe._aggregate_loss_device(store, self.res, *self.loss_args)
        return parts, store.attention_store

    def _grad_body(self, loss):
        with torch.enable_grad():
            return torch.autograd.grad(loss, [self.lat_g], retain_graph=True)[0]

    def _cfg_body(self, store):
        with torch.no_grad():
            out = self.pipe.unet(self.lat2, self.t_dev, encoder_hidden_states=self.embeds,
                                 time_projection=self.tp[2]).sample
        return out, store.attention_store

    def _joint_body(self, store):
        pipe = self.pipe
        with torch.no_grad():
            out = pipe.unet(self.lat3, self.t_dev, encoder_hidden_states=self.embeds3, time_projection=self.tp[3]).sample
            # the guidance evaluation is sample 0: the loss reads its head-maps only (the reference's guidance forward
            # has batch 1); what stays published afterwards is what the CFG pass would have left: samples 1 and 2
            full = store.attention_store
            store.attention_store = {k: [p[: p.shape[0] // 3] for p in v] for k, v in full.items()}
            parts = pipe._aggregate_loss_device(store, self.res, *self.loss_args)
            snap = {k: [p[p.shape[0] // 3:] for p in v] for k, v in full.items()}
            store.attention_store = snap
        return out, parts, snap

    def _capture(self, store):
        calls = dict(self.pipe.unet_calls)
        self._set_t(981, *self.tp)
        torch.cuda.synchronize()
        # warm-up and capture run on the package's ONE side stream per device: the split-K scratch is kept per (device, stream)
        # and must exist before the capture opens (ops.prepare_device); the captured launches keep that stream's set
        side = self.stream = ops.side_stream(self.lat_g.device)
        ops.prepare_device(self.lat_g.device, side)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):  # warm-up: library algorithm selection and workspace allocation happen here
                parts, _ = self._eval_body(store)
                self._grad_body(parts[1])
                self._cfg_body(store)
                if self.joint:
                    self._joint_body(store)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        # No automatic garbage collection while a capture is open: a cycle collection that happens to run inside one frees
        # whatever became unreachable earlier — an older runner's hipGraphs, streams, events — and destroying those while this
        # thread's stream is capturing aborts the process (seen once in the full GPU suite, at a capture after ~500 tests).
        # torch.cuda.graph() collects on entry; between entry and exit nothing may.
        # The captured launches read derived weight tensors (LayerNorm-folded weights, transposes, convolution packs) by raw
        # pointer: the runner keeps every one handed out during the capture alive for as long as it lives itself — the caches
        # that made them may evict as they like
        with ops.no_gc(), ops.keepalive_scope() as keep:
            self._capture_graphs(store)
        self._keep = keep.tensors
        torch.cuda.synchronize()
        self.pipe.unet_calls.update(calls)  # capture / warm-up passes are not image work
        # The captured kernels read the cached text K/V projections of the static prompt buffers by raw pointer:
        # pin those cache entries so that eager passes with other contexts can never evict (= free) them.
        self._storages = {b.untyped_storage().data_ptr() for b in (self.embeds, getattr(self, "embeds3", None))
                          if b is not None}
        self.pinned = pin_context_projections(self.pipe.unet, self._storages, +1)
        GraphRunner.captures += 1

    def _capture_graphs(self, store):
        self.g_eval = torch.cuda.CUDAGraph()
        with ops.census_scope() as c_eval, torch.cuda.graph(self.g_eval, stream=self.stream):
            self.parts, self.store_eval = self._eval_body(store)
        self.loss = self.parts[1]
        self.g_grad = torch.cuda.CUDAGraph()
        with ops.census_scope() as c_grad, torch.cuda.graph(self.g_grad, pool=self.g_eval.pool(), stream=self.stream):
            self.grad = self._grad_body(self.loss)
        self.g_cfg = torch.cuda.CUDAGraph()
        with ops.census_scope() as c_cfg, torch.cuda.graph(self.g_cfg, pool=self.g_eval.pool(), stream=self.stream):
            self.noise, self.store_cfg = self._cfg_body(store)
        self.launches = {"eval": c_eval.launches, "grad": c_grad.launches, "cfg": c_cfg.launches}
        if self.joint:
            self.g_joint = torch.cuda.CUDAGraph()
            with ops.census_scope() as c_joint, torch.cuda.graph(self.g_joint, pool=self.g_eval.pool(), stream=self.stream):
                self.noise3, self.parts_joint, self.store_joint = self._joint_body(store)
            self.launches["joint"] = c_joint.launches

    captures = 0  # how many runners were ever captured (tests assert graph reuse across seeds)

    def release(self):
        """Unpin the text K/V entries (the runner is being dropped; its graphs must not be replayed afterwards)."""
        if self._storages:
            pin_context_projections(self.pipe.unet, self._storages, -1)
            self._storages = set()
            # every completed split-K / fused-loss launch returns its arrival tickets to zero; a word left non-zero means a
            # launch died half-way and every later launch on that tile silently skipped its epilogue
            if not ops.tickets_are_zero(self.lat_g.device):
                raise ops.GaError("an arrival-ticket word is non-zero after the runner's last replay: a split-K launch did not complete")

    # -- replays
    def _publish(self, store, snapshot):
        if store is not None and hasattr(store, "attention_store"):
            store.attention_store = snapshot  # "the maps of the most recent forward", as after an eager pass
            store.cur_step += 1

    def evaluate(self, latents, t, store):
        """-> (the latents leaf the captured loss depends on, loss parts).  The packed table of the parts is a
        PendingLossTable: the replay overwrites the static device table at the next evaluation, the pinned copy taken right
        behind this one does not, so the host may read it after enqueuing more work."""
        self._set_t(t, 1)
        with torch.no_grad():
            self.lat_g.copy_(latents)
        self.g_eval.replay()
        ops.add_census(self.launches["eval"])
        self._publish(store, self.store_eval)
        slot = self._host_turn % self.HOST_SLOTS
        self._host_turn += 1
        self._host_rows[slot].copy_(self.parts[4], non_blocking=True)
        self._host_events[slot].record()
        return self.lat_g, self.parts[:4] + (PendingLossTable(self._host_rows[slot], self._host_events[slot]),)

    def backward(self):
        self.g_grad.replay()
        ops.add_census(self.launches["grad"])
        return self.grad

    def cfg_forward(self, latents, t, store):
        self._set_t(t, 2)
        self.lat2.copy_(latents[0:1].expand_as(self.lat2))   # one broadcast copy instead of one launch per sample
        self.g_cfg.replay()
        ops.add_census(self.launches["cfg"])
        self._publish(store, self.store_cfg)
        return self.noise

    def joint_forward(self, latents, t, store):
        """-> (loss parts of the guidance evaluation, CFG noise prediction (2, ...)) from one batch-3 replay."""
        self._set_t(t, 3)
        self.lat3.copy_(latents[0:1].expand_as(self.lat3))
        self.g_joint.replay()
        ops.add_census(self.launches["joint"])
        self._publish(store, self.store_joint)
        return self.parts_joint, self.noise3[1:3]
