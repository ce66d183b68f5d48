"""The training step as two hipGraphs -- the host out of the hot loop (plnlp/model.py:147-171).

An eager collab step is ~46 kernel launches behind ~1.5 ms of Python / ctypes / autograd work on the host, next to
1.6 ms of GPU work: the host is 10 % from being the bottleneck, and any hiccup on it (a CPU quota, a GC pause) lands
in the step time.  Everything a step enqueues depends on the batch only through buffer CONTENTS -- except three
things, each of which gets a device-side or bucketed form here:

  * the dropout seeds, Adam's bias corrections and the learning rate change every step -> they live in device
    memory (ops.StepScalars; plnlp_epilogue.dropout_seed_ptr / adam_scalars, plnlp_adam_tensor.step_scalars), the
    host uploads 120 bytes per step, computed exactly as the eager launchers compute them;
  * the number T of nodes the batch touches sizes the row-sparse forward / backward -> the launches are sized for
    T rounded UP to a bucket of 512 rows (the compact lists are already padded with empty rows past T, so any
    extent >= T gives the same bits: that is how the eager step pads to 32), one captured graph per bucket (T
    varies by a fraction of a percent from batch to batch: one or two buckets in practice);
  * the batch's index structures (incidence lists, touched-row compaction, hot-row tables) are built by a second,
    smaller graph one batch AHEAD on the side stream, into one of two static buffer sets; its last node copies T to
    pinned memory, so the host knows the bucket of step i + 1 while step i runs -- no read-back on the critical path.

Per step the host then issues: three small copies of the batch into the static inputs, one replay of the prologue
graph (side stream), one 120-byte upload, one replay of the step graph, two event records.  Replayed steps are
bit-identical to eager ones (tests/test_hip_round3.py): same kernels, same arguments, same order.

torch.cuda.CUDAGraph is hipGraph capture / instantiate / launch on ROCm; capture happens in this process (never an
exec), allocations made while capturing come from the graph's private pool and stay put across replays."""
from __future__ import annotations

import os
from typing import Optional

import torch

from . import ops

# OPT-IN (PLNLP_CAPTURE=1, or BaseModel.pipeline(capture=True)).  Measured on MI355X (profiles/r03_capture_ab.txt,
# interleaved runs): replayed, the collab step costs the host 0.25 ms instead of 1.06 ms of work -- but the GPU needs
# 1.607 ms instead of 1.576 ms for it (+2 %; ddi +1.2 %, citation2 +0.7 %): with ROCm's graph packet capture off
# (it has to be: plnlp_amd/__init__.py) a graph's kernels are dispatched with slightly larger gaps than the same
# kernels queued eagerly back to back, and the step is GPU-bound with the host two steps ahead either way.  So the
# eager loop stays the default for throughput; the captured loop is for hosts that cannot keep up (a CPU quota shared
# by many ranks, a slower host): it makes the step time independent of host speed.
CAPTURE = {"enabled": os.environ.get("PLNLP_CAPTURE", "0") == "1", "bucket": 512, "warm_steps": 3}


class _Slot:
    """one of the two static buffer sets a batch's inputs and index structures live in"""

    def __init__(self, device, batch: int, num_neg: int, weighted: bool):
        self.pos = torch.zeros(batch, 2, dtype=torch.int64, device=device)
        self.neg = torch.zeros(batch, num_neg, 2, dtype=torch.int64, device=device)
        self.w = torch.zeros(batch, dtype=torch.float32, device=device) if weighted else None
        self.count_host = torch.zeros(1, dtype=torch.int64, pin_memory=True)
        self.batch: Optional[ops.EdgeBatch] = None
        self.pro_graph = None
        self.main = {}               # (bucket, accumulator) -> (graph, loss tensor, seed slots, feeds the accumulator)
        self.pro_done = None         # event: the prologue replay that filled this slot has finished
        self.main_done = None        # event: the step that read this slot has finished


class StepPipeline:
    """prepare(pos, neg, w) -> handle (a batch's pre-processing, started on the side stream);
    step(handle) -> detached loss.  Eager (BaseModel.prepare_edges + train_step) until the model is warm or when the
    configuration cannot be captured; the two-graph form above afterwards.  One instance per (model, data graph,
    batch shape); BaseModel.train and bench.py both drive the hot loop through it."""

    def __init__(self, model, data, num_neg: int, batch_size: int, weighted: bool, capture: Optional[bool] = None):
        self.model, self.data, self.k, self.B, self.weighted = model, data, int(num_neg), int(batch_size), bool(weighted)
        want = CAPTURE["enabled"] if capture is None else capture
        self.why_eager = self._capturable() if want else "capture not requested (PLNLP_CAPTURE=1 / capture=True turns it on)"
        self.captured = self.why_eager is None
        self.steps = 0
        self.replays = 0
        self.last_fed = False        # did the step just enqueued feed ops.LOSS_ACC (the epoch's running loss sum)?
        self._prepared = 0
        if self.captured:
            dev = model.device
            self.slots = [_Slot(dev, self.B, self.k, self.weighted) for _ in range(2)]
            self.scalars = ops.StepScalars(dev)
            self.pool = torch.cuda.graph_pool_handle()
            self.side = ops.side_stream(dev)

    # ------------------------------------------------------------------ what can be captured ----
    def _capturable(self) -> Optional[str]:
        m = self.model
        from .layer import BaseGNN, DotPredictor, MLPPredictor
        if m.device.type != "cuda":
            return "not on the GPU"
        import plnlp_amd
        if not plnlp_amd.GRAPH_REPLAY_SAFE:
            return ("ROCm graph packet capture is on (DEBUG_CLR_GRAPH_PACKET_CAPTURE != 0 when the HIP runtime "
                    "initialised): replayed graphs fault after a device-to-host read; import plnlp_amd before "
                    "touching the GPU, or export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0")
        if m.process_group is not None:
            return "data-parallel steps run collectives and per-step host logic (not captured yet)"
        if not m._fused_step:
            return "the optimiser is not the fused HIP Adam"
        if not isinstance(m.encoder, BaseGNN) or type(m.predictor) not in (DotPredictor, MLPPredictor):
            return "foreign encoder / predictor modules"
        if m._loss_override is not None:
            return "a caller-supplied loss function"
        if len(m.optimizer.param_groups) != 1:
            return "more than one optimiser parameter group"
        if not ops.PROLOGUE_OVERLAP["enabled"] or not m._throttled():
            return "prologue overlap / step throttle disabled"
        return None

    # ------------------------------------------------------------------------------ eager ----
    def _prepare_eager(self, pos, neg, w):
        return ("eager", pos, neg, w, self.model.prepare_edges(pos, neg, edges_ready=True))

    # --------------------------------------------------------------------------- prepare ----
    def prepare(self, pos, neg, w=None):
        """start the edge-only pre-processing of a future batch.  pos [b, 2], neg [b, k, 2], w [b] | None must not
        be the output of work still pending on the CURRENT stream (resident slices, or produced on the side stream)"""
        full = pos.size(0) == self.B and (w is not None) == self.weighted
        if not self.captured or not full or self._prepared < CAPTURE["warm_steps"]:
            self._prepared += 1
            return self._prepare_eager(pos, neg, w)
        slot = self.slots[self._prepared % 2]
        self._prepared += 1
        side = self.side
        with torch.cuda.stream(side):
            if slot.main_done is not None:          # the step that last read this buffer set must be through
                side.wait_event(slot.main_done)
            slot.pos.copy_(pos)
            slot.neg.copy_(neg.reshape(slot.neg.shape))
            if self.weighted:
                slot.w.copy_(w)
            if slot.pro_graph is None:
                self._capture_prologue(slot)
            slot.pro_graph.replay()
            slot.pro_done = torch.cuda.Event(blocking=True)
            slot.pro_done.record(side)
        return ("captured", slot)

    def _capture_prologue(self, slot: _Slot) -> None:
        m = self.model
        negf = slot.neg.reshape(-1, 2)
        n_edges = slot.pos.size(0) + negf.size(0)
        fused, use_channel, build, sparse_fwd = m._edge_flags(n_edges, True, True)
        g = torch.cuda.CUDAGraph()
        # (capture on the side stream itself: the replays run there)
        with torch.cuda.graph(g, stream=self.side, capture_error_mode="thread_local"):
            slot.batch = ops.EdgeBatch([slot.pos[:, 0], negf[:, 0]], [slot.pos[:, 1], negf[:, 1]], m.num_nodes,
                                       build=build, compact=use_channel, overlap=False, inputs_ready=True,
                                       compact_endpoints=sparse_fwd, record_streams=False,
                                       count_host=slot.count_host if use_channel else None)
        slot.batch._flags = (n_edges, fused, use_channel, build, sparse_fwd)
        slot.pro_graph = g

    # ------------------------------------------------------------------------------ step ----
    def step(self, handle, global_count=None):
        m = self.model
        self.steps += 1
        if handle[0] == "eager":
            _, pos, neg, w, batch = handle
            fed0 = ops.LOSS_ACC["fed"]
            loss = m.train_step(self.data, pos, neg, self.k, w, edges_ready=True, global_count=global_count,
                                prepared=batch)
            self.last_fed = ops.LOSS_ACC["fed"] != fed0
            return loss
        slot = handle[1]
        main = torch.cuda.current_stream(m.device)
        inc = slot.batch.incidence
        bucket = 0
        if isinstance(inc, ops.CompactIncidence):
            if not slot.pro_done.query():           # a batch ahead: normally long done (the host sleeps otherwise)
                import time
                t0 = time.perf_counter()
                slot.pro_done.synchronize()
                ops.StepThrottle.waited_s += time.perf_counter() - t0
            t = int(slot.count_host.item())
            q = CAPTURE["bucket"]
            bucket = min(inc._rows_cap.numel(), (t + q - 1) // q * q)
            inc._count = bucket
        if not slot.pro_done.query():
            main.wait_event(slot.pro_done)
        # (a captured loss kernel carries the accumulator it was captured with: one graph per accumulator and weight)
        acc = ops.LOSS_ACC["buf"]
        key = (bucket, None if acc is None else (acc.data_ptr(), float(ops.LOSS_ACC["weight"])))
        entry = slot.main.get(key)
        if entry is None:
            entry = slot.main[key] = self._capture_step(slot, global_count)
        graph, loss, n_seeds, self.last_fed = entry
        # ---- the scalars of THIS step, computed as the eager launchers compute them
        group = m.optimizer.param_groups[0]
        steps = {m.optimizer.state[p]["step"] for p in group["params"] if m.optimizer.state.get(p)}
        if len(steps) != 1:
            raise RuntimeError("captured step: the parameters' Adam step counts differ")
        t_adam = steps.pop() + 1
        self.scalars.upload(group["lr"], group["betas"][0], group["betas"][1], t_adam,
                            [ops.next_seed() for _ in range(n_seeds)])
        graph.replay()
        for p in group["params"]:
            st = m.optimizer.state.get(p)
            if st:
                st["step"] = t_adam
        slot.main_done = torch.cuda.Event()
        slot.main_done.record(main)
        self.replays += 1
        m._throttle(keep=None)
        return loss

    def _capture_step(self, slot: _Slot, global_count):
        m = self.model
        opt = m.optimizer
        saved = {p: opt.state[p]["step"] for p in opt.param_groups[0]["params"] if opt.state.get(p)}
        sc = self.scalars
        first_slot = sc.seed_slots = 0
        fed0 = ops.LOSS_ACC["fed"]
        ops._step_scalars["active"] = sc
        g = torch.cuda.CUDAGraph()
        try:
            # (thread_local: the permutation thread of BaseModel.train keeps issuing its copies meanwhile)
            with torch.cuda.graph(g, pool=self.pool, capture_error_mode="thread_local"):
                loss, _ = m._train_step_core(self.data, slot.pos, slot.neg, self.k, slot.w, True, global_count,
                                             slot.batch)
        finally:
            ops._step_scalars["active"] = None
            for p, s in saved.items():             # the capture enqueued nothing: the counters must not move
                opt.state[p]["step"] = s
            m._adam_sink = None
        return g, loss, sc.seed_slots - first_slot, ops.LOSS_ACC["fed"] != fed0
