"""Batched streaming harness around the native forward.

Reproduces the call pattern and OUTPUT ORDER of the reference's hot loop
(``/root/reference/inference.py:146-205``) for a sequence of already-resized uint8 HWC frames
(video decode, ``cv2.resize`` and the video writer stay outside - they are host codec I/O):

    frame1 = first frame
    for every next frame2 (every ``frame_interval``-th loop turn, :161-164):
        for i in 1..interpolation_factor:              # :173-184
            write denormalize(model(frame1, frame2))   # identical arguments for every i (alpha unused, :174)
        write denormalize(normalize(frame1))           # :187-188  (predictions come BEFORE the earlier frame)
        frame1 = frame2
    write the last frame: as read (raw uint8) when the loop ends in the pair branch (:164-167),
                          denormalize(normalize(frame1)) when it ends in the skip branch (:198-201, frame_interval > 1)

What changes is how the work is scheduled, not what is computed:
  * pairs are processed in batches (the reference runs batch 1 with a blocking D2H per frame, :53);
  * ToTensor/Normalize and denormalize/clip/uint8 run on the GPU (emavfi_preprocess_u8 / _postprocess_u8);
  * the ``interpolation_factor`` identical forwards of a pair are computed once and emitted that many
    times (bit-identical to recomputing them);
  * THREE streams (round 5; the reference: one, with a blocking D2H per frame, inference.py:53): the uint8 frames of batch k + 1
    travel host -> HBM by hipMemcpyAsync (the SDMA engines: no compute unit involved) and are normalised by a short device kernel
    on a high-priority side stream while the caller's stream computes batch k; the predictions of batch k - 1 are turned into
    uint8 by a device kernel and copied HBM -> host on another one - chained by events.  Two buffer slots of pinned host
    memory: the host fills slot k+1 and drains slot k-1 while the GPU works on slot k; consecutive pairs share a frame, so a
    batch of n pairs moves n+1 frames (frame_interval 1), not 2n.  Measured beside back-to-back B = 8 x 720p forwards
    (tools/copy_beside_compute.py, profiles/r05_stream_*): a copy + device kernel leg costs the forward 0.6-1.2 % of its rate,
    where the round-1..4 form - the pre / post kernels reading / writing the pinned buffers themselves over PCIe
    (``zero_copy=True`` keeps it) - costs 4.6-5.5 % per leg even on a side stream: a kernel that waits on PCIe holds its
    wave slots, and the persistent convolution kernels need all of them.
``mode="recursive"`` (opt-in, not in the reference, which has no timestep input) replaces the repeated
identical prediction by recursive midpoints: factor 1 -> [1/2]; factor 3 -> [1/4, 1/2, 3/4]; factor 7 -> eighths.
``reference_quirks=False`` drops the de-normalisation of the already-[0,1] model output (appendix A of
SURVEY.md) and passes source frames through untouched; order and counts stay the same.
"""
from __future__ import annotations

from typing import Iterable, Iterator, List, Optional

import numpy as np
import torch

from . import lib as _lib


class FrameInterpolator:
    def __init__(self, model, interpolation_factor: int = 1, frame_interval: int = 1, batch_pairs: int = 8,
                 reference_quirks: bool = True, mode: str = "reference", device=None, copy_out: bool = True, zero_copy: bool = False):
        if interpolation_factor < 0 or frame_interval < 1 or batch_pairs < 1:
            raise ValueError("interpolation_factor >= 0, frame_interval >= 1, batch_pairs >= 1 required")
        if mode not in ("reference", "recursive"):
            raise ValueError("mode must be 'reference' (the reference's repeated identical prediction) or 'recursive'")
        if mode == "recursive" and (interpolation_factor + 1) & interpolation_factor:
            raise ValueError("recursive midpoints need interpolation_factor = 2^k - 1 (1, 3, 7, ...)")
        self.model = model
        self.factor = int(interpolation_factor)
        self.interval = int(frame_interval)
        self.batch_pairs = int(batch_pairs)
        self.quirks = bool(reference_quirks)
        self.mode = mode
        self.device = torch.device(device) if device is not None else next(model.parameters()).device
        if self.device.type != "cuda":
            raise RuntimeError("FrameInterpolator needs the model on a ROCm device (no CPU path)")
        self._shape = None
        self._norm = None
        # False: yield views into the pinned result buffers instead of fresh arrays (valid until the generator is
        # advanced again - enough for a writer that consumes each frame at once; saves a page-faulting 2.8-6 MB
        # allocation + copy per frame)
        self.copy_out = bool(copy_out)
        # True: the round-1..4 transport (the pre / post kernels read / write the pinned host buffers in place) instead of SDMA copies
        self.zero_copy = bool(zero_copy)

    # ---- the reference's frame selection (inference.py:158-201), as (pairs, tail) over frame indices
    @staticmethod
    def schedule(n_frames: int, frame_interval: int):
        """Returns (pairs, last, last_roundtrip): pairs = [(i1, i2)] in processing order, last = index of the frame
        written at the end (None for an empty input), last_roundtrip = the loop ended in the skip branch
        (inference.py:198-201), where the reference writes denormalize_frame(frame1_tensor) - the float32 normalise ->
        float64 de-normalise -> truncate round trip, which changes some pixels by one count - instead of the raw frame."""
        if n_frames <= 0:
            return [], None, False
        pairs, cur, frame_num, nxt = [], 0, 0, 1
        while True:
            frame_num += 1
            if nxt >= n_frames:          # cap.read() fails: both branches write frame1 and stop
                return pairs, cur, frame_num % frame_interval != 0
            if frame_num % frame_interval == 0:
                pairs.append((cur, nxt))
            cur, nxt = nxt, nxt + 1      # in the skip branch the reference also advances frame1

    def count_outputs(self, n_frames: int) -> int:
        pairs, last, _ = self.schedule(n_frames, self.interval)
        return 0 if last is None else len(pairs) * (self.factor + 1) + 1

    # ---- segment sharding (BASELINE configs[4]): one process per GPU, each with a contiguous
    # run of the stream's frame pairs.  No exchange between ranks: a pair's two frames are all a forward needs.
    @staticmethod
    def segment(n_frames: int, frame_interval: int, rank: int = 0, world: int = 1):
        """Rank ``rank``'s share of a stream of ``n_frames`` frames: ``(pairs, lo, hi, tail)``.

        ``pairs``: its contiguous slice of ``schedule()``'s pair list (sizes differ by at most one, earlier ranks take
        the remainder: ``dist.shard_range``); ``[lo, hi)``: the frame indices it has to read - with ``frame_interval`` 1 its
        last frame is the next rank's first (segment boundaries share one frame); ``tail``: it writes the stream's final
        frame (the highest rank does, also when it owns no pair).  Concatenating the ranks' outputs in rank order gives
        exactly the single-process sequence (``tests/test_stream.py``)."""
        from .dist import shard_range
        pairs, last, _ = FrameInterpolator.schedule(n_frames, frame_interval)
        a, b = shard_range(len(pairs), rank, world)
        mine = pairs[a:b]
        tail = last is not None and rank == world - 1
        need = [f for p in mine for f in p] + ([last] if tail else [])
        return mine, (min(need) if need else 0), (max(need) + 1 if need else 0), tail

    @staticmethod
    def emission_plan(n_frames: int, interpolation_factor: int, frame_interval: int, rank: int = 0, world: int = 1,
                      reference_quirks: bool = True):
        """What ``run(frames, rank, world)`` yields, symbolically and in order: ("pred", i1, i2, j) - prediction j of the
        pair (identical for every j in reference mode) -, ("src", i1) - the earlier frame of the pair, round-tripped when
        ``reference_quirks`` -, ("tail", index, roundtrip).  Pure host logic (no device needed)."""
        mine, _, _, tail = FrameInterpolator.segment(n_frames, frame_interval, rank, world)
        _, last, last_roundtrip = FrameInterpolator.schedule(n_frames, frame_interval)
        plan = []
        for i1, i2 in mine:
            plan += [("pred", i1, i2, j) for j in range(interpolation_factor)]
            plan.append(("src", i1))
        if tail:
            plan.append(("tail", last, bool(last_roundtrip and reference_quirks)))
        return plan

    # ---- buffers: two slots of pinned host memory the kernels read / write in place, the preprocessed frames of a slot in HBM
    def _alloc(self, shape):
        if self._shape == shape:
            return
        H, W, C = shape
        nb, nout = self.batch_pairs, max(self.factor if self.mode == "recursive" else 1, 1)
        self._shape = shape
        self._slots = []
        for _ in range(2):
            self._slots.append({
                "h_in": torch.empty(2 * nb, H, W, C, dtype=torch.uint8).pin_memory(),
                "h_pred": torch.empty(nb * nout, H, W, C, dtype=torch.uint8).pin_memory(),
                "h_src": torch.empty(nb, H, W, C, dtype=torch.uint8).pin_memory(),
                "x": torch.empty(2 * nb, C, H, W, dtype=torch.float32, device=self.device),
                # device-side images of the three pinned buffers (the SDMA copies' other end)
                "d_in": torch.empty(2 * nb, H, W, C, dtype=torch.uint8, device=self.device),
                "d_pred": torch.empty(nb * nout, H, W, C, dtype=torch.uint8, device=self.device),
                "d_src": torch.empty(nb, H, W, C, dtype=torch.uint8, device=self.device),
                # consumed: the preprocess kernel has read h_in (the host may restage it); pre: x is ready; fwd: the forward has read x
                # and written its predictions; done: the postprocess kernels have written h_pred / h_src (the host may drain them)
                "consumed": torch.cuda.Event(), "pre": torch.cuda.Event(), "fwd": torch.cuda.Event(), "done": torch.cuda.Event(),
                "src": torch.cuda.Event(),
            })
        # high-priority lanes: a 9-frame preprocess needs < 1 % of the CUs' time, the priority gets its workgroups dispatched between
        # those of the compute kernels (torch: lower number = higher priority)
        self._pre = _lib.side_stream(self.device, which=2, priority=-1)
        self._post = _lib.side_stream(self.device, which=3, priority=-1)

    _pool = None

    @classmethod
    def _copy_pool(cls):
        if cls._pool is None:
            from concurrent.futures import ThreadPoolExecutor
            cls._pool = ThreadPoolExecutor(max_workers=8, thread_name_prefix="emavfi-stage")
        return cls._pool

    def _stage(self, slot, frames, chunk):
        """Copy the distinct frames of `chunk` into the slot's pinned input buffer.
        Returns (number of staged frames, positions of each pair's first / second frame)."""
        order, pos = [], {}
        for a, b in chunk:
            for f in (a, b):
                if f not in pos:
                    pos[f] = len(order)
                    order.append(f)
        slot["consumed"].synchronize()    # the preprocess kernel of this slot's previous batch has read the buffer
        # plain single-threaded memcpy: a torch CPU copy_ fans out over the intra-op thread pool, which on a
        # CPU-share-limited box (more threads than granted cores) was seen to stall for 40-160 ms at a time
        h_in = slot["h_in"].numpy()
        if len(order) >= 4 and h_in[0].nbytes >= (1 << 20):
            # a few plain memcpy threads (numpy releases the GIL inside copyto): the FIRST batch's staging is the one piece of host work
            # nothing overlaps (2.5-5 ms for nine 720p frames on one thread = 4-6 % of a 64-pair run)
            list(self._copy_pool().map(lambda t: np.copyto(h_in[t[0]], frames[t[1]]), enumerate(order)))
        else:
            for i, f in enumerate(order):
                np.copyto(h_in[i], frames[f])
        # positions stay on the host: a device index tensor would be a synchronous pageable copy on the
        # main stream, i.e. the host would block behind the compute it has just enqueued
        return len(order), [pos[a] for a, _ in chunk], [pos[b] for _, b in chunk]

    @staticmethod
    def _rows(x, idx):
        """x[idx] without a device index tensor: a view when idx is a run of consecutive rows (the usual case:
        consecutive pairs share frames), otherwise a stack of row views."""
        if all(idx[k + 1] == idx[k] + 1 for k in range(len(idx) - 1)):
            return x[idx[0]:idx[0] + len(idx)]
        return torch.stack([x[i] for i in idx])

    def _predict(self, x1, x2):
        """[n, k, 3, H, W] predictions per pair: k = 1 (reference mode) or `factor` recursive midpoints."""
        with torch.no_grad():
            if self.mode == "reference" or self.factor <= 1:
                return self.model(x1, x2).unsqueeze(1)
            # the model consumes normalised frames and returns [0,1] images: re-normalise midpoints to recurse.
            # The constants are created once: torch.tensor(..., device=) is a synchronous pageable copy, i.e. the host
            # would block behind the forward it has just enqueued and stop staging / draining beside it.
            if self._norm is None:
                self._norm = (torch.tensor(_lib.IMAGENET_MEAN, device=x1.device).view(1, 3, 1, 1),
                              torch.tensor(_lib.IMAGENET_STD, device=x1.device).view(1, 3, 1, 1))
            mean, std = self._norm

            def rec(a, b, depth):
                m = self.model(a, b)
                if depth == 1:
                    return [m]
                mn = (m - mean) / std
                return rec(a, mn, depth - 1) + [m] + rec(mn, b, depth - 1)

            levels = (self.factor + 1).bit_length() - 1
            return torch.stack(rec(x1, x2, levels), dim=1)

    def run(self, frames, rank: int = 0, world: int = 1) -> Iterator[np.ndarray]:
        """Yields uint8 HWC frames in the order the reference's writer receives them.

        ``frames``: the whole stream - an iterable, or (sharded use) any object with ``len()`` and integer indexing, of which
        only this rank's segment ``[lo, hi)`` (``segment()``) is touched, e.g. a lazy video reader.  ``rank`` / ``world``:
        this process's share (one process per GPU); the default is the whole stream."""
        if not (hasattr(frames, "__len__") and hasattr(frames, "__getitem__")):
            frames = list(frames)
        n_total = len(frames)
        mine, lo, hi, tail = self.segment(n_total, self.interval, rank, world)
        pairs, last, last_roundtrip = self.schedule(n_total, self.interval)
        if last is None or (not mine and not tail):
            return
        pairs = mine
        frames = {i: np.ascontiguousarray(frames[i]) for i in range(lo, hi)}   # this rank's segment only
        first = frames[lo]
        for f in frames.values():
            if f.dtype != np.uint8 or f.ndim != 3 or f.shape != first.shape:
                raise ValueError("FrameInterpolator.run: same-shape uint8 HWC frames expected")
        self._alloc(first.shape)
        main = torch.cuda.current_stream(self.device)
        # ramp-up: with three or more batches to come the FIRST one is half-size - the GPU starts after five staged frames instead of
        # nine, and (64 pairs at batch 8: 4 + 7 x 8 + 4) the last one's drain is half as long; every other batch is full.  Per-sample
        # results do not depend on the batch a pair travels in (tests/test_gpu_parity.py::test_forward_config2_batch16_256)
        bp = self.batch_pairs
        head = bp // 2 if (bp >= 2 and len(pairs) > 2 * bp) else 0
        chunks = ([pairs[:head]] if head else []) + [pairs[i:i + bp] for i in range(head, len(pairs), bp)]
        npred = self.factor if self.mode == "recursive" else 1

        def drain(slot, chunk):
            slot["done"].synchronize()            # this batch's frames have been written into the pinned buffers
            pred_h, src_h = slot["h_pred"].numpy(), slot["h_src"].numpy()
            own = (lambda v: v.copy()) if self.copy_out else (lambda v: v)
            for k, (a, _) in enumerate(chunk):
                if self.mode == "recursive":
                    for j in range(npred):
                        yield own(pred_h[k * npred + j])
                else:
                    for _ in range(self.factor):
                        yield own(pred_h[k])
                yield own(src_h[k]) if self.quirks else frames[a]

        staged = self._stage(self._slots[0], frames, chunks[0]) if chunks else None
        prev = None
        for ci, chunk in enumerate(chunks):
            slot = self._slots[ci & 1]
            nup, ia, ib = staged
            n = len(chunk)
            with torch.cuda.stream(self._pre):
                # the slot's x was last read by the forward of batch ci - 2 (main) and by the round-trip postprocess of its frames (post)
                self._pre.wait_event(slot["fwd"])
                self._pre.wait_event(slot["done"])
                if self.zero_copy:
                    x = _lib.preprocess_u8(slot["h_in"][:nup], device=self.device, out=slot["x"][:nup])   # distinct frames, read over PCIe, normalised once
                    slot["consumed"].record(self._pre)
                else:
                    slot["d_in"][:nup].copy_(slot["h_in"][:nup], non_blocking=True)                       # hipMemcpyAsync pinned -> HBM (SDMA)
                    slot["consumed"].record(self._pre)
                    x = _lib.preprocess_u8(slot["d_in"][:nup], out=slot["x"][:nup])                       # distinct frames, normalised once
                slot["pre"].record(self._pre)
                src_here = self.quirks and not self.zero_copy and all(ia[k + 1] == ia[k] + 1 for k in range(n - 1))
                if src_here:
                    # the reference's round trip of every pair's earlier frame (inference.py:187-188) depends on the preprocess only:
                    # it leaves from this lane, ahead of the forward, instead of queueing behind the predictions at the end of the batch
                    _lib.postprocess_u8(x[ia[0]:ia[0] + n], denormalize=True, out=slot["d_src"][:n])
                    slot["h_src"][:n].copy_(slot["d_src"][:n], non_blocking=True)
                    slot["src"].record(self._pre)
            main.wait_event(slot["pre"])
            x1, x2 = self._rows(x, ia), self._rows(x, ib)
            pred = self._predict(x1, x2)                                      # [n, k, 3, H, W], on the caller's stream
            slot["fwd"].record(main)
            with torch.cuda.stream(self._post):
                self._post.wait_event(slot["fwd"])
                flat = pred.reshape(-1, *pred.shape[2:])
                flat.record_stream(self._post)                                # allocated on the caller's stream, read on this one
                if self.quirks and (x1.data_ptr() < slot["x"].data_ptr() or x1.data_ptr() >= slot["x"].data_ptr() + slot["x"].numel() * 4):
                    x1.record_stream(self._post)                              # a gathered copy (non-consecutive rows), not a view of the slot
                if self.zero_copy:
                    _lib.postprocess_u8(flat, denormalize=self.quirks, out=slot["h_pred"][:n * npred])
                    if self.quirks:
                        _lib.postprocess_u8(x1, denormalize=True, out=slot["h_src"][:n])
                else:
                    _lib.postprocess_u8(flat, denormalize=self.quirks, out=slot["d_pred"][:n * npred])
                    slot["h_pred"][:n * npred].copy_(slot["d_pred"][:n * npred], non_blocking=True)       # HBM -> pinned (SDMA)
                    if self.quirks and not src_here:
                        _lib.postprocess_u8(x1, denormalize=True, out=slot["d_src"][:n])
                        slot["h_src"][:n].copy_(slot["d_src"][:n], non_blocking=True)
                    if src_here:
                        self._post.wait_event(slot["src"])                    # `done` covers both lanes' writes into the pinned buffers
                slot["done"].record(self._post)
            if ci + 1 < len(chunks):                  # host-side staging of the next batch overlaps this batch's compute
                staged = self._stage(self._slots[(ci + 1) & 1], frames, chunks[ci + 1])
            if prev is not None:                      # emit the previous batch (its slot is reused only after this)
                yield from drain(*prev)
            prev = (slot, chunk)
        if prev is not None:
            yield from drain(*prev)
        if not tail:                         # the stream's final frame belongs to the highest rank
            return
        if last_roundtrip and self.quirks:   # skip-branch ending: the reference writes the round-tripped frame
            src = torch.from_numpy(frames[last]).unsqueeze(0).to(self.device)
            yield _lib.postprocess_u8(_lib.preprocess_u8(src), denormalize=True).cpu().numpy()[0]
        else:
            yield frames[last]
