"""The fused per-frame path: frames resident in HBM -> 6D pose.

Device part (one hipGraph, ``bp_pipeline_run``): Pillow-exact bicubic stretch ->
YOLOv3 -> decode + arg-max objectness -> box rescale + crop -> FastPose -> heat-map
arg-max; 316 floats per frame come back.  Host part (``finish_record``): key-point
decoding, pPose-NMS (n = 1), key-point pruning, PnP.  Together they replace
``DetectionLoader.update`` -> ``DetectionProcessor.update`` -> the KPD main loop ->
``DataWriter.update`` of the reference (dataloader.py:330-401,438-457,678-741;
betapose_evaluate.py:145-176) for the evaluation stream.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional

import numpy as np

from . import _lib
from .eval import decode_keypoints
from .ops import solve_pnp
from .pPose_nms import pose_nms

RESULT_FLOATS = _lib.RESULT_FLOATS


class FramePipeline:
    def __init__(self, det_model, pose_model, frame_h: int = 480, frame_w: int = 640, batch: int = 1,
                 confidence: float = 0.01, num_classes: int = 80, use_graph: bool = True, keep_heatmaps: bool = False,
                 frames=None):
        import torch
        _lib.require_gpu()
        self.det, self.pose = det_model, getattr(pose_model, "pyranet", pose_model)
        self.H, self.W, self.batch = int(frame_h), int(frame_w), int(batch)
        self.use_graph = bool(use_graph)
        dev = "cuda:%d" % self.det._device if self.det._device is not None else "cuda"
        self.det.cuda()
        self.pose.cuda()
        dev = "cuda:%d" % self.det._device
        # ``frames``: a device frame buffer shared with other pipelines (several objects looking at the same frame)
        self.frames = frames if frames is not None else torch.zeros((self.batch, self.H, self.W, 3), dtype=torch.uint8, device=dev)
        assert tuple(self.frames.shape) == (self.batch, self.H, self.W, 3) and self.frames.dtype == torch.uint8
        self.results = torch.zeros((self.batch, RESULT_FLOATS), dtype=torch.float32, device=dev)
        self.heatmaps = torch.zeros((self.batch, 50, 80, 64), dtype=torch.float32, device=dev) if keep_heatmaps else None
        h = C.c_void_p()
        _lib.check(_lib.lib().bp_pipeline_create(self.det.handle, self.pose.handle, self.H, self.W, self.batch,
                                                 float(confidence), int(num_classes), self.frames.data_ptr(),
                                                 self.results.data_ptr(),
                                                 self.heatmaps.data_ptr() if keep_heatmaps else None, C.byref(h)))
        self._h = h
        self._faults_seen = 0

    def __del__(self):
        try:
            if getattr(self, "_h", None) is not None:
                _lib.lib().bp_pipeline_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def set_fixed_box(self, box_xyxy=None):
        """Throughput runs with a deterministic crop: box in frame pixels, or None."""
        if box_xyxy is None:
            _lib.check(_lib.lib().bp_pipeline_set_fixed_box(self._h, None))
        else:
            b = np.ascontiguousarray(box_xyxy, dtype=np.float32)
            _lib.check(_lib.lib().bp_pipeline_set_fixed_box(self._h, b.ctypes.data))

    def enqueue(self, stream: Optional[int] = None):
        """Launch the device part on ``stream`` (default: torch's current stream).  ``self.frames`` must
        already hold the batch; ``self.results`` is valid once the stream reaches this point."""
        _lib.check(_lib.lib().bp_pipeline_run(self._h, int(self.use_graph), stream if stream is not None else _lib.current_stream()))
        if any(getattr(m, "_latency_mode", False) for m in (self.det, self.pose)):
            self._latency_check()

    def latency_faults(self) -> int:
        """Frames this pipeline ran twice because the lone-frame latency mode's placement check failed (bp_pipeline_latency_faults)."""
        return int(_lib.lib().bp_pipeline_latency_faults(self._h))

    def prepare(self):
        """Set-up: build the frame's hipGraph now (capture + instantiate, nothing executes) instead of inside the first ``enqueue``."""
        if self.use_graph:
            _lib.check(_lib.lib().bp_pipeline_prepare(self._h))
        return self

    def kernel_count(self) -> int:
        return _lib.lib().bp_pipeline_kernel_count(self._h)

    def run(self, frames_bgr_u8) -> np.ndarray:
        """Convenience: upload frames (numpy [B,H,W,3] u8 or cuda tensor), run, return records [B,316] (host)."""
        import torch
        f = frames_bgr_u8 if hasattr(frames_bgr_u8, "is_cuda") else torch.from_numpy(np.ascontiguousarray(frames_bgr_u8))
        if f.dim() == 3:
            f = f.unsqueeze(0)
        self.frames.copy_(f, non_blocking=True)
        self.enqueue()
        return self.results.cpu().numpy()

    def _latency_check(self) -> bool:
        """Lone-frame latency mode only (Darknet.set_prefetch).  ``bp_pipeline_run`` itself waits for the frame in that mode, reads the
        engines' placement error words and -- when a launch reported a K slice on the wrong XCD, i.e. left a tile unstored -- switches
        the mode off and runs the same frame again, whichever caller drove it (``run``, ``StreamedRunner``, a bare ``enqueue``).  This
        mirrors that into the Python objects: a warning, and the engines' ``_latency_mode`` flags off (the mode stays off)."""
        n = self.latency_faults()
        if n == self._faults_seen:
            return False
        self._faults_seen = n
        import warnings
        warnings.warn("betapose_amd: block placement is not the round robin the lone-frame latency mode relies on; mode switched off, frame re-run")
        for m in (self.det, self.pose):
            if hasattr(m, "set_prefetch"):
                m.set_prefetch(False)
        return True


class StreamedRunner:
    """Keeps ``streams`` frames in flight: one engine clone + one hipGraph per HIP stream over shared filters, frame
    uploads straight from the loader's pinned slots, records copied back into a ring of pinned buffers.  At batch 1
    most layers cannot fill 256 CUs and every kernel carries a few microseconds of fixed cost, so independent frames
    overlapping on the chip is where the throughput comes from."""

    def __init__(self, det_model, pose_model, frame_h: int = 480, frame_w: int = 640, streams: int = 4,
                 confidence: float = 0.01, num_classes: int = 80, use_graph: bool = True, batch: int = 1):
        """``batch`` frames per launch and stream (the reference's ``--detbatch``, dataloader.py:284-289): the engines
        must have been created with ``max_batch >= batch``.  More frames per launch mean fewer launches, K slices and
        hand-offs per frame: 1 -> 2 -> 4 frames per launch run 945 -> 1 100 -> 1 290 frames/s."""
        import torch
        S = max(1, int(streams))
        B = max(1, int(batch))
        pose = getattr(pose_model, "pyranet", pose_model)
        dets = [det_model] + [det_model.clone() for _ in range(S - 1)]
        poses = [pose] + [pose.clone() for _ in range(S - 1)]
        self.pipes = [FramePipeline(dets[k], poses[k], frame_h, frame_w, batch=B, confidence=confidence,
                                    num_classes=num_classes, use_graph=use_graph) for k in range(S)]
        dev = self.pipes[0].frames.device
        self.S, self.B, self.H, self.W = S, B, int(frame_h), int(frame_w)
        self.streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
        self._pinned = [torch.empty((B, RESULT_FLOATS), dtype=torch.float32).pin_memory() for _ in range(2 * S)]
        self._events = [torch.cuda.Event() for _ in range(2 * S)]

    def run(self, source, on_record) -> int:
        """``source`` yields ``(index, frame[H,W,3] u8 BGR, host_address)`` and has ``release(index)`` (FrameLoader);
        ``on_record(index, rec[316])`` is called in source order once the frame's record is on the host.
        Returns the number of frames processed."""
        import torch
        L = _lib.lib()
        S, B, NS, nbytes = self.S, self.B, 2 * self.S, self.H * self.W * 3
        inflight = []          # (launch number, [source indices of its frames])
        pending = []           # source indices uploaded into the current launch's batch slots

        def finish():
            j, idxs = inflight.pop(0)
            self._events[j % NS].synchronize()
            recs = self._pinned[j % NS].numpy().copy()
            for b, idx in enumerate(idxs):
                source.release(idx)
            for b, idx in enumerate(idxs):
                on_record(idx, recs[b])

        def launch(j):
            k = j % S
            st = self.streams[k]
            with torch.cuda.stream(st):
                self.pipes[k].enqueue(st.cuda_stream)
                self._pinned[j % NS].copy_(self.pipes[k].results, non_blocking=True)
                self._events[j % NS].record(st)
            inflight.append((j, list(pending)))
            pending.clear()

        j, n = 0, 0
        try:
            for idx, frame, addr in source:
                if frame.shape != (self.H, self.W, 3):
                    source.release(idx)
                    raise ValueError("frame %d is %s, pipeline was built for %s" % (idx, frame.shape, (self.H, self.W, 3)))
                k = j % S
                st = self.streams[k]
                with torch.cuda.stream(st):
                    _lib.check(L.bp_upload(self.pipes[k].frames.data_ptr() + len(pending) * nbytes, addr, nbytes, st.cuda_stream))
                pending.append(idx)
                n += 1
                if len(pending) == B:
                    launch(j)
                    j += 1
                    if len(inflight) > S:      # (the stream's own order keeps a launch's frame slots safe from the next upload)
                        finish()
            if pending:                        # ragged last launch: the unused slots keep their previous frames, whose
                launch(j)                      # records nobody reads
                j += 1
            while inflight:
                finish()
        finally:
            # an error mid-stream (a broken frame, a failed launch): let the device drain, then hand the loader its
            # slots back so it can be closed or iterated further
            if inflight or pending:
                for st in self.streams:
                    st.synchronize()
                for idx in [i for _, idxs in inflight for i in idxs] + pending:
                    try:
                        source.release(idx)
                    except Exception:
                        pass
                inflight.clear()
                pending.clear()
        return n


class MultiObjectRunner:
    """Occlusion-LineMod as a multi-object workload (occlusion_betapose_evaluate.py:89-90,204,218-257 runs ONE object
    per process and re-decodes every frame per object): the unit of work is a (frame, object) pair.  Every
    object keeps its own detector + key-point weights resident; a frame is decoded ONCE (loader slot), and each of its
    units uploads it from that slot to the stream it lands on and runs that object's graph.  ``streams`` units are in
    flight; unit u = frame_position * n_objects + object_position, and only the units in ``owned`` are run (the caller
    shards them ``u % world``).

    ``engines``: {obj_id: (Darknet, FastPoseHIP)} for the objects this rank owns units of."""

    def __init__(self, engines: dict, obj_ids: List[int], frame_h: int = 480, frame_w: int = 640, streams: int = 4,
                 confidence: float = 0.01, num_classes: int = 80, use_graph: bool = True):
        import torch
        self.obj_ids = list(obj_ids)
        S = max(1, int(streams))
        self.S, self.H, self.W = S, int(frame_h), int(frame_w)
        self.pipes = {}            # (stream, obj_id) -> FramePipeline over the stream's shared frame buffer
        dev = None
        self.frame_bufs = []
        for k in range(S):
            buf = None
            for oid, (det, pose) in engines.items():
                pose = getattr(pose, "pyranet", pose)
                d, p_ = (det, pose) if k == 0 else (det.clone(), pose.clone())
                fp = FramePipeline(d, p_, frame_h, frame_w, batch=1, confidence=confidence, num_classes=num_classes,
                                   use_graph=use_graph, frames=buf)
                buf = fp.frames
                dev = buf.device
                self.pipes[(k, oid)] = fp
            self.frame_bufs.append(buf)
        self.streams = [torch.cuda.Stream(device=dev) for _ in range(S)] if dev is not None else []
        self._pinned = [torch.empty((1, RESULT_FLOATS), dtype=torch.float32).pin_memory() for _ in range(2 * S)]
        self._events = [torch.cuda.Event() for _ in range(2 * S)]

    def run(self, source, frame_positions: List[int], owned, on_record) -> int:
        """``source``: FrameLoader over the frames this rank touches (in ``frame_positions`` order: position of each
        in the global frame list); ``owned(u)`` tells whether unit u belongs to this rank;
        ``on_record(u, rec[316])`` receives every owned unit's record.  Returns the number of units run."""
        import torch
        L = _lib.lib()
        S, NS, nbytes, K = self.S, 2 * self.S, self.H * self.W * 3, len(self.obj_ids)
        inflight = []              # (sequence number, unit, loader index or None when this is not the frame's last unit)
        pending = {}               # loader index -> units still in flight

        def finish():
            j, u, idx = inflight.pop(0)
            self._events[j % NS].synchronize()
            rec = self._pinned[j % NS].numpy()[0].copy()
            pending[idx] -= 1
            if pending[idx] == 0:
                del pending[idx]
                source.release(idx)
            on_record(u, rec)

        j = 0
        try:
            for idx, frame, addr in source:
                if frame.shape != (self.H, self.W, 3):
                    source.release(idx)
                    raise ValueError("frame %d is %s, pipeline was built for %s" % (idx, frame.shape, (self.H, self.W, 3)))
                units = [frame_positions[idx] * K + oi for oi in range(K) if owned(frame_positions[idx] * K + oi)]
                if not units:
                    source.release(idx)
                    continue
                pending[idx] = len(units)
                for u in units:
                    k = j % S
                    st = self.streams[k]
                    # the stream's frame buffer is only rewritten after the stream's previous unit finished reading it
                    # (same stream: in order)
                    with torch.cuda.stream(st):
                        _lib.check(L.bp_upload(self.frame_bufs[k].data_ptr(), addr, nbytes, st.cuda_stream))
                        self.pipes[(k, self.obj_ids[u % K])].enqueue(st.cuda_stream)
                        self._pinned[j % NS].copy_(self.pipes[(k, self.obj_ids[u % K])].results, non_blocking=True)
                        self._events[j % NS].record(st)
                    inflight.append((j, u, idx))
                    j += 1
                    if len(inflight) > S:
                        finish()
            while inflight:
                finish()
        finally:
            if inflight:
                for st in self.streams:
                    st.synchronize()
                for idx in list(pending):
                    try:
                        source.release(idx)
                    except Exception:
                        pass
                inflight.clear()
        return j


def finish_record(rec: np.ndarray, imgname: str, kp_3d: np.ndarray, cam_K: np.ndarray, left_number: int = 50) -> dict:
    """Host tail for one frame: 316-float record -> the dict ``DataWriter.update`` appends to
    ``final_result`` (dataloader.py:704-727): {'imgname', 'result', 'cam_R', 'cam_t'} (+ the raw boxes)."""
    rec = np.ascontiguousarray(rec, dtype=np.float32)
    idx = int(rec[:1].view(np.int32)[0])
    if idx < 0:     # no detection: the reference forwards the frame with boxes=None and records nothing
        return {"imgname": imgname, "result": [], "cam_R": [], "cam_t": [], "boxes": None}
    boxes = rec[12:16].reshape(1, 4).copy()
    scores = rec[5:6].reshape(1, 1).copy()
    pt1, pt2 = rec[8:10].reshape(1, 2), rec[10:12].reshape(1, 2)
    kp = rec[16:].reshape(1, 50, 6)
    _, preds_img, preds_scores = decode_keypoints(kp, pt1, pt2)
    result = pose_nms(boxes, scores, preds_img, preds_scores)
    out = {"imgname": imgname, "result": result, "boxes": boxes, "scores": scores, "yolo_index": idx}
    if result:
        kp_score = np.array(result[0]["kp_score"][:, 0])
        kp_2d = np.array(result[0]["keypoints"])
        k3 = np.array(kp_3d)
        while len(kp_2d) > left_number:          # dataloader.py:718-722
            d = int(np.argmin(kp_score, axis=0))
            kp_score = np.delete(kp_score, d)
            kp_2d = np.delete(kp_2d, d, axis=0)
            k3 = np.delete(k3, d, axis=0)
        R, t = solve_pnp(k3, kp_2d, cam_K)
        out.update({"cam_R": R, "cam_t": t})
    else:
        out.update({"cam_R": [], "cam_t": []})
    return out
