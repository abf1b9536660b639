"""Pipeline stages with the reference's class names, constructor arguments, methods and queue tuples
(3_6Dpose_estimator/dataloader.py:90-190,285-466,649-763), so ``betapose_evaluate.py`` runs with only its imports
changed.  Threads + bounded queues as under the reference's ``--sp``; the models are the HIP engines."""
from __future__ import annotations

import os
import time
from queue import Queue
from threading import Lock, Thread

import numpy as np

from .darknet import Darknet
from .eval import getPrediction
from .img import crop_from_dets, crop_from_dets_frame, load_frame_bgr  # noqa: F401  (crop_from_dets: reference name)
from .ops import solve_pnp
from .opt import opt
from .pPose_nms import pose_nms
from .yolo_util import dynamic_write_results


class ImageLoader:
    """dataloader.py:90-189 (format 'yolo'): per image the BGR u8 frame, the bicubic-stretched RGB tensor the detector
    sees (Pillow, exactly as torchvision's Resize(interpolation=3) + ToTensor) and (w, h, w, h)."""

    def __init__(self, im_names, batchSize=1, format='yolo', queueSize=50, reso=608):
        if format != 'yolo':
            raise NotImplementedError(format)
        from .dist import limit_host_threads
        limit_host_threads()      # stage threads + a per-core intra-op pool collapse on many-core hosts
        self.img_dir = opt.inputpath
        self.imglist = im_names
        self.reso = int(reso)
        self.batchSize = batchSize
        self.datalen = len(self.imglist)
        self.num_batches = (self.datalen + batchSize - 1) // batchSize
        self.Q = Queue(maxsize=queueSize)

    def start(self):
        p = Thread(target=self.getitem_yolo, args=(), daemon=True)
        p.start()
        return self

    def _load_one(self, k):
        """One frame: (tensor [1,3,reso,reso] RGB 0..1, BGR u8 frame, path, (w, h))."""
        import torch
        from PIL import Image
        name = os.path.join(self.img_dir, self.imglist[k].rstrip('\n').rstrip('\r'))
        orig = load_frame_bgr(name)
        pil = Image.fromarray(np.ascontiguousarray(orig[:, :, ::-1])).resize((self.reso, self.reso), 3)
        t = torch.from_numpy(np.asarray(pil, dtype=np.uint8).transpose(2, 0, 1).copy()).float().div(255)
        return t.unsqueeze(0), orig, name, (orig.shape[1], orig.shape[0])

    def getitem_yolo(self):
        # the reference decodes and resizes on this one thread; decode and Pillow's resize release the GIL, so a few
        # workers run ahead (bounded), results are consumed strictly in list order
        import torch
        from collections import deque
        from concurrent.futures import ThreadPoolExecutor
        ahead = max(2 * self.batchSize, 8)
        try:
            self._produce(ahead, torch, deque, ThreadPoolExecutor)
        except BaseException as e:   # an unreadable frame must not leave the consumers blocked on an empty queue
            self.Q.put(e)

    def _produce(self, ahead, torch, deque, ThreadPoolExecutor):
        with ThreadPoolExecutor(max_workers=4) as pool:
            pending, nxt = deque(), 0
            for i in range(self.num_batches):
                lo, hi = i * self.batchSize, min((i + 1) * self.batchSize, self.datalen)
                while nxt < self.datalen and nxt < hi + ahead:
                    pending.append(pool.submit(self._load_one, nxt))
                    nxt += 1
                items = [pending.popleft().result() for _ in range(lo, hi)]
                img = torch.cat([it[0] for it in items])
                orig_img = [it[1] for it in items]
                im_name = [it[2] for it in items]
                im_dim_list = torch.FloatTensor([it[3] for it in items]).repeat(1, 2)
                self.Q.put((img, orig_img, im_name, im_dim_list))

    def getitem_ssd(self):
        raise NotImplementedError("the SSD input format (dataloader.py:111-148) is not part of this path; use format='yolo'")

    def getitem(self):
        item = self.Q.get()
        if isinstance(item, BaseException):
            self.Q.put(item)          # every later reader sees it too
            raise RuntimeError("ImageLoader: frame input failed: %s" % item) from item
        return item

    def length(self):
        return len(self.imglist)

    def len(self):
        return self.Q.qsize()


class _Stage:
    """A worker thread feeding a bounded queue: what every stage of the reference's ``--sp`` pipeline is.  Subclasses
    implement ``update`` (the thread body); ``start`` / ``read`` / ``len`` are the reference's method names."""

    def __init__(self, queueSize):
        self.Q = Queue(maxsize=queueSize)
        self.stopped = False

    def start(self):
        Thread(target=self.update, daemon=True).start()
        return self

    def read(self):
        return self.Q.get()

    def len(self):
        return self.Q.qsize()

    def _emit(self, *fields):
        self.Q.put(tuple(fields))


_EMPTY7 = (None,) * 7


def _default_detector(obj_id, batchSize):
    """The detector the reference builds when none is handed in: its hard-coded cfg and per-object weights
    (dataloader.py:287-293)."""
    paths = {"cfg": "yolo/cfg/yolov3-single.cfg", "weights": "models/yolo/%02d.weights" % obj_id}
    model = Darknet(paths["cfg"], reso=int(opt.inp_dim), max_batch=batchSize)
    model.load_weights(paths["weights"])
    for what in ("cfg", "weights"):
        print("Loading YOLO %s from" % what, paths[what])
    return model


class DetectionLoader(_Stage):
    """dataloader.py:285-409: detector + box selection + rescale to frame pixels; per frame one
    (orig_img, im_name, boxes, scores, inps, pt1, pt2) tuple, the last three as zero place-holders of the crop stage's
    shapes.  ``det_model`` may be passed in (already loaded)."""

    def __init__(self, dataloder, obj_id, batchSize=1, queueSize=1024, det_model=None):
        super().__init__(queueSize)
        self.det_model = det_model if det_model is not None else _default_detector(obj_id, batchSize)
        self.det_model.net_info['height'] = opt.inp_dim
        self.det_inp_dim = reso = int(opt.inp_dim)
        if reso % 32 != 0 or reso <= 32:
            raise AssertionError("detector input size must be a multiple of 32 and larger than 32, got %d" % reso)
        self.det_model.cuda().eval()
        self.dataloder = dataloder
        self.batchSize = batchSize
        self.datalen = dataloder.length()
        self.num_batches = -(-self.datalen // batchSize)

    def _frame_boxes(self, dets, im_dim_list):
        """Detections [n, 8] (frame index first) in detector-input pixels -> boxes in frame pixels and their scores."""
        import torch
        scale = im_dim_list.index_select(0, dets[:, 0].long()) / float(self.det_inp_dim)     # (w, h, w, h) / reso per box
        return dets[:, 1:5] * scale, dets[:, 5:6]

    def update(self):
        import torch
        for _ in range(self.num_batches):
            img, orig_img, im_name, im_dim_list = self.dataloder.getitem()
            if img is None:
                self._emit(*_EMPTY7)
                return
            dets = dynamic_write_results(self.det_model(img.cuda()), opt.confidence, opt.num_classes, nms=True,
                                         nms_conf=opt.nms_thesh)
            found = not isinstance(dets, int) and dets.shape[0] > 0
            if found:
                dets = dets.cpu()
                boxes, scores = self._frame_boxes(dets, im_dim_list)
                owner = dets[:, 0]
            for k, (frame, name) in enumerate(zip(orig_img, im_name)):
                mine = (owner == k) if found else None
                n = int(mine.sum()) if found else 0
                if n == 0:
                    self._emit(frame, name, None, None, None, None, None)
                    continue
                self._emit(frame, name, boxes[mine], scores[mine], torch.zeros(n, 3, opt.inputResH, opt.inputResW),
                           torch.zeros(n, 2), torch.zeros(n, 2))


class DetectionProcessor(_Stage):
    """dataloader.py:412-465: BGR->RGB, /255, mean subtraction, box padding, crop + bilinear resize -- one HIP kernel.
    Output tuples: (inps, orig_img, im_name, boxes, scores, pt1, pt2)."""

    def __init__(self, detectionLoader, queueSize=1024):
        super().__init__(queueSize)
        self.detectionLoader = detectionLoader
        self.datalen = detectionLoader.datalen

    def update(self):
        for _ in range(self.datalen):
            frame, name, boxes, scores = self.detectionLoader.read()[:4]
            if frame is None:
                self._emit(*_EMPTY7)
                return
            if boxes is None or boxes.nelement() == 0:
                self._emit(None, frame, name, boxes, scores, None, None)
            else:
                inps, pt1, pt2 = crop_from_dets_frame(frame, boxes, opt.inputResH, opt.inputResW)
                self._emit(inps, frame, name, boxes, scores, pt1, pt2)


def keep_best_keypoints(kp_2d, kp_score, kp_3d, left_number):
    """The reference drops the lowest-scoring key point, one at a time, until ``left_number`` remain
    (dataloader.py:703-712).  Removing the current minimum repeatedly = removing the (n - left_number) smallest in
    first-occurrence order on ties, which a stable sort gives in one step; survivors keep their original order."""
    kp_2d, kp_score, kp_3d = np.asarray(kp_2d), np.asarray(kp_score), np.asarray(kp_3d)
    drop = len(kp_2d) - int(left_number)
    if drop <= 0:
        return kp_2d, kp_score, kp_3d
    keep = np.sort(np.argsort(kp_score, kind="stable")[drop:])
    return kp_2d[keep], kp_score[keep], kp_3d[keep]


class DataWriter(_Stage):
    """dataloader.py:649-763: heat-maps -> key points -> pPose-NMS -> key-point pruning -> PnP."""

    def __init__(self, cam_K, left_number, kp_model_vertices, save_video=False, savepath='examples/res/1.avi',
                 fourcc=0, fps=25, frameSize=(640, 480), queueSize=1024):
        # same positional signature as the reference (dataloader.py:650-653).  save_video: the annotated frames go to a
        # Motion-JPEG .avi (video.MJPEGWriter stands in for cv2.VideoWriter; `fourcc` is accepted and ignored)
        super().__init__(queueSize)
        self.save_video = bool(save_video)
        self.stream = None
        if self.save_video:
            from .video import MJPEGWriter
            self.stream = MJPEGWriter(savepath, fps, frameSize)
        self.final_result = []
        self.kp_3d = kp_model_vertices
        self.cam_K = cam_K
        self.left_number = left_number
        self._pending = 0                 # items handed to save() and not yet fully processed by the writer thread
        self._pending_lock = Lock()
        self._thread = None
        self.error = None                 # first exception of the writer thread (re-raised by stop() / results())

    def start(self):
        self._thread = Thread(target=self.update, daemon=True)
        self._thread.start()
        return self

    def _pose_of(self, boxes, scores, hm_data, pt1, pt2, im_name):
        """One frame's result dict: key points from the heat-maps, pPose-NMS, the best ``left_number`` points into PnP."""
        _, preds_img, preds_scores = getPrediction(hm_data, pt1, pt2, opt.inputResH, opt.inputResW, opt.outputResH,
                                                   opt.outputResW)
        poses = pose_nms(boxes, scores, preds_img, preds_scores)
        result = {'imgname': im_name, 'result': poses, 'cam_R': [], 'cam_t': []}
        if poses:
            best = poses[0]
            kp_2d, _, kp_3d = keep_best_keypoints(best['keypoints'], np.asarray(best['kp_score'])[:, 0], self.kp_3d,
                                                  self.left_number)
            result['cam_R'], result['cam_t'] = solve_pnp(kp_3d, kp_2d, self.cam_K)
        return result

    def update(self):
        from queue import Empty
        while not self.stopped:
            try:
                boxes, scores, hm_data, pt1, pt2, orig_img, im_name = self.Q.get(timeout=0.001)
            except Empty:
                continue
            # the item always leaves the pending count, whatever happens to it: an exception in PnP / NMS / the video stream must
            # not leave running() true forever with callers spinning on it (round-4 advisor finding); the first one is kept and
            # re-raised by stop() / results()
            try:
                frame_out = None
                if boxes is not None:
                    result = self._pose_of(boxes, scores, hm_data, pt1, pt2, im_name)
                    self.final_result.append(result)
                    if self.save_video:
                        from .video import vis_frame
                        frame_out = vis_frame(orig_img, result)
                elif self.save_video and orig_img is not None:
                    frame_out = np.asarray(orig_img)
                if frame_out is not None:
                    self.stream.write(frame_out)
            except Exception as exc:          # noqa: BLE001 -- forwarded to the caller's thread
                if self.error is None:
                    self.error = exc
            finally:
                with self._pending_lock:
                    self._pending -= 1

    def running(self):
        # the reference only tests Q.empty() (racy: the last item may still be in flight, dataloader.py:743-746)
        time.sleep(0.002)
        if self._thread is not None and not self._thread.is_alive():
            return False                      # a dead writer will never drain the queue
        with self._pending_lock:
            return self._pending > 0

    def save(self, boxes, scores, hm_data, pt1, pt2, orig_img, im_name):
        with self._pending_lock:
            self._pending += 1            # counted before the item is visible to the writer: running() can never miss it
        self.Q.put((boxes, scores, hm_data, pt1, pt2, orig_img, im_name))

    def stop(self):
        self.stopped = True
        if self._thread is not None:
            self._thread.join()                 # the writer finishes the item it holds (PnP, stream.write) before the stream is released
        if self.stream is not None:
            self.stream.release()
        self._raise_pending_error()

    def _raise_pending_error(self):
        if self.error is not None:
            exc, self.error = self.error, None
            raise exc

    def results(self):
        self._raise_pending_error()
        return self.final_result


class Mscoco:
    """dataloader.py:765-791 -- the (unused at inference) dataset stub InferenNet_fast takes."""

    def __init__(self, train=True, sigma=1, scale_factor=(0.2, 0.3), rot_factor=40, label_type='Gaussian'):
        self.inputResH, self.inputResW = opt.inputResH, opt.inputResW
        self.outputResH, self.outputResW = opt.outputResH, opt.outputResW
        self.nJoints = 50
        self.accIdxs = tuple(range(1, 51))
        self.flipRef = ()
