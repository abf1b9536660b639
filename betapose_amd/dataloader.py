"""Pipeline stages with the reference's class names, constructor arguments, methods and queue tuples
(3_6Dpose_estimator/dataloader.py:90-190,285-466,649-763), so ``betapose_evaluate.py`` runs with only its imports
changed.  Threads + bounded queues as under the reference's ``--sp``; the models are the HIP engines."""
from __future__ import annotations

import os
import time
from queue import Queue
from threading import Thread

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


class DetectionLoader:
    """dataloader.py:285-409.  ``det_model`` may be passed in (already loaded); otherwise the reference's hard-coded
    paths are used (cfg ``yolo/cfg/yolov3-single.cfg``, weights ``models/yolo/%02d.weights``)."""

    def __init__(self, dataloder, obj_id, batchSize=1, queueSize=1024, det_model=None):
        if det_model is None:
            cfg_path = "yolo/cfg/yolov3-single.cfg"
            weights_path = 'models/yolo/{:02d}.weights'.format(obj_id)
            det_model = Darknet(cfg_path, reso=int(opt.inp_dim), max_batch=batchSize)
            det_model.load_weights(weights_path)
            print("Loading YOLO cfg from", cfg_path)
            print("Loading YOLO weights from", weights_path)
        self.det_model = det_model
        self.det_model.net_info['height'] = opt.inp_dim
        self.det_inp_dim = int(self.det_model.net_info['height'])
        assert self.det_inp_dim % 32 == 0
        assert self.det_inp_dim > 32
        self.det_model.cuda()
        self.det_model.eval()
        self.stopped = False
        self.dataloder = dataloder
        self.batchSize = batchSize
        self.datalen = self.dataloder.length()
        self.num_batches = (self.datalen + batchSize - 1) // batchSize
        self.Q = Queue(maxsize=queueSize)

    def start(self):
        Thread(target=self.update, args=(), daemon=True).start()
        return self

    def update(self):
        import torch
        for i in range(self.num_batches):
            img, orig_img, im_name, im_dim_list = self.dataloder.getitem()
            if img is None:
                self.Q.put((None, None, None, None, None, None, None))
                return
            prediction = self.det_model(img.cuda())
            dets = dynamic_write_results(prediction, opt.confidence, opt.num_classes, nms=True, nms_conf=opt.nms_thesh)
            if isinstance(dets, int) or dets.shape[0] == 0:
                for k in range(len(orig_img)):
                    self.Q.put((orig_img[k], im_name[k], None, None, None, None, None))
                continue
            dets = dets.cpu()
            reso = self.det_inp_dim
            dims = torch.index_select(im_dim_list, 0, dets[:, 0].long())
            w_ratio, h_ratio = dims[:, 0] / reso, dims[:, 1] / reso
            boxes = dets[:, 1:5]
            boxes[:, 0] = boxes[:, 0] * w_ratio
            boxes[:, 1] = boxes[:, 1] * h_ratio
            boxes[:, 2] = boxes[:, 2] * w_ratio
            boxes[:, 3] = boxes[:, 3] * h_ratio
            scores = dets[:, 5:6]
            for k in range(len(orig_img)):
                boxes_k = boxes[dets[:, 0] == k]
                if boxes_k.shape[0] == 0:
                    self.Q.put((orig_img[k], im_name[k], None, None, None, None, None))
                    continue
                inps = torch.zeros(boxes_k.size(0), 3, opt.inputResH, opt.inputResW)
                pt1 = torch.zeros(boxes_k.size(0), 2)
                pt2 = torch.zeros(boxes_k.size(0), 2)
                self.Q.put((orig_img[k], im_name[k], boxes_k, scores[dets[:, 0] == k], inps, pt1, pt2))

    def read(self):
        return self.Q.get()

    def len(self):
        return self.Q.qsize()


class DetectionProcessor:
    """dataloader.py:412-465: BGR->RGB, /255, mean subtraction, box padding, crop + bilinear resize -- one HIP kernel."""

    def __init__(self, detectionLoader, queueSize=1024):
        self.detectionLoader = detectionLoader
        self.stopped = False
        self.datalen = self.detectionLoader.datalen
        self.Q = Queue(maxsize=queueSize)

    def start(self):
        Thread(target=self.update, args=(), daemon=True).start()
        return self

    def update(self):
        for i in range(self.datalen):
            (orig_img, im_name, boxes, scores, inps, pt1, pt2) = self.detectionLoader.read()
            if orig_img is None:
                self.Q.put((None, None, None, None, None, None, None))
                return
            if boxes is None or boxes.nelement() == 0:
                self.Q.put((None, orig_img, im_name, boxes, scores, None, None))
                continue
            inps, pt1, pt2 = crop_from_dets_frame(orig_img, boxes, opt.inputResH, opt.inputResW)
            self.Q.put((inps, orig_img, im_name, boxes, scores, pt1, pt2))

    def read(self):
        return self.Q.get()

    def len(self):
        return self.Q.qsize()


class DataWriter:
    """dataloader.py:649-763: heat-maps -> key points -> pPose-NMS -> key-point pruning -> PnP."""

    def __init__(self, cam_K, left_number, kp_model_vertices, save_video=False, savepath='examples/res/1.avi',
                 fourcc=0, fps=25, frameSize=(640, 480), queueSize=1024):
        # same positional signature as the reference (dataloader.py:650-653).  save_video: the annotated frames go to a
        # Motion-JPEG .avi (video.MJPEGWriter stands in for cv2.VideoWriter; `fourcc` is accepted and ignored)
        self.save_video = bool(save_video)
        if self.save_video:
            from .video import MJPEGWriter
            self.stream = MJPEGWriter(savepath, fps, frameSize)
        self.stopped = False
        self.final_result = []
        self.Q = Queue(maxsize=queueSize)
        self.kp_3d = kp_model_vertices
        self.cam_K = cam_K
        self.left_number = left_number
        self._busy = False

    def start(self):
        Thread(target=self.update, args=(), daemon=True).start()
        return self

    def update(self):
        while True:
            if self.stopped:
                return
            if self.Q.empty():
                time.sleep(0.001)
                continue
            self._busy = True
            (boxes, scores, hm_data, pt1, pt2, orig_img, im_name) = self.Q.get()
            if boxes is not None:
                preds_hm, preds_img, preds_scores = getPrediction(hm_data, pt1, pt2, opt.inputResH, opt.inputResW,
                                                                  opt.outputResH, opt.outputResW)
                result = {'imgname': im_name, 'result': pose_nms(boxes, scores, preds_img, preds_scores)}
                if result['result']:
                    kp_score = np.array(result['result'][0]['kp_score'][:, 0])
                    kp_2d = np.array(result['result'][0]['keypoints'])
                    kp_3d = np.array(self.kp_3d)
                    while len(kp_2d) > self.left_number:
                        d = int(np.argmin(kp_score, axis=0))
                        kp_score = np.delete(kp_score, d)
                        kp_2d = np.delete(kp_2d, d, axis=0)
                        kp_3d = np.delete(kp_3d, d, axis=0)
                    R, t = solve_pnp(kp_3d, kp_2d, self.cam_K)
                    result.update({'cam_R': R, 'cam_t': t})
                else:
                    result.update({'cam_R': [], 'cam_t': []})
                self.final_result.append(result)
                if self.save_video:
                    from .video import vis_frame
                    self.stream.write(vis_frame(orig_img, result))
            elif self.save_video and orig_img is not None:
                self.stream.write(np.asarray(orig_img))
            self._busy = False

    def running(self):
        # the reference only tests Q.empty() (racy: the last item may still be in flight, dataloader.py:743-746)
        time.sleep(0.002)
        return (not self.Q.empty()) or self._busy

    def save(self, boxes, scores, hm_data, pt1, pt2, orig_img, im_name):
        self._busy = True
        self.Q.put((boxes, scores, hm_data, pt1, pt2, orig_img, im_name))

    def stop(self):
        self.stopped = True
        time.sleep(0.01)
        if self.save_video:
            self.stream.release()

    def results(self):
        return self.final_result

    def len(self):
        return self.Q.qsize()


class Mscoco:
    """dataloader.py:765-791 -- the (unused at inference) dataset stub InferenNet_fast takes."""

    def __init__(self, train=True, sigma=1, scale_factor=(0.2, 0.3), rot_factor=40, label_type='Gaussian'):
        self.inputResH, self.inputResW = opt.inputResH, opt.inputResW
        self.outputResH, self.outputResW = opt.outputResH, opt.outputResW
        self.nJoints = 50
        self.accIdxs = tuple(range(1, 51))
        self.flipRef = ()
