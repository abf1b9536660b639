"""Video / webcam input and result visualisation with the reference's class surface:
``VideoLoader`` (dataloader.py:192-282), ``VideoDetectionLoader`` (:468-591), ``WebcamLoader`` (:594-647),
``letterbox_image`` / ``prep_frame`` (yolo/preprocess.py:18-60) and ``vis_frame`` (fn.py:88-220, commented out there).

The reference reads video through ``cv2.VideoCapture``; OpenCV (and any other codec library) is not part of this
image, so ``FrameSource`` stands in for it with the same ``read() -> (grabbed, frame_bgr)`` contract over what can be
decoded here: a directory / list of frame images, or a Motion-JPEG stream (RIFF ``.avi`` with MJPG chunks, or
concatenated JPEGs) decoded with Pillow.  Compressed inter-frame codecs (H.264, MPEG-4) raise a clear error.
``letterbox_image`` restates ``cv2.resize(..., INTER_CUBIC)`` for 8-bit images (4-tap, a = -0.75, 11-bit fixed-point
coefficients, replicated border, no antialiasing -- OpenCV imgproc/resize.cpp): parity against a real cv2 is NOT pinned.
None of these classes is used by the reference's 6D evaluation scripts; they are here so that a user of its loaders
finds the same names.  The detector / key-point engines behind them are the HIP ones.
"""
from __future__ import annotations

import io
import os
import struct
import sys
import time
from queue import LifoQueue, Queue
from threading import Thread
from typing import List, Optional, Tuple

import numpy as np

from .img import im_to_torch, load_frame_bgr
from .opt import opt


# ----------------------------------------------------------------------------------------------- frame sources
class FrameSource:
    """``cv2.VideoCapture`` stand-in: ``isOpened()``, ``read()``, ``frame_count``, ``fps``, ``frame_size`` (w, h)."""

    def __init__(self, path):
        self.path = path
        self._frames: Optional[List[str]] = None      # image sequence
        self._jpegs: Optional[List[Tuple[int, int]]] = None   # (offset, length) into the stream file
        self._fh = None
        self._pos = 0
        self.fps = 25.0
        self.fourcc = 0
        if isinstance(path, (list, tuple)):
            self._frames = list(path)
        elif os.path.isdir(str(path)):
            names = sorted(f for f in os.listdir(path) if f.lower().endswith((".png", ".jpg", ".jpeg", ".bmp")))
            self._frames = [os.path.join(path, f) for f in names]
        elif os.path.isfile(str(path)):
            self._index_stream(str(path))
        else:
            raise IOError("Cannot capture source %r (no such file or directory; camera devices need a capture library "
                          "this image does not have)" % (path,))
        self.frame_count = len(self._frames) if self._frames is not None else len(self._jpegs)
        self.frame_size = (0, 0)
        if self.frame_count:
            ok, f0 = self._decode(0)
            self.frame_size = (f0.shape[1], f0.shape[0])

    def _index_stream(self, path):
        data = open(path, "rb").read()
        if data[:4] == b"RIFF" and data[8:12] == b"AVI ":
            self._jpegs = []
            pos = 12
            end = len(data)

            def walk(lo, hi):
                p = lo
                while p + 8 <= hi:
                    cid, sz = data[p:p + 4], struct.unpack("<I", data[p + 4:p + 8])[0]
                    if cid == b"LIST":
                        walk(p + 12, min(hi, p + 8 + sz))
                    elif cid == b"avih" and sz >= 4:
                        us = struct.unpack("<I", data[p + 8:p + 12])[0]
                        if us:
                            self.fps = 1e6 / us
                    elif cid == b"strh" and sz >= 8 and data[p + 8:p + 12] == b"vids":
                        self.fourcc = struct.unpack("<I", data[p + 12:p + 16])[0]
                        if data[p + 12:p + 16].upper() not in (b"MJPG", b"JPEG"):
                            raise IOError("AVI video stream is %r: only Motion-JPEG can be decoded without a codec "
                                          "library" % data[p + 12:p + 16])
                    elif cid[2:] in (b"dc", b"db") and sz > 2 and data[p + 8:p + 10] == b"\xff\xd8":
                        self._jpegs.append((p + 8, sz))
                    p += 8 + sz + (sz & 1)
            walk(pos, end)
        elif data[:2] == b"\xff\xd8":
            # concatenated JPEGs: frames start at SOI and end at the matching EOI
            self._jpegs = []
            p = 0
            while True:
                s = data.find(b"\xff\xd8\xff", p)
                if s < 0:
                    break
                e = data.find(b"\xff\xd9", s + 2)
                if e < 0:
                    break
                self._jpegs.append((s, e + 2 - s))
                p = e + 2
        else:
            raise IOError("%s: not a Motion-JPEG stream (inter-frame codecs need a codec library this image lacks)" % path)
        self._fh = open(path, "rb")

    def _decode(self, i):
        if self._frames is not None:
            return True, load_frame_bgr(self._frames[i])
        from PIL import Image
        off, ln = self._jpegs[i]
        self._fh.seek(off)
        img = Image.open(io.BytesIO(self._fh.read(ln))).convert("RGB")
        return True, np.ascontiguousarray(np.asarray(img)[:, :, ::-1])

    def isOpened(self):
        return self.frame_count > 0

    def read(self):
        if self._pos >= self.frame_count:
            return False, None
        ok, f = self._decode(self._pos)
        self._pos += 1
        return ok, f

    def release(self):
        if self._fh is not None:
            self._fh.close()
            self._fh = None


class MJPEGWriter:
    """``cv2.VideoWriter`` stand-in for ``DataWriter(save_video=True)``: a Motion-JPEG ``.avi`` (one 00dc chunk per
    frame, idx1 index) that ``FrameSource`` and ordinary players read back."""

    def __init__(self, path, fps=25, frame_size=(640, 480), quality=90):
        self.path, self.fps, self.size, self.quality = path, float(fps), tuple(frame_size), quality
        self._chunks: List[bytes] = []

    def isOpened(self):
        return True

    def write(self, frame_bgr):
        from PIL import Image
        buf = io.BytesIO()
        Image.fromarray(np.ascontiguousarray(frame_bgr[:, :, ::-1])).save(buf, format="JPEG", quality=self.quality)
        self._chunks.append(buf.getvalue())

    def release(self):
        w, h = self.size
        n = len(self._chunks)
        movi = b"".join(b"00dc" + struct.pack("<I", len(c)) + c + (b"\0" if len(c) & 1 else b"") for c in self._chunks)
        idx, off = b"", 4
        for c in self._chunks:
            idx += b"00dc" + struct.pack("<III", 0x10, off, len(c))
            off += 8 + len(c) + (len(c) & 1)
        avih = struct.pack("<IIIIIIIIII4I", int(1e6 / self.fps), 0, 0, 0x10, n, 0, 1, 0, w, h, 0, 0, 0, 0)
        strh = b"vids" + b"MJPG" + struct.pack("<IHHIIIIIIII4H", 0, 0, 0, 0, 1, int(self.fps), 0, n, 0, 0xFFFFFFFF, 0, 0, 0, w, h)
        strf = struct.pack("<IiiHH4sIiiII", 40, w, h, 1, 24, b"MJPG", w * h * 3, 0, 0, 0, 0)

        def chunk(cid, body):
            return cid + struct.pack("<I", len(body)) + body + (b"\0" if len(body) & 1 else b"")

        def lst(kind, body):
            return b"LIST" + struct.pack("<I", len(body) + 4) + kind + body
        hdrl = lst(b"hdrl", chunk(b"avih", avih) + lst(b"strl", chunk(b"strh", strh) + chunk(b"strf", strf)))
        body = b"AVI " + hdrl + lst(b"movi", movi) + chunk(b"idx1", idx)
        os.makedirs(os.path.dirname(os.path.abspath(self.path)), exist_ok=True)
        with open(self.path, "wb") as f:
            f.write(b"RIFF" + struct.pack("<I", len(body)) + body)


# ----------------------------------------------------------------------------------------------- preprocessing
def _cubic_coeffs(x, A=-0.75):
    """OpenCV's interpolateCubic: weights of taps -1, 0, +1, +2 for fractional offset x."""
    c0 = ((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A
    c1 = ((A + 2) * x - (A + 3)) * x * x + 1
    c2 = ((A + 2) * (1 - x) - (A + 3)) * (1 - x) * (1 - x) + 1
    return np.stack([c0, c1, c2, 1.0 - c0 - c1 - c2], axis=-1)


def _resize_axis_tables(src, dst):
    scale = src / dst
    f = (np.arange(dst) + 0.5) * scale - 0.5
    s = np.floor(f).astype(np.int64)
    w = _cubic_coeffs((f - s).astype(np.float32).astype(np.float64))
    # 11-bit fixed-point coefficients (INTER_RESIZE_COEF_BITS), as cv2 rounds them to short
    wi = np.rint(w * 2048.0).astype(np.int64)
    idx = np.clip(s[:, None] + np.arange(-1, 3)[None, :], 0, src - 1)       # replicated border
    return idx, wi


def cv_resize_cubic(img_u8, new_w, new_h):
    """``cv2.resize(img, (new_w, new_h), interpolation=cv2.INTER_CUBIC)`` for u8 HWC, restated (module docstring)."""
    img = np.asarray(img_u8, dtype=np.int64)
    ix, wx = _resize_axis_tables(img.shape[1], new_w)
    iy, wy = _resize_axis_tables(img.shape[0], new_h)
    rows = (img[:, ix, :] * wx[None, :, :, None]).sum(axis=2)                       # horizontal pass, 11 fractional bits
    out = (rows[iy, :, :] * wy[:, :, None, None]).sum(axis=1)                       # vertical pass, 22 fractional bits
    return np.clip((out + (1 << 21)) >> 22, 0, 255).astype(np.uint8)


def letterbox_image(img, inp_dim):
    """yolo/preprocess.py:18-31: resize with unchanged aspect ratio, grey (128) padding."""
    img_w, img_h = img.shape[1], img.shape[0]
    w, h = inp_dim
    new_w = int(img_w * min(w / img_w, h / img_h))
    new_h = int(img_h * min(w / img_w, h / img_h))
    resized = cv_resize_cubic(img, new_w, new_h)
    canvas = np.full((inp_dim[1], inp_dim[0], 3), 128, dtype=np.int64)
    canvas[(h - new_h) // 2:(h - new_h) // 2 + new_h, (w - new_w) // 2:(w - new_w) // 2 + new_w, :] = resized
    return canvas


def prep_frame(img, inp_dim):
    """yolo/preprocess.py:47-60: frame (BGR u8) -> (tensor [1,3,D,D] RGB 0..1, the frame, (w, h))."""
    import torch
    orig_im = img
    dim = orig_im.shape[1], orig_im.shape[0]
    lb = letterbox_image(orig_im, (inp_dim, inp_dim))
    img_ = lb[:, :, ::-1].transpose((2, 0, 1)).copy()
    return torch.from_numpy(img_).float().div(255.0).unsqueeze(0), orig_im, dim


# ----------------------------------------------------------------------------------------------- loaders
class VideoLoader:
    """dataloader.py:192-282: batches of letterboxed frames from a video source."""

    def __init__(self, path, batchSize=1, queueSize=50):
        self.path = path
        self.stream = FrameSource(path)
        assert self.stream.isOpened(), 'Cannot capture source'
        self.stopped = False
        self.batchSize = batchSize
        self.datalen = int(self.stream.frame_count)
        self.num_batches = self.datalen // batchSize + (1 if self.datalen % batchSize else 0)
        self.Q = Queue(maxsize=queueSize)

    def length(self):
        return self.datalen

    def start(self):
        Thread(target=self.update, args=(), daemon=True).start()
        return self

    def update(self):
        import torch
        stream = FrameSource(self.path)
        assert stream.isOpened(), 'Cannot capture source'
        for i in range(self.num_batches):
            img, orig_img, im_name, im_dim_list = [], [], [], []
            for k in range(i * self.batchSize, min((i + 1) * self.batchSize, self.datalen)):
                grabbed, frame = stream.read()
                if not grabbed:
                    self.Q.put((None, None, None, None))
                    print('===========================> This video get ' + str(k) + ' frames in total.')
                    sys.stdout.flush()
                    return
                img_k, orig_img_k, im_dim_list_k = prep_frame(frame, int(opt.inp_dim))
                img.append(img_k)
                orig_img.append(orig_img_k)
                im_name.append(str(k) + '.jpg')
                im_dim_list.append(im_dim_list_k)
            self.Q.put((torch.cat(img), orig_img, im_name, torch.FloatTensor(im_dim_list).repeat(1, 2)))

    def videoinfo(self):
        return (self.stream.fourcc, self.stream.fps, self.stream.frame_size)

    def getitem(self):
        return self.Q.get()

    def len(self):
        return self.Q.qsize()


def _letterbox_boxes(dets, im_dim_list, det_inp_dim):
    """dataloader.py:548-560: undo the letterbox on detections (x1, y1, x2, y2 in columns 1..4)."""
    import torch
    im_dim_list = torch.index_select(im_dim_list, 0, dets[:, 0].long())
    scaling_factor = torch.min(det_inp_dim / im_dim_list, 1)[0].view(-1, 1)
    dets[:, [1, 3]] -= (det_inp_dim - scaling_factor * im_dim_list[:, 0].view(-1, 1)) / 2
    dets[:, [2, 4]] -= (det_inp_dim - scaling_factor * im_dim_list[:, 1].view(-1, 1)) / 2
    dets[:, 1:5] /= scaling_factor
    for j in range(dets.shape[0]):
        dets[j, [1, 3]] = torch.clamp(dets[j, [1, 3]], 0.0, float(im_dim_list[j, 0]))
        dets[j, [2, 4]] = torch.clamp(dets[j, [2, 4]], 0.0, float(im_dim_list[j, 1]))
    return dets


class VideoDetectionLoader:
    """dataloader.py:468-591: video frames -> detector -> per-frame (inp, orig_img, boxes, scores).  The reference
    hard-codes AlphaPose's person detector (yolov3-spp, NMS on); here the object detector of this path is passed in (or
    built from ``models/yolo/<obj>.weights``) and ``dynamic_write_results`` keeps its one box per frame."""

    def __init__(self, path, batchSize=4, queueSize=256, det_model=None, obj_id=None):
        from .darknet import Darknet
        from .yolo_util import dynamic_write_results
        self._write_results = dynamic_write_results
        if det_model is None:
            det_model = Darknet("yolo/cfg/yolov3-single.cfg", reso=int(opt.inp_dim), max_batch=batchSize)
            det_model.load_weights('models/yolo/{:02d}.weights'.format(int(obj_id if obj_id is not None else opt.obj_id)))
        self.det_model = det_model
        self.det_model.net_info['height'] = opt.inp_dim
        self.det_inp_dim = int(self.det_model.net_info['height'])
        assert self.det_inp_dim % 32 == 0
        assert self.det_inp_dim > 32
        self.det_model.cuda()
        self.det_model.eval()
        self.stream = FrameSource(path)
        assert self.stream.isOpened(), 'Cannot capture source'
        self.stopped = False
        self.batchSize = batchSize
        self.datalen = int(self.stream.frame_count)
        self.num_batches = self.datalen // batchSize + (1 if self.datalen % batchSize else 0)
        self.Q = Queue(maxsize=queueSize)

    def length(self):
        return self.datalen

    def len(self):
        return self.Q.qsize()

    def start(self):
        Thread(target=self.update, args=(), daemon=True).start()
        return self

    def update(self):
        import torch
        for i in range(self.num_batches):
            img, inp, orig_img, im_dim_list = [], [], [], []
            for k in range(i * self.batchSize, min((i + 1) * self.batchSize, self.datalen)):
                grabbed, frame = self.stream.read()
                if not grabbed:
                    self.stop()
                    return
                img_k, orig_img_k, im_dim_list_k = prep_frame(frame, int(opt.inp_dim))
                img.append(img_k)
                inp.append(im_to_torch(orig_img_k))
                orig_img.append(orig_img_k)
                im_dim_list.append(im_dim_list_k)
            with torch.no_grad():
                im_dims = torch.FloatTensor(im_dim_list).repeat(1, 2)
                prediction = self.det_model(torch.cat(img)).cpu()
                dets = self._write_results(prediction, opt.confidence, opt.num_classes, nms=True, nms_conf=opt.nms_thesh)
                if isinstance(dets, int) or dets.shape[0] == 0:
                    for k in range(len(inp)):
                        self.Q.put((inp[k], orig_img[k], None, None))
                    continue
                dets = _letterbox_boxes(dets.clone(), im_dims, self.det_inp_dim)
                boxes, scores = dets[:, 1:5], dets[:, 5:6]
            for k in range(len(inp)):
                self.Q.put((inp[k], orig_img[k], boxes[dets[:, 0] == k], scores[dets[:, 0] == k]))

    def videoinfo(self):
        return (self.stream.fourcc, self.stream.fps, self.stream.frame_size)

    def read(self):
        return self.Q.get()

    def more(self):
        return self.Q.qsize() > 0

    def stop(self):
        self.stopped = True


class WebcamLoader:
    """dataloader.py:594-647: newest-frame-first (LIFO) queue of letterboxed frames.  ``webcam``: a camera index in the
    reference; here anything ``FrameSource`` opens (a growing frame directory, an MJPEG stream) -- a bare index raises
    because no capture library is present."""

    def __init__(self, webcam, queueSize=256):
        if isinstance(webcam, int) or (isinstance(webcam, str) and webcam.isdigit()):
            raise IOError("Cannot capture source: camera index %s needs a capture library (V4L2 / OpenCV) this image "
                          "does not have; pass a frame directory or an MJPEG stream" % webcam)
        self.stream = FrameSource(webcam)
        assert self.stream.isOpened(), 'Cannot capture source'
        self.stopped = False
        self.Q = LifoQueue(maxsize=queueSize)

    def start(self):
        Thread(target=self.update, args=(), daemon=True).start()
        return self

    def update(self):
        import torch
        while True:
            if not self.Q.full():
                grabbed, frame = self.stream.read()
                if not grabbed:
                    self.stop()
                    return
                img, orig_img, dim = prep_frame(frame, int(opt.inp_dim))
                self.Q.put((img, orig_img, im_to_torch(orig_img), torch.FloatTensor([dim]).repeat(1, 2)))
            else:
                with self.Q.mutex:
                    self.Q.queue.clear()

    def videoinfo(self):
        return (self.stream.fourcc, self.stream.fps, self.stream.frame_size)

    def read(self):
        return self.Q.get()

    def len(self):
        return self.Q.qsize()

    def stop(self):
        self.stopped = True


# ----------------------------------------------------------------------------------------------- visualisation
def vis_frame(frame, im_res, format='coco'):
    """fn.py:144-220 (commented out in the reference): draw every result's box and key points on a BGR frame and
    return the annotated BGR u8 image.  ``im_res`` = one entry of ``DataWriter.results()``; the human-skeleton limb
    table of the original does not apply to 50 object key points, so points are drawn score-coloured without limbs."""
    from PIL import Image, ImageDraw
    img = Image.fromarray(np.ascontiguousarray(np.asarray(frame)[:, :, ::-1]))
    draw = ImageDraw.Draw(img)
    for human in im_res.get('result', []):
        kp = np.asarray(human['keypoints'], dtype=np.float64)
        sc = np.asarray(human['kp_score'], dtype=np.float64).reshape(-1)
        if 'bbox' in human:
            x1, y1, x2, y2 = [float(v) for v in np.asarray(human['bbox']).reshape(-1)[:4]]
            draw.rectangle([x1, y1, x2, y2], outline=(0, 255, 0))
        for (x, y), s in zip(kp, sc):
            if s <= 0.05:                                   # fn.py:175
                continue
            c = int(max(0.0, min(1.0, s)) * 255)
            draw.ellipse([x - 2, y - 2, x + 2, y + 2], fill=(255 - c, c, 64))
    return np.ascontiguousarray(np.asarray(img)[:, :, ::-1])


vis_frame_fast = vis_frame
