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
        with open(path, "rb") as fh:
            data = fh.read()
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
    frame, idx1 index) that ``FrameSource`` and ordinary players read back.  Frames are appended to the file as they
    arrive (headers of fixed size are written first and patched in ``release``); only the 16-byte index entries stay
    in memory."""

    def __init__(self, path, fps=25, frame_size=(640, 480), quality=90):
        self.path, self.fps, self.size, self.quality = path, float(fps), tuple(frame_size), quality
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        self._fh = open(path, "wb")
        self._index: List[Tuple[int, int]] = []       # (offset inside movi, length)
        self._fh.write(self._header(0, 0))            # place-holder of the final size
        self._movi_at = self._fh.tell()               # first chunk goes here; offsets in idx1 count from the 'movi' tag
        self._cursor = 4

    @staticmethod
    def _chunk(cid, body):
        return cid + struct.pack("<I", len(body)) + body + (b"\0" if len(body) & 1 else b"")

    @staticmethod
    def _list(kind, body_len_or_body):
        if isinstance(body_len_or_body, int):
            return b"LIST" + struct.pack("<I", body_len_or_body + 4) + kind
        return b"LIST" + struct.pack("<I", len(body_len_or_body) + 4) + kind + body_len_or_body

    def _header(self, n, movi_bytes):
        w, h = self.size
        avih = struct.pack("<IIIIIIIIII4I", int(1e6 / self.fps), 0, 0, 0x10, n, 0, 1, 0, w, h, 0, 0, 0, 0)
        strh = b"vids" + b"MJPG" + struct.pack("<IHHIIIIIIII4H", 0, 0, 0, 0, 1, int(self.fps), 0, n, 0, 0xFFFFFFFF, 0, 0, 0, w, h)
        strf = struct.pack("<IiiHH4sIiiII", 40, w, h, 1, 24, b"MJPG", w * h * 3, 0, 0, 0, 0)
        hdrl = self._list(b"hdrl", self._chunk(b"avih", avih) + self._list(b"strl", self._chunk(b"strh", strh) + self._chunk(b"strf", strf)))
        idx_bytes = 8 + 16 * n
        riff_len = 4 + len(hdrl) + 12 + movi_bytes + idx_bytes
        return b"RIFF" + struct.pack("<I", riff_len) + b"AVI " + hdrl + self._list(b"movi", movi_bytes)

    def isOpened(self):
        return self._fh is not None

    def write(self, frame_bgr):
        from PIL import Image
        buf = io.BytesIO()
        Image.fromarray(np.ascontiguousarray(frame_bgr[:, :, ::-1])).save(buf, format="JPEG", quality=self.quality)
        jpg = buf.getvalue()
        self._fh.write(self._chunk(b"00dc", jpg))
        self._index.append((self._cursor, len(jpg)))
        self._cursor += 8 + len(jpg) + (len(jpg) & 1)

    def release(self):
        if self._fh is None:
            return
        idx = b"".join(b"00dc" + struct.pack("<III", 0x10, off, ln) for off, ln in self._index)
        self._fh.write(self._chunk(b"idx1", idx))
        self._fh.seek(0)
        self._fh.write(self._header(len(self._index), self._cursor - 4))
        self._fh.close()
        self._fh = None


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
    canvas = np.full((inp_dim[1], inp_dim[0], 3), 128, dtype=np.uint8)
    top, left = (h - new_h) // 2, (w - new_w) // 2
    canvas[top:top + new_h, left:left + new_w, :] = resized
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
def _frame_batches(stream, datalen, batchSize):
    """Frames of ``stream`` in lists of up to ``batchSize`` with the index of each list's first frame; a short read ends
    the iteration after yielding (first index, frames read so far, False)."""
    for first in range(0, datalen, batchSize):
        frames = []
        for _ in range(first, min(first + batchSize, datalen)):
            grabbed, frame = stream.read()
            if not grabbed:
                yield first, frames, False
                return
            frames.append(frame)
        yield first, frames, True


def _prep_frames(frames):
    """``prep_frame`` over a list: (tensor [n,3,D,D], the frames, [n,4] (w, h, w, h))."""
    import torch
    tensors, dims = [], []
    for f in frames:
        t, _, wh = prep_frame(f, int(opt.inp_dim))
        tensors.append(t)
        dims.append(wh)
    return torch.cat(tensors), torch.FloatTensor(dims).repeat(1, 2)


class _VideoStage:
    """What the three video loaders share: a frame source, a bounded queue and a worker thread running ``update``."""

    def __init__(self, source, queue):
        self.stream = source if isinstance(source, FrameSource) else FrameSource(source)
        if not self.stream.isOpened():
            raise AssertionError('Cannot capture source')
        self.stopped = False
        self.Q = queue

    def start(self):
        Thread(target=self.update, daemon=True).start()
        return self

    def videoinfo(self):
        return (self.stream.fourcc, self.stream.fps, self.stream.frame_size)

    def len(self):
        return self.Q.qsize()

    def stop(self):
        self.stopped = True
        self.stream.release()


class VideoLoader(_VideoStage):
    """dataloader.py:192-282: batches (img, orig_img, im_name, im_dim_list) of letterboxed frames from a video source;
    frame k is named '<k>.jpg'."""

    def __init__(self, path, batchSize=1, queueSize=50):
        super().__init__(path, Queue(maxsize=queueSize))
        self.path = path
        self.batchSize = batchSize
        self.datalen = int(self.stream.frame_count)
        self.num_batches = -(-self.datalen // batchSize)

    def length(self):
        return self.datalen

    def update(self):
        for first, frames, complete in _frame_batches(self.stream, self.datalen, self.batchSize):
            if not complete:
                self.Q.put((None, None, None, None))
                print('===========================> This video get %d frames in total.' % (first + len(frames)), flush=True)
                break
            img, dims = _prep_frames(frames)
            self.Q.put((img, frames, ['%d.jpg' % (first + j) for j in range(len(frames))], dims))
        self.stream.release()

    def getitem(self):
        return self.Q.get()


def unletterbox_boxes(dets, im_dim_list, det_inp_dim):
    """dataloader.py:548-560: detections (frame index, x1, y1, x2, y2, ...) from the letterboxed detector input back to
    frame pixels: remove the grey border, divide by the letterbox scale, clip to the frame.  Vectorised over boxes;
    returns a new tensor."""
    import torch
    wh = im_dim_list.index_select(0, dets[:, 0].long())[:, :2]            # [n, 2] frame (w, h)
    scale = (float(det_inp_dim) / wh).min(dim=1, keepdim=True)[0]        # [n, 1]
    border = (float(det_inp_dim) - scale * wh) / 2                       # [n, 2] grey border (x, y)
    out = dets.clone()
    xyxy = (dets[:, 1:5] - border.repeat(1, 2)) / scale
    out[:, 1:5] = torch.minimum(xyxy.clamp(min=0.0), wh.repeat(1, 2))
    return out


_letterbox_boxes = unletterbox_boxes      # (the name round 2 used)


class VideoDetectionLoader(_VideoStage):
    """dataloader.py:468-591: video frames -> detector -> per-frame (inp, orig_img, boxes, scores).  The reference
    hard-codes AlphaPose's person detector (yolov3-spp, NMS on); here the object detector of this path is passed in (or
    built from ``models/yolo/<obj>.weights``) and ``dynamic_write_results`` keeps its one box per frame."""

    def __init__(self, path, batchSize=4, queueSize=256, det_model=None, obj_id=None):
        from .dataloader import _default_detector
        from .yolo_util import dynamic_write_results
        super().__init__(path, Queue(maxsize=queueSize))
        self._write_results = dynamic_write_results
        if det_model is None:
            det_model = _default_detector(int(obj_id if obj_id is not None else opt.obj_id), batchSize)
        self.det_model = det_model
        self.det_model.net_info['height'] = opt.inp_dim
        self.det_inp_dim = int(opt.inp_dim)
        if self.det_inp_dim % 32 != 0 or self.det_inp_dim <= 32:
            raise AssertionError("detector input size must be a multiple of 32 and larger than 32")
        self.det_model.cuda().eval()
        self.batchSize = batchSize
        self.datalen = int(self.stream.frame_count)
        self.num_batches = -(-self.datalen // batchSize)

    def length(self):
        return self.datalen

    def update(self):
        import torch
        for _, frames, complete in _frame_batches(self.stream, self.datalen, self.batchSize):
            if not complete:
                self.stop()
                return
            img, dims = _prep_frames(frames)
            inps = [im_to_torch(f) for f in frames]
            with torch.no_grad():
                dets = self._write_results(self.det_model(img).cpu(), opt.confidence, opt.num_classes, nms=True,
                                           nms_conf=opt.nms_thesh)
            found = not isinstance(dets, int) and dets.shape[0] > 0
            if found:
                dets = unletterbox_boxes(dets, dims, self.det_inp_dim)
            for k, (inp, frame) in enumerate(zip(inps, frames)):
                if not found:
                    self.Q.put((inp, frame, None, None))
                    continue
                mine = dets[:, 0] == k
                self.Q.put((inp, frame, dets[mine, 1:5], dets[mine, 5:6]))

    def read(self):
        return self.Q.get()

    def more(self):
        return self.Q.qsize() > 0


class WebcamLoader(_VideoStage):
    """dataloader.py:594-647: newest-frame-first (LIFO) queue of letterboxed frames.  ``webcam``: a camera index in the
    reference; here anything ``FrameSource`` opens (a growing frame directory, an MJPEG stream) -- a bare index raises
    because no capture library is present."""

    def __init__(self, webcam, queueSize=256):
        if isinstance(webcam, int) or (isinstance(webcam, str) and webcam.isdigit()):
            raise IOError("Cannot capture source: camera index %s needs a capture library (V4L2 / OpenCV) this image "
                          "does not have; pass a frame directory or an MJPEG stream" % webcam)
        super().__init__(webcam, LifoQueue(maxsize=queueSize))

    def update(self):
        import torch
        while not self.stopped:
            if self.Q.full():                      # the consumer fell behind: drop the backlog, keep only fresh frames
                with self.Q.mutex:
                    self.Q.queue.clear()
                continue
            grabbed, frame = self.stream.read()
            if not grabbed:
                self.stop()
                break
            img, _, dim = prep_frame(frame, int(opt.inp_dim))
            self.Q.put((img, frame, im_to_torch(frame), torch.FloatTensor([dim]).repeat(1, 2)))

    def read(self):
        return self.Q.get()


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
            draw.rectangle([min(x1, x2), min(y1, y2), max(x1, x2), max(y1, y2)], outline=(0, 255, 0))
        for (x, y), s in zip(kp, sc):
            if s <= 0.05:                                   # fn.py:175
                continue
            c = int(max(0.0, min(1.0, s)) * 255)
            draw.ellipse([x - 2, y - 2, x + 2, y + 2], fill=(255 - c, c, 64))
    return np.ascontiguousarray(np.asarray(img)[:, :, ::-1])


vis_frame_fast = vis_frame
