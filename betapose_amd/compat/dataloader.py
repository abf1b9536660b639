"""``from dataloader import ImageLoader, DetectionLoader, DetectionProcessor, DataWriter, Mscoco, crop_from_dets``."""
from betapose_amd.dataloader import (DataWriter, DetectionLoader, DetectionProcessor, ImageLoader, Mscoco,  # noqa: F401
                                     crop_from_dets)
from betapose_amd.video import VideoDetectionLoader, VideoLoader, WebcamLoader  # noqa: F401
